/*
 * landing_casadi_abi.h -- the CasADi external-function ABI exported by the drop-in libraries
 *   landingCtrller_IPOPT_mi355x.so      (N = 20 intervals: replaces the reference's
 *                                        optimizations/landing/codegen_casadi/landingCtrller_IPOPT.so)
 *   landingCtrller_IPOPT_N40_mi355x.so  (N = 40 intervals)
 * Symbol set and semantics are those of the reference's generated C
 * (landingCtrller_IPOPT.c:10916-10993 for `nlp`, the same block after every function):
 *
 *   for F in { nlp, nlp_f, nlp_g, nlp_grad, nlp_grad_f, nlp_hess_l, nlp_jac_g }:
 *     int F(const double** arg, double** res, long long* iw, double* w, int mem);
 *     int F_alloc_mem(void); int F_init_mem(int); void F_free_mem(int);
 *     int F_checkout(void);  void F_release(int); void F_incref(void); void F_decref(void);
 *     long long F_n_in(void); long long F_n_out(void); double F_default_in(long long);
 *     const char* F_name_in(long long); const char* F_name_out(long long);
 *     const long long* F_sparsity_in(long long); const long long* F_sparsity_out(long long);
 *     int F_work(long long* sz_arg, long long* sz_res, long long* sz_iw, long long* sz_w);
 *
 * arg[i]==NULL is read as zeros, res[i]==NULL is skipped (landingCtrller_IPOPT.c:69-70,11163);
 * every evaluation runs the HIP kernels of liblanding_mi355x.so with a batch of one (return 0 on
 * success, 1 on failure -- CasADi then reports an evaluation error, oracle_function.cpp:218-225).
 * CasADi loads it with external("nlp", path) (casadi/core/nlpsol.cpp:100-108).
 */
#ifndef LANDING_CASADI_ABI_H
#define LANDING_CASADI_ABI_H
#ifdef __cplusplus
extern "C" {
#endif
#define LANDING_CASADI_DECL(F)                                                              \
  int F(const double** arg, double** res, long long* iw, double* w, int mem);               \
  int F##_alloc_mem(void); int F##_init_mem(int mem); void F##_free_mem(int mem);           \
  int F##_checkout(void); void F##_release(int mem); void F##_incref(void); void F##_decref(void); \
  long long F##_n_in(void); long long F##_n_out(void); double F##_default_in(long long i);  \
  const char* F##_name_in(long long i); const char* F##_name_out(long long i);              \
  const long long* F##_sparsity_in(long long i); const long long* F##_sparsity_out(long long i); \
  int F##_work(long long* sz_arg, long long* sz_res, long long* sz_iw, long long* sz_w);
LANDING_CASADI_DECL(nlp)
LANDING_CASADI_DECL(nlp_f)
LANDING_CASADI_DECL(nlp_g)
LANDING_CASADI_DECL(nlp_grad)
LANDING_CASADI_DECL(nlp_grad_f)
LANDING_CASADI_DECL(nlp_hess_l)
LANDING_CASADI_DECL(nlp_jac_g)
#ifdef __cplusplus
}
#endif
#endif
