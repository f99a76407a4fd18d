// casadi_abi.cpp -- CasADi external-function drop-in (include/landing_casadi_abi.h).
// Host-side shim only: every evaluation is one launch of the HIP sweep kernel (batch of one)
// through the C ABI of liblanding_mi355x.so.  Mirrors landingCtrller_IPOPT.c:10916-10993 et seq.
#include <cstdlib>
#include <mutex>
#include <vector>

#include "../../include/landing_casadi_abi.h"
#include "../../include/landing_nlp.h"

#ifndef LANDING_N
#define LANDING_N 20
#endif
// LANDING_CCC = 1: the NLP of the reference's N=41 script (generate_quadruped_SRBM_CCC.m; its generated library would be
// codegen_casadi/nlp_quad_SRBM.so, :340-341 -- absent from the reference tree): kin-box .05/.05/.27 (:169-171), running cost (:81-89), the
// script's own parameter vector (np = 37N+112: Uref, QX, Qc, Qf are parameters), Hessian in the extended pattern landing_pattern_hess_rc
#ifndef LANDING_CCC
#define LANDING_CCC 0
#endif
// LANDING_KD = 1: the kinodynamic refinement NLP (generate_solver/generate_landingCtrller_KNITRO.m:360-377 generates and loads ./landingCtrller_KNITRO.so;
// a missing blob of the reference tree): x = [X; jpos; U] (48 N + 12), p = the 13 N + 113 active Opti parameters, g in Opti's canonical form, patterns and
// values from landing_kinodyn_casadi_* (include/landing_nlp.h); the context carries the reference's 'quad3D' / 'mc3D' model (landing_rbd_model_mc3d)
#ifndef LANDING_KD
#define LANDING_KD 0
#endif

namespace {
const int N = LANDING_N;
std::mutex g_mu;
landing_ctx* g_ctx = nullptr;
int g_refs = 0;
std::vector<long long> s_x, s_p, s_one, s_g, s_hess, s_jac;
std::vector<double> zeros;

void dense(std::vector<long long>& s, long long n) {
  s.clear(); s.push_back(n); s.push_back(1); s.push_back(0); s.push_back(n);
  for (long long i = 0; i < n; ++i) s.push_back(i);
}
#if LANDING_KD
landing_ctx* g_kd_ctx = nullptr;
landing_ctx* kd_ctx_locked() {      // (g_mu held) the kinodynamic patterns come from the derivative kernels themselves: the context is needed before the tables
  if (!g_kd_ctx) {
    const char* d = std::getenv("LANDING_DEVICE");
    g_kd_ctx = landing_create(N, d ? std::atoi(d) : 0, nullptr);
    if (g_kd_ctx) { landing_rbd_model m; landing_rbd_model_mc3d(&m); if (landing_rbd_set_model(g_kd_ctx, &m)) { landing_destroy(g_kd_ctx); g_kd_ctx = nullptr; } }
  }
  return g_kd_ctx;
}
void build_sparsity() {
  if (!s_x.empty()) return;
  long long nx = 0, ng = 0;
  landing_kinodyn_nlp_dims(N, &nx, &ng);
  const long long np = landing_kinodyn_casadi_np(N);
  landing_ctx* c = kd_ctx_locked();
  if (!c) return;
  std::vector<long long> sj, sh;
  for (int which = 0; which < 2; ++which) {
    const long long *ci = nullptr, *r = nullptr; long long nnz = 0;
    if (landing_kinodyn_casadi_pattern(c, N, which, &ci, &r, &nnz)) return;
    std::vector<long long>& s = which ? sh : sj;
    s.push_back(which ? nx : ng); s.push_back(nx);
    s.insert(s.end(), ci, ci + nx + 1); s.insert(s.end(), r, r + nnz);
  }
  dense(s_x, nx); dense(s_p, np); dense(s_one, 1); dense(s_g, ng);
  s_jac = sj; s_hess = sh;
  zeros.assign((size_t)std::max(std::max(nx, ng), np), 0.0);
}
#else
void build_sparsity() {
  if (!s_x.empty()) return;
  const long long nx = landing_nx(N), ng = landing_ng(N), np = LANDING_CCC ? landing_np_ccc(N) : landing_np(N);
  dense(s_x, nx); dense(s_p, np); dense(s_one, 1); dense(s_g, ng);
  std::vector<long long> ci(nx + 1), r(landing_nnz_jac(N));
  landing_pattern_jac(N, ci.data(), r.data());
  s_jac.clear(); s_jac.push_back(ng); s_jac.push_back(nx);
  s_jac.insert(s_jac.end(), ci.begin(), ci.end()); s_jac.insert(s_jac.end(), r.begin(), r.end());
  r.resize(LANDING_CCC ? landing_nnz_hess_rc(N) : landing_nnz_hess(N));
  if (LANDING_CCC) landing_pattern_hess_rc(N, ci.data(), r.data()); else landing_pattern_hess(N, ci.data(), r.data());
  s_hess.clear(); s_hess.push_back(nx); s_hess.push_back(nx);
  s_hess.insert(s_hess.end(), ci.begin(), ci.end()); s_hess.insert(s_hess.end(), r.begin(), r.end());
  zeros.assign((size_t)std::max(std::max(nx, ng), np), 0.0);
}
#endif
landing_ctx* ctx() {
  std::lock_guard<std::mutex> lk(g_mu);
  build_sparsity();
#if LANDING_KD
  g_ctx = kd_ctx_locked();
#endif
  if (!g_ctx) {
    const char* d = std::getenv("LANDING_DEVICE");
    landing_form form;
    landing_form_default(&form);
    if (LANDING_CCC) { form.kin_box[0] = 0.05; form.kin_box[1] = 0.05; form.kin_box[2] = 0.27; form.run_cost = 2; }
    g_ctx = landing_create(N, d ? std::atoi(d) : 0, &form);
  }
  return g_ctx;
}
// arg[i] == NULL means "all zeros" (landingCtrller_IPOPT.c:69-70).  The zero vector is part of the lazily built tables, so
// they are built HERE, before the pointer is taken (the arguments of eval() are evaluated before eval() itself runs).
const double* in(const double** arg, int i) {
  if (arg && arg[i]) return arg[i];
  { std::lock_guard<std::mutex> lk(g_mu); build_sparsity(); }
  return zeros.data();
}

// one evaluation; which outputs are wanted is decided by the caller's res[] pointers
int eval(const double* x, const double* p, const double* lam_f, const double* lam_g, double* f, double* g,
         double* grad_f, double* jac, double* hess, double* ggx, double* ggp) {
  landing_ctx* c = ctx();
  if (!c) return 1;
  if (!f && !g && !grad_f && !jac && !hess && !ggx && !ggp) return 0;
#if LANDING_KD
  return landing_kinodyn_casadi_eval_host(c, N, x, p, lam_f, lam_g, f, g, grad_f, jac, hess, ggx, ggp) == 0 ? 0 : 1;
#endif
  if (LANDING_CCC && hess) {      // the running cost adds diagonals casadi_s4 does not hold: extended pattern, own entry point
    if (landing_eval_hess_rc_batch_host(c, 1, x, p, lam_f, lam_g, hess) != 0) return 1;
    hess = nullptr;
    if (!f && !g && !grad_f && !jac && !ggx && !ggp) return 0;
  }
  return landing_eval_batch_host(c, 1, x, p, lam_f, lam_g, f, g, grad_f, jac, hess, ggx, ggp) == 0 ? 0 : 1;
}
void addref() { std::lock_guard<std::mutex> lk(g_mu); ++g_refs; }
void dropref() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (--g_refs <= 0 && g_ctx) {
#if LANDING_KD
    landing_kinodyn_casadi_release(g_ctx); g_kd_ctx = nullptr; s_x.clear();      // (the pattern tables point into the context's cache)
#endif
    landing_destroy(g_ctx); g_ctx = nullptr; g_refs = 0;
  }
}
const long long* sp(int which) {
  { std::lock_guard<std::mutex> lk(g_mu); build_sparsity(); }
  switch (which) { case 0: return s_x.data(); case 1: return s_p.data(); case 2: return s_one.data(); case 3: return s_g.data();
                   case 4: return s_hess.data(); case 5: return s_jac.data(); default: return nullptr; }
}
}  // namespace

#define META(F, NIN, NOUT, NAMES_IN, NAMES_OUT, SP_IN, SP_OUT)                                   \
  int F##_alloc_mem(void) { return 0; }                                                           \
  int F##_init_mem(int) { return 0; }                                                             \
  void F##_free_mem(int) {}                                                                       \
  int F##_checkout(void) { return 0; }                                                            \
  void F##_release(int) {}                                                                        \
  void F##_incref(void) { addref(); }                                                             \
  void F##_decref(void) { dropref(); }                                                            \
  long long F##_n_in(void) { return NIN; }                                                        \
  long long F##_n_out(void) { return NOUT; }                                                      \
  double F##_default_in(long long) { return 0; }                                                  \
  const char* F##_name_in(long long i) { static const char* n[] = NAMES_IN; return (i >= 0 && i < NIN) ? n[i] : 0; }   \
  const char* F##_name_out(long long i) { static const char* n[] = NAMES_OUT; return (i >= 0 && i < NOUT) ? n[i] : 0; } \
  const long long* F##_sparsity_in(long long i) { static const int s[] = SP_IN; return (i >= 0 && i < NIN) ? sp(s[i]) : 0; }   \
  const long long* F##_sparsity_out(long long i) { static const int s[] = SP_OUT; return (i >= 0 && i < NOUT) ? sp(s[i]) : 0; } \
  int F##_work(long long* a, long long* r, long long* iw, long long* w) {                         \
    if (a) *a = NIN; if (r) *r = NOUT; if (iw) *iw = 0; if (w) *w = 0; return 0; }

#define L(...) {__VA_ARGS__}

extern "C" {
// nlp:(x,p)->(f,g)   landingCtrller_IPOPT.c:66
int nlp(const double** arg, double** res, long long*, double*, int) {
  return eval(in(arg, 0), in(arg, 1), nullptr, nullptr, res[0], res[1], nullptr, nullptr, nullptr, nullptr, nullptr);
}
META(nlp, 2, 2, L("x", "p"), L("f", "g"), L(0, 1), L(2, 3))
// nlp_f:(x,p)->(f)   :10994
int nlp_f(const double** arg, double** res, long long*, double*, int) {
  return eval(in(arg, 0), in(arg, 1), nullptr, nullptr, res[0], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
META(nlp_f, 2, 1, L("x", "p"), L("f"), L(0, 1), L(2))
// nlp_g:(x,p)->(g)   :11160
int nlp_g(const double** arg, double** res, long long*, double*, int) {
  return eval(in(arg, 0), in(arg, 1), nullptr, nullptr, nullptr, res[0], nullptr, nullptr, nullptr, nullptr, nullptr);
}
META(nlp_g, 2, 1, L("x", "p"), L("g"), L(0, 1), L(3))
// nlp_grad:(x,p,lam_f,lam_g)->(f,g,grad_gamma_x,grad_gamma_p)   :22014
int nlp_grad(const double** arg, double** res, long long*, double*, int) {
  return eval(in(arg, 0), in(arg, 1), in(arg, 2), in(arg, 3), res[0], res[1], nullptr, nullptr, nullptr, res[2], res[3]);
}
META(nlp_grad, 4, 4, L("x", "p", "lam_f", "lam_g"), L("f", "g", "grad_gamma_x", "grad_gamma_p"), L(0, 1, 2, 3), L(2, 3, 0, 1))
// nlp_grad_f:(x,p)->(f,grad_f_x)   :52601
int nlp_grad_f(const double** arg, double** res, long long*, double*, int) {
  return eval(in(arg, 0), in(arg, 1), nullptr, nullptr, res[0], nullptr, res[1], nullptr, nullptr, nullptr, nullptr);
}
META(nlp_grad_f, 2, 2, L("x", "p"), L("f", "grad_f_x"), L(0, 1), L(2, 0))
// nlp_hess_l:(x,p,lam_f,lam_g)->(hess_gamma_x_x)   :53526
int nlp_hess_l(const double** arg, double** res, long long*, double*, int) {
  return eval(in(arg, 0), in(arg, 1), in(arg, 2), in(arg, 3), nullptr, nullptr, nullptr, nullptr, res[0], nullptr, nullptr);
}
META(nlp_hess_l, 4, 1, L("x", "p", "lam_f", "lam_g"), L("hess_gamma_x_x"), L(0, 1, 2, 3), L(4))
// nlp_jac_g:(x,p)->(g,jac_g_x)   :94013
int nlp_jac_g(const double** arg, double** res, long long*, double*, int) {
  return eval(in(arg, 0), in(arg, 1), nullptr, nullptr, nullptr, res[0], nullptr, res[1], nullptr, nullptr, nullptr);
}
META(nlp_jac_g, 2, 2, L("x", "p"), L("g", "jac_g_x"), L(0, 1), L(3, 5))
}  // extern "C"
