// solver_kernels.hip -- batched primal-dual interior-point solver for the SRBM landing NLP (gfx950).
//
// ONE WAVEFRONT = ONE NLP, from the initial guess to the KKT point, in a single persistent kernel:
// members never wait for each other (no lock-step batch iterations), the hardware workgroup
// dispatcher is the work queue.  What replaces the reference's CasADi Nlpsol('ipopt') + MA57 path
// (generate_landingCtrller_IPOPT.m:231-264,277,314; casadi/core/nlpsol.cpp:560-640):
//   * the NLP IPOPT sees through the CasADi boundary: every bound lives in g (lbx/ubx = +-inf), so
//     every inequality row gets a slack and a pair of bound multipliers;
//   * monotone barrier update, fraction-to-the-boundary rule, filter line search (Waechter & Biegler);
//   * the KKT system is condensed stage-wise and solved by a Riccati recursion with state
//     (X_k, c_k) and control (f_k, c_{k+1}) -- the no-slip rows couple U_k and U_{k+1}, carrying the
//     feet as state restores the optimal-control sparsity; inertia correction = retry with a larger
//     delta_w when a stage Cholesky meets a non-positive pivot (IPOPT's rule of thumb).
// Function values / derivatives come from the same srbm_stage.hpp code as the function layer, in the
// reference's CCS order; the assembly into stage blocks is table driven (solver_capi.inc).
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/landing_nlp.h"

// inlining policy of the solver's phase functions (development switches; the defaults are what the product build uses).
// Measured round 3, A/B on one box: with every phase inlined into landing_ipm_kernel the callee-saved-register traffic disappears
// (209 -> 192 GB of HBM traffic per launch) but the monolithic kernel is allocated and scheduled far worse -- condensation 0.104 ->
// 0.231 ms, forward sweep 0.082 -> 0.146 ms per iteration under load, 96 -> 141 ms per batch.  Out of line it is.
#ifndef LANDING_INL_COND
#define LANDING_INL_COND __noinline__
#endif
#ifndef LANDING_INL_BACK
#define LANDING_INL_BACK __noinline__
#endif
#ifndef LANDING_INL_FWD
#define LANDING_INL_FWD __noinline__
#endif
#ifndef LANDING_INL_ROWP
#define LANDING_INL_ROWP __noinline__
#endif

namespace landing {

constexpr int NZ_JX = 0, NZ_JU = 157, NZ_JUN = 385, NZ_HX = 613, NZ_HU = 642, NZ_HUN = 802, NZ_TOT = 962;
constexpr int GS = 49;    // LDS row stride of G (48x48)
constexpr int PS = 25;    // LDS row stride of 24x24 matrices
constexpr int YS = 37;    // LDS row stride of 24x36 / 12x36 matrices
// per-stage Riccati record (doubles): K 24x24 | kappa 24 | A^ 12x36 | b 12 | P_k rows of X (12x24) | p_k X part 12
// stage record: gains K (nu x 24) | kappa (24) | closed-loop state map of the forward sweep X+ = Mt sigma + mv (12 x 24 | 12) |
// state rows of the cost-to-go P (12 x 24) | p (12)
constexpr int RIC_K = 0, RIC_KAP = 576, RIC_MT = 600, RIC_MV = 888, RIC_PX = 900, RIC_PV = 1188, RIC_STRIDE = 1200;
constexpr int RIC_FWD0 = 288, RIC_FWDN = 612;   // forward chain reads rec[288, 900): K rows of c+ | kappa | Mt | mv
constexpr int FILT_CAP = 64;
constexpr int SOLVER_NMAX = 96;   // longest horizon the solver kernel takes: sigma_0..sigma_N of the forward sweep live in the 48 x 49 LDS array, one table row per stage
constexpr int ES = 26;    // LDS row stride of the elimination side block [gamma_u | I] (24 x 25)
constexpr int SOLVER_THREADS = 256;
// condensed stage data (LDS staging area of the backward sweep): [G targets (table order) | gamma 48 | A^ values] of one stage
constexpr int COND_GAM = 480, COND_AH = 528, COND_STRIDE = 704;
constexpr int RCG = 36;   // per stage in the member's workspace: gradient of the running cost w.r.t. (X_k, c_k, f_k) (forms with a running cost)
// constant Hessian entries of the running cost per stage, stored right behind the Hessian nonzeros so that the
// condensation tables address them like any other entry of [J | H]: X diagonal (12) | (pos_a, c_leg,a) (12) | c diagonal (12) | f diagonal (12)
constexpr int RUNC = 48;

// Packed condensation term (8 bytes), stage-local: value = JH[a] * (has_b ? JH[b] : 1) * coeff, coeff = sigma[row] (has_b) /
// rho[row] (!has_b) for rtype 0, +1 / -1 / 0 for rtype 1 / 2 / 3; summed into the open destination, stored to
// stg[dst] when `closes`.  a, b = positions in the stage's LDS copy of its nonzeros (segment sa at NZ_*[sa], offset oa inside):
// segments 0..6 = J X_k | J U_k | J U_{k+1} | H X_k | H U_k | H U_{k+1} | running-cost constants of stage k.
constexpr int CTAB_MLMAX = 6;
__host__ __device__ inline unsigned long long cterm_pack(int sa, int oa, int sb, int ob, bool has_b, int rowq, int rtype, int dst, bool closes) {
  const int nz[7] = {NZ_JX, NZ_JU, NZ_JUN, NZ_HX, NZ_HU, NZ_HUN, NZ_TOT};
  const unsigned pa = (unsigned)(nz[sa] + oa), pb = (unsigned)(nz[sb] + ob);
  const unsigned lo = pa | (pb << 11) | ((unsigned)(has_b ? 1 : 0) << 22) | ((unsigned)rowq << 23) | ((unsigned)rtype << 30);
  const unsigned hi = (unsigned)dst | ((unsigned)(closes ? 1 : 0) << 10);
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__host__ __device__ inline bool cterm_closes(unsigned long long t) { return ((t >> 42) & 1ull) != 0; }
// Assembly of a stage inside the backward sweep (asm_terms): operands of the condensation terms are positions in the LDS array
// cx = [nonzeros of the stage (NZ_TOT + RUNC) | sigma (104) | rho (104) | 1, -1, 0]
constexpr int CX_SR = NZ_TOT + 48, CX_ONE = CX_SR + 208, CX_MONE = CX_ONE + 1, CX_ZERO = CX_ONE + 2, CX_LEN = CX_ONE + 3;
constexpr int ATAB_TB = 6, ATAB_LTMAX = 6;       // terms per batch; longest per-thread list (a multiple of ATAB_TB): the 1 138..1 234 terms of a stage give 5 per thread
constexpr int ASM_NSLOT = 512;                   // partial-sum slots of one stage

struct SolverWorkspace {
  double* buf = nullptr; size_t cap = 0;
  int* d_order = nullptr; int order_cap = 0;
  hipEvent_t done = nullptr;     // recorded behind every solve launch: the next launch (any stream) and any re-allocation wait for it
  int* d_tab = nullptr; int* d_stage_tab = nullptr; int n_tab = 0;
  int* d_rterm = nullptr; int rlen = 0;
  int *d_ctab = nullptr, *d_ctype = nullptr, *d_ccomb = nullptr; int c_ml = 0, c_mid = 0;     // packed per-stage-type assembly tables (aterm_pack): [type][c_ml][256], most frequent type
  static size_t member_stride(const Layout& L) {
    return (size_t)4 * L.nx + (size_t)14 * L.ng + L.nnz_jac + L.nnz_hess + (size_t)L.N * RUNC + (size_t)(L.N + 1) * RIC_STRIDE + (size_t)L.N * RCG;
  }
  int ensure(const Layout& L, int B, hipStream_t stream);
  void release();
};

// optional per-member phase timers (wall_clock64 ticks, 100 MHz) -- enabled when SolveArgs.prof != nullptr
enum { PH_EVAL = 0, PH_ERR, PH_SIGRHO, PH_BACK, PH_FWD, PH_DUAL, PH_LS, PH_ACCEPT, PH_NFACT, PH_NTRIAL, PH_NITER, PH_NSTAGE_OK, PH_B_ASM, PH_NSTAGE, PH_B_ELIM, PH_B_POST,
#ifdef LANDING_STAGE_PROF      // development build (tools/dev/stage_prof.py): time of wave 0 between marks inside block_eliminate, slots 16..
       PH_COUNT = 32 };
#else
       PH_COUNT = 16 };
#endif   // 11 / 13: stage eliminations that succeeded / were attempted (stage-0 foot block included)
#define PROF_ADD(slot, tstart) do { if (SH.prof_on) { const long long n_ = (long long)wall_clock64(); if (threadIdx.x == 0) { SH.prof[slot] += (double)(n_ - (tstart)); (tstart) = n_; } } } while (0)

struct SolveArgs {
  Layout L; int B; landing_solver_opts o; double* prof;
  const double* p; const double* x0;
  double* x_out; double* f_out; double* lam_out; int* status; int* iters; double* kkt;
  double* ws; size_t ws_stride;
  const unsigned long long* rterm; int rlen;                               // row-product records [rlen][256]
  const unsigned long long* ctab; const int* ctype; int c_ml, c_mid;      // packed assembly tables [type][c_ml][256] (aterm_pack), type of every stage
  const unsigned long long* ccomb;                                         // ... and the destinations summed from partial slots [type][256] (acomb_pack)
  const int* edge_map;   // compaction map of the tiled Jacobian write-out (landing_ctx::d_edge_map)
  const int* order;      // dispatch order: workgroup b solves member order[b] (hard-first, see landing_order_kernel); nullptr = identity
};

// ---- block-wide reductions through LDS (deterministic order), K values at once ----------------------
enum { RSUM = 0, RMAX = 1, RMIN = 2 };
__device__ __forceinline__ double red_op(double a, double b, int op) { return op == RSUM ? a + b : (op == RMAX ? fmax(a, b) : fmin(a, b)); }
template <int K>
__device__ __forceinline__ void block_reduce(double (&v)[K], const int (&op)[K], double* red) {
  const int tid = threadIdx.x, nwave = (blockDim.x + 63) >> 6;
  for (int i = 0; i < K; ++i) {
#pragma unroll
    for (int mask = 32; mask >= 1; mask >>= 1) v[i] = red_op(v[i], __shfl_xor(v[i], mask), op[i]);   // butterfly: every lane gets the wave result
  }
  if ((tid & 63) == 0) for (int i = 0; i < K; ++i) red[(tid >> 6) * K + i] = v[i];
  __syncthreads();
  for (int i = 0; i < K; ++i) {
    double r = red[i];
    for (int w = 1; w < nwave; ++w) r = red_op(r, red[w * K + i], op[i]);
    v[i] = r;
  }
  __syncthreads();
}
__device__ __forceinline__ double block_reduce1(double v, int op, double* red) {
  double a[1] = {v}; const int o[1] = {op};
  block_reduce<1>(a, o, red);
  return a[0];
}

// The four largest of a set of values spread over the block (clip_k rule of the primal step length): every thread keeps
// a sorted quadruple t[0] >= .. >= t[3]; the merge is a max-type operation, hence independent of the order.
__device__ __forceinline__ void top4_push(double (&t)[4], double v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double hi = fmax(t[i], v); v = fmin(t[i], v); t[i] = hi; }
}
__device__ __forceinline__ void block_top4(double (&t)[4], double* red) {
  const int tid = threadIdx.x, nwave = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int mask = 32; mask >= 1; mask >>= 1) {
    double o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = __shfl_xor(t[i], mask);
#pragma unroll
    for (int i = 0; i < 4; ++i) top4_push(t, o[i]);
  }
  if ((tid & 63) == 0) for (int i = 0; i < 4; ++i) red[(tid >> 6) * 4 + i] = t[i];
  __syncthreads();
  for (int i = 0; i < 4; ++i) t[i] = red[i];
  for (int w = 1; w < nwave; ++w) for (int i = 0; i < 4; ++i) top4_push(t, red[w * 4 + i]);
  __syncthreads();
}

struct MemberMem {
  double *x, *xt, *dx, *gx;
  double *g, *gt, *s, *ds, *zL, *zU, *y, *yn, *sig, *rho;
  double *J, *H, *Hc, *ric, *cond;
  double *en, *ep, *wn, *wp;      // feasibility phase: violation variables of the lower / upper side of every inequality row and their multipliers
};

__device__ __forceinline__ MemberMem carve(const Layout& L, double* w) {
  MemberMem M;
  M.x = w; w += L.nx; M.xt = w; w += L.nx; M.dx = w; w += L.nx; M.gx = w; w += L.nx;
  M.g = w; w += L.ng; M.gt = w; w += L.ng; M.s = w; w += L.ng; M.ds = w; w += L.ng;
  M.zL = w; w += L.ng; M.zU = w; w += L.ng;
  M.y = w; w += L.ng; M.yn = w; w += L.ng;
  M.sig = w; w += L.ng; M.rho = w; w += L.ng;
  M.J = w; w += L.nnz_jac; M.H = w; w += L.nnz_hess; M.Hc = w; w += (size_t)L.N * RUNC; M.ric = w; w += (size_t)(L.N + 1) * RIC_STRIDE; M.cond = w; w += (size_t)L.N * RCG;
  M.en = w; w += L.ng; M.ep = w; w += L.ng; M.wn = w; w += L.ng; M.wp = w;
  return M;
}

// Scalar state of the interior-point loop, kept in LDS (S.ks) and never in registers across phases.  Every thread reads it in
// place; ONLY thread 0 writes it, inside KS_BEGIN / KS_END blocks that are fenced by barriers on both sides -- so the scalar
// control logic of the solver (error test, restarts, barrier update, regularisation schedule, filter, watchdog) runs once per
// workgroup instead of 256 times, and no value is live across a phase.  Round 2 held these ~45 scalars in VGPRs: the kernel body
// alone carried 767 static scratch (private-segment) accesses around its calls, executed every iteration by all 256 threads
// (VERDICT r2 item 8); now 233, none of them on the per-iteration path of the row passes.
struct IpmState {
  double mu, delta_last, th_max, c_pr, c_co, c_cm, c_ys, c_zs, c_nz, e_pr, e_du, e_co;
  double tau, omt, a_pr, a_du, th0, ph0, dphi, alpha, s_corr, delta, ft;
  long long tp;
  int nfilt, it, status, need_reg_streak, nreset, last_reset_it, ncrawl, clip_k_cur, last_mu_it, cutstreak, wd_count, first_failed, force_step;
  int clip_now, accepted, armijo_step, fact_ok, skipped_zero, attempt;
  int jamrun, stag, full_prev;      // jam_clip / stag_relief (landing_nlp.h): iterations in a row with a tiny step to the boundary; full steps of the last barrier problem that did not halve the error
  double e_prev;
  int action, flag, fresh, ls_done, need_corr, fallback;
  // feasibility (restoration) phase, landing_nlp.h: 1 while the elastic problem is being solved; lim = iteration limit in force
  int feas, feas_used, lim, fact_failed;
  int fjam, fstat; double v1_ref;      // feas_jam / feas_stat (landing_nlp.h): leaky count of iterations with a tiny accepted step; iterations with a stationary violation, its reference value
  double c_rn, f_vmax, f_v1;      // |z + w - rho|_inf of the elastic rows; max-norm and 1-norm violation of the inequality rows at x
  // round 6 (landing_nlp.h: feas_max, feas_back, feas_resume, feas_polish): entries into the phase so far; 1 once a stalled phase has handed its point back (or the
  // line search jammed with no entry left); 1 once the regularisation was dropped at a stationary violation; violation at the entry point (1-norm, equality rows
  // included) and 1-norm residual of the equality rows at x; iteration count nothing runs beyond; 1: the entry pass of the phase has to record th_entry
  int n_feas, stalled, polished, hard_lim, want_entry; double th_entry, f_theq;
  double fdc;      // factor between the regularisation of the last iteration and the first one tried in this one inside the phase (feas_delta_dec, adapted: squared after an
                   // iteration whose first factorisation succeeded, square root (<= 0.7) after one that needed more)
};
enum { ACT_GO = 0, ACT_STOP = 1, ACT_RESET = 2, ACT_FEAS = 3, ACT_BACK = 4 };

// LDS of one member
constexpr int FWD_DXL_NMAX = 64;   // horizons up to this keep a copy of dx in LDS for row_products: 36 N + 12 doubles in [P | A1 | A^]
#ifndef LANDING_FWD_WAVE
#define LANDING_FWD_WAVE 0         // 1: the state recursion of the forward sweep by one wave (forward_pass; round-6 experiment)
#endif
#ifndef LANDING_PIVOT_2X2
#define LANDING_PIVOT_2X2 1        // the 4 x 4 pivot block through its 2 x 2 partition (pivot_block_step; 0: LDL^T + two triangular solves, rounds 2-4)
#endif
#ifndef LANDING_PIVOT_BLOCK
#define LANDING_PIVOT_BLOCK 4      // pivot-block size of the fp64 stage elimination (8: built, measured, slower -- pivot_block_step)
#endif
constexpr int XCH = 112 * LANDING_PIVOT_BLOCK;   // one exchange buffer of the blocked elimination: pivot rows [64][PB] + pivot columns [48][PB]
static_assert(2 * XCH >= 24 * YS, "A1 also holds Y (24 x YS)");
struct Lds {
  double G[48 * GS];
  double P[24 * PS];
  double A1[2 * XCH];          // Y = P(:,0:12)*A^ (24 x YS = 888) while T^T P T is formed, then the elimination side block
                              // Ex (24 x ES): col 0 = gamma_u -> z, cols 1.. = I -> unit-lower inverse
  double Ah[2 * 12 * YS];      // A^ of the stage being eliminated and of the one being assembled (copy k & 1 belongs to stage k)
  double gam[48], pv[24], q[24], bv[2 * 12], sig[24], w[48], dinv[24];
  double red[(SOLVER_THREADS / 64) * 6];
  double filt_th[FILT_CAP], filt_ph[FILT_CAP];
  double prof[32];
  int flag;
  // member context, written once by every thread with identical values (read back as LDS broadcasts by the
  // __noinline__ phases so that they carry no register state across calls)
  // assembly (asm_issue / asm_copy / asm_terms): term table of the most frequent stage type (the others are read from L2) and, per stage, the
  // bases of the seven [J | H | Hc] segments + the type id (slot 7)
  int segb[SOLVER_NMAX * 8];
  double carry[ASM_NSLOT];          // partial sums of the destinations that are summed by more than one thread
  double rcl[RCG];                  // gradient of the running cost of the stage whose G / gamma are assembled
  double jhl[CX_LEN];               // cx: nonzeros of the stage being assembled, the seven [J | H | Hc] segments side by side (coalesced copy) | sigma | rho of its rows | 1, -1, 0
  double dump[64];                  // where the stores of terms that close nothing go
  const unsigned long long* ctab; const unsigned long long* ccomb; int c_ml, c_mid, rc_on;
  // bounds of the rows: lbg/ubg depend on the row's position inside its stage only (boundary rows | rows of a stage |
  // rows of the last stage, which has another layout), so 244 (lb, ub) pairs in LDS replace two ng-long workspace arrays
  // that every row pass used to stream (6 of ~30 array passes per iteration)
  double bnd_lb[36 + 2 * 104], bnd_ub[36 + 2 * 104];
  MemberMem M; Layout L; const double* p; int prof_on;
  IpmState ks;
  unsigned long long atab_mid[ATAB_LTMAX * SOLVER_THREADS];      // (behind everything the tables address: their offsets are 16 bits)
  unsigned long long acomb_mid[SOLVER_THREADS];
};
// One instance per workgroup (= per NLP).  Namespace scope keeps the LDS address space visible to every
// phase function (ds_* instructions instead of flat_*).
__shared__ Lds SH;
// Packed terms of the assembly tables, with every place given as a BYTE OFFSET inside the LDS block (the host builds the tables with
// offsetof: solver_capi.inc): value = [pa] * [pb] * [pc], summed into the open PIECE of a destination and stored to [d] (+ the
// distance to the second copy of A^ when `ah` is set and the stage's copy is 1); the sum restarts behind the store unless `keep`
// (the store of a term that closes nothing goes to a per-lane dump slot).  A piece is a whole destination -- an entry of G (upper
// triangle), gamma or A^ -- or one of the partial sums that asm_combine adds up in slot order (pieces are cut where the 256 equal
// chunks of rounds 1-4 were, so every sum keeps its association).
static_assert(36 * FWD_DXL_NMAX + 12 <= 24 * PS + 2 * XCH + 2 * 12 * YS && offsetof(Lds, A1) == offsetof(Lds, P) + sizeof(double) * 24 * PS && offsetof(Lds, Ah) == offsetof(Lds, A1) + sizeof(double) * 2 * XCH,
              "dx (x layout, N <= FWD_DXL_NMAX) fits the adjacent arrays [P | A1 | A^] (forward_pass / row_products)");
static_assert(offsetof(Lds, atab_mid) <= 65528, "the assembly tables address G, A^, gamma, cx, the partial sums and the dump slots with 16-bit byte offsets");
__host__ __device__ inline unsigned long long aterm_pack(int pa, int pb, int pc, int d, bool keep, bool ah) {
  const unsigned lo = (unsigned)pa | ((unsigned)pb << 16);
  const unsigned hi = (unsigned)pc | (((unsigned)d | (keep ? 1u : 0u) | (ah ? 2u : 0u)) << 16);
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
// destination made of n partial sums (slots at byte offset s0 onwards, added in that order): lo = n | s0 << 16, hi = d | ah << 1; n = 0: nothing
__host__ __device__ inline unsigned long long acomb_pack(int n, int s0, int d, bool ah) {
  return (unsigned long long)((unsigned)n | ((unsigned)s0 << 16)) | ((unsigned long long)((unsigned)d | (ah ? 2u : 0u)) << 32);
}

// fp64 matrix-core tile: D(16x16) = C + A(16 x 4KT) B(4KT x 16) with v_mfma_f64_16x16x4_f64.  Operand layout
// (cdna_hip_programming.md section 3): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the accumulator holds
// D[row = (l>>4) + 4r][col = l&15], r = 0..3.  fa(i,k) / fb(k,j) fetch operands (LDS, zero outside the matrix).
typedef double f64x4 __attribute__((vector_size(32)));
template <int KT, class FA, class FB>
__device__ __forceinline__ f64x4 mfma_tile(f64x4 c, FA fa, FB fb) {
  const int l = threadIdx.x & 63, ij = l & 15, kq = l >> 4;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) c = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(ij, kt * 4 + kq), fb(kt * 4 + kq, ij), c, 0, 0, 0);
  return c;
}

// 1/d by v_rcp_f64 + two Newton steps (<= 1 ulp measured, tools/dev/rcp_prec.hip; the IEEE division sequence is 4x longer).
__device__ __forceinline__ double fast_rcp(double d) {
  double i = __builtin_amdgcn_rcp(d);
  i = fma(i, fma(-d, i, 1.0), i);
  return fma(i, fma(-d, i, 1.0), i);
}

// Broadcast of one lane's fp64 value to the whole wave through the scalar unit (v_readlane_b32 x2; `src` must be
// wave-uniform -- it is a compile-time constant in the unrolled elimination below).
__device__ __forceinline__ double lane_bcast(double v, int src) {
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  return __hiloint2double(hi, lo);
}

// Elimination of the controls of one stage by ONE wavefront, entirely in registers: Gauss-Jordan on
// [G_uu | G_us | gamma_u], lane c owning column c (NU + 24 + 1 <= 49 lanes, NU values each).  The pivot of step j is
// broadcast from lane j, the multipliers from lane j's column; no LDS traffic and no barrier inside the steps.
// Used for the 12 x 12 block of the free feet of stage 0 only (the stages themselves go through block_eliminate).  On success Kl (LDS, NU x 24, row stride 24) holds
// K = G_uu^-1 G_us and kl the vector kappa = G_uu^-1 gamma_u.  A non-positive / non-finite pivot (wrong inertia)
// is reported through *flag = 0; the caller raises delta_w.
template <int NU>
__device__ __noinline__ void gauss_jordan_wave(const double* G, const double* gam, double* Kl, double* kl, int* flag) {
  const int lane = threadIdx.x;        // called by the first wave only (threadIdx.x < 64)
  double col[NU];
  const bool isUU = lane < NU, isUS = lane >= NU && lane < NU + 24, isG = lane == NU + 24;
  {   // branch-free column fetch: every lane walks its own (base, stride); idle lanes re-read column 0
    const double* src = isG ? gam + 24 : G + 24 * GS + (isUU ? 24 + lane : (isUS ? lane - NU : 0));
    const int stride = isG ? 1 : GS;
#pragma unroll
    for (int i = 0; i < NU; ++i) col[i] = src[i * stride];
  }
  bool ok = true;
  // Rolled pivot loop with a rotating row file: step j finds its pivot row in col[0]; every update writes row i
  // into slot i-1 (the FMA destination differs from its accumulator source, so the rotation costs nothing) and the
  // normalised pivot row goes to the last slot.  After NU steps the rows are back in place.  One small loop body
  // (instruction-cache resident) instead of NU unrolled copies; the lane select of the broadcasts is the loop
  // counter (scalar).
#pragma unroll 1
  for (int j = 0; j < NU; ++j) {
    const double d = lane_bcast(col[0], j);
    if (!(d > 0.0) || !(d < 1e300)) ok = false;        // wave-uniform; keep going (values are discarded)
    double inv = __builtin_amdgcn_rcp(d);               // v_rcp_f64 + two Newton steps instead of the IEEE division
    inv = fma(inv, fma(-d, inv, 1.0), inv);             // sequence: the reciprocal sits on the serial pivot chain
    inv = fma(inv, fma(-d, inv, 1.0), inv);
    const double pj = col[0] * inv;
    // multipliers first (independent scalar broadcasts, back to back), then the row updates
    double m[NU];
#pragma unroll
    for (int i = 1; i < NU; ++i) m[i] = lane_bcast(col[i], j);
    __builtin_amdgcn_sched_barrier(0);      // all broadcasts in flight before the first update consumes one
#pragma unroll
    for (int i = 1; i < NU; ++i) col[i - 1] = fma(-m[i], pj, col[i]);
    col[NU - 1] = pj;
  }
  if (isUS) {
#pragma unroll
    for (int i = 0; i < NU; ++i) Kl[i * 24 + (lane - NU)] = col[i];
  }
  if (isG) {
#pragma unroll
    for (int i = 0; i < NU; ++i) kl[i] = col[i];
  }
  if (lane == 0) *flag = ok ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward sweep, round 5: the ASSEMBLY of stage k - 1 rides on the elimination of stage k.
// While the four waves eliminate the controls of stage k on the matrix cores, they copy stage k - 1's CCS nonzeros, sigma / rho of its
// rows and the residual of its dynamics rows from the workspace into LDS (coalesced loads issued behind the first block step), form
// the condensation terms  G = H + J_d^T Sigma J_d, gamma = J_d^T rho, A^ = -dg_dyn/d(X,c,f)  and store them straight into G, gamma
// and the NEXT stage's copy of A^ / b between two later block steps.  Rounds 1-4 ran the condensation as a phase of its own over all
// stages (0.106 ms of the 0.727 ms an iteration took under load) that wrote a 704-double block per stage to the member's workspace,
// which the sweep read back (473 KB per member and iteration, 12 % of the kernel's HBM traffic) and scattered into G between two
// barriers at the head of every stage (0.6-0.9 us of the 7 us a stage took).  The arithmetic and its order are unchanged: results are
// bit-identical to round 4's (tools/dev/emu_ab.py).  Built on the way, measured on the GPU and rejected (profiles/r05_ab_experiments.txt):
//  * the round-4 term tables (segment / offset pairs, destination codes, carries between threads) decoded inside the block steps, operands
//    gathered from the workspace: +2.3 us per stage (60 gather instructions of 64 scattered addresses per stage); operands staged through
//    LDS first: +2.0 us (500 instructions per thread and stage): whatever the four waves execute together is on the chain, so the
//    per-term work has to be a handful of instructions;
//  * an ASSEMBLER WAVE: waves 0..2 eliminate (wave 2 carrying the right-hand-side column as a second tile), wave 3 assembles alone: the
//    elimination itself went from 5.0 to 7.2 us per stage (second tile: 148+ registers, callee-saved spills whose reloads sit in front
//    of every return) and one wave needs 1.1 us more than that for 1 280 terms.
//
// The work is described by packed 8-byte terms (aterm_pack) in stage-LOCAL form: one table per stage type (first / middle /
// penultimate / last), the middle one resident in LDS.  A thread owns whole destinations (balanced on the host: 5-6 terms per
// thread).  Rounds 1-4 cut the term list into 256 equal chunks, one per thread, and a destination whose terms straddled chunk borders
// was finished from per-thread carries in thread order; now a destination is summed by one thread, which restarts the partial sum
// where those borders were (`bnd`), so every sum keeps its association.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(1)))* landing_gptr;    // global address space: global_load, not flat_load
#else                                                                     // (a flat load also ties up the LDS counter)
typedef const double* landing_gptr;
#endif
constexpr int ASM_SEG_LEN[7] = {NZ_JU - NZ_JX, NZ_JUN - NZ_JU, NZ_HX - NZ_JUN, NZ_HU - NZ_HX, NZ_HUN - NZ_HU, NZ_TOT - NZ_HUN, RUNC};
constexpr int ASM_SEG_POS[7] = {NZ_JX, NZ_JU, NZ_JUN, NZ_HX, NZ_HU, NZ_HUN, NZ_TOT};
static_assert(NZ_JUN - NZ_JU <= SOLVER_THREADS && 2 * 104 <= SOLVER_THREADS, "one load per thread and segment");
struct AsmRegs { double jh[7], sr, g, rc; };
// (1) coalesced loads of stage k's nonzeros (one per segment), sigma / rho of its rows, the residual of its dynamics rows; nothing is consumed here
__device__ __forceinline__ void asm_issue(int k, AsmRegs& R) {
  const Lds& S = SH;
  const MemberMem& M = S.M;
  const int tid = threadIdx.x, ng = S.L.ng;
  landing_gptr JH = (landing_gptr)M.J;       // [J | H | Hc] are contiguous
  landing_gptr SR = (landing_gptr)M.sig;     // [sigma | rho] are contiguous
  const int* sb = S.segb + k * 8;
  const int g0 = S.L.g_stage(k);
#pragma unroll
  for (int sgm = 0; sgm < 7; ++sgm) R.jh[sgm] = JH[sb[sgm] + (tid < ASM_SEG_LEN[sgm] ? tid : 0)];
  R.sr = SR[tid < 104 ? g0 + tid : (tid < 208 ? ng + g0 + tid - 104 : 0)];      // (the last stage has 80 rows: the tail reads the workspace behind them, never used)
  R.g = ((landing_gptr)M.g)[g0 + (tid < 12 ? tid : 0)];
  R.rc = 0.0;
  if (S.rc_on) R.rc = ((landing_gptr)M.cond)[(size_t)k * RCG + (tid < RCG ? tid : 0)];
}
// (2) ... into LDS; a barrier follows before asm_terms
__device__ __forceinline__ void asm_copy(const AsmRegs& R) {
  Lds& S = SH;
  const int tid = threadIdx.x;
#pragma unroll
  for (int sgm = 0; sgm < 7; ++sgm) if (tid < ASM_SEG_LEN[sgm]) S.jhl[ASM_SEG_POS[sgm] + tid] = R.jh[sgm];
  if (tid < 208) S.jhl[CX_SR + tid] = R.sr;
}
// (3) products and stores: G and gamma (dead since the tile fetch of the running elimination), A^ / b of copy `nb`.  ATAB_TB terms
// at a time: codes, then all operands, then the arithmetic and the stores -- three LDS round trips per batch instead of three per term
__device__ __forceinline__ void asm_terms(int k, int nb, const AsmRegs& R) {
  Lds& S = SH;
  const int tid = threadIdx.x, LT = S.c_ml, tp = S.segb[k * 8 + 7];
  char* const lds0 = reinterpret_cast<char*>(&S);
  const unsigned ahoff = nb ? (unsigned)(12 * YS * sizeof(double)) : 0u;
  const bool mid = (tp == S.c_mid);
  const unsigned long long* gt = S.ctab + (size_t)tp * LT * SOLVER_THREADS;
  double acc = 0.0;
  for (int j0 = 0; j0 < LT; j0 += ATAB_TB) {
    unsigned long long t[ATAB_TB];
    if (mid) {
#pragma unroll
      for (int u = 0; u < ATAB_TB; ++u) t[u] = S.atab_mid[(j0 + u) * SOLVER_THREADS + tid];
    } else {
#pragma unroll
      for (int u = 0; u < ATAB_TB; ++u) t[u] = gt[(j0 + u) * SOLVER_THREADS + tid];
    }
    double a[ATAB_TB], b[ATAB_TB], c[ATAB_TB];
#pragma unroll
    for (int u = 0; u < ATAB_TB; ++u) {
      const unsigned lo = (unsigned)t[u], hi = (unsigned)(t[u] >> 32);
      a[u] = *reinterpret_cast<const double*>(lds0 + (lo & 0xffffu)); b[u] = *reinterpret_cast<const double*>(lds0 + (lo >> 16)); c[u] = *reinterpret_cast<const double*>(lds0 + (hi & 0xffffu));
    }
#pragma unroll
    for (int u = 0; u < ATAB_TB; ++u) {
      const unsigned dd = (unsigned)(t[u] >> 48);
      acc += a[u] * b[u] * c[u];
      *reinterpret_cast<double*>(lds0 + ((dd & 0xfff8u) + ((dd & 2u) ? ahoff : 0u))) = acc;
      acc = (dd & 1u) ? acc : 0.0;
    }
  }
  if (tid < 12) { const int i = tid; S.bv[nb * 12 + (i < 6 ? i : (i < 9 ? i + 3 : i - 3))] = -R.g; }
  if (tid < RCG) S.rcl[tid] = R.rc;      // gradient of the running cost (w order X, c, f): added to gamma when its tile is fetched
}
// (4) behind the next barrier: the destinations whose terms more than one thread summed
__device__ __forceinline__ void asm_combine(int k, int nb) {
  Lds& S = SH;
  const int tid = threadIdx.x, tp = S.segb[k * 8 + 7];
  const unsigned long long code = tp == S.c_mid ? S.acomb_mid[tid] : S.ccomb[tp * SOLVER_THREADS + tid];
  const unsigned lo = (unsigned)code, dd = (unsigned)(code >> 32);
  const int n = (int)(lo & 7u);
  if (n > 0) {
    char* const lds0 = reinterpret_cast<char*>(&S);
    const double* sl = reinterpret_cast<const double*>(lds0 + (lo >> 16));
    double tot = 0.0;
    for (int i = 0; i < n; ++i) tot += sl[i];
    *reinterpret_cast<double*>(lds0 + ((dd & 0xfff8u) + ((dd & 2u) ? (nb ? (unsigned)(12 * YS * sizeof(double)) : 0u) : 0u))) = tot;
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef double __attribute__((address_space(1)))* landing_gptr_w;
#else
typedef double* landing_gptr_w;
#endif

// Elimination of the controls of one stage on the matrix cores (waves 0..2): blocked Gauss-Jordan with
// 4 x 4 pivot blocks on the (NU + 24) x (NU + 25) array  [G_uu G_us gamma_u ; G_su G_ss gamma_s]  (rows/columns: controls
// first).  After NU/4 block steps the control rows hold [I | K | kappa] and the state rows hold the Schur complement
// [0 | P_k | p_k] -- gains and cost-to-go in one pass.  The array lives in v_mfma_f64_16x16x4 accumulator tiles: wave w
// owns column tile w (16 columns x 48 rows = 3 tiles; with NU = 24 the right-hand side is column 48, the only live column of
// wave 3's tile).  One block step = one LDS exchange (the owner of the pivot
// columns publishes them, every wave publishes its slice of the 4 pivot rows), one barrier, the 4 x 4 LDL^T
// (recomputed by every lane from the broadcast block: no further communication), the normalised pivot rows of the
// own columns, and ONE rank-4 MFMA per tile:  T -= C R  (C = pivot columns with the pivot rows blanked; the pivot
// rows are then overwritten by R itself -- forming them as W - (D - I) R would cancel at the scale of W).  The pivots of the 4 x 4 LDL^T are the scalar pivots of the unblocked
// elimination, so the inertia test (all pivots positive) is unchanged.  Returns false on a non-positive pivot (at the end of the stage).
//
// One step of the blocked Gauss-Jordan elimination with a PS x PS pivot block (PS = 4 or 8) at rows / columns [OFF, OFF + PS) of the
// tile array T (wave ct owns column tile ct, accumulator layout: T[rt][r] = element (row 16 rt + lk + 4 r, column 16 ct + lj)).
// Exchange through LDS (buffer STEP & 1 of S.A1): every lane publishes the PS pivot-row entries of its column, W[c][PS]; the PS
// lane-columns that hold the pivot columns publish them, C[row][PS]; one barrier; then every lane factors the pivot block
// D = L diag(d) L^T redundantly from the broadcast copy (no further communication), solves D r = w for its own column and the
// wave applies the rank-PS update T -= C R with PS / 4 matrix-core instructions per tile; the pivot rows become R itself
// (forming them as W - (D - I) R would cancel at the scale of W).  The scalar pivots d are those of the unblocked elimination
// (inertia test unchanged).  PS = 4 is the product setting.  8 x 8 blocks (3 exchange + barrier rounds per stage
// instead of 6) were built and measured in round 3 and are SLOWER: the redundant per-lane factorisation grows with PS^3 -- about 210
// fp64 operations per lane and block at 4 issue cycles each against 2 x 40 -- which costs more than the three barrier rounds it
// saves: backward sweep 0.353 instead of 0.293 ms per iteration with the CU to itself (tools/dev/ab.sh, -DLANDING_PIVOT_BLOCK=8).
template <int NU, int PS, int OFF, int STEP>
__device__ __forceinline__ bool pivot_block_step(f64x4 (&T)[3], int ct, int lj, int lk, int c) {
  static_assert(PS == 4 || PS == 8, "pivot blocks of 4 or 8");
  static_assert((OFF & 3) == 0 && (OFF >> 4) == ((OFF + PS - 1) >> 4), "a pivot block lies inside one row tile");
  Lds& S = SH;
  constexpr int RTB = OFF >> 4, R0 = (OFF & 15) >> 2, NQ = PS / 4;
  constexpr int WSZ = 64 * PS;
  static_assert(WSZ + 48 * PS <= XCH, "exchange buffer");
  double* W = S.A1 + (STEP & 1) * XCH;
  double* C = W + WSZ;
#pragma unroll
  for (int q = 0; q < NQ; ++q) W[c * PS + 4 * q + lk] = T[RTB][R0 + q];
  if (ct == RTB && lj >= (OFF & 15) && lj < (OFF & 15) + PS) {
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(16 * rt + lk + 4 * r) * PS + (lj - (OFF & 15))] = T[rt][r];
  }
  __syncthreads();
  // pivot block (uniform reads), pivot rows of the own column, pivot-column operands
  double a[PS][PS];
#pragma unroll
  for (int i = 0; i < PS; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) a[i][j] = C[(OFF + i) * PS + j];
  double w[PS];
#pragma unroll
  for (int i = 0; i < PS; ++i) w[i] = W[c * PS + i];
  double am[NQ][3];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) { const int row = 16 * rt + lj; const double cv = C[row * PS + 4 * q + lk]; am[q][rt] = (row >= OFF && row < OFF + PS) ? 0.0 : cv; }   // pivot rows: no update
  auto recip = [](double d) { double i = __builtin_amdgcn_rcp(d); i = fma(i, fma(-d, i, 1.0), i); return fma(i, fma(-d, i, 1.0), i); };
#if LANDING_PIVOT_2X2
  static_assert(PS == 4, "2 x 2 partitioned pivot block");
  // The 4 x 4 pivot block through its 2 x 2 partition  D = [A B^T; B C]:  A^-1 from its determinant, E = B A^-1, Schur complement S = C - E B^T, S^-1 from its
  // determinant, then  r_2 = S^-1 (w_2 - E w_1),  r_1 = A^-1 w_1 - E^T r_2.  Two reciprocals on the serial chain instead of four and ~24 dependent operations
  // instead of ~64 of the LDL^T + two triangular solves (every lane computes this between the barrier and its matrix-core update: it is latency, not
  // throughput).  Inertia: the four leading principal minors a00, det A, det A * s00, det A * det S are positive exactly when the four pivots of the
  // unblocked elimination are (Sylvester), so the test -- a00, det A, s00, det S in (2^-1022, ~1e300) -- accepts the same matrices up to rounding.
  const double a00 = a[0][0], a10 = a[1][0], a11 = a[1][1], a20 = a[2][0], a21 = a[2][1], a22 = a[2][2], a30 = a[3][0], a31 = a[3][1], a32 = a[3][2], a33 = a[3][3];
  const double detA = fma(a00, a11, -a10 * a10), iA = recip(detA);
  const double e00 = fma(a20, a11, -a21 * a10) * iA, e01 = fma(a21, a00, -a20 * a10) * iA, e10 = fma(a30, a11, -a31 * a10) * iA, e11 = fma(a31, a00, -a30 * a10) * iA;
  const double s00 = fma(-e01, a21, fma(-e00, a20, a22)), s10 = fma(-e11, a21, fma(-e10, a20, a32)), s11 = fma(-e11, a31, fma(-e10, a30, a33));
  const double detS = fma(s00, s11, -s10 * s10), iS = recip(detS);
  unsigned hm = 0u;
  { const double piv[4] = {a00, detA, s00, detS};
#pragma unroll
    for (int j = 0; j < 4; ++j) { const unsigned h = (unsigned)__double2hiint(piv[j]) - 0x00100000u; hm = h > hm ? h : hm; } }
  const bool ok = hm < (0x7e37e43cu - 0x00100000u);
  auto solve = [&](const double (&wv)[PS], double (&R)[NQ]) {
    const double t2 = fma(-e01, wv[1], fma(-e00, wv[0], wv[2])), t3 = fma(-e11, wv[1], fma(-e10, wv[0], wv[3]));
    const double r2 = fma(s11, t2, -s10 * t3) * iS, r3 = fma(s00, t3, -s10 * t2) * iS;
    const double q0 = fma(a11, wv[0], -a10 * wv[1]) * iA, q1 = fma(a00, wv[1], -a10 * wv[0]) * iA;
    const double r0 = fma(-e10, r3, fma(-e00, r2, q0)), r1 = fma(-e11, r3, fma(-e01, r2, q1));
    R[0] = lk == 0 ? r0 : (lk == 1 ? r1 : (lk == 2 ? r2 : r3));
  };
#else
  // D = L diag(d) L^T: t[i][j] = l[i][j] d[j] = a[i][j] - sum_{k<j} l[i][k] t[j][k]
  double l[PS][PS], t[PS][PS], inv[PS];
  unsigned hm = 0u;
#pragma unroll
  for (int j = 0; j < PS; ++j) {
#pragma unroll
    for (int i = j; i < PS; ++i) {
      double acc = a[i][j];
#pragma unroll
      for (int kk = 0; kk < j; ++kk) acc = fma(-l[i][kk], t[j][kk], acc);
      t[i][j] = acc;
    }
    inv[j] = recip(t[j][j]);
#pragma unroll
    for (int i = j + 1; i < PS; ++i) l[i][j] = t[i][j] * inv[j];
    // all pivots in (2^-1022, ~1e300): one unsigned range test on the high words (negative, zero, subnormal, huge, inf and NaN
    // pivots all fall outside)
    const unsigned h = (unsigned)__double2hiint(t[j][j]) - 0x00100000u;
    hm = h > hm ? h : hm;
  }
  const bool ok = hm < (0x7e37e43cu - 0x00100000u);
  // normalised pivot rows of a column: D r = w by the two triangular solves (no explicit inverse: a badly conditioned
  // pivot block costs no more accuracy than the scalar elimination would); lane group lk keeps r[4 q + lk]
  auto solve = [&](const double (&wv)[PS], double (&R)[NQ]) {
    double y[PS], r[PS];
#pragma unroll
    for (int i = 0; i < PS; ++i) {
      double acc = wv[i];
#pragma unroll
      for (int kk = 0; kk < i; ++kk) acc = fma(-l[i][kk], y[kk], acc);
      y[i] = acc;
    }
#pragma unroll
    for (int i = PS - 1; i >= 0; --i) {
      double acc = y[i] * inv[i];
#pragma unroll
      for (int kk = i + 1; kk < PS; ++kk) acc = fma(-l[kk][i], r[kk], acc);
      r[i] = acc;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) R[q] = lk == 0 ? r[4 * q] : (lk == 1 ? r[4 * q + 1] : (lk == 2 ? r[4 * q + 2] : r[4 * q + 3]));
  };
#endif
  double R[NQ];
  solve(w, R);
  if (16 * ct + 16 > OFF) {                              // tiles whose columns are all eliminated already stay as they are
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) T[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[q][rt], -R[q], T[rt], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) T[RTB][R0 + q] = R[q];  // the pivot rows become the normalised rows, exactly
  }
  return ok;                                             // identical in every lane of the workgroup (tested after the update so
}                                                        // that the operand fetches are not held behind it)

// (`k` = stage, its copy of A^ / b is k & 1; the assembly of stage k - 1 rides along)
template <int NU, int PB>
__device__ __noinline__ bool block_eliminate(double* __restrict__ rec, double delta, int k) {
  Lds& S = SH;
  constexpr int NR = NU + 24;                         // rows; column NR is gamma
  const int tid = threadIdx.x, ct = tid >> 6, l = tid & 63, lj = l & 15, lk = l >> 4;
  const int c = 16 * ct + lj;
  const bool isg = (c == NR), live = (c <= NR);
  const int bcol = c < NU ? 24 + c : (c < NR ? c - NU : 0);        // position of the own column in the (sigma, f, c+) order of G
  const int cb = k & 1;
  const double* Ah = S.Ah + cb * (12 * YS);
  const double* bv = S.bv + cb * 12;
#ifdef LANDING_STAGE_PROF
  long long stg_t_ = SH.prof_on ? (long long)wall_clock64() : 0;
#define STG_T(i) do { if (SH.prof_on && threadIdx.x == 0) { const long long n_ = (long long)wall_clock64(); SH.prof[16 + (i)] += (double)(n_ - stg_t_); stg_t_ = n_; } } while (0)
#else
#define STG_T(i) do { } while (0)
#endif
  f64x4 T[3];
  // Every operand of the prologue is ONE unconditional LDS load: where a lane has no operand (dead column, row outside the array, structural zero)
  // the index points at a slot that holds 0.0 (cx's constant).  Written as `cond ? S.Ah[i] : 0.0` the compiler turned each of the ~40 operand fetches
  // into a branch around a load with a wait behind it -- a chain of LDS round trips, 1.4 us of the 7 us a stage takes (round 5).
  const double* const lds0 = reinterpret_cast<const double*>(&S);
  const int oG = (int)(offsetof(Lds, G) / sizeof(double)), oGam = (int)(offsetof(Lds, gam) / sizeof(double)), oP = (int)(offsetof(Lds, P) / sizeof(double)),
            oPv = (int)(offsetof(Lds, pv) / sizeof(double)), oAh = (int)(offsetof(Lds, Ah) / sizeof(double)) + cb * (12 * YS), oBv = (int)(offsetof(Lds, bv) / sizeof(double)) + cb * 12,
            oZ = (int)(offsetof(Lds, jhl) / sizeof(double)) + CX_ZERO;
  {   // tile fetch: every lane walks its own column of the condensed G (the assembly fills the upper triangle only) / of gamma; delta_w on the diagonal
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rho = 16 * rt + lk + 4 * r;
        const int a = rho < NU ? 24 + rho : (rho < NR ? rho - NU : 0);
        const bool in = live && rho < NR;
        const double v = lds0[in ? (isg ? oGam + a : oG + (a < bcol ? a * GS + bcol : bcol * GS + a)) : oZ] + (rho == c ? delta : 0.0);
        T[rt][r] = in ? v : 0.0;
      }
    if (S.rc_on) {      // (uniform) + gradient of the running cost of the stage's variables (X, c, f) in the column of gamma
#pragma unroll
      for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rho = 16 * rt + lk + 4 * r;
          const int a = rho < NU ? 24 + rho : (rho < NR ? rho - NU : 0);
          const double g = S.rcl[a < RCG ? a : 0];
          T[rt][r] += (isg && rho < NR && a < RCG) ? g : 0.0;
        }
    }
  }
  {   // + T^T P T and T^T (P b + p) of the next stage's cost-to-go, formed where it is consumed.  With
      // A_ext = [A^ | b] (12 rows) the own column of Y = P(:,0:12) A_ext comes out of the matrix cores in accumulator
      // layout, which IS the B-operand layout of the next product (row 4kt+k of k-step kt sits in lane group k):
      // Y never touches LDS.  Columns of c+ and the p-part of gamma enter P T directly.
    const bool cplus = live && !isg && bcol >= 36;
    double be[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) be[kt] = lds0[isg ? oBv + 4 * kt + lk : ((live && bcol < 36) ? oAh + (4 * kt + lk) * YS + bcol : oZ)];
    double pa1[3], add1[3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) pa1[kt] = lds0[oP + lj * PS + 4 * kt + lk];
#pragma unroll
    for (int r = 0; r < 3; ++r) { const int row = lk + 4 * r; add1[r] = lds0[cplus ? oP + row * PS + 12 + bcol - 36 : (isg ? oPv + row : oZ)]; }
    double av[3][3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
      const int rho = 16 * rt + lj;
      const int a = rho < NU ? 24 + rho : (rho < NR ? rho - NU : 99);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) av[rt][kt] = lds0[a < 36 ? oAh + (4 * kt + lk) * YS + a : oZ];
    }
    double pa2[3], add2[3];
    if (NU == 24) {
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) pa2[kt] = lds0[lj < 12 ? oP + (12 + lj) * PS + 4 * kt + lk : oZ];
#pragma unroll
      for (int r = 0; r < 3; ++r) { const int row = 12 + lk + 4 * r; add2[r] = lds0[cplus ? oP + row * PS + 12 + bcol - 36 : (isg ? oPv + row : oZ)]; }
    }
    f64x4 Y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) Y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1[kt], be[kt], Y1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) Y1[r] += add1[r];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) T[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt][kt], Y1[kt], T[rt], 0, 0, 0);
    if (NU == 24) {   // rows of c+ (control rows 12..23): + rows 12..23 of P T
      f64x4 Y2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa2[kt], be[kt], Y2, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 3; ++r) Y2[r] += add2[r];
      T[0][3] += Y2[0]; T[1][0] += Y2[1]; T[1][1] += Y2[2];
    }
  }
  // assembly of stage k - 1 (asm_issue / asm_copy / asm_terms / asm_combine): loads behind the first block step (every wave is past
  // the tile fetch: G and gamma are free), LDS copy before block step SC, whose barrier publishes it, products and stores behind
  // that step, sums of several threads behind the next one.
  // A non-positive pivot does not end the stage early: the sweep is abandoned by the caller, a few block steps later
  constexpr int NSTEP = (NU + PB - 1) / PB, SC = NSTEP >= 5 ? 3 : 1;
  static_assert(NSTEP >= 3 && NSTEP <= 6 && SC + 2 <= NSTEP, "3..6 block steps per stage");
  AsmRegs nxt;
  const bool more = k > 0;                               // (uniform)
#ifndef LANDING_DEV_ASM_LEVEL      // development probe (tools/dev/stage_time.py): 0 = no assembly, 1 = loads and LDS copy, 2 = + products, 3 = everything (results are garbage below 3)
#define LANDING_DEV_ASM_LEVEL 3
#endif
#define ASM_HOOK(step) do { if constexpr (SC == (step) && LANDING_DEV_ASM_LEVEL >= 1) { if (more) asm_copy(nxt); } if constexpr (SC + 1 == (step) && LANDING_DEV_ASM_LEVEL >= 2) { if (more) asm_terms(k - 1, cb ^ 1, nxt); } if constexpr (SC + 2 == (step) && LANDING_DEV_ASM_LEVEL >= 3) { if (more) asm_combine(k - 1, cb ^ 1); } } while (0)
  STG_T(0);      // prologue
  bool ok = pivot_block_step<NU, (PB > NU ? NU : PB), 0, 0>(T, ct, lj, lk, c);
  STG_T(1);      // block step 0
  if (more && LANDING_DEV_ASM_LEVEL >= 1) asm_issue(k - 1, nxt);
  STG_T(2);      // asm_issue
  ASM_HOOK(1);
  if constexpr (NU > PB) ok &= pivot_block_step<NU, (NU - PB >= PB ? PB : NU - PB), PB, 1>(T, ct, lj, lk, c);
  ASM_HOOK(2);
  if constexpr (NU > 2 * PB) ok &= pivot_block_step<NU, (NU - 2 * PB >= PB ? PB : NU - 2 * PB), 2 * PB, 2>(T, ct, lj, lk, c);
  STG_T(3);      // block steps 1, 2
  ASM_HOOK(3);
  STG_T(4);      // asm_copy (NU = 24)
  if constexpr (NU > 3 * PB) ok &= pivot_block_step<NU, (NU - 3 * PB >= PB ? PB : NU - 3 * PB), 3 * PB, 3>(T, ct, lj, lk, c);
  STG_T(5);      // block step 3
  ASM_HOOK(4);
  STG_T(6);      // asm_terms
  if constexpr (NU > 4 * PB) ok &= pivot_block_step<NU, (NU - 4 * PB >= PB ? PB : NU - 4 * PB), 4 * PB, 4>(T, ct, lj, lk, c);
  STG_T(7);      // block step 4
  ASM_HOOK(5);
  STG_T(8);      // asm_combine
  if constexpr (NU > 5 * PB) ok &= pivot_block_step<NU, (NU - 5 * PB >= PB ? PB : NU - 5 * PB), 5 * PB, 5>(T, ct, lj, lk, c);
  STG_T(9);      // block step 5
  ASM_HOOK(6);
#undef ASM_HOOK
  if (!ok) return false;                                 // (identical in every lane)
  {   // closed-loop state map for the forward sweep: X+ = A^_sigma sigma + A^_f f + b with f = -(K_f sigma + kappa_f), i.e.
      // Mt = A^_sigma - A^_f K_f, mv = b - A^_f kappa_f.  K_f / kappa_f are rows 0..11 of the first row tile, already in
      // B-operand layout (k-step kt = accumulator kt).
    f64x4 Mq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const double af = lds0[lj < 12 ? oAh + lj * YS + 24 + 4 * kt + lk : oZ];
      Mq = __builtin_amdgcn_mfma_f64_16x16x4f64(af, T[0][kt], Mq, 0, 0, 0);
    }
    if (c >= NU && c <= NR) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int i = lk + 4 * r;
        if (c < NR) rec[RIC_MT + i * 24 + (c - NU)] = Ah[i * YS + (c - NU)] - Mq[r];
        else rec[RIC_MV + i] = bv[i] - Mq[r];
      }
    }
  }
  // gains to the stage record, cost-to-go to LDS (+ its state rows to the record)
  if (c >= NU && c <= NR) {
    const int sj = c - NU;
#pragma unroll
    for (int rt = 0; rt < 3; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rho = 16 * rt + lk + 4 * r;
        const double v = T[rt][r];
        if (rho < NU) {
          if (c < NR) rec[RIC_K + rho * 24 + sj] = v; else rec[RIC_KAP + rho] = v;
        } else if (rho < NR) {
          const int i = rho - NU;
          if (c < NR) { S.P[i * PS + sj] = v; if (i < 12) rec[RIC_PX + i * 24 + sj] = v; }
          else { S.pv[i] = v; if (i < 12) rec[RIC_PV + i] = v; }
        }
      }
  }
  STG_T(10);     // epilogue
  return true;
}

// One backward Riccati step (templated on the control dimension: 24 = (f_k, c_{k+1}), 12 = last stage): elimination of the
// controls of stage k (G + T^T P T; P_k, p_k -> LDS, gains -> record k) with the assembly of stage k - 1 riding along.
template <int NU>
__device__ __forceinline__ bool riccati_step(double* rec, double delta, int k) {
  const bool ok = block_eliminate<NU, LANDING_PIVOT_BLOCK>(rec, delta, k);
  __syncthreads();
  return ok;
}

// One backward Riccati sweep with regularisation delta (terminal cost-to-go, stages N-1..0, free feet of
// stage 0).  false = a pivot was not positive (wrong inertia): the caller raises delta and retries.
__device__ LANDING_INL_BACK bool riccati_backward(double delta) {
  Lds& S = SH;
  const Layout& L = S.L;
  const MemberMem& M = S.M;
  const double* p = S.p;
  const int N = L.N, lane = threadIdx.x, NT = blockDim.x;
  (void)p; (void)N; (void)lane; (void)NT;
  bool ok = true;
  // terminal cost-to-go on sigma_N = X_N: diagonal (terminal rows are copies of X_N, gen:94-97)
  for (int e = lane; e < 24 * PS; e += NT) S.P[e] = 0.0;
  __syncthreads();
  if (lane < 12) {
    const int i = lane;
    const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
    const double qn2 = S.ks.feas ? 0.0 : 2.0 * p[L.o_QN + i];      // (the feasibility phase has no objective)
    S.P[i * PS + i] = qn2 + M.sig[ra] + M.sig[rb] + delta;
    S.pv[i] = qn2 * (M.x[12 * N + i] - p[12 * N + i]) + M.rho[ra] + M.rho[rb];
    double* rec = M.ric + (size_t)N * RIC_STRIDE;      // record N: P_N (diag), p_N
    for (int j = 0; j < 24; ++j) rec[RIC_PX + i * 24 + j] = (j == i) ? S.P[i * PS + i] : 0.0;
    rec[RIC_PV + i] = S.pv[i];
  }
  // background of the condensed stage data: the scatter below writes the structural nonzeros only (the patterns
  // of the stages nest along the sweep unless the tables say otherwise), nothing else writes G / A^ any more
  for (int e = lane; e < 48 * GS; e += NT) S.G[e] = 0.0;
  for (int e = lane; e < 2 * 12 * YS; e += NT) S.Ah[e] = 0.0;
  if (lane < 24) S.pv[lane] = (lane < 12) ? S.pv[lane] : 0.0;
  __syncthreads();
  {   // the only exposed assembly of the sweep
    AsmRegs first;
    asm_issue(N - 1, first);
    asm_copy(first);
    __syncthreads();
    asm_terms(N - 1, (N - 1) & 1, first);
    __syncthreads();
    asm_combine(N - 1, (N - 1) & 1);
  }
  __syncthreads();
  for (int k = N - 1; k >= 0 && ok; --k) {
    const bool last = (k == N - 1);
    long long tb_ = S.prof_on ? (long long)wall_clock64() : 0;
    // ---- G + T^T P T, elimination of the controls: P_k, p_k, gains -> record k; the assembly of stage k - 1 rides along
    double* rec = M.ric + (size_t)k * RIC_STRIDE;
    ok = last ? riccati_step<12>(rec, delta, k) : riccati_step<24>(rec, delta, k);
    if (lane == 0) { S.prof[PH_NSTAGE] += 1.0; if (ok) S.prof[PH_NSTAGE_OK] += 1.0; }
    PROF_ADD(PH_B_ELIM, tb_);
  }
  if (ok) {
    // ---- stage 0: X_0 fixed, feet c_0 free: P_cc dc0 = -(p_c + P_cx dX0), same elimination on a 12x12 block
    if (lane < 12) {
      const int i = lane;
      const double x0i = (i < 6) ? p[L.o_q_init + i] : p[L.o_qd_init + i - 6];
      S.sig[i] = x0i - M.x[i];
    }
    for (int e = lane; e < 144; e += NT) { const int i = e / 12, j = e % 12; S.G[(24 + i) * GS + 24 + j] = S.P[(12 + i) * PS + 12 + j]; }
    __syncthreads();
    if (lane < 12) {
      double v = S.pv[12 + lane];
      for (int t = 0; t < 12; ++t) v += S.P[(12 + lane) * PS + t] * S.sig[t];
      S.gam[24 + lane] = v;
    }
    __syncthreads();
    if (lane < 64) gauss_jordan_wave<12>(S.G, S.gam, S.A1, S.A1 + 24 * 24, &S.flag);
    __syncthreads();
    ok = S.flag != 0;
    if (lane == 0) { S.prof[PH_NSTAGE] += 1.0; if (ok) S.prof[PH_NSTAGE_OK] += 1.0; }
    if (ok) {
      if (lane < 12) S.sig[12 + lane] = -S.A1[24 * 24 + lane];
      __syncthreads();
    }
  }
  __syncthreads();
  return ok;
}

// Forward sweep: dx of every stage, next states, multipliers of the dynamics rows.
// The serial part is the state recursion alone: sigma_{k+1} = [Mt; -K_c] sigma_k + [mv; -kappa_c] (one 24 x 24
// product and one barrier per stage; the 612 doubles of a stage are prefetched four stages ahead into registers
// and passed on through a double-buffered LDS slot).  The forces f_k = -(K_f sigma_k + kappa_f) and the multipliers
// of the dynamics rows y_k = -(P_{k+1} sigma_{k+1} + p_{k+1})_X do not feed the recursion: they are evaluated for
// all stages at once afterwards.
__device__ LANDING_INL_FWD void forward_pass() {
  Lds& S = SH;
  const Layout& L = S.L;
  const MemberMem& M = S.M;
  const int N = L.N, tid = threadIdx.x, NT = blockDim.x;
  double* sg = S.G;                    // sigma_k, k = 0..N  at sg[24 k]           (N <= SOLVER_NMAX = 96: 2328 of the 2352 doubles of G)
  double* buf = S.jhl;                 // two stage slots of RIC_FWDN doubles: the assembler's copy of a stage's nonzeros (free here) and A1
  double* buf1 = S.A1;
  static_assert(24 * (SOLVER_NMAX + 1) <= 48 * GS && RIC_FWDN <= NZ_TOT + RUNC && RIC_FWDN <= XCH * 2, "forward scratch fits");
  auto slot = [&](int k) { return (k & 1) ? buf1 : buf; };
  auto fetch = [&](int k, double (&r)[3]) {
    landing_gptr rec = (landing_gptr)(M.ric + (size_t)(k < N ? k : N - 1) * RIC_STRIDE + RIC_FWD0);   // global_load: a flat load would also tie up the LDS counter
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int e = tid + j * 256; r[j] = rec[e < RIC_FWDN ? e : RIC_FWDN - 1]; }   // unconditional (clamped):
  };                                                                                                     // the memory counter stays countable
  auto stash = [&](int k, const double (&r)[3]) {
    double* b = slot(k);
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int e = tid + j * 256; if (k < N && e < RIC_FWDN) b[e] = r[j]; }
  };
  if (tid < 24) sg[tid] = S.sig[tid];
#if LANDING_FWD_WAVE
  // Round 6 experiment (profiles/r06_ab_experiments.txt): the recursion is a 24 x 24 product per stage -- ONE wave runs it (two lanes per row, the same association of the sums as the
  // eight-lanes-per-row form below, so the same bits), with a wave-local ordering between the stages instead of a workgroup barrier; the other three waves stage the records of the next
  // two stages into LDS meanwhile (loads issued two groups ahead).  One workgroup barrier per TWO stages.
  {
    constexpr int FW = RIC_FWDN;
    static_assert(FW <= NZ_TOT + RUNC && FW <= 2 * 12 * YS && 2 * FW <= 24 * PS + 2 * XCH, "four stage slots: cx | A^ | [P | A1] (two)");
    double* const slotp[4] = {S.jhl, S.Ah, S.P, S.P + FW};      // (P and A1 are adjacent members of Lds)
    static_assert(offsetof(Lds, A1) == offsetof(Lds, P) + sizeof(double) * 24 * PS, "P and A1 adjacent");
    const int wave = tid >> 6, l = tid & 63, lt = tid - 64;
    auto gfetch = [&](int grp, double (&r)[7]) {      // stages 2 grp, 2 grp + 1: 2 x 612 doubles over the 192 loader threads
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        int e = lt + 192 * i; e = e < 2 * FW ? e : 2 * FW - 1;
        const int st = e >= FW ? 1 : 0; int k = 2 * grp + st; k = k < N ? k : N - 1;
        r[i] = ((landing_gptr)(M.ric + (size_t)k * RIC_STRIDE + RIC_FWD0))[e - st * FW];
      }
    };
    auto gstash = [&](int grp, const double (&r)[7]) {
      double* const b0 = slotp[2 * (grp & 1)]; double* const b1 = slotp[2 * (grp & 1) + 1];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int e = lt + 192 * i, st = e >= FW ? 1 : 0;
        if (e < 2 * FW && 2 * grp + st < N) (st ? b1 : b0)[e - st * FW] = r[i];
      }
    };
    auto group = [&](int grp) {      // the compute wave: stages 2 grp, 2 grp + 1 from the slots of pair grp & 1
      const int r = l & 31, h = l >> 5; const bool valid = r < 24; const int rr = valid ? r : 0;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int k = 2 * grp + st;
        if (k < N) {      // (uniform)
          const double* b = slotp[2 * (grp & 1) + st];
          const bool lastk = (k == N - 1);
          const double* row = (rr < 12) ? b + 312 + rr * 24 : b + (rr - 12) * 24;
          const double* sk = sg + 24 * k;
          double a[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { const int j = 4 * h + jj; a[jj] = row[3 * j] * sk[3 * j] + row[3 * j + 1] * sk[3 * j + 1] + row[3 * j + 2] * sk[3 * j + 2]; }
          double acc = (a[0] + a[1]) + (a[2] + a[3]);
          acc += __shfl_xor(acc, 32);
          if (valid && h == 0) sg[24 * (k + 1) + r] = (r < 12) ? b[600 + r] + acc : (lastk ? 0.0 : -(b[300 + (r - 12)] + acc));
        }
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // sigma_{k+1} is read by the other lanes of this wave in the next stage (LDS is in order within a wave)
#endif
        __builtin_amdgcn_wave_barrier();
      }
    };
    double rA[7], rB[7];
    if (wave > 0) { gfetch(0, rA); gfetch(1, rB); gstash(0, rA); gfetch(2, rA); }
    __syncthreads();
    const int NG = (N + 1) / 2;
    for (int g0 = 0; g0 < NG; g0 += 2) {
      if (wave == 0) group(g0); else { gstash(g0 + 1, rB); gfetch(g0 + 3, rB); }
      __syncthreads();
      if (wave == 0) group(g0 + 1); else { gstash(g0 + 2, rA); gfetch(g0 + 4, rA); }
      __syncthreads();
    }
  }
#else
  double r0[3], r1[3], r2[3], r3[3];
  fetch(0, r0); fetch(1, r1); fetch(2, r2); fetch(3, r3);
  stash(0, r0); fetch(4, r0);
  __syncthreads();
  auto stage = [&](int k, double (&rn)[3]) {
    // rn holds stage k+1 (loaded four stages ago): hand it to LDS, refill it with stage k+5
    if (k < N) {
      const double* b = slot(k);                 // [0,288) K rows of c+ | [288,312) kappa | [312,600) Mt | [600,612) mv
      const bool lastk = (k == N - 1);
      if (tid < 192) {
        const int r = tid >> 3, j = tid & 7;
        const double* row = (r < 12) ? b + 312 + r * 24 : b + (r - 12) * 24;
        const double* sk = sg + 24 * k;
        double acc = row[3 * j] * sk[3 * j] + row[3 * j + 1] * sk[3 * j + 1] + row[3 * j + 2] * sk[3 * j + 2];
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (j == 0) sg[24 * (k + 1) + r] = (r < 12) ? b[600 + r] + acc : (lastk ? 0.0 : -(b[300 + (r - 12)] + acc));
      }
    }
    stash(k + 1, rn);
    fetch(k + 5, rn);
    __syncthreads();
  };
  for (int k0 = 0; k0 < N; k0 += 4) {
    stage(k0, r1); stage(k0 + 1, r2); stage(k0 + 2, r3); stage(k0 + 3, r0);
  }
#endif
  // ---- everything that hangs off the states, all stages at once.  Round 6: dx also goes to LDS in the x layout ([P | A1 | A^] are free from here to the next backward sweep) for
  // row_products, whose gathers of dx were one of its two remaining global gathers per term (horizons up to FWD_DXL_NMAX; longer ones gather from the workspace as before)
  double* const dxl = S.P;
  const bool use_dxl = N <= FWD_DXL_NMAX;
  for (int e = tid; e < 24 * (N + 1); e += NT) {
    const int k = e / 24, i = e % 24;
    if (i < 12) { M.dx[L.x_X(k) + i] = sg[e]; if (use_dxl) dxl[L.x_X(k) + i] = sg[e]; }
    else if (k < N) { M.dx[L.x_U(k) + (i - 12)] = sg[e]; if (use_dxl) dxl[L.x_U(k) + (i - 12)] = sg[e]; }
  }
  for (int e = tid; e < 24 * N; e += NT) {
    const int k = e / 24, h = e % 24, i = h % 12;
    const double* sk = sg + 24 * (h < 12 ? k : k + 1);
    if (h < 12) {            // forces of stage k
      landing_gptr rec = (landing_gptr)(M.ric + (size_t)k * RIC_STRIDE);
      double acc = rec[RIC_KAP + i];
#pragma unroll
      for (int t = 0; t < 24; ++t) acc += rec[RIC_K + i * 24 + t] * sk[t];
      M.dx[L.x_U(k) + 12 + i] = -acc;
      if (use_dxl) dxl[L.x_U(k) + 12 + i] = -acc;
    } else {                 // multipliers of the dynamics rows of stage k (state order -> row order)
      landing_gptr recn = (landing_gptr)(M.ric + (size_t)(k + 1) * RIC_STRIDE);
      double acc = recn[RIC_PV + i];
#pragma unroll
      for (int t = 0; t < 24; ++t) acc += recn[RIC_PX + i * 24 + t] * sk[t];
      const int q = i < 6 ? i : (i < 9 ? i + 3 : i - 3);
      M.yn[L.g_stage(k) + q] = -acc;
    }
  }
  if (tid < 12) {
    const int i = tid;
    const double v = sg[24 * N + i];
    const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
    M.ds[ra] = v + (M.g[ra] - M.s[ra]);
    M.ds[rb] = v + (M.g[rb] - M.s[rb]);
  }
  __syncthreads();
}

// ds = J_d dx + (g - s) for every stage inequality row: fixed per-thread term lists over the whole horizon (chunks cut at row
// boundaries), 8-byte records {jac index (16 bits) | x index (16) | dst row (16) | closes (1) | valid (1)}, batches of 8 terms
// with the records of the next batch fetched under the gathers of this one.  (Tried in round 2: the stage-local LDS
// table scheme of condense() -- one term per thread and stage, carries through LDS, a barrier per 4 stages -- is slower
// here, alone 0.065 vs 0.039 ms and under load 0.105 vs 0.091: the list is short, 15 k terms, and latency-bound.)
__device__ LANDING_INL_ROWP void row_products(const unsigned long long* __restrict__ rterm, int rlen) {
  Lds& S = SH;
  const MemberMem& M = S.M;
  const int tid = threadIdx.x, NT = blockDim.x;
  const double* __restrict__ Jn = M.J; const double* __restrict__ dxv = M.dx;
  const double* dxl = S.P; const bool use_dxl = S.L.N <= FWD_DXL_NMAX;      // dx in LDS, x layout (forward_pass)
  double* __restrict__ dsv = M.ds;      // (round 6: the `+ (g - s)` of a row is added by the dual pass, which loads g and s of every row anyway -- two of the four gathers per term are gone)
  double acc = 0.0;
  constexpr int BW = 8;
  unsigned long long tn[BW];
#pragma unroll
  for (int u = 0; u < BW; ++u) tn[u] = (u < rlen) ? rterm[(size_t)u * NT + tid] : 0ull;
  for (int j0 = 0; j0 < rlen; j0 += BW) {
    unsigned long long t[BW];
#pragma unroll
    for (int u = 0; u < BW; ++u) t[u] = tn[u];
#pragma unroll
    for (int u = 0; u < BW; ++u) { const int j = j0 + BW + u; tn[u] = (j < rlen) ? rterm[(size_t)j * NT + tid] : 0ull; }
    double a[BW], b[BW];
#pragma unroll
    for (int u = 0; u < BW; ++u) {
      const int ij = (int)(t[u] & 0xffffu), ix = (int)((t[u] >> 16) & 0xffffu);
      a[u] = Jn[ij]; b[u] = use_dxl ? dxl[ix] : dxv[ix];
    }
#pragma unroll
    for (int u = 0; u < BW; ++u) {
      const bool valid = (t[u] >> 49) & 1ull, closes = (t[u] >> 48) & 1ull;
      acc += valid ? a[u] * b[u] : 0.0;
      if (closes) { dsv[(int)((t[u] >> 32) & 0xffffu)] = acc; acc = 0.0; }
    }
  }
  __syncthreads();
}

// Running-cost pieces of the solver (landing_form.run_cost, generate_quadruped_SRBM_CCC.m:81-89): out of line -- cold for the default
// terminal-cost objective, and their 36-entry gradient arrays stay out of the register allocation of the main loop.
__device__ __noinline__ void rc_init_hc() {      // constant Hessian entries of the running cost (layout: RUNC)
  const Layout& L = SH.L; const double* p = SH.p;
  for (int e = threadIdx.x; e < L.N * RUNC; e += blockDim.x) {
    const int k = e / RUNC, j = e % RUNC, a = j % 3;
    const double dt2 = 2.0 * p[L.o_dt + k];
    double v;
    if (j < 12) v = dt2 * (rc_QX(L, p, j) + (j < 3 ? 4.0 * rc_Qc(L, p, j) : 0.0));
    else if (j < 24) v = -dt2 * rc_Qc(L, p, a);
    else if (j < 36) v = dt2 * rc_Qc(L, p, a);
    else v = dt2 * rc_Qf(L, p, a);
    SH.M.Hc[e] = v;
  }
}
__device__ __noinline__ void rc_add_grad() {     // objective gradient of the stage variables into gx
  const Layout& L = SH.L; const MemberMem& M = SH.M;
  for (int k = threadIdx.x; k < L.N; k += blockDim.x) { double* gU = M.gx + L.x_U(k); (void)run_cost_stage(L, M.x, SH.p, k, M.gx + L.x_X(k), gU, gU + 12); }
}
__device__ __noinline__ void rc_add_gamma() {    // ... and, per stage, for the right-hand sides gamma_k (w order X, c, f): added when the tile of gamma is fetched (block_eliminate)
  const Layout& L = SH.L; const MemberMem& M = SH.M;
  for (int k = threadIdx.x; k < L.N; k += blockDim.x) {
    double gr[36];
#pragma unroll
    for (int a = 0; a < 36; ++a) gr[a] = 0.0;
    (void)run_cost_stage(L, M.x, SH.p, k, gr, gr + 12, gr + 24);
    double* gm = M.cond + (size_t)k * RCG;
#pragma unroll
    for (int a = 0; a < 36; ++a) gm[a] = gr[a];
  }
}
__device__ __noinline__ void rc_f_dphi(double& f0, double& dphi) {     // running cost at x and its directional derivative along dx (per-thread partial sums)
  const Layout& L = SH.L; const MemberMem& M = SH.M;
  for (int k0 = 0; k0 < L.N; k0 += blockDim.x) {
    const int k = k0 + threadIdx.x;
    if (k >= L.N) continue;
    double gr[36];
#pragma unroll
    for (int a = 0; a < 36; ++a) gr[a] = 0.0;
    f0 += run_cost_stage(L, M.x, SH.p, k, gr, gr + 12, gr + 24);
    const double* dX = M.dx + L.x_X(k); const double* dU = M.dx + L.x_U(k);
#pragma unroll
    for (int a = 0; a < 12; ++a) dphi += gr[a] * dX[a] + gr[12 + a] * dU[a] + gr[24 + a] * dU[12 + a];
  }
}
__device__ __noinline__ double rc_f(const double* x) {                 // running cost at x (per-thread partial sum)
  const Layout& L = SH.L;
  double f = 0.0;
  // (uniform trip count with a lane predicate: a loop whose trip count differs per lane right in front of the block
  // reduction hung the kernel on gfx950 / ROCm 7.2 even with the branch not taken -- bisected, tools/dev)
  for (int k0 = 0; k0 < L.N; k0 += blockDim.x) { const int k = k0 + threadIdx.x; if (k < L.N) f += run_cost_stage(L, x, SH.p, k, nullptr, nullptr, nullptr); }
  return f;
}

// ---- feasibility (restoration) phase: one side of an elastic inequality row ------------------------------------------------------
// The row  lb <= s  becomes  a = s - lb + n >= 0,  n >= 0  with the price rho_pen * n  (upper side: b = ub + q - s, q >= 0); multipliers
// z (of a >= 0) and w (of n >= 0), stationarity  rho_pen - z - w = 0.  Eliminating (dn, dw) from the primal-dual Newton system leaves
//   dz = c - sd * (z / D) ds,   D = a + z n / w,   c = (mu - a z - z (mu - n w + n (z + w - rho_pen)) / w) / D
// (sd = +1 lower side, -1 upper side: d(dist) = sd ds + dn), i.e. the row enters the condensed system with sigma = z / D exactly like
// a plain slack row with sigma = z / d -- condensation, Riccati sweep and forward sweep are those of the normal iteration.
struct ElStep { double dz, dw, dn, da; };
__device__ __forceinline__ ElStep el_step(double sd, double a, double n, double z, double w, double mu, double rho_pen, double ds) {
  ElStep e;
  const double D = a + z * n / w, rn = z + w - rho_pen;
  e.dz = (mu - a * z - z * (mu - n * w + n * rn) / w - sd * z * ds) / D;
  e.dw = -e.dz - rn;
  e.dn = (mu - n * w - n * e.dw) / w;
  e.da = sd * ds + e.dn;
  return e;
}

// Workgroups per CU the register budget is sized for.  Measured on MI355X (N = 40, end of round 1, IPRA off): 2 per CU
// (256 VGPRs) beats 3 (168) at batch 1024 -- 5220 vs 4760 NLPs/s -- and ties beyond (2048: 5410 vs 5330, 8192: 7080 vs
// 7050); 4 per CU (128 VGPRs, and the 46 KB of LDS would only fit three) is far behind.  Every phase is latency-bound:
// more resident workgroups mostly add contention, and the row passes / derivative phases spill less with 256 registers.
#ifndef LANDING_MIN_WAVES
#define LANDING_MIN_WAVES 2
#endif
// Receding-horizon shift (SURVEY 8f row N3 / BASELINE configs[4]): the initial guess of the next control tick is the previous
// solution advanced by one stage -- X(:,k) <- X(:,k+1), U(:,k) <- U(:,k+1), the last column held -- with the measured state
// in X(:,0); the parameter vector gets the new q_init / qd_init.  (The reference's warm-start variant re-solves from the stored
// previous solution unshifted, codegen_casadi/test_loadCasadi_ws.m:73-88; the shift is what a 100 Hz loop adds.)
__global__ void landing_mpc_shift_kernel(Layout L, int B, const double* __restrict__ x_prev, const double* __restrict__ state,
                                         double* __restrict__ p, double* __restrict__ x0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * L.nx) return;
  const int m = (int)(idx / L.nx), i = (int)(idx % L.nx), N = L.N, nX = 12 * (N + 1);
  const double* xp = x_prev + (size_t)m * L.nx;
  double v;
  if (i < 12) v = state[(size_t)m * 12 + i];
  else if (i < nX) { const int k = i / 12, r = i % 12; v = xp[12 * (k < N ? k + 1 : N) + r]; }
  else { const int j = i - nX, k = j / 24, r = j % 24; v = xp[nX + 24 * (k < N - 1 ? k + 1 : N - 1) + r]; }
  x0[idx] = v;
  if (i < 6) p[(size_t)m * L.np + L.o_q_init + i] = v;
  else if (i < 12) p[(size_t)m * L.np + L.o_qd_init + (i - 6)] = v;
}

// Dispatch order of a batch: workgroups are handed to the CUs in blockIdx order and a batch larger than the resident
// capacity (2 workgroups per CU) runs in waves, so the batch time is the finishing time of the slowest member -- which
// is much later when that member only starts in the second wave.  Members likely to need many iterations go first.
// Difficulty proxy = initial body height z0 = q_init(3) (it grows with |pitch| and the drop speed through the callers'
// touch-down rule, generate_training_data_automated.m:52-60): correlation 0.65 with the iteration count on the bench
// batches (tests/dev/ipm_lab.py), as good as what a 30-iteration probe predicts.  rank = number of members with a larger
// key (ties: lower index first) -- O(B^2) comparisons, deterministic, no atomics.  NaN keys sort last.
__global__ void landing_order_kernel(Layout L, int B, const double* __restrict__ p, int* __restrict__ order) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= B) return;
  const int off = L.o_q_init + 2;
  double key = p[(size_t)m * L.np + off];
  if (!(key == key)) key = -INFINITY;
  int rank = 0;
  for (int j = 0; j < B; ++j) {
    double kj = p[(size_t)j * L.np + off];
    if (!(kj == kj)) kj = -INFINITY;
    rank += (kj > key) || (kj == key && j < m);
  }
  order[rank] = m;
}

__global__ void __launch_bounds__(SOLVER_THREADS, LANDING_MIN_WAVES) landing_ipm_kernel(SolveArgs A) {
  if ((int)blockIdx.x >= A.B) return;
  const int m = A.order ? A.order[blockIdx.x] : (int)blockIdx.x;
  const Layout& L = A.L;
  const int N = L.N, lane = threadIdx.x, NT = blockDim.x;
  const int nx = L.nx, ng = L.ng;
  const double* p = A.p + (size_t)m * L.np;
  const landing_solver_opts& o = A.o;
  const MemberMem M = carve(L, A.ws + (size_t)m * A.ws_stride);
  Lds& S = SH;
  const double INF = INFINITY;
  // the workspace arrays never overlap: tell the compiler so that the row passes can batch their loads
  auto bidx = [N](int r) { if (r < 36) return r; const int k = (r - 36) / 104, q = (r - 36) - 104 * k; return 36 + (k == N - 1 ? 104 : 0) + q; };
  double* __restrict__ r_g = M.g; double* __restrict__ r_gt = M.gt; double* __restrict__ r_s = M.s; double* __restrict__ r_ds = M.ds;
  double* __restrict__ r_zL = M.zL; double* __restrict__ r_zU = M.zU;
  double* __restrict__ r_y = M.y; double* __restrict__ r_yn = M.yn; double* __restrict__ r_sig = M.sig; double* __restrict__ r_rho = M.rho;
  S.M = M; S.L = L; S.p = p; S.prof_on = A.prof != nullptr;
  S.ctab = A.ctab; S.ccomb = A.ccomb; S.c_ml = A.c_ml; S.c_mid = A.c_mid; S.rc_on = 0;
  if (lane < SOLVER_THREADS) S.acomb_mid[lane] = A.ccomb[A.c_mid * SOLVER_THREADS + lane];
  for (int e = lane; e < 64; e += NT) S.dump[e] = 0.0;
  for (int e = lane; e < A.c_ml * SOLVER_THREADS; e += NT) S.atab_mid[e] = A.ctab[(size_t)A.c_mid * A.c_ml * SOLVER_THREADS + e];      // term table of the most frequent stage type
  if (lane == 0) { S.jhl[CX_ONE] = 1.0; S.jhl[CX_MONE] = -1.0; S.jhl[CX_ZERO] = 0.0; }
  if (lane < 32) S.prof[lane] = 0.0;
  {   // condensation: segment bases of every stage, packed term table of the most frequent stage type
    const int nj = L.nnz_jac, nh = L.nnz_hess;
    for (int k = lane; k < N; k += NT) {
      int* sb = S.segb + k * 8;
      sb[0] = L.jx(k); sb[1] = L.ju(k); sb[2] = k < N - 1 ? L.ju(k + 1) : 0;
      sb[3] = nj + L.hx(k); sb[4] = nj + L.hu(k); sb[5] = k < N - 1 ? nj + L.hu(k + 1) : 0;
      sb[6] = nj + nh + k * RUNC; sb[7] = A.ctype[k];
    }
  }
  __syncthreads();

  // ------------------------------------------------------------------ initial point
  for (int i = lane; i < nx; i += NT) {
    double v = A.x0[(size_t)m * nx + i];
    if (i < 6) v = p[L.o_q_init + i]; else if (i < 12) v = p[L.o_qd_init + i - 6];   // X(:,1) is fixed (gen:90-91)
    M.x[i] = v;
  }
  for (int e = lane; e < 36 + 2 * 104; e += NT) {
    const int r = e < 36 ? e : (e < 36 + 104 ? L.g_stage(0) + (e - 36) : L.g_stage(N - 1) + (e - 36 - 104));
    double lb = 0.0, ub = 0.0;
    if (r < ng) bound_of(L, p, r, lb, ub);          // (the last stage has 80 rows: the tail of its slot is never addressed)
    S.bnd_lb[e] = lb; S.bnd_ub[e] = ub;
  }
  __syncthreads();
  if (L.run_cost) rc_init_hc();
  __syncthreads();
  member_eval_g_rare(L, M.x, p, M.g);
  __syncthreads();
  auto init_slacks = [&]() {     // slack pushed into the interior (IPOPT bound_push/frac), z = 1, y = z_U - z_L, y_dyn = 0
    for (int r = lane; r < ng; r += NT) {
      const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)];
      double sv = 0.0, zl = 0.0, zu = 0.0;
      if (r >= 12 && lb != ub) {
        const bool hL = lb > -INF, hU = ub < INF;
        sv = M.g[r];
        double pl, pu;
        if (hL && hU) { pl = fmin(o.bound_push * fmax(1.0, fabs(lb)), o.bound_frac * (ub - lb)); pu = fmin(o.bound_push * fmax(1.0, fabs(ub)), o.bound_frac * (ub - lb)); }
        else { pl = o.bound_push * fmax(1.0, hL ? fabs(lb) : 0.0); pu = o.bound_push * fmax(1.0, hU ? fabs(ub) : 0.0); }
        if (hL) sv = fmax(sv, lb + pl);
        if (hU) sv = fmin(sv, ub - pu);
        zl = hL ? 1.0 : 0.0; zu = hU ? 1.0 : 0.0;
      }
      M.s[r] = sv; M.zL[r] = zl; M.zU[r] = zu; M.y[r] = zu - zl;
    }
    __syncthreads();
  };
  // ... at the point a feasibility phase hands back (landing_nlp.h, feas_ret_push): slacks pushed only feas_ret_push off their bounds, bound multipliers
  // mu / distance -- what the phase gained in feasibility is kept
  auto init_slacks_return = [&](double mu_) {
    const double push = o.feas_ret_push;
    for (int r = lane; r < ng; r += NT) {
      const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)];
      double sv = 0.0, zl = 0.0, zu = 0.0;
      if (r >= 12 && lb != ub) {
        const bool hL = lb > -INF, hU = ub < INF;
        sv = M.g[r];
        double pl, pu;
        if (hL && hU) { pl = fmin(push * fmax(1.0, fabs(lb)), push * (ub - lb)); pu = fmin(push * fmax(1.0, fabs(ub)), push * (ub - lb)); }
        else { pl = push * fmax(1.0, hL ? fabs(lb) : 0.0); pu = push * fmax(1.0, hU ? fabs(ub) : 0.0); }
        if (hL) sv = fmax(sv, lb + pl);
        if (hU) sv = fmin(sv, ub - pu);
        zl = hL ? fmin(fmax(mu_ / (sv - lb), 1e-8), 1e3) : 0.0; zu = hU ? fmin(fmax(mu_ / (ub - sv), 1e-8), 1e3) : 0.0;
      }
      M.s[r] = sv; M.zL[r] = zl; M.zU[r] = zu; M.y[r] = zu - zl;
    }
    __syncthreads();
  };
  init_slacks();
  // primal / complementarity errors, Sigma and rho of the CURRENT point for barrier parameter mu_ (one pass over the
  // rows; the accept pass below produces the same quantities for the next iterate, so this runs only at the start,
  // after a multiplier reset and when mu changes)
  IpmState& K = S.ks;
  // K is written by thread 0 only, between barriers (all other threads only read it, after the closing barrier)
#define KS_BEGIN() __syncthreads(); if (lane == 0) {
#define KS_BEGIN_SYNCED() if (lane == 0) {      /* directly behind a barrier (block_reduce / block_top4 end with one) */
#define KS_END() } __syncthreads()
  // Row passes: every thread owns the rows lane + NT j.  They are processed RB at a time with ALL loads of a batch issued
  // up-front and unconditionally (every array is fully allocated; out-of-range rows re-read the last row and are masked):
  // one memory round trip per batch instead of two or three dependent ones per row behind the bound-type branches.
#ifndef LANDING_RB
#define LANDING_RB 4
#endif
  constexpr int RB = LANDING_RB;
  // primal / complementarity errors, Sigma and rho of the CURRENT point for barrier parameter mu_ (one pass over the
  // rows; the accept pass below produces the same quantities for the next iterate, so this runs only at the start,
  // after a multiplier reset and when mu changes).  Leaves c_pr, c_co, c_cm, |y|_1, |z|_1 and the number of bound multipliers in K.
  // the same for the elastic problem of the feasibility phase (plain row loops: the phase is rare); also leaves the violation of the
  // inequality rows at x (max norm, 1-norm) and |z + w - rho_pen|_inf in K
  auto feas_point_pass = [&](double mu_) {
    const double frho = o.feas_rho;
    double pr = 0.0, co = 0.0, cm = 0.0, rn = 0.0, ys = 0.0, zs = 0.0, nz = 0.0, vmax = 0.0, v1 = 0.0, teq = 0.0;
    for (int r = lane; r < ng; r += NT) {
      const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)];
      double sg = 0.0, rh = 0.0;
      if (r >= 12) {
        const double g = r_g[r];
        ys += fabs(r_y[r]);
        if (lb == ub) { pr = fmax(pr, fabs(g - lb)); teq += fabs(g - lb); }
        else {
          const double s = r_s[r], v = fmax(fmax(lb - g, g - ub), 0.0);
          pr = fmax(pr, fabs(g - s)); vmax = fmax(vmax, v); v1 += v;
          if (lb > -INF) {
            const double n = M.en[r], a = s - lb + n, z = r_zL[r], w = M.wn[r], D = a + z * n / w;
            const double c = (mu_ - a * z - z * (mu_ - n * w + n * (z + w - frho)) / w) / D;
            co = fmax(co, fmax(a * z, n * w)); cm = fmax(cm, fmax(fabs(a * z - mu_), fabs(n * w - mu_))); rn = fmax(rn, fabs(z + w - frho));
            sg += z / D; rh -= z + c; zs += z; nz += 1.0;
          }
          if (ub < INF) {
            const double q = M.ep[r], b = ub + q - s, z = r_zU[r], w = M.wp[r], D = b + z * q / w;
            const double c = (mu_ - b * z - z * (mu_ - q * w + q * (z + w - frho)) / w) / D;
            co = fmax(co, fmax(b * z, q * w)); cm = fmax(cm, fmax(fabs(b * z - mu_), fabs(q * w - mu_))); rn = fmax(rn, fabs(z + w - frho));
            sg += z / D; rh += z + c; zs += z; nz += 1.0;
          }
          rh += sg * (g - s);
        }
      }
      r_sig[r] = sg; r_rho[r] = rh;
    }
    double v[6] = {pr, co, cm, ys, zs, nz}; const int op[6] = {RMAX, RMAX, RMAX, RSUM, RSUM, RSUM};
    block_reduce<6>(v, op, S.red);
    double u[4] = {rn, vmax, v1, teq}; const int op4[4] = {RMAX, RMAX, RSUM, RSUM};
    block_reduce<4>(u, op4, S.red);
    KS_BEGIN_SYNCED()
      K.c_pr = v[0]; K.c_co = v[1]; K.c_cm = v[2]; K.c_ys = v[3]; K.c_zs = v[4]; K.c_nz = fmax(v[5], 1.0); K.c_rn = u[0]; K.f_vmax = u[1]; K.f_v1 = u[2]; K.f_theq = u[3];
      if (K.want_entry) { K.th_entry = u[2] + u[3]; K.want_entry = 0; }      // first pass of a phase: the violation it starts from
    KS_END();
  };
  auto point_pass = [&](double mu_) {
    if (K.feas) { feas_point_pass(mu_); return; }
    double pr = 0.0, co = 0.0, cm = 0.0, ys = 0.0, zs = 0.0, nz = 0.0;
    for (int rb = lane; rb < ng; rb += NT * RB) {
      double lbv[RB], ubv[RB], gv[RB], sv[RB], zlv[RB], zuv[RB], yv[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j) { const int r = rb + j * NT, rr = r < ng ? r : ng - 1; lbv[j] = S.bnd_lb[bidx(rr)]; ubv[j] = S.bnd_ub[bidx(rr)]; gv[j] = r_g[rr]; sv[j] = r_s[rr]; zlv[j] = r_zL[rr]; zuv[j] = r_zU[rr]; yv[j] = r_y[rr]; }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int r = rb + j * NT;
        if (r >= ng) continue;
        const double lb = lbv[j], ub = ubv[j];
        double sg = 0.0, rh = 0.0;
        if (r >= 12) {
          const double g = gv[j];
          ys += fabs(yv[j]);
          if (lb == ub) pr = fmax(pr, fabs(g - lb));
          else {
            const double s = sv[j];
            pr = fmax(pr, fabs(g - s));
            if (lb > -INF) { const double d = s - lb, rd = fast_rcp(d), zl = zlv[j]; co = fmax(co, d * zl); cm = fmax(cm, fabs(d * zl - mu_)); sg += zl * rd; rh -= mu_ * rd; zs += zl; nz += 1.0; }
            if (ub < INF) { const double d = ub - s, rd = fast_rcp(d), zu = zuv[j]; co = fmax(co, d * zu); cm = fmax(cm, fabs(d * zu - mu_)); sg += zu * rd; rh += mu_ * rd; zs += zu; nz += 1.0; }
            rh += sg * (g - s);
          }
        }
        r_sig[r] = sg; r_rho[r] = rh;
      }
    }
    double v[6] = {pr, co, cm, ys, zs, nz}; const int op[6] = {RMAX, RMAX, RMAX, RSUM, RSUM, RSUM};
    block_reduce<6>(v, op, S.red);
    KS_BEGIN_SYNCED() K.c_pr = v[0]; K.c_co = v[1]; K.c_cm = v[2]; K.c_ys = v[3]; K.c_zs = v[4]; K.c_nz = fmax(v[5], 1.0); KS_END();
  };

  if (lane == 0) {
    K.tp = A.prof ? (long long)wall_clock64() : 0;
    K.mu = o.mu_init; K.delta_last = 0.0; K.th_max = 0.0;
    K.c_pr = 0.0; K.c_co = 0.0; K.c_cm = 0.0; K.c_ys = 0.0; K.c_zs = 0.0; K.c_nz = 1.0;
    K.nfilt = 0; K.it = 0; K.status = LANDING_MAX_ITER; K.need_reg_streak = 0; K.nreset = 0; K.first_failed = 0;
    K.last_reset_it = 0; K.ncrawl = 0; K.clip_k_cur = o.clip_k; K.last_mu_it = 0; K.cutstreak = 0; K.wd_count = 0; K.force_step = 0;
    K.e_pr = 0; K.e_du = 0; K.e_co = 0;
    K.jamrun = 0; K.stag = 0; K.full_prev = 0; K.e_prev = 1e300;
    K.feas = 0; K.feas_used = 0; K.fact_failed = 0; K.lim = o.max_iter; K.fjam = 0; K.fstat = 0; K.v1_ref = 0.0; K.c_rn = 0.0; K.f_vmax = 0.0; K.f_v1 = 0.0;
    K.n_feas = 0; K.stalled = 0; K.polished = 0; K.want_entry = 0; K.th_entry = 0.0; K.f_theq = 0.0; K.hard_lim = o.max_iter > 0 ? 3 * o.max_iter : 0; K.fdc = o.feas_delta_dec > 0.0 ? o.feas_delta_dec : o.delta_dec;
  }
  __syncthreads();
  // the lane = stage phases are called by the lanes that have work only: the callee-saved registers an out-of-line function touches
  // are saved and restored in scratch memory by every lane that is active at the call
  const int nst = N > 36 ? N : 36;

  for (;;) {
    // ---------------------------------------------------------------- derivatives at (x, y)
    if (A.prof && lane == 0) K.tp = (long long)wall_clock64();
    // (round 2: the coalesced tile write-out of landing_sweep_kernel<0> was tried here for the Jacobian task -- eval_task_jac_tiled,
    // tiles in the dead G array -- and is slower: 0.049 vs 0.039 ms alone, 0.094 vs 0.085 under load; every lane then runs the
    // full middle-stage stream and the wave serialises on 24 tile flushes, while the scattered stores of this version drain
    // asynchronously behind the arithmetic of the other two waves)
    if ((lane & 63) < nst) member_eval_jh(L, M.x, p, M.y, M.J, M.H, M.gx, nullptr, nullptr, K.feas ? 0.0 : 1.0);
    __syncthreads();
    if (L.run_cost && !K.feas) { rc_add_grad(); __syncthreads(); }   // objective gradient of the stage variables (the terminal part is in member_eval_jh)
    PROF_ADD(PH_EVAL, K.tp);
    // ---------------------------------------------------------------- optimality error (unscaled)
    if (K.it == 0) point_pass(K.mu);
    {
      double du = 0.0;
      for (int i = lane + 12; i < nx; i += NT) du = fmax(du, fabs(M.gx[i]));
      du = block_reduce1(du, RMAX, S.red);
      KS_BEGIN_SYNCED()      // ---- what happens with this iterate: stop, restart, or another iteration
        const double pr = K.c_pr, co = K.c_co, mu = K.mu;
        const int it = K.it, nreset = K.nreset;
        if (K.feas) du = fmax(du, K.c_rn);      // the elastic problem has the extra stationarity rows rho_pen - z - w = 0
        K.e_pr = pr; K.e_du = du; K.e_co = co;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(LANDING_EMU_TRACE)
        if (getenv("LO_TRACE")) fprintf(stderr, "it %4d pr %9.2e du %9.2e co %9.2e mu %8.1e dlast %8.1e nreset %d nfilt %d feas %d fdc %.3f\n", it, pr, du, co, mu, K.delta_last, nreset, K.nfilt, K.feas, K.fdc);
#endif
        if (o.stag_relief > 0) {      // full Newton steps in the last barrier problem that do not halve the error: the proximal term is what holds them back
          const double E = fmax(pr, du);
          K.stag = (!K.feas && mu <= o.tol / 10.0 * 1.0000001 && K.full_prev && E > 0.5 * K.e_prev) ? K.stag + 1 : 0;
          K.e_prev = E;
        }
        int act = ACT_GO;
        bool give_up = false;
        if (K.feas) {
          // feasibility phase: a feasible point (or an elastic KKT point with negligible violation) restarts the solve from here, an elastic
          // KKT point with positive violation is the certificate of local infeasibility
          const bool conv = fmax(du, fmax(pr, co)) <= o.tol;
          if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { K.status = LANDING_NUMERICAL; act = ACT_STOP; }
          else if (K.f_vmax <= 1e-9 && pr <= o.tol) act = ACT_BACK;
          else if (conv && K.f_v1 > o.feas_cert) { K.status = LANDING_INFEASIBLE; act = ACT_STOP; }      // KKT point of the elastic problem with positive violation: the certificate
          else if (conv) act = ACT_BACK;
          else if (o.feas_back > 0.0 && !K.feas_used && K.f_v1 + K.f_theq <= o.feas_back * K.th_entry) act = ACT_BACK;      // the violation has come down: IPOPT leaves its restoration phase here
          else if (o.feas_stat > 0) {      // stationary violation (landing_nlp.h): NOT a certificate
            const double v1 = K.f_v1;
            if (K.fstat < 0 || !(fabs(v1 - K.v1_ref) <= 0.05 * K.v1_ref)) { K.v1_ref = v1; K.fstat = 0; } else K.fstat++;
            if (K.fstat >= o.feas_stat && mu <= 1e-4 && pr <= 1e-3) {
              if (v1 <= o.feas_cert) act = ACT_BACK;
              else if (o.feas_polish > 0.0 && !K.polished) {      // once: the regularisation drops -- a stationary point is then a few Newton steps from the elastic KKT point
                K.polished = 1; K.fstat = -1; K.delta_last = o.feas_polish / (o.feas_delta_dec > 0.0 ? o.feas_delta_dec : o.delta_dec); K.need_reg_streak = 2;
              }
              else if (o.feas_resume && !K.stalled) { K.stalled = 1; K.feas_used = 1; act = ACT_BACK; }      // the interior-point iteration resumes from this point, once
              else { K.status = LANDING_STALLED; act = ACT_STOP; }
            }
          }
          if (act == ACT_GO && it >= K.lim) act = ACT_STOP;
          if (act == ACT_BACK) {
            K.feas = 0; K.lim = it + (o.max_iter > 1 ? o.max_iter : 1); if (K.lim > K.hard_lim) K.lim = K.hard_lim; K.status = LANDING_MAX_ITER;
            K.mu = (o.feas_ret_push > 0.0 && o.feas_ret_mu > 0.0) ? o.feas_ret_mu : o.mu_init; K.fjam = 0; K.nfilt = 0; K.delta_last = 0.0; K.need_reg_streak = 0; K.wd_count = 0; K.th_max = 0.0; K.nreset = 0; K.last_reset_it = it; K.ncrawl = 0;
            K.cutstreak = 0; K.force_step = 0;
            K.it = it + 1;
          }
        }
        else if (K.fact_failed) { K.status = LANDING_NUMERICAL; K.fact_failed = 0; give_up = true; }      // no regularisation made the last step computable
        else if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { K.status = LANDING_NUMERICAL; give_up = true; }
        else if (fmax(du, fmax(pr, co)) <= o.tol) { K.status = LANDING_CONVERGED; act = ACT_STOP; }
        else if (it >= K.lim) give_up = true;      // (>=: K.it runs ahead of a limit set from a max_iter < 1)
        else if (du > o.reset_du && nreset >= o.max_resets && o.max_resets > 0) { K.status = LANDING_NUMERICAL; give_up = true; }   // jammed again: give up
        else if (o.feas_jam > 0 && K.fjam >= o.feas_jam && pr > 1e-3 && o.feas_phase) { give_up = true; if (K.feas_used) K.stalled = 1; }      // jammed line search: the feasibility phase starts now (landing_nlp.h); no entry left: the solve ends here, status 4
        else {
          // crawling: still in the first barrier problem (mu never decreased) restart_period iterations after the last (re)start
          const bool stalled = o.restart_period > 0 && it - K.last_reset_it >= o.restart_period && mu >= o.mu_init && nreset < o.max_resets && K.ncrawl < ((o.fresh_restart & 4) ? 2 : 1);
          const bool overreg = o.reset_delta > 0.0 && K.delta_last > o.reset_delta && nreset < o.max_resets;
          // a LATER barrier problem not solved 2 restart_period iterations after it began has wandered off (the dual infeasibility stays
          // far below reset_du, nothing else catches it): restarted in place like a crawling iterate
          const bool lost = (o.fresh_restart & 8) && o.restart_period > 0 && mu < o.mu_init && pr > 1e-3 && nreset < o.max_resets &&
                            ((it - K.last_mu_it >= 2 * o.restart_period && it - K.last_reset_it >= o.restart_period) || K.wd_count >= 3);     // ... or crawls on although the watchdog has fired three times
          if ((du > o.reset_du && nreset < o.max_resets) || stalled || overreg || lost) {
            act = ACT_RESET;
            K.last_reset_it = it;
            if (stalled) K.ncrawl++;
            // jammed iterate (multipliers blown up): keep x, re-initialise slacks, multipliers, barrier parameter and
            // filter -- the role IPOPT's restoration phase plays on this problem class
            K.nreset = nreset + 1;
            // the restart in place did not help (second restart) or the iterate is jammed: back to the caller's initial guess with
            // another step rule (landing_nlp.h: the members that fail from it with clip_k = 4 solve with clip_k = 2)
            K.fresh = (((o.fresh_restart & 2) && nreset + 1 == 2) || ((o.fresh_restart & 1) && nreset + 1 == 1 && !stalled && !lost)) ? 1 : 0;
            if (K.fresh) { if (K.clip_k_cur > 1) K.clip_k_cur = 2; K.th_max = 0.0; }
            K.mu = o.mu_init; K.nfilt = 0; K.delta_last = 0.0; K.need_reg_streak = 0; K.wd_count = 0;
            K.it = it + 1;
          }
        }
        if (give_up) {
          // the solve would end here as NUMERICAL / MAX_ITER: enter the feasibility phase once (landing_solver_opts::feas_phase)
          if (K.stalled) K.status = LANDING_STALLED;      // the point a stalled phase handed back did not lead anywhere either
          if (o.feas_phase && !K.feas_used && o.max_iter > 0 && it < K.hard_lim) {      // (max_iter < 1: the caller asked for no iteration at all, the phase would get none either)
            act = ACT_FEAS;
            K.feas = 1; K.n_feas++; K.feas_used = K.n_feas >= o.feas_max ? 1 : 0; K.status = LANDING_MAX_ITER; K.lim = it + o.max_iter; if (K.lim > K.hard_lim) K.lim = K.hard_lim; K.fstat = -1; K.fjam = 0; K.want_entry = 1; K.fdc = o.feas_delta_dec > 0.0 ? o.feas_delta_dec : o.delta_dec;
            K.mu = o.mu_init; K.nfilt = 0; K.th_max = 0.0; K.delta_last = 0.0; K.need_reg_streak = 0; K.cutstreak = 0; K.force_step = 0; K.wd_count = 0;
            K.it = it + 1;
          } else act = ACT_STOP;
        }
        K.action = act;
      KS_END();
    }
    if (K.action == ACT_STOP) break;
    if (K.action == ACT_FEAS) {
      // ---- enter the feasibility phase from the current point (from the caller's initial guess when the iterate is not finite)
      double big = 0.0;
      for (int i = lane; i < nx; i += NT) { const double v = fabs(M.x[i]); big = fmax(big, v < 1e6 ? v : 1e300); }
      big = block_reduce1(big, RMAX, S.red);
      if (!(big < 1e6)) {
        for (int i = lane; i < nx; i += NT) {
          double v = A.x0[(size_t)m * nx + i];
          if (i < 6) v = p[L.o_q_init + i]; else if (i < 12) v = p[L.o_qd_init + i - 6];
          M.x[i] = v;
        }
      }
      __syncthreads();
      member_eval_g_rare(L, M.x, p, M.g);
      if (L.run_cost) for (int e = lane; e < N * RUNC; e += NT) M.Hc[e] = 0.0;      // no objective: its constant Hessian entries leave the condensation
      __syncthreads();
      {
        const double frho = o.feas_rho, mu0 = o.mu_init;
        for (int r = lane + 12; r < ng; r += NT) {
          const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)], g = M.g[r];
          if (lb == ub) { M.y[r] = 0.0; continue; }
          // slack on the row value; violation variables sized so that both distances start at a comfortable value
          double zl = 0.0, zu = 0.0, n0 = 0.0, q0 = 0.0, wl = 0.0, wu = 0.0;
          if (lb > -INF) { const double v = lb - g; n0 = fmax(v, 0.0) + fmax(1e-2, 0.1 * fabs(v)); zl = fmin(mu0 / (g - lb + n0), 0.5 * frho); wl = frho - zl; }
          if (ub < INF) { const double v = g - ub; q0 = fmax(v, 0.0) + fmax(1e-2, 0.1 * fabs(v)); zu = fmin(mu0 / (ub + q0 - g), 0.5 * frho); wu = frho - zu; }
          M.s[r] = g; M.en[r] = n0; M.ep[r] = q0; M.zL[r] = zl; M.zU[r] = zu; M.wn[r] = wl; M.wp[r] = wu; M.y[r] = zu - zl;
        }
      }
      __syncthreads();
      point_pass(K.mu);
      continue;
    }
    if (K.action == ACT_BACK) {
      // ---- a feasible point (or negligible violation): the interior-point solve restarts from it
      if (L.run_cost) rc_init_hc();
      for (int r = lane + 12; r < ng; r += NT) if (S.bnd_lb[bidx(r)] == S.bnd_ub[bidx(r)]) M.y[r] = 0.0;
      __syncthreads();
      if (o.feas_ret_push > 0.0) init_slacks_return(K.mu); else init_slacks();
      point_pass(K.mu);
      continue;
    }
    if (K.action == ACT_RESET) {
      if (K.fresh) {
        for (int i = lane; i < nx; i += NT) {
          double v = A.x0[(size_t)m * nx + i];
          if (i < 6) v = p[L.o_q_init + i]; else if (i < 12) v = p[L.o_qd_init + i - 6];
          M.x[i] = v;
        }
        __syncthreads();
        member_eval_g_rare(L, M.x, p, M.g);
        __syncthreads();
      }
      init_slacks();
      point_pass(K.mu);
      continue;
    }
    // ---------------------------------------------------------------- barrier parameter (monotone)
    for (;;) {
      KS_BEGIN()
        double sd = 1.0, sc = 1.0;      // IPOPT's scaling of the optimality error in the barrier-subproblem test (landing_nlp.h)
        if (o.barrier_smax > 0.0) {
          sd = fmax(o.barrier_smax, (K.c_ys + K.c_zs) / ((double)(ng - 12) + K.c_nz)) / o.barrier_smax;
          sc = fmax(o.barrier_smax, K.c_zs / K.c_nz) / o.barrier_smax;
        }
        const double mu = K.mu;
        if (fmax(K.e_du / sd, fmax(K.c_pr, K.c_cm / sc)) <= o.kappa_eps * mu && mu > o.tol / 10.0) {
          K.mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
          K.nfilt = 0; K.last_mu_it = K.it; K.wd_count = 0;
          K.flag = 1;
        } else {
          K.flag = 0;
          K.tau = fmax(o.tau_min, 1.0 - mu);
        }
      KS_END();
      if (!K.flag) break;
      point_pass(K.mu);                     // complementarity error, Sigma, rho for the new mu
    }
    PROF_ADD(PH_ERR, K.tp);

    // (the condensation G_k = H_k + J_d^T Sigma J_d, gamma_k, A^_k is part of the backward sweep: cond_issue / cond_finish)
    { const int rc = (L.run_cost && !K.feas) ? 1 : 0; if (lane == 0) S.rc_on = rc; if (rc) rc_add_gamma(); __syncthreads(); }   // gradient of the running cost for the stage right-hand sides
    PROF_ADD(PH_SIGRHO, K.tp);
    // ================================================================ Riccati factorisation with inertia correction
    // IPOPT's inertia-correction schedule (delta_w = 0 first, then max(1e-20, delta_last/3), then x8 / x100),
    // except that an iteration following a regularised one starts from delta_last/3 directly when the
    // unregularised attempt failed twice in a row (saves one full factorisation in nonconvex phases)
    // (o.sticky_delta = 1: when the first trial of the previous iteration failed, start from delta_last itself)
    KS_BEGIN()
      const double dl = K.delta_last;
      const double ddec = (K.feas && o.feas_delta_dec > 0.0) ? K.fdc : o.delta_dec;      // (the elastic problem has flat directions: its regularisation has to fall faster, landing_nlp.h)
      K.delta = (K.need_reg_streak >= 2 && dl > 0.0) ? fmax(1e-20, dl * ((o.sticky_delta && K.first_failed) ? 1.0 : ddec)) : 0.0;
      if (!L.run_cost && !K.feas) {      // proximal term of the terminal-cost form (landing_nlp.h)
        double fl = o.delta_floor;
        if (o.stag_relief > 0 && K.stag >= o.stag_relief) {      // ... a tenth of it per stagnating iteration
          for (int e = K.stag - o.stag_relief; e >= 0 && fl >= 1e-12; --e) fl *= 0.1;
          if (fl < 1e-12) fl = 0.0;
        }
        K.delta = fmax(K.delta, fl);
      }
      K.skipped_zero = K.delta > 0.0;
      K.fact_ok = 0; K.attempt = 0; K.flag = 1;
      S.prof[PH_NFACT] += 1.0;
    KS_END();
    for (;;) {
      const bool ok = riccati_backward(K.delta);
      KS_BEGIN()
        K.fact_ok = ok ? 1 : 0;
        if (K.attempt == 0) K.first_failed = (K.skipped_zero && !ok) ? 1 : 0;
        K.attempt++;
        K.flag = 0;
        if (!ok && K.attempt < 60) {      // next attempt with a larger regularisation
          double d = K.delta; const double dl = K.delta_last;
          if (d == 0.0) d = (dl == 0.0) ? o.delta_init : fmax(1e-20, dl * ((K.feas && o.feas_delta_dec > 0.0) ? K.fdc : o.delta_dec));
          else if (K.feas && o.feas_delta_dec > 0.0 && K.attempt == 1 && d < dl) d = dl;      // the regularisation of the last iteration is the best guess of what this one needs
          else d *= (dl == 0.0 ? o.delta_inc_first : o.delta_inc);
          if (!(d > 1e40)) { K.delta = d; K.flag = 1; S.prof[PH_NFACT] += 1.0; }
        }
      KS_END();
      if (!K.flag) break;
    }
    if (!K.fact_ok) {      // the step cannot be computed: give up (status NUMERICAL), or -- once -- continue in the feasibility phase
      KS_BEGIN() K.status = K.stalled ? LANDING_STALLED : LANDING_NUMERICAL; K.fact_failed = (o.feas_phase && !K.feas_used && !K.feas) ? 1 : 0; KS_END();
      if (K.fact_failed) continue;
      break;
    }
    KS_BEGIN()
      const bool adapt = K.feas && o.feas_delta_dec > 0.0;
      if (adapt) K.fdc = K.attempt <= 1 ? fmax(o.feas_delta_dec, K.fdc * K.fdc) : fmin(0.7, sqrt(K.fdc));      // (attempt counts the factorisations of this iteration)
      if (K.delta > 0.0) { K.delta_last = K.delta; K.need_reg_streak++; } else K.need_reg_streak = 0;
      if (K.need_reg_streak > 8) K.need_reg_streak = adapt ? 2 : 0;      // probe delta = 0 again from time to time (not inside the phase: the elastic problem has no objective)
    KS_END();
    PROF_ADD(PH_BACK, K.tp);

    forward_pass();
#ifndef LANDING_DEV_SKIP_ROWP      // development probe (timing only: the results are garbage without it)
    row_products(A.rterm, A.rlen);
#endif

    PROF_ADD(PH_FWD, K.tp);
    // ================================================================ dual steps, step bounds, merit data
    // fraction-to-the-boundary as tau / max(-ds/d), tau / max(-dz/z): reciprocals instead of divisions in the row
    // loop, one logarithm per row (log of the product of the two distances)
    // clip_k rule (landing_nlp.h): while the point is far from feasible the step length comes from the clip_k-th largest
    // ratio |ds| / distance; the slacks with a larger one stop at (1 - tau) of their distance (omt > 0 in the passes below)
    {
      const double mu = K.mu;
      const bool feas = K.feas != 0;
      const bool clip_now = !feas && K.clip_k_cur > 1 && (K.c_pr > o.clip_until || (o.jam_clip > 0 && K.jamrun >= o.jam_clip));      // far from feasible, or jammed (landing_nlp.h)
      double top[4] = {0.0, 0.0, 0.0, 0.0};
      double m_pr = 0.0, m_du = 0.0, th0 = 0.0, bar = 0.0, dphi = 0.0;
      double f0 = 0.0;
      if (feas) {      // elastic rows: steps of the eliminated variables, step bounds (a, n, b, q and their multipliers stay positive), merit data
        const double frho = o.feas_rho;
        for (int r = lane + 12; r < ng; r += NT) {
          const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)], g = r_g[r];
          if (lb == ub) { th0 += fabs(g - lb); continue; }
          const double s = r_s[r];
          double ds = r_ds[r];
          if (r >= 36) { ds = ds + (g - s); r_ds[r] = ds; }      // row_products left J_d dx: the row's `+ (g - s)` is added here (the terminal rows 12..35 are complete)
          th0 += fabs(g - s);
          if (lb > -INF) {
            const double n = M.en[r], a = s - lb + n, z = r_zL[r], w = M.wn[r];
            const ElStep e = el_step(1.0, a, n, z, w, mu, frho, ds);
            m_pr = fmax(m_pr, fmax(-e.da / a, -e.dn / n)); m_du = fmax(m_du, fmax(-e.dz / z, -e.dw / w));
            bar -= log(a * n); dphi += frho * e.dn - mu * (e.da / a + e.dn / n); f0 += frho * n;
          }
          if (ub < INF) {
            const double q = M.ep[r], b = ub + q - s, z = r_zU[r], w = M.wp[r];
            const ElStep e = el_step(-1.0, b, q, z, w, mu, frho, ds);
            m_pr = fmax(m_pr, fmax(-e.da / b, -e.dn / q)); m_du = fmax(m_du, fmax(-e.dz / z, -e.dw / w));
            bar -= log(b * q); dphi += frho * e.dn - mu * (e.da / b + e.dn / q); f0 += frho * q;
          }
        }
      } else
      for (int rb = lane + 12; rb < ng; rb += NT * RB) {
        double lbv[RB], ubv[RB], gv[RB], sv[RB], dsv[RB], zlv[RB], zuv[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) { const int r = rb + j * NT, rr = r < ng ? r : ng - 1; lbv[j] = S.bnd_lb[bidx(rr)]; ubv[j] = S.bnd_ub[bidx(rr)]; gv[j] = r_g[rr]; sv[j] = r_s[rr]; dsv[j] = r_ds[rr]; zlv[j] = r_zL[rr]; zuv[j] = r_zU[rr]; }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          if (rb + j * NT >= ng) continue;
          const double lb = lbv[j], ub = ubv[j], g = gv[j];
          if (lb == ub) { th0 += fabs(g - lb); continue; }
          const double s = sv[j];
          double ds = dsv[j];
          { const int r = rb + j * NT; if (r >= 36) { ds = ds + (g - s); r_ds[r] = ds; } }      // row_products left J_d dx: the row's `+ (g - s)` is added here (the terminal rows 12..35 are complete)
          th0 += fabs(g - s);
          double dprod = 1.0;
          if (lb > -INF) {
            const double d = s - lb, rd = fast_rcp(d), zl = zlv[j];
            const double dz = fma(-zl * rd, ds, mu * rd - zl);
            m_pr = fmax(m_pr, -ds * rd); top4_push(top, -ds * rd);
            m_du = fmax(m_du, -dz * fast_rcp(zl));
            dprod = d; dphi -= mu * ds * rd;
          }
          if (ub < INF) {
            const double d = ub - s, rd = fast_rcp(d), zu = zuv[j];
            const double dz = fma(zu * rd, ds, mu * rd - zu);
            m_pr = fmax(m_pr, ds * rd); top4_push(top, ds * rd);
            m_du = fmax(m_du, -dz * fast_rcp(zu));
            dprod *= d; dphi += mu * ds * rd;
          }
          bar -= log(dprod);
        }
      }
      if (lane < 12 && !feas) {
        const double d = M.x[12 * N + lane] - p[12 * N + lane], qn = p[L.o_QN + lane];
        f0 = qn * d * d; dphi += 2.0 * qn * d * M.dx[12 * N + lane];
      }
      if (L.run_cost && !feas) rc_f_dphi(f0, dphi);
      double v[6] = {m_pr, m_du, th0, bar, dphi, f0}; const int op[6] = {RMAX, RMAX, RSUM, RSUM, RSUM, RSUM};
      block_reduce<6>(v, op, S.red);
      if (clip_now) block_top4(top, S.red);         // (uniform: clip_now comes from K)
      KS_BEGIN_SYNCED()
        const double tau = K.tau;
        double a_pr = (v[0] > tau) ? tau / v[0] : 1.0;
        K.a_du = (v[1] > tau) ? tau / v[1] : 1.0;
        if (clip_now) {
          const double rk = top[(K.clip_k_cur > 4 ? 4 : K.clip_k_cur) - 1];
          a_pr = (rk > tau) ? tau / rk : 1.0;
        }
        K.a_pr = a_pr;
        K.clip_now = clip_now ? 1 : 0; K.omt = clip_now ? 1.0 - tau : -1.0;
        K.th0 = v[2]; K.dphi = v[4]; K.ph0 = v[5] + mu * v[3];
        if (K.th_max == 0.0) K.th_max = 1e4 * fmax(1.0, v[2]);
        // line search state
        K.alpha = a_pr; K.s_corr = 0.0; K.accepted = 0; K.armijo_step = 0; K.ls_done = a_pr > 1e-10 ? 0 : 1;
      KS_END();
    }
    PROF_ADD(PH_DUAL, K.tp);
    // ================================================================ filter line search
    while (!K.ls_done) {
      const double alpha = K.alpha, omt = K.omt, mu = K.mu;
      if (lane == 0) S.prof[PH_NTRIAL] += 1.0;
      for (int i = lane; i < nx; i += NT) M.xt[i] = M.x[i] + alpha * M.dx[i];
      __syncthreads();
      if (lane < nst) member_eval_g(L, M.xt, p, M.gt);
      __syncthreads();
      double tht = 0.0, bt = 0.0, ft = 0.0;
      const bool feas = K.feas != 0;
      if (feas) {      // elastic rows at the trial step length: theta over all rows, merit = rho_pen (n + q) - mu sum of logs (mu applied below)
        const double frho = o.feas_rho;
        for (int r = lane + 12; r < ng; r += NT) {
          const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)], g = r_gt[r];
          if (lb == ub) { tht += fabs(g - lb); continue; }
          const double s0 = r_s[r], ds = r_ds[r], s = s0 + alpha * ds;
          tht += fabs(g - s);
          if (lb > -INF) { const double n0 = M.en[r]; const ElStep e = el_step(1.0, s0 - lb + n0, n0, r_zL[r], M.wn[r], mu, frho, ds); const double n = n0 + alpha * e.dn; bt -= log((s - lb + n) * n); ft += frho * n; }
          if (ub < INF) { const double q0 = M.ep[r]; const ElStep e = el_step(-1.0, ub + q0 - s0, q0, r_zU[r], M.wp[r], mu, frho, ds); const double q = q0 + alpha * e.dn; bt -= log((ub + q - s) * q); ft += frho * q; }
        }
      } else
      for (int rb = lane + 12; rb < ng; rb += NT * RB) {
        double lbv[RB], ubv[RB], gv[RB], sv[RB], dsv[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) { const int r = rb + j * NT, rr = r < ng ? r : ng - 1; lbv[j] = S.bnd_lb[bidx(rr)]; ubv[j] = S.bnd_ub[bidx(rr)]; gv[j] = r_gt[rr]; sv[j] = r_s[rr]; dsv[j] = r_ds[rr]; }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          if (rb + j * NT >= ng) continue;
          const double lb = lbv[j], ub = ubv[j], g = gv[j];
          if (lb == ub) { tht += fabs(g - lb); continue; }
          double s = sv[j] + alpha * dsv[j];
          if (omt > 0.0) { if (lb > -INF) s = fmax(s, fma(omt, sv[j] - lb, lb)); if (ub < INF) s = fmin(s, fma(-omt, ub - sv[j], ub)); }
          tht += fabs(g - s);
          bt -= log((lb > -INF ? s - lb : 1.0) * (ub < INF ? ub - s : 1.0));
        }
      }
      if (lane < 12 && !feas) { const double d = M.xt[12 * N + lane] - p[12 * N + lane]; ft = p[L.o_QN + lane] * d * d; }
      if (L.run_cost && !feas) ft += rc_f(M.xt);
      { double v[3] = {tht, bt, ft}; const int op[3] = {RSUM, RSUM, RSUM}; block_reduce<3>(v, op, S.red); tht = v[0]; bt = v[1]; ft = v[2]; }
      KS_BEGIN_SYNCED()
        const double th_min = 1e-4, th_floor = o.theta_floor * o.tol;     // violations below the tolerance count as equal (landing_nlp.h)
        const double th0 = K.th0, ph0 = K.ph0, dphi = K.dphi;
        const int nfilt = K.nfilt;
        const double pht = ft + mu * bt;
        bool ok_f = (tht <= K.th_max) && (pht < 1e300) && (pht > -1e300) && (tht < 1e300);
        for (int e = 0; e < nfilt && ok_f; ++e) if (tht >= fmax(S.filt_th[e], th_floor) && pht >= S.filt_ph[e]) ok_f = false;
        const bool switching = (dphi < 0.0) && (th0 <= th_min) && (alpha * pow(-dphi, 2.3) > 1.0 * pow(th0, 1.1));
        bool accepted = false, done = false;
        if (ok_f) {
          if (switching) {
            if (pht <= ph0 + 1e-8 * alpha * dphi) { accepted = true; K.armijo_step = 1; }
          } else if (tht <= fmax((1.0 - 1e-5) * th0, th_floor) || pht <= ph0 - 1e-8 * th0) {
            accepted = true;
          }
        }
        if (K.force_step && ok_f) { accepted = true; K.nfilt = 0; done = true; }      // watchdog (landing_nlp.h): the step to the boundary is taken without the sufficient-decrease / switching tests;
                                                                                      // it must still pass theta <= theta_max and must not be dominated by a filter entry (ok_f), then the filter restarts
        if (accepted) done = true;
        K.need_corr = 0;
        if (!done) {
          if (o.slack_corr > 0.0 && !K.feas && alpha == K.a_pr && tht >= th0) { K.need_corr = 1; K.ft = ft; }
          else { K.alpha = alpha * 0.5; if (!(K.alpha > 1e-10)) done = true; }
        }
        K.accepted = accepted ? 1 : 0;
        K.ls_done = done ? 1 : 0;
      KS_END();
      if (K.need_corr) {
        // slack correction (landing_nlp.h): the rejected first trial point once more with the inequality slacks moved to g(x_trial)
        double tht2 = 0.0, bt2 = 0.0;
        for (int rb = lane + 12; rb < ng; rb += NT * RB) {
          double lbv[RB], ubv[RB], gv[RB], sv[RB], dsv[RB];
#pragma unroll
          for (int j = 0; j < RB; ++j) { const int r = rb + j * NT, rr = r < ng ? r : ng - 1; lbv[j] = S.bnd_lb[bidx(rr)]; ubv[j] = S.bnd_ub[bidx(rr)]; gv[j] = r_gt[rr]; sv[j] = r_s[rr]; dsv[j] = r_ds[rr]; }
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            if (rb + j * NT >= ng) continue;
            const double lb = lbv[j], ub = ubv[j], g = gv[j];
            if (lb == ub) { tht2 += fabs(g - lb); continue; }
            double s = sv[j] + alpha * dsv[j];
            if (omt > 0.0) { if (lb > -INF) s = fmax(s, fma(omt, sv[j] - lb, lb)); if (ub < INF) s = fmin(s, fma(-omt, ub - sv[j], ub)); }
            const double lo = lb > -INF ? lb + o.slack_corr * (s - lb) : -INF, hi = ub < INF ? ub - o.slack_corr * (ub - s) : INF;
            s = fmin(fmax(g, lo), hi);
            tht2 += fabs(g - s);
            bt2 -= log((lb > -INF ? s - lb : 1.0) * (ub < INF ? ub - s : 1.0));
          }
        }
        { double v[2] = {tht2, bt2}; const int op[2] = {RSUM, RSUM}; block_reduce<2>(v, op, S.red); tht2 = v[0]; bt2 = v[1]; }
        KS_BEGIN_SYNCED()
          const double th_floor = o.theta_floor * o.tol, th0 = K.th0, ph0 = K.ph0;
          const int nfilt = K.nfilt;
          const double pht2 = K.ft + mu * bt2;
          bool ok2 = (tht2 <= K.th_max) && (pht2 < 1e300) && (pht2 > -1e300);
          for (int e = 0; e < nfilt && ok2; ++e) if (tht2 >= fmax(S.filt_th[e], th_floor) && pht2 >= S.filt_ph[e]) ok2 = false;
          if (ok2 && (tht2 <= fmax((1.0 - 1e-5) * th0, th_floor) || pht2 <= ph0 - 1e-8 * th0)) { K.accepted = 1; K.s_corr = o.slack_corr; K.ls_done = 1; }
          else { K.alpha = alpha * 0.5; if (!(K.alpha > 1e-10)) K.ls_done = 1; }
        KS_END();
      }
    }
    KS_BEGIN()
      const double a_pr = K.a_pr;
      K.force_step = 0;
      if (o.watchdog > 0 && !K.feas) {      // successive iterations with step lengths <= 1/16 of the step to the boundary arm the watchdog
        if (K.accepted && K.alpha <= 0.0625 * a_pr) { if (++K.cutstreak >= o.watchdog) { K.force_step = 1; K.cutstreak = 0; K.wd_count++; } }
        else K.cutstreak = 0;
      }
      K.fallback = 0;
      if (!K.accepted) {
        // no acceptable step: take a short step along the Newton direction and restart the filter
        K.nfilt = 0;
        K.alpha = fmin(a_pr, o.alpha_fallback);
        K.fallback = 1;
      } else if (!K.armijo_step) {
        int nfilt = K.nfilt;
        if (nfilt == FILT_CAP) {     // drop the oldest entry (serial, rare)
          for (int e = 0; e + 1 < FILT_CAP; ++e) { S.filt_th[e] = S.filt_th[e + 1]; S.filt_ph[e] = S.filt_ph[e + 1]; }
          nfilt = FILT_CAP - 1;
        }
        S.filt_th[nfilt] = (1.0 - 1e-5) * K.th0; S.filt_ph[nfilt] = K.ph0 - 1e-8 * K.th0;
        K.nfilt = nfilt + 1;
      }
      if (o.dual_step_cap > 0.0) K.a_du = fmin(K.a_du, o.dual_step_cap * K.alpha);      // the multipliers do not run ahead of a blocked primal step (landing_nlp.h)
      if (o.jam_clip > 0) K.jamrun = (!K.clip_now && K.a_pr < 0.02) ? K.jamrun + 1 : 0;
      if (o.feas_jam > 0) K.fjam = (!K.feas && K.alpha < 1e-2) ? K.fjam + 1 : (K.fjam > 2 ? K.fjam - 2 : 0);
      K.full_prev = (K.accepted && K.alpha == 1.0 && K.a_du == 1.0 && K.attempt <= 1) ? 1 : 0;
    KS_END();
    if (K.fallback) {
      const double alpha = K.alpha;
      for (int i = lane; i < nx; i += NT) M.xt[i] = M.x[i] + alpha * M.dx[i];
      __syncthreads();
      member_eval_g_rare(L, M.xt, p, M.gt);
      __syncthreads();
    }
    PROF_ADD(PH_LS, K.tp);
    // ================================================================ accept the trial point; the same pass produces the
    // primal / complementarity errors, Sigma and rho of the new iterate (what point_pass computes)
    for (int i = lane; i < nx; i += NT) M.x[i] = M.xt[i];
    if (K.feas) {      // elastic rows: primal variables with alpha, multipliers with a_du (kept inside the kappa_Sigma band), then the quantities of the new point
      const double alpha = K.alpha, a_du = K.a_du, mu = K.mu, frho = o.feas_rho;
      for (int r = lane; r < ng; r += NT) {
        const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)];
        r_g[r] = r_gt[r];
        if (r < 12) continue;
        if (lb == ub) { r_y[r] = r_y[r] + alpha * (r_yn[r] - r_y[r]); continue; }
        const double s0 = r_s[r], ds = r_ds[r], s = s0 + alpha * ds;
        double zl = 0.0, zu = 0.0;
        if (lb > -INF) {
          const double n0 = M.en[r], z0 = r_zL[r], w0 = M.wn[r];
          const ElStep e = el_step(1.0, s0 - lb + n0, n0, z0, w0, mu, frho, ds);
          const double n = n0 + alpha * e.dn, a = s - lb + n;
          zl = fmin(fmax(z0 + a_du * e.dz, 1e-10 * mu / a), 1e10 * mu / a);
          M.wn[r] = fmin(fmax(w0 + a_du * e.dw, 1e-10 * mu / n), 1e10 * mu / n); M.en[r] = n;
        }
        if (ub < INF) {
          const double q0 = M.ep[r], z0 = r_zU[r], w0 = M.wp[r];
          const ElStep e = el_step(-1.0, ub + q0 - s0, q0, z0, w0, mu, frho, ds);
          const double q = q0 + alpha * e.dn, b = ub + q - s;
          zu = fmin(fmax(z0 + a_du * e.dz, 1e-10 * mu / b), 1e10 * mu / b);
          M.wp[r] = fmin(fmax(w0 + a_du * e.dw, 1e-10 * mu / q), 1e10 * mu / q); M.ep[r] = q;
        }
        r_s[r] = s; r_zL[r] = zl; r_zU[r] = zu; r_y[r] = zu - zl;
      }
      __syncthreads();
      feas_point_pass(mu);
      KS_BEGIN() K.it++; KS_END();
    } else {
      const double alpha = K.alpha, a_du = K.a_du, omt = K.omt, mu = K.mu, s_corr = K.s_corr;
      double npr = 0.0, nco = 0.0, ncm = 0.0, nys = 0.0, nzs = 0.0, nnz = 0.0;
      for (int rb = lane; rb < ng; rb += NT * RB) {
        double lbv[RB], ubv[RB], gv[RB], sv[RB], dsv[RB], zlv[RB], zuv[RB], yv[RB], ynv[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int r = rb + j * NT, rr = r < ng ? r : ng - 1;
          lbv[j] = S.bnd_lb[bidx(rr)]; ubv[j] = S.bnd_ub[bidx(rr)]; gv[j] = r_gt[rr]; sv[j] = r_s[rr]; dsv[j] = r_ds[rr]; zlv[j] = r_zL[rr]; zuv[j] = r_zU[rr]; yv[j] = r_y[rr]; ynv[j] = r_yn[rr];
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int r = rb + j * NT;
          if (r >= ng) continue;
          const double lb = lbv[j], ub = ubv[j], g = gv[j];
          r_g[r] = g;
          double sg = 0.0, rh = 0.0;
          if (r >= 12) {
            if (lb == ub) { const double yn_ = yv[j] + alpha * (ynv[j] - yv[j]); r_y[r] = yn_; nys += fabs(yn_); npr = fmax(npr, fabs(g - lb)); }
            else {
              const double so = sv[j], ds = dsv[j];
              double s = so + alpha * ds;
              if (omt > 0.0) { if (lb > -INF) s = fmax(s, fma(omt, so - lb, lb)); if (ub < INF) s = fmin(s, fma(-omt, ub - so, ub)); }
              if (s_corr > 0.0) {
                const double lo = lb > -INF ? lb + s_corr * (s - lb) : -INF, hi = ub < INF ? ub - s_corr * (ub - s) : INF;
                s = fmin(fmax(g, lo), hi);
              }
              double zl = 0.0, zu = 0.0;
              if (lb > -INF) {
                const double dold = so - lb, ro = fast_rcp(dold), zo = zlv[j], dz = fma(-zo * ro, ds, mu * ro - zo), d = s - lb, rd = fast_rcp(d);
                zl = fmin(fmax(zo + a_du * dz, 1e-10 * mu * rd), 1e10 * mu * rd);
                nco = fmax(nco, d * zl); ncm = fmax(ncm, fabs(d * zl - mu)); sg += zl * rd; rh -= mu * rd; nzs += zl; nnz += 1.0;
              }
              if (ub < INF) {
                const double dold = ub - so, ro = fast_rcp(dold), zo = zuv[j], dz = fma(zo * ro, ds, mu * ro - zo), d = ub - s, rd = fast_rcp(d);
                zu = fmin(fmax(zo + a_du * dz, 1e-10 * mu * rd), 1e10 * mu * rd);
                nco = fmax(nco, d * zu); ncm = fmax(ncm, fabs(d * zu - mu)); sg += zu * rd; rh += mu * rd; nzs += zu; nnz += 1.0;
              }
              npr = fmax(npr, fabs(g - s));
              rh += sg * (g - s);
              r_s[r] = s; r_zL[r] = zl; r_zU[r] = zu; r_y[r] = zu - zl; nys += fabs(zu - zl);
            }
          }
          r_sig[r] = sg; r_rho[r] = rh;
        }
      }
      double v[6] = {npr, nco, ncm, nys, nzs, nnz}; const int op[6] = {RMAX, RMAX, RMAX, RSUM, RSUM, RSUM};
      block_reduce<6>(v, op, S.red);
      KS_BEGIN_SYNCED()
        K.c_pr = v[0]; K.c_co = v[1]; K.c_cm = v[2]; K.c_ys = v[3]; K.c_zs = v[4]; K.c_nz = fmax(v[5], 1.0);
        K.it++;
      KS_END();
    }
    PROF_ADD(PH_ACCEPT, K.tp);
  }
  __syncthreads();
  if (A.prof && lane == 0) { S.prof[PH_NITER] = (double)K.it; for (int i = 0; i < PH_COUNT; ++i) A.prof[(size_t)m * PH_COUNT + i] = S.prof[i]; }

  // -------------------------------------------------------------------- outputs
  // multipliers of the initial-state rows from stationarity of X(:,1): lam = -(grad f + J^T y)
  double fo = 0.0;
  if (lane < 12) { M.y[lane] = -M.gx[lane]; const double d = M.x[12 * N + lane] - p[12 * N + lane]; fo = p[L.o_QN + lane] * d * d; }
  if (L.run_cost) fo += rc_f(M.x);
  fo = block_reduce1(fo, RSUM, S.red);
  // reference-consistent KKT residual: max_viol(g), ||grad f + J^T lam||_inf, |lam * dist|
  double kp = 0.0, kc = 0.0;
  for (int r = lane; r < ng; r += NT) {
    const double lb = S.bnd_lb[bidx(r)], ub = S.bnd_ub[bidx(r)], g = M.g[r], lam = M.y[r];
    kp = fmax(kp, fmax(lb - g, fmax(g - ub, 0.0)));
    if (lb != ub) {
      const double dist = lam > 0.0 ? ub - g : g - lb;
      if (lam != 0.0 && dist < INF) kc = fmax(kc, fabs(lam * dist));
    }
  }
  { double v[2] = {kp, kc}; const int op[2] = {RMAX, RMAX}; block_reduce<2>(v, op, S.red); kp = v[0]; kc = v[1]; }
  for (int i = lane; i < nx; i += NT) A.x_out[(size_t)m * nx + i] = M.x[i];
  if (A.lam_out) for (int r = lane; r < ng; r += NT) A.lam_out[(size_t)m * ng + r] = M.y[r];
  if (lane == 0) {
    if (A.f_out) A.f_out[m] = fo;
    if (A.status) A.status[m] = K.status;
    if (A.iters) A.iters[m] = K.it;
    if (A.kkt) { A.kkt[3 * m] = kp; A.kkt[3 * m + 1] = K.e_du; A.kkt[3 * m + 2] = kc; }
  }
#undef KS_BEGIN
#undef KS_BEGIN_SYNCED
#undef KS_END
}

}  // namespace landing
