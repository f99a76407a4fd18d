// kd_solver_kernels.hip -- batched interior-point solver of the KINODYNAMIC REFINEMENT NLP (gfx950, fp64).
//
// The NLP: optimizations/landing/main_scripts/landing_optimization.m:38-201 (= generate_solver/generate_landingCtrller_KNITRO.m:45-215),
// the step after the SRBM solve in every production caller (:300-322 SRBM solution as the initial guess, :360-376 / :398-435 the solves,
// which the reference hands to KNITRO -- a commercial solver whose artefacts are absent from the tree).  Variables
//   x = [X (12 x (N+1)); jpos (12 x N); U (24 x N: c; f_grf)],  rows g in the script's order (rbd_kernels.hip, kd_stage_rows).
// What is solved here is the same primal-dual interior-point iteration as the SRBM solver's (solver_kernels.hip): slack + bound
// multiplier pair on every inequality row, fraction-to-the-boundary rule, filter line search, monotone barrier update, inertia correction
// by delta_w -- on this NLP's stage structure:
//   state   sigma_k = (X_k, c_k)            24      (X_0 and c_0 are fixed by the script's initial conditions, :89-91)
//   control u_k     = (f_k, jpos_k, c_k+1)  36      (24 in the last interval: there is no c_N)
//   dynamics X_k+1 = X_k + dt (...)  (the Euler defects :125-128 are linear in X_k+1 with unit coefficient), c_k+1 = part of u_k
// The joint angles are stage-local (they enter the FK band, the leg-torque rows and their own limits of interval k only), so they are
// eliminated with the forces inside the stage: the Riccati recursion runs on the 24-dimensional state.
//
// Division of work per interior-point iteration (host loop in kd_capi.inc, all members of the batch in lock step, finished members skip):
//   landing_kinodyn_nlp_jac_kernel / _hess_kernel   (rbd_kernels.hip, whole batch)   exact J blocks (141 x 72) and Hessian blocks of lam' g
//                                                    (72 x 72) of every interval by forward-mode AD, as CasADi provides them to the reference
//   landing_kd_head_kernel                          (this file, ONE WORKGROUP = ONE NLP)   error test, stop / restart / phase decisions, barrier update
//   landing_kd_condense_kernel                      (one workgroup per member and interval, round 6)   J_I' Sigma J_I and J_I' rho of the inequality rows on the
//       fp64 matrix cores (v_mfma_f64_16x16x4) -- independent of the Riccati recursion
//   landing_kd_iter_kernel                          (ONE WORKGROUP = ONE NLP)   Q_k = (H_k + delta_w I) + that part per stage, Riccati sweep with inertia correction,
//       forward sweep, step bounds, filter line search with g evaluated in the kernel (kd_stage_rows<double>, one lane per interval), acceptance.
// The iteration state of a member (mu, delta_w, filter, counters) lives in its workspace between launches.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/landing_nlp.h"

namespace landing {

constexpr int KD_NSIG = 24, KD_NV = 60;          // state; stage variables v = (sigma, f, jpos, c+) of a middle interval (48 in the last one)
constexpr int KD_MS = 62;                        // LDS row stride of the stage array [M | m] (column 60 = right-hand side)
constexpr int KD_PS = 25;                        // LDS row stride of the cost-to-go P (24 x 24)
constexpr int KD_AS = 37;                        // LDS row stride of [A^ | b] (12 x 37)
#ifndef KD_JC_ROWS_DEF
#define KD_JC_ROWS_DEF 44
#endif
constexpr int KD_JC_ROWS = KD_JC_ROWS_DEF, KD_JC_S = 65;     // rows per chunk of the staged Jacobian (the 129 inequality rows of an interval = 3 chunks of 44; round 5: 4 of 36), LDS row stride (64 columns + pad)
// per-interval record of the backward sweep (doubles): gains K (36 x 24) | kappa (36) | [A^ | b] (12 x 37) | state rows of the cost-to-go
// P_k (12 x 24) | p_k (12)
constexpr int KD_REC_K = 0, KD_REC_KAP = 864, KD_REC_AH = 900, KD_REC_PX = 1344, KD_REC_PV = 1632, KD_REC = 1648;
constexpr int KD_GC = KD_NV * KD_NV + KD_NV;      // per interval: the inequality rows' part of the stage array, J_I' Sigma J_I (row major, 60 x 60) | J_I' rho (60)
constexpr int KD_FILT = 48;
constexpr int KD_THREADS = 256;
// row r (0..11) of the Euler defects (v, omega, pos, rpy -- the script's order :125-128) <-> entry of X_k+1 it is linear in
__device__ __constant__ int KD_ROW2X[12] = {9, 10, 11, 6, 7, 8, 0, 1, 2, 3, 4, 5};

// out-of-line phases of the iteration kernel
#if defined(__HIP_DEVICE_COMPILE__)
#define KD_PHASE __device__ __noinline__
#else
#define KD_PHASE __device__ __noinline__
#endif
struct KdState {
  double mu, delta_last, th_max, c_pr, c_co, c_cm, c_ys, c_zs, c_nz, e_pr, e_du, e_co;
  double tau, a_pr, a_du, th0, ph0, dphi, alpha, s_corr, delta, ft, fval, omt;
  double filt_th[KD_FILT], filt_ph[KD_FILT];
  int nfilt, it, status, done, need_reg_streak, first_failed, cutstreak, force_step, wd_count, last_mu_it;
  int accepted, armijo_step, fact_ok, skipped_zero, attempt, flag, ls_done, need_corr, fallback, nfact, ntrial, nreset;
  int last_reset_it, ncrawl, clip_k_cur, fresh, reg_it;
  int pending;      // the inertia correction of this iteration continues in the next launch (landing_kd_iter_kernel, KD_TRIES_PER_ROUND)
  int stage;        // 1: landing_kd_head_kernel has prepared this iteration (error test passed, barrier parameter, delta_w of the first attempt): landing_kd_condense_kernel and
                    // landing_kd_iter_kernel act on it; 0: nothing to do for them this round (finished, restarted, or resumed through `pending`)
  int stag, full_prev; double e_prev;      // stag_relief (landing_nlp.h): full steps of the last barrier problem that did not halve the error
  // feasibility (restoration) phase, round 5 -- the scheme of landing_ipm_kernel (solver_kernels.hip, landing_nlp.h feas_phase / feas_jam / feas_stat):
  // feas = 1 while the elastic problem is being solved, lim = iteration limit in force, fjam / fstat / v1_ref = the two rules' counters
  int feas, feas_used, lim, fjam, fstat, polished; double v1_ref, c_rn, f_vmax, f_v1;
  // round 6 (landing_nlp.h feas_max / feas_back / feas_ret_push / feas_delta_dec / feas_resume): entries into the phase, stalled flag, hard iteration limit, "record the
  // entry violation" flag; violation at the entry (1-norm, equality rows included), 1-norm residual of the equality rows at x, adapted regularisation factor of the phase
  int n_feas, stalled, hard_lim, want_entry; double th_entry, f_theq, fdc;
  double prof[8]; long long tp;      // development aid: wall_clock64 ticks (100 MHz) per phase, summed over the iterations: grad | mu | backward | forward | dual | line search | accept
};

struct KdMem {
  double *x, *xt, *dx, *gx;
  double *g, *gt, *s, *ds, *zL, *zU, *y, *yn, *sig, *rho;
  double *J, *H, *rec, *wbuf;      // wbuf [N][72]: gathered stage variables of the in-kernel row evaluation
  double* jty;                     // [N][72]: J_k' y_k per interval and block column, written by the Jacobian kernel (rbd_kernels.hip, KdNlpArgs::jty)
  double *en, *ep, *wn, *wp;       // feasibility phase: violation variables of the lower / upper side of every inequality row and their multipliers
  double* gc;                      // [N][KD_GC]: J_I' Sigma J_I (60 x 60) and J_I' rho (60) of every interval, written by landing_kd_condense_kernel (round 6)
  KdState* st;
};
__host__ __device__ inline size_t kd_ws_stride(int N) {
  const size_t nx = (size_t)kd_nx(N), ng = (size_t)kd_ng(N);
  return 4 * nx + 10 * ng + (size_t)N * KD_JCS + (size_t)N * KD_NW * KD_NW + (size_t)(N + 1) * KD_REC + 2 * (size_t)N * KD_NW + 4 * ng + (size_t)N * KD_GC + (sizeof(KdState) + 7) / 8 + 8;
}
__device__ __forceinline__ KdMem kd_carve(int N, double* w) {
  const size_t nx = (size_t)kd_nx(N), ng = (size_t)kd_ng(N);
  KdMem M;
  M.x = w; w += nx; M.xt = w; w += nx; M.dx = w; w += nx; M.gx = w; w += nx;
  M.g = w; w += ng; M.gt = w; w += ng; M.s = w; w += ng; M.ds = w; w += ng; M.zL = w; w += ng; M.zU = w; w += ng;
  M.y = w; w += ng; M.yn = w; w += ng; M.sig = w; w += ng; M.rho = w; w += ng;
  M.J = w; w += (size_t)N * KD_JCS; M.H = w; w += (size_t)N * KD_NW * KD_NW; M.rec = w; w += (size_t)(N + 1) * KD_REC;      // (J: compact Jacobian blocks, rbd_kernels.hip KD_JCS)
  M.wbuf = w; w += (size_t)N * KD_NW;
  M.jty = w; w += (size_t)N * KD_NW;
  M.en = w; w += ng; M.ep = w; w += ng; M.wn = w; w += ng; M.wp = w; w += ng;
  M.gc = w; w += (size_t)N * KD_GC;
  M.st = reinterpret_cast<KdState*>(w);
  return M;
}

// Structural non-zeros of the inequality rows (12 ..) of an interval's Jacobian block, [0] = every interval but the last, [1] = the last one: the rows in the order of
// falling counts (perm: the lanes of a wavefront then run rows of the same length), their entries as ranges rp[j] .. rp[j + 1] of the column list cl.  Built once per
// context by asking the Jacobian kernel itself (kd_ensure_jpat, solver_capi.inc: an entry of a forward-mode derivative is exactly 0.0 where the row does not depend on the
// variable): 529 of the 129 x 72 entries of a middle interval, at most 9 in a row.
constexpr int KD_JP_ROWS = 132, KD_JP_NNZ = 640, KD_JP_MAX = 12;
struct KdJPat { unsigned short rp[2][KD_JP_ROWS]; unsigned char perm[2][KD_JP_ROWS]; unsigned char cl[2][KD_JP_NNZ]; };
static_assert(sizeof(KdJPat) % 8 == 0, "copied in 8-byte words");
// The condensation J_I' Sigma J_I, J_I' rho of an interval over those non-zeros (landing_kd_condense_kernel): entry t of the block's non-zero list sits in row position
// erow[t]; destination d = (va <= vb) of the 60 x 60 array (v = stage variable: block column w < 48, or w - 12 for w >= 60) sums the terms dterm[drp[d] .. drp[d + 1]),
// each (ta | tb << 10 | row position << 20): (J[ta] sigma_row) J[tb], rows ascending; the right-hand side of variable va sums rterm[rrp[va] .. rrp[va + 1]), each
// (ta | row position << 10): J[ta] rho_row.  3 305 products per middle interval instead of 129 x 60 x 60.  Built with KdJPat (kd_ensure_jpat).
constexpr int KD_CP_ND = 1024, KD_CP_NT = 3072;
struct KdCPat {
  int nd[2], nt[2];
  unsigned char erow[2][KD_JP_NNZ];
  unsigned short drp[2][KD_CP_ND + 1]; unsigned short dab[2][KD_CP_ND];      // dab = va | vb << 8
  unsigned int dterm[2][KD_CP_NT];
  unsigned short rrp[2][KD_NV + 4]; unsigned int rterm[2][KD_JP_NNZ];
  unsigned jcol[2][KD_NW][KD_JCOL];    // per column the stored rows of a block and their places in its compact form (KdNlpArgs::jcol)
};
static_assert(KD_JP_NNZ == KD_JC_NNZ, "one capacity");

struct KdSolveArgs {
  const RbdModel* model; KdNlpParams P; int B, N; landing_solver_opts o;
  const double* x0;        // [B][nx]
  const double* lb; const double* ub;      // [B][ng]  lbg / ubg (the script's values: kinodyn.py / landing_kinodyn_bounds)
  const double* cost;      // [B][24]  QN (12) | Xref(:, end) (12)   -- the terminal cost of :83-86
  double* ws; size_t ws_stride;
  double* x_out; double* f_out; double* lam_out; int* status; int* iters; double* kkt;
  int* n_active;           // number of members still iterating (written by the iteration kernel)
  const int* n_dcur; const int* dlist_cur;      // ... and the list of THIS round (the head kernel is gridded over it; pending members are on it too: the derivative kernels skip them by their flag)
  int* n_dnext; int* dlist_next;    // members that need their derivatives in the NEXT round of launches (count, list [B]): appended wherever a member ends a launch with a new point (round 6: the derivative
                                    // kernels are gridded over this list -- in the lock-step tail the launches over the whole batch were mostly workgroups that leave at once)
  const KdCPat* cpat;      // the condensation over those non-zeros (device copy)
  const KdJPat* jpat;      // structural non-zeros of the Jacobian blocks' inequality rows (device copy, built once per context)
  int* n_cond; int* cond_list;      // members whose head kernel has prepared an iteration this round (count, list [B]: the work list of landing_kd_condense_kernel)
  int* done;               // [B] 1 = the member has finished (read by the function-layer kernels: finished members are skipped)
  // Portfolio (round 5, landing_nlp.h kd_clone_after): members B0 .. B-1 are CLONE slots -- workspace blocks without a problem of their own.  After
  // kd_clone_after rounds the members still iterating are posed again in KD_NVAR clone slots each, from the callers' guess under another option set
  // (ov[v]); the first member of such a family that converges ends the others, the finish kernel reports it under the original's index.
  int B0, F, m_lo;         // clone slot of (wave w, variant v, family i) = B0 + ((w * KD_NVAR + v) * F + i); m_lo: first member of an init launch
  int* src;                // [B - B0] the original of a clone slot, -1 = unused
  const int* win_prev;     // [B0] the winners as they stood BEFORE this round of launches (copied by the host loop): what ends a member -- a relative that converges earlier in the SAME launch
                           // must not (with a plain read of `win` the family's result depended on the order in which the hardware ran the workgroups of one launch)
  int* win;                // [B0] the member of the family of original a that converged first (KD_NOWIN: none yet; of several in one round the lowest index: atomicMin)
  int* cloned;             // [B0] 1 = the original has clones
  landing_solver_opts ov[3];
};
#ifndef KD_NVAR_DEF
#define KD_NVAR_DEF 3
#endif
#ifndef KD_NWAVE_DEF
#define KD_NWAVE_DEF 4
#endif
constexpr int KD_NVAR = KD_NVAR_DEF, KD_NWAVE = KD_NWAVE_DEF;
constexpr int KD_NOWIN = 0x7f7f7f7f;      // (a byte pattern: the host sets it with one memset)
__device__ __forceinline__ int kd_problem_of(const KdSolveArgs& A, int m) { return m < A.B0 ? m : A.src[m - A.B0]; }
__device__ __forceinline__ const landing_solver_opts& kd_opts_of(const KdSolveArgs& A, int m) { return m < A.B0 ? A.o : A.ov[((m - A.B0) / A.F) % KD_NVAR]; }

// variable j of v = (X_k, c_k, f_k, jpos_k, c_k+1) -> column of the interval's block over w = (X_k, c_k, f_k, jpos_k, X_k+1, c_k+1)
__host__ __device__ inline int kd_v2w(int j) { return j < 48 ? j : j + 12; }

// LDS of one member's workgroup
struct KdLds {
  double Ms[KD_NV * KD_MS];            // stage array [M | m]
  double Pm[KD_NSIG * KD_PS];          // cost-to-go of the next stage
  double pv[KD_NSIG];
  double Ah[12 * KD_AS];               // [A^ | b]
  double Y[KD_NSIG * KD_AS];           // P [A^ | b] (+ p): rows of sigma+
  union {      // (the chunk buffer lives in the backward sweep, the knot steps in the forward sweep: one piece of LDS -- what pays for 44 instead of 36 rows per chunk)
    double Jc[KD_JC_ROWS * KD_JC_S];     // chunk of the interval's inequality rows (v columns, zero padded to 64)
    struct { double dsg[KD_NSIG * 65];   // d sigma_k of every knot (N <= 64)
             KdJPat jp; };               // the Jacobian blocks' non-zeros (forward sweep: ds = J_I dx)
  };
  double dxw[KD_NW];
  double red[(KD_THREADS / 64) * 6];
  int hist[64];                        // clip_k > 4: histogram of the blocking slacks over half-octaves of |ds| / distance
  int flag;
  KdState ks;
};
__shared__ KdLds KSH;

// g at x for one member: one lane per interval (callers pass every thread of the block, in uniform control flow; lanes >= N only write boundary rows / idle).
// The 72 stage variables of every interval are gathered by the WHOLE block into LDS (the arrays of the sweeps, free wherever g is evaluated) with all loads of a
// thread in flight, and read from there where the rows use them.  (Round 4-5: each interval's lane gathered its own 72 into the member's workspace, load after store
// after load -- 72 exposed round trips, most of the line search's 0.11 ms; as a local array they are promoted to 144 VGPRs and put the phase at 253 VGPRs + spills.)
KD_PHASE void kd_member_eval_g(const KdNlpParams& P, const RbdModel& M, int N, const double* x, double* g, double* wbuf) {
  (void)wbuf;
  static_assert(64 * KD_NW <= KD_NV * KD_MS + KD_NSIG * KD_PS + KD_NSIG + 12 * KD_AS + KD_NSIG * KD_AS, "the gathered variables of N <= 64 intervals fit the arrays in front of Jc");
  double* ws = KSH.Ms;
  const int tid = threadIdx.x, NT = blockDim.x, tot = N * KD_NW;
  constexpr int NB = 6;
  for (int e0 = 0; e0 < tot; e0 += NB * NT) {
    double t[NB]; int ix[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { const int e = e0 + tid + q * NT, ee = e < tot ? e : 0; ix[q] = kd_w_index(N, ee / KD_NW, ee % KD_NW); t[q] = x[ix[q] >= 0 ? ix[q] : 0]; }
#pragma unroll
    for (int q = 0; q < NB; ++q) { const int e = e0 + tid + q * NT; if (e < tot) ws[e] = ix[q] >= 0 ? t[q] : 0.0; }
  }
  __syncthreads();
  // Round 6: two passes.  (1) One lane per interval: every row but the legs' kinematics (legmask 0: zeros in those rows, c - fkv = 0).  (2) One lane per (interval, leg): the
  // leg's hip-relative foot position, its torques and its forward kinematics -- the expensive part, which one lane used to run for its four legs one after the other (the
  // line search's 0.07 ms per member-iteration was mostly that) -- into the seven rows of the leg and its six rows c - fkv.  Same functions on the same operands: the same bits.
  for (int k = tid; k < N; k += NT) {
    const double* w = ws + (size_t)k * KD_NW;
    // the rows go straight to the member's g array (a local out[141] is promoted to registers by the unrolled row code: 255 VGPRs + AGPR spills,
    // one workgroup per CU); the last interval writes its 117 rows only
    KdRowArray<double> rows{g + KD_BND + k * KD_ROWS};
    kd_stage_rows<double>(P, M, k, k == N - 1, w, rows, 0);
  }
  __syncthreads();
  for (int e = tid; e < 4 * N; e += NT) {
    const int k = e >> 2, l = e & 3;
    const bool last = k == N - 1;
    const double* w = ws + (size_t)k * KD_NW;
    double R[9], E0[9], r0[3], o7[7], fk[3];
    kd_base_frames_d(P, M, w, R, E0, r0);
    kd_leg_kin_d(M, l, w, R, E0, r0, o7, fk);
    double* gk = g + KD_BND + k * KD_ROWS;
    const int lb = last ? 9 : 15, o1 = 16 + l * lb + (last ? 2 : 8), o2 = 16 + 4 * lb + 17 + 3 * l;
    for (int a = 0; a < 7; ++a) gk[o1 + a] = o7[a];
    for (int j = 0; j < 3; ++j) { const double v = w[12 + 3 * l + j] - fk[j]; gk[o2 + j] = v; gk[o2 + 12 + j] = v; }
  }
  if (tid >= 64 && tid < 64 + 48) {      // boundary rows (coordinate picks), by a wave that has no interval to evaluate
    const int i = tid - 64, oU = 12 * (N + 1) + 12 * N;
    double v;
    if (i < 12) v = x[i];
    else if (i < 24) v = x[oU + (i - 12)];
    else { const int j = i - 24; v = j < 12 ? x[12 * N + (j % 6)] : x[12 * N + 6 + (j % 6)]; }
    g[i] = v;
  }
  __syncthreads();
}

#define KD_BEGIN() __syncthreads(); if (threadIdx.x == 0) {
#define KD_BEGIN_SYNCED() if (threadIdx.x == 0) {
#define KD_END() } __syncthreads()
#define KD_PROF(slot) do { if (threadIdx.x == 0) { const long long n_ = (long long)wall_clock64(); KSH.ks.prof[slot] += (double)(n_ - KSH.ks.tp); KSH.ks.tp = n_; } } while (0)

// ---- condensation of interval k -------------------------------------------------------------------------------------------------
// Round 6: split in two.  (a) kd_condense_rows -- the inequality rows' part J_I' Sigma J_I and J_I' rho of EVERY interval of every member that iterates this round,
// one workgroup per (member, interval) in landing_kd_condense_kernel between the head and the iteration kernel: it does not depend on the Riccati recursion, and
// inside the backward sweep its three chunks per stage (loads of 62 KB of J behind one chunk of matrix-core work) were 15 % of a batch of law main and 18 % of a round
// of the lock-step tail, where one member's chain is all there is (tools/dev/gpu_r06p.sh: builds that run the loop 1 / 2 / 3 times).  (b) kd_assemble_stage -- inside the
// sweep: M = (H + delta_w I) + [that part], m, [A^ | b], every load of a thread in flight together.  Same sums in the same order as the fused form of rounds 4-5:
// bit-identical iterates.
#ifndef KD_COND_UNROLL
#define KD_COND_UNROLL 44
#endif
#ifndef KD_COND_WGS
#define KD_COND_WGS 3      // (168 registers: at 4 the chunk loop spills -- 35 against 19 ms of kernel time per batch, tools/dev/gpu_r06t.sh)
#endif
#ifndef KD_COND_DENSE
#define KD_COND_DENSE 0      // 1: the matrix-core form (kd_condense_rows); 0: over the structural non-zeros (kd_condense_rows_sparse)
#endif
#ifndef KD_COND_GRID
#define KD_COND_GRID 2048
#endif
struct KdCondLds {
  double Jc[KD_JC_ROWS * KD_JC_S];     // chunk of the interval's inequality rows (v columns, zero padded to 64)
  double sgc[KD_JC_ROWS], rhc[KD_JC_ROWS];
};
__shared__ KdCondLds KCS;

// (a) J_I' Sigma J_I and m = J_I' rho over the inequality rows 12 .. nr-1 of interval k, in chunks of KD_JC_ROWS rows staged in LDS; the product runs on the fp64
// matrix cores: wave w owns row tile w of M (16 rows), four column tiles; the loads of the next chunk are in flight while the matrix cores work on this one
#if KD_COND_DENSE
#error "the matrix-core form of the condensation reads dense Jacobian blocks: the solver keeps them in compact form since round 6 (rbd_kernels.hip KD_JCS)"
#endif
__device__ __forceinline__ void kd_condense_rows(const KdMem& M, int N, int k) {
  KdCondLds& S = KCS;
  const int tid = threadIdx.x;
  const bool last = k == N - 1;
  const int nv = last ? 48 : KD_NV, nr = last ? KD_ROWS_LAST : KD_ROWS;
  // (global address space: global_load, not flat_load -- a flat load also counts on the LDS counter, so every wait for an LDS read would wait for the
  // prefetched rows as well: solver_kernels.hip landing_gptr)
  const landing_gptr Jk = (landing_gptr)(M.J + (size_t)k * KD_ROWS * KD_NW);
  const landing_gptr Gsig = (landing_gptr)M.sig, Grho = (landing_gptr)M.rho;
  const int g0 = KD_BND + k * KD_ROWS;
  const int wave = tid >> 6, l = tid & 63, lj = l & 15, lk = l >> 4;
  f64x4 acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
  double macc = 0.0;
  constexpr int NE = (KD_JC_ROWS * 64 + KD_THREADS - 1) / KD_THREADS;
  double pre[NE], pre_sg = 0.0, pre_rh = 0.0;
  auto fetch = [&](int r0) {
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      const int e = tid + q * KD_THREADS, rr = e >> 6, c = e & 63, r = r0 + rr;
      const bool in = rr < KD_JC_ROWS && r < nr && c < nv;
      const double v = Jk[(in ? r : 12) * KD_NW + kd_v2w(in ? c : 0)];      // unconditional load (clamped): all NE loads are issued together
      pre[q] = in ? v : 0.0;
    }
    { const int r = r0 + tid; const bool in = tid < KD_JC_ROWS && r < nr; const double a = Gsig[g0 + (in ? r : 12)], b = Grho[g0 + (in ? r : 12)]; pre_sg = in ? a : 0.0; pre_rh = in ? b : 0.0; }
  };
  fetch(12);
#ifdef KD_DEV_COND_REPS      // timing probe (tools/dev): the chunk loop KD_DEV_COND_REPS times, the last pass counts -- same results, the difference of two builds is the loop's cost
  for (int rep_ = 0; rep_ < KD_DEV_COND_REPS; ++rep_) {
  if (rep_ > 0) { for (int t = 0; t < 4; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0}; macc = 0.0; __syncthreads(); fetch(12); }
#endif
  for (int r0 = 12; r0 < nr; r0 += KD_JC_ROWS) {
#pragma unroll
    for (int q = 0; q < NE; ++q) { const int e = tid + q * KD_THREADS, rr = e >> 6, c = e & 63; if (rr < KD_JC_ROWS) S.Jc[rr * KD_JC_S + c] = pre[q]; }
    if (tid < KD_JC_ROWS) { S.sgc[tid] = pre_sg; S.rhc[tid] = pre_rh; }
    __syncthreads();
    if (r0 + KD_JC_ROWS < nr) fetch(r0 + KD_JC_ROWS);      // (uniform)
    for (int t = 0; t < 4; ++t)
      acc[t] = mfma_tile<KD_JC_ROWS / 4>(acc[t], [&](int i, int kk) { return S.Jc[kk * KD_JC_S + 16 * wave + i] * S.sgc[kk]; },
                                         [&](int kk, int j) { return S.Jc[kk * KD_JC_S + 16 * t + j]; });
#if defined(__HIP_DEVICE_COMPILE__)
    for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(acc[t]));
#endif
    if (tid < nv) {
#pragma unroll KD_COND_UNROLL
      for (int kk = 0; kk < KD_JC_ROWS; ++kk) macc += S.Jc[kk * KD_JC_S + tid] * S.rhc[kk];      // (one wave; unrolled: the 88 LDS reads are issued ahead of the chain of sums, whose order stays)
    }
    __syncthreads();
  }
#ifdef KD_DEV_COND_REPS
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(macc));
#endif
  }
#endif
  double* gk = M.gc + (size_t)k * KD_GC;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 4; ++r) {
      const int a = 16 * wave + lk + 4 * r, b = 16 * t + lj;
      if (a < nv && b < nv) gk[a * KD_NV + b] = acc[t][r];
    }
  if (tid < nv) gk[KD_NV * KD_NV + tid] = macc;
}

// (a') the same over the structural non-zeros of the block (KdJPat / KdCPat): the entries of the block into LDS (529 gathers), then one lane per destination of the
// 60 x 60 array (upper triangle, mirrored on the way out) and per right-hand side.  Entries of the array that no row couples are never written: the workspace starts
// cleared.  (The dense form above on the matrix cores: 0.51 ms per round of the full batch, this one: see DESIGN.md 4.8b.)
struct KdCondSparseLds { double val[KD_JP_NNZ]; double sg[KD_JP_ROWS], rh[KD_JP_ROWS]; };
__shared__ KdCondSparseLds KCP;
__device__ __forceinline__ void kd_condense_rows_sparse(const KdMem& M, int N, int k, const KdJPat* __restrict__ jp, const KdCPat* __restrict__ cp) {
  KdCondSparseLds& S = KCP;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int lp = k == N - 1 ? 1 : 0, nrow = (lp ? KD_ROWS_LAST : KD_ROWS) - 12, nnz = jp->rp[lp][nrow];
  const landing_gptr Jk = (landing_gptr)(M.J + (size_t)k * KD_JCS + KD_JC_DEF);      // the block's non-zeros in table order (compact form)
  const landing_gptr Gsig = (landing_gptr)M.sig, Grho = (landing_gptr)M.rho;
  const int g0 = KD_BND + k * KD_ROWS;
  for (int t = tid; t < nnz; t += NT) S.val[t] = Jk[t];
  for (int j = tid; j < nrow; j += NT) { const int r = jp->perm[lp][j]; S.sg[j] = Gsig[g0 + r]; S.rh[j] = Grho[g0 + r]; }
  __syncthreads();
  double* gk = M.gc + (size_t)k * KD_GC;
  const int nd = cp->nd[lp];
  for (int d = tid; d < nd; d += NT) {
    const int t0 = cp->drp[lp][d], t1 = cp->drp[lp][d + 1], ab = cp->dab[lp][d], va = ab & 255, vb = ab >> 8;
    double acc = 0.0;
    for (int t = t0; t < t1; ++t) { const unsigned w = cp->dterm[lp][t]; acc += (S.val[w & 1023u] * S.sg[w >> 20]) * S.val[(w >> 10) & 1023u]; }
    gk[va * KD_NV + vb] = acc; gk[vb * KD_NV + va] = acc;
  }
  if (tid < KD_NV) {
    double acc = 0.0;
    for (int t = cp->rrp[lp][tid]; t < cp->rrp[lp][tid + 1]; ++t) { const unsigned w = cp->rterm[lp][t]; acc += S.val[w & 1023u] * S.rh[w >> 10]; }
    gk[KD_NV * KD_NV + tid] = acc;
  }
}

// (b) stage array of interval k into KSH.Ms (nv x nv + rhs), KSH.Ah
KD_PHASE void kd_assemble_stage(const KdMem& M, int N, int k, double delta) {
  KdLds& S = KSH;
  const int tid = threadIdx.x;
  const bool last = k == N - 1;
  const int nv = last ? 48 : KD_NV;
  const landing_gptr Jk = (landing_gptr)(M.J + (size_t)k * KD_JCS);      // (the 12 defect rows are the dense head of the compact block)
  const landing_gptr Hk = (landing_gptr)(M.H + (size_t)k * KD_NW * KD_NW);
  const landing_gptr Ck = (landing_gptr)(M.gc + (size_t)k * KD_GC);
  const landing_gptr Gg = (landing_gptr)M.g;
  const int g0 = KD_BND + k * KD_ROWS;
  // Hessian block of lam' g over v with delta_w on the diagonal + the condensed inequality rows; right-hand side J_I' rho; [A^ | b] from the defect rows:
  // X_k+1 = A^ (sigma_k, f_k) + b in step form (rows in the order of X).  All loads of a thread are issued together, unconditionally (clamped addresses), and only then
  // stored: with the load under the bounds test each round waited for its own load (round 5)
  constexpr int NH = (KD_NV * KD_MS + KD_THREADS - 1) / KD_THREADS, NA = (12 * KD_AS + KD_THREADS - 1) / KD_THREADS;
  double hv[NH], cv[NH], av[NA];
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    const int e = tid + q * KD_THREADS, a = e / KD_MS, b = e % KD_MS;
    const bool in = a < nv && b < nv, rhs = a < nv && b == KD_NV;
    hv[q] = Hk[kd_v2w(in ? a : 0) * KD_NW + kd_v2w(in ? b : 0)];
    cv[q] = Ck[in ? a * KD_NV + b : (rhs ? KD_NV * KD_NV + a : 0)];
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int e = tid + q * KD_THREADS, ee = e < 12 * KD_AS ? e : 0, r = ee / KD_AS, c = ee % KD_AS;
    const double vj = Jk[r * KD_NW + (c < 36 ? c : 0)], vg = Gg[g0 + r];
    av[q] = c < 36 ? -vj : -vg;
  }
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    const int e = tid + q * KD_THREADS, a = e / KD_MS, b = e % KD_MS;
    const bool in = a < nv && b < nv, rhs = a < nv && b == KD_NV;
    if (e < KD_NV * KD_MS) S.Ms[e] = in ? (hv[q] + (a == b ? delta : 0.0)) + cv[q] : (rhs ? cv[q] : 0.0);
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int e = tid + q * KD_THREADS, ee = e < 12 * KD_AS ? e : 0, r = ee / KD_AS, c = ee % KD_AS;
    if (e < 12 * KD_AS) S.Ah[KD_ROW2X[r] * KD_AS + c] = av[q];
  }
  __syncthreads();
}

// One step of the blocked Gauss-Jordan elimination of the stage array (64 x 64 in accumulator tiles: T[rt][r] = element (row 16 rt + lk + 4 r,
// column 16 ct + lj), wave ct owns column tile ct) with the 4 x 4 pivot block at rows / columns [OFF, OFF + 4).  Exchange through LDS (buffer
// STEP & 1 in KSH.Jc, free during the elimination): every lane publishes the 4 pivot-row entries of its column, W[c][4]; the lanes that hold
// the pivot columns publish them, C[row][4]; one barrier; then every lane factors the pivot block D = L diag(d) L^T redundantly, solves
// D r = w for its own column and the wave applies T -= C R with one matrix-core instruction per row tile; the pivot rows become R itself.
// The scalar pivots d are those of the unblocked elimination (inertia test unchanged).  Same scheme as pivot_block_step of solver_kernels.hip;
// here the state columns 0..23 are never pivots, so only the tile of columns 32..47 may be skipped once it is eliminated.
template <int OFF, int STEP>
__device__ __forceinline__ bool kd_pivot_block_step(f64x4 (&T)[4], int ct, int lj, int lk, int c) {
  static_assert((OFF & 3) == 0 && OFF >= KD_NSIG && OFF + 4 <= KD_NV, "pivot blocks of the controls");
  KdLds& S = KSH;
  constexpr int RTB = OFF >> 4, R0 = (OFF & 15) >> 2;
  static_assert(2 * (64 * 4 + 64 * 4) <= KD_JC_ROWS * KD_JC_S, "exchange buffers fit the chunk buffer");
  double* W = S.Jc + (STEP & 1) * 512;
  double* C = W + 256;
  W[c * 4 + lk] = T[RTB][R0];
  if (ct == RTB && lj >= (OFF & 15) && lj < (OFF & 15) + 4) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(16 * rt + lk + 4 * r) * 4 + (lj - (OFF & 15))] = T[rt][r];
  }
  __syncthreads();
  double a[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) a[i][j] = C[(OFF + i) * 4 + j];
  double w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = W[c * 4 + i];
  double am[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) { const int row = 16 * rt + lj; const double cv = C[row * 4 + lk]; am[rt] = (row >= OFF && row < OFF + 4) ? 0.0 : cv; }   // pivot rows: no update
  auto recip = [](double d) { double i = __builtin_amdgcn_rcp(d); i = fma(i, fma(-d, i, 1.0), i); return fma(i, fma(-d, i, 1.0), i); };
  double l[4][4], t[4][4], inv[4];
  unsigned hm = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = j; i < 4; ++i) {
      double acc = a[i][j];
#pragma unroll
      for (int kk = 0; kk < j; ++kk) acc = fma(-l[i][kk], t[j][kk], acc);
      t[i][j] = acc;
    }
    inv[j] = recip(t[j][j]);
#pragma unroll
    for (int i = j + 1; i < 4; ++i) l[i][j] = t[i][j] * inv[j];
    const unsigned h = (unsigned)__double2hiint(t[j][j]) - 0x00100000u;      // pivot in (2^-1022, ~1e300): one unsigned range test on the high word
    hm = h > hm ? h : hm;
  }
  const bool ok = hm < (0x7e37e43cu - 0x00100000u);
  double y[4], r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double acc = w[i];
#pragma unroll
    for (int kk = 0; kk < i; ++kk) acc = fma(-l[i][kk], y[kk], acc);
    y[i] = acc;
  }
#pragma unroll
  for (int i = 3; i >= 0; --i) {
    double acc = y[i] * inv[i];
#pragma unroll
    for (int kk = i + 1; kk < 4; ++kk) acc = fma(-l[kk][i], r[kk], acc);
    r[i] = acc;
  }
  const double R = lk == 0 ? r[0] : (lk == 1 ? r[1] : (lk == 2 ? r[2] : r[3]));
  if (!(ct == 2 && OFF >= 48)) {      // (the tile of columns 32..47 holds eliminated controls only from then on)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) T[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[rt], -R, T[rt], 0, 0, 0);
    T[RTB][R0] = R;      // the pivot rows become the normalised rows, exactly
  }
  return ok;
}

// ---- one backward step: adds T' P+ T, T'(P+ t0 + p+) of the next stage's cost-to-go, eliminates the controls (Gauss-Jordan, scalar pivots),
// leaves gains / cost-to-go in the record and in KSH.Pm, KSH.pv.  ns_next = 24 (12 for the last interval: sigma_N = X_N).  false = a pivot
// was not positive (wrong inertia).
KD_PHASE bool kd_riccati_stage(const KdMem& M, int N, int k) {
  KdLds& S = KSH;
  const int tid = threadIdx.x, NT = blockDim.x;
  const bool last = k == N - 1;
  const int nv = last ? 48 : KD_NV, nu = nv - KD_NSIG;
#ifdef LANDING_KD_DEV
  const long long t_in_ = (long long)wall_clock64();
#endif
  // Gauss-Jordan on the control rows / columns 24 .. nv-1 of [M | m] on the fp64 matrix cores (round 5): the 64 x 64 array lives in
  // v_mfma_f64_16x16x4 accumulator tiles (wave w owns column tile w, four row tiles), pivot blocks of 4 x 4 -- kd_pivot_block_step, the
  // scheme of the SRBM solver's block_eliminate (solver_kernels.hip): 9 exchange + barrier rounds per stage instead of 36.  (Round 4: scalar
  // pivots with the array in registers, one barrier per pivot: 1.1 ms per factorisation.)
  bool ok = true;
  {
    const int ct = tid >> 6, l = tid & 63, lj = l & 15, lk = l >> 4, c = 16 * ct + lj;
    f64x4 T[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 16 * rt + lk + 4 * r; T[rt][r] = (row < KD_NV && c < KD_MS) ? S.Ms[row * KD_MS + c] : 0.0; }
    {   // + the cost-to-go of the next stage, sigma+ = (X+, c+) with X+ = A^ (sigma, f) + b:  E' P E and E' (P e + p), formed on the matrix cores where it is consumed (round 6,
        // the scheme of the SRBM solver's block_eliminate).  The own column of Y = P(:, X) [A^ | b] comes out of the matrix cores in accumulator layout, which IS the
        // B-operand layout of the next product; columns of c+ and the p-part of the right-hand side enter P directly.  Rounds 4-5 formed Y and A^' Y in scalar loops through
        // LDS with two barriers: 88 of the 349 us of a backward sweep per member-iteration under load (development timer).
      const bool isg = c == KD_NV, sf = c < 36, cpl = c >= 48 && c < KD_NV;
      const int cj = cpl ? 12 + (c - 48) : 0;
      double be[3], pa1[3], pa2[3], add1[3], add2[3], av[3][3];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const double v = S.Ah[(4 * kt + lk) * KD_AS + (sf ? c : 36)];
        be[kt] = (sf || isg) ? v : 0.0;
        pa1[kt] = S.Pm[lj * KD_PS + 4 * kt + lk];
        const double q = S.Pm[(12 + (lj < 12 ? lj : 0)) * KD_PS + 4 * kt + lk];
        pa2[kt] = lj < 12 ? q : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int row = 4 * r + lk;
        const double a1 = S.Pm[row * KD_PS + cj], a2 = S.Pm[(12 + row) * KD_PS + cj], g1 = S.pv[row], g2 = S.pv[12 + row];
        add1[r] = cpl ? a1 : (isg ? g1 : 0.0);
        add2[r] = cpl ? a2 : (isg ? g2 : 0.0);
      }
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) {
        const int a = 16 * rt + lj;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) { const double v = S.Ah[(4 * kt + lk) * KD_AS + (a < 36 ? a : 0)]; av[rt][kt] = a < 36 ? v : 0.0; }
      }
      f64x4 Y1 = {0.0, 0.0, 0.0, 0.0}, Y2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) { Y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1[kt], be[kt], Y1, 0, 0, 0); Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa2[kt], be[kt], Y2, 0, 0, 0); }
#pragma unroll
      for (int r = 0; r < 3; ++r) { Y1[r] += add1[r]; Y2[r] += add2[r]; }
#pragma unroll
      for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) T[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt][kt], Y1[kt], T[rt], 0, 0, 0);
      T[3][0] += Y2[0]; T[3][1] += Y2[1]; T[3][2] += Y2[2];      // rows of c+ (48..59): + rows 12..23 of P E
    }
#ifdef LANDING_KD_DEV      // development timer: the products with the cost-to-go of the next stage (prof[7], part of the backward sweep's slot)
    if (tid == 0) S.ks.prof[7] += (double)((long long)wall_clock64() - t_in_);
#endif
    ok &= kd_pivot_block_step<24, 0>(T, ct, lj, lk, c);
    ok &= kd_pivot_block_step<28, 1>(T, ct, lj, lk, c);
    ok &= kd_pivot_block_step<32, 2>(T, ct, lj, lk, c);
    ok &= kd_pivot_block_step<36, 3>(T, ct, lj, lk, c);
    ok &= kd_pivot_block_step<40, 4>(T, ct, lj, lk, c);
    ok &= kd_pivot_block_step<44, 5>(T, ct, lj, lk, c);
    if (!last) {      // (uniform)
      ok &= kd_pivot_block_step<48, 6>(T, ct, lj, lk, c);
      ok &= kd_pivot_block_step<52, 7>(T, ct, lj, lk, c);
      ok &= kd_pivot_block_step<56, 8>(T, ct, lj, lk, c);
    }
    if (!ok) { __syncthreads(); return false; }      // (identical in every lane)
    // gains and kappa to the interval's record, cost-to-go of this stage to LDS (+ its state rows to the record) -- straight from the accumulator tiles (round 6: rounds
    // 4-5 stored the array back to LDS behind a barrier and read it again)
    double* rec = M.rec + (size_t)k * KD_REC;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rt + lk + 4 * r;
        const double v = T[rt][r];
        if (row >= KD_NV) continue;
        if (row >= KD_NSIG) {      // control rows: [I | K | kappa]
          const int a = row - KD_NSIG;
          if (c < KD_NSIG) rec[KD_REC_K + a * 24 + c] = a < nu ? v : 0.0;
          else if (c == KD_NV) rec[KD_REC_KAP + a] = a < nu ? v : 0.0;
        } else {                   // state rows: the Schur complement [P_k | p_k]
          if (c < KD_NSIG) { S.Pm[row * KD_PS + c] = v; if (row < 12) rec[KD_REC_PX + row * 24 + c] = v; }
          else if (c == KD_NV) { S.pv[row] = v; if (row < 12) rec[KD_REC_PV + row] = v; }
        }
      }
    for (int e = tid; e < 12 * KD_AS; e += NT) rec[KD_REC_AH + e] = S.Ah[e];
  }
  __syncthreads();
  return true;
}

// terminal cost-to-go on X_N: 2 QN (terminal cost :83-86) + Sigma of the four terminal row groups (:94-97) + delta
__device__ __forceinline__ void kd_terminal(const KdMem& M, int N, const double* cost, double delta) {
  KdLds& S = KSH;
  const int tid = threadIdx.x, NT = blockDim.x;
  for (int e = tid; e < KD_NSIG * KD_PS; e += NT) S.Pm[e] = 0.0;
  if (tid < KD_NSIG) S.pv[tid] = 0.0;
  __syncthreads();
  if (tid < 12) {
    const int i = tid, ra = i < 6 ? 24 + i : 36 + (i - 6), rb = i < 6 ? 30 + i : 42 + (i - 6);
    const double qn2 = S.ks.feas ? 0.0 : 2.0 * cost[i];      // (the feasibility phase has no objective)
    const double pd = qn2 + M.sig[ra] + M.sig[rb] + delta, pg = qn2 * (M.x[12 * N + i] - cost[12 + i]) + M.rho[ra] + M.rho[rb];
    S.Pm[i * KD_PS + i] = pd; S.pv[i] = pg;
    double* rec = M.rec + (size_t)N * KD_REC;
    for (int j = 0; j < 24; ++j) rec[KD_REC_PX + i * 24 + j] = j == i ? pd : 0.0;
    rec[KD_REC_PV + i] = pg;
  }
  __syncthreads();
}

// whole backward sweep with regularisation delta
KD_PHASE bool kd_backward(const KdMem& M, int N, const double* cost, double delta) {
  kd_terminal(M, N, cost, delta);
  for (int k = N - 1; k >= 0; --k) {
    kd_assemble_stage(M, N, k, delta);
    if (!kd_riccati_stage(M, N, k)) return false;
  }
  return true;
}

// forward sweep: dx of every variable, multipliers of the defect rows (yn), ds of every inequality row
#ifndef KD_DS_U
#define KD_DS_U 16
#endif
KD_PHASE void kd_forward(const KdMem& M, int N, const double* lbm, const KdJPat* jpat) {
  KdLds& S = KSH;
  const int tid = threadIdx.x, NT = blockDim.x;
  const int oJ = 12 * (N + 1), oU = oJ + 12 * N;
  // sigma_0 is fixed by the initial conditions (rows 0..23: lb = ub = q_init, qd_init, c_init)
  if (tid < 24) { const int xi = tid < 12 ? tid : oU + (tid - 12); S.dsg[tid] = lbm[tid] - M.x[xi]; }
  __syncthreads();
  // the records (gains K 36 x 24, kappa, [A^ | b]: the first KD_REC_PX doubles) travel one stage ahead through registers into a double
  // buffer in the stage array (free here): round 4 read them from the workspace inside the serial chain, two exposed round trips per stage
  constexpr int NREC = KD_REC_PX, NQ = (NREC + KD_THREADS - 1) / KD_THREADS;
  static_assert(2 * NREC <= KD_NV * KD_MS, "two records fit the stage array");
  double nxt[NQ];
  auto fetch = [&](int k) {
    const landing_gptr rec = (landing_gptr)(M.rec + (size_t)(k < N ? k : N - 1) * KD_REC);      // (global_load: see kd_condense_rows)
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int e = tid + q * KD_THREADS; nxt[q] = rec[e < NREC ? e : NREC - 1]; }
  };
  auto stash = [&](int k) {
    double* b = S.Ms + (k & 1) * NREC;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int e = tid + q * KD_THREADS; if (e < NREC) b[e] = nxt[q]; }
  };
  fetch(0); stash(0); fetch(1);
  __syncthreads();
  for (int k = 0; k < N; ++k) {
    const bool last = k == N - 1;
    const int nu = last ? 24 : 36;
    const double* rec = S.Ms + (k & 1) * NREC;
    const double* sk = S.dsg + 24 * k;
    // du = -(K dsigma + kappa)
    if (tid < nu) {
      double a = rec[KD_REC_KAP + tid];
      for (int t = 0; t < 24; ++t) a += rec[KD_REC_K + tid * 24 + t] * sk[t];
      S.dxw[24 + tid] = -a;      // dxw[24..] = (df, djpos, dc+)
    }
    stash(k + 1); fetch(k + 2);      // (the other buffer: nobody reads it during this stage)
    __syncthreads();
    if (tid < 12) {              // dX+ = A^ (dsigma, df) + b
      double a = rec[KD_REC_AH + tid * KD_AS + 36];
      for (int t = 0; t < 24; ++t) a += rec[KD_REC_AH + tid * KD_AS + t] * sk[t];
      for (int t = 0; t < 12; ++t) a += rec[KD_REC_AH + tid * KD_AS + 24 + t] * S.dxw[24 + t];
      S.dsg[24 * (k + 1) + tid] = a;
    } else if (tid < 24) S.dsg[24 * (k + 1) + tid] = last ? 0.0 : S.dxw[24 + 24 + (tid - 12)];
    if (tid >= 64 && tid < 64 + 24) {          // steps of the stage variables into dx
      const int j = tid - 64;
      M.dx[j < 12 ? 12 * k + j : oU + 24 * k + (j - 12)] = sk[j];
    }
    if (tid >= 128 && tid < 128 + 24) {
      const int j = tid - 128;
      M.dx[j < 12 ? oU + 24 * k + 12 + j : oJ + 12 * k + (j - 12)] = S.dxw[24 + j];
    }
    __syncthreads();
  }
  if (tid < 12) M.dx[12 * N + tid] = S.dsg[24 * N + tid];
  // multipliers of the defect rows of interval k: y = -(P_k+1 dsigma_k+1 + p_k+1)_X in the row order of the defects
  for (int e = tid; e < 12 * N; e += NT) {
    const int k = e / 12, r = e % 12, i = KD_ROW2X[r];
    const landing_gptr recn = (landing_gptr)(M.rec + (size_t)(k + 1) * KD_REC);
    const double* sn = S.dsg + 24 * (k + 1);
    double a = recn[KD_REC_PV + i];
    for (int t = 0; t < 24; ++t) a += recn[KD_REC_PX + i * 24 + t] * sn[t];
    M.yn[KD_BND + k * KD_ROWS + r] = -a;
  }
  __syncthreads();
  // ds = J_I dx + (g - s).  The steps of all intervals' block variables are gathered first (into the stage array, free here), so that no barrier separates the intervals.
  // Round 6: over the STRUCTURAL non-zeros of the rows (KdJPat: 4.1 entries per row on average, at most 9, of 72): one lane per row, its entries gathered from the dense
  // block with all loads of the row in flight, summed in column order -- no cross-lane reduction.  Rounds 4-5 streamed the dense rows (one wave per row, lanes over the 72
  // columns, a 6-level shuffle reduction per row): 9.5 % of a batch of law main, 0.18 ms per round of the lock-step tail (tools/dev/gpu_r06w.sh).
  static_assert(64 * KD_NW <= KD_NV * KD_MS + KD_NSIG * KD_PS + KD_NSIG + 12 * KD_AS + KD_NSIG * KD_AS, "the gathered steps of N <= 64 intervals fit the arrays in front of Jc");
  double* dxa = S.Ms;      // [N][72]  (runs on into Pm, pv, Ah, Y for long horizons: all free during the forward sweep)
  for (int e = tid; e < N * KD_NW; e += NT) { const int k = e / KD_NW, i = kd_w_index(N, k, e % KD_NW); dxa[e] = i >= 0 ? M.dx[i] : 0.0; }
  { const unsigned long long* src = reinterpret_cast<const unsigned long long*>(jpat); unsigned long long* dst = reinterpret_cast<unsigned long long*>(&S.jp);
    for (int e = tid; e < (int)(sizeof(KdJPat) / 8); e += NT) dst[e] = src[e]; }
  __syncthreads();
#ifdef KD_DEV_DS_REPS      // timing probe (tools/dev): the pass KD_DEV_DS_REPS times, same results
  for (int rep_ = 0; rep_ < KD_DEV_DS_REPS; ++rep_)
#endif
  {
    constexpr int RM = KD_ROWS - 12, RL = KD_ROWS_LAST - 12;      // inequality rows of a middle / of the last interval
    const int nmid = (N - 1) * RM, ntot = nmid + RL;
    const landing_gptr Gg = (landing_gptr)M.g, Gs = (landing_gptr)M.s;
    for (int i0 = 0; i0 < ntot; i0 += NT) {
      const int idx = i0 + tid; const bool on = idx < ntot; const int ii = on ? idx : 0;
      const int lp = ii >= nmid ? 1 : 0, k = lp ? N - 1 : ii / RM, j = lp ? ii - nmid : ii % RM;
      const int r = S.jp.perm[lp][j], t0 = S.jp.rp[lp][j], cnt = S.jp.rp[lp][j + 1] - t0;
      const landing_gptr Jr = (landing_gptr)(M.J + (size_t)k * KD_JCS + KD_JC_DEF + t0);      // the row's entries, contiguous in the compact block (global_load: the LDS reads below wait on the LDS counter only)
      const double* dk = dxa + k * KD_NW;
      const int g = KD_BND + k * KD_ROWS + r;
      double v[KD_JP_MAX]; int c[KD_JP_MAX];
#pragma unroll
      for (int u = 0; u < KD_JP_MAX; ++u) { const int uu = u < cnt ? u : 0; c[u] = S.jp.cl[lp][t0 + uu]; v[u] = Jr[uu]; }      // (clamped: every load is issued, unconditionally)
      const double gs = Gg[g] - Gs[g];
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < KD_JP_MAX; ++u) acc += u < cnt ? v[u] * dk[c[u]] : 0.0;
      if (on) M.ds[g] = acc + gs;
    }
  }
  __syncthreads();
  if (tid < 24) {      // terminal rows: copies of X_N
    const int r = 24 + tid, xi = tid < 12 ? (tid % 6) : 6 + (tid % 6);
    M.ds[r] = M.dx[12 * N + xi] + (M.g[r] - M.s[r]);
  }
  __syncthreads();
}

// gx = grad f + J' y over the free rows (rows 24 .. ng-1): one thread per variable, the (at most two) block columns that hold it.  The column
// products J_k' y_k are a by-product of the Jacobian kernel (M.jty; round 4 re-read every J block here: 1.5 MB and 0.49 ms per member-iteration)
KD_PHASE void kd_grad_lag(const KdMem& M, int N, const double* cost, double obj) {      // obj: 1, or 0 in the feasibility phase
  const int tid = threadIdx.x, NT = blockDim.x, nx = kd_nx(N);
  const int oJ = 12 * (N + 1), oU = oJ + 12 * N;
  for (int i = tid; i < nx; i += NT) {
    int k, j0, j1 = -1;      // own interval and column, column in the previous interval's block
    if (i < oJ) { k = i / 12; j0 = i % 12; j1 = 48 + j0; }
    else if (i < oU) { k = (i - oJ) / 12; j0 = 36 + (i - oJ) % 12; }
    else { k = (i - oU) / 24; const int q = (i - oU) % 24; j0 = 12 + q; if (q < 12) j1 = 60 + q; }
    double a = 0.0;
    if (k < N) a += M.jty[(size_t)k * KD_NW + j0];
    if (j1 >= 0 && k >= 1) a += M.jty[(size_t)(k - 1) * KD_NW + j1];
    if (i >= 12 * N && i < 12 * N + 12) {      // X_N: terminal cost and the four terminal row groups
      const int q = i - 12 * N;
      a += obj * 2.0 * cost[q] * (M.x[i] - cost[12 + q]);
      a += q < 6 ? M.y[24 + q] + M.y[30 + q] : M.y[36 + (q - 6)] + M.y[42 + (q - 6)];
    }
    M.gx[i] = a;
  }
  __syncthreads();
}

// primal / complementarity errors, Sigma and rho of the current point for barrier parameter mu_ -> K.c_*, M.sig, M.rho
KD_PHASE void kd_point_pass(const KdMem& M, int ng, const double* lbm, const double* ubm, double mu_) {
  KdLds& S = KSH;
  const int tid = threadIdx.x, NT = blockDim.x;
  const double INF = INFINITY;
  double pr = 0.0, co = 0.0, cm = 0.0, ys = 0.0, zs = 0.0, nz = 0.0;
  for (int r = tid; r < ng; r += NT) {
    const double lb = lbm[r], ub = ubm[r];
    double sg = 0.0, rh = 0.0;
    if (r >= 24) {
      const double g = M.g[r];
      ys += fabs(M.y[r]);
      if (lb == ub) pr = fmax(pr, fabs(g - lb));
      else {
        const double s = M.s[r];
        pr = fmax(pr, fabs(g - s));
        if (lb > -INF) { const double d = s - lb, zl = M.zL[r]; co = fmax(co, d * zl); cm = fmax(cm, fabs(d * zl - mu_)); sg += zl / d; rh -= mu_ / d; zs += zl; nz += 1.0; }
        if (ub < INF) { const double d = ub - s, zu = M.zU[r]; co = fmax(co, d * zu); cm = fmax(cm, fabs(d * zu - mu_)); sg += zu / d; rh += mu_ / d; zs += zu; nz += 1.0; }
        rh += sg * (g - s);
      }
    }
    M.sig[r] = sg; M.rho[r] = rh;
  }
  double v[6] = {pr, co, cm, ys, zs, nz}; const int op[6] = {RMAX, RMAX, RMAX, RSUM, RSUM, RSUM};
  block_reduce<6>(v, op, S.red);
  KD_BEGIN_SYNCED() S.ks.c_pr = v[0]; S.ks.c_co = v[1]; S.ks.c_cm = v[2]; S.ks.c_ys = v[3]; S.ks.c_zs = v[4]; S.ks.c_nz = fmax(v[5], 1.0); KD_END();
}

// the same for the elastic problem of the feasibility phase (solver_kernels.hip, el_step): every inequality row lb <= s <= ub becomes
// a = s - lb + n >= 0, n >= 0 (b = ub + q - s >= 0, q >= 0) at the price rho_pen (n + q); the row enters the condensed system with sigma = z / D.
// Also leaves the violation of the inequality rows at x (max norm, 1-norm) and |z + w - rho_pen|_inf in K
KD_PHASE void kd_feas_point_pass(const KdMem& M, int ng, const double* lbm, const double* ubm, double mu_, double frho) {
  KdLds& S = KSH;
  const int tid = threadIdx.x, NT = blockDim.x;
  const double INF = INFINITY;
  double pr = 0.0, co = 0.0, cm = 0.0, rn = 0.0, ys = 0.0, zs = 0.0, nz = 0.0, vmax = 0.0, v1 = 0.0, teq = 0.0;
  for (int r = tid; r < ng; r += NT) {
    const double lb = lbm[r], ub = ubm[r];
    double sg = 0.0, rh = 0.0;
    if (r >= 24) {
      const double g = M.g[r];
      ys += fabs(M.y[r]);
      if (lb == ub) { pr = fmax(pr, fabs(g - lb)); teq += fabs(g - lb); }
      else {
        const double s = M.s[r], v = fmax(fmax(lb - g, g - ub), 0.0);
        pr = fmax(pr, fabs(g - s)); vmax = fmax(vmax, v); v1 += v;
        if (lb > -INF) {
          const double n = M.en[r], a = s - lb + n, z = M.zL[r], w = M.wn[r], D = a + z * n / w;
          const double c = (mu_ - a * z - z * (mu_ - n * w + n * (z + w - frho)) / w) / D;
          co = fmax(co, fmax(a * z, n * w)); cm = fmax(cm, fmax(fabs(a * z - mu_), fabs(n * w - mu_))); rn = fmax(rn, fabs(z + w - frho));
          sg += z / D; rh -= z + c; zs += z; nz += 1.0;
        }
        if (ub < INF) {
          const double q = M.ep[r], b = ub + q - s, z = M.zU[r], w = M.wp[r], D = b + z * q / w;
          const double c = (mu_ - b * z - z * (mu_ - q * w + q * (z + w - frho)) / w) / D;
          co = fmax(co, fmax(b * z, q * w)); cm = fmax(cm, fmax(fabs(b * z - mu_), fabs(q * w - mu_))); rn = fmax(rn, fabs(z + w - frho));
          sg += z / D; rh += z + c; zs += z; nz += 1.0;
        }
        rh += sg * (g - s);
      }
    }
    M.sig[r] = sg; M.rho[r] = rh;
  }
  double v[6] = {pr, co, cm, ys, zs, nz}; const int op[6] = {RMAX, RMAX, RMAX, RSUM, RSUM, RSUM};
  block_reduce<6>(v, op, S.red);
  double u[4] = {rn, vmax, v1, teq}; const int op4[4] = {RMAX, RMAX, RSUM, RSUM};
  block_reduce<4>(u, op4, S.red);
  KD_BEGIN_SYNCED()
    S.ks.c_pr = v[0]; S.ks.c_co = v[1]; S.ks.c_cm = v[2]; S.ks.c_ys = v[3]; S.ks.c_zs = v[4]; S.ks.c_nz = fmax(v[5], 1.0); S.ks.c_rn = u[0]; S.ks.f_vmax = u[1]; S.ks.f_v1 = u[2]; S.ks.f_theq = u[3];
    if (S.ks.want_entry) { S.ks.th_entry = u[2] + u[3]; S.ks.want_entry = 0; }      // first pass of a phase: the violation it starts from
  KD_END();
}

KD_PHASE void kd_init_slacks(const KdMem& M, int ng, const double* lbm, const double* ubm, const landing_solver_opts& o) {
  const double INF = INFINITY;
  for (int r = threadIdx.x; r < ng; r += blockDim.x) {
    const double lb = lbm[r], ub = ubm[r];
    double sv = 0.0, zl = 0.0, zu = 0.0;
    if (r >= 24 && lb != ub) {
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
}

// ... at the point a feasibility phase hands back (landing_nlp.h feas_ret_push, round 6): slacks pushed only feas_ret_push off their bounds, multipliers mu / distance
KD_PHASE void kd_init_slacks_return(const KdMem& M, int ng, const double* lbm, const double* ubm, const landing_solver_opts& o, double mu_) {
  const double INF = INFINITY, push = o.feas_ret_push;
  for (int r = threadIdx.x; r < ng; r += blockDim.x) {
    const double lb = lbm[r], ub = ubm[r];
    double sv = 0.0, zl = 0.0, zu = 0.0;
    if (r >= 24 && lb != ub) {
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
}

// ---- start of a solve: initial point, slacks, multipliers, iteration state ------------------------------------------------------
__global__ void __launch_bounds__(KD_THREADS) landing_kd_init_kernel(KdSolveArgs A) {
  const int m = blockIdx.x + A.m_lo;
  if (m >= A.B) return;
  const int pm = kd_problem_of(A, m);      // (clone slots: the original's problem data, the variant's options)
  if (pm < 0) return;
  const landing_solver_opts& o = kd_opts_of(A, m);
  const int N = A.N, nx = kd_nx(N), ng = kd_ng(N), tid = threadIdx.x, NT = blockDim.x;
  const KdMem M = kd_carve(N, A.ws + (size_t)m * A.ws_stride);
  const double* lbm = A.lb + (size_t)pm * ng; const double* ubm = A.ub + (size_t)pm * ng;
  const int oU = 12 * (N + 1) + 12 * N;
  for (int i = tid; i < nx; i += NT) {
    double v = A.x0[(size_t)pm * nx + i];
    if (i < 12) v = lbm[i];                                  // X(:,1) and c(:,1) are fixed (:89-91)
    else if (i >= oU && i < oU + 12) v = lbm[12 + (i - oU)];
    M.x[i] = v;
  }
  __syncthreads();
  kd_member_eval_g(A.P, *A.model, N, M.x, M.g, M.wbuf);
  __syncthreads();
  kd_init_slacks(M, ng, lbm, ubm, o);
  KdState& K = KSH.ks;
  if (tid == 0) {
    K.mu = o.mu_init; K.delta_last = 0.0; K.th_max = 0.0; K.c_pr = K.c_co = K.c_cm = K.c_ys = K.c_zs = 0.0; K.c_nz = 1.0;
    K.e_pr = K.e_du = K.e_co = 0.0; K.tau = 0.0; K.a_pr = K.a_du = 0.0; K.th0 = K.ph0 = K.dphi = K.alpha = K.s_corr = K.delta = K.ft = K.fval = 0.0; K.omt = -1.0;
    K.nfilt = 0; K.it = 0; K.status = LANDING_MAX_ITER; K.done = 0; K.need_reg_streak = 0; K.first_failed = 0; K.cutstreak = 0; K.force_step = 0;
    K.wd_count = 0; K.last_mu_it = 0; K.accepted = 0; K.armijo_step = 0; K.fact_ok = 0; K.skipped_zero = 0; K.attempt = 0; K.flag = 0; K.ls_done = 0;
    K.need_corr = 0; K.fallback = 0; K.nfact = 0; K.ntrial = 0; K.nreset = 0; K.last_reset_it = 0; K.ncrawl = 0; K.clip_k_cur = o.clip_k; K.fresh = 0; K.reg_it = -1000; K.pending = 0; K.stage = 0; K.stag = 0; K.full_prev = 0; K.e_prev = 1e300;
    K.feas = 0; K.feas_used = 0; K.lim = o.max_iter; K.fjam = 0; K.fstat = 0; K.polished = 0; K.v1_ref = 0.0; K.c_rn = 0.0; K.f_vmax = 0.0; K.f_v1 = 0.0;
    K.n_feas = 0; K.stalled = 0; K.want_entry = 0; K.th_entry = 0.0; K.f_theq = 0.0; K.hard_lim = o.max_iter > 0 ? 3 * o.max_iter : 0; K.fdc = o.feas_delta_dec > 0.0 ? o.feas_delta_dec : o.delta_dec;
    for (int i = 0; i < 8; ++i) K.prof[i] = 0.0; K.tp = 0;
  }
  __syncthreads();
  kd_point_pass(M, ng, lbm, ubm, K.mu);
  // Presolve: rows of the first interval that depend on the FIXED variables X(:,1), c(:,1) only -- foot height, kinematic box, leg length of
  // the initial stance (:138, :157-164) and the z bound (:181).  The script fixes c(:,1) to the nominal stance under the hips of the initial
  // attitude (:232-236) and bounds p_rel in WORLD axes, so a steep initial pitch / roll with a small velocity-dependent box violates them
  // whatever the solver does (190 of 1024 drop states of the callers' sampling law): such a member is reported as LANDING_INFEASIBLE at
  // once, kkt[0] = the violation, instead of driving the interior-point iteration into a numerical failure.
  double viol = 0.0;
  if (tid < 21 && N >= 2) {
    const int l = tid / 5, j = tid % 5;
    const int r = tid == 20 ? KD_BND + 92 : KD_BND + 16 + 15 * l + (j == 0 ? 0 : 7 + j);      // per leg (15 rows): c_z (0) | p_rel x, y, z (8..10) | |p_rel|^2 (11); row 92 = z
    const double g = M.g[r];
    viol = fmax(fmax(lbm[r] - g, g - ubm[r]), 0.0);
  }
  viol = block_reduce1(viol, RMAX, KSH.red);
  if (tid == 0) {
    if (viol > o.tol) { K.status = LANDING_INFEASIBLE; K.done = 1; K.e_pr = viol; }
    *M.st = K;
    A.done[m] = K.done;
    if (!K.done) A.dlist_next[atomicAdd(A.n_dnext, 1)] = m;      // (the host passes the list the NEXT derivative launches read)
  }
}

// ---- one interior-point iteration of one member (J and H blocks of the current (x, y) are in the workspace) ---------------------------
#ifndef KD_DELTA_JUMP
#define KD_DELTA_JUMP 12     // (round 5; 0 = off: rounds 3-4)
#endif
#ifndef KD_TRIES_PER_ROUND
#define KD_TRIES_PER_ROUND 2      // (round 5: 3 with the matrix-core elimination, 0.7 instead of 1.8 ms per attempt -- 1.23 -> 1.07 s per batch of 1024; round 4: 1.  With the
                                  // portfolio and the delta_w continuation 2: the tail rounds wait for their slowest member's attempts -- 0.408 -> 0.394 s, 1 attempt 0.411)
#endif
// Round 6: three launches per round.  landing_kd_head_kernel -- error test, stop / restart / phase decisions, barrier parameter, delta_w of the first attempt (one
// workgroup per member); landing_kd_condense_kernel -- J_I' Sigma J_I and J_I' rho of every interval (one workgroup per member and interval: sigma and rho are final
// once the head has set the barrier parameter); landing_kd_iter_kernel -- Riccati sweeps with inertia correction, forward sweep, line search, acceptance.  The state
// travels in the member's workspace (KdState::stage says whether the head has prepared an iteration).
__global__ void __launch_bounds__(KD_THREADS, 2) landing_kd_head_kernel(KdSolveArgs A) {
  if ((int)blockIdx.x >= *A.n_dcur) return;
  const int m = A.dlist_cur[blockIdx.x];
  if (m >= A.B) return;
  const int N = A.N, nx = kd_nx(N), ng = kd_ng(N), tid = threadIdx.x, NT = blockDim.x;
  const KdMem M = kd_carve(N, A.ws + (size_t)m * A.ws_stride);
  const int pm = kd_problem_of(A, m);
  if (pm < 0) return;                                       // unused clone slot
  if (M.st->done) return;                                   // (uniform: one global word per member)
  if (A.win) {      // portfolio: a relative of this member has converged -- the family's result is there
    const int w = A.win_prev[pm];
    if (w != KD_NOWIN && w != m) { if (threadIdx.x == 0) { M.st->done = 1; A.done[m] = 1; } return; }
  }
  const landing_solver_opts& o = kd_opts_of(A, m);
  const double* lbm = A.lb + (size_t)pm * ng; const double* ubm = A.ub + (size_t)pm * ng;
  const double* cost = A.cost + (size_t)pm * 24;
  const double INF = INFINITY;
  KdLds& S = KSH;
  KdState& K = S.ks;
  if (tid == 0) { K = *M.st; K.tp = (long long)wall_clock64(); }
  __syncthreads();
  const int oU = 12 * (N + 1) + 12 * N;
  // The host loop runs the members in lock step (derivative kernels between the iterations), so a launch lasts as long as its slowest member: one
  // that needs five regularisation attempts (1.8 ms each) used to hold all others back -- at full batch the launch took 15 ms for 2 x 3.3 ms of work
  // per slot.  After KD_TRIES_PER_ROUND failed factorisations the member therefore saves its state and RETURNS (pending): the next launch resumes
  // its inertia correction where it stopped, the derivative kernels skip it meanwhile (A.done[m] = 2: x has not moved).
  if (K.pending != 0) {            // (uniform) its inertia correction continues in landing_kd_iter_kernel; nothing of the member has moved (the condensation repeats itself: same J, sigma, rho)
    if (tid == 0) A.cond_list[atomicAdd(A.n_cond, 1)] = m;
    return;
  }
  {
    if (tid == 0) K.stage = 0;
    // ---------------------------------------------------------------- optimality error (unscaled), stop test
    kd_grad_lag(M, N, cost, K.feas ? 0.0 : 1.0);
    KD_PROF(0);
    {
      double du = 0.0;
      for (int i = tid; i < nx; i += NT) { const bool fixed = i < 12 || (i >= oU && i < oU + 12); if (!fixed) du = fmax(du, fabs(M.gx[i])); }
      du = block_reduce1(du, RMAX, S.red);
      KD_BEGIN_SYNCED()
        const double pr = K.c_pr, co = K.c_co;
        if (K.feas) du = fmax(du, K.c_rn);      // the elastic problem has the extra stationarity rows rho_pen - z - w = 0
        K.e_pr = pr; K.e_du = du; K.e_co = co;
        if (o.stag_relief > 0) {      // as in landing_ipm_kernel: the proximal term turns the last Newton steps into a linear iteration (one member of the
          const double E = fmax(pr, du);      // bench batch: 400 full steps from pr 1.5e-5 to 1e-6, a quarter of the batch's wall time)
          K.stag = (!K.feas && K.mu <= o.tol / 10.0 * 1.0000001 && K.full_prev && E > 0.5 * K.e_prev) ? K.stag + 1 : 0;
          K.e_prev = E;
        }
        K.flag = 0;      // 0: iterate, 1: stop, 2: restart, 3: back from the feasibility phase, 4: into the feasibility phase
        bool give_up = false;
        if (K.feas) {
          // feasibility phase (landing_nlp.h): a feasible point (or an elastic KKT point with negligible violation) restarts the solve from here, an elastic
          // KKT point with positive violation is the certificate of local infeasibility; a violation that has been stationary for feas_stat iterations with
          // the equality rows at 1e-3 is not (LANDING_STALLED, round 6)
          const bool conv = fmax(du, fmax(pr, co)) <= o.tol;
          if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { K.status = LANDING_NUMERICAL; K.flag = 1; }
          else if (K.f_vmax <= 1e-9 && pr <= o.tol) K.flag = 3;
          else if (conv && K.f_v1 > o.feas_cert) { K.status = LANDING_INFEASIBLE; K.flag = 1; }
          else if (conv) K.flag = 3;
          else if (o.feas_back > 0.0 && !K.feas_used && K.f_v1 + K.f_theq <= o.feas_back * K.th_entry) K.flag = 3;      // the violation has come down: back to the interior-point iteration (IPOPT's restoration phase)
          else if (o.feas_stat > 0) {
            const double v1 = K.f_v1;
            if (K.fstat < 0 || !(fabs(v1 - K.v1_ref) <= 0.05 * K.v1_ref)) { K.v1_ref = v1; K.fstat = 0; } else K.fstat++;
            if (K.fstat >= o.feas_stat && K.mu <= 1e-4 && pr <= 1e-3) {      // round 6 (landing_nlp.h): a stationary violation is not a certificate -- the regularisation is dropped once
              if (v1 <= o.feas_cert) K.flag = 3;                             // (a stationary point is then a few Newton steps from the elastic KKT point, status 3 above); after that LANDING_STALLED
              else if (o.feas_polish > 0.0 && !K.polished) { K.polished = 1; K.fstat = -1; K.delta_last = o.feas_polish / (o.feas_delta_dec > 0.0 ? o.feas_delta_dec : o.delta_dec); K.need_reg_streak = 2; }
              else if (o.feas_resume && !K.stalled) { K.stalled = 1; K.feas_used = 1; K.flag = 3; }      // the interior-point iteration resumes from this point, once
              else { K.status = LANDING_STALLED; K.flag = 1; }
            }
          }
          if (K.flag == 0 && K.it >= K.lim) { K.status = K.stalled ? LANDING_STALLED : LANDING_MAX_ITER; K.flag = 1; }
          if (K.flag == 3) {
            K.feas = 0; K.lim = K.it + (o.max_iter > 1 ? o.max_iter : 1); if (K.lim > K.hard_lim) K.lim = K.hard_lim; K.status = LANDING_MAX_ITER;
            K.mu = (o.feas_ret_push > 0.0 && o.feas_ret_mu > 0.0) ? o.feas_ret_mu : o.mu_init; K.fjam = 0; K.nfilt = 0; K.delta_last = 0.0; K.need_reg_streak = 0; K.wd_count = 0; K.th_max = 0.0; K.nreset = 0; K.last_reset_it = K.it; K.ncrawl = 0;
            K.cutstreak = 0; K.force_step = 0;
            K.it = K.it + 1;
          }
        }
        else if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { K.status = LANDING_NUMERICAL; give_up = true; }
        else if (fmax(du, fmax(pr, co)) <= o.tol) { K.status = LANDING_CONVERGED; K.flag = 1; }
        else if (K.it >= K.lim) { K.status = LANDING_MAX_ITER; give_up = true; }
        else if (du > o.reset_du && K.nreset >= o.max_resets && o.max_resets > 0) { K.status = LANDING_NUMERICAL; give_up = true; }      // jammed again: give up
        else if (o.feas_jam > 0 && K.fjam >= o.feas_jam && pr > 1e-3 && o.feas_phase) { K.status = LANDING_MAX_ITER; give_up = true; if (K.feas_used) K.stalled = 1; }      // jammed line search (landing_nlp.h); no entry left: status 4
        else {
          // restart rules of the SRBM solver (solver_kernels.hip, landing_nlp.h fresh_restart): a jammed iterate (multipliers blown up), a first
          // barrier problem that crawls, a later one that has wandered off -> slacks, multipliers, barrier parameter and filter are re-initialised,
          // at the current x or (after a jam) at the caller's initial guess with the step rule clip_k = 2
          const int it = K.it, nreset = K.nreset; const double mu = K.mu;
          const bool jam = du > o.reset_du && nreset < o.max_resets;
          const bool stalled = o.restart_period > 0 && it - K.last_reset_it >= o.restart_period && mu >= o.mu_init && nreset < o.max_resets && K.ncrawl < ((o.fresh_restart & 4) ? 2 : 1);
          const bool overreg = o.reset_delta > 0.0 && K.delta_last > o.reset_delta && nreset < o.max_resets;
          const bool lost = (o.fresh_restart & 8) && o.restart_period > 0 && mu < o.mu_init && pr > 1e-3 && nreset < o.max_resets &&
                            ((it - K.last_mu_it >= 2 * o.restart_period && it - K.last_reset_it >= o.restart_period) || K.wd_count >= 3);
          if (jam || stalled || overreg || lost) {
            K.flag = 2;
            K.last_reset_it = it;
            if (stalled) K.ncrawl++;
            K.nreset = nreset + 1;
            K.fresh = (((o.fresh_restart & 2) && nreset + 1 == 2) || ((o.fresh_restart & 1) && nreset + 1 == 1 && !stalled && !lost)) ? 1 : 0;
            if (K.fresh) { if (K.clip_k_cur > 1) K.clip_k_cur = 2; K.th_max = 0.0; }
            K.mu = o.mu_init; K.nfilt = 0; K.delta_last = 0.0; K.need_reg_streak = 0; K.wd_count = 0; K.cutstreak = 0; K.force_step = 0;
            K.it = it + 1;
          }
        }
        if (give_up) {      // the solve would end here as NUMERICAL / MAX_ITER: enter the feasibility phase once
          if (K.stalled) K.status = LANDING_STALLED;
          if (o.feas_phase && !K.feas_used && o.max_iter > 0 && K.it < K.hard_lim) {
            K.flag = 4;
            K.feas = 1; K.n_feas++; K.feas_used = K.n_feas >= o.feas_max ? 1 : 0; K.status = LANDING_MAX_ITER; K.lim = K.it + o.max_iter; if (K.lim > K.hard_lim) K.lim = K.hard_lim; K.fstat = -1;
            K.fjam = 0; K.want_entry = 1; K.fdc = o.feas_delta_dec > 0.0 ? o.feas_delta_dec : o.delta_dec;
            K.mu = o.mu_init; K.nfilt = 0; K.th_max = 0.0; K.delta_last = 0.0; K.need_reg_streak = 0; K.cutstreak = 0; K.force_step = 0; K.wd_count = 0;
            K.it = K.it + 1;
          } else K.flag = 1;
        }
      KD_END();
    }
    if (K.flag == 1) {
      if (tid == 0) {
        K.done = 1; *M.st = K; A.done[m] = 1;
        if (A.win && K.status == LANDING_CONVERGED) atomicMin(&A.win[pm], m);      // (relatives that converge in the same round: the lowest index, whatever the order)
      }
      return;
    }
    if (K.flag == 2) {      // restart: the next round of launches evaluates the derivatives at the re-initialised point
      if (K.fresh) {
        for (int i = tid; i < nx; i += NT) {
          double v = A.x0[(size_t)pm * nx + i];
          if (i < 12) v = lbm[i]; else if (i >= oU && i < oU + 12) v = lbm[12 + (i - oU)];
          M.x[i] = v;
        }
        __syncthreads();
        kd_member_eval_g(A.P, *A.model, N, M.x, M.g, M.wbuf);
        __syncthreads();
      }
      kd_init_slacks(M, ng, lbm, ubm, o);
      kd_point_pass(M, ng, lbm, ubm, K.mu);
      if (tid == 0) { *M.st = K; atomicAdd(A.n_active, 1); A.dlist_next[atomicAdd(A.n_dnext, 1)] = m; }
      return;
    }
    if (K.flag == 3) {      // a feasible point (or negligible violation): the interior-point solve restarts from it; derivatives at the new multipliers next round
      for (int r = tid + 24; r < ng; r += NT) if (lbm[r] == ubm[r]) M.y[r] = 0.0;
      __syncthreads();
      if (o.feas_ret_push > 0.0) kd_init_slacks_return(M, ng, lbm, ubm, o, K.mu); else kd_init_slacks(M, ng, lbm, ubm, o);
      kd_point_pass(M, ng, lbm, ubm, K.mu);
      if (tid == 0) { *M.st = K; A.done[m] = 0; atomicAdd(A.n_active, 1); A.dlist_next[atomicAdd(A.n_dnext, 1)] = m; }
      return;
    }
    if (K.flag == 4) {      // into the feasibility phase from the current point (from the caller's initial guess when the iterate is not finite)
      double big = 0.0;
      for (int i = tid; i < nx; i += NT) { const double v = fabs(M.x[i]); big = fmax(big, v < 1e6 ? v : 1e300); }
      big = block_reduce1(big, RMAX, S.red);
      if (!(big < 1e6)) {
        for (int i = tid; i < nx; i += NT) {
          double v = A.x0[(size_t)pm * nx + i];
          if (i < 12) v = lbm[i]; else if (i >= oU && i < oU + 12) v = lbm[12 + (i - oU)];
          M.x[i] = v;
        }
      }
      __syncthreads();
      kd_member_eval_g(A.P, *A.model, N, M.x, M.g, M.wbuf);
      __syncthreads();
      {
        const double frho = o.feas_rho, mu0 = o.mu_init;
        for (int r = tid + 24; r < ng; r += NT) {
          const double lb = lbm[r], ub = ubm[r], g = M.g[r];
          if (lb == ub) { M.y[r] = 0.0; continue; }
          // slack on the row value; violation variables sized so that both distances start at a comfortable value
          double zl = 0.0, zu = 0.0, n0 = 0.0, q0 = 0.0, wl = 0.0, wu = 0.0;
          if (lb > -INF) { const double v = lb - g; n0 = fmax(v, 0.0) + fmax(1e-2, 0.1 * fabs(v)); zl = fmin(mu0 / (g - lb + n0), 0.5 * frho); wl = frho - zl; }
          if (ub < INF) { const double v = g - ub; q0 = fmax(v, 0.0) + fmax(1e-2, 0.1 * fabs(v)); zu = fmin(mu0 / (ub + q0 - g), 0.5 * frho); wu = frho - zu; }
          M.s[r] = g; M.en[r] = n0; M.ep[r] = q0; M.zL[r] = zl; M.zU[r] = zu; M.wn[r] = wl; M.wp[r] = wu; M.y[r] = zu - zl;
        }
      }
      __syncthreads();
      kd_feas_point_pass(M, ng, lbm, ubm, K.mu, o.feas_rho);
      if (tid == 0) { *M.st = K; A.done[m] = 0; atomicAdd(A.n_active, 1); A.dlist_next[atomicAdd(A.n_dnext, 1)] = m; }
      return;
    }
    // ---------------------------------------------------------------- barrier parameter (monotone)
    for (;;) {
      KD_BEGIN()
        double sd = 1.0, sc = 1.0;
        if (o.barrier_smax > 0.0) {
          sd = fmax(o.barrier_smax, (K.c_ys + K.c_zs) / ((double)(ng - 24) + K.c_nz)) / o.barrier_smax;
          sc = fmax(o.barrier_smax, K.c_zs / K.c_nz) / o.barrier_smax;
        }
        const double mu = K.mu;
        if (fmax(K.e_du / sd, fmax(K.c_pr, K.c_cm / sc)) <= o.kappa_eps * mu && mu > o.tol / 10.0) {
          K.mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
          K.nfilt = 0; K.last_mu_it = K.it; K.wd_count = 0;
          K.flag = 1;
        } else { K.flag = 0; K.tau = fmax(o.tau_min, 1.0 - mu); }
      KD_END();
      if (!K.flag) break;
      if (K.feas) kd_feas_point_pass(M, ng, lbm, ubm, K.mu, o.feas_rho); else kd_point_pass(M, ng, lbm, ubm, K.mu);
    }
    KD_PROF(1);
    // ================================================================ Riccati factorisation with inertia correction (IPOPT's schedule)
    KD_BEGIN()
      const double dl = K.delta_last;
      K.delta = (K.need_reg_streak >= 2 && dl > 0.0) ? fmax(1e-20, dl * ((K.feas && o.feas_delta_dec > 0.0) ? K.fdc : o.delta_dec)) : 0.0;      // (inside the phase the regularisation falls faster, adapted: landing_nlp.h feas_delta_dec)
      { double fl = K.feas ? 0.0 : o.delta_floor;    // proximal term (the cost is terminal only: landing_nlp.h delta_floor; off in the feasibility phase)
        if (o.stag_relief > 0 && K.stag >= o.stag_relief) { for (int e = K.stag - o.stag_relief; e >= 0 && fl >= 1e-12; --e) fl *= 0.1; if (fl < 1e-12) fl = 0.0; }
        K.delta = fmax(K.delta, fl); }
      K.skipped_zero = K.delta > 0.0; K.fact_ok = 0; K.attempt = 0; K.flag = 1; K.nfact++;
      K.stage = 1;
      *M.st = K;
      A.cond_list[atomicAdd(A.n_cond, 1)] = m;      // (the order of the list is the order the hardware ran the workgroups in: every entry is independent work)
    KD_END();
  }
}

// J_I' Sigma J_I and J_I' rho of every interval of the members whose head kernel has prepared an iteration this round: work item w = (entry w / N of the list, interval
// w % N), a fixed grid of workgroups strides over the items (four workgroups per CU: 128 registers, 23.6 KB of LDS) -- in the lock-step tail a handful of members
// iterate, and a launch of one workgroup per (member, interval) of the whole batch cost 0.3 ms of empty workgroups per round
__global__ void __launch_bounds__(KD_THREADS, KD_COND_WGS) landing_kd_condense_kernel(KdSolveArgs A) {
  const int N = A.N, total = *A.n_cond * N;                 // (uniform)
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int m = A.cond_list[w / N], k = w % N;
    const KdMem M = kd_carve(N, A.ws + (size_t)m * A.ws_stride);
#if KD_COND_DENSE
    kd_condense_rows(M, N, k);
#else
    kd_condense_rows_sparse(M, N, k, A.jpat, A.cpat);
#endif
    __syncthreads();
  }
}

__global__ void __launch_bounds__(KD_THREADS, 2) landing_kd_iter_kernel(KdSolveArgs A) {
  if ((int)blockIdx.x >= *A.n_cond) return;      // the members the head kernel has passed on: prepared iterations and pending inertia corrections
  const int m = A.cond_list[blockIdx.x];
  if (m >= A.B) return;
  const int N = A.N, nx = kd_nx(N), ng = kd_ng(N), tid = threadIdx.x, NT = blockDim.x;
  const KdMem M = kd_carve(N, A.ws + (size_t)m * A.ws_stride);
  const int pm = kd_problem_of(A, m);
  if (pm < 0) return;                                       // unused clone slot
  if (M.st->done) return;                                   // (uniform: one global word per member)
  if (!M.st->pending && M.st->stage != 1) return;           // finished, restarted or sent into / out of the feasibility phase by the head kernel: the next round starts afresh
  const landing_solver_opts& o = kd_opts_of(A, m);
  const double* lbm = A.lb + (size_t)pm * ng; const double* ubm = A.ub + (size_t)pm * ng;
  const double* cost = A.cost + (size_t)pm * 24;
  const double INF = INFINITY;
  KdLds& S = KSH;
  KdState& K = S.ks;
  if (tid == 0) { K = *M.st; K.tp = (long long)wall_clock64(); K.pending = 0; K.stage = 0; }
  __syncthreads();
  int tries = 0;
  for (;;) {
    const bool ok = kd_backward(M, N, cost, K.delta);
    KD_BEGIN()
      K.fact_ok = ok ? 1 : 0;
      if (K.attempt == 0) K.first_failed = (K.skipped_zero && !ok) ? 1 : 0;
      K.attempt++;
      K.flag = 0;
      if (!ok && K.attempt < 60) {
        double d = K.delta; const double dl = K.delta_last;
        const bool adapt = K.feas && o.feas_delta_dec > 0.0;
        if (d == 0.0) d = (dl == 0.0) ? o.delta_init : fmax(1e-20, dl * (adapt ? K.fdc : o.delta_dec));
        else if (adapt && K.attempt == 1 && d < dl) d = dl;      // inside the phase: the regularisation of the last iteration is the best guess of what this one needs
#if KD_DELTA_JUMP > 0
        // the first failure at the proximal floor is IPOPT's failure at delta = 0: continue from the last successful regularisation (if that was at
        // most KD_DELTA_JUMP iterations ago), not fourfold from the floor -- a member that needs delta ~ 1e2 .. 1e4 in its first barrier problem
        // re-probes the floor every ninth iteration (need_reg_streak) and spent up to eight attempts = three rounds of the lock-step loop on the way back
        else if (K.attempt == 1 && dl * o.delta_dec > d * o.delta_inc && K.it - K.reg_it <= KD_DELTA_JUMP) d = dl * o.delta_dec;
#endif
        else d *= (dl == 0.0 ? o.delta_inc_first : o.delta_inc);
        if (!(d > 1e40)) { K.delta = d; K.flag = 1; K.nfact++; }
      }
    KD_END();
    if (!K.flag) break;
    if (++tries >= KD_TRIES_PER_ROUND) {
      if (tid == 0) { K.pending = 1; *M.st = K; A.done[m] = 2; atomicAdd(A.n_active, 1); A.dlist_next[atomicAdd(A.n_dnext, 1)] = m; }      // (on the list for the head kernel; the derivative kernels skip it)
      return;
    }
  }
  if (!K.fact_ok) {      // no regularisation made the step computable: give up (status NUMERICAL) or -- once -- continue in the feasibility phase: the next
    // round's stop test sees a non-finite error and takes that path (the point itself is kept)
    if (o.feas_phase && !K.feas_used && !K.feas && o.max_iter > 0) {
      if (tid == 0) { K.c_pr = INFINITY; *M.st = K; A.done[m] = 0; atomicAdd(A.n_active, 1); A.dlist_next[atomicAdd(A.n_dnext, 1)] = m; }
      return;
    }
    if (tid == 0) { K.status = K.stalled ? LANDING_STALLED : LANDING_NUMERICAL; K.done = 1; *M.st = K; A.done[m] = 1; }
    return;
  }
  KD_BEGIN()
    const bool adapt = K.feas && o.feas_delta_dec > 0.0;
    if (adapt) K.fdc = K.attempt <= 1 ? fmax(o.feas_delta_dec, K.fdc * K.fdc) : fmin(0.7, sqrt(K.fdc));      // (attempt counts the factorisations of this iteration)
    if (K.delta > (K.feas ? 0.0 : o.delta_floor)) { K.delta_last = K.delta; K.reg_it = K.it; K.need_reg_streak++; } else K.need_reg_streak = 0;
    if (K.need_reg_streak > 8) K.need_reg_streak = adapt ? 2 : 0;      // (no probe of delta_w = 0 inside the phase: the elastic problem has no objective)
  KD_END();
  KD_PROF(2);
  kd_forward(M, N, lbm, A.jpat);
  KD_PROF(3);
  // ================================================================ dual steps, step bounds, merit data
  {
    const double mu = K.mu;
    // clip_k rule (landing_nlp.h): while the point is far from feasible the step length comes from the clip_k-th largest ratio |ds| / distance;
    // the slacks with a larger one stop at (1 - tau) of their distance (omt > 0 in the passes below).  Without it ONE slack after the other
    // cuts the step by 1 - tau per iteration from the callers' guess (measured on the first GPU batch: a_pr 2e-1, 3e-2, 3e-3 ... 3e-15)
    const bool feas = K.feas != 0;
    const bool clip_now = !feas && K.clip_k_cur > 1 && K.c_pr > o.clip_until;
    double top[4] = {0.0, 0.0, 0.0, 0.0};
    double m_pr = 0.0, m_du = 0.0, th0 = 0.0, bar = 0.0, dphi = 0.0, f0 = 0.0;
    // clip_k > 4: the step length that leaves at most clip_k - 1 slacks blocked comes from a histogram (half-octave buckets of ratio / tau)
    const bool clip_hist = clip_now && K.clip_k_cur > 4;
    const double rtau = 1.0 / K.tau;
    if (clip_hist) { if (tid < 64) S.hist[tid] = 0; __syncthreads(); }
    auto hist_push = [&](double rt) { if (clip_hist && rt * rtau > 1.0) { int b = (int)ceil(2.0 * log2(rt * rtau)); atomicAdd(&S.hist[b < 1 ? 1 : b > 63 ? 63 : b], 1); } };
    if (feas) {      // elastic rows: steps of the eliminated variables, step bounds (a, n, b, q and their multipliers stay positive), merit data
      const double frho = o.feas_rho;
      for (int r = tid + 24; r < ng; r += NT) {
        const double lb = lbm[r], ub = ubm[r], g = M.g[r];
        if (lb == ub) { th0 += fabs(g - lb); continue; }
        const double s = M.s[r], ds = M.ds[r];
        th0 += fabs(g - s);
        if (lb > -INF) {
          const double n = M.en[r], a = s - lb + n, z = M.zL[r], w = M.wn[r];
          const ElStep e = el_step(1.0, a, n, z, w, mu, frho, ds);
          m_pr = fmax(m_pr, fmax(-e.da / a, -e.dn / n)); m_du = fmax(m_du, fmax(-e.dz / z, -e.dw / w));
          bar -= log(a * n); dphi += frho * e.dn - mu * (e.da / a + e.dn / n); f0 += frho * n;
        }
        if (ub < INF) {
          const double q = M.ep[r], b = ub + q - s, z = M.zU[r], w = M.wp[r];
          const ElStep e = el_step(-1.0, b, q, z, w, mu, frho, ds);
          m_pr = fmax(m_pr, fmax(-e.da / b, -e.dn / q)); m_du = fmax(m_du, fmax(-e.dz / z, -e.dw / w));
          bar -= log(b * q); dphi += frho * e.dn - mu * (e.da / b + e.dn / q); f0 += frho * q;
        }
      }
    } else
    for (int r = tid + 24; r < ng; r += NT) {
      // (every array of the row is loaded before the first test: under the tests each load waited for its own round trip -- 3 per row instead of 1)
      const double lb = lbm[r], ub = ubm[r], g = M.g[r], s = M.s[r], ds = M.ds[r], zl_ = M.zL[r], zu_ = M.zU[r];
      if (lb == ub) { th0 += fabs(g - lb); continue; }
      th0 += fabs(g - s);
      double dprod = 1.0;
      if (lb > -INF) {
        const double d = s - lb, rd = 1.0 / d, zl = zl_;
        const double dz = -zl * rd * ds + (mu * rd - zl);
        m_pr = fmax(m_pr, -ds * rd); top4_push(top, -ds * rd); hist_push(-ds * rd); m_du = fmax(m_du, -dz / zl);
        dprod = d; dphi -= mu * ds * rd;
      }
      if (ub < INF) {
        const double d = ub - s, rd = 1.0 / d, zu = zu_;
        const double dz = zu * rd * ds + (mu * rd - zu);
        m_pr = fmax(m_pr, ds * rd); top4_push(top, ds * rd); hist_push(ds * rd); m_du = fmax(m_du, -dz / zu);
        dprod *= d; dphi += mu * ds * rd;
      }
      bar -= log(dprod);
    }
    if (tid < 12 && !feas) { const double d = M.x[12 * N + tid] - cost[12 + tid], qn = cost[tid]; f0 = qn * d * d; dphi += 2.0 * qn * d * M.dx[12 * N + tid]; }
    double v[6] = {m_pr, m_du, th0, bar, dphi, f0}; const int op[6] = {RMAX, RMAX, RSUM, RSUM, RSUM, RSUM};
    block_reduce<6>(v, op, S.red);
    if (clip_now) block_top4(top, S.red);         // (uniform: clip_now comes from K)
    KD_BEGIN_SYNCED()
      const double tau = K.tau;
      K.a_pr = (v[0] > tau) ? tau / v[0] : 1.0;
      if (clip_now) { const double rk = top[(K.clip_k_cur > 4 ? 4 : K.clip_k_cur) - 1]; K.a_pr = (rk > tau) ? tau / rk : 1.0; }
      if (clip_hist) {
        int cum = 0, b = 63;
        for (; b >= 1; --b) { if (cum + S.hist[b] > K.clip_k_cur - 1) break; cum += S.hist[b]; }
        K.a_pr = b >= 1 ? fmax(K.a_pr, exp2(-0.5 * (double)b)) : 1.0;      // (never below the 4th-ratio rule: the top bucket is open-ended)
      }
      K.omt = clip_now ? 1.0 - tau : -1.0;
      K.a_du = (v[1] > tau) ? tau / v[1] : 1.0;
      K.th0 = v[2]; K.dphi = v[4]; K.ph0 = v[5] + mu * v[3]; K.fval = v[5];
      if (K.th_max == 0.0) K.th_max = 1e4 * fmax(1.0, v[2]);
      K.alpha = K.a_pr; K.s_corr = 0.0; K.accepted = 0; K.armijo_step = 0; K.ls_done = K.a_pr > 1e-10 ? 0 : 1;
    KD_END();
#ifdef LANDING_KD_BLOCKERS      // development aid: the rows whose slack sets the primal step length of this iteration
    if (!feas) {
      const double apr = K.a_pr;
      for (int r = tid + 24; r < ng; r += NT) {
        const double lb = lbm[r], ub = ubm[r]; if (lb == ub) continue;
        const double s = M.s[r], ds = M.ds[r];
        const double rl = lb > -INF ? -ds / (s - lb) : 0.0, ru = ub < INF ? ds / (ub - s) : 0.0;
        if (fmax(rl, ru) * apr >= 0.25) printf("  blk it %d a_pr %.2e row %d k %d j %d %s ratio %.2e dist %.2e ds %.2e g %.3e s %.3e z %.2e\n", K.it, apr, r, r < 48 ? -1 : (r - 48) / 141, r < 48 ? r : (r - 48) % 141, rl > ru ? "L" : "U", fmax(rl, ru), rl > ru ? s - lb : ub - s, ds, M.g[r], s, rl > ru ? M.zL[r] : M.zU[r]);
      }
    }
#endif
  }
  KD_PROF(4);
  // ================================================================ filter line search
  while (!K.ls_done) {
    const double alpha = K.alpha, mu = K.mu, omt = K.omt;
    for (int i = tid; i < nx; i += NT) M.xt[i] = M.x[i] + alpha * M.dx[i];
    __syncthreads();
    kd_member_eval_g(A.P, *A.model, N, M.xt, M.gt, M.wbuf);
    __syncthreads();
    double tht = 0.0, bt = 0.0, ft = 0.0;
    const bool feas = K.feas != 0;
    if (feas) {      // elastic rows at the trial step length: theta over all rows, merit = rho_pen (n + q) - mu sum of logs (mu applied below)
      const double frho = o.feas_rho;
      for (int r = tid + 24; r < ng; r += NT) {
        const double lb = lbm[r], ub = ubm[r], g = M.gt[r];
        if (lb == ub) { tht += fabs(g - lb); continue; }
        const double s0 = M.s[r], ds = M.ds[r], s = s0 + alpha * ds;
        tht += fabs(g - s);
        if (lb > -INF) { const double n0 = M.en[r]; const ElStep e = el_step(1.0, s0 - lb + n0, n0, M.zL[r], M.wn[r], mu, frho, ds); const double n = n0 + alpha * e.dn; bt -= log((s - lb + n) * n); ft += frho * n; }
        if (ub < INF) { const double q0 = M.ep[r]; const ElStep e = el_step(-1.0, ub + q0 - s0, q0, M.zU[r], M.wp[r], mu, frho, ds); const double q = q0 + alpha * e.dn; bt -= log((ub + q - s) * q); ft += frho * q; }
      }
    } else
    for (int r = tid + 24; r < ng; r += NT) {
      const double lb = lbm[r], ub = ubm[r], g = M.gt[r], so = M.s[r], dso = M.ds[r];      // (loads before the tests: see the step-bound pass)
      if (lb == ub) { tht += fabs(g - lb); continue; }
      double s = so + alpha * dso;
      if (omt > 0.0) { if (lb > -INF) s = fmax(s, lb + omt * (so - lb)); if (ub < INF) s = fmin(s, ub - omt * (ub - so)); }
      tht += fabs(g - s);
      bt -= log((lb > -INF ? s - lb : 1.0) * (ub < INF ? ub - s : 1.0));
    }
    if (tid < 12 && !feas) { const double d = M.xt[12 * N + tid] - cost[12 + tid]; ft = cost[tid] * d * d; }
    { double v[3] = {tht, bt, ft}; const int op[3] = {RSUM, RSUM, RSUM}; block_reduce<3>(v, op, S.red); tht = v[0]; bt = v[1]; ft = v[2]; }
    KD_BEGIN_SYNCED()
      K.ntrial++;
      const double th_min = 1e-4, th_floor = o.theta_floor * o.tol;
      const double th0 = K.th0, ph0 = K.ph0, dphi = K.dphi;
      const int nfilt = K.nfilt;
      const double pht = ft + mu * bt;
      bool ok_f = (tht <= K.th_max) && (pht < 1e300) && (pht > -1e300) && (tht < 1e300);
      for (int e = 0; e < nfilt && ok_f; ++e) if (tht >= fmax(K.filt_th[e], th_floor) && pht >= K.filt_ph[e]) ok_f = false;
      const bool switching = (dphi < 0.0) && (th0 <= th_min) && (alpha * pow(-dphi, 2.3) > 1.0 * pow(th0, 1.1));
      bool accepted = false, done = false;
      if (ok_f) {
        if (switching) { if (pht <= ph0 + 1e-8 * alpha * dphi) { accepted = true; K.armijo_step = 1; } }
        else if (tht <= fmax((1.0 - 1e-5) * th0, th_floor) || pht <= ph0 - 1e-8 * th0) accepted = true;
      }
      if (K.force_step && ok_f) { accepted = true; K.nfilt = 0; done = true; }      // watchdog: the step to the boundary is taken (it must still pass theta_max and the filter entries)
      if (accepted) done = true;
      K.need_corr = 0;
      if (!done) {
        if (o.slack_corr > 0.0 && !K.feas && alpha == K.a_pr && tht >= th0) { K.need_corr = 1; K.ft = ft; }
        else { K.alpha = alpha * 0.5; if (!(K.alpha > 1e-10)) done = true; }
      }
      K.accepted = accepted ? 1 : 0; K.ls_done = done ? 1 : 0;
    KD_END();
    if (K.need_corr) {
      // slack correction (landing_nlp.h): the rejected first trial point once more with the inequality slacks moved to g(x_trial)
      double tht2 = 0.0, bt2 = 0.0;
      for (int r = tid + 24; r < ng; r += NT) {
        const double lb = lbm[r], ub = ubm[r], g = M.gt[r];
        if (lb == ub) { tht2 += fabs(g - lb); continue; }
        double s = M.s[r] + alpha * M.ds[r];
        if (omt > 0.0) { const double so = M.s[r]; if (lb > -INF) s = fmax(s, lb + omt * (so - lb)); if (ub < INF) s = fmin(s, ub - omt * (ub - so)); }
        const double lo = lb > -INF ? lb + o.slack_corr * (s - lb) : -INF, hi = ub < INF ? ub - o.slack_corr * (ub - s) : INF;
        s = fmin(fmax(g, lo), hi);
        tht2 += fabs(g - s);
        bt2 -= log((lb > -INF ? s - lb : 1.0) * (ub < INF ? ub - s : 1.0));
      }
      { double v[2] = {tht2, bt2}; const int op[2] = {RSUM, RSUM}; block_reduce<2>(v, op, S.red); tht2 = v[0]; bt2 = v[1]; }
      KD_BEGIN_SYNCED()
        const double th_floor = o.theta_floor * o.tol, th0 = K.th0, ph0 = K.ph0;
        const int nfilt = K.nfilt;
        const double pht2 = K.ft + mu * bt2;
        bool ok2 = (tht2 <= K.th_max) && (pht2 < 1e300) && (pht2 > -1e300);
        for (int e = 0; e < nfilt && ok2; ++e) if (tht2 >= fmax(K.filt_th[e], th_floor) && pht2 >= K.filt_ph[e]) ok2 = false;
        if (ok2 && (tht2 <= fmax((1.0 - 1e-5) * th0, th_floor) || pht2 <= ph0 - 1e-8 * th0)) { K.accepted = 1; K.s_corr = o.slack_corr; K.ls_done = 1; }
        else { K.alpha = alpha * 0.5; if (!(K.alpha > 1e-10)) K.ls_done = 1; }
      KD_END();
    }
  }
  KD_BEGIN()
    const double a_pr = K.a_pr;
    K.force_step = 0;
    if (o.watchdog > 0 && !K.feas) {
      if (K.accepted && K.alpha <= 0.0625 * a_pr) { if (++K.cutstreak >= o.watchdog) { K.force_step = 1; K.cutstreak = 0; K.wd_count++; } }
      else K.cutstreak = 0;
    }
    K.fallback = 0;
    if (!K.accepted) { K.nfilt = 0; K.alpha = fmin(a_pr, o.alpha_fallback); K.fallback = 1; }
    else if (!K.armijo_step) {
      int nfilt = K.nfilt;
      if (nfilt == KD_FILT) { for (int e = 0; e + 1 < KD_FILT; ++e) { K.filt_th[e] = K.filt_th[e + 1]; K.filt_ph[e] = K.filt_ph[e + 1]; } nfilt = KD_FILT - 1; }
      K.filt_th[nfilt] = (1.0 - 1e-5) * K.th0; K.filt_ph[nfilt] = K.ph0 - 1e-8 * K.th0;
      K.nfilt = nfilt + 1;
    }
    if (o.dual_step_cap > 0.0) K.a_du = fmin(K.a_du, o.dual_step_cap * K.alpha);
    K.full_prev = (K.accepted && K.alpha == 1.0 && K.a_du == 1.0 && K.attempt <= 1) ? 1 : 0;
    if (o.feas_jam > 0) K.fjam = (!K.feas && K.alpha < 1e-2) ? K.fjam + 1 : (K.fjam > 2 ? K.fjam - 2 : 0);
  KD_END();
  if (K.fallback) {
    const double alpha = K.alpha;
    for (int i = tid; i < nx; i += NT) M.xt[i] = M.x[i] + alpha * M.dx[i];
    __syncthreads();
    kd_member_eval_g(A.P, *A.model, N, M.xt, M.gt, M.wbuf);
    __syncthreads();
  }
  KD_PROF(5);
  // ================================================================ accept the trial point; errors, Sigma, rho of the new iterate
  for (int i = tid; i < nx; i += NT) M.x[i] = M.xt[i];
  if (K.feas) {      // elastic rows: primal variables with alpha, multipliers with a_du (kept inside the kappa_Sigma band), then the quantities of the new point
    const double alpha = K.alpha, a_du = K.a_du, mu = K.mu, frho = o.feas_rho;
    for (int r = tid; r < ng; r += NT) {
      const double lb = lbm[r], ub = ubm[r];
      M.g[r] = M.gt[r];
      if (r < 24) continue;
      if (lb == ub) { M.y[r] = M.y[r] + alpha * (M.yn[r] - M.y[r]); continue; }
      const double s0 = M.s[r], ds = M.ds[r], s = s0 + alpha * ds;
      double zl = 0.0, zu = 0.0;
      if (lb > -INF) {
        const double n0 = M.en[r], z0 = M.zL[r], w0 = M.wn[r];
        const ElStep e = el_step(1.0, s0 - lb + n0, n0, z0, w0, mu, frho, ds);
        const double n = n0 + alpha * e.dn, a = s - lb + n;
        zl = fmin(fmax(z0 + a_du * e.dz, 1e-10 * mu / a), 1e10 * mu / a);
        M.wn[r] = fmin(fmax(w0 + a_du * e.dw, 1e-10 * mu / n), 1e10 * mu / n); M.en[r] = n;
      }
      if (ub < INF) {
        const double q0 = M.ep[r], z0 = M.zU[r], w0 = M.wp[r];
        const ElStep e = el_step(-1.0, ub + q0 - s0, q0, z0, w0, mu, frho, ds);
        const double q = q0 + alpha * e.dn, b = ub + q - s;
        zu = fmin(fmax(z0 + a_du * e.dz, 1e-10 * mu / b), 1e10 * mu / b);
        M.wp[r] = fmin(fmax(w0 + a_du * e.dw, 1e-10 * mu / q), 1e10 * mu / q); M.ep[r] = q;
      }
      M.s[r] = s; M.zL[r] = zl; M.zU[r] = zu; M.y[r] = zu - zl;
    }
    __syncthreads();
    kd_feas_point_pass(M, ng, lbm, ubm, mu, frho);
    KD_BEGIN()
      K.it++;
      { const long long n_ = (long long)wall_clock64(); K.prof[6] += (double)(n_ - K.tp); K.tp = n_; }
      *M.st = K;
      A.done[m] = 0;
      atomicAdd(A.n_active, 1); A.dlist_next[atomicAdd(A.n_dnext, 1)] = m;
    KD_END();
  } else {
    const double alpha = K.alpha, a_du = K.a_du, mu = K.mu, s_corr = K.s_corr, omt = K.omt;
    double npr = 0.0, nco = 0.0, ncm = 0.0, nys = 0.0, nzs = 0.0, nnz = 0.0;
    for (int r = tid; r < ng; r += NT) {
      const double lb = lbm[r], ub = ubm[r], g = M.gt[r];
      const double y_ = M.y[r], yn0_ = M.yn[r], so = M.s[r], ds = M.ds[r], zlo_ = M.zL[r], zuo_ = M.zU[r];      // (loads before the tests: see the step-bound pass)
      M.g[r] = g;
      double sg = 0.0, rh = 0.0;
      if (r >= 24) {
        if (lb == ub) { const double yn_ = y_ + alpha * (yn0_ - y_); M.y[r] = yn_; nys += fabs(yn_); npr = fmax(npr, fabs(g - lb)); }
        else {
          double s = so + alpha * ds;
          if (omt > 0.0) { if (lb > -INF) s = fmax(s, lb + omt * (so - lb)); if (ub < INF) s = fmin(s, ub - omt * (ub - so)); }
          if (s_corr > 0.0) { const double lo = lb > -INF ? lb + s_corr * (s - lb) : -INF, hi = ub < INF ? ub - s_corr * (ub - s) : INF; s = fmin(fmax(g, lo), hi); }
          double zl = 0.0, zu = 0.0;
          if (lb > -INF) {
            const double dold = so - lb, zo = zlo_, dz = -zo / dold * ds + (mu / dold - zo), d = s - lb;
            zl = fmin(fmax(zo + a_du * dz, 1e-10 * mu / d), 1e10 * mu / d);
            nco = fmax(nco, d * zl); ncm = fmax(ncm, fabs(d * zl - mu)); sg += zl / d; rh -= mu / d; nzs += zl; nnz += 1.0;
          }
          if (ub < INF) {
            const double dold = ub - so, zo = zuo_, dz = zo / dold * ds + (mu / dold - zo), d = ub - s;
            zu = fmin(fmax(zo + a_du * dz, 1e-10 * mu / d), 1e10 * mu / d);
            nco = fmax(nco, d * zu); ncm = fmax(ncm, fabs(d * zu - mu)); sg += zu / d; rh += mu / d; nzs += zu; nnz += 1.0;
          }
          npr = fmax(npr, fabs(g - s));
          rh += sg * (g - s);
          M.s[r] = s; M.zL[r] = zl; M.zU[r] = zu; M.y[r] = zu - zl; nys += fabs(zu - zl);
        }
      }
      M.sig[r] = sg; M.rho[r] = rh;
    }
    double v[6] = {npr, nco, ncm, nys, nzs, nnz}; const int op[6] = {RMAX, RMAX, RMAX, RSUM, RSUM, RSUM};
    block_reduce<6>(v, op, S.red);
    KD_BEGIN_SYNCED()
      K.c_pr = v[0]; K.c_co = v[1]; K.c_cm = v[2]; K.c_ys = v[3]; K.c_zs = v[4]; K.c_nz = fmax(v[5], 1.0);
      K.it++;
      { const long long n_ = (long long)wall_clock64(); K.prof[6] += (double)(n_ - K.tp); K.tp = n_; }
      *M.st = K;
      A.done[m] = 0;      // (2 while the member was pending: the next launch of the derivative kernels must see its new x)
      atomicAdd(A.n_active, 1); A.dlist_next[atomicAdd(A.n_dnext, 1)] = m;
    KD_END();
  }
}

// ---- portfolio: the originals still iterating get their clone slots (one launch, one thread; wave w uses its own slot range) -----------
__global__ void __launch_bounds__(KD_THREADS) landing_kd_clone_kernel(KdSolveArgs A, int wave) {
  if (blockIdx.x != 0) return;
  __shared__ int cnt[KD_THREADS];
  const int tid = threadIdx.x, per = (A.B0 + KD_THREADS - 1) / KD_THREADS, a0 = tid * per, a1 = (a0 + per < A.B0) ? a0 + per : A.B0;
  int n = 0;
  for (int a = a0; a < a1; ++a) n += (A.done[a] != 1 && !A.cloned[a]) ? 1 : 0;
  cnt[tid] = n;
  __syncthreads();
  int nf = 0;
  for (int t = 0; t < tid; ++t) nf += cnt[t];
  for (int a = a0; a < a1 && nf < A.F; ++a) {
    if (A.done[a] == 1 || A.cloned[a]) continue;
    A.cloned[a] = 1;
    for (int v = 0; v < KD_NVAR; ++v) A.src[(wave * KD_NVAR + v) * A.F + nf] = a;
    ++nf;
  }
}

// ---- end of a solve: outputs (the J blocks of the final (x, y) are in the workspace) ------------------------------------------------
__global__ void __launch_bounds__(KD_THREADS) landing_kd_finish_kernel(KdSolveArgs A) {
  const int m = blockIdx.x;
  if (m >= A.B0) return;
  const int N = A.N, nx = kd_nx(N), ng = kd_ng(N), tid = threadIdx.x, NT = blockDim.x;
  int wm = (A.win && A.win[m] != KD_NOWIN) ? A.win[m] : m;      // portfolio: the member of the family that converged first (else the original)
  if (A.win && wm == m && A.cloned[m]) {      // no member of the family converged and the original ends undecided: a clone's certificate of local infeasibility (an elastic KKT point reached on
    const KdState* s0 = kd_carve(N, A.ws + (size_t)m * A.ws_stride).st;      // another path) decides it -- the lowest slot, whatever order the hardware ran things in (it used to be discarded)
    int best = KD_NOWIN;
    if (!(s0->done && (s0->status == LANDING_CONVERGED || s0->status == LANDING_INFEASIBLE))) {
      for (int c = tid; c < A.B - A.B0; c += NT)
        if (A.src[c] == m) { const KdState* sc = kd_carve(N, A.ws + (size_t)(A.B0 + c) * A.ws_stride).st; if (sc->done && sc->status == LANDING_INFEASIBLE) best = best < A.B0 + c ? best : A.B0 + c; }
    }
    KSH.hist[tid & 63] = KD_NOWIN;
    __syncthreads();
    if (best != KD_NOWIN) atomicMin(&KSH.hist[0], best);
    __syncthreads();
    if (KSH.hist[0] != KD_NOWIN) wm = KSH.hist[0];
    __syncthreads();
  }
  const KdMem M = kd_carve(N, A.ws + (size_t)wm * A.ws_stride);
  const double* lbm = A.lb + (size_t)m * ng; const double* ubm = A.ub + (size_t)m * ng;
  const double* cost = A.cost + (size_t)m * 24;
  const double INF = INFINITY;
  KdLds& S = KSH;
  const int oU = 12 * (N + 1) + 12 * N;
  kd_grad_lag(M, N, cost, 1.0);
  // multipliers of the fixed rows from stationarity of X(:,1), c(:,1): lam = -(grad f + J' y)
  if (tid < 24) M.y[tid] = -M.gx[tid < 12 ? tid : oU + (tid - 12)];
  double du = 0.0, fo = 0.0;
  for (int i = tid; i < nx; i += NT) { const bool fixed = i < 12 || (i >= oU && i < oU + 12); if (!fixed) du = fmax(du, fabs(M.gx[i])); }
  if (tid < 12) { const double d = M.x[12 * N + tid] - cost[12 + tid]; fo = cost[tid] * d * d; }
  __syncthreads();
  double kp = 0.0, kc = 0.0;
  for (int r = tid; r < ng; r += NT) {
    const double lb = lbm[r], ub = ubm[r], g = M.g[r], lam = M.y[r];
    kp = fmax(kp, fmax(lb - g, fmax(g - ub, 0.0)));
    if (lb != ub) {
      const double dist = lam > 0.0 ? ub - g : g - lb;
      if (lam != 0.0 && dist < INF) kc = fmax(kc, fabs(lam * dist));
    }
  }
  { double v[4] = {kp, kc, du, fo}; const int op[4] = {RMAX, RMAX, RMAX, RSUM}; block_reduce<4>(v, op, S.red); kp = v[0]; kc = v[1]; du = v[2]; fo = v[3]; }
  for (int i = tid; i < nx; i += NT) A.x_out[(size_t)m * nx + i] = M.x[i];
  if (A.lam_out) for (int r = tid; r < ng; r += NT) A.lam_out[(size_t)m * ng + r] = M.y[r];
  if (tid == 0) {
    const KdState& K = *M.st;
    int status = K.status;
    if (!K.done) status = LANDING_MAX_ITER;
    // the stop test of the iteration kernel used the slack-based primal error; the report is the reference-consistent residual
    if (A.f_out) A.f_out[m] = fo;
    if (A.status) A.status[m] = status;
    if (A.iters) A.iters[m] = K.it;
    if (A.kkt) { A.kkt[3 * m] = kp; A.kkt[3 * m + 1] = du; A.kkt[3 * m + 2] = kc; }
  }
}

#undef KD_BEGIN
#undef KD_BEGIN_SYNCED
#undef KD_END
#undef KD_PROF

}  // namespace landing
