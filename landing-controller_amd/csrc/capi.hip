// capi.hip -- C ABI of liblanding_mi355x.so (include/landing_nlp.h): context, launches, host copies.
// There is no CPU fallback anywhere in this file: every entry point either runs the HIP kernels on
// the selected gfx950 device or returns an error code.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <atomic>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/landing_nlp.h"
#include "layout.hpp"
#include "srbm_stage.hpp"
#include "eval_kernels.hip"
#include "solver_kernels.hip"
#include "vbl_kernels.hip"
#include "rbd_kernels.hip"
#include "wb_kernels.hip"
#include "kd_solver_kernels.hip"

using landing::Layout;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(LANDING_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct landing_ctx {
  Layout L;
  int device;
  int* d_edge_map = nullptr;
  landing::SolverWorkspace ws;
  double* d_prof = nullptr;
  double* d_vbl = nullptr;
  double* h_vbl = nullptr; hipEvent_t vbl_copied = nullptr;      // pinned staging block of landing_riccati_gains_batch and the event behind its copy
  landing::RbdModel* d_rbd = nullptr;     // uploaded by landing_rbd_set_model
  int2* d_rc_map = nullptr;               // landing_eval_hess_rc_batch: source nonzero + running-cost code of every entry of the extended pattern
  double* d_fb_scratch = nullptr; size_t fb_scratch_n = 0;   // qdd and H^-1 per knot of the exact floating-base linearisation
  double* d_h4 = nullptr; size_t h4_cap = 0;   // ... and its scratch for the casadi_s4-pattern nonzeros
  // function layer: the Jacobian, Hessian and residual kernels of one landing_eval_batch call are independent; for large
  // batches they run on two auxiliary streams forked from / joined to the caller's stream so that their ramps and tails overlap
  hipStream_t host_stream = nullptr;  // stream of the *_host entry points (copies + launch), created on first use
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  int sweep_split = 0;               // dev (LANDING_SWEEP_SPLIT=<n>): bit 0: the Jacobian stream as two launches (X_k columns, U_k columns), bit 1: the Hessian stream too -- measured, no gain
  bool sweep_concurrent = true;      // Q (576) | F (576) | 1/diag(R) (12) of the last landing_riccati_gains_batch call
  // every scratch block above is re-used by the next call of its entry point, possibly on another stream: the launches that use it are
  // fenced by this event (recorded behind them, waited for before the next writer / reader touches the block) -- ADVICE r2
  hipEvent_t scratch_done = nullptr;
  double* d_kd_ws = nullptr; size_t kd_cap = 0; int* d_kd_active = nullptr; int* d_kd_done = nullptr; int kd_done_cap = 0;      // workspace of landing_kinodyn_solve_batch (kd_capi.inc), count of members still iterating
  int kd_jpat_nnz[2] = {0, 0}; size_t kd_cpat_off = 0;
  void* d_kd_jpat = nullptr;      // landing::KdJPat: structural non-zeros of the kinodynamic NLP's Jacobian blocks (kd_ensure_jpat, solver_capi.inc)
  unsigned char* d_kd_pairs = nullptr; int kd_npair = 0; int rbd_std_base = 0;      // structurally non-zero pairs of a Hessian block of the kinodynamic NLP ([2][kd_npair]: i | j; solver_capi.inc, kd_ensure_pairs)
  const double* wb_skip = nullptr;      // landing_wb_skip_taken: consumed by the next landing_wb_rollout
  int wb_semi = 0;             // integrator of the whole-body loop: 0 explicit Euler, 1 semi-implicit Euler (landing_wb_set_integrator)
  bool rbd_arrow = false;      // the model set by landing_rbd_set_model is "six base joints + four 3-joint legs on the base": H is block-arrow (wb_kernels.hip)
  std::mutex mu;      // serialises landing_solve_batch calls on one context (the workspace belongs to the context)
  std::mutex hs_mu; landing_stream* hs_obj = nullptr; int hs_lanes = 0;      // stream of landing_solve_stream_host, one call at a time (child contexts and their workspaces are kept between calls)
};
// call with ctx->mu held: the stream waits for the previous user of the context's scratch blocks
static hipError_t scratch_acquire(landing_ctx* ctx, hipStream_t s) {
  if (!ctx->scratch_done) { hipError_t e = hipEventCreateWithFlags(&ctx->scratch_done, hipEventDisableTiming); if (e != hipSuccess) return e; return hipSuccess; }
  return hipStreamWaitEvent(s, ctx->scratch_done, 0);
}
static hipError_t scratch_release(landing_ctx* ctx, hipStream_t s) { return hipEventRecord(ctx->scratch_done, s); }

extern "C" {

const char* landing_last_error(void) { return g_err.c_str(); }

void landing_form_default(landing_form* f) {
  f->kin_box[0] = 0.15; f->kin_box[1] = 0.15; f->kin_box[2] = 0.30;
  f->kin_z_off = 0.05; f->comp_eps = 1e-3; f->slip_eps = 1e-2;
  f->run_cost = 0;
  { const landing::Layout d = landing::make_layout(3);
    for (int i = 0; i < 12; ++i) { f->QX[i] = 0.0; f->p_hip[i] = d.p_hip[i]; }
    for (int i = 0; i < 3; ++i) { f->Qc[i] = 0.0; f->Qf[i] = 0.0; f->f_ref[i] = 0.0; } }
}

void landing_solver_opts_default(landing_solver_opts* o) {
  memset(o, 0, sizeof(*o));
  o->tol = 1e-6; o->max_iter = 3000; o->mu_init = 0.0 /* auto: 0.5 for the terminal-cost form, the reference's 0.1 for forms with a running cost (landing_nlp.h) */; o->bound_push = 0.0 /* auto: 1.0 / the reference's 0.5 */; o->bound_frac = 0.1;
  o->kappa_eps = 0.0 /* auto: 120 for the terminal-cost form, 10 for forms with a running cost (landing_nlp.h) */; o->kappa_mu = 0.2; o->theta_mu = 0.0 /* auto: 1.8 / IPOPT's 1.5 */; o->max_soc = 0; o->max_resets = 8; o->reset_du = 1e9;
  o->delta_init = 1e-4; o->delta_inc_first = 10.0; o->delta_inc = 4.0; o->delta_dec = 0.5; o->tau_min = 0.9; o->alpha_fallback = 1e-2;
  o->stage_local_reg = 0; o->sticky_delta = 0; o->restart_period = 75; o->reset_delta = 1e5; o->dispatch_order = 1;
  o->clip_k = 4; o->clip_until = 0.03; o->theta_floor = 30.0; o->fresh_restart = 9; o->dual_step_cap = 1.0; o->slack_corr = 0.9; o->watchdog = 3; o->barrier_smax = 1.0; o->factor_fp32 = 0; o->jam_clip = 2; o->stag_relief = 3; o->feas_jam = 8; o->feas_stat = 25;
  o->kd_clone_after = 0; o->kd_clone_max = 0; o->kd_clone_iter = 0;
  o->feas_phase = 1; o->feas_rho = 1000.0; o->feas_cert = 1e-4;
  o->delta_floor = 3e-4;
  o->feas_back = 0.2; o->feas_max = 3; o->feas_delta_dec = 0.1; o->feas_ret_push = 0.01; o->feas_ret_mu = 0.01; o->feas_resume = 1; o->feas_polish = 1e-8;
}

void landing_solver_opts_warm(landing_solver_opts* o) {
  landing_solver_opts_default(o);
  o->bound_push = 1e-4; o->bound_frac = 1e-4; o->mu_init = 1e-4;
  o->clip_k = 0;                 // a shifted plan starts next to the boundary of many rows: classic rule
  o->fresh_restart = 0;          // the initial guess of a tick IS the previous plan
  o->restart_period = 0;         // the crawl detector is tied to mu_init; a warm start is not expected to crawl
  o->jam_clip = 0; o->stag_relief = 0; o->feas_jam = 0;      // cold-start rules
  o->feas_phase = 0;             // a tick that hits its iteration cap continues at the next tick; no restoration inside a tick
  o->max_iter = 14;              // real-time iteration cap: a tick never runs longer than ~14 x 0.56 ms at one NLP per CU; a member
                                 // that needs more keeps its improved iterate and continues at the next tick (status 1)
}

long long landing_nx(int N) { return 36LL * N + 12; }
long long landing_ng(int N) { return 104LL * N + 12; }
long long landing_np(int N) { return 13LL * N + 94; }
long long landing_np_ccc(int N) { return 37LL * N + 112; }      /* parameter vector of the N=41 script (landing_form.run_cost = 2) */
long long landing_ctx_np(const landing_ctx* ctx) { return ctx ? ctx->L.np : 0; }
long long landing_nnz_jac(int N) { return 36 + 385LL * (N - 1) + 313; }
long long landing_nnz_hess(int N) { return 177LL * N + 12LL * (N - 1) + 12; }
long long landing_sweep_bytes_per_member(int N) {
  // read x,p,lam_g ; write g, grad_f, jac nz, hess nz
  return 8 * (landing_nx(N) + landing_np(N) + landing_ng(N) + landing_ng(N) + landing_nx(N) + landing_nnz_jac(N) + landing_nnz_hess(N));
}
const char* landing_kernel_name_sweep(void) { return "landing_sweep_kernel"; }

int landing_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- CCS patterns: recorded from the very emission order the kernels use -----------------------
namespace {
struct RecJ {
  std::vector<long long>* rows; std::vector<long long>* colind; long long base_own, base_prev; bool first;
  void col() { colind->push_back((long long)rows->size()); }
  void end() {}
  void put(int r, double) {
    long long g;
    if (r >= 0) g = base_own + r;
    else if (r > -100) { const int i = -1 - r; g = first ? i : base_prev + landing::dyn_row_of_state(i); }
    else { const int c = -100 - r; g = base_prev + 16 + 12 * (c / 6) + 2 + (c % 6); }
    rows->push_back(g);
  }
};
}  // namespace

int landing_pattern_jac(int N, long long* colind, long long* row) {
  if (N < 2 || !colind || !row) return fail(LANDING_E_ARG, "landing_pattern_jac: bad argument");
  const Layout L = landing::make_layout(N);
  std::vector<long long> rx, cx, ru, cu;
  srbm::StageVars z; srbm::StageParams P;
  memset(&z, 0, sizeof(z)); memset(&P, 0, sizeof(P));
  P.mass = 1.0;
  const double fzp[4] = {0, 0, 0, 0};
  for (int k = 0; k < N; ++k) {
    RecJ ex{&rx, &cx, L.g_stage(k), k ? L.g_stage(k - 1) : 0, k == 0};
    RecJ eu{&ru, &cu, L.g_stage(k), k ? L.g_stage(k - 1) : 0, k == 0};
    srbm::stage_jac(z, P, k == 0, k == N - 1, fzp, ex, eu);
  }
  // X_N columns: terminal rows then the identity of stage N-1
  for (int i = 0; i < 12; ++i) {
    cx.push_back((long long)rx.size());
    if (i < 6) { rx.push_back(12 + i); rx.push_back(18 + i); } else { rx.push_back(24 + i - 6); rx.push_back(30 + i - 6); }
    rx.push_back(L.g_stage(N - 1) + landing::dyn_row_of_state(i));
  }
  if ((long long)(rx.size() + ru.size()) != L.nnz_jac) return fail(LANDING_E_ARG, "internal: jac pattern size");
  long long n = 0;
  for (size_t c = 0; c < cx.size(); ++c) colind[n++] = cx[c];
  for (size_t c = 0; c < cu.size(); ++c) colind[n++] = (long long)rx.size() + cu[c];
  colind[n] = L.nnz_jac;
  std::copy(rx.begin(), rx.end(), row);
  std::copy(ru.begin(), ru.end(), row + rx.size());
  return 0;
}

int landing_pattern_hess(int N, long long* colind, long long* row) {
  if (N < 2 || !colind || !row) return fail(LANDING_E_ARG, "landing_pattern_hess: bad argument");
  const Layout L = landing::make_layout(N);
  long long n = 0, c = 0;
  // X columns (order of srbm::stage_hess: pos diag; (pos,e),(e,e); (e,omega),(omega,omega))
  for (int k = 0; k <= N; ++k) {
    const long long X = L.x_X(k);
    if (k == N) { for (int i = 0; i < 12; ++i) { colind[c++] = n; row[n++] = X + i; } break; }
    for (int j = 0; j < 3; ++j) { colind[c++] = n; row[n++] = X + j; }
    for (int a = 0; a < 3; ++a) { colind[c++] = n; for (int i = 0; i < 3; ++i) row[n++] = X + i; for (int b = 0; b <= a; ++b) row[n++] = X + 3 + b; }
    colind[c++] = n; row[n++] = X + 4; row[n++] = X + 5;
    colind[c++] = n; row[n++] = X + 3; row[n++] = X + 4; row[n++] = X + 5; row[n++] = X + 6;
    colind[c++] = n; row[n++] = X + 3; row[n++] = X + 4; row[n++] = X + 5; row[n++] = X + 6; row[n++] = X + 7;
    for (int j = 0; j < 3; ++j) colind[c++] = n;   // v columns: empty
  }
  for (int k = 0; k < N; ++k) {
    const long long X = L.x_X(k), U = L.x_U(k);
    for (int l = 0; l < 4; ++l) for (int j = 0; j < 3; ++j) {
      colind[c++] = n;
      row[n++] = X + j; row[n++] = X + 3; row[n++] = X + 4; row[n++] = X + 5;
      if (k > 0) row[n++] = L.x_U(k - 1) + 12 + 3 * l + 2;
      row[n++] = U + 3 * l + j;
    }
    for (int l = 0; l < 4; ++l) for (int j = 0; j < 3; ++j) {
      colind[c++] = n;
      for (int i = 0; i < 3; ++i) if (i != j) row[n++] = X + i;
      row[n++] = X + 3; row[n++] = X + 4; row[n++] = X + 5;
      for (int i = 0; i < 3; ++i) if (i != j || j == 2) row[n++] = U + 3 * l + i;
    }
  }
  colind[c] = n;
  if (n != L.nnz_hess || c != L.nx) return fail(LANDING_E_ARG, "internal: hess pattern size");
  return 0;
}

long long landing_nnz_hess_rc(int N) { return landing_nnz_hess(N) + 18LL * N; }

// casadi_s4 plus the diagonals the running cost touches and s4 lacks: (omega, omega), (v, v) of X_0..X_{N-1} and (f, f) of every stage
static bool rc_extra_diag(const Layout& L, long long col) {
  if (col < L.x_X(L.N)) return (col % 12) >= 6;
  if (col >= L.x_U(0)) return ((col - L.x_U(0)) % 24) >= 12;
  return false;
}
int landing_pattern_hess_rc(int N, long long* colind, long long* row) {
  if (N < 2 || !colind || !row) return fail(LANDING_E_ARG, "landing_pattern_hess_rc: bad argument");
  const Layout L = landing::make_layout(N);
  std::vector<long long> c4(L.nx + 1), r4(L.nnz_hess);
  if (int e = landing_pattern_hess(N, c4.data(), r4.data())) return e;
  long long n = 0;
  for (long long c = 0; c < L.nx; ++c) {
    colind[c] = n;
    n = std::copy(r4.begin() + c4[c], r4.begin() + c4[c + 1], row + n) - row;
    if (rc_extra_diag(L, c)) row[n++] = c;        // a diagonal: the largest row of an upper-triangular column
  }
  colind[L.nx] = n;
  if (n != landing_nnz_hess_rc(N)) return fail(LANDING_E_ARG, "internal: hess_rc pattern size");
  return 0;
}

// ---- context ----------------------------------------------------------------------------------
landing_ctx* landing_create(int N, int device, const landing_form* form) {
  if (N < 2 || N > 256) { fail(LANDING_E_ARG, "landing_create: N must be in [2,256]"); return nullptr; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { fail(LANDING_E_NODEV, "landing_create: no HIP device (this library has no CPU path)"); return nullptr; }
  if (device < 0 || device >= n) { fail(LANDING_E_ARG, "landing_create: bad device index"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { fail(LANDING_E_HIP, "hipSetDevice failed"); return nullptr; }
  landing_ctx* c = new landing_ctx();
  c->L = landing::make_layout(N);
  c->device = device;
#ifdef LANDING_DEV_SWITCHES      // development builds only (tools/dev): environment switches of the measurements recorded in profiles/r0*_ab_experiments.txt
  { const char* e = getenv("LANDING_SWEEP_SERIAL"); c->sweep_concurrent = !(e && e[0] == '1'); }
  { const char* e = getenv("LANDING_SWEEP_SPLIT"); if (e && e[0] >= '0' && e[0] <= '3') c->sweep_split = e[0] - '0'; }
#endif
  {  // positions of the U_k Jacobian entries of stages 0 / N-1 inside the uniform (middle-stage) emission sequence
    struct RecCodes { std::vector<int>* v; void col() {} void end() {} void put(int r, double) { v->push_back(r); } };
    std::vector<int> cx, cu;
    srbm::StageVars z; srbm::StageParams P;
    memset(&z, 0, sizeof(z)); memset(&P, 0, sizeof(P)); P.mass = 1.0;
    const double fzp[4] = {0, 0, 0, 0};
    RecCodes ex{&cx}, eu{&cu};
    srbm::stage_jac(z, P, false, false, fzp, ex, eu);
    std::vector<int> map(456, -1);
    int nf = 0, nl = 0;
    for (size_t i = 0; i < cu.size() && i < 228; ++i) {
      const int r = cu[i];
      const bool prev_entry = r <= -100;                                   // no-slip rows of stage k-1: absent for stage 0
      const bool slip_entry = r >= 16 && r < 64 && ((r - 16) % 12) >= 2 && ((r - 16) % 12) < 8;   // own no-slip rows: absent for stage N-1
      if (!prev_entry) map[i] = nf++;
      if (!slip_entry) map[228 + i] = nl++;
    }
    if (cu.size() != 228 || nf != 204 || nl != 180 || hipMalloc((void**)&c->d_edge_map, 456 * sizeof(int)) != hipSuccess ||
        hipMemcpy(c->d_edge_map, map.data(), 456 * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      fail(LANDING_E_HIP, "landing_create: edge map setup failed"); delete c; return nullptr;
    }
  }
  if (form) {
    for (int i = 0; i < 3; ++i) c->L.kin_box[i] = form->kin_box[i];
    c->L.kin_z_off = form->kin_z_off; c->L.comp_eps = form->comp_eps; c->L.slip_eps = form->slip_eps;
    c->L.run_cost = form->run_cost == 2 ? 2 : (form->run_cost ? 1 : 0);
    if (c->L.run_cost == 2) landing::layout_ccc_params(c->L);      // the N=41 script's own parameter vector: Uref, QX, Qc, Qf are entries of p
    for (int i = 0; i < 12; ++i) { c->L.QX[i] = form->QX[i]; c->L.p_hip[i] = form->p_hip[i]; }
    for (int i = 0; i < 3; ++i) { c->L.Qc[i] = form->Qc[i]; c->L.Qf[i] = form->Qf[i]; c->L.f_ref[i] = form->f_ref[i]; }
  }
  return c;
}

void landing_destroy(landing_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  if (ctx->hs_obj) { landing_stream_destroy(ctx->hs_obj); ctx->hs_obj = nullptr; }
  ctx->ws.release();
  if (ctx->d_edge_map) (void)hipFree(ctx->d_edge_map);
  if (ctx->d_vbl) (void)hipFree(ctx->d_vbl);
  if (ctx->h_vbl) { (void)hipEventSynchronize(ctx->vbl_copied); (void)hipHostFree(ctx->h_vbl); (void)hipEventDestroy(ctx->vbl_copied); }
  if (ctx->d_rbd) (void)hipFree(ctx->d_rbd);
  if (ctx->d_rc_map) (void)hipFree(ctx->d_rc_map);
  if (ctx->d_h4) (void)hipFree(ctx->d_h4);
  if (ctx->d_fb_scratch) (void)hipFree(ctx->d_fb_scratch);
  for (int i = 0; i < 3; ++i) { if (ctx->aux[i]) (void)hipStreamDestroy(ctx->aux[i]); if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->d_kd_pairs) (void)hipFree(ctx->d_kd_pairs);
  if (ctx->d_kd_jpat) (void)hipFree(ctx->d_kd_jpat);
  if (ctx->d_kd_ws) (void)hipFree(ctx->d_kd_ws);
  if (ctx->d_kd_active) (void)hipFree(ctx->d_kd_active);
  if (ctx->d_kd_done) (void)hipFree(ctx->d_kd_done);
  if (ctx->scratch_done) { (void)hipEventSynchronize(ctx->scratch_done); (void)hipEventDestroy(ctx->scratch_done); }
  if (ctx->host_stream) (void)hipStreamDestroy(ctx->host_stream);
  delete ctx;
}

// ---- function layer -----------------------------------------------------------------------------
int landing_eval_batch(landing_ctx* ctx, int B, const double* d_x, const double* d_p, const double* d_lam_f,
                       const double* d_lam_g, double* d_f, double* d_g, double* d_grad_f, double* d_jac,
                       double* d_hess, double* d_ggx, double* d_ggp, void* stream) {
  if (ctx && B == 0) return 0;          // empty batch: nothing to do
  if (!ctx || B < 0 || !d_x || !d_p) return fail(LANDING_E_ARG, "landing_eval_batch: bad argument");
  if ((d_hess || d_ggx || d_ggp) && !d_lam_g) return fail(LANDING_E_ARG, "landing_eval_batch: lam_g required for hess/grad_gamma");
  if (ctx->L.run_cost && d_hess) return fail(LANDING_E_ARG, "landing_eval_batch: with a running cost the Hessian does not fit the CCS pattern of the terminal-cost NLP; use landing_eval_hess_rc_batch");
  HIP_TRY(hipSetDevice(ctx->device));
  landing::EvalArgs A{d_x, d_p, d_lam_f, d_lam_g, d_f, d_g, d_grad_f, d_jac, d_hess, d_ggx, d_ggp, ctx->d_edge_map, d_g ? 1 : 0};
  if (ctx->L.N < 3) return fail(LANDING_E_ARG, "landing_eval_batch: N >= 3 required");
  hipStream_t s0 = (hipStream_t)stream, sj = s0, sh = s0, sh2 = s0;
  const bool fork = ctx->sweep_concurrent && B >= 512 && d_jac && d_hess;
  if (fork) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->aux[0]) {
      for (int i = 0; i < 3; ++i) { HIP_TRY(hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming)); }
      HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    }
    sj = ctx->aux[0]; sh = ctx->aux[1]; sh2 = ctx->aux[2];
    HIP_TRY(hipEventRecord(ctx->ev_fork, s0));
    HIP_TRY(hipStreamWaitEvent(sj, ctx->ev_fork, 0)); HIP_TRY(hipStreamWaitEvent(sh, ctx->ev_fork, 0)); HIP_TRY(hipStreamWaitEvent(sh2, ctx->ev_fork, 0));
  }
  const size_t lds = (size_t)landing::landing_sweep_tile_rows(ctx->L.N) * landing::TILE_LD * sizeof(double);
  if (d_hess && ctx->sweep_split & 2) {
    hipLaunchKernelGGL((landing::landing_sweep_kernel<1, 2>), dim3(B), dim3(64), lds, sh, ctx->L, B, A);
    hipLaunchKernelGGL((landing::landing_sweep_kernel<1, 1>), dim3(B), dim3(64), lds, sh, ctx->L, B, A);
  } else if (d_hess) hipLaunchKernelGGL((landing::landing_sweep_kernel<1, 0>), dim3(B), dim3(64), lds, sh, ctx->L, B, A);
  if (d_jac && ctx->sweep_split & 1) {      // the Jacobian stream as two launches: U_k columns, X_k columns
    hipLaunchKernelGGL((landing::landing_sweep_kernel<0, 2>), dim3(B), dim3(64), lds + 456 * sizeof(int), sj, ctx->L, B, A);
    hipLaunchKernelGGL((landing::landing_sweep_kernel<0, 1>), dim3(B), dim3(64), lds, sh2, ctx->L, B, A);
  } else if (d_jac) hipLaunchKernelGGL((landing::landing_sweep_kernel<0, 0>), dim3(B), dim3(64), lds + 456 * sizeof(int), sj, ctx->L, B, A);
  if (d_g) hipLaunchKernelGGL((landing::landing_sweep_kernel<2, 0>), dim3(B), dim3(64), lds, s0, ctx->L, B, A);
  // the part of the sweep that needs no stage evaluation has its own light instantiation (eval_kernels.hip)
  const bool heavy = (ctx->L.run_cost && (d_f || d_grad_f)) || (d_g && !A.g_staged) || d_ggx || d_ggp;
  if (heavy) hipLaunchKernelGGL(landing::landing_sweep_misc_kernel<false>, dim3(B), dim3(64), 0, s0, ctx->L, B, A);
  else hipLaunchKernelGGL(landing::landing_sweep_misc_kernel<true>, dim3(B), dim3(64), 0, s0, ctx->L, B, A);
  HIP_TRY(hipGetLastError());
  if (fork) {
    HIP_TRY(hipEventRecord(ctx->ev_join[0], sj)); HIP_TRY(hipEventRecord(ctx->ev_join[1], sh)); HIP_TRY(hipEventRecord(ctx->ev_join[2], sh2));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipStreamWaitEvent(s0, ctx->ev_join[i], 0));
  }
  return 0;
}

// Hessian of the Lagrangian including the running cost, nonzeros in the pattern of landing_pattern_hess_rc
int landing_eval_hess_rc_batch(landing_ctx* ctx, int B, const double* d_x, const double* d_p, const double* d_lam_f,
                               const double* d_lam_g, double* d_hess_rc, void* stream) {
  if (ctx && B == 0) return 0;
  if (!ctx || B < 0 || !d_x || !d_p || !d_lam_g || !d_hess_rc) return fail(LANDING_E_ARG, "landing_eval_hess_rc_batch: bad argument");
  if (ctx->L.N < 3) return fail(LANDING_E_ARG, "landing_eval_hess_rc_batch: N >= 3 required");
  HIP_TRY(hipSetDevice(ctx->device));
  const Layout& L = ctx->L;
  const int nrc = (int)landing_nnz_hess_rc(L.N);
  std::lock_guard<std::mutex> lock(ctx->mu);      // the scratch belongs to the context
  if (!ctx->d_rc_map) {
    std::vector<long long> c4(L.nx + 1), r4(L.nnz_hess), cr(L.nx + 1), rr(nrc);
    if (int e = landing_pattern_hess(L.N, c4.data(), r4.data())) return e;
    if (int e = landing_pattern_hess_rc(L.N, cr.data(), rr.data())) return e;
    std::vector<int2> map(nrc);
    for (long long c = 0; c < L.nx; ++c) {
      long long i4 = c4[c];
      for (long long j = cr[c]; j < cr[c + 1]; ++j) {
        const long long r = rr[j];
        int src = -1, code = 0;
        if (i4 < c4[c + 1] && r4[i4] == r) src = (int)i4++;
        // which running-cost term lands on (r, c)?
        if (r == c && c < L.x_X(L.N)) code = 1 | (int)(c % 12) << 4 | (int)(c / 12) << 8;
        else if (c >= L.x_U(0)) {
          const int k = (int)((c - L.x_U(0)) / 24), u = (int)((c - L.x_U(0)) % 24), a = u % 3;
          if (r == c) code = (u < 12 ? 3 : 4) | a << 4 | k << 8;
          else if (u < 12 && r == L.x_X(k) + a) code = 2 | a << 4 | k << 8;
        }
        map[j] = make_int2(src, code);
      }
      if (i4 != c4[c + 1]) return fail(LANDING_E_ARG, "internal: hess_rc pattern does not contain casadi_s4");
    }
    HIP_TRY(hipMalloc((void**)&ctx->d_rc_map, sizeof(int2) * (size_t)nrc));
    HIP_TRY(hipMemcpy(ctx->d_rc_map, map.data(), sizeof(int2) * (size_t)nrc, hipMemcpyHostToDevice));
  }
  const size_t need = (size_t)B * L.nnz_hess;
  if (ctx->h4_cap < need) {
    if (ctx->d_h4) HIP_TRY(hipFree(ctx->d_h4));      // hipFree waits for the device: no launch still reads the old block
    ctx->d_h4 = nullptr; ctx->h4_cap = 0;
    HIP_TRY(hipMalloc((void**)&ctx->d_h4, need * sizeof(double)));
    ctx->h4_cap = need;
  }
  hipStream_t s0 = (hipStream_t)stream;
  HIP_TRY(scratch_acquire(ctx, s0));
  landing::EvalArgs A{d_x, d_p, d_lam_f, d_lam_g, nullptr, nullptr, nullptr, nullptr, ctx->d_h4, nullptr, nullptr, ctx->d_edge_map, 0};
  hipLaunchKernelGGL((landing::landing_sweep_kernel<1, 0>), dim3(B), dim3(64), (size_t)landing::landing_sweep_tile_rows(L.N) * landing::TILE_LD * sizeof(double), s0, L, B, A);
  hipLaunchKernelGGL(landing::landing_sweep_misc_kernel<true>, dim3(B), dim3(64), 0, s0, L, B, A);     // terminal-cost block of the Hessian
  hipLaunchKernelGGL(landing::landing_hess_rc_kernel, dim3((nrc + 255) / 256, B), dim3(256), 0, s0, L, B, nrc, ctx->d_rc_map, ctx->d_h4, d_p, d_lam_f, d_hess_rc);
  HIP_TRY(hipGetLastError());
  HIP_TRY(scratch_release(ctx, s0));
  return 0;
}

namespace {
struct DevBuf {
  double* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, n * sizeof(double)); }
};
}  // namespace

int landing_eval_batch_host(landing_ctx* ctx, int B, const double* x, const double* p, const double* lam_f,
                            const double* lam_g, double* f, double* g, double* grad_f, double* jac, double* hess,
                            double* ggx, double* ggp) {
  if (!ctx || B <= 0 || !x || !p) return fail(LANDING_E_ARG, "landing_eval_batch_host: bad argument");
  const Layout& L = ctx->L;
  HIP_TRY(hipSetDevice(ctx->device));
  DevBuf dx, dp, dlf, dlg, df, dg, dgf, dj, dh, dgx, dgp;
  const size_t b = (size_t)B;
  HIP_TRY(dx.alloc(b * L.nx)); HIP_TRY(dp.alloc(b * L.np));
  HIP_TRY(hipMemcpy(dx.p, x, b * L.nx * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dp.p, p, b * L.np * 8, hipMemcpyHostToDevice));
  if (lam_f) { HIP_TRY(dlf.alloc(b)); HIP_TRY(hipMemcpy(dlf.p, lam_f, b * 8, hipMemcpyHostToDevice)); }
  if (lam_g) { HIP_TRY(dlg.alloc(b * L.ng)); HIP_TRY(hipMemcpy(dlg.p, lam_g, b * L.ng * 8, hipMemcpyHostToDevice)); }
  if (f) HIP_TRY(df.alloc(b));
  if (g) HIP_TRY(dg.alloc(b * L.ng));
  if (grad_f) HIP_TRY(dgf.alloc(b * L.nx));
  if (jac) HIP_TRY(dj.alloc(b * L.nnz_jac));
  if (hess) HIP_TRY(dh.alloc(b * L.nnz_hess));
  if (ggx) HIP_TRY(dgx.alloc(b * L.nx));
  if (ggp) HIP_TRY(dgp.alloc(b * L.np));
  int rc = landing_eval_batch(ctx, B, dx.p, dp.p, dlf.p, dlg.p, df.p, dg.p, dgf.p, dj.p, dh.p, dgx.p, dgp.p, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  if (f) HIP_TRY(hipMemcpy(f, df.p, b * 8, hipMemcpyDeviceToHost));
  if (g) HIP_TRY(hipMemcpy(g, dg.p, b * L.ng * 8, hipMemcpyDeviceToHost));
  if (grad_f) HIP_TRY(hipMemcpy(grad_f, dgf.p, b * L.nx * 8, hipMemcpyDeviceToHost));
  if (jac) HIP_TRY(hipMemcpy(jac, dj.p, b * L.nnz_jac * 8, hipMemcpyDeviceToHost));
  if (hess) HIP_TRY(hipMemcpy(hess, dh.p, b * L.nnz_hess * 8, hipMemcpyDeviceToHost));
  if (ggx) HIP_TRY(hipMemcpy(ggx, dgx.p, b * L.nx * 8, hipMemcpyDeviceToHost));
  if (ggp) HIP_TRY(hipMemcpy(ggp, dgp.p, b * L.np * 8, hipMemcpyDeviceToHost));
  return 0;
}

int landing_eval_hess_rc_batch_host(landing_ctx* ctx, int B, const double* x, const double* p, const double* lam_f,
                                    const double* lam_g, double* hess_rc) {
  if (!ctx || B <= 0 || !x || !p || !lam_g || !hess_rc) return fail(LANDING_E_ARG, "landing_eval_hess_rc_batch_host: bad argument");
  const Layout& L = ctx->L;
  HIP_TRY(hipSetDevice(ctx->device));
  DevBuf dx, dp, dlf, dlg, dh;
  const size_t b = (size_t)B, nrc = (size_t)landing_nnz_hess_rc(L.N);
  HIP_TRY(dx.alloc(b * L.nx)); HIP_TRY(dp.alloc(b * L.np)); HIP_TRY(dlg.alloc(b * L.ng)); HIP_TRY(dh.alloc(b * nrc));
  HIP_TRY(hipMemcpy(dx.p, x, b * L.nx * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dp.p, p, b * L.np * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dlg.p, lam_g, b * L.ng * 8, hipMemcpyHostToDevice));
  if (lam_f) { HIP_TRY(dlf.alloc(b)); HIP_TRY(hipMemcpy(dlf.p, lam_f, b * 8, hipMemcpyHostToDevice)); }
  if (int rc = landing_eval_hess_rc_batch(ctx, B, dx.p, dp.p, dlf.p, dlg.p, dh.p, nullptr)) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(hess_rc, dh.p, b * nrc * 8, hipMemcpyDeviceToHost));
  return 0;
}

int landing_bounds_batch(landing_ctx* ctx, int B, const double* d_p, double* d_lbg, double* d_ubg, void* stream) {
  if (ctx && B == 0) return 0;
  if (!ctx || B < 0 || !d_p || !d_lbg || !d_ubg) return fail(LANDING_E_ARG, "landing_bounds_batch: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t n = (size_t)B * ctx->L.ng;
  hipLaunchKernelGGL(landing::landing_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ctx->L, B, d_p, d_lbg, d_ubg);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"

#include "solver_capi.inc"
#include "multi_capi.inc"
#include "stream_capi.inc"
#include "kd_capi.inc"
#include "kd_casadi_capi.inc"
