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

namespace landing {

constexpr int NZ_JX = 0, NZ_JU = 157, NZ_JUN = 385, NZ_HX = 613, NZ_HU = 642, NZ_HUN = 802, NZ_TOT = 962;
constexpr int GS = 49;    // LDS row stride of G (48x48)
constexpr int PS = 25;    // LDS row stride of 24x24 matrices
constexpr int YS = 37;    // LDS row stride of 24x36 / 12x36 matrices
// per-stage Riccati record (doubles): K 24x24 | kappa 24 | A^ 12x36 | b 12 | P_k rows of X (12x24) | p_k X part 12
constexpr int RIC_K = 0, RIC_KAP = 576, RIC_AH = 600, RIC_B = 1032, RIC_PX = 1044, RIC_PV = 1332, RIC_STRIDE = 1344;
constexpr int FILT_CAP = 64;

struct SolverWorkspace {
  double* buf = nullptr; size_t cap = 0;
  int* d_tab = nullptr; int* d_stage_tab = nullptr; int n_tab = 0;
  static size_t member_stride(const Layout& L) {
    return (size_t)4 * L.nx + (size_t)14 * L.ng + L.nnz_jac + L.nnz_hess + (size_t)(L.N + 1) * RIC_STRIDE;
  }
  int ensure(const Layout& L, int B);
  void release();
};

struct SolveArgs {
  Layout L; int B; landing_solver_opts o;
  const double* p; const double* x0;
  double* x_out; double* f_out; double* lam_out; int* status; int* iters; double* kkt;
  double* ws; size_t ws_stride;
  const int* tab; const int* stage_tab;
};

// ---- block-wide reductions through LDS (deterministic order) -----------------------------------------
enum { RSUM = 0, RMAX = 1, RMIN = 2 };
__device__ __forceinline__ double block_reduce(double v, int op, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  double r = red[0];
  for (unsigned t = 1; t < blockDim.x; ++t) {
    const double u = red[t];
    r = (op == RSUM) ? r + u : (op == RMAX ? fmax(r, u) : fmin(r, u));
  }
  __syncthreads();
  return r;
}

struct MemberMem {
  double *x, *xt, *dx, *gx;
  double *g, *gt, *s, *ds, *zL, *zU, *dzL, *dzU, *y, *yn, *lb, *ub, *sig, *rho;
  double *J, *H, *ric;
};

__device__ __forceinline__ MemberMem carve(const Layout& L, double* w) {
  MemberMem M;
  M.x = w; w += L.nx; M.xt = w; w += L.nx; M.dx = w; w += L.nx; M.gx = w; w += L.nx;
  M.g = w; w += L.ng; M.gt = w; w += L.ng; M.s = w; w += L.ng; M.ds = w; w += L.ng;
  M.zL = w; w += L.ng; M.zU = w; w += L.ng; M.dzL = w; w += L.ng; M.dzU = w; w += L.ng;
  M.y = w; w += L.ng; M.yn = w; w += L.ng; M.lb = w; w += L.ng; M.ub = w; w += L.ng;
  M.sig = w; w += L.ng; M.rho = w; w += L.ng;
  M.J = w; w += L.nnz_jac; M.H = w; w += L.nnz_hess; M.ric = w;
  return M;
}

// LDS of one member
struct Lds {
  double G[48 * GS];
  double P[24 * PS];
  double A1[NZ_TOT];          // stage nonzeros during assembly, then Y = P(:,0:12)*A^  (24 x YS = 888)
  double Li[24 * PS];         // inverse Cholesky factor
  double V[24 * PS];          // Li * G_us
  double Ah[12 * YS];
  double Sg[104], rh[104];
  double gam[48], pv[24], q[24], bv[12], vv[24], sig[24], w[48], sgn[24];
  double red[64];
  double filt_th[FILT_CAP], filt_ph[FILT_CAP];
};

// copy the CCS segments of stage k into the staging buffer
__device__ __forceinline__ void load_stage_nz(const Layout& L, const MemberMem& M, int k, double* nz, bool with_h) {
  const int N = L.N, lane = threadIdx.x, NT = blockDim.x;
  const bool last = (k == N - 1);
  for (int i = lane; i < 157; i += NT) nz[NZ_JX + i] = M.J[L.jx(k) + i];
  for (int i = lane, n = L.ju_len(k); i < n; i += NT) nz[NZ_JU + i] = M.J[L.ju(k) + i];
  if (!last) for (int i = lane, n = L.ju_len(k + 1); i < n; i += NT) nz[NZ_JUN + i] = M.J[L.ju(k + 1) + i];
  if (with_h) {
    for (int i = lane; i < 29; i += NT) nz[NZ_HX + i] = M.H[L.hx(k) + i];
    for (int i = lane, n = (k == 0 ? 148 : 160); i < n; i += NT) nz[NZ_HU + i] = M.H[L.hu(k) + i];
    if (!last) for (int i = lane; i < 160; i += NT) nz[NZ_HUN + i] = M.H[L.hu(k + 1) + i];
  }
}

// In-place Cholesky of the n x n block at A (row stride ld), lower triangle; on success the strict lower
// part holds L(i,j)*sqrt(d_j) un-normalised columns and rj[j] = 1/sqrt(d_j).  Returns false on a
// non-positive / non-finite pivot (wave-uniform).
__device__ __forceinline__ bool chol_lower(double* A, int ld, int n, double* rj) {
  const int lane = threadIdx.x, NT = blockDim.x;
  for (int j = 0; j < n; ++j) {
    const double d = A[j * ld + j];
    if (!(d > 0.0) || !(d < 1e300)) return false;
    const double inv = 1.0 / d;
    const int m = n - 1 - j;                   // trailing size
    for (int e = lane; e < m * m; e += NT) {   // (i,c) over the trailing square, lower part only
      const int i = j + 1 + e / m, c = j + 1 + e % m;
      if (c <= i) A[i * ld + c] -= A[i * ld + j] * A[c * ld + j] * inv;
    }
    if (lane == 0) rj[j] = 1.0 / sqrt(d);
    __syncthreads();
  }
  // normalise: L(i,j) = A(i,j) * rj[j], L(j,j) = sqrt(d_j) = 1/rj[j]
  for (int e = lane; e < n * n; e += NT) {
    const int i = e / n, j = e % n;
    if (j < i) A[i * ld + j] *= rj[j];
    else if (j == i) A[i * ld + j] = 1.0 / rj[j];
  }
  __syncthreads();
  return true;
}

// Li = L^{-1} (lower triangular), column j by lane j (uniform control flow, broadcast reads of L)
__device__ __forceinline__ void tri_inverse(const double* Lm, int ld, int n, double* Li) {
  const int j = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    if (j < n) {
      double acc = (i == j) ? 1.0 : 0.0;
      for (int t = 0; t < i; ++t) acc -= Lm[i * ld + t] * Li[t * PS + j];
      Li[i * PS + j] = (j <= i) ? acc / Lm[i * ld + i] : 0.0;
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(64) landing_ipm_kernel(SolveArgs A) {
  const int m = blockIdx.x;
  if (m >= A.B) return;
  const Layout& L = A.L;
  const int N = L.N, lane = threadIdx.x, NT = blockDim.x;
  const int nx = L.nx, ng = L.ng;
  const double* p = A.p + (size_t)m * L.np;
  const landing_solver_opts& o = A.o;
  const MemberMem M = carve(L, A.ws + (size_t)m * A.ws_stride);
  __shared__ Lds S;
  const double INF = INFINITY;

  // ------------------------------------------------------------------ initial point
  for (int i = lane; i < nx; i += NT) {
    double v = A.x0[(size_t)m * nx + i];
    if (i < 6) v = p[L.o_q_init + i]; else if (i < 12) v = p[L.o_qd_init + i - 6];   // X(:,1) is fixed (gen:90-91)
    M.x[i] = v;
  }
  for (int r = lane; r < ng; r += NT) { double lb, ub; bound_of(L, p, r, lb, ub); M.lb[r] = lb; M.ub[r] = ub; }
  __syncthreads();
  member_eval_g(L, M.x, p, M.g);
  __syncthreads();
  for (int r = lane; r < ng; r += NT) {
    const double lb = M.lb[r], ub = M.ub[r];
    double sv = 0.0, zl = 0.0, zu = 0.0;
    if (r >= 12 && lb != ub) {               // inequality row: slack pushed into the interior (IPOPT bound_push/frac)
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

  double mu = o.mu_init, delta_last = 0.0, th_max = 0.0;
  int nfilt = 0, it = 0, status = LANDING_MAX_ITER;
  double e_pr = 0, e_du = 0, e_co = 0;

  for (it = 0; it <= o.max_iter; ++it) {
    // ---------------------------------------------------------------- derivatives at (x, y)
    member_eval_jh(L, M.x, p, M.y, M.J, M.H, M.gx);
    __syncthreads();
    // ---------------------------------------------------------------- optimality error (unscaled)
    double du = 0.0, pr = 0.0, co = 0.0;
    for (int i = lane + 12; i < nx; i += NT) du = fmax(du, fabs(M.gx[i]));
    for (int r = lane + 12; r < ng; r += NT) {
      const double lb = M.lb[r], ub = M.ub[r], g = M.g[r];
      if (lb == ub) { pr = fmax(pr, fabs(g - lb)); continue; }
      const double s = M.s[r];
      pr = fmax(pr, fabs(g - s));
      if (lb > -INF) co = fmax(co, (s - lb) * M.zL[r]);
      if (ub < INF) co = fmax(co, (ub - s) * M.zU[r]);
    }
    du = block_reduce(du, RMAX, S.red); pr = block_reduce(pr, RMAX, S.red); co = block_reduce(co, RMAX, S.red);
    e_pr = pr; e_du = du; e_co = co;
    if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { status = LANDING_NUMERICAL; break; }
    if (fmax(du, fmax(pr, co)) <= o.tol) { status = LANDING_CONVERGED; break; }
    if (it == o.max_iter) break;
    // ---------------------------------------------------------------- barrier parameter (monotone)
    for (;;) {
      double cm = 0.0;
      for (int r = lane + 12; r < ng; r += NT) {
        const double lb = M.lb[r], ub = M.ub[r];
        if (lb == ub) continue;
        const double s = M.s[r];
        if (lb > -INF) cm = fmax(cm, fabs((s - lb) * M.zL[r] - mu));
        if (ub < INF) cm = fmax(cm, fabs((ub - s) * M.zU[r] - mu));
      }
      cm = block_reduce(cm, RMAX, S.red);
      if (fmax(du, fmax(pr, cm)) <= o.kappa_eps * mu && mu > o.tol / 10.0) {
        mu = fmax(o.tol / 10.0, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
        nfilt = 0;
      } else break;
    }
    const double tau = fmax(0.99, 1.0 - mu);
    // ---------------------------------------------------------------- Sigma, rho per inequality row
    for (int r = lane; r < ng; r += NT) {
      const double lb = M.lb[r], ub = M.ub[r];
      double sg = 0.0, rh = 0.0;
      if (r >= 12 && lb != ub) {
        const double s = M.s[r];
        if (lb > -INF) { const double d = s - lb; sg += M.zL[r] / d; rh -= mu / d; }
        if (ub < INF) { const double d = ub - s; sg += M.zU[r] / d; rh += mu / d; }
        rh += sg * (M.g[r] - s);
      }
      M.sig[r] = sg; M.rho[r] = rh;
    }
    __syncthreads();

    // ================================================================ Riccati factorisation with inertia correction
    double delta = 0.0;
    bool fact_ok = false;
    for (int attempt = 0; attempt < 60 && !fact_ok; ++attempt) {
      if (attempt > 0) {
        if (delta == 0.0) delta = (delta_last == 0.0) ? 1e-4 : fmax(1e-20, delta_last / 3.0);
        else delta *= (delta_last == 0.0 ? 100.0 : 8.0);
        if (delta > 1e40) break;
      }
      bool ok = true;
      // terminal cost-to-go on sigma_N = X_N: diagonal (terminal rows are copies of X_N, gen:94-97)
      for (int e = lane; e < 24 * PS; e += NT) S.P[e] = 0.0;
      __syncthreads();
      if (lane < 12) {
        const int i = lane;
        const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
        const double qn2 = 2.0 * p[L.o_QN + i];
        S.P[i * PS + i] = qn2 + M.sig[ra] + M.sig[rb] + delta;
        S.pv[i] = qn2 * (M.x[12 * N + i] - p[12 * N + i]) + M.rho[ra] + M.rho[rb];
        double* rec = M.ric + (size_t)N * RIC_STRIDE;      // record N: P_N (diag), p_N
        for (int j = 0; j < 24; ++j) rec[RIC_PX + i * 24 + j] = (j == i) ? S.P[i * PS + i] : 0.0;
        rec[RIC_PV + i] = S.pv[i];
      }
      __syncthreads();
      for (int k = N - 1; k >= 0 && ok; --k) {
        const bool last = (k == N - 1);
        const int nu = last ? 12 : 24, nsn = last ? 12 : 24, nw = 24 + nu;
        const int* tb = A.tab + A.stage_tab[k];
        const int g0 = L.g_stage(k), nr = L.rows(k);
        // ---- stage data into LDS
        load_stage_nz(L, M, k, S.A1, true);
        for (int r = lane; r < nr; r += NT) { S.Sg[r] = M.sig[g0 + r]; S.rh[r] = M.rho[g0 + r]; }
        for (int e = lane; e < 48 * GS; e += NT) S.G[e] = 0.0;
        for (int e = lane; e < 12 * YS; e += NT) S.Ah[e] = 0.0;
        __syncthreads();
        // ---- G = H + J^T Sigma J (sparse targets), gamma = J^T rho, A^, b
        {
          const int nT = tb[0];
          const int* ab = A.tab + tb[1]; const int* st = A.tab + tb[2]; const int* tm = A.tab + tb[3];
          for (int t = lane; t < nT; t += NT) {
            double acc = 0.0;
            for (int e = st[t]; e < st[t + 1]; ++e) {
              const int r = tm[3 * e], i1 = tm[3 * e + 1], i2 = tm[3 * e + 2];
              acc += (r < 0) ? S.A1[i1] : S.Sg[r] * S.A1[i1] * S.A1[i2];
            }
            const int a = ab[t] & 255, b = ab[t] >> 8;
            S.G[a * GS + b] = acc; S.G[b * GS + a] = acc;
          }
          const int* gs = A.tab + tb[4]; const int* gt = A.tab + tb[5];
          for (int a = lane; a < 48; a += NT) {
            double acc = 0.0;
            for (int e = gs[a]; e < gs[a + 1]; ++e) acc += S.rh[gt[2 * e]] * S.A1[gt[2 * e + 1]];
            S.gam[a] = acc;
          }
          const int nA = tb[6]; const int* at = A.tab + tb[7];
          for (int t = lane; t < nA; t += NT) S.Ah[at[3 * t + 1] * YS + at[3 * t + 2]] = -S.A1[at[3 * t]];
          if (lane < 12) S.bv[lane < 6 ? lane : (lane < 9 ? lane + 3 : lane - 3)] = -M.g[g0 + lane];
        }
        __syncthreads();
        for (int a = lane; a < nw; a += NT) S.G[a * GS + a] += delta;
        // ---- Y = P(:,0:12) A^  (nsn x 36), q = P(:,0:12) b + p   (A1 is free again: Y lives there)
        double* Y = S.A1;
        __syncthreads();
        for (int e = lane; e < nsn * 36; e += NT) {
          const int i = e / 36, j = e % 36;
          double acc = 0.0;
          for (int t = 0; t < 12; ++t) acc += S.P[i * PS + t] * S.Ah[t * YS + j];
          Y[i * YS + j] = acc;
        }
        for (int i = lane; i < nsn; i += NT) {
          double acc = S.pv[i];
          for (int t = 0; t < 12; ++t) acc += S.P[i * PS + t] * S.bv[t];
          S.q[i] = acc;
        }
        __syncthreads();
        // ---- G += T^T P T, gamma += T^T q
        for (int e = lane; e < 36 * 36; e += NT) {
          const int i = e / 36, j = e % 36;
          double acc = 0.0;
          for (int t = 0; t < 12; ++t) acc += S.Ah[t * YS + i] * Y[t * YS + j];
          S.G[i * GS + j] += acc;
        }
        if (!last) {
          for (int e = lane; e < 12 * 36; e += NT) {
            const int i = e / 36, j = e % 36;
            const double v = Y[(12 + i) * YS + j];
            S.G[(36 + i) * GS + j] += v; S.G[j * GS + 36 + i] += v;
          }
          for (int e = lane; e < 144; e += NT) { const int i = e / 12, j = e % 12; S.G[(36 + i) * GS + 36 + j] += S.P[(12 + i) * PS + 12 + j]; }
        }
        for (int j = lane; j < 36; j += NT) {
          double acc = 0.0;
          for (int t = 0; t < 12; ++t) acc += S.Ah[t * YS + j] * S.q[t];
          S.gam[j] += acc;
        }
        if (!last && lane < 12) S.gam[36 + lane] += S.q[12 + lane];
        __syncthreads();
        // ---- Cholesky of G_uu, inverse factor
        ok = chol_lower(S.G + 24 * GS + 24, GS, nu, S.sgn);
        if (!ok) break;
        tri_inverse(S.G + 24 * GS + 24, GS, nu, S.Li);
        // ---- V = Li G_us (nu x 24), vv = Li gamma_u
        for (int e = lane; e < nu * 24; e += NT) {
          const int i = e / 24, j = e % 24;
          double acc = 0.0;
          for (int t = 0; t <= i; ++t) acc += S.Li[i * PS + t] * S.G[(24 + t) * GS + j];
          S.V[i * PS + j] = acc;
        }
        for (int i = lane; i < nu; i += NT) {
          double acc = 0.0;
          for (int t = 0; t <= i; ++t) acc += S.Li[i * PS + t] * S.gam[24 + t];
          S.vv[i] = acc;
        }
        __syncthreads();
        // ---- P_k = G_ss - V^T V, p_k = gamma_s - V^T vv ; gains K = Li^T V, kappa = Li^T vv -> record k
        double* rec = M.ric + (size_t)k * RIC_STRIDE;
        for (int e = lane; e < 24 * 24; e += NT) {
          const int i = e / 24, j = e % 24;
          double acc = S.G[i * GS + j];
          for (int t = 0; t < nu; ++t) acc -= S.V[t * PS + i] * S.V[t * PS + j];
          S.P[i * PS + j] = acc;
          if (i < 12) rec[RIC_PX + i * 24 + j] = acc;
        }
        for (int i = lane; i < 24; i += NT) {
          double acc = S.gam[i];
          for (int t = 0; t < nu; ++t) acc -= S.V[t * PS + i] * S.vv[t];
          S.pv[i] = acc;
          if (i < 12) rec[RIC_PV + i] = acc;
        }
        for (int e = lane; e < nu * 24; e += NT) {
          const int i = e / 24, j = e % 24;
          double acc = 0.0;
          for (int t = i; t < nu; ++t) acc += S.Li[t * PS + i] * S.V[t * PS + j];
          rec[RIC_K + i * 24 + j] = acc;
        }
        for (int i = lane; i < nu; i += NT) {
          double acc = 0.0;
          for (int t = i; t < nu; ++t) acc += S.Li[t * PS + i] * S.vv[t];
          rec[RIC_KAP + i] = acc;
        }
        for (int e = lane; e < 12 * 36; e += NT) rec[RIC_AH + e] = S.Ah[(e / 36) * YS + e % 36];
        if (lane < 12) rec[RIC_B + lane] = S.bv[lane];
        __syncthreads();
      }
      if (ok) {
        // ---- stage 0: X_0 fixed, feet c_0 free: P_cc dc0 = -(p_c + P_cx dX0)
        for (int e = lane; e < 144; e += NT) { const int i = e / 12, j = e % 12; S.V[i * PS + j] = S.P[(12 + i) * PS + 12 + j]; }
        if (lane < 12) {
          const int i = lane;
          const double x0i = (i < 6) ? p[L.o_q_init + i] : p[L.o_qd_init + i - 6];
          S.sig[i] = x0i - M.x[i];
        }
        __syncthreads();
        ok = chol_lower(S.V, PS, 12, S.sgn);
        if (ok) {
          tri_inverse(S.V, PS, 12, S.Li);
          if (lane < 12) {
            double acc = S.pv[12 + lane];
            for (int t = 0; t < 12; ++t) acc += S.P[(12 + lane) * PS + t] * S.sig[t];
            S.q[lane] = acc;
          }
          __syncthreads();
          if (lane < 12) { double acc = 0.0; for (int t = 0; t <= lane; ++t) acc += S.Li[lane * PS + t] * S.q[t]; S.vv[lane] = acc; }
          __syncthreads();
          if (lane < 12) { double acc = 0.0; for (int t = lane; t < 12; ++t) acc += S.Li[t * PS + lane] * S.vv[t]; S.sig[12 + lane] = -acc; }
          __syncthreads();
        }
      }
      fact_ok = ok;
    }
    if (!fact_ok) { status = LANDING_NUMERICAL; break; }
    if (delta > 0.0) delta_last = delta;

    // ================================================================ forward pass: dx, ds (stage rows), y_dyn
    for (int k = 0; k < N; ++k) {
      const bool last = (k == N - 1);
      const int nu = last ? 12 : 24;
      const double* rec = M.ric + (size_t)k * RIC_STRIDE;
      const double* recn = M.ric + (size_t)(k + 1) * RIC_STRIDE;
      const int* tb = A.tab + A.stage_tab[k];
      const int g0 = L.g_stage(k), nr = L.rows(k);
      load_stage_nz(L, M, k, S.A1, false);
      if (lane < nu) {
        double acc = rec[RIC_KAP + lane];
        for (int t = 0; t < 24; ++t) acc += rec[RIC_K + lane * 24 + t] * S.sig[t];
        S.w[24 + lane] = -acc;
      }
      if (lane < 24) S.w[lane] = S.sig[lane];
      __syncthreads();
      if (lane < 12) { M.dx[L.x_X(k) + lane] = S.w[lane]; M.dx[L.x_U(k) + lane] = S.w[12 + lane]; M.dx[L.x_U(k) + 12 + lane] = S.w[24 + lane]; }
      {   // ds = J_d w + (g - s) for the stage's inequality rows
        const int* rs = A.tab + tb[8]; const int* rt = A.tab + tb[9];
        for (int r = lane + 12; r < nr; r += NT) {
          double acc = 0.0;
          for (int e = rs[r]; e < rs[r + 1]; ++e) acc += S.A1[rt[2 * e]] * S.w[rt[2 * e + 1]];
          M.ds[g0 + r] = acc + (M.g[g0 + r] - M.s[g0 + r]);
        }
      }
      // next state: X+ = A^ [sigma; f] + b ; c+ = u_c
      if (lane < 12) {
        double acc = rec[RIC_B + lane];
        for (int t = 0; t < 36; ++t) acc += rec[RIC_AH + lane * 36 + t] * S.w[t];
        S.q[lane] = acc;
      } else if (lane < 24) {
        S.q[lane] = last ? 0.0 : S.w[36 + (lane - 12)];
      }
      __syncthreads();
      if (lane < 24) S.sig[lane] = S.q[lane];
      __syncthreads();
      // multipliers of the dynamics rows: y = -(P_{k+1} sigma_{k+1} + p_{k+1})_X   (state order -> row order)
      if (lane < 12) {
        double acc = recn[RIC_PV + lane];
        const int nn = last ? 12 : 24;
        for (int t = 0; t < nn; ++t) acc += recn[RIC_PX + lane * 24 + t] * S.sig[t];
        const int q = lane < 6 ? lane : (lane < 9 ? lane + 3 : lane - 3);   // state index -> dyn row
        M.yn[g0 + q] = -acc;
      }
      __syncthreads();
    }
    if (lane < 12) {
      const int i = lane;
      M.dx[12 * N + i] = S.sig[i];
      const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
      M.ds[ra] = S.sig[i] + (M.g[ra] - M.s[ra]);
      M.ds[rb] = S.sig[i] + (M.g[rb] - M.s[rb]);
    }
    __syncthreads();

    // ================================================================ dual steps, step bounds, merit data
    double a_pr = 1.0, a_du = 1.0, th0 = 0.0, bar = 0.0, dphi = 0.0;
    for (int r = lane + 12; r < ng; r += NT) {
      const double lb = M.lb[r], ub = M.ub[r], g = M.g[r];
      if (lb == ub) { th0 += fabs(g - lb); continue; }
      const double s = M.s[r], ds = M.ds[r];
      th0 += fabs(g - s);
      double yn = M.sig[r] * ds;
      if (lb > -INF) {
        const double d = s - lb, zl = M.zL[r];
        const double dz = mu / d - zl - zl / d * ds;
        M.dzL[r] = dz; yn -= mu / d;
        if (ds < 0.0) a_pr = fmin(a_pr, -tau * d / ds);
        if (dz < 0.0) a_du = fmin(a_du, -tau * zl / dz);
        bar -= log(d); dphi -= mu * ds / d;
      } else M.dzL[r] = 0.0;
      if (ub < INF) {
        const double d = ub - s, zu = M.zU[r];
        const double dz = mu / d - zu + zu / d * ds;
        M.dzU[r] = dz; yn += mu / d;
        if (ds > 0.0) a_pr = fmin(a_pr, tau * d / ds);
        if (dz < 0.0) a_du = fmin(a_du, -tau * zu / dz);
        bar -= log(d); dphi += mu * ds / d;
      } else M.dzU[r] = 0.0;
      M.yn[r] = yn;
    }
    double f0 = 0.0;
    if (lane < 12) {
      const double d = M.x[12 * N + lane] - p[12 * N + lane], qn = p[L.o_QN + lane];
      f0 = qn * d * d; dphi += 2.0 * qn * d * M.dx[12 * N + lane];
    }
    a_pr = block_reduce(a_pr, RMIN, S.red); a_du = block_reduce(a_du, RMIN, S.red);
    th0 = block_reduce(th0, RSUM, S.red); bar = block_reduce(bar, RSUM, S.red);
    dphi = block_reduce(dphi, RSUM, S.red); f0 = block_reduce(f0, RSUM, S.red);
    const double ph0 = f0 + mu * bar;
    if (it == 0) th_max = 1e4 * fmax(1.0, th0);
    const double th_min = 1e-4;

    // ================================================================ filter line search
    double alpha = a_pr;
    bool accepted = false, armijo_step = false;
    while (alpha > 1e-10) {
      for (int i = lane; i < nx; i += NT) M.xt[i] = M.x[i] + alpha * M.dx[i];
      __syncthreads();
      member_eval_g(L, M.xt, p, M.gt);
      __syncthreads();
      double tht = 0.0, bt = 0.0, ft = 0.0;
      for (int r = lane + 12; r < ng; r += NT) {
        const double lb = M.lb[r], ub = M.ub[r], g = M.gt[r];
        if (lb == ub) { tht += fabs(g - lb); continue; }
        const double s = M.s[r] + alpha * M.ds[r];
        tht += fabs(g - s);
        if (lb > -INF) bt -= log(s - lb);
        if (ub < INF) bt -= log(ub - s);
      }
      if (lane < 12) { const double d = M.xt[12 * N + lane] - p[12 * N + lane]; ft = p[L.o_QN + lane] * d * d; }
      tht = block_reduce(tht, RSUM, S.red); bt = block_reduce(bt, RSUM, S.red); ft = block_reduce(ft, RSUM, S.red);
      const double pht = ft + mu * bt;
      bool ok_f = (tht <= th_max) && (pht < 1e300) && (pht > -1e300) && (tht < 1e300);
      for (int e = 0; e < nfilt && ok_f; ++e) if (tht >= S.filt_th[e] && pht >= S.filt_ph[e]) ok_f = false;
      const bool switching = (dphi < 0.0) && (th0 <= th_min) && (alpha * pow(-dphi, 2.3) > 1.0 * pow(th0, 1.1));
      if (ok_f) {
        if (switching) {
          if (pht <= ph0 + 1e-8 * alpha * dphi) { accepted = true; armijo_step = true; }
        } else if (tht <= (1.0 - 1e-5) * th0 || pht <= ph0 - 1e-8 * th0) {
          accepted = true;
        }
      }
      if (accepted) break;
      alpha *= 0.5;
    }
    if (!accepted) {
      // no acceptable step: take a short step along the Newton direction and restart the filter
      nfilt = 0;
      alpha = fmin(a_pr, 1e-2);
      for (int i = lane; i < nx; i += NT) M.xt[i] = M.x[i] + alpha * M.dx[i];
      __syncthreads();
      member_eval_g(L, M.xt, p, M.gt);
      __syncthreads();
    } else if (!armijo_step) {
      __syncthreads();
      if (nfilt == FILT_CAP) {     // drop the oldest entry (serial, rare)
        if (lane == 0) for (int e = 0; e + 1 < FILT_CAP; ++e) { S.filt_th[e] = S.filt_th[e + 1]; S.filt_ph[e] = S.filt_ph[e + 1]; }
        nfilt = FILT_CAP - 1;
      }
      if (lane == 0) { S.filt_th[nfilt] = (1.0 - 1e-5) * th0; S.filt_ph[nfilt] = ph0 - 1e-8 * th0; }
      nfilt++;
      __syncthreads();
    }
    // ================================================================ accept the trial point
    for (int i = lane; i < nx; i += NT) M.x[i] = M.xt[i];
    for (int r = lane; r < ng; r += NT) {
      const double lb = M.lb[r], ub = M.ub[r];
      M.g[r] = M.gt[r];
      if (r < 12) continue;
      if (lb == ub) { M.y[r] += alpha * (M.yn[r] - M.y[r]); continue; }
      const double s = M.s[r] + alpha * M.ds[r];
      double zl = 0.0, zu = 0.0;
      if (lb > -INF) { const double d = s - lb; zl = M.zL[r] + a_du * M.dzL[r]; zl = fmin(fmax(zl, mu / (1e10 * d)), 1e10 * mu / d); }
      if (ub < INF) { const double d = ub - s; zu = M.zU[r] + a_du * M.dzU[r]; zu = fmin(fmax(zu, mu / (1e10 * d)), 1e10 * mu / d); }
      M.s[r] = s; M.zL[r] = zl; M.zU[r] = zu; M.y[r] = zu - zl;
    }
    __syncthreads();
  }

  // -------------------------------------------------------------------- outputs
  // multipliers of the initial-state rows from stationarity of X(:,1): lam = -(grad f + J^T y)
  double fo = 0.0;
  if (lane < 12) { M.y[lane] = -M.gx[lane]; const double d = M.x[12 * N + lane] - p[12 * N + lane]; fo = p[L.o_QN + lane] * d * d; }
  fo = block_reduce(fo, RSUM, S.red);
  // reference-consistent KKT residual: max_viol(g), ||grad f + J^T lam||_inf, |lam * dist|
  double kp = 0.0, kc = 0.0;
  for (int r = lane; r < ng; r += NT) {
    const double lb = M.lb[r], ub = M.ub[r], g = M.g[r], lam = M.y[r];
    kp = fmax(kp, fmax(lb - g, fmax(g - ub, 0.0)));
    if (lb != ub) {
      const double dist = lam > 0.0 ? ub - g : g - lb;
      if (lam != 0.0 && dist < INF) kc = fmax(kc, fabs(lam * dist));
    }
  }
  kp = block_reduce(kp, RMAX, S.red); kc = block_reduce(kc, RMAX, S.red);
  for (int i = lane; i < nx; i += NT) A.x_out[(size_t)m * nx + i] = M.x[i];
  if (A.lam_out) for (int r = lane; r < ng; r += NT) A.lam_out[(size_t)m * ng + r] = M.y[r];
  if (lane == 0) {
    if (A.f_out) A.f_out[m] = fo;
    if (A.status) A.status[m] = status;
    if (A.iters) A.iters[m] = it;
    if (A.kkt) { A.kkt[3 * m] = kp; A.kkt[3 * m + 1] = e_du; A.kkt[3 * m + 2] = kc; }
  }
  (void)e_pr; (void)e_co;
}

}  // namespace landing
