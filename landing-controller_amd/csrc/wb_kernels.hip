// wb_kernels.hip -- SQP (Gauss-Newton / iLQR) loop on the full 18-DoF floating-base dynamics (gfx950, fp64).
// SURVEY section 8(f) row N2 / BASELINE configs[3]: "full spatial_v2 floating-base (18-DoF) dynamics linearisation in the SQP loop,
// N = 40, batch = 1024".  The reference holds the dynamics (casadi_compatible_dynamics.m:12-143 over spatial_v2 HandC.m) and hands
// them to CasADi / an NLP solver; it has no loop of its own around them.  The loop here is the standard multiple-shooting
// Gauss-Newton iteration for a trajectory-tracking problem on that model:
//   state x = [q; qd] (36), control u = joint torques (12; the six base coordinates are unactuated), known foot forces f_k (the
//   SRBM solution's), explicit Euler like the SRBM NLP (generate_landingCtrller_IPOPT.m:127-130):  q+ = q + dt qd,
//   qd+ = qd + dt qdd(q, qd, [0; u], f);  cost  sum_k 1/2 |x_k - xref_k|^2_Q + 1/2 |u_k|^2_R  +  1/2 |x_N - xref_N|^2_QN.
// One SQP iteration = (1) exact linearisation of the dynamics at every knot (rbd_kernels.hip, one thread per (knot, tangent)),
// (2) landing_wb_backward_kernel: the LQ subproblem by a Riccati recursion, one wavefront per member, the 36 x 36 value function and
// the stage matrices in LDS, (3) landing_wb_rollout_kernel: the nonlinear dynamics rolled out under the feedback policy for a set
// of step lengths, one thread per (step length, member); the host backtracks over them (landing-controller_amd/wb.py).
#include <hip/hip_runtime.h>
#include <math.h>

namespace landing {

constexpr int WB_NX = 36, WB_NU = 12, WB_LD = 37;     // LDS row stride 37: conflict-free column walks

struct WbBackArgs {
  int B, N; double dt, reg;
  const double* x; const double* u; const double* xref;          // [B][N+1][36], [B][N][12], [B][N+1][36]
  const double* A; const double* Hinv;                           // [B*N][18][36], [B*N][18][18] (knot index b*N + k)
  double Q[WB_NX], R[WB_NU], QN[WB_NX];                          // diagonal weights
  double* K; double* kff; double* dV; int* ok;                   // [B][N][12][36], [B][N][12], [B][2], [B]
};

// LQ backward pass of one member.  64 threads; every product is a plain LDS-resident loop (36^3 flops four times per knot).
__global__ void __launch_bounds__(64) landing_wb_backward_kernel(WbBackArgs a) {
  const int b = blockIdx.x, t = threadIdx.x, N = a.N;
  if (b >= a.B) return;
  __shared__ double V[WB_NX * WB_LD], Ak[WB_NX * WB_LD], VA[WB_NX * WB_LD], Bk[18 * WB_NU], VB[WB_NX * WB_NU];
  __shared__ double Quu[WB_NU * (WB_NU + 1)], Qux[WB_NU * WB_LD], Kk[WB_NU * WB_LD];
  __shared__ double v[WB_NX], Qx[WB_NX], Qu[WB_NU], kf[WB_NU], dx[WB_NX], acc[2];
  __shared__ int good;
  const double* xb = a.x + (size_t)b * (N + 1) * WB_NX;
  const double* rb = a.xref + (size_t)b * (N + 1) * WB_NX;
  for (int e = t; e < WB_NX * WB_NX; e += 64) { const int i = e / WB_NX, j = e % WB_NX; V[i * WB_LD + j] = (i == j) ? a.QN[i] : 0.0; }
  if (t < WB_NX) v[t] = a.QN[t] * (xb[(size_t)N * WB_NX + t] - rb[(size_t)N * WB_NX + t]);
  if (t == 0) { acc[0] = 0.0; acc[1] = 0.0; good = 1; }
  __syncthreads();
  for (int k = N - 1; k >= 0; --k) {
    const double* Ad = a.A + ((size_t)b * N + k) * 18 * 36;
    const double* Hi = a.Hinv + ((size_t)b * N + k) * 18 * 18;
    // A_k = [I, dt I; dt dqdd/dq, I + dt dqdd/dqd],  B_k = dt [0; Hinv(:, 6:18)] (only the lower 18 rows are stored)
    for (int e = t; e < WB_NX * WB_NX; e += 64) {
      const int i = e / WB_NX, j = e % WB_NX;
      double val;
      if (i < 18) val = (j == i ? 1.0 : 0.0) + (j == 18 + i ? a.dt : 0.0);
      else val = a.dt * Ad[(i - 18) * 36 + j] + (j == i ? 1.0 : 0.0);
      Ak[i * WB_LD + j] = val;
    }
    for (int e = t; e < 18 * WB_NU; e += 64) { const int i = e / WB_NU, c = e % WB_NU; Bk[e] = a.dt * Hi[i * 18 + 6 + c]; }
    if (t < WB_NX) dx[t] = xb[(size_t)k * WB_NX + t] - rb[(size_t)k * WB_NX + t];
    __syncthreads();
    // VA = V A, VB = V B
    for (int e = t; e < WB_NX * WB_NX; e += 64) {
      const int i = e / WB_NX, j = e % WB_NX;
      double s = 0.0;
      for (int m = 0; m < WB_NX; ++m) s += V[i * WB_LD + m] * Ak[m * WB_LD + j];
      VA[i * WB_LD + j] = s;
    }
    for (int e = t; e < WB_NX * WB_NU; e += 64) {
      const int i = e / WB_NU, c = e % WB_NU;
      double s = 0.0;
      for (int m = 0; m < 18; ++m) s += V[i * WB_LD + 18 + m] * Bk[m * WB_NU + c];
      VB[e] = s;
    }
    __syncthreads();
    // Qux = B' VA (12 x 36), Quu = R + B' VB + reg, Qu = R u + B' v, Qx = Q dx + A' v
    for (int e = t; e < WB_NU * WB_NX; e += 64) {
      const int c = e / WB_NX, j = e % WB_NX;
      double s = 0.0;
      for (int m = 0; m < 18; ++m) s += Bk[m * WB_NU + c] * VA[(18 + m) * WB_LD + j];
      Qux[c * WB_LD + j] = s;
    }
    for (int e = t; e < WB_NU * WB_NU; e += 64) {
      const int c = e / WB_NU, d = e % WB_NU;
      double s = (c == d) ? a.R[c] + a.reg : 0.0;
      for (int m = 0; m < 18; ++m) s += Bk[m * WB_NU + c] * VB[(18 + m) * WB_NU + d];
      Quu[c * (WB_NU + 1) + d] = s;
    }
    if (t < WB_NU) {
      double s = a.R[t] * a.u[((size_t)b * N + k) * WB_NU + t];
      for (int m = 0; m < 18; ++m) s += Bk[m * WB_NU + t] * v[18 + m];
      Qu[t] = s;
    }
    if (t < WB_NX) {
      double s = a.Q[t] * dx[t];
      for (int m = 0; m < WB_NX; ++m) s += Ak[m * WB_LD + t] * v[m];
      Qx[t] = s;
    }
    __syncthreads();
    // Cholesky of Quu (12 x 12, serial: 300 flops), then K = -Quu^-1 Qux and kff = -Quu^-1 Qu column by column
    if (t == 0) {
      for (int j = 0; j < WB_NU; ++j) {
        double d = Quu[j * (WB_NU + 1) + j];
        for (int m = 0; m < j; ++m) d -= Quu[j * (WB_NU + 1) + m] * Quu[j * (WB_NU + 1) + m];
        if (!(d > 0.0)) { good = 0; d = 1.0; }
        d = sqrt(d); Quu[j * (WB_NU + 1) + j] = d;
        for (int i = j + 1; i < WB_NU; ++i) {
          double s = Quu[i * (WB_NU + 1) + j];
          for (int m = 0; m < j; ++m) s -= Quu[i * (WB_NU + 1) + m] * Quu[j * (WB_NU + 1) + m];
          Quu[i * (WB_NU + 1) + j] = s / d;
        }
      }
    }
    __syncthreads();
    if (t <= WB_NX) {            // thread j < 36: column j of Qux; thread 36: the vector Qu
      double y[WB_NU];
      for (int i = 0; i < WB_NU; ++i) {
        double s = (t < WB_NX) ? Qux[i * WB_LD + t] : Qu[i];
        for (int m = 0; m < i; ++m) s -= Quu[i * (WB_NU + 1) + m] * y[m];
        y[i] = s / Quu[i * (WB_NU + 1) + i];
      }
      for (int i = WB_NU - 1; i >= 0; --i) {
        double s = y[i];
        for (int m = i + 1; m < WB_NU; ++m) s -= Quu[m * (WB_NU + 1) + i] * y[m];
        y[i] = s / Quu[i * (WB_NU + 1) + i];
      }
      for (int i = 0; i < WB_NU; ++i) { if (t < WB_NX) Kk[i * WB_LD + t] = -y[i]; else kf[i] = -y[i]; }
    }
    __syncthreads();
    // outputs of the stage; then V <- Q + A' VA + Qux' K (symmetrised), v <- Qx + Qux' kff
    for (int e = t; e < WB_NU * WB_NX; e += 64) a.K[(((size_t)b * N + k) * WB_NU + e / WB_NX) * WB_NX + e % WB_NX] = Kk[(e / WB_NX) * WB_LD + e % WB_NX];
    if (t < WB_NU) a.kff[((size_t)b * N + k) * WB_NU + t] = kf[t];
    if (t == 0) {
      double d1 = 0.0;
      for (int i = 0; i < WB_NU; ++i) d1 += kf[i] * Qu[i];
      acc[0] += d1; acc[1] -= 0.5 * d1;          // kff' Quu kff = -kff' Qu
    }
    // every thread owns the elements e = t + 64 n of the upper triangle (666 entries): computed into registers while V, VA and
    // A_k are still inputs of other threads, written after the barrier
    {
      double nv[11];
      int cnt = 0;
      for (int e = t; e < WB_NX * (WB_NX + 1) / 2; e += 64, ++cnt) {
        int i = 0, rem = e;
        while (rem >= WB_NX - i) { rem -= WB_NX - i; ++i; }
        const int j = i + rem;
        double s = (i == j) ? a.Q[i] : 0.0;
        for (int m = 0; m < WB_NX; ++m) s += Ak[m * WB_LD + i] * VA[m * WB_LD + j];
        for (int c = 0; c < WB_NU; ++c) s += 0.5 * (Qux[c * WB_LD + i] * Kk[c * WB_LD + j] + Qux[c * WB_LD + j] * Kk[c * WB_LD + i]);
        nv[cnt] = s;
      }
      double nvec = 0.0;
      if (t < WB_NX) { nvec = Qx[t]; for (int c = 0; c < WB_NU; ++c) nvec += Qux[c * WB_LD + t] * kf[c]; }
      __syncthreads();
      cnt = 0;
      for (int e = t; e < WB_NX * (WB_NX + 1) / 2; e += 64, ++cnt) {
        int i = 0, rem = e;
        while (rem >= WB_NX - i) { rem -= WB_NX - i; ++i; }
        const int j = i + rem;
        V[i * WB_LD + j] = nv[cnt]; V[j * WB_LD + i] = nv[cnt];
      }
      if (t < WB_NX) v[t] = nvec;
    }
    __syncthreads();
  }
  if (t == 0) { a.dV[2 * b] = acc[0]; a.dV[2 * b + 1] = acc[1]; a.ok[b] = good; }
}

struct WbRollArgs {
  const RbdModel* model; int B, N, nalpha; double dt;
  const double* alphas;                                          // [nalpha]
  const double* x; const double* u; const double* xref; const double* f_foot;   // current trajectory, reference, foot forces [B][N][12] or null
  const double* K; const double* kff;                            // null: open-loop rollout of u (initialisation)
  double Q[WB_NX], R[WB_NU], QN[WB_NX];
  double* xnew; double* unew; double* cost;                      // [nalpha][B][N+1][36], [nalpha][B][N][12], [nalpha][B]
};

// nonlinear rollout under u = u_k + alpha kff_k + K_k (x - x_k), one thread per (step length, member)
__global__ void __launch_bounds__(64) landing_wb_rollout_kernel(WbRollArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.nalpha * a.B) return;
  const int ia = idx / a.B, b = idx % a.B, N = a.N;
  const RbdModel& M = *a.model;
  const double alpha = a.alphas[ia];
  const double* xb = a.x + (size_t)b * (N + 1) * WB_NX; const double* ub = a.u + (size_t)b * N * WB_NU;
  const double* rb = a.xref + (size_t)b * (N + 1) * WB_NX;
  double* xo = a.xnew + ((size_t)ia * a.B + b) * (N + 1) * WB_NX; double* uo = a.unew + ((size_t)ia * a.B + b) * N * WB_NU;
  double xs[WB_NX], un[WB_NU], H[RB_NB * RB_NB], C[RB_NB], rhs[RB_NB];
  for (int i = 0; i < WB_NX; ++i) { xs[i] = xb[i]; xo[i] = xs[i]; }
  double cost = 0.0; bool ok = true;
  for (int k = 0; k < N; ++k) {
    for (int c = 0; c < WB_NU; ++c) {
      double s = ub[k * WB_NU + c];
      if (a.K) {
        s += alpha * a.kff[((size_t)b * N + k) * WB_NU + c];
        const double* Kr = a.K + (((size_t)b * N + k) * WB_NU + c) * WB_NX;
        for (int j = 0; j < WB_NX; ++j) s += Kr[j] * (xs[j] - xb[k * WB_NX + j]);
      }
      un[c] = s; uo[k * WB_NU + c] = s;
      cost += 0.5 * a.R[c] * s * s;
    }
    for (int i = 0; i < WB_NX; ++i) { const double d = xs[i] - rb[k * WB_NX + i]; cost += 0.5 * a.Q[i] * d * d; }
    hand_c(M, xs, xs + 18, a.f_foot ? a.f_foot + ((size_t)b * N + k) * 12 : nullptr, H, C);
    for (int i = 0; i < RB_NB; ++i) rhs[i] = (i >= 6 ? un[i - 6] : 0.0) - C[i];
    ok = chol_solve18(H, rhs) && ok;
    for (int i = 0; i < 18; ++i) { const double qd = xs[18 + i]; xs[i] += a.dt * qd; xs[18 + i] = qd + a.dt * rhs[i]; }
    for (int i = 0; i < WB_NX; ++i) xo[(k + 1) * WB_NX + i] = xs[i];
  }
  for (int i = 0; i < WB_NX; ++i) { const double d = xs[i] - rb[N * WB_NX + i]; cost += 0.5 * a.QN[i] * d * d; }
  a.cost[(size_t)ia * a.B + b] = (ok && cost == cost) ? cost : INFINITY;
}

}  // namespace landing
