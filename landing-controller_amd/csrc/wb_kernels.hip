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
  int B, N; double dt, reg; int semi;      // semi = 1: semi-implicit (symplectic) Euler, qd+ = qd + dt qdd, q+ = q + dt qd+ (landing_wb_set_integrator)
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
    // A_k = [I, dt I; dt dqdd/dq, I + dt dqdd/dqd],  B_k = dt [0; Hinv(:, 6:18)] (only the lower 18 rows Bl are stored).  Semi-implicit Euler:
    // the q rows of both are [I 0] + dt x (their qd rows), i.e. B = [dt Bl; Bl] and B' w = Bl' (w_qd + dt w_q) -- the factor sd below
    for (int e = t; e < WB_NX * WB_NX; e += 64) {
      const int i = e / WB_NX, j = e % WB_NX;
      double val;
      if (i < 18) val = (j == i ? 1.0 : 0.0) + (a.semi ? a.dt * (a.dt * Ad[i * 36 + j] + (j == 18 + i ? 1.0 : 0.0)) : (j == 18 + i ? a.dt : 0.0));      // semi: q rows = [I 0] + dt (qd rows)
      else val = a.dt * Ad[(i - 18) * 36 + j] + (j == i ? 1.0 : 0.0);
      Ak[i * WB_LD + j] = val;
    }
    for (int e = t; e < 18 * WB_NU; e += 64) { const int i = e / WB_NU, c = e % WB_NU; Bk[e] = a.dt * Hi[i * 18 + 6 + c]; }
    if (t < WB_NX) dx[t] = xb[(size_t)k * WB_NX + t] - rb[(size_t)k * WB_NX + t];
    __syncthreads();
    const double sd = a.semi ? a.dt : 0.0;
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
      for (int m = 0; m < 18; ++m) s += (V[i * WB_LD + 18 + m] + sd * V[i * WB_LD + m]) * Bk[m * WB_NU + c];
      VB[e] = s;
    }
    __syncthreads();
    // Qux = B' VA (12 x 36), Quu = R + B' VB + reg, Qu = R u + B' v, Qx = Q dx + A' v
    for (int e = t; e < WB_NU * WB_NX; e += 64) {
      const int c = e / WB_NX, j = e % WB_NX;
      double s = 0.0;
      for (int m = 0; m < 18; ++m) s += Bk[m * WB_NU + c] * (VA[(18 + m) * WB_LD + j] + sd * VA[m * WB_LD + j]);
      Qux[c * WB_LD + j] = s;
    }
    for (int e = t; e < WB_NU * WB_NU; e += 64) {
      const int c = e / WB_NU, d = e % WB_NU;
      double s = (c == d) ? a.R[c] + a.reg : 0.0;
      for (int m = 0; m < 18; ++m) s += Bk[m * WB_NU + c] * (VB[(18 + m) * WB_NU + d] + sd * VB[m * WB_NU + d]);
      Quu[c * (WB_NU + 1) + d] = s;
    }
    if (t < WB_NU) {
      double s = a.R[t] * a.u[((size_t)b * N + k) * WB_NU + t];
      for (int m = 0; m < 18; ++m) s += Bk[m * WB_NU + t] * (v[18 + m] + sd * v[m]);
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
  const RbdModel* model; int B, N, nalpha; double dt; int semi; int arrow;      // semi: semi-implicit Euler (WbBackArgs); arrow: the model's H is block-arrow (base 6 + four 3-joint legs on the base): structured solve
  const double* alphas;                                          // [nalpha]
  const double* x; const double* u; const double* xref; const double* f_foot;   // current trajectory, reference, foot forces [B][N][12] or null
  const double* K; const double* kff;                            // null: open-loop rollout of u (initialisation)
  double Q[WB_NX], R[WB_NU], QN[WB_NX];
  double* xnew; double* unew; double* cost;                      // [nalpha][B][N+1][36], [nalpha][B][N][12], [nalpha][B]
  const double* skip;                                            // [B] or null: members with skip[b] != 0 (a step length already taken, landing_wb_select) are not rolled out: cost = inf
};

// nonlinear rollout under u = u_k + alpha kff_k + K_k (x - x_k), one thread per (step length, member)
__global__ void __launch_bounds__(64) landing_wb_rollout_kernel(WbRollArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.nalpha * a.B) return;
  const int ia = idx / a.B, b = idx % a.B, N = a.N;
  if (a.skip && a.skip[b] != 0.0) { a.cost[(size_t)ia * a.B + b] = INFINITY; return; }
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
    for (int i = 0; i < 18; ++i) { const double qd = xs[18 + i], qn = qd + a.dt * rhs[i]; xs[i] += a.dt * (a.semi ? qn : qd); xs[18 + i] = qn; }
    for (int i = 0; i < WB_NX; ++i) xo[(k + 1) * WB_NX + i] = xs[i];
  }
  for (int i = 0; i < WB_NX; ++i) { const double d = xs[i] - rb[N * WB_NX + i]; cost += 0.5 * a.QN[i] * d * d; }
  a.cost[(size_t)ia * a.B + b] = (ok && cost == cost) ? cost : INFINITY;
}

// ---- the same rollout with every per-thread array in LDS -------------------------------------------------------------------------
// The kernel above keeps E[18][9], r[18][3], v, a, f (18 spatial vectors each), the composite inertias, H (18 x 18) and the state in private
// memory (8.5 KB per thread): the 18-body recursion indexes them through the model's parent array, so they live in scratch and one forward-dynamics
// evaluation costs ~325 us of dependent scratch round trips (13 ms per rollout of 40 knots).  Here the arrays of WB_TPB
// threads sit in LDS, element-major ([element][thread]: the WB_TPB lanes of a wave touch consecutive banks), one workgroup = one wave with WB_TPB
// active lanes.  Same arithmetic in the same order as hand_c / chol_solve18 / landing_wb_rollout_kernel: results are identical bit for bit.
constexpr int WB_TPB = 16;
constexpr int WS_E = 0, WS_R = 162, WS_V = 216, WS_F = 324, WS_U = 432, WS_H = 612, WS_C = 936, WS_RHS = 954, WS_X = 972, WS_UN = 1008, WS_PER = 1020;
static_assert(WB_TPB * WS_PER * 8 + 4096 <= 160 * 1024, "LDS of one workgroup");
struct WsView {
  double* base;
  __device__ __forceinline__ double& operator()(int i) const { return base[i * WB_TPB]; }
};
__device__ __forceinline__ SV ws_ld_sv(const WsView& W, int off, int i) { SV s; s.a = mk3(W(off + 6 * i), W(off + 6 * i + 1), W(off + 6 * i + 2)); s.l = mk3(W(off + 6 * i + 3), W(off + 6 * i + 4), W(off + 6 * i + 5)); return s; }
__device__ __forceinline__ void ws_st_sv(const WsView& W, int off, int i, SV s) { W(off + 6 * i) = s.a.x; W(off + 6 * i + 1) = s.a.y; W(off + 6 * i + 2) = s.a.z; W(off + 6 * i + 3) = s.l.x; W(off + 6 * i + 4) = s.l.y; W(off + 6 * i + 5) = s.l.z; }
__device__ __forceinline__ void ws_ld_xf(const WsView& W, int i, double* E, double* r) {
#pragma unroll
  for (int j = 0; j < 9; ++j) E[j] = W(WS_E + 9 * i + j);
#pragma unroll
  for (int j = 0; j < 3; ++j) r[j] = W(WS_R + 3 * i + j);
}
// H -> W(WS_H + 18 i + j), C -> W(WS_C + i); q = W(WS_X + i), qd = W(WS_X + 18 + i)   (hand_c above, line for line)
__device__ void hand_c_lds(const RbdModel& M, const double* f_foot, const WsView& W) {
  double Ei[9], ri[3];
  for (int i = 0; i < RB_NB; ++i) {
    joint_xform(M.jtype[i], W(WS_X + i), M.E[i], M.r[i], Ei, ri);
#pragma unroll
    for (int j = 0; j < 9; ++j) W(WS_E + 9 * i + j) = Ei[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) W(WS_R + 3 * i + j) = ri[j];
    const SV vJ = sunit(M.jtype[i], W(WS_X + 18 + i));
    const int pa = M.parent[i];
    SV vi, ai;
    if (pa == 0) {
      SV g; g.a = mk3(0, 0, 0); g.l = mk3(0, 0, 9.81);
      vi = vJ; ai = xmotion(Ei, ri, g);
    } else {
      const SV vp = xmotion(Ei, ri, ws_ld_sv(W, WS_V, pa - 1));
      vi.a = add3(vp.a, vJ.a); vi.l = add3(vp.l, vJ.l);
      const SV ap = xmotion(Ei, ri, ws_ld_sv(W, WS_U, pa - 1)), cv = crm_mul(vi, vJ);
      ai.a = add3(ap.a, cv.a); ai.l = add3(ap.l, cv.l);
    }
    ws_st_sv(W, WS_V, i, vi); ws_st_sv(W, WS_U, i, ai);
    const SV Ia = inertia_mul(M.m[i], M.h[i], M.I[i], ai), Iv = inertia_mul(M.m[i], M.h[i], M.I[i], vi), cf = crf_mul(vi, Iv);
    SV fi; fi.a = add3(Ia.a, cf.a); fi.l = add3(Ia.l, cf.l);
    ws_st_sv(W, WS_F, i, fi);
  }
  if (f_foot) {
    double E0[9], r0[3];
    for (int j = 0; j < 9; ++j) E0[j] = (j % 4 == 0) ? 1.0 : 0.0;
    r0[0] = r0[1] = r0[2] = 0.0;
    auto compose = [](const double* Eu, const double* ru, double* Ea, double* ra) {
      const V3d t = mulT3(Ea, mk3(ru[0], ru[1], ru[2]));
      double En[9];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) En[3 * a + b] = Eu[3 * a] * Ea[b] + Eu[3 * a + 1] * Ea[3 + b] + Eu[3 * a + 2] * Ea[6 + b];
      for (int j = 0; j < 9; ++j) Ea[j] = En[j];
      ra[0] += t.x; ra[1] += t.y; ra[2] += t.z;
    };
    for (int i = 0; i < 6; ++i) { ws_ld_xf(W, i, Ei, ri); compose(Ei, ri, E0, r0); }
    for (int leg = 0; leg < 4; ++leg) {
      double El[9], rl[3];
      for (int j = 0; j < 9; ++j) El[j] = E0[j];
      for (int j = 0; j < 3; ++j) rl[j] = r0[j];
      const int jb = M.b_foot[leg] - 1;
      for (int i = jb - 2; i <= jb; ++i) { ws_ld_xf(W, i, Ei, ri); compose(Ei, ri, El, rl); }
      const V3d pf = add3(mk3(rl[0], rl[1], rl[2]), mulT3(El, mk3(M.foot_r[leg][0], M.foot_r[leg][1], M.foot_r[leg][2])));
      const V3d fw = mk3(f_foot[3 * leg], f_foot[3 * leg + 1], f_foot[3 * leg + 2]);
      const V3d nb = crs3(sub3(pf, mk3(rl[0], rl[1], rl[2])), fw);
      SV fj = ws_ld_sv(W, WS_F, jb);
      fj.a = sub3(fj.a, mul3(El, nb)); fj.l = sub3(fj.l, mul3(El, fw));
      ws_st_sv(W, WS_F, jb, fj);
    }
  }
  for (int i = RB_NB - 1; i >= 0; --i) {
    const SV fi = ws_ld_sv(W, WS_F, i);
    W(WS_C + i) = sdot(M.jtype[i], fi);
    const int pa = M.parent[i];
    if (pa != 0) {
      ws_ld_xf(W, i, Ei, ri);
      const SV t = xforceT(Ei, ri, fi);
      SV fp = ws_ld_sv(W, WS_F, pa - 1);
      fp.a = add3(fp.a, t.a); fp.l = add3(fp.l, t.l);
      ws_st_sv(W, WS_F, pa - 1, fp);
    }
  }
  // composite inertias in the region of the (dead) accelerations: cm at WS_U + i, ch at WS_U + 18 + 3 i, cI at WS_U + 72 + 6 i
  constexpr int CM = WS_U, CH = WS_U + 18, CI = WS_U + 72;
  for (int i = 0; i < RB_NB; ++i) { W(CM + i) = M.m[i]; for (int j = 0; j < 3; ++j) W(CH + 3 * i + j) = M.h[i][j]; for (int j = 0; j < 6; ++j) W(CI + 6 * i + j) = M.I[i][j]; }
  for (int i = RB_NB - 1; i >= 0; --i) {
    const int pa = M.parent[i];
    if (pa == 0) continue;
    ws_ld_xf(W, i, Ei, ri);
    const double cmi = W(CM + i);
    const V3d hp = mulT3(Ei, mk3(W(CH + 3 * i), W(CH + 3 * i + 1), W(CH + 3 * i + 2))), rr = mk3(ri[0], ri[1], ri[2]);
    const V3d hn = add3(hp, scl3(cmi, rr));
    double T[9];
    { double I6[6];
      for (int j = 0; j < 6; ++j) I6[j] = W(CI + 6 * i + j);
      const double Is[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
      double A[9];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[3 * a + b] = Is[3 * a] * Ei[b] + Is[3 * a + 1] * Ei[3 + b] + Is[3 * a + 2] * Ei[6 + b];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) T[3 * a + b] = Ei[a] * A[b] + Ei[3 + a] * A[3 + b] + Ei[6 + a] * A[6 + b];
    }
    auto add_ss = [&](V3d a, V3d b, double sgn) {
      const double d = a.x * b.x + a.y * b.y + a.z * b.z;
      const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
      for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) T[3 * x + y] += sgn * (bv[x] * av[y] - (x == y ? d : 0.0));
    };
    add_ss(rr, hp, -1.0); add_ss(hn, rr, -1.0);
    const int ip = CI + 6 * (pa - 1);
    W(ip) += T[0]; W(ip + 1) += 0.5 * (T[1] + T[3]); W(ip + 2) += 0.5 * (T[2] + T[6]); W(ip + 3) += T[4]; W(ip + 4) += 0.5 * (T[5] + T[7]); W(ip + 5) += T[8];
    W(CM + pa - 1) += cmi; W(CH + 3 * (pa - 1)) += hn.x; W(CH + 3 * (pa - 1) + 1) += hn.y; W(CH + 3 * (pa - 1) + 2) += hn.z;
  }
  for (int i = 0; i < RB_NB * RB_NB; ++i) W(WS_H + i) = 0.0;
  for (int i = 0; i < RB_NB; ++i) {
    double hh[3], II[6];
    for (int j = 0; j < 3; ++j) hh[j] = W(CH + 3 * i + j);
    for (int j = 0; j < 6; ++j) II[j] = W(CI + 6 * i + j);
    SV fh = inertia_mul(W(CM + i), hh, II, sunit(M.jtype[i], 1.0));
    W(WS_H + i * RB_NB + i) = sdot(M.jtype[i], fh);
    int j = i;
    while (M.parent[j] > 0) {
      ws_ld_xf(W, j, Ei, ri);
      fh = xforceT(Ei, ri, fh);
      j = M.parent[j] - 1;
      const double hij = sdot(M.jtype[j], fh);
      W(WS_H + i * RB_NB + j) = hij; W(WS_H + j * RB_NB + i) = hij;
    }
  }
}
// chol_solve18 on W(WS_H ..), W(WS_RHS ..)
__device__ bool chol_solve18_lds(const WsView& W) {
  for (int j = 0; j < RB_NB; ++j) {
    double d = W(WS_H + j * RB_NB + j);
    for (int k = 0; k < j; ++k) { const double l = W(WS_H + j * RB_NB + k); d -= l * l; }
    if (!(d > 0.0)) return false;
    d = sqrt(d); W(WS_H + j * RB_NB + j) = d;
    for (int i = j + 1; i < RB_NB; ++i) {
      double s = W(WS_H + i * RB_NB + j);
      for (int k = 0; k < j; ++k) s -= W(WS_H + i * RB_NB + k) * W(WS_H + j * RB_NB + k);
      W(WS_H + i * RB_NB + j) = s / d;
    }
  }
  for (int i = 0; i < RB_NB; ++i) { double s = W(WS_RHS + i); for (int k = 0; k < i; ++k) s -= W(WS_H + i * RB_NB + k) * W(WS_RHS + k); W(WS_RHS + i) = s / W(WS_H + i * RB_NB + i); }
  for (int i = RB_NB - 1; i >= 0; --i) { double s = W(WS_RHS + i); for (int k = i + 1; k < RB_NB; ++k) s -= W(WS_H + k * RB_NB + i) * W(WS_RHS + k); W(WS_RHS + i) = s / W(WS_H + i * RB_NB + i); }
  return true;
}
// The joint-space inertia of the quadruped is block-arrow: the six base coordinates couple with everything, the 3 x 3 blocks of the four
// legs only with the base and themselves (different branches of the tree).  Solve H x = rhs by eliminating the legs first -- the Cholesky
// factorisation in the order legs, base has no fill-in: per leg a 3 x 3 factor, Y = A^-1 [B | r] (3 x 7), Schur update of the 6 x 6 base
// block; then the base; then x_leg = y - Y x_base.  Every loop has constant bounds (registers, pipelined LDS loads): ~1000 flops against
// the 2700 of the dense factorisation with its ~2000 dependent LDS round trips, which was 80 % of a forward-dynamics evaluation.
// x -> W(WS_RHS ..); false = a pivot was not positive.
__device__ bool arrow_solve18_lds(const WsView& W) {
  double S[6][6], rb[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { rb[i] = W(WS_RHS + i);
#pragma unroll
    for (int j = 0; j < 6; ++j) S[i][j] = W(WS_H + i * RB_NB + j); }
  bool ok = true;
  double Y[4][3][7];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int o = 6 + 3 * l;
    double A[3][3], Bm[3][7];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int b = 0; b < 3; ++b) A[a][b] = W(WS_H + (o + a) * RB_NB + o + b);
#pragma unroll
      for (int b = 0; b < 6; ++b) Bm[a][b] = W(WS_H + (o + a) * RB_NB + b);
      Bm[a][6] = W(WS_RHS + o + a);
    }
    // A = L L'
    const double d0 = A[0][0]; ok = ok && (d0 > 0.0); const double l00 = sqrt(d0 > 0.0 ? d0 : 1.0);
    const double l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    const double d1 = A[1][1] - l10 * l10; ok = ok && (d1 > 0.0); const double l11 = sqrt(d1 > 0.0 ? d1 : 1.0);
    const double l21 = (A[2][1] - l20 * l10) / l11;
    const double d2 = A[2][2] - l20 * l20 - l21 * l21; ok = ok && (d2 > 0.0); const double l22 = sqrt(d2 > 0.0 ? d2 : 1.0);
#pragma unroll
    for (int cix = 0; cix < 7; ++cix) {
      const double z0 = Bm[0][cix] / l00, z1 = (Bm[1][cix] - l10 * z0) / l11, z2 = (Bm[2][cix] - l20 * z0 - l21 * z1) / l22;
      const double y2 = z2 / l22, y1 = (z1 - l21 * y2) / l11, y0 = (z0 - l10 * y1 - l20 * y2) / l00;
      Y[l][0][cix] = y0; Y[l][1][cix] = y1; Y[l][2][cix] = y2;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j) S[i][j] -= Bm[0][i] * Y[l][0][j] + Bm[1][i] * Y[l][1][j] + Bm[2][i] * Y[l][2][j];
      rb[i] -= Bm[0][i] * Y[l][0][6] + Bm[1][i] * Y[l][1][6] + Bm[2][i] * Y[l][2][6];
    }
  }
  // base block: dense 6 x 6 Cholesky in registers
  double Lb[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = S[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= Lb[j][k] * Lb[j][k];
    ok = ok && (d > 0.0);
    d = sqrt(d > 0.0 ? d : 1.0); Lb[j][j] = d;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = 0.5 * (S[i][j] + S[j][i]);
#pragma unroll
      for (int k = 0; k < j; ++k) s -= Lb[i][k] * Lb[j][k];
      Lb[i][j] = s / d;
    }
  }
  double xb[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { double s = rb[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= Lb[i][k] * xb[k];
    xb[i] = s / Lb[i][i]; }
#pragma unroll
  for (int i = 5; i >= 0; --i) { double s = xb[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= Lb[k][i] * xb[k];
    xb[i] = s / Lb[i][i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) W(WS_RHS + i) = xb[i];
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double s = Y[l][a][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) s -= Y[l][a][j] * xb[j];
      W(WS_RHS + 6 + 3 * l + a) = s;
    }
  return ok;
}

__global__ void __launch_bounds__(64) landing_wb_rollout_lds_kernel(WbRollArgs a) {
  __shared__ double ws[WB_TPB * WS_PER];
  __shared__ RbdModel Ms;                       // the model (3.5 KB) next to the arrays: the recursion reads parent / jtype / Xtree / inertia of every body
  {
    static_assert(sizeof(RbdModel) % 4 == 0, "word copy");
    const unsigned* src = reinterpret_cast<const unsigned*>(a.model); unsigned* dst = reinterpret_cast<unsigned*>(&Ms);
    for (int e = threadIdx.x; e < (int)(sizeof(RbdModel) / 4); e += blockDim.x) dst[e] = src[e];
  }
  __syncthreads();
  const int idx = blockIdx.x * WB_TPB + threadIdx.x;
  if ((int)threadIdx.x >= WB_TPB || idx >= a.nalpha * a.B) return;
  const WsView W{ws + threadIdx.x};
  const int ia = idx / a.B, b = idx % a.B, N = a.N;
  if (a.skip && a.skip[b] != 0.0) { a.cost[(size_t)ia * a.B + b] = INFINITY; return; }
  const RbdModel& M = Ms;
  const double alpha = a.alphas[ia];
  const double* xb = a.x + (size_t)b * (N + 1) * WB_NX; const double* ub = a.u + (size_t)b * N * WB_NU;
  const double* rb = a.xref + (size_t)b * (N + 1) * WB_NX;
  double* xo = a.xnew + ((size_t)ia * a.B + b) * (N + 1) * WB_NX; double* uo = a.unew + ((size_t)ia * a.B + b) * N * WB_NU;
  for (int i = 0; i < WB_NX; ++i) { const double v = xb[i]; W(WS_X + i) = v; xo[i] = v; }
  double cost = 0.0; bool ok = true;
  for (int k = 0; k < N; ++k) {
    for (int c = 0; c < WB_NU; ++c) {
      double s = ub[k * WB_NU + c];
      if (a.K) {
        s += alpha * a.kff[((size_t)b * N + k) * WB_NU + c];
        const double* Kr = a.K + (((size_t)b * N + k) * WB_NU + c) * WB_NX;
        for (int j = 0; j < WB_NX; ++j) s += Kr[j] * (W(WS_X + j) - xb[k * WB_NX + j]);
      }
      W(WS_UN + c) = s; uo[k * WB_NU + c] = s;
      cost += 0.5 * a.R[c] * s * s;
    }
    for (int i = 0; i < WB_NX; ++i) { const double d = W(WS_X + i) - rb[k * WB_NX + i]; cost += 0.5 * a.Q[i] * d * d; }
    hand_c_lds(M, a.f_foot ? a.f_foot + ((size_t)b * N + k) * 12 : nullptr, W);
    for (int i = 0; i < RB_NB; ++i) W(WS_RHS + i) = (i >= 6 ? W(WS_UN + i - 6) : 0.0) - W(WS_C + i);
    ok = (a.arrow ? arrow_solve18_lds(W) : chol_solve18_lds(W)) && ok;
    for (int i = 0; i < 18; ++i) { const double qd = W(WS_X + 18 + i), qn = qd + a.dt * W(WS_RHS + i); W(WS_X + i) += a.dt * (a.semi ? qn : qd); W(WS_X + 18 + i) = qn; }
    for (int i = 0; i < WB_NX; ++i) xo[(k + 1) * WB_NX + i] = W(WS_X + i);
  }
  for (int i = 0; i < WB_NX; ++i) { const double d = W(WS_X + i) - rb[N * WB_NX + i]; cost += 0.5 * a.QN[i] * d * d; }
  a.cost[(size_t)ia * a.B + b] = (ok && cost == cost) ? cost : INFINITY;
}


// Step-length selection on the device: member b keeps the FIRST of the nalpha rollouts (in the order of the list) whose cost is below its
// current one -- trajectory, controls, cost and the step length taken (0 = none) -- unless its backward pass failed (ok = 0).  Replaces one
// rollout launch + torch.where merges per step length of the host loop.
struct WbSelArgs { int B, N, nalpha, keep; const double* alphas; const int* ok; const double* xnew; const double* unew; const double* costnew; double* x; double* u; double* cost; double* step; };
__global__ void __launch_bounds__(256) landing_wb_select_kernel(WbSelArgs a) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (b >= a.B) return;
  int pick = -1;
  const double c0 = a.cost[b];
  const bool taken = a.keep && a.step && a.step[b] != 0.0;      // second stage of a two-stage search: the member already took a step length
  if (a.ok[b] && !taken) for (int ia = 0; ia < a.nalpha && pick < 0; ++ia) if (a.costnew[(size_t)ia * a.B + b] < c0) pick = ia;      // (uniform: every thread reads the same words)
  __syncthreads();      // every thread has read cost[b] before thread 0 overwrites it
  if (pick >= 0) {
    const size_t nxs = (size_t)(a.N + 1) * WB_NX, nus = (size_t)a.N * WB_NU;
    const double* xs = a.xnew + ((size_t)pick * a.B + b) * nxs; const double* us = a.unew + ((size_t)pick * a.B + b) * nus;
    for (size_t e = t; e < nxs; e += blockDim.x) a.x[(size_t)b * nxs + e] = xs[e];
    for (size_t e = t; e < nus; e += blockDim.x) a.u[(size_t)b * nus + e] = us[e];
  }
  if (t == 0) { if (pick >= 0) a.cost[b] = a.costnew[(size_t)pick * a.B + b]; if (a.step && !taken) a.step[b] = pick >= 0 ? a.alphas[pick] : 0.0; }
}

}  // namespace landing
