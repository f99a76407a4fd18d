// vbl_kernels.hip -- SRBM variational linearisation A (24x24), B (24x12) and the Riccati differential equation for the
// tracking gains along a solved landing trajectory, batched (gfx950, fp64).  SURVEY section 8(f) row N3.
//
// What it computes follows the reference's tracking-controller synthesis
//   utilities_general/srbm-utilities/generateVariationalDynamics.m:29-62   (error dynamics on SO(3), A = d(dxdot)/d(dx), B = d(dxdot)/d(df))
//   utilities_general/srbm-utilities/generateRiccatiIntegrator.m:24-62     (Pdot = A'P + PA - PB R^-1 B'P + Q; backward step P0 = Pf + dt*k1)
//   optimizations/landing/quadruped_SRBM_NLP.m:428-535                     (weights, backward sweep over the sampled trajectory)
// which the reference builds symbolically with CasADi and runs one step at a time from MATLAB.  Here: one workgroup
// (4 wavefronts) per trajectory; P, A, B live in LDS as zero-padded 32 x 32 arrays and every 24 x 24 product runs on the
// fp64 matrix cores (v_mfma_f64_16x16x4, wave w owns output tile (w >> 1, w & 1)); the closed forms of A and B are
// written out by hand (the reference differentiates the error dynamics symbolically -- they are linear in the error).
//
// State order [dp(3) deta(3) domega(3) dv(3) dpf(12)], control = the twelve ground-reaction force components.
#include <hip/hip_runtime.h>
#include <math.h>

namespace landing {

constexpr int VN = 24, VM = 12, VP = 32, VLD = 33;    // states, controls, padded size, LDS row stride

struct VblConst { double Ib[9], Ibi[9], inv_m; };     // body inertia (row-major 3x3), its inverse, 1/mass

// closed forms of generateVariationalDynamics.m:33-55.  A, B: LDS arrays [VP][VLD], zero outside the 24 x 24 / 24 x 12 blocks.
// Called by all threads of the workgroup; the caller synchronises afterwards.
__device__ __forceinline__ void vbl_fill(const double* __restrict__ xr, const double* __restrict__ fr, const VblConst& C, double* A, double* Bm) {
  const int tid = threadIdx.x, NT = blockDim.x;
  for (int e = tid; e < VP * VLD; e += NT) { A[e] = 0.0; Bm[e] = 0.0; }
  __syncthreads();
  if (tid == 0) {
    // rotation: R = rpyToRotMat(rpy)' with rpyToRotMat = rz' ry' rx' (rpyToRotMat.m:2)  =>  R' = rz' ry' rx'
    double sp, cp, st, ct, ss, cs;
    sincos(xr[3], &sp, &cp); sincos(xr[4], &st, &ct); sincos(xr[5], &ss, &cs);
    // Rt = rz(psi)' * ry(theta)' * rx(phi)'  (rx.m, ry.m, rz.m are the coordinate-transform forms)
    const double Rt[3][3] = {{cs * ct, cs * st * sp - ss * cp, cs * st * cp + ss * sp},
                             {ss * ct, ss * st * sp + cs * cp, ss * st * cp - cs * sp},
                             {-st, ct * sp, ct * cp}};
    const double p[3] = {xr[0], xr[1], xr[2]}, w[3] = {xr[6], xr[7], xr[8]};
    double fs[3] = {0, 0, 0}, tau[3] = {0, 0, 0};                    // sum f_l, sum R' (pf_l - p) x f_l
    for (int l = 0; l < 4; ++l) {
      const double r[3] = {xr[12 + 3 * l] - p[0], xr[13 + 3 * l] - p[1], xr[14 + 3 * l] - p[2]};
      const double f[3] = {fr[3 * l], fr[3 * l + 1], fr[3 * l + 2]};
      const double c[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
      for (int i = 0; i < 3; ++i) { fs[i] += f[i]; tau[i] += Rt[i][0] * c[0] + Rt[i][1] * c[1] + Rt[i][2] * c[2]; }
    }
    auto skew = [](const double* v, double S[3][3]) { S[0][0] = 0; S[0][1] = -v[2]; S[0][2] = v[1]; S[1][0] = v[2]; S[1][1] = 0; S[1][2] = -v[0]; S[2][0] = -v[1]; S[2][1] = v[0]; S[2][2] = 0; };
    auto put3 = [&](double* M, int r0, int c0, const double X[3][3]) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[(r0 + i) * VLD + c0 + j] = X[i][j]; };
    auto mm = [](const double X[3][3], const double Y[3][3], double Z[3][3]) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Z[i][j] = X[i][0] * Y[0][j] + X[i][1] * Y[1][j] + X[i][2] * Y[2][j]; };
    double Ibi[3][3], Ib[3][3], S[3][3], T[3][3], U[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Ibi[i][j] = C.Ibi[3 * i + j]; Ib[i][j] = C.Ib[3 * i + j]; }
    for (int i = 0; i < 3; ++i) A[i * VLD + 9 + i] = 1.0;                               // :33  dp' = dv
    skew(w, S);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[(3 + i) * VLD + 3 + j] = -S[i][j];   // :34  deta' = -w x deta + domega
    for (int i = 0; i < 3; ++i) A[(3 + i) * VLD + 6 + i] = 1.0;
    // :36-52  domega' = Ib^-1 ( skew(tau) deta + R'( -sum skew(f_l) dpf_l + skew(sum f) dp + sum skew(pf_l - p) df_l ) + (skew(Ib w) - skew(w) Ib) domega )
    { double IR[3][3]; mm(Ibi, Rt, IR);
      skew(fs, S); mm(IR, S, T); put3(A, 6, 0, T);                                       // dp
      skew(tau, S); mm(Ibi, S, T); put3(A, 6, 3, T);                                     // deta
      double Iw[3]; for (int i = 0; i < 3; ++i) Iw[i] = Ib[i][0] * w[0] + Ib[i][1] * w[1] + Ib[i][2] * w[2];
      double S2[3][3]; skew(Iw, S); skew(w, S2); mm(S2, Ib, U);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[i][j] -= U[i][j];
      mm(Ibi, S, T); put3(A, 6, 6, T);                                                    // domega
      for (int l = 0; l < 4; ++l) {
        const double f[3] = {fr[3 * l], fr[3 * l + 1], fr[3 * l + 2]};
        skew(f, S); mm(IR, S, T);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i][j] = -T[i][j];
        put3(A, 6, 12 + 3 * l, T);                                                        // dpf_l
        const double r[3] = {xr[12 + 3 * l] - p[0], xr[13 + 3 * l] - p[1], xr[14 + 3 * l] - p[2]};
        skew(r, S); mm(IR, S, T); put3(Bm, 6, 3 * l, T);                                  // df_l
        for (int i = 0; i < 3; ++i) Bm[(9 + i) * VLD + 3 * l + i] = C.inv_m;             // :54  dv' = 1/m sum df_l
      } }
    for (int i = 12; i < 24; ++i) A[i * VLD + i] = -0.00001;                             // :56  small stabilising term
  }
}

// C tile (ti, tj) of X(32 x 4KT) * Y(4KT x 32) on the fp64 matrix cores; X, Y in LDS (row stride VLD); YT: read Y transposed
template <int KT, bool YT>
__device__ __forceinline__ void mm_tile(const double* X, const double* Y, double (&c)[4]) {
  typedef double f64x4 __attribute__((vector_size(32)));
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, ti = w >> 1, tj = w & 1, ij = l & 15, kq = l >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = 4 * kt + kq;
    const double a = X[(16 * ti + ij) * VLD + k];
    const double b = YT ? Y[(16 * tj + ij) * VLD + k] : Y[k * VLD + 16 * tj + ij];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = acc[r];
}

struct VblLds { double P[VP * VLD], A[VP * VLD], B[VP * VLD], PA[VP * VLD], PB[VP * VLD], PBr[VP * VLD], W[VP * VLD], Kacc[VP * VLD]; };

// D = Pin A + (Pin A)' - (Pin B) R^-1 (Pin B)' + Q  into the caller's accumulator-layout registers (tile of the wave)
__device__ __forceinline__ void rde_rhs(VblLds& S, const double* Pin, const double* __restrict__ Qm, const double* __restrict__ rinv, double (&d)[4]) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, ti = w >> 1, tj = w & 1;
  double c[4];
  mm_tile<6, false>(Pin, S.A, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) S.PA[(16 * ti + (l >> 4) + 4 * r) * VLD + 16 * tj + (l & 15)] = c[r];
  mm_tile<6, false>(Pin, S.B, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 16 * ti + (l >> 4) + 4 * r, j = 16 * tj + (l & 15);
    S.PB[i * VLD + j] = c[r];
    S.PBr[i * VLD + j] = j < VM ? c[r] * rinv[j] : 0.0;
  }
  __syncthreads();
  mm_tile<3, true>(S.PBr, S.PB, c);           // (PB R^-1) (PB)'
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 16 * ti + (l >> 4) + 4 * r, j = 16 * tj + (l & 15);
    const bool in = i < VN && j < VN;
    d[r] = in ? S.PA[i * VLD + j] + S.PA[j * VLD + i] - c[r] + Qm[i * VN + j] : 0.0;
  }
  __syncthreads();
}

struct RdeArgs {
  int B, n, rk4; double dt;
  const double* xref; const double* fref;      // [B][n][24], [B][n][12]
  const double* Q; const double* rinv; const double* F;   // device: 24x24 row-major, 1/diag(R) (12), terminal 24x24
  double* P; double* K;                         // [B][n][576] row-major, [B][n][12*24] (may be null)
  double* Aout; double* Bout;                   // optional [B][n][576], [B][n][288]
  VblConst C;
};

// One workgroup (256 threads) per trajectory.  Backward sweep j = n-1 .. 1:  P[j-1] = P[j] + dt * f(P[j]; ref_j)
// (generateRiccatiIntegrator.m:47, `P0 = Pf + dt*k1`; rk4 = 1 selects the classical RK4 combination of :43-46 that the
// reference keeps commented out for the backward step and uses for its forward check).  Gains K_j = R^-1 B_j' P_j.
__global__ void __launch_bounds__(256) landing_rde_kernel(RdeArgs a) {
  __shared__ VblLds S;
  const int m = blockIdx.x;
  if (m >= a.B) return;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, ti = w >> 1, tj = w & 1;
  double* Pm = a.P ? a.P + (size_t)m * a.n * VN * VN : nullptr;
  for (int e = tid; e < VP * VLD; e += 256) { const int i = e / VLD, j = e % VLD; S.P[e] = (a.F && i < VN && j < VN) ? a.F[i * VN + j] : 0.0; }
  __syncthreads();
  for (int j = a.n - 1; j >= 0; --j) {
    const double* xr = a.xref + ((size_t)m * a.n + j) * 24;
    const double* fr = a.fref + ((size_t)m * a.n + j) * 12;
    vbl_fill(xr, fr, a.C, S.A, S.B);
    __syncthreads();
    // outputs of grid point j: P_j, K_j = R^-1 B_j' P_j, A_j, B_j
    if (Pm) for (int e = tid; e < VN * VN; e += 256) Pm[(size_t)j * VN * VN + e] = S.P[(e / VN) * VLD + e % VN];
    if (a.Aout) for (int e = tid; e < VN * VN; e += 256) a.Aout[((size_t)m * a.n + j) * VN * VN + e] = S.A[(e / VN) * VLD + e % VN];
    if (a.Bout) for (int e = tid; e < VN * VM; e += 256) a.Bout[((size_t)m * a.n + j) * VN * VM + e] = S.B[(e / VM) * VLD + e % VM];
    if (a.K) {
      for (int e = tid; e < VM * VN; e += 256) {
        const int u = e / VN, s = e % VN;
        double acc = 0.0;
        for (int t = 0; t < VN; ++t) acc += S.B[t * VLD + u] * S.P[t * VLD + s];
        a.K[((size_t)m * a.n + j) * VM * VN + e] = acc * a.rinv[u];
      }
    }
    if (j == 0) break;
    double d[4];
    if (!a.rk4) {
      rde_rhs(S, S.P, a.Q, a.rinv, d);
#pragma unroll
      for (int r = 0; r < 4; ++r) S.P[(16 * ti + (l >> 4) + 4 * r) * VLD + 16 * tj + (l & 15)] += a.dt * d[r];
    } else {
      double k[4], acc[4];
      rde_rhs(S, S.P, a.Q, a.rinv, k);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int e = (16 * ti + (l >> 4) + 4 * r) * VLD + 16 * tj + (l & 15); acc[r] = k[r]; S.W[e] = S.P[e] + 0.5 * a.dt * k[r]; }
      __syncthreads();
      rde_rhs(S, S.W, a.Q, a.rinv, k);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int e = (16 * ti + (l >> 4) + 4 * r) * VLD + 16 * tj + (l & 15); acc[r] += 2.0 * k[r]; S.W[e] = S.P[e] + 0.5 * a.dt * k[r]; }
      __syncthreads();
      rde_rhs(S, S.W, a.Q, a.rinv, k);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int e = (16 * ti + (l >> 4) + 4 * r) * VLD + 16 * tj + (l & 15); acc[r] += 2.0 * k[r]; S.W[e] = S.P[e] + a.dt * k[r]; }
      __syncthreads();
      rde_rhs(S, S.W, a.Q, a.rinv, k);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int e = (16 * ti + (l >> 4) + 4 * r) * VLD + 16 * tj + (l & 15); S.P[e] += a.dt * (acc[r] + k[r]) / 6.0; }
    }
    __syncthreads();
  }
}

}  // namespace landing
