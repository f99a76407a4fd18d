// srbm_stage.hpp -- device math of ONE shooting stage of the SRBM landing NLP (gfx950, fp64).
//
// Hand-written CDNA4 device code (not generated, not a hipify of the reference's CasADi C).
// What it computes follows the reference formulation
//   optimizations/landing/generate_solver/generate_landingCtrller_IPOPT.m:106-170  (stage rows)
//   utilities_general/dynamics-utilities/rpyToRotMat.m:2, Binv.m:13-17            (rotation, Euler rates)
// and emits values in the order of the reference's CasADi sparsity patterns
//   landingCtrller_IPOPT.c:63 (casadi_s4, upper-triangular Hessian CCS) and :64 (casadi_s5, Jacobian CCS)
// so that one thread = one stage produces contiguous CCS segments (DESIGN.md "CCS segments").
//
// Differences from the reference's expression graph (mathematically identical, rounding-level):
//  * Binv(rpy)*R(rpy) is used in its closed form T(phi,theta) (independent of yaw), so the
//    structurally-present yaw entries of the rpy rows are emitted as exact zeros;
//  * sum_l e_j x f_l is evaluated as e_j x (sum_l f_l).
#pragma once

namespace srbm {

struct V3 { double x, y, z; };
struct M3 { double a[3][3]; };

__host__ __device__ __forceinline__ V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__host__ __device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__host__ __device__ __forceinline__ V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__host__ __device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__host__ __device__ __forceinline__ double comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
__host__ __device__ __forceinline__ V3 unit(int i) { return v3(i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0); }
__host__ __device__ __forceinline__ V3 mul(const M3& m, V3 v) {
  return v3(m.a[0][0] * v.x + m.a[0][1] * v.y + m.a[0][2] * v.z,
            m.a[1][0] * v.x + m.a[1][1] * v.y + m.a[1][2] * v.z,
            m.a[2][0] * v.x + m.a[2][1] * v.y + m.a[2][2] * v.z);
}
__host__ __device__ __forceinline__ V3 mulT(const M3& m, V3 v) {
  return v3(m.a[0][0] * v.x + m.a[1][0] * v.y + m.a[2][0] * v.z,
            m.a[0][1] * v.x + m.a[1][1] * v.y + m.a[2][1] * v.z,
            m.a[0][2] * v.x + m.a[1][2] * v.y + m.a[2][2] * v.z);
}
// hip location is (hx,hy,0): R*h only needs the first two columns
__host__ __device__ __forceinline__ V3 mul_hip(const M3& m, double hx, double hy) {
  return v3(m.a[0][0] * hx + m.a[0][1] * hy, m.a[1][0] * hx + m.a[1][1] * hy, m.a[2][0] * hx + m.a[2][1] * hy);
}

// hipSrbmLocation, get_robot_params.m:90-91 (legs FR, FL, BR, BL)
__host__ __device__ __forceinline__ double hip_x(int l) { return l < 2 ? 0.19 : -0.19; }
__host__ __device__ __forceinline__ double hip_y(int l) { return (l & 1) ? 0.1 : -0.1; }

// Rz(psi) * Q  and  Rz'(psi) * Q, Rz''(psi)*Q  (rz.m:8-13 transposed: [c -s 0; s c 0; 0 0 1])
__host__ __device__ __forceinline__ M3 rz_mul(double c, double s, const M3& q) {
  M3 r;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r.a[0][j] = c * q.a[0][j] - s * q.a[1][j];
    r.a[1][j] = s * q.a[0][j] + c * q.a[1][j];
    r.a[2][j] = q.a[2][j];
  }
  return r;
}
__host__ __device__ __forceinline__ M3 rz1_mul(double c, double s, const M3& q) {  // d/dpsi
  M3 r;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r.a[0][j] = -s * q.a[0][j] - c * q.a[1][j];
    r.a[1][j] = c * q.a[0][j] - s * q.a[1][j];
    r.a[2][j] = 0.0;
  }
  return r;
}

// Rotation set: R = rz(psi)' ry(theta)' rx(phi)' and derivatives w.r.t. e=(phi,theta,psi).
// d1[a] are stored; the six second derivatives D2(a, b) (p=phi, t=theta, s=psi: pp pt ps tt ts ss) are formed where they are
// consumed (each is used once, in the (e_b, e_a) block of the Hessian): holding them cost 108 VGPRs over the whole Hessian stream,
// which then needed the AGPR overflow of a 1-wave-per-SIMD launch bound (round 2: 444 VGPRs + 188 AGPRs).
template <bool SECOND>
struct RotSet {
  M3 R, d1[3];
  double sp, cp, st, ct, ss, cs;
  __host__ __device__ __forceinline__ void eval(double phi, double th, double psi) {
    sincos(phi, &sp, &cp);
    sincos(th, &st, &ct);
    sincos(psi, &ss, &cs);
    // Q = Ry*Rx and its phi/theta derivatives (ry.m, rx.m transposed)
    M3 Q = {{{ct, st * sp, st * cp}, {0.0, cp, -sp}, {-st, ct * sp, ct * cp}}};
    M3 Qp = {{{0.0, st * cp, -st * sp}, {0.0, -sp, -cp}, {0.0, ct * cp, -ct * sp}}};
    M3 Qt = {{{-st, ct * sp, ct * cp}, {0.0, 0.0, 0.0}, {-ct, -st * sp, -st * cp}}};
    R = rz_mul(cs, ss, Q);
    d1[0] = rz_mul(cs, ss, Qp);
    d1[1] = rz_mul(cs, ss, Qt);
    d1[2] = rz1_mul(cs, ss, Q);
  }
  __host__ __device__ __forceinline__ M3 D2(int a, int b) const {  // symmetric index
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    if (lo == 0 && hi == 0) { const M3 Qpp = {{{0.0, -st * sp, -st * cp}, {0.0, -cp, sp}, {0.0, -ct * sp, -ct * cp}}}; return rz_mul(cs, ss, Qpp); }
    if (lo == 0 && hi == 1) { const M3 Qpt = {{{0.0, ct * cp, -ct * sp}, {0.0, 0.0, 0.0}, {0.0, -st * cp, st * sp}}}; return rz_mul(cs, ss, Qpt); }
    if (lo == 0 && hi == 2) { const M3 Qp = {{{0.0, st * cp, -st * sp}, {0.0, -sp, -cp}, {0.0, ct * cp, -ct * sp}}}; return rz1_mul(cs, ss, Qp); }
    if (lo == 1 && hi == 1) { const M3 Qtt = {{{-ct, -st * sp, -st * cp}, {0.0, 0.0, 0.0}, {st, -ct * sp, -ct * cp}}}; return rz_mul(cs, ss, Qtt); }
    if (lo == 1 && hi == 2) { const M3 Qt = {{{-st, ct * sp, ct * cp}, {0.0, 0.0, 0.0}, {-ct, -st * sp, -st * cp}}}; return rz1_mul(cs, ss, Qt); }
    // Rz'' Q = -(rows 0,1 of R), row 2 = 0
    M3 r;
#pragma unroll
    for (int j = 0; j < 3; ++j) { r.a[0][j] = -R.a[0][j]; r.a[1][j] = -R.a[1][j]; r.a[2][j] = 0.0; }
    return r;
  }
};

// Per-stage constants (from p: dt_k, mu, mass, Ib, Ib_inv; formulation: kin_z_off)
struct StageParams {
  double dt, dt_over_m, km, Ib[3], Ibi[3], kin_z_off, mass;
};

// Decision variables a stage touches. X=[pos rpy omega v], c/f per leg, next state and next feet.
struct StageVars {
  double X[12], c[12], f[12], Xn[12], cn[12];
};

// ------------------------------------------------------------------------------------------------
// residual rows of one stage, written at their row offsets. `last`: k==N-1.
// ------------------------------------------------------------------------------------------------
template <typename Out>
__host__ __device__ __forceinline__ void stage_g(const StageVars& z, const StageParams& P, bool last, Out& out) {
  RotSet<false> RS;
  RS.eval(z.X[3], z.X[4], z.X[5]);
  const V3 pos = v3(z.X[0], z.X[1], z.X[2]), w = v3(z.X[6], z.X[7], z.X[8]), v = v3(z.X[9], z.X[10], z.X[11]);
  const double sec = 1.0 / RS.ct, tt = RS.st * sec;
  const double aa = RS.sp * w.y + RS.cp * w.z, bb = RS.cp * w.y - RS.sp * w.z;
  const V3 edot = v3(w.x + tt * aa, bb, aa * sec);            // Binv(rpy)*(R*omega), gen:128
  V3 fs = v3(0, 0, 0), tau = v3(0, 0, 0);
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const V3 r = v3(z.c[3 * l] - pos.x, z.c[3 * l + 1] - pos.y, z.c[3 * l + 2] - pos.z);
    const V3 f = v3(z.f[3 * l], z.f[3 * l + 1], z.f[3 * l + 2]);
    fs = fs + f;
    tau = tau + cross(r, f);                                   // gen:120-123
  }
  const V3 taub = mulT(RS.R, tau);
  const V3 Iw = v3(P.Ib[0] * w.x, P.Ib[1] * w.y, P.Ib[2] * w.z);
  const V3 nn = cross(w, Iw);                                  // gen:124
  const V3 omd = v3(P.Ibi[0] * (taub.x - nn.x), P.Ibi[1] * (taub.y - nn.y), P.Ibi[2] * (taub.z - nn.z));
  out.put(0, z.Xn[0] - pos.x - v.x * P.dt);                    // gen:127
  out.put(1, z.Xn[1] - pos.y - v.y * P.dt);
  out.put(2, z.Xn[2] - pos.z - v.z * P.dt);
  out.put(3, z.Xn[3] - z.X[3] - edot.x * P.dt);                // gen:128
  out.put(4, z.Xn[4] - z.X[4] - edot.y * P.dt);
  out.put(5, z.Xn[5] - z.X[5] - edot.z * P.dt);
  out.put(6, z.Xn[9] - v.x - (fs.x * P.dt_over_m));            // gen:129 (gravity only in z)
  out.put(7, z.Xn[10] - v.y - (fs.y * P.dt_over_m));
  out.put(8, z.Xn[11] - v.z - (fs.z * P.dt_over_m - 9.81 * P.dt));
  out.put(9, z.Xn[6] - w.x - omd.x * P.dt);                    // gen:130
  out.put(10, z.Xn[7] - w.y - omd.y * P.dt);
  out.put(11, z.Xn[8] - w.z - omd.z * P.dt);
  // rows are emitted in increasing order (sequential emitters rely on it)
  const int stride = last ? 6 : 12, kin = last ? 2 : 8, fric = last ? 40 : 64, box = last ? 56 : 80;
#pragma unroll
  for (int l = 0; l < 4; ++l) out.put(12 + l, z.f[3 * l + 2]);                        // gen:133
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int rb = 16 + stride * l;
    const double fz = z.f[3 * l + 2], cz = z.c[3 * l + 2];
    out.put(rb, cz);                                           // gen:139
    out.put(rb + 1, fz * cz);                                  // gen:140
    if (!last) {
      const double d0 = fz * (z.cn[3 * l] - z.c[3 * l]), d1 = fz * (z.cn[3 * l + 1] - z.c[3 * l + 1]), d2 = fz * (z.cn[3 * l + 2] - z.c[3 * l + 2]);
      out.put(rb + 2, d0); out.put(rb + 3, d1); out.put(rb + 4, d2);   // gen:143
      out.put(rb + 5, d0); out.put(rb + 6, d1); out.put(rb + 7, d2);   // gen:144
    }
    const V3 Rh = mul_hip(RS.R, hip_x(l), hip_y(l));
    const V3 pr = v3(z.c[3 * l] - (pos.x + Rh.x), z.c[3 * l + 1] - (pos.y + Rh.y), z.c[3 * l + 2] - (pos.z + Rh.z));
    out.put(rb + kin, pr.x);                                   // gen:153-156
    out.put(rb + kin + 1, pr.y);
    out.put(rb + kin + 2, pr.z + P.kin_z_off);
    out.put(rb + kin + 3, dot(pr, pr));
  }
#pragma unroll
  for (int l = 0; l < 4; ++l) out.put(fric + l, z.f[3 * l] - P.km * z.f[3 * l + 2]);            // gen:160-163
#pragma unroll
  for (int l = 0; l < 4; ++l) out.put(fric + 4 + l, -P.km * z.f[3 * l + 2] - z.f[3 * l]);
#pragma unroll
  for (int l = 0; l < 4; ++l) out.put(fric + 8 + l, z.f[3 * l + 1] - P.km * z.f[3 * l + 2]);
#pragma unroll
  for (int l = 0; l < 4; ++l) out.put(fric + 12 + l, -P.km * z.f[3 * l + 2] - z.f[3 * l + 1]);
#pragma unroll
  for (int i = 0; i < 6; ++i) out.put(box + i, z.X[i]);                               // gen:166-169
#pragma unroll
  for (int i = 0; i < 6; ++i) out.put(box + 6 + i, z.X[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) out.put(box + 12 + i, z.X[6 + i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) out.put(box + 18 + i, z.X[6 + i]);
}

// ------------------------------------------------------------------------------------------------
// Jacobian nonzeros of the columns (X_k, U_k) in casadi_s5 order.
//   ex: X_k columns, 157 values (first entry of every column is the 1.0 of the initial-state row
//       (k==0) or of stage k-1's "X+" identity);
//   eu: U_k columns: 228 values (k>=1, not last), 204 (k==0), 180 (last);  fz_prev = f_z of stage k-1.
// Emitter interface: e.col() starts a column, e.put(row, v) adds a nonzero; row >= 0 is a row of this
// stage (offsets of SURVEY App. A), PREV_ID(i) the identity/initial row of state i, PREV_SLIP(l,j,q)
// the no-slip row (q=0: "<=" row, q=1: ">=" row) of leg l, axis j in stage k-1.
// ------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int PREV_ID(int i) { return -1 - i; }
__host__ __device__ __forceinline__ int PREV_SLIP(int l, int j, int q) { return -100 - (6 * l + 3 * q + j); }

template <typename EX, typename EU>
__host__ __device__ __forceinline__ void stage_jac(const StageVars& z, const StageParams& P, bool first, bool last,
                                          const double fz_prev[4], EX& ex, EU& eu) {
  RotSet<false> RS;
  RS.eval(z.X[3], z.X[4], z.X[5]);
  const V3 pos = v3(z.X[0], z.X[1], z.X[2]), w = v3(z.X[6], z.X[7], z.X[8]);
  const double sec = 1.0 / RS.ct, tt = RS.st * sec, sec2 = sec * sec;
  const double aa = RS.sp * w.y + RS.cp * w.z, bb = RS.cp * w.y - RS.sp * w.z;
  const double dt = P.dt;
  const int stride = last ? 6 : 12, kin = last ? 2 : 8, fric = last ? 40 : 64, box = last ? 56 : 80;
  V3 F = v3(0, 0, 0), tau = v3(0, 0, 0);
  V3 r[4], f[4], prel[4], Rah[3][4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    r[l] = v3(z.c[3 * l] - pos.x, z.c[3 * l + 1] - pos.y, z.c[3 * l + 2] - pos.z);
    f[l] = v3(z.f[3 * l], z.f[3 * l + 1], z.f[3 * l + 2]);
    F = F + f[l];
    tau = tau + cross(r[l], f[l]);
    const V3 Rh = mul_hip(RS.R, hip_x(l), hip_y(l));
    prel[l] = r[l] - Rh;
#pragma unroll
    for (int a = 0; a < 3; ++a) Rah[a][l] = mul_hip(RS.d1[a], hip_x(l), hip_y(l));
  }
  // ---- X_k columns ----
#pragma unroll
  for (int j = 0; j < 3; ++j) {                  // pos_j
    ex.col();
    ex.put(PREV_ID(j), 1.0);
    ex.put(j, -1.0);                             // g_pos_j
    const V3 t = mulT(RS.R, cross(unit(j), F));  // + dt*Ibi*R^T(e_j x F)
    ex.put(9, dt * P.Ibi[0] * t.x); ex.put(10, dt * P.Ibi[1] * t.y); ex.put(11, dt * P.Ibi[2] * t.z);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const int rk = 16 + stride * l + kin;
      ex.put(rk + j, -1.0); ex.put(rk + 3, -2.0 * comp(prel[l], j));
    }
    ex.put(box + j, 1.0); ex.put(box + 6 + j, 1.0);
  }
  {
    // d edot / d e : phi, theta (psi: 0)
    const V3 de[3] = {v3(tt * bb, -aa, bb * sec), v3(aa * sec2, 0.0, aa * sec * tt), v3(0, 0, 0)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {                // rpy_a
      ex.col();
      ex.put(PREV_ID(3 + a), 1.0);
      ex.put(3, (a == 0 ? -1.0 : 0.0) - dt * de[a].x);
      ex.put(4, (a == 1 ? -1.0 : 0.0) - dt * de[a].y);
      ex.put(5, (a == 2 ? -1.0 : 0.0) - dt * de[a].z);
      const V3 t = mulT(RS.d1[a], tau);
      if (a != 0) ex.put(9, -dt * P.Ibi[0] * t.x);  // R(:,1) has no roll dependence: not in the pattern
      ex.put(10, -dt * P.Ibi[1] * t.y); ex.put(11, -dt * P.Ibi[2] * t.z);
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const int rk = 16 + stride * l + kin;
        ex.put(rk, -Rah[a][l].x); ex.put(rk + 1, -Rah[a][l].y);
        if (a != 2) ex.put(rk + 2, -Rah[a][l].z);   // (R*hip)_z has no yaw dependence
        ex.put(rk + 3, -2.0 * dot(prel[l], Rah[a][l]));
      }
      ex.put(box + 3 + a, 1.0); ex.put(box + 9 + a, 1.0);
    }
  }
  {
    // T = Binv*R closed form; dn/dw
    const double T[3][3] = {{1.0, RS.sp * tt, RS.cp * tt}, {0.0, RS.cp, -RS.sp}, {0.0, RS.sp * sec, RS.cp * sec}};
    const double dn[3][3] = {{0.0, (P.Ib[2] - P.Ib[1]) * w.z, (P.Ib[2] - P.Ib[1]) * w.y},
                             {(P.Ib[0] - P.Ib[2]) * w.z, 0.0, (P.Ib[0] - P.Ib[2]) * w.x},
                             {(P.Ib[1] - P.Ib[0]) * w.y, (P.Ib[1] - P.Ib[0]) * w.x, 0.0}};
#pragma unroll
    for (int j = 0; j < 3; ++j) {                // omega_j
      ex.col();
      ex.put(PREV_ID(6 + j), 1.0);
#pragma unroll
      for (int i = 0; i < 3; ++i) ex.put(3 + i, -dt * T[i][j]);
#pragma unroll
      for (int i = 0; i < 3; ++i) ex.put(9 + i, (i == j ? -1.0 : 0.0) + dt * P.Ibi[i] * dn[i][j]);
      ex.put(box + 12 + j, 1.0); ex.put(box + 18 + j, 1.0);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {                  // v_j
    ex.col();
    ex.put(PREV_ID(9 + j), 1.0); ex.put(j, -dt); ex.put(6 + j, -1.0); ex.put(box + 15 + j, 1.0); ex.put(box + 21 + j, 1.0);
  }
  ex.end();                                      // X part complete (emitters that share a buffer between the parts switch here)
  // ---- U_k columns: feet ----
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const double fz = f[l].z;
    const int rb = 16 + stride * l, rk = rb + kin;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      eu.col();
      if (!first) { eu.put(PREV_SLIP(l, j, 0), fz_prev[l]); eu.put(PREV_SLIP(l, j, 1), fz_prev[l]); }
      const V3 t = mulT(RS.R, cross(unit(j), f[l]));           // d tau_b/d c_lj = R^T(e_j x f_l)
      eu.put(9, -dt * P.Ibi[0] * t.x); eu.put(10, -dt * P.Ibi[1] * t.y); eu.put(11, -dt * P.Ibi[2] * t.z);
      if (j == 2) { eu.put(rb, 1.0); eu.put(rb + 1, fz); }      // c_z row, f_z*c_z row
      if (!last) { eu.put(rb + 2 + j, -fz); eu.put(rb + 5 + j, -fz); }   // no-slip "<=" / ">=" rows
      eu.put(rk + j, 1.0);                                      // p_rel_j
      eu.put(rk + 3, 2.0 * comp(prel[l], j));                   // |p_rel|^2
    }
  }
  // ---- U_k columns: forces ----
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int rb = 16 + stride * l;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      eu.col();
      eu.put(6 + j, -P.dt_over_m);                              // g_v_j
      const V3 t = mulT(RS.R, cross(r[l], unit(j)));            // d tau_b/d f_lj = R^T(r_l x e_j)
      eu.put(9, -dt * P.Ibi[0] * t.x); eu.put(10, -dt * P.Ibi[1] * t.y); eu.put(11, -dt * P.Ibi[2] * t.z);
      if (j == 2) {
        eu.put(12 + l, 1.0);                                    // f_z row
        eu.put(rb + 1, z.c[3 * l + 2]);                         // f_z*c_z
        if (!last) {
          const double d0 = z.cn[3 * l] - z.c[3 * l], d1 = z.cn[3 * l + 1] - z.c[3 * l + 1], d2 = z.cn[3 * l + 2] - z.c[3 * l + 2];
          eu.put(rb + 2, d0); eu.put(rb + 3, d1); eu.put(rb + 4, d2); eu.put(rb + 5, d0); eu.put(rb + 6, d1); eu.put(rb + 7, d2);
        }
        eu.put(fric + l, -P.km); eu.put(fric + 4 + l, -P.km); eu.put(fric + 8 + l, -P.km); eu.put(fric + 12 + l, -P.km);
      } else {
        eu.put(fric + 8 * j + l, 1.0); eu.put(fric + 8 * j + 4 + l, -1.0);   // friction rows (<=, >=)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Upper-triangular Lagrangian-Hessian nonzeros of the columns (X_k, U_k) in casadi_s4 order.
//   lam: multipliers of this stage's rows (row offsets as in stage_g); lam_prev_slip[l][i]: sum of the
//   two no-slip multipliers (ub+lb row) of stage k-1.  hx: 29 values; hu: 160 (k>=1) / 148 (k==0).
// ------------------------------------------------------------------------------------------------
template <typename Lam, typename EX, typename EU>
__host__ __device__ __forceinline__ void stage_hess(const StageVars& z, const StageParams& P, bool first, bool last,
                                           const Lam& lam, const double lam_prev_slip[12], EX& hx, EU& hu) {
  RotSet<true> RS;
  RS.eval(z.X[3], z.X[4], z.X[5]);
  const V3 pos = v3(z.X[0], z.X[1], z.X[2]), w = v3(z.X[6], z.X[7], z.X[8]);
  const double sec = 1.0 / RS.ct, tt = RS.st * sec, sec2 = sec * sec;
  const double aa = RS.sp * w.y + RS.cp * w.z, bb = RS.cp * w.y - RS.sp * w.z;
  const double dt = P.dt;
  const int stride = last ? 6 : 12, kin = last ? 2 : 8;
  const V3 le = v3(lam(3), lam(4), lam(5));
  const V3 mub = v3(lam(9) * P.Ibi[0], lam(10) * P.Ibi[1], lam(11) * P.Ibi[2]);
  const V3 m = mul(RS.R, mub);
  V3 Ram[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) Ram[a] = mul(RS.d1[a], mub);
  V3 tau = v3(0, 0, 0);
  V3 r[4], f[4], prel[4], Rah[3][4];
  double lL[4], lslip[4][3], lcomp[4];
  V3 lbox[4];
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const int rb = 16 + stride * l;
    r[l] = v3(z.c[3 * l] - pos.x, z.c[3 * l + 1] - pos.y, z.c[3 * l + 2] - pos.z);
    f[l] = v3(z.f[3 * l], z.f[3 * l + 1], z.f[3 * l + 2]);
    tau = tau + cross(r[l], f[l]);
    prel[l] = r[l] - mul_hip(RS.R, hip_x(l), hip_y(l));
#pragma unroll
    for (int a = 0; a < 3; ++a) Rah[a][l] = mul_hip(RS.d1[a], hip_x(l), hip_y(l));
    lcomp[l] = lam(rb + 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) lslip[l][i] = last ? 0.0 : (lam(rb + 2 + i) + lam(rb + 5 + i));
    lbox[l] = v3(lam(rb + kin), lam(rb + kin + 1), lam(rb + kin + 2));
    lL[l] = lam(rb + kin + 3);
  }
  // (pos,pos) diagonal
  const double hpp = 2.0 * (lL[0] + lL[1] + lL[2] + lL[3]);
  hx.put(hpp); hx.put(hpp); hx.put(hpp);
  // (pos,e_a), (e_b,e_a)
  // second derivatives of edot (closed form of Binv*R): pp, pt, tt ; anything with psi = 0
  const V3 dde_pp = v3(-tt * aa, -bb, -aa * sec), dde_pt = v3(bb * sec2, 0.0, bb * sec * tt),
           dde_tt = v3(2.0 * aa * sec2 * tt, 0.0, aa * sec * (tt * tt + sec2));
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    V3 hp = v3(0, 0, 0);
#pragma unroll
    for (int l = 0; l < 4; ++l) hp = hp + dt * cross(f[l], Ram[a]) + (2.0 * lL[l]) * Rah[a][l];
    hx.put(hp.x); hx.put(hp.y); hx.put(hp.z);
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      const M3 Rab = RS.D2(a, b);
      double s = -dt * dot(mul(Rab, mub), tau);
      if (a == 0 && b == 0) s += -dt * dot(le, dde_pp);
      if (a == 1 && b == 0) s += -dt * dot(le, dde_pt);
      if (a == 1 && b == 1) s += -dt * dot(le, dde_tt);
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const V3 t = mul_hip(Rab, hip_x(l), hip_y(l));
        s += -dot(lbox[l], t) + 2.0 * lL[l] * (dot(Rah[a][l], Rah[b][l]) - dot(prel[l], t));
      }
      hx.put(s);
    }
  }
  // (e_a, omega_j), (omega_i, omega_j)
  {
    const double Tp[3][3] = {{0.0, RS.cp * tt, -RS.sp * tt}, {0.0, -RS.sp, -RS.cp}, {0.0, RS.cp * sec, -RS.sp * sec}};
    const double Tt[3][3] = {{0.0, RS.sp * sec2, RS.cp * sec2}, {0.0, 0.0, 0.0}, {0.0, RS.sp * sec * tt, RS.cp * sec * tt}};
    const double lev[3] = {le.x, le.y, le.z};
    double hew[3][3];  // [a][j]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      hew[0][j] = -dt * (lev[0] * Tp[0][j] + lev[1] * Tp[1][j] + lev[2] * Tp[2][j]);
      hew[1][j] = -dt * (lev[0] * Tt[0][j] + lev[1] * Tt[1][j] + lev[2] * Tt[2][j]);
      hew[2][j] = 0.0;
    }
    const double hw01 = dt * mub.z * (P.Ib[1] - P.Ib[0]), hw02 = dt * mub.y * (P.Ib[0] - P.Ib[2]),
                 hw12 = dt * mub.x * (P.Ib[2] - P.Ib[1]);
    hx.put(hew[1][0]); hx.put(hew[2][0]);                               // omega_x: rows theta, psi
    hx.put(hew[0][1]); hx.put(hew[1][1]); hx.put(hew[2][1]); hx.put(hw01);
    hx.put(hew[0][2]); hx.put(hew[1][2]); hx.put(hew[2][2]); hx.put(hw02); hx.put(hw12);
  }
  hx.end();
  // ---- U_k columns: feet ----
#pragma unroll
  for (int l = 0; l < 4; ++l) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      hu.put(-2.0 * lL[l]);                                             // (pos_j, c_lj)
#pragma unroll
      for (int a = 0; a < 3; ++a)
        hu.put(-dt * comp(cross(f[l], Ram[a]), j) - 2.0 * lL[l] * comp(Rah[a][l], j));
      if (!first) hu.put(lam_prev_slip[3 * l + j]);                     // (f_z of stage k-1, c_lj)
      hu.put(2.0 * lL[l]);                                              // diagonal
    }
  }
  // ---- U_k columns: forces ----
  const double mm[3] = {m.x, m.y, m.z};
#pragma unroll
  for (int l = 0; l < 4; ++l) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // eps_{ijq} m_q for i != j
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (i == j) continue;
        const int q = 3 - i - j;
        const double e = ((j - i + 3) % 3 == 1) ? 1.0 : -1.0;
        hu.put(dt * e * mm[q]);                                         // (pos_i, f_lj)
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) hu.put(-dt * comp(cross(Ram[a], r[l]), j));   // (e_a, f_lj)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (i == j && j != 2) continue;
        double val = 0.0;
        if (i != j) {
          const int q = 3 - i - j;
          const double e = ((j - i + 3) % 3 == 1) ? 1.0 : -1.0;
          val = -dt * e * mm[q];
        }
        if (j == 2) { val -= lslip[l][i]; if (i == 2) val += lcomp[l]; }
        hu.put(val);                                                    // (c_li, f_lj)
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// d(lam^T g_k)/dp contributions of one stage (nlp_grad's grad_gamma_p, landingCtrller_IPOPT.c:22015):
//   out[0]=d/d dt_k, out[1]=d/d mu, out[2]=d/d mass, out[3..5]=d/d Ib, out[6..8]=d/d Ib_inv
// ------------------------------------------------------------------------------------------------
template <typename Lam>
__host__ __device__ __forceinline__ void stage_gradp(const StageVars& z, const StageParams& P, bool last, const Lam& lam, double out[9]) {
  RotSet<false> RS;
  RS.eval(z.X[3], z.X[4], z.X[5]);
  const V3 pos = v3(z.X[0], z.X[1], z.X[2]), w = v3(z.X[6], z.X[7], z.X[8]), v = v3(z.X[9], z.X[10], z.X[11]);
  const double sec = 1.0 / RS.ct, tt = RS.st * sec;
  const double aa = RS.sp * w.y + RS.cp * w.z, bb = RS.cp * w.y - RS.sp * w.z;
  const V3 edot = v3(w.x + tt * aa, bb, aa * sec);
  V3 fs = v3(0, 0, 0), tau = v3(0, 0, 0);
  double smu = 0.0;
  const int fric = last ? 40 : 64;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const V3 r = v3(z.c[3 * l] - pos.x, z.c[3 * l + 1] - pos.y, z.c[3 * l + 2] - pos.z);
    const V3 f = v3(z.f[3 * l], z.f[3 * l + 1], z.f[3 * l + 2]);
    fs = fs + f;
    tau = tau + cross(r, f);
    smu += f.z * (lam(fric + l) + lam(fric + 4 + l) + lam(fric + 8 + l) + lam(fric + 12 + l));
  }
  const V3 taub = mulT(RS.R, tau);
  const V3 Iw = v3(P.Ib[0] * w.x, P.Ib[1] * w.y, P.Ib[2] * w.z);
  const V3 nn = cross(w, Iw);
  const V3 tb = taub - nn;
  const V3 omd = v3(P.Ibi[0] * tb.x, P.Ibi[1] * tb.y, P.Ibi[2] * tb.z);
  const double inv_m = 1.0 / P.mass;
  const V3 rdd = v3(fs.x * inv_m, fs.y * inv_m, fs.z * inv_m - 9.81);
  out[0] = -(lam(0) * v.x + lam(1) * v.y + lam(2) * v.z + lam(3) * edot.x + lam(4) * edot.y + lam(5) * edot.z +
             lam(6) * rdd.x + lam(7) * rdd.y + lam(8) * rdd.z + lam(9) * omd.x + lam(10) * omd.y + lam(11) * omd.z);
  out[1] = -0.71 * smu;
  out[2] = P.dt * inv_m * inv_m * (lam(6) * fs.x + lam(7) * fs.y + lam(8) * fs.z);
  out[3] = P.dt * (lam(10) * P.Ibi[1] * w.z * w.x - lam(11) * P.Ibi[2] * w.x * w.y);
  out[4] = P.dt * (-lam(9) * P.Ibi[0] * w.y * w.z + lam(11) * P.Ibi[2] * w.x * w.y);
  out[5] = P.dt * (lam(9) * P.Ibi[0] * w.y * w.z - lam(10) * P.Ibi[1] * w.x * w.z);
  out[6] = -P.dt * lam(9) * tb.x;
  out[7] = -P.dt * lam(10) * tb.y;
  out[8] = -P.dt * lam(11) * tb.z;
}

}  // namespace srbm
