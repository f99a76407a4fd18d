// rbd_kernels.hip -- batched floating-base rigid-body routines for the 18-body quadruped model (gfx950, fp64).
// SURVEY section 8(f) rows N2 (BASELINE configs[3]: full 18-DoF dynamics and its linearisation at every knot of every
// member) and N1 (the rows the kinodynamic refinement adds to a stage).
//
// What it computes follows
//   utilities_general/spatial_v2/dynamics/HandC.m:14-62                       tau = H(q) qdd + C(q, qd, f_ext)
//   utilities_general/dynamics-utilities/casadi_compatible_dynamics.m:12-143  the same recursion with foot forces (:53-60)
//   utilities_general/dynamics-utilities/get_forward_kin_foot.m:4-25          foot positions
//   utilities_general/dynamics-utilities/get_foot_jacobians_mc.m:12-24        closed-form leg Jacobians
//   optimizations/landing/main_scripts/landing_optimization.m:152-189         torque rows J_f'(-R' f), FK-consistency rows
// The reference evaluates them with 6 x 6 Pluecker matrices (and differentiates through CasADi).  Here every transform is
// kept in its compact form plux(E, r) (12 numbers), every rigid-body inertia as (m, h = m c, Ibar) (10 numbers) -- both
// closed under the operations of the recursions -- so one thread carries a whole 18-body evaluation in 5.8 KB of private
// memory; the linearisation is a central-difference sweep of the forward dynamics with one thread per (knot, column).
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

namespace landing {

constexpr int RB_NB = 18;
struct RbdModel {                    // host-built (landing-controller_amd/rbd.py from constants.py), resident in HBM
  int parent[RB_NB];                 // 1-based, 0 = fixed base
  int jtype[RB_NB];                  // 0 Rx, 1 Ry, 2 Rz, 3 Px, 4 Py, 5 Pz
  double E[RB_NB][9], r[RB_NB][3];   // Xtree = plux(E, r)
  double m[RB_NB], h[RB_NB][3], I[RB_NB][6];   // link inertia: mass, m*com, rotational inertia about the link origin (xx xy xz yy yz zz)
  int b_foot[4]; double foot_r[4][3];          // Xfoot = plux(1, foot_r) on body b_foot (1-based)
  double l1, l2, l3, l4;             // leg lengths of get_foot_jacobians_mc.m:5-8
};

struct V3d { double x, y, z; };
__device__ __forceinline__ V3d mk3(double x, double y, double z) { V3d v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3d add3(V3d a, V3d b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3d sub3(V3d a, V3d b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3d scl3(double s, V3d a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3d crs3(V3d a, V3d b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3d mul3(const double* E, V3d v) { return mk3(E[0] * v.x + E[1] * v.y + E[2] * v.z, E[3] * v.x + E[4] * v.y + E[5] * v.z, E[6] * v.x + E[7] * v.y + E[8] * v.z); }
__device__ __forceinline__ V3d mulT3(const double* E, V3d v) { return mk3(E[0] * v.x + E[3] * v.y + E[6] * v.z, E[1] * v.x + E[4] * v.y + E[7] * v.z, E[2] * v.x + E[5] * v.y + E[8] * v.z); }
__device__ __forceinline__ V3d sym3(const double* I, V3d v) { return mk3(I[0] * v.x + I[1] * v.y + I[2] * v.z, I[1] * v.x + I[3] * v.y + I[4] * v.z, I[2] * v.x + I[4] * v.y + I[5] * v.z); }
struct SV { V3d a, l; };             // spatial vector: angular / linear (motion) or moment / force

// joint transform applied on top of Xtree: Xup = XJ * plux(E, r)   (jcalc.m:22-40, plux.m)
// The axis is a template parameter (joint_xform below dispatches): with a run-time axis the rows of E are addressed through computed indices, and
// the callers' per-body E arrays then cannot live in registers (round 4: 8.9 KB -> 1.3 KB of scratch in the tangent kernel, 7.6 KB in the hyper-dual one).
template <int JT>
__device__ __forceinline__ void joint_xform_t(double q, const double* Et, const double* rt, double* E, double* r) {
  if (JT < 3) {
    double s, c; sincos(q, &s, &c);
    // rows of rx/ry/rz (coordinate transforms): rx = [1 0 0; 0 c s; 0 -s c], ry = [c 0 -s; 0 1 0; s 0 c], rz = [c s 0; -s c 0; 0 0 1]
    constexpr int a = JT < 3 ? JT : 0, b = (a + 1) % 3, d = (a + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      E[3 * a + j] = Et[3 * a + j];
      E[3 * b + j] = c * Et[3 * b + j] + s * Et[3 * d + j];
      E[3 * d + j] = -s * Et[3 * b + j] + c * Et[3 * d + j];
    }
    r[0] = rt[0]; r[1] = rt[1]; r[2] = rt[2];
  } else {   // xlt(q e_a) * plux(E, r) = plux(E, r + E' (q e_a))
    constexpr int a = JT >= 3 ? JT - 3 : 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) E[j] = Et[j];
    r[0] = rt[0] + Et[3 * a] * q; r[1] = rt[1] + Et[3 * a + 1] * q; r[2] = rt[2] + Et[3 * a + 2] * q;
  }
}
__device__ __forceinline__ void joint_xform(int jt, double q, const double* Et, const double* rt, double* E, double* r) {
  switch (jt) {
    case 0: joint_xform_t<0>(q, Et, rt, E, r); break;
    case 1: joint_xform_t<1>(q, Et, rt, E, r); break;
    case 2: joint_xform_t<2>(q, Et, rt, E, r); break;
    case 3: joint_xform_t<3>(q, Et, rt, E, r); break;
    case 4: joint_xform_t<4>(q, Et, rt, E, r); break;
    default: joint_xform_t<5>(q, Et, rt, E, r); break;
  }
}
__device__ __forceinline__ SV xmotion(const double* E, const double* r, SV v) {      // X v
  const V3d rr = mk3(r[0], r[1], r[2]);
  SV o; o.a = mul3(E, v.a); o.l = mul3(E, sub3(v.l, crs3(rr, v.a))); return o;
}
__device__ __forceinline__ SV xforceT(const double* E, const double* r, SV f) {      // X' f  (child -> parent)
  const V3d rr = mk3(r[0], r[1], r[2]);
  SV o; o.l = mulT3(E, f.l); o.a = add3(mulT3(E, f.a), crs3(rr, o.l)); return o;
}
__device__ __forceinline__ SV crm_mul(SV v, SV w) { SV o; o.a = crs3(v.a, w.a); o.l = add3(crs3(v.a, w.l), crs3(v.l, w.a)); return o; }
__device__ __forceinline__ SV crf_mul(SV v, SV f) { SV o; o.a = add3(crs3(v.a, f.a), crs3(v.l, f.l)); o.l = crs3(v.a, f.l); return o; }
__device__ __forceinline__ SV inertia_mul(double m, const double* h, const double* I, SV v) {
  const V3d hh = mk3(h[0], h[1], h[2]);
  SV o; o.a = add3(sym3(I, v.a), crs3(hh, v.l)); o.l = sub3(scl3(m, v.l), crs3(hh, v.a)); return o;
}
__device__ __forceinline__ double sdot(int jt, SV f) { return jt == 0 ? f.a.x : (jt == 1 ? f.a.y : (jt == 2 ? f.a.z : (jt == 3 ? f.l.x : (jt == 4 ? f.l.y : f.l.z)))); }
__device__ __forceinline__ SV sunit(int jt, double s) {
  SV o; o.a = mk3(jt == 0 ? s : 0.0, jt == 1 ? s : 0.0, jt == 2 ? s : 0.0); o.l = mk3(jt == 3 ? s : 0.0, jt == 4 ? s : 0.0, jt == 5 ? s : 0.0); return o;
}

// H (row-major 18 x 18, may be null) and C (18) for one configuration; f_foot: 12 world-frame foot forces or null
__device__ void hand_c(const RbdModel& M, const double* q, const double* qd, const double* f_foot, double* H, double* C) {
  double E[RB_NB][9], r[RB_NB][3];
  SV v[RB_NB], fvp[RB_NB];
  double E0[9], r0[3];                       // transform from the world to the current chain body (only the 4 foot bodies need it)
  double E0f[4][9], r0f[4][3];
  {
    SV avp[RB_NB];
    for (int i = 0; i < RB_NB; ++i) {
      joint_xform(M.jtype[i], q[i], M.E[i], M.r[i], E[i], r[i]);
      const SV vJ = sunit(M.jtype[i], qd[i]);
      const int pa = M.parent[i];
      if (pa == 0) {
        SV g; g.a = mk3(0, 0, 0); g.l = mk3(0, 0, 9.81);      // -a_grav
        v[i] = vJ; avp[i] = xmotion(E[i], r[i], g);
      } else {
        const SV vp = xmotion(E[i], r[i], v[pa - 1]);
        v[i].a = add3(vp.a, vJ.a); v[i].l = add3(vp.l, vJ.l);
        const SV ap = xmotion(E[i], r[i], avp[pa - 1]), cv = crm_mul(v[i], vJ);
        avp[i].a = add3(ap.a, cv.a); avp[i].l = add3(ap.l, cv.l);
      }
      const SV Ia = inertia_mul(M.m[i], M.h[i], M.I[i], avp[i]), Iv = inertia_mul(M.m[i], M.h[i], M.I[i], v[i]), cf = crf_mul(v[i], Iv);
      fvp[i].a = add3(Ia.a, cf.a); fvp[i].l = add3(Ia.l, cf.l);
    }
  }
  if (f_foot) {   // external forces at the feet, casadi_compatible_dynamics.m:53-60: fvp -= X0^{-T} f_world
    // world transforms of the four foot bodies: the base chain (bodies 1..6) then the leg (3 bodies)
    for (int j = 0; j < 9; ++j) E0[j] = (j % 4 == 0) ? 1.0 : 0.0;
    r0[0] = r0[1] = r0[2] = 0.0;
    auto compose = [](const double* Eu, const double* ru, double* Ea, double* ra) {   // (Ea, ra) <- plux(Eu, ru) * plux(Ea, ra)
      const V3d t = mulT3(Ea, mk3(ru[0], ru[1], ru[2]));
      double En[9];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) En[3 * a + b] = Eu[3 * a] * Ea[b] + Eu[3 * a + 1] * Ea[3 + b] + Eu[3 * a + 2] * Ea[6 + b];
      for (int j = 0; j < 9; ++j) Ea[j] = En[j];
      ra[0] += t.x; ra[1] += t.y; ra[2] += t.z;
    };
    for (int i = 0; i < 6; ++i) compose(E[i], r[i], E0, r0);
    for (int leg = 0; leg < 4; ++leg) {
      for (int j = 0; j < 9; ++j) E0f[leg][j] = E0[j];
      for (int j = 0; j < 3; ++j) r0f[leg][j] = r0[j];
      const int jb = M.b_foot[leg] - 1;
      for (int i = jb - 2; i <= jb; ++i) compose(E[i], r[i], E0f[leg], r0f[leg]);
      // foot point in world coordinates, spatial force about the world origin, moved into body coordinates: X0^* f = [E0 (n - r0 x f); E0 f]
      const V3d pf = add3(mk3(r0f[leg][0], r0f[leg][1], r0f[leg][2]), mulT3(E0f[leg], mk3(M.foot_r[leg][0], M.foot_r[leg][1], M.foot_r[leg][2])));
      const V3d fw = mk3(f_foot[3 * leg], f_foot[3 * leg + 1], f_foot[3 * leg + 2]);
      const V3d nb = crs3(sub3(pf, mk3(r0f[leg][0], r0f[leg][1], r0f[leg][2])), fw);
      fvp[jb].a = sub3(fvp[jb].a, mul3(E0f[leg], nb)); fvp[jb].l = sub3(fvp[jb].l, mul3(E0f[leg], fw));
    }
  }
  for (int i = RB_NB - 1; i >= 0; --i) {
    C[i] = sdot(M.jtype[i], fvp[i]);
    const int pa = M.parent[i];
    if (pa != 0) { const SV t = xforceT(E[i], r[i], fvp[i]); fvp[pa - 1].a = add3(fvp[pa - 1].a, t.a); fvp[pa - 1].l = add3(fvp[pa - 1].l, t.l); }
  }
  if (!H) return;
  // composite rigid-body inertias in (m, h, Ibar) form: parent += X' I X with X = plux(E, r):
  //   m' = m, h' = E'h + m r, Ibar' = E' Ibar E - skew(r) skew(E'h) - skew(E'h + m r) skew(r)
  double cm[RB_NB], ch[RB_NB][3], cI[RB_NB][6];
  for (int i = 0; i < RB_NB; ++i) { cm[i] = M.m[i]; for (int j = 0; j < 3; ++j) ch[i][j] = M.h[i][j]; for (int j = 0; j < 6; ++j) cI[i][j] = M.I[i][j]; }
  for (int i = RB_NB - 1; i >= 0; --i) {
    const int pa = M.parent[i];
    if (pa == 0) continue;
    const double* Ei = E[i];
    const V3d hp = mulT3(Ei, mk3(ch[i][0], ch[i][1], ch[i][2])), rr = mk3(r[i][0], r[i][1], r[i][2]);
    const V3d hn = add3(hp, scl3(cm[i], rr));
    // E' Ibar E (symmetric)
    double T[9];
    { const double* I6 = cI[i];
      const double Is[9] = {I6[0], I6[1], I6[2], I6[1], I6[3], I6[4], I6[2], I6[4], I6[5]};
      double A[9];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[3 * a + b] = Is[3 * a] * Ei[b] + Is[3 * a + 1] * Ei[3 + b] + Is[3 * a + 2] * Ei[6 + b];     // Ibar E
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) T[3 * a + b] = Ei[a] * A[b] + Ei[3 + a] * A[3 + b] + Ei[6 + a] * A[6 + b];                   // E' (Ibar E)
    }
    // - skew(r) skew(hp) - skew(hn) skew(r):  skew(a) skew(b) = b a' - (a.b) 1
    auto add_ss = [&](V3d a, V3d b, double sgn) {
      const double d = a.x * b.x + a.y * b.y + a.z * b.z;
      const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
      for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) T[3 * x + y] += sgn * (bv[x] * av[y] - (x == y ? d : 0.0));
    };
    add_ss(rr, hp, -1.0); add_ss(hn, rr, -1.0);
    double* Ip = cI[pa - 1];
    Ip[0] += T[0]; Ip[1] += 0.5 * (T[1] + T[3]); Ip[2] += 0.5 * (T[2] + T[6]); Ip[3] += T[4]; Ip[4] += 0.5 * (T[5] + T[7]); Ip[5] += T[8];
    cm[pa - 1] += cm[i]; ch[pa - 1][0] += hn.x; ch[pa - 1][1] += hn.y; ch[pa - 1][2] += hn.z;
  }
  for (int i = 0; i < RB_NB * RB_NB; ++i) H[i] = 0.0;
  for (int i = 0; i < RB_NB; ++i) {
    SV fh = inertia_mul(cm[i], ch[i], cI[i], sunit(M.jtype[i], 1.0));
    H[i * RB_NB + i] = sdot(M.jtype[i], fh);
    int j = i;
    while (M.parent[j] > 0) {
      fh = xforceT(E[j], r[j], fh);
      j = M.parent[j] - 1;
      const double hij = sdot(M.jtype[j], fh);
      H[i * RB_NB + j] = hij; H[j * RB_NB + i] = hij;
    }
  }
}

// in-place Cholesky solve of the 18 x 18 system H x = b (H destroyed); returns false if H is not positive definite
__device__ bool chol_solve18(double* H, double* b) {
  for (int j = 0; j < RB_NB; ++j) {
    double d = H[j * RB_NB + j];
    for (int k = 0; k < j; ++k) d -= H[j * RB_NB + k] * H[j * RB_NB + k];
    if (!(d > 0.0)) return false;
    d = sqrt(d); H[j * RB_NB + j] = d;
    for (int i = j + 1; i < RB_NB; ++i) {
      double s = H[i * RB_NB + j];
      for (int k = 0; k < j; ++k) s -= H[i * RB_NB + k] * H[j * RB_NB + k];
      H[i * RB_NB + j] = s / d;
    }
  }
  for (int i = 0; i < RB_NB; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= H[i * RB_NB + k] * b[k]; b[i] = s / H[i * RB_NB + i]; }
  for (int i = RB_NB - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < RB_NB; ++k) s -= H[k * RB_NB + i] * b[k]; b[i] = s / H[i * RB_NB + i]; }
  return true;
}

struct FbArgs {
  const RbdModel* model; int npts;
  const double* q; const double* qd; const double* tau; const double* f_foot;   // [npts][18] x3, [npts][12] or null
  double* H; double* C; double* qdd; double* A; double* Hinv;                  // [npts][324], [npts][18], [npts][18], [npts][18*36], [npts][324]; any may be null
  double fd_h;
  int arrow;      // the model has the quadruped topology rnea_tangent_quad is written for (landing_rbd_set_model)
};

// one thread per knot: H, C (and qdd when tau is given)
__global__ void __launch_bounds__(64) landing_fb_hc_kernel(FbArgs a) {
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= a.npts) return;
  const RbdModel& M = *a.model;
  double H[RB_NB * RB_NB], C[RB_NB];
  hand_c(M, a.q + (size_t)pt * RB_NB, a.qd + (size_t)pt * RB_NB, a.f_foot ? a.f_foot + (size_t)pt * 12 : nullptr, H, C);
  if (a.H) for (int i = 0; i < RB_NB * RB_NB; ++i) a.H[(size_t)pt * RB_NB * RB_NB + i] = H[i];
  if (a.C) for (int i = 0; i < RB_NB; ++i) a.C[(size_t)pt * RB_NB + i] = C[i];
  if (a.qdd && a.tau) {
    double b[RB_NB];
    for (int i = 0; i < RB_NB; ++i) b[i] = a.tau[(size_t)pt * RB_NB + i] - C[i];
    const bool ok = chol_solve18(H, b);
    for (int i = 0; i < RB_NB; ++i) a.qdd[(size_t)pt * RB_NB + i] = ok ? b[i] : NAN;
  }
}

// linearisation of the forward dynamics qdd(q, qd, tau) = H^-1 (tau - C): one thread per (knot, column).
// columns 0..35: d qdd / d [q; qd] by central differences with step fd_h (the reference differentiates the same recursion
// through CasADi); columns 36..53: column of H^-1 = d qdd / d tau.
__global__ void __launch_bounds__(64) landing_fb_lin_kernel(FbArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int pt = idx / 54, col = idx % 54;
  if (pt >= a.npts) return;
  const RbdModel& M = *a.model;
  double q[RB_NB], qd[RB_NB], H[RB_NB * RB_NB], C[RB_NB], bp[RB_NB], bm[RB_NB];
  for (int i = 0; i < RB_NB; ++i) { q[i] = a.q[(size_t)pt * RB_NB + i]; qd[i] = a.qd[(size_t)pt * RB_NB + i]; }
  const double* ff = a.f_foot ? a.f_foot + (size_t)pt * 12 : nullptr;
  if (col >= 36) {
    if (!a.Hinv) return;
    hand_c(M, q, qd, ff, H, C);
    for (int i = 0; i < RB_NB; ++i) bp[i] = (i == col - 36) ? 1.0 : 0.0;
    const bool ok = chol_solve18(H, bp);
    for (int i = 0; i < RB_NB; ++i) a.Hinv[((size_t)pt * RB_NB + i) * RB_NB + (col - 36)] = ok ? bp[i] : NAN;
    return;
  }
  if (!a.A) return;
  double* var = col < RB_NB ? q : qd;
  const int j = col % RB_NB;
  const double x0 = var[j];
  bool ok = true;
  var[j] = x0 + a.fd_h;
  hand_c(M, q, qd, ff, H, C);
  for (int i = 0; i < RB_NB; ++i) bp[i] = a.tau[(size_t)pt * RB_NB + i] - C[i];
  ok = chol_solve18(H, bp) && ok;
  var[j] = x0 - a.fd_h;
  hand_c(M, q, qd, ff, H, C);
  for (int i = 0; i < RB_NB; ++i) bm[i] = a.tau[(size_t)pt * RB_NB + i] - C[i];
  ok = chol_solve18(H, bm) && ok;
  for (int i = 0; i < RB_NB; ++i) a.A[((size_t)pt * RB_NB + i) * 36 + col] = ok ? (bp[i] - bm[i]) / (2.0 * a.fd_h) : NAN;
}

// ---- exact linearisation (forward mode) ------------------------------------------------------------------------------------------
// d qdd / d z = -H^-1 d ID(q, qd, qdd, f) / d z at fixed qdd, z in [q; qd], with ID = H qdd + C the inverse dynamics (the recursion of
// hand_c with the joint accelerations S qdd added, ID.m:20-40).  The reference gets these derivatives from CasADi's algorithmic
// differentiation of the same recursion (casadi_compatible_dynamics.m is called on SX symbols); here every thread pushes ONE tangent
// direction through the recursion with dual numbers (value, derivative) -- exact to rounding, one O(n) pass per column instead of the
// two O(n^2..n^3) forward-dynamics evaluations of the central-difference kernel above.
struct Dual { double v, d; };
__device__ __forceinline__ Dual D_(double v, double d = 0.0) { Dual o; o.v = v; o.d = d; return o; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return D_(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return D_(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return D_(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return D_(a.v * b.v, fma(a.v, b.d, a.d * b.v)); }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return D_(a * b.v, a * b.d); }
struct V3D { Dual x, y, z; };
__device__ __forceinline__ V3D mk3D(Dual x, Dual y, Dual z) { V3D v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3D mk3D(double x, double y, double z) { return mk3D(D_(x), D_(y), D_(z)); }
__device__ __forceinline__ V3D add3(V3D a, V3D b) { return mk3D(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3D sub3(V3D a, V3D b) { return mk3D(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3D scl3(double s, V3D a) { return mk3D(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3D crs3(V3D a, V3D b) { return mk3D(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3D mul3(const Dual* E, V3D v) { return mk3D(E[0] * v.x + E[1] * v.y + E[2] * v.z, E[3] * v.x + E[4] * v.y + E[5] * v.z, E[6] * v.x + E[7] * v.y + E[8] * v.z); }
__device__ __forceinline__ V3D mulT3(const Dual* E, V3D v) { return mk3D(E[0] * v.x + E[3] * v.y + E[6] * v.z, E[1] * v.x + E[4] * v.y + E[7] * v.z, E[2] * v.x + E[5] * v.y + E[8] * v.z); }
__device__ __forceinline__ V3D sym3(const double* I, V3D v) { return mk3D(I[0] * v.x + I[1] * v.y + I[2] * v.z, I[1] * v.x + I[3] * v.y + I[4] * v.z, I[2] * v.x + I[4] * v.y + I[5] * v.z); }
struct SVD { V3D a, l; };
__device__ __forceinline__ SVD addS(SVD p, SVD q) { SVD o; o.a = add3(p.a, q.a); o.l = add3(p.l, q.l); return o; }
// (the axis is a template parameter: with a run-time axis the rows of E are addressed through computed indices and the callers' E arrays
// cannot live in registers)
template <int JT>
__device__ __forceinline__ void joint_xform_t(Dual q, const double* Et, const double* rt, Dual* E, Dual* r) {
  if (JT < 3) {
    double s, c; sincos(q.v, &s, &c);
    const Dual S = D_(s, c * q.d), Cc = D_(c, -s * q.d);
    constexpr int a = JT < 3 ? JT : 0, b = (a + 1) % 3, d = (a + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      E[3 * a + j] = D_(Et[3 * a + j]);
      E[3 * b + j] = Et[3 * b + j] * Cc + Et[3 * d + j] * S;
      E[3 * d + j] = Et[3 * d + j] * Cc - Et[3 * b + j] * S;
    }
    r[0] = D_(rt[0]); r[1] = D_(rt[1]); r[2] = D_(rt[2]);
  } else {
    constexpr int a = JT >= 3 ? JT - 3 : 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) E[j] = D_(Et[j]);
    r[0] = D_(rt[0]) + Et[3 * a] * q; r[1] = D_(rt[1]) + Et[3 * a + 1] * q; r[2] = D_(rt[2]) + Et[3 * a + 2] * q;
  }
}
__device__ __forceinline__ void joint_xform(int jt, Dual q, const double* Et, const double* rt, Dual* E, Dual* r) {
  switch (jt) {
    case 0: joint_xform_t<0>(q, Et, rt, E, r); break;
    case 1: joint_xform_t<1>(q, Et, rt, E, r); break;
    case 2: joint_xform_t<2>(q, Et, rt, E, r); break;
    case 3: joint_xform_t<3>(q, Et, rt, E, r); break;
    case 4: joint_xform_t<4>(q, Et, rt, E, r); break;
    default: joint_xform_t<5>(q, Et, rt, E, r); break;
  }
}
__device__ __forceinline__ SVD xmotion(const Dual* E, const Dual* r, SVD v) {
  const V3D rr = mk3D(r[0], r[1], r[2]);
  SVD o; o.a = mul3(E, v.a); o.l = mul3(E, sub3(v.l, crs3(rr, v.a))); return o;
}
__device__ __forceinline__ SVD xforceT(const Dual* E, const Dual* r, SVD f) {
  const V3D rr = mk3D(r[0], r[1], r[2]);
  SVD o; o.l = mulT3(E, f.l); o.a = add3(mulT3(E, f.a), crs3(rr, o.l)); return o;
}
__device__ __forceinline__ SVD crm_mul(SVD v, SVD w) { SVD o; o.a = crs3(v.a, w.a); o.l = add3(crs3(v.a, w.l), crs3(v.l, w.a)); return o; }
__device__ __forceinline__ SVD crf_mul(SVD v, SVD f) { SVD o; o.a = add3(crs3(v.a, f.a), crs3(v.l, f.l)); o.l = crs3(v.a, f.l); return o; }
__device__ __forceinline__ SVD inertia_mul(double m, const double* h, const double* I, SVD v) {
  const V3D hh = mk3D(h[0], h[1], h[2]);
  SVD o; o.a = add3(sym3(I, v.a), crs3(hh, v.l)); o.l = sub3(scl3(m, v.l), crs3(hh, v.a)); return o;
}
__device__ __forceinline__ Dual sdot(int jt, SVD f) { return jt == 0 ? f.a.x : (jt == 1 ? f.a.y : (jt == 2 ? f.a.z : (jt == 3 ? f.l.x : (jt == 4 ? f.l.y : f.l.z)))); }
__device__ __forceinline__ SVD sunit(int jt, Dual s) {
  const Dual z = D_(0.0);
  SVD o; o.a = mk3D(jt == 0 ? s : z, jt == 1 ? s : z, jt == 2 ? s : z); o.l = mk3D(jt == 3 ? s : z, jt == 4 ? s : z, jt == 5 ? s : z); return o;
}
// derivative of tau = ID(q, qd, qdd, f_foot) along the direction (dir < 18: q_dir, else qd_{dir-18}); dtau[18]
__device__ void rnea_tangent(const RbdModel& M, const double* q, const double* qd, const double* qdd, const double* f_foot, int dir, double* dtau) {
  Dual E[RB_NB][9], r[RB_NB][3];
  SVD v[RB_NB], fvp[RB_NB], avp[RB_NB];
  for (int i = 0; i < RB_NB; ++i) {
    const Dual qi = D_(q[i], dir == i ? 1.0 : 0.0), qdi = D_(qd[i], dir == RB_NB + i ? 1.0 : 0.0);
    joint_xform(M.jtype[i], qi, M.E[i], M.r[i], E[i], r[i]);
    const SVD vJ = sunit(M.jtype[i], qdi), aJ = sunit(M.jtype[i], D_(qdd[i]));
    const int pa = M.parent[i];
    if (pa == 0) {
      SVD g; g.a = mk3D(0.0, 0.0, 0.0); g.l = mk3D(0.0, 0.0, 9.81);
      v[i] = vJ; avp[i] = addS(xmotion(E[i], r[i], g), aJ);
    } else {
      v[i] = addS(xmotion(E[i], r[i], v[pa - 1]), vJ);
      avp[i] = addS(addS(xmotion(E[i], r[i], avp[pa - 1]), crm_mul(v[i], vJ)), aJ);
    }
    fvp[i] = addS(inertia_mul(M.m[i], M.h[i], M.I[i], avp[i]), crf_mul(v[i], inertia_mul(M.m[i], M.h[i], M.I[i], v[i])));
  }
  if (f_foot) {
    Dual E0[9], r0[3];
    for (int j = 0; j < 9; ++j) E0[j] = D_((j % 4 == 0) ? 1.0 : 0.0);
    r0[0] = r0[1] = r0[2] = D_(0.0);
    auto compose = [](const Dual* Eu, const Dual* ru, Dual* Ea, Dual* ra) {
      const V3D t = mulT3(Ea, mk3D(ru[0], ru[1], ru[2]));
      Dual En[9];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) En[3 * a + b] = Eu[3 * a] * Ea[b] + Eu[3 * a + 1] * Ea[3 + b] + Eu[3 * a + 2] * Ea[6 + b];
      for (int j = 0; j < 9; ++j) Ea[j] = En[j];
      ra[0] = ra[0] + t.x; ra[1] = ra[1] + t.y; ra[2] = ra[2] + t.z;
    };
    for (int i = 0; i < 6; ++i) compose(E[i], r[i], E0, r0);
    for (int leg = 0; leg < 4; ++leg) {
      Dual Ef[9], rf[3];
      for (int j = 0; j < 9; ++j) Ef[j] = E0[j];
      for (int j = 0; j < 3; ++j) rf[j] = r0[j];
      const int jb = M.b_foot[leg] - 1;
      for (int i = jb - 2; i <= jb; ++i) compose(E[i], r[i], Ef, rf);
      const V3D rb = mk3D(rf[0], rf[1], rf[2]);
      const V3D pf = add3(rb, mulT3(Ef, mk3D(M.foot_r[leg][0], M.foot_r[leg][1], M.foot_r[leg][2])));
      const V3D fw = mk3D(f_foot[3 * leg], f_foot[3 * leg + 1], f_foot[3 * leg + 2]);
      const V3D nb = crs3(sub3(pf, rb), fw);
      fvp[jb].a = sub3(fvp[jb].a, mul3(Ef, nb)); fvp[jb].l = sub3(fvp[jb].l, mul3(Ef, fw));
    }
  }
  for (int i = RB_NB - 1; i >= 0; --i) {
    dtau[i] = sdot(M.jtype[i], fvp[i]).d;
    const int pa = M.parent[i];
    if (pa != 0) fvp[pa - 1] = addS(fvp[pa - 1], xforceT(E[i], r[i], fvp[i]));
  }
}
// The same tangent for the reference's quadruped topology (landing_rbd_set_model checks it: six single-DoF base joints in a chain, bodies 0..5,
// and four legs of three joints hanging off body 5 -- parent = [0 1 2 3 4 5 | 6 7 8 | 6 10 11 | 6 13 14 | 6 16 17]).  The generic recursion above
// indexes its per-body arrays with the model's parent table: 1080 doubles of dual numbers per thread in private memory (8.9 KB of scratch, the
// kernel's whole time).  Here every index is a compile-time constant and a leg is finished -- forward, foot force, backward into body 5 -- before
// the next one starts; the base chain's transforms are formed again on the way back instead of being kept.  Same operations in the same order per
// body as rnea_tangent: the results agree to rounding.
__device__ void rnea_tangent_quad(const RbdModel& M, const double* q, const double* qd, const double* qdd, const double* f_foot, int dir, double* dtau) {
  SVD fb[6], v, a;
  Dual E0[9], r0[3];
#pragma unroll
  for (int j = 0; j < 9; ++j) E0[j] = D_((j % 4 == 0) ? 1.0 : 0.0);
  r0[0] = r0[1] = r0[2] = D_(0.0);
  auto compose = [](const Dual* Eu, const Dual* ru, Dual* Ea, Dual* ra) {
    const V3D t = mulT3(Ea, mk3D(ru[0], ru[1], ru[2]));
    Dual En[9];
#pragma unroll
    for (int aa = 0; aa < 3; ++aa)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) En[3 * aa + bb] = Eu[3 * aa] * Ea[bb] + Eu[3 * aa + 1] * Ea[3 + bb] + Eu[3 * aa + 2] * Ea[6 + bb];
#pragma unroll
    for (int j = 0; j < 9; ++j) Ea[j] = En[j];
    ra[0] = ra[0] + t.x; ra[1] = ra[1] + t.y; ra[2] = ra[2] + t.z;
  };
  // ---- base chain, forward
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    Dual E[9], r[3];
    const Dual qi = D_(q[i], dir == i ? 1.0 : 0.0), qdi = D_(qd[i], dir == RB_NB + i ? 1.0 : 0.0);
    joint_xform(M.jtype[i], qi, M.E[i], M.r[i], E, r);
    const SVD vJ = sunit(M.jtype[i], qdi), aJ = sunit(M.jtype[i], D_(qdd[i]));
    if (i == 0) {
      SVD g; g.a = mk3D(0.0, 0.0, 0.0); g.l = mk3D(0.0, 0.0, 9.81);
      v = vJ; a = addS(xmotion(E, r, g), aJ);
    } else {
      const SVD vn = addS(xmotion(E, r, v), vJ);
      a = addS(addS(xmotion(E, r, a), crm_mul(vn, vJ)), aJ);
      v = vn;
    }
    fb[i] = addS(inertia_mul(M.m[i], M.h[i], M.I[i], a), crf_mul(v, inertia_mul(M.m[i], M.h[i], M.I[i], v)));
    if (f_foot) compose(E, r, E0, r0);
  }
  // ---- the four legs, one after the other
#pragma unroll 1
  for (int leg = 0; leg < 4; ++leg) {
    Dual El[3][9], rl[3][3];
    SVD fl[3], vv = v, aa = a;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = 6 + 3 * leg + j;
      const Dual qi = D_(q[i], dir == i ? 1.0 : 0.0), qdi = D_(qd[i], dir == RB_NB + i ? 1.0 : 0.0);
      joint_xform(M.jtype[i], qi, M.E[i], M.r[i], El[j], rl[j]);
      const SVD vJ = sunit(M.jtype[i], qdi), aJ = sunit(M.jtype[i], D_(qdd[i]));
      const SVD vn = addS(xmotion(El[j], rl[j], vv), vJ);
      aa = addS(addS(xmotion(El[j], rl[j], aa), crm_mul(vn, vJ)), aJ);
      vv = vn;
      fl[j] = addS(inertia_mul(M.m[i], M.h[i], M.I[i], aa), crf_mul(vv, inertia_mul(M.m[i], M.h[i], M.I[i], vv)));
    }
    if (f_foot) {
      Dual Ef[9], rf[3];
#pragma unroll
      for (int j = 0; j < 9; ++j) Ef[j] = E0[j];
#pragma unroll
      for (int j = 0; j < 3; ++j) rf[j] = r0[j];
#pragma unroll
      for (int j = 0; j < 3; ++j) compose(El[j], rl[j], Ef, rf);
      const V3D rb = mk3D(rf[0], rf[1], rf[2]);
      const V3D pf = add3(rb, mulT3(Ef, mk3D(M.foot_r[leg][0], M.foot_r[leg][1], M.foot_r[leg][2])));
      const V3D fw = mk3D(f_foot[3 * leg], f_foot[3 * leg + 1], f_foot[3 * leg + 2]);
      const V3D nb = crs3(sub3(pf, rb), fw);
      fl[2].a = sub3(fl[2].a, mul3(Ef, nb)); fl[2].l = sub3(fl[2].l, mul3(Ef, fw));
    }
#pragma unroll
    for (int j = 2; j >= 0; --j) {
      const int i = 6 + 3 * leg + j;
      dtau[i] = sdot(M.jtype[i], fl[j]).d;
      if (j > 0) fl[j - 1] = addS(fl[j - 1], xforceT(El[j], rl[j], fl[j]));
      else fb[5] = addS(fb[5], xforceT(El[0], rl[0], fl[0]));
    }
  }
  // ---- base chain, backward (transforms formed again)
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    dtau[i] = sdot(M.jtype[i], fb[i]).d;
    if (i > 0) {
      Dual E[9], r[3];
      joint_xform(M.jtype[i], D_(q[i], dir == i ? 1.0 : 0.0), M.E[i], M.r[i], E, r);
      fb[i - 1] = addS(fb[i - 1], xforceT(E, r, fb[i]));
    }
  }
}
// pass 1 (one thread per knot): H, C, qdd and H^-1 (scratch or caller buffers); pass 2 (one thread per (knot, column)): A(:, col) = -H^-1 dtau
__global__ void __launch_bounds__(64) landing_fb_lin_exact_prep_kernel(FbArgs a, double* qdd_out, double* hinv_out) {
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= a.npts) return;
  const RbdModel& M = *a.model;
  double H[RB_NB * RB_NB], C[RB_NB], b[RB_NB];
  hand_c(M, a.q + (size_t)pt * RB_NB, a.qd + (size_t)pt * RB_NB, a.f_foot ? a.f_foot + (size_t)pt * 12 : nullptr, H, C);
  for (int i = 0; i < RB_NB; ++i) b[i] = a.tau[(size_t)pt * RB_NB + i] - C[i];
  bool ok = true;     // Cholesky factor in place, then qdd and the 18 columns of the inverse by substitution
  for (int j = 0; j < RB_NB && ok; ++j) {
    double d = H[j * RB_NB + j];
    for (int k = 0; k < j; ++k) d -= H[j * RB_NB + k] * H[j * RB_NB + k];
    if (!(d > 0.0)) { ok = false; break; }
    d = sqrt(d); H[j * RB_NB + j] = d;
    for (int i = j + 1; i < RB_NB; ++i) {
      double s = H[i * RB_NB + j];
      for (int k = 0; k < j; ++k) s -= H[i * RB_NB + k] * H[j * RB_NB + k];
      H[i * RB_NB + j] = s / d;
    }
  }
  auto subst = [&](double* x) {
    for (int i = 0; i < RB_NB; ++i) { double s = x[i]; for (int k = 0; k < i; ++k) s -= H[i * RB_NB + k] * x[k]; x[i] = s / H[i * RB_NB + i]; }
    for (int i = RB_NB - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < RB_NB; ++k) s -= H[k * RB_NB + i] * x[k]; x[i] = s / H[i * RB_NB + i]; }
  };
  subst(b);
  for (int i = 0; i < RB_NB; ++i) qdd_out[(size_t)pt * RB_NB + i] = ok ? b[i] : NAN;
  for (int c = 0; c < RB_NB; ++c) {
    double e[RB_NB];
    for (int i = 0; i < RB_NB; ++i) e[i] = (i == c) ? 1.0 : 0.0;
    subst(e);
    for (int i = 0; i < RB_NB; ++i) hinv_out[((size_t)pt * RB_NB + i) * RB_NB + c] = ok ? e[i] : NAN;
  }
}
template <bool QUAD>      // QUAD: the model has the topology of rnea_tangent_quad (FbArgs::arrow); its own instantiation, so that it does not inherit the generic recursion's scratch
#ifndef LANDING_FBLIN_WAVES
#define LANDING_FBLIN_WAVES 1
#endif
__global__ void __launch_bounds__(64, QUAD ? LANDING_FBLIN_WAVES : 1) landing_fb_lin_exact_kernel(FbArgs a, const double* qdd_in, const double* hinv_in) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int pt = idx / 36, col = idx % 36;
  if (pt >= a.npts) return;
  const RbdModel& M = *a.model;
  double dtau[RB_NB];
  if (QUAD) rnea_tangent_quad(M, a.q + (size_t)pt * RB_NB, a.qd + (size_t)pt * RB_NB, qdd_in + (size_t)pt * RB_NB, a.f_foot ? a.f_foot + (size_t)pt * 12 : nullptr, col, dtau);
  else rnea_tangent(M, a.q + (size_t)pt * RB_NB, a.qd + (size_t)pt * RB_NB, qdd_in + (size_t)pt * RB_NB, a.f_foot ? a.f_foot + (size_t)pt * 12 : nullptr, col, dtau);
  const double* Hi = hinv_in + (size_t)pt * RB_NB * RB_NB;
  for (int i = 0; i < RB_NB; ++i) {
    double s = 0.0;
    for (int k = 0; k < RB_NB; ++k) s -= Hi[i * RB_NB + k] * dtau[k];
    a.A[((size_t)pt * RB_NB + i) * 36 + col] = s;
  }
}

// rows the kinodynamic refinement adds per stage (landing_optimization.m:152-189), one thread per (member, stage):
// foot positions by forward kinematics of [q6; jpos] (get_forward_kin_foot.m), FK consistency c - FK, leg torques
// tau = J_f' (-R_world_to_body f) with the closed-form Jacobian of get_foot_jacobians_mc.m:12-24 and rpyToRotMat_xyz.m:2
struct KdArgs { const RbdModel* model; int npts; const double* q6; const double* c; const double* f; const double* jpos; double* fk; double* fk_err; double* tau; };
__global__ void __launch_bounds__(64) landing_kinodyn_rows_kernel(KdArgs a) {
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= a.npts) return;
  const RbdModel& M = *a.model;
  const double* q6 = a.q6 + (size_t)pt * 6; const double* jp = a.jpos + (size_t)pt * 12;
  double E0[9], r0[3];
  for (int j = 0; j < 9; ++j) E0[j] = (j % 4 == 0) ? 1.0 : 0.0;
  r0[0] = r0[1] = r0[2] = 0.0;
  auto compose = [](const double* Eu, const double* ru, double* Ea, double* ra) {
    const V3d t = mulT3(Ea, mk3(ru[0], ru[1], ru[2]));
    double En[9];
    for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) En[3 * x + y] = Eu[3 * x] * Ea[y] + Eu[3 * x + 1] * Ea[3 + y] + Eu[3 * x + 2] * Ea[6 + y];
    for (int j = 0; j < 9; ++j) Ea[j] = En[j];
    ra[0] += t.x; ra[1] += t.y; ra[2] += t.z;
  };
  double E[9], r[3];
  for (int i = 0; i < 6; ++i) { joint_xform(M.jtype[i], q6[i], M.E[i], M.r[i], E, r); compose(E, r, E0, r0); }
  // R_world_to_body = E0 (the coordinate transform world -> body accumulated by the chain = (rx' ry' rz')')
  for (int leg = 0; leg < 4; ++leg) {
    double El[9], rl[3];
    for (int j = 0; j < 9; ++j) El[j] = E0[j];
    for (int j = 0; j < 3; ++j) rl[j] = r0[j];
    const int jb = M.b_foot[leg] - 1;
    for (int i = jb - 2; i <= jb; ++i) { joint_xform(M.jtype[i], jp[3 * leg + (i - (jb - 2))], M.E[i], M.r[i], E, r); compose(E, r, El, rl); }
    const V3d pf = add3(mk3(rl[0], rl[1], rl[2]), mulT3(El, mk3(M.foot_r[leg][0], M.foot_r[leg][1], M.foot_r[leg][2])));
    const double pfv[3] = {pf.x, pf.y, pf.z};
    for (int j = 0; j < 3; ++j) {
      if (a.fk) a.fk[(size_t)pt * 12 + 3 * leg + j] = pfv[j];
      if (a.fk_err) a.fk_err[(size_t)pt * 12 + 3 * leg + j] = a.c[(size_t)pt * 12 + 3 * leg + j] - pfv[j];
    }
    if (a.tau) {
      const double ss = (leg & 1) ? 1.0 : -1.0;            // sideSign = [-1, 1, -1, 1]
      double s1, c1, s2, c2, s3, c3;
      sincos(jp[3 * leg], &s1, &c1); sincos(jp[3 * leg + 1], &s2, &c2); sincos(jp[3 * leg + 2], &s3, &c3);
      const double c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3, l14 = M.l1 + M.l4;
      const double J[3][3] = {{0.0, M.l3 * c23 + M.l2 * c2, M.l3 * c23},
                              {M.l3 * c1 * c23 + M.l2 * c1 * c2 - l14 * s1 * ss, -M.l3 * s1 * s23 - M.l2 * s1 * s2, -M.l3 * s1 * s23},
                              {M.l3 * s1 * c23 + M.l2 * c2 * s1 + l14 * ss * c1, M.l3 * c1 * s23 + M.l2 * c1 * s2, M.l3 * c1 * s23}};
      const V3d fb = mul3(E0, mk3(-a.f[(size_t)pt * 12 + 3 * leg], -a.f[(size_t)pt * 12 + 3 * leg + 1], -a.f[(size_t)pt * 12 + 3 * leg + 2]));
      for (int j = 0; j < 3; ++j) a.tau[(size_t)pt * 12 + 3 * leg + j] = J[0][j] * fb.x + J[1][j] * fb.y + J[2][j] * fb.z;
    }
  }
}


// ---- function layer of the kinodynamic refinement NLP ------------------------------------------------------------
// g(x) and its Jacobian for the NLP of optimizations/landing/main_scripts/landing_optimization.m:38-189 (N+1 knots, N intervals):
//   x = [X(:) (12 x (N+1): pos, rpy, omega_body, v_world); jpos(:) (12 x N); U(:) (24 x N: c; f_grf)]   -- the script's declaration order (:39-42)
//   g = [q(:,1); qdot(:,1); c(:,1)  (24, :89-91) | q(:,N) twice, qdot(:,N) twice (24, :94-97) | per interval k the rows of :113-189 in the
//        script's order: v / omega / pos / rpy Euler defects (12), f_z (4), per leg [c_z, f_z c_z, f_z (c+ - c) twice (k < N-1), p_rel x y z,
//        |p_rel|^2, leg torques (3)], friction (16), z (1), c - FK twice (24), jpos twice (24)]:  141 rows (117 in the last interval)
// Rows that the script states twice (two one-sided inequalities on the same expression) appear twice, so that lbg / ubg can be the script's.
// One stage function, templated on the scalar: double for g, Dual for one tangent direction -- the Jacobian block of a stage
// (rows x 72 columns over w = [X_k, c_k, f_k, jpos_k, X_k+1, c_k+1]) is produced exactly, one thread per (member, interval, column),
// the way the reference gets it from CasADi's algorithmic differentiation.
struct KdNlpParams { double dt[64]; double mass, Ib[3], Ibi[3], mu; int std_base = 0; };      // std_base: the tree's base is Px Py Pz Rx Ry Rz with identity tree transforms (landing_rbd_set_model)
constexpr int KD_NW = 72, KD_ROWS = 141, KD_ROWS_LAST = 117, KD_BND = 48;
__host__ __device__ inline int kd_ng(int N) { return KD_BND + (N - 1) * KD_ROWS + KD_ROWS_LAST; }
__host__ __device__ inline int kd_nx(int N) { return 12 * (N + 1) + 12 * N + 24 * N; }

__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const double q = a.v / b.v; return D_(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return D_(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator+(Dual a, double b) { return D_(a.v + b, a.d); }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return D_(a.v - b, a.d); }
__device__ __forceinline__ void sincos_t(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ __forceinline__ void sincos_t(Dual x, Dual& s, Dual& c) { double sv, cv; sincos(x.v, &sv, &cv); s = D_(sv, cv * x.d); c = D_(cv, -sv * x.d); }
__device__ __forceinline__ double lit(double, double v) { return v; }
__device__ __forceinline__ Dual lit(Dual, double v) { return D_(v); }
__device__ __forceinline__ V3D mk3(Dual x, Dual y, Dual z) { return mk3D(x, y, z); }
template <class T> struct KdVec { typedef V3d type; };
template <> struct KdVec<Dual> { typedef V3D type; };
// second-order forward mode: value, two first-order parts and the mixed second-order part along directions (e_i, e_j); the same stage
// function instantiated on this scalar gives d^2 rows / dw_i dw_j exactly (hyper-dual numbers)
struct HDual { double v, a, b, ab; };
__device__ __forceinline__ HDual H_(double v, double a = 0.0, double b = 0.0, double ab = 0.0) { HDual o; o.v = v; o.a = a; o.b = b; o.ab = ab; return o; }
__device__ __forceinline__ HDual operator+(HDual x, HDual y) { return H_(x.v + y.v, x.a + y.a, x.b + y.b, x.ab + y.ab); }
__device__ __forceinline__ HDual operator-(HDual x, HDual y) { return H_(x.v - y.v, x.a - y.a, x.b - y.b, x.ab - y.ab); }
__device__ __forceinline__ HDual operator*(HDual x, HDual y) { return H_(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b, x.ab * y.v + x.a * y.b + x.b * y.a + x.v * y.ab); }
__device__ __forceinline__ HDual operator*(HDual x, double s) { return H_(x.v * s, x.a * s, x.b * s, x.ab * s); }
__device__ __forceinline__ HDual operator*(double s, HDual x) { return H_(x.v * s, x.a * s, x.b * s, x.ab * s); }
__device__ __forceinline__ HDual operator+(HDual x, double s) { return H_(x.v + s, x.a, x.b, x.ab); }
__device__ __forceinline__ HDual operator-(HDual x, double s) { return H_(x.v - s, x.a, x.b, x.ab); }
__device__ __forceinline__ HDual operator/(HDual x, HDual y) {
  const double f = 1.0 / y.v, f1 = -f * f, f2 = -2.0 * f * f1;
  return x * H_(f, f1 * y.a, f1 * y.b, f2 * y.a * y.b + f1 * y.ab);
}
__device__ __forceinline__ void sincos_t(HDual x, HDual& s, HDual& c) {
  double sv, cv; sincos(x.v, &sv, &cv);
  s = H_(sv, cv * x.a, cv * x.b, -sv * x.a * x.b + cv * x.ab); c = H_(cv, -sv * x.a, -sv * x.b, -cv * x.a * x.b - sv * x.ab);
}
__device__ __forceinline__ HDual lit(HDual, double v) { return H_(v); }
struct V3H { HDual x, y, z; };
__device__ __forceinline__ V3H mk3(HDual x, HDual y, HDual z) { V3H v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3H add3(V3H a, V3H b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3H sub3(V3H a, V3H b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3H crs3(V3H a, V3H b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3H mul3(const HDual* E, V3H v) { return mk3(E[0] * v.x + E[1] * v.y + E[2] * v.z, E[3] * v.x + E[4] * v.y + E[5] * v.z, E[6] * v.x + E[7] * v.y + E[8] * v.z); }
__device__ __forceinline__ V3H mulT3(const HDual* E, V3H v) { return mk3(E[0] * v.x + E[3] * v.y + E[6] * v.z, E[1] * v.x + E[4] * v.y + E[7] * v.z, E[2] * v.x + E[5] * v.y + E[8] * v.z); }
template <> struct KdVec<HDual> { typedef V3H type; };
template <int JT>
__device__ __forceinline__ void joint_xform_t(HDual q, const double* Et, const double* rt, HDual* E, HDual* r) {
  if (JT < 3) {
    HDual S, Cc; sincos_t(q, S, Cc);
    constexpr int a = JT < 3 ? JT : 0, b = (a + 1) % 3, d = (a + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      E[3 * a + j] = H_(Et[3 * a + j]);
      E[3 * b + j] = Et[3 * b + j] * Cc + Et[3 * d + j] * S;
      E[3 * d + j] = Et[3 * d + j] * Cc - Et[3 * b + j] * S;
    }
    r[0] = H_(rt[0]); r[1] = H_(rt[1]); r[2] = H_(rt[2]);
  } else {
    constexpr int a = JT >= 3 ? JT - 3 : 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) E[j] = H_(Et[j]);
    r[0] = H_(rt[0]) + Et[3 * a] * q; r[1] = H_(rt[1]) + Et[3 * a + 1] * q; r[2] = H_(rt[2]) + Et[3 * a + 2] * q;
  }
}
__device__ __forceinline__ void joint_xform(int jt, HDual q, const double* Et, const double* rt, HDual* E, HDual* r) {
  switch (jt) {
    case 0: joint_xform_t<0>(q, Et, rt, E, r); break;
    case 1: joint_xform_t<1>(q, Et, rt, E, r); break;
    case 2: joint_xform_t<2>(q, Et, rt, E, r); break;
    case 3: joint_xform_t<3>(q, Et, rt, E, r); break;
    case 4: joint_xform_t<4>(q, Et, rt, E, r); break;
    default: joint_xform_t<5>(q, Et, rt, E, r); break;
  }
}


template <class T>
__device__ void kd_compose(const T* Eu, const T* ru, T* Ea, T* ra) {      // (Ea, ra) <- plux(Eu, ru) * plux(Ea, ra)
  const typename KdVec<T>::type t = mulT3(Ea, mk3(ru[0], ru[1], ru[2]));
  T En[9];
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) En[3 * a + b] = Eu[3 * a] * Ea[b] + Eu[3 * a + 1] * Ea[3 + b] + Eu[3 * a + 2] * Ea[6 + b];
  for (int j = 0; j < 9; ++j) Ea[j] = En[j];
  ra[0] = ra[0] + t.x; ra[1] = ra[1] + t.y; ra[2] = ra[2] + t.z;
}

// (Ea, ra) <- plux(rot_A(q) Et, rt) * plux(Ea, ra) for a revolute joint about axis A whose sine / cosine the caller holds (the leg rows need them for the torque
// rows anyway): t = Ea' rt and G = Et Ea cost products with the model's constants only, then two rows of G are rotated -- 12 products of T numbers instead of the 27 of
// the general composition of joint_xform's dense E (and no second sincos per joint)
template <int A, class T>
__device__ __forceinline__ void kd_rot_compose_t(const T& S, const T& Cc, const double* Et, const double* rt, T* Ea, T* ra) {
  constexpr int b = (A + 1) % 3, d = (A + 2) % 3;
  const T tx = Ea[0] * rt[0] + Ea[3] * rt[1] + Ea[6] * rt[2], ty = Ea[1] * rt[0] + Ea[4] * rt[1] + Ea[7] * rt[2], tz = Ea[2] * rt[0] + Ea[5] * rt[1] + Ea[8] * rt[2];
  T G[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) G[3 * r + c] = Ea[c] * Et[3 * r] + Ea[3 + c] * Et[3 * r + 1] + Ea[6 + c] * Et[3 * r + 2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Ea[3 * A + c] = G[3 * A + c];
    Ea[3 * b + c] = Cc * G[3 * b + c] + S * G[3 * d + c];
    Ea[3 * d + c] = Cc * G[3 * d + c] - S * G[3 * b + c];
  }
  ra[0] = ra[0] + tx; ra[1] = ra[1] + ty; ra[2] = ra[2] + tz;
}
template <class T>
__device__ __forceinline__ void kd_rot_compose(int jt, const T& S, const T& Cc, const double* Et, const double* rt, T* Ea, T* ra) {
  switch (jt) {
    case 0: kd_rot_compose_t<0>(S, Cc, Et, rt, Ea, ra); break;
    case 1: kd_rot_compose_t<1>(S, Cc, Et, rt, Ea, ra); break;
    default: kd_rot_compose_t<2>(S, Cc, Et, rt, Ea, ra); break;
  }
}

// Kinematic rows of one leg of an interval (landing_optimization.m:148-171, 182): p_rel x y z, |p_rel|^2, the three leg torques -> o7[0..6];
// foot position of the tree -> fk3.  Everything it needs comes through pointers (w = the interval's 72 variables, R body -> world, (E0, r0) the
// world -> base transform of the tree).  For T = double the function is called out of line (kd_leg_kin_d): inlined four times into the row
// code it put the in-kernel row evaluation of the interior-point solver at 253 VGPRs + 32 AGPRs of spill, one workgroup per CU.
template <class T>
__device__ __forceinline__ void kd_leg_kin(const RbdModel& M, int l, const T* w, const T* R, const T* E0, const T* r0, T* o7, T* fk3) {
  typedef typename KdVec<T>::type V;
  const T zero = lit(w[0], 0.0);
  const T* X = w; const T* c = w + 12; const T* f = w + 24; const T* jp = w + 36;
  const V pos = mk3(X[0], X[1], X[2]);
  const double l14 = M.l1 + M.l4;
  T Ej[9], rj[3];
  int q = 0;
    const double hx = l < 2 ? 0.19 : -0.19, hy = (l & 1) ? 0.1 : -0.1;        // params.hipSrbmLocation (get_robot_params.m:90-91)
    const V pr = sub3(mk3(c[3 * l], c[3 * l + 1], c[3 * l + 2]), add3(pos, mk3(R[0] * hx + R[1] * hy, R[3] * hx + R[4] * hy, R[6] * hx + R[7] * hy)));
    o7[q++] = pr.x; o7[q++] = pr.y; o7[q++] = pr.z;                                                                       // :157-163
    o7[q++] = pr.x * pr.x + pr.y * pr.y + pr.z * pr.z;                                                                      // :164
    // leg torques J_f'(-R_world_to_body f)  (:167-171, get_foot_jacobians_mc.m:12-24)
    T s1, c1, s2, c2, s3, c3;
    sincos_t(jp[3 * l], s1, c1); sincos_t(jp[3 * l + 1], s2, c2); sincos_t(jp[3 * l + 2], s3, c3);
    const T c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
    const double ss = (l & 1) ? 1.0 : -1.0;
    const T J[3][3] = {{zero, c23 * M.l3 + c2 * M.l2, c23 * M.l3},
                       {c1 * c23 * M.l3 + c1 * c2 * M.l2 - s1 * (l14 * ss), zero - s1 * s23 * M.l3 - s1 * s2 * M.l2, zero - s1 * s23 * M.l3},
                       {s1 * c23 * M.l3 + c2 * s1 * M.l2 + c1 * (l14 * ss), c1 * s23 * M.l3 + c1 * s2 * M.l2, c1 * s23 * M.l3}};
    const V fb = mulT3(R, mk3(zero - f[3 * l], zero - f[3 * l + 1], zero - f[3 * l + 2]));
    for (int j = 0; j < 3; ++j) o7[q++] = J[0][j] * fb.x + J[1][j] * fb.y + J[2][j] * fb.z;
    // foot position of the tree (get_forward_kin_foot.m)
    T El[9], rl[3];
    for (int j = 0; j < 9; ++j) El[j] = E0[j];
    for (int j = 0; j < 3; ++j) rl[j] = r0[j];
    const int jb = M.b_foot[l] - 1;
    {
      const T* Sj[3] = {&s1, &s2, &s3}; const T* Cj[3] = {&c1, &c2, &c3};
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = jb - 2 + u, jt = M.jtype[i];
        if (jt < 3) kd_rot_compose(jt, *Sj[u], *Cj[u], M.E[i], M.r[i], El, rl);      // (uniform: the model)
        else { joint_xform(jt, jp[3 * l + u], M.E[i], M.r[i], Ej, rj); kd_compose(Ej, rj, El, rl); }
      }
    }
    const double fr0 = M.foot_r[l][0], fr1 = M.foot_r[l][1], fr2 = M.foot_r[l][2];      // (El' foot_r with the model's constants as plain doubles)
    const V pf = mk3(rl[0] + (El[0] * fr0 + El[3] * fr1 + El[6] * fr2), rl[1] + (El[1] * fr0 + El[4] * fr1 + El[7] * fr2), rl[2] + (El[2] * fr0 + El[5] * fr1 + El[8] * fr2));
  fk3[0] = pf.x; fk3[1] = pf.y; fk3[2] = pf.z;
}
__device__ __noinline__ void kd_leg_kin_d(const RbdModel& M, int l, const double* w, const double* R, const double* E0, const double* r0, double* o7, double* fk3) {
  kd_leg_kin<double>(M, l, w, R, E0, r0, o7, fk3);
}

// The frames kd_stage_rows<double> hands to kd_leg_kin_d -- R_body_to_world and the world -> base transform of the tree -- for callers that evaluate one leg of an interval
// on its own lane (kd_solver_kernels.hip kd_member_eval_g): the same expressions in the same order.
__device__ __noinline__ void kd_base_frames_d(const KdNlpParams& P, const RbdModel& M, const double* w, double* R, double* E0, double* r0) {
  const double* X = w;
  double sr, cr, sp, cp, sy, cy;
  sincos_t(X[3], sr, cr); sincos_t(X[4], sp, cp); sincos_t(X[5], sy, cy);
  const double zero = 0.0;
  const double Rr[9] = {cp * cy, zero - cp * sy, sp,
                        cr * sy + sr * sp * cy, cr * cy - sr * sp * sy, zero - sr * cp,
                        sr * sy - cr * sp * cy, sr * cy + cr * sp * sy, cr * cp};
  for (int j = 0; j < 9; ++j) R[j] = Rr[j];
  if (P.std_base) {
    for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) E0[3 * a + b2] = R[3 * b2 + a];
    r0[0] = X[0]; r0[1] = X[1]; r0[2] = X[2];
  } else {
    double Ej[9], rj[3];
    for (int j = 0; j < 9; ++j) E0[j] = (j % 4 == 0) ? 1.0 : 0.0;
    r0[0] = r0[1] = r0[2] = 0.0;
    for (int i = 0; i < 6; ++i) { joint_xform(M.jtype[i], X[i], M.E[i], M.r[i], Ej, rj); kd_compose(Ej, rj, E0, r0); }
  }
}

// legmask: bit l set = the rows of leg l that need the leg's kinematics (hip-relative position, leg torques, forward kinematics: the
// expensive part of the function) are evaluated; a cleared bit writes zeros there.  The value / Jacobian kernels pass 15; the Hessian kernel
// passes the one leg a pair of directions belongs to (second derivatives of the other legs' rows vanish for that pair).
// `out` is an emitter: out.put(v) receives the rows one after the other.  (Round 4: the callers used to pass an array of KD_ROWS values -- 4.5 KB of
// hyper-dual numbers per thread in private memory in the Hessian kernel, which only needs lam' out; KdRowArray below is the array form.)
template <class T> struct KdRowArray { T* p; __device__ __forceinline__ void put(const T& v) { *p++ = v; } };
// w: the stage's 72 variables -- an array of T, or a view that forms them on access (KdSeedView: the hyper-dual kernel keeps 72 doubles and two seed
// indices instead of 72 hyper-dual numbers)
// STDB: 1 / 0 = the base-transform form is decided at compile time (the derivative kernels exist in both forms: a run-time test keeps the chain's registers alive,
// 0.348 against 0.335 s per batch), -1 = by P.std_base (the value paths)
template <class T, class OUT, class WIN, int STDB = -1>
__device__ void kd_stage_rows(const KdNlpParams& P, const RbdModel& M, int k, bool last, const WIN& w, OUT& out, int legmask = 15) {
  typedef typename KdVec<T>::type V;
  const T zero = lit(w[0], 0.0);
  const auto X = w + 0; const auto c = w + 12; const auto f = w + 24; const auto jp = w + 36; const auto Xn = w + 48; const auto cn = w + 60;
  const double dt = P.dt[k];
  const V pos = mk3(X[0], X[1], X[2]), om = mk3(X[6], X[7], X[8]), v = mk3(X[9], X[10], X[11]);
  T sr, cr, sp, cp, sy, cy;
  sincos_t(X[3], sr, cr); sincos_t(X[4], sp, cp); sincos_t(X[5], sy, cy);
  // R_body_to_world = rx(r)' ry(p)' rz(y)'  (rpyToRotMat_xyz.m:2), row-major
  const T R[9] = {cp * cy, zero - cp * sy, sp,
                  cr * sy + sr * sp * cy, cr * cy - sr * sp * sy, zero - sr * cp,
                  sr * sy - cr * sp * cy, sr * cy + cr * sp * sy, cr * cp};
  V fs = mk3(zero, zero, zero), tw = mk3(zero, zero, zero);
  for (int l = 0; l < 4; ++l) {
    const V fl = mk3(f[3 * l], f[3 * l + 1], f[3 * l + 2]);
    fs = add3(fs, fl);
    tw = add3(tw, crs3(sub3(mk3(c[3 * l], c[3 * l + 1], c[3 * l + 2]), pos), fl));
  }
  const V tb = mulT3(R, tw);                                                        // R_world_to_body * torque
  const V Iw = mk3(om.x * P.Ib[0], om.y * P.Ib[1], om.z * P.Ib[2]);
  const V nn = crs3(om, Iw);
  const V omd = mk3((tb.x - nn.x) * P.Ibi[0], (tb.y - nn.y) * P.Ibi[1], (tb.z - nn.z) * P.Ibi[2]);
  const double im = 1.0 / P.mass;
  const V rdd = mk3(fs.x * im, fs.y * im, fs.z * im - 9.81);
  const V Rw = mul3(R, om);
  // Binv(rpy) (Binv.m:13-17), psi = yaw, theta = pitch
  const T ict = lit(w[0], 1.0) / cp, tt = sp * ict;
  const V ed = mk3((cy * Rw.x + sy * Rw.y) * ict, cy * Rw.y - sy * Rw.x, (cy * Rw.x + sy * Rw.y) * tt + Rw.z);
  out.put(Xn[9] - X[9] - rdd.x * dt); out.put(Xn[10] - X[10] - rdd.y * dt); out.put(Xn[11] - X[11] - rdd.z * dt);       // :125
  out.put(Xn[6] - X[6] - omd.x * dt); out.put(Xn[7] - X[7] - omd.y * dt); out.put(Xn[8] - X[8] - omd.z * dt);          // :126
  out.put(Xn[0] - X[0] - v.x * dt); out.put(Xn[1] - X[1] - v.y * dt); out.put(Xn[2] - X[2] - v.z * dt);                // :127
  out.put(Xn[3] - X[3] - ed.x * dt); out.put(Xn[4] - X[4] - ed.y * dt); out.put(Xn[5] - X[5] - ed.z * dt);             // :128
  for (int l = 0; l < 4; ++l) out.put(f[3 * l + 2]);                                                                       // :131
  // world -> base transform of the tree (the six base joints take pos, rpy), for the forward kinematics of :182
  T E0[9], r0[3], Ej[9], rj[3];
  if (STDB == 1 || (STDB < 0 && P.std_base)) {      // (uniform: a template argument or a kernel argument)
    // floating base Px Py Pz Rx Ry Rz with identity tree transforms (rbd.quad3d_model / landing_rbd_model_mc3d; checked by landing_rbd_set_model): the chain of the
    // six base joints is rz(y) ry(p) rx(r) and the position -- the transpose of R and pos, which the rows above have formed already.  In the hyper-dual kernel the chain
    // was six 3 x 3 products of hyper-dual numbers: 0.363 -> 0.334 s per batch of the refinement
    for (int a = 0; a < 3; ++a) for (int b2 = 0; b2 < 3; ++b2) E0[3 * a + b2] = R[3 * b2 + a];
    r0[0] = pos.x; r0[1] = pos.y; r0[2] = pos.z;
  } else {
    for (int j = 0; j < 9; ++j) E0[j] = lit(w[0], (j % 4 == 0) ? 1.0 : 0.0);
    r0[0] = r0[1] = r0[2] = zero;
    for (int i = 0; i < 6; ++i) { joint_xform(M.jtype[i], X[i], M.E[i], M.r[i], Ej, rj); kd_compose(Ej, rj, E0, r0); }
  }
  T fkv[12];
  const double l14 = M.l1 + M.l4;
  for (int l = 0; l < 4; ++l) {
    const T cz = c[3 * l + 2], fz = f[3 * l + 2];
    out.put(cz);                                                                                                           // :138
    out.put(fz * cz);                                                                                                      // :139
    if (!last) {
      for (int a = 0; a < 3; ++a) out.put(fz * (cn[3 * l + a] - c[3 * l + a]));                                              // :142
      for (int a = 0; a < 3; ++a) out.put(fz * (cn[3 * l + a] - c[3 * l + a]));                                              // :143
    }
    if (!((legmask >> l) & 1)) {      // rows of this leg's kinematics are not needed by the caller
      for (int a = 0; a < 7; ++a) out.put(zero);
      fkv[3 * l] = c[3 * l]; fkv[3 * l + 1] = c[3 * l + 1]; fkv[3 * l + 2] = c[3 * l + 2];
      continue;
    }
    // hip-relative foot position (4 rows), leg torques (3 rows), foot position of the tree: kd_leg_kin (out of line for T = double)
    if constexpr (std::is_same<T, double>::value) { double o7[7]; kd_leg_kin_d(M, l, w, R, E0, r0, o7, fkv + 3 * l); for (int a = 0; a < 7; ++a) out.put(o7[a]); continue; }
    // (the dual / hyper-dual instantiations of the derivative kernels keep the rows inline: the same code as kd_leg_kin, in place)
    const double hx = l < 2 ? 0.19 : -0.19, hy = (l & 1) ? 0.1 : -0.1;        // params.hipSrbmLocation (get_robot_params.m:90-91)
    const V pr = sub3(mk3(c[3 * l], c[3 * l + 1], c[3 * l + 2]), add3(pos, mk3(R[0] * hx + R[1] * hy, R[3] * hx + R[4] * hy, R[6] * hx + R[7] * hy)));
    out.put(pr.x); out.put(pr.y); out.put(pr.z);                                                                       // :157-163
    out.put(pr.x * pr.x + pr.y * pr.y + pr.z * pr.z);                                                                      // :164
    // leg torques J_f'(-R_world_to_body f)  (:167-171, get_foot_jacobians_mc.m:12-24)
    T s1, c1, s2, c2, s3, c3;
    sincos_t(jp[3 * l], s1, c1); sincos_t(jp[3 * l + 1], s2, c2); sincos_t(jp[3 * l + 2], s3, c3);
    const T c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
    const double ss = (l & 1) ? 1.0 : -1.0;
    const T J[3][3] = {{zero, c23 * M.l3 + c2 * M.l2, c23 * M.l3},
                       {c1 * c23 * M.l3 + c1 * c2 * M.l2 - s1 * (l14 * ss), zero - s1 * s23 * M.l3 - s1 * s2 * M.l2, zero - s1 * s23 * M.l3},
                       {s1 * c23 * M.l3 + c2 * s1 * M.l2 + c1 * (l14 * ss), c1 * s23 * M.l3 + c1 * s2 * M.l2, c1 * s23 * M.l3}};
    const V fb = mulT3(R, mk3(zero - f[3 * l], zero - f[3 * l + 1], zero - f[3 * l + 2]));
    for (int j = 0; j < 3; ++j) out.put(J[0][j] * fb.x + J[1][j] * fb.y + J[2][j] * fb.z);
    // foot position of the tree (get_forward_kin_foot.m)
    T El[9], rl[3];
    for (int j = 0; j < 9; ++j) El[j] = E0[j];
    for (int j = 0; j < 3; ++j) rl[j] = r0[j];
    const int jb = M.b_foot[l] - 1;
    {      // (kd_rot_compose: the joints' sines / cosines are those of the torque rows above)
      const T* Sj[3] = {&s1, &s2, &s3}; const T* Cj[3] = {&c1, &c2, &c3};
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int i = jb - 2 + u, jt = M.jtype[i];
        if (jt < 3) kd_rot_compose(jt, *Sj[u], *Cj[u], M.E[i], M.r[i], El, rl);      // (uniform: the model)
        else { joint_xform(jt, jp[3 * l + u], M.E[i], M.r[i], Ej, rj); kd_compose(Ej, rj, El, rl); }
      }
    }
    const double fr0 = M.foot_r[l][0], fr1 = M.foot_r[l][1], fr2 = M.foot_r[l][2];      // (El' foot_r with the model's constants as plain doubles)
    const V pf = mk3(rl[0] + (El[0] * fr0 + El[3] * fr1 + El[6] * fr2), rl[1] + (El[1] * fr0 + El[4] * fr1 + El[7] * fr2), rl[2] + (El[2] * fr0 + El[5] * fr1 + El[8] * fr2));
    fkv[3 * l] = pf.x; fkv[3 * l + 1] = pf.y; fkv[3 * l + 2] = pf.z;
  }
  const double km = 0.71 * P.mu;
  for (int l = 0; l < 4; ++l) out.put(f[3 * l] - f[3 * l + 2] * km);                                                         // :175
  for (int l = 0; l < 4; ++l) out.put(f[3 * l + 2] * (-km) - f[3 * l]);      // :176  f_x >= -km f_z: neither side is parametric, so Opti holds it as  -km f_z - f_x <= 0  (optistack_internal.cpp:793-806: e = args[j] - args[j+1] of `(-km f_z) <= f_x`)
  for (int l = 0; l < 4; ++l) out.put(f[3 * l + 1] - f[3 * l + 2] * km);                                                     // :177
  for (int l = 0; l < 4; ++l) out.put(f[3 * l + 2] * (-km) - f[3 * l + 1]);  // :178  likewise
  out.put(X[2]);                                                                                                           // :181
  for (int j = 0; j < 12; ++j) out.put(c[j] - fkv[j]);                                                                     // :186
  for (int j = 0; j < 12; ++j) out.put(c[j] - fkv[j]);                                                                     // :187
  for (int j = 0; j < 12; ++j) out.put(jp[j]);                                                                             // :188
  for (int j = 0; j < 12; ++j) out.put(jp[j]);                                                                             // :189
}

// Compact form of an interval's Jacobian block for the solver (round 6): the 12 defect rows dense (12 x 72), then the structural non-zeros of the inequality rows in the order of
// the solver's table (kd_solver_kernels.hip KdJPat: 529 / 457 entries); KdNlpArgs::jcol lists, per column of the 141 x 72 block, the rows that are stored and their places there.
// The dense block is 93 % zeros and the Jacobian kernel was bound by storing them (1.37 GB per round of the full batch).
constexpr int KD_JC_DEF = 12 * KD_NW, KD_JC_NNZ = 640, KD_JCS = KD_JC_DEF + KD_JC_NNZ, KD_JCOL = 96;
struct KdNlpArgs {
  const RbdModel* model; KdNlpParams P; int B, N; const double* x; double* g; double* jac; const double* lam; double* hess;
  // member strides in doubles (0 = dense arrays [B][nx], [B][ng], [B][N][141][72], [B][N][72][72]) and an optional per-member skip flag: the
  // interior-point solver (kd_solver_kernels.hip) evaluates straight into its per-member workspace and skips members that have finished
  long long sx, sg, sj, sh; const int* skip;
  const unsigned* jcol = nullptr;   // optional [2][72][KD_JCOL]: per column of a block (middle intervals | last interval) its stored rows in rising order, (row << 16) | place in the compact form, closed by row 0xffff:
                                    // the Jacobian kernel writes compact blocks of KD_JCS doubles per interval instead of dense ones
  const int* list = nullptr; const int* n_list = nullptr;      // optional work list (round 6): block index b stands for member list[b], b < *n_list (the solver's members that need derivatives this round)
  double* jty = nullptr;      // optional [member stride sj][N][72]: J_k' lam_k per interval and column, a by-product of the Jacobian kernel (the solver's grad f + J' y)
  __host__ __device__ size_t ox(int b) const { return (size_t)b * (sx ? (size_t)sx : (size_t)(12 * (N + 1) + 36 * N)); }
  __host__ __device__ size_t og(int b) const { return (size_t)b * (sg ? (size_t)sg : (size_t)(48 + (N - 1) * 141 + 117)); }
  __host__ __device__ size_t oj(int b) const { return (size_t)b * (sj ? (size_t)sj : (size_t)N * 141 * 72); }
  __host__ __device__ size_t oh(int b) const { return (size_t)b * (sh ? (size_t)sh : (size_t)N * 72 * 72); }
};
// index of w[j] of interval k in x
__device__ __forceinline__ int kd_w_index(int N, int k, int j) {
  const int oJ = 12 * (N + 1), oU = oJ + 12 * N;
  if (j < 12) return 12 * k + j;
  if (j < 24) return oU + 24 * k + (j - 12);
  if (j < 36) return oU + 24 * k + 12 + (j - 24);
  if (j < 48) return oJ + 12 * k + (j - 36);
  if (j < 60) return 12 * (k + 1) + (j - 48);
  return k + 1 < N ? oU + 24 * (k + 1) + (j - 60) : -1;          // c of the next interval (the last interval has none)
}
// g: one thread per (member, interval); the thread of interval 0 also writes the 48 boundary rows
__global__ void __launch_bounds__(64) landing_kinodyn_nlp_g_kernel(KdNlpArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.B * a.N) return;
  const int b = idx / a.N, k = idx % a.N, N = a.N;
  if (a.skip && a.skip[b]) return;
  const double* x = a.x + a.ox(b);
  double* g = a.g + a.og(b);
  double w[KD_NW];
#pragma unroll
  for (int j = 0; j < KD_NW; ++j) { const int i = kd_w_index(N, k, j); w[j] = i >= 0 ? x[i] : 0.0; }
  const bool last = k == N - 1;
  KdRowArray<double> out{g + KD_BND + k * KD_ROWS};      // (the last interval emits KD_ROWS_LAST rows)
  kd_stage_rows<double>(a.P, *a.model, k, last, w, out);
  if (k == 0) {
    for (int i = 0; i < 12; ++i) g[i] = x[i];                                            // q(:,1), qdot(:,1)
    for (int i = 0; i < 12; ++i) g[12 + i] = x[12 * (N + 1) + 12 * N + i];               // c(:,1)
    for (int i = 0; i < 6; ++i) { g[24 + i] = x[12 * N + i]; g[30 + i] = x[12 * N + i]; g[36 + i] = x[12 * N + 6 + i]; g[42 + i] = x[12 * N + 6 + i]; }
  }
}
// Jacobian blocks: jac[b][k][row][col], col over w (72), one thread per (member, interval, column)
// One block = KD_JAC_STAGES intervals of ONE member x 72 columns (4 intervals = 4.5 wavefronts: 0.435 s per batch; 8 = 9 full wavefronts but 168 VGPRs: 0.443; 2: 0.453); the stage variables of those intervals live in LDS and the seeded
// dual numbers are formed on access (as in the Hessian kernel below) instead of 72 dual numbers = 144 VGPRs per lane.  Grid = B * ceil(N / KD_JAC_STAGES).
#ifndef KD_JAC_STAGES_DEF
#define KD_JAC_STAGES_DEF 4
#endif
// Round 6: wavefront-pure columns, as in the Hessian kernel.  A column of leg l (c_l, f_l, jpos_l, c_l of the next interval) only moves that leg's kinematic rows -- the expensive part of
// kd_stage_rows -- so wave l (0..3) takes the 12 columns of leg l of the block's four intervals (48 lanes, legmask 1 << l), wave 4 the 12 columns of X (every leg: pos and rpy move them all),
// wave 5 the 12 columns of X_k+1 (no leg: the defect rows only).  4.5 waves x 4 legs of kinematics became 4 x 1 + 1 x 4: 0.95 -> see DESIGN.md 4.8b ms per round of the full batch.
constexpr int KD_JAC_STAGES = KD_JAC_STAGES_DEF, KD_JAC_THREADS = 6 * 64;
static_assert(KD_JAC_STAGES == 4, "the lane map of landing_kinodyn_nlp_jac_kernel: 4 intervals x 12 columns = 48 lanes of a wavefront");
__host__ __device__ inline long long kd_jac_blocks(long long B, int N) { return B * ((N + KD_JAC_STAGES - 1) / KD_JAC_STAGES); }
template <int STDB>
__global__ void __launch_bounds__(KD_JAC_THREADS) landing_kinodyn_nlp_jac_kernel(KdNlpArgs a) {
  const int N = a.N, nblk = (N + KD_JAC_STAGES - 1) / KD_JAC_STAGES;
  int b = (int)(blockIdx.x / nblk); const int k0 = (int)(blockIdx.x % nblk) * KD_JAC_STAGES;
  if (a.list) { if (b >= *a.n_list) return; b = a.list[b]; }
  if (b >= a.B) return;
  if (a.skip && a.skip[b]) return;
  __shared__ double xs[KD_JAC_STAGES][KD_NW];
  const double* x = a.x + a.ox(b);
  for (int e = (int)threadIdx.x; e < KD_JAC_STAGES * KD_NW; e += KD_JAC_THREADS) {
    const int kk = k0 + e / KD_NW, i = kk < N ? kd_w_index(N, kk, e % KD_NW) : -1;
    xs[e / KD_NW][e % KD_NW] = i >= 0 ? x[i] : 0.0;
  }
  __syncthreads();
  const int wv = (int)threadIdx.x >> 6, ln = (int)threadIdx.x & 63;
  if (ln >= 48) return;
  const int ks = ln / 12, q = ln % 12, k = k0 + ks;
  const int col = wv == 4 ? q : (wv == 5 ? 48 + q : (q < 9 ? 12 + 12 * (q / 3) + 3 * wv + q % 3 : 60 + 3 * wv + (q - 9)));
  const int legmask = wv < 4 ? (1 << wv) : (wv == 4 ? 15 : 0);
  if (k >= N) return;
  struct DualSeedView {      // w[q] = x_q + eps [q == col], formed on access
    const double* xv; int col, off;
    __device__ __forceinline__ Dual operator[](int q) const { const int qq = q + off; return D_(xv[qq], qq == col ? 1.0 : 0.0); }
    __device__ __forceinline__ DualSeedView operator+(int o) const { return DualSeedView{xv, col, off + o}; }
  };
  const DualSeedView w{xs[ks], col, 0};
  const bool last = k == N - 1;
  // row after row of column col; with lam given the column's product with the multipliers of the interval's rows comes along (rows in order)
  struct ColOut { double* J; bool zero; const double* y; double acc; const unsigned* lst; unsigned nxt; int row;      // lst: the column's stored rows (compact form); nxt = the next of them
                  __device__ __forceinline__ void put(const Dual& v) {
                    const double d = zero ? 0.0 : v.d;
                    if (lst) { if (row == (int)(nxt >> 16)) { J[nxt & 0xffffu] = d; nxt = *++lst; } ++row; } else { *J = d; J += KD_NW; }
                    if (y) { acc += d * *y; ++y; } } };
  const unsigned* lst0 = a.jcol ? a.jcol + ((size_t)(last ? KD_NW : 0) + col) * KD_JCOL : nullptr;
  ColOut out{a.jcol ? a.jac + a.oj(b) + (size_t)k * KD_JCS : a.jac + a.oj(b) + ((size_t)k * KD_ROWS) * KD_NW + col, last && col >= 60,
             (a.jty && a.lam) ? a.lam + a.og(b) + KD_BND + (size_t)k * KD_ROWS : nullptr, 0.0, lst0, lst0 ? *lst0 : 0u, 0};
  kd_stage_rows<Dual, ColOut, DualSeedView, STDB>(a.P, *a.model, k, last, w, out, legmask);
  if (a.jty && a.lam) a.jty[a.oj(b) + (size_t)k * KD_NW + col] = out.acc;
}

// Hessian of lam' g restricted to one interval: hess[b][k][i][j] = sum_r lam_r d^2 row_r / dw_i dw_j (72 x 72, symmetric; the 48 boundary rows are
// linear), one thread per (member, interval, pair i <= j) pushing the two directions through the stage function in second-order forward mode.
// X_k+1 enters every row linearly and c_k+1 only through f_z (c_k+1 - c_k): pairs inside [X_k+1, c_k+1] are structural zeros and are skipped.
// The NLP's Hessian of the Lagrangian is the sum of these blocks at their positions in x (+ the constant 2 QN of the terminal cost, :84-86).
// A first, exact implementation for the next round's solver to be checked against -- not tuned (2 628 stage evaluations per interval).
// Structural non-zeros of the block (the rest is written as zeros by a memset): the velocity X[9..11] and X_k+1 enter linearly; the variables
// of a leg (c_l, f_l, jpos_l) couple with themselves and with the base (pos, rpy, omega) only; c_k+1 of a leg only with that leg's f_z:
// 45 + 4 x 81 + 4 x 45 + 12 = 561 pairs instead of 2 628.
// Order: the 45 pairs inside the base first (they need the rows of every leg), then leg by leg the pairs that involve a variable of that leg
// (they need that leg's rows only: kd_stage_rows' legmask) -- the threads of a wavefront mostly share one mask.
__host__ __device__ inline int kd_pair_leg(int i, int j) {      // -1: both directions in the base; else the leg the pair belongs to
  auto leg_of = [](int v) { return v < 12 ? -1 : (v < 48 ? ((v - 12) % 12) / 3 : (v >= 60 ? (v - 60) / 3 : -1)); };
  const int li = leg_of(i), lj = leg_of(j);
  return li >= 0 ? li : lj;
}
__host__ __device__ inline int kd_pair_list(unsigned char* pi, unsigned char* pj) {      // fills (i <= j) pairs, returns their number (561)
  int n = 0;
  auto leg_of = [](int v) { return v < 12 ? -1 : (v < 48 ? ((v - 12) % 12) / 3 : -2); };      // base = -1, leg 0..3 for c / f / jpos entries
  for (int pass = -1; pass < 4; ++pass) {
    for (int i = 0; i < 48; ++i) {
      if (i >= 9 && i < 12) continue;
      for (int j = i; j < 48; ++j) {
        if (j >= 9 && j < 12) continue;
        const int li = leg_of(i), lj = leg_of(j);
        if (li >= 0 && lj >= 0 && li != lj) continue;
        if ((li >= 0 ? li : lj) != pass) continue;
        if (pi) { pi[n] = (unsigned char)i; pj[n] = (unsigned char)j; }
        ++n;
      }
    }
    if (pass >= 0) for (int a = 0; a < 3; ++a) { if (pi) { pi[n] = (unsigned char)(24 + 3 * pass + 2); pj[n] = (unsigned char)(60 + 3 * pass + a); } ++n; }
  }
  return n;
}
constexpr int KD_NPAIR = 561;
// (npair pairs per block: the KD_NPAIR candidates of kd_pair_list, or the subset of them whose entry is not structurally zero -- solver_capi.inc, kd_ensure_pairs)
// One block (one wave) = 64 pairs of ONE (member, interval): the 72 stage variables are common to the block and live in LDS (read on access, one broadcast
// ds_read each) instead of 144 VGPRs per lane -- the kernel ran at 256 VGPRs + 256 AGPRs + 1.6 KB of scratch per lane with them.  Grid = B * N * ceil(npair / 64).
// (The pairs of KD_HESS_G = 2 consecutive intervals are numbered through: 2 x 286 = 572 pairs fill 9 wavefronts but for 4 lanes; interval by interval the fifth
// wavefront of each had 30 of 64 lanes at work.)
#ifndef KD_HESS_G_DEF
#define KD_HESS_G_DEF 2
#endif
constexpr int KD_HESS_G = KD_HESS_G_DEF;
__host__ __device__ inline long long kd_hess_blocks(long long B, int N, int npair) { return B * ((N + KD_HESS_G - 1) / KD_HESS_G) * ((KD_HESS_G * npair + 63) / 64); }
#ifndef KD_HESS_WAVES
#define KD_HESS_WAVES 1
#endif
template <int STDB>
__global__ void __launch_bounds__(64, KD_HESS_WAVES) landing_kinodyn_nlp_hess_kernel(KdNlpArgs a, const unsigned char* __restrict__ pair_i, const unsigned char* __restrict__ pair_j, int npair) {
  const int nch = (KD_HESS_G * npair + 63) / 64, N = a.N, ngr = (N + KD_HESS_G - 1) / KD_HESS_G;
  const long long blk = blockIdx.x;
  const int ch = (int)(blk % nch); const int gr = (int)((blk / nch) % ngr); int b = (int)(blk / ((long long)nch * ngr));
  if (a.list) { if (b >= *a.n_list) return; b = a.list[b]; }
  if (b >= a.B) return;
  if (a.skip && a.skip[b]) return;
  __shared__ double xs[KD_HESS_G][KD_NW];
  const double* x = a.x + a.ox(b);
  for (int t = threadIdx.x; t < KD_HESS_G * KD_NW; t += 64) {
    const int kk = gr * KD_HESS_G + t / KD_NW, q = t % KD_NW;
    const int ix = kk < N ? kd_w_index(N, kk, q) : -1;
    xs[t / KD_NW][q] = ix >= 0 ? x[ix] : 0.0;
  }
  __syncthreads();
  const int p = ch * 64 + (int)threadIdx.x;
  const int ks = p / npair, pr = p - ks * npair, k = gr * KD_HESS_G + ks;
  if (ks >= KD_HESS_G || k >= N) return;
  const double* xv = xs[ks];
  const int i = pair_i[pr], j = pair_j[pr];
  if (i == 255) return;      // padding of the wavefront-pure order (solver_capi.inc kd_ensure_pairs)
  double* Hk = a.hess + a.oh(b) + ((size_t)k * KD_NW) * KD_NW;
  const bool last = k == N - 1;
  if (last && j >= 60) return;
  const double* lam = a.lam + a.og(b) + KD_BND + (size_t)k * KD_ROWS;
  struct KdSeedView {      // w[q] = x_q + eps1 [q == i] + eps2 [q == j], formed on access
    const double* xv; int i, j, off;
    __device__ __forceinline__ HDual operator[](int q) const { const int qq = q + off; return H_(xv[qq], qq == i ? 1.0 : 0.0, qq == j ? 1.0 : 0.0, 0.0); }
    __device__ __forceinline__ KdSeedView operator+(int o) const { return KdSeedView{xv, i, j, off + o}; }
  };
  const KdSeedView w{xv, i, j, 0};
  const int pl = kd_pair_leg(i, j);
  struct LamOut { const double* lam; double s; __device__ __forceinline__ void put(const HDual& v) { s += *lam++ * v.ab; } };      // lam' (second-order part), row after row
  LamOut out{lam, 0.0};
  kd_stage_rows<HDual, LamOut, KdSeedView, STDB>(a.P, *a.model, k, last, w, out, pl < 0 ? 15 : (1 << pl));
  const double s = out.s;
  Hk[i * KD_NW + j] = s; Hk[j * KD_NW + i] = s;
}

// Leg inverse kinematics for the kinodynamic screen: joint angles of every leg such that FK([q6; jpos]) = c (the foot positions of an
// SRBM solution), by damped Newton steps on the tree FK with a central-difference 3 x 3 Jacobian, clamped to the joint limits of
// landing_optimization.m:246-247 (the reference seeds its kinodynamic NLP the same way, through quadInverseKinematics / fsolve,
// misc/inverse_kinematics.m:2).  One thread per (point, leg); res = |FK - c| after the last step.
struct IkArgs { const RbdModel* model; int npts; const double* q6; const double* c; double* jpos; double* res; int iters; double jmin[3], jmax[3]; };
__device__ __forceinline__ V3d leg_fk(const RbdModel& M, const double* E0, const double* r0, int leg, const double* q3) {
  double El[9], rl[3], E[9], r[3];
  for (int j = 0; j < 9; ++j) El[j] = E0[j];
  for (int j = 0; j < 3; ++j) rl[j] = r0[j];
  const int jb = M.b_foot[leg] - 1;
  for (int i = jb - 2; i <= jb; ++i) {
    joint_xform(M.jtype[i], q3[i - (jb - 2)], M.E[i], M.r[i], E, r);
    const V3d t = mulT3(El, mk3(r[0], r[1], r[2]));
    double En[9];
    for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) En[3 * x + y] = E[3 * x] * El[y] + E[3 * x + 1] * El[3 + y] + E[3 * x + 2] * El[6 + y];
    for (int j = 0; j < 9; ++j) El[j] = En[j];
    rl[0] += t.x; rl[1] += t.y; rl[2] += t.z;
  }
  return add3(mk3(rl[0], rl[1], rl[2]), mulT3(El, mk3(M.foot_r[leg][0], M.foot_r[leg][1], M.foot_r[leg][2])));
}
__global__ void __launch_bounds__(64) landing_leg_ik_kernel(IkArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int pt = idx >> 2, leg = idx & 3;
  if (pt >= a.npts) return;
  const RbdModel& M = *a.model;
  const double* q6 = a.q6 + (size_t)pt * 6;
  double E0[9], r0[3], E[9], r[3];
  for (int j = 0; j < 9; ++j) E0[j] = (j % 4 == 0) ? 1.0 : 0.0;
  r0[0] = r0[1] = r0[2] = 0.0;
  for (int i = 0; i < 6; ++i) {
    joint_xform(M.jtype[i], q6[i], M.E[i], M.r[i], E, r);
    const V3d t = mulT3(E0, mk3(r[0], r[1], r[2]));
    double En[9];
    for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) En[3 * x + y] = E[3 * x] * E0[y] + E[3 * x + 1] * E0[3 + y] + E[3 * x + 2] * E0[6 + y];
    for (int j = 0; j < 9; ++j) E0[j] = En[j];
    r0[0] += t.x; r0[1] += t.y; r0[2] += t.z;
  }
  const V3d target = mk3(a.c[(size_t)pt * 12 + 3 * leg], a.c[(size_t)pt * 12 + 3 * leg + 1], a.c[(size_t)pt * 12 + 3 * leg + 2]);
  double q[3] = {0.0, -0.8, 1.6};
  double err = 0.0;
  for (int it = 0; it <= a.iters; ++it) {
    const V3d e = sub3(leg_fk(M, E0, r0, leg, q), target);
    err = sqrt(e.x * e.x + e.y * e.y + e.z * e.z);
    if (it == a.iters || err < 1e-13) break;
    double J[3][3];
    const double h = 1e-6;
    for (int j = 0; j < 3; ++j) {
      const double qj = q[j];
      q[j] = qj + h; const V3d fp = leg_fk(M, E0, r0, leg, q);
      q[j] = qj - h; const V3d fm = leg_fk(M, E0, r0, leg, q);
      q[j] = qj;
      J[0][j] = (fp.x - fm.x) / (2 * h); J[1][j] = (fp.y - fm.y) / (2 * h); J[2][j] = (fp.z - fm.z) / (2 * h);
    }
    // dq = -(J'J + lam 1)^-1 J' e  (Levenberg damping keeps the step finite at the stretched-knee singularity)
    double A[3][3], b[3];
    const double ev[3] = {e.x, e.y, e.z};
    for (int x = 0; x < 3; ++x) { b[x] = 0.0; for (int y = 0; y < 3; ++y) { A[x][y] = (x == y ? 1e-9 : 0.0); for (int t = 0; t < 3; ++t) A[x][y] += J[t][x] * J[t][y]; } for (int t = 0; t < 3; ++t) b[x] += J[t][x] * ev[t]; }
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (!(fabs(det) > 0.0)) break;
    const double dq0 = (b[0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (b[1] * A[2][2] - A[1][2] * b[2]) + A[0][2] * (b[1] * A[2][1] - A[1][1] * b[2])) / det;
    const double dq1 = (A[0][0] * (b[1] * A[2][2] - A[1][2] * b[2]) - b[0] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * b[2] - b[1] * A[2][0])) / det;
    const double dq2 = (A[0][0] * (A[1][1] * b[2] - b[1] * A[2][1]) - A[0][1] * (A[1][0] * b[2] - b[1] * A[2][0]) + b[0] * (A[1][0] * A[2][1] - A[1][1] * A[2][0])) / det;
    const double dq[3] = {dq0, dq1, dq2};
    for (int j = 0; j < 3; ++j) { const double step = fmax(-0.5, fmin(0.5, dq[j])); q[j] = fmax(a.jmin[j], fmin(a.jmax[j], q[j] - step)); }
  }
  for (int j = 0; j < 3; ++j) a.jpos[(size_t)pt * 12 + 3 * leg + j] = q[j];
  if (a.res) a.res[(size_t)pt * 4 + leg] = err;
}

}  // namespace landing
