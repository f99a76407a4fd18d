// layout.hpp -- sizes and offsets of the reference's vectors for N intervals (host + device POD).
//   x, p, g: generate_landingCtrller_IPOPT.m:41-75, SURVEY rows a1/a2/App. A
//   CCS segments of casadi_s5 / casadi_s4 (landingCtrller_IPOPT.c:63-64): see DESIGN.md
#pragma once

namespace landing {

struct Layout {
  int N;
  int nx, ng, np, nnz_jac, nnz_hess;
  // offsets into p
  int o_dt, o_q_min, o_q_max, o_qd_min, o_qd_max, o_q_init, o_qd_init, o_q_term_min, o_q_term_max,
      o_qd_term_min, o_qd_term_max, o_QN, o_mu, o_l_leg_max, o_f_max, o_mass, o_Ib, o_Ib_inv;
  int o_Uref, o_QX, o_Qc, o_Qf;      // only in the parameter vector of the N=41 script (run_cost == 2); -1 otherwise
  // formulation constants
  double kin_box[3], kin_z_off, comp_eps, slip_eps;
  // running cost of the N=41 script (generate_quadruped_SRBM_CCC.m:81-89): 0 = off (default); 1 = on, weights and force reference are
  // the constants below, p as in the IPOPT variant; 2 = on with THAT script's own parameter vector (layout_ccc_params): Uref, QX, Qc, Qf
  // are entries of p and grad_gamma_p has entries for them
  int run_cost;
  double QX[12], Qc[3], Qf[3], f_ref[3], p_hip[12];

  __host__ __device__ int x_X(int k) const { return 12 * k; }
  __host__ __device__ int x_U(int k) const { return 12 * (N + 1) + 24 * k; }
  __host__ __device__ int g_stage(int k) const { return 36 + 104 * k; }
  __host__ __device__ int rows(int k) const { return k == N - 1 ? 80 : 104; }
  // Jacobian CCS: [X_0..X_{N-1} (157 each) | X_N (36) | U_0 (204) | U_1..U_{N-2} (228 each) | U_{N-1} (180)]
  __host__ __device__ int jx(int k) const { return 157 * k; }
  __host__ __device__ int ju(int k) const { return 157 * N + 36 + (k == 0 ? 0 : 204 + 228 * (k - 1)); }
  __host__ __device__ int ju_len(int k) const { return k == 0 ? (N == 1 ? 156 : 204) : (k == N - 1 ? 180 : 228); }
  // Hessian CCS: [X_0..X_{N-1} (29 each) | X_N (12) | U_0 (148) | U_k (160 each)]
  __host__ __device__ int hx(int k) const { return 29 * k; }
  __host__ __device__ int hu(int k) const { return 29 * N + 12 + (k == 0 ? 0 : 148 + 160 * (k - 1)); }
};

inline Layout make_layout(int N) {
  Layout L;
  L.N = N;
  L.nx = 36 * N + 12; L.ng = 104 * N + 12; L.np = 13 * N + 94;
  L.nnz_jac = 36 + 385 * (N - 1) + 313; L.nnz_hess = 177 * N + 12 * (N - 1) + 12;
  L.o_dt = 12 * (N + 1);
  const int b = L.o_dt + N;
  L.o_q_min = b; L.o_q_max = b + 6; L.o_qd_min = b + 12; L.o_qd_max = b + 18; L.o_q_init = b + 24; L.o_qd_init = b + 30;
  L.o_q_term_min = b + 36; L.o_q_term_max = b + 42; L.o_qd_term_min = b + 48; L.o_qd_term_max = b + 54;
  L.o_QN = b + 60; L.o_mu = b + 72; L.o_l_leg_max = b + 73; L.o_f_max = b + 74; L.o_mass = b + 75; L.o_Ib = b + 76; L.o_Ib_inv = b + 79;
  L.kin_box[0] = 0.15; L.kin_box[1] = 0.15; L.kin_box[2] = 0.30; L.kin_z_off = 0.05; L.comp_eps = 1e-3; L.slip_eps = 1e-2;
  L.run_cost = 0;
  L.o_Uref = L.o_QX = L.o_Qc = L.o_Qf = -1;
  { const double ph[12] = {0.19, -0.1, -0.2, 0.19, 0.1, -0.2, -0.19, -0.1, -0.2, -0.19, 0.1, -0.2};
    for (int i = 0; i < 12; ++i) { L.QX[i] = 0.0; L.p_hip[i] = ph[i]; }
    for (int i = 0; i < 3; ++i) { L.Qc[i] = 0.0; L.Qf[i] = 0.0; L.f_ref[i] = 0.0; } }
  return L;
}

// Parameter vector of the reference's N=41 script, in Opti's order of the ACTIVE parameters (generate_quadruped_SRBM_CCC.m:49-71; c_init
// is declared :58 but unused -- its constraint is commented out :98 -- so Opti drops it like Uref of the IPOPT variant):
//   Xref 12(N+1) | Uref 24N | dt N | q_min q_max qd_min qd_max q_init qd_init q_term_min q_term_max qd_term_min qd_term_max (6 each) |
//   QX 12 | QN 12 | Qc 3 | Qf 3 | mu l_leg_max f_max mass | Ib 3 | Ib_inv 3            np = 37 N + 112  ("np grows by 24N+18")
inline void layout_ccc_params(Layout& L) {
  const int N = L.N;
  L.run_cost = 2;
  L.o_Uref = 12 * (N + 1);
  L.o_dt = L.o_Uref + 24 * N;
  const int b = L.o_dt + N;
  L.o_q_min = b; L.o_q_max = b + 6; L.o_qd_min = b + 12; L.o_qd_max = b + 18; L.o_q_init = b + 24; L.o_qd_init = b + 30;
  L.o_q_term_min = b + 36; L.o_q_term_max = b + 42; L.o_qd_term_min = b + 48; L.o_qd_term_max = b + 54;
  L.o_QX = b + 60; L.o_QN = b + 72; L.o_Qc = b + 84; L.o_Qf = b + 87;
  L.o_mu = b + 90; L.o_l_leg_max = b + 91; L.o_f_max = b + 92; L.o_mass = b + 93; L.o_Ib = b + 94; L.o_Ib_inv = b + 97;
  L.np = b + 100;
}

}  // namespace landing
