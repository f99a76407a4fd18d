"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference's tracking-controller synthesis.

  * delta_xdot(): the SRBM variational (error) dynamics exactly as written in
    utilities_general/srbm-utilities/generateVariationalDynamics.m:29-56 (R = rpyToRotMat(rpy)', skew() terms t1, t2a, t2b,
    t2c, t3, the -1e-5 foot term);
  * vbl_AB(): A = jacobian(delta_xdot, delta_x), B = jacobian(delta_xdot, delta_fgrf) (:59-60).  The error dynamics are
    linear in (delta_x, delta_f), so the Jacobians are obtained EXACTLY by evaluating delta_xdot on unit vectors -- no
    closed form is restated here, which keeps this oracle independent of the kernel's hand-written A and B;
  * rde_backward(): generateRiccatiIntegrator.m:24-47 -- Pdot = A'P + PA - P B (R \\ B'P) + Q and the backward step
    `P0 = Pf + dt*k1` (:47; rk4=True: the RK4 combination of :43-46) applied along the grid as
    optimizations/landing/quadruped_SRBM_NLP.m:487-503 does;
  * sample_reference(): the interpolation of (X*, U*) onto the Riccati grid, quadruped_SRBM_NLP.m:489-499.
The reference builds these with CasADi symbolics (absent here: libcasadi is a missing blob), so parity is pinned to this
restatement of its formulas; weights of quadruped_SRBM_NLP.m:437-481 are in reference_weights().
Only tests/ may import this module."""
import numpy as np


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def rpy_to_rot(rpy):
    """rpyToRotMat.m:2  R = rz(y)' * ry(p)' * rx(r)'  with the spatial_v2 coordinate transforms rx/ry/rz (rx.m:8-13 ...)"""
    r, p, y = rpy
    c, s = np.cos, np.sin
    rx = np.array([[1, 0, 0], [0, c(r), s(r)], [0, -s(r), c(r)]])
    ry = np.array([[c(p), 0, -s(p)], [0, 1, 0], [s(p), 0, c(p)]])
    rz = np.array([[c(y), s(y), 0], [-s(y), c(y), 0], [0, 0, 1.0]])
    return rz.T @ ry.T @ rx.T


def delta_xdot(xref, fgrf, dx, df, Ib, mass):
    p, rpy, omega, pf = xref[0:3], xref[3:6], xref[6:9], xref[12:24]
    d_p, d_eta, d_om, d_v, d_pf = dx[0:3], dx[3:6], dx[6:9], dx[9:12], dx[12:24]
    Ib_inv = np.linalg.inv(Ib)
    R = rpy_to_rot(rpy).T                                              # :31  R transforms body to world
    out = np.zeros(24)
    out[0:3] = d_v                                                     # :32
    out[3:6] = -skew(omega) @ d_eta + d_om                             # :33
    t1 = skew(sum(R.T @ skew(pf[3 * l:3 * l + 3] - p) @ fgrf[3 * l:3 * l + 3] for l in range(4))) @ d_eta          # :35-38
    t2a = -sum(skew(fgrf[3 * l:3 * l + 3]) @ d_pf[3 * l:3 * l + 3] for l in range(4))                                # :39-42
    t2b = skew(sum(fgrf[3 * l:3 * l + 3] for l in range(4))) @ d_p                                                  # :43
    t2c = sum(skew(pf[3 * l:3 * l + 3] - p) @ df[3 * l:3 * l + 3] for l in range(4))                                # :44-47
    t3 = skew(Ib @ omega) @ d_om - skew(omega) @ Ib @ d_om                                                          # :48
    out[6:9] = Ib_inv @ (t1 + R.T @ (t2a + t2b + t2c) + t3)            # :49
    out[9:12] = df.reshape(4, 3).sum(axis=0) / mass                    # :51
    out[12:24] = -0.00001 * d_pf                                       # :53
    return out


def vbl_AB(xref, fgrf, Ib, mass):
    A = np.zeros((24, 24)); B = np.zeros((24, 12)); z24, z12 = np.zeros(24), np.zeros(12)
    for i in range(24):
        e = z24.copy(); e[i] = 1.0
        A[:, i] = delta_xdot(xref, fgrf, e, z12, Ib, mass)
    for i in range(12):
        e = z12.copy(); e[i] = 1.0
        B[:, i] = delta_xdot(xref, fgrf, z24, e, Ib, mass)
    return A, B


def rde_rhs(P, A, B, Q, R):
    return A.T @ P + P @ A - P @ B @ np.linalg.solve(R, B.T @ P) + Q          # generateRiccatiIntegrator.m:27


def rde_backward(xref, fref, Ib, mass, Q, R, F, dt, rk4=False):
    """xref [n,24], fref [n,12] on the grid; returns P [n,24,24] (P[n-1] = F) and K [n,12,24] = R^-1 B' P"""
    n = xref.shape[0]
    P = np.zeros((n, 24, 24)); K = np.zeros((n, 12, 24)); P[n - 1] = F
    for j in range(n - 1, -1, -1):
        A, B = vbl_AB(xref[j], fref[j], Ib, mass)
        K[j] = np.linalg.solve(R, B.T @ P[j])
        if j == 0:
            break
        f = lambda M: rde_rhs(M, A, B, Q, R)
        if rk4:
            k1 = f(P[j]); k2 = f(P[j] + dt / 2 * k1); k3 = f(P[j] + dt / 2 * k2); k4 = f(P[j] + dt * k3)
            P[j - 1] = P[j] + dt * (k1 + 2 * k2 + 2 * k3 + k4) / 6
        else:
            P[j - 1] = P[j] + dt * f(P[j])                                    # :47
    return P, K


def sample_reference(X_star, U_star, t_star, dt_r, n):
    """quadruped_SRBM_NLP.m:487-499: xd = interpolation of [X*(1:12); U*(1:12)] between knots, ud = U*(13:24, k_opt)"""
    N = U_star.shape[1]
    xd = np.zeros((n, 24)); ud = np.zeros((n, 12))
    for k in range(1, n + 1):                      # MATLAB index k, t_int = (k-1) dt
        t = (k - 1) * dt_r
        ko = 1
        while t > t_star[ko] and ko < N - 1:        # t_star(k_opt+1), k_opt < N-1  (1-based k_opt)
            ko += 1
        ki = (t_star[ko] - t) / (t_star[ko] - t_star[ko - 1])
        xd[k - 1] = ki * np.concatenate([X_star[:, ko - 1], U_star[:12, ko - 1]]) + (1 - ki) * np.concatenate([X_star[:, ko], U_star[:12, ko - 1]])
        ud[k - 1] = U_star[12:24, ko - 1]
    return xd, ud


def reference_weights():
    """quadruped_SRBM_NLP.m:437-481: terminal F, running Q (24x24, only the body block is non-zero), R = 90 I"""
    F = np.zeros((24, 24)); Q = np.zeros((24, 24))
    F[:12, :12] = np.diag([1, 1, 1, 5, 5, 5, 4, 4, 4, 3, 3, 3.0])
    Q[:12, :12] = np.diag([.25, .25, .25, 1, 1, 1, .5, .5, .5, 1, 1, 1.0])
    return F, Q, 90.0 * np.eye(12)
