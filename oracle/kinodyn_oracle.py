"""TEST INFRASTRUCTURE -- numpy restatement of the rows of the reference's kinodynamic refinement NLP:
optimizations/landing/main_scripts/landing_optimization.m:100-189, one function per row group, evaluated on a whole trajectory
X [12, N+1] (q = x y z roll pitch yaw; qdot = omega_body, v_world), U [24, N] (c; f_grf -- the script's order, :40-42), jpos [12, N].
Rotation rpyToRotMat_xyz.m:2, Euler rates Binv.m:13-17, foot Jacobians get_foot_jacobians_mc.m, forward kinematics
get_forward_kin_foot.m (through oracle/rbd_oracle.py).  Pinned by reference-held data: the two stored solutions of
optimizations/landing/test_scripts (tests/golden/n1_kinodyn_solutions.npz, tests/test_n1_rows.py) satisfy every row group below
to the feasibility tolerance of the solver that produced them (KNITRO feastol 1e-4 relative / 1e-3 absolute, :392-393).
Only tests/ may import this module."""
import numpy as np

from oracle import rbd_oracle as ro

HIP = np.array([[0.19, -0.1, 0.0], [0.19, 0.1, 0.0], [-0.19, -0.1, 0.0], [-0.19, 0.1, 0.0]])      # params.hipSrbmLocation, get_robot_params.m:90-91
TAU_MAX = np.array([18.0, 18.0, 28.0])          # model.gr .* motorTauMax = [6 6 9.33] * 3, get_robot_model.m:237-241 (rounded as the script's comparison needs)
JPOS_MIN = np.tile([-np.pi / 3, -np.pi / 2, 0.0], 4)
JPOS_MAX = np.tile([np.pi / 3, np.pi / 2, 3 * np.pi / 4], 4)
REFERENCE_DT = np.array([0.05] + [0.02] * 15 + [0.05, 0.05, 0.1, 0.2])      # landing_optimization.m:28


def binv(rpy):
    """Binv.m:13-17"""
    th, ps = rpy[1], rpy[2]
    return np.array([[np.cos(ps) / np.cos(th), np.sin(ps) / np.cos(th), 0.0], [-np.sin(ps), np.cos(ps), 0.0],
                     [np.cos(ps) * np.tan(th), np.sin(ps) * np.tan(th), 1.0]])


def dynamics_defects(X, U, dt, mass, Ib, Ib_inv):
    """:113-129  explicit-Euler defects, [12, N] in the script's order (v, omega, pos, rpy)"""
    N = U.shape[1]
    out = np.zeros((12, N))
    for k in range(N):
        q, qd, c, f = X[:6, k], X[6:, k], U[:12, k].reshape(4, 3), U[12:, k].reshape(4, 3)
        R = ro.rpy_to_rot_xyz(q[3:6])                                   # body -> world
        rdd = f.sum(axis=0) / mass + np.array([0.0, 0.0, -9.81])
        tau = sum(np.cross(c[l] - q[:3], f[l]) for l in range(4))
        omd = Ib_inv * (R.T @ tau - np.cross(qd[:3], Ib * qd[:3]))
        out[0:3, k] = X[9:12, k + 1] - qd[3:6] - rdd * dt[k]
        out[3:6, k] = X[6:9, k + 1] - qd[0:3] - omd * dt[k]
        out[6:9, k] = X[0:3, k + 1] - q[0:3] - qd[3:6] * dt[k]
        out[9:12, k] = X[3:6, k + 1] - q[3:6] - binv(q[3:6]) @ (R @ qd[0:3]) * dt[k]
    return out


def contact_rows(U):
    """:132-146  f_z >= 0, c_z >= 0, f_z c_z <= 1e-3 (LCP), |f_z (c+ - c)| <= 1e-3 (no slip): returns (f_z, c_z, f_z c_z, f_z dc [N-1, 4, 3])"""
    c, f = U[:12].T.reshape(-1, 4, 3), U[12:].T.reshape(-1, 4, 3)
    return f[:, :, 2], c[:, :, 2], f[:, :, 2] * c[:, :, 2], f[:-1, :, 2:3] * (c[1:] - c[:-1])


def friction_rows(U, mu):
    """:175-178  |f_x|, |f_y| <= 0.71 mu f_z: returns the four slacks (>= 0 when feasible), [N, 4, 4]"""
    f = U[12:].T.reshape(-1, 4, 3)
    lim = 0.71 * mu * f[:, :, 2]
    return np.stack([lim - f[:, :, 0], f[:, :, 0] + lim, lim - f[:, :, 1], f[:, :, 1] + lim], axis=-1)


def hip_relative(X, U):
    """:148-149  p_rel = c - (r + R hip), world frame, [N, 4, 3]"""
    N = U.shape[1]
    out = np.zeros((N, 4, 3))
    for k in range(N):
        R = ro.rpy_to_rot_xyz(X[3:6, k])
        out[k] = U[:12, k].reshape(4, 3) - (X[:3, k] + HIP @ R.T)
    return out


def kinematic_rows(X, U, jpos):
    """:166-171, :184-189  returns (c - FK([q; jpos]) [N, 12], leg torques J_f'(-R' f) [N, 12])"""
    N = U.shape[1]
    fk_err, tau = np.zeros((N, 12)), np.zeros((N, 12))
    for k in range(N):
        _, fk_err[k], tau[k] = ro.kinodyn_rows(X[:6, k], U[:12, k], U[12:, k], jpos[:, k])
    return fk_err, tau
