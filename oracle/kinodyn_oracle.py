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


# ---- the whole NLP in the layout of include/landing_nlp.h (landing_kinodyn_nlp_eval): x = [X(:); jpos(:); U(:)], rows in the script's order ----
def nlp_dims(N):
    return 48 * N + 12, 48 + 141 * (N - 1) + 117


def stage_rows(w, dt, last, mass, Ib, Ib_inv, mu):
    """rows of one interval (landing_optimization.m:113-189) for w = [X_k, c_k, f_k, jpos_k, X_k+1, c_k+1] (72): 141 values, 117 when `last`"""
    X, c, f, jp, Xn, cn = w[:12], w[12:24], w[24:36], w[36:48], w[48:60], w[60:72]
    pos, rpy, om, v = X[0:3], X[3:6], X[6:9], X[9:12]
    R = ro.rpy_to_rot_xyz(rpy)
    cf, ff = c.reshape(4, 3), f.reshape(4, 3)
    rdd = ff.sum(axis=0) / mass + np.array([0.0, 0.0, -9.81])
    tau = sum(np.cross(cf[l] - pos, ff[l]) for l in range(4))
    omd = Ib_inv * (R.T @ tau - np.cross(om, Ib * om))
    out = [Xn[9:12] - v - rdd * dt, Xn[6:9] - om - omd * dt, Xn[0:3] - pos - v * dt, Xn[3:6] - rpy - binv(rpy) @ (R @ om) * dt, ff[:, 2]]
    fk, fk_err, tq = ro.kinodyn_rows(X[:6], c, f, jp)
    for l in range(4):
        out.append([cf[l, 2], ff[l, 2] * cf[l, 2]])
        if not last:
            d = ff[l, 2] * (cn[3 * l:3 * l + 3] - cf[l])
            out += [d, d]
        pr = cf[l] - (pos + R @ HIP[l])
        out += [pr, [pr @ pr], tq[3 * l:3 * l + 3]]
    km = 0.71 * mu
    # (the two `>=` friction rows in Opti's canonical form: neither side of `f_xy >= -km f_z` is parametric, so the row is (-km f_z) - f_xy <= 0, optistack_internal.cpp:793-806)
    out += [ff[:, 0] - km * ff[:, 2], -km * ff[:, 2] - ff[:, 0], ff[:, 1] - km * ff[:, 2], -km * ff[:, 2] - ff[:, 1], [pos[2]], fk_err, fk_err, jp, jp]
    return np.concatenate([np.atleast_1d(np.asarray(o, float)) for o in out])


def w_index(N, k, j):
    oJ, oU = 12 * (N + 1), 12 * (N + 1) + 12 * N
    if j < 12: return 12 * k + j
    if j < 24: return oU + 24 * k + (j - 12)
    if j < 36: return oU + 24 * k + 12 + (j - 24)
    if j < 48: return oJ + 12 * k + (j - 36)
    if j < 60: return 12 * (k + 1) + (j - 48)
    return oU + 24 * (k + 1) + (j - 60) if k + 1 < N else -1


def gather_w(x, N, k):
    return np.array([x[w_index(N, k, j)] if w_index(N, k, j) >= 0 else 0.0 for j in range(72)])


def nlp_g(x, N, dt, mass, Ib, Ib_inv, mu):
    nx, ng = nlp_dims(N)
    assert x.shape == (nx,)
    oU = 12 * (N + 1) + 12 * N
    g = [x[0:12], x[oU:oU + 12], x[12 * N:12 * N + 6], x[12 * N:12 * N + 6], x[12 * N + 6:12 * N + 12], x[12 * N + 6:12 * N + 12]]
    for k in range(N):
        g.append(stage_rows(gather_w(x, N, k), dt[k], k == N - 1, mass, Ib, Ib_inv, mu))
    g = np.concatenate(g)
    assert g.shape == (ng,)
    return g


def stage_jacobian(w, dt, last, mass, Ib, Ib_inv, mu, h=2e-3):
    """d rows / d w [rows, 72] by Richardson-extrapolated central differences (steps h, h/2: error O(h^4))"""
    def cd(hh):
        J = np.zeros((117 if last else 141, 72))
        for j in range(72):
            e = np.zeros(72); e[j] = hh
            J[:, j] = (stage_rows(w + e, dt, last, mass, Ib, Ib_inv, mu) - stage_rows(w - e, dt, last, mass, Ib, Ib_inv, mu)) / (2 * hh)
        return J
    return (4.0 * cd(0.5 * h) - cd(h)) / 3.0


def pack_x(X, U, J):
    """X [12, N+1], U [24, N] (c; f), jpos [12, N] -> x in the layout above"""
    return np.concatenate([X.flatten(order="F"), J.flatten(order="F"), U.flatten(order="F")])


def nlp_jacobian(x, N, dt, mass, Ib, Ib_inv, mu):
    """dense Jacobian [ng, nx] of nlp_g from the Richardson stage blocks (boundary rows: coordinate picks)"""
    nx, ng = nlp_dims(N)
    Jf = np.zeros((ng, nx))
    oU = 12 * (N + 1) + 12 * N
    for i in range(12):
        Jf[i, i] = 1.0; Jf[12 + i, oU + i] = 1.0
    for i in range(6):
        Jf[24 + i, 12 * N + i] = 1.0; Jf[30 + i, 12 * N + i] = 1.0; Jf[36 + i, 12 * N + 6 + i] = 1.0; Jf[42 + i, 12 * N + 6 + i] = 1.0
    for k in range(N):
        last = k == N - 1
        Jk = stage_jacobian(gather_w(x, N, k), dt[k], last, mass, Ib, Ib_inv, mu)
        for j in range(72):
            ix = w_index(N, k, j)
            if ix >= 0:
                Jf[48 + 141 * k:48 + 141 * k + Jk.shape[0], ix] += Jk[:, j]
    return Jf


def kkt(x, lam_g, N, dt, mass, Ib, Ib_inv, mu, lbg, ubg, grad_f):
    """unscaled KKT residuals of (x, lam_g) in the convention of the SRBM oracle (lo_kkt: lam > 0 pushes against ubg, lam < 0 against lbg):
    (primal infeasibility, dual infeasibility |grad f + J' lam|_inf, complementarity max |lam_r| * distance to the bound it pushes against)"""
    g = nlp_g(x, N, dt, mass, Ib, Ib_inv, mu)
    pr = float(np.maximum(np.maximum(lbg - g, g - ubg), 0.0).max())
    du = float(np.abs(np.asarray(grad_f) + nlp_jacobian(x, N, dt, mass, Ib, Ib_inv, mu).T @ lam_g).max())
    eq = lbg == ubg
    dist = np.where(lam_g > 0, np.where(np.isfinite(ubg), np.maximum(ubg - g, 0.0), np.inf), np.where(np.isfinite(lbg), np.maximum(g - lbg, 0.0), np.inf))
    with np.errstate(invalid="ignore"):
        co = np.where(eq | (lam_g == 0.0), 0.0, np.abs(lam_g) * dist)
    return pr, du, float(np.nanmax(co))


# ---- the same rows for MANY stages at once (numpy over a leading batch axis) -------------------------------------------------------------
# Used to certify whole batches of solver results: the row functions above, restated with arrays [n, ...] in place of scalars, on any
# numpy dtype.  On complex arguments they give the Jacobian by the complex-step method (every row is an analytic function of w: products,
# sin, cos, tan): d row / d w_j = Im row(w + i h e_j) / h with h = 1e-30 is exact to rounding -- no difference quotient, no step-size error,
# and a different derivative mechanism than the kernels' dual numbers.  tests/test_kd_solver_cpu.py pins the batch form to the scalar one.
_CH = 1e-30


def _rot_b(axis, t):
    c, s = np.cos(t), np.sin(t)
    o, z = np.ones_like(c), np.zeros_like(c)
    if axis == 0: rows = [[o, z, z], [z, c, s], [z, -s, c]]
    elif axis == 1: rows = [[c, z, -s], [z, o, z], [s, z, c]]
    else: rows = [[c, s, z], [-s, c, z], [z, z, o]]
    return np.stack([np.stack(r, axis=-1) for r in rows], axis=-2)


def _skew_b(r):
    z = np.zeros_like(r[..., 0])
    return np.stack([np.stack([z, -r[..., 2], r[..., 1]], -1), np.stack([r[..., 2], z, -r[..., 0]], -1), np.stack([-r[..., 1], r[..., 0], z], -1)], -2)


def _jcalc_b(jtype, q):
    """jcalc.m:22-40 on an array of joint positions: XJ [n, 6, 6]"""
    n = q.shape[0]
    X = np.zeros((n, 6, 6), q.dtype)
    if jtype[0] == "R":
        E = _rot_b("xyz".index(jtype[1]), q)
        X[:, :3, :3] = E; X[:, 3:, 3:] = E
    else:
        r = np.zeros((n, 3), q.dtype); r[:, "xyz".index(jtype[1])] = q
        X[:, :3, :3] = np.eye(3); X[:, 3:, 3:] = np.eye(3); X[:, 3:, :3] = -_skew_b(r)
    return X


def fk_batch(q18):
    """get_forward_kin_foot.m on rows of [q6, jpos] ([n, 18]): foot positions [n, 12] (6 x 6 Pluecker chain, as rbd_oracle.forward_kin_foot)"""
    M = ro.quad3d_model()
    X0 = [None] * ro.NB
    for i in range(ro.NB):
        Xup = _jcalc_b(M["jtype"][i], q18[:, i]) @ M["Xtree"][i]
        pa = M["parent"][i]
        X0[i] = Xup if pa == 0 else Xup @ X0[pa - 1]
    out = []
    for leg in range(4):
        X = M["Xfoot"][leg] @ X0[M["b_foot"][leg] - 1]
        E = X[:, :3, :3]
        S = -(np.swapaxes(E, 1, 2) @ X[:, 3:, :3])
        out.append(np.stack([S[:, 2, 1], S[:, 0, 2], S[:, 1, 0]], -1))
    return np.concatenate(out, -1)


def stage_rows_batch(W, dt, last, mass, Ib, Ib_inv, mu):
    """stage_rows for n stages at once: W [n, 72], dt [n] -> [n, 141] (117 when `last`)"""
    W = np.asarray(W); n = W.shape[0]; dt = np.asarray(dt, float).reshape(n, 1)
    X, c, f, jp, Xn, cn = W[:, :12], W[:, 12:24], W[:, 24:36], W[:, 36:48], W[:, 48:60], W[:, 60:72]
    pos, rpy, om, v = X[:, 0:3], X[:, 3:6], X[:, 6:9], X[:, 9:12]
    R = np.swapaxes(_rot_b(0, rpy[:, 0]), 1, 2) @ np.swapaxes(_rot_b(1, rpy[:, 1]), 1, 2) @ np.swapaxes(_rot_b(2, rpy[:, 2]), 1, 2)     # rpyToRotMat_xyz.m:2
    cf, ff = c.reshape(n, 4, 3), f.reshape(n, 4, 3)
    rdd = ff.sum(axis=1) / mass + np.array([0.0, 0.0, -9.81])
    tau = np.cross(cf - pos[:, None, :], ff).sum(axis=1)
    omd = np.asarray(Ib_inv) * ((np.swapaxes(R, 1, 2) @ tau[:, :, None])[:, :, 0] - np.cross(om, np.asarray(Ib) * om))
    th, ps = rpy[:, 1], rpy[:, 2]
    z, o = np.zeros_like(th), np.ones_like(th)
    Bi = np.stack([np.stack([np.cos(ps) / np.cos(th), np.sin(ps) / np.cos(th), z], -1), np.stack([-np.sin(ps), np.cos(ps), z], -1),
                   np.stack([np.cos(ps) * np.tan(th), np.sin(ps) * np.tan(th), o], -1)], -2)                                 # Binv.m:13-17
    ed = (Bi @ (R @ om[:, :, None]))[:, :, 0]
    out = [Xn[:, 9:12] - v - rdd * dt, Xn[:, 6:9] - om - omd * dt, Xn[:, 0:3] - pos - v * dt, Xn[:, 3:6] - rpy - ed * dt, ff[:, :, 2]]
    fk = fk_batch(np.concatenate([X[:, :6], jp], axis=1))
    fk_err = c - fk
    Rw2b = np.swapaxes(R, 1, 2)
    l1, l2, l3, l4 = 0.062, 0.209, 0.195, 0.004                                                                              # get_foot_jacobians_mc.m:5-8
    for l in range(4):
        out.append(np.stack([cf[:, l, 2], ff[:, l, 2] * cf[:, l, 2]], -1))
        if not last:
            d = ff[:, l, 2:3] * (cn[:, 3 * l:3 * l + 3] - cf[:, l])
            out += [d, d]
        pr = cf[:, l] - (pos + (R @ HIP[l])[:, :])
        out += [pr, (pr * pr).sum(axis=1, keepdims=True)]
        q1, q2, q3 = jp[:, 3 * l], jp[:, 3 * l + 1], jp[:, 3 * l + 2]
        s1, s2, s3, c1, c2, c3 = np.sin(q1), np.sin(q2), np.sin(q3), np.cos(q1), np.cos(q2), np.cos(q3)
        c23 = c2 * c3 - s2 * s3; s23 = s2 * c3 + c2 * s3
        ss = (-1.0, 1.0, -1.0, 1.0)[l]
        zz = np.zeros_like(q1)
        J = np.stack([np.stack([zz, l3 * c23 + l2 * c2, l3 * c23], -1),
                      np.stack([l3 * c1 * c23 + l2 * c1 * c2 - (l1 + l4) * s1 * ss, -l3 * s1 * s23 - l2 * s1 * s2, -l3 * s1 * s23], -1),
                      np.stack([l3 * s1 * c23 + l2 * c2 * s1 + (l1 + l4) * ss * c1, l3 * c1 * s23 + l2 * c1 * s2, l3 * c1 * s23], -1)], -2)
        fb = -(Rw2b @ ff[:, l, :, None])
        out.append((np.swapaxes(J, 1, 2) @ fb)[:, :, 0])
    km = 0.71 * mu
    out += [ff[:, :, 0] - km * ff[:, :, 2], -km * ff[:, :, 2] - ff[:, :, 0], ff[:, :, 1] - km * ff[:, :, 2], -km * ff[:, :, 2] - ff[:, :, 1], pos[:, 2:3], fk_err, fk_err, jp, jp]
    return np.concatenate(out, axis=1)


def _w_map(N):
    """x index of w[j] of interval k ([N, 72]; -1 where the variable does not exist)"""
    return np.array([[w_index(N, k, j) for j in range(72)] for k in range(N)])


def nlp_g_batch(Xs, N, dt, mass, Ib, Ib_inv, mu):
    """nlp_g for B members at once: Xs [B, nx] -> [B, ng]"""
    Xs = np.asarray(Xs); B = Xs.shape[0]
    wm = _w_map(N)
    oU = 12 * (N + 1) + 12 * N
    g = [Xs[:, 0:12], Xs[:, oU:oU + 12], Xs[:, 12 * N:12 * N + 6], Xs[:, 12 * N:12 * N + 6], Xs[:, 12 * N + 6:12 * N + 12], Xs[:, 12 * N + 6:12 * N + 12]]
    for k in range(N):
        W = np.where(wm[k] >= 0, Xs[:, np.maximum(wm[k], 0)], 0.0)
        g.append(stage_rows_batch(W, np.full(B, dt[k]), k == N - 1, mass, Ib, Ib_inv, mu))
    return np.concatenate(g, axis=1)


def grad_lagrangian_batch(Xs, Lam, N, dt, mass, Ib, Ib_inv, mu, grad_f):
    """grad f + J' lam for B members ([B, nx]), J by the complex-step method, interval by interval"""
    Xs = np.asarray(Xs, float); Lam = np.asarray(Lam, float); B = Xs.shape[0]
    nx, ng = nlp_dims(N)
    wm = _w_map(N)
    oU = 12 * (N + 1) + 12 * N
    out = np.array(grad_f, float).reshape(B, nx).copy()
    out[:, 0:12] += Lam[:, 0:12]; out[:, oU:oU + 12] += Lam[:, 12:24]
    out[:, 12 * N:12 * N + 6] += Lam[:, 24:30] + Lam[:, 30:36]; out[:, 12 * N + 6:12 * N + 12] += Lam[:, 36:42] + Lam[:, 42:48]
    for k in range(N):
        last = k == N - 1
        nr = 117 if last else 141
        W = np.where(wm[k] >= 0, Xs[:, np.maximum(wm[k], 0)], 0.0)                      # [B, 72]
        Wc = np.repeat(W[:, None, :], 72, axis=1).astype(complex)                      # [B, 72 directions, 72]
        Wc[:, np.arange(72), np.arange(72)] += 1j * _CH
        rows = stage_rows_batch(Wc.reshape(B * 72, 72), np.full(B * 72, dt[k]), last, mass, Ib, Ib_inv, mu)
        Jk = (rows.imag / _CH).reshape(B, 72, nr)                                       # Jk[b, j, r] = d row_r / d w_j
        lam_k = Lam[:, 48 + 141 * k:48 + 141 * k + nr]
        contrib = np.einsum("bjr,br->bj", Jk, lam_k)
        for j in range(72):
            if wm[k, j] >= 0:
                out[:, wm[k, j]] += contrib[:, j]
    return out


def kkt_batch(Xs, Lam, N, dt, mass, Ib, Ib_inv, mu, lbg, ubg, grad_f):
    """kkt() for B members: [B, 3] = (pr_inf, du_inf, compl) in the convention of the SRBM oracle (lo_kkt)"""
    Xs = np.asarray(Xs, float); Lam = np.asarray(Lam, float); lbg = np.asarray(lbg, float); ubg = np.asarray(ubg, float)
    g = nlp_g_batch(Xs, N, dt, mass, Ib, Ib_inv, mu)
    pr = np.maximum(np.maximum(lbg - g, g - ubg), 0.0).max(axis=1)
    du = np.abs(grad_lagrangian_batch(Xs, Lam, N, dt, mass, Ib, Ib_inv, mu, grad_f)).max(axis=1)
    eq = lbg == ubg
    dist = np.where(Lam > 0, np.where(np.isfinite(ubg), np.maximum(ubg - g, 0.0), np.inf), np.where(np.isfinite(lbg), np.maximum(g - lbg, 0.0), np.inf))
    with np.errstate(invalid="ignore"):
        co = np.where(eq | (Lam == 0.0), 0.0, np.abs(Lam) * dist)
    return np.stack([pr, du, np.nanmax(co, axis=1)], axis=1)
