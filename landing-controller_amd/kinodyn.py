"""Host side of the kinodynamic refinement NLP (optimizations/landing/main_scripts/landing_optimization.m): the variable
layout, the bounds lbg / ubg in the row order of landing_kinodyn_nlp_eval (include/landing_nlp.h) with the script's values, its parameter
code (velocity-dependent kinematic box, :251 with test_scripts/kin_box_limits.m) and the terminal cost (:83-86).  g, Jacobian and Hessian
blocks come from the GPU (rbd.Rbd.kinodyn_nlp_eval / kinodyn_nlp_hess), the solve from rbd.Rbd.kinodyn_solve_host / kinodyn_solve_24
(landing_kinodyn_solve_batch, csrc/kd_solver_kernels.hip)."""
import numpy as np

INF = np.inf
SIDE_SIGN = (-1.0, 1.0, -1.0, 1.0)                                  # :156
TAU_MAX = np.array([18.0, 18.0, 3.0 * 9.33])                        # model.tauMax = gr .* motorTauMax = [6 6 9.33] * 3 (get_robot_model.m:237-241)
JPOS_MIN = np.tile([-np.pi / 3, -np.pi / 2, 0.0], 4)                # :246
JPOS_MAX = np.tile([np.pi / 3, np.pi / 2, 3 * np.pi / 4], 4)        # :247


def dims(N):
    """N intervals (the script's N - 1): nx, ng"""
    return 48 * N + 12, 48 + 141 * (N - 1) + 117


def pack_x(X, U, jpos):
    """X [12, N+1], U [24, N] = [c; f_grf], jpos [12, N] -> x (the script's declaration order X, jpos, U, :39-42)"""
    return np.concatenate([np.asarray(X).flatten(order="F"), np.asarray(jpos).flatten(order="F"), np.asarray(U).flatten(order="F")])


def unpack_x(x, N):
    x = np.asarray(x)
    a, b = 12 * (N + 1), 12 * (N + 1) + 12 * N
    return x[:a].reshape(12, N + 1, order="F"), x[b:].reshape(24, N, order="F"), x[a:b].reshape(12, N, order="F")


def kin_box_limits(v, direction):
    """test_scripts/kin_box_limits.m: adjustment of the kinematic box with the body-frame velocity"""
    box_max = 0.15 if direction == "x" else 0.25
    return abs(v * (box_max / 2.0)) if abs(v) < 2.0 else box_max


def kin_box_of(rpy0, v_world0):
    """:249-251  kin_box_val from the initial attitude and velocity"""
    r, p, y = rpy0
    rx = np.array([[1, 0, 0], [0, np.cos(r), np.sin(r)], [0, -np.sin(r), np.cos(r)]])
    ry = np.array([[np.cos(p), 0, -np.sin(p)], [0, 1, 0], [np.sin(p), 0, np.cos(p)]])
    rz = np.array([[np.cos(y), np.sin(y), 0], [-np.sin(y), np.cos(y), 0], [0, 0, 1]])
    vb = (rx.T @ ry.T @ rz.T).T @ np.asarray(v_world0, float)
    return kin_box_limits(vb[0], "x"), kin_box_limits(vb[1], "y")


def bounds(N, q_init, qd_init, c_init, kin_box, q_term_min=(-10, -10, 0.15, -0.1, -0.1, -10), q_term_max=(10, 10, 5, 0.1, 0.1, 10),
           qd_term_min=(-10, -10, -10, -.5, -.5, -.5), qd_term_max=(10, 10, 10, .5, .5, .5), z_min=0.075, l_leg_max=0.4,
           jpos_min=JPOS_MIN, jpos_max=JPOS_MAX, tau_max=TAU_MAX, comp_eps=1e-3, slip_eps=1e-3, fk_band=0.01, kin_box_y0=0.10):
    """lbg, ubg [ng] in the row order of landing_kinodyn_nlp_eval; defaults = the script's values (:208-258).  kin_box_y0: 0.10 in
    landing_optimization.m:150 (landing_kinodyn_form_default), 0.125 in generate_landingCtrller_KNITRO.m:154 (landing_kinodyn_form_knitro --
    what the 24-argument solver function uses)"""
    ng = dims(N)[1]
    lb, ub = np.zeros(ng), np.zeros(ng)
    lb[0:6] = ub[0:6] = q_init; lb[6:12] = ub[6:12] = qd_init; lb[12:24] = ub[12:24] = c_init
    lb[24:30] = q_term_min; ub[24:30] = INF; lb[30:36] = -INF; ub[30:36] = q_term_max
    lb[36:42] = qd_term_min; ub[36:42] = INF; lb[42:48] = -INF; ub[42:48] = qd_term_max
    kbx, kby = 0.125 + kin_box[0], kin_box_y0 + kin_box[1]
    for k in range(N):
        last = k == N - 1
        o = 48 + 141 * k
        r = o + 12                                                  # (Euler defects: 0 = 0)
        lb[r:r + 4] = 0.0; ub[r:r + 4] = INF; r += 4                 # f_z >= 0
        for l in range(4):
            lb[r] = 0.0; ub[r] = INF; r += 1                         # c_z >= 0
            lb[r] = -INF; ub[r] = comp_eps; r += 1                   # f_z c_z <= 1e-3
            if not last:
                lb[r:r + 3] = -INF; ub[r:r + 3] = slip_eps; r += 3
                lb[r:r + 3] = -slip_eps; ub[r:r + 3] = INF; r += 3
            lb[r] = -kbx; ub[r] = kbx; r += 1
            if SIDE_SIGN[l] < 0: lb[r], ub[r] = -kby, 0.05
            else: lb[r], ub[r] = -0.05, kby
            r += 1
            lb[r] = -0.4; ub[r] = -0.075; r += 1
            lb[r] = -INF; ub[r] = l_leg_max ** 2; r += 1
            lb[r:r + 3] = -np.asarray(tau_max); ub[r:r + 3] = tau_max; r += 3
        for s in range(4):                                           # friction: all four groups are `g1 - g2 <= 0` in Opti's canonical form (round 6)
            lb[r:r + 4] = -INF; ub[r:r + 4] = 0.0
            r += 4
        lb[r] = z_min; ub[r] = INF; r += 1
        lb[r:r + 12] = -fk_band; ub[r:r + 12] = INF; r += 12
        lb[r:r + 12] = -INF; ub[r:r + 12] = fk_band; r += 12
        lb[r:r + 12] = jpos_min; ub[r:r + 12] = INF; r += 12
        lb[r:r + 12] = -INF; ub[r:r + 12] = jpos_max; r += 12
        assert r == o + (117 if last else 141)
    return lb, ub


def terminal_cost(x, N, x_ref_end, QN=(0, 0, 100, 10, 10, 0, 10, 10, 10, 10, 10, 10)):
    """:83-86  (X(:,end) - Xref(:,end))' diag(QN) (...)  and its gradient in x (Hessian: 2 diag(QN) on X(:, N+1))"""
    e = np.asarray(x)[12 * N:12 * N + 12] - np.asarray(x_ref_end, float)
    g = np.zeros(dims(N)[0]); g[12 * N:12 * N + 12] = 2.0 * np.asarray(QN, float) * e
    return float(e @ (np.asarray(QN, float) * e)), g


# ---- the solve (landing_kinodyn_solve_batch, include/landing_nlp.h) ---------------------------------------------------------------------
QN_DEFAULT = (0, 0, 100, 10, 10, 0, 10, 10, 10, 10, 10, 10)          # :253
Q_TERM_REF = (0, 0, 0.25, 0, 0, 0)                                   # :228
C_REL_INIT = (0.2, 0.15, -0.3)                                       # :235  p_foot_rel of c_init
SIDE_SIGN_C = np.array([1, -1, 1, 1, 1, 1, -1, -1, 1, -1, 1, 1], float)      # :204


def rot_xyz(rpy):
    """rpyToRotMat_xyz.m:2  R = rx(r)' ry(p)' rz(y)'  (body -> world)"""
    r, p, y = rpy
    rx = np.array([[1, 0, 0], [0, np.cos(r), np.sin(r)], [0, -np.sin(r), np.cos(r)]])
    ry = np.array([[np.cos(p), 0, -np.sin(p)], [0, 1, 0], [np.sin(p), 0, np.cos(p)]])
    rz = np.array([[np.cos(y), np.sin(y), 0], [-np.sin(y), np.cos(y), 0], [0, 0, 1]])
    return rx.T @ ry.T @ rz.T


def c_init_of(q_init):
    """:232-236  initial foot positions under the hips of the initial pose"""
    q = np.asarray(q_init, float)
    R = rot_xyz(q[3:6])
    return np.concatenate([q[:3] + R @ (SIDE_SIGN_C[3 * l:3 * l + 3] * np.asarray(C_REL_INIT)) for l in range(4)])


def member_problem(N, q_init, qd_init, x_srbm, jpos_guess=None, **bound_kw):
    """One member of the refinement batch as the production callers pose it (landing_optimization.m:203-322, generate_training_data_automated.m
    :62-156): bounds, terminal cost data [QN | Xref(:, end)] and the initial guess x0 = [X*(:); jpos_guess; U*(:)] from the SRBM solution x_srbm
    ([X(:); U(:)], nx = 36N + 12).  jpos_guess None -> the data-generation caller's constant guess (0, -pi/4, pi/2) per leg (:143)."""
    q_init = np.asarray(q_init, float); qd_init = np.asarray(qd_init, float)
    kb = kin_box_of(q_init[3:6], qd_init[3:6])
    lb, ub = bounds(N, q_init, qd_init, c_init_of(q_init), kb, **bound_kw)
    xs = np.asarray(x_srbm, float)
    X = xs[:12 * (N + 1)].reshape(12, N + 1, order="F"); U = xs[12 * (N + 1):].reshape(24, N, order="F")
    jp = np.tile(np.tile([0.0, -np.pi / 4, np.pi / 2], 4).reshape(12, 1), (1, N)) if jpos_guess is None else np.asarray(jpos_guess, float).reshape(12, N)
    cost = np.concatenate([np.asarray(QN_DEFAULT, float), np.concatenate([Q_TERM_REF, np.zeros(6)])])
    return lb, ub, cost, pack_x(X, U, jp)


def make_args24(N, q_init, qd_init, x_srbm, dt, mass, Ib, Ib_inv, mu=0.75, l_leg_max=0.4, jpos_guess=None):
    """The 24 arguments of the reference's kinodynamic solver function for B drop states (generate_training_data_automated.m:62-156), MATLAB-shaped
    with a trailing batch axis; q_init, qd_init [B, 6], x_srbm [B, 36N+12] the SRBM solutions used as the initial guess."""
    q_init = np.atleast_2d(np.asarray(q_init, float)); qd_init = np.atleast_2d(np.asarray(qd_init, float)); xs = np.atleast_2d(np.asarray(x_srbm, float))
    B = q_init.shape[0]
    rep = lambda v: np.repeat(np.asarray(v, float).reshape(-1, 1), B, axis=1)
    Xref = np.zeros((12, N + 1, B)); x0 = np.zeros((48 * N + 12, B)); c_init = np.zeros((12, B)); kb = np.zeros((2, B))
    for b in range(B):
        for i in range(6):
            Xref[i, :, b] = np.linspace(q_init[b, i], Q_TERM_REF[i], N + 1); Xref[6 + i, :, b] = np.linspace(qd_init[b, i], 0.0, N + 1)
        c_init[:, b] = c_init_of(q_init[b]); kb[:, b] = kin_box_of(q_init[b, 3:6], qd_init[b, 3:6])
        jp = None if jpos_guess is None else jpos_guess[b]
        x0[:, b] = member_problem(N, q_init[b], qd_init[b], xs[b], jp)[3]
    return dict(Xref=Xref, Uref=None, dt=np.repeat(np.asarray(dt, float).reshape(1, N, 1), B, axis=2), q_min=rep([-10, -10, 0.075, -10, -10, -10]),
                q_max=rep([10, 10, 1.0, 10, 10, 10]), qd_min=rep([-10, -10, -10, -40, -40, -40]), qd_max=rep([10, 10, 10, 40, 40, 40]),
                q_init=q_init.T.copy(), qd_init=qd_init.T.copy(), c_init=c_init, q_term_min=rep([-10, -10, 0.15, -0.1, -0.1, -10]),
                q_term_max=rep([10, 10, 5, 0.1, 0.1, 10]), qd_term_min=rep([-10, -10, -10, -.5, -.5, -.5]), qd_term_max=rep([10, 10, 10, .5, .5, .5]),
                QN=rep(QN_DEFAULT), x0=x0, jpos_min=rep(JPOS_MIN), jpos_max=rep(JPOS_MAX), kin_box=kb, mu=rep([mu]), l_leg_max=rep([l_leg_max]),
                mass=rep([mass]), Ib=rep(Ib), Ib_inv=rep(Ib_inv))


KNITRO_PARAMS = ("Xref", "dt", "q_init", "qd_init", "c_init", "jpos_min", "jpos_max", "q_term_min", "q_term_max", "qd_term_min", "qd_term_max", "q_min", "QN", "mu", "l_leg_max",
                 "mass", "Ib", "Ib_inv", "kin_box")


def knitro_param_offsets(N):
    """offsets of the ACTIVE Opti parameters of generate_landingCtrller_KNITRO.m (:51-82) inside p, declaration order (landing_kinodyn_casadi_offsets)"""
    lens = (12 * (N + 1), N, 6, 6, 12, 12, 12, 6, 6, 6, 6, 6, 12, 1, 1, 1, 3, 3, 2)
    off, o = {}, 0
    for n, l in zip(KNITRO_PARAMS, lens):
        off[n] = (o, o + l); o += l
    return off, o


def pack_params_knitro(N, **kw):
    """p of the CasADi-external face (landingCtrller_KNITRO_mi355x.so) from the script's parameter values; Xref 12 x (N+1) column-major"""
    off, n = knitro_param_offsets(N)
    p = np.zeros(n)
    for name, (a, b) in off.items():
        v = np.asarray(kw[name], float)
        p[a:b] = v.flatten(order="F") if v.ndim > 1 else v
    return p
