"""Host-side mirror of how the reference's callers build the inputs of the landing solver function.

The unit of work is one call of the 21-input solver function
``landingCtrller_IPOPT(Xref, Uref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, q_term_min,
q_term_max, qd_term_min, qd_term_max, QN, x0, mu, l_leg_max, f_max, mass, Ib, Ib_inv) -> (x*, f*)``
(generate_landingCtrller_IPOPT.m:323-327).  Everything here is numpy; a batch simply carries a
leading member dimension.

Layouts:
  x = [X(:); U(:)], X 12x(N+1) column-major = [pos rpy omega_body v_world], U 24xN = [c(12) f(12)]
  p = [Xref(:) (12(N+1)); dt (N); q_min q_max qd_min qd_max q_init qd_init q_term_min q_term_max
       qd_term_min qd_term_max (6 each); QN (12); mu; l_leg_max; f_max; mass; Ib(3); Ib_inv(3)]
      (Uref is an inactive Opti parameter and is not part of p.)
"""
from dataclasses import dataclass

import numpy as np

from .constants import HIP_SRBM, robot_constants


def nx(N):
    return 36 * N + 12


def ng(N):
    return 104 * N + 12


def n_p(N):
    return 13 * N + 94


def nnz_jac(N):
    return 36 + 385 * (N - 1) + 313


def nnz_hess(N):
    return 177 * N + 12 * (N - 1) + 12


def param_offsets(N):
    b = 12 * (N + 1) + N
    names = ["q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "q_term_min", "q_term_max",
             "qd_term_min", "qd_term_max"]
    o = {"Xref": 0, "dt": 12 * (N + 1)}
    for i, n in enumerate(names):
        o[n] = b + 6 * i
    o.update(QN=b + 60, mu=b + 72, l_leg_max=b + 73, f_max=b + 74, mass=b + 75, Ib=b + 76, Ib_inv=b + 79,
             np=b + 82)
    return o


def rpy_to_rot_xyz(rpy):
    """rpyToRotMat_xyz.m: rx(r)' * ry(p)' * rz(y)' (used by the callers for sampling/refs only)."""
    r, p, y = rpy
    cx, sx, cy, sy, cz, sz = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def rpy_to_rot(rpy):
    """rpyToRotMat.m:2: rz(y)' * ry(p)' * rx(r)' (the rotation used inside the NLP)."""
    r, p, y = rpy
    cx, sx, cy, sy, cz, sz = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


SIDE_SIGN = np.array([1, -1, 1, 1, 1, 1, -1, -1, 1, -1, 1, 1], float)


@dataclass
class CallerConstants:
    """Fixed arguments of the batch callers (generate_training_data_automated.m:62-102,
    main_scripts/landing_optimization.m:219-258)."""
    q_min: tuple = (-10, -10, 0.075, -10, -10, -10)
    q_max: tuple = (10, 10, 1.0, 10, 10, 10)
    qd_min: tuple = (-10, -10, -10, -40, -40, -40)
    qd_max: tuple = (10, 10, 10, 40, 40, 40)
    q_term_min: tuple = (-10, -10, 0.15, -0.1, -0.1, -10)
    q_term_max: tuple = (10, 10, 5, 0.1, 0.1, 10)
    qd_term_min: tuple = (-10, -10, -10, -0.5, -0.5, -0.5)
    qd_term_max: tuple = (10, 10, 10, 0.5, 0.5, 0.5)
    q_term_ref: tuple = (0, 0, 0.25, 0, 0, 0)
    qd_term_ref: tuple = (0, 0, 0, 0, 0, 0)
    c_ref: tuple = (0.2, 0.2, -0.3)
    QN: tuple = (0, 0, 100, 10, 10, 0, 10, 10, 10, 10, 10, 10)
    mu: float = 0.75
    l_leg_max: float = 0.4
    f_max: float = 500.0
    td_nom: float = 0.35


# The time grid every production caller of the N=20 solver function uses (main_scripts/landing_optimization.m:28,
# generate_data/generate_training_data_automated.m:28, generate_data/nn_warmstart.m:49): 20 intervals, 0.75 s, fine steps
# around touch-down.  `dt_grid="reference"` of the builders below selects it (N must be 20); "uniform" = T/N
# (analysis/eval_SRBM_CCC.m:22-24, the N=41 script and SURVEY 8(d)'s synthetic bench workload).
REFERENCE_DT_GRID = np.array([0.05] + [0.02] * 15 + [0.05, 0.05, 0.1, 0.2])

# Drop-state sampling laws of the two batch callers (the third column of numbers is what differs):
#   "main"    main_scripts/landing_optimization.m:207-208:            v_xy = 1.0 (2U-1),  v_z = -4.5 U - 0.5, f_max 300 (:258)
#   "datagen" generate_data/generate_training_data_automated.m:47,50: v_xy = 1.75 (2U-1), v_z = -3 U - 3,     f_max 500 (:102)
DROP_LAWS = {"main": (1.0, -4.5, -0.5), "datagen": (1.75, -3.0, -3.0)}


def dt_of(N, T, dt_grid="uniform"):
    """dt[N] of a caller: "uniform" (T/N) or "reference" (the production grid above, N = 20 only; T is ignored)."""
    if isinstance(dt_grid, str):
        if dt_grid == "uniform":
            return np.full(N, T / N)
        if dt_grid == "reference":
            if N != len(REFERENCE_DT_GRID):
                raise ValueError("the reference's production time grid has %d intervals (N = %d asked)" % (len(REFERENCE_DT_GRID), N))
            return REFERENCE_DT_GRID.copy()
        raise ValueError("dt_grid: 'uniform', 'reference' or an array of N step lengths")
    dt = np.asarray(dt_grid, float).reshape(-1)
    if dt.size != N or not (dt > 0).all():
        raise ValueError("dt_grid must hold N positive step lengths")
    return dt.copy()


def sample_drop_states(B, seed, dt1, consts=None, law="main"):
    """Random drop states of the batch callers (DROP_LAWS; the reference never seeds ``rand``; the seed is ours).
    Returns q_init[B,6], qd_init[B,6]."""
    c = consts or CallerConstants()
    vxy, vz_a, vz_b = DROP_LAWS[law]
    rng = np.random.default_rng(seed)
    u = rng.random((B, 9))
    q = np.zeros((B, 6))
    qd = np.zeros((B, 6))
    q[:, 3] = 0.25 * (2 * u[:, 0] - 1)
    q[:, 4] = (np.pi / 3) * (2 * u[:, 1] - 1)
    q[:, 5] = 0.25 * (2 * u[:, 2] - 1)
    qd[:, 0:3] = 0.5 * (2 * u[:, 3:6] - 1)
    qd[:, 3:5] = vxy * (2 * u[:, 6:8] - 1)
    qd[:, 5] = vz_a * u[:, 8] + vz_b
    for b in range(B):
        R = rpy_to_rot_xyz(q[b, 3:6])
        hip_z = (R @ HIP_SRBM.T)[2, :]
        q[b, 2] = c.td_nom + abs(hip_z.min()) + abs(dt1 * qd[b, 5])
    return q, qd


def reference_trajectories(N, q_init, qd_init, consts=None):
    """Xref[12,N+1], Uref[24,N] of one member (generate_training_data_automated.m:105-119)."""
    c = consts or CallerConstants()
    Xref = np.zeros((12, N + 1))
    for i in range(6):
        Xref[i] = np.linspace(q_init[i], c.q_term_ref[i], N + 1)
        Xref[6 + i] = np.linspace(qd_init[i], c.qd_term_ref[i], N + 1)
    c_ref = SIDE_SIGN * np.tile(np.asarray(c.c_ref, float), 4)
    Uref = np.zeros((24, N))
    for k in range(N):
        R = rpy_to_rot_xyz(Xref[3:6, k])
        for leg in range(4):
            Uref[3 * leg:3 * leg + 3, k] = Xref[0:3, k] + R @ c_ref[3 * leg:3 * leg + 3]
    return Xref, Uref


def pack_params(N, Xref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, q_term_min, q_term_max,
                qd_term_min, qd_term_max, QN, mu, l_leg_max, f_max, mass, Ib, Ib_inv):
    """p vector in Opti's parameter order (generate_landingCtrller_IPOPT.m:51-75)."""
    parts = [np.asarray(Xref, float).reshape(12, N + 1).flatten(order="F"), np.asarray(dt, float).reshape(N)]
    for v in (q_min, q_max, qd_min, qd_max, q_init, qd_init, q_term_min, q_term_max, qd_term_min, qd_term_max):
        parts.append(np.asarray(v, float).reshape(6))
    parts.append(np.asarray(QN, float).reshape(12))
    parts.append(np.array([mu, l_leg_max, f_max, mass], float))
    parts.append(np.asarray(Ib, float).reshape(3))
    parts.append(np.asarray(Ib_inv, float).reshape(3))
    p = np.concatenate(parts)
    assert p.size == n_p(N)
    return p


def n_p_ccc(N):
    """length of the N=41 script's own parameter vector (generate_quadruped_SRBM_CCC.m:49-71): 37N + 112"""
    return 37 * N + 112


def pack_params_ccc(N, Xref, Uref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, q_term_min, q_term_max,
                    qd_term_min, qd_term_max, QX, QN, Qc, Qf, mu, l_leg_max, f_max, mass, Ib, Ib_inv):
    """p of the N=41 script in Opti's order of its active parameters (c_init is declared but unused: dropped)"""
    parts = [np.asarray(Xref, float).reshape(12, N + 1).flatten(order="F"), np.asarray(Uref, float).reshape(24, N).flatten(order="F"),
             np.asarray(dt, float).reshape(N)]
    for v in (q_min, q_max, qd_min, qd_max, q_init, qd_init, q_term_min, q_term_max, qd_term_min, qd_term_max):
        parts.append(np.asarray(v, float).reshape(6))
    parts += [np.asarray(QX, float).reshape(12), np.asarray(QN, float).reshape(12), np.asarray(Qc, float).reshape(3), np.asarray(Qf, float).reshape(3)]
    parts.append(np.array([mu, l_leg_max, f_max, mass], float))
    parts += [np.asarray(Ib, float).reshape(3), np.asarray(Ib_inv, float).reshape(3)]
    p = np.concatenate(parts)
    assert p.size == n_p_ccc(N)
    return p


def ccc_from_ipopt_params(N, p, Uref, QX, Qc, Qf):
    """the N=41 script's parameter vector holding the same problem as an IPOPT-variant p (np = 13N+94) plus Uref and the running-cost weights"""
    o = param_offsets(N)
    six = [p[o[n]:o[n] + 6] for n in ("q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "q_term_min", "q_term_max", "qd_term_min", "qd_term_max")]
    return pack_params_ccc(N, p[:12 * (N + 1)].reshape(12, N + 1, order="F"), Uref, p[o["dt"]:o["dt"] + N], *six, QX, p[o["QN"]:o["QN"] + 12], Qc, Qf,
                           p[o["mu"]], p[o["l_leg_max"]], p[o["f_max"]], p[o["mass"]], p[o["Ib"]:o["Ib"] + 3], p[o["Ib_inv"]:o["Ib_inv"] + 3])


def make_member(N, T, q_init, qd_init, consts=None, dt_grid="uniform"):
    """(p, x0, Xref, Uref) for one drop state with the callers' fixed arguments; x0=[Xref(:);Uref(:)]."""
    c = consts or CallerConstants()
    mass, Ib, Ib_inv = robot_constants()
    dt = dt_of(N, T, dt_grid)
    Xref, Uref = reference_trajectories(N, q_init, qd_init, c)
    p = pack_params(N, Xref, dt, c.q_min, c.q_max, c.qd_min, c.qd_max, q_init, qd_init, c.q_term_min,
                    c.q_term_max, c.qd_term_min, c.qd_term_max, c.QN, c.mu, c.l_leg_max, c.f_max, mass, Ib, Ib_inv)
    x0 = np.concatenate([Xref.flatten(order="F"), Uref.flatten(order="F")])
    return p, x0, Xref, Uref


def make_args21(B, N=40, T=0.6, seed=20211, consts=None, dt_grid="uniform", law="main"):
    """The 21 arguments of the solver function for B sampled drop states, shaped as the MATLAB callers hold them with a
    trailing batch axis (generate_training_data_automated.m:62-136): dict name -> array."""
    c = consts or CallerConstants()
    mass, Ib, Ib_inv = robot_constants()
    dtv = dt_of(N, T, dt_grid)
    q, qd = sample_drop_states(B, seed, dtv[0], c, law)
    Xref = np.zeros((12, N + 1, B)); Uref = np.zeros((24, N, B)); x0 = np.zeros((nx(N), B))
    for b in range(B):
        Xref[:, :, b], Uref[:, :, b] = reference_trajectories(N, q[b], qd[b], c)
        x0[:, b] = np.concatenate([Xref[:, :, b].flatten(order="F"), Uref[:, :, b].flatten(order="F")])
    rep = lambda v: np.repeat(np.asarray(v, float).reshape(-1, 1), B, axis=1)
    return dict(Xref=Xref, Uref=Uref, dt=np.repeat(dtv.reshape(1, N, 1), B, axis=2), q_min=rep(c.q_min), q_max=rep(c.q_max), qd_min=rep(c.qd_min),
                qd_max=rep(c.qd_max), q_init=q.T.copy(), qd_init=qd.T.copy(), q_term_min=rep(c.q_term_min), q_term_max=rep(c.q_term_max),
                qd_term_min=rep(c.qd_term_min), qd_term_max=rep(c.qd_term_max), QN=rep(c.QN), x0=x0, mu=rep([c.mu]),
                l_leg_max=rep([c.l_leg_max]), f_max=rep([c.f_max]), mass=rep([mass]), Ib=rep(Ib), Ib_inv=rep(Ib_inv))


def make_batch(B, N=40, T=0.6, seed=20211, consts=None, dt_grid="uniform", law="main"):
    """Drop-state batch: returns P[B,np], X0[B,nx], q_init, qd_init.  Defaults = the synthetic workload of SURVEY 8(d)
    (uniform T/N grid, law "main"); ``dt_grid="reference"`` poses the problem the reference's production callers pose at
    N = 20 (REFERENCE_DT_GRID; touch-down height from ITS first step, :52-60), ``law`` picks their sampling law (DROP_LAWS)."""
    dtv = dt_of(N, T, dt_grid)
    q, qd = sample_drop_states(B, seed, dtv[0], consts, law)
    P = np.zeros((B, n_p(N)))
    X0 = np.zeros((B, nx(N)))
    for b in range(B):
        P[b], X0[b], _, _ = make_member(N, T, q[b], qd[b], consts, dtv)
    return P, X0, q, qd


def production_constants(law="main"):
    """CallerConstants of the two production callers: they differ in f_max only (landing_optimization.m:258 -> 300,
    generate_training_data_automated.m:102 -> 500)."""
    return CallerConstants(f_max=300.0 if law == "main" else 500.0)


def split_solution(N, x):
    """X*[12,N+1], U*[24,N] (generate_training_data_automated.m:139-141)."""
    x = np.asarray(x)
    return x[:12 * (N + 1)].reshape(12, N + 1, order="F"), x[12 * (N + 1):].reshape(24, N, order="F")


def reference_default_problem(N=20, T=0.6):
    """The fixed drop the generator script itself solves for verification, BASELINE configs[0]
    (generate_landingCtrller_IPOPT.m:173-224,332-338): returns (p, x0) with x0 = [Xref(:); Uref(:)]."""
    mass, Ib, Ib_inv = robot_constants()
    q_init = np.array([0, 0, 0.6, 0, np.pi / 4, -np.pi / 6]); qd_init = np.array([0, 4, 5, 1.3, -2, -2.0])
    q_term_ref = np.array([0, 0, 0.275, 0, 0, 0]); qd_term_ref = np.zeros(6)
    Xref = np.zeros((12, N + 1))
    for i in range(6):
        Xref[i] = np.linspace(q_init[i], q_term_ref[i], N + 1)
        Xref[6 + i] = np.linspace(qd_init[i], qd_term_ref[i], N + 1)
    c_ref = SIDE_SIGN * np.tile([0.2, 0.1, -0.2], 4)                       # :189
    Uref = np.zeros((24, N))
    for leg in range(4):
        for xyz in range(3):
            Uref[3 * leg + xyz] = Xref[xyz, :-1] + c_ref[3 * leg + xyz]     # :205 (no rotation in this script)
    p = pack_params(N, Xref, np.full(N, T / N), [-10, -10, 0.1, -10, -10, -10], [10, 10, 1.0, 10, 10, 10],
                    [-10, -10, -10, -40, -40, -40], [10, 10, 10, 40, 40, 40], q_init, qd_init,
                    [-10, -10, 0.2, -0.1, -0.1, -10], [10, 10, 5, 0.1, 0.1, 10], [-10, -10, -10, -40, -40, -40],
                    [10, 10, 10, 40, 40, 40], [0, 0, 100, 100, 100, 0, 10, 10, 10, 10, 10, 10], 1.0, 0.35, 200.0, mass, Ib, Ib_inv)
    x0 = np.concatenate([Xref.flatten(order="F"), Uref.flatten(order="F")])
    return p, x0
