"""Writes tests/golden/n1_kd_solved.npz: three members of the kinodynamic refinement batch SOLVED ON AN MI355X by this repository's solver
(landing_kinodyn_solve_batch) -- drop state, the SRBM solution used as the initial guess, the returned x*, lam_g* and the kernel's KKT report.
Data only (our own outputs, not reference material); the CPU suite re-solves them through the host emulation from a perturbed x* and certifies
both with the complex-step oracle.  Run on the GPU box:  python tests/make_golden_kd.py gpurun_out/n1_kd_solved.npz"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P_ = importlib.import_module("landing-controller_amd.problem"); capi = importlib.import_module("landing-controller_amd.capi")
rbd = importlib.import_module("landing-controller_amd.rbd"); kd = importlib.import_module("landing-controller_amd.kinodyn")
K = importlib.import_module("landing-controller_amd.constants")
N, pick = 20, [0, 1, 3]
consts = P_.production_constants("main")
P, X0, q, qd = P_.make_batch(8, N, 0.6, seed=7, consts=consts, dt_grid="reference", law="main")
P, X0, q, qd = P[pick], X0[pick], q[pick], qd[pick]
L = capi.LandingLib(N, device=0)
R = rbd.Rbd(L)
srbm = L.solve_host(P, X0)
assert (srbm["status"] == 0).all()
mass, Ib, Ibi = K.robot_constants()
prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b]) for b in range(len(pick))]
lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
s = R.kinodyn_solve_host(N, lb, ub, cost, x0, P_.REFERENCE_DT_GRID, mass, Ib, Ibi, consts.mu)
assert (s["status"] == 0).all(), s["status"]
np.savez_compressed(sys.argv[1], q_init=q, qd_init=qd, x_srbm=srbm["x"], x=s["x"], lam_g=s["lam_g"], kkt=s["kkt"], iters=s["iters"], f=s["f"])
print("wrote", sys.argv[1], s["iters"], s["kkt"].max(axis=0))
