"""dev: the outliers of a kinodynamic bench batch under other option sets -- would a clone with another slack initialisation / first barrier parameter /
step rule have converged sooner?   python tools/dev/kd_variants.py SEED m1,m2,...  (law main, B = 1024 as tools/bench_kd_solve.py poses it)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
P_ = importlib.import_module("landing-controller_amd.problem"); capi = importlib.import_module("landing-controller_amd.capi")
rbd = importlib.import_module("landing-controller_amd.rbd"); kd = importlib.import_module("landing-controller_amd.kinodyn"); K = importlib.import_module("landing-controller_amd.constants")
N, B = 20, 1024
seed = int(sys.argv[1]); idx = [int(v) for v in sys.argv[2].split(",")]; law = sys.argv[3] if len(sys.argv) > 3 else "main"
consts = P_.production_constants(law)
P, X0, q, qd = P_.make_batch(B, N, 0.6, seed=seed, consts=consts, dt_grid="reference", law=law)
L = capi.LandingLib(N, device=0, lib_path=os.environ.get("LANDING_LIB")); R = rbd.Rbd(L)
srbm = L.solve_host(P, X0)
mass, Ib, Ibi = K.robot_constants()
prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b], None) for b in idx]
lbs, ubs, costs, x0s = (np.array([p[i] for p in prob]) for i in range(4))
variants = [("defaults", {}), ("push0.1", dict(bound_push=0.1, bound_frac=0.1)), ("mu0.02", dict(mu_init=0.02)), ("clip16", dict(clip_k=16)), ("clip2", dict(clip_k=2)),
            ("push0.1+clip16", dict(bound_push=0.1, bound_frac=0.1, clip_k=16)), ("mu1", dict(mu_init=1.0)), ("keps10", dict(kappa_eps=10.0))]
print("members", idx)
for name, kv in variants:
    o = R.kinodyn_default_opts(); o.max_iter = 500
    for k_, v_ in kv.items(): setattr(o, k_, type(getattr(o, k_))(v_))
    s = R.kinodyn_solve_host(N, lbs, ubs, costs, x0s, P_.REFERENCE_DT_GRID, mass, Ib, Ibi, consts.mu, o)
    print("%-16s status %s iters %s" % (name, s["status"].tolist(), s["iters"].tolist()), flush=True)
