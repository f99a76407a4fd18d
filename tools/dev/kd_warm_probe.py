"""Development probe (round 6): iterations of the refinement's re-solve from x* under variants of the warm-start preset.  python tools/dev/kd_warm_probe.py"""
import importlib, sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lc = lambda m: importlib.import_module("landing-controller_amd." + m)
P, kd, capi, rbd, K = lc("problem"), lc("kinodyn"), lc("capi"), lc("rbd"), lc("constants")
N, B = 20, 256
L = capi.LandingLib(N, device=0); R = rbd.Rbd(L)
consts = P.production_constants("main")
Pp, X0, q, qd = P.make_batch(B, N, 0.6, seed=12, consts=consts, dt_grid="reference", law="main")
srbm = L.solve_host(Pp, X0)
mass, Ib, Ibi = K.robot_constants(); dt = P.REFERENCE_DT_GRID
prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b]) for b in range(B)]
lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
cold = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, consts.mu, R.kinodyn_default_opts())
ok = cold["status"] == 0
for push, mu, extra in ((1e-4, 1e-4, {}), (1e-5, 1e-5, {}), (1e-6, 1e-6, {}), (1e-3, 1e-3, {}), (1e-4, 1e-6, {}), (1e-6, 1e-4, {}), (1e-4, 1e-4, dict(delta_floor=0.0)), (1e-5, 1e-5, dict(kappa_eps=10.0)), (1e-5, 1e-5, dict(watchdog=0, slack_corr=0.0))):
    w = R.kinodyn_warm_opts(); w.bound_push = push; w.bound_frac = push; w.mu_init = mu
    for k, v in extra.items(): setattr(w, k, v)
    r = R.kinodyn_solve_host(N, lb[ok], ub[ok], cost[ok], cold["x"][ok], dt, mass, Ib, Ibi, consts.mu, w)
    it = r["iters"]
    print("push %g mu %g %s: converged %d / %d, iterations mean %.1f median %.0f p90 %.0f p99 %.0f max %d" % (push, mu, extra, (r["status"] == 0).sum(), ok.sum(), it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max()), flush=True)
