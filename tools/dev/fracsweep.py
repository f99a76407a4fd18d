import importlib, sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
for N, B, seed in ((40, 1024, 1), (40, 1024, 777), (40, 2048, 20211), (20, 1024, 5)):
    P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=seed)
    L = capi.LandingLib(N, 0)
    for frac in (0.5, 0.3, 0.2, 0.1, 0.05, 0.02):
        o = L.default_opts(); o.max_iter = 300; o.bound_frac = frac
        L.solve_host(P[:8], X0[:8], o)
        t = time.time(); r = L.solve_host(P, X0, o); dt = time.time() - t
        c = r['status'] == 0
        print('N %d B %d seed %5d frac %.2f: conv %4d  iters mean %.1f med %.0f p90 %.0f max(conv) %d  host-sec %.3f' % (N, B, seed, frac, c.sum(), r['iters'].mean(), np.median(r['iters']), np.percentile(r['iters'], 90), r['iters'][c].max(), dt), flush=True)
    L.close()
