import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0)
o = L.default_opts(); o.max_iter = 300
r = L.solve_host(P, X0, o)
it = r['iters']; idx = np.argsort(-it)[:12]
print('percentiles 50/90/95/99/99.5/100:', np.percentile(it, [50, 90, 95, 99, 99.5, 100]))
for i in idx: print(i, it[i], 'rpy', np.round(q[i, 3:6], 3), 'z', round(q[i, 2], 3), 'v', np.round(qd[i, 3:6], 2), 'w', np.round(qd[i, :3], 2), 'f*', r['f'][i])
