import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40
P, X0, _, _ = problem.make_batch(2, N, 0.6, seed=20211)
A = capi.LandingLib(N, 0); Bl = capi.LandingLib(N, 0, lib_path=sys.argv[1])
for K in (0, 1, 2, 3, 5, 10):
    oa = A.default_opts(); oa.max_iter = K; ob = Bl.default_opts(); ob.max_iter = K
    ra = A.solve_host(P, X0, oa); rb = Bl.solve_host(P, X0, ob)
    print(K, 'dx', np.abs(ra['x'] - rb['x']).max(), 'dlam', np.abs(ra['lam_g'] - rb['lam_g']).max(), 'kkt a', ra['kkt'][0], 'kkt b', rb['kkt'][0], 'it', ra['iters'], rb['iters'], flush=True)
oa = A.default_opts(); oa.max_iter = 1; ra = A.solve_host(P, X0, oa); rb = Bl.solve_host(P, X0, oa)
d = np.abs(ra['lam_g'][0] - rb['lam_g'][0]); idx = np.nonzero(d > 1e-6 * (1 + np.abs(ra['lam_g'][0])))[0]
print('rows differing', len(idx), idx[:40].tolist())
print('stage/local of first ones', [((i - 36) // 104, (i - 36) % 104) for i in idx[:20]])
dx = np.abs(ra['x'][0] - rb['x'][0]); ix = np.nonzero(dx > 1e-9)[0]; print('x differing', len(ix), ix[:30].tolist())
