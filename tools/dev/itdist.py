import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0)
res = {}
for mi in (300, 1000):
    o = L.default_opts(); o.max_iter = mi
    r = L.solve_host(P, X0, o); res[mi] = r
    c = r['status'] == 0
    it = r['iters'][c]
    print(mi, 'converged', c.sum(), 'iters of converged: p50 %d p90 %d p95 %d p98 %d p99 %d max %d' % tuple(np.percentile(it, [50, 90, 95, 98, 99, 100])))
    print('   histogram of converged iters >150:', np.sort(it[it > 150]).tolist())
r3, r10 = res[300], res[1000]
bad = np.nonzero(r3['status'] != 0)[0]
print('not converged at 300:', len(bad))
for b in bad:
    print('  m%4d st300 %d kkt300 %s | st1000 %d it %d kkt %s' % (b, r3['status'][b], np.array2string(r3['kkt'][b], precision=2), r10['status'][b], r10['iters'][b], np.array2string(r10['kkt'][b], precision=2)))
