"""dev: barrier-parameter schedule (kappa_mu, theta_mu, mu_init) on eight bench batches: iterations, tail, time per batch"""
import importlib, sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
L = capi.LandingLib(N, 0)
NBAT = int(os.environ.get('NBAT', '8'))
batches = [problem.make_batch(B, N, 0.6, seed=(20211 if NBAT == 8 else 777000) + 1000 * i)[:2] for i in range(NBAT)]
cfgs = [dict(), dict(kappa_mu=0.1), dict(theta_mu=1.8), dict(kappa_mu=0.1, theta_mu=1.8), dict(kappa_mu=0.05, theta_mu=2.0), dict(kappa_mu=0.3, theta_mu=1.3),
        dict(mu_init=0.03), dict(mu_init=0.3), dict(kappa_mu=0.1, mu_init=0.03), dict(tau_min=0.95), dict(tau_min=0.8), dict(tau_min=0.99)]
if len(sys.argv) > 1:
    cfgs = [eval("dict(%s)" % a) for a in sys.argv[1:]]
for cfg in cfgs:
    o = L.default_opts(); o.max_iter = 300
    for k, v in cfg.items(): setattr(o, k, v)
    L.solve_host(batches[0][0][:64], batches[0][1][:64], o)
    its, conv, ts, worst = [], 0, [], 0
    for P, X0 in batches:
        t = time.time(); r = L.solve_host(P, X0, o); ts.append(time.time() - t)
        c = r['status'] == 0; conv += int(c.sum()); its.append(r['iters']); worst = max(worst, int(r['iters'].max()))
    its = np.concatenate(its)
    print('%-40s conv %5d/%d  iters mean %.2f p99 %.0f max %d  ms/batch (host path) mean %.1f min %.1f' % (cfg, conv, NBAT * B, its.mean(), np.percentile(its, 99), worst, 1e3 * np.mean(ts), 1e3 * np.min(ts)), flush=True)
