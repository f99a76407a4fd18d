"""Development probe: do alternative settings solve the slowest members of a batch faster?"""
import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
L = capi.LandingLib(N, 0)
for seed in (20211, 5150, 1):
    P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=seed)
    o = L.default_opts(); o.max_iter = 300
    r = L.solve_host(P, X0, o)
    slow = np.argsort(-r['iters'])[:10]
    print('seed', seed, 'slowest', r['iters'][slow].tolist(), flush=True)
    def trial(label, **kw):
        o = L.default_opts(); o.max_iter = 300
        for k, v in kw.items(): setattr(o, k, v)
        rr = L.solve_host(P[slow], X0[slow], o)
        print('   %-26s iters %s  status %s' % (label, rr['iters'].tolist(), rr['status'].tolist()), flush=True)
    trial('mu_init 1', mu_init=1.0)
    trial('mu_init 0.01', mu_init=0.01)
    trial('bound_push 0.2', bound_push=0.2)
    trial('bound_frac 0.02', bound_frac=0.02)
    trial('tau_min 0.99', tau_min=0.99)
    trial('restart 40', restart_period=40)
    trial('kappa_eps 30', kappa_eps=30.0)
