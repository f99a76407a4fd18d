"""Development probe: which alternative settings solve the members the default settings do not."""
import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 2048
P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0)
o = L.default_opts(); o.max_iter = 300
r = L.solve_host(P, X0, o)
bad = np.nonzero(r['status'] != 0)[0]
print('default: failed', len(bad), 'of', B, flush=True)
def trial(label, x0=None, **kw):
    o = L.default_opts(); o.max_iter = 300
    for k, v in kw.items(): setattr(o, k, v)
    rr = L.solve_host(P[bad], X0[bad] if x0 is None else x0, o)
    ok = rr['status'] == 0
    print('%-40s solves %3d of %d  (iters med %d)' % (label, ok.sum(), len(bad), np.median(rr['iters'][ok]) if ok.any() else -1), flush=True)
    return ok
a = trial('same settings again')
b = trial('mu_init 1', mu_init=1.0)
c = trial('mu_init 0.01', mu_init=0.01)
d = trial('bound_push/frac 0.1', bound_push=0.1, bound_frac=0.1)
e = trial('bound_push/frac 0.01', bound_push=0.01, bound_frac=0.01)
f = trial('tau_min 0.99', tau_min=0.99)
g = trial('delta_inc 8', delta_inc=8.0)
h = trial('kappa_mu 0.5', kappa_mu=0.5)
i = trial('restart from failed x, mu 1', x0=r['x'][bad], mu_init=1.0)
j = trial('restart from failed x', x0=r['x'][bad])
print('union of all:', (a | b | c | d | e | f | g | h | i | j).sum(), 'of', len(bad))
print('union mu1|push.1|fromx:', (b | d | j).sum())
