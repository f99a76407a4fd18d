"""dev: objective reached under option overrides against the reference's option values (mu_init 0.1, bound_push 0.5), member by member, terminal-cost
form, eight bench batches: same (rel 1e-3) / better / worse local minimum.   python3 tools/dev/fstar_cmp.py "mu_init=0.5" "mu_init=0.5,bound_push=1.0" """
import importlib, sys, os, numpy as np, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
L = capi.LandingLib(N, 0)
batches = [problem.make_batch(B, N, 0.6, seed=20211 + 1000 * i)[:2] for i in range(8)]
def run(cfg):
    o = L.default_opts(); o.max_iter = 300
    for k, v in cfg.items(): setattr(o, k, v)
    out = [L.solve_host(P, X0, o) for P, X0 in batches]
    return np.concatenate([r["f"] for r in out]), np.concatenate([r["status"] for r in out]), np.concatenate([r["iters"] for r in out])
f0, s0, i0 = run(dict(mu_init=0.1, bound_push=0.5))
print("reference values: converged %d, iterations mean %.2f, f* median %.4g" % ((s0 == 0).sum(), i0.mean(), np.median(f0)))
for a in sys.argv[1:]:
    f, s, it = run(eval("dict(%s)" % a))
    ok = (s == 0) & (s0 == 0); rel = (f[ok] - f0[ok]) / np.maximum(np.abs(f0[ok]), 1e-12)
    print("%-40s converged %d  iters mean %.2f  same %d  better %d  worse %d  (worse by: median %.2g max %.2g; better by: median %.2g max %.2g)" % (
        a, (s == 0).sum(), it.mean(), (np.abs(rel) <= 1e-3).sum(), (rel < -1e-3).sum(), (rel > 1e-3).sum(),
        np.median(rel[rel > 1e-3]) if (rel > 1e-3).any() else 0, rel.max(), np.median(-rel[rel < -1e-3]) if (rel < -1e-3).any() else 0, -rel.min()), flush=True)
