import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40
P, X0, _, _ = problem.make_batch(1, N, 0.6, seed=20211)
for lp in sys.argv[1:]:
    L = capi.LandingLib(N, 0, lib_path=lp)
    prof = torch.zeros(1, 16, device='cuda', dtype=torch.float64)
    L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
    o = L.default_opts(); o.max_iter = 1
    L.solve_host(P, X0, o)
    ph = prof.cpu().numpy()[0]
    print("nfact", ph[8], "ntrial", ph[9], "niter", ph[10]); print("alpha at entry", ph[6], "after eval_g", ph[7]); print(lp, dict(a_pr=ph[0], a_du=ph[1], bar=ph[2], f0=ph[3], bt=ph[4], ft=ph[5], th0=ph[11], ph0=ph[12], dphi=ph[13], tht=ph[14], pht=ph[15]))
