import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0)
prof = torch.zeros(B, 16, device='cuda', dtype=torch.float64)
L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
o = L.default_opts(); o.max_iter = 300
r = L.solve_host(P, X0, o)
ph = prof.cpu().numpy()
print('iters', ph[:, 10].sum(), 'facts', ph[:, 8].sum(), 'failed: foot-block', ph[:, 12].sum(), ' k<=3', ph[:, 13].sum(), ' middle', ph[:, 11].sum(), ' k>=N-4', ph[:, 15].sum())
