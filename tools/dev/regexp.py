import importlib, sys, os, numpy as np, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0)
prof = torch.zeros(B, 16, device='cuda', dtype=torch.float64)
dP, dX0 = torch.tensor(P, device='cuda'), torch.tensor(X0, device='cuda')
mk = lambda *s, dt=torch.float64: torch.empty(*s, device='cuda', dtype=dt)
x, f, lam, kkt = mk(B, L.nx), mk(B), mk(B, L.ng), mk(B, 3); st, it = mk(B, dt=torch.int32), mk(B, dt=torch.int32)
def run(label, **kw):
    o = L.default_opts(); o.max_iter = 300
    for k, v in kw.items(): setattr(o, k, v)
    L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr()); prof.zero_()
    L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), f.data_ptr(), lam.data_ptr(), st.data_ptr(), it.data_ptr(), kkt.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); ph = prof.cpu().numpy(); L.lib.landing_set_profile_buffer(L.ctx, None)
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t = time.time()
        L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), f.data_ptr(), lam.data_ptr(), st.data_ptr(), it.data_ptr(), kkt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); ts.append(time.time() - t)
    dt = min(ts); c = (st.cpu().numpy() == 0); its = it.cpu().numpy()
    print('%-34s conv %4d  iters mean %.1f med %.0f p90 %.0f  fact/iter %.2f  sec %.3f  nlp/s %.0f' % (label, c.sum(), its.mean(), np.median(its), np.percentile(its, 90), ph[:, 8].sum() / ph[:, 10].sum(), dt, c.sum() / dt), flush=True)










run('default')
run('pivot fix floor 1e-4', stage_local_reg=1)
run('pivot fix floor 1e-2', stage_local_reg=1, delta_init=1e-2)
run('pivot fix floor 1', stage_local_reg=1, delta_init=1.0)
run('pivot fix floor 1e-6', stage_local_reg=1, delta_init=1e-6)
run('default (push .5)')
run('push/frac .2', bound_push=0.2, bound_frac=0.2)
run('push/frac .1', bound_push=0.1, bound_frac=0.1)
run('push/frac .05', bound_push=0.05, bound_frac=0.05)
run('push .1 frac .5', bound_push=0.1, bound_frac=0.5)
run('push .5 frac .1', bound_push=0.5, bound_frac=0.1)
