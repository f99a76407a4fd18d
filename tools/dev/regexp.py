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










for sd in (20211, 5150, 1, 99):
    P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=sd)
    dP.copy_(torch.tensor(P)); dX0.copy_(torch.tensor(X0))
    print('seed', sd)
    def run2(label, **kw):
        run(label, **kw); its = it.cpu().numpy(); print('      max iters %d  p99 %d  top5 %s' % (its.max(), np.percentile(its, 99), np.sort(its)[-5:].tolist()))
    run2('default')
    run2('reset_du 1e5', reset_du=1e5)
    run2('reset_du 1e6', reset_du=1e6)
    run2('reset_du 1e7', reset_du=1e7)
    run2('max_resets 16 reset_du 1e6', reset_du=1e6, max_resets=16)
