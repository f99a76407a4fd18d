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










for sd in (20211, 5150):
    P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=sd)
    dP.copy_(torch.tensor(P)); dX0.copy_(torch.tensor(X0))
    print('seed', sd)
    run('default')
    run('mu_init 0.03', mu_init=0.03)
    run('mu_init 0.3', mu_init=0.3)
    run('mu_init 1', mu_init=1.0)
    run('kappa_mu 0.1', kappa_mu=0.1)
    run('kappa_mu 0.3', kappa_mu=0.3)
    run('theta_mu 1.8', theta_mu=1.8)
    run('tau_min 0.95', tau_min=0.95)
    run('tau_min 0.99', tau_min=0.99)
    run('tau_min 0.8', tau_min=0.8)
    run('bound_push 0.2', bound_push=0.2)
    run('bound_push 1.0', bound_push=1.0)
    run('kappa_eps 5', kappa_eps=5.0)
    run('kappa_eps 20', kappa_eps=20.0)
