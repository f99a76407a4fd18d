"""Development probe: per-member iteration counts, phase timers and counters of the bench batch -> gpurun_out/itdump_<tag>.npz"""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
tag = sys.argv[1] if len(sys.argv) > 1 else "x"
N, B = 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0)
o = L.default_opts(); o.max_iter = 300
prof = torch.zeros(B, 16, device='cuda', dtype=torch.float64)
L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
r = L.solve_host(P, X0, o)
L.lib.landing_set_profile_buffer(L.ctx, None)
ph = prof.cpu().numpy()
t = time.time(); r2 = L.solve_host(P, X0, o); dt = time.time() - t
it = r['iters']; c = r['status'] == 0
print('conv', c.sum(), 'iters mean %.1f med %.0f p90 %.0f p99 %.0f max %d' % (it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max()))
print('fact/iter %.3f trials/iter %.3f  host-path sec %.3f' % (ph[:, 8].sum() / ph[:, 10].sum(), ph[:, 9].sum() / ph[:, 10].sum(), dt))
tot = ph[:, :8].sum(axis=0) / 1e5 / ph[:, 10].sum()     # ms per iteration (100 MHz ticks)
print('ms/iter under load by phase (eval err sigrho back fwd dual ls accept):', np.round(tot, 4), 'sum %.4f' % tot.sum())
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'itdump_%s.npz' % tag), iters=it, status=r['status'], prof=ph, kkt=r['kkt'])
