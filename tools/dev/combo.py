"""Development probe: batch time (B=1024, N=40) for dispatch_order x restart_period over several seeds."""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
L = capi.LandingLib(N, 0)
dev = "cuda"
mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
x, st, it = mk(B, L.nx), mk(B, dt=torch.int32), mk(B, dt=torch.int32)
stream = torch.cuda.current_stream().cuda_stream
tot = {}
for seed in (20211, 20212, 5150, 1, 7):
    P, X0, _, _ = problem.make_batch(B, N, 0.6, seed=seed)
    dP, dX0 = torch.tensor(P, device=dev), torch.tensor(X0, device=dev)
    for order in (0, 1):
        for rp in (60, 80):
            o = L.default_opts(); o.max_iter = 300; o.dispatch_order = order; o.restart_period = rp
            ts = []
            for rep in range(3):
                torch.cuda.synchronize(); t = time.perf_counter()
                L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, stream)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
            ms = 1e3 * min(ts[1:]); ith = it.cpu().numpy(); c = int((st == 0).sum())
            tot[(order, rp)] = tot.get((order, rp), 0) + ms
            print("seed %5d order %d restart %d: %.1f ms  conv %d  iters mean %.1f max %d" % (seed, order, rp, ms, c, ith.mean(), ith.max()), flush=True)
for k, v in sorted(tot.items()): print("order %d restart %d: total %.1f ms over 5 seeds -> %.0f NLPs/s" % (k[0], k[1], v, 5 * B / v * 1e3))
