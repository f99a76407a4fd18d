#!/usr/bin/env python3
"""Quick solver timing on one GPU (development tool)."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=40)
ap.add_argument("--steps", type=int, default=2); ap.add_argument("--seed", type=int, default=20211); ap.add_argument("--max_iter", type=int, default=3000)
a = ap.parse_args()
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
L = capi.LandingLib(a.N, 0)
P, X0, _, _ = problem.make_batch(a.B, a.N, 0.6, seed=a.seed)
dev = "cuda"
dP, dX0 = torch.tensor(P, device=dev), torch.tensor(X0, device=dev)
mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
x, f, lam, kkt = mk(a.B, L.nx), mk(a.B), mk(a.B, L.ng), mk(a.B, 3)
st, it = mk(a.B, dt=torch.int32), mk(a.B, dt=torch.int32)
o = L.default_opts(); o.max_iter = a.max_iter
s = torch.cuda.current_stream().cuda_stream
for step in range(a.steps):
    torch.cuda.synchronize(); t = time.time()
    L.solve_device(a.B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), f.data_ptr(), lam.data_ptr(), st.data_ptr(), it.data_ptr(), kkt.data_ptr(), s)
    torch.cuda.synchronize(); dt_ = time.time() - t
    sth, ith, kh = st.cpu().numpy(), it.cpu().numpy(), kkt.cpu().numpy()
    conv = sth == 0
    print(json.dumps({"B": a.B, "N": a.N, "sec": dt_, "converged": int(conv.sum()), "nlp_per_s": float(conv.sum() / dt_),
                      "iters_mean": float(ith.mean()), "iters_med": float(np.median(ith)), "iters_max": int(ith.max()),
                      "iters_p90": float(np.percentile(ith, 90)), "status_counts": np.bincount(sth, minlength=3).tolist(),
                      "kkt_max_conv": kh[conv].max(axis=0).tolist() if conv.any() else None}))
