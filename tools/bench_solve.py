#!/usr/bin/env python3
"""Quick solver timing on one GPU (development tool)."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=40)
ap.add_argument("--steps", type=int, default=2); ap.add_argument("--seed", type=int, default=20211); ap.add_argument("--max_iter", type=int, default=3000); ap.add_argument("--prof", action="store_true"); ap.add_argument("--lib", default=None)
a = ap.parse_args()
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
L = capi.LandingLib(a.N, 0, lib_path=a.lib)
P, X0, _, _ = problem.make_batch(a.B, a.N, 0.6, seed=a.seed)
dev = "cuda"
dP, dX0 = torch.tensor(P, device=dev), torch.tensor(X0, device=dev)
mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
x, f, lam, kkt = mk(a.B, L.nx), mk(a.B), mk(a.B, L.ng), mk(a.B, 3)
st, it = mk(a.B, dt=torch.int32), mk(a.B, dt=torch.int32)
o = L.default_opts(); o.max_iter = a.max_iter
s = torch.cuda.current_stream().cuda_stream
prof = mk(a.B, 16) if a.prof else None
if a.prof:
    prof.zero_(); L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
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
    if a.prof:
        ph = prof.cpu().numpy()
        names = ["eval", "err", "sigrho", "back", "fwd", "dual", "ls", "accept"]
        tot = ph[:, :8].sum(axis=0) / 100e6  # seconds summed over members
        nit, nfact, ntrial = ph[:, 10].sum(), ph[:, 8].sum(), ph[:, 9].sum()
        print(json.dumps({"phase_ms_per_iter": {n: 1e3 * t / nit for n, t in zip(names, tot)}, "ms_per_iter_total": 1e3 * tot.sum() / nit,
                          "fact_per_iter": nfact / nit, "trials_per_iter": ntrial / nit, "ms_per_fact": 1e3 * tot[3] / nfact, "ms_per_trial": 1e3 * tot[6] / ntrial,
                          "back_us_per_stage_fact": {n: 1e6 * ph[:, 11 + i].sum() / 100e6 / nfact / a.N for i, n in enumerate(["load", "asm", "tpt", "elim+post", "pivot_chain"])}}))
