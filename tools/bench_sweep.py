#!/usr/bin/env python3
"""Times the function-layer sweep kernel (one full derivative sweep per member) with HIP events."""
import argparse, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch

ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=40)
ap.add_argument("--steps", type=int, default=50); ap.add_argument("--noise", type=float, default=0.01); ap.add_argument("--presolve", type=int, default=0); ap.add_argument("--lib", default=None); ap.add_argument("--warmup", type=int, default=5)
a = ap.parse_args()
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
L = capi.LandingLib(a.N, 0, lib_path=a.lib)
nb = min(a.B, 256)
P, X0, _, _ = problem.make_batch(nb, a.N, 0.6, seed=1)
reps = (a.B + nb - 1) // nb
P = np.tile(P, (reps, 1))[:a.B]; X0 = np.tile(X0, (reps, 1))[:a.B]
rng = np.random.default_rng(0)
dev = "cuda"
dX = torch.tensor(X0 + a.noise * rng.normal(size=X0.shape), device=dev); dP = torch.tensor(P, device=dev)
dlam = torch.tensor(rng.normal(size=(a.B, L.ng)), device=dev)
mk = lambda *s: torch.empty(*s, device=dev, dtype=torch.float64)
g, gf, jac, hess = mk(a.B, L.ng), mk(a.B, L.nx), mk(a.B, L.nnz_jac), mk(a.B, L.nnz_hess)
st = torch.cuda.current_stream().cuda_stream
if a.presolve:   # the solver first (what bench.py does before its sweep leg): allocates the 1 GB workspace, leaves the GPU in its loaded state
    Ps, Xs, _, _ = problem.make_batch(1024, a.N, 0.6, seed=20211)
    for _ in range(a.presolve): L.solve_host(Ps, Xs)
def run():
    L.eval_device(a.B, dX.data_ptr(), dP.data_ptr(), 0, dlam.data_ptr(), 0, g.data_ptr(), gf.data_ptr(), jac.data_ptr(), hess.data_ptr(), 0, 0, st)
for _ in range(a.warmup): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.steps
by = L.lib.landing_sweep_bytes_per_member(a.N) * a.B
print(json.dumps({"kernel": "landing_sweep_kernel", "B": a.B, "N": a.N, "ms_per_launch": ms, "algorithmic_bytes": by, "GBps": by / ms / 1e6, "frac_of_8TBps": by / ms / 1e6 / 8000}))
