"""LO_TRACE of one member of the production problem through the CPU port:  python tools/dev/feas_trace.py <global index> [k=v,...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as om
problem = importlib.import_module("landing-controller_amd.problem")
gi = int(sys.argv[1]); law = os.environ.get("LAW", "datagen"); N = 20
opts = {}
for kv in (sys.argv[2] if len(sys.argv) > 2 else "").split(","):
    if "=" in kv:
        k, v = kv.split("="); opts[k] = float(v) if ("." in v or "e" in v) else int(v)
O = om.Oracle(N)
P, X0, _, _ = problem.make_batch(1024, N, 0.6, seed=100000 + gi // 1024, consts=problem.production_constants(law), dt_grid="reference", law=law)
m = gi % 1024
os.environ["LO_TRACE"] = "1"
r = om.cpu_solve_batch(O, P[m:m + 1], X0[m:m + 1], threads=1, max_iter=int(os.environ.get("MAXIT", 300)), **opts)
print("status", r["status"], "iters", r["iters"], "kkt", r["kkt"], file=sys.stderr)
