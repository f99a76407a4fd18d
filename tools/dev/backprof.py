"""Development probe: per-stage cost of the backward sweep (scatter / elimination timers of the kernel), alone and under load.
   python tools/dev/backprof.py [lib.so]"""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40
path = sys.argv[1] if len(sys.argv) > 1 else None
L = capi.LandingLib(N, 0, lib_path=os.path.join(ROOT, path) if path else None)
o = L.default_opts(); o.max_iter = 300
for kv in os.environ.get("VAR_OPTS", "").split(","):
    if "=" in kv:
        k_, v_ = kv.split("="); setattr(o, k_, type(getattr(o, k_))(float(v_)))
for label, nb in (("alone", 8), ("load", 1024)):
    P, X0, _, _ = problem.make_batch(nb, N, 0.6, seed=20211)
    dP, dX0 = torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")
    x = torch.empty(nb, L.nx, device="cuda", dtype=torch.float64); st = torch.empty(nb, device="cuda", dtype=torch.int32); it = torch.empty_like(st)
    prof = torch.zeros(nb, 16, device="cuda", dtype=torch.float64)
    L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
    L.solve_device(nb, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    L.lib.landing_set_profile_buffer(L.ctx, None)
    ph = prof.cpu().numpy().sum(axis=0)
    nst = ph[13]; tick = 1e-2   # us per tick (100 MHz)
    print("%-5s stage eliminations %.0f: back %.2f us/stage = scatter %.2f + elim %.2f (+ sweep overhead %.2f); prologue slot %.2f" %
          (label, nst, ph[3] * tick / nst, ph[12] * tick / nst, ph[14] * tick / nst, (ph[3] - ph[12] - ph[14]) * tick / nst, ph[15] * tick / nst))
