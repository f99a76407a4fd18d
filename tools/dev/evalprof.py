"""Development probe: time of the derivative phase (first iterations only) of library builds, alone and under load.
   python tools/dev/evalprof.py [name=lib.so ...]"""
import importlib, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40
for spec in sys.argv[1:] or ["cur=cur"]:
    name, path = spec.split("=")
    L = capi.LandingLib(N, 0, lib_path=None if path == "cur" else os.path.join(ROOT, path))
    o = L.default_opts(); o.max_iter = 3
    out = []
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
        out.append("%s eval %.4f ms per evaluation" % (label, ph[0] * 1e-5 / (4 * nb)))     # 4 evaluations in 3 iterations (+ the final one)
    print("%-8s %s" % (name, " | ".join(out)), flush=True)
    L.close()
