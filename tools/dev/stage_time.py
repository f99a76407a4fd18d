"""dev probe: time of one stage of the backward sweep (PH_B_ELIM / PH_NSTAGE of the kernel's phase timers), alone (8 members) and under load (1024),
for library variants whose results may be garbage (-DLANDING_DEV_ASM_LEVEL builds: the assembler wave does part of its work only).
    python tools/dev/stage_time.py lib1.so lib2.so ..."""
import importlib, sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40
P, X0, _, _ = problem.make_batch(1024, N, 0.6, seed=20211)
for path in sys.argv[1:]:
    L = capi.LandingLib(N, 0, lib_path=os.path.join(ROOT, path))
    dP, dX0 = torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")
    x = torch.empty(1024, L.nx, device="cuda", dtype=torch.float64); st = torch.empty(1024, device="cuda", dtype=torch.int32); it = torch.empty_like(st)
    o = L.default_opts(); o.max_iter = 20; o.feas_phase = 0
    out = {"lib": path}
    for label, nb in (("load", 1024), ("alone", 8)):
        prof = torch.zeros(nb, 16, device="cuda", dtype=torch.float64)
        L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
        L.solve_device(nb, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        L.lib.landing_set_profile_buffer(L.ctx, None)
        ph = prof.cpu().numpy()
        out[label + "_us_per_stage"] = round(float(ph[:, 14].sum() / 100.0 / max(ph[:, 13].sum(), 1.0)), 3)
        out[label + "_stages"] = int(ph[:, 13].sum()); out[label + "_ok"] = int(ph[:, 11].sum())
    print(json.dumps(out), flush=True)
    L.close()
