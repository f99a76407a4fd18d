"""dev probe (-DLANDING_STAGE_PROF build): where a stage of the backward sweep spends its time (wave 0, 100 MHz ticks between marks inside block_eliminate),
alone (8 members) and under load (1024)."""
import importlib, sys, os, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40
P, X0, _, _ = problem.make_batch(1024, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0, lib_path=os.path.join(ROOT, sys.argv[1]))
dP, dX0 = torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")
x = torch.empty(1024, L.nx, device="cuda", dtype=torch.float64); st = torch.empty(1024, device="cuda", dtype=torch.int32); it = torch.empty_like(st)
o = L.default_opts(); o.max_iter = 300
names = ["prologue", "step0", "asm_issue", "steps1-2", "asm_copy", "step3", "asm_terms", "step4", "asm_combine", "step5", "epilogue"]
for label, nb in (("load", 1024), ("alone", 8)):
    prof = torch.zeros(nb, 32, device="cuda", dtype=torch.float64)
    L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
    L.solve_device(nb, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); L.lib.landing_set_profile_buffer(L.ctx, None)
    ph = prof.cpu().numpy(); ns = ph[:, 13].sum()
    print(label, "us per stage: total(B_ELIM) %.2f |" % (ph[:, 14].sum() / 100 / ns), " ".join("%s %.2f" % (n, ph[:, 16 + i].sum() / 100 / ns) for i, n in enumerate(names)), "| sum %.2f" % (ph[:, 16:27].sum() / 100 / ns))
