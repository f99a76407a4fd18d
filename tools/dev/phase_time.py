"""Development probe (round 6): time and phase timers of single hard members of the production problem on the GPU, new phase rules against round 5's.
   python tools/dev/phase_time.py"""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 20
L = capi.LandingLib(N, 0)
names = ["eval", "err", "sigrho", "back", "fwd", "dual", "ls", "accept", "nfact", "ntrial", "niter", "nstage_ok", "b_asm", "nstage", "b_elim", "b_post"]
R5 = dict(feas_max=1, feas_back=0.0, feas_delta_dec=0.0, feas_ret_push=0.0, feas_resume=0, feas_polish=0.0)
for seed, m in ((100006, 933), (100000, 870), (100010, 457), (100000, 145), (100000, 5)):
    P, X0, _, _ = problem.make_batch(1024, N, 0.6, seed=seed, consts=problem.production_constants("datagen"), dt_grid="reference", law="datagen")
    for label, ov in (("new", {}), ("r5 ", R5)):
        o = L.default_opts(); o.max_iter = 300
        for k, v in ov.items(): setattr(o, k, v)
        dP, dX0 = torch.tensor(P[m:m + 1], device="cuda"), torch.tensor(X0[m:m + 1], device="cuda")
        x = torch.empty(1, L.nx, device="cuda", dtype=torch.float64); st = torch.empty(1, device="cuda", dtype=torch.int32); it = torch.empty_like(st)
        prof = torch.zeros(1, 16, device="cuda", dtype=torch.float64)
        s = torch.cuda.current_stream().cuda_stream
        L.solve_device(1, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, s); torch.cuda.synchronize()
        t = time.perf_counter(); L.solve_device(1, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, s); torch.cuda.synchronize(); t = time.perf_counter() - t
        L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
        L.solve_device(1, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, s); torch.cuda.synchronize()
        L.lib.landing_set_profile_buffer(L.ctx, None)
        ph = prof.cpu().numpy()[0]; n = int(it.item())
        print("(%d,%d) %s status %d iters %d  %.1f ms = %.3f ms/it | per it (us): " % (seed, m, label, st.item(), n, 1e3 * t, 1e3 * t / max(n, 1)) +
              " ".join("%s %.0f" % (names[i], ph[i] * 1e-2 / max(n, 1)) for i in range(8)) + " | nfact/it %.2f ntrial/it %.2f" % (ph[8] / max(n, 1), ph[9] / max(n, 1)))
