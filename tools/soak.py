"""Robustness sweep of the solver over many synthetic batches (SURVEY 8d sampling law): convergence rate, iteration statistics,
batch times and the worst members (seed, index, iterations, status) -- the cases to look at next.
   python tools/soak.py [--batches 64] [--B 1024] [--N 40]"""
import argparse, importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
ap = argparse.ArgumentParser(); ap.add_argument("--batches", type=int, default=64); ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=40)
ap.add_argument("--grid", default="uniform", help="uniform | reference (the production callers' N=20 grid, problem.REFERENCE_DT_GRID)"); ap.add_argument("--law", default="main", help="main | datagen (problem.DROP_LAWS)")
ap.add_argument("--form", default="terminal", help="terminal | running (the running cost of generate_quadruped_SRBM_CCC.m:81-89 with the weights of tests/test_gpu_solver.py::RUN_COST) | ccc (kin-box .05/.05/.27, GRF cost only: the formulation of the stored N=40 solutions)")
ap.add_argument("--seed0", type=int, default=100000); ap.add_argument("--seed-step", type=int, default=1); ap.add_argument("--opts", default=""); a = ap.parse_args()
FORMS = {"terminal": {}, "running": dict(run_cost=dict(QX=[0, 0, 10, 1, 1, 0, .1, .1, .1, .1, .1, .1], Qc=[1.0, 1.0, 0.5], Qf=[1e-4, 1e-4, 1e-3], f_ref=[0, 0, 20.0])),
         "ccc": dict(kin_box=(0.05, 0.05, 0.27), run_cost=dict(QX=[0] * 12, Qc=[0, 0, 0], Qf=[1e-4, 1e-4, 1e-3], f_ref=[0, 0, 0]))}
L = capi.LandingLib(a.N, 0, **FORMS[a.form])
o = L.default_opts(); o.max_iter = 300
for kv in a.opts.split(","):
    if "=" in kv:
        k_, v_ = kv.split("="); setattr(o, k_, type(getattr(o, k_))(float(v_)))
consts = problem.production_constants(a.law) if a.grid == "reference" else None
vz_fail = []
mk = lambda *s, dt=torch.float64: torch.empty(*s, device="cuda", dtype=dt)
x, st, it, kkt = mk(a.B, L.nx), mk(a.B, dt=torch.int32), mk(a.B, dt=torch.int32), mk(a.B, 3)
stream = torch.cuda.current_stream().cuda_stream
its, sts, ms, worst = [], [], [], []
for b in range(a.batches):
    seed = a.seed0 + b * a.seed_step
    P, X0, _, qd_ = problem.make_batch(a.B, a.N, 0.6, seed=seed, consts=consts, dt_grid=a.grid, law=a.law)
    dP, dX0 = torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter()
    L.solve_device(a.B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), kkt.data_ptr(), stream)
    torch.cuda.synchronize(); ms.append(1e3 * (time.perf_counter() - t))
    ih, sh, kh = it.cpu().numpy(), st.cpu().numpy(), kkt.cpu().numpy()
    its.append(ih); sts.append(sh); vz_fail += [float(v) for v in qd_[sh != 0, 5]]
    for m in np.argsort(-ih)[:3]:
        worst.append((int(ih[m]), seed, int(m), int(sh[m])))
    for m in np.nonzero(sh != 0)[0]:
        worst.append((int(ih[m]), seed, int(m), int(sh[m])))
its, sts, ms = np.concatenate(its), np.concatenate(sts), np.array(ms[1:] if len(ms) > 1 else ms)
worst = sorted(set(worst), reverse=True)[:16]
print(json.dumps({"workload": "%d batches x %d drop states, N=%d, dt grid %s, sampling law %s, objective form %s, max_iter 300, KKT tol 1e-6" % (a.batches, a.B, a.N, a.grid, a.law, a.form), "options": a.opts,
                  "vz_of_unconverged_min_max": [min(vz_fail), max(vz_fail)] if vz_fail else None,
                  "members": int(its.size), "converged": int((sts == 0).sum()), "max_iter_hit": int((sts == 1).sum()), "numerical": int((sts == 2).sum()), "certified_locally_infeasible": int((sts == 3).sum()), "stalled": int((sts == 4).sum()),
                  "iters_mean": float(its.mean()), "iters_p50": float(np.median(its)), "iters_p99": float(np.percentile(its, 99)), "iters_p999": float(np.percentile(its, 99.9)), "iters_max": int(its.max()),
                  "batch_ms_mean": float(ms.mean()), "batch_ms_min": float(ms.min()), "batch_ms_max": float(ms.max()), "nlps_per_s_mean": float(a.B / ms.mean() * 1e3),
                  "worst_members_iters_seed_index_status": worst}))
