#!/usr/bin/env python3
"""Kinodynamic refinement at batch size: SRBM solve (N = 20, production grid) -> refinement of the same drop states through
the device-pointer entry point (landing_kinodyn_solve_batch), wall time per batch and the outcome counts.  One JSON line."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024); ap.add_argument("--seed", type=int, default=20211); ap.add_argument("--law", default="main")
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--max-iter", type=int, default=500)
ap.add_argument("--opt", action="append", default=[], help="solver option override k=v (repeatable)")
ap.add_argument("--inflight", type=int, default=0, help="K > 1: also K batches in flight (K contexts on K streams, one host thread each): what a data-generation job streaming batches does")
ap.add_argument("--dump", default="", help="write the per-member status / iteration arrays there (.npz)")
ap.add_argument("--ik", type=int, default=0, help="1: joint-angle guess = inverse kinematics of the SRBM feet (Rbd.kinodynamic_screen) instead of the data-generation caller's constant guess")
a = ap.parse_args()
import torch
P_ = importlib.import_module("landing-controller_amd.problem"); capi = importlib.import_module("landing-controller_amd.capi")
rbd = importlib.import_module("landing-controller_amd.rbd"); kd = importlib.import_module("landing-controller_amd.kinodyn")
K = importlib.import_module("landing-controller_amd.constants")
N, B = 20, a.B
consts = P_.production_constants(a.law)
P, X0, q, qd = P_.make_batch(B, N, 0.6, seed=a.seed, consts=consts, dt_grid="reference", law=a.law)
L = capi.LandingLib(N, device=0, lib_path=os.environ.get("LANDING_LIB")); R = rbd.Rbd(L)
t = time.perf_counter(); srbm = L.solve_host(P, X0); t_srbm = time.perf_counter() - t
mass, Ib, Ibi = K.robot_constants()
jp_guess = [None] * B
t_ik = 0.0
if a.ik:
    torch.cuda.synchronize(); t = time.perf_counter()
    scr = R.kinodynamic_screen(N, torch.tensor(srbm["x"], device="cuda"))
    jp_guess = scr["jpos"].cpu().numpy().transpose(0, 2, 1)      # [B, 12, N]
    t_ik = time.perf_counter() - t
prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b], jp_guess[b]) for b in range(B)]
lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
T = lambda v: torch.tensor(v, device="cuda")
dl, du, dc, dx0 = T(lb), T(ub), T(cost), T(x0)
nx, ng = kd.dims(N)
x = torch.empty(B, nx, device="cuda", dtype=torch.float64); kk = torch.empty(B, 3, device="cuda", dtype=torch.float64)
st = torch.empty(B, device="cuda", dtype=torch.int32); it = torch.empty(B, device="cuda", dtype=torch.int32)
o = R.kinodyn_default_opts(); o.max_iter = a.max_iter
for kv in a.opt:
    k_, v_ = kv.split("="); setattr(o, k_, type(getattr(o, k_))(float(v_)))
times = []
for _ in range(a.reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    R.kinodyn_solve_device(B, N, dl.data_ptr(), du.data_ptr(), dc.data_ptr(), dx0.data_ptr(), P_.REFERENCE_DT_GRID, mass, Ib, Ibi, consts.mu, o, x.data_ptr(),
                           d_status=st.data_ptr(), d_iters=it.data_ptr(), d_kkt=kk.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t)
piped = None
if a.inflight > 1:      # the latency-bound tail rounds of one batch (a handful of members, most CUs idle) overlap with the full rounds of another
    import threading
    lanes = []
    for j in range(a.inflight):
        Lj = capi.LandingLib(N, device=0, lib_path=os.environ.get("LANDING_LIB")); Rj = rbd.Rbd(Lj)
        lanes.append((Lj, Rj, torch.cuda.Stream(), torch.empty_like(x), torch.empty_like(st), torch.empty_like(it)))
    def work(j, n):
        Lj, Rj, sj, xj, stj, itj = lanes[j]
        for _ in range(n):
            Rj.kinodyn_solve_device(B, N, dl.data_ptr(), du.data_ptr(), dc.data_ptr(), dx0.data_ptr(), P_.REFERENCE_DT_GRID, mass, Ib, Ibi, consts.mu, o, xj.data_ptr(),
                                    d_status=stj.data_ptr(), d_iters=itj.data_ptr(), stream=sj.cuda_stream)
        sj.synchronize()
    for j in range(a.inflight): work(j, 1)      # (contexts' tables and workspaces)
    torch.cuda.synchronize(); t = time.perf_counter()
    th = [threading.Thread(target=work, args=(j, a.reps)) for j in range(a.inflight)]
    for h in th: h.start()
    for h in th: h.join()
    torch.cuda.synchronize(); tp = time.perf_counter() - t
    same = all(bool(torch.equal(l[3], x)) and bool(torch.equal(l[4], st)) for l in lanes)
    piped = {"batches_in_flight": a.inflight, "batches": a.inflight * a.reps, "wall_s": tp, "s_per_batch": tp / (a.inflight * a.reps), "members_per_s": B * a.inflight * a.reps / tp,
             "same_results_as_one_at_a_time": same}
    for l in lanes: l[0].close()
s, i, k = st.cpu().numpy(), it.cpu().numpy(), kk.cpu().numpy()
ok = s == 0
if a.dump:
    np.savez(a.dump, status=s, iters=i, kkt=k)
print(json.dumps({"opts": a.opt, "what": "kinodynamic refinement of %d SRBM solutions (N = 20, production grid, law %s, seed %d)" % (B, a.law, a.seed), "batch": B,
                  "srbm_solve_s": t_srbm, "jpos_guess": "inverse kinematics of the SRBM feet (%.4f s)" % t_ik if a.ik else "constant (generate_training_data_automated.m:143)", "srbm_converged": int((srbm["status"] == 0).sum()), "refinement_s": times, "refinement_s_best": min(times),
                  "status_counts": np.bincount(s, minlength=4).tolist(), "converged": int(ok.sum()), "certified_infeasible": int((s == 3).sum()),
                  "iters_mean_converged": float(i[ok].mean()), "iters_p99_converged": float(np.percentile(i[ok], 99)), "iters_max": int(i.max()),
                  "kkt_max_converged": k[ok].max(axis=0).tolist(), "refined_per_s": float(ok.sum() / min(times)), "members_per_s": B / min(times), "in_flight": piped, "rounds": int(i.max()) + 1}))
