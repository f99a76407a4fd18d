#!/usr/bin/env python3
"""bench.py -- landing NLPs solved/sec (SRBM, N=40 intervals, batch 1024 per GPU) on 1..8 MI355X.

A "step" is one pass of the hot path over one batch: landing_solve_batch() on the rank's 1024 synthetic
drop states (inputs already resident in HBM), followed -- for --gpus > 1 -- by the RCCL all-gather of the
solved trajectories and status words (the only communication of the path).  One process per
GPU; N>1 is launched by torch.distributed.run.  Rank 0 prints ONE JSON line.

  value            = NLPs that reached the KKT tolerance (status CONVERGED, unscaled pr/du/compl <= 1e-6)
                     over all ranks and timed steps / wall time (barrier + synchronize on both sides, max over ranks)
  roofline         = the dominant kernel (landing_ipm_kernel): algorithmic fp64 flops of the implemented
                     recursion (counted per launch from the kernel's own iteration / factorisation / trial
                     counters, model in DESIGN.md) / its HIP-event duration, against the fp64 matrix peak
  sweep_roofline   = the function-layer sweep kernel (landing_sweep_kernel, HBM bound): algorithmic bytes
                     (278 288 B per NLP at N=40) / HIP-event duration
  cpu_baseline     = oracle/landing_solver_cpu.c (scalar fp64 port of the same algorithm, OpenMP over
                     members) on a bounded sample of the same workload, on this box's host cores (rank 0 only)
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_INTERVALS = 40
BATCH_PER_GPU = 1024
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (AMD datasheet)
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md


def flop_model(N):
    """fp64 flops of the implemented recursion (DESIGN.md 'flop model'), per factorisation / iteration / trial.
    Algorithmic counts (no tile padding): multiply-add = 2."""
    def stage(nu):
        nr, nc = nu + 24, nu + 25                      # rows (u, sigma) and columns (u, sigma, gamma) of the stage array
        y = 24 * 12 * 37 * 2                           # Y = P(:,0:12) [A^ | b]
        tpt = 36 * nc * 12 * 2 + 24 * nc               # G += A^T Y  (+ the c+ rows / columns and p entering directly)
        gj = (nu // 4) * (nr * nc * 4 * 2 + nc * 28 + 60)   # blocked Gauss-Jordan: rank-4 updates, 4x4 solves, 4x4 LDL^T
        cl = 12 * 12 * 25 * 2 + 12 * 25                # closed-loop map Mt = A^_s - A^_f K_f, mv
        return y + tpt + gj + cl
    fact = (N - 1) * stage(24) + stage(12) + 12 * 12 * 13 * 2 + 12 * 24 * 2      # + stage-0 foot block
    nnz_j = 36 + 385 * (N - 1) + 313
    n_terms = 1125 * N                                  # condensation terms per stage (tables in solver_capi.inc)
    it = (2 * 3500 * N          # Jacobian (twice: store + J^T y) and Hessian values, ~3.5 kflop per stage sweep each
          + 3 * n_terms + N * (24 * 24 * 2 * 2 + 12 * 24 * 2) + 2 * (nnz_j - 149 * N) + 40 * (104 * N + 12))
    trial = 360 * N + 12 * (104 * N + 12)
    return fact, it, trial


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="NLPs per GPU (weak scaling)")
    ap.add_argument("--max-iter", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    capi = importlib.import_module("landing-controller_amd.capi")
    problem = importlib.import_module("landing-controller_amd.problem")
    sharding = importlib.import_module("landing-controller_amd.sharding")
    N, B = N_INTERVALS, a.batch
    lib = capi.LandingLib(N, device=local)

    # synthetic drop states of SURVEY 8(d); every rank its own shard of the sweep (seed = 20211 + rank)
    P, X0, _, _ = problem.make_batch(B, N, 0.6, seed=20211 + rank)
    dP, dX0 = torch.tensor(P, device=dev), torch.tensor(X0, device=dev)
    mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
    x, f, lam, kkt = mk(B, lib.nx), mk(B), mk(B, lib.ng), mk(B, 3)
    st, it = mk(B, dt=torch.int32), mk(B, dt=torch.int32)
    if world > 1:
        xg, stg = mk(world * B, lib.nx), mk(world * B, dt=torch.int32)
    opts = lib.default_opts()
    opts.max_iter = a.max_iter
    stream = torch.cuda.current_stream().cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kernel_ms = []

    def step(timed):
        if timed:
            ev0.record()
        lib.solve_device(B, dP.data_ptr(), dX0.data_ptr(), opts, x.data_ptr(), f.data_ptr(), lam.data_ptr(), st.data_ptr(),
                         it.data_ptr(), kkt.data_ptr(), stream)
        if timed:
            ev1.record()
        if world > 1:   # collect the solved trajectories (RCCL over xGMI)
            sharding.gather_solutions(x, st, xg, stg)
        return (ev0, ev1)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
        torch.cuda.synchronize()            # events of this step are complete; the launch is asynchronous otherwise
        kernel_ms.append(ev0.elapsed_time(ev1))
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    per_rank = None
    if world > 1:      # per-GPU time spread (inter-GPU load imbalance is reported, not balanced)
        allel = [torch.zeros_like(el) for _ in range(world)]
        dist.all_gather(allel, el)
        per_rank = [1e3 * float(t.item()) / a.steps for t in allel]
    conv = (st == 0).sum().to(torch.float64).reshape(1)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(conv, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    solved_per_step = float(conv.item())

    out = None
    if rank == 0:
        sth, ith, kh = st.cpu().numpy(), it.cpu().numpy(), kkt.cpu().numpy()
        ok = sth == 0
        # ---- roofline of the dominant kernel: counters from one extra (untimed) instrumented pass
        prof = mk(B, 16)
        prof.zero_()
        lib.lib.landing_set_profile_buffer(lib.ctx, prof.data_ptr())
        step(False)
        torch.cuda.synchronize()
        lib.lib.landing_set_profile_buffer(lib.ctx, None)
        ph = prof.cpu().numpy()
        n_fact, n_trial, n_iter = ph[:, 8].sum(), ph[:, 9].sum(), ph[:, 10].sum()
        f_fact, f_it, f_trial = flop_model(N)
        flops = n_fact * f_fact + n_iter * f_it + n_trial * f_trial
        k_ms = float(np.mean(kernel_ms))
        achieved = flops / (k_ms * 1e-3) / 1e12
        roofline = {"kernel": "landing_ipm_kernel", "bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None, "launch_ms": k_ms, "flops_per_launch": flops,
                    "note": "fp64; latency-bound persistent kernel (one workgroup per NLP, 3 per CU): serial chain of 40 stages x 6 block-pivot steps per factorisation; T^T P T, the blocked Gauss-Jordan elimination and the closed-loop map run on v_mfma_f64_16x16x4"}
        # ---- function-layer sweep kernel (HBM bound)
        Bs = 4096
        reps = (Bs + B - 1) // B
        sx = dX0.repeat(reps, 1)[:Bs].contiguous(); sp = dP.repeat(reps, 1)[:Bs].contiguous()
        sl = torch.randn(Bs, lib.ng, device=dev, dtype=torch.float64)
        sg, sgf, sj, sh = mk(Bs, lib.ng), mk(Bs, lib.nx), mk(Bs, lib.nnz_jac), mk(Bs, lib.nnz_hess)
        run = lambda: lib.eval_device(Bs, sx.data_ptr(), sp.data_ptr(), 0, sl.data_ptr(), 0, sg.data_ptr(), sgf.data_ptr(), sj.data_ptr(), sh.data_ptr(), 0, 0, stream)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(20):
            run()
        ev1.record()
        torch.cuda.synchronize()
        s_ms = ev0.elapsed_time(ev1) / 20
        by = lib.lib.landing_sweep_bytes_per_member(N) * Bs
        sweep = {"kernel": "landing_sweep_kernel<0>+<1>+<2>+landing_sweep_misc_kernel (one landing_eval_batch call)", "bound": "hbm", "achieved": by / s_ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": by / s_ms / 1e6 / HBM_PEAK_GBPS, "traffic": None, "launch_ms": s_ms, "members": Bs}
        # ---- CPU baseline (oracle port) on a bounded sample of the same workload
        cpu = None
        if not a.no_cpu_baseline and world == 1:      # CPU legs on rank 0 of the single-GPU run only
            from oracle import oracle as orc
            orc.build()
            O = orc.Oracle(N)
            cores = os.cpu_count() or 1
            ns = int(min(B, max(16, 4 * cores)))
            tc = time.perf_counter()
            r = orc.cpu_solve_batch(O, P[:ns], X0[:ns], threads=cores, max_iter=a.max_iter)
            tcpu = time.perf_counter() - tc
            # function layer on the host: full derivative sweeps/s of the CPU restatement, one core and all cores
            nsw = int(min(B, 4 * cores))
            lam_h = np.random.default_rng(0).normal(size=(nsw, lib.ng))
            t1 = time.perf_counter(); orc.cpu_sweep_batch(O, X0[:16], P[:16], lam_h[:16], reps=4, threads=1); t1 = time.perf_counter() - t1
            ta = time.perf_counter(); orc.cpu_sweep_batch(O, X0[:nsw], P[:nsw], lam_h, reps=4, threads=cores); ta = time.perf_counter() - ta
            cpu_fn = {"unit": "derivative sweeps/s (278 288 algorithmic bytes each)", "one_core": 64 / t1, "all_cores": 4 * nsw / ta, "cores": cores,
                      "kind": "port", "sample": f"64 sweeps on one core, {4 * nsw} sweeps on {cores} cores (oracle/landing_oracle.c, OpenMP over members)",
                      "gpu_sweeps_per_s": Bs / (s_ms * 1e-3)}
            cpu = {"value": float((r["status"] == 0).sum() / tcpu), "unit": "NLPs/s", "cores": cores, "kind": "port", "function_layer": cpu_fn,
                   "sample": f"first {ns} members of rank 0's batch (N=40), max_iter {a.max_iter}, OpenMP over members, {tcpu:.1f} s",
                   "converged": int((r["status"] == 0).sum()), "gpu_converged_same_members": int(ok[:ns].sum())}
        out = {
            "metric": "landing NLPs solved/sec (SRBM, N=40, batch)", "value": solved_per_step * a.steps / elapsed, "unit": "NLPs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D-SRBM landing NLP, N=40 intervals, batch=%d random drop heights/attitudes per GPU, fp64 (BASELINE configs[1])" % B,
                       "global_batch": B * world, "max_iter": a.max_iter, "kkt_tol": 1e-6, "parallelism": "batch-sharded x%d, RCCL all-gather of x*" % world},
            "solved_per_step": solved_per_step, "members_per_step": B * world, "ms_per_step_by_rank": per_rank,
            "kkt_max_over_solved": kh[ok].max(axis=0).tolist() if ok.any() else None,
            "iters_median": float(np.median(ith)), "iters_mean": float(ith.mean()),
            "roofline": roofline, "sweep_roofline": sweep, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
