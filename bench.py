#!/usr/bin/env python3
"""bench.py -- landing NLPs solved/sec (SRBM, N=40 intervals, batch 1024 per GPU) on 1..8 MI355X.

A "step" is one pass of the hot path over one batch: landing_solve_batch() on the rank's 1024 synthetic
drop states (inputs already resident in HBM), followed -- for --gpus > 1 -- by the RCCL all-gather of the
solved trajectories and status words (the only communication of the path).  One process per GPU.

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself:
the parent -- BEFORE importing torch or touching HIP -- runs `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>` as a child process, relays its
output and exits with its code.  Under an external torchrun (RANK / LOCAL_RANK / WORLD_SIZE set) it runs as one rank.
Rank 0 prints ONE JSON line.  `--backend gloo --dry` exercises exactly this multi-rank code path (rendezvous,
sharding, gather, per-rank timing, reductions, JSON) on CPU with the solve replaced by a stub; the line then says
"dry": true and carries no measurement (CPU test of the launcher, tests/test_bench_launcher_cpu.py).

  value            = NLPs that reached the KKT tolerance (status CONVERGED, unscaled pr/du/compl <= 1e-6)
                     over all ranks and timed steps / wall time (barrier + synchronize on both sides, max over ranks);
                     inputs resident in HBM.  `pcie_inclusive` repeats the measurement with H2D of p, x0 from pinned host
                     memory and D2H of x*, status inside the timed step (SURVEY 8d's wording of the metric).
  roofline         = the dominant kernel (landing_ipm_kernel): fp64 flops per launch / its HIP-event duration against the
                     fp64 matrix peak, two ways: `frac` counts SURVEY 8(d)'s algorithmic figure (4.4 MFLOP per KKT solve +
                     0.14 MFLOP of callbacks per iteration) x iterations; `frac_implemented` counts the flops of the
                     implemented recursion (model in DESIGN.md) for the SUCCESSFUL stage eliminations only.
                     `traffic` = HBM bytes per launch from the rocprofv3 --pmc passes committed under profiles/ (same
                     command; bench.py cannot run the profiler on itself).
  sweep_roofline   = the function-layer sweep kernel (HBM bound): algorithmic bytes (278 288 B per NLP at
                     N=40) / HIP-event duration
  cpu_baseline     = oracle/landing_solver_cpu.c (scalar fp64 port of the same algorithm, OpenMP over
                     members) on a bounded sample of the same workload, on this box's host cores (rank 0, N=1 only)
  next_rows        = SURVEY 8(f) rows measured beside the headline, behind the timed region, never part of `value` (N=1 only):
                     kinodyn_refinement = row N1, one batch of 1024 through landing_kinodyn_solve_batch (tools/bench_kd_solve.py)
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_INTERVALS = 40
BATCH_PER_GPU = 1024
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector = matrix peak (AMD datasheet)
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md
SURVEY_8D_KKT_FLOPS = 4.4e6  # block-tridiagonal factor+solve per interior-point iteration at N=40
SURVEY_8D_CALLBACK_FLOPS = 0.14e6
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_ipm.json")      # default of --pmc-file


def kernel_source_sha():
    """sha256 over the sources that make up landing_ipm_kernel: stamped into profiles/*_pmc_ipm.json by tools/pmc_summary.py and
    compared here, so that a PMC traffic figure measured on another build of the kernel is never attached to this run"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "landing-controller_amd", "csrc")
    for f in ("solver_kernels.hip", "eval_kernels.hip", "srbm_stage.hpp", "layout.hpp", "Makefile"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def flop_model(N):
    """fp64 flops of the implemented recursion (DESIGN.md 'flop model'): per stage elimination (middle / last stage),
    per iteration outside the factorisation, per trial point.  Algorithmic counts (no tile padding): multiply-add = 2."""
    def stage(nu):
        nr, nc = nu + 24, nu + 25                      # rows (u, sigma) and columns (u, sigma, gamma) of the stage array
        y = 24 * 12 * 37 * 2                           # Y = P(:,0:12) [A^ | b]
        tpt = 36 * nc * 12 * 2 + 24 * nc               # G += A^T Y  (+ the c+ rows / columns and p entering directly)
        gj = (nu // 4) * (nr * nc * 4 * 2 + nc * 28 + 60)   # blocked Gauss-Jordan: rank-4 updates, 4x4 solves, 4x4 LDL^T
        cl = 12 * 12 * 25 * 2 + 12 * 25                # closed-loop map Mt = A^_s - A^_f K_f, mv
        return y + tpt + gj + cl
    foot = 12 * 12 * 13 * 2 + 12 * 24 * 2             # stage-0 foot block
    nnz_j = 36 + 385 * (N - 1) + 313
    n_terms = 1125 * N                                  # condensation terms per stage (tables in solver_capi.inc)
    it = (2 * 3500 * N          # Jacobian (twice: store + J^T y) and Hessian values, ~3.5 kflop per stage sweep each
          + 3 * n_terms + N * (24 * 24 * 2 * 2 + 12 * 24 * 2) + 2 * (nnz_j - 149 * N) + 40 * (104 * N + 12))
    trial = 360 * N + 12 * (104 * N + 12)
    return stage(24), stage(12), foot, it, trial


def usable_cores():
    """cores this process may run on: the affinity mask, cut by the cgroup CPU quota (v2 cpu.max, v1 cfs quota) when one is set"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return max(1, n)


# the reference's own generated C (codegen_casadi/landingCtrller_IPOPT.c, gcc -O1, N = 20 intervals) timed through
# ctypes in the build container (8 cores) -- the constant carried when the compiled library is not beside this script
REFERENCE_C_CONTAINER_US = {"nlp_f": 20.0, "nlp_g": 16.0, "nlp_grad_f": 13.0, "nlp_jac_g": 30.0, "nlp_hess_l": 30.0}


def reference_generated_c_leg(np):
    """The reference's CPU path as far as it exists as code: the five CasADi callbacks IPOPT calls once per iteration, from the reference's
    own generated C compiled in place by oracle/Makefile into oracle/_ref/ (N = 20 -- the reference ships no N = 40 code; IPOPT + MA57 are
    absent third-party binaries, so the solver itself cannot be timed anywhere).  Timed live on this box's host (one core) when the library
    travelled with the repository, else the container-measured constants of BASELINE.md section 2, labelled as such."""
    from oracle import oracle as orc
    path = os.path.join(ROOT, "oracle", "_ref", "liblanding_ref.so")
    out = {"kind": "reference", "N": 20, "unit": "us per call, one core",
           "what": "nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l of codegen_casadi/landingCtrller_IPOPT.c (gcc -O1) through ctypes"}
    if not os.path.exists(path):
        out.update({"source": "BASELINE.md section 2 (build container, 8-core CPU, not this box): oracle/_ref/liblanding_ref.so did not travel",
                    "per_call_us": REFERENCE_C_CONTAINER_US, "callback_set_us": sum(REFERENCE_C_CONTAINER_US.values())})
        return out
    R = orc.RefOracle(path)
    rng = np.random.default_rng(5)
    x, p, lam = rng.normal(size=R.nx) * 0.1, np.abs(rng.normal(size=R.np_)) + 0.1, rng.normal(size=R.ng)
    calls = {"nlp_f": lambda: R.f(x, p), "nlp_g": lambda: R.g(x, p), "nlp_grad_f": lambda: R.grad_f(x, p), "nlp_jac_g": lambda: R.jac_g(x, p),
             "nlp_hess_l": lambda: R.hess_l(x, p, 1.0, lam)}
    per = {}
    for k_, fn in calls.items():
        for _ in range(20):
            fn()
        n = 400
        t = time.perf_counter()
        for _ in range(n):
            fn()
        per[k_] = 1e6 * (time.perf_counter() - t) / n
    out.update({"source": "measured in this run on this box's host CPU (ctypes call overhead included), 400 calls each after 20 warm-up calls",
                "per_call_us": per, "callback_set_us": sum(per.values()),
                "note": "one callback set per IPOPT iteration; the KKT factorisation (MA57) that dominates the reference's iteration is not in the tree"})
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Parent of a multi-GPU run: start n ranks through torch.distributed.run.  Nothing in this process has imported
    torch or initialised HIP (an exec/fork after GPU initialisation is forbidden on this pool)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="NLPs per GPU (weak scaling)")
    ap.add_argument("--max-iter", type=int, default=300)
    ap.add_argument("--distinct-batches", type=int, default=8, help="timed steps cycle through this many different synthetic batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc-file", default=PMC_FILE, help="rocprofv3 --pmc summary (tools/profile_round.sh) the HBM traffic of the roofline object is read from; ignored unless it was measured on this very kernel source")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo only with --dry (CPU launcher test)")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and two-batches-in-flight legs (profiling runs: only the headline launches)")
    ap.add_argument("--dry", action="store_true", help="no GPU work: exercise the multi-rank path with a stub solve")
    ap.add_argument("--force-dist", action="store_true", help="run the collective path (process group, RCCL gather, reductions) at world size 1 too: GPU smoke test of the multi-GPU code on a 1-GPU box")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.backend == "gloo" and not a.dry:
        raise SystemExit("--backend gloo is the CPU launcher test and needs --dry: the product has no CPU path")

    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and env_world is None:
        sys.exit(spawn_ranks(a.gpus, sys.argv[1:]))
    world = int(env_world) if env_world is not None else 1
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d -- launch with matching values" % (a.gpus, world))
    multi = world > 1 or a.force_dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    if a.dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product has no CPU path")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)

    problem = importlib.import_module("landing-controller_amd.problem")
    sharding = importlib.import_module("landing-controller_amd.sharding")
    N, B = N_INTERVALS, a.batch
    nx, ng = problem.nx(N), problem.ng(N)
    lib = None
    if not a.dry:
        capi = importlib.import_module("landing-controller_amd.capi")
        lib = capi.LandingLib(N, device=local)

    # synthetic drop states of SURVEY 8(d); every rank its own shard of the sweep (seed = 20211 + rank), and every timed
    # step its own batch (seed + 1000 * step): the batch time is set by the slowest of the 1024 members, which varies by
    # +-15 % from one random batch to the next -- the metric is the mean over the K batches, not one lucky or unlucky draw
    def batch_of(step):
        Pq, Xq, _, _ = problem.make_batch(B if not a.dry else min(B, 8), N, 0.6, seed=20211 + rank + 1000 * step)
        if a.dry:
            Pq = np.resize(Pq, (B, Pq.shape[1])); Xq = np.resize(Xq, (B, Xq.shape[1]))
        return Pq, Xq
    n_batches = max(1, min(a.steps, a.distinct_batches))
    host_batches = [batch_of(i) for i in range(n_batches)]
    dev_batches = [(torch.tensor(Pq, device=dev), torch.tensor(Xq, device=dev)) for Pq, Xq in host_batches]
    P, X0 = host_batches[0]
    dP, dX0 = dev_batches[0]
    mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
    x, f, lam, kkt = mk(B, nx), mk(B), mk(B, ng), mk(B, 3)
    st, it = mk(B, dt=torch.int32), mk(B, dt=torch.int32)
    xg = stg = None
    if multi:
        xg, stg = mk(world * B, nx), mk(world * B, dt=torch.int32)
    if a.dry:
        cuda_sync = lambda: None
        ev0 = ev1 = None
        opts = None
        stream = 0
    else:
        cuda_sync = torch.cuda.synchronize
        opts = lib.default_opts()
        opts.max_iter = a.max_iter
        stream = torch.cuda.current_stream().cuda_stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    kernel_ms = []

    def solve(dp, dx0):
        if a.dry:     # stub: "solution" = the initial guess, everything converged
            x.copy_(dx0); st.zero_(); it.fill_(1)
            return
        lib.solve_device(B, dp.data_ptr(), dx0.data_ptr(), opts, x.data_ptr(), f.data_ptr(), lam.data_ptr(), st.data_ptr(),
                         it.data_ptr(), kkt.data_ptr(), stream)

    def step(timed, i=0):
        if timed and ev0 is not None:
            ev0.record()
        solve(*dev_batches[i % n_batches])
        if timed and ev1 is not None:
            ev1.record()
        if multi:   # collect the solved trajectories (RCCL over xGMI)
            sharding.gather_solutions(x, st, xg, stg)

    def sync():
        cuda_sync()
        if multi:
            dist.barrier()
            cuda_sync()

    # warm-up steps are the timed step verbatim (events, the count of solved members, the event read-back): the first use of
    # every torch kernel loads its code object, ~0.2 s in total that otherwise lands in the first timed steps
    n_conv = torch.zeros(1, device=dev, dtype=torch.float64)
    for _ in range(a.warmup):
        step(True)
        cuda_sync()
        n_conv += (st == 0).sum()
        if ev0 is not None:
            ev0.elapsed_time(ev1)
    sync()
    n_conv.zero_()
    sync()
    step_ms = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        ts = time.perf_counter()
        step(True, i)
        cuda_sync()                         # events of this step are complete; the launch is asynchronous otherwise
        n_conv += (st == 0).sum()
        if ev0 is not None:
            kernel_ms.append(ev0.elapsed_time(ev1))
        step_ms.append(1e3 * (time.perf_counter() - ts))
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    per_rank = None
    if multi:      # per-GPU time spread (inter-GPU load imbalance is reported, not balanced)
        allel = [torch.zeros_like(el) for _ in range(world)]
        dist.all_gather(allel, el)
        per_rank = [1e3 * float(t.item()) / a.steps for t in allel]
    conv = n_conv / a.steps          # converged members per step, averaged over the timed steps
    if multi:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(conv, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    solved_per_step = float(conv.item())
    gathered_ok = None
    if multi:      # the gathered block of this rank must be its own solution (cheap self-check of the collective)
        gathered_ok = bool(torch.equal(xg[rank * B:(rank + 1) * B], x) and torch.equal(stg[rank * B:(rank + 1) * B], st))

    sweep_timing = measure_sweep(lib, torch, dev, mk, B, dX0, dP, stream, ev0, ev1) if (rank == 0 and not a.dry) else None

    # ---- the same step with the PCIe legs inside (SURVEY 8d wording): pinned host p, x0 -> HBM, x*, status -> host
    pcie = None
    if not a.dry and not a.no_extras:
        pinned = [(torch.tensor(Pq).pin_memory(), torch.tensor(Xq).pin_memory()) for Pq, Xq in host_batches]
        hx, hst = torch.empty(B, nx, dtype=torch.float64).pin_memory(), torch.empty(B, dtype=torch.int32).pin_memory()
        dP2, dX02 = torch.empty_like(dP), torch.empty_like(dX0)

        def step_pcie(i=0):
            hP, hX0 = pinned[i % n_batches]
            dP2.copy_(hP, non_blocking=True); dX02.copy_(hX0, non_blocking=True)
            solve(dP2, dX02)
            hx.copy_(x, non_blocking=True); hst.copy_(st, non_blocking=True)
            if multi:
                sharding.gather_solutions(x, st, xg, stg)
        step_pcie(); sync()
        tp = time.perf_counter()
        for i in range(a.steps):
            step_pcie(i)
            cuda_sync()
        sync()
        tp = time.perf_counter() - tp
        tpe = torch.tensor([tp], device=dev, dtype=torch.float64)
        if multi:
            dist.all_reduce(tpe, op=dist.ReduceOp.MAX)
        tp = float(tpe.item())
        pcie = {"value": solved_per_step * a.steps / tp, "unit": "NLPs/s", "ms_per_step": 1e3 * tp / a.steps,
                "bytes_per_step_per_gpu": int(8 * B * (P.shape[1] + 2 * nx) + 4 * B),
                "note": "H2D of p, x0 (pinned) and D2H of x*, status inside the timed step"}

    # ---- the same K steps through the library's own streaming entry points (round 6): ONE context, ONE caller stream, two launches in
    # flight on the library's lanes (landing_stream_create / _submit / _wait); the host waits for the ticket of submission i - 2, counts its converged
    # members and submits i while i - 1 runs (tools/dev/stream_probe.py: a stream-side wait on a third stream costs 12 % -- the HIP streams share hardware queues)
    streamed = None
    if not a.dry and a.steps >= 2 and not a.no_extras:
        S = lib.stream(2)
        sl = [(mk(B, nx), mk(B, dt=torch.int32), mk(B, dt=torch.int32)) for _ in range(2)]
        conv_s = torch.zeros(1, device=dev, dtype=torch.float64)
        def ssub(i):
            xq, stq, itq = sl[i % 2]
            dPq, dXq = dev_batches[i % n_batches]
            return S.submit(B, dPq.data_ptr(), dXq.data_ptr(), opts, xq.data_ptr(), 0, 0, stq.data_ptr(), itq.data_ptr(), 0, in_stream=stream)
        t_ = [ssub(0), ssub(1)]; S.sync(); sync()
        tk = []
        tsr = time.perf_counter()
        for i in range(a.steps):
            if i >= 2:
                S.wait(tk[i - 2]); conv_s += (sl[i % 2][1] == 0).sum()      # the host waits for submission i - 2 (a data-generation caller reads its results here); i - 1 keeps the GPU busy
            tk.append(ssub(i))
        for i in range(max(0, a.steps - 2), a.steps):
            S.wait(tk[i]); conv_s += (sl[i % 2][1] == 0).sum()
        sync()
        tsr = time.perf_counter() - tsr
        tse = torch.tensor([tsr], device=dev, dtype=torch.float64)
        if multi:
            dist.all_reduce(tse, op=dist.ReduceOp.MAX); dist.all_reduce(conv_s, op=dist.ReduceOp.SUM)
        streamed = {"value": float(conv_s.item()) / float(tse.item()), "unit": "NLPs/s", "lanes": 2, "ms_per_step": 1e3 * float(tse.item()) / a.steps,
                    "note": "same K steps through landing_stream_submit / landing_stream_wait: one context, one host thread, two launches in flight inside the library; the host waits for submission i - 2 before it submits i"}
        S.close()

    # ---- the same K steps with two batches in flight (two contexts on two streams): the tail of one batch -- a handful of
    # members that need 2-4x the mean iteration count while most CUs idle -- overlaps with the bulk of the next one.  What a
    # data-generation job streaming batches through the GPU does; reported beside the one-batch-at-a-time `value`.
    piped = None
    if not a.dry and a.steps >= 2 and not a.no_extras:
        capi2 = importlib.import_module("landing-controller_amd.capi")
        lanes = []
        for i in range(2):
            Lq = lib if i == 0 else capi2.LandingLib(N, device=local)
            sq = torch.cuda.Stream()
            lanes.append((Lq, sq, mk(B, nx), mk(B, dt=torch.int32), mk(B, dt=torch.int32)))
        def pstep(i):
            Lq, sq, xq, stq, itq = lanes[i % 2]
            dPq, dXq = dev_batches[i % n_batches]
            Lq.solve_device(B, dPq.data_ptr(), dXq.data_ptr(), opts, xq.data_ptr(), 0, 0, stq.data_ptr(), itq.data_ptr(), 0, sq.cuda_stream)
        pstep(0); pstep(1); sync()
        # converged-member counts are accumulated ON the lane's stream (one accumulator per lane), stream-ordered between the solve that
        # wrote the status buffer and the next solve of that lane that overwrites it (the read used to sit on another stream)
        conv_lane = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(2)]
        def pcount(i):
            with torch.cuda.stream(lanes[i % 2][1]):
                conv_lane[i % 2] += (lanes[i % 2][3] == 0).sum()
        tq = time.perf_counter()
        for i in range(a.steps):
            if i >= 2:
                pcount(i)
            pstep(i)
        for i in range(min(2, a.steps)):
            pcount(i)
        sync()
        conv_p = conv_lane[0] + conv_lane[1]
        tq = time.perf_counter() - tq
        tqe = torch.tensor([tq], device=dev, dtype=torch.float64)
        if multi:
            dist.all_reduce(tqe, op=dist.ReduceOp.MAX); dist.all_reduce(conv_p, op=dist.ReduceOp.SUM)
        piped = {"value": float(conv_p.item()) / float(tqe.item()), "unit": "NLPs/s", "batches_in_flight": 2, "ms_per_step": 1e3 * float(tqe.item()) / a.steps,
                 "note": "same K steps, two contexts on two streams, no synchronisation between steps"}
        lanes[1][0].close()

    if rank == 0:
        cfg = {"workload": "3D-SRBM landing NLP, N=40 intervals, batch=%d random drop heights/attitudes per GPU, fp64 (BASELINE configs[1]%s)" % (B, "; x%d GPUs = configs[2]" % world if world > 1 else ""),
               "global_batch": B * world, "distinct_batches": n_batches, "max_iter": a.max_iter, "kkt_tol": 1e-6,
               "solver_options": "library defaults; kappa_eps / theta_mu / mu_init / bound_push chosen by the formulation (terminal-cost form: 120 / 1.8 / 0.5 / 1.0; include/landing_nlp.h)", "parallelism": "batch-sharded x%d, RCCL all-gather of x*" % world}
        out = {"metric": "landing NLPs solved/sec (SRBM, N=40, batch)", "value": solved_per_step * a.steps / elapsed, "unit": "NLPs/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": cfg,
               "solved_per_step": solved_per_step, "members_per_step": B * world, "ms_per_step_by_rank": per_rank, "gather_self_check": gathered_ok,
               "ms_by_step_rank0": [round(v, 2) for v in step_ms]}
        if a.dry:
            out.update({"dry": True, "value": 0.0, "backend": a.backend, "note": "launcher dry run: stub solve, no measurement"})
        else:
            out.update(measure_details(a, lib, np, torch, dev, mk, B, N, P, X0, dP, dX0, st, it, kkt, kernel_ms, solve, world, stream, ev0, ev1, dev_batches, sweep_timing))
            out["pcie_inclusive"] = pcie
            out["two_batches_in_flight"] = piped
            out["streamed"] = streamed
            out["value_basis"] = "inputs resident in HBM when the timed region starts, ONE batch in flight (bench contract of the task statement: the PCIe-inclusive rate 'is never value'); pcie_inclusive and streamed are the same K steps with the PCIe legs inside / through the library's streaming entry points"

            if world == 1 and not a.no_extras:
                out["next_rows"] = {"kinodyn_refinement": measure_kinodyn(np, torch, local)}
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def measure_kinodyn(np, torch, local, B=1024, N=20, seed=20211, reps=2):
    """SURVEY 8f row N1 beside the headline (rank 0, behind the timed region, never part of `value`): the production callers' pipeline on one batch --
    SRBM solve (N = 20, production grid, law "main") -> kinodynamic refinement of the same drop states through the device-pointer entry point.  The same
    measurement as tools/bench_kd_solve.py (profiles/r06_kd_bench.json).  Any failure is reported as a string: this leg must not take the headline down."""
    try:
        P_ = importlib.import_module("landing-controller_amd.problem"); capi = importlib.import_module("landing-controller_amd.capi")
        rbd = importlib.import_module("landing-controller_amd.rbd"); kd = importlib.import_module("landing-controller_amd.kinodyn")
        K = importlib.import_module("landing-controller_amd.constants")
        consts = P_.production_constants("main")
        P, X0, q, qd = P_.make_batch(B, N, 0.6, seed=seed, consts=consts, dt_grid="reference", law="main")
        L = capi.LandingLib(N, device=local); R = rbd.Rbd(L)
        srbm = L.solve_host(P, X0)
        mass, Ib, Ibi = K.robot_constants()
        prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b], None) for b in range(B)]
        lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
        T = lambda v: torch.tensor(v, device="cuda:%d" % local)
        dl, du, dc, dx0 = T(lb), T(ub), T(cost), T(x0)
        nx, ng = kd.dims(N)
        x = torch.empty(B, nx, device=dl.device, dtype=torch.float64); st = torch.empty(B, device=dl.device, dtype=torch.int32); it = torch.empty_like(st)
        o = R.kinodyn_default_opts()
        times = []
        for _ in range(reps + 1):      # (the first call builds the context's tables and workspace)
            torch.cuda.synchronize(); t = time.perf_counter()
            R.kinodyn_solve_device(B, N, dl.data_ptr(), du.data_ptr(), dc.data_ptr(), dx0.data_ptr(), P_.REFERENCE_DT_GRID, mass, Ib, Ibi, consts.mu, o, x.data_ptr(),
                                   d_status=st.data_ptr(), d_iters=it.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t)
        s, i = st.cpu().numpy(), it.cpu().numpy()
        L.close()
        ok = s == 0
        return {"what": "kinodynamic refinement (row N1) of %d SRBM solutions, N = 20, production grid, law main, seed %d; library defaults (portfolio of clone slots, DESIGN 4.8)" % (B, seed),
                "refinement_s": min(times[1:]), "members_per_s": B / min(times[1:]), "converged_per_s": float(ok.sum() / min(times[1:])),
                "status_counts": np.bincount(s, minlength=4).tolist(), "undecided": int(np.isin(s, (1, 2)).sum()), "iters_mean_converged": float(i[ok].mean()), "iters_max": int(i.max()),
                "srbm_converged": int((srbm["status"] == 0).sum())}
    except Exception as e:      # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def measure_sweep(lib, torch, dev, mk, B, dX0, dP, stream, ev0, ev1):
    """function-layer sweep (HBM bound): one landing_eval_batch call over 4096 members, HIP events.  Called right behind the timed region of
    the headline steps, before the side legs build their second context (their allocations and frees leave the sweep's 0.9 GB of outputs on
    other pages: measured 8 % slower behind them than in a fresh process -- tools/bench_sweep.py reproduces the fresh-process figure)"""
    Bs = 4096
    reps = (Bs + B - 1) // B
    sx = dX0.repeat(reps, 1)[:Bs].contiguous(); sp = dP.repeat(reps, 1)[:Bs].contiguous()
    sl = torch.randn(Bs, lib.ng, device=dev, dtype=torch.float64)
    sg, sgf, sj, sh = mk(Bs, lib.ng), mk(Bs, lib.nx), mk(Bs, lib.nnz_jac), mk(Bs, lib.nnz_hess)
    run = lambda: lib.eval_device(Bs, sx.data_ptr(), sp.data_ptr(), 0, sl.data_ptr(), 0, sg.data_ptr(), sgf.data_ptr(), sj.data_ptr(), sh.data_ptr(), 0, 0, stream)
    SW_REPS = 100      # ~35 ms timed
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(SW_REPS):
        run()
    ev1.record()
    torch.cuda.synchronize()
    return Bs, ev0.elapsed_time(ev1) / SW_REPS


def measure_details(a, lib, np, torch, dev, mk, B, N, P, X0, dP, dX0, st, it, kkt, kernel_ms, solve, world, stream, ev0, ev1, dev_batches, sweep_timing):
    """rank 0: roofline of the solver kernel, the sweep kernel, the CPU baseline"""
    # ---- roofline of the dominant kernel: counters from one extra (untimed) instrumented pass over the timed batches;
    # per-launch figures = mean over those batches
    nb = min(len(dev_batches), len(kernel_ms))
    cnt = np.zeros(16)
    ith_all, ok_all, kk_all = [], [], []
    prof = mk(B, 16)
    for b in range(nb):
        prof.zero_()
        lib.lib.landing_set_profile_buffer(lib.ctx, prof.data_ptr())
        solve(*dev_batches[b])
        torch.cuda.synchronize()
        lib.lib.landing_set_profile_buffer(lib.ctx, None)
        cnt += prof.cpu().numpy().sum(axis=0)
        ith_all.append(it.cpu().numpy().copy()); ok_all.append(st.cpu().numpy() == 0); kk_all.append(kkt.cpu().numpy().copy())
    cnt /= nb
    ith, ok, kh = np.concatenate(ith_all), np.concatenate(ok_all), np.concatenate(kk_all)
    n_fact, n_trial, n_iter = cnt[8], cnt[9], cnt[10]
    n_stage_ok, n_stage_all = cnt[11], cnt[13]      # stage eliminations: successful / attempted
    f_mid, f_last, f_foot, f_it, f_trial = flop_model(N)
    # successful eliminations only: every iteration ends with exactly one complete sweep (N-1 middle stages, the last
    # stage, the foot block); eliminations redone after an inertia failure are overhead, not useful work
    flops_impl = n_iter * ((N - 1) * f_mid + f_last + f_foot) + n_iter * f_it + n_trial * f_trial
    flops_8d = n_iter * (SURVEY_8D_KKT_FLOPS + SURVEY_8D_CALLBACK_FLOPS)
    k_ms = float(np.mean([np.mean(kernel_ms[b::len(dev_batches)]) for b in range(nb)]))
    ach8d = flops_8d / (k_ms * 1e-3) / 1e12
    achim = flops_impl / (k_ms * 1e-3) / 1e12
    traffic = None
    pmc_note = "no PMC file (%s)" % os.path.relpath(a.pmc_file, ROOT)
    if os.path.exists(a.pmc_file):
        try:
            pm = json.load(open(a.pmc_file))
            if pm.get("kernel_source_sha256") == kernel_source_sha():
                traffic = pm.get("traffic_bytes_per_launch")
                pmc_note = "%s (kernel sources %s..., command: %s); %s" % (os.path.relpath(a.pmc_file, ROOT), pm["kernel_source_sha256"][:12], pm.get("what", "?"), pm.get("note", ""))
            else:
                pmc_note = "%s was measured on other kernel sources (sha %s..., this build %s...): traffic not reported" % (
                    os.path.relpath(a.pmc_file, ROOT), str(pm.get("kernel_source_sha256"))[:12], kernel_source_sha()[:12])
        except Exception as e:   # noqa: BLE001
            pmc_note = "unreadable PMC file: %s" % e
    roofline = {"kernel": "landing_ipm_kernel", "bound": "mfma", "achieved": ach8d, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach8d / FP64_PEAK_TFLOPS, "traffic": traffic, "launch_ms": k_ms,
                "flops_per_launch": flops_8d, "basis": "SURVEY 8(d): (4.4 + 0.14) MFLOP x %d iterations" % int(n_iter),
                "achieved_implemented": achim, "frac_implemented": achim / FP64_PEAK_TFLOPS, "flops_per_launch_implemented": flops_impl,
                "iterations": int(n_iter), "sweeps_started": int(n_fact), "stage_eliminations_ok": int(n_stage_ok),
                "stage_eliminations_attempted": int(n_stage_all),
                "factorisation_equivalents_per_iteration": float(n_stage_all / max(1.0, n_iter * N)),
                "trial_points": int(n_trial), "traffic_source": pmc_note,
                "note": "fp64; latency-bound persistent kernel (one 256-thread workgroup per NLP, 2 per CU): serial chain of 40 stages x 6 block-pivot steps per sweep; T^T P T, the blocked Gauss-Jordan elimination and the closed-loop map run on v_mfma_f64_16x16x4"}
    # ---- function-layer sweep kernel (HBM bound): timed by measure_sweep() right behind the headline steps
    Bs, s_ms = sweep_timing
    by = lib.lib.landing_sweep_bytes_per_member(N) * Bs
    sweep = {"kernel": "one landing_eval_batch call (g, grad f, Jacobian and Hessian nonzeros)", "bound": "hbm", "achieved": by / s_ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": by / s_ms / 1e6 / HBM_PEAK_GBPS, "traffic": None, "launch_ms": s_ms, "members": Bs, "algorithmic_bytes": by}
    sw_file = os.path.join(os.path.dirname(a.pmc_file), os.path.basename(a.pmc_file).replace("pmc_ipm", "pmc_sweep").replace("summary.json", "summary_sweep.json"))
    if os.path.exists(sw_file) and sw_file != a.pmc_file:      # HBM bytes of one call from the PMC passes of tools/profile_round.sh -- same source-stamp rule as above
        try:
            sw = json.load(open(sw_file))
            if sw.get("kernel_source_sha256") == kernel_source_sha() and sw.get("members") == Bs:
                sweep["traffic"] = sw.get("traffic_bytes_per_call"); sweep["traffic_source"] = os.path.relpath(sw_file, ROOT)
            else:
                sweep["traffic_source"] = "%s was measured on other kernel sources or another batch: not reported" % os.path.relpath(sw_file, ROOT)
        except Exception as e:   # noqa: BLE001
            sweep["traffic_source"] = "unreadable: %s" % e
    # ---- CPU baseline (oracle port) on a bounded sample of the same workload
    cpu = None
    if not a.no_cpu_baseline and world == 1:      # CPU legs on rank 0 of the single-GPU run only
        from oracle import oracle as orc
        orc.build()
        O = orc.Oracle(N)
        cores = usable_cores()      # what this process may actually run on (affinity mask and cgroup quota), not the box's core count:
                                    # round 3 started os.cpu_count() = 256 threads on a box whose quota is far smaller and measured the
                                    # oversubscription (50 ms per iteration and thread against 3.5 ms on an unloaded core)
        # warm-up: thread pool, page faults of the per-member work arrays, code pages
        orc.cpu_solve_batch(O, P[:min(B, cores)], X0[:min(B, cores)], threads=cores, max_iter=3)
        t1 = time.perf_counter(); r1 = orc.cpu_solve_batch(O, P[:2], X0[:2], threads=1, max_iter=a.max_iter); t1 = time.perf_counter() - t1
        per_nlp = max(t1 / 2, 1e-3)
        ns = int(min(B, cores * max(1, min(64, int(15.0 / per_nlp)))))      # about 15 s of wall time at most (10-30 s of CPU work per core is the contract's sample size)
        tc = time.perf_counter()
        r = orc.cpu_solve_batch(O, P[:ns], X0[:ns], threads=cores, max_iter=a.max_iter)
        tcpu = time.perf_counter() - tc
        # function layer on the host: full derivative sweeps/s of the CPU restatement, one core and all cores, >= 32 sweeps per thread
        nsw = int(min(B, cores))
        lam_h = np.random.default_rng(0).normal(size=(max(nsw, 16), lib.ng))
        orc.cpu_sweep_batch(O, X0[:16], P[:16], lam_h[:16], reps=2, threads=1)                        # warm
        ts1 = time.perf_counter(); orc.cpu_sweep_batch(O, X0[:16], P[:16], lam_h[:16], reps=16, threads=1); ts1 = time.perf_counter() - ts1
        orc.cpu_sweep_batch(O, X0[:nsw], P[:nsw], lam_h[:nsw], reps=2, threads=cores)                   # warm
        ta = time.perf_counter(); orc.cpu_sweep_batch(O, X0[:nsw], P[:nsw], lam_h[:nsw], reps=32, threads=cores); ta = time.perf_counter() - ta
        cpu_fn = {"unit": "derivative sweeps/s (278 288 algorithmic bytes each)", "one_core": 256 / ts1, "all_cores": 32 * nsw / ta, "cores": cores,
                  "kind": "port", "sample": f"256 sweeps on one core, {32 * nsw} sweeps on {cores} threads (32 per thread; oracle/landing_oracle.c, OpenMP over members), warmed",
                  "gpu_sweeps_per_s": Bs / (s_ms * 1e-3)}
        cpu = {"value": float((r["status"] == 0).sum() / tcpu), "unit": "NLPs/s", "cores": cores, "kind": "port", "function_layer": cpu_fn,
               "one_core": {"value": float((r1["status"] == 0).sum() / t1), "unit": "NLPs/s", "ms_per_iteration": 1e3 * t1 / max(1, int(r1["iters"].sum()))},
               "ms_per_iteration_per_thread": 1e3 * tcpu * min(cores, ns) / max(1, int(r["iters"].sum())),
               "host_cpu_count": os.cpu_count(),
               "sample": f"first {ns} members of rank 0's batch (N=40), max_iter {a.max_iter}, OpenMP over members on {cores} usable cores, warmed, {tcpu:.1f} s",
               "converged": int((r["status"] == 0).sum()), "gpu_converged_same_members": int(ok_all[0][:ns].sum()),
               "reference_generated_c": reference_generated_c_leg(np)}
    return {"kkt_max_over_solved": kh[ok].max(axis=0).tolist() if ok.any() else None,
            "iters_median": float(np.median(ith)), "iters_mean": float(ith.mean()), "iters_max": int(ith.max()),
            "roofline": dict(roofline, sweep=sweep), "sweep_roofline": sweep, "cpu_baseline": cpu}      # (the sweep object also sits INSIDE roofline: the driver's `parsed` keeps the contract's keys only)


if __name__ == "__main__":
    main()
