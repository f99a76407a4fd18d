"""Development probe: per-member iteration counts, phase timers and counters of the bench batch -> gpurun_out/itdump_<tag>.npz"""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
tag = sys.argv[1] if len(sys.argv) > 1 else "x"
N, B = 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=20211)
L = capi.LandingLib(N, 0)
o = L.default_opts(); o.max_iter = 300
prof = torch.zeros(B, 16, device='cuda', dtype=torch.float64)
L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
r = L.solve_host(P, X0, o)
L.lib.landing_set_profile_buffer(L.ctx, None)
ph = prof.cpu().numpy()
t = time.time(); r2 = L.solve_host(P, X0, o); dt = time.time() - t
it = r['iters']; c = r['status'] == 0
print('conv', c.sum(), 'iters mean %.1f med %.0f p90 %.0f p99 %.0f max %d' % (it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max()))
print('fact/iter %.3f trials/iter %.3f  host-path sec %.3f' % (ph[:, 8].sum() / ph[:, 10].sum(), ph[:, 9].sum() / ph[:, 10].sum(), dt))
tot = ph[:, :8].sum(axis=0) / 1e5 / ph[:, 10].sum()     # ms per iteration (100 MHz ticks)
print('ms/iter under load by phase (eval err sigrho back fwd dual ls accept):', np.round(tot, 4), 'sum %.4f' % tot.sum())
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', 'itdump_%s.npz' % tag), iters=it, status=r['status'], prof=ph, kkt=r['kkt'])
import json, heapq
def _makespan(t, order, slots=512):      # list scheduling: the next member of `order` goes to the slot that frees first
    h = [0.0] * slots; heapq.heapify(h)
    for m in order: heapq.heappush(h, heapq.heappop(h) + t[m])
    return float(max(h))
hist, edges = np.histogram(it, bins=[0, 30, 40, 50, 60, 70, 80, 90, 100, 120, 140, 160, 200, 250, 301])
busy_ms = ph[:, :8].sum(axis=1) / 1e5          # per member: time its workgroup spent in the timed phases (100 MHz ticks)
json.dump({"what": "bench batch (seed 20211, B=%d, N=40, max_iter 300): iteration histogram and per-member busy time of the solver kernel" % B,
           "iters_hist": {"%d-%d" % (edges[i], edges[i + 1] - 1): int(hist[i]) for i in range(len(hist))},
           "iters_mean": float(it.mean()), "iters_p50": float(np.median(it)), "iters_p90": float(np.percentile(it, 90)), "iters_p99": float(np.percentile(it, 99)), "iters_max": int(it.max()),
           "member_busy_ms": {"mean": float(busy_ms.mean()), "p50": float(np.median(busy_ms)), "p99": float(np.percentile(busy_ms, 99)), "max": float(busy_ms.max()), "sum": float(busy_ms.sum())},
           "slots": 512, "balanced_bound_ms": float(busy_ms.sum() / 512), "host_path_batch_ms": 1e3 * dt,
           "list_scheduling_of_the_busy_times_ms": {"note": "1024 members on 512 slots = two members per slot: what ANY dispatch order can reach is the best pairing of a long with a short member, not the balanced bound",
                                                    "library_order_initial_height": _makespan(busy_ms, np.argsort(-P[:, problem.param_offsets(N)["q_init"] + 2], kind="stable")),
                                                    "index_order": _makespan(busy_ms, np.arange(B)), "longest_first_with_hindsight": _makespan(busy_ms, np.argsort(-busy_ms, kind="stable"))},
           "phase_ms_per_iter_under_load": dict(zip(["eval", "err", "cond", "back", "fwd", "dual", "ls", "accept"], [float(v) for v in tot])),
           "sweeps_per_iter": float(ph[:, 8].sum() / ph[:, 10].sum()), "stage_elims_per_iter": float(ph[:, 13].sum() / ph[:, 10].sum()), "trials_per_iter": float(ph[:, 9].sum() / ph[:, 10].sum())},
          open(os.path.join(ROOT, 'gpurun_out', 'iter_hist_%s.json' % tag), 'w'), indent=1)
