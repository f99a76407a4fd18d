"""CPU-port lab for the feasibility-phase rules on the reference's production problem (N = 20, non-uniform grid): the drop states of
tools/soak.py --N 20 --grid reference --law <law> --batches <n> (same seeds), solved by oracle/landing_solver_cpu.c on the host cores.
   python tools/dev/feas_lab.py [--law datagen] [--batches 16] [--opts k=v,...] [--base k=v,...]
Prints outcome counts, iteration statistics and a makespan proxy (list scheduling of the iteration counts on 512 slots in dispatch
order = initial height, hard first) of one option set, and of a baseline set to compare member by member."""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as om
problem = importlib.import_module("landing-controller_amd.problem")
ap = argparse.ArgumentParser(); ap.add_argument("--law", default="datagen"); ap.add_argument("--B", type=int, default=1024); ap.add_argument("--seed0", type=int, default=100000)
ap.add_argument("--batches", type=int, default=16); ap.add_argument("--opts", default=""); ap.add_argument("--base", default=None); ap.add_argument("--N", type=int, default=20)
ap.add_argument("--max_iter", type=int, default=300); ap.add_argument("--save", default=""); ap.add_argument("--grid", default="reference")
a = ap.parse_args()
O = om.Oracle(a.N)
def parse(s):
    d = {}
    for kv in (s or "").split(","):
        if "=" in kv:
            k, v = kv.split("="); d[k] = float(v) if ("." in v or "e" in v) else int(v)
    return d
def makespan(it, order, slots=512):
    import heapq
    h = [0.0] * min(slots, len(it)); heapq.heapify(h)
    for m in order: heapq.heappush(h, heapq.heappop(h) + it[m])
    return max(h)
def run(opts):
    st, it, ms, t0 = [], [], [], time.time()
    for b in range(a.batches):
        consts = problem.production_constants(a.law) if a.grid == "reference" else None
        P, X0, _, _ = problem.make_batch(a.B, a.N, 0.6, seed=a.seed0 + b, consts=consts, dt_grid=a.grid, law=a.law)
        r = om.cpu_solve_batch(O, P, X0, threads=8, max_iter=a.max_iter, **opts)
        off = O.param_offsets()["q_init"] + 2
        order = np.argsort(-P[:, off], kind="stable")
        st.append(r["status"]); it.append(r["iters"]); ms.append(makespan(r["iters"], order))
    return dict(status=np.concatenate(st), iters=np.concatenate(it), ms=np.array(ms), t=time.time() - t0)
def summ(r):
    st, it = r["status"], r["iters"]
    return dict(converged=int((st == 0).sum()), cert=int((st == 3).sum()), stalled=int((st == 4).sum()), undecided=int(np.isin(st, (1, 2)).sum()), it_mean=round(float(it.mean()), 2),
                it_p99=float(np.percentile(it, 99)), it_p999=float(np.percentile(it, 99.9)), it_max=int(it.max()), makespan_mean=round(float(r["ms"].mean()), 1), makespan_max=float(r["ms"].max()), secs=round(r["t"], 1))
new = run(parse(a.opts)); print("new ", json.dumps(summ(new)), flush=True)
if a.save: np.savez(a.save, status=new["status"], iters=new["iters"])
if a.base is not None:
    old = run(parse(a.base)); print("base", json.dumps(summ(old)))
    ch = np.nonzero(new["status"] != old["status"])[0]
    print("%d status changes (member: base -> new, iters base -> new):" % len(ch), [(int(m), int(old["status"][m]), int(new["status"][m]), int(old["iters"][m]), int(new["iters"][m])) for m in ch][:80])
