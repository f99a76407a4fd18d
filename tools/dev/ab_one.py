"""dev probe, one library build per process (so that rocprofv3 can attribute counters): solves the bench batches (seeds 20211 + 1000 i) with the library at
argv[1] and prints one line: batch ms per seed, NLPs/s, iterations, phase timers under load.   python tools/dev/ab_one.py <lib.so|cur> [nseeds] [--noprof]"""
import importlib, sys, os, time, json, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
path = sys.argv[1]; nseed = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 4
N, B = 40, 1024
L = capi.LandingLib(N, 0, lib_path=None if path == "cur" else os.path.join(ROOT, path))
mk = lambda *s, dt=torch.float64: torch.empty(*s, device="cuda", dtype=dt)
x, st, it = mk(B, L.nx), mk(B, dt=torch.int32), mk(B, dt=torch.int32)
stream = torch.cuda.current_stream().cuda_stream
o = L.default_opts(); o.max_iter = 300
ms, conv, its = [], [], []
for i in range(nseed):
    P, X0, _, _ = problem.make_batch(B, N, 0.6, seed=20211 + 1000 * i)
    dP, dX0 = torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")
    ts = []
    for rep in range(2 if i else 3):
        torch.cuda.synchronize(); t = time.perf_counter()
        L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, stream)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    ms.append(min(ts[1:] if len(ts) > 2 else ts)); conv.append(int((st == 0).sum())); its.append(float(it.float().mean()))
out = {"lib": path, "ms": [round(v, 1) for v in ms], "nlps": round(nseed * B / sum(ms) * 1e3), "conv": conv, "iters": [round(v, 2) for v in its]}
if "--noprof" not in sys.argv:
    names = ("eval", "err", "cond", "back", "fwd", "dual", "ls", "accept")
    for label, nb in (("load", B), ("alone", 8)):
        prof = torch.zeros(nb, 16, device="cuda", dtype=torch.float64)
        L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
        L.solve_device(nb, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        L.lib.landing_set_profile_buffer(L.ctx, None)
        ph = prof.cpu().numpy()
        v = ph[:, :8].sum(axis=0) / 1e5 / ph[:, 10].sum()
        out[label] = {"total": round(float(v.sum()), 4), **{n: round(float(a), 4) for n, a in zip(names, v)}}
        out[label + "_sweeps_per_it"] = round(float(ph[:, 8].sum() / ph[:, 10].sum()), 3)
print(json.dumps(out), flush=True)
