"""Development probe: A/B of library builds on ONE box: batch time over seeds + phase timers under load and alone.
   python tools/dev/variants.py name=path [name=path ...]   (path relative to the repo root; 'cur' = the product library)"""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
dev = "cuda"
mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
batches = {}
SEEDS = tuple(int(v) for v in os.environ.get('VAR_SEEDS', '20211,21211,22211,23211,24211,25211,26211,27211').split(','))
for seed in SEEDS:
    P, X0, _, _ = problem.make_batch(B, N, 0.6, seed=seed)
    batches[seed] = (torch.tensor(P, device=dev), torch.tensor(X0, device=dev))
names = ("eval", "err", "cond", "back", "fwd", "dual", "ls", "accept")
for spec in sys.argv[1:]:
    name, path = spec.split("=")
    L = capi.LandingLib(N, 0, lib_path=None if path == "cur" else os.path.join(ROOT, path))
    x, st, it = mk(B, L.nx), mk(B, dt=torch.int32), mk(B, dt=torch.int32)
    stream = torch.cuda.current_stream().cuda_stream
    o = L.default_opts(); o.max_iter = 300
    for kv in os.environ.get("VAR_OPTS", "").split(","):
        if "=" in kv:
            k_, v_ = kv.split("="); setattr(o, k_, type(getattr(o, k_))(float(v_)))
    tot = 0.0; line = []
    for seed, (dP, dX0) in batches.items():
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, stream)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        ms = 1e3 * min(ts[1:]); tot += ms
        line.append("%d: %.0f ms (%d, %.1f, %d)" % (seed, ms, int((st == 0).sum()), it.float().mean().item(), int(it.max())))
    # phase timers: under load (whole batch) and alone (8 members)
    res = {}
    for label, nb in (("load", B), ("alone", 8)):
        prof = torch.zeros(nb, 16, device=dev, dtype=torch.float64)
        L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr())
        dP, dX0 = batches[SEEDS[0]]
        L.solve_device(nb, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, stream)
        torch.cuda.synchronize()
        L.lib.landing_set_profile_buffer(L.ctx, None)
        ph = prof.cpu().numpy()
        res[label] = ph[:, :8].sum(axis=0) / 1e5 / ph[:, 10].sum()
    print("%-10s total %.1f ms -> %.0f NLPs/s | seed: ms (conv, mean it, max it) %s" % (name, tot, len(SEEDS) * B / tot * 1e3, "  ".join(line)))
    for label in ("load", "alone"):
        print("    %-5s ms/iter %.4f : %s" % (label, res[label].sum(), "  ".join("%s %.4f" % (n, v) for n, v in zip(names, res[label]))), flush=True)
    L.close()
