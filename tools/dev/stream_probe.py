"""Development probe (round 6): throughput of landing_stream_* under different consumption patterns, against two contexts on two streams.
   python tools/dev/stream_probe.py"""
import importlib, sys, os, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B, K = 40, 1024, 24
L = capi.LandingLib(N, 0); o = L.default_opts(); o.max_iter = 300
bat = [tuple(torch.tensor(a, device="cuda") for a in problem.make_batch(B, N, 0.6, seed=20211 + 1000 * i)[:2]) for i in range(8)]
mk = lambda *s, dt=torch.float64: torch.empty(*s, device="cuda", dtype=dt)
def run(label, lanes, mode):
    S = L.stream(lanes)
    outs = [(mk(B, L.nx), mk(B, dt=torch.int32)) for _ in range(lanes)]
    cur = torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    conv = torch.zeros(1, device="cuda", dtype=torch.float64)
    def sub(i, ins):
        x, st = outs[i % lanes]; dP, dX = bat[i % 8]
        return S.submit(B, dP.data_ptr(), dX.data_ptr(), o, x.data_ptr(), d_status=st.data_ptr(), in_stream=ins)
    for i in range(lanes): sub(i, cur)
    S.sync(); torch.cuda.synchronize()
    tk = []; t = time.perf_counter()
    for i in range(K):
        if mode == "nocount":
            tk.append(sub(i, cur))
        elif mode == "count_cur":
            if i >= lanes: S.wait(tk[i - lanes], stream=cur); conv += (outs[i % lanes][1] == 0).sum()
            tk.append(sub(i, cur))
        elif mode == "count_side":
            if i >= lanes:
                S.wait(tk[i - lanes], stream=side.cuda_stream)
                with torch.cuda.stream(side): conv += (outs[i % lanes][1] == 0).sum()
            tk.append(sub(i, side.cuda_stream))
        elif mode == "hostwait":
            if i >= lanes: S.wait(tk[i - lanes]); conv += (outs[i % lanes][1] == 0).sum()
            tk.append(sub(i, cur))
    S.sync(); torch.cuda.synchronize(); t = time.perf_counter() - t
    print("%-28s lanes %d: %.1f ms per batch, %.0f NLPs/s" % (label + " " + mode, lanes, 1e3 * t / K, K * B / t), flush=True)
    S.close()
for lanes in (1, 2, 3):
    for mode in ("nocount", "count_cur", "count_side", "hostwait"):
        run("stream", lanes, mode)
