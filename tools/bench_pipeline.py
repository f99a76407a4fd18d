#!/usr/bin/env python3
"""Throughput with several independent batches in flight (one context + stream each): what a data-generation job that
streams batches through one GPU gets, next to bench.py's one-batch-at-a-time figure (development / documentation tool)."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=1024); ap.add_argument("--N", type=int, default=40)
ap.add_argument("--inflight", type=int, default=2); ap.add_argument("--batches", type=int, default=8); ap.add_argument("--max_iter", type=int, default=300)
a = ap.parse_args()
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
dev = "cuda"
slots = []
for s in range(a.inflight):
    L = capi.LandingLib(a.N, 0)
    mk = lambda *sh, dt=torch.float64: torch.empty(*sh, device=dev, dtype=dt)
    slots.append(dict(L=L, stream=torch.cuda.Stream(), x=mk(a.B, L.nx), f=mk(a.B), lam=mk(a.B, L.ng), kkt=mk(a.B, 3),
                      st=mk(a.B, dt=torch.int32), it=mk(a.B, dt=torch.int32)))
inputs = []
for b in range(a.batches):
    P, X0, _, _ = problem.make_batch(a.B, a.N, 0.6, seed=20211 + b)
    inputs.append((torch.tensor(P, device=dev), torch.tensor(X0, device=dev)))
def launch(slot, inp):
    L = slot["L"]; o = L.default_opts(); o.max_iter = a.max_iter
    L.solve_device(a.B, inp[0].data_ptr(), inp[1].data_ptr(), o, slot["x"].data_ptr(), slot["f"].data_ptr(), slot["lam"].data_ptr(),
                   slot["st"].data_ptr(), slot["it"].data_ptr(), slot["kkt"].data_ptr(), slot["stream"].cuda_stream)
for s in slots: launch(s, inputs[0])          # warm-up (tables, workspaces)
torch.cuda.synchronize()
solved = 0
t0 = time.perf_counter()
for b in range(a.batches):
    s = slots[b % a.inflight]
    s["stream"].synchronize()                  # previous batch of this slot is done: count it, reuse the buffers
    if b >= a.inflight: solved += int((s["st"] == 0).sum().item())
    launch(s, inputs[b])
for i, s in enumerate(slots):
    s["stream"].synchronize(); solved += int((s["st"] == 0).sum().item())
dt = time.perf_counter() - t0
print(json.dumps({"B": a.B, "N": a.N, "inflight": a.inflight, "batches": a.batches, "sec": dt, "solved": solved, "of": a.B * a.batches, "nlp_per_s": solved / dt}))
