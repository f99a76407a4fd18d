"""Streaming batches through one GPU with several of them in flight (the data-generation use case of
generate_data/generate_training_data_automated.m:38: thousands of independent drop states).

The time of ONE batch of 1024 NLPs is set by its slowest member and by the granularity of two members per resident slot
(DESIGN.md 4.2); a second batch in flight fills the CUs that idle in its tail: 24.4 k instead of 18.7 k NLPs/s on one MI355X
(round 5, profiles/r05_bench.json `two_batches_in_flight`).  One solver
context (workspace, tables) and one HIP stream per lane; a lane is reused as soon as its previous batch has been read.

    pipe = BatchPipeline(N=40, depth=2)
    for P, X0 in batches:                      # numpy [B, np], [B, nx]
        done = pipe.submit(P, X0)              # returns the results of the batch that left the pipeline, or None
    for res in pipe.drain(): ...
"""
import importlib

import numpy as np
import torch


class BatchPipeline:
    def __init__(self, N, depth=2, device=0, opts=None):
        capi = importlib.import_module(__package__ + ".capi")
        self.N, self.depth, self.dev = N, depth, torch.device("cuda", device)
        self.lanes = []
        for _ in range(depth):
            lib = capi.LandingLib(N, device=device)
            self.lanes.append(dict(lib=lib, stream=torch.cuda.Stream(device=self.dev), busy=False, bufs=None, tag=None))
        self.opts = opts or self.lanes[0]["lib"].default_opts()
        self.next_lane = 0
        self.n_submitted = 0

    def _collect(self, lane):
        lane["stream"].synchronize()
        b = lane["bufs"]
        lane["busy"] = False
        return dict(tag=lane["tag"], x=b["x"].cpu().numpy(), f=b["f"].cpu().numpy(), status=b["st"].cpu().numpy(), iters=b["it"].cpu().numpy(), kkt=b["kkt"].cpu().numpy())

    def submit(self, P, X0, tag=None):
        lane = self.lanes[self.next_lane]
        self.next_lane = (self.next_lane + 1) % self.depth
        out = self._collect(lane) if lane["busy"] else None
        lib, B = lane["lib"], P.shape[0]
        with torch.cuda.stream(lane["stream"]):
            f64 = dict(device=self.dev, dtype=torch.float64)
            b = dict(p=torch.as_tensor(np.ascontiguousarray(P), **f64), x0=torch.as_tensor(np.ascontiguousarray(X0), **f64),
                     x=torch.empty(B, lib.nx, **f64), f=torch.empty(B, **f64), kkt=torch.empty(B, 3, **f64),
                     st=torch.empty(B, device=self.dev, dtype=torch.int32), it=torch.empty(B, device=self.dev, dtype=torch.int32))
            lib.solve_device(B, b["p"].data_ptr(), b["x0"].data_ptr(), self.opts, b["x"].data_ptr(), b["f"].data_ptr(), 0, b["st"].data_ptr(), b["it"].data_ptr(),
                             b["kkt"].data_ptr(), lane["stream"].cuda_stream)
        lane.update(busy=True, bufs=b, tag=self.n_submitted if tag is None else tag)
        self.n_submitted += 1
        return out

    def drain(self):
        outs = []
        for i in range(self.depth):
            lane = self.lanes[(self.next_lane + i) % self.depth]
            if lane["busy"]:
                outs.append(self._collect(lane))
        return outs

    def close(self):
        for lane in self.lanes:
            lane["lib"].close()
