"""Streaming batches through one GPU with several of them in flight (the data-generation use case of
generate_data/generate_training_data_automated.m:38: thousands of independent drop states, produced batch by batch).

The time of ONE batch of 1024 NLPs is set by its slowest member and by the granularity of two members per resident slot
(DESIGN.md 4.2); a second batch in flight fills the CUs that idle in its tail: 24-25 k instead of 19 k NLPs/s on one MI355X.
Rounds 3-5 did this here, with one solver context and one torch stream per lane; since round 6 the library does it itself
(landing_stream_* of include/landing_nlp.h: ONE context, the lanes and their HIP streams live behind landing_stream_submit / _wait)
and this class is the host-array convenience around it: uploads on a copy stream, the submission ordered behind them.

    pipe = BatchPipeline(N=40, depth=2)
    for P, X0 in batches:                      # numpy [B, np], [B, nx]
        done = pipe.submit(P, X0)              # returns the results of the batch that left the pipeline, or None
    for res in pipe.drain(): ...
"""
import importlib
from collections import deque

import numpy as np
import torch


class BatchPipeline:
    def __init__(self, N, depth=2, device=0, opts=None, **form):
        capi = importlib.import_module(__package__ + ".capi")
        self.N, self.depth, self.dev = N, depth, torch.device("cuda", device)
        self.lib = capi.LandingLib(N, device=device, **form)
        self.S = self.lib.stream(depth)
        self.copy = torch.cuda.Stream(device=self.dev)
        self.opts = opts or self.lib.default_opts()
        self.inflight = deque()
        self.n_submitted = 0

    def _collect(self):
        ticket, tag, b = self.inflight.popleft()
        self.S.wait(ticket)                    # host waits for that submission only; the later ones keep the GPU busy
        return dict(tag=tag, x=b["x"].cpu().numpy(), f=b["f"].cpu().numpy(), status=b["st"].cpu().numpy(), iters=b["it"].cpu().numpy(), kkt=b["kkt"].cpu().numpy())

    def submit(self, P, X0, tag=None):
        out = self._collect() if len(self.inflight) >= self.depth else None
        lib, B = self.lib, P.shape[0]
        f64 = dict(device=self.dev, dtype=torch.float64)
        with torch.cuda.stream(self.copy):
            b = dict(p=torch.as_tensor(np.ascontiguousarray(P), **f64), x0=torch.as_tensor(np.ascontiguousarray(X0), **f64),
                     x=torch.empty(B, lib.nx, **f64), f=torch.empty(B, **f64), kkt=torch.empty(B, 3, **f64),
                     st=torch.empty(B, device=self.dev, dtype=torch.int32), it=torch.empty(B, device=self.dev, dtype=torch.int32))
        ticket = self.S.submit(B, b["p"].data_ptr(), b["x0"].data_ptr(), self.opts, b["x"].data_ptr(), b["f"].data_ptr(), 0, b["st"].data_ptr(), b["it"].data_ptr(),
                               b["kkt"].data_ptr(), in_stream=self.copy.cuda_stream)
        self.inflight.append((ticket, self.n_submitted if tag is None else tag, b))
        self.n_submitted += 1
        return out

    def drain(self):
        outs = []
        while self.inflight:
            outs.append(self._collect())
        return outs

    def close(self):
        self.S.sync(); self.S.close(); self.lib.close()
