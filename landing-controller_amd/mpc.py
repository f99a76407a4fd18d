"""Receding-horizon landing controller loop on the batched solver (BASELINE configs[4]).

The reference only has the building block: a warm-started re-solve from the stored previous solution
(codegen_casadi/test_loadCasadi_ws.m:73-88, options of generate_landingCtrller_IPOPT_warmstart.m:246-247,263).  A control
loop repeats it at the controller rate: every tick the previous solution is advanced by one stage, the measured state
becomes the initial condition, and the NLP is re-solved from that guess.  Everything stays in HBM between ticks; per tick
the host enqueues one shift kernel and one solver launch per batch.

    ctl = RecedingHorizon(lib, P, X0)          # cold solve of the B drop states (device tensors inside)
    for t in range(ticks):
        u0 = ctl.first_controls()              # [B, 24] feet and forces applied during the tick
        x_meas = plant(ctl.predicted_next_state(), ...)   # the caller's plant / state estimator, [B, 12] on the device
        info = ctl.tick(x_meas)                # shift + warm-started solve; info: iterations, status, kkt
"""
import torch


class RecedingHorizon:
    def __init__(self, lib, P, X0, opts_cold=None, opts_warm=None, device="cuda"):
        self.lib, self.N, self.B = lib, lib.N, P.shape[0]
        dev = torch.device(device)
        f64 = dict(device=dev, dtype=torch.float64)
        self.p = torch.as_tensor(P, **f64).contiguous().clone()
        self.x0 = torch.as_tensor(X0, **f64).contiguous().clone()
        self.x = torch.empty_like(self.x0)
        self.f = torch.empty(self.B, **f64); self.kkt = torch.empty(self.B, 3, **f64)
        self.status = torch.empty(self.B, device=dev, dtype=torch.int32); self.iters = torch.empty(self.B, device=dev, dtype=torch.int32)
        self.opts_cold = opts_cold or lib.default_opts()
        self.opts_warm = opts_warm or lib.warm_opts()
        self.stream = torch.cuda.current_stream().cuda_stream
        self._solve(self.opts_cold)

    def _solve(self, opts):
        self.lib.solve_device(self.B, self.p.data_ptr(), self.x0.data_ptr(), opts, self.x.data_ptr(), self.f.data_ptr(), 0, self.status.data_ptr(),
                              self.iters.data_ptr(), self.kkt.data_ptr(), self.stream)

    def first_controls(self):
        """U(:,0) of the current plan: foot positions (12) and ground-reaction forces (12) per member"""
        nX = 12 * (self.N + 1)
        return self.x[:, nX:nX + 24]

    def predicted_next_state(self):
        """X(:,1) of the current plan: where the NLP's own discretisation puts the body after one stage"""
        return self.x[:, 12:24]

    def tick(self, state):
        """state [B, 12] = measured [q; qd] (device tensor).  Shifts the plan, re-solves warm; returns the device tensors."""
        state = state.contiguous()
        self.lib.mpc_shift_device(self.B, self.x.data_ptr(), state.data_ptr(), self.p.data_ptr(), self.x0.data_ptr(), self.stream)
        self._solve(self.opts_warm)
        return dict(status=self.status, iters=self.iters, kkt=self.kkt, f=self.f)
