"""SQP (Gauss-Newton / iLQR) loop on the 18-DoF floating-base model (BASELINE configs[3]).

Host side of include/landing_nlp.h's landing_wb_* entry points: per iteration one exact linearisation of the dynamics at every knot
(landing_fb_dynamics_batch, fd_h = 0), one LQ backward pass (landing_wb_backward) and one set of nonlinear rollouts
(landing_wb_rollout) per step length of a backtracking search; a member keeps the first step length that lowers its cost.  Tensors live where `device` says ("cuda" for the product library,
"cpu" for the host emulation of tests/emu -- the C ABI only sees pointers).

    sqp = WholeBodySQP(lib, rbd, N=40, dt=0.015, Q=..., R=..., QN=...)
    out = sqp.solve(x0, u_init, xref, f_foot, iters=5)        # x [B, N+1, 36], u [B, N, 12], cost history [iters+1, B]
"""
import ctypes as C

import numpy as np
import torch


class WholeBodySQP:
    def __init__(self, lib, rbd, N, dt, Q, R, QN, device="cuda", alphas=(1.0, 0.5, 0.25, 0.1, 0.03), semi_implicit=False, fused=True):
        """semi_implicit: qd+ = qd + dt qdd, q+ = q + dt qd+ (landing_wb_set_integrator; the scheme of test_scripts/test_integrationDifference.m:30-40)
        instead of explicit Euler.  fused: ALL step lengths in one landing_wb_rollout launch and the choice per member on the device
        (landing_wb_select) -- no host synchronisation inside an iteration; False = the round-3 loop (one launch + torch.where merges per step length)"""
        self.L, self.R_, self.N, self.dt, self.dev = lib, rbd, int(N), float(dt), torch.device(device)
        self.fused = bool(fused)
        lib.lib.landing_wb_set_integrator.argtypes = [C.c_void_p, C.c_int]
        lib._check(lib.lib.landing_wb_set_integrator(lib.ctx, 1 if semi_implicit else 0), "landing_wb_set_integrator")
        lib.lib.landing_wb_select.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10
        lib.lib.landing_wb_skip_taken.argtypes = [C.c_void_p, C.c_void_p]
        self.Q = np.ascontiguousarray(Q, float); self.R = np.ascontiguousarray(R, float); self.QN = np.ascontiguousarray(QN, float)
        assert self.Q.shape == (36,) and self.R.shape == (12,) and self.QN.shape == (36,)
        self.alphas = torch.tensor(list(alphas), dtype=torch.float64, device=self.dev)
        vp, dp = C.c_void_p, C.POINTER(C.c_double)
        lib.lib.landing_wb_backward.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp, vp, vp, vp, dp, dp, dp, vp, vp, vp, vp, vp]
        lib.lib.landing_wb_rollout.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_double, vp, vp, vp, vp, vp, vp, dp, dp, dp, vp, vp, vp, vp]
        self._w = [a.ctypes.data_as(dp) for a in (self.Q, self.R, self.QN)]

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream if self.dev.type == "cuda" else None

    def _mk(self, *s, dt=torch.float64):
        return torch.zeros(*s, device=self.dev, dtype=dt)

    def rollout(self, x, u, xref, f_foot, K=None, kff=None, alphas=None):
        B = x.shape[0]; al = self.alphas[:1] if K is None else (self.alphas if alphas is None else alphas)
        na = al.shape[0]
        xn, un, cost = self._mk(na, B, self.N + 1, 36), self._mk(na, B, self.N, 12), self._mk(na, B)
        p = lambda t: t.data_ptr() if t is not None else None
        self.L._check(self.L.lib.landing_wb_rollout(self.L.ctx, B, self.N, na, p(al), self.dt, p(x), p(u), p(xref), p(f_foot), p(K), p(kff),
                                                    self._w[0], self._w[1], self._w[2], p(xn), p(un), p(cost), self._stream()), "landing_wb_rollout")
        return xn, un, cost

    def linearise(self, x, u, f_foot):
        B = x.shape[0]; n = B * self.N
        q = x[:, :self.N, :18].reshape(n, 18).contiguous(); qd = x[:, :self.N, 18:].reshape(n, 18).contiguous()
        tau = torch.cat([self._mk(n, 6), u.reshape(n, 12)], dim=1).contiguous()
        ff = f_foot.reshape(n, 12).contiguous() if f_foot is not None else None
        A, Hinv = self._mk(n, 18, 36), self._mk(n, 18, 18)
        self.R_.fb_dynamics(n, q.data_ptr(), qd.data_ptr(), tau.data_ptr(), ff.data_ptr() if ff is not None else 0, d_A=A.data_ptr(), d_Hinv=Hinv.data_ptr(),
                            fd_h=0.0, stream=self._stream() or 0)
        return A, Hinv

    def backward(self, x, u, xref, A, Hinv, reg=0.0):
        B = x.shape[0]
        K, kff, dV, ok = self._mk(B, self.N, 12, 36), self._mk(B, self.N, 12), self._mk(B, 2), self._mk(B, dt=torch.int32)
        self.L._check(self.L.lib.landing_wb_backward(self.L.ctx, B, self.N, self.dt, reg, x.data_ptr(), u.data_ptr(), xref.data_ptr(), A.data_ptr(), Hinv.data_ptr(),
                                                     self._w[0], self._w[1], self._w[2], K.data_ptr(), kff.data_ptr(), dV.data_ptr(), ok.data_ptr(), self._stream()),
                      "landing_wb_backward")
        return K, kff, dV, ok

    def solve(self, x0, u_init, xref, f_foot=None, iters=5, reg=0.0, K_init=None, rel_tol=0.0):
        """x0 [B, 36], u_init [B, N, 12], xref [B, N+1, 36], f_foot [B, N, 12] or None (tensors on self.dev).  K_init [12, 36]: feedback
        gain of the initial rollout, u = u_init + K_init (x - xref) -- e.g. a joint PD law; an open-loop rollout of constant torques
        over the whole horizon does not stay near the reference"""
        B = x0.shape[0]
        u = u_init.contiguous().clone(); xref = xref.contiguous(); f_foot = f_foot.contiguous() if f_foot is not None else None
        if K_init is None:
            xs = self._mk(B, self.N + 1, 36); xs[:, 0] = x0
            xn, un, c0 = self.rollout(xs, u, xref, f_foot)
        else:
            nom = xref.clone(); nom[:, 0] = x0
            K0 = torch.as_tensor(K_init, dtype=torch.float64, device=self.dev).reshape(1, 1, 12, 36).expand(B, self.N, 12, 36).contiguous()
            xn, un, c0 = self.rollout(nom.contiguous(), u, xref, f_foot, K0, self._mk(B, self.N, 12), alphas=self._mk(1))
        x, u, cost = xn[0].contiguous(), un[0].contiguous(), c0[0].clone()
        hist, steps = [cost.clone()], []
        for _ in range(iters):
            A, Hinv = self.linearise(x, u, f_foot)
            K, kff, dV, ok = self.backward(x, u, xref, A, Hinv, reg)
            # backtracking: the step lengths in decreasing order, one rollout launch each, a member keeps the FIRST one that lowers its
            # cost; the loop ends as soon as every member has one (normally after alpha = 1)
            if self.fused:
                # two stages, no host synchronisation: the full step for everybody, then the shorter ones for the members that did not take it
                # (the second rollout returns at once for all others -- normally for every member)
                step = torch.zeros_like(cost)
                x_base, u_base = x.clone(), u.clone()      # both stages roll out from the trajectory of the backward pass
                x = x.contiguous(); u = u.contiguous(); cost = cost.contiguous()
                na = int(self.alphas.shape[0])
                for stage, (lo, hi) in enumerate(((0, 1), (1, na))):
                    if hi <= lo:
                        continue
                    if stage == 1:
                        self.L._check(self.L.lib.landing_wb_skip_taken(self.L.ctx, step.data_ptr()), "landing_wb_skip_taken")
                    al = self.alphas[lo:hi].contiguous()
                    xn, un, cn = self.rollout(x_base, u_base, xref, f_foot, K, kff, alphas=al)
                    self.L._check(self.L.lib.landing_wb_select(self.L.ctx, B, self.N, (hi - lo) * (-1 if stage else 1), al.data_ptr(), ok.data_ptr(), xn.data_ptr(), un.data_ptr(),
                                                               cn.data_ptr(), x.data_ptr(), u.data_ptr(), cost.data_ptr(), step.data_ptr(), self._stream()), "landing_wb_select")
                hist.append(cost.clone()); steps.append(step)
                if rel_tol > 0.0 and float(((hist[-2] - hist[-1]) / hist[-1].clamp_min(1e-300)).max()) <= rel_tol:
                    break
                continue
            x_old, u_old = x, u
            done = ~ok.bool(); step = torch.zeros_like(cost)
            for ia in range(self.alphas.shape[0]):
                xn, un, cn = self.rollout(x_old, u_old, xref, f_foot, K, kff, alphas=self.alphas[ia:ia + 1])
                acc = (~done) & (cn[0] < cost)
                x = torch.where(acc[:, None, None], xn[0], x).contiguous(); u = torch.where(acc[:, None, None], un[0], u).contiguous()
                cost = torch.where(acc, cn[0], cost); step = torch.where(acc, self.alphas[ia], step)
                done = done | acc
                if bool(done.all()):
                    break
            hist.append(cost.clone()); steps.append(step)
            if float(((hist[-2] - hist[-1]) / hist[-1].clamp_min(1e-300)).max()) <= rel_tol:      # no member improved any more
                break
        return dict(x=x, u=u, cost=torch.stack(hist), alpha=torch.stack(steps) if steps else None, expected=dV)
