"""SURVEY 8(f) row N2 / BASELINE configs[3]: the SQP (Gauss-Newton / iLQR) loop on the 18-DoF floating-base dynamics -- exact
linearisation, LQ backward pass (landing_wb_backward_kernel) and nonlinear rollouts (landing_wb_rollout_kernel) -- against the numpy
oracle of oracle/wb_oracle.py (6 x 6 Pluecker dynamics, Richardson-extrapolated derivatives, dense textbook LQ pass).
CPU: the kernels through tests/emu on a short horizon, iterate for iterate.  GPU: N = 40 steps of 1 ms, a batch of members: monotone cost,
agreement of one member with the oracle after one iteration.  Tolerances (fp64): 1e-6 relative on trajectories and costs (the two
sides differentiate the dynamics differently: forward mode vs extrapolated differences, 1e-8)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")
DT = 0.001        # explicit Euler on the leg links (joint-space inertias 5e-4 .. 6e-3 kg m^2 under 24 N foot forces) is unstable beyond ~1.5 ms
Q = np.concatenate([np.full(3, 200.0), np.full(3, 100.0), np.full(12, 20.0), np.full(18, 0.5)])
R = np.full(12, 1e-3)
QN = 5.0 * Q
_JI = np.tile([0.0064, 0.0056, 0.00049], 4)                     # joint-space inertias of the leg joints at the nominal pose (diag of H)
KPD = np.zeros((12, 36)); KPD[np.arange(12), 6 + np.arange(12)] = -0.05 * _JI / DT ** 2; KPD[np.arange(12), 24 + np.arange(12)] = -0.3 * _JI / DT     # joint PD law of the initial rollout (inside the stability region of the explicit Euler step)


def _problem(rng, B, N):
    """a crouching reference: the base sinks and pitches slightly, the knees bend; constant supporting foot forces"""
    x0 = np.zeros((B, 36)); xref = np.zeros((B, N + 1, 36)); f = np.zeros((B, N, 12)); u0 = np.zeros((B, N, 12))
    for b in range(B):
        q0 = np.concatenate([[0.0, 0.0, 0.30 + 0.02 * rng.normal()], 0.05 * rng.normal(size=3), np.tile([0.0, -0.8, 1.6], 4) + 0.05 * rng.normal(size=12)])
        x0[b, :18] = q0; x0[b, 18:] = 0.1 * rng.normal(size=18); x0[b, 20] -= 0.5
        for k in range(N + 1):
            s = k / N
            xref[b, k, :18] = q0; xref[b, k, 2] = q0[2] - 0.01 * s; xref[b, k, 4] = q0[4] + 0.02 * s
            xref[b, k, 6:18] = q0[6:] + np.tile([0.0, -0.03 * s, 0.06 * s], 4)
        f[b] = np.tile([0.0, 0.0, 8.252 * 9.81 / 4 * 1.2], 4)
        from oracle import rbd_oracle as ro
        _, Cq = ro.hand_c(q0, np.zeros(18), f[b, 0].reshape(4, 3))
        u0[b] = Cq[6:]                      # joint torques that hold the legs against the foot forces at the initial pose
    return x0, u0, xref, f


def _sqp(L, dev, N, **kw):
    import torch  # noqa: F401
    R_ = lc("rbd").Rbd(L)
    return lc("wb").WholeBodySQP(L, R_, N, DT, Q, R, QN, device=dev, **kw)


@pytest.mark.parametrize("semi,fused", [(False, True), (True, True), (False, False)])
def test_wb_sqp_emulated_follows_oracle(semi, fused):
    """explicit and semi-implicit Euler (landing_wb_set_integrator), step lengths chosen on the device (landing_wb_select) or by the round-3 host loop"""
    import torch
    from oracle import wb_oracle as wo
    wo.SEMI = semi
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    N, B, iters = 5, 2, 2
    x0, u0, xref, f = _problem(np.random.default_rng(3), B, N)
    S = _sqp(L, "cpu", N, semi_implicit=semi, fused=fused)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    out = S.solve(t(x0), t(u0), t(xref), t(f), iters=iters, K_init=KPD)
    cost = out["cost"].numpy()
    print(cost)
    for b in range(B):
        try:
            xs, us, hist = wo.solve(x0[b], u0[b], xref[b], f[b], DT, Q, R, QN, iters, K_init=KPD)
        finally:
            wo.SEMI = False if b == B - 1 else semi
        assert np.allclose(cost[:, b], hist, rtol=1e-6), (cost[:, b], hist)
        assert np.max(np.abs(out["x"][b].numpy() - xs)) <= 1e-6 * max(1.0, np.abs(xs).max())
        assert np.max(np.abs(out["u"][b].numpy() - us)) <= 1e-5 * max(1.0, np.abs(us).max())
    assert (np.diff(cost, axis=0) <= 1e-12).all() and (cost[-1] < 0.9 * cost[0]).all(), cost


@pytest.mark.gpu
def test_wb_sqp_gpu_n40():
    import torch
    from oracle import wb_oracle as wo
    L = lc("capi").LandingLib(40, device=0)
    N, B = 40, 64
    x0, u0, xref, f = _problem(np.random.default_rng(5), B, N)
    S = _sqp(L, "cuda", N)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda")
    out = S.solve(t(x0), t(u0), t(xref), t(f), iters=6, K_init=KPD)
    cost = out["cost"].cpu().numpy()
    assert np.isfinite(cost).all() and (np.diff(cost, axis=0) <= 1e-9 * cost[0]).all()
    assert (cost[-1] < 0.5 * cost[0]).all(), (cost[0][:4], cost[-1][:4])
    assert (cost[-1] - cost[-2] >= -0.05 * cost[-1]).mean() > 0.8          # the iteration has settled on most members
    # one member, one iteration, against the oracle (N = 40: 40 knots x 144 + dynamics evaluations in numpy)
    o1 = S.solve(t(x0[:1]), t(u0[:1]), t(xref[:1]), t(f[:1]), iters=1, K_init=KPD)
    xs, us, hist = wo.solve(x0[0], u0[0], xref[0], f[0], DT, Q, R, QN, 1, K_init=KPD)
    assert np.allclose(o1["cost"].cpu().numpy()[:, 0], hist, rtol=1e-6)
    assert np.max(np.abs(o1["x"][0].cpu().numpy() - xs)) <= 1e-6 * max(1.0, np.abs(xs).max())
    L.close()


@pytest.mark.gpu
def test_wb_sqp_configs3_full_size():
    """BASELINE configs[3] at its stated size: the SQP loop (exact linearisation of the 18-DoF floating-base dynamics at every knot +
    LQ backward pass + nonlinear rollouts) at N = 40, batch = 1024.  Every member's cost decreases monotonically and settles; the
    first iteration of a sample of members equals the oracle's (wb_oracle.py) to 1e-6 relative."""
    import torch
    from oracle import wb_oracle as wo
    L = lc("capi").LandingLib(40, device=0)
    N, B = 40, 1024
    x0, u0, xref, f = _problem(np.random.default_rng(5), 64, N)
    rng = np.random.default_rng(1)
    rep = B // 64
    x0 = np.tile(x0, (rep, 1)) + 1e-3 * rng.normal(size=(B, 36)); u0 = np.tile(u0, (rep, 1, 1)); xref = np.tile(xref, (rep, 1, 1)); f = np.tile(f, (rep, 1, 1))
    S = _sqp(L, "cuda", N)
    t = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda")
    out = S.solve(t(x0), t(u0), t(xref), t(f), iters=5, K_init=KPD)
    cost = out["cost"].cpu().numpy()
    assert cost.shape[1] == B and np.isfinite(cost).all()
    assert (np.diff(cost, axis=0) <= 1e-9 * cost[0]).all()
    assert (cost[-1] < 0.5 * cost[0]).all()
    assert (cost[-1] - cost[-2] >= -0.05 * cost[-1]).mean() > 0.8
    for b in (0, 517, 1023):
        o1 = S.solve(t(x0[b:b + 1]), t(u0[b:b + 1]), t(xref[b:b + 1]), t(f[b:b + 1]), iters=1, K_init=KPD)
        xs, us, hist = wo.solve(x0[b], u0[b], xref[b], f[b], DT, Q, R, QN, 1, K_init=KPD)
        assert np.allclose(o1["cost"].cpu().numpy()[:, 0], hist, rtol=1e-6)
        assert np.max(np.abs(o1["x"][0].cpu().numpy() - xs)) <= 1e-6 * max(1.0, np.abs(xs).max())
    L.close()
