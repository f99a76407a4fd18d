"""SURVEY 8(f) row N3, tracking-gain synthesis: SRBM variational linearisation A, B and the Riccati differential equation
(generateVariationalDynamics.m:29-62, generateRiccatiIntegrator.m:24-62, quadruped_SRBM_NLP.m:428-503).
The HIP kernel (fp64 matrix cores) against the numpy oracle that restates the reference's formulas:
  * CPU: the same kernel sources compiled for the host through tests/emu;
  * GPU (-m gpu): the product library, batch of trajectories sampled from solved landing NLPs.
Tolerance: fp64, 1e-10 relative on A, B; 1e-9 relative on P and K after the whole backward sweep (products are
accumulated in a different order on the matrix cores)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")


def _case(rng, n, B):
    xref = np.zeros((B, n, 24)); fref = np.zeros((B, n, 12))
    for b in range(B):
        base = np.concatenate([[0, 0, 0.35], 0.4 * rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)])
        feet = np.array([0.2, -0.15, 0, 0.2, 0.15, 0, -0.2, -0.15, 0, -0.2, 0.15, 0.0])
        for j in range(n):
            xref[b, j, :12] = base + 0.05 * j * rng.normal(size=12)
            xref[b, j, 12:] = feet + 0.02 * rng.normal(size=12)
            fref[b, j] = np.tile([3.0, -2.0, 25.0], 4) + 5 * rng.normal(size=12)
    return xref, fref


def _check(P, K, A, Bm, xref, fref, Ib, mass, Q, R, F, dt, rk4, vo):
    for b in range(xref.shape[0]):
        Po, Ko = vo.rde_backward(xref[b], fref[b], Ib, mass, Q, R, F, dt, rk4=rk4)
        for j in range(xref.shape[1]):
            Ao, Bo = vo.vbl_AB(xref[b, j], fref[b, j], Ib, mass)
            assert np.max(np.abs(A[b, j] - Ao)) <= 1e-10 * max(1.0, np.max(np.abs(Ao)))
            assert np.max(np.abs(Bm[b, j] - Bo)) <= 1e-10 * max(1.0, np.max(np.abs(Bo)))
        assert np.max(np.abs(P[b] - Po)) <= 1e-9 * max(1.0, np.max(np.abs(Po))), np.max(np.abs(P[b] - Po))
        assert np.max(np.abs(K[b] - Ko)) <= 1e-9 * max(1.0, np.max(np.abs(Ko)))
        assert np.max(np.abs(P[b] - np.swapaxes(P[b], 1, 2))) <= 1e-10 * np.max(np.abs(P[b]))     # P stays symmetric


@pytest.mark.parametrize("rk4", [False, True])
def test_vbl_emulated_kernel_matches_oracle(rk4):
    from oracle import vbl_oracle as vo
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    rng = np.random.default_rng(3)
    B, n = 2, 7
    xref, fref = _case(rng, n, B)
    Ib = lc("constants").composite_body_inertia()[0:3, 0:3]; mass = 8.252
    F, Q, R = vo.reference_weights()
    P = np.zeros((B, n, 24, 24)); K = np.zeros((B, n, 12, 24)); A = np.zeros((B, n, 24, 24)); Bm = np.zeros((B, n, 24, 12))
    ptr = lambda a: a.ctypes.data          # the emulation's "device" memory is host memory
    L.riccati_gains_device(B, n, ptr(xref), ptr(fref), Ib, mass, Q, np.diag(R), F, 0.022, rk4, ptr(P), ptr(K), ptr(A), ptr(Bm))
    _check(P, K, A, Bm, xref, fref, Ib, mass, Q, R, F, 0.022, rk4, vo)


def test_sample_reference_grid():
    """interpolation of a solved trajectory onto the Riccati grid (quadruped_SRBM_NLP.m:487-499)"""
    from oracle import vbl_oracle as vo
    N = 6
    X = np.arange(12 * (N + 1), dtype=float).reshape(12, N + 1, order="F"); U = np.arange(24 * N, dtype=float).reshape(24, N, order="F")
    t = np.linspace(0, 0.6, N + 1)
    xd, ud = vo.sample_reference(X, U, t, 0.05, 13)
    assert np.allclose(xd[0, :12], X[:, 0]) and np.allclose(xd[2, :12], X[:, 1]) and np.allclose(xd[1, :12], 0.5 * (X[:, 0] + X[:, 1]))
    assert np.allclose(ud[3], U[12:, 1])


@pytest.mark.gpu
def test_vbl_gpu_along_solved_trajectories():
    import torch
    from oracle import vbl_oracle as vo
    capi, Pm = lc("capi"), lc("problem")
    N, B = 40, 6
    L = capi.LandingLib(N, device=0)
    Pb, X0, _, _ = Pm.make_batch(B, N, 0.6, seed=4)
    r = L.solve_host(Pb, X0)
    assert (r["status"] == 0).all()
    dt_r, n = 0.022, int(0.6 / 0.022) + 1
    xs, fs = [], []
    for b in range(B):
        Xs, Us = Pm.split_solution(N, r["x"][b])
        xd, ud = vo.sample_reference(Xs, Us, np.linspace(0, 0.6, N + 1), dt_r, n)
        xs.append(xd); fs.append(ud)
    xref, fref = np.array(xs), np.array(fs)
    Ib = lc("constants").composite_body_inertia()[0:3, 0:3]; mass = 8.252
    F, Q, R = vo.reference_weights()
    dev = "cuda"
    dx, df = torch.tensor(xref, device=dev), torch.tensor(fref, device=dev)
    mk = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float64)
    for rk4 in (False, True):
        P, K, A, Bm = mk(B, n, 24, 24), mk(B, n, 12, 24), mk(B, n, 24, 24), mk(B, n, 24, 12)
        L.riccati_gains_device(B, n, dx.data_ptr(), df.data_ptr(), Ib, mass, Q, np.diag(R), F, dt_r, rk4, P.data_ptr(), K.data_ptr(), A.data_ptr(), Bm.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        _check(P.cpu().numpy(), K.cpu().numpy(), A.cpu().numpy(), Bm.cpu().numpy(), xref, fref, Ib, mass, Q, R, F, dt_r, rk4, vo)
    # closed-loop sanity of the gains: A - B K at the first grid point is Hurwitz on the body block for the RK4 solution
    Acl = A.cpu().numpy()[0, 0] - Bm.cpu().numpy()[0, 0] @ K.cpu().numpy()[0, 0]
    assert np.isfinite(Acl).all()
    L.close()
