"""Lagrangian Hessian / gradients of the running-cost formulation (generate_quadruped_SRBM_CCC.m:81-99) in the function layer:
landing_eval_hess_rc_batch (extended pattern: casadi_s4 + 18 N diagonals) and the running-cost parts of grad_gamma_x /
grad_gamma_p, against the oracle -- whose own entries are pinned by finite differences of its gradient here."""
import importlib
import os

import numpy as np
import pytest

from oracle.oracle import Oracle

PKG = "landing-controller_amd"
capi = importlib.import_module(PKG + ".capi")
problem = importlib.import_module(PKG + ".problem")
RC = dict(QX=[0, 0, 10, 1, 1, 0, .1, .1, .1, .1, .1, .1], Qc=[1.0, 1.0, 0.5], Qf=[1e-4, 1e-4, 1e-3], f_ref=[0, 0, 20.0])
EMU = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "liblanding_emu.so")


def _inputs(N, B, seed):
    P, X0, _, _ = problem.make_batch(B, N, 0.6, seed=seed)
    rng = np.random.default_rng(seed)
    return X0 + 0.05 * rng.normal(size=X0.shape), P, rng.normal(size=(B, 104 * N + 12)), 0.5 + rng.random(B)


def _dense(ci, r, h, n):
    H = np.zeros((n, n))
    for c in range(n):
        H[r[ci[c]:ci[c + 1]], c] = h[ci[c]:ci[c + 1]]
    return H + np.triu(H, 1).T


def test_oracle_hess_rc_matches_finite_differences():
    N = 5; O = Oracle(N, run_cost=RC)
    X, P, lam, lf = _inputs(N, 1, 3)
    x, p, lam, lf = X[0], P[0], lam[0], lf[0]
    ci, r = O.pattern_hess_rc()
    assert len(r) == O.nnz_hess + 18 * N and ci[-1] == len(r)
    for c in range(O.nx):      # upper triangular, strictly increasing rows
        rows = r[ci[c]:ci[c + 1]]
        assert np.all(np.diff(rows) > 0) and (len(rows) == 0 or rows[-1] <= c)
    H = _dense(ci, r, O.hess_l_rc(x, p, lf, lam), O.nx)
    e = 1e-6; Hfd = np.zeros_like(H)
    for i in range(O.nx):
        d = np.zeros(O.nx); d[i] = e
        Hfd[:, i] = (O.grad(x + d, p, lf, lam)[2] - O.grad(x - d, p, lf, lam)[2]) / (2 * e)
    assert np.abs(H - Hfd).max() < 1e-6 * max(1.0, np.abs(H).max())
    gp = O.grad(x, p, lf, lam)[3]

    def gamma(pp):
        f, g, _, _ = O.grad(x, pp, lf, lam)
        return lf * f + lam @ g
    for i in list(range(0, 12 * N, 7)) + list(range(12 * (N + 1), 12 * (N + 1) + N)):     # Xref_k and dt_k entries
        d = np.zeros(len(p)); d[i] = e
        assert abs((gamma(p + d) - gamma(p - d)) / (2 * e) - gp[i]) < 1e-6 * max(1.0, abs(gp[i]))


def _check(lib, N, B, seed):
    O = Oracle(N, run_cost=RC)
    X, P, lam, lf = _inputs(N, B, seed)
    ci, r = lib.pattern_hess_rc(); co, ro = O.pattern_hess_rc()
    assert np.array_equal(ci, co) and np.array_equal(r, ro)
    h = lib.hess_rc_host(X, P, lf, lam)
    out = lib.eval_host(X, P, lf, lam, want=("f", "grad_f", "grad_gamma_x", "grad_gamma_p"))
    for b in range(B):
        ho = O.hess_l_rc(X[b], P[b], lf[b], lam[b])
        assert np.max(np.abs(h[b] - ho)) <= 1e-12 * max(1.0, np.abs(ho).max())
        f, g, gx, gp = O.grad(X[b], P[b], lf[b], lam[b])
        assert abs(out["f"][b] - f) <= 1e-12 * max(1.0, abs(f))
        assert np.max(np.abs(out["grad_gamma_x"][b] - gx)) <= 1e-11 * max(1.0, np.abs(gx).max())
        assert np.max(np.abs(out["grad_gamma_p"][b] - gp)) <= 1e-11 * max(1.0, np.abs(gp).max())
    with pytest.raises(RuntimeError):        # the casadi_s4-pattern Hessian cannot hold the running cost
        lib.eval_host(X, P, lf, lam, want=("hess",))


@pytest.mark.skipif(not os.path.exists(EMU), reason="host emulation library not built (make -C landing-controller_amd/csrc emu)")
def test_hess_rc_emulated_matches_oracle():
    lib = capi.LandingLib(6, device=0, lib_path=EMU, run_cost=RC)
    _check(lib, 6, 3, 5)
    # without a running cost the extended pattern carries the casadi_s4 values and zeros
    lib0 = capi.LandingLib(6, device=0, lib_path=EMU); O0 = Oracle(6)
    X, P, lam, lf = _inputs(6, 2, 9)
    h = lib0.hess_rc_host(X, P, lf, lam)
    for b in range(2):
        assert np.max(np.abs(h[b] - O0.hess_l_rc(X[b], P[b], lf[b], lam[b]))) < 1e-11


@pytest.mark.gpu
def test_hess_rc_gpu_matches_oracle():
    lib = capi.LandingLib(40, device=0, run_cost=RC)
    _check(lib, 40, 16, 7)
