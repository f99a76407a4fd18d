"""SURVEY 8(f) row N1 -- the rows of the reference's kinodynamic refinement NLP (landing_optimization.m:100-189), pinned by
reference-held data: the two kinodynamic solutions stored beside the reference's test scripts (tests/golden/n1_kinodyn_solutions.npz,
written by tests/make_golden_n1.py; both on the uniform grid dt = 0.03, N = 20) satisfy every row group of oracle/kinodyn_oracle.py to
the tolerance of the solver that produced them -- Euler defects under rpyToRotMat_xyz / Binv, contact / LCP / no-slip rows, friction
pyramid, the forward-kinematics band |c - FK([q; jpos])| (ACTIVE in both: 1e-3 in the older file, 1e-2 in the newer one, the value of
landing_optimization.m:186-187), leg torques J_f'(-R' f) within tauMax, joint limits, kinematic box.  A wrong rotation convention, leg
geometry, side sign or inertia shows up here as a violation orders of magnitude above these tolerances (e.g. the production grid
instead of dt = 0.03: defects of 1.1).  CPU: oracle + the kernel through tests/emu; GPU: landing_kinodyn_rows_batch on the same data."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")
DT = np.full(20, 0.03)


def _sol(tag):
    d = np.load(os.path.join(GOLDEN, "n1_kinodyn_solutions.npz"))
    return d["X_" + tag], d["U_" + tag], d["J_" + tag]


@pytest.mark.parametrize("tag,mu,band", [("a", 0.5, 1e-3), ("b", 1.0, 1e-2)])
def test_stored_kinodynamic_solutions_satisfy_the_oracle_rows(tag, mu, band):
    from oracle import kinodyn_oracle as ko
    mass, Ib, Ibi = lc("constants").robot_constants()
    X, U, J = _sol(tag)
    assert X.shape == (12, 21) and U.shape == (24, 20) and J.shape == (12, 20)
    dd = ko.dynamics_defects(X, U, DT, mass, np.asarray(Ib), np.asarray(Ibi))
    assert np.abs(dd).max() <= 5e-5, np.abs(dd).reshape(4, 3, -1).max(axis=(1, 2))          # measured: 2.3e-5 (omega rows) / 3e-7
    fz, cz, lcp, slip = ko.contact_rows(U)
    assert fz.min() >= -1e-6 and cz.min() >= -1e-5 and lcp.max() <= 1e-3 * 1.001 and np.abs(slip).max() <= 1e-3 * 1.001
    assert lcp.max() >= 0.8e-3                                                               # (the complementarity row is active)
    assert ko.friction_rows(U, mu).min() >= -1e-5                                            # friction pyramid at the file's mu (a: active)
    fk_err, tau = ko.kinematic_rows(X, U, J)
    assert 0.98 * band <= np.abs(fk_err).max() <= band * 1.002                               # the FK band is ACTIVE at its bound
    assert (np.abs(tau).reshape(-1, 4, 3).max(axis=(0, 1)) <= ko.TAU_MAX).all()
    assert (J.T >= ko.JPOS_MIN - 1e-6).all() and (J.T <= ko.JPOS_MAX + 1e-6).all()
    pr = ko.hip_relative(X, U)
    assert pr[..., 2].max() <= -0.075 + 1e-3 and pr[..., 2].min() >= -0.4 - 1e-3 and np.linalg.norm(pr, axis=-1).max() <= 0.4 + 1e-3
    assert X[2].min() >= 0.075


def _rows_through(L, dev):
    import torch
    R = lc("rbd").Rbd(L)
    out = {}
    for tag in "ab":
        X, U, J = _sol(tag)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
        q6, c, f, jp = t(X[:6, :20].T), t(U[:12].T), t(U[12:].T), t(J.T)
        fk, err, tau = (torch.zeros(20, 12, dtype=torch.float64, device=dev) for _ in range(3))
        st = torch.cuda.current_stream().cuda_stream if dev == "cuda" else 0
        R.kinodyn_rows(20, q6.data_ptr(), c.data_ptr(), f.data_ptr(), jp.data_ptr(), fk.data_ptr(), err.data_ptr(), tau.data_ptr(), st)
        if dev == "cuda":
            torch.cuda.synchronize()
        out[tag] = (err.cpu().numpy(), tau.cpu().numpy())
    return out


def _check_rows(out):
    from oracle import kinodyn_oracle as ko
    for tag, band in (("a", 1e-3), ("b", 1e-2)):
        X, U, J = _sol(tag)
        fe, to = ko.kinematic_rows(X, U, J)
        err, tau = out[tag]
        assert np.abs(err - fe).max() <= 1e-12 and np.abs(tau - to).max() <= 1e-11 * max(1.0, np.abs(to).max())
        assert 0.98 * band <= np.abs(err).max() <= band * 1.002 and (np.abs(tau).reshape(-1, 4, 3).max(axis=(0, 1)) <= ko.TAU_MAX).all()


def test_kinodyn_rows_kernel_on_stored_solutions_emulated():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    _check_rows(_rows_through(L, "cpu"))


@pytest.mark.gpu
def test_kinodyn_rows_kernel_on_stored_solutions_gpu():
    L = lc("capi").LandingLib(20, device=0)
    _check_rows(_rows_through(L, "cuda"))
    L.close()


# ---- function layer of the whole refinement NLP (landing_kinodyn_nlp_eval): g and the exact Jacobian blocks --------------------------------
def _nlp_check(L, dev, N, B, seed, n_jac=2):
    """random points around a plausible posture: g vs the numpy oracle (1e-12), Jacobian blocks vs Richardson-extrapolated differences of the
    ORACLE's rows (1e-7 relative: the kernel differentiates exactly, the tolerance is the oracle's)"""
    import torch
    from oracle import kinodyn_oracle as ko
    mass, Ib, Ibi = lc("constants").robot_constants()
    Ib, Ibi = np.asarray(Ib), np.asarray(Ibi)
    R = lc("rbd").Rbd(L)
    nx, ng = R.kinodyn_nlp_dims(N)
    assert (nx, ng) == ko.nlp_dims(N)
    rng = np.random.default_rng(seed)
    dt = 0.02 + 0.03 * rng.random(N); mu = 0.75
    xs = np.zeros((B, nx))
    for b in range(B):
        X = np.zeros((12, N + 1)); U = np.zeros((24, N)); J = np.zeros((12, N))
        X[2] = 0.3 + 0.05 * rng.normal(size=N + 1); X[3:6] = 0.3 * rng.normal(size=(3, N + 1)); X[6:] = rng.normal(size=(6, N + 1)); X[:2] = 0.1 * rng.normal(size=(2, N + 1))
        U[:12] = (np.tile([0.2, -0.15, 0.0, 0.2, 0.15, 0.0, -0.2, -0.15, 0.0, -0.2, 0.15, 0.0], (N, 1)).T + 0.05 * rng.normal(size=(12, N)))
        U[12:] = 20.0 * rng.normal(size=(12, N)); U[14::3] = np.abs(U[14::3])
        J[:] = (np.tile([0.0, -0.8, 1.6], 4)[:, None] + 0.2 * rng.normal(size=(12, N)))
        xs[b] = ko.pack_x(X, U, J)
    dx = torch.tensor(xs, dtype=torch.float64, device=dev)
    g = torch.zeros(B, ng, dtype=torch.float64, device=dev); jac = torch.zeros(B, N, 141, 72, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream if dev == "cuda" else 0
    R.kinodyn_nlp_eval(B, N, dx.data_ptr(), dt, mass, Ib, Ibi, mu, g.data_ptr(), jac.data_ptr(), st)
    if dev == "cuda":
        torch.cuda.synchronize()
    g, jac = g.cpu().numpy(), jac.cpu().numpy()
    for b in range(B):
        go = ko.nlp_g(xs[b], N, dt, mass, Ib, Ibi, mu)
        assert np.abs(g[b] - go).max() <= 1e-12 * max(1.0, np.abs(go).max()), np.abs(g[b] - go).max()
    for b in range(min(B, n_jac)):
        for k in (0, N - 1):
            last = k == N - 1
            Jo = ko.stage_jacobian(ko.gather_w(xs[b], N, k), dt[k], last, mass, Ib, Ibi, mu)
            Jk = jac[b, k, :Jo.shape[0]]
            assert np.abs(Jk - Jo).max() <= 1e-7 * max(1.0, np.abs(Jo).max()), (k, np.abs(Jk - Jo).max())
            if last:
                assert (Jk[:, 60:] == 0.0).all()
    return R


def test_kinodyn_nlp_function_layer_emulated():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    _nlp_check(L, "cpu", N=3, B=2, seed=1, n_jac=1)


def _stored_solution_rows(L, dev):
    """the reference's stored kinodynamic solutions through the kernel: every row group inside the script's bounds"""
    import torch
    from oracle import kinodyn_oracle as ko
    mass, Ib, Ibi = lc("constants").robot_constants()
    R = lc("rbd").Rbd(L)
    N = 20
    for tag, mu, band in (("a", 0.5, 1e-3), ("b", 1.0, 1e-2)):
        X, U, J = _sol(tag)
        x = ko.pack_x(X, U, J)
        dx = torch.tensor(x[None], dtype=torch.float64, device=dev)
        g = torch.zeros(1, ko.nlp_dims(N)[1], dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream if dev == "cuda" else 0
        R.kinodyn_nlp_eval(1, N, dx.data_ptr(), DT, mass, np.asarray(Ib), np.asarray(Ibi), mu, g.data_ptr(), 0, st)
        if dev == "cuda":
            torch.cuda.synchronize()
        g = g.cpu().numpy()[0]
        assert np.abs(g - ko.nlp_g(x, N, DT, mass, np.asarray(Ib), np.asarray(Ibi), mu)).max() <= 1e-12
        for k in range(N):
            last = k == N - 1
            r = g[48 + 141 * k: 48 + 141 * k + (117 if last else 141)]
            assert np.abs(r[:12]).max() <= 5e-5                                       # Euler defects
            assert r[12:16].min() >= -1e-6                                            # f_z >= 0
            S = 9 if last else 15
            for l in range(4):
                q = r[16 + S * l: 16 + S * (l + 1)]
                assert q[0] >= -1e-5 and q[1] <= 1e-3 * 1.001                         # c_z >= 0, LCP
                if not last:
                    assert np.abs(q[2:8]).max() <= 1e-3 * 1.001                       # no slip
                assert (np.abs(q[-3:]) <= ko.TAU_MAX).all()                           # leg torques
            o = 16 + 4 * S
            assert r[o:o + 16].max() <= 1e-5      # friction pyramid: all four groups are `lhs - rhs <= 0` (Opti's canonical form, round 6)
            assert np.abs(r[o + 17:o + 29]).max() <= band * 1.002                     # FK band


def test_kinodyn_nlp_rows_on_stored_solutions_emulated():
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    _stored_solution_rows(L, "cpu")


@pytest.mark.gpu
def test_kinodyn_nlp_function_layer_gpu():
    """N = 20 intervals (the script's size), 1024 members: g of every member and sampled Jacobian blocks against the oracle; the stored solutions"""
    L = lc("capi").LandingLib(20, device=0)
    _nlp_check(L, "cuda", N=20, B=1024, seed=2, n_jac=2)
    _stored_solution_rows(L, "cuda")
    L.close()


def _hess_check(L, dev, N, B, seed, oracle_cols):
    """Hessian blocks of lam' g: symmetric, structural zeros in the [X_k+1, c_k+1] block, equal to central differences of the kernel's own exact
    Jacobian (J' lam at w +- h e_j, every column of two intervals, 1e-6) and -- for a few columns -- of the ORACLE's Richardson Jacobian (1e-5)"""
    import torch
    from oracle import kinodyn_oracle as ko
    mass, Ib, Ibi = lc("constants").robot_constants()
    Ib, Ibi = np.asarray(Ib), np.asarray(Ibi)
    R = lc("rbd").Rbd(L)
    nx, ng = R.kinodyn_nlp_dims(N)
    rng = np.random.default_rng(seed)
    dt = 0.02 + 0.03 * rng.random(N); mu = 0.75
    xs = 0.3 * rng.normal(size=(B, nx)); xs[:, 2:12 * (N + 1):12] += 0.3
    lam = rng.normal(size=(B, ng))
    t = lambda a_: torch.tensor(np.ascontiguousarray(a_), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream if dev == "cuda" else 0
    sync = (lambda: torch.cuda.synchronize()) if dev == "cuda" else (lambda: None)
    dx, dl = t(xs), t(lam)
    H = torch.zeros(B, N, 72, 72, dtype=torch.float64, device=dev)
    R.kinodyn_nlp_hess(B, N, dx.data_ptr(), dt, mass, Ib, Ibi, mu, dl.data_ptr(), H.data_ptr(), st); sync()
    H = H.cpu().numpy()
    assert np.array_equal(H, H.transpose(0, 1, 3, 2)) and np.isfinite(H).all()
    assert (H[:, :, 48:, 48:] == 0.0).all() and (H[:, :, :, 48:60] == 0.0).all() and (H[:, N - 1, :, 60:] == 0.0).all()
    assert np.abs(H[:, :, :48, :48]).max() > 1e-3
    h = 1e-5
    def jt_lam(xp):      # J_k' lam_k of every interval at the points xp [n, nx] (kernel Jacobian)
        n = xp.shape[0]
        J = torch.zeros(n, N, 141, 72, dtype=torch.float64, device=dev)
        dxp = t(xp)
        R.kinodyn_nlp_eval(n, N, dxp.data_ptr(), dt, mass, Ib, Ibi, mu, 0, J.data_ptr(), st); sync()
        return J.cpu().numpy()
    b = 0
    for k in (0, N - 1):
        nr = 117 if k == N - 1 else 141
        lk = lam[b, 48 + 141 * k: 48 + 141 * k + nr]
        cols = [j for j in range(72) if ko.w_index(N, k, j) >= 0]
        xp = np.repeat(xs[b:b + 1], 2 * len(cols), axis=0)
        for q, j in enumerate(cols):
            xp[2 * q, ko.w_index(N, k, j)] += h; xp[2 * q + 1, ko.w_index(N, k, j)] -= h
        J = jt_lam(xp)[:, k, :nr]
        for q, j in enumerate(cols):
            col = ((J[2 * q] - J[2 * q + 1]) / (2 * h)).T @ lk
            assert np.abs(H[b, k, :, j] - col).max() <= 1e-6 * max(1.0, np.abs(col).max()), (k, j, np.abs(H[b, k, :, j] - col).max())
    k = 0
    w = ko.gather_w(xs[b], N, k); lk = lam[b, 48:48 + 141]
    for j in oracle_cols:
        e = np.zeros(72); e[j] = 1e-3
        col = (ko.stage_jacobian(w + e, dt[k], False, mass, Ib, Ibi, mu) - ko.stage_jacobian(w - e, dt[k], False, mass, Ib, Ibi, mu)).T @ lk / 2e-3
        assert np.abs(H[b, k, :, j] - col).max() <= 1e-5 * max(1.0, np.abs(col).max()), (j, np.abs(H[b, k, :, j] - col).max())


def test_kinodyn_nlp_hessian_emulated():
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    _hess_check(L, "cpu", N=2, B=1, seed=3, oracle_cols=(4, 26, 38))      # a rotation angle, a vertical force, a hip joint


def test_kinodyn_hessian_pair_table_does_not_depend_on_the_first_call_emulated():
    """the table of structurally non-zero Hessian pairs is probed once per context; probed with the first call's parameter VALUES, a first call with
    a symmetric inertia (omega x I omega = 0: 36 exact zeros) left every later call on that context with those entries missing.  Now the probe uses fixed generic
    constants and landing_rbd_set_model invalidates the table: symmetric inertia first, the real one second == the real one on a fresh context, bit for bit."""
    import torch
    mass, Ib, Ibi = lc("constants").robot_constants()
    Ib, Ibi = np.asarray(Ib), np.asarray(Ibi)
    N, B = 2, 1
    rng = np.random.default_rng(5)
    emu = os.path.join(ROOT, "tests", "emu", "liblanding_emu.so")
    def hess(L, R, Ib_, Ibi_):
        nx, ng = R.kinodyn_nlp_dims(N)
        r2 = np.random.default_rng(6)
        xs = 0.3 * r2.normal(size=(B, nx)); xs[:, 2:12 * (N + 1):12] += 0.3
        lam = r2.normal(size=(B, ng)); dt = np.array([0.03, 0.04])
        dx, dl = torch.tensor(xs), torch.tensor(lam)
        H = torch.zeros(B, N, 72, 72, dtype=torch.float64)
        R.kinodyn_nlp_hess(B, N, dx.data_ptr(), dt, mass, Ib_, Ibi_, 0.75, dl.data_ptr(), H.data_ptr(), 0)
        return H.numpy().copy()
    L1 = lc("capi").LandingLib(20, lib_path=emu); R1 = lc("rbd").Rbd(L1)
    sym = np.full(3, 0.03)
    Hs = hess(L1, R1, sym, 1.0 / sym)
    assert (Hs[0, :, 6:9, 6:9] == 0.0).all()                      # omega x (I omega) vanishes for a spherical inertia
    H2 = hess(L1, R1, Ib, Ibi)                                    # ... same context, the real inertia
    L2 = lc("capi").LandingLib(20, lib_path=emu); R2 = lc("rbd").Rbd(L2)
    H3 = hess(L2, R2, Ib, Ibi)
    assert np.abs(H3[0, :, 6:9, 6:9]).max() > 1e-3 and np.array_equal(H2, H3)
    R1b = lc("rbd").Rbd(L1)                                       # landing_rbd_set_model again: the table is invalidated and rebuilt at the next call
    assert np.array_equal(hess(L1, R1b, Ib, Ibi), H3)
    L1.close(); L2.close()


@pytest.mark.gpu
def test_kinodyn_nlp_hessian_gpu():
    L = lc("capi").LandingLib(20, device=0)
    _hess_check(L, "cuda", N=20, B=64, seed=4, oracle_cols=(3, 8, 14, 26, 40, 71))
    L.close()


def _base_forms_check(L, dev, N, B):
    """The derivative kernels exist in two forms (csrc/rbd_kernels.hip kd_stage_rows): with the base transform of the tree taken from R and pos -- the reference model's
    floating base Px Py Pz Rx Ry Rz, what landing_rbd_set_model recognises -- and composed joint by joint.  A model that differs from the standard one by a tree offset of
    1e-300 m takes the second form: g, the Jacobian blocks and the Hessian blocks of both agree to rounding; an offset of 0.25 m shows up in g (the second form runs)."""
    import torch
    mass, Ib, Ibi = lc("constants").robot_constants()
    Ib, Ibi = np.asarray(Ib), np.asarray(Ibi)
    R = lc("rbd").Rbd(L)
    nx, ng = R.kinodyn_nlp_dims(N)
    rng = np.random.default_rng(5)
    dt = 0.02 + 0.03 * rng.random(N); mu = 0.75
    xs = 0.3 * rng.normal(size=(B, nx)); xs[:, 2:12 * (N + 1):12] += 0.3
    lam = rng.normal(size=(B, ng))
    t = lambda a_: torch.tensor(np.ascontiguousarray(a_), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream if dev == "cuda" else 0
    sync = (lambda: torch.cuda.synchronize()) if dev == "cuda" else (lambda: None)
    dx, dl = t(xs), t(lam)
    out = []
    for offset in (0.0, 1e-300, 0.25):
        R.model.r[0][0] = offset
        L._check(L.lib.landing_rbd_set_model(L.ctx, __import__("ctypes").byref(R.model)), "landing_rbd_set_model")
        g = torch.zeros(B, ng, dtype=torch.float64, device=dev); J = torch.zeros(B, N, 141, 72, dtype=torch.float64, device=dev); H = torch.zeros(B, N, 72, 72, dtype=torch.float64, device=dev)
        R.kinodyn_nlp_eval(B, N, dx.data_ptr(), dt, mass, Ib, Ibi, mu, g.data_ptr(), J.data_ptr(), st)
        R.kinodyn_nlp_hess(B, N, dx.data_ptr(), dt, mass, Ib, Ibi, mu, dl.data_ptr(), H.data_ptr(), st); sync()
        out.append((g.cpu().numpy(), J.cpu().numpy(), H.cpu().numpy()))
    for a, b, name in zip(out[0], out[1], ("g", "jac", "hess")):
        assert np.isfinite(a).all() and np.abs(a).max() > 1e-3, name
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(a).max()), (name, np.abs(a - b).max())
    # a real offset of the tree's root moves every forward-kinematics row by a constant: only the joint-by-joint form can see it -- it is the form a non-standard model gets
    assert np.abs(out[2][0] - out[0][0]).max() > 0.1
    for a, b, name in zip(out[0][1:], out[2][1:], ("jac", "hess")):
        assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(a).max()), (name, np.abs(a - b).max())
    R.model.r[0][0] = 0.0
    L._check(L.lib.landing_rbd_set_model(L.ctx, __import__("ctypes").byref(R.model)), "landing_rbd_set_model")


def test_kinodyn_base_transform_forms_agree_emulated():
    subprocess.run(["make", "-C", os.path.join(ROOT, "landing-controller_amd", "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(3, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    _base_forms_check(L, "cpu", N=3, B=2)
    L.close()


@pytest.mark.gpu
def test_kinodyn_base_transform_forms_agree_gpu():
    L = lc("capi").LandingLib(20, device=0)
    _base_forms_check(L, "cuda", N=20, B=64)
    L.close()


def test_kinodyn_bounds_hold_on_the_stored_solution():
    """host side (landing-controller_amd/kinodyn.py): lbg / ubg in the kernel's row order with the script's values -- the stored solution of the
    CURRENT formulation (prevSoln.mat: FK band 1e-2) lies inside them to the producing solver's tolerance, rows through the oracle"""
    from oracle import kinodyn_oracle as ko
    kd = lc("kinodyn")
    mass, Ib, Ibi = lc("constants").robot_constants()
    X, U, J = _sol("b")
    N = 20
    assert kd.dims(N) == ko.nlp_dims(N) and np.array_equal(kd.pack_x(X, U, J), ko.pack_x(X, U, J))
    Xr, Ur, Jr = kd.unpack_x(kd.pack_x(X, U, J), N)
    assert np.array_equal(Xr, X) and np.array_equal(Ur, U) and np.array_equal(Jr, J)
    x = kd.pack_x(X, U, J)
    g = ko.nlp_g(x, N, DT, mass, np.asarray(Ib), np.asarray(Ibi), 1.0)
    kb = kd.kin_box_of(X[3:6, 0], X[9:12, 0])
    lb, ub = kd.bounds(N, X[:6, 0], X[6:, 0], U[:12, 0], kb)
    viol = np.maximum(np.maximum(lb - g, g - ub), 0.0)
    # the friction rows depend on the run's mu (1.0 reproduces the file), the terminal box on that run's settings: every other row group holds
    stage = np.arange(48, g.size)
    assert viol[stage].max() <= 2e-3, (viol[stage].max(), int(np.argmax(viol[stage])))
    assert viol[:24].max() == 0.0
    # known answer: the stored trajectory ends ON the script's terminal reference (q_term_ref z = 0.25, :222-223): f* = 0 for this instance, the
    # point is a feasible global minimiser of the NLP as defined here (gradient of f ~ 1e-5: stationary with zero multipliers)
    f, gr = kd.terminal_cost(x, N, [0, 0, 0.25, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    assert 0.0 <= f <= 1e-10 and np.abs(gr).max() <= 1e-4 and np.count_nonzero(gr) <= 12


def test_kinodyn_oracle_kkt_of_the_stored_solution():
    """the certificate the next round's solver will be held to (oracle/kinodyn_oracle.py::kkt, convention of the SRBM oracle): the reference's stored
    solution with zero multipliers is a KKT point of the NLP as defined here up to the producing solver's feasibility tolerance"""
    from oracle import kinodyn_oracle as ko
    kd = lc("kinodyn")
    mass, Ib, Ibi = lc("constants").robot_constants()
    X, U, J = _sol("b"); N = 20
    x = kd.pack_x(X, U, J)
    lb, ub = kd.bounds(N, X[:6, 0], X[6:, 0], U[:12, 0], kd.kin_box_of(X[3:6, 0], X[9:12, 0]))
    f, gf = kd.terminal_cost(x, N, [0, 0, 0.25, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    pr, du, co = ko.kkt(x, np.zeros(lb.size), N, DT, mass, np.asarray(Ib), np.asarray(Ibi), 1.0, lb, ub, gf)
    assert pr <= 2e-3 and du <= 1e-4 and co == 0.0, (pr, du, co)


def _casadi_face_check(L, N, seed, hess_cols, tol_h):
    """the CasADi-external face of the kinodynamic NLP (landing_kinodyn_casadi_eval_host = what landingCtrller_KNITRO_mi355x.so forwards to) against the oracle:
    g, the CCS Jacobian, grad_gamma_x (complex-step oracle), the CCS Hessian (central differences of the oracle's exact gradient), grad_gamma_p (differences of the
    oracle's Lagrangian in every parameter that enters f or g), f / grad_f, and lbg / ubg from p"""
    from oracle import kinodyn_oracle as ko
    kd = lc("kinodyn")
    mass, Ib, Ibi = lc("constants").robot_constants()
    Ib, Ibi = np.asarray(Ib), np.asarray(Ibi)
    R = lc("rbd").Rbd(L)
    nx, ng = R.kinodyn_nlp_dims(N)
    rng = np.random.default_rng(seed)
    dt = 0.02 + 0.03 * rng.random(N); mu = 0.75
    x = 0.3 * rng.normal(size=nx); x[2:12 * (N + 1):12] += 0.3
    lam = rng.normal(size=ng); lam_f = 0.7
    Xref = rng.normal(size=(12, N + 1)); QN = np.array(kd.QN_DEFAULT, float) + rng.random(12)
    q_init = np.array([0, 0, 0.6, 0.1, -0.3, 0.05]); qd_init = np.array([0.1, -0.2, 0.3, 0.5, -0.4, -3.0])
    vals = dict(Xref=Xref, dt=dt, q_init=q_init, qd_init=qd_init, c_init=kd.c_init_of(q_init), jpos_min=kd.JPOS_MIN, jpos_max=kd.JPOS_MAX,
                q_term_min=[-10, -10, 0.15, -0.1, -0.1, -10], q_term_max=[10, 10, 5, 0.1, 0.1, 10], qd_term_min=[-10, -10, -10, -.5, -.5, -.5], qd_term_max=[10, 10, 10, .5, .5, .5],
                q_min=[-10, -10, 0.075, -10, -10, -10], QN=QN, mu=mu, l_leg_max=0.4, mass=mass, Ib=Ib, Ib_inv=Ibi, kin_box=kd.kin_box_of(q_init[3:6], qd_init[3:6]))
    p = kd.pack_params_knitro(N, **vals)
    off, npar = kd.knitro_param_offsets(N)
    assert npar == 13 * N + 113 == R.kinodyn_casadi_np(N)
    r = R.kinodyn_casadi_eval(N, x, p, lam_f, lam)
    d = x[12 * N:12 * N + 12] - Xref[:, N]
    gf = np.zeros(nx); gf[12 * N:12 * N + 12] = 2 * QN * d
    assert abs(r["f"] - QN @ d ** 2) <= 1e-12 and np.array_equal(r["grad_f"], gf)
    g0 = ko.nlp_g(x, N, dt, mass, Ib, Ibi, mu)
    assert np.abs(r["g"] - g0).max() <= 1e-11
    (jc, jr), (hc, hr) = R.kinodyn_casadi_pattern(N, 0), R.kinodyn_casadi_pattern(N, 1)
    J = np.zeros((ng, nx)); H = np.zeros((nx, nx))
    for c in range(nx):
        J[jr[jc[c]:jc[c + 1]], c] = r["jac"][jc[c]:jc[c + 1]]
        H[hr[hc[c]:hc[c + 1]], c] = r["hess"][hc[c]:hc[c + 1]]
        assert (hr[hc[c]:hc[c + 1]] <= c).all()                     # upper triangle
    H = H + np.triu(H, 1).T
    assert np.abs(J - ko.nlp_jacobian(x, N, dt, mass, Ib, Ibi, mu)).max() <= 1e-6
    gl = lambda xx: ko.grad_lagrangian_batch(xx[None], lam[None], N, dt, mass, Ib, Ibi, mu, lam_f * 2 * np.where(np.arange(nx) // 12 == N, 1.0, 0.0)[None] * 0)[0]
    def grad_gamma(xx):
        gfx = np.zeros(nx); gfx[12 * N:12 * N + 12] = lam_f * 2 * QN * (xx[12 * N:12 * N + 12] - Xref[:, N])
        return ko.grad_lagrangian_batch(xx[None], lam[None], N, dt, mass, Ib, Ibi, mu, gfx[None])[0]
    assert np.abs(r["ggx"] - grad_gamma(x)).max() <= 1e-9 * max(1.0, np.abs(r["ggx"]).max())
    h = 1e-5
    for j in hess_cols:
        e = np.zeros(nx); e[j] = h
        col = (grad_gamma(x + e) - grad_gamma(x - e)) / (2 * h)
        assert np.abs(H[:, j] - col).max() <= tol_h * max(1.0, np.abs(col).max()), (j, np.abs(H[:, j] - col).max())
    def gamma(pp):
        o = {k: pp[a:b] for k, (a, b) in off.items()}
        dd = x[12 * N:12 * N + 12] - o["Xref"][12 * N:]
        return lam_f * (o["QN"] @ dd ** 2) + lam @ ko.nlp_g(x, N, o["dt"], o["mass"][0], o["Ib"], o["Ib_inv"], o["mu"][0])
    for name in ("Xref", "dt", "QN", "mu", "mass", "Ib", "Ib_inv", "kin_box", "q_init"):
        a, b = off[name]
        for i in sorted({a, b - 1}):
            hp = 1e-6 * max(1.0, abs(p[i])); e = np.zeros(npar); e[i] = hp
            fd = (gamma(p + e) - gamma(p - e)) / (2 * hp)
            assert abs(r["ggp"][i] - fd) <= 1e-6 * max(1.0, abs(fd)), (name, i, r["ggp"][i], fd)
    lb, ub = R.kinodyn_casadi_bounds(N, p)
    l0, u0 = kd.bounds(N, q_init, qd_init, vals["c_init"], vals["kin_box"])
    assert np.array_equal(lb, l0) and np.array_equal(ub, u0)
    # the `>=` friction rows are held the way Opti holds them (optistack_internal.cpp:793-813): -km f_z - f_xy <= 0
    k0 = 48
    fr = slice(k0 + 12 + 4 + 4 * 15 + 4, k0 + 12 + 4 + 4 * 15 + 8)      # second friction group of interval 0: [defects 12 | f_z 4 | 4 legs x 15 | friction 16 ...]
    assert np.isneginf(lb[fr]).all() and (ub[fr] == 0.0).all()
    U0 = x[12 * (N + 1) + 12 * N:12 * (N + 1) + 12 * N + 24]
    assert np.allclose(r["g"][fr], -0.71 * mu * U0[12 + 2::3] - U0[12 + 0::3], atol=1e-13)


def test_kinodyn_casadi_face_emulated():
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    _casadi_face_check(L, N=2, seed=11, hess_cols=(3, 7, 30, 40, 60, 75, 100), tol_h=1e-6)
    L.close()
