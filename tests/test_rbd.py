"""SURVEY 8(f) rows N2 (18-DoF floating-base dynamics and its linearisation, BASELINE configs[3]) and N1 (rows of the
kinodynamic refinement stage): the HIP kernels -- compact plux(E, r) / (m, h, Ibar) algebra, one thread per evaluation --
against the numpy oracle that restates the reference with full 6 x 6 Pluecker matrices (HandC.m, casadi_compatible_dynamics.m,
get_forward_kin_foot.m, get_foot_jacobians_mc.m, landing_optimization.m:152-189).
CPU: the same sources through tests/emu.  GPU: N=40 knots x members at BASELINE configs[3]'s shape (sampled).
Tolerances (fp64): H, C, FK, torques 1e-11 relative; qdd 1e-9 (18 x 18 solve, cond ~1e4); finite-difference Jacobians
1e-5 relative against the oracle's own central differences (same step)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")


def _points(rng, n):
    q = np.zeros((n, 18)); qd = rng.normal(size=(n, 18)); tau = 5 * rng.normal(size=(n, 18)); f = np.zeros((n, 12))
    for i in range(n):
        q[i, :3] = [rng.normal() * 0.3, rng.normal() * 0.3, 0.3 + 0.2 * rng.random()]
        q[i, 3:6] = 0.5 * rng.normal(size=3)
        q[i, 6:] = np.tile([0.0, -0.8, 1.6], 4) + 0.3 * rng.normal(size=12)
        f[i] = np.tile([3.0, -2.0, 25.0], 4) + 6 * rng.normal(size=12)
    return q, qd, tau, f


def _check_dynamics(ro, q, qd, tau, f, H, Cb, qdd, A, Hinv, fd_h, with_f):
    for i in range(q.shape[0]):
        ff = f[i].reshape(4, 3) if with_f else None
        Ho, Co = ro.hand_c(q[i], qd[i], ff)
        assert np.max(np.abs(H[i] - Ho)) <= 1e-11 * np.max(np.abs(Ho))
        assert np.max(np.abs(Cb[i] - Co)) <= 1e-11 * max(1.0, np.max(np.abs(Co)))
        qo = np.linalg.solve(Ho, tau[i] - Co)
        assert np.max(np.abs(qdd[i] - qo)) <= 1e-9 * max(1.0, np.max(np.abs(qo)))
        assert np.max(np.abs(Hinv[i] - np.linalg.inv(Ho))) <= 1e-9 * np.max(np.abs(np.linalg.inv(Ho)))
    _, Ao, _ = ro.fd_linearisation(q[0], qd[0], tau[0], f[0].reshape(4, 3) if with_f else None, h=fd_h)
    assert np.max(np.abs(A[0] - Ao)) <= 1e-5 * max(1.0, np.max(np.abs(Ao))), np.max(np.abs(A[0] - Ao))


@pytest.mark.parametrize("with_f", [False, True])
def test_fb_dynamics_emulated(with_f):
    from oracle import rbd_oracle as ro
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    R = lc("rbd").Rbd(L)
    n = 3
    q, qd, tau, f = _points(np.random.default_rng(5), n)
    H = np.zeros((n, 18, 18)); Cb = np.zeros((n, 18)); qdd = np.zeros((n, 18)); A = np.zeros((n, 18, 36)); Hinv = np.zeros((n, 18, 18))
    p = lambda a: a.ctypes.data
    R.fb_dynamics(n, p(q), p(qd), p(tau), p(f) if with_f else 0, p(H), p(Cb), p(qdd), p(A), p(Hinv), 1e-6)
    _check_dynamics(ro, q, qd, tau, f, H, Cb, qdd, A, Hinv, 1e-6, with_f)


@pytest.mark.parametrize("with_f", [False, True])
def test_fb_exact_linearisation_emulated(with_f):
    """fd_h = 0: the exact linearisation (forward-mode tangents of the inverse dynamics, -H^-1 dID/dz) against the oracle's
    Richardson-extrapolated central differences (1e-8 relative; the central-difference kernel only reaches 1e-5), with and without
    caller buffers for qdd / H^-1 (the context scratch path)"""
    from oracle import rbd_oracle as ro
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    R = lc("rbd").Rbd(L)
    n = 2
    q, qd, tau, f = _points(np.random.default_rng(7), n)
    p = lambda a: a.ctypes.data
    A = np.zeros((n, 18, 36)); A2 = np.zeros((n, 18, 36)); qdd = np.zeros((n, 18)); Hinv = np.zeros((n, 18, 18))
    R.fb_dynamics(n, p(q), p(qd), p(tau), p(f) if with_f else 0, d_qdd=p(qdd), d_A=p(A), d_Hinv=p(Hinv), fd_h=0.0)
    R.fb_dynamics(n, p(q), p(qd), p(tau), p(f) if with_f else 0, d_A=p(A2), fd_h=0.0)          # scratch for qdd / H^-1
    assert np.array_equal(A, A2)
    for i in range(n):
        ff = f[i].reshape(4, 3) if with_f else None
        Ao = ro.richardson_linearisation(q[i], qd[i], tau[i], ff)
        assert np.max(np.abs(A[i] - Ao)) <= 1e-8 * max(1.0, np.max(np.abs(Ao))), np.max(np.abs(A[i] - Ao))
        Ho, Co = ro.hand_c(q[i], qd[i], ff)
        assert np.max(np.abs(qdd[i] - np.linalg.solve(Ho, tau[i] - Co))) <= 1e-9 * max(1.0, np.max(np.abs(qdd[i])))
        assert np.max(np.abs(Hinv[i] - np.linalg.inv(Ho))) <= 1e-9 * np.max(np.abs(np.linalg.inv(Ho)))


def test_model_matches_reference_constants():
    """the compact model reproduces the reference's composite inertia at the home pose"""
    from oracle import rbd_oracle as ro
    K = lc("constants")
    q = np.concatenate([np.zeros(6), np.tile(K.Q_LEG_HOME, 4)])
    H, Cg = ro.hand_c(q, np.zeros(18))
    mass, Ib, _ = K.robot_constants()
    assert abs(H[2, 2] - mass) < 1e-12 and np.allclose(np.diag(H)[3:6], Ib, rtol=1e-12) and abs(Cg[2] - 9.81 * mass) < 1e-10
    assert np.allclose(ro.quad3d_model()["tau_max"], lc("rbd").TAU_MAX)


def test_kinodyn_rows_emulated():
    from oracle import rbd_oracle as ro
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    R = lc("rbd").Rbd(L)
    rng = np.random.default_rng(2)
    n = 5
    q, _, _, f = _points(rng, n)
    q6 = np.ascontiguousarray(q[:, :6]); jp = np.ascontiguousarray(q[:, 6:]); c = rng.normal(size=(n, 12)) * 0.3
    fk = np.zeros((n, 12)); err = np.zeros((n, 12)); tau = np.zeros((n, 12))
    p = lambda a: a.ctypes.data
    R.kinodyn_rows(n, p(q6), p(c), p(f), p(jp), p(fk), p(err), p(tau))
    for i in range(n):
        fko, eo, to = ro.kinodyn_rows(q6[i], c[i], f[i], jp[i])
        assert np.max(np.abs(fk[i] - fko)) <= 1e-12 and np.max(np.abs(err[i] - eo)) <= 1e-12
        assert np.max(np.abs(tau[i] - to)) <= 1e-11 * max(1.0, np.max(np.abs(to)))


@pytest.mark.gpu
def test_fb_dynamics_gpu_config4_shape():
    """BASELINE configs[3] shape: N=40 knots x 1024 members = 40 960 configurations, H / C / qdd / linearisation in one call each;
    a sample is compared with the oracle, the rest through properties (H symmetric positive definite, H Hinv = 1)"""
    import torch
    from oracle import rbd_oracle as ro
    L = lc("capi").LandingLib(40, device=0)
    R = lc("rbd").Rbd(L)
    n = 40 * 1024
    rng = np.random.default_rng(11)
    q, qd, tau, f = _points(rng, 64)
    rep = n // 64
    Q = np.tile(q, (rep, 1)) + 1e-3 * rng.normal(size=(n, 18)); QD = np.tile(qd, (rep, 1)); TAU = np.tile(tau, (rep, 1)); F = np.tile(f, (rep, 1))
    dev = "cuda"
    t = lambda a: torch.tensor(a, device=dev)
    dq, dqd, dtau, df = t(Q), t(QD), t(TAU), t(F)
    mk = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float64)
    H, Cb, qdd, A, Hinv = mk(n, 18, 18), mk(n, 18), mk(n, 18), mk(n, 18, 36), mk(n, 18, 18)
    R.fb_dynamics(n, dq.data_ptr(), dqd.data_ptr(), dtau.data_ptr(), df.data_ptr(), H.data_ptr(), Cb.data_ptr(), qdd.data_ptr(), A.data_ptr(), Hinv.data_ptr(), 1e-6,
                  torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Hh, Ch, qh, Ah, Hi = H.cpu().numpy(), Cb.cpu().numpy(), qdd.cpu().numpy(), A.cpu().numpy(), Hinv.cpu().numpy()
    idx = np.arange(0, n, n // 8)[:8]
    _check_dynamics(ro, Q[idx], QD[idx], TAU[idx], F[idx], Hh[idx], Ch[idx], qh[idx], Ah[idx], Hi[idx], 1e-6, True)
    assert np.isfinite(Ah).all() and np.abs(Hh - np.swapaxes(Hh, 1, 2)).max() == 0.0
    E = torch.matmul(H, Hinv) - torch.eye(18, device=dev, dtype=torch.float64)
    assert E.abs().max().item() < 1e-8
    # exact linearisation at the same shape: equals the central differences to their accuracy everywhere, the oracle's Richardson
    # extrapolation to 1e-8 on a sample
    Ax = mk(n, 18, 36)
    R.fb_dynamics(n, dq.data_ptr(), dqd.data_ptr(), dtau.data_ptr(), df.data_ptr(), d_A=Ax.data_ptr(), fd_h=0.0, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Axh = Ax.cpu().numpy()
    assert np.isfinite(Axh).all() and np.abs(Axh - Ah).max() <= 2e-5 * max(1.0, np.abs(Ah).max())
    for i in idx[:3]:
        Ao = ro.richardson_linearisation(Q[i], QD[i], TAU[i], F[i].reshape(4, 3))
        assert np.max(np.abs(Axh[i] - Ao)) <= 1e-8 * max(1.0, np.max(np.abs(Ao)))
    L.close()


@pytest.mark.gpu
def test_kinodyn_rows_gpu_on_solved_batch():
    """kinodynamic feasibility screen of SRBM solutions: joint angles by a few Newton steps of the FK, then the torque rows of
    landing_optimization.m:165-171 for every stage of every member; GPU rows == oracle rows on a sample"""
    import torch
    from oracle import rbd_oracle as ro
    capi, Pm = lc("capi"), lc("problem")
    N, B = 40, 32
    L = capi.LandingLib(N, device=0)
    R = lc("rbd").Rbd(L)
    P, X0, _, _ = Pm.make_batch(B, N, 0.6, seed=4)
    r = L.solve_host(P, X0)
    assert (r["status"] == 0).all()
    q6 = np.zeros((B, N, 6)); c = np.zeros((B, N, 12)); f = np.zeros((B, N, 12))
    for b in range(B):
        Xs, Us = Pm.split_solution(N, r["x"][b])
        q6[b] = Xs[:6, :N].T; c[b] = Us[:12].T; f[b] = Us[12:].T
    jp = np.tile(np.tile([0.0, -0.8, 1.6], 4), (B, N, 1)) + 0.05 * np.random.default_rng(0).normal(size=(B, N, 12))
    n = B * N
    t = lambda a: torch.tensor(a.reshape(n, -1), device="cuda")
    dq, dc, df, dj = t(q6), t(c), t(f), t(jp)
    fk, err, tau = (torch.zeros(n, 12, device="cuda", dtype=torch.float64) for _ in range(3))
    R.kinodyn_rows(n, dq.data_ptr(), dc.data_ptr(), df.data_ptr(), dj.data_ptr(), fk.data_ptr(), err.data_ptr(), tau.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    fkh, eh, th = fk.cpu().numpy(), err.cpu().numpy(), tau.cpu().numpy()
    for i in range(0, n, 97):
        fko, eo, to = ro.kinodyn_rows(q6.reshape(n, 6)[i], c.reshape(n, 12)[i], f.reshape(n, 12)[i], jp.reshape(n, 12)[i])
        assert np.max(np.abs(fkh[i] - fko)) <= 1e-12 and np.max(np.abs(eh[i] - eo)) <= 1e-12 and np.max(np.abs(th[i] - to)) <= 1e-10 * max(1.0, np.max(np.abs(to)))
    L.close()


def test_leg_ik_emulated_roundtrip():
    """IK of the kinodynamic screen: FK(q6, IK(q6, FK(q6, jpos*))) == FK(q6, jpos*) for joint angles inside the limits"""
    from oracle import rbd_oracle as ro
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(20, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    R = lc("rbd").Rbd(L)
    rng = np.random.default_rng(9)
    n = 6
    q6 = np.concatenate([rng.normal(size=(n, 2)) * 0.2, 0.3 + 0.1 * rng.random((n, 1)), 0.4 * rng.normal(size=(n, 3))], axis=1)
    jp_true = np.tile([0.0, -0.9, 1.7], (n, 4)) + 0.25 * rng.normal(size=(n, 12))
    c = np.array([ro.forward_kin_foot(np.concatenate([q6[i], jp_true[i]])).reshape(12) for i in range(n)])
    jp = np.zeros((n, 12)); res = np.zeros((n, 4))
    p = lambda a: a.ctypes.data
    R.leg_ik(n, p(q6), p(c), p(jp), p(res), iters=20)
    assert res.max() < 1e-9, res
    for i in range(n):
        assert np.max(np.abs(ro.forward_kin_foot(np.concatenate([q6[i], jp[i]])).reshape(12) - c[i])) < 1e-9
    lo, hi = lc("rbd").JPOS_MIN, lc("rbd").JPOS_MAX
    assert (jp >= lo - 1e-12).all() and (jp <= hi + 1e-12).all()


@pytest.mark.gpu
def test_kinodynamic_screen_of_solved_batch():
    """SRBM solutions -> joint angles by IK at every stage -> FK band and torque limits of landing_optimization.m:165-171,186-187:
    the screen runs on the whole batch; FK residuals after IK are tiny wherever the foot is reachable, torques are finite, and the
    verdict agrees with the oracle's rows on a sample"""
    import torch
    from oracle import rbd_oracle as ro
    capi, Pm = lc("capi"), lc("problem")
    N, B = 40, 64
    L = capi.LandingLib(N, device=0)
    R = lc("rbd").Rbd(L)
    P, X0, _, _ = Pm.make_batch(B, N, 0.6, seed=8)
    r = L.solve_host(P, X0)
    xs = torch.tensor(r["x"], device="cuda")
    # the SRBM NLP uses the ZYX rotation, the kinodynamic model XYZ Euler angles; the screen is meaningful for small
    # roll / yaw -- here it is exercised as is
    s = R.kinodynamic_screen(N, xs)
    torch.cuda.synchronize()
    jp = s["jpos"].cpu().numpy(); ratio = s["torque_ratio_max"].cpu().numpy(); fkb = s["fk_err_max"].cpu().numpy()
    assert np.isfinite(jp).all() and np.isfinite(ratio).all() and (ratio >= 0).all()
    b = int(np.argmin(fkb))
    Xs, Us = Pm.split_solution(N, r["x"][b])
    worst = 0.0; tr = 0.0
    for k in range(N):
        fk, err, tau = ro.kinodyn_rows(Xs[:6, k], Us[:12, k], Us[12:, k], jp[b, k])
        worst = max(worst, np.abs(err).max()); tr = max(tr, (np.abs(tau) / lc("rbd").TAU_MAX).max())
    assert abs(worst - fkb[b]) < 1e-9 and abs(tr - ratio[b]) < 1e-9 * max(1.0, tr)
    L.close()
