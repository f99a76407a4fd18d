"""The reference's N=41 script declares its cost weights as Opti PARAMETERS (generate_quadruped_SRBM_CCC.m:67-68: QX, QN, Qc, Qf) and uses the
force part of Uref in the running cost (:81-89), so its NLP has its own parameter vector, np = 37N + 112 (VERDICT r2 "missing" 6):
    p = [Xref | Uref | dt | q_min .. qd_term_max | QX | QN | Qc | Qf | mu l_leg_max f_max mass | Ib | Ib_inv]
`landing_form.run_cost = 2` selects it.  Pinned here:
  * the oracle's grad_gamma_p against central differences of gamma = lam_f f + lam_g' g in EVERY entry of p (incl. Uref, QX, Qc, Qf);
  * f, grad f, grad_gamma_x / p and the extended-pattern Hessian of the kernels (host emulation on CPU, MI355X with -m gpu) against the oracle;
  * the same problem posed with constants of the context (run_cost 1) and through p (run_cost 2) is solved to the same bits;
  * landing_pack_args25 (the script's 25-argument solver function, analysis/eval_SRBM_CCC.m:72-78) = the Python mirror."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")
RC = dict(QX=[0.3, 0.2, 10, 1, 1, 0.4, .1, .2, .1, .3, .1, .2], Qc=[1.0, 0.8, 0.5], Qf=[1e-4, 2e-4, 1e-3], f_ref=[0.5, -0.25, 20.0])
KB = (0.05, 0.05, 0.27)


def _problem(N, B, seed):
    Pm = lc("problem")
    P, X0, _, _ = Pm.make_batch(B, N, 0.6, seed=seed)
    rng = np.random.default_rng(seed)
    Pc = np.zeros((B, Pm.n_p_ccc(N)))
    for b in range(B):
        Uref = X0[b][12 * (N + 1):].reshape(24, N, order="F").copy()
        Uref[12:] = np.tile(RC["f_ref"], 4)[:, None] + (0.3 * rng.normal(size=(12, N)) if b else 0.0)     # member 0: the per-axis constant of run_cost 1
        Pc[b] = Pm.ccc_from_ipopt_params(N, P[b], Uref, RC["QX"], RC["Qc"], RC["Qf"])
    return P, Pc, X0


def test_oracle_grad_gamma_p_all_entries(oracle_mod):
    N = 5
    O2 = oracle_mod.Oracle(N, kin_box=KB, run_cost=RC, ccc_params=True)
    O1 = oracle_mod.Oracle(N, kin_box=KB, run_cost=RC)
    P, Pc, X0 = _problem(N, 2, 3)
    assert O2.np_ == 37 * N + 112 and Pc.shape[1] == O2.np_
    rng = np.random.default_rng(0)
    x = X0[1] + 0.05 * rng.normal(size=X0[1].shape); lam = rng.normal(size=O2.ng); lam_f = 0.7
    # member 0 holds exactly the constants of the run_cost-1 form: same objective, same gradient
    assert abs(O2.f(X0[0] + 0.01, Pc[0]) - O1.f(X0[0] + 0.01, P[0])) <= 1e-12 * max(1.0, abs(O1.f(X0[0] + 0.01, P[0])))
    f, g, gx, gp = O2.grad(x, Pc[1], lam_f, lam)
    gam = lambda p_: lam_f * O2.f(x, p_) + lam @ O2.g(x, p_)
    o = O2.param_offsets()
    fd = np.zeros_like(gp)
    for i in range(O2.np_):
        h = 1e-6 * max(1.0, abs(Pc[1][i]))
        pp, pm = Pc[1].copy(), Pc[1].copy(); pp[i] += h; pm[i] -= h
        fd[i] = (gam(pp) - gam(pm)) / (2 * h)
    skip = set()      # bounds enter lbg / ubg only (not g): zero gradient on both sides, compared anyway
    err = np.abs(fd - gp) / np.maximum(1.0, np.abs(gp))
    assert err.max() <= 2e-6, (int(err.argmax()), fd[err.argmax()], gp[err.argmax()])
    for name, n in (("QX", 12), ("Qc", 3), ("Qf", 3)):
        assert (np.abs(gp[o[name]:o[name] + n]) > 0).all(), name           # the new entries are really there
    U = gp[o["Uref"]:o["Uref"] + 24 * N].reshape(24, N, order="F")
    assert not U[:12].any() and (np.abs(U[12:]) > 0).all()                # foot part inactive, force part active


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "all", "emu"], check=True, capture_output=True)
    return os.path.join(ROOT, "tests", "emu", "liblanding_emu.so")


def _check_eval(L, O, x, Pc, lam, lam_f):
    B = x.shape[0]
    e = L.eval_host(x, Pc, lam_f=np.full(B, lam_f), lam_g=lam, want=("f", "grad_f", "grad_gamma_x", "grad_gamma_p"))
    for b in range(B):
        f, g, gx, gp = O.grad(x[b], Pc[b], lam_f, lam[b])
        assert abs(e["f"][b] - O.f(x[b], Pc[b])) <= 1e-12 * max(1.0, abs(e["f"][b]))
        assert np.allclose(e["grad_f"][b], O.grad_f(x[b], Pc[b])[1], rtol=1e-11, atol=1e-12)
        assert np.allclose(e["grad_gamma_x"][b], gx, rtol=1e-10, atol=1e-11)
        assert np.allclose(e["grad_gamma_p"][b], gp, rtol=1e-10, atol=1e-11), np.abs(e["grad_gamma_p"][b] - gp).argmax()
    h = L.hess_rc_host(x, Pc, np.full(B, lam_f), lam)
    for b in range(B):
        assert np.allclose(h[b], O.hess_l_rc(x[b], Pc[b], lam_f, lam[b]), rtol=1e-11, atol=1e-12)


def test_emulated_kernels_follow_oracle_and_pack25(emu, oracle_mod):
    capi, Pm = lc("capi"), lc("problem")
    N, B = 6, 3
    O = oracle_mod.Oracle(N, kin_box=KB, run_cost=RC, ccc_params=True)
    L = capi.LandingLib(N, lib_path=emu, kin_box=KB, run_cost=RC, ccc_params=True)
    assert L.np_ == 37 * N + 112 == L.lib.landing_np_ccc(N)
    P, Pc, X0 = _problem(N, B, 5)
    rng = np.random.default_rng(1)
    x = X0 + 0.03 * rng.normal(size=X0.shape); lam = rng.normal(size=(B, L.ng))
    _check_eval(L, O, x, Pc, lam, 0.6)
    # the script's 25 arguments, MATLAB-shaped, packed by the C entry point = the Python mirror
    o = Pm.param_offsets(N)
    col = lambda name, n: P[:, o[name]:o[name] + n].T.copy()
    args = dict(Xref=P[:, :12 * (N + 1)].T.reshape(12, N + 1, B, order="F"), Uref=np.stack([Pc[b][O.param_offsets()["Uref"]:][:24 * N].reshape(24, N, order="F") for b in range(B)], axis=2),
                dt=col("dt", N).reshape(1, N, B), QX=np.tile(np.array(RC["QX"])[:, None], (1, B)), Qc=np.tile(np.array(RC["Qc"])[:, None], (1, B)), Qf=np.tile(np.array(RC["Qf"])[:, None], (1, B)),
                QN=col("QN", 12), x0=X0.T.copy(), mu=col("mu", 1), l_leg_max=col("l_leg_max", 1), f_max=col("f_max", 1), mass=col("mass", 1), Ib=col("Ib", 3), Ib_inv=col("Ib_inv", 3), c_init=None)
    for n in ("q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "q_term_min", "q_term_max", "qd_term_min", "qd_term_max"):
        args[n] = col(n, 6)
    a, keep, b = capi.matlab_args25(N, args)
    p = np.zeros((B, L.np_))
    assert b == B and L.lib.landing_pack_args25(N, B, C.byref(a), p.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert np.array_equal(p, Pc)
    # ... and solving through them = solving the packed p; the 21-argument entry point refuses this context
    o_ = L.default_opts(); o_.max_iter = 3
    r0 = L.solve_host(Pc, X0, o_); r1 = L.solve_args25(args, o_)
    for k in ("x", "f", "lam_g", "status", "iters", "kkt"):
        assert np.array_equal(r0[k], r1[k]), k
    with pytest.raises(RuntimeError, match="25 arguments"):
        L.solve_args21({n: args.get(n) for n in capi.ARGS21}, o_)
    L.close()


def test_emulated_solver_same_bits_as_context_constants(emu):
    """member 0 of _problem carries exactly the run_cost-1 constants: weights read from p or from the context give the same iterates"""
    capi = lc("capi")
    N = 6
    P, Pc, X0 = _problem(N, 1, 7)
    L1 = capi.LandingLib(N, lib_path=emu, kin_box=KB, run_cost=RC)
    L2 = capi.LandingLib(N, lib_path=emu, kin_box=KB, run_cost=RC, ccc_params=True)
    o = L1.default_opts(); o.max_iter = 40
    a = L1.solve_host(P, X0, o); b = L2.solve_host(Pc, X0, o)
    assert a["iters"][0] > 5
    for k in ("x", "f", "lam_g", "status", "iters", "kkt"):
        assert np.array_equal(a[k], b[k]), k
    L1.close(); L2.close()


@pytest.mark.gpu
def test_gpu_ccc_params_eval_and_solve(oracle_mod):
    """N = 40 (the reference's N=41 script) on the MI355X: function layer against the oracle in every output incl. the new grad_gamma_p entries;
    the solver reaches KKT <= 1e-6 under the oracle's functions with the weights read from p, same bits as with context constants"""
    capi = lc("capi")
    N, B = 40, 8
    O = oracle_mod.Oracle(N, kin_box=KB, run_cost=RC, ccc_params=True)
    L = capi.LandingLib(N, device=0, kin_box=KB, run_cost=RC, ccc_params=True)
    L1 = capi.LandingLib(N, device=0, kin_box=KB, run_cost=RC)
    P, Pc, X0 = _problem(N, B, 11)
    rng = np.random.default_rng(2)
    x = X0 + 0.02 * rng.normal(size=X0.shape); lam = rng.normal(size=(B, L.ng))
    _check_eval(L, O, x, Pc, lam, 0.8)
    r = L.solve_host(Pc, X0)
    ok = r["status"] == 0
    assert ok.sum() >= B - 1, r["status"]
    for b in np.nonzero(ok)[0]:
        assert O.kkt(r["x"][b], Pc[b], r["lam_g"][b]).max() <= 1e-6 * 1.0001
    r1 = L1.solve_host(P[:1], X0[:1])
    assert np.array_equal(r1["x"][0], r["x"][0]) and r1["iters"][0] == r["iters"][0]
    L.close(); L1.close()
