"""CPU tests of the kinodynamic refinement solve (no GPU):
  * the batch form of the oracle's rows (numpy over a leading axis, complex-step Jacobian) equals the scalar restatement that the reference's
    stored solutions pin (tests/test_n1_rows.py);
  * landing_kinodyn_bounds (pure host code of the product library) equals the Python mirror kinodyn.bounds;
  * the solver kernels, compiled for the host through tests/emu, re-solve a member that an MI355X solved (tests/golden/n1_kd_solved.npz,
    written by tests/make_golden_kd.py) from a perturbed copy of its solution and end at a KKT point <= 1e-6 under the ORACLE's functions;
    the stored GPU solution itself passes the same certificate;
  * the presolve certificate (a fixed-variable row of the first interval violated) and the CCS pattern export."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")
N = 20


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "emu"], check=True, capture_output=True)
    L = lc("capi").LandingLib(N, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so"))
    return L, lc("rbd").Rbd(L)


def _consts():
    mass, Ib, Ibi = lc("constants").robot_constants()
    return mass, np.asarray(Ib), np.asarray(Ibi), lc("problem").REFERENCE_DT_GRID


def _certify(x, lam, lb, ub, cost, mu=0.75):
    from oracle import kinodyn_oracle as ko
    kd = lc("kinodyn")
    mass, Ib, Ibi, dt = _consts()
    x = np.atleast_2d(x); lam = np.atleast_2d(lam)
    gf = np.array([kd.terminal_cost(x[b], N, np.atleast_2d(cost)[b][12:], np.atleast_2d(cost)[b][:12])[1] for b in range(x.shape[0])])
    return ko.kkt_batch(x, lam, N, dt, mass, Ib, Ibi, mu, np.atleast_2d(lb), np.atleast_2d(ub), gf)


def test_batch_oracle_equals_scalar_oracle():
    from oracle import kinodyn_oracle as ko
    mass, Ib, Ibi, dt = _consts()
    rng = np.random.default_rng(3)
    nx, ng = ko.nlp_dims(N)
    X = rng.normal(size=(2, nx)) * 0.3
    for b in range(2):
        assert np.abs(ko.nlp_g(X[b], N, dt, mass, Ib, Ibi, 0.75) - ko.nlp_g_batch(X[b:b + 1], N, dt, mass, Ib, Ibi, 0.75)[0]).max() <= 1e-14
    lam = rng.normal(size=(1, ng))
    J = ko.nlp_jacobian(X[0], N, dt, mass, Ib, Ibi, 0.75)           # Richardson-extrapolated differences of the scalar rows
    gl = ko.grad_lagrangian_batch(X[:1], lam, N, dt, mass, Ib, Ibi, 0.75, np.zeros((1, nx)))[0]      # complex step of the batch rows
    assert np.abs(J.T @ lam[0] - gl).max() <= 1e-9 * max(1.0, np.abs(gl).max())


def test_bounds_entry_point_equals_python_mirror():
    kd, P = lc("kinodyn"), lc("problem")
    lib = lc("capi").LandingLib.__new__(lc("capi").LandingLib)
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "all"], check=True, capture_output=True)
    import ctypes as C
    raw = lc("capi").load(os.path.join(PKG, "liblanding_mi355x.so"))
    _, _, q, qd = P.make_batch(5, N, 0.6, seed=7, consts=P.production_constants("main"), dt_grid="reference")
    B = 5
    dp = C.POINTER(C.c_double)
    ci = np.array([kd.c_init_of(q[b]) for b in range(B)]); kb = np.array([kd.kin_box_of(q[b][3:6], qd[b][3:6]) for b in range(B)])
    rep = lambda v: np.ascontiguousarray(np.tile(np.asarray(v, float), (B, 1)))
    args = [np.ascontiguousarray(q), np.ascontiguousarray(qd), ci, rep([-10, -10, 0.075, -10, -10, -10]), rep([-10, -10, 0.15, -0.1, -0.1, -10]), rep([10, 10, 5, 0.1, 0.1, 10]),
            rep([-10, -10, -10, -.5, -.5, -.5]), rep([10, 10, 10, .5, .5, .5]), rep(kd.JPOS_MIN), rep(kd.JPOS_MAX), np.ascontiguousarray(kb), np.full(B, 0.4)]
    ng = kd.dims(N)[1]
    lb = np.zeros((B, ng)); ub = np.zeros((B, ng))
    raw.landing_kinodyn_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p] + [dp] * 14
    assert raw.landing_kinodyn_bounds(N, B, None, *[a.ctypes.data_as(dp) for a in args], lb.ctypes.data_as(dp), ub.ctypes.data_as(dp)) == 0
    for b in range(B):
        l0, u0 = kd.bounds(N, q[b], qd[b], ci[b], kb[b])
        assert np.array_equal(lb[b], l0) and np.array_equal(ub[b], u0)
    del lib


def test_gpu_solutions_pass_the_oracle():
    kd = lc("kinodyn")
    d = np.load(os.path.join(GOLDEN, "n1_kd_solved.npz"))
    mass, Ib, Ibi, dt = _consts()
    prob = [kd.member_problem(N, d["q_init"][b], d["qd_init"][b], d["x_srbm"][b]) for b in range(3)]
    lb, ub, cost, _ = (np.array([p[i] for p in prob]) for i in range(4))
    # (a) what the MI355X returned is a KKT point <= 1e-6 under the oracle's rows and complex-step Jacobian
    k_gpu = _certify(d["x"], d["lam_g"], lb, ub, cost)
    assert k_gpu.max() <= 1e-6 * 1.0001, k_gpu
    assert np.allclose(k_gpu, d["kkt"], rtol=1e-4, atol=1e-10)          # ... and the kernel's own report is that residual


def test_emulated_kernels_solve_a_short_horizon_member(emu):
    """the whole iteration -- derivative kernels, condensation on the (emulated) matrix cores, Riccati sweep with the joint angles eliminated
    inside the stage, filter line search -- from a cold start on a 6-interval member (dt = 50 ms, a gentle drop; the full 20-interval grid takes
    minutes under the fiber emulation): a KKT point <= 1e-6 under the oracle's rows / complex-step Jacobian, f* = 0 (reachable terminal reference)"""
    from oracle import kinodyn_oracle as ko
    kd, P = lc("kinodyn"), lc("problem")
    mass, Ib, Ibi, _ = _consts()
    Ns, dtv = 6, np.full(6, 0.05)
    Ls = lc("capi").LandingLib(Ns, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so")); Rs = lc("rbd").Rbd(Ls)
    q = np.array([0, 0, 0.0, 0.05, 0.15, -0.05]); qd = np.array([0.1, -0.1, 0.05, 0.2, -0.1, -1.0])
    q[2] = 0.35 + abs(min((kd.rot_xyz(q[3:6]) @ np.array([sx * 0.19, sy * 0.1, 0.0]))[2] for sx in (1, -1) for sy in (1, -1))) + abs(dtv[0] * qd[5])
    _, x0s, _, _ = P.make_member(Ns, 0.3, q, qd, P.production_constants("main"), dtv)
    lb, ub, cost, x0 = kd.member_problem(Ns, q, qd, x0s)
    o = Rs.kinodyn_default_opts(); o.max_iter = 80
    s = Rs.kinodyn_solve_host(Ns, lb, ub, cost, x0, dtv, mass, Ib, Ibi, 0.75, o)
    assert s["status"][0] == 0 and s["iters"][0] <= 60, (s["status"], s["iters"], s["kkt"])
    gf = kd.terminal_cost(s["x"][0], Ns, cost[12:], cost[:12])[1]
    k = ko.kkt_batch(s["x"], s["lam_g"], Ns, dtv, mass, Ib, Ibi, 0.75, lb[None], ub[None], gf[None])
    assert k.max() <= 1e-6 * 1.0001, k
    assert np.allclose(k[0], s["kkt"][0], rtol=1e-4, atol=1e-10)          # the kernel's report is the oracle's residual
    assert s["f"][0] <= 1e-7
    assert np.array_equal(s["x"][0][:12], np.concatenate([q, qd]))        # initial conditions met exactly
    Ls.close()


def test_portfolio_reports_the_first_member_of_a_family_that_converges():
    """landing_solver_opts::kd_clone_after (round 5): the member still iterating after 4 rounds is posed again in three clone slots under other option
    sets; whichever member of the family converges first is reported under the original's index -- a KKT point <= 1e-6 under the oracle like any other.
    A clone time the solve never reaches leaves every bit of the result alone."""
    from oracle import kinodyn_oracle as ko
    kd, P = lc("kinodyn"), lc("problem")
    mass, Ib, Ibi, _ = _consts()
    Ns, dtv = 6, np.full(6, 0.05)
    Ls = lc("capi").LandingLib(Ns, lib_path=os.path.join(ROOT, "tests", "emu", "liblanding_emu.so")); Rs = lc("rbd").Rbd(Ls)
    q = np.array([0, 0, 0.0, 0.05, 0.15, -0.05]); qd = np.array([0.1, -0.1, 0.05, 0.2, -0.1, -1.0])
    q[2] = 0.35 + abs(min((kd.rot_xyz(q[3:6]) @ np.array([sx * 0.19, sy * 0.1, 0.0]))[2] for sx in (1, -1) for sy in (1, -1))) + abs(dtv[0] * qd[5])
    _, x0s, _, _ = P.make_member(Ns, 0.3, q, qd, P.production_constants("main"), dtv)
    lb, ub, cost, x0 = kd.member_problem(Ns, q, qd, x0s)
    res = {}
    for name, after, lim in (("off", 0, 6), ("never", 400, 6), ("early", 4, 80)):      # (the first two stop at the iteration limit: the emulation is slow)
        o = Rs.kinodyn_default_opts(); o.max_iter = lim; o.feas_phase = 0; o.kd_clone_after = after; o.kd_clone_max = 1; o.kd_clone_iter = 80
        res[name] = Rs.kinodyn_solve_host(Ns, lb, ub, cost, x0, dtv, mass, Ib, Ibi, 0.75, o)
    for k in ("x", "lam_g", "kkt", "status", "iters"):
        assert np.array_equal(res["off"][k], res["never"][k]), k
    s = res["early"]
    assert s["status"][0] == 0 and s["iters"][0] <= 60, (s["status"], s["iters"])
    gf = kd.terminal_cost(s["x"][0], Ns, cost[12:], cost[:12])[1]
    k = ko.kkt_batch(s["x"], s["lam_g"], Ns, dtv, mass, Ib, Ibi, 0.75, lb[None], ub[None], gf[None])
    assert k.max() <= 1e-6 * 1.0001, k
    assert np.allclose(k[0], s["kkt"][0], rtol=1e-4, atol=1e-10)
    assert np.array_equal(s["x"][0][:12], np.concatenate([q, qd]))
    assert res["off"]["status"][0] == 1 and res["off"]["iters"][0] == 6
    Ls.close()


def test_presolve_certificate_and_patterns(emu):
    L, R = emu
    kd, P = lc("kinodyn"), lc("problem")
    mass, Ib, Ibi, dt = _consts()
    # a steep initial pitch with a slow drop: the nominal stance under the hips violates the kinematic box of the FIRST interval, which only
    # holds fixed variables (landing_optimization.m:89-91,157-164,232-236) -- reported at once, no iteration
    q = np.array([0, 0, 0.6, 0.0, 0.9, 0.0]); qd = np.array([0, 0, 0, 0, 0, -0.6])
    _, x0s, _, _ = P.make_member(N, 0.6, q, qd, P.production_constants("main"), P.REFERENCE_DT_GRID)
    lb, ub, cost, x0 = kd.member_problem(N, q, qd, x0s)
    s = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, 0.75)
    assert s["status"][0] == 3 and s["iters"][0] == 0
    from oracle import kinodyn_oracle as ko
    g = ko.nlp_g(s["x"][0], N, dt, mass, Ib, Ibi, 0.75)
    rows = 48 + 16 + 15 * np.repeat(np.arange(4), 5) + np.tile([0, 8, 9, 10, 11], 4)
    viol = np.maximum(np.maximum(lb - g, g - ub), 0.0)
    assert viol[rows].max() > 1e-3 and abs(s["kkt"][0, 0] - viol.max()) <= 1e-12
    # CCS patterns from the derivative kernels: column counts, symmetry of use, sizes
    cj, rj = R.kinodyn_pattern(N, 0)
    ch, rh = R.kinodyn_pattern(N, 1)
    nx, ng = kd.dims(N)
    assert cj.shape == (nx + 1,) and cj[-1] == rj.size and rj.max() < ng and np.all(np.diff(cj) >= 1)
    assert ch[-1] == rh.size and all(rh[ch[c]:ch[c + 1]].max(initial=-1) <= c for c in range(nx))      # upper triangle
    assert (rj.size, rh.size) == (NNZ_JAC, NNZ_HESS)
    # round 6: the solver's own table of a block's non-zeros (the forward sweep forms ds = J dx over these entries only) against the CCS pattern above -- two probes with
    # different random points and constants: per interval exactly the entries of the inequality rows 12.. of jac_g_x
    (rm, cm), (rl, cl) = R.kinodyn_block_nonzeros()
    assert (rm.size, rl.size) == (529, 457) and np.bincount(np.bincount(rm)[12:]).max() >= 1 and np.bincount(rm).max() <= 12
    from_ccs = {k: set() for k in range(N)}
    oJ, oU, BND, NR = 12 * (N + 1), 12 * (N + 1) + 12 * N, 48, 141
    def widx(k, j):
        if j < 12: return 12 * k + j
        if j < 36: return oU + 24 * k + (j - 12)
        if j < 48: return oJ + 12 * k + (j - 36)
        if j < 60: return 12 * (k + 1) + (j - 48)
        return oU + 24 * (k + 1) + (j - 60) if k + 1 < N else -1
    col_of = {}
    for k in range(N):
        for j in range(72):
            if widx(k, j) >= 0: col_of[(k, widx(k, j))] = j
    for c in range(nx):
        for r in rj[cj[c]:cj[c + 1]]:
            if r >= BND and (r - BND) % NR >= 12:
                k = (r - BND) // NR
                from_ccs[k].add(((r - BND) % NR, col_of[(k, c)]))
    assert from_ccs[3] == set(zip(rm.tolist(), cm.tolist())) and from_ccs[N - 1] == set(zip(rl.tolist(), cl.tolist()))


NNZ_JAC, NNZ_HESS = 13536, 5720      # N = 20 (three random points agree; the reference ships no generated code of this NLP to compare with)


def test_c_model_builder_equals_python_model_and_gateway_compiles(tmp_path):
    """landing_rbd_model_mc3d (what C / mex callers use) fills the struct exactly as rbd.quad3d_model does; matlab/landing_refine_mex.c compiles
    against the mex.h stub and links to the product library"""
    import ctypes as C
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "all"], check=True, capture_output=True)
    raw = lc("capi").load(os.path.join(PKG, "liblanding_mi355x.so"))
    rbd = lc("rbd")
    m = rbd.RbdModel()
    raw.landing_rbd_model_mc3d.argtypes = [C.POINTER(rbd.RbdModel)]
    raw.landing_rbd_model_mc3d.restype = None
    raw.landing_rbd_model_mc3d(C.byref(m))
    ref = rbd.quad3d_model()
    for name, _ in rbd.RbdModel._fields_:
        a, b = np.ctypeslib.as_array(getattr(m, name)) if hasattr(getattr(m, name), "_length_") else getattr(m, name), \
               np.ctypeslib.as_array(getattr(ref, name)) if hasattr(getattr(ref, name), "_length_") else getattr(ref, name)
        assert np.allclose(a, b, rtol=1e-13, atol=1e-19), name      # (the inertia about the link origin is formed in another order of operations: last-bit differences)
    so = os.path.join(str(tmp_path), "refine_gateway.so")
    subprocess.run(["gcc", "-O1", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-shared", "-Wall", "-Werror", "-Wno-unused-function", "-I", os.path.join(ROOT, "tests", "stubs"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "matlab", "landing_refine_mex.c"), "-o", so, "-L", PKG, "-llanding_mi355x", "-Wl,-rpath," + PKG], check=True)
    assert "mexFunction" in subprocess.run(["nm", "-D", so], capture_output=True, text=True).stdout


def test_kinodyn_bounds_forms_match_the_two_reference_scripts():
    """landing_solve_kinodyn_24[_on] stand for the function generate_landingCtrller_KNITRO.m builds, whose lateral kinematic box is
    kin_box_y = 0.125 + kin_box(2) (:154); main_scripts/landing_optimization.m:150 has 0.10.  landing_kinodyn_form_knitro / _default are those two
    literal sets; landing_kinodyn_bounds (pure host code) with either equals the numpy restatement of the rows (kinodyn.bounds), and the two
    forms differ in the eight lateral-box bounds of every interval only."""
    import ctypes as C
    capi, kd = lc("capi"), lc("kinodyn")
    lib = capi.load()
    N, B = 20, 2

    class Form(C.Structure):
        _fields_ = [("comp_eps", C.c_double), ("slip_eps", C.c_double), ("fk_band", C.c_double), ("kin_box_x0", C.c_double), ("kin_box_y0", C.c_double),
                    ("kin_box_y_in", C.c_double), ("kin_box_z_lo", C.c_double), ("kin_box_z_hi", C.c_double), ("tau_max", C.c_double * 3)]
    fd, fk = Form(), Form()
    lib.landing_kinodyn_form_default(C.byref(fd)); lib.landing_kinodyn_form_knitro(C.byref(fk))
    assert (fd.kin_box_x0, fd.kin_box_y0, fk.kin_box_x0, fk.kin_box_y0) == (0.125, 0.10, 0.125, 0.125)
    for f in ("comp_eps", "slip_eps", "fk_band", "kin_box_y_in", "kin_box_z_lo", "kin_box_z_hi"):
        assert getattr(fd, f) == getattr(fk, f)
    rng = np.random.default_rng(3)
    q = np.column_stack([np.zeros((B, 2)), 0.5 + 0.1 * rng.random(B), 0.2 * rng.normal(size=(B, 3))]); qd = rng.normal(size=(B, 6))
    nx, ng = kd.dims(N)
    dp = C.POINTER(C.c_double)
    arr = lambda v: np.ascontiguousarray(v, float)
    ci = arr([kd.c_init_of(q[b]) for b in range(B)]); kb = arr([kd.kin_box_of(q[b, 3:6], qd[b, 3:6]) for b in range(B)])
    rep = lambda v: arr(np.tile(np.asarray(v, float), (B, 1)))
    ins = [arr(q), arr(qd), ci, rep([-10, -10, 0.075, -10, -10, -10]), rep([-10, -10, 0.15, -0.1, -0.1, -10]), rep([10, 10, 5, 0.1, 0.1, 10]),
           rep([-10, -10, -10, -.5, -.5, -.5]), rep([10, 10, 10, .5, .5, .5]), rep(kd.JPOS_MIN), rep(kd.JPOS_MAX), kb, arr(np.full((B, 1), 0.4))]
    out = {}
    for name, form, y0 in (("default", fd, 0.10), ("knitro", fk, 0.125)):
        lb = np.zeros((B, ng)); ub = np.zeros((B, ng))
        rc = lib.landing_kinodyn_bounds(C.c_int(N), C.c_int(B), C.byref(form), *[a.ctypes.data_as(dp) for a in ins], lb.ctypes.data_as(dp), ub.ctypes.data_as(dp))
        assert rc == 0
        for b in range(B):
            l2, u2 = kd.bounds(N, q[b], qd[b], ci[b], kb[b], kin_box_y0=y0)
            assert np.array_equal(lb[b], l2) and np.array_equal(ub[b], u2), name
        out[name] = (lb, ub)
    diff = (out["default"][0] != out["knitro"][0]) | (out["default"][1] != out["knitro"][1])
    d = lambda i: np.abs(np.where(diff, out["default"][i], 0.0) - np.where(diff, out["knitro"][i], 0.0))
    assert diff.sum() == B * N * 4 and np.allclose((d(0) + d(1))[diff], 0.025)
