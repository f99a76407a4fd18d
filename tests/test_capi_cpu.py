"""CPU tests of the product's C ABI and host logic (no GPU needed):
  * the hipcc-built library loads and exports every symbol include/*.h declares;
  * the CCS patterns it reports equal the reference's casadi_s4/casadi_s5 (golden fixture);
  * the kernels' indexing/emission logic, compiled for the host through tests/emu (the same
    sources, fibers instead of lanes), reproduces the oracle -- no compute call touches a GPU here.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")


@pytest.fixture(scope="session")
def built():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "all", "emu"], check=True, capture_output=True)
    return True


def _declared(header, pat):
    txt = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(pat, txt)))


def test_library_exports_every_declared_symbol(built):
    lib = C.CDLL(os.path.join(PKG, "liblanding_mi355x.so"))
    names = _declared("landing_nlp.h", r"\b(landing_[a-z0-9_]+)\s*\(")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.parametrize("so", ["landingCtrller_IPOPT_mi355x.so", "landingCtrller_IPOPT_N40_mi355x.so", "nlp_quad_SRBM_mi355x.so", "landingCtrller_KNITRO_mi355x.so"])
def test_casadi_dropin_exports_reference_symbol_set(built, so):
    lib = C.CDLL(os.path.join(PKG, so))
    suffixes = ["", "_alloc_mem", "_init_mem", "_free_mem", "_checkout", "_release", "_incref", "_decref", "_n_in",
                "_n_out", "_default_in", "_name_in", "_name_out", "_sparsity_in", "_sparsity_out", "_work"]
    for f in ["nlp", "nlp_f", "nlp_g", "nlp_grad", "nlp_grad_f", "nlp_hess_l", "nlp_jac_g"]:  # landingCtrller_IPOPT.c:10916-10993
        for s in suffixes:
            assert hasattr(lib, f + s), f + s
    # metadata CasADi asserts on (external.cpp:325-363)
    lib.nlp_jac_g_name_out.restype = C.c_char_p; lib.nlp_jac_g_name_out.argtypes = [C.c_longlong]
    assert lib.nlp_jac_g_name_out(1) == b"jac_g_x" and lib.nlp_jac_g_name_out(2) is None
    lib.nlp_hess_l_name_out.restype = C.c_char_p; lib.nlp_hess_l_name_out.argtypes = [C.c_longlong]
    assert lib.nlp_hess_l_name_out(0) == b"hess_gamma_x_x"
    lib.nlp_grad_n_in.restype = C.c_longlong
    assert lib.nlp_grad_n_in() == 4
    sz = (C.c_longlong * 4)()
    assert lib.nlp_work(C.byref(sz, 0), C.byref(sz, 8), C.byref(sz, 16), C.byref(sz, 24)) == 0 and list(sz) == [2, 2, 0, 0]


def test_dropin_sparsity_equals_reference_pattern(built):
    lib = C.CDLL(os.path.join(PKG, "landingCtrller_IPOPT_mi355x.so"))
    d = np.load(os.path.join(GOLDEN, "n20_patterns.npz"))
    for fn, idx, ci_k, r_k, nr, nc in (("nlp_jac_g_sparsity_out", 1, "jac_colind", "jac_row", 2092, 732),
                                       ("nlp_hess_l_sparsity_out", 0, "hess_colind", "hess_row", 732, 732)):
        f = getattr(lib, fn); f.restype = C.POINTER(C.c_longlong); f.argtypes = [C.c_longlong]
        ptr = f(idx)
        assert (ptr[0], ptr[1]) == (nr, nc)
        ci = np.array([ptr[2 + i] for i in range(nc + 1)])
        r = np.array([ptr[3 + nc + i] for i in range(int(ci[-1]))])
        assert np.array_equal(ci, d[ci_k]) and np.array_equal(r, d[r_k])
    f = lib.nlp_f_sparsity_in; f.restype = C.POINTER(C.c_longlong); f.argtypes = [C.c_longlong]
    ptr = f(1)
    assert [ptr[i] for i in range(4)] == [354, 1, 0, 354]   # casadi_s1, landingCtrller_IPOPT.c:60


def test_ccc_dropin_metadata(built):
    """nlp_quad_SRBM_mi355x.so = the NLP of the reference's N=41 script (generate_quadruped_SRBM_CCC.m:340-341): x 1452, its own parameter
    vector p 37N+112 = 1592, g 4172, the Hessian in the extended (running-cost) pattern -- upper triangular, rows sorted"""
    lib = C.CDLL(os.path.join(PKG, "nlp_quad_SRBM_mi355x.so"))
    main = C.CDLL(os.path.join(PKG, "liblanding_mi355x.so"))
    main.landing_nnz_hess_rc.restype = C.c_longlong
    f = lib.nlp_f_sparsity_in; f.restype = C.POINTER(C.c_longlong); f.argtypes = [C.c_longlong]
    assert [f(0)[i] for i in range(2)] == [1452, 1] and [f(1)[i] for i in range(4)] == [1592, 1, 0, 1592]
    h = lib.nlp_hess_l_sparsity_out; h.restype = C.POINTER(C.c_longlong); h.argtypes = [C.c_longlong]
    ptr = h(0)
    nnz = main.landing_nnz_hess_rc(40)
    assert (ptr[0], ptr[1]) == (1452, 1452) and ptr[2 + 1452] == nnz == 7560 + 18 * 40
    ci = np.array([ptr[2 + i] for i in range(1453)]); r = np.array([ptr[3 + 1452 + i] for i in range(nnz)])
    for c in range(1452):
        rows = r[ci[c]:ci[c + 1]]
        assert (np.diff(rows) > 0).all() and (rows <= c).all()
    gp = lib.nlp_grad_sparsity_out; gp.restype = C.POINTER(C.c_longlong); gp.argtypes = [C.c_longlong]
    assert gp(3)[0] == 1592                                      # grad_gamma_p has an entry for every parameter, QX / Qc / Qf / Uref included


def test_no_device_means_loud_failure(built):
    """The product has no CPU path: without a GPU landing_create must fail with a message."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    capi = lc("capi")
    with pytest.raises(RuntimeError, match="no HIP device|landing_create failed"):
        capi.LandingLib(20)


@pytest.fixture(scope="module")
def emu(built):
    return os.path.join(ROOT, "tests", "emu", "liblanding_emu.so")


@pytest.mark.parametrize("N", [20, 40])
def test_emulated_kernels_match_oracle(emu, oracle_mod, N):
    capi = lc("capi")
    O = oracle_mod.Oracle(N)
    lib = capi.LandingLib(N, lib_path=emu)
    assert all(np.array_equal(a, b) for a, b in zip(lib.pattern_jac(), O.pattern_jac()))
    assert all(np.array_equal(a, b) for a, b in zip(lib.pattern_hess(), O.pattern_hess()))
    rng = np.random.default_rng(N)
    B = 2
    x = rng.normal(size=(B, O.nx)) * 0.5; p = rng.uniform(0.5, 1.5, size=(B, O.np_))
    lam = rng.normal(size=(B, O.ng)); lf = rng.uniform(0.5, 2, size=B)
    out = lib.eval_host(x, p, lf, lam)
    for b in range(B):
        f, gf = O.grad_f(x[b], p[b]); g, jac = O.jac_g(x[b], p[b])
        h = O.hess_l(x[b], p[b], lf[b], lam[b]); _, _, gx, gp = O.grad(x[b], p[b], lf[b], lam[b])
        assert abs(out["f"][b] - f) < 1e-12 and np.max(np.abs(out["grad_f"][b] - gf)) < 1e-12
        assert np.max(np.abs(out["g"][b] - g)) < 1e-12 and np.max(np.abs(out["jac"][b] - jac)) < 1e-12
        assert np.max(np.abs(out["hess"][b] - h)) < 1e-11
        assert np.max(np.abs(out["grad_gamma_x"][b] - gx)) < 1e-11 and np.max(np.abs(out["grad_gamma_p"][b] - gp)) < 1e-10


def test_emulated_kernels_match_reference_fixture(emu):
    """the reference's own outputs (tests/golden/n20_eval.npz) through the emulated kernels"""
    capi = lc("capi")
    lib = capi.LandingLib(20, lib_path=emu)
    d = np.load(os.path.join(GOLDEN, "n20_eval.npz"))
    for c in range(3):
        g = lambda k: d[f"c{c}_{k}"]
        out = lib.eval_host(g("x"), g("p"), np.array([float(g("lam_f"))]), g("lam_g"))
        assert np.max(np.abs(out["g"][0] - g("g"))) < 1e-12
        assert np.max(np.abs(out["jac"][0] - g("jac"))) < 1e-12
        assert np.max(np.abs(out["hess"][0] - g("hess"))) < 1e-11
        assert np.max(np.abs(out["grad_gamma_x"][0] - g("grad_gamma_x"))) < 1e-11
        assert np.max(np.abs(out["grad_gamma_p"][0] - g("grad_gamma_p"))) < 1e-10


def test_solver_opts_mirror_matches_the_header():
    """the ctypes mirror of landing_solver_opts (capi.SolverOpts) has the header's field order and types: every default written by
    the C side reads back at the right name (a shifted field would show up as a wrong value), both for the cold and the warm set"""
    import ctypes as C
    capi = lc("capi")
    lib = capi.load()
    o = capi.SolverOpts(); lib.landing_solver_opts_default(C.byref(o))
    want = dict(tol=1e-6, max_iter=3000, mu_init=0.0, bound_push=0.0, bound_frac=0.1, kappa_eps=0.0, kappa_mu=0.2, theta_mu=0.0, max_soc=0, max_resets=8,
                reset_du=1e9, stage_local_reg=0, sticky_delta=0, restart_period=75, dispatch_order=1, delta_init=1e-4, delta_inc_first=10.0, delta_inc=4.0,
                delta_dec=0.5, tau_min=0.9, alpha_fallback=1e-2, reset_delta=1e5, clip_k=4, clip_until=0.03, theta_floor=30.0, fresh_restart=9,
                dual_step_cap=1.0, slack_corr=0.9, watchdog=3, barrier_smax=1.0, factor_fp32=0, feas_phase=1, feas_rho=1000.0, feas_cert=1e-4, delta_floor=3e-4, jam_clip=2, stag_relief=3, feas_jam=8, feas_stat=25,
                kd_clone_after=0, kd_clone_max=0, kd_clone_iter=0,
                feas_back=0.2, feas_max=3, feas_delta_dec=0.1, feas_ret_push=0.01, feas_ret_mu=0.01, feas_resume=1, feas_polish=1e-8)
    assert {n: getattr(o, n) for n, _ in capi.SolverOpts._fields_} == want
    w = capi.SolverOpts(); lib.landing_solver_opts_warm(C.byref(w))
    assert (w.bound_push, w.bound_frac, w.mu_init, w.restart_period, w.max_iter, w.clip_k, w.fresh_restart, w.factor_fp32) == (1e-4, 1e-4, 1e-4, 0, 14, 0, 0, 0)
    assert (w.dual_step_cap, w.slack_corr, w.watchdog, w.barrier_smax, w.theta_floor, w.feas_phase) == (1.0, 0.9, 3, 1.0, 30.0, 0)
    # the header declares the fields in the same order
    import re
    hdr = open(os.path.join(ROOT, "include", "landing_nlp.h")).read()
    body = hdr[hdr.index("typedef struct {\n  double tol;"):hdr.index("} landing_solver_opts;")]
    names = re.findall(r"^  (?:int|double) (\w+);", body, flags=re.M)
    assert names == [n for n, _ in capi.SolverOpts._fields_], names
