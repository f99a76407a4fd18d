"""CPU tests (no GPU): the oracle restatement against the reference's golden data.

Bars: <=1e-12 absolute (values are O(1..20)) against the outputs of the reference's generated C
stored in tests/golden/n20_eval.npz; exact equality of the CCS patterns with casadi_s4/casadi_s5.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, lc

TOL = 1e-12


@pytest.fixture(scope="module")
def O20(oracle_mod):
    return oracle_mod.Oracle(20)


def test_sizes(oracle_mod):
    for N, want in ((20, (732, 2092, 354, 7664, 3780)), (40, (1452, 4172, 614, 15364, 7560))):
        O = oracle_mod.Oracle(N)
        assert (O.nx, O.ng, O.np_, O.nnz_jac, O.nnz_hess) == want  # landingCtrller_IPOPT.c:66,53526,94013


def test_patterns_equal_reference(O20):
    d = np.load(os.path.join(GOLDEN, "n20_patterns.npz"))
    ci, r = O20.pattern_jac()
    assert np.array_equal(ci, d["jac_colind"]) and np.array_equal(r, d["jac_row"])
    ci, r = O20.pattern_hess()
    assert np.array_equal(ci, d["hess_colind"]) and np.array_equal(r, d["hess_row"])
    assert np.all(np.diff(ci) >= 0)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_callbacks_match_reference_fixture(O20, case):
    d = np.load(os.path.join(GOLDEN, "n20_eval.npz"))
    g = lambda k: d[f"c{case}_{k}"]
    x, p, lam, lam_f = g("x"), g("p"), g("lam_g"), float(g("lam_f"))
    assert abs(O20.f(x, p) - g("f")) <= TOL * max(1, abs(g("f")))
    f, gf = O20.grad_f(x, p)
    assert np.max(np.abs(gf - g("grad_f"))) <= TOL
    assert np.max(np.abs(O20.g(x, p) - g("g"))) <= TOL
    gg, jac = O20.jac_g(x, p)
    assert np.max(np.abs(gg - g("g"))) <= TOL
    assert np.max(np.abs(jac - g("jac"))) <= TOL
    assert np.max(np.abs(O20.hess_l(x, p, lam_f, lam) - g("hess"))) <= 10 * TOL
    f2, g2, gx, gp = O20.grad(x, p, lam_f, lam)
    assert np.max(np.abs(gx - g("grad_gamma_x"))) <= 10 * TOL
    assert np.max(np.abs(gp - g("grad_gamma_p"))) <= 100 * TOL


def test_against_reference_library_when_present(oracle_mod, O20):
    """Direct comparison with oracle/_ref (the reference's own generated C) on fresh seeds."""
    try:
        R = oracle_mod.RefOracle()
    except FileNotFoundError:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(7)
    for _ in range(3):
        x = rng.normal(size=732) * 0.7
        p = rng.uniform(0.3, 2.0, size=354)
        lam = rng.normal(size=2092)
        assert np.max(np.abs(O20.g(x, p) - R.g(x, p))) <= TOL
        assert np.max(np.abs(O20.jac_g(x, p)[1] - R.jac_g(x, p)[1])) <= TOL
        assert np.max(np.abs(O20.hess_l(x, p, 1.3, lam) - R.hess_l(x, p, 1.3, lam))) <= 10 * TOL
        a, b = O20.grad(x, p, 1.3, lam), R.grad(x, p, 1.3, lam)
        assert np.max(np.abs(a[2] - b[2])) <= 10 * TOL and np.max(np.abs(a[3] - b[3])) <= 100 * TOL


def test_golden_trajectory_known_answer(O20):
    """test_scripts/1.5msDrop30Pitch.mat is feasible and has f<=2e-5 under the reference NLP."""
    d = np.load(os.path.join(GOLDEN, "n20_golden_1p5ms30pitch.npz"))
    x, p = d["x"], d["p"]
    assert abs(O20.f(x, p) - d["f_ref"]) < 1e-15 and d["f_ref"] < 2e-5
    g = O20.g(x, p)
    assert np.max(np.abs(g - d["g_ref"])) <= TOL
    lb, ub = O20.bounds(p)
    viol = np.maximum(np.maximum(lb - g, g - ub), 0)
    assert viol.max() < 3e-5


def test_n40_stored_solutions_are_feasible(oracle_mod):
    """data/*.mat N=40 solutions satisfy the N-generic restatement (IPOPT constr_viol_tol is 1e-3)."""
    P, Cn = lc("problem"), lc("constants")
    d = np.load(os.path.join(GOLDEN, "n40_golden.npz"))
    O = oracle_mod.Oracle(40, kin_box=(0.05, 0.05, 0.27))  # generate_quadruped_SRBM_CCC.m:169-171
    mass, Ib, Ibi = Cn.robot_constants()
    for x in d["x"]:
        X = x[:12 * 41].reshape(12, 41, order="F")
        p = P.pack_params(40, np.zeros((12, 41)), np.full(40, 0.015), [-10, -10, .15, -10, -10, -10], [10, 10, 1, 10, 10, 10],
                          [-10, -10, -10, -40, -40, -40], [10, 10, 10, 40, 40, 40], X[:6, 0], X[6:, 0],
                          [-10, -10, .15, -.1, -.1, -10], [10, 10, 5, .1, .1, 10], [-10, -10, -10, -40, -40, -40],
                          [10, 10, 10, 40, 40, 40], [0, 0, 100, 100, 100, 0, 10, 10, 10, 10, 10, 10], 1.0, .35, 250., mass, Ib, Ibi)  # eval_SRBM_CCC.m:29-56
        g = O.g(x, p)
        lb, ub = O.bounds(p)
        viol = np.maximum(np.maximum(lb - g, g - ub), 0)
        eq = lb == ub
        assert viol[eq].max() < 1e-4 and viol[~eq].max() < 1e-3


def test_n40_finite_differences(oracle_mod):
    """N=40 has no generated C in the reference: check J and H of the restatement by central differences."""
    O = oracle_mod.Oracle(40)
    P = lc("problem")
    pb, x0b, _, _ = P.make_batch(1, 40, 0.6, seed=3)
    p, x = pb[0], x0b[0].copy()
    rng = np.random.default_rng(5)
    x[12 * 41 + 12:] += 0  # keep refs
    x += rng.normal(size=x.size) * 0.05
    lam = rng.normal(size=O.ng)
    g0, jac = O.jac_g(x, p)
    ci, r = O.pattern_jac()
    hci, hr = O.pattern_hess()
    hess = O.hess_l(x, p, 1.0, lam)
    h = 1e-6
    for j in rng.choice(O.nx, size=40, replace=False):
        e = np.zeros(O.nx); e[j] = h
        col = (O.g(x + e, p) - O.g(x - e, p)) / (2 * h)
        dense = np.zeros(O.ng); dense[r[ci[j]:ci[j + 1]]] = jac[ci[j]:ci[j + 1]]
        assert np.max(np.abs(col - dense)) < 2e-7
        gp_, gm_ = O.grad(x + e, p, 1.0, lam)[2], O.grad(x - e, p, 1.0, lam)[2]
        hcol = (gp_ - gm_) / (2 * h)
        dense = np.zeros(O.nx)
        for c in range(O.nx):  # symmetric fill of column j from the upper-triangular CCS
            pass
        # upper part: entries (row<=j) of column j ; lower part: entries (j, c) for c>j
        dense[hr[hci[j]:hci[j + 1]]] = hess[hci[j]:hci[j + 1]]
        for c in range(j + 1, O.nx):
            rows = hr[hci[c]:hci[c + 1]]
            k = np.searchsorted(rows, j)
            if k < rows.size and rows[k] == j:
                dense[c] = hess[hci[c] + k]
        assert np.max(np.abs(hcol - dense)) < 5e-6 * max(1.0, np.max(np.abs(dense)))


def test_constants_match_survey():
    Cn = lc("constants")
    want = json.load(open(os.path.join(GOLDEN, "constants.json")))["survey_a15"]
    mass, Ib, Ibi = Cn.robot_constants()
    assert abs(mass - want["mass"]) < 1e-9
    assert np.allclose(Ib, want["Ib"], atol=1e-7) and np.allclose(Ibi, want["Ib_inv"], atol=1e-5)


def test_bounds_rules(O20):
    """Opti canonical forms: equality rows, one-sided rows, parameter-dependent bounds."""
    d = np.load(os.path.join(GOLDEN, "n20_golden_1p5ms30pitch.npz"))
    p = d["p"]
    lb, ub = O20.bounds(p)
    o = O20.param_offsets()
    assert np.array_equal(lb[:12], ub[:12]) and np.array_equal(lb[:6], p[o["q_init"]:o["q_init"] + 6])
    base = 36 + 104 * 3
    assert np.all(lb[base:base + 12] == 0) and np.all(ub[base:base + 12] == 0)
    assert ub[base + 12] == p[o["f_max"]] and lb[base + 12] == 0
    assert ub[base + 17] == 1e-3 and np.isneginf(lb[base + 17])
    assert ub[base + 27] == p[o["l_leg_max"]] ** 2
    last = 36 + 104 * 19
    assert np.all(ub[last + 40:last + 56] == 0) and np.all(np.isneginf(lb[last + 40:last + 56]))
