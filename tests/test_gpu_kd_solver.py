"""GPU tests of the kinodynamic refinement solve (landing_optimization.m:38-201,300-322,360-376), through the C ABI:
the production callers' pipeline -- SRBM solve (landing_solve_batch, N = 20, production grid) -> its solution as the initial guess of the
refinement NLP -> landing_solve_kinodyn_24 -- on 1024 drop states, EVERY returned member re-certified under the oracle's rows and
complex-step Jacobian (oracle/kinodyn_oracle.py, pinned by the reference's stored kinodynamic solutions, tests/test_n1_rows.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, lc

pytestmark = pytest.mark.gpu
N = 20
KKT_TOL = 1e-6


@pytest.fixture(scope="module")
def ctx():
    L = lc("capi").LandingLib(N, device=0)
    R = lc("rbd").Rbd(L)
    yield L, R
    L.close()


def _consts():
    mass, Ib, Ibi = lc("constants").robot_constants()
    return mass, np.asarray(Ib), np.asarray(Ibi), lc("problem").REFERENCE_DT_GRID


def _certify(x, lam, lb, ub, cost, dt, mu):
    from oracle import kinodyn_oracle as ko
    kd = lc("kinodyn")
    mass, Ib, Ibi, _ = _consts()
    out = np.zeros((x.shape[0], 3))
    for lo in range(0, x.shape[0], 128):      # (chunks bound the size of the complex-step arrays)
        sl = slice(lo, lo + 128)
        gf = np.array([kd.terminal_cost(x[b], N, cost[b][12:], cost[b][:12])[1] for b in range(*sl.indices(x.shape[0]))])
        out[sl] = ko.kkt_batch(x[sl], lam[sl], N, dt, mass, Ib, Ibi, mu, lb[sl], ub[sl], gf)
    return out


def test_refinement_of_1024_srbm_solutions_every_member_decided_and_certified(ctx):
    """1024 drop states of the production sampling law (landing_optimization.m:207-218), f_max / grid of that caller.  Every member must be
    DECIDED: a KKT point <= 1e-6 (unscaled, under the oracle -- every converged member is re-certified), or the presolve certificate: a row
    of the first interval that holds FIXED variables only (the nominal stance under the hips of a steep initial attitude against the
    velocity-dependent kinematic box, :157-164 with :232-236,249-251) is violated, which no solver can repair; at most 0.5 % undecided.
    Measured on an MI355X (round 4): seeds 7 / 8 / 9: 834 + 190 + 0, 865 + 159 + 0, 860 + 163 + 1; f* <= 1e-9 for every converged member."""
    L, R = ctx
    P, kd = lc("problem"), lc("kinodyn")
    from oracle import kinodyn_oracle as ko
    B = 1024
    consts = P.production_constants("main")
    Pp, X0, q, qd = P.make_batch(B, N, 0.6, seed=20211, consts=consts, dt_grid="reference", law="main")
    srbm = L.solve_host(Pp, X0)
    assert (srbm["status"] == 0).mean() >= 0.99
    mass, Ib, Ibi, dt = _consts()
    args = kd.make_args24(N, q, qd, srbm["x"], dt, mass, Ib, Ibi, mu=consts.mu)
    s = R.kinodyn_solve_24(N, args)
    ok, cert = s["status"] == 0, s["status"] == 3
    assert (ok | cert).mean() >= 0.995, np.bincount(s["status"], minlength=4)
    assert ok.sum() >= 0.75 * B
    prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b], kin_box_y0=0.125) for b in range(B)]      # the 24-argument function = generate_landingCtrller_KNITRO.m's: kin_box_y = 0.125 + kin_box(2) (:154)
    lb, ub, cost = (np.array([p[i] for p in prob]) for i in range(3))
    # every converged member: the oracle's KKT residual, and it is what the kernel reported
    k = _certify(s["x"][ok], s["lam_g"][ok], lb[ok], ub[ok], cost[ok], dt, consts.mu)
    assert k.max() <= KKT_TOL * 1.0001, (k.max(axis=0), int(np.argmax(k.max(axis=1))))
    assert np.allclose(k, s["kkt"][ok], rtol=1e-3, atol=1e-9)
    assert s["f"][ok].max() <= 1e-7                       # the terminal reference is reachable: f* = 0 (cf. the stored solution, tests/test_n1_rows.py)
    oU = 12 * (N + 1) + 12 * N
    assert np.array_equal(s["x"][ok][:, :6], q[ok]) and np.array_equal(s["x"][ok][:, 6:12], qd[ok])
    assert np.array_equal(s["x"][ok][:, oU:oU + 12], args["c_init"].T[ok])
    # every certificate: the oracle's rows of the first interval over the fixed variables are violated by what the kernel reports
    g = ko.nlp_g_batch(s["x"][cert], N, dt, mass, Ib, Ibi, consts.mu)
    rows = 48 + 16 + 15 * np.repeat(np.arange(4), 5) + np.tile([0, 8, 9, 10, 11], 4)
    viol = np.maximum(np.maximum(lb[cert] - g, g - ub[cert]), 0.0)
    assert (viol[:, rows].max(axis=1) > KKT_TOL).all() and (s["iters"][cert] == 0).all()
    assert np.allclose(viol.max(axis=1), s["kkt"][cert][:, 0], rtol=1e-9, atol=1e-12)
    print("refinement of 1024: converged %d, certified infeasible %d, undecided %d; iterations mean %.1f max %d" % (ok.sum(), cert.sum(), B - ok.sum() - cert.sum(), s["iters"][ok].mean(), s["iters"][ok].max()))


def test_stored_drop_known_answer(ctx):
    """the drop of test_scripts/1.5msDrop30Pitch.mat (pure pitch 30 deg, v_z = -1.5, uniform dt = 0.03; the file holds a kinodynamic solution of
    an older variant of the script, FK band 1e-3): from ITS initial state and stance, with the current script's bounds (:208-258), the pipeline
    returns a KKT point with f* <= 2e-5 (the terminal reference is reachable, f* = 0) that is feasible under those bounds."""
    L, R = ctx
    P, kd = lc("problem"), lc("kinodyn")
    d = np.load(os.path.join(GOLDEN, "n1_kinodyn_solutions.npz"))
    X, U = d["X_a"], d["U_a"]
    q, qd = X[:6, 0].copy(), X[6:, 0].copy()
    dt = np.full(N, 0.03)
    consts = P.production_constants("main")
    p, x0s, _, _ = P.make_member(N, 0.6, q, qd, consts, dt)
    srbm = L.solve_host(p[None], x0s[None])
    assert srbm["status"][0] == 0
    mass, Ib, Ibi, _ = _consts()
    args = kd.make_args24(N, q[None], qd[None], srbm["x"], dt, mass, Ib, Ibi, mu=consts.mu)
    args["c_init"] = U[:12, 0].reshape(12, 1).copy()      # the stance the file starts from
    args["x0"][12 * (N + 1) + 12 * N:12 * (N + 1) + 12 * N + 12, 0] = U[:12, 0]
    s = R.kinodyn_solve_24(N, args)
    assert s["status"][0] == 0 and s["f"][0] <= 2e-5, (s["status"], s["f"], s["kkt"])
    lb, ub = kd.bounds(N, q, qd, U[:12, 0], kd.kin_box_of(q[3:6], qd[3:6]), kin_box_y0=0.125)      # (the 24-argument function's form: landing_kinodyn_form_knitro)
    cost = np.concatenate([kd.QN_DEFAULT, kd.Q_TERM_REF, np.zeros(6)])
    k = _certify(s["x"], s["lam_g"], lb[None], ub[None], cost[None], dt, consts.mu)
    assert k.max() <= KKT_TOL * 1.0001, k


def test_device_entry_point_is_deterministic_and_batch_independent(ctx):
    """landing_kinodyn_solve_batch on device pointers: the same batch twice and a member alone give identical bits (fixed summation orders,
    the only atomic is the integer count of members still iterating)"""
    import torch
    L, R = ctx
    P, kd = lc("problem"), lc("kinodyn")
    B = 8
    consts = P.production_constants("main")
    Pp, X0, q, qd = P.make_batch(B, N, 0.6, seed=8, consts=consts, dt_grid="reference", law="main")
    srbm = L.solve_host(Pp, X0)
    mass, Ib, Ibi, dt = _consts()
    prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b]) for b in range(B)]
    lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
    a = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, consts.mu)
    T = lambda v: torch.tensor(v, device="cuda")
    dl, du_, dc, dx0 = T(lb), T(ub), T(cost), T(x0)
    nx, ng = kd.dims(N)
    x = torch.empty(B, nx, device="cuda", dtype=torch.float64); st = torch.empty(B, device="cuda", dtype=torch.int32); it = torch.empty(B, device="cuda", dtype=torch.int32)
    R.kinodyn_solve_device(B, N, dl.data_ptr(), du_.data_ptr(), dc.data_ptr(), dx0.data_ptr(), dt, mass, Ib, Ibi, consts.mu, None, x.data_ptr(), d_status=st.data_ptr(),
                           d_iters=it.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy(), a["x"]) and np.array_equal(it.cpu().numpy(), a["iters"]) and np.array_equal(st.cpu().numpy(), a["status"])
    b = int(np.nonzero(a["status"] == 0)[0][0])
    c = R.kinodyn_solve_host(N, lb[b], ub[b], cost[b], x0[b], dt, mass, Ib, Ibi, consts.mu)
    assert np.array_equal(c["x"][0], a["x"][b]) and c["iters"][0] == a["iters"][b]


def test_retry_ladder_on_the_hard_sampling_law(ctx):
    """landing_kinodyn_solve_batch_host with the library defaults (opts = NULL) re-solves the members its first pass left undecided with another
    slack initialisation, then with a smaller first barrier parameter (csrc/kd_capi.inc).  256 drop states of the data-generation law (faster
    drops; measured on 1024: 19 undecided after one pass, 5 after the ladder): fewer undecided members than one pass with the same defaults passed
    explicitly, at most 1 %, the members decided in the first pass keep their bits, and every converged member is a KKT point under the oracle."""
    L, R = ctx
    P, kd = lc("problem"), lc("kinodyn")
    B = 256
    consts = P.production_constants("datagen")
    Pp, X0, q, qd = P.make_batch(B, N, 0.6, seed=7, consts=consts, dt_grid="reference", law="datagen")
    srbm = L.solve_host(Pp, X0)
    mass, Ib, Ibi, dt = _consts()
    lbs, ubs, costs, x0s = [], [], [], []
    for b in range(B):
        lb, ub, cost, x0 = kd.member_problem(N, q[b], qd[b], srbm["x"][b], None)
        lbs.append(lb); ubs.append(ub); costs.append(cost); x0s.append(x0)
    lbs, ubs, costs, x0s = np.array(lbs), np.array(ubs), np.array(costs), np.array(x0s)
    one = R.kinodyn_solve_host(N, lbs, ubs, costs, x0s, dt, mass, Ib, Ibi, consts.mu, R.kinodyn_default_opts())
    lad = R.kinodyn_solve_host(N, lbs, ubs, costs, x0s, dt, mass, Ib, Ibi, consts.mu, None)
    und1 = (one["status"] == 1) | (one["status"] == 2); und2 = (lad["status"] == 1) | (lad["status"] == 2)
    assert und2.sum() <= und1.sum() and und2.mean() <= 0.01, (und1.sum(), und2.sum())
    dec1 = ~und1
    assert np.array_equal(one["status"][dec1], lad["status"][dec1]) and np.array_equal(one["x"][dec1], lad["x"][dec1])
    ok = lad["status"] == 0
    kk = _certify(lad["x"][ok], lad["lam_g"][ok], lbs[ok], ubs[ok], costs[ok], dt, consts.mu)
    assert kk.max() <= KKT_TOL * 1.0001, kk.max()
    print("hard law, 256 members: undecided after one pass %d, after the ladder %d; converged %d, certified infeasible %d" % (und1.sum(), und2.sum(), ok.sum(), (lad["status"] == 3).sum()))


def test_feasibility_phase_decides_the_hard_law_in_one_pass(ctx):
    """Round 5: the kinodynamic solver has the SRBM solver's feasibility (restoration) phase (landing_kd_iter_kernel: elastic rows, el_step).  Law "datagen"
    (faster drops, generate_training_data_automated.m:47-50), 1024 drop states, ONE pass with explicit options (the host entry point's retry ladder only
    runs without options): at most 0.5 % undecided (round 4: 19 of 1024 after one pass, 5 after the ladder), every converged member re-certified under the
    oracle, and every certificate of local infeasibility checked: the violation the kernel reports is the oracle's, the dynamics and the fixed initial rows
    hold to 1e-3 where the certificate comes from the phase (a presolve certificate does not iterate at all)."""
    L, R = ctx
    P, kd = lc("problem"), lc("kinodyn")
    from oracle import kinodyn_oracle as ko
    B = 1024
    consts = P.production_constants("datagen")
    Pp, X0, q, qd = P.make_batch(B, N, 0.6, seed=7, consts=consts, dt_grid="reference", law="datagen")
    srbm = L.solve_host(Pp, X0)
    mass, Ib, Ibi, dt = _consts()
    prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b]) for b in range(B)]
    lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
    o = R.kinodyn_default_opts()
    assert (o.feas_phase, o.feas_jam, o.feas_stat) == (1, 0, 25)
    assert (o.clip_k, o.restart_period, o.kd_clone_after, o.kd_clone_max, o.kd_clone_iter) == (16, 30, 56, 96, 200)      # round 5: portfolio (landing_nlp.h)
    s = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, consts.mu, o)
    ok, cert = s["status"] == 0, s["status"] == 3
    print("law datagen, one pass: %d converged + %d certified + %d undecided; iterations max %d" % (ok.sum(), cert.sum(), (~ok & ~cert).sum(), s["iters"].max()))
    assert (~ok & ~cert).sum() <= 5 and ok.sum() >= 0.94 * B
    k = _certify(s["x"][ok], s["lam_g"][ok], lb[ok], ub[ok], cost[ok], dt, consts.mu)
    assert k.max() <= KKT_TOL * 1.0001, (k.max(axis=0), int(np.argmax(k.max(axis=1))))
    g = ko.nlp_g_batch(s["x"][cert], N, dt, mass, Ib, Ibi, consts.mu)
    lbc, ubc = lb[cert], ub[cert]
    viol = np.maximum(np.maximum(lbc - g, g - ubc), 0.0)
    assert np.allclose(viol.max(axis=1), s["kkt"][cert, 0], rtol=1e-6, atol=1e-9) and (viol.max(axis=1) > KKT_TOL).all()
    phase = cert & (s["iters"] > 0)      # certificates of the feasibility phase (the presolve ones have no iteration)
    assert phase.sum() >= 1
    eq = lbc == ubc
    ineq_v = np.where(~eq, viol, 0.0).max(axis=1); eq_v = np.where(eq, viol, 0.0).max(axis=1)
    pm = phase[cert]
    assert (eq_v[pm] <= 1e-3 * 1.0001).all() and (ineq_v[pm] > 1e-6).all()
    o.feas_phase = 0      # ... and without the phase those members end undecided
    s0 = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, consts.mu, o)
    assert (s0["status"][phase] != 0).all() and np.isin(s0["status"], (1, 2)).sum() > (~ok & ~cert).sum()


def test_portfolio_takes_the_outlier_out_of_the_batch(ctx):
    """landing_solver_opts::kd_clone_after (round 5): 1024 drop states of law "main", seed 10 -- without the portfolio (clip_k 16 as in the defaults) one member needs 670 iterations
    and the lock-step loop waits for it (1.54 s for the batch through the host entry point); with it (the defaults) that member's family converges after
    105 (0.52 s), nobody needs more than 250 iterations of its own, nobody is undecided, every member that converged without the portfolio still converges, and every converged member -- the
    clones' winners included -- is a KKT point <= 1e-6 under the oracle.  Members that never met the clone time keep every bit."""
    import time
    L, R = ctx
    P, kd = lc("problem"), lc("kinodyn")
    B = 1024
    consts = P.production_constants("main")
    Pp, X0, q, qd = P.make_batch(B, N, 0.6, seed=10, consts=consts, dt_grid="reference", law="main")
    srbm = L.solve_host(Pp, X0)
    mass, Ib, Ibi, dt = _consts()
    prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b]) for b in range(B)]
    lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
    on = R.kinodyn_default_opts()
    off = R.kinodyn_default_opts(); off.kd_clone_after = 0
    res, secs = {}, {}
    for name, o in (("off", off), ("on", on)):
        t = time.perf_counter(); res[name] = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, consts.mu, o); secs[name] = time.perf_counter() - t
    a, s = res["off"], res["on"]
    again = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, consts.mu, on)      # relatives that converge in one round: the lowest index wins, whatever the timing
    assert np.array_equal(again["x"], s["x"]) and np.array_equal(again["iters"], s["iters"]) and np.array_equal(again["status"], s["status"])
    und = lambda r: int(np.isin(r["status"], (1, 2)).sum())
    print("portfolio off: %d undecided, slowest member %d iterations, %.2f s; on: %d undecided, slowest %d, %.2f s (host entry point, copies included)" % (
        und(a), a["iters"].max(), secs["off"], und(s), s["iters"].max(), secs["on"]))
    assert a["iters"].max() >= 400                                   # the outlier is there ...
    assert und(s) == 0 and s["iters"].max() <= 250                    # ... and gone
    assert np.all(s["status"][a["status"] == 0] == 0) and (s["status"] == 0).sum() >= (a["status"] == 0).sum()
    early = a["iters"] < 50                                           # finished before the clone time (56 rounds): untouched
    assert early.sum() >= 0.7 * B and np.array_equal(a["x"][early], s["x"][early]) and np.array_equal(a["iters"][early], s["iters"][early])
    ok = s["status"] == 0
    kk = _certify(s["x"][ok], s["lam_g"][ok], lb[ok], ub[ok], cost[ok], dt, consts.mu)
    assert kk.max() <= KKT_TOL * 1.0001, kk.max()
    assert np.allclose(kk, s["kkt"][ok], rtol=1e-3, atol=1e-9)


def test_warm_start_preset_resolves_from_the_previous_solution_in_a_few_iterations(ctx):
    """landing_kinodyn_solver_opts_warm (round 6): every production caller solves the refinement NLP twice, the second time from the first solution
    (`prevSoln`, main_scripts/landing_optimization.m:395-435; generate_solver/generate_landingCtrller_KNITRO_warmstart.m builds the `_ws` function for it).
    256 drop states of law "main": cold solve, then the re-solve of every converged member from its x* under the preset -- all converge again, 99 % of them in
    at most 10 iterations (measured: mean 4.0, median 3, p99 5; one member in 216 needs ~110 under every setting tried, tools/dev/kd_warm_probe.py; cold: ~31 on
    average), to KKT points <= 1e-6 under the oracle; with the cold-start defaults the same re-solve needs several times as many."""
    L, R = ctx
    P, kd = lc("problem"), lc("kinodyn")
    B = 256
    consts = P.production_constants("main")
    Pp, X0, q, qd = P.make_batch(B, N, 0.6, seed=12, consts=consts, dt_grid="reference", law="main")
    srbm = L.solve_host(Pp, X0)
    mass, Ib, Ibi, dt = _consts()
    prob = [kd.member_problem(N, q[b], qd[b], srbm["x"][b]) for b in range(B)]
    lb, ub, cost, x0 = (np.array([p[i] for p in prob]) for i in range(4))
    cold = R.kinodyn_solve_host(N, lb, ub, cost, x0, dt, mass, Ib, Ibi, consts.mu, R.kinodyn_default_opts())
    ok = cold["status"] == 0
    assert ok.sum() >= 0.75 * B
    w = R.kinodyn_warm_opts()
    assert (w.bound_push, w.bound_frac, w.mu_init, w.kd_clone_after, w.max_iter) == (1e-6, 1e-6, 1e-6, 0, 100)
    warm = R.kinodyn_solve_host(N, lb[ok], ub[ok], cost[ok], cold["x"][ok], dt, mass, Ib, Ibi, consts.mu, w)
    again = R.kinodyn_solve_host(N, lb[ok], ub[ok], cost[ok], cold["x"][ok], dt, mass, Ib, Ibi, consts.mu, R.kinodyn_default_opts())
    print("re-solve from x*: warm preset %d / %d converged, iterations mean %.1f max %d; cold-start defaults mean %.1f max %d; the cold solve itself mean %.1f" % (
        (warm["status"] == 0).sum(), ok.sum(), warm["iters"].mean(), warm["iters"].max(), again["iters"].mean(), again["iters"].max(), cold["iters"][ok].mean()))
    assert (warm["status"] == 0).all() and np.percentile(warm["iters"], 99) <= 10 and np.median(warm["iters"]) <= 5
    assert again["iters"].mean() >= 2.0 * warm["iters"].mean()
    kk = _certify(warm["x"], warm["lam_g"], lb[ok], ub[ok], cost[ok], dt, consts.mu)
    assert kk.max() <= KKT_TOL * 1.0001
    rel = (np.abs(warm["x"] - cold["x"][ok]) / np.maximum(1.0, np.abs(cold["x"][ok]))).max(axis=1)
    assert np.median(rel) <= 5e-3 and np.percentile(rel, 90) <= 0.1      # the same solution, polished (the optimum f* = 0 is a continuum: a member may slide a few centimetres / newtons along it, and the
                                                                          # one member in ~200 that needs a hundred iterations ends at another KKT point altogether)
