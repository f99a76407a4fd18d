"""GPU tests of the batched interior-point solver, through the C ABI (run with -m gpu).

Solver parity is defined as SURVEY section 7 "hard parts" prescribes: the KKT residual of OUR solution
under the REFERENCE-EQUIVALENT functions (the oracle, pinned to the reference's generated C), to the
tolerance north_star states (fp64, <= 1e-6 on pr_inf / du_inf / compl, unscaled), plus -- where the
reference pins a number -- the optimal objective (golden known-answer: f* <= 2e-5 for 1.5msDrop30Pitch).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, lc

pytestmark = pytest.mark.gpu
KKT_TOL = 1e-6


@pytest.fixture(scope="module")
def libs():
    capi = lc("capi")
    return {N: capi.LandingLib(N, device=0) for N in (20, 40)}


@pytest.mark.parametrize("N,B", [(20, 24), (40, 32)])
def test_solver_reaches_kkt_under_oracle_functions(libs, oracle_mod, N, B):
    O = oracle_mod.Oracle(N)
    # N = 40: the uniform 15 ms grid of SURVEY 8(d).  N = 20: the grid the reference's callers pose (problem.REFERENCE_DT_GRID)
    Pm = lc("problem")
    P, X0, _, qd = Pm.make_batch(B, N, 0.6, seed=1, consts=Pm.production_constants("main") if N == 20 else None, dt_grid="reference" if N == 20 else "uniform")
    r = libs[N].solve_host(P, X0)
    conv = r["status"] == 0
    assert conv.all(), f"{(~conv).sum()} members failed: {np.nonzero(~conv)[0]}"
    for b in np.nonzero(conv)[0]:
        k = O.kkt(r["x"][b], P[b], r["lam_g"][b])
        assert k.max() <= KKT_TOL * 1.0001, (b, k)
        assert np.allclose(k, r["kkt"][b], rtol=1e-6, atol=1e-12)      # the kernel reports the same residual
        assert abs(O.f(r["x"][b], P[b]) - r["f"][b]) < 1e-12
        assert np.array_equal(r["x"][b][:12], np.concatenate([P[b][O.param_offsets()["q_init"]:][:6], P[b][O.param_offsets()["qd_init"]:][:6]]))


def test_solver_golden_known_answer(libs, oracle_mod):
    """reference golden (test_scripts/1.5msDrop30Pitch.mat): for that p the optimum is f* in [0, 1.64e-5]"""
    O = oracle_mod.Oracle(20)
    d = np.load(os.path.join(GOLDEN, "n20_golden_1p5ms30pitch.npz"))
    p = d["p"]
    # the callers' initial guess: linear references (generate_training_data_automated.m:105-119)
    o = O.param_offsets()
    q0, qd0 = p[o["q_init"]:o["q_init"] + 6], p[o["qd_init"]:o["qd_init"] + 6]
    _, x0, _, _ = lc("problem").make_member(20, 0.6, q0, qd0)
    r = libs[20].solve_host(p[None], x0[None])
    assert r["status"][0] == 0
    assert r["f"][0] <= 2e-5
    assert O.kkt(r["x"][0], p, r["lam_g"][0]).max() <= KKT_TOL * 1.0001


def test_solver_reference_default_problem(libs, oracle_mod):
    """BASELINE configs[0]: the generator script's own verification drop (generate_landingCtrller_IPOPT.m:173-224), f* = 0"""
    O = oracle_mod.Oracle(20)
    p, x0 = lc("problem").reference_default_problem()
    r = libs[20].solve_host(p[None], x0[None])
    assert r["status"][0] == 0 and r["f"][0] <= 1e-8
    assert O.kkt(r["x"][0], p, r["lam_g"][0]).max() <= KKT_TOL * 1.0001


def test_failed_member_does_not_poison_batch(libs, oracle_mod):
    """a member with NaN parameters is flagged and the others still converge (failure isolation)"""
    N = 20
    P, X0, _, _ = lc("problem").make_batch(4, N, 0.6, seed=3)
    P[2, :] = np.nan
    r = libs[N].solve_host(P, X0)
    assert r["status"][2] == 2
    assert (r["status"][[0, 1, 3]] == 0).all()


def test_warm_start_converges_faster(libs):
    """re-solve from the previous solution (the reference's *_ws variant, test_loadCasadi_ws.m:73-88)"""
    N = 20
    P, X0, _, _ = lc("problem").make_batch(4, N, 0.6, seed=5)
    L = libs[N]
    cold = L.solve_host(P, X0)
    o = L.default_opts(); o.bound_push = 5e-3; o.bound_frac = 5e-3    # generate_landingCtrller_IPOPT_warmstart.m:246-247
    warm = L.solve_host(P, cold["x"], o)
    ok = (cold["status"] == 0) & (warm["status"] == 0)
    assert ok.sum() >= 3 and warm["iters"][ok].sum() < cold["iters"][ok].sum()
    w2 = L.solve_host(P, cold["x"], L.warm_opts())                    # the receding-horizon options: a handful of iterations from a solution
    ok2 = (cold["status"] == 0)
    assert (w2["iters"][ok2] <= 14).all() and (w2["status"][ok2] == 0).sum() >= ok2.sum() - 1


def test_solver_other_horizons_and_limits(oracle_mod):
    """N is a runtime parameter of the solver too (N <= 96: the lane = stage phases loop over 64-stage chunks, the forward sweep keeps sigma_0..N in LDS)."""
    capi = lc("capi")
    for N in (16, 30, 80):      # (80 > 64: two chunks of the lane = stage phases, round 4; CPU port: 6 of 6 in 45..51 iterations)
        O = oracle_mod.Oracle(N)
        L = capi.LandingLib(N, device=0)
        P, X0, _, _ = lc("problem").make_batch(6, N, 0.6, seed=4)
        r = L.solve_host(P, X0)
        ok = r["status"] == 0; cert = r["status"] == 3
        # every member is DECIDED: a KKT point, or a certificate of local infeasibility from the feasibility phase (round 3).  Measured: N = 16
        # (uniform dt = 37.5 ms, a grid no caller of the reference uses) 5 converge + 1 certified (round 2: "about three quarters solve"); N = 30: all solve
        assert (ok | cert).all() and ok.sum() >= (5 if N == 16 else 6), r["status"]
        for b in np.nonzero(ok)[0]:
            assert O.kkt(r["x"][b], P[b], r["lam_g"][b]).max() <= KKT_TOL * 1.0001
        for b in np.nonzero(cert)[0]:      # the certificate: equality rows met, a positive violation of the inequality rows (1-norm above feas_cert)
            g = O.g(r["x"][b], P[b]); lb, ub = O.bounds(P[b]); eq = lb == ub
            assert np.abs(g[eq] - lb[eq]).max() <= 1e-6 and np.maximum(np.maximum(lb - g, g - ub), 0.0)[~eq].sum() > 1e-4
        L.close()
    L = capi.LandingLib(100, device=0)
    P, X0, _, _ = lc("problem").make_batch(1, 100, 0.6, seed=4)
    with pytest.raises(RuntimeError, match="N <= 96"):
        L.solve_host(P, X0)
    L.close()


def test_solver_is_deterministic_and_batch_independent(libs):
    """same member alone or inside a batch, run twice: identical bits (no atomics, fixed summation order)"""
    N = 20
    P, X0, _, _ = lc("problem").make_batch(5, N, 0.6, seed=8)
    a = libs[N].solve_host(P, X0)
    b = libs[N].solve_host(P, X0)
    c = libs[N].solve_host(P[2:3], X0[2:3])
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["iters"], b["iters"])
    assert np.array_equal(a["x"][2], c["x"][0]) and a["iters"][2] == c["iters"][0]


def test_solver_full_size_batch_properties(libs, oracle_mod):
    """BASELINE configs[1] size (N=40, B=1024), device-pointer entry point: >= 93 % converge within 300 iterations,
    every converged member satisfies the reported KKT bound, initial state rows are met exactly, every
    converged member is re-certified with the oracle's functions."""
    import torch
    N, B = 40, 1024
    L = libs[N]
    O = oracle_mod.Oracle(N)
    P, X0, _, _ = lc("problem").make_batch(B, N, 0.6, seed=20211)
    dev = "cuda"
    dP, dX0 = torch.tensor(P, device=dev), torch.tensor(X0, device=dev)
    mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
    x, f, lam, kkt = mk(B, L.nx), mk(B), mk(B, L.ng), mk(B, 3)
    st, it = mk(B, dt=torch.int32), mk(B, dt=torch.int32)
    o = L.default_opts(); o.max_iter = 300
    L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), f.data_ptr(), lam.data_ptr(), st.data_ptr(), it.data_ptr(), kkt.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sth, kh, xh, lh = st.cpu().numpy(), kkt.cpu().numpy(), x.cpu().numpy(), lam.cpu().numpy()
    ok = sth == 0
    assert ok.sum() >= B - 1, f"{ok.sum()}/{B}"        # measured: 1024/1024 (rounds 1 and 2); one straggler is tolerated
    assert kh[ok].max() <= KKT_TOL * 1.0001
    ith = it.cpu().numpy()
    # measured: round 2 mean 50.3, p99 79, max 95; round 3 (delta_floor, kappa_eps 80): mean 40.1, p99 52, max 75
    assert ith.mean() <= 44 and np.percentile(ith, 99) <= 65 and ith.max() <= 120, (ith.mean(), np.percentile(ith, 99), ith.max())
    po = O.param_offsets()
    assert np.array_equal(xh[:, :6], P[:, po["q_init"]:po["q_init"] + 6]) and np.array_equal(xh[:, 6:12], P[:, po["qd_init"]:po["qd_init"] + 6])
    for b in np.nonzero(ok)[0]:      # EVERY converged member re-certified under the oracle's (reference-pinned) functions (was every 97th)
        assert O.kkt(xh[b], P[b], lh[b]).max() <= KKT_TOL * 1.0001, b


def test_solver_config3_shard_size_on_one_gpu(libs, oracle_mod):
    """BASELINE configs[2]'s total (B = 8192, N = 40) on ONE GPU: the workspace (9 GB of the 288 GB), the dispatch of 8192
    workgroups over 512 resident slots and failure isolation at that size.  >= 99 % converge within 300 iterations; every one of
    the converged members is re-certified under the oracle's functions; member i equals member i of the 1024-batch with the
    same seed (batch independence at full size)."""
    import torch
    N, B = 40, 8192
    L = libs[N]
    O = oracle_mod.Oracle(N)
    Ps, X0s = [], []
    for r in range(8):                                     # the eight rank shards of bench.py --gpus 8 (seed 20211 + rank)
        Pr, Xr, _, _ = lc("problem").make_batch(1024, N, 0.6, seed=20211 + r)
        Ps.append(Pr); X0s.append(Xr)
    P, X0 = np.concatenate(Ps), np.concatenate(X0s)
    dev = "cuda"
    dP, dX0 = torch.tensor(P, device=dev), torch.tensor(X0, device=dev)
    mk = lambda *s, dt=torch.float64: torch.empty(*s, device=dev, dtype=dt)
    x, lam, kkt, st, it = mk(B, L.nx), mk(B, L.ng), mk(B, 3), mk(B, dt=torch.int32), mk(B, dt=torch.int32)
    o = L.default_opts(); o.max_iter = 300
    L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, lam.data_ptr(), st.data_ptr(), it.data_ptr(), kkt.data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sth, kh, xh, lh = st.cpu().numpy(), kkt.cpu().numpy(), x.cpu().numpy(), lam.cpu().numpy()
    ok = sth == 0
    assert ok.mean() >= 0.999, f"{ok.sum()}/{B}"      # measured: 8192 / 8192
    assert kh[ok].max() <= KKT_TOL * 1.0001
    for b in np.nonzero(ok)[0]:      # every converged member of the 8192 (a C call each: a few seconds in total)
        assert O.kkt(xh[b], P[b], lh[b]).max() <= KKT_TOL * 1.0001, b
    x1 = mk(1024, L.nx)
    L.solve_device(1024, dP.data_ptr(), dX0.data_ptr(), o, x1.data_ptr(), 0, 0, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(x1, x[:1024])


RUN_COST = dict(QX=[0, 0, 10, 1, 1, 0, .1, .1, .1, .1, .1, .1], Qc=[1.0, 1.0, 0.5], Qf=[1e-4, 1e-4, 1e-3], f_ref=[0, 0, 20.0])


@pytest.mark.parametrize("N,B", [(20, 16), (40, 32)])
def test_solver_with_running_cost(oracle_mod, N, B):
    """objective of the reference's N=41 script (generate_quadruped_SRBM_CCC.m:81-89): KKT point certified by the oracle"""
    O = oracle_mod.Oracle(N, run_cost=RUN_COST)
    L = lc("capi").LandingLib(N, device=0, run_cost=RUN_COST)
    P, X0, _, _ = lc("problem").make_batch(B, N, 0.6, seed=2)
    r = L.solve_host(P, X0)
    conv = r["status"] == 0
    # round 3: every member is decided (converged, or certified locally infeasible by the feasibility phase); measured N = 20: 15 + 1, N = 40: 32 + 0
    assert (conv | (r["status"] == 3)).all() and conv.mean() >= (0.9 if N == 20 else 0.96), f"only {conv.sum()}/{B} members converged: {r['status']}"
    for b in np.nonzero(conv)[0][:12]:
        assert O.kkt(r["x"][b], P[b], r["lam_g"][b]).max() <= KKT_TOL * 1.0001
        assert abs(O.f(r["x"][b], P[b]) - r["f"][b]) < 1e-9 * max(1.0, abs(r["f"][b]))
    # function layer: f and grad_f carry the running cost, the Hessian of that variant is refused
    x = X0 + 0.01 * np.random.default_rng(0).normal(size=X0.shape)
    e = L.eval_host(x[:4], P[:4], want=("f", "grad_f"))
    for b in range(4):
        assert abs(e["f"][b] - O.f(x[b], P[b])) < 1e-10 * max(1.0, abs(e["f"][b]))
        assert np.allclose(e["grad_f"][b], np.asarray(O.grad_f(x[b], P[b])[-1] if isinstance(O.grad_f(x[b], P[b]), tuple) else O.grad_f(x[b], P[b])).ravel(), rtol=1e-11, atol=1e-12)
    with pytest.raises(RuntimeError):
        L.eval_host(x[:1], P[:1], lam_g=np.zeros((1, L.ng)), want=("hess",))


def test_n40_reference_solutions_known_answer(oracle_mod):
    """data/*.mat of the reference (N=40, produced by its N=41 script: kin-box .05/.05/.27, running GRF cost Qf, terminal QN;
    parameters of analysis/eval_SRBM_CCC.m:29-56): starting from the callers' linear references, the solver must reach a KKT
    point whose objective is not worse than the stored (IPOPT tol 1e-4) trajectory's objective."""
    P, Cn = lc("problem"), lc("constants")
    d = np.load(os.path.join(GOLDEN, "n40_golden.npz"))
    rc = dict(QX=[0] * 12, Qc=[0, 0, 0], Qf=[1e-4, 1e-4, 1e-3], f_ref=[0, 0, 0])
    kb = (0.05, 0.05, 0.27)
    O = oracle_mod.Oracle(40, kin_box=kb, run_cost=rc)
    L = lc("capi").LandingLib(40, device=0, kin_box=kb, run_cost=rc)
    mass, Ib, Ibi = Cn.robot_constants()
    N = 40
    Ps, X0s = [], []
    for x in d["x"]:
        X = x[:12 * 41].reshape(12, 41, order="F")
        q0, qd0 = X[:6, 0], X[6:, 0]
        Xref = np.zeros((12, N + 1))
        for i in range(6):
            Xref[i] = np.linspace(q0[i], [0, 0, 0.2, 0, 0, 0][i], N + 1); Xref[6 + i] = np.linspace(qd0[i], 0.0, N + 1)
        c_ref = P.SIDE_SIGN * np.tile([0.2, 0.1, -0.35], 4)
        Uref = np.zeros((24, N))
        for j in range(12):
            Uref[j] = Xref[j % 3, :-1] + c_ref[j]
        Ps.append(P.pack_params(N, Xref, np.full(N, 0.015), [-10, -10, .15, -10, -10, -10], [10, 10, 1, 10, 10, 10],
                                [-10, -10, -10, -40, -40, -40], [10, 10, 10, 40, 40, 40], q0, qd0,
                                [-10, -10, .15, -.1, -.1, -10], [10, 10, 5, .1, .1, 10], [-10, -10, -10, -40, -40, -40],
                                [10, 10, 10, 40, 40, 40], [0, 0, 100, 100, 100, 0, 10, 10, 10, 10, 10, 10], 1.0, .35, 250., mass, Ib, Ibi))
        X0s.append(np.concatenate([Xref.flatten(order="F"), Uref.flatten(order="F")]))
    Ps, X0s = np.array(Ps), np.array(X0s)
    r = L.solve_host(Ps, X0s)
    ok = r["status"] == 0
    assert ok.sum() >= len(Ps) - 1, r["status"]        # measured: 17/17 (round 1), 16/17 (first half of round 2), 17/17 (final)
    same = better = 0
    for b in np.nonzero(ok)[0]:
        assert O.kkt(r["x"][b], Ps[b], r["lam_g"][b]).max() <= KKT_TOL * 1.0001
        f_ref = O.f(d["x"][b], Ps[b])
        same += abs(r["f"][b] - f_ref) <= 1e-3 * f_ref          # the same local minimum (the stored one is feasible to 1e-3 only)
        better += r["f"][b] <= f_ref * 1.001
    # measured in round 1: 6 of 17 coincide to <= 1e-4 relative, 8 are better, 3 end in another (worse) local minimum;
    # round 2 (restart_period 60, delta_dec 0.5): 5 coincide, 6 are better, 5 end in a worse local minimum, 1 does not converge
    # round 2 with clip_k / theta_floor: 17 of 17 converge, 7 coincide, 12 same or better; final (+ dual_step_cap): 17 of 17, 5 coincide,
    # 13 have the same or a better objective, 4 a worse local minimum
    # round 3 (kappa_eps 80): 17 of 17, 5 coincide, 11 same or better, 6 larger (two of them by 1-2 %); kappa_eps 10..30 give 13, bound_push 1.0
    # or clip_k 8 give 9 / 8 -- the reason those faster settings are not the defaults (profiles/r03_delta_floor.txt)
    # (non-convex NLP: which KKT point a run reaches depends on the regularisation path; every returned point is certified above)
    print("N=40 stored reference solutions: converged %d of %d, same local minimum %d, same or better objective %d" % (ok.sum(), len(Ps), same, better))
    # round 4: kappa_eps chosen by formulation (0 = automatic: IPOPT's 10 for the forms with a running cost, which these stored solutions are;
    # 80 for the terminal-cost form of the bench): 17 of 17, 5 coincide, 13 same or better -- the round-2 level, assertion restored
    assert same >= 5 and better >= 13, (same, better, ok.sum())


def test_reference_bound_frac_is_a_supported_configuration(libs, oracle_mod):
    """the reference's own slack initialisation, bound_push = bound_frac = 0.5 (generate_landingCtrller_IPOPT.m:241-242): slower
    (every two-sided slack starts at its interval mid-point) but it must work: >= 95 % of a 256-member N=40 batch reach
    KKT <= 1e-6 within 600 iterations (measured round 1: 97.6 % within 300)"""
    N, B = 40, 256
    L = libs[N]
    P, X0, _, _ = lc("problem").make_batch(B, N, 0.6, seed=20211)
    o = L.default_opts(); o.bound_push = 0.5; o.bound_frac = 0.5; o.mu_init = 0.1; o.max_iter = 600
    r = L.solve_host(P, X0, o)
    ok = r["status"] == 0
    assert ok.mean() >= 0.95, ok.sum()
    assert r["kkt"][ok].max() <= KKT_TOL * 1.0001


@pytest.mark.parametrize("law,min_conv", [("main", 0.995), ("datagen", 0.965)])
def test_n20_production_problem_full_batch(libs, oracle_mod, law, min_conv):
    """BASELINE configs[0] as the reference's production callers pose it: N = 20 on the non-uniform grid
    dt = [0.05, 0.02 x 15, 0.05, 0.05, 0.1, 0.2] (landing_optimization.m:28, generate_training_data_automated.m:28, nn_warmstart.m:49), both
    sampling laws (problem.DROP_LAWS), each caller's own f_max, 1024 drop states per law, from the callers' linear references.
    Converged members are KKT points <= 1e-6 by the kernel's report, EVERY one re-certified under the oracle's (reference-pinned) functions.
    Round 6: status 3 means ONE thing again -- a KKT point of the elastic problem with positive violation, equality rows
    <= 1e-6 under the oracle for EVERY certificate; a stationary violation that is not such a point is LANDING_STALLED (4), counted as undecided
    here; no certified member may be one that the plain iteration (no phase) solves within 150 iterations (`lost == 0`); at most 0.5 % undecided.
    The phase now works the way IPOPT's restoration phase does (landing_nlp.h: feas_back, feas_max, feas_ret_push, feas_delta_dec, feas_polish, feas_resume).
    CPU port on these very batches: main 1020 converged + 4 certified, datagen 994 + 28 + 2 stalled; 16 x 1024 per law on the GPU: profiles/r06_soak_n20_*.json."""
    N, B = 20, 1024
    Pm = lc("problem")
    O = oracle_mod.Oracle(N)
    P, X0, _, qd = Pm.make_batch(B, N, 0.6, seed=7, consts=Pm.production_constants(law), dt_grid="reference", law=law)
    o = libs[N].default_opts(); o.max_iter = 300
    r = libs[N].solve_host(P, X0, o)
    ok = r["status"] == 0
    print("N=20 production grid, law %s: %d / %d converged, iterations mean %.1f p99 %.0f max %d; v_z of the others: %s" %
          (law, ok.sum(), B, r["iters"][ok].mean(), np.percentile(r["iters"][ok], 99), r["iters"][ok].max(), np.round(np.sort(qd[~ok, 5]), 2)[:12]))
    cert = r["status"] == 3
    print("   certified locally infeasible: %d (largest violation %.1e .. %.1e), stalled: %d, undecided otherwise: %d" % (
        cert.sum(), r["kkt"][cert, 0].min() if cert.any() else 0, r["kkt"][cert, 0].max() if cert.any() else 0, (r["status"] == 4).sum(), np.isin(r["status"], (1, 2)).sum()))
    assert ok.mean() >= min_conv, f"{ok.sum()}/{B}"
    assert (ok | cert).mean() >= 0.995, np.bincount(r["status"])
    assert r["kkt"][ok].max() <= KKT_TOL * 1.0001
    for b in np.nonzero(ok)[0]:      # every converged member is re-certified
        assert O.kkt(r["x"][b], P[b], r["lam_g"][b]).max() <= KKT_TOL * 1.0001
    for b in np.nonzero(cert)[0]:      # EVERY certificate: dynamics and initial state hold to 1e-6, the violation of the inequality rows is what the kernel reports
        g = O.g(r["x"][b], P[b]); lbb, ubb = O.bounds(P[b])
        eq = lbb == ubb
        assert np.abs(g[eq] - lbb[eq]).max() <= 1e-6 * 1.0001
        viol = np.maximum(np.maximum(lbb - g, g - ubb), 0.0)
        assert abs(viol.max() - r["kkt"][b, 0]) <= 1e-9 and viol.sum() > 1e-4
    o.feas_phase = 0      # ... and without the phase the certified members end undecided (NUMERICAL / MAX_ITER) or converge only after hundreds of
    r0 = libs[N].solve_host(P, X0, o)      # iterations; whoever the jam rule (feas_jam) never touched converges to the same bits
    ok0 = r0["status"] == 0
    lost = cert & ok0 & (r0["iters"] <= 150)      # sent into the phase by the jam rule although the plain iteration would have converged soon
    assert lost.sum() == 0 and ok0.sum() <= ok.sum() + 3, (lost.sum(), cert.sum(), ok0.sum(), ok.sum())
    same = ok0 & ok & (r0["iters"] == r["iters"])
    assert same.sum() >= 0.98 * ok0.sum() and np.array_equal(r0["x"][same], r["x"][same])


@pytest.mark.gpu
def test_soak_failures_are_rescued(oracle_mod):
    """members found by tools/soak.py (65 536 fresh drop states) that hit max_iter with IPOPT's independent dual step length and
    restarts in place only -- ordinary drop states, each solvable from the same initial guess with another step rule: they converge
    with the defaults (landing_solver_opts::dual_step_cap = 1 removes the jam itself; fresh_restart = 9 is the safety net), certified by
    the oracle; with every rule off they fail, and the proximal term (delta_floor) or the restart rules alone rescue them"""
    N = 40
    cases = [(100062, 614), (100062, 438), (100041, 890), (100039, 349), (100031, 450), (100027, 126), (100059, 370), (100044, 440)]
    O = oracle_mod.Oracle(N)
    Ps, Xs = [], []
    for seed, m in cases:
        P, X0, _, _ = lc("problem").make_batch(1024, N, 0.6, seed=seed)
        Ps.append(P[m]); Xs.append(X0[m])
    Ps, Xs = np.array(Ps), np.array(Xs)
    L = lc("capi").LandingLib(N, device=0)
    o = L.default_opts(); o.max_iter = 300
    assert (o.fresh_restart, o.dual_step_cap, o.slack_corr) == (9, 1.0, 0.9)
    r = L.solve_host(Ps, Xs, o)
    assert (r["status"] == 0).all() and r["iters"].max() <= 120, (r["status"], r["iters"])       # measured: 44..72 iterations
    for b in range(len(cases)):
        assert O.kkt(r["x"][b], Ps[b], r["lam_g"][b]).max() <= 1e-6 * 1.0001
    o.fresh_restart = 0; o.dual_step_cap = 0.0; o.slack_corr = 0.0; o.watchdog = 0; o.barrier_smax = 0.0; o.feas_phase = 0
    r2 = L.solve_host(Ps, Xs, o)
    assert (r2["status"] == 0).sum() >= 7, r2["status"]          # the proximal term alone (delta_floor, round 3) un-jams them: measured 8 of 8
    o.delta_floor = 0.0; o.kappa_eps = 10.0; o.mu_init = 0.1; o.bound_push = 0.5; o.theta_mu = 1.5     # ... the round-2 schedule they were found with (round 4: with the automatic mu_init / bound_push of this form, 0.5 / 1.0, 7 of the 8 converge even so)
    r0 = L.solve_host(Ps, Xs, o)
    assert (r0["status"] != 0).sum() >= 4, r0["status"]          # measured: 8 of 8 fail
    o.fresh_restart = 15
    r1 = L.solve_host(Ps, Xs, o)
    assert (r1["status"] == 0).sum() >= 7, r1["status"]          # the restart rules alone (measured: 8 of 8, 94..207 iterations)
    L.close()


@pytest.mark.gpu
def test_tail_rules_of_round4(libs, oracle_mod):
    """jam_clip / stag_relief (include/landing_nlp.h) through the C ABI on the bench batch (seed 20211): the slowest member (304: 88 iterations, 45 of them
    full Newton steps held back by the proximal term) ends within 60 iterations with the defaults, no member gets slower by more than 15 iterations, every
    member of both runs is a KKT point under the oracle."""
    N, B = 40, 1024
    L = libs[N]
    O = oracle_mod.Oracle(N)
    P, X0, _, _ = lc("problem").make_batch(B, N, 0.6, seed=20211)
    o = L.default_opts(); o.max_iter = 300
    assert (o.jam_clip, o.stag_relief) == (2, 3)
    o.kappa_eps = 80.0; o.theta_mu = 1.5      # the barrier schedule the slow member was found with (the automatic one moved on: 120 / 1.8)
    r1 = L.solve_host(P, X0, o)
    o.jam_clip = 0; o.stag_relief = 0
    r0 = L.solve_host(P, X0, o)
    assert (r0["status"] == 0).all() and (r1["status"] == 0).all()
    assert r0["iters"][304] >= 80 and r1["iters"][304] <= 60, (r0["iters"][304], r1["iters"][304])
    assert r1["iters"].max() < r0["iters"].max() and (r1["iters"] - r0["iters"]).max() <= 15
    for r in (r0, r1):
        for b in range(B):
            assert O.kkt(r["x"][b], P[b], r["lam_g"][b]).max() <= KKT_TOL * 1.0001, b
