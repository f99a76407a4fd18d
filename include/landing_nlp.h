/*
 * landing_nlp.h -- C ABI of the MI355X-native batched SRBM landing-NLP library
 * (liblanding_mi355x.so).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Two boundaries of the reference are covered:
 *
 *  (1) the batched forms of the CasADi external functions the reference's generated library
 *      exports (optimizations/landing/codegen_casadi/landingCtrller_IPOPT.c):
 *        nlp_f       :10995  -> f
 *        nlp_g       :11161  -> g
 *        nlp_grad_f  :52602  -> f, grad_f_x
 *        nlp_jac_g   :94014  -> g, jac_g_x (CCS nonzeros, pattern casadi_s5 :64)
 *        nlp_hess_l  :53527  -> hess_gamma_x_x (upper-triangular CCS nonzeros, casadi_s4 :63)
 *        nlp_grad    :22015  -> f, g, grad_gamma_x, grad_gamma_p
 *      landing_eval_batch() evaluates any subset of these for B independent (x,p[,lam]) in one
 *      launch; the single-problem CasADi ABI itself (same symbol names, `int F(const double**
 *      arg, double** res, long long* iw, double* w, int mem)` plus the 15 metadata functions per
 *      symbol, landingCtrller_IPOPT.c:10916-10993) is exported by the drop-in library
 *      landingCtrller_IPOPT_mi355x.so (include/landing_casadi_abi.h), which forwards to this one.
 *
 *  (2) the solver function the MATLAB callers invoke,
 *        [x*, f*] = landingCtrller_IPOPT(Xref, Uref, dt, q_min, q_max, qd_min, qd_max, q_init,
 *                     qd_init, q_term_min, q_term_max, qd_term_min, qd_term_max, QN, x0, mu,
 *                     l_leg_max, f_max, mass, Ib, Ib_inv)
 *      (generate_solver/generate_landingCtrller_IPOPT.m:323-327; call sites
 *      main_scripts/landing_optimization.m:305-311, generate_data/generate_training_data_automated.m:130-136),
 *      as landing_solve_batch(): B drop states at once, each argument with a leading batch
 *      dimension, the active parameters packed into p exactly as CasADi packs them.
 *
 * Layouts (all fp64, member-major, C order):
 *   x   [B][nx]   nx = 36N+12 : [X(:) ; U(:)], X 12x(N+1) column-major, U 24xN column-major
 *   p   [B][np]   np = 13N+94 : [Xref(:) ; dt ; q_min q_max qd_min qd_max q_init qd_init q_term_min
 *                               q_term_max qd_term_min qd_term_max ; QN ; mu l_leg_max f_max mass ; Ib ; Ib_inv]
 *   g, lam_g [B][ng]  ng = 104N+12 ;  jac [B][nnz_jac] ; hess [B][nnz_hess]
 * "d_" arguments are DEVICE pointers (HBM resident); `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  Every function returns 0 on success, a negative LANDING_E_* code on
 * error; no function falls back to a CPU path.
 */
#ifndef LANDING_NLP_H
#define LANDING_NLP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct landing_ctx landing_ctx;

#define LANDING_E_ARG (-1)     /* bad argument */
#define LANDING_E_HIP (-2)     /* HIP runtime error (see landing_last_error) */
#define LANDING_E_NODEV (-3)   /* no usable gfx950 device */

/* Formulation constants that are literals in the reference scripts (not parameters). */
typedef struct {
  double kin_box[3];  /* generate_landingCtrller_IPOPT.m:149-151 (.15,.15,.30); CCC variant .05,.05,.27 */
  double kin_z_off;   /* :155  0.05 */
  double comp_eps;    /* :140  1e-3 */
  double slip_eps;    /* :143-144  1e-2 */
  /* running cost of the reference's N=41 script (generate_quadruped_SRBM_CCC.m:81-89); 0 = terminal cost only
   * (generate_landingCtrller_IPOPT.m:83-87, the default):
   *   sum_k dt_k ( |X_k - Xref_k|^2_QX + sum_legs |pos_k + p_hip - c_k|^2_Qc + sum_legs |f_k - f_ref|^2_Qf )
   * QX, Qc, Qf are parameters of that script whose callers pass constants; here they are constants of the context.
   * Supported by landing_solve_batch and by every output of landing_eval_batch except d_hess: the running cost adds
   * diagonal entries that casadi_s4 does not hold, so its Hessian is returned by landing_eval_hess_rc_batch in the
   * extended pattern of landing_pattern_hess_rc. */
  int run_cost;       /* 0 off; 1 on, weights / force reference = the constants below, p as in the IPOPT variant (np = 13N+94);
                         2 on, with the N=41 script's OWN parameter vector (generate_quadruped_SRBM_CCC.m:49-71, Opti's order of the active
                         parameters): p = [Xref 12(N+1) | Uref 24N | dt N | q_min .. qd_term_max 60 | QX 12 | QN 12 | Qc 3 | Qf 3 | mu l_leg_max
                         f_max mass | Ib 3 | Ib_inv 3], np = 37N+112 (landing_np_ccc); QX, Qc, Qf and the force part of Uref are read from p,
                         grad_gamma_p has entries for them, the fields QX / Qc / Qf / f_ref below are ignored.  The solver function of that
                         script has 25 arguments: landing_pack_args25 / landing_solve_args25                                              */
  double QX[12], Qc[3], Qf[3];
  double f_ref[3];    /* Uref(13:24,k) = f_ref per leg in the callers (test_loadCasadi_ws.m:68-72) */
  double p_hip[12];   /* CCC :76-79 */
} landing_form;

/* Options of the interior-point solver; names follow the IPOPT options the reference sets
 * (generate_landingCtrller_IPOPT.m:231-264) where the meaning is the same. */
typedef struct {
  double tol;            /* unscaled KKT tolerance on pr/du/compl (default 1e-6)   */
  int max_iter;          /* default 3000 (:232)                                    */
  double mu_init;        /* first barrier parameter.  Default 0 = AUTOMATIC (round 4), like kappa_eps below: the reference's 0.1 (:247) for the
                            forms with a running cost -- on the 17 stored N = 40 solutions of the reference, which are of that form, 0.1 ends in
                            the stored local minimum or a better one 13 times, 0.2 / 0.3 / 0.5 / 1.0 only 10 / 12 / 11 / 11 times
                            (tools/dev/quality17.py) -- and 0.5 for the terminal-cost form: with this solver's monotone schedule a member
                            leaves the first barrier problem better centred.  MI355X, 32 fresh batches of 1024 (N = 40, tools/dev/musweep.py):
                            0.1 / 0.3 / 0.5 / 0.7 / 1.0 -> 40.1 / 36.8 / 36.5 / 35.8 / 35.1 iterations on average, slowest member 89 / 115 / 80 /
                            86 / 85, 79.8 / 75.3 / 72.2 / 71.5 / 71.8 ms per batch through the host path; 2.0 loses one member in 8192.
                            A positive value is taken as given (the warm-start options set 1e-4).
                            Hold-out of 128 fresh batches (tools/soak.py): 131 072 / 131 072 converged with both, iterations mean 40.1 -> 36.5,
                            p99.9 62 -> 57, 72.7 -> 66.8 ms per batch.  On the reference's N = 20 production grid (16 x 1024 drop states, law
                            main) the mean falls too (43.9 -> 38.4) but the slow tail is longer: within 300 iterations 16 346 + 33 certified +
                            5 undecided with 0.1, 16 337 + 34 + 13 with 0.5; within 1500 iterations 16 346 + 37 + 1 against 16 348 + 35 + 1
                            (the members 0.5 leaves for later converge, where 0.1 certifies some of them infeasible) -- a caller of that
                            problem who caps the iterations tightly sets 0.1.                                                                  */
  double bound_push;     /* slack initialisation: distance a slack is pushed off its bounds.  Default 0 = AUTOMATIC (round 4): the reference's 0.5
                            (:242) for the forms with a running cost, 1.0 for the terminal-cost form.  1.0 is faster on every family tried
                            (round 3, MI355X, 64 fresh batches of 1024, N = 40, kappa_eps 80: 40.1 -> 36.7 iterations, 72.0 -> 68.1 ms per batch;
                            CPU port: N = 20 production grid 42.6 -> 38.6, running cost 60.0 -> 56.1) but ends at a WORSE local minimum than
                            the reference's stored N = 40 solutions -- which are of the running-cost form -- more often (same-or-better
                            objective on 9 of 17 instead of 11..13, tests/test_gpu_solver.py): there the reference's value stays.  The
                            terminal-cost form has no such question: its minimum is f* = 0 (the terminal reference is reachable), every
                            KKT point found is the global minimum (tools/dev/fstar_cmp.py: f* ~ 1e-10 under any setting).  With the automatic
                            mu_init, MI355X, two hold-out sets of 128 fresh batches (tools/soak.py): 262 144 / 262 144 converged, iterations
                            mean 34.1 (reference values 40.1), p99.9 53 (62), slowest member 112, 64.0 ms per batch (72.7); N = 20 production
                            grid, law main, within 300 iterations: 16 350 converged + 32 certified + 2 undecided (reference values 16 346 + 33 + 5),
                            law datagen 15 920 + 412 + 52 (15 950 + 410 + 24).  A positive value is taken as given.                          */
  double bound_frac;     /* default 0.1.  The reference sets 0.5 for IPOPT (:241); with 0.5 the slack of every two-sided row starts at
                            the mid-point of its interval whatever the initial guess says.  Measured on three seeded batches of
                            1024..2048 drop states (N=40, tools/dev/fracsweep.py): 0.5 -> 97.6 % solved, mean 80 iterations;
                            any value in 0.02..0.2 -> 100 % solved, mean 64 iterations.                                          */
  double kappa_eps;      /* barrier-subproblem tolerance factor (IPOPT's barrier_tol_factor, default there 10): a barrier problem counts as
                            solved when its scaled optimality error is <= kappa_eps mu.  Default 0 = AUTOMATIC (round 4): 120 for the
                            terminal-cost form (80 until the end of round 4, below; 120 together with theta_mu 1.8, see there), IPOPT's 10 for the forms with a running cost (landing_form.run_cost 1 / 2) -- on the
                            17 stored N = 40 solutions of the reference, which are solutions of the running-cost form, 10 ends in the stored
                            local minimum or a better one 13 times, 80 only 11 times (tests/test_gpu_solver.py), and the bench workload is
                            the terminal-cost form.  A positive value is taken as given.  The 80 of round 3:  IPOPT's 10 belongs to its
                            monotone mode, which the reference does not use (mu_strategy adaptive, :244): this solver's mu schedule IS the
                            monotone one, and with 10 a member spends 30-40 of its ~48 iterations polishing the first barrier problem
                            (mu = 0.1) before mu may fall.  MI355X, 64 fresh batches of 1024 (N = 40, delta_floor on): 5 / 10 / 20 / 40 /
                            80 / 160 / 400 -> 49.1 / 47.8 / 46.0 / 42.5 / 40.1 / 38.6 / 36.1 iterations on average, 86.7 / 84.9 / 80.7 /
                            76.4 / 72.0 / 74.0 / 79.4 ms per batch (beyond 80 the tail grows: p99.9 61 -> 70 -> 106).  With 80 and
                            bound_push 1: two hold-out sets of 128 batches each, 262 144 / 262 144 converged, worst member 122 iterations,
                            67.7 ms per batch; with the default bound_push see profiles/r03_delta_floor.txt.  One member in ~50 000 then leans on watchdog / theta_floor / fresh_restart (a 350 .. 690
                            iteration crawl without them), which at kappa_eps 10 had become no-ops (profiles/r03_delta_floor.txt)          */
  double kappa_mu;       /* 0.2                                                    */
  double theta_mu;       /* superlinear decrease of the barrier parameter, mu <- min(kappa_mu mu, mu^theta_mu).  Default 0 = AUTOMATIC (round 4): IPOPT's
                            1.5 for the forms with a running cost (on the 17 stored reference solutions 1.8 keeps 13 same-or-better but needs 135
                            instead of 124 iterations), 1.8 for the terminal-cost form together with kappa_eps 120.  MI355X, hold-out of 128 fresh
                            batches of 1024 (N = 40, device time per batch, tools/soak.py --seed0 500000; second set 700000): 1.5 / 80 -> 64.1 ms (63.6),
                            iterations mean 34.1, p99 43; 1.8 / 80 -> 63.2; 1.5 / 120 -> 63.4; 1.8 / 120 -> 60.7 (60.4), mean 32.3, p99 40, slowest
                            batch 74 instead of 89 ms; 1.8 / 160 -> 60.6; 2.0 / 120 -> 64.6 and 1.9 / 140 -> 63.3 (p99.9 59..61: the tail grows);
                            kappa_mu 0.1 -> slower.  All 131 072 members converge in every variant.                                            */
  int max_soc;           /* reserved, ignored: second-order corrections proper were measured in the CPU port in round 6 (2-3 % of the iterations meet the case, iteration counts
                            unchanged: DESIGN.md 4.3b, profiles/r06_ab_experiments.txt A14) and not built into the kernel; slack_corr below repairs the same rejections without a solve */
  int max_resets;        /* multiplier resets allowed per NLP (default 8), see reset_du.  (2 was tried in round 2: it stops the rare
                            locally infeasible member ~130 iterations earlier, but the N=41-script formulation -- kin-box
                            .05/.05/.27 with the running cost -- then loses 4 of its 17 stored reference cases.)            */
  double reset_du;       /* dual infeasibility above which slacks/multipliers/mu are re-initialised
                            at the current x (jammed iterate; IPOPT would enter restoration), 1e9  */
  int stage_local_reg;   /* ignored (a per-stage delta_w was tried in round 1 and removed); kept for ABI stability            */
  int sticky_delta;      /* experimental: restart from delta_last when the previous first trial failed; default 0          */
  int restart_period;    /* re-initialise slacks/multipliers/filter at the current x when the first barrier problem (mu = mu_init)
                            is still not solved this many iterations after the last (re)start (crawling iterate; counts
                            against max_resets); 0 = never; default 75.  (First half of round 2: 60 instead of 80 -- slowest member of
                            four seeded batches 149..160 instead of 177..236 iterations.  With the watchdog catching the crawling
                            iterates early the detector fires needlessly at 60: 75 over 64 fresh batches p99 of the iteration count
                            90 -> 79, mean batch time 101.9 -> 99.5 ms; 90: the same.)                                              */
  int dispatch_order;    /* 1 (default): members with a large initial body height are dispatched first (they tend to need the most
                            iterations and would otherwise set the batch time from the second wave); 0: batch order.  Results do
                            not depend on it (every member is solved independently).  Applied only to batches that do not fit the GPU at
                            once (more than 512 members) and up to 16 384 members (the ranking kernel compares all pairs)               */
  double delta_init;     /* first trial regularisation when none was needed before (IPOPT first_hessian_perturbation, 1e-4) */
  double delta_inc_first;/* growth factor while no regularised iteration happened yet (IPOPT 100; default 10)            */
  double delta_inc;      /* growth factor afterwards (IPOPT 8; default 4: finer steps over-regularise less, tools/strag.py) */
  double delta_dec;      /* first trial = delta_last * delta_dec (IPOPT 1/3; default 1/2: 5 % fewer stage eliminations at unchanged
                            iteration counts on three seeded batches, tests/dev/ipm_lab.py round 2)                        */
  double tau_min;        /* fraction-to-the-boundary floor (IPOPT 0.99; default 0.9)                                     */
  double alpha_fallback; /* step taken (and filter restarted) when the line search finds no acceptable point (1e-2)      */
  double reset_delta;    /* regularisation above which the iterate counts as jammed too (steps degenerate to damped
                            gradient steps); <= 0 disables; default 1e5                                                   */
  int clip_k;            /* fraction-to-the-boundary rule of the primal step: the step length is set by the clip_k-th most blocking
                            slack (1..4; default 4); the clip_k - 1 slacks that would limit it further stop at (1 - tau) of their
                            current distance to the bound instead -- exactly where the rule would have left them had each been the
                            only one -- and their row shows up in theta of the trial point, so the filter still decides.  Why: from
                            the callers' straight-line guess the Newton step is long against the slack distances and ONE slack at a
                            time cuts every step to 1..10 % for 3-4 iterations until its multiplier has grown (round-2 traces,
                            DESIGN.md 4.2); letting the few worst jam together instead of one after the other: mean 63 -> 53
                            iterations, slowest member of eight seeded batches 160..246 -> 100..133, inertia-failure retries
                            1.28 -> 1.15 sweeps per iteration (tests/dev/ipm_lab.py).  0 / 1 = the classic rule (IPOPT).
                            landing_kinodyn_solve_batch also takes values above 4 (default there 16, round 5): the step length that
                            leaves at most clip_k - 1 slacks blocked, from a histogram of the ratios over half-octaves (never below
                            the 4-slack rule's); landing_solve_batch treats values above 4 as 4                                    */
  double clip_until;     /* ... applied only while the primal infeasibility (max norm, slack rows included) is above this value
                            (default 0.03): close to feasibility the classic rule is kept -- without the switch 1 member in 1000
                            parks at pr ~ 2e-2 with diverging multipliers                                                     */
  double theta_floor;    /* filter line search: constraint violations (theta, 1-norm over the ~4000 rows) below theta_floor * tol count as
                            equal -- a trial point that stays below is never rejected for its theta.  Default 30.  At the last barrier
                            problems theta sits at ~1e-7, far below the tolerance, while the dual infeasibility still needs full Newton
                            steps whose second-order terms raise theta a little; the relative-decrease test alone then cuts every step to
                            1/64 .. 1e-7 (IPOPT gets past this with second-order corrections / its acceptable-point stop).  Convergence is
                            still decided by the max-norm residuals <= tol.  Measured (tools/soak.py, 64 fresh batches): floor 0 -> one
                            212-iteration member in the bench batches; 1 -> 99, but three members of the 65 536 still need 203..251
                            iterations with theta pinned AT the floor; 30 -> they need 51..69, mean batch time 110.2 -> 106.2 ms, slowest
                            batch 223 -> 142 ms; 10: 107.6 ms, 100: 106.1 ms.  0 = off                                              */
  int fresh_restart;     /* restart rules beyond "re-initialise slacks, multipliers and mu at the current x", a bit mask (default 9 = 1 | 8):
                            1: a restart that follows a JAM (dual infeasibility above reset_du, regularisation above reset_delta) goes back
                               to the caller's initial guess with clip_k = 2 -- a restart in place repeats the failure from a bad x;
                            2: so does every member's second restart;   4: the crawl detector may fire twice;
                            8: a LATER barrier problem (mu < mu_init) still unsolved 2 restart_period iterations after it began, with a
                               primal infeasibility above 1e-3 -- or one in which the watchdog has fired three times without effect --
                               has wandered off and is restarted in place (nothing else catches it).
                            History (tools/soak.py, 64 fresh batches = 65 536 drop states, profiles/r02_soak*.json): with restarts in
                            place only 12 members hit max_iter -- none infeasible or unusual, each solves in 50..100 iterations from the
                            same guess with another step rule -- and such a member sets the time of its batch (250 instead of 105 ms).
                            Rules 1|2|4|8 rescued all twelve; once dual_step_cap (below) removed the cause of the jams, rules 2 and 4 only
                            cost time (a 239-iteration member in the bench batches) and rule 8 alone keeps 65 536 of 65 536.  0 = none    */
  double dual_step_cap;  /* the step length of the bound multipliers is at most dual_step_cap times the accepted primal step length
                            (default 1: a_du <= alpha; 0 = IPOPT's independent dual step length).  Found through the members that
                            jammed (fresh_restart above): the jam starts where ONE slack cuts the primal step to 1e-3..1e-4 while the
                            multipliers keep taking steps of 0.1..0.7 -- x and s stay, z runs ahead, the dual infeasibility grows from 1e3
                            to 1e10 within eight iterations.  With the cap all twelve members that hit max_iter in the 65 536-member sweep
                            converge in 44..72 iterations WITHOUT any restart; over the 64 fresh batches p99.9 of the iteration count
                            133 -> 106 and the mean batch time 127.7 -> 110.3 ms (8 020 -> 9 290 NLPs/s), the bench batches unchanged
                            (109 ms).  2: mean 138 ms; 4: 128 ms; 0.5: 171 iterations on average                     */
  double slack_corr;     /* slack correction at a rejected first trial point (default 0.9; 0 = off): when the first trial point of the line
                            search (alpha = the step to the boundary) is rejected with theta ABOVE its current value, the same point is
                            tried once more with every inequality slack moved to g(x_trial) -- but never closer to a bound than slack_corr
                            times its linearised distance -- and, if the filter accepts that, taken.  No new linear solve: it removes the
                            part of the Maratos effect that comes from the curvature of the inequality rows, which is what IPOPT's
                            second-order corrections (reference: max_soc 4) would repair with another solve.  The members that set the time
                            of a typical batch are of this type (20..60 iterations at mu = 0.1 with a full step admissible but cut to
                            1/8..1/16): CPU port, the stragglers 137 / 203 / 166 / 157 / 145 -> 57 / 58 / 57 / 93 / 68 iterations; numbers at
                            scale in DESIGN.md 4.2.  Applied at every trial point instead it doubles the mean iteration count       */
  int watchdog;          /* after this many successive iterations whose accepted step length is at most 1/16 of the step to the boundary, the
                            next iteration takes the first trial point (the step to the boundary) without the sufficient-decrease / switching
                            tests -- it must still pass theta <= theta_max and must not be dominated by a filter entry -- and restarts the
                            filter (default 3; 0 = off; cf. IPOPT's watchdog_shortened_iter_trigger).  The members that
                            were left as the slowest of the 65 536-member sweep sat for 50..60 iterations with steps of 1e-3 until the
                            line search failed outright and its fall-back step (alpha_fallback) freed them: CPU port, 161 / 158 / 138 /
                            136 / 133 / 129 / 115 -> 84 / 69 / 81 / 94 / 83 / 67 / 68 iterations, the bench batch unchanged               */
  double barrier_smax;   /* the test that ends a barrier subproblem, E_mu <= kappa_eps mu, uses IPOPT's SCALED optimality error: the dual
                            infeasibility divided by s_d = max(s_max, (|y|_1 + |z|_1) / (m + n_z)) / s_max, the complementarity error by
                            s_c = max(s_max, |z|_1 / n_z) / s_max (Waechter & Biegler 2006, eq. 6; IPOPT's s_max is 100).  Default
                            s_max = 1: with multipliers of the order of the foot forces the unscaled test over-solves every subproblem by
                            two or three iterations.  The FINAL test stays the unscaled tol on pr / du / compl.  Tried in the first half
                            of the round (lab): mean 53.5 -> 48.5 iterations but 1 member in 1000 wandered off at mu = 1e-4; with the
                            watchdog, the slack correction and the later-barrier-problem restart in place it is safe: CPU port on two
                            batches mean 53.2 -> 50.3, median 52 -> 49, slowest 107 -> 95 (numbers at scale in DESIGN.md).  0 = unscaled */
  int factor_fp32;       /* RETIRED in round 5 -- ignored: a non-zero value runs the fp64 factor like 0 (one warning on stderr per process).  Rounds 2-4 had a
                            variant of the stage elimination in single precision on v_mfma_f32_16x16x4_f32 (BASELINE configs[4]'s "fp32
                            MFMA KKT factor"; everything outside the factor fp64).  It reached the same fp64 KKT tolerance but never
                            paid: a stage elimination was 11 % SLOWER than the fp64 one (0.325 vs 0.292 ms of backward sweep per
                            iteration), a warm-started tick needed 9.3 instead of 8.1 iterations, median tick 4.2 vs 4.0 ms and
                            98 % instead of 100 % of the ticks inside the 10 ms budget at batch 256 (profiles/r03_mpc_fp32.json): the
                            sweep is bound by latency and instruction issue, not by the matrix cores, and the 4 x 4 pivot blocks are
                            factored redundantly in registers either way.  configs[4] runs the fp64 factor.  The field
                            stays so that the struct layout of rounds 2-4 callers is unchanged                                       */
  int feas_phase;        /* feasibility (restoration) phase, default 1 (0 in landing_solver_opts_warm).  The reference configures IPOPT, whose
                            answer to a jammed iterate is its restoration phase; here a solve that would end as LANDING_NUMERICAL or
                            LANDING_MAX_ITER continues, once, on the ELASTIC problem: every inequality row may be violated by n, q >= 0 at the
                            price feas_rho (n + q), no objective, same condensation / Riccati sweep / filter line search (the elastic row
                            enters the condensed system exactly like a slack row, solver_kernels.hip el_step).  Outcomes: a feasible point ->
                            the interior-point solve restarts from it with max_iter fresh iterations; a KKT point of the elastic problem
                            with positive violation -> LANDING_INFEASIBLE.  Measured on the reference's production problem (N = 20,
                            non-uniform grid, 1024 drop states per sampling law, CPU port): law "datagen" 984 -> 996 converged + 27
                            certified locally infeasible + 1 undecided; law "main" 1020 converged + 4 certified.  Members that converge
                            without it are untouched (bit-identical)                                                                   */
  double feas_rho;       /* price of a unit of violation in the elastic problem (IPOPT's restoration phase: 1000)                     */
  double feas_cert;      /* 1-norm violation above which the elastic KKT point counts as a certificate (1e-4)                        */
  double delta_floor;    /* proximal term of the terminal-cost form: the first regularisation tried in an iteration is delta_floor instead
                            of 0 (forms with a running cost and the feasibility phase: always 0).  With the terminal cost only, the feet in
                            flight, the lateral translation ... have NO curvature in the Lagrangian: the Newton step along them is set by
                            whatever delta_w the inertia correction last left behind, and the later barrier problems crawl under a saw-tooth
                            delta_w (1e-6 .. 3e-5, wild steps each time it touches the inertia threshold).  A constant (1/2) delta |x - x_k|^2
                            damps exactly those directions; the KKT conditions at the solution do not see it.  Measured (CPU port, 4 x 1024
                            drop states of the bench distribution, N = 40): iterations mean 50.4 -> 48.0, p99 79 -> 61, max 96..116 -> 64..72;
                            N = 20 production grid: law "main" 52.6 -> 49.5, law "datagen" 69.8 -> 62.1; N = 64: 57.2 -> 52.9.  A batch of
                            1024 on 512 slots ends with its slowest late starter, so the tail is what the launch time follows: MI355X, 64 fresh
                            batches of 1024: 95.7 -> 84.9 ms per batch; hold-out 128 batches 131 072 / 131 072 converged, worst member 110
                            iterations (profiles/r03_delta_floor.txt).  3e-3 and above slows every member (linear rate delta / (sigma +
                            delta)); with kappa_eps 10, 4e-4 .. 1e-3 are equivalent within the box-to-box spread; with the final kappa_eps 80 the slow-down
                            already starts at 1e-3 (44.9 instead of 40.1 iterations, one member of 65 536 lost) while 3e-4 and 5e-4 are
                            equivalent (72.7 / 71.3 ms per batch): the default is 3e-4, a factor 3 below that cliff (two hold-out sets of
                            131 072 drop states: all converged, worst member 167 iterations).  0 = the plain IPOPT schedule (first trial
                            delta_w = 0)                                                                                                 */
  int jam_clip;          /* (round 4) the clip_k rule also NEAR feasibility once the classic fraction-to-the-boundary rule has allowed a step
                            to the boundary below 0.02 in this many iterations in a row (default 2; 0 = never): the slow members left after
                            the option changes of round 4 (65..78 iterations against a mean of 34) sit in a later barrier problem with
                            pr ~ 1e-3, where clip_until has switched the rule off, behind one or two slacks that allow steps of 1e-6..1e-2
                            for 25 iterations.  The rule ends with the first iteration whose classic step bound is >= 0.02 again.      */
  int stag_relief;       /* (round 4) delta_floor is divided by 10 per iteration once this many FULL steps (alpha = alpha_dual = 1, first
                            factorisation accepted) in the last barrier problem have not halved max(pr, du) (default 3; 0 = never): against a
                            curvature of ~1e-5 the proximal term turns Newton's method into a linear iteration of rate delta / (sigma + delta)
                            -- one member of the bench batch needed 45 full steps from pr 1e-5 to 1e-6.  Any other step resets it.      */
  int feas_jam;          /* (round 5) the feasibility phase starts EARLY, at most once per solve, when the line search has been jammed:
                            a leaky count of iterations whose accepted step length is below 1e-2 (+1 per such iteration, -2 per other
                            iteration) reaches this number while the primal infeasibility is above 1e-3.  Default 8; 0 = the phase starts
                            only where the solve would end as NUMERICAL / MAX_ITER (rounds 3-4); 0 in landing_solver_opts_warm.  IPOPT
                            enters its restoration phase when the step length falls below alpha_min; rounds 3-4 let such a member run
                            into the iteration limit first -- on the reference's production problem (N = 20, non-uniform grid) the
                            locally infeasible drop states (1.8 % of the data-generation law) took 300 iterations to reach the phase and
                            the slowest member of every batch 600, against 42 for a converging member.  No member of the bench family
                            (N = 40, uniform grid) meets the rule (same iterates, 131 072 drop states)                                  */
  int feas_stat;         /* (round 5) the feasibility phase ends when the 1-norm violation of the inequality rows has stayed within 5 % of a
                            reference value for this many iterations in a row, mu <= 1e-4 and the equality rows (dynamics, slack
                            definitions) hold to 1e-3 (default 25; 0 = off): with a violation
                            above feas_cert the member is reported LANDING_INFEASIBLE (the violation is stationary: what is left of the
                            elastic problem's optimality error is the polishing of its last barrier problems, which took 200+ iterations
                            of wandering full steps for such members), otherwise the solve restarts from the point like after an
                            elastic KKT point.  kkt[0] is the violation in both cases                                                  */
  int kd_clone_after;    /* (round 5, landing_kinodyn_solve_batch only; landing_solve_batch ignores the three kd_ fields) PORTFOLIO of the
                            kinodynamic refinement: after this many rounds of the lock-step loop every member that is still iterating
                            is posed AGAIN, from the callers' initial guess, in three clone slots under three other option sets (the other
                            step rule: clip_k 4 if the caller's is above 4, else 16 | clip_k 16 with bound_push = bound_frac = 0.1 |
                            mu_init = 1; everything else as the caller set it, iteration limit kd_clone_iter); three more waves
                            follow after 2, 3 and 4 times as many rounds for members that had no slot before.  The first member of such
                            a family (the original included) that converges ends the others and is reported under the original's
                            index (x, lam_g, kkt, iters = that member's own count); when nobody converged and the original ends without a
                            decision, a clone's certificate of local infeasibility (status 3) is reported (round 6; it does not end the
                            family early).  0 = off.  Why: which member of a batch is
                            slow depends on the path, not on the problem -- of 5 bench batches of 1024 (law "main") three hold a
                            member that needs 400 .. 1000 iterations (batch 1.1 .. 2.0 s instead of 0.72 s: the loop runs as long
                            as its slowest member), and each of these members converges in 28 .. 73 iterations under at least one
                            of the three sets (profiles/r05_ab_experiments.txt).  landing_kinodyn_solver_opts_default: 56 / 96 / 200
                            (with clip_k 16 and restart_period 30: 0.51 .. 0.59 s per batch on eight seeds, every member decided);
                            0 in landing_solver_opts_default                                                              */
  int kd_clone_max;      /* families per wave (workspace: 4 waves x 3 variants x kd_clone_max member blocks behind the batch)          */
  int kd_clone_iter;     /* iteration limit of a clone (its feasibility phase included in the usual way: limit + limit); 0 = 200      */
  /* ---- round 6: the feasibility phase the way IPOPT runs its restoration phase (fields appended: the layout of rounds 2-5 callers is a prefix).
     Measured with the CPU port on the reference's production problem (N = 20, non-uniform grid, data-generation law, the 16 x 1024 drop states of
     tools/soak.py; tools/dev/feas_lab.py; round 5 in brackets): 15 926 converged (15 858) + 438 certificates, every one a KKT point of the elastic problem
     to tol (488 "certificates", 45 % of them stationary-violation exits) + 13 stalled + 7 undecided (38), slowest member of a batch 386 iterations on
     average (424); GPU numbers: DESIGN.md 4.3 */
  double feas_back;      /* an entry into the phase that is not the last one allowed RETURNS to the interior-point iteration as soon as the violation of the
                            rows (1-norm, equality rows included) has fallen to feas_back times its value at the entry -- IPOPT leaves its restoration
                            phase as soon as the violation has come down and the filter accepts (required_infeasibility_reduction 0.9).  Default 0.2;
                            0 = the phase always runs to a feasible point / an elastic KKT point (rounds 3-5)                                      */
  int feas_max;          /* entries into the phase per solve (default 3; rounds 3-5: 1).  The last one has no early return.  Everything together -- the
                            interior-point iterations, the phases, what follows them -- ends after 3 max_iter iterations at the latest (as before) */
  double feas_delta_dec; /* inside the phase the first regularisation tried is delta_last * feas_delta_dec instead of delta_last * delta_dec (default 0.1;
                            0 = delta_dec), ADAPTED while the phase runs: the factor is squared (never below feas_delta_dec) after an iteration whose first
                            factorisation had the right inertia and replaced by its square root (never above 0.7) after one that needed more; a failed
                            first attempt is followed by delta_last itself, and delta_w = 0 is not probed inside the phase (MI355X, single hard members:
                            2.1-2.3 -> 1.3-1.6 factorisations per iteration, 157 -> 134 ms per batch of the data-generation law).  The elastic problem has no objective: along its flat directions the step is gradient / delta_w, and the
                            iterate reaches the vertex it is heading for only once delta_w has fallen to ~1e-10 -- 38 iterations of halving from the
                            1e-3 the first barrier problems leave behind (traces: tools/dev/feas_trace.py), during which the equality rows are thrown
                            off by orders of magnitude and come back ("wandering", DESIGN.md 4.3 of round 5)                                         */
  double feas_ret_push;  /* the point a phase hands back is taken over WARM: slacks pushed only feas_ret_push off their bounds (bound_push and bound_frac
                            of that re-initialisation), bound multipliers mu / distance with mu = feas_ret_mu, so that the feasibility the phase gained is
                            kept (rounds 3-5 re-initialised as at the start: bound_push 1.0 and mu 0.5 threw the iterate back to a violation of the
                            size it had entered with).  Defaults 0.01 / 0.01; feas_ret_push = 0: the re-initialisation of the start                  */
  double feas_ret_mu;
  int feas_resume;       /* 1 (default): a phase that ends at a stationary violation (feas_stat) which the polishing step below could not turn into a KKT
                            point hands its point back to the interior-point iteration ONCE instead of ending the solve; if that does not lead anywhere
                            either (line search jammed again, iteration limit, no factorisation) the solve ends as LANDING_STALLED.  0: LANDING_STALLED
                            at once                                                                                                               */
  double feas_polish;    /* at the first stationary violation of a phase the regularisation is dropped to this value (default 1e-8; 0 = off) and the
                            stationarity count restarts: of 21 members per 4096 that round 5 reported "infeasible" from such a point, 19 are 5 Newton
                            steps away from the KKT point of the elastic problem (an honest status 3), the others are not stationary at all          */
} landing_solver_opts;

/* status codes written per batch member by landing_solve_batch */
#define LANDING_CONVERGED 0
#define LANDING_MAX_ITER 1
#define LANDING_NUMERICAL 2     /* NaN/Inf or regularisation blow-up; other members unaffected */
#define LANDING_INFEASIBLE 3    /* the feasibility phase (landing_solver_opts::feas_phase) ended at a KKT point of the elastic problem (stationarity,
                                   equality rows and complementarity all within tol) with positive violation: a certificate of LOCAL
                                   infeasibility (what IPOPT reports as "converged to a point of local infeasibility"); x is that point,
                                   kkt[0] its largest violation.  Since round 6 nothing else is reported under this code                     */
#define LANDING_STALLED 4       /* no certificate and no solution: the feasibility phase ended at a violation that had been stationary for feas_stat
                                   iterations at mu <= 1e-4 (equality rows within 1e-3) without being a KKT point of the elastic problem, or the
                                   line search jammed again with no entry into the phase left (feas_max).  x is the last iterate, kkt[0] its largest
                                   violation.  Rounds 3-5 reported the first case as LANDING_INFEASIBLE; a caller may re-pose such a member with
                                   other options -- the outcome says nothing about the problem                                                */

void landing_form_default(landing_form* f);
void landing_solver_opts_default(landing_solver_opts* o);

/* sizes for N intervals */
long long landing_nx(int N);
long long landing_ng(int N);
long long landing_np(int N);
long long landing_nnz_jac(int N);
long long landing_nnz_hess(int N);
/* CCS patterns (colind[nx+1], row[nnz]); equal to casadi_s5 / casadi_s4 for N=20. Host arrays. */
int landing_pattern_jac(int N, long long* colind, long long* row);
int landing_pattern_hess(int N, long long* colind, long long* row);

/* context: device selection, pattern tables, solver workspace (grown on demand) */
landing_ctx* landing_create(int N, int device, const landing_form* form /* NULL = default */);
void landing_destroy(landing_ctx* ctx);
const char* landing_last_error(void);
int landing_device_count(void);

/* Batched function layer.  Any output pointer may be NULL (skipped), as with CasADi's res[i]==0.
 * d_lam_f: [B] or NULL (=1.0); d_lam_g: [B][ng], required for d_hess / d_grad_gamma_*.        */
int landing_eval_batch(landing_ctx* ctx, int B, const double* d_x, const double* d_p,
                       const double* d_lam_f, const double* d_lam_g,
                       double* d_f, double* d_g, double* d_grad_f, double* d_jac, double* d_hess,
                       double* d_grad_gamma_x, double* d_grad_gamma_p, void* stream);
/* Hessian of the Lagrangian of the running-cost formulation (form.run_cost = 1; generate_quadruped_SRBM_CCC.m:81-99, for
 * which the reference ships no generated C and hence no CCS pattern).  Pattern: casadi_s4 plus the 18 N diagonal entries
 * (omega, omega), (v, v) of X_0..X_{N-1} and (f, f) of every stage -- upper triangular CCS, rows sorted inside a column.
 * d_hess_rc: [B][landing_nnz_hess_rc(N)].  With run_cost = 0 the extra entries are zero.                                   */
long long landing_nnz_hess_rc(int N);
int landing_pattern_hess_rc(int N, long long* colind, long long* row);
int landing_eval_hess_rc_batch(landing_ctx* ctx, int B, const double* d_x, const double* d_p, const double* d_lam_f,
                               const double* d_lam_g, double* d_hess_rc, void* stream);
int landing_eval_hess_rc_batch_host(landing_ctx* ctx, int B, const double* x, const double* p, const double* lam_f,
                                    const double* lam_g, double* hess_rc);
/* same, host pointers (copies in and out; used by the CasADi drop-in library) */
int landing_eval_batch_host(landing_ctx* ctx, int B, const double* x, const double* p,
                            const double* lam_f, const double* lam_g,
                            double* f, double* g, double* grad_f, double* jac, double* hess,
                            double* grad_gamma_x, double* grad_gamma_p);
/* lbg/ubg of every member from p (Opti canonicalisation, optistack_internal.cpp:742-856);
 * infinite bounds are +-INFINITY.  d_lbg/d_ubg: [B][ng]. */
int landing_bounds_batch(landing_ctx* ctx, int B, const double* d_p, double* d_lbg, double* d_ubg, void* stream);

/* Batched solver: B independent NLPs.  Outputs (device, any may be NULL except d_x):
 *   d_x [B][nx] solution, d_f [B] objective, d_lam_g [B][ng] multipliers (CasADi sign: >0 at upper
 *   bounds), d_status [B] (LANDING_*), d_iters [B], d_kkt [B][3] = pr_inf, du_inf, compl (unscaled). */
int landing_solve_batch(landing_ctx* ctx, int B, const double* d_p, const double* d_x0,
                        const landing_solver_opts* opts,
                        double* d_x, double* d_f, double* d_lam_g, int* d_status, int* d_iters,
                        double* d_kkt, void* stream);
int landing_solve_batch_host(landing_ctx* ctx, int B, const double* p, const double* x0,
                             const landing_solver_opts* opts,
                             double* x, double* f, double* lam_g, int* status, int* iters, double* kkt);

/* ---- the reference's 21-argument solver function, batched ---------------------------------------
 * [x*, f*] = landingCtrller_IPOPT(Xref, Uref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, q_term_min, q_term_max,
 *                                 qd_term_min, qd_term_max, QN, x0, mu, l_leg_max, f_max, mass, Ib, Ib_inv)
 * (generate_solver/generate_landingCtrller_IPOPT.m:323-327; call sites main_scripts/landing_optimization.m:305-311,
 * generate_data/generate_training_data_automated.m:130-136, generate_data/nn_warmstart.m:193-199).
 * HOST pointers, column-major dense doubles exactly as MATLAB holds them, with a TRAILING batch dimension:
 *   Xref 12 x (N+1) x B | Uref 24 x N x B | dt 1 x N x B | the ten 6-vectors 6 x B | QN 12 x B | x0 nx x B |
 *   mu, l_leg_max, f_max, mass 1 x B | Ib, Ib_inv 3 x B
 * i.e. member b of an argument with n values per member starts at arg + b*n.  Uref is an inactive Opti parameter in the
 * terminal-cost NLP and may be NULL; it is not part of p.  Outputs (host): x_star [nx x B], f_star [B]
 * (may be NULL), status / iters [B] (may be NULL), kkt [3 x B] (may be NULL).
 * landing_pack_args21 only builds p [np x B] in Opti's active-parameter order (pure host code, no device needed). */
typedef struct {
  const double *Xref, *Uref, *dt, *q_min, *q_max, *qd_min, *qd_max, *q_init, *qd_init, *q_term_min, *q_term_max,
      *qd_term_min, *qd_term_max, *QN, *x0, *mu, *l_leg_max, *f_max, *mass, *Ib, *Ib_inv;
} landing_args21;
int landing_pack_args21(int N, int B, const landing_args21* a, double* p);
int landing_solve_args21(landing_ctx* ctx, int B, const landing_args21* a, const landing_solver_opts* opts,
                         double* x_star, double* f_star, int* status, int* iters, double* kkt);
/* the same with the 21 arguments spelled out in the reference's order (what a mex / FFI stub binds) */
int landing_solve_21(landing_ctx* ctx, int B, const double* Xref, const double* Uref, const double* dt,
                     const double* q_min, const double* q_max, const double* qd_min, const double* qd_max,
                     const double* q_init, const double* qd_init, const double* q_term_min, const double* q_term_max,
                     const double* qd_term_min, const double* qd_term_max, const double* QN, const double* x0,
                     const double* mu, const double* l_leg_max, const double* f_max, const double* mass,
                     const double* Ib, const double* Ib_inv, const landing_solver_opts* opts,
                     double* x_star, double* f_star, int* status, int* iters, double* kkt);

/* ---- the 25-argument solver function of the reference's N=41 script ----------------------------------------------------------
 *   [x*, f*] = f_quad_SRBM(Xref, Uref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, c_init, q_term_min, q_term_max, qd_term_min,
 *                          qd_term_max, QX, QN, Qc, Qf, x0, mu, l_leg_max, f_max, mass, Ib, Ib_inv)
 * (generate_solver/generate_quadruped_SRBM_CCC.m; call site analysis/eval_SRBM_CCC.m:72-78).  c_init is an inactive parameter of that
 * script (its constraint is commented out, :98) and may be NULL.  Contexts created with landing_form.run_cost = 2 and that script's
 * kin-box (.05, .05, .27).  Arrays as in landing_args21: column-major with a trailing batch axis; lam_g may be NULL. */
typedef struct {
  const double *Xref, *Uref, *dt, *q_min, *q_max, *qd_min, *qd_max, *q_init, *qd_init, *c_init, *q_term_min, *q_term_max,
      *qd_term_min, *qd_term_max, *QX, *QN, *Qc, *Qf, *x0, *mu, *l_leg_max, *f_max, *mass, *Ib, *Ib_inv;
} landing_args25;
long long landing_np_ccc(int N);                       /* 37N + 112 */
long long landing_ctx_np(const landing_ctx* ctx);      /* length of p for this context's formulation */
int landing_pack_args25(int N, int B, const landing_args25* a, double* p);
int landing_solve_args25(landing_ctx* ctx, int B, const landing_args25* a, const landing_solver_opts* opts,
                         double* x_star, double* f_star, double* lam_g, int* status, int* iters, double* kkt);

/* ---- the same solver function sharded over several devices from the C boundary ---------------------------------
 * Replaces the serial loop of generate_data/generate_training_data_automated.m:38,130-136: the B drop states are cut into
 * contiguous shards (landing_shard_range: sizes differ by at most one), one per entry of the device list; every shard is
 * packed, uploaded, solved and downloaded by its own host thread on its own context (one process, n_dev contexts -- what a mex
 * file can do).  Members are independent: no collective, every thread writes its slice of the caller's HOST buffers.  A device
 * index may repeat (two contexts on one GPU).  Results are bit-identical to the single-context calls above whatever the list.
 * All arguments carry a trailing batch axis as in landing_solve_args21; lam_g [ng x B] (CasADi sign convention) may be NULL. */
typedef struct landing_multi landing_multi;
landing_multi* landing_multi_create(int N, const int* devices, int n_dev, const landing_form* form /* NULL = defaults */);
void landing_multi_destroy(landing_multi* m);
int landing_multi_count(const landing_multi* m);
void landing_shard_range(int B, int n_shards, int i, int* lo, int* hi);
int landing_multi_solve_args21(landing_multi* m, int B, const landing_args21* a, const landing_solver_opts* opts,
                               double* x_star, double* f_star, double* lam_g, int* status, int* iters, double* kkt);
/* one-call form for FFI stubs (matlab/landing_solve_mex.c): contexts are cached inside the library per (N, device list) */
int landing_solve_21_multi(const int* devices, int n_dev, int N, int B, const double* Xref, const double* Uref, const double* dt,
                           const double* q_min, const double* q_max, const double* qd_min, const double* qd_max,
                           const double* q_init, const double* qd_init, const double* q_term_min, const double* q_term_max,
                           const double* qd_term_min, const double* qd_term_max, const double* QN, const double* x0,
                           const double* mu, const double* l_leg_max, const double* f_max, const double* mass,
                           const double* Ib, const double* Ib_inv, const landing_solver_opts* opts,
                           double* x_star, double* f_star, double* lam_g, int* status, int* iters, double* kkt);
void landing_multi_release_cached(void);

/* ---- streaming: consecutive batches through ONE context with the GPU kept busy across batch boundaries (round 6) --------------------
 * Replaces the serial loop of generate_data/generate_training_data_automated.m:38,130-136 for a caller that produces its drop states batch by batch.
 * One launch ends with its slowest member: a batch of 1024 takes 53 ms on the 512 resident slots of an MI355X while the slots are busy for 40 ms on
 * average; with the next batch's launch in flight behind it the freed slots are refilled at once (19.2 k -> 24-25 k NLPs/s, bench.py `streamed`).
 *   landing_stream_create   `lanes` launches may be in flight (0 = 2, 1..8); lane 0 is the context itself, every further lane a child context with the
 *                           same formulation and its own solver workspace (1.1 MB per member at N = 40); each lane has its own non-blocking HIP stream
 *   landing_stream_submit   = landing_solve_batch on the next lane (round robin), asynchronous: returns a ticket >= 0 (or a negative LANDING_E_*).
 *                           `in_stream`: the stream on which d_p / d_x0 become ready (NULL = the default stream); the caller keeps inputs and outputs of
 *                           a submission alive and untouched until it has waited for the ticket; outputs of submissions in flight must not overlap
 *   landing_stream_wait     stream != NULL: that stream waits for the submission (no host synchronisation); NULL: the host waits
 *   landing_stream_sync     ... for everything submitted so far
 * Results are bit-identical to landing_solve_batch one call at a time (members are independent; a lane runs the very same launch).  A stream object is
 * driven by ONE host thread (several objects on one context, or one per context, may be driven by several); B = 0 is a valid (empty) submission.
 *   landing_solve_stream_host   host arrays of ANY number of members cut into chunks (0 = 1024) that go through such a stream, uploads of chunk i + 1
 *                           and downloads of chunk i - 1 under the solve of chunk i: what matlab/landing_solve_mex.c calls for batches above 2048 */
typedef struct landing_stream landing_stream;
landing_stream* landing_stream_create(landing_ctx* ctx, int lanes);
void landing_stream_destroy(landing_stream* s);
int landing_stream_lanes(const landing_stream* s);
long long landing_stream_submit(landing_stream* s, int B, const double* d_p, const double* d_x0, const landing_solver_opts* opts,
                                double* d_x, double* d_f, double* d_lam_g, int* d_status, int* d_iters, double* d_kkt, void* in_stream);
int landing_stream_wait(landing_stream* s, long long ticket, void* stream);
int landing_stream_sync(landing_stream* s, void* stream);
int landing_solve_stream_host(landing_ctx* ctx, int B, int chunk, int lanes, const double* p, const double* x0, const landing_solver_opts* opts,
                              double* x, double* f, double* lam_g, int* status, int* iters, double* kkt);

/* ---- tracking-controller synthesis along solved trajectories -------------------------------------
 * SRBM variational linearisation A (24 x 24), B (24 x 12) (utilities_general/srbm-utilities/generateVariationalDynamics.m:29-62)
 * and the Riccati differential equation Pdot = A'P + PA - P B R^-1 B'P + Q integrated backward along the sampled
 * trajectory (generateRiccatiIntegrator.m:24-62, quadruped_SRBM_NLP.m:428-503), B trajectories at once.
 *   d_xref [B][n][24] = [p rpy omega v pf(12)] and d_fref [B][n][12] at the n grid points (device);
 *   Ib3x3: body inertia, row-major 3x3 (host); Q, F: 24 x 24 row-major (host); r_diag: diagonal of R (12, host);
 *   rk4 = 0: the reference's backward step `P0 = Pf + dt*k1` (:47); 1: classical RK4 (:43-46)
 * Outputs (device, any may be NULL): d_P [B][n][576] row-major with P[n-1] = F, d_K [B][n][12*24] = R^-1 B'P (tracking
 * gains), d_A [B][n][576], d_B [B][n][288]. */
int landing_riccati_gains_batch(landing_ctx* ctx, int B, int n, const double* d_xref, const double* d_fref,
                                const double* Ib3x3, double mass, const double* Q, const double* r_diag, const double* F,
                                double dt, int rk4, double* d_P, double* d_K, double* d_A, double* d_B, void* stream);

/* Receding-horizon loop (BASELINE configs[4]: 100 Hz warm-started re-solves).  One control tick =
 *   landing_mpc_shift(previous solution, measured states) -> x0, p;  landing_solve_batch(p, x0, warm-start options).
 * landing_mpc_shift: x0 = previous solution advanced by one stage (last column held) with d_state [B][12] = measured
 * [q; qd] in X(:,0); q_init / qd_init of d_p [B][np] are overwritten in place.  d_x0 must not alias d_x_prev.
 * Warm-start options: landing_solver_opts_warm() -- the reference's `_ws` variant sets bound_push = bound_frac = 5e-3
 * (generate_landingCtrller_IPOPT_warmstart.m:246-247); with a shifted solution 1e-4 for both and mu_init = 1e-4 converge in
 * 8 iterations instead of 27 (tests/dev, round 2), which is what fits a 10 ms tick; max_iter = 14 bounds the tick time
 * (real-time iteration: an unconverged member keeps its iterate and continues at the next tick). */
int landing_mpc_shift(landing_ctx* ctx, int B, const double* d_x_prev, const double* d_state, double* d_p, double* d_x0, void* stream);
void landing_solver_opts_warm(landing_solver_opts* o);

/* ---- floating-base rigid-body routines of the 18-body model ------------------------------------
 * landing_rbd_model: the kinematic tree in compact form -- Xtree = plux(E, r) per body (get_robot_model.m:134-234), link
 * inertias as (mass, m*com, rotational inertia about the link origin [xx xy xz yy yz zz]) -- built by the caller
 * (landing-controller_amd/rbd.py restates the reference's 'quad3D' model) and uploaded once with landing_rbd_set_model.
 * landing_fb_dynamics_batch: tau = H(q) qdd + C(q, qd, f_foot) at npts configurations (spatial_v2 HandC.m:14-62 with the
 *   foot forces of casadi_compatible_dynamics.m:53-60; q, qd, tau [npts][18], f_foot [npts][12] world-frame forces or NULL);
 *   outputs (device, any may be NULL): H [npts][18][18], C [npts][18], qdd [npts][18] (needs tau), the linearisation of the
 *   forward dynamics A = d qdd / d [q; qd] [npts][18][36] (needs tau) and Hinv = d qdd / d tau [npts][18][18].
 *   fd_h <= 0: A is EXACT -- -H^-1 d ID(q, qd, qdd, f) / d [q; qd] with the 36 tangent directions pushed through the
 *   inverse-dynamics recursion in forward mode (dual numbers), what the reference obtains from CasADi's algorithmic
 *   differentiation of casadi_compatible_dynamics.m; fd_h > 0: central differences of the forward dynamics with that step
 *   (round-2 first half; kept as a cross-check).
 * landing_kinodyn_rows_batch: the rows the kinodynamic refinement adds per stage (landing_optimization.m:152-189) at npts
 *   (member, stage) points: q6 = [pos; rpy (XYZ convention)] [npts][6], c / f [npts][12] feet and forces, jpos [npts][12];
 *   outputs fk [npts][12] (get_forward_kin_foot.m), fk_err = c - fk, tau = J_f'(-R_world_to_body f) (get_foot_jacobians_mc.m). */
typedef struct {
  int parent[18], jtype[18];            /* parent 1-based (0 = fixed base); joint 0..5 = Rx Ry Rz Px Py Pz */
  double E[18][9], r[18][3];
  double m[18], h[18][3], I[18][6];
  int b_foot[4]; double foot_r[4][3];
  double l1, l2, l3, l4;
} landing_rbd_model;
int landing_rbd_set_model(landing_ctx* ctx, const landing_rbd_model* model);
int landing_fb_dynamics_batch(landing_ctx* ctx, int npts, const double* d_q, const double* d_qd, const double* d_tau, const double* d_f_foot,
                              double* d_H, double* d_C, double* d_qdd, double* d_A, double* d_Hinv, double fd_h, void* stream);
/* leg inverse kinematics (damped Newton on the tree FK, `iters` steps, clamped to jpos_min/max[3] per leg): joint angles
 * d_jpos [npts][12] with FK([q6; jpos]) = d_c, residual |FK - c| per leg in d_res [npts][4] (may be NULL) */
int landing_leg_ik_batch(landing_ctx* ctx, int npts, const double* d_q6, const double* d_c, const double* jpos_min3, const double* jpos_max3,
                         int iters, double* d_jpos, double* d_res, void* stream);
int landing_kinodyn_rows_batch(landing_ctx* ctx, int npts, const double* d_q6, const double* d_c, const double* d_f, const double* d_jpos,
                               double* d_fk, double* d_fk_err, double* d_tau, void* stream);

/* ---- function layer of the kinodynamic refinement NLP (landing_optimization.m:38-189) -------------------------------
 * x = [X(:) (12 x (N+1)); jpos(:) (12 x N); U(:) (24 x N, c then f_grf per column)] -- the script's declaration order (:39-42), nx = 48 N + 12;
 * g = [q(:,1); qdot(:,1); c(:,1) | q(:,N), q(:,N), qdot(:,N), qdot(:,N) | per interval the rows of :113-189 in the script's order], every
 * expression the script bounds twice appears twice: ng = 48 + 141 (N - 1) + 117.  N here = number of intervals (the script's N - 1 = 20).
 * landing_kinodyn_nlp_eval: d_g [B][ng] and / or d_jac [B][N][141][72] -- the Jacobian block of interval k over
 *   w = [X_k (12), c_k (12), f_k (12), jpos_k (12), X_k+1 (12), c_k+1 (12)] (rows beyond 117 of the last interval and its c_k+1 columns are not
 *   written / zero); the 48 boundary rows are coordinate picks.  Exact derivatives (forward mode, one tangent per thread), the way the
 *   reference gets them from CasADi.  Needs landing_rbd_set_model (forward kinematics of get_forward_kin_foot.m, leg lengths of
 *   get_foot_jacobians_mc.m).  Bounds and the solve itself are not part of this entry point.                          */
typedef struct { double dt[64]; double mass, Ib[3], Ib_inv[3], mu; } landing_kinodyn_params;
int landing_kinodyn_nlp_dims(int N, long long* nx, long long* ng);
int landing_kinodyn_nlp_eval(landing_ctx* ctx, int B, int N, const double* d_x, const landing_kinodyn_params* prm, double* d_g, double* d_jac, void* stream);
/* Hessian of lam_g' g per interval: d_hess [B][N][72][72] (symmetric blocks over w; the Hessian of the Lagrangian of the NLP is their sum at
 * the positions of w in x, plus the constant terminal-cost term 2 diag(QN) on X(:, N+1)).  Exact (second-order forward mode, one pair of
 * directions per thread over the 561 structurally non-zero pairs of a block); a first implementation for the solver of the next round to be
 * checked against: 60 ms per 1024 members at N = 20 on an MI355X. */
int landing_kinodyn_nlp_hess(landing_ctx* ctx, int B, int N, const double* d_x, const landing_kinodyn_params* prm, const double* d_lam_g, double* d_hess, void* stream);

/* ---- the kinodynamic refinement SOLVE ---------------------------------------------------------------------------------
 * The step after the SRBM solve in every production caller (main_scripts/landing_optimization.m:300-322 SRBM solution as the initial guess,
 * :360-376 / :398-435 the solves; generate_data/generate_training_data_automated.m:125-175).  The reference hands this NLP to KNITRO through
 * the solver function of generate_solver/generate_landingCtrller_KNITRO.m:365-377,
 *   [x*, f*] = landingCtrller_KNITRO(Xref, Uref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, c_init, q_term_min, q_term_max, qd_term_min,
 *                                    qd_term_max, QN, x0, jpos_min, jpos_max, kin_box, mu, l_leg_max, mass, Ib, Ib_inv)
 * (KNITRO is a commercial solver; its generated artefacts are missing blobs of the reference tree).  Here: the interior-point method of
 * landing_solve_batch on this NLP's stage structure (state (X_k, c_k), controls (f_k, jpos_k, c_k+1): the joint angles are stage-local and are
 * eliminated inside the stage), exact first and second derivatives from landing_kinodyn_nlp_eval / _hess, B members per call.
 *   landing_kinodyn_form     the literals of the script's constraint set (:139-189)
 *   landing_kinodyn_bounds   lbg / ubg [ng x B] from the script's arguments (host arrays, trailing batch axis; Opti's canonicalisation,
 *                            optistack_internal.cpp:742-856: a side without decision variables becomes a bound of g = the other side, an inequality with
 *                            variables on both sides becomes g = lhs - rhs in (-inf, 0].  Round 6 settled the 8 friction rows of an interval written with
 *                            `>=`, `f_xy >= -0.71 mu f_z` (:176,178): MATLAB's a >= b is CasADi's le(b, a), so Opti holds them as
 *                            g = -0.71 mu f_z - f_xy in (-inf, 0] -- which is what the rows, jac_g and lam_g of this library now are (rounds 3-5 emitted
 *                            f_xy + 0.71 mu f_z in [0, inf): same feasible set, opposite sign of g, of the Jacobian rows and of lam_g for those rows))
 *   landing_kinodyn_solve_batch   device pointers: d_lbg, d_ubg [B][ng]; d_cost [B][24] = QN (12) | Xref(:, end) (12) (terminal cost :83-86);
 *                            d_x0 [B][nx]; dt, mass, Ib, Ib_inv, mu shared by the batch (prm).  Outputs as landing_solve_batch: d_x [B][nx],
 *                            d_f [B], d_lam_g [B][ng] (CasADi sign), d_status [B] (LANDING_*), d_iters [B], d_kkt [B][3] (pr, du, compl unscaled)
 *   landing_solve_kinodyn_24 the 24 arguments in the reference's order, HOST pointers, column-major with a trailing batch axis
 *                            (Xref 12 x (N+1) x B, Uref 24 x N x B (inactive, may be NULL), dt 1 x N x B, 6-vectors 6 x B, c_init 12 x B, QN 12 x B,
 *                            x0 nx x B, jpos_min / jpos_max 12 x B, kin_box 2 x B, mu / l_leg_max / mass 1 x B, Ib / Ib_inv 3 x B); dt, mu, mass, Ib,
 *                            Ib_inv must be the same for every member of one call (they are constants of every caller in the reference).
 * Needs landing_rbd_set_model.  N = number of intervals (the script's N - 1 = 20), N <= 64.
 * Options: landing_kinodyn_solver_opts_default = landing_solver_opts_default with max_iter 500, bound_push = bound_frac = 0.01, mu_init 0.1, theta_mu 1.5
 * and feas_jam 0; round 5 (late): clip_k 16, restart_period 30 and the portfolio kd_clone_after 56 / kd_clone_max 96 / kd_clone_iter 200 (see those fields: the
 * members still iterating after 56 rounds race three clones of themselves; d_iters of such a member is the winner's own count).  The feasibility (restoration) phase of landing_solve_batch exists here too (round 5; feas_phase, feas_rho, feas_cert, feas_stat): a member
 * that would end as NUMERICAL / MAX_ITER continues on the elastic problem and ends as a KKT point of the original NLP, with a certificate of local
 * infeasibility (status 3), or undecided.  The host-array entry points re-solve the few members a first pass leaves undecided with two other slack / barrier
 * initialisations when the caller passes no options (retry ladder).                                                                              */
typedef struct {
  double comp_eps, slip_eps;      /* :139  f_z c_z <= 1e-3;  :142-143  |f_z (c+ - c)| <= 1e-3 */
  double fk_band;                 /* :186-187  |c - FK(q, jpos)| <= 0.01 */
  double kin_box_x0, kin_box_y0;  /* :151-152  0.125 + kin_box(1), 0.10 + kin_box(2)  (generate_landingCtrller_KNITRO.m:110 uses 0.125 for y as well) */
  double kin_box_y_in;            /* :159-163  inner lateral bound 0.05 */
  double kin_box_z_lo, kin_box_z_hi;  /* :153-154  -0.4, -0.075 */
  double tau_max[3];              /* model.tauMax = gr .* motorTauMax (get_robot_model.m:236-240): 18, 18, 27.99 */
} landing_kinodyn_form;
void landing_kinodyn_form_default(landing_kinodyn_form* f);       /* the literals of main_scripts/landing_optimization.m (kin_box_y = 0.10 + kin_box(2), :150) */
void landing_kinodyn_form_knitro(landing_kinodyn_form* f);        /* ... of generate_solver/generate_landingCtrller_KNITRO.m (kin_box_y = 0.125 + kin_box(2), :154): what
                                                                      landing_solve_kinodyn_24[_on] use when no form is passed -- they stand for the function that script builds */
void landing_kinodyn_solver_opts_default(landing_solver_opts* o);
void landing_kinodyn_solver_opts_warm(landing_solver_opts* o);     /* the `_ws` re-solve from a previous solution (landing_optimization.m:395-435, generate_landingCtrller_KNITRO_warmstart.m):
                                                                      bound_push = bound_frac = mu_init = 1e-6, no cold-start rules, no portfolio, max_iter 100: 4 iterations on average (cold start: 31) */
int landing_kinodyn_bounds(int N, int B, const landing_kinodyn_form* form, const double* q_init, const double* qd_init, const double* c_init,
                           const double* q_min, const double* q_term_min, const double* q_term_max, const double* qd_term_min, const double* qd_term_max,
                           const double* jpos_min, const double* jpos_max, const double* kin_box, const double* l_leg_max, double* lbg, double* ubg);
int landing_kinodyn_solve_batch(landing_ctx* ctx, int B, int N, const landing_kinodyn_params* prm, const double* d_lbg, const double* d_ubg,
                                const double* d_cost, const double* d_x0, const landing_solver_opts* opts,
                                double* d_x, double* d_f, double* d_lam_g, int* d_status, int* d_iters, double* d_kkt, void* stream);
/* Host arrays.  With opts = NULL (the defaults) the members the first pass leaves undecided -- iteration limit or numerical failure -- are solved
   again from the same initial guess: pass 2 with bound_push = bound_frac = 0.1, pass 3 with mu_init = 0.02; a member keeps its first decided
   outcome and iters counts all its passes (measured on 1024 drop states of the hard sampling law: 19 -> 5 undecided).  With options given: one pass. */
int landing_kinodyn_solve_batch_host(landing_ctx* ctx, int B, int N, const landing_kinodyn_params* prm, const double* lbg, const double* ubg,
                                     const double* cost, const double* x0, const landing_solver_opts* opts,
                                     double* x, double* f, double* lam_g, int* status, int* iters, double* kkt);

int landing_solve_kinodyn_24(landing_ctx* ctx, int N, int B, const landing_kinodyn_form* form /* NULL = landing_kinodyn_form_knitro */,
                             const double* Xref, const double* Uref, const double* dt, const double* q_min, const double* q_max, const double* qd_min,
                             const double* qd_max, const double* q_init, const double* qd_init, const double* c_init, const double* q_term_min,
                             const double* q_term_max, const double* qd_term_min, const double* qd_term_max, const double* QN, const double* x0,
                             const double* jpos_min, const double* jpos_max, const double* kin_box, const double* mu, const double* l_leg_max,
                             const double* mass, const double* Ib, const double* Ib_inv, const landing_solver_opts* opts,
                             double* x_star, double* f_star, double* lam_g, int* status, int* iters, double* kkt);
/* the reference's 'quad3D' tree with the 'mc3D' parameters in the model struct above, for C / mex callers (rbd.py builds the same in Python), and the
 * one-call form of landing_solve_kinodyn_24 for FFI stubs: a context per (N, device) with that model is cached inside the library */
void landing_rbd_model_mc3d(landing_rbd_model* model);
int landing_solve_kinodyn_24_on(int device, int N, int B, const double* Xref, const double* Uref, const double* dt, const double* q_min, const double* q_max,
                                const double* qd_min, const double* qd_max, const double* q_init, const double* qd_init, const double* c_init,
                                const double* q_term_min, const double* q_term_max, const double* qd_term_min, const double* qd_term_max, const double* QN,
                                const double* x0, const double* jpos_min, const double* jpos_max, const double* kin_box, const double* mu, const double* l_leg_max,
                                const double* mass, const double* Ib, const double* Ib_inv, const landing_solver_opts* opts,
                                double* x_star, double* f_star, double* lam_g, int* status, int* iters, double* kkt);
/* ... with a constraint-set form (NULL = landing_kinodyn_form_knitro): the same call, `form` in front of the 24 arguments */
int landing_solve_kinodyn_24_on_form(int device, int N, int B, const landing_kinodyn_form* form, const double* Xref, const double* Uref, const double* dt,
                                     const double* q_min, const double* q_max, const double* qd_min, const double* qd_max, const double* q_init, const double* qd_init,
                                     const double* c_init, const double* q_term_min, const double* q_term_max, const double* qd_term_min, const double* qd_term_max,
                                     const double* QN, const double* x0, const double* jpos_min, const double* jpos_max, const double* kin_box, const double* mu,
                                     const double* l_leg_max, const double* mass, const double* Ib, const double* Ib_inv, const landing_solver_opts* opts,
                                     double* x_star, double* f_star, double* lam_g, int* status, int* iters, double* kkt);
/* CCS patterns of this NLP in CasADi's compressed form, which = 0: jac_g_x (ng x nx), 1: upper triangle of hess_gamma_x_x; colind [nx + 1],
 * row [*nnz] (pass row = NULL to get the count first).  Derived from the derivative kernels themselves (device needed). */
int landing_kinodyn_pattern(landing_ctx* ctx, int N, int which, long long* colind, long long* row, long long* nnz);
/* Diagnostic: the solver's own table of the structural non-zeros of a Jacobian block's inequality rows (rows 12.. of the 141 x 72 block of an interval; the forward
 * sweep forms ds = J dx over these entries only, round 6): counts for an interval that is not the last / for the last one (529 / 457 with the Mini-Cheetah model), and,
 * when `entries` is not NULL, the table itself as (row, column) byte pairs, the first *nnz_mid pairs for a middle interval, then *nnz_last for the last
 * (room for 2 x 1280 bytes).  Built once per context by asking the Jacobian kernel (device needed); it must agree with landing_kinodyn_pattern. */
int landing_kinodyn_block_nonzeros(landing_ctx* ctx, int* nnz_mid, int* nnz_last, unsigned char* entries);

/* ---- CasADi-external face of the kinodynamic refinement NLP (round 6) ------------------------------------------------------------------------
 * What landingCtrller_KNITRO_mi355x.so (csrc/casadi_abi.cpp with -DLANDING_KD=1) forwards to: the seven nlp_* functions of the library the reference generates
 * for this NLP (generate_solver/generate_landingCtrller_KNITRO.m:360-377; its own artefacts are missing blobs) for ONE problem on host arrays.
 *   p   the ACTIVE Opti parameters in declaration order (:51-82): Xref 12(N+1) | dt N | q_init 6 | qd_init 6 | c_init 12 | jpos_min 12 | jpos_max 12 |
 *       q_term_min 6 | q_term_max 6 | qd_term_min 6 | qd_term_max 6 | q_min 6 | QN 12 | mu | l_leg_max | mass | Ib 3 | Ib_inv 3 | kin_box 2, np = 13 N + 113
 *       (Uref, q_max, qd_min, qd_max are declared and never used: inactive, dropped -- the rule SURVEY row a2 verified on landingCtrller_IPOPT.c)
 *   g   rows of landing_kinodyn_nlp_eval = Opti's canonical forms; jac / hess: CCS nonzeros in the patterns of landing_kinodyn_casadi_pattern
 *       (= landing_kinodyn_pattern, pointers stay valid while the context lives); lam_g in CasADi's sign; lbg / ubg from p: landing_kinodyn_casadi_bounds
 *   any output may be NULL; lam_f NULL = 1; grad_gamma_p: exact for Xref / QN, central differences for dt, mass, Ib, Ib_inv, mu, 0 for parameters of the bounds */
long long landing_kinodyn_casadi_np(int N);
int landing_kinodyn_casadi_offsets(int N, long long off[19]);
int landing_kinodyn_casadi_bounds(int N, const landing_kinodyn_form* form, const double* p, double* lbg, double* ubg);
int landing_kinodyn_casadi_pattern(landing_ctx* ctx, int N, int which, const long long** colind, const long long** row, long long* nnz);
int landing_kinodyn_casadi_eval_host(landing_ctx* ctx, int N, const double* x, const double* p, const double* lam_f, const double* lam_g,
                                     double* f, double* g, double* grad_f, double* jac, double* hess, double* grad_gamma_x, double* grad_gamma_p);
void landing_kinodyn_casadi_release(landing_ctx* ctx);

/* ---- SQP (Gauss-Newton / iLQR) loop on the 18-DoF model (BASELINE configs[3]) --------------------------------
 * Trajectory-tracking problem per member: state x = [q; qd] (36), control u = the 12 joint torques (base unactuated), known foot
 * forces f_k, explicit Euler  q+ = q + dt qd, qd+ = qd + dt qdd(q, qd, [0; u], f)  (the discretisation of the SRBM NLP,
 * generate_landingCtrller_IPOPT.m:127-130), cost  sum_k 1/2 |x_k - xref_k|^2_Q + 1/2 |u_k|^2_R + 1/2 |x_N - xref_N|^2_QN  with
 * diagonal weights.  The reference has the dynamics (casadi_compatible_dynamics.m) but no loop around them; one iteration here is
 *   landing_fb_dynamics_batch(npts = B N, q/qd/tau = knots of (x, u), ..., d_A, d_Hinv, fd_h = 0)   exact linearisation,
 *   landing_wb_backward     LQ subproblem by a Riccati recursion (one wavefront per member, 36 x 36 value function in LDS):
 *                           feedback gains d_K [B][N][12][36], feed-forward d_kff [B][N][12], expected decrease d_dV [B][2]
 *                           (cost change ~ alpha dV[0] + alpha^2 dV[1]), d_ok [B] = 0 where a stage Hessian was not positive definite
 *                           (raise reg), and
 *   landing_wb_rollout      the nonlinear dynamics under u = u_k + alpha kff_k + K_k (x - x_k) for nalpha step lengths, one thread
 *                           per (step length, member): trajectories d_xnew [nalpha][B][N+1][36], d_unew [nalpha][B][N][12] and their
 *                           costs d_cost [nalpha][B] (inf where the dynamics failed); d_K = d_kff = NULL rolls out u open loop
 *                           (initialisation: a dynamically consistent trajectory and its cost).
 * The host keeps the best step length per member and adapts reg (landing-controller_amd/wb.py). */
int landing_wb_backward(landing_ctx* ctx, int B, int N, double dt, double reg, const double* d_x, const double* d_u, const double* d_xref,
                        const double* d_A, const double* d_Hinv, const double* Q36, const double* R12, const double* QN36,
                        double* d_K, double* d_kff, double* d_dV, int* d_ok, void* stream);
int landing_wb_rollout(landing_ctx* ctx, int B, int N, int nalpha, const double* d_alphas, double dt, const double* d_x, const double* d_u,
                       const double* d_xref, const double* d_f_foot, const double* d_K, const double* d_kff, const double* Q36, const double* R12,
                       const double* QN36, double* d_xnew, double* d_unew, double* d_cost, void* stream);

/* Integrator of landing_wb_backward / landing_wb_rollout: 0 (default) explicit Euler as above; 1 semi-implicit (symplectic) Euler,
 *   qd+ = qd + dt qdd(q, qd, u),  q+ = q + dt qd+   (the scheme the reference compares with explicit Euler in
 * test_scripts/test_integrationDifference.m:30-40).  Explicit Euler on this model is unstable beyond dt ~ 1.5 ms (light leg links under the foot
 * forces); the semi-implicit scheme runs the loop on the landing NLP's own 15 ms grid.  The linearisation of the backward pass follows:
 * A = [I 0; 0 0] + [dt A_qd; A_qd],  B = [dt B_qd; B_qd]  with A_qd = [dt dqdd/dq, I + dt dqdd/dqd], B_qd = dt Hinv(:, 6:18).
 * landing_wb_select: after ONE landing_wb_rollout launch with all step lengths, every member keeps the first rollout of the list that lowers its
 * cost (d_x, d_u, d_cost updated in place, d_step [B] = the step length taken, 0 = none; members with d_ok = 0 keep their trajectory). */
int landing_wb_set_integrator(landing_ctx* ctx, int semi_implicit);
/* two-stage search: roll out the full step first (the usual winner), select, then landing_wb_skip_taken(d_step) + a rollout of the shorter step
 * lengths, in which every member that already took a step returns at once, + landing_wb_select with nalpha NEGATED (= keep earlier choices) */
int landing_wb_skip_taken(landing_ctx* ctx, const double* d_step);
int landing_wb_select(landing_ctx* ctx, int B, int N, int nalpha, const double* d_alphas, const int* d_ok, const double* d_xnew, const double* d_unew,
                      const double* d_costnew, double* d_x, double* d_u, double* d_cost, double* d_step, void* stream);

/* development aid: d_prof [B][16] doubles receives per-member phase timers of the next solves (100 MHz
 * wall-clock ticks: eval, error, sigma/rho, backward, forward, dual, line search, accept; then counts of
 * factorisations, trial points, iterations); NULL disables. */
int landing_set_profile_buffer(landing_ctx* ctx, double* d_prof);
/* diagnostic: device pointer and per-member stride (doubles) of the solver workspace left by the last landing_solve_batch
 * (layout: landing-controller_amd/csrc/solver_kernels.hip, carve()) */
int landing_debug_workspace(landing_ctx* ctx, double** d_ws, unsigned long long* stride);

/* name of the dominant kernels (for profilers) and per-launch algorithmic bytes of the sweep */
const char* landing_kernel_name_sweep(void);
long long landing_sweep_bytes_per_member(int N);

#ifdef __cplusplus
}
#endif
#endif
