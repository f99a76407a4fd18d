"""ctypes bindings for the CPU oracle (TEST INFRASTRUCTURE, not product code).

* ``Oracle(N)``      -- liblanding_oracle.so, the N-generic plain-C restatement (landing_oracle.c)
* ``RefOracle()``    -- oracle/_ref/liblanding_ref.so, the reference's own CasADi-generated C
                        (optimizations/landing/codegen_casadi/landingCtrller_IPOPT.c, N=20 only),
                        called through the CasADi external ABI (landingCtrller_IPOPT.c:10916-10993).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ll = C.c_longlong


def build(quiet=True):
    """make -C oracle (also builds oracle/_ref when /root/reference is present)."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


class _Form(C.Structure):
    _fields_ = [("N", C.c_int), ("kin_box", C.c_double * 3), ("kin_z_off", C.c_double),
                ("comp_eps", C.c_double), ("slip_eps", C.c_double), ("run_cost", C.c_int), ("QX", C.c_double * 12),
                ("Qc", C.c_double * 3), ("Qf", C.c_double * 3), ("f_ref", C.c_double * 3), ("p_hip", C.c_double * 12)]


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Oracle:
    def __init__(self, N, kin_box=None, run_cost=None, ccc_params=False):
        path = os.path.join(HERE, "liblanding_oracle.so")
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        for name in ("lo_nx", "lo_ng", "lo_np", "lo_nnz_jac", "lo_nnz_hess", "lo_nnz_hess_rc"):
            getattr(lib, name).restype = _ll
            getattr(lib, name).argtypes = [C.c_int]
        self.N = N
        self.form = _Form()
        lib.lo_form_default(C.byref(self.form), N)
        if kin_box is not None:
            for i in range(3):
                self.form.kin_box[i] = kin_box[i]
        if run_cost is not None:      # dict(QX=[12], Qc=[3], Qf=[3], f_ref=[3]): generate_quadruped_SRBM_CCC.m:81-89
            self.form.run_cost = 1
            for i in range(12):
                self.form.QX[i] = run_cost["QX"][i]
            for i in range(3):
                self.form.Qc[i] = run_cost["Qc"][i]; self.form.Qf[i] = run_cost["Qf"][i]; self.form.f_ref[i] = run_cost.get("f_ref", (0, 0, 0))[i]
        if ccc_params:                # the N=41 script's own parameter vector (lo_param_offsets_form): weights / force reference from p
            self.form.run_cost = 2
        lib.lo_np_form.restype = _ll
        self.nx, self.ng, self.np_ = lib.lo_nx(N), lib.lo_ng(N), lib.lo_np_form(C.byref(self.form))
        self.nnz_jac, self.nnz_hess = lib.lo_nnz_jac(N), lib.lo_nnz_hess(N)
        self._F = C.byref(self.form)

    # -- patterns ---------------------------------------------------------------------
    def pattern_jac(self):
        ci = np.zeros(self.nx + 1, np.int64); r = np.zeros(self.nnz_jac, np.int64)
        self.lib.lo_pattern_jac(self.N, ci.ctypes.data_as(C.POINTER(_ll)), r.ctypes.data_as(C.POINTER(_ll)))
        return ci, r

    def pattern_hess(self):
        ci = np.zeros(self.nx + 1, np.int64); r = np.zeros(self.nnz_hess, np.int64)
        self.lib.lo_pattern_hess(self.N, ci.ctypes.data_as(C.POINTER(_ll)), r.ctypes.data_as(C.POINTER(_ll)))
        return ci, r

    def param_offsets(self):
        names = ["Xref", "dt", "q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "q_term_min",
                 "q_term_max", "qd_term_min", "qd_term_max", "QN", "mu", "l_leg_max", "f_max", "mass",
                 "Ib", "Ib_inv", "np", "Uref", "QX", "Qc", "Qf"]
        arr = (C.c_int * len(names))()
        self.lib.lo_param_offsets_form(C.byref(self.form), arr)
        return dict(zip(names, list(arr)))

    # -- callbacks --------------------------------------------------------------------
    def f(self, x, p):
        out = C.c_double()
        self.lib.lo_nlp_f(self._F, _p(x), _p(p), C.byref(out))
        return out.value

    def grad_f(self, x, p):
        out = C.c_double(); g = np.zeros(self.nx)
        self.lib.lo_nlp_grad_f(self._F, _p(x), _p(p), C.byref(out), _p(g))
        return out.value, g

    def g(self, x, p):
        g = np.zeros(self.ng)
        self.lib.lo_nlp_g(self._F, _p(x), _p(p), _p(g))
        return g

    def jac_g(self, x, p):
        g = np.zeros(self.ng); j = np.zeros(self.nnz_jac)
        self.lib.lo_nlp_jac_g(self._F, _p(x), _p(p), _p(g), _p(j))
        return g, j

    def hess_l(self, x, p, lam_f, lam_g):
        h = np.zeros(self.nnz_hess)
        self.lib.lo_nlp_hess_l(self._F, _p(x), _p(p), C.c_double(lam_f), _p(lam_g), _p(h))
        return h

    def pattern_hess_rc(self):
        n = self.lib.lo_nnz_hess_rc(self.N)
        ci = np.zeros(self.nx + 1, np.int64); r = np.zeros(n, np.int64)
        self.lib.lo_pattern_hess_rc(C.c_int(self.N), ci.ctypes.data_as(C.POINTER(_ll)), r.ctypes.data_as(C.POINTER(_ll)))
        return ci, r

    def hess_l_rc(self, x, p, lam_f, lam_g):
        h = np.zeros(self.lib.lo_nnz_hess_rc(self.N))
        self.lib.lo_nlp_hess_l_rc(self._F, _p(np.ascontiguousarray(x, float)), _p(np.ascontiguousarray(p, float)), C.c_double(lam_f), _p(np.ascontiguousarray(lam_g, float)), _p(h))
        return h

    def grad(self, x, p, lam_f, lam_g):
        f = C.c_double(); g = np.zeros(self.ng); gx = np.zeros(self.nx); gp = np.zeros(self.np_)
        self.lib.lo_nlp_grad(self._F, _p(x), _p(p), C.c_double(lam_f), _p(lam_g), C.byref(f), _p(g), _p(gx), _p(gp))
        return f.value, g, gx, gp

    def bounds(self, p):
        lb = np.zeros(self.ng); ub = np.zeros(self.ng)
        self.lib.lo_bounds(self._F, _p(p), _p(lb), _p(ub))
        return lb, ub

    def stage_eval(self, k, x, p, lam=None, want_J=True):
        """dense per-stage blocks: g[104], J[104,60], H[60,60] (locals: X_k,U_k,X_{k+1},c_{k+1})."""
        g = np.zeros(104); J = np.zeros((104, 60)) if want_J else None
        H = np.zeros((60, 60)) if lam is not None else None
        self.lib.lo_stage_eval(self._F, C.c_int(k), _p(x), _p(p), _p(lam), _p(g), _p(J), _p(H))
        return g, J, H

    def kkt(self, x, p, lam_g):
        out = np.zeros(3)
        self.lib.lo_kkt(self._F, _p(x), _p(p), _p(lam_g), _p(out))
        return out


class _SolverOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int), ("mu_init", C.c_double), ("bound_push", C.c_double),
                ("bound_frac", C.c_double), ("kappa_eps", C.c_double), ("kappa_mu", C.c_double), ("theta_mu", C.c_double),
                ("max_resets", C.c_int), ("reset_du", C.c_double), ("delta_init", C.c_double), ("delta_inc_first", C.c_double),
                ("delta_inc", C.c_double), ("delta_dec", C.c_double), ("tau_min", C.c_double), ("alpha_fallback", C.c_double),
                ("restart_period", C.c_int), ("reset_delta", C.c_double), ("barrier_smax", C.c_double), ("watchdog", C.c_int), ("slack_corr", C.c_double), ("dual_step_cap", C.c_double), ("fresh_restart", C.c_int), ("theta_floor", C.c_double), ("clip_k", C.c_int), ("clip_until", C.c_double),
                ("feas_phase", C.c_int), ("feas_rho", C.c_double), ("feas_cert", C.c_double), ("delta_floor", C.c_double), ("jam_clip", C.c_int), ("stag_relief", C.c_int), ("feas_jam", C.c_int), ("feas_stat", C.c_int),
                ("feas_back", C.c_double), ("feas_max", C.c_int), ("feas_delta_dec", C.c_double), ("feas_ret_push", C.c_double), ("feas_ret_mu", C.c_double), ("feas_resume", C.c_int), ("feas_polish", C.c_double), ("max_soc", C.c_int)]


def cpu_solve_batch(O, P, X0, threads=0, max_iter=None, tol=None, **opts):
    """oracle/landing_solver_cpu.c: scalar CPU port of the interior-point algorithm (cpu_baseline / cross-check)."""
    P = np.ascontiguousarray(np.atleast_2d(P), float); X0 = np.ascontiguousarray(np.atleast_2d(X0), float)
    B = P.shape[0]
    o = _SolverOpts()
    O.lib.lo_solver_opts_default(C.byref(o))
    if max_iter is not None:
        o.max_iter = max_iter
    if tol is not None:
        o.tol = tol
    for k_, v_ in opts.items():          # any other field of lo_solver_opts (clip_k, theta_floor, ...)
        setattr(o, k_, v_)
    x = np.zeros((B, O.nx)); lam = np.zeros((B, O.ng)); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    kkt = np.zeros((B, 3)); cnt = np.zeros(2, np.int64)
    ip = C.POINTER(C.c_int)
    O.lib.lo_solve_batch(O._F, C.c_int(B), _p(P), _p(X0), C.byref(o), C.c_int(threads), _p(x), _p(lam),
                         st.ctypes.data_as(ip), it.ctypes.data_as(ip), _p(kkt), cnt.ctypes.data_as(C.POINTER(_ll)))
    return dict(x=x, lam_g=lam, status=st, iters=it, kkt=kkt, factorizations=int(cnt[0]), trials=int(cnt[1]))


def cpu_sweep_batch(O, X, P, LAM, reps=1, threads=0):
    """oracle/landing_solver_cpu.c: `reps` full derivative sweeps of the batch on the host (bench.py's CPU function-layer leg)."""
    X = np.ascontiguousarray(X, float); P = np.ascontiguousarray(P, float); LAM = np.ascontiguousarray(LAM, float)
    return int(O.lib.lo_sweep_batch(O._F, C.c_int(X.shape[0]), _p(X), _p(P), _p(LAM), C.c_int(reps), C.c_int(threads)))


class RefOracle:
    """The reference's generated C through its CasADi external ABI (N=20: x[732], p[354])."""
    N, nx, np_, ng, nnz_jac, nnz_hess = 20, 732, 354, 2092, 7664, 3780

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "_ref", "liblanding_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        for fn in ("nlp", "nlp_f", "nlp_g", "nlp_grad", "nlp_grad_f", "nlp_hess_l", "nlp_jac_g"):
            f = getattr(self.lib, fn)
            f.restype = C.c_int
            f.argtypes = [C.POINTER(_dp), C.POINTER(_dp), C.POINTER(_ll), _dp, C.c_int]

    def _call(self, name, ins, outs):
        arg = (_dp * len(ins))(*[_p(a) for a in ins])
        res = (_dp * len(outs))(*[_p(a) for a in outs])
        rc = getattr(self.lib, name)(arg, res, None, None, 0)
        if rc != 0:
            raise RuntimeError(f"{name} returned {rc}")

    def sparsity(self, fn, which, idx):
        f = getattr(self.lib, f"{fn}_sparsity_{which}")
        f.restype = C.POINTER(_ll); f.argtypes = [_ll]
        ptr = f(idx)
        nrow, ncol = ptr[0], ptr[1]
        colind = np.array([ptr[2 + i] for i in range(ncol + 1)], np.int64)
        nnz = int(colind[-1])
        row = np.array([ptr[3 + ncol + i] for i in range(nnz)], np.int64)
        return int(nrow), int(ncol), colind, row

    def f(self, x, p):
        o = np.zeros(1); self._call("nlp_f", [x, p], [o]); return o[0]

    def grad_f(self, x, p):
        o = np.zeros(1); g = np.zeros(self.nx); self._call("nlp_grad_f", [x, p], [o, g]); return o[0], g

    def g(self, x, p):
        g = np.zeros(self.ng); self._call("nlp_g", [x, p], [g]); return g

    def jac_g(self, x, p):
        g = np.zeros(self.ng); j = np.zeros(self.nnz_jac); self._call("nlp_jac_g", [x, p], [g, j]); return g, j

    def hess_l(self, x, p, lam_f, lam_g):
        h = np.zeros(self.nnz_hess); lf = np.array([lam_f], float)
        self._call("nlp_hess_l", [x, p, lf, lam_g], [h]); return h

    def grad(self, x, p, lam_f, lam_g):
        f = np.zeros(1); g = np.zeros(self.ng); gx = np.zeros(self.nx); gp = np.zeros(self.np_)
        lf = np.array([lam_f], float)
        self._call("nlp_grad", [x, p, lf, lam_g], [f, g, gx, gp]); return f[0], g, gx, gp
