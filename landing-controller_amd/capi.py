"""ctypes binding of the C ABI (include/landing_nlp.h) of liblanding_mi355x.so.

The library is the product: HIP kernels for gfx950 behind plain-C entry points.  This module fails
loudly (ImportError / RuntimeError) when the library is missing or no GPU is usable -- there is no
CPU fallback.  ``load(path)`` lets the CPU test-suite bind the host-emulated build of the same
sources (tests/emu) for logic tests; product code never passes a path.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "liblanding_mi355x.so"
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_llp = C.POINTER(C.c_longlong)


class LandingForm(C.Structure):
    _fields_ = [("kin_box", C.c_double * 3), ("kin_z_off", C.c_double), ("comp_eps", C.c_double), ("slip_eps", C.c_double),
                ("run_cost", C.c_int), ("QX", C.c_double * 12), ("Qc", C.c_double * 3), ("Qf", C.c_double * 3),
                ("f_ref", C.c_double * 3), ("p_hip", C.c_double * 12)]


class SolverOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int), ("mu_init", C.c_double), ("bound_push", C.c_double),
                ("bound_frac", C.c_double), ("kappa_eps", C.c_double), ("kappa_mu", C.c_double), ("theta_mu", C.c_double),
                ("max_soc", C.c_int), ("max_resets", C.c_int), ("reset_du", C.c_double), ("stage_local_reg", C.c_int), ("sticky_delta", C.c_int), ("restart_period", C.c_int), ("dispatch_order", C.c_int),
                ("delta_init", C.c_double), ("delta_inc_first", C.c_double), ("delta_inc", C.c_double), ("delta_dec", C.c_double),
                ("tau_min", C.c_double), ("alpha_fallback", C.c_double), ("reset_delta", C.c_double), ("clip_k", C.c_int), ("clip_until", C.c_double), ("theta_floor", C.c_double), ("fresh_restart", C.c_int), ("dual_step_cap", C.c_double), ("slack_corr", C.c_double), ("watchdog", C.c_int), ("barrier_smax", C.c_double), ("factor_fp32", C.c_int),
                ("feas_phase", C.c_int), ("feas_rho", C.c_double), ("feas_cert", C.c_double), ("delta_floor", C.c_double), ("jam_clip", C.c_int), ("stag_relief", C.c_int), ("feas_jam", C.c_int), ("feas_stat", C.c_int),
                ("kd_clone_after", C.c_int), ("kd_clone_max", C.c_int), ("kd_clone_iter", C.c_int),
                ("feas_back", C.c_double), ("feas_max", C.c_int), ("feas_delta_dec", C.c_double), ("feas_ret_push", C.c_double), ("feas_ret_mu", C.c_double), ("feas_resume", C.c_int), ("feas_polish", C.c_double)]


ARGS21 = ["Xref", "Uref", "dt", "q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "q_term_min", "q_term_max",
          "qd_term_min", "qd_term_max", "QN", "x0", "mu", "l_leg_max", "f_max", "mass", "Ib", "Ib_inv"]


ARGS25 = ["Xref", "Uref", "dt", "q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "c_init", "q_term_min", "q_term_max",
          "qd_term_min", "qd_term_max", "QX", "QN", "Qc", "Qf", "x0", "mu", "l_leg_max", "f_max", "mass", "Ib", "Ib_inv"]


class Args25(C.Structure):
    """landing_args25: the 25 solver-function arguments of the reference's N=41 script (analysis/eval_SRBM_CCC.m:72-78), host pointers"""
    _fields_ = [(n, _dp) for n in ARGS25]


class Args21(C.Structure):
    """landing_args21: the reference's 21 solver-function arguments (generate_landingCtrller_IPOPT.m:323-327), host pointers"""
    _fields_ = [(n, _dp) for n in ARGS21]


EXPORTS = ["landing_last_error", "landing_form_default", "landing_solver_opts_default", "landing_nx", "landing_ng",
           "landing_np", "landing_nnz_jac", "landing_nnz_hess", "landing_pattern_jac", "landing_pattern_hess",
           "landing_create", "landing_destroy", "landing_device_count", "landing_eval_batch", "landing_eval_batch_host",
           "landing_bounds_batch", "landing_solve_batch", "landing_solve_batch_host", "landing_kernel_name_sweep",
           "landing_sweep_bytes_per_member", "landing_set_profile_buffer", "landing_debug_workspace",
           "landing_pack_args21", "landing_solve_args21", "landing_solve_21",
           "landing_multi_create", "landing_multi_destroy", "landing_multi_count", "landing_shard_range", "landing_multi_solve_args21",
           "landing_solve_21_multi", "landing_multi_release_cached",
           "landing_np_ccc", "landing_ctx_np", "landing_pack_args25", "landing_solve_args25", "landing_riccati_gains_batch", "landing_mpc_shift", "landing_solver_opts_warm", "landing_rbd_set_model", "landing_fb_dynamics_batch",
           "landing_kinodyn_rows_batch", "landing_kinodyn_nlp_dims", "landing_kinodyn_nlp_eval", "landing_kinodyn_nlp_hess", "landing_leg_ik_batch", "landing_nnz_hess_rc", "landing_pattern_hess_rc",
           "landing_eval_hess_rc_batch", "landing_eval_hess_rc_batch_host",
           "landing_stream_create", "landing_stream_destroy", "landing_stream_lanes", "landing_stream_submit", "landing_stream_wait", "landing_stream_sync", "landing_solve_stream_host"]


def load(path=None):
    path = path or os.path.join(HERE, LIB_NAME)
    try:  # torch ships its own libamdhip64: it must be the first HIP runtime loaded in the process,
        import torch  # noqa: F401  (otherwise torch.cuda later reports "No HIP GPUs are available")
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with __graft_entry__.build() (hipcc, gfx950). No CPU fallback exists.")
    lib = C.CDLL(path)
    lib.landing_last_error.restype = C.c_char_p
    lib.landing_kernel_name_sweep.restype = C.c_char_p
    for n in ("landing_nx", "landing_ng", "landing_np", "landing_np_ccc", "landing_nnz_jac", "landing_nnz_hess", "landing_sweep_bytes_per_member"):
        getattr(lib, n).restype = C.c_longlong
        getattr(lib, n).argtypes = [C.c_int]
    lib.landing_ctx_np.restype = C.c_longlong
    lib.landing_ctx_np.argtypes = [C.c_void_p]
    lib.landing_pattern_jac.argtypes = [C.c_int, _llp, _llp]
    lib.landing_pattern_hess.argtypes = [C.c_int, _llp, _llp]
    lib.landing_create.restype = C.c_void_p
    lib.landing_create.argtypes = [C.c_int, C.c_int, C.POINTER(LandingForm)]
    lib.landing_destroy.argtypes = [C.c_void_p]
    vp = C.c_void_p
    lib.landing_eval_batch.argtypes = [vp, C.c_int] + [vp] * 11 + [vp]
    lib.landing_eval_batch_host.argtypes = [vp, C.c_int] + [_dp] * 11
    lib.landing_bounds_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    if hasattr(lib, "landing_eval_hess_rc_batch"):
        lib.landing_nnz_hess_rc.restype = C.c_longlong; lib.landing_nnz_hess_rc.argtypes = [C.c_int]
        lib.landing_pattern_hess_rc.argtypes = [C.c_int, _llp, _llp]
        lib.landing_eval_hess_rc_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
        lib.landing_eval_hess_rc_batch_host.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, _dp]
    lib.landing_solve_batch.argtypes = [vp, C.c_int, vp, vp, C.POINTER(SolverOpts), vp, vp, vp, vp, vp, vp, vp]
    lib.landing_set_profile_buffer.argtypes = [vp, vp]
    lib.landing_solve_batch_host.argtypes = [vp, C.c_int, _dp, _dp, C.POINTER(SolverOpts), _dp, _dp, _dp, _ip, _ip, _dp]
    lib.landing_pack_args21.argtypes = [C.c_int, C.c_int, C.POINTER(Args21), _dp]
    lib.landing_solve_args21.argtypes = [vp, C.c_int, C.POINTER(Args21), C.POINTER(SolverOpts), _dp, _dp, _ip, _ip, _dp]
    if hasattr(lib, "landing_riccati_gains_batch"):      # (older development builds used by tools/dev/variants.py lack it)
        lib.landing_riccati_gains_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, _dp, C.c_double, _dp, _dp, _dp, C.c_double, C.c_int, vp, vp, vp, vp, vp]
    if hasattr(lib, "landing_mpc_shift"):
        lib.landing_mpc_shift.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    if hasattr(lib, "landing_solve_args25"):
        lib.landing_pack_args25.argtypes = [C.c_int, C.c_int, C.POINTER(Args25), _dp]
        lib.landing_solve_args25.argtypes = [vp, C.c_int, C.POINTER(Args25), C.POINTER(SolverOpts), _dp, _dp, _dp, _ip, _ip, _dp]
    if hasattr(lib, "landing_multi_create"):
        lib.landing_multi_create.restype = C.c_void_p
        lib.landing_multi_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(LandingForm)]
        lib.landing_multi_destroy.argtypes = [vp]
        lib.landing_multi_solve_args21.argtypes = [vp, C.c_int, C.POINTER(Args21), C.POINTER(SolverOpts), _dp, _dp, _dp, _ip, _ip, _dp]
        lib.landing_solve_21_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int] + [_dp] * 21 + [C.POINTER(SolverOpts), _dp, _dp, _dp, _ip, _ip, _dp]
    lib.landing_solve_21.argtypes = [vp, C.c_int] + [_dp] * 21 + [C.POINTER(SolverOpts), _dp, _dp, _ip, _ip, _dp]
    if hasattr(lib, "landing_stream_create"):      # streaming entry points (round 6)
        lib.landing_stream_create.restype = C.c_void_p
        lib.landing_stream_create.argtypes = [vp, C.c_int]
        lib.landing_stream_destroy.argtypes = [vp]
        lib.landing_stream_lanes.argtypes = [vp]
        lib.landing_stream_submit.restype = C.c_longlong
        lib.landing_stream_submit.argtypes = [vp, C.c_int, vp, vp, C.POINTER(SolverOpts), vp, vp, vp, vp, vp, vp, vp]
        lib.landing_stream_wait.argtypes = [vp, C.c_longlong, vp]
        lib.landing_stream_sync.argtypes = [vp, vp]
        lib.landing_solve_stream_host.argtypes = [vp, C.c_int, C.c_int, C.c_int, _dp, _dp, C.POINTER(SolverOpts), _dp, _dp, _dp, _ip, _ip, _dp]
    return lib


class SolveStream:
    """landing_stream_* (include/landing_nlp.h): consecutive batches through one context with `lanes` launches in flight.  Device pointers; the caller keeps
    the buffers of a submission alive until it has waited for its ticket."""

    def __init__(self, lib, lanes=2):
        self.L = lib
        self.h = lib.lib.landing_stream_create(lib.ctx, lanes)
        if not self.h:
            raise RuntimeError("landing_stream_create: " + lib.lib.landing_last_error().decode())
        self.lanes = lib.lib.landing_stream_lanes(self.h)

    def submit(self, B, d_p, d_x0, opts, d_x, d_f=0, d_lam_g=0, d_status=0, d_iters=0, d_kkt=0, in_stream=0):
        t = self.L.lib.landing_stream_submit(self.h, B, d_p, d_x0, C.byref(opts), d_x, d_f or None, d_lam_g or None, d_status or None, d_iters or None, d_kkt or None, in_stream or None)
        if t < 0:
            raise RuntimeError("landing_stream_submit: " + self.L.lib.landing_last_error().decode())
        return t

    def wait(self, ticket, stream=0):
        self.L._check(self.L.lib.landing_stream_wait(self.h, ticket, stream or None), "landing_stream_wait")

    def sync(self, stream=0):
        self.L._check(self.L.lib.landing_stream_sync(self.h, stream or None), "landing_stream_sync")

    def close(self):
        if self.h:
            self.L.lib.landing_stream_destroy(self.h); self.h = None


def matlab_args25(N, args):
    """the same for the 25 arguments of the N=41 script (c_init may be missing: inactive)"""
    keep, a = [], Args25()
    B = np.asarray(args["x0"]).shape[-1] if np.asarray(args["x0"]).ndim > 1 else 1
    for n in ARGS25:
        v = args.get(n)
        if v is None:
            setattr(a, n, None)
            continue
        buf = np.ascontiguousarray(np.asarray(v, float).reshape((-1, B), order="F").T)
        keep.append(buf)
        setattr(a, n, buf.ctypes.data_as(_dp))
    return a, keep, B


def matlab_args21(N, args):
    """dict of the 21 arguments as MATLAB would hold them for a batch of B members -- arrays whose LAST axis is the batch
    (Xref [12, N+1, B], dt [1, N, B], 6-vectors [6, B], x0 [nx, B], scalars [1, B] ...) -- flattened column-major into the
    member-major host buffers the C ABI takes.  Returns (Args21, keep-alive list, B)."""
    keep, a = [], Args21()
    B = np.asarray(args["x0"]).shape[-1] if np.asarray(args["x0"]).ndim > 1 else 1
    for n in ARGS21:
        v = args.get(n)
        if v is None:
            setattr(a, n, None)
            continue
        buf = np.ascontiguousarray(np.asarray(v, float).reshape((-1, B), order="F").T)     # [B, n] member-major
        keep.append(buf)
        setattr(a, n, buf.ctypes.data_as(_dp))
    return a, keep, B


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


class LandingLib:
    """Thin object wrapper: one context per (N, device)."""

    def __init__(self, N, device=0, kin_box=None, lib_path=None, run_cost=None, ccc_params=False):
        self.lib = load(lib_path)
        self.N = N
        form = LandingForm()
        self.lib.landing_form_default(C.byref(form))
        if kin_box is not None:
            for i in range(3):
                form.kin_box[i] = kin_box[i]
        if run_cost is not None:      # dict(QX=[12], Qc=[3], Qf=[3], f_ref=[3]): generate_quadruped_SRBM_CCC.m:81-89
            form.run_cost = 1
            for i in range(12):
                form.QX[i] = run_cost["QX"][i]
            for i in range(3):
                form.Qc[i] = run_cost["Qc"][i]; form.Qf[i] = run_cost["Qf"][i]; form.f_ref[i] = run_cost.get("f_ref", (0, 0, 0))[i]
        if ccc_params:                # the N=41 script's own parameter vector: weights and force reference are entries of p (run_cost = 2)
            form.run_cost = 2
        self.form = form
        self.ctx = self.lib.landing_create(N, device, C.byref(form))
        if not self.ctx:
            raise RuntimeError("landing_create failed: " + self.lib.landing_last_error().decode())
        self.nx, self.ng, self.np_ = self.lib.landing_nx(N), self.lib.landing_ng(N), self.lib.landing_ctx_np(self.ctx)
        self.nnz_jac, self.nnz_hess = self.lib.landing_nnz_jac(N), self.lib.landing_nnz_hess(N)

    def close(self):
        if self.ctx:
            self.lib.landing_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.landing_last_error().decode()}")

    def default_opts(self):
        o = SolverOpts()
        self.lib.landing_solver_opts_default(C.byref(o))
        return o

    def warm_opts(self):
        o = SolverOpts()
        self.lib.landing_solver_opts_warm(C.byref(o))
        return o

    def mpc_shift_device(self, B, d_x_prev, d_state, d_p, d_x0, stream=0):
        self._check(self.lib.landing_mpc_shift(self.ctx, B, d_x_prev, d_state, d_p, d_x0, stream or None), "landing_mpc_shift")

    def pattern_jac(self):
        ci = np.zeros(self.nx + 1, np.int64); r = np.zeros(self.nnz_jac, np.int64)
        self._check(self.lib.landing_pattern_jac(self.N, ci.ctypes.data_as(_llp), r.ctypes.data_as(_llp)), "pattern_jac")
        return ci, r

    def pattern_hess(self):
        ci = np.zeros(self.nx + 1, np.int64); r = np.zeros(self.nnz_hess, np.int64)
        self._check(self.lib.landing_pattern_hess(self.N, ci.ctypes.data_as(_llp), r.ctypes.data_as(_llp)), "pattern_hess")
        return ci, r

    def pattern_hess_rc(self):
        """pattern of the Lagrangian Hessian of the running-cost formulation: casadi_s4 + 18 N diagonals"""
        n = self.lib.landing_nnz_hess_rc(self.N)
        ci = np.zeros(self.nx + 1, np.int64); r = np.zeros(n, np.int64)
        self._check(self.lib.landing_pattern_hess_rc(self.N, ci.ctypes.data_as(_llp), r.ctypes.data_as(_llp)), "pattern_hess_rc")
        return ci, r

    def hess_rc_host(self, x, p, lam_f, lam_g):
        x = np.ascontiguousarray(np.atleast_2d(x), float); p = np.ascontiguousarray(np.atleast_2d(p), float)
        lam_g = np.ascontiguousarray(np.atleast_2d(lam_g), float)
        lam_f = None if lam_f is None else np.ascontiguousarray(np.atleast_1d(lam_f), float)
        h = np.full((x.shape[0], self.lib.landing_nnz_hess_rc(self.N)), np.nan)
        self._check(self.lib.landing_eval_hess_rc_batch_host(self.ctx, x.shape[0], _p(x), _p(p), _p(lam_f), _p(lam_g), _p(h)), "landing_eval_hess_rc_batch_host")
        return h

    # ---- host-pointer entry points (numpy in / numpy out) --------------------------------------
    def eval_host(self, x, p, lam_f=None, lam_g=None, want=("f", "g", "grad_f", "jac", "hess", "grad_gamma_x", "grad_gamma_p")):
        x = np.ascontiguousarray(np.atleast_2d(x), float); p = np.ascontiguousarray(np.atleast_2d(p), float)
        B = x.shape[0]
        if lam_g is not None:
            lam_g = np.ascontiguousarray(np.atleast_2d(lam_g), float)
        if lam_f is not None:
            lam_f = np.ascontiguousarray(np.atleast_1d(lam_f), float)
        shapes = dict(f=(B,), g=(B, self.ng), grad_f=(B, self.nx), jac=(B, self.nnz_jac), hess=(B, self.nnz_hess),
                      grad_gamma_x=(B, self.nx), grad_gamma_p=(B, self.np_))
        out = {k: (np.full(shapes[k], np.nan) if k in want else None) for k in shapes}
        rc = self.lib.landing_eval_batch_host(self.ctx, B, _p(x), _p(p), _p(lam_f), _p(lam_g), _p(out["f"]), _p(out["g"]),
                                              _p(out["grad_f"]), _p(out["jac"]), _p(out["hess"]), _p(out["grad_gamma_x"]),
                                              _p(out["grad_gamma_p"]))
        self._check(rc, "landing_eval_batch_host")
        return {k: v for k, v in out.items() if v is not None}

    def solve_host(self, p, x0, opts=None):
        p = np.ascontiguousarray(np.atleast_2d(p), float); x0 = np.ascontiguousarray(np.atleast_2d(x0), float)
        B = p.shape[0]
        opts = opts or self.default_opts()
        x = np.zeros((B, self.nx)); f = np.zeros(B); lam = np.zeros((B, self.ng))
        status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
        rc = self.lib.landing_solve_batch_host(self.ctx, B, _p(p), _p(x0), C.byref(opts), _p(x), _p(f), _p(lam),
                                               status.ctypes.data_as(_ip), iters.ctypes.data_as(_ip), _p(kkt))
        self._check(rc, "landing_solve_batch_host")
        return dict(x=x, f=f, lam_g=lam, status=status, iters=iters, kkt=kkt)

    def solve_stream_host(self, p, x0, opts=None, chunk=1024, lanes=2):
        """landing_solve_stream_host: any number of members, cut into chunks that go through a stream of `lanes` launches in flight"""
        p = np.ascontiguousarray(np.atleast_2d(p), float); x0 = np.ascontiguousarray(np.atleast_2d(x0), float)
        B = p.shape[0]
        opts = opts or self.default_opts()
        x = np.zeros((B, self.nx)); f = np.zeros(B); lam = np.zeros((B, self.ng))
        status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
        rc = self.lib.landing_solve_stream_host(self.ctx, B, chunk, lanes, _p(p), _p(x0), C.byref(opts), _p(x), _p(f), _p(lam),
                                                status.ctypes.data_as(_ip), iters.ctypes.data_as(_ip), _p(kkt))
        self._check(rc, "landing_solve_stream_host")
        return dict(x=x, f=f, lam_g=lam, status=status, iters=iters, kkt=kkt)

    def stream(self, lanes=2):
        """landing_stream_create on this context: SolveStream with submit / wait / sync / close"""
        return SolveStream(self, lanes)

    def pack_args21(self, args):
        """p [B, np] from the 21 MATLAB-shaped arguments (landing_pack_args21; host only)"""
        a, keep, B = matlab_args21(self.N, args)
        p = np.zeros((B, self.np_))
        self._check(self.lib.landing_pack_args21(self.N, B, C.byref(a), _p(p)), "landing_pack_args21")
        return p

    def solve_args21(self, args, opts=None, spelled_out=False):
        """the reference's solver-function call, batched: args = dict of the 21 MATLAB-shaped arrays (batch = last axis)"""
        a, keep, B = matlab_args21(self.N, args)
        opts = opts or self.default_opts()
        x = np.zeros((B, self.nx)); f = np.zeros(B); status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
        outs = (_p(x), _p(f), status.ctypes.data_as(_ip), iters.ctypes.data_as(_ip), _p(kkt))
        if spelled_out:
            rc = self.lib.landing_solve_21(self.ctx, B, *[getattr(a, n) for n in ARGS21], C.byref(opts), *outs)
        else:
            rc = self.lib.landing_solve_args21(self.ctx, B, C.byref(a), C.byref(opts), *outs)
        self._check(rc, "landing_solve_args21")
        return dict(x=x, f=f, status=status, iters=iters, kkt=kkt)

    def solve_args25(self, args, opts=None):
        """the N=41 script's solver-function call, batched: args = dict of its 25 MATLAB-shaped arrays (batch = last axis)"""
        a, keep, B = matlab_args25(self.N, args)
        opts = opts or self.default_opts()
        x = np.zeros((B, self.nx)); f = np.zeros(B); lam = np.zeros((B, self.ng)); status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
        rc = self.lib.landing_solve_args25(self.ctx, B, C.byref(a), C.byref(opts), _p(x), _p(f), _p(lam), status.ctypes.data_as(_ip), iters.ctypes.data_as(_ip), _p(kkt))
        self._check(rc, "landing_solve_args25")
        return dict(x=x, f=f, lam_g=lam, status=status, iters=iters, kkt=kkt)

    def solve_args21_multi(self, args, devices, opts=None, one_call=False, want_lam=True):
        """the reference's solver-function call sharded over a device list from the C boundary (landing_multi_solve_args21 /
        landing_solve_21_multi): contiguous shards, one host thread + context per entry of `devices` (an index may repeat)"""
        a, keep, B = matlab_args21(self.N, args)
        opts = opts or self.default_opts()
        x = np.zeros((B, self.nx)); f = np.zeros(B); status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
        lam = np.zeros((B, self.ng)) if want_lam else None
        outs = (_p(x), _p(f), _p(lam), status.ctypes.data_as(_ip), iters.ctypes.data_as(_ip), _p(kkt))
        dev = (C.c_int * len(devices))(*devices)
        if one_call:
            self.lib.landing_solve_21_multi.restype = C.c_int
            rc = self.lib.landing_solve_21_multi(dev, len(devices), self.N, B, *[getattr(a, n) for n in ARGS21], C.byref(opts), *outs)
        else:
            self.lib.landing_multi_create.restype = C.c_void_p
            m = self.lib.landing_multi_create(self.N, dev, len(devices), C.byref(self.form))
            if not m:
                raise RuntimeError("landing_multi_create: " + self.lib.landing_last_error().decode())
            try:
                rc = self.lib.landing_multi_solve_args21(C.c_void_p(m), B, C.byref(a), C.byref(opts), *outs)
            finally:
                self.lib.landing_multi_destroy(C.c_void_p(m))
        self._check(rc, "landing_multi_solve_args21")
        return dict(x=x, f=f, lam_g=lam, status=status, iters=iters, kkt=kkt)

    def riccati_gains_device(self, B, n, d_xref, d_fref, Ib3x3, mass, Q, r_diag, F, dt, rk4=False, d_P=0, d_K=0, d_A=0, d_B=0, stream=0):
        """landing_riccati_gains_batch: VBL linearisation + Riccati tracking gains along B sampled trajectories (device pointers)"""
        Ib3x3 = np.ascontiguousarray(Ib3x3, float); Q = np.ascontiguousarray(Q, float); F = np.ascontiguousarray(F, float); r = np.ascontiguousarray(r_diag, float)
        rc = self.lib.landing_riccati_gains_batch(self.ctx, B, n, d_xref, d_fref, _p(Ib3x3), float(mass), _p(Q), _p(r), _p(F), float(dt), int(bool(rk4)),
                                                  d_P or None, d_K or None, d_A or None, d_B or None, stream or None)
        self._check(rc, "landing_riccati_gains_batch")

    # ---- device-pointer entry points (integers = device addresses, e.g. torch tensor.data_ptr()) --
    def eval_device(self, B, d_x, d_p, d_lam_f=0, d_lam_g=0, d_f=0, d_g=0, d_grad_f=0, d_jac=0, d_hess=0, d_ggx=0, d_ggp=0, stream=0):
        rc = self.lib.landing_eval_batch(self.ctx, B, d_x, d_p, d_lam_f or None, d_lam_g or None, d_f or None, d_g or None,
                                         d_grad_f or None, d_jac or None, d_hess or None, d_ggx or None, d_ggp or None, stream or None)
        self._check(rc, "landing_eval_batch")

    def bounds_device(self, B, d_p, d_lbg, d_ubg, stream=0):
        self._check(self.lib.landing_bounds_batch(self.ctx, B, d_p, d_lbg, d_ubg, stream or None), "landing_bounds_batch")

    def solve_device(self, B, d_p, d_x0, opts, d_x, d_f=0, d_lam_g=0, d_status=0, d_iters=0, d_kkt=0, stream=0):
        rc = self.lib.landing_solve_batch(self.ctx, B, d_p, d_x0, C.byref(opts), d_x, d_f or None, d_lam_g or None,
                                          d_status or None, d_iters or None, d_kkt or None, stream or None)
        self._check(rc, "landing_solve_batch")
