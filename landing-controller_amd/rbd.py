"""Host side of the floating-base rigid-body routines: the reference's 'quad3D' Mini-Cheetah tree
(utilities_general/dynamics-utilities/get_robot_model.m:134-234 with the 'mc3D' parameters of get_robot_params.m:50-115)
in the compact form the HIP kernels use -- Xtree = plux(E, r), link inertia = (mass, m*com, rotational inertia about the
link origin) -- and the ctypes binding of landing_rbd_set_model / landing_fb_dynamics_batch / landing_kinodyn_rows_batch."""
import ctypes as C

import numpy as np

from . import constants as K


class RbdModel(C.Structure):
    _fields_ = [("parent", C.c_int * 18), ("jtype", C.c_int * 18), ("E", (C.c_double * 9) * 18), ("r", (C.c_double * 3) * 18),
                ("m", C.c_double * 18), ("h", (C.c_double * 3) * 18), ("I", (C.c_double * 6) * 18),
                ("b_foot", C.c_int * 4), ("foot_r", (C.c_double * 3) * 4), ("l1", C.c_double), ("l2", C.c_double), ("l3", C.c_double), ("l4", C.c_double)]


def _rbi(m, com, rot):
    """(m, h, Ibar about the origin): spatialInertia.m restated in compact form"""
    com = np.asarray(com, float); c = K._skew(com)
    Ibar = np.asarray(rot, float) + m * (c @ c.T)
    return m, m * com, np.array([Ibar[0, 0], Ibar[0, 1], Ibar[0, 2], Ibar[1, 1], Ibar[1, 2], Ibar[2, 2]])


def _flip_y_rbi(m, h, I6):
    """flipAlongAxis(I, 'Y') (get_robot_model.m:852-889): mirror the link through the x-z plane"""
    return m, h * np.array([1, -1, 1.0]), I6 * np.array([1, -1, 1, 1, -1, 1.0])


def quad3d_model():
    M = RbdModel()
    abad = _rbi(0.54, [0, 0.036, 0], 1e-6 * np.array([[381, 58, 0.45], [58, 560, 0.95], [0.45, 0.95, 444]]))
    hip = _rbi(0.634, [0, 0.016, -0.02], 1e-6 * np.array([[1983, 245, 13], [245, 2103, 1.5], [13, 1.5, 408]]))
    knee = _rbi(0.064, [0, 0, -0.061], 1e-6 * np.array([[6, 0, 0], [0, 248, 0], [0, 0, 245.0]]))
    body = _rbi(3.3, [0, 0, 0], 1e-6 * np.diag([11253.0, 36203.0, 42673.0]))
    zero = (0.0, np.zeros(3), np.zeros(6))
    abad_loc, hip_loc, knee_loc, foot_loc = np.array([0.19, 0.049, 0.0]), np.array([0, 0.062, 0.0]), np.array([0, 0, -0.209]), np.array([0, 0, -0.195])
    side = np.array([[1, 1, -1, -1], [-1, 1, -1, 1], [1, 1, 1, 1]], float)
    rzpi = np.array([[np.cos(np.pi), np.sin(np.pi), 0], [-np.sin(np.pi), np.cos(np.pi), 0], [0, 0, 1.0]])
    bodies = [(i, t, np.eye(3), np.zeros(3), zero) for i, t in zip(range(6), (3, 4, 5, 0, 1, 2))]     # Px Py Pz Rx Ry Rz
    bodies[5] = (5, 2, np.eye(3), np.zeros(3), body)
    leg_side = -1
    for leg in range(4):
        s = side[:, leg]
        links = [abad, hip, knee] if leg_side > 0 else [_flip_y_rbi(*abad), _flip_y_rbi(*hip), _flip_y_rbi(*knee)]
        n0 = len(bodies)
        bodies.append((6, 0, np.eye(3), s * abad_loc, links[0]))
        bodies.append((n0 + 1, 1, rzpi, s * hip_loc, links[1]))           # plux(rz(pi), 0) * plux(1, r) = plux(rz(pi), r)
        bodies.append((n0 + 2, 1, np.eye(3), s * knee_loc, links[2]))
        M.b_foot[leg] = n0 + 3
        for j in range(3):
            M.foot_r[leg][j] = (s * foot_loc)[j]
        leg_side = -leg_side
    for i, (pa, jt, E, r, (m, h, I6)) in enumerate(bodies):
        M.parent[i] = pa; M.jtype[i] = jt; M.m[i] = m
        for j in range(9): M.E[i][j] = E.flatten()[j]
        for j in range(3): M.r[i][j] = r[j]; M.h[i][j] = h[j]
        for j in range(6): M.I[i][j] = I6[j]
    M.l1, M.l2, M.l3, M.l4 = 0.062, 0.209, 0.195, 0.004                       # get_foot_jacobians_mc.m:5-8
    return M


TAU_MAX = np.tile(np.array([6.0, 6.0, 9.33]) * 3.0, 4)     # model.gr .* motorTauMax (get_robot_model.m:236-240, get_robot_params.m:103-108)
JPOS_MIN = np.tile([-np.pi / 3, -np.pi / 2, 0.0], 4)       # landing_optimization.m:246-247
JPOS_MAX = np.tile([np.pi / 3, np.pi / 2, 3 * np.pi / 4], 4)


class KinodynParams(C.Structure):
    """landing_kinodyn_params of include/landing_nlp.h"""
    _fields_ = [("dt", C.c_double * 64), ("mass", C.c_double), ("Ib", C.c_double * 3), ("Ib_inv", C.c_double * 3), ("mu", C.c_double)]


class Rbd:
    """binds the three entry points on an existing LandingLib (its context owns the uploaded model)"""

    def __init__(self, lib):
        self.L = lib
        vp = C.c_void_p
        lib.lib.landing_rbd_set_model.argtypes = [vp, C.POINTER(RbdModel)]
        lib.lib.landing_fb_dynamics_batch.argtypes = [vp, C.c_int] + [vp] * 9 + [C.c_double, vp]
        lib.lib.landing_kinodyn_rows_batch.argtypes = [vp, C.c_int] + [vp] * 7 + [vp]
        dp = C.POINTER(C.c_double)
        lib.lib.landing_leg_ik_batch.argtypes = [vp, C.c_int, vp, vp, dp, dp, C.c_int, vp, vp, vp]
        lib.lib.landing_kinodyn_nlp_dims.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        lib.lib.landing_kinodyn_nlp_eval.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(KinodynParams), vp, vp, vp]
        lib.lib.landing_kinodyn_nlp_hess.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(KinodynParams), vp, vp, vp]
        ip = C.POINTER(C.c_int)
        lib.lib.landing_kinodyn_solve_batch_host.argtypes = [vp, C.c_int, C.c_int, C.POINTER(KinodynParams), dp, dp, dp, dp, vp, dp, dp, dp, ip, ip, dp]
        lib.lib.landing_kinodyn_solve_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(KinodynParams)] + [vp] * 4 + [vp] + [vp] * 6 + [vp]
        self.model = quad3d_model()
        lib._check(lib.lib.landing_rbd_set_model(lib.ctx, C.byref(self.model)), "landing_rbd_set_model")

    def fb_dynamics(self, npts, d_q, d_qd, d_tau=0, d_f_foot=0, d_H=0, d_C=0, d_qdd=0, d_A=0, d_Hinv=0, fd_h=0.0, stream=0):
        """fd_h = 0: exact linearisation (forward-mode tangents of the inverse dynamics); fd_h > 0: central differences with that step"""
        n = lambda v: v or None
        self.L._check(self.L.lib.landing_fb_dynamics_batch(self.L.ctx, npts, d_q, d_qd, n(d_tau), n(d_f_foot), n(d_H), n(d_C), n(d_qdd), n(d_A), n(d_Hinv), fd_h, n(stream)),
                      "landing_fb_dynamics_batch")

    def kinodyn_rows(self, npts, d_q6, d_c, d_f, d_jpos, d_fk=0, d_fk_err=0, d_tau=0, stream=0):
        n = lambda v: v or None
        self.L._check(self.L.lib.landing_kinodyn_rows_batch(self.L.ctx, npts, d_q6, n(d_c), n(d_f), d_jpos, n(d_fk), n(d_fk_err), n(d_tau), n(stream)), "landing_kinodyn_rows_batch")

    def kinodyn_nlp_dims(self, N):
        nx, ng = C.c_longlong(), C.c_longlong()
        self.L._check(self.L.lib.landing_kinodyn_nlp_dims(N, C.byref(nx), C.byref(ng)), "landing_kinodyn_nlp_dims")
        return nx.value, ng.value

    @staticmethod
    def _kd_params(N, dt, mass, Ib, Ib_inv, mu):
        prm = KinodynParams()
        for k in range(N):
            prm.dt[k] = float(dt[k])
        prm.mass = float(mass); prm.mu = float(mu)
        for i in range(3):
            prm.Ib[i] = float(Ib[i]); prm.Ib_inv[i] = float(Ib_inv[i])
        return prm

    def kinodyn_nlp_eval(self, B, N, d_x, dt, mass, Ib, Ib_inv, mu, d_g=0, d_jac=0, stream=0):
        """function layer of the kinodynamic refinement NLP (include/landing_nlp.h): g [B, ng] and / or the Jacobian blocks [B, N, 141, 72]"""
        prm = self._kd_params(N, dt, mass, Ib, Ib_inv, mu)
        n = lambda v: v or None
        self.L._check(self.L.lib.landing_kinodyn_nlp_eval(self.L.ctx, B, N, d_x, C.byref(prm), n(d_g), n(d_jac), n(stream)), "landing_kinodyn_nlp_eval")

    def kinodyn_nlp_hess(self, B, N, d_x, dt, mass, Ib, Ib_inv, mu, d_lam_g, d_hess, stream=0):
        """Hessian blocks of lam_g' g per interval [B, N, 72, 72] (include/landing_nlp.h)"""
        prm = self._kd_params(N, dt, mass, Ib, Ib_inv, mu)
        self.L._check(self.L.lib.landing_kinodyn_nlp_hess(self.L.ctx, B, N, d_x, C.byref(prm), d_lam_g, d_hess, stream or None), "landing_kinodyn_nlp_hess")

    def kinodyn_default_opts(self):
        from .capi import SolverOpts
        o = SolverOpts()
        self.L.lib.landing_kinodyn_solver_opts_default(C.byref(o))
        return o

    def kinodyn_warm_opts(self):
        """landing_kinodyn_solver_opts_warm: the re-solve from a previous solution"""
        from .capi import SolverOpts
        o = SolverOpts()
        self.L.lib.landing_kinodyn_solver_opts_warm(C.byref(o))
        return o

    def kinodyn_solve_host(self, N, lbg, ubg, cost, x0, dt, mass, Ib, Ib_inv, mu, opts=None):
        """landing_kinodyn_solve_batch_host: B kinodynamic refinement NLPs (host arrays [B, ng], [B, ng], [B, 24], [B, nx]) -> dict"""
        lbg = np.ascontiguousarray(np.atleast_2d(lbg), float); ubg = np.ascontiguousarray(np.atleast_2d(ubg), float)
        cost = np.ascontiguousarray(np.atleast_2d(cost), float); x0 = np.ascontiguousarray(np.atleast_2d(x0), float)
        B = x0.shape[0]; nx, ng = self.kinodyn_nlp_dims(N)
        assert x0.shape == (B, nx) and lbg.shape == (B, ng) and ubg.shape == (B, ng) and cost.shape == (B, 24)
        prm = self._kd_params(N, dt, mass, Ib, Ib_inv, mu)
        x = np.zeros((B, nx)); f = np.zeros(B); lam = np.zeros((B, ng)); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int); P = lambda a: a.ctypes.data_as(dp)
        rc = self.L.lib.landing_kinodyn_solve_batch_host(self.L.ctx, B, N, C.byref(prm), P(lbg), P(ubg), P(cost), P(x0), C.byref(opts) if opts is not None else None,
                                                         P(x), P(f), P(lam), st.ctypes.data_as(ip), it.ctypes.data_as(ip), P(kkt))
        self.L._check(rc, "landing_kinodyn_solve_batch_host")
        return dict(x=x, f=f, lam_g=lam, status=st, iters=it, kkt=kkt)

    def kinodyn_solve_device(self, B, N, d_lbg, d_ubg, d_cost, d_x0, dt, mass, Ib, Ib_inv, mu, opts, d_x, d_f=0, d_lam=0, d_status=0, d_iters=0, d_kkt=0, stream=0):
        prm = self._kd_params(N, dt, mass, Ib, Ib_inv, mu)
        n = lambda v: v or None
        rc = self.L.lib.landing_kinodyn_solve_batch(self.L.ctx, B, N, C.byref(prm), d_lbg, d_ubg, d_cost, d_x0, C.byref(opts) if opts is not None else None,
                                                    d_x, n(d_f), n(d_lam), n(d_status), n(d_iters), n(d_kkt), n(stream))
        self.L._check(rc, "landing_kinodyn_solve_batch")

    ARGS24 = ("Xref", "Uref", "dt", "q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "c_init", "q_term_min", "q_term_max", "qd_term_min",
              "qd_term_max", "QN", "x0", "jpos_min", "jpos_max", "kin_box", "mu", "l_leg_max", "mass", "Ib", "Ib_inv")

    def kinodyn_solve_24(self, N, args, opts=None, form=None):
        """landing_solve_kinodyn_24: the reference's 24-argument solver function (generate_landingCtrller_KNITRO.m:373-377), `args` = dict name ->
        MATLAB-shaped array (column-major, batch = last axis; Uref may be None)"""
        dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
        keep = []
        def conv(name):
            v = args.get(name)
            if v is None:
                return None
            a = np.asfortranarray(np.asarray(v, float)); keep.append(a)
            return a.ctypes.data_as(dp)
        ptrs = [conv(n) for n in self.ARGS24]
        B = int(np.asarray(args["x0"]).reshape(48 * N + 12, -1, order="F").shape[1])
        nx, ng = self.kinodyn_nlp_dims(N)
        x = np.zeros((B, nx)); f = np.zeros(B); lam = np.zeros((B, ng)); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
        P = lambda a: a.ctypes.data_as(dp)
        fn = self.L.lib.landing_solve_kinodyn_24
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p] + [dp] * 24 + [C.c_void_p, dp, dp, dp, ip, ip, dp]
        rc = fn(self.L.ctx, N, B, C.byref(form) if form is not None else None, *ptrs, C.byref(opts) if opts is not None else None,
                P(x), P(f), P(lam), st.ctypes.data_as(ip), it.ctypes.data_as(ip), P(kkt))
        self.L._check(rc, "landing_solve_kinodyn_24")
        return dict(x=x, f=f, lam_g=lam, status=st, iters=it, kkt=kkt)

    def kinodyn_bounds(self, N, B, q_init, qd_init, c_init, q_min, q_term_min, q_term_max, qd_term_min, qd_term_max, jpos_min, jpos_max, kin_box, l_leg_max):
        """landing_kinodyn_bounds (pure host code): arrays [B, n] -> lbg, ubg [B, ng]"""
        dp = C.POINTER(C.c_double)
        a = [np.ascontiguousarray(v, float) for v in (q_init, qd_init, c_init, q_min, q_term_min, q_term_max, qd_term_min, qd_term_max, jpos_min, jpos_max, kin_box, l_leg_max)]
        ng = 48 + 141 * (N - 1) + 117
        lb = np.zeros((B, ng)); ub = np.zeros((B, ng))
        fn = self.L.lib.landing_kinodyn_bounds
        fn.argtypes = [C.c_int, C.c_int, C.c_void_p] + [dp] * 14
        self.L._check(fn(N, B, None, *[v.ctypes.data_as(dp) for v in a], lb.ctypes.data_as(dp), ub.ctypes.data_as(dp)), "landing_kinodyn_bounds")
        return lb, ub

    # ---- CasADi-external face of the kinodynamic NLP (include/landing_nlp.h, round 6): what landingCtrller_KNITRO_mi355x.so forwards to
    def kinodyn_casadi_np(self, N):
        fn = self.L.lib.landing_kinodyn_casadi_np; fn.restype = C.c_longlong; fn.argtypes = [C.c_int]
        return int(fn(N))

    def kinodyn_casadi_pattern(self, N, which):
        lp = C.POINTER(C.c_longlong)
        ci, r, nnz = lp(), lp(), C.c_longlong()
        fn = self.L.lib.landing_kinodyn_casadi_pattern
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(lp), C.POINTER(lp), lp]
        self.L._check(fn(self.L.ctx, N, which, C.byref(ci), C.byref(r), C.byref(nnz)), "landing_kinodyn_casadi_pattern")
        nx = 48 * N + 12
        return np.array(ci[:nx + 1], np.int64), np.array(r[:nnz.value], np.int64)

    def kinodyn_casadi_bounds(self, N, p):
        dp = C.POINTER(C.c_double)
        p = np.ascontiguousarray(p, float); ng = 48 + 141 * (N - 1) + 117
        lb, ub = np.zeros(ng), np.zeros(ng)
        fn = self.L.lib.landing_kinodyn_casadi_bounds
        fn.argtypes = [C.c_int, C.c_void_p, dp, dp, dp]
        self.L._check(fn(N, None, p.ctypes.data_as(dp), lb.ctypes.data_as(dp), ub.ctypes.data_as(dp)), "landing_kinodyn_casadi_bounds")
        return lb, ub

    def kinodyn_casadi_eval(self, N, x, p, lam_f=1.0, lam_g=None, want=("f", "g", "grad_f", "jac", "hess", "ggx", "ggp")):
        """landing_kinodyn_casadi_eval_host: one problem, host arrays; returns a dict of the requested outputs (jac / hess as CCS nonzeros)"""
        dp = C.POINTER(C.c_double)
        nx, ng, npar = 48 * N + 12, 48 + 141 * (N - 1) + 117, self.kinodyn_casadi_np(N)
        x = np.ascontiguousarray(x, float); p = np.ascontiguousarray(p, float)
        assert x.shape == (nx,) and p.shape == (npar,)
        nj, nh = (len(self.kinodyn_casadi_pattern(N, w)[1]) for w in (0, 1))
        size = dict(f=1, g=ng, grad_f=nx, jac=nj, hess=nh, ggx=nx, ggp=npar)
        out = {k: np.zeros(size[k]) for k in want}
        ptr = lambda k: out[k].ctypes.data_as(dp) if k in out else None
        lf = C.c_double(lam_f)
        lg = np.ascontiguousarray(lam_g, float) if lam_g is not None else None
        fn = self.L.lib.landing_kinodyn_casadi_eval_host
        fn.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp] + [dp] * 7
        self.L._check(fn(self.L.ctx, N, x.ctypes.data_as(dp), p.ctypes.data_as(dp), C.byref(lf), lg.ctypes.data_as(dp) if lg is not None else None,
                         *[ptr(k) for k in ("f", "g", "grad_f", "jac", "hess", "ggx", "ggp")]), "landing_kinodyn_casadi_eval_host")
        if "f" in out:
            out["f"] = float(out["f"][0])
        return out

    def kinodyn_pattern(self, N, which):
        """landing_kinodyn_pattern: CCS (colind, row) of jac_g_x (which = 0) or triu(hess_gamma_x_x) (which = 1)"""
        nx = 48 * N + 12
        colind = np.zeros(nx + 1, np.int64); nnz = C.c_longlong()
        lp = C.POINTER(C.c_longlong)
        fn = self.L.lib.landing_kinodyn_pattern
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, lp, lp, lp]
        self.L._check(fn(self.L.ctx, N, which, colind.ctypes.data_as(lp), None, C.byref(nnz)), "landing_kinodyn_pattern")
        row = np.zeros(nnz.value, np.int64)
        self.L._check(fn(self.L.ctx, N, which, colind.ctypes.data_as(lp), row.ctypes.data_as(lp), C.byref(nnz)), "landing_kinodyn_pattern")
        return colind, row

    def kinodyn_block_nonzeros(self):
        """landing_kinodyn_block_nonzeros: the solver's table of the structural non-zeros of a Jacobian block's inequality rows -> ((rows, cols) of a middle interval,
        (rows, cols) of the last one); rows count from the top of the 141 x 72 block"""
        n0, n1 = C.c_int(), C.c_int(); buf = np.zeros(2 * 1280, np.uint8)
        fn = self.L.lib.landing_kinodyn_block_nonzeros
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_ubyte)]
        self.L._check(fn(self.L.ctx, C.byref(n0), C.byref(n1), buf.ctypes.data_as(C.POINTER(C.c_ubyte))), "landing_kinodyn_block_nonzeros")
        e = buf[:2 * (n0.value + n1.value)].reshape(-1, 2).astype(int)
        return (e[:n0.value, 0], e[:n0.value, 1]), (e[n0.value:, 0], e[n0.value:, 1])

    def leg_ik(self, npts, d_q6, d_c, d_jpos, d_res=0, iters=12, jmin=None, jmax=None, stream=0):
        jmin = np.ascontiguousarray(JPOS_MIN[:3] if jmin is None else jmin, float); jmax = np.ascontiguousarray(JPOS_MAX[:3] if jmax is None else jmax, float)
        dp = C.POINTER(C.c_double)
        self.L._check(self.L.lib.landing_leg_ik_batch(self.L.ctx, npts, d_q6, d_c, jmin.ctypes.data_as(dp), jmax.ctypes.data_as(dp), iters, d_jpos, d_res or None, stream or None),
                      "landing_leg_ik_batch")

    def kinodynamic_screen(self, N, x_star, kin_tol=0.01):
        """SRBM solutions x* [B, nx] (device tensor) -> per member: worst FK residual after IK, worst torque ratio |tau| / tau_max and
        whether every stage satisfies the FK band and the torque limits of landing_optimization.m:165-171,186-187"""
        import torch
        B = x_star.shape[0]; nX = 12 * (N + 1); n = B * N
        X = x_star[:, :nX].reshape(B, N + 1, 12); U = x_star[:, nX:].reshape(B, N, 24)
        q6 = X[:, :N, :6].reshape(n, 6).contiguous(); c = U[:, :, :12].reshape(n, 12).contiguous(); f = U[:, :, 12:].reshape(n, 12).contiguous()
        mk = lambda *s: torch.zeros(*s, device=x_star.device, dtype=torch.float64)
        jp, res, tau, err = mk(n, 12), mk(n, 4), mk(n, 12), mk(n, 12)
        st = torch.cuda.current_stream().cuda_stream
        self.leg_ik(n, q6.data_ptr(), c.data_ptr(), jp.data_ptr(), res.data_ptr(), stream=st)
        self.kinodyn_rows(n, q6.data_ptr(), c.data_ptr(), f.data_ptr(), jp.data_ptr(), 0, err.data_ptr(), tau.data_ptr(), stream=st)
        ratio = (tau.abs() / torch.as_tensor(TAU_MAX, device=x_star.device)).reshape(B, N * 12).max(dim=1).values
        fk_bad = err.abs().reshape(B, N * 12).max(dim=1).values
        return dict(jpos=jp.reshape(B, N, 12), fk_err_max=fk_bad, torque_ratio_max=ratio, feasible=(fk_bad <= kin_tol) & (ratio <= 1.0))
