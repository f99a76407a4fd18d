"""Development lab driver (test infrastructure): builds tests/dev/ipm_lab.c against the oracle's functions and reports the
iteration-count distribution / factorisation work of a batch for the switches given in the environment (see ipm_lab.c).
  LAB_SOC=4 LAB_THMIN=1 python tests/dev/ipm_lab.py [B] [seed] [N]
"""
import ctypes as C, importlib, os, subprocess, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc
problem = importlib.import_module("landing-controller_amd.problem")


def build():
    os.makedirs(os.path.join(HERE, "_lab"), exist_ok=True)
    out = os.path.join(HERE, "_lab", "liblanding_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", "landing_oracle.c"), os.path.join(HERE, "ipm_lab.c")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-std=c11", "-o", out] + srcs + ["-lm"], check=True)
    return out


def solve(O, P, X0, threads=8, max_iter=300, tol=None, lam0=None, **kw):
    B = P.shape[0]
    o = orc._SolverOpts(); O.lib.lo_solver_opts_default(C.byref(o)); o.max_iter = max_iter
    if tol: o.tol = tol
    for k, v in kw.items(): setattr(o, k, v)
    if os.environ.get("LAB_RESETDU"): o.reset_du = float(os.environ["LAB_RESETDU"])
    if os.environ.get("LAB_MAXRESETS"): o.max_resets = int(os.environ["LAB_MAXRESETS"])
    for nm in ("delta_dec", "delta_inc", "delta_inc_first", "delta_init", "theta_mu", "kappa_mu", "kappa_eps", "bound_push", "bound_frac", "mu_init", "tau_min"):
        if os.environ.get("LAB_" + nm.upper()): setattr(o, nm, float(os.environ["LAB_" + nm.upper()]))
    x = np.zeros((B, O.nx)); lam = np.zeros((B, O.ng)) if lam0 is None else np.ascontiguousarray(lam0, float).copy(); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
    kkt = np.zeros((B, 3)); cnt = np.zeros(5, np.int64)
    ip = C.POINTER(C.c_int)
    O.lib.lo_solve_batch(O._F, C.c_int(B), orc._p(np.ascontiguousarray(P)), orc._p(np.ascontiguousarray(X0)), C.byref(o), C.c_int(threads), orc._p(x), orc._p(lam),
                         st.ctypes.data_as(ip), it.ctypes.data_as(ip), orc._p(kkt), cnt.ctypes.data_as(C.POINTER(C.c_longlong)))
    return dict(x=x, lam_g=lam, status=st, iters=it, kkt=kkt, cnt=cnt)


if __name__ == "__main__":
    build()
    orc.HERE = os.path.join(HERE, "_lab")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20211
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    only = [int(v) for v in os.environ.get("LAB_ONLY", "").split(",") if v]
    O = orc.Oracle(N)
    P, X0, q, qd = problem.make_batch(max(B, max(only) + 1 if only else 0), N, 0.6, seed=seed)
    if only: P, X0 = P[only], X0[only]
    else: P, X0 = P[:B], X0[:B]
    t = time.time(); r = solve(O, P, X0, threads=int(os.environ.get("LAB_THREADS", "8")), max_iter=int(os.environ.get("LAB_MAXIT", "300"))); dt = time.time() - t
    it = r["iters"]; c = r["status"] == 0; cnt = r["cnt"]
    kk = r["kkt"][c].max() if c.any() else float("nan")
    tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("LAB_"))
    print("%-40s conv %d/%d  iters mean %.1f med %.0f p90 %.0f p99 %.0f max %d | sweeps/it %.3f stage-elims/it %.1f (=%.3f sweeps) trials/it %.2f soc/it %.3f (acc %.3f) kktmax %.1e  %.0fs"
          % (tag, c.sum(), len(it), it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max(), cnt[0] / it.sum(), cnt[2] / it.sum(),
             cnt[2] / it.sum() / (N + 1), cnt[1] / it.sum(), cnt[3] / it.sum(), cnt[4] / max(1, it.sum()), kk, dt))
    if only or os.environ.get("LAB_LIST"): print(it.tolist())
