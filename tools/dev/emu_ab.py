"""dev probe (CPU, no GPU): two builds of the HOST EMULATION of the kernels (tests/emu) on the same members -- are the results
identical bit for bit?  Used for refactorings of landing_ipm_kernel that must not change the arithmetic (round 3: LDS-resident
iteration state; round 5: condensation fused into the backward sweep, fused row passes).
    python tools/dev/emu_ab.py BASE.so NEW.so [--cases n20,n40,rc,ccc,feas,short] [--iters K]
BASE is typically built from a git worktree of the previous commit (make -C landing-controller_amd/csrc emu there).
Each library is driven in its own child process (two copies of the emulation in one process would share the fiber runtime's symbols)."""
import argparse
import importlib
import os
import pickle
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run_cases(lib_path, cases, iters):
    capi = importlib.import_module("landing-controller_amd.capi")
    problem = importlib.import_module("landing-controller_amd.problem")
    out = {}
    for c in cases:
        if c == "n20":
            L = capi.LandingLib(20, lib_path=lib_path); P, X0, _, _ = problem.make_batch(2, 20, 0.6, seed=1); o = L.default_opts()
        elif c == "n40":
            L = capi.LandingLib(40, lib_path=lib_path); P, X0, _, _ = problem.make_batch(2, 40, 0.6, seed=20211); o = L.default_opts()
        elif c == "short":      # stops at the iteration limit, feasibility phase off
            L = capi.LandingLib(20, lib_path=lib_path); P, X0, _, _ = problem.make_batch(2, 20, 0.6, seed=3); o = L.default_opts(); o.max_iter = 7; o.feas_phase = 0
        elif c == "feas":       # iteration limit hit early -> the feasibility phase runs, then the solve restarts
            L = capi.LandingLib(20, lib_path=lib_path); P, X0, _, _ = problem.make_batch(2, 20, 0.6, seed=5); o = L.default_opts(); o.max_iter = 12
        elif c == "rc":
            rc = dict(QX=[0, 0, 10, 10, 10, 0, 1, 1, 1, 1, 1, 1], Qc=[1, 1, 1], Qf=[1e-4, 1e-4, 1e-4], f_ref=[0, 0, 20.0])
            L = capi.LandingLib(20, lib_path=lib_path, run_cost=rc); P, X0, _, _ = problem.make_batch(2, 20, 0.6, seed=2); o = L.default_opts()
        elif c == "prod":       # the reference's production grid (non-uniform dt), N = 20
            L = capi.LandingLib(20, lib_path=lib_path)
            P, X0, _, _ = problem.make_batch(2, 20, 0.6, seed=9, grid="reference") if "grid" in problem.make_batch.__code__.co_varnames else problem.make_batch(2, 20, 0.6, seed=9)
            o = L.default_opts()
        else:
            raise SystemExit("unknown case " + c)
        if iters and c not in ("short", "feas"):
            o.max_iter = iters
        r = L.solve_host(P, X0, o)
        out[c] = {k: np.asarray(v).copy() for k, v in r.items()}
        L.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("base"); ap.add_argument("new", nargs="?")
    ap.add_argument("--cases", default="n20,short,feas,rc,n40")
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--child", default="")
    a = ap.parse_args()
    cases = a.cases.split(",")
    if a.child:
        pickle.dump(run_cases(a.base, cases, a.iters), open(a.child, "wb"))
        sys.exit(0)
    res = []
    procs = []
    for i, lib in enumerate((a.base, a.new)):
        f = "/tmp/emu_ab_%d_%d.pkl" % (os.getpid(), i)
        procs.append((f, subprocess.Popen([sys.executable, __file__, os.path.abspath(lib), "--cases", a.cases, "--iters", str(a.iters), "--child", f])))
    for f, p in procs:
        if p.wait() != 0:
            raise SystemExit("child failed")
        res.append(pickle.load(open(f, "rb"))); os.remove(f)
    bad = 0
    for c in cases:
        A, B = res[0][c], res[1][c]
        line = []
        for k in sorted(A):
            same = A[k].shape == B[k].shape and np.array_equal(A[k], B[k], equal_nan=True) if A[k].dtype.kind == "f" else np.array_equal(A[k], B[k])
            if not same:
                d = np.nanmax(np.abs(A[k].astype(float) - B[k].astype(float))) if A[k].shape == B[k].shape else float("nan")
                line.append("%s DIFF(max %.3e)" % (k, d)); bad += 1
        print("%-6s status %s iters %s : %s" % (c, A["status"].tolist(), A["iters"].tolist(), "identical" if not line else ", ".join(line)))
    sys.exit(1 if bad else 0)
