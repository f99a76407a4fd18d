import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def lc(sub=None):
    """import landing-controller_amd[.sub] (the package directory name carries a hyphen)."""
    name = "landing-controller_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


class MexGateway:
    """matlab/landing_solve_mex.c compiled against tests/stubs/mex.h and linked to `lib_dir`/lib`lib_name`.so; call() hands it
    MATLAB-shaped arrays (dict name -> array, column-major, batch = last axis) and returns its outputs (or raises with the text of
    mexErrMsgTxt)."""

    def __init__(self, tmp_dir, lib_dir, lib_name):
        import ctypes as C
        import subprocess
        self.C = C
        so = os.path.join(str(tmp_dir), "gateway_%s.so" % lib_name)
        subprocess.run(["gcc", "-O1", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-shared", "-Wall", "-Werror", "-Wno-unused-function", "-I", os.path.join(ROOT, "tests", "stubs"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "matlab", "landing_solve_mex.c"),
                        os.path.join(ROOT, "tests", "stubs", "mex_driver.c"), "-o", so, "-L", lib_dir, "-l" + lib_name,
                        "-Wl,-rpath," + lib_dir], check=True)
        self.gw = C.CDLL(so)
        self.gw.gateway_error.restype = C.c_char_p

    def call(self, N, args, names, opts=None, devices=None, nlhs=6, single=()):
        import numpy as np
        C = self.C
        bufs = [np.asfortranarray(np.asarray(args[n], float)) for n in names]
        bufs = [b if b.ndim >= 2 else b.reshape(-1, 1) for b in bufs]
        B = bufs[0].shape[2] if bufs[0].ndim > 2 else 1
        dpt = C.POINTER(C.c_double)
        data = (dpt * 21)(*[b.ctypes.data_as(dpt) for b in bufs])
        ndim = (C.c_int * 21)(*[b.ndim for b in bufs])
        dims = (C.c_int * 84)(*sum([list(b.shape) + [1] * (4 - b.ndim) for b in bufs], []))
        cls = (C.c_int * 21)(*[7 if n in single else 6 for n in names])
        nx, ng = 36 * N + 12, 104 * N + 12
        X = np.zeros((B, nx)); F = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kk = np.zeros((B, 3)); lam = np.zeros((B, ng))
        dp = lambda a: a.ctypes.data_as(dpt); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        if opts is None and devices is None:
            n_opt, on, ov = -1, None, None
        else:
            items = list((opts or {}).items())
            n_opt = len(items)
            on = (C.c_char_p * max(n_opt, 1))(*[k.encode() for k, _ in items]); ov = (C.c_double * max(n_opt, 1))(*[float(v) for _, v in items])
        dv = (C.c_double * max(len(devices or []), 1))(*[float(d) for d in (devices or [])])
        rc = self.gw.call_gateway(data, ndim, dims, cls, n_opt, on, ov, len(devices or []), dv, nlhs, dp(X), dp(F), ip(st), ip(it), dp(kk), dp(lam), nx, ng, B)
        if rc == 1:
            raise RuntimeError(self.gw.gateway_error().decode())
        assert rc == 0, "the gateway created an output that was not asked for"
        return dict(x=X, f=F, status=st, iters=it, kkt=kk, lam_g=lam)
