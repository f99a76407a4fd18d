"""SURVEY row a14 on CPU: the reference's 21-argument solver function as C entry points.
  * landing_pack_args21 (pure host code of the product library) packs p in Opti's active-parameter order: equal to the
    Python mirror pack_params() and to the golden p of the reference-generated fixtures;
  * matlab/landing_solve_mex.c is compiled against a test stub of mex.h (no MATLAB in the image) and its mexFunction is
    CALLED with MATLAB-shaped arrays; linked to the host-emulated build of the kernels it must return exactly what
    landing_solve_batch_host returns for the packed p."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, lc

PKG = os.path.join(ROOT, "landing-controller_amd")


@pytest.fixture(scope="module")
def libs():
    subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "all", "emu"], check=True, capture_output=True)
    return os.path.join(PKG, "liblanding_mi355x.so"), os.path.join(ROOT, "tests", "emu", "liblanding_emu.so")


def test_pack_args21_equals_python_mirror(libs):
    capi, P = lc("capi"), lc("problem")
    N, B = 20, 3
    lib = capi.load(libs[0])
    args = P.make_args21(B, N, 0.6, seed=5)
    a, keep, b = capi.matlab_args21(N, args)
    p = np.zeros((B, P.n_p(N)))
    assert b == B and lib.landing_pack_args21(N, B, C.byref(a), p.ctypes.data_as(C.POINTER(C.c_double))) == 0
    Pb, X0, q, qd = P.make_batch(B, N, 0.6, seed=5)
    assert np.array_equal(p, Pb)                                    # bit-identical to the callers' packing
    assert np.array_equal(args["x0"].T, X0)
    # Uref is inactive: NULL is accepted; a missing active argument is refused
    a.Uref = None
    assert lib.landing_pack_args21(N, B, C.byref(a), p.ctypes.data_as(C.POINTER(C.c_double))) == 0
    a.QN = None
    assert lib.landing_pack_args21(N, B, C.byref(a), p.ctypes.data_as(C.POINTER(C.c_double))) != 0


def test_mex_gateway_compiles_and_matches_packed_path(libs, tmp_path):
    capi, P = lc("capi"), lc("problem")
    N, B = 8, 1           # small horizon: the emulated kernel runs the gateway's default options to convergence in seconds
    so = str(tmp_path / "gateway.so")
    emu_dir = os.path.dirname(libs[1])
    subprocess.run(["gcc", "-O1", "-std=c99", "-fPIC", "-shared", "-Wall", "-Werror", "-Wno-unused-function", "-I", os.path.join(ROOT, "tests", "stubs"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "matlab", "landing_solve_mex.c"),
                    os.path.join(ROOT, "tests", "stubs", "mex_driver.c"), "-o", so, "-L", emu_dir, "-llanding_emu",
                    "-Wl,-rpath," + emu_dir], check=True)
    gw = C.CDLL(so)
    args = P.make_args21(B, N, 0.6, seed=9)
    bufs = [np.asfortranarray(np.asarray(args[n], float)) for n in capi.ARGS21]
    data = (C.POINTER(C.c_double) * 21)(*[b.ctypes.data_as(C.POINTER(C.c_double)) for b in bufs])
    ndim = (C.c_int * 21)(*[b.ndim for b in bufs])
    dims = (C.c_int * 84)(*sum([list(b.shape) + [1] * (4 - b.ndim) for b in bufs], []))
    nx = P.nx(N)
    X = np.zeros((B, nx)); F = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kk = np.zeros((B, 3))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    assert gw.call_gateway(data, ndim, dims, dp(X), dp(F), ip(st), ip(it), dp(kk), nx, B) == 0
    L = capi.LandingLib(N, lib_path=libs[1])
    Pb, X0, _, _ = P.make_batch(B, N, 0.6, seed=9)
    ref = L.solve_host(Pb, X0)
    assert st.tolist() == [0] and it.tolist() == ref["iters"].tolist()
    assert np.array_equal(X, ref["x"]) and np.array_equal(F, ref["f"]) and np.array_equal(kk, ref["kkt"])
    # ... and the two C spellings of the entry point agree with it (a few iterations are enough for bit-equality)
    o = L.default_opts(); o.max_iter = 3
    r0 = L.solve_host(Pb, X0, o); r1 = L.solve_args21(args, o); r2 = L.solve_args21(args, o, spelled_out=True)
    for r in (r1, r2):
        assert np.array_equal(r["x"], r0["x"]) and np.array_equal(r["status"], r0["status"]) and np.array_equal(r["kkt"], r0["kkt"])
