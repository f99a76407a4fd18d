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
    from conftest import MexGateway
    capi, P = lc("capi"), lc("problem")
    N, B = 8, 1           # small horizon: the emulated kernel runs the gateway's default options to convergence in seconds
    gw = MexGateway(tmp_path, os.path.dirname(libs[1]), "landing_emu")
    args = P.make_args21(B, N, 0.6, seed=9)
    out = gw.call(N, args, capi.ARGS21)
    L = capi.LandingLib(N, lib_path=libs[1])
    Pb, X0, _, _ = P.make_batch(B, N, 0.6, seed=9)
    ref = L.solve_host(Pb, X0)
    assert out["status"].tolist() == [0] and out["iters"].tolist() == ref["iters"].tolist()
    for k in ("x", "f", "kkt", "lam_g"):
        assert np.array_equal(out[k], ref[k]), k
    # ... and the two C spellings of the entry point agree with it (a few iterations are enough for bit-equality)
    o = L.default_opts(); o.max_iter = 3
    r0 = L.solve_host(Pb, X0, o); r1 = L.solve_args21(args, o); r2 = L.solve_args21(args, o, spelled_out=True)
    for r in (r1, r2):
        assert np.array_equal(r["x"], r0["x"]) and np.array_equal(r["status"], r0["status"]) and np.array_equal(r["kkt"], r0["kkt"])


def test_mex_gateway_validates_broadcasts_and_takes_options(libs, tmp_path):
    """class and element count of every argument are checked; arguments that hold ONE member's worth of values
    (what the reference's callers pass for q_min, mu, mass, ...) are shared by the batch; outputs are created only when asked for;
    the options struct reaches landing_solver_opts and the device list shards the batch (two contexts, emulated: sequential)."""
    from conftest import MexGateway
    capi, P = lc("capi"), lc("problem")
    N, B = 6, 3
    gw = MexGateway(tmp_path, os.path.dirname(libs[1]), "landing_emu")
    args = P.make_args21(B, N, 0.6, seed=4)
    few = dict(max_iter=3, feas_phase=0)
    full = gw.call(N, args, capi.ARGS21, opts=few)
    assert (full["iters"] == 3).all() and (full["status"] == 1).all()          # the option arrived (3 iterations, LANDING_MAX_ITER)
    # shared constants as 6x1 / 1x1 arrays, exactly as generate_training_data_automated.m:130-136 passes them
    shared = dict(args)
    for n in ("q_min", "q_max", "qd_min", "qd_max", "q_term_min", "q_term_max", "qd_term_min", "qd_term_max", "QN", "mu", "l_leg_max", "f_max", "mass", "Ib", "Ib_inv", "dt"):
        shared[n] = np.asarray(args[n])[..., 0]
    sh = gw.call(N, shared, capi.ARGS21, opts=few)
    for k in ("x", "f", "status", "iters", "kkt", "lam_g"):
        assert np.array_equal(sh[k], full[k]), k
    # sharded over two contexts: bit-identical, ragged shards (3 members over 2) included
    two = gw.call(N, shared, capi.ARGS21, opts=few, devices=[0, 0])
    for k in ("x", "f", "status", "iters", "kkt", "lam_g"):
        assert np.array_equal(two[k], full[k]), k
    # only the outputs asked for are created (the driver reports rc 2 otherwise)
    one = gw.call(N, args, capi.ARGS21, opts=few, nlhs=1)
    assert np.array_equal(one["x"], full["x"]) and not one["f"].any()
    # refusals
    bad = dict(args); bad["mu"] = np.ones((1, 2))
    with pytest.raises(RuntimeError, match=r"argument 16 \(mu\) has 2 elements"):
        gw.call(N, bad, capi.ARGS21, opts=few)
    bad = dict(args); bad["x0"] = np.zeros((P.nx(N) - 1, B))
    with pytest.raises(RuntimeError, match=r"argument 15 \(x0\)"):
        gw.call(N, bad, capi.ARGS21, opts=few)
    with pytest.raises(RuntimeError, match="must be a full real double array"):
        gw.call(N, args, capi.ARGS21, opts=few, single=("Ib",))
    bad = dict(args); bad["Xref"] = np.zeros((11, N + 1, B))
    with pytest.raises(RuntimeError, match="Xref must be 12"):
        gw.call(N, bad, capi.ARGS21, opts=few)
    with pytest.raises(RuntimeError, match="landing_multi_create"):
        gw.call(N, args, capi.ARGS21, opts=few, devices=[7])      # the emulation has one device


def test_multi_device_entry_points_shard_ranges_and_identity(libs):
    """landing_shard_range = the contiguous split of sharding.py; landing_multi_solve_args21 over 1, 2 and 3 contexts (emulated)
    returns exactly the single-context result, ragged shards and B < n_dev included"""
    capi, P, sh = lc("capi"), lc("problem"), lc("sharding")
    lib = capi.load(libs[1])
    lo, hi = C.c_int(), C.c_int()
    for B in (0, 1, 7, 8, 1024, 8191):
        for n in (1, 2, 3, 8):
            got = []
            for i in range(n):
                lib.landing_shard_range(B, n, i, C.byref(lo), C.byref(hi)); got.append((lo.value, hi.value))
            assert got[0][0] == 0 and got[-1][1] == B and all(got[i][1] == got[i + 1][0] for i in range(n - 1))
            assert max(h - l for l, h in got) - min(h - l for l, h in got) <= 1
            assert got == [sh.shard_range(B, n, i) for i in range(n)]
    N, B = 6, 4
    L = capi.LandingLib(N, lib_path=libs[1])
    args = P.make_args21(B, N, 0.6, seed=2)
    o = L.default_opts(); o.max_iter = 4
    ref = L.solve_args21(args, o)
    Pb, X0, _, _ = P.make_batch(B, N, 0.6, seed=2)
    lam_ref = L.solve_host(Pb, X0, o)["lam_g"]
    for devs, one_call in (([0], False), ([0, 0], False), ([0, 0, 0], True), ([0] * 6, False)):
        r = L.solve_args21_multi(args, devs, o, one_call=one_call)
        for k in ("x", "f", "status", "iters", "kkt"):
            assert np.array_equal(r[k], ref[k]), (devs, k)
        assert np.array_equal(r["lam_g"], lam_ref)
    lib.landing_multi_release_cached()


def test_multi_device_rejects_the_25_argument_formulation_and_null_arguments(libs):
    """a context built with the N=41 script's own parameter vector (run_cost = 2, np = 37N+112) must not be fed the 21-argument
    p (13N+94 values: a heap over-read); NULL arguments are reported for every shard, not dereferenced by shards with lo > 0"""
    capi, P = lc("capi"), lc("problem")
    N, B = 6, 4
    args = P.make_args21(B, N, 0.6, seed=2)
    Lc = capi.LandingLib(N, lib_path=libs[1], kin_box=(0.05, 0.05, 0.27), run_cost=dict(QX=[1.0] * 12, Qc=[1.0] * 3, Qf=[1e-3] * 3, f_ref=[0, 0, 20.0]), ccc_params=True)
    with pytest.raises(RuntimeError, match="25 arguments"):
        Lc.solve_args21_multi(args, [0, 0])
    L = capi.LandingLib(N, lib_path=libs[1])
    bad = dict(args); bad.pop("mass")
    a, keep, _ = capi.matlab_args21(N, args)
    a.mass = None
    x = np.zeros((B, L.nx))
    dev = (C.c_int * 2)(0, 0)
    L.lib.landing_multi_create.restype = C.c_void_p
    m = L.lib.landing_multi_create(N, dev, 2, C.byref(L.form))
    assert m
    o = L.default_opts()
    rc = L.lib.landing_multi_solve_args21(C.c_void_p(m), B, C.byref(a), C.byref(o), x.ctypes.data_as(C.POINTER(C.c_double)), None, None, None, None, None)
    L.lib.landing_multi_destroy(C.c_void_p(m))
    assert rc == -1 and b"NULL argument" in L.lib.landing_last_error()


def test_stream_entry_points_are_the_single_calls_bit_for_bit(libs):
    """landing_stream_* / landing_solve_stream_host (round 6): batches through one context with several launches in flight -- every lane runs the very
    launch landing_solve_batch would, so the results are the single-call results bit for bit, whatever the number of lanes, ragged last chunk and a
    running-cost formulation (the child contexts of the lanes inherit it) included.  Emulated kernels: streams and events are no-ops there, the
    GPU test (tests/test_gpu_dataset.py) runs the same comparison with real concurrency."""
    import torch
    capi, P = lc("capi"), lc("problem")
    N, B = 6, 5
    for form in ({}, dict(run_cost=dict(QX=[0, 0, 10, 1, 1, 0, .1, .1, .1, .1, .1, .1], Qc=[1.0, 1.0, 0.5], Qf=[1e-4, 1e-4, 1e-3], f_ref=[0, 0, 20.0]))):
        L = capi.LandingLib(N, lib_path=libs[1], **form)
        Pb, X0, _, _ = P.make_batch(B, N, 0.6, seed=3)
        o = L.default_opts(); o.max_iter = 2
        ref = L.solve_host(Pb, X0, o)
        for lanes, chunk in (((2, 2), (3, 1), (2, 8)) if not form else ((2, 2),)):
            r = L.solve_stream_host(Pb, X0, o, chunk=chunk, lanes=lanes)
            for k in ("x", "f", "lam_g", "status", "iters", "kkt"):
                assert np.array_equal(r[k], ref[k]), (lanes, chunk, k)
        # device-pointer form: three submissions on two lanes, waited for in another order
        S = L.stream(2)
        assert S.lanes == 2
        dP, dX0 = torch.tensor(Pb), torch.tensor(X0)
        outs = [(torch.zeros(B, L.nx, dtype=torch.float64), torch.zeros(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32)) for _ in range(3)]
        tk = [S.submit(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), d_status=st.data_ptr(), d_iters=it.data_ptr()) for x, st, it in outs]
        assert tk == [0, 1, 2]
        S.wait(tk[2]); S.wait(tk[0]); S.sync()
        for x, st, it in outs:
            assert np.array_equal(x.numpy(), ref["x"]) and np.array_equal(st.numpy(), ref["status"]) and np.array_equal(it.numpy(), ref["iters"])
        with pytest.raises(RuntimeError):
            S.wait(7)
        S.close(); L.close()
