"""SURVEY row a14 on the GPU: the reference's 21-argument solver function, called through the C ABI with arrays shaped
exactly as the MATLAB call sites pass them (generate_landingCtrller_IPOPT.m:323-327, landing_optimization.m:305-311),
must return bit-identical x*, f*, status to the packed-p path it wraps."""
import numpy as np
import pytest

from conftest import lc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,B", [(20, 6), (40, 16)])
def test_args21_equals_packed_path(N, B, oracle_mod):
    capi, P = lc("capi"), lc("problem")
    L = capi.LandingLib(N, device=0)
    args = P.make_args21(B, N, 0.6, seed=77)
    assert args["Xref"].shape == (12, N + 1, B) and args["dt"].shape == (1, N, B) and args["x0"].shape == (P.nx(N), B)
    Pb, X0, _, _ = P.make_batch(B, N, 0.6, seed=77)
    ref = L.solve_host(Pb, X0)
    r1 = L.solve_args21(args)
    r2 = L.solve_args21(args, spelled_out=True)
    noU = dict(args); noU["Uref"] = None                     # inactive Opti parameter: may be omitted
    r3 = L.solve_args21(noU)
    for r in (r1, r2, r3):
        for k in ("x", "f", "status", "iters", "kkt"):
            assert np.array_equal(r[k], ref[k]), k
    ok = ref["status"] == 0
    assert ok.sum() >= B - 1, ref["status"]                  # (N = 20: a few per cent of the synthetic drop states do not solve)
    O = oracle_mod.Oracle(N)
    for b in np.nonzero(ok)[0][::5]:                          # and it is a KKT point of the reference-equivalent NLP
        assert max(ref["kkt"][b]) <= 1e-6 * 1.0001 and abs(O.f(r1["x"][b], Pb[b]) - r1["f"][b]) < 1e-12
    L.close()


def test_args21_single_member_matlab_call_shape():
    """B = 1 exactly as landing_optimization.m:305-311 calls it: 2-D arrays, no batch axis"""
    capi, P = lc("capi"), lc("problem")
    N = 20
    L = capi.LandingLib(N, device=0)
    a = P.make_args21(1, N, 0.6, seed=3)
    flat = {k: (v[..., 0] if v.ndim == 3 else v) for k, v in a.items()}       # Xref 12x(N+1), dt 1xN, q_min 6x1, mu 1x1 ...
    flat = {k: (v if v.ndim == 2 else v.reshape(-1, 1)) for k, v in flat.items()}
    r = L.solve_args21(flat)
    Pb, X0, _, _ = P.make_batch(1, N, 0.6, seed=3)
    ref = L.solve_host(Pb, X0)
    assert np.array_equal(r["x"], ref["x"]) and r["status"][0] == 0
    L.close()


def test_multi_device_entry_two_contexts_bit_identical(tmp_path):
    """landing_multi_solve_args21 / landing_solve_21_multi on the 1-GPU box with devices = {0, 0}: two contexts, two host threads,
    contiguous (ragged) shards -- x*, f*, lam_g, status, iterations, KKT bit-identical to the single-context call; through the mex
    gateway (matlab/landing_solve_mex.c against the mex.h stub) with shared constants, an options struct and a device list too."""
    import os
    from conftest import ROOT, MexGateway
    capi, P = lc("capi"), lc("problem")
    N, B = 20, 33
    L = capi.LandingLib(N, device=0)
    c = P.production_constants("main")
    args = P.make_args21(B, N, 0.6, seed=31, consts=c, dt_grid="reference")
    Pb, X0, _, _ = P.make_batch(B, N, 0.6, seed=31, consts=c, dt_grid="reference")
    o = L.default_opts(); o.max_iter = 300
    ref = L.solve_host(Pb, X0, o)
    assert (ref["status"] == 0).sum() >= B - 1
    for devs, one_call in (([0, 0], False), ([0, 0, 0], True), ([0], True)):
        r = L.solve_args21_multi(args, devs, o, one_call=one_call)
        for k in ("x", "f", "lam_g", "status", "iters", "kkt"):
            assert np.array_equal(r[k], ref[k]), (devs, k)
    gw = MexGateway(tmp_path, os.path.join(ROOT, "landing-controller_amd"), "landing_mi355x")
    shared = dict(args)
    for n in ("q_min", "q_max", "qd_min", "qd_max", "q_term_min", "q_term_max", "qd_term_min", "qd_term_max", "QN", "mu", "l_leg_max", "f_max", "mass", "Ib", "Ib_inv", "dt"):
        shared[n] = np.asarray(args[n])[..., 0]
    out = gw.call(N, shared, capi.ARGS21, opts=dict(max_iter=300), devices=[0, 0])
    for k in ("x", "f", "lam_g", "status", "iters", "kkt"):
        assert np.array_equal(out[k], ref[k]), k
    with pytest.raises(RuntimeError, match="landing_multi_create"):
        L.solve_args21_multi(args, [0, 99], o)
    L.lib.landing_multi_release_cached()
    L.close()
