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
