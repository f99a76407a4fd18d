"""SURVEY 8(f) row N4 on the GPU: the data-generation loop of generate_training_data_automated.m:38-219 as ONE batched solve --
sample drop states, solve, keep the converged members, write `training_data.{input,output}` in the reference's layout,
normalise as its NN pipeline expects and invert the normalisation."""
import numpy as np
import pytest

from conftest import lc

pytestmark = pytest.mark.gpu


def test_batched_data_generation_roundtrip(tmp_path):
    capi, P, ds = lc("capi"), lc("problem"), lc("dataset")
    N, B = 40, 96
    L = capi.LandingLib(N, device=0)
    Pb, X0, q, qd = P.make_batch(B, N, 0.6, seed=31)
    r = L.solve_host(Pb, X0)
    inp, out = ds.training_pairs(N, q, qd, r["x"], r["status"])
    M = int((r["status"] == 0).sum())
    assert M >= B - 1 and inp.shape == (9, M) and out.shape == (P.nx(N), M)
    f = tmp_path / "training_data_landing.mat"
    ds.save_training_mat(f, inp, out)
    i2, o2 = ds.load_training_mat(f)
    assert np.array_equal(o2, out)
    ds.write_member_log(tmp_path / "members.jsonl", r["status"], r["iters"], r["kkt"], r["f"])
    inp_n, out_n, stats = ds.normalise(N, i2, o2, mass=Pb[0, P.param_offsets(N)["mass"]])
    keep = np.nonzero(r["status"] == 0)[0]
    for e in (0, M // 2, M - 1):
        X, U, _ = ds.denormalise(out_n[:, e], stats)
        Xs, Us = P.split_solution(N, r["x"][keep[e]])
        m = np.ones((12, N + 1), bool); m[0:2, 0] = False
        assert np.allclose(X[m], Xs[m], atol=1e-10) and np.allclose(U[:12], Us[:12], atol=1e-10)
        for leg in range(4):
            t0 = int(out_n[-4 + leg, e]) - 1
            assert np.allclose(U[12 + 3 * leg:15 + 3 * leg, t0:], Us[12 + 3 * leg:15 + 3 * leg, t0:], atol=1e-9)
    L.close()


def test_two_batches_in_flight_give_the_same_results():
    """pipeline.BatchPipeline: batches streamed through two contexts / two streams return, in order, exactly what one solve
    at a time returns (every member is solved independently; nothing is shared between the lanes)"""
    capi, P, pl = lc("capi"), lc("problem"), lc("pipeline")
    N, B = 40, 256
    batches = [P.make_batch(B, N, 0.6, seed=900 + i)[:2] for i in range(5)]
    pipe = pl.BatchPipeline(N, depth=2)
    got = []
    for Pb, X0 in batches:
        r = pipe.submit(Pb, X0)
        if r is not None:
            got.append(r)
    got += pipe.drain()
    pipe.close()
    assert [g["tag"] for g in got] == list(range(5))
    L = capi.LandingLib(N, device=0)
    for (Pb, X0), g in zip(batches, got):
        ref = L.solve_host(Pb, X0)
        assert np.array_equal(ref["x"], g["x"]) and np.array_equal(ref["status"], g["status"]) and np.array_equal(ref["iters"], g["iters"])
    L.close()
