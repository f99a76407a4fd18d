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
    """pipeline.BatchPipeline over the library's streaming entry points (landing_stream_create / _submit / _wait, round 6: ONE context, two launches in
    flight on the library's own lanes): batches return, in order, exactly what one solve at a time returns, bit for bit (every member is solved
    independently; a lane runs the very launch landing_solve_batch would)"""
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


def test_stream_host_entry_point_chunks_a_large_batch_bit_for_bit():
    """landing_solve_stream_host: 2500 drop states in chunks of 1024 through two lanes (uploads / downloads under the solves, ragged last chunk) == one
    landing_solve_batch_host call; the device-pointer form with three submissions on two lanes, consumed through stream waits only (no host sync between)."""
    import torch
    capi, P = lc("capi"), lc("problem")
    N, B = 40, 2500
    L = capi.LandingLib(N, device=0)
    Pb, X0, _, _ = P.make_batch(B, N, 0.6, seed=77)
    o = L.default_opts(); o.max_iter = 300
    halves = [L.solve_host(Pb[i:i + 1250], X0[i:i + 1250], o) for i in (0, 1250)]      # (<= 2048 members: one launch each, no streaming)
    ref = {k: np.concatenate([h[k] for h in halves]) for k in halves[0]}
    r = L.solve_stream_host(Pb, X0, o, chunk=1024, lanes=2)
    auto = L.solve_host(Pb, X0, o)                                                       # above 2048 members landing_solve_batch_host streams by itself
    for k in ("x", "f", "lam_g", "status", "iters", "kkt"):
        assert np.array_equal(r[k], ref[k]) and np.array_equal(auto[k], ref[k]), k
    assert (r["status"] == 0).all()
    S = L.stream(2)
    dP, dX0 = torch.tensor(Pb[:600], device="cuda"), torch.tensor(X0[:600], device="cuda")
    outs = [(torch.zeros(200, L.nx, device="cuda", dtype=torch.float64), torch.zeros(200, device="cuda", dtype=torch.int32)) for _ in range(3)]
    cur = torch.cuda.current_stream().cuda_stream
    tk = [S.submit(200, dP[200 * i:].data_ptr(), dX0[200 * i:].data_ptr(), o, x.data_ptr(), d_status=st.data_ptr(), in_stream=cur) for i, (x, st) in enumerate(outs)]
    for t in tk:
        S.wait(t, stream=cur)                    # the current stream waits; the copies below are ordered behind it
    got = [x.cpu().numpy() for x, _ in outs]
    for i in range(3):
        assert np.array_equal(got[i], ref["x"][200 * i:200 * (i + 1)])
    S.close(); L.close()


def test_streamed_data_generation_writes_the_same_shard_as_one_batch_at_a_time(tmp_path):
    """dataset.generate_streamed (round 6): four batches of 128 drop states streamed through ONE context with two launches in flight, converged members appended to the shard in the
    reference's layout -- the shard equals the one assembled from one solve at a time, column for column"""
    capi, P, ds = lc("capi"), lc("problem"), lc("dataset")
    N, B, nb = 40, 128, 4
    r = ds.generate_streamed(N, nb, B, tmp_path / "shard", seed0=4100, log_path=tmp_path / "members.jsonl")
    assert r["batches"] == nb and sum(r["status_counts"].values()) == nb * B and r["status_counts"].get(0, 0) >= nb * B - 2
    with np.load(str(tmp_path / "shard.npz")) as d:
        inp, out = d["input"], d["output"]
    assert inp.shape[1] == r["samples_written"] == r["status_counts"][0]
    L = capi.LandingLib(N, device=0)
    cols_i, cols_o = [], []
    for i in range(nb):
        Pb, X0, q, qd = P.make_batch(B, N, 0.6, seed=4100 + i)
        s = L.solve_host(Pb, X0)
        a, b = ds.training_pairs(N, q, qd, s["x"], s["status"])
        cols_i.append(a); cols_o.append(b)
    L.close()
    assert np.array_equal(inp, np.concatenate(cols_i, axis=1)) and np.array_equal(out, np.concatenate(cols_o, axis=1))
    assert len(open(tmp_path / "members.jsonl").read().splitlines()) == nb * B
