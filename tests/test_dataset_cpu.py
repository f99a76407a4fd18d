"""host-side writer of the reference's training pairs (generate_training_data_automated.m:204-219)"""
import numpy as np

from conftest import lc


def test_training_pairs_layout(tmp_path):
    ds, P = lc("dataset"), lc("problem")
    N, B = 20, 5
    Pb, X0, q, qd = P.make_batch(B, N, 0.6, seed=2)
    status = np.array([0, 1, 0, 2, 0])
    inp, out = ds.training_pairs(N, q, qd, X0, status)
    assert inp.shape == (9, 3) and out.shape == (P.nx(N), 3)
    assert np.array_equal(inp[:3, 1], q[2, 3:6]) and np.array_equal(inp[3:, 1], qd[2])
    Xs, Us = P.split_solution(N, out[:, 2])                 # X* = reshape(x(1:12(N+1)),12,N+1), U* the rest
    assert np.array_equal(Xs[:, 0], X0[4, :12]) and Us.shape == (24, N)
    f = tmp_path / "shard.npz"
    assert ds.append_shard(str(f), inp, out) == 3 and ds.append_shard(str(f), inp[:, :1], out[:, :1]) == 4
    g = tmp_path / "noext"                                    # extension-less path: both appends must land in one file
    assert ds.append_shard(str(g), inp[:, :2], out[:, :2]) == 2 and ds.append_shard(str(g), inp[:, 2:], out[:, 2:]) == 3
    with np.load(str(g) + ".npz") as d:
        assert np.array_equal(d["input"], inp) and np.array_equal(d["output"], out)
    jp = np.zeros((B, 12 * N))
    assert ds.training_pairs(N, q, qd, X0, status, jp)[1].shape[0] == P.nx(N) + 12 * N


def test_reference_mat_layout_and_normalisation(tmp_path):
    """training_data.{input,output} as the reference grows them (generate_training_data_automated.m:204-219) and the
    normalisation / denormalisation of its NN pipeline (data_normalization.m:38-114, data_denormalization.m:17-38)"""
    ds, P = lc("dataset"), lc("problem")
    N, B = 20, 12
    rng = np.random.default_rng(0)
    Pb, X0, q, qd = P.make_batch(B, N, 0.6, seed=3)
    x = X0.copy()
    nX = 12 * (N + 1)
    for b in range(B):                                 # synthetic "solutions": forces that load after a touch-down index
        U = x[b, nX:].reshape(24, N, order="F")
        for leg in range(4):
            t0 = 2 + (b + leg) % 5
            U[12 + 3 * leg:15 + 3 * leg, t0:] = np.array([[3.0], [-2.0], [40.0]]) + rng.normal(size=(3, N - t0))
        x[b, nX:] = U.flatten(order="F")
    x[:, :nX] += 0.01 * rng.normal(size=(B, nX))
    jp = np.tile([0, -0.8, 1.6] * 4, (B, N)).reshape(B, 12 * N) + 0.05 * rng.normal(size=(B, 12 * N))
    inp, out = ds.training_pairs(N, q, qd, x, None, jp)
    f = tmp_path / "training_data_landing.mat"
    ds.save_training_mat(f, inp, out)
    i2, o2 = ds.load_training_mat(f)
    assert np.array_equal(i2, inp) and np.array_equal(o2, out) and out.shape == (nX + 24 * N + 12 * N, B)
    inp_n, out_n, stats = ds.normalise(N, inp, out, mass=8.252, with_jpos=True)
    assert out_n.shape == (out.shape[0] + 4, B) and np.allclose(inp_n.mean(axis=1), 0, atol=1e-12)
    assert np.allclose(np.std(inp_n, axis=1, ddof=1)[3:], 1.0)                      # z-scores with MATLAB's std(x, 0, 2)
    for e in (0, 5, 11):
        X, U, J = ds.denormalise(out_n[:, e], stats, with_jpos=True)
        Xs, Us = P.split_solution(N, x[e])
        m = np.ones((12, N + 1), bool); m[0:2, 0] = False                           # X_norm(1:2, 1) = 0 is not invertible by design
        assert np.allclose(X[m], Xs[m], atol=1e-12) and np.allclose(U[:12], Us[:12], atol=1e-12) and np.allclose(J, jp[e].reshape(12, N, order="F"), atol=1e-12)
        for leg in range(4):                                                          # forces: exact from touch-down on, zero before it
            t0 = int(out_n[-4 + leg, e]) - 1
            assert np.allclose(U[12 + 3 * leg:15 + 3 * leg, t0:], Us[12 + 3 * leg:15 + 3 * leg, t0:], atol=1e-12)
            assert not U[12 + 3 * leg:15 + 3 * leg, :t0].any()
    log = tmp_path / "members.jsonl"
    ds.write_member_log(log, [0, 1], [61, 300], [[1e-9, 2e-7, 1e-7], [1e-2, 3.0, 0.1]], f=[0.1, 0.2])
    import json
    recs = [json.loads(l) for l in open(log)]
    assert recs[1]["status"] == 1 and recs[0]["iterations"] == 61 and recs[0]["du_inf"] == 2e-7
