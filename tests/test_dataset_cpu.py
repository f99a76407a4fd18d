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
