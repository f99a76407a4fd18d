"""Training-pair writer for the reference's NN warm-start pipeline.

generate_data/generate_training_data_automated.m:204-219 appends, per solved drop state,
    input  column = [q_init(4:6) ; qd_init(:)]            (9 values: rpy0, omega0, v0)
    output column = [X*(:) ; U*(:) ; jpos*(:)]
to ``training_data`` and saves it after every sample.  The SRBM stage produces X* and U*; the joint
trajectories jpos* come from the KNITRO kinodynamic refinement (out of scope), so they are optional here
and the block is simply absent when not given.  Members whose status is not CONVERGED are skipped, as
the interactive "Save trajectory for training?" prompt of the reference would.
"""
import numpy as np


def training_pairs(N, q_init, qd_init, x_star, status=None, jpos_star=None):
    """-> (input [9, M], output [12(N+1)+24N(+12N), M]) for the M accepted members, column per sample."""
    q_init = np.atleast_2d(q_init); qd_init = np.atleast_2d(qd_init); x_star = np.atleast_2d(x_star)
    keep = np.ones(x_star.shape[0], bool) if status is None else (np.asarray(status) == 0)
    inp = np.concatenate([q_init[keep, 3:6], qd_init[keep, :]], axis=1).T
    out = x_star[keep].T          # x = [X(:); U(:)] is already the column-major stacking the script writes
    if jpos_star is not None:
        out = np.concatenate([out, np.atleast_2d(jpos_star)[keep].T], axis=0)
    return inp, out


def append_shard(path, inp, out):
    """append samples to an .npz shard (the reference re-saves a growing .mat after every sample,
    generate_training_data_automated.m:214-219).  The shard is rewritten atomically: a crash mid-write leaves the
    previous file intact."""
    import os
    import tempfile
    path = str(path)
    if not path.endswith(".npz"):
        path += ".npz"            # numpy appends the suffix on save but not on load: normalise once for both
    if os.path.exists(path):
        with np.load(path) as d:
            inp = np.concatenate([d["input"], inp], axis=1)
            out = np.concatenate([d["output"], out], axis=1)
    fd, tmp = tempfile.mkstemp(suffix=".npz", dir=os.path.dirname(os.path.abspath(path)))
    try:
        with os.fdopen(fd, "wb") as fh:
            np.savez_compressed(fh, input=inp, output=out)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise
    return inp.shape[1]


# ---- the reference's .mat layout and the normalisation its NN pipeline expects ------------------------------
def save_training_mat(path, inp, out):
    """`training_data.input` / `training_data.output` as generate_training_data_automated.m:204-219 grows them (one column per
    sample).  Written as a MAT v5 file through scipy (the reference passes '-V7.3', i.e. HDF5, for which this image has no
    writer; MATLAB's `load` reads both)."""
    from scipy.io import savemat
    savemat(str(path), {"training_data": {"input": np.asarray(inp, float), "output": np.asarray(out, float)}}, do_compression=True)


def load_training_mat(path):
    from scipy.io import loadmat
    td = loadmat(str(path), squeeze_me=True, struct_as_record=False)["training_data"]
    return np.atleast_2d(td.input), np.atleast_2d(td.output)


def _split_output(N, col, with_jpos):
    nX, nU = 12 * (N + 1), 24 * N
    X = col[:nX].reshape(12, N + 1, order="F"); U = col[nX:nX + nU].reshape(24, N, order="F")
    J = col[nX + nU:nX + nU + 12 * N].reshape(12, N, order="F") if with_jpos else None
    return X, U, J


def normalise(N, inp, out, mass, with_jpos=False):
    """generate_data/data_normalization.m:38-114.  inp [9, M], out [nX + nU (+ 12N), M] -> (input_n, output_n [.. + 4, M], stats):
    z-scores of the input, the states, the foot positions (and joint angles) with MATLAB's `std(x, 0, 2)` (n-1 normalisation);
    X_norm(1:2, 1) = 0; ground-reaction forces of every leg shifted to their touch-down index td = first column with f_z > 1,
    padded with the last column, divided by the body weight; td (1-based, 4 values) appended.  Deviations, stated: a zero standard
    deviation maps to 0 instead of NaN, a leg that never loads gets td = 1 (the MATLAB script errors on both)."""
    inp = np.asarray(inp, float); out = np.asarray(out, float)
    M = out.shape[1]
    sd = lambda a: np.std(a, axis=1, ddof=1) if a.shape[1] > 1 else np.zeros(a.shape[0])
    div = lambda a, s: np.divide(a, s, out=np.zeros_like(a), where=s != 0)
    mean_in, std_in = inp.mean(axis=1), sd(inp)
    mean_out, std_out = out.mean(axis=1), sd(out)
    mX, mU, mJ = _split_output(N, mean_out, with_jpos); sX, sU, sJ = _split_output(N, std_out, with_jpos)
    stats = dict(mean_input=mean_in, std_input=std_in, mean_X=mX, mean_U=mU, mean_jpos=mJ, std_X=sX, std_U=sU, std_jpos=sJ, td_scale=1, mass=mass, N=N)
    inp_n = div(inp - mean_in[:, None], np.broadcast_to(std_in[:, None], inp.shape))
    cols = []
    for e in range(M):
        X, U, J = _split_output(N, out[:, e], with_jpos)
        Un = np.zeros_like(U); td = np.ones(4)
        for leg in range(4):
            f = U[12 + 3 * leg:15 + 3 * leg]
            hit = np.nonzero(f[2] > 1.0)[0]
            t0 = int(hit[0]) if hit.size else 0                      # 0-based; td is stored 1-based as in MATLAB
            fo = np.concatenate([f[:, t0:], np.repeat(f[:, -1:], t0, axis=1)], axis=1)
            Un[12 + 3 * leg:15 + 3 * leg] = fo / (mass * 9.81)
            td[leg] = t0 + 1
        Xn = div(X - mX, sX); Xn[0:2, 0] = 0.0
        Un[:12] = div(U[:12] - mU[:12], sU[:12])
        parts = [Xn.flatten(order="F"), Un.flatten(order="F")]
        if with_jpos:
            parts.append(div(J - mJ, sJ).flatten(order="F"))
        cols.append(np.concatenate(parts + [td]))
    return inp_n, np.array(cols).T, stats


def denormalise(nn_col, stats, with_jpos=False):
    """generate_data/data_denormalization.m:17-38: one normalised output column -> (X, U, jpos)"""
    N = stats["N"]
    Xn, Un, Jn = _split_output(N, nn_col, with_jpos)
    td = np.rint(nn_col[-4:]).astype(int)
    X = Xn * stats["std_X"] + stats["mean_X"]
    U = np.zeros((24, N))
    U[:12] = Un[:12] * stats["std_U"][:12] + stats["mean_U"][:12]
    for leg in range(4):
        fo = Un[12 + 3 * leg:15 + 3 * leg]
        t0 = td[leg] - 1
        U[12 + 3 * leg:15 + 3 * leg] = np.concatenate([np.zeros((3, t0)), fo[:, :N - t0]], axis=1) * (stats["mass"] * 9.81)
    J = Jn * stats["std_jpos"] + stats["mean_jpos"] if with_jpos else None
    return X, U, J


def write_member_log(path, status, iters, kkt, f=None, extra=None):
    """JSON lines, one record per batch member (SURVEY section 5 hook: iterations, pr_inf, du_inf, compl, status)"""
    import json
    with open(path, "a") as fh:
        for m in range(len(status)):
            rec = {"member": m, "status": int(status[m]), "iterations": int(iters[m]), "pr_inf": float(kkt[m][0]), "du_inf": float(kkt[m][1]), "compl": float(kkt[m][2])}
            if f is not None:
                rec["f"] = float(f[m])
            if extra:
                rec.update(extra)
            fh.write(json.dumps(rec) + "\n")


def generate_streamed(N, n_batches, B, shard_path, seed0=0, T=0.6, dt_grid="uniform", law="main", opts=None, depth=2, device=0, log_path=None, **form):
    """The data-generation loop of generate_data/generate_training_data_automated.m:38-219 as a STREAM of batches through one GPU (round 6): the drop states of batch i + 1 are sampled on
    the host and queued (`pipeline.BatchPipeline` over the library's landing_stream_submit / _wait: one context, `depth` launches in flight) while batch i is solved; the converged members of
    every batch that leaves the pipeline are appended to the shard in the reference's layout (training_pairs / append_shard), every member is logged (write_member_log).  Returns the
    counts per status and the number of samples written.  Results per batch are bit-identical to one solve at a time (tests/test_gpu_dataset.py)."""
    import importlib
    problem = importlib.import_module(__package__ + ".problem"); pipeline = importlib.import_module(__package__ + ".pipeline")
    consts = problem.production_constants(law) if dt_grid == "reference" else None
    pipe = pipeline.BatchPipeline(N, depth=depth, device=device, opts=opts, **form)
    meta, counts, written = {}, {}, 0

    def take(res):
        nonlocal written
        q, qd = meta.pop(res["tag"])
        inp, out = training_pairs(N, q, qd, res["x"], res["status"])
        if inp.shape[1]:
            written = append_shard(shard_path, inp, out)
        if log_path:
            write_member_log(log_path, res["status"], res["iters"], res["kkt"], res["f"], extra={"batch": int(res["tag"])})
        for s in res["status"]:
            counts[int(s)] = counts.get(int(s), 0) + 1
    try:
        for i in range(n_batches):
            P, X0, q, qd = problem.make_batch(B, N, T, seed=seed0 + i, consts=consts, dt_grid=dt_grid, law=law)
            meta[i] = (q, qd)
            done = pipe.submit(P, X0, tag=i)
            if done is not None:
                take(done)
        for done in pipe.drain():
            take(done)
    finally:
        pipe.close()
    return dict(status_counts=counts, samples_written=written, batches=n_batches)
