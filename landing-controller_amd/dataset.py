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
