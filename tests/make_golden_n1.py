"""Writes tests/golden/n1_kinodyn_solutions.npz: the two kinodynamic solutions the reference keeps beside its test scripts
(optimizations/landing/test_scripts/1.5msDrop30Pitch.mat and prevSoln.mat: X_star [12, 21], U_star [24, 20] = [c; f], jpos_star [12, 20]),
as plain arrays.  Data only; run in the build container where /root/reference exists:  python tests/make_golden_n1.py"""
import os

import numpy as np
import scipy.io as sio

SRC = "/root/reference/optimizations/landing/test_scripts"
out = {}
for tag, name in (("a", "1.5msDrop30Pitch.mat"), ("b", "prevSoln.mat")):
    d = sio.loadmat(os.path.join(SRC, name))
    out["X_" + tag], out["U_" + tag], out["J_" + tag] = d["X_star"], d["U_star"], d["jpos_star"]
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "n1_kinodyn_solutions.npz"), **out)
