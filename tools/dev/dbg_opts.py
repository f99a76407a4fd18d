import importlib, sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B = 40, 1024
L = capi.LandingLib(N, 0)
P, X0, _, _ = problem.make_batch(B, N, 0.6, seed=20211)
for jc, sr in ((0, 0), (2, 3), (2, 0), (0, 3)):
    o = L.default_opts(); o.max_iter = 300; o.jam_clip = jc; o.stag_relief = sr
    r = L.solve_host(P, X0, o); it = r["iters"]
    print(jc, sr, "sizeof", __import__("ctypes").sizeof(o), "conv", (r["status"] == 0).sum(), "mean %.2f" % it.mean(), "m304", it[304], "top", np.sort(it)[-5:])
