import importlib, sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("landing-controller_amd.capi"); P = importlib.import_module("landing-controller_amd.problem"); Cn = importlib.import_module("landing-controller_amd.constants")
from oracle import oracle as orc
L = capi.LandingLib(40, 0)
Pb, X0, _, _ = P.make_batch(256, 40, 0.6, seed=20211)
for mr in (2, 8):
    for rp in (60, 0):
        o = L.default_opts(); o.bound_frac = 0.5; o.max_iter = 600; o.max_resets = mr; o.restart_period = rp
        r = L.solve_host(Pb, X0, o); print('bound_frac 0.5 max_resets', mr, 'restart', rp, 'conv', (r['status'] == 0).sum(), 'mean it', r['iters'].mean())
d = np.load(os.path.join(ROOT, "tests", "golden", "n40_golden.npz"))
rc = dict(QX=[0] * 12, Qc=[0, 0, 0], Qf=[1e-4, 1e-4, 1e-3], f_ref=[0, 0, 0]); kb = (0.05, 0.05, 0.27)
O = orc.Oracle(40, kin_box=kb, run_cost=rc); L2 = capi.LandingLib(40, device=0, kin_box=kb, run_cost=rc)
mass, Ib, Ibi = Cn.robot_constants(); N = 40
Ps, X0s = [], []
for x in d["x"]:
    X = x[:12 * 41].reshape(12, 41, order="F"); q0, qd0 = X[:6, 0], X[6:, 0]
    Xref = np.zeros((12, N + 1))
    for i in range(6):
        Xref[i] = np.linspace(q0[i], [0, 0, 0.2, 0, 0, 0][i], N + 1); Xref[6 + i] = np.linspace(qd0[i], 0.0, N + 1)
    c_ref = P.SIDE_SIGN * np.tile([0.2, 0.1, -0.35], 4); Uref = np.zeros((24, N))
    for j in range(12): Uref[j] = Xref[j % 3, :-1] + c_ref[j]
    Ps.append(P.pack_params(N, Xref, np.full(N, 0.015), [-10, -10, .15, -10, -10, -10], [10, 10, 1, 10, 10, 10], [-10, -10, -10, -40, -40, -40], [10, 10, 10, 40, 40, 40], q0, qd0,
                            [-10, -10, .15, -.1, -.1, -10], [10, 10, 5, .1, .1, 10], [-10, -10, -10, -40, -40, -40], [10, 10, 10, 40, 40, 40], [0, 0, 100, 100, 100, 0, 10, 10, 10, 10, 10, 10], 1.0, .35, 250., mass, Ib, Ibi))
    X0s.append(np.concatenate([Xref.flatten(order="F"), Uref.flatten(order="F")]))
Ps, X0s = np.array(Ps), np.array(X0s)
r = L2.solve_host(Ps, X0s); ok = r["status"] == 0
same = better = 0
for b in np.nonzero(ok)[0]:
    f_ref = O.f(d["x"][b], Ps[b]); same += abs(r["f"][b] - f_ref) <= 1e-3 * f_ref; better += r["f"][b] <= f_ref * 1.001
    print(b, 'f ours %.6f ref %.6f iters %d' % (r["f"][b], f_ref, r["iters"][b]))
print('known answer: ok', ok.sum(), 'of', len(Ps), 'same', same, 'better', better, 'status', r['status'])
