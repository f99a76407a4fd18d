"""BASELINE configs[3] shape: full 18-DoF floating-base dynamics and its linearisation at every knot of every member
(N = 40 knots x 1024 members = 40 960 configurations).  Times H / C / qdd, the exact (forward-mode) linearisation and the
central-difference one with HIP events.    python tools/bench_rbd.py [--members 1024]"""
import argparse, importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); rbd = importlib.import_module("landing-controller_amd.rbd")
ap = argparse.ArgumentParser(); ap.add_argument("--members", type=int, default=1024); ap.add_argument("--reps", type=int, default=5); a = ap.parse_args()
N = 40; n = N * a.members
L = capi.LandingLib(N, 0); R = rbd.Rbd(L)
rng = np.random.default_rng(11)
q = np.zeros((n, 18)); q[:, 2] = 0.3 + 0.2 * rng.random(n); q[:, :2] = 0.3 * rng.normal(size=(n, 2)); q[:, 3:6] = 0.5 * rng.normal(size=(n, 3))
q[:, 6:] = np.tile([0.0, -0.8, 1.6], 4) + 0.3 * rng.normal(size=(n, 12))
t = lambda v: torch.tensor(v, device="cuda")
dq, dqd, dtau, df = t(q), t(rng.normal(size=(n, 18))), t(5 * rng.normal(size=(n, 18))), t(np.tile([3.0, -2.0, 25.0], 4) + 6 * rng.normal(size=(n, 12)))
mk = lambda *s: torch.zeros(*s, device="cuda", dtype=torch.float64)
H, Cb, qdd, A, A2, Hinv = mk(n, 18, 18), mk(n, 18), mk(n, 18), mk(n, 18, 36), mk(n, 18, 36), mk(n, 18, 18)
st = torch.cuda.current_stream().cuda_stream
def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps
P = lambda x: x.data_ptr()
t_hc = timed(lambda: R.fb_dynamics(n, P(dq), P(dqd), P(dtau), P(df), d_H=P(H), d_C=P(Cb), d_qdd=P(qdd), stream=st))
t_ex = timed(lambda: R.fb_dynamics(n, P(dq), P(dqd), P(dtau), P(df), d_qdd=P(qdd), d_A=P(A), d_Hinv=P(Hinv), fd_h=0.0, stream=st))
t_fd = timed(lambda: R.fb_dynamics(n, P(dq), P(dqd), P(dtau), P(df), d_A=P(A2), d_Hinv=P(Hinv), fd_h=1e-6, stream=st))
dev = (A - A2).abs().max().item() / max(1.0, A2.abs().max().item())
print(json.dumps({"workload": "18-DoF floating-base dynamics at N=40 x %d members = %d configurations (BASELINE configs[3] shape), fp64" % (a.members, n),
                  "H_C_qdd_ms": t_hc, "linearisation_exact_ms": t_ex, "linearisation_central_differences_ms": t_fd,
                  "linearisations_per_s_exact": n / (t_ex * 1e-3), "linearisations_per_s_central_differences": n / (t_fd * 1e-3),
                  "exact_vs_central_difference_rel_dev": dev,
                  "note": "exact = qdd, H^-1 and d qdd / d [q; qd] = -H^-1 dID/dz by forward-mode tangents; central differences = 72 forward-dynamics evaluations per knot"}))
