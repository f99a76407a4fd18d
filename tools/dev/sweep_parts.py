"""dev: the streams of one landing_eval_batch call timed alone (HIP events): residual rows, Jacobian, Hessian, all together"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40
L = capi.LandingLib(N, 0, lib_path=sys.argv[2] if len(sys.argv) > 2 else None)
for B in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024,4096,16384").split(",")]:
    P, X0, _, _ = problem.make_batch(256, N, 0.6, seed=1)
    reps = (B + 255) // 256
    P = np.tile(P, (reps, 1))[:B]; X0 = np.tile(X0, (reps, 1))[:B]
    rng = np.random.default_rng(0)
    dX = torch.tensor(X0 + 0.01 * rng.normal(size=X0.shape), device="cuda"); dP = torch.tensor(P, device="cuda"); dlam = torch.tensor(rng.normal(size=(B, L.ng)), device="cuda")
    mk = lambda *s: torch.empty(*s, device="cuda", dtype=torch.float64)
    g, gf, jac, hess = mk(B, L.ng), mk(B, L.nx), mk(B, L.nnz_jac), mk(B, L.nnz_hess)
    st = torch.cuda.current_stream().cuda_stream
    out = {"B": B}
    for name, kw in (("g", dict(d_g=g.data_ptr())), ("jac", dict(d_jac=jac.data_ptr())), ("hess", dict(d_hess=hess.data_ptr())),
                     ("all", dict(d_g=g.data_ptr(), d_grad_f=gf.data_ptr(), d_jac=jac.data_ptr(), d_hess=hess.data_ptr()))):
        run = lambda: L.eval_device(B, dX.data_ptr(), dP.data_ptr(), 0, dlam.data_ptr(), stream=st, **kw)
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        out[name + "_us"] = round(1e3 * e0.elapsed_time(e1) / 30, 1)
    print(json.dumps(out), flush=True)
