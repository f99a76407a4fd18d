"""Development probe: iteration counts of many fresh batches with their drop states -> gpurun_out/iters_fit.npz (dispatch-order studies)."""
import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N, B, nb = 40, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = capi.LandingLib(N, 0); o = L.default_opts(); o.max_iter = 300
x = torch.empty(B, L.nx, device="cuda", dtype=torch.float64); st = torch.empty(B, device="cuda", dtype=torch.int32); it = torch.empty_like(st)
Q, QD, IT = [], [], []
for b in range(nb):
    P, X0, q, qd = problem.make_batch(B, N, 0.6, seed=100000 + b)
    dP, dX0 = torch.tensor(P, device="cuda"), torch.tensor(X0, device="cuda")
    L.solve_device(B, dP.data_ptr(), dX0.data_ptr(), o, x.data_ptr(), 0, 0, st.data_ptr(), it.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    Q.append(q); QD.append(qd); IT.append(it.cpu().numpy().astype(float))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "iters_fit.npz"), q=np.array(Q), qd=np.array(QD), it=np.array(IT))
print("saved", nb, "batches; mean iterations %.2f" % np.mean(IT))
