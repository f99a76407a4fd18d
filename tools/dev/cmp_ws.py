"""Development probe: diff the solver workspace of two builds after K iterations (which phase goes wrong first)."""
import importlib, sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("landing-controller_amd.capi"); problem = importlib.import_module("landing-controller_amd.problem")
N = 40; K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P, X0, _, _ = problem.make_batch(1, N, 0.6, seed=20211)
def run(lib_path):
    L = capi.LandingLib(N, 0, lib_path=lib_path)
    o = L.default_opts(); o.max_iter = K
    r = L.solve_host(P, X0, o)
    ptr = C.c_void_p(); st = C.c_ulonglong()
    L.lib.landing_debug_workspace.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ulonglong)]
    assert L.lib.landing_debug_workspace(L.ctx, C.byref(ptr), C.byref(st)) == 0
    n = st.value
    buf = torch.empty(n, dtype=torch.float64, device="cuda")
    hip = C.CDLL("libamdhip64.so"); hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(buf.data_ptr(), ptr.value, n * 8, 3) == 0
    torch.cuda.synchronize()
    return buf.cpu().numpy(), L
a, L = run(None); b, _ = run(sys.argv[1])
nx, ng, nj, nh = L.nx, L.ng, L.nnz_jac, L.nnz_hess
names = ["x", "xt", "dx", "gx"] + ["g", "gt", "s", "ds", "zL", "zU", "y", "yn", "lb", "ub", "sig", "rho"] + ["J", "H", "Hc", "ric", "cond"]
sizes = [nx] * 4 + [ng] * 12 + [nj, nh, N * 48, (N + 1) * 1200, N * 704]
off = 0
for nm, sz in zip(names, sizes):
    da, db = a[off:off + sz], b[off:off + sz]
    with np.errstate(all='ignore'):
        bad = ~(np.isclose(da, db, rtol=1e-9, atol=1e-12) | (np.isnan(da) & np.isnan(db)))
    print('%-5s size %6d  differing %6d  first %s' % (nm, sz, bad.sum(), np.nonzero(bad)[0][:6].tolist()))
    off += sz
print('total', off, len(a))
x0 = X0[0].copy(); o = L.param_offsets() if hasattr(L, 'param_offsets') else None
dxa = a[2 * nx:3 * nx]; i = np.argmax(np.abs(dxa[12:])) + 12
print('alpha a', (a[i] - x0[i]) / dxa[i], 'alpha b', (b[i] - x0[i]) / b[2 * nx + i])
