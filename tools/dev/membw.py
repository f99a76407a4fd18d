"""Development probe: achievable HBM write / copy bandwidth with plain torch kernels (ceiling for the store-dominated sweep)."""
import torch, json
n = 1 << 28   # 2 GiB of fp64
a = torch.empty(n, device="cuda", dtype=torch.float64); b = torch.empty(n, device="cuda", dtype=torch.float64)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms_fill = t(lambda: a.fill_(1.0)); ms_copy = t(lambda: b.copy_(a)); ms_read = t(lambda: a.sum())
print(json.dumps({"fill_GBps": n * 8 / ms_fill / 1e6, "copy_GBps(read+write)": 2 * n * 8 / ms_copy / 1e6, "sum_read_GBps": n * 8 / ms_read / 1e6}))
