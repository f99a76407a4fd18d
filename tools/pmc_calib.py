"""Calibration workload for the FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md: 'calibrate on a known byte count in
your own access pattern'): known-byte kernels with the solver's access widths.
  * calib_stream8: y[i] = 2 x[i], fp64, 8 bytes per lane, coalesced           -> reads 1 GiB, writes 1 GiB per launch
  * calib_gather8: y[i] = x[perm[i]], fp64 gather through a random permutation within 4 KB windows (like the term lists)
Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; tools/pmc_summary.py reads the ratios."""
import torch
n = 1 << 27
x = torch.randn(n, device="cuda", dtype=torch.float64)
y = torch.empty_like(x)
w = 512                                           # 4 KB windows
perm = (torch.arange(n, device="cuda").view(-1, w) // w * w + torch.stack([torch.randperm(w, device="cuda") for _ in range(64)]).repeat(n // w // 64, 1)).view(-1)
torch.cuda.synchronize()
for _ in range(3):
    torch.mul(x, 2.0, out=y)                      # elementwise kernel: 8 B / lane
torch.cuda.synchronize()
for _ in range(3):
    torch.index_select(x, 0, perm, out=y)         # gather kernel
torch.cuda.synchronize()
print("calib done", n * 8, "bytes per array")
