// Development probe: write bandwidth of the sweep kernels' store pattern.  One wave per "member"; every store instruction writes four runs of
// 16 doubles (lane c = lane & 15 of rows lane >> 4 ...), the runs of one wave sit SEG doubles apart (the CCS segment length), starting at
// offset OFF doubles from a 128-byte boundary.  Compares aligned (SEG % 16 == 0, OFF = 0) with the real layout (SEG = 157, any OFF) and
// with full-line rows of 32.     hipcc --offload-arch=gfx950 -O3 tools/dev/wbench.hip -o tools/dev/wbench && tools/dev/wbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int W>      // run length in doubles (16 or 32)
__global__ void __launch_bounds__(64) wk(double* out, long long per_member, int nrow, int seg, int off, int nflush) {
  const int lane = threadIdx.x, c = lane % W, r0 = lane / W;
  constexpr int RPI = 64 / W;      // rows per store instruction
  double* base = out + (long long)blockIdx.x * per_member + off;
  for (int f = 0; f < nflush; ++f)
    for (int row = r0; row < nrow; row += RPI) base[(long long)row * seg + f * W + c] = (double)(f + row);
}
int main() {
  const int B = 4096, nrow = 40;
  const long long per_member = 16384;      // doubles per member (128 KB)
  double* d; hipMalloc(&d, (size_t)B * per_member * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { const char* name; int W, seg, off, nflush; } cfgs[] = {
    {"aligned  seg 160 off 0  W16", 16, 160, 0, 10}, {"real     seg 157 off 0  W16", 16, 157, 0, 9}, {"real     seg 157 off 5  W16", 16, 157, 5, 9},
    {"aligned  seg 384 off 0  W16", 16, 384, 0, 24}, {"real     seg 385 off 3  W16", 16, 385, 3, 24},
    {"aligned  seg 384 off 0  W32", 32, 384, 0, 12}, {"real     seg 385 off 3  W32", 32, 385, 3, 12}};
  for (auto& c : cfgs) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (c.W == 16) hipLaunchKernelGGL(wk<16>, dim3(B), dim3(64), 0, 0, d, per_member, nrow, c.seg, c.off, c.nflush);
      else hipLaunchKernelGGL(wk<32>, dim3(B), dim3(64), 0, 0, d, per_member, nrow, c.seg, c.off, c.nflush);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)B * nrow * c.nflush * c.W * 8;
    printf("%s : %.1f us  %.2f TB/s  (%.0f MB)\n", c.name, best * 1e3, bytes / best / 1e9, bytes / 1e6);
  }
  return 0;
}
