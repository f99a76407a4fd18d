// Development probe 2: the sweep kernels' write pattern with the occupancy, the vector width and dependent arithmetic between write-outs as knobs.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/wbench2.hip -o tools/dev/wbench2 && tools/dev/wbench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
struct alignas(16) D2 { double a, b; };
// VEC 1: 16 lanes x 8 B per row (4 rows per instruction); VEC 2: 8 lanes x 16 B per row (8 rows per instruction)
template <int VEC>
__global__ void __launch_bounds__(64) wk(double* out, long long stride, int nrow, int seg, int nflush, int work, double seed, int mode) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  double* base = out + (long long)blockIdx.x * stride;
  const int ga = (int)((((unsigned long long)base) >> 3) & 15);
  double acc = seed + lane;
  for (int f = 0; f < nflush; ++f) {
    for (int i = 0; i < work; ++i) acc = fma(acc, 1.0000001, 1e-9);      // dependent chain: `work` x 4+ cycles
    if (VEC == 1) {
      const int c = lane & 15;
      for (int row = lane >> 4; row < nrow; row += 4) { const int s = row * seg, h = (16 - ((ga + s) & 15)) & 15; base[s + h + f * 16 + c] = acc; }
    } else {
      const int c = lane & 7;
      for (int row = lane >> 3; row < nrow; row += 8) { const int s = row * seg, h = (16 - ((ga + s) & 15)) & 15; *reinterpret_cast<D2*>(base + s + h + f * 16 + 2 * c) = D2{acc, acc}; }
    }
    if (mode == 1 && f == 1) {      // head [0, h) of every row, by words
      const int c = lane & 15;
      for (int row = lane >> 4; row < nrow; row += 4) { const int s = row * seg, h = (16 - ((ga + s) & 15)) & 15; if (c < h) base[s + c] = acc; }
    }
  }
  if (mode == 1) {      // tail [h + 16 nflush, seg) of every row, by words
    const int c = lane & 15;
    for (int row = lane >> 4; row < nrow; row += 4) { const int s = row * seg, h = (16 - ((ga + s) & 15)) & 15, p = h + 16 * nflush + c; if (p < seg) base[s + p] = acc; }
  }
  if (mode == 2) {      // the line shared by the tail of row r and the head of row r + 1: one full-line store
    const int c = lane & 15;
    for (int row = lane >> 4; row < nrow; row += 4) { const int s = row * seg, h = (16 - ((ga + s) & 15)) & 15; base[s + h + 16 * nflush + c] = acc; }
  }
  if (acc == 12345.678) lds[lane] = acc;
}
int main() {
  const int B = 4096, nrow = 40;
  double* d; hipMalloc(&d, (size_t)B * 16384 * 8 + 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Cfg { const char* name; int vec; long long stride; int seg, nflush, ldsKB, work, mode; } cfgs[] = {
    {"16B seg 385 occ 8/CU blocks only          ", 2, 15364, 385, 23, 20, 0, 0},
    {"16B seg 385 occ 8/CU + head, tail (words) ", 2, 15364, 385, 23, 20, 0, 1},
    {"16B seg 385 occ 8/CU + joint line at end  ", 2, 15364, 385, 23, 20, 0, 2},
    {"8B  seg 385 occ 8/CU blocks only          ", 1, 15364, 385, 23, 20, 0, 0},
    {"8B  seg 385 occ 8/CU + head, tail (words) ", 1, 15364, 385, 23, 20, 0, 1},
    {"8B  seg 385 occ 8/CU + joint line at end  ", 1, 15364, 385, 23, 20, 0, 2},
    {"8B  seg 157 occ 8/CU blocks only          ", 1, 15364, 157, 9, 20, 0, 0},
    {"8B  seg 157 occ 8/CU + head, tail (words) ", 1, 15364, 157, 9, 20, 0, 1},
    {"8B  seg 157 occ 8/CU + joint line at end  ", 1, 15364, 157, 9, 20, 0, 2},
    {"8B  seg 385 occ 8/CU work 30 blocks only  ", 1, 15364, 385, 23, 20, 30, 0},
    {"8B  seg 385 occ 8/CU work 30 head, tail   ", 1, 15364, 385, 23, 20, 30, 1},
    {"8B  seg 385 occ 8/CU work 30 joint line   ", 1, 15364, 385, 23, 20, 30, 2},
  };
  for (int B2 : {1024, 4096, 8192}) for (auto& c : cfgs) {
    if (B2 != 4096 && c.work != 30) continue;
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      const int nb = B2 > 4096 ? 4096 : B2;      // 8192: two launches back to back over the same buffer
      for (int l = 0; l < (B2 + 4095) / 4096; ++l) {
        if (c.vec == 1) hipLaunchKernelGGL(wk<1>, dim3(nb), dim3(64), c.ldsKB * 1024, 0, d, c.stride, nrow, c.seg, c.nflush, c.work, 1.0, c.mode);
        else hipLaunchKernelGGL(wk<2>, dim3(nb), dim3(64), c.ldsKB * 1024, 0, d, c.stride, nrow, c.seg, c.nflush, c.work, 1.0, c.mode);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)B2 * nrow * (c.mode ? c.seg : c.nflush * 16) * 8;
    printf("B %5d  %s : %7.1f us  %.2f TB/s  (%.0f MB)\n", B2, c.name, best * 1e3, bytes / best / 1e9, bytes / 1e6);
  }
  return 0;
}
