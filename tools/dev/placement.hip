// Development probe: where do the waves of a 256-thread / 38 KB-LDS workgroup land (XCD, CU, SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 4) void probe(unsigned* out, int spin) {
  __shared__ double pad[4800];
  const int w = threadIdx.x >> 6;
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID, 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
  pad[threadIdx.x] = hw;
  __syncthreads();
  double a = pad[(threadIdx.x * 7) & 255];
  for (int i = 0; i < spin; ++i) a = a * 1.0000001 + 1e-9;   // keep the block resident for a while
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + w) * 2] = hw; out[(blockIdx.x * 4 + w) * 2 + 1] = xcc + (a == 12345.0); }
}
int main() {
  const int B = 1024;
  unsigned* d; hipMalloc(&d, B * 8 * sizeof(unsigned));
  probe<<<B, 256>>>(d, 200000);
  std::vector<unsigned> h(B * 8); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<int>> cu;
  int simd_hist[4][4] = {};
  for (int b = 0; b < B; ++b) {
    for (int w = 0; w < 4; ++w) { unsigned hw = h[(b * 4 + w) * 2]; simd_hist[w][(hw >> 4) & 3]++; }
    unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
    unsigned key = (xcc << 16) | (hw & 0xff00);
    cu[key].push_back(b);
  }
  printf("distinct CUs: %zu\n", cu.size());
  for (int w = 0; w < 4; ++w) printf("wave %d simd histogram: %d %d %d %d\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
  int shown = 0;
  for (auto& kv : cu) { if (shown++ >= 12) break; printf("cu %05x:", kv.first); for (int b : kv.second) printf(" b%d(simd0=%u)", b, (h[b * 8] >> 4) & 3); printf("\n"); }
  std::map<size_t, int> occ; for (auto& kv : cu) occ[kv.second.size()]++;
  for (auto& o : occ) printf("CUs with %zu blocks: %d\n", o.first, o.second);
  return 0;
}
