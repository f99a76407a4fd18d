// Development probe: single-wave issue cost (shader cycles per instruction) of the instructions the pivot chain uses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((vector_size(32)));
#define REP 64
template <int MODE>
__global__ void k(long long* out, double* sink, int lanesel) {
  __shared__ double lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.0 + threadIdx.x * 1e-3 + i;
  double x = 1.0000001 + threadIdx.x * 1e-9;
  f64x4 acc = {0, 0, 0, 0};
  int idx = (threadIdx.x * 8) & 1023;
  long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
    if (MODE == 0) {          // 16 independent DFMA
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = fma(a[i], x, x);
    } else if (MODE == 1) {   // 16 x (2 readlane -> SGPR) + DFMA using it
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int hi = __builtin_amdgcn_readlane(__double2hiint(a[i]), lanesel), lo = __builtin_amdgcn_readlane(__double2loint(a[i]), lanesel);
        a[i] = fma(__hiloint2double(hi, lo), x, a[i]);
      }
    } else if (MODE == 2) {   // 32 readlanes only (xor-folded into one scalar)
      int s = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s ^= __builtin_amdgcn_readlane(__double2hiint(a[i]), lanesel); s ^= __builtin_amdgcn_readlane(__double2loint(a[i]), lanesel); }
      a[0] += s;
    } else if (MODE == 3) {   // dependent DFMA chain (latency)
#pragma unroll
      for (int i = 0; i < 16; ++i) x = fma(x, x, 1e-9);
    } else if (MODE == 4) {   // MFMA f64 16x16x4, dependent accumulator
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], x, acc, 0, 0, 0);
    } else if (MODE == 5) {   // 4 independent MFMA accumulators
      f64x4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
#pragma unroll
      for (int i = 0; i < 4; ++i) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], x, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i + 4], x, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i + 8], x, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i + 12], x, c3, 0, 0, 0); }
      acc = c0 + c1 + c2 + c3;
    } else if (MODE == 6) {   // dependent LDS read chain (latency): 16 hops
#pragma unroll
      for (int i = 0; i < 16; ++i) idx = ((int)lds[idx] * 8 + 8) & 1023;
    } else if (MODE == 7) {   // 16 rcp f64
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_rcp(a[i]);
    } else if (MODE == 8) {   // 32 bpermute
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int hi = __builtin_amdgcn_ds_bpermute(lanesel << 2, __double2hiint(a[i])), lo = __builtin_amdgcn_ds_bpermute(lanesel << 2, __double2loint(a[i]));
        a[i] = __hiloint2double(hi, lo) + 1.0;
      }
    } else if (MODE == 10) {  // the 4x4 LDL^T + solve of one block step, inputs perturbed by the previous result (dependent)
      const double a00 = a[0] + x * 1e-30, a10 = a[1] * 1e-3, a11 = a[2], a20 = a[3] * 1e-3, a21 = a[4] * 1e-3, a22 = a[5], a30 = a[6] * 1e-3, a31 = a[7] * 1e-3, a32 = a[8] * 1e-3, a33 = a[9];
      const double w0 = a[10], w1 = a[11], w2 = a[12], w3 = a[13];
      auto recip = [](double d) { double i = __builtin_amdgcn_rcp(d); i = fma(i, fma(-d, i, 1.0), i); return fma(i, fma(-d, i, 1.0), i); };
      const double d0 = a00, i0 = recip(d0);
      const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
      const double d1 = fma(-l10, a10, a11), i1 = recip(d1);
      const double t21 = fma(-l20, a10, a21), t31 = fma(-l30, a10, a31);
      const double l21 = t21 * i1, l31 = t31 * i1;
      const double d2 = fma(-l21, t21, fma(-l20, a20, a22)), i2 = recip(d2);
      const double t32 = fma(-l31, t21, fma(-l30, a20, a32));
      const double l32 = t32 * i2;
      const double d3 = fma(-l32, t32, fma(-l31, t31, fma(-l30, a30, a33))), i3 = recip(d3);
      const double y1 = fma(-l10, w0, w1);
      const double y2 = fma(-l21, y1, fma(-l20, w0, w2));
      const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, w0, w3)));
      const double r3 = y3 * i3;
      const double r2 = fma(-l32, r3, y2 * i2);
      const double r1 = fma(-l31, r3, fma(-l21, r2, y1 * i1));
      const double r0 = fma(-l30, r3, fma(-l20, r2, fma(-l10, r1, w0 * i0)));
      const int lk = threadIdx.x >> 4;
      x = lk == 0 ? r0 : (lk == 1 ? r1 : (lk == 2 ? r2 : r3));
    } else if (MODE == 11) {  // LDS exchange round trip: write, barrier, uniform + per-lane reads, dependent
      lds[threadIdx.x * 4 + (r & 3)] = x;
      __syncthreads();
      x = lds[(r * 16) & 1023] + lds[((r * 16) & 1023) + 4] + lds[(threadIdx.x * 4 + 1) & 1023];
    } else if (MODE == 12) {  // barrier only
      __syncthreads();
    } else if (MODE == 9) {   // 16 independent LDS reads b64 + add
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] += lds[(idx + i * 64) & 1023];
    }
  }
  long long t1 = clock64();
  double s = x + idx;
  for (int i = 0; i < 16; ++i) s += a[i];
  s += acc[0] + acc[1] + acc[2] + acc[3];
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[MODE] = t1 - t0;
}
int main() {
  long long* d; double* s; hipMalloc(&d, 16 * 8); hipMalloc(&s, 1024 * 8); hipMemset(d, 0, 128);
#define RUN(M) k<M><<<1, 64>>>(d, s, 3); k<M><<<1, 64>>>(d, s, 3);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
  long long h[16]; hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
  const char* nm[] = {"16 indep DFMA", "16 x (2 readlane + DFMA)", "32 readlane", "16 dependent DFMA", "16 dependent MFMA f64 16x16x4", "16 MFMA (4 indep accumulators)",
                      "16 dependent LDS hops (+cvt)", "16 rcp f64", "32 bpermute (+16 add)", "16 indep LDS read b64 + add", "LDL4+solve (x16 for per-iteration)", "LDS write+barrier+3 reads (x16)", "barrier (x16)"};
  for (int m = 0; m < 13; ++m) printf("%-36s %8.1f cycles per group of 16  (%lld total)\n", nm[m], (double)h[m] / REP, h[m]);
  return 0;
}
