// hip_emu.h -- TEST INFRASTRUCTURE: a minimal host emulation of the HIP constructs the kernels in
// landing-controller_amd/csrc use, so that their indexing / control logic can be exercised by the
// CPU test-suite (-m "not gpu") in a container without a GPU.  The product library is always
// built by hipcc for gfx950 from the very same sources; this header is never part of it.
//
// Model: one block at a time; the threads of a block are ucontext fibers scheduled round-robin,
// __syncthreads() yields to the next fiber (valid because the kernels only communicate through
// __shared__/global memory separated by __syncthreads()).  __shared__ maps to `static`.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__
#define __launch_bounds__(...)
#define __noinline__
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct int4 { int x, y, z, w; };
struct int2 { int x, y; };
inline int2 make_int2(int x, int y) { return int2{x, y}; }
struct emu_idx { unsigned x, y, z; };
extern emu_idx threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
#define hipHostMallocDefault 0
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void* p) { std::free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipEventDisableTiming 2
#define hipStreamNonBlocking 1
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)1; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)1; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline long long wall_clock64() { return 0; }
inline void sincos(double a, double* s, double* c) { *s = std::sin(a); *c = std::cos(a); }

namespace hip_emu {
void yield_barrier();        // block-wide barrier (generation counted)
void wave_barrier();         // barrier among the 64 fibers of the caller's wave
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
}
inline void __syncthreads() { hip_emu::yield_barrier(); }
// wave-level data exchange: only the 64 fibers of one wave have to call these convergently (as on the device)
inline double __shfl_xor(double v, int mask) {
  static double buf[1024];
  buf[threadIdx.x] = v;
  hip_emu::wave_barrier();
  const double r = buf[threadIdx.x ^ (unsigned)mask];
  hip_emu::wave_barrier();
  return r;
}
inline int __builtin_amdgcn_readlane(int v, int lane) {
  // double-buffered per wave: one barrier per call is enough (a slot is rewritten two calls later)
  static int buf[2][1024];
  static unsigned cnt[1024];                      // per-fiber call counter: all lanes of a wave make the same calls
  const unsigned p = cnt[threadIdx.x]++ & 1u;
  buf[p][threadIdx.x] = v;
  hip_emu::wave_barrier();
  return buf[p][(threadIdx.x & ~63u) + (unsigned)lane];
}
inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int v) {     // every lane reads lane (byte_addr / 4) & 63
  static int buf[2][1024];
  static unsigned cnt[1024];
  const unsigned p = cnt[threadIdx.x]++ & 1u;
  buf[p][threadIdx.x] = v;
  hip_emu::wave_barrier();
  return buf[p][(threadIdx.x & ~63u) + (((unsigned)byte_addr >> 2) & 63u)];
}
typedef double hip_emu_f64x4 __attribute__((vector_size(32)));
// v_mfma_f64_16x16x4_f64: lane l supplies A[l&15][l>>4], B[l>>4][l&15]; D[(l>>4)+4r][l&15] (all 64 lanes of the wave call it)
inline hip_emu_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hip_emu_f64x4 c, int, int, int) {
  static double bufA[2][1024], bufB[2][1024];
  static unsigned cnt[1024];
  const unsigned p = cnt[threadIdx.x]++ & 1u, base = threadIdx.x & ~63u, l = threadIdx.x & 63u;
  bufA[p][threadIdx.x] = a; bufB[p][threadIdx.x] = b;
  hip_emu::wave_barrier();
  const unsigned col = l & 15u;
  for (unsigned r = 0; r < 4; ++r) {
    const unsigned row = (l >> 4) + 4 * r;
    double acc = c[r];
    for (unsigned k = 0; k < 4; ++k) acc += bufA[p][base + k * 16 + row] * bufB[p][base + k * 16 + col];
    c[r] = acc;
  }
  return c;
}
typedef float hip_emu_f32x4 __attribute__((vector_size(16)));
// v_mfma_f32_16x16x4_f32: A / B as above (one float per lane), D[4 (l>>4) + r][l&15] -- NOT the f64 row map
inline hip_emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hip_emu_f32x4 c, int, int, int) {
  static float bufA[2][1024], bufB[2][1024];
  static unsigned cnt[1024];
  const unsigned p = cnt[threadIdx.x]++ & 1u, base = threadIdx.x & ~63u, l = threadIdx.x & 63u;
  bufA[p][threadIdx.x] = a; bufB[p][threadIdx.x] = b;
  hip_emu::wave_barrier();
  const unsigned col = l & 15u;
  for (unsigned r = 0; r < 4; ++r) {
    const unsigned row = 4 * (l >> 4) + r;
    float acc = c[r];
    for (unsigned k = 0; k < 4; ++k) acc = std::fmaf(bufA[p][base + k * 16 + row], bufB[p][base + k * 16 + col], acc);
    c[r] = acc;
  }
  return c;
}
inline int atomicAdd(int* p, int v) { const int o = *p; *p += v; return o; }      // (one block at a time, fibers never preempt)
inline int atomicCAS(int* p, int cmp, int v) { const int o = *p; if (o == cmp) *p = v; return o; }
inline int atomicMin(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_wave_barrier() { hip_emu::wave_barrier(); }
inline int __double2hiint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b >> 32); }
inline int __double2loint(double d) { long long b; std::memcpy(&b, &d, 8); return (int)(b & 0xffffffffLL); }
inline double __hiloint2double(int hi, int lo) { long long b = (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo); double d; std::memcpy(&d, &b, 8); return d; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hip_emu::run_grid(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
