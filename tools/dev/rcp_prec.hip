// Development probe: relative error of v_rcp_f64 and of one / two Newton refinements against 1/x.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  double d = x[i], a = __builtin_amdgcn_rcp(d);
  r0[i] = a; a = fma(a, fma(-d, a, 1.0), a); r1[i] = a; a = fma(a, fma(-d, a, 1.0), a); r2[i] = a;
}
int main() {
  const int n = 1 << 22; std::vector<double> h(n); std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-40, 40), m(1, 2);
  for (auto& v : h) v = std::ldexp(m(g), (int)u(g));
  double *x, *r0, *r1, *r2; hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8);
  hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(x, r0, r1, r2, n);
  std::vector<double> a(n), b(n), c(n); hipMemcpy(a.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), r1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), r2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) { long double t = 1.0L / (long double)h[i]; e0 = fmax(e0, (double)fabsl((a[i] - t) / t)); e1 = fmax(e1, (double)fabsl((b[i] - t) / t)); e2 = fmax(e2, (double)fabsl((c[i] - t) / t)); }
  printf("max rel err: rcp %.3e  +1 Newton %.3e  +2 Newton %.3e  (eps = %.3e)\n", e0, e1, e2, 2.22e-16);
  return 0;
}
