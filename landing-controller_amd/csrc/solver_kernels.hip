// solver_kernels.hip -- batched interior-point solver kernels (placeholder until the solver lands).
#include <hip/hip_runtime.h>
namespace landing {
struct SolverWorkspace {
  void* buf = nullptr; size_t bytes = 0;
  void release() { if (buf) hipFree(buf); buf = nullptr; bytes = 0; }
};
}  // namespace landing
