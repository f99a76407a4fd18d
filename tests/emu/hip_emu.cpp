// hip_emu.cpp -- TEST INFRASTRUCTURE (see hip_emu.h): fiber scheduler for emulated thread blocks.
#include "hip_emu.h"

emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace hip_emu {
namespace {
struct Fiber { ucontext_t ctx; std::vector<char> stack; bool done = false; };
std::vector<Fiber>* g_fibers = nullptr;
ucontext_t g_sched;
int g_cur = -1;
const std::function<void()>* g_body = nullptr;
void trampoline() {
  (*g_body)();
  (*g_fibers)[g_cur].done = true;
  swapcontext(&(*g_fibers)[g_cur].ctx, &g_sched);
}
}  // namespace

void yield_barrier() { swapcontext(&(*g_fibers)[g_cur].ctx, &g_sched); }

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
  const size_t STACK = 1u << 20;
  const unsigned nthreads = block.x * block.y * block.z;
  std::vector<Fiber> fibers(nthreads);
  for (auto& f : fibers) f.stack.resize(STACK);
  gridDim = grid; blockDim = block;
  g_fibers = &fibers; g_body = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = fibers[t];
      f.done = false;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack.data();
      f.ctx.uc_stack.ss_size = STACK;
      f.ctx.uc_link = &g_sched;
      makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    bool any = true;
    while (any) {
      any = false;
      for (unsigned t = 0; t < nthreads; ++t) {
        if (fibers[t].done) continue;
        any = true;
        g_cur = (int)t;
        blockIdx = {bx, by, bz};
        threadIdx = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        swapcontext(&g_sched, &fibers[t].ctx);
      }
    }
  }
  g_fibers = nullptr; g_body = nullptr;
}
}  // namespace hip_emu
