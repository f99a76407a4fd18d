// hip_emu.cpp -- TEST INFRASTRUCTURE (see hip_emu.h): fiber scheduler for emulated thread blocks.
#include "hip_emu.h"

emu_idx threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace hip_emu {
namespace {
struct Fiber { ucontext_t ctx; std::vector<char> stack; bool done = false; const unsigned* wait_ptr = nullptr; unsigned wait_val = 0; };
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

static void yield_once() { const int me = g_cur; swapcontext(&(*g_fibers)[me].ctx, &g_sched); }

// generation-counted barriers: a fiber that arrives re-yields until everybody of its group has arrived, so groups
// (block / wave) may synchronise independently of each other
static unsigned g_block_arrived = 0, g_block_gen = 0, g_block_size = 0;
static unsigned g_wave_arrived[64], g_wave_gen[64], g_wave_size[64];

static void wait_for_change(const unsigned* gen_ptr, unsigned gen) {
  Fiber& f = (*g_fibers)[g_cur];
  f.wait_ptr = gen_ptr; f.wait_val = gen;      // the scheduler skips this fiber until *gen_ptr changes
  yield_once();
}
void yield_barrier() {
  const unsigned gen = g_block_gen;
  if (++g_block_arrived == g_block_size) { g_block_arrived = 0; ++g_block_gen; return; }
  wait_for_change(&g_block_gen, gen);
}
void wave_barrier() {
  const unsigned w = (unsigned)g_cur >> 6;
  const unsigned gen = g_wave_gen[w];
  if (++g_wave_arrived[w] == g_wave_size[w]) { g_wave_arrived[w] = 0; ++g_wave_gen[w]; return; }
  wait_for_change(&g_wave_gen[w], gen);
}

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
      f.done = false; f.wait_ptr = nullptr;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack.data();
      f.ctx.uc_stack.ss_size = STACK;
      f.ctx.uc_link = &g_sched;
      makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    g_block_arrived = 0; g_block_gen = 0; g_block_size = nthreads;
    for (unsigned w = 0; w < 64; ++w) { g_wave_arrived[w] = 0; g_wave_gen[w] = 0; const unsigned lo = w * 64; g_wave_size[w] = nthreads > lo ? (nthreads - lo < 64 ? nthreads - lo : 64) : 0; }
    bool any = true;
    while (any) {
      any = false;
      bool ran = false;
      for (unsigned t = 0; t < nthreads; ++t) {
        if (fibers[t].done) continue;
        any = true;
        if (fibers[t].wait_ptr) { if (*fibers[t].wait_ptr == fibers[t].wait_val) continue; fibers[t].wait_ptr = nullptr; }
        ran = true;
        g_cur = (int)t;
        blockIdx = {bx, by, bz};
        threadIdx = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        swapcontext(&g_sched, &fibers[t].ctx);
      }
      if (any && !ran) {   // every live fiber waits at a barrier that can no longer complete: a divergent __syncthreads()
        unsigned waiting = 0, finished = 0;     // (some threads of the block left, or wait at another barrier) -- on the GPU this hangs
        for (unsigned t = 0; t < nthreads; ++t) { if (fibers[t].done) ++finished; else ++waiting; }
        std::fprintf(stderr, "hip_emu: barrier deadlock in block (%u,%u,%u): %u threads wait, %u have exited, block barrier %u/%u arrived\n",
                     bx, by, bz, waiting, finished, g_block_arrived, g_block_size);
        std::abort();
      }
    }
  }
  g_fibers = nullptr; g_body = nullptr;
}
}  // namespace hip_emu
