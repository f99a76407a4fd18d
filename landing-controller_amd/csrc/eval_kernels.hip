// eval_kernels.hip -- batched function layer of the SRBM landing NLP on gfx950.
//
// One workgroup (64 threads = one wavefront) per batch member; lane = shooting stage (stages beyond
// 64 wrap).  Each lane evaluates its stage with srbm_stage.hpp and produces the contiguous CCS
// segments of the reference's patterns (layout.hpp), i.e. the batched form of
//   nlp_f :10995, nlp_g :11161, nlp_grad_f :52602, nlp_jac_g :94014, nlp_hess_l :53527, nlp_grad :22015
// of optimizations/landing/codegen_casadi/landingCtrller_IPOPT.c.
#include <hip/hip_runtime.h>
#include <math.h>

#include "layout.hpp"
#include "srbm_stage.hpp"

namespace landing {

struct EvalArgs {
  const double* x; const double* p; const double* lam_f; const double* lam_g;
  double* f; double* g; double* grad_f; double* jac; double* hess; double* ggx; double* ggp;
  const int* edge_map;
  int g_staged;          // 1: the stage rows of g are written by landing_sweep_kernel<2> (misc kernel writes the 36 boundary rows only)   // [0..227] U-part Jacobian positions of stage 0, [228..455] of stage N-1 (-1 = entry absent)
};

__host__ __device__ __forceinline__ int dyn_row_of_state(int i) {  // state index -> dynamics row (pos,rpy,v,omega)
  return i < 6 ? i : (i < 9 ? i + 3 : i - 3);
}

// Weights and force reference of the running cost: constants of the context (run_cost 1) or entries of p (run_cost 2: the N=41
// script's own parameter vector, layout_ccc_params).  Value accessors (no pointers into the by-value Layout kernel argument).
__device__ __forceinline__ double rc_QX(const Layout& L, const double* p, int i) { return L.run_cost == 2 ? p[L.o_QX + i] : L.QX[i]; }
__device__ __forceinline__ double rc_Qc(const Layout& L, const double* p, int a) { return L.run_cost == 2 ? p[L.o_Qc + a] : L.Qc[a]; }
__device__ __forceinline__ double rc_Qf(const Layout& L, const double* p, int a) { return L.run_cost == 2 ? p[L.o_Qf + a] : L.Qf[a]; }
__device__ __forceinline__ double rc_fref(const Layout& L, const double* p, int k, int l, int a) { return L.run_cost == 2 ? p[L.o_Uref + 24 * k + 12 + 3 * l + a] : L.f_ref[a]; }

// Running cost of stage k (generate_quadruped_SRBM_CCC.m:81-89) and, optionally, its gradient added to gX / gc / gf.
__device__ __forceinline__ double run_cost_stage(const Layout& L, const double* x, const double* p, int k, double* gX, double* gc, double* gf) {
  const double* X = x + L.x_X(k); const double* U = x + L.x_U(k);
  const double dt = p[L.o_dt + k];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const double e = X[i] - p[12 * k + i], q = rc_QX(L, p, i);
    s += q * e * e;
    if (gX) gX[i] += 2.0 * dt * q * e;
  }
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double r = X[a] + L.p_hip[3 * l + a] - U[3 * l + a], u = U[12 + 3 * l + a] - rc_fref(L, p, k, l, a);
      const double qc = rc_Qc(L, p, a), qf = rc_Qf(L, p, a);
      s += qc * r * r + qf * u * u;
      if (gX) gX[a] += 2.0 * dt * qc * r;
      if (gc) gc[3 * l + a] -= 2.0 * dt * qc * r;
      if (gf) gf[3 * l + a] += 2.0 * dt * qf * u;
    }
  return dt * s;
}

__device__ __forceinline__ void load_stage(const Layout& L, const double* x, const double* p, int k,
                                           srbm::StageVars& z, srbm::StageParams& P) {
  const double* Xk = x + L.x_X(k);
  const double* Uk = x + L.x_U(k);
  const double* Xn = x + L.x_X(k + 1);
  for (int i = 0; i < 12; ++i) { z.X[i] = Xk[i]; z.c[i] = Uk[i]; z.f[i] = Uk[12 + i]; z.Xn[i] = Xn[i]; }
  if (k < L.N - 1) { const double* Un = x + L.x_U(k + 1); for (int i = 0; i < 12; ++i) z.cn[i] = Un[i]; }
  else { for (int i = 0; i < 12; ++i) z.cn[i] = 0.0; }
  P.dt = p[L.o_dt + k];
  P.mass = p[L.o_mass];
  P.dt_over_m = P.dt / P.mass;
  P.km = 0.71 * p[L.o_mu];
  for (int i = 0; i < 3; ++i) { P.Ib[i] = p[L.o_Ib + i]; P.Ibi[i] = p[L.o_Ib_inv + i]; }
  P.kin_z_off = L.kin_z_off;
}

struct RowStore { double* g; __device__ __forceinline__ void put(int r, double v) { g[r] = v; } };
#ifdef LANDING_DEV_NOSTORE      // development probe (timing only, garbage results): the values are summed and ONE value per segment is stored -- what do the lane-strided stores of the solver's derivative tasks cost?
struct SeqStoreJ { double* q; double acc = 0.0; __device__ __forceinline__ void col() {} __device__ __forceinline__ void end() { *q = acc; } __device__ __forceinline__ void put(int, double v) { acc += v; } };
struct SeqStoreH { double* q; double acc = 0.0; __device__ __forceinline__ void end() { *q = acc; } __device__ __forceinline__ void put(double v) { acc += v; } };
#else
struct SeqStoreJ { double* q; __device__ __forceinline__ void col() {} __device__ __forceinline__ void end() {} __device__ __forceinline__ void put(int, double v) { *q++ = v; } };
struct SeqStoreH { double* q; __device__ __forceinline__ void end() {} __device__ __forceinline__ void put(double v) { *q++ = v; } };
#endif
struct LamStage { const double* l; __device__ __forceinline__ double operator()(int r) const { return l[r]; } };
// the multipliers of a stage addressed by the row numbers of a MIDDLE stage (104 rows); `last`: the lane holds the last stage (80 rows: the six
// no-slip rows of every foot are absent -> 0, the rows behind them move up).  r is a compile-time constant at every call site.
struct LamRemap {
  const double* l; bool last;
  __device__ __forceinline__ double operator()(int r) const {
    const double xm = l[r];
    if (r < 16) return xm;
    const int t = (r - 16) % 12, leg = (r - 16) / 12;
    if (t >= 2 && t < 8) return last ? 0.0 : xm;
    const double xl = l[t < 2 ? 16 + 6 * leg + t : 16 + 6 * leg + t - 6];
    return last ? xl : xm;
  }
};

// column-wise accumulation of J^T lam (grad_gamma_x)
struct DotLam {
  const double* lam_own; const double* lam_prev; bool first; double* out; double acc; bool open;
  __device__ __forceinline__ void col() { if (open) *out++ = acc; acc = 0.0; open = true; }
  __device__ __forceinline__ void end() {}
  __device__ __forceinline__ void put(int r, double v) {
    double l;
    if (r >= 0) l = lam_own[r];
    else if (r > -100) { const int i = -1 - r; l = first ? lam_prev[i] : lam_prev[dyn_row_of_state(i)]; }
    else { const int c = -100 - r; const int leg = c / 6, rem = c % 6; l = lam_prev[16 + 12 * leg + 2 + rem]; }
    acc += v * l;
  }
  __device__ __forceinline__ void finish() { if (open) *out++ = acc; open = false; }
};

// Sequential store of one CCS segment per lane, coalesced through an LDS tile: every lane of the wavefront appends its value to its own
// tile row; every 16 values the wave writes 16-value runs out row by row, so that one store instruction covers four 128-byte runs instead of 64
// scattered 8-byte words (the direct per-lane stores are bound by the L2 request rate, not by bytes: profiles/README.md).  All lanes emit the
// SAME sequence -- stages 0 and N-1, whose segments lack the entries of the neighbouring stage's no-slip rows, emit placeholders there and the
// write-out compacts them through `map` (position in the uniform sequence -> position in the segment, -1 = absent).
//
// Round 4: the runs are ALIGNED to 128-byte lines and written 16 bytes per lane.  The CCS segments start wherever the reference's nonzero order
// puts them (157 k, 228 k + ... doubles, members 15 364 doubles apart), so a run of the 16 values emitted last straddles two lines; measured with the
// store pattern alone (tools/dev/wbench.hip, 4096 waves x 40 rows): 3.4 TB/s against 6.7 TB/s for line-aligned runs.  And the write-out of round 3
// (one 8-byte word per lane and row visit, ~30 instructions each) was 2.5 x the arithmetic of the Jacobian stream in issued instructions.  A tile
// row is therefore a RING of the last 32 values (slot 32 mirrors slot 0, so that a pair of consecutive positions is always contiguous): from the
// second round on a row writes the aligned block that became complete one round earlier, positions [h + 16 (F - 2), h + 16 (F - 1)) with h = 1..16
// the distance of the segment start to the next line boundary -- eight lanes per row, one global_store_dwordx4 each (tile_round).  The head [0, h)
// and the tail (<= 30 values) go out at the end of the part by 8-byte words (tile_rest: in full lines too).  The rows of the edge
// stages, whose positions go through the compaction maps, are written by words in the emitted windows as before.  X_k and U_k columns of a stage
// are emitted one after the other (srbm_stage.hpp), so both parts share ONE tile: the second emitter takes the tile over at end() of the first.
constexpr int TILE_LD = 49;      // ring of 32 + mirror of slot 0 + the first 16 values of the part (TILE_HEAD)
constexpr int TILE_HEAD = 33;
// hides a wave-uniform value from the optimiser (empty asm on an SGPR); a no-op for the g++ host emulation of tests/emu
#if defined(__HIP__)
#define LANDING_OPAQUE_UNIFORM(x) asm volatile("" : "+s"(x))
#define LANDING_OPAQUE_LANE(x) asm volatile("" : "+v"(x))
#else
#define LANDING_OPAQUE_UNIFORM(x) ((void)0)
#define LANDING_OPAQUE_LANE(x) ((void)0)
#endif
// tile rows (per lane) whose LDS reads are issued together in the block write-out: a lane visits rows lane>>3, +8, ... -- 5 visits at N = 40
#ifndef LANDING_FLUSH_GROUP_J
#define LANDING_FLUSH_GROUP_J 3
#endif
#ifndef LANDING_FLUSH_GROUP_H
#define LANDING_FLUSH_GROUP_H 5
#endif
#ifndef LANDING_FLUSH_GROUP_G
#define LANDING_FLUSH_GROUP_G 5
#endif
// KIND 0: Jacobian X_k columns, 1: Jacobian U_k columns, 2: Hessian X_k columns, 3: Hessian U_k columns, 4: g rows.
// start of stage k's segment in the member's array, and the position of emitted value `pos` inside it (-1: a placeholder of an edge stage)
template <int KIND>
__device__ __forceinline__ int tile_seg(const Layout* L, int k) {
  return KIND == 0 ? L->jx(k) : (KIND == 1 ? L->ju(k) : (KIND == 2 ? L->hx(k) : (KIND == 3 ? L->hu(k) : L->g_stage(k))));
}
template <int KIND> constexpr bool tile_first_edge() { return KIND == 1 || KIND == 3; }      // stage 0 is compacted
template <int KIND> constexpr bool tile_last_edge() { return KIND == 1 || KIND == 4; }       // stage N-1 is compacted
template <int KIND>
__device__ __forceinline__ bool tile_edge(int k, int N) {      // rows whose positions are compacted: written by words, window by window
  return (tile_first_edge<KIND>() && k == 0) || (tile_last_edge<KIND>() && k == N - 1);
}
template <int KIND>
__device__ __forceinline__ int tile_pos(const int* map, int k, int N, int pos) {
  if (KIND == 1) { if (k == 0) pos = map[pos]; else if (k == N - 1) pos = map[228 + pos]; }
  else if (KIND == 3) { if (k == 0) pos = (pos < 72) ? ((pos % 6 == 4) ? -1 : pos - (pos / 6) - (pos % 6 > 4 ? 1 : 0)) : pos - 12; }
  else if (KIND == 4) {   // residual rows; the last stage has no no-slip rows (80 instead of 104 rows)
    if (k == N - 1) {
      if (pos >= 64) pos -= 24;
      else if (pos >= 16) { const int l = (pos - 16) / 12, t = (pos - 16) % 12; pos = t < 2 ? 16 + 6 * l + t : (t < 8 ? -1 : 16 + 6 * l + t - 6); }
    }
  }
  return pos;
}
struct alignas(16) TilePairOfDoubles { double a, b; };
// Regular round (cnt = 16 F): the line-aligned block [h + 16 (F - 2), + 16) of every interior row, two values per lane (F >= 2; at F = 2 also
// the head [0, h), by words), and the window emitted since the last round of the edge rows (by words; lanes 0-15 stage 0, lanes 16-31 stage N-1).
// Inlined (an out-of-line call makes every write-out wait for its stores at the return), but the position counter is laundered through an
// empty asm so that the call sites of a stage are not specialised and hoisted into one giant live range (that version spilled 1.9 KB per lane);
// ONE function for the whole round: the stage functions' loops are only unrolled (and the round tests resolved) while their body stays small.
// ga = (address of gbase / 8) mod 16: with it the distance of a segment start to the next 128-byte line is 32-bit arithmetic.
template <int KIND>
__device__ __forceinline__ void tile_round(const double* tile, double* gbase, int ga, const Layout* L, const int* map, int k0, int nrow, int cnt_) {
  int cnt = cnt_;
  LANDING_OPAQUE_UNIFORM(cnt);
  __builtin_amdgcn_wave_barrier();
  // the lane number is laundered too: everything derived from it is recomputed in every round (a few integer operations) instead of living in
  // registers across the stage's arithmetic -- at the 256-register bound those were SPILLED, and their reload at the start of a round
  // (s_waitcnt vmcnt(0)) also waited for every store of the previous round: the rounds were serialised on the store latency
  int lane = threadIdx.x & 63;
  LANDING_OPAQUE_LANE(lane);
  const int N = L->N;
  constexpr int FG = KIND <= 1 ? LANDING_FLUSH_GROUP_J : (KIND <= 3 ? LANDING_FLUSH_GROUP_H : LANDING_FLUSH_GROUP_G);
  if (cnt >= 32) {
    const int lag = cnt - 32 + 2 * (lane & 7);
#pragma unroll 1
    for (int row0 = lane >> 3; row0 < nrow; row0 += 8 * FG) {
      double va[FG], vb[FG]; int q[FG];
#pragma unroll
      for (int j = 0; j < FG; ++j) {
        const bool in = row0 + 8 * j < nrow;
        const int row = in ? row0 + 8 * j : row0;
        const int k = k0 + row;
        const int seg = tile_seg<KIND>(L, k);
        const int p = 16 - ((ga + seg) & 15) + lag;
        const double* tr = tile + row * TILE_LD;
        va[j] = tr[p & 31]; vb[j] = tr[(p & 31) + 1];
        q[j] = in && !tile_edge<KIND>(k, N) ? seg + p : -1;
      }
#pragma unroll
      for (int j = 0; j < FG; ++j)
        if (q[j] >= 0) *reinterpret_cast<TilePairOfDoubles*>(gbase + q[j]) = TilePairOfDoubles{va[j], vb[j]};
    }
  }
  if (tile_first_edge<KIND>() || tile_last_edge<KIND>()) {
    const int k = (lane >> 4) == 0 ? (tile_first_edge<KIND>() ? 0 : -1) : ((lane >> 4) == 1 && tile_last_edge<KIND>() ? N - 1 : -1);
    const int row = k - k0;
    if (k >= 0 && row >= 0 && row < nrow) {
      const int p = cnt - 16 + (lane & 15), d = tile_pos<KIND>(map, k, N, p);
      if (d >= 0) gbase[tile_seg<KIND>(L, k) + d] = tile[row * TILE_LD + (p & 31)];
    }
  }
  __builtin_amdgcn_wave_barrier();
}
// The rest of a part, by 8-byte words, 16 lanes per row: the tail of every row (<= 30 values of an interior row, <= 15 of an edge row) and the
// heads.  A segment ends where the next stage's begins, so the tail of row r and the head [0, h) of row r + 1 share a 128-byte line: the lanes
// past the end of row r write the head of row r + 1 (kept in the TILE_HEAD slots) IN THE SAME STORE -- a full line instead of two partial
// ones at different times (the store pattern alone: 4.2 -> 5.2 TB/s, tools/dev/wbench2.hip).  A row whose predecessor cannot do that (first row
// of the tile, or the predecessor is an edge row) writes its own head.
template <int KIND>
__device__ __forceinline__ void tile_rest(const double* tile, double* gbase, int ga, const Layout* L, const int* map, int k0, int nrow, int cnt_) {
  int cnt = cnt_;
  LANDING_OPAQUE_UNIFORM(cnt);
  __builtin_amdgcn_wave_barrier();
  int lane = threadIdx.x & 63;
  LANDING_OPAQUE_LANE(lane);
  const int c = lane & 15, N = L->N;
  constexpr int FG = 2;
  const int F = cnt >> 4;
#pragma unroll 1
  for (int row0 = lane >> 4; row0 < nrow; row0 += 4 * FG) {
    double v[FG], v2[FG], v3[FG]; int q[FG], q2[FG], q3[FG];
#pragma unroll
    for (int j = 0; j < FG; ++j) {
      const bool in = row0 + 4 * j < nrow;
      const int row = in ? row0 + 4 * j : row0;
      const int k = k0 + row;
      const int seg = tile_seg<KIND>(L, k);
      const int h = 16 - ((ga + seg) & 15);      // 1..16
      const double* tr = tile + row * TILE_LD;
      int p1, p2 = -1, p3 = -1;
      bool n1 = false, n2 = false;      // the position belongs to the head of the next row
      if (tile_edge<KIND>(k, N)) p1 = c < (cnt & 15) ? (cnt & ~15) + c : -1;
      else if (F < 2) { p1 = c; p2 = 16 + c; if (p1 >= cnt) p1 = -1; if (p2 >= cnt) p2 = -1; }      // short part: nothing written yet, the row writes all of itself
      else {
        p1 = h + 16 * (F - 1) + c; p2 = p1 + 16;
        // the next row's head: allowed when that row is in the tile, not an edge row and really contiguous
        const bool join = row + 1 < nrow && !tile_edge<KIND>(k + 1, N) && tile_seg<KIND>(L, k + 1) == seg + cnt;
        if (p1 >= cnt) { n1 = join && p1 - cnt < 16 - ((ga + seg + cnt) & 15); if (!n1) p1 = -1; }
        if (p2 >= cnt) { n2 = join && p2 - cnt < 16 - ((ga + seg + cnt) & 15); if (!n2) p2 = -1; }
        if ((row == 0 || tile_edge<KIND>(k - 1, N) || tile_seg<KIND>(L, k - 1) + cnt != seg) && c < h) p3 = c;      // own head
      }
      if (!in) p1 = p2 = p3 = -1;
      v[j] = n1 ? tr[TILE_LD + TILE_HEAD + p1 - cnt] : tr[(p1 < 0 ? 0 : p1) & 31];
      v2[j] = n2 ? tr[TILE_LD + TILE_HEAD + p2 - cnt] : tr[(p2 < 0 ? 0 : p2) & 31];
      v3[j] = tr[TILE_HEAD + (p3 < 0 ? 0 : p3)];
      const int d1 = p1 < 0 ? -1 : (n1 ? p1 : tile_pos<KIND>(map, k, N, p1));
      q[j] = d1 < 0 ? -1 : seg + d1;
      q2[j] = p2 < 0 ? -1 : seg + p2;      // p2, p3 only on interior rows: no compaction
      q3[j] = p3 < 0 ? -1 : seg + p3;
    }
#pragma unroll
    for (int j = 0; j < FG; ++j) {
      if (q[j] >= 0) gbase[q[j]] = v[j];
      if (q2[j] >= 0) gbase[q2[j]] = v2[j];
      if (q3[j] >= 0) gbase[q3[j]] = v3[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int tile_ga(const double* g) { return (int)((((unsigned long long)g) >> 3) & 15ull); }
// the two emitters of a stage (X_k columns, then U_k columns) over ONE tile: end() of the X part (srbm_stage.hpp) writes its rest out
template <int KX, int KU>
struct TilePair {
  double* tile;          // LDS, 64 x TILE_LD: per lane a ring of the last 32 emitted values (+ the mirror of slot 0)
  double* gbase;         // member's J / H / g array
  int ga;                // (gbase / 8) mod 16
  const Layout* L;
  const int* map;        // edge_map (KIND 1) or nullptr
  int k0, nrow;          // stage of tile row 0, rows really written
  int cx = 0, cu = 0;
  bool xdone = false;
  template <int KIND>
  __device__ __forceinline__ void put_(double v, int& cnt) {
    double* t = tile + ((int)(threadIdx.x & 63) < nrow ? (int)(threadIdx.x & 63) : nrow) * TILE_LD;      // (stays live: 385 uses per stage); idle lanes share the spare row
    t[cnt & 31] = v;
    if ((cnt & 31) == 0) t[32] = v;
    if (cnt < 16) t[TILE_HEAD + cnt] = v;
    ++cnt;      // cnt is a compile-time constant at every call site (the stage functions are straight-line code): the tests below cost nothing
    if ((cnt & 15) == 0 && (cnt >= 32 || tile_first_edge<KIND>() || tile_last_edge<KIND>())) tile_round<KIND>(tile, gbase, ga, L, map, k0, nrow, cnt);
  }
  __device__ __forceinline__ void putx(double v) { put_<KX>(v, cx); }
  __device__ __forceinline__ void putu(double v) { put_<KU>(v, cu); }
  __device__ __forceinline__ void endx() { tile_rest<KX>(tile, gbase, ga, L, map, k0, nrow, cx); xdone = true; }
  __device__ __forceinline__ void finish() {
    if (!xdone) tile_rest<KX>(tile, gbase, ga, L, map, k0, nrow, cx);
    else tile_rest<KU>(tile, gbase, ga, L, map, k0, nrow, cu);
  }
  struct X { TilePair& t; __device__ __forceinline__ void col() {} __device__ __forceinline__ void end() { t.endx(); } __device__ __forceinline__ void put(int, double v) { t.putx(v); } __device__ __forceinline__ void put(double v) { t.putx(v); } };
  struct U { TilePair& t; __device__ __forceinline__ void col() {} __device__ __forceinline__ void end() {} __device__ __forceinline__ void put(int, double v) { t.putu(v); } __device__ __forceinline__ void put(double v) { t.putu(v); } };
};

// Jacobian (FAM 0), Hessian (FAM 1) nonzeros or residual rows (FAM 2) of every stage of one member: one wavefront per member, lane = stage,
// every lane runs the middle-stage instruction stream (first = last = false) and the tile write-out drops the
// placeholders of the two edge stages.  Reported by landing_kernel_name_sweep() for profilers.
// Occupancy: the tile is dynamic LDS of (min(N, 64) + 1) x TILE_LD doubles (the extra row takes the puts of the idle lanes) -- 16 KB at
// N = 40, 9 wavefronts per CU -- and the registers allow 2 wavefronts per SIMD for the Jacobian and residual streams (256 VGPRs), 1 for the
// Hessian stream (AGPR overflow of the default bound; capping it costs 750 B of scratch per lane and doubles its time).  Measured in round 4:
// compiling the streams for 3 wavefronts per SIMD (smaller tile, <= 168 registers, ~100 B of spills) is slower, and so is launching the X_k and
// U_k columns of a stream as two kernels (PART 1, 2: half the instruction stream each) -- the streams are bound by the write path, not by issue.
#ifndef LANDING_SWEEP_WAVES
#define LANDING_SWEEP_WAVES(FAM, PART) ((FAM) == 1 && (PART) == 0 ? 1 : 2)
#endif
__host__ __device__ inline int landing_sweep_tile_rows(int N) { return (N < 64 ? N : 64) + 1; }
struct NullEmit {      // the part of a stage another launch writes: its arithmetic is dead code here
  __device__ __forceinline__ void col() {}
  __device__ __forceinline__ void end() {}
  __device__ __forceinline__ void put(int, double) {}
  __device__ __forceinline__ void put(double) {}
};
// PART 0: both parts of a stage by one wavefront; 1: the X_k columns only; 2: the U_k columns only (the Jacobian stream is launched as 1 + 2:
// half the instruction stream and fewer live values per wavefront, twice the wavefronts)
template <int FAM, int PART = 0>
__global__ void __launch_bounds__(64, LANDING_SWEEP_WAVES(FAM, PART)) landing_sweep_kernel(Layout L, int B, EvalArgs A) {
  const int m = blockIdx.x;
  if (m >= B) return;
  const int N = L.N, ln = threadIdx.x;
  const double* x = A.x + (size_t)m * L.nx;
  const double* p = A.p + (size_t)m * L.np;
  const double* lam_g = A.lam_g ? A.lam_g + (size_t)m * L.ng : nullptr;
#if defined(__HIP__)
  extern __shared__ double tile[];      // landing_sweep_tile_rows(N) x TILE_LD doubles, then (FAM 0) the compaction maps
#else
  static double tile[65 * TILE_LD + 228];      // host emulation (tests/emu): no dynamic LDS
#endif
  int* emap = reinterpret_cast<int*>(tile + landing_sweep_tile_rows(N) * TILE_LD);      // of the edge stages: read at every U-column write-out -- from LDS, not through a dependent global load
  if (FAM == 0 && PART != 1) {
    int e[8];      // all eight loads in flight before the first is waited for
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = ln + 64 * j < 456 ? A.edge_map[ln + 64 * j] : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (ln + 64 * j < 456) emap[ln + 64 * j] = e[j];
    __builtin_amdgcn_wave_barrier();
  }
  for (int k0 = 0; k0 < N; k0 += 64) {
    const int rows_here = N - k0 < 64 ? N - k0 : 64;
    int k = k0 + ln;
    if (k > N - 1) k = N - 1;                       // idle lanes replay the last stage (never written out)
    srbm::StageVars z; srbm::StageParams P;
    load_stage(L, x, p, k, z, P);
    if (FAM == 0) {
      double fz_prev[4] = {0, 0, 0, 0};
      if (PART != 1 && k > 0) { const double* Up = x + L.x_U(k - 1); for (int l = 0; l < 4; ++l) fz_prev[l] = Up[12 + 3 * l + 2]; }
      double* J = A.jac + (size_t)m * L.nnz_jac;
      if (PART == 0) {
        TilePair<0, 1> t{tile, J, tile_ga(J), &L, emap, k0, rows_here};
        TilePair<0, 1>::X ex{t}; TilePair<0, 1>::U eu{t};
        srbm::stage_jac(z, P, false, false, fz_prev, ex, eu);
        t.finish();
      } else if (PART == 1) {
        TilePair<0, 0> t{tile, J, tile_ga(J), &L, nullptr, k0, rows_here};
        TilePair<0, 0>::X ex{t}; NullEmit eu;
        srbm::stage_jac(z, P, false, false, fz_prev, ex, eu);
        t.finish();
      } else {
        TilePair<1, 1> t{tile, J, tile_ga(J), &L, emap, k0, rows_here};
        NullEmit ex; TilePair<1, 1>::X eu{t};
        srbm::stage_jac(z, P, false, false, fz_prev, ex, eu);
        t.finish();
      }
    } else if (FAM == 2) {
      double* G = A.g + (size_t)m * L.ng;
      TilePair<4, 4> t{tile, G, tile_ga(G), &L, nullptr, k0, rows_here};
      TilePair<4, 4>::X og{t};
      srbm::stage_g(z, P, false, og);
      t.finish();
    } else {
      double* H = A.hess + (size_t)m * L.nnz_hess;
      // Multipliers: no branches around the loads.  (Read through LamStage with a runtime `last`, the twelve no-slip sums of stage_hess became
      // twelve conditional blocks of two loads and an s_waitcnt vmcnt(0) each -- a chain of ~15 memory round trips per wavefront, at one
      // wavefront per SIMD a large part of the stream's time.)  Every lane runs stage_hess with last = false (the emitted sequence is the
      // same for every lane anyway); the lane of the last stage, whose 80 rows are numbered differently and have no no-slip rows, reads
      // through LamRemap: both numberings are loaded unconditionally (all rows lie inside the 80 rows of the shortest stage) and selected.
      const LamRemap lam{lam_g + L.g_stage(k), k == N - 1};
      double lps[12];
      {
        const double* lp = lam_g + L.g_stage(k > 0 ? k - 1 : 0);      // stage 0: loaded from its own rows, multiplied away
        const double on = k > 0 ? 1.0 : 0.0;
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
          for (int i = 0; i < 3; ++i) lps[3 * l + i] = on * (lp[16 + 12 * l + 2 + i] + lp[16 + 12 * l + 5 + i]);
      }
      if (PART == 0) {
        TilePair<2, 3> t{tile, H, tile_ga(H), &L, nullptr, k0, rows_here};
        TilePair<2, 3>::X hx{t}; TilePair<2, 3>::U hu{t};
        srbm::stage_hess(z, P, false, false, lam, lps, hx, hu);
        t.finish();
      } else if (PART == 1) {
        TilePair<2, 2> t{tile, H, tile_ga(H), &L, nullptr, k0, rows_here};
        TilePair<2, 2>::X hx{t}; NullEmit hu;
        srbm::stage_hess(z, P, false, false, lam, lps, hx, hu);
        t.finish();
      } else {
        TilePair<3, 3> t{tile, H, tile_ga(H), &L, nullptr, k0, rows_here};
        NullEmit hx; TilePair<3, 3>::X hu{t};
        srbm::stage_hess(z, P, false, false, lam, lps, hx, hu);
        t.finish();
      }
    }
  }
}

// name is reported by landing_kernel_name_sweep() for profilers
// Everything else of the function layer (direct stores): f, grad_f, boundary rows, g of every stage, the X_N blocks
// of J and H, grad_gamma_x / grad_gamma_p.  `A.jac` / `A.hess` stage segments are written by the kernels above.
// LIGHT = true: only what needs no stage evaluation -- terminal-cost f and grad_f, boundary rows, the X_N blocks -- for the common call
// (g, grad_f, Jacobian, Hessian of the terminal-cost NLP; landing_eval_batch picks it).  The full kernel carries the stage code of
// grad_gamma_x / grad_gamma_p / the running cost and is allocated 256 VGPRs + 100 AGPRs for it: one wave per SIMD and no co-residency
// with the Jacobian / Hessian streams, so that in round 2 this trivial part of the sweep took 246 us on the critical path of every call
// (rocprofv3 kernel trace at 4096 members, profiles/r03_sweep_timeline.txt).
template <bool LIGHT>
__global__ void __launch_bounds__(64) landing_sweep_misc_kernel(Layout L, int B, EvalArgs A) {
  const int m = blockIdx.x;
  if (m >= B) return;
  const int N = L.N;
  const double* x = A.x + (size_t)m * L.nx;
  const double* p = A.p + (size_t)m * L.np;
  const double* lam_g = A.lam_g ? A.lam_g + (size_t)m * L.ng : nullptr;
  const double lam_f = A.lam_f ? A.lam_f[m] : 1.0;
  constexpr int NGP = 9 + 18;      // shared-parameter sums over the stages: dt-free part (9, stage_gradp) + QX (12), Qc (3), Qf (3) of the running cost
  __shared__ double red[64][NGP];

  // ---- objective: terminal cost only (gen:83-87) ----
  if (threadIdx.x == 0 && (A.f || A.grad_f || A.ggx || A.ggp)) {
    double s = 0.0;
    for (int i = 0; i < 12; ++i) {
      const double d = x[12 * N + i] - p[12 * N + i];
      s += d * p[L.o_QN + i] * d;
    }
    if (!LIGHT && L.run_cost) for (int k = 0; k < N; ++k) s += run_cost_stage(L, x, p, k, nullptr, nullptr, nullptr);
    if (A.f) A.f[m] = s;
  }
  if (A.grad_f) {
    double* gf = A.grad_f + (size_t)m * L.nx;
    for (int i = threadIdx.x; i < L.nx; i += blockDim.x) {
      const int t = i - 12 * N;
      gf[i] = (t >= 0 && t < 12) ? 2.0 * p[L.o_QN + t] * (x[i] - p[12 * N + t]) : 0.0;
    }
    if (!LIGHT && L.run_cost) {
      __builtin_amdgcn_wave_barrier();
      for (int k = threadIdx.x; k < N; k += blockDim.x) { double* gU = gf + L.x_U(k); (void)run_cost_stage(L, x, p, k, gf + L.x_X(k), gU, gU + 12); }
    }
  }
  // ---- boundary rows (gen:90-97) ----
  if (A.g) {
    double* g = A.g + (size_t)m * L.ng;
    for (int r = threadIdx.x; r < 36; r += blockDim.x) {
      double v;
      if (r < 12) v = x[r];
      else if (r < 24) v = x[12 * N + (r - 12) % 6];
      else v = x[12 * N + 6 + (r - 24) % 6];
      g[r] = v;
    }
  }
  if (A.jac) {
    double* J = A.jac + (size_t)m * L.nnz_jac + L.jx(N);
    for (int i = threadIdx.x; i < 36; i += blockDim.x) J[i] = 1.0;
  }
  if (A.hess) {
    double* H = A.hess + (size_t)m * L.nnz_hess + L.hx(N);
    for (int i = threadIdx.x; i < 12; i += blockDim.x) H[i] = 2.0 * lam_f * p[L.o_QN + i];
  }
  if (LIGHT) return;      // (uniform: everything below needs stage evaluations)
  if (A.ggx && lam_g) {
    double* gx = A.ggx + (size_t)m * L.nx;
    for (int i = threadIdx.x; i < 12; i += blockDim.x) {
      const double* lp = lam_g + L.g_stage(N - 1);
      double v = lam_f * 2.0 * p[L.o_QN + i] * (x[12 * N + i] - p[12 * N + i]) + lp[dyn_row_of_state(i)];
      v += (i < 6) ? lam_g[12 + i] + lam_g[18 + i] : lam_g[24 + i - 6] + lam_g[30 + i - 6];
      gx[12 * N + i] = v;
    }
  }
  double gp_acc[NGP];
  for (int i = 0; i < NGP; ++i) gp_acc[i] = 0.0;

  // ---- stages (only when a per-stage output is left for this kernel) ----
  const bool stage_work = (A.g && !A.g_staged) || (A.ggx && lam_g) || (A.ggp && lam_g);
  for (int k = threadIdx.x; stage_work && k < N; k += blockDim.x) {
    const bool first = (k == 0), last = (k == N - 1);
    srbm::StageVars z; srbm::StageParams P;
    load_stage(L, x, p, k, z, P);
    if (A.g && !A.g_staged) {
      RowStore out{A.g + (size_t)m * L.ng + L.g_stage(k)};
      srbm::stage_g(z, P, last, out);
    }
    double fz_prev[4] = {0, 0, 0, 0};
    if (!first) { const double* Up = x + L.x_U(k - 1); for (int l = 0; l < 4; ++l) fz_prev[l] = Up[12 + 3 * l + 2]; }
    if (A.ggx && lam_g) {
      double* gx = A.ggx + (size_t)m * L.nx;
      const double* lprev = first ? lam_g : lam_g + L.g_stage(k - 1);
      DotLam ex{lam_g + L.g_stage(k), lprev, first, gx + L.x_X(k), 0.0, false};
      DotLam eu{lam_g + L.g_stage(k), lprev, first, gx + L.x_U(k), 0.0, false};
      srbm::stage_jac(z, P, first, last, fz_prev, ex, eu);
      ex.finish(); eu.finish();
    }
    if (A.ggp && lam_g) {
      double o[9];
      LamStage lam{lam_g + L.g_stage(k)};
      srbm::stage_gradp(z, P, last, lam, o);
      A.ggp[(size_t)m * L.np + L.o_dt + k] = o[0];
      for (int i = 1; i < 9; ++i) gp_acc[i] += o[i];
    }
  }
  if (L.run_cost && A.ggx && lam_g) {   // lam_f * gradient of the running cost; stage k owns X_k and U_k (CCC :81-89)
    __syncthreads();                   // the J^T lam sums above also write X_{k+1} of the neighbouring lane
    double* gx = A.ggx + (size_t)m * L.nx;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
      double gX[12], gU[24];
      for (int i = 0; i < 12; ++i) gX[i] = 0.0;
      for (int i = 0; i < 24; ++i) gU[i] = 0.0;
      (void)run_cost_stage(L, x, p, k, gX, gU, gU + 12);
      for (int i = 0; i < 12; ++i) gx[L.x_X(k) + i] += lam_f * gX[i];
      for (int i = 0; i < 24; ++i) gx[L.x_U(k) + i] += lam_f * gU[i];
    }
  }
  if (L.run_cost && A.ggp && lam_g) {   // d/d dt_k of the running cost (the stage loop above wrote the constraint part)
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
      const double dt = p[L.o_dt + k];
      A.ggp[(size_t)m * L.np + L.o_dt + k] += lam_f * run_cost_stage(L, x, p, k, nullptr, nullptr, nullptr) / dt;
      if (L.run_cost == 2) {      // the weights are parameters: d/dQX_i = lam_f sum_k dt_k e_i^2, d/dQc_a, d/dQf_a likewise (per-lane partial sums)
        const double* X = x + L.x_X(k); const double* U = x + L.x_U(k);
        for (int i = 0; i < 12; ++i) { const double e = X[i] - p[12 * k + i]; gp_acc[9 + i] += lam_f * dt * e * e; }
        for (int l = 0; l < 4; ++l) for (int a = 0; a < 3; ++a) {
          const double r = X[a] + L.p_hip[3 * l + a] - U[3 * l + a], u = U[12 + 3 * l + a] - rc_fref(L, p, k, l, a);
          gp_acc[21 + a] += lam_f * dt * r * r; gp_acc[24 + a] += lam_f * dt * u * u;
        }
      }
    }
  }
  if (A.ggp && lam_g) {   // uniform branch: reduce the shared-parameter sums over stages
    for (int i = 0; i < NGP; ++i) red[threadIdx.x][i] = gp_acc[i];
    __syncthreads();
    double* gp = A.ggp + (size_t)m * L.np;
    for (int i = threadIdx.x; i < L.np; i += blockDim.x) {
      if (i >= L.o_dt && i < L.o_dt + N) continue;   // written per stage above
      double v = 0.0;
      int slot = -1;
      if (i == L.o_mu) slot = 1; else if (i == L.o_mass) slot = 2;
      else if (i >= L.o_Ib && i < L.o_Ib + 3) slot = 3 + (i - L.o_Ib);
      else if (i >= L.o_Ib_inv && i < L.o_Ib_inv + 3) slot = 6 + (i - L.o_Ib_inv);
      else if (L.run_cost == 2 && i >= L.o_QX && i < L.o_QX + 12) slot = 9 + (i - L.o_QX);
      else if (L.run_cost == 2 && i >= L.o_Qc && i < L.o_Qc + 3) slot = 21 + (i - L.o_Qc);
      else if (L.run_cost == 2 && i >= L.o_Qf && i < L.o_Qf + 3) slot = 24 + (i - L.o_Qf);
      if (slot >= 0) { for (int t = 0; t < (int)blockDim.x; ++t) v += red[t][slot]; }
      else if (i >= 12 * N && i < 12 * N + 12) { const int t = i - 12 * N; v = -2.0 * lam_f * p[L.o_QN + t] * (x[12 * N + t] - p[12 * N + t]); }
      else if (i >= L.o_QN && i < L.o_QN + 12) { const int t = i - L.o_QN; const double d = x[12 * N + t] - p[12 * N + t]; v = lam_f * d * d; }
      else if (L.run_cost && i < 12 * N) { const int k = i / 12; v = -2.0 * lam_f * p[L.o_dt + k] * rc_QX(L, p, i - 12 * k) * (x[i] - p[i]); }   // Xref_k
      else if (L.run_cost == 2 && i >= L.o_Uref && i < L.o_Uref + 24 * N) {      // Uref: the foot part is inactive in the cost (0), the force part is f_ref
        const int k = (i - L.o_Uref) / 24, j = (i - L.o_Uref) % 24;
        if (j >= 12) v = -2.0 * lam_f * p[L.o_dt + k] * rc_Qf(L, p, (j - 12) % 3) * (x[L.x_U(k) + j] - p[i]);
      }
      gp[i] = v;
    }
  }
}

// Hessian of the Lagrangian with the running cost, in the extended pattern (landing_pattern_hess_rc): entry j takes nonzero
// map[j].x of the casadi_s4-pattern Hessian (or nothing) plus the constant second derivative of the running cost that
// map[j].y encodes -- kind | axis << 4 | stage << 8, kind 1: QX[a] (+ 4 Qc[a] on pos), 2: -Qc[a], 3: Qc[a], 4: Qf[a] -- times
// 2 lam_f dt_k (generate_quadruped_SRBM_CCC.m:81-89).
__global__ void __launch_bounds__(256) landing_hess_rc_kernel(Layout L, int B, int nnz_rc, const int2* __restrict__ map, const double* __restrict__ h4,
                                                              const double* __restrict__ p, const double* __restrict__ lam_f, double* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (j >= nnz_rc || m >= B) return;
  const int2 e = map[j];
  double v = e.x >= 0 ? h4[(size_t)m * L.nnz_hess + e.x] : 0.0;
  const int kind = e.y & 15;
  if (kind && L.run_cost) {
    const int a = (e.y >> 4) & 15, k = e.y >> 8;
    const double* pm = p + (size_t)m * L.np;
    const double w = kind == 1 ? rc_QX(L, pm, a) + (a < 3 ? 4.0 * rc_Qc(L, pm, a) : 0.0) : kind == 2 ? -rc_Qc(L, pm, a) : kind == 3 ? rc_Qc(L, pm, a) : rc_Qf(L, pm, a);
    v += 2.0 * (lam_f ? lam_f[m] : 1.0) * pm[L.o_dt + k] * w;
  }
  out[(size_t)m * nnz_rc + j] = v;
}

// ---- member-level device functions shared with the solver kernel (lane = stage) -------------------
// inlining policy of the solver's phase functions (development switches; the defaults are what the product build uses)
// (measured round 3, A/B on one box: with every phase inlined into landing_ipm_kernel the callee-saved-register traffic disappears
// -- 209 -> 192 GB of HBM traffic per launch -- but the monolithic kernel is allocated and scheduled far worse: condensation 0.104 ->
// 0.231 ms, forward sweep 0.082 -> 0.146 ms per iteration under load, 96 -> 141 ms per batch.  Out of line it is.)
#ifndef LANDING_INL_EVAL_G
#define LANDING_INL_EVAL_G __noinline__
#endif
#ifndef LANDING_INL_TASK
#define LANDING_INL_TASK __noinline__
#endif
#ifndef LANDING_INL_EVAL_JH
#define LANDING_INL_EVAL_JH __noinline__
#endif
// residual g(x) of one member (boundary rows + all stages); the caller synchronises afterwards.
__device__ LANDING_INL_EVAL_G void member_eval_g(const Layout& L, const double* x, const double* p, double* g) {
  const int N = L.N;
  for (int r = threadIdx.x; r < 36; r += blockDim.x) {
    double v;
    if (r < 12) v = x[r];
    else if (r < 24) v = x[12 * N + (r - 12) % 6];
    else v = x[12 * N + 6 + (r - 24) % 6];
    g[r] = v;
  }
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    srbm::StageVars z; srbm::StageParams P;
    load_stage(L, x, p, k, z, P);
    RowStore out{g + L.g_stage(k)};
    srbm::stage_g(z, P, k == N - 1, out);
  }
}
// out-of-line copy for the rare call sites of the solver (initial point, restart from the initial guess, line-search fall-back)
__device__ __noinline__ void member_eval_g_rare(const Layout& L, const double* x, const double* p, double* g) { member_eval_g(L, x, p, g); }
// The derivative tasks of one stage.  Each loads the stage's variables itself: handing them over as `const StageVars&` made the caller
// write the two structs to its stack frame (77 private-memory stores per lane and call) and every task read them back from there.
struct StageIn { srbm::StageVars z; srbm::StageParams P; double fz_prev[4]; };
__device__ __forceinline__ void load_stage_in(const Layout& L, const double* x, const double* p, int k, StageIn& I) {
  load_stage(L, x, p, k, I.z, I.P);
  for (int l = 0; l < 4; ++l) I.fz_prev[l] = 0.0;
  if (k > 0) { const double* Up = x + L.x_U(k - 1); for (int l = 0; l < 4; ++l) I.fz_prev[l] = Up[12 + 3 * l + 2]; }
}
__device__ LANDING_INL_TASK void eval_task_jac(const Layout& L, const double* x, const double* p, int k, double* J) {
  StageIn I; load_stage_in(L, x, p, k, I);
  SeqStoreJ ex{J + L.jx(k)}, eu{J + L.ju(k)};
  srbm::stage_jac(I.z, I.P, k == 0, k == L.N - 1, I.fz_prev, ex, eu);
}
__device__ LANDING_INL_TASK void eval_task_jty(const Layout& L, const double* x, const double* p, int k, const double* y, double* gx) {
  StageIn I; load_stage_in(L, x, p, k, I);
  const bool first = (k == 0);
  const double* lprev = first ? y : y + L.g_stage(k - 1);
  DotLam ex{y + L.g_stage(k), lprev, first, gx + L.x_X(k), 0.0, false};
  DotLam eu{y + L.g_stage(k), lprev, first, gx + L.x_U(k), 0.0, false};
  srbm::stage_jac(I.z, I.P, first, k == L.N - 1, I.fz_prev, ex, eu);
  ex.finish(); eu.finish();
}
// Hessian values of stage k: PART 0 = both column groups, 1 = the X_k columns (29 values), 2 = the U_k columns (148 / 160).  The solver
// runs 1 and 2 on two waves (round 5; the fourth wave used to idle through the derivative phase)
template <int PART>
__device__ LANDING_INL_TASK void eval_task_hess(const Layout& L, const double* x, const double* p, int k, const double* y, double* H) {
  StageIn I; load_stage_in(L, x, p, k, I);
  const bool first = (k == 0);
  double lps[12];
  for (int i = 0; i < 12; ++i) lps[i] = 0.0;
  if (!first) {
    const double* lp = y + L.g_stage(k - 1);
    for (int l = 0; l < 4; ++l) for (int i = 0; i < 3; ++i) lps[3 * l + i] = lp[16 + 12 * l + 2 + i] + lp[16 + 12 * l + 5 + i];
  }
  LamStage lam{y + L.g_stage(k)};
  if (PART == 0) { SeqStoreH hx{H + L.hx(k)}, hu{H + L.hu(k)}; srbm::stage_hess(I.z, I.P, first, k == L.N - 1, lam, lps, hx, hu); }
  else if (PART == 1) { SeqStoreH hx{H + L.hx(k)}; NullEmit hu; srbm::stage_hess(I.z, I.P, first, k == L.N - 1, lam, lps, hx, hu); }
  else { NullEmit hx; SeqStoreH hu{H + L.hu(k)}; srbm::stage_hess(I.z, I.P, first, k == L.N - 1, lam, lps, hx, hu); }
}

// Jacobian / Hessian nonzeros (CCS order) and gx = grad f + J^T y of one member.
__device__ LANDING_INL_EVAL_JH void member_eval_jh(const Layout& L, const double* x, const double* p, const double* y,
                                            double* J, double* H, double* gx, double* tiles = nullptr, const int* edge_map = nullptr, double obj = 1.0) {
  const int N = L.N;      // obj: 1, or 0 in the solver's feasibility phase (no objective)
  for (int i = threadIdx.x; i < 36; i += blockDim.x) J[L.jx(N) + i] = 1.0;
  for (int i = threadIdx.x; i < 12; i += blockDim.x) {
    H[L.hx(N) + i] = obj * 2.0 * p[L.o_QN + i];
    const double* lp = y + L.g_stage(N - 1);
    double v = obj * 2.0 * p[L.o_QN + i] * (x[12 * N + i] - p[12 * N + i]) + lp[dyn_row_of_state(i)];
    v += (i < 6) ? y[12 + i] + y[18 + i] : y[24 + i - 6] + y[30 + i - 6];
    gx[12 * N + i] = v;
  }
  // (stage, task) pairs over the threads: task 0 = Jacobian values, 1 = J^T y (column dot products), 2 = Hessian X_k columns, 3 = Hessian U_k columns
  // one wavefront per task (no divergent calls); with fewer than 4 waves the tasks are looped
  const int nwave = (blockDim.x + 63) >> 6, wave = threadIdx.x >> 6;
  (void)tiles; (void)edge_map;      // (round 2's tiled write-out of the Jacobian task inside the solver: measured slower, removed)
  for (int task = wave; task < 4; task += nwave)
  for (int k = threadIdx.x & 63; k < N; k += 64) {
#ifdef LANDING_DEV_SKIP_TASK          // development probe (tools/dev): which task bounds the derivative phase
    if (task == LANDING_DEV_SKIP_TASK) continue;
#endif
    if (task == 0) eval_task_jac(L, x, p, k, J);
    else if (task == 1) eval_task_jty(L, x, p, k, y, gx);
    else if (task == 2) eval_task_hess<1>(L, x, p, k, y, H);
    else eval_task_hess<2>(L, x, p, k, y, H);
  }
}

// lbg/ubg of one row from p (Opti canonical forms)
__device__ __forceinline__ void bound_of(const Layout& L, const double* p, int r, double& lb, double& ub) {
  const double inf = INFINITY;
  if (r < 36) {
    const int i = r % 6;
    if (r < 6) lb = ub = p[L.o_q_init + i];
    else if (r < 12) lb = ub = p[L.o_qd_init + i];
    else if (r < 18) { lb = p[L.o_q_term_min + i]; ub = inf; }
    else if (r < 24) { lb = -inf; ub = p[L.o_q_term_max + i]; }
    else if (r < 30) { lb = p[L.o_qd_term_min + i]; ub = inf; }
    else { lb = -inf; ub = p[L.o_qd_term_max + i]; }
  } else {
    const int k = (r - 36) / 104, q = (r - 36) % 104;
    const bool last = (k == L.N - 1);
    const int stride = last ? 6 : 12, kin = last ? 2 : 8, fric = last ? 40 : 64, box = last ? 56 : 80;
    if (q < 12) { lb = ub = 0.0; }
    else if (q < 16) { lb = 0.0; ub = p[L.o_f_max]; }
    else if (q < fric) {
      const int t = (q - 16) % stride;
      if (t == 0) { lb = 0.0; ub = inf; }
      else if (t == 1) { lb = -inf; ub = L.comp_eps; }
      else if (t < kin) { if (t < 5) { lb = -inf; ub = L.slip_eps; } else { lb = -L.slip_eps; ub = inf; } }
      else if (t == kin) { lb = -L.kin_box[0]; ub = L.kin_box[0]; }
      else if (t == kin + 1) { lb = -L.kin_box[1]; ub = L.kin_box[1]; }
      else if (t == kin + 2) { lb = -L.kin_box[2]; ub = 0.0; }
      else { lb = -inf; ub = p[L.o_l_leg_max] * p[L.o_l_leg_max]; }
    } else if (q < box) { lb = -inf; ub = 0.0; }
    else {
      const int t = q - box, i = t % 6;
      if (t < 6) { lb = -inf; ub = p[L.o_q_max + i]; }
      else if (t < 12) { lb = p[L.o_q_min + i]; ub = inf; }
      else if (t < 18) { lb = -inf; ub = p[L.o_qd_max + i]; }
      else { lb = p[L.o_qd_min + i]; ub = inf; }
    }
  }
}

// lbg/ubg from p (Opti canonical forms).  One thread per (member,row).
__global__ void landing_bounds_kernel(Layout L, int B, const double* p_all, double* lbg, double* ubg) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * L.ng) return;
  const int m = (int)(idx / L.ng), r = (int)(idx % L.ng);
  double lb, ub;
  bound_of(L, p_all + (size_t)m * L.np, r, lb, ub);
  lbg[idx] = lb; ubg[idx] = ub;
}

}  // namespace landing
