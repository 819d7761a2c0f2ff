// Shared device helpers for the flatland-marl hot-path kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/flmarl.h"

#define FLM_WAVE 64

// Raises a kernel's dynamic-LDS limit to what the CU has (160 KiB less the kernel's static LDS), once per device and
// kernel, under std::call_once: the launch path makes no attribute call afterwards and concurrent host threads cannot
// race on it.  Use inside an entry point (returns FLM_E_LAUNCH from it on failure); every use is a cache of its own.
#define FLM_RAISE_LDS_ONCE(kernel)                                                                            \
  do {                                                                                                        \
    static std::once_flag flm_once_[64];                                                                      \
    static bool flm_ok_[64];                                                                                  \
    int flm_dev_ = -1;                                                                                        \
    if (hipGetDevice(&flm_dev_) != hipSuccess || flm_dev_ < 0 || flm_dev_ >= 64) return FLM_E_LAUNCH;         \
    std::call_once(flm_once_[flm_dev_], [&] { flm_ok_[flm_dev_] = flm::raise_lds_limit(reinterpret_cast<const void*>(kernel)); }); \
    if (!flm_ok_[flm_dev_]) return FLM_E_LAUNCH;                                                              \
  } while (0)

#define FLM_CHECK_LAUNCH()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH; \
  } while (0)

namespace flm {

inline bool raise_lds_limit(const void* kernel) {
  hipFuncAttributes attr;
  if (hipFuncGetAttributes(&attr, kernel) != hipSuccess) return false;
  const int room = 160 * 1024 - (int)attr.sharedSizeBytes;
  return room > 0 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room) == hipSuccess;
}

__device__ __forceinline__ int dr(int h) { return (h == 0) ? -1 : ((h == 2) ? 1 : 0); }
__device__ __forceinline__ int dc(int h) { return (h == 1) ? 1 : ((h == 3) ? -1 : 0); }
__device__ __forceinline__ int nibble(unsigned code, int h) { return (code >> ((3 - h) * 4)) & 0xF; }
__device__ __forceinline__ int has_exit(int nib, int n) { return (nib >> (3 - n)) & 1; }

// Heading from cell p1 towards cell p2 (row axis wins ties; equal cells give South).
__device__ __forceinline__ int direction_to_point(int r1, int c1, int r2, int c2) {
  int a = r1 - r2, b = c1 - c2;
  if (a * a >= b * b) return a > 0 ? 0 : 2;
  return b > 0 ? 3 : 1;
}

// Candidate moves of an agent heading h in a cell, in the order left, forward, right.
// act[i] in {1,2,3}; nd[i] = new heading.  Returns the count (0..3).
__device__ __forceinline__ int valid_moves(unsigned code, int h, int* act, int* nd) {
  int nib = nibble(code, h);
  int n = 0;
  if (__popc(code & 0xFFFFu) == 1) {  // dead end: only the U-turn, reported as "forward"
    int ex = (h + 2) & 3;
    if (has_exit(nib, ex)) { act[0] = 2; nd[0] = ex; n = 1; }
  } else {
    bool single = __popc(nib) == 1;
#pragma unroll
    for (int i = -1; i <= 1; i++) {
      int d = (h + i + 4) & 3;
      if (has_exit(nib, d)) { act[n] = single ? 2 : 2 + i; nd[n] = d; n++; }
    }
  }
  return n;
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
  unsigned lane = threadIdx.x & 63;
  return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

// Exclusive prefix of a per-thread flag over the workgroup (up to 1024 threads), plus the workgroup total.
// scratch: 16 ints of LDS; contains two barriers.
__device__ __forceinline__ int block_flag_scan(bool flag, int* scratch, int* total) {
  const unsigned long long m = __ballot(flag);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const int within = __popcll(m & lanemask_lt());
  if ((threadIdx.x & 63) == 0) scratch[w] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; i++) {
    const int v = scratch[i];
    if (i < w) base += v;
    tot += v;
  }
  __syncthreads();
  *total = tot;
  return base + within;
}

// K flags per thread (bit k of `bits`), ranked in the order (k, thread) - the order of a strided loop
// `for (k) item = k * blockDim.x + threadIdx.x` - with three barriers for all K instead of two per flag.
// multi_flag_counts leaves the exclusive prefix of every (k, wave) and the total in scratch (16 * K + 1 ints of LDS)
// and returns the total; multi_flag_rank then gives the rank of this thread's flag k.
template <int K>
__device__ __forceinline__ int multi_flag_counts(unsigned bits, int* scratch) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();   // an earlier use of scratch is over
#pragma unroll
  for (int k = 0; k < K; k++) {
    const unsigned long long m = __ballot((bits >> k) & 1u);
    if (lane == 0) scratch[k * 16 + w] = __popcll(m);
  }
  __syncthreads();
  if (w == 0) {
    constexpr int C = (16 * K + 63) / 64;   // consecutive entries per lane
    int v[C], sum = 0;
#pragma unroll
    for (int c = 0; c < C; c++) {
      const int i = lane * C + c;
      v[c] = (i < 16 * K && (i & 15) < nw) ? scratch[i] : 0;
      sum += v[c];
    }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    int run = inc - sum;
#pragma unroll
    for (int c = 0; c < C; c++) {
      const int i = lane * C + c;
      if (i < 16 * K) scratch[i] = run;
      run += v[c];
    }
    if (lane == 63) scratch[16 * K] = inc;
  }
  __syncthreads();
  return scratch[16 * K];
}

__device__ __forceinline__ int multi_flag_rank(unsigned bits, int k, const int* scratch) {
  const unsigned long long m = __ballot((bits >> k) & 1u);
  return scratch[k * 16 + (threadIdx.x >> 6)] + __popcll(m & lanemask_lt());
}

// Exclusive prefix sum of a per-thread int over the workgroup (up to 1024 threads), plus the workgroup total.
__device__ __forceinline__ int block_int_scan(int v, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) scratch[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; i++) {
    const int s = scratch[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// Greedy descent on a distance row (dist[cell_idx*4+heading], 0xFFFF = unreachable) from (cell, h)
// to tcell: at most max_depth moves, ties broken left, forward, right; the best distance seen so far is
// carried from step to step.  Elements are written while they fit in cap.  Returns the element count
// (may exceed cap), or -1 when some step has no admissible move.
// succ (optional): for every state with exactly one possible move that is not on any target cell, the
// state that move leads to (0xFFFF otherwise).  Such a step needs no distance look-up: on a reachable
// state the single successor is one move closer, so the walk follows succ and only evaluates candidates
// where there is a choice.
// choice (optional, with succ): per state four uint16: the states the moves left, forward, right lead to
// (0xFFFF = no such move) and a flag "the move is reported as forward" (a single exit, or the U-turn of a
// dead end, which sits in the forward slot).  With it a step with a choice reads one table entry and the
// candidates' distances instead of the cell code and the state numbers of the neighbouring cells.
// WIDE: elements are collected four at a time and leave as 16-byte stores (many walks side by side: fewer store
// instructions); otherwise every element is stored as it is found (one walk alone: the shortest dependent chain per
// step).  The tables may live in LDS or in global memory.
template <bool WIDE = true>
__device__ __forceinline__ int greedy_path(const uint16_t* __restrict__ grid, const int32_t* __restrict__ cell_idx,
                                           const uint16_t* __restrict__ dist, const uint16_t* __restrict__ succ,
                                           const uint16_t* __restrict__ choice, int H, int W, int cell, int h, int tcell,
                                           int max_depth, int32_t* pc, uint32_t* pm, int cap) {
  unsigned best_d = 0xFFFFFFFFu;
  int n = 0, depth = 0;
  int s = -1;                 // compact state number of (cell, h) while it is known
  bool reach = false;         // the current state is known to reach the target, best_d is its distance
  // elements are collected four at a time and stored as 16 bytes when the output rows are 16-byte aligned
  const bool vec = WIDE && ((reinterpret_cast<size_t>(pc) | reinterpret_cast<size_t>(pm)) & 15) == 0;
  int bc[4] = {0, 0, 0, 0};
  unsigned bm[4] = {0, 0, 0, 0};
  auto emit = [&](int c, unsigned m) {
    if (vec) {
      bc[n & 3] = c;
      bm[n & 3] = m;
      if ((n & 3) == 3 && n < cap) {
        *reinterpret_cast<int4*>(pc + n - 3) = make_int4(bc[0], bc[1], bc[2], bc[3]);
        *reinterpret_cast<uint4*>(pm + n - 3) = make_uint4(bm[0], bm[1], bm[2], bm[3]);
      }
    } else if (n < cap) {
      pc[n] = c;
      pm[n] = m;
    }
    n++;
  };
  auto flush = [&]() {     // the last, partial group of four
    if (vec)
      for (int i = n & ~3; i < n && i < cap; i++) {
        pc[i] = bc[i & 3];
        pm[i] = bm[i & 3];
      }
  };
  if (succ && cell >= 0 && cell < H * W) {
    const int idx = cell_idx[cell];
    if (idx >= 0) s = idx * 4 + h;
  }
  while (cell != tcell && depth < max_depth) {
    if (s >= 0) {
      const unsigned nx = succ[s];
      if (nx != 0xFFFFu) {
        if (!reach) {
          const unsigned d0 = dist[s];
          if (d0 == FLM_DIST_INF) return -1;      // nothing can be reached from here
          best_d = d0;
          reach = true;
        }
        // the whole forced run: no state of it lies on a target cell (such states have no forced successor), so
        // the walk cannot arrive inside it; only the horizon can end it early
        unsigned nxt = nx;
        do {
          const int nd = (int)(nxt & 3u);
          emit(cell, (unsigned)h | (2u << 8) | ((unsigned)nd << 16));
          depth++;
          cell += dr(nd) * W + dc(nd);
          h = nd;
          s = (int)nxt;
          best_d -= 1;
          if (depth >= max_depth) break;
          nxt = succ[s];
        } while (nxt != 0xFFFFu);
        continue;
      }
    }
    if (choice && s >= 0) {
      const uint2 raw = *reinterpret_cast<const uint2*>(choice + (size_t)s * 4);
      const unsigned ns[3] = {raw.x & 0xFFFFu, raw.x >> 16, raw.y & 0xFFFFu};
      int best = -1;
      unsigned best_s = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        if (ns[i] == 0xFFFFu) continue;
        const unsigned dd = dist[ns[i]];
        if (dd != FLM_DIST_INF && dd < best_d) {
          best = i;
          best_d = dd;
          best_s = ns[i];
        }
      }
      if (best < 0) return -1;
      const int nd1 = (int)(best_s & 3u);
      const unsigned action = (raw.y >> 16) & 1u ? 2u : (unsigned)(1 + best);
      emit(cell, (unsigned)h | (action << 8) | ((unsigned)nd1 << 16));
      depth++;
      cell += dr(nd1) * W + dc(nd1);
      h = nd1;
      s = (int)best_s;
      reach = true;
      continue;
    }
    int r = cell / W, c = cell - r * W;
    int act[3], nd[3];
    int k = valid_moves(grid[cell], h, act, nd);
    int best = -1, best_idx = -1;
    for (int i = 0; i < k; i++) {
      int nr = r + dr(nd[i]), nc = c + dc(nd[i]);
      if (nr < 0 || nr >= H || nc < 0 || nc >= W) continue;
      int idx = cell_idx[nr * W + nc];
      if (idx < 0) continue;
      unsigned dd = dist[idx * 4 + nd[i]];
      if (dd != FLM_DIST_INF && dd < best_d) {
        best = i;
        best_d = dd;
        best_idx = idx;
      }
    }
    if (best < 0) return -1;
    emit(cell, (unsigned)h | ((unsigned)act[best] << 8) | ((unsigned)nd[best] << 16));
    depth++;
    cell = (r + dr(nd[best])) * W + c + dc(nd[best]);
    h = nd[best];
    if (succ) s = best_idx * 4 + h;
    reach = true;
  }
  if (depth < max_depth) emit(cell, (unsigned)h | (4u << 8) | ((unsigned)h << 16));   // closing element: stop on the cell reached
  flush();
  return n;
}

}  // namespace flm
// Diagnostic only (flm_debug_fused_stamps): when set, the one-workgroup-per-environment kernels of the reset path
// write shader-clock stamps at their phase boundaries to this device buffer, [3][E][16] 64-bit words
// (0: anchor_decisions / walk_columns, 1: graph_build, 2: chain_index).  Nothing else reads it.
extern unsigned long long* g_flm_stamps;
extern int g_flm_stamps_envs;
static inline unsigned long long* flm_stamps_for(int E) { return E <= g_flm_stamps_envs ? g_flm_stamps : nullptr; }


