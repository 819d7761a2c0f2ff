// Distance-to-target maps by chain compression.
//
// A state is (rail cell, heading); a move follows one allowed exit of the cell.  Most states have exactly
// one move, so their distance is "one more than their successor's": only states with a choice (two or
// more moves) need a minimum.  Per environment (chain_index_kernel):
//   * anchors = states with >= 2 moves, plus every state of a cell that is some agent's target;
//   * every other state follows its single move until it reaches an anchor: pointer doubling gives, in
//     ceil(log2 S) rounds, the anchor it ends in and the number of steps (dead ends and anchor-free loops
//     get "none");
//   * each anchor gets up to four weighted edges to the anchors its moves lead to.
// Per (environment, group of 32 distinct targets) (distance_chain_kernel): a label-correcting
// shortest-path iteration over the few hundred anchors, all 32 targets side by side in LDS (sources: the
// four states of the target cell at distance 0), then one coalesced pass writes
// dist[state] = steps[state] + dist[anchor of state].  The distances are the unique shortest move counts,
// identical to a breadth-first search over the states, whatever the iteration order.
#include "flm_common.h"

namespace flm {

#define FLM_NONE16 0xFFFFu

// Orders the LDS accesses of one wave's lanes: a wave's LDS operations complete in program order, so it is
// enough to keep the compiler from moving them across this point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

__global__ __launch_bounds__(1024, 8) void chain_index_kernel(
    const uint16_t* __restrict__ grid_all, const int32_t* __restrict__ cell_idx_all,
    const int32_t* __restrict__ n_cells_all, const int32_t* __restrict__ target_cell, int A, int H, int W, int capCells,
    int capAnch, uint16_t* __restrict__ root_all, uint16_t* __restrict__ steps_all, uint32_t* __restrict__ edges_all,
    int32_t* __restrict__ n_anchor_all, int32_t* __restrict__ counts_all, uint16_t* __restrict__ succ_all,
    uint16_t* __restrict__ choice_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int e = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int HW = H * W;
  const int capS = capCells * 4;
  const int S = n_cells_all[e] * 4;
  const uint16_t* grid = grid_all + (size_t)e * HW;
  const int32_t* cell_idx = cell_idx_all + (size_t)e * HW;
  uint16_t* jmp0 = reinterpret_cast<uint16_t*>(smem);   // [capS]
  uint16_t* jmp1 = jmp0 + capS;
  uint16_t* stp0 = jmp1 + capS;
  uint16_t* stp1 = stp0 + capS;
  uint16_t* aidx = stp1 + capS;                          // [capS] anchor number or NONE
  unsigned* flags = reinterpret_cast<unsigned*>(aidx + capS + (capS & 1));  // [capS/32+1] anchor bits
  int* scratch = reinterpret_cast<int*>(flags + (capS + 31) / 32 + 1);       // 16 ints
  uint16_t* root = root_all + (size_t)e * capS;
  uint16_t* steps = steps_all + (size_t)e * capS;
  uint32_t* edges = edges_all + (size_t)e * capAnch * 4;

  for (int i = tid; i < (capS + 31) / 32 + 1; i += T) flags[i] = 0;
  __syncthreads();
  // cells that are a target of some agent: all four states become anchors
  for (int a = tid; a < A; a += T) {
    int tc = target_cell[(size_t)e * A + a];
    if (tc >= 0 && tc < HW) {
      int ci = cell_idx[tc];
      if (ci >= 0) atomicOr(&flags[(ci * 4) >> 5], 0xFu << ((ci * 4) & 31));
    }
  }
  __syncthreads();
  // classify every state; successor of single-move states
  for (int cell = tid; cell < HW; cell += T) {
    const int ci = cell_idx[cell];
    if (ci < 0) continue;
    const int r = cell / W, c = cell - r * W;
    const unsigned code = grid[cell];
    int nbr[4];   // compact number of the rail cell on each side some heading may leave through, -1 = none
    const unsigned exits = (code | (code >> 4) | (code >> 8) | (code >> 12)) & 0xFu;
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int nr = r + dr(n), nc = c + dc(n);
      nbr[n] = (has_exit(exits, n) && nr >= 0 && nr < H && nc >= 0 && nc < W) ? cell_idx[nr * W + nc] : -1;
    }
    const bool dead_end = __popc(code & 0xFFFFu) == 1;
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const int s = ci * 4 + h;
      const int nib = nibble(code, h);
      int deg = 0, nxt = 0;
#pragma unroll
      for (int n = 0; n < 4; n++) {
        if (!has_exit(nib, n) || nbr[n] < 0) continue;
        deg++;
        nxt = nbr[n] * 4 + n;
      }
      const bool is_target = (flags[s >> 5] >> (s & 31)) & 1;
      if (deg >= 2 && !is_target) atomicOr(&flags[s >> 5], 1u << (s & 31));
      const bool anchor = is_target || deg >= 2;
      // forced step of a path walk: the heading allows exactly one exit (reported as "forward"), it leads to a
      // rail cell, and the cell is neither a dead end nor a target
      // (staged in the anchor-number array, which is not in use yet, and copied out with coalesced stores)
      aidx[s] = (!is_target && deg == 1 && __popc(nib) == 1 && __popc(code & 0xFFFFu) != 1) ? (uint16_t)nxt : (uint16_t)FLM_NONE16;
      if (choice_all && aidx[s] == FLM_NONE16) {
        // (only where the walk has no forced successor: it never looks at the other entries)
        // the moves a path walk may take from this state (get_valid_move_actions_: left, forward, right; in a
        // dead end only the U-turn, which counts as forward) and the states they lead to
        unsigned slot[3] = {FLM_NONE16, FLM_NONE16, FLM_NONE16};
        if (dead_end) {
          const int ex = (h + 2) & 3;
          if (has_exit(nib, ex) && nbr[ex] >= 0) slot[1] = (unsigned)(nbr[ex] * 4 + ex);
        } else {
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const int d = (h + i + 3) & 3;                      // heading h-1, h, h+1
            if (has_exit(nib, d) && nbr[d] >= 0) slot[i] = (unsigned)(nbr[d] * 4 + d);
          }
        }
        const unsigned single = (dead_end || __popc(nib) == 1) ? 1u : 0u;
        *reinterpret_cast<uint2*>(choice_all + ((size_t)e * capS + s) * 4) =
            make_uint2(slot[0] | (slot[1] << 16), slot[2] | (single << 16));
      }
      if (anchor || deg == 0) {
        jmp0[s] = (uint16_t)s;   // fixed point: anchors end a chain, dead states lead nowhere
        stp0[s] = 0;
      } else {
        jmp0[s] = (uint16_t)nxt;
        stp0[s] = 1;
      }
    }
  }
  __syncthreads();
  if (succ_all) {
    uint16_t* succ = succ_all + (size_t)e * capS;
    for (int s = tid; s < S; s += T) succ[s] = aidx[s];
  }
  // pointer doubling
  uint16_t* ja = jmp0; uint16_t* jb = jmp1; uint16_t* sa = stp0; uint16_t* sb = stp1;
  for (int span = 1; span < S; span <<= 1) {
    for (int s = tid; s < S; s += T) {
      const int j = ja[s];
      unsigned st = (unsigned)sa[s] + (unsigned)sa[j];
      jb[s] = ja[j];
      sb[s] = (uint16_t)(st > 0xFFFEu ? 0xFFFEu : st);
    }
    __syncthreads();
    uint16_t* t = ja; ja = jb; jb = t;
    t = sa; sa = sb; sb = t;
  }
  // number the anchors in state order
  int n_anchor = 0;
  for (int base = 0; base < S; base += T) {
    const int s = base + tid;
    const bool anc = s < S && ((flags[s >> 5] >> (s & 31)) & 1);
    // 1024-thread flag scan: ballot per wave + wave totals through LDS
    unsigned long long m = __ballot(anc);
    const int w = tid >> 6;
    if ((tid & 63) == 0) scratch[w] = __popcll(m);
    __syncthreads();
    int before = 0, tot = 0;
    for (int i = 0; i < (T >> 6); i++) {
      int v = scratch[i];
      if (i < w) before += v;
      tot += v;
    }
    __syncthreads();
    if (s < S) aidx[s] = anc ? (uint16_t)min(n_anchor + before + __popcll(m & lanemask_lt()), 0xFFFE) : (uint16_t)FLM_NONE16;
    n_anchor += tot;
  }
  __syncthreads();
  const bool overflow = n_anchor > capAnch;
  if (tid == 0) {
    n_anchor_all[e] = overflow ? capAnch : n_anchor;
    if (overflow) atomicOr(&counts_all[e * 4 + 3], FLM_GRAPH_ERR_STATES);
  }
  // per state: anchor reached and steps to it
  for (int s = tid; s < S; s += T) {
    const int j = ja[s];
    const unsigned aj = aidx[j];
    const bool ok = aj != FLM_NONE16 && aj < (unsigned)capAnch;
    root[s] = ok ? (uint16_t)aj : (uint16_t)FLM_NONE16;
    steps[s] = ok ? sa[s] : 0;
  }
  // per anchor: one edge per move
  for (int cell = tid; cell < HW; cell += T) {
    const int ci = cell_idx[cell];
    if (ci < 0) continue;
    const int r = cell / W, c = cell - r * W;
    const unsigned code = grid[cell];
    for (int h = 0; h < 4; h++) {
      const int s = ci * 4 + h;
      const unsigned a = aidx[s];
      if (a == FLM_NONE16 || a >= (unsigned)capAnch) continue;
      const int nib = nibble(code, h);
      uint32_t out[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
      int k = 0;
      for (int n = 0; n < 4; n++) {
        if (!has_exit(nib, n)) continue;
        const int nr = r + dr(n), nc = c + dc(n);
        if (nr < 0 || nr >= H || nc < 0 || nc >= W) continue;
        const int ni = cell_idx[nr * W + nc];
        if (ni < 0) continue;
        const int ns = ni * 4 + n;
        const int j = ja[ns];
        const unsigned aj = aidx[j];
        if (aj == FLM_NONE16 || aj >= (unsigned)capAnch) continue;   // leads to a dead end or an endless loop
        const unsigned w = 1u + sa[ns];
        out[k++] = aj | (w << 16);
      }
      reinterpret_cast<uint4*>(edges)[a] = make_uint4(out[0], out[1], out[2], out[3]);
    }
  }
}

// One workgroup per (environment, 32 target slots).  LDS per anchor: its forward edges, its distance for
// each of the 32 targets, its predecessors (reverse edges) and a 32-bit "dirty" set.  Targets are
// independent problems, so each of the 16 waves owns 2 of them and iterates on its own, without workgroup
// barriers (a wave's LDS operations complete in program order).  Per wave, work lists hold (anchor, target)
// pairs packed in 16 bits.  A round has two phases, one lane per pair:
//   1. every pair whose label got smaller marks the same target dirty at the anchor's predecessors; a
//      pair that becomes dirty is appended once (the LDS atomic OR returns the previous set);
//   2. every dirty pair re-evaluates its label as the minimum over the anchor's moves and, if it got
//      smaller, is appended to the list of changed pairs.
// Each label has one writer per round.  If a list ever fills up, the pairs that did not fit are made up
// for after the rounds by sweeping all labels until nothing changes (labels are always lengths of real
// paths, so the sweep converges to the same shortest distances).
#define FLM_DC_THREADS 1024
#define FLM_DC_LIST 256    // pairs per wave and list
#ifndef FLM_DC_WAVES
#define FLM_DC_WAVES 16     // waves that iterate (32 / FLM_DC_WAVES targets each); the other waves only help to set up and fill
#endif
#define FLM_DA_ROW 34   // uint16 per anchor row: 32 targets + 2 of padding, so that rows start on different LDS banks
__global__ __launch_bounds__(FLM_DC_THREADS) void distance_chain_kernel(
    const uint16_t* __restrict__ root_all, const uint16_t* __restrict__ steps_all, const uint32_t* __restrict__ edges_all,
    const int32_t* __restrict__ n_anchor_all, const int32_t* __restrict__ n_cells_all,
    const int32_t* __restrict__ cell_idx_all, const int32_t* __restrict__ slot_target,
    const int32_t* __restrict__ n_slots, int A, int HW, int capCells, int capAnch, int ldsAnch,
    uint16_t* __restrict__ dist_all, int32_t* __restrict__ counts_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_cnt[FLM_DC_THREADS / 64][2];   // per wave: changed-list size, dirty-list size
  __shared__ int s_lost;                          // some pair did not fit a list
  const int e = blockIdx.y, tid = threadIdx.x, T = FLM_DC_THREADS;
  const int ns = n_slots[e];
  const int base_slot = blockIdx.x * 32;
  if (base_slot >= ns) return;
  const int nslot = min(32, ns - base_slot);
  const int capS = capCells * 4;
  const int S = n_cells_all[e] * 4;
  const int NA = n_anchor_all[e];
  if (NA > ldsAnch) {   // more anchors than the LDS carve holds: flag and leave this environment's rows unwritten
    if (tid == 0) atomicOr(&counts_all[e * 4 + 3], FLM_GRAPH_ERR_STATES);
    return;
  }
  uint4* ed = reinterpret_cast<uint4*>(smem);                        // [ldsAnch] forward edges
  uint16_t* da = reinterpret_cast<uint16_t*>(ed + ldsAnch);          // [ldsAnch][FLM_DA_ROW]
  unsigned* dirty = reinterpret_cast<unsigned*>(da + (size_t)ldsAnch * FLM_DA_ROW);   // [ldsAnch]
  unsigned* rev_off = dirty + ldsAnch;                               // [ldsAnch + 1]
  uint16_t* rev_list = reinterpret_cast<uint16_t*>(rev_off + ldsAnch + 1);    // [4*ldsAnch]
  uint16_t* lists = rev_list + 4 * (size_t)ldsAnch;                  // per wave: changed pairs, dirty pairs
  const uint16_t* root = root_all + (size_t)e * capS;
  const uint16_t* steps = steps_all + (size_t)e * capS;

  for (int a = tid; a < NA; a += T) {
    ed[a] = reinterpret_cast<const uint4*>(edges_all + (size_t)e * capAnch * 4)[a];
    dirty[a] = 0;     // first used as in-degree counter / fill cursor while the reverse lists are built
  }
  for (int i = tid; i < NA * FLM_DA_ROW; i += T) da[i] = (uint16_t)FLM_NONE16;
  if (tid < 2 * (T / 64)) (&s_cnt[0][0])[tid] = 0;
  if (tid == 0) s_lost = 0;
  __syncthreads();
  // reverse adjacency (predecessor anchors)
  for (int a = tid; a < NA; a += T) {
    const uint4 eb = ed[a];
    const unsigned ev[4] = {eb.x, eb.y, eb.z, eb.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (ev[k] != 0xFFFFFFFFu) atomicAdd(&dirty[ev[k] & 0xFFFFu], 1u);
  }
  __syncthreads();
  if (tid < 64) {   // exclusive scan of the in-degrees by one wave
    unsigned carry = 0;
    for (int base = 0; base < NA; base += 64) {
      const int a = base + tid;
      unsigned v = a < NA ? dirty[a] : 0, inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        unsigned t = __shfl_up(inc, o);
        if (tid >= o) inc += t;
      }
      if (a < NA) rev_off[a] = carry + inc - v;
      carry += __shfl(inc, 63);
    }
    if (tid == 0) rev_off[NA] = carry;
  }
  __syncthreads();
  for (int a = tid; a < NA; a += T) dirty[a] = 0;
  __syncthreads();
  for (int a = tid; a < NA; a += T) {
    const uint4 eb = ed[a];
    const unsigned ev[4] = {eb.x, eb.y, eb.z, eb.w};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (ev[k] != 0xFFFFFFFFu) {
        const unsigned t = ev[k] & 0xFFFFu;
        rev_list[rev_off[t] + atomicAdd(&dirty[t], 1u)] = (uint16_t)a;
      }
  }
  __syncthreads();
  for (int a = tid; a < NA; a += T) dirty[a] = 0;
  __syncthreads();
  {
    const int w = tid >> 6, lane = tid & 63;
    uint16_t* lc = lists + (size_t)(w < FLM_DC_WAVES ? w : 0) * 2 * FLM_DC_LIST;   // changed pairs
    uint16_t* ld = lc + FLM_DC_LIST;                      // dirty pairs
    int* cnt = s_cnt[w];
    // sources: the four states of each target cell (this wave's targets are w, w + FLM_DC_WAVES, ...)
    if (w < FLM_DC_WAVES && lane < 4 * (32 / FLM_DC_WAVES)) {
      const int j = w + FLM_DC_WAVES * (lane >> 2), h = lane & 3;
      if (j < nslot) {
        const int tcell = slot_target[(size_t)e * A + base_slot + j];
        if (tcell >= 0 && tcell < HW) {
          const int ci = cell_idx_all[(size_t)e * HW + tcell];
          if (ci >= 0) {
            const unsigned a = root[ci * 4 + h];   // target states are anchors: root = own anchor number
            if (a != FLM_NONE16) {
              da[a * FLM_DA_ROW + j] = 0;
              lc[atomicAdd(&cnt[0], 1)] = (uint16_t)(a | (j << 11));
            }
          }
        }
      }
    }
    wave_sync();
    // list sizes live in (wave-uniform) registers; appends are placed by ballot ranks, so the only LDS
    // atomics left are the dirty-set updates, which rarely meet on one address
    int n_c = w < FLM_DC_WAVES ? min(cnt[0], FLM_DC_LIST) : 0;
    for (int round = 0; round < 65536; round++) {
      if (n_c == 0) break;
      // phase 1: predecessors of the changed pairs become dirty
      int n_d = 0;
      for (int base = 0; base < n_c; base += 64) {
        const int i = base + lane;
        const bool have = i < n_c;
        const unsigned pr = have ? lc[i] : 0u;
        const unsigned r = pr & 0x7FFu, bit = 1u << (pr >> 11);
        unsigned q = have ? rev_off[r] : 0u;
        const unsigned qe = have ? rev_off[r + 1] : 0u;
        while (__ballot(q < qe)) {
          bool fresh = false;
          unsigned bb = 0;
          if (q < qe) {
            bb = rev_list[q++];
            fresh = !(atomicOr(&dirty[bb], bit) & bit);
          }
          const unsigned long long mk = __ballot(fresh);
          if (fresh) {
            const int pos = n_d + __popcll(mk & lanemask_lt());
            if (pos < FLM_DC_LIST) ld[pos] = (uint16_t)(bb | (pr & 0xF800u));
            else s_lost = 1;
          }
          n_d += __popcll(mk);
        }
      }
      wave_sync();
      const int nd = min(n_d, FLM_DC_LIST);
      // phase 2: re-evaluate the dirty pairs; the changed list has been consumed and is refilled
      n_c = 0;
      for (int base = 0; base < nd; base += 64) {
        const int i = base + lane;
        bool improved = false;
        unsigned pr = 0;
        if (i < nd) {
          pr = ld[i];
          const unsigned bb = pr & 0x7FFu, j = pr >> 11;
          const uint4 eb = ed[bb];
          const unsigned ev[4] = {eb.x, eb.y, eb.z, eb.w};
          const unsigned cur = da[bb * FLM_DA_ROW + j];
          unsigned best = cur;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (ev[k] == 0xFFFFFFFFu) continue;
            const unsigned d = da[(ev[k] & 0xFFFFu) * FLM_DA_ROW + j];
            if (d == FLM_NONE16) continue;
            const unsigned cand = d + (ev[k] >> 16);
            if (cand < best) best = cand;
          }
          atomicAnd(&dirty[bb], ~(1u << j));
          if (best < cur) {
            da[bb * FLM_DA_ROW + j] = (uint16_t)(best > 0xFFFEu ? 0xFFFEu : best);
            improved = true;
          }
        }
        const unsigned long long mk = __ballot(improved);
        if (improved) {
          const int pos = n_c + __popcll(mk & lanemask_lt());
          if (pos < FLM_DC_LIST) lc[pos] = (uint16_t)pr;
          else s_lost = 1;
        }
        n_c += __popcll(mk);
      }
      n_c = min(n_c, FLM_DC_LIST);
      wave_sync();
    }
  }
  __syncthreads();
  if (s_lost) {
    // some pairs did not fit a list: sweep all labels until none changes
    const int j = tid & 31;
    for (int it = 0; it <= NA * 32; it++) {
      int changed = 0;
      for (int b = tid >> 5; b < NA; b += (T >> 5)) {
        const uint4 eb = ed[b];
        const unsigned ev[4] = {eb.x, eb.y, eb.z, eb.w};
        const unsigned cur = da[b * FLM_DA_ROW + j];
        unsigned best = cur;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (ev[k] == 0xFFFFFFFFu) continue;
          const unsigned d = da[(ev[k] & 0xFFFFu) * FLM_DA_ROW + j];
          if (d == FLM_NONE16) continue;
          const unsigned cand = d + (ev[k] >> 16);
          if (cand < best) best = cand;
        }
        if (best < cur) {
          da[b * FLM_DA_ROW + j] = (uint16_t)(best > 0xFFFEu ? 0xFFFEu : best);
          changed = 1;
        }
      }
      if (!__syncthreads_or(changed)) break;
    }
  }
  // every state: steps to its anchor + the anchor's distance.  A thread owns 8 consecutive states and
  // writes them to each target's row with one 16-byte store (rows are 16-byte aligned: capCells is even).
  for (int s0 = tid * 8; s0 < S; s0 += T * 8) {
    unsigned an[8], st[8];
    {
      const uint4 r4 = *reinterpret_cast<const uint4*>(root + s0);
      const uint4 t4 = *reinterpret_cast<const uint4*>(steps + s0);
      const unsigned rw[4] = {r4.x, r4.y, r4.z, r4.w}, tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        an[2 * i] = rw[i] & 0xFFFFu; an[2 * i + 1] = rw[i] >> 16;
        st[2 * i] = tw[i] & 0xFFFFu; st[2 * i + 1] = tw[i] >> 16;
      }
    }
    uint16_t* dist = dist_all + ((size_t)e * A + base_slot) * capS + s0;
    for (int jj = 0; jj < nslot; jj++) {
      unsigned d[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        unsigned v = FLM_NONE16;
        if (an[i] != FLM_NONE16 && s0 + i < S) {
          const unsigned d0 = da[an[i] * FLM_DA_ROW + jj];
          if (d0 != FLM_NONE16) {
            v = d0 + st[i];
            if (v > 0xFFFEu) v = 0xFFFEu;
          }
        }
        d[i] = v;
      }
      *reinterpret_cast<uint4*>(dist + (size_t)jj * capS) =
          make_uint4(d[0] | (d[1] << 16), d[2] | (d[3] << 16), d[4] | (d[5] << 16), d[6] | (d[7] << 16));
    }
  }
}

}  // namespace flm

extern "C" int flm_chain_index(const uint16_t* grid, const int32_t* cell_idx, const int32_t* n_cells,
                               const int32_t* target_cell, int E, int A, int H, int W, int capCells, int capAnch,
                               uint16_t* root, uint16_t* steps, uint32_t* edges, int32_t* n_anchor, int32_t* counts,
                               uint16_t* succ, uint16_t* choice, void* stream) {
  if (!grid || !cell_idx || !n_cells || !target_cell || !root || !steps || !edges || !n_anchor || !counts) return FLM_E_ARG;
  if (E <= 0 || A <= 0 || H <= 0 || W <= 0 || capCells <= 0 || capCells * 4 >= 65535 || capAnch <= 0 || capAnch > 0xFFF0)
    return FLM_E_ARG;
  const size_t capS = (size_t)capCells * 4;
  size_t lds = capS * 2 * 5 + 16 + ((capS + 31) / 32 + 1) * 4 + 128;
  if (lds > 160 * 1024 - 512) return FLM_E_ARG;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(flm::chain_index_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return FLM_E_LAUNCH;
  hipLaunchKernelGGL(flm::chain_index_kernel, dim3(E), dim3(1024), lds, (hipStream_t)stream, grid, cell_idx, n_cells,
                     target_cell, A, H, W, capCells, capAnch, root, steps, edges, n_anchor, counts, succ, choice);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_distance_chains(const uint16_t* root, const uint16_t* steps, const uint32_t* edges,
                                   const int32_t* n_anchor, const int32_t* n_cells, const int32_t* cell_idx,
                                   const int32_t* slot_target, const int32_t* n_slots, int E, int A, int H, int W,
                                   int capCells, int capAnch, int ldsAnch, uint16_t* dist, int32_t* counts,
                                   void* stream) {
  if (!root || !steps || !edges || !n_anchor || !n_cells || !cell_idx || !slot_target || !n_slots || !dist || !counts)
    return FLM_E_ARG;
  if (E <= 0 || E > 65535 || A <= 0 || H <= 0 || W <= 0 || capCells <= 0 || capAnch <= 0 || ldsAnch <= 0 ||
      ldsAnch > capAnch)
    return FLM_E_ARG;
  if (ldsAnch > 2047) return FLM_E_ARG;   // anchors are packed in 11 bits on the work lists
  if (capCells & 1) return FLM_E_ARG;     // 16-byte row stores
  size_t lds = (size_t)ldsAnch * (16 + 2 * FLM_DA_ROW + 4 + 4 + 8) + 16 + FLM_DC_WAVES * 2 * FLM_DC_LIST * 2;
  if (lds > 160 * 1024 - 512) return FLM_E_ARG;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(flm::distance_chain_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return FLM_E_LAUNCH;
  dim3 grid((A + 31) / 32, E);
  hipLaunchKernelGGL(flm::distance_chain_kernel, grid, dim3(FLM_DC_THREADS), lds, (hipStream_t)stream, root, steps, edges, n_anchor,
                     n_cells, cell_idx, slot_target, n_slots, A, H * W, capCells, capAnch, ldsAnch, dist, counts);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
