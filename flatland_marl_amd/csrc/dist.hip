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
// shortest-path iteration over the few hundred anchors (flm_sweep.h: one wave per four targets, the labels
// of an anchor side by side in LDS; sources: the four states of the target cell at distance 0), then one
// coalesced pass writes
// dist[state] = steps[state] + dist[anchor of state].  The distances are the unique shortest move counts,
// identical to a breadth-first search over the states, whatever the iteration order.
#include "flm_common.h"
#include "flm_sweep.h"

namespace flm {

#define FLM_NONE16 0xFFFFu

__global__ __launch_bounds__(1024, 8) void chain_index_kernel(
    const uint16_t* __restrict__ grid_all, const int32_t* __restrict__ cell_idx_all,
    const int32_t* __restrict__ n_cells_all, const int32_t* __restrict__ target_cell, int A, int H, int W, int capCells,
    int capAnch, uint16_t* __restrict__ root_all, uint16_t* __restrict__ steps_all, uint32_t* __restrict__ edges_all,
    int32_t* __restrict__ n_anchor_all, int32_t* __restrict__ counts_all, uint16_t* __restrict__ succ_all,
    uint16_t* __restrict__ choice_all, uint16_t* __restrict__ walk_all, uint32_t* __restrict__ move_edge_all,
    uint16_t* __restrict__ move_succ_all, const int32_t* __restrict__ idx_cell_all,
    unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int e = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  auto stamp = [&](int i) {   // diagnostic only
    if (stamps && threadIdx.x == 0) stamps[((size_t)2 * gridDim.x + blockIdx.x) * 16 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  const int HW = H * W;
  const int capS = capCells * 4;
  const int S = n_cells_all[e] * 4;
  const uint16_t* grid = grid_all + (size_t)e * HW;
  const int32_t* cell_idx = cell_idx_all + (size_t)e * HW;
  // the rail cells are visited through the compact numbering's inverse table when the caller has one (a fraction
  // of the H*W grid cells), otherwise by scanning the grid
  const int32_t* idx_cell = idx_cell_all ? idx_cell_all + (size_t)e * capCells : nullptr;
  const int n_visit = idx_cell ? S / 4 : HW;
  // per state one word: the state its forced moves have been followed to so far | the number of moves << 16.
  // Pointer doubling updates the words in place: whatever mixture of old and new words a thread reads, "d moves
  // from s lead to j" stays a true statement, and the fixed point (every pointer names an anchor or a state
  // without moves) does not depend on the order.
  unsigned* jw = reinterpret_cast<unsigned*>(smem);       // [capS]
  uint16_t* aidx = reinterpret_cast<uint16_t*>(jw + capS);  // [capS] anchor number or NONE
  unsigned* flags = reinterpret_cast<unsigned*>(aidx + capS + (capS & 1));  // [capS/32+1] anchor bits
  int* scratch = reinterpret_cast<int*>(flags + (capS + 31) / 32 + 1);       // 16 * 8 + 1 ints (multi_flag_counts)
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
  // the cell visited in pass `it`: its compact number, transition code and the compact numbers of the rail cells
  // on the sides some heading may leave through (-1 = none)
  auto visit = [&](int it, int& ci, unsigned& code, int (&nbr)[4]) -> bool {
    const int cell = idx_cell ? idx_cell[it] : it;
    ci = idx_cell ? it : cell_idx[cell];
    if (ci < 0) return false;
    const int r = cell / W, c = cell - r * W;
    code = grid[cell];
    const unsigned exits = (code | (code >> 4) | (code >> 8) | (code >> 12)) & 0xFu;
#pragma unroll
    for (int n = 0; n < 4; n++) {
      const int nr = r + dr(n), nc = c + dc(n);
      nbr[n] = (has_exit(exits, n) && nr >= 0 && nr < H && nc >= 0 && nc < W) ? cell_idx[nr * W + nc] : -1;
    }
    return true;
  };
  auto classify = [&](int ci, unsigned code, const int (&nbr)[4]) {
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
      // forced step of a path walk: the heading allows exactly one exit (reported as "forward"), it leads to a
      // rail cell, and the cell is neither a dead end nor a target
      const bool forced = !is_target && deg == 1 && __popc(nib) == 1 && __popc(code & 0xFFFFu) != 1;
      // every live state that is not forced is an anchor: a walk only ever decides at anchors
      const bool anchor = is_target || deg >= 2 || (deg == 1 && !forced);
      if (anchor && !is_target) atomicOr(&flags[s >> 5], 1u << (s & 31));
      // (staged in the anchor-number array, which is not in use yet, and copied out with coalesced stores)
      aidx[s] = forced ? (uint16_t)nxt : (uint16_t)FLM_NONE16;
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
      // fixed point: anchors end a chain, dead states lead nowhere; the others take their one move
      jw[s] = (anchor || deg == 0) ? (unsigned)s : ((unsigned)nxt | (1u << 16));
    }
  };
  // The first passes keep what they looked up in registers for the edge phase below (same thread, same cell): the
  // look-ups are chains of dependent global loads.
  constexpr int KI = 4;
  unsigned kept_code[KI], kept_n01[KI], kept_n23[KI];
  {
    // three rounds of loads, none of them conditional (an index that does not exist reads a neighbouring one and
    // the value is dropped), so that the loads of all passes are in flight together
    int v_cell[KI], v_ci[KI], v_nbr[KI][4];
    unsigned v_code[KI];
    const int last = n_visit > 0 ? n_visit - 1 : 0;
#pragma unroll
    for (int k = 0; k < KI; k++) {
      const int itc = min(k * T + tid, last);
      v_cell[k] = idx_cell ? idx_cell[itc] : itc;
      if (n_visit <= 0 || v_cell[k] < 0 || v_cell[k] >= HW) v_cell[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < KI; k++) {
      v_code[k] = grid[v_cell[k]];
      v_ci[k] = idx_cell ? min(k * T + tid, last) : cell_idx[v_cell[k]];
    }
#pragma unroll
    for (int k = 0; k < KI; k++) {
      const int r = v_cell[k] / W, c = v_cell[k] - r * W;
      const unsigned code = v_code[k];
      const unsigned exits = (code | (code >> 4) | (code >> 8) | (code >> 12)) & 0xFu;
#pragma unroll
      for (int n = 0; n < 4; n++) {
        const int nr = r + dr(n), nc = c + dc(n);
        const bool ok = has_exit(exits, n) && nr >= 0 && nr < H && nc >= 0 && nc < W;
        const int got = cell_idx[ok ? nr * W + nc : v_cell[k]];
        v_nbr[k][n] = ok ? got : -1;
      }
    }
#pragma unroll
    for (int k = 0; k < KI; k++) {
      const bool live = k * T + tid < n_visit && v_ci[k] >= 0;
      if (live) classify(v_ci[k], v_code[k], v_nbr[k]);
      kept_code[k] = live ? (v_code[k] | 0x10000u | ((unsigned)v_ci[k] << 17)) : 0u;   // code, valid, compact number
      kept_n01[k] = (unsigned)(v_nbr[k][0] & 0xFFFF) | ((unsigned)(v_nbr[k][1] & 0xFFFF) << 16);
      kept_n23[k] = (unsigned)(v_nbr[k][2] & 0xFFFF) | ((unsigned)(v_nbr[k][3] & 0xFFFF) << 16);
    }
  }
  for (int it = KI * T + tid; it < n_visit; it += T) {
    int ci, nbr[4];
    unsigned code;
    if (visit(it, ci, code, nbr)) classify(ci, code, nbr);
  }
  __syncthreads();
  stamp(1);
  if (succ_all) {
    uint16_t* succ = succ_all + (size_t)e * capS;
    for (int s = tid; s < S; s += T) succ[s] = aidx[s];
  }
  // pointer doubling
  // (a round in which every pointer already names a fixed point changes nothing: the loop ends there, after
  // about log2 of the longest run of forced moves; only anchor-free loops keep it going to log2 S)
  for (int span = 1; span < S; span <<= 1) {
    int more = 0;
    for (int s = tid; s < S; s += T) {
      const unsigned w = jw[s];
      const unsigned j = w & 0xFFFFu;
      const unsigned wj = jw[j];
      const unsigned jj = wj & 0xFFFFu;
      unsigned st = (w >> 16) + (wj >> 16);
      if (st > 0xFFFEu) st = 0xFFFEu;
      more |= jj != j;
      jw[s] = jj | (st << 16);
    }
    more = __syncthreads_or(more);
    if (!more) break;
  }
  stamp(2);
  // number the anchors in state order: eight strided passes over the states per ranking
  int n_anchor = 0;
  constexpr int KB = 8;
  for (int base0 = 0; base0 < S; base0 += KB * T) {
    unsigned anc_bits = 0;
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int s = base0 + k * T + tid;
      if (s < S && ((flags[s >> 5] >> (s & 31)) & 1)) anc_bits |= 1u << k;
    }
    const int tot = multi_flag_counts<KB>(anc_bits, scratch);
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int s = base0 + k * T + tid;
      const int rank = n_anchor + multi_flag_rank(anc_bits, k, scratch);
      if (s < S) aidx[s] = ((anc_bits >> k) & 1u) ? (uint16_t)min(rank, 0xFFFE) : (uint16_t)FLM_NONE16;
    }
    n_anchor += tot;
  }
  __syncthreads();
  stamp(3);
  const bool overflow = n_anchor > capAnch;
  if (tid == 0) {
    n_anchor_all[e] = overflow ? capAnch : n_anchor;
    if (overflow) atomicOr(&counts_all[e * 4 + 3], FLM_GRAPH_ERR_STATES);
  }
  // per state: anchor reached and steps to it
  for (int s = tid; s < S; s += T) {
    const int j = (int)(jw[s] & 0xFFFFu);
    const unsigned aj = aidx[j];
    const bool ok = aj != FLM_NONE16 && aj < (unsigned)capAnch;
    root[s] = ok ? (uint16_t)aj : (uint16_t)FLM_NONE16;
    steps[s] = ok ? (uint16_t)(jw[s] >> 16) : 0;
    if (walk_all) {
      // what a path walk does in state s: a forced state names its successor, an anchor its own number
      // (bit 15 set), a state without moves 0xFFFF.  The successor is the value this thread stored to succ above.
      const unsigned own = aidx[s];
      unsigned wv = FLM_NONE16;
      if (own != FLM_NONE16) wv = own < (unsigned)capAnch && own < 0x7FFFu ? (0x8000u | own) : FLM_NONE16;
      else wv = succ_all[(size_t)e * capS + s];
      walk_all[(size_t)e * capS + s] = (uint16_t)wv;
    }
  }
  stamp(4);
  // per anchor: one edge per move
  auto emit_edges = [&](int ci, unsigned code, const int (&nbr)[4]) {
    const bool dead_end = __popc(code & 0xFFFFu) == 1;
#pragma unroll
    for (int h = 0; h < 4; h++) {
      const int s = ci * 4 + h;
      const unsigned a = aidx[s];
      if (a == FLM_NONE16 || a >= (unsigned)capAnch) continue;
      const int nib = nibble(code, h);
      uint32_t o0 = 0xFFFFFFFFu, o1 = 0xFFFFFFFFu, o2 = 0xFFFFFFFFu, o3 = 0xFFFFFFFFu;   // the edges, packed to the front
      uint32_t via[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};         // the edge through each side
      int k = 0;
#pragma unroll
      for (int n = 0; n < 4; n++) {
        if (!has_exit(nib, n) || nbr[n] < 0) continue;
        const int ns = nbr[n] * 4 + n;
        const unsigned wn = jw[ns];
        const unsigned aj = aidx[wn & 0xFFFFu];
        if (aj == FLM_NONE16 || aj >= (unsigned)capAnch) continue;   // leads to a dead end or an endless loop
        via[n] = aj | ((1u + (wn >> 16)) << 16);
        if (k == 0) o0 = via[n];
        else if (k == 1) o1 = via[n];
        else if (k == 2) o2 = via[n];
        else o3 = via[n];
        k++;
      }
      reinterpret_cast<uint4*>(edges)[a] = make_uint4(o0, o1, o2, o3);
      if (move_edge_all) {
        // the same edges in the order a path walk tries its moves (left, forward, right; in a dead end only the
        // U-turn, in the forward slot) beside the states those moves lead to; the fourth entries hold the flag
        // "the move is reported as forward" (a single exit, or the U-turn of a dead end)
        uint32_t me[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        unsigned ms[3] = {FLM_NONE16, FLM_NONE16, FLM_NONE16};
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const int d = dead_end ? ((h + 2) & 3) : ((h + i + 3) & 3);
          if (dead_end && i != 1) continue;
          if (!has_exit(nib, d) || nbr[d] < 0) continue;
          ms[i] = (unsigned)(nbr[d] * 4 + d);
          me[i] = via[d];
        }
        const unsigned single = (dead_end || __popc(nib) == 1) ? 1u : 0u;
        reinterpret_cast<uint4*>(move_edge_all + (size_t)e * capAnch * 4)[a] = make_uint4(me[0], me[1], me[2], single);
        reinterpret_cast<uint2*>(move_succ_all + (size_t)e * capAnch * 4)[a] =
            make_uint2(ms[0] | (ms[1] << 16), ms[2] | (single << 16));
      }
    }
  };
#pragma unroll
  for (int k = 0; k < KI; k++) {
    if (!(kept_code[k] & 0x10000u)) continue;
    const int nbr[4] = {(int)(int16_t)(kept_n01[k] & 0xFFFFu), (int)(int16_t)(kept_n01[k] >> 16),
                        (int)(int16_t)(kept_n23[k] & 0xFFFFu), (int)(int16_t)(kept_n23[k] >> 16)};
    emit_edges((int)(kept_code[k] >> 17), kept_code[k] & 0xFFFFu, nbr);
  }
  for (int it = KI * T + tid; it < n_visit; it += T) {
    int ci, nbr[4];
    unsigned code;
    if (visit(it, ci, code, nbr)) emit_edges(ci, code, nbr);
  }
  stamp(5);
}

// One workgroup per (environment, 32 target slots).  LDS: the anchors' out-edges and, per group of four
// targets, one 8-byte row of labels per anchor.  Targets are independent problems: wave g relaxes the labels
// of targets 4g..4g+3 on its own (anchor_sweep, no workgroup barrier in the loop); then all waves write
// dist[state] = steps[state] + label[anchor of state] with coalesced 16-byte stores.
#define FLM_DC_THREADS 1024
#define FLM_DC_TPG 4                      // targets per wave
#define FLM_DC_GROUPS (32 / FLM_DC_TPG)   // waves that iterate; the others only help to set up and fill
__global__ __launch_bounds__(FLM_DC_THREADS) void distance_chain_kernel(
    const uint16_t* __restrict__ root_all, const uint16_t* __restrict__ steps_all, const uint32_t* __restrict__ edges_all,
    const int32_t* __restrict__ n_anchor_all, const int32_t* __restrict__ n_cells_all,
    const int32_t* __restrict__ cell_idx_all, const int32_t* __restrict__ slot_target,
    const int32_t* __restrict__ n_slots, int A, int HW, int capCells, int capAnch, int ldsAnch,
    uint16_t* __restrict__ dist_all, int32_t* __restrict__ counts_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
  uint4* ed = reinterpret_cast<uint4*>(smem);                                  // [ldsAnch] out-edges
  unsigned* rows = reinterpret_cast<unsigned*>(ed + ldsAnch);                  // [FLM_DC_GROUPS][ldsAnch][2]
  const uint16_t* lab = reinterpret_cast<const uint16_t*>(rows);               // the same labels, one by one
  const uint16_t* root = root_all + (size_t)e * capS;
  const uint16_t* steps = steps_all + (size_t)e * capS;

  for (int a = tid; a < NA; a += T) {
    uint4 ev = reinterpret_cast<const uint4*>(edges_all + (size_t)e * capAnch * 4)[a];
    if (ev.x == FLM_NONE32) ev.x = (unsigned)a;      // a missing edge becomes a zero-weight loop
    if (ev.y == FLM_NONE32) ev.y = (unsigned)a;
    if (ev.z == FLM_NONE32) ev.z = (unsigned)a;
    if (ev.w == FLM_NONE32) ev.w = (unsigned)a;
    ed[a] = ev;
  }
  for (int i = tid; i < FLM_DC_GROUPS * ldsAnch * 2; i += T) rows[i] = 0xFFFFFFFFu;
  __syncthreads();
  // sources: the four states of each target cell are anchors (root = own anchor number) at distance 0
  if (tid < 4 * nslot) {
    const int j = tid >> 2, h = tid & 3;
    const int tcell = slot_target[(size_t)e * A + base_slot + j];
    if (tcell >= 0 && tcell < HW) {
      const int ci = cell_idx_all[(size_t)e * HW + tcell];
      if (ci >= 0) {
        const unsigned a = root[ci * 4 + h];
        if (a != FLM_NONE16)
          reinterpret_cast<uint16_t*>(rows)[((size_t)(j / FLM_DC_TPG) * ldsAnch + a) * FLM_DC_TPG + (j % FLM_DC_TPG)] = 0;
      }
    }
  }
  __syncthreads();
  {
    const int w = tid >> 6, lane = tid & 63;
    if (w * FLM_DC_TPG < nslot) anchor_sweep<FLM_DC_TPG / 2>(rows + (size_t)w * ldsAnch * (FLM_DC_TPG / 2), ed, NA, lane);
  }
  __syncthreads();
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
      const uint16_t* col = lab + (size_t)(jj / FLM_DC_TPG) * ldsAnch * FLM_DC_TPG + (jj % FLM_DC_TPG);
      unsigned d[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        unsigned v = FLM_NONE16;
        if (an[i] != FLM_NONE16 && s0 + i < S) {
          const unsigned d0 = col[an[i] * FLM_DC_TPG];
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
                               uint16_t* succ, uint16_t* choice, uint16_t* walk, uint32_t* move_edge, uint16_t* move_succ,
                               const int32_t* idx_cell, void* stream) {
  if (!grid || !cell_idx || !n_cells || !target_cell || !root || !steps || !edges || !n_anchor || !counts) return FLM_E_ARG;
  if ((walk || move_edge || move_succ) && !(walk && move_edge && move_succ && succ)) return FLM_E_ARG;   // all or none
  if (walk && capCells * 4 > 32768) return FLM_E_ARG;   // bit 15 of a walk entry marks an anchor
  if (E <= 0 || A <= 0 || H <= 0 || W <= 0 || capCells <= 0 || capCells * 4 >= 65535 || capAnch <= 0 || capAnch > 0xFFF0)
    return FLM_E_ARG;
  const size_t capS = (size_t)capCells * 4;
  size_t lds = capS * 6 + 16 + ((capS + 31) / 32 + 1) * 4 + 4 * (16 * 8 + 1) + 64;
  if (lds > 160 * 1024 - 512) return FLM_E_ARG;
  FLM_RAISE_LDS_ONCE(flm::chain_index_kernel);
  hipLaunchKernelGGL(flm::chain_index_kernel, dim3(E), dim3(1024), lds, (hipStream_t)stream, grid, cell_idx, n_cells,
                     target_cell, A, H, W, capCells, capAnch, root, steps, edges, n_anchor, counts, succ, choice, walk,
                     move_edge, move_succ, idx_cell, flm_stamps_for(E));
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
  if (capCells & 1) return FLM_E_ARG;     // 16-byte row stores
  size_t lds = (size_t)ldsAnch * (16 + 32 * 2);
  if (lds > 160 * 1024 - 512) return FLM_E_ARG;
  FLM_RAISE_LDS_ONCE(flm::distance_chain_kernel);
  dim3 grid((A + 31) / 32, E);
  hipLaunchKernelGGL(flm::distance_chain_kernel, grid, dim3(FLM_DC_THREADS), lds, (hipStream_t)stream, root, steps, edges, n_anchor,
                     n_cells, cell_idx, slot_target, n_slots, A, H * W, capCells, capAnch, ldsAnch, dist, counts);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
