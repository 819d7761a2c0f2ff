// Shortest paths and occupancy columns of every agent straight from the chain index, without the dense
// distance map: what env.reset() makes get_many compute (src/rail_observations.py:78-95 through
// src/predictions.py:36-72 and src/utils/shortest_path.py:11-90 of the reference).  Two launches, one workgroup
// per environment each, everything inside a launch in LDS:
//
// anchor_decisions_kernel
//   1. anchor distances: label-correcting sweeps over the environment's anchor graph (flm_sweep.h), one wave
//      per group of four targets; the dense [target, state] map of flm_distance_chains is never written;
//   2. decisions: for every (anchor, target) the move a greedy walk takes there - the first of left, forward,
//      right among the moves of least distance (src/utils/shortest_path.py:59-65) - two bits each, and per
//      agent whether its start state reaches its target at all.  Only these leave the kernel.
//
// walk_columns_kernel
//   3. hops: one lane per agent goes from anchor to anchor (decision -> edge -> next anchor) and notes the
//      state each anchor contributes to the path and the first state of every forced run in between;
//   4. runs: all threads fill the forced runs in parallel by following the per-state successor table;
//   5. paths (optional) leave as coalesced rows in the encoding of flm_shortest_paths;
//   6. columns: one wave per agent expands its states to the predicted cell per time step
//      (src/predictions.py:85-143) and runs the scans of flm_columns.h.
#include <stdlib.h>

#include "flm_columns.h"
#include "flm_common.h"
#include "flm_sweep.h"

namespace flm {

#define FLM_NONE16 0xFFFFu
#define FLM_TPG 4                         // targets per sweeping wave
#define FLM_WC_THREADS 768                // walk_columns_kernel

__host__ __device__ inline unsigned align16(unsigned x) { return (x + 15u) & ~15u; }

// ---------------------------------------------------------------------------------------------------------
// LDS of anchor_decisions_kernel: out-edges, then one 8-byte label row per (group of four targets, anchor).
__host__ __device__ inline unsigned decisions_lds_bytes(int ldsAnch, int ldsSlots) {
  return align16(16u * ldsAnch) + align16(2u * (unsigned)ldsAnch * (unsigned)ldsSlots);
}

// PM: anchors per lane the sweep keeps in registers; WPS: waves per SIMD the register budget is cut for.  Tables up
// to 80 KB leave room for two workgroups per CU (PM 10 in 80 registers); larger ones run alone on their CU and
// get 128 registers, enough for 13 anchors per lane (832 anchors).
template <int PM, int WPS>
__global__ __launch_bounds__(1024, WPS) void anchor_decisions_kernel(
    const uint16_t* __restrict__ root_all, const uint32_t* __restrict__ edges_all, const uint32_t* __restrict__ move_edge_all,
    const int32_t* __restrict__ n_anchor_all, const int32_t* __restrict__ cell_idx_all,
    const int32_t* __restrict__ tslot_all, const int32_t* __restrict__ n_slots_all, const int32_t* __restrict__ slot_target_all,
    const int32_t* __restrict__ start_cell_all, const int32_t* __restrict__ start_dir_all, int A, int HW, int capCells,
    int capAnch, int ldsAnch, int ldsSlots, uint16_t* __restrict__ bits_all, int32_t* __restrict__ reach_all,
    int32_t* __restrict__ counts_all, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = FLM_TPG / 2;
  const int e = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63;
  auto stamp = [&](int i) {   // diagnostic only: shader-clock stamps to a buffer nothing else reads
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  const int capS = capCells * 4;
  const int NA = n_anchor_all[e];
  const int ns = n_slots_all[e];
  if (NA > ldsAnch || ns > ldsSlots) {   // tables larger than the LDS carve: flag, leave this environment's outputs unwritten
    if (tid == 0) atomicOr(&counts_all[e * 4 + 3], FLM_GRAPH_ERR_STATES);
    return;
  }
  uint4* ed = reinterpret_cast<uint4*>(smem);
  unsigned* rows = reinterpret_cast<unsigned*>(smem + align16(16u * ldsAnch));
  uint16_t* lab = reinterpret_cast<uint16_t*>(rows);
  const int NG8 = ldsSlots / 8;
  const int NGRP = (ns + FLM_TPG - 1) / FLM_TPG;
  const size_t ea = (size_t)e * A;
  const int32_t* cell_idx = cell_idx_all + (size_t)e * HW;
  const uint16_t* root = root_all + (size_t)e * capS;
  auto label_at = [&](int a, int j) -> unsigned {
    return lab[((size_t)(j / FLM_TPG) * ldsAnch + a) * FLM_TPG + (j % FLM_TPG)];
  };

  for (int a = tid; a < NA; a += T) {
    uint4 ev = reinterpret_cast<const uint4*>(edges_all + (size_t)e * capAnch * 4)[a];
    // a missing edge becomes a zero-weight loop: it changes nothing and keeps the lanes of a sweep together
    if (ev.x == FLM_NONE32) ev.x = (unsigned)a;
    if (ev.y == FLM_NONE32) ev.y = (unsigned)a;
    if (ev.z == FLM_NONE32) ev.z = (unsigned)a;
    if (ev.w == FLM_NONE32) ev.w = (unsigned)a;
    ed[a] = ev;
  }
  for (int i = tid; i < NGRP * ldsAnch * NW; i += T) rows[i] = 0xFFFFFFFFu;
  __syncthreads();
  for (int i = tid; i < 4 * ns; i += T) {         // sources: the four states of each target cell, distance 0
    const int j = i >> 2, h = i & 3;
    const int tcell = slot_target_all[ea + j];
    if (tcell >= 0 && tcell < HW) {
      const int ci = cell_idx[tcell];
      if (ci >= 0) {
        const unsigned a = root[ci * 4 + h];
        if (a != FLM_NONE16 && (int)a < NA) lab[((size_t)(j / FLM_TPG) * ldsAnch + a) * FLM_TPG + (j % FLM_TPG)] = 0;
      }
    }
  }
  __syncthreads();
  stamp(1);
  for (int g = wave; g < NGRP; g += T / 64) {
    const int rounds = anchor_sweep<NW, PM>(rows + (size_t)g * ldsAnch * NW, ed, NA, lane);
    if (stamps && g == 0 && lane == 0) stamps[(size_t)blockIdx.x * 16 + 15] = (unsigned long long)rounds;
  }
  __syncthreads();
  stamp(2);

  // decisions per (anchor, 8 targets): the labels of a move's successor for four targets are one 8-byte row
  static_assert(FLM_TPG == 4, "the decision phase unpacks rows of four labels");
  uint16_t* bits = bits_all + (size_t)e * ldsAnch * NG8;
  for (int it = tid; it < NA * NG8; it += T) {
    const int a = it / NG8, g8 = it - a * NG8;
    const uint4 me = reinterpret_cast<const uint4*>(move_edge_all + (size_t)e * capAnch * 4)[a];
    const unsigned mv[3] = {me.x, me.y, me.z};
    unsigned lab4[2][3][2];   // [half of the eight targets][move][word]: 0xFFFF where there is no such move or group
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      const int g = g8 * 2 + hf;
#pragma unroll
      for (int i = 0; i < 3; i++) {
        uint2 v = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        if (mv[i] != FLM_NONE32 && g < NGRP) v = *reinterpret_cast<const uint2*>(rows + ((size_t)g * ldsAnch + (mv[i] & 0xFFFFu)) * NW);
        lab4[hf][i][0] = v.x;
        lab4[hf][i][1] = v.y;
      }
    }
    unsigned out = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int j = g8 * 8 + t;
      unsigned sel = 3u, best = 0u;
      if (j < ns) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
          if (mv[i] == FLM_NONE32) continue;
          const unsigned d = (lab4[t >> 2][i][(t >> 1) & 1] >> (16 * (t & 1))) & 0xFFFFu;
          if (d == FLM_NONE16) continue;
          const unsigned cand = d + (mv[i] >> 16);       // distance of the move's successor + 1
          if (sel == 3u || cand < best) {                 // strictly smaller wins: ties go to the earlier move
            best = cand;
            sel = (unsigned)i;
          }
        }
      }
      out |= sel << (2 * t);
    }
    bits[it] = (uint16_t)out;
  }
  // does the agent's start state reach its target?
  for (int a = tid; a < A; a += T) {
    const size_t g = ea + a;
    const int c0 = start_cell_all[g], h0 = start_dir_all[g] & 3, j = tslot_all[g];
    int ok = 0;
    if (c0 >= 0 && c0 < HW && j >= 0 && j < ns) {
      const int ci = cell_idx[c0];
      if (ci >= 0) {
        const unsigned r0 = root[ci * 4 + h0];
        if (r0 != FLM_NONE16 && (int)r0 < NA) ok = label_at((int)r0, j) != FLM_NONE16;
      }
    }
    reach_all[g] = ok;
  }
  stamp(3);
}

// ---------------------------------------------------------------------------------------------------------
struct WalkLayout {
  unsigned off_bits;     // uint16 [ldsAnch][NG8]
  unsigned off_medge;    // uint4 [ldsAnch]: the moves' edges (anchor | weight << 16), fourth word = flag "reported as forward"
  unsigned off_msucc;    // uint2 [ldsAnch]: the states the moves lead to
  unsigned off_astate;   // uint16 [ldsAnch]: the state of each anchor
  unsigned off_cinfo;    // unsigned [capCells]: row << 8 | column | (rail, 0x8000 | switch, 0xFFFF) << 16
  unsigned off_rinfo;    // unsigned [capR]: packed rail ends
  unsigned off_agent;    // int [13][A]: per-agent scalars (see AG_* below)
  unsigned off_rec;      // uint16 [A][Lr]: the state of every path element, Lr = Lcap + 1 rounded up to even
  unsigned off_walk;     // uint16 [capS]; the per-wave column scratch reuses these bytes afterwards
  unsigned Lr;
  unsigned total;
};

__host__ __device__ inline WalkLayout walk_layout(int A, int capCells, int ldsAnch, int ldsSlots, int capR, int Lcap, int T1) {
  WalkLayout L;
  unsigned o = 0;
  L.Lr = ((unsigned)Lcap + 2u) & ~1u;
  L.off_bits = o; o = align16(o + 2u * ldsAnch * ((unsigned)ldsSlots / 8u));
  L.off_medge = o; o = align16(o + 16u * ldsAnch);
  L.off_msucc = o; o = align16(o + 8u * ldsAnch);
  L.off_astate = o; o = align16(o + 2u * ldsAnch);
  L.off_cinfo = o; o = align16(o + 4u * capCells);
  L.off_rinfo = o; o = align16(o + 4u * capR);
  L.off_agent = o; o = align16(o + 4u * 13u * A);
  L.off_rec = o; o = align16(o + 2u * A * L.Lr);
  L.off_walk = o;
  const unsigned walk_bytes = 2u * 4u * capCells;
  const unsigned scratch_bytes = (FLM_WC_THREADS / 64) * align16(8u * T1);   // per wave: cell, rail, switch, sign (16 bits each)
  o = align16(o + (walk_bytes > scratch_bytes ? walk_bytes : scratch_bytes));
  L.total = o;
  return L;
}

__global__ __launch_bounds__(FLM_WC_THREADS) void walk_columns_kernel(
    const uint16_t* __restrict__ walk_all, const uint16_t* __restrict__ root_all, const uint16_t* __restrict__ steps_all,
    const uint32_t* __restrict__ move_edge_all, const uint16_t* __restrict__ move_succ_all,
    const uint16_t* __restrict__ bits_all, const int32_t* __restrict__ reach_all, const int32_t* __restrict__ n_anchor_all,
    const int32_t* __restrict__ n_cells_all, const int32_t* __restrict__ n_slots_all,
    const int32_t* __restrict__ cell_idx_all, const int32_t* __restrict__ idx_cell_all,
    const int32_t* __restrict__ cell_rail_all,
    const int32_t* __restrict__ node_id_all, const int32_t* __restrict__ edge_info_all, const int32_t* __restrict__ graph_counts,
    const int32_t* __restrict__ tslot_all, const int32_t* __restrict__ start_cell_all, const int32_t* __restrict__ start_dir_all,
    const int32_t* __restrict__ pos_cell_all, const int32_t* __restrict__ target_cell_all,
    const int32_t* __restrict__ status_all, const int32_t* __restrict__ tpc_all, int A, int H, int W, int capCells,
    int capAnch, int ldsAnch, int ldsSlots, int capR, int Rp, int T1, int max_depth, int Lcap, int shift,
    int32_t* __restrict__ path_cell_all, uint32_t* __restrict__ path_meta_all, int32_t* __restrict__ path_len_all,
    int16_t* __restrict__ cols_all, int32_t* __restrict__ err_all, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int e = blockIdx.x, tid = threadIdx.x, T = FLM_WC_THREADS;
  const int wave = tid >> 6, lane = tid & 63;
  auto stamp = [&](int i) {
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(4);
  const int HW = H * W;
  const int capS = capCells * 4;
  const int NA = n_anchor_all[e];
  const int nC = n_cells_all[e];
  if (NA > ldsAnch || n_slots_all[e] > ldsSlots) return;   // flagged by anchor_decisions_kernel
  const WalkLayout L = walk_layout(A, capCells, ldsAnch, ldsSlots, capR, Lcap, T1);
  uint16_t* bits = reinterpret_cast<uint16_t*>(smem + L.off_bits);
  uint4* medge = reinterpret_cast<uint4*>(smem + L.off_medge);
  uint2* msucc = reinterpret_cast<uint2*>(smem + L.off_msucc);
  uint16_t* astate = reinterpret_cast<uint16_t*>(smem + L.off_astate);
  unsigned* cinfo = reinterpret_cast<unsigned*>(smem + L.off_cinfo);
  unsigned* rinfo = reinterpret_cast<unsigned*>(smem + L.off_rinfo);
  // per-agent scalars, each an int [A]
  int* ag = reinterpret_cast<int*>(smem + L.off_agent);
  enum { AG_LEN, AG_LIM, AG_TOT, AG_START, AG_DIR, AG_TCI, AG_CI0, AG_SLOT, AG_POS, AG_POSID, AG_TGT, AG_TPC, AG_TGTRC };
  int* plen = ag + AG_LEN * A;       // path length (-1 = none)
  int* plim = ag + AG_LIM * A;       // last index that holds a state
  int* ptot = ag + AG_TOT * A;       // total moves up to the target or the horizon
  uint16_t* rec = reinterpret_cast<uint16_t*>(smem + L.off_rec);
  uint16_t* walk = reinterpret_cast<uint16_t*>(smem + L.off_walk);
  const int Lr = (int)L.Lr;
  const int NG8 = ldsSlots / 8;
  const size_t ea = (size_t)e * A;
  const int32_t* cell_idx = cell_idx_all + (size_t)e * HW;
  const int nR = min(graph_counts[e * 4 + 1], capR);

  // ---- tables.  The look-ups with a dependent second step (cell -> what stands in it, agent -> its cells) are
  // issued first and consumed last, so that their two round trips to memory overlap the plain copies in between.
  auto pack_rc = [&](int cell) -> unsigned {   // row << 8 | column (H, W <= 256); a cell off the grid gives 0xFFFF
    if (cell < 0 || cell >= HW) return 0xFFFFu;
    const int r = cell / W;
    return (unsigned)((r << 8) | (cell - r * W));
  };
  const int32_t* cell_rail = cell_rail_all + (size_t)e * HW;
  const int32_t* node_id = node_id_all + (size_t)e * HW;
  auto id_of = [](int rl, int nd) -> unsigned {   // what stands in a cell: a rail, 0x8000 | switch, or 0xFFFF
    return rl >= 0 ? (unsigned)rl : (nd >= 0 ? (0x8000u | (unsigned)nd) : 0xFFFFu);
  };
  constexpr int CU = 4;                      // cells per thread and batch
  int c_cell[CU], c_rl[CU], c_nd[CU];
  const int lastC = nC > 0 ? nC - 1 : 0;
#pragma unroll
  for (int u = 0; u < CU; u++) {
    c_cell[u] = idx_cell_all[(size_t)e * capCells + min(u * T + tid, lastC)];
    if (nC <= 0 || c_cell[u] < 0 || c_cell[u] >= HW) c_cell[u] = 0;
  }
  // agent scalars (first step): thread a of the first waves
  const int aa = tid < A ? tid : 0;
  const size_t ga = ea + aa;
  const int a_c0 = start_cell_all[ga], a_t = target_cell_all[ga], a_pos = pos_cell_all[ga], a_tpc = tpc_all[ga];
  const int a_dir = start_dir_all[ga], a_slot = tslot_all[ga], a_status = status_all[ga], a_reach = reach_all[ga];
#pragma unroll
  for (int u = 0; u < CU; u++) {
    c_rl[u] = cell_rail[c_cell[u]];
    c_nd[u] = node_id[c_cell[u]];
  }
  const int a_posc = (a_pos >= 0 && a_pos < HW) ? a_pos : 0;
  const int a_tci = cell_idx[(a_t >= 0 && a_t < HW) ? a_t : 0], a_ci0 = cell_idx[(a_c0 >= 0 && a_c0 < HW) ? a_c0 : 0];
  const int a_prl = cell_rail[a_posc], a_pnd = node_id[a_posc];

  for (int i = tid; i < NA * NG8; i += T) bits[i] = bits_all[(size_t)e * ldsAnch * NG8 + i];
  for (int a = tid; a < NA; a += T) {
    medge[a] = reinterpret_cast<const uint4*>(move_edge_all + (size_t)e * capAnch * 4)[a];
    msucc[a] = reinterpret_cast<const uint2*>(move_succ_all + (size_t)e * capAnch * 4)[a];
  }
  for (int s = tid; s < nC * 4; s += T) {
    const unsigned wv = walk_all[(size_t)e * capS + s];
    walk[s] = (uint16_t)wv;
    if ((wv & 0x8000u) && wv != FLM_NONE16 && (int)(wv & 0x7FFFu) < NA) astate[wv & 0x7FFFu] = (uint16_t)s;
  }
  for (int r = tid; r < nR; r += T) {
    const int32_t* ei = edge_info_all + ((size_t)e * capR + r) * 5;
    rinfo[r] = pack_rail_ends(ei[0], ei[1], ei[2], ei[3]);
  }
#pragma unroll
  for (int u = 0; u < CU; u++) {
    const int ci = u * T + tid;
    if (ci < nC) cinfo[ci] = pack_rc(c_cell[u]) | (id_of(c_rl[u], c_nd[u]) << 16);
  }
  for (int base = CU * T; base < nC; base += T) {     // maps with more rail cells than CU per thread
    const int ci = base + tid;
    if (ci < nC) {
      const int cell = idx_cell_all[(size_t)e * capCells + ci];
      const bool ok = cell >= 0 && cell < HW;
      cinfo[ci] = pack_rc(cell) | ((ok ? id_of(cell_rail[cell], node_id[cell]) : 0xFFFFu) << 16);
    }
  }
  if (tid < A) {
    const int a = tid;
    ag[AG_START * A + a] = a_c0;
    ag[AG_DIR * A + a] = a_dir & 3;
    ag[AG_TCI * A + a] = (a_t >= 0 && a_t < HW) ? a_tci : -1;
    ag[AG_CI0 * A + a] = (a_c0 >= 0 && a_c0 < HW) ? a_ci0 : -1;
    ag[AG_SLOT * A + a] = a_slot;
    ag[AG_POS * A + a] = (int)pack_rc(a_pos);
    ag[AG_POSID * A + a] = (a_pos >= 0 && a_pos < HW) ? (int)id_of(a_prl, a_pnd) : 0xFFFF;
    ag[AG_TGT * A + a] = a_t;
    ag[AG_TGTRC * A + a] = (a_t >= 0 && a_t < HW) ? (int)pack_rc(a_t) : -1;
    ag[AG_TPC * A + a] = a_tpc < 1 ? 1 : a_tpc;
    // status 3 (DONE_REMOVED) has no path; an unreachable target neither
    ag[AG_LEN * A + a] = (a_status != 3 ? 1 : 0) | (a_reach ? 2 : 0);
  }
  for (int i = tid; i < A * Lr / 2; i += T) reinterpret_cast<unsigned*>(rec)[i] = 0xFFFFFFFFu;
  __syncthreads();
  stamp(5);

  // ---- hops, one lane per agent
  for (int a = tid; a < A; a += T) {
    const size_t g = ea + a;
    uint16_t* my = rec + (size_t)a * Lr;
    int len = -1, lim = -1, total = 0;
    const int flags = plen[a];
    if (flags & 1) {                      // not DONE_REMOVED
      const int c0 = ag[AG_START * A + a], h0 = ag[AG_DIR * A + a], tcell = ag[AG_TGT * A + a], j = ag[AG_SLOT * A + a];
      const int tci = ag[AG_TCI * A + a], ci0 = ag[AG_CI0 * A + a];
      if (c0 == tcell) {                  // already there: only the closing element
        len = max_depth > 0 ? 1 : 0;
        lim = 0;
        my[0] = ci0 >= 0 ? (uint16_t)(ci0 * 4 + h0) : (uint16_t)0;
      } else if (ci0 >= 0 && (flags & 2)) {
        const unsigned s0 = (unsigned)(ci0 * 4 + h0);
        const unsigned w0 = walk[s0];
        int k = 0;
        unsigned an;
        bool fail = false;
        my[0] = (uint16_t)s0;
        if (!(w0 & 0x8000u)) {            // forced start: the run up to its anchor is filled below
          k = steps_all[(size_t)e * capS + s0];
          an = root_all[(size_t)e * capS + s0];
        } else {
          an = w0 & 0x7FFFu;
        }
        // the path's element k is the state of anchor an
        while (true) {
          const unsigned sa = astate[an];
          if (k <= max_depth) my[k] = (uint16_t)sa;
          if (k >= max_depth || (int)(sa >> 2) == tci) break;
          const unsigned mv = (bits[an * NG8 + (j >> 3)] >> (2 * (j & 7))) & 3u;
          if (mv == 3u) { fail = true; break; }
          const uint4 me = medge[an];
          const uint2 ms = msucc[an];
          const unsigned ev = mv == 0 ? me.x : (mv == 1 ? me.y : me.z);
          const unsigned nxt = mv == 0 ? (ms.x & 0xFFFFu) : (mv == 1 ? (ms.x >> 16) : (ms.y & 0xFFFFu));
          if (ev == FLM_NONE32) { fail = true; break; }
          const int w = (int)(ev >> 16);
          if (w > 1 && k + 1 <= max_depth) my[k + 1] = (uint16_t)nxt;   // first state of a forced run
          k += w;
          an = ev & 0xFFFFu;
        }
        if (!fail) {
          total = k;
          len = k < max_depth ? k + 1 : max_depth;   // the closing element when the target comes first
          lim = min(k, max_depth);
        }
      }
    }
    plen[a] = len;
    plim[a] = lim;
    ptot[a] = total;
    path_len_all[g] = len;
  }
  __syncthreads();
  stamp(6);

  // ---- forced runs: every entry noted above that is a forced state and has an empty successor is the head of
  // a run; all heads are found first (nothing is written meanwhile), then every thread fills the runs it found
  {
    constexpr int PER = 16;               // entries per thread and pass
    const int n_ent = A * Lr;
    for (int base = 0; base < n_ent; base += PER * T) {
      unsigned heads = 0;                 // bit q: entry base + q*T + tid is a head
#pragma unroll
      for (int q = 0; q < PER; q++) {
        const int idx = base + q * T + tid;
        if (idx < n_ent) {
          const int a = idx / Lr, i = idx - a * Lr;
          const unsigned s = rec[idx];
          if (s != FLM_NONE16 && i + 1 <= plim[a] && rec[idx + 1] == FLM_NONE16 && !(walk[s] & 0x8000u)) heads |= 1u << q;
        }
      }
      __syncthreads();
      while (heads) {
        const int q = __ffs(heads) - 1;
        heads &= heads - 1;
        int idx = base + q * T + tid;
        const int a = idx / Lr;
        const int last = a * Lr + plim[a];
        unsigned s = rec[idx];
        while (idx + 1 <= last && rec[idx + 1] == FLM_NONE16) {
          s = walk[s];
          rec[++idx] = (uint16_t)s;
        }
      }
      __syncthreads();
    }
  }
  stamp(7);

  // ---- the paths leave in the encoding of flm_shortest_paths
  if (path_cell_all) {
    for (int a = wave; a < A; a += T / 64) {
      const size_t g = ea + a;
      const int len = min(plen[a], Lcap);
      const bool closed = ptot[a] < max_depth;
      const int c0 = ag[AG_START * A + a], h0 = ag[AG_DIR * A + a], j = ag[AG_SLOT * A + a];
      for (int k = lane; k < len; k += 64) {
        const unsigned s = rec[(size_t)a * Lr + k];
        const unsigned rc = cinfo[s >> 2] & 0xFFFFu;
        const int cell = k == 0 ? c0 : (int)((rc >> 8) * W + (rc & 255u));
        const unsigned h = k == 0 ? (unsigned)h0 : (s & 3u);
        unsigned act = 2u, nh = h;
        if (closed && k == len - 1) {
          act = 4u;
        } else {
          nh = rec[(size_t)a * Lr + k + 1] & 3u;
          const unsigned wv = walk[s];
          if (wv & 0x8000u) {              // an anchor: the move it took, unless reported as forward
            const unsigned an = wv & 0x7FFFu;
            const unsigned mv = (bits[an * NG8 + (j >> 3)] >> (2 * (j & 7))) & 3u;
            act = (medge[an].w & 1u) ? 2u : 1u + mv;
          }
        }
        path_cell_all[g * Lcap + k] = cell;
        path_meta_all[g * Lcap + k] = h | (act << 8) | (nh << 16);
      }
    }
  }
  __syncthreads();
  stamp(8);

  // ---- columns, one wave per agent (the scratch tables reuse the successor table's bytes)
  {
    const unsigned per_wave = align16(8u * T1);
    uint16_t* s_cell = reinterpret_cast<uint16_t*>(smem + L.off_walk + (size_t)wave * per_wave);
    int16_t* s_rail = reinterpret_cast<int16_t*>(s_cell + T1);
    int16_t* s_node = s_rail + T1;
    int16_t* s_dir = s_node + T1;
    const int KT = (T1 + 63) >> 6;
    for (int a = wave; a < A; a += T / 64) {
      const size_t g = ea + a;
      // cells are compared and stored as row << 8 | column here
      const int pos = ag[AG_POS * A + a], tgt = ag[AG_TGTRC * A + a], tpc = ag[AG_TPC * A + a];
      const float inv_tpc = 1.0f / (float)tpc;
      int len = plen[a];
      if (len > Lcap) len = Lcap;
      const uint16_t* my = rec + (size_t)a * Lr;
      const unsigned pos_id = (unsigned)ag[AG_POSID * A + a];   // what stands in the agent's own cell (path element 0)
      // the agent stops advancing at the first path element on the target cell
      int kstop = len > 0 ? len - 1 : 0;
      for (int base = 0; base < len; base += 64) {
        const int k = base + lane;
        const int fl = first_lane(k < len && ((k == 0) ? pos : (int)(cinfo[my[k] >> 2] & 0xFFFFu)) == tgt);
        if (fl < 64) {
          kstop = base + fl;
          break;
        }
      }
      int n = T1;                   // steps kept: cut after the first visit of the target
      for (int q = 0; q < KT; q++) {
        const int ts = lane + 64 * q;
        bool at_target = false;
        if (ts < T1) {
          int k = element_of_step(ts, tpc, inv_tpc);
          if (k > kstop) k = kstop;
          int c = pos;
          unsigned id = pos_id;
          if (k != 0) {
            const unsigned ci = cinfo[my[k] >> 2];
            c = (int)(ci & 0xFFFFu);
            id = ci >> 16;
          }
          s_cell[ts] = (uint16_t)c;
          s_rail[ts] = (id & 0x8000u) ? (int16_t)-1 : (int16_t)id;
          s_node[ts] = (id & 0x8000u) ? (id == 0xFFFFu ? (int16_t)-1 : (int16_t)(id & 0x7FFFu)) : (int16_t)-1;
          at_target = c == tgt;
        }
        const int fl = first_lane(at_target);
        if (fl < 64 && n == T1) n = 64 * q + fl + 1;
      }
      __builtin_amdgcn_wave_barrier();
      const int err = columns_from_steps<uint16_t, int16_t>(s_cell, s_rail, s_node, s_dir, n, T1, CellPacked{}, RailEndsPacked{rinfo}, Rp,
                                                           shift, lane, cols_all + g * T1);
      if (lane == 0) err_all[g] = err;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (stamps) {
    __syncthreads();
    stamp(9);
  }
}

}  // namespace flm

// Diagnostic hook (not part of the C ABI in include/): when set, every launch of flm_paths_columns, flm_graph_build
// and flm_chain_index writes sixteen 64-bit words per environment to its third of the buffer ([3][E][16]): shader-clock
// stamps at the phase boundaries and the rounds of one sweep.
// The caller states how many environments the buffer was sized for; a launch over more environments gets no stamps.
unsigned long long* g_flm_stamps = nullptr;
int g_flm_stamps_envs = 0;
extern "C" void flm_debug_fused_stamps(void* device_buffer, int capacity_envs) {
  g_flm_stamps = static_cast<unsigned long long*>(device_buffer);
  g_flm_stamps_envs = device_buffer ? capacity_envs : 0;
}

extern "C" long flm_paths_columns_lds_bytes(int A, int capCells, int ldsAnch, int ldsSlots, int capR, int Lcap, int T1) {
  if (A <= 0 || capCells <= 0 || ldsAnch <= 0 || ldsSlots <= 0 || capR <= 0 || Lcap <= 0 || T1 <= 0) return 0;
  const long a = (long)flm::decisions_lds_bytes(ldsAnch, ldsSlots);
  const long b = (long)flm::walk_layout(A, capCells, ldsAnch, ldsSlots, capR, Lcap, T1).total;
  return a > b ? a : b;
}

extern "C" long flm_paths_columns_workspace(int E, int A, int ldsAnch, int ldsSlots) {
  if (E <= 0 || A <= 0 || ldsAnch <= 0 || ldsSlots <= 0) return 0;
  // decisions: 2 bytes per (anchor, 8 targets), rounded up to whole int32 words per launch; reach flags: one int32 per agent
  return ((long)E * ldsAnch * (ldsSlots / 8) + 1) / 2 + (long)E * A;
}

extern "C" int flm_paths_columns(const uint16_t* walk, const uint16_t* root, const uint16_t* steps, const uint32_t* edges,
                                 const uint32_t* move_edge, const uint16_t* move_succ, const int32_t* n_anchor,
                                 const int32_t* n_cells, const int32_t* cell_idx, const int32_t* idx_cell,
                                 const int32_t* cell_rail, const int32_t* node_id, const int32_t* edge_info, const int32_t* tslot,
                                 const int32_t* n_slots, const int32_t* slot_target, const int32_t* start_cell,
                                 const int32_t* start_dir, const int32_t* pos_cell, const int32_t* target_cell,
                                 const int32_t* status, const int32_t* tpc, int E, int A, int H, int W, int capCells,
                                 int capAnch, int ldsAnch, int ldsSlots, int capR, int Rp, int T1, int max_depth, int Lcap,
                                 int shift, int32_t* workspace, int32_t* path_cell, uint32_t* path_meta, int32_t* path_len,
                                 int16_t* cols, int32_t* err, int32_t* counts, void* stream_) {
  if (!walk || !root || !steps || !edges || !move_edge || !move_succ || !n_anchor || !n_cells || !cell_idx || !idx_cell || !cell_rail ||
      !node_id || !edge_info || !tslot || !n_slots || !slot_target || !start_cell || !start_dir || !pos_cell ||
      !target_cell || !status || !tpc || !workspace || !path_len || !cols || !err || !counts)
    return FLM_E_ARG;
  if ((path_cell == nullptr) != (path_meta == nullptr)) return FLM_E_ARG;
  if (E <= 0 || A <= 0 || A > FLM_WC_THREADS || H <= 0 || W <= 0 || H > 256 || W > 256 || capCells <= 0 || capCells * 4 > 32768 ||
      capAnch <= 0 || ldsAnch <= 0 || ldsAnch > capAnch || ldsAnch > 0x7FFE || ldsSlots <= 0 || (ldsSlots & 7) || capR <= 0 ||
      Rp <= 0 || Rp > 16383 || T1 <= 1 || T1 > 64 * FLM_COL_KT || max_depth < 0 || Lcap < max_depth + 1 || shift < 0)
    return FLM_E_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned lds1 = flm::decisions_lds_bytes(ldsAnch, ldsSlots);
  const flm::WalkLayout L = flm::walk_layout(A, capCells, ldsAnch, ldsSlots, capR, Lcap, T1);
  if (lds1 > 160 * 1024 - 512 || L.total > 160 * 1024 - 512) return FLM_E_CAPACITY;
  uint16_t* bits = reinterpret_cast<uint16_t*>(workspace);
  int32_t* reach = workspace + ((long)E * ldsAnch * (ldsSlots / 8) + 1) / 2;
  // one wave per group of four targets (at least four waves for the set-up loops, at most sixteen)
  int waves = (ldsSlots + FLM_TPG - 1) / FLM_TPG;
  waves = waves < 4 ? 4 : (waves > 16 ? 16 : waves);
  const bool alone = lds1 > 80 * 1024;   // one workgroup per CU anyway: a larger register budget costs nothing
  if (alone) FLM_RAISE_LDS_ONCE((flm::anchor_decisions_kernel<13, 4>));
  else FLM_RAISE_LDS_ONCE((flm::anchor_decisions_kernel<10, 6>));
  if (alone)
    hipLaunchKernelGGL((flm::anchor_decisions_kernel<13, 4>), dim3(E), dim3(waves * 64), lds1, stream, root, edges, move_edge,
                       n_anchor, cell_idx, tslot, n_slots, slot_target, start_cell, start_dir, A, H * W, capCells, capAnch,
                       ldsAnch, ldsSlots, bits, reach, counts, flm_stamps_for(E));
  else
    hipLaunchKernelGGL((flm::anchor_decisions_kernel<10, 6>), dim3(E), dim3(waves * 64), lds1, stream, root, edges, move_edge,
                       n_anchor, cell_idx, tslot, n_slots, slot_target, start_cell, start_dir, A, H * W, capCells, capAnch,
                       ldsAnch, ldsSlots, bits, reach, counts, flm_stamps_for(E));
  FLM_CHECK_LAUNCH();
  FLM_RAISE_LDS_ONCE(flm::walk_columns_kernel);
  hipLaunchKernelGGL(flm::walk_columns_kernel, dim3(E), dim3(FLM_WC_THREADS), L.total, stream, walk, root, steps, move_edge,
                     move_succ, bits, reach, n_anchor, n_cells, n_slots, cell_idx, idx_cell, cell_rail, node_id, edge_info, counts, tslot,
                     start_cell, start_dir, pos_cell, target_cell, status, tpc, A, H, W, capCells, capAnch, ldsAnch, ldsSlots,
                     capR, Rp, T1, max_depth, Lcap, shift, path_cell, path_meta, path_len, cols, err, flm_stamps_for(E));
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
