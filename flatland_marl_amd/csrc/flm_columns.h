// Column form of a rail-occupancy plane from the cell an agent is predicted to occupy at every time step,
// one wavefront per path, without a sequential pass (src/rail_observations.py:323-407 of the reference).
// Lane l owns the time steps l, l+64, ...  The two order-dependent quantities of the sequential definition
// become bit scans over wave ballots (one 64-bit mask per 64 steps):
//   * the node through which the current rail was entered = the last earlier step that left a switch
//     cell (highest set bit below the step in the "leaves a switch" masks), or the initial entry;
//   * a step on a switch cell shows the rail and sign of the next step that is on a rail
//     (lowest set bit above the step in the "on a rail" masks).
// The first error in time order is the one reported, as in the sequential definition.
#pragma once
#include "flm_common.h"

namespace flm {

#define FLM_COL_KT 8   // up to 512 time steps

// Index of the first lane of the wave whose flag is set (64 = none): a ballot and a bit scan, no lane exchange.
__device__ __forceinline__ int first_lane(bool flag) {
  const unsigned long long m = __ballot(flag);
  return m ? __ffsll((long long)m) - 1 : 64;
}

// Row and column of a cell as the per-step table stores it.
struct CellLinear {             // r*W + c
  int W;
  __device__ __forceinline__ void operator()(int cell, int& r, int& c) const { r = cell / W; c = cell - r * W; }
};
struct CellPacked {             // r << 8 | c (grids of at most 256 x 256)
  __device__ __forceinline__ void operator()(int cell, int& r, int& c) const { r = cell >> 8; c = cell & 255; }
};

// The two ends of a rail: (switch, cardinal point) of its source and of its destination.
struct RailEnds { int n0, cp0, n1, cp1; };

// Rail ends from the [capR,5] table flm_graph_build writes (global memory).
struct RailEndsGlobal {
  const int32_t* __restrict__ edge_info;
  __device__ __forceinline__ RailEnds operator()(int rail) const {
    const int32_t* ei = edge_info + rail * 5;
    return RailEnds{ei[0], ei[1], ei[2], ei[3]};
  }
};

// Rail ends from a packed copy (LDS): source switch | source point << 12 | destination switch << 14 | destination
// point << 26 (switch numbers stay below 2048, so an absent end, -1, matches no switch).
__device__ __forceinline__ unsigned pack_rail_ends(int n0, int cp0, int n1, int cp1) {
  return (unsigned)(n0 & 0xFFF) | ((unsigned)(cp0 & 3) << 12) | ((unsigned)(n1 & 0xFFF) << 14) | ((unsigned)(cp1 & 3) << 26);
}
struct RailEndsPacked {
  const unsigned* table;
  __device__ __forceinline__ RailEnds operator()(int rail) const {
    const unsigned v = table[rail];
    return RailEnds{(int)(v & 0xFFF), (int)((v >> 12) & 3), (int)((v >> 14) & 0xFFF), (int)((v >> 26) & 3)};
  }
};

// The wave's per-time-step tables are filled for ts < T1 (s_cell: cell r*W + c, s_rail / s_node: rail / switch
// in that cell or -1; CellT / IdT are the storage types, read as int), n = number of steps kept (cut after the
// first visit of the target), ends(rail) gives a rail's two ends.  Writes col[0..T1) (rail*2 + (sign > 0) at step
// ts + shift, -1 = empty) and returns the path's error code (wave-uniform).  s_dir is a scratch table of the wave.

// KTM = compile-time bound on the number of 64-step masks (T1 <= 64 * KTM).
template <int KTM, typename CellT, typename IdT, typename Ends, typename Coord>
__device__ __forceinline__ int columns_from_steps_kt(const CellT* s_cell, const IdT* s_rail, const IdT* s_node, IdT* s_dir,
                                                     int n, int T1, const Coord coord, const Ends ends, int Rp,
                                                     int shift, int lane, int16_t* __restrict__ col) {
  auto heading = [&](int c0, int c1) {          // direction_to_point(c0, c1)
    int r0, k0, r1, k1;
    coord(c0, r0, k0);
    coord(c1, r1, k1);
    return direction_to_point(r0, k0, r1, k1);
  };
  const int KT = (T1 + 63) >> 6;
  // masks over the kept steps
  unsigned long long m_rail[KTM], m_ev[KTM];
  int key_err_ts = 0x7FFFFFFF;   // first step at which leaving a cell that is neither rail nor node fails (wave-uniform)
#pragma unroll
  for (int q = 0; q < KTM; q++) {
    m_rail[q] = 0;
    m_ev[q] = 0;
    if (q < KT) {
      const int ts = lane + 64 * q;
      bool on_rail = false, ev = false, bad = false;
      if (ts < n) {
        on_rail = (int)s_rail[ts] != -1;
        ev = !on_rail && ts + 1 < n && (int)s_cell[ts] != (int)s_cell[ts + 1];
        bad = ev && s_node[ts] < 0;
      }
      m_rail[q] = __ballot(on_rail);
      m_ev[q] = __ballot(ev);
      const int fl = first_lane(bad);
      if (fl < 64 && key_err_ts == 0x7FFFFFFF) key_err_ts = 64 * q + fl;
    }
  }
  // initial entry (wave-uniform)
  int err = 0, en0 = -1, ecp0 = -1;
  if ((int)s_rail[0] != -1) {
    int i = -1;   // first kept step >= 1 that is not on a rail
#pragma unroll
    for (int q = KTM - 1; q >= 0; q--)
      if (q < KT) {
        unsigned long long valid = (n - 64 * q >= 64) ? ~0ull : ((n - 64 * q) > 0 ? ((1ull << (n - 64 * q)) - 1) : 0ull);
        unsigned long long holes = ~m_rail[q] & valid;
        if (q == 0) holes &= ~1ull;
        if (holes) i = 64 * q + __ffsll((long long)holes) - 1;
      }
    if (i < 0) err = FLM_AGENT_INDEX_ERROR;
    else {
      const int nid = s_node[i];
      if (nid < 0) err = FLM_AGENT_KEY_ERROR;
      else {
        const int c0 = s_cell[i - 1], c1 = s_cell[i];
        const int cp = (heading(c0, c1) + 2) & 3;
        const RailEnds ei = ends(s_rail[0]);
        if (nid == ei.n1 && cp == ei.cp1) { en0 = ei.n0; ecp0 = ei.cp0; }
        else if (nid == ei.n0 && cp == ei.cp0) { en0 = ei.n1; ecp0 = ei.cp1; }
      }
    }
  } else {
    const int nid = s_node[0];
    if (nid < 0) err = FLM_AGENT_KEY_ERROR;
    else if (n < 2) err = FLM_AGENT_INDEX_ERROR;
    else {
      const int c0 = s_cell[0], c1 = s_cell[1];
      en0 = nid;
      ecp0 = heading(c0, c1);
    }
  }
  // sign of every step on a rail
  int assert_err_ts = 0x7FFFFFFF;   // first step on a rail that was entered through neither of its ends (wave-uniform)
#pragma unroll
  for (int q = 0; q < KTM; q++)
    if (q < KT) {
      const int ts = lane + 64 * q;
      bool no_dir = false;
      if (ts < n && ((m_rail[q] >> lane) & 1)) {
        // last earlier step that left a switch cell
        int le = -1;
        unsigned long long below = m_ev[q] & lanemask_lt();
        if (below) le = 64 * q + 63 - __clzll((long long)below);
        else {
#pragma unroll
          for (int p = KTM - 1; p >= 0; p--)
            if (p < q && le < 0 && m_ev[p]) le = 64 * p + 63 - __clzll((long long)m_ev[p]);
        }
        int en = en0, ecp = ecp0;
        if (le >= 0) {
          const int c0 = s_cell[le], c1 = s_cell[le + 1];
          en = s_node[le];
          ecp = heading(c0, c1);
        }
        const RailEnds ei = ends(s_rail[ts]);
        int dir = 0;
        if (en == ei.n1 && ecp == ei.cp1) dir = 1;
        else if (en == ei.n0 && ecp == ei.cp0) dir = -1;
        no_dir = dir == 0;
        s_dir[ts] = dir;
      }
      const int fl = first_lane(no_dir);
      if (fl < 64 && assert_err_ts == 0x7FFFFFFF) assert_err_ts = 64 * q + fl;
    }
  if (!err) {
    const int k_ts = key_err_ts, a_ts = assert_err_ts;
    if (k_ts != 0x7FFFFFFF || a_ts != 0x7FFFFFFF) err = (k_ts < a_ts) ? FLM_AGENT_KEY_ERROR : FLM_AGENT_ASSERT_ERROR;
  }
  __builtin_amdgcn_wave_barrier();
  // output: a step on a rail shows itself, a step on a switch cell the next step that is on a rail
  for (int t = lane; t < T1; t += 64) {
    const int ts = t - shift;
    int v = -1;
    if (!err && ts >= 0 && ts < n) {
      const int q = ts >> 6, b = ts & 63;
      int src = -1;
      unsigned long long mq = 0;
#pragma unroll
      for (int p = 0; p < KTM; p++)
        if (p == q) mq = m_rail[p];
      if ((mq >> b) & 1) src = ts;
      else {
        unsigned long long above = (b == 63) ? 0ull : (mq & (~0ull << (b + 1)));
        if (above) src = 64 * q + __ffsll((long long)above) - 1;
        else {
#pragma unroll
          for (int p = 0; p < KTM; p++)
            if (p > q && src < 0 && m_rail[p]) src = 64 * p + __ffsll((long long)m_rail[p]) - 1;
        }
      }
      if (src >= 0) {
        const int rail = s_rail[src];
        if (rail < Rp) v = rail * 2 + (s_dir[src] > 0 ? 1 : 0);
      }
    }
    col[t] = (int16_t)v;
  }
  return err;
}

template <typename CellT, typename IdT, typename Ends, typename Coord>
__device__ __forceinline__ int columns_from_steps(const CellT* s_cell, const IdT* s_rail, const IdT* s_node, IdT* s_dir,
                                                  int n, int T1, const Coord coord, const Ends ends, int Rp, int shift,
                                                  int lane, int16_t* __restrict__ col) {
  // horizons of up to 192 steps (the usual depth of 150) take the version with three masks
  if (T1 <= 64 * 3) return columns_from_steps_kt<3>(s_cell, s_rail, s_node, s_dir, n, T1, coord, ends, Rp, shift, lane, col);
  return columns_from_steps_kt<FLM_COL_KT>(s_cell, s_rail, s_node, s_dir, n, T1, coord, ends, Rp, shift, lane, col);
}

// Path element shown at time step ts when an element lasts tpc steps: ts / tpc without an integer division
// (ts < 2^16; the quotient of (ts + 0.5) / tpc sits at least 0.5 / tpc away from an integer).
__device__ __forceinline__ int element_of_step(int ts, int tpc, float inv_tpc) {
  return tpc > 65535 ? 0 : (int)(((float)ts + 0.5f) * inv_tpc);
}

}  // namespace flm
