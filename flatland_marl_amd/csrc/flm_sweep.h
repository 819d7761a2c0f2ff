// Shortest distances over the anchor graph of one environment by wave-synchronous label correction.
//
// The anchor graph (see dist.hip) has a few hundred nodes with up to four weighted out-edges each.  One
// wavefront owns a group of targets; the labels of the group sit side by side in one LDS row per anchor
// (16-bit labels, two per 32-bit word), so that a lane relaxes all targets of the group for one anchor with
// packed 16-bit arithmetic: label(a) = min(label(a), label(head of edge) + weight) over the anchor's
// out-edges.  A round sweeps all anchors in place, every lane walking its own run of consecutive anchors (a
// label lowered early in the round is already seen by what follows); the wave stops after a round in which no
// lane lowered a label.
// Labels are always lengths of real paths, so the fixed point is the unique shortest distance, whatever the
// order of the updates: the same integers a breadth-first search over the states gives.
#pragma once
#include "flm_common.h"

namespace flm {

#define FLM_NONE32 0xFFFFFFFFu

typedef unsigned short flm_us2 __attribute__((ext_vector_type(2)));

// per 16-bit half: min(a + b, 0xFFFF)
__device__ __forceinline__ unsigned pk_add_sat_u16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(flm_us2, a), __builtin_bit_cast(flm_us2, b)));
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(flm_us2, a), __builtin_bit_cast(flm_us2, b)));
}

// Orders the LDS accesses of one wave's lanes: a wave's LDS operations complete in program order, so it is
// enough to keep the compiler from moving them across this point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

template <int NW> struct LabelRow { unsigned w[NW]; };

template <int NW>
__device__ __forceinline__ LabelRow<NW> load_row(const unsigned* p) {
  static_assert(NW == 2 || NW == 4, "a row is 8 or 16 bytes");
  LabelRow<NW> r;
  if constexpr (NW == 4) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w;
  } else {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    r.w[0] = v.x; r.w[1] = v.y;
  }
  return r;
}

template <int NW>
__device__ __forceinline__ void store_row(unsigned* p, const LabelRow<NW>& r) {
  if constexpr (NW == 4) *reinterpret_cast<uint4*>(p) = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
  else *reinterpret_cast<uint2*>(p) = make_uint2(r.w[0], r.w[1]);
}

// One wave relaxes the labels of one target group until nothing changes.
//   rows  LDS [NA][NW] 32-bit words: 2*NW labels per anchor (0xFFFF = unreachable), sources preset to 0
//   ed    LDS [NA] out-edges: head anchor | weight << 16; a missing edge is stored as a zero-weight loop to the
//         anchor itself (it changes nothing and keeps the lanes together); real edges come first
// Lane l owns the P = ceil(NA / 64) consecutive anchors l*P .. l*P + P - 1 and relaxes them in ascending order,
// in place, so a label lowered early in a round is seen by everything after it (anchors are numbered in state
// order: neighbours on the grid tend to be neighbours in a lane's run).  The next anchor's edges and labels
// are fetched while the current one is relaxed.  Returns the number of rounds (statistics only).
template <int NW>
__device__ __forceinline__ int anchor_sweep_chunks(unsigned* rows, const uint4* __restrict__ ed, int NA, int lane) {
  const int P = (NA + 63) >> 6;
  const int a0 = lane * P;
  int round = 0;
  for (; round <= NA; round++) {     // label correction converges in at most NA - 1 rounds; one more sees no change
    bool changed = false;
    int me = a0 < NA ? a0 : 0;
    uint4 e = ed[me];
    LabelRow<NW> cur = load_row<NW>(rows + (size_t)me * NW);
    for (int p = 0; p < P; p++) {
      const int a = a0 + p;
      const bool have = a < NA;
      // neighbours' labels, all requested before any is used
      const LabelRow<NW> nb0 = load_row<NW>(rows + (size_t)(e.x & 0xFFFFu) * NW);
      const LabelRow<NW> nb1 = load_row<NW>(rows + (size_t)(e.y & 0xFFFFu) * NW);
      const bool more2 = __ballot(e.z != (unsigned)me) != 0, more3 = __ballot(e.w != (unsigned)me) != 0;   // wave-uniform
      LabelRow<NW> nb2 = cur, nb3 = cur;
      if (more2) nb2 = load_row<NW>(rows + (size_t)(e.z & 0xFFFFu) * NW);
      if (more3) nb3 = load_row<NW>(rows + (size_t)(e.w & 0xFFFFu) * NW);
      // the next anchor of this lane
      const int men = (p + 1 < P && a + 1 < NA) ? a + 1 : 0;
      const uint4 en = ed[men];
      const LabelRow<NW> curn = load_row<NW>(rows + (size_t)men * NW);
      LabelRow<NW> best = cur;
      const unsigned w0 = __builtin_amdgcn_perm(e.x, e.x, 0x07060706u), w1 = __builtin_amdgcn_perm(e.y, e.y, 0x07060706u);
#pragma unroll
      for (int i = 0; i < NW; i++) {
        best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb0.w[i], w0));
        best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb1.w[i], w1));
      }
      if (more2) {
        const unsigned w2 = __builtin_amdgcn_perm(e.z, e.z, 0x07060706u);
#pragma unroll
        for (int i = 0; i < NW; i++) best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb2.w[i], w2));
      }
      if (more3) {
        const unsigned w3 = __builtin_amdgcn_perm(e.w, e.w, 0x07060706u);
#pragma unroll
        for (int i = 0; i < NW; i++) best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb3.w[i], w3));
      }
      unsigned diff = 0;
#pragma unroll
      for (int i = 0; i < NW; i++) diff |= best.w[i] ^ cur.w[i];
      if (have && diff) {
        store_row<NW>(rows + (size_t)a * NW, best);
        changed = true;
      }
      e = en;
      cur = curn;
      me = men;
    }
    wave_lds_sync();
    if (!__ballot(changed)) break;
  }
  return round;
}

// The same sweep with the lane's own anchors held in registers: the first two out-edges and the labels of its
// P <= PM anchors (a row is written by its owner only, so the register copy stays exact); the rare third and
// fourth edges are read from LDS where some lane of the wave has one.  Per anchor and round the LDS then serves
// the two neighbour rows and, when a label drops, one store.
template <int NW, int PM>
__device__ __forceinline__ int anchor_sweep_regs(unsigned* rows, const uint4* __restrict__ ed, int NA, int lane) {
  const int P = (NA + 63) >> 6;
  const int a0 = lane * P;
  unsigned ex[PM], ey[PM];
  LabelRow<NW> cur[PM];
  unsigned more = 0;                  // bit p: some lane's p-th anchor has a third or fourth edge (wave-uniform)
#pragma unroll
  for (int p = 0; p < PM; p++) {
    ex[p] = ey[p] = 0;
    for (int i = 0; i < NW; i++) cur[p].w[i] = 0xFFFFFFFFu;
    if (p < P) {
      const int a = a0 + p;
      const int me = a < NA ? a : 0;
      const uint4 e = ed[me];
      ex[p] = e.x;
      ey[p] = e.y;
      cur[p] = load_row<NW>(rows + (size_t)me * NW);
      if (__ballot(e.z != (unsigned)me || e.w != (unsigned)me)) more |= 1u << p;
    }
  }
  int round = 0;
  for (; round <= NA; round++) {
    bool changed = false;
#pragma unroll
    for (int p = 0; p < PM; p++) {
      if (p < P) {                    // wave-uniform
        const int a = a0 + p;
        const bool have = a < NA;
        const LabelRow<NW> nb0 = load_row<NW>(rows + (size_t)(ex[p] & 0xFFFFu) * NW);
        const LabelRow<NW> nb1 = load_row<NW>(rows + (size_t)(ey[p] & 0xFFFFu) * NW);
        const unsigned w0 = __builtin_amdgcn_perm(ex[p], ex[p], 0x07060706u), w1 = __builtin_amdgcn_perm(ey[p], ey[p], 0x07060706u);
        LabelRow<NW> best = cur[p];
#pragma unroll
        for (int i = 0; i < NW; i++) {
          best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb0.w[i], w0));
          best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb1.w[i], w1));
        }
        if ((more >> p) & 1u) {
          const uint4 e = ed[have ? a : 0];
          const LabelRow<NW> nb2 = load_row<NW>(rows + (size_t)(e.z & 0xFFFFu) * NW);
          const LabelRow<NW> nb3 = load_row<NW>(rows + (size_t)(e.w & 0xFFFFu) * NW);
          const unsigned w2 = __builtin_amdgcn_perm(e.z, e.z, 0x07060706u), w3 = __builtin_amdgcn_perm(e.w, e.w, 0x07060706u);
#pragma unroll
          for (int i = 0; i < NW; i++) {
            best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb2.w[i], w2));
            best.w[i] = pk_min_u16(best.w[i], pk_add_sat_u16(nb3.w[i], w3));
          }
        }
        unsigned diff = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) diff |= best.w[i] ^ cur[p].w[i];
        if (have && diff) {
          store_row<NW>(rows + (size_t)a * NW, best);
          cur[p] = best;
          changed = true;
        }
      }
    }
    wave_lds_sync();
    if (!__ballot(changed)) break;
  }
  return round;
}

// Picks the register-resident sweep when a lane's share of the anchors fits it (PM anchors per lane: what the
// kernel's register budget holds).
template <int NW, int PM = 10>
__device__ __forceinline__ int anchor_sweep(unsigned* rows, const uint4* __restrict__ ed, int NA, int lane) {
  const int P = (NA + 63) >> 6;
  if (P <= PM) return anchor_sweep_regs<NW, PM>(rows, ed, NA, lane);
  return anchor_sweep_chunks<NW>(rows, ed, NA, lane);
}

}  // namespace flm
