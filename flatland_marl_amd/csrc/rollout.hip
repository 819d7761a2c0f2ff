// Rollout-time kernels: alternative paths at a switch, and the occupancy-plane bookkeeping
// (shift, delay behind the last train on the next rail, crash checks).
#include "flm_common.h"

namespace flm {

// ---- alternative paths ------------------------------------------------------------------------------
// A group of G lanes per request (G = power of two >= cap_alt, 64 / G requests per wave).  The group's
// first lane enumerates the branching prefix (the cell the agent stands on, then every switch cell crossed
// before the first plain cell) depth-first in the order left, forward, right and writes one prefix per
// leaf; then every lane completes one leaf by a greedy descent of at most max_depth moves, so the walks of
// a request run side by side; last, leaves whose descent found no way are squeezed out in order.
// More than cap_alt leaves is FLM_AGENT_OVERFLOW.
// STAGE: one request per wave (G = 64).  The wave first works out, for every state of the environment, the move a
// greedy walk towards this request's target takes there (the forced successor, or among the moves of least distance
// the first of left, forward, right: what greedy_path() decides step by step) and keeps it in LDS as one word per
// state - successor, action, "no way" - and, beside it, the move after that one.  A descent is then a chain of
// look-ups of the same kind whatever the states are, two path elements per look-up, and the lanes of a request's
// alternatives run in step.  The descent of greedy_path() carries the distance of the state it stands on and admits
// only moves that lower it; in a distance map every state's best move does, so the table holds the same choice (the
// first element, taken before any distance is known, is handled apart).
#define FLM_ALT_STACK 32
#define FLM_ALT_FRAME_INTS (2 * (FLM_ALT_STACK + 2) + 4 * FLM_ALT_STACK + 6 * FLM_ALT_STACK)   // per request

template <bool STAGE>
__global__ __launch_bounds__(64) void altpaths_kernel(
    const uint16_t* __restrict__ grid_all, const int32_t* __restrict__ cell_idx_all,
    const uint16_t* __restrict__ dist_all, const uint16_t* __restrict__ succ_all,
    const uint16_t* __restrict__ choice_all, const int32_t* __restrict__ idx_cell_all, const int32_t* __restrict__ tslot,
    const int32_t* __restrict__ node_id_all, const int32_t* __restrict__ req_agent, int n_req,
    const int32_t* __restrict__ pos_cell, const int32_t* __restrict__ direction,
    const int32_t* __restrict__ target_cell, int A, int H, int W, int capCells, int max_depth, int cap_alt, int Lcap,
    int G, int32_t* __restrict__ alt_count, int32_t* __restrict__ alt_len, int32_t* __restrict__ alt_cell,
    uint32_t* __restrict__ alt_meta, int32_t* __restrict__ err_out, const int32_t* __restrict__ n_live, int live_base,
    int32_t* __restrict__ agent_rep) {
  __shared__ int s_cell[64], s_h[64], s_base[64], s_res[64], s_n[64], s_err[64], s_cnt[64];
  extern __shared__ __attribute__((aligned(16))) int alt_frames[];     // [64 / G][FLM_ALT_FRAME_INTS], then the staged table
  const int lane = threadIdx.x, grp = lane / G, j = lane - grp * G;
  const int g = blockIdx.x * (64 / G) + grp;
  // n_live (optional): the launch was sized before the number of requests was known to the host; the kernel that
  // counted them left it at *n_live, and this launch covers requests live_base .. live_base + n_req - 1 of them
  if (n_live) n_req = min(n_req, max(*n_live - live_base, 0));
  if (STAGE && blockIdx.x >= n_req) return;                            // (one request per workgroup)
  const bool live = g < n_req;
  const int ga = live ? req_agent[g] : 0;
  // agent_rep (optional) [n_req, cap_alt]: the request's agent once per alternative, the list flm_columns takes
  if (agent_rep && live && j < cap_alt) agent_rep[(size_t)g * cap_alt + j] = ga;
  const int e = ga / A;
  const int HW = H * W;
  const uint16_t* grid = grid_all + (size_t)e * HW;
  const int32_t* cell_idx = cell_idx_all + (size_t)e * HW;
  const int32_t* node_id = node_id_all + (size_t)e * HW;
  const uint16_t* dist = dist_all + ((size_t)e * A + tslot[ga]) * capCells * 4;
  const uint16_t* succ = succ_all ? succ_all + (size_t)e * capCells * 4 : nullptr;
  const uint16_t* choice = succ_all && choice_all ? choice_all + (size_t)e * capCells * 16 : nullptr;
  // STAGE: [capCells * 4] next move, [capCells * 4] the move after it (32 bits each), [capCells] grid cell of every rail
  // cell, then the descents' records [cap_alt][Lcap]; before the descents that last region holds a copy of the target's
  // distance row (2 bytes per state), from which - with the environment's successor and choice tables, read from
  // memory - the move tables are worked out (capCells is even: every table is whole 16-byte pieces)
  uint32_t* next_lds = reinterpret_cast<uint32_t*>(alt_frames + FLM_ALT_FRAME_INTS);
  uint32_t* next2_lds = next_lds + (size_t)capCells * 4;
  int32_t* icell_lds = reinterpret_cast<int32_t*>(next2_lds + (size_t)capCells * 4);
  uint32_t* rec_lds = reinterpret_cast<uint32_t*>(icell_lds + ((capCells + 3) & ~3));   // (on a 16-byte boundary)
  uint16_t* dist_lds = reinterpret_cast<uint16_t*>(rec_lds);
  constexpr uint32_t NO_MOVE = 0xFFFFFFFFu;
  const int tcell_s = target_cell[ga];
  const int tci_s = (STAGE && tcell_s >= 0 && tcell_s < HW) ? cell_idx[tcell_s] : -1;
  if (STAGE) {
    // (G = 64: the whole wave serves one request, e and the target are the same for every lane)
    // Only the target's distance row is looked up at random (by successor state): it is staged.  The successor table
    // is read once, front to back, and the choice table once per state that has a choice: both come straight from
    // memory, eight states per lane and pass, which keeps the request's LDS at 44 bytes per cell - five requests
    // per CU instead of two.
    const uint4* src_d = reinterpret_cast<const uint4*>(dist);
    const int32_t* src_i = idx_cell_all + (size_t)e * capCells;
    uint4* dst_d = reinterpret_cast<uint4*>(dist_lds);
    for (int i = lane; i < capCells / 2; i += 64) dst_d[i] = src_d[i];   // capCells * 8 bytes, 16 per copy
    for (int i = lane; i < capCells; i += 64) icell_lds[i] = src_i[i];
    __syncthreads();
    const int capS = capCells * 4;                             // (a multiple of 8: capCells is even)
    for (int s8 = lane * 8; s8 < capS; s8 += 64 * 8) {
      const uint4 sv = *reinterpret_cast<const uint4*>(succ + s8);
      const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w};
      unsigned nx[8];
      uint2 raw[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        nx[k] = (sw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
        raw[k] = make_uint2(0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (nx[k] == 0xFFFFu) raw[k] = *reinterpret_cast<const uint2*>(choice + (size_t)(s8 + k) * 4);
      uint32_t wv[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        uint32_t w = NO_MOVE;
        if (nx[k] != 0xFFFFu) {
          w = nx[k] | (2u << 16);                              // forced: recorded as "forward"
        } else {
          const unsigned ns[3] = {raw[k].x & 0xFFFFu, raw[k].x >> 16, raw[k].y & 0xFFFFu};
          unsigned best_d = 0xFFFFFFFFu;
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const unsigned dd = ns[i] != 0xFFFFu ? (unsigned)dist_lds[ns[i]] : (unsigned)FLM_DIST_INF;
            if (dd != FLM_DIST_INF && dd < best_d) {
              best_d = dd;
              w = ns[i] | (((raw[k].y >> 16) & 1u ? 2u : (unsigned)(1 + i)) << 16);
            }
          }
        }
        wv[k] = w;
      }
      *reinterpret_cast<uint4*>(next_lds + s8) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
      *reinterpret_cast<uint4*>(next_lds + s8 + 4) = make_uint4(wv[4], wv[5], wv[6], wv[7]);
    }
    __syncthreads();
    // the move after the next one; none where the next state is on the target cell (a walk ends there)
    for (int st = lane; st < capS; st += 64) {
      const uint32_t w = next_lds[st];
      uint32_t w2 = NO_MOVE;
      if (w != NO_MOVE && (int)((w & 0xFFFFu) >> 2) != tci_s) w2 = next_lds[w & 0xFFFFu];
      next2_lds[st] = w2;
    }
    __syncthreads();                  // the staged copies are done with: the records take their place
  }
  const int tcell = target_cell[ga];
  int32_t* out_len = alt_len + (size_t)g * cap_alt;
  int32_t* slot_cell = alt_cell + (size_t)g * cap_alt * Lcap;
  uint32_t* slot_meta = alt_meta + (size_t)g * cap_alt * Lcap;
  int n_leaf = 0, err = 0;

  if (live && j == 0) {
    const int pos = pos_cell[ga], h0 = direction[ga] & 3;
    // prefix elements (cell, meta) and DFS frames over switch cells: the group's slice of LDS (private arrays of
    // this size would live in scratch memory)
    int* frame = alt_frames + grp * FLM_ALT_FRAME_INTS;
    int* pre_cell = frame;                                                   // [FLM_ALT_STACK + 2]
    unsigned* pre_meta = reinterpret_cast<unsigned*>(frame + FLM_ALT_STACK + 2);   // [FLM_ALT_STACK + 2]
    int* f_cell = frame + 2 * (FLM_ALT_STACK + 2);                           // [FLM_ALT_STACK] each
    int* f_h = f_cell + FLM_ALT_STACK;
    int* f_i = f_h + FLM_ALT_STACK;
    int* f_k = f_i + FLM_ALT_STACK;
    int(*f_act)[3] = reinterpret_cast<int(*)[3]>(f_k + FLM_ALT_STACK);        // [FLM_ALT_STACK][3] each
    int(*f_nd)[3] = reinterpret_cast<int(*)[3]>(f_k + FLM_ALT_STACK + 3 * FLM_ALT_STACK);
    if (pos == tcell || pos < 0 || pos >= HW) {
      err = FLM_AGENT_ASSERT_ERROR;
    } else {
      int act0[3], nd0[3];
      const int k0 = valid_moves(grid[pos], h0, act0, nd0);
      const int pr = pos / W, pc = pos - pr * W;
      for (int i0 = 0; i0 < k0 && !err; i0++) {
        // the first element is always recorded as "forward"
        pre_cell[0] = pos;
        pre_meta[0] = (unsigned)h0 | (2u << 8) | ((unsigned)nd0[i0] << 16);
        int np = 1, depth = 0;
        {
          int nr = pr + dr(nd0[i0]), nc = pc + dc(nd0[i0]);
          if (nr < 0 || nr >= H || nc < 0 || nc >= W) { err = FLM_AGENT_ASSERT_ERROR; break; }
          f_cell[0] = nr * W + nc;
          f_h[0] = nd0[i0];
          f_i[0] = 0;
          f_k[0] = valid_moves(grid[f_cell[0]], f_h[0], f_act[0], f_nd[0]);
          depth = 1;
        }
        while (depth > 0 && !err) {
          const int d = depth - 1;
          const int cell = f_cell[d], h = f_h[d];
          const int r = cell / W, c = cell - r * W;
          if (node_id[cell] >= 0) {
            if (f_i[d] > 0) np--;  // back from the previous branch
            if (f_i[d] < f_k[d]) {
              const int i = f_i[d]++;
              pre_cell[np] = cell;
              pre_meta[np] = (unsigned)h | ((unsigned)f_act[d][i] << 8) | ((unsigned)f_nd[d][i] << 16);
              np++;
              int nr = r + dr(f_nd[d][i]), nc = c + dc(f_nd[d][i]);
              if (depth >= FLM_ALT_STACK || nr < 0 || nr >= H || nc < 0 || nc >= W) { err = FLM_AGENT_OVERFLOW; break; }
              f_cell[depth] = nr * W + nc;
              f_h[depth] = f_nd[d][i];
              f_i[depth] = 0;
              f_k[depth] = valid_moves(grid[f_cell[depth]], f_h[depth], f_act[depth], f_nd[depth]);
              depth++;
            } else {
              depth--;
            }
          } else {
            if (f_k[d] != 1) { err = FLM_AGENT_ASSERT_ERROR; break; }
            if (n_leaf >= cap_alt || np + 1 > Lcap) { err = FLM_AGENT_OVERFLOW; break; }
            int32_t* oc = slot_cell + (size_t)n_leaf * Lcap;
            uint32_t* om = slot_meta + (size_t)n_leaf * Lcap;
            for (int q = 0; q < np; q++) { oc[q] = pre_cell[q]; om[q] = pre_meta[q]; }
            oc[np] = cell;
            om[np] = (unsigned)h | ((unsigned)f_act[d][0] << 8) | ((unsigned)f_nd[d][0] << 16);
            const int nr = r + dr(f_nd[d][0]), nc = c + dc(f_nd[d][0]);
            // where the descent starts: -1 = the leaf is the target itself, -2 = the move leaves the grid
            s_cell[grp * G + n_leaf] = cell == tcell ? -1 : (nr >= 0 && nr < H && nc >= 0 && nc < W) ? nr * W + nc : -2;
            s_h[grp * G + n_leaf] = f_nd[d][0];
            s_base[grp * G + n_leaf] = np + 1;
            n_leaf++;
            depth--;
          }
        }
      }
    }
    s_n[grp] = n_leaf;
    s_err[grp] = err;
  }
  __syncthreads();
  if (live) {
    n_leaf = s_n[grp];
    err = s_err[grp];
    if (!err && j < n_leaf) {
      const int start = s_cell[lane], base = s_base[lane];
      int res = -1;                       // elements of the finished alternative; -1 = no way, -2 = does not fit
      if (start == -1) {
        res = base;
      } else if (start >= 0) {
        int n;
        if (STAGE) {
          // The descent of greedy_path() on the request's move tables: an element is recorded as
          // state | action << 16 | next heading << 20; cells and the output layout follow when all lanes copy the
          // records out.
          uint32_t* rec = rec_lds + (size_t)j * Lcap;
          const int cap = Lcap - base;
          const int tci = tci_s;
          const int idx0 = cell_idx[start];
          n = -1;
          if (idx0 >= 0) {
            int st = idx0 * 4 + s_h[lane], depth = 0;
            bool dead = false;
            n = 0;
            // a walk that starts on a forced move from a state that reaches nothing ends at once (greedy_path() learns
            // the distance at its first forced step); everywhere else "no way" is the table's own mark
            if ((st >> 2) != tci && max_depth > 0 && succ[st] != 0xFFFFu && dist[st] == FLM_DIST_INF) dead = true;
            while (!dead && (st >> 2) != tci && depth < max_depth) {
              const uint32_t w1 = next_lds[st], w2 = next2_lds[st];
              if (w1 == NO_MOVE) { dead = true; break; }
              const unsigned s1 = w1 & 0xFFFFu;
              if (n < cap) rec[n] = (unsigned)st | (w1 & 0x70000u) | ((s1 & 3u) << 20);
              n++;
              depth++;
              st = (int)s1;
              if (w2 != NO_MOVE && depth < max_depth) {          // (s1 is not on the target cell: the table says so)
                const unsigned s2 = w2 & 0xFFFFu;
                if (n < cap) rec[n] = s1 | (w2 & 0x70000u) | ((s2 & 3u) << 20);
                n++;
                depth++;
                st = (int)s2;
              }
            }
            if (dead) {
              n = -1;
            } else if (depth < max_depth) {     // closing element: stop on the cell reached
              if (n < cap) rec[n] = (unsigned)st | (4u << 16) | (((unsigned)st & 3u) << 20);
              n++;
            }
          }
          s_cnt[lane] = n < 0 ? 0 : (n < cap ? n : cap);
        } else {
          n = greedy_path<true>(grid, cell_idx, dist, succ, choice, H, W, start, s_h[lane], tcell, max_depth,
                                slot_cell + (size_t)j * Lcap + base, slot_meta + (size_t)j * Lcap + base, Lcap - base);
        }
        if (n >= 0) res = base + n > Lcap ? -2 : base + n;
      }
      s_res[lane] = res;
    }
  }
  __syncthreads();
  if (STAGE && live && !err) {
    // the records of every descent become path elements: all lanes, coalesced
    for (int q = 0; q < n_leaf; q++) {
      if (s_cell[q] < 0) continue;        // no descent ran for this leaf
      const int cnt = s_cnt[q], base = s_base[q];
      const uint32_t* rec = rec_lds + (size_t)q * Lcap;
      for (int i = lane; i < cnt; i += 64) {
        const unsigned v = rec[i];
        const unsigned st = v & 0xFFFFu;
        slot_cell[(size_t)q * Lcap + base + i] = icell_lds[st >> 2];
        slot_meta[(size_t)q * Lcap + base + i] = (st & 3u) | (((v >> 16) & 7u) << 8) | (((v >> 20) & 3u) << 16);
      }
    }
    __syncthreads();                       // the squeeze below reads what other lanes have just written
  }
  int n_alt = 0;
  if (live && !err) {
    for (int q = 0; q < n_leaf; q++) {
      const int res = s_res[grp * G + q];
      if (res == -2) { err = FLM_AGENT_OVERFLOW; break; }
      if (res < 0) continue;
      if (n_alt != q)                     // an earlier leaf fell out: move this one down (each lane its own elements)
        for (int i = j; i < res; i += G) {
          slot_cell[(size_t)n_alt * Lcap + i] = slot_cell[(size_t)q * Lcap + i];
          slot_meta[(size_t)n_alt * Lcap + i] = slot_meta[(size_t)q * Lcap + i];
        }
      if (j == 0) out_len[n_alt] = res;
      n_alt++;
    }
  }
  if (live) {
    // unused alternatives have length 0 (the caller need not clear the array)
    for (int q = (err ? 0 : n_alt) + j; q < cap_alt; q += G) out_len[q] = 0;
    if (j == 0) {
      alt_count[g] = err ? 0 : n_alt;
      err_out[g] = err;
    }
  }
}

// ---- occupancy-plane bookkeeping ---------------------------------------------------------------------
// One 256-thread workgroup per environment; the planes of an environment are [A][plane_stride] int8
// with rows of T1 bytes.  All helpers are called by the whole workgroup.

struct EnvMaps {
  int8_t* base;
  int A, Rp, T1, plane_stride;
  __device__ __forceinline__ int8_t* plane(int a) const { return base + (size_t)a * plane_stride; }
};

// First rail with a non-zero entry in column ts of agent a's plane (0 when the column is empty) and the
// entry found there.  s_tmp: one LDS int.
__device__ int blk_rail_dir(const EnvMaps& m, int a, int ts, int* s_tmp, int* dir) {
  if (threadIdx.x == 0) *s_tmp = 0x7FFFFFFF;
  __syncthreads();
  const int8_t* p = m.plane(a);
  for (int r = threadIdx.x; r < m.Rp; r += blockDim.x)
    if (p[(size_t)r * m.T1 + ts] != 0) atomicMin(s_tmp, r);
  __syncthreads();
  int rail = *s_tmp;
  if (rail == 0x7FFFFFFF) rail = 0;
  *dir = p[(size_t)rail * m.T1 + ts];
  __syncthreads();
  return rail;
}

// index of the first zero at or after `from` in a row, relative to `from`; 0 when there is none
__device__ __forceinline__ int first_zero(const int8_t* row, int from, int T1) {
  for (int t = from; t < T1; t++)
    if (row[t] == 0) return t - from;
  return 0;
}

__device__ __forceinline__ int thread_rail_at0(const EnvMaps& m, int a) {
  const int8_t* p = m.plane(a);
  for (int r = 0; r < m.Rp; r++)
    if (p[(size_t)r * m.T1] != 0) return r;
  return 0;
}

// The train (other than a, already departed) that leaves `rail` last, and its expected exit time.
// s_tmp: one LDS int.  Returns exit time (0 = nobody), *last = that train.
__device__ int blk_last_train_on_rail(const EnvMaps& m, const int32_t* status, const int32_t* tpc, int a, int rail,
                                      int* s_tmp, int* last) {
  if (threadIdx.x == 0) *s_tmp = 0;
  __syncthreads();
  for (int o = threadIdx.x; o < m.A; o += blockDim.x) {
    if (o == a || status[o] == 0) continue;
    const int8_t* p = m.plane(o);
    const int8_t* row = p + (size_t)rail * m.T1;
    int t = tpc[o];
    int ex = -1;
    if (row[0] != 0) {
      ex = first_zero(row, 0, m.T1);
    } else if (t >= 1 && t - 1 < m.T1 && row[t - 1] != 0) {
      int orail = thread_rail_at0(m, o);
      ex = 0;
      if (orail != rail) ex = first_zero(p + (size_t)orail * m.T1, 0, m.T1);
      ex += first_zero(row, ex, m.T1);
    }
    // sequential rule: a later train replaces the current one only with a strictly larger exit time
    if (ex > 0) atomicMax(s_tmp, (ex << 12) | (4095 - o));
  }
  __syncthreads();
  int key = *s_tmp;
  __syncthreads();
  if (key == 0) {
    *last = 0;
    return 0;
  }
  *last = 4095 - (key & 4095);
  return key >> 12;
}

__device__ bool blk_headon_crash(const EnvMaps& m, const int32_t* status, const int32_t* tpc, int a, int rail, int dir,
                                 int* s_tmp) {
  int last;
  int le = blk_last_train_on_rail(m, status, tpc, a, rail, s_tmp, &last);
  if (le > 0) return (le - 1 < m.T1 ? m.plane(last)[(size_t)rail * m.T1 + le - 1] : 0) != dir;
  return false;
}

__global__ __launch_bounds__(256) void check_crash_kernel(const int8_t* __restrict__ maps, const int32_t* __restrict__ status_all,
                                                          const int32_t* __restrict__ tpc_all,
                                                          const int32_t* __restrict__ pos_all,
                                                          const int32_t* __restrict__ init_all,
                                                          const int32_t* __restrict__ agent_sel,
                                                          const int32_t* __restrict__ before_switch,
                                                          const int32_t* __restrict__ next_cell, int A, int Rp, int T1,
                                                          int plane_stride, int32_t* __restrict__ crash_out) {
  __shared__ int s_tmp[2];
  const int e = blockIdx.x;
  const int a = agent_sel[e];
  if (a < 0 || a >= A) return;
  EnvMaps m{const_cast<int8_t*>(maps) + (size_t)e * A * plane_stride, A, Rp, T1, plane_stride};
  const int32_t* status = status_all + (size_t)e * A;
  const int32_t* tpc = tpc_all + (size_t)e * A;
  const int32_t* pos = pos_all + (size_t)e * A;
  bool crash = false;
  if (status[a] == 0) {
    const int init = init_all[(size_t)e * A + a];
    if (threadIdx.x == 0) s_tmp[1] = 0;
    __syncthreads();
    for (int o = threadIdx.x; o < A; o += blockDim.x)
      if (o != a && pos[o] == init) s_tmp[1] = 1;
    __syncthreads();
    crash = s_tmp[1] != 0;
    __syncthreads();
    if (!crash) {
      int dir;
      int rail = blk_rail_dir(m, a, 1, s_tmp, &dir);
      crash = blk_headon_crash(m, status, tpc, a, rail, dir, s_tmp);
    }
  } else if (before_switch[e]) {
    int t = tpc[a];
    if (t >= T1) t = T1 - 1;
    int dir;
    int rail = blk_rail_dir(m, a, t, s_tmp, &dir);
    crash = blk_headon_crash(m, status, tpc, a, rail, dir, s_tmp);
  } else {
    const int nxt = next_cell[e];
    if (threadIdx.x == 0) s_tmp[1] = 0;
    __syncthreads();
    if (nxt >= 0)
      for (int o = threadIdx.x; o < A; o += blockDim.x)
        if (o != a && pos[o] == nxt) s_tmp[1] = 1;
    __syncthreads();
    crash = s_tmp[1] != 0;
  }
  if (threadIdx.x == 0) crash_out[e] = crash ? 1 : 0;
}

// Shift agent a's plane by `delay` to the right, clear the first delay+tpc columns, keep the current rail
// for tpc columns and hold the next rail for `delay` more columns.
__device__ void blk_delay(const EnvMaps& m, const int32_t* tpc, int a, int rail, int dir, int delay, int* s_tmp) {
  int odir;
  const int orail = blk_rail_dir(m, a, 0, s_tmp, &odir);
  const int t = tpc[a];
  int8_t* p = m.plane(a);
  int z = delay + t;
  if (z > m.T1) z = m.T1;
  for (int r = threadIdx.x; r < m.Rp; r += blockDim.x) {
    int8_t* row = p + (size_t)r * m.T1;
    for (int c = m.T1 - 1; c >= z; c--) row[c] = row[c - delay];
    for (int c = 0; c < z; c++) row[c] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int c = 0; c < t && c < m.T1; c++) p[(size_t)orail * m.T1 + c] = (int8_t)odir;
    for (int c = t; c < t + delay && c < m.T1; c++) p[(size_t)rail * m.T1 + c] = (int8_t)dir;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void update_bitmaps_kernel(int8_t* __restrict__ maps, const int32_t* __restrict__ status_all,
                                                             const int32_t* __restrict__ tpc_all,
                                                             const int32_t* __restrict__ agent_sel,
                                                             const int32_t* __restrict__ before_switch, int A, int Rp,
                                                             int T1, int plane_stride) {
  __shared__ int s_tmp[2];
  const int e = blockIdx.x;
  const int a = agent_sel[e];
  if (a < 0 || a >= A) return;
  EnvMaps m{maps + (size_t)e * A * plane_stride, A, Rp, T1, plane_stride};
  const int32_t* status = status_all + (size_t)e * A;
  const int32_t* tpc = tpc_all + (size_t)e * A;
  int8_t* p = m.plane(a);
  if (before_switch[e]) {
    int t = tpc[a];
    if (t >= T1) t = T1 - 1;
    int dir, last;
    const int rail = blk_rail_dir(m, a, t, s_tmp, &dir);
    const int le = blk_last_train_on_rail(m, status, tpc, a, rail, s_tmp, &last);
    if (le > 0) {
      const int cur = first_zero(p + (size_t)rail * T1, t, T1) + t;
      if (cur <= le) blk_delay(m, tpc, a, rail, dir, le + t - cur, s_tmp);
    }
  }
  // one step passes: every row moves one column to the left
  for (int r = threadIdx.x; r < Rp; r += blockDim.x) {
    int8_t* row = p + (size_t)r * T1;
    for (int c = 0; c + 1 < T1; c++) row[c] = row[c + 1];
    row[T1 - 1] = 0;
  }
}

}  // namespace flm

static int altpaths_impl(const uint16_t* grid, const int32_t* cell_idx, const uint16_t* dist, const uint16_t* succ,
                         const uint16_t* choice, const int32_t* idx_cell, const int32_t* tslot, const int32_t* node_id,
                         const int32_t* req_agent, int n_req, const int32_t* n_live, int live_base, int32_t* agent_rep,
                         const int32_t* pos_cell,
                         const int32_t* direction, const int32_t* target_cell, int E, int A, int H, int W,
                         int capCells, int max_depth, int cap_alt, int Lcap, int32_t* alt_count, int32_t* alt_len,
                         int32_t* alt_cell, uint32_t* alt_meta, int32_t* err, void* stream) {
  if (!grid || !cell_idx || !dist || !tslot || !node_id || !req_agent || !pos_cell || !direction || !target_cell ||
      !alt_count || !alt_len || !alt_cell || !alt_meta || !err)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || H <= 0 || W <= 0 || n_req <= 0 || capCells <= 0 || max_depth < 0 || cap_alt <= 0 ||
      cap_alt > 64 || Lcap < max_depth + 4)
    return FLM_E_ARG;
  int G = 1;                              // lanes per request
  while (G < cap_alt) G *= 2;
  // one request per wave with its move tables, the cell table and the descents' records in LDS when the successor and
  // choice tables exist, are 8-byte aligned per environment (an even capCells) and everything fits beside the frames
  const size_t staged_lds = (size_t)FLM_ALT_FRAME_INTS * sizeof(int) + (size_t)capCells * 32 + (size_t)((capCells + 3) & ~3) * 4 +
                            std::max((size_t)capCells * 8, (size_t)cap_alt * Lcap * 4) + 16;
  const bool stage = succ && choice && idx_cell && (capCells & 1) == 0 && capCells * 4 <= 65535 &&
                     (((uintptr_t)succ | (uintptr_t)choice | (uintptr_t)dist) & 15) == 0 && staged_lds <= 64 * 1024;
  if (stage) {
    FLM_RAISE_LDS_ONCE(flm::altpaths_kernel<true>);
    hipLaunchKernelGGL(flm::altpaths_kernel<true>, dim3(n_req), dim3(64), staged_lds, (hipStream_t)stream, grid, cell_idx,
                       dist, succ, choice, idx_cell, tslot, node_id, req_agent, n_req, pos_cell, direction, target_cell, A, H, W,
                       capCells, max_depth, cap_alt, Lcap, 64, alt_count, alt_len, alt_cell, alt_meta, err, n_live, live_base,
                       agent_rep);
  } else {
    const int per_block = 64 / G;
    const size_t lds = (size_t)per_block * FLM_ALT_FRAME_INTS * sizeof(int);
    if (lds > 160 * 1024 - 512) return FLM_E_ARG;
    FLM_RAISE_LDS_ONCE(flm::altpaths_kernel<false>);
    hipLaunchKernelGGL(flm::altpaths_kernel<false>, dim3((n_req + per_block - 1) / per_block), dim3(64), lds,
                       (hipStream_t)stream, grid, cell_idx, dist, succ, choice, idx_cell, tslot, node_id, req_agent, n_req, pos_cell,
                       direction, target_cell, A, H, W, capCells, max_depth, cap_alt, Lcap, G, alt_count, alt_len, alt_cell,
                       alt_meta, err, n_live, live_base, agent_rep);
  }
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_altpaths(const uint16_t* grid, const int32_t* cell_idx, const uint16_t* dist, const uint16_t* succ,
                            const uint16_t* choice, const int32_t* idx_cell, const int32_t* tslot, const int32_t* node_id,
                            const int32_t* req_agent, int n_req, const int32_t* pos_cell,
                            const int32_t* direction, const int32_t* target_cell, int E, int A, int H, int W,
                            int capCells, int max_depth, int cap_alt, int Lcap, int32_t* alt_count, int32_t* alt_len,
                            int32_t* alt_cell, uint32_t* alt_meta, int32_t* err, void* stream) {
  return altpaths_impl(grid, cell_idx, dist, succ, choice, idx_cell, tslot, node_id, req_agent, n_req, nullptr, 0, nullptr, pos_cell,
                       direction, target_cell, E, A, H, W, capCells, max_depth, cap_alt, Lcap, alt_count, alt_len, alt_cell,
                       alt_meta, err, stream);
}

extern "C" int flm_altpaths_counted(const uint16_t* grid, const int32_t* cell_idx, const uint16_t* dist, const uint16_t* succ,
                                    const uint16_t* choice, const int32_t* idx_cell, const int32_t* tslot,
                                    const int32_t* node_id, const int32_t* req_agent, int n_max, const int32_t* n_live,
                                    int live_base, int32_t* agent_rep, const int32_t* pos_cell, const int32_t* direction,
                                    const int32_t* target_cell, int E, int A, int H, int W, int capCells, int max_depth,
                                    int cap_alt, int Lcap, int32_t* alt_count, int32_t* alt_len, int32_t* alt_cell,
                                    uint32_t* alt_meta, int32_t* err, void* stream) {
  if (!n_live || live_base < 0) return FLM_E_ARG;
  return altpaths_impl(grid, cell_idx, dist, succ, choice, idx_cell, tslot, node_id, req_agent, n_max, n_live, live_base,
                       agent_rep, pos_cell, direction, target_cell, E, A, H, W, capCells, max_depth, cap_alt, Lcap, alt_count, alt_len,
                       alt_cell, alt_meta, err, stream);
}

extern "C" int flm_check_crash(const int8_t* maps, const int32_t* status, const int32_t* tpc, const int32_t* pos_cell,
                               const int32_t* init_cell, const int32_t* agent_sel, const int32_t* before_switch,
                               const int32_t* next_cell, int E, int A, int Rp, int T1, int plane_stride, int32_t* crash,
                               void* stream) {
  if (!maps || !status || !tpc || !pos_cell || !init_cell || !agent_sel || !before_switch || !next_cell || !crash)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || A > 4095 || Rp <= 0 || T1 <= 1 || T1 > 8000 || (long long)plane_stride < (long long)Rp * T1)
    return FLM_E_ARG;
  hipLaunchKernelGGL(flm::check_crash_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, maps, status, tpc, pos_cell,
                     init_cell, agent_sel, before_switch, next_cell, A, Rp, T1, plane_stride, crash);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_update_bitmaps(int8_t* maps, const int32_t* status, const int32_t* tpc, const int32_t* agent_sel,
                                  const int32_t* before_switch, int E, int A, int Rp, int T1, int plane_stride,
                                  void* stream) {
  if (!maps || !status || !tpc || !agent_sel || !before_switch) return FLM_E_ARG;
  if (E <= 0 || A <= 0 || A > 4095 || Rp <= 0 || T1 <= 1 || T1 > 8000 || (long long)plane_stride < (long long)Rp * T1)
    return FLM_E_ARG;
  hipLaunchKernelGGL(flm::update_bitmaps_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, maps, status, tpc, agent_sel,
                     before_switch, A, Rp, T1, plane_stride);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
