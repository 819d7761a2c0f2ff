// Path -> rail-occupancy plane, and the per-agent observation (own plane + two heat planes).
#include "flm_common.h"
#include "flm_columns.h"


namespace flm {

// One 256-thread workgroup per (environment, path).  All threads clear the plane with 16-byte stores
// while they look up, per time step, the cell the agent is predicted to occupy and the rail / node in
// that cell.  Thread 0 then runs the short sequential pass that tracks through which node the agent
// entered its current rail (that decides the sign) and back-fills the steps spent on switch cells
// with the rail that follows.  After a barrier every step writes its single byte.
__global__ __launch_bounds__(256) void bitmaps_kernel(
    const int32_t* __restrict__ path_cell, const int32_t* __restrict__ path_len, const int32_t* __restrict__ agent_flat,
    const int32_t* __restrict__ pos_cell, const int32_t* __restrict__ target_cell, const int32_t* __restrict__ tpc_all,
    const int32_t* __restrict__ node_id_all, const int32_t* __restrict__ cell_rail_all,
    const int32_t* __restrict__ edge_info_all, int A, int H, int W, int capR, int Rp, int T1, int plane_stride,
    int Lcap, int shift, int8_t* __restrict__ planes, int16_t* __restrict__ cols, int32_t* __restrict__ err_out) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  int* s_cell = lds;             // [T1] predicted cell
  int* s_rail = lds + T1;        // [T1] rail in that cell or -1
  int* s_node = lds + 2 * T1;    // [T1] node in that cell or -1
  int* s_orail = lds + 3 * T1;   // [T1] rail to paint at the step, -1 = none
  int* s_oval = lds + 4 * T1;    // [T1] its sign
  int* s_misc = lds + 5 * T1;    // [0] kstop, [1] n (steps kept), [2] error
  const int gp = blockIdx.x;
  const int ga = agent_flat ? agent_flat[gp] : gp;  // e*A + a
  const int e = ga / A;
  const int HW = H * W;
  const int tid = threadIdx.x;
  const int pos = pos_cell[ga], tgt = target_cell[ga];
  int tpc = tpc_all[ga];
  if (tpc < 1) tpc = 1;
  int len = path_len[gp];
  if (len > Lcap) len = Lcap;
  const int32_t* pcell = path_cell + (size_t)gp * Lcap;
  const int32_t* node_id = node_id_all + (size_t)e * HW;
  const int32_t* cell_rail = cell_rail_all + (size_t)e * HW;
  const int32_t* edge_info = edge_info_all + (size_t)e * capR * 5;
  int8_t* plane = planes ? planes + (size_t)gp * plane_stride : nullptr;

  if (tid == 0) {
    s_misc[0] = len > 0 ? len - 1 : 0;
    s_misc[1] = T1;
    s_misc[2] = 0;
  }
  // clear the plane
  if (plane) {
    uint4 z = make_uint4(0, 0, 0, 0);
    uint4* p4 = reinterpret_cast<uint4*>(plane);
    for (int i = tid; i < plane_stride / 16; i += 256) p4[i] = z;
  }
  __syncthreads();
  // the agent stops advancing at the first path element on the target cell (element 0 is where it stands)
  for (int k = tid; k < len; k += 256) {
    int c = (k == 0) ? pos : pcell[k];
    if (c == tgt) atomicMin(&s_misc[0], k);
  }
  __syncthreads();
  const int kstop = s_misc[0];
  for (int ts = tid; ts < T1; ts += 256) {
    int k = ts / tpc;
    if (k > kstop) k = kstop;
    int c = (k == 0) ? pos : pcell[k];
    s_cell[ts] = c;
    bool inside = c >= 0 && c < HW;
    s_rail[ts] = inside ? cell_rail[c] : -1;
    s_node[ts] = inside ? node_id[c] : -1;
    s_orail[ts] = -1;
    if (c == tgt) atomicMin(&s_misc[1], ts + 1);  // cut after the first visit of the target
  }
  __syncthreads();
  const int n = s_misc[1];
  if (tid == 0) {
    int err = 0;
    int en = -1, ecp = -1;  // (node, cardinal point) through which the current rail was entered
    int i = 0;
    if (s_rail[0] != -1) {
      const int initial = s_rail[0];
      int rail = initial;
      while (rail != -1) {
        i++;
        if (i >= n) { err = FLM_AGENT_INDEX_ERROR; break; }
        rail = s_rail[i];
      }
      if (!err) {
        int nid = s_node[i];
        if (nid < 0) err = FLM_AGENT_KEY_ERROR;
        else {
          int c0 = s_cell[i - 1], c1 = s_cell[i];
          int cp = (direction_to_point(c0 / W, c0 % W, c1 / W, c1 % W) + 2) & 3;
          const int32_t* ei = edge_info + initial * 5;
          if (nid == ei[2] && cp == ei[3]) { en = ei[0]; ecp = ei[1]; }
          else if (nid == ei[0] && cp == ei[1]) { en = ei[2]; ecp = ei[3]; }
        }
      }
    } else {
      int nid = s_node[0];
      if (nid < 0) err = FLM_AGENT_KEY_ERROR;
      else if (n < 2) err = FLM_AGENT_INDEX_ERROR;
      else {
        int c0 = s_cell[0], c1 = s_cell[1];
        en = nid;
        ecp = direction_to_point(c0 / W, c0 % W, c1 / W, c1 % W);
      }
    }
    int holes = 0;
    for (int ts = 0; ts < n && !err; ts++) {
      int rail = s_rail[ts];
      if (rail == -1) {
        holes++;
        if (ts + 1 < n && s_cell[ts] != s_cell[ts + 1]) {
          int nid = s_node[ts];
          if (nid < 0) { err = FLM_AGENT_KEY_ERROR; break; }
          int c0 = s_cell[ts], c1 = s_cell[ts + 1];
          en = nid;
          ecp = direction_to_point(c0 / W, c0 % W, c1 / W, c1 % W);
        }
      } else {
        const int32_t* ei = edge_info + rail * 5;
        int dir = 0;
        if (en == ei[2] && ecp == ei[3]) dir = 1;
        else if (en == ei[0] && ecp == ei[1]) dir = -1;
        if (dir == 0) { err = FLM_AGENT_ASSERT_ERROR; break; }
        for (int b = ts - holes; b <= ts; b++) {
          s_orail[b] = rail;
          s_oval[b] = dir;
        }
        holes = 0;
      }
    }
    s_misc[2] = err;
    err_out[gp] = err;
  }
  __syncthreads();  // also orders the clearing stores before the bytes written below
  const bool ok = s_misc[2] == 0;
  if (plane && ok) {
    for (int ts = tid; ts < n; ts += 256) {
      int rail = s_orail[ts];
      int t = ts + shift;
      if (rail >= 0 && rail < Rp && t < T1) plane[(size_t)rail * T1 + t] = (int8_t)s_oval[ts];
    }
  }
  if (cols) {
    // column form of the same plane: per time step, rail*2 + (sign > 0), or -1 when the column is empty
    int16_t* col = cols + (size_t)gp * T1;
    for (int t = tid; t < T1; t += 256) {
      int ts = t - shift;
      int v = -1;
      if (ok && ts >= 0 && ts < n) {
        int rail = s_orail[ts];
        if (rail >= 0 && rail < Rp) v = rail * 2 + (s_oval[ts] > 0 ? 1 : 0);
      }
      col[t] = (int16_t)v;
    }
  }
}

// ---- column form, one wavefront per path (the scans are in flm_columns.h) ---------------------------------
__global__ __launch_bounds__(256) void columns_wave_kernel(
    const int32_t* __restrict__ path_cell, const int32_t* __restrict__ path_len, const int32_t* __restrict__ agent_flat,
    const int32_t* __restrict__ pos_cell, const int32_t* __restrict__ target_cell, const int32_t* __restrict__ tpc_all,
    const int32_t* __restrict__ node_id_all, const int32_t* __restrict__ cell_rail_all,
    const int32_t* __restrict__ edge_info_all, int A, int n_paths, int H, int W, int capR, int Rp, int T1, int Lcap,
    int shift, int16_t* __restrict__ cols, int32_t* __restrict__ err_out) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gp = blockIdx.x * 4 + wave;
  if (gp >= n_paths) return;   // wave-uniform; no workgroup barrier is used below
  int* s_cell = lds + wave * 4 * T1;
  int* s_rail = s_cell + T1;
  int* s_node = s_rail + T1;
  int* s_dir = s_node + T1;
  const int ga = agent_flat ? agent_flat[gp] : gp;
  const int e = ga / A;
  const int HW = H * W;
  const int pos = pos_cell[ga], tgt = target_cell[ga];
  int tpc = tpc_all[ga];
  if (tpc < 1) tpc = 1;
  int len = path_len[gp];
  if (len > Lcap) len = Lcap;
  const int32_t* pcell = path_cell + (size_t)gp * Lcap;
  const int32_t* node_id = node_id_all + (size_t)e * HW;
  const int32_t* cell_rail = cell_rail_all + (size_t)e * HW;
  const int32_t* edge_info = edge_info_all + (size_t)e * capR * 5;
  const int KT = (T1 + 63) >> 6;

  // the agent stops advancing at the first path element on the target cell
  int kstop = len > 0 ? len - 1 : 0;
  for (int base = 0; base < len; base += 64) {
    const int k = base + lane;
    const int fl = first_lane(k < len && ((k == 0) ? pos : pcell[k]) == tgt);
    if (fl < 64) {
      kstop = base + fl;
      break;
    }
  }
  int n = T1;                     // steps kept: cut after the first visit of the target
  for (int q = 0; q < KT; q++) {
    const int ts = lane + 64 * q;
    bool at_target = false;
    if (ts < T1) {
      int k = ts / tpc;
      if (k > kstop) k = kstop;
      const int c = (k == 0) ? pos : pcell[k];
      const bool inside = c >= 0 && c < HW;
      s_cell[ts] = c;
      s_rail[ts] = inside ? cell_rail[c] : -1;
      s_node[ts] = inside ? node_id[c] : -1;
      at_target = c == tgt;
    }
    const int fl = first_lane(at_target);
    if (fl < 64 && n == T1) n = 64 * q + fl + 1;
  }
  __builtin_amdgcn_wave_barrier();

  const int err = columns_from_steps<int, int>(s_cell, s_rail, s_node, s_dir, n, T1, CellLinear{W}, RailEndsGlobal{edge_info}, Rp,
                                               shift, lane, cols + (size_t)gp * T1);
  if (lane == 0) err_out[gp] = err;
}

// Dense plane of every path from its column form: clear with 16-byte stores, then one byte per column.
__global__ __launch_bounds__(256) void planes_from_columns_kernel(const int16_t* __restrict__ cols, int T1,
                                                                  int plane_stride, int8_t* __restrict__ planes) {
  const int gp = blockIdx.x, tid = threadIdx.x;
  int8_t* plane = planes + (size_t)gp * plane_stride;
  const uint4 z = make_uint4(0, 0, 0, 0);
  uint4* p4 = reinterpret_cast<uint4*>(plane);
  for (int i = tid; i < plane_stride / 16; i += 256) p4[i] = z;
  __syncthreads();   // orders the clearing stores of the workgroup before the bytes below
  const int16_t* col = cols + (size_t)gp * T1;
  for (int t = tid; t < T1; t += 256) {
    const int c = col[t];
    if (c >= 0) plane[(size_t)(c >> 1) * T1 + t] = (c & 1) ? (int8_t)1 : (int8_t)-1;
  }
}

// ---- observations ---------------------------------------------------------------------------------
// A workgroup owns 1 KiB of plane positions (16 per lane) of one environment.  Its NW waves split the
// agents; each wave keeps its agents' 16-byte pieces in registers, counts the +1 and the -1 entries in
// packed bytes, the per-wave counts meet in LDS, and every wave then writes, for its agents, the own
// plane and "total minus self" for both signs.  Each input byte is read once and each output byte
// written once, all as 16-byte accesses of 1 KiB per wave instruction.
__device__ __forceinline__ unsigned pos_bits(unsigned x) { return x & ~(x >> 1) & 0x01010101u; }
__device__ __forceinline__ unsigned neg_bits(unsigned x) { return (x >> 1) & 0x01010101u; }
__device__ __forceinline__ uint4 pos_bits4(uint4 v) { return make_uint4(pos_bits(v.x), pos_bits(v.y), pos_bits(v.z), pos_bits(v.w)); }
__device__ __forceinline__ uint4 neg_bits4(uint4 v) { return make_uint4(neg_bits(v.x), neg_bits(v.y), neg_bits(v.z), neg_bits(v.w)); }
__device__ __forceinline__ uint4 add4(uint4 a, uint4 b) { return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ uint4 sub4(uint4 a, uint4 b) { return make_uint4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <bool NT>
__device__ __forceinline__ void store16(uint8_t* p, uint4 v) {
  typedef unsigned vec4u __attribute__((ext_vector_type(4)));
  if (NT) {
    vec4u t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vec4u*>(p));
  } else {
    *reinterpret_cast<uint4*>(p) = v;
  }
}

// APW = agents per wave, CH = 1-KiB chunks per workgroup (each lane owns CH pieces of 16 bytes),
// NT = non-temporal stores for the write-once output.
template <int APW, int CH, bool NT>
__global__ __launch_bounds__(1024) void observations_kernel(const int8_t* __restrict__ maps, const int8_t* __restrict__ own,
                                                            int A, int plane_stride, uint8_t* __restrict__ obs) {
  extern __shared__ __attribute__((aligned(16))) uint4 part[];  // [nw][2*CH][64]
  const int nw = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.y;
  const size_t off0 = (size_t)blockIdx.x * (1024 * CH) + (size_t)lane * 16;
  const size_t env_base = (size_t)e * A;
  uint4 v[APW][CH];
  uint4 psum[CH], nsum[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) psum[c] = nsum[c] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < APW; i++) {
    int a = wave + i * nw;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      size_t off = off0 + (size_t)c * 1024;
      v[i][c] = make_uint4(0, 0, 0, 0);
      if (off < (size_t)plane_stride && a < A)
        v[i][c] = *reinterpret_cast<const uint4*>(maps + (env_base + a) * plane_stride + off);
    }
  }
#pragma unroll
  for (int i = 0; i < APW; i++)
#pragma unroll
    for (int c = 0; c < CH; c++) {
      psum[c] = add4(psum[c], pos_bits4(v[i][c]));
      nsum[c] = add4(nsum[c], neg_bits4(v[i][c]));
    }
#pragma unroll
  for (int c = 0; c < CH; c++) {
    part[((wave * CH + c) * 2 + 0) * 64 + lane] = psum[c];
    part[((wave * CH + c) * 2 + 1) * 64 + lane] = nsum[c];
  }
  __syncthreads();
  uint4 ptot[CH], ntot[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) ptot[c] = ntot[c] = make_uint4(0, 0, 0, 0);
  for (int w = 0; w < nw; w++)
#pragma unroll
    for (int c = 0; c < CH; c++) {
      ptot[c] = add4(ptot[c], part[((w * CH + c) * 2 + 0) * 64 + lane]);
      ntot[c] = add4(ntot[c], part[((w * CH + c) * 2 + 1) * 64 + lane]);
    }
#pragma unroll
  for (int i = 0; i < APW; i++) {
    int a = wave + i * nw;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      size_t off = off0 + (size_t)c * 1024;
      if (off < (size_t)plane_stride && a < A) {
        uint4 mine = v[i][c];
        uint4 shown = own ? *reinterpret_cast<const uint4*>(own + (env_base + a) * plane_stride + off) : mine;
        uint8_t* o = obs + (env_base + a) * 3 * plane_stride + off;
        store16<NT>(o, shown);
        store16<NT>(o + plane_stride, sub4(ptot[c], pos_bits4(mine)));
        store16<NT>(o + 2 * (size_t)plane_stride, sub4(ntot[c], neg_bits4(mine)));
      }
    }
  }
}

// Observations straight from the column form (no dense input planes): a workgroup owns CH KiB of plane
// positions of one environment.  Every wave walks the columns of its agents once, keeps the few entries
// that fall into the tile in registers and counts them into two packed-byte totals in LDS.  After the
// barrier it writes, per agent, the own plane (zeros except those entries, staged through a per-wave LDS
// tile) and total-minus-self for both signs.  The kernel only writes to HBM: 3 bytes per position and agent.
// Every workgroup scans all A*T1 column entries of its environment, so a larger tile (CH) pays when there
// are many of them.
template <int APW, int KT, int CH>
__global__ __launch_bounds__(1024) void observations_columns_kernel(const int16_t* __restrict__ cols, int A, int T1,
                                                                    int plane_stride, uint8_t* __restrict__ obs) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds_u[];
  constexpr int TILE = 1024 * CH;
  unsigned* tot_pos = lds_u;                // [TILE/4] packed bytes
  unsigned* tot_neg = lds_u + TILE / 4;     // [TILE/4]
  const int nw = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned char* own = reinterpret_cast<unsigned char*>(lds_u + TILE / 2) + wave * TILE;   // per-wave tile
  const int e = blockIdx.y;
  const int p0 = blockIdx.x * TILE;
  for (int i = threadIdx.x; i < TILE / 2; i += blockDim.x) lds_u[i] = 0;
  __syncthreads();
  int hit[APW][KT];     // (offset inside the tile) * 2 + (sign > 0), -1 = none
#pragma unroll
  for (int i = 0; i < APW; i++) {
    const int a = wave + i * nw;
#pragma unroll
    for (int k = 0; k < KT; k++) {
      const int t = lane + 64 * k;
      hit[i][k] = -1;
      if (a < A && t < T1) {
        const int c = cols[((size_t)e * A + a) * T1 + t];
        if (c >= 0) {
          const int p = (c >> 1) * T1 + t - p0;
          if (p >= 0 && p < TILE) {
            hit[i][k] = p * 2 + (c & 1);
            atomicAdd(((c & 1) ? tot_pos : tot_neg) + (p >> 2), 1u << ((p & 3) * 8));
          }
        }
      }
    }
  }
  __syncthreads();
  uint4 ptot[CH], ntot[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) {
    ptot[c] = reinterpret_cast<const uint4*>(tot_pos)[c * 64 + lane];
    ntot[c] = reinterpret_cast<const uint4*>(tot_neg)[c * 64 + lane];
  }
  const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < APW; i++) {
    const int a = wave + i * nw;
    if (a >= A) continue;   // wave-uniform
    bool any = false;
#pragma unroll
    for (int k = 0; k < KT; k++) any |= hit[i][k] >= 0;
    uint4 mine[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) mine[c] = zero;
    if (__ballot(any)) {   // wave-uniform: this agent has entries in the tile
#pragma unroll
      for (int c = 0; c < CH; c++) reinterpret_cast<uint4*>(own)[c * 64 + lane] = zero;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < KT; k++)
        if (hit[i][k] >= 0) own[hit[i][k] >> 1] = (hit[i][k] & 1) ? (unsigned char)1 : (unsigned char)0xff;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < CH; c++) mine[c] = reinterpret_cast<const uint4*>(own)[c * 64 + lane];
      __builtin_amdgcn_wave_barrier();
    }
    uint8_t* o = obs + ((size_t)e * A + a) * 3 * plane_stride;
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const size_t off = (size_t)p0 + (size_t)c * 1024 + (size_t)lane * 16;
      if (off < (size_t)plane_stride) {
        store16<true>(o + off, mine[c]);
        store16<true>(o + plane_stride + off, sub4(ptot[c], pos_bits4(mine[c])));
        store16<true>(o + 2 * (size_t)plane_stride + off, sub4(ntot[c], neg_bits4(mine[c])));
      }
    }
  }
}

// Observation of selected agents only: request g asks for agent req_agent[g] (= e*A + a); the plane shown
// as its own is own_planes[g] when given, else its plane in maps.  One workgroup per (1 KiB tile, request).
__global__ __launch_bounds__(64) void observations_req_kernel(const int8_t* __restrict__ maps,
                                                              const int32_t* __restrict__ req_agent,
                                                              const int8_t* __restrict__ own_planes, int A,
                                                              int plane_stride, uint8_t* __restrict__ out) {
  const int g = blockIdx.y;
  const int ga = req_agent[g];
  const int e = ga / A, a = ga - e * A;
  const size_t off = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 16;
  if (off >= (size_t)plane_stride) return;
  uint4 psum = make_uint4(0, 0, 0, 0), nsum = make_uint4(0, 0, 0, 0), mine = make_uint4(0, 0, 0, 0);
  const int8_t* env = maps + (size_t)e * A * plane_stride + off;
  for (int o = 0; o < A; o++) {
    uint4 v = *reinterpret_cast<const uint4*>(env + (size_t)o * plane_stride);
    if (o == a) mine = v;
    else {
      psum = add4(psum, pos_bits4(v));
      nsum = add4(nsum, neg_bits4(v));
    }
  }
  if (own_planes) mine = *reinterpret_cast<const uint4*>(own_planes + (size_t)g * plane_stride + off);
  uint8_t* o = out + (size_t)g * 3 * plane_stride + off;
  *reinterpret_cast<uint4*>(o) = mine;
  *reinterpret_cast<uint4*>(o + plane_stride) = psum;
  *reinterpret_cast<uint4*>(o + 2 * (size_t)plane_stride) = nsum;
}

// Observation of selected agents straight from the column form: request g asks for agent req_agent[g]
// (= e*A + a); the plane shown as its own comes from own_cols when given (an alternative: row own_row[g], or
// row g without own_row; a negative own_row means the agent's own plane), else from the agent's own columns;
// the heat planes sum the other agents' columns.  One workgroup per (tile of up to 16 KiB, request): the
// environment's A*T1 column entries are scanned once per tile, so small planes take a single scan.
#define FLM_REQ_TILE 16384
__global__ __launch_bounds__(256) void observations_req_columns_kernel(const int16_t* __restrict__ cols,
                                                                       const int32_t* __restrict__ req_agent,
                                                                       const int16_t* __restrict__ own_cols,
                                                                       const int32_t* __restrict__ own_row, int A, int T1,
                                                                       int plane_stride, int tile_bytes,
                                                                       const int32_t* __restrict__ out_row,
                                                                       uint8_t* __restrict__ out,
                                                                       const int32_t* __restrict__ n_live, int plane_limit) {
  extern __shared__ __attribute__((aligned(16))) unsigned req_tile[];   // [3][tile_bytes/4]: own plane, positive heat, negative heat (packed bytes)
  const int g = blockIdx.y, tid = threadIdx.x;
  // n_live (optional): the launch was sized by an upper bound; the kernel that counted the requests left their number there
  if (n_live && g >= *n_live) return;
  // out_row (optional): the row of `out` that request g writes (negative: the request is skipped), else row g
  const int dst_row = out_row ? out_row[g] : g;
  if (dst_row < 0) return;
  const int ga = req_agent[g];
  const int e = ga / A, a = ga - e * A;
  const int p0 = blockIdx.x * tile_bytes;
  const int tw = tile_bytes / 4;
  for (int i = tid; i < 3 * tw / 4; i += 256) reinterpret_cast<uint4*>(req_tile)[i] = make_uint4(0u, 0u, 0u, 0u);   // tile_bytes % 16 == 0
  __syncthreads();
  const int16_t* env = cols + (size_t)e * A * T1;
  // the environment's A*T1 column entries, eight per thread in flight at once (no load is conditional)
  const int n_ent = A * T1;
  const float inv_T1 = 1.0f / (float)T1;
  for (int base = 0; base < n_ent; base += 8 * 256) {
    int cv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) cv[u] = env[min(base + u * 256 + tid, n_ent - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = base + u * 256 + tid;
      const int c = cv[u];
      if (i >= n_ent || c < 0) continue;
      int o = (int)((float)i * inv_T1);          // i / T1, corrected below (i < 2^24: the estimate is off by at most one)
      if (o * T1 > i) o--;
      else if ((o + 1) * T1 <= i) o++;
      if (o == a) continue;
      const int p = (c >> 1) * T1 + (i - o * T1) - p0;
      if (p >= 0 && p < tile_bytes) atomicAdd(&req_tile[((c & 1) ? 1 : 2) * tw + (p >> 2)], 1u << ((p & 3) * 8));
    }
  }
  const int orow = own_cols ? (own_row ? own_row[g] : g) : -1;
  const int16_t* src = orow >= 0 ? own_cols + (size_t)orow * T1 : env + (size_t)a * T1;
  for (int t = tid; t < T1; t += 256) {
    const int c = src[t];
    if (c < 0) continue;
    const int p = (c >> 1) * T1 + t - p0;
    if (p >= 0 && p < tile_bytes) reinterpret_cast<unsigned char*>(req_tile)[p] = (c & 1) ? (unsigned char)1 : (unsigned char)0xff;
  }
  __syncthreads();
  // (plane_limit <= plane_stride: the bytes of every plane that are written; the caller keeps the rest)
  const int n16 = min(tile_bytes, plane_limit - p0) / 16;
  uint8_t* o = out + (size_t)dst_row * 3 * plane_stride + p0;
  for (int i = tid; i < n16; i += 256) {
    reinterpret_cast<uint4*>(o)[i] = reinterpret_cast<const uint4*>(req_tile)[i];
    reinterpret_cast<uint4*>(o + plane_stride)[i] = reinterpret_cast<const uint4*>(req_tile + tw)[i];
    reinterpret_cast<uint4*>(o + 2 * (size_t)plane_stride)[i] = reinterpret_cast<const uint4*>(req_tile + 2 * tw)[i];
  }
}

// Rail reordering, src/preprocessing.py:32-59: the rails the agent's own plane visits, in order of first
// visit, become rows 0..ns-1 of all three planes and every other row is cleared.  One workgroup per
// observation, out of place.  The visiting order is a chain of dependent look-ups (next time step = first
// zero of the current rail's row, next rail = the row occupied at that step), so one thread walks it over a
// per-time-step "first occupied row" table that all threads build; the row gather is parallel.
// err[g] = FLM_AGENT_NONTERM where the reference never returns (the current rail stays occupied up to the
// last time step: np.argmax of an all-False slice is 0 and its loop stops advancing).
__global__ __launch_bounds__(256) void reorder_rails_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                            int Rp, int T1, int plane_stride,
                                                            int32_t* __restrict__ err) {
  extern __shared__ int rr_lds[];
  int* first_row = rr_lds;            // [T1] first row with a non-zero entry at time step t
  int* swap = rr_lds + T1;            // [T1 + 1]
  __shared__ int ns_s, bad_s;
  const int g = blockIdx.x, tid = threadIdx.x;
  const uint8_t* src = in + (size_t)g * 3 * plane_stride;
  uint8_t* dst = out + (size_t)g * 3 * plane_stride;
  for (int t = tid; t < T1; t += 256) first_row[t] = 0x7fffffff;
  __syncthreads();
  const int nbytes = Rp * T1;
  for (int w = tid; w * 4 < nbytes; w += 256) {
    uint32_t v = *reinterpret_cast<const uint32_t*>(src + (size_t)w * 4);
    if (v == 0) continue;
    for (int b = 0; b < 4; b++) {
      int i = w * 4 + b;
      if (i < nbytes && ((v >> (8 * b)) & 0xff)) {
        int r = i / T1;
        atomicMin(&first_row[i - r * T1], r);
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    int ns = 0, bad = 0, ts = 0;
    int rail = first_row[0] == 0x7fffffff ? 0 : first_row[0];
    if (src[(size_t)rail * T1] == 0 && T1 > 1) {      // agent not departed (:44-46)
      ts = 1;
      rail = first_row[1] == 0x7fffffff ? 0 : first_row[1];
    }
    while (src[(size_t)rail * T1 + ts] != 0) {          // :49-52
      if (ns > T1) { bad = 1; break; }
      swap[ns++] = rail;
      int k = 0;
      while (ts + k < T1 && src[(size_t)rail * T1 + ts + k] != 0) k++;
      if (ts + k >= T1) { bad = 1; break; }
      ts += k;
      rail = first_row[ts] == 0x7fffffff ? 0 : first_row[ts];
    }
    ns_s = ns;
    bad_s = bad;
    err[g] = bad ? FLM_AGENT_NONTERM : 0;
  }
  __syncthreads();
  const int ns = bad_s ? 0 : ns_s;
  const bool keep = ns == 0;            // empty own plane: the reference leaves all three planes as they are
  for (int p = 0; p < 3; p++) {
    const uint8_t* sp = src + (size_t)p * plane_stride;
    uint8_t* dp = dst + (size_t)p * plane_stride;
    for (int w = tid; w * 4 < plane_stride; w += 256) {
      uint32_t v = 0;
      if (keep) v = *reinterpret_cast<const uint32_t*>(sp + (size_t)w * 4);
      else if (w * 4 < ns * T1) {
        for (int b = 0; b < 4; b++) {
          int i = w * 4 + b, r = i / T1;
          if (r < ns) v |= (uint32_t)sp[(size_t)swap[r] * T1 + (i - r * T1)] << (8 * b);
        }
      }
      *reinterpret_cast<uint32_t*>(dp + (size_t)w * 4) = v;
    }
  }
}

}  // namespace flm

static int launch_columns(const int32_t* path_cell, const int32_t* path_len, const int32_t* agent_flat,
                          const int32_t* pos_cell, const int32_t* target_cell, const int32_t* tpc,
                          const int32_t* node_id, const int32_t* cell_rail, const int32_t* edge_info, int A, int n_paths,
                          int H, int W, int capR, int Rp, int T1, int Lcap, int shift, int16_t* cols, int32_t* err,
                          hipStream_t stream) {
  if (T1 <= 64 * FLM_COL_KT) {
    size_t lds = sizeof(int) * 16 * (size_t)T1;
    hipLaunchKernelGGL(flm::columns_wave_kernel, dim3((n_paths + 3) / 4), dim3(256), lds, stream, path_cell, path_len,
                       agent_flat, pos_cell, target_cell, tpc, node_id, cell_rail, edge_info, A, n_paths, H, W, capR, Rp, T1,
                       Lcap, shift, cols, err);
  } else {   // longer horizons: the sequential formulation, one workgroup per path
    size_t lds = sizeof(int) * (5 * (size_t)T1 + 4);
    hipLaunchKernelGGL(flm::bitmaps_kernel, dim3(n_paths), dim3(256), lds, stream, path_cell, path_len, agent_flat, pos_cell,
                       target_cell, tpc, node_id, cell_rail, edge_info, A, H, W, capR, Rp, T1, 0, Lcap, shift,
                       (int8_t*)nullptr, cols, err);
  }
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_columns(const int32_t* path_cell, const int32_t* path_len, const int32_t* agent_flat,
                           const int32_t* pos_cell, const int32_t* target_cell, const int32_t* tpc,
                           const int32_t* node_id, const int32_t* cell_rail, const int32_t* edge_info, int E, int A,
                           int n_paths, int H, int W, int capR, int Rp, int T1, int Lcap, int shift, int16_t* cols,
                           int32_t* err, void* stream) {
  if (!path_cell || !path_len || !pos_cell || !target_cell || !tpc || !node_id || !cell_rail || !edge_info || !cols || !err)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || n_paths <= 0 || H <= 0 || W <= 0 || capR <= 0 || Rp <= 0 || Rp > 16383 || T1 <= 1 ||
      Lcap <= 0 || shift < 0 || T1 > 8000)
    return FLM_E_ARG;
  if (!agent_flat && n_paths != E * A) return FLM_E_ARG;
  return launch_columns(path_cell, path_len, agent_flat, pos_cell, target_cell, tpc, node_id, cell_rail, edge_info, A,
                        n_paths, H, W, capR, Rp, T1, Lcap, shift, cols, err, (hipStream_t)stream);
}

extern "C" int flm_bitmaps(const int32_t* path_cell, const int32_t* path_len, const int32_t* agent_flat,
                           const int32_t* pos_cell, const int32_t* target_cell, const int32_t* tpc,
                           const int32_t* node_id, const int32_t* cell_rail, const int32_t* edge_info, int E, int A,
                           int n_paths, int H, int W, int capR, int Rp, int T1, int plane_stride, int Lcap, int shift,
                           int16_t* cols_ws, int8_t* planes, int32_t* err, void* stream) {
  if (!path_cell || !path_len || !pos_cell || !target_cell || !tpc || !node_id || !cell_rail || !edge_info || !planes ||
      !err || !cols_ws)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || n_paths <= 0 || H <= 0 || W <= 0 || capR <= 0 || Rp <= 0 || Rp > 16383 || T1 <= 1 ||
      Lcap <= 0 || shift < 0 || plane_stride % 16 != 0 || (long long)plane_stride < (long long)Rp * T1 || T1 > 8000)
    return FLM_E_ARG;
  if (!agent_flat && n_paths != E * A) return FLM_E_ARG;
  int rc = launch_columns(path_cell, path_len, agent_flat, pos_cell, target_cell, tpc, node_id, cell_rail, edge_info, A,
                          n_paths, H, W, capR, Rp, T1, Lcap, shift, cols_ws, err, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(flm::planes_from_columns_kernel, dim3(n_paths), dim3(256), 0, (hipStream_t)stream, cols_ws, T1,
                     plane_stride, planes);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}


template <int APW, int CH, bool NT>
static int launch_obs(const int8_t* maps, const int8_t* own, int E, int A, int plane_stride, uint8_t* obs, int nw,
                      hipStream_t stream) {
  dim3 grid((plane_stride + 1024 * CH - 1) / (1024 * CH), E);
  size_t lds = sizeof(uint4) * 2 * 64 * (size_t)nw * CH;
  hipLaunchKernelGGL((flm::observations_kernel<APW, CH, NT>), grid, dim3(nw * 64), lds, stream, maps, own, A,
                     plane_stride, obs);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

template <int CH, bool NT>
static int launch_obs_apw(const int8_t* maps, const int8_t* own, int E, int A, int plane_stride, uint8_t* obs, int nw,
                          hipStream_t stream) {
  int apw = (A + nw - 1) / nw;
  if (apw <= 1) return launch_obs<1, CH, NT>(maps, own, E, A, plane_stride, obs, nw, stream);
  if (apw <= 2) return launch_obs<2, CH, NT>(maps, own, E, A, plane_stride, obs, nw, stream);
  if (apw <= 4) return launch_obs<4, CH, NT>(maps, own, E, A, plane_stride, obs, nw, stream);
  if (apw <= 7) return launch_obs<7, CH, NT>(maps, own, E, A, plane_stride, obs, nw, stream);
  if (CH * 13 * 4 <= 128) {
    if (apw <= 13) return launch_obs<13, CH, NT>(maps, own, E, A, plane_stride, obs, nw, stream);
    if (apw <= 16 && CH == 1) return launch_obs<16, 1, NT>(maps, own, E, A, plane_stride, obs, nw, stream);
  }
  return FLM_E_ARG;
}

extern "C" int flm_observations(const int8_t* maps, const int8_t* own, int E, int A, int plane_stride, uint8_t* obs,
                                void* stream_) {
  if (!maps || !obs) return FLM_E_ARG;
  if (E <= 0 || E > 65535 || A <= 0 || A > 255 || plane_stride <= 0 || plane_stride % 16 != 0) return FLM_E_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  // 16 waves (fewest agents per wave), 1 KiB per workgroup, non-temporal stores: the fastest of the measured shapes
  const int nw = A < 16 ? A : 16;
  return launch_obs_apw<1, true>(maps, own, E, A, plane_stride, obs, nw, stream);
}

template <int APW, int KT, int CH>
static int launch_obs_cols(const int16_t* cols, int E, int A, int T1, int plane_stride, uint8_t* obs, int nw,
                           hipStream_t stream) {
  dim3 grid((plane_stride + 1024 * CH - 1) / (1024 * CH), E);
  size_t lds = (size_t)CH * (2048 + 1024 * (size_t)nw);
  if (lds > 160 * 1024 - 512) return FLM_E_ARG;
  if (lds > 64 * 1024) FLM_RAISE_LDS_ONCE((flm::observations_columns_kernel<APW, KT, CH>));
  hipLaunchKernelGGL((flm::observations_columns_kernel<APW, KT, CH>), grid, dim3(nw * 64), lds, stream, cols, A, T1,
                     plane_stride, obs);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

template <int KT, int CH>
static int launch_obs_cols_apw(const int16_t* cols, int E, int A, int T1, int plane_stride, uint8_t* obs, int nw,
                               hipStream_t stream) {
  int apw = (A + nw - 1) / nw;
  if (apw <= 1) return launch_obs_cols<1, KT, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (apw <= 2) return launch_obs_cols<2, KT, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (apw <= 4) return launch_obs_cols<4, KT, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (apw <= 8) return launch_obs_cols<8, KT, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (apw <= 16) return launch_obs_cols<16, KT, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  return FLM_E_ARG;
}

template <int CH>
static int launch_obs_cols_kt(const int16_t* cols, int E, int A, int T1, int plane_stride, uint8_t* obs, int nw,
                              hipStream_t stream) {
  int kt = (T1 + 63) / 64;
  if (kt <= 1) return launch_obs_cols_apw<1, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (kt <= 2) return launch_obs_cols_apw<2, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (kt <= 3) return launch_obs_cols_apw<3, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (kt <= 5) return launch_obs_cols_apw<5, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
  return launch_obs_cols_apw<8, CH>(cols, E, A, T1, plane_stride, obs, nw, stream);
}

extern "C" int flm_observations_columns(const int16_t* cols, int E, int A, int Rp, int T1, int plane_stride,
                                        uint8_t* obs, void* stream_) {
  if (!cols || !obs) return FLM_E_ARG;
  if (E <= 0 || E > 65535 || A <= 0 || A > 255 || Rp <= 0 || T1 <= 1 || T1 > 512 || plane_stride % 16 != 0 ||
      (long long)plane_stride < (long long)Rp * T1)
    return FLM_E_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int nw = A < 16 ? A : 16;
  // tile size: 1 KiB while an environment has few column entries, 2 or 4 KiB when scanning them dominates
  const int ch = (long)A * T1 >= 32768 ? 4 : (long)A * T1 >= 12288 ? 2 : 1;
  if (ch >= 4) return launch_obs_cols_kt<4>(cols, E, A, T1, plane_stride, obs, nw, stream);
  if (ch >= 2) return launch_obs_cols_kt<2>(cols, E, A, T1, plane_stride, obs, nw, stream);
  return launch_obs_cols_kt<1>(cols, E, A, T1, plane_stride, obs, nw, stream);
}

extern "C" int flm_observations_req(const int8_t* maps, const int32_t* req_agent, const int8_t* own_planes, int n_req,
                                    int E, int A, int plane_stride, uint8_t* out, void* stream) {
  if (!maps || !req_agent || !out) return FLM_E_ARG;
  if (n_req <= 0 || n_req > 65535 || E <= 0 || A <= 0 || A > 255 || plane_stride <= 0 || plane_stride % 16 != 0)
    return FLM_E_ARG;
  dim3 grid((plane_stride + 1023) / 1024, n_req);
  hipLaunchKernelGGL(flm::observations_req_kernel, grid, dim3(64), 0, (hipStream_t)stream, maps, req_agent, own_planes,
                     A, plane_stride, out);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

static int observations_req_columns_impl(const int16_t* cols, const int32_t* req_agent, const int16_t* own_cols,
                                         const int32_t* own_row, int n_req, const int32_t* n_live, int plane_limit, int E,
                                         int A, int Rp, int T1, int plane_stride, uint8_t* out, void* stream) {
  if (!cols || !req_agent || !out) return FLM_E_ARG;
  if (plane_limit < 0 || plane_limit > plane_stride || plane_limit % 16 != 0) return FLM_E_ARG;
  if (plane_limit == 0) plane_limit = plane_stride;
  if (n_req <= 0 || n_req > 65535 || E <= 0 || A <= 0 || A > 255 || Rp <= 0 || T1 <= 0 || plane_stride % 16 != 0 ||
      (long long)plane_stride < (long long)Rp * T1)
    return FLM_E_ARG;
  const int tile = plane_limit < FLM_REQ_TILE ? plane_limit : FLM_REQ_TILE;        // a multiple of 16
  dim3 grid((plane_limit + tile - 1) / tile, n_req);
  hipLaunchKernelGGL(flm::observations_req_columns_kernel, grid, dim3(256), 3 * (size_t)tile, (hipStream_t)stream, cols,
                     req_agent, own_cols, own_row, A, T1, plane_stride, tile, nullptr, out, n_live, plane_limit);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_observations_req_columns(const int16_t* cols, const int32_t* req_agent, const int16_t* own_cols,
                                            const int32_t* own_row, int n_req, int E, int A, int Rp, int T1,
                                            int plane_stride, uint8_t* out, void* stream) {
  return observations_req_columns_impl(cols, req_agent, own_cols, own_row, n_req, nullptr, 0, E, A, Rp, T1, plane_stride,
                                       out, stream);
}

extern "C" int flm_observations_req_columns_counted(const int16_t* cols, const int32_t* req_agent, const int16_t* own_cols,
                                                    const int32_t* own_row, int n_max, const int32_t* n_live,
                                                    int plane_limit, int E, int A, int Rp, int T1, int plane_stride,
                                                    uint8_t* out, void* stream) {
  if (!n_live) return FLM_E_ARG;
  return observations_req_columns_impl(cols, req_agent, own_cols, own_row, n_max, n_live, plane_limit, E, A, Rp, T1,
                                       plane_stride, out, stream);
}

extern "C" int flm_observations_req_columns_rows(const int16_t* cols, const int32_t* req_agent, const int32_t* out_row,
                                                 int n_req, int E, int A, int Rp, int T1, int plane_stride, uint8_t* out,
                                                 void* stream) {
  if (!cols || !req_agent || !out_row || !out) return FLM_E_ARG;
  if (n_req <= 0 || E <= 0 || A <= 0 || A > 255 || Rp <= 0 || T1 <= 0 || plane_stride % 16 != 0 ||
      (long long)plane_stride < (long long)Rp * T1)
    return FLM_E_ARG;
  const int tile = plane_stride < FLM_REQ_TILE ? plane_stride : FLM_REQ_TILE;      // a multiple of 16
  for (int lo = 0; lo < n_req; lo += 65535) {                // the grid's y extent is 65,535
    const int cnt = n_req - lo < 65535 ? n_req - lo : 65535;
    dim3 grid((plane_stride + tile - 1) / tile, cnt);
    hipLaunchKernelGGL(flm::observations_req_columns_kernel, grid, dim3(256), 3 * (size_t)tile, (hipStream_t)stream, cols,
                       req_agent + lo, nullptr, nullptr, A, T1, plane_stride, tile, out_row + lo, out, nullptr, plane_stride);
    FLM_CHECK_LAUNCH();
  }
  return FLM_OK;
}

extern "C" int flm_reorder_rails(const uint8_t* obs_in, uint8_t* obs_out, int n, int Rp, int T1, int plane_stride,
                                 int32_t* err, void* stream) {
  if (!obs_in || !obs_out || !err || obs_in == obs_out) return FLM_E_ARG;
  if (n <= 0 || Rp <= 0 || T1 <= 0 || plane_stride < Rp * T1 || plane_stride % 16 != 0) return FLM_E_ARG;
  size_t lds = sizeof(int) * (2 * (size_t)T1 + 1);
  if (lds > 60000) return FLM_E_ARG;
  hipLaunchKernelGGL(flm::reorder_rails_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, obs_in, obs_out, Rp, T1,
                     plane_stride, err);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

// ---- flm_bitmaps without a caller-provided workspace and without any allocation -------------------------
// The column-form workspace of flm_bitmaps lives in a file-scope device array.  In host code the array's name
// is only a handle: its device address is looked up once per device (with that device current) and kept.
// FLM_BITMAPS_STATIC_MAX_COLS int16 entries (n_paths * T1 of one call): 4 MiB
static __device__ __align__(256) int16_t scratch[FLM_BITMAPS_STATIC_MAX_COLS];

static std::once_flag scratch_once[64];
static int16_t* scratch_ptr[64];        // per device ordinal; nullptr = the look-up failed

// Device address of this device's copy of `scratch` (each GPU holds its own copy at its own address), looked up
// once per device with that device current.
static int16_t* scratch_for_current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::call_once(scratch_once[dev], [dev] {
    void* p = nullptr;
    scratch_ptr[dev] = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) == hipSuccess ? static_cast<int16_t*>(p) : nullptr;
  });
  return scratch_ptr[dev];
}

extern "C" int flm_bitmaps_static(const int32_t* path_cell, const int32_t* path_len, const int32_t* agent_flat,
                                  const int32_t* pos_cell, const int32_t* target_cell, const int32_t* tpc,
                                  const int32_t* node_id, const int32_t* cell_rail, const int32_t* edge_info, int E, int A,
                                  int n_paths, int H, int W, int capR, int Rp, int T1, int plane_stride, int Lcap,
                                  int shift, int8_t* planes, int32_t* err, void* stream) {
  if (n_paths <= 0 || T1 <= 1) return FLM_E_ARG;
  if ((long long)n_paths * T1 > (long long)FLM_BITMAPS_STATIC_MAX_COLS) return FLM_E_CAPACITY;
  int16_t* ws = scratch_for_current_device();
  if (!ws) return FLM_E_LAUNCH;
  return flm_bitmaps(path_cell, path_len, agent_flat, pos_cell, target_cell, tpc, node_id, cell_rail, edge_info, E, A,
                     n_paths, H, W, capR, Rp, T1, plane_stride, Lcap, shift, ws, planes, err, stream);
}

extern "C" int flm_abi_version(void) { return 4; }
