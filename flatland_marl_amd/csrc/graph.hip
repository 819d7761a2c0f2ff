// Switch-graph construction, one 1024-thread workgroup per environment.
//
// The numbering the sequential definition produces (src/rail_observations.py:410-510 of the
// reference) is reproduced without a sequential pass:
//   * nodes are numbered row-major  -> ballot/prefix scan over the cells;
//   * every (node, cardinal point) "port" that has a connection walks the rail to the next node,
//     independently of all other ports (parallel);
//   * the sequential rule "create the edge unless the far port was already visited" is equivalent to:
//     port p (index 4*node+cp) with far port q creates an edge iff q was not itself processed at or
//     before p (q has no connection, or q > p) and p is the lowest such port reaching q.  Edge ids
//     are the prefix count of creating ports in port order.
#include "flm_common.h"

namespace flm {

struct Walk {
  int end_port;  // 4*node2 + cp2, or -1 when the walk makes no edge
  int len;       // cells strictly between the two nodes
};

// Follows the rail leaving cell `cell` towards heading `dir` until a node is met.
// EMIT: also writes the cells (cell*4+heading) to out[] and lowers cell_rail[] to `rail`.
template <bool EMIT, typename NodeT>
__device__ Walk walk_rail(const uint16_t* grid, const NodeT* node_id, int H, int W, int cell, int dir, int32_t* out,
                          int out_cap, int32_t* cell_rail, int rail) {
  Walk w;
  w.end_port = -1;
  w.len = 0;
  int r = cell / W, c = cell - r * W;
  const int cap_steps = 4 * H * W;  // the sequential definition does not terminate on a cycle of plain cells
  for (int step = 0; step < cap_steps; step++) {
    int nr = r + dr(dir), nc = c + dc(dir);
    if (nr < 0 || nr >= H || nc < 0 || nc >= W) return w;  // leaves the grid: no edge
    int ncell = nr * W + nc;
    int id2 = node_id[ncell];
    if (id2 >= 0) {
      w.end_port = id2 * 4 + ((dir + 2) & 3);
      return w;
    }
    if (EMIT) {
      if (w.len < out_cap) out[w.len] = ncell * 4 + dir;
      atomicMin(reinterpret_cast<unsigned*>(cell_rail) + ncell, (unsigned)rail);
    }
    w.len++;
    r = nr;
    c = nc;
    // next heading: first allowed exit other than the side we came from; North when there is none
    int nib = nibble(grid[ncell], dir) & ~(1 << (3 - ((dir + 2) & 3)));
    int t = nib ? (3 - (31 - __clz(nib))) : 0;
    int tr = r + dr(t), tc = c + dc(t);
    if (tr < 0 || tr >= H || tc < 0 || tc >= W) return w;
    dir = t;
  }
  w.end_port = -1;
  return w;
}

// The same walk over one 32-bit word per cell staged in LDS: transition code | switch number << 16 (0xFFF = none)
// | border << 28 (bit d: heading d leaves the grid from here).  One look-up per step, no row/column bookkeeping.
#define FLM_WORD_NONODE 0xFFFu
template <bool EMIT>
__device__ Walk walk_rail_words(const unsigned* words, int H, int W, int cell, int dir, int32_t* out, int out_cap,
                                int32_t* cell_rail, int rail) {
  Walk w;
  w.end_port = -1;
  w.len = 0;
  unsigned word = words[cell];
  const int cap_steps = 4 * H * W;  // the sequential definition does not terminate on a cycle of plain cells
  for (int step = 0; step < cap_steps; step++) {
    if ((word >> (28 + dir)) & 1u) return w;  // leaves the grid: no edge
    const int ncell = cell + (dir == 0 ? -W : (dir == 1 ? 1 : (dir == 2 ? W : -1)));
    const unsigned nw = words[ncell];
    const unsigned id2 = (nw >> 16) & 0xFFFu;
    if (id2 != FLM_WORD_NONODE) {
      w.end_port = (int)id2 * 4 + ((dir + 2) & 3);
      return w;
    }
    if (EMIT) {
      if (w.len < out_cap) out[w.len] = ncell * 4 + dir;
      atomicMin(reinterpret_cast<unsigned*>(cell_rail) + ncell, (unsigned)rail);
    }
    w.len++;
    // next heading: first allowed exit other than the side we came from; North when there is none (whether it
    // leaves the grid is seen at the top of the next step)
    const int nib = nibble(nw & 0xFFFFu, dir) & ~(1 << (3 - ((dir + 2) & 3)));
    dir = nib ? (3 - (31 - __clz(nib))) : 0;
    cell = ncell;
    word = nw;
  }
  w.end_port = -1;
  return w;
}

// LDS_GRID: the transition codes, node numbers and border flags of the environment are staged in LDS (one 32-bit
// word per cell), so that the rail walks, chains of dependent look-ups, run at LDS latency.
template <bool LDS_GRID>
__global__ __launch_bounds__(1024, 8) void graph_build_kernel(
    const uint16_t* __restrict__ grid_all, int H, int W, int capN, int capR, int capC, int32_t* __restrict__ node_id_all,
    int32_t* __restrict__ node_cell_all, int32_t* __restrict__ edge_info_all, int32_t* __restrict__ edge_cell_off_all,
    int32_t* __restrict__ edge_cells_all, int32_t* __restrict__ cell_rail_all, int32_t* __restrict__ counts_all,
    unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  const int e = blockIdx.x;
  auto stamp = [&](int i) {   // diagnostic only
    if (stamps && threadIdx.x == 0) stamps[((size_t)gridDim.x + blockIdx.x) * 16 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  const int HW = H * W;
  const uint16_t* grid_g = grid_all + (size_t)e * HW;
  int32_t* node_id = node_id_all + (size_t)e * HW;
  int32_t* node_cell = node_cell_all + (size_t)e * capN;
  int32_t* edge_info = edge_info_all + (size_t)e * capR * 5;
  int32_t* edge_cell_off = edge_cell_off_all + (size_t)e * (capR + 1);
  int32_t* edge_cells = edge_cells_all + (size_t)e * capC;
  int32_t* cell_rail = cell_rail_all + (size_t)e * HW;
  int32_t* counts = counts_all + (size_t)e * 4;

  int* scratch = lds;                 // 16 ints
  int* mscan = lds + 16;              // 16 * 8 + 1 ints (multi_flag_counts); 272 reserved
  int* p_end = lds + 16 + 272;        // [4*capN]
  int* p_len = p_end + 4 * capN;      // [4*capN]
  int* p_first = p_len + 4 * capN;    // [4*capN]
  int* conn = p_first + 4 * capN;     // [capN] connection mask per node
  unsigned* words = reinterpret_cast<unsigned*>(conn + capN);     // [HW] when LDS_GRID
  const uint16_t* grid = grid_g;
  const int tid = threadIdx.x, T = blockDim.x;
  int err = 0;

  // ---- nodes: row-major numbering, eight strided passes over the cells per ranking (multi_flag_counts)
  constexpr int KB = 8;
  int n_nodes = 0;
  for (int base0 = 0; base0 < HW; base0 += KB * T) {
    unsigned node_bits = 0;
    unsigned codes[KB];
#pragma unroll
    for (int k = 0; k < KB; k++) codes[k] = grid_g[min(base0 + k * T + tid, HW - 1)];   // unconditional: all in flight at once
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int cell = base0 + k * T + tid;
      if (cell < HW) {
        const unsigned code = codes[k];
        if (LDS_GRID) {
          const int r = cell / W, c = cell - r * W;
          const unsigned border = (r == 0 ? 1u : 0u) | (c == W - 1 ? 2u : 0u) | (r == H - 1 ? 4u : 0u) | (c == 0 ? 8u : 0u);
          words[cell] = code | (border << 28) | (FLM_WORD_NONODE << 16);
        }
        // a switch: some heading has a choice of exits; the diamond crossing counts as one as well
        const bool is_node = code == 0x8421u || __popc(code & 0xFu) > 1 || __popc(code & 0xF0u) > 1 ||
                             __popc(code & 0xF00u) > 1 || __popc(code & 0xF000u) > 1;
        cell_rail[cell] = -1;
        if (is_node) node_bits |= 1u << k;
      }
    }
    const int tot = multi_flag_counts<KB>(node_bits, mscan);
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int cell = base0 + k * T + tid;
      const bool is_node = (node_bits >> k) & 1u;
      const int id = n_nodes + multi_flag_rank(node_bits, k, mscan);
      if (cell < HW) {
        node_id[cell] = (is_node && id < capN) ? id : -1;
        if (LDS_GRID && is_node && id < capN) words[cell] = (words[cell] & 0xF000FFFFu) | ((unsigned)id << 16);
      }
      if (is_node && id < capN) {
        // sides with a connection: where some heading may leave to, and where it comes in from
        const unsigned code = LDS_GRID ? (words[cell] & 0xFFFFu) : (unsigned)grid_g[cell];   // (this thread stored it above)
        unsigned mask = 0;
        if (code == 0x8421u) {
          mask = 0xF;
        } else {
#pragma unroll
          for (int d = 0; d < 4; d++) {
            const int nib = nibble(code, d);
            if (nib) mask |= 1u << ((d + 2) & 3);
#pragma unroll
            for (int t = 0; t < 4; t++)
              if (has_exit(nib, t)) mask |= 1u << t;
          }
        }
        node_cell[id] = cell;
        conn[id] = (int)mask;
      }
    }
    n_nodes += tot;
  }
  if (n_nodes > capN) {
    err |= FLM_GRAPH_ERR_NODES;
    n_nodes = capN;
  }
  __syncthreads();  // node_id / node_cell / conn visible to the whole workgroup
  stamp(1);

  // ---- every connected port walks to the next node
  const int n_ports = n_nodes * 4;
  for (int p = tid; p < n_ports; p += T) {
    p_first[p] = 0x7FFFFFFF;
    p_end[p] = -1;
    p_len[p] = 0;
    int n = p >> 2, cp = p & 3;
    if ((conn[n] >> cp) & 1) {
      Walk w = LDS_GRID ? walk_rail_words<false>(words, H, W, node_cell[n], cp, nullptr, 0, nullptr, 0)
                        : walk_rail<false>(grid, node_id, H, W, node_cell[n], cp, nullptr, 0, nullptr, 0);
      p_end[p] = w.end_port;
      p_len[p] = w.len;
    }
  }
  __syncthreads();
  stamp(2);
  for (int p = tid; p < n_ports; p += T) {
    int q = p_end[p];
    if (q >= 0) {
      bool q_done_before = ((conn[q >> 2] >> (q & 3)) & 1) && q <= p;
      if (q_done_before) p_end[p] = -1;  // not a candidate
      else atomicMin(&p_first[q], p);
    }
  }
  __syncthreads();

  stamp(3);
  // ---- edge ids = prefix count of creating ports; cell offsets = prefix sum of their lengths
  int n_edges = 0, n_cells = 0;
  for (int base = 0; base < n_ports; base += T) {
    int p = base + tid;
    bool creates = false;
    int len = 0;
    if (p < n_ports) {
      int q = p_end[p];
      creates = q >= 0 && p_first[q] == p;
      len = creates ? p_len[p] : 0;
    }
    int tot_e, tot_c;
    int id = n_edges + block_flag_scan(creates, scratch, &tot_e);
    int off = n_cells + block_int_scan(len, scratch, &tot_c);
    if (creates && id < capR) {
      int q = p_end[p];
      int32_t* ei = edge_info + id * 5;
      ei[0] = p >> 2;
      ei[1] = p & 3;
      ei[2] = q >> 2;
      ei[3] = q & 3;
      ei[4] = len;
      edge_cell_off[id] = off;
      int room = capC - off;
      if (LDS_GRID)
        walk_rail_words<true>(words, H, W, node_cell[p >> 2], p & 3, edge_cells + off, room < 0 ? 0 : room, cell_rail, id);
      else
        walk_rail<true>(grid, node_id, H, W, node_cell[p >> 2], p & 3, edge_cells + off, room < 0 ? 0 : room, cell_rail, id);
    }
    n_edges += tot_e;
    n_cells += tot_c;
  }
  stamp(4);
  if (n_edges > capR) err |= FLM_GRAPH_ERR_RAILS;
  if (n_cells > capC) err |= FLM_GRAPH_ERR_CELLS;
  if (tid == 0) {
    if (n_edges <= capR) edge_cell_off[n_edges] = n_cells;
    counts[0] = n_nodes;
    counts[1] = n_edges;
    counts[2] = n_cells;
    counts[3] = err;
  }
}

// Compact numbering of rail cells + (optionally) the reverse-move table, one 1024-thread workgroup per environment.
__global__ __launch_bounds__(1024) void state_index_kernel(const uint16_t* __restrict__ grid_all, int H, int W, int capCells,
                                                           int32_t* __restrict__ cell_idx_all, int32_t* __restrict__ n_cells_all,
                                                           uint16_t* __restrict__ pred_all, int32_t* __restrict__ counts_all,
                                                           int32_t* __restrict__ idx_cell_all) {
  __shared__ int mscan[16 * 16 + 1];
  const int e = blockIdx.x, tid = threadIdx.x, HW = H * W;
  const uint16_t* grid = grid_all + (size_t)e * HW;
  int32_t* cell_idx = cell_idx_all + (size_t)e * HW;
  constexpr int KB = 16;   // strided passes over the cells per ranking
  int n = 0;
  for (int base0 = 0; base0 < HW; base0 += KB * 1024) {
    unsigned rail_bits = 0;
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int cell = base0 + k * 1024 + tid;
      const unsigned code = grid[min(cell, HW - 1)];   // unconditional: all loads in flight at once
      if (cell < HW && code != 0) rail_bits |= 1u << k;
    }
    const int tot = multi_flag_counts<KB>(rail_bits, mscan);
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int cell = base0 + k * 1024 + tid;
      const bool rail = (rail_bits >> k) & 1u;
      const int id = n + multi_flag_rank(rail_bits, k, mscan);
      if (cell < HW) cell_idx[cell] = (rail && id < capCells) ? id : -1;
      if (idx_cell_all && rail && id < capCells) idx_cell_all[(size_t)e * capCells + id] = cell;
    }
    n += tot;
  }
  if (tid == 0) {
    n_cells_all[e] = n > capCells ? capCells : n;
    if (n > capCells) atomicOr(&counts_all[e * 4 + 3], FLM_GRAPH_ERR_STATES);
  }
  if (!pred_all) return;
  __syncthreads();
  uint16_t* pred = pred_all + (size_t)e * capCells * 16;
  // state (cell, h) is reached in one move from (cell - move(h), o) for every o allowed to leave towards h
  for (int cell = tid; cell < HW; cell += 1024) {
    int idx = cell_idx[cell];
    if (idx < 0) continue;
    int r = cell / W, c = cell - r * W;
#pragma unroll
    for (int h = 0; h < 4; h++) {
      uint16_t out[4] = {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF};
      int pr = r - dr(h), pc = c - dc(h);
      if (pr >= 0 && pr < H && pc >= 0 && pc < W) {
        int pcell = pr * W + pc;
        int pidx = cell_idx[pcell];
        unsigned code = grid[pcell];
        if (pidx >= 0) {
          int k = 0;
#pragma unroll
          for (int o = 0; o < 4; o++)
            if (has_exit(nibble(code, o), h)) out[k++] = (uint16_t)(pidx * 4 + o);
        }
      }
      ushort4 v = make_ushort4(out[0], out[1], out[2], out[3]);
      reinterpret_cast<ushort4*>(pred)[idx * 4 + h] = v;
    }
  }
}

}  // namespace flm

extern "C" int flm_graph_build(const uint16_t* grid, int E, int H, int W, int capN, int capR, int capC,
                               int32_t* node_id, int32_t* node_cell, int32_t* edge_info, int32_t* edge_cell_off,
                               int32_t* edge_cells, int32_t* cell_rail, int32_t* counts, void* stream) {
  if (!grid || !node_id || !node_cell || !edge_info || !edge_cell_off || !edge_cells || !cell_rail || !counts)
    return FLM_E_ARG;
  if (E <= 0 || H <= 0 || W <= 0 || capN <= 0 || capN > FLM_MAX_NODES || capR <= 0 || capC <= 0) return FLM_E_ARG;
  const size_t lds_ports = sizeof(int) * (16 + 272 + (size_t)capN * 13);
  const size_t lds_grid = lds_ports + 4 * (size_t)H * W;
  // stage the grid in LDS when two workgroups still fit a CU (80 KB each)
  const bool staged = lds_grid <= 80 * 1024;
  const size_t lds = staged ? lds_grid : lds_ports;
  if (lds > 160 * 1024 - 512) return FLM_E_ARG;
  if (staged) FLM_RAISE_LDS_ONCE(flm::graph_build_kernel<true>);
  else FLM_RAISE_LDS_ONCE(flm::graph_build_kernel<false>);
  if (staged)
    hipLaunchKernelGGL(flm::graph_build_kernel<true>, dim3(E), dim3(1024), lds, (hipStream_t)stream, grid, H, W, capN, capR,
                       capC, node_id, node_cell, edge_info, edge_cell_off, edge_cells, cell_rail, counts, flm_stamps_for(E));
  else
    hipLaunchKernelGGL(flm::graph_build_kernel<false>, dim3(E), dim3(1024), lds, (hipStream_t)stream, grid, H, W, capN, capR,
                       capC, node_id, node_cell, edge_info, edge_cell_off, edge_cells, cell_rail, counts, flm_stamps_for(E));
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_state_index(const uint16_t* grid, int E, int H, int W, int capCells, int32_t* cell_idx,
                               int32_t* n_cells, uint16_t* pred, int32_t* counts, int32_t* idx_cell, void* stream) {
  if (!grid || !cell_idx || !n_cells || !counts) return FLM_E_ARG;   // pred may be NULL: numbering only
  if (E <= 0 || H <= 0 || W <= 0 || capCells <= 0 || capCells * 4 >= 65535) return FLM_E_ARG;
  hipLaunchKernelGGL(flm::state_index_kernel, dim3(E), dim3(1024), 0, (hipStream_t)stream, grid, H, W, capCells,
                     cell_idx, n_cells, pred, counts, idx_cell);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
