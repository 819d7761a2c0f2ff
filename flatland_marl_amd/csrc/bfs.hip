// Distance-to-target BFS over (rail cell, heading) states, batched over environments.
//
// One wavefront per (environment, distinct target).  The wave keeps its FIFO and its visited bit set
// in LDS; the reverse-move table of the environment is staged once per workgroup in LDS and shared by
// the workgroup's waves.  A BFS level is expanded 64 states at a time: each lane reads one state's
// (up to four) predecessors, claims the unvisited ones with an LDS atomic OR, and the claimed states
// are appended to the FIFO at ballot-prefix positions, so the visiting order is deterministic.  The
// distance is the level number, written straight to HBM (unreachable states keep the 0xFFFF fill).
#include "flm_common.h"


namespace flm {

// slot j of an environment = j-th distinct target in agent order
__global__ __launch_bounds__(256) void target_slots_kernel(const int32_t* __restrict__ target_cell, int A,
                                                           int32_t* __restrict__ tslot, int32_t* __restrict__ n_slots,
                                                           int32_t* __restrict__ slot_target) {
  extern __shared__ __attribute__((aligned(16))) int lds[];
  int* tgt = lds;            // [A]
  int* first = lds + A;      // [A] 1 when the agent is the first with its target
  const int e = blockIdx.x;
  for (int a = threadIdx.x; a < A; a += blockDim.x) tgt[a] = target_cell[(size_t)e * A + a];
  __syncthreads();
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    int f = 1;
    for (int b = 0; b < a; b++)
      if (tgt[b] == tgt[a]) { f = 0; break; }
    first[a] = f;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    int slot = 0, owner = a;
    for (int b = 0; b < a; b++) {
      if (tgt[b] == tgt[a]) { owner = b; break; }
    }
    for (int b = 0; b < owner; b++) slot += first[b];
    tslot[(size_t)e * A + a] = slot;
    if (first[a]) slot_target[(size_t)e * A + slot] = tgt[a];
  }
  if (threadIdx.x == 0) {
    int n = 0;
    for (int b = 0; b < A; b++) n += first[b];
    n_slots[e] = n;
  }
}

template <bool PRED_LDS>
__global__ __launch_bounds__(1024) void distance_bfs_kernel(const uint16_t* __restrict__ pred_all,
                                                           const int32_t* __restrict__ cell_idx_all,
                                                           const int32_t* __restrict__ n_cells_all,
                                                           const int32_t* __restrict__ slot_target,
                                                           const int32_t* __restrict__ n_slots, int A, int HW, int capCells,
                                                           uint16_t* __restrict__ dist_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int e = blockIdx.y;
  const int nw = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ns = n_slots[e];
  if ((int)blockIdx.x * nw >= ns) return;  // whole workgroup idle (uniform)
  const int S = n_cells_all[e] * 4;
  const int capS = capCells * 4;
  const uint16_t* pred_g = pred_all + (size_t)e * capS * 4;

  // LDS carve: [pred: capS*8 B if PRED_LDS][per wave: queue capS*2 B, visited ceil(capS/32)*4 B]
  size_t off = 0;
  const uint2* pred_l = nullptr;
  if (PRED_LDS) {
    uint2* pl = reinterpret_cast<uint2*>(smem);
    for (int s = threadIdx.x; s < S; s += blockDim.x) pl[s] = reinterpret_cast<const uint2*>(pred_g)[s];
    pred_l = pl;
    off = ((size_t)capS * 8 + 15) & ~(size_t)15;
  }
  const int vis_words = (capS + 31) >> 5;
  const size_t per_wave = (((size_t)capS * 2 + 15) & ~(size_t)15) + (((size_t)vis_words * 4 + 15) & ~(size_t)15);
  volatile uint16_t* queue = reinterpret_cast<volatile uint16_t*>(smem + off + per_wave * wave);
  unsigned* vis = reinterpret_cast<unsigned*>(smem + off + per_wave * wave + (((size_t)capS * 2 + 15) & ~(size_t)15));
  for (int i = lane; i < vis_words; i += 64) vis[i] = 0;
  __syncthreads();  // staged table visible; no barrier is used after this point

  const int slot = blockIdx.x * nw + wave;
  if (slot >= ns) return;
  const int tcell = slot_target[(size_t)e * A + slot];
  if (tcell < 0 || tcell >= HW) return;
  const int tidx = cell_idx_all[(size_t)e * HW + tcell];
  if (tidx < 0) return;  // target on an empty cell: nothing reaches it
  uint16_t* dist = dist_all + ((size_t)e * A + slot) * capS;

  int head = 0, tail = 4;
  if (lane < 4) {
    int s = tidx * 4 + lane;
    queue[lane] = (uint16_t)s;
    atomicOr(&vis[s >> 5], 1u << (s & 31));
    dist[s] = 0;
  }
  int level = 0;
  while (head < tail) {
    const int level_end = tail;
    const unsigned d = (unsigned)(level + 1);
    for (int base = head; base < level_end; base += 64) {
      const int i = base + lane;
      const bool active = i < level_end;
      uint2 pv = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      if (active) {
        int s = queue[i];
        pv = PRED_LDS ? pred_l[s] : reinterpret_cast<const uint2*>(pred_g)[s];
      }
      unsigned p[4] = {pv.x & 0xFFFFu, pv.x >> 16, pv.y & 0xFFFFu, pv.y >> 16};
      bool won[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        won[k] = false;
        if (p[k] != 0xFFFFu) {
          unsigned bit = 1u << (p[k] & 31);
          unsigned old = atomicOr(&vis[p[k] >> 5], bit);
          won[k] = !(old & bit);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        unsigned long long m = __ballot(won[k]);
        if (won[k]) {
          queue[tail + __popcll(m & lanemask_lt())] = (uint16_t)p[k];
          dist[p[k]] = (uint16_t)d;
        }
        tail += __popcll(m);
      }
    }
    head = level_end;
    level++;
    if (level >= 0xFFFE) break;  // distances are stored in 16 bits
  }
}

__global__ void shortest_paths_kernel(const uint16_t* __restrict__ grid_all, const int32_t* __restrict__ cell_idx_all,
                                      const uint16_t* __restrict__ dist_all, const uint16_t* __restrict__ succ_all,
                                      const uint16_t* __restrict__ choice_all, const int32_t* __restrict__ tslot,
                                      const int32_t* __restrict__ start_cell, const int32_t* __restrict__ start_dir,
                                      const int32_t* __restrict__ target_cell, const int32_t* __restrict__ status, int E,
                                      int A, int H, int W, int capCells, int max_depth, int Lcap,
                                      int32_t* __restrict__ path_cell, uint32_t* __restrict__ path_meta,
                                      int32_t* __restrict__ path_len) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= E * A) return;
  const int e = g / A;
  const int HW = H * W;
  if (status[g] == 3) {  // DONE_REMOVED: no path
    path_len[g] = -1;
    return;
  }
  const uint16_t* grid = grid_all + (size_t)e * HW;
  const int32_t* cell_idx = cell_idx_all + (size_t)e * HW;
  const uint16_t* dist = dist_all + ((size_t)e * A + tslot[g]) * capCells * 4;
  int32_t* pc = path_cell + (size_t)g * Lcap;
  uint32_t* pm = path_meta + (size_t)g * Lcap;
  const uint16_t* succ = succ_all ? succ_all + (size_t)e * capCells * 4 : nullptr;
  const uint16_t* choice = succ_all && choice_all ? choice_all + (size_t)e * capCells * 16 : nullptr;
  path_len[g] = greedy_path(grid, cell_idx, dist, succ, choice, H, W, start_cell[g], start_dir[g] & 3, target_cell[g], max_depth, pc,
                            pm, Lcap);
}

}  // namespace flm

static const size_t kLdsBudget = 160 * 1024 - 512;

extern "C" int flm_distance_bfs(const uint16_t* pred, const int32_t* cell_idx, const int32_t* n_cells,
                                const int32_t* target_cell, int E, int A, int H, int W, int capCells, int32_t* tslot,
                                int32_t* n_slots, int32_t* slot_target, uint16_t* dist, void* stream_) {
  if (!pred || !cell_idx || !n_cells || !target_cell || !tslot || !n_slots || !slot_target || !dist) return FLM_E_ARG;
  if (E <= 0 || E > 65535 || A <= 0 || A > 8192 || H <= 0 || W <= 0 || capCells <= 0 || capCells * 4 >= 65535)
    return FLM_E_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t capS = (size_t)capCells * 4;
  if (hipMemsetAsync(dist, 0xFF, (size_t)E * A * capS * 2, stream) != hipSuccess) return FLM_E_LAUNCH;
  hipLaunchKernelGGL(flm::target_slots_kernel, dim3(E), dim3(256), sizeof(int) * 2 * (size_t)A, stream, target_cell, A,
                     tslot, n_slots, slot_target);
  FLM_CHECK_LAUNCH();
  const size_t pred_bytes = (capS * 8 + 15) & ~(size_t)15;
  const size_t per_wave = ((capS * 2 + 15) & ~(size_t)15) + ((((capS + 31) / 32) * 4 + 15) & ~(size_t)15);
  bool pred_lds = pred_bytes + 2 * per_wave <= kLdsBudget;
  size_t room = kLdsBudget - (pred_lds ? pred_bytes : 0);
  int nw = (int)(room / per_wave);
  if (nw < 1) return FLM_E_ARG;
  if (nw > 16) nw = 16;
  if (nw > A) nw = A;
  size_t lds = (pred_lds ? pred_bytes : 0) + per_wave * nw;
  dim3 grid((A + nw - 1) / nw, E);
  if (pred_lds) {
    FLM_RAISE_LDS_ONCE(flm::distance_bfs_kernel<true>);
    hipLaunchKernelGGL(flm::distance_bfs_kernel<true>, grid, dim3(nw * 64), lds, stream, pred, cell_idx, n_cells,
                       slot_target, n_slots, A, H * W, capCells, dist);
  } else {
    FLM_RAISE_LDS_ONCE(flm::distance_bfs_kernel<false>);
    hipLaunchKernelGGL(flm::distance_bfs_kernel<false>, grid, dim3(nw * 64), lds, stream, pred, cell_idx, n_cells,
                       slot_target, n_slots, A, H * W, capCells, dist);
  }
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_target_slots(const int32_t* target_cell, int E, int A, int32_t* tslot, int32_t* n_slots,
                                int32_t* slot_target, void* stream) {
  if (!target_cell || !tslot || !n_slots || !slot_target || E <= 0 || A <= 0 || A > 8192) return FLM_E_ARG;
  hipLaunchKernelGGL(flm::target_slots_kernel, dim3(E), dim3(256), sizeof(int) * 2 * (size_t)A, (hipStream_t)stream,
                     target_cell, A, tslot, n_slots, slot_target);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_shortest_paths(const uint16_t* grid, const int32_t* cell_idx, const uint16_t* dist,
                                  const uint16_t* succ, const uint16_t* choice, const int32_t* tslot,
                                  const int32_t* start_cell, const int32_t* start_dir,
                                  const int32_t* target_cell, const int32_t* status, int E, int A, int H, int W,
                                  int capCells, int max_depth, int Lcap, int32_t* path_cell, uint32_t* path_meta,
                                  int32_t* path_len, void* stream) {
  if (!grid || !cell_idx || !dist || !tslot || !start_cell || !start_dir || !target_cell || !status || !path_cell ||
      !path_meta || !path_len)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || H <= 0 || W <= 0 || capCells <= 0 || max_depth < 0 || Lcap < max_depth + 1) return FLM_E_ARG;
  int total = E * A;
  hipLaunchKernelGGL(flm::shortest_paths_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, grid,
                     cell_idx, dist, succ, choice, tslot, start_cell, start_dir, target_cell, status, E, A, H, W, capCells,
                     max_depth, Lcap, path_cell, path_meta, path_len);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
