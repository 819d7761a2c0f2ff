// Batched rollout step on the column form of the occupancy planes.
//
// A plane holds at most one entry per time step, so an agent's plane is kept as T1 int16 "columns":
// rail*2 + (sign > 0), or -1 for an empty column.  Every bookkeeping rule of the training loop (shift by
// one step, delay behind the last train of the next rail, head-on crash checks) reads or moves whole
// columns, so a rollout step touches A*T1*2 bytes per environment instead of A*Rp*T1.
//
// One wave per environment walks the agents in handle order, exactly as the sequential loop does: an
// agent's checks see the planes of lower-numbered agents already updated for this step and those of
// higher-numbered agents not yet.
#include "flm_common.h"

namespace flm {

#define FLM_STEP_KT 8   // T1 <= 64 * FLM_STEP_KT

__device__ __forceinline__ bool on_rail(int c, int rail) { return c >= 0 && (c >> 1) == rail; }
__device__ __forceinline__ int col_dir(int c) { return c < 0 ? 0 : ((c & 1) ? 1 : -1); }
__device__ __forceinline__ int col_enc(int rail, int dir) { return dir == 0 ? -1 : rail * 2 + (dir > 0 ? 1 : 0); }

// np.argmax(row[from:] == 0) for the row `rail` of a plane given by its columns, by the whole wave: 64 columns
// per look, every lane returns the same value (the arguments are the same in every lane)
__device__ __forceinline__ int wave_first_zero_col(const int16_t* row, int rail, int from, int T1) {
  for (int base = from; base < T1; base += 64) {
    const int t = base + (int)threadIdx.x;
    const bool zero = t < T1 && !on_rail(row[t], rail);
    const unsigned long long mk = __ballot(zero);
    if (mk) return base + __ffsll(mk) - 1 - from;
  }
  return 0;
}

struct ColEnv {
  int16_t* cols;            // [A][T1]
  int A, T1;
  const int32_t* status;    // [A]
  const int32_t* tpc;       // [A]
  const int32_t* pos;       // [A] cell on the grid, -1 = not on the grid
  __device__ __forceinline__ int16_t* row(int a) const { return cols + (size_t)a * T1; }
};

// src/rail_observations.py:161-164
__device__ __forceinline__ int rail_dir(const ColEnv& m, int a, int ts, int* dir) {
  const int c = m.row(a)[ts];
  *dir = col_dir(c);
  return c < 0 ? 0 : (c >> 1);
}

// src/rail_observations.py:257-299: the train (other than a, already departed) that leaves `rail` last and its
// expected exit time (0 = nobody).  Ties keep the lowest agent number.
__device__ __forceinline__ int wave_last_train(const ColEnv& m, int a, int rail, int* last) {
  const int lane = threadIdx.x;
  int key = 0;
  for (int o0 = 0; o0 < m.A; o0 += 64) {
    // one lane per train decides whether it is on the rail now (1) or when it leaves its current rail (2) ...
    const int o = o0 + lane;
    int kind = 0;
    if (o < m.A && o != a && m.status[o] != 0) {
      const int16_t* row = m.row(o);
      const int t = m.tpc[o];
      if (on_rail(row[0], rail)) kind = 1;
      else if (t >= 1 && t - 1 < m.T1 && on_rail(row[t - 1], rail)) kind = 2;
    }
    // ... and the wave scans the columns of the few trains that are
    unsigned long long cand = __ballot(kind != 0);
    while (cand) {
      const int l = __ffsll(cand) - 1;
      cand &= cand - 1;
      const int oo = o0 + l;
      const int16_t* row = m.row(oo);
      int ex;
      if (__shfl(kind, l) == 1) {
        ex = wave_first_zero_col(row, rail, 0, m.T1);
      } else {
        const int c0 = row[0];
        const int orail = c0 < 0 ? 0 : (c0 >> 1);
        ex = 0;
        if (orail != rail) ex = wave_first_zero_col(row, orail, 0, m.T1);
        ex += wave_first_zero_col(row, rail, ex, m.T1);
      }
      if (ex > 0) key = max(key, (ex << 12) | (4095 - oo));
    }
  }
  if (key == 0) {
    *last = 0;
    return 0;
  }
  *last = 4095 - (key & 4095);
  return key >> 12;
}

// src/rail_observations.py:192-204
__device__ __forceinline__ bool wave_headon_crash(const ColEnv& m, int a, int rail, int dir) {
  int last;
  const int le = wave_last_train(m, a, rail, &last);
  if (le > 0) {
    // (an exit time past the horizon would be an IndexError in the reference: read it as an empty column)
    const int c = le - 1 < m.T1 ? m.row(last)[le - 1] : -1;
    const int last_dir = on_rail(c, rail) ? col_dir(c) : 0;
    return last_dir != dir;
  }
  return false;
}

// src/rail_observations.py:182-190
__device__ __forceinline__ bool wave_cell_occupied(const ColEnv& m, int a, int cell) {
  bool hit = false;
  for (int o = threadIdx.x; o < m.A; o += 64)
    if (o != a && m.pos[o] == cell) hit = true;
  return __ballot(hit) != 0;
}

// src/rail_observations.py:206-230.  next_cell: the cell the agent's path enters next (-1 = empty path).
__device__ __forceinline__ bool wave_check_crash(const ColEnv& m, int a, bool before_switch, int init_cell, int next_cell) {
  if (m.status[a] == 0) {
    if (wave_cell_occupied(m, a, init_cell)) return true;
    int dir;
    const int rail = rail_dir(m, a, 1, &dir);
    return wave_headon_crash(m, a, rail, dir);
  }
  if (before_switch) {
    int t = m.tpc[a];
    if (t >= m.T1) t = m.T1 - 1;
    int dir;
    const int rail = rail_dir(m, a, t, &dir);
    return wave_headon_crash(m, a, rail, dir);
  }
  if (next_cell >= 0) return wave_cell_occupied(m, a, next_cell);
  return false;
}

// src/rail_observations.py:167-180
__device__ __forceinline__ void wave_delay(const ColEnv& m, int a, int rail, int dir, int delay) {
  const int lane = threadIdx.x;
  int16_t* row = m.row(a);
  int odir;
  const int orail = rail_dir(m, a, 0, &odir);
  const int t = m.tpc[a];
  int z = delay + t;
  if (z > m.T1) z = m.T1;
  int v[FLM_STEP_KT];
#pragma unroll
  for (int k = 0; k < FLM_STEP_KT; k++) {
    const int c = lane + 64 * k;
    int nv = -1;
    if (c < m.T1) {
      if (c >= z) nv = row[c - delay];
      else if (c < t) nv = col_enc(orail, odir);
      else nv = col_enc(rail, dir);   // t <= c < t + delay
    }
    v[k] = nv;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < FLM_STEP_KT; k++) {
    const int c = lane + 64 * k;
    if (c < m.T1) row[c] = (int16_t)v[k];
  }
  __syncthreads();
}

// src/rail_observations.py:232-252
__device__ __forceinline__ void wave_update_bitmaps(const ColEnv& m, int a, bool before_switch) {
  const int lane = threadIdx.x;
  int16_t* row = m.row(a);
  if (before_switch) {
    int t = m.tpc[a];
    if (t >= m.T1) t = m.T1 - 1;
    int dir, last;
    const int rail = rail_dir(m, a, t, &dir);
    const int le = wave_last_train(m, a, rail, &last);
    if (le > 0) {
      const int cur = wave_first_zero_col(row, rail, t, m.T1) + t;
      if (cur <= le) wave_delay(m, a, rail, dir, le + t - cur);
    }
  }
  int v[FLM_STEP_KT];
#pragma unroll
  for (int k = 0; k < FLM_STEP_KT; k++) {
    const int c = lane + 64 * k;
    v[k] = (c + 1 < m.T1) ? row[c + 1] : -1;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < FLM_STEP_KT; k++) {
    const int c = lane + 64 * k;
    if (c < m.T1) row[c] = (int16_t)v[k];
  }
  __syncthreads();
}

// Which of the five cases of the training loop (src/main.py:146-249) each agent is in this step:
// 0 done, 1 ready to depart, 2 before a switch with an action required, 3 action required, 4 otherwise.
// before_switch follows src/rail_observations.py:142-159 (an exhausted or missing path counts as "before a switch").
__global__ void rollout_classify_kernel(const int32_t* __restrict__ status, const int32_t* __restrict__ pos_cell,
                                        const int32_t* __restrict__ action_required,
                                        const int32_t* __restrict__ path_cell, const uint32_t* __restrict__ path_meta,
                                        const int32_t* __restrict__ path_head, const int32_t* __restrict__ path_len,
                                        const int32_t* __restrict__ cell_rail, int n, int A, int H, int W, int Lp,
                                        int32_t* __restrict__ branch, int32_t* __restrict__ before_switch) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const int e = g / A;
  const int st = status[g];
  int bsw = 0;
  if (st == 1) {
    const int head = path_head[g], len = path_len[g];
    if (len >= 0 && head < len) {
      const int32_t* cr = cell_rail + (size_t)e * H * W;
      const int pos = pos_cell[g];
      const int cell = path_cell[(size_t)g * Lp + head];
      const unsigned meta = path_meta[(size_t)g * Lp + head];
      const int act = (meta >> 8) & 0xFF, nh = (meta >> 16) & 3;
      int next_rail = -1;
      if (act == 4) {
        next_rail = cr[cell];
      } else {
        const int r = cell / W + dr(nh), c = cell % W + dc(nh);
        if (r >= 0 && r < H && c >= 0 && c < W) next_rail = cr[r * W + c];
      }
      const int cur_rail = (pos >= 0 && pos < H * W) ? cr[pos] : -1;
      bsw = (cur_rail != -1 && next_rail == -1) ? 1 : 0;
    } else {
      bsw = 1;
    }
  }
  int b;
  if (st >= 2) b = 0;
  else if (st == 0) b = 1;
  else if (bsw && action_required[g]) b = 2;
  else if (action_required[g]) b = 3;
  else b = 4;
  branch[g] = b;
  before_switch[g] = bsw;
}

// The per-agent part of one training-loop step (src/main.py:146-249) for every environment: one wave per
// environment, agents in handle order.  net_in is the stop/go decision (1 = go) of the agents in cases 1
// and 2; alt_slot/best pick the alternative a case-2 agent commits to.
// The walk is a chain of dependent look-ups.  The per-agent scalars it reads (status, time per cell, position, case,
// decision, and the head entry of each agent's path) are fetched for all agents at once into LDS before it starts,
// and its results (action, crash flag, decision, new head of the path) are collected there and stored after it: a
// store inside the walk would have to complete before the next barrier, once per agent.  LDS = the environment's
// columns are staged there too.
template <bool LDS>
__global__ __launch_bounds__(64) void rollout_apply_kernel(
    int16_t* __restrict__ cols_all, const int32_t* __restrict__ status_all, const int32_t* __restrict__ tpc_all,
    const int32_t* __restrict__ pos_all, const int32_t* __restrict__ init_all, const int32_t* __restrict__ branch_all,
    const int32_t* __restrict__ net_in, const int32_t* __restrict__ alt_slot, const int32_t* __restrict__ best_all,
    const int32_t* __restrict__ alt_count, const int32_t* __restrict__ alt_len, const int32_t* __restrict__ alt_cell,
    const uint32_t* __restrict__ alt_meta, const int16_t* __restrict__ alt_cols, int32_t* __restrict__ path_cell,
    uint32_t* __restrict__ path_meta, int32_t* __restrict__ path_head, int32_t* __restrict__ path_len, int A, int W,
    int T1, int Lp, int cap_alt, int col_elems, int32_t* __restrict__ action_out, int32_t* __restrict__ crash_out,
    int32_t* __restrict__ net_out) {
  extern __shared__ __attribute__((aligned(16))) int16_t s_cols[];
  const int e = blockIdx.x, lane = threadIdx.x;
  // the columns (A*T1*2 bytes) go to LDS when they fit, otherwise they stay in global memory (L2-resident)
  int16_t* cols_env = cols_all + (size_t)e * A * T1;
  const int n_cols = A * T1;
  int32_t* s_agent = reinterpret_cast<int32_t*>(s_cols + (LDS ? col_elems : 0));   // [12][A]
  const size_t g0 = (size_t)e * A;
  if (LDS)
    for (int i = lane; i < n_cols; i += 64) s_cols[i] = cols_env[i];
  for (int o = lane; o < A; o += 64) {
    const size_t g = g0 + o;
    const int head = path_head[g], len = path_len[g];
    int cell = -1;
    unsigned meta = 0;
    if (len >= 0 && head < len) {
      cell = path_cell[g * Lp + head];
      meta = path_meta[g * Lp + head];
  }
    s_agent[o] = status_all[g];
    s_agent[A + o] = tpc_all[g];
    s_agent[2 * A + o] = pos_all[g];
    s_agent[3 * A + o] = branch_all[g];
    s_agent[4 * A + o] = net_in[g];
    s_agent[5 * A + o] = head;
    s_agent[6 * A + o] = len;
    s_agent[7 * A + o] = cell;
    s_agent[8 * A + o] = (int32_t)meta;
  }
  __syncthreads();
  ColEnv m{LDS ? s_cols : cols_env, A, T1, s_agent, s_agent + A, s_agent + 2 * A};
  for (int a = 0; a < A; a++) {
    const size_t g = g0 + a;
    const int br = s_agent[3 * A + a];
    int net = s_agent[4 * A + a], action = 0, crash = 0;
    int32_t* pc = path_cell + g * Lp;
    uint32_t* pm = path_meta + g * Lp;
    if (br == 0) {
      for (int c = lane; c < T1; c += 64) m.row(a)[c] = -1;
      __syncthreads();
      net = 0;
    } else if (br == 1) {
      if (net != 0) {
        crash = wave_check_crash(m, a, false, init_all[g], -1) ? 1 : 0;
        if (crash) {
          net = 0;
          action = 4;
        } else {
          wave_update_bitmaps(m, a, false);
          action = 2;   // a departing agent moves forward (src/rail_observations.py:117-118)
        }
      }
    } else if (br == 2) {
      // the agent commits to one of its alternatives: that path replaces its own, read from its first entry
      const int slot = alt_slot[g];
      const int n_alt = slot >= 0 ? alt_count[slot] : 0;
      int len = -1;
      unsigned meta0 = 0;
      if (n_alt > 0) {
        int b = best_all[g];
        if (b < 0 || b >= n_alt) b = 0;
        const size_t q = (size_t)slot * cap_alt + b;
        len = alt_len[q];
        if (len > 0) meta0 = alt_meta[q * Lp];
        for (int c = lane; c < T1; c += 64) m.row(a)[c] = alt_cols[q * T1 + c];
        for (int i = lane; i < len && i < Lp; i += 64) {
          pc[i] = alt_cell[q * Lp + i];
          pm[i] = alt_meta[q * Lp + i];
        }
        if (lane == 0) {
          s_agent[5 * A + a] = 0;
          path_len[g] = len;
        }
        __syncthreads();
      } else {
        net = 0;
      }
      if (net == 0) {
        action = 4;
      } else {
        crash = wave_check_crash(m, a, true, -1, -1) ? 1 : 0;
        if (crash) {
          net = 0;
          action = 4;
        } else {
          if (len <= 0) {
            action = 4;
          } else {
            const int act = (meta0 >> 8) & 0xFF;
            action = (act >= 1 && act <= 3) ? act : 0;
            if (lane == 0) s_agent[5 * A + a] = 1;
          }
          wave_update_bitmaps(m, a, true);
        }
      }
    } else if (br == 3) {
      const int head = s_agent[5 * A + a], len = s_agent[6 * A + a], cell = s_agent[7 * A + a];
      const unsigned meta = (unsigned)s_agent[8 * A + a];
      const bool on_path = len >= 0 && head < len;
      const int act = (meta >> 8) & 0xFF, nh = (meta >> 16) & 3;
      const int next_cell = !on_path ? -1 : (act == 4 ? cell : cell + dr(nh) * W + dc(nh));
      crash = wave_check_crash(m, a, false, -1, next_cell) ? 1 : 0;
      if (crash) {
        net = 0;
        action = 4;
      } else {
        net = 1;
        if (!on_path) {
          action = 4;
        } else {
          action = (act >= 1 && act <= 3) ? act : 0;
          if (lane == 0) s_agent[5 * A + a] = head + 1;
        }
        wave_update_bitmaps(m, a, false);
      }
    } else {
      net = 1;
      wave_update_bitmaps(m, a, false);
    }
    if (lane == 0) {
      s_agent[9 * A + a] = action;
      s_agent[10 * A + a] = crash;
      s_agent[11 * A + a] = net;
    }
    __syncthreads();
  }
  for (int o = lane; o < A; o += 64) {
    path_head[g0 + o] = s_agent[5 * A + o];
    action_out[g0 + o] = s_agent[9 * A + o];
    crash_out[g0 + o] = s_agent[10 * A + o];
    net_out[g0 + o] = s_agent[11 * A + o];
  }
  if (LDS)
    for (int i = lane; i < n_cols; i += 64) cols_env[i] = s_cols[i];
}

__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// One environment step for every environment (agents in handle order, one wave per environment): a train
// leaves its start cell on a move action if that cell is free; at a cell boundary the action is
// interpreted (an impossible left/right falls back to forward, an impossible move stops the train); a
// moving train advances by `speed` per step and enters the next cell when the fraction reaches 1 and the
// cell is free; a train that reaches its target leaves the grid.  Reward: -speed per step for a train that
// has not arrived, 0 on arrival and afterwards, +1 to every agent in the step in which the last one arrives.
//
// Malfunction process (malf_all != nullptr; flatland's malfunction_from_params): before the step every agent
// that is not broken breaks with probability threshold / 2^32 for min + draw % (max - min + 1) + 1 steps and
// remembers whether it was moving; a broken train on the grid ignores its action and stays where it is;
// after the agent's step the counter goes down by one and `moving` is restored when it reaches zero.  The
// draws are a counter hash of (environment seed, step, agent), not flatland's MT19937 stream.
__global__ __launch_bounds__(64) void env_step_kernel(
    const uint16_t* __restrict__ grid_all, const int32_t* __restrict__ action_all, const int32_t* __restrict__ init_all,
    const int32_t* __restrict__ init_dir_all, const int32_t* __restrict__ target_all,
    const double* __restrict__ speed_all, int32_t* __restrict__ status_all, int32_t* __restrict__ pos_all,
    int32_t* __restrict__ dir_all, int32_t* __restrict__ moving_all, double* __restrict__ frac_all,
    int32_t* __restrict__ exit_action_all, int32_t* __restrict__ done_all, int A, int H, int W,
    double* __restrict__ reward_all, int32_t* __restrict__ action_required, int32_t* __restrict__ all_done,
    int32_t* __restrict__ malf_all, int32_t* __restrict__ malf_moving_all, const uint32_t* __restrict__ env_seed,
    uint32_t malf_threshold, int malf_min, int malf_span, int step_index) {
  extern __shared__ int s_pos[];
  const int e = blockIdx.x, lane = threadIdx.x;
  const size_t base = (size_t)e * A;
  const uint16_t* grid = grid_all + (size_t)e * H * W;
  for (int o = lane; o < A; o += 64) s_pos[o] = pos_all[base + o];
  __syncthreads();
  auto cell_free = [&](int cell) {
    bool hit = false;
    for (int o = lane; o < A; o += 64) hit |= s_pos[o] == cell;
    return __ballot(hit) == 0;
  };
  // (new cell valid && transition valid), new heading, new cell for a move action from (pos, dir)
  auto check_action = [&](int pos, int dir, int action, int* nd_out, int* np_out) {
    const int nib = nibble(grid[pos], dir);
    const int num = __popc(nib);
    int nd = dir, tv = -1;
    if (action == 1) {
      nd = dir - 1;
      if (num <= 1) tv = 0;
    } else if (action == 3) {
      nd = dir + 1;
      if (num <= 1) tv = 0;
    }
    nd &= 3;
    if (action == 2 && num == 1) {
      for (int n = 0; n < 4; n++)
        if (has_exit(nib, n)) { nd = n; break; }
      tv = 1;
    }
    const int r = pos / W + dr(nd), c = pos % W + dc(nd);
    const bool inside = r >= 0 && r < H && c >= 0 && c < W;
    const bool cv = inside && grid[r * W + c] != 0;
    if (tv < 0) tv = has_exit(nib, nd);
    *nd_out = nd;
    *np_out = inside ? r * W + c : -1;
    return cv && tv != 0;
  };
  const uint32_t seed = malf_all ? env_seed[e] : 0u;
  const bool already_over = all_done[e] != 0;
  int n_done = 0;
  for (int i = 0; i < A; i++) {
    const size_t g = base + i;
    int status = status_all[g];
    double reward = 0.0;
    const double speed = speed_all[g];
    int pos = s_pos[i], dir = dir_all[g], moving = moving_all[g];
    double frac = frac_all[g];
    int exit_action = exit_action_all[g];
    int done = done_all[g];
    int malf = 0, malf_moving = 0;
    if (malf_all) {
      malf = malf_all[g];
      malf_moving = malf_moving_all[g];
      if (malf < 1) {
        const uint32_t h1 = fmix32(seed ^ fmix32((uint32_t)step_index * 0x9E3779B1u + (uint32_t)i * 0x85EBCA77u + 1u));
        if (h1 < malf_threshold) {
          malf = malf_min + (int)(fmix32(h1 ^ 0x27D4EB2Fu) % (uint32_t)malf_span) + 1;
          malf_moving = moving;
        }
      }
    }
    if (status >= 2) {
      // finished: nothing moves
    } else if (status == 0) {
      const int action = action_all[g];
      if (action >= 1 && action <= 3 && cell_free(init_all[g])) {
        status = 1;
        pos = init_all[g];
        dir = init_dir_all[g];
        moving = 1;
      }
      reward = -speed;
    } else if (malf > 0) {
      reward = -speed;
    } else {
      int action = action_all[g];
      if (fabs(frac) < 1e-9) {
        if (action == 0 && moving) action = 2;
        if (action == 4 && moving) moving = 0;
        if (!moving && action != 0 && action != 4) moving = 1;
        if (moving) {
          bool valid = false;
          if (action != 0 && action != 4) {
            int nd, np;
            if (check_action(pos, dir, action, &nd, &np)) {
              valid = true;
            } else if (action == 1 || action == 3) {
              action = 2;
              valid = check_action(pos, dir, action, &nd, &np);
            }
          }
          if (valid) exit_action = action;
          else moving = 0;
        }
      }
      if (moving) {
        frac += speed;
        if (frac > 1.0 - 1e-9) {
          int nd, np;
          if (check_action(pos, dir, exit_action, &nd, &np) && cell_free(np)) {
            pos = np;
            dir = nd;
            frac = 0.0;
          }
        }
        if (pos == target_all[g]) {
          status = 3;
          pos = -1;
          moving = 0;
          done = 1;
        }
      }
      reward = done ? 0.0 : -speed;
    }
    if (malf >= 1) {
      malf--;
      if (malf == 0) moving = malf_moving;
    }
    n_done += done ? 1 : 0;
    __syncthreads();
    if (lane == 0) {
      s_pos[i] = pos;
      status_all[g] = status;
      pos_all[g] = pos;
      dir_all[g] = dir;
      moving_all[g] = moving;
      frac_all[g] = frac;
      exit_action_all[g] = exit_action;
      done_all[g] = done;
      reward_all[g] = reward;
      action_required[g] = (status == 0 || (status == 1 && fabs(frac) < 1e-9)) ? 1 : 0;
      if (malf_all) {
        malf_all[g] = malf;
        malf_moving_all[g] = malf_moving;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  if (lane == 0) all_done[e] = n_done == A ? 1 : 0;
  // the step that finishes the last agent pays the global reward to every agent
  if (n_done == A && !already_over)
    for (int o = lane; o < A; o += 64) reward_all[base + o] = 1.0;
}

// ---- request lists and the stop/go policy on the device ---------------------------------------------------
// The three kernels below replace what would otherwise be dozens of tiny element-wise launches per step.

// Agents that consult the policy this step, in agent order: ready[] (case 1), sw[] (case 2); counts[0..1] their
// numbers; alt_slot[g] = position of agent g in sw[] or -1; net/best cleared; the batch's view of the agents
// (cell, heading, status) brought up to date for the path kernels.  Two launches over one thread per agent:
// the first counts the two kinds per workgroup (blk[2*b], blk[2*b+1]), the second places every agent after
// the agents of the workgroups before its own, so the lists come out ordered.
__global__ __launch_bounds__(1024) void rollout_requests_count_kernel(
    const int32_t* __restrict__ branch, const int32_t* __restrict__ status, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ heading, const int32_t* __restrict__ start_cell, int n, int32_t* __restrict__ blk,
    int32_t* __restrict__ alt_slot, int32_t* __restrict__ net, int32_t* __restrict__ best, int32_t* __restrict__ b_pos,
    int32_t* __restrict__ b_dir, int32_t* __restrict__ b_status) {
  __shared__ int scratch[16];
  const int g = blockIdx.x * 1024 + threadIdx.x;
  int br = -1;
  if (g < n) {
    br = branch[g];
    const int p = pos[g];
    b_pos[g] = p >= 0 ? p : start_cell[g];
    b_dir[g] = heading[g];
    b_status[g] = status[g];
    net[g] = 0;
    best[g] = 0;
    alt_slot[g] = -1;
  }
  int t1, t2;
  block_flag_scan(br == 1, scratch, &t1);
  block_flag_scan(br == 2, scratch, &t2);
  if (threadIdx.x == 0) {
    blk[2 * blockIdx.x] = t1;
    blk[2 * blockIdx.x + 1] = t2;
  }
}

__global__ __launch_bounds__(1024) void rollout_requests_fill_kernel(const int32_t* __restrict__ branch, int n,
                                                                     const int32_t* __restrict__ blk,
                                                                     int32_t* __restrict__ ready, int32_t* __restrict__ sw,
                                                                     int32_t* __restrict__ counts,
                                                                     int32_t* __restrict__ alt_slot) {
  __shared__ int scratch[16];
  const int g = blockIdx.x * 1024 + threadIdx.x;
  int s1 = 0, s2 = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 1024) {
    s1 += blk[2 * b];
    s2 += blk[2 * b + 1];
  }
  int base1, base2, t1, t2;
  block_int_scan(s1, scratch, &base1);
  block_int_scan(s2, scratch, &base2);
  const int br = g < n ? branch[g] : -1;
  const int r1 = block_flag_scan(br == 1, scratch, &t1);
  const int r2 = block_flag_scan(br == 2, scratch, &t2);
  if (br == 1) ready[base1 + r1] = g;
  if (br == 2) {
    sw[base2 + r2] = g;
    alt_slot[g] = base2 + r2;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    counts[0] = base1 + t1;
    counts[1] = base2 + t2;
  }
}

// After the alternatives of the n_switch agents in sw[] have been computed: an agent whose alternatives raised
// (path or plane) is flagged in agent_err (first error kept) and left without alternatives; row_offset[i] =
// number of alternatives of the agents before i; the rows the network evaluates are listed in row_req /
// row_own: first the n_ready departing agents (own plane: -1), then every alternative (i*K + k) of every
// agent of sw[]; counts[2] = number of rows.
__global__ __launch_bounds__(1024) void rollout_alt_finish_kernel(
    int32_t* __restrict__ alt_count, const int32_t* __restrict__ alt_err, const int32_t* __restrict__ col_err,
    const int32_t* __restrict__ sw, int n_switch, int K, const int32_t* __restrict__ ready, int n_ready,
    int32_t* __restrict__ agent_err, int32_t* __restrict__ row_offset, int32_t* __restrict__ row_req,
    int32_t* __restrict__ row_own, int32_t* __restrict__ counts) {
  __shared__ int scratch[16];
  const int tid = threadIdx.x;
  const int per = (n_switch + 1023) / 1024;
  const int lo = min(n_switch, tid * per), hi = min(n_switch, lo + per);
  int sum = 0;
  for (int i = lo; i < hi; i++) {
    int cnt = alt_count[i];
    const int e0 = alt_err[i];
    int first = e0, bad = e0 != 0;
    int worst = 0;
    for (int k = 0; k < K && k < cnt; k++) {
      const int ce = col_err[i * K + k];
      bad |= ce != 0;
      worst = max(worst, ce);
    }
    if (e0 == 0) first = worst;
    const int g = sw[i];
    if (agent_err[g] == 0 && first != 0) agent_err[g] = first;
    if (bad) cnt = 0;
    alt_count[i] = cnt;
    sum += cnt;
  }
  int total;
  int off = block_int_scan(sum, scratch, &total);
  for (int i = lo; i < hi; i++) {
    const int cnt = alt_count[i];
    row_offset[i] = off;
    for (int k = 0; k < cnt; k++) {
      row_req[n_ready + off + k] = sw[i];
      row_own[n_ready + off + k] = i * K + k;
    }
    off += cnt;
  }
  for (int r = tid; r < n_ready; r += 1024) {
    row_req[r] = ready[r];
    row_own[r] = -1;
  }
  if (tid == 0) {
    row_offset[n_switch] = total;
    counts[2] = n_ready + total;
  }
}

__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t step, uint32_t g, uint32_t k) {
  const uint32_t h = fmix32(seed ^ fmix32(step * 0x9E3779B1u + g * 0x85EBCA77u + k * 0xC2B2AE3Du + 0x165667B1u));
  return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);     // in (0, 1): eps = 0 never explores
}

// Epsilon-greedy decisions from the Q-values of the rows listed by rollout_alt_finish_kernel
// (src/main.py:168-172 for a departing agent, :201-208 for an agent in front of a switch: the best of all
// (alternative, action) pairs, or with probability eps an action and an alternative drawn independently).
// net[g] / best[g] are written for the consulting agents; sel_row / sel_act (one entry per consulting agent,
// departing agents first) name the row whose observation the agent acted on (-1: it had no alternative) and
// the action chosen.  Random numbers are a counter hash of (seed, step, agent).
__global__ void policy_select_kernel(const float* __restrict__ q, const int32_t* __restrict__ ready, int n_ready,
                                     const int32_t* __restrict__ sw, int n_switch,
                                     const int32_t* __restrict__ row_offset, const int32_t* __restrict__ alt_count,
                                     float eps, uint32_t seed, uint32_t step, int32_t* __restrict__ net,
                                     int32_t* __restrict__ best, int32_t* __restrict__ sel_row,
                                     int32_t* __restrict__ sel_act) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_ready) {
    const int g = ready[r];
    int a = q[r * 2 + 1] > q[r * 2] ? 1 : 0;
    if (uniform01(seed, step, g, 0) <= eps) a = uniform01(seed, step, g, 1) < 0.5f ? 1 : 0;
    net[g] = a;
    sel_row[r] = r;
    sel_act[r] = a;
  } else if (r < n_ready + n_switch) {
    const int i = r - n_ready;
    const int g = sw[i];
    const int cnt = alt_count[i];
    int a = 0, b = 0, row = -1;
    if (cnt > 0) {
      const int base = n_ready + row_offset[i];
      float top = q[base * 2];
      int arg = 0;
      for (int j = 1; j < 2 * cnt; j++) {
        const float v = q[base * 2 + j];
        if (v > top) { top = v; arg = j; }
      }
      a = arg & 1;
      b = arg >> 1;
      if (uniform01(seed, step, g, 0) <= eps) {
        a = uniform01(seed, step, g, 1) < 0.5f ? 1 : 0;
        b = min((int)(uniform01(seed, step, g, 2) * cnt), cnt - 1);
      }
      row = base + b;
    }
    net[g] = a;
    best[g] = b;
    sel_row[r] = row;
    sel_act[r] = a;
  }
}

}  // namespace flm

extern "C" int flm_rollout_classify(const int32_t* status, const int32_t* pos_cell, const int32_t* action_required,
                                    const int32_t* path_cell, const uint32_t* path_meta, const int32_t* path_head,
                                    const int32_t* path_len, const int32_t* cell_rail, int E, int A, int H, int W, int Lp,
                                    int32_t* branch, int32_t* before_switch, void* stream) {
  if (!status || !pos_cell || !action_required || !path_cell || !path_meta || !path_head || !path_len || !cell_rail ||
      !branch || !before_switch)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || H <= 0 || W <= 0 || Lp <= 0 || (long long)E * A > 0x7FFFFFFF) return FLM_E_ARG;
  const int n = E * A;
  hipLaunchKernelGGL(flm::rollout_classify_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, status,
                     pos_cell, action_required, path_cell, path_meta, path_head, path_len, cell_rail, n, A, H, W, Lp,
                     branch, before_switch);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_rollout_apply(int16_t* cols, const int32_t* status, const int32_t* tpc, const int32_t* pos_cell,
                                 const int32_t* init_cell, const int32_t* branch, const int32_t* net_in,
                                 const int32_t* alt_slot, const int32_t* best, const int32_t* alt_count,
                                 const int32_t* alt_len, const int32_t* alt_cell, const uint32_t* alt_meta,
                                 const int16_t* alt_cols, int32_t* path_cell, uint32_t* path_meta, int32_t* path_head,
                                 int32_t* path_len, int E, int A, int W, int T1, int Lp, int cap_alt, int32_t* action,
                                 int32_t* crash, int32_t* net_out, void* stream) {
  if (!cols || !status || !tpc || !pos_cell || !init_cell || !branch || !net_in || !alt_slot || !best || !path_cell ||
      !path_meta || !path_head || !path_len || !action || !crash || !net_out)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || A > 512 || W <= 0 || T1 <= 1 || T1 > 64 * FLM_STEP_KT || Lp <= 0 || cap_alt <= 0)
    return FLM_E_ARG;
  // twelve words per agent and up to 32 KiB of columns per environment go to LDS (several environments still share
  // a CU)
  const size_t col_bytes = ((size_t)A * T1 * sizeof(int16_t) + 15) & ~(size_t)15;
  const bool lds = col_bytes <= 32 * 1024;
  const size_t shm = (lds ? col_bytes : 0) + (size_t)12 * A * sizeof(int32_t);
  const int col_elems = (int)(col_bytes / sizeof(int16_t));
#define FLM_APPLY(L)                                                                                                  \
  hipLaunchKernelGGL(flm::rollout_apply_kernel<L>, dim3(E), dim3(64), shm, (hipStream_t)stream, cols, status, tpc,      \
                     pos_cell, init_cell, branch, net_in, alt_slot, best, alt_count, alt_len, alt_cell, alt_meta,      \
                     alt_cols, path_cell, path_meta, path_head, path_len, A, W, T1, Lp, cap_alt, col_elems, action,    \
                     crash, net_out)
  if (lds) FLM_APPLY(true);
  else FLM_APPLY(false);
#undef FLM_APPLY
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_env_step(const uint16_t* grid, const int32_t* action, const int32_t* init_cell,
                            const int32_t* init_dir, const int32_t* target_cell, const double* speed, int32_t* status,
                            int32_t* pos_cell, int32_t* direction, int32_t* moving, double* fraction,
                            int32_t* exit_action, int32_t* done, int E, int A, int H, int W, double* reward,
                            int32_t* action_required, int32_t* all_done, int32_t* malfunction,
                            int32_t* malfunction_moving, const uint32_t* env_seed, uint32_t malfunction_threshold,
                            int min_duration, int max_duration, int step_index, void* stream) {
  if (!grid || !action || !init_cell || !init_dir || !target_cell || !speed || !status || !pos_cell || !direction ||
      !moving || !fraction || !exit_action || !done || !reward || !action_required || !all_done)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || A > 8192 || H <= 0 || W <= 0) return FLM_E_ARG;
  if (malfunction && (!malfunction_moving || !env_seed || min_duration < 0 || max_duration < min_duration ||
                      step_index < 0))
    return FLM_E_ARG;
  hipLaunchKernelGGL(flm::env_step_kernel, dim3(E), dim3(64), sizeof(int) * (size_t)A, (hipStream_t)stream, grid, action,
                     init_cell, init_dir, target_cell, speed, status, pos_cell, direction, moving, fraction, exit_action,
                     done, A, H, W, reward, action_required, all_done, malfunction, malfunction_moving, env_seed,
                     malfunction_threshold, min_duration, max_duration - min_duration + 1, step_index);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_rollout_requests(const int32_t* branch, const int32_t* status, const int32_t* pos_cell,
                                    const int32_t* direction, const int32_t* init_cell, int E, int A, int32_t* ready,
                                    int32_t* sw, int32_t* counts, int32_t* alt_slot, int32_t* net, int32_t* best,
                                    int32_t* batch_pos_cell, int32_t* batch_direction, int32_t* batch_status,
                                    int32_t* workspace, void* stream) {
  if (!branch || !status || !pos_cell || !direction || !init_cell || !ready || !sw || !counts || !alt_slot || !net ||
      !best || !batch_pos_cell || !batch_direction || !batch_status || !workspace)
    return FLM_E_ARG;
  if (E <= 0 || A <= 0 || (long long)E * A > 0x3FFFFFFF) return FLM_E_ARG;
  const int n = E * A, nb = (n + 1023) / 1024;
  hipLaunchKernelGGL(flm::rollout_requests_count_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, branch, status,
                     pos_cell, direction, init_cell, n, workspace, alt_slot, net, best, batch_pos_cell, batch_direction,
                     batch_status);
  FLM_CHECK_LAUNCH();
  hipLaunchKernelGGL(flm::rollout_requests_fill_kernel, dim3(nb), dim3(1024), 0, (hipStream_t)stream, branch, n, workspace,
                     ready, sw, counts, alt_slot);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_rollout_alt_finish(int32_t* alt_count, const int32_t* alt_err, const int32_t* col_err,
                                      const int32_t* sw, int n_switch, int cap_alt, const int32_t* ready, int n_ready,
                                      int32_t* agent_err, int32_t* row_offset, int32_t* row_req, int32_t* row_own,
                                      int32_t* counts, void* stream) {
  if (!alt_count || !alt_err || !col_err || !sw || !agent_err || !row_offset || !row_req || !row_own || !counts ||
      (n_ready > 0 && !ready))
    return FLM_E_ARG;
  if (n_switch <= 0 || cap_alt <= 0 || n_ready < 0 || (long long)n_switch * cap_alt + n_ready > 0x3FFFFFFF)
    return FLM_E_ARG;
  hipLaunchKernelGGL(flm::rollout_alt_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, alt_count, alt_err,
                     col_err, sw, n_switch, cap_alt, ready, n_ready, agent_err, row_offset, row_req, row_own, counts);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_policy_select(const float* q, const int32_t* ready, int n_ready, const int32_t* sw, int n_switch,
                                 const int32_t* row_offset, const int32_t* alt_count, double eps, uint32_t seed,
                                 int step, int32_t* net, int32_t* best, int32_t* sel_row, int32_t* sel_act,
                                 void* stream) {
  if (!q || !net || !best || !sel_row || !sel_act || (n_ready > 0 && !ready) ||
      (n_switch > 0 && (!sw || !row_offset || !alt_count)))
    return FLM_E_ARG;
  if (n_ready < 0 || n_switch < 0 || n_ready + n_switch <= 0 || step < 0) return FLM_E_ARG;
  const int n = n_ready + n_switch;
  hipLaunchKernelGGL(flm::policy_select_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, ready,
                     n_ready, sw, n_switch, row_offset, alt_count, (float)eps, seed, (uint32_t)step, net, best, sel_row,
                     sel_act);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
