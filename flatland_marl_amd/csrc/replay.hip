// Uniform sampling without replacement from a device-resident replay memory and the gather of the sampled
// transitions (ReplayBuffer.sample of the reference, src/agent.py:159-174: random.sample of the deque, then one
// stacked tensor per field).
//
// Sampling: a pseudo-random permutation of the entries 0..size-1, keyed by (seed, draw); the sample is the images
// of 0..batch-1, which are distinct because the map is a bijection.  The permutation is a balanced Feistel network
// over the smallest even number of bits that covers `size` (eight rounds of a 32-bit mixing function), walked along
// its cycle until the image falls below `size`: one thread per sample, a few dozen integer operations each, no pass
// over the memory (a selection of the smallest keys over all entries took 24 us per draw at 20,000 entries).
#include "flm_common.h"

namespace flm {

__device__ __forceinline__ uint32_t replay_mix(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// image of x under the keyed permutation of 0 .. 2^(2*hbits) - 1
__device__ __forceinline__ uint32_t replay_permute(uint32_t x, int hbits, uint32_t k0, uint32_t k1) {
  const uint32_t mask = (1u << hbits) - 1u;
  uint32_t left = x >> hbits, right = x & mask;
#pragma unroll
  for (uint32_t r = 0; r < 8; r++) {
    const uint32_t f = replay_mix((right * 0x9E3779B1u + k0 + r * 0x85EBCA77u) ^ k1) & mask;
    const uint32_t t = right;
    right = left ^ f;
    left = t;
  }
  return (left << hbits) | right;
}

// entry j of the sample of (seed, draw) from 0 .. size-1
__device__ __forceinline__ int32_t replay_sampled_index(int size, uint32_t seed, uint32_t draw, int j) {
  const int bits = size > 1 ? 32 - __clz(size - 1) : 1;
  const int hbits = (bits + 1) >> 1;                         // 2^(2*hbits) >= size, and < 4 * size
  const uint32_t k0 = replay_mix(seed ^ 0x27D4EB2Fu), k1 = replay_mix(draw * 0x9E3779B1u + seed);
  uint32_t x = (uint32_t)j;
  do {
    x = replay_permute(x, hbits, k0, k1);
  } while (x >= (uint32_t)size);                             // (ends: the walk returns to j < size at the latest)
  return (int32_t)x;
}

__global__ __launch_bounds__(256) void replay_sample_kernel(int size, int batch, uint32_t seed, uint32_t draw,
                                                            int32_t* __restrict__ idx_out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= batch) return;
  idx_out[j] = replay_sampled_index(size, seed, draw, j);
}

// One workgroup per (sampled transition, 16 KiB piece of a row): both observation rows with 16-byte copies;
// piece 0 also copies the three scalars.
__global__ __launch_bounds__(256) void replay_gather_kernel(const uint8_t* __restrict__ state, const uint8_t* __restrict__ next_state,
                                                            const int64_t* __restrict__ action, const float* __restrict__ reward,
                                                            const float* __restrict__ done, const int32_t* __restrict__ idx,
                                                            long row_bytes, uint8_t* __restrict__ out_state,
                                                            uint8_t* __restrict__ out_next, int64_t* __restrict__ out_action,
                                                            float* __restrict__ out_reward, float* __restrict__ out_done) {
  const int b = blockIdx.y;
  const long src = (long)idx[b] * row_bytes, dst = (long)b * row_bytes;
  const long lo = (long)blockIdx.x * 16384;
  const long hi = lo + 16384 < row_bytes ? lo + 16384 : row_bytes;
  for (long o = lo + (long)threadIdx.x * 16; o < hi; o += 256 * 16) {
    *reinterpret_cast<uint4*>(out_state + dst + o) = *reinterpret_cast<const uint4*>(state + src + o);
    *reinterpret_cast<uint4*>(out_next + dst + o) = *reinterpret_cast<const uint4*>(next_state + src + o);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_action[b] = action[idx[b]];
    out_reward[b] = reward[idx[b]];
    out_done[b] = done[idx[b]];
  }
}

// The same gather for rows of n_planes planes of which only the first used_bytes matter (observations whose rails
// beyond the environments' own are empty): those bytes of every plane are copied, the rest of the output rows is left
// as it is.  One workgroup per (sampled transition, 16 KiB piece of the bytes to copy).
__global__ __launch_bounds__(256) void replay_gather_planes_kernel(const uint8_t* __restrict__ state, const uint8_t* __restrict__ next_state,
                                                                   const int64_t* __restrict__ action, const float* __restrict__ reward,
                                                                   const float* __restrict__ done, const int32_t* __restrict__ idx,
                                                                   long row_bytes, int n_planes, long plane_bytes, long used_bytes,
                                                                   uint8_t* __restrict__ out_state, uint8_t* __restrict__ out_next,
                                                                   int64_t* __restrict__ out_action, float* __restrict__ out_reward,
                                                                   float* __restrict__ out_done) {
  const int b = blockIdx.y;
  const long src = (long)idx[b] * row_bytes, dst = (long)b * row_bytes;
  const long total = used_bytes * n_planes;
  const long lo = (long)blockIdx.x * 16384;
  const long hi = lo + 16384 < total ? lo + 16384 : total;
  for (long o = lo + (long)threadIdx.x * 16; o < hi; o += 256 * 16) {
    const long pl = o / used_bytes, at = pl * plane_bytes + (o - pl * used_bytes);
    *reinterpret_cast<uint4*>(out_state + dst + at) = *reinterpret_cast<const uint4*>(state + src + at);
    *reinterpret_cast<uint4*>(out_next + dst + at) = *reinterpret_cast<const uint4*>(next_state + src + at);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_action[b] = action[idx[b]];
    out_reward[b] = reward[idx[b]];
    out_done[b] = done[idx[b]];
  }
}

// flm_replay_sample and the gather in one launch: every workgroup works out the entry of its transition itself (the
// same few dozen integer operations, on values that are the same for the whole workgroup), so the list of indices
// never has to reach memory before the rows are fetched; it is still written, for callers that want it.
// used_bytes == plane_bytes and n_planes * plane_bytes == row_bytes copies whole rows.
// (pieces of 4 KiB: one 16-byte copy per thread and field, so that a few hundred rows still fill the chip)
constexpr unsigned DRAW_GATHER_PIECE = 4096;
__global__ __launch_bounds__(256) void replay_draw_gather_kernel(
    const uint8_t* __restrict__ state, const uint8_t* __restrict__ next_state, const int64_t* __restrict__ action,
    const float* __restrict__ reward, const float* __restrict__ done, int size, uint32_t seed, uint32_t draw, long row_bytes,
    unsigned n_planes, unsigned plane_bytes, unsigned used_bytes, uint8_t* __restrict__ out_state,
    uint8_t* __restrict__ out_next, int64_t* __restrict__ out_action, float* __restrict__ out_reward,
    float* __restrict__ out_done, int32_t* __restrict__ idx_out) {
  const int b = blockIdx.y;
  const int32_t x = replay_sampled_index(size, seed, draw, b);
  const long src = (long)x * row_bytes, dst = (long)b * row_bytes;
  const unsigned total = used_bytes * n_planes;
  const unsigned lo = blockIdx.x * DRAW_GATHER_PIECE;
  const unsigned hi = lo + DRAW_GATHER_PIECE < total ? lo + DRAW_GATHER_PIECE : total;
  for (unsigned o = lo + threadIdx.x * 16u; o < hi; o += 256u * 16u) {
    const unsigned pl = o / used_bytes;
    const long at = (long)pl * plane_bytes + (o - pl * used_bytes);
    *reinterpret_cast<uint4*>(out_state + dst + at) = *reinterpret_cast<const uint4*>(state + src + at);
    *reinterpret_cast<uint4*>(out_next + dst + at) = *reinterpret_cast<const uint4*>(next_state + src + at);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_action[b] = action[x];
    out_reward[b] = reward[x];
    out_done[b] = done[x];
    idx_out[b] = x;
  }
}

// ---- storing a rollout step's transitions (default replay feeding of the training script, src/main.py:275-278) ----
// Request i of the step (the departing agents, then the agents in front of a switch) is stored when it had a row to
// act on (sel_row[i] >= 0); the stored requests take consecutive ring slots from `head` on in request order; if there
// are more of them than the ring holds, only the last `capacity` are kept (what appending to deque(maxlen) leaves).
// One workgroup: count, then rank; writes slot[i] (-1 = not stored), the concatenated agent list, the three scalars
// of every stored transition (the action after the crash checks, src/main.py:180,225; reward and done of the step)
// and the number of transitions that the training loop counts (all of them, kept or not).
__global__ __launch_bounds__(1024) void replay_store_index_kernel(
    const int32_t* __restrict__ ready, int n_ready, const int32_t* __restrict__ sw, int n_switch,
    const int32_t* __restrict__ sel_row, const int32_t* __restrict__ net, const double* __restrict__ reward,
    const int32_t* __restrict__ done, int head, int capacity, int64_t* __restrict__ ring_action,
    float* __restrict__ ring_reward, float* __restrict__ ring_done, int32_t* __restrict__ slot,
    int32_t* __restrict__ req_out, int32_t* __restrict__ count_out) {
  __shared__ int scratch[16];
  const int tid = threadIdx.x, n = n_ready + n_switch;
  int mine = 0;
  for (int i = tid; i < n; i += 1024) mine += sel_row[i] >= 0 ? 1 : 0;
  int total;
  block_int_scan(mine, scratch, &total);
  const int drop = total > capacity ? total - capacity : 0;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool ok = i < n && sel_row[i] >= 0;
    int tot;
    const int rank = base + block_flag_scan(ok, scratch, &tot);
    base += tot;
    if (i < n) {
      const int g = i < n_ready ? ready[i] : sw[i - n_ready];
      req_out[i] = g;
      int sl = -1;
      if (ok && rank >= drop) {
        sl = (int)(((long)head + rank - drop) % capacity);
        ring_action[sl] = (int64_t)net[g];
        ring_reward[sl] = (float)reward[g];
        ring_done[sl] = done[g] ? 1.f : 0.f;
      }
      slot[i] = sl;
    }
  }
  if (tid == 0) count_out[0] = total;
}

// One workgroup per (16 KiB piece of a row, request): the observation the agent acted on goes to its ring slot.
__global__ __launch_bounds__(256) void replay_store_rows_kernel(const uint8_t* __restrict__ obs, const int32_t* __restrict__ sel_row,
                                                                const int32_t* __restrict__ slot, long row_bytes,
                                                                uint8_t* __restrict__ ring_state) {
  const int i = blockIdx.y;
  const int sl = slot[i];
  if (sl < 0) return;
  const uint8_t* src = obs + (long)sel_row[i] * row_bytes;
  uint8_t* dst = ring_state + (long)sl * row_bytes;
  const long lo = (long)blockIdx.x * 16384;
  const long hi = lo + 16384 < row_bytes ? lo + 16384 : row_bytes;
  for (long o = lo + (long)threadIdx.x * 16; o < hi; o += 256 * 16)
    *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src + o);
}


// ---- the feeding modes that keep an observation per agent between steps -------------------------------------------
// The crash penalty (src/main.py:271-273) and switch-to-switch (src/main.py:280-296) work on curr_obs (the observation
// an agent last acted on) and, switch-to-switch, buffer_obs / buffer_rew (the observation kept since the agent's last
// "go" and the reward collected since).  Per step, in stream order:
//   replay_slots_curr_kernel   every request with a row: curr_obs[agent] = that row, has_curr = 1;
//   replay_slots_index_kernel  one workgroup lists the step's transitions in the order they enter the ring -
//                              penalties (agents ascending: crash flag set and an observation to blame), then either
//                              the default transitions (requests in order) or the matured switch-to-switch ones
//                              (buffer_obs -> curr_obs, reward collected) followed by the stops (curr_obs -> curr_obs) -
//                              with ring slots as deque(maxlen) would leave them, the scalars of each, and where its
//                              two rows come from;
//   replay_slots_rows_kernel   copies those rows into the ring;
//   replay_slots_keep_kernel   switch-to-switch: a request that goes keeps curr_obs in buffer_obs;
//   replay_slots_reward_kernel switch-to-switch: buffer_rew += reward for every agent holding a buffer_obs.
constexpr unsigned SRC_OBS = 0u, SRC_CURR = 1u, SRC_BUF = 2u, SRC_NONE = 3u;   // upper two bits of a row source, index below

__global__ __launch_bounds__(256) void replay_slots_curr_kernel(const int32_t* __restrict__ ready, int n_ready,
                                                                const int32_t* __restrict__ sw, const int32_t* __restrict__ sel_row,
                                                                int i0, const uint8_t* __restrict__ obs, long row_bytes,
                                                                uint8_t* __restrict__ curr_obs, uint8_t* __restrict__ has_curr) {
  const int i = i0 + blockIdx.y;
  const int row = sel_row[i];
  if (row < 0) return;
  const int g = i < n_ready ? ready[i] : sw[i - n_ready];
  const uint8_t* src = obs + (long)row * row_bytes;
  uint8_t* dst = curr_obs + (long)g * row_bytes;
  const long lo = (long)blockIdx.x * 16384;
  const long hi = lo + 16384 < row_bytes ? lo + 16384 : row_bytes;
  for (long o = lo + (long)threadIdx.x * 16; o < hi; o += 256 * 16)
    *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src + o);
  if (blockIdx.x == 0 && threadIdx.x == 0) has_curr[g] = 1;
}

__global__ __launch_bounds__(1024) void replay_slots_index_kernel(
    const int32_t* __restrict__ ready, int n_ready, const int32_t* __restrict__ sw, int n_switch,
    const int32_t* __restrict__ sel_row, const int32_t* __restrict__ net, const double* __restrict__ reward,
    const int32_t* __restrict__ done, const int32_t* __restrict__ crash, int n_agents, int crash_penalty, int switch2switch,
    const uint8_t* __restrict__ has_curr, uint8_t* __restrict__ has_buf, double* __restrict__ buf_rew, int head, int capacity,
    int64_t* __restrict__ ring_action, float* __restrict__ ring_reward, float* __restrict__ ring_done,
    int32_t* __restrict__ desc, int max_t, int32_t* __restrict__ slot_req, int32_t* __restrict__ req_out,
    int32_t* __restrict__ count_out) {
  __shared__ int scratch[16];
  const int tid = threadIdx.x, n = n_ready + n_switch;
  auto agent_of = [&](int i) { return i < n_ready ? ready[i] : sw[i - n_ready]; };
  // how many of each kind
  int c_pen = 0, c_a = 0, c_b = 0;          // penalties; default or matured; stops
  if (crash_penalty)
    for (int g = tid; g < n_agents; g += 1024) c_pen += (crash[g] != 0 && has_curr[g]) ? 1 : 0;
  for (int i = tid; i < n; i += 1024) {
    if (sel_row[i] < 0) continue;
    const int g = agent_of(i);
    if (!switch2switch) {
      c_a++;
    } else {
      c_a += has_buf[g] ? 1 : 0;
      c_b += net[g] == 0 ? 1 : 0;
    }
  }
  int n_pen, n_a, n_b;
  block_int_scan(c_pen, scratch, &n_pen);
  block_int_scan(c_a, scratch, &n_a);
  block_int_scan(c_b, scratch, &n_b);
  const int total = n_pen + n_a + n_b;
  const int drop = total > capacity ? total - capacity : 0;
  // transition t: its slot (or -1), its scalars, its row sources
  auto emit = [&](int t, unsigned src_state, unsigned src_next, int64_t action, float rew, float dn) {
    int sl = -1;
    if (t >= drop) {
      sl = (int)(((long)head + t - drop) % capacity);
      ring_action[sl] = action;
      ring_reward[sl] = rew;
      ring_done[sl] = dn;
    }
    desc[t] = sl;
    desc[max_t + t] = (int32_t)src_state;
    desc[2 * max_t + t] = (int32_t)src_next;
    return sl;
  };
  int base = 0;
  if (crash_penalty)
    for (int g0 = 0; g0 < n_agents; g0 += 1024) {
      const int g = g0 + tid;
      const bool hit = g < n_agents && crash[g] != 0 && has_curr[g];
      int tot;
      const int rank = base + block_flag_scan(hit, scratch, &tot);
      base += tot;
      if (hit) emit(rank, (SRC_CURR << 30) | (unsigned)g, (SRC_CURR << 30) | (unsigned)g, 1, -100.f, 1.f);
    }
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool ok = i < n && sel_row[i] >= 0;
    const int g = i < n ? agent_of(i) : 0;
    const bool first = ok && (!switch2switch || has_buf[g]);
    int tot;
    const int rank = base + block_flag_scan(first, scratch, &tot);
    base += tot;
    if (i < n) {
      int sl = -1;
      if (first && !switch2switch) {
        // the decision stored is the one left by the crash checks (src/main.py:180, 225); the next state is written
        // by the observation kernel afterwards
        sl = emit(rank, (SRC_OBS << 30) | (unsigned)sel_row[i], SRC_NONE << 30, net[g], (float)reward[g], done[g] ? 1.f : 0.f);
      } else if (first) {
        // an observation kept since the previous decision meets the current one (src/main.py:283-287)
        emit(rank, (SRC_BUF << 30) | (unsigned)g, (SRC_CURR << 30) | (unsigned)g, 1, (float)buf_rew[g], done[g] ? 1.f : 0.f);
        has_buf[g] = 0;
        buf_rew[g] = 0.0;
      }
      if (slot_req) slot_req[i] = sl;
      if (req_out) req_out[i] = g;
    }
  }
  if (switch2switch)
    for (int i0 = 0; i0 < n; i0 += 1024) {
      const int i = i0 + tid;
      const int g = i < n ? agent_of(i) : 0;
      const bool stop = i < n && sel_row[i] >= 0 && net[g] == 0;
      int tot;
      const int rank = base + block_flag_scan(stop, scratch, &tot);
      base += tot;
      // stored with action 1, as the script does (src/main.py:289)
      if (stop) emit(rank, (SRC_CURR << 30) | (unsigned)g, (SRC_CURR << 30) | (unsigned)g, 1, (float)reward[g], 0.f);
    }
  for (int t = total + tid; t < max_t; t += 1024) desc[t] = -1;
  if (tid == 0) count_out[0] = total;
}

__global__ __launch_bounds__(256) void replay_slots_rows_kernel(const int32_t* __restrict__ desc, int max_t, int t0,
                                                                const uint8_t* __restrict__ obs, const uint8_t* __restrict__ curr_obs,
                                                                const uint8_t* __restrict__ buf_obs, long row_bytes,
                                                                uint8_t* __restrict__ ring_state, uint8_t* __restrict__ ring_next) {
  const int t = t0 + blockIdx.y;
  const int sl = desc[t];
  if (sl < 0) return;
  auto source = [&](unsigned d) -> const uint8_t* {
    const unsigned kind = d >> 30;
    const long idx = (long)(d & 0x3FFFFFFFu);
    return kind == SRC_OBS ? obs + idx * row_bytes : kind == SRC_CURR ? curr_obs + idx * row_bytes
           : kind == SRC_BUF ? buf_obs + idx * row_bytes : nullptr;
  };
  const uint8_t* s0 = source((unsigned)desc[max_t + t]);
  const uint8_t* s1 = source((unsigned)desc[2 * max_t + t]);
  const long lo = (long)blockIdx.x * 16384;
  const long hi = lo + 16384 < row_bytes ? lo + 16384 : row_bytes;
  for (long o = lo + (long)threadIdx.x * 16; o < hi; o += 256 * 16) {
    if (s0) *reinterpret_cast<uint4*>(ring_state + (long)sl * row_bytes + o) = *reinterpret_cast<const uint4*>(s0 + o);
    if (s1) *reinterpret_cast<uint4*>(ring_next + (long)sl * row_bytes + o) = *reinterpret_cast<const uint4*>(s1 + o);
  }
}

__global__ __launch_bounds__(256) void replay_slots_keep_kernel(const int32_t* __restrict__ ready, int n_ready,
                                                                const int32_t* __restrict__ sw, const int32_t* __restrict__ sel_row,
                                                                int i0, const int32_t* __restrict__ net, long row_bytes,
                                                                const uint8_t* __restrict__ curr_obs, uint8_t* __restrict__ buf_obs,
                                                                uint8_t* __restrict__ has_buf) {
  const int i = i0 + blockIdx.y;
  if (sel_row[i] < 0) return;
  const int g = i < n_ready ? ready[i] : sw[i - n_ready];
  if (net[g] != 1) return;
  const uint8_t* src = curr_obs + (long)g * row_bytes;
  uint8_t* dst = buf_obs + (long)g * row_bytes;
  const long lo = (long)blockIdx.x * 16384;
  const long hi = lo + 16384 < row_bytes ? lo + 16384 : row_bytes;
  for (long o = lo + (long)threadIdx.x * 16; o < hi; o += 256 * 16)
    *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src + o);
  if (blockIdx.x == 0 && threadIdx.x == 0) has_buf[g] = 1;
}

__global__ void replay_slots_reward_kernel(const uint8_t* __restrict__ has_buf, const double* __restrict__ reward, int n_agents,
                                           double* __restrict__ buf_rew) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_agents && has_buf[g]) buf_rew[g] += reward[g];
}

}  // namespace flm

extern "C" int flm_replay_store(const int32_t* ready, int n_ready, const int32_t* sw, int n_switch, const int32_t* sel_row,
                                const void* obs, long row_bytes, const int32_t* net, const double* reward,
                                const int32_t* done, int head, int capacity, void* ring_state, int64_t* ring_action,
                                float* ring_reward, float* ring_done, int32_t* slot, int32_t* req_out, int32_t* count_out,
                                void* stream) {
  if (!sel_row || !obs || !net || !reward || !done || !ring_state || !ring_action || !ring_reward || !ring_done || !slot ||
      !req_out || !count_out || (n_ready > 0 && !ready) || (n_switch > 0 && !sw))
    return FLM_E_ARG;
  if (n_ready < 0 || n_switch < 0 || (long)n_ready + n_switch <= 0 || (long)n_ready + n_switch > 0x3FFFFFFF) return FLM_E_ARG;
  if (capacity <= 0 || head < 0 || head >= capacity || row_bytes <= 0 || (row_bytes & 15)) return FLM_E_ARG;
  if (((uintptr_t)obs | (uintptr_t)ring_state) & 15) return FLM_E_ARG;
  const int n = n_ready + n_switch;
  hipLaunchKernelGGL(flm::replay_store_index_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ready, n_ready, sw,
                     n_switch, sel_row, net, reward, done, head, capacity, ring_action, ring_reward, ring_done, slot, req_out,
                     count_out);
  FLM_CHECK_LAUNCH();
  const unsigned pieces = (unsigned)((row_bytes + 16383) / 16384);
  for (int lo = 0; lo < n; lo += 65535) {                  // the grid's y extent is 65,535
    const int cnt = n - lo < 65535 ? n - lo : 65535;
    hipLaunchKernelGGL(flm::replay_store_rows_kernel, dim3(pieces, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)obs, sel_row + lo, slot + lo, row_bytes, (uint8_t*)ring_state);
    FLM_CHECK_LAUNCH();
  }
  return FLM_OK;
}

extern "C" long flm_replay_store_slots_workspace(int n_agents, int n_requests) {
  if (n_agents <= 0 || n_requests < 0) return 0;
  return 3L * ((long)n_agents + 2L * n_requests);
}

extern "C" int flm_replay_store_slots(const int32_t* ready, int n_ready, const int32_t* sw, int n_switch,
                                      const int32_t* sel_row, const void* obs, long row_bytes, const int32_t* net,
                                      const double* reward, const int32_t* done, const int32_t* crash, int n_agents,
                                      int crash_penalty, int switch2switch, void* curr_obs, uint8_t* has_curr, void* buf_obs,
                                      uint8_t* has_buf, double* buf_rew, int head, int capacity, void* ring_state,
                                      void* ring_next, int64_t* ring_action, float* ring_reward, float* ring_done,
                                      int32_t* workspace, int32_t* slot, int32_t* req_out, int32_t* count_out, void* stream) {
  const long n_l = (long)n_ready + n_switch;
  if (!net || !reward || !done || !crash || !curr_obs || !has_curr || !ring_state || !ring_next || !ring_action ||
      !ring_reward || !ring_done || !workspace || !count_out)
    return FLM_E_ARG;
  if (n_ready < 0 || n_switch < 0 || n_l > 0x1FFFFFFF || n_agents <= 0 || n_agents > 0x1FFFFFFF) return FLM_E_ARG;
  if (n_l > 0 && (!sel_row || !obs || !slot || !req_out || (n_ready > 0 && !ready) || (n_switch > 0 && !sw))) return FLM_E_ARG;
  if (switch2switch && (!buf_obs || !has_buf || !buf_rew)) return FLM_E_ARG;
  if (capacity <= 0 || head < 0 || head >= capacity || row_bytes <= 0 || (row_bytes & 15)) return FLM_E_ARG;
  if (((uintptr_t)obs | (uintptr_t)ring_state | (uintptr_t)ring_next | (uintptr_t)curr_obs | (uintptr_t)buf_obs) & 15)
    return FLM_E_ARG;
  const int n = (int)n_l;
  const long max_t_l = (long)n_agents + 2 * n_l;
  if (max_t_l > 0x3FFFFFFF) return FLM_E_ARG;
  const int max_t = (int)max_t_l;
  hipStream_t st = (hipStream_t)stream;
  const unsigned pieces = (unsigned)((row_bytes + 16383) / 16384);
  for (int lo = 0; lo < n; lo += 65535) {                  // the grid's y extent is 65,535
    const int cnt = n - lo < 65535 ? n - lo : 65535;
    hipLaunchKernelGGL(flm::replay_slots_curr_kernel, dim3(pieces, (unsigned)cnt), dim3(256), 0, st, ready, n_ready, sw,
                       sel_row, lo, (const uint8_t*)obs, row_bytes, (uint8_t*)curr_obs, has_curr);
    FLM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(flm::replay_slots_index_kernel, dim3(1), dim3(1024), 0, st, ready, n_ready, sw, n_switch, sel_row, net,
                     reward, done, crash, n_agents, crash_penalty, switch2switch, has_curr, has_buf, buf_rew, head, capacity,
                     ring_action, ring_reward, ring_done, workspace, max_t, slot, req_out, count_out);
  FLM_CHECK_LAUNCH();
  // at most this many transitions: every agent a penalty, every request one (default) or two (switch-to-switch)
  const int most = (crash_penalty ? n_agents : 0) + (switch2switch ? 2 * n : n);
  for (int lo = 0; lo < most; lo += 65535) {
    const int cnt = most - lo < 65535 ? most - lo : 65535;
    hipLaunchKernelGGL(flm::replay_slots_rows_kernel, dim3(pieces, (unsigned)cnt), dim3(256), 0, st, workspace, max_t, lo,
                       (const uint8_t*)obs, (const uint8_t*)curr_obs, (const uint8_t*)buf_obs, row_bytes, (uint8_t*)ring_state,
                       (uint8_t*)ring_next);
    FLM_CHECK_LAUNCH();
  }
  if (switch2switch) {
    for (int lo = 0; lo < n; lo += 65535) {
      const int cnt = n - lo < 65535 ? n - lo : 65535;
      hipLaunchKernelGGL(flm::replay_slots_keep_kernel, dim3(pieces, (unsigned)cnt), dim3(256), 0, st, ready, n_ready, sw,
                         sel_row, lo, net, row_bytes, (const uint8_t*)curr_obs, (uint8_t*)buf_obs, has_buf);
      FLM_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(flm::replay_slots_reward_kernel, dim3((unsigned)((n_agents + 255) / 256)), dim3(256), 0, st, has_buf,
                       reward, n_agents, buf_rew);
    FLM_CHECK_LAUNCH();
  }
  return FLM_OK;
}

extern "C" int flm_replay_sample(int size, int batch, uint32_t seed, uint32_t draw, int32_t* idx, void* stream) {
  if (!idx || size <= 0 || batch <= 0 || batch > size) return FLM_E_ARG;
  hipLaunchKernelGGL(flm::replay_sample_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, size,
                     batch, seed, draw, idx);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_replay_gather(const void* state, const void* next_state, const int64_t* action, const float* reward,
                                 const float* done, const int32_t* idx, int batch, long row_bytes, void* out_state,
                                 void* out_next, int64_t* out_action, float* out_reward, float* out_done, void* stream) {
  if (!state || !next_state || !action || !reward || !done || !idx || !out_state || !out_next || !out_action || !out_reward ||
      !out_done)
    return FLM_E_ARG;
  if (batch <= 0 || batch > 65535 || row_bytes <= 0 || (row_bytes & 15)) return FLM_E_ARG;
  if (((uintptr_t)state | (uintptr_t)next_state | (uintptr_t)out_state | (uintptr_t)out_next) & 15) return FLM_E_ARG;
  dim3 grid((unsigned)((row_bytes + 16383) / 16384), (unsigned)batch);
  hipLaunchKernelGGL(flm::replay_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)state,
                     (const uint8_t*)next_state, action, reward, done, idx, row_bytes, (uint8_t*)out_state, (uint8_t*)out_next,
                     out_action, out_reward, out_done);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_replay_draw_gather(const void* state, const void* next_state, const int64_t* action, const float* reward,
                                      const float* done, int size, uint32_t seed, uint32_t draw, int batch, long row_bytes,
                                      int n_planes, long plane_bytes, long used_bytes, void* out_state, void* out_next,
                                      int64_t* out_action, float* out_reward, float* out_done, int32_t* idx, void* stream) {
  if (!state || !next_state || !action || !reward || !done || !idx || !out_state || !out_next || !out_action || !out_reward ||
      !out_done)
    return FLM_E_ARG;
  if (size <= 0 || batch <= 0 || batch > size || batch > 65535) return FLM_E_ARG;
  if (row_bytes <= 0 || (row_bytes & 15) || n_planes <= 0 || plane_bytes <= 0 || (plane_bytes & 15) || used_bytes <= 0 ||
      (used_bytes & 15) || used_bytes > plane_bytes || (long)n_planes * plane_bytes > row_bytes || row_bytes >= (1L << 31))
    return FLM_E_ARG;
  if (((uintptr_t)state | (uintptr_t)next_state | (uintptr_t)out_state | (uintptr_t)out_next) & 15) return FLM_E_ARG;
  dim3 grid((unsigned)((used_bytes * n_planes + flm::DRAW_GATHER_PIECE - 1) / flm::DRAW_GATHER_PIECE), (unsigned)batch);
  hipLaunchKernelGGL(flm::replay_draw_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)state,
                     (const uint8_t*)next_state, action, reward, done, size, seed, draw, row_bytes, (unsigned)n_planes,
                     (unsigned)plane_bytes, (unsigned)used_bytes, (uint8_t*)out_state, (uint8_t*)out_next, out_action,
                     out_reward, out_done, idx);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}

extern "C" int flm_replay_gather_planes(const void* state, const void* next_state, const int64_t* action, const float* reward,
                                        const float* done, const int32_t* idx, int batch, long row_bytes, int n_planes,
                                        long plane_bytes, long used_bytes, void* out_state, void* out_next,
                                        int64_t* out_action, float* out_reward, float* out_done, void* stream) {
  if (!state || !next_state || !action || !reward || !done || !idx || !out_state || !out_next || !out_action || !out_reward ||
      !out_done)
    return FLM_E_ARG;
  if (batch <= 0 || batch > 65535 || row_bytes <= 0 || (row_bytes & 15) || n_planes <= 0 || plane_bytes <= 0 ||
      (plane_bytes & 15) || used_bytes <= 0 || (used_bytes & 15) || used_bytes > plane_bytes ||
      (long)n_planes * plane_bytes > row_bytes)
    return FLM_E_ARG;
  if (((uintptr_t)state | (uintptr_t)next_state | (uintptr_t)out_state | (uintptr_t)out_next) & 15) return FLM_E_ARG;
  dim3 grid((unsigned)((used_bytes * n_planes + 16383) / 16384), (unsigned)batch);
  hipLaunchKernelGGL(flm::replay_gather_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)state,
                     (const uint8_t*)next_state, action, reward, done, idx, row_bytes, n_planes, plane_bytes, used_bytes,
                     (uint8_t*)out_state, (uint8_t*)out_next, out_action, out_reward, out_done);
  FLM_CHECK_LAUNCH();
  return FLM_OK;
}
