// Uniform sampling without replacement from a device-resident replay memory and the gather of the sampled
// transitions (ReplayBuffer.sample of the reference, src/agent.py:159-174: random.sample of the deque, then one
// stacked tensor per field).
//
// Sampling: entry i of the memory gets the 32-bit key hash(seed, draw, i); the sample is the `batch` entries with
// the smallest keys (ties by index), which is a uniformly random subset.  One workgroup finds the key of rank
// `batch` by a four-pass radix select over LDS histograms and then lists the chosen entries in index order:
// no sort of the whole memory, no scratch buffer.
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

__device__ __forceinline__ uint32_t replay_key(uint32_t seed, uint32_t draw, uint32_t i) {
  return replay_mix(seed ^ replay_mix(draw * 0x9E3779B1u + i * 0x85EBCA77u + 0x27D4EB2Fu));
}

__global__ __launch_bounds__(1024) void replay_sample_kernel(int size, int batch, uint32_t seed, uint32_t draw,
                                                             int32_t* __restrict__ idx_out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need;
  __shared__ int scratch[16 * 8 + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  // key of rank `batch` (1-based): narrow the prefix eight bits at a time
  unsigned prefix = 0, need = (unsigned)batch;   // `need` = rank still to find among the keys that match the prefix
  for (int pass = 0; pass < 4; pass++) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < size; i += 1024) {
      const uint32_t k = replay_key(seed, draw, (uint32_t)i);
      if (pass == 0 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // the bin in which the running count reaches `need`: every lane sums four bins, the wave scans the sums
      const unsigned h0 = hist[lane * 4], h1 = hist[lane * 4 + 1], h2 = hist[lane * 4 + 2], h3 = hist[lane * 4 + 3];
      const unsigned sum = h0 + h1 + h2 + h3;
      unsigned inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      const unsigned before = inc - sum;           // keys in the bins below this lane's four
      if (before < need && inc >= need) {           // exactly one lane: counts are non-negative and the total >= need
        unsigned acc = before, b = (unsigned)lane * 4;
        if (acc + h0 < need) { acc += h0; b++;
          if (acc + h1 < need) { acc += h1; b++;
            if (acc + h2 < need) { acc += h2; b++; } } }
        s_prefix = prefix | (b << shift);
        s_need = need - acc;
      }
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
  }
  // every key below `prefix` is in; of the keys equal to it, the first `need` in index order.  Entries are ranked
  // eight strided passes at a time (multi_flag_counts), ties first, then the chosen ones.
  int n_out = 0, n_tie = 0;
  constexpr int KB = 8;
  for (int base0 = 0; base0 < size; base0 += KB * 1024) {
    unsigned below_bits = 0, tie_bits = 0;
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int i = base0 + k * 1024 + tid;
      if (i < size) {
        const uint32_t key = replay_key(seed, draw, (uint32_t)i);
        if (key < prefix) below_bits |= 1u << k;
        if (key == prefix) tie_bits |= 1u << k;
      }
    }
    const int tot_t = multi_flag_counts<KB>(tie_bits, scratch);
    unsigned take_bits = below_bits;
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int rank_t = n_tie + multi_flag_rank(tie_bits, k, scratch);
      if (((tie_bits >> k) & 1u) && rank_t < (int)need) take_bits |= 1u << k;
    }
    const int tot_o = multi_flag_counts<KB>(take_bits, scratch);
#pragma unroll
    for (int k = 0; k < KB; k++) {
      const int pos = n_out + multi_flag_rank(take_bits, k, scratch);
      if (((take_bits >> k) & 1u) && pos < batch) idx_out[pos] = base0 + k * 1024 + tid;
    }
    n_tie += tot_t;
    n_out += tot_o;
  }
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

extern "C" int flm_replay_sample(int size, int batch, uint32_t seed, uint32_t draw, int32_t* idx, void* stream) {
  if (!idx || size <= 0 || batch <= 0 || batch > size) return FLM_E_ARG;
  hipLaunchKernelGGL(flm::replay_sample_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, size, batch, seed, draw, idx);
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
