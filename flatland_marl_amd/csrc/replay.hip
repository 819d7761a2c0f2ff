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
  __shared__ int scratch[16];
  const int tid = threadIdx.x;
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
    if (tid == 0) {
      unsigned acc = 0, b = 0;
      for (; b < 256; b++) {
        if (acc + hist[b] >= need) break;
        acc += hist[b];
      }
      s_prefix = prefix | (b << shift);
      s_need = need - acc;
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    __syncthreads();
  }
  // every key below `prefix` is in; of the keys equal to it, the first `need` in index order
  int n_out = 0, n_tie = 0;
  for (int base = 0; base < size; base += 1024) {
    const int i = base + tid;
    uint32_t k = 0xFFFFFFFFu;
    bool below = false, tie = false;
    if (i < size) {
      k = replay_key(seed, draw, (uint32_t)i);
      below = k < prefix;
      tie = k == prefix;
    }
    int tot_t, tot_o;
    const int rank_t = n_tie + block_flag_scan(tie, scratch, &tot_t);
    const bool take = below || (tie && rank_t < (int)need);
    const int pos = n_out + block_flag_scan(take, scratch, &tot_o);
    if (take && pos < batch) idx_out[pos] = i;
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

}  // namespace flm

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
