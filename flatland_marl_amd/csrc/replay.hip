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
