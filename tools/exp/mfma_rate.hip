// Rate of v_mfma_f32_32x32x16_bf16 from registers: waves per workgroup and independent accumulators varied.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void rate(const uint4* in, float* out, int iters) {
  bf16x8 a = __builtin_bit_cast(bf16x8, in[threadIdx.x]), b = __builtin_bit_cast(bf16x8, in[threadIdx.x + 256]);
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; n++)
    for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int n = 0; n < NACC; n++)
#pragma unroll
      for (int k = 0; k < 6; k++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; n++)
    for (int r = 0; r < 16; r++) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(const char* name, int wgs, int threads, const uint4* in, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  rate<NACC><<<wgs, threads>>>(in, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<NACC><<<wgs, threads>>>(in, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * (threads / 64) * iters * NACC * 6 * 32768.0;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
  uint4* in;
  float* out;
  hipMalloc(&in, 512 * 16);
  hipMemset(in, 0, 512 * 16);
  hipMalloc(&out, 4096 * 256 * 4);
  run<1>("256 wg x 4 waves, 1 acc chain", 256, 256, in, out);
  run<4>("256 wg x 4 waves, 4 acc chains", 256, 256, in, out);
  run<4>("512 wg x 4 waves, 4 acc chains", 512, 256, in, out);
  run<4>("1024 wg x 4 waves, 4 acc chains", 1024, 256, in, out);
  run<1>("512 wg x 4 waves, 1 acc chain", 512, 256, in, out);
  return 0;
}
