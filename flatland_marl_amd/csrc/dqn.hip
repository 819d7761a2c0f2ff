// Dueling Double-DQN of flatland-marl on gfx950: fp32 values and fp32 accumulation throughout.  The large matrix
// products run on the bf16 matrix cores with every fp32 operand split into three bf16 terms (gemm_x3_kernel: the
// result has the accuracy of an fp32 multiply-add); the convolution from bytes and the small-batch kernels use
// fp32 instructions (v_mfma_f32_32x32x2_f32, v_fma_f32).
//
// Network (src/model.py:11-43 of the reference), H3 = 3*max_rails rows, T1 = depth+1 columns:
//   x[B,1,H3,T1] -> Conv2d(1,64,(1,T1)) + ReLU  == per row a dot product with 64 filters
//                -> flatten as [c*H3 + h]       -> fc1_adv, fc1_val (64*H3 -> 512) + ReLU
//                -> fc2_adv (512 -> NA), fc2_val (512 -> 1) -> q = val + adv - mean(adv)
// NA = the reference's action_space (2 in src/main.py:92), at most FLM_DQN_MAX_ACTIONS.
// Parameters live in one flat fp32 buffer in the order of the module's parameters():
//   conv1.weight[64,T1] conv1.bias[64] fc1_adv.weight[512,K1] fc1_adv.bias[512] fc1_val.weight[512,K1]
//   fc1_val.bias[512] fc2_adv.weight[NA,512] fc2_adv.bias[NA] fc2_val.weight[1,512] fc2_val.bias[1],  K1 = 64*H3.
#include <cstdlib>
#include <algorithm>
#include <type_traits>

#include "flm_common.h"

namespace flm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DqnLayout {
  int H3, T1, NA;
  long K1;
  long conv_w, conv_b, a1_w, a1_b, v1_w, v1_b, a2_w, a2_b, v2_w, v2_b, total;
  __host__ __device__ DqnLayout(int h3, int t1, int na) : H3(h3), T1(t1), NA(na) {
    K1 = 64L * h3;
    conv_w = 0;
    conv_b = conv_w + 64L * t1;
    a1_w = conv_b + 64;
    a1_b = a1_w + 512 * K1;
    v1_w = a1_b + 512;
    v1_b = v1_w + 512 * K1;
    a2_w = v1_b + 512;
    a2_b = a2_w + 512L * na;
    v2_w = a2_b + na;
    v2_b = v2_w + 512;
    total = v2_b + 1;
  }
};

// ---- tiled matrix product ---------------------------------------------------------------------------
// C[M,N] (+)= A[M,K] * B[K,N].  A is addressed A[m*lda + k] when A_KC (k contiguous) else A[k*lda + m];
// B is B[n*ldb + k] when B_KC else B[k*ldb + n].  Workgroup tile 128x128.  The z index of the grid selects a
// batch / K-slice through the z strides.
enum { EPI_PLAIN = 0, EPI_CONV = 1, EPI_MASK = 2 };

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;   // per n, or nullptr
  const float* mask;   // EPI_MASK: same indexing as C; output kept where mask > 0
  int M, N, K;
  long lda, ldb, ldc;
  long a_zs, b_zs, c_zs;
  int relu, accumulate;
  int H3;              // EPI_CONV: rows are (b,h) with h < H3; C index = b*N*H3 + n*H3 + h
  int ksplit;          // > 0: blockIdx.z = group*nks + slice; the slice selects the K range
                       // [slice*ksplit, min(K, (slice+1)*ksplit)) and C += slice*c_zs; the group (groups > 1)
                       // selects an independent product: A += group*a_zs, B += group*b_zs, C += group*c_gs
  int groups;
  long c_gs;
  int x_slowest;       // order in which workgroups are dealt to the XCDs (split-operand kernel): see there
  int b_break_k;       // > 0 (B not k-contiguous only): rows k >= b_break_k of B start b_break_off floats further on
  long b_break_off;    // (two [K/2, N] blocks that are not adjacent in memory multiplied as one [K, N] matrix)
  // Segment mapping (kernels with SEG != 0): an index i of the product stands for element
  // seg(i) = (i / b_seg_len) * b_seg_stride + i % b_seg_len of a wider row in memory (b_seg_len a multiple of 4, the
  // index range below 65536; b_seg_magic = ceil(2^32 / b_seg_len) turns the division into a multiply).  SEG 1: the k of
  // a k-contiguous B, B[n*ldb + seg(k)]; SEG 2: the n of a row-contiguous B, B[k*ldb + seg(n)]; SEG 3: the n of C,
  // C[m*ldc + seg(n)] (plain epilogue).
  int b_seg_len, b_seg_stride;
  unsigned b_seg_magic;
};

#define GBM 128
#define GBN 128

// ---- the product on the bf16 matrix cores, fp32 operands split three ways -----------------------------------
// An fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 significant bits, taken by truncation:
// hi = top half of the word, mid = top half of x - hi, lo = x - hi - mid).  Of the nine partial products of
// (a_hi + a_mid + a_lo) * (b_hi + b_mid + b_lo) the six of order 2^-16 and above are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16; every partial product is exact, and what is left out (a_mid*b_lo, a_lo*b_mid,
// a_lo*b_lo) is below 2^-21 of the product, 2^-23 on average: the rounding of an fp32 multiply-add.  Six bf16
// instructions of 32 cycles cover the sixteen k of eight fp32 instructions of 64 cycles.  Infinities and NaNs
// in the operands give NaN.
// Workgroup tile 128x128, 8 waves as 2x4, each wave 2x1 tiles of 32x32 (at most 128 registers: four waves per
// SIMD with two workgroups on a CU).  Memory is read 32 k at a time into
// registers (two fetches in flight), LDS is filled 16 k at a time: a stage holds [split][row][16 k] for both
// operands, a lane's operand of one instruction is one 16-byte read, rows are 48 bytes apart (an odd number of
// 16-byte slots: the reads of a 16-lane group fall into 16 different slots).  There are two stages: while the
// matrix cores work on one, the same wave splits the next 16 k into the other, so its vector instructions run
// in the shadow of its matrix instructions; one barrier per stage.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define XBK 32                          // k per fetch
#define XROW 12                         // 32-bit words per LDS row
#define XSTAGE (3 * 128 * XROW)         // words per operand and stage
#define X3_LDS_BYTES (2 * 2 * XSTAGE * 4)

// bytes 2,3 of a below bytes 2,3 of b
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));        // at most 8 significant bits are left
}

// One thread's share of a 128-row x 32-k operand slice: 8 floats, v[half] the four of one stage.  512 threads.
// KC (k contiguous): thread (row = tid >> 2, kq = tid & 3); v[half][i] = row, k 16*half + 4*kq + i.
// otherwise (rows contiguous): thread (rq = tid >> 4, k16 = tid & 15); v[half][i] = k 16*half + k16, row 4*rq + i.
// FAST: the slice lies inside the matrix and its rows are 16-byte aligned: no test around any load.
// SEG: the operand's contiguous axis (k when KC, the rows otherwise) is mapped through segments (GemmArgs::b_seg_*);
// the 16-byte loads are then only 4-byte aligned, which global memory allows.
struct __attribute__((aligned(4))) F4U { float v[4]; };
template <bool KC, bool FAST, bool SEG = false>
__device__ __forceinline__ void fetch_x3(const float* __restrict__ P, long ld, int r0, int rmax, int k0, int kmax, bool vec,
                                         int tid, float (&v)[2][4], int break_k = 0, long break_off = 0, int seg_len = 0,
                                         int seg_stride = 0, unsigned seg_magic = 0) {
#pragma unroll
  for (int half = 0; half < 2; half++) {
    int r, k;
    const float* p;
    bool ok;
    if (KC) {
      r = r0 + (tid >> 2);
      k = k0 + 16 * half + 4 * (tid & 3);
      if (SEG) {
        const unsigned sg = __umulhi((unsigned)k, seg_magic);
        p = P + (long)r * ld + (long)sg * seg_stride + ((unsigned)k - sg * (unsigned)seg_len);
      } else {
        p = P + (long)r * ld + k;
      }
      ok = r < rmax && k + 3 < kmax;
    } else {
      r = r0 + 4 * (tid >> 4);
      k = k0 + 16 * half + (tid & 15);
      long rr = r;
      if (SEG) {                      // the rows of this operand are the mapped axis (four consecutive ones stay together)
        const unsigned sg = __umulhi((unsigned)r, seg_magic);
        rr = (long)sg * seg_stride + ((unsigned)r - sg * (unsigned)seg_len);
      }
      p = P + (long)k * ld + rr + ((break_k > 0 && k >= break_k) ? break_off : 0);
      ok = k < kmax && r + 3 < rmax;
    }
    if (SEG && (FAST || (vec && ok))) {
      const F4U t = *reinterpret_cast<const F4U*>(p);
      v[half][0] = t.v[0]; v[half][1] = t.v[1]; v[half][2] = t.v[2]; v[half][3] = t.v[3];
    } else if (FAST || (vec && ok)) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      v[half][0] = t.x; v[half][1] = t.y; v[half][2] = t.z; v[half][3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) v[half][i] = (KC ? (r < rmax && k + i < kmax) : (k < kmax && r + i < rmax)) ? p[i] : 0.f;
    }
  }
}

// value of the neighbouring lane (lane ^ 1)
__device__ __forceinline__ float from_pair_lane(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}

// the thread's four floats of one half, split, into a stage S = [3][128][XROW] words
template <bool KC>
__device__ __forceinline__ void place_x3(unsigned* __restrict__ S, int tid, const float (&x)[4]) {
  if (KC) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; i++) split3(x[i], h[i], m[i], l[i]);
    unsigned* d = S + (tid >> 2) * XROW + (tid & 3) * 2;
    *reinterpret_cast<uint2*>(d) = make_uint2(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]));
    *reinterpret_cast<uint2*>(d + 128 * XROW) = make_uint2(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]));
    *reinterpret_cast<uint2*>(d + 2 * 128 * XROW) = make_uint2(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]));
  } else {
    // Two neighbouring lanes hold the same four rows at an even and an odd k.  They swap halves: the even lane
    // ends up with both k of rows 0 and 1, the odd lane with both k of rows 2 and 3, and each writes whole words.
    const bool even = (tid & 1) == 0;
    const float g0 = from_pair_lane(even ? x[2] : x[0]), g1 = from_pair_lane(even ? x[3] : x[1]);
    const float ke[2] = {even ? x[0] : g0, even ? x[1] : g1};      // the even k of this lane's two rows
    const float ko[2] = {even ? g0 : x[2], even ? g1 : x[3]};      // the odd k
    unsigned* d = S + (4 * (tid >> 4) + (even ? 0 : 2)) * XROW + ((tid & 15) >> 1);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3(ke[i], h0, m0, l0);
      split3(ko[i], h1, m1, l1);
      d[i * XROW] = pack_hi16(h0, h1);
      d[i * XROW + 128 * XROW] = pack_hi16(m0, m1);
      d[i * XROW + 2 * 128 * XROW] = pack_hi16(l0, l1);
    }
  }
}

// end of a stage: the workgroup's barrier, and nothing of the next stage is moved in front of it (its split
// would otherwise be hoisted to where its loads have not arrived yet)
__device__ __forceinline__ void stage_end() {
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
}

// Order of the instructions of one stage, given to the compiler's scheduler: the operand reads first, then per
// matrix instruction a few of the vector instructions of the split, its LDS writes and (WITH_FETCH) the loads of
// a later slice in between.
template <bool WITH_FETCH>
__device__ __forceinline__ void stage_schedule() {
  __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);        // LDS reads
#pragma unroll
  for (int i = 0; i < 12; i++) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one matrix instruction
    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // vector instructions
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // an LDS write
    if (WITH_FETCH && i >= 4 && i < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // a load
  }
}

template <bool A_KC, bool B_KC, int EPI, int SEG = 0>
__global__ __launch_bounds__(512, 4) void gemm_x3_kernel(GemmArgs g) {
  constexpr bool B_SEG = SEG == 1 || SEG == 2;
  static_assert(SEG == 0 || (SEG == 1 && B_KC) || (SEG == 2 && !B_KC) || (SEG == 3 && EPI == EPI_PLAIN), "segment mapping");
  extern __shared__ __attribute__((aligned(16))) unsigned x3_lds[];   // [stage][A, B][split][row][XROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;                 // 8 waves as 2 x 4, each 2 x 1 tiles of 32 x 32
  // Workgroups go round-robin to the 8 XCDs, each with an L2 of its own.  The launch index is turned into a
  // logical one that is consecutive within an XCD, and logical neighbours are the workgroups that read the same
  // operand tiles: with x_slowest the column tile changes last (few row tiles share a wide B), otherwise the
  // z index (the K slice) does.
  unsigned li = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
  if ((nwg & 7) == 0) li = (li & 7) * (nwg >> 3) + (li >> 3);
  unsigned bx, by, bz;
  if (g.x_slowest) {
    by = li % gridDim.y; li /= gridDim.y;
    bz = li % gridDim.z; bx = li / gridDim.z;
  } else {
    bx = li % gridDim.x; li /= gridDim.x;
    by = li % gridDim.y; bz = li / gridDim.y;
  }
  const int m0 = by * GBM, n0 = bx * GBN;
  const int nks = g.ksplit ? (int)gridDim.z / (g.groups > 1 ? g.groups : 1) : 1;
  const int grp = g.ksplit ? (int)bz / nks : (int)bz;
  const int slice = g.ksplit ? (int)bz - grp * nks : 0;
  const float* A = g.A + (long)grp * g.a_zs;
  const float* B = g.B + (long)grp * g.b_zs;
  float* C = g.C + (g.ksplit ? (long)slice * g.c_zs + (long)grp * g.c_gs : (long)grp * g.c_zs);
  const int k_begin = slice * g.ksplit;
  const int k_end = g.ksplit ? min(g.K, k_begin + g.ksplit) : g.K;
  const bool a_vec = (g.lda & 3) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0;
  const bool b_vec = B_SEG || ((g.ldb & 3) == 0 && (reinterpret_cast<size_t>(B) & 15) == 0 && (g.b_break_off & 3) == 0);
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

  const int r32 = lane & 31, h4 = (lane >> 5) * 4;
  // the 16 k of one stage: 9 operand reads, 12 matrix instructions.  The reads stand in front of the split of
  // the next stage in program order (the compiler cannot tell that the two stages do not overlap), so that
  // the matrix instructions depend on nothing behind them and can be spread over the split.
  bf16x8 fa[2][3], fb[3];
  auto operands = [&](int stage) {
    const unsigned* As = x3_lds + stage * 2 * XSTAGE;
    const unsigned* Bs = As + XSTAGE;
#pragma unroll
    for (int s = 0; s < 3; s++) {
      fb[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + (s * 128 + wn * 32 + r32) * XROW + h4));
#pragma unroll
      for (int i = 0; i < 2; i++)
        fa[i][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + (s * 128 + wm * 64 + i * 32 + r32) * XROW + h4));
    }
  };
  auto products = [&]() {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      f32x16 c = acc[i];
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[0], c, 0, 0, 0);
      acc[i] = c;
    }
  };
  auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int nf = (k_end - k_begin + XBK - 1) / XBK;          // fetches of 32 k; stages 2f and 2f+1 hold fetch f
    float a0[2][4], b0[2][4], a1[2][4], b1[2][4];
    auto fetch = [&](int f, float (&ra)[2][4], float (&rb)[2][4]) {
      // past the end the last slice is fetched again (FAST has no tests) and never placed
      const int k0 = k_begin + (FAST ? min(f, nf - 1) : f) * XBK;
      fetch_x3<A_KC, FAST>(A, g.lda, m0, g.M, k0, k_end, a_vec, tid, ra);
      fetch_x3<B_KC, FAST, B_SEG>(B, g.ldb, n0, g.N, k0, k_end, b_vec, tid, rb, g.b_break_k, g.b_break_off, g.b_seg_len,
                                  g.b_seg_stride, g.b_seg_magic);
    };
    auto place = [&](int stage, int half, const float (&ra)[2][4], const float (&rb)[2][4]) {
      unsigned* As = x3_lds + stage * 2 * XSTAGE;
      place_x3<A_KC>(As, tid, ra[half]);
      place_x3<B_KC>(As + XSTAGE, tid, rb[half]);
    };
    if (nf <= 0) return;
    fetch(0, a0, b0);
    fetch(1, a1, b1);
    place(0, 0, a0, b0);
    __syncthreads();
    // Four stages per trip.  At its top a0/b0 hold fetch f, a1/b1 fetch f+1, and stage 0 holds the first half of
    // fetch f.  A fetch past the end is never multiplied: the trip that would place it is not run.
    int f = 0;
    for (; f + 1 < nf; f += 2) {
      operands(0);
      place(1, 1, a0, b0);
      fetch(f + 2, a0, b0);
      products();
      stage_schedule<true>();
      stage_end();
      operands(1);
      place(0, 0, a1, b1);
      products();
      stage_schedule<false>();
      stage_end();
      operands(0);
      place(1, 1, a1, b1);
      fetch(f + 3, a1, b1);
      products();
      stage_schedule<true>();
      stage_end();
      operands(1);
      place(0, 0, a0, b0);
      products();
      stage_schedule<false>();
      stage_end();
    }
    if (f < nf) {      // an odd number of fetches: the last one is in a0/b0, its first half already in stage 0
      operands(0);
      place(1, 1, a0, b0);
      products();
      __syncthreads();
      operands(1);
      products();
    }
  };
  const bool full = a_vec && b_vec && m0 + GBM <= g.M && n0 + GBN <= g.N && ((k_end - k_begin) & (XBK - 1)) == 0;
  if (full) run(std::true_type{});
  else run(std::false_type{});
  {
    const int n = n0 + wn * 32 + (lane & 31);
    if (n < g.N) {
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m >= g.M) continue;
          float v = acc[i][r] + bv;
          long idx;
          if (EPI == EPI_CONV) {
            const int b = m / g.H3, h = m - b * g.H3;
            idx = ((long)b * g.N + n) * g.H3 + h;
          } else if (SEG == 3) {
            const unsigned sg = __umulhi((unsigned)n, g.b_seg_magic);
            idx = (long)m * g.ldc + (long)sg * g.b_seg_stride + ((unsigned)n - sg * (unsigned)g.b_seg_len);
          } else {
            idx = (long)m * g.ldc + n;
          }
          if (g.accumulate) v += C[idx];
          if (g.relu) v = v > 0.f ? v : 0.f;
          if (EPI == EPI_MASK) v = g.mask[idx] > 0.f ? v : 0.f;
          C[idx] = v;
        }
    }
  }
}

template <bool A_KC, bool B_KC, int EPI, int SEG = 0>
static int launch_gemm(const GemmArgs& g, int nz, hipStream_t stream) {
  dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, nz);
  // the kernel's LDS limit is raised once per device (the attribute belongs to the device's copy of the kernel)
  FLM_RAISE_LDS_ONCE((gemm_x3_kernel<A_KC, B_KC, EPI, SEG>));
  hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, EPI, SEG>), grid, dim3(512), X3_LDS_BYTES, stream, g);
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

// out[n] = sum over z of part[z*n_elem + n]  (deterministic reduction: 16 waves take every 16th slice each, their
// sums meet in LDS in wave order)
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ part, int nz, long n_elem,
                                                               float* __restrict__ out) {
  __shared__ float s[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (i < n_elem)
    for (int z = w; z < nz; z += 16) acc += part[(long)z * n_elem + i];
  s[w][lane] = acc;
  __syncthreads();
  if (w == 0 && i < n_elem) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += s[k][lane];
    out[i] = t;
  }
}

// out[m*ldo + n] = relu(bias[n] + sum over z of part[z][m][n])  (split-K reduction with the layer's epilogue)
// The bias of column n is bias0[n] for n < n_half and bias1[n - n_half] above (two layers side by side).
// out[i] = mask[i] > 0 ? sum over z of part[z*n + i] : 0, four elements per thread, slices added in order
// (n a multiple of 4, 16-byte aligned buffers)
__global__ __launch_bounds__(256) void reduce_mask_kernel(const float* __restrict__ part, int nz, long n,
                                                          const float* __restrict__ mask, float* __restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 s = *reinterpret_cast<const float4*>(part + i);
  for (int z = 1; z < nz; z++) {
    const float4 v = *reinterpret_cast<const float4*>(part + (long)z * n + i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float4 m = *reinterpret_cast<const float4*>(mask + i);
  s.x = m.x > 0.f ? s.x : 0.f; s.y = m.y > 0.f ? s.y : 0.f; s.z = m.z > 0.f ? s.z : 0.f; s.w = m.w > 0.f ? s.w : 0.f;
  *reinterpret_cast<float4*>(out + i) = s;
}

__global__ void reduce_bias_relu_kernel(const float* __restrict__ part, int nz, int M, int N, const float* __restrict__ bias0,
                                        const float* __restrict__ bias1, int n_half, float* __restrict__ out, long ldo) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (long)m * N);
  float s = 0.f;
  for (int z = 0; z < nz; z++) s += part[(long)z * M * N + i];
  s += n < n_half ? bias0[n] : bias1[n - n_half];
  out[(long)m * ldo + n] = s > 0.f ? s : 0.f;
}

// column sums: out[n] = sum_m X[m*ld + n], one workgroup per 64 columns, fixed summation order
// (columns from n_split on go to out1: two destinations that are not adjacent, one launch)
// copy (optional): all N sums once more, contiguous
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ X, int M, int N, long ld, float* __restrict__ out,
                                                      int n_split, float* __restrict__ out1, float* __restrict__ copy) {
  __shared__ float s[16][64];
  const int lane = threadIdx.x & 63, n = blockIdx.x * 64 + lane, w = threadIdx.x >> 6;
  // 16 waves split the rows; four independent partial sums per lane keep four loads in flight; every
  // partial sum and their combination have a fixed order
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (n < N) {
    int m = w;
    for (; m + 48 < M; m += 64) {
      a0 += X[(long)m * ld + n];
      a1 += X[(long)(m + 16) * ld + n];
      a2 += X[(long)(m + 32) * ld + n];
      a3 += X[(long)(m + 48) * ld + n];
    }
    for (; m < M; m += 16) a0 += X[(long)m * ld + n];
  }
  s[w][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (w == 0 && n < N) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += s[i][lane];
    if (n < n_split) out[n] = acc;
    else out1[n - n_split] = acc;
    if (copy) copy[n] = acc;
  }
}

// conv bias gradient: out[c] = sum over b,h of dX1[b][c][h]
// (blockIdx.y selects one of gridDim.y interleaved parts of the batch; out[y*64 + c] are partial sums that the
// caller adds up in a fixed order)
__global__ __launch_bounds__(1024) void conv_bias_grad_kernel(const float* __restrict__ dX1, int B, int H3, float* __restrict__ out) {
  __shared__ float s[16];
  const int c = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc = 0.f;
  for (int b = blockIdx.y * 16 + w; b < B; b += 16 * gridDim.y) {
    const float* row = dX1 + ((long)b * 64 + c) * H3;
    for (int h = lane; h < H3; h += 64) acc += row[h];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) s[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += s[k];
    out[blockIdx.y * 64 + c] = t;
  }
}

// ---- small-batch convolution (act: B <= 8, float input) -----------------------------------------------------
// x1[b][c][h] = relu(bias[c] + sum_k x[b][h][k] * Wc[c][k]) for four rows (b,h) per workgroup: the filter bank is
// staged in LDS transposed ([k][c], padded so that neither the staging nor the reads conflict), thread (c, r)
// owns one output, and the four rows of one filter leave as one 16-byte run.
__global__ __launch_bounds__(256) void conv_rows_kernel(const float* __restrict__ X, const float* __restrict__ Wc,
                                                        const float* __restrict__ bias, long M, int T1, int H3,
                                                        float* __restrict__ x1) {
  extern __shared__ __attribute__((aligned(16))) float conv_lds[];
  float* wt = conv_lds;                    // [T1][65]
  float* xs = conv_lds + (size_t)T1 * 65;  // [4][T1]
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * 4;
  for (int i = tid; i < 64 * T1; i += 256) {
    const int c = i / T1, k = i - c * T1;
    wt[k * 65 + c] = Wc[i];
  }
  for (int i = tid; i < 4 * T1; i += 256) {
    const long row = row0 + i / T1;
    xs[i] = row < M ? X[row0 * T1 + i] : 0.f;
  }
  __syncthreads();
  const int c = tid >> 2, r = tid & 3;
  const float* xr = xs + r * T1;
  float acc = 0.f;
  for (int k = 0; k < T1; k++) acc = fmaf(xr[k], wt[k * 65 + c], acc);
  const long row = row0 + r;
  if (row < M) {
    const long b = row / H3, h = row - b * H3;
    const float v = acc + bias[c];
    x1[(b * 64 + c) * H3 + h] = v > 0.f ? v : 0.f;
  }
}

// ---- small-batch first dense layer (act: B <= 8): one wave per output neuron streams its weight row
template <int BMAX>
__global__ __launch_bounds__(256) void fc1_small_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, int B, long K, float* __restrict__ out,
                                                        int ldo, int n_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const float4* w4 = reinterpret_cast<const float4*>(W + (long)row * K);
  float acc[BMAX];
#pragma unroll
  for (int b = 0; b < BMAX; b++) acc[b] = 0.f;
  const long K4 = K >> 2;
  for (long i = lane; i < K4; i += 64) {
    const float4 w = w4[i];
#pragma unroll
    for (int b = 0; b < BMAX; b++)
      if (b < B) {
        const float4 x = reinterpret_cast<const float4*>(X + (long)b * K)[i];
        acc[b] = fmaf(w.x, x.x, acc[b]);
        acc[b] = fmaf(w.y, x.y, acc[b]);
        acc[b] = fmaf(w.z, x.z, acc[b]);
        acc[b] = fmaf(w.w, x.w, acc[b]);
      }
  }
#pragma unroll
  for (int b = 0; b < BMAX; b++) {
    float v = acc[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0 && b < B) {
      v += bias[row];
      out[(long)b * ldo + row] = v > 0.f ? v : 0.f;
    }
  }
}

// ---- small-batch first dense layer, both streams in one launch ---------------------------------------------
// The layer is bound by streaming its weights (2 x 512 x K floats: 78.6 MB at max_rails 100) once.  A workgroup
// of 16 waves owns four output neurons; its 1024 threads take the 16-byte pieces of the four weight rows
// round-robin (every load of a wave is 1 KiB of one row), so a lane has its x pieces in registers and about
// twenty independent weight loads in flight; 256 workgroups cover the 1024 neurons of the two streams, one per
// CU.  Lanes reduce inside the wave, the sixteen waves meet in LDS and are summed in a fixed order.
template <int BMAX>
__global__ __launch_bounds__(1024) void fc1_rows_kernel(const float* __restrict__ X, const float* __restrict__ Wadv,
                                                        const float* __restrict__ Wval, const float* __restrict__ badv,
                                                        const float* __restrict__ bval, int B, long K, float* __restrict__ out) {
  constexpr int R = 4;
  __shared__ float part[16][R][BMAX];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = blockIdx.x * R;                       // 0..1023: advantage stream first, then the value stream
  const float* W = row0 < 512 ? Wadv + (long)row0 * K : Wval + (long)(row0 - 512) * K;
  const long K4 = K >> 2;
  float acc[R][BMAX];
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int b = 0; b < BMAX; b++) acc[r][b] = 0.f;
  typedef float v4f __attribute__((ext_vector_type(4)));
  // U pieces of every row per trip, all loads of a trip issued before the first product (no load is conditional)
  auto trip = [&](long i0, auto u_count) {
    constexpr int U = decltype(u_count)::value;
    v4f w[U][R];
    float4 x[U][BMAX];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = i0 + (long)u * 1024;
#pragma unroll
      for (int r = 0; r < R; r++)
        w[u][r] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(W + (long)r * K) + i);   // read once
#pragma unroll
      for (int b = 0; b < BMAX; b++)
        x[u][b] = reinterpret_cast<const float4*>(X + (long)(b < B ? b : 0) * K)[i];
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the trip is in flight before the first product waits
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int b = 0; b < BMAX; b++) {
          acc[r][b] = fmaf(w[u][r].x, x[u][b].x, acc[r][b]);
          acc[r][b] = fmaf(w[u][r].y, x[u][b].y, acc[r][b]);
          acc[r][b] = fmaf(w[u][r].z, x[u][b].z, acc[r][b]);
          acc[r][b] = fmaf(w[u][r].w, x[u][b].w, acc[r][b]);
        }
  };
  constexpr int UT = BMAX == 1 ? 4 : (BMAX <= 4 ? 2 : 1);   // pieces per trip: what the registers hold beside x
  long i0 = tid;
  for (; i0 + (UT - 1) * 1024 < K4; i0 += UT * 1024) trip(i0, std::integral_constant<int, UT>{});
  for (; i0 < K4; i0 += 1024) trip(i0, std::integral_constant<int, 1>{});
#pragma unroll
  for (int r = 0; r < R; r++)
#pragma unroll
    for (int b = 0; b < BMAX; b++) {
      float v = acc[r][b];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
      if (lane == 0) part[wave][r][b] = v;
    }
  __syncthreads();
  if (tid < R * BMAX) {
    const int r = tid / BMAX, b = tid - r * BMAX;
    if (b < B) {
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 16; w2++) v += part[w2][r][b];
      const int row = row0 + r;
      v += row < 512 ? badv[row] : bval[row - 512];
      out[(long)b * 1024 + row] = v > 0.f ? v : 0.f;
    }
  }
}

// ---- output layers + dueling combination, one wave per sample ------------------------------------------
// hid[B,1024]: columns 0..511 advantage stream, 512..1023 value stream.  q[B,NA], NA <= FLM_DQN_MAX_ACTIONS.
__global__ __launch_bounds__(64) void head_forward_kernel(const float* __restrict__ hid, const float* __restrict__ a2w,
                                                          const float* __restrict__ a2b, const float* __restrict__ v2w,
                                                          const float* __restrict__ v2b, int B, int NA, float* __restrict__ q) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const float* h = hid + (long)b * 1024;
  float sa[FLM_DQN_MAX_ACTIONS], sv = 0.f;
#pragma unroll
  for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++) sa[a] = 0.f;
  for (int i = lane; i < 512; i += 64) {
    const float ha = h[i], hv = h[512 + i];
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) sa[a] = fmaf(ha, a2w[a * 512 + i], sa[a]);
    sv = fmaf(hv, v2w[i], sv);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) sa[a] += __shfl_down(sa[a], o);
    sv += __shfl_down(sv, o);
  }
  if (lane == 0) {
    const float val = sv + v2b[0];
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) {
        sa[a] += a2b[a];
        sum += sa[a];
      }
    const float mean = NA == 2 ? sum * 0.5f : sum / (float)NA;
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) q[b * NA + a] = val + sa[a] - mean;
  }
}

// Rails that no environment has: rows [R4, Rp) of all three planes of every observation are zero, so the convolution
// gives relu(conv bias) there for every sample, and their share of the first dense layer is the same vector for every
// sample: bias2[o] = bias[o] + sum over filters c of relu(conv_b[c]) * wpad[o][c], where wpad[o][c] is the sum of
// W[o][c*H3 + plane*Rp + rail] over the planes and the rails >= R4.  One workgroup per output unit o (advantage stream
// first); all sums in a fixed order.  wpad_out (optional): wpad as [64][1024] (filter-major), which the backward pass needs as well.
// (workgroups 1024.. do the same for a second network, j1: a learn step needs both networks' vectors)
struct PaddingBiasJob {
  const float *Wa, *Wv, *bias_a, *bias_v, *conv_b;
  float *bias2, *wpad_out, *rcb_out;
};
__global__ __launch_bounds__(256) void fc1_padding_bias_kernel(PaddingBiasJob j0, PaddingBiasJob j1, long K1, int H3, int Rp,
                                                               int R4) {
  __shared__ float lane_sum[64][65];
  __shared__ float red[64];
  const PaddingBiasJob& j = blockIdx.x >= 1024 ? j1 : j0;
  const float* __restrict__ Wa = j.Wa;
  const float* __restrict__ Wv = j.Wv;
  const float* __restrict__ bias_a = j.bias_a;
  const float* __restrict__ bias_v = j.bias_v;
  const float* __restrict__ conv_b = j.conv_b;
  float* __restrict__ bias2 = j.bias2;
  float* __restrict__ wpad_out = j.wpad_out;
  float* __restrict__ rcb_out = j.rcb_out;
  const int o = blockIdx.x & 1023, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float* W = o < 512 ? Wa + (long)o * K1 : Wv + (long)(o - 512) * K1;
  const int L = Rp - R4;
  // a wave takes every fourth filter, its lanes the consecutive weights of a segment: 48 independent loads per lane
  // and pass (16 filters x 3 planes); the lanes' sums of a filter meet in LDS and are added in lane order
  float a[16];
#pragma unroll
  for (int j = 0; j < 16; j++) a[j] = 0.f;
  for (int i = lane; i < L; i += 64) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const float* p = W + (long)(w + 4 * j) * H3 + R4 + i;
      a[j] += (p[0] + p[Rp]) + p[2 * Rp];
    }
  }
#pragma unroll
  for (int j = 0; j < 16; j++) lane_sum[w + 4 * j][lane] = a[j];
  __syncthreads();
  if (t < 64) {
    float wp = 0.f;
    for (int l = 0; l < 64; l++) wp += lane_sum[t][l];
    if (wpad_out) wpad_out[(long)t * 1024 + o] = wp;             // [filter][output unit]
    if (rcb_out && o == 0) rcb_out[t] = fmaxf(conv_b[t], 0.f);   // what every filter gives on an empty row
    red[t] = fmaxf(conv_b[t], 0.f) * wp;
  }
  __syncthreads();
  if (t == 0) {
    float acc = 0.f;
    for (int c = 0; c < 64; c++) acc += red[c];
    bias2[o] = (o < 512 ? bias_a[o] : bias_v[o - 512]) + acc;
  }
}

// The split-K reduction of the first dense layer (both streams) with its bias and ReLU, then the output layers and
// the dueling combination: reduce_bias_relu_kernel and head_forward_kernel in one launch, one workgroup per sample.
// Thread t owns hidden units 4t..4t+3 (threads 0..127 the advantage stream, 128..255 the value stream): it sums
// their K slices in slice order (the same bits as reduce_bias_relu_kernel), stores them, and multiplies them with
// its four columns of the output weights; lanes meet in the wave, the four waves in LDS, in a fixed order.
// part, hid and the two bias vectors must be 16-byte aligned.
__global__ __launch_bounds__(256) void fc1_reduce_head_kernel(const float* __restrict__ part, int nz, int B,
                                                              const float* __restrict__ bias_a, const float* __restrict__ bias_v,
                                                              float* __restrict__ hid, const float* __restrict__ a2w,
                                                              const float* __restrict__ a2b, const float* __restrict__ v2w,
                                                              const float* __restrict__ v2b, int NA, float* __restrict__ q) {
  __shared__ float red[4][FLM_DQN_MAX_ACTIONS];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const long slice = (long)B * 256;                                      // float4 per K slice
  const float4* p = reinterpret_cast<const float4*>(part) + (long)b * 256 + t;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int z = 0;
  for (; z + 4 <= nz; z += 4) {                                          // four loads in flight, added in slice order
    const float4 v0 = p[(long)z * slice], v1 = p[(long)(z + 1) * slice], v2 = p[(long)(z + 2) * slice],
                 v3 = p[(long)(z + 3) * slice];
    s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
    s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
    s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
    s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
  }
  for (; z < nz; z++) {
    const float4 v0 = p[(long)z * slice];
    s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
  }
  const bool adv = t < 128;
  const int u = adv ? 4 * t : 4 * (t - 128);                             // first of the thread's units inside its stream
  const float4 bb = *reinterpret_cast<const float4*>((adv ? bias_a : bias_v) + u);
  s.x = fmaxf(s.x + bb.x, 0.f); s.y = fmaxf(s.y + bb.y, 0.f); s.z = fmaxf(s.z + bb.z, 0.f); s.w = fmaxf(s.w + bb.w, 0.f);
  reinterpret_cast<float4*>(hid)[(long)b * 256 + t] = s;
  float acc[FLM_DQN_MAX_ACTIONS];
#pragma unroll
  for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++) acc[a] = 0.f;
  if (adv) {
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) {
        const float* wr = a2w + a * 512 + u;
        acc[a] = fmaf(s.w, wr[3], fmaf(s.z, wr[2], fmaf(s.y, wr[1], s.x * wr[0])));
      }
  } else {
    const float* wr = v2w + u;
    acc[0] = fmaf(s.w, wr[3], fmaf(s.z, wr[2], fmaf(s.y, wr[1], s.x * wr[0])));
  }
#pragma unroll
  for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
    if (a < NA || a == 0) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[a] += __shfl_xor(acc[a], o);
      if (lane == 0) red[w][a] = acc[a];
    }
  __syncthreads();
  if (t == 0) {
    const float val = (red[2][0] + red[3][0]) + v2b[0];
    float av[FLM_DQN_MAX_ACTIONS], sum = 0.f;
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) {
        av[a] = (red[0][a] + red[1][a]) + a2b[a];
        sum += av[a];
      }
    const float mean = NA == 2 ? sum * 0.5f : sum / (float)NA;
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) q[b * NA + a] = val + av[a] - mean;
  }
}

// Double-DQN target and loss gradient (src/agent.py:97-111):
//   a* = argmax_a q_next_local[b,a];  target = r + gamma * q_next_target[b,a*] * (1 - done)
//   loss = mean_b (q[b,action[b]] - target)^2;  dq[b,a] = 2*(q - target)/B on the taken action, else 0
// loss_out != nullptr: the launch is one workgroup that covers the batch (B <= 1024) and also writes the mean of the
// squared errors, summed in the order of mean_kernel.
__global__ __launch_bounds__(1024) void td_loss_kernel(const float* __restrict__ q, const float* __restrict__ q_next_local,
                                                       const float* __restrict__ q_next_target,
                                                       const int64_t* __restrict__ action, const float* __restrict__ reward,
                                                       const float* __restrict__ done, float gamma, int B, int NA,
                                                       float* __restrict__ dq, float* __restrict__ sq_err,
                                                       float* __restrict__ loss_out) {
  __shared__ float s_sq[1024];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    int astar = 0;                                   // the first maximum
    for (int a = 1; a < NA; a++)
      if (q_next_local[b * NA + a] > q_next_local[b * NA + astar]) astar = a;
    const float target = reward[b] + (gamma * q_next_target[b * NA + astar]) * (1.f - done[b]);
    const int a = (int)min(max(action[b], (int64_t)0), (int64_t)(NA - 1));   // a bad index must not leave the row
    const float diff = q[b * NA + a] - target;
    sq_err[b] = diff * diff;
    if (loss_out) s_sq[threadIdx.x] = diff * diff;
    for (int k = 0; k < NA; k++) dq[b * NA + k] = k == a ? 2.f * diff / (float)B : 0.f;
  }
  if (!loss_out) return;
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 64) s += s_sq[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) *loss_out = s / (float)B;
  }
}

// The reference's target without double DQN (src/agent.py:103-105).  Its Q_targets_next keeps a trailing axis
// of length one, so rewards + gamma * Q_targets_next * (1 - dones) is a [B,B] matrix, target[i,j] = r[j] +
// gamma * max_a q_next_target[i,a] * (1 - done[j]), and mse_loss compares q[j,action[j]] with every row i of
// column j: loss = mean over (i,j) of (q_j - target[i,j])^2.  In closed form, with m1 / m2 the batch means of
// the maxima and of their squares, u_j = q_j - r_j and c_j = gamma * (1 - done[j]):
//   loss = mean_j (u_j^2 - 2 u_j c_j m1 + c_j^2 m2),   dloss/dq_j = 2 (u_j - c_j m1) / B.
// One workgroup: the two means in a fixed order, then one thread per sample.
__global__ __launch_bounds__(1024) void td_loss_plain_kernel(const float* __restrict__ q, const float* __restrict__ q_next_target,
                                                             const int64_t* __restrict__ action,
                                                             const float* __restrict__ reward, const float* __restrict__ done,
                                                             float gamma, int B, int NA, float* __restrict__ dq,
                                                             float* __restrict__ sq_err) {
  __shared__ double s1[1024], s2[1024];
  const int tid = threadIdx.x;
  double a1 = 0.0, a2 = 0.0;
  for (int b = tid; b < B; b += 1024) {
    float m = q_next_target[b * NA];
    for (int k = 1; k < NA; k++) m = fmaxf(m, q_next_target[b * NA + k]);
    a1 += (double)m;
    a2 += (double)m * (double)m;
  }
  s1[tid] = a1;
  s2[tid] = a2;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) {
      s1[tid] += s1[tid + o];
      s2[tid] += s2[tid + o];
    }
    __syncthreads();
  }
  const float m1 = (float)(s1[0] / (double)B), m2 = (float)(s2[0] / (double)B);
  for (int b = tid; b < B; b += 1024) {
    const int a = (int)min(max(action[b], (int64_t)0), (int64_t)(NA - 1));
    const float u = q[b * NA + a] - reward[b];
    const float c = gamma * (1.f - done[b]);
    sq_err[b] = u * u - 2.f * u * c * m1 + c * c * m2;
    for (int k = 0; k < NA; k++) dq[b * NA + k] = k == a ? 2.f * (u - c * m1) / (float)B : 0.f;
  }
}

__global__ __launch_bounds__(64) void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) *out = s / (float)n;
}

// Backward of the head: from dq[B,NA] to the hidden-layer gradient dhid[B,1024] (ReLU applied) and the
// gradients of the four small tensors.  One wave per sample writes dhid; the small tensors are reduced
// by a second kernel over the batch in a fixed order.
__global__ __launch_bounds__(64) void head_backward_hidden_kernel(const float* __restrict__ hid, const float* __restrict__ dq,
                                                                  const float* __restrict__ a2w, const float* __restrict__ v2w,
                                                                  int B, int NA, float* __restrict__ dhid,
                                                                  float* __restrict__ dadv, float* __restrict__ dval) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  // q = val + adv - mean(adv):  dval = sum_a dq[a],  dadv[a] = dq[a] - dval / NA
  float da[FLM_DQN_MAX_ACTIONS], dv = 0.f;
#pragma unroll
  for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++) {
    da[a] = a < NA ? dq[b * NA + a] : 0.f;
    dv += da[a];
  }
  const float mean = NA == 2 ? 0.5f * dv : dv / (float)NA;
#pragma unroll
  for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++) da[a] -= mean;
  const float* h = hid + (long)b * 1024;
  float* dh = dhid + (long)b * 1024;
  for (int i = lane; i < 512; i += 64) {
    float ga = 0.f;
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) ga = a == 0 ? da[0] * a2w[i] : ga + da[a] * a2w[a * 512 + i];
    dh[i] = h[i] > 0.f ? ga : 0.f;
    const float gv = dv * v2w[i];
    dh[512 + i] = h[512 + i] > 0.f ? gv : 0.f;
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++)
      if (a < NA) dadv[b * NA + a] = da[a];
    dval[b] = dv;
  }
}

// g_a2w[a,i] = sum_b dadv[b,a]*hid[b,i]; g_a2b[a] = sum_b dadv[b,a]; g_v2w[i] = sum_b dval[b]*hid[b,512+i]; g_v2b
__global__ __launch_bounds__(1024) void head_backward_params_kernel(const float* __restrict__ hid, const float* __restrict__ dadv,
                                                                    const float* __restrict__ dval, int B, int NA,
                                                                    float* __restrict__ g_a2w, float* __restrict__ g_a2b,
                                                                    float* __restrict__ g_v2w, float* __restrict__ g_v2b) {
  // workgroup = (64 hidden units, one output: blockIdx.y < NA an action's advantage row, blockIdx.y == NA the value
  // row); 16 waves take every 16th sample each, their sums meet in LDS in wave order
  __shared__ float s[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane, a = blockIdx.y;
  const bool value = a == NA;
  float acc = 0.f;
  for (int b = w; b < B; b += 16) {
    const float d = value ? dval[b] : dadv[b * NA + a];
    acc = fmaf(d, hid[(long)b * 1024 + (value ? 512 : 0) + i], acc);
  }
  s[w][lane] = acc;
  __syncthreads();
  if (w == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += s[k][lane];
    if (value) g_v2w[i] = t;
    else g_a2w[a * 512 + i] = t;
  }
  if (blockIdx.x == 0 && w == 1) {   // this output's bias gradient: one wave, lanes stride over the batch
    float bsum = 0.f;
    for (int b = lane; b < B; b += 64) bsum += value ? dval[b] : dadv[b * NA + a];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bsum += __shfl_xor(bsum, o);
    if (lane == 0) {
      if (value) g_v2b[0] = bsum;
      else g_a2b[a] = bsum;
    }
  }
}

// td_loss_kernel, head_backward_hidden_kernel, head_backward_params_kernel and the column sums of dhid
// (colsum_kernel) in one launch, for a batch that one workgroup covers (B <= 1024) under double DQN: four dependent
// launches of a few microseconds each are mostly launch latency.  Every workgroup works out the loss gradient of all
// B samples itself (a few loads and a dozen operations per sample) and keeps it in LDS; workgroups 0..15 then own 64
// hidden units each (dhid for every sample, summed over the samples in colsum_kernel's order: the first dense
// layer's bias gradient), workgroups 16.. are head_backward_params_kernel's (64 hidden units, one output).  Every
// sum keeps the order of the kernel it comes from; workgroup 0 also writes dq, the squared errors, their mean, dadv
// and dval.
__global__ __launch_bounds__(1024) void head_backward_fused_kernel(
    const float* __restrict__ q, const float* __restrict__ q_next_local, const float* __restrict__ q_next_target,
    const int64_t* __restrict__ action, const float* __restrict__ reward, const float* __restrict__ done, float gamma, int B,
    int NA, const float* __restrict__ hid, const float* __restrict__ a2w, const float* __restrict__ v2w,
    float* __restrict__ dq, float* __restrict__ sq_err, float* __restrict__ loss_out, float* __restrict__ dhid,
    float* __restrict__ dadv, float* __restrict__ dval, float* __restrict__ g_a2w, float* __restrict__ g_a2b,
    float* __restrict__ g_v2w, float* __restrict__ g_v2b, float* __restrict__ g_a1b, float* __restrict__ g_v1b,
    float* __restrict__ sums_copy) {
  __shared__ float s_da[FLM_DQN_MAX_ACTIONS][1024], s_dv[1024], s_sq[1024], s_red[16][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const bool first = blockIdx.x == 0;
  if (tid < B) {
    const int b = tid;
    int astar = 0;                                   // the first maximum
    for (int a = 1; a < NA; a++)
      if (q_next_local[b * NA + a] > q_next_local[b * NA + astar]) astar = a;
    const float target = reward[b] + (gamma * q_next_target[b * NA + astar]) * (1.f - done[b]);
    const int act = (int)min(max(action[b], (int64_t)0), (int64_t)(NA - 1));
    const float diff = q[b * NA + act] - target;
    s_sq[b] = diff * diff;
    const float d_taken = 2.f * diff / (float)B;
    // q = val + adv - mean(adv):  dval = sum_a dq[a],  dadv[a] = dq[a] - dval / NA
    float da[FLM_DQN_MAX_ACTIONS], dv = 0.f;
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++) {
      da[a] = a < NA && a == act ? d_taken : 0.f;
      dv += da[a];
    }
    const float mean = NA == 2 ? 0.5f * dv : dv / (float)NA;
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++) {
      if (first && a < NA) dq[b * NA + a] = da[a];
      da[a] -= mean;
      s_da[a][b] = da[a];
      if (first && a < NA) dadv[b * NA + a] = da[a];
    }
    s_dv[b] = dv;
    if (first) {
      sq_err[b] = diff * diff;
      dval[b] = dv;
    }
  }
  __syncthreads();
  if (first && tid < 64) {                           // the mean of the squared errors, in mean_kernel's order
    float s = 0.f;
    for (int i = tid; i < B; i += 64) s += s_sq[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (tid == 0) *loss_out = s / (float)B;
  }
  if (blockIdx.x < 16) {
    // 64 hidden units: dhid of every sample, and its sum over the samples
    const int n = blockIdx.x * 64 + lane;
    const bool adv = blockIdx.x < 8;
    const int i = adv ? n : n - 512;
    float wcol[FLM_DQN_MAX_ACTIONS];
#pragma unroll
    for (int a = 0; a < FLM_DQN_MAX_ACTIONS; a++) wcol[a] = adv ? (a < NA ? a2w[a * 512 + i] : 0.f) : (a == 0 ? v2w[i] : 0.f);
    auto row = [&](int m) -> float {
      float g;
      if (adv) {
        g = s_da[0][m] * wcol[0];
#pragma unroll
        for (int a = 1; a < FLM_DQN_MAX_ACTIONS; a++)
          if (a < NA) g = fmaf(s_da[a][m], wcol[a], g);
      } else {
        g = s_dv[m] * wcol[0];
      }
      const float out = hid[(long)m * 1024 + n] > 0.f ? g : 0.f;
      dhid[(long)m * 1024 + n] = out;
      return out;
    };
    // (the activations of eight rows are fetched before the first of them is used: the loop is a chain of loads otherwise)
    auto fetch = [&](int m) -> float { return hid[(long)m * 1024 + n]; };
    auto row_with = [&](int m, float h) -> float {
      float g;
      if (adv) {
        g = s_da[0][m] * wcol[0];
#pragma unroll
        for (int a = 1; a < FLM_DQN_MAX_ACTIONS; a++)
          if (a < NA) g = fmaf(s_da[a][m], wcol[a], g);
      } else {
        g = s_dv[m] * wcol[0];
      }
      const float out = h > 0.f ? g : 0.f;
      dhid[(long)m * 1024 + n] = out;
      return out;
    };
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int m = w;
    for (; m + 112 < B; m += 128) {
      float h[8];
#pragma unroll
      for (int u = 0; u < 8; u++) h[u] = fetch(m + 16 * u);
      a0 += row_with(m, h[0]);
      a1 += row_with(m + 16, h[1]);
      a2 += row_with(m + 32, h[2]);
      a3 += row_with(m + 48, h[3]);
      a0 += row_with(m + 64, h[4]);
      a1 += row_with(m + 80, h[5]);
      a2 += row_with(m + 96, h[6]);
      a3 += row_with(m + 112, h[7]);
    }
    for (; m + 48 < B; m += 64) {
      a0 += row(m);
      a1 += row(m + 16);
      a2 += row(m + 32);
      a3 += row(m + 48);
    }
    for (; m < B; m += 16) a0 += row(m);
    s_red[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 16; k++) acc += s_red[k][lane];
      if (adv) g_a1b[i] = acc;
      else g_v1b[i] = acc;
      if (sums_copy) sums_copy[n] = acc;
    }
  } else {
    const int job = blockIdx.x - 16, bx = job & 7, a = job >> 3;
    const int i = bx * 64 + lane;
    const bool value = a == NA;
    const float* d = value ? s_dv : s_da[a];
    float acc = 0.f;
    const float* hcol = hid + (value ? 512 : 0) + i;
    int b = w;
    for (; b + 112 < B; b += 128) {                  // eight loads in flight, the products added in the order of the samples
      float h[8];
#pragma unroll
      for (int u = 0; u < 8; u++) h[u] = hcol[(long)(b + 16 * u) * 1024];
#pragma unroll
      for (int u = 0; u < 8; u++) acc = fmaf(d[b + 16 * u], h[u], acc);
    }
    for (; b < B; b += 16) acc = fmaf(d[b], hcol[(long)b * 1024], acc);
    s_red[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; k++) t += s_red[k][lane];
      if (value) g_v2w[i] = t;
      else g_a2w[a * 512 + i] = t;
    }
    if (bx == 0 && w == 1) {   // this output's bias gradient: one wave, lanes stride over the batch
      float bsum = 0.f;
      for (int b = lane; b < B; b += 64) bsum += d[b];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bsum += __shfl_xor(bsum, o);
      if (lane == 0) {
        if (value) g_v2b[0] = bsum;
        else g_a2b[a] = bsum;
      }
    }
  }
}

// ---- optimiser and target network --------------------------------------------------------------------
// torch.optim.Adam defaults (no weight decay, no amsgrad), one step over the flat buffers.  The products
// are kept unfused so that every intermediate is rounded where the tensor-by-tensor formulation rounds it.
// One element of the step.  Contraction into fused multiply-adds is switched off here (the _rn intrinsics alone do not
// prevent it), so that the three kernels below give the same bits whatever surrounds the call.
struct AdamConsts { float one_minus_b1, beta2, one_minus_b2, eps, step_size, bc2_sqrt, tau, one_minus_tau; };
__device__ __forceinline__ void adam_element(float& p, float g, float& m, float& v, const AdamConsts& k) {
#pragma clang fp contract(off)
  const float mi = m + k.one_minus_b1 * (g - m);
  const float vi = v * k.beta2 + (k.one_minus_b2 * g) * g;
  m = mi;
  v = vi;
  const float denom = __fdiv_rn(__fsqrt_rn(vi), k.bc2_sqrt) + k.eps;
  p = p + (-k.step_size) * __fdiv_rn(mi, denom);
}
__device__ __forceinline__ float soft_element(float p, float t, const AdamConsts& k) {
#pragma clang fp contract(off)
  return k.tau * p + k.one_minus_tau * t;
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float one_minus_b1, float beta2, float one_minus_b2, float eps, float step_size,
                            float bc2_sqrt) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AdamConsts k = {one_minus_b1, beta2, one_minus_b2, eps, step_size, bc2_sqrt, 0.f, 0.f};
  float pi = p[i], mi = m[i], vi = v[i];
  adam_element(pi, g[i], mi, vi, k);
  m[i] = mi;
  v[i] = vi;
  p[i] = pi;
}

// Adam step followed by the soft update of the target network with the new parameters, in one pass
__global__ void adam_soft_update_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                        float* __restrict__ v, float* __restrict__ target, long n, float one_minus_b1,
                                        float beta2, float one_minus_b2, float eps, float step_size, float bc2_sqrt,
                                        float tau, float one_minus_tau) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AdamConsts k = {one_minus_b1, beta2, one_minus_b2, eps, step_size, bc2_sqrt, tau, one_minus_tau};
  float pi = p[i], mi = m[i], vi = v[i];
  adam_element(pi, g[i], mi, vi, k);
  m[i] = mi;
  v[i] = vi;
  p[i] = pi;
  target[i] = soft_element(pi, target[i], k);
}

// The same pass after a learn step over the used rails that left the first dense layer's weight gradient under the
// empty rows unwritten: there it is rcb[c] * g[bias of o] (fc1_wgrad_padding_kernel's product), taken from the layer's
// bias gradient instead of being read - 39 MB less to write and to read at 20 of 70 rails.  Indices of the two weight
// blocks are below 2^24: the quotients come from float estimates corrected by one.
__device__ __forceinline__ unsigned div_small(unsigned n, unsigned d, float inv_d) {
  unsigned q = (unsigned)((float)n * inv_d);
  if (q * d > n) q--;
  else if ((q + 1) * d <= n) q++;
  return q;
}
// COMPACT: the optimizer's own data of the first dense layer's weights (moments and gradient) is kept without the
// empty rows.  (a) The weights under the empty rows have received the same gradient in every step so far (the caller's
// condition), so their moments are the same along the rows: mv_in [2][512][64] pairs (m, v) holds them once per
// (stream, output unit, filter); the pass reads them from there and writes the new pair to mv_out (another buffer:
// other threads are still reading mv_in).  (b) The moments under the used rails live in mpk / vpk [2][512][Kc]
// (Kc = 64*3*R4 columns: filter, plane, rail < R4), and the gradient is read in the same packed form from the start of
// each weight block of g, as the learn step leaves it when asked to.  m and v of the two weight blocks are not touched:
// reading 20 floats out of every 70 moves nearly every 128-byte line anyway, the packed form moves a third.
// The parameters and the target are updated element by element as before: the same bits.
template <bool COMPACT>
__global__ __launch_bounds__(256) void adam_soft_update_rails_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
    float* __restrict__ target, long n, float one_minus_b1, float beta2, float one_minus_b2, float eps, float step_size,
    float bc2_sqrt, float tau, float one_minus_tau, long a1_w, long a1_b, long v1_w, long v1_b, unsigned K1, unsigned H3,
    unsigned Rp, unsigned R4, float inv_K1, float inv_H3, float inv_Rp, const float* __restrict__ rcb,
    const float2* __restrict__ mv_in, float2* __restrict__ mv_out, float* __restrict__ mpk, float* __restrict__ vpk) {
  // four consecutive parameters per thread (the two weight blocks start and end on multiples of four, so a thread's
  // four lie inside one or outside both): their place (output unit, filter, rail) is worked out once and stepped
  const long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i0 >= n) return;
  const AdamConsts k = {one_minus_b1, beta2, one_minus_b2, eps, step_size, bc2_sqrt, tau, one_minus_tau};
  const bool adv = i0 >= a1_w && i0 < a1_b, val = i0 >= v1_w && i0 < v1_b;
  unsigned o = 0, c = 0, rail = 0, pl = 0;
  if (adv || val) {
    const unsigned j = (unsigned)(i0 - (adv ? a1_w : v1_w));
    o = div_small(j, K1, inv_K1);
    const unsigned rem = j - o * K1;
    c = div_small(rem, H3, inv_H3);
    const unsigned r2 = rem - c * H3;
    pl = div_small(r2, Rp, inv_Rp);
    rail = r2 - pl * Rp;
  }
  const float* gb = g + (adv ? a1_b : v1_b);
  const unsigned Kc = 192u * R4;
  float pv[4], mv[4], vv[4], tv[4], gv[4];
  bool synth[4], owner[4];
  unsigned pair[4], pk[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    synth[e] = (adv || val) && rail >= R4;
    gv[e] = synth[e] ? rcb[c] * gb[o] : 0.f;
    pair[e] = ((adv ? 0u : 512u) + o) * 64u + c;
    pk[e] = o * Kc + (c * 3u + pl) * R4 + rail;               // (meaningful under the used rails)
    owner[e] = synth[e] && pl == 0 && rail == R4;
    if (++rail == Rp) {
      rail = 0;
      if (++pl == 3) {
        pl = 0;
        if (++c == 64) { c = 0; o++; }
      }
    }
  }
  if (COMPACT && (adv || val)) {
    const float4 p4 = *reinterpret_cast<const float4*>(p + i0), t4 = *reinterpret_cast<const float4*>(target + i0);
    pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
    tv[0] = t4.x; tv[1] = t4.y; tv[2] = t4.z; tv[3] = t4.w;
    const float* gp = g + (adv ? a1_w : v1_w);
    float* mp = mpk + (adv ? 0 : (size_t)512 * Kc);
    float* vp = vpk + (adv ? 0 : (size_t)512 * Kc);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (synth[e]) {
        const float2 mvp = mv_in[pair[e]];
        mv[e] = mvp.x;
        vv[e] = mvp.y;
      } else {
        gv[e] = gp[pk[e]];
        mv[e] = mp[pk[e]];
        vv[e] = vp[pk[e]];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      adam_element(pv[e], gv[e], mv[e], vv[e], k);
      tv[e] = soft_element(pv[e], tv[e], k);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (!synth[e]) {
        mp[pk[e]] = mv[e];
        vp[pk[e]] = vv[e];
      } else if (owner[e]) {
        mv_out[pair[e]] = make_float2(mv[e], vv[e]);
      }
    }
    *reinterpret_cast<float4*>(p + i0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    *reinterpret_cast<float4*>(target + i0) = make_float4(tv[0], tv[1], tv[2], tv[3]);
    return;
  }
  const bool whole = i0 + 3 < n;
  if (whole) {
    const float4 p4 = *reinterpret_cast<const float4*>(p + i0), t4 = *reinterpret_cast<const float4*>(target + i0);
    const float4 m4 = *reinterpret_cast<const float4*>(m + i0), v4 = *reinterpret_cast<const float4*>(v + i0);
    pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
    tv[0] = t4.x; tv[1] = t4.y; tv[2] = t4.z; tv[3] = t4.w;
    mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
    vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
    if (!(synth[0] && synth[1] && synth[2] && synth[3])) {
      const float4 g4 = *reinterpret_cast<const float4*>(g + i0);
      if (!synth[0]) gv[0] = g4.x;
      if (!synth[1]) gv[1] = g4.y;
      if (!synth[2]) gv[2] = g4.z;
      if (!synth[3]) gv[3] = g4.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; e++)
      if (i0 + e < n) {
        pv[e] = p[i0 + e]; mv[e] = m[i0 + e]; vv[e] = v[i0 + e]; tv[e] = target[i0 + e];
        if (!synth[e]) gv[e] = g[i0 + e];
      } else {
        pv[e] = mv[e] = vv[e] = tv[e] = gv[e] = 0.f;
      }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) {
    adam_element(pv[e], gv[e], mv[e], vv[e], k);
    tv[e] = soft_element(pv[e], tv[e], k);
  }
  if (whole) {
    *reinterpret_cast<float4*>(p + i0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    *reinterpret_cast<float4*>(target + i0) = make_float4(tv[0], tv[1], tv[2], tv[3]);
    *reinterpret_cast<float4*>(m + i0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *reinterpret_cast<float4*>(v + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; e++)
      if (i0 + e < n) { p[i0 + e] = pv[e]; m[i0 + e] = mv[e]; v[i0 + e] = vv[e]; target[i0 + e] = tv[e]; }
  }
}

// The optimizer's moments of the two weight blocks, between their two forms.  One workgroup per (stream, output unit).
// take: pair[(stream, o, c)] = (m, v) at plane 0, rail R4 (one of the equal ones under the empty rows), and the moments
// under the used rails into mpk / vpk [2][512][Kc].  fill: the other way, every m and v under the empty rows from its pair.
template <bool FILL>
__global__ __launch_bounds__(256) void adam_moments_kernel(float* __restrict__ m, float* __restrict__ v, long a1_w, long v1_w,
                                                           long K1, int H3, int Rp, int R4, float2* __restrict__ pairs,
                                                           float* __restrict__ mpk, float* __restrict__ vpk) {
  const int o = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;      // o: stream*512 + unit
  const long base = (o < 512 ? a1_w + (long)o * K1 : v1_w + (long)(o - 512) * K1);
  const long Kc = 192L * R4;
  const int L = Rp - R4;
  for (int sg = w; sg < 192; sg += 4) {                                        // segment = (filter, plane)
    const int c = sg / 3, pl = sg - 3 * c;
    const long at = base + (long)c * H3 + (long)pl * Rp;
    const long pat = (long)o * Kc + (long)sg * R4;
    if (FILL) {
      const float2 mvp = pairs[o * 64 + c];
      for (int i = lane; i < L; i += 64) {
        m[at + R4 + i] = mvp.x;
        v[at + R4 + i] = mvp.y;
      }
      for (int i = lane; i < R4; i += 64) {
        m[at + i] = mpk[pat + i];
        v[at + i] = vpk[pat + i];
      }
    } else {
      if (pl == 0 && lane == 0) pairs[o * 64 + c] = make_float2(m[at + R4], v[at + R4]);
      for (int i = lane; i < R4; i += 64) {
        mpk[pat + i] = m[at + i];
        vpk[pat + i] = v[at + i];
      }
    }
  }
}

// target <- tau*local + (1-tau)*target (src/agent.py:120-131)
__global__ void soft_update_kernel(float* __restrict__ target, const float* __restrict__ local, long n, float tau,
                                   float one_minus_tau) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const AdamConsts k = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, tau, one_minus_tau};
  target[i] = soft_element(local[i], target[i], k);
}

// uint8 observation planes (own plane is int8) -> fp32 rows [n, H3=3*Rp, T1]
// Workgroup = (piece of a plane, plane of a sample): no division per element; a thread converts four consecutive
// bytes and stores 16 bytes where the output is aligned for it.
__global__ __launch_bounds__(256) void obs_to_f32_kernel(const uint8_t* __restrict__ obs, int n, int Rp, int T1, int plane_stride,
                                                         float* __restrict__ out) {
  const long per = (long)Rp * T1;
  const int sp = blockIdx.y;                              // sample * 3 + plane
  const int pl = sp % 3;
  const uint8_t* src = obs + (long)sp * plane_stride;
  float* dst = out + (long)sp * per;
  const bool al = (reinterpret_cast<size_t>(dst) & 15) == 0, al_src = (reinterpret_cast<size_t>(src) & 3) == 0;
  for (long o = ((long)blockIdx.x * 256 + threadIdx.x) * 4; o < per; o += (long)gridDim.x * 1024) {
    float f[4];
    if (al_src && o + 3 < per) {
      const unsigned w = *reinterpret_cast<const unsigned*>(src + o);
#pragma unroll
      for (int i = 0; i < 4; i++) f[i] = pl == 0 ? (float)(int8_t)(w >> (8 * i)) : (float)((w >> (8 * i)) & 0xffu);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint8_t u = o + i < per ? src[o + i] : 0;
        f[i] = pl == 0 ? (float)(int8_t)u : (float)u;
      }
    }
    if (al && o + 3 < per) {
      *reinterpret_cast<float4*>(dst + o) = make_float4(f[0], f[1], f[2], f[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
        if (o + i < per) dst[o + i] = f[i];
    }
  }
}

// pieces of 4 KiB a plane is cut into (grid x of obs_to_f32_kernel), at most 8
static unsigned obs_to_f32_pieces(int Rp, int T1) {
  const long per = (long)Rp * T1;
  long p = (per + 4095) / 4096;
  return (unsigned)(p < 1 ? 1 : p > 8 ? 8 : p);
}

// ---- convolution straight from uint8 observations ----------------------------------------------------
// x1[b][c][h] = relu(bias[c] + sum_k W[c][k] * obs(b, h, k)) for the 64 filters, where row h = plane*Rp + rail
// of an observation [3][plane_stride] (plane 0 holds int8, the heat planes uint8).
// Workgroups are persistent, one per CU (8 waves) while the rows fit the LDS: the filter bank (T1 x 64 floats)
// is staged in LDS once, then each wave
// works through tiles of 64 consecutive rows (b,h) on its own, without workgroup barriers.  Per tile a wave
//   * writes the tile's bytes, fetched into registers while the previous tile was being multiplied, into its
//     own 64 rows of LDS (a row starts at any byte address, so its bytes keep their offset within the first
//     word; the few bytes copied beyond the row meet zero filter weights).  Row addresses are wave-uniform and
//     advance in scalar registers; a lane always fetches the same word of every row,
//   * starts the loads of its next tile,
//   * multiplies filters (matrix rows) by observation rows (matrix columns) on the matrix cores, 2x2 tiles of
//     32x32: a lane then holds 32 filters of one observation row, and a store instruction writes 32
//     consecutive rows of one filter, which is the layout of x1.
// BF: the products run on the bf16 matrix cores.  An observation byte (-1..1 or a count below 255) is exact in
// bf16, and a filter weight is the exact sum of three bf16 terms (split3), so w*x is the sum of three exact partial
// products accumulated in fp32: three v_mfma_f32_32x32x16_bf16 of 32 cycles cover the sixteen k of eight fp32
// instructions of 64.  The filter bank is staged split as [term][filter][k] with k contiguous (a lane's operand is
// one 16-byte read; rows an odd number of 16-byte slots apart), a lane's eight bytes of an observation row are
// cut out of three aligned words (the row keeps its byte offset) and turned into bf16 in registers.
#define CV_KP16(T1) (((T1) + 15) & ~15)                         // K padded to whole bf16 instructions
#define CV_BROW(T1) (2 * CV_KP16(T1) + 16)                      // bytes per filter and term in LDS
#define CV_ASTRIDE(T1) ((((T1) + 6) / 4 * 4) | 4)              // bytes per row in LDS: covers 3 + T1 bytes, word count odd
#define CV_WSTRIDE 65                                            // floats per k in LDS
__device__ __forceinline__ void cv_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// four bytes of a word as two words of bf16 pairs; mode 0: unsigned bytes, 1: signed bytes, 2: signed where sg
__device__ __forceinline__ uint2 bytes_to_bf16(unsigned w, int mode, bool sg) {
  float f[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float fu = (float)((w >> (8 * i)) & 0xffu);
    const float fs = (float)(int)(int8_t)(w >> (8 * i));
    f[i] = mode == 0 ? fu : mode == 1 ? fs : (sg ? fs : fu);
  }
  return make_uint2(pack_hi16(__float_as_uint(f[0]), __float_as_uint(f[1])), pack_hi16(__float_as_uint(f[2]), __float_as_uint(f[3])));
}

// WPL: words a lane fetches per row (rows of up to 64*WPL words); NWAVE: waves per workgroup; JT: tiles of 32 rows a
// wave works on at a time (2: 64 rows share every filter fragment read; 1: half the rows per wave in LDS, so that
// twice the waves fit a CU and hide one another's memory latency)
struct ConvSecondJob {
  const uint8_t* obs;
  const float* Wc;
  const float* bias;
  long M, obs_bytes;
  float* x1;
  unsigned first_wg;     // >= gridDim.x: no second job
};

template <int WPL, int NWAVE, bool BF, int JT>
__global__ __launch_bounds__(64 * NWAVE, NWAVE / 4) void conv_u8_kernel(
    const uint8_t* __restrict__ obs, const float* __restrict__ Wc, const float* __restrict__ bias, long M, int Rp, int T1,
    int H3, int plane_stride, long obs_bytes, float* __restrict__ x1, ConvSecondJob j2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cv_lds[];
  constexpr int CV_TILE = 32 * JT;
  // a launch may carry a second convolution of the same shape with its own filters, input and output (the target
  // network beside the local one in a learn step): the workgroups from j2.first_wg on work on it
  const bool second = blockIdx.x >= j2.first_wg;
  if (second) {
    obs = j2.obs; Wc = j2.Wc; bias = j2.bias; M = j2.M; obs_bytes = j2.obs_bytes; x1 = j2.x1;
  }
  const unsigned n_wg = second ? gridDim.x - j2.first_wg : min(gridDim.x, j2.first_wg);
  const unsigned wg = second ? blockIdx.x - j2.first_wg : blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int astr = CV_ASTRIDE(T1);
  const int KP = (T1 + 1) & ~1;                                  // K padded to a whole MFMA step
  const int nword = (T1 + 6) >> 2;                               // words that cover any alignment of T1 bytes
  unsigned char* Ab = cv_lds + (size_t)wave * CV_TILE * astr;    // this wave's rows: [CV_TILE][astr]
  unsigned char* bank = cv_lds + (((size_t)NWAVE * CV_TILE * astr + 15) & ~(size_t)15);
  float* Wl = reinterpret_cast<float*>(bank);                    // !BF: [KP][CV_WSTRIDE] floats
  const int KP16 = CV_KP16(T1), brow = CV_BROW(T1);              // BF: [3][64][brow] bytes
  float* bl = BF ? reinterpret_cast<float*>(bank + (size_t)3 * 64 * brow) : Wl + (size_t)KP * CV_WSTRIDE;   // [64]
  if (BF) {
    for (int i = tid; i < 64 * KP16; i += 64 * NWAVE) {
      const int c = i / KP16, k = i - c * KP16;
      unsigned h, m, l;
      split3(k < T1 ? Wc[(long)c * T1 + k] : 0.f, h, m, l);
      unsigned short* d = reinterpret_cast<unsigned short*>(bank + (size_t)c * brow) + k;
      d[0] = (unsigned short)(h >> 16);
      d[(size_t)64 * brow / 2] = (unsigned short)(m >> 16);
      d[(size_t)2 * 64 * brow / 2] = (unsigned short)(l >> 16);
    }
  } else {
    // filters, transposed to [k][c]: a wave takes 64/NWAVE filters, lanes along k, all its loads in flight
    constexpr int FPW = 64 / NWAVE;
    for (int kb = 0; kb < KP; kb += 64) {
      const int k = kb + lane;
      float v[FPW];
#pragma unroll
      for (int ci = 0; ci < FPW; ci++) v[ci] = k < T1 ? Wc[(long)(wave * FPW + ci) * T1 + k] : 0.f;
      if (k < KP) {
#pragma unroll
        for (int ci = 0; ci < FPW; ci++) Wl[k * CV_WSTRIDE + wave * FPW + ci] = v[ci];
      }
    }
  }
  if (tid < 64) bl[tid] = bias[tid];
  __syncthreads();

  const unsigned uM = (unsigned)M, uH3 = (unsigned)H3, uRp = (unsigned)Rp;      // M < 2^31 (checked by the host)
  const unsigned NT = (uM + CV_TILE - 1) / CV_TILE;
  const unsigned GW = n_wg * NWAVE;
  // byte offsets are taken from the 4-byte aligned address at or below obs
  const unsigned mis = (unsigned)(reinterpret_cast<size_t>(obs) & 3);
  const uint8_t* obs_al = obs - mis;
  const size_t end_off = ((size_t)mis + (size_t)obs_bytes + 3) & ~(size_t)3;
  // the words of a row this lane fetches; lanes beyond the row's last word repeat that word
  int wc[WPL];
#pragma unroll
  for (int x = 0; x < WPL; x++) wc[x] = min(lane + 64 * x, nword - 1);
  unsigned pre[CV_TILE * WPL];
  // of the tile being fetched: rows it has, element offset of its first row in x1 (both wave-uniform); per row
  // (row = lane): byte offset within the first word | 4 if the plane is signed, and its element offset in x1
  // relative to the first row's
  unsigned nxt_rows = 0;
  long nxt_out0 = 0;
  int nxt_os = 0;
  unsigned nxt_drel = 0;
  auto prefetch = [&](unsigned tile) {
    const unsigned m0 = tile * CV_TILE;
    unsigned b = m0 / uH3;
    const unsigned h0 = m0 - b * uH3;
    unsigned pl = h0 / uRp, rail = h0 - pl * uRp;
    size_t off = (size_t)mis + ((size_t)b * 3 + pl) * (size_t)plane_stride + (size_t)rail * T1;
    nxt_rows = min((unsigned)CV_TILE, uM - m0);
    nxt_out0 = (long)b * 64 * H3 + h0;
    {   // this lane's row
      const unsigned m = m0 + lane;
      const unsigned bb = m / uH3, hh = m - bb * uH3, pp = hh / uRp, rr = hh - pp * uRp;
      const size_t o = (size_t)mis + ((size_t)bb * 3 + pp) * (size_t)plane_stride + (size_t)rr * T1;
      nxt_os = (int)(o & 3) | (pp == 0 ? 4 : 0);
      nxt_drel = (bb - b) * 64u * uH3 + hh - h0;
    }
#pragma unroll
    for (int j = 0; j < CV_TILE; j++) {
      if ((unsigned)j < nxt_rows) {
        const size_t a = off & ~(size_t)3;
        const unsigned* src = reinterpret_cast<const unsigned*>(obs_al + a);
        if (a + (size_t)nword * 4 <= end_off) {
#pragma unroll
          for (int x = 0; x < WPL; x++) pre[j * WPL + x] = src[wc[x]];
        } else {                           // the last rows of the buffer: no word beyond its end is read
#pragma unroll
          for (int x = 0; x < WPL; x++) {
            const bool ok = a + (size_t)wc[x] * 4 + 4 <= end_off;
            pre[j * WPL + x] = src[ok ? wc[x] : 0];
          }
        }
        off += (size_t)T1;
        if (++rail == uRp) {
          rail = 0;
          if (++pl == 3) { pl = 0; b++; }
          off = (size_t)mis + ((size_t)b * 3 + pl) * (size_t)plane_stride;
        }
      }
    }
  };

  const int c32 = lane & 31, kh = lane >> 5;
  const float* wp = Wl + kh * CV_WSTRIDE + c32;
  auto stash = [&]() {   // the fetched words go to this wave's rows
#pragma unroll
    for (int j = 0; j < CV_TILE; j++) {
      if ((unsigned)j < nxt_rows) {
#pragma unroll
        for (int x = 0; x < WPL; x++) *reinterpret_cast<unsigned*>(Ab + j * astr + wc[x] * 4) = pre[j * WPL + x];
      }
    }
  };
  // Software pipeline over this wave's tiles: while tile t is multiplied its successor's bytes are in flight;
  // they are written to LDS right after the products, and only then are the outputs of tile t stored, so that
  // those stores drain during the next tile's products instead of in front of a wait.
  // (tile t goes to workgroup t % n_wg: when the tiles are not a multiple of the waves, the waves with one
  // tile more are spread over all CUs instead of filling the first few)
  unsigned t = __builtin_amdgcn_readfirstlane(wave * n_wg + wg);
  unsigned cur_rows = 0, cur_drel = 0;
  long cur_out0 = 0;
  int cur_os = 0;
  if (t < NT) {
    prefetch(t);
    stash();
    cur_rows = nxt_rows; cur_out0 = nxt_out0; cur_os = nxt_os; cur_drel = nxt_drel;
    cv_wave_sync();
    if (t + GW < NT) prefetch(t + GW);
  }
  for (; t < NT; t += GW) {
    int os[JT];                              // of the lane's row in every row tile: byte offset | 4 if signed
#pragma unroll
    for (int j = 0; j < JT; j++) os[j] = __shfl(cur_os, c32 + 32 * j);
    // the accumulators start from the bias of their filter (32 look-ups per tile instead of one per stored value)
    f32x16 acc[2][JT];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float bv = bl[32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh];
#pragma unroll
        for (int j = 0; j < JT; j++) acc[i][j][r] = bv;
      }
    if (BF) {
      // per row tile: the lane's row, the byte offset of the row inside its first word, and whether all / none /
      // some of the tile's rows are signed (wave-uniform)
      const unsigned char* rp[JT];
      unsigned sh[JT];
      bool sg[JT];
      int mode[JT];
#pragma unroll
      for (int j = 0; j < JT; j++) {
        rp[j] = Ab + (c32 + 32 * j) * astr + 8 * kh;
        sh[j] = (unsigned)os[j] & 3u;
        sg[j] = (os[j] & 4) != 0;
        const unsigned long long bs = __ballot(sg[j]);
        mode[j] = bs == 0ull ? 0 : bs == ~0ull ? 1 : 2;
      }
      const unsigned char* ap = bank + (size_t)c32 * brow + 16 * kh;
      for (int st = 0; st < KP16; st += 16) {
        bf16x8 fa[2][3], fb[JT];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int t = 0; t < 3; t++)
            fa[i][t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ap + ((size_t)t * 64 + 32 * i) * brow + 2 * st));
#pragma unroll
        for (int j = 0; j < JT; j++) {
          const unsigned* wp3 = reinterpret_cast<const unsigned*>(rp[j] + st);
          const unsigned w0 = wp3[0], w1 = wp3[1], w2 = wp3[2];
          const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, sh[j]), hi = __builtin_amdgcn_alignbyte(w2, w1, sh[j]);
          const uint2 p = bytes_to_bf16(lo, mode[j], sg[j]), q = bytes_to_bf16(hi, mode[j], sg[j]);
          fb[j] = __builtin_bit_cast(bf16x8, make_uint4(p.x, p.y, q.x, q.y));
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < JT; j++) {
            f32x16 c = acc[i][j];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j], c, 0, 0, 0);
            acc[i][j] = c;
          }
      }
    } else {
    // sign extension from bit 7 for the signed plane, from bit 8 (never set) for the others
    unsigned wd[JT];
    const unsigned char* bp[JT];
#pragma unroll
    for (int j = 0; j < JT; j++) {
      wd[j] = os[j] & 4 ? 8u : 9u;
      bp[j] = Ab + (c32 + 32 * j) * astr + kh + (os[j] & 3);
    }
    auto kstep = [&](int kk) {
      const float f0 = wp[kk * CV_WSTRIDE], f1 = wp[kk * CV_WSTRIDE + 32];
#pragma unroll
      for (int j = 0; j < JT; j++) {
        const unsigned u = bp[j][kk];
        const float x = (float)(int)__builtin_amdgcn_sbfe((int)u, 0u, wd[j]);
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, x, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1, x, acc[1][j], 0, 0, 0);
      }
    };
    int kk = 0;
    for (; kk + 8 <= KP; kk += 8) {
#pragma unroll
      for (int u = 0; u < 4; u++) kstep(kk + 2 * u);
    }
    for (; kk < KP; kk += 2) kstep(kk);
    }
    cv_wave_sync();                        // every lane has read its rows before the next tile overwrites them
    const unsigned o_rows = cur_rows;
    const unsigned d_lane = cur_drel;
    float* dst0 = x1 + cur_out0;
    if (t + GW < NT) {
      stash();
      cur_rows = nxt_rows; cur_out0 = nxt_out0; cur_os = nxt_os; cur_drel = nxt_drel;
      cv_wave_sync();
      if (t + 2 * GW < NT) prefetch(t + 2 * GW);
    }
    // acc[i][j][r]: filter 32*i + (r&3) + 8*(r>>2) + 4*kh of observation row 32*j + c32
#pragma unroll
    for (int j = 0; j < JT; j++) {
      const unsigned d = __shfl(d_lane, c32 + 32 * j) + 4u * kh * uH3;
      if ((unsigned)(c32 + 32 * j) < o_rows) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int c = 32 * i + (r & 3) + 8 * (r >> 2);
            dst0[d + (unsigned)c * uH3] = fmaxf(acc[i][j][r], 0.f);
          }
      }
    }
  }
}

// ---- backward pass over the used rails: what the empty rails contribute ---------------------------------------------
// The first dense layer's weights under the empty rails see the constant relu(conv_b[c]) for every sample, so their
// gradient is dW[o][c*H3 + plane*Rp + rail] = relu(conv_b[c]) * gsum[o] (gsum = the layer's bias gradient, the column
// sums of dhid) for rail >= R4.  One workgroup per output unit, lanes along a segment's consecutive weights.
__global__ __launch_bounds__(256) void fc1_wgrad_padding_kernel(const float* __restrict__ gsum, const float* __restrict__ conv_b,
                                                                float* __restrict__ dWa, float* __restrict__ dWv, long K1, int H3,
                                                                int Rp, int R4) {
  const int o = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* W = o < 512 ? dWa + (long)o * K1 : dWv + (long)(o - 512) * K1;
  const float go = gsum[o];
  const int L = Rp - R4;
  for (int sg = w; sg < 192; sg += 4) {
    const int c = sg / 3, pl = sg - 3 * c;
    const float v = fmaxf(conv_b[c], 0.f) * go;
    float* p = W + (long)c * H3 + (long)pl * Rp + R4;
    for (int i = lane; i < L; i += 64) p[i] = v;
  }
}

// ... and the convolution's bias gradient gains, for every filter that is active on an empty row (conv_b[c] > 0), the
// sum over those rows of the activation gradient: sum over o of gsum[o] * wpad[c][o] (wpad: fc1_padding_bias_kernel).
// conv_wgrad_reduce_kernel adds it where it writes the bias gradient.

// ---- the convolution's weight and bias gradients straight from uint8 observations --------------------------------
// dWc[c][k] = sum over samples b and rows h of dx1[b][c][h] * s[b][h][k],  dbias[c] = sum of dx1[b][c][h].
// On the matrix cores with the rows (b,h) as the inner dimension, 16 rows per instruction: A = dx1, fp32 split into
// three bf16 terms, B = the observation bytes, exact in bf16, so three instructions per 16 rows and 32x32 tile and
// every partial product exact (the product from float states needs six, on tiles half empty).  Column T1 of B is all
// ones: its column of the result is the bias gradient.  Both operands come straight from memory: a lane's eight
// consecutive rows of one channel of dx1 are contiguous, and its eight rows of one byte column of s are eight byte
// loads (no transposition through LDS).
// One workgroup per CU, eight waves = 2 channel halves x 4; the four waves of a half share out the 16-row chunks of the
// workgroup's samples (sample blockIdx.x, blockIdx.x + gridDim.x, ...), each accumulating a [32, 32*NT] tile in
// registers with the next chunk's loads in flight; the four tiles of a half meet in LDS in a fixed order, the workgroup
// writes one partial [64, 32*NT], and conv_wgrad_reduce_kernel adds the workgroups' partials in a fixed order.
template <int NT>
__global__ __launch_bounds__(512, 2) void conv_wgrad_u8_kernel(const float* __restrict__ dx1, const uint8_t* __restrict__ obs,
                                                               int B, int Rp, int T1, int plane_stride,
                                                               float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];   // [2 halves][3 waves][NT * 16][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = wave & 1, q = wave >> 1;
  const int n = lane & 31, kg = lane >> 5;
  const int H3 = 3 * Rp;
  const int NC = (H3 + 15) >> 4;                       // chunks of 16 rows per sample
  const int G = gridDim.x;
  const int n_task = ((B - 1 - (int)blockIdx.x) / G + 1) * NC;
  const int c = 32 * half + n;
  f32x16 acc[NT];
#pragma unroll
  for (int kt = 0; kt < NT; kt++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[kt][r] = 0.f;

  // this lane's share of chunk t: eight rows of channel c of dx1 (zero past the sample's last row), the same eight
  // rows of NT byte columns of s, and which of the rows are signed (plane 0).  Every load is unconditional, from an
  // address clamped into the arrays, and what must not count is replaced afterwards: no branch stands between the
  // loads, so those of the next chunk stay in flight under the products of this one.
  const int kcol[2] = {min(32 * (NT - 1) + n, T1 - 1), 0};
  auto fetch = [&](int t, float (&a)[8], unsigned (&bb)[NT][8], unsigned& sgn) {
    t = min(t, n_task - 1);
    const int si = t / NC, j = t - si * NC;
    const int b = (int)blockIdx.x + si * G;
    const int hb = 16 * j + 8 * kg;
    const float* pa = dx1 + ((long)b * 64 + c) * H3;
    const uint8_t* ob = obs + (long)b * 3 * plane_stride;
    sgn = 0;
    // eight consecutive floats as two 16-byte loads (4-byte aligned is enough for global memory; one load per lane
    // and cache line instead of four).  Past the sample's last row they run into the next channel's rows or, at the
    // very end of dx1, into the caller's next buffer (at most 15 floats; the workspace continues there): masked below.
    struct __attribute__((aligned(4))) F4 { float v[4]; };
    const F4 lo = *reinterpret_cast<const F4*>(pa + hb), hi = *reinterpret_cast<const F4*>(pa + hb + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a[i] = lo.v[i];
      a[4 + i] = hi.v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int h = min(hb + i, H3 - 1);               // (a row past the end is multiplied by zero)
      const int pl = (h >= Rp ? 1 : 0) + (h >= 2 * Rp ? 1 : 0);
      const uint8_t* row = ob + (long)pl * plane_stride + (long)(h - pl * Rp) * T1;
      sgn |= pl == 0 ? 1u << i : 0u;
#pragma unroll
      for (int kt = 0; kt < NT; kt++) bb[kt][i] = row[kt + 1 < NT ? 32 * kt + n : kcol[0]];
    }
  };
  // (after the loads) rows past the sample's end and columns past the horizon: zero, the bias column: one
  auto mask = [&](int t, float (&a)[8], unsigned (&bb)[NT][8]) {
    t = min(t, n_task - 1);
    const int j = t % NC;
    const int hb = 16 * j + 8 * kg;
    const int k = 32 * (NT - 1) + n;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      a[i] = hb + i < H3 ? a[i] : 0.f;
      bb[NT - 1][i] = k < T1 ? bb[NT - 1][i] : (k == T1 ? 1u : 0u);
    }
  };
  auto compute = [&](const float (&a)[8], const unsigned (&bb)[NT][8], unsigned sgn) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; i++) split3(a[i], h[i], m[i], l[i]);
    bf16x8 fa[3];
    fa[0] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7])));
    fa[1] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7])));
    fa[2] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7])));
    bf16x8 fb[NT];
#pragma unroll
    for (int kt = 0; kt < NT; kt++) {
      unsigned f[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int v = (sgn >> i) & 1u ? (int)(signed char)bb[kt][i] : (int)bb[kt][i];
        f[i] = __float_as_uint((float)v);             // -1..255: exact in the upper half of the word
      }
      fb[kt] = __builtin_bit_cast(bf16x8, make_uint4(pack_hi16(f[0], f[1]), pack_hi16(f[2], f[3]), pack_hi16(f[4], f[5]), pack_hi16(f[6], f[7])));
    }
#pragma unroll
    for (int sp = 2; sp >= 0; sp--)                    // smallest terms first
#pragma unroll
      for (int kt = 0; kt < NT; kt++) acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sp], fb[kt], acc[kt], 0, 0, 0);
  };

  float a0[8], a1[8];
  unsigned b0[NT][8], b1[NT][8], s0 = 0, s1 = 0;
  fetch(q, a0, b0, s0);
  for (int t = q; t < n_task; t += 8) {
    fetch(t + 4, a1, b1, s1);
    mask(t, a0, b0);
    compute(a0, b0, s0);
    fetch(t + 8, a0, b0, s0);
    if (t + 4 < n_task) {
      mask(t + 4, a1, b1);
      compute(a1, b1, s1);
    }
  }
  // the four tiles of a half, added in wave order
  if (q > 0) {
    float* slot = wg_lds + (size_t)((half * 3 + (q - 1)) * NT * 16) * 64;
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) slot[(kt * 16 + r) * 64 + lane] = acc[kt][r];
  }
  __syncthreads();
  if (q == 0) {
    for (int qq = 0; qq < 3; qq++) {
      const float* slot = wg_lds + (size_t)((half * 3 + qq) * NT * 16) * 64;
#pragma unroll
      for (int kt = 0; kt < NT; kt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[kt][r] += slot[(kt * 16 + r) * 64 + lane];
    }
    float* out = part + (size_t)blockIdx.x * 64 * (32 * NT);
#pragma unroll
    for (int kt = 0; kt < NT; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int cc = 32 * half + (r & 3) + 8 * (r >> 2) + 4 * kg;
        out[(size_t)cc * (32 * NT) + 32 * kt + n] = acc[kt][r];
      }
  }
}

// dW[c][k] (k < T1) and db[c] (k == T1) = sum over the G workgroup partials part[g][c][k] of row width W: sixteen waves
// take every sixteenth partial each (two sums in flight per lane), their sums meet in LDS in wave order.
// gsum != nullptr (learn step over the used rails): the workgroup that holds the bias element of a filter with
// conv_b[c] > 0 also adds the empty rows' share, sum over o of gsum[o] * wpad[c][o]: thread t < 256 takes o = t, t + 256,
// t + 512, t + 768, the 256 partial sums meet in LDS by halving.
__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int G, int W, int T1,
                                                                 float* __restrict__ dW, float* __restrict__ db,
                                                                 const float* __restrict__ gsum, const float* __restrict__ wpad,
                                                                 const float* __restrict__ conv_b) {
  __shared__ float sm[16][64];
  __shared__ float pad[256], padv[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const bool valid = idx < 64 * (T1 + 1);
  const int c = valid ? idx / (T1 + 1) : 0, k = valid ? idx - c * (T1 + 1) : 0;
  const float* p = part + (size_t)c * W + k;
  const size_t gs = (size_t)64 * W;
  float s0 = 0.f, s1 = 0.f;
  int g = w;
  for (; g + 16 < G; g += 32) {
    s0 += p[(size_t)g * gs];
    s1 += p[(size_t)(g + 16) * gs];
  }
  if (g < G) s0 += p[(size_t)g * gs];
  sm[w][lane] = s0 + s1;
  // the bias element of filter cb lies at index cb*(T1+1) + T1: those among this workgroup's 64 indices, one by one
  // (the loop is the same for the whole workgroup)
  const int first = blockIdx.x * 64, t = threadIdx.x;
  if (gsum != nullptr) {
    for (int cb = first / (T1 + 1); cb < 64; cb++) {          // (the smallest cb with cb*(T1+1) + T1 >= first)
      const int at = cb * (T1 + 1) + T1;
      if (at >= first + 64) break;
      if (t < 256) {
        const float* wp = wpad + (long)cb * 1024;
        pad[t] = (gsum[t] * wp[t] + gsum[t + 256] * wp[t + 256]) + (gsum[t + 512] * wp[t + 512] + gsum[t + 768] * wp[t + 768]);
      }
      __syncthreads();
      for (int st = 128; st > 0; st >>= 1) {
        if (t < st) pad[t] += pad[t + st];
        __syncthreads();
      }
      if (t == 0) padv[at - first] = pad[0];
      __syncthreads();
    }
  }
  __syncthreads();
  if (w == 0 && valid) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) v += sm[i][lane];
    if (k < T1) dW[(size_t)c * T1 + k] = v;
    else db[c] = gsum != nullptr && conv_b[c] > 0.f ? v + padv[lane] : v;
  }
}

// the byte form of the convolution's weight gradient covers horizons up to depth 158 (five column tiles with the
// bias column) 
static int conv_wgrad_u8_tiles(int T1) { return T1 + 1 <= 160 ? (T1 + 1 + 31) / 32 : 0; }
static int conv_wgrad_u8_groups(int B) { return B < 256 ? B : 256; }

template <int NT>
static int launch_conv_wgrad_u8(const float* dx1, const uint8_t* obs, int B, int Rp, int T1, int plane_stride, float* part,
                                float* dW, float* db, hipStream_t stream, const float* gsum = nullptr,
                                const float* wpad = nullptr, const float* conv_b = nullptr) {
  const int G = conv_wgrad_u8_groups(B);
  const size_t lds = (size_t)6 * NT * 16 * 64 * sizeof(float);
  FLM_RAISE_LDS_ONCE((conv_wgrad_u8_kernel<NT>));
  hipLaunchKernelGGL((conv_wgrad_u8_kernel<NT>), dim3(G), dim3(512), lds, stream, dx1, obs, B, Rp, T1, plane_stride, part);
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((64 * (T1 + 1) + 63) / 64)), dim3(1024), 0, stream, part, G,
                     32 * NT, T1, dW, db, gsum, wpad, conv_b);
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

static size_t conv_u8_lds(int T1, int nwave, bool bf, int tile_rows = 64) {
  const size_t rows = ((size_t)nwave * tile_rows * CV_ASTRIDE(T1) + 15) & ~(size_t)15;
  // BF: 16 spare bytes behind the bias: the last row's third word may lie behind the rows of the last wave
  return bf ? rows + (size_t)3 * 64 * CV_BROW(T1) + 64 * 4 + 16 : rows + ((size_t)((T1 + 1) & ~1) * CV_WSTRIDE + 64) * 4;
}

// second (optional): another convolution of the same shape in the same launch - its own filters (params2: a network's
// flat parameters), observations and output; the 256 workgroups are shared out by the number of tiles
struct ConvSecond {
  const float* Wc;
  const float* bias;
  const uint8_t* obs;
  long M, obs_bytes;
  float* x1;
};

template <int WPL, int NWAVE, bool BF, int JT = 2>
static int launch_conv_u8(const uint8_t* obs_u8, const float* Wc, const float* bias, long M, int Rp, int T1, int H3,
                          int plane_stride, long obs_bytes, float* x1, hipStream_t stream, const ConvSecond* second = nullptr) {
  constexpr int CV_TILE = 32 * JT;
  const size_t lds = conv_u8_lds(T1, NWAVE, BF, CV_TILE);
  if (lds > 160 * 1024 - 512) return FLM_E_ARG;
  FLM_RAISE_LDS_ONCE((conv_u8_kernel<WPL, NWAVE, BF, JT>));
  // persistent workgroups: one per CU (256 CUs)
  const long tiles = (M + CV_TILE - 1) / CV_TILE;
  long wgs = (tiles + NWAVE - 1) / NWAVE;
  if (wgs > 256) wgs = 256;
  if (wgs < 1) wgs = 1;
  ConvSecondJob j2{};
  j2.first_wg = 0xFFFFFFFFu;
  if (second) {
    const long tiles2 = (second->M + CV_TILE - 1) / CV_TILE;
    long wgs2 = (tiles2 + NWAVE - 1) / NWAVE;
    if (wgs2 < 1) wgs2 = 1;
    if (wgs + wgs2 > 256) {                   // share the CUs by the work
      wgs = (256 * tiles + (tiles + tiles2) / 2) / (tiles + tiles2);
      if (wgs < 1) wgs = 1;
      if (wgs > 255) wgs = 255;
      wgs2 = 256 - wgs;
    }
    j2 = ConvSecondJob{second->obs, second->Wc, second->bias, second->M, second->obs_bytes, second->x1, (unsigned)wgs};
    wgs += wgs2;
  }
  hipLaunchKernelGGL((conv_u8_kernel<WPL, NWAVE, BF, JT>), dim3((unsigned)wgs), dim3(64 * NWAVE), lds, stream, obs_u8, Wc, bias,
                     M, Rp, T1, H3, plane_stride, obs_bytes, x1, j2);
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

}  // namespace flm

// K slices of the first dense layer (advantage and value streams side by side: 8 column tiles).
// fc1_k_slices_cap(B): the most slices a batch of B rows is ever given - what workspaces are sized for: the count with
// the least rounds(n) / n of a pass over K (rounds = the launch's workgroups over the 512 that fit the chip, rounded up)
// plus n partial sums to write and reduce.
static int fc1_k_slices_cap(int B) {
  const long tiles = (long)((B + GBM - 1) / GBM) * 8;
  int best = 1;
  double best_cost = 0;
  for (int n = 1; n <= 32; n++) {
    const double cost = (double)((tiles * n + 511) / 512) / n + n * (double)B * 1e-6;
    if (n == 1 || cost < best_cost) {
      best = n;
      best_cost = cost;
    }
  }
  return best;
}

// The slices used for a product over K columns: a launch takes as long as its fullest CU, and a CU works its
// workgroups off at one rate whether one or two are resident, so the estimate is, in units of one k of one tile,
// ceil(workgroups / 256) * (K / n + 30) for the products and each workgroup's start and end, plus 0.028 * n * B for
// the partial sums.  Measured points it reproduces (forward over 20 of 70 rails, K = 3,840): 512 rows 8 slices
// (74 us; 16: 78, 4: 87), 1,024 rows 4 (97; 8: 99, 2: 123), 2,250 rows 3 or 7 (186, 188; 4: 195, 2: 220).
static int fc1_k_slices(int B, long K) {
  const long tiles = (long)((B + GBM - 1) / GBM) * 8;
  const int cap = fc1_k_slices_cap(B);
  int best = 1;
  double best_cost = 0;
  for (int n = 1; n <= cap; n++) {
    const double cost = (double)((tiles * n + 255) / 256) * ((double)K / n + 30.0) + 0.028 * n * (double)B;
    if (n == 1 || cost < best_cost) {
      best = n;
      best_cost = cost;
    }
  }
  return best;
}

using flm::DqnLayout;
using flm::GemmArgs;

static bool actions_ok(int NA) { return NA >= 1 && NA <= FLM_DQN_MAX_ACTIONS; }

extern "C" long flm_dqn_param_count(int H3, int T1, int NA) { return actions_ok(NA) ? DqnLayout(H3, T1, NA).total : 0; }

// workspace floats needed by flm_dqn_forward for a batch of B samples
extern "C" long flm_dqn_forward_workspace(int B, int H3) {
  return (long)B * 64 * H3 + (long)B * 1024 + (long)fc1_k_slices_cap(B) * B * 1024 + 1024;   // x1, hid, K slices, a bias vector
}

// small batches (act): the first dense layer streams both weight matrices in one launch
static bool dqn_rows_path(const float* params, const DqnLayout& L, int B, const float* x1) {
  return B <= 8 && (L.K1 & 3) == 0 &&
         (((uintptr_t)(params + L.a1_w) | (uintptr_t)(params + L.v1_w) | (uintptr_t)x1) & 15) == 0;
}

// The convolution: x1[B,64,H3] from x, the float input [B,H3,T1], or (x == nullptr) from obs_u8 [B,3,plane_stride]
// with H3 = 3*Rp.
// params2 (byte input only): a second network's convolution over obs2 [B2,3,plane_stride] into x1_2, in the same launch.
static int dqn_conv_stage(const float* params, const float* x, const uint8_t* obs_u8, int plane_stride, int B, int H3,
                          int T1, int NA, float* x1, hipStream_t stream, const float* params2 = nullptr,
                          const uint8_t* obs2 = nullptr, int B2 = 0, float* x1_2 = nullptr) {
  DqnLayout L(H3, T1, NA);
  GemmArgs g{};
  int rc;
  const bool rows_path = dqn_rows_path(params, L, B, x1);
  const size_t conv_rows_lds = sizeof(float) * ((size_t)T1 * 65 + 4 * (size_t)T1);
  if (x && rows_path && conv_rows_lds <= 64 * 1024) {
    const long M = (long)B * H3;
    hipLaunchKernelGGL(flm::conv_rows_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), conv_rows_lds, stream, x,
                       params + L.conv_w, params + L.conv_b, M, T1, H3, x1);
    if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  } else if (x) {
    // conv as GEMM: rows (b,h), K = T1, 64 filters; stored as [b][c][h] with ReLU
    g.A = x; g.B = params + L.conv_w; g.C = x1; g.bias = params + L.conv_b; g.mask = nullptr;
    g.M = B * H3; g.N = 64; g.K = T1; g.lda = T1; g.ldb = T1; g.ldc = 0; g.a_zs = g.b_zs = g.c_zs = 0;
    g.relu = 1; g.accumulate = 0; g.H3 = H3;
    rc = flm::launch_gemm<true, true, flm::EPI_CONV>(g, 1, stream);
    if (rc) return rc;
  } else {
    const long M = (long)B * H3;
    const long obs_bytes = (long)B * 3 * plane_stride;
    if (M >= (1L << 31)) return FLM_E_ARG;
    const int nword = (T1 + 6) >> 2;
    const float* cw = params + L.conv_w;
    const float* cb = params + L.conv_b;
    const size_t cap = 160 * 1024 - 512;
    flm::ConvSecond sec{};
    const flm::ConvSecond* second = nullptr;
    if (params2) {
      if ((long)B2 * H3 >= (1L << 31)) return FLM_E_ARG;
      sec = flm::ConvSecond{params2 + L.conv_w, params2 + L.conv_b, obs2, (long)B2 * H3, (long)B2 * 3 * plane_stride, x1_2};
      second = &sec;
    }
    // on the bf16 matrix cores while the split filter bank fits the LDS beside the rows of the waves: sixteen waves
    // with 32 rows each first (four per SIMD hide one another's memory latency), then eight waves with 64 or 32
    // rows, then four with 32 (depth 300: the bank alone is 120 KB); beyond that on fp32 instructions
    if (nword <= 64 && flm::conv_u8_lds(T1, 16, true, 32) <= cap)
      rc = flm::launch_conv_u8<1, 16, true, 1>(obs_u8, cw, cb, M, H3 / 3, T1, H3, plane_stride, obs_bytes, x1, stream, second);
    else if (nword <= 64 && flm::conv_u8_lds(T1, 8, true) <= cap)
      rc = flm::launch_conv_u8<1, 8, true>(obs_u8, cw, cb, M, H3 / 3, T1, H3, plane_stride, obs_bytes, x1, stream, second);
    else if (nword <= 64 && flm::conv_u8_lds(T1, 8, true, 32) <= cap)       // (the LDS of four waves with 64 rows)
      rc = flm::launch_conv_u8<1, 8, true, 1>(obs_u8, cw, cb, M, H3 / 3, T1, H3, plane_stride, obs_bytes, x1, stream, second);
    else if (nword <= 128 && flm::conv_u8_lds(T1, 4, true, 32) <= cap)      // long horizons: the bank leaves room for four
      rc = flm::launch_conv_u8<2, 4, true, 1>(obs_u8, cw, cb, M, H3 / 3, T1, H3, plane_stride, obs_bytes, x1, stream, second);
    else if (nword <= 64)
      rc = flm::launch_conv_u8<1, 4, false>(obs_u8, cw, cb, M, H3 / 3, T1, H3, plane_stride, obs_bytes, x1, stream, second);
    else if (nword <= 128)
      rc = flm::launch_conv_u8<2, 4, false>(obs_u8, cw, cb, M, H3 / 3, T1, H3, plane_stride, obs_bytes, x1, stream, second);
    else
      rc = FLM_E_ARG;
    if (rc) return rc;
  }
  return FLM_OK;
}

// The dense layers behind the convolution: hid[B,1024] and q[B,NA] from x1[B,64*H3]; part is workspace for the K
// slices of the first dense layer (fc1_k_slices_cap(B) * B * 1024 floats).
static int dqn_dense_stage(const float* params, int B, int H3, int T1, int NA, const float* x1, float* hid, float* part,
                           float* q, hipStream_t stream) {
  DqnLayout L(H3, T1, NA);
  GemmArgs g{};
  int rc;
  const bool rows_path = dqn_rows_path(params, L, B, x1);
  if (rows_path) {
    if (B == 1)
      hipLaunchKernelGGL(flm::fc1_rows_kernel<1>, dim3(256), dim3(1024), 0, stream, x1, params + L.a1_w, params + L.v1_w,
                         params + L.a1_b, params + L.v1_b, B, L.K1, hid);
    else if (B <= 4)
      hipLaunchKernelGGL(flm::fc1_rows_kernel<4>, dim3(256), dim3(1024), 0, stream, x1, params + L.a1_w, params + L.v1_w,
                         params + L.a1_b, params + L.v1_b, B, L.K1, hid);
    else
      hipLaunchKernelGGL(flm::fc1_rows_kernel<8>, dim3(256), dim3(1024), 0, stream, x1, params + L.a1_w, params + L.v1_w,
                         params + L.a1_b, params + L.v1_b, B, L.K1, hid);
    if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  } else if (B <= 8 && (L.K1 & 3) == 0) {
    hipLaunchKernelGGL(flm::fc1_small_kernel<8>, dim3(128), dim3(256), 0, stream, x1, params + L.a1_w, params + L.a1_b, B,
                       L.K1, hid, 1024, 512);
    hipLaunchKernelGGL(flm::fc1_small_kernel<8>, dim3(128), dim3(256), 0, stream, x1, params + L.v1_w, params + L.v1_b, B,
                       L.K1, hid + 512, 1024, 512);
    if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  } else {
    // K = 64*H3 is long and M x N small: split K over workgroups, then reduce with bias + ReLU.  Both streams
    // in one launch: group 0 = advantage, group 1 = value (their weight blocks are a fixed stride apart).
    const int nks = fc1_k_slices(B, L.K1);
    const int ks = (int)(((L.K1 + nks - 1) / nks + XBK - 1) / XBK * XBK);
    g = GemmArgs{};
    g.A = x1; g.B = params + L.a1_w; g.C = part;
    g.M = B; g.N = 512; g.K = (int)L.K1; g.lda = L.K1; g.ldb = L.K1; g.ldc = 1024;
    g.a_zs = 0; g.b_zs = L.v1_w - L.a1_w; g.c_zs = (long)B * 1024; g.c_gs = 512; g.groups = 2; g.ksplit = ks;
    rc = flm::launch_gemm<true, true, flm::EPI_PLAIN>(g, 2 * nks, stream);
    if (rc) return rc;
    // the reduction over the K slices with bias and ReLU, and the output layers behind it, in one launch where the
    // buffers allow 16-byte accesses
    if ((((uintptr_t)part | (uintptr_t)hid | (uintptr_t)(params + L.a1_b) | (uintptr_t)(params + L.v1_b)) & 15) == 0) {
      hipLaunchKernelGGL(flm::fc1_reduce_head_kernel, dim3(B), dim3(256), 0, stream, part, nks, B, params + L.a1_b,
                         params + L.v1_b, hid, params + L.a2_w, params + L.a2_b, params + L.v2_w, params + L.v2_b, NA, q);
      if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
      return FLM_OK;
    }
    const long ne = (long)B * 1024;
    hipLaunchKernelGGL(flm::reduce_bias_relu_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream, part, nks, B,
                       1024, params + L.a1_b, params + L.v1_b, 512, hid, 1024L);
  }
  hipLaunchKernelGGL(flm::head_forward_kernel, dim3(B), dim3(64), 0, stream, hid, params + L.a2_w, params + L.a2_b,
                     params + L.v2_w, params + L.v2_b, B, NA, q);
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

static int dqn_forward_impl(const float* params, const float* x, const uint8_t* obs_u8, int plane_stride, int B, int H3,
                            int T1, int NA, float* x1, float* hid, float* part, float* q, hipStream_t stream) {
  const int rc = dqn_conv_stage(params, x, obs_u8, plane_stride, B, H3, T1, NA, x1, stream);
  return rc ? rc : dqn_dense_stage(params, B, H3, T1, NA, x1, hid, part, q, stream);
}

// rails actually used by the observations, rounded up to whole groups of four; 0 = the padded rows are too few to be
// worth a pass of their own, the batch is one of the small ones that stream the weights, or the indices do not fit
static int dqn_rails_used(int B, int Rp, int rails_used) {
  if (rails_used <= 0 || B <= 8) return 0;
  const int R4 = (rails_used + 3) & ~3;
  if (R4 + 4 > Rp || 192L * R4 >= 65536) return 0;
  return R4;
}

// The forward pass from uint8 observations whose rails [R4, Rp) are empty in every plane: the convolution runs over
// the R4 used rails of each plane only (x1c [B,64,3*R4]), the first dense layer multiplies those columns of its weights
// (segments of R4 in rows of 3*Rp per filter) and the empty rails enter through the bias (fc1_padding_bias_kernel).
// bias2: 1024 floats of workspace.
// The dense layers behind a convolution over the used rails: x1c [B,64,3*R4] -> hid, q.  bias2: 1024 floats of
// workspace; wpad_out: 1024*64 floats or nullptr (fc1_padding_bias_kernel).
static flm::PaddingBiasJob padding_bias_job(const float* params, const DqnLayout& L, float* bias2, float* wpad_out,
                                            float* rcb_out) {
  return flm::PaddingBiasJob{params + L.a1_w, params + L.v1_w, params + L.a1_b, params + L.v1_b, params + L.conv_b,
                             bias2, wpad_out, rcb_out};
}

static int dqn_dense_stage_rails(const float* params, int B, int Rp, int R4, int T1, int NA, const float* x1c, float* hid,
                                 float* part, float* bias2, float* wpad_out, float* q, hipStream_t stream,
                                 float* rcb_out = nullptr, bool bias2_ready = false) {
  const int H3 = 3 * Rp;
  DqnLayout L(H3, T1, NA);
  int rc;
  if (!bias2_ready) {
    const flm::PaddingBiasJob j = padding_bias_job(params, L, bias2, wpad_out, rcb_out);
    hipLaunchKernelGGL(flm::fc1_padding_bias_kernel, dim3(1024), dim3(256), 0, stream, j, j, L.K1, H3, Rp, R4);
  }
  const long Kc = 192L * R4;
  const int nks = fc1_k_slices(B, Kc);
  const int ks = (int)(((Kc + nks - 1) / nks + XBK - 1) / XBK * XBK);
  GemmArgs g{};
  g.A = x1c; g.B = params + L.a1_w; g.C = part;
  g.M = B; g.N = 512; g.K = (int)Kc; g.lda = Kc; g.ldb = L.K1; g.ldc = 1024;
  g.a_zs = 0; g.b_zs = L.v1_w - L.a1_w; g.c_zs = (long)B * 1024; g.c_gs = 512; g.groups = 2; g.ksplit = ks;
  g.b_seg_len = R4; g.b_seg_stride = Rp; g.b_seg_magic = (unsigned)((0x100000000ULL + (unsigned)R4 - 1) / (unsigned)R4);
  if ((rc = flm::launch_gemm<true, true, flm::EPI_PLAIN, 1>(g, 2 * nks, stream))) return rc;
  if ((((uintptr_t)part | (uintptr_t)hid | (uintptr_t)bias2) & 15) == 0) {
    hipLaunchKernelGGL(flm::fc1_reduce_head_kernel, dim3(B), dim3(256), 0, stream, part, nks, B, bias2, bias2 + 512, hid,
                       params + L.a2_w, params + L.a2_b, params + L.v2_w, params + L.v2_b, NA, q);
  } else {
    const long ne = (long)B * 1024;
    hipLaunchKernelGGL(flm::reduce_bias_relu_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, stream, part, nks, B,
                       1024, bias2, bias2 + 512, 512, hid, 1024L);
    hipLaunchKernelGGL(flm::head_forward_kernel, dim3(B), dim3(64), 0, stream, hid, params + L.a2_w, params + L.a2_b,
                       params + L.v2_w, params + L.v2_b, B, NA, q);
  }
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

static int dqn_forward_rails(const float* params, const uint8_t* obs_u8, int plane_stride, int B, int Rp, int R4, int T1,
                             int NA, float* x1c, float* hid, float* part, float* bias2, float* q, hipStream_t stream) {
  // the byte convolution addresses a row as (sample, plane, rail) with plane_stride bytes between planes: with R4 rails
  // per plane instead of Rp it reads the used rails and writes the compact activations
  const int rc = dqn_conv_stage(params, nullptr, obs_u8, plane_stride, B, 3 * R4, T1, NA, x1c, stream);
  return rc ? rc : dqn_dense_stage_rails(params, B, Rp, R4, T1, NA, x1c, hid, part, bias2, nullptr, q, stream);
}

extern "C" int flm_dqn_forward(const float* params, const float* x, int B, int H3, int T1, int NA, float* workspace,
                               float* q, void* stream) {
  if (!params || !x || !workspace || !q || B <= 0 || H3 <= 0 || T1 <= 0 || !actions_ok(NA)) return FLM_E_ARG;
  float* x1 = workspace;
  float* hid = workspace + (long)B * 64 * H3;
  float* part = hid + (long)B * 1024;
  return dqn_forward_impl(params, x, nullptr, 0, B, H3, T1, NA, x1, hid, part, q, (hipStream_t)stream);
}

extern "C" int flm_dqn_forward_u8(const float* params, const uint8_t* obs, int n, int Rp, int T1, int plane_stride, int NA,
                                  float* workspace, float* q, void* stream) {
  if (!params || !obs || !workspace || !q || n <= 0 || Rp <= 0 || T1 <= 0 || (long)plane_stride < (long)Rp * T1 ||
      !actions_ok(NA))
    return FLM_E_ARG;
  const int H3 = 3 * Rp;
  float* x1 = workspace;
  float* hid = workspace + (long)n * 64 * H3;
  float* part = hid + (long)n * 1024;
  return dqn_forward_impl(params, nullptr, obs, plane_stride, n, H3, T1, NA, x1, hid, part, q, (hipStream_t)stream);
}

extern "C" int flm_dqn_forward_u8_rails(const float* params, const uint8_t* obs, int n, int Rp, int T1, int plane_stride,
                                        int NA, int rails_used, float* workspace, float* q, void* stream) {
  if (!params || !obs || !workspace || !q || n <= 0 || Rp <= 0 || T1 <= 0 || (long)plane_stride < (long)Rp * T1 ||
      !actions_ok(NA) || rails_used < 0 || rails_used > Rp)
    return FLM_E_ARG;
  const int R4 = dqn_rails_used(n, Rp, rails_used);
  if (!R4) return flm_dqn_forward_u8(params, obs, n, Rp, T1, plane_stride, NA, workspace, q, stream);
  const int H3 = 3 * Rp;
  float* x1c = workspace;                                   // (the layout of flm_dqn_forward_u8: the compact x1 is smaller)
  float* hid = workspace + (long)n * 64 * H3;
  float* part = hid + (long)n * 1024;
  float* bias2 = part + (long)fc1_k_slices_cap(n) * n * 1024;
  return dqn_forward_rails(params, obs, plane_stride, n, Rp, R4, T1, NA, x1c, hid, part, bias2, q, (hipStream_t)stream);
}

extern "C" int flm_obs_to_f32(const uint8_t* obs, int n, int Rp, int T1, int plane_stride, float* out, void* stream) {
  if (!obs || !out || n <= 0 || Rp <= 0 || T1 <= 0 || (long)plane_stride < (long)Rp * T1) return FLM_E_ARG;
  if ((long)n * 3 > 65535 * 64L) return FLM_E_ARG;
  for (long s0 = 0; s0 < (long)n * 3; s0 += 65535)      // the grid's y extent is 65,535
    hipLaunchKernelGGL(flm::obs_to_f32_kernel, dim3(flm::obs_to_f32_pieces(Rp, T1), (unsigned)std::min(65535L, (long)n * 3 - s0)),
                       dim3(256), 0, (hipStream_t)stream, obs + s0 * plane_stride, n, Rp, T1, plane_stride,
                       out + s0 * (long)Rp * T1);
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

// split-K partials of a learn step: the local network's two forward passes run as one batch of 2B
static long learn_part_floats(int B) {
  const long one = (long)fc1_k_slices_cap(B) * B * 1024, two = (long)fc1_k_slices_cap(2 * B) * 2 * B * 1024;
  return one > two ? one : two;
}

// the convolution's weight-gradient partials: one [64,T1] per sample (from float states) or one [64,32*tiles] per
// workgroup (from bytes)
static long learn_cpart_floats(int B, int T1) {
  const long per_sample = (long)B * 64 * T1;
  const long per_group = (long)flm::conv_wgrad_u8_groups(B) * 64 * 32 * flm::conv_wgrad_u8_tiles(T1);
  return per_sample > per_group ? per_sample : per_group;
}

// workspace floats needed by flm_dqn_learn_grads
extern "C" long flm_dqn_learn_workspace(int B, int H3, int T1, int NA) {
  if (!actions_ok(NA)) return 0;
  const long act = (long)B * 64 * H3 + (long)B * 1024;        // x1 + hid of one forward
  return 3 * act                                              // three forwards
         + (long)B * 1024                                     // dhid
         + (long)B * 64 * H3                                  // dx1
         + learn_cpart_floats(B, T1)                          // conv weight-gradient partials
         + learn_part_floats(B)                               // split-K partials of the first dense layer
         + (((5L * NA + 2) * B + 64 + 3) & ~3L)               // three q's, dq, dadv; dval, squared errors (whole 16 bytes)
         + 2 * 1024 + 1024 * 64 + 1024 + 64;                  // used-rails form: two bias vectors, wpad, column sums of dhid, relu(conv bias)
}

// Forward passes, loss and backward of one learn step (src/agent.py:94-114); grads gets dLoss/dparams of
// the local network in the flat layout, loss_out the scalar mean-squared error.  Inputs are either float
// states (s, s_next) or uint8 observations (s_u8, sn_u8 with plane_stride): then the three forward passes read
// the bytes and s points at workspace that receives the float copy of the states the convolution's weight
// gradient needs.
// The step is written as two halves over one workspace.  The first: forward passes, loss, and the gradients of the
// output layers and of the first dense layer (everything in grads from flm_dqn_grad_split() on: 98 % of the buffer,
// complete and final).  The second: the gradient w.r.t. the convolution's activations and the convolution's own
// gradients (the floats before the split).  A data-parallel caller sends the first part on its way between the two.
struct LearnWorkspace {
  float *x1, *hid, *x1n, *hidn, *x1t, *hidt, *dhid, *dx1, *cpart, *part, *q, *qn, *qt, *dq, *dadv, *dval, *sq;
  float *bias2, *wpad, *gsum, *rcb;   // used-rails form
  LearnWorkspace(float* ws, int B, int H3, int T1, int NA) {
    // x1 / x1n, hid / hidn and q / qn are adjacent: the local network sees [s; s'] as one batch of 2B
    const long n_x1 = (long)B * 64 * H3, n_hid = (long)B * 1024;
    x1 = ws;             x1n = x1 + n_x1;
    hid = x1n + n_x1;    hidn = hid + n_hid;
    x1t = hidn + n_hid;  hidt = x1t + n_x1;
    dhid = hidt + n_hid;
    dx1 = dhid + n_hid;
    cpart = dx1 + n_x1;
    part = cpart + learn_cpart_floats(B, T1);
    q = part + learn_part_floats(B);
    qn = q + (long)NA * B; qt = qn + (long)NA * B; dq = qt + (long)NA * B;
    dadv = dq + (long)NA * B; dval = dadv + (long)NA * B; sq = dval + B;
    bias2 = q + (((5L * NA + 2) * B + 64 + 3) & ~3L);
    wpad = bias2 + 2 * 1024;
    gsum = wpad + 1024 * 64;
    rcb = gsum + 1024;
  }
};

static int learn_dense_half(const float* local, const float* target, const float* s, const float* s_next,
                            const uint8_t* s_u8, const uint8_t* sn_u8, int plane_stride, const int64_t* action,
                            const float* reward, const float* done, float gamma, int double_dqn, int B, int H3, int T1,
                            int NA, const LearnWorkspace& w, float* grads, float* loss_out, hipStream_t stream, int R4 = 0,
                            int fill_padding = 1) {
  DqnLayout L(H3, T1, NA);
  int rc;
  // R4 > 0 (uint8 observations only): rails [R4, Rp) are empty in every plane of every observation; the convolutions and
  // the products of the first dense layer run over the used rails (activations [.,64,3*R4]), the empty ones enter in
  // closed form (fc1_padding_bias_kernel, fc1_wgrad_padding_kernel, conv_wgrad_reduce_kernel's bias share)
  const int Rp = H3 / 3;
  const int H3c = R4 ? 3 * R4 : H3;                 // rows per sample the convolution computes
  const long n_x1c = (long)B * 64 * H3c;
  if (s_u8 && !flm::conv_wgrad_u8_tiles(T1)) {      // (the byte form of the weight gradient needs no float copy)
    if ((long)B * 3 > 65535) return FLM_E_ARG;
    hipLaunchKernelGGL(flm::obs_to_f32_kernel, dim3(flm::obs_to_f32_pieces(H3 / 3, T1), B * 3), dim3(256), 0, stream, s_u8, B,
                       H3 / 3, T1, plane_stride, const_cast<float*>(s));
    if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  }
  // The local network on the states and, under double DQN (where its view of the next states picks the action), on
  // the next states as well: two convolutions into adjacent halves, then the dense layers once over 2B samples (the
  // weights of the first dense layer are read once, and its K slices are half as many).
  // (One launch when the next states lie right behind the states in memory, as a replay memory that gathers both into
  // one buffer leaves them: the same rows, fewer idle waves at the end of the launch.)
  const bool adjacent = double_dqn && (s_u8 ? sn_u8 == s_u8 + (size_t)B * 3 * plane_stride : s_next == s + (size_t)B * H3 * T1) &&
                        (long)2 * B * H3 < (1L << 31) && (s_u8 || B > 8);   // (float input: the same kernel either way)
  bool target_conv_done = false;
  if (adjacent && s_u8) {
    // ... and the target network's convolution over the next states rides in the same launch
    if ((rc = dqn_conv_stage(local, nullptr, s_u8, plane_stride, 2 * B, H3c, T1, NA, w.x1, stream, target, sn_u8, B, w.x1t)))
      return rc;
    target_conv_done = true;
  } else if (adjacent) {
    if ((rc = dqn_conv_stage(local, s, nullptr, plane_stride, 2 * B, H3, T1, NA, w.x1, stream))) return rc;
  } else {
    if ((rc = dqn_conv_stage(local, s_u8 ? nullptr : s, s_u8, plane_stride, B, H3c, T1, NA, w.x1, stream))) return rc;
    if (double_dqn && (rc = dqn_conv_stage(local, s_u8 ? nullptr : s_next, sn_u8, plane_stride, B, H3c, T1, NA, w.x1 + n_x1c,
                                           stream)))
      return rc;
  }
  const int n_local = double_dqn ? 2 * B : B;
  if (R4) {
    // the padding's share of the bias for both networks: one launch
    hipLaunchKernelGGL(flm::fc1_padding_bias_kernel, dim3(2048), dim3(256), 0, stream,
                       padding_bias_job(local, L, w.bias2, w.wpad, w.rcb),
                       padding_bias_job(target, L, w.bias2 + 1024, nullptr, nullptr), L.K1, H3, Rp, R4);
    if ((rc = dqn_dense_stage_rails(local, n_local, Rp, R4, T1, NA, w.x1, w.hid, w.part, w.bias2, w.wpad, w.q, stream, w.rcb,
                                    true)))
      return rc;
    if (!target_conv_done && (rc = dqn_conv_stage(target, nullptr, sn_u8, plane_stride, B, H3c, T1, NA, w.x1t, stream))) return rc;
    rc = dqn_dense_stage_rails(target, B, Rp, R4, T1, NA, w.x1t, w.hidt, w.part, w.bias2 + 1024, nullptr, w.qt, stream, nullptr,
                               true);
  } else {
    if ((rc = dqn_dense_stage(local, n_local, H3, T1, NA, w.x1, w.hid, w.part, w.q, stream))) return rc;
    if (target_conv_done)
      rc = dqn_dense_stage(target, B, H3, T1, NA, w.x1t, w.hidt, w.part, w.qt, stream);
    else
      rc = dqn_forward_impl(target, s_u8 ? nullptr : s_next, sn_u8, plane_stride, B, H3, T1, NA, w.x1t, w.hidt, w.part, w.qt,
                            stream);
  }
  if (rc) return rc;
  if (double_dqn && B <= 1024) {
    // loss, head and the first dense layer's bias gradient (the column sums of dhid) in one launch
    // (a copy of the sums in the workspace for the used-rails form: a data-parallel caller may be reducing grads by the
    // time the second half reads them)
    hipLaunchKernelGGL(flm::head_backward_fused_kernel, dim3(16 + 8 * (NA + 1)), dim3(1024), 0, stream, w.q, w.qn, w.qt, action,
                       reward, done, gamma, B, NA, w.hid, local + L.a2_w, local + L.v2_w, w.dq, w.sq, loss_out, w.dhid, w.dadv,
                       w.dval, grads + L.a2_w, grads + L.a2_b, grads + L.v2_w, grads + L.v2_b, grads + L.a1_b, grads + L.v1_b,
                       R4 ? w.gsum : (float*)nullptr);
  } else {
    if (double_dqn) {
      hipLaunchKernelGGL(flm::td_loss_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, w.q, w.qn, w.qt, action, reward, done,
                         gamma, B, NA, w.dq, w.sq, (float*)nullptr);
    } else {
      hipLaunchKernelGGL(flm::td_loss_plain_kernel, dim3(1), dim3(1024), 0, stream, w.q, w.qt, action, reward, done, gamma, B,
                         NA, w.dq, w.sq);
    }
    hipLaunchKernelGGL(flm::mean_kernel, dim3(1), dim3(64), 0, stream, w.sq, B, loss_out);
    // head
    hipLaunchKernelGGL(flm::head_backward_hidden_kernel, dim3(B), dim3(64), 0, stream, w.hid, w.dq, local + L.a2_w,
                       local + L.v2_w, B, NA, w.dhid, w.dadv, w.dval);
    hipLaunchKernelGGL(flm::head_backward_params_kernel, dim3(8, NA + 1), dim3(1024), 0, stream, w.hid, w.dadv, w.dval, B, NA,
                       grads + L.a2_w, grads + L.a2_b, grads + L.v2_w, grads + L.v2_b);
    // the first dense layer's bias gradients = column sums of dhid
    hipLaunchKernelGGL(flm::colsum_kernel, dim3(16), dim3(1024), 0, stream, w.dhid, B, 1024, 1024L, grads + L.a1_b, 512,
                       grads + L.v1_b, R4 ? w.gsum : (float*)nullptr);
  }
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  // first dense layer: weight gradients dW[n,j] = sum_b dhid[b,n] * x1[b,j], both streams as two groups of one launch
  // (more workgroups per launch: fewer idle matrix pipes in the last round)
  GemmArgs g{};
  g.A = w.dhid; g.B = w.x1; g.C = grads + L.a1_w;
  g.M = 512; g.N = (int)L.K1; g.K = B; g.lda = 1024; g.ldb = L.K1; g.ldc = L.K1;
  g.ksplit = B; g.groups = 2; g.a_zs = 512; g.b_zs = 0; g.c_zs = 0; g.c_gs = L.v1_w - L.a1_w; g.x_slowest = 1;
  if (R4) {
    // over the used rails: the columns of the product are scattered into the rows of dW, the rest of dW is filled from
    // the column sums
    const long Kc = 192L * R4;
    g.N = (int)Kc; g.ldb = Kc;
    g.b_seg_len = R4; g.b_seg_stride = Rp; g.b_seg_magic = (unsigned)((0x100000000ULL + (unsigned)R4 - 1) / (unsigned)R4);
    if (fill_padding == 2) {
      // packed: the product as it is, [512][Kc] at the start of each weight block of grads (the optimizer pass in its
      // compact form reads it there and forms the gradient under the empty rows itself)
      g.ldc = Kc;
      g.b_seg_len = 0;
      if ((rc = flm::launch_gemm<false, false, flm::EPI_PLAIN>(g, 2, stream))) return rc;
    } else {
      if ((rc = flm::launch_gemm<false, false, flm::EPI_PLAIN, 3>(g, 2, stream))) return rc;
      if (fill_padding)     // (otherwise the optimizer pass forms these products itself: flm_dqn_adam_soft_update_rails)
        hipLaunchKernelGGL(flm::fc1_wgrad_padding_kernel, dim3(1024), dim3(256), 0, stream, w.gsum, local + L.conv_b,
                           grads + L.a1_w, grads + L.v1_w, L.K1, H3, Rp, R4);
    }
  } else {
    if ((rc = flm::launch_gemm<false, false, flm::EPI_PLAIN>(g, 2, stream))) return rc;
  }
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

static int learn_conv_half(const float* local, const float* s, const uint8_t* s_u8, int plane_stride, int B, int H3, int T1,
                           int NA, const LearnWorkspace& w, float* grads, hipStream_t stream, int R4 = 0) {
  DqnLayout L(H3, T1, NA);
  int rc;
  const int Rp = H3 / 3;
  // gradient w.r.t. the conv activations: dx1 = [dhid_adv dhid_val] * [W_adv; W_val], zero where x1 == 0: one product
  // over K = 1024 (the two weight blocks are not adjacent: the value block starts v1_w - a1_w - 512*K1 floats late)
  {
    GemmArgs g{};
    g.A = w.dhid; g.B = local + L.a1_w; g.C = w.dx1; g.mask = w.x1;
    g.M = B; g.N = (int)L.K1; g.K = 1024; g.lda = 1024; g.ldb = L.K1; g.ldc = L.K1;
    g.b_break_k = 512; g.b_break_off = (L.v1_w - L.a1_w) - 512L * L.K1; g.x_slowest = 1;
    if (R4) {            // over the used rails: the columns of the product are those of the compact activations
      const long Kc = 192L * R4;
      g.N = (int)Kc; g.ldc = Kc;
      g.b_seg_len = R4; g.b_seg_stride = Rp; g.b_seg_magic = (unsigned)((0x100000000ULL + (unsigned)R4 - 1) / (unsigned)R4);
      // 4 x 30 tiles (B = 512, 20 rails) would leave half the CUs idle: K is cut into slices whose partial sums (in the
      // forward pass's K-slice buffer) are added up, and masked, by a second launch.  As few slices as give every CU
      // about one workgroup: measured at 120 tiles, product + second launch take 56 / 36.6 / 39.7 / 52.6 us with
      // 1 / 2 / 4 / 8 slices.
      const long n_out = (long)B * Kc;
      const long fit = learn_part_floats(B) / n_out;
      const long tiles = (long)((B + GBM - 1) / GBM) * ((Kc + GBN - 1) / GBN);
      int nks = 1;
      while (nks < 8 && tiles * nks < 192 && 2 * nks <= fit) nks *= 2;
      if (nks >= 2 && (((uintptr_t)w.part | (uintptr_t)w.x1 | (uintptr_t)w.dx1) & 15) == 0) {
        g.C = w.part; g.mask = nullptr; g.ksplit = 1024 / nks; g.c_zs = n_out;
        if ((rc = flm::launch_gemm<true, false, flm::EPI_PLAIN, 2>(g, nks, stream))) return rc;
        hipLaunchKernelGGL(flm::reduce_mask_kernel, dim3((unsigned)((n_out / 4 + 255) / 256)), dim3(256), 0, stream, w.part, nks,
                           n_out, w.x1, w.dx1);
      } else {
        if ((rc = flm::launch_gemm<true, false, flm::EPI_MASK, 2>(g, 1, stream))) return rc;
      }
    } else {
      if ((rc = flm::launch_gemm<true, false, flm::EPI_MASK>(g, 1, stream))) return rc;
    }
  }
  // conv weights and bias from the observation bytes where the horizon allows: one launch and its reduction
  const int nt = s_u8 ? flm::conv_wgrad_u8_tiles(T1) : 0;
  if (nt) {
    float* dW = grads + L.conv_w;
    float* db = grads + L.conv_b;
    const int rails = R4 ? R4 : Rp;          // rails per plane in dx1 (the kernel steps through the planes by plane_stride)
    // (over the used rails the bias gradient's share of the empty rows rides in the reduction)
    const float* gs = R4 ? w.gsum : nullptr;
    const float* cb = local + L.conv_b;
    switch (nt) {
      case 1: rc = flm::launch_conv_wgrad_u8<1>(w.dx1, s_u8, B, rails, T1, plane_stride, w.cpart, dW, db, stream, gs, w.wpad, cb); break;
      case 2: rc = flm::launch_conv_wgrad_u8<2>(w.dx1, s_u8, B, rails, T1, plane_stride, w.cpart, dW, db, stream, gs, w.wpad, cb); break;
      case 3: rc = flm::launch_conv_wgrad_u8<3>(w.dx1, s_u8, B, rails, T1, plane_stride, w.cpart, dW, db, stream, gs, w.wpad, cb); break;
      case 4: rc = flm::launch_conv_wgrad_u8<4>(w.dx1, s_u8, B, rails, T1, plane_stride, w.cpart, dW, db, stream, gs, w.wpad, cb); break;
      default: rc = flm::launch_conv_wgrad_u8<5>(w.dx1, s_u8, B, rails, T1, plane_stride, w.cpart, dW, db, stream, gs, w.wpad, cb); break;
    }
    return rc;
  }
  // from float states: per sample dWc_b[64,T1] = dx1_b[64,H3] * s_b[H3,T1], then a fixed-order sum over the batch
  {
    GemmArgs g{};
    g.A = w.dx1; g.B = s; g.C = w.cpart;
    g.M = 64; g.N = T1; g.K = H3; g.lda = H3; g.ldb = T1; g.ldc = T1;
    g.a_zs = 64L * H3; g.b_zs = (long)H3 * T1; g.c_zs = 64L * T1;
    if ((rc = flm::launch_gemm<true, false, flm::EPI_PLAIN>(g, B, stream))) return rc;
    const long ne = 64L * T1;
    hipLaunchKernelGGL(flm::reduce_partials_kernel, dim3((unsigned)((ne + 63) / 64)), dim3(1024), 0, stream, w.cpart, B, ne,
                       grads + L.conv_w);
    // bias: four interleaved parts of the batch side by side (cpart is free again: stream order), then their sum
    hipLaunchKernelGGL(flm::conv_bias_grad_kernel, dim3(64, 4), dim3(1024), 0, stream, w.dx1, B, H3, w.cpart);
    hipLaunchKernelGGL(flm::reduce_partials_kernel, dim3(1), dim3(1024), 0, stream, w.cpart, 4, 64L, grads + L.conv_b);
  }
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

// phase 0: the whole step; 1: the first half; 2: the second half (same arguments, workspace untouched in between)
static int learn_grads_impl(const float* local, const float* target, const float* s, const float* s_next,
                            const uint8_t* s_u8, const uint8_t* sn_u8, int plane_stride, const int64_t* action,
                            const float* reward, const float* done, float gamma, int double_dqn, int B, int H3, int T1,
                            int NA, float* ws, float* grads, float* loss_out, hipStream_t stream, int phase = 0, int R4 = 0,
                            int fill_padding = 1) {
  const LearnWorkspace w(ws, B, H3, T1, NA);
  int rc = FLM_OK;
  if (phase != 2)
    rc = learn_dense_half(local, target, s, s_next, s_u8, sn_u8, plane_stride, action, reward, done, gamma, double_dqn, B, H3,
                          T1, NA, w, grads, loss_out, stream, R4, fill_padding);
  if (rc == FLM_OK && phase != 1) rc = learn_conv_half(local, s, s_u8, plane_stride, B, H3, T1, NA, w, grads, stream, R4);
  return rc;
}

extern "C" int flm_dqn_learn_grads(const float* local, const float* target, const float* s, const float* s_next,
                                   const int64_t* action, const float* reward, const float* done, float gamma,
                                   int double_dqn, int B, int H3, int T1, int NA, float* ws, float* grads,
                                   float* loss_out, void* stream) {
  if (!local || !target || !s || !s_next || !action || !reward || !done || !ws || !grads || !loss_out) return FLM_E_ARG;
  if (B <= 0 || H3 <= 0 || T1 <= 0 || !actions_ok(NA)) return FLM_E_ARG;
  return learn_grads_impl(local, target, s, s_next, nullptr, nullptr, 0, action, reward, done, gamma, double_dqn, B, H3, T1,
                          NA, ws, grads, loss_out, (hipStream_t)stream);
}

// first float of grads that phase 1 of a split learn step completes (the convolution's gradients lie before it)
extern "C" long flm_dqn_grad_split(int H3, int T1, int NA) { return actions_ok(NA) ? DqnLayout(H3, T1, NA).a1_w : 0; }

extern "C" int flm_dqn_learn_grads_phase(const float* local, const float* target, const float* s, const float* s_next,
                                         const int64_t* action, const float* reward, const float* done, float gamma,
                                         int double_dqn, int B, int H3, int T1, int NA, float* ws, float* grads,
                                         float* loss_out, int phase, void* stream) {
  if (!local || !target || !s || !s_next || !action || !reward || !done || !ws || !grads || !loss_out) return FLM_E_ARG;
  if (B <= 0 || H3 <= 0 || T1 <= 0 || !actions_ok(NA) || phase < 1 || phase > 2) return FLM_E_ARG;
  return learn_grads_impl(local, target, s, s_next, nullptr, nullptr, 0, action, reward, done, gamma, double_dqn, B, H3, T1,
                          NA, ws, grads, loss_out, (hipStream_t)stream, phase);
}

extern "C" long flm_dqn_learn_workspace_u8(int B, int Rp, int T1, int NA) {
  if (!actions_ok(NA)) return 0;
  return flm_dqn_learn_workspace(B, 3 * Rp, T1, NA) + (long)B * 3 * Rp * T1;
}

extern "C" int flm_dqn_learn_grads_u8(const float* local, const float* target, const uint8_t* s, const uint8_t* s_next,
                                      int plane_stride, const int64_t* action, const float* reward, const float* done,
                                      float gamma, int double_dqn, int B, int Rp, int T1, int NA, float* ws,
                                      float* grads, float* loss_out, void* stream) {
  if (!local || !target || !s || !s_next || !action || !reward || !done || !ws || !grads || !loss_out) return FLM_E_ARG;
  if (B <= 0 || Rp <= 0 || T1 <= 0 || (long)plane_stride < (long)Rp * T1 || !actions_ok(NA)) return FLM_E_ARG;
  const int H3 = 3 * Rp;
  float* s_f32 = ws + flm_dqn_learn_workspace(B, H3, T1, NA);   // the float copy of s lives behind the common workspace
  return learn_grads_impl(local, target, s_f32, nullptr, s, s_next, plane_stride, action, reward, done, gamma, double_dqn, B,
                          H3, T1, NA, ws, grads, loss_out, (hipStream_t)stream);
}

extern "C" int flm_dqn_learn_grads_u8_phase(const float* local, const float* target, const uint8_t* s, const uint8_t* s_next,
                                            int plane_stride, const int64_t* action, const float* reward,
                                            const float* done, float gamma, int double_dqn, int B, int Rp, int T1, int NA,
                                            float* ws, float* grads, float* loss_out, int phase, void* stream) {
  if (!local || !target || !s || !s_next || !action || !reward || !done || !ws || !grads || !loss_out) return FLM_E_ARG;
  if (B <= 0 || Rp <= 0 || T1 <= 0 || (long)plane_stride < (long)Rp * T1 || !actions_ok(NA) || phase < 1 || phase > 2)
    return FLM_E_ARG;
  const int H3 = 3 * Rp;
  float* s_f32 = ws + flm_dqn_learn_workspace(B, H3, T1, NA);
  return learn_grads_impl(local, target, s_f32, nullptr, s, s_next, plane_stride, action, reward, done, gamma, double_dqn, B,
                          H3, T1, NA, ws, grads, loss_out, (hipStream_t)stream, phase);
}

// rails per plane that flm_dqn_learn_grads_u8_rails computes over for these sizes (a multiple of four), 0 = it runs the
// plain form: what a caller may leave out of the observations it hands over
extern "C" long flm_dqn_learn_rails(int B, int Rp, int T1, int rails_used) {
  if (B <= 0 || Rp <= 0 || T1 <= 0 || rails_used < 0 || rails_used > Rp) return 0;
  return flm::conv_wgrad_u8_tiles(T1) ? dqn_rails_used(B, Rp, rails_used) : 0;
}

extern "C" int flm_dqn_learn_grads_u8_rails(const float* local, const float* target, const uint8_t* s, const uint8_t* s_next,
                                            int plane_stride, const int64_t* action, const float* reward, const float* done,
                                            float gamma, int double_dqn, int B, int Rp, int T1, int NA, int rails_used,
                                            float* ws, float* grads, float* loss_out, int fill_padding, int phase,
                                            void* stream) {
  if (!local || !target || !s || !s_next || !action || !reward || !done || !ws || !grads || !loss_out) return FLM_E_ARG;
  if (B <= 0 || Rp <= 0 || T1 <= 0 || (long)plane_stride < (long)Rp * T1 || !actions_ok(NA) || phase < 0 || phase > 2 ||
      rails_used < 0 || rails_used > Rp || fill_padding < 0 || fill_padding > 2)
    return FLM_E_ARG;
  const int H3 = 3 * Rp;
  // the used-rails form needs the byte form of the convolution's weight gradient (no float copy of the states exists)
  const int R4 = (int)flm_dqn_learn_rails(B, Rp, T1, rails_used);
  if (fill_padding != 1 && !R4) return FLM_E_ARG;
  float* s_f32 = ws + flm_dqn_learn_workspace(B, H3, T1, NA);
  return learn_grads_impl(local, target, s_f32, nullptr, s, s_next, plane_stride, action, reward, done, gamma, double_dqn, B,
                          H3, T1, NA, ws, grads, loss_out, (hipStream_t)stream, phase, R4, fill_padding);
}

extern "C" int flm_dqn_adam(float* params, const float* grads, float* m, float* v, long n, double lr, double beta1,
                            double beta2, double eps, int step, void* stream) {
  if (!params || !grads || !m || !v || n <= 0 || step <= 0) return FLM_E_ARG;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(flm::adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, m,
                     v, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1),
                     (float)sqrt(bc2));
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

extern "C" int flm_dqn_adam_soft_update(float* params, const float* grads, float* m, float* v, float* target, long n,
                                        double lr, double beta1, double beta2, double eps, int step, double tau,
                                        void* stream) {
  if (!params || !grads || !m || !v || !target || n <= 0 || step <= 0) return FLM_E_ARG;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(flm::adam_soft_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     params, grads, m, v, target, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)(lr / bc1), (float)sqrt(bc2), (float)tau, (float)(1.0 - tau));
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

// float offset, inside the workspace of flm_dqn_learn_grads_u8_rails, of the 64 values relu(conv bias) the step left
// there (what flm_dqn_adam_soft_update_rails needs of it)
extern "C" long flm_dqn_learn_rails_rcb_offset(int B, int Rp, int T1, int NA) {
  if (B <= 0 || Rp <= 0 || T1 <= 0 || !actions_ok(NA)) return -1;
  float* base = nullptr;
  const LearnWorkspace w(base, B, 3 * Rp, T1, NA);
  return (long)(w.rcb - base);
}

extern "C" int flm_dqn_adam_soft_update_rails(float* params, const float* grads, float* m, float* v, float* target, long n,
                                              double lr, double beta1, double beta2, double eps, int step, double tau, int Rp,
                                              int T1, int NA, int rails4, const float* rcb, const float* moments_in,
                                              float* moments_out, float* m_packed, float* v_packed, void* stream) {
  if (!params || !grads || !m || !v || !target || !rcb || n <= 0 || step <= 0 || Rp <= 0 || T1 <= 0 || !actions_ok(NA))
    return FLM_E_ARG;
  const DqnLayout L(3 * Rp, T1, NA);
  if (n != L.total || rails4 <= 0 || rails4 >= Rp || (rails4 & 3) || 512L * L.K1 >= (1L << 24)) return FLM_E_ARG;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)m | (uintptr_t)v | (uintptr_t)target) & 15) return FLM_E_ARG;
  const bool compact = moments_in != nullptr;
  if (compact != (moments_out != nullptr) || compact != (m_packed != nullptr) || compact != (v_packed != nullptr) ||
      (compact && moments_in == moments_out) || (((uintptr_t)moments_in | (uintptr_t)moments_out) & 7))
    return FLM_E_ARG;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const dim3 grid((unsigned)((n + 1023) / 1024));
#define FLM_ADAM_RAILS(COMPACT)                                                                                              \
  hipLaunchKernelGGL(flm::adam_soft_update_rails_kernel<COMPACT>, grid, dim3(256), 0, (hipStream_t)stream, params, grads, m, v,  \
                     target, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1),       \
                     (float)sqrt(bc2), (float)tau, (float)(1.0 - tau), L.a1_w, L.a1_b, L.v1_w, L.v1_b, (unsigned)L.K1,         \
                     (unsigned)(3 * Rp), (unsigned)Rp, (unsigned)rails4, 1.0f / (float)L.K1, 1.0f / (float)(3 * Rp),           \
                     1.0f / (float)Rp, rcb, reinterpret_cast<const float2*>(moments_in), reinterpret_cast<float2*>(moments_out), \
                     m_packed, v_packed)
  if (compact) FLM_ADAM_RAILS(true);
  else FLM_ADAM_RAILS(false);
#undef FLM_ADAM_RAILS
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

extern "C" int flm_dqn_adam_moments(float* m, float* v, long n, int Rp, int T1, int NA, int rails4, float* moments,
                                    float* m_packed, float* v_packed, int fill, void* stream) {
  if (!m || !v || !moments || !m_packed || !v_packed || n <= 0 || Rp <= 0 || T1 <= 0 || !actions_ok(NA) ||
      ((uintptr_t)moments & 7))
    return FLM_E_ARG;
  const DqnLayout L(3 * Rp, T1, NA);
  if (n != L.total || rails4 <= 0 || rails4 >= Rp) return FLM_E_ARG;
  if (fill)
    hipLaunchKernelGGL(flm::adam_moments_kernel<true>, dim3(1024), dim3(256), 0, (hipStream_t)stream, m, v, L.a1_w, L.v1_w, L.K1,
                       3 * Rp, Rp, rails4, reinterpret_cast<float2*>(moments), m_packed, v_packed);
  else
    hipLaunchKernelGGL(flm::adam_moments_kernel<false>, dim3(1024), dim3(256), 0, (hipStream_t)stream, m, v, L.a1_w, L.v1_w,
                       L.K1, 3 * Rp, Rp, rails4, reinterpret_cast<float2*>(moments), m_packed, v_packed);
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}

extern "C" int flm_dqn_soft_update(float* target, const float* local, long n, double tau, void* stream) {
  if (!target || !local || n <= 0) return FLM_E_ARG;
  hipLaunchKernelGGL(flm::soft_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, target,
                     local, n, (float)tau, (float)(1.0 - tau));
  if (hipGetLastError() != hipSuccess) return FLM_E_LAUNCH;
  return FLM_OK;
}
