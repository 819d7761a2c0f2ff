// Write-only patterns over an observation-sized buffer: how the order in which workgroups cover
// [E][A][3][plane_stride] bytes changes the achieved HBM write rate.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <bool NT>
__device__ __forceinline__ void st16(unsigned char* p, uint4 v) {
  typedef unsigned vec4u __attribute__((ext_vector_type(4)));
  if (NT) {
    vec4u t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vec4u*>(p));
  } else {
    *reinterpret_cast<uint4*>(p) = v;
  }
}

// A: workgroup = (1 KiB tile, env); wave w writes agents w, w+16, ... (three planes each)
template <bool NT>
__global__ __launch_bounds__(1024) void pat_a(unsigned char* obs, int A, size_t ps) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, e = blockIdx.y;
  const size_t off = (size_t)blockIdx.x * 1024 + l * 16;
  if (off >= ps) return;
  const uint4 v = make_uint4(e, w, l, 1);
  for (int a = w; a < A; a += 16) {
    unsigned char* o = obs + ((size_t)e * A + a) * 3 * ps + off;
    st16<NT>(o, v);
    st16<NT>(o + ps, v);
    st16<NT>(o + 2 * ps, v);
  }
}

// B: workgroup = (16 KiB of positions, env); all 16 waves write the same plane of the same agent side by side
template <bool NT>
__global__ __launch_bounds__(1024) void pat_b(unsigned char* obs, int A, size_t ps) {
  const int e = blockIdx.y;
  const size_t off = (size_t)blockIdx.x * 16384 + threadIdx.x * 16;
  if (off >= ps) return;
  const uint4 v = make_uint4(e, threadIdx.x, 2, 1);
  for (int a = 0; a < A; a++) {
    unsigned char* o = obs + ((size_t)e * A + a) * 3 * ps + off;
    st16<NT>(o, v);
    st16<NT>(o + ps, v);
    st16<NT>(o + 2 * ps, v);
  }
}

// C: workgroup = (env, agent): the three planes of one agent front to back (265 KB contiguous)
template <bool NT>
__global__ __launch_bounds__(1024) void pat_c(unsigned char* obs, int A, size_t ps) {
  unsigned char* o = obs + ((size_t)blockIdx.y * A + blockIdx.x) * 3 * ps;
  const uint4 v = make_uint4(blockIdx.x, threadIdx.x, 3, 1);
  for (size_t off = (size_t)threadIdx.x * 16; off < 3 * ps; off += 16384) st16<NT>(o + off, v);
}

// D: like B with 256-thread workgroups covering 4 KiB
template <bool NT>
__global__ __launch_bounds__(256) void pat_d(unsigned char* obs, int A, size_t ps) {
  const int e = blockIdx.y;
  const size_t off = (size_t)blockIdx.x * 4096 + threadIdx.x * 16;
  if (off >= ps) return;
  const uint4 v = make_uint4(e, threadIdx.x, 4, 1);
  for (int a = 0; a < A; a++) {
    unsigned char* o = obs + ((size_t)e * A + a) * 3 * ps + off;
    st16<NT>(o, v);
    st16<NT>(o + ps, v);
    st16<NT>(o + 2 * ps, v);
  }
}

// M: like A with 4 KiB tiles (wave w writes agents w, w+16, ..., four 1 KiB rows per plane).  With a plane
// stride that is a multiple of 32 KiB every chunk a workgroup writes has the same 4 KiB residue mod 8, and that
// residue is blockIdx.x mod 8 (+ ROT): the XCD the workgroup runs on when workgroups go round-robin to the XCDs.
template <bool NT, int ROT>
__global__ __launch_bounds__(1024) void pat_m(unsigned char* obs, int A, size_t ps, size_t ps_used) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, e = blockIdx.y;
  const size_t tile = (blockIdx.x & ~(size_t)7) | ((blockIdx.x + ROT) & 7);
  const size_t off = tile * 4096 + l * 16;
  const uint4 v = make_uint4(e, w, l, 1);
  for (int a = w; a < A; a += 16) {
    unsigned char* o = obs + ((size_t)e * A + a) * 3 * ps + off;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int c = 0; c < 4; c++)
        if (off + c * 1024 < ps_used) st16<NT>(o + p * ps + c * 1024, v);
  }
}

// E: linear order, one short-lived workgroup per contiguous chunk of CHUNK bytes
template <bool NT, int THREADS, int CHUNK>
__global__ __launch_bounds__(THREADS) void pat_e(unsigned char* obs, size_t n) {
  const size_t base = (size_t)blockIdx.x * CHUNK;
  const uint4 v = make_uint4(blockIdx.x, threadIdx.x, 5, 1);
#pragma unroll
  for (int off = threadIdx.x * 16; off < CHUNK; off += THREADS * 16)
    if (base + off < n) st16<NT>(obs + base + off, v);
}

// F: 4 KiB chunks in linear order with the chunk index remapped: MODE 0 = identity, 1 = neighbours swapped,
// 2 = rotated by the group number (every workgroup slot of a group of 8 visits every residue over time),
// 3 = low 6 bits reversed
template <int MODE>
__global__ __launch_bounds__(256) void pat_f(unsigned char* obs, size_t n) {
  size_t b = blockIdx.x;
  if (MODE == 1) b ^= 1;
  if (MODE == 2) b = (b & ~(size_t)7) | ((b + (b >> 3)) & 7);
  if (MODE == 3) {
    unsigned lo = b & 63, r = 0;
    for (int i = 0; i < 6; i++) r |= ((lo >> i) & 1) << (5 - i);
    b = (b & ~(size_t)63) | r;
  }
  const size_t off = b * 4096 + threadIdx.x * 16;
  if (off < n) st16<false>(obs + off, make_uint4(b, threadIdx.x, 6, 1));
}

// G: long-lived workgroups of 256 threads; workgroup w writes ITER chunks of 4 KiB that all have the residue
// (w mod 8) but lie far apart (stride = 8 * gridDim chunks)
template <int ITER>
__global__ __launch_bounds__(256) void pat_g(unsigned char* obs, size_t n) {
  const size_t nb = gridDim.x;
  for (int i = 0; i < ITER; i++) {
    const size_t chunk = (size_t)blockIdx.x + (size_t)i * nb;
    const size_t off = chunk * 4096 + threadIdx.x * 16;
    if (off < n) st16<false>(obs + off, make_uint4(i, threadIdx.x, 7, 1));
  }
}

// H: like G but the residue of the chunks a workgroup writes changes every iteration
template <int ITER>
__global__ __launch_bounds__(256) void pat_h(unsigned char* obs, size_t n) {
  const size_t nb = gridDim.x;
  for (int i = 0; i < ITER; i++) {
    const size_t chunk = (((size_t)blockIdx.x & ~(size_t)7) | ((blockIdx.x + i) & 7)) + (size_t)i * nb;
    const size_t off = chunk * 4096 + threadIdx.x * 16;
    if (off < n) st16<false>(obs + off, make_uint4(i, threadIdx.x, 8, 1));
  }
}

// I: linear 4 KiB chunks, one short-lived workgroup each, with the dependent work a fused streaming kernel
// would do before its single store: one 2-byte load per lane from a small table, an LDS tile built from it,
// a 16-byte load of a running total (L2 resident), then the store.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void pat_i(unsigned char* obs, size_t n, const short* cols, const uint4* totals) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[THREADS * 16];
  const size_t b = blockIdx.x;
  const int tid = threadIdx.x;
  reinterpret_cast<uint4*>(tile)[tid] = make_uint4(0, 0, 0, 0);
  int c = -1;
  if (tid < 151) c = cols[(b % 4096) * 151 + tid];
  __syncthreads();
  if (c >= 0 && c < THREADS * 16) tile[c] = 1;
  __syncthreads();
  const uint4 mine = reinterpret_cast<const uint4*>(tile)[tid];
  const uint4 t = totals[(b % 2048) * THREADS + tid];
  const size_t off = b * (THREADS * 16) + tid * 16;
  if (off < n) st16<false>(obs + off, make_uint4(t.x - mine.x, t.y - mine.y, t.z - mine.z, t.w - mine.w));
}

__device__ __forceinline__ uint4 bits4(uint4 v, unsigned m) {   // bytes equal to m -> 1
  auto f = [m](unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 4; i++) r |= ((((x >> (8 * i)) & 0xff) == m) ? 1u : 0u) << (8 * i);
    return r;
  };
  return make_uint4(f(v.x), f(v.y), f(v.z), f(v.w));
}

// J: the real streaming kernel.  GRID3: (chunks per plane, 3*A, E) else 1-D with the coordinates divided out;
// SMALLTOT: totals wrapped into a 32 MB window.
template <bool GRID3, bool SMALLTOT, int VAR = 0>
__global__ __launch_bounds__(256) void pat_j(const short* __restrict__ cols, const unsigned char* __restrict__ totals,
                                             int A, int T1, int ps, unsigned char* __restrict__ obs) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[4096];
  const int tid = threadIdx.x;
  int x, y, e;
  if (GRID3) { x = blockIdx.x; y = blockIdx.y; e = blockIdx.z; }
  else {
    const unsigned cpp = ps / 4096, b = blockIdx.x;
    const unsigned pl = b / cpp;
    x = b - pl * cpp;
    e = pl / (3 * A);
    y = pl - e * 3 * A;
  }
  reinterpret_cast<uint4*>(tile)[tid] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const int a = y / 3, p = y - a * 3;
  const int off0 = x * 4096;
  const short* ac = cols + ((size_t)(VAR == 1 ? (e & 7) : e) * A + a) * T1;
  for (int t = tid; t < T1; t += 256) {
    const int c = ac[t];
    const int q = (c >> 1) * T1 + t - off0;
    if (VAR != 2 && c >= 0 && q >= 0 && q < 4096) tile[q] = (c & 1) ? (unsigned char)1 : (unsigned char)0xff;
  }
  __syncthreads();
  const uint4 mine = reinterpret_cast<const uint4*>(tile)[tid];
  uint4 v = mine;
  const int off = off0 + tid * 16;
  if (p && VAR != 3) {
    const size_t te = SMALLTOT ? (size_t)(e & 127) : (size_t)e;
    const uint4 tot = *reinterpret_cast<const uint4*>(totals + (te * 2 + (p == 1 ? 1 : 0)) * ps + off);
    const uint4 m = bits4(mine, p == 1 ? 1u : 0xffu);
    v = make_uint4(tot.x - m.x, tot.y - m.y, tot.z - m.z, tot.w - m.w);
  }
  *reinterpret_cast<uint4*>(obs + (((size_t)e * A + a) * 3 + p) * ps + off) = v;
}

// K: the streaming kernel with both global loads issued before the first barrier; THREADS x PER 16-byte pieces
// per 4 KiB chunk (256 x 1 or 128 x 2), 1-D or 3-D grid.
template <int THREADS, bool GRID3, bool SKIP = false>
__global__ __launch_bounds__(THREADS) void pat_k(const short* __restrict__ cols, const unsigned char* __restrict__ totals,
                                                 int A, int T1, int ps, unsigned char* __restrict__ obs) {
  constexpr int PER = 256 / THREADS;
  __shared__ __attribute__((aligned(16))) unsigned char tile[4096];
  const int tid = threadIdx.x;
  int x, y, e;
  if (GRID3) { x = blockIdx.x; y = blockIdx.y; e = blockIdx.z; }
  else {
    const unsigned cpp = ps / 4096, b = blockIdx.x;
    const unsigned pl = b / cpp;
    x = b - pl * cpp;
    e = pl / (3 * A);
    y = pl - e * 3 * A;
  }
  const int a = y / 3, p = y - a * 3;
  const int off0 = x * 4096;
  const short* ac = cols + ((size_t)e * A + a) * T1;
  constexpr int KT = THREADS >= 256 ? 1 : THREADS == 128 ? 2 : 3;
  int c[KT];
#pragma unroll
  for (int k = 0; k < KT; k++) {
    c[k] = -1;
    if (tid + k * THREADS < T1) c[k] = ac[tid + k * THREADS];
  }
  uint4 tot[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    tot[k] = make_uint4(0, 0, 0, 0);
    if (p) tot[k] = *reinterpret_cast<const uint4*>(totals + ((size_t)e * 2 + (p == 1 ? 1 : 0)) * ps + off0 + (tid + k * THREADS) * 16);
  }
  int q[KT];
  bool any = false;
#pragma unroll
  for (int k = 0; k < KT; k++) {
    q[k] = (c[k] >> 1) * T1 + tid + k * THREADS - off0;
    if (c[k] < 0 || q[k] < 0 || q[k] >= 4096) q[k] = -1;
    any |= q[k] >= 0;
  }
  const bool use_tile = !SKIP || __syncthreads_or(any);
  if (use_tile) {
#pragma unroll
    for (int k = 0; k < PER; k++) reinterpret_cast<uint4*>(tile)[tid + k * THREADS] = make_uint4(0, 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KT; k++)
      if (q[k] >= 0) tile[q[k]] = (c[k] & 1) ? (unsigned char)1 : (unsigned char)0xff;
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const uint4 mine = use_tile ? reinterpret_cast<const uint4*>(tile)[tid + k * THREADS] : make_uint4(0, 0, 0, 0);
    uint4 v = mine;
    if (p) {
      const unsigned sh = p == 1 ? 0 : 1;
      auto f = [sh](unsigned xx) { return sh ? (xx >> 1) & 0x01010101u : xx & ~(xx >> 1) & 0x01010101u; };
      v = make_uint4(tot[k].x - f(mine.x), tot[k].y - f(mine.y), tot[k].z - f(mine.z), tot[k].w - f(mine.w));
    }
    *reinterpret_cast<uint4*>(obs + (((size_t)e * A + a) * 3 + p) * ps + off0 + (tid + k * THREADS) * 16) = v;
  }
}

// L: lean version of the streaming kernel: 256 threads, one column entry per thread (T1 <= 256), no
// data-dependent branches (a miss goes to a spare LDS byte), both loads before the first barrier.
template <bool GRID3, int VAR = 0>
__global__ __launch_bounds__(256) void pat_l(const short* __restrict__ cols, const unsigned char* __restrict__ totals,
                                             int A, int T1, int ps, unsigned char* __restrict__ obs) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[4096 + 256];
  const int tid = threadIdx.x;
  int x, y, e;
  if (GRID3) { x = blockIdx.x; y = blockIdx.y; e = blockIdx.z; }
  else {
    const unsigned cpp = ps / 4096, b = blockIdx.x;
    const unsigned pl = b / cpp;
    x = b - pl * cpp;
    e = pl / (3 * A);
    y = pl - e * 3 * A;
  }
  const int a = y / 3, p = y - a * 3;
  const int off0 = x * 4096;
  const unsigned agent = (unsigned)e * A + a;
  const unsigned agent_c = VAR & 1 ? (unsigned)(e & 7) * A + a : agent;
  const int c = cols[(size_t)agent_c * T1 + min(tid, T1 - 1)];
  uint4 tot = make_uint4(0, 0, 0, 0);
  if (p) tot = *reinterpret_cast<const uint4*>(totals + ((size_t)(VAR & 2 ? (e & 7) : e) * 2 + (p == 1 ? 1 : 0)) * ps + off0 + tid * 16);
  reinterpret_cast<uint4*>(tile)[tid] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const unsigned q = (unsigned)((c >> 1) * T1 + tid - off0);
  const bool hit = c >= 0 && tid < T1 && q < 4096u;
  tile[hit ? q : 4096u + tid] = (c & 1) ? (unsigned char)1 : (unsigned char)0xff;
  __syncthreads();
  const uint4 mine = reinterpret_cast<const uint4*>(tile)[tid];
  uint4 v = mine;
  if (p) {
    const unsigned sh = p == 1 ? 0 : 1;
    auto f = [sh](unsigned xx) { return sh ? (xx >> 1) & 0x01010101u : xx & ~(xx >> 1) & 0x01010101u; };
    v = make_uint4(tot.x - f(mine.x), tot.y - f(mine.y), tot.z - f(mine.z), tot.w - f(mine.w));
  }
  *reinterpret_cast<uint4*>(obs + ((size_t)agent * 3 + p) * ps + off0 + tid * 16) = v;
}

// P: persistent linear sweep.  Every wave works through 4 KiB chunks of its XCD's residue class in address
// order; the inputs of chunk i + D (the agent's column entries, the dense totals) are fetched while chunk i is
// written, so a read that misses the L2 never stalls a store.  No workgroup barrier.
template <int D, int VAR>
__global__ __launch_bounds__(256) void pat_p(const short* __restrict__ cols, const unsigned char* __restrict__ totals,
                                             int A, int T1, int ps, unsigned char* __restrict__ obs, unsigned n_chunks) {
  __shared__ __attribute__((aligned(16))) unsigned char tiles[4][4096 + 256];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
  unsigned char* tile = tiles[wave];
  const unsigned r = blockIdx.x & 7, q = blockIdx.x >> 3, S = (gridDim.x >> 3) * nwv, s = q * nwv + wave;
  const unsigned cpp = ps / 4096;
  const unsigned iters = (n_chunks / 8 + S - 1) / S;
  int cc[D][3];
  uint4 tot[D][4];
  auto fetch = [&](int k, unsigned i) {
    unsigned J = r + 8 * (s + i * S);
    if (J >= n_chunks) J = n_chunks - 1;
    const unsigned P = J / cpp;
    const unsigned x = J - P * cpp;
    const unsigned Q = P / 3;
    const unsigned p = P - Q * 3;
    const short* ac = cols + (size_t)Q * T1;
#pragma unroll
    for (int j = 0; j < 3; j++) cc[k][j] = ac[min(lane + 64 * j, T1 - 1)];
    const unsigned e = Q / (unsigned)A;
    const unsigned char* tp = totals + ((size_t)(VAR & 2 ? (e & 7) : e) * 2 + (p == 1 ? 1 : 0)) * ps + x * 4096 + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; j++) tot[k][j] = *reinterpret_cast<const uint4*>(tp + j * 1024);
  };
#pragma unroll
  for (int k = 0; k < D; k++) fetch(k, k);
  for (unsigned i0 = 0; i0 < iters; i0 += D) {
#pragma unroll
    for (int k = 0; k < D; k++) {
      const unsigned i = i0 + k;
      const unsigned J = r + 8 * (s + i * S);
      const unsigned P = J / cpp;
      const unsigned x = J - P * cpp;
      const unsigned p = P % 3;
      const int off0 = x * 4096;
      if (!(VAR & 4)) {
#pragma unroll
      for (int j = 0; j < 4; j++) reinterpret_cast<uint4*>(tile)[j * 64 + lane] = make_uint4(0, 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int t = lane + 64 * j, c = cc[k][j];
        const unsigned qq = (unsigned)((c >> 1) * T1 + t - off0);
        const bool hit = c >= 0 && t < T1 && qq < 4096u;
        tile[hit ? qq : 4096u + lane] = (c & 1) ? (unsigned char)1 : (unsigned char)0xff;
      }
      __builtin_amdgcn_wave_barrier();
      }
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint4 mine = make_uint4(cc[k][0], cc[k][1], cc[k][2], 0);
        if (!(VAR & 4)) mine = reinterpret_cast<const uint4*>(tile)[j * 64 + lane];
        v[j] = mine;
        if (p) {
          const unsigned sh = p == 1 ? 0 : 1;
          auto f = [sh](unsigned xx) { return sh ? (xx >> 1) & 0x01010101u : xx & ~(xx >> 1) & 0x01010101u; };
          v[j] = make_uint4(tot[k][j].x - f(mine.x), tot[k][j].y - f(mine.y), tot[k][j].z - f(mine.z), tot[k][j].w - f(mine.w));
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (J < n_chunks) {
        unsigned char* o = obs + (size_t)J * 4096 + lane * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<uint4*>(o + j * 1024) = v[j];
      }
      fetch(k, i + D);
    }
  }
}

#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(r)); return 1; } } while (0)

template <class F>
static int timeit(const char* name, F launch, double bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; i++) launch();
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= 5;
  printf("%-28s %7.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
  return 0;
}

int main(int argc, char** argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 4096, A = 50;
  printf("E = %d\n", E);
  const size_t ps = 88320;
  const size_t n = (size_t)E * A * 3 * ps;
  unsigned char* obs;
  CK(hipMalloc(&obs, n));
  const double bytes = (double)n;
  dim3 ga((ps + 1023) / 1024, E), gb((ps + 16383) / 16384, E), gc(A, E), gd((ps + 4095) / 4096, E);
  timeit("memset", [&] { (void)hipMemsetAsync(obs, 1, n, 0); }, bytes);
  timeit("A 1KiB tiles, NT", [&] { pat_a<true><<<ga, 1024>>>(obs, A, ps); }, bytes);
  timeit("A 1KiB tiles, plain", [&] { pat_a<false><<<ga, 1024>>>(obs, A, ps); }, bytes);
  timeit("B 16KiB bursts, NT", [&] { pat_b<true><<<gb, 1024>>>(obs, A, ps); }, bytes);
  timeit("B 16KiB bursts, plain", [&] { pat_b<false><<<gb, 1024>>>(obs, A, ps); }, bytes);
  timeit("C agent front to back, NT", [&] { pat_c<true><<<gc, 1024>>>(obs, A, ps); }, bytes);
  timeit("C agent front to back, plain", [&] { pat_c<false><<<gc, 1024>>>(obs, A, ps); }, bytes);
  timeit("D 4KiB bursts 256 thr, NT", [&] { pat_d<true><<<gd, 256>>>(obs, A, ps); }, bytes);
  timeit("D 4KiB bursts 256 thr, plain", [&] { pat_d<false><<<gd, 256>>>(obs, A, ps); }, bytes);
  timeit("E linear 4KiB/256thr, plain", [&] { pat_e<false, 256, 4096><<<(unsigned)((n + 4095) / 4096), 256>>>(obs, n); }, bytes);
  timeit("E linear 8KiB/256thr, plain", [&] { pat_e<false, 256, 8192><<<(unsigned)((n + 8191) / 8192), 256>>>(obs, n); }, bytes);
  timeit("E linear 8KiB/256thr, NT", [&] { pat_e<true, 256, 8192><<<(unsigned)((n + 8191) / 8192), 256>>>(obs, n); }, bytes);
  timeit("E linear 16KiB/1024thr, plain", [&] { pat_e<false, 1024, 16384><<<(unsigned)((n + 16383) / 16384), 1024>>>(obs, n); }, bytes);
  timeit("E linear 64KiB/1024thr, plain", [&] { pat_e<false, 1024, 65536><<<(unsigned)((n + 65535) / 65536), 1024>>>(obs, n); }, bytes);
  timeit("E linear 64KiB/256thr, plain", [&] { pat_e<false, 256, 65536><<<(unsigned)((n + 65535) / 65536), 256>>>(obs, n); }, bytes);
  const unsigned nch = (unsigned)((n + 4095) / 4096);
  timeit("F identity", [&] { pat_f<0><<<nch, 256>>>(obs, n); }, bytes);
  timeit("F neighbours swapped", [&] { pat_f<1><<<nch, 256>>>(obs, n); }, bytes);
  timeit("F rotated", [&] { pat_f<2><<<nch, 256>>>(obs, n); }, bytes);
  timeit("F 6 bits reversed", [&] { pat_f<3><<<nch, 256>>>(obs, n); }, bytes);
  timeit("G same residue, 150 iters", [&] { pat_g<150><<<(nch / 150 + 7) / 8 * 8, 256>>>(obs, n); }, bytes);
  timeit("H changing residue, 150 iters", [&] { pat_h<150><<<(nch / 150 + 7) / 8 * 8, 256>>>(obs, n); }, bytes);
  timeit("G same residue, 38 iters", [&] { pat_g<38><<<(nch / 38 + 7) / 8 * 8, 256>>>(obs, n); }, bytes);
  if (argc > 2 && argv[2][0] == 'a') {   // residue-aligned plane hopping: stride 98304 = 3 * 32 KiB, 88,184 bytes of each plane written
    CK(hipFree(obs));
    const size_t psa = 98304, used = 88192;
    const size_t na = (size_t)E * A * 3 * psa;
    CK(hipMalloc(&obs, na));
    printf("obs %p\n", (void*)obs);
    const double bu = (double)E * A * 3 * used;
    dim3 gm(psa / 4096, E), gm2((used + 4095) / 4096, E);
    timeit("M aligned, residue = xcd, NT", [&] { pat_m<true, 0><<<gm, 1024>>>(obs, A, psa, used); }, bu);
    timeit("M aligned, residue = xcd", [&] { pat_m<false, 0><<<gm, 1024>>>(obs, A, psa, used); }, bu);
    timeit("M aligned, residue = xcd+1", [&] { pat_m<false, 1><<<gm, 1024>>>(obs, A, psa, used); }, bu);
    timeit("M aligned, residue = xcd+2", [&] { pat_m<false, 2><<<gm, 1024>>>(obs, A, psa, used); }, bu);
    timeit("M aligned, residue = xcd+4", [&] { pat_m<false, 4><<<gm, 1024>>>(obs, A, psa, used); }, bu);
    timeit("M aligned, 22 tiles per env", [&] { pat_m<false, 0><<<gm2, 1024>>>(obs, A, psa, used); }, bu);
    timeit("M stride 88320 (unaligned)", [&] { pat_m<false, 0><<<gm2, 1024>>>(obs, A, 88320, used); }, bu);
    CK(hipFree(obs));
    return 0;
  }
  short* cols;
  uint4* totals;
  CK(hipMalloc(&cols, 4096 * 151 * 2));
  CK(hipMalloc(&totals, 2048 * 1024 * 16));
  CK(hipMemset(cols, 0x11, 4096 * 151 * 2));
  CK(hipMemset(totals, 0x22, 2048 * 1024 * 16));
  timeit("I fused-like, 4KiB/256thr", [&] { pat_i<256><<<nch, 256>>>(obs, n, cols, totals); }, bytes);
  timeit("I fused-like, 1KiB/64thr", [&] { pat_i<64><<<(unsigned)((n + 1023) / 1024), 64>>>(obs, n, cols, totals); }, bytes);
  {
    const int ps2 = 90112, T1 = 151;
    unsigned char *obs2, *tot2;
    short* cols2;
    const size_t n2 = (size_t)E * A * 3 * ps2;
    CK(hipMalloc(&obs2, n2));
    CK(hipMalloc(&tot2, (size_t)E * 2 * ps2));
    CK(hipMalloc(&cols2, (size_t)E * A * T1 * 2));
    CK(hipMemset(tot2, 0x33, (size_t)E * 2 * ps2));
    short* h = (short*)malloc((size_t)E * A * T1 * 2);
    unsigned r = 12345;
    for (size_t i = 0; i < (size_t)E * A * T1; i++) {
      if ((i % T1) % 12 == 0) r = r * 1103515245u + 12345u;      // a train stays on a rail for a dozen steps
      h[i] = (short)(((r >> 16) % 584) * 2 + ((r >> 8) & 1));
    }
    CK(hipMemcpy(cols2, h, (size_t)E * A * T1 * 2, hipMemcpyHostToDevice));
    free(h);
    const double by2 = (double)n2;
    dim3 g3(ps2 / 4096, 3 * A, E);
    const unsigned g1 = (unsigned)(n2 / 4096);
    timeit("J real, 3-D grid", [&] { pat_j<true, false><<<g3, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("J real, 3-D grid, small totals", [&] { pat_j<true, true><<<g3, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("J 3-D, cols wrapped small", [&] { pat_j<true, true, 1><<<g3, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("J 3-D, no LDS scatter", [&] { pat_j<true, true, 2><<<g3, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("J 3-D, no totals", [&] { pat_j<true, true, 3><<<g3, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K hoisted loads, 256thr, 3-D", [&] { pat_k<256, true><<<g3, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K hoisted loads, 256thr, 1-D", [&] { pat_k<256, false><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K hoisted loads, 128thr, 3-D", [&] { pat_k<128, true><<<g3, 128>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K hoisted loads, 128thr, 1-D", [&] { pat_k<128, false><<<g1, 128>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K hoisted loads, 64thr, 3-D", [&] { pat_k<64, true><<<g3, 64>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K hoisted loads, 64thr, 1-D", [&] { pat_k<64, false><<<g1, 64>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K 64thr, 3-D, skip empty", [&] { pat_k<64, true, true><<<g3, 64>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("K 128thr, 3-D, skip empty", [&] { pat_k<128, true, true><<<g3, 128>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    if (argc > 2 && argv[2][0] == 'p') {
      const unsigned nc = (unsigned)(n2 / 4096);
      timeit("P persistent D=2, 512 wg", [&] { pat_p<2, 0><<<512, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P persistent D=4, 512 wg", [&] { pat_p<4, 0><<<512, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P persistent D=4, 1024 wg", [&] { pat_p<4, 0><<<1024, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P persistent D=4, 2048 wg", [&] { pat_p<4, 0><<<2048, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P persistent D=8, 512 wg", [&] { pat_p<8, 0><<<512, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P persistent D=8, 1024 wg", [&] { pat_p<8, 0><<<1024, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P D=4, 1024 wg, totals small", [&] { pat_p<4, 2><<<1024, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P D=4, 1024 wg, no LDS", [&] { pat_p<4, 4><<<1024, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P D=4, 1024 wg, no LDS, totals small", [&] { pat_p<4, 6><<<1024, 256>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P D=4, 1024 wg x 64 thr", [&] { pat_p<4, 0><<<1024, 64>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P D=4, 4096 wg x 64 thr", [&] { pat_p<4, 0><<<4096, 64>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("P D=4, 8192 wg x 64 thr", [&] { pat_p<4, 0><<<8192, 64>>>(cols2, tot2, A, T1, ps2, obs2, nc); }, by2);
      timeit("L lean, 1-D", [&] { pat_l<false><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
      return 0;
    }
    timeit("L lean, 3-D", [&] { pat_l<true><<<g3, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("L lean, 1-D, cols small", [&] { pat_l<false, 1><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("L lean, 1-D, totals small", [&] { pat_l<false, 2><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("L lean, 1-D, both small", [&] { pat_l<false, 3><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("L lean, 1-D", [&] { pat_l<false><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("J real, 1-D grid", [&] { pat_j<false, false><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    timeit("J real, 1-D grid, small totals", [&] { pat_j<false, true><<<g1, 256>>>(cols2, tot2, A, T1, ps2, obs2); }, by2);
    CK(hipFree(obs2));
  }
  CK(hipFree(obs));
  return 0;
}
