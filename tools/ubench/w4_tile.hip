// Microbenchmark: the "one wave per SIMD" GEMM geometry (256 threads = 4 waves as 2x2, wave tile 128x128 of a 256x256x64 block tile,
// 256 accumulator registers per lane) -- can hipcc hold it without spills and what do D (LDS-DMA), R (ds_read_b128) and
// M (MFMA) cost per K-tile?   Per K-tile and workgroup: D 64 KiB (16 x 1 KiB per wave), R 128 KiB (32 x ds_read_b128 per wave),
// M 512 x 16x16x32 (128 per wave) or 256 x 32x32x16 (64 per wave).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/w4_tile.hip -o tools/ubench/w4_tile.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// MODE bits: 1 = D, 2 = R, 4 = M, 8 = deferred DMA wait (vmcnt(16)), 16 = sched_group_barrier interleave
template <int MODE, int SHAPE>
__global__ void __launch_bounds__(256, 1) k(const char* __restrict__ src, float* __restrict__ sink, int iters, size_t window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + ((size_t)blockIdx.x * 65536) % window;
  constexpr int NB = SHAPE == 16 ? 8 : 4;          // blocks per wave-tile side
  constexpr int KS = SHAPE == 16 ? 2 : 4;          // k-steps per K-tile (k32 / k16)
  using acc_t = typename std::conditional<SHAPE == 16, f32x4_t, f32x16_t>::type;
  acc_t acc[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < (SHAPE == 16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;
  bf16x8_t af[NB], bfr[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) { af[j] = __builtin_bit_cast(bf16x8_t, make_uint4(lane, j, 1, 2)); bfr[j] = __builtin_bit_cast(bf16x8_t, make_uint4(lane, j, 3, 4)); }
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if ((MODE & 1) && !(MODE & 16)) {
#pragma unroll
      for (int c = 0; c < 16; ++c)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + ((size_t)(it & 63) * 65536 + (wid * 16 + c) * 1024 + lane * 16) % window),
                                         (lptr_t)(smem + (buf ^ 1) * 65536 + (wid * 16 + c) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (MODE & 2) {
#pragma unroll
        for (int j = 0; j < NB; ++j) af[j] = *reinterpret_cast<const bf16x8_t*>(smem + buf * 65536 + ((ks * 2 * NB + j) * 1024 + lane * 16 + wid * 8192) % 65536);
#pragma unroll
        for (int j = 0; j < NB; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(smem + buf * 65536 + ((ks * 2 * NB + NB + j) * 1024 + lane * 16 + wid * 8192) % 65536);
      }
      if (MODE & 4) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          if ((MODE & 16) && (MODE & 1)) {       // 16 DMAs spread over the KS*NB row groups of the K-tile
            constexpr int per = 16 / (KS * NB);  // 1 for both shapes
#pragma unroll
            for (int e = 0; e < per; ++e) {
              const int c = (ks * NB + i) * per + e;
              __builtin_amdgcn_global_load_lds((gptr_t)(base + ((size_t)(it & 63) * 65536 + (wid * 16 + c) * 1024 + lane * 16) % window),
                                               (lptr_t)(smem + (buf ^ 1) * 65536 + (wid * 16 + c) * 1024), 16, 0, 0);
            }
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            if constexpr (SHAPE == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) { asm volatile("" ::"v"(af[i])); asm volatile("" ::"v"(bfr[i])); }
      }
    }
    if (MODE & 8) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < (SHAPE == 16 ? 4 : 16); ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int SHAPE>
float run(const char* src, float* sink, int iters, size_t window) {
  hipFuncSetAttribute((const void*)k<MODE, SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, SHAPE>), dim3(256), dim3(256), 131072, 0, src, sink, 8, window);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, SHAPE>), dim3(256), dim3(256), 131072, 0, src, sink, iters, window);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

template <int SHAPE>
void sweep(const char* src, float* sink, size_t w) {
  const int iters = 2000;
  printf(" MFMA %s: D %.3f  R %.3f  M %.3f  D+R %.3f  D+M %.3f  R+M %.3f  D+R+M %.3f | D+R+M deferred %.3f  interleaved %.3f  both %.3f | D+M both %.3f  (us per K-tile)\n",
         SHAPE == 16 ? "16x16x32" : "32x32x16", run<1, SHAPE>(src, sink, iters, w), run<2, SHAPE>(src, sink, iters, w), run<4, SHAPE>(src, sink, iters, w),
         run<3, SHAPE>(src, sink, iters, w), run<5, SHAPE>(src, sink, iters, w), run<6, SHAPE>(src, sink, iters, w), run<7, SHAPE>(src, sink, iters, w),
         run<7 + 8, SHAPE>(src, sink, iters, w), run<7 + 16, SHAPE>(src, sink, iters, w), run<7 + 24, SHAPE>(src, sink, iters, w),
         run<5 + 24, SHAPE>(src, sink, iters, w));
}

int main() {
  const size_t window = 64ull << 20;
  char* src; float* sink;
  hipMalloc(&src, window + (1 << 20)); hipMemset(src, 1, window + (1 << 20)); hipMalloc(&sink, 256 * 256 * 4);
  for (int pass = 0; pass < 2; ++pass) {
    const size_t w = pass == 0 ? (2ull << 20) : window;
    printf("window %zu MiB\n", w >> 20);
    sweep<16>(src, sink, w);
    sweep<32>(src, sink, w);
  }
  return 0;
}
