// Microbenchmark behind NOTEBOOK.md "GEMM experiments": what do LDS-DMA (global_load_lds), ds_read_b128 and MFMA cost alone and
// together on one CU-resident workgroup per CU (512 threads, 128 KiB LDS, the GEMM's geometry)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_mfma.hip -o gpurun_out/ubench && gpurun_out/ubench
// Per "iteration" (= one GEMM K-tile) a workgroup does, depending on the mode bits:
//   D: 64 x global_load_lds_dwordx4 (64 KiB from an L2-resident 4 MiB window)      R: 192 x ds_read_b128 (192 KiB)
//   M: 512 x v_mfma_f32_16x16x32_bf16 (the K-tile's MFMAs, 64 per wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int SHAPE = 16>
__global__ void __launch_bounds__(512) k(const char* __restrict__ src, float* __restrict__ sink, int iters, size_t window) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src;   // window is a power of two: offsets are masked, not divided (a 64-bit '%' costs ~130 VALU instructions)
  f32x4_t acc[8][4];
  f32x16_t acc32[4][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  bf16x8_t wf[4], xf[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) wf[j] = __builtin_bit_cast(bf16x8_t, make_uint4(lane, j, 1, 2));
#pragma unroll
  for (int i = 0; i < 8; ++i) xf[i] = __builtin_bit_cast(bf16x8_t, make_uint4(lane, i, 3, 4));
  uint4 stg[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) stg[c] = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (MODE & 32) {  // register staging instead of LDS-DMA: 8 x global_load_dwordx4 now, 8 x ds_write_b128 at the end of the iteration
#pragma unroll
      for (int c = 0; c < 8; ++c)
        stg[c] = *reinterpret_cast<const uint4*>(base + (((size_t)(it & 63) * 65536 + (size_t)blockIdx.x * 65536 + (wid * 8 + c) * 1024 + lane * 16) & (window - 1)));
    }
    if ((MODE & 1) && !(MODE & 16)) {  // DMA: 8 per wave, 1 KiB each, into the other buffer
#pragma unroll
      for (int c = 0; c < 8; ++c)
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (((size_t)(it & 63) * 65536 + (size_t)blockIdx.x * 65536 + (wid * 8 + c) * 1024 + lane * 16) & (window - 1))),
                                         (lptr_t)(smem + (buf ^ 1) * 65536 + (wid * 8 + c) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (MODE & 2) {  // fragment reads (conflict-free pattern: consecutive 16-B per lane)
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(smem + buf * 65536 + ((ks * 12 + j) * 1024 + lane * 16 + wid * 4096) % 65536);
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(smem + buf * 65536 + ((ks * 12 + 4 + i) * 1024 + lane * 16 + wid * 4096) % 65536);
      }
      if (MODE & 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if ((MODE & 16) && (MODE & 1) && (i & 1) == 0) {   // interleaved issue: one DMA per 8 MFMAs
            const int c = ks * 4 + (i >> 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (((size_t)(it & 63) * 65536 + (size_t)blockIdx.x * 65536 + (wid * 8 + c) * 1024 + lane * 16) & (window - 1))),
                                             (lptr_t)(smem + (buf ^ 1) * 65536 + (wid * 8 + c) * 1024), 16, 0, 0);
          }
          if constexpr (SHAPE == 16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
          } else {   // same fragment bytes per k-step pair, half as many (twice as long) MFMAs: 32 per K-tile and wave
#pragma unroll
            for (int j = 0; j < 2; ++j) acc32[i & 3][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2 * j + (i >> 2)], xf[i], acc32[i & 3][j], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(xf[i]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(wf[j]));
      }
    }
    if (MODE & 32) {
#pragma unroll
      for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(smem + (buf ^ 1) * 65536 + (wid * 8 + c) * 1024 + lane * 16) = stg[c];
    }
    if (MODE & 64) {   // W: the K-tile's 64 KiB written with ds_write_b128 from registers (no global traffic): LDS write-port cost only
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        stg[c].x += it;
        *reinterpret_cast<uint4*>(smem + (buf ^ 1) * 65536 + (wid * 8 + c) * 1024 + lane * 16) = stg[c];
      }
    }
    if (MODE & 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // deferred: only the PREVIOUS iteration's DMA must have landed
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc32[i][j][r];
  if (s == 12345.678f) sink[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int SHAPE = 16>
float run(const char* src, float* sink, int iters, size_t window, int blocks) {
  hipFuncSetAttribute((const void*)k<MODE, SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, SHAPE>), dim3(blocks), dim3(512), 131072, 0, src, sink, 8, window);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<MODE, SHAPE>), dim3(blocks), dim3(512), 131072, 0, src, sink, iters, window);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  const size_t window = 64ull << 20;   // 64 MiB: L2-miss / MALL-hit stream;  use 2 MiB per-XCD-resident window for the L2-hit case
  char* src; float* sink;
  hipMalloc(&src, window + (1 << 20)); hipMemset(src, 1, window + (1 << 20)); hipMalloc(&sink, 256 * 512 * 4 * 4);
  const int iters = 2000, blocks = 256;
  const char* names[8] = {"-", "D", "R", "D+R", "M", "D+M", "R+M", "D+R+M"};
  for (int pass = 0; pass < 2; ++pass) {
    const size_t w = pass == 0 ? (2ull << 20) : window;
    printf("window %zu MiB (%s)\n", w >> 20, pass == 0 ? "L2-resident" : "beyond L2: MALL/HBM");
    float t[8];
    t[1] = run<1>(src, sink, iters, w, blocks); t[2] = run<2>(src, sink, iters, w, blocks); t[3] = run<3>(src, sink, iters, w, blocks);
    t[4] = run<4>(src, sink, iters, w, blocks); t[5] = run<5>(src, sink, iters, w, blocks); t[6] = run<6>(src, sink, iters, w, blocks);
    t[7] = run<7>(src, sink, iters, w, blocks);
    for (int m = 1; m < 8; ++m) printf("  %-6s %8.3f ms  = %6.3f us per K-tile-equivalent\n", names[m], t[m], t[m] * 1e3 / iters);
    printf("  D+M   deferred wait      %6.3f us | interleaved issue %6.3f us | both %6.3f us\n", run<5 + 8>(src, sink, iters, w, blocks) * 1e3 / iters,
           run<5 + 16>(src, sink, iters, w, blocks) * 1e3 / iters, run<5 + 24>(src, sink, iters, w, blocks) * 1e3 / iters);
    printf("  D+R+M deferred wait      %6.3f us | interleaved issue %6.3f us | both %6.3f us\n", run<7 + 8>(src, sink, iters, w, blocks) * 1e3 / iters,
           run<7 + 16>(src, sink, iters, w, blocks) * 1e3 / iters, run<7 + 24>(src, sink, iters, w, blocks) * 1e3 / iters);
    printf("  MFMA 32x32x16 instead of 16x16x32:  M %6.3f | R+M %6.3f | D+M both %6.3f | D+R+M %6.3f | D+R+M deferred %6.3f  interleaved %6.3f  both %6.3f us\n",
           run<4, 32>(src, sink, iters, w, blocks) * 1e3 / iters, run<6, 32>(src, sink, iters, w, blocks) * 1e3 / iters,
           run<5 + 24, 32>(src, sink, iters, w, blocks) * 1e3 / iters, run<7, 32>(src, sink, iters, w, blocks) * 1e3 / iters,
           run<7 + 8, 32>(src, sink, iters, w, blocks) * 1e3 / iters, run<7 + 16, 32>(src, sink, iters, w, blocks) * 1e3 / iters,
           run<7 + 24, 32>(src, sink, iters, w, blocks) * 1e3 / iters);
    printf("  ds_write_b128 of the 64 KiB from registers (no global loads):  W %6.3f | W+R %6.3f | W+M %6.3f | W+R+M %6.3f us\n",
           run<64>(src, sink, iters, w, blocks) * 1e3 / iters, run<64 + 2>(src, sink, iters, w, blocks) * 1e3 / iters,
           run<64 + 4>(src, sink, iters, w, blocks) * 1e3 / iters, run<64 + 6>(src, sink, iters, w, blocks) * 1e3 / iters);
    printf("  D     deferred wait      %6.3f us\n", run<1 + 8>(src, sink, iters, w, blocks) * 1e3 / iters);
    printf("  register staging (global_load_dwordx4 -> ds_write_b128):  G %6.3f us | G+R %6.3f us | G+M %6.3f us | G+R+M %6.3f us\n",
           run<32>(src, sink, iters, w, blocks) * 1e3 / iters, run<32 + 2>(src, sink, iters, w, blocks) * 1e3 / iters,
           run<32 + 4>(src, sink, iters, w, blocks) * 1e3 / iters, run<32 + 6>(src, sink, iters, w, blocks) * 1e3 / iters);
  }
  return 0;
}
