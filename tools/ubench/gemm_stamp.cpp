// Debug harness: where does the time go at the seams of the persistent GEMM tile loop?  Links nothing: dlopens a -DGRIT_GEMM_STAMP build of
// gemm_bf16.hip (tools/ubench/build_gemm_stamp.sh), runs one GEMM shape, prints per-tile cycle differences between the stamp points:
//   0 tile start | 1 after K-tile 0 | 2 after K-tile 1 | 3 after the steady loop | 4 after K-tile nk-2 | 5 after the last K-tile | 6 after the epilogue
// usage: gemm_stamp.bin M N K epi
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef int (*gemm_fn)(const void*, const void*, void*, int64_t, int, int, int64_t, int64_t, int64_t, int, const void*, int64_t, void*);
typedef int (*stamp_fn)(unsigned long long*);
__global__ void fill(uint16_t* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float f = ((x & 0xffffff) / 16777216.0f * 2.0f - 1.0f) * scale;
    uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); p[i] = (uint16_t)(u >> 16);
  }
}
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 131072; const int N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096, epi = argc > 4 ? atoi(argv[4]) : 0;
  void* h = dlopen("tools/ubench/_var/libgemm_stamp.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  gemm_fn gemm = (gemm_fn)dlsym(h, "grit_gemm_bf16_nt"); stamp_fn stamps = (stamp_fn)dlsym(h, "grit_debug_gemm_stamps");
  uint16_t *A, *W, *C, *R;
  const int outN = epi == 2 ? N / 2 : N;
  CK(hipMalloc(&A, M * K * 2)); CK(hipMalloc(&W, (int64_t)N * K * 2)); CK(hipMalloc(&C, M * outN * 2)); CK(hipMalloc(&R, M * N * 2));
  fill<<<2048, 256>>>(A, M * K, 1, 1.0f); fill<<<2048, 256>>>(W, (int64_t)N * K, 2, 0.05f); fill<<<2048, 256>>>(R, M * N, 3, 1.0f);
  for (int it = 0; it < 3; ++it) { int rc = gemm(A, W, C, M, N, K, K, K, outN, epi, epi == 1 ? R : nullptr, N, nullptr); if (rc) { fprintf(stderr, "rc %d\n", rc); return 3; } }
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> s(8 * 2 * 64 * 8);
  if (stamps(s.data())) { fprintf(stderr, "stamp copy failed\n"); return 4; }
  const char* names[7] = {"ktile0", "ktile1", "steady", "ktile nk-2", "last ktile", "epilogue", "-> next tile start"};
  for (int wg = 0; wg < 8; wg += 3) for (int g = 0; g < 2; ++g) {
    printf("workgroup slot %d wave group %d (cycles)\n  tile:", wg, g);
    for (int t = 1; t < 6; ++t) printf(" %9d", t);
    printf("\n");
    for (int p = 0; p < 7; ++p) {
      printf("  %-20s", names[p]);
      for (int t = 1; t < 6; ++t) {
        const unsigned long long* a = &s[((wg * 2 + g) * 64 + t) * 8];
        const unsigned long long b = p < 6 ? a[p + 1] : s[((wg * 2 + g) * 64 + t + 1) * 8];
        printf(" %9lld", (long long)(b - a[p < 6 ? p : 6]));
      }
      printf("\n");
    }
    const unsigned long long* a = &s[((wg * 2 + g) * 64 + 2) * 8];
    printf("  steady K-tiles: %d per tile -> %.1f cycles per K-tile; whole tile %lld cycles\n", K / 64 - 4, (double)(a[3] - a[2]) / (K / 64 - 4),
           (long long)(s[((wg * 2 + g) * 64 + 3) * 8] - a[0]));
  }
  return 0;
}
