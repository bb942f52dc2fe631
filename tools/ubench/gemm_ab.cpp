// A/B harness for the bf16 NT GEMM: the library under test (gritlm_amd/libgritlm_hip.so) against the previous round's kernel
// (tools/ubench/_r01/libgemm_r01.so, built from git history by build_gemm_ab.sh).  Both accumulate every output in the same
// k order, so the outputs must be BIT-IDENTICAL: small/ragged shapes (odd K-tile counts, M/N tails, grouped + gathered rows,
// all epilogues) are compared word for word over many repetitions (race screen), then the four GritLM-7B shapes are timed
// interleaved on random data.   usage: gemm_ab.bin [check|time|all] [reps]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef int (*gemm_fn)(const void*, const void*, void*, int64_t, int, int, int64_t, int64_t, int64_t, int, const void*, int64_t, void*);
typedef int (*rope_fn)(const void*, const void*, void*, int64_t, int, int, int64_t, int64_t, int64_t, const float*, const float*,
                       const int32_t*, int, int, int, void*);
typedef int (*grouped_fn)(const void*, const int32_t*, const void*, void*, const int32_t*, int, int64_t, int, int, int64_t, int64_t,
                          int64_t, int64_t, int, void*);
typedef const char* (*err_fn)(void);
struct Lib { gemm_fn gemm; rope_fn rope; grouped_fn grouped; err_fn err; };

static Lib load(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(2); }
  Lib l;
  l.gemm = (gemm_fn)dlsym(h, "grit_gemm_bf16_nt");
  l.rope = (rope_fn)dlsym(h, "grit_gemm_bf16_nt_rope");
  l.grouped = (grouped_fn)dlsym(h, "grit_gemm_bf16_nt_grouped");
  l.err = (err_fn)dlsym(h, "grit_last_error");
  if (!l.gemm || !l.rope || !l.grouped) { fprintf(stderr, "%s: missing symbols\n", path); exit(2); }
  return l;
}

__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32) * 40503u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float f = ((x & 0xffffff) / 16777216.0f * 2.0f - 1.0f) * scale;   // uniform [-scale, scale), full sign/mantissa activity
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
  }
}
__global__ void fill_f32(float* p, int64_t n, uint32_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
    p[i] = (x & 0xffffff) / 16777216.0f * 2.0f - 1.0f;
  }
}
__global__ void count_diff(const uint16_t* a, const uint16_t* b, int64_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}

static uint16_t* dev_bf16(int64_t n, uint32_t seed, float scale) {
  uint16_t* p; CK(hipMalloc(&p, n * 2 + 64));
  fill_bf16<<<2048, 256>>>(p, n, seed, scale);
  return p;
}

enum { STORE = 0, RESIDUAL = 1, SWIGLU = 2, ROPE = 3, GROUPED_STORE = 10, GROUPED_SWIGLU = 12 };

struct Case { int64_t M; int N, K, epi; };

static unsigned long long run_case(const Lib& a, const Lib& b, const Case& c, int reps, bool time_it, FILE* log) {
  const int64_t M = c.M; const int N = c.N, K = c.K;
  const bool grouped = c.epi >= 10;
  const int epi = grouped ? c.epi - 10 : c.epi;
  const int outN = epi == SWIGLU ? N / 2 : (epi == 6 ? 2 * N : N);        // 6 = SWIGLU_BWD: C = [d_gate | d_up], residual = saved [gate | up]
  const int ngroups = 5;
  uint16_t* A = dev_bf16(M * K, 11, 1.0f);
  uint16_t* W = dev_bf16((int64_t)N * K * (grouped ? ngroups : 1), 23, 0.05f);
  uint16_t* R = epi == RESIDUAL ? dev_bf16(M * N, 37, 1.0f) : (epi == 6 ? dev_bf16(M * 2 * N, 37, 1.0f) : nullptr);
  uint16_t *Ca, *Cb;
  CK(hipMalloc(&Ca, M * outN * 2 + 64)); CK(hipMalloc(&Cb, M * outN * 2 + 64));
  float *cosT = nullptr, *sinT = nullptr;
  if (epi == ROPE) {
    CK(hipMalloc(&cosT, 512 * 64 * 4)); CK(hipMalloc(&sinT, 512 * 64 * 4));
    fill_f32<<<64, 256>>>(cosT, 512 * 64, 5); fill_f32<<<64, 256>>>(sinT, 512 * 64, 7);
  }
  int32_t *counts = nullptr, *rows = nullptr;
  if (grouped) {
    std::vector<int32_t> hc(ngroups), hr(M);
    int64_t left = M;
    for (int g = 0; g < ngroups; ++g) { hc[g] = g == ngroups - 1 ? (int32_t)left : (int32_t)((M / ngroups) + (g % 2 ? 37 : -37)) ; if (hc[g] > left) hc[g] = (int32_t)left; left -= hc[g]; }
    hc[1] += 0;
    for (int64_t i = 0; i < M; ++i) hr[i] = (int32_t)((i * 7919) % M);      // a permutation when gcd(7919, M) = 1; any map is fine for A/B
    CK(hipMalloc(&counts, ngroups * 4)); CK(hipMalloc(&rows, M * 4));
    CK(hipMemcpy(counts, hc.data(), ngroups * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rows, hr.data(), M * 4, hipMemcpyHostToDevice));
  }
  unsigned long long* dcount; CK(hipMalloc(&dcount, 8)); CK(hipMemset(dcount, 0, 8));
  auto call = [&](const Lib& l, uint16_t* C) {
    int rc;
    if (grouped) rc = l.grouped(A, rows, W, C, counts, ngroups, M, N, K, K, K, (int64_t)N * K, outN, epi, nullptr);
    else if (epi == ROPE) rc = l.rope(A, W, C, M, N, K, K, K, N, cosT, sinT, nullptr, 512, 512, (N / 128) * 128 - (N >= 256 ? 128 : 0), nullptr);
    else rc = l.gemm(A, W, C, M, N, K, K, K, outN, epi, R, epi == 6 ? 2 * N : N, nullptr);
    if (rc != 0) { fprintf(stderr, "launch rc=%d (%s)\n", rc, l.err ? l.err() : "?"); exit(3); }
  };
  unsigned long long total = 0;
  if (!time_it) {
    CK(hipMemset(Ca, 0xff, M * outN * 2)); 
    call(a, Ca);
    for (int r = 0; r < reps; ++r) {
      CK(hipMemset(Cb, 0xee, M * outN * 2));
      call(b, Cb);
      count_diff<<<1024, 256>>>(Ca, Cb, M * outN, dcount);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&total, dcount, 8, hipMemcpyDeviceToHost));
    fprintf(log, "check M=%lld N=%d K=%d epi=%d reps=%d : %llu differing words%s\n", (long long)M, N, K, c.epi, reps, total, total ? "  <-- MISMATCH" : "");
    if (total) {
      std::vector<uint16_t> ha(M * outN), hb(M * outN);
      CK(hipMemcpy(ha.data(), Ca, M * outN * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), Cb, M * outN * 2, hipMemcpyDeviceToHost));
      int shown = 0; long long rows_bad = 0, last_row = -1;
      for (int64_t i = 0; i < M * outN; ++i) if (ha[i] != hb[i]) {
        if (i / outN != last_row) { ++rows_bad; last_row = i / outN; }
        if (shown < 24) { fprintf(log, "   diff at row %lld col %lld : r01 %04x new %04x\n", (long long)(i / outN), (long long)(i % outN), ha[i], hb[i]); ++shown; }
      }
      fprintf(log, "   rows with differences: %lld\n", rows_bad);
    }
  } else {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * M * N * K;
    call(a, Ca); call(b, Cb); CK(hipDeviceSynchronize());
    double best[2] = {1e30, 1e30}, sum[2] = {0, 0};
    const int rounds = reps, inner = 3;
    for (int r = 0; r < rounds; ++r)
      for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < inner; ++i) call(w ? b : a, w ? Cb : Ca);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= inner;
        if (ms < best[w]) best[w] = ms;
        sum[w] += ms;
      }
    count_diff<<<1024, 256>>>(Ca, Cb, M * outN, dcount);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&total, dcount, 8, hipMemcpyDeviceToHost));
    fprintf(log, "time M=%lld N=%d K=%d epi=%d : r01 %.3f ms (%.0f TF, min %.3f)   new %.3f ms (%.0f TF, min %.3f)   new/r01 speed %.3f   diff words %llu\n",
            (long long)M, N, K, c.epi, sum[0] / rounds, flop / (sum[0] / rounds * 1e-3) / 1e12, best[0], sum[1] / rounds,
            flop / (sum[1] / rounds * 1e-3) / 1e12, best[1], sum[0] / sum[1], total);
  }
  fflush(log);
  (void)hipFree(A); (void)hipFree(W); if (R) (void)hipFree(R); (void)hipFree(Ca); (void)hipFree(Cb); if (cosT) { (void)hipFree(cosT); (void)hipFree(sinT); }
  if (counts) { (void)hipFree(counts); (void)hipFree(rows); }
  (void)hipFree(dcount);
  return total;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  const int reps = (argc > 2 && strcmp(mode, "case")) ? atoi(argv[2]) : 10;
  const char* newlib = getenv("GEMM_NEW") ? getenv("GEMM_NEW") : "gritlm_amd/libgritlm_hip.so";
  const char* oldlib = getenv("GEMM_OLD") ? getenv("GEMM_OLD") : "tools/ubench/_r01/libgemm_r01.so";
  Lib a = load(oldlib), b = load(newlib);
  if (getenv("GEMM_SELF_AB")) {
    // library "a" is a COPY of the library under test: latch GRIT_GEMM_NOPERSIST=1 into its launch knobs (read at its first launch),
    // so the run compares the persistent launch form (b) with the one-workgroup-per-tile form (a) of the SAME kernels bit for bit
    setenv("GRIT_GEMM_NOPERSIST", "1", 1);
    uint16_t* t; CK(hipMalloc(&t, 256 * 256 * 2 * 3)); CK(hipMemset(t, 0, 256 * 256 * 2 * 3));
    a.gemm(t, t + 65536, t + 131072, 256, 256, 64, 64, 64, 256, 0, nullptr, 0, nullptr);
    CK(hipDeviceSynchronize());
    unsetenv("GRIT_GEMM_NOPERSIST");
    (void)hipFree(t);
  }
  unsigned long long bad = 0;
  if (!strcmp(mode, "check") || !strcmp(mode, "all")) {
    const Case cases[] = {
        {256, 256, 64, STORE}, {256, 256, 128, STORE}, {256, 256, 192, STORE}, {300, 272, 320, STORE}, {1000, 1040, 448, STORE},
        {4096, 1024, 512, RESIDUAL}, {777, 528, 576, RESIDUAL}, {4096, 2048, 512, SWIGLU}, {1111, 1088, 192, SWIGLU},
        {2048, 1536, 512, ROPE}, {1500, 768, 320, ROPE}, {5000, 768, 256, GROUPED_STORE}, {5000, 1024, 320, GROUPED_SWIGLU},
        {8192, 4096, 4096, STORE}, {8192, 4096, 14336, RESIDUAL}, {4096, 28672, 4096, SWIGLU}, {8192, 6144, 4096, ROPE},
        // persistent path (more tiles than CUs, even K-tile count >= 4) with ragged M / N edges and a tile count that is not a multiple of 256
        {9000, 4352, 512, STORE}, {9000, 4352, 512, RESIDUAL}, {9000, 4352, 512, SWIGLU}, {9000, 4352, 256, ROPE}, {4100, 4352, 256, STORE},
        {70000, 1280, 384, RESIDUAL},
        // SWIGLU_BWD (epi 6: C = [d_gate | d_up], residual = saved [gate | up]): per-tile and persistent launches, ragged M / N edges
        {300, 272, 320, 6}, {777, 528, 576, 6}, {4096, 2048, 512, 6}, {9000, 4352, 512, 6}, {70000, 1280, 384, 6}, {4100, 14336, 4096, 6},
    };
    for (const Case& c : cases) bad += run_case(a, b, c, c.M * (int64_t)c.N > (1 << 24) ? 3 : reps * 3, false, stdout);
  }
  if (!strcmp(mode, "case")) {                  // gemm_ab.bin case M N K epi [reps]
    const Case c{atoll(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5])};
    bad += run_case(a, b, c, argc > 6 ? atoi(argv[6]) : 4, true, stdout);
  }
  if (!strcmp(mode, "time") || !strcmp(mode, "all")) {
    const int64_t M = getenv("AB_M") ? atoll(getenv("AB_M")) : 131072;
    const Case cases[] = {{M, 6144, 4096, ROPE}, {M, 4096, 4096, RESIDUAL}, {M, 28672, 4096, SWIGLU}, {M, 4096, 14336, RESIDUAL}, {M, 6144, 4096, STORE}};
    for (const Case& c : cases) bad += run_case(a, b, c, 4, true, stdout);
  }
  printf(bad ? "RESULT: MISMATCH (%llu words)\n" : "RESULT: bit-identical\n", bad);
  return bad ? 1 : 0;
}
