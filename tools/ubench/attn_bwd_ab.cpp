// A/B harness for the attention backward: library under test vs the previous kernels (a private build of an older
// attention_bwd.hip, tools/ubench/build_attn_bwd_ab.sh).  The arithmetic (MFMA order, exp2 arguments, bf16 roundings) is unchanged
// between the two, so dq / dk / dv must be BIT-IDENTICAL; then both are timed interleaved.   usage: attn_bwd_ab.bin [check|time|all]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef int (*fpad_fn)(const void*, const uint64_t*, void*, float*, int, int, int, int, int, int64_t, int64_t, float, void*);
typedef int (*fvar_fn)(const void*, const int32_t*, void*, float*, int, int, int, int, int, int64_t, int64_t, float, void*);
typedef int (*bpad_fn)(const void*, const uint64_t*, const void*, const void*, const float*, float*, void*, int, int, int, int, int, int64_t, int64_t, float, void*);
typedef int (*bvar_fn)(const void*, const int32_t*, const void*, const void*, const float*, float*, void*, int, int, int64_t, int, int, int, int64_t, int64_t, float, void*);
struct Lib { bpad_fn bidir, causal; bvar_fn vbidir, vcausal; };
static void* must(void* h, const char* n) { void* p = dlsym(h, n); if (!p) { fprintf(stderr, "missing %s\n", n); exit(2); } return p; }
static void* open_lib(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(2); }
  return h;
}
static Lib load(const char* path) {
  void* h = open_lib(path);
  return Lib{(bpad_fn)must(h, "grit_attn_bidir_bwd"), (bpad_fn)must(h, "grit_attn_causal_bwd"), (bvar_fn)must(h, "grit_attn_bidir_varlen_bwd"),
             (bvar_fn)must(h, "grit_attn_causal_varlen_bwd")};
}
__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32) * 40503u;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float f = ((x & 0xffffff) / 16777216.0f * 2.0f - 1.0f) * scale;
    uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
  }
}
__global__ void count_diff(const uint32_t* a, const uint32_t* b, int64_t n, unsigned long long* out) {
  unsigned long long c = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(out, c);
}
static fpad_fn fwd_pad, fwd_pad_causal; static fvar_fn fwd_var, fwd_var_causal;
static unsigned long long run(const Lib& a, const Lib& b, int B, int S, int nq, int nkv, bool varlen, bool causal, bool ragged, bool timing) {
  const int d = 128; const int64_t stride = (int64_t)(nq + 2 * nkv) * d, ostride = (int64_t)nq * d;
  std::vector<int> lens(B);
  for (int i = 0; i < B; ++i) lens[i] = ragged ? 1 + (int)((i * 2654435761u >> 8) % S) : S;
  if (ragged) lens[0] = S;
  int64_t T = 0; std::vector<int32_t> cu(B + 1, 0);
  for (int i = 0; i < B; ++i) cu[i + 1] = cu[i] + lens[i];
  T = varlen ? cu[B] : (int64_t)B * S;
  const int W = (S + 63) / 64;
  std::vector<uint64_t> bits((size_t)B * W, 0);
  for (int i = 0; i < B; ++i) for (int k = 0; k < lens[i]; ++k) bits[(size_t)i * W + k / 64] |= 1ull << (k % 64);
  if (ragged && !varlen && B > 2) bits[(size_t)2 * W] &= ~0xff00ull;           // holes inside a row (padded layout only)
  uint16_t *qkv, *out, *dout, *ga, *gb; float *lse, *delta; uint64_t* dbits; int32_t* dcu; unsigned long long* dc;
  const int64_t nl = varlen ? T * nq : (int64_t)B * nq * S;
  CK(hipMalloc(&qkv, T * stride * 2)); fill_bf16<<<2048, 256>>>(qkv, T * stride, 77, 2.0f);
  CK(hipMalloc(&dout, T * ostride * 2)); fill_bf16<<<2048, 256>>>(dout, T * ostride, 1234, 1.0f);
  CK(hipMalloc(&out, T * ostride * 2)); CK(hipMalloc(&ga, T * stride * 2)); CK(hipMalloc(&gb, T * stride * 2));
  CK(hipMalloc(&lse, nl * 4)); CK(hipMalloc(&delta, nl * 4));
  CK(hipMalloc(&dbits, bits.size() * 8)); CK(hipMemcpy(dbits, bits.data(), bits.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dcu, (B + 1) * 4)); CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dc, 8)); CK(hipMemset(dc, 0, 8));
  CK(hipMemset(ga, 0x11, T * stride * 2)); CK(hipMemset(gb, 0x11, T * stride * 2)); CK(hipMemset(out, 0, T * ostride * 2));
  const float scale = 0.08838834764831845f;
  int rc = varlen ? (causal ? fwd_var_causal : fwd_var)(qkv, dcu, out, lse, B, S, nq, nkv, d, stride, ostride, scale, nullptr)
                  : (causal ? fwd_pad_causal : fwd_pad)(qkv, dbits, out, lse, B, S, nq, nkv, d, stride, ostride, scale, nullptr);
  if (rc) { fprintf(stderr, "fwd rc=%d\n", rc); exit(3); }
  auto call = [&](const Lib& l, uint16_t* g) {
    int r;
    if (varlen) r = (causal ? l.vcausal : l.vbidir)(qkv, dcu, out, dout, lse, delta, g, B, S, T, nq, nkv, d, stride, ostride, scale, nullptr);
    else r = (causal ? l.causal : l.bidir)(qkv, dbits, out, dout, lse, delta, g, B, S, nq, nkv, d, stride, ostride, scale, nullptr);
    if (r) { fprintf(stderr, "bwd rc=%d\n", r); exit(3); }
  };
  call(a, ga); call(b, gb);
  count_diff<<<1024, 256>>>((const uint32_t*)ga, (const uint32_t*)gb, T * stride / 2, dc);
  CK(hipDeviceSynchronize());
  unsigned long long bad; CK(hipMemcpy(&bad, dc, 8, hipMemcpyDeviceToHost));
  double ms[2] = {0, 0};
  if (timing) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 6; ++r) for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 4; ++i) call(w ? b : a, w ? gb : ga);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[w] += t / 4;
    }
    ms[0] /= 6; ms[1] /= 6;
  }
  double flop = 0; for (int i = 0; i < B; ++i) flop += 10.0 * nq * (double)lens[i] * lens[i] * d * (causal ? 0.5 : 1.0);   // 5 products (S recomputed twice: 7 executed)
  printf("%s%s%s B=%d S=%d nq=%d nkv=%d : %llu differing words", varlen ? "varlen " : "padded ", causal ? "causal " : "bidir ", ragged ? "ragged" : "full", B, S, nq, nkv, bad);
  if (timing) printf("   old %.3f ms (%.0f TF)  new %.3f ms (%.0f TF)  speed %.3f", ms[0], flop / ms[0] / 1e9, ms[1], flop / ms[1] / 1e9, ms[0] / ms[1]);
  printf("%s\n", bad ? "  <-- MISMATCH" : "");
  (void)hipFree(qkv); (void)hipFree(out); (void)hipFree(dout); (void)hipFree(ga); (void)hipFree(gb); (void)hipFree(lse); (void)hipFree(delta);
  (void)hipFree(dbits); (void)hipFree(dcu); (void)hipFree(dc);
  return bad;
}
int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  const char* newp = getenv("ATTN_NEW") ? getenv("ATTN_NEW") : "gritlm_amd/libgritlm_hip.so";
  Lib a = load(getenv("ATTN_OLD") ? getenv("ATTN_OLD") : "tools/ubench/_r01/libattn_bwd_old.so");
  Lib b = load(newp);
  void* hf = open_lib(getenv("ATTN_FWD") ? getenv("ATTN_FWD") : "gritlm_amd/libgritlm_hip.so");
  fwd_pad = (fpad_fn)must(hf, "grit_attn_bidir_fwd"); fwd_pad_causal = (fpad_fn)must(hf, "grit_attn_causal_fwd");
  fwd_var = (fvar_fn)must(hf, "grit_attn_bidir_varlen_fwd"); fwd_var_causal = (fvar_fn)must(hf, "grit_attn_causal_varlen_fwd");
  unsigned long long bad = 0;
  if (strcmp(mode, "time")) {
    for (int rep = 0; rep < 2; ++rep) {
      bad += run(a, b, 5, 512, 8, 2, false, false, true, false);  bad += run(a, b, 5, 500, 8, 2, true, false, true, false);
      bad += run(a, b, 3, 333, 4, 2, false, true, true, false);   bad += run(a, b, 4, 1000, 4, 1, true, true, true, false);
      bad += run(a, b, 2, 64, 2, 1, false, false, false, false);  bad += run(a, b, 3, 33, 2, 1, true, false, true, false);
      bad += run(a, b, 2, 2048, 4, 1, false, false, true, false); bad += run(a, b, 9, 700, 8, 2, true, true, true, false);
      bad += run(a, b, 20, 640, 32, 8, false, false, true, false); bad += run(a, b, 3, 129, 8, 8, false, false, true, false);
      bad += run(a, b, 11, 1536, 32, 8, true, false, true, false);
    }
  }
  if (strcmp(mode, "check")) {
    bad += run(a, b, 32, 512, 32, 8, false, false, false, true);      // one GradCache chunk of the contrastive bench
    bad += run(a, b, 32, 512, 32, 8, true, false, true, true);
    bad += run(a, b, 8, 2048, 32, 8, false, false, false, true);
    bad += run(a, b, 8, 2048, 32, 8, false, true, false, true);
  }
  printf(bad ? "RESULT: MISMATCH\n" : "RESULT: bit-identical\n");
  return bad ? 1 : 0;
}
