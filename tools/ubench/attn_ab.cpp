// A/B harness for the attention forward: library under test vs the previous kernel (tools/ubench/_r01/libattn_old.so built from git
// history by build_attn_ab.sh).  The arithmetic (MFMA order, softmax) is unchanged between the two, so outputs and log-sum-exp rows
// must be BIT-IDENTICAL; then both are timed interleaved.   usage: attn_ab.bin [check|time|all]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef int (*pad_fn)(const void*, const uint64_t*, void*, float*, int, int, int, int, int, int64_t, int64_t, float, void*);
typedef int (*var_fn)(const void*, const int32_t*, void*, float*, int, int, int, int, int, int64_t, int64_t, float, void*);
struct Lib { pad_fn bidir, causal; var_fn vbidir, vcausal; };
static Lib load(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); exit(2); }
  Lib l{(pad_fn)dlsym(h, "grit_attn_bidir_fwd"), (pad_fn)dlsym(h, "grit_attn_causal_fwd"), (var_fn)dlsym(h, "grit_attn_bidir_varlen_fwd"),
        (var_fn)dlsym(h, "grit_attn_causal_varlen_fwd")};
  if (!l.bidir || !l.causal || !l.vbidir || !l.vcausal) { fprintf(stderr, "%s: missing symbols\n", path); exit(2); }
  return l;
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
__global__ void max_diff_bf16(const uint16_t* a, const uint16_t* b, int64_t n, unsigned int* out) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)a[i] << 16), y = __uint_as_float((uint32_t)b[i] << 16);
    const float d = fabsf(x - y);
    m = (d == d) ? fmaxf(m, d) : 1e30f;
  }
  atomicMax(out, __float_as_uint(m));          // non-negative floats order like their bit patterns
}
__global__ void max_diff_f32(const float* a, const float* b, int64_t n, unsigned int* out) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (a[i] == b[i]) continue;                  // covers equal infinities
    const float d = fabsf(a[i] - b[i]);
    m = (d == d) ? fmaxf(m, d) : 1e30f;
  }
  atomicMax(out, __float_as_uint(m));
}
static unsigned long long run(const Lib& a, const Lib& b, int B, int S, int nq, int nkv, bool varlen, bool causal, bool ragged, bool timing) {
  const int d = 128; const int64_t stride = (int64_t)(nq + 2 * nkv) * d, ostride = (int64_t)nq * d;
  std::vector<int> lens(B);
  for (int i = 0; i < B; ++i) lens[i] = ragged ? 1 + (int)((i * 2654435761u >> 8) % S) : S;
  if (ragged) lens[0] = S;
  int64_t T = 0; std::vector<int32_t> cu(B + 1, 0);
  for (int i = 0; i < B; ++i) { cu[i + 1] = cu[i] + lens[i]; }
  T = varlen ? cu[B] : (int64_t)B * S;
  const int W = (S + 63) / 64;
  std::vector<uint64_t> bits((size_t)B * W, 0);
  for (int i = 0; i < B; ++i) for (int k = 0; k < lens[i]; ++k) bits[(size_t)i * W + k / 64] |= 1ull << (k % 64);
  if (ragged && !varlen && B > 2) bits[(size_t)2 * W] &= ~0xff00ull;           // holes inside a row (padded layout only)
  uint16_t* qkv; CK(hipMalloc(&qkv, T * stride * 2)); fill_bf16<<<2048, 256>>>(qkv, T * stride, 77, 2.0f);
  uint16_t *oa, *ob; float *la, *lb; uint64_t* dbits; int32_t* dcu; unsigned long long* dc;
  const int64_t nl = varlen ? T * nq : (int64_t)B * nq * S;
  CK(hipMalloc(&oa, T * ostride * 2)); CK(hipMalloc(&ob, T * ostride * 2)); CK(hipMalloc(&la, nl * 4)); CK(hipMalloc(&lb, nl * 4));
  CK(hipMalloc(&dbits, bits.size() * 8)); CK(hipMemcpy(dbits, bits.data(), bits.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dcu, (B + 1) * 4)); CK(hipMemcpy(dcu, cu.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dc, 8)); CK(hipMemset(dc, 0, 8));
  CK(hipMemset(oa, 0x11, T * ostride * 2)); CK(hipMemset(ob, 0x11, T * ostride * 2)); CK(hipMemset(la, 0x11, nl * 4)); CK(hipMemset(lb, 0x11, nl * 4));
  const float scale = 0.08838834764831845f;
  auto call = [&](const Lib& l, uint16_t* o, float* ls) {
    int rc;
    if (varlen) rc = (causal ? l.vcausal : l.vbidir)(qkv, dcu, o, ls, B, S, nq, nkv, d, stride, ostride, scale, nullptr);
    else rc = (causal ? l.causal : l.bidir)(qkv, dbits, o, ls, B, S, nq, nkv, d, stride, ostride, scale, nullptr);
    if (rc) { fprintf(stderr, "rc=%d\n", rc); exit(3); }
  };
  call(a, oa, la); call(b, ob, lb);
  count_diff<<<1024, 256>>>((const uint32_t*)oa, (const uint32_t*)ob, T * ostride / 2, dc);
  count_diff<<<1024, 256>>>((const uint32_t*)la, (const uint32_t*)lb, nl, dc);
  CK(hipDeviceSynchronize());
  unsigned long long bad; CK(hipMemcpy(&bad, dc, 8, hipMemcpyDeviceToHost));
  // ATTN_TOL=<x>: the two libraries may differ in rounding (e.g. deferred softmax rescale): report max |diff| of O (bf16) and LSE
  // instead of bit equality; a pair counts as a mismatch when the O difference exceeds x
  float dmax_o = 0.f, dmax_l = 0.f;
  const char* tol_s = getenv("ATTN_TOL");
  if (tol_s) {
    unsigned int* dm; CK(hipMalloc(&dm, 8)); CK(hipMemset(dm, 0, 8));
    max_diff_bf16<<<1024, 256>>>(oa, ob, T * ostride, dm);
    max_diff_f32<<<1024, 256>>>(la, lb, nl, dm + 1);
    unsigned int hm[2]; CK(hipMemcpy(hm, dm, 8, hipMemcpyDeviceToHost));
    memcpy(&dmax_o, &hm[0], 4); memcpy(&dmax_l, &hm[1], 4);
    (void)hipFree(dm);
    bad = (dmax_o > atof(tol_s) || dmax_l > 1e-3f) ? 1 : 0;
  }
  double ms[2] = {0, 0};
  if (timing) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 6; ++r) for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 4; ++i) call(w ? b : a, w ? ob : oa, w ? lb : la);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); ms[w] += t / 4;
    }
    ms[0] /= 6; ms[1] /= 6;
  }
  double flop = 0; for (int i = 0; i < B; ++i) flop += 4.0 * nq * (double)lens[i] * lens[i] * d * (causal ? 0.5 : 1.0);
  if (tol_s) printf("%s%s%s B=%d S=%d nq=%d nkv=%d : max|dO| %.3g max|dLSE| %.3g", varlen ? "varlen " : "padded ", causal ? "causal " : "bidir ", ragged ? "ragged" : "full", B, S, nq, nkv, dmax_o, dmax_l);
  else printf("%s%s%s B=%d S=%d nq=%d nkv=%d : %llu differing words", varlen ? "varlen " : "padded ", causal ? "causal " : "bidir ", ragged ? "ragged" : "full", B, S, nq, nkv, bad);
  if (timing) printf("   old %.3f ms (%.0f TF)  new %.3f ms (%.0f TF)  speed %.3f", ms[0], flop / ms[0] / 1e9, ms[1], flop / ms[1] / 1e9, ms[0] / ms[1]);
  printf("%s\n", bad ? "  <-- MISMATCH" : "");
  (void)hipFree(qkv); (void)hipFree(oa); (void)hipFree(ob); (void)hipFree(la); (void)hipFree(lb); (void)hipFree(dbits); (void)hipFree(dcu); (void)hipFree(dc);
  return bad;
}
int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  Lib a = load(getenv("ATTN_OLD") ? getenv("ATTN_OLD") : "tools/ubench/_r01/libattn_old.so");
  Lib b = load(getenv("ATTN_NEW") ? getenv("ATTN_NEW") : "gritlm_amd/libgritlm_hip.so");
  unsigned long long bad = 0;
  if (strcmp(mode, "time")) {
    for (int rep = 0; rep < 3; ++rep) {
      bad += run(a, b, 5, 512, 8, 2, false, false, true, false);  bad += run(a, b, 5, 500, 8, 2, true, false, true, false);
      bad += run(a, b, 3, 333, 4, 2, false, true, true, false);   bad += run(a, b, 4, 1000, 4, 1, true, true, true, false);
      bad += run(a, b, 2, 64, 2, 1, false, false, false, false);  bad += run(a, b, 3, 33, 2, 1, true, false, true, false);
      bad += run(a, b, 2, 4096, 4, 1, false, false, true, false);
      bad += run(a, b, 9, 700, 8, 2, true, true, true, false);    bad += run(a, b, 70, 640, 32, 8, false, false, true, false);
      bad += run(a, b, 3, 129, 8, 8, false, false, true, false);  bad += run(a, b, 11, 1536, 32, 8, true, false, true, false);
    }
  }
  if (strcmp(mode, "check")) {
    bad += run(a, b, 256, 512, 32, 8, false, false, false, true);
    bad += run(a, b, 64, 2048, 32, 8, false, false, false, true);
    bad += run(a, b, 256, 512, 32, 8, true, false, true, true);
    bad += run(a, b, 64, 2048, 32, 8, false, true, false, true);
    bad += run(a, b, 4, 8192, 32, 8, false, false, false, true);
    bad += run(a, b, 8, 512, 32, 8, false, false, false, true);
    bad += run(a, b, 1, 2048, 32, 8, false, false, false, true);
  }
  printf(bad ? "RESULT: MISMATCH\n" : (getenv("ATTN_TOL") ? "RESULT: within tolerance\n" : "RESULT: bit-identical\n"));
  return bad ? 1 : 0;
}
