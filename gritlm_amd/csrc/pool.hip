// Masked mean / weighted-mean / cls / last-token pooling fused with L2-normalise, and its backward.
//
// Replaces GritLM.pooling + F.normalize (gritlm/gritlm.py:178-218, :156-158; training/model.py:151-165).
// The reference materialises hidden * mask.float() as a [B,S,H] fp32 temporary (2.1 GB at 256x512x4096)
// and reduces it; here one 512-thread workgroup per document streams the bf16 rows once (16 B per lane,
// 1 KiB per wave-instruction), skips rows whose pooling weight is zero (padding / instruction tokens are
// never read), accumulates in fp32 registers and normalises in the same launch.  HBM-bound:
// algorithmic bytes = B*S*H*2 read + B*H*4 written.
#include <atomic>

#include "common.h"

namespace grit {

constexpr int POOL_THREADS = 512;

// Pooling weights for one document, computed by wave 0 into LDS:
//   cw[k] = (position s_k, weight w_k) for the nz positions with non-zero weight, *den = denominator.
// mean: w = mask value; weightedmean: w = mask * cumsum(mask) (gritlm.py:211); cls: w[0] = 1;
// lasttoken: w[last s with mask != 0] = mask there (index clamped to 0 when the row is empty, :190-208).
__device__ void pool_weights(const int64_t* __restrict__ mrow, int instr, int S, int mode, int* cs, float* cw, int* nz_out,
                             float* den_out) {
  const int lane = threadIdx.x & 63;
  int nz = 0;
  if (mode == GRIT_POOL_CLS) {
    if (lane == 0) { cs[0] = 0; cw[0] = 1.f; *nz_out = 1; *den_out = 1.f; }
    return;
  }
  if (mode == GRIT_POOL_LASTTOKEN) {
    int last = -1;
    for (int base = 0; base < S; base += 64) {
      const int s = base + lane;
      const bool on = s < S && s >= instr && mrow[s] != 0;
      const unsigned long long b = __ballot(on);
      if (b) last = base + 63 - __builtin_clzll(b);
    }
    if (lane == 0) {
      const int idx = last < 0 ? 0 : last;
      const float w = (last < 0) ? 0.f : (float)mrow[idx];
      cs[0] = idx; cw[0] = w; *nz_out = 1; *den_out = 1.f;  // empty row: one zero-weight entry == hidden*0
    }
    return;
  }
  long long carry = 0, den = 0;
  for (int base = 0; base < S; base += 64) {
    const int s = base + lane;
    long long m = (s < S && s >= instr) ? (long long)mrow[s] : 0;
    long long w = m;
    if (mode == GRIT_POOL_WEIGHTEDMEAN) {
      long long inc = m;  // inclusive scan over the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      w = m * (carry + inc);
      carry += __shfl(inc, 63, 64);
    }
    long long ws = w;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ws += __shfl_xor(ws, o, 64);
    den += ws;
    const bool on = w != 0;
    const unsigned long long b = __ballot(on);
    if (on) {
      const int pos = nz + __popcll(b & ((1ull << lane) - 1ull));
      cs[pos] = s; cw[pos] = (float)w;
    }
    nz += __popcll(b);
  }
  if (lane == 0) { *nz_out = nz; *den_out = (float)den; }
}

__device__ __forceinline__ void fma8(float (&a)[8], const uint4& v, float w) {
  a[0] += w * bflo(v.x); a[1] += w * bfhi(v.x); a[2] += w * bflo(v.y); a[3] += w * bfhi(v.y);
  a[4] += w * bflo(v.z); a[5] += w * bfhi(v.z); a[6] += w * bflo(v.w); a[7] += w * bfhi(v.w);
}

__global__ void __launch_bounds__(POOL_THREADS) pool_norm_fwd_k(const uint16_t* __restrict__ hidden, const int64_t* __restrict__ mask,
                                                                const int32_t* __restrict__ instr_len, float* __restrict__ out,
                                                                float* __restrict__ inv_norm, int S, int H, int mode, int normalize) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* cs = reinterpret_cast<int*>(smem);                   // [S]
  float* cw = reinterpret_cast<float*>(smem + 4 * S);       // [S]
  float* red = reinterpret_cast<float*>(smem + 8 * S);      // [8] + nz + den
  int* nzp = reinterpret_cast<int*>(red + 8);
  float* denp = red + 9;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) pool_weights(mask + (int64_t)b * S, instr_len ? instr_len[b] : 0, S, mode, cs, cw, nzp, denp);
  __syncthreads();
  const int nz = *nzp;
  const float den = *denp;  // den == 0 -> NaN/inf rows exactly like the unguarded reference (:213-214)
  const int HC = H >> 3;
  const uint4* hb = reinterpret_cast<const uint4*>(hidden + (int64_t)b * S * H);
  float* ob = out + (int64_t)b * H;
  float ssq = 0.f;
  for (int cc = tid; cc < HC; cc += POOL_THREADS) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= nz; k += 8) {  // 8 row loads (128 B) in flight per lane: one workgroup per CU has to cover the HBM latency alone
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = hb[(int64_t)cs[k + u] * HC + cc];
#pragma unroll
      for (int u = 0; u < 8; ++u) fma8(a, v[u], cw[k + u]);       // same accumulation order as one row at a time
    }
    for (; k + 4 <= nz; k += 4) {
      const uint4 v0 = hb[(int64_t)cs[k] * HC + cc], v1 = hb[(int64_t)cs[k + 1] * HC + cc];
      const uint4 v2 = hb[(int64_t)cs[k + 2] * HC + cc], v3 = hb[(int64_t)cs[k + 3] * HC + cc];
      fma8(a, v0, cw[k]); fma8(a, v1, cw[k + 1]); fma8(a, v2, cw[k + 2]); fma8(a, v3, cw[k + 3]);
    }
    for (; k < nz; ++k) fma8(a, hb[(int64_t)cs[k] * HC + cc], cw[k]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = a[e] / den; ssq += a[e] * a[e]; }
    float4* o4 = reinterpret_cast<float4*>(ob + cc * 8);
    o4[0] = make_float4(a[0], a[1], a[2], a[3]);
    o4[1] = make_float4(a[4], a[5], a[6], a[7]);
  }
  if (!normalize) {
    if (inv_norm && tid == 0) inv_norm[b] = 1.f;
    return;
  }
  ssq = wave_sum(ssq);
  if (lane == 0) red[wave] = ssq;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < POOL_THREADS / 64; ++w) tot += red[w];
  const float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);  // F.normalize eps
  if (inv_norm && tid == 0) inv_norm[b] = inv;
  for (int cc = tid; cc < HC; cc += POOL_THREADS) {  // each thread rescales what it wrote itself
    float4* o4 = reinterpret_cast<float4*>(ob + cc * 8);
    float4 x = o4[0], y = o4[1];
    x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv; y.x *= inv; y.y *= inv; y.z *= inv; y.w *= inv;
    o4[0] = x; o4[1] = y;
  }
}

// Packed (un-padded) batches: document b occupies rows [cu[b], cu[b+1]) of hidden [T,H]; every row is a real token, the first
// instr_len[b] of them are attended to but not pooled.  Same arithmetic as pool_norm_fwd_k with mask == 1 on [instr, len).
__global__ void __launch_bounds__(POOL_THREADS) pool_norm_varlen_fwd_k(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ cu,
                                                                       const int32_t* __restrict__ instr_len, float* __restrict__ out,
                                                                       float* __restrict__ inv_norm, int H, int mode, int normalize) {
  __shared__ float red[POOL_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t start = cu[b];
  const int len = cu[b + 1] - cu[b];
  int instr = instr_len ? instr_len[b] : 0;
  instr = instr < len ? instr : len;
  int s0 = instr, s1 = len;
  float den = (float)(len - instr);
  if (mode == GRIT_POOL_WEIGHTEDMEAN) { const float n = (float)(len - instr); den = 0.5f * n * (n + 1.f); }
  if (mode == GRIT_POOL_CLS) { s0 = 0; s1 = len > 0 ? 1 : 0; den = 1.f; }
  if (mode == GRIT_POOL_LASTTOKEN) { s0 = len > 0 ? len - 1 : 0; s1 = len; den = 1.f; }
  const bool ramp = (mode == GRIT_POOL_WEIGHTEDMEAN);
  const int HC = H >> 3;
  const uint4* hb = reinterpret_cast<const uint4*>(hidden) + start * HC;
  float* ob = out + (int64_t)b * H;
  float ssq = 0.f;
  for (int cc = tid; cc < HC; cc += POOL_THREADS) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = s0;
    for (; s + 8 <= s1; s += 8) {  // 8 row loads in flight per lane (see pool_norm_fwd_k)
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = hb[(int64_t)(s + u) * HC + cc];
      const float w0 = ramp ? (float)(s - instr + 1) : 1.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) fma8(a, v[u], ramp ? w0 + (float)u : 1.f);
    }
    for (; s + 4 <= s1; s += 4) {
      const uint4 v0 = hb[(int64_t)s * HC + cc], v1 = hb[(int64_t)(s + 1) * HC + cc];
      const uint4 v2 = hb[(int64_t)(s + 2) * HC + cc], v3 = hb[(int64_t)(s + 3) * HC + cc];
      const float w0 = ramp ? (float)(s - instr + 1) : 1.f;
      fma8(a, v0, w0); fma8(a, v1, ramp ? w0 + 1.f : 1.f); fma8(a, v2, ramp ? w0 + 2.f : 1.f); fma8(a, v3, ramp ? w0 + 3.f : 1.f);
    }
    for (; s < s1; ++s) fma8(a, hb[(int64_t)s * HC + cc], ramp ? (float)(s - instr + 1) : 1.f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = a[e] / den; ssq += a[e] * a[e]; }
    float4* o4 = reinterpret_cast<float4*>(ob + cc * 8);
    o4[0] = make_float4(a[0], a[1], a[2], a[3]);
    o4[1] = make_float4(a[4], a[5], a[6], a[7]);
  }
  if (!normalize) {
    if (inv_norm && tid == 0) inv_norm[b] = 1.f;
    return;
  }
  ssq = wave_sum(ssq);
  if (lane == 0) red[wave] = ssq;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < POOL_THREADS / 64; ++w) tot += red[w];
  const float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
  if (inv_norm && tid == 0) inv_norm[b] = inv;
  for (int cc = tid; cc < HC; cc += POOL_THREADS) {
    float4* o4 = reinterpret_cast<float4*>(ob + cc * 8);
    float4 x = o4[0], y = o4[1];
    x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv; y.x *= inv; y.y *= inv; y.z *= inv; y.w *= inv;
    o4[0] = x; o4[1] = y;
  }
}

// dhidden[b,s,:] = (w[s]/den) * g,  g = (dy - y (y.dy)) * inv_norm  (normalize)  or dy
__global__ void __launch_bounds__(POOL_THREADS) pool_norm_bwd_k(const float* __restrict__ y, const float* __restrict__ dy,
                                                                const float* __restrict__ inv_norm, const int64_t* __restrict__ mask,
                                                                const int32_t* __restrict__ instr_len, uint16_t* __restrict__ dhidden,
                                                                int S, int H, int mode, int normalize) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* g = reinterpret_cast<float*>(smem);                        // [H]
  int* cs = reinterpret_cast<int*>(smem + 4 * (size_t)H);           // [S]
  float* cw = reinterpret_cast<float*>(smem + 4 * (size_t)H + 4 * S);
  float* ws = reinterpret_cast<float*>(smem + 4 * (size_t)H + 8 * S);  // [S] dense weights
  float* red = ws + S;                                              // [8] + nz + den
  int* nzp = reinterpret_cast<int*>(red + 8);
  float* denp = red + 9;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) pool_weights(mask + (int64_t)b * S, instr_len ? instr_len[b] : 0, S, mode, cs, cw, nzp, denp);
  for (int s = tid; s < S; s += POOL_THREADS) ws[s] = 0.f;
  const float* yb = y + (int64_t)b * H;
  const float* dyb = dy + (int64_t)b * H;
  float dot = 0.f;
  if (normalize)
    for (int i = tid; i < H; i += POOL_THREADS) dot += yb[i] * dyb[i];
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < POOL_THREADS / 64; ++w) tot += red[w];
  const float inv = normalize ? inv_norm[b] : 1.f;
  for (int i = tid; i < H; i += POOL_THREADS) g[i] = normalize ? (dyb[i] - yb[i] * tot) * inv : dyb[i];
  const int nz = *nzp;
  const float inv_den = 1.0f / *denp;
  for (int k = tid; k < nz; k += POOL_THREADS) ws[cs[k]] = cw[k] * inv_den;
  __syncthreads();
  const int HC = H >> 3;
  uint4* db = reinterpret_cast<uint4*>(dhidden + (int64_t)b * S * H);
  for (int s = wave; s < S; s += POOL_THREADS / 64) {
    const float w = ws[s];
    for (int cc = lane; cc < HC; cc += 64) {
      uint4 o = make_uint4(0, 0, 0, 0);
      if (w != 0.f) {
        const float4 g0 = *reinterpret_cast<const float4*>(g + cc * 8), g1 = *reinterpret_cast<const float4*>(g + cc * 8 + 4);
        o.x = pack2bf(w * g0.x, w * g0.y); o.y = pack2bf(w * g0.z, w * g0.w);
        o.z = pack2bf(w * g1.x, w * g1.y); o.w = pack2bf(w * g1.z, w * g1.w);
      }
      db[(int64_t)s * HC + cc] = o;
    }
  }
}

// packed rows: dhidden[cu[b]+s, :] = w(s) * g with the weights of pool_norm_varlen_fwd_k (no mask: every packed row is a token)
__global__ void __launch_bounds__(POOL_THREADS) pool_norm_varlen_bwd_k(const float* __restrict__ y, const float* __restrict__ dy,
                                                                       const float* __restrict__ inv_norm, const int32_t* __restrict__ cu,
                                                                       const int32_t* __restrict__ instr_len,
                                                                       uint16_t* __restrict__ dhidden, int H, int mode, int normalize) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* g = reinterpret_cast<float*>(smem);  // [H]
  float* red = g + H;                         // [POOL_THREADS/64]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t start = cu[b];
  const int len = cu[b + 1] - cu[b];
  int instr = instr_len ? instr_len[b] : 0;
  instr = instr < len ? instr : len;
  int s0 = instr, s1 = len;
  float den = (float)(len - instr);
  if (mode == GRIT_POOL_WEIGHTEDMEAN) { const float n = (float)(len - instr); den = 0.5f * n * (n + 1.f); }
  if (mode == GRIT_POOL_CLS) { s0 = 0; s1 = len > 0 ? 1 : 0; den = 1.f; }
  if (mode == GRIT_POOL_LASTTOKEN) { s0 = len > 0 ? len - 1 : 0; s1 = len; den = 1.f; }
  const bool ramp = (mode == GRIT_POOL_WEIGHTEDMEAN);
  const float* yb = y + (int64_t)b * H;
  const float* dyb = dy + (int64_t)b * H;
  float dot = 0.f;
  if (normalize)
    for (int i = tid; i < H; i += POOL_THREADS) dot += yb[i] * dyb[i];
  dot = wave_sum(dot);
  if (lane == 0) red[wave] = dot;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < POOL_THREADS / 64; ++w) tot += red[w];
  const float inv = normalize ? inv_norm[b] : 1.f;
  for (int i = tid; i < H; i += POOL_THREADS) g[i] = normalize ? (dyb[i] - yb[i] * tot) * inv : dyb[i];
  __syncthreads();
  const float inv_den = 1.0f / den;
  const int HC = H >> 3;
  uint4* db = reinterpret_cast<uint4*>(dhidden) + start * HC;
  for (int s = wave; s < len; s += POOL_THREADS / 64) {
    float w = 0.f;
    if (s >= s0 && s < s1) w = (ramp ? (float)(s - instr + 1) : 1.f) * inv_den;
    for (int cc = lane; cc < HC; cc += 64) {
      uint4 o = make_uint4(0, 0, 0, 0);
      if (w != 0.f) {
        const float4 g0 = *reinterpret_cast<const float4*>(g + cc * 8), g1 = *reinterpret_cast<const float4*>(g + cc * 8 + 4);
        o.x = pack2bf(w * g0.x, w * g0.y); o.y = pack2bf(w * g0.z, w * g0.w);
        o.z = pack2bf(w * g1.x, w * g1.y); o.w = pack2bf(w * g1.z, w * g1.w);
      }
      db[(int64_t)s * HC + cc] = o;
    }
  }
}

}  // namespace grit

using namespace grit;

// A dynamic-LDS opt-in is a PER-DEVICE function attribute: set once per (kernel, device), from whichever thread gets there first
// (autograd worker threads call into the backward entry points concurrently).
template <typename KernelT>
static void lds_optin_once(KernelT kernel, std::atomic<uint64_t>& done, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
}

extern "C" {

int grit_pool_norm_fwd(const void* hidden, const int64_t* mask, const int32_t* instr_len, float* out, float* inv_norm, int B, int S,
                       int H, int mode, int normalize, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(hidden && mask && out, GRIT_E_BADARG, "grit_pool_norm_fwd: null pointer");
  GRIT_REQUIRE(B >= 0 && S > 0 && H > 0, GRIT_E_BADARG, "grit_pool_norm_fwd: bad sizes");
  GRIT_REQUIRE(mode >= GRIT_POOL_MEAN && mode <= GRIT_POOL_LASTTOKEN, GRIT_E_BADARG, "grit_pool_norm_fwd: unknown pooling mode %d", mode);
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_pool_norm_fwd: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(S <= 16384, GRIT_E_UNSUPPORTED, "grit_pool_norm_fwd: S=%d > 16384", S);
  GRIT_REQUIRE(aligned16(hidden) && aligned16(out), GRIT_E_BADARG, "grit_pool_norm_fwd: pointers must be 16-byte aligned");
  if (B == 0) return GRIT_OK;
  const size_t lds = 8 * (size_t)S + 64;
  static std::atomic<uint64_t> optin_f{0};
  lds_optin_once(pool_norm_fwd_k, optin_f, 160 * 1024);
  hipLaunchKernelGGL(pool_norm_fwd_k, dim3(B), dim3(POOL_THREADS), lds, (hipStream_t)stream, (const uint16_t*)hidden, mask, instr_len,
                     out, inv_norm, S, H, mode, normalize);
  GRIT_CHECK_LAUNCH("grit_pool_norm_fwd");
  return GRIT_OK;
}

int grit_pool_norm_varlen_fwd(const void* hidden, const int32_t* cu_seqlens, const int32_t* instr_len, float* out, float* inv_norm, int B,
                              int H, int mode, int normalize, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(hidden && cu_seqlens && out, GRIT_E_BADARG, "grit_pool_norm_varlen_fwd: null pointer");
  GRIT_REQUIRE(B >= 0 && H > 0, GRIT_E_BADARG, "grit_pool_norm_varlen_fwd: bad sizes");
  GRIT_REQUIRE(mode >= GRIT_POOL_MEAN && mode <= GRIT_POOL_LASTTOKEN, GRIT_E_BADARG, "grit_pool_norm_varlen_fwd: unknown pooling mode %d", mode);
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_pool_norm_varlen_fwd: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(aligned16(hidden) && aligned16(out), GRIT_E_BADARG, "grit_pool_norm_varlen_fwd: pointers must be 16-byte aligned");
  if (B == 0) return GRIT_OK;
  hipLaunchKernelGGL(pool_norm_varlen_fwd_k, dim3(B), dim3(POOL_THREADS), 0, (hipStream_t)stream, (const uint16_t*)hidden, cu_seqlens,
                     instr_len, out, inv_norm, H, mode, normalize);
  GRIT_CHECK_LAUNCH("grit_pool_norm_varlen_fwd");
  return GRIT_OK;
}

int grit_pool_norm_bwd(const float* y, const float* dy, const float* inv_norm, const int64_t* mask, const int32_t* instr_len,
                       void* dhidden, int B, int S, int H, int mode, int normalize, void* stream) {
  GRIT_REQUIRE(y && dy && mask && dhidden, GRIT_E_BADARG, "grit_pool_norm_bwd: null pointer");
  GRIT_REQUIRE(!normalize || inv_norm, GRIT_E_BADARG, "grit_pool_norm_bwd: inv_norm required when normalize");
  GRIT_REQUIRE(B >= 0 && S > 0 && H > 0, GRIT_E_BADARG, "grit_pool_norm_bwd: bad sizes");
  GRIT_REQUIRE(mode >= GRIT_POOL_MEAN && mode <= GRIT_POOL_LASTTOKEN, GRIT_E_BADARG, "grit_pool_norm_bwd: unknown pooling mode %d", mode);
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_pool_norm_bwd: H=%d must be a multiple of 8", H);
  const size_t lds = 4 * (size_t)H + 12 * (size_t)S + 64;
  GRIT_REQUIRE(lds <= 160 * 1024, GRIT_E_UNSUPPORTED, "grit_pool_norm_bwd: H=%d S=%d exceed LDS", H, S);
  GRIT_REQUIRE(aligned16(dhidden), GRIT_E_BADARG, "grit_pool_norm_bwd: pointers must be 16-byte aligned");
  if (B == 0) return GRIT_OK;
  static std::atomic<uint64_t> optin_b{0};
  lds_optin_once(pool_norm_bwd_k, optin_b, 160 * 1024);
  hipLaunchKernelGGL(pool_norm_bwd_k, dim3(B), dim3(POOL_THREADS), lds, (hipStream_t)stream, y, dy, inv_norm, mask, instr_len,
                     (uint16_t*)dhidden, S, H, mode, normalize);
  GRIT_CHECK_LAUNCH("grit_pool_norm_bwd");
  return GRIT_OK;
}

int grit_pool_norm_varlen_bwd(const float* y, const float* dy, const float* inv_norm, const int32_t* cu_seqlens, const int32_t* instr_len,
                              void* dhidden, int B, int H, int mode, int normalize, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(y && dy && cu_seqlens && dhidden, GRIT_E_BADARG, "grit_pool_norm_varlen_bwd: null pointer");
  GRIT_REQUIRE(!normalize || inv_norm, GRIT_E_BADARG, "grit_pool_norm_varlen_bwd: inv_norm required when normalize");
  GRIT_REQUIRE(B >= 0 && H > 0, GRIT_E_BADARG, "grit_pool_norm_varlen_bwd: bad sizes");
  GRIT_REQUIRE(mode >= GRIT_POOL_MEAN && mode <= GRIT_POOL_LASTTOKEN, GRIT_E_BADARG, "grit_pool_norm_varlen_bwd: unknown pooling mode %d", mode);
  GRIT_REQUIRE(H % 8 == 0 && H <= 32768, GRIT_E_UNSUPPORTED, "grit_pool_norm_varlen_bwd: H=%d must be a multiple of 8, <= 32768", H);
  GRIT_REQUIRE(aligned16(dhidden), GRIT_E_BADARG, "grit_pool_norm_varlen_bwd: pointers must be 16-byte aligned");
  const size_t lds = 4 * (size_t)H + 64;
  static std::atomic<uint64_t> optin_vb{0};
  lds_optin_once(pool_norm_varlen_bwd_k, optin_vb, 160 * 1024);
  hipLaunchKernelGGL(pool_norm_varlen_bwd_k, dim3(B), dim3(POOL_THREADS), lds, (hipStream_t)stream, y, dy, inv_norm, cu_seqlens, instr_len,
                     (uint16_t*)dhidden, H, mode, normalize);
  GRIT_CHECK_LAUNCH("grit_pool_norm_varlen_bwd");
  return GRIT_OK;
}
}  // extern "C"
