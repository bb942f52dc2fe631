// Token-by-token decode kernels (generation from cached document KV, rag/eval.py:237-302 -> model.generate with
// past_key_values).  At 1-8 rows per step every projection is an HBM-bound GEMV over the bf16 weights (14.5 GB per token at
// the 7B shape), attention is a read of the sequence's KV; nothing here uses MFMA.
//   gemv        out[b,n] = x[b,:] . W[n,:]  (+ residual), bf16 in/out, fp32 accumulate -- nn.Linear of q/k/v/o/down/lm_head
//   gemv_swiglu act[b,p] = silu(x.Wg[p]) * (x.Wu[p]) on the GRIT_EPI_SWIGLU weight layout (gate/up rows interleaved in blocks of 16)
//   kv_append   k,v of the new token (post-RoPE) -> cache[b, h, lens[b], :]
//   attn_decode one query row per (sequence, head) against cache[:, :lens[b]+1] (flash-decoding split over the keys) + combine
//   argmax      greedy next token (lowest index on ties) and lens[b] += 1
// Every kernel reads its dynamic sizes (lens) from DEVICE memory so that a whole step can be captured in one HIP graph.
#include "common.h"

namespace grit {

__device__ __forceinline__ float silu_d(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float dot8(const uint4 a, const uint4 b) {
  return bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) + bfhi(a.y) * bfhi(b.y) + bflo(a.z) * bflo(b.z) +
         bfhi(a.z) * bfhi(b.z) + bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
}

constexpr int GV_ROWS = 4;  // weight rows per wave (loads of 4 rows in flight per lane)

// MODE 0: store, 1: + residual, 2: SwiGLU pairs (rows r, r+16 of the interleaved layout)
template <int NB, int MODE>
__global__ void __launch_bounds__(256) gemv_bf16_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ W, uint16_t* __restrict__ out,
                                                   const uint16_t* __restrict__ res, int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo,
                                                   int64_t ldr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 4 + wave;                     // one unit = GV_ROWS weight rows
  int rows[GV_ROWS];
  if (MODE == 2) {
    // unit covers 2 (gate, up) pairs: pair p -> gate row (p/16)*32 + p%16, up row = gate row + 16
    const int p0 = unit * 2;
    if (p0 >= N / 2) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int p = p0 + i; if (p > N / 2 - 1) p = N / 2 - 1;
      rows[2 * i] = (p >> 4) * 32 + (p & 15);
      rows[2 * i + 1] = rows[2 * i] + 16;
    }
  } else {
    const int r0 = unit * GV_ROWS;
    if (r0 >= N) return;
#pragma unroll
    for (int i = 0; i < GV_ROWS; ++i) rows[i] = r0 + i < N ? r0 + i : N - 1;
  }
  float acc[GV_ROWS][NB];
#pragma unroll
  for (int i = 0; i < GV_ROWS; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
  const int KC = K >> 3;
  for (int c = lane; c < KC; c += 64) {
    uint4 wv[GV_ROWS];
#pragma unroll
    for (int i = 0; i < GV_ROWS; ++i) wv[i] = reinterpret_cast<const uint4*>(W + (int64_t)rows[i] * ldw)[c];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint4 xv = reinterpret_cast<const uint4*>(x + (int64_t)(b < B ? b : 0) * ldx)[c];
#pragma unroll
      for (int i = 0; i < GV_ROWS; ++i) acc[i][b] += dot8(wv[i], xv);
    }
  }
#pragma unroll
  for (int i = 0; i < GV_ROWS; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = wave_sum(acc[i][b]);
  if (lane != 0) return;
  if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = unit * 2 + i;
      if (p >= N / 2) break;
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (b < B) out[(int64_t)b * ldo + p] = (uint16_t)f2bf(round_bf(silu_d(round_bf(acc[2 * i][b]))) * round_bf(acc[2 * i + 1][b]));
    }
  } else {
#pragma unroll
    for (int i = 0; i < GV_ROWS; ++i) {
      const int n = unit * GV_ROWS + i;
      if (n >= N) break;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b >= B) continue;
        float v = acc[i][b];
        if (MODE == 1) v = round_bf(v) + bf2f(res[(int64_t)b * ldr + n]);
        out[(int64_t)b * ldo + n] = (uint16_t)f2bf(v);
      }
    }
  }
}

// ---- append the new token's k, v (already rotated) to the cache: cache [B, nkv, Lmax, d]
__global__ void __launch_bounds__(256) kv_append_k(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ck, uint16_t* __restrict__ cv,
                                                   const int32_t* __restrict__ lens, int nq, int nkv, int d, int Lmax, int64_t qkv_stride) {
  const int b = blockIdx.x;
  const int pos = lens[b];
  if (pos >= Lmax) return;
  const int per = nkv * d / 8;                  // 16-B chunks of k (and of v)
  for (int i = threadIdx.x; i < 2 * per; i += 256) {
    const int which = i / per, j = i - which * per;
    const int h = j / (d / 8), c = j - h * (d / 8);
    const uint4 v = reinterpret_cast<const uint4*>(qkv + (int64_t)b * qkv_stride + (int64_t)(nq + which * nkv + h) * d)[c];
    uint16_t* dst = (which ? cv : ck) + (((int64_t)b * nkv + h) * Lmax + pos) * d;
    reinterpret_cast<uint4*>(dst)[c] = v;
  }
}

// ---- decode attention, head_dim 128.  grid (splits, nkv, B), 256 threads = 4 waves x 64 keys = 256 keys per workgroup.
constexpr int AD_D = 128, AD_CH = 256, AD_G = 8;   // up to 8 query heads per kv head
__global__ void __launch_bounds__(256) attn_decode_k(const uint16_t* __restrict__ q, const uint16_t* __restrict__ ck,
                                                     const uint16_t* __restrict__ cv, const int32_t* __restrict__ lens, float* __restrict__ part,
                                                     int nq, int nkv, int Lmax, int64_t q_stride, float scale, int max_splits) {
  __shared__ float qs[AD_G][AD_D];           // query heads of this kv head, pre-scaled
  __shared__ float ps[4][AD_G][64];          // per wave: probabilities of its 64 keys
  __shared__ float ws_m[4][AD_G], ws_l[4][AD_G];
  __shared__ float wo[4][AD_G][AD_D];
  const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = nq / nkv;
  const int L = lens[b] + 1;                 // keys 0 .. lens[b] (the new token was appended)
  const int k0 = split * AD_CH;
  float* pbase = part + (((int64_t)b * nkv + hk) * max_splits + split) * G * (AD_D + 2);
  if (k0 >= L) {                             // empty split: neutral element
    for (int i = tid; i < G * (AD_D + 2); i += 256) pbase[i] = (i % (AD_D + 2) == 0) ? -INFINITY : 0.f;
    return;
  }
  for (int i = tid; i < G * AD_D; i += 256) {
    const int g = i / AD_D, e = i - g * AD_D;
    qs[g][e] = bf2f(q[(int64_t)b * q_stride + (int64_t)(hk * G + g) * AD_D + e]) * scale;
  }
  __syncthreads();
  const int key = k0 + wave * 64 + lane;
  const bool live = key < L;
  float s[AD_G];
#pragma unroll
  for (int g = 0; g < AD_G; ++g) s[g] = 0.f;
  {
    const uint4* kr = reinterpret_cast<const uint4*>(ck + (((int64_t)b * nkv + hk) * Lmax + (live ? key : L - 1)) * AD_D);
#pragma unroll 4
    for (int c = 0; c < AD_D / 8; ++c) {
      const uint4 kv = kr[c];
      const float kf[8] = {bflo(kv.x), bfhi(kv.x), bflo(kv.y), bfhi(kv.y), bflo(kv.z), bfhi(kv.z), bflo(kv.w), bfhi(kv.w)};
#pragma unroll
      for (int g = 0; g < AD_G; ++g) {
        if (g >= G) break;
        const float* qg = &qs[g][c * 8];
        s[g] += kf[0] * qg[0] + kf[1] * qg[1] + kf[2] * qg[2] + kf[3] * qg[3] + kf[4] * qg[4] + kf[5] * qg[5] + kf[6] * qg[6] + kf[7] * qg[7];
      }
    }
  }
#pragma unroll
  for (int g = 0; g < AD_G; ++g) {
    if (g >= G) break;
    const float sv = live ? s[g] : -INFINITY;
    const float m = wave_max(sv);
    const float p = live ? __expf(sv - m) : 0.f;
    const float l = wave_sum(p);
    ps[wave][g][lane] = p;
    if (lane == 0) { ws_m[wave][g] = m; ws_l[wave][g] = l; }
  }
  __syncthreads();
  // O partial of this wave: lane owns dims 2*lane, 2*lane+1
  float o[AD_G][2];
#pragma unroll
  for (int g = 0; g < AD_G; ++g) { o[g][0] = 0.f; o[g][1] = 0.f; }
  const int nk = min(64, L - (k0 + wave * 64));
  const uint32_t* vr = reinterpret_cast<const uint32_t*>(cv + (((int64_t)b * nkv + hk) * Lmax + k0 + wave * 64) * AD_D) + lane;
  for (int j = 0; j < nk; ++j) {
    const uint32_t vv = vr[(int64_t)j * (AD_D / 2)];
    const float v0 = bflo(vv), v1 = bfhi(vv);
#pragma unroll
    for (int g = 0; g < AD_G; ++g) {
      if (g >= G) break;
      const float p = ps[wave][g][j];
      o[g][0] += p * v0; o[g][1] += p * v1;
    }
  }
#pragma unroll
  for (int g = 0; g < AD_G; ++g) {
    if (g >= G) break;
    wo[wave][g][2 * lane] = o[g][0]; wo[wave][g][2 * lane + 1] = o[g][1];
  }
  __syncthreads();
  // combine the 4 waves -> one partial per (head): [m, l, o[128]]
  for (int i = tid; i < G * AD_D; i += 256) {
    const int g = i / AD_D, e = i - g * AD_D;
    float m = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) m = fmaxf(m, ws_m[w][g]);
    float l = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = ws_m[w][g] == -INFINITY ? 0.f : __expf(ws_m[w][g] - m);
      l += ws_l[w][g] * f; acc += wo[w][g][e] * f;
    }
    float* pg = pbase + g * (AD_D + 2);
    pg[2 + e] = acc;
    if (e == 0) { pg[0] = m; pg[1] = l; }
  }
}

__global__ void __launch_bounds__(128) attn_decode_combine_k(const float* __restrict__ part, uint16_t* __restrict__ out, int nq, int nkv,
                                                             int max_splits, int64_t out_stride) {
  const int h = blockIdx.x, b = blockIdx.y, e = threadIdx.x;
  const int G = nq / nkv, hk = h / G, g = h - hk * G;
  const float* base = part + ((int64_t)b * nkv + hk) * max_splits * G * (AD_D + 2) + g * (AD_D + 2);
  float m = -INFINITY;
  for (int s = 0; s < max_splits; ++s) m = fmaxf(m, base[(int64_t)s * G * (AD_D + 2)]);
  float l = 0.f, acc = 0.f;
  for (int s = 0; s < max_splits; ++s) {
    const float* ps_ = base + (int64_t)s * G * (AD_D + 2);
    const float f = ps_[0] == -INFINITY ? 0.f : __expf(ps_[0] - m);
    l += ps_[1] * f; acc += ps_[2 + e] * f;
  }
  out[(int64_t)b * out_stride + (int64_t)h * AD_D + e] = (uint16_t)f2bf(l > 0.f ? acc / l : 0.f);
}

// ---- greedy sampling + advance: next[b] = argmax_v logits[b, v] (lowest index on ties), lens[b] += 1
__global__ void __launch_bounds__(256) argmax_advance_k(const uint16_t* __restrict__ logits, int64_t ld, int V, int64_t* __restrict__ next,
                                                        int32_t* __restrict__ lens, int64_t* __restrict__ history, int64_t hist_stride,
                                                        const int32_t* __restrict__ step) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int v = tid; v < V; v += 256) {
    const float x = bf2f(logits[(int64_t)b * ld + v]);
    if (x > best) { best = x; idx = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
  }
  if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    next[b] = idx;
    if (history) history[(int64_t)b * hist_stride + step[0]] = idx;
    if (lens) lens[b] += 1;
  }
}

__global__ void bump_k(int32_t* v) { v[0] += 1; }

}  // namespace grit

using namespace grit;

template <int MODE>
static int launch_gemv(const void* x, const void* W, void* out, const void* res, int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo,
                       int64_t ldr, hipStream_t st) {
  const int units = MODE == 2 ? (N / 2 + 1) / 2 : (N + GV_ROWS - 1) / GV_ROWS;
  const dim3 grid((unsigned)((units + 3) / 4));
#define GRIT_GEMV(NB_)                                                                                                                   \
  hipLaunchKernelGGL((gemv_bf16_k<NB_, MODE>), grid, dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)W, (uint16_t*)out, (const uint16_t*)res, \
                     B, N, K, ldx, ldw, ldo, ldr)
  if (B == 1) GRIT_GEMV(1); else if (B == 2) GRIT_GEMV(2); else if (B <= 4) GRIT_GEMV(4); else GRIT_GEMV(8);
  GRIT_CHECK_LAUNCH("grit_gemv_bf16");
  return GRIT_OK;
}

extern "C" int grit_gemv_bf16(const void* x, const void* W, void* out, int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int epilogue,
                              const void* residual, int64_t ldr, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(x && W && out, GRIT_E_BADARG, "grit_gemv_bf16: null pointer");
  GRIT_REQUIRE(B > 0 && B <= 8, GRIT_E_UNSUPPORTED, "grit_gemv_bf16: B=%d rows (1..8; larger batches use grit_gemm_bf16_nt)", B);
  GRIT_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, GRIT_E_BADARG, "grit_gemv_bf16: bad sizes");
  GRIT_REQUIRE(aligned16(x) && aligned16(W), GRIT_E_BADARG, "grit_gemv_bf16: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE: GRIT_REQUIRE(ldo >= N, GRIT_E_BADARG, "grit_gemv_bf16: ldo < N");
      return launch_gemv<0>(x, W, out, nullptr, B, N, K, ldx, ldw, ldo, 0, st);
    case GRIT_EPI_RESIDUAL: GRIT_REQUIRE(residual && ldo >= N && ldr >= N, GRIT_E_BADARG, "grit_gemv_bf16: RESIDUAL needs residual, ldo, ldr >= N");
      return launch_gemv<1>(x, W, out, residual, B, N, K, ldx, ldw, ldo, ldr, st);
    case GRIT_EPI_SWIGLU: GRIT_REQUIRE(N % 32 == 0 && ldo >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemv_bf16: SWIGLU needs N %% 32 == 0, ldo >= N/2");
      return launch_gemv<2>(x, W, out, nullptr, B, N, K, ldx, ldw, ldo, 0, st);
    default: GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemv_bf16: unknown epilogue %d", epilogue);
  }
  return GRIT_OK;
}

extern "C" int grit_kv_append(const void* qkv, void* cache_k, void* cache_v, const int32_t* lens, int B, int nq, int nkv, int d, int Lmax,
                              int64_t qkv_stride, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(qkv && cache_k && cache_v && lens, GRIT_E_BADARG, "grit_kv_append: null pointer");
  GRIT_REQUIRE(B > 0 && nq > 0 && nkv > 0 && d % 8 == 0 && Lmax > 0 && qkv_stride % 8 == 0, GRIT_E_BADARG, "grit_kv_append: bad sizes");
  GRIT_REQUIRE(aligned16(qkv) && aligned16(cache_k) && aligned16(cache_v), GRIT_E_BADARG, "grit_kv_append: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(kv_append_k, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, (uint16_t*)cache_k, (uint16_t*)cache_v,
                     lens, nq, nkv, d, Lmax, qkv_stride);
  GRIT_CHECK_LAUNCH("grit_kv_append");
  return GRIT_OK;
}

extern "C" int64_t grit_attn_decode_workspace_floats(int B, int nq, int nkv, int Lmax) {
  const int splits = (Lmax + AD_CH - 1) / AD_CH;
  return (int64_t)B * nkv * splits * (nq / nkv) * (AD_D + 2);
}

extern "C" int grit_attn_decode(const void* q, const void* cache_k, const void* cache_v, const int32_t* lens, void* out, float* workspace, int B,
                                int nq, int nkv, int d, int Lmax, int64_t q_stride, int64_t out_stride, float scale, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(q && cache_k && cache_v && lens && out && workspace, GRIT_E_BADARG, "grit_attn_decode: null pointer");
  GRIT_REQUIRE(d == AD_D, GRIT_E_UNSUPPORTED, "grit_attn_decode: head_dim=%d (only 128 is built)", d);
  GRIT_REQUIRE(nq % nkv == 0 && nq / nkv <= AD_G, GRIT_E_UNSUPPORTED, "grit_attn_decode: %d query heads per kv head (max %d)", nq / nkv, AD_G);
  GRIT_REQUIRE(B > 0 && Lmax > 0 && B <= 65535 && nkv <= 65535, GRIT_E_BADARG, "grit_attn_decode: bad sizes");
  const int splits = (Lmax + AD_CH - 1) / AD_CH;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_decode_k, dim3((unsigned)splits, (unsigned)nkv, (unsigned)B), dim3(256), 0, st, (const uint16_t*)q, (const uint16_t*)cache_k,
                     (const uint16_t*)cache_v, lens, workspace, nq, nkv, Lmax, q_stride, scale, splits);
  GRIT_CHECK_LAUNCH("grit_attn_decode");
  hipLaunchKernelGGL(attn_decode_combine_k, dim3((unsigned)nq, (unsigned)B), dim3(AD_D), 0, st, (const float*)workspace, (uint16_t*)out, nq, nkv,
                     splits, out_stride);
  GRIT_CHECK_LAUNCH("grit_attn_decode: combine");
  return GRIT_OK;
}

extern "C" int grit_argmax_advance(const void* logits, int64_t ld, int V, int64_t* next, int32_t* lens, int64_t* history, int64_t hist_stride,
                                   int32_t* step, int B, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(logits && next, GRIT_E_BADARG, "grit_argmax_advance: null pointer");
  GRIT_REQUIRE(V > 0 && ld >= V && B > 0 && (!history || step), GRIT_E_BADARG, "grit_argmax_advance: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(argmax_advance_k, dim3((unsigned)B), dim3(256), 0, st, (const uint16_t*)logits, ld, V, next, lens, history, hist_stride, step);
  GRIT_CHECK_LAUNCH("grit_argmax_advance");
  if (step) {
    hipLaunchKernelGGL(bump_k, dim3(1), dim3(1), 0, st, step);
    GRIT_CHECK_LAUNCH("grit_argmax_advance: step");
  }
  return GRIT_OK;
}
