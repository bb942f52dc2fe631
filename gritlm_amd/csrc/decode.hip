// Token-by-token decode kernels (generation from cached document KV, rag/eval.py:237-302 -> model.generate with
// past_key_values).  At 1-8 rows per step every projection is an HBM-bound GEMV over the bf16 weights (14.5 GB per token at
// the 7B shape), attention is a read of the sequence's KV; nothing here uses MFMA.
//   gemv        out[b,n] = x[b,:] . W[n,:]  (+ residual), bf16 in/out, fp32 accumulate -- nn.Linear of q/k/v/o/down/lm_head
//   gemv_swiglu act[b,p] = silu(x.Wg[p]) * (x.Wu[p]) on the GRIT_EPI_SWIGLU weight layout (gate/up rows interleaved in blocks of 16)
//   kv_append   k,v of the new token (post-RoPE) -> cache[b, h, lens[b], :]
//   attn_decode one query row per (sequence, head) against cache[:, :lens[b]+1] (flash-decoding split over the keys) + combine
//   argmax      greedy next token (lowest index on ties) and lens[b] += 1
// Every kernel reads its dynamic sizes (lens) from DEVICE memory so that a whole step can be captured in one HIP graph.
//
// F16 (round 6): the same kernels on IEEE fp16 weights and an fp16 KV cache, around an fp32 residual stream -- the decode half of the
// encoder's "f16_operands" policy (the RAG flow continues from the fp16 K/V of encode(get_cache=True)).  Formats under F16: the stream h,
// the fused q|k|v row and the logits are fp32 (residual adds, the rotation of q and the argmax see unrounded values); every GEMV operand
// -- the stream's copy h16 in front of a norm, ctx, act -- is fp16, rounded once from fp32; K / V are rounded to fp16 once when they
// enter the cache.  Values beyond the fp16 range raise the per-device flag word every fp16 kernel of the library reports through (common.h).
#include "common.h"

namespace grit {


__device__ __forceinline__ float dot8(const uint4 a, const uint4 b) {
  return bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) + bfhi(a.y) * bfhi(b.y) + bflo(a.z) * bflo(b.z) +
         bfhi(a.z) * bfhi(b.z) + bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
}
// eight 16-bit values of the operand format -> fp32
template <bool F16>
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  f[0] = lo16_op<F16>(v.x); f[1] = hi16_op<F16>(v.x); f[2] = lo16_op<F16>(v.y); f[3] = hi16_op<F16>(v.y);
  f[4] = lo16_op<F16>(v.z); f[5] = hi16_op<F16>(v.z); f[6] = lo16_op<F16>(v.w); f[7] = hi16_op<F16>(v.w);
}
// two fp32 values -> one 32-bit word of the operand format, RNE: the bit-trick bf16 pack every RoPE kernel of the library uses, or fp16
template <bool F16>
__device__ __forceinline__ uint32_t pack2_rne(float lo, float hi) {
  if constexpr (F16) return pack2h_hw(lo, hi);
  else return pack2bf(lo, hi);
}
// one fp32 value -> the 16 bits of its fp16 rounding (RNE; beyond 65504: inf, reported by the caller)
__device__ __forceinline__ uint16_t f2h_bits(float f) { return (uint16_t)(pack2h_hw(f, 0.f) & 0xffffu); }

// All-reduce of NG independent values over the 64 lanes in registers: four row rotations by DPP (all-reduce inside every 16-lane row),
// then v_permlane16_swap / v_permlane32_swap for the rows (gfx950) -- no LDS-queue instruction.  (`wave_max` / `wave_sum` of common.h
// are six DEPENDENT ds_bpermute round trips each, ~3 us for the eight reductions of a GQA group of four when they run one after the
// other; here the NG chains are interleaved step by step.)  Sum order differs from wave_sum's: used only where no bit pattern is pinned.
template <int NG, bool IS_MAX>
__device__ __forceinline__ void wave_allreduce(float (&v)[NG]) {
  auto op = [](float a, float b) { return IS_MAX ? fmaxf(a, b) : a + b; };
#define GRIT_DPP_STEP(CTRL)                                                                                                      \
  _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                                                               \
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v[g]), (CTRL), 0xf, 0xf, false);                                 \
    v[g] = op(v[g], __int_as_float(t));                                                                                          \
  }
  GRIT_DPP_STEP(0x121)      // row_ror:1
  GRIT_DPP_STEP(0x122)      // row_ror:2
  GRIT_DPP_STEP(0x124)      // row_ror:4
  GRIT_DPP_STEP(0x128)      // row_ror:8
#undef GRIT_DPP_STEP
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[g]), __float_as_uint(v[g]), false, false);
    v[g] = op(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[g]), __float_as_uint(v[g]), false, false);
    v[g] = op(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
  }
}

#ifndef GRIT_GV_ROWS
#define GRIT_GV_ROWS 4      // (A/B builds: tools/decode_variants.sh)
#endif
constexpr int GV_ROWS = GRIT_GV_ROWS;  // weight rows per workgroup; its 4 waves each take a quarter of K (split-K, LDS reduce)
#ifndef GRIT_GV_ROWS_SWIGLU
#define GRIT_GV_ROWS_SWIGLU GRIT_GV_ROWS
#endif
constexpr int GV_ROWS_SW = GRIT_GV_ROWS_SWIGLU;   // the same for the gate|up launches (SwiGLU epilogue: rows come in (gate, up) pairs);
                                                  // 4 / 8 / 16 rows: 2.757 / 2.798 / 2.824 ms per token (profiles/r06_decode_f16_ab.log)

// MODE 0: store, 1: + residual, 2: SwiGLU pairs (rows r, r+16 of the interleaved layout).
// PRENORM 1: x is the raw residual stream and the kernel applies MistralRMSNorm on the fly (x_n = bf16(w_ln * bf16(x * rsqrt(mean x^2 + eps))),
// modeling_mistral_gritlm.py:84-89) -- saves the separate RMSNorm launch of a decode step (a 1-row kernel is pure launch latency), same bits
// as the two launches.  Its price: the rounding needs the row's RMS BEFORE the first product, so every wave of every workgroup first reads
// the whole row again (as much load traffic as the workgroup's weights) and reduces it.
// PRENORM 2 (round 5, "deferred"): out = rsqrt(mean x^2 + eps) * sum_k W[n,k] (x[k] w_ln[k]) -- the scale is applied to the finished dot
// product, the sum of squares is accumulated from the x pieces the lane loads for the product anyway (the four waves' split-K quarters
// cover the row exactly once) and reduced beside it: no second pass over x, no dependency in front of the weight stream.  x_n is never
// rounded to bf16 (the reference rounds it twice): one rounding fewer than the reference's arithmetic, not the same bits.
// F16: W and x in fp16; MODE 0 / 1 write fp32 (q|k|v row, logits / the stream: out = res + dot, nothing rounded), MODE 2 writes the fp16
// activation (one rounding of silu(g) * u).  MODE 1 additionally writes out16 = fp16(out): the operand copy of the stream the next
// norm + GEMV reads (an fp32 x would double the x traffic of the 7168 gate|up workgroups: measured +2.8 us per launch, 3 % of a token).
// PRENORM 1 (the exact bf16 form) has no fp16 counterpart.  ld* are in elements of the respective format.
template <int NB, int MODE, int PRENORM, int R, bool F16 = false, int WV = 4>
__global__ void __launch_bounds__(64 * WV) gemv_bf16_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ W, uint16_t* __restrict__ out,
                                                   const uint16_t* __restrict__ res, const uint16_t* __restrict__ ln_w, float eps, int B, int N,
                                                   int K, int64_t ldx, int64_t ldw, int64_t ldo, int64_t ldr, unsigned int* __restrict__ flag,
                                                   uint16_t* __restrict__ out16, int64_t ldo16, const int32_t* __restrict__ widx, int64_t w_estride) {
  static_assert(!(F16 && PRENORM == 1), "the exact fused norm rounds to bf16: bf16 only");
  // widx (nullable): W is a stack of matrices [E, N, K] and this launch multiplies by matrix widx[0] -- the expert a sparse-MoE decode step
  // routed the row to, read from DEVICE memory so that the step stays one HIP graph
  if (widx) W += (int64_t)widx[0] * w_estride;
  __shared__ float red[WV][R][NB];                            // WV waves per workgroup, each takes 1 / WV of K (split-K, LDS reduce)
  __shared__ float red_ss[WV][NB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x;                                // one unit = R weight rows
  int rows[R];
  if (MODE == 2) {
    // unit covers R/2 (gate, up) pairs: pair p -> gate row (p/16)*32 + p%16, up row = gate row + 16
    const int p0 = unit * (R / 2);
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      int p = p0 + i; if (p > N / 2 - 1) p = N / 2 - 1;
      rows[2 * i] = (p >> 4) * 32 + (p & 15);
      rows[2 * i + 1] = rows[2 * i] + 16;
    }
  } else {
    const int r0 = unit * R;
#pragma unroll
    for (int i = 0; i < R; ++i) rows[i] = r0 + i < N ? r0 + i : N - 1;
  }
  const int KC = K >> 3;
  // The weights are streamed ONCE per token and shared with nobody: non-temporal loads (no L2 / Infinity Cache allocation that would only
  // evict the activations and the KV cache; measured on this chip as the `nt-weights` row of the decode price list: issue -> landed
  // -18 %; here 3.59 -> 3.47 ms per token, profiles/r03_decode_ab.log).  The first weight pieces of the lane go out BEFORE the RMSNorm statistics are reduced: the HBM latency of the stream's
  // head overlaps the (L2-resident, serial) sum of squares instead of following it -- the 50 MB q|k|v GEMV is one workgroup wave deep.
  auto wload = [&](int i, int c) -> uint4 {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(W + (int64_t)rows[i] * ldw) + c);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };
  // eight x values of row b at chunk c, as fp32
  auto xload = [&](int b, int c, float (&xf)[8]) {
    unpack8<F16>(reinterpret_cast<const uint4*>(x + (int64_t)(b < B ? b : 0) * ldx)[c], xf);
  };
  const int c0 = wave * 64 + lane;
  uint4 wv[R];
  if (c0 < KC) {
#pragma unroll
    for (int i = 0; i < R; ++i) wv[i] = wload(i, c0);
  }
  float inv[NB];
  float ssq[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) ssq[b] = 0.f;
  if (PRENORM == 1) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float ss = 0.f;
      const uint4* xr = reinterpret_cast<const uint4*>(x + (int64_t)(b < B ? b : 0) * ldx);
      for (int c = lane; c < KC; c += 64) {
        const uint4 v = xr[c];
        ss += bflo(v.x) * bflo(v.x) + bfhi(v.x) * bfhi(v.x) + bflo(v.y) * bflo(v.y) + bfhi(v.y) * bfhi(v.y) + bflo(v.z) * bflo(v.z) +
              bfhi(v.z) * bfhi(v.z) + bflo(v.w) * bflo(v.w) + bfhi(v.w) * bfhi(v.w);
      }
      inv[b] = rsqrtf(wave_sum(ss) / (float)K + eps);
    }
  }
  float acc[R][NB];
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
  for (int c = c0; c < KC; c += 64 * WV) {
    // (a one-step software pipeline of the weight loads -- next iteration's pieces requested before this iteration's arithmetic -- was
    //  measured 8 % SLOWER at the 7B shape: 16 more registers per lane, fewer workgroups in flight; profiles/r03_decode_ab.log)
    if (c != c0) {
#pragma unroll
      for (int i = 0; i < R; ++i) wv[i] = wload(i, c);
    }
    uint4 lw = make_uint4(0, 0, 0, 0);
    if (PRENORM) lw = reinterpret_cast<const uint4*>(ln_w)[c];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (PRENORM == 2 || F16) {
        float xs[8];
        xload(b, c, xs);
        if constexpr (F16) {
          // The fp16-operand forms PIN their contraction (one product, then an fma chain): left to the compiler, the NB = 1 / 4 and the
          // NB = 2 / 8 instantiations of the deferred-norm form contracted differently and a row's bits depended on how many rows shared
          // the launch (tools/gemv_rows_probe.py) -- the prompt chunk (8 rows per launch) must reproduce the token-by-token prompt.
#pragma clang fp contract(off)
          if (PRENORM == 2) {
            const float lf[8] = {bflo(lw.x), bfhi(lw.x), bflo(lw.y), bfhi(lw.y), bflo(lw.z), bfhi(lw.z), bflo(lw.w), bfhi(lw.w)};   // (norm weights stay bf16)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ssq[b] = __builtin_fmaf(xs[e], xs[e], ssq[b]); xs[e] = xs[e] * lf[e]; }
          }
#pragma unroll
          for (int i = 0; i < R; ++i) {
            float wf[8];
            unpack8<true>(wv[i], wf);
            float t = wf[0] * xs[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) t = __builtin_fmaf(wf[e], xs[e], t);
            acc[i][b] = acc[i][b] + t;
          }
          continue;
        }
        if (PRENORM == 2) {
          const float lf[8] = {bflo(lw.x), bfhi(lw.x), bflo(lw.y), bfhi(lw.y), bflo(lw.z), bfhi(lw.z), bflo(lw.w), bfhi(lw.w)};   // (norm weights stay bf16)
#pragma unroll
          for (int e = 0; e < 8; ++e) { ssq[b] += xs[e] * xs[e]; xs[e] = xs[e] * lf[e]; }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const uint4 w = wv[i];
          acc[i][b] += lo16_op<F16>(w.x) * xs[0] + hi16_op<F16>(w.x) * xs[1] + lo16_op<F16>(w.y) * xs[2] + hi16_op<F16>(w.y) * xs[3] +
                       lo16_op<F16>(w.z) * xs[4] + hi16_op<F16>(w.z) * xs[5] + lo16_op<F16>(w.w) * xs[6] + hi16_op<F16>(w.w) * xs[7];
        }
        continue;
      }
      uint4 xv = reinterpret_cast<const uint4*>(x + (int64_t)(b < B ? b : 0) * ldx)[c];
      if (PRENORM == 1) {
        const float s_ = inv[b];
        xv.x = pack2bf(round_bf(bflo(xv.x) * s_) * bflo(lw.x), round_bf(bfhi(xv.x) * s_) * bfhi(lw.x));
        xv.y = pack2bf(round_bf(bflo(xv.y) * s_) * bflo(lw.y), round_bf(bfhi(xv.y) * s_) * bfhi(lw.y));
        xv.z = pack2bf(round_bf(bflo(xv.z) * s_) * bflo(lw.z), round_bf(bfhi(xv.z) * s_) * bfhi(lw.z));
        xv.w = pack2bf(round_bf(bflo(xv.w) * s_) * bflo(lw.w), round_bf(bfhi(xv.w) * s_) * bfhi(lw.w));
      }
#pragma unroll
      for (int i = 0; i < R; ++i) acc[i][b] += dot8(wv[i], xv);
    }
  }
  // the wave's R x NB (+ NB) partial sums are reduced TOGETHER in registers (wave_allreduce: DPP row rotations + permlane swaps, the chains
  // interleaved step by step) -- R x NB calls of wave_sum are as many chains of six dependent ds_bpermute round trips at the tail of a
  // kernel that is one workgroup wave deep
  {
    float rv[R * NB + (PRENORM == 2 ? NB : 0)];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) rv[i * NB + b] = acc[i][b];
    if (PRENORM == 2) {
#pragma unroll
      for (int b = 0; b < NB; ++b) rv[R * NB + b] = ssq[b];
    }
    wave_allreduce<R * NB + (PRENORM == 2 ? NB : 0), false>(rv);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < R; ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) red[wave][i][b] = rv[i * NB + b];
      if (PRENORM == 2) {
#pragma unroll
        for (int b = 0; b < NB; ++b) red_ss[wave][b] = rv[R * NB + b];
      }
    }
  }
  __syncthreads();
  // thread t < R*NB finishes output (row i, batch b)
  const int t = threadIdx.x;
  auto row_scale = [&](int b) -> float {
    if (PRENORM != 2) return 1.0f;
    float ss = red_ss[0][b];
#pragma unroll
    for (int w = 1; w < WV; ++w) ss += red_ss[w][b];
    return rsqrtf(ss / (float)K + eps);
  };
  auto total = [&](int i, int b) -> float {                   // the waves' partial sums, in wave order
    float v = red[0][i][b];
#pragma unroll
    for (int w = 1; w < WV; ++w) v += red[w][i][b];
    return v;
  };
  if (MODE == 2) {
    if (t < (R / 2) * NB) {
      const int i = t / NB, b = t - i * NB, p = unit * (R / 2) + i;
      if (p < N / 2 && b < B) {
        const float rs = row_scale(b);
        const float g = total(2 * i, b) * rs;
        const float u = total(2 * i + 1, b) * rs;
        if constexpr (F16) {
          const uint16_t hb = f2h_bits(silu_f(g) * u);
          if ((hb & 0x7c00u) == 0x7c00u) atomicOr(flag, 1u);
          out[(int64_t)b * ldo + p] = hb;
        } else {
          out[(int64_t)b * ldo + p] = (uint16_t)f2bf(round_bf(silu_f(round_bf(g))) * round_bf(u));
        }
      }
    }
  } else if (t < R * NB) {
    const int i = t / NB, b = t - i * NB, n = unit * R + i;
    if (n < N && b < B) {
      float v = total(i, b) * row_scale(b);
      if constexpr (F16) {
        if (MODE == 1) v += reinterpret_cast<const float*>(res)[(int64_t)b * ldr + n];
        reinterpret_cast<float*>(out)[(int64_t)b * ldo + n] = v;
        if (MODE == 1 && out16) {
          const uint16_t hb = f2h_bits(v);
          if ((hb & 0x7c00u) == 0x7c00u) atomicOr(flag, 1u);
          out16[(int64_t)b * ldo16 + n] = hb;
        }
      } else {
        if (MODE == 1) v = round_bf(v) + bf2f(res[(int64_t)b * ldr + n]);
        out[(int64_t)b * ldo + n] = (uint16_t)f2bf(v);
      }
    }
  }
}

// ---- RoPE of the new token's q, k (at position lens[b]) + append of its k, v to the cache, one launch.
//      qkv [B, qkv_stride] (q rotated in place), cache [B, nkv, Lmax, d], tables [Lmax, d/2] fp32 (rounded like the encoder's).
//      cache_row (nullable): row b of qkv belongs to sequence cache_row[b] of the cache -- several rows may be consecutive tokens of ONE
//      sequence (a prompt chunk on top of a cached prefix: lens[b] = prefix + its index in the chunk).
//      F16: the row is fp32 (q is rotated in place without a rounding), k (rotated) and v are rounded to fp16 once, into fp16 caches.
template <bool F16 = false>
__global__ void __launch_bounds__(256) rope_kv_append_k(uint16_t* __restrict__ qkv, const float* __restrict__ cos_tab, const float* __restrict__ sin_tab,
                                                        uint16_t* __restrict__ ck, uint16_t* __restrict__ cv, const int32_t* __restrict__ lens,
                                                        const int32_t* __restrict__ cache_row, int nq, int nkv, int d, int Lmax, int64_t qkv_stride,
                                                        unsigned int* __restrict__ flag) {
  const int b = blockIdx.x;
  const int pos = lens[b];
  if (pos >= Lmax) return;
  const int cb = cache_row ? cache_row[b] : b;
  const int half = d >> 1, jc_n = d >> 4;               // chunks of 8 elements per half head
  uint16_t* row = qkv + (int64_t)b * qkv_stride;
  float* rowf = reinterpret_cast<float*>(qkv) + (int64_t)b * qkv_stride;
  const int rot_items = (nq + nkv) * jc_n, cp_items = nkv * (d >> 3);
  uint32_t bad = 0;
  for (int i = threadIdx.x; i < rot_items + cp_items; i += 256) {
    if (i < rot_items) {
      const int head = i / jc_n, jc = i - head * jc_n;
      const int64_t off = (int64_t)head * d + jc * 8;
      const float* ct = cos_tab + (int64_t)pos * half + jc * 8;
      const float* stb = sin_tab + (int64_t)pos * half + jc * 8;
      float a[8], bb[8], lo[8], hi[8];
      if constexpr (F16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = rowf[off + e]; bb[e] = rowf[off + half + e]; }
      } else {
        unpack8<false>(*reinterpret_cast<const uint4*>(row + off), a);
        unpack8<false>(*reinterpret_cast<const uint4*>(row + off + half), bb);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lo[e] = rope_lo(a[e], bb[e], ct[e], stb[e]);       // x*cos + rotate_half(x)*sin, first half: -x2 * sin
        hi[e] = rope_hi(a[e], bb[e], ct[e], stb[e]);       // second half: +x1 * sin
      }
      if (F16 && head < nq) {                              // q: rotated in place, fp32
#pragma unroll
        for (int e = 0; e < 8; ++e) { rowf[off + e] = lo[e]; rowf[off + half + e] = hi[e]; }
        continue;
      }
      const uint4 r1 = make_uint4(pack2_rne<F16>(lo[0], lo[1]), pack2_rne<F16>(lo[2], lo[3]), pack2_rne<F16>(lo[4], lo[5]), pack2_rne<F16>(lo[6], lo[7]));
      const uint4 r2 = make_uint4(pack2_rne<F16>(hi[0], hi[1]), pack2_rne<F16>(hi[2], hi[3]), pack2_rne<F16>(hi[4], hi[5]), pack2_rne<F16>(hi[6], hi[7]));
      if (head < nq) {
        *reinterpret_cast<uint4*>(row + off) = r1; *reinterpret_cast<uint4*>(row + off + half) = r2;
      } else {
        if constexpr (F16) bad |= h2_nonfinite(r1.x) | h2_nonfinite(r1.y) | h2_nonfinite(r1.z) | h2_nonfinite(r1.w) | h2_nonfinite(r2.x) |
                                  h2_nonfinite(r2.y) | h2_nonfinite(r2.z) | h2_nonfinite(r2.w);
        uint16_t* dst = ck + (((int64_t)cb * nkv + (head - nq)) * Lmax + pos) * d + jc * 8;
        *reinterpret_cast<uint4*>(dst) = r1; *reinterpret_cast<uint4*>(dst + half) = r2;
      }
    } else {
      const int j = i - rot_items, h = j / (d >> 3), c = j - h * (d >> 3);
      uint4 v;
      if constexpr (F16) {
        const float* vs = rowf + (int64_t)(nq + nkv + h) * d + c * 8;
        v = make_uint4(pack2h_hw(vs[0], vs[1]), pack2h_hw(vs[2], vs[3]), pack2h_hw(vs[4], vs[5]), pack2h_hw(vs[6], vs[7]));
        bad |= h2_nonfinite(v.x) | h2_nonfinite(v.y) | h2_nonfinite(v.z) | h2_nonfinite(v.w);
      } else {
        v = reinterpret_cast<const uint4*>(row + (int64_t)(nq + nkv + h) * d)[c];
      }
      reinterpret_cast<uint4*>(cv + (((int64_t)cb * nkv + h) * Lmax + pos) * d)[c] = v;
    }
  }
  if constexpr (F16) {
    if (bad) atomicOr(flag, 1u);
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

// ---- decode attention, head_dim 128.  grid (splits of 64 keys, nkv, B), ONE wave per workgroup: at 1-8 sequences the work is tiny, what
//      matters is that every 64-key slice of the cache is read by its own wave somewhere on the chip (2 k keys x 8 kv heads = 264 waves).
constexpr int AD_D = 128, AD_CH = 64, AD_G = 8;   // up to 8 query heads per kv head
// ROPE: q arrives un-rotated in the fused qkv row; every workgroup rotates its G query heads itself (position lens[b]) and the one whose
// slice contains that position also rotates the new k, appends k and v to the cache and then reads them back like any other key --
// the RoPE + KV-append launch of a decode step disappears.
// G = query heads per kv head, a TEMPLATE argument (round 5): with the run-time G of rounds 1-4 every loop over the heads carried an
// `if (g >= G) break`, the compiler could neither unroll nor interleave the heads' dot products and reductions, and the one wave of a
// workgroup executed them back to back -- the kernel was bound by its own dependent-instruction chains (11.4 us per layer at L = 2 k
// for 32 KB of K / V per workgroup), not by memory.
// PH ("per head", G = 1): one workgroup per QUERY head instead of per kv head -- blockIdx.y is the query head, the gq heads of a kv group
// are gq workgroups that read the same K / V slice (the duplicates hit L2) and each does a quarter of the arithmetic: the kernel is one
// wave deep and bound by its own instruction stream (a slice of 64 keys x 4 heads is ~2000 dependent-free FMAs per lane), not by the
// 32 KB it reads.  Every one of them rotates the new key itself; they write identical cache rows.
// F16: the q (ROPE: q|k|v) row is fp32 -- q is rotated and scaled without a rounding, the new k / v are rounded to fp16 once, on their way
// into the cache -- and the cache is fp16.
template <bool ROPE, int G, bool PH = false, bool F16 = false>
__global__ void __launch_bounds__(64) attn_decode_k(const uint16_t* __restrict__ q, uint16_t* __restrict__ ck, uint16_t* __restrict__ cv,
                                                    const int32_t* __restrict__ lens, float* __restrict__ part, const float* __restrict__ cos_tab,
                                                    const float* __restrict__ sin_tab, int nq, int nkv, int Lmax, int64_t q_stride, float scale,
                                                    int max_splits, unsigned int* __restrict__ flag, const int32_t* __restrict__ cache_row) {
  __shared__ __attribute__((aligned(16))) float qs[G][AD_D];   // query heads of this kv head, pre-scaled
  __shared__ float ps[G][64];                                  // probabilities of the 64 keys
  __shared__ __attribute__((aligned(16))) uint16_t newk[AD_D]; // ROPE, owner workgroup: the new key (rotated) and value rows, handed to the
  __shared__ __attribute__((aligned(16))) uint32_t newv[AD_D / 2];   // lanes that hold them in the K / V register layouts
  const int gq = PH ? nq / nkv : G;                       // query heads per kv head in the q row / the partials' layout
  const int split = blockIdx.x, hk = PH ? (int)blockIdx.y / gq : (int)blockIdx.y, b = blockIdx.z;
  const int g0 = PH ? (int)blockIdx.y - hk * gq : 0;      // this workgroup's first (PH: only) head inside the group
  const int lane = threadIdx.x;
  const int cb = cache_row ? cache_row[b] : b;   // the sequence whose cache row b attends to (several rows may be tokens of one sequence)
  const int L = lens[b] + 1;                 // keys 0 .. lens[b] (the new token was appended)
  const int k0 = split * AD_CH;
  float* pbase = part + ((((int64_t)b * nkv + hk) * max_splits + split) * gq + g0) * (AD_D + 2);
  if (k0 >= L) {                             // empty split: neutral element
    for (int i = lane; i < G * (AD_D + 2); i += 64) pbase[i] = (i % (AD_D + 2) == 0) ? -INFINITY : 0.f;
    return;
  }
  // The slice's K and V rows are requested FIRST: 32 independent 16-byte loads per lane go out before the query heads are fetched,
  // rotated and staged -- also in the one workgroup whose slice receives the NEW key (round 5: it used to append first and load
  // afterwards, two memory latencies in a row on the kernel's critical path): its loads of the new key's cache row return whatever the
  // slot held, and the rotated k / the v row replace those registers through LDS after the barrier.
  const int key = k0 + lane;
  const bool live = key < L;
  const int kg = lane >> 4, dc = lane & 15;                  // PV layout: key group kg (16 keys), dim chunk dc (8 dims)
  const int nk = min(64, L - k0);
  uint4 kreg[AD_D / 8], vreg[16];
  auto load_kv = [&]() {
    const uint4* kr = reinterpret_cast<const uint4*>(ck + (((int64_t)cb * nkv + hk) * Lmax + (live ? key : L - 1)) * AD_D);
#pragma unroll
    for (int c = 0; c < AD_D / 8; ++c) kreg[c] = kr[c];       // 16 independent loads in flight
    const uint16_t* vbase = cv + (((int64_t)cb * nkv + hk) * Lmax + k0) * AD_D;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int kk = kg * 16 + j;
      vreg[j] = reinterpret_cast<const uint4*>(vbase + (int64_t)(kk < nk ? kk : nk - 1) * AD_D)[dc];
    }
  };
  const bool owner = ROPE && (L - 1 >= k0) && (L - 1 < k0 + AD_CH);      // workgroup-uniform
  load_kv();
  if constexpr (ROPE) {
    const int pos = L - 1;
    const float c = cos_tab[(int64_t)pos * 64 + lane], sn = sin_tab[(int64_t)pos * 64 + lane];     // lane <-> pair (e, e + 64)
    const uint16_t* row = q + (int64_t)b * q_stride;
    const float* rowf = reinterpret_cast<const float*>(q) + (int64_t)b * q_stride;       // F16: the fp32 q|k|v row
    auto qk_at = [&](int64_t i) -> float { if constexpr (F16) return rowf[i]; else return bf2f(row[i]); };
    float x1[G], x2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {                             // all 2 G loads in flight before the first use
      x1[g] = qk_at((int64_t)(hk * gq + g0 + g) * AD_D + lane); x2[g] = qk_at((int64_t)(hk * gq + g0 + g) * AD_D + 64 + lane);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if constexpr (F16) {
        qs[g][lane] = rope_lo(x1[g], x2[g], c, sn) * scale; qs[g][64 + lane] = rope_hi(x1[g], x2[g], c, sn) * scale;
      } else {
        const uint32_t r = pack2bf(rope_lo(x1[g], x2[g], c, sn), rope_hi(x1[g], x2[g], c, sn));         // one rounding, as rope_k
        qs[g][lane] = bflo(r) * scale; qs[g][64 + lane] = bfhi(r) * scale;
      }
    }
    if (owner) {                                          // this workgroup owns the new key: rotate k, append k and v
      const int64_t ko = (int64_t)(nq + hk) * AD_D, vo = (int64_t)(nq + nkv + hk) * AD_D;
      const float y1 = qk_at(ko + lane), y2 = qk_at(ko + 64 + lane);
      const uint32_t r = F16 ? pack2h_hw(rope_lo(y1, y2, c, sn), rope_hi(y1, y2, c, sn)) : pack2bf(rope_lo(y1, y2, c, sn), rope_hi(y1, y2, c, sn));
      uint16_t* kd = ck + (((int64_t)cb * nkv + hk) * Lmax + pos) * AD_D;
      kd[lane] = (uint16_t)(r & 0xffff); kd[64 + lane] = (uint16_t)(r >> 16);
      uint32_t vw;
      if constexpr (F16) {
        vw = pack2h_hw(rowf[vo + 2 * lane], rowf[vo + 2 * lane + 1]);
        if (h2_nonfinite(r) | h2_nonfinite(vw)) atomicOr(flag, 1u);
      } else {
        vw = reinterpret_cast<const uint32_t*>(row + vo)[lane];
      }
      reinterpret_cast<uint32_t*>(cv + (((int64_t)cb * nkv + hk) * Lmax + pos) * AD_D)[lane] = vw;
      newk[lane] = (uint16_t)(r & 0xffff); newk[64 + lane] = (uint16_t)(r >> 16);
      newv[lane] = vw;
    }
  } else {
    for (int i = lane; i < G * AD_D; i += 64) {
      const int g = i / AD_D, e = i - g * AD_D;
      const int64_t qi = (int64_t)b * q_stride + (int64_t)(hk * gq + g0 + g) * AD_D + e;
      qs[g][e] = (F16 ? reinterpret_cast<const float*>(q)[qi] : bf2f(q[qi])) * scale;
    }
  }
  __syncthreads();
  if (owner) {                                                 // the new key's rows, from LDS into the register layouts
    const int lk = L - 1 - k0;                                 // its index in the slice
    if (lane == lk) {
#pragma unroll
      for (int c = 0; c < AD_D / 8; ++c) kreg[c] = reinterpret_cast<const uint4*>(newk)[c];
    }
    const uint4 nv = reinterpret_cast<const uint4*>(newv)[dc];
#pragma unroll
    for (int j = 0; j < 16; ++j)                               // the new key is the slice's last one: every clamped row index (keys past
      if (kg * 16 + j >= lk) vreg[j] = nv;                     // the sequence, probability 0) pointed at its slot as well
  }
  float s[G];
#pragma unroll
  for (int g = 0; g < G; ++g) s[g] = 0.f;
#pragma unroll
  for (int c = 0; c < AD_D / 8; ++c) {
    float kf[8];
    unpack8<F16>(kreg[c], kf);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 qa = *reinterpret_cast<const float4*>(&qs[g][c * 8]), qb = *reinterpret_cast<const float4*>(&qs[g][c * 8 + 4]);
      s[g] += kf[0] * qa.x + kf[1] * qa.y + kf[2] * qa.z + kf[3] * qa.w + kf[4] * qb.x + kf[5] * qb.y + kf[6] * qb.z + kf[7] * qb.w;
    }
  }
  float mg[G], lg[G];
#pragma unroll
  for (int g = 0; g < G; ++g) mg[g] = live ? s[g] : -INFINITY;
  wave_allreduce<G, true>(mg);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float p = live ? __expf(s[g] - mg[g]) : 0.f;
    lg[g] = p;
    ps[g][lane] = p;
  }
  wave_allreduce<G, false>(lg);
  __syncthreads();
  // O partial: lane = (key group kg = lane>>4, dim chunk dc = lane&15): 16 INDEPENDENT 16-byte loads per lane (keys kg*16 .. +15, dims
  // 8dc .. +7; requested at the top of the kernel), then the 4 key groups are folded through the two row swaps
  float o[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float vf[8];
    unpack8<F16>(vreg[j], vf);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float p = ps[g][kg * 16 + j];            // 0 for keys past the sequence
#pragma unroll
      for (int e = 0; e < 8; ++e) o[g][e] += p * vf[e];
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {                    // fold the four key groups (rows of 16 lanes): same dims sit 16 / 32 lanes apart
      const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[g][e]), __float_as_uint(o[g][e]), false, false);
      const float t = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
      const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
      o[g][e] = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
    }
    float* pg = pbase + g * (AD_D + 2);
    if (kg == 0) {
      *reinterpret_cast<float4*>(pg + 2 + 8 * dc) = make_float4(o[g][0], o[g][1], o[g][2], o[g][3]);
      *reinterpret_cast<float4*>(pg + 2 + 8 * dc + 4) = make_float4(o[g][4], o[g][5], o[g][6], o[g][7]);
    }
    if (lane == 0) { pg[0] = mg[g]; pg[1] = lg[g]; }
  }
}

template <bool F16 = false>
__global__ void __launch_bounds__(128) attn_decode_combine_k(const float* __restrict__ part, uint16_t* __restrict__ out, int nq, int nkv,
                                                             int max_splits, int64_t out_stride) {
  // (round 5) every wave derives the global maximum and the denominator for itself, in registers: the (m_s, l_s) pairs of all splits
  // are fetched in ONE pass (two loads per split and lane, all in flight), reduced with the register all-reduce above -- one memory
  // round trip and one barrier (the rescale factors go through LDS) instead of three of each
  __shared__ float fs[2][512];               // per wave: exp(m_s - m) of every split
  const int h = blockIdx.x, b = blockIdx.y, e = threadIdx.x, lane = e & 63, wave = e >> 6;
  const int G = nq / nkv, hk = h / G, g = h - hk * G;
  const int64_t sstride = (int64_t)G * (AD_D + 2);
  const float* base = part + ((int64_t)b * nkv + hk) * max_splits * sstride + g * (AD_D + 2);
  float ms[8], ls[8];                        // max_splits <= 512 = 8 per lane
  // the partial outputs of the first 64 splits (L <= 4096: all of them) are requested NOW, beside the (m_s, l_s) pairs: they do not
  // depend on the rescale factors, so the kernel pays one memory round trip instead of two
  float pv[64];
#pragma unroll
  for (int sp = 0; sp < 64; ++sp) pv[sp] = sp < max_splits ? base[sp * sstride + 2 + e] : 0.f;
  float m[1] = {-INFINITY};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int sp = lane + 64 * i;
    ms[i] = sp < max_splits ? base[sp * sstride] : -INFINITY;
    ls[i] = sp < max_splits ? base[sp * sstride + 1] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) m[0] = fmaxf(m[0], ms[i]);
  wave_allreduce<1, true>(m);
  float l[1] = {0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int sp = lane + 64 * i;
    const float f = ms[i] == -INFINITY ? 0.f : __expf(ms[i] - m[0]);
    if (sp < max_splits) fs[wave][sp] = f;
    l[0] += ls[i] * f;
  }
  wave_allreduce<1, false>(l);
  __syncthreads();
  // thread e owns output dim e; the loads of different splits are independent
  float acc = 0.f;
#pragma unroll
  for (int sp = 0; sp < 64; ++sp) acc += pv[sp] * (sp < max_splits ? fs[wave][sp] : 0.f);      // same order as the loop below
#pragma unroll 8
  for (int sp = 64; sp < max_splits; ++sp) acc += base[sp * sstride + 2 + e] * fs[wave][sp];
  const float o = l[0] > 0.f ? acc / l[0] : 0.f;                    // a convex combination of cached values: inside their range
  out[(int64_t)b * out_stride + (int64_t)h * AD_D + e] = F16 ? f2h_bits(o) : (uint16_t)f2bf(o);
}

// ---- greedy sampling + advance: next[b] = argmax_v logits[b, v] (lowest index on ties), lens[b] += 1
// L32: fp32 logits (the fp16-operand decode step keeps them unrounded)
template <bool L32 = false>
__global__ void __launch_bounds__(256) argmax_advance_k(const uint16_t* __restrict__ logits, int64_t ld, int V, int64_t* __restrict__ next,
                                                        int32_t* __restrict__ lens, int64_t* __restrict__ history, int64_t hist_stride,
                                                        const int32_t* __restrict__ step) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  const uint16_t* row = logits + (int64_t)b * ld;
  const float* rowf = reinterpret_cast<const float*>(logits) + (int64_t)b * ld;
  const int VC = V >> 3;                                      // chunks of 8 (ld % 8 == 0 keeps the rows aligned)
  for (int c = tid; c < VC; c += 256) {
    float x[8];
    if constexpr (L32) {
      const float4 u = reinterpret_cast<const float4*>(rowf)[2 * c], w = reinterpret_cast<const float4*>(rowf)[2 * c + 1];
      x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = w.x; x[5] = w.y; x[6] = w.z; x[7] = w.w;
    } else {
      unpack8<false>(reinterpret_cast<const uint4*>(row)[c], x);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (x[e] > best) { best = x[e]; idx = c * 8 + e; }      // ascending index inside a thread: first maximum wins
  }
  for (int v = (VC << 3) + tid; v < V; v += 256) {
    const float x = L32 ? rowf[v] : bf2f(row[v]);
    if (x > best || (x == best && v < idx)) { best = x; idx = v; }
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

// Sparse-MoE decode on fp16 operands: h[b] += w[b,0] y[2b] + w[b,1] y[2b+1] in fp32 (y: the two chosen experts' fp32 outputs of row b) and
// h16 = fp16(h), the operand copy the next norm + GEMV reads -- one launch for the four elementwise passes a host-side combine would take.
__global__ void __launch_bounds__(256) moe_decode_combine_f32_k(float* __restrict__ h, uint16_t* __restrict__ h16, const float* __restrict__ y,
                                                                const float* __restrict__ w, int H, unsigned int* __restrict__ flag) {
  const int b = blockIdx.y;
  const float w0 = w[2 * b], w1 = w[2 * b + 1];
  uint32_t bad = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256) {
    const float v = h[(int64_t)b * H + i] + (w0 * y[(int64_t)(2 * b) * H + i] + w1 * y[(int64_t)(2 * b + 1) * H + i]);
    h[(int64_t)b * H + i] = v;
    const uint16_t hb = f2h_bits(v);
    bad |= ((hb & 0x7c00u) == 0x7c00u);
    h16[(int64_t)b * H + i] = hb;
  }
  if (bad) atomicOr(flag, 1u);
}

}  // namespace grit

using namespace grit;

template <int MODE, int PRENORM, bool F16 = false, int R = (MODE == 2 ? GV_ROWS_SW : GV_ROWS)>
static int launch_gemv(const void* x, const void* W, void* out, const void* res, const void* ln_w, float eps, int B, int N, int K, int64_t ldx,
                       int64_t ldw, int64_t ldo, int64_t ldr, hipStream_t st, void* out16 = nullptr, int64_t ldo16 = 0,
                       const int32_t* widx = nullptr, int64_t w_estride = 0) {
  const int units = MODE == 2 ? (N / 2 + R / 2 - 1) / (R / 2) : (N + R - 1) / R;
  const dim3 grid((unsigned)units);
  unsigned int* flag = F16 ? f16_flag_ptr() : nullptr;
  if (F16 && !flag) return GRIT_E_LAUNCH;
#define GRIT_GEMV_W(NB_, WV_)                                                                                                             \
  hipLaunchKernelGGL((gemv_bf16_k<NB_, MODE, PRENORM, R, F16, WV_>), grid, dim3(64 * WV_), 0, st, (const uint16_t*)x, (const uint16_t*)W,       \
                     (uint16_t*)out, (const uint16_t*)res, (const uint16_t*)ln_w, eps, B, N, K, ldx, ldw, ldo, ldr, flag, (uint16_t*)out16, ldo16,   \
                     widx, w_estride)
  // GRIT_GV_WAVES_SHORT_K (A/B builds): workgroups of 8 waves where K <= 4096 (q|k|v, o_proj: every lane then covers its share of the row in
  // ONE iteration -- one memory latency instead of two -- at twice the requests in flight per workgroup); 1- and 2-row steps only.
  // Measured (profiles/r06_decode_f16_ab.log, block 4): 2.754 -> 2.747 ms per token on bf16, 2.766 -> 2.743 on fp16 operands: not the default.
#ifdef GRIT_GV_WAVES_SHORT_K
#define GRIT_GEMV(NB_) do { if (K <= 4096 && NB_ <= 2 && PRENORM != 1) GRIT_GEMV_W(NB_, GRIT_GV_WAVES_SHORT_K); else GRIT_GEMV_W(NB_, 4); } while (0)
#else
#define GRIT_GEMV(NB_) GRIT_GEMV_W(NB_, 4)
#endif
  if (B == 1) GRIT_GEMV(1); else if (B == 2) GRIT_GEMV(2); else if (B <= 4) GRIT_GEMV(4); else GRIT_GEMV(8);
#undef GRIT_GEMV
#undef GRIT_GEMV_W
  GRIT_CHECK_LAUNCH(F16 ? "grit_gemv_f16" : "grit_gemv_bf16");
  return GRIT_OK;
}

// the fp16-operand forms (formats in the kernel's header comment); ln_w: the deferred norm
static int gemv_entry_f16(const char* name, const void* x, const void* W, void* out, const void* ln_w, float eps, int B, int N, int K, int64_t ldx,
                          int64_t ldw, int64_t ldo, int epilogue, const void* residual, int64_t ldr, void* out16, int64_t ldo16, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(x && W && out, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && B <= 8, GRIT_E_UNSUPPORTED, "%s: B=%d rows (1..8; larger batches use grit_gemm_f16_nt)", name, B);
  GRIT_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(aligned16(x) && aligned16(W) && (!ln_w || aligned16(ln_w)), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  hipStream_t st = (hipStream_t)stream;
  const bool pn = ln_w != nullptr;
  switch (epilogue) {
    case GRIT_EPI_STORE: GRIT_REQUIRE(ldo >= N, GRIT_E_BADARG, "%s: ldo < N", name);
      return pn ? launch_gemv<0, 2, true>(x, W, out, nullptr, ln_w, eps, B, N, K, ldx, ldw, ldo, 0, st)
                : launch_gemv<0, 0, true>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st);
    case GRIT_EPI_RESIDUAL: GRIT_REQUIRE(residual && ldo >= N && ldr >= N && !pn, GRIT_E_BADARG, "%s: RESIDUAL needs residual, ldo, ldr >= N (no pre-norm)", name);
      GRIT_REQUIRE(!out16 || ldo16 >= N, GRIT_E_BADARG, "%s: ldo16 < N", name);
      return launch_gemv<1, 0, true>(x, W, out, residual, nullptr, 0.f, B, N, K, ldx, ldw, ldo, ldr, st, out16, ldo16);
    case GRIT_EPI_SWIGLU: GRIT_REQUIRE(N % 32 == 0 && ldo >= N / 2, GRIT_E_UNSUPPORTED, "%s: SWIGLU needs N %% 32 == 0, ldo >= N/2", name);
      return pn ? launch_gemv<2, 2, true>(x, W, out, nullptr, ln_w, eps, B, N, K, ldx, ldw, ldo, 0, st)
                : launch_gemv<2, 0, true>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st);
    default: GRIT_REQUIRE(false, GRIT_E_BADARG, "%s: unknown epilogue %d", name, epilogue);
  }
  return GRIT_OK;
}

static int gemv_entry(const char* name, const void* x, const void* W, void* out, const void* ln_w, float eps, int B, int N, int K, int64_t ldx,
                      int64_t ldw, int64_t ldo, int epilogue, const void* residual, int64_t ldr, void* stream, bool deferred = false) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(x && W && out, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && B <= 8, GRIT_E_UNSUPPORTED, "%s: B=%d rows (1..8; larger batches use grit_gemm_bf16_nt)", name, B);
  GRIT_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(aligned16(x) && aligned16(W) && (!ln_w || aligned16(ln_w)), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  hipStream_t st = (hipStream_t)stream;
  const bool pn = ln_w != nullptr;
  switch (epilogue) {
    case GRIT_EPI_STORE: GRIT_REQUIRE(ldo >= N, GRIT_E_BADARG, "%s: ldo < N", name);
      return pn ? (deferred ? launch_gemv<0, 2>(x, W, out, nullptr, ln_w, eps, B, N, K, ldx, ldw, ldo, 0, st)
                            : launch_gemv<0, 1>(x, W, out, nullptr, ln_w, eps, B, N, K, ldx, ldw, ldo, 0, st))
                : launch_gemv<0, 0>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st);
    case GRIT_EPI_RESIDUAL: GRIT_REQUIRE(residual && ldo >= N && ldr >= N && !pn, GRIT_E_BADARG, "%s: RESIDUAL needs residual, ldo, ldr >= N (no pre-norm)", name);
      // (rows per workgroup of these launches -- o_proj, down: N = 4096 -> 1024 workgroups of 4 rows, half a workgroup wave -- and the
      //  number of K-loop pieces a lane keeps in flight were A/B'd on one box: 1 / 2 / 4 rows 2.914 / 2.902 / 2.903 ms per token; 2 / 4
      //  pieces in flight 2.963 / 3.139 against 2.921 -- more requests in flight are SLOWER; profiles/r05_decode_{rows,unroll}_ab.log)
      return launch_gemv<1, 0>(x, W, out, residual, nullptr, 0.f, B, N, K, ldx, ldw, ldo, ldr, st);
    case GRIT_EPI_SWIGLU: GRIT_REQUIRE(N % 32 == 0 && ldo >= N / 2, GRIT_E_UNSUPPORTED, "%s: SWIGLU needs N %% 32 == 0, ldo >= N/2", name);
      return pn ? (deferred ? launch_gemv<2, 2>(x, W, out, nullptr, ln_w, eps, B, N, K, ldx, ldw, ldo, 0, st)
                            : launch_gemv<2, 1>(x, W, out, nullptr, ln_w, eps, B, N, K, ldx, ldw, ldo, 0, st))
                : launch_gemv<2, 0>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st);
    default: GRIT_REQUIRE(false, GRIT_E_BADARG, "%s: unknown epilogue %d", name, epilogue);
  }
  return GRIT_OK;
}

extern "C" int grit_gemv_bf16(const void* x, const void* W, void* out, int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int epilogue,
                              const void* residual, int64_t ldr, void* stream) {
  return gemv_entry("grit_gemv_bf16", x, W, out, nullptr, 0.f, B, N, K, ldx, ldw, ldo, epilogue, residual, ldr, stream);
}

extern "C" int grit_rmsnorm_gemv_bf16(const void* x, const void* ln_weight, float eps, const void* W, void* out, int B, int N, int K, int64_t ldx,
                                      int64_t ldw, int64_t ldo, int epilogue, void* stream) {
  GRIT_REQUIRE(ln_weight, GRIT_E_BADARG, "grit_rmsnorm_gemv_bf16: null pointer");
  return gemv_entry("grit_rmsnorm_gemv_bf16", x, W, out, ln_weight, eps, B, N, K, ldx, ldw, ldo, epilogue, nullptr, 0, stream);
}

extern "C" int grit_gemv_f16(const void* x, const void* W, void* out, int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int epilogue,
                             const void* residual, int64_t ldr, void* out16, int64_t ldo16, void* stream) {
  GRIT_REQUIRE(!out16 || epilogue == GRIT_EPI_RESIDUAL, GRIT_E_BADARG, "grit_gemv_f16: out16 goes with the RESIDUAL epilogue");
  return gemv_entry_f16("grit_gemv_f16", x, W, out, nullptr, 0.f, B, N, K, ldx, ldw, ldo, epilogue, residual, ldr, out16, ldo16, stream);
}

extern "C" int grit_rmsnorm_gemv_f16_deferred(const void* x, const void* ln_weight, float eps, const void* W, void* out, int B, int N, int K,
                                              int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream) {
  GRIT_REQUIRE(ln_weight, GRIT_E_BADARG, "grit_rmsnorm_gemv_f16_deferred: null pointer");
  GRIT_REQUIRE(epilogue != GRIT_EPI_RESIDUAL, GRIT_E_BADARG, "grit_rmsnorm_gemv_f16_deferred: STORE or SWIGLU");
  return gemv_entry_f16("grit_rmsnorm_gemv_f16_deferred", x, W, out, ln_weight, eps, B, N, K, ldx, ldw, ldo, epilogue, nullptr, 0, nullptr, 0, stream);
}

// Sparse-MoE decode (ABI 5): x [B,K] times ONE matrix of a stack W [E,N,K] -- matrix expert[0], an index in DEVICE memory (the router's
// choice for the row; w_expert_stride = elements between consecutive matrices) -- with the STORE or SWIGLU epilogue.  f16: the fp16-operand
// formats of grit_gemv_f16 (STORE writes fp32, SWIGLU fp16).
static int gemv_expert_entry(const char* name, bool f16, const void* x, const void* W, void* out, const int32_t* expert, int64_t w_expert_stride,
                             int B, int N, int K, int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(x && W && out && expert, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && B <= 8, GRIT_E_UNSUPPORTED, "%s: B=%d rows (1..8)", name, B);
  GRIT_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K && w_expert_stride % 8 == 0 &&
               ldw <= w_expert_stride / N, GRIT_E_BADARG, "%s: bad sizes", name);       // (stride >= N * ldw, without the product: fuzzed ldw overflows it)
  GRIT_REQUIRE(aligned16(x) && aligned16(W), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE: GRIT_REQUIRE(ldo >= N, GRIT_E_BADARG, "%s: ldo < N", name);
      return f16 ? launch_gemv<0, 0, true>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st, nullptr, 0, expert, w_expert_stride)
                 : launch_gemv<0, 0>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st, nullptr, 0, expert, w_expert_stride);
    case GRIT_EPI_SWIGLU: GRIT_REQUIRE(N % 32 == 0 && ldo >= N / 2, GRIT_E_UNSUPPORTED, "%s: SWIGLU needs N %% 32 == 0, ldo >= N/2", name);
      return f16 ? launch_gemv<2, 0, true>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st, nullptr, 0, expert, w_expert_stride)
                 : launch_gemv<2, 0>(x, W, out, nullptr, nullptr, 0.f, B, N, K, ldx, ldw, ldo, 0, st, nullptr, 0, expert, w_expert_stride);
    default: GRIT_REQUIRE(false, GRIT_E_BADARG, "%s: epilogue %d (STORE or SWIGLU)", name, epilogue);
  }
  return GRIT_OK;
}

extern "C" int grit_gemv_bf16_expert(const void* x, const void* W, void* out, const int32_t* expert, int64_t w_expert_stride, int B, int N, int K,
                                     int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream) {
  return gemv_expert_entry("grit_gemv_bf16_expert", false, x, W, out, expert, w_expert_stride, B, N, K, ldx, ldw, ldo, epilogue, stream);
}

extern "C" int grit_gemv_f16_expert(const void* x, const void* W, void* out, const int32_t* expert, int64_t w_expert_stride, int B, int N, int K,
                                    int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream) {
  return gemv_expert_entry("grit_gemv_f16_expert", true, x, W, out, expert, w_expert_stride, B, N, K, ldx, ldw, ldo, epilogue, stream);
}

// the DEFERRED form of the fused norm (PRENORM 2 above): one launch, no second pass over x; x_n is not rounded to bf16
extern "C" int grit_rmsnorm_gemv_bf16_deferred(const void* x, const void* ln_weight, float eps, const void* W, void* out, int B, int N, int K,
                                               int64_t ldx, int64_t ldw, int64_t ldo, int epilogue, void* stream) {
  GRIT_REQUIRE(ln_weight, GRIT_E_BADARG, "grit_rmsnorm_gemv_bf16_deferred: null pointer");
  return gemv_entry("grit_rmsnorm_gemv_bf16_deferred", x, W, out, ln_weight, eps, B, N, K, ldx, ldw, ldo, epilogue, nullptr, 0, stream, true);
}

template <bool F16>
static int rope_kv_append_launch(const char* name, void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v,
                                 const int32_t* lens, const int32_t* cache_row, int B, int nq, int nkv, int d, int Lmax, int64_t qkv_stride,
                                 void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(qkv && cos_tab && sin_tab && cache_k && cache_v && lens, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && nq > 0 && nkv > 0 && d % 16 == 0 && Lmax > 0 && qkv_stride % 8 == 0, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(qkv_stride >= ((int64_t)nq + 2 * (int64_t)nkv) * d, GRIT_E_BADARG, "%s: qkv_stride too small", name);
  GRIT_REQUIRE(aligned16(qkv) && aligned16(cache_k) && aligned16(cache_v), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  unsigned int* flag = F16 ? f16_flag_ptr() : nullptr;
  if (F16 && !flag) return GRIT_E_LAUNCH;
  hipLaunchKernelGGL(rope_kv_append_k<F16>, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkv, cos_tab, sin_tab, (uint16_t*)cache_k,
                     (uint16_t*)cache_v, lens, cache_row, nq, nkv, d, Lmax, qkv_stride, flag);
  GRIT_CHECK_LAUNCH(name);
  return GRIT_OK;
}

extern "C" int grit_rope_kv_append(void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v, const int32_t* lens, int B,
                                   int nq, int nkv, int d, int Lmax, int64_t qkv_stride, void* stream) {
  return rope_kv_append_launch<false>("grit_rope_kv_append", qkv, cos_tab, sin_tab, cache_k, cache_v, lens, nullptr, B, nq, nkv, d, Lmax, qkv_stride,
                                      stream);
}

// Prompt chunk on top of a cached prefix (ABI 5): V rows that are tokens of B <= V sequences -- row v belongs to sequence cache_row[v] and sits
// at position lens[v] of it (consecutive tokens of one sequence: lens = prefix, prefix + 1, ...).  First every row's k / v is appended
// (grit_rope_kv_append_rows), then every row attends to keys 0 .. lens[v] of ITS sequence (grit_attn_decode_rows): causal attention over
// the prefix and the chunk without a token-by-token loop.  f16 != 0: the fp16-operand formats (fp32 q|k|v rows, fp16 caches and ctx).
extern "C" int grit_rope_kv_append_rows(void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v, const int32_t* lens,
                                        const int32_t* cache_row, int V, int nq, int nkv, int d, int Lmax, int64_t qkv_stride, int f16,
                                        void* stream) {
  GRIT_REQUIRE(cache_row, GRIT_E_BADARG, "grit_rope_kv_append_rows: null pointer");
  return f16 ? rope_kv_append_launch<true>("grit_rope_kv_append_rows", qkv, cos_tab, sin_tab, cache_k, cache_v, lens, cache_row, V, nq, nkv, d, Lmax,
                                           qkv_stride, stream)
             : rope_kv_append_launch<false>("grit_rope_kv_append_rows", qkv, cos_tab, sin_tab, cache_k, cache_v, lens, cache_row, V, nq, nkv, d, Lmax,
                                            qkv_stride, stream);
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
  // a size query has no error channel: 0 for sizes the compute call rejects
  if (B <= 0 || nq <= 0 || nkv <= 0 || Lmax <= 0 || B > 65535 || nkv > 65535 || nq > 65535 || Lmax > 512 * AD_CH) return 0;
  const int64_t splits = ((int64_t)Lmax + AD_CH - 1) / AD_CH;
  return (int64_t)B * nkv * splits * (nq / nkv) * (AD_D + 2);
}

template <bool F16>
static int attn_decode_launch(const char* name, const void* q, void* cache_k, void* cache_v, const int32_t* lens, void* out, float* workspace,
                              const float* cos_tab, const float* sin_tab, int B, int nq, int nkv, int d, int Lmax, int64_t q_stride,
                              int64_t out_stride, float scale, void* stream, const int32_t* cache_row = nullptr) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(q && cache_k && cache_v && lens && out && workspace, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(d == AD_D, GRIT_E_UNSUPPORTED, "%s: head_dim=%d (only 128 is built)", name, d);
  GRIT_REQUIRE(B > 0 && Lmax > 0 && nq > 0 && nkv > 0 && B <= 65535 && nkv <= 65535 && nq <= 65535, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(nq % nkv == 0 && nq / nkv <= AD_G, GRIT_E_UNSUPPORTED, "%s: %d query heads per kv head (max %d)", name, nq / nkv, AD_G);
  GRIT_REQUIRE(Lmax <= 512 * AD_CH, GRIT_E_UNSUPPORTED, "%s: Lmax=%d > %d", name, Lmax, 512 * AD_CH);
  const int splits = (Lmax + AD_CH - 1) / AD_CH;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)splits, (unsigned)nkv, (unsigned)B);
  unsigned int* flag = F16 ? f16_flag_ptr() : nullptr;
  if (F16 && !flag) return GRIT_E_LAUNCH;
  // one workgroup per query head (PH above) for GQA models: 8.74 -> 7.52 us per launch at L = 2 k, 32 / 8 heads (kernel trace,
  // profiles/r05_decode_kernel_stats_perhead.csv); GRIT_ATTN_DECODE_PER_HEAD=0 is the A/B knob
  static const int per_head = getenv("GRIT_ATTN_DECODE_PER_HEAD") ? atoi(getenv("GRIT_ATTN_DECODE_PER_HEAD")) : 1;
  if (per_head && nq / nkv > 1) {
    const dim3 gridh((unsigned)splits, (unsigned)nq, (unsigned)B);
    if (cos_tab)
      hipLaunchKernelGGL((attn_decode_k<true, 1, true, F16>), gridh, dim3(64), 0, st, (const uint16_t*)q, (uint16_t*)cache_k, (uint16_t*)cache_v, lens, workspace,
                         cos_tab, sin_tab, nq, nkv, Lmax, q_stride, scale, splits, flag, cache_row);
    else
      hipLaunchKernelGGL((attn_decode_k<false, 1, true, F16>), gridh, dim3(64), 0, st, (const uint16_t*)q, (uint16_t*)cache_k, (uint16_t*)cache_v, lens, workspace,
                         cos_tab, sin_tab, nq, nkv, Lmax, q_stride, scale, splits, flag, cache_row);
  } else {
#define GRIT_AD_LAUNCH(R, GG)                                                                                                           \
  hipLaunchKernelGGL((attn_decode_k<R, GG, false, F16>), grid, dim3(64), 0, st, (const uint16_t*)q, (uint16_t*)cache_k, (uint16_t*)cache_v, lens, workspace, \
                     cos_tab, sin_tab, nq, nkv, Lmax, q_stride, scale, splits, flag, cache_row)
#define GRIT_AD_BY_G(R)                                                                                                                 \
  switch (nq / nkv) {                                                                                                                   \
    case 1: GRIT_AD_LAUNCH(R, 1); break;                                                                                                \
    case 2: GRIT_AD_LAUNCH(R, 2); break;                                                                                                \
    case 4: GRIT_AD_LAUNCH(R, 4); break;                                                                                                \
    case 8: GRIT_AD_LAUNCH(R, 8); break;                                                                                                \
    default: GRIT_REQUIRE(false, GRIT_E_UNSUPPORTED, "%s: %d query heads per kv head (1, 2, 4 or 8 are built)", name, nq / nkv);       \
  }
  if (cos_tab) { GRIT_AD_BY_G(true) } else { GRIT_AD_BY_G(false) }
#undef GRIT_AD_BY_G
#undef GRIT_AD_LAUNCH
  }
  GRIT_CHECK_LAUNCH(name);
  hipLaunchKernelGGL(attn_decode_combine_k<F16>, dim3((unsigned)nq, (unsigned)B), dim3(AD_D), 0, st, (const float*)workspace, (uint16_t*)out, nq, nkv,
                     splits, out_stride);
  GRIT_CHECK_LAUNCH(name);
  return GRIT_OK;
}

extern "C" int grit_attn_decode(const void* q, const void* cache_k, const void* cache_v, const int32_t* lens, void* out, float* workspace, int B,
                                int nq, int nkv, int d, int Lmax, int64_t q_stride, int64_t out_stride, float scale, void* stream) {
  return attn_decode_launch<false>("grit_attn_decode", q, (void*)cache_k, (void*)cache_v, lens, out, workspace, nullptr, nullptr, B, nq, nkv, d, Lmax,
                            q_stride, out_stride, scale, stream);
}

extern "C" int grit_attn_decode_rows(const void* q, const void* cache_k, const void* cache_v, const int32_t* lens, const int32_t* cache_row, void* out,
                                     float* workspace, int V, int nq, int nkv, int d, int Lmax, int64_t q_stride, int64_t out_stride, float scale,
                                     int f16, void* stream) {
  GRIT_REQUIRE(cache_row, GRIT_E_BADARG, "grit_attn_decode_rows: null pointer");
  return f16 ? attn_decode_launch<true>("grit_attn_decode_rows", q, (void*)cache_k, (void*)cache_v, lens, out, workspace, nullptr, nullptr, V, nq, nkv, d,
                                        Lmax, q_stride, out_stride, scale, stream, cache_row)
             : attn_decode_launch<false>("grit_attn_decode_rows", q, (void*)cache_k, (void*)cache_v, lens, out, workspace, nullptr, nullptr, V, nq, nkv, d,
                                         Lmax, q_stride, out_stride, scale, stream, cache_row);
}

extern "C" int grit_attn_decode_rope(const void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v, const int32_t* lens,
                                     void* out, float* workspace, int B, int nq, int nkv, int d, int Lmax, int64_t qkv_stride, int64_t out_stride,
                                     float scale, void* stream) {
  GRIT_REQUIRE(cos_tab && sin_tab, GRIT_E_BADARG, "grit_attn_decode_rope: null pointer");
  GRIT_REQUIRE(nq > 0 && nkv > 0 && nq <= 65535 && nkv <= 65535 && d > 0 && d <= 65535, GRIT_E_BADARG, "grit_attn_decode_rope: bad sizes");
  GRIT_REQUIRE(qkv_stride >= ((int64_t)nq + 2 * (int64_t)nkv) * d, GRIT_E_BADARG, "grit_attn_decode_rope: qkv_stride too small");
  return attn_decode_launch<false>("grit_attn_decode_rope", qkv, cache_k, cache_v, lens, out, workspace, cos_tab, sin_tab, B, nq, nkv, d, Lmax, qkv_stride,
                                   out_stride, scale, stream);
}

// fp16-operand form: qkv is the fp32 fused projection row of the new token, the caches and out (ctx) are fp16
extern "C" int grit_attn_decode_rope_f16(const void* qkv, const float* cos_tab, const float* sin_tab, void* cache_k, void* cache_v, const int32_t* lens,
                                         void* out, float* workspace, int B, int nq, int nkv, int d, int Lmax, int64_t qkv_stride,
                                         int64_t out_stride, float scale, void* stream) {
  GRIT_REQUIRE(cos_tab && sin_tab, GRIT_E_BADARG, "grit_attn_decode_rope_f16: null pointer");
  GRIT_REQUIRE(nq > 0 && nkv > 0 && nq <= 65535 && nkv <= 65535 && d > 0 && d <= 65535, GRIT_E_BADARG, "grit_attn_decode_rope_f16: bad sizes");
  GRIT_REQUIRE(qkv_stride >= ((int64_t)nq + 2 * (int64_t)nkv) * d, GRIT_E_BADARG, "grit_attn_decode_rope_f16: qkv_stride too small");
  return attn_decode_launch<true>("grit_attn_decode_rope_f16", qkv, cache_k, cache_v, lens, out, workspace, cos_tab, sin_tab, B, nq, nkv, d, Lmax,
                                  qkv_stride, out_stride, scale, stream);
}

template <bool L32>
static int argmax_launch(const char* name, const void* logits, int64_t ld, int V, int64_t* next, int32_t* lens, int64_t* history, int64_t hist_stride,
                         int32_t* step, int B, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(logits && next, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(V > 0 && ld >= V && ld % 8 == 0 && B > 0 && (!history || step), GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(aligned16(logits), GRIT_E_BADARG, "%s: logits must be 16-byte aligned", name);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(argmax_advance_k<L32>, dim3((unsigned)B), dim3(256), 0, st, (const uint16_t*)logits, ld, V, next, lens, history, hist_stride, step);
  GRIT_CHECK_LAUNCH(name);
  if (step) {
    hipLaunchKernelGGL(bump_k, dim3(1), dim3(1), 0, st, step);
    GRIT_CHECK_LAUNCH(name);
  }
  return GRIT_OK;
}

extern "C" int grit_argmax_advance(const void* logits, int64_t ld, int V, int64_t* next, int32_t* lens, int64_t* history, int64_t hist_stride,
                                   int32_t* step, int B, void* stream) {
  return argmax_launch<false>("grit_argmax_advance", logits, ld, V, next, lens, history, hist_stride, step, B, stream);
}

extern "C" int grit_moe_decode_combine_f32(float* h, void* h16, const float* y, const float* weights, int B, int H, void* stream) {
  if (B == 0) return GRIT_OK;
  GRIT_REQUIRE(h && h16 && y && weights, GRIT_E_BADARG, "grit_moe_decode_combine_f32: null pointer");
  GRIT_REQUIRE(B > 0 && B <= 65535 && H > 0, GRIT_E_BADARG, "grit_moe_decode_combine_f32: bad sizes");
  unsigned int* flag = f16_flag_ptr();
  if (!flag) return GRIT_E_LAUNCH;
  hipLaunchKernelGGL(moe_decode_combine_f32_k, dim3((unsigned)((H + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, h, (uint16_t*)h16, y,
                     weights, H, flag);
  GRIT_CHECK_LAUNCH("grit_moe_decode_combine_f32");
  return GRIT_OK;
}

// fp32 logits (the fp16-operand decode step)
extern "C" int grit_argmax_advance_f32(const void* logits, int64_t ld, int V, int64_t* next, int32_t* lens, int64_t* history, int64_t hist_stride,
                                       int32_t* step, int B, void* stream) {
  return argmax_launch<true>("grit_argmax_advance_f32", logits, ld, V, next, lens, history, hist_stride, step, B, stream);
}
