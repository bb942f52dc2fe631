// Sparse-MoE plumbing around the grouped GEMM (Mixtral's MixtralSparseMoeBlock, scripts/modeling_mixtral_gritlm.py:839-882):
//   router  : gate Linear (bf16-rounded logits) -> softmax fp32 -> top-2 -> renormalise -> bf16     (:843-849)
//   index   : stable counting sort of the (token, k) pairs by expert -> per-expert row ranges        (replaces the per-expert
//             torch.where + .tolist() host round trips of :859-870; counts stay on the device)
//   combine : out = residual + (w_a * y_a  (+)  w_b * y_b) with the reference's bf16 rounding points (:876, :880, decoder :945)
// The expert MLPs themselves are two launches of grit_gemm_bf16_nt_grouped (gathered A rows, SwiGLU epilogue; then w2).
// All three kernels are HBM/latency-bound byte work; no MFMA.
#include <atomic>

#include "common.h"

namespace grit {

constexpr int MOE_MAX_E = 16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// ---- router: one wave per token, gate weights staged in LDS ([E,H] bf16)
template <int E>
__global__ void __launch_bounds__(256) moe_router_top2_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gate_w, int64_t T,
                                                         int H, int32_t* __restrict__ experts, float* __restrict__ weights) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* gw = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HC = H >> 3;
  for (int i = tid; i < E * HC; i += 256) gw[i] = reinterpret_cast<const uint4*>(gate_w)[i];
  __syncthreads();
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < T; t += (int64_t)gridDim.x * 4) {
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    const uint4* xr = reinterpret_cast<const uint4*>(x) + t * HC;
    for (int c0 = lane; c0 < HC; c0 += 8 * 64) {
      // eight 16-byte loads of the token row in flight per lane before the first use (H = 4096: the whole row)
      uint4 xv8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv8[u] = (c0 + 64 * u < HC) ? xr[c0 + 64 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 64 * u;
        if (c >= HC) break;
        const uint4 xv = xv8[u];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          // v_dot2c_f32_bf16: two bf16 products accumulated in fp32 per instruction, no unpacking (the kernel was VALU-bound on bflo/bfhi + fma)
          const uint4 wv = gw[e * HC + c];
          float a_ = acc[e];
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.x), __builtin_bit_cast(bf16x2_t, wv.x), a_, false);
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.y), __builtin_bit_cast(bf16x2_t, wv.y), a_, false);
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.z), __builtin_bit_cast(bf16x2_t, wv.z), a_, false);
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.w), __builtin_bit_cast(bf16x2_t, wv.w), a_, false);
          acc[e] = a_;
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < E; ++e) { acc[e] = round_bf(wave_sum(acc[e])); mx = fmaxf(mx, acc[e]); }   // nn.Linear output in the model dtype
    float p[E], den = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { p[e] = expf(acc[e] - mx); den += p[e]; }
    int e0 = 0, e1 = -1;
    float p0 = -1.f, p1 = -1.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {                    // descending, lowest index first on ties (torch.topk on equal values)
      const float pe = p[e] / den;
      if (pe > p0) { p1 = p0; e1 = e0; p0 = pe; e0 = e; }
      else if (pe > p1) { p1 = pe; e1 = e; }
    }
    if (lane == 0) {
      const float s = p0 + p1;
      experts[2 * t] = e0; experts[2 * t + 1] = e1;
      weights[2 * t] = round_bf(p0 / s); weights[2 * t + 1] = round_bf(p1 / s);
    }
  }
}

// ---- router of the "f16_operands" policy (fp32 residual stream): NOTHING is rounded -- what the reference computes when the model runs in
//      fp32 (scripts/modeling_mixtral_gritlm.py:843-849 with hidden_states fp32): logits = RMSNorm(h) Wg^T, softmax, top-2, renormalise.
//      The kernel reads the residual stream h itself (fp32) and folds the post-attention RMSNorm in (deferred form: the row scale
//      rsqrt(mean h^2 + eps) multiplies the finished dot products of (h * w_ln) with the gate rows), so the routing decision does not see
//      the fp16 rounding of the expert GEMMs' A operand.  One wave per token, gate [E,H] + w_ln [H] (bf16, exact) staged in LDS.
template <int E>
__global__ void __launch_bounds__(256) moe_router_top2_f32_k(const float* __restrict__ h, const uint16_t* __restrict__ ln_w, float eps,
                                                             const uint16_t* __restrict__ gate_w, int64_t T, int H,
                                                             int32_t* __restrict__ experts, float* __restrict__ weights) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint2* gw = reinterpret_cast<uint2*>(smem);                   // [E][H/4] : 4 bf16 per entry
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HQ = H >> 2;
  uint2* lw = gw + (size_t)E * HQ;                              // [H/4]
  for (int i = tid; i < E * HQ; i += 256) gw[i] = reinterpret_cast<const uint2*>(gate_w)[i];
  for (int i = tid; i < HQ; i += 256) lw[i] = reinterpret_cast<const uint2*>(ln_w)[i];
  __syncthreads();
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < T; t += (int64_t)gridDim.x * 4) {
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    float ss = 0.f;
    const float4* xr = reinterpret_cast<const float4*>(h) + t * HQ;
    for (int c0 = lane; c0 < HQ; c0 += 8 * 64) {
      float4 xv8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv8[u] = (c0 + 64 * u < HQ) ? xr[c0 + 64 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 64 * u;
        if (c >= HQ) break;
        const float4 xv = xv8[u];
        const uint2 l = lw[c];
        ss += xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
        const float x0 = xv.x * bflo(l.x), x1 = xv.y * bfhi(l.x), x2 = xv.z * bflo(l.y), x3 = xv.w * bfhi(l.y);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const uint2 g = gw[e * HQ + c];
          acc[e] += x0 * bflo(g.x) + x1 * bfhi(g.x) + x2 * bflo(g.y) + x3 * bfhi(g.y);
        }
      }
    }
    const float inv = rsqrtf(wave_sum(ss) / (float)H + eps);
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < E; ++e) { acc[e] = wave_sum(acc[e]) * inv; mx = fmaxf(mx, acc[e]); }
    float p[E], den = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { p[e] = expf(acc[e] - mx); den += p[e]; }
    int e0 = 0, e1 = -1;
    float p0 = -1.f, p1 = -1.f;
#pragma unroll
    for (int e = 0; e < E; ++e) {                    // descending, lowest index first on ties (torch.topk on equal values)
      const float pe = p[e] / den;
      if (pe > p0) { p1 = p0; e1 = e0; p0 = pe; e0 = e; }
      else if (pe > p1) { p1 = pe; e1 = e; }
    }
    if (lane == 0) {
      const float s = p0 + p1;
      experts[2 * t] = e0; experts[2 * t + 1] = e1;
      weights[2 * t] = p0 / s; weights[2 * t + 1] = p1 / s;
    }
  }
}

// ---- router backward (training; scripts/modeling_mixtral_gritlm.py:843-849 differentiated): w = renormalised top-2 of softmax(x Wg^T).
//      One wave per token, gate matrix in LDS (the forward router's structure): the wave recomputes the token's E logits in fp32, forms
//        p = softmax(logits);  s = p[e0] + p[e1];  w_k = p[e_k] / s;  dsel_k = (dw_k - sum_j dw_j w_j) / s;  inner = sum_k dsel_k p[e_k]
//        dlogits[e] = p[e] * ((e == e_k ? dsel_k : 0) - inner)  (+ aux_dlogits[t, e]: the auxiliary load-balancing loss's pull)
//      writes the [T, E] fp32 gradient of the logits and, in the same pass over the row,
//        dx_out[t, :] = bf16(f32(dx_in[t, :]) + sum_e dlogits[e] * Wg[e, :])          (the gate Linear's input gradient joins the experts')
//      HBM-bound byte work: x read, dx_in read, dx_out written, once each.
template <int E>
__global__ void __launch_bounds__(256) moe_router_bwd_k(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gate_w,
                                                        const int32_t* __restrict__ experts, const float* __restrict__ dw,
                                                        const float* __restrict__ aux, const uint16_t* __restrict__ dx_in,
                                                        uint16_t* __restrict__ dx_out, float* __restrict__ dlogits, int64_t T, int H) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* gw = reinterpret_cast<uint4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HC = H >> 3;
  for (int i = tid; i < E * HC; i += 256) gw[i] = reinterpret_cast<const uint4*>(gate_w)[i];
  __syncthreads();
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < T; t += (int64_t)gridDim.x * 4) {
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    const uint4* xr = reinterpret_cast<const uint4*>(x) + t * HC;
    for (int c0 = lane; c0 < HC; c0 += 8 * 64) {
      uint4 xv8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv8[u] = (c0 + 64 * u < HC) ? xr[c0 + 64 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 64 * u;
        if (c >= HC) break;
        const uint4 xv = xv8[u];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const uint4 wv = gw[e * HC + c];
          float a_ = acc[e];
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.x), __builtin_bit_cast(bf16x2_t, wv.x), a_, false);
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.y), __builtin_bit_cast(bf16x2_t, wv.y), a_, false);
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.z), __builtin_bit_cast(bf16x2_t, wv.z), a_, false);
          a_ = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, xv.w), __builtin_bit_cast(bf16x2_t, wv.w), a_, false);
          acc[e] = a_;
        }
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < E; ++e) { acc[e] = wave_sum(acc[e]); mx = fmaxf(mx, acc[e]); }      // fp32 logits (the backward's recompute keeps them unrounded)
    float p[E], den = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { p[e] = expf(acc[e] - mx); den += p[e]; }
    const float rden = 1.0f / den;
    const int e0 = experts[2 * t], e1 = experts[2 * t + 1];
    const float dw0 = dw[2 * t], dw1 = dw[2 * t + 1];
    float pe0 = 0.f, pe1 = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { p[e] *= rden; pe0 = (e == e0) ? p[e] : pe0; pe1 = (e == e1) ? p[e] : pe1; }
    const float ssel = pe0 + pe1;
    const float g = (dw0 * pe0 + dw1 * pe1) / ssel;                 // sum_j dw_j w_j
    const float ds0 = (dw0 - g) / ssel, ds1 = (dw1 - g) / ssel;
    const float inner = ds0 * pe0 + ds1 * pe1;
    float dl[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float dpe = (e == e0 ? ds0 : 0.f) + (e == e1 ? ds1 : 0.f);
      dl[e] = p[e] * (dpe - inner) + (aux != nullptr ? aux[t * E + e] : 0.f);
    }
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) dlogits[t * E + e] = dl[e];
    }
    // dx_out = dx_in + dlogits @ Wg
    const uint4* dir = dx_in != nullptr ? reinterpret_cast<const uint4*>(dx_in) + t * HC : nullptr;
    uint4* dor = reinterpret_cast<uint4*>(dx_out) + t * HC;
    for (int c = lane; c < HC; c += 64) {
      const uint4 dv = dir != nullptr ? dir[c] : make_uint4(0, 0, 0, 0);
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint4 wv = gw[e * HC + c];
        o[0] = fmaf(dl[e], bflo(wv.x), o[0]); o[1] = fmaf(dl[e], bfhi(wv.x), o[1]);
        o[2] = fmaf(dl[e], bflo(wv.y), o[2]); o[3] = fmaf(dl[e], bfhi(wv.y), o[3]);
        o[4] = fmaf(dl[e], bflo(wv.z), o[4]); o[5] = fmaf(dl[e], bfhi(wv.z), o[5]);
        o[6] = fmaf(dl[e], bflo(wv.w), o[6]); o[7] = fmaf(dl[e], bfhi(wv.w), o[7]);
      }
      dor[c] = make_uint4(pack2bf(bflo(dv.x) + o[0], bfhi(dv.x) + o[1]), pack2bf(bflo(dv.y) + o[2], bfhi(dv.y) + o[3]),
                          pack2bf(bflo(dv.z) + o[4], bfhi(dv.z) + o[5]), pack2bf(bflo(dv.w) + o[6], bfhi(dv.w) + o[7]));
    }
  }
}

// ---- gate weight gradient: dWg[e, h] = sum_t dlogits[t, e] * x[t, h].  Deterministic two-level sum: workgroup (column block, slab) owns
//      512 columns (one dword = two bf16 columns per thread) of a slab of RW_SLAB tokens and writes fp32 partials [slab, E, H]; the
//      reduce kernel adds the slabs in slab order and folds the sum into the bf16 gradient the way `grad.add_(dW.to(bf16))` does.
constexpr int RW_SLAB = 512;
template <int E>
__global__ void __launch_bounds__(256) moe_router_wgrad_k(const uint16_t* __restrict__ x, const float* __restrict__ dlogits, float* __restrict__ part,
                                                          int64_t T, int H) {
  const int col = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 2;
  const int64_t t0 = (int64_t)blockIdx.y * RW_SLAB;
  const int64_t t1 = t0 + RW_SLAB < T ? t0 + RW_SLAB : T;
  float a0[E], a1[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
  if (col < H) {
    const uint16_t* xc = x + col;
    int64_t t = t0;
    for (; t + 4 <= t1; t += 4) {                      // four row dwords in flight per thread
      uint32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint32_t*>(xc + (t + u) * H);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* d = dlogits + (t + u) * E;        // wave-uniform address: scalar loads
#pragma unroll
        for (int e = 0; e < E; ++e) { a0[e] = fmaf(d[e], bflo(v[u]), a0[e]); a1[e] = fmaf(d[e], bfhi(v[u]), a1[e]); }
      }
    }
    for (; t < t1; ++t) {
      const uint32_t v = *reinterpret_cast<const uint32_t*>(xc + t * H);
      const float* d = dlogits + t * E;
#pragma unroll
      for (int e = 0; e < E; ++e) { a0[e] = fmaf(d[e], bflo(v), a0[e]); a1[e] = fmaf(d[e], bfhi(v), a1[e]); }
    }
    float* po = part + (int64_t)blockIdx.y * E * H + col;
#pragma unroll
    for (int e = 0; e < E; ++e) *reinterpret_cast<float2*>(po + (int64_t)e * H) = make_float2(a0[e], a1[e]);
  }
}

__global__ void __launch_bounds__(256) moe_router_wgrad_reduce_k(const float* __restrict__ part, uint16_t* __restrict__ grad, int nslab, int64_t EH) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= EH) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += part[(int64_t)k * EH + i];
  grad[i] = (uint16_t)f2bf(bf2f(grad[i]) + round_bf(s));            // grad.add_(dW.to(bfloat16)) in bf16 arithmetic
}

// ---- index: stable counting sort of the n = 2T (token, k) entries by expert, three launches:
//      (1) per-chunk expert histograms (4096 entries per workgroup), (2) one small workgroup turns them into chunk bases,
//      (3) every workgroup ranks its chunk in entry order (ballot + popcount per expert, running bases across the 16 rounds of 256).
//      Stable: inside an expert the rows are ordered by token, the order torch.where produces in the reference (:861).
constexpr int IDX_CHUNK = 4096, IDX_T = 256;
__global__ void __launch_bounds__(IDX_T) moe_hist_k(const int32_t* __restrict__ experts, int64_t n, int E, int32_t* __restrict__ chunk_counts) {
  __shared__ int32_t h[MOE_MAX_E];
  const int tid = threadIdx.x;
  if (tid < MOE_MAX_E) h[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * IDX_CHUNK;
  for (int j = tid; j < IDX_CHUNK; j += IDX_T) {
    const int64_t i = base + j;
    if (i < n) atomicAdd(&h[experts[i]], 1);
  }
  __syncthreads();
  if (tid < E) chunk_counts[(int64_t)blockIdx.x * E + tid] = h[tid];
}

// chunk_counts [nchunks, E] -> chunk_base [nchunks, E] (first sorted row of expert e's entries of chunk c), counts [E]
__global__ void __launch_bounds__(64) moe_scan_k(const int32_t* __restrict__ chunk_counts, int nchunks, int E, int32_t* __restrict__ chunk_base,
                                                 int32_t* __restrict__ counts) {
  __shared__ int32_t tot[MOE_MAX_E];
  const int lane = threadIdx.x;
  // lane e < E walks expert e's column (nchunks <= a few hundred)
  if (lane < E) {
    int32_t run = 0;
    for (int c = 0; c < nchunks; ++c) { const int32_t v = chunk_counts[(int64_t)c * E + lane]; chunk_base[(int64_t)c * E + lane] = run; run += v; }
    tot[lane] = run; counts[lane] = run;
  }
  __syncthreads();
  if (lane < E) {
    int32_t off = 0;
    for (int e = 0; e < lane; ++e) off += tot[e];
    for (int c = 0; c < nchunks; ++c) chunk_base[(int64_t)c * E + lane] += off;
  }
}

__global__ void __launch_bounds__(IDX_T) moe_rank_k(const int32_t* __restrict__ experts, int64_t n, int E, const int32_t* __restrict__ chunk_base,
                                                    int32_t* __restrict__ row_token, int32_t* __restrict__ rows) {
  __shared__ int32_t run[MOE_MAX_E];                 // next free sorted row per expert
  __shared__ int32_t wtot[IDX_T / 64][MOE_MAX_E];    // per wave: entries of expert e in this round
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < E) run[tid] = chunk_base[(int64_t)blockIdx.x * E + tid];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * IDX_CHUNK;
  for (int r = 0; r < IDX_CHUNK / IDX_T; ++r) {
    const int64_t i = base + r * IDX_T + tid;
    const int e = i < n ? experts[i] : -1;
    int my_rank = 0;
    for (int k = 0; k < E; ++k) {
      const uint64_t m = __ballot(e == k);
      if (e == k) my_rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wtot[wave][k] = __popcll(m);
    }
    __syncthreads();
    if (e >= 0) {
      int32_t pos = run[e] + my_rank;
      for (int w = 0; w < wave; ++w) pos += wtot[w][e];
      rows[i] = pos;
      row_token[pos] = (int32_t)(i >> 1);
    }
    __syncthreads();
    if (tid < E) run[tid] += wtot[0][tid] + wtot[1][tid] + wtot[2][tid] + wtot[3][tid];
    __syncthreads();
  }
}

// ---- combine: out[t] = bf16(res[t] + bf16(bf16(w0 * y[r0]) + bf16(w1 * y[r1])))
__global__ void __launch_bounds__(256) moe_combine_k(const uint16_t* __restrict__ y, const int32_t* __restrict__ rows,
                                                     const float* __restrict__ weights, const uint16_t* __restrict__ res,
                                                     uint16_t* __restrict__ out, int64_t T, int H) {
  const int HC = H >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * HC) return;
  const int64_t t = i / HC;
  const int c = (int)(i - t * HC);
  const int r0 = rows[2 * t], r1 = rows[2 * t + 1];
  const float w0 = weights[2 * t], w1 = weights[2 * t + 1];
  const uint4 a = reinterpret_cast<const uint4*>(y)[(int64_t)r0 * HC + c];
  const uint4 b = reinterpret_cast<const uint4*>(y)[(int64_t)r1 * HC + c];
  const uint4 r = res ? reinterpret_cast<const uint4*>(res)[i] : make_uint4(0, 0, 0, 0);
  const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, rv[4] = {r.x, r.y, r.z, r.w};
  uint32_t o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float lo = round_bf(round_bf(w0 * bflo(av[k])) + round_bf(w1 * bflo(bv[k])));
    const float hi = round_bf(round_bf(w0 * bfhi(av[k])) + round_bf(w1 * bfhi(bv[k])));
    o[k] = res ? pack2bf(lo + bflo(rv[k]), hi + bfhi(rv[k])) : pack2bf(lo, hi);
  }
  reinterpret_cast<uint4*>(out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- combine of the "f16_operands" policy: out[t] (fp32) = res[t] (fp32) + w0 * y[r0] + w1 * y[r1], y fp16 (the grouped w2 GEMM's one
//      rounding), everything else in fp32 -- the reference's arithmetic when the model runs in fp32 (:876-880, decoder :945)
__global__ void __launch_bounds__(256) moe_combine_f32_k(const uint16_t* __restrict__ y, const int32_t* __restrict__ rows,
                                                         const float* __restrict__ weights, const float* __restrict__ res,
                                                         float* __restrict__ out, int64_t T, int H) {
  const int HC = H >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * HC) return;
  const int64_t t = i / HC;
  const int c = (int)(i - t * HC);
  const int r0 = rows[2 * t], r1 = rows[2 * t + 1];
  const float w0 = weights[2 * t], w1 = weights[2 * t + 1];
  const uint4 a = reinterpret_cast<const uint4*>(y)[(int64_t)r0 * HC + c];
  const uint4 b = reinterpret_cast<const uint4*>(y)[(int64_t)r1 * HC + c];
  const float4 ra = res ? reinterpret_cast<const float4*>(res)[2 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 rb = res ? reinterpret_cast<const float4*>(res)[2 * i + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  const float rv[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
  float o[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = rv[2 * k] + (w0 * hlo(av[k]) + w1 * hlo(bv[k]));
    o[2 * k + 1] = rv[2 * k + 1] + (w0 * hhi(av[k]) + w1 * hhi(bv[k]));
  }
  reinterpret_cast<float4*>(out)[2 * i] = make_float4(o[0], o[1], o[2], o[3]);
  reinterpret_cast<float4*>(out)[2 * i + 1] = make_float4(o[4], o[5], o[6], o[7]);
}

// ---- backward of combine (autograd of final_hidden_states += expert_out * routing_weight, modeling_mixtral_gritlm.py:861-880):
//      one wave per routed row r:  dy[r] = bf16(w(r) * dout[token(r)]),  dw[token(r), slot(r)] = <y[r], dout[token(r)]>  (fp32)
__global__ void __launch_bounds__(256) moe_combine_bwd_k(const uint16_t* __restrict__ dout, const uint16_t* __restrict__ y,
                                                         const int32_t* __restrict__ row_token, const int32_t* __restrict__ rows,
                                                         const float* __restrict__ weights, uint16_t* __restrict__ dy, float* __restrict__ dw,
                                                         int64_t R, int H) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int HC = H >> 3;
  const int t = row_token[r];
  const int slot = (rows[2 * (int64_t)t] == (int32_t)r) ? 0 : 1;
  const float w = weights[2 * (int64_t)t + slot];
  const uint4* gp = reinterpret_cast<const uint4*>(dout) + (int64_t)t * HC;
  const uint4* yp = reinterpret_cast<const uint4*>(y) + r * HC;
  uint4* op = reinterpret_cast<uint4*>(dy) + r * HC;
  float dot = 0.f;
  for (int c = lane; c < HC; c += 64) {
    const uint4 g = gp[c], yv = yp[c];
    const uint32_t ga[4] = {g.x, g.y, g.z, g.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g0 = bflo(ga[k]), g1 = bfhi(ga[k]);
      dot += g0 * bflo(ya[k]) + g1 * bfhi(ya[k]);
      o[k] = pack2bf_hw(w * g0, w * g1);
    }
    op[c] = make_uint4(o[0], o[1], o[2], o[3]);
  }
  dot = wave_sum(dot);
  if (lane == 0) dw[2 * (int64_t)t + slot] = dot;
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

extern "C" int grit_moe_combine_bwd(const void* dout, const void* y, const int32_t* row_token, const int32_t* rows, const float* weights,
                                    void* dy, float* dw, int64_t T, int H, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(dout && y && row_token && rows && weights && dy && dw, GRIT_E_BADARG, "grit_moe_combine_bwd: null pointer");
  GRIT_REQUIRE(T > 0 && H > 0 && H % 8 == 0, GRIT_E_BADARG, "grit_moe_combine_bwd: bad sizes T=%lld H=%d", (long long)T, H);
  GRIT_REQUIRE(aligned16(dout) && aligned16(y) && aligned16(dy), GRIT_E_BADARG, "grit_moe_combine_bwd: pointers must be 16-byte aligned");
  const int64_t R = 2 * T;
  hipLaunchKernelGGL(moe_combine_bwd_k, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dout,
                     (const uint16_t*)y, row_token, rows, weights, (uint16_t*)dy, dw, R, H);
  GRIT_CHECK_LAUNCH("grit_moe_combine_bwd");
  return GRIT_OK;
}

extern "C" int grit_moe_router_top2(const void* x, const void* gate_w, int32_t* experts, float* weights, int64_t T, int H, int E,
                                    void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(x && gate_w && experts && weights, GRIT_E_BADARG, "grit_moe_router_top2: null pointer");
  GRIT_REQUIRE(T > 0 && H > 0 && H % 8 == 0, GRIT_E_BADARG, "grit_moe_router_top2: bad sizes T=%lld H=%d", (long long)T, H);
  GRIT_REQUIRE(E == 4 || E == 8 || E == 16, GRIT_E_UNSUPPORTED, "grit_moe_router_top2: num_experts=%d (4, 8 and 16 are built)", E);
  GRIT_REQUIRE((size_t)E * H * 2 <= 160 * 1024, GRIT_E_UNSUPPORTED, "grit_moe_router_top2: gate [%d,%d] exceeds LDS", E, H);
  GRIT_REQUIRE(aligned16(x) && aligned16(gate_w), GRIT_E_BADARG, "grit_moe_router_top2: pointers must be 16-byte aligned");
  const size_t lds = (size_t)E * H * 2;
  int64_t nb = (T + 3) / 4;
  if (nb > 1024) nb = 1024;
  hipStream_t st = (hipStream_t)stream;
#define GRIT_ROUTER(E_)                                                                                                       \
  do {                                                                                                                        \
    static std::atomic<uint64_t> optin_{0};                                                                                   \
    lds_optin_once(moe_router_top2_k<E_>, optin_, 160 * 1024);                                                                \
    hipLaunchKernelGGL(moe_router_top2_k<E_>, dim3((unsigned)nb), dim3(256), lds, st, (const uint16_t*)x, (const uint16_t*)gate_w, T, H,  \
                       experts, weights);                                                                                     \
  } while (0)
  if (E == 4) GRIT_ROUTER(4); else if (E == 8) GRIT_ROUTER(8); else GRIT_ROUTER(16);
  GRIT_CHECK_LAUNCH("grit_moe_router_top2");
  return GRIT_OK;
}

// Router of the "f16_operands" policy: h [T,H] fp32 (the residual stream entering the block's RMSNorm), ln_w [H] bf16, gate_w [E,H] bf16 ->
// experts [T,2] int32, weights [T,2] fp32 (NOT rounded).  See moe_router_top2_f32_k.
extern "C" int grit_moe_router_top2_f32(const float* h, const void* ln_w, float eps, const void* gate_w, int32_t* experts, float* weights,
                                        int64_t T, int H, int E, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(h && ln_w && gate_w && experts && weights, GRIT_E_BADARG, "grit_moe_router_top2_f32: null pointer");
  GRIT_REQUIRE(T > 0 && H > 0 && H % 8 == 0, GRIT_E_BADARG, "grit_moe_router_top2_f32: bad sizes T=%lld H=%d", (long long)T, H);
  GRIT_REQUIRE(E == 4 || E == 8 || E == 16, GRIT_E_UNSUPPORTED, "grit_moe_router_top2_f32: num_experts=%d (4, 8 and 16 are built)", E);
  GRIT_REQUIRE((size_t)(E + 1) * H * 2 <= 160 * 1024, GRIT_E_UNSUPPORTED, "grit_moe_router_top2_f32: gate [%d,%d] + norm weight exceed LDS", E, H);
  GRIT_REQUIRE(aligned16(h) && aligned16(gate_w) && aligned16(ln_w), GRIT_E_BADARG, "grit_moe_router_top2_f32: pointers must be 16-byte aligned");
  const size_t lds = (size_t)(E + 1) * H * 2;
  int64_t nb = (T + 3) / 4;
  if (nb > 2048) nb = 2048;
  hipStream_t st = (hipStream_t)stream;
#define GRIT_ROUTER32(E_)                                                                                                     \
  do {                                                                                                                        \
    static std::atomic<uint64_t> optin_{0};                                                                                   \
    lds_optin_once(moe_router_top2_f32_k<E_>, optin_, 160 * 1024);                                                            \
    hipLaunchKernelGGL(moe_router_top2_f32_k<E_>, dim3((unsigned)nb), dim3(256), lds, st, h, (const uint16_t*)ln_w, eps,      \
                       (const uint16_t*)gate_w, T, H, experts, weights);                                                      \
  } while (0)
  if (E == 4) GRIT_ROUTER32(4); else if (E == 8) GRIT_ROUTER32(8); else GRIT_ROUTER32(16);
  GRIT_CHECK_LAUNCH("grit_moe_router_top2_f32");
  return GRIT_OK;
}

// Combine of the "f16_operands" policy: y [2T,H] fp16, residual (nullable) / out [T,H] fp32 (out may alias residual).
extern "C" int grit_moe_combine_f32(const void* y, const int32_t* rows, const float* weights, const float* residual, float* out, int64_t T,
                                    int H, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(y && rows && weights && out, GRIT_E_BADARG, "grit_moe_combine_f32: null pointer");
  GRIT_REQUIRE(T > 0 && H > 0 && H % 8 == 0, GRIT_E_BADARG, "grit_moe_combine_f32: bad sizes");
  GRIT_REQUIRE(aligned16(y) && aligned16(out) && (!residual || aligned16(residual)), GRIT_E_BADARG,
               "grit_moe_combine_f32: pointers must be 16-byte aligned");
  const int64_t n = T * (H >> 3);
  hipLaunchKernelGGL(moe_combine_f32_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y, rows, weights,
                     residual, out, T, H);
  GRIT_CHECK_LAUNCH("grit_moe_combine_f32");
  return GRIT_OK;
}

// Router backward of the sparse-MoE block (training).  dw [T,2] fp32 = d loss / d routing weights (grit_moe_combine_bwd), experts [T,2] the
// forward's selection, aux_dlogits [T,E] fp32 or NULL; dx_in [T,H] bf16 or NULL (the experts' input gradient); outputs dx_out [T,H] bf16
// (may alias dx_in) and dlogits [T,E] fp32.
extern "C" int grit_moe_router_bwd(const void* x, const void* gate_w, const int32_t* experts, const float* dw, const float* aux_dlogits,
                                   const void* dx_in, void* dx_out, float* dlogits, int64_t T, int H, int E, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(x && gate_w && experts && dw && dx_out && dlogits, GRIT_E_BADARG, "grit_moe_router_bwd: null pointer");
  GRIT_REQUIRE(T > 0 && T < (1ll << 40) && H > 0 && H % 8 == 0, GRIT_E_BADARG, "grit_moe_router_bwd: bad sizes T=%lld H=%d", (long long)T, H);
  GRIT_REQUIRE(E == 4 || E == 8 || E == 16, GRIT_E_UNSUPPORTED, "grit_moe_router_bwd: num_experts=%d (4, 8 and 16 are built)", E);
  GRIT_REQUIRE((size_t)E * H * 2 <= 160 * 1024, GRIT_E_UNSUPPORTED, "grit_moe_router_bwd: gate [%d,%d] exceeds LDS", E, H);
  GRIT_REQUIRE(aligned16(x) && aligned16(gate_w) && aligned16(dx_out) && (!dx_in || aligned16(dx_in)), GRIT_E_BADARG,
               "grit_moe_router_bwd: pointers must be 16-byte aligned");
  const size_t lds = (size_t)E * H * 2;
  int64_t nb = (T + 3) / 4;
  if (nb > 1024) nb = 1024;
  hipStream_t st = (hipStream_t)stream;
#define GRIT_ROUTER_BWD(E_)                                                                                                   \
  do {                                                                                                                        \
    static std::atomic<uint64_t> optin_{0};                                                                                   \
    lds_optin_once(moe_router_bwd_k<E_>, optin_, 160 * 1024);                                                                 \
    hipLaunchKernelGGL(moe_router_bwd_k<E_>, dim3((unsigned)nb), dim3(256), lds, st, (const uint16_t*)x, (const uint16_t*)gate_w, experts, dw, \
                       aux_dlogits, (const uint16_t*)dx_in, (uint16_t*)dx_out, dlogits, T, H);                                \
  } while (0)
  if (E == 4) GRIT_ROUTER_BWD(4); else if (E == 8) GRIT_ROUTER_BWD(8); else GRIT_ROUTER_BWD(16);
  GRIT_CHECK_LAUNCH("grit_moe_router_bwd");
  return GRIT_OK;
}

extern "C" int64_t grit_moe_router_wgrad_workspace_floats(int64_t T, int H, int E) {
  if (T <= 0 || T > (1ll << 40) || H <= 0 || H > (1 << 20) || E <= 0 || E > MOE_MAX_E) return 0;       // 0 for sizes the compute call rejects
  return ((T + RW_SLAB - 1) / RW_SLAB) * (int64_t)E * H;
}

// gate.weight gradient: grad [E,H] bf16 (in/out) += bf16(dlogits^T [E,T] @ x [T,H]); workspace: grit_moe_router_wgrad_workspace_floats floats.
extern "C" int grit_moe_router_wgrad(const void* x, const float* dlogits, void* grad, float* workspace, int64_t T, int H, int E, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(x && dlogits && grad && workspace, GRIT_E_BADARG, "grit_moe_router_wgrad: null pointer");
  GRIT_REQUIRE(T > 0 && T < (1ll << 40) && H > 0 && H <= (1 << 20) && H % 2 == 0, GRIT_E_BADARG, "grit_moe_router_wgrad: bad sizes T=%lld H=%d", (long long)T, H);
  GRIT_REQUIRE(E == 4 || E == 8 || E == 16, GRIT_E_UNSUPPORTED, "grit_moe_router_wgrad: num_experts=%d (4, 8 and 16 are built)", E);
  const int64_t nslab = (T + RW_SLAB - 1) / RW_SLAB;
  GRIT_REQUIRE(nslab <= 65535, GRIT_E_UNSUPPORTED, "grit_moe_router_wgrad: T=%lld exceeds 65535 slabs of %d tokens", (long long)T, RW_SLAB);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((H / 2 + 255) / 256), (unsigned)nslab);
  if (E == 4) hipLaunchKernelGGL(moe_router_wgrad_k<4>, grid, dim3(256), 0, st, (const uint16_t*)x, dlogits, workspace, T, H);
  else if (E == 8) hipLaunchKernelGGL(moe_router_wgrad_k<8>, grid, dim3(256), 0, st, (const uint16_t*)x, dlogits, workspace, T, H);
  else hipLaunchKernelGGL(moe_router_wgrad_k<16>, grid, dim3(256), 0, st, (const uint16_t*)x, dlogits, workspace, T, H);
  GRIT_CHECK_LAUNCH("grit_moe_router_wgrad");
  const int64_t EH = (int64_t)E * H;
  hipLaunchKernelGGL(moe_router_wgrad_reduce_k, dim3((unsigned)((EH + 255) / 256)), dim3(256), 0, st, (const float*)workspace, (uint16_t*)grad,
                     (int)nslab, EH);
  GRIT_CHECK_LAUNCH("grit_moe_router_wgrad: reduce");
  return GRIT_OK;
}

extern "C" int64_t grit_moe_index_workspace_ints(int64_t T, int E) {
  if (T < 0 || E <= 0 || T > (1ll << 40)) return 0;         // 0 for sizes grit_moe_index rejects
  const int64_t nch = (2 * T + IDX_CHUNK - 1) / IDX_CHUNK;
  return 2 * (nch > 0 ? nch : 1) * E;
}

extern "C" int grit_moe_index(const int32_t* experts, int64_t T, int E, int32_t* counts, int32_t* row_token, int32_t* rows, int32_t* workspace,
                              void* stream) {
  GRIT_REQUIRE(counts && workspace, GRIT_E_BADARG, "grit_moe_index: null pointer");
  GRIT_REQUIRE(E > 0 && E <= MOE_MAX_E, GRIT_E_UNSUPPORTED, "grit_moe_index: num_experts=%d (max %d)", E, MOE_MAX_E);
  GRIT_REQUIRE(T >= 0 && T < (1ll << 30), GRIT_E_BADARG, "grit_moe_index: bad T");
  GRIT_REQUIRE(T == 0 || (experts && row_token && rows), GRIT_E_BADARG, "grit_moe_index: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = 2 * T;
  const int nch = (int)((n + IDX_CHUNK - 1) / IDX_CHUNK);
  int32_t* chunk_counts = workspace;
  int32_t* chunk_base = workspace + (int64_t)(nch > 0 ? nch : 1) * E;
  if (nch > 0) {
    hipLaunchKernelGGL(moe_hist_k, dim3((unsigned)nch), dim3(IDX_T), 0, st, experts, n, E, chunk_counts);
    GRIT_CHECK_LAUNCH("grit_moe_index: histogram");
  }
  hipLaunchKernelGGL(moe_scan_k, dim3(1), dim3(64), 0, st, chunk_counts, nch, E, chunk_base, counts);
  GRIT_CHECK_LAUNCH("grit_moe_index: scan");
  if (nch > 0) {
    hipLaunchKernelGGL(moe_rank_k, dim3((unsigned)nch), dim3(IDX_T), 0, st, experts, n, E, chunk_base, row_token, rows);
    GRIT_CHECK_LAUNCH("grit_moe_index: rank");
  }
  return GRIT_OK;
}

extern "C" int grit_moe_combine(const void* y, const int32_t* rows, const float* weights, const void* residual, void* out, int64_t T, int H,
                                void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(y && rows && weights && out, GRIT_E_BADARG, "grit_moe_combine: null pointer");
  GRIT_REQUIRE(T > 0 && H > 0 && H % 8 == 0, GRIT_E_BADARG, "grit_moe_combine: bad sizes");
  GRIT_REQUIRE(aligned16(y) && aligned16(out) && (!residual || aligned16(residual)), GRIT_E_BADARG,
               "grit_moe_combine: pointers must be 16-byte aligned");
  const int64_t n = T * (H >> 3);
  hipLaunchKernelGGL(moe_combine_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y, rows, weights,
                     (const uint16_t*)residual, (uint16_t*)out, T, H);
  GRIT_CHECK_LAUNCH("grit_moe_combine");
  return GRIT_OK;
}
