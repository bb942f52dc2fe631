// Backward of the bidirectional flash attention (recompute form): dq, dk, dv from (q,k,v,out,dout,lse).
//
// Autograd of MistralSdpaAttention's core (scripts/modeling_mistral_gritlm.py:690-698) for the contrastive
// training step (GradCache pass 2, grad_cache.py:213-242).  No S x S tensor is ever stored:
//   delta[q]  = sum_d dO[q,d] O[q,d]
//   P         = exp(S*scale - lse),  dP = dO V^T,  dS = P (dP - delta)
//   dV = P^T dO      dK = scale dS^T Q      dQ = scale dS K
// Two launches (no atomics, deterministic):
//   attn_bwd_dkdv_k : one workgroup per (batch, kv head, 128 keys); each lane OWNS one key, K/V fragments stay in
//                     registers, loops over the q heads of the GQA group and all query tiles;
//   attn_bwd_dq_k   : one workgroup per (batch, q head, 128 queries); each lane OWNS one query, loops over KV tiles.
// As in the forward every product is arranged so that the owned index is the MFMA "column" (lane&31): the softmax
// statistics are lane-local (dq kernel) or a broadcast LDS read (dkdv kernel), and P / dS feed the next MFMA from
// registers with the contraction-index permutation applied on the transposed-LDS-image side.
#include "common.h"

namespace grit {

constexpr int AB_D = 128;
constexpr int AB_PITCH = 136;                 // transposed image: [128 d][64 rows] bf16, 136-B row pitch
constexpr int AB_RM = 64 * AB_D * 2;          // 16384: row-major image [64 rows][128 d], 16-B slots XOR (row&15)
constexpr int AB_TR = AB_D * AB_PITCH;        // 17408

__device__ __forceinline__ uint32_t lo16b(uint32_t w) { return w & 0xffffu; }
__device__ __forceinline__ uint32_t hi16b(uint32_t w) { return w >> 16; }

// Stage a [64 rows][128] bf16 tile (rows r0.. of `base`, clamped to row < limit) with 256 threads.
// "pair" mapping (2 rows x 16 B per item, 2 items per thread): can write the row-major image, the transposed image, or both.
__device__ __forceinline__ void stage_pairs(const uint16_t* __restrict__ base, int64_t stride, int r0, int limit, char* rm, char* tr) {
  const int tid = threadIdx.x, lane32 = tid & 31, half = tid >> 5;
  const int kp = (half & 3) * 8 + (lane32 & 7);
  int ra = r0 + 2 * kp, rb = ra + 1;
  ra = ra < limit ? ra : limit - 1; rb = rb < limit ? rb : limit - 1;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int dg = (half >> 2) * 4 + (lane32 >> 3) + 8 * it;
    const uint4 a = *reinterpret_cast<const uint4*>(base + (int64_t)ra * stride + dg * 8);
    const uint4 c = *reinterpret_cast<const uint4*>(base + (int64_t)rb * stride + dg * 8);
    if (rm != nullptr) {
      const int row = 2 * kp;
      *reinterpret_cast<uint4*>(rm + row * 256 + ((dg ^ (row & 15)) << 4)) = a;
      *reinterpret_cast<uint4*>(rm + (row + 1) * 256 + ((dg ^ ((row + 1) & 15)) << 4)) = c;
    }
    if (tr != nullptr) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(tr + (dg * 8) * AB_PITCH + kp * 4);
      constexpr int P4 = AB_PITCH / 4;
      dst[0 * P4] = lo16b(a.x) | (lo16b(c.x) << 16); dst[1 * P4] = hi16b(a.x) | (hi16b(c.x) << 16);
      dst[2 * P4] = lo16b(a.y) | (lo16b(c.y) << 16); dst[3 * P4] = hi16b(a.y) | (hi16b(c.y) << 16);
      dst[4 * P4] = lo16b(a.z) | (lo16b(c.z) << 16); dst[5 * P4] = hi16b(a.z) | (hi16b(c.z) << 16);
      dst[6 * P4] = lo16b(a.w) | (lo16b(c.w) << 16); dst[7 * P4] = hi16b(a.w) | (hi16b(c.w) << 16);
    }
  }
}

// A-operand fragment from a row-major image: A[row = rb*32 + (lane&31)][k = 16ks + 8hi + j]
__device__ __forceinline__ bf16x8_t frag_rm(const char* rm, int rb, int ks, int lane) {
  const int row = rb * 32 + (lane & 31);
  return *reinterpret_cast<const bf16x8_t*>(rm + row * 256 + (((2 * ks + (lane >> 5)) ^ (row & 15)) << 4));
}
// A-operand fragment from a transposed image: A[row = d = db*32 + (lane&31)][k = (hi,j)] = X[r = rb*32 + 16c + 8(j>>2) + 4hi + (j&3)][d]
__device__ __forceinline__ bf16x8_t frag_tr(const char* tr, int db, int rb, int c, int lane) {
  const char* p = tr + (db * 32 + (lane & 31)) * AB_PITCH + (rb * 32 + c * 16 + 4 * (lane >> 5)) * 2;
  const uint2 v0 = *reinterpret_cast<const uint2*>(p);
  const uint2 v1 = *reinterpret_cast<const uint2*>(p + 16);
  return __builtin_bit_cast(bf16x8_t, make_uint4(v0.x, v0.y, v1.x, v1.y));
}
__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& v, int c) {
  return __builtin_bit_cast(bf16x8_t, make_uint4(pack2bf_hw(v[8 * c + 0], v[8 * c + 1]), pack2bf_hw(v[8 * c + 2], v[8 * c + 3]),
                                                  pack2bf_hw(v[8 * c + 4], v[8 * c + 5]), pack2bf_hw(v[8 * c + 6], v[8 * c + 7])));
}

// ------------------------------------------------------------------ delta = rowsum(dO * O) per (b, head, q)
// padded layout: delta [B, nq, S];  varlen (S == 0): delta [T, nq]
__global__ void __launch_bounds__(256) attn_delta_k(const uint16_t* __restrict__ out, const uint16_t* __restrict__ dout,
                                                    float* __restrict__ delta, int64_t T, int S, int nq, int64_t out_stride) {
  const int64_t item = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;  // (token, head), 16 lanes each
  const int sub = threadIdx.x & 15;
  float acc = 0.f;
  const bool on = item < T * nq;
  int64_t t = 0; int h = 0;
  if (on) {
    t = item / nq; h = (int)(item - t * nq);
    const int64_t off = t * out_stride + (int64_t)h * AB_D + sub * 8;
    const uint4 o = *reinterpret_cast<const uint4*>(out + off), g = *reinterpret_cast<const uint4*>(dout + off);
    acc = bflo(o.x) * bflo(g.x) + bfhi(o.x) * bfhi(g.x) + bflo(o.y) * bflo(g.y) + bfhi(o.y) * bfhi(g.y) +
          bflo(o.z) * bflo(g.z) + bfhi(o.z) * bfhi(g.z) + bflo(o.w) * bflo(g.w) + bfhi(o.w) * bfhi(g.w);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (on && sub == 0) {
    if (S == 0) {
      delta[t * nq + h] = acc;
    } else {
      const int64_t b = t / S; const int s = (int)(t - b * S);
      delta[(b * nq + h) * S + s] = acc;
    }
  }
}

// ------------------------------------------------------------------ dK, dV
template <bool VARLEN>
__global__ void __launch_bounds__(256)
attn_bwd_dkdv_k(const uint16_t* __restrict__ qkv, const uint64_t* __restrict__ key_bits, const int32_t* __restrict__ cu_seqlens,
                const uint16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
                uint16_t* __restrict__ dqkv, int S_arg, int nq, int nkv, int64_t qkv_stride, int64_t out_stride, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_rm = smem; char* q_tr = q_rm + AB_RM; char* d_rm = q_tr + AB_TR; char* d_tr = d_rm + AB_RM;
  float* st = reinterpret_cast<float*>(d_tr + AB_TR);  // [64] lse (log2 domain), [64] delta

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int kblk = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const int group = nq / nkv;
  int S = S_arg;
  int64_t row0 = (int64_t)b * S_arg;
  if constexpr (VARLEN) {
    row0 = cu_seqlens[b];
    S = cu_seqlens[b + 1] - cu_seqlens[b];
    if (kblk * 128 >= S) return;
  }
  const int key = kblk * 128 + wave * 32 + (lane & 31);
  bool key_ok = key < S;
  if constexpr (!VARLEN) {
    const int W = (S + 63) >> 6;
    key_ok = key_ok && ((key_bits[(int64_t)b * W + (key >> 6)] >> (key & 63)) & 1ull);
  }
  const int key_ld = key < S ? key : S - 1;
  const float scale_log2 = scale * 1.4426950408889634f;

  // K, V fragments of the owned key (B operands): X[key][16ks + 8hi .. +8]
  bf16x8_t kf[8], vf[8];
  {
    const uint16_t* kp = qkv + (row0 + key_ld) * qkv_stride + (int64_t)(nq + hk) * AB_D + hi * 8;
    const uint16_t* vp = kp + (int64_t)nkv * AB_D;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + ks * 16);
      vf[ks] = *reinterpret_cast<const bf16x8_t*>(vp + ks * 16);
    }
  }
  f32x16_t dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  const int nqt = (S + 63) >> 6;
  for (int g = 0; g < group; ++g) {
    const int h = hk * group + g;
    const uint16_t* qbase = qkv + row0 * qkv_stride + (int64_t)h * AB_D;
    const uint16_t* dobase = dout + row0 * out_stride + (int64_t)h * AB_D;
    // statistics of query q of head h: padded layout [B,nq,S] (stride 1 over q), packed layout [T,nq] (stride nq over q)
    const float* lrow = VARLEN ? lse + row0 * nq + h : lse + ((int64_t)b * nq + h) * S;
    const float* drow = VARLEN ? delta + row0 * nq + h : delta + ((int64_t)b * nq + h) * S;
    const int64_t qs = VARLEN ? nq : 1;
    for (int qt = causal ? 2 * kblk : 0; qt < nqt; ++qt) {   // causal: query tiles before this key block see none of its keys
      stage_pairs(qbase, qkv_stride, qt * 64, S, q_rm, q_tr);
      stage_pairs(dobase, out_stride, qt * 64, S, d_rm, d_tr);
      if (tid < 64) {
        const int q = qt * 64 + tid;
        st[tid] = q < S ? lrow[q * qs] * 1.4426950408889634f : INFINITY;  // +inf -> P = 0 for rows past the sequence
        st[64 + tid] = q < S ? drow[q * qs] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x16_t s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(q_rm, qb, ks, lane), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(d_rm, qb, ks, lane), vf[ks], dp, 0, 0, 0);
        }
        // regs 4g..4g+3 <-> q = 32qb + 8g + 4hi + 0..3
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 l4 = *reinterpret_cast<const float4*>(st + qb * 32 + 8 * gq + 4 * hi);
          const float4 d4 = *reinterpret_cast<const float4*>(st + 64 + qb * 32 + 8 * gq + 4 * hi);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * gq + e;
            const bool seen = key_ok && (!causal || key <= qt * 64 + qb * 32 + 8 * gq + 4 * hi + e);
            const float p = seen ? __builtin_amdgcn_exp2f(s[r] * scale_log2 - lv[e]) : 0.f;
            s[r] = p;
            dp[r] = p * (dp[r] - dl[e]);
          }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8_t pb = pack8(s, c), dsb = pack8(dp, c);
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(d_tr, db, qb, c, lane), pb, dv[db], 0, 0, 0);
            dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(q_tr, db, qb, c, lane), dsb, dk[db], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
  }
  if (key < S) {
    uint16_t* kp = dqkv + (row0 + key) * qkv_stride + (int64_t)(nq + hk) * AB_D + 4 * hi;
    uint16_t* vp = kp + (int64_t)nkv * AB_D;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        *reinterpret_cast<uint2*>(kp + db * 32 + gq * 8) = make_uint2(pack2bf(dk[db][4 * gq] * scale, dk[db][4 * gq + 1] * scale),
                                                                     pack2bf(dk[db][4 * gq + 2] * scale, dk[db][4 * gq + 3] * scale));
        *reinterpret_cast<uint2*>(vp + db * 32 + gq * 8) = make_uint2(pack2bf(dv[db][4 * gq], dv[db][4 * gq + 1]),
                                                                     pack2bf(dv[db][4 * gq + 2], dv[db][4 * gq + 3]));
      }
  }
}

// ------------------------------------------------------------------ dQ
template <bool VARLEN>
__global__ void __launch_bounds__(256)
attn_bwd_dq_k(const uint16_t* __restrict__ qkv, const uint64_t* __restrict__ key_bits, const int32_t* __restrict__ cu_seqlens,
              const uint16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
              uint16_t* __restrict__ dqkv, int S_arg, int nq, int nkv, int64_t qkv_stride, int64_t out_stride, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_rm = smem; char* k_tr = k_rm + AB_RM; char* v_rm = k_tr + AB_TR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (nq / nkv);
  int S = S_arg;
  int64_t row0 = (int64_t)b * S_arg;
  const uint64_t* bits = nullptr;
  int ntiles = 0;
  if constexpr (VARLEN) {
    row0 = cu_seqlens[b];
    S = cu_seqlens[b + 1] - cu_seqlens[b];
    if (qblk * 128 >= S) return;
    ntiles = (S + 63) >> 6;
  } else {
    const int W = (S + 63) >> 6;
    bits = key_bits + (int64_t)b * W;
    for (int w = W - 1; w >= 0; --w)
      if (bits[w] != 0) { ntiles = w + 1; break; }
  }
  if (causal) ntiles = ntiles < 2 * qblk + 2 ? ntiles : 2 * qblk + 2;
  const int q = qblk * 128 + wave * 32 + (lane & 31);
  const int q_ld = q < S ? q : S - 1;
  const float scale_log2 = scale * 1.4426950408889634f;
  const int64_t sidx = VARLEN ? (row0 + q_ld) * nq + h : ((int64_t)b * nq + h) * S + q_ld;
  const float lse2 = lse[sidx] * 1.4426950408889634f;
  const float dlt = delta[sidx];

  bf16x8_t qf[8], dof[8];
  {
    const uint16_t* qp = qkv + (row0 + q_ld) * qkv_stride + (int64_t)h * AB_D + hi * 8;
    const uint16_t* dp = dout + (row0 + q_ld) * out_stride + (int64_t)h * AB_D + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
      dof[ks] = *reinterpret_cast<const bf16x8_t*>(dp + ks * 16);
    }
  }
  f32x16_t dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  const uint16_t* kbase = qkv + row0 * qkv_stride + (int64_t)(nq + hk) * AB_D;
  const uint16_t* vbase = kbase + (int64_t)nkv * AB_D;
  for (int t = 0; t < ntiles; ++t) {
    stage_pairs(kbase, qkv_stride, t * 64, S, k_rm, k_tr);
    stage_pairs(vbase, qkv_stride, t * 64, S, v_rm, nullptr);
    __syncthreads();
    uint64_t word;
    if constexpr (VARLEN) {
      const int rem = S - t * 64;
      word = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    } else {
      word = bits[t];
    }
    if (causal) {
      const int n = q - t * 64 + 1;
      word &= n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
    }
    const uint32_t wlo = (uint32_t)(word >> (4 * hi)), whi = (uint32_t)(word >> (32 + 4 * hi));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(k_rm, kb, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rm(v_rm, kb, ks, lane), dof[ks], dp, 0, 0, 0);
      }
      const uint32_t wsel = kb ? whi : wlo;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kbit = (r & 3) + 8 * (r >> 2);
        const float p = ((wsel >> kbit) & 1u) ? __builtin_amdgcn_exp2f(s[r] * scale_log2 - lse2) : 0.f;
        dp[r] = p * (dp[r] - dlt);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bf16x8_t dsb = pack8(dp, c);
#pragma unroll
        for (int db = 0; db < 4; ++db)
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(k_tr, db, kb, c, lane), dsb, dq[db], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (q < S) {
    uint16_t* op = dqkv + (row0 + q) * qkv_stride + (int64_t)h * AB_D + 4 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<uint2*>(op + db * 32 + gq * 8) = make_uint2(pack2bf(dq[db][4 * gq] * scale, dq[db][4 * gq + 1] * scale),
                                                                     pack2bf(dq[db][4 * gq + 2] * scale, dq[db][4 * gq + 3] * scale));
  }
}

}  // namespace grit

using namespace grit;

static int attn_bwd_launch(bool varlen, int causal, const void* qkv, const uint64_t* key_bits, const int32_t* cu, const void* out, const void* dout,
                           const float* lse, float* delta, void* dqkv, int B, int S_or_maxlen, int64_t T, int nq, int nkv,
                           int64_t qkv_stride, int64_t out_stride, float scale, hipStream_t st) {
  const int64_t items = T * nq;
  hipLaunchKernelGGL(attn_delta_k, dim3((unsigned)((items * 16 + 255) / 256)), dim3(256), 0, st, (const uint16_t*)out,
                     (const uint16_t*)dout, delta, T, varlen ? 0 : S_or_maxlen, nq, out_stride);
  GRIT_CHECK_LAUNCH("grit_attn_bidir_bwd: delta");
  static bool attr_set = false;
  const int lds_kv = 2 * AB_RM + 2 * AB_TR + 512, lds_q = 2 * AB_RM + AB_TR;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
    attr_set = true;
  }
  const unsigned nblk = (unsigned)((S_or_maxlen + 127) / 128);
  if (varlen) {
    hipLaunchKernelGGL(attn_bwd_dkdv_k<true>, dim3(nblk, (unsigned)nkv, (unsigned)B), dim3(256), lds_kv, st, (const uint16_t*)qkv, key_bits, cu,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, S_or_maxlen, nq, nkv, qkv_stride, out_stride, scale, causal);
    GRIT_CHECK_LAUNCH("grit_attn_bidir_varlen_bwd: dkdv");
    hipLaunchKernelGGL(attn_bwd_dq_k<true>, dim3(nblk, (unsigned)nq, (unsigned)B), dim3(256), lds_q, st, (const uint16_t*)qkv, key_bits, cu,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, S_or_maxlen, nq, nkv, qkv_stride, out_stride, scale, causal);
    GRIT_CHECK_LAUNCH("grit_attn_bidir_varlen_bwd: dq");
  } else {
    hipLaunchKernelGGL(attn_bwd_dkdv_k<false>, dim3(nblk, (unsigned)nkv, (unsigned)B), dim3(256), lds_kv, st, (const uint16_t*)qkv, key_bits, cu,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, S_or_maxlen, nq, nkv, qkv_stride, out_stride, scale, causal);
    GRIT_CHECK_LAUNCH("grit_attn_bidir_bwd: dkdv");
    hipLaunchKernelGGL(attn_bwd_dq_k<false>, dim3(nblk, (unsigned)nq, (unsigned)B), dim3(256), lds_q, st, (const uint16_t*)qkv, key_bits, cu,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, S_or_maxlen, nq, nkv, qkv_stride, out_stride, scale, causal);
    GRIT_CHECK_LAUNCH("grit_attn_bidir_bwd: dq");
  }
  return GRIT_OK;
}

static int attn_bwd_padded(int causal, const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                           float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                           int64_t out_stride, float scale, void* stream) {
  GRIT_REQUIRE(qkv && key_bits && out && dout && lse && delta && dqkv, GRIT_E_BADARG, "grit_attn_bidir_bwd: null pointer");
  GRIT_REQUIRE(B > 0 && S > 0 && nq > 0 && nkv > 0, GRIT_E_BADARG, "grit_attn_bidir_bwd: bad sizes");
  GRIT_REQUIRE(d == AB_D, GRIT_E_UNSUPPORTED, "grit_attn_bidir_bwd: head_dim=%d (only 128 is built)", d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "grit_attn_bidir_bwd: nq not a multiple of nkv");
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "grit_attn_bidir_bwd: bad strides");
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), GRIT_E_BADARG,
               "grit_attn_bidir_bwd: pointers must be 16-byte aligned");
  GRIT_REQUIRE(nq <= 65535 && B <= 65535, GRIT_E_UNSUPPORTED, "grit_attn_bidir_bwd: grid too large");
  return attn_bwd_launch(false, causal, qkv, key_bits, nullptr, out, dout, lse, delta, dqkv, B, S, (int64_t)B * S, nq, nkv, qkv_stride,
                         out_stride, scale, (hipStream_t)stream);
}
extern "C" int grit_attn_bidir_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                                   float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                                   int64_t out_stride, float scale, void* stream) {
  return attn_bwd_padded(0, qkv, key_bits, out, dout, lse, delta, dqkv, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_causal_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                                    float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                                    int64_t out_stride, float scale, void* stream) {
  return attn_bwd_padded(1, qkv, key_bits, out, dout, lse, delta, dqkv, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}

static int attn_bwd_varlen(int causal, const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                           float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                           int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  GRIT_REQUIRE(qkv && cu_seqlens && out && dout && lse && delta && dqkv, GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: null pointer");
  GRIT_REQUIRE(B > 0 && max_len > 0 && T > 0 && nq > 0 && nkv > 0, GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: bad sizes");
  GRIT_REQUIRE(d == AB_D, GRIT_E_UNSUPPORTED, "grit_attn_bidir_varlen_bwd: head_dim=%d (only 128 is built)", d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: nq not a multiple of nkv");
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: bad strides");
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), GRIT_E_BADARG,
               "grit_attn_bidir_varlen_bwd: pointers must be 16-byte aligned");
  GRIT_REQUIRE(nq <= 65535 && B <= 65535, GRIT_E_UNSUPPORTED, "grit_attn_bidir_varlen_bwd: grid too large");
  return attn_bwd_launch(true, causal, qkv, nullptr, cu_seqlens, out, dout, lse, delta, dqkv, B, max_len, T, nq, nkv, qkv_stride, out_stride,
                         scale, (hipStream_t)stream);
}
extern "C" int grit_attn_bidir_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                                          float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                                          int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_bwd_varlen(0, qkv, cu_seqlens, out, dout, lse, delta, dqkv, B, max_len, T, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_causal_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                                           float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                                           int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_bwd_varlen(1, qkv, cu_seqlens, out, dout, lse, delta, dqkv, B, max_len, T, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
