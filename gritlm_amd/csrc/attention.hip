// Bidirectional (non-causal) flash attention forward for gfx950, GQA, head_dim 128, key-padding bitmask.
//
// Replaces repeat_kv + the additive [B,1,S,S] mask + F.scaled_dot_product_attention of
// scripts/modeling_mistral_gritlm.py (:182-191, :1017-1036, :690-698).  No repeat_kv copy (the kv head is
// index arithmetic), no mask tensor (one uint64 per 64 keys), no S x S score matrix.
//
// Structure: one 256-thread workgroup = 128 query rows of one (batch, head); each wave owns 32 rows; two workgroups per CU.
// KV tiles of 64 keys go HBM -> LDS by direct LDS-DMA (global_load_lds, 16 B per lane, 1 KiB = 4 key rows per wave instruction) into
// a two-stage ring: the DMA of tile t+1 is issued right after the single barrier of tile t and lands under tile t's MFMAs and
// softmax -- no staging registers, no ds_write pass, one barrier per tile.  Both images are row-major [key][256 B] with the 16-byte
// units XOR-swizzled through the per-lane SOURCE address (the LDS image of a DMA is lane-linear): K unit ^= key & 15
// (conflict-free ds_read_b128 of a 32-key fragment), V unit ^= 4 (key & 3) (conflict-free ds_read_b64_tr_b16: the 16 lanes of a
// transposing read touch 4 keys x 32 B, the XOR puts them -- and the second 16-lane group -- on 16 distinct units of one 256-B bank row).
//   S^T = K Q^T     v_mfma_f32_32x32x16_bf16(A = K rows, B = Q)  -> lane (q = lane&31) holds 32 keys' scores
//   O^T = V^T P^T   v_mfma_f32_32x32x16_bf16(A = V^T rows, B = P) -> lane (q = lane&31) holds 64 of its d's
// Both products are "swapped" so that every softmax statistic (max, sum, rescale) is lane-local: the only
// cross-lane traffic per tile is one shuffle with lane^32 for the row max.  The P operand needs no
// permlane/LDS round trip: the MFMA contraction index is permuted identically on the V^T side
// (key(kb,c,hi,j) = 32kb + 16c + 8(j>>2) + 4hi + (j&3)): two transposing 8-byte LDS reads whose per-lane
// addresses select exactly those keys.
#include "common.h"

namespace grit {

constexpr int ATT_D = 128;
constexpr int ATT_QB = 128;   // query rows per workgroup
constexpr int ATT_KB = 64;    // keys per tile
constexpr int V_PITCH = 256;  // bytes per key row of the row-major V image
constexpr int K_LDS_BYTES = ATT_KB * ATT_D * 2;  // 16384
constexpr int V_LDS_BYTES = ATT_KB * V_PITCH;    // 16384
constexpr int ATT_STAGE_BYTES = K_LDS_BYTES + V_LDS_BYTES;   // 32 KiB per stage, two stages
typedef const __attribute__((address_space(1))) void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ uint32_t lo16(uint32_t w) { return w & 0xffffu; }
__device__ __forceinline__ uint32_t hi16(uint32_t w) { return w >> 16; }

// VARLEN: sequences are packed back to back (no padding rows at all); cu_seqlens[b] is the first row of sequence b and
// every key of a sequence is valid, so the key bitmask is synthesised from the length.
// CAUSAL: additionally key <= query (the generative branch of unified training, MistralSdpaAttention with is_causal=True,
// modeling_mistral_gritlm.py:690-698 / :1017-1036); tiles past the workgroup's last query are skipped.
template <bool VARLEN, bool CAUSAL>
__global__ void __launch_bounds__(256, 2)
attn_bidir_fwd_k(const uint16_t* __restrict__ qkv, const uint64_t* __restrict__ key_bits, const int32_t* __restrict__ cu_seqlens,
                 uint16_t* __restrict__ out, float* __restrict__ lse, int S_arg, int nq, int nkv, int64_t qkv_stride,
                 int64_t out_stride, float scale_log2) {
  __shared__ __attribute__((aligned(16))) char smem[2 * ATT_STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (nq / nkv);
  int S = S_arg;
  int64_t row0 = (int64_t)b * S_arg;
  int ntiles = 0;
  const uint64_t* bits = nullptr;
  if constexpr (VARLEN) {
    row0 = cu_seqlens[b];
    S = cu_seqlens[b + 1] - cu_seqlens[b];
    if (qb * ATT_QB >= S) return;              // uniform per workgroup
    ntiles = (S + 63) >> 6;
  } else {
    const int W = (S + 63) >> 6;
    bits = key_bits + (int64_t)b * W;
    // number of KV tiles that contain at least one valid key (trailing padding is never loaded)
    for (int w = W - 1; w >= 0; --w)
      if (bits[w] != 0) { ntiles = w + 1; break; }
  }

  if constexpr (CAUSAL) {
    const int lim = 2 * qb + 2;                 // tiles holding keys <= the last query of this workgroup
    ntiles = ntiles < lim ? ntiles : lim;
  }

  const uint16_t* qbase = qkv + (int64_t)h * ATT_D;
  const uint16_t* kbase = qkv + (int64_t)(nq + hk) * ATT_D;
  const uint16_t* vbase = qkv + (int64_t)(nq + nkv + hk) * ATT_D;

  const int ql = lane & 31, hi = lane >> 5;
  const int q_row = qb * ATT_QB + wave * 32 + ql;
  const int q_ld = q_row < S ? q_row : S - 1;

  // ---- Q fragments (B operand): lane holds Q[q][16ks + 8hi .. +8]
  bf16x8_t qf[8];
  {
    const uint16_t* qp = qbase + (row0 + q_ld) * qkv_stride + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16);
  }

  // ---- LDS-DMA roles: wave w stages keys 16w .. 16w+15 of a tile, four 1-KiB instructions for K and four for V (4 keys each);
  //      lane -> key 16w + 4i + (lane>>4), physical 16-byte unit lane&15, which holds the LOGICAL unit (lane&15) ^ swizzle(key)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int st_key = 16 * wv + (lane >> 4);                                     // + 4i
  const int v_unit = (lane & 15) ^ (4 * ((lane >> 4) & 3));                     // V: unit ^= 4 (key & 3); key & 3 == (lane>>4) & 3
  auto stage_tile = [&](int t, int buf) {
    char* kdst = smem + buf * ATT_STAGE_BYTES + wv * 4096;
    char* vdst = kdst + K_LDS_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * ATT_KB + st_key + 4 * i;
      key = key < S ? key : S - 1;
      const int k_unit = (lane & 15) ^ ((4 * i + (lane >> 4)) & 15);           // K: unit ^= key & 15
      const uint16_t* rowp = qkv + (row0 + key) * qkv_stride;
      __builtin_amdgcn_global_load_lds((att_gptr_t)(rowp + (int64_t)(nq + hk) * ATT_D + k_unit * 8), (att_lptr_t)(kdst + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((att_gptr_t)(rowp + (int64_t)(nq + nkv + hk) * ATT_D + v_unit * 8), (att_lptr_t)(vdst + i * 1024), 16, 0, 0);
    }
  };

  f32x16_t oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // K fragment address: row = 32kb + (lane&31), d-slot = 2ks + hi, swizzle by row&15 == lane&15
  const int kf_row = ql * 256, kf_x = lane & 15;
  // V fragment (A operand of O^T += V^T P^T) straight from the ROW-MAJOR V image with ds_read_b64_tr_b16: in every 16-lane group lane j
  // points at V[k0 + j/4][d0 + 4 (j%4)] and lane c receives V[k0 .. k0+3][d0 + c] (the hardware transposes the group's 4 x 16 block);
  // d0 = 32db + 16 ((lane>>4)&1) makes c <-> the MFMA row lane&31, k0 = 32kb + 16c + 4hi (+8 for the second half of the k-slice)
  // (key & 3) of every key a lane addresses is (lane>>2)&3, so the V swizzle turns the d-block offset db*64 into (db ^ r)*64, r = (lane>>2)&3:
  // vt_lane carries r in byte bits 7:6 and the read address of d-block db is vt_lane ^ (db << 6)
  const int vt_lane = (((lane & 15) >> 2) + 4 * hi) * V_PITCH + ((((lane >> 2) & 3) * 4) << 4) + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;

  if (ntiles > 0) stage_tile(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed (this wave's share: vmcnt; everybody's: the barrier) and every wave is done reading the other stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 1 < ntiles) stage_tile(t + 1, (t + 1) & 1);
    const char* k_lds = smem + (t & 1) * ATT_STAGE_BYTES;
    const char* v_lds = k_lds + K_LDS_BYTES;

    // ---- S^T = K Q^T (scores for 64 keys x 32 q per wave)
    f32x16_t sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(k_lds + kb * 32 * 256 + kf_row + (((2 * ks + hi) ^ kf_x) << 4));
        // the first product takes the constant 0 as its accumulator input (an inline operand: no 16 v_mov per chain)
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? zero16 : sacc[kb], 0, 0, 0);
      }
    }

    // ---- mask + online softmax (all lane-local except one exchange with lane^32)
    uint64_t word;
    if constexpr (VARLEN) {
      const int rem = S - t * ATT_KB;
      word = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    } else {
      word = bits[t];
    }
    bool fast = (word == ~0ull);
    if constexpr (CAUSAL) {
      if (t * ATT_KB + ATT_KB - 1 > qb * ATT_QB + wave * 32) {   // tile reaches past this wave's first query: per-lane bound
        const int n = q_row - t * ATT_KB + 1;                    // keys of this tile the lane's query may see
        word &= n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
        fast = false;
      }
    }
    float mx = -INFINITY;
    if (fast) {                   // every key of the tile is valid (all tiles but a ragged last one): no per-element mask
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
    } else {
      const uint32_t wlo = (uint32_t)(word >> (4 * hi)), whi = (uint32_t)(word >> (32 + 4 * hi));
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const uint32_t wsel = kb ? whi : wlo;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kbit = (r & 3) + 8 * (r >> 2);  // key within the 32-block, minus 4*hi (already shifted)
          const float s = ((wsel >> kbit) & 1u) ? sacc[kb][r] : -INFINITY;
          sacc[kb][r] = s;
          mx = fmaxf(mx, s);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2;      // scale > 0: max commutes with the scaling
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
    m_run = m_new;
    float psum = 0.f;
    bf16x8_t pb[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        uint32_t pk[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          // exp2(s*scale - m): one fma + one v_exp per score (masked scores are -inf -> 0)
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sacc[kb][8 * c + 2 * jj], scale_log2, -m_use));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sacc[kb][8 * c + 2 * jj + 1], scale_log2, -m_use));
          psum += p0 + p1;
          pk[jj] = pack2bf_hw(p0, p1);
        }
        pb[kb][c] = __builtin_bit_cast(bf16x8_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;

    // ---- O^T += V^T P^T
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const char* vp = v_lds + (vt_lane ^ (db << 6)) + (kb * 32 + c * 16) * V_PITCH;
          const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp));
          const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 8 * V_PITCH));
          const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, (__attribute__((ext_vector_type(8))) short){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kb][c], oacc[db], 0, 0, 0);
        }
    }
  }

  // ---- epilogue: lane holds O[q][32db + 8g + 4hi + 0..3] in regs 4g..4g+3 of oacc[db]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  // 16-byte stores: v_permlane32_swap exchanges the two 32-lane halves of the register groups g and g+1, so that a lane ends up with
  // 8 CONSECUTIVE dims of its row (half 0: group g, half 1: group g+1) -- 8 stores per lane instead of 16 eight-byte ones (guide T21)
  uint4 ost[4][2];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      float a[4], bq[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(oacc[db][8 * gp + e] * inv_l), __float_as_uint(oacc[db][8 * gp + 4 + e] * inv_l),
                                                         false, false);
        a[e] = __uint_as_float(sw[0]); bq[e] = __uint_as_float(sw[1]);
      }
      ost[db][gp] = make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(bq[0], bq[1]), pack2bf(bq[2], bq[3]));
    }
  if (q_row < S) {
    uint16_t* op = out + (row0 + q_row) * out_stride + (int64_t)h * ATT_D + 8 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) *reinterpret_cast<uint4*>(op + db * 32 + gp * 16) = ost[db][gp];
    if (lse != nullptr && hi == 0) {
      if constexpr (VARLEN) lse[(row0 + q_row) * nq + h] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;  // [T, nq]
      else lse[((int64_t)b * nq + h) * S + q_row] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }
  }
}

}  // namespace grit

using namespace grit;

static int attn_fwd_padded(const char* name, bool causal, const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S,
                           int nq, int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  GRIT_REQUIRE(qkv && key_bits && out, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && S > 0 && nq > 0 && nkv > 0, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(d == ATT_D, GRIT_E_UNSUPPORTED, "%s: head_dim=%d (only 128 is built)", name, d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "%s: nq=%d not a multiple of nkv=%d", name, nq, nkv);
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "%s: bad strides", name);
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  GRIT_REQUIRE(nq <= 65535 && B <= 65535, GRIT_E_UNSUPPORTED, "%s: grid too large", name);
  const dim3 grid((unsigned)((S + ATT_QB - 1) / ATT_QB), (unsigned)nq, (unsigned)B);
  if (causal)
    hipLaunchKernelGGL((attn_bidir_fwd_k<false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, key_bits,
                       (const int32_t*)nullptr, (uint16_t*)out, lse, S, nq, nkv, qkv_stride, out_stride, scale * 1.4426950408889634f);
  else
    hipLaunchKernelGGL((attn_bidir_fwd_k<false, false>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, key_bits,
                       (const int32_t*)nullptr, (uint16_t*)out, lse, S, nq, nkv, qkv_stride, out_stride, scale * 1.4426950408889634f);
  GRIT_CHECK_LAUNCH(name);
  return GRIT_OK;
}

static int attn_fwd_varlen(const char* name, bool causal, const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len,
                           int nq, int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  GRIT_REQUIRE(qkv && cu_seqlens && out, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && max_len > 0 && nq > 0 && nkv > 0, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(d == ATT_D, GRIT_E_UNSUPPORTED, "%s: head_dim=%d (only 128 is built)", name, d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "%s: nq=%d not a multiple of nkv=%d", name, nq, nkv);
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "%s: bad strides", name);
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  GRIT_REQUIRE(nq <= 65535 && B <= 65535, GRIT_E_UNSUPPORTED, "%s: grid too large", name);
  const dim3 grid((unsigned)((max_len + ATT_QB - 1) / ATT_QB), (unsigned)nq, (unsigned)B);
  if (causal)
    hipLaunchKernelGGL((attn_bidir_fwd_k<true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, (const uint64_t*)nullptr,
                       cu_seqlens, (uint16_t*)out, lse, max_len, nq, nkv, qkv_stride, out_stride, scale * 1.4426950408889634f);
  else
    hipLaunchKernelGGL((attn_bidir_fwd_k<true, false>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)qkv, (const uint64_t*)nullptr,
                       cu_seqlens, (uint16_t*)out, lse, max_len, nq, nkv, qkv_stride, out_stride, scale * 1.4426950408889634f);
  GRIT_CHECK_LAUNCH(name);
  return GRIT_OK;
}

extern "C" int grit_attn_bidir_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv,
                                   int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_padded("grit_attn_bidir_fwd", false, qkv, key_bits, out, lse, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_causal_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv,
                                    int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_padded("grit_attn_causal_fwd", true, qkv, key_bits, out, lse, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_bidir_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                          int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_varlen("grit_attn_bidir_varlen_fwd", false, qkv, cu_seqlens, out, lse, B, max_len, nq, nkv, d, qkv_stride, out_stride, scale,
                         stream);
}
extern "C" int grit_attn_causal_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                           int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_varlen("grit_attn_causal_varlen_fwd", true, qkv, cu_seqlens, out, lse, B, max_len, nq, nkv, d, qkv_stride, out_stride, scale,
                         stream);
}
