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
// Both stream their 64-row tiles HBM -> LDS by direct LDS-DMA into a two-stage ring (the DMA of tile t+1 is issued right after the
// single barrier of tile t), keep ONE row-major image per operand and read it both as row fragments and, through
// ds_read_b64_tr_b16, as fragments of the transposed tile; gradients leave through an LDS transposition as full-line stores;
// workgroups that share their streamed operands sit on the same XCD (block v runs on XCD v % 8).
// As in the forward every product is arranged so that the owned index is the MFMA "column" (lane&31): the softmax
// statistics are lane-local (dq kernel) or a broadcast LDS read (dkdv kernel), and P / dS feed the next MFMA from
// registers with the contraction-index permutation applied on the transposed-LDS-image side.
#include <atomic>

#include "common.h"

namespace grit {

constexpr int AB_D = 128;
constexpr int AB_IMG = 64 * 256;              // one staged tile: row-major [64 rows][256 B]
constexpr int AB_STAGE = 2 * AB_IMG;          // two images per ring stage (Q + dO, or K + V)
constexpr int AB_RING = 2 * AB_STAGE;         // 64 KiB
typedef const __attribute__((address_space(1))) void* ab_gptr_t;
typedef __attribute__((address_space(3))) void* ab_lptr_t;
typedef __attribute__((ext_vector_type(4))) short ab_s16x4_t;

// s_waitcnt through the builtin (gfx9 encoding) so that the waitcnt insertion pass sees it (cf. attention.hip)
#define AB_WAIT_VM0()                       \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_waitcnt(0x0F70);     \
    asm volatile("" ::: "memory");          \
  } while (0)
#define AB_SCHED_FENCE()                    \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
#define AB_WAIT_LGKM0()                     \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_waitcnt(0xC07F);     \
    asm volatile("" ::: "memory");          \
  } while (0)

// Every staged image is read BOTH as row fragments (ds_read_b128: 16 rows x one 16-byte unit per lane group) and through the
// transposing ds_read_b64_tr_b16 (4 consecutive rows x 4 units per 32 lanes), so its 16-byte units are XOR-swizzled with
//   g(row) = ((row & 3) << 2) | ((row >> 2) & 3)
// -- a bijection of row & 15 (conflict-free row fragments) whose bits 3:2 enumerate 4 consecutive rows (conflict-free transposing
// reads).  The image of an LDS-DMA is lane-linear, so the swizzle goes into the per-lane SOURCE address.
//
// Stage rows r0 .. r0+63 (clamped to < limit) of a [rows][stride] bf16 matrix, 128 columns from `base`: wave w fills rows 16w..16w+15
// with four 1-KiB instructions (4 rows x 256 B each).
__device__ __forceinline__ void stage_img(const char* base, uint32_t stride_b, int r0, int limit, char* img, int wv, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = r0 + 16 * wv + 4 * i + (lane >> 4);
    row = row < limit ? row : limit - 1;
    const uint32_t unit = (uint32_t)((lane & 15) ^ ((((lane >> 4) & 3) << 2) | i));     // g(16w + 4i + (lane>>4))
    __builtin_amdgcn_global_load_lds((ab_gptr_t)(base + ((uint32_t)row * stride_b + (unit << 4))), (ab_lptr_t)(img + (16 * wv + 4 * i) * 256), 16, 0, 0);
  }
}
// A-operand fragment from the row-major image: A[row = rb*32 + (lane&31)][k = 16ks + 8hi + j]
__device__ __forceinline__ bf16x8_t frag_rm(const char* img, int rb, int ks, int lane) {
  const int row = rb * 32 + (lane & 31);
  const int gl = ((lane & 3) << 2) | ((lane >> 2) & 3);                                   // g(row): row & 15 == lane & 15
  return *reinterpret_cast<const bf16x8_t*>(img + row * 256 + (((2 * ks + (lane >> 5)) ^ gl) << 4));
}
// A-operand fragment of the TRANSPOSED tile straight from the row-major image (ds_read_b64_tr_b16, cf. attention.hip):
//   A[row = d = db*32 + (lane&31)][k = (hi,j)] = X[r = rb*32 + 16c + 8(j>>2) + 4hi + (j&3)][d]
// tr0 / tr1: per-lane byte offsets of the two reads (rows +0 / +8) for db = 0, without the tile-row offset
struct TrLane {
  int o0, o1;
};
__device__ __forceinline__ TrLane tr_lane(int lane) {
  const int hi = lane >> 5, i = (lane & 15) >> 2, a = (lane >> 4) & 1, b = (lane & 3) >> 1;
  TrLane t;
  t.o0 = (4 * hi + i) * 256 + (i << 6) + ((((2 * a + b) ^ hi)) << 4) + 8 * (lane & 1);             // g: bits 3:2 = i, bits 1:0 = (row>>2)&3 = hi
  t.o1 = (8 + 4 * hi + i) * 256 + (i << 6) + ((((2 * a + b) ^ (2 + hi))) << 4) + 8 * (lane & 1);   //                         ... = 2 + hi
  return t;
}
__device__ __forceinline__ bf16x8_t frag_tr(const char* img, int db, int rb, int c, const TrLane& t) {
  const char* p = img + (rb * 32 + c * 16) * 256;
  const ab_s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ab_s16x4_t*)(p + (t.o0 ^ (db << 6))));
  const ab_s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ab_s16x4_t*)(p + (t.o1 ^ (db << 6))));
  return __builtin_bit_cast(bf16x8_t, (__attribute__((ext_vector_type(8))) short){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
}
// Store a wave's [32 rows][128] fp32 accumulator tile (MFMA "transposed" layout: lane (row = lane&31, hi) holds
// X[row][32db + 8g + 4hi + 0..3] in regs 4g..4g+3 of acc[db]) as bf16 with FULL-LINE stores: v_permlane32_swap gives every lane
// 16-byte pieces, which go through an 8 KiB LDS buffer (unit ^= row & 15) and leave as 4 rows x 256 B per instruction.
__device__ __forceinline__ void store_tile_rows(const f32x16_t (&acc)[4], float mul, char* xs, char* gbase, uint32_t stride_b, int row_first,
                                                int limit, int lane) {
  const int ql = lane & 31, hi = lane >> 5;
  char* wp = xs + ql * 256;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      float a[4], bq[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[db][8 * gp + e] * mul), __float_as_uint(acc[db][8 * gp + 4 + e] * mul),
                                                         false, false);
        a[e] = __uint_as_float(sw[0]); bq[e] = __uint_as_float(sw[1]);
      }
      *reinterpret_cast<uint4*>(wp + (((db * 4 + gp * 2 + hi) ^ (ql & 15)) << 4)) =
          make_uint4(pack2bf_hw(a[0], a[1]), pack2bf_hw(a[2], a[3]), pack2bf_hw(bq[0], bq[1]), pack2bf_hw(bq[2], bq[3]));
    }
  AB_WAIT_LGKM0();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 4 * j + (lane >> 4);
    const uint4 piece = *reinterpret_cast<const uint4*>(xs + j * 1024 + lane * 16);
    if (row_first + r < limit)
      *reinterpret_cast<uint4*>(gbase + ((uint32_t)(row_first + r) * stride_b + (uint32_t)(((lane & 15) ^ (r & 15)) << 4))) = piece;
  }
  AB_WAIT_LGKM0();
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
// Workgroups of one (batch, kv head) -- its key blocks -- stream the same Q / dO tiles and sit on the same XCD.
// CW = false (round 6): bidirectional attention -- the embedding tower of the contrastive step -- where `causal` and `window` are known to be
// 0 and a score's visibility is the key's validity alone; with the run-time flags the loop evaluated two compares and two mask merges
// per score (~130 of ~800 instructions per tile).  Together with the buffer-addressed staging below: 943 -> 677 us for the three backward
// launches of a 32 x 512 chunk (profiles/r06_attn_bwd_split_ab.json, "fused"; same bits).
// (Also measured this round and NOT kept -- tools/ubench/attn_bwd_variants/: the loop specialised to dV only / dK only, each within 256
//  registers so that two workgroups share a CU: bit-identical, 0.94-0.96x of this form -- the pair executes 80 products per tile
//  instead of 64 and streams the Q / dO tiles twice.)
template <bool VARLEN, bool CW = true>
__global__ void __launch_bounds__(256)
attn_bwd_dkdv_k(const uint16_t* __restrict__ qkv, const uint64_t* __restrict__ key_bits, const int32_t* __restrict__ cu_seqlens,
                const uint16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
                uint16_t* __restrict__ dqkv, int S_arg, int nq, int nkv, int64_t qkv_stride, int64_t out_stride, float scale, int causal,
                int nkb, int n_sets, int window) {
  extern __shared__ __attribute__((aligned(256))) char smem[];          // 256: the asm read addresses OR / XOR lane constants into bits 7:4

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int kx = (int)blockIdx.x >> 3;
  const int set = (kx / nkb) * 8 + ((int)blockIdx.x & 7), kblk = kx % nkb;
  if (set >= n_sets) return;
  const int b = set / nkv, hk = set - b * nkv;
  const int group = nq / nkv;
  int S = S_arg;
  int64_t row0 = (int64_t)b * S_arg;
  if constexpr (VARLEN) {
    row0 = cu_seqlens[b];
    S = cu_seqlens[b + 1] - cu_seqlens[b];
  }
  if (kblk * 128 >= S) return;
  const int key = kblk * 128 + wave * 32 + (lane & 31);
  bool key_ok = key < S;
  if constexpr (!VARLEN) {
    const int W = (S + 63) >> 6;
    key_ok = key_ok && ((key_bits[(int64_t)b * W + (key >> 6)] >> (key & 63)) & 1ull);
  }
  const int key_ld = key < S ? key : S - 1;
  const bool all_keys_ok = __builtin_amdgcn_ballot_w64(!key_ok) == 0ull;               // wave-uniform
  const float scale_log2 = scale * 1.4426950408889634f;
  const uint32_t qkv_stride_b = (uint32_t)qkv_stride * 2u, out_stride_b = (uint32_t)out_stride * 2u;
  const char* seq_qkv = reinterpret_cast<const char*>(qkv + row0 * qkv_stride);      // workgroup-uniform bases, 32-bit per-lane offsets
  const char* seq_do = reinterpret_cast<const char*>(dout + row0 * out_stride);

  // flat tile list: the query tiles (qt0 .. nqt-1) of each of the group's heads
  int nqt = (S + 63) >> 6;
  const int qt0 = causal ? 2 * kblk : 0;           // causal: query tiles before this key block see none of its keys
  if (window > 0) {                                // sliding window: nor do the tiles behind query (last key of the block) + window - 1
    const int qt_end = ((kblk * 128 + 126 + window) >> 6) + 1;
    nqt = nqt < qt_end ? nqt : qt_end;
  }
  const int per_head = nqt - qt0;
  const int ntl = group * per_head;
  auto tile_of = [&](int n, int& h, int& qt) {
    const int g = n / per_head;
    qt = qt0 + (n - g * per_head);
    h = hk * group + g;
  };
  // Buffer-addressed LDS-DMA (round 6, as the forward kernel): the descriptor's range check zero-fills rows past the sequence (their
  // statistics are lse = +inf -> P = 0, so the row's content never matters) -- no per-tile clamp / 64-bit multiply per piece.  The lane's
  // offset of piece i is [(lane>>4) rows + the swizzled unit of piece 0] ^ (i << 4): the strides are multiples of 256 B, so the unit sits
  // alone in bits 7:4 and g(row) of piece i differs from piece 0's in bits 5:4 only -- one v_xor per piece from two values that are
  // recomputed from a laundered lane id in every call (kept across the tile loop they are the first registers hipcc spills, and a
  // scratch reload's vmcnt wait would sit between the DMA pieces); the piece's 4 i rows, the wave's 16 rows, the tile and the head
  // go into the scalar offset.
  const auto q_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(seq_qkv), (short)0, (int)((uint32_t)(S - 1) * qkv_stride_b + (uint32_t)nq * AB_D * 2u), 0x00020000);
  const auto do_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(seq_do), (short)0, (int)((uint32_t)(S - 1) * out_stride_b + (uint32_t)nq * AB_D * 2u), 0x00020000);
  auto stage = [&](int n, int buf) {
    int h, qt;
    tile_of(n, h, qt);
    char* sb = smem + buf * AB_STAGE + wv * 4096;
    const uint32_t hq = (uint32_t)__builtin_amdgcn_readfirstlane(h * AB_D * 2), rq = (uint32_t)__builtin_amdgcn_readfirstlane(qt * 64 + 16 * wv);
    const uint32_t off_q = rq * qkv_stride_b + hq, off_d = rq * out_stride_b + hq;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const uint32_t u0_b = (uint32_t)(((ln & 15) ^ (((ln >> 4) & 3) << 2)) << 4);          // g(16w + (lane>>4)) << 4, cf. stage_img
    const uint32_t xq = (uint32_t)(ln >> 4) * qkv_stride_b + u0_b, xd = (uint32_t)(ln >> 4) * out_stride_b + u0_b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (ab_lptr_t)(sb + i * 1024), 16, (int)(xq ^ (uint32_t)(i << 4)), (int)(off_q + 4u * i * qkv_stride_b), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(do_rsrc, (ab_lptr_t)(sb + AB_IMG + i * 1024), 16, (int)(xd ^ (uint32_t)(i << 4)), (int)(off_d + 4u * i * out_stride_b), 0, 0);
    }
  };
  if (ntl > 0) stage(0, 0);

  // K, V fragments of the owned key (B operands): X[key][16ks + 8hi .. +8]
  bf16x8_t kf[8], vf[8];
  {
    const char* kp = seq_qkv + ((uint32_t)key_ld * qkv_stride_b + (uint32_t)((nq + hk) * AB_D * 2 + hi * 16));
    const char* vp = kp + nkv * AB_D * 2;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8_t*>(kp + ks * 32);
      vf[ks] = *reinterpret_cast<const bf16x8_t*>(vp + ks * 32);
    }
  }
  f32x16_t dk[4], dv[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
  const TrLane trl = tr_lane(lane);

  // statistics of a tile's 64 queries: every wave keeps its own copy, lane l <-> query qt*64 + l (one coalesced load per array), and
  // hands it to the lanes that need it through its 512-byte LDS table (DK_STATS below; rounds 1-5: 32 ds_bpermute per half).  Rows past the
  // sequence get lse = +inf -> P = 0.
  auto load_stats = [&](int n_, float& l_, float& d_) {
    int h_s, qt_s;
    tile_of(n_, h_s, qt_s);
    const int qc = qt_s * 64 + lane < S ? qt_s * 64 + lane : S - 1;
    // padded layout [B,nq,S] (stride 1 over q), packed layout [T,nq] (stride nq over q)
    const float* lrow = VARLEN ? lse + row0 * nq + h_s : lse + ((int64_t)b * nq + h_s) * S;
    const float* drow = VARLEN ? delta + row0 * nq + h_s : delta + ((int64_t)b * nq + h_s) * S;
    const int64_t qs = VARLEN ? nq : 1;
    l_ = lrow[qc * qs];
    d_ = drow[qc * qs];
  };
  // the statistics of tile n + 1 are requested during tile n, BEHIND its LDS-DMA, and are covered by the vmcnt(0) at the top of tile
  // n + 1 -- requested at the top of their own tile, hipcc's wait for them (vmcnt(0): the DMA sits in a conditional branch, so it
  // cannot count) would pull the whole DMA of the next tile in front of the first softmax
  float next_l = 0.f, next_d = 0.f;
  if (ntl > 0) load_stats(0, next_l, next_d);
  for (int n = 0; n < ntl; ++n) {
    AB_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int h_t, qt;
    tile_of(n, h_t, qt);
    float raw_l, raw_d, my_l = 0.f, my_d = 0.f;
    const bool q_in = qt * 64 + lane < S;
    raw_l = next_l; raw_d = next_d;
    const char* q_img = smem + (n & 1) * AB_STAGE;
    const char* d_img = q_img + AB_IMG;
    // Round 3, second half: every LDS fragment read of the tile is inline asm with counted lgkmcnt (cf. attention.hip).  What it buys here:
    //  * the row fragments stream through THREE register sets two k-slices ahead of their products (24 registers instead of the 64 of a
    //    whole half requested up front), the transposed fragments come in two groups of 8 (the second one in flight under the first one's
    //    products): 96 registers of fragments in flight instead of 128, every read waited for exactly when its product needs it;
    //  * no ds_read_b64_tr_b16 builtin follows an LDS-DMA any more (hipcc puts a vmcnt(0) in front of each), so the next tile's DMA goes
    //    out at the TOP of the tile instead of behind its last transposing read: a whole tile of flight instead of a quarter.
    if (n + 1 < ntl) {
      stage(n + 1, (n + 1) & 1);
      load_stats(n + 1, next_l, next_d);
    }
    {
      const uint32_t img_b = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)q_img);
      // row fragments: (2 ks + hi) ^ g == (2 ks) ^ (hi ^ g) -> the lane's constant sits in address bits 7:4, a k-slice is one v_xor
      const uint32_t a_base = (img_b + (uint32_t)((lane & 31) * 256)) | (uint32_t)((hi ^ (((lane & 3) << 2) | ((lane >> 2) & 3))) << 4);
      uint32_t t0[4], t1[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) { t0[db] = img_b + (uint32_t)(trl.o0 ^ (db << 6)); t1[db] = img_b + (uint32_t)(trl.o1 ^ (db << 6)); }
      bf16x8_t fa0[3], fa1[3];
      ab_s16x4_t fbr[2][4][2][2];                                // [group c][db][operand][half]
      f32x4_t lq[4], dq4[4];                                     // statistics of the half's rows 8j + 4hi .. +3
      const uint32_t st_base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)(smem + AB_RING)) + (uint32_t)wv * 512u;
      const uint32_t st_w = st_base + (uint32_t)lane * 4u;
      const uint32_t st_r = st_base + (uint32_t)hi * 16u;
      const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16_t s, dp;
#define DK_A_READ(QB, KS)                                                                                                         \
  do {                                                                                                                            \
    const uint32_t ka = a_base ^ (uint32_t)((KS) << 5);                                                                           \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa0[(KS) % 3]) : "v"(ka), "i"((QB) * 8192));                              \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa1[(KS) % 3]) : "v"(ka), "i"((QB) * 8192 + AB_IMG));                     \
  } while (0)
  // (The products as inline asm with the register FILE of every accumulator fixed -- S / dP "+v", dK / dV "+a" -- remove ALL ~540
  //  v_accvgpr moves per tile that hipcc generates around its one-form-for-all MFMA selection; built, bit-identical, and measured EQUAL
  //  to this form (+1.9 % at S 512, -1.5 % at S 2048: profiles/r03_attn_bwd_ab_asm_reads.log): the moves ride in issue slots the
  //  matrix pipe leaves free.  Not kept.)
#define DK_MFMA_V0(D, A, B) D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, zero16, 0, 0, 0)
#define DK_MFMA_V(D, A, B) D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, D, 0, 0, 0)
#define DK_MFMA_A(D, A, B) D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, D, 0, 0, 0)
#define DK_A_MMA(KS, N)                                                                                                           \
  do {                                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fa0[(KS) % 3]), "+v"(fa1[(KS) % 3]) : : "memory");                            \
    if ((KS) == 0) { DK_MFMA_V0(s, fa0[(KS) % 3], kf[KS]); DK_MFMA_V0(dp, fa1[(KS) % 3], vf[KS]); }                               \
    else { DK_MFMA_V(s, fa0[(KS) % 3], kf[KS]); DK_MFMA_V(dp, fa1[(KS) % 3], vf[KS]); }                                           \
  } while (0)
#define DK_B_READ1(QB, C, DB, OP)                                                                                                 \
  do {                                                                                                                            \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fbr[C][DB][OP][0]) : "v"(t0[DB]), "i"(((QB) * 32 + (C) * 16) * 256 + (OP) * AB_IMG)); \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fbr[C][DB][OP][1]) : "v"(t1[DB]), "i"(((QB) * 32 + (C) * 16) * 256 + (OP) * AB_IMG)); \
  } while (0)
#define DK_B_READ(QB, C)                                                                                                          \
  do {                                                                                                                            \
    DK_B_READ1(QB, C, 0, 0); DK_B_READ1(QB, C, 0, 1); DK_B_READ1(QB, C, 1, 0); DK_B_READ1(QB, C, 1, 1);                           \
    DK_B_READ1(QB, C, 2, 0); DK_B_READ1(QB, C, 2, 1); DK_B_READ1(QB, C, 3, 0); DK_B_READ1(QB, C, 3, 1);                           \
  } while (0)
#define DK_B_WAIT(C, N)                                                                                                           \
  do {                                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                                      \
                 : "+v"(fbr[C][0][0][0]), "+v"(fbr[C][0][0][1]), "+v"(fbr[C][0][1][0]), "+v"(fbr[C][0][1][1]), "+v"(fbr[C][1][0][0]), \
                   "+v"(fbr[C][1][0][1]), "+v"(fbr[C][1][1][0]), "+v"(fbr[C][1][1][1]) : : "memory");                             \
    asm volatile(""                                                                                                               \
                 : "+v"(fbr[C][2][0][0]), "+v"(fbr[C][2][0][1]), "+v"(fbr[C][2][1][0]), "+v"(fbr[C][2][1][1]), "+v"(fbr[C][3][0][0]), \
                   "+v"(fbr[C][3][0][1]), "+v"(fbr[C][3][1][0]), "+v"(fbr[C][3][1][1]));                                          \
  } while (0)
#define DK_FRAG(C, DB, OP)                                                                                                        \
  __builtin_bit_cast(bf16x8_t, (__attribute__((ext_vector_type(8))) short){fbr[C][DB][OP][0][0], fbr[C][DB][OP][0][1], fbr[C][DB][OP][0][2], \
                                                                           fbr[C][DB][OP][0][3], fbr[C][DB][OP][1][0], fbr[C][DB][OP][1][1], \
                                                                           fbr[C][DB][OP][1][2], fbr[C][DB][OP][1][3]})
#define DK_B_MMA(C)                                                                                                               \
  _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                                              \
    const bf16x8_t f1_ = DK_FRAG(C, db, 1), f0_ = DK_FRAG(C, db, 0);                                                              \
    DK_MFMA_A(dv[db], f1_, pb[C]);                                                                                                \
    DK_MFMA_A(dk[db], f0_, dsb[C]);                                                                                               \
  }
// The 64 queries' statistics (lse, delta) reach the lanes that need them -- 16 rows per half and lane, the same for every key column --
// through a 512-byte per-wave LDS table (round 6): lane l writes the pair of query l once per tile, every lane reads its 4 + 4 aligned
// 16-byte groups per half (queries 8j + 4hi .. +3) with ds_read_b128.  8 LDS instructions per half instead of 32 ds_bpermute, requested
// IN FRONT of the half's transposing reads and waited for with a counted lgkmcnt, so those stay in flight under the softmax.
// Inline asm like every LDS access of the loop (a C++ LDS load behind an LDS-DMA gets a vmcnt(0) from hipcc's wait-count pass).
#define DK_ST1(QB, J)                                                                                                             \
  do {                                                                                                                            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lq[J]) : "v"(st_r), "i"((QB) * 128 + (J) * 32));                          \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dq4[J]) : "v"(st_r), "i"((QB) * 128 + (J) * 32 + 256));                   \
  } while (0)
#define DK_STATS(QB)                                                                                                              \
  do {                                                                                                                            \
    if ((QB) == 0) {                                                                                                              \
      my_l = q_in ? raw_l * 1.4426950408889634f : INFINITY;                                                                       \
      my_d = q_in ? raw_d : 0.f;                                                                                                  \
      asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:256" : : "v"(st_w), "v"(my_l), "v"(my_d) : "memory");       \
    }                                                                                                                             \
    DK_ST1(QB, 0); DK_ST1(QB, 1); DK_ST1(QB, 2); DK_ST1(QB, 3);                                                                   \
  } while (0)
#define DK_SOFTMAX(QB)                                                                                                            \
  do {                                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(15)"      /* (the counter saturates at 15) 16 transposing reads were requested behind the statistics */ \
                 : "+v"(lq[0]), "+v"(lq[1]), "+v"(lq[2]), "+v"(lq[3]), "+v"(dq4[0]), "+v"(dq4[1]), "+v"(dq4[2]), "+v"(dq4[3]) : : "memory"); \
    if (!CW && all_keys_ok) {             /* every key of the wave is valid (all blocks but a ragged last one): no select per score */ \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                            \
        const float e = __builtin_amdgcn_exp2f(s[r] * scale_log2 - lq[r >> 2][r & 3]);                                            \
        s[r] = e;                                                                                                                 \
        dp[r] = e * (dp[r] - dq4[r >> 2][r & 3]);                                                                                 \
      }                                                                                                                           \
    } else {                                                                                                                      \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                              \
      const int ql = (QB) * 32 + 8 * (r >> 2) + (r & 3);                                                                          \
      const int qd = qt * 64 + ql + 4 * hi - key;                                                                                 \
      const bool seen = CW ? (key_ok & ((causal == 0) | (qd >= 0)) & ((window == 0) | (qd < window))) : key_ok;                   \
      const float e = __builtin_amdgcn_exp2f(s[r] * scale_log2 - lq[r >> 2][r & 3]);                                              \
      const float p = seen ? e : 0.f;                                                                                             \
      s[r] = p;                                                                                                                   \
      dp[r] = p * (dp[r] - dq4[r >> 2][r & 3]);                                                                                   \
    }                                                                                                                             \
    }                                                                                                                             \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) { pb[c] = pack8(s, c); dsb[c] = pack8(dp, c); }                                 \
  } while (0)
#define DK_HALF(QB)                                                                                                               \
  do {                                                                                                                            \
    DK_A_READ(QB, 2); DK_A_MMA(0, 4);                                                                                             \
    DK_A_READ(QB, 3); DK_A_MMA(1, 4);                                                                                             \
    DK_A_READ(QB, 4); DK_A_MMA(2, 4);                                                                                             \
    DK_A_READ(QB, 5); DK_A_MMA(3, 4);                                                                                             \
    DK_A_READ(QB, 6); DK_A_MMA(4, 4);                                                                                             \
    DK_A_READ(QB, 7); DK_A_MMA(5, 4);                                                                                             \
    DK_A_MMA(6, 2); DK_A_MMA(7, 0);                                                                                               \
    DK_STATS(QB);                          /* the rows' statistics, then ... */                                                   \
    DK_B_READ(QB, 0);                      /* ... 16 transposing reads in flight under the softmax */                             \
    bf16x8_t pb[2], dsb[2];                                                                                                       \
    DK_SOFTMAX(QB);                                                                                                               \
    DK_B_WAIT(0, 0);                                                                                                              \
    DK_B_READ(QB, 1);                      /* ... the second group under the first group's products */                           \
    DK_B_MMA(0)                                                                                                                   \
    if ((QB) == 0) { DK_A_READ(1, 0); DK_A_READ(1, 1); DK_B_WAIT(1, 4); } else { DK_B_WAIT(1, 0); }                               \
    DK_B_MMA(1)                                                                                                                   \
  } while (0)
      DK_A_READ(0, 0); DK_A_READ(0, 1);
      DK_HALF(0);
      DK_HALF(1);
#undef DK_A_READ
#undef DK_A_MMA
#undef DK_MFMA_V0
#undef DK_MFMA_V
#undef DK_MFMA_A
#undef DK_B_READ1
#undef DK_B_READ
#undef DK_B_WAIT
#undef DK_FRAG
#undef DK_B_MMA
#undef DK_SOFTMAX
#undef DK_STATS
#undef DK_ST1
#undef DK_HALF
    }
  }
  // every wave is done with the ring: it becomes the transposition buffer of the full-line gradient stores
  AB_WAIT_VM0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  char* xs = smem + wv * 8192;
  char* seq_dqkv = reinterpret_cast<char*>(dqkv + row0 * qkv_stride);
  store_tile_rows(dk, scale, xs, seq_dqkv + (int64_t)(nq + hk) * AB_D * 2, qkv_stride_b, kblk * 128 + wave * 32, S, lane);
  store_tile_rows(dv, 1.0f, xs, seq_dqkv + (int64_t)(nq + nkv + hk) * AB_D * 2, qkv_stride_b, kblk * 128 + wave * 32, S, lane);
}

// ------------------------------------------------------------------ dQ
// Workgroups of one (batch, kv head) -- the group's heads x query blocks -- stream the same K / V tiles and sit on the same XCD.
template <bool VARLEN>
__global__ void __launch_bounds__(256, 2)
attn_bwd_dq_k(const uint16_t* __restrict__ qkv, const uint64_t* __restrict__ key_bits, const int32_t* __restrict__ cu_seqlens,
              const uint16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ delta,
              uint16_t* __restrict__ dqkv, int S_arg, int nq, int nkv, int64_t qkv_stride, int64_t out_stride, float scale, int causal,
              int nqb, int n_sets, int window) {
  __shared__ __attribute__((aligned(256))) char smem[AB_RING];   // 256: the asm read addresses OR the lane constant into bits 7:4

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int gqa = nq / nkv, U = gqa * nqb;
  const int kx = (int)blockIdx.x >> 3;
  const int set = (kx / U) * 8 + ((int)blockIdx.x & 7), member = kx % U;
  if (set >= n_sets) return;
  const int b = set / nkv, hk = set - b * nkv;
  const int h = hk * gqa + member % gqa, qblk = member / gqa;
  int S = S_arg;
  int64_t row0 = (int64_t)b * S_arg;
  if constexpr (VARLEN) {
    row0 = cu_seqlens[b];
    S = cu_seqlens[b + 1] - cu_seqlens[b];
  }
  if (qblk * 128 >= S) return;
  const uint32_t qkv_stride_b = (uint32_t)qkv_stride * 2u, out_stride_b = (uint32_t)out_stride * 2u;
  const char* seq_qkv = reinterpret_cast<const char*>(qkv + row0 * qkv_stride);
  const char* k_base = seq_qkv + (int64_t)(nq + hk) * AB_D * 2;
  // buffer-addressed LDS-DMA (round 6, cf. attn_bwd_dkdv_k): ONE descriptor serves K and V (same row stride; V sits v_delta bytes behind
  // K, a scalar), rows past the sequence are zero-filled by the range check (their keys are masked by the tile's key word)
  const uint32_t v_delta_b = (uint32_t)nkv * AB_D * 2u;
  const auto kv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(k_base), (short)0, (int)((uint32_t)(S - 1) * qkv_stride_b + 256u + v_delta_b), 0x00020000);
  auto stage = [&](int t, int buf) {
    char* sb = smem + buf * AB_STAGE + wv * 4096;
    const uint32_t off_k = (uint32_t)__builtin_amdgcn_readfirstlane(t * 64 + 16 * wv) * qkv_stride_b;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const uint32_t xk = (uint32_t)(ln >> 4) * qkv_stride_b + (uint32_t)(((ln & 15) ^ (((ln >> 4) & 3) << 2)) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t vo = xk ^ (uint32_t)(i << 4), so = off_k + 4u * i * qkv_stride_b;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(kv_rsrc, (ab_lptr_t)(sb + i * 1024), 16, (int)vo, (int)so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(kv_rsrc, (ab_lptr_t)(sb + AB_IMG + i * 1024), 16, (int)vo, (int)(so + v_delta_b), 0, 0);
    }
  };
  // first K/V tile: 0 (always exists: S > 0), or the one holding the first key of the block's first query's window
  const int t_first = window > 0 && qblk * 128 - window + 1 > 0 ? (qblk * 128 - window + 1) >> 6 : 0;
  stage(t_first, t_first & 1);

  const uint64_t* bits = nullptr;
  int ntiles = 0;
  if constexpr (VARLEN) {
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
    const char* qp = seq_qkv + ((uint32_t)q_ld * qkv_stride_b + (uint32_t)(h * AB_D * 2 + hi * 16));
    const char* dp = reinterpret_cast<const char*>(dout + row0 * out_stride) + ((uint32_t)q_ld * out_stride_b + (uint32_t)(h * AB_D * 2 + hi * 16));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32);
      dof[ks] = *reinterpret_cast<const bf16x8_t*>(dp + ks * 32);
    }
  }
  f32x16_t dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
  const TrLane trl = tr_lane(lane);

  // padded layout: the mask word of tile t + 1 is fetched at the end of tile t (a scalar load issued at the top of its own tile is
  // waited for right there -- hipcc needs it for the causal bounds -- with its whole round trip exposed behind the barrier)
  uint64_t word_next = 0;
  if constexpr (!VARLEN) {
    if (t_first < ntiles) word_next = bits[t_first];
  }
  for (int t = t_first; t < ntiles; ++t) {
    AB_WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* k_img = smem + (t & 1) * AB_STAGE;
    const char* v_img = k_img + AB_IMG;
    uint64_t word;
    if constexpr (VARLEN) {
      const int rem = S - t * 64;
      word = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    } else {
      word = word_next;
    }
    if (causal) {
      const int n = q - t * 64 + 1;
      word &= n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
      if (window > 0) {
        const int lo = n - window;                  // keys of this tile in front of the query's window
        word &= lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
      }
    }
    // wave-uniform: without causal bounds `word` is the tile's key word (a scalar); with them it is per lane
    const bool all_keys = !causal && __builtin_amdgcn_readfirstlane((int)(word == ~0ull)) != 0;
    const uint32_t wlo = (uint32_t)(word >> (4 * hi)), whi = (uint32_t)(word >> (32 + 4 * hi));
    // (The same inline-asm read scheme as attn_bwd_dkdv_k -- row fragments two k-slices ahead, the transposing reads in one counted group,
    //  the next tile's DMA at the tile top -- was built for this loop too: bit-identical, 1.017 / 1.000 / 0.996 / 0.994 of this form on
    //  the four harness shapes (profiles/r03_attn_bwd_ab_asm_reads.log).  With two waves per SIMD the partner wave covers what the
    //  pinned groups below leave open; not kept.)
    // hipcc sinks every fragment read to its use and waits for each (one read in flight); the reads are therefore issued in pinned
    // groups a group ahead of the products that consume them (two waves per SIMD leave ~30 registers for that: 2 k-slices x {K, V}
    // per group, double-buffered), and the transposed-K fragments of the dQ products go out ahead of the softmax arithmetic
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      bf16x8_t fa[2][4];
      auto read_a = [&](int g, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          fa[buf][2 * j] = frag_rm(k_img, kb, 2 * g + j, lane);
          fa[buf][2 * j + 1] = frag_rm(v_img, kb, 2 * g + j, lane);
        }
      };
      auto mma_a = [&](int g, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][2 * j], qf[2 * g + j], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][2 * j + 1], dof[2 * g + j], dp, 0, 0, 0);
        }
      };
      read_a(0, 0);
      AB_SCHED_FENCE();
      read_a(1, 1);
      AB_SCHED_FENCE();
      mma_a(0, 0);
      read_a(2, 0);
      AB_SCHED_FENCE();
      mma_a(1, 1);
      read_a(3, 1);
      AB_SCHED_FENCE();
      mma_a(2, 0);
      AB_SCHED_FENCE();
      mma_a(3, 1);
      bf16x8_t fb[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int db = 0; db < 4; ++db) fb[c][db] = frag_tr(k_img, db, kb, c, trl);
      AB_SCHED_FENCE();
      // the next tile's DMA goes out behind the tile's LAST transposing reads (hipcc waits vmcnt(0) in front of every
      // ds_read_b64_tr_b16 that follows an LDS-DMA)
      if (kb == 1 && t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
      const uint32_t wsel = kb ? whi : wlo;
      if (all_keys) {             // (round 6) every key of the tile is visible to every query of the wave -- all tiles of a bidirectional
#pragma unroll                    // pass but a ragged last one: no per-score bit test (a shift, a compare and a select per score: ~40 % of
        for (int r = 0; r < 16; ++r) {                 // the loop's VALU instructions); same bits as the masked form with all bits set
          const float e = __builtin_amdgcn_exp2f(s[r] * scale_log2 - lse2);
          dp[r] = e * (dp[r] - dlt);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kbit = (r & 3) + 8 * (r >> 2);
          const float e = __builtin_amdgcn_exp2f(s[r] * scale_log2 - lse2);   // unconditional: a select, not a branch per element
          const float p = ((wsel >> kbit) & 1u) ? e : 0.f;
          dp[r] = p * (dp[r] - dlt);
        }
      }
      bf16x8_t dsb[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) dsb[c] = pack8(dp, c);
      AB_SCHED_FENCE();
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int db = 0; db < 4; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[c][db], dsb[c], dq[db], 0, 0, 0);
    }
    // (requested at the END of the tile: while a scalar load is outstanding hipcc turns every counted LDS wait into lgkmcnt(0), so it
    //  must not sit inside the pinned read groups; here its round trip overlaps the wait for the next tile's DMA and the barrier)
    if constexpr (!VARLEN) word_next = bits[t + 1 < ntiles ? t + 1 : t];
  }
  AB_WAIT_VM0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  store_tile_rows(dq, scale, smem + wv * 8192, reinterpret_cast<char*>(dqkv + row0 * qkv_stride) + (int64_t)h * AB_D * 2, qkv_stride_b,
                  qblk * 128 + wave * 32, S, lane);
}

}  // namespace grit

using namespace grit;

static int attn_bwd_launch(bool varlen, int causal, int window, const void* qkv, const uint64_t* key_bits, const int32_t* cu, const void* out, const void* dout,
                           const float* lse, float* delta, void* dqkv, int B, int S_or_maxlen, int64_t T, int nq, int nkv,
                           int64_t qkv_stride, int64_t out_stride, float scale, hipStream_t st) {
  const int64_t items = T * nq;
  hipLaunchKernelGGL(attn_delta_k, dim3((unsigned)((items * 16 + 255) / 256)), dim3(256), 0, st, (const uint16_t*)out,
                     (const uint16_t*)dout, delta, T, varlen ? 0 : S_or_maxlen, nq, out_stride);
  GRIT_CHECK_LAUNCH("grit_attn_bidir_bwd: delta");
  const int lds_kv = AB_RING + 4 * 512;          // ring + the four waves' statistics tables (attn_bwd_dkdv_k)
  static std::atomic<uint64_t> optin{0};                       // per-device function attribute
  {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(optin.load(std::memory_order_acquire) & bit)) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_k<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_k<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_k<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_k<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
      optin.fetch_or(bit, std::memory_order_release);
    }
  }
  const int nblk = (S_or_maxlen + 127) / 128;
  const int n_sets = B * nkv, sets8 = 8 * ((n_sets + 7) / 8);
  const dim3 grid_kv((unsigned)(sets8 * nblk)), grid_q((unsigned)(sets8 * (nq / nkv) * nblk));
#define AB_LAUNCH_KV(VL, CWF)                                                                                                       \
  hipLaunchKernelGGL((attn_bwd_dkdv_k<VL, CWF>), grid_kv, dim3(256), lds_kv, st, (const uint16_t*)qkv, key_bits, cu, (const uint16_t*)dout, lse,  \
                     delta, (uint16_t*)dqkv, S_or_maxlen, nq, nkv, qkv_stride, out_stride, scale, causal, nblk, n_sets, window)
  const bool cw = causal != 0 || window > 0;
  if (varlen) {
    if (cw) { AB_LAUNCH_KV(true, true); } else { AB_LAUNCH_KV(true, false); }
    GRIT_CHECK_LAUNCH("grit_attn_bidir_varlen_bwd: dkdv");
    hipLaunchKernelGGL(attn_bwd_dq_k<true>, grid_q, dim3(256), 0, st, (const uint16_t*)qkv, key_bits, cu,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, S_or_maxlen, nq, nkv, qkv_stride, out_stride, scale, causal, nblk, n_sets, window);
    GRIT_CHECK_LAUNCH("grit_attn_bidir_varlen_bwd: dq");
  } else {
    if (cw) { AB_LAUNCH_KV(false, true); } else { AB_LAUNCH_KV(false, false); }
    GRIT_CHECK_LAUNCH("grit_attn_bidir_bwd: dkdv");
    hipLaunchKernelGGL(attn_bwd_dq_k<false>, grid_q, dim3(256), 0, st, (const uint16_t*)qkv, key_bits, cu,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, S_or_maxlen, nq, nkv, qkv_stride, out_stride, scale, causal, nblk, n_sets, window);
    GRIT_CHECK_LAUNCH("grit_attn_bidir_bwd: dq");
  }
#undef AB_LAUNCH_KV
  return GRIT_OK;
}

static int attn_bwd_padded(int causal, int window, const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                           float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                           int64_t out_stride, float scale, void* stream) {
  GRIT_REQUIRE(qkv && key_bits && out && dout && lse && delta && dqkv, GRIT_E_BADARG, "grit_attn_bidir_bwd: null pointer");
  GRIT_REQUIRE(B > 0 && S > 0 && nq > 0 && nkv > 0 && S <= (1 << 30) && nq <= 65535 && nkv <= 65535, GRIT_E_BADARG, "grit_attn_bidir_bwd: bad sizes");
  GRIT_REQUIRE(d == AB_D, GRIT_E_UNSUPPORTED, "grit_attn_bidir_bwd: head_dim=%d (only 128 is built)", d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "grit_attn_bidir_bwd: nq not a multiple of nkv");
  GRIT_REQUIRE(window >= 0 && (causal || window == 0), GRIT_E_BADARG, "grit_attn_bidir_bwd: window=%d (causal attention only)", window);
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "grit_attn_bidir_bwd: bad strides");
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), GRIT_E_BADARG,
               "grit_attn_bidir_bwd: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)B * nq * ((S + 127) / 128) < (1ll << 30), GRIT_E_UNSUPPORTED, "grit_attn_bidir_bwd: grid too large");
  GRIT_REQUIRE((int64_t)S * qkv_stride * 2 < (1ll << 31) && (int64_t)S * out_stride * 2 < (1ll << 31), GRIT_E_UNSUPPORTED,
               "grit_attn_bidir_bwd: one sequence spans more than 2 GiB (32-bit row offsets)");
  return attn_bwd_launch(false, causal, window, qkv, key_bits, nullptr, out, dout, lse, delta, dqkv, B, S, (int64_t)B * S, nq, nkv, qkv_stride,
                         out_stride, scale, (hipStream_t)stream);
}
extern "C" int grit_attn_bidir_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                                   float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                                   int64_t out_stride, float scale, void* stream) {
  return attn_bwd_padded(0, 0, qkv, key_bits, out, dout, lse, delta, dqkv, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_causal_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                                    float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                                    int64_t out_stride, float scale, void* stream) {
  return attn_bwd_padded(1, 0, qkv, key_bits, out, dout, lse, delta, dqkv, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}

static int attn_bwd_varlen(int causal, int window, const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                           float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                           int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  GRIT_REQUIRE(qkv && cu_seqlens && out && dout && lse && delta && dqkv, GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: null pointer");
  GRIT_REQUIRE(B > 0 && max_len > 0 && T > 0 && nq > 0 && nkv > 0 && max_len <= (1 << 30) && nq <= 65535 && nkv <= 65535, GRIT_E_BADARG,
               "grit_attn_bidir_varlen_bwd: bad sizes");
  GRIT_REQUIRE(d == AB_D, GRIT_E_UNSUPPORTED, "grit_attn_bidir_varlen_bwd: head_dim=%d (only 128 is built)", d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: nq not a multiple of nkv");
  GRIT_REQUIRE(window >= 0 && (causal || window == 0), GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: window=%d (causal attention only)", window);
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "grit_attn_bidir_varlen_bwd: bad strides");
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), GRIT_E_BADARG,
               "grit_attn_bidir_varlen_bwd: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)B * nq * ((max_len + 127) / 128) < (1ll << 30), GRIT_E_UNSUPPORTED, "grit_attn_bidir_varlen_bwd: grid too large");
  GRIT_REQUIRE((int64_t)max_len * qkv_stride * 2 < (1ll << 31) && (int64_t)max_len * out_stride * 2 < (1ll << 31), GRIT_E_UNSUPPORTED,
               "grit_attn_bidir_varlen_bwd: one sequence spans more than 2 GiB (32-bit row offsets)");
  return attn_bwd_launch(true, causal, window, qkv, nullptr, cu_seqlens, out, dout, lse, delta, dqkv, B, max_len, T, nq, nkv, qkv_stride, out_stride,
                         scale, (hipStream_t)stream);
}
extern "C" int grit_attn_bidir_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                                          float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                                          int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_bwd_varlen(0, 0, qkv, cu_seqlens, out, dout, lse, delta, dqkv, B, max_len, T, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_causal_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout, const float* lse,
                                           float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv, int d,
                                           int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_bwd_varlen(1, 0, qkv, cu_seqlens, out, dout, lse, delta, dqkv, B, max_len, T, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
// backward of grit_attn_causal_window_fwd / _varlen_fwd (same window: keys q - window + 1 .. q)
extern "C" int grit_attn_causal_window_bwd(const void* qkv, const uint64_t* key_bits, const void* out, const void* dout, const float* lse,
                                           float* delta, void* dqkv, int B, int S, int nq, int nkv, int d, int64_t qkv_stride,
                                           int64_t out_stride, float scale, int window, void* stream) {
  GRIT_REQUIRE(window >= 1, GRIT_E_BADARG, "grit_attn_causal_window_bwd: window=%d must be >= 1", window);
  return attn_bwd_padded(1, window, qkv, key_bits, out, dout, lse, delta, dqkv, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_causal_window_varlen_bwd(const void* qkv, const int32_t* cu_seqlens, const void* out, const void* dout,
                                                  const float* lse, float* delta, void* dqkv, int B, int max_len, int64_t T, int nq, int nkv,
                                                  int d, int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream) {
  GRIT_REQUIRE(window >= 1, GRIT_E_BADARG, "grit_attn_causal_window_varlen_bwd: window=%d must be >= 1", window);
  return attn_bwd_varlen(1, window, qkv, cu_seqlens, out, dout, lse, delta, dqkv, B, max_len, T, nq, nkv, d, qkv_stride, out_stride, scale,
                         stream);
}
