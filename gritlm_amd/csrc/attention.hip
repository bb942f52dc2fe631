// Bidirectional (non-causal) flash attention forward for gfx950, GQA, head_dim 128, key-padding bitmask.
//
// Replaces repeat_kv + the additive [B,1,S,S] mask + F.scaled_dot_product_attention of
// scripts/modeling_mistral_gritlm.py (:182-191, :1017-1036, :690-698).  No repeat_kv copy (the kv head is
// index arithmetic), no mask tensor (one uint64 per 64 keys), no S x S score matrix.
//
// Structure: one 256-thread workgroup walks up to `qpw` consecutive 128-row query blocks of one (batch, head); each wave owns 32 rows of
// a block; two workgroups per CU.  The K/V tile stream runs THROUGH the block seams (the first tile of the next block is staged, and
// its Q rows are fetched into the same registers, under the last tile of the current one), so the per-block prologue (Q + first-tile latency) is paid once
// per workgroup and the output stores of a block drain under the next block's first tile.  Workgroups that share K/V (the GQA
// group's heads x query-block groups of one (batch, kv head)) are dealt to the SAME XCD back to back (block v runs on XCD v % 8).
// KV tiles of 64 keys go HBM -> LDS by direct LDS-DMA (buffer_load ... lds, 16 B per lane, 1 KiB = 4 key rows per wave instruction;
// the descriptor's range check zero-fills rows past the sequence) into a two-stage ring: the eight pieces of tile t+1 are issued
// BETWEEN the QK products of tile t and land under its softmax and PV products -- no staging registers, no ds_write pass, one barrier
// per tile.  Every LDS fragment read of the tile loop is inline asm with counted lgkmcnt (round 3): K fragments two k-slices ahead of
// their products, V fragments in four groups of 8 with the first two requested in front of the softmax; nothing in the loop makes hipcc
// wait for more than it needs (its own lgkmcnt(0) / vmcnt(0) in front of builtin LDS reads, scalar loads and ds_bpermute cost 12 %).  Both images are row-major [key][256 B] with the 16-byte
// units XOR-swizzled through the per-lane SOURCE address (the LDS image of a DMA is lane-linear): K unit ^= key & 15
// (conflict-free ds_read_b128 of a 32-key fragment), V unit ^= 4 (key & 3) (conflict-free ds_read_b64_tr_b16: the 16 lanes of a
// transposing read touch 4 keys x 32 B, the XOR puts them -- and the second 16-lane group -- on 16 distinct units of one 256-B bank row).
//   S^T = K Q^T     v_mfma_f32_32x32x16_bf16(A = K rows, B = Q)  -> lane (q = lane&31) holds 32 keys' scores
//   O^T = V^T P^T   v_mfma_f32_32x32x16_bf16(A = V^T rows, B = P) -> lane (q = lane&31) holds 64 of its d's
// Both products are "swapped" so that every softmax statistic (max, sum, rescale) is lane-local: the only
// cross-lane traffic per tile is one v_permlane32_swap for the row max.  The P operand needs no
// permlane/LDS round trip: the MFMA contraction index is permuted identically on the V^T side
// (key(kb,c,hi,j) = 32kb + 16c + 8(j>>2) + 4hi + (j&3)): two transposing 8-byte LDS reads whose per-lane
// addresses select exactly those keys.
#include <stdlib.h>

#include <atomic>

#include "common.h"

// s_waitcnt vmcnt(0) through the builtin (gfx9 encoding: expcnt 7, lgkmcnt 15 = "don't wait"): unlike an asm statement the waitcnt
// insertion pass SEES it, so it does not add its own vmcnt(0) in front of the first MFMA that reads the re-fetched Q registers -- that
// wait would sit behind the freshly issued DMA of the next tile and serialise it
#define ATT_WAIT_VM0()                      \
  do {                                      \
    asm volatile("" ::: "memory");          \
    __builtin_amdgcn_s_waitcnt(0x0F70);     \
    asm volatile("" ::: "memory");          \
  } while (0)

#ifndef ATT_DEFER_MAX
#define ATT_DEFER_MAX 1
#endif
// ATT_TIMING (diagnostic build, tools/attn_phase_probe.sh): thread 0 of every workgroup writes s_memtime stamps (cycles since kernel entry) as
// raw u32 into the workgroup's LSE rows instead of the LSE -- where a workgroup's time goes between prologue, tiles, seams and stores
// (profiles/r05_attn_phase_probe.json).  Three scheduling experiments built on that picture were measured and did NOT ship
// (profiles/r05_attn_{prio,hpw}_ab.log; the priority variants are in git history, commit "ATT_PRIO_MODE"): a static
// s_setprio asymmetry between the two co-resident workgroups (+-1 %), 2 / 4 heads of a kv group per workgroup (-2 % / -12 %), and
// dropping the two lgkmcnt(0) of the output transposition (0 %).
#ifdef ATT_TIMING
#define ATT_STAMP() do { if (tid == 0 && att_ns < att_cap) { att_st[att_ns] = (uint32_t)(__builtin_readcyclecounter() - att_t0); ++att_ns; } } while (0)
#else
#define ATT_STAMP() do { } while (0)
#endif
// (ATT_DEFER_MAX and ATT_ABLATE_STORES are the A/B knobs of tools/ubench/attn_ab.cpp.  The levers of round 3 -- asm reads, spread /
//  buffer-addressed / unconditional DMA pieces, early V reads, s_setprio over QK, xor K addresses, hoisted mask word, permlane row
//  maximum -- were each A/B'd as a compile-time variant against the kernel of the previous commit, bit-identical every time
//  (profiles/r03_attn_fwd_ab_asm_reads.log, 773 -> 870 TF at B 256 x S 512); the variants are resolved in this file, the round-2
//  kernel is rebuilt from git history by tools/ubench/build_attn_ab.sh.)

namespace grit {

constexpr int ATT_D = 128;
constexpr int ATT_QB = 128;   // query rows per workgroup
constexpr int ATT_KB = 64;    // keys per tile
constexpr int V_PITCH = 256;  // bytes per key row of the row-major V image
constexpr int K_LDS_BYTES = ATT_KB * ATT_D * 2;  // 16384
constexpr int V_LDS_BYTES = ATT_KB * V_PITCH;    // 16384
constexpr int ATT_STAGE_BYTES = K_LDS_BYTES + V_LDS_BYTES;   // 32 KiB per stage, two stages
constexpr int ATT_XPOSE_BYTES = 4096;                         // per wave: 32 rows x 128 B, the Q-in / O-out transposition buffer
constexpr int ATT_LDS_BYTES = 2 * ATT_STAGE_BYTES + 4 * ATT_XPOSE_BYTES;   // 80 KiB: two workgroups fill the CU's 160 KiB exactly
typedef const __attribute__((address_space(1))) void* att_gptr_t;
typedef __attribute__((address_space(3))) void* att_lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ uint32_t lo16(uint32_t w) { return w & 0xffffu; }
__device__ __forceinline__ uint32_t hi16(uint32_t w) { return w >> 16; }

// VARLEN: sequences are packed back to back (no padding rows at all); cu_seqlens[b] is the first row of sequence b and
// every key of a sequence is valid, so the key bitmask is synthesised from the length.
// CAUSAL: additionally key <= query (the generative branch of unified training, MistralSdpaAttention with is_causal=True,
// modeling_mistral_gritlm.py:690-698 / :1017-1036); tiles past the workgroup's last query are skipped.
// CAUSAL with window > 0 (Mistral's sliding window, modeling_mistral_gritlm.py:381-385 / the sliding-window causal mask of
// _prepare_4d_causal_attention_mask, :1005-1036): a query sees the `window` keys q - window + 1 .. q; tiles that lie wholly in front
// of a query block's first visible key are skipped as well.
// F16 (round 5, the encoder's "f16_operands" precision policy): q | k | v and the output hold IEEE fp16, P is rounded to fp16 --
// v_mfma_f32_32x32x16_f16 runs at the bf16 rate, the LDS-DMA staging and the transposing reads are 16-bit-format-agnostic, so the
// instruction stream is the bf16 one with the two MFMA opcodes and the two pack instructions exchanged.  The deferred rescale keeps
// P <= 2^8, far inside the fp16 range; the output is a convex combination of V rows: nothing here can overflow.
__device__ __forceinline__ f32x16_t mma32(bf16x8_t a, bf16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_t mma32(f16x8_t a, f16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
template <bool F16> struct att_frag { typedef bf16x8_t type; };
template <> struct att_frag<true> { typedef f16x8_t type; };

template <bool VARLEN, bool CAUSAL, bool F16 = false>
__global__ void __launch_bounds__(256, 2)
attn_bidir_fwd_k(const uint16_t* __restrict__ qkv, const uint64_t* __restrict__ key_bits, const int32_t* __restrict__ cu_seqlens,
                 uint16_t* __restrict__ out, float* __restrict__ lse, int S_arg, int nq, int nkv, int64_t qkv_stride,
                 int64_t out_stride, float scale_log2, int qpw, int ngx, int n_sets, int window) {
  extern __shared__ __attribute__((aligned(256))) char smem[];          // 256: the asm read addresses OR / XOR lane constants into bits 7:4
  typedef typename att_frag<F16>::type frag_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware decode: the k-th workgroup of XCD x belongs to K/V set (k / U) * 8 + x, U = (heads per kv head) x (query-block groups)
  const int gqa = nq / nkv, U = gqa * ngx;
  const int kx = (int)blockIdx.x >> 3;
  const int set = (kx / U) * 8 + ((int)blockIdx.x & 7), member = kx % U;
  if (set >= n_sets) return;
  const int b = set / nkv, hk = set - b * nkv;
  const int h = hk * gqa + member % gqa;
  const int qb_first = (member / gqa) * qpw;
  int S = S_arg;
  int64_t row0 = (int64_t)b * S_arg;
  if constexpr (VARLEN) {
    row0 = cu_seqlens[b];
    S = cu_seqlens[b + 1] - cu_seqlens[b];
  }
  if (qb_first * ATT_QB >= S) return;            // uniform per workgroup
#ifdef ATT_TIMING
  const uint64_t att_t0 = __builtin_readcyclecounter();
  uint32_t* att_st = reinterpret_cast<uint32_t*>(lse + ((int64_t)b * nq + h) * S_arg + (int64_t)qb_first * ATT_QB);
  int att_ns = 0;
  const int att_cap = qpw * ATT_QB;
  ATT_STAMP();                                                                      // [0] = 0
  if (tid == 0) {
    att_st[att_ns++] = (uint32_t)__builtin_amdgcn_s_memrealtime();                   // [1] = 100 MHz wall clock at entry
    att_st[att_ns++] = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);          // [2] = HW_REG_HW_ID
    att_st[att_ns++] = (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);         // [3] = HW_REG_XCC_ID
  }
#endif
  const int nqb = (S + ATT_QB - 1) / ATT_QB;
  const int nblk = (nqb - qb_first) < qpw ? (nqb - qb_first) : qpw;

  // Global addressing = workgroup-uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: no 64-bit per-lane pointers to keep
  // alive (or spill) around the tile loop.  The launcher guarantees rows * stride * 2 < 2^31.
  const char* q_base = reinterpret_cast<const char*>(qkv + row0 * qkv_stride + (int64_t)h * ATT_D);
  const char* k_base = reinterpret_cast<const char*>(qkv + row0 * qkv_stride + (int64_t)(nq + hk) * ATT_D);
  const char* v_base = reinterpret_cast<const char*>(qkv + row0 * qkv_stride + (int64_t)(nq + nkv + hk) * ATT_D);
  char* o_base = reinterpret_cast<char*>(out + row0 * out_stride + (int64_t)h * ATT_D);
  const uint32_t qkv_stride_b = (uint32_t)qkv_stride * 2u, out_stride_b = (uint32_t)out_stride * 2u;

  const int ql = lane & 31, hi = lane >> 5;

  // ---- LDS-DMA roles: wave w stages keys 16w .. 16w+15 of a tile, four 1-KiB instructions for K and four for V (4 keys each);
  //      lane -> key 16w + 4i + (lane>>4), physical 16-byte unit lane&15, which holds the LOGICAL unit (lane&15) ^ swizzle(key)
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int st_key = 16 * wv + (lane >> 4);                                     // + 4i
  const uint32_t v_unit_b = (uint32_t)(((lane & 15) ^ (4 * ((lane >> 4) & 3))) << 4);   // V: unit ^= 4 (key & 3); key & 3 == (lane>>4) & 3
  auto stage_tile = [&](int t, int buf) {
    char* kdst = smem + buf * ATT_STAGE_BYTES + wv * 4096;
    char* vdst = kdst + K_LDS_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int key = t * ATT_KB + st_key + 4 * i;
      key = key < S ? key : S - 1;
      const uint32_t k_unit_b = (uint32_t)(((lane & 15) ^ ((4 * i + (lane >> 4)) & 15)) << 4);   // K: unit ^= key & 15
      const uint32_t row_b = (uint32_t)key * qkv_stride_b;
      __builtin_amdgcn_global_load_lds((att_gptr_t)(k_base + (row_b + k_unit_b)), (att_lptr_t)(kdst + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((att_gptr_t)(v_base + (row_b + v_unit_b)), (att_lptr_t)(vdst + i * 1024), 16, 0, 0);
    }
  };
  // one 1-KiB piece of a tile (the eight pieces of the next tile are issued BETWEEN the QK products of the current one --
  // an LDS-DMA instruction costs 60-185 issue cycles (guide, 'LDS-DMA piece issue cost'), eight of them in front of the first K read
  // held the whole tile back; between MFMAs the cost sits under the matrix pipe)
  // Buffer-addressed LDS-DMA (buffer_load_dwordx4 ... offen lds): the descriptor's range check zero-fills rows past the sequence (their
  // keys are masked anyway: finite K -> score -> -inf, P = 0 x finite V), so a piece needs no per-lane clamp / 32-bit multiply: the
  // lane's offset inside a tile (row + swizzled unit; V: + the distance of the V heads from the K heads) is a constant VGPR per piece,
  // the tile's row offset ONE scalar operand for all eight pieces, and ONE descriptor serves K and V (a V row of the last valid key
  // ends exactly at num_records; the next row of either operand starts (nq + nkv) x 256 - 256 >= 0 bytes behind it).
  const uint32_t v_delta_b = (uint32_t)nkv * ATT_D * 2u;
  const auto kv_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(k_base), (short)0, (int)((uint32_t)(S - 1) * qkv_stride_b + 256u + v_delta_b), 0x00020000);
  uint32_t pc_k[4], pc_v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pc_k[i] = (uint32_t)(st_key + 4 * i) * qkv_stride_b + (uint32_t)(((lane & 15) ^ ((4 * i + (lane >> 4)) & 15)) << 4);
    pc_v[i] = (uint32_t)(st_key + 4 * i) * qkv_stride_b + v_unit_b + v_delta_b;
  }
  auto stage_piece = [&](int t, int buf, int i, int is_v) {
    char* dst = smem + buf * ATT_STAGE_BYTES + wv * 4096 + (is_v ? K_LDS_BYTES : 0) + i * 1024;
    const uint32_t tile_b = (uint32_t)__builtin_amdgcn_readfirstlane(t * ATT_KB) * qkv_stride_b;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(kv_rsrc, (att_lptr_t)dst, 16, (int)(is_v ? pc_v[i] : pc_k[i]), (int)tile_b, 0, 0);
  };
  // the first tile's DMA goes out before anything else (tile 0 always exists: S > 0); with a window the first tile of the first block
  // is known only after the key bitmask has been scanned (below)
  if (!(CAUSAL && window > 0)) stage_tile(0, 0);

  // ---- Q fragments (B operand): lane holds Q[q][16ks + 8hi .. +8] of query block qb.  A row-per-lane global load touches 32 rows x 32 B
  //      per instruction (measured: the per-block Q fetch + O store in that shape cost 18 % of the kernel at S = 512), so Q comes
  //      in through the wave's private 4 KiB transposition buffer instead, one 64-column half (32 rows x 128 B) at a time: LDS-DMA of
  //      whole 128-byte row segments (4 instructions x 8 rows, 16-byte units swizzled by (row>>1)&7), then ds_read_b128 of the half's
  //      four k-slices.
  frag_t qf[8];
  char* xs = smem + 2 * ATT_STAGE_BYTES + wv * ATT_XPOSE_BYTES;
  const int q_swz = (ql >> 1) & 7;                                               // swizzle of the lane's own row
  // (per-block address arithmetic is recomputed from a laundered lane id where it is used: hoisted to kernel entry it would be spilled
  //  around the tile loop, and a scratch reload's vmcnt wait would sit in front of the hand-placed DMA)
  auto q_stage_half = [&](int qb, int half) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int x_row = ln >> 3, x_unit = ln & 7;                                  // row (+ 8j) and physical unit of a DMA piece
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 8 * j + x_row;
      int qr = qb * ATT_QB + wave * 32 + r;
      qr = qr < S ? qr : S - 1;
      const uint32_t unit_b = (uint32_t)((half * 8 + (x_unit ^ ((r >> 1) & 7))) << 4);   // logical unit held by physical unit x_unit of row r
      __builtin_amdgcn_global_load_lds((att_gptr_t)(q_base + ((uint32_t)qr * qkv_stride_b + unit_b)), (att_lptr_t)(xs + j * 1024), 16, 0, 0);
    }
  };
  auto q_read_half = [&](int half) {
    const char* rp = xs + ql * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[half * 4 + ks] = *reinterpret_cast<const frag_t*>(rp + (((2 * ks + hi) ^ q_swz) << 4));
    // the reads must have left the buffer before it is refilled (DMA) or rewritten (O staging)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);                                           // lgkmcnt(0)
    asm volatile("" ::: "memory");
  };
  // the workgroup's FIRST block takes its Q rows straight from global memory (row-per-lane loads, one round trip that overlaps the
  // first tile's DMA; two dependent passes through the 4 KiB buffer would put two memory latencies in front of the first product)
  {
    const int qr0 = qb_first * ATT_QB + wave * 32 + ql;
    const char* qp = q_base + ((uint32_t)(qr0 < S ? qr0 : S - 1) * qkv_stride_b + (uint32_t)hi * 16u);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qf[ks] = *reinterpret_cast<const frag_t*>(qp + ks * 32);
    }
  }

  // number of KV tiles that contain at least one valid key (trailing padding is never loaded)
  int ntiles_all = 0;
  const uint64_t* bits = nullptr;
  if constexpr (VARLEN) {
    ntiles_all = (S + 63) >> 6;
  } else {
    const int W = (S + 63) >> 6;
    bits = key_bits + (int64_t)b * W;
    for (int w = W - 1; w >= 0; --w)
      if (bits[w] != 0) { ntiles_all = w + 1; break; }
  }

  // K fragment address: row = 32kb + (lane&31), d-slot = 2ks + hi, swizzle by row&15 == lane&15
  const int kf_row = ql * 256, kf_x = lane & 15;
  // V fragment (A operand of O^T += V^T P^T) straight from the ROW-MAJOR V image with ds_read_b64_tr_b16: in every 16-lane group lane j
  // points at V[k0 + j/4][d0 + 4 (j%4)] and lane c receives V[k0 .. k0+3][d0 + c] (the hardware transposes the group's 4 x 16 block);
  // d0 = 32db + 16 ((lane>>4)&1) makes c <-> the MFMA row lane&31, k0 = 32kb + 16c + 4hi (+8 for the second half of the k-slice)
  // (key & 3) of every key a lane addresses is (lane>>2)&3, so the V swizzle turns the d-block offset db*64 into (db ^ r)*64, r = (lane>>2)&3:
  // vt_lane carries r in byte bits 7:6 and the read address of d-block db is vt_lane ^ (db << 6)
  const int vt_lane = (((lane & 15) >> 2) + 4 * hi) * V_PITCH + ((((lane >> 2) & 3) * 4) << 4) + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;

  // KV tiles [t0, t1) of query block qb
  auto tile_range = [&](int qb, int& t0, int& t1) {
    t0 = 0;
    t1 = ntiles_all;
    if constexpr (CAUSAL) {
      const int lim = 2 * qb + 2;               // tiles holding keys <= the last query of this block
      t1 = t1 < lim ? t1 : lim;
      if (window > 0) {
        const int lo = qb * ATT_QB - window + 1;                // first key the block's FIRST query sees
        t0 = lo > 0 ? lo >> 6 : 0;
        t0 = t0 < t1 ? t0 : (t1 > 0 ? t1 - 1 : 0);             // never an empty range while the sequence has keys (the K/V stream
      }                                                        // through the block seams counts on one tile per block); its keys are
    }                                                          // then masked for every row
  };
  if (CAUSAL && window > 0) {
    int f0, f1;
    tile_range(qb_first, f0, f1);
    stage_tile(f0, 0);
  }

  int gt = 0;                                   // tiles consumed so far by this workgroup: tile g lives in stage g & 1
  ATT_WAIT_VM0();                               // first tile + first Q rows
  ATT_STAMP();                                  // [4] prologue done
  for (int qi = 0; qi < nblk; ++qi) {
    const int qb = qb_first + qi;
    const int q_row = qb * ATT_QB + wave * 32 + ql;
    int t_first, ntiles, next_first = 0, next_end;
    tile_range(qb, t_first, ntiles);
    const bool more = qi + 1 < nblk;
    if (more) tile_range(qb + 1, next_first, next_end);

    f32x16_t oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int t = t_first; t < ntiles; ++t, ++gt) {
      // tile t has landed (this wave's share: vmcnt; everybody's: the barrier) and every wave is done reading the other stage.  The
      // first tile of a block was waited for before the block loop / at the seam (before the previous block stored its output:
      // vmcnt counts stores, and the stores should drain under this tile, not in front of it).
      if (t > t_first) ATT_WAIT_VM0();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ATT_STAMP();                              // per tile: past the barrier
      // the eight pieces are issued unconditionally: behind the workgroup's very last tile they re-stage that tile into the idle stage
      // (nobody reads it; the wait in front of the block's output stores covers it) instead of costing a uniform branch per piece
      const int st_t = (t + 1 < ntiles) ? t + 1 : (more ? next_first : t), st_buf = (gt + 1) & 1;
      // next block's Q, first 64 columns: fetched a whole tile ahead (the buffer is idle), so the wait at the top of the last tile
      // already covers it
      if (more && t + 2 == ntiles) q_stage_half(qb + 1, 0);
      const char* k_lds = smem + (gt & 1) * ATT_STAGE_BYTES;
      const char* v_lds = k_lds + K_LDS_BYTES;
      // the tile's key-mask word is fetched HERE (a scalar load; its latency sits under the QK products) and turned into `fast` before
      // the asm LDS reads of the PV products are requested: fetched where it is used, hipcc's lgkmcnt(0) for the scalar load waited for
      // the load's own round trip AND for the sixteen V reads just issued -- once per tile
      uint64_t word;
      if constexpr (VARLEN) {
        const int rem = S - t * ATT_KB;
        word = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
      } else {
        word = bits[t];
      }
      bool fast = (word == ~0ull);
      if constexpr (CAUSAL) {
        const int qw0 = qb * ATT_QB + wave * 32;                   // this wave's first query
        // tile reaches past the wave's first query, or starts in front of the first key the wave's LAST query sees: per-lane bounds
        if (t * ATT_KB + ATT_KB - 1 > qw0 || (window > 0 && t * ATT_KB < qw0 + 32 - window)) {
          const int n = q_row - t * ATT_KB + 1;                    // keys of this tile the lane's query may see
          word &= n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
          if (window > 0) {
            const int lo = n - window;                             // keys of this tile in front of the lane's window
            word &= lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
          }
          fast = false;
        }
      }

      // ---- S^T = K Q^T (scores for 64 keys x 32 q per wave)
      // the two 32-key halves are two INDEPENDENT accumulation chains, issued alternately: a v_mfma_f32_32x32x16_bf16 occupies the pipe
      // for 32 cycles but its result is ready after 64, so a product that accumulates onto the one issued just before it has to
      // wait (hipcc keeps MFMA source order; A/B against "8 products on one half, then 8 on the other": +2.5-4.5 %, bit-identical)
      f32x16_t sacc[2];
      // K fragments as inline-asm ds_read_b128 with counted lgkmcnt, requested two k-slices (2 reads each) ahead of the products that
      // consume them (hipcc issues them in small batches and waits lgkmcnt(0) seven times per tile)
      {
        const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // (2 ks + hi) ^ kf_x == (2 ks) ^ (hi ^ kf_x): one per-tile base with the lane's constant in address bits 7:4, one v_xor per k-slice
        const uint32_t kbase = ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)k_lds) + (uint32_t)kf_row) | (uint32_t)((hi ^ kf_x) << 4);
        frag_t kr[8][2];
#define ATT_K_READ(KS)                                                                                                              \
  do {                                                                                                                              \
    const uint32_t ka = kbase ^ (uint32_t)((KS) << 5);                                                                              \
    asm volatile("ds_read_b128 %0, %1" : "=v"(kr[KS][0]) : "v"(ka));                                                                \
    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(kr[KS][1]) : "v"(ka));                                                    \
  } while (0)
#define ATT_K_MMA(KS, N)                                                                                                            \
  do {                                                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(kr[KS][0]), "+v"(kr[KS][1]) : : "memory");                                      \
    sacc[0] = mma32(kr[KS][0], qf[KS], (KS) == 0 ? zero16 : sacc[0]);                                                               \
    sacc[1] = mma32(kr[KS][1], qf[KS], (KS) == 0 ? zero16 : sacc[1]);                                                               \
    stage_piece(st_t, st_buf, (KS) >> 1, (KS) & 1);                                                                                 \
    asm volatile("" ::: "memory");                                                                                                  \
  } while (0)
        // K fragments two k-slices ahead of their products (three ahead: no gain); one LDS-DMA piece of the next tile behind every
        // product pair; s_setprio 1 over the section: 0 .. +2 % (over the PV section as well: -1.5 %)
        __builtin_amdgcn_s_setprio(1);
        ATT_K_READ(0); ATT_K_READ(1);
        ATT_K_READ(2); ATT_K_MMA(0, 4);
        ATT_K_READ(3); ATT_K_MMA(1, 4);
        ATT_K_READ(4); ATT_K_MMA(2, 4);
        ATT_K_READ(5); ATT_K_MMA(3, 4);
        ATT_K_READ(6); ATT_K_MMA(4, 4);
        ATT_K_READ(7); ATT_K_MMA(5, 4);
        ATT_K_MMA(6, 2); ATT_K_MMA(7, 0);
        __builtin_amdgcn_s_setprio(0);
#undef ATT_K_READ
#undef ATT_K_MMA
      }

      // the next block's Q rows replace this block's as soon as its last QK product has read them: the fetch lands under the
      // softmax and PV of the last tile, no second register set
      const bool q_next = more && t + 1 == ntiles;
      if (q_next) {                 // this block's last QK products have read qf: refill it for the next block
        if (ntiles - t_first == 1) { q_stage_half(qb + 1, 0); ATT_WAIT_VM0(); }
        q_read_half(0);
        q_stage_half(qb + 1, 1);    // the other 64 columns land under the softmax and the PV products
      }

      // The transposing V reads as inline asm with COUNTED lgkmcnt: hipcc's wait-count pass treats a ds_read_b64_tr_b16 builtin as aliasing the
      // pending LDS-DMA and puts an s_waitcnt vmcnt(0) in front of the first one -- the NEXT tile's DMA then had to land before this
      // tile's PV products could start.  Four groups (kb, c) of 8 reads feed 4 MFMAs each; group g + 1 is requested before group g is
      // waited for (lgkmcnt(8): LDS returns in order), two register sets of 16.  The first two groups are requested HERE,
      // in front of the softmax (the V tile has been in LDS since the barrier at the top of the tile; the K fragments' registers are free).
      const uint32_t vbase = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)v_lds);
      uint32_t va[4];
#pragma unroll
      for (int db = 0; db < 4; ++db) va[db] = vbase + (uint32_t)(vt_lane ^ (db << 6));
      s16x4_t vr[2][4][2];
#define ATT_TR_GROUP(G, BUF)                                                                                                          \
  _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                                                  \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[BUF][db][0]) : "v"(va[db]), "i"(((G) >> 1) * 32 * V_PITCH + ((G) & 1) * 16 * V_PITCH));                \
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vr[BUF][db][1]) : "v"(va[db]), "i"(((G) >> 1) * 32 * V_PITCH + ((G) & 1) * 16 * V_PITCH + 8 * V_PITCH)); \
  }
      {
        int fast_i = __builtin_amdgcn_readfirstlane(fast ? 1 : 0);   // wave-uniform; made opaque HERE so that the compare -- and with it hipcc's wait for the
        asm volatile("" : "+s"(fast_i));            // mask word's scalar load -- sits in front of the V reads, not behind them
        fast = fast_i != 0;
      }
      ATT_TR_GROUP(0, 0)
      ATT_TR_GROUP(1, 1)

      // ---- mask + online softmax (all lane-local except one exchange with lane^32)
      float mx = -INFINITY;
      if (fast) {                   // every key of the tile is valid (all tiles but a ragged last one): no per-element mask
        // four independent chains (max is exact: any order gives the same bits); one chain of 16 dependent v_max3 is a latency chain
        float m4[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          const int kb = c4 >> 1, r0 = (c4 & 1) * 8;
          m4[c4] = fmaxf(fmaxf(sacc[kb][r0], sacc[kb][r0 + 1]), sacc[kb][r0 + 2]);
#pragma unroll
          for (int r = 3; r < 8; ++r) m4[c4] = fmaxf(m4[c4], sacc[kb][r0 + r]);
        }
        mx = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
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
      {
        // max over the two 32-lane halves through v_permlane32_swap (a VALU instruction): __shfl_xor is a ds_bpermute, an LDS-queue
        // instruction whose lgkmcnt(0) also waits for the V reads requested in front of the softmax
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * scale_log2;      // scale > 0: max commutes with the scaling
      }
      const float m_new = fmaxf(m_run, mx);
#if ATT_DEFER_MAX
      // deferred rescale: a row keeps its old reference maximum as long as its maximum grows by less than 2^8 (P <= 256, exact in
      // the bf16 exponent range; l and O stay consistent with m_run); when NO row of the wave has to move, the 64 accumulator
      // multiplies + exp2 of the rescale are skipped.  The decision is per row (rows that stay multiply by exactly 1), so a row's
      // result never depends on which other rows share its wave -- the packed and the padded layout stay bit-identical.
      const bool grow = !(m_new - m_run <= 8.0f);                 // also true for m_run = -inf (first tile, or all keys masked so far)
      if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
        const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_ref) : 1.0f;  // m_run = -inf -> 0
        m_run = grow ? m_new : m_run;
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      }
      const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
#else
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
      m_run = m_new;
#endif
      float psum = 0.f;
      frag_t pb[2][2];
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
            pk[jj] = pack2_op<F16>(p0, p1);
          }
          pb[kb][c] = __builtin_bit_cast(frag_t, make_uint4(pk[0], pk[1], pk[2], pk[3]));
        }
#if ATT_DEFER_MAX
      l_run += psum;
#else
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#endif


      // ---- O^T += V^T P^T   (the four d-blocks are four independent accumulators: round-robin, never the same one twice in a row)
      {
#define ATT_TR_WAIT(N, BUF)                                                                                                           \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                                            \
               : "+v"(vr[BUF][0][0]), "+v"(vr[BUF][0][1]), "+v"(vr[BUF][1][0]), "+v"(vr[BUF][1][1]), "+v"(vr[BUF][2][0]),              \
                 "+v"(vr[BUF][2][1]), "+v"(vr[BUF][3][0]), "+v"(vr[BUF][3][1]))
#define ATT_TR_MMA(G, BUF)                                                                                                            \
  _Pragma("unroll") for (int db = 0; db < 4; ++db) {                                                                                  \
    const s16x4_t v0 = vr[BUF][db][0], v1 = vr[BUF][db][1];                                                                           \
    const frag_t vf = __builtin_bit_cast(frag_t, (__attribute__((ext_vector_type(8))) short){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}); \
    oacc[db] = mma32(vf, pb[(G) >> 1][(G) & 1], oacc[db]);                                                                            \
  }
        ATT_TR_WAIT(8, 0);
        ATT_TR_MMA(0, 0)
        ATT_TR_GROUP(2, 0)
        ATT_TR_WAIT(8, 1);
        ATT_TR_MMA(1, 1)
        ATT_TR_GROUP(3, 1)
        ATT_TR_WAIT(8, 0);
        ATT_TR_MMA(2, 0)
        ATT_TR_WAIT(0, 1);
        ATT_TR_MMA(3, 1)
#undef ATT_TR_GROUP
#undef ATT_TR_WAIT
#undef ATT_TR_MMA
      }
    }

    // ---- block seam: the next block's first tile and Q rows (issued one tile ago) are waited for BEFORE this block's stores go out
    ATT_STAMP();                                // block: tile loop done
    ATT_WAIT_VM0();
    if (more) q_read_half(1);
    ATT_STAMP();                                // block: seam wait done

    // ---- epilogue: lane holds O[q][32db + 8g + 4hi + 0..3] in regs 4g..4g+3 of oacc[db]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);     // (once per block; through v_permlane32_swap like the tile maximum: measured 2 % slower at S >= 2048)
    const float inv_l = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    // full-line stores through the transposition buffer (a row-per-lane store touches 32 lines x 32 B per instruction): one 64-column
    // half at a time, every lane writes its 4 pieces of the half (unit ^= (row>>1)&7), then stores 16 B of an 8-row x 128-B piece
    const int q_wave0 = qb * ATT_QB + wave * 32;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int x_row = ln >> 3, x_unit = ln & 7;                                  // row (+ 8j) and physical unit of a store piece
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      char* wp = xs + (ln & 31) * 128;
#pragma unroll
      for (int dbl = 0; dbl < 2; ++dbl)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          // v_permlane32_swap exchanges the two 32-lane halves of the register groups g and g+1, so that a lane ends up with 8
          // CONSECUTIVE dims of its row (half 0: group g, half 1: group g+1): one 16-byte piece
          const int db = half * 2 + dbl;
          float a[4], bq[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(oacc[db][8 * gp + e] * inv_l),
                                                             __float_as_uint(oacc[db][8 * gp + 4 + e] * inv_l), false, false);
            a[e] = __uint_as_float(sw[0]); bq[e] = __uint_as_float(sw[1]);
          }
          *reinterpret_cast<uint4*>(wp + (((dbl * 4 + gp * 2 + hi) ^ q_swz) << 4)) =
              make_uint4(pack2_op<F16>(a[0], a[1]), pack2_op<F16>(a[2], a[3]), pack2_op<F16>(bq[0], bq[1]), pack2_op<F16>(bq[2], bq[3]));
        }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);                                         // lgkmcnt(0): the wave's own writes are in the buffer
      asm volatile("" ::: "memory");
      uint4 piece[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) piece[j] = *reinterpret_cast<const uint4*>(xs + j * 1024 + ln * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + x_row;
        const int qr = q_wave0 + r;
#ifdef ATT_ABLATE_STORES
        if (qr < S && scale_log2 < -1e30f)
#else
        if (qr < S)
#endif
        {
          typedef __attribute__((ext_vector_type(4))) unsigned int att_u32x4_t;
          att_u32x4_t* op = reinterpret_cast<att_u32x4_t*>(o_base + ((uint32_t)qr * out_stride_b + (uint32_t)((half * 8 + (x_unit ^ ((r >> 1) & 7))) << 4)));
          const att_u32x4_t pv = {piece[j].x, piece[j].y, piece[j].z, piece[j].w};
          *op = pv;
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);                                         // the pieces are in registers before the buffer is reused
      asm volatile("" ::: "memory");
    }
    ATT_STAMP();                                // block: output stores issued
#ifdef ATT_TIMING
    if (qi + 1 == nblk && tid == 0 && att_ns + 1 < att_cap) { att_st[att_ns++] = (uint32_t)__builtin_amdgcn_s_memrealtime(); att_st[att_ns++] = 0xffffffffu; }
#else
    if (q_row < S && lse != nullptr && hi == 0) {
      if constexpr (VARLEN) lse[(row0 + q_row) * nq + h] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;  // [T, nq]
      else lse[((int64_t)b * nq + h) * S + q_row] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
    }
#endif
  }
}


}  // namespace grit

using namespace grit;

// Launch geometry: query blocks per workgroup (the K/V stream runs through the block seams, so more blocks per workgroup amortise the
// prologue) -- as many as 4 while the launch still has >= 4 workgroups per CU-slot pair -- and the XCD-aware 1-D grid.
// the 80 KiB dynamic-LDS opt-in is a per-device function attribute: set it once per (instantiation, device)
template <typename KernelT>
static void attn_lds_optin(KernelT kernel, std::atomic<uint64_t>& done, int bytes = ATT_LDS_BYTES) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
}
template <bool VARLEN, bool CAUSAL, bool F16 = false>
static void attn_launch(dim3 grid, hipStream_t st, const uint16_t* qkv, const uint64_t* key_bits, const int32_t* cu, uint16_t* out, float* lse,
                        int S, int nq, int nkv, int64_t qkv_stride, int64_t out_stride, float scale_log2, int qpw, int ngx, int n_sets,
                        int window) {
  static std::atomic<uint64_t> optin{0};
  attn_lds_optin(attn_bidir_fwd_k<VARLEN, CAUSAL, F16>, optin);
  hipLaunchKernelGGL((attn_bidir_fwd_k<VARLEN, CAUSAL, F16>), grid, dim3(256), ATT_LDS_BYTES, st, qkv, key_bits, cu, out, lse, S, nq, nkv, qkv_stride,
                     out_stride, scale_log2, qpw, ngx, n_sets, window);
}

struct AttnGeom {
  int qpw, ngx, n_sets;
  unsigned grid;
};
static AttnGeom attn_geom(int B, int max_len, int nq, int nkv, bool causal) {
  const int nqb = (max_len + ATT_QB - 1) / ATT_QB;
  static const int forced = getenv("GRIT_ATTN_QPW") ? atoi(getenv("GRIT_ATTN_QPW")) : 0;      // A/B knob
  int qpw = 1;
  // causal: consecutive query blocks see 2, 4, 6, ... KV tiles, so a workgroup walking 4 of them is up to 4x longer than its neighbour;
  // 2 blocks per workgroup balance better (B 64 x S 2048: 869 -> 893 TF, profiles/r03_attn_fwd_ab_asm_reads.log)
  for (int c = causal ? 2 : 4; c >= 1; c >>= 1)
    if ((int64_t)B * nq * ((nqb + c - 1) / c) >= 2048 || c == 1) { qpw = c; break; }
  if (forced > 0) qpw = forced;
  if (qpw > nqb) qpw = nqb;
  AttnGeom g;
  g.qpw = qpw;
  g.ngx = (nqb + qpw - 1) / qpw;
  g.n_sets = B * nkv;
  g.grid = (unsigned)(8 * ((g.n_sets + 7) / 8) * (nq / nkv) * g.ngx);
  return g;
}

static int attn_fwd_padded(const char* name, bool causal, int window, const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S,
                           int nq, int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream, bool f16 = false) {
  GRIT_REQUIRE(qkv && key_bits && out, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && S > 0 && nq > 0 && nkv > 0 && S <= (1 << 30) && nq <= 65535 && nkv <= 65535, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(d == ATT_D, GRIT_E_UNSUPPORTED, "%s: head_dim=%d (only 128 is built)", name, d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "%s: nq=%d not a multiple of nkv=%d", name, nq, nkv);
  GRIT_REQUIRE(window >= 0 && (causal || window == 0), GRIT_E_BADARG, "%s: window=%d (>= 1 keys per query, causal attention only)", name, window);
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "%s: bad strides", name);
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  GRIT_REQUIRE((int64_t)B * nq * ((S + ATT_QB - 1) / ATT_QB) < (1ll << 30), GRIT_E_UNSUPPORTED, "%s: grid too large", name);
  GRIT_REQUIRE((int64_t)S * qkv_stride * 2 < (1ll << 31) && (int64_t)S * out_stride * 2 < (1ll << 31), GRIT_E_UNSUPPORTED,
               "%s: one sequence spans more than 2 GiB (32-bit row offsets)", name);
  const AttnGeom g = attn_geom(B, S, nq, nkv, causal);
  const dim3 grid(g.grid);
  if (causal && f16)
    attn_launch<false, true, true>(grid, (hipStream_t)stream, (const uint16_t*)qkv, key_bits, nullptr, (uint16_t*)out, lse, S, nq, nkv, qkv_stride,
                                   out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, window);
  else if (causal)
    attn_launch<false, true>(grid, (hipStream_t)stream, (const uint16_t*)qkv, key_bits, nullptr, (uint16_t*)out, lse, S, nq, nkv, qkv_stride,
                             out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, causal ? window : 0);
  else if (f16)
    attn_launch<false, false, true>(grid, (hipStream_t)stream, (const uint16_t*)qkv, key_bits, nullptr, (uint16_t*)out, lse, S, nq, nkv, qkv_stride,
                                    out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, 0);
  else
    attn_launch<false, false>(grid, (hipStream_t)stream, (const uint16_t*)qkv, key_bits, nullptr, (uint16_t*)out, lse, S, nq, nkv, qkv_stride,
                              out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, causal ? window : 0);
  GRIT_CHECK_LAUNCH(name);
  return GRIT_OK;
}

static int attn_fwd_varlen(const char* name, bool causal, int window, const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len,
                           int nq, int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream, bool f16 = false) {
  GRIT_REQUIRE(qkv && cu_seqlens && out, GRIT_E_BADARG, "%s: null pointer", name);
  GRIT_REQUIRE(B > 0 && max_len > 0 && nq > 0 && nkv > 0 && max_len <= (1 << 30) && nq <= 65535 && nkv <= 65535, GRIT_E_BADARG, "%s: bad sizes", name);
  GRIT_REQUIRE(d == ATT_D, GRIT_E_UNSUPPORTED, "%s: head_dim=%d (only 128 is built)", name, d);
  GRIT_REQUIRE(nq % nkv == 0, GRIT_E_BADARG, "%s: nq=%d not a multiple of nkv=%d", name, nq, nkv);
  GRIT_REQUIRE(window >= 0 && (causal || window == 0), GRIT_E_BADARG, "%s: window=%d (>= 1 keys per query, causal attention only)", name, window);
  GRIT_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (int64_t)(nq + 2 * nkv) * d && out_stride % 8 == 0 && out_stride >= (int64_t)nq * d,
               GRIT_E_BADARG, "%s: bad strides", name);
  GRIT_REQUIRE(aligned16(qkv) && aligned16(out), GRIT_E_BADARG, "%s: pointers must be 16-byte aligned", name);
  GRIT_REQUIRE((int64_t)B * nq * ((max_len + ATT_QB - 1) / ATT_QB) < (1ll << 30), GRIT_E_UNSUPPORTED, "%s: grid too large", name);
  GRIT_REQUIRE((int64_t)max_len * qkv_stride * 2 < (1ll << 31) && (int64_t)max_len * out_stride * 2 < (1ll << 31), GRIT_E_UNSUPPORTED,
               "%s: one sequence spans more than 2 GiB (32-bit row offsets)", name);
  const AttnGeom g = attn_geom(B, max_len, nq, nkv, causal);
  const dim3 grid(g.grid);
  if (causal && f16)
    attn_launch<true, true, true>(grid, (hipStream_t)stream, (const uint16_t*)qkv, nullptr, cu_seqlens, (uint16_t*)out, lse, max_len, nq, nkv, qkv_stride,
                                  out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, window);
  else if (causal)
    attn_launch<true, true>(grid, (hipStream_t)stream, (const uint16_t*)qkv, nullptr, cu_seqlens, (uint16_t*)out, lse, max_len, nq, nkv, qkv_stride,
                            out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, causal ? window : 0);
  else if (f16)
    attn_launch<true, false, true>(grid, (hipStream_t)stream, (const uint16_t*)qkv, nullptr, cu_seqlens, (uint16_t*)out, lse, max_len, nq, nkv, qkv_stride,
                                   out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, 0);
  else
    attn_launch<true, false>(grid, (hipStream_t)stream, (const uint16_t*)qkv, nullptr, cu_seqlens, (uint16_t*)out, lse, max_len, nq, nkv, qkv_stride,
                             out_stride, scale * 1.4426950408889634f, g.qpw, g.ngx, g.n_sets, causal ? window : 0);
  GRIT_CHECK_LAUNCH(name);
  return GRIT_OK;
}

extern "C" int grit_attn_bidir_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv,
                                   int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_padded("grit_attn_bidir_fwd", false, 0, qkv, key_bits, out, lse, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
// fp16-operand policy: qkv and out hold IEEE fp16.  Bidirectional (the embedding path) and -- ABI 5 -- causal, optionally with a sliding
// window of `window` keys (0: none): the causal prompt pass of a unified / generative model and the 'cc' embedding attention
extern "C" int grit_attn_causal_f16_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv,
                                        int d, int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream) {
  return attn_fwd_padded("grit_attn_causal_f16_fwd", true, window, qkv, key_bits, out, lse, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream, true);
}
extern "C" int grit_attn_causal_varlen_f16_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                               int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream) {
  return attn_fwd_varlen("grit_attn_causal_varlen_f16_fwd", true, window, qkv, cu_seqlens, out, lse, B, max_len, nq, nkv, d, qkv_stride, out_stride,
                         scale, stream, true);
}
extern "C" int grit_attn_bidir_f16_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv,
                                       int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_padded("grit_attn_bidir_f16_fwd", false, 0, qkv, key_bits, out, lse, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream, true);
}
extern "C" int grit_attn_bidir_varlen_f16_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                              int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_varlen("grit_attn_bidir_varlen_f16_fwd", false, 0, qkv, cu_seqlens, out, lse, B, max_len, nq, nkv, d, qkv_stride, out_stride, scale,
                         stream, true);
}
extern "C" int grit_attn_causal_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv,
                                    int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_padded("grit_attn_causal_fwd", true, 0, qkv, key_bits, out, lse, B, S, nq, nkv, d, qkv_stride, out_stride, scale, stream);
}
extern "C" int grit_attn_bidir_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                          int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_varlen("grit_attn_bidir_varlen_fwd", false, 0, qkv, cu_seqlens, out, lse, B, max_len, nq, nkv, d, qkv_stride, out_stride, scale,
                         stream);
}
extern "C" int grit_attn_causal_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                           int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, void* stream) {
  return attn_fwd_varlen("grit_attn_causal_varlen_fwd", true, 0, qkv, cu_seqlens, out, lse, B, max_len, nq, nkv, d, qkv_stride, out_stride, scale,
                         stream);
}

// Sliding-window causal attention: query q sees keys q - window + 1 .. q (window >= 1; window >= S is plain causal attention).
extern "C" int grit_attn_causal_window_fwd(const void* qkv, const uint64_t* key_bits, void* out, float* lse, int B, int S, int nq, int nkv,
                                           int d, int64_t qkv_stride, int64_t out_stride, float scale, int window, void* stream) {
  GRIT_REQUIRE(window >= 1, GRIT_E_BADARG, "grit_attn_causal_window_fwd: window=%d must be >= 1", window);
  return attn_fwd_padded("grit_attn_causal_window_fwd", true, window, qkv, key_bits, out, lse, B, S, nq, nkv, d, qkv_stride, out_stride, scale,
                         stream);
}
extern "C" int grit_attn_causal_window_varlen_fwd(const void* qkv, const int32_t* cu_seqlens, void* out, float* lse, int B, int max_len, int nq,
                                                  int nkv, int d, int64_t qkv_stride, int64_t out_stride, float scale, int window,
                                                  void* stream) {
  GRIT_REQUIRE(window >= 1, GRIT_E_BADARG, "grit_attn_causal_window_varlen_fwd: window=%d must be >= 1", window);
  return attn_fwd_varlen("grit_attn_causal_window_varlen_fwd", true, window, qkv, cu_seqlens, out, lse, B, max_len, nq, nkv, d, qkv_stride,
                         out_stride, scale, stream);
}
