// bf16 "NT" GEMM on MFMA for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (nn.Linear without bias), fp32 accumulate.
//
// Replaces q/k/v/o_proj and the three MLP projections of scripts/modeling_mistral_gritlm.py
// (:225-228, :655-657, :703, :177-178) -- 98 % of the encoder FLOPs.
//
// Structure (CDNA4-first, see DESIGN.md "GEMM"):
//   * 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave),
//     BK = 64, v_mfma_f32_16x16x32_bf16, 32 accumulator fragments (128 fp32 regs) per wave;
//   * both operands are K-contiguous, staged HBM->LDS by direct LDS-DMA (global_load_lds, 16 B/lane,
//     1 KiB per wave-instruction) into two 64 KiB stages, each split into four 16 KiB HALF-TILES
//     (A_h0, A_h1, W_h0, W_h1: the activation rows / weight rows every wave needs for one QUADRANT of
//     its 128x64 output);
//   * main loop = 4 phases per K-tile, one output quadrant (16 MFMAs) each; a phase is
//     {ds_read the quadrant's new fragments, issue ONE half-tile of LDS-DMA, counted vmcnt} | barrier |
//     {16 MFMAs at raised priority} | barrier.  The two wave groups (wr = 0 / 1; one wave of each on
//     every SIMD) run ONE BARRIER APART, so on every SIMD one wave feeds the matrix pipe while its
//     partner issues LDS reads and DMA: the load segments sit beside the other group's MFMAs instead
//     of in front of its own.  LDS-DMA runs 4-5 phases ahead of its first reader (never vmcnt(0));
//   * LDS image is lane-linear; the bank-conflict swizzle (16-B slot ^= (row>>1)&7, two 128-B rows =
//     one 256-B bank row) is applied to the per-lane SOURCE address and to the ds_read_b128 address;
//   * the W fragment is the MFMA "A" operand and the activation fragment the "B" operand, so each lane
//     ends up with 4 CONSECUTIVE output columns of one row -> packed bf16 stores and a
//     register-local SwiGLU (gate/up weight rows interleaved in blocks of 16);
//   * XCD-aware block remap (block b runs on XCD b % 8): every XCD walks a contiguous range of tiles,
//     4 m-tiles x 8 n-tiles in flight per XCD share A/W panels in that XCD's 4 MiB L2;
//   * big dense launches are PERSISTENT (round 3): one workgroup per CU pulls tiles from its XCD's queue (one device counter per
//     XCD: the next tile goes to whichever CU is ready first, exactly the order the hardware dispatcher would produce, so the tiles
//     an XCD runs at one time stay neighbours and keep sharing panels -- the round-2 persistent form walked a STATIC tile list
//     per CU, its CUs drifted apart and its L2-miss traffic doubled) and the K-tile stream runs THROUGH the tile boundaries: the
//     last two K-tiles of a tile stage the first half-tiles of the next one, the epilogue's stores drain under the next tile's first
//     K-tile.  Per tile this removes the workgroup launch, the cold 96 KiB prologue burst every CU issues at the same moment and
//     the exposed store drain: ~10 us of a 101 us tile at K = 4096 (tools/gemm_sustained_ab.py, profiles/r03_gemm_*).
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace grit {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 64 KiB
constexpr int HALF_BYTES = 128 * BK * 2;         // 16 KiB: 128 LDS rows x 128 B
constexpr int A_BYTES = BM * BK * 2;             // 32 KiB
constexpr int PERSIST_LDS_BYTES = 2 * STAGE_BYTES + 64 + 4 * 512 * 8;   // ring + tile mailbox + per-thread source-offset table (144.06 KiB)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;


// Grouped ("MoE") mode: the M rows are the concatenation of n_groups row ranges (counts[g] rows each, read from DEVICE memory so
// the host never syncs on the routing), group g multiplies against W + g * w_stride.  tiles_m is then an upper bound
// (sum_g ceil(counts[g]/256) <= floor(M/256) + n_groups); surplus workgroups exit.  a_rows (optional) gathers the A rows:
// sorted row r is A[a_rows[r]] -- the token permutation is folded into the per-lane LDS-DMA source address.
struct GemmGroups {
  const int32_t* counts;
  const int32_t* a_rows;
  int64_t w_stride;
  int n_groups;
};

// GRIT_EPI_ROPE: STORE with apply_rotary_pos_emb (modeling_mistral_gritlm.py:138-163) fused for the leading rope_cols columns (the q and k
// heads of the fused QKV projection, head_dim 128).  A wave then owns the column blocks {c, c+16, c+64, c+80} of one head instead of 64
// consecutive columns, so that the rotation partners (col, col+64) sit in the same lane (fragments j and j+2): the rotation is
// register-local, exactly the arithmetic of the stand-alone kernel (round the projection to bf16, rotate in fp32, round once).
struct GemmRope {
  const float* cos_tab;        // [table rows, 64] fp32
  const float* sin_tab;
  const int32_t* positions;    // nullable: position of row m = m % S
  int S;
  int rope_cols;
};

// A SECOND problem riding in the same launch (grit_gemm_bf16_nt_pair: same K, same epilogue): tiles [0, first_tiles) of the XCD-remapped
// order belong to the launch's own operands, the rest to these.  Two weight-gradient GEMMs whose tile counts are no multiples of the CU
// count (q|k|v: 384 tiles = 1.5 waves of 256 CUs, down: 896 = 3.5) fill whole waves together (1280 = 5).
struct GemmSecond {
  const uint16_t* A;
  const uint16_t* W;
  uint16_t* C;
  const uint16_t* R;
  int64_t M, lda, ldw, ldc, ldr;
  int N, tiles_m, tiles_n, first_tiles;
};

// first tile row of W fragment j of the waves in column wc (= first output column of the fragment inside the 256-wide tile)
template <bool ROPE>
__device__ __forceinline__ int wrow_of(int wc, int j) {
  return ROPE ? (wc >> 1) * 128 + (j >> 1) * 64 + (wc & 1) * 32 + (j & 1) * 16 : wc * 64 + j * 16;
}

// compiler fence between the segments of a phase: hipcc does not model the LDS-DMA builtin as a store to LDS (it would hoist or
// merge ds_reads across barriers), and it moves register-only MFMAs across asm waits (guide rule 18)
#define GRIT_SEG_FENCE()                     \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);       \
  } while (0)
#define GRIT_BARRIER()                       \
  do {                                       \
    GRIT_SEG_FENCE();                        \
    __builtin_amdgcn_s_barrier();            \
    GRIT_SEG_FENCE();                        \
  } while (0)
// K-loop barriers.  Default (round 2/3): two per phase -- load segment | B | 16 MFMAs | B -- with the second wave group one barrier
// behind, so every barrier interval has one group in its MFMAs and the other in its load segment.  -DGRIT_GEMM_BAR1: ONE per phase,
// group 0 takes it in FRONT of its MFMAs and group 1 BEHIND them:
//     group 0:  L(p) | B(p) | M(p)  L(p+1) | B(p+1) | ...          group 1:  L(p) M(p) | B(p) | L(p+1) M(p+1) | B(p+1) | ...
// so in every interval group 0's MFMAs still sit beside group 1's load segment and the other way round, but the matrix pipe changes
// hands at a barrier only every other time (the mid-interval hand-over needs no rendezvous).  The hazards hold unchanged: what L(p)
// reads was waited for by every wave (counted vmcnt at the end of its L(p-1)) in front of B(p-1), and L(p) runs behind B(p-1) in both
// groups; a slot's last reads (phase q) are complete (lgkmcnt(0) at the top of M(q)) in front of B(q+1) in both groups, and the DMA
// that refills it is issued in L(p), p >= q + 2, behind B(p-1).
#ifdef GRIT_GEMM_BAR1
#define GRIT_BAR_PRE()                                 \
  do {                                                 \
    GRIT_SEG_FENCE();                                  \
    if (wr == 0) __builtin_amdgcn_s_barrier();         \
    GRIT_SEG_FENCE();                                  \
  } while (0)
#define GRIT_BAR_POST()                                \
  do {                                                 \
    GRIT_SEG_FENCE();                                  \
    if (wr == 1) __builtin_amdgcn_s_barrier();         \
    GRIT_SEG_FENCE();                                  \
  } while (0)
#define GRIT_STAGGER(G) do { } while (0)
#else
#define GRIT_BAR_PRE() GRIT_BARRIER()
#define GRIT_BAR_POST() GRIT_BARRIER()
#define GRIT_STAGGER(G) do { if (wr == (G)) GRIT_BARRIER(); } while (0)
#endif

// A/B builds only (tools/ubench): -DGRIT_SWIGLU_BWD_DIRECT keeps the direct (row-per-lane) epilogue for SWIGLU_BWD
#ifdef GRIT_SWIGLU_BWD_DIRECT
#define GRIT_SWB_DIRECT(EPI) ((EPI) == GRIT_EPI_SWIGLU_BWD)
#else
#define GRIT_SWB_DIRECT(EPI) false
#endif
#ifdef GRIT_GEMM_STAMP
// debug build only (tools/ubench/gemm_stamp.cpp): shader-clock stamps of two waves (one per wave group) of a few workgroups at the
// seams of the persistent tile loop
__device__ unsigned long long g_stamps[8 * 2 * 64 * 8];      // [workgroup slot][wave group][tile][point]
#endif
// MFMA of one 16x16x32 step in the operand format of the instantiation
__device__ __forceinline__ f32x4_t mma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4_t mma16(f16x8_t a, f16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// F16 (round 5, the encoder's "f16_operands" precision policy): A and W hold IEEE fp16 instead of bf16 -- v_mfma_f32_16x16x32_f16 runs at
// the bf16 rate and the 16-bit staging (LDS-DMA, swizzle, fragment reads) is format-agnostic, so the K loop is the same instruction
// stream with one opcode changed.  The epilogues round ONCE, fp32 accumulator -> f16 (the bf16 instantiations reproduce the reference's
// bf16 op-by-op roundings: Linear output, rotation, silu, product): STORE, ROPE (rotation on the fp32 accumulators), SWIGLU
// (silu(gate) * up in fp32) and RESIDUAL_F32 (nothing rounded).  Values that do not fit fp16 (|v| >= 65520 -> inf) set the device's
// overflow flag word `ovf` (common.h h2_nonfinite): the host turns it into an error instead of a silently saturated embedding.
template <int EPI, bool PERSIST, bool F16 = false>
__global__ void __launch_bounds__(512) gemm_bf16_nt_k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W_all,
                                                      uint16_t* C, const uint16_t* Rsd, int64_t M_all,
                                                      int N, int K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr,
                                                      int tiles_m, int tiles_n, int GM, int remap, GemmGroups groups, GemmRope rope,
                                                      unsigned int* tile_ctr, GemmSecond second, unsigned int* ovf) {
  static_assert(!F16 || EPI == GRIT_EPI_STORE || EPI == GRIT_EPI_ROPE || EPI == GRIT_EPI_SWIGLU || EPI == GRIT_EPI_SWIGLU_STACKED ||
                    EPI == GRIT_EPI_RESIDUAL_F32 || EPI == GRIT_EPI_RESIDUAL,
                "fp16 operands: forward epilogues only");

  using frag_t = std::conditional_t<F16, f16x8_t, bf16x8_t>;
  uint32_t ovf_acc = 0;                                        // F16: OR of h2_nonfinite() over everything this lane stores
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // pair launch (non-persistent, dense): pick the problem of this workgroup from its position in the XCD-remapped order of the WHOLE
  // grid, then continue with that problem's operands and a problem-local tile id (remap = 3: "already remapped")
  int pair_wg = 0;
  if constexpr (!PERSIST) {
    if (second.first_tiles > 0) {
      const int v = (int)blockIdx.x, nv = (int)gridDim.x, xcd = v & 7, q8 = nv >> 3, r8 = nv & 7;
      pair_wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
      if (pair_wg >= second.first_tiles) {
        pair_wg -= second.first_tiles;
        A = second.A; W_all = second.W; C = second.C; Rsd = second.R; M_all = second.M; N = second.N;
        lda = second.lda; ldw = second.ldw; ldc = second.ldc; ldr = second.ldr; tiles_m = second.tiles_m; tiles_n = second.tiles_n;
      }
      remap = 3;
    }
  }
  constexpr bool ROPE = (EPI == GRIT_EPI_ROPE);
  constexpr bool STACKED = (EPI == GRIT_EPI_SWIGLU_STACKED || EPI == GRIT_EPI_SWIGLU_STACKED_SAVE);
  constexpr bool SWIGLU = (EPI == GRIT_EPI_SWIGLU || STACKED);

  // ---- tile of block v: XCD-aware id (bijective remap, guide T1) + grouped ordering (GM m-tiles per group): the 32 workgroups an
  //      XCD runs at a time are 32 consecutive tile ids = 4 m-tiles x 8 n-tiles sharing A / W panels in that XCD's 4 MiB L2.  The
  //      sharing lives on the tiles of an XCD staying IN STEP (the L2 holds about two K-tiles of the XCD's traffic): one workgroup per
  //      tile keeps them in step (equal tile times, in-order dispatch); a persistent one-workgroup-per-CU variant with the K-tile stream
  //      running through the tile boundaries was built in round 2 -- bit-identical, +3 % in isolation, equal in the model (the chip is
  //      power-limited) at TWICE the L2-miss traffic because its CUs drift apart -- and removed (profiles/r02_gemm_persistent_*.log).
  //      PERSIST: v = 8 * (index in the XCD's queue) + XCD, the same numbering one workgroup per tile would have.
  const int n_virtual = PERSIST ? tiles_m * tiles_n : (int)gridDim.x;
  const int group_sz = GM * tiles_n;
  auto tile_of = [&](int v, int64_t& m0, int64_t& M, const uint16_t*& W, int& n0) -> bool {
    const int xcd = v & 7, q8 = n_virtual >> 3, r8 = n_virtual & 7;
    int grp, in_grp;
    if (!PERSIST && remap == 2) {
      // grouped (MoE) launches: tile groups are dealt round-robin to the XCDs (XCD x runs groups x, x+8, ...), so the eight XCDs work
      // on neighbouring row blocks -- i.e. on the SAME expert -- at any time and that expert's weights stay in the Infinity Cache;
      // contiguous per-XCD ranges would keep all experts' weights (1.9 GB at the 8x7B shape) live at once
      const int li = v >> 3;
      grp = (li / group_sz) * 8 + xcd;
      in_grp = li - (li / group_sz) * group_sz;
    } else {
      const int wg = (!PERSIST && remap == 3) ? pair_wg
                     : ((PERSIST || remap) ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3) : v);
      grp = wg / group_sz;
      in_grp = wg - grp * group_sz;
    }
    const int first_m = grp * GM;
    if (first_m >= tiles_m) return false;                       // (remap == 2: surplus workgroups of the rounded-up grid)
    const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    if (in_grp >= gm * tiles_n) return false;
    const int tm = first_m + in_grp % gm, tn = in_grp / gm;
    m0 = (int64_t)tm * BM; M = M_all; W = W_all;
    if (!PERSIST && groups.counts != nullptr) {
      int t = tm, g = 0;
      int64_t off = 0;
      for (; g < groups.n_groups; ++g) {
        const int c = groups.counts[g], nt = (c + BM - 1) / BM;
        if (t < nt) break;
        t -= nt; off += c;
      }
      if (g == groups.n_groups) return false;                   // surplus tile of the upper-bound grid
      m0 = off + (int64_t)t * BM;
      M = off + groups.counts[g];                                // row limit of this group
      W = W_all + (int64_t)g * groups.w_stride;
    }
    n0 = tn * BN;
    return true;
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // PERSIST: the XCD's tile queue.  Thread 0 draws queue positions with one returning atomic each and publishes them through two
  // mailbox words BEHIND the staging ring (the same dynamic LDS array: a second __shared__ object would make hipcc drain the LDS-DMA
  // queue in front of every fragment read); a position is drawn a whole tile before it is needed (in the previous tile's epilogue,
  // where the DMA queue is drained anyway), so neither the atomic's latency nor the compiler's wait for its result touches the K loop.
  volatile int* mailbox = reinterpret_cast<volatile int*>(smem + 2 * STAGE_BYTES);
  const int my_xcd = (int)(blockIdx.x & 7);
  auto draw = [&](int slot) {                      // thread 0 only
    const unsigned pos = __hip_atomic_fetch_add(tile_ctr + my_xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mailbox[slot] = (int)(pos > 0x0fffffffu ? 0x0fffffffu : pos);
  };
  auto vtile_of_pos = [&](int pos) { return pos >= ((n_virtual >> 3) + (my_xcd < (n_virtual & 7) ? 1 : 0)) ? -1 : pos * 8 + my_xcd; };

  int64_t m0, M;
  const uint16_t* W;
  int n0;
  int vtile = blockIdx.x;
  int tile_no = 0;                                 // PERSIST: tiles done by this workgroup (mailbox slot parity)
  if constexpr (PERSIST) {
    if (tid == 0) { draw(0); draw(1); }
    __syncthreads();
    vtile = vtile_of_pos(__builtin_amdgcn_readfirstlane(mailbox[0]));
    if (vtile < 0) return;                         // more workgroups than tiles in this XCD's queue
  }
  if (!tile_of(vtile, m0, M, W, n0)) return;

  // ---- LDS-DMA sources.  Half-tile kinds: 0 = A_h0, 1 = A_h1, 2 = W_h0, 3 = W_h1; every half-tile is 128 LDS rows of 128 B and wave
  //      `wid` fills LDS rows wid*16 .. wid*16+15 of it with two 1-KiB instructions (8 rows each).
  //      A_h, LDS row p  <->  tile row (p>>6)*128 + h*64 + (p&63)             (64 rows of each M-half of the workgroup)
  //      W_h, LDS row p  <->  tile row wrow(p>>5, 2h + ((p>>4)&1)) + (p&15)   (fragments 2h, 2h+1 of each of the 4 wave columns)
  // Non-persistent launches keep one 64-bit pointer per (half-tile kind, chunk) and lane in registers (16 VGPRs).  PERSIST cannot
  // afford them: the kernel sits at 242 of 256 VGPRs, its epilogue runs INSIDE the tile loop, and every register hipcc spills there
  // comes back as a scratch load in the middle of the LDS-DMA stream -- which it waits for with vmcnt(0), draining the DMA queue
  // (the first build of this loop did that ten times per tile and gained nothing).  So PERSIST keeps a 32-bit BYTE OFFSET per (kind,
  // chunk) and lane in LDS, behind the staging ring next to the mailbox (A rows against the first row of their tile, W rows against W:
  // the host routes operands whose offsets would not fit 32 bits to the per-tile launch), and every load segment fetches the two
  // offsets of the half-tile it stages with one 8-byte LDS read (lgkmcnt, not vmcnt).  Every thread reads only what it wrote itself:
  // program order is all the synchronisation the table needs.
  const uint16_t* src[4][2];
  const uint16_t* abase[2];                                    // PERSIST: first row of the tile the A_h0 / A_h1 offsets refer to
  auto set_src = [&](int h, int64_t tm0, int64_t tM, const uint16_t* tW, int tn0) {     // sources of A_h and W_h of the tile at (tm0, tn0)
    int ln = lane;
    if (PERSIST) asm volatile("" : "+v"(ln));                  // recomputed per tile from the lane id: nothing of this is kept in
                                                               // registers across the K loop (the compiler would hoist and spill)
    const int srow = ln >> 3;                                  // row inside the 8-row chunk
    if constexpr (PERSIST) abase[h] = A + tm0 * lda;
    const int row_lim = (int)((tM - 1 - tm0) < 255 ? (tM - 1 - tm0) : 255);              // last valid row of the tile, tile-relative
    uint32_t oa[2], ow[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int p = wid * 16 + c * 8 + srow;
      const int slot = (ln & 7) ^ ((p >> 1) & 7);              // logical 16-B slot held by this physical slot
      const int ra = (p >> 6) * 128 + h * 64 + (p & 63);
      const int rw = wrow_of<ROPE>(p >> 5, 2 * h + ((p >> 4) & 1)) + (p & 15);
      int gn_row = tn0 + rw; if (gn_row > N - 1) gn_row = N - 1;
      // stacked [gate; up] weights: interleaved row n = 32 q + r is gate row 16 q + r (r < 16) or up row 16 q + r - 16
      if (STACKED) gn_row = ((gn_row >> 5) << 4) + (gn_row & 15) + ((gn_row & 16) ? (N >> 1) : 0);
      if constexpr (PERSIST) {
        const int rr = ra < row_lim ? ra : row_lim;
        oa[c] = (uint32_t)rr * (uint32_t)(lda * 2) + (uint32_t)(slot * 16);
        ow[c] = (uint32_t)gn_row * (uint32_t)(ldw * 2) + (uint32_t)(slot * 16);
      } else {
        int64_t gm_row = tm0 + ra; if (gm_row > tM - 1) gm_row = tM - 1;
        if (groups.a_rows != nullptr) gm_row = groups.a_rows[gm_row];
        src[h][c] = A + gm_row * lda + slot * 8;
        src[2 + h][c] = tW + (int64_t)gn_row * ldw + slot * 8;
      }
    }
    if constexpr (PERSIST) {        // (inline asm stores: invisible to hipcc's wait-count pass, which would order them behind the pending LDS-DMA)
      const uint32_t ta = (uint32_t)(2 * STAGE_BYTES + 64 + (h * 512 + wid * 64 + ln) * 8);
      const uint64_t va = (uint64_t)oa[0] | ((uint64_t)oa[1] << 32), vw = (uint64_t)ow[0] | ((uint64_t)ow[1] << 32);
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8192" : : "v"(ta), "v"(va), "v"(vw) : "memory");
    }
  };
  set_src(0, m0, M, W, n0);
  set_src(1, m0, M, W, n0);
  const int nk = K / BK;
  // PERSIST: the two offsets a stage() call needs are fetched ONE PHASE AHEAD by an inline-asm ds_read_b64 (`prefetch_off`), so the
  // LDS latency sits under the 16 MFMAs in between and hipcc's wait-count pass never sees the read (a C++ LDS load placed here gets an
  // s_waitcnt vmcnt(0) in front of it -- the pass orders every LDS load it knows about behind all pending LDS-DMA -- and the DMA queue
  // would drain once per phase).  The read is covered by the lgkmcnt(0) that opens the next MFMA segment; `cur_off` is made opaque right
  // before its use so that nothing computed from it can be scheduled above that wait.
  uint32_t soff_addr;                                                    // this thread's slot of kind 0 (kind k: + 4096 k)
  uint64_t cur_off = 0;                                                  // {chunk 0, chunk 1} offsets of the next stage() call
#define GRIT_PREFETCH_OFF(KIND)                                                                                                     \
  do {                                                                                                                              \
    if constexpr (PERSIST) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(cur_off) : "v"(soff_addr), "i"((KIND) * 4096) : "memory"); \
  } while (0)
  auto stage = [&](int kind, int buf, int64_t kt_) {           // K-tile kt_ of the half-tile whose row / column pointers are in src[kind]
    const int64_t kofs = kt_ * BK;
    char* base = smem + buf * STAGE_BYTES + kind * HALF_BYTES + wid * 2048;
    if constexpr (PERSIST) {
      asm volatile("" : "+v"(cur_off));
      const uint64_t ubu = reinterpret_cast<uint64_t>((kind < 2 ? abase[kind] : W_all) + kofs);    // wave-uniform: force scalar registers
      const char* ub = reinterpret_cast<const char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ubu >> 32)) << 32) |
                                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ubu));
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + (uint32_t)cur_off), (lptr_t)base, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + (uint32_t)(cur_off >> 32)), (lptr_t)(base + 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds((gptr_t)(src[kind][0] + kofs), (lptr_t)base, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src[kind][1] + kofs), (lptr_t)(base + 1024), 16, 0, 0);
    }
  };
  // ---- fragment read offsets (bytes inside a stage); the swizzle term is lane-constant because every fragment starts at a
  //      multiple of 16 LDS rows: (row>>1)&7 == (lane>>1)&7
  //      PERSIST: all of these lane constants are REBUILT from a laundered lane id at the top of every tile, so that none of them is
  //      live across the epilogue (hipcc otherwise spills a handful of them in the prologue and reloads them inside the first K-tiles
  //      of every tile -- scratch loads in the LDS-DMA stream)
  int frow, kq, s_off[2], a_off, w_off;
  auto lane_consts = [&]() {
    int ln = lane;
    if (PERSIST) asm volatile("" : "+v"(ln));
    frow = ln & 15; kq = ln >> 4;
    const int swz = (ln >> 1) & 7;
    s_off[0] = (kq ^ swz) << 4; s_off[1] = ((4 + kq) ^ swz) << 4;
    a_off = (wr * 64 + frow) * 128;                        // A frag i: + (i>>2)*HALF_BYTES + (i&3)*2048
    w_off = 2 * HALF_BYTES + (wc * 32 + frow) * 128;       // W frag j: + (j>>1)*HALF_BYTES + (j&1)*2048
    soff_addr = (uint32_t)(2 * STAGE_BYTES + 64 + (wid * 64 + ln) * 8);
  };
  lane_consts();

  f32x4_t acc[8][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // PERSIST: the zeros are made opaque, i.e. 128 v_mov per tile instead of a constant-0 accumulator operand on the first K-tile's
        // MFMAs.  With the folded form the accumulators were not live at the top of the first K-tile, hipcc packed other values into
        // their registers and then SPILLED one 16-byte accumulator fragment in phase 1 -- its scratch reload (vmcnt(0): a full drain of
        // the LDS-DMA queue) sat in the middle of phase 2's MFMAs, once per tile, in every variant but RoPE / RESIDUAL.  Opaque: 0 spills,
        // 0 scratch in all seven persistent instantiations (tools/kernel_resources.py).  -DGRIT_GEMM_FOLDED_ZERO: the old form (A/B).
#ifndef GRIT_GEMM_FOLDED_ZERO
        if (PERSIST) asm volatile("" : "+v"(acc[i][j]));
#endif
      }
  };
  zero_acc();

  frag_t wf0[2][2][2], wf1[2][2], xf[4][2];                        // [buffer][fragment][k-step]
#define GRIT_READ_W(WF, H, SB)                                                                        \
  _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                  \
      WF[jj][ks] = *reinterpret_cast<const frag_t*>((SB) + w_off + (H) * HALF_BYTES + jj * 2048 + s_off[ks])
#define GRIT_READ_X(H, SB)                                                                            \
  _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                  \
      xf[ii][ks] = *reinterpret_cast<const frag_t*>((SB) + a_off + (H) * HALF_BYTES + ii * 2048 + s_off[ks])
// The barrier BEHIND an MFMA segment is executed GRIT_GEMM_BAR_EARLY products before the segment's end (round 4; default 1, i.e. in
// front of the LAST product).  That barrier only hands the matrix pipe to the other wave group -- this wave's LDS reads completed at
// the top of the segment and the products touch registers only, so every hazard rule above holds with the barrier anywhere inside the
// segment -- and the other group's wake-up (barrier release -> lgkmcnt(0) -> s_setprio -> first v_mfma) now overlaps this group's last
// product instead of following it.  Same-box interleaved A/B against the barrier behind the segment (-DGRIT_GEMM_BAR_EARLY=0), outputs
// bit-identical on all 29 harness cases: q|k|v (RoPE) 1.024-1.025x, o_proj 1.020-1.026x, gate|up 1.010x, down 1.012-1.016x on two boxes;
// 2 products early 0.92-0.94x (the two groups' products interleave and the finishing wave's load segment starts late; with the tail at
// s_setprio 3: 1.006-1.018x), 3: 1.00-1.01x, 6: 0.97-0.99x.  The opposite change -- ONE barrier per phase, group 0 in front of its
// products and group 1 behind them (-DGRIT_GEMM_BAR1) -- is 0.964-0.988x: without the mid-interval rendezvous a group that finishes its
// load segment early issues its products BESIDE the other group's, and the pipe then idles at the end of the interval while the
// other group loads alone.  profiles/r04_gemm_barrier_ab.log.
#ifndef GRIT_GEMM_BAR_EARLY
#define GRIT_GEMM_BAR_EARLY 1
#endif
#if GRIT_GEMM_BAR_EARLY > 0 && !defined(GRIT_GEMM_BAR1)
#ifdef GRIT_GEMM_BAR_TAILPRIO
#define GRIT_TAIL_PRIO() __builtin_amdgcn_s_setprio(GRIT_GEMM_BAR_TAILPRIO)
#else
#define GRIT_TAIL_PRIO() do { } while (0)
#endif
#define GRIT_MMA(WF, I0, J0)                                                                          \
  do {                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                                               \
      const int ks = q_ >> 3, ii = (q_ >> 1) & 3, jj = q_ & 1;                                        \
      acc[(I0) + ii][(J0) + jj] =                                                                     \
          mma16(WF[jj][ks], xf[ii][ks], acc[(I0) + ii][(J0) + jj]);                                   \
      if (q_ == 15 - (GRIT_GEMM_BAR_EARLY)) {                                                         \
        GRIT_BARRIER();                                                                               \
        GRIT_TAIL_PRIO();                                                                             \
      }                                                                                               \
    }                                                                                                 \
    __builtin_amdgcn_s_setprio(0);                                                                    \
  } while (0)
#undef GRIT_BAR_POST
#define GRIT_BAR_POST() GRIT_SEG_FENCE()
#else
#define GRIT_MMA(WF, I0, J0)                                                                          \
  do {                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                  \
      _Pragma("unroll") for (int ii = 0; ii < 4; ++ii)                                                \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                              \
          acc[(I0) + ii][(J0) + jj] =                                                                 \
              mma16(WF[jj][ks], xf[ii][ks], acc[(I0) + ii][(J0) + jj]);                                   \
    __builtin_amdgcn_s_setprio(0);                                                                    \
  } while (0)
#endif
  // load segment: the LDS reads are issued ahead of the two LDS-DMA instructions, then the counted wait:
  // vmcnt(8) = "all but the 4 newest half-tiles have landed"
  // WAIT: 1 = the counted wait; 2 = drain; 3 = the counted wait unless `skip_waits` (a wave-uniform run-time flag: ONE copy of the
  // first K-tile's code serves the first tile of a workgroup and the tiles that follow an epilogue -- two copies made hipcc spill a
  // fragment at their join, and a spill reload inside the K loop is a vmcnt(0))
#ifdef GRIT_GEMM_LGKM_PRE   /* A/B: the fragment reads are waited for (and the priority raised) in FRONT of the barrier that starts the MFMA segment */
#define GRIT_LSEG_TAIL()                                         \
  do {                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_s_setprio(1);                               \
  } while (0)
#else
#define GRIT_LSEG_TAIL() do { } while (0)
#endif
#define GRIT_LSEG_END(NREADS, WAIT)                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0);                                           \
    __builtin_amdgcn_sched_group_barrier(0x10, 2, 0);                                                 \
    if constexpr ((WAIT) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                       \
    if constexpr ((WAIT) == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
    if constexpr ((WAIT) == 3) { if (!skip_waits) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }  \
    GRIT_LSEG_TAIL();                                                                                 \
  } while (0)

  // One K-tile = 4 phases, one output quadrant (16 MFMAs) each.  Half-tile stream (issue order):
  //   ... W_h0(t) A_h0(t) W_h1(t) A_h1(t) W_h0(t+1) ...;  phase p of K-tile t issues the half-tile 6 positions ahead of the newest one it
  // reads, into a slot whose last reader ran >= 2 phases earlier (the other wave group is half a phase behind, so 2 phases is the
  // minimum that has every reader's lgkmcnt(0) behind a barrier).  What a phase reads was waited for -- by EVERY wave, vmcnt(8) --
  // before the first barrier of the previous phase.  Phase 4 has no fragments of its own to read (W_h0 stays in registers), so it
  // reads the NEXT K-tile's W_h0 instead: LDS reads per phase 8 / 4 / 8 / 4.
  // k31 / k20: K-tile index of the half-tiles staged in phases 1,2 (W_h1, A_h1 of K-tile t+1) and 3,4 (W_h0, A_h0 of K-tile t+2).
  // PERSIST, `pos`: 0 = inside a tile; 1 = FIRST K-tile of a tile that follows another one in this workgroup: everything its four phases
  // read landed before the previous tile's epilogue, and the epilogue's stores are still draining (they share the vmcnt counter with
  // the LDS-DMA and are OLDER than anything issued since), so its phases do not wait at all -- the first counted wait, one K-tile
  // later, needs exactly what it always needs (the half-tile staged 4 phases earlier) and finds the stores retired under ~1.6 us of
  // matrix work; 2 = LAST K-tile of a tile: W_h0 of the next tile is read after the epilogue, and everything staged so far (the next
  // tile's first K-tile and a half) must have landed before the stores go out.
  bool skip_waits = false;
  auto ktile = [&](int64_t k31, int64_t k20, auto bufc, auto posc) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr int POS = decltype(posc)::value;
    constexpr int WAIT = POS == 1 ? 3 : 1;
    const char* sb = smem + BUF * STAGE_BYTES;
    const char* sbn = smem + (BUF ^ 1) * STAGE_BYTES;
    // phase 1: quadrant (A_h0, W_h0)
    GRIT_READ_X(0, sb);
    stage(3, BUF ^ 1, k31);
    GRIT_PREFETCH_OFF(1);
    GRIT_LSEG_END(8, WAIT);           // W_h1(t) landed (read in phase 2)
    GRIT_BAR_PRE();
    GRIT_MMA(wf0[BUF], 0, 0);
    GRIT_BAR_POST();
    // phase 2: quadrant (A_h0, W_h1)
    GRIT_READ_W(wf1, 1, sb);
    stage(1, BUF ^ 1, k31);
    GRIT_PREFETCH_OFF(2);
    GRIT_LSEG_END(4, WAIT);           // A_h1(t) landed (read in phase 3)
    GRIT_BAR_PRE();
    GRIT_MMA(wf1, 0, 2);
    GRIT_BAR_POST();
    // phase 3: quadrant (A_h1, W_h1)
    GRIT_READ_X(1, sb);
    stage(2, BUF, k20);
    GRIT_PREFETCH_OFF(0);
    GRIT_LSEG_END(8, WAIT);           // W_h0(t+1) landed (read in phase 4)
    GRIT_BAR_PRE();
    GRIT_MMA(wf1, 4, 2);
    GRIT_BAR_POST();
    // phase 4: quadrant (A_h1, W_h0); W_h0 of the next K-tile -> the other buffer's fragment registers
    if constexpr (POS != 2) GRIT_READ_W(wf0[BUF ^ 1], 0, sbn);
    stage(0, BUF, k20);
    GRIT_PREFETCH_OFF(3);
    GRIT_LSEG_END(POS == 2 ? 0 : 4, POS == 2 ? 2 : WAIT);      // A_h0(t+1) landed (read in phase 1 of the next K-tile)
    GRIT_BAR_PRE();
    GRIT_MMA(wf0[BUF], 4, 0);
    GRIT_BAR_POST();
  };
  const std::integral_constant<int, 0> B0{};
  const std::integral_constant<int, 1> B1{};
  const std::integral_constant<int, 0> MID{};
  const std::integral_constant<int, 1> FIRST{};
  const std::integral_constant<int, 2> LAST{};
  auto kclamp = [&](int kt) { return (int64_t)(kt < nk ? kt : nk - 1); };        // past the end: re-stage the last K-tile (nobody reads it)

  // prologue (PERSIST: every offset pair is fetched and waited for in place; set_src's table writes are this thread's own)
#define GRIT_OFF_NOW(KIND)                                                                \
  do {                                                                                    \
    GRIT_PREFETCH_OFF(KIND);                                                              \
    if constexpr (PERSIST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
  } while (0)
  GRIT_OFF_NOW(2); stage(2, 0, 0); GRIT_OFF_NOW(0); stage(0, 0, 0); GRIT_OFF_NOW(3); stage(3, 0, 0); GRIT_OFF_NOW(1); stage(1, 0, 0);
  GRIT_OFF_NOW(2); stage(2, 1, kclamp(1)); GRIT_OFF_NOW(0); stage(0, 1, kclamp(1));
  GRIT_PREFETCH_OFF(3);                   // for phase 1 of the first K-tile (covered by the lgkmcnt(0) of the W_h0 fragment read below)
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // W_h0(0), A_h0(0)
  GRIT_BARRIER();
  GRIT_READ_W(wf0[0], 0, smem);
  GRIT_STAGGER(1);                    // the second wave group runs one barrier behind the first (BAR1: no stagger in the count)

  // F16: the overflow bits of a tile are published at the END OF ITS EPILOGUE and the accumulator word cleared, so that it is not live
  // across the K loop (in the persistent form one more live VGPR there pushed the RESIDUAL instantiation over the 256-register cliff:
  // 35 spills = scratch reloads, each a vmcnt(0) drain of the LDS-DMA queue).  (Rows beyond M are computed from clamped, valid rows:
  // no false alarms.)
#define GRIT_OVF_FLUSH()                                                   \
  do {                                                                     \
    if constexpr (F16) {                                                   \
      if (ovf_acc != 0 && ovf != nullptr) atomicOr(ovf, 1u);               \
      ovf_acc = 0;                                                         \
    }                                                                      \
  } while (0)
  // ---- epilogue: lane holds C[m][n..n+3], m = frag row base + (lane&15), n = frag col base + (lane>>4)*4
  auto wrow = [&](int j) { return wrow_of<ROPE>(wc, j); };
  auto epilogue = [&](int64_t m0, int64_t M, int n0) {
  int ln_e = lane;
  if (PERSIST) asm volatile("" : "+v"(ln_e));               // per-lane output coordinates are rebuilt per tile, not carried through the K loop
  const int frow = ln_e & 15, kq = ln_e >> 4;
  const int64_t mrow = m0 + wr * 128 + frow;
  const int ncol = n0 + wc * 64 + kq * 4;
  // The 8 row blocks of the lane go out in batches of RB: all table / residual loads of a batch are issued up front, so the epilogue pays
  // one memory latency per batch.  One workgroup per tile: one batch (the K loop's registers are dead).  PERSIST: the K loop's state
  // stays live across the epilogue, smaller batches keep it out of scratch memory.
  constexpr int RB = !PERSIST ? 8 : (ROPE ? 2 : 8);
#pragma unroll
  for (int ib = 0; ib < 8; ib += RB) {
  if (PERSIST) __builtin_amdgcn_sched_barrier(0);
  if constexpr (ROPE) {
    if (n0 + (wc >> 1) * 128 < rope.rope_cols) {        // this wave's head is a q or k head (uniform per wave)
      const int m0_mod = (int)(m0 % rope.S);              // block-uniform: the only 64-bit division
#pragma unroll
      for (int i = ib; i < ib + RB; ++i) {
        const int64_t m = mrow + i * 16;
        const int64_t mc = m < M ? m : M - 1;
        const int pos = rope.positions ? rope.positions[mc] : (m0_mod + (int)(mc - m0)) % rope.S;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          const int c1 = (wc & 1) * 32 + p2 * 16 + kq * 4;                       // head-local column of fragment p2, lane's 4 columns
          const float4 cs = *reinterpret_cast<const float4*>(rope.cos_tab + (int64_t)pos * 64 + c1);
          const float4 sn = *reinterpret_cast<const float4*>(rope.sin_tab + (int64_t)pos * 64 + c1);
          const float cc[4] = {cs.x, cs.y, cs.z, cs.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x1 = acc[i][p2][r], x2 = acc[i][p2 + 2][r];                     // F16: the rotation runs on the fp32 accumulators
            if constexpr (!F16) {
              const uint32_t xr = pack2bf_hw(x1, x2);                             // q/k = bf16(linear) first (:655-657)
              x1 = bflo(xr); x2 = bfhi(xr);
            }
            acc[i][p2][r] = rope_lo(x1, x2, cc[r], ss[r]);
            acc[i][p2 + 2][r] = rope_hi(x1, x2, cc[r], ss[r]);
          }
        }
      }
    }
  }
  uint4 rpre[8][2];
  if constexpr (EPI == GRIT_EPI_RESIDUAL) {
#pragma unroll
    for (int i = ib; i < ib + RB; ++i)
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        const int64_t m = mrow + i * 16;
        const int n = n0 + wrow(2 * jq + (kq & 1)) + (kq >> 1) * 8;
        rpre[i][jq] = (m < M && n < N) ? *reinterpret_cast<const uint4*>(Rsd + m * ldr + n) : make_uint4(0, 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = ib; i < ib + RB; ++i) {
    const int64_t m = mrow + i * 16;
    if (m >= M) continue;
    if constexpr (SWIGLU) {
      // fragments (0,1) and (2,3) are (gate, up) pairs -> two 16-column output blocks; the same permlane16 exchange gives every
      // lane 8 consecutive output columns
      const int nb = n0 + wc * 64;             // multiple of 64
      if (nb < N) {
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (F16) {            // one rounding: silu(gate) * up in fp32, rounded to f16 by the store below
            o0[r] = silu_f(acc[i][0][r]) * acc[i][1][r];
            o1[r] = silu_f(acc[i][2][r]) * acc[i][3][r];
          } else {
          // bf16 roundings through v_cvt_pk_bf16_f32 (RNE, two per instruction) instead of the 5-op bit trick
          const uint32_t gu0 = pack2bf_hw(acc[i][0][r], acc[i][1][r]), gu1 = pack2bf_hw(acc[i][2][r], acc[i][3][r]);
          const uint32_t ss = pack2bf_hw(silu_f(bflo(gu0)), silu_f(bflo(gu1)));
          o0[r] = bflo(ss) * bfhi(gu0);
          o1[r] = bfhi(ss) * bfhi(gu1);
          }
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0[r]), __float_as_uint(o1[r]), false, false);
          o0[r] = __uint_as_float(sw[0]); o1[r] = __uint_as_float(sw[1]);
        }
        const int oc = (nb >> 1) + (kq & 1) * 16 + (kq >> 1) * 8;
        if (2 * oc < N) {
          const uint4 ov = make_uint4(pack2_op<F16>(o0[0], o0[1]), pack2_op<F16>(o0[2], o0[3]), pack2_op<F16>(o1[0], o1[1]), pack2_op<F16>(o1[2], o1[3]));
          if constexpr (F16) ovf_acc |= h2_nonfinite(ov.x) | h2_nonfinite(ov.y) | h2_nonfinite(ov.z) | h2_nonfinite(ov.w);
          *reinterpret_cast<uint4*>(C + m * ldc + oc) = ov;
        }
        if constexpr (EPI == GRIT_EPI_SWIGLU_STACKED_SAVE) {
          // the bf16 pre-activations the backward pass needs, [gate | up] at the SAME columns as the activation (the exchange that gives
          // the lane 8 consecutive activation columns gives it the 8 matching gate and up columns)
          uint32_t gq[2][4], uq[2][4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t gu0 = pack2bf_hw(acc[i][0][r], acc[i][1][r]), gu1 = pack2bf_hw(acc[i][2][r], acc[i][3][r]);
            const auto sg = __builtin_amdgcn_permlane16_swap(gu0 & 0xffffu, gu1 & 0xffffu, false, false);
            const auto su = __builtin_amdgcn_permlane16_swap(gu0 >> 16, gu1 >> 16, false, false);
            gq[0][r] = sg[0]; gq[1][r] = sg[1]; uq[0][r] = su[0]; uq[1][r] = su[1];
          }
          if (2 * oc < N) {
            uint16_t* gsave = const_cast<uint16_t*>(Rsd) + m * ldr + oc;
            *reinterpret_cast<uint4*>(gsave) = make_uint4(gq[0][0] | (gq[0][1] << 16), gq[0][2] | (gq[0][3] << 16), gq[1][0] | (gq[1][1] << 16),
                                                          gq[1][2] | (gq[1][3] << 16));
            *reinterpret_cast<uint4*>(gsave + (N >> 1)) = make_uint4(uq[0][0] | (uq[0][1] << 16), uq[0][2] | (uq[0][3] << 16),
                                                                     uq[1][0] | (uq[1][1] << 16), uq[1][2] | (uq[1][3] << 16));
          }
        }
      }
    } else {
      // 16-byte stores: v_permlane16_swap exchanges the 16-lane rows of two adjacent n-fragments so that every lane ends up with
      // 8 CONSECUTIVE columns (rows 0/2 of the wave keep fragment j, rows 1/3 take fragment j+1) -- half the store (and residual
      // load) instructions of the natural 4-columns-per-lane layout; the store tail is issue-bound (guide T21)
#pragma unroll
      for (int jp = 0; jp < 4; jp += 2) {
        f32x4_t lo = acc[i][jp], hi4 = acc[i][jp + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo[r]), __float_as_uint(hi4[r]), false, false);
          lo[r] = __uint_as_float(sw[0]); hi4[r] = __uint_as_float(sw[1]);
        }
        const int n = n0 + wrow(jp + (kq & 1)) + (kq >> 1) * 8;
        if (n >= N) continue;
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        if constexpr (EPI == GRIT_EPI_RESIDUAL) {
          // the reference rounds the Linear output to bf16 before the residual add (:769,:775)
          const uint4 rv = rpre[i][jp >> 1];
          const uint32_t ra[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t rr = pack2_op<F16>(v[2 * e], v[2 * e + 1]);
            v[2 * e] = lo16_op<F16>(rr) + lo16_op<F16>(ra[e]); v[2 * e + 1] = hi16_op<F16>(rr) + hi16_op<F16>(ra[e]);
          }
        }
        if constexpr (EPI == GRIT_EPI_SWIGLU_BWD) {
          // d_act = bf16(acc) (what the un-fused path stores and re-reads), then the arithmetic of grit_swiglu_bwd on the saved [gate | up]
          const uint4 gv = *reinterpret_cast<const uint4*>(Rsd + m * ldr + n), uv = *reinterpret_cast<const uint4*>(Rsd + m * ldr + N + n);
          const uint32_t ga[4] = {gv.x, gv.y, gv.z, gv.w}, ua[4] = {uv.x, uv.y, uv.z, uv.w};
          uint32_t og[4], ou[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t dd = pack2bf_hw(v[2 * e], v[2 * e + 1]);
            float dg0, du0, dg1, du1;
            swiglu_bwd_elem(bflo(dd), bflo(ga[e]), bflo(ua[e]), dg0, du0);
            swiglu_bwd_elem(bfhi(dd), bfhi(ga[e]), bfhi(ua[e]), dg1, du1);
            og[e] = pack2bf_hw(dg0, dg1);
            ou[e] = pack2bf_hw(du0, du1);
          }
          *reinterpret_cast<uint4*>(C + m * ldc + n) = make_uint4(og[0], og[1], og[2], og[3]);
          *reinterpret_cast<uint4*>(C + m * ldc + N + n) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
          continue;
        }
        const uint4 pk = make_uint4(pack2_op<F16>(v[0], v[1]), pack2_op<F16>(v[2], v[3]), pack2_op<F16>(v[4], v[5]), pack2_op<F16>(v[6], v[7]));
        if constexpr (F16) ovf_acc |= h2_nonfinite(pk.x) | h2_nonfinite(pk.y) | h2_nonfinite(pk.z) | h2_nonfinite(pk.w);
        *reinterpret_cast<uint4*>(C + m * ldc + n) = pk;
      }
    }
  }
  }
  GRIT_OVF_FLUSH();
  };

  // ---- epilogue through LDS (round 3; used by STORE / RESIDUAL, written for RoPE and SwiGLU too).  In the MFMA layout consecutive lanes hold
  //      consecutive ROWS (lane & 15 = row, lane >> 4 = which 16-byte piece), so one 16-byte store (or residual load) instruction
  //      touches 16 rows x 64 B: every lane of a 16-lane pass hits a different cache line -- the epilogue was issue-bound in the texture
  //      path (stamps: 5.6-8.5 k cycles per tile for the stores alone, 17-19 k with the residual loads).  Each wave now turns its row
  //      block (16 rows x 128 B) through a private 4 KiB piece of the staging ring (dead at this point: the A_h1 / W_h1 slots of stage 1)
  //      and goes to memory with lanes 8 r .. 8 r + 7 on the 8 pieces of ONE row: 8 full 128-byte lines per instruction (RoPE: two 64-byte
  //      segments per row; SwiGLU: 4 lanes per 64-byte output row).  The arithmetic is untouched -- the residual is added after the
  //      turn, to the same bf16-rounded value -- so the results are bit-identical to the direct epilogue (tools/ubench/gemm_ab.cpp).
  auto epilogue_lds = [&](int64_t m0, int64_t M, int n0) {
    int ln_e = lane;
    if (PERSIST) asm volatile("" : "+v"(ln_e));
    const int frow = ln_e & 15, kq = ln_e >> 4;
    char* xp = smem + STAGE_BYTES + (wid < 4 ? HALF_BYTES : 3 * HALF_BYTES) + (wid & 3) * 4096;
    const int64_t mbase = m0 + wr * 128;
    if constexpr (EPI == GRIT_EPI_RESIDUAL_F32) {
      // fp32 residual stream: C (fp32, may alias the residual) = residual + acc, nothing rounded.  A row block of the wave is 16 rows x
      // 256 B of fp32: the lane's four 16-byte fragments pieces go into the wave's 4 KiB transposition space (unit ^= row & 7: conflict-free
      // ds_write_b128 / ds_read_b128), come back with 16 lanes on the 16 pieces of ONE row -- 4 rows x 256 B = 8 full lines per load /
      // store instruction instead of 16 rows x 64 B -- and meet the residual, fetched four row blocks ahead (ring of 4 x 4 float4).
      const float* R32 = reinterpret_cast<const float*>(Rsd);
      float* C32 = reinterpret_cast<float*>(C);
      const int tr4 = ln_e >> 4, tu16 = ln_e & 15;
      const int ncol32 = n0 + wc * 64 + tu16 * 4;
      float4 rpre32[4][4];
      auto r32_fetch = [&](int i) {
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
          const int64_t m = mbase + i * 16 + p4 * 4 + tr4;
          rpre32[i & 3][p4] = (m < M && ncol32 < N) ? *reinterpret_cast<const float4*>(R32 + m * ldr + ncol32) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) r32_fetch(i);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PERSIST && (i & 1) == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4_t*>(xp + frow * 256 + (((j * 4 + kq) ^ (frow & 7)) << 4)) = acc[i][j];
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                                     // lgkmcnt(0): the wave's own pieces are in the buffer
        asm volatile("" ::: "memory");
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
          const int row = p4 * 4 + tr4;
          const float4 piece = *reinterpret_cast<const float4*>(xp + row * 256 + ((tu16 ^ (row & 7)) << 4));
          const int64_t m = mbase + i * 16 + row;
          const float4 rv = rpre32[i & 3][p4];
          if (m < M && ncol32 < N)
            *reinterpret_cast<float4*>(C32 + m * ldc + ncol32) = make_float4(piece.x + rv.x, piece.y + rv.y, piece.z + rv.z, piece.w + rv.w);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);                                     // the pieces are in registers before the buffer is rewritten
        asm volatile("" ::: "memory");
        if (i + 4 < 8) r32_fetch(i + 4);
      }
      return;
    }
    const int tr = ln_e >> 3, tu = ln_e & 7;                                   // transposed side: row (+ 8) and 16-byte unit of the row
    const int ncol_t = n0 + wrow(tu >> 1) + (tu & 1) * 8;
    uint4 rpre[8][2];
    if constexpr (EPI == GRIT_EPI_RESIDUAL) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int64_t m = mbase + i * 16 + hh * 8 + tr;
          rpre[i][hh] = (m < M && ncol_t < N) ? *reinterpret_cast<const uint4*>(Rsd + m * ldr + ncol_t) : make_uint4(0, 0, 0, 0);
        }
    }
    // SWIGLU_BWD: the saved [gate | up] rows of a row block are fetched FOUR row blocks ahead into a ring of 4 x 2 x 2 16-byte registers
    // (all 32 at once would be 128 VGPRs next to the 128 accumulators); after the turn a lane owns 8 consecutive columns of one row, so
    // each of these loads -- and each of the two stores -- is a full 128-byte line per 8 lanes instead of 16 rows x 64 B per instruction
    uint4 gpre[4][2], upre[4][2];
    auto swb_fetch = [&](int i) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int64_t m = mbase + i * 16 + hh * 8 + tr;
        const bool in = m < M && ncol_t < N;
        gpre[i & 3][hh] = in ? *reinterpret_cast<const uint4*>(Rsd + m * ldr + ncol_t) : make_uint4(0, 0, 0, 0);
        upre[i & 3][hh] = in ? *reinterpret_cast<const uint4*>(Rsd + m * ldr + N + ncol_t) : make_uint4(0, 0, 0, 0);
      }
    };
    if constexpr (EPI == GRIT_EPI_SWIGLU_BWD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) swb_fetch(i);
    }
    bool rotate = false;
    int m0_mod = 0;
    if constexpr (ROPE) {
      rotate = n0 + (wc >> 1) * 128 < rope.rope_cols;                           // this wave's head is a q or k head (uniform per wave)
      m0_mod = (int)(m0 % rope.S);                                              // block-uniform: the only 64-bit division
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (PERSIST && (i & 1) == 0) __builtin_amdgcn_sched_barrier(0);           // bound the live range of the per-row-block temporaries
      if constexpr (ROPE) {
        if (rotate) {
          const int64_t m = mbase + frow + i * 16;
          const int64_t mc = m < M ? m : M - 1;
          const int pos = rope.positions ? rope.positions[mc] : (m0_mod + (int)(mc - m0)) % rope.S;
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            const int c1 = (wc & 1) * 32 + p2 * 16 + kq * 4;                     // head-local column of fragment p2, lane's 4 columns
            const float4 cs = *reinterpret_cast<const float4*>(rope.cos_tab + (int64_t)pos * 64 + c1);
            const float4 sn = *reinterpret_cast<const float4*>(rope.sin_tab + (int64_t)pos * 64 + c1);
            const float cc[4] = {cs.x, cs.y, cs.z, cs.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x1 = acc[i][p2][r], x2 = acc[i][p2 + 2][r];
              if constexpr (!F16) {
                const uint32_t xr = pack2bf_hw(x1, x2);                           // q/k = bf16(linear) first (:655-657)
                x1 = bflo(xr); x2 = bfhi(xr);
              }
              acc[i][p2][r] = rope_lo(x1, x2, cc[r], ss[r]);
              acc[i][p2 + 2][r] = rope_hi(x1, x2, cc[r], ss[r]);
            }
          }
        }
      }
      char* xb = xp + (i & 1) * 2048;
      if constexpr (SWIGLU) {
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (F16) {
            o0[r] = silu_f(acc[i][0][r]) * acc[i][1][r];
            o1[r] = silu_f(acc[i][2][r]) * acc[i][3][r];
          } else {
          const uint32_t gu0 = pack2bf_hw(acc[i][0][r], acc[i][1][r]), gu1 = pack2bf_hw(acc[i][2][r], acc[i][3][r]);
          const uint32_t ss = pack2bf_hw(silu_f(bflo(gu0)), silu_f(bflo(gu1)));
          o0[r] = bflo(ss) * bfhi(gu0);
          o1[r] = bfhi(ss) * bfhi(gu1);
          }
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0[r]), __float_as_uint(o1[r]), false, false);
          o0[r] = __uint_as_float(sw[0]); o1[r] = __uint_as_float(sw[1]);
        }
        // 16 rows x 64 B (32 output columns of the wave): unit q of the row, swizzled by (row >> 1) & 3
        const int q = (kq & 1) * 2 + (kq >> 1);
        const uint4 ov = make_uint4(pack2_op<F16>(o0[0], o0[1]), pack2_op<F16>(o0[2], o0[3]), pack2_op<F16>(o1[0], o1[1]), pack2_op<F16>(o1[2], o1[3]));
        if constexpr (F16) ovf_acc |= h2_nonfinite(ov.x) | h2_nonfinite(ov.y) | h2_nonfinite(ov.z) | h2_nonfinite(ov.w);
        *reinterpret_cast<uint4*>(xb + frow * 64 + ((q ^ ((frow >> 1) & 3)) << 4)) = ov;
      } else {
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          f32x4_t lo = acc[i][jp], hi4 = acc[i][jp + 1];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo[r]), __float_as_uint(hi4[r]), false, false);
            lo[r] = __uint_as_float(sw[0]); hi4[r] = __uint_as_float(sw[1]);
          }
          const int pidx = (jp + (kq & 1)) * 2 + (kq >> 1);                      // 16-byte unit of the row: fragment jp + (kq & 1), half kq >> 1
          const uint4 ov = make_uint4(pack2_op<F16>(lo[0], lo[1]), pack2_op<F16>(lo[2], lo[3]), pack2_op<F16>(hi4[0], hi4[1]), pack2_op<F16>(hi4[2], hi4[3]));
          if constexpr (F16) ovf_acc |= h2_nonfinite(ov.x) | h2_nonfinite(ov.y) | h2_nonfinite(ov.z) | h2_nonfinite(ov.w);
          *reinterpret_cast<uint4*>(xb + frow * 128 + ((pidx ^ (frow & 7)) << 4)) = ov;
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);                                       // lgkmcnt(0): the wave's own pieces are in the buffer
      asm volatile("" ::: "memory");
      if constexpr (SWIGLU) {
        const int row = ln_e >> 2, unit = ln_e & 3;
        const uint4 piece = *reinterpret_cast<const uint4*>(xb + row * 64 + ((unit ^ ((row >> 1) & 3)) << 4));
        const int64_t m = mbase + i * 16 + row;
        const int oc = ((n0 + wc * 64) >> 1) + unit * 8;
        if (m < M && 2 * oc < N) *reinterpret_cast<uint4*>(C + m * ldc + oc) = piece;
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int row = hh * 8 + tr;
          uint4 piece = *reinterpret_cast<const uint4*>(xb + row * 128 + ((tu ^ (row & 7)) << 4));
          const int64_t m = mbase + i * 16 + row;
          if constexpr (EPI == GRIT_EPI_RESIDUAL) {
            // the reference rounds the Linear output to bf16 before the residual add (:769,:775): `piece` holds the rounded values
            const uint4 rv = rpre[i][hh];
            if constexpr (F16) {
              // fp16 residual stream: f16(acc) + residual as ONE v_pk_add_f16 per pair -- the IEEE fp16 sum of two fp16 values, i.e. exactly
              // what unpack / fp32 add / round would give, without the unpack temporaries (they spilled 35 VGPRs in the persistent form)
              piece = make_uint4(pk_add_f16(piece.x, rv.x), pk_add_f16(piece.y, rv.y), pk_add_f16(piece.z, rv.z), pk_add_f16(piece.w, rv.w));
              ovf_acc |= h2_nonfinite(piece.x) | h2_nonfinite(piece.y) | h2_nonfinite(piece.z) | h2_nonfinite(piece.w);
            } else {
            piece = make_uint4(pack2bf_hw(bflo(piece.x) + bflo(rv.x), bfhi(piece.x) + bfhi(rv.x)), pack2bf_hw(bflo(piece.y) + bflo(rv.y), bfhi(piece.y) + bfhi(rv.y)),
                               pack2bf_hw(bflo(piece.z) + bflo(rv.z), bfhi(piece.z) + bfhi(rv.z)), pack2bf_hw(bflo(piece.w) + bflo(rv.w), bfhi(piece.w) + bfhi(rv.w)));
            }
          }
          if constexpr (EPI == GRIT_EPI_SWIGLU_BWD) {
            // `piece` = d_act rounded to bf16 (what the un-fused path stores and re-reads), then the arithmetic of grit_swiglu_bwd
            const uint4 gv = gpre[i & 3][hh], uv = upre[i & 3][hh];
            const uint32_t da[4] = {piece.x, piece.y, piece.z, piece.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ua[4] = {uv.x, uv.y, uv.z, uv.w};
            uint32_t og[4], ou[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float dg0, du0, dg1, du1;
              swiglu_bwd_elem(bflo(da[e]), bflo(ga[e]), bflo(ua[e]), dg0, du0);
              swiglu_bwd_elem(bfhi(da[e]), bfhi(ga[e]), bfhi(ua[e]), dg1, du1);
              og[e] = pack2bf_hw(dg0, dg1);
              ou[e] = pack2bf_hw(du0, du1);
            }
            if (m < M && ncol_t < N) {
              *reinterpret_cast<uint4*>(C + m * ldc + ncol_t) = make_uint4(og[0], og[1], og[2], og[3]);
              *reinterpret_cast<uint4*>(C + m * ldc + N + ncol_t) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
            }
            continue;
          }
          if (m < M && ncol_t < N) *reinterpret_cast<uint4*>(C + m * ldc + ncol_t) = piece;
        }
        if constexpr (EPI == GRIT_EPI_SWIGLU_BWD) {
          if (i + 4 < 8) swb_fetch(i + 4);            // the slot just consumed takes the rows of the block four ahead
        }
      }
    }
    GRIT_OVF_FLUSH();
  };
  // Measured against the direct epilogue on the same box (ratios to the round-2 kernel, M = 131072; profiles/r03_gemm_ab_lds_epilogue.log):
  // RESIDUAL K = 4096 1.019 -> 1.049, K = 14336 1.003 -> 1.011, STORE 1.029 -> 1.039, RoPE 1.026 -> 1.025, SwiGLU 1.027 -> 1.017 (its output
  // rows are 64 bytes: nothing to merge, one more LDS round trip).  So: the turn for the full-width epilogues, the direct form for the rest.
  // What the stamps say is left of the residual epilogue (14.5 k cycles per tile): 128 KiB read + 128 KiB written per CU by all 256 CUs
  // at the same moment = 64 MB at ~7 TB/s -- the seam of a lock-stepped launch is an HBM burst, not an issue problem.
  // SWIGLU_BWD (round 3, second half): two 16-byte loads + two 16-byte stores per 8 outputs made the direct form the slowest epilogue of the
  // training step (1135 TF on the d_act GEMM against 1400-1470 for the other dense launches of a chunk); through the turn they are full lines.
  constexpr bool LDS_EPI = (EPI == GRIT_EPI_STORE || EPI == GRIT_EPI_RESIDUAL || EPI == GRIT_EPI_SWIGLU_BWD || EPI == GRIT_EPI_RESIDUAL_F32) && !GRIT_SWB_DIRECT(EPI);

  if constexpr (!PERSIST) {
    for (int kt = 0; kt < nk; kt += 2) {
      ktile(kclamp(kt + 1), kclamp(kt + 2), B0, MID);
      if (kt + 1 < nk) ktile(kclamp(kt + 2), kclamp(kt + 3), B1, MID);
    }
    GRIT_STAGGER(0);                    // barrier counts of the two groups match again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail DMAs must land before the LDS is released
    GRIT_SEG_FENCE();
    if constexpr (LDS_EPI) {
      GRIT_BARRIER();                   // ... everybody's, before the ring is reused as the epilogue's transposition space
      epilogue_lds(m0, M, n0);
    } else {
      epilogue(m0, M, n0);
    }
  } else {
#ifdef GRIT_GEMM_STAMP
#define GRIT_STAMP(ID)                                                                                                              \
  do {                                                                                                                              \
    if ((tid & 255) == 0 && (blockIdx.x % 37) == 0 && blockIdx.x / 37 < 8 && tile_no < 64)                                          \
      g_stamps[(((blockIdx.x / 37) * 2 + (tid >> 8)) * 64 + tile_no) * 8 + (ID)] = __builtin_readcyclecounter();                    \
  } while (0)
#else
#define GRIT_STAMP(ID) do { } while (0)
#endif
    // One workgroup per CU walks the tiles it draws from its XCD's queue with the K-tile stream running THROUGH the tile boundaries.
    // (nk even and >= 4: the host guarantees it.)
    for (;;) {
      GRIT_STAMP(0);
      ktile(1, 2, B0, FIRST);                              // (skip_waits: false for the workgroup's first tile)
      GRIT_STAMP(1);
      ktile(2, 3, B1, MID);
      GRIT_STAMP(2);
      for (int kt = 2; kt < nk - 2; kt += 2) {
        ktile(kt + 1, kt + 2, B0, MID);
        ktile(kt + 2, kt + 3, B1, MID);
      }
      GRIT_STAMP(3);
      // next tile (drawn one tile ago); no next tile: re-stage this one (harmless, nobody reads it)
      int64_t m0n = m0, Mn = M;
      const uint16_t* Wn = W;
      int n0n = n0;
      // (inline asm: a C++ LDS load here would get hipcc's vmcnt(0) in front of it and drain the DMA queue once per tile)
      int pos_next;
      {
        const uint32_t mb_addr = (uint32_t)(2 * STAGE_BYTES + 4 * ((tile_no + 1) & 1));
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos_next) : "v"(mb_addr) : "memory");
      }
      const int vnext = vtile_of_pos(__builtin_amdgcn_readfirstlane(pos_next));
      const bool more = vnext >= 0 && tile_of(vnext, m0n, Mn, Wn, n0n);
      set_src(0, m0n, Mn, Wn, n0n);                        // W_h0 / A_h0 of the current tile were last staged two K-tiles ago
      ktile(nk - 1, 0, B0, MID);
      GRIT_STAMP(4);
      set_src(1, m0n, Mn, Wn, n0n);
      GRIT_OFF_NOW(3);                                     // the pair fetched in the previous phase 4 predates set_src(1)
      ktile(0, 1, B1, LAST);
      GRIT_STAMP(5);
      // The two wave groups run one barrier apart inside a tile; at the seam they are brought IN STEP (the leading group waits half a
      // phase for the other one's last MFMAs) so that both run their epilogues at the same time -- left staggered, each group sat at
      // its next barrier for the whole length of the other group's epilogue (stamped: 2 x 4.3 k cycles per tile with the plain-store
      // epilogue, 2 x 13 k with the residual one) -- and the stagger is re-established in front of the next tile.
      GRIT_STAGGER(0);
      if (tid == 0 && more) draw(tile_no & 1);             // queue position of the tile after the next one (this tile's slot is free)
      if constexpr (LDS_EPI) {
        // (the groups are in step: every wave has passed the last barrier of the last K-tile, so every read of the A_h1 / W_h1 slots of
        //  stage 1 -- the transposition space -- is done)
        epilogue_lds(m0, M, n0);
        GRIT_BARRIER();                                    // nobody stages the next tile's W_h1 / A_h1 into them before everybody is out
      } else {
        epilogue(m0, M, n0);
      }
      GRIT_STAMP(6);
      if (!more) break;
      lane_consts();
      zero_acc();
      GRIT_READ_W(wf0[0], 0, smem);                        // W_h0(0) of the next tile: landed before the last K-tile's final wait
      GRIT_STAGGER(1);                                     // one barrier behind again
      vtile = vnext; m0 = m0n; M = Mn; W = Wn; n0 = n0n; ++tile_no; skip_waits = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GRIT_SEG_FENCE();
  }
#undef GRIT_OVF_FLUSH
#undef GRIT_READ_W
#undef GRIT_PREFETCH_OFF
#undef GRIT_OFF_NOW
#undef GRIT_READ_X
#undef GRIT_MMA
#undef GRIT_LSEG_END
}

// Launch knobs for A/B runs (read once, thread-safe static initialisation): GRIT_GEMM_GM=<n> m-tiles per scheduling group (default 4:
// 4 m x 8 n tiles in flight per XCD), GRIT_GEMM_NOREMAP=1 disables the XCD remap, GRIT_GEMM_RR=1 deals tile groups round-robin to the
// XCDs for dense launches too (default: grouped launches only), GRIT_GEMM_NOPERSIST=1 always launches one workgroup per tile,
// GRIT_GEMM_PERSIST_MAXKT=<n> lifts the K-tile bound of the persistent form (default 128).
struct GemmKnobs {
  int gm, remap, rr_all, persist, persist_max_kt;
};
static const GemmKnobs& gemm_knobs() {
  static const GemmKnobs k = [] {
    GemmKnobs v;
    const char* e = getenv("GRIT_GEMM_GM");
    v.gm = (e && atoi(e) > 0) ? atoi(e) : 4;
    v.remap = getenv("GRIT_GEMM_NOREMAP") ? 0 : 1;
    v.rr_all = getenv("GRIT_GEMM_RR") ? 1 : 0;
    v.persist = getenv("GRIT_GEMM_NOPERSIST") ? 0 : 1;
    e = getenv("GRIT_GEMM_PERSIST_MAXKT");                       // A/B knob: largest K-tile count that still takes the persistent form
    v.persist_max_kt = (e && atoi(e) > 0) ? atoi(e) : 1024;
    return v;
  }();
  return k;
}

static int device_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int v = cached[dev & 63].load(std::memory_order_relaxed);
  if (v == 0) {
    (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    if (v <= 0) v = 256;
    cached[dev & 63].store(v, std::memory_order_relaxed);
  }
  return v;
}

// Tile-queue counters of the persistent launches: CTR_SETS sets of 8 counters (one per XCD) in a module-scope device array (the library
// never allocates), organised as ONE RING PER STREAM: up to CTR_STREAMS (device, stream) pairs each own CTR_PER_STREAM consecutive sets.
// A launch takes the next set of ITS STREAM'S ring, clears it with a 32-byte memset node on that stream and hands it to the kernel.
// Why per stream: a set is re-cleared CTR_PER_STREAM launches later, and only stream order guarantees that the kernel that used it has
// finished by then -- with one process-wide ring (round 3) a persistent kernel still pending on another stream (a side stream, an
// autograd worker's stream, a stream parked on an event) could have had its live queue re-zeroed after the ring wrapped: tiles computed
// twice or skipped, silently.  Rings are RECLAIMED (round 5): when every slot is taken, the least recently used
// ring whose stream has nothing pending (hipStreamQuery == success -- its kernels are done, its counters dead -- or the handle is no
// longer a stream at all: a destroyed side stream) goes to the new (device, stream) pair, so 8 GPUs x (compute + side stream) plus
// re-created CU-masked communication streams no longer exhaust the table for the life of the process.  Only when all CTR_STREAMS rings
// have work in flight does a launch take the per-tile form (reported once on stderr).
constexpr int CTR_STREAMS = 16, CTR_PER_STREAM = 256, CTR_SETS = CTR_STREAMS * CTR_PER_STREAM;
__device__ unsigned int g_tile_ctr[CTR_SETS * 8];
struct CtrRing {
  int dev;
  hipStream_t st;
  uint32_t next;
  bool used;
  uint64_t last_use;
};
static unsigned int* next_counter_set(hipStream_t st) {
  static std::mutex mu;
  static CtrRing rings[CTR_STREAMS];
  static std::atomic<unsigned int*> base[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  unsigned int* b = base[dev & 63].load(std::memory_order_acquire);
  if (b == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tile_ctr)) != hipSuccess || p == nullptr) return nullptr;
    b = (unsigned int*)p;
    base[dev & 63].store(b, std::memory_order_release);
  }
  int slot = -1;
  uint32_t idx = 0;
  {
    static uint64_t tick = 0;
    static bool warned = false;
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < CTR_STREAMS && slot < 0; ++i)
      if (rings[i].used && rings[i].dev == dev && rings[i].st == st) slot = i;
    for (int i = 0; i < CTR_STREAMS && slot < 0; ++i)
      if (!rings[i].used) {
        rings[i] = CtrRing{dev, st, 0u, true, 0};
        slot = i;
      }
    if (slot < 0) {
      // reclaim: least recently used ring whose stream is idle (or gone).  A ring of THIS device only: its counter sets live in this
      // device's copy of g_tile_ctr, and a stream of another device cannot be queried without switching devices.
      uint64_t best = ~0ull;
      for (int i = 0; i < CTR_STREAMS; ++i) {
        if (rings[i].dev != dev || rings[i].last_use >= best) continue;
        const hipError_t q = hipStreamQuery(rings[i].st);
        // reclaim ONLY a stream that is idle (its kernels are done, its counters dead) or provably gone (an invalid handle: a destroyed
        // side stream).  Everything else is "busy": not-ready, and in particular the capture-related errors -- a live stream that is being
        // graph-captured also fails the query, and a graph already captured on it would replay on counter sets handed to another stream
        // 
        if (q != hipSuccess) {
          (void)hipGetLastError();
          if (q != hipErrorInvalidHandle && q != hipErrorInvalidResourceHandle && q != hipErrorContextIsDestroyed) continue;
        }
        best = rings[i].last_use;
        slot = i;
      }
      if (slot >= 0) rings[slot] = CtrRing{dev, st, 0u, true, 0};
    }
    if (slot < 0) {                                   // every ring busy: per-tile launch form
      if (!warned) {
        warned = true;
        fprintf(stderr, "gritlm_hip: %d streams with persistent GEMMs in flight; further streams take the per-tile launch form\n", CTR_STREAMS);
      }
      return nullptr;
    }
    rings[slot].last_use = ++tick;
    idx = rings[slot].next++ % CTR_PER_STREAM;
    // the clearing memset is enqueued while the table is still locked: from the moment the slot is assigned the stream has work pending,
    // so a concurrent reclaim (hipStreamQuery under the same mutex) can never see it idle between the assignment and the launch
    unsigned int* set = b + ((size_t)slot * CTR_PER_STREAM + idx) * 8;
    if (hipMemsetAsync(set, 0, 8 * sizeof(unsigned int), st) != hipSuccess) return nullptr;
    return set;
  }
}

// the 128 KiB dynamic-LDS opt-in is a per-device function attribute: set it once per (instantiation, device)
template <typename KernelT>
static void ensure_lds_optin(KernelT kernel, std::atomic<uint64_t>& done, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
}

template <int EPI, bool F16 = false>
static int launch_gemm(const void* A, const void* W, void* C, const void* R, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                       int64_t ldc, int64_t ldr, hipStream_t st, GemmGroups grp = GemmGroups{nullptr, nullptr, 0, 0},
                       GemmRope rope = GemmRope{nullptr, nullptr, nullptr, 0, 0}) {
  unsigned int* ovf = nullptr;
  if constexpr (F16) {
    ovf = f16_flag_ptr();
    GRIT_REQUIRE(ovf != nullptr, GRIT_E_LAUNCH, "grit_gemm_f16_nt: the overflow flag word of this device is not reachable");
  }
  const int tiles_m = grp.counts ? (int)(M / BM) + grp.n_groups : (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
  static std::atomic<uint64_t> optin{0}, optin_p{0};
  const GemmKnobs& kn = gemm_knobs();
  // grouped launches: round-robin tile groups over the XCDs (remap 2) on a grid rounded up to 8 x whole groups
  const int total_groups = (tiles_m + kn.gm - 1) / kn.gm;
  const bool rr = (grp.counts || kn.rr_all) && kn.remap;
  const unsigned nblocks = rr ? (unsigned)(8 * ((total_groups + 7) / 8) * kn.gm * tiles_n) : (unsigned)(tiles_m * tiles_n);
  const int remap_mode = rr ? 2 : kn.remap;
  // persistent form: dense launches with an even number (>= 4) of K-tiles and at least two tiles per CU; everything else takes one
  // workgroup per tile.  (Until the accumulator spill of the first K-tile was removed -- zero_acc above -- the form measured 0.99x at
  // K = 14336, where the seam is only 3 % of a tile, and was bounded to <= 128 K-tiles; without the per-tile queue drain it wins there
  // too: down_proj 1.013x at M = 131072, 1.042x at a training chunk, the gate|up weight gradient (K = 16384 tokens) 1.031x, bit-identical;
  // profiles/r03_gemm_ab_opaque_zero.log.  GRIT_GEMM_PERSIST_MAXKT restores a bound for A/B runs.)
  const int n_cu = device_cu_count();
  const bool off32 = 256 * lda * 2 + 128 < (1ll << 32) && (int64_t)N * ldw * 2 + 128 < (1ll << 32);     // PERSIST's 32-bit source offsets
  if (kn.persist && !rr && grp.counts == nullptr && (K / BK) % 2 == 0 && K / BK >= 4 && K / BK <= kn.persist_max_kt && (int64_t)tiles_m * tiles_n >= 2 * (int64_t)n_cu &&
      n_cu % 8 == 0 && off32) {
    unsigned int* ctr = next_counter_set(st);
    if (ctr != nullptr) {
      ensure_lds_optin(gemm_bf16_nt_k<EPI, true, F16>, optin_p, PERSIST_LDS_BYTES);
      hipLaunchKernelGGL((gemm_bf16_nt_k<EPI, true, F16>), dim3((unsigned)n_cu), dim3(512), PERSIST_LDS_BYTES, st, (const uint16_t*)A,
                         (const uint16_t*)W, (uint16_t*)C, (const uint16_t*)R, M, N, K, lda, ldw, ldc, ldr, tiles_m, tiles_n, kn.gm,
                         remap_mode, grp, rope, ctr, GemmSecond{}, ovf);
      GRIT_CHECK_LAUNCH("grit_gemm_bf16_nt (persistent)");
      return GRIT_OK;
    }
    (void)hipGetLastError();          // no counter set (symbol lookup / memset refused, e.g. inside a stream capture): per-tile launch
  }
  ensure_lds_optin(gemm_bf16_nt_k<EPI, false, F16>, optin, 2 * STAGE_BYTES);
  hipLaunchKernelGGL((gemm_bf16_nt_k<EPI, false, F16>), dim3(nblocks), dim3(512), 2 * STAGE_BYTES, st, (const uint16_t*)A, (const uint16_t*)W,
                     (uint16_t*)C, (const uint16_t*)R, M, N, K, lda, ldw, ldc, ldr, tiles_m, tiles_n, kn.gm, remap_mode, grp, rope,
                     (unsigned int*)nullptr, GemmSecond{}, ovf);
  GRIT_CHECK_LAUNCH("grit_gemm_bf16_nt");
  return GRIT_OK;
}

template <int EPI>
static int launch_gemm_pair(const void* A1, const void* W1, void* C1, const void* R1, int64_t M1, int N1, int64_t lda1, int64_t ldw1,
                            int64_t ldc1, int64_t ldr1, const void* A2, const void* W2, void* C2, const void* R2, int64_t M2, int N2,
                            int64_t lda2, int64_t ldw2, int64_t ldc2, int64_t ldr2, int K, hipStream_t st) {
  const int tm1 = (int)((M1 + BM - 1) / BM), tn1 = (N1 + BN - 1) / BN, tm2 = (int)((M2 + BM - 1) / BM), tn2 = (N2 + BN - 1) / BN;
  static std::atomic<uint64_t> optin{0};
  const GemmKnobs& kn = gemm_knobs();
  const GemmSecond sec{(const uint16_t*)A2, (const uint16_t*)W2, (uint16_t*)C2, (const uint16_t*)R2, M2, lda2, ldw2, ldc2, ldr2, N2, tm2, tn2, tm1 * tn1};
  ensure_lds_optin(gemm_bf16_nt_k<EPI, false>, optin, 2 * STAGE_BYTES);
  hipLaunchKernelGGL((gemm_bf16_nt_k<EPI, false>), dim3((unsigned)(tm1 * tn1 + tm2 * tn2)), dim3(512), 2 * STAGE_BYTES, st, (const uint16_t*)A1,
                     (const uint16_t*)W1, (uint16_t*)C1, (const uint16_t*)R1, M1, N1, K, lda1, ldw1, ldc1, ldr1, tm1, tn1, kn.gm, 1,
                     GemmGroups{nullptr, nullptr, 0, 0}, GemmRope{nullptr, nullptr, nullptr, 0, 0}, (unsigned int*)nullptr, sec,
                     (unsigned int*)nullptr);
  GRIT_CHECK_LAUNCH("grit_gemm_bf16_nt_pair");
  return GRIT_OK;
}

}  // namespace grit

using namespace grit;

#ifdef GRIT_GEMM_STAMP
extern "C" int grit_debug_gemm_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(grit::g_stamps), sizeof(grit::g_stamps));
}
#endif
extern "C" int grit_swiglu_block(void) { return 16; }

extern "C" int grit_gemm_bf16_nt_grouped(const void* A, const int32_t* a_rows, const void* W, void* C, const int32_t* group_counts,
                                         int num_groups, int64_t M_total, int N, int K, int64_t lda, int64_t ldw, int64_t w_group_stride,
                                         int64_t ldc, int epilogue, void* stream) {
  if (M_total == 0) return GRIT_OK;
  GRIT_REQUIRE(A && W && C && group_counts, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped: null pointer");
  GRIT_REQUIRE(M_total >= 0 && N > 0 && K > 0 && num_groups > 0 && num_groups <= 1024, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped: bad sizes");
  GRIT_REQUIRE(K % 64 == 0 && N % 16 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_grouped: K=%d must be a multiple of 64, N=%d of 16", K, N);
  GRIT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && w_group_stride % 8 == 0 && lda >= K && ldw >= K, GRIT_E_BADARG,
               "grit_gemm_bf16_nt_grouped: bad leading dimensions");
  GRIT_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C), GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)(M_total / BM + num_groups) * ((N + BN - 1) / BN) < (1ll << 31), GRIT_E_UNSUPPORTED,
               "grit_gemm_bf16_nt_grouped: too many tiles");
  const GemmGroups grp{group_counts, a_rows, w_group_stride, num_groups};
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE:
      GRIT_REQUIRE(ldc >= N, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped: ldc < N");
      return launch_gemm<GRIT_EPI_STORE>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    case GRIT_EPI_SWIGLU:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_grouped: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    default:
      GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped: epilogue %d not available (STORE, SWIGLU)", epilogue);
  }
  return GRIT_OK;
}

// Grouped launch with the training epilogues (Mixtral's expert MLP, forward with saved pre-activations and backward):
// STORE, SWIGLU (interleaved weights), SWIGLU_STACKED, SWIGLU_STACKED_SAVE ([gate | up] of every sorted row through `residual`),
// SWIGLU_BWD (saved [gate | up] of the sorted rows in `residual`).  Same semantics per group as grit_gemm_bf16_nt.
extern "C" int grit_gemm_bf16_nt_grouped_epi(const void* A, const int32_t* a_rows, const void* W, void* C, const void* residual,
                                             const int32_t* group_counts, int num_groups, int64_t M_total, int N, int K, int64_t lda,
                                             int64_t ldw, int64_t w_group_stride, int64_t ldc, int64_t ldr, int epilogue, void* stream) {
  if (M_total == 0) return GRIT_OK;
  GRIT_REQUIRE(A && W && C && group_counts, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped_epi: null pointer");
  GRIT_REQUIRE(M_total >= 0 && N > 0 && K > 0 && num_groups > 0 && num_groups <= 1024, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped_epi: bad sizes");
  GRIT_REQUIRE(K % 64 == 0 && N % 16 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_grouped_epi: K=%d must be a multiple of 64, N=%d of 16", K, N);
  GRIT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && w_group_stride % 8 == 0 && lda >= K && ldw >= K, GRIT_E_BADARG,
               "grit_gemm_bf16_nt_grouped_epi: bad leading dimensions");
  GRIT_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C), GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped_epi: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)(M_total / BM + num_groups) * ((N + BN - 1) / BN) < (1ll << 31), GRIT_E_UNSUPPORTED,
               "grit_gemm_bf16_nt_grouped_epi: too many tiles");
  const GemmGroups grp{group_counts, a_rows, w_group_stride, num_groups};
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE:
      GRIT_REQUIRE(ldc >= N, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped_epi: ldc < N");
      return launch_gemm<GRIT_EPI_STORE>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    case GRIT_EPI_SWIGLU:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_grouped_epi: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    case GRIT_EPI_SWIGLU_STACKED:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_grouped_epi: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU_STACKED>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    case GRIT_EPI_SWIGLU_STACKED_SAVE:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_grouped_epi: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      GRIT_REQUIRE(residual && ldr % 8 == 0 && ldr >= N && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_bf16_nt_grouped_epi: SWIGLU_STACKED_SAVE writes [gate | up] through `residual` (ldr >= N)");
      return launch_gemm<GRIT_EPI_SWIGLU_STACKED_SAVE>(A, W, C, residual, M_total, N, K, lda, ldw, ldc, ldr, st, grp);
    case GRIT_EPI_SWIGLU_BWD:
      GRIT_REQUIRE(residual && ldr % 8 == 0 && ldr >= 2 * (int64_t)N && ldc >= 2 * (int64_t)N && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_bf16_nt_grouped_epi: SWIGLU_BWD needs the saved [gate | up] in `residual` (ldr >= 2N) and ldc >= 2N");
      return launch_gemm<GRIT_EPI_SWIGLU_BWD>(A, W, C, residual, M_total, N, K, lda, ldw, ldc, ldr, st, grp);
    default:
      GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemm_bf16_nt_grouped_epi: epilogue %d not available", epilogue);
  }
  return GRIT_OK;
}

extern "C" int grit_gemm_bf16_nt_rope(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc,
                                      const float* cos_tab, const float* sin_tab, const int32_t* positions, int S, int table_rows,
                                      int rope_cols, void* stream) {
  if (M == 0) return GRIT_OK;
  GRIT_REQUIRE(A && W && C && cos_tab && sin_tab, GRIT_E_BADARG, "grit_gemm_bf16_nt_rope: null pointer");
  GRIT_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, GRIT_E_BADARG, "grit_gemm_bf16_nt_rope: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
  GRIT_REQUIRE(N % 128 == 0 && rope_cols % 128 == 0 && rope_cols >= 0 && rope_cols <= N, GRIT_E_UNSUPPORTED,
               "grit_gemm_bf16_nt_rope: N=%d and rope_cols=%d must be multiples of the head size 128", N, rope_cols);
  GRIT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= K && ldw >= K && ldc >= N, GRIT_E_BADARG,
               "grit_gemm_bf16_nt_rope: bad leading dimensions");
  GRIT_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && aligned16(cos_tab) && aligned16(sin_tab), GRIT_E_BADARG,
               "grit_gemm_bf16_nt_rope: pointers must be 16-byte aligned");
  GRIT_REQUIRE((positions != nullptr) ? table_rows > 0 : (S > 0 && table_rows >= S), GRIT_E_BADARG,
               "grit_gemm_bf16_nt_rope: positions or S (<= table rows) required");
  GRIT_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31), GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_rope: too many tiles");
  const GemmRope rope{cos_tab, sin_tab, positions, S > 0 ? S : 1, rope_cols};
  return launch_gemm<GRIT_EPI_ROPE>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, (hipStream_t)stream, GemmGroups{nullptr, nullptr, 0, 0}, rope);
}

// Two dense GEMMs with the same K and epilogue (STORE or RESIDUAL) in ONE launch: the weight-gradient GEMMs of a layer whose tile counts
// leave half a wave of CUs idle when launched alone.  Per problem the semantics (and the bits) are those of grit_gemm_bf16_nt.
extern "C" int grit_gemm_bf16_nt_pair(const void* A1, const void* W1, void* C1, const void* R1, int64_t M1, int N1, int64_t lda1, int64_t ldw1,
                                      int64_t ldc1, int64_t ldr1, const void* A2, const void* W2, void* C2, const void* R2, int64_t M2, int N2,
                                      int64_t lda2, int64_t ldw2, int64_t ldc2, int64_t ldr2, int K, int epilogue, void* stream) {
  GRIT_REQUIRE(A1 && W1 && C1 && A2 && W2 && C2, GRIT_E_BADARG, "grit_gemm_bf16_nt_pair: null pointer");
  GRIT_REQUIRE(M1 > 0 && N1 > 0 && M2 > 0 && N2 > 0 && K > 0, GRIT_E_BADARG, "grit_gemm_bf16_nt_pair: bad sizes");
  GRIT_REQUIRE(K % 64 == 0 && N1 % 16 == 0 && N2 % 16 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_pair: K must be a multiple of 64, N of 16");
  GRIT_REQUIRE(lda1 % 8 == 0 && ldw1 % 8 == 0 && ldc1 % 8 == 0 && lda2 % 8 == 0 && ldw2 % 8 == 0 && ldc2 % 8 == 0 && lda1 >= K && ldw1 >= K &&
                   lda2 >= K && ldw2 >= K && ldc1 >= N1 && ldc2 >= N2,
               GRIT_E_BADARG, "grit_gemm_bf16_nt_pair: bad leading dimensions");
  GRIT_REQUIRE(aligned16(A1) && aligned16(W1) && aligned16(C1) && aligned16(A2) && aligned16(W2) && aligned16(C2), GRIT_E_BADARG,
               "grit_gemm_bf16_nt_pair: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)((M1 + BM - 1) / BM) * ((N1 + BN - 1) / BN) + (int64_t)((M2 + BM - 1) / BM) * ((N2 + BN - 1) / BN) < (1ll << 31),
               GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt_pair: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE:
      return launch_gemm_pair<GRIT_EPI_STORE>(A1, W1, C1, nullptr, M1, N1, lda1, ldw1, ldc1, 0, A2, W2, C2, nullptr, M2, N2, lda2, ldw2, ldc2, 0, K, st);
    case GRIT_EPI_RESIDUAL:
      GRIT_REQUIRE(R1 && R2 && ldr1 % 8 == 0 && ldr2 % 8 == 0 && ldr1 >= N1 && ldr2 >= N2 && aligned16(R1) && aligned16(R2), GRIT_E_BADARG,
                   "grit_gemm_bf16_nt_pair: RESIDUAL epilogue needs both residuals (ldr >= N)");
      return launch_gemm_pair<GRIT_EPI_RESIDUAL>(A1, W1, C1, R1, M1, N1, lda1, ldw1, ldc1, ldr1, A2, W2, C2, R2, M2, N2, lda2, ldw2, ldc2, ldr2, K, st);
    default:
      GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemm_bf16_nt_pair: epilogue %d not available (STORE, RESIDUAL)", epilogue);
  }
  return GRIT_OK;
}

extern "C" int grit_gemm_bf16_nt(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                                 int64_t ldc, int epilogue, const void* residual, int64_t ldr, void* stream) {
  if (M == 0) return GRIT_OK;  // empty batch (empty tensors have null data pointers)
  GRIT_REQUIRE(A && W && C, GRIT_E_BADARG, "grit_gemm_bf16_nt: null pointer");
  GRIT_REQUIRE(M >= 0 && N > 0 && K > 0, GRIT_E_BADARG, "grit_gemm_bf16_nt: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
  GRIT_REQUIRE(K % 64 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt: K=%d must be a multiple of 64", K);
  GRIT_REQUIRE(N % 16 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt: N=%d must be a multiple of 16", N);
  GRIT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= K && ldw >= K, GRIT_E_BADARG,
               "grit_gemm_bf16_nt: bad leading dimensions lda=%lld ldw=%lld ldc=%lld", (long long)lda, (long long)ldw, (long long)ldc);
  GRIT_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C), GRIT_E_BADARG, "grit_gemm_bf16_nt: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31), GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt: too many tiles");
  if (M == 0) return GRIT_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE:
      GRIT_REQUIRE(ldc >= N, GRIT_E_BADARG, "grit_gemm_bf16_nt: ldc < N");
      return launch_gemm<GRIT_EPI_STORE>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, st);
    case GRIT_EPI_RESIDUAL:
      GRIT_REQUIRE(residual && ldr % 8 == 0 && ldr >= N && ldc >= N && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_bf16_nt: RESIDUAL epilogue needs residual with ldr >= N");
      return launch_gemm<GRIT_EPI_RESIDUAL>(A, W, C, residual, M, N, K, lda, ldw, ldc, ldr, st);
    case GRIT_EPI_RESIDUAL_F32:
      GRIT_REQUIRE(residual && ldr % 4 == 0 && ldr >= N && ldc >= N && ldc % 4 == 0 && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_bf16_nt: RESIDUAL_F32 epilogue needs an fp32 residual with ldr >= N (C and residual are fp32, ldc / ldr in floats)");
      return launch_gemm<GRIT_EPI_RESIDUAL_F32>(A, W, C, residual, M, N, K, lda, ldw, ldc, ldr, st);
    case GRIT_EPI_SWIGLU:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, st);
    case GRIT_EPI_SWIGLU_STACKED:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU_STACKED>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, st);
    case GRIT_EPI_SWIGLU_STACKED_SAVE:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      GRIT_REQUIRE(residual && ldr % 8 == 0 && ldr >= N && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_bf16_nt: SWIGLU_STACKED_SAVE writes [gate | up] through `residual` (ldr >= N)");
      return launch_gemm<GRIT_EPI_SWIGLU_STACKED_SAVE>(A, W, C, residual, M, N, K, lda, ldw, ldc, ldr, st);
    case GRIT_EPI_SWIGLU_BWD:
      GRIT_REQUIRE(residual && ldr % 8 == 0 && ldr >= 2 * (int64_t)N && ldc >= 2 * (int64_t)N && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_bf16_nt: SWIGLU_BWD needs the saved [gate | up] in `residual` (ldr >= 2N) and ldc >= 2N");
      return launch_gemm<GRIT_EPI_SWIGLU_BWD>(A, W, C, residual, M, N, K, lda, ldw, ldc, ldr, st);
    default:
      GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemm_bf16_nt: unknown epilogue %d", epilogue);
  }
  return GRIT_OK;
}

// ---- fp16-operand instantiations (the encoder's "f16_operands" precision policy; forward only, dense) ----
// Same contract as grit_gemm_bf16_nt with A, W (and C for STORE / SWIGLU) holding IEEE fp16.  STORE: C = f16(acc); SWIGLU: C =
// f16(silu(gate) * up) from the fp32 accumulators (one rounding; interleaved weight rows, grit_swiglu_block()); RESIDUAL: C (fp16) =
// f16(f16(acc) + residual) with an fp16 residual (the fp16 residual stream of the "f16_stream" policy); RESIDUAL_F32: C (fp32)
// = residual (fp32) + acc.  A result beyond the fp16 range sets the device's overflow flag (grit_f16_overflow_flag).
extern "C" int grit_gemm_f16_nt(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                                int64_t ldc, int epilogue, const void* residual, int64_t ldr, void* stream) {
  if (M == 0) return GRIT_OK;
  GRIT_REQUIRE(A && W && C, GRIT_E_BADARG, "grit_gemm_f16_nt: null pointer");
  GRIT_REQUIRE(M >= 0 && N > 0 && K > 0, GRIT_E_BADARG, "grit_gemm_f16_nt: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
  GRIT_REQUIRE(K % 64 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt: K=%d must be a multiple of 64", K);
  GRIT_REQUIRE(N % 16 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt: N=%d must be a multiple of 16", N);
  GRIT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= K && ldw >= K, GRIT_E_BADARG,
               "grit_gemm_f16_nt: bad leading dimensions lda=%lld ldw=%lld ldc=%lld", (long long)lda, (long long)ldw, (long long)ldc);
  GRIT_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C), GRIT_E_BADARG, "grit_gemm_f16_nt: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31), GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE:
      GRIT_REQUIRE(ldc >= N, GRIT_E_BADARG, "grit_gemm_f16_nt: ldc < N");
      return launch_gemm<GRIT_EPI_STORE, true>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, st);
    case GRIT_EPI_RESIDUAL:
      GRIT_REQUIRE(residual && ldr % 8 == 0 && ldr >= N && ldc >= N && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_f16_nt: RESIDUAL epilogue needs an fp16 residual with ldr >= N");
      return launch_gemm<GRIT_EPI_RESIDUAL, true>(A, W, C, residual, M, N, K, lda, ldw, ldc, ldr, st);
    case GRIT_EPI_RESIDUAL_F32:
      GRIT_REQUIRE(residual && ldr % 4 == 0 && ldr >= N && ldc >= N && ldc % 4 == 0 && aligned16(residual), GRIT_E_BADARG,
                   "grit_gemm_f16_nt: RESIDUAL_F32 epilogue needs an fp32 residual with ldr >= N (C and residual are fp32, ldc / ldr in floats)");
      return launch_gemm<GRIT_EPI_RESIDUAL_F32, true>(A, W, C, residual, M, N, K, lda, ldw, ldc, ldr, st);
    case GRIT_EPI_SWIGLU:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU, true>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, st);
    case GRIT_EPI_SWIGLU_STACKED:     // the training engine's [gate; up] weights (pass 1 of GradCache under an fp16 policy)
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU_STACKED, true>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, st);
    default:
      GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemm_f16_nt: epilogue %d not available (STORE, SWIGLU, SWIGLU_STACKED, RESIDUAL, RESIDUAL_F32)", epilogue);
  }
  return GRIT_OK;
}

// grit_gemm_bf16_nt_grouped on fp16 operands (Mixtral's expert MLP under the "f16_operands" policy): A (gathered through a_rows), W
// [E,N,K] and C in fp16, one rounding of the fp32 accumulator (SWIGLU / SWIGLU_STACKED: of silu(gate) * up evaluated in fp32).
extern "C" int grit_gemm_f16_nt_grouped(const void* A, const int32_t* a_rows, const void* W, void* C, const int32_t* group_counts,
                                        int num_groups, int64_t M_total, int N, int K, int64_t lda, int64_t ldw, int64_t w_group_stride,
                                        int64_t ldc, int epilogue, void* stream) {
  if (M_total == 0) return GRIT_OK;
  GRIT_REQUIRE(A && W && C && group_counts, GRIT_E_BADARG, "grit_gemm_f16_nt_grouped: null pointer");
  GRIT_REQUIRE(M_total >= 0 && N > 0 && K > 0 && num_groups > 0 && num_groups <= 1024, GRIT_E_BADARG, "grit_gemm_f16_nt_grouped: bad sizes");
  GRIT_REQUIRE(K % 64 == 0 && N % 16 == 0, GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt_grouped: K=%d must be a multiple of 64, N=%d of 16", K, N);
  GRIT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && w_group_stride % 8 == 0 && lda >= K && ldw >= K, GRIT_E_BADARG,
               "grit_gemm_f16_nt_grouped: bad leading dimensions");
  GRIT_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C), GRIT_E_BADARG, "grit_gemm_f16_nt_grouped: pointers must be 16-byte aligned");
  GRIT_REQUIRE((int64_t)(M_total / BM + num_groups) * ((N + BN - 1) / BN) < (1ll << 31), GRIT_E_UNSUPPORTED,
               "grit_gemm_f16_nt_grouped: too many tiles");
  const GemmGroups grp{group_counts, a_rows, w_group_stride, num_groups};
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case GRIT_EPI_STORE:
      GRIT_REQUIRE(ldc >= N, GRIT_E_BADARG, "grit_gemm_f16_nt_grouped: ldc < N");
      return launch_gemm<GRIT_EPI_STORE, true>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    case GRIT_EPI_SWIGLU:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt_grouped: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU, true>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    case GRIT_EPI_SWIGLU_STACKED:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt_grouped: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU_STACKED, true>(A, W, C, nullptr, M_total, N, K, lda, ldw, ldc, 0, st, grp);
    default:
      GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemm_f16_nt_grouped: epilogue %d not available (STORE, SWIGLU, SWIGLU_STACKED)", epilogue);
  }
  return GRIT_OK;
}

// grit_gemm_bf16_nt_rope on fp16 operands: the rotation runs on the fp32 accumulators with the UNROUNDED fp32 tables the caller passes,
// q | k | v are rounded once, to fp16.
extern "C" int grit_gemm_f16_nt_rope(const void* A, const void* W, void* C, int64_t M, int N, int K, int64_t lda, int64_t ldw, int64_t ldc,
                                     const float* cos_tab, const float* sin_tab, const int32_t* positions, int S, int table_rows,
                                     int rope_cols, void* stream) {
  if (M == 0) return GRIT_OK;
  GRIT_REQUIRE(A && W && C && cos_tab && sin_tab, GRIT_E_BADARG, "grit_gemm_f16_nt_rope: null pointer");
  GRIT_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, GRIT_E_BADARG, "grit_gemm_f16_nt_rope: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
  GRIT_REQUIRE(N % 128 == 0 && rope_cols % 128 == 0 && rope_cols >= 0 && rope_cols <= N, GRIT_E_UNSUPPORTED,
               "grit_gemm_f16_nt_rope: N=%d and rope_cols=%d must be multiples of the head size 128", N, rope_cols);
  GRIT_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= K && ldw >= K && ldc >= N, GRIT_E_BADARG,
               "grit_gemm_f16_nt_rope: bad leading dimensions");
  GRIT_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && aligned16(cos_tab) && aligned16(sin_tab), GRIT_E_BADARG,
               "grit_gemm_f16_nt_rope: pointers must be 16-byte aligned");
  GRIT_REQUIRE((positions != nullptr) ? table_rows > 0 : (S > 0 && table_rows >= S), GRIT_E_BADARG,
               "grit_gemm_f16_nt_rope: positions or S (<= table rows) required");
  GRIT_REQUIRE((int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) < (1ll << 31), GRIT_E_UNSUPPORTED, "grit_gemm_f16_nt_rope: too many tiles");
  const GemmRope rope{cos_tab, sin_tab, positions, S > 0 ? S : 1, rope_cols};
  return launch_gemm<GRIT_EPI_ROPE, true>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, (hipStream_t)stream, GemmGroups{nullptr, nullptr, 0, 0}, rope);
}
