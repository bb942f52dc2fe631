// bf16 "NT" GEMM on MFMA for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (nn.Linear without bias), fp32 accumulate.
//
// Replaces q/k/v/o_proj and the three MLP projections of scripts/modeling_mistral_gritlm.py
// (:225-228, :655-657, :703, :177-178) -- 98 % of the encoder FLOPs.
//
// Structure (CDNA4-first, see DESIGN.md "GEMM"):
//   * 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave),
//     BK = 64, v_mfma_f32_16x16x32_bf16, 32 accumulator fragments (128 fp32 regs) per wave;
//   * both operands are K-contiguous, staged HBM->LDS by direct LDS-DMA (global_load_lds, 16 B/lane,
//     1 KiB per wave-instruction), two 64 KiB LDS stages, one barrier per K-tile;
//   * LDS image is lane-linear; the bank-conflict swizzle (16-B slot ^= (row>>1)&7, two 128-B rows =
//     one 256-B bank row) is applied to the per-lane SOURCE address and to the ds_read_b128 address;
//   * the W fragment is the MFMA "A" operand and the activation fragment the "B" operand, so each lane
//     ends up with 4 CONSECUTIVE output columns of one row -> 8-byte packed bf16 stores and a
//     register-local SwiGLU (gate/up weight rows interleaved in blocks of 16);
//   * XCD-aware block remap (block b runs on XCD b % 8): every XCD walks a contiguous range of tiles,
//     8 m-tiles x 4 n-tiles in flight per XCD share A/W panels in that XCD's 4 MiB L2.
#include <stdlib.h>

#include "common.h"

namespace grit {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 64 KiB
constexpr int A_BYTES = BM * BK * 2;             // 32 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// Workgroup barrier that also retires this wave's LDS-DMA (global_load_lds) and LDS reads.  hipcc does not model the
// LDS-DMA builtin as a store to LDS: it may hoist later ds_reads above a plain __syncthreads() and omit the vmcnt
// wait when the DMA was issued in a previous loop iteration (seen in the ISA of the pipelined kernel) -- so the wait
// is explicit and both sides are fenced for the compiler with "memory" clobbers.
__device__ __forceinline__ void lds_dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

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

template <int EPI, int ABL = 0>
__global__ void __launch_bounds__(512) gemm_bf16_nt_k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W_all,
                                                      uint16_t* C, const uint16_t* Rsd, int64_t M_all,
                                                      int N, int K, int64_t lda, int64_t ldw, int64_t ldc, int64_t ldr,
                                                      int tiles_m, int tiles_n, int GM, int remap, GemmGroups groups, GemmRope rope) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool ROPE = (EPI == GRIT_EPI_ROPE);

  // ---- XCD-aware tile id (bijective remap, guide T1) + grouped ordering (GM m-tiles per group)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int group_sz = GM * tiles_n;
  int grp, in_grp;
  if (remap == 2) {
    // grouped (MoE) launches: tile groups are dealt round-robin to the XCDs (XCD x runs groups x, x+8, ...), so the eight XCDs work
    // on neighbouring row blocks -- i.e. on the SAME expert -- at any time and that expert's weights stay in the Infinity Cache;
    // contiguous per-XCD ranges would keep all experts' weights (1.9 GB at the 8x7B shape) live at once
    const int li = bid >> 3;
    grp = (li / group_sz) * 8 + xcd;
    in_grp = li - (li / group_sz) * group_sz;
  } else {
    const int wg = remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3) : bid;
    grp = wg / group_sz;
    in_grp = wg - grp * group_sz;
  }
  const int first_m = grp * GM;
  if (first_m >= tiles_m) return;                             // (remap == 2: surplus workgroups of the rounded-up grid)
  const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  if (in_grp >= gm * tiles_n) return;
  const int tm = first_m + in_grp % gm, tn = in_grp / gm;
  int64_t m0 = (int64_t)tm * BM, M = M_all;
  const uint16_t* W = W_all;
  if (groups.counts != nullptr) {
    int t = tm, g = 0;
    int64_t off = 0;
    for (; g < groups.n_groups; ++g) {
      const int c = groups.counts[g], nt = (c + BM - 1) / BM;
      if (t < nt) break;
      t -= nt; off += c;
    }
    if (g == groups.n_groups) return;                             // surplus tile of the upper-bound grid
    m0 = off + (int64_t)t * BM;
    M = off + groups.counts[g];                                    // row limit of this group
    W = W_all + (int64_t)g * groups.w_stride;
  }
  const int n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // ---- staging addresses: wave `wid` fills chunks wid*4..wid*4+3 (8 rows x 128 B each) of A and of W
  const int srow = lane >> 3;                                  // row inside the 8-row chunk
  const uint16_t* a_src[4];
  const uint16_t* w_src[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = (wid * 4 + c) * 8 + srow;                    // tile row 0..255
    const int slot = (lane & 7) ^ ((r >> 1) & 7);              // logical 16-B slot held by this physical slot
    int64_t gm_row = m0 + r; if (gm_row > M - 1) gm_row = M - 1;
    if (groups.a_rows != nullptr) gm_row = groups.a_rows[gm_row];
    int gn_row = n0 + r; if (gn_row > N - 1) gn_row = N - 1;
    if (ABL == 14) { gm_row &= 255; gn_row &= 255; }           // timing experiment: L2-resident operands (256 rows x 512 k each)
    a_src[c] = A + gm_row * lda + slot * 8;
    w_src[c] = W + (int64_t)gn_row * ldw + slot * 8;
  }

  auto stage = [&](int buf, int kt) {
    if (ABL == 14) kt &= 7;
    char* base = smem + buf * STAGE_BYTES + wid * 4096;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[c] + (int64_t)kt * BK), (lptr_t)(base + c * 1024), 16, 0, 0);
      if (ABL != 10 || kt == 0)
        __builtin_amdgcn_global_load_lds((gptr_t)(w_src[c] + (int64_t)kt * BK), (lptr_t)(base + A_BYTES + c * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside a stage); swizzle term is lane-constant because every
  //      fragment starts at a multiple of 16 rows: (row>>1)&7 == (lane>>1)&7
  const int frow = lane & 15, kq = lane >> 4, swz = (lane >> 1) & 7;
  const int a_off = (wr * 128 + frow) * 128;            // + i*2048
  // first tile row of W fragment j of this wave (= first output column of the fragment inside the tile)
  auto wrow = [&](int j) { return ROPE ? (wc >> 1) * 128 + (j >> 1) * 64 + (wc & 1) * 32 + (j & 1) * 16 : wc * 64 + j * 16; };
  const int w_off = A_BYTES + frow * 128;               // + wrow(j)*128
  const int s_off0 = ((kq) ^ swz) << 4, s_off1 = ((4 + kq) ^ swz) << 4;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  stage(0, 0);
  lds_dma_barrier();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (ABL == 0 || ABL == 8 || ABL == 9 || ABL == 12 || ABL == 13 || ABL == 14) stage(cur ^ 1, kt + 1 < nk ? kt + 1 : kt);   // branch-free body (re-stages the last tile once, harmless)
    else if (kt + 1 < nk && ABL != 1) stage(cur ^ 1, kt + 1);
    const char* sb = smem + (ABL == 2 ? 0 : cur * STAGE_BYTES);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ks ? s_off1 : s_off0;
      bf16x8_t wf[4], xf[8];
      if (ABL != 2 || kt == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(sb + w_off + wrow(j) * 128 + so);
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] = *reinterpret_cast<const bf16x8_t*>(sb + a_off + i * 2048 + so);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = __builtin_bit_cast(bf16x8_t, make_uint4(kt, j, ks, lane));
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] = __builtin_bit_cast(bf16x8_t, make_uint4(kt, i, ks, lane));
      }
      if (ABL == 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(xf[i]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(wf[j]));
      } else {
        if (ABL == 5) __builtin_amdgcn_s_setprio(1);
        if (ABL == 6) __builtin_amdgcn_iglp_opt(0);
        if (ABL == 7) __builtin_amdgcn_iglp_opt(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        if (ABL == 5) __builtin_amdgcn_s_setprio(0);
      }
    }
    if (ABL == 8) {
      // explicit issue order for the K-tile body (LLVM sched_group_barrier: 0x10 VMEM, 0x100 DS read, 0x8 MFMA): every pair of
      // activation fragments is read one 8-MFMA group AHEAD of the group that consumes it, so the lgkmcnt waits are counted
      // instead of lgkmcnt(0) right behind the read
      __builtin_amdgcn_sched_group_barrier(0x10, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // w0-3, x0-3 (ks 0)
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G0: x0,x1
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // x4,x5
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G1: x2,x3
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // x6,x7
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G2: x4,x5
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // ks 1: w0-3, x0,x1
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G3: x6,x7
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G4
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G5
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G6
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G7
    }
    if (ABL == 12) {  // LDS-DMA spread: one per 8-MFMA group
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 10, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 6, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
    }
    if (ABL == 13) {  // LDS-DMA spread over the first half: one per 4 MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0); __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
    }
    if (ABL == 0 || ABL == 9 || ABL == 14) {  // default: fragment reads issued two MFMA groups ahead of their consumers
      __builtin_amdgcn_sched_group_barrier(0x10, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);  // w0-3, x0-5 (ks 0)
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G0
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // x6,x7
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G1
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);   // ks 1: w0-3, x0,x1
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G2
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G3
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G4
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G5
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G6
      __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);     // G7
    }
    lds_dma_barrier();
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = frag row base + (lane&15), n = frag col base + (lane>>4)*4
  const int64_t mrow = m0 + wr * 128 + frow;
  const int ncol = n0 + wc * 64 + kq * 4;
  // RESIDUAL: all 16 residual loads of the lane are issued up front (the fragment registers are dead by now), so the epilogue pays one
  // memory latency instead of one per output row
  if constexpr (ROPE) {
    if (n0 + (wc >> 1) * 128 < rope.rope_cols) {        // this wave's head is a q or k head (uniform per wave)
      const int m0_mod = (int)(m0 % rope.S);              // block-uniform: the only 64-bit division
#pragma unroll
      for (int i = 0; i < 8; ++i) {
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
            const uint32_t xr = pack2bf_hw(acc[i][p2][r], acc[i][p2 + 2][r]);     // q/k = bf16(linear) first (:655-657)
            const float x1 = bflo(xr), x2 = bfhi(xr);
            acc[i][p2][r] = x1 * cc[r] - x2 * ss[r];
            acc[i][p2 + 2][r] = x2 * cc[r] + x1 * ss[r];
          }
        }
      }
    }
  }
  uint4 rpre[8][2];
  if constexpr (EPI == GRIT_EPI_RESIDUAL) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jq = 0; jq < 2; ++jq) {
        const int64_t m = mrow + i * 16;
        const int n = n0 + wrow(2 * jq + (kq & 1)) + (kq >> 1) * 8;
        rpre[i][jq] = (m < M && n < N) ? *reinterpret_cast<const uint4*>(Rsd + m * ldr + n) : make_uint4(0, 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = mrow + i * 16;
    if (m >= M) continue;
    if constexpr (EPI == GRIT_EPI_SWIGLU) {
      // fragments (0,1) and (2,3) are (gate, up) pairs -> two 16-column output blocks; the same permlane16 exchange gives every
      // lane 8 consecutive output columns
      const int nb = n0 + wc * 64;             // multiple of 64
      if (nb < N) {
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // bf16 roundings through v_cvt_pk_bf16_f32 (RNE, two per instruction) instead of the 5-op bit trick
          const uint32_t gu0 = pack2bf_hw(acc[i][0][r], acc[i][1][r]), gu1 = pack2bf_hw(acc[i][2][r], acc[i][3][r]);
          const uint32_t ss = pack2bf_hw(silu_f(bflo(gu0)), silu_f(bflo(gu1)));
          o0[r] = bflo(ss) * bfhi(gu0);
          o1[r] = bfhi(ss) * bfhi(gu1);
          const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0[r]), __float_as_uint(o1[r]), false, false);
          o0[r] = __uint_as_float(sw[0]); o1[r] = __uint_as_float(sw[1]);
        }
        const int oc = (nb >> 1) + (kq & 1) * 16 + (kq >> 1) * 8;
        if (2 * oc < N)
          *reinterpret_cast<uint4*>(C + m * ldc + oc) = make_uint4(pack2bf_hw(o0[0], o0[1]), pack2bf_hw(o0[2], o0[3]), pack2bf_hw(o1[0], o1[1]),
                                                                  pack2bf_hw(o1[2], o1[3]));
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
            const uint32_t rr = pack2bf_hw(v[2 * e], v[2 * e + 1]);
            v[2 * e] = bflo(rr) + bflo(ra[e]); v[2 * e + 1] = bfhi(rr) + bfhi(ra[e]);
          }
        }
        const uint4 pk = make_uint4(pack2bf_hw(v[0], v[1]), pack2bf_hw(v[2], v[3]), pack2bf_hw(v[4], v[5]), pack2bf_hw(v[6], v[7]));
        if (ABL == 11) { asm volatile("" ::"v"(pk.x), "v"(pk.y), "v"(pk.z), "v"(pk.w)); continue; }   // timing experiment: no stores
        *reinterpret_cast<uint4*>(C + m * ldc + n) = pk;
      }
    }
  }
}

constexpr int KH = 32;                          // k per ring slot (variant 8)
constexpr int SLOT_BYTES = (BM + BN) * KH * 2;  // 32 KiB
constexpr int SLOT_A = BM * KH * 2;             // 16 KiB

// ======================================================================================================
// Variant 8: K-half ring (4 x 32 KiB slots) + tile-level REGISTER double buffering + explicit issue order.
//   iteration h:  LDS-DMA of K-half h+4 into the slot whose fragments already sit in registers (slot h&3),
//                 12 ds_read_b128 of K-half h+1 interleaved (sched_group_barrier) with the 32 MFMAs (16x16x32) of K-half h,
//                 counted vmcnt(8) (K-half h+2 landed; h+3, h+4 stay in flight), ONE barrier.
// DMA issue -> first use is 3 iterations (~1.5 us of MFMAs), fragment reads never wait behind the barrier, and the
// body is branch-free (tail indices are clamped: the last K-half is re-staged into slots nobody reads).
// ======================================================================================================
template <int EPI>
__global__ void __launch_bounds__(512) gemm_bf16_nt_v8_k(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* C,
                                                         const uint16_t* Rsd, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                                                         int64_t ldc, int64_t ldr, int tiles_m, int tiles_n, int GM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int group_sz = GM * tiles_n;
  const int grp = wg / group_sz, first_m = grp * GM;
  const int gm = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int in_grp = wg - grp * group_sz;
  const int tm = first_m + in_grp % gm, tn = in_grp / gm;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // DMA roles: 16-row chunks 2*wid, 2*wid+1 of the A half and of the W half; 64-B rows, slot ^= (-(row>>2))&3
  const uint16_t* a_src[2];
  const uint16_t* w_src[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int r = (wid * 2 + c) * 16 + (lane >> 2);
    const int slot = (lane & 3) ^ ((0 - (r >> 2)) & 3);   // f(row) = (-(row>>2)) & 3: conflict-free for the 16x16x32 lane groups
    int64_t gm_row = m0 + r; if (gm_row > M - 1) gm_row = M - 1;
    int gn_row = n0 + r; if (gn_row > N - 1) gn_row = N - 1;
    a_src[c] = A + gm_row * lda + slot * 8;
    w_src[c] = W + (int64_t)gn_row * ldw + slot * 8;
  }
  const int nh = K / KH;
  auto stage = [&](int h) {   // h may exceed nh-1 near the tail: clamp the SOURCE, keep the slot
    const int hs = h < nh ? h : nh - 1;
    char* base = smem + (h & 3) * SLOT_BYTES + wid * 2048;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[c] + (int64_t)hs * KH), (lptr_t)(base + c * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[c] + (int64_t)hs * KH), (lptr_t)(base + SLOT_A + c * 1024), 16, 0, 0);
    }
  };
  // fragments (16x16x32): row = 16f + (lane&15); the 32-k slice of a row is 4 slots; lane reads slot kq ^ ((row>>2)&3)
  const int frow = lane & 15, kq = lane >> 4, swz = (0 - (frow >> 2)) & 3;
  const int x_off = (wr * 128 + frow) * 64 + ((kq ^ swz) << 4);          // + i*1024 (16 rows x 64 B)
  const int w_off = SLOT_A + (wc * 64 + frow) * 64 + ((kq ^ swz) << 4);  // + j*1024

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  bf16x8_t wa[4], xa[8], wb[4], xb[8];
#define V8_READ(WF, XF, SLOT)                                                                        \
  do {                                                                                               \
    const char* sb_ = smem + (SLOT) * SLOT_BYTES;                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) WF[j] = *reinterpret_cast<const bf16x8_t*>(sb_ + w_off + j * 1024); \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) XF[i] = *reinterpret_cast<const bf16x8_t*>(sb_ + x_off + i * 1024); \
  } while (0)
#define V8_MMA(WF, XF)                                                                               \
  do {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[j], XF[i], acc[i][j], 0, 0, 0);       \
  } while (0)
#define V8_SCHED()                                                                                   \
  do {                                                                                               \
    __builtin_amdgcn_sched_group_barrier(0x10, 4, 0);                                                \
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x8, 6, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x8, 6, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x8, 6, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x8, 6, 0);                                                 \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);                                                 \
  } while (0)
#define V8_SYNC()                                                                                    \
  do {                                                                                               \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_s_barrier();                                                                    \
    asm volatile("" ::: "memory");                                                                   \
  } while (0)

  stage(0); stage(1); stage(2); stage(3);
  V8_SYNC();                               // K-halves 0 and 1 have landed, 2 and 3 in flight
  V8_READ(wa, xa, 0);
  for (int h = 0; h < nh; h += 2) {
    // even half-step: compute K-half h (registers A), fetch K-half h+1 into registers B, refill slot h&3 with K-half h+4
    stage(h + 4);
    V8_READ(wb, xb, (h + 1) & 3);
    V8_MMA(wa, xa);
    V8_SCHED();
    V8_SYNC();
    // odd half-step
    stage(h + 5);
    V8_READ(wa, xa, (h + 2) & 3);
    V8_MMA(wb, xb);
    V8_SCHED();
    V8_SYNC();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped tail DMAs before the LDS is released
#undef V8_READ
#undef V8_MMA
#undef V8_SCHED
#undef V8_SYNC

  const int64_t mrow = m0 + wr * 128 + frow;
  const int ncol = n0 + wc * 64 + kq * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = mrow + i * 16;
    if (m >= M) continue;
    if constexpr (EPI == GRIT_EPI_SWIGLU) {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int nb = n0 + wc * 64 + j * 16;
        if (nb >= N) continue;
        const f32x4_t g = acc[i][j], u = acc[i][j + 1];
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = round_bf(silu_f(round_bf(g[r]))) * round_bf(u[r]);
        *reinterpret_cast<uint2*>(C + m * ldc + (nb >> 1) + kq * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = ncol + j * 16;
        if (n >= N) continue;
        f32x4_t v = acc[i][j];
        if constexpr (EPI == GRIT_EPI_RESIDUAL) {
          const uint2 rv = *reinterpret_cast<const uint2*>(Rsd + m * ldr + n);
          v[0] = round_bf(v[0]) + bflo(rv.x); v[1] = round_bf(v[1]) + bfhi(rv.x);
          v[2] = round_bf(v[2]) + bflo(rv.y); v[3] = round_bf(v[3]) + bfhi(rv.y);
        }
        *reinterpret_cast<uint2*>(C + m * ldc + n) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
    }
  }
}

// kernel generation: 1 = two 64 KiB stages, one barrier per K-tile (default); 8 = K-half ring + register double buffering
// (GRIT_GEMM_VARIANT=8: ties variant 1).  Removed after measurement (DESIGN.md "GEMM experiments"): 32x32x16 variants,
// burst-read / ring variants, one-wave-per-SIMD, weights-direct-to-registers.
// GRIT_GEMM_ABLATE=<n> (variant 1, STORE epilogue; timing experiments, results WRONG for 1,2,3,10):
//   1 no LDS-DMA in the K loop, 2 no ds_read, 3 no MFMA, 10 weight half of the DMA skipped; 5 setprio, 6/7 iglp_opt(0/1),
//   8 explicit issue order one group ahead, 0/9 = default (reads two MFMA groups ahead), 11 no epilogue stores,
//   12/13 LDS-DMA issue spread over the MFMAs (1 per 8 / 1 per 4), 14 operands made L2-resident (rows & 255, k-tiles & 7).
// GRIT_GEMM_GM=<n> m-tiles per scheduling group (default 4: 4 m x 8 n tiles in flight per XCD; measured 2/4/8/16/32 ->
// 5.41/5.33/5.47/6.03/6.66 ms on the QKV shape, no remap 5.65 ms), GRIT_GEMM_NOREMAP=1 disables the XCD remap (A/B only).
static int gemm_variant() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("GRIT_GEMM_VARIANT");
    v = (e != nullptr && e[0] == '8') ? 8 : 1;
  }
  return v;
}

template <int EPI>
static int launch_gemm(const void* A, const void* W, void* C, const void* R, int64_t M, int N, int K, int64_t lda, int64_t ldw,
                       int64_t ldc, int64_t ldr, hipStream_t st, GemmGroups grp = GemmGroups{nullptr, nullptr, 0, 0},
                       GemmRope rope = GemmRope{nullptr, nullptr, nullptr, 0, 0}) {
  const int tiles_m = grp.counts ? (int)(M / BM) + grp.n_groups : (int)((M + BM - 1) / BM), tiles_n = (N + BN - 1) / BN;
  static bool attr_set = false;  // idempotent; benign race
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    attr_set = true;
  }
  static int abl = -1;
  if (abl < 0) { const char* e = getenv("GRIT_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
  static int gm_knob = 0, remap_knob = 1;
  if (gm_knob == 0) {
    const char* e = getenv("GRIT_GEMM_GM"); gm_knob = (e && atoi(e) > 0) ? atoi(e) : 4;
    remap_knob = getenv("GRIT_GEMM_NOREMAP") ? 0 : 1;
  }
  // grouped launches: round-robin tile groups over the XCDs (remap 2) on a grid rounded up to 8 x whole groups
  const int total_groups = (tiles_m + gm_knob - 1) / gm_knob;
  static int rr_all = -1;
  if (rr_all < 0) rr_all = getenv("GRIT_GEMM_RR") ? 1 : 0;     // experiment: round-robin groups for the dense launches too
  const bool rr = (grp.counts || rr_all) && remap_knob;
  const unsigned nblocks = rr ? (unsigned)(8 * ((total_groups + 7) / 8) * gm_knob * tiles_n) : (unsigned)(tiles_m * tiles_n);
  const int remap_mode = rr ? 2 : remap_knob;
#define GRIT_LAUNCH_ABL(A_)                                                                                                          \
  hipLaunchKernelGGL((gemm_bf16_nt_k<EPI, A_>), dim3(nblocks), dim3(512), 2 * STAGE_BYTES, st, (const uint16_t*)A, \
                     (const uint16_t*)W, (uint16_t*)C, (const uint16_t*)R, M, N, K, lda, ldw, ldc, ldr, tiles_m, tiles_n, gm_knob, remap_mode, grp, rope)
  if (abl > 0 && EPI == GRIT_EPI_STORE && gemm_variant() == 1) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 11>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_k<EPI, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    if (abl == 1) GRIT_LAUNCH_ABL(1); else if (abl == 2) GRIT_LAUNCH_ABL(2); else if (abl == 3) GRIT_LAUNCH_ABL(3);
    else if (abl == 5) GRIT_LAUNCH_ABL(5); else if (abl == 6) GRIT_LAUNCH_ABL(6); else if (abl == 7) GRIT_LAUNCH_ABL(7); else if (abl == 8) GRIT_LAUNCH_ABL(8); else if (abl == 9) GRIT_LAUNCH_ABL(9); else if (abl == 10) GRIT_LAUNCH_ABL(10); else if (abl == 11) GRIT_LAUNCH_ABL(11); else if (abl == 12) GRIT_LAUNCH_ABL(12); else if (abl == 14) GRIT_LAUNCH_ABL(14); else GRIT_LAUNCH_ABL(13);
  } else if (gemm_variant() == 8 && grp.counts == nullptr && EPI != GRIT_EPI_ROPE) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_v8_k<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * SLOT_BYTES);
    hipLaunchKernelGGL(gemm_bf16_nt_v8_k<EPI>, dim3((unsigned)(tiles_m * tiles_n)), dim3(512), 4 * SLOT_BYTES, st, (const uint16_t*)A,
                       (const uint16_t*)W, (uint16_t*)C, (const uint16_t*)R, M, N, K, lda, ldw, ldc, ldr, tiles_m, tiles_n, gm_knob);
  } else
    GRIT_LAUNCH_ABL(0);
  GRIT_CHECK_LAUNCH("grit_gemm_bf16_nt");
  return GRIT_OK;
}

}  // namespace grit

using namespace grit;

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
    case GRIT_EPI_SWIGLU:
      GRIT_REQUIRE(N % 64 == 0 && ldc >= N / 2, GRIT_E_UNSUPPORTED, "grit_gemm_bf16_nt: SWIGLU epilogue needs N %% 64 == 0 and ldc >= N/2");
      return launch_gemm<GRIT_EPI_SWIGLU>(A, W, C, nullptr, M, N, K, lda, ldw, ldc, 0, st);
    default:
      GRIT_REQUIRE(false, GRIT_E_BADARG, "grit_gemm_bf16_nt: unknown epilogue %d", epilogue);
  }
  return GRIT_OK;
}
