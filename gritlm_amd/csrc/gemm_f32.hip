// Exact-f32 strided GEMM on the f32 matrix pipe (v_mfma_f32_32x32x2_f32 -- bitwise an fmaf chain, 157 TF peak = 1/16 of the bf16 rate).
//
//   C[m,n] = alpha * sum_k A(m,k) B(k,n),   A(m,k) = A[m*sam + k*sak],   B(k,n) = B[k*sbk + n*sbn]
//
// Users: the InfoNCE similarity matrix  scores = (1/tau) q p^T  and its two gradient products (gritlm/training/model.py:42-47, :62-64),
// and the index search  scores = queries @ embeddings  (rag/index.py:97-104).  The representations stay fp32 on purpose: at tau = 0.02
// a cosine error of 2e-5 moves a logit by 1e-3 (SURVEY §7 hard part 3), so no bf16-input MFMA here.
//
// Round 3 rewrite (the round-2 kernel staged 16-deep K slices with scalar 4-byte loads behind two barriers: ~5 % of the pipe).
//   * tile BM x BN x 32 per 256-thread workgroup, 4 waves, every wave a (BM/WM) x (BN/WN) block of 32x32 MFMA tiles; three shapes:
//     128x128 (big products), 64x64 (few tiles: the local-row gradient products), 32x128 (a handful of query rows against a long index:
//     HBM-bound, no padded MFMA work);
//   * operands come from HBM as 16-byte loads into registers one K-step AHEAD of the MFMAs that consume them (the loads of step t+1
//     are in flight during the 16 x TM x TN MFMAs of step t), go to LDS with ds_write_b128, one LDS image, two barriers per step; 2-3
//     workgroups per CU cover each other's barriers (37 KB of LDS, ~130 VGPRs);
//   * LDS images are chosen per operand by which index is contiguous in memory, so the HBM -> LDS copy is always a straight 16-byte copy
//     (no transposition pass):
//       - k-contiguous (q, p, dS rows):   [row][36]  (32 k + 4 pad floats).  A lane fetches FOUR consecutive k of its row with one
//         ds_read_b128 and feeds them to four MFMAs: lanes 0-31 hold k = 8s+j, lanes 32-63 hold k = 8s+4+j for MFMA j -- a permutation of
//         the summation index, applied to both operands alike (the sum is over k: any order is a valid dot product; the order is FIXED,
//         so results are bit-reproducible).  36-float rows make the 16-lane groups of ds_read_b128 hit 64 distinct banks.
//       - row-contiguous (p as B of dq; dS^T and q of dp; an [H,N] index): [k][rows], one ds_read_b32 per MFMA operand at the same
//         permuted k (consecutive lanes = consecutive banks).
//   * operands that are not 16-byte loadable (odd leading dimensions / offsets, ragged K) take a guarded scalar-load path into the same
//     images;
//   * workgroup ids are dealt to the 8 XCDs in contiguous ranges, m-tiles fastest inside groups of 16, so the tiles an L2 serves at one
//     time share their B panel.
#include "common.h"

namespace grit {

constexpr int FKT = 32;          // K-step
constexpr int FLD = FKT + 4;     // row pitch (floats) of a k-contiguous image

template <int R, bool KFAST>
struct F32Stage {
  static constexpr int NV = R * FKT / 4 / 256;      // float4 per thread per K-step (R = 32: 1, 64: 2, 128: 4)
  static_assert(NV >= 1, "tile too small for 256 threads");
  static constexpr int LDS_FLOATS = KFAST ? R * FLD : FKT * R;

  // element (row, k) of the operand lives at X[row * s_row + k * s_k]; rows [row0, row0 + R), k in [k0, k0 + 32)
  static __device__ __forceinline__ void load(f32x4_t (&v)[NV], const float* __restrict__ X, int64_t s_row, int64_t s_k, int row0, int k0,
                                              int rows, int K, bool vec, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + 256 * i;
      int row, k;
      if (KFAST) { row = row0 + (f >> 3); k = k0 + 4 * (f & 7); }
      else       { k = k0 + f / (R / 4); row = row0 + 4 * (f % (R / 4)); }
      f32x4_t x = {0.f, 0.f, 0.f, 0.f};
      if (vec) {      // the 4 elements are contiguous, 16-byte aligned and in range together (host-checked: extent % 4 == 0)
        if (row < rows && k < K) x = *reinterpret_cast<const f32x4_t*>(X + (int64_t)row * s_row + (int64_t)k * s_k);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rr = KFAST ? row : row + j, kk = KFAST ? k + j : k;
          if (rr < rows && kk < K) x[j] = X[(int64_t)rr * s_row + (int64_t)kk * s_k];
        }
      }
      v[i] = x;
    }
  }

  static __device__ __forceinline__ void store(const f32x4_t (&v)[NV], float* __restrict__ S, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + 256 * i;
      if (KFAST) *reinterpret_cast<f32x4_t*>(S + (f >> 3) * FLD + 4 * (f & 7)) = v[i];
      else       *reinterpret_cast<f32x4_t*>(S + (f / (R / 4)) * R + 4 * (f % (R / 4))) = v[i];
    }
  }

  // the four operand values of MFMAs j = 0..3 of k-group s for the 32 rows starting at r0: value j belongs to k = 8s + 4*half + j
  static __device__ __forceinline__ f32x4_t frag(const float* __restrict__ S, int r0, int s, int lane) {
    const int r = r0 + (lane & 31), kb = 8 * s + 4 * (lane >> 5);
    if (KFAST) return *reinterpret_cast<const f32x4_t*>(S + r * FLD + kb);
    f32x4_t x;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = S[(kb + j) * R + r];
    return x;
  }
};

template <int BM, int BN, int WM, int WN, bool AK, bool BK>
__global__ void __launch_bounds__(256, 2) gemm_f32_k(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C, int M, int N,
                                                     int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t ldc, float alpha,
                                                     int vec_a, int vec_b, int tiles_m, int tiles_n) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using SA = F32Stage<BM, AK>;
  using SB = F32Stage<BN, BK>;
  __shared__ __attribute__((aligned(16))) float As[SA::LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[SB::LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // workgroup -> tile: contiguous id range per XCD (bijective for any grid size), m fastest inside groups of 16 m-tiles
  const int nwg = tiles_m * tiles_n;
  int id;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  constexpr int GM = 16;
  const int per_group = GM * tiles_n, g = id / per_group, first_m = g * GM;
  const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  const int tm_ = first_m + (id % per_group) % gsz, tn_ = (id % per_group) / gsz;
  const int m0 = tm_ * BM, n0 = tn_ * BN;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4_t va[SA::NV], vb[SB::NV];
  SA::load(va, A, sam, sak, m0, 0, M, K, vec_a != 0, tid);
  SB::load(vb, Bm, sbn, sbk, n0, 0, N, K, vec_b != 0, tid);
  for (int k0 = 0; k0 < K; k0 += FKT) {
    __syncthreads();                       // every wave is done reading the previous K-step's image
    SA::store(va, As, tid);
    SB::store(vb, Bs, tid);
    __syncthreads();
    if (k0 + FKT < K) {                    // next K-step's operands: in flight during this step's MFMAs
      SA::load(va, A, sam, sak, m0, k0 + FKT, M, K, vec_a != 0, tid);
      SB::load(vb, Bm, sbn, sbk, n0, k0 + FKT, N, K, vec_b != 0, tid);
    }
#pragma unroll
    for (int s = 0; s < FKT / 8; ++s) {
      f32x4_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = SA::frag(As, wm * (BM / WM) + i * 32, s, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = SB::frag(Bs, wn * (BN / WN) + j * 32, s, lane);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][jj], b[j][jj], acc[i][j], 0, 0, 0);
    }
  }
  // D[row][col]: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n < N) C[(int64_t)m * ldc + n] = alpha * acc[i][j][r];
      }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_shape(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                        int64_t ldc, float alpha, bool ak, bool bk, int vec_a, int vec_b, hipStream_t st) {
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  const dim3 grid((unsigned)(tm * tn)), block(256);
#define GRIT_F32_LAUNCH(AKv, BKv)                                                                                                  \
  hipLaunchKernelGGL((gemm_f32_k<BM, BN, WM, WN, AKv, BKv>), grid, block, 0, st, A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, \
                     vec_a, vec_b, tm, tn)
  if (ak && bk) GRIT_F32_LAUNCH(true, true);
  else if (ak) GRIT_F32_LAUNCH(true, false);
  else if (bk) GRIT_F32_LAUNCH(false, true);
  else GRIT_F32_LAUNCH(false, false);
#undef GRIT_F32_LAUNCH
  GRIT_CHECK_LAUNCH("f32 gemm");
  return GRIT_OK;
}

// shared by infonce.hip and knn.hip
int launch_f32_gemm_strided(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                            int64_t ldc, float alpha, hipStream_t st) {
  if (M <= 0 || N <= 0) return GRIT_OK;
  GRIT_REQUIRE(K > 0 && M <= (1 << 30) && N <= (1 << 30) && K <= (1 << 30), GRIT_E_BADARG, "f32 gemm: bad sizes M=%d N=%d K=%d", M, N, K);
  GRIT_REQUIRE((int64_t)((M + 31) / 32) * ((N + 63) / 64) < (1ll << 31), GRIT_E_UNSUPPORTED, "f32 gemm: grid too large (M=%d N=%d)", M, N);
  // LDS image by the contiguous index; an operand with no unit stride is staged k-major through the scalar path
  const bool ak = (sak == 1) || (sam != 1), bk = (sbk == 1) || (sbn != 1);
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  const int vec_a = al16(A) && (ak ? (sak == 1 && sam % 4 == 0 && K % 4 == 0) : (sam == 1 && sak % 4 == 0 && M % 4 == 0));
  const int vec_b = al16(B) && (bk ? (sbk == 1 && sbn % 4 == 0 && K % 4 == 0) : (sbn == 1 && sbk % 4 == 0 && N % 4 == 0));
  const int64_t big = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  if (M <= 32) return launch_shape<32, 128, 1, 4>(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, ak, bk, vec_a, vec_b, st);
  if (big >= 256) return launch_shape<128, 128, 2, 2>(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, ak, bk, vec_a, vec_b, st);
  return launch_shape<64, 64, 2, 2>(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, ak, bk, vec_a, vec_b, st);
}

}  // namespace grit
