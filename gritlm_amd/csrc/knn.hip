// Brute-force k-nearest-neighbour search of the RAG index (rag/index.py:97-104: scores = queries @ embeddings; torch.topk(scores, k)):
// inner-product scores on the exact-f32 MFMA GEMM of infonce.hip, then a per-query top-k by chunked bitonic sorts in LDS
// (2048 candidates per workgroup -> its k best, repeated until one chunk is left).  Descending scores, lower index first on ties.
#include "common.h"

namespace grit {

int launch_f32_gemm_strided(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                            int64_t ldc, float alpha, hipStream_t st);   // infonce.hip

constexpr int TK_C = 2048;      // candidates per workgroup
constexpr int TK_T = 256;

__device__ __forceinline__ bool tk_before(float va, int64_t ia, float vb, int64_t ib) { return va > vb || (va == vb && ia < ib); }

// vals [Q, n] (row stride ldv); idx (nullable) [Q, n]: candidate ids (implicit: position).  Workgroup (c, q) sorts candidates
// [c*TK_C, (c+1)*TK_C) of row q and writes its best k to out_v/out_i [Q, nchunks, k].
__global__ void __launch_bounds__(TK_T) topk_chunk_k(const float* __restrict__ vals, const int64_t* __restrict__ idx, int64_t n, int64_t ldv,
                                                    int64_t ldi, int k, float* __restrict__ out_v, int64_t* __restrict__ out_i, int nchunks) {
  __shared__ float sv[TK_C];
  __shared__ int64_t si[TK_C];
  const int c = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
  const int64_t base = (int64_t)c * TK_C;
  for (int j = tid; j < TK_C; j += TK_T) {
    const int64_t p = base + j;
    const bool in = p < n;
    sv[j] = in ? vals[(int64_t)q * ldv + p] : -INFINITY;
    si[j] = in ? (idx ? idx[(int64_t)q * ldi + p] : p) : INT64_MAX;
  }
  __syncthreads();
  // bitonic sort, "tk_before" order first
  for (int size = 2; size <= TK_C; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < TK_C / 2; t += TK_T) {
        const int lo = 2 * t - (t & (stride - 1));     // index with bit `stride` clear
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);            // this sub-sequence sorts "best first"
        const float va = sv[lo], vb = sv[hi];
        const int64_t ia = si[lo], ib = si[hi];
        const bool swap = up ? tk_before(vb, ib, va, ia) : tk_before(va, ia, vb, ib);
        if (swap) { sv[lo] = vb; sv[hi] = va; si[lo] = ib; si[hi] = ia; }
      }
      __syncthreads();
    }
  }
  float* ov = out_v + ((int64_t)q * nchunks + c) * k;
  int64_t* oi = out_i + ((int64_t)q * nchunks + c) * k;
  for (int j = tid; j < k; j += TK_T) { ov[j] = sv[j]; oi[j] = si[j]; }
}

}  // namespace grit

using namespace grit;

extern "C" int64_t grit_knn_workspace_bytes(int Q, int64_t N, int k) {
  if (Q <= 0 || N <= 0 || k <= 0 || Q > 65535 || N >= (1ll << 31) || k > TK_C / 2) return 0;   // 0 for sizes grit_knn_topk rejects
  const int64_t c1 = (N + TK_C - 1) / TK_C;
  // scores [Q,N] fp32 + two candidate buffers (values fp32 + ids int64) of the first level's size
  return (int64_t)Q * N * 4 + 2 * (int64_t)Q * c1 * k * 12 + 256;
}

extern "C" int grit_knn_topk(const float* queries, const float* embeddings, int Q, int64_t N, int H, int64_t emb_stride_n, int64_t emb_stride_h,
                             int k, void* workspace, float* out_scores, int64_t* out_index, void* stream) {
  if (Q == 0) return GRIT_OK;
  GRIT_REQUIRE(queries && embeddings && workspace && out_scores && out_index, GRIT_E_BADARG, "grit_knn_topk: null pointer");
  GRIT_REQUIRE(Q > 0 && N > 0 && H > 0 && N < (1ll << 31), GRIT_E_BADARG, "grit_knn_topk: bad sizes Q=%d N=%lld H=%d", Q, (long long)N, H);
  GRIT_REQUIRE(k > 0 && k <= TK_C / 2 && k <= N, GRIT_E_UNSUPPORTED, "grit_knn_topk: k=%d (1..min(N, %d))", k, TK_C / 2);
  GRIT_REQUIRE(Q <= 65535, GRIT_E_UNSUPPORTED, "grit_knn_topk: Q=%d > 65535 (batch the queries)", Q);
  hipStream_t st = (hipStream_t)stream;
  float* scores = (float*)workspace;
  const int64_t c1 = (N + TK_C - 1) / TK_C;
  char* p = (char*)workspace + (((int64_t)Q * N * 4 + 15) / 16) * 16;
  float* cv[2];
  int64_t* ci[2];
  for (int s = 0; s < 2; ++s) {
    ci[s] = (int64_t*)p; p += (int64_t)Q * c1 * k * 8;
    cv[s] = (float*)p; p += (((int64_t)Q * c1 * k * 4 + 15) / 16) * 16;
  }
  // scores[q, n] = sum_h queries[q,h] * embeddings(n,h)
  int rc = launch_f32_gemm_strided(queries, embeddings, scores, Q, (int)N, H, H, 1, emb_stride_h, emb_stride_n, N, 1.0f, st);
  if (rc) return rc;
  const float* in_v = scores;
  const int64_t* in_i = nullptr;
  int64_t n = N, ldv = N, ldi = 0;
  int cur = 0;
  for (;;) {
    const int64_t nch = (n + TK_C - 1) / TK_C;
    const bool last = nch == 1;
    float* ov = last ? out_scores : cv[cur];
    int64_t* oi = last ? out_index : ci[cur];
    hipLaunchKernelGGL(topk_chunk_k, dim3((unsigned)nch, (unsigned)Q), dim3(TK_T), 0, st, in_v, in_i, n, ldv, ldi, k, ov, oi, (int)nch);
    GRIT_CHECK_LAUNCH("grit_knn_topk: top-k");
    if (last) break;
    in_v = ov; in_i = oi; n = nch * k; ldv = n; ldi = n;
    cur ^= 1;
  }
  return GRIT_OK;
}
