// In-batch InfoNCE: temperature-scaled similarity matrix + cross-entropy, forward and backward.
//
// Replaces DistributedContrastiveLoss.__call__/compute_similarity (gritlm/training/model.py:36-47, :62-64)
// after the cross-rank gather.  The representations stay fp32 (the reference pools/normalises in fp32,
// gritlm/gritlm.py:212-214) and the products run on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32,
// bitwise an fmaf chain): at tau = 0.02 a cosine error of 2e-5 already moves a logit by 1e-3, so a
// bf16-input MFMA would miss the 1e-3 loss tolerance (SURVEY.md §7 hard part 3).
//
//   1. scores = (1/tau) q p^T                       strided f32 MFMA GEMM (gemm_f32.hip)
//   2. per row: lse, loss += (lse - s[i, i*G])/Nq;  scores <- (softmax - onehot) / (Nq tau)   (in place)
//   3. dq = dS[q_off.., :] p ,  dp = dS[:, p_off..]^T q   for the caller's own rows only -- the rows
//      that carry grad after `_dist_gather_tensor` re-inserts the local shard (:49-60).
#include "common.h"

namespace grit {

// exact-f32 MFMA GEMM with arbitrary operand strides: gemm_f32.hip
int launch_f32_gemm_strided(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk, int64_t sbn,
                            int64_t ldc, float alpha, hipStream_t st);

// one workgroup per query row: logsumexp, the row's loss term (loss_rows[i]; summed in a fixed order by infonce_loss_k -- no float
// atomics: the loss is bit-reproducible from run to run), then d loss / d raw-scores in place
__global__ void __launch_bounds__(256) infonce_ce_k(float* __restrict__ scores, float* __restrict__ loss_rows, int Nq, int Np, int group,
                                                    float inv_temperature, int want_grad) {
  __shared__ float red[8];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* row = scores + (int64_t)i * Np;
  float mx = -INFINITY;
  for (int j = tid; j < Np; j += 256) mx = fmaxf(mx, row[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  // sum of exp over the NON-target columns only: when the positive dominates (a trained model) both the
  // loss (log1p form) and d/ds_target = -(sum_others / sum) keep full relative precision, where
  // softmax - 1 (what torch computes) cancels catastrophically in fp32
  const int tgt = i * group;
  float sum = 0.f;
  for (int j = tid; j < Np; j += 256) sum += (j == tgt) ? 0.f : expf(row[j] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float sum_others = red[0] + red[1] + red[2] + red[3];
  const float s_t = row[tgt];
  const float e_t = expf(s_t - mx);
  const float tot = sum_others + e_t;
  const float li = (s_t == mx) ? log1pf(sum_others) : (mx - s_t) + logf(tot);
  if (tid == 0) loss_rows[i] = li;
  if (!want_grad) return;
  __syncthreads();  // row[tgt] read above before anyone overwrites it
  const float sc = inv_temperature / (float)Nq, inv_tot = 1.0f / tot;
  for (int j = tid; j < Np; j += 256) {
    const float pj = (j == tgt) ? -sum_others * inv_tot : expf(row[j] - mx) * inv_tot;
    row[j] = pj * sc;
  }
}

// loss[0] = mean(loss_rows[0 .. Nq)): one workgroup, fixed summation tree (thread t adds rows t, t+256, ... in order; then a fixed
// wave / cross-wave reduction)
__global__ void __launch_bounds__(256) infonce_loss_k(float* __restrict__ loss, const float* __restrict__ loss_rows, int Nq) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int i = tid; i < Nq; i += 256) acc += loss_rows[i];
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)Nq;
}

static int launch_f32_gemm(const float* A, const float* B, float* C, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbk,
                           int64_t sbn, int64_t ldc, float alpha, hipStream_t st) {
  return launch_f32_gemm_strided(A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, alpha, st);
}

}  // namespace grit

using namespace grit;

// (ABI 2: the round-2 entry point `grit_infonce_fwd_bwd` took ONE loss pointer that silently grew from 1 to 1 + Nq floats; the per-row
// terms now have their own argument and the symbol a new name, so a caller built against the old header fails at load time.)
extern "C" int grit_infonce_rows_fwd_bwd(const float* q, const float* p, float inv_temperature, float* scores, float* loss, float* loss_rows,
                                         float* dq, float* dp, int Nq, int Np, int H, int q_off, int nq_loc, int p_off, int np_loc,
                                         void* stream) {
  GRIT_REQUIRE(q && p && scores && loss && loss_rows, GRIT_E_BADARG, "grit_infonce_rows_fwd_bwd: null pointer");
  GRIT_REQUIRE(Nq > 0 && Np > 0 && H > 0, GRIT_E_BADARG, "grit_infonce_rows_fwd_bwd: bad sizes");
  GRIT_REQUIRE(Np % Nq == 0, GRIT_E_BADARG, "grit_infonce_rows_fwd_bwd: Np=%d is not a multiple of Nq=%d (target = i * Np/Nq)", Np, Nq);
  GRIT_REQUIRE((dq == nullptr) || (q_off >= 0 && nq_loc > 0 && (int64_t)q_off + nq_loc <= Nq), GRIT_E_BADARG, "grit_infonce_rows_fwd_bwd: bad q range");
  GRIT_REQUIRE((dp == nullptr) || (p_off >= 0 && np_loc > 0 && (int64_t)p_off + np_loc <= Np), GRIT_E_BADARG, "grit_infonce_rows_fwd_bwd: bad p range");
  hipStream_t st = (hipStream_t)stream;
  // scores[i,j] = inv_t * sum_h q[i,h] p[j,h]
  int rc = launch_f32_gemm(q, p, scores, Nq, Np, H, H, 1, 1, H, Np, inv_temperature, st);
  if (rc) return rc;
  const int want_grad = (dq != nullptr) || (dp != nullptr);
  hipLaunchKernelGGL(infonce_ce_k, dim3(Nq), dim3(256), 0, st, scores, loss_rows, Nq, Np, Np / Nq, inv_temperature, want_grad);
  GRIT_CHECK_LAUNCH("grit_infonce_rows_fwd_bwd: ce");
  hipLaunchKernelGGL(infonce_loss_k, dim3(1), dim3(256), 0, st, loss, loss_rows, Nq);
  GRIT_CHECK_LAUNCH("grit_infonce_rows_fwd_bwd: loss");
  if (dq) {  // dq[m,h] = sum_j dS[q_off+m, j] p[j,h]
    rc = launch_f32_gemm(scores + (int64_t)q_off * Np, p, dq, nq_loc, H, Np, Np, 1, H, 1, H, 1.0f, st);
    if (rc) return rc;
  }
  if (dp) {  // dp[m,h] = sum_i dS[i, p_off+m] q[i,h]
    rc = launch_f32_gemm(scores + p_off, q, dp, np_loc, H, Nq, 1, Np, H, 1, H, 1.0f, st);
    if (rc) return rc;
  }
  return GRIT_OK;
}
