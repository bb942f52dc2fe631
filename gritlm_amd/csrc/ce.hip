// Next-token cross entropy over the vocabulary for the generative branch of unified training:
// NextTokenLoss (gritlm/training/model.py:66-107) on logits = lm_head(hidden).float() (modeling_mistral_gritlm.py:1176-1177).
// One workgroup per token row; logits are bf16 (the lm_head GEMM output), all arithmetic fp32.
//   fwd: lse[t] = logsumexp(logits[t,:]),  loss_row[t] = lse[t] - logits[t, label[t]]   (0 for label == -100, torch's ignore_index)
//   bwd: logits[t,:] <- (exp(logits[t,:] - lse[t]) - onehot(label[t])) * scale * (*dev_scale)   in place, bf16 (0 for ignored rows)
// HBM-bound: fwd reads the row once (online max/sum), bwd reads and writes it once.
#include "common.h"

namespace grit {

constexpr int CE_THREADS = 256;

__global__ void __launch_bounds__(CE_THREADS) ce_fwd_k(const uint16_t* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                       float* __restrict__ lse, float* __restrict__ loss_row, int V) {
  __shared__ float red_m[CE_THREADS / 64], red_s[CE_THREADS / 64];
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* row = logits + t * ld;
  float m = -INFINITY, s = 0.f;
  const int VC = V >> 3;
  for (int c = tid; c < VC; c += CE_THREADS) {
    const uint4 v = reinterpret_cast<const uint4*>(row)[c];
    const float x[8] = {bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y), bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w)};
    float cm = x[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) cm = fmaxf(cm, x[e]);
    const float nm = fmaxf(m, cm);
    float cs = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) cs += __expf(x[e] - nm);
    s = s * __expf(m - nm) + cs;      // m = -inf on the first chunk: exp(-inf) = 0
    m = nm;
  }
  for (int i = (VC << 3) + tid; i < V; i += CE_THREADS) {   // tail when V is not a multiple of 8
    const float x = bf2f(row[i]);
    const float nm = fmaxf(m, x);
    s = s * __expf(m - nm) + __expf(x - nm);
    m = nm;
  }
  // combine (m, s) pairs: wave, then block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64), os = __shfl_xor(s, o, 64);
    const float nm = fmaxf(m, om);
    s = (nm == -INFINITY) ? 0.f : s * __expf(m - nm) + os * __expf(om - nm);
    m = nm;
  }
  if (lane == 0) { red_m[wave] = m; red_s[wave] = s; }
  __syncthreads();
  if (tid == 0) {
    float M = red_m[0], S = red_s[0];
#pragma unroll
    for (int w = 1; w < CE_THREADS / 64; ++w) {
      const float nm = fmaxf(M, red_m[w]);
      S = (nm == -INFINITY) ? 0.f : S * __expf(M - nm) + red_s[w] * __expf(red_m[w] - nm);
      M = nm;
    }
    const float l = M + logf(S);
    lse[t] = l;
    const int64_t lab = labels[t];
    loss_row[t] = (lab >= 0 && lab < V) ? l - bf2f(row[lab]) : 0.f;
  }
}

__global__ void __launch_bounds__(CE_THREADS) ce_bwd_k(uint16_t* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                       const float* __restrict__ lse, const float* __restrict__ dev_scale, float scale, int V) {
  const int64_t t = blockIdx.x;
  const int tid = threadIdx.x;
  uint16_t* row = logits + t * ld;
  const int64_t lab = labels[t];
  const bool live = lab >= 0 && lab < V;
  const float sc = live ? scale * (dev_scale ? *dev_scale : 1.f) : 0.f;
  const float l = lse[t];
  const int VC = V >> 3;
  for (int c = tid; c < VC; c += CE_THREADS) {
    uint4 v = reinterpret_cast<uint4*>(row)[c];
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i0 = c * 8 + 2 * e;
      const float g0 = (__expf(bflo(w[e]) - l) - (i0 == lab ? 1.f : 0.f)) * sc;
      const float g1 = (__expf(bfhi(w[e]) - l) - (i0 + 1 == lab ? 1.f : 0.f)) * sc;
      w[e] = pack2bf(g0, g1);
    }
    reinterpret_cast<uint4*>(row)[c] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  for (int i = (VC << 3) + tid; i < V; i += CE_THREADS)
    row[i] = (uint16_t)f2bf((__expf(bf2f(row[i]) - l) - (i == lab ? 1.f : 0.f)) * sc);
}

}  // namespace grit

using namespace grit;

extern "C" int grit_ce_fwd(const void* logits, int64_t ld, const int64_t* labels, float* lse, float* loss_row, int64_t T, int V, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(logits && labels && lse && loss_row, GRIT_E_BADARG, "grit_ce_fwd: null pointer");
  GRIT_REQUIRE(T > 0 && V > 0 && ld >= V && ld % 8 == 0, GRIT_E_BADARG, "grit_ce_fwd: bad sizes T=%lld V=%d ld=%lld", (long long)T, V, (long long)ld);
  GRIT_REQUIRE(aligned16(logits), GRIT_E_BADARG, "grit_ce_fwd: logits must be 16-byte aligned");
  GRIT_REQUIRE(T < (1ll << 31), GRIT_E_UNSUPPORTED, "grit_ce_fwd: too many rows");
  hipLaunchKernelGGL(ce_fwd_k, dim3((unsigned)T), dim3(CE_THREADS), 0, (hipStream_t)stream, (const uint16_t*)logits, ld, labels, lse, loss_row, V);
  GRIT_CHECK_LAUNCH("grit_ce_fwd");
  return GRIT_OK;
}

extern "C" int grit_ce_bwd(void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* dev_scale, float scale, int64_t T,
                           int V, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(logits && labels && lse, GRIT_E_BADARG, "grit_ce_bwd: null pointer");
  GRIT_REQUIRE(T > 0 && V > 0 && ld >= V && ld % 8 == 0, GRIT_E_BADARG, "grit_ce_bwd: bad sizes");
  GRIT_REQUIRE(aligned16(logits), GRIT_E_BADARG, "grit_ce_bwd: logits must be 16-byte aligned");
  GRIT_REQUIRE(T < (1ll << 31), GRIT_E_UNSUPPORTED, "grit_ce_bwd: too many rows");
  hipLaunchKernelGGL(ce_bwd_k, dim3((unsigned)T), dim3(CE_THREADS), 0, (hipStream_t)stream, (uint16_t*)logits, ld, labels, lse, dev_scale, scale, V);
  GRIT_CHECK_LAUNCH("grit_ce_bwd");
  return GRIT_OK;
}
