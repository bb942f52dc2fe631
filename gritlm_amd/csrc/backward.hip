// HBM-bound backward pieces of the contrastive training step (autograd of the ops in elementwise.hip /
// the MLP activation): RMSNorm backward (+ fused residual-gradient add), SwiGLU forward/backward on the
// concatenated [gate | up] layout used by the training engine, embedding scatter-add, small helpers.
#include <atomic>

#include "common.h"

namespace grit {

// ---------------------------------------------------------------- RMSNorm backward
// y = w * (x * rs),  rs = rsqrt(mean(x^2) + eps)   (scripts/modeling_mistral_gritlm.py:84-89)
//   dx = rs * (dy*w) - x * rs^3 * mean(dy*w*x)  (+ dres),   dw += sum_t dy * x * rs
// One wave per row (grid-stride); the block's dw contribution is accumulated in LDS and written to
// dw_partial[blockIdx]; rmsnorm_dw_reduce_k sums the partials.
__global__ void __launch_bounds__(256) rmsnorm_bwd_k(const uint4* __restrict__ dy, const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                     const uint4* dres, uint4* dx, float* __restrict__ dw_partial, int64_t T, int H,
                                                     float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* acc = reinterpret_cast<float*>(smem);  // [H]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int HC = H >> 3;
  for (int i = tid; i < H; i += 256) acc[i] = 0.f;
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < T; row += (int64_t)gridDim.x * 4) {
    const uint4* xr = x + row * HC;
    const uint4* gr = dy + row * HC;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < HC; c += 64) {
      const uint4 xv = xr[c], gv = gr[c], wv = w[c];
      const uint32_t xa[4] = {xv.x, xv.y, xv.z, xv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, wa[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = bflo(xa[e]), x1 = bfhi(xa[e]);
        ss += x0 * x0 + x1 * x1;
        dot += bflo(ga[e]) * bflo(wa[e]) * x0 + bfhi(ga[e]) * bfhi(wa[e]) * x1;
      }
    }
    ss = wave_sum(ss); dot = wave_sum(dot);
    const float rs = rsqrtf(ss / (float)H + eps);
    const float k2 = rs * rs * rs * dot / (float)H;
    for (int c = lane; c < HC; c += 64) {
      const uint4 xv = xr[c], gv = gr[c], wv = w[c];
      uint4 rv = make_uint4(0, 0, 0, 0);
      if (dres != nullptr) rv = dres[row * HC + c];
      const uint32_t xa[4] = {xv.x, xv.y, xv.z, xv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, wa[4] = {wv.x, wv.y, wv.z, wv.w},
                     ra[4] = {rv.x, rv.y, rv.z, rv.w};
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = bflo(xa[e]), x1 = bfhi(xa[e]), g0 = bflo(ga[e]), g1 = bfhi(ga[e]);
        const float d0 = rs * g0 * bflo(wa[e]) - x0 * k2 + bflo(ra[e]);
        const float d1 = rs * g1 * bfhi(wa[e]) - x1 * k2 + bfhi(ra[e]);
        o[e] = pack2bf(d0, d1);
        atomicAdd(&acc[c * 8 + 2 * e], g0 * x0 * rs);
        atomicAdd(&acc[c * 8 + 2 * e + 1], g1 * x1 * rs);
      }
      dx[row * HC + c] = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
  __syncthreads();
  float* outp = dw_partial + (int64_t)blockIdx.x * H;
  for (int i = tid; i < H; i += 256) outp[i] = acc[i];
}

// one workgroup per 64 columns: 4 waves split the partial rows, lanes are consecutive columns (coalesced 256-B rows)
// Fast path for H = NCH * 512 (every lane owns the same 8*NCH columns in every row): the row stays in registers between the two
// sweeps and the weight gradient accumulates in registers; one LDS reduction per workgroup at the end instead of LDS atomics per row.
template <int NCH>
__global__ void __launch_bounds__(256, NCH >= 8 ? 2 : 1) rmsnorm_bwd_reg_k(const uint4* __restrict__ dy, const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                         const uint4* dres, uint4* dx, float* __restrict__ dw_partial, int64_t T, int H,
                                                         float eps) {
  __shared__ float red[3][NCH * 512];
  // the weight row lives in LDS (8 KiB at H = 4096), not in 4 * NCH registers per lane: with x, dy, dres (all three requested up front,
  // ONE memory round trip per row) and the 8 * NCH weight-gradient accumulators the H = 4096 instantiation then fits 256 registers =
  // two waves per SIMD; the round-2 form (weights in registers, dres fetched after the row statistics) needed 339 = one wave per SIMD
  // with two round trips per row and ran at 3.6 TB/s
  __shared__ uint4 wsh[NCH * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform row pointers: scalar base + one per-lane offset register
  const int HC = H >> 3;
  float dwa[NCH][8];
  for (int i = tid; i < NCH * 64; i += 256) wsh[i] = w[i];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) dwa[c][e] = 0.f;
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < T; row += (int64_t)gridDim.x * 4) {
    uint4 xv[NCH], gv[NCH], rv[NCH];
    const uint4* xrow = x + row * HC;
    const uint4* grow = dy + row * HC;
    const uint4* rrow = dres != nullptr ? dres + row * HC : nullptr;
    uint4* orow = dx + row * HC;
#pragma unroll
    for (int c = 0; c < NCH; ++c) { xv[c] = xrow[c * 64 + lane]; gv[c] = grow[c * 64 + lane]; }
#pragma unroll
    for (int c = 0; c < NCH; ++c) rv[c] = rrow != nullptr ? rrow[c * 64 + lane] : make_uint4(0, 0, 0, 0);
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint4 wv = wsh[c * 64 + lane];
      const uint32_t xa[4] = {xv[c].x, xv[c].y, xv[c].z, xv[c].w}, ga[4] = {gv[c].x, gv[c].y, gv[c].z, gv[c].w},
                     wa[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = bflo(xa[e]), x1 = bfhi(xa[e]);
        ss += x0 * x0 + x1 * x1;
        dot += bflo(ga[e]) * bflo(wa[e]) * x0 + bfhi(ga[e]) * bfhi(wa[e]) * x1;
      }
      if (NCH >= 8) __builtin_amdgcn_sched_barrier(0);          // one column block at a time: bounds the live fp32 temporaries
    }
    ss = wave_sum(ss); dot = wave_sum(dot);
    const float rs = rsqrtf(ss / (float)H + eps);
    const float k2 = rs * rs * rs * dot / (float)H;
    // the packed rows are made opaque here: otherwise the fp32 unpackings of the first sweep (64 + 64 registers at H = 4096) are kept
    // alive across the reduction for the second sweep and the kernel spills
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      asm volatile("" : "+v"(xv[c].x), "+v"(xv[c].y), "+v"(xv[c].z), "+v"(xv[c].w), "+v"(gv[c].x), "+v"(gv[c].y), "+v"(gv[c].z), "+v"(gv[c].w));
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint4 wv = wsh[c * 64 + lane];
      const uint32_t xa[4] = {xv[c].x, xv[c].y, xv[c].z, xv[c].w}, ga[4] = {gv[c].x, gv[c].y, gv[c].z, gv[c].w},
                     wa[4] = {wv.x, wv.y, wv.z, wv.w}, ra[4] = {rv[c].x, rv[c].y, rv[c].z, rv[c].w};
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = bflo(xa[e]), x1 = bfhi(xa[e]), g0 = bflo(ga[e]), g1 = bfhi(ga[e]);
        o[e] = pack2bf_hw(rs * g0 * bflo(wa[e]) - x0 * k2 + bflo(ra[e]), rs * g1 * bfhi(wa[e]) - x1 * k2 + bfhi(ra[e]));
        dwa[c][2 * e] += g0 * x0 * rs;
        dwa[c][2 * e + 1] += g1 * x1 * rs;
      }
      orow[c * 64 + lane] = make_uint4(o[0], o[1], o[2], o[3]);
      if (NCH >= 8) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // waves 1..3 park their partial sums in LDS, wave 0 adds them up and writes the workgroup's row of dw_partial
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave - 1][(c * 64 + lane) * 8 + e] = dwa[c][e];
  }
  __syncthreads();
  if (wave == 0) {
    float* outp = dw_partial + (int64_t)blockIdx.x * H;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int col = (c * 64 + lane) * 8 + e;
        outp[col] = dwa[c][e] + red[0][col] + red[1][col] + red[2][col];
      }
  }
}

__global__ void __launch_bounds__(256) rmsnorm_dw_reduce_k(const float* __restrict__ partial, float* __restrict__ dw, int nblk, int H) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  // eight independent row loads in flight per lane (one load at a time made the 512-row reduction a 40 us latency chain, 0.34 % of a
  // training step); the summation order is fixed: row b goes to accumulator (b / 4) % 8 of wave b % 4, accumulators and waves are
  // folded in index order
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < H) {
    int b = wave;
    for (; b + 28 < nblk; b += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 4 * u) * H + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += v[u];
    }
    for (int u = 0; b < nblk; b += 4, ++u) s[u] += partial[(int64_t)b * H + i];
  }
  red[wave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (wave == 0 && i < H) dw[i] += red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// ---------------------------------------------------------------- SwiGLU on the concatenated layout gu = [gate | up], [T, 2I]

__global__ void __launch_bounds__(256) swiglu_fwd_k(const uint16_t* __restrict__ gu, uint16_t* __restrict__ act, int64_t T, int I) {
  const int IC = I >> 3;
  const int64_t total = T * IC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / IC; const int c = (int)(i - t * IC);
    const uint4 g = *reinterpret_cast<const uint4*>(gu + t * 2 * I + c * 8);
    const uint4 u = *reinterpret_cast<const uint4*>(gu + t * 2 * I + I + c * 8);
    const uint32_t ga[4] = {g.x, g.y, g.z, g.w}, ua[4] = {u.x, u.y, u.z, u.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = bflo(ga[e]), g1 = bfhi(ga[e]);
      o[e] = pack2bf(round_bf(silu_f(g0)) * bflo(ua[e]), round_bf(silu_f(g1)) * bfhi(ua[e]));
    }
    *reinterpret_cast<uint4*>(act + t * I + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

__global__ void __launch_bounds__(256) swiglu_bwd_k(const uint16_t* __restrict__ gu, const uint16_t* __restrict__ dact,
                                                    uint16_t* __restrict__ dgu, int64_t T, int I) {
  const int IC = I >> 3;
  const int64_t total = T * IC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / IC; const int c = (int)(i - t * IC);
    const uint4 g = *reinterpret_cast<const uint4*>(gu + t * 2 * I + c * 8);
    const uint4 u = *reinterpret_cast<const uint4*>(gu + t * 2 * I + I + c * 8);
    const uint4 d = *reinterpret_cast<const uint4*>(dact + t * I + c * 8);
    const uint32_t ga[4] = {g.x, g.y, g.z, g.w}, ua[4] = {u.x, u.y, u.z, u.w}, da[4] = {d.x, d.y, d.z, d.w};
    uint32_t og[4], ou[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dg0, du0, dg1, du1;
      swiglu_bwd_elem(bflo(da[e]), bflo(ga[e]), bflo(ua[e]), dg0, du0);
      swiglu_bwd_elem(bfhi(da[e]), bfhi(ga[e]), bfhi(ua[e]), dg1, du1);
      og[e] = pack2bf_hw(dg0, dg1);
      ou[e] = pack2bf_hw(du0, du1);
    }
    *reinterpret_cast<uint4*>(dgu + t * 2 * I + c * 8) = make_uint4(og[0], og[1], og[2], og[3]);
    *reinterpret_cast<uint4*>(dgu + t * 2 * I + I + c * 8) = make_uint4(ou[0], ou[1], ou[2], ou[3]);
  }
}

// ---------------------------------------------------------------- embedding backward, deterministic (no atomics)
// The token rows are visited in the order of a STABLE sort of their ids (`order`; `sorted_ids[i] = ids[order[i]]`): workgroup i owns the run
// of equal ids that STARTS at sorted position i (every other workgroup exits), sums the run's rows of dh in that fixed order in fp32 and
// folds the sum into the bf16 gradient row:  grad[id] = bf16(grad[id] + sum_{t in run} dh[t]).  Only touched rows are read or written
// (the round-2 kernel scattered with fp32 atomics into a dense fp32 [V,H] table -- 524 MB at the 7B shape -- that was then folded and
// cleared in full for every GradCache chunk, and whose sums depended on the arrival order of the atomics).
__global__ void __launch_bounds__(256) embed_scatter_sorted_k(const uint4* __restrict__ dh, const int64_t* __restrict__ sorted_ids,
                                                              const int64_t* __restrict__ order, uint4* __restrict__ grad, int64_t T, int HC,
                                                              int64_t V) {
  // ids are clamped to [0, V) BEFORE the runs are detected: a sorted sequence stays sorted under the clamp, so all out-of-range ids fall
  // into the run of row 0 / row V - 1 and every gradient row still has exactly one owning workgroup (clamping per run, as round 3 did,
  // let two distinct out-of-range ids -- or -1 and 0 -- update the same row from two workgroups without synchronisation: ADVICE r03)
  auto clampv = [V](int64_t x) { return x < 0 ? (int64_t)0 : (x >= V ? V - 1 : x); };
  const int64_t i = blockIdx.x;
  const int64_t row = clampv(sorted_ids[i]);
  if (i > 0 && clampv(sorted_ids[i - 1]) == row) return;
  int64_t end = i + 1;
  while (end < T && clampv(sorted_ids[end]) == row) ++end;
  for (int c = threadIdx.x; c < HC; c += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t j = i; j < end; ++j) {
      const uint4 v = dh[order[j] * HC + c];
      acc[0] += bflo(v.x); acc[1] += bfhi(v.x); acc[2] += bflo(v.y); acc[3] += bfhi(v.y);
      acc[4] += bflo(v.z); acc[5] += bfhi(v.z); acc[6] += bflo(v.w); acc[7] += bfhi(v.w);
    }
    uint4 g = grad[row * HC + c];
    g.x = pack2bf(bflo(g.x) + acc[0], bfhi(g.x) + acc[1]); g.y = pack2bf(bflo(g.y) + acc[2], bfhi(g.y) + acc[3]);
    g.z = pack2bf(bflo(g.z) + acc[4], bfhi(g.z) + acc[5]); g.w = pack2bf(bflo(g.w) + acc[6], bfhi(g.w) + acc[7]);
    grad[row * HC + c] = g;
  }
}

// acc_bf16[i] = bf16(acc_bf16[i] + x_f32[i])  : fold an fp32 gradient into a bf16 .grad buffer
__global__ void __launch_bounds__(256) accum_bf16_from_f32_k(uint16_t* __restrict__ acc, const float* __restrict__ x, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc[i] = (uint16_t)f2bf(bf2f(acc[i]) + x[i]);
}

static inline int grid_for(int64_t items) {
  int64_t g = (items + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace grit

using namespace grit;

// A dynamic-LDS opt-in is a PER-DEVICE function attribute: set once per (kernel, device), from whichever thread gets there first
// (autograd worker threads call into the backward entry points concurrently).
template <typename KernelT>
static void lds_optin_once(KernelT kernel, std::atomic<uint64_t>& done, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
}

extern "C" {

int64_t grit_rmsnorm_bwd_workspace_rows(int64_t T) {
  if (T > 2048) return 512;
  int64_t g = (T + 3) / 4;
  return g < 1 ? 1 : g;
}

int grit_rmsnorm_bwd(const void* dy, const void* x, const void* w, const void* dres, void* dx, float* dw_partial, float* dw, int64_t T,
                     int H, float eps, void* stream) {
  GRIT_REQUIRE(dy && x && w && dx && dw_partial && dw, GRIT_E_BADARG, "grit_rmsnorm_bwd: null pointer");
  GRIT_REQUIRE(T > 0 && H > 0, GRIT_E_BADARG, "grit_rmsnorm_bwd: bad sizes");
  GRIT_REQUIRE(H % 8 == 0 && H <= 32768, GRIT_E_UNSUPPORTED, "grit_rmsnorm_bwd: H=%d must be a multiple of 8 and <= 32768", H);
  GRIT_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(w) && aligned16(dx) && (dres == nullptr || aligned16(dres)), GRIT_E_BADARG,
               "grit_rmsnorm_bwd: pointers must be 16-byte aligned");
  const int nblk = (int)grit_rmsnorm_bwd_workspace_rows(T);
  static std::atomic<uint64_t> optin{0};
  lds_optin_once(rmsnorm_bwd_k, optin, 128 * 1024);
  hipStream_t st = (hipStream_t)stream;
#define GRIT_RMSBWD_REG(NCH_)                                                                                                   \
  hipLaunchKernelGGL(rmsnorm_bwd_reg_k<NCH_>, dim3(nblk), dim3(256), 0, st, (const uint4*)dy, (const uint4*)x, (const uint4*)w,      \
                     (const uint4*)dres, (uint4*)dx, dw_partial, T, H, eps)
  if (H == 512) GRIT_RMSBWD_REG(1);
  else if (H == 1024) GRIT_RMSBWD_REG(2);
  else if (H == 2048) GRIT_RMSBWD_REG(4);
  else if (H == 4096) GRIT_RMSBWD_REG(8);
  else
    hipLaunchKernelGGL(rmsnorm_bwd_k, dim3(nblk), dim3(256), (size_t)H * 4, st, (const uint4*)dy, (const uint4*)x, (const uint4*)w,
                       (const uint4*)dres, (uint4*)dx, dw_partial, T, H, eps);
  GRIT_CHECK_LAUNCH("grit_rmsnorm_bwd");
  hipLaunchKernelGGL(rmsnorm_dw_reduce_k, dim3((H + 63) / 64), dim3(256), 0, st, (const float*)dw_partial, dw, nblk, H);
  GRIT_CHECK_LAUNCH("grit_rmsnorm_bwd: reduce");
  return GRIT_OK;
}

int grit_swiglu_fwd(const void* gu, void* act, int64_t T, int I, void* stream) {
  GRIT_REQUIRE(gu && act, GRIT_E_BADARG, "grit_swiglu_fwd: null pointer");
  GRIT_REQUIRE(T > 0 && I > 0 && I % 8 == 0, GRIT_E_UNSUPPORTED, "grit_swiglu_fwd: I=%d must be a positive multiple of 8", I);
  GRIT_REQUIRE(aligned16(gu) && aligned16(act), GRIT_E_BADARG, "grit_swiglu_fwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(swiglu_fwd_k, dim3(grid_for(T * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)gu, (uint16_t*)act, T, I);
  GRIT_CHECK_LAUNCH("grit_swiglu_fwd");
  return GRIT_OK;
}

int grit_swiglu_bwd(const void* gu, const void* dact, void* dgu, int64_t T, int I, void* stream) {
  GRIT_REQUIRE(gu && dact && dgu, GRIT_E_BADARG, "grit_swiglu_bwd: null pointer");
  GRIT_REQUIRE(T > 0 && I > 0 && I % 8 == 0, GRIT_E_UNSUPPORTED, "grit_swiglu_bwd: I=%d must be a positive multiple of 8", I);
  GRIT_REQUIRE(aligned16(gu) && aligned16(dact) && aligned16(dgu), GRIT_E_BADARG, "grit_swiglu_bwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(swiglu_bwd_k, dim3(grid_for(T * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)gu,
                     (const uint16_t*)dact, (uint16_t*)dgu, T, I);
  GRIT_CHECK_LAUNCH("grit_swiglu_bwd");
  return GRIT_OK;
}

int grit_embed_scatter_add_sorted(const void* dh, const int64_t* sorted_ids, const int64_t* order, void* grad, int64_t T, int H, int64_t V,
                                 void* stream) {
  GRIT_REQUIRE(dh && sorted_ids && order && grad, GRIT_E_BADARG, "grit_embed_scatter_add_sorted: null pointer");
  GRIT_REQUIRE(T > 0 && T < (1ll << 31) && H > 0 && V > 0 && H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_embed_scatter_add_sorted: bad sizes");
  GRIT_REQUIRE(aligned16(dh) && aligned16(grad), GRIT_E_BADARG, "grit_embed_scatter_add_sorted: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(embed_scatter_sorted_k, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, (const uint4*)dh, sorted_ids, order,
                     (uint4*)grad, T, H / 8, V);
  GRIT_CHECK_LAUNCH("grit_embed_scatter_add_sorted");
  return GRIT_OK;
}

int grit_accum_bf16_from_f32(void* acc, const float* x, int64_t n, void* stream) {
  GRIT_REQUIRE(acc && x, GRIT_E_BADARG, "grit_accum_bf16_from_f32: null pointer");
  GRIT_REQUIRE(n > 0, GRIT_E_BADARG, "grit_accum_bf16_from_f32: bad size");
  hipLaunchKernelGGL(accum_bf16_from_f32_k, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)acc, x, n);
  GRIT_CHECK_LAUNCH("grit_accum_bf16_from_f32");
  return GRIT_OK;
}

}  // extern "C"
