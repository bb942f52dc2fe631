// HBM-bound pieces of the encoder forward: embedding gather, RMSNorm, RoPE, mask packing, transpose.
// All loads/stores are 16 B per lane (8 bf16), one wave (64 lanes) covers 1 KiB contiguous.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace grit {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- embedding gather
// scripts/modeling_mistral_gritlm.py:994  inputs_embeds = self.embed_tokens(input_ids)
__global__ void __launch_bounds__(256) embed_gather_k(const uint4* __restrict__ table, const int64_t* __restrict__ ids,
                                                      uint4* __restrict__ out, int64_t T, int HC, int64_t V) {
  const int64_t total = T * HC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / HC;
    const int c = (int)(i - t * HC);
    int64_t id = ids[t];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    out[i] = table[id * HC + c];
  }
}

// fp32 residual stream (opt-in, DESIGN "precision policy"): the same gather, rows widened to fp32 (exact)
__global__ void __launch_bounds__(256) embed_gather_f32_k(const uint4* __restrict__ table, const int64_t* __restrict__ ids,
                                                          float4* __restrict__ out, int64_t T, int HC, int64_t V) {
  const int64_t total = T * HC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / HC;
    const int c = (int)(i - t * HC);
    int64_t id = ids[t];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const uint4 v = table[id * HC + c];
    out[2 * i] = make_float4(bflo(v.x), bfhi(v.x), bflo(v.y), bfhi(v.y));
    out[2 * i + 1] = make_float4(bflo(v.z), bfhi(v.z), bflo(v.w), bfhi(v.w));
  }
}

// ---------------------------------------------------------------- RMSNorm forward
// scripts/modeling_mistral_gritlm.py:84-89.  One wave per row, row held in registers when H = NCH*512.
__device__ __forceinline__ float sumsq8(const uint4& v) {
  float s = 0.f, a;
  a = bflo(v.x); s += a * a; a = bfhi(v.x); s += a * a;
  a = bflo(v.y); s += a * a; a = bfhi(v.y); s += a * a;
  a = bflo(v.z); s += a * a; a = bfhi(v.z); s += a * a;
  a = bflo(v.w); s += a * a; a = bfhi(v.w); s += a * a;
  return s;
}
__device__ __forceinline__ uint32_t norm2(uint32_t x, uint32_t w, float rs) {
  // bf16(w * bf16(x*rs)) for both halves: the reference rounds after the normalise and after the weight
  const float lo = bflo(w) * round_bf(bflo(x) * rs);
  const float hi = bfhi(w) * round_bf(bfhi(x) * rs);
  return pack2bf(lo, hi);
}
__device__ __forceinline__ uint4 norm8(const uint4& x, const uint4& w, float rs) {
  uint4 o;
  o.x = norm2(x.x, w.x, rs); o.y = norm2(x.y, w.y, rs); o.z = norm2(x.z, w.z, rs); o.w = norm2(x.w, w.w, rs);
  return o;
}

// PW (a handful of rows: the decode step's one-row norms): the weight row is requested together with x, in front of the reduction --
// a launch that small is one dependent chain, and the weights' L2 round trip otherwise FOLLOWS the reduction (4.6 us per launch at
// T = 1).  With many rows other waves hide that latency and the 32 extra registers would only cost occupancy, so PW is off there.
template <int NCH, bool PW = false>
__global__ void __launch_bounds__(256) rmsnorm_fwd_reg_k(const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                         uint4* __restrict__ y, int64_t T, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int HC = H >> 3;
  const uint4* xr = x + row * HC;
  uint4 v[NCH], wv[PW ? NCH : 1];
#pragma unroll
  for (int c = 0; c < NCH; ++c) v[c] = xr[c * 64 + lane];
  if constexpr (PW) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) wv[c] = w[c * 64 + lane];
  }
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) ss += sumsq8(v[c]);
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)H + eps);
  uint4* yr = y + row * HC;
#pragma unroll
  for (int c = 0; c < NCH; ++c) yr[c * 64 + lane] = norm8(v[c], PW ? wv[c] : w[c * 64 + lane], rs);
}

__global__ void __launch_bounds__(256) rmsnorm_fwd_generic_k(const uint4* __restrict__ x, const uint4* __restrict__ w,
                                                             uint4* __restrict__ y, int64_t T, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int HC = H >> 3;
  const uint4* xr = x + row * HC;
  float ss = 0.f;
  for (int c = lane; c < HC; c += 64) ss += sumsq8(xr[c]);
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)H + eps);
  uint4* yr = y + row * HC;
  for (int c = lane; c < HC; c += 64) yr[c] = norm8(xr[c], w[c], rs);
}

// RMSNorm of an fp32 residual stream: x [T,H] fp32 -> y bf16 = bf16(w * (x * rsqrt(mean(x^2) + eps))), ONE rounding (what the reference
// computes when it runs in fp32, :84-89, rounded once to the bf16 operand the next GEMM takes).  One wave per row; a lane owns the
// float4 at index c*64 + lane of every 256-element chunk c, so every load instruction covers 1 KiB contiguous and every store 512 B.
// F16: y = fp16 (the "f16_operands" policy; w stays bf16); a value beyond the fp16 range sets the overflow flag word `ovf`.
template <int NCH, bool F16 = false>   // H = NCH * 256
__global__ void __launch_bounds__(256) rmsnorm_fwd_f32in_reg_k(const float4* __restrict__ x, const uint2* __restrict__ w,
                                                               uint2* __restrict__ y, int64_t T, int H, float eps, unsigned int* ovf) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int HQ = H >> 2;
  const float4* xr = x + row * HQ;
  float4 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) v[c] = xr[c * 64 + lane];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)H + eps);
  uint2* yr = y + row * HQ;
  uint32_t bad = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const uint2 ww = w[c * 64 + lane];
    const uint2 o = make_uint2(pack2_op<F16>(bflo(ww.x) * (v[c].x * rs), bfhi(ww.x) * (v[c].y * rs)),
                               pack2_op<F16>(bflo(ww.y) * (v[c].z * rs), bfhi(ww.y) * (v[c].w * rs)));
    if constexpr (F16) bad |= h2_nonfinite(o.x) | h2_nonfinite(o.y);
    yr[c * 64 + lane] = o;
  }
  if constexpr (F16) {
    if (bad != 0 && ovf != nullptr) atomicOr(ovf, 1u);
  }
}
template <bool F16 = false>
__global__ void __launch_bounds__(256) rmsnorm_fwd_f32in_generic_k(const float4* __restrict__ x, const uint2* __restrict__ w,
                                                                   uint2* __restrict__ y, int64_t T, int H, float eps, unsigned int* ovf) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int HQ = H >> 2;
  const float4* xr = x + row * HQ;
  float ss = 0.f;
  for (int c = lane; c < HQ; c += 64) { const float4 a = xr[c]; ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; }
  ss = wave_sum(ss);
  const float rs = rsqrtf(ss / (float)H + eps);
  uint2* yr = y + row * HQ;
  uint32_t bad = 0;
  for (int c = lane; c < HQ; c += 64) {
    const float4 a = xr[c];
    const uint2 ww = w[c];
    const uint2 o = make_uint2(pack2_op<F16>(bflo(ww.x) * (a.x * rs), bfhi(ww.x) * (a.y * rs)), pack2_op<F16>(bflo(ww.y) * (a.z * rs), bfhi(ww.y) * (a.w * rs)));
    if constexpr (F16) bad |= h2_nonfinite(o.x) | h2_nonfinite(o.y);
    yr[c] = o;
  }
  if constexpr (F16) {
    if (bad != 0 && ovf != nullptr) atomicOr(ovf, 1u);
  }
}

// RMSNorm of an fp16 residual stream (the "f16_stream" policy): x [T,H] fp16 -> y = round(w * (x * rsqrt(mean(x^2) + eps))) in fp32 with ONE
// rounding, to fp16 (OUT_F16: the operand of the next GEMM) or to bf16 (last_hidden_state, the pooling kernels' input).  w is bf16.
template <bool OUT_F16>
__device__ __forceinline__ uint4 norm8_h16(const uint4& x, const uint4& w, float rs) {
  uint4 o;
  o.x = pack2_op<OUT_F16>(bflo(w.x) * (hlo(x.x) * rs), bfhi(w.x) * (hhi(x.x) * rs));
  o.y = pack2_op<OUT_F16>(bflo(w.y) * (hlo(x.y) * rs), bfhi(w.y) * (hhi(x.y) * rs));
  o.z = pack2_op<OUT_F16>(bflo(w.z) * (hlo(x.z) * rs), bfhi(w.z) * (hhi(x.z) * rs));
  o.w = pack2_op<OUT_F16>(bflo(w.w) * (hlo(x.w) * rs), bfhi(w.w) * (hhi(x.w) * rs));
  return o;
}
__device__ __forceinline__ float sumsq8_h16(const uint4& v) {
  float s = 0.f, a;
  a = hlo(v.x); s += a * a; a = hhi(v.x); s += a * a;
  a = hlo(v.y); s += a * a; a = hhi(v.y); s += a * a;
  a = hlo(v.z); s += a * a; a = hhi(v.z); s += a * a;
  a = hlo(v.w); s += a * a; a = hhi(v.w); s += a * a;
  return s;
}
template <int NCH, bool OUT_F16>      // NCH > 0: H = NCH * 512, row held in registers; NCH = 0: any H % 8 == 0
__global__ void __launch_bounds__(256) rmsnorm_fwd_h16_k(const uint4* __restrict__ x, const uint4* __restrict__ w, uint4* __restrict__ y,
                                                         int64_t T, int H, float eps, unsigned int* ovf) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= T) return;
  const int HC = H >> 3;
  const uint4* xr = x + row * HC;
  uint4* yr = y + row * HC;
  uint32_t bad = 0;
  if constexpr (NCH > 0) {
    uint4 v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = xr[c * 64 + lane];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) ss += sumsq8_h16(v[c]);
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint4 o = norm8_h16<OUT_F16>(v[c], w[c * 64 + lane], rs);
      if constexpr (OUT_F16) bad |= h2_nonfinite(o.x) | h2_nonfinite(o.y) | h2_nonfinite(o.z) | h2_nonfinite(o.w);
      yr[c * 64 + lane] = o;
    }
  } else {
    float ss = 0.f;
    for (int c = lane; c < HC; c += 64) ss += sumsq8_h16(xr[c]);
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
    for (int c = lane; c < HC; c += 64) {
      const uint4 o = norm8_h16<OUT_F16>(xr[c], w[c], rs);
      if constexpr (OUT_F16) bad |= h2_nonfinite(o.x) | h2_nonfinite(o.y) | h2_nonfinite(o.z) | h2_nonfinite(o.w);
      yr[c] = o;
    }
  }
  if constexpr (OUT_F16) {
    if (bad != 0 && ovf != nullptr) atomicOr(ovf, 1u);
  }
}

// ---------------------------------------------------------------- fp16 overflow flag of the "f16_operands" policy
// One word per device (module-scope device variable: the library never allocates).  Every kernel that rounds to fp16 ORs 1 into it when
// it stores an inf / nan; grit_f16_overflow_flag() reads (and optionally clears) it on the caller's stream.
__device__ unsigned int g_f16_overflow[1];
unsigned int* f16_flag_ptr() {
  static unsigned int* cached[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  unsigned int* p = __atomic_load_n(&cached[dev & 63], __ATOMIC_ACQUIRE);
  if (p == nullptr) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_f16_overflow)) != hipSuccess || q == nullptr) {
      (void)hipGetLastError();
      return nullptr;
    }
    p = (unsigned int*)q;
    __atomic_store_n(&cached[dev & 63], p, __ATOMIC_RELEASE);
  }
  return p;
}

// ---------------------------------------------------------------- RoPE (in place on q,k of the fused qkv rows)
// scripts/modeling_mistral_gritlm.py:138-163: x' = x*cos + rotate_half(x)*sin, fp32 math, one rounding.
__device__ __forceinline__ void rot2(uint32_t a, uint32_t b, float c0, float s0, float c1, float s1, uint32_t& oa,
                                     uint32_t& ob) {
  const float a0 = bflo(a), a1 = bfhi(a), b0 = bflo(b), b1 = bfhi(b);
  oa = pack2bf(rope_lo(a0, b0, c0, s0), rope_lo(a1, b1, c1, s1));
  ob = pack2bf(rope_hi(a0, b0, c0, s0), rope_hi(a1, b1, c1, s1));
}
__global__ void __launch_bounds__(256) rope_k(uint16_t* __restrict__ qkv, const float4* __restrict__ cos_tab,
                                              const float4* __restrict__ sin_tab, const int32_t* __restrict__ positions, int64_t T,
                                              int S, int nheads, int d, int64_t row_stride, float sgn) {
  const int jc_n = d >> 4;  // 16-byte chunks per half head
  const int64_t total = T * nheads * jc_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int jc = (int)(i % jc_n);
    const int64_t r = i / jc_n;
    const int head = (int)(r % nheads);
    const int64_t t = r / nheads;
    const int pos = positions ? positions[t] : (int)(t % S);
    uint16_t* base = qkv + t * row_stride + (int64_t)head * d + jc * 8;
    uint4 x1 = *reinterpret_cast<uint4*>(base);
    uint4 x2 = *reinterpret_cast<uint4*>(base + (d >> 1));
    const int64_t tb = ((int64_t)pos * (d >> 1) + jc * 8) >> 2;
    const float4 c0 = cos_tab[tb], c1 = cos_tab[tb + 1];
    float4 s0 = sin_tab[tb], s1 = sin_tab[tb + 1];
    s0.x *= sgn; s0.y *= sgn; s0.z *= sgn; s0.w *= sgn; s1.x *= sgn; s1.y *= sgn; s1.z *= sgn; s1.w *= sgn;
    uint4 o1, o2;
    rot2(x1.x, x2.x, c0.x, s0.x, c0.y, s0.y, o1.x, o2.x);
    rot2(x1.y, x2.y, c0.z, s0.z, c0.w, s0.w, o1.y, o2.y);
    rot2(x1.z, x2.z, c1.x, s1.x, c1.y, s1.y, o1.z, o2.z);
    rot2(x1.w, x2.w, c1.z, s1.z, c1.w, s1.w, o1.w, o2.w);
    *reinterpret_cast<uint4*>(base) = o1;
    *reinterpret_cast<uint4*>(base + (d >> 1)) = o2;
  }
}

// ---------------------------------------------------------------- key-padding bitmask
__global__ void __launch_bounds__(256) mask_pack_k(const int64_t* __restrict__ mask, uint64_t* __restrict__ bits, int B,
                                                   int S, int W) {
  const int lane = threadIdx.x & 63;
  const int64_t word = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (word >= (int64_t)B * W) return;
  const int b = (int)(word / W), wi = (int)(word % W);
  const int s = wi * 64 + lane;
  const bool on = (s < S) && (mask[(int64_t)b * S + s] != 0);
  const uint64_t m = __ballot(on);
  if (lane == 0) bits[word] = m;
}

// ---------------------------------------------------------------- bf16 transpose [R,C] -> [C,R]
// 64 x 256 tile per workgroup: eight 16-byte loads in flight per lane, row-major LDS image (544-byte pitch: 4 consecutive rows sit 8
// banks apart), read back TRANSPOSED by the hardware (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block, lane c
// receives column c's 4 rows) -- two reads give a lane 8 consecutive source rows of one column = 16 bytes of one output row; the four
// lane groups of a wave hold four neighbouring pieces of the same 16 output rows (64 contiguous bytes per row and store).
// (Round 1 read the image back with sixteen 2-byte LDS reads per lane from a 64 x 64 tile: 5.05 TB/s over a layer's eight operand
//  shapes in the training step; this form 5.2 TB/s there, 5.9 TB/s on 16384 x 4096 alone.)
constexpr int TR_ROWS = 64, TR_COLS = 256, TR_PITCH = 544;
typedef __attribute__((ext_vector_type(4))) short tr_s16x4_t;
__global__ void __launch_bounds__(256) transpose_k(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int64_t R,
                                                   int64_t C, int64_t ld_in, int64_t ld_out) {
  __shared__ __attribute__((aligned(16))) char tile[TR_ROWS * TR_PITCH];
  const int64_t r0 = (int64_t)blockIdx.y * TR_ROWS, c0 = (int64_t)blockIdx.x * TR_COLS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4 v[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = tid + it * 256;                   // 2048 chunks of 8 elements: 32 per row
    const int r = idx >> 5, cc = (idx & 31) * 8;
    v[it] = make_uint4(0, 0, 0, 0);
    if (r0 + r < R && c0 + cc < C) v[it] = *reinterpret_cast<const uint4*>(in + (r0 + r) * ld_in + c0 + cc);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int idx = tid + it * 256;
    *reinterpret_cast<uint4*>(tile + (idx >> 5) * TR_PITCH + (idx & 31) * 16) = v[it];
  }
  __syncthreads();
  const int g = lane >> 4, j = lane & 15;
  const int rd = (j >> 2) * TR_PITCH + (j & 3) * 8;   // lane j of a group points at block row j/4, block columns 4 (j%4) .. +3
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int u = wave * 8 + it;                      // 32 units per workgroup: 16 column blocks x 2 row halves
    const int d0 = (u >> 1) * 16, k0 = (u & 1) * 32 + g * 8;
    const char* p = tile + k0 * TR_PITCH + d0 * 2 + rd;
    const tr_s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_s16x4_t*)(p));
    const tr_s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_s16x4_t*)(p + 4 * TR_PITCH));
    const int64_t c = c0 + d0 + j;                    // output row = source column; 8 consecutive source rows from k0
    if (c < C && r0 + k0 < R) {
      const uint4 w = make_uint4((uint16_t)a[0] | ((uint32_t)(uint16_t)a[1] << 16), (uint16_t)a[2] | ((uint32_t)(uint16_t)a[3] << 16),
                                 (uint16_t)b[0] | ((uint32_t)(uint16_t)b[1] << 16), (uint16_t)b[2] | ((uint32_t)(uint16_t)b[3] << 16));
      *reinterpret_cast<uint4*>(out + c * ld_out + r0 + k0) = w;
    }
  }
}

static inline int grid_for(int64_t items, int threads) {
  int64_t g = (items + threads - 1) / threads;
  const int64_t cap = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace grit

using namespace grit;

extern "C" {

int grit_version(void) { return GRIT_ABI_VERSION; }
const char* grit_last_error_string(void) { return g_err; }

int grit_embed_gather(const void* table, const int64_t* ids, void* out, int64_t T, int H, int64_t V, void* stream) {
  if (T == 0) return GRIT_OK;  // empty batch: nothing to do (empty tensors have null data pointers)
  GRIT_REQUIRE(table && ids && out, GRIT_E_BADARG, "grit_embed_gather: null pointer");
  GRIT_REQUIRE(T >= 0 && H > 0 && V > 0, GRIT_E_BADARG, "grit_embed_gather: bad sizes T=%lld H=%d V=%lld", (long long)T, H, (long long)V);
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_embed_gather: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(aligned16(table) && aligned16(out), GRIT_E_BADARG, "grit_embed_gather: pointers must be 16-byte aligned");
  if (T == 0) return GRIT_OK;
  const int HC = H / 8;
  hipLaunchKernelGGL(embed_gather_k, dim3(grid_for(T * HC, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)table, ids,
                     (uint4*)out, T, HC, V);
  GRIT_CHECK_LAUNCH("grit_embed_gather");
  return GRIT_OK;
}

int grit_rmsnorm_fwd(const void* x, const void* w, void* y, int64_t T, int H, float eps, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(x && w && y, GRIT_E_BADARG, "grit_rmsnorm_fwd: null pointer");
  GRIT_REQUIRE(T >= 0 && H > 0, GRIT_E_BADARG, "grit_rmsnorm_fwd: bad sizes");
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_rmsnorm_fwd: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), GRIT_E_BADARG, "grit_rmsnorm_fwd: pointers must be 16-byte aligned");
  if (T == 0) return GRIT_OK;
  const dim3 grid((unsigned)((T + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const uint4 *xp = (const uint4*)x, *wp = (const uint4*)w;
  uint4* yp = (uint4*)y;
  if (H % 512 == 0 && H / 512 <= 16) {
    switch (H / 512) {
      case 1: hipLaunchKernelGGL(rmsnorm_fwd_reg_k<1>, grid, block, 0, st, xp, wp, yp, T, H, eps); break;
      case 2: hipLaunchKernelGGL(rmsnorm_fwd_reg_k<2>, grid, block, 0, st, xp, wp, yp, T, H, eps); break;
      case 4: hipLaunchKernelGGL(rmsnorm_fwd_reg_k<4>, grid, block, 0, st, xp, wp, yp, T, H, eps); break;
      case 8:
        if (T <= 16) hipLaunchKernelGGL((rmsnorm_fwd_reg_k<8, true>), grid, block, 0, st, xp, wp, yp, T, H, eps);
        else hipLaunchKernelGGL(rmsnorm_fwd_reg_k<8>, grid, block, 0, st, xp, wp, yp, T, H, eps);
        break;
      case 16: hipLaunchKernelGGL(rmsnorm_fwd_reg_k<16>, grid, block, 0, st, xp, wp, yp, T, H, eps); break;
      default: hipLaunchKernelGGL(rmsnorm_fwd_generic_k, grid, block, 0, st, xp, wp, yp, T, H, eps); break;
    }
  } else {
    hipLaunchKernelGGL(rmsnorm_fwd_generic_k, grid, block, 0, st, xp, wp, yp, T, H, eps);
  }
  GRIT_CHECK_LAUNCH("grit_rmsnorm_fwd");
  return GRIT_OK;
}

int grit_embed_gather_f32(const void* table, const int64_t* ids, float* out, int64_t T, int H, int64_t V, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(table && ids && out, GRIT_E_BADARG, "grit_embed_gather_f32: null pointer");
  GRIT_REQUIRE(T >= 0 && H > 0 && V > 0, GRIT_E_BADARG, "grit_embed_gather_f32: bad sizes T=%lld H=%d V=%lld", (long long)T, H, (long long)V);
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_embed_gather_f32: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(aligned16(table) && aligned16(out), GRIT_E_BADARG, "grit_embed_gather_f32: pointers must be 16-byte aligned");
  const int HC = H / 8;
  hipLaunchKernelGGL(embed_gather_f32_k, dim3(grid_for(T * HC, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)table, ids,
                     (float4*)out, T, HC, V);
  GRIT_CHECK_LAUNCH("grit_embed_gather_f32");
  return GRIT_OK;
}

int grit_rmsnorm_fwd_f32in(const float* x, const void* w, void* y, int64_t T, int H, float eps, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(x && w && y, GRIT_E_BADARG, "grit_rmsnorm_fwd_f32in: null pointer");
  GRIT_REQUIRE(T >= 0 && H > 0, GRIT_E_BADARG, "grit_rmsnorm_fwd_f32in: bad sizes");
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_rmsnorm_fwd_f32in: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), GRIT_E_BADARG, "grit_rmsnorm_fwd_f32in: pointers must be 16-byte aligned");
  const dim3 grid((unsigned)((T + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const float4* xp = (const float4*)x;
  const uint2* wp = (const uint2*)w;
  uint2* yp = (uint2*)y;
  switch (H % 256 == 0 ? H / 256 : 0) {
    case 1: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<1, false>), grid, block, 0, st, xp, wp, yp, T, H, eps, (unsigned int*)nullptr); break;
    case 2: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<2, false>), grid, block, 0, st, xp, wp, yp, T, H, eps, (unsigned int*)nullptr); break;
    case 4: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<4, false>), grid, block, 0, st, xp, wp, yp, T, H, eps, (unsigned int*)nullptr); break;
    case 8: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<8, false>), grid, block, 0, st, xp, wp, yp, T, H, eps, (unsigned int*)nullptr); break;
    case 16: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<16, false>), grid, block, 0, st, xp, wp, yp, T, H, eps, (unsigned int*)nullptr); break;
    default: hipLaunchKernelGGL(rmsnorm_fwd_f32in_generic_k<false>, grid, block, 0, st, xp, wp, yp, T, H, eps, (unsigned int*)nullptr); break;
  }
  GRIT_CHECK_LAUNCH("grit_rmsnorm_fwd_f32in");
  return GRIT_OK;
}

/* fp16-operand policy: the same RMSNorm of the fp32 residual stream, y rounded once to IEEE fp16 (w stays bf16) */
int grit_rmsnorm_fwd_f32in_f16(const float* x, const void* w, void* y, int64_t T, int H, float eps, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(x && w && y, GRIT_E_BADARG, "grit_rmsnorm_fwd_f32in_f16: null pointer");
  GRIT_REQUIRE(T >= 0 && H > 0, GRIT_E_BADARG, "grit_rmsnorm_fwd_f32in_f16: bad sizes");
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_rmsnorm_fwd_f32in_f16: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), GRIT_E_BADARG, "grit_rmsnorm_fwd_f32in_f16: pointers must be 16-byte aligned");
  unsigned int* ovf = f16_flag_ptr();
  GRIT_REQUIRE(ovf != nullptr, GRIT_E_LAUNCH, "grit_rmsnorm_fwd_f32in_f16: the overflow flag word of this device is not reachable");
  const dim3 grid((unsigned)((T + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const float4* xp = (const float4*)x;
  const uint2* wp = (const uint2*)w;
  uint2* yp = (uint2*)y;
  switch (H % 256 == 0 ? H / 256 : 0) {
    case 1: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<1, true>), grid, block, 0, st, xp, wp, yp, T, H, eps, ovf); break;
    case 2: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<2, true>), grid, block, 0, st, xp, wp, yp, T, H, eps, ovf); break;
    case 4: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<4, true>), grid, block, 0, st, xp, wp, yp, T, H, eps, ovf); break;
    case 8: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<8, true>), grid, block, 0, st, xp, wp, yp, T, H, eps, ovf); break;
    case 16: hipLaunchKernelGGL((rmsnorm_fwd_f32in_reg_k<16, true>), grid, block, 0, st, xp, wp, yp, T, H, eps, ovf); break;
    default: hipLaunchKernelGGL(rmsnorm_fwd_f32in_generic_k<true>, grid, block, 0, st, xp, wp, yp, T, H, eps, ovf); break;
  }
  GRIT_CHECK_LAUNCH("grit_rmsnorm_fwd_f32in_f16");
  return GRIT_OK;
}

/* fp16 residual stream (the "f16_stream" policy): MistralRMSNorm (:84-89) of an fp16 row, ONE rounding; y fp16 (out_is_f16 != 0: the next
 * GEMM's operand) or bf16 (last_hidden_state for the pooling kernels); w bf16 */
int grit_rmsnorm_fwd_f16in(const void* x, const void* w, void* y, int out_is_f16, int64_t T, int H, float eps, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(x && w && y, GRIT_E_BADARG, "grit_rmsnorm_fwd_f16in: null pointer");
  GRIT_REQUIRE(T >= 0 && H > 0, GRIT_E_BADARG, "grit_rmsnorm_fwd_f16in: bad sizes");
  GRIT_REQUIRE(H % 8 == 0, GRIT_E_UNSUPPORTED, "grit_rmsnorm_fwd_f16in: H=%d must be a multiple of 8", H);
  GRIT_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), GRIT_E_BADARG, "grit_rmsnorm_fwd_f16in: pointers must be 16-byte aligned");
  unsigned int* ovf = f16_flag_ptr();
  GRIT_REQUIRE(ovf != nullptr, GRIT_E_LAUNCH, "grit_rmsnorm_fwd_f16in: the overflow flag word of this device is not reachable");
  const dim3 grid((unsigned)((T + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const uint4* xp = (const uint4*)x;
  const uint4* wp = (const uint4*)w;
  uint4* yp = (uint4*)y;
#define GRIT_H16(NCH_)                                                                                                     \
  do {                                                                                                                     \
    if (out_is_f16) hipLaunchKernelGGL((rmsnorm_fwd_h16_k<NCH_, true>), grid, block, 0, st, xp, wp, yp, T, H, eps, ovf);   \
    else hipLaunchKernelGGL((rmsnorm_fwd_h16_k<NCH_, false>), grid, block, 0, st, xp, wp, yp, T, H, eps, ovf);             \
  } while (0)
  switch (H % 512 == 0 ? H / 512 : 0) {
    case 1: GRIT_H16(1); break;
    case 2: GRIT_H16(2); break;
    case 4: GRIT_H16(4); break;
    case 8: GRIT_H16(8); break;
    case 16: GRIT_H16(16); break;
    default: GRIT_H16(0); break;
  }
#undef GRIT_H16
  GRIT_CHECK_LAUNCH("grit_rmsnorm_fwd_f16in");
  return GRIT_OK;
}

/* The current device's fp16 overflow flag: *host_flag = 1 when a kernel of the f16_operands policy has stored an inf / nan since the last
 * clear, else 0.  Runs on `stream` and WAITS for it (one 4-byte D2H copy); clear != 0 resets the flag behind the read. */
int grit_f16_overflow_flag(int* host_flag, int clear, void* stream) {
  GRIT_REQUIRE(host_flag != nullptr, GRIT_E_BADARG, "grit_f16_overflow_flag: null pointer");
  unsigned int* ovf = f16_flag_ptr();
  GRIT_REQUIRE(ovf != nullptr, GRIT_E_LAUNCH, "grit_f16_overflow_flag: the overflow flag word of this device is not reachable");
  hipStream_t st = (hipStream_t)stream;
  unsigned int v = 0;
  hipError_t e = hipMemcpyAsync(&v, ovf, sizeof(v), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && clear) e = hipMemsetAsync(ovf, 0, sizeof(v), st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    set_error("grit_f16_overflow_flag: %s", hipGetErrorString(e));
    return GRIT_E_LAUNCH;
  }
  *host_flag = v != 0 ? 1 : 0;
  return GRIT_OK;
}

static int rope_launch(void* qkv, const float* cos_tab, const float* sin_tab, const int32_t* positions, int64_t T, int S, int nq, int nkv,
                       int d, int64_t row_stride, int inverse, void* stream);

int grit_rope_qk_inplace(void* qkv, const float* cos_tab, const float* sin_tab, int64_t T, int S, int nq, int nkv, int d,
                         int64_t row_stride, int inverse, void* stream) {
  return rope_launch(qkv, cos_tab, sin_tab, nullptr, T, S, nq, nkv, d, row_stride, inverse, stream);
}

int grit_rope_qk_inplace_pos(void* qkv, const float* cos_tab, const float* sin_tab, const int32_t* positions, int64_t T, int table_rows,
                             int nq, int nkv, int d, int64_t row_stride, int inverse, void* stream) {
  GRIT_REQUIRE(positions, GRIT_E_BADARG, "grit_rope_qk_inplace_pos: null positions");
  return rope_launch(qkv, cos_tab, sin_tab, positions, T, table_rows, nq, nkv, d, row_stride, inverse, stream);
}

static int rope_launch(void* qkv, const float* cos_tab, const float* sin_tab, const int32_t* positions, int64_t T, int S, int nq, int nkv,
                       int d, int64_t row_stride, int inverse, void* stream) {
  if (T == 0) return GRIT_OK;
  GRIT_REQUIRE(qkv && cos_tab && sin_tab, GRIT_E_BADARG, "grit_rope_qk_inplace: null pointer");
  GRIT_REQUIRE(T >= 0 && S > 0 && nq > 0 && nkv >= 0 && d > 0, GRIT_E_BADARG, "grit_rope_qk_inplace: bad sizes");
  GRIT_REQUIRE(d % 16 == 0, GRIT_E_UNSUPPORTED, "grit_rope_qk_inplace: head_dim=%d must be a multiple of 16", d);
  GRIT_REQUIRE(row_stride % 8 == 0 && row_stride >= ((int64_t)nq + nkv) * d, GRIT_E_BADARG, "grit_rope_qk_inplace: bad row_stride");
  GRIT_REQUIRE((int64_t)nq + nkv <= INT32_MAX, GRIT_E_BADARG, "grit_rope_qk_inplace: bad sizes");
  GRIT_REQUIRE(aligned16(qkv) && aligned16(cos_tab) && aligned16(sin_tab), GRIT_E_BADARG, "grit_rope_qk_inplace: pointers must be 16-byte aligned");
  if (T == 0) return GRIT_OK;
  const int nheads = nq + nkv;
  const int64_t items = T * nheads * (d / 16);
  hipLaunchKernelGGL(rope_k, dim3(grid_for(items, 256)), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkv,
                     (const float4*)cos_tab, (const float4*)sin_tab, positions, T, S, nheads, d, row_stride, inverse ? -1.0f : 1.0f);
  GRIT_CHECK_LAUNCH("grit_rope_qk_inplace");
  return GRIT_OK;
}

int grit_mask_pack(const int64_t* mask, uint64_t* bits, int B, int S, void* stream) {
  GRIT_REQUIRE(mask && bits, GRIT_E_BADARG, "grit_mask_pack: null pointer");
  GRIT_REQUIRE(B > 0 && S > 0 && S <= (1 << 30), GRIT_E_BADARG, "grit_mask_pack: bad sizes");
  const int W = (S + 63) / 64;
  const int64_t words = (int64_t)B * W;
  hipLaunchKernelGGL(mask_pack_k, dim3((unsigned)((words + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mask, bits, B, S, W);
  GRIT_CHECK_LAUNCH("grit_mask_pack");
  return GRIT_OK;
}

int grit_transpose_bf16(const void* in, void* out, int64_t R, int64_t C, int64_t ld_in, int64_t ld_out, void* stream) {
  GRIT_REQUIRE(in && out, GRIT_E_BADARG, "grit_transpose_bf16: null pointer");
  GRIT_REQUIRE(R > 0 && C > 0, GRIT_E_BADARG, "grit_transpose_bf16: bad sizes");
  GRIT_REQUIRE(C % 8 == 0, GRIT_E_UNSUPPORTED, "grit_transpose_bf16: C=%lld must be a multiple of 8", (long long)C);
  // output rows are written in 16-byte groups of 8 source rows: the tail group is zero-filled up to the next multiple of 8
  GRIT_REQUIRE(ld_in >= C && ld_out >= (R + 7) / 8 * 8 && ld_in % 8 == 0 && ld_out % 8 == 0, GRIT_E_BADARG,
               "grit_transpose_bf16: bad leading dimensions (ld_out must cover R rounded up to 8)");
  GRIT_REQUIRE(aligned16(in) && aligned16(out), GRIT_E_BADARG, "grit_transpose_bf16: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(transpose_k, dim3((unsigned)((C + TR_COLS - 1) / TR_COLS), (unsigned)((R + TR_ROWS - 1) / TR_ROWS)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)in, (uint16_t*)out, R, C, ld_in, ld_out);
  GRIT_CHECK_LAUNCH("grit_transpose_bf16");
  return GRIT_OK;
}

}  // extern "C"
