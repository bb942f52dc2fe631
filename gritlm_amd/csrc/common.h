// Shared device/host helpers for libgritlm_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gritlm_hip.h"

namespace grit {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

void set_error(const char* fmt, ...);

#define GRIT_REQUIRE(cond, code, ...)       \
  do {                                      \
    if (!(cond)) {                          \
      ::grit::set_error(__VA_ARGS__);       \
      return (code);                        \
    }                                       \
  } while (0)

#define GRIT_CHECK_LAUNCH(name)                                                 \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      ::grit::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return GRIT_E_LAUNCH;                                                     \
    }                                                                           \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- bf16 <-> f32 (bit tricks; RNE like torch's .to(bfloat16)) ----
__device__ __forceinline__ float bf2f(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }
// hardware pack (v_cvt_pk_bf16_f32, RNE): one instruction for two conversions.  Written as a vector conversion, NOT as inline asm: the
// compiler then knows it is a VALU write and inserts the wait states an MFMA that reads the result as SrcA/B needs -- with an opaque asm
// statement directly in front of the MFMA, part of the lanes saw stale operands (found with the tr-read attention variant, NOTEBOOK.md §8)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_pk_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_pk_t;
__device__ __forceinline__ uint32_t pack2bf_hw(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_pk_t){lo, hi}, bf16x2_pk_t));
}
__device__ __forceinline__ float round_bf(float f) { return __uint_as_float(f2bf(f) << 16); }

// ---- fp16 operands (the "f16_operands" precision policy of the encoder: same MFMA rate as bf16, 3 more mantissa bits) ----
// v_cvt_pk_f16_f32 (RNE, gfx950) for two conversions; values beyond 65504 become inf -- every kernel that rounds to f16 ORs
// h2_nonfinite() of what it stores into a per-device flag word (f16_flag_ptr()) that the host reads after the forward pass.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_pk_t;
__device__ __forceinline__ uint32_t pack2h_hw(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_pk_t){lo, hi}, f16x2_pk_t));
}
__device__ __forceinline__ float hlo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_pk_t, w)[0]; }
__device__ __forceinline__ float hhi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_pk_t, w)[1]; }
// non-zero iff one of the two packed halves is inf / nan (exponent field all ones): strip the signs, add 1 to the exponent's top bit
__device__ __forceinline__ uint32_t h2_nonfinite(uint32_t pk) { return ((pk & 0x7fff7fffu) + 0x04000400u) & 0x80008000u; }
// operand-format switch of the templated kernels: F16 = false -> bf16 (the reference's arithmetic type), true -> fp16
template <bool F16>
__device__ __forceinline__ uint32_t pack2_op(float lo, float hi) {
  if constexpr (F16) return pack2h_hw(lo, hi);
  else return pack2bf_hw(lo, hi);
}
// packed fp16 add (v_pk_add_f16): the correctly rounded fp16 sums of two pairs of fp16 values
__device__ __forceinline__ uint32_t pk_add_f16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_bit_cast(f16x2_pk_t, a) + __builtin_bit_cast(f16x2_pk_t, b));
}
// 16-bit unpack in the operand format (e.g. the residual reads of the GEMM's RESIDUAL epilogue: an fp16 residual stream under F16)
template <bool F16>
__device__ __forceinline__ float lo16_op(uint32_t w) {
  if constexpr (F16) return hlo(w);
  else return bflo(w);
}
template <bool F16>
__device__ __forceinline__ float hi16_op(uint32_t w) {
  if constexpr (F16) return hhi(w);
  else return bfhi(w);
}
// host: device address of this device's f16 overflow flag word (elementwise.hip; nullptr if the symbol lookup fails)
unsigned int* f16_flag_ptr();

// Rotary embedding arithmetic with a FIXED contraction (one rounded product + one fma), so every kernel that rotates -- rope_k, the QKV
// GEMM's epilogue in both launch forms, the decode kernels -- produces the same bits whatever the surrounding code makes the compiler
// prefer:  lo = x1*cos - x2*sin,  hi = x2*cos + x1*sin   (modeling_mistral_gritlm.py:138-163, x*cos + rotate_half(x)*sin)
__device__ __forceinline__ float rope_lo(float x1, float x2, float c, float s) {
#pragma clang fp contract(off)
  const float t = x2 * s;
  return __builtin_fmaf(x1, c, -t);
}
__device__ __forceinline__ float rope_hi(float x1, float x2, float c, float s) {
#pragma clang fp contract(off)
  const float t = x1 * s;
  return __builtin_fmaf(x2, c, t);
}

// silu(x) = x * sigmoid(x) for every kernel that evaluates the MLP activation (GEMM epilogues, the stand-alone SwiGLU kernel, the decode
// GEMV): ONE definition, so the fused and un-fused paths the checks compare stay bit-identical.  Correctly rounded division: the
// v_rcp_f32 form (-DGRIT_SILU_RCP) moved 2 of 1.9 G activations by one bf16 ulp and measured 0.98x on the gate|up GEMM (within the
// run-to-run noise of that launch, profiles/r03_gemm_ab_opaque_zero.log), so the forward arithmetic stays what round 1 pinned.
__device__ __forceinline__ float silu_f(float x) {
#pragma clang fp contract(off)
#ifdef GRIT_SILU_RCP
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#else
  return x / (1.0f + __expf(-x));
#endif
}

// SwiGLU backward of one element on bf16-rounded operands: d_gate = d * u * s * (1 + g (1 - s)), d_up = d * g * s, s = sigmoid(g).
// ONE definition with a fixed contraction for the stand-alone kernel (backward.hip) and both GEMM epilogues (gemm_bf16.hip), which the
// checks compare bit for bit.  The sigmoid's reciprocal is v_rcp_f32 (1 ulp) instead of the correctly rounded division (~10 VALU
// instructions per element in an epilogue that handles 128 elements per lane and tile): 1 bf16 output in 1.7 million moves by one ulp,
// the d_act GEMM gains 4-6 % (profiles/r03_gemm_ab_swiglu_bwd_epilogue.log; -DGRIT_SWIGLU_BWD_DIV restores the division for A/B builds).
__device__ __forceinline__ void swiglu_bwd_elem(float d, float g, float u, float& d_gate, float& d_up) {
#pragma clang fp contract(off)
#ifdef GRIT_SWIGLU_BWD_DIV
  const float s = 1.0f / (1.0f + __expf(-g));
#else
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
#endif
  const float t = g * (1.f - s);
  d_gate = d * u * s * (1.f + t);
  d_up = d * g * s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace grit
