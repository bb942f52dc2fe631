// tr-read fragment -> MFMA: O^T[d][q] = sum over 16 keys of V[key][d] * 1 ; V[key][d] = d + key/64  -> expect 16 d + (sum key)/64, same for every q
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int V_PITCH = 320;
__device__ uint16_t f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) char smem[16384 + 64 * V_PITCH];
  char* v_lds = smem + 16384;
  for (int i = threadIdx.x; i < 64 * 128; i += 64) { int key = i / 128, d = i % 128; *(uint16_t*)(v_lds + key * V_PITCH + d * 2) = f2bf((float)d + (key == 3 ? 64.f : 0.f)); }
  __syncthreads();
  const int lane = threadIdx.x, hi = lane >> 5;
  const int vt_lane = (((lane & 15) >> 2) + 4 * hi) * V_PITCH + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;
  f32x16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const uint32_t one2 = 0x3F803F80u;
  const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, make_uint4(one2, one2, one2, one2));
  const char* vp = v_lds + vt_lane;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp));
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 8 * V_PITCH));
  const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, (__attribute__((ext_vector_type(8))) short){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]});
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4); float h[1024];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int dd = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);           // D row held in register r of lane l
    const float want = 16.f * dd + 64.f;                             // 16 keys (0..15), key 3 carries +64
    if (h[l * 16 + r] != want) { if (bad < 10) printf("lane %d (q %d) r %d (d %d): got %g want %g\n", l, l & 31, r, dd, h[l * 16 + r], want); ++bad; }
  }
  printf("mismatches %d\n", bad);
  return 0;
}
