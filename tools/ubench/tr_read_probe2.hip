#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
constexpr int V_PITCH = 320;
__global__ void k(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) char smem[16384 + 64 * V_PITCH];
  char* v_lds = smem + 16384;
  for (int i = threadIdx.x; i < 64 * 128; i += 64) { int key = i / 128, d = i % 128; *(uint16_t*)(v_lds + key * V_PITCH + d * 2) = (uint16_t)(key * 128 + d); }
  __syncthreads();
  const int lane = threadIdx.x, hi = lane >> 5;
  const int vt_lane = (((lane & 15) >> 2) + 4 * hi) * V_PITCH + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2;
  const int db = 1, kb = 1, c = 0;
  const char* vp = v_lds + vt_lane + (kb * 32 + c * 16) * V_PITCH + db * 64;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp));
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 8 * V_PITCH));
  for (int e = 0; e < 4; ++e) { out[lane * 8 + e] = v0[e]; out[lane * 8 + 4 + e] = v1[e]; }
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 8 * 2); uint16_t h[512];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
    int hi = l >> 5, key = 32 + 0 + 4 * hi + (e & 3) + 8 * (e >> 2), dd = 32 + (l & 31);
    int want = key * 128 + dd;
    if (h[l * 8 + e] != want) { if (bad < 8) printf("lane %d e %d got (key %d d %d) want (key %d d %d)\n", l, e, h[l*8+e] / 128, h[l*8+e] % 128, key, dd); ++bad; }
  }
  printf("mismatches %d\n", bad);
  return 0;
}
