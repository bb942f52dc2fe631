// Does ds_read_b64_tr_b16 write anything besides its two destination VGPRs?  Sentinels in v[40:47], tr read into v[42:43].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  uint32_t addr = (uint32_t)(uintptr_t)lds + (((lane & 15) >> 2) * 320 + (((lane >> 4) & 1) * 16 + 4 * (lane & 3)) * 2);
  uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
  asm volatile(
      "v_mov_b32 v40, 0x11111111\n v_mov_b32 v41, 0x22222222\n v_mov_b32 v42, 0x33333333\n v_mov_b32 v43, 0x44444444\n"
      "v_mov_b32 v44, 0x55555555\n v_mov_b32 v45, 0x66666666\n v_mov_b32 v46, 0x77777777\n v_mov_b32 v47, 0x88888888\n"
      "ds_read_b64_tr_b16 v[42:43], %8\n s_waitcnt lgkmcnt(0)\n s_nop 7\n"
      "v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42\n v_mov_b32 %3, v43\n v_mov_b32 %4, v44\n v_mov_b32 %5, v45\n v_mov_b32 %6, v46\n v_mov_b32 %7, v47\n"
      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
      : "v"(addr)
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
  uint32_t* o = out + lane * 8;
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5; o[6] = r6; o[7] = r7;
}
int main() {
  uint32_t* d; hipMalloc(&d, 64 * 8 * 4); uint32_t h[512];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const uint32_t want[8] = {0x11111111, 0x22222222, 0, 0, 0x55555555, 0x66666666, 0x77777777, 0x88888888};
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 8; ++r) if (r != 2 && r != 3 && h[l * 8 + r] != want[r]) { if (bad < 10) printf("lane %d reg v%d clobbered: %08x\n", l, 40 + r, h[l * 8 + r]); ++bad; }
  printf("clobbered sentinels: %d\n", bad);
  return 0;
}
