// What does ds_read_b64_tr_b16 return?  LDS holds u16 value = its own element index; every lane reads 8 bytes at `addr(lane)`.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = lane * 8;                                   // contiguous 8 B per lane
  else if (mode == 1) addr = (lane & 15) * 64 + (lane >> 4) * 8;    // 16 rows of 64 B (32 elements), lane group g reads 8 B at column block g
  else addr = (lane & 15) * 32 + (lane >> 4) * 8;                   // 16 rows of 32 B (16 elements)
  uint32_t base = (uint32_t)(uintptr_t)lds + addr;
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base) : "memory");
  out[lane * 4 + 0] = v.x & 0xffff; out[lane * 4 + 1] = v.x >> 16; out[lane * 4 + 2] = v.y & 0xffff; out[lane * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("L%02d:%4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l % 4 == 3) ? "\n" : " | "); }
  }
  return 0;
}
