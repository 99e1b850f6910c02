#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// what does ds_read_b64_tr_b16 return?  LDS holds u16 value = element index; lane l passes byte address addr[l]
__global__ void k(const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  uint32_t a = base + (uint32_t)addr[threadIdx.x];
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
  int h_addr[64]; uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
  for (int test = 0; test < 3; ++test) {
    for (int l = 0; l < 64; ++l) {
      if (test == 0) h_addr[l] = l * 8;                 // lane l -> 4 consecutive elements 4l..4l+3
      else if (test == 1) h_addr[l] = l * 128;          // lane l -> row l of a [64][64] u16 matrix (128 B rows), cols 0..3
      else h_addr[l] = (l & 15) * 128 + (l >> 4) * 8;   // 16 rows x 4 column groups
    }
    hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("test %d\n", test);
    for (int l = 0; l < 64; ++l) printf("  lane %2d addr %5d (elem %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return 0;
}
