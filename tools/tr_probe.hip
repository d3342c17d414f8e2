#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  int q = (l & 15) >> 2, p = l & 3, g = l >> 4;
  short* addr = sm + (q * 64) + g * 16 + 4 * p;   // row q (64 shorts per row), cols g*16 + 4p..
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}
int main() {
  short* d; hipMalloc(&d, 64*4*2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d: ", l); for (int i=0;i<4;++i) printf("(r%d,c%2d) ", h[l*4+i]/64, h[l*4+i]%64); printf("\n"); }
  return 0;
}
