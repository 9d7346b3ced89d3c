#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LROW 40
__global__ void k(short* out) {
  __shared__ short img[64 * LROW];
  for (int i = threadIdx.x; i < 64 * LROW; i += 64) img[i] = (short)((i / LROW) * 100 + (i % LROW));  // row*100+col
  __syncthreads();
  int l = threadIdx.x;
  int row = 8 * (l >> 5) + ((l & 15) >> 2);
  int col = 16 * ((l >> 4) & 1) + 4 * (l & 3);
  auto p = (__attribute__((address_space(3))) s16x4*)(img + row * LROW + col);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
