// semantic check of __builtin_amdgcn_global_load_lds (16-byte) on gfx950: LDS destination = wave base + lane * 16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__global__ void k(const uint4* __restrict__ src, uint4* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // lane l of wave w fetches source element w*64 + (l ^ 5); destination slot (linear) w*64 + l
  const uint4* g = src + wid * 64 + (lane ^ 5);
  unsigned char* base = smem + (size_t)wid * 1024;  // wave-uniform
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)base, 16, 0, 0);
  __syncthreads();
  out[tid] = reinterpret_cast<const uint4*>(smem)[tid];
}
int main() {
  const int n = 256;
  uint4 h[n], r[n];
  for (int i = 0; i < n; ++i) h[i] = make_uint4(i, i * 3 + 1, 7 * i, 0xabc00000u + i);
  uint4 *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 4096, 0, d, o);
  hipMemcpy(r, o, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const int s = (i / 64) * 64 + ((i & 63) ^ 5);
    if (r[i].x != h[s].x || r[i].y != h[s].y || r[i].z != h[s].z || r[i].w != h[s].w) ++bad;
  }
  printf("glds 16-byte: %d mismatches of %d\n", bad, n);
  return bad != 0;
}
