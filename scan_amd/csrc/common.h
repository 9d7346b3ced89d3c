// Shared helpers for the gfx950 kernels of libscan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/scan_hip.h"

#define SCAN_WAVE 64

extern "C" void scan_set_error(const char* fmt, ...);

#define SCAN_CHECK_ARG(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      scan_set_error(__VA_ARGS__);         \
      return -1;                           \
    }                                      \
  } while (0)

#define SCAN_LAUNCH_CHECK(name)                                          \
  do {                                                                   \
    hipError_t _e = hipGetLastError();                                   \
    if (_e != hipSuccess) {                                              \
      scan_set_error("%s: launch failed: %s", name, hipGetErrorString(_e)); \
      return -2;                                                         \
    }                                                                    \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// memory-bound kernels: cap the grid at 256 CUs x 8 blocks and grid-stride the rest
static inline int grid_for(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) smem4[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = smem4[0] + smem4[1] + smem4[2] + smem4[3];
  __syncthreads();
  return r;
}

// pyramid row decode: m -> (level, image, y, x)
struct RowCoord {
  int lvl, n, y, x;
};
__device__ __forceinline__ RowCoord decode_row(const scan_pyramid_t& d, int64_t m) {
  RowCoord rc;
  int l = 0;
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
    if (i < d.n_levels && m >= d.row_off[i]) l = i;
  int64_t r = m - d.row_off[l];
  const int hw = d.h[l] * d.w[l];
  rc.lvl = l;
  rc.n = (int)(r / hw);
  int rem = (int)(r - (int64_t)rc.n * hw);
  rc.y = rem / d.w[l];
  rc.x = rem - rc.y * d.w[l];
  return rc;
}
