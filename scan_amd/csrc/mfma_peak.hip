// Measurement entry point (no counterpart in the reference): the bf16 matrix-pipe rate THIS board sustains -- a register-only
// v_mfma_f32_16x16x32_bf16 loop (no LDS, no memory) on every CU.  The nominal dense peak bench.py prices the conv kernels
// against (2.5 PFLOP/s = 2.4 GHz x 256 CUs x 4 SIMDs x 1024 FLOP/cycle, MI355X_MICROARCH.md) is not reachable under the
// 1.4 kW package cap with random operands: the loop below holds 2.0-2.2 GHz at the cap (tools/mfma_peak.sh samples rocm-smi
// beside it; zeros as operands run at 2.4 GHz and half the power).  bench.py reports the figure as
// roofline.board_sustained next to the nominal fraction.
#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void mfma_sustained_kernel(float* out, long long iters, int random) {
  bf16x8 a[4], b[4];
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // random sign and mantissa at magnitude 2^-3: the sums stay finite over any run length that matters
      const unsigned ha = hash32(id * 64u + i * 8u + e), hb = hash32(id * 64u + 32u + i * 8u + e);
      a[i][e] = __builtin_bit_cast(__bf16, random ? (unsigned short)((ha & 0x807fu) | (0x7cu << 7)) : (unsigned short)0);
      b[i][e] = __builtin_bit_cast(__bf16, random ? (unsigned short)((hb & 0x807fu) | (0x7cu << 7)) : (unsigned short)0);
    }
  f32x4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  for (long long k = 0; k < iters; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)  // (inline asm: the intrinsic form compiles with accumulator copies between the MFMAs)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  f32x4v s = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[id] = s[0] + s[1] + s[2] + s[3];
}

// Runs the loop for about `seconds` (after a short calibration launch) with two waves per SIMD on the current device and
// writes the measured rate in TFLOP/s.  Blocking: synchronises `stream`.  random != 0: random-sign, random-mantissa operands
// (what a convolution multiplies); 0: zeros (the pipe's issue limit without the power cap).
extern "C" int scan_mfma_sustained_bf16(double seconds, int32_t random, double* tflops, void* stream) {
  SCAN_CHECK_ARG(tflops != nullptr && seconds > 0.0 && seconds <= 30.0, "mfma_sustained_bf16: seconds in (0, 30], tflops != NULL");
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
    scan_set_error("mfma_sustained_bf16: no device");
    return -2;
  }
  const int blocks = p.multiProcessorCount * 2;  // 256 threads = one wave per SIMD; two blocks per CU
  hipStream_t st = as_stream(stream);
  float* out = nullptr;
  if (hipMalloc(&out, sizeof(float) * blocks * 256) != hipSuccess) {
    scan_set_error("mfma_sustained_bf16: hipMalloc failed");
    return -2;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  long long iters = 100000;
  float ms = 0.f;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 calibrates the run length
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_sustained_kernel, dim3(blocks), dim3(256), 0, st, out, iters, (int)random);
    hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) {
      hipFree(out);
      scan_set_error("mfma_sustained_bf16: timing failed");
      return -2;
    }
    if (pass == 0) iters = (long long)((double)iters * seconds * 1e3 / ms) + 1;
  }
  *tflops = (double)blocks * 4.0 * (double)iters * 16.0 * 16384.0 / ((double)ms * 1e-3) * 1e-12;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(out);
  return 0;
}
