// What the bf16 matrix pipe of THIS board sustains: a register-only v_mfma_f32_16x16x32_bf16 loop (no LDS, no memory) on
// every CU for a few seconds, timed with HIP events, while the caller samples rocm-smi (tools/mfma_peak.sh).  The nominal
// dense peak (2.5 PFLOP/s) is 2.4 GHz x 256 CUs x 4 SIMDs x 1024 FLOP/cycle; under the 1.4 kW package cap an MFMA-dense
// loop on random operands holds a lower clock -- this is the measured ceiling the conv kernels' fractions can be read against.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/mfma_peak.hip
//   gpurun_out/mfma_peak <seconds> <waves per SIMD: 1|2|4> <data: 0 zeros | 1 random | 2 random with one operand half zeros | 3 random, no operand shared by consecutive MFMAs> <shape: 0 16x16x32 | 1 32x32x16>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void mfma_loop(float* out, long iters, int random) {
  bf16x8 a[4], b[4];
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // random sign and mantissa, magnitudes around 2^-3 (the sums stay finite over any run length that matters)
      const unsigned ha = hash32(id * 64u + i * 8u + e), hb = hash32(id * 64u + 32u + i * 8u + e);
      const unsigned short ua = random ? (unsigned short)((ha & 0x807fu) | (0x7cu << 7)) : 0;
      unsigned short ub = random ? (unsigned short)((hb & 0x807fu) | (0x7cu << 7)) : 0;
      if (random == 2 && (hb & 0x10000u)) ub = 0;  // half of one operand's elements are zero (what a ReLU leaves)
      a[i][e] = __builtin_bit_cast(__bf16, ua);
      b[i][e] = __builtin_bit_cast(__bf16, ub);
    }
  f32x4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  if (random == 3) {  // diagonal order: consecutive MFMAs share neither operand
    for (long k = 0; k < iters; ++k) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][(i + d) & 3]) : "v"(a[i]), "v"(b[(i + d) & 3]));
    }
  } else {
    for (long k = 0; k < iters; ++k) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)  // (inline asm: the intrinsic form was compiled with accumulator copies between the MFMAs)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
    }
  }
  f32x4v s = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[id] = s[0] + s[1] + s[2] + s[3];
}

typedef float f32x16v __attribute__((ext_vector_type(16)));
// the same loop on v_mfma_f32_32x32x16_bf16 (same FLOP per fragment pair and per cycle; a different operand broadcast inside the array)
__global__ __launch_bounds__(256) void mfma_loop_32(float* out, long iters, int random) {
  bf16x8 a[2], b[2];
  const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned ha = hash32(id * 64u + i * 8u + e), hb = hash32(id * 64u + 32u + i * 8u + e);
      a[i][e] = __builtin_bit_cast(__bf16, random ? (unsigned short)((ha & 0x807fu) | (0x7cu << 7)) : (unsigned short)0);
      b[i][e] = __builtin_bit_cast(__bf16, random ? (unsigned short)((hb & 0x807fu) | (0x7cu << 7)) : (unsigned short)0);
    }
  f32x16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (long k = 0; k < iters; ++k) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[id] = s;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int wps = argc > 2 ? atoi(argv[2]) : 2;
  const int random = argc > 3 ? atoi(argv[3]) : 1;
  const int shape32 = argc > 4 ? atoi(argv[4]) : 0;
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
  float* out;
  (void)hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  long iters = 200000;
  double tf = 0.0;
  for (int pass = 0; pass < 2; ++pass) {  // pass 0 calibrates the run length
    (void)hipEventRecord(e0);
    if (shape32)
      hipLaunchKernelGGL(mfma_loop_32, dim3(blocks), dim3(256), 0, 0, out, iters, random);
    else
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, random);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16 * 16384.0;
    tf = flop / (ms * 1e-3) * 1e-12;
    if (pass == 0) iters = (long)(iters * seconds * 1e3 / ms);
    else
      printf("{\"mfma\": \"%s\", \"cus\": %d, \"waves_per_simd\": %d, \"data\": \"%s\", \"seconds\": %.2f, \"bf16_tflops\": %.1f, "
             "\"frac_of_2500\": %.3f, \"implied_mhz_at_full_issue\": %.0f}\n", shape32 ? "32x32x16" : "16x16x32", cus, wps, random == 0 ? "zeros" : random == 1 ? "random" : random == 2 ? "random, one operand half zeros" : "random, diagonal order", ms * 1e-3, tf,
             tf / 2500.0, tf * 1e12 / (cus * 4 * 1024.0) * 1e-6);
  }
  return 0;
}
