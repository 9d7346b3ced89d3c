// fp32 convolutions on the bf16 matrix cores by operand splitting -- what the conv kernels share.
//
// An fp32 value is cut into NP bf16 PIECES of decreasing magnitude, x = p0 + p1 (+ p2), each the round-to-nearest bf16 of
// what the earlier pieces left over (every subtraction is exact in fp32).  Two pieces carry 16 significand bits; three
// pieces carry all 24: the three-piece split of a normal fp32 number is EXACT (8 + 8 + 8 bits, the signs of the
// residuals absorb the carries).  A product x * w is accumulated in fp32 from the piece products x_i * w_j with
// i + j <= NP - 1, smallest terms first:
//     NP = 2 ("bf16x3"):  x0 w1, x1 w0, x0 w0                              3 bf16 MFMAs, dropped terms ~2^-16 of the product
//     NP = 3 ("bf16x6"):  x0 w2, x1 w1, x2 w0, x0 w1, x1 w0, x0 w0         6 bf16 MFMAs, dropped terms (x1 w2, x2 w1, x2 w2)
//                                                                          <= 2^-23 of the product, i.e. below the rounding
//                                                                          of the fp32 accumulation itself
// bf16 x bf16 products are exact in fp32, so bf16x6 is the reference's arithmetic (fp32 multiply, fp32 accumulate:
// torch.nn.Conv2d, fcos_core/modeling/backbone/mmdetection/vgg.py:8-33) up to the order of summation, on a pipe whose
// dense peak is 16x that of v_mfma_f32_32x32x2_f32: ceiling 2.5 PFLOP/s / 6 = 417 TFLOP/s fp32-equivalent against 157.
#pragma once
#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// pieces of four values: q[0] = hi ... q[NP - 1] = lo.  Written on PAIRS so that each step is one packed instruction:
// v_cvt_pk_bf16_f32 rounds and packs two values, the two floats behind a packed pair are a shift and a mask of its dword,
// the residual is one v_pk_add_f32 -- 8 vector instructions per float4 and residual stage instead of the 11-12 the
// element-wise form compiled to (the split runs beside the MFMA stream and every vector instruction of it takes an issue
// slot from the matrix pipe: profiles/r04_wgrad_exp.txt).
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NP>
__device__ __forceinline__ void split4_np(const float4 v, bf16x4 (&q)[NP]) {
  f32x2v r0 = {v.x, v.y}, r1 = {v.z, v.w};
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bf16x2 b0 = __builtin_convertvector(r0, bf16x2), b1 = __builtin_convertvector(r1, bf16x2);
    q[p][0] = b0[0];
    q[p][1] = b0[1];
    q[p][2] = b1[0];
    q[p][3] = b1[1];
    if (p + 1 < NP) {
      const unsigned u0 = __builtin_bit_cast(unsigned, b0), u1 = __builtin_bit_cast(unsigned, b1);
      const f32x2v f0 = {__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xffff0000u)};
      const f32x2v f1 = {__builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xffff0000u)};
      r0 -= f0;
      r1 -= f1;
    }
  }
}

template <int NP>
__device__ __forceinline__ void split1_np(float v, __bf16 (&q)[NP]) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    q[p] = (__bf16)v;
    if (p + 1 < NP) v -= (float)q[p];
  }
}

// buffer descriptor from wave-uniform inputs, made PROVABLY uniform for the compiler (cdna_hip_programming.md T20): a
// descriptor it cannot prove uniform gets a waterfall loop around every load
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc_b(const void* base, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// workgroup index -> position in a list that keeps consecutive positions on ONE XCD (blockIdx % 8 is the XCD)
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / 8;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
