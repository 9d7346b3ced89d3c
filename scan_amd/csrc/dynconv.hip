// Semantic-conditioned dynamic 1x1 convolution + channel softmax for gfx950.
// Replaces GRAPHModule.dynamic_conv + softmax(dim=1)
// (reference rpn/fcos/condgraph.py:619-629, 344-346, 509-510, 541-542).
//
// feat [M,256] x kernels[K,256]^T -> logits [M,K] -> softmax -> probs [M,K].
// HBM-bound (1 KiB read, 72 B written per pixel).  Forward contracts on the matrix
// cores with v_mfma_f32_16x16x4_f32 (pixels x classes tiles, exact fp32), so the
// 256-deep dot products need no cross-lane reduction; the softmax runs on the 16-lane
// class groups of the accumulator layout.  Backward is a streaming VALU kernel
// (rank-K updates) with a deterministic two-stage reduction for d_kernels.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DC_C 256

template <int K>
__global__ __launch_bounds__(256) void dynconv_fwd_kernel(const float* __restrict__ feat,
                                                          const float* __restrict__ kernels, int64_t M,
                                                          float* __restrict__ logits, float* __restrict__ probs) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, q = lane >> 4;  // A: row = col-index of lane, k-quarter q
  // B fragments: lane supplies kernels[col][16j + 4q + e] for MFMA (j, e)
  float4 bw[16];
#pragma unroll
  for (int j = 0; j < 16; ++j)
    bw[j] = (col < K) ? *reinterpret_cast<const float4*>(kernels + col * DC_C + 16 * j + 4 * q)
                      : make_float4(0, 0, 0, 0);
  const int64_t groups = (M + 15) / 16;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t gidx = wave_id; gidx < groups; gidx += n_waves) {
    const int64_t row = gidx * 16 + col;  // as A operand this lane feeds pixel row `col` of the tile
    float4 av[16];
    if (row < M) {
#pragma unroll
      for (int j = 0; j < 16; ++j) av[j] = *reinterpret_cast<const float4*>(feat + row * DC_C + 16 * j + 4 * q);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) av[j] = make_float4(0, 0, 0, 0);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bw[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bw[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bw[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bw[j].w, acc, 0, 0, 0);
    }
    // C/D map (16x16): class = lane & 15, pixel = 4*(lane>>4) + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float z = acc[r];
      float mx = (col < K) ? z : -INFINITY;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float e = (col < K) ? expf(z - mx) : 0.f;
      float den = e;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) den += __shfl_xor(den, off, 64);
      const int64_t orow = gidx * 16 + 4 * q + r;
      if (col < K && orow < M) {
        logits[orow * K + col] = z;
        probs[orow * K + col] = e / den;
      }
    }
  }
}

// backward: lane -> 4 channels, wave -> FOUR consecutive rows per iteration.  part[block][K][256] partial d_kernels.
// The K softmax-backward coefficients of a row are computed ONCE by K lanes (16-lane group r handles row r: three
// coalesced 36-byte segment loads per wave instead of 27 same-address loads per lane and row), reduced with four
// xor-shuffles and broadcast to the wave as scalars (v_readlane -> SGPR operands of the FMAs).  Four rows in flight
// per wave hide the HBM latency the one-row-per-iteration version was bound by (1.1 -> ? TB/s, see profiles/).
template <int K>
__global__ __launch_bounds__(256) void dynconv_bwd_kernel(const float* __restrict__ feat,
                                                          const float* __restrict__ kernels,
                                                          const float* __restrict__ probs,
                                                          const float* __restrict__ d_logits_in,
                                                          const float* __restrict__ d_probs, int64_t M,
                                                          float* __restrict__ d_feat, float* __restrict__ part) {
  __shared__ float red[4][K][DC_C];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int gr = lane >> 4, gk = lane & 15;
  float4 w[K], dwacc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    w[k] = *reinterpret_cast<const float4*>(kernels + k * DC_C + 4 * lane);
    dwacc[k] = make_float4(0, 0, 0, 0);
  }
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wid;
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int64_t chunks = (M + 3) >> 2;
  for (int64_t ch = wave_id; ch < chunks; ch += n_waves) {
    const int64_t row0 = ch << 2;
    float4 f[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      f[r] = (row0 + r < M) ? *reinterpret_cast<const float4*>(feat + (row0 + r) * DC_C + 4 * lane)
                            : make_float4(0, 0, 0, 0);
    const int64_t myrow = row0 + gr;
    const bool act = gk < K && myrow < M;
    const int64_t idx = myrow * K + gk;
    const float pl = (act && d_probs != nullptr) ? probs[idx] : 0.f;
    const float dp = (act && d_probs != nullptr) ? d_probs[idx] : 0.f;
    const float dl = (act && d_logits_in != nullptr) ? d_logits_in[idx] : 0.f;
    float dot = pl * dp;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) dot += __shfl_xor(dot, off, 64);
    const float dzv = dl + pl * (dp - dot);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float4 o = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float dz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dzv), r * 16 + k));
        o.x += dz * w[k].x;
        o.y += dz * w[k].y;
        o.z += dz * w[k].z;
        o.w += dz * w[k].w;
        dwacc[k].x += dz * f[r].x;
        dwacc[k].y += dz * f[r].y;
        dwacc[k].z += dz * f[r].z;
        dwacc[k].w += dz * f[r].w;
      }
      if (row0 + r < M) *reinterpret_cast<float4*>(d_feat + (row0 + r) * DC_C + 4 * lane) = o;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) *reinterpret_cast<float4*>(&red[wid][k][4 * lane]) = dwacc[k];
  __syncthreads();
  for (int i = threadIdx.x; i < K * DC_C; i += 256) {
    const int k = i / DC_C, c = i - k * DC_C;
    part[(int64_t)blockIdx.x * K * DC_C + i] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
  }
}

__global__ void dynconv_reduce_kernel(const float* __restrict__ part, int nb, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[(int64_t)b * n + i];
  out[i] = s;
}

static int dc_bwd_blocks(int64_t M) {
  int64_t b = (M + 63) / 64;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int64_t scan_dynconv_ws_floats(int64_t M, int32_t C, int32_t K) { return (int64_t)dc_bwd_blocks(M) * K * C; }

extern "C" int scan_dynconv_softmax_forward(const float* feat, const float* kernels, int64_t M, int32_t C, int32_t K,
                                            float* logits, float* probs, void* stream) {
  SCAN_CHECK_ARG(C == DC_C, "dynconv_softmax_forward: only C=256 is built (got %d)", C);
  SCAN_CHECK_ARG(K == 9 || K == 2, "dynconv_softmax_forward: only K in {2, 9} is built (got %d)", K);
  SCAN_CHECK_ARG(M >= 0, "dynconv_softmax_forward: bad M");
  if (M == 0) return 0;
  SCAN_CHECK_ARG(feat && kernels && logits && probs, "dynconv_softmax_forward: null pointer");
  const int64_t groups = (M + 15) / 16;
  int64_t blocks = (groups + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = as_stream(stream);
  if (K == 9)
    hipLaunchKernelGGL((dynconv_fwd_kernel<9>), dim3((int)blocks), dim3(256), 0, st, feat, kernels, M, logits, probs);
  else
    hipLaunchKernelGGL((dynconv_fwd_kernel<2>), dim3((int)blocks), dim3(256), 0, st, feat, kernels, M, logits, probs);
  SCAN_LAUNCH_CHECK("dynconv_fwd");
  return 0;
}

extern "C" int scan_dynconv_softmax_backward(const float* feat, const float* kernels, const float* probs,
                                             const float* d_logits_in, const float* d_probs, int64_t M, int32_t C,
                                             int32_t K, float* d_feat, float* d_kernels, float* ws, void* stream) {
  SCAN_CHECK_ARG(C == DC_C, "dynconv_softmax_backward: only C=256 is built (got %d)", C);
  SCAN_CHECK_ARG(K == 9 || K == 2, "dynconv_softmax_backward: only K in {2, 9} is built (got %d)", K);
  SCAN_CHECK_ARG(M > 0, "dynconv_softmax_backward: M must be positive");
  SCAN_CHECK_ARG(feat && kernels && probs && d_feat && d_kernels && ws, "dynconv_softmax_backward: null pointer");
  const int nb = dc_bwd_blocks(M);
  hipStream_t st = as_stream(stream);
  if (K == 9)
    hipLaunchKernelGGL((dynconv_bwd_kernel<9>), dim3(nb), dim3(256), 0, st, feat, kernels, probs, d_logits_in, d_probs,
                       M, d_feat, ws);
  else
    hipLaunchKernelGGL((dynconv_bwd_kernel<2>), dim3(nb), dim3(256), 0, st, feat, kernels, probs, d_logits_in, d_probs,
                       M, d_feat, ws);
  SCAN_LAUNCH_CHECK("dynconv_bwd");
  hipLaunchKernelGGL(dynconv_reduce_kernel, dim3((K * C + 255) / 256), dim3(256), 0, st, ws, nb, K * C, d_kernels);
  SCAN_LAUNCH_CHECK("dynconv_reduce");
  return 0;
}
