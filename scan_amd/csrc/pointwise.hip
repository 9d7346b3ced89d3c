// HBM-bound pointwise / reduction kernels of the SCAN hot path for gfx950:
// SigmoidFocalLoss, IoU loss, (weighted) BCE-with-logits, GRL scale, softmax focal
// loss, fused SGD.  64-lane wavefront reductions, float4 streaming where the layout
// allows, grid capped at 2048 blocks with grid-stride loops.
#include <float.h>

#include "common.h"
#include <algorithm>

// ------------------------------------------------------------------ sigmoid focal loss
// follows the reference's stable CUDA formula, csrc/cuda/SigmoidFocalLoss_cuda.cu:20-101
// Shared pieces: p = sigmoid(x) and expf(x - 2x[x>=0]) = expf(-|x|) are evaluated once; gamma == 2 (the shipped
// MODEL.FCOS.LOSS_GAMMA) squares instead of calling powf; the positive-class term is only evaluated by lanes that
// need it (a wave with no positive label skips it entirely).
template <bool G2>
__device__ __forceinline__ float focal_pow(float b, float gamma) {
  return G2 ? b * b : powf(b, gamma);
}
// One v_exp_f32, one v_rcp_f32 and one v_log_f32 per element, shared by both branches (round 4; the kernels were bound
// by vector-ALU issue, not by HBM: profiles/r03_pointwise_counters.txt -- two expf, a full-precision division and one or
// two logf per element through the math library):
//     e = exp(-|x|) in (0, 1]        r = 1 / (1 + e)        L = log(1 + e)
//     sigmoid(x) = x >= 0 ? r : e r         1 - sigmoid(x) = x >= 0 ? e r : r
//     log sigmoid(x) = min(x, 0) - L        the reference's second term -x [x >= 0] - log(1 + exp(x - 2 x [x >= 0])) = -max(x, 0) - L
// v_exp_f32 / v_log_f32 are good to ~1 ulp of their result and v_rcp_f32 to 1 ulp; L switches to its series below 2^-7
// where log(1 + e) would lose the low bits of e.  Against the C oracle (the reference's CUDA formula in libm floats) the
// element losses agree to rtol 1e-5 / atol 1e-7 as before (tests/test_gpu_kernels.py::test_sigmoid_focal_*).
struct SigParts {
  float p, q, L;  // sigmoid(x), 1 - sigmoid(x), log(1 + exp(-|x|))
};
__device__ __forceinline__ float fast_log1p_unit(float e) {  // log(1 + e), 0 <= e <= 1
  return e < 0.0078125f ? e * (1.f - e * (0.5f - e * 0.33333334f)) : __logf(1.f + e);
}
__device__ __forceinline__ SigParts sig_parts(float x) {
  const float e = __expf(-fabsf(x));
  const float r = __builtin_amdgcn_rcpf(1.f + e);
  const float er = e * r;
  SigParts s;
  s.p = x >= 0.f ? r : er;
  s.q = x >= 0.f ? er : r;
  s.L = fast_log1p_unit(e);
  return s;
}
template <bool G2>
__device__ __forceinline__ float focal_fwd_elem(float x, int t, int d, float gamma, float alpha) {
  if (t < 0) return 0.f;
  const SigParts s = sig_parts(x);
  if (t == d + 1) {
    const float logp = fmaxf(fminf(x, 0.f) - s.L, -87.33654f);  // log(max(p, FLT_MIN))
    return -focal_pow<G2>(s.q, gamma) * logp * alpha;
  }
  return focal_pow<G2>(s.p, gamma) * (fmaxf(x, 0.f) + s.L) * (1.f - alpha);
}
template <bool G2>
__device__ __forceinline__ float focal_bwd_elem(float x, int t, int d, float gamma, float alpha) {
  if (t < 0) return 0.f;
  const SigParts s = sig_parts(x);
  if (t == d + 1) {
    const float logp = fmaxf(fminf(x, 0.f) - s.L, -87.33654f);
    return -focal_pow<G2>(s.q, gamma) * (s.q - s.p * gamma * logp) * alpha;
  }
  return -focal_pow<G2>(s.p, gamma) * ((-fmaxf(x, 0.f) - s.L) * s.q * gamma - s.p) * (1.f - alpha);
}

// row (label index) and class of the 4 consecutive elements starting at i0 = 4q.  CT = compile-time class count of the
// two shipped configurations (8: Cityscapes, one label per two float4; 1: Sim10k / KITTI, one label per element),
// 0 = any C (one integer division per float4 instead of the 64-bit division per element this kernel used to pay).
template <int CT>
__device__ __forceinline__ void focal_rowcol(int64_t q, int C, const int* __restrict__ targets, int64_t total,
                                             int (&t)[4], int (&d)[4]) {
  if (CT == 8) {
    const int tt = targets[q >> 1];
    const int d0 = (int)(q & 1) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[e] = tt;
      d[e] = d0 + e;
    }
  } else if (CT == 1) {
    const int4 tt = reinterpret_cast<const int4*>(targets)[q];
    t[0] = tt.x; t[1] = tt.y; t[2] = tt.z; t[3] = tt.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = 0;
  } else {
    const int64_t i0 = q << 2;
    int64_t n = i0 / C;
    int dd = (int)(i0 - n * C);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[e] = (i0 + e < total) ? targets[n] : -1;
      d[e] = dd;
      if (++dd == C) {
        dd = 0;
        ++n;
      }
    }
  }
}

template <int CT, bool G2>
__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ logits,
                                                        const int* __restrict__ targets, int64_t total, int C,
                                                        float gamma, float alpha, float* __restrict__ losses,
                                                        float* __restrict__ loss_sum) {
  __shared__ float red[4];
  float acc = 0.f;
  const int64_t n4 = (total + 3) >> 2;
  const bool vec = (total & 3) == 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q << 2;
    float x[4], l[4];
    if (vec) {
      const float4 v = reinterpret_cast<const float4*>(logits)[q];
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (i0 + e < total) ? logits[i0 + e] : 0.f;
    }
    int t[4], d[4];
    focal_rowcol<CT>(q, C, targets, total, t, d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      l[e] = (i0 + e < total) ? focal_fwd_elem<G2>(x[e], t[e], d[e], gamma, alpha) : 0.f;
      acc += l[e];
    }
    if (losses != nullptr) {
      if (vec) {
        reinterpret_cast<float4*>(losses)[q] = make_float4(l[0], l[1], l[2], l[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i0 + e < total) losses[i0 + e] = l[e];
      }
    }
  }
  if (loss_sum != nullptr) {
    const float s = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss_sum, s);
  }
}

template <int CT, bool G2>
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits,
                                                        const int* __restrict__ targets,
                                                        const float* __restrict__ d_losses, float d_scale,
                                                        int64_t total, int C, float gamma, float alpha,
                                                        float* __restrict__ d_logits) {
  const int64_t n4 = (total + 3) >> 2;
  const bool vec = (total & 3) == 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = q << 2;
    float x[4], up[4], g[4];
    if (vec) {
      const float4 v = reinterpret_cast<const float4*>(logits)[q];
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
      if (d_losses != nullptr) {
        const float4 u = reinterpret_cast<const float4*>(d_losses)[q];
        up[0] = u.x; up[1] = u.y; up[2] = u.z; up[3] = u.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = (i0 + e < total) ? logits[i0 + e] : 0.f;
        if (d_losses != nullptr) up[e] = (i0 + e < total) ? d_losses[i0 + e] : 0.f;
      }
    }
    int t[4], d[4];
    focal_rowcol<CT>(q, C, targets, total, t, d);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      g[e] = ((i0 + e < total) ? focal_bwd_elem<G2>(x[e], t[e], d[e], gamma, alpha) : 0.f) *
             (d_losses != nullptr ? up[e] : d_scale);
    if (vec) {
      reinterpret_cast<float4*>(d_logits)[q] = make_float4(g[0], g[1], g[2], g[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i0 + e < total) d_logits[i0 + e] = g[e];
    }
  }
}

// loss reductions end in one float atomic per block on ONE address: keep the block count low (same-address atomics
// serialise at ~13 ns each: 2048 blocks cost ~55 us whatever the tensor size) and give each thread more elements
// scan_tune "reduce_blocks": the cap for the kernels bound by vector-ALU issue (focal losses: they need the occupancy of 8 blocks
// per CU); `light` = kernels with little arithmetic per byte, whose run time the end-of-block atomics are a visible part of:
// half the cap.  Round 5 sweep at M = 2^24 (tools/pointwise_roofline.py, same process): IoU forward 132 -> 113 us (0.57 -> 0.67 of
// the HBM roof), CKA BCE forward 220 -> 206 us (0.65 -> 0.69) at 1,024; sigmoid focal fused-sum forward 161 -> 193 us (worse).
int g_scan_reduce_blocks = 2048;
static inline int grid_reduce(int64_t work_items, int block, bool light = false) {
  int64_t g = (work_items + (int64_t)block * 8 - 1) / ((int64_t)block * 8);
  int cap = light ? g_scan_reduce_blocks / 2 : g_scan_reduce_blocks;
  if (cap < 1) cap = 1;  // scan_tune accepts any int: every setting must still give a launchable grid
  if (g < 1) g = 1;
  if (g > cap) g = cap;  // large tensors: the ~27 us of serialised atomics hide behind >= 100 us of streaming
  return (int)g;
}

#define FOCAL_DISPATCH(KERNEL, GRID, ...)                                                                           \
  do {                                                                                                              \
    const bool g2 = gamma == 2.0f;                                                                                  \
    const bool al = (total & 3) == 0;                                                                               \
    if (C == 8 && al) {                                                                                             \
      if (g2) hipLaunchKernelGGL((KERNEL<8, true>), dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__);      \
      else hipLaunchKernelGGL((KERNEL<8, false>), dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__);        \
    } else if (C == 1 && al && (reinterpret_cast<uintptr_t>(targets) & 15) == 0) {                                  \
      if (g2) hipLaunchKernelGGL((KERNEL<1, true>), dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__);      \
      else hipLaunchKernelGGL((KERNEL<1, false>), dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__);        \
    } else {                                                                                                        \
      if (g2) hipLaunchKernelGGL((KERNEL<0, true>), dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__);      \
      else hipLaunchKernelGGL((KERNEL<0, false>), dim3(GRID), dim3(256), 0, as_stream(stream), __VA_ARGS__);        \
    }                                                                                                               \
  } while (0)

extern "C" int scan_sigmoid_focal_loss_forward(const float* logits, const int32_t* targets, int64_t M, int32_t C,
                                               float gamma, float alpha, float* losses, float* loss_sum,
                                               void* stream) {
  SCAN_CHECK_ARG(M >= 0 && C > 0, "sigmoid_focal_loss_forward: bad shape M=%lld C=%d", (long long)M, C);
  SCAN_CHECK_ARG(losses || loss_sum, "sigmoid_focal_loss_forward: no output requested");
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && targets, "sigmoid_focal_loss_forward: null input");
  const int64_t total = M * C;
  const int grid = loss_sum ? grid_reduce((total + 3) / 4, 256) : grid_for((total + 3) / 4, 256);
  FOCAL_DISPATCH(focal_fwd_kernel, grid, logits, targets, total, C, gamma, alpha, losses, loss_sum);
  SCAN_LAUNCH_CHECK("focal_fwd");
  return 0;
}

extern "C" int scan_sigmoid_focal_loss_backward(const float* logits, const int32_t* targets, const float* d_losses,
                                                float d_scale, int64_t M, int32_t C, float gamma, float alpha,
                                                float* d_logits, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && C > 0, "sigmoid_focal_loss_backward: bad shape M=%lld C=%d", (long long)M, C);
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && targets && d_logits, "sigmoid_focal_loss_backward: null pointer");
  const int64_t total = M * C;
  const int grid = grid_for((total + 3) / 4, 256);
  FOCAL_DISPATCH(focal_bwd_kernel, grid, logits, targets, d_losses, d_scale, total, C, gamma, alpha, d_logits);
  SCAN_LAUNCH_CHECK("focal_bwd");
  return 0;
}

// ------------------------------------------------------------------ IoU loss (layers/iou_loss.py:5-36)
struct IouTerms {
  float loss, hI, wI, U, I, pw, ph;
};
__device__ __forceinline__ IouTerms iou_terms(const float4 p, const float4 t) {
  IouTerms r;
  const float ta = (t.x + t.z) * (t.y + t.w);
  r.pw = p.x + p.z;
  r.ph = p.y + p.w;
  const float pa = r.pw * r.ph;
  r.wI = fminf(p.x, t.x) + fminf(p.z, t.z);
  r.hI = fminf(p.w, t.w) + fminf(p.y, t.y);
  r.I = r.wI * r.hI;
  r.U = ta + pa - r.I;
  r.loss = -logf((r.I + 1.0f) / (r.U + 1.0f));
  return r;
}

__global__ __launch_bounds__(256) void iou_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ weight, int64_t P,
                                                      float* __restrict__ out2) {
  __shared__ float red[4];
  float num = 0.f, den = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = reinterpret_cast<const float4*>(pred)[i];
    const float4 t = reinterpret_cast<const float4*>(target)[i];
    const float w = weight ? weight[i] : 1.f;
    num += iou_terms(p, t).loss * w;
    den += w;
  }
  const float sn = block_sum_256(num, red);
  const float sd = block_sum_256(den, red);
  if (threadIdx.x == 0) {
    atomicAdd(out2, sn);
    atomicAdd(out2 + 1, sd);
  }
}

// torch.min(a, b) backward: the gradient goes to the smaller input, split evenly on exact ties
__device__ __forceinline__ float min_grad(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

__global__ __launch_bounds__(256) void iou_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                      const float* __restrict__ weight, int64_t P,
                                                      const float* __restrict__ g_num, float* __restrict__ d_pred) {
  const float g = g_num[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = reinterpret_cast<const float4*>(pred)[i];
    const float4 t = reinterpret_cast<const float4*>(target)[i];
    const IouTerms r = iou_terms(p, t);
    const float w = (weight ? weight[i] : 1.f) * g;
    // loss = log(U+1) - log(I+1);  U = ta + pa - I
    const float dU = 1.f / (r.U + 1.f);
    const float dI = -dU - 1.f / (r.I + 1.f);  // dU/dI = -1
    float4 o;
    o.x = w * (dU * r.ph + dI * r.hI * min_grad(p.x, t.x));  // left
    o.z = w * (dU * r.ph + dI * r.hI * min_grad(p.z, t.z));  // right
    o.y = w * (dU * r.pw + dI * r.wI * min_grad(p.y, t.y));  // top
    o.w = w * (dU * r.pw + dI * r.wI * min_grad(p.w, t.w));  // bottom
    reinterpret_cast<float4*>(d_pred)[i] = o;
  }
}

extern "C" int scan_iou_loss_forward(const float* pred, const float* target, const float* weight, int64_t P,
                                     float* out2, void* stream) {
  SCAN_CHECK_ARG(P >= 0 && out2, "iou_loss_forward: bad arguments");
  if (P == 0) return 0;
  SCAN_CHECK_ARG(pred && target, "iou_loss_forward: null input");
  hipLaunchKernelGGL(iou_fwd_kernel, dim3(grid_reduce(P, 256, true)), dim3(256), 0, as_stream(stream), pred, target, weight, P,
                     out2);
  SCAN_LAUNCH_CHECK("iou_fwd");
  return 0;
}

extern "C" int scan_iou_loss_backward(const float* pred, const float* target, const float* weight, int64_t P,
                                      const float* g_num_dev, float* d_pred, void* stream) {
  SCAN_CHECK_ARG(P >= 0, "iou_loss_backward: bad arguments");
  if (P == 0) return 0;
  SCAN_CHECK_ARG(pred && target && g_num_dev && d_pred, "iou_loss_backward: null pointer");
  hipLaunchKernelGGL(iou_bwd_kernel, dim3(grid_for(P, 256)), dim3(256), 0, as_stream(stream), pred, target, weight, P,
                     g_num_dev, d_pred);
  SCAN_LAUNCH_CHECK("iou_bwd");
  return 0;
}

// ------------------------------------------------------------------ BCE with logits (optionally weighted)
// VEC: logits / targets / weight (unit stride) read as float4 -- four times fewer load instructions for the same bytes (the
// scalar form ran at 0.25 of the HBM roof at M = 2^24, bound by load issue: profiles/r04_pointwise_roofline.json)
template <bool VEC>
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ logits,
                                                      const float* __restrict__ targets, float const_target,
                                                      const float* __restrict__ weight, int64_t w_stride, int64_t M,
                                                      float* __restrict__ out2) {
  __shared__ float red[4];
  float num = 0.f, den = 0.f;
  auto elem = [&](float x, float t, float w) {
    // max(x,0) - x*t + log(1 + exp(-|x|))
    const float l = fmaxf(x, 0.f) - x * t + fast_log1p_unit(__expf(-fabsf(x)));
    num += l * w;
    den += w;
  };
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {
    const int64_t m4 = M >> 2;
#pragma unroll 4
    for (int64_t q = tid; q < m4; q += nthr) {
      const float4 x = reinterpret_cast<const float4*>(logits)[q];
      const float4 t = targets ? reinterpret_cast<const float4*>(targets)[q]
                               : make_float4(const_target, const_target, const_target, const_target);
      const float4 w = weight ? reinterpret_cast<const float4*>(weight)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
      elem(x.x, t.x, w.x);
      elem(x.y, t.y, w.y);
      elem(x.z, t.z, w.z);
      elem(x.w, t.w, w.w);
    }
    for (int64_t i = (m4 << 2) + tid; i < M; i += nthr) elem(logits[i], targets ? targets[i] : const_target, weight ? weight[i] : 1.f);
  } else {
    for (int64_t i = tid; i < M; i += nthr)
      elem(logits[i], targets ? targets[i] : const_target, weight ? weight[i * w_stride] : 1.f);
  }
  const float sn = block_sum_256(num, red);
  const float sd = block_sum_256(den, red);
  if (threadIdx.x == 0) {
    atomicAdd(out2, sn);
    atomicAdd(out2 + 1, sd);
  }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ logits,
                                                      const float* __restrict__ targets, float const_target,
                                                      const float* __restrict__ weight, int64_t w_stride, int64_t M,
                                                      const float* __restrict__ g_dev, float* __restrict__ d_logits) {
  const float g = g_dev[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = logits[i];
    const float t = targets ? targets[i] : const_target;
    const float w = weight ? weight[i * w_stride] : 1.f;
    const float s = 1.f / (1.f + expf(-x));
    d_logits[i] = g * w * (s - t);
  }
}

extern "C" int scan_bce_logits_forward(const float* logits, const float* targets, float const_target,
                                       const float* weight, int64_t w_stride, int64_t M, float* out2, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && out2, "bce_logits_forward: bad arguments");
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits, "bce_logits_forward: null input");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (M >= 4096 && al16(logits) && (!targets || al16(targets)) && (!weight || (w_stride == 1 && al16(weight))))
    // the kernel ends in two float atomics per block on one cache line (~13 ns each, serialised): at 8 bytes per element the
    // 2,048 blocks of grid_reduce cost more in atomics (53 us) than the stream itself (27 us at M = 2^24) -- 512 blocks here
    hipLaunchKernelGGL(bce_fwd_kernel<true>, dim3(std::min(512, grid_reduce((M + 3) / 4, 256))), dim3(256), 0, as_stream(stream),
                       logits, targets, const_target, weight, w_stride, M, out2);
  else
    hipLaunchKernelGGL(bce_fwd_kernel<false>, dim3(grid_reduce(M, 256)), dim3(256), 0, as_stream(stream), logits, targets,
                       const_target, weight, w_stride, M, out2);
  SCAN_LAUNCH_CHECK("bce_fwd");
  return 0;
}

extern "C" int scan_bce_logits_backward(const float* logits, const float* targets, float const_target,
                                        const float* weight, int64_t w_stride, int64_t M, const float* g_dev,
                                        float* d_logits, void* stream) {
  SCAN_CHECK_ARG(M >= 0, "bce_logits_backward: bad arguments");
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && g_dev && d_logits, "bce_logits_backward: null pointer");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(M, 256)), dim3(256), 0, as_stream(stream), logits, targets,
                     const_target, weight, w_stride, M, g_dev, d_logits);
  SCAN_LAUNCH_CHECK("bce_bwd");
  return 0;
}

// ------------------------------------------------------------------ CKA class-conditional BCE
#define CKA_MAXC 16
// fin != nullptr: the block that finishes last (ticket fin[0], zeroed by the caller) turns the completed sums into the
// loss  fin[1] = sum_c (num_c / den_c) / Cf  -- the reference's per-class weighted means averaged over the classes
// (discriminator/fcos_head_discriminator_con.py:119-121) -- so the layer needs no select / div / sum / div kernels of
// its own behind this launch.  The sums were added with float atomics (executed at the memory side): they are read
// back the same way.
__device__ __forceinline__ void cka_finish(float* __restrict__ out, int Cf, float* __restrict__ fin) {
  __shared__ unsigned last_flag;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(reinterpret_cast<unsigned*>(fin), 1u);
    last_flag = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (last_flag && threadIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < Cf; ++c) s += atomicAdd(out + 2 * c, 0.f) / atomicAdd(out + 2 * c + 1, 0.f);
    fin[1] = s / (float)Cf;
  }
}
__global__ __launch_bounds__(256) void cka_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ act,
                                                      int64_t M, int Cf, float t, float* __restrict__ out,
                                                      float* __restrict__ fin) {
  __shared__ float red[4];
  float num[CKA_MAXC], den[CKA_MAXC];
#pragma unroll
  for (int c = 0; c < CKA_MAXC; ++c) num[c] = den[c] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < CKA_MAXC; ++c) {
      if (c < Cf) {
        const float x = logits[i * Cf + c];
        const float w = act[i * (Cf + 1) + c + 1];
        const float l = fmaxf(x, 0.f) - x * t + fast_log1p_unit(__expf(-fabsf(x)));
        num[c] += l * w;
        den[c] += w;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CKA_MAXC; ++c) {
    if (c < Cf) {
      const float sn = block_sum_256(num[c], red);
      const float sd = block_sum_256(den[c], red);
      if (threadIdx.x == 0) {
        atomicAdd(out + 2 * c, sn);
        atomicAdd(out + 2 * c + 1, sd);
      }
    }
  }
  if (fin != nullptr) cka_finish(out, Cf, fin);
}

// Cf == 8 (Cityscapes): a thread owns one float4 of logits = half a row, i.e. ALWAYS the same four classes (the grid
// stride is even), so its eight partial sums stay in registers; the act-map row (9 floats, unaligned) is read by the two
// lanes of a row as 4 + 4 scalars of one contiguous wave-wide segment.  Even / odd lanes are reduced separately.
__global__ __launch_bounds__(256) void cka_fwd8_kernel(const float* __restrict__ logits, const float* __restrict__ act,
                                                       int64_t M, float t, float* __restrict__ out,
                                                       float* __restrict__ fin) {
  __shared__ float red[4][16];
  float num[4] = {0.f, 0.f, 0.f, 0.f}, den[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t n4 = M * 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(logits)[q];
    const float* a = act + (q >> 1) * 9 + 1 + (q & 1) * 4;
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float w = a[e];
      const float l = fmaxf(x[e], 0.f) - x[e] * t + fast_log1p_unit(__expf(-fabsf(x[e])));
      num[e] += l * w;
      den[e] += w;
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int off = 32; off >= 2; off >>= 1) {  // keeps lane parity: lane 0 <- even lanes, lane 1 <- odd lanes
      num[e] += __shfl_down(num[e], off, 64);
      den[e] += __shfl_down(den[e], off, 64);
    }
    if (lane < 2) {
      red[wid][2 * (lane * 4 + e)] = num[e];       // class c = (lane & 1) * 4 + e
      red[wid][2 * (lane * 4 + e) + 1] = den[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) atomicAdd(out + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (fin != nullptr) cka_finish(out, 8, fin);
}

// sums == nullptr: g [Cf] are the per-class coefficients; else g [1] is the gradient of the loss scalar and the
// coefficient of class c is g / (Cf * den_c), den_c = sums[2 c + 1] (what the layer's backward used to compute with
// three small kernels)
__global__ __launch_bounds__(256) void cka_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ act,
                                                      int64_t M, int Cf, float t, const float* __restrict__ g,
                                                      const float* __restrict__ sums, float* __restrict__ d_logits) {
  __shared__ float coef[CKA_MAXC];
  if (threadIdx.x < Cf) coef[threadIdx.x] = sums ? g[0] / ((float)Cf * sums[2 * threadIdx.x + 1]) : g[threadIdx.x];
  __syncthreads();
  const int64_t total = M * Cf;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / Cf;
    const int c = (int)(i - m * Cf);
    const float x = logits[i];
    const float s = 1.f / (1.f + expf(-x));
    d_logits[i] = coef[c] * act[m * (Cf + 1) + c + 1] * (s - t);
  }
}

static int cka_forward_launch(const float* logits, const float* act, int64_t M, int32_t Cf, float target, float* out,
                              float* fin, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && Cf > 0 && Cf <= CKA_MAXC && out, "cka_bce_forward: bad arguments (Cf=%d)", Cf);
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && act, "cka_bce_forward: null input");
  if (Cf == 8 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
    int g = grid_reduce(M * 2, 256, true);
    hipLaunchKernelGGL(cka_fwd8_kernel, dim3(g), dim3(256), 0, as_stream(stream), logits, act, M, target, out, fin);
  } else {
    hipLaunchKernelGGL(cka_fwd_kernel, dim3(grid_reduce(M, 256)), dim3(256), 0, as_stream(stream), logits, act, M, Cf,
                       target, out, fin);
  }
  SCAN_LAUNCH_CHECK("cka_fwd");
  return 0;
}

extern "C" int scan_cka_bce_forward(const float* logits, const float* act, int64_t M, int32_t Cf, float target,
                                    float* out, void* stream) {
  return cka_forward_launch(logits, act, M, Cf, target, out, nullptr, stream);
}

// out: 2 * Cf + 2 floats, zeroed by the caller: the sums, a ticket word, the loss (written by the last block)
extern "C" int scan_cka_bce_forward_loss(const float* logits, const float* act, int64_t M, int32_t Cf, float target,
                                         float* out, void* stream) {
  SCAN_CHECK_ARG(M > 0, "cka_bce_forward_loss: needs at least one row (the loss is a ratio of sums)");
  return cka_forward_launch(logits, act, M, Cf, target, out, out ? out + 2 * Cf : nullptr, stream);
}

extern "C" int scan_cka_bce_backward(const float* logits, const float* act, int64_t M, int32_t Cf, float target,
                                     const float* g_dev, float* d_logits, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && Cf > 0 && Cf <= CKA_MAXC, "cka_bce_backward: bad arguments (Cf=%d)", Cf);
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && act && g_dev && d_logits, "cka_bce_backward: null pointer");
  hipLaunchKernelGGL(cka_bwd_kernel, dim3(grid_for(M * Cf, 256)), dim3(256), 0, as_stream(stream), logits, act, M, Cf,
                     target, g_dev, (const float*)nullptr, d_logits);
  SCAN_LAUNCH_CHECK("cka_bwd");
  return 0;
}

// g_loss [1]: gradient of the loss scalar of scan_cka_bce_forward_loss; sums: that call's out
extern "C" int scan_cka_bce_backward_loss(const float* logits, const float* act, int64_t M, int32_t Cf, float target,
                                          const float* g_loss, const float* sums, float* d_logits, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && Cf > 0 && Cf <= CKA_MAXC, "cka_bce_backward_loss: bad arguments (Cf=%d)", Cf);
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && act && g_loss && sums && d_logits, "cka_bce_backward_loss: null pointer");
  hipLaunchKernelGGL(cka_bwd_kernel, dim3(grid_for(M * Cf, 256)), dim3(256), 0, as_stream(stream), logits, act, M, Cf,
                     target, g_loss, sums, d_logits);
  SCAN_LAUNCH_CHECK("cka_bwd");
  return 0;
}

// ------------------------------------------------------------------ y = alpha * x (GRL)
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, float alpha, float* __restrict__ y,
                                                    int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    reinterpret_cast<float4*>(y)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[(n4 << 2) + threadIdx.x] = alpha * x[(n4 << 2) + threadIdx.x];
}

extern "C" int scan_scale(const float* x, float alpha, float* y, int64_t n, void* stream) {
  SCAN_CHECK_ARG(n >= 0, "scale: bad n");
  if (n == 0) return 0;
  SCAN_CHECK_ARG(x && y, "scale: null pointer");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, as_stream(stream), x, alpha, y, n);
  SCAN_LAUNCH_CHECK("scale");
  return 0;
}

// ------------------------------------------------------------------ softmax focal loss (alpha = 1)
// layers/sigmoid_focal_loss_wbg.py:38-64:  p = softmax(z)[label] clamped at 1e-15; -(1-p)^g log p
#define SFL_MAXK 16
// Rows are K floats (36 B at K = 9): a lane reading its own row touches a new cache line every 3-4 lanes.  A block
// therefore moves its 256 rows as ONE contiguous 256*K-float segment with coalesced dword accesses through LDS and each
// lane then walks its row in LDS (stride K floats: conflict-free for odd K).
// Round 4: no workgroup barriers.  A WAVE moves its own 64 rows (64 * K floats, contiguous) through a private LDS region:
// K coalesced dword loads per lane (lane + 64 j), written to LDS, then every lane walks its row there (stride K floats:
// conflict-free for odd K); LDS operations of one wave execute in order, so a wave-level fence is all that is needed.
// Two 64-row batches are in flight per wave and iteration (the first cut -- 256 rows per workgroup between two
// __syncthreads -- spent 72 % of its wave cycles parked: profiles/r03_pointwise_counters.txt).
template <bool BWD>
__global__ __launch_bounds__(256) void sfl_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                  int64_t M, int K, float gamma, float d_scale,
                                                  float* __restrict__ loss_sum, float* __restrict__ d_logits) {
  constexpr int NB = 2;  // batches of 64 rows per wave and iteration
  __shared__ float rows_all[4 * NB * 64 * SFL_MAXK];
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* wrows = rows_all + wid * (NB * 64 * SFL_MAXK);
  const bool g2 = gamma == 2.0f;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(logits) & 15) == 0;
  float acc = 0.f;
  const int64_t nbatch = (M + 63) / 64;
  const int64_t stride = (int64_t)gridDim.x * 4 * NB;
  for (int64_t b0 = ((int64_t)blockIdx.x * 4 + wid) * NB; b0 < nbatch; b0 += stride) {
    constexpr int NV = SFL_MAXK / 4;  // float4 per lane and batch: 64 rows x K floats = 16 K float4, lane + 64 j
    float4 v[NB][NV];
    int nrow[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int64_t r0 = (b0 + u) * 64;
      nrow[u] = r0 < M ? (int)((M - r0 < 64) ? (M - r0) : 64) : 0;
      const float* src = logits + r0 * K;  // 64 * K * 4 bytes per batch: 16-byte aligned whenever logits is
      const int nf = nrow[u] * K;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int i4 = 4 * (lane + 64 * j);
        v[u][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i4 + 3 < nf && vec_ok) {
          v[u][j] = *reinterpret_cast<const float4*>(src + i4);
        } else if (i4 < nf) {
          v[u][j].x = src[i4];
          if (i4 + 1 < nf) v[u][j].y = src[i4 + 1];
          if (i4 + 2 < nf) v[u][j].z = src[i4 + 2];
          if (i4 + 3 < nf) v[u][j].w = src[i4 + 3];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NB; ++u)
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (4 * 64 * j < 64 * K) *reinterpret_cast<float4*>(wrows + u * 64 * SFL_MAXK + 4 * (lane + 64 * j)) = v[u][j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int64_t r0 = (b0 + u) * 64;
      float* z = wrows + u * 64 * SFL_MAXK + lane * K;
      if (lane < nrow[u]) {
        float e[SFL_MAXK];
        float mx = z[0];
#pragma unroll
        for (int k = 1; k < SFL_MAXK; ++k)
          if (k < K) mx = fmaxf(mx, z[k]);
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < SFL_MAXK; ++k) {
          e[k] = (k < K) ? __expf(z[k] - mx) : 0.f;
          den += e[k];
        }
        const int lab = (int)labels[r0 + lane];
        float el = 0.f;
#pragma unroll
        for (int k = 0; k < SFL_MAXK; ++k) el = (k == lab) ? e[k] : el;
        const float inv = __builtin_amdgcn_rcpf(den);
        const float pr = el * inv;
        if (!BWD) {
          const float p = fmaxf(pr, 1e-15f);
          const float om = 1.f - p;
          // log p through its series near p = 1: log(p) there loses the low bits of 1 - p
          const float lp = om < 0.0078125f ? -om * (1.f + om * (0.5f + om * 0.33333334f)) : __logf(p);
          acc += -(g2 ? om * om : powf(om, gamma)) * lp;
        } else {
          float dLdp = 0.f;  // clamp(min=1e-15) has zero gradient below the clamp
          if (pr >= 1e-15f) {
            const float om = 1.f - pr;
            const float lp = om < 0.0078125f ? -om * (1.f + om * (0.5f + om * 0.33333334f)) : __logf(pr);
            const float ip = __builtin_amdgcn_rcpf(pr);
            dLdp = g2 ? (2.f * om * lp - om * om * ip) : (gamma * powf(om, gamma - 1.f) * lp - powf(om, gamma) * ip);
          }
          const float c = dLdp * pr * d_scale;
#pragma unroll
          for (int k = 0; k < SFL_MAXK; ++k)
            if (k < K) z[k] = c * ((k == lab ? 1.f : 0.f) - e[k] * inv);
        }
      }
    }
    if (BWD) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int64_t r0 = (b0 + u) * 64;
        float* dst = d_logits + r0 * K;
        const int nf = nrow[u] * K;
        const bool st_ok = (reinterpret_cast<uintptr_t>(d_logits) & 15) == 0;
#pragma unroll
        for (int j = 0; j < SFL_MAXK / 4; ++j) {
          const int i4 = 4 * (lane + 64 * j);
          const float4 o = *reinterpret_cast<const float4*>(wrows + u * 64 * SFL_MAXK + i4);
          if (i4 + 3 < nf && st_ok) {
            *reinterpret_cast<float4*>(dst + i4) = o;
          } else if (i4 < nf) {
            dst[i4] = o.x;
            if (i4 + 1 < nf) dst[i4 + 1] = o.y;
            if (i4 + 2 < nf) dst[i4 + 2] = o.z;
            if (i4 + 3 < nf) dst[i4 + 3] = o.w;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the region is rewritten by the next iteration's batches
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (!BWD) {
    const float sum = block_sum_256(acc, red);
    if (threadIdx.x == 0) atomicAdd(loss_sum, sum);
  }
}

extern "C" int scan_softmax_focal_forward(const float* logits, const int64_t* labels, int64_t M, int32_t K,
                                          float gamma, float* loss_sum, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && K > 0 && K <= SFL_MAXK && loss_sum, "softmax_focal_forward: bad arguments (K=%d)", K);
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && labels, "softmax_focal_forward: null input");
  int64_t g = (M + 511) / 512;  // a workgroup iteration covers 4 waves x 2 batches x 64 rows
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(sfl_kernel<false>, dim3((int)g), dim3(256), 0, as_stream(stream), logits, labels, M, K, gamma, 0.f,
                     loss_sum, (float*)nullptr);
  SCAN_LAUNCH_CHECK("sfl_fwd");
  return 0;
}

extern "C" int scan_softmax_focal_backward(const float* logits, const int64_t* labels, int64_t M, int32_t K,
                                           float gamma, float d_scale, float* d_logits, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && K > 0 && K <= SFL_MAXK, "softmax_focal_backward: bad arguments (K=%d)", K);
  if (M == 0) return 0;
  SCAN_CHECK_ARG(logits && labels && d_logits, "softmax_focal_backward: null pointer");
  int64_t g = (M + 511) / 512;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(sfl_kernel<true>, dim3((int)g), dim3(256), 0, as_stream(stream), logits, labels, M, K, gamma,
                     d_scale, (float*)nullptr, d_logits);
  SCAN_LAUNCH_CHECK("sfl_bwd");
  return 0;
}

// ------------------------------------------------------------------ fused SGD + momentum + weight decay
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, int64_t n, float lr, float wd,
                                                  float momentum, int first) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float w = p[i];
    const float gg = g[i] + wd * w;
    const float b = first ? gg : momentum * buf[i] + gg;
    buf[i] = b;
    p[i] = w - lr * b;
  }
}

extern "C" int scan_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float wd, float momentum,
                                 int32_t first_step, void* stream) {
  SCAN_CHECK_ARG(n >= 0, "sgd_momentum: bad n");
  if (n == 0) return 0;
  SCAN_CHECK_ARG(p && g && buf, "sgd_momentum: null pointer");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), p, g, buf, n, lr, wd,
                     momentum, first_step);
  SCAN_LAUNCH_CHECK("sgd");
  return 0;
}

// ------------------------------------------------------------------ 2x2 / stride-2 max pooling on NHWC rows
// (replaces nn.MaxPool2d(2, 2) of the VGG body, reference backbone/mmdetection/vgg.py:33).  HBM-bound:
// forward reads 4 and writes 1 float4 per lane; backward re-derives the argmax (first maximum in window
// order, like F.max_pool2d) from x and y instead of storing indices.
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                           float* __restrict__ y) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const float4* b = reinterpret_cast<const float4*>(x + (((int64_t)n * H + 2 * yo) * W + 2 * xo) * C) + c4;
    const float4 a0 = b[0], a1 = b[C4], a2 = b[(int64_t)W * C4], a3 = b[(int64_t)W * C4 + C4];
    float4 m;
    m.x = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
    m.y = fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y));
    m.z = fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z));
    m.w = fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w));
    reinterpret_cast<float4*>(y)[i] = m;
  }
}

__device__ __forceinline__ void pool_bwd1(float a0, float a1, float a2, float a3, float m, float g, float& d0,
                                          float& d1, float& d2, float& d3) {
  d0 = d1 = d2 = d3 = 0.f;
  if (a0 == m) d0 = g;
  else if (a1 == m) d1 = g;
  else if (a2 == m) d2 = g;
  else d3 = g;
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, int N, int H, int W, int C,
                                                           float* __restrict__ dx, int relu_mask) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int64_t off = ((((int64_t)n * H + 2 * yo) * W + 2 * xo) * C) / 4 + c4;
    const float4* b = reinterpret_cast<const float4*>(x) + off;
    float4* o = reinterpret_cast<float4*>(dx) + off;
    const int64_t s1 = C4, s2 = (int64_t)W * C4, s3 = s2 + C4;
    const float4 a0 = b[0], a1 = b[s1], a2 = b[s2], a3 = b[s3];
    const float4 m = reinterpret_cast<const float4*>(y)[i];
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    if (relu_mask) {  // x is a ReLU output whose own backward was deferred to its consumers: dx *= (x > 0)
      g.x = m.x > 0.f ? g.x : 0.f;
      g.y = m.y > 0.f ? g.y : 0.f;
      g.z = m.z > 0.f ? g.z : 0.f;
      g.w = m.w > 0.f ? g.w : 0.f;
    }
    float4 d0, d1, d2, d3;
    pool_bwd1(a0.x, a1.x, a2.x, a3.x, m.x, g.x, d0.x, d1.x, d2.x, d3.x);
    pool_bwd1(a0.y, a1.y, a2.y, a3.y, m.y, g.y, d0.y, d1.y, d2.y, d3.y);
    pool_bwd1(a0.z, a1.z, a2.z, a3.z, m.z, g.z, d0.z, d1.z, d2.z, d3.z);
    pool_bwd1(a0.w, a1.w, a2.w, a3.w, m.w, g.w, d0.w, d1.w, d2.w, d3.w);
    o[0] = d0;
    o[s1] = d1;
    o[s2] = d2;
    o[s3] = d3;
  }
}

extern "C" int scan_maxpool2x2_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y,
                                       void* stream) {
  SCAN_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_forward: bad arguments");
  SCAN_CHECK_ARG((H & 1) == 0 && (W & 1) == 0 && (C & 3) == 0, "maxpool2x2_forward: H, W must be even and C %% 4 == 0");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), x, N, H, W, C, y);
  SCAN_LAUNCH_CHECK("maxpool2_fwd");
  return 0;
}

extern "C" int scan_maxpool2x2_backward(const float* x, const float* y, const float* dy, int32_t N, int32_t H,
                                        int32_t W, int32_t C, float* dx, int32_t relu_mask, void* stream) {
  SCAN_CHECK_ARG(x && y && dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_backward: bad arguments");
  SCAN_CHECK_ARG((H & 1) == 0 && (W & 1) == 0 && (C & 3) == 0, "maxpool2x2_backward: H, W must be even and C %% 4 == 0");
  const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), x, y, dy, N, H, W,
                     C, dx, relu_mask);
  SCAN_LAUNCH_CHECK("maxpool2_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------
// ResNet stem pooling: 3x3 / stride 2 / pad 1 max-pool on NHWC rows (F.max_pool2d(x, 3, 2, 1) in BaseStem.forward,
// reference backbone/resnet.py:335).  Forward only: the stem is frozen whenever FREEZE_CONV_BODY_AT >= 1 (default 2).
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                             int Ho, int Wo, float* __restrict__ y) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * Ho * Wo * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int xo = (int)(p % Wo);
    p /= Wo;
    const int yo = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = 2 * yo - 1 + dy;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * xo - 1 + dx;
        if (xx < 0 || xx >= W) continue;
        const float4 v = reinterpret_cast<const float4*>(x + (((int64_t)n * H + yy) * W + xx) * C)[c4];
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    reinterpret_cast<float4*>(y)[i] = m;
  }
}

extern "C" int scan_maxpool3x3s2_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y,
                                         void* stream) {
  SCAN_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0,
                 "maxpool3x3s2_forward: bad arguments (C %% 4 == 0 required)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), x, N, H, W, C,
                     Ho, Wo, y);
  SCAN_LAUNCH_CHECK("maxpool3s2_fwd");
  return 0;
}

// residual join of a bottleneck block: y = max(a + b, 0)  (out += identity; F.relu_(out), resnet.py:312-313).
// The backward is scan_relu_backward(dy, y) handed to both branches.
__global__ __launch_bounds__(256) void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ y, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
    float4 o;
    o.x = fmaxf(u.x + v.x, 0.f);
    o.y = fmaxf(u.y + v.y, 0.f);
    o.z = fmaxf(u.z + v.z, 0.f);
    o.w = fmaxf(u.w + v.w, 0.f);
    reinterpret_cast<float4*>(y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 << 2; i < n; ++i) y[i] = fmaxf(a[i] + b[i], 0.f);
}

extern "C" int scan_add_relu(const float* a, const float* b, float* y, int64_t n, void* stream) {
  SCAN_CHECK_ARG(a && b && y && n >= 0, "add_relu: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_relu_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, as_stream(stream), a, b, y, n);
  SCAN_LAUNCH_CHECK("add_relu");
  return 0;
}

// ------------------------------------------------------------------ FPN top-down join (reference backbone/fpn.py:62-75)
// y = lateral + F.interpolate(coarse, scale_factor=2, mode="nearest") on NHWC rows: lateral [N, 2h, 2w, C], coarse
// [N, h, w, C].  One pass (read 1.25 C floats per output pixel, write C) instead of the expand / reshape copy of the
// up-sampled map followed by an add.  Backward: d_lateral = g itself; d_coarse = the 2x2 window sums of g.
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float* __restrict__ lat, const float* __restrict__ coarse,
                                                             int N, int h, int w, int C4, float* __restrict__ y) {
  const int64_t total = (int64_t)N * 4 * h * w * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t p = i / C4;
    const int x = (int)(p % (2 * w));
    p /= 2 * w;
    const int yy = (int)(p % (2 * h));
    const int n = (int)(p / (2 * h));
    const float4 a = reinterpret_cast<const float4*>(lat)[i];
    const float4 b = reinterpret_cast<const float4*>(coarse)[(((int64_t)n * h + (yy >> 1)) * w + (x >> 1)) * C4 + c];
    reinterpret_cast<float4*>(y)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

__global__ __launch_bounds__(256) void downsample2x_sum_kernel(const float* __restrict__ g, int N, int h, int w, int C4,
                                                               float* __restrict__ d) {
  const int64_t total = (int64_t)N * h * w * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t p = i / C4;
    const int x = (int)(p % w);
    p /= w;
    const int yy = (int)(p % h);
    const int n = (int)(p / h);
    const float4* r0 = reinterpret_cast<const float4*>(g) + (((int64_t)n * 2 * h + 2 * yy) * 2 * w + 2 * x) * C4 + c;
    const float4* r1 = r0 + (int64_t)2 * w * C4;
    const float4 a = r0[0], b = r0[C4], e = r1[0], f = r1[C4];
    // the order of torch's sum over the (2, 2) window axes of the expanded view: rows first, then columns
    reinterpret_cast<float4*>(d)[i] = make_float4((a.x + b.x) + (e.x + f.x), (a.y + b.y) + (e.y + f.y),
                                                  (a.z + b.z) + (e.z + f.z), (a.w + b.w) + (e.w + f.w));
  }
}

extern "C" int scan_upsample2x_add(const float* lat, const float* coarse, int32_t N, int32_t h, int32_t w, int32_t C,
                                   float* y, void* stream) {
  SCAN_CHECK_ARG(lat && coarse && y && N >= 1 && h >= 1 && w >= 1 && C >= 4 && C % 4 == 0,
                 "upsample2x_add: bad arguments (C=%d must be a multiple of 4)", C);
  const int64_t n4 = (int64_t)N * 4 * h * w * (C / 4);
  hipLaunchKernelGGL(upsample2x_add_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, as_stream(stream), lat, coarse, N, h, w,
                     C / 4, y);
  SCAN_LAUNCH_CHECK("upsample2x_add");
  return 0;
}

extern "C" int scan_downsample2x_sum(const float* g, int32_t N, int32_t h, int32_t w, int32_t C, float* d, void* stream) {
  SCAN_CHECK_ARG(g && d && N >= 1 && h >= 1 && w >= 1 && C >= 4 && C % 4 == 0,
                 "downsample2x_sum: bad arguments (C=%d must be a multiple of 4)", C);
  const int64_t n4 = (int64_t)N * h * w * (C / 4);
  hipLaunchKernelGGL(downsample2x_sum_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, as_stream(stream), g, N, h, w, C / 4, d);
  SCAN_LAUNCH_CHECK("downsample2x_sum");
  return 0;
}

// ---- gradient of "the rows of images [i0, i1) of a pyramid" (ops.take_images): the full pyramid's gradient in ONE pass --
// rows of the taken images copied from g (the sub-pyramid's rows, level-major), every other row zero.  Written with torch ops
// this is a zero fill of the whole matrix plus one slice copy per level.
__global__ __launch_bounds__(256) void take_images_bwd_kernel(const float* __restrict__ g, scan_pyramid_t d, int i0, int i1,
                                                              int C, float* __restrict__ out) {
  const int64_t total = d.row_off[d.n_levels] * (int64_t)C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = e / C;
    const int c = (int)(e - m * C);
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
      if (i < d.n_levels && m >= d.row_off[i]) lvl = i;
    const int64_t hw = (int64_t)d.h[lvl] * d.w[lvl];
    const int64_t r = m - d.row_off[lvl];
    const int64_t n = r / hw;
    float v = 0.f;
    if (n >= i0 && n < i1) {
      // rows of the sub-pyramid before this level: (i1 - i0) images of every earlier level
      const int64_t sub_off = d.row_off[lvl] / d.n_images * (i1 - i0);
      v = g[(sub_off + (n - i0) * hw + (r - n * hw)) * C + c];
    }
    out[e] = v;
  }
}

extern "C" int scan_take_images_backward(const float* g, const scan_pyramid_t* d, int32_t i0, int32_t i1, int32_t C, float* out,
                                         void* stream) {
  SCAN_CHECK_ARG(g && d && out && C >= 1 && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && 0 <= i0 && i0 < i1 &&
                     i1 <= d->n_images,
                 "take_images_backward: bad arguments");
  const int64_t total = d->row_off[d->n_levels] * (int64_t)C;
  hipLaunchKernelGGL(take_images_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), g, *d, i0, i1, C, out);
  SCAN_LAUNCH_CHECK("take_images_backward");
  return 0;
}

// ---- column block copy between row matrices of different row pitch (round 6): dst[r][dc0 + c] = src[r][sc0 + c] for c < ncols,
// and zeros in dst columns [dc0 + ncols, dc0 + ncols + ztail).  What the torch tier spelled as F.pad of the [M, 9] act maps to
// [M, 12] (a fill plus a narrow strided copy, and a strided slice copy back in the backward) and as buf[:, C:C + e] = extra in
// front of the class branches: one coalesced pass each here.
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst,
                                                        int ld_dst, int64_t M, int ncols, int ztail) {
  const int w = ncols + ztail;
  const int64_t total = M * (int64_t)w;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / w;
    const int c = (int)(e - r * w);
    dst[r * ld_dst + c] = c < ncols ? src[r * ld_src + c] : 0.f;
  }
}

extern "C" int scan_copy_cols(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int64_t M, int32_t ncols,
                              int32_t ztail, void* stream) {
  SCAN_CHECK_ARG(M >= 0 && ncols >= 0 && ztail >= 0 && ld_src >= ncols && ld_dst >= ncols + ztail,
                 "copy_cols: bad arguments (M=%lld ncols=%d ztail=%d ld_src=%d ld_dst=%d)", (long long)M, ncols, ztail, ld_src, ld_dst);
  if (M == 0 || ncols + ztail == 0) return 0;
  SCAN_CHECK_ARG(src && dst, "copy_cols: null pointer");
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(M * (int64_t)(ncols + ztail), 256)), dim3(256), 0, as_stream(stream), src,
                     ld_src, dst, ld_dst, M, ncols, ztail);
  SCAN_LAUNCH_CHECK("copy_cols");
  return 0;
}

// ---- paradigm (prototype) update of the middle head, reference condgraph.py:586-606 with COSINE_UPDATE_ON (round 6: one launch
// instead of ~25 one-row torch launches on the step's critical small-kernel stretch).  P [K][C][T] (slot t of class k, channel c
// at (k * C + c) * T + t), pb [K][C] the batch's class means (zero rows: class not seen).  it = the paradigm counter's value:
//   slot = it - 1 if it == T else it;   cur = P[:, :, slot]
//   m    = cosine_similarity(cur, pb, dim = 1, eps = 1e-8)      (rows normalised first, like torch: x / max(||x||, eps))
//   upd  = pb.sum(1) != 0 ? cur * m + pb * (1 - m) : cur
//   it == T: slots 0 .. T-2 <- slots 1 .. T-1;   P[:, :, slot] = upd
// One workgroup per class, C <= 1024 channels over 256 threads.
__global__ __launch_bounds__(256) void paradigm_update_kernel(float* __restrict__ P, const float* __restrict__ pb, int C, int T,
                                                              int it) {
  __shared__ float red[4];
  const int k = blockIdx.x;
  const int slot = it == T ? it - 1 : it;
  float cur[4], b[4];
  float s_cc = 0.f, s_bb = 0.f, s_b = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = threadIdx.x + 256 * j;
    cur[j] = c < C ? P[((int64_t)k * C + c) * T + slot] : 0.f;
    b[j] = c < C ? pb[(int64_t)k * C + c] : 0.f;
    s_cc += cur[j] * cur[j];
    s_bb += b[j] * b[j];
    s_b += b[j];
  }
  auto bsum = [&](float v) {
    float r = block_sum_256(v, red);
    __shared__ float bc;
    if (threadIdx.x == 0) bc = r;
    __syncthreads();
    r = bc;
    __syncthreads();
    return r;
  };
  const float n_c = fmaxf(sqrtf(bsum(s_cc)), 1e-8f), n_b = fmaxf(sqrtf(bsum(s_bb)), 1e-8f);
  const bool exist = bsum(s_b) != 0.f;
  float dotp = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) dotp += (cur[j] / n_c) * (b[j] / n_b);
  const float m = bsum(dotp);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = threadIdx.x + 256 * j;
    if (c >= C) continue;
    float* row = P + ((int64_t)k * C + c) * T;
    const float upd = exist ? cur[j] * m + b[j] * (1.f - m) : cur[j];
    if (it == T)
      for (int t = 0; t + 1 < T; ++t) row[t] = row[t + 1];
    row[slot] = upd;
  }
}

extern "C" int scan_paradigm_update(float* P, const float* pb, int32_t K, int32_t C, int32_t T, int32_t it, void* stream) {
  SCAN_CHECK_ARG(P && pb && K >= 1 && C >= 1 && C <= 1024 && T >= 1 && it >= 0 && it <= T,
                 "paradigm_update: bad arguments (K=%d C=%d T=%d it=%d)", K, C, T, it);
  hipLaunchKernelGGL(paradigm_update_kernel, dim3(K), dim3(256), 0, as_stream(stream), P, pb, C, T, it);
  SCAN_LAUNCH_CHECK("paradigm_update");
  return 0;
}
