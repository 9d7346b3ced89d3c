// GroupNorm(32) + ReLU on pyramid activations [M, 256] for gfx950.
// HBM-bound: every kernel streams rows with one float4 per lane (64 lanes x 16 B =
// one 1 KiB row per wave instruction), statistics are accumulated in fp64.
//
// Replaces nn.GroupNorm(32, C) + nn.ReLU in the reference towers
// (rpn/fcos/condgraph.py:99-105, rpn/fcos/fcos.py:36-49,
//  discriminator/fcos_head_discriminator_con.py:31-32).
#include "common.h"

#define GN_C 256
#define GN_RPB 128  // rows per block (4 waves x 32 rows)

// blockIdx.y = level*N + image, blockIdx.x = row chunk inside that image
struct GnBlock {
  int64_t row0;  // first pyramid row of this block
  int rows;      // rows in this block (0 => nothing to do)
  int il;        // image-level index
  int hw;
};
__device__ __forceinline__ GnBlock gn_block(const scan_pyramid_t& d) {
  GnBlock b;
  b.il = blockIdx.y;
  const int lvl = b.il / d.n_images, n = b.il - lvl * d.n_images;
  b.hw = d.h[lvl] * d.w[lvl];
  const int64_t start = (int64_t)blockIdx.x * GN_RPB;
  b.rows = 0;
  b.row0 = 0;
  if (start < b.hw) {
    b.rows = (int)((b.hw - start) < GN_RPB ? (b.hw - start) : GN_RPB);
    b.row0 = d.row_off[lvl] + (int64_t)n * b.hw + start;
  }
  return b;
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, scan_pyramid_t d, int G,
                                                       double* __restrict__ ws) {
  __shared__ double red[4][32][2];
  const GnBlock b = gn_block(d);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double s = 0.0, q = 0.0;
  for (int r = wid; r < b.rows; r += 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + (b.row0 + r) * GN_C + 4 * lane);
    s += (double)((v.x + v.y) + (v.z + v.w));
    q += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
  }
  // 8 channels per group = 2 adjacent lanes
  s += __shfl_xor(s, 1, 64);
  q += __shfl_xor(q, 1, 64);
  if ((lane & 1) == 0) {
    red[wid][lane >> 1][0] = s;
    red[wid][lane >> 1][1] = q;
  }
  __syncthreads();
  if (threadIdx.x < 32 && b.rows > 0) {
    const int g = threadIdx.x;
    const double ss = red[0][g][0] + red[1][g][0] + red[2][g][0] + red[3][g][0];
    const double qq = red[0][g][1] + red[1][g][1] + red[2][g][1] + red[3][g][1];
    atomicAdd(&ws[((int64_t)b.il * G + g) * 2 + 0], ss);
    atomicAdd(&ws[((int64_t)b.il * G + g) * 2 + 1], qq);
  }
}

// y before the ReLU.  One fixed operation order (explicit fma) shared by the forward and by the backward kernels,
// which recompute it from x to get the ReLU mask instead of reading y back from HBM.
__device__ __forceinline__ float gn_affine(float v, float mean, float rstd, float ga, float be) {
  return __fmaf_rn((v - mean) * rstd, ga, be);
}

__global__ void gn_stats_final_kernel(const double* __restrict__ ws, scan_pyramid_t d, int G, float eps,
                                      float* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = d.n_levels * d.n_images * G;
  if (i >= total) return;
  const int il = i / G, lvl = il / d.n_images;
  const double cnt = (double)d.h[lvl] * d.w[lvl] * (GN_C / G);
  const double mean = ws[2 * i] / cnt;
  double var = ws[2 * i + 1] / cnt - mean * mean;
  if (var < 0) var = 0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, scan_pyramid_t d, int G,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu,
                                                       float* __restrict__ y) {
  const GnBlock b = gn_block(d);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 1;
  const float mean = stats[((int64_t)b.il * G + g) * 2], rstd = stats[((int64_t)b.il * G + g) * 2 + 1];
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  const float4 be = *reinterpret_cast<const float4*>(beta + 4 * lane);
  for (int r = wid; r < b.rows; r += 4) {
    const int64_t off = (b.row0 + r) * GN_C + 4 * lane;
    const float4 v = *reinterpret_cast<const float4*>(x + off);
    float4 o;
    o.x = gn_affine(v.x, mean, rstd, ga.x, be.x);
    o.y = gn_affine(v.y, mean, rstd, ga.y, be.y);
    o.z = gn_affine(v.z, mean, rstd, ga.z, be.z);
    o.w = gn_affine(v.w, mean, rstd, ga.w, be.w);
    if (relu) {
      o.x = fmaxf(o.x, 0.f);
      o.y = fmaxf(o.y, 0.f);
      o.z = fmaxf(o.z, 0.f);
      o.w = fmaxf(o.w, 0.f);
    }
    *reinterpret_cast<float4*>(y + off) = o;
  }
}

// backward pass 1: group sums S1 = sum dyh*gamma, S2 = sum dyh*gamma*xhat; channel sums dgamma, dbeta
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                                            const float* __restrict__ dy, scan_pyramid_t d, int G,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, int relu,
                                                            double* __restrict__ ws_g, double* __restrict__ ws_c) {
  __shared__ double redg[4][32][2];
  __shared__ double redc[4][GN_C][2];
  const GnBlock b = gn_block(d);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 1;
  const float mean = stats[((int64_t)b.il * G + g) * 2], rstd = stats[((int64_t)b.il * G + g) * 2 + 1];
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  const float4 be = relu ? *reinterpret_cast<const float4*>(beta + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
  double s1 = 0, s2 = 0;
  float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  for (int r = wid; r < b.rows; r += 4) {
    const int64_t off = (b.row0 + r) * GN_C + 4 * lane;
    const float4 xv = *reinterpret_cast<const float4*>(x + off);
    float4 gv = *reinterpret_cast<const float4*>(dy + off);
    if (relu) {
      gv.x = gn_affine(xv.x, mean, rstd, ga.x, be.x) > 0.f ? gv.x : 0.f;
      gv.y = gn_affine(xv.y, mean, rstd, ga.y, be.y) > 0.f ? gv.y : 0.f;
      gv.z = gn_affine(xv.z, mean, rstd, ga.z, be.z) > 0.f ? gv.z : 0.f;
      gv.w = gn_affine(xv.w, mean, rstd, ga.w, be.w) > 0.f ? gv.w : 0.f;
    }
    const float h0 = (xv.x - mean) * rstd, h1 = (xv.y - mean) * rstd, h2 = (xv.z - mean) * rstd,
                h3 = (xv.w - mean) * rstd;
    dg[0] += gv.x * h0;
    dg[1] += gv.y * h1;
    dg[2] += gv.z * h2;
    dg[3] += gv.w * h3;
    db[0] += gv.x;
    db[1] += gv.y;
    db[2] += gv.z;
    db[3] += gv.w;
    const float a0 = gv.x * ga.x, a1 = gv.y * ga.y, a2 = gv.z * ga.z, a3 = gv.w * ga.w;
    s1 += (double)((a0 + a1) + (a2 + a3));
    s2 += (double)((a0 * h0 + a1 * h1) + (a2 * h2 + a3 * h3));
  }
  s1 += __shfl_xor(s1, 1, 64);
  s2 += __shfl_xor(s2, 1, 64);
  if ((lane & 1) == 0) {
    redg[wid][g][0] = s1;
    redg[wid][g][1] = s2;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    redc[wid][4 * lane + e][0] = (double)dg[e];
    redc[wid][4 * lane + e][1] = (double)db[e];
  }
  __syncthreads();
  if (b.rows > 0) {
    if (threadIdx.x < 32) {
      const int gg = threadIdx.x;
      atomicAdd(&ws_g[((int64_t)b.il * G + gg) * 2 + 0], redg[0][gg][0] + redg[1][gg][0] + redg[2][gg][0] + redg[3][gg][0]);
      atomicAdd(&ws_g[((int64_t)b.il * G + gg) * 2 + 1], redg[0][gg][1] + redg[1][gg][1] + redg[2][gg][1] + redg[3][gg][1]);
    }
    const int c = threadIdx.x;  // 256 threads == 256 channels
    atomicAdd(&ws_c[2 * c + 0], redc[0][c][0] + redc[1][c][0] + redc[2][c][0] + redc[3][c][0]);
    atomicAdd(&ws_c[2 * c + 1], redc[0][c][1] + redc[1][c][1] + redc[2][c][1] + redc[3][c][1]);
  }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                                           const float* __restrict__ dy, scan_pyramid_t d, int G,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, int relu,
                                                           const double* __restrict__ ws_g, float* __restrict__ dx) {
  const GnBlock b = gn_block(d);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 1;
  const float mean = stats[((int64_t)b.il * G + g) * 2], rstd = stats[((int64_t)b.il * G + g) * 2 + 1];
  const float inv_cnt = 1.0f / ((float)b.hw * (GN_C / G));
  const float S1 = (float)ws_g[((int64_t)b.il * G + g) * 2] * inv_cnt;
  const float S2 = (float)ws_g[((int64_t)b.il * G + g) * 2 + 1] * inv_cnt;
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  const float4 be = relu ? *reinterpret_cast<const float4*>(beta + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = wid; r < b.rows; r += 4) {
    const int64_t off = (b.row0 + r) * GN_C + 4 * lane;
    const float4 xv = *reinterpret_cast<const float4*>(x + off);
    float4 gv = *reinterpret_cast<const float4*>(dy + off);
    if (relu) {
      gv.x = gn_affine(xv.x, mean, rstd, ga.x, be.x) > 0.f ? gv.x : 0.f;
      gv.y = gn_affine(xv.y, mean, rstd, ga.y, be.y) > 0.f ? gv.y : 0.f;
      gv.z = gn_affine(xv.z, mean, rstd, ga.z, be.z) > 0.f ? gv.z : 0.f;
      gv.w = gn_affine(xv.w, mean, rstd, ga.w, be.w) > 0.f ? gv.w : 0.f;
    }
    float4 o;
    o.x = rstd * (gv.x * ga.x - (S1 + (xv.x - mean) * rstd * S2));
    o.y = rstd * (gv.y * ga.y - (S1 + (xv.y - mean) * rstd * S2));
    o.z = rstd * (gv.z * ga.z - (S1 + (xv.z - mean) * rstd * S2));
    o.w = rstd * (gv.w * ga.w - (S1 + (xv.w - mean) * rstd * S2));
    *reinterpret_cast<float4*>(dx + off) = o;
  }
}

__global__ void gn_bwd_param_final_kernel(const double* __restrict__ ws_c, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, int accumulate) {
  const int c = threadIdx.x;
  const float a = (float)ws_c[2 * c], b = (float)ws_c[2 * c + 1];
  dgamma[c] = accumulate ? dgamma[c] + a : a;
  dbeta[c] = accumulate ? dbeta[c] + b : b;
}

static int gn_check(const scan_pyramid_t* d, int C, int G, const char* who) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS, "%s: bad pyramid", who);
  SCAN_CHECK_ARG(C == GN_C && G == 32, "%s: only C=256, G=32 is built (got C=%d G=%d)", who, C, G);
  return 0;
}
static dim3 gn_grid(const scan_pyramid_t* d) {
  int maxhw = 1;
  for (int l = 0; l < d->n_levels; ++l)
    if (d->h[l] * d->w[l] > maxhw) maxhw = d->h[l] * d->w[l];
  return dim3((maxhw + GN_RPB - 1) / GN_RPB, d->n_levels * d->n_images);
}

extern "C" int64_t scan_groupnorm_ws_floats(const scan_pyramid_t* d, int32_t C, int32_t G) {
  return 2 * ((int64_t)d->n_levels * d->n_images * G * 2 + (int64_t)C * 2);
}

extern "C" int scan_groupnorm_stats(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G, float eps,
                                    float* stats, float* ws, void* stream) {
  if (gn_check(d, C, G, "groupnorm_stats")) return -1;
  SCAN_CHECK_ARG(x && stats && ws, "groupnorm_stats: null pointer");
  hipStream_t st = as_stream(stream);
  double* wsd = reinterpret_cast<double*>(ws);
  const int total = d->n_levels * d->n_images * G;
  if (hipMemsetAsync(wsd, 0, sizeof(double) * 2 * total, st) != hipSuccess) {
    scan_set_error("groupnorm_stats: memset failed");
    return -2;
  }
  hipLaunchKernelGGL(gn_stats_kernel, gn_grid(d), dim3(256), 0, st, x, *d, G, wsd);
  SCAN_LAUNCH_CHECK("gn_stats");
  hipLaunchKernelGGL(gn_stats_final_kernel, dim3((total + 255) / 256), dim3(256), 0, st, wsd, *d, G, eps, stats);
  SCAN_LAUNCH_CHECK("gn_stats_final");
  return 0;
}

// (mean, rstd) from sums accumulated elsewhere (scan_conv3x3_gn_bf16x3's epilogue): ws = fp64 [n_levels*N*G][2]
extern "C" int scan_groupnorm_stats_from_sums(const float* ws, const scan_pyramid_t* d, int32_t C, int32_t G, float eps,
                                              float* stats, void* stream) {
  if (gn_check(d, C, G, "groupnorm_stats_from_sums")) return -1;
  SCAN_CHECK_ARG(ws && stats, "groupnorm_stats_from_sums: null pointer");
  const int total = d->n_levels * d->n_images * G;
  hipLaunchKernelGGL(gn_stats_final_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const double*>(ws), *d, G, eps, stats);
  SCAN_LAUNCH_CHECK("gn_stats_final");
  return 0;
}

extern "C" int scan_groupnorm_relu_forward(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G,
                                           const float* stats, const float* gamma, const float* beta, int32_t relu,
                                           float* y, void* stream) {
  if (gn_check(d, C, G, "groupnorm_relu_forward")) return -1;
  SCAN_CHECK_ARG(x && stats && gamma && beta && y, "groupnorm_relu_forward: null pointer");
  hipLaunchKernelGGL(gn_apply_kernel, gn_grid(d), dim3(256), 0, as_stream(stream), x, *d, G, stats, gamma, beta, relu, y);
  SCAN_LAUNCH_CHECK("gn_apply");
  return 0;
}

extern "C" int scan_groupnorm_relu_backward(const float* x, const float* beta, const float* dy, const scan_pyramid_t* d,
                                            int32_t C, int32_t G, const float* stats, const float* gamma, int32_t relu,
                                            float* dx, float* dgamma, float* dbeta, int32_t accumulate, float* ws,
                                            void* stream) {
  if (gn_check(d, C, G, "groupnorm_relu_backward")) return -1;
  SCAN_CHECK_ARG(x && dy && stats && gamma && dx && dgamma && dbeta && ws && (beta || !relu),
                 "groupnorm_relu_backward: null pointer");
  hipStream_t st = as_stream(stream);
  double* ws_g = reinterpret_cast<double*>(ws);
  const int64_t ng = (int64_t)d->n_levels * d->n_images * G * 2;
  double* ws_c = ws_g + ng;
  if (hipMemsetAsync(ws_g, 0, sizeof(double) * (ng + 2 * C), st) != hipSuccess) {
    scan_set_error("groupnorm_relu_backward: memset failed");
    return -2;
  }
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, gn_grid(d), dim3(256), 0, st, x, beta, dy, *d, G, stats, gamma, relu, ws_g, ws_c);
  SCAN_LAUNCH_CHECK("gn_bwd_reduce");
  hipLaunchKernelGGL(gn_bwd_apply_kernel, gn_grid(d), dim3(256), 0, st, x, beta, dy, *d, G, stats, gamma, relu, ws_g, dx);
  SCAN_LAUNCH_CHECK("gn_bwd_apply");
  hipLaunchKernelGGL(gn_bwd_param_final_kernel, dim3(1), dim3(GN_C), 0, st, ws_c, dgamma, dbeta, accumulate);
  SCAN_LAUNCH_CHECK("gn_bwd_param_final");
  return 0;
}
