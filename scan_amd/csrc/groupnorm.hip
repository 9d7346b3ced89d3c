// GroupNorm(32) + ReLU on pyramid activations [M, 256] for gfx950.
// HBM-bound: every kernel streams rows with one float4 per lane (64 lanes x 16 B =
// one 1 KiB row per wave instruction), statistics are accumulated in fp64.
//
// Replaces nn.GroupNorm(32, C) + nn.ReLU in the reference towers
// (rpn/fcos/condgraph.py:99-105, rpn/fcos/fcos.py:36-49,
//  discriminator/fcos_head_discriminator_con.py:31-32).
#include "common.h"

#define GN_C 256
#define GN_RPB 256  // rows per block (4 waves x 64 rows, four rows in flight per wave)
#define GN_REP 8    // replicas of the per-channel fp64 sums (block b adds into replica b % 8: 8x fewer adders per address)

// Compact 1-D grid: block -> (level, image, row chunk) through a per-level prefix table, so small levels do not
// launch empty blocks (a 2-D grid sized by the largest level wasted 3 of 4 blocks on a five-level pyramid).
struct GnTab {
  int blk_off[SCAN_MAX_LEVELS + 1];  // first block of each level
  int per_img[SCAN_MAX_LEVELS];      // blocks per image on that level
};
struct GnBlock {
  int64_t row0;  // first pyramid row of this block
  int rows;      // rows in this block
  int il;        // image-level index
  int hw;
  bool first;    // first row chunk of its (level, image)
};
__device__ __forceinline__ GnBlock gn_block(const scan_pyramid_t& d, const GnTab& t) {
  GnBlock b;
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
    if (i < d.n_levels && (int)blockIdx.x >= t.blk_off[i]) lvl = i;
  const int r = blockIdx.x - t.blk_off[lvl];
  const int n = r / t.per_img[lvl], chunk = r - n * t.per_img[lvl];
  b.il = lvl * d.n_images + n;
  b.first = chunk == 0;
  b.hw = d.h[lvl] * d.w[lvl];
  const int64_t start = (int64_t)chunk * GN_RPB;
  b.rows = (int)((b.hw - start) < GN_RPB ? (b.hw - start) : GN_RPB);
  b.row0 = d.row_off[lvl] + (int64_t)n * b.hw + start;
  return b;
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, scan_pyramid_t d, GnTab tab, int G,
                                                       double* __restrict__ ws) {
  __shared__ double red[4][32][2];
  const GnBlock b = gn_block(d, tab);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double s = 0.0, q = 0.0;
  const float* base = x + b.row0 * GN_C + 4 * lane;
  for (int r = wid; r < b.rows; r += 16) {  // rows r, r+4, r+8, r+12 of this wave: four loads in flight
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = (r + 4 * u < b.rows) ? *reinterpret_cast<const float4*>(base + (int64_t)(r + 4 * u) * GN_C)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += (double)((v[u].x + v[u].y) + (v[u].z + v[u].w));
      q += (double)((v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w));
    }
  }
  // 8 channels per group = 2 adjacent lanes
  s += __shfl_xor(s, 1, 64);
  q += __shfl_xor(q, 1, 64);
  if ((lane & 1) == 0) {
    red[wid][lane >> 1][0] = s;
    red[wid][lane >> 1][1] = q;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
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

__device__ __forceinline__ float4 gn_affine4(const float4 v, float mean, float rstd, const float4 ga, const float4 be) {
  float4 o;
  o.x = gn_affine(v.x, mean, rstd, ga.x, be.x);
  o.y = gn_affine(v.y, mean, rstd, ga.y, be.y);
  o.z = gn_affine(v.z, mean, rstd, ga.z, be.z);
  o.w = gn_affine(v.w, mean, rstd, ga.w, be.w);
  return o;
}

// sums != nullptr: (mean, rstd) are derived here from the fp64 (sum, sum of squares) a conv epilogue accumulated -- the
// arithmetic of gn_stats_final_kernel, so the launch of its own goes away -- and the first block of every (level, image)
// writes them to stats_out for the backward pass.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, scan_pyramid_t d, GnTab tab, int G,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu,
                                                       float* __restrict__ y, const double* __restrict__ sums,
                                                       float eps, float* __restrict__ stats_out, int ldy) {
  const GnBlock b = gn_block(d, tab);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 1;
  float mean, rstd;
  if (sums != nullptr) {
    const int64_t i = (int64_t)b.il * G + g;
    const double cnt = (double)b.hw * (GN_C / G);
    const double m = sums[2 * i] / cnt;
    double var = sums[2 * i + 1] / cnt - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (b.first && wid == 0 && (lane & 1) == 0) {
      stats_out[2 * i] = mean;
      stats_out[2 * i + 1] = rstd;
    }
  } else {
    mean = stats[((int64_t)b.il * G + g) * 2];
    rstd = stats[((int64_t)b.il * G + g) * 2 + 1];
  }
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  const float4 be = *reinterpret_cast<const float4*>(beta + 4 * lane);
  const int64_t base = b.row0 * GN_C + 4 * lane;
  for (int r = wid; r < b.rows; r += 16) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + 4 * u < b.rows) v[u] = *reinterpret_cast<const float4*>(x + base + (int64_t)(r + 4 * u) * GN_C);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + 4 * u < b.rows) {
        float4 o = gn_affine4(v[u], mean, rstd, ga, be);
        if (relu) {
          o.x = fmaxf(o.x, 0.f);
          o.y = fmaxf(o.y, 0.f);
          o.z = fmaxf(o.z, 0.f);
          o.w = fmaxf(o.w, 0.f);
        }
        *reinterpret_cast<float4*>(y + (b.row0 + r + 4 * u) * (int64_t)ldy + 4 * lane) = o;
      }
    }
  }
}

// backward pass 1: group sums S1 = sum dyh*gamma, S2 = sum dyh*gamma*xhat; channel sums dgamma, dbeta
__device__ __forceinline__ float4 gn_masked(const float4 xv, float4 gv, float mean, float rstd, const float4 ga,
                                            const float4 be) {
  gv.x = gn_affine(xv.x, mean, rstd, ga.x, be.x) > 0.f ? gv.x : 0.f;
  gv.y = gn_affine(xv.y, mean, rstd, ga.y, be.y) > 0.f ? gv.y : 0.f;
  gv.z = gn_affine(xv.z, mean, rstd, ga.z, be.z) > 0.f ? gv.z : 0.f;
  gv.w = gn_affine(xv.w, mean, rstd, ga.w, be.w) > 0.f ? gv.w : 0.f;
  return gv;
}

__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                                            const float* __restrict__ dy, scan_pyramid_t d, GnTab tab,
                                                            int G, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, int relu,
                                                            double* __restrict__ ws_g, double* __restrict__ ws_c,
                                                            int lddy) {
  __shared__ double redg[4][32][2];
  __shared__ float redc[4][GN_C][2];
  const GnBlock b = gn_block(d, tab);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 1;
  const float mean = stats[((int64_t)b.il * G + g) * 2], rstd = stats[((int64_t)b.il * G + g) * 2 + 1];
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  const float4 be = relu ? *reinterpret_cast<const float4*>(beta + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
  double s1 = 0, s2 = 0;
  float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
  const int64_t base = b.row0 * GN_C + 4 * lane;
  for (int r = wid; r < b.rows; r += 16) {
    float4 xv[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = r + 4 * u < b.rows;
      const int64_t off = base + (int64_t)(r + 4 * u) * GN_C;
      xv[u] = ok ? *reinterpret_cast<const float4*>(x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
      gv[u] = ok ? *reinterpret_cast<const float4*>(dy + (b.row0 + r + 4 * u) * (int64_t)lddy + 4 * lane)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 gg = relu ? gn_masked(xv[u], gv[u], mean, rstd, ga, be) : gv[u];
      const float h0 = (xv[u].x - mean) * rstd, h1 = (xv[u].y - mean) * rstd, h2 = (xv[u].z - mean) * rstd,
                  h3 = (xv[u].w - mean) * rstd;
      dg[0] += gg.x * h0;
      dg[1] += gg.y * h1;
      dg[2] += gg.z * h2;
      dg[3] += gg.w * h3;
      db[0] += gg.x;
      db[1] += gg.y;
      db[2] += gg.z;
      db[3] += gg.w;
      const float a0 = gg.x * ga.x, a1 = gg.y * ga.y, a2 = gg.z * ga.z, a3 = gg.w * ga.w;
      s1 += (double)((a0 + a1) + (a2 + a3));
      s2 += (double)((a0 * h0 + a1 * h1) + (a2 * h2 + a3 * h3));
    }
  }
  s1 += __shfl_xor(s1, 1, 64);
  s2 += __shfl_xor(s2, 1, 64);
  if ((lane & 1) == 0) {
    redg[wid][g][0] = s1;
    redg[wid][g][1] = s2;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    redc[wid][4 * lane + e][0] = dg[e];
    redc[wid][4 * lane + e][1] = db[e];
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int gg = threadIdx.x;
    atomicAdd(&ws_g[((int64_t)b.il * G + gg) * 2 + 0], redg[0][gg][0] + redg[1][gg][0] + redg[2][gg][0] + redg[3][gg][0]);
    atomicAdd(&ws_g[((int64_t)b.il * G + gg) * 2 + 1], redg[0][gg][1] + redg[1][gg][1] + redg[2][gg][1] + redg[3][gg][1]);
  }
  const int c = threadIdx.x;  // 256 threads == 256 channels
  double* rep = ws_c + (int64_t)(blockIdx.x % GN_REP) * 2 * GN_C;
  atomicAdd(&rep[2 * c + 0], ((double)redc[0][c][0] + (double)redc[1][c][0]) + ((double)redc[2][c][0] + (double)redc[3][c][0]));
  atomicAdd(&rep[2 * c + 1], ((double)redc[0][c][1] + (double)redc[1][c][1]) + ((double)redc[2][c][1] + (double)redc[3][c][1]));
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ beta,
                                                           const float* __restrict__ dy, scan_pyramid_t d, GnTab tab,
                                                           int G, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, int relu,
                                                           const double* __restrict__ ws_g, float* __restrict__ dx,
                                                           const double* __restrict__ ws_c, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate, int lddy) {
  if (blockIdx.x == 0) {
    // the channel sums are final (the reduce kernel ran before this launch): fold the GN_REP replicas into dgamma /
    // dbeta here instead of in a one-block launch of its own (256 threads == 256 channels)
    const int c = threadIdx.x;
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int r = 0; r < GN_REP; ++r) {
      sa += ws_c[(int64_t)r * 2 * GN_C + 2 * c];
      sb += ws_c[(int64_t)r * 2 * GN_C + 2 * c + 1];
    }
    const float a = (float)sa, bb = (float)sb;
    dgamma[c] = accumulate ? dgamma[c] + a : a;
    dbeta[c] = accumulate ? dbeta[c] + bb : bb;
  }
  const GnBlock b = gn_block(d, tab);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int g = lane >> 1;
  const float mean = stats[((int64_t)b.il * G + g) * 2], rstd = stats[((int64_t)b.il * G + g) * 2 + 1];
  const float inv_cnt = 1.0f / ((float)b.hw * (GN_C / G));
  const float S1 = (float)ws_g[((int64_t)b.il * G + g) * 2] * inv_cnt;
  const float S2 = (float)ws_g[((int64_t)b.il * G + g) * 2 + 1] * inv_cnt;
  const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  const float4 be = relu ? *reinterpret_cast<const float4*>(beta + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t base = b.row0 * GN_C + 4 * lane;
  for (int r = wid; r < b.rows; r += 16) {
    float4 xv[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + 4 * u < b.rows) {
        const int64_t off = base + (int64_t)(r + 4 * u) * GN_C;
        xv[u] = *reinterpret_cast<const float4*>(x + off);
        gv[u] = *reinterpret_cast<const float4*>(dy + (b.row0 + r + 4 * u) * (int64_t)lddy + 4 * lane);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + 4 * u < b.rows) {
        const float4 gg = relu ? gn_masked(xv[u], gv[u], mean, rstd, ga, be) : gv[u];
        float4 o;
        o.x = rstd * (gg.x * ga.x - (S1 + (xv[u].x - mean) * rstd * S2));
        o.y = rstd * (gg.y * ga.y - (S1 + (xv[u].y - mean) * rstd * S2));
        o.z = rstd * (gg.z * ga.z - (S1 + (xv[u].z - mean) * rstd * S2));
        o.w = rstd * (gg.w * ga.w - (S1 + (xv[u].w - mean) * rstd * S2));
        *reinterpret_cast<float4*>(dx + base + (int64_t)(r + 4 * u) * GN_C) = o;
      }
    }
  }
}

static int gn_check(const scan_pyramid_t* d, int C, int G, const char* who) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS, "%s: bad pyramid", who);
  SCAN_CHECK_ARG(C == GN_C && G == 32, "%s: only C=256, G=32 is built (got C=%d G=%d)", who, C, G);
  return 0;
}
static int gn_tab(const scan_pyramid_t* d, GnTab* t) {
  t->blk_off[0] = 0;
  for (int l = 0; l < SCAN_MAX_LEVELS; ++l) {
    if (l < d->n_levels) {
      t->per_img[l] = (d->h[l] * d->w[l] + GN_RPB - 1) / GN_RPB;
      t->blk_off[l + 1] = t->blk_off[l] + d->n_images * t->per_img[l];
    } else {
      t->per_img[l] = 1;
      t->blk_off[l + 1] = t->blk_off[l];
    }
  }
  return t->blk_off[d->n_levels];
}

extern "C" int64_t scan_groupnorm_ws_floats(const scan_pyramid_t* d, int32_t C, int32_t G) {
  return 2 * ((int64_t)d->n_levels * d->n_images * G * 2 + (int64_t)C * 2 * GN_REP);
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
  GnTab tab;
  const int nblk = gn_tab(d, &tab);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk), dim3(256), 0, st, x, *d, tab, G, wsd);
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

// ld variants: y (forward) / dy (backward) may be a column slice of a wider row-major matrix -- ldy / lddy = its row
// stride in floats (a multiple of 4, >= C; the slice starts at the pointer).  The CKA discriminators normalise straight
// into the first 256 columns of the [M, 256 + Cf] class-branch input and take the gradient back from the same columns
// of its data gradient: no concatenation copy forward, no contiguous() copy backward.
static int gn_ld_check(int ld, int C, const char* who) {
  SCAN_CHECK_ARG(ld >= C && ld % 4 == 0, "%s: row stride %d must be a multiple of 4 and >= C", who, ld);
  return 0;
}

extern "C" int scan_groupnorm_relu_forward_ld(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G,
                                              const float* stats, const float* gamma, const float* beta, int32_t relu,
                                              float* y, int32_t ldy, void* stream) {
  if (gn_check(d, C, G, "groupnorm_relu_forward") || gn_ld_check(ldy, C, "groupnorm_relu_forward")) return -1;
  SCAN_CHECK_ARG(x && stats && gamma && beta && y, "groupnorm_relu_forward: null pointer");
  GnTab tab;
  const int nblk = gn_tab(d, &tab);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), x, *d, tab, G, stats, gamma, beta, relu, y,
                     (const double*)nullptr, 0.f, (float*)nullptr, ldy);
  SCAN_LAUNCH_CHECK("gn_apply");
  return 0;
}

extern "C" int scan_groupnorm_relu_forward(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G,
                                           const float* stats, const float* gamma, const float* beta, int32_t relu,
                                           float* y, void* stream) {
  return scan_groupnorm_relu_forward_ld(x, d, C, G, stats, gamma, beta, relu, y, C, stream);
}

// scan_groupnorm_stats_from_sums + scan_groupnorm_relu_forward in one launch: sums = fp64 [n_levels*N*G][2] from
// scan_conv3x3_gn_bf16x3's epilogue; stats [n_levels*N*G][2] is written for the backward pass.
extern "C" int scan_groupnorm_relu_forward_from_sums_ld(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G,
                                                        const float* sums, float eps, const float* gamma,
                                                        const float* beta, int32_t relu, float* y, int32_t ldy,
                                                        float* stats, void* stream) {
  if (gn_check(d, C, G, "groupnorm_relu_forward_from_sums") || gn_ld_check(ldy, C, "groupnorm_relu_forward_from_sums"))
    return -1;
  SCAN_CHECK_ARG(x && sums && stats && gamma && beta && y, "groupnorm_relu_forward_from_sums: null pointer");
  GnTab tab;
  const int nblk = gn_tab(d, &tab);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(nblk), dim3(256), 0, as_stream(stream), x, *d, tab, G, (const float*)nullptr,
                     gamma, beta, relu, y, reinterpret_cast<const double*>(sums), eps, stats, ldy);
  SCAN_LAUNCH_CHECK("gn_apply_from_sums");
  return 0;
}

extern "C" int scan_groupnorm_relu_forward_from_sums(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G,
                                                     const float* sums, float eps, const float* gamma,
                                                     const float* beta, int32_t relu, float* y, float* stats,
                                                     void* stream) {
  return scan_groupnorm_relu_forward_from_sums_ld(x, d, C, G, sums, eps, gamma, beta, relu, y, C, stats, stream);
}

extern "C" int scan_groupnorm_relu_backward_ld(const float* x, const float* beta, const float* dy, int32_t lddy,
                                               const scan_pyramid_t* d, int32_t C, int32_t G, const float* stats,
                                               const float* gamma, int32_t relu, float* dx, float* dgamma, float* dbeta,
                                               int32_t accumulate, float* ws, void* stream) {
  if (gn_check(d, C, G, "groupnorm_relu_backward") || gn_ld_check(lddy, C, "groupnorm_relu_backward")) return -1;
  SCAN_CHECK_ARG(x && dy && stats && gamma && dx && dgamma && dbeta && ws && (beta || !relu),
                 "groupnorm_relu_backward: null pointer");
  hipStream_t st = as_stream(stream);
  double* ws_g = reinterpret_cast<double*>(ws);
  const int64_t ng = (int64_t)d->n_levels * d->n_images * G * 2;
  double* ws_c = ws_g + ng;
  // accumulate bit 1: ws arrives cleared (a slice of the caller's once-per-iteration cleared buffer)
  if (!(accumulate & 2) && hipMemsetAsync(ws_g, 0, sizeof(double) * (ng + 2 * C * GN_REP), st) != hipSuccess) {
    scan_set_error("groupnorm_relu_backward: memset failed");
    return -2;
  }
  GnTab tab;
  const int nblk = gn_tab(d, &tab);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, x, beta, dy, *d, tab, G, stats, gamma, relu, ws_g, ws_c,
                     lddy);
  SCAN_LAUNCH_CHECK("gn_bwd_reduce");
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nblk), dim3(256), 0, st, x, beta, dy, *d, tab, G, stats, gamma, relu, ws_g, dx,
                     ws_c, dgamma, dbeta, accumulate & 1, lddy);
  SCAN_LAUNCH_CHECK("gn_bwd_apply");
  return 0;
}

extern "C" int scan_groupnorm_relu_backward(const float* x, const float* beta, const float* dy, const scan_pyramid_t* d,
                                            int32_t C, int32_t G, const float* stats, const float* gamma, int32_t relu,
                                            float* dx, float* dgamma, float* dbeta, int32_t accumulate, float* ws,
                                            void* stream) {
  return scan_groupnorm_relu_backward_ld(x, beta, dy, C, d, C, G, stats, gamma, relu, dx, dgamma, dbeta, accumulate, ws,
                                         stream);
}
