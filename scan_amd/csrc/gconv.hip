// Grouped 3x3 convolution "G groups of 128 channels -> one output channel per group" on pyramid rows.
//
// The CKA discriminator ends, per foreground class c, in conv3x3(128 -> 1) on that class's 128 hidden channels
// (reference modeling/discriminator/fcos_head_discriminator_con.py:44-62,104-121).  Stacked over the classes this is a
// conv [M, G*128] -> [M, G] with a block-diagonal weight: on the dense MFMA kernel it costs 64x the useful
// multiply-adds (G of G*G blocks are non-zero, and the narrowest tile is 64 output channels wide for 8 live ones:
// 0.67 ms forward at P3).  The useful work is one multiply-add per input element and tap -- 1.2 GMAC at P3 against
// 537 MB of input -- so these kernels are plain fp32 FMA code bound by HBM:
//
//   forward   (a) per input pixel q and group g the nine tap products T[q][g][t] = <h[q, g*128 ..], w[g][t][..]>
//                 (one 16-byte load per lane, 32-lane butterfly sums), (b) y[p][g] = bias[g] + sum_t T[p + off(t)][g][t]
//                 over the in-bounds taps -- h is read exactly once, T is 9 floats per (pixel, group)
//   dgrad     dh[q, c] = (h[q, c] > 0) * sum_t dy[q - off(t)][g] * w[g][t][c]      (reads h only for the mask)
//   wgrad     dw[g][t][c] = sum_q h[q, c] * dy[q - off(t)][g]: register accumulators over a row chunk per workgroup,
//             per-workgroup partials in a slab, fixed-order reduction (deterministic)
//
// Weight layout: the stacked conv weight [G][9][G*128] (scan_cka_stack_weights' w2): only the diagonal blocks
// w[g][t][g*128 + i] are read / written.  Lane map: thread t of 256 handles the channel quad t % (GC/4) of pixel slot
// t / (GC/4), so 32 consecutive lanes = one group of 128 channels.
#include "common.h"

#define GC_G 128  // channels per group

__device__ __forceinline__ float half_wave_sum(float v) {  // sum over the 32-lane half of a wave, result in every lane
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// pixel row m of the pyramid -> row of its neighbour (dy, dx), or -1 outside the image
__device__ __forceinline__ int64_t neighbour_row(const scan_pyramid_t& d, int64_t m, const RowCoord& rc, int oy, int ox) {
  const int y = rc.y + oy, x = rc.x + ox;
  if (y < 0 || y >= d.h[rc.lvl] || x < 0 || x >= d.w[rc.lvl]) return -1;
  return m + (int64_t)oy * d.w[rc.lvl] + ox;
}

__global__ __launch_bounds__(256) void gconv_taps_kernel(const float* __restrict__ x, int64_t M, int G, int GC,
                                                         const float* __restrict__ w, float* __restrict__ T) {
  const int quads = GC >> 2;                  // channel quads per pixel (32 per group)
  const int slots = 256 / quads;              // pixels per workgroup iteration
  const int cq = threadIdx.x % quads, slot = threadIdx.x / quads;
  const int g = cq >> 5, l32 = cq & 31;
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(w + ((int64_t)g * 9 + t) * GC + 4 * cq);
  for (int64_t q = (int64_t)blockIdx.x * slots + slot; q < M; q += (int64_t)gridDim.x * slots) {
    const float4 h = *reinterpret_cast<const float4*>(x + q * GC + 4 * cq);
    float mine = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float s = __fmaf_rn(h.x, wv[t].x, __fmaf_rn(h.y, wv[t].y, __fmaf_rn(h.z, wv[t].z, h.w * wv[t].w)));
      s = half_wave_sum(s);
      if (l32 == t) mine = s;
    }
    if (l32 < 9) T[(q * G + g) * 9 + l32] = mine;
  }
}

__global__ __launch_bounds__(256) void gconv_gather_kernel(const float* __restrict__ T, scan_pyramid_t d, int G,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int Ns) {
  const int64_t M = d.row_off[d.n_levels];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * Ns; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Ns;
    const int g = (int)(i - p * Ns);
    float acc = 0.f;
    if (g < G) {
      const RowCoord rc = decode_row(d, p);
      acc = bias != nullptr ? bias[g] : 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int64_t q = neighbour_row(d, p, rc, t / 3 - 1, t % 3 - 1);
        if (q >= 0) acc += T[(q * G + g) * 9 + t];
      }
    }
    y[i] = acc;
  }
}

__global__ __launch_bounds__(256) void gconv_dgrad_kernel(const float* __restrict__ dy, int Ns, scan_pyramid_t d, int G,
                                                          int GC, const float* __restrict__ w,
                                                          const float* __restrict__ mask, float* __restrict__ dx) {
  const int64_t M = d.row_off[d.n_levels];
  const int quads = GC >> 2, slots = 256 / quads;
  const int cq = threadIdx.x % quads, slot = threadIdx.x / quads;
  const int g = cq >> 5;
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(w + ((int64_t)g * 9 + t) * GC + 4 * cq);
  for (int64_t q = (int64_t)blockIdx.x * slots + slot; q < M; q += (int64_t)gridDim.x * slots) {
    const RowCoord rc = decode_row(d, q);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // y[p] took h[p + off(t)] * w[t], so h[q] feeds y[q - off(t)]
      const int64_t p = neighbour_row(d, q, rc, 1 - t / 3, 1 - t % 3);
      if (p >= 0) {
        const float gy = dy[p * Ns + g];
        o.x = __fmaf_rn(gy, wv[t].x, o.x);
        o.y = __fmaf_rn(gy, wv[t].y, o.y);
        o.z = __fmaf_rn(gy, wv[t].z, o.z);
        o.w = __fmaf_rn(gy, wv[t].w, o.w);
      }
    }
    if (mask != nullptr) {
      const float4 mk = *reinterpret_cast<const float4*>(mask + q * GC + 4 * cq);
      o.x = mk.x > 0.f ? o.x : 0.f;
      o.y = mk.y > 0.f ? o.y : 0.f;
      o.z = mk.z > 0.f ? o.z : 0.f;
      o.w = mk.w > 0.f ? o.w : 0.f;
    }
    *reinterpret_cast<float4*>(dx + q * GC + 4 * cq) = o;
  }
}

// slab[block][t][GC]: the block's partial of dw[g(c)][t][c]
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          int Ns, scan_pyramid_t d, int G, int GC, int64_t rows_per_block,
                                                          float* __restrict__ slab) {
  __shared__ float red[9 * 1024];
  const int64_t M = d.row_off[d.n_levels];
  const int quads = GC >> 2, slots = 256 / quads;
  const int cq = threadIdx.x % quads, slot = threadIdx.x / quads;
  const int g = cq >> 5;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t q0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t q1 = q0 + rows_per_block < M ? q0 + rows_per_block : M;
  for (int64_t q = q0 + slot; q < q1; q += slots) {
    const RowCoord rc = decode_row(d, q);
    const float4 h = *reinterpret_cast<const float4*>(x + q * GC + 4 * cq);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int64_t p = neighbour_row(d, q, rc, 1 - t / 3, 1 - t % 3);
      if (p >= 0) {
        const float gy = dy[p * Ns + g];
        acc[t].x = __fmaf_rn(gy, h.x, acc[t].x);
        acc[t].y = __fmaf_rn(gy, h.y, acc[t].y);
        acc[t].z = __fmaf_rn(gy, h.z, acc[t].z);
        acc[t].w = __fmaf_rn(gy, h.w, acc[t].w);
      }
    }
  }
  float* out = slab + (int64_t)blockIdx.x * 9 * GC;
  if (slots == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t) *reinterpret_cast<float4*>(out + (int64_t)t * GC + 4 * cq) = acc[t];
    return;
  }
  // several pixel slots per workgroup (GC < 1024): add them up through LDS in slot order
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<float4*>(red + (t * 256 + threadIdx.x) * 4) = acc[t];
  __syncthreads();
  if (slot == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float4 s = acc[t];
      for (int k = 1; k < slots; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(red + (t * 256 + k * quads + cq) * 4);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
      }
      *reinterpret_cast<float4*>(out + (int64_t)t * GC + 4 * cq) = s;
    }
  }
}

__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ slab, int blocks, int G, int GC,
                                                                 float* __restrict__ dw, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 9 * GC
  if (i >= 9 * GC) return;
  const int t = i / GC, c = i - t * GC;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += slab[(int64_t)b * 9 * GC + i];
  float* dst = dw + ((int64_t)(c / GC_G) * 9 + t) * GC + c;
  *dst = accumulate ? *dst + s : s;
}

static int gconv_check(const scan_pyramid_t* d, int G, int Cg, const char* who) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS, "%s: bad pyramid", who);
  SCAN_CHECK_ARG(Cg == GC_G, "%s: only 128 channels per group are built (got %d)", who, Cg);
  SCAN_CHECK_ARG(G == 1 || G == 2 || G == 4 || G == 8, "%s: G=%d must be 1, 2, 4 or 8", who, G);
  return 0;
}

extern "C" int64_t scan_gconv3x3_to1_ws_floats(const scan_pyramid_t* d, int32_t G, int32_t Cg) {
  if (!d) return -1;
  const int64_t M = d->row_off[d->n_levels];
  const int64_t taps = M * G * 9;                    // forward
  const int64_t slab = (int64_t)1024 * 9 * G * Cg;   // wgrad partials (at most 1024 workgroups)
  return taps > slab ? taps : slab;
}

extern "C" int scan_gconv3x3_to1_forward(const float* x, const scan_pyramid_t* d, int32_t G, int32_t Cg, const float* w,
                                         const float* bias, float* y, int32_t Ns, float* ws, void* stream) {
  if (gconv_check(d, G, Cg, "gconv3x3_to1_forward")) return -1;
  SCAN_CHECK_ARG(x && w && y && ws && Ns >= G, "gconv3x3_to1_forward: bad arguments (Ns=%d)", Ns);
  const int64_t M = d->row_off[d->n_levels];
  if (M == 0) return 0;
  const int GC = G * Cg, slots = 256 / (GC / 4);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(gconv_taps_kernel, dim3(grid_for((M + slots - 1) / slots, 1)), dim3(256), 0, st, x, M, G, GC, w, ws);
  SCAN_LAUNCH_CHECK("gconv_taps");
  hipLaunchKernelGGL(gconv_gather_kernel, dim3(grid_for(M * Ns, 256)), dim3(256), 0, st, ws, *d, G, bias, y, Ns);
  SCAN_LAUNCH_CHECK("gconv_gather");
  return 0;
}

extern "C" int scan_gconv3x3_to1_dgrad(const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G, int32_t Cg,
                                       const float* w, const float* mask, float* dx, void* stream) {
  if (gconv_check(d, G, Cg, "gconv3x3_to1_dgrad")) return -1;
  SCAN_CHECK_ARG(dy && w && dx && Ns >= G, "gconv3x3_to1_dgrad: bad arguments (Ns=%d)", Ns);
  const int64_t M = d->row_off[d->n_levels];
  if (M == 0) return 0;
  const int GC = G * Cg, slots = 256 / (GC / 4);
  hipLaunchKernelGGL(gconv_dgrad_kernel, dim3(grid_for((M + slots - 1) / slots, 1)), dim3(256), 0, as_stream(stream), dy,
                     Ns, *d, G, GC, w, mask, dx);
  SCAN_LAUNCH_CHECK("gconv_dgrad");
  return 0;
}

extern "C" int scan_gconv3x3_to1_wgrad(const float* x, const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G,
                                       int32_t Cg, float* dw, int32_t accumulate, float* ws, void* stream) {
  if (gconv_check(d, G, Cg, "gconv3x3_to1_wgrad")) return -1;
  SCAN_CHECK_ARG(x && dy && dw && ws && Ns >= G, "gconv3x3_to1_wgrad: bad arguments (Ns=%d)", Ns);
  const int64_t M = d->row_off[d->n_levels];
  const int GC = G * Cg;
  hipStream_t st = as_stream(stream);
  int blocks = 1024;
  int64_t rpb = (M + blocks - 1) / blocks;
  if (rpb < 16) rpb = 16;  // small levels: fewer, fuller workgroups
  blocks = (int)((M + rpb - 1) / rpb);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gconv_wgrad_kernel, dim3(blocks), dim3(256), 0, st, x, dy, Ns, *d, G, GC, rpb, ws);
  SCAN_LAUNCH_CHECK("gconv_wgrad");
  hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3((9 * GC + 255) / 256), dim3(256), 0, st, ws, blocks, G, GC, dw,
                     accumulate);
  SCAN_LAUNCH_CHECK("gconv_wgrad_reduce");
  return 0;
}
