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

// Sum over each 32-lane half of the wave on the DPP path (no LDS crossbar): quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror leave every lane of a 16-lane row with the row's sum; row_bcast15 then adds row 0 into row 1 and row 2 into
// row 3.  The total is valid in the lanes with (lane & 16) != 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false);
  return v + __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float half_wave_sum_hi(float v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  return v;
}

// pixel row m of the pyramid -> row of its neighbour (dy, dx), or -1 outside the image
__device__ __forceinline__ int64_t neighbour_row(const scan_pyramid_t& d, int64_t m, const RowCoord& rc, int oy, int ox) {
  const int y = rc.y + oy, x = rc.x + ox;
  if (y < 0 || y >= d.h[rc.lvl] || x < 0 || x >= d.w[rc.lvl]) return -1;
  return m + (int64_t)oy * d.w[rc.lvl] + ox;
}

#define GC_UNROLL 2   // pixels in flight per thread, backward (36 + 36 + 26 live registers per lane)
#define GC_UNROLL_F 4 // forward: 4 KiB per wave in flight

__global__ __launch_bounds__(256) void gconv_taps_kernel(const float* __restrict__ x, int64_t M, int G, int GC,
                                                         const float* __restrict__ w, float* __restrict__ T) {
  const int quads = GC >> 2;                  // channel quads per pixel (32 per group)
  const int slots = 256 / quads;              // pixels per workgroup iteration
  const int cq = threadIdx.x % quads, slot = threadIdx.x / quads;
  const int g = cq >> 5, l32 = cq & 31;
  float4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4*>(w + ((int64_t)g * 9 + t) * GC + 4 * cq);
  const int64_t stride = (int64_t)gridDim.x * slots;
  for (int64_t q0 = (int64_t)blockIdx.x * slots + slot; q0 < M; q0 += GC_UNROLL_F * stride) {
    float4 h[GC_UNROLL_F];
#pragma unroll
    for (int u = 0; u < GC_UNROLL_F; ++u) {
      const int64_t q = q0 + u * stride;
      h[u] = q < M ? *reinterpret_cast<const float4*>(x + q * GC + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < GC_UNROLL_F; ++u) {
      const int64_t q = q0 + u * stride;
      float mine = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float s = __fmaf_rn(h[u].x, wv[t].x, __fmaf_rn(h[u].y, wv[t].y, __fmaf_rn(h[u].z, wv[t].z, h[u].w * wv[t].w)));
        s = half_wave_sum_hi(s);
        if (l32 == 16 + t) mine = s;
      }
      if (q < M && l32 >= 16 && l32 < 25) T[(q * G + g) * 9 + (l32 - 16)] = mine;
    }
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

// Backward in one pass over x: per input pixel q the nine dy[q - off(t)][g] values serve both the data gradient
// (dx[q] = mask * sum_t dy_t * w[t]) and the weight gradient (acc[t] += dy_t * x[q]); x is read once.
// slab[block][t][GC]: the block's partial of dw[g(c)][t][c].
template <bool DO_DX, bool DO_DW>
__global__ __launch_bounds__(256) void gconv_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int Ns,
                                                        scan_pyramid_t d, int G, int GC, const float* __restrict__ w,
                                                        int relu_mask, float* __restrict__ dx, int64_t rows_per_block,
                                                        float* __restrict__ slab) {
  __shared__ float red[DO_DW ? 9 * 1024 : 4];
  const int64_t M = d.row_off[d.n_levels];
  const int quads = GC >> 2, slots = 256 / quads;
  const int cq = threadIdx.x % quads, slot = threadIdx.x / quads;
  const int g = cq >> 5;
  float4 wv[9], acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wv[t] = DO_DX ? *reinterpret_cast<const float4*>(w + ((int64_t)g * 9 + t) * GC + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // weight gradient alone: three pixels in flight per thread (the fused variant has no registers to spare)
  constexpr int UN = (DO_DW && !DO_DX) ? 3 : GC_UNROLL;
  // Rows are dealt to the workgroups round-robin in groups of UN * slots (grid-stride), not as one contiguous run
  // each: with contiguous runs of 128 rows x 4 KiB = 512 KiB the 1024 workgroups stream through addresses that
  // differ by a power of two, i.e. through the same HBM channels at the same time (1.8 TB/s measured; the sum a
  // workgroup accumulates does not depend on which rows it gets).  rows_per_block only sizes the grid.
  (void)rows_per_block;
  const int64_t qe = M;
  const bool need_x = DO_DW || relu_mask;
  for (int64_t q0 = (int64_t)blockIdx.x * UN * slots + slot; q0 < qe; q0 += (int64_t)gridDim.x * UN * slots) {
    float4 h[UN];
    float gy[UN][9];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      int64_t q = q0 + u * slots;
      if (slots == 1)  // one pixel per workgroup iteration: the row decode below is wave-uniform -> scalar unit
        q = ((int64_t)__builtin_amdgcn_readfirstlane((int)(q >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
      h[u] = (q < qe && need_x) ? *reinterpret_cast<const float4*>(x + q * GC + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < qe) {
        const RowCoord rc = decode_row(d, q);
        if (slots == 1) {
          // one pixel per workgroup iteration: the wave's two groups (lanes 0..31 / 32..63) need two wave-uniform dy
          // values per tap.  Lanes 0..17 fetch the 9 x 2 values with ONE load and every lane picks its own by
          // v_readlane -- as nine same-address vector loads per KiB of x the gathers cost more address-unit time
          // than the x stream itself (418 -> 234 us at P3 without them).
          const int g0 = 2 * (int)(threadIdx.x >> 6);
          const int l = threadIdx.x & 63;
          const int tl = l % 9, gl = l / 9;
          const int64_t pl = l < 18 ? neighbour_row(d, q, rc, 1 - tl / 3, 1 - tl % 3) : -1;
          const float v = pl >= 0 ? dy[pl * Ns + g0 + gl] : 0.f;
          const int vb = __builtin_bit_cast(int, v);
          const bool hi = (l & 32) != 0;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vb, t));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vb, 9 + t));
            gy[u][t] = hi ? b : a;
          }
        } else {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            // y[p] took x[p + off(t)] * w[t], so x[q] feeds y[q - off(t)]
            const int64_t p = neighbour_row(d, q, rc, 1 - t / 3, 1 - t % 3);
            gy[u][t] = p >= 0 ? dy[p * Ns + g] : 0.f;
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 9; ++t) gy[u][t] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t q = q0 + u * slots;
      if (DO_DW) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          acc[t].x = __fmaf_rn(gy[u][t], h[u].x, acc[t].x);
          acc[t].y = __fmaf_rn(gy[u][t], h[u].y, acc[t].y);
          acc[t].z = __fmaf_rn(gy[u][t], h[u].z, acc[t].z);
          acc[t].w = __fmaf_rn(gy[u][t], h[u].w, acc[t].w);
        }
      }
      if (DO_DX && q < qe) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          o.x = __fmaf_rn(gy[u][t], wv[t].x, o.x);
          o.y = __fmaf_rn(gy[u][t], wv[t].y, o.y);
          o.z = __fmaf_rn(gy[u][t], wv[t].z, o.z);
          o.w = __fmaf_rn(gy[u][t], wv[t].w, o.w);
        }
        if (relu_mask) {
          o.x = h[u].x > 0.f ? o.x : 0.f;
          o.y = h[u].y > 0.f ? o.y : 0.f;
          o.z = h[u].z > 0.f ? o.z : 0.f;
          o.w = h[u].w > 0.f ? o.w : 0.f;
        }
        *reinterpret_cast<float4*>(dx + q * GC + 4 * cq) = o;
      }
    }
  }
  if (!DO_DW) return;
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

// ---- matrix-core versions (v_mfma_f32_16x16x4_f32, exact fp32 products like the VALU kernels) -----------------------
// Forward taps as a GEMM per (16 consecutive pixels, group): [16 px x 128 ch] . [128 ch x 9 taps] -- the layout of the
// dynamic-conv forward kernel: a lane holds eight float4 of ITS pixel row (channels 16 j + 4 q .. + 3), which are four
// consecutive k-steps each, so the 128-long dot products need no cross-lane sums at all and a wave keeps 8 KiB of loads
// in flight.  relu_bits (optional): bit 4 j + e of word [(row * G + g) * 4 + q] = (x[row][g * 128 + 16 j + 4 q + e] > 0),
// the producer's ReLU mask for the backward pass (1/32 of re-reading x).
typedef float f32x4g __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gconv_taps_mfma_kernel(const float* __restrict__ x, int64_t M, int G, int GC,
                                                              const float* __restrict__ w, float* __restrict__ T,
                                                              uint32_t* __restrict__ relu_bits) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, q = lane >> 4;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;  // a multiple of G (grid sizes are multiples of 2)
  const int g = (int)(wave_id % G);
  float4 bw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    bw[j] = col < 9 ? *reinterpret_cast<const float4*>(w + ((int64_t)g * 9 + col) * GC + g * GC_G + 16 * j + 4 * q)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t groups = (M + 15) / 16;
  for (int64_t gidx = wave_id / G; gidx < groups; gidx += n_waves / G) {
    const int64_t row = gidx * 16 + col;
    float4 av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      av[j] = row < M ? *reinterpret_cast<const float4*>(x + row * GC + g * GC_G + 16 * j + 4 * q)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4g acc = {0.f, 0.f, 0.f, 0.f};
    uint32_t bits = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bw[j].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bw[j].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bw[j].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bw[j].w, acc, 0, 0, 0);
      bits |= (av[j].x > 0.f ? 1u : 0u) << (4 * j) | (av[j].y > 0.f ? 2u : 0u) << (4 * j) |
              (av[j].z > 0.f ? 4u : 0u) << (4 * j) | (av[j].w > 0.f ? 8u : 0u) << (4 * j);
    }
    if (relu_bits != nullptr && row < M) relu_bits[(row * G + g) * 4 + q] = bits;
    // C/D map (16x16): column = lane & 15 = tap, row = 4 * (lane >> 4) + reg = pixel of the group
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t orow = gidx * 16 + 4 * q + r;
      if (col < 9 && orow < M) T[(orow * G + g) * 9 + col] = acc[r];
    }
  }
}

// Data gradient as a GEMM per (16 consecutive pixels, group): dx^T [128 ch x 16 px] = w^T [128 ch x 9 taps] . gy [9 x 16]
// with gy[t][px] = dy[px - off(t)][g] gathered by the lanes (three per lane).  A lane ends up with four consecutive
// channels of its pixel per 16-channel tile -- the float4 layout of the forward loads, so the ReLU bits of the forward
// (or x itself) mask the result in place.
__global__ __launch_bounds__(256) void gconv_dx_mfma_kernel(const float* __restrict__ dy, int Ns, scan_pyramid_t d, int G,
                                                            int GC, const float* __restrict__ w,
                                                            const float* __restrict__ xmask,
                                                            const uint32_t* __restrict__ relu_bits,
                                                            float* __restrict__ dx) {
  const int64_t M = d.row_off[d.n_levels];
  const int lane = threadIdx.x & 63;
  const int col = lane & 15, q = lane >> 4;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  const int g = (int)(wave_id % G);
  // A operand: row = channel `col` of tile, k = tap 4 s + q
  float aw[8][3];
#pragma unroll
  for (int tile = 0; tile < 8; ++tile)
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
      const int t = 4 * s_ + q;
      aw[tile][s_] = t < 9 ? w[((int64_t)g * 9 + t) * GC + g * GC_G + 16 * tile + col] : 0.f;
    }
  const int64_t groups = (M + 15) / 16;
  for (int64_t gidx = wave_id / G; gidx < groups; gidx += n_waves / G) {
    const int64_t row = gidx * 16 + col;  // B operand / output: this lane's pixel
    float gy[3] = {0.f, 0.f, 0.f};
    if (row < M) {
      const RowCoord rc = decode_row(d, row);
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const int t = 4 * s_ + q;
        if (t < 9) {
          // y[p] took x[p + off(t)] * w[t], so x[row] feeds y[row - off(t)]
          const int64_t p = neighbour_row(d, row, rc, 1 - t / 3, 1 - t % 3);
          if (p >= 0) gy[s_] = dy[p * Ns + g];
        }
      }
    }
    uint32_t bits = 0xffffffffu;
    float4 mk[8];
    if (relu_bits != nullptr) {
      if (row < M) bits = relu_bits[(row * G + g) * 4 + q];
    } else if (xmask != nullptr) {
#pragma unroll
      for (int tile = 0; tile < 8; ++tile)
        mk[tile] = row < M ? *reinterpret_cast<const float4*>(xmask + row * GC + g * GC_G + 16 * tile + 4 * q)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int tile = 0; tile < 8; ++tile) {
      f32x4g acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[tile][s_], gy[s_], acc, 0, 0, 0);
      // C/D map: column = lane & 15 = pixel, row = 4 * (lane >> 4) + reg = channel inside the tile
      float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (relu_bits != nullptr) {
        o.x = (bits >> (4 * tile + 0)) & 1u ? o.x : 0.f;
        o.y = (bits >> (4 * tile + 1)) & 1u ? o.y : 0.f;
        o.z = (bits >> (4 * tile + 2)) & 1u ? o.z : 0.f;
        o.w = (bits >> (4 * tile + 3)) & 1u ? o.w : 0.f;
      } else if (xmask != nullptr) {
        o.x = mk[tile].x > 0.f ? o.x : 0.f;
        o.y = mk[tile].y > 0.f ? o.y : 0.f;
        o.z = mk[tile].z > 0.f ? o.z : 0.f;
        o.w = mk[tile].w > 0.f ? o.w : 0.f;
      }
      if (row < M) *reinterpret_cast<float4*>(dx + row * GC + g * GC_G + 16 * tile + 4 * q) = o;
    }
  }
}

// slab [blocks][n] -> partial [parts][n]: part p sums its contiguous run of blocks in order (grid.y = parts)
__global__ __launch_bounds__(256) void gconv_slab_partial_kernel(const float* __restrict__ slab, int blocks, int n,
                                                                 int per_part, float* __restrict__ partial) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b0 = blockIdx.y * per_part;
  const int b1 = b0 + per_part < blocks ? b0 + per_part : blocks;
  float s = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) s += slab[(int64_t)b * n + i];
  partial[(int64_t)blockIdx.y * n + i] = s;
}

__global__ __launch_bounds__(256) void gconv_wgrad_reduce_kernel(const float* __restrict__ partial, int parts, int G, int GC,
                                                                 float* __restrict__ dw, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 9 * GC
  if (i >= 9 * GC) return;
  const int t = i / GC, c = i - t * GC;
  float s = 0.f;
  for (int b = 0; b < parts; ++b) s += partial[(int64_t)b * 9 * GC + i];
  float* dst = dw + ((int64_t)(c / GC_G) * 9 + t) * GC + c;
  *dst = accumulate ? *dst + s : s;
}

#define GC_MAX_BLOCKS 1024
#define GC_PARTS 32

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
  const int64_t slab = (int64_t)(GC_MAX_BLOCKS + GC_PARTS) * 9 * G * Cg;  // wgrad partials of <= 1024 workgroups + 32 sums
  return taps > slab ? taps : slab;
}

// scan_tune "gconv_mfma": 0 (default) = the fp32 FMA kernels (DPP half-wave sums forward; ONE backward pass over x that
// yields dx and the dw partials); 1 = tap products and data gradient on v_mfma_f32_16x16x4_f32 with the ReLU mask kept
// as bits by the forward.  Same products, different summation order.  Measured in the training step on one box
// (3 alternating runs each): FMA 68.8 / 69.1 / 68.7 ms, matrix cores 69.1 / 70.3 / 68.7 ms -- the matrix-core kernels
// are individually a little faster forward (147 vs 171 us at P3) but need two passes backward (dx 200 us + dw 235 us vs
// 480 us fused), so nothing is gained; all of them sit at 2.3-3.7 TB/s, bound by bytes in flight, not by arithmetic.
int g_scan_gconv_mfma = 0;

static int gconv_wave_grid(int64_t M, int G) {
  int64_t tasks = ((M + 15) / 16) * G;  // (16-pixel group, channel group) pairs, one per wave iteration
  int64_t blocks = (tasks + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 2) blocks = 2;
  return (int)((blocks + 1) / 2 * 2);   // even: 4 waves per block -> the wave count is a multiple of G <= 8
}

static int gconv_forward(const float* x, const scan_pyramid_t* d, int G, int Cg, const float* w, const float* bias, float* y,
                         int Ns, float* ws, uint32_t* relu_bits, hipStream_t st) {
  if (gconv_check(d, G, Cg, "gconv3x3_to1_forward")) return -1;
  SCAN_CHECK_ARG(x && w && y && ws && Ns >= G, "gconv3x3_to1_forward: bad arguments (Ns=%d)", Ns);
  const int64_t M = d->row_off[d->n_levels];
  if (M == 0) return 0;
  const int GC = G * Cg, slots = 256 / (GC / 4);
  if (g_scan_gconv_mfma || relu_bits != nullptr) {
    hipLaunchKernelGGL(gconv_taps_mfma_kernel, dim3(gconv_wave_grid(M, G)), dim3(256), 0, st, x, M, G, GC, w, ws, relu_bits);
  } else {
    hipLaunchKernelGGL(gconv_taps_kernel, dim3(grid_for((M + slots - 1) / slots, 1)), dim3(256), 0, st, x, M, G, GC, w, ws);
  }
  SCAN_LAUNCH_CHECK("gconv_taps");
  hipLaunchKernelGGL(gconv_gather_kernel, dim3(grid_for(M * Ns, 256)), dim3(256), 0, st, ws, *d, G, bias, y, Ns);
  SCAN_LAUNCH_CHECK("gconv_gather");
  return 0;
}

extern "C" int scan_gconv3x3_to1_forward(const float* x, const scan_pyramid_t* d, int32_t G, int32_t Cg, const float* w,
                                         const float* bias, float* y, int32_t Ns, float* ws, void* stream) {
  return gconv_forward(x, d, G, Cg, w, bias, y, Ns, ws, nullptr, as_stream(stream));
}

// the same, also leaving the ReLU bit mask of x behind (relu_bits: M * G * 4 words) for scan_gconv3x3_to1_backward_bits
extern "C" int scan_gconv3x3_to1_forward_bits(const float* x, const scan_pyramid_t* d, int32_t G, int32_t Cg,
                                              const float* w, const float* bias, float* y, int32_t Ns, float* ws,
                                              uint32_t* relu_bits, void* stream) {
  SCAN_CHECK_ARG(relu_bits, "gconv3x3_to1_forward_bits: null bit mask");
  return gconv_forward(x, d, G, Cg, w, bias, y, Ns, ws, relu_bits, as_stream(stream));
}

static int gconv_backward(const float* x, const float* dy, int Ns, const scan_pyramid_t* d, int G, int Cg, const float* w,
                          int relu_mask, const uint32_t* relu_bits, float* dx, float* dw, int accumulate, float* ws,
                          hipStream_t st, const char* who) {
  if (gconv_check(d, G, Cg, who)) return -1;
  SCAN_CHECK_ARG(dy && Ns >= G && (dx || dw), "%s: bad arguments (Ns=%d)", who, Ns);
  SCAN_CHECK_ARG(!dx || w, "%s: the data gradient needs the weights", who);
  SCAN_CHECK_ARG(!dw || (x && ws), "%s: the weight gradient needs x and a workspace", who);
  SCAN_CHECK_ARG(!(dx && relu_mask && !relu_bits) || x, "%s: the ReLU mask needs x or its bit mask", who);
  const int64_t M = d->row_off[d->n_levels];
  if (M == 0) return 0;
  const int GC = G * Cg;
  int blocks = GC_MAX_BLOCKS;
  int64_t rpb = (M + blocks - 1) / blocks;
  if (rpb < 16) rpb = 16;  // small levels: fewer, fuller workgroups
  blocks = (int)((M + rpb - 1) / rpb);
  const bool mfma = g_scan_gconv_mfma || relu_bits != nullptr;
  if (dx && mfma) {
    hipLaunchKernelGGL(gconv_dx_mfma_kernel, dim3(gconv_wave_grid(M, G)), dim3(256), 0, st, dy, Ns, *d, G, GC, w,
                       (relu_mask && !relu_bits) ? x : (const float*)nullptr, relu_bits, dx);
    SCAN_LAUNCH_CHECK(who);
    if (dw) hipLaunchKernelGGL((gconv_bwd_kernel<false, true>), dim3(blocks), dim3(256), 0, st, x, dy, Ns, *d, G, GC, w, 0,
                               dx, rpb, ws);
  } else if (dx && dw) {
    hipLaunchKernelGGL((gconv_bwd_kernel<true, true>), dim3(blocks), dim3(256), 0, st, x, dy, Ns, *d, G, GC, w, relu_mask, dx,
                       rpb, ws);
  } else if (dx) {
    hipLaunchKernelGGL((gconv_bwd_kernel<true, false>), dim3(blocks), dim3(256), 0, st, x, dy, Ns, *d, G, GC, w, relu_mask, dx,
                       rpb, ws);
  } else {
    hipLaunchKernelGGL((gconv_bwd_kernel<false, true>), dim3(blocks), dim3(256), 0, st, x, dy, Ns, *d, G, GC, w, 0, dx, rpb,
                       ws);
  }
  SCAN_LAUNCH_CHECK(who);
  if (dw) {
    // fixed-order two-level sum of the per-workgroup partials: 32 runs of consecutive blocks, then the 32 run sums
    const int n = 9 * GC, per = (blocks + GC_PARTS - 1) / GC_PARTS, parts = (blocks + per - 1) / per;
    float* partial = ws + (int64_t)GC_MAX_BLOCKS * n;
    hipLaunchKernelGGL(gconv_slab_partial_kernel, dim3((n + 255) / 256, parts), dim3(256), 0, st, ws, blocks, n, per,
                       partial);
    SCAN_LAUNCH_CHECK("gconv_slab_partial");
    hipLaunchKernelGGL(gconv_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, partial, parts, G, GC, dw,
                       accumulate);
    SCAN_LAUNCH_CHECK("gconv_wgrad_reduce");
  }
  return 0;
}

extern "C" int scan_gconv3x3_to1_dgrad(const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G, int32_t Cg,
                                       const float* w, const float* mask, float* dx, void* stream) {
  SCAN_CHECK_ARG(dx, "gconv3x3_to1_dgrad: null output");
  return gconv_backward(mask, dy, Ns, d, G, Cg, w, mask != nullptr, nullptr, dx, nullptr, 0, nullptr, as_stream(stream),
                        "gconv3x3_to1_dgrad");
}

extern "C" int scan_gconv3x3_to1_wgrad(const float* x, const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G,
                                       int32_t Cg, float* dw, int32_t accumulate, float* ws, void* stream) {
  SCAN_CHECK_ARG(dw, "gconv3x3_to1_wgrad: null output");
  return gconv_backward(x, dy, Ns, d, G, Cg, nullptr, 0, nullptr, nullptr, dw, accumulate, ws, as_stream(stream),
                        "gconv3x3_to1_wgrad");
}

// both gradients (relu_mask != 0: dx is multiplied by (x > 0), the producer's deferred ReLU)
extern "C" int scan_gconv3x3_to1_backward(const float* x, const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G,
                                          int32_t Cg, const float* w, int32_t relu_mask, float* dx, float* dw,
                                          int32_t accumulate, float* ws, void* stream) {
  SCAN_CHECK_ARG(x && dx && dw, "gconv3x3_to1_backward: null pointer");
  return gconv_backward(x, dy, Ns, d, G, Cg, w, relu_mask, nullptr, dx, dw, accumulate, ws, as_stream(stream),
                        "gconv3x3_to1_backward");
}

// both gradients with the ReLU mask taken from the forward's bit mask (x is read once, by the weight gradient)
extern "C" int scan_gconv3x3_to1_backward_bits(const float* x, const float* dy, int32_t Ns, const scan_pyramid_t* d,
                                               int32_t G, int32_t Cg, const float* w, const uint32_t* relu_bits, float* dx,
                                               float* dw, int32_t accumulate, float* ws, void* stream) {
  SCAN_CHECK_ARG(x && dx && dw && relu_bits, "gconv3x3_to1_backward_bits: null pointer");
  return gconv_backward(x, dy, Ns, d, G, Cg, w, 1, relu_bits, dx, dw, accumulate, ws, as_stream(stream),
                        "gconv3x3_to1_backward_bits");
}
