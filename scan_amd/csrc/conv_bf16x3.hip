// 3x3 / stride-1 convolution (forward and data-gradient) on gfx950 matrix cores with fp32-grade
// accuracy from bf16 MFMA: every fp32 operand x is split as x = hi + lo (two bf16 values, 16
// mantissa bits together) and the product is accumulated in fp32 as
//      hi_a*hi_b + hi_a*lo_b + lo_a*hi_b          (3 x v_mfma_f32_32x32x16_bf16)
// bf16 x bf16 products are exact in fp32, so the only error is the dropped lo*lo term and the split
// truncation (~2^-17 relative per operand): measured loss error of a full DA iteration vs the fp32
// reference is 2e-6 relative.  bf16 MFMA runs 16x the fp32-MFMA rate on CDNA4, so three
// passes are ~5x faster than v_mfma_f32_32x32x2_f32.
//
// This file holds the FIRST generation of the forward / data-gradient kernel (v_mfma_f32_32x32x16_bf16; since round 2
// the launches go to conv_bf16x3_v2.hip unless scan_tune("conv_v2", 0) -- it stays as the A/B reference and as the
// arithmetic the second kernel is tested against), the weight split, and the weight-gradient kernels.
//
// Structure of the first-generation forward kernel (per 512-thread workgroup, one workgroup per CU: 95 KB of LDS):
//   output tile   16 x 16 pixels (one image, one pyramid level) x BN = 128 or 256 output channels
//                 (8 x 16 pixels x 64 channels with 256 threads for Cout <= 64)
//   K loop        input channels in chunks of 32; per chunk the 18 x 18 x 32 input HALO patch is read from HBM/L2 ONCE
//                 as fp32, split to bf16 hi/lo while being written to LDS, and then reused by all 9 taps (the taps are
//                 pure LDS address offsets)
//   weights       pre-split once per step into bf16 hi/lo [O][9][Csw] (scan_weight_split), staged per (chunk, tap)
//                 through a double-buffered LDS tile
//   waves         4 x 2, each 64 pixels x 64 (BN = 256: 128) channels = 2 x 2 (2 x 4) MFMA tiles of 32x32
// LDS pixel rows are 80 B (64 B of data + 16 B pad) and patch rows 1536 B so the 16-byte fragment reads of consecutive
// pixels / channels fall on distinct bank groups.
//
// dgrad reuses the same kernel: dX = conv3x3(dY, W') with W'[c][t][o] = W[o][8-t][c]
// (scan_weight_split mode 1 writes the flipped + transposed copy).
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef SCAN_EXP_MFMA16
// TIMING EXPERIMENT ONLY (wrong results): every v_mfma_f32_32x32x16_bf16 is replaced by two v_mfma_f32_16x16x32_bf16 on
// 4-register slices of the same accumulator -- same MFMA cycles, same LDS traffic, same registers -- to see what clock
// the chip holds with the 16x16 shape (MI355X_MICROARCH.md, DVFS give-back item 7) before rewriting the fragment layout.
__device__ __forceinline__ f32x16 mma_exp(bf16x8 a, bf16x8 b, f32x16 c) {
  f32x4v p0 = __builtin_shufflevector(c, c, 0, 1, 2, 3), p1 = __builtin_shufflevector(c, c, 4, 5, 6, 7);
  p0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, p0, 0, 0, 0);
  p1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, p1, 0, 0, 0);
  c[0] = p0[0]; c[1] = p0[1]; c[2] = p0[2]; c[3] = p0[3];
  c[4] = p1[0]; c[5] = p1[1]; c[6] = p1[2]; c[7] = p1[3];
  return c;
}
#define SCAN_MMA(a, b, c) mma_exp(a, b, c)
#else
#define SCAN_MMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define TW 16
#define PW (TW + 2)
#define CK 32             // channels per K chunk
#define LROW 40           // bf16 elements per LDS row (32 data + 8 pad = 80 B)
#define PPITCH 768        // bf16 elements per halo-patch ROW of 18 pixels: 1440 B of data padded to 1536 B, so a
                          // fragment read that spans two tile rows (lanes 0-15 / 16-31) stays on 16 distinct
                          // 16-byte slots (5 * pixel mod 16 is a bijection only if the row step is 0 mod 16 slots)

struct TileTab {
  int tile_off[SCAN_MAX_LEVELS + 1];
  int tiles_x[SCAN_MAX_LEVELS];
  int tiles_y[SCAN_MAX_LEVELS];
};

__device__ __forceinline__ int xcd_remap3(int orig, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / 8;
}

__device__ __forceinline__ void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
  hi[0] = (__bf16)v.x;
  hi[1] = (__bf16)v.y;
  hi[2] = (__bf16)v.z;
  hi[3] = (__bf16)v.w;
  lo[0] = (__bf16)(v.x - (float)hi[0]);
  lo[1] = (__bf16)(v.y - (float)hi[1]);
  lo[2] = (__bf16)(v.z - (float)hi[2]);
  lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// BN: output channels per workgroup; TH: tile height in pixels (tile = TH x 16); NT: threads (TH * 32);
// KS: 3 (3x3, pad 1, stride 1) or 1 (1x1: no halo, one tap).  The 1x1 instance also serves the stride-2 1x1 convs of
// the ResNet bottlenecks through MAP: 0 source pixel = output pixel (same pyramid), 1 source = 2 * output (forward of
// a stride-2 conv, source pyramid sd is the finer one), 2 source = output / 2 where both coordinates are even, zero
// elsewhere (its data gradient: sd is the coarser dY pyramid).
template <int BN, int TH, int NT, int KS = 3>
__global__ __launch_bounds__(NT, NT == 512 ? 2 : 3) void conv3x3_bf16x3_kernel(
    const float* __restrict__ src, scan_pyramid_t d, int Cs, const __bf16* __restrict__ wh,
    const __bf16* __restrict__ wl, int Csw, const float* __restrict__ bias, const float* __restrict__ mask,
    float* __restrict__ dst, int Nout, int Ns, int relu, TileTab tt, int n_tiles, scan_pyramid_t sd, int map,
    double* __restrict__ gn_ws) {
  constexpr int HALO = KS / 2, NTAPS = KS * KS;
  constexpr int PH = TH + 2 * HALO;
  constexpr int PWK = TW + 2 * HALO;
  constexpr int NPATCH = PH * PWK;                // halo pixels: 180 (TH 8) or 324 (TH 16); 256 for the 1x1
  constexpr int WAVES = NT / 64;
  constexpr int WN_WAVES = BN >= 128 ? 2 : 1;     // 2 (BN=128 / 256) or 1 (BN=64)
  constexpr int WM_WAVES = WAVES / WN_WAVES;
  constexpr int TM = (TH * TW / 32) / WM_WAVES;   // 32-pixel MFMA tiles per wave
  constexpr int TN = BN / (32 * WN_WAVES);        // 32-channel MFMA tiles per wave: 2, or 4 for BN=256
  constexpr int ASLOTS = (NPATCH * 8 + NT - 1) / NT;  // float4 of the halo patch per thread per chunk
  constexpr int BSEG = BN * 4 * 2 / NT;           // 16-byte weight segments per thread per (chunk, tap)

  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* Ah = reinterpret_cast<__bf16*>(smem_raw);  // [PH][PPITCH]
  __bf16* Al = Ah + PH * PPITCH;                     // [PH][PPITCH]
  __bf16* Bs = Al + PH * PPITCH;                     // [2 buf][2 plane][BN][LROW]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int bid = xcd_remap3(blockIdx.x, gridDim.x);
  const int n_tile = bid % n_tiles;
  const int tile = bid / n_tiles;
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
    if (i < d.n_levels && tile >= tt.tile_off[i]) lvl = i;
  const int H = d.h[lvl], W = d.w[lvl];
  int t = tile - tt.tile_off[lvl];
  const int per_img = tt.tiles_x[lvl] * tt.tiles_y[lvl];
  const int img = t / per_img;
  t -= img * per_img;
  const int ty0 = (t / tt.tiles_x[lvl]) * TH, tx0 = (t % tt.tiles_x[lvl]) * TW;
  const int64_t rowbase = d.row_off[lvl] + (int64_t)img * H * W;
  const int n0 = n_tile * BN;
  const int nchunks = (Cs + CK - 1) / CK;

  // ---- A patch staging roles: 180 pixels x 8 float4 = 1440 slots, 6 per thread
  float4 ra[ASLOTS];
  auto load_a = [&](int cc) {
    const int c0 = cc * CK;
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
      const int slot = tid + NT * i;
      const int q = slot >> 3, c = c0 + 4 * (slot & 7);
      const int py = q / PWK, px = q - py * PWK;
      const int y = ty0 - HALO + py, x = tx0 - HALO + px;
      bool ok = (slot < NPATCH * 8) && y >= 0 && y < H && x >= 0 && x < W && c < Cs;
      int64_t row = rowbase + (int64_t)y * W + x;
      if (KS == 1 && map != 0) {
        const int Hs = sd.h[lvl], Ws = sd.w[lvl];
        int sy, sx;
        if (map == 1) {
          sy = 2 * y;
          sx = 2 * x;
        } else {
          ok = ok && ((y | x) & 1) == 0;
          sy = y >> 1;
          sx = x >> 1;
        }
        ok = ok && sy < Hs && sx < Ws;
        row = sd.row_off[lvl] + ((int64_t)img * Hs + sy) * Ws + sx;
      }
      ra[i] = ok ? *reinterpret_cast<const float4*>(src + row * Cs + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
      const int slot = tid + NT * i;
      if (slot < NPATCH * 8) {
        const int q = slot >> 3, c4 = slot & 7;
        const int py = q / PWK, px = q - py * PWK;
        bf16x4 hi, lo;
        split4(ra[i], hi, lo);
        *reinterpret_cast<bf16x4*>(Ah + py * PPITCH + px * LROW + 4 * c4) = hi;
        *reinterpret_cast<bf16x4*>(Al + py * PPITCH + px * LROW + 4 * c4) = lo;
      }
    }
  };
  // ---- B staging roles: BN rows x 4 segments x 2 planes
  uint4 rb[BSEG];
  auto load_b = [&](int cc, int tap) {
#pragma unroll
    for (int i = 0; i < BSEG; ++i) {
      const int slot = tid + NT * i;
      const int plane = slot / (BN * 4);
      const int rem = slot - plane * BN * 4;
      const int row = rem >> 2, seg = rem & 3;
      const int o = n0 + row, c = cc * CK + 8 * seg;
      const __bf16* base = plane ? wl : wh;
      rb[i] = (o < Nout && c < Csw) ? *reinterpret_cast<const uint4*>(base + ((int64_t)o * NTAPS + tap) * Csw + c)
                                    : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BSEG; ++i) {
      const int slot = tid + NT * i;
      const int plane = slot / (BN * 4);
      const int rem = slot - plane * BN * 4;
      const int row = rem >> 2, seg = rem & 3;
      *reinterpret_cast<uint4*>(Bs + ((buf * 2 + plane) * BN + row) * LROW + 8 * seg) = rb[i];
    }
  };

  // ---- MFMA roles
  const int wm = wid / WN_WAVES, wn = wid % WN_WAVES;
  const int lr = lane & 31, lh = lane >> 5;
  int a_off[TM];  // bf16 offset of this lane's pixel (tap 0,0) + its k-half
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int p = (wm * TM + tm) * 32 + lr;  // pixel within the 8x16 tile
    a_off[tm] = (p >> 4) * PPITCH + (p & 15) * LROW + 8 * lh;
  }
  const int b_off = (wn * 32 * TN + lr) * LROW + 8 * lh;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  load_a(0);
  load_b(0, 0);
  for (int cc = 0; cc < nchunks; ++cc) {
    __syncthreads();  // every wave is done reading the previous chunk's patch
    store_a();
    if (cc + 1 < nchunks) load_a(cc + 1);
#pragma unroll 1
    for (int tap = 0; tap < NTAPS; ++tap) {
      const int buf = (cc * NTAPS + tap) & 1;
      store_b(buf);
      if (tap < NTAPS - 1)
        load_b(cc, tap + 1);
      else if (cc + 1 < nchunks)
        load_b(cc + 1, 0);
      __syncthreads();
      const int ky = tap / KS, kx = tap - KS * ky;
      const int shift = ky * PPITCH + kx * LROW;
      const __bf16* bh = Bs + (buf * 2 + 0) * BN * LROW + b_off;
      const __bf16* bl = Bs + (buf * 2 + 1) * BN * LROW + b_off;
      if constexpr (TN > 2) {
        // 128 accumulator registers: keep only one channel tile's weight fragments live at a time
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 ah[TM], al[TM];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) {
            ah[tm] = *reinterpret_cast<const bf16x8*>(Ah + a_off[tm] + shift + 16 * s);
            al[tm] = *reinterpret_cast<const bf16x8*>(Al + a_off[tm] + shift + 16 * s);
          }
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const bf16x8 bhv = *reinterpret_cast<const bf16x8*>(bh + tn * 32 * LROW + 16 * s);
            const bf16x8 blv = *reinterpret_cast<const bf16x8*>(bl + tn * 32 * LROW + 16 * s);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
              acc[tm][tn] = SCAN_MMA(al[tm], bhv, acc[tm][tn]);
              acc[tm][tn] = SCAN_MMA(ah[tm], blv, acc[tm][tn]);
              acc[tm][tn] = SCAN_MMA(ah[tm], bhv, acc[tm][tn]);
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ah[TM], al[TM], bhv[TN], blv[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          ah[tm] = *reinterpret_cast<const bf16x8*>(Ah + a_off[tm] + shift + 16 * s);
          al[tm] = *reinterpret_cast<const bf16x8*>(Al + a_off[tm] + shift + 16 * s);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          bhv[tn] = *reinterpret_cast<const bf16x8*>(bh + tn * 32 * LROW + 16 * s);
          blv[tn] = *reinterpret_cast<const bf16x8*>(bl + tn * 32 * LROW + 16 * s);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            acc[tm][tn] = SCAN_MMA(al[tm], bhv[tn], acc[tm][tn]);
            acc[tm][tn] = SCAN_MMA(ah[tm], blv[tn], acc[tm][tn]);
            acc[tm][tn] = SCAN_MMA(ah[tm], bhv[tn], acc[tm][tn]);
          }
      }
    }
  }

  // ---- epilogue.  C/D map of 32x32: col = lane&31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int o = n0 + wn * 32 * TN + tn * 32 + lr;
    const float bv = (bias != nullptr && o < Nout) ? bias[o] : 0.f;
    float gs = 0.f, gq = 0.f;
    if (relu & 2) {
      // fused 2x2 / stride-2 max-pool (frozen VGG stages, single-level pyramid): a window's four pixels are the
      // registers r, r+1 (x, x+1) and r+8, r+9 (next row) of ONE lane, so the pooled tensor is written directly and
      // the full-resolution activation never reaches HBM
      const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 2 * q;  // 0, 2, 4, 6
          const int p = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int y = ty0 + (p >> 4), x = tx0 + (p & 15);
          if (y < H && x < W && o < Nout) {
            float v = fmaxf(fmaxf(acc[tm][tn][r], acc[tm][tn][r + 1]), fmaxf(acc[tm][tn][r + 8], acc[tm][tn][r + 9])) + bv;
            if (relu & 1) v = fmaxf(v, 0.f);
            dst[((int64_t)img * Hp * Wp + (int64_t)(y >> 1) * Wp + (x >> 1)) * Ns + o] = v;
          }
        }
      continue;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      // the ReLU mask of a data gradient: fetch the 16 values of this MFMA tile first so the loads overlap
      // (one dependent load per store serialises on the memory latency and costs ~20 us per tile)
      float mk[16];
      if (mask != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int y = ty0 + (p >> 4), x = tx0 + (p & 15);
          const bool ok = y < H && x < W && o < Nout;
          mk[r] = ok ? mask[(rowbase + (int64_t)y * W + x) * Ns + o] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int y = ty0 + (p >> 4), x = tx0 + (p & 15);
        if (y < H && x < W && o < Nout) {
          const int64_t m = rowbase + (int64_t)y * W + x;
          float v = acc[tm][tn][r] + bv;
          if (relu & 1) v = fmaxf(v, 0.f);
          if (mask != nullptr) v = (mk[r] > 0.f) ? v : 0.f;
          dst[m * Ns + o] = v;
          gs += v;
          gq += v * v;
        }
      }
    }
    if (gn_ws != nullptr) {
      // GroupNorm(32) statistics of the 256-channel output this conv feeds: sum / sum of squares per (level, image,
      // group of 8 channels = 8 adjacent lanes, both lane halves), one fp64 atomic pair per group and wave
      double ds = (double)gs, dq = (double)gq;
#pragma unroll
      for (int sh = 1; sh <= 4; sh <<= 1) {
        ds += __shfl_xor(ds, sh, 64);
        dq += __shfl_xor(dq, sh, 64);
      }
      ds += __shfl_xor(ds, 32, 64);
      dq += __shfl_xor(dq, 32, 64);
      if ((lr & 7) == 0 && lh == 0 && o < Nout) {
        const int64_t slot = ((int64_t)(lvl * d.n_images + img) * 32 + (o >> 3)) * 2;
        atomicAdd(&gn_ws[slot], ds);
        atomicAdd(&gn_ws[slot + 1], dq);
      }
    }
  }
}

// w [O][T][Cs] fp32 -> bf16 hi / lo planes.
//   mode 0: out[o][t][c]            (O rows, row length Csw >= Cs, zero padded)       -- forward
//   mode 1: out[c][T-1-t][o]        (Cs rows, row length Csw >= O, zero padded)       -- dgrad (flip + transpose)
__global__ void weight_split_kernel(const float* __restrict__ w, int O, int T, int Cs, int mode, int rows, int Csw,
                                    __bf16* __restrict__ wh, __bf16* __restrict__ wl) {
  const int64_t total = (int64_t)rows * T * Csw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % Csw);
    const int64_t rt = i / Csw;
    const int tt = (int)(rt % T);
    const int row = (int)(rt / T);
    float v = 0.f;
    if (mode == 0) {
      if (col < Cs) v = w[((int64_t)row * T + tt) * Cs + col];
    } else {
      if (col < O) v = w[((int64_t)col * T + (T - 1 - tt)) * Cs + row];
    }
    const __bf16 h = (__bf16)v;
    wh[i] = h;
    wl[i] = (__bf16)(v - (float)h);
  }
}

// tuning switch (tests / A-B measurements): 0 keeps every launch on the 128-channel instance
int g_scan_conv_bn256 = 1;
#define g_bn256 g_scan_conv_bn256
// 1 (default): forward / data-gradient launches go to the 16x16x32-MFMA kernel (conv_bf16x3_v2.hip); 0: the 32x32x16
// kernel of this file (kept for A/B measurements and as the reference the second kernel is tested against)
int g_scan_conv_v2 = 1;
int conv3x3_bf16x3_v2_launch(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                             int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout, int32_t Ns,
                             int32_t relu, void* stream, double* gn_ws);
int conv1x1_bf16x3_v2_launch(const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* wh, const void* wl,
                             int32_t Csw, const float* bias, const float* mask, float* y, const scan_pyramid_t* yd,
                             int32_t Nout, int32_t Ns, int32_t relu, int32_t map, void* stream);
static inline bool v2_ok(const void* y, const void* mask, int32_t Ns) {
  return g_scan_conv_v2 && (Ns & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
}

static void make_tiles(const scan_pyramid_t* d, TileTab* tt, int TH) {
  tt->tile_off[0] = 0;
  for (int l = 0; l < SCAN_MAX_LEVELS; ++l) {
    if (l < d->n_levels) {
      tt->tiles_x[l] = (d->w[l] + TW - 1) / TW;
      tt->tiles_y[l] = (d->h[l] + TH - 1) / TH;
      tt->tile_off[l + 1] = tt->tile_off[l] + d->n_images * tt->tiles_x[l] * tt->tiles_y[l];
    } else {
      tt->tiles_x[l] = tt->tiles_y[l] = 1;
      tt->tile_off[l + 1] = tt->tile_off[l];
    }
  }
}

extern "C" int scan_weight_split(const float* w, int32_t O, int32_t T, int32_t Cs, int32_t mode, void* wh, void* wl,
                                 int32_t Csw, void* stream) {
  SCAN_CHECK_ARG(w && wh && wl && O > 0 && T > 0 && Cs > 0, "weight_split: bad arguments");
  SCAN_CHECK_ARG(mode == 0 || mode == 1, "weight_split: mode must be 0 or 1");
  SCAN_CHECK_ARG(Csw % 8 == 0 && Csw >= (mode == 0 ? Cs : O), "weight_split: Csw=%d must be a multiple of 8 and cover the row",
                 Csw);
  const int rows = mode == 0 ? O : Cs;
  const int64_t total = (int64_t)rows * T * Csw;
  hipLaunchKernelGGL(weight_split_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), w, O, T, Cs,
                     mode, rows, Csw, reinterpret_cast<__bf16*>(wh), reinterpret_cast<__bf16*>(wl));
  SCAN_LAUNCH_CHECK("weight_split");
  return 0;
}

// y[M][Ns] = conv3x3_s1(x[M][Cs]) with pre-split weights wh/wl [Nout][9][Csw]; same pyramid in and out.
static int conv3x3_bf16x3_launch(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                                 int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout, int32_t Ns,
                                 int32_t relu, void* stream, double* gn_ws = nullptr) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1,
                 "conv3x3_bf16x3: bad pyramid");
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "conv3x3_bf16x3: Cs=%d must be a positive multiple of 4", Cs);
  SCAN_CHECK_ARG(Csw % 8 == 0 && Csw >= Cs, "conv3x3_bf16x3: Csw=%d must be a multiple of 8 and >= Cs", Csw);
  SCAN_CHECK_ARG(Nout > 0 && Ns >= Nout, "conv3x3_bf16x3: Nout=%d Ns=%d", Nout, Ns);
  SCAN_CHECK_ARG(x && wh && wl && y, "conv3x3_bf16x3: null pointer");
  if (v2_ok(y, mask, Ns)) return conv3x3_bf16x3_v2_launch(x, d, Cs, wh, wl, Csw, bias, mask, y, Nout, Ns, relu, stream, gn_ws);
  TileTab tt;
  hipStream_t st = as_stream(stream);
  const __bf16* h = reinterpret_cast<const __bf16*>(wh);
  const __bf16* l = reinterpret_cast<const __bf16*>(wl);
  if (Nout > 64) {
    // 16 x 16 pixel tiles, 512 threads (8 waves = 4 x 2).  BN = 128: each wave 64 px x 64 ch.
    // (A/B on one device, tower layer: 8x16/256 thr x2 per CU 305 TF, this 332 TF, a 3-deep software-pipelined
    //  variant of it 330 TF -- not kept.)
    // BN = 256 (each wave 64 px x 128 ch, 128 accumulator registers): the halo patch is staged and split once
    // per 256 output channels instead of once per 128, and each barrier interval carries twice the MFMA work.  Used
    // when the output channels fill 256-wide tiles and the launch still has >= 2 workgroups per CU.
    make_tiles(d, &tt, 16);
    const int tiles = tt.tile_off[d->n_levels];
    const bool wide = g_bn256 && Nout % 256 == 0 && (int64_t)tiles * (Nout / 256) >= 512;
    if (wide) {
      const int n_tiles = Nout / 256;
      const size_t sh = (size_t)(2 * 18 * PPITCH + 4 * 256 * LROW) * sizeof(__bf16);
      static bool done = false;
      if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16x3_kernel<256, 16, 512>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        done = true;
      }
      hipLaunchKernelGGL((conv3x3_bf16x3_kernel<256, 16, 512>), dim3(tiles * n_tiles), dim3(512), sh, st, x, *d, Cs, h, l,
                         Csw, bias, mask, y, Nout, Ns, relu, tt, n_tiles, *d, 0, gn_ws);
      SCAN_LAUNCH_CHECK("conv3x3_bf16x3");
      return 0;
    }
    const int n_tiles = (Nout + 127) / 128;
    const size_t sh = (size_t)(2 * 18 * PPITCH + 4 * 128 * LROW) * sizeof(__bf16);
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16x3_kernel<128, 16, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      done = true;
    }
    hipLaunchKernelGGL((conv3x3_bf16x3_kernel<128, 16, 512>), dim3(tiles * n_tiles), dim3(512), sh, st, x, *d, Cs, h, l,
                       Csw, bias, mask, y, Nout, Ns, relu, tt, n_tiles, *d, 0, gn_ws);
  } else {
    make_tiles(d, &tt, 8);
    const int tiles = tt.tile_off[d->n_levels];
    const size_t sh = (size_t)(2 * 10 * PPITCH + 4 * 64 * LROW) * sizeof(__bf16);
    hipLaunchKernelGGL((conv3x3_bf16x3_kernel<64, 8, 256>), dim3(tiles), dim3(256), sh, st, x, *d, Cs, h, l, Csw, bias,
                       mask, y, Nout, Ns, relu, tt, 1, *d, 0, gn_ws);
  }
  SCAN_LAUNCH_CHECK("conv3x3_bf16x3");
  return 0;
}

extern "C" int scan_conv3x3_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                                   int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout,
                                   int32_t Ns, int32_t relu, void* stream) {
  return conv3x3_bf16x3_launch(x, d, Cs, wh, wl, Csw, bias, mask, y, Nout, Ns, relu ? 1 : 0, stream);
}

// conv3x3 + bias whose output feeds GroupNorm(32, 256): the epilogue also accumulates the per-(level, image, group)
// sum and sum of squares into gn_ws (fp64, n_levels * n_images * 32 * 2 values, zeroed here), which
// scan_groupnorm_stats_from_sums turns into (mean, rstd) -- the separate statistics pass over y disappears.
extern "C" int scan_conv3x3_gn_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                                      int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns, float* gn_ws,
                                      void* stream) {
  SCAN_CHECK_ARG(Nout == 256 && gn_ws, "conv3x3_gn_bf16x3: needs Nout == 256 (GroupNorm(32, 256)) and a workspace");
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1, "conv3x3_gn_bf16x3: bad pyramid");
  const size_t bytes = sizeof(double) * 2 * 32 * (size_t)d->n_levels * d->n_images;
  if (hipMemsetAsync(gn_ws, 0, bytes, as_stream(stream)) != hipSuccess) {
    scan_set_error("conv3x3_gn_bf16x3: memset failed");
    return -2;
  }
  return conv3x3_bf16x3_launch(x, d, Cs, wh, wl, Csw, bias, nullptr, y, Nout, Ns, 0, stream,
                               reinterpret_cast<double*>(gn_ws));
}

// the same with the sums ADDED to gn_ws as it is: the caller cleared it (scan_amd/ops.py hands out slices of one buffer it
// clears with one memset per training iteration instead of one memset launch per call)
extern "C" int scan_conv3x3_gn_acc_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                                          int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns, float* gn_ws,
                                          void* stream) {
  SCAN_CHECK_ARG(Nout == 256 && gn_ws, "conv3x3_gn_acc_bf16x3: needs Nout == 256 (GroupNorm(32, 256)) and a workspace");
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1,
                 "conv3x3_gn_acc_bf16x3: bad pyramid");
  return conv3x3_bf16x3_launch(x, d, Cs, wh, wl, Csw, bias, nullptr, y, Nout, Ns, 0, stream,
                               reinterpret_cast<double*>(gn_ws));
}

// conv3x3 + bias (+ ReLU) + 2x2 / stride-2 max-pool in one launch: y [N, H/2, W/2, Ns] (forward only; single-level
// pyramid with even H, W).  max and the monotone bias / ReLU commute, so the result equals pooling the conv output.
extern "C" int scan_conv3x3_pool2_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh,
                                         const void* wl, int32_t Csw, const float* bias, float* y, int32_t Nout,
                                         int32_t Ns, int32_t relu, void* stream) {
  SCAN_CHECK_ARG(d && d->n_levels == 1 && (d->h[0] & 1) == 0 && (d->w[0] & 1) == 0,
                 "conv3x3_pool2_bf16x3: needs a single-level pyramid with even H and W");
  return conv3x3_bf16x3_launch(x, d, Cs, wh, wl, Csw, bias, nullptr, y, Nout, Ns, (relu ? 1 : 0) | 2, stream);
}

// y[Mo][Ns] = conv1x1(x[Mi][Cs]) with pre-split weights wh/wl [Nout][1][Csw].  map 0: stride 1 (xd == yd);
// map 1: stride 2 forward (yd = xd.conv_out(1, 2)); map 2: data gradient of a stride-2 1x1 conv (x = dY on the coarse
// pyramid xd, y = dX on the fine pyramid yd, zero where a coordinate is odd).
extern "C" int scan_conv1x1_bf16x3(const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* wh, const void* wl,
                                   int32_t Csw, const float* bias, const float* mask, float* y,
                                   const scan_pyramid_t* yd, int32_t Nout, int32_t Ns, int32_t relu, int32_t map,
                                   void* stream) {
  SCAN_CHECK_ARG(xd && yd && yd->n_levels >= 1 && yd->n_levels <= SCAN_MAX_LEVELS && yd->n_images >= 1 &&
                     xd->n_levels == yd->n_levels && xd->n_images == yd->n_images,
                 "conv1x1_bf16x3: bad pyramids");
  SCAN_CHECK_ARG(map >= 0 && map <= 2, "conv1x1_bf16x3: map=%d must be 0, 1 or 2", map);
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "conv1x1_bf16x3: Cs=%d must be a positive multiple of 4", Cs);
  SCAN_CHECK_ARG(Csw % 8 == 0 && Csw >= Cs, "conv1x1_bf16x3: Csw=%d must be a multiple of 8 and >= Cs", Csw);
  SCAN_CHECK_ARG(Nout > 0 && Ns >= Nout, "conv1x1_bf16x3: Nout=%d Ns=%d", Nout, Ns);
  SCAN_CHECK_ARG(x && wh && wl && y, "conv1x1_bf16x3: null pointer");
  for (int l = 0; l < yd->n_levels; ++l) {
    const int eh = map == 0 ? xd->h[l] : map == 1 ? (xd->h[l] - 1) / 2 + 1 : yd->h[l];
    const int ew = map == 0 ? xd->w[l] : map == 1 ? (xd->w[l] - 1) / 2 + 1 : yd->w[l];
    SCAN_CHECK_ARG(eh == yd->h[l] && ew == yd->w[l] &&
                       (map != 2 || ((yd->h[l] - 1) / 2 + 1 == xd->h[l] && (yd->w[l] - 1) / 2 + 1 == xd->w[l])),
                   "conv1x1_bf16x3: level %d sizes do not match map %d", l, map);
  }
  if (v2_ok(y, mask, Ns))
    return conv1x1_bf16x3_v2_launch(x, xd, Cs, wh, wl, Csw, bias, mask, y, yd, Nout, Ns, relu, map, stream);
  TileTab tt;
  hipStream_t st = as_stream(stream);
  const __bf16* h = reinterpret_cast<const __bf16*>(wh);
  const __bf16* l = reinterpret_cast<const __bf16*>(wl);
  if (Nout > 64) {
    make_tiles(yd, &tt, 16);
    const int tiles = tt.tile_off[yd->n_levels];
    const int n_tiles = (Nout + 127) / 128;
    const size_t sh = (size_t)(2 * 16 * PPITCH + 4 * 128 * LROW) * sizeof(__bf16);
    static bool done = false;
    if (!done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_bf16x3_kernel<128, 16, 512, 1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      done = true;
    }
    hipLaunchKernelGGL((conv3x3_bf16x3_kernel<128, 16, 512, 1>), dim3(tiles * n_tiles), dim3(512), sh, st, x, *yd, Cs, h,
                       l, Csw, bias, mask, y, Nout, Ns, relu, tt, n_tiles, *xd, map, nullptr);
  } else {
    make_tiles(yd, &tt, 8);
    const int tiles = tt.tile_off[yd->n_levels];
    const size_t sh = (size_t)(2 * 8 * PPITCH + 4 * 64 * LROW) * sizeof(__bf16);
    hipLaunchKernelGGL((conv3x3_bf16x3_kernel<64, 8, 256, 1>), dim3(tiles), dim3(256), sh, st, x, *yd, Cs, h, l, Csw,
                       bias, mask, y, Nout, Ns, relu, tt, 1, *xd, map, nullptr);
  }
  SCAN_LAUNCH_CHECK("conv1x1_bf16x3");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient of the 3x3 / stride-1 conv on the bf16 matrix cores (same hi/lo split):
//     dW[o][ky][kx][c] = sum_m dY[m][o] * X[m + (ky-1, kx-1)][c]
// GEMM with K = pixels.  Both operands live in memory pixel-major ([pixel][channel]), but an MFMA lane
// needs 8 consecutive k (pixels) of one channel: the tiles are staged in LDS in their natural
// [pixel][channel] layout and read with ds_read_b64_tr_b16, gfx950's transposing LDS read (each 16-lane
// group fetches a 4-pixel x 16-channel block and receives it channel-major), so no transpose pass exists.
//
// The kernel is bound by operand traffic, not by the matrix cores (one tap per block re-reads every
// dY / X element 18 times and saturates the fabric at ~8 TB/s), so a 512-thread workgroup computes the
// 128 (o) x 128 (c) tile for the THREE kx taps of one ky from a single staged dY chunk and one staged X row
// segment: K chunks are 32 consecutive pixels of one image row, the X segment carries one halo pixel each
// side ([x0-1, x0+32], zero outside the row), and tap kx is just "LDS row + kx" for the transposed reads.
// Deterministic split-K over chunk ranges into fp32 slabs; the splits of one tile group are placed on the
// same XCD so the re-reads of a chunk hit that XCD's L2.  The bias gradient (column sums of dY) rides along
// in the ky == 1, c-tile 0 workgroups, which already stream dY.
// LDS rows are 320 B (256 B data + 64 B pad): the four pixel rows of a transposed read then fall on
// disjoint 32-byte bank groups for both 16-lane groups of a half wave.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define WROW 160  // bf16 elements per LDS row
#define WK 64     // pixels per K chunk (one image-row segment): ~2 us of MFMA work per chunk, enough to cover the
                  // HBM latency of the next chunk's loads, which are prefetched into registers meanwhile
#define WNA (WK / 16)             // dY float4 per thread per chunk
#define WNB(KX) ((WK + (KX) - 1 + 15) / 16)  // X float4 per thread per chunk (KX/2-pixel halo each side)
#define WBUF(KX) ((WK + WK + (KX) - 1) * 2 * WROW)  // bf16 elements of the LDS stage: dY hi/lo [WK], X hi/lo [WK+KX-1]

struct ChunkTab {
  long long chunk_off[SCAN_MAX_LEVELS + 1];
  int segs[SCAN_MAX_LEVELS];
};

__device__ __forceinline__ bf16x8 tr_read8(const __bf16* p0) {
  // p0: this lane's address for pixels k..k+3; pixels k+4..k+7 are 4 rows further
  auto q0 = (__attribute__((address_space(3))) s16x4*)(p0);
  auto q1 = (__attribute__((address_space(3))) s16x4*)(p0 + 4 * WROW);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q1);
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8, r);
}

// MFMA work of one staged chunk for one wave: TMN = number of live 32-row o tiles (2, or 1 when the second lies
// beyond Nout)
template <int TMN, int KX, int TMAX = 2>
__device__ __forceinline__ void wgrad_mma(const __bf16* Ah, const __bf16* Al, const __bf16* Bh, const __bf16* Bl,
                                          int tr_off, int a_col, int b_col, f32x16 (&acc)[KX][TMAX]) {
#pragma unroll
  for (int s = 0; s < WK / 16; ++s) {
    bf16x8 ah[TMN], al[TMN];
#pragma unroll
    for (int t = 0; t < TMN; ++t) {
      const int oa = tr_off + 16 * s * WROW + a_col + 32 * t;
      ah[t] = tr_read8(Ah + oa);
      al[t] = tr_read8(Al + oa);
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
      const int ob = tr_off + (16 * s + kx) * WROW + b_col;
      const bf16x8 bh = tr_read8(Bh + ob);
      const bf16x8 bl = tr_read8(Bl + ob);
#pragma unroll
      for (int tm = 0; tm < TMN; ++tm) {
        acc[kx][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh, acc[kx][tm], 0, 0, 0);
        acc[kx][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl, acc[kx][tm], 0, 0, 0);
        acc[kx][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh, acc[kx][tm], 0, 0, 0);
      }
    }
  }
}

// KX = 3: the 3x3 / stride-1 conv (three kx taps of one ky per workgroup).  KX = 1: 1x1 convs (one tap, no halo), whose
// X operand may be gathered with stride S from the finer pyramid xd (ResNet's stride-2 1x1 convs); d is always the
// pyramid of dY, over which the K chunks run.
// NT = 512: 8 waves = 2 (o) x 4 (c), each 64 o x 32 c; NT = 1024: 16 waves = 4 x 4, each 32 o x 32 c (48 accumulator
// registers per lane, four waves per SIMD)
template <int KX, int S, int NT = 512>
__global__ __launch_bounds__(NT, NT == 1024 ? 4 : 2) void conv3x3_wgrad_bf16x3_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, scan_pyramid_t xd) {
  constexpr int HALO = KX / 2, T = KX * KX;
  constexpr int RG = NT / 32;                       // pixel-row groups of the staging roles
  constexpr int NA = WK / RG;                       // dY float4 per thread per chunk
  constexpr int NB = (WK + KX - 1 + RG - 1) / RG;   // X float4 per thread per chunk
  constexpr int WO = NT / 256;                      // waves along o
  constexpr int TMAX = 128 / (32 * WO);             // 32-row o tiles per wave (2 or 1)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware placement: ids b and b+8 share an XCD; all tiles of one split get the same b % 8
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int qq = L >> 3;
  int tile = qq % n_tiles;
  const int split = (qq / n_tiles) * 8 + xcd;
  const int c_tile = tile % c_tiles;
  tile /= c_tiles;
  const int ky = tile % KX;
  const int o_tile = tile / KX;
  const int o0 = o_tile * 128, c0 = c_tile * 128;
  const long long total_chunks = ct.chunk_off[d.n_levels];
  const long long ch_begin = (long long)split * chunks_per_split;
  long long ch_end = ch_begin + chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  const bool do_bias = (bias_slab != nullptr) && ky == HALO && c_tile == 0;

  // staging roles: float4 column q4, pixel rows rr + RG i
  const int q4 = tid & 31, rr = tid >> 5;
  float4 ra[NA], rb[NB];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_chunk = [&](long long ch) {
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
      if (i < d.n_levels && ch >= ct.chunk_off[i]) lvl = i;
    const int H = d.h[lvl], W = d.w[lvl];
    const long long r = ch - ct.chunk_off[lvl];
    const int seg = (int)(r % ct.segs[lvl]);
    const long long row = r / ct.segs[lvl];  // n * H + y
    const int y = (int)(row % H);
    const int x0 = seg * WK;
    const long long rowbase = d.row_off[lvl] + row * W;
    const int o = o0 + 4 * q4, c = c0 + 4 * q4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int k = rr + RG * i;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 + k < W) {
        const long long m = rowbase + x0 + k;
        if ((Ns & 3) == 0 && o + 3 < Ns) {
          ra[i] = *reinterpret_cast<const float4*>(dy + m * Ns + o);
        } else {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
          for (int e = 0; e < 4; ++e)
            if (o + e < Ns) t[e] = dy[m * Ns + o + e];
          ra[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
    const int yy = y + ky - HALO;
    const bool yok = yy >= 0 && yy < H && c < Cs;
    // X row: same pyramid for the 3x3; for the 1x1 the (possibly finer) pyramid xd sampled with stride S
    const int Wx = (KX == 1) ? xd.w[lvl] : W;
    const long long xrow = (KX == 1) ? xd.row_off[lvl] + ((row / H) * xd.h[lvl] + (long long)S * y) * Wx
                                     : rowbase + (long long)(ky - 1) * W;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      const int xx = x0 - HALO + j;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < WK + KX - 1 && yok && xx >= 0 && xx < W)
        rb[i] = *reinterpret_cast<const float4*>(x + (xrow + (long long)S * xx) * Cs + c);
    }
  };
  auto store_chunk = [&]() {
    __bf16* Ah = sm;
    __bf16* Al = Ah + WK * WROW;
    __bf16* Bh = Al + WK * WROW;
    __bf16* Bl = Bh + (WK + KX - 1) * WROW;
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int off = (rr + RG * i) * WROW + 4 * q4;
      split4(ra[i], hi, lo);
      *reinterpret_cast<bf16x4*>(Ah + off) = hi;
      *reinterpret_cast<bf16x4*>(Al + off) = lo;
      if (do_bias) {
        bsum.x += ra[i].x;
        bsum.y += ra[i].y;
        bsum.z += ra[i].z;
        bsum.w += ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      if (j < WK + KX - 1) {
        const int off = j * WROW + 4 * q4;
        split4(rb[i], hi, lo);
        *reinterpret_cast<bf16x4*>(Bh + off) = hi;
        *reinterpret_cast<bf16x4*>(Bl + off) = lo;
      }
    }
  };

  // 8 waves: 2 (o) x 4 (c); each wave 64 o x 32 c = 2 x 1 MFMA tiles, for each of the 3 kx taps.  Waves that share
  // a c column group sit on different SIMDs (wid % 4), so a tile with one live column group keeps two SIMDs busy
  const int wm = wid % WO, wn = wid / WO;
  const int lr = lane & 31, lh = lane >> 5;
  // transposed-read lane address: pixel row 8h + (l&15)>>2, channel column 16*((l>>4)&1) + 4*(l&3)
  const int tr_off = (8 * lh + ((lane & 15) >> 2)) * WROW + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int a_col = wm * (32 * TMAX), b_col = wn * 32;
  const bool c_act = c0 + b_col < Cs;
  const bool o_act0 = o0 + a_col < Nout, o_act1 = o0 + a_col + 32 < Nout;

  f32x16 acc[KX][TMAX];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int b = 0; b < TMAX; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // One LDS stage; the next chunk is prefetched into registers while this one computes and is converted /
  // written after the barrier that retires the current chunk.
  if (ch_begin < ch_end) load_chunk(ch_begin);
  const __bf16* Ah = sm;
  const __bf16* Al = Ah + WK * WROW;
  const __bf16* Bh = Al + WK * WROW;
  const __bf16* Bl = Bh + (WK + KX - 1) * WROW;
  for (long long ch = ch_begin; ch < ch_end; ++ch) {
#ifdef SCAN_EXP_WGRAD_NOSTAGE
    // TIMING ABLATION ONLY (wrong results): the split + LDS writes happen for the first chunk only; the loads stay
    if (ch == ch_begin) store_chunk();
    else {
#pragma unroll
      for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w));
#pragma unroll
      for (int i = 0; i < NB; ++i) asm volatile("" ::"v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
    }
#else
    store_chunk();
#endif
    if (ch + 1 < ch_end) load_chunk(ch + 1);
    __syncthreads();
    // wave-uniform skips: a wave whose 32 c columns lie beyond Cs (third c tile of Cin = 264 / 265) or whose o rows
    // lie beyond Nout (Cout = 8 / 5 / 1 heads) has nothing to contribute; it still stages and synchronises
    if (TMAX == 2 && c_act && o_act1)
      wgrad_mma<TMAX, KX, TMAX>(Ah, Al, Bh, Bl, tr_off, a_col, b_col, acc);
    else if (c_act && o_act0)
      wgrad_mma<1, KX, TMAX>(Ah, Al, Bh, Bl, tr_off, a_col, b_col, acc);
    __syncthreads();  // every wave is done with this chunk's LDS image
  }

  float* out = slab + (long long)split * Nout * T * Cs;
  const int c = c0 + b_col + lr;
#pragma unroll
  for (int kx = 0; kx < KX; ++kx)
#pragma unroll
    for (int tm = 0; tm < TMAX; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + a_col + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][tm][r];
      }

  if (do_bias) {  // column sums of this split's dY rows: reduce the 16 pixel-row groups through LDS
    float* red = reinterpret_cast<float*>(smem_raw);  // [RG][128]
    *reinterpret_cast<float4*>(red + rr * 128 + 4 * q4) = bsum;
    __syncthreads();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) sum += red[g * 128 + tid];
      if (o0 + tid < Nout) bias_slab[(long long)split * Nout + o0 + tid] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Second generation of the weight-gradient kernel: the same workgroup decomposition (128 o x 128 c x 3 kx taps of one
// ky, 64-pixel K chunks, deterministic split-K slabs) on v_mfma_f32_16x16x32_bf16 -- the shape on which the chip holds
// the higher clock in MFMA-dense loops (see conv_bf16x3_v2.hip).  A k-step is now 32 pixels: lane (col = l & 15,
// kg = l >> 4) of the A operand (dY^T, rows = 16 output channels) and of the B operand (X, columns = 16 input
// channels) needs pixels 8 kg .. 8 kg + 7 of its column, i.e. two ds_read_b64_tr_b16 on rows 8 kg + q and 8 kg + 4 + q.
// The two 16-lane groups of a half wave read rows 8 apart in the same 16 columns; with the 320-byte row pitch those
// fall on the same banks, so the 32-byte column group of a row is XOR-ed with bit 3 of the row index (applied by the
// staging writes and by every lane's read address).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wsw(int row, int col) {  // bf16 element offset of (row, col) in a swizzled stage plane
  return row * WROW + ((((col >> 4) ^ ((row >> 3) & 1)) << 4) | (col & 15));
}

__device__ __forceinline__ bf16x8 tr_read8_v2(const __bf16* p0, const __bf16* p1) {
  auto q0 = (__attribute__((address_space(3))) s16x4*)(p0);
  auto q1 = (__attribute__((address_space(3))) s16x4*)(p1);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q1);
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8, r);
}

// TO: live 16-row o tiles of this wave (4, 2 or 1)
template <int TO, int KX, int TOMAX>
__device__ __forceinline__ void wgrad_mma_v2(const __bf16* Ah, const __bf16* Al, const __bf16* Bh, const __bf16* Bl,
                                             int row_lane, int col4, int a_col, int b_col,
                                             f32x4v (&acc)[KX][TOMAX][2]) {
#pragma unroll
  for (int s = 0; s < WK / 32; ++s) {
    // dY rows of this lane: r0 = 32 s + 8 kg + q and r0 + 4 (bit 3 of both = kg & 1: one swizzle per lane)
    const int ra0 = 32 * s + row_lane, ra1 = ra0 + 4;
    bf16x8 ah[TO], al[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const int c = a_col + 16 * t + col4;
      const int o0 = wsw(ra0, c), o1 = wsw(ra1, c);
      ah[t] = tr_read8_v2(Ah + o0, Ah + o1);
      al[t] = tr_read8_v2(Al + o0, Al + o1);
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
      const int rb0 = ra0 + kx, rb1 = ra1 + kx;  // X row j <-> pixel x0 - HALO + j: tap kx is a row shift
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = b_col + 16 * tc + col4;
        const int o0 = wsw(rb0, c), o1 = wsw(rb1, c);
        const bf16x8 bh = tr_read8_v2(Bh + o0, Bh + o1);
        const bf16x8 bl = tr_read8_v2(Bl + o0, Bl + o1);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[to], bh, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bl, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bh, acc[kx][to][tc], 0, 0, 0);
      }
    }
  }
}

// NT = 512: 8 waves = 2 (o) x 4 (c), each 64 o x 32 c; NT = 1024: 16 waves = 4 x 4, each 32 o x 32 c (48 accumulator
// registers, <= 128 registers per lane: four waves per SIMD to hide the staging phases and the transposed-read latency)
// wgrad_mma_v2 with a hook after every (k-step, tap, column tile) block of MFMAs: the fourth-generation kernel issues the
// NEXT chunk's buffer loads there, one per block, so that the texture-address path works through them while the matrix
// pipe is busy (issued together before the barrier, the 72 wave-instructions of a workgroup queue up behind each other
// and the MFMA phase starts late: profiles/r03_wgrad_v4_ab.txt)
template <int TO, int KX, int TOMAX, typename F>
__device__ __forceinline__ void wgrad_mma_v2_hook(const __bf16* Ah, const __bf16* Al, const __bf16* Bh, const __bf16* Bl,
                                                  int row_lane, int col4, int a_col, int b_col,
                                                  f32x4v (&acc)[KX][TOMAX][2], F&& hook) {
  int blk = 0;
#pragma unroll
  for (int s = 0; s < WK / 32; ++s) {
    const int ra0 = 32 * s + row_lane, ra1 = ra0 + 4;
    bf16x8 ah[TO], al[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const int c = a_col + 16 * t + col4;
      const int o0 = wsw(ra0, c), o1 = wsw(ra1, c);
      ah[t] = tr_read8_v2(Ah + o0, Ah + o1);
      al[t] = tr_read8_v2(Al + o0, Al + o1);
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
      const int rb0 = ra0 + kx, rb1 = ra1 + kx;
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = b_col + 16 * tc + col4;
        const int o0 = wsw(rb0, c), o1 = wsw(rb1, c);
        const bf16x8 bh = tr_read8_v2(Bh + o0, Bh + o1);
        const bf16x8 bl = tr_read8_v2(Bl + o0, Bl + o1);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[to], bh, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bl, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bh, acc[kx][to][tc], 0, 0, 0);
        hook(blk);
        ++blk;
      }
    }
  }
}

// EXP != 0: TIMING ABLATIONS ONLY (wrong results), selected by scan_tune "wgrad_exp" and never by default:
//   1 every split-K slab walks the chunk range of slab 0 (operands hot in L2: what the memory side costs);
//   2 no MFMA phase (loads + staging + barriers only);  3 loads and staging for the first chunk only (MFMA phase +
//   barriers only);  4 global loads for the first chunk only, the split + LDS writes stay (what waiting for loads costs).
template <int KX, int S, int NT, int EXP = 0>
__global__ __launch_bounds__(NT, NT == 1024 ? 4 : 2) void conv_wgrad_bf16x3_v2_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, scan_pyramid_t xd) {
  constexpr int HALO = KX / 2, T = KX * KX;
  constexpr int RG = NT / 32;                         // pixel-row groups of the staging roles (16 or 32)
  constexpr int NA = WK / RG;                         // dY float4 per thread per chunk
  constexpr int NB = (WK + KX - 1 + RG - 1) / RG;     // X float4 per thread per chunk
  constexpr int WO = NT / 256;                        // waves along o (2 or 4)
  constexpr int TOMAX = 128 / (16 * WO);              // 16-row o tiles per wave (4 or 2)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int qq = L >> 3;
  int tile = qq % n_tiles;
  const int split = (qq / n_tiles) * 8 + xcd;
  const int c_tile = tile % c_tiles;
  tile /= c_tiles;
  const int ky = tile % KX;
  const int o_tile = tile / KX;
  const int o0 = o_tile * 128, c0 = c_tile * 128;
  const long long total_chunks = ct.chunk_off[d.n_levels];
  const long long ch_begin = EXP == 1 ? 0 : (long long)split * chunks_per_split;
  long long ch_end = ch_begin + chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  const bool do_bias = (bias_slab != nullptr) && ky == HALO && c_tile == 0;

  const int q4 = tid & 31, rr = tid >> 5;
  float4 ra[NA], rb[NB];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_chunk = [&](long long ch) {
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
      if (i < d.n_levels && ch >= ct.chunk_off[i]) lvl = i;
    const int H = d.h[lvl], W = d.w[lvl];
    const long long r = ch - ct.chunk_off[lvl];
    const int seg = (int)(r % ct.segs[lvl]);
    const long long row = r / ct.segs[lvl];  // n * H + y
    const int y = (int)(row % H);
    const int x0 = seg * WK;
    const long long rowbase = d.row_off[lvl] + row * W;
    const int o = o0 + 4 * q4, c = c0 + 4 * q4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int k = rr + RG * i;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 + k < W) {
        const long long m = rowbase + x0 + k;
        if ((Ns & 3) == 0 && o + 3 < Ns) {
          ra[i] = *reinterpret_cast<const float4*>(dy + m * Ns + o);
        } else {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
          for (int e = 0; e < 4; ++e)
            if (o + e < Ns) t[e] = dy[m * Ns + o + e];
          ra[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
    const int yy = y + ky - HALO;
    const bool yok = yy >= 0 && yy < H && c < Cs;
    const int Wx = (KX == 1) ? xd.w[lvl] : W;
    const long long xrow = (KX == 1) ? xd.row_off[lvl] + ((row / H) * xd.h[lvl] + (long long)S * y) * Wx
                                     : rowbase + (long long)(ky - 1) * W;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      const int xx = x0 - HALO + j;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < WK + KX - 1 && yok && xx >= 0 && xx < W)
        rb[i] = *reinterpret_cast<const float4*>(x + (xrow + (long long)S * xx) * Cs + c);
    }
  };
  auto store_chunk = [&]() {
    __bf16* Ah = sm;
    __bf16* Al = Ah + WK * WROW;
    __bf16* Bh = Al + WK * WROW;
    __bf16* Bl = Bh + (WK + KX - 1) * WROW;
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int off = wsw(rr + RG * i, 4 * q4);
      split4(ra[i], hi, lo);
      *reinterpret_cast<bf16x4*>(Ah + off) = hi;
      *reinterpret_cast<bf16x4*>(Al + off) = lo;
      if (do_bias) {
        bsum.x += ra[i].x;
        bsum.y += ra[i].y;
        bsum.z += ra[i].z;
        bsum.w += ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      if (j < WK + KX - 1) {
        const int off = wsw(j, 4 * q4);
        split4(rb[i], hi, lo);
        *reinterpret_cast<bf16x4*>(Bh + off) = hi;
        *reinterpret_cast<bf16x4*>(Bl + off) = lo;
      }
    }
  };

  // waves: WO (o) x 4 (c); each wave (128 / WO) o x 32 c = TOMAX x 2 MFMA tiles of 16 x 16, for each of the KX taps
  const int wm = wid % WO, wn = wid / WO;
  const int lr = lane & 15, kg = lane >> 4;
  const int row_lane = 8 * kg + (lr >> 2), col4 = 4 * (lane & 3);
  const int a_col = wm * (16 * TOMAX), b_col = wn * 32;
  const bool c_act = c0 + b_col < Cs;
  const int o_left = Nout - (o0 + a_col);  // live o rows of this wave

  f32x4v acc[KX][TOMAX][2];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int b = 0; b < TOMAX; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (ch_begin < ch_end) load_chunk(ch_begin);
  const __bf16* Ah = sm;
  const __bf16* Al = Ah + WK * WROW;
  const __bf16* Bh = Al + WK * WROW;
  const __bf16* Bl = Bh + (WK + KX - 1) * WROW;
  for (long long ch = ch_begin; ch < ch_end; ++ch) {
    if (EXP != 3 || ch == ch_begin) store_chunk();
    if (EXP != 3 && EXP != 4 && ch + 1 < ch_end) load_chunk(ch + 1);
    __syncthreads();
    // wave-uniform skips of dead tiles (third c tile of Cin = 264 / 265; Cout = 8 / 5 / 1 heads)
    if (EXP == 2) {
    } else if (TOMAX == 4 && c_act && o_left > 32)
      wgrad_mma_v2<TOMAX, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 16)
      wgrad_mma_v2<2, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 0)
      wgrad_mma_v2<1, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    __syncthreads();
  }

  // C/D map of 16x16: column = lane & 15 = input channel c, row = 4 * (lane >> 4) + reg = output channel o
  float* out = slab + (long long)split * Nout * T * Cs;
#pragma unroll
  for (int kx = 0; kx < KX; ++kx)
#pragma unroll
    for (int to = 0; to < TOMAX; ++to)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = c0 + b_col + 16 * tc + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + a_col + 16 * to + 4 * kg + r;
          if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][to][tc][r];
        }
      }

  if (do_bias) {
    float* red = reinterpret_cast<float*>(smem_raw);  // [RG][128]
    *reinterpret_cast<float4*>(red + rr * 128 + 4 * q4) = bsum;
    __syncthreads();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) sum += red[g * 128 + tid];
      if (o0 + tid < Nout) bias_slab[(long long)split * Nout + o0 + tid] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fourth generation of the weight-gradient kernel: the 16x16x32 kernel above (same tiles, same LDS image, same K order,
// bit-identical slabs) with the K-chunk STAGING rewritten for instruction count.
//
// What the ablations of the second generation showed (scan_tune "wgrad_exp", profiles/r03_wgrad_exp.txt; conv3_x, us):
// whole kernel 1690; MFMA phase + barriers alone 1100; loads + staging alone 772; data hot in L2: 1612.  The two phases
// did not overlap although the next chunk's loads are issued before the MFMA phase -- because issuing them cost ~700
// instructions per wave and chunk: 64-bit divisions to decode the chunk index into (level, image, row, segment), one
// exec-masked branch per load for the edge conditions, 64-bit address arithmetic per lane.  All eight waves run that
// code between the same two barriers, so the matrix cores idle for its whole length.  Here
//   * the chunk position (level, image, row, segment) is wave-uniform state advanced by a few scalar instructions per
//     chunk (the divisions run once per workgroup);
//   * both operands are fetched with buffer loads whose descriptor (base = first pixel of the chunk's row segment,
//     num_records = bytes up to its last valid pixel) is rebuilt per chunk from scalars: the hardware range check
//     returns zeros beyond the row end / for rows outside the image (num_records = 0), so no load is predicated;
//   * a lane's byte offsets inside a chunk are the same for every chunk (pixel k, channel column 4 q4) and live in
//     registers; columns beyond the channel count carry an out-of-range offset.
// Per chunk and lane: NA + NB buffer_load_dwordx4 with constant offsets, one v_cndmask (left image edge), no address
// arithmetic.  Needs Ns % 4 == 0, Cs % 4 == 0 and row segments below 2 GiB (64 pixels x channels x 4 B).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lvl_pick(const int32_t (&a)[SCAN_MAX_LEVELS], int l) {
  int v = a[0];
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i) v = (l == i) ? a[i] : v;
  return v;
}
__device__ __forceinline__ long long lvl_pick64(const int64_t (&a)[SCAN_MAX_LEVELS + 1], int l) {
  long long v = a[0];
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i) v = (l == i) ? (long long)a[i] : v;
  return v;
}

// buffer descriptor from wave-uniform inputs, made PROVABLY uniform for the compiler (cdna_hip_programming.md T20): a
// descriptor it cannot prove uniform gets a waterfall loop around every load
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const float* base, int bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  float* p = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int KX, int S, int NT, int IL = 0>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_bf16x3_v4_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, scan_pyramid_t xd) {
  constexpr int HALO = KX / 2, T = KX * KX;
  constexpr int RG = NT / 32;                         // pixel-row groups of the staging roles
  constexpr int NA = WK / RG;                         // dY float4 per thread per chunk
  constexpr int NB = (WK + KX - 1 + RG - 1) / RG;     // X float4 per thread per chunk
  constexpr int WO = NT / 256;                        // waves along o
  constexpr int TOMAX = 128 / (16 * WO);              // 16-row o tiles per wave
  constexpr unsigned BAD = 0x80000000u;               // a byte offset beyond every descriptor: the load returns zeros
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int qq = L >> 3;
  // integer division runs on the vector ALU: pin the (wave-uniform) quotients back into scalar registers so that
  // everything derived from them -- the chunk walk, the buffer descriptors -- stays scalar
  int tile = __builtin_amdgcn_readfirstlane(qq % n_tiles);
  const int split = __builtin_amdgcn_readfirstlane(qq / n_tiles) * 8 + xcd;
  const int c_tile = __builtin_amdgcn_readfirstlane(tile % c_tiles);
  tile = __builtin_amdgcn_readfirstlane(tile / c_tiles);
  const int ky = __builtin_amdgcn_readfirstlane(tile % KX);
  const int o_tile = __builtin_amdgcn_readfirstlane(tile / KX);
  const int o0 = o_tile * 128, c0 = c_tile * 128;
  const long long total_chunks = ct.chunk_off[d.n_levels];
  const long long ch_begin = (long long)split * chunks_per_split;
  long long ch_end = ch_begin + chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  const bool do_bias = (bias_slab != nullptr) && ky == HALO && c_tile == 0;

  // ---- per-lane byte offsets inside a chunk: constant for the whole kernel
  const int q4 = tid & 31, rr = tid >> 5;
  unsigned offa[NA], offb[NB];
  {
    const int o = o0 + 4 * q4, c = c0 + 4 * q4;
#pragma unroll
    for (int i = 0; i < NA; ++i) offa[i] = (o < Ns) ? (unsigned)(((rr + RG * i) * Ns + o) * 4) : BAD;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      offb[i] = (c < Cs && j < WK + KX - 1) ? (unsigned)((S * j * Cs + c) * 4) : BAD;
    }
  }

  // ---- wave-uniform chunk position: level, image, row, row segment (the divisions run once)
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
    if (i < d.n_levels && ch_begin >= ct.chunk_off[i]) lvl = i;
  int segs = lvl_pick(ct.segs, lvl), H = lvl_pick(d.h, lvl), W = lvl_pick(d.w, lvl);
  long long row0 = lvl_pick64(d.row_off, lvl);
  int seg, n, y;
  {
    const long long r = (ch_begin < ch_end ? ch_begin : 0) - ct.chunk_off[lvl];
    const long long rowl = r / segs;
    // the 64-bit divisions run on the vector ALU: bring the (wave-uniform) results back to scalar registers, or every
    // address and descriptor derived from them stays in VGPRs and each buffer load gets a waterfall loop
    seg = __builtin_amdgcn_readfirstlane((int)(r - rowl * segs));
    n = __builtin_amdgcn_readfirstlane((int)(rowl / H));
    y = __builtin_amdgcn_readfirstlane((int)(rowl - (long long)(rowl / H) * H));
  }

  float4 ra[NA], rb[NB];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  __amdgpu_buffer_rsrc_t ra_src, rb_src;
  bool left_edge = false;
  // descriptors of the chunk at (lvl, n, y, seg): scalar work only.  live = false: zero records, every load of the
  // "chunk" returns zeros without touching memory (the interleaved variant issues its loads unconditionally, also
  // behind the last chunk, so that the MFMA phase has no control flow in it)
  auto prepare_loads = [&](bool live = true) {
    const int x0 = seg * WK;
    const long long rowbase = row0 + ((long long)n * H + y) * W;
    const int kmax = (W - x0 < WK) ? W - x0 : WK;
    ra_src = uniform_rsrc(dy + (rowbase + x0) * Ns, live ? kmax * Ns * 4 : 0);
    const float* bbase;
    int nrec;
    if (KX == 1) {
      const int Hx = lvl_pick(xd.h, lvl), Wx = lvl_pick(xd.w, lvl);
      const long long xrow = lvl_pick64(xd.row_off, lvl) + ((long long)n * Hx + (long long)S * y) * Wx;
      bbase = x + (xrow + (long long)S * x0) * Cs;
      nrec = ((kmax - 1) * S + 1) * Cs * 4;
    } else {
      const int yy = y + ky - HALO;
      const int jmax = (W - x0 + HALO < WK + KX - 1) ? W - x0 + HALO : WK + KX - 1;
      bbase = x + (rowbase + (long long)(ky - HALO) * W + x0 - HALO) * Cs;  // never dereferenced where it lies outside
      nrec = (yy >= 0 && yy < H) ? jmax * Cs * 4 : 0;
    }
    rb_src = uniform_rsrc(bbase, live ? nrec : 0);
    left_edge = seg == 0;
  };
  auto issue_one = [&](int k) {  // load k of the NA + NB of a chunk
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (k == i) ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra_src, (int)offa[i], 0, 0));
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (k == NA + i) {
        unsigned off = offb[i];
        if (KX > 1 && i == 0) off = (left_edge && rr < HALO) ? BAD : off;  // pixel x0 - HALO + j left of the image
        rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb_src, (int)off, 0, 0));
      }
  };
  auto issue_loads = [&]() {
    prepare_loads();
#pragma unroll
    for (int k = 0; k < NA + NB; ++k) issue_one(k);
  };
  auto advance = [&]() {
    if (++seg == segs) {
      seg = 0;
      if (++y == H) {
        y = 0;
        if (++n == d.n_images) {
          n = 0;
          ++lvl;
          segs = lvl_pick(ct.segs, lvl);
          H = lvl_pick(d.h, lvl);
          W = lvl_pick(d.w, lvl);
          row0 = lvl_pick64(d.row_off, lvl);
        }
      }
    }
  };
  auto store_chunk = [&]() {
    __bf16* Ah = sm;
    __bf16* Al = Ah + WK * WROW;
    __bf16* Bh = Al + WK * WROW;
    __bf16* Bl = Bh + (WK + KX - 1) * WROW;
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int off = wsw(rr + RG * i, 4 * q4);
      split4(ra[i], hi, lo);
      *reinterpret_cast<bf16x4*>(Ah + off) = hi;
      *reinterpret_cast<bf16x4*>(Al + off) = lo;
      if (do_bias) {
        bsum.x += ra[i].x;
        bsum.y += ra[i].y;
        bsum.z += ra[i].z;
        bsum.w += ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      if (j < WK + KX - 1) {
        const int off = wsw(j, 4 * q4);
        split4(rb[i], hi, lo);
        *reinterpret_cast<bf16x4*>(Bh + off) = hi;
        *reinterpret_cast<bf16x4*>(Bl + off) = lo;
      }
    }
  };

  const int wm = wid % WO, wn = wid / WO;
  const int lr = lane & 15, kg = lane >> 4;
  const int row_lane = 8 * kg + (lr >> 2), col4 = 4 * (lane & 3);
  const int a_col = wm * (16 * TOMAX), b_col = wn * 32;
  const bool c_act = c0 + b_col < Cs;
  const int o_left = Nout - (o0 + a_col);

  f32x4v acc[KX][TOMAX][2];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int b = 0; b < TOMAX; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (ch_begin < ch_end) issue_loads();
  const __bf16* Ah = sm;
  const __bf16* Al = Ah + WK * WROW;
  const __bf16* Bh = Al + WK * WROW;
  const __bf16* Bl = Bh + (WK + KX - 1) * WROW;
  for (long long ch = ch_begin; ch < ch_end; ++ch) {
    store_chunk();
    const bool more = ch + 1 < ch_end;
    if (IL) {
      if (more) advance();
      prepare_loads(more);
    } else if (more) {
      advance();
      issue_loads();
    }
    __syncthreads();
    if (IL) {
      // one load after each of the first NA + NB MFMA blocks (12 blocks for the 3x3, 4 for the 1x1: the rest follow the
      // last block); the sched_barrier keeps the compiler from gathering them at either end of the phase
      constexpr int NBLK = (WK / 32) * KX * 2, ILD = IL > 0 ? IL : 1;
      auto hook = [&](int blk) {
#pragma unroll
        for (int k = 0; k < NA + NB; ++k)
          if (k / ILD == blk || (blk == NBLK - 1 && k / ILD >= NBLK)) issue_one(k);  // IL loads per block
        __builtin_amdgcn_sched_barrier(0);
      };
      if (TOMAX == 4 && c_act && o_left > 32) {
        wgrad_mma_v2_hook<TOMAX, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc, hook);
      } else if (c_act && o_left > 16) {
        wgrad_mma_v2_hook<2, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc, hook);
      } else if (c_act && o_left > 0) {
        wgrad_mma_v2_hook<1, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc, hook);
      } else {
#pragma unroll
        for (int k = 0; k < NA + NB; ++k) issue_one(k);
      }
    } else if (TOMAX == 4 && c_act && o_left > 32)
      wgrad_mma_v2<TOMAX, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 16)
      wgrad_mma_v2<2, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 0)
      wgrad_mma_v2<1, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    __syncthreads();
  }

  float* out = slab + (long long)split * Nout * T * Cs;
#pragma unroll
  for (int kx = 0; kx < KX; ++kx)
#pragma unroll
    for (int to = 0; to < TOMAX; ++to)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = c0 + b_col + 16 * tc + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + a_col + 16 * to + 4 * kg + r;
          if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][to][tc][r];
        }
      }

  if (do_bias) {
    float* red = reinterpret_cast<float*>(smem_raw);  // [RG][128]
    *reinterpret_cast<float4*>(red + rr * 128 + 4 * q4) = bsum;
    __syncthreads();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) sum += red[g * 128 + tid];
      if (o0 + tid < Nout) bias_slab[(long long)split * Nout + o0 + tid] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Third generation of the weight-gradient kernel: the 16x16x32 kernel above with the staging phase taken off the
// critical path.  The stage image drops its row padding (256-byte rows, 16-byte chunk ch of row r stored at
// ch ^ (((r & 3) << 2) | ((r >> 2) & 3)): conflict-free for the transposed reads of two lane groups 8 rows apart and for
// the 8-byte staging writes), which shrinks a stage from 83 KB to 66.5 KB -- TWO stages fit.  Each iteration computes on
// stage i & 1 while the next chunk is split and written into the other stage; ONE barrier per chunk instead of two.
// The two waves of a SIMD (w and w + 4) run the two halves in opposite order -- waves 0..3 stage first and then issue
// their MFMAs, waves 4..7 issue their MFMAs first -- so the matrix pipe of a SIMD is fed by one partner while the other
// converts (MI355X_MICROARCH.md: "two waves that run the same program with one barrier per block: try a stagger").
// ------------------------------------------------------------------------------------------------
#define W3ROW 128  // bf16 elements per LDS row: no padding
#define W3STAGE(KX) ((2 * WK + 2 * (WK + (KX) - 1)) * W3ROW)  // bf16 elements per stage
__device__ __forceinline__ int wsw3(int row, int col) {
  const int f = ((row & 3) << 2) | ((row >> 2) & 3);
  return row * W3ROW + ((((col >> 3) ^ f) << 3) | (col & 7));
}

// TO: live 16-row o tiles of this wave (4, 2 or 1)
template <int TO, int KX, int TOMAX>
__device__ __forceinline__ void wgrad_mma_v3(const __bf16* Ah, const __bf16* Al, const __bf16* Bh, const __bf16* Bl,
                                             int row_lane, int col4, int a_col, int b_col,
                                             f32x4v (&acc)[KX][TOMAX][2]) {
#pragma unroll
  for (int s = 0; s < WK / 32; ++s) {
    // dY rows of this lane: r0 = 32 s + 8 kg + q and r0 + 4 (bit 3 of both = kg & 1: one swizzle per lane)
    const int ra0 = 32 * s + row_lane, ra1 = ra0 + 4;
    bf16x8 ah[TO], al[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const int c = a_col + 16 * t + col4;
      const int o0 = wsw3(ra0, c), o1 = wsw3(ra1, c);
      ah[t] = tr_read8_v2(Ah + o0, Ah + o1);
      al[t] = tr_read8_v2(Al + o0, Al + o1);
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
      const int rb0 = ra0 + kx, rb1 = ra1 + kx;  // X row j <-> pixel x0 - HALO + j: tap kx is a row shift
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = b_col + 16 * tc + col4;
        const int o0 = wsw3(rb0, c), o1 = wsw3(rb1, c);
        const bf16x8 bh = tr_read8_v2(Bh + o0, Bh + o1);
        const bf16x8 bl = tr_read8_v2(Bl + o0, Bl + o1);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[to], bh, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bl, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bh, acc[kx][to][tc], 0, 0, 0);
      }
    }
  }
}

// NT = 512: 8 waves = 2 (o) x 4 (c), each 64 o x 32 c; NT = 1024: 16 waves = 4 x 4, each 32 o x 32 c (48 accumulator
// registers, <= 128 registers per lane: four waves per SIMD to hide the staging phases and the transposed-read latency)
template <int KX, int S, int NT>
__global__ __launch_bounds__(NT, NT == 1024 ? 4 : 2) void conv_wgrad_bf16x3_v3_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, scan_pyramid_t xd) {
  constexpr int HALO = KX / 2, T = KX * KX;
  constexpr int RG = NT / 32;                         // pixel-row groups of the staging roles (16 or 32)
  constexpr int NA = WK / RG;                         // dY float4 per thread per chunk
  constexpr int NB = (WK + KX - 1 + RG - 1) / RG;     // X float4 per thread per chunk
  constexpr int WO = NT / 256;                        // waves along o (2 or 4)
  constexpr int TOMAX = 128 / (16 * WO);              // 16-row o tiles per wave (4 or 2)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int qq = L >> 3;
  int tile = qq % n_tiles;
  const int split = (qq / n_tiles) * 8 + xcd;
  const int c_tile = tile % c_tiles;
  tile /= c_tiles;
  const int ky = tile % KX;
  const int o_tile = tile / KX;
  const int o0 = o_tile * 128, c0 = c_tile * 128;
  const long long total_chunks = ct.chunk_off[d.n_levels];
  const long long ch_begin = (long long)split * chunks_per_split;
  long long ch_end = ch_begin + chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  const bool do_bias = (bias_slab != nullptr) && ky == HALO && c_tile == 0;

  const int q4 = tid & 31, rr = tid >> 5;
  float4 ra[NA], rb[NB];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_chunk = [&](long long ch) {
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
      if (i < d.n_levels && ch >= ct.chunk_off[i]) lvl = i;
    const int H = d.h[lvl], W = d.w[lvl];
    const long long r = ch - ct.chunk_off[lvl];
    const int seg = (int)(r % ct.segs[lvl]);
    const long long row = r / ct.segs[lvl];  // n * H + y
    const int y = (int)(row % H);
    const int x0 = seg * WK;
    const long long rowbase = d.row_off[lvl] + row * W;
    const int o = o0 + 4 * q4, c = c0 + 4 * q4;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int k = rr + RG * i;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 + k < W) {
        const long long m = rowbase + x0 + k;
        if ((Ns & 3) == 0 && o + 3 < Ns) {
          ra[i] = *reinterpret_cast<const float4*>(dy + m * Ns + o);
        } else {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
          for (int e = 0; e < 4; ++e)
            if (o + e < Ns) t[e] = dy[m * Ns + o + e];
          ra[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
    const int yy = y + ky - HALO;
    const bool yok = yy >= 0 && yy < H && c < Cs;
    const int Wx = (KX == 1) ? xd.w[lvl] : W;
    const long long xrow = (KX == 1) ? xd.row_off[lvl] + ((row / H) * xd.h[lvl] + (long long)S * y) * Wx
                                     : rowbase + (long long)(ky - 1) * W;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      const int xx = x0 - HALO + j;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < WK + KX - 1 && yok && xx >= 0 && xx < W)
        rb[i] = *reinterpret_cast<const float4*>(x + (xrow + (long long)S * xx) * Cs + c);
    }
  };
  auto store_chunk = [&](int stage) {
    __bf16* Ah = sm + stage * W3STAGE(KX);
    __bf16* Al = Ah + WK * W3ROW;
    __bf16* Bh = Al + WK * W3ROW;
    __bf16* Bl = Bh + (WK + KX - 1) * W3ROW;
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int off = wsw3(rr + RG * i, 4 * q4);
      split4(ra[i], hi, lo);
      *reinterpret_cast<bf16x4*>(Ah + off) = hi;
      *reinterpret_cast<bf16x4*>(Al + off) = lo;
      if (do_bias) {
        bsum.x += ra[i].x;
        bsum.y += ra[i].y;
        bsum.z += ra[i].z;
        bsum.w += ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      if (j < WK + KX - 1) {
        const int off = wsw3(j, 4 * q4);
        split4(rb[i], hi, lo);
        *reinterpret_cast<bf16x4*>(Bh + off) = hi;
        *reinterpret_cast<bf16x4*>(Bl + off) = lo;
      }
    }
  };

  // waves: WO (o) x 4 (c); each wave (128 / WO) o x 32 c = TOMAX x 2 MFMA tiles of 16 x 16, for each of the KX taps
  const int wm = wid % WO, wn = wid / WO;
  const int lr = lane & 15, kg = lane >> 4;
  const int row_lane = 8 * kg + (lr >> 2), col4 = 4 * (lane & 3);
  const int a_col = wm * (16 * TOMAX), b_col = wn * 32;
  const bool c_act = c0 + b_col < Cs;
  const int o_left = Nout - (o0 + a_col);  // live o rows of this wave

  f32x4v acc[KX][TOMAX][2];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int b = 0; b < TOMAX; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  auto mma = [&](int stage) {
    const __bf16* Ah = sm + stage * W3STAGE(KX);
    const __bf16* Al = Ah + WK * W3ROW;
    const __bf16* Bh = Al + WK * W3ROW;
    const __bf16* Bl = Bh + (WK + KX - 1) * W3ROW;
    // wave-uniform skips of dead tiles (third c tile of Cin = 264 / 265; Cout = 8 / 5 / 1 heads)
    if (TOMAX == 4 && c_act && o_left > 32)
      wgrad_mma_v3<TOMAX, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 16)
      wgrad_mma_v3<2, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 0)
      wgrad_mma_v3<1, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
  };
  auto stage_next = [&](long long ch, int stage) {  // convert chunk ch + 1 (in registers) into `stage`, fetch chunk ch + 2
    if (ch + 1 < ch_end) {
      store_chunk(stage);
      if (ch + 2 < ch_end) load_chunk(ch + 2);
    }
  };
  const bool late = wid >= 4;  // the SIMD partner of wave w is wave w + 4
  if (ch_begin < ch_end) {
    load_chunk(ch_begin);
    store_chunk(0);
    if (ch_begin + 1 < ch_end) load_chunk(ch_begin + 1);
  }
  __syncthreads();
  for (long long ch = ch_begin; ch < ch_end; ++ch) {
    const int cur = (int)(ch - ch_begin) & 1;
    if (!late) {
      stage_next(ch, cur ^ 1);
      mma(cur);
    } else {
      mma(cur);
      stage_next(ch, cur ^ 1);
    }
    __syncthreads();  // stage cur is free for chunk ch + 2, stage cur ^ 1 is complete
  }

  // C/D map of 16x16: column = lane & 15 = input channel c, row = 4 * (lane >> 4) + reg = output channel o
  float* out = slab + (long long)split * Nout * T * Cs;
#pragma unroll
  for (int kx = 0; kx < KX; ++kx)
#pragma unroll
    for (int to = 0; to < TOMAX; ++to)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = c0 + b_col + 16 * tc + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + a_col + 16 * to + 4 * kg + r;
          if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][to][tc][r];
        }
      }

  if (do_bias) {
    float* red = reinterpret_cast<float*>(smem_raw);  // [RG][128]
    *reinterpret_cast<float4*>(red + rr * 128 + 4 * q4) = bsum;
    __syncthreads();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) sum += red[g * 128 + tid];
      if (o0 + tid < Nout) bias_slab[(long long)split * Nout + o0 + tid] = sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fifth generation: the fourth generation's staging (scalar chunk walk, range-checked buffer loads) in the third
// generation's structure -- unpadded swizzled LDS image, TWO stages, one barrier per chunk, the two waves of a SIMD
// running "split + write chunk k+1, fetch chunk k+2" and "MFMAs of chunk k" in opposite order.  With the staging down
// to a few dozen instructions the overlap the third generation was built for becomes visible (it measured slower in
// round 2 because both halves of a SIMD pair spent most of a chunk in address arithmetic).  Same K order: bit-identical
// slabs.
// ------------------------------------------------------------------------------------------------
template <int KX, int S, int NT>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_bf16x3_v5_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, scan_pyramid_t xd) {
  constexpr int HALO = KX / 2, T = KX * KX;
  constexpr int RG = NT / 32;                         // pixel-row groups of the staging roles
  constexpr int NA = WK / RG;                         // dY float4 per thread per chunk
  constexpr int NB = (WK + KX - 1 + RG - 1) / RG;     // X float4 per thread per chunk
  constexpr int WO = NT / 256;                        // waves along o
  constexpr int TOMAX = 128 / (16 * WO);              // 16-row o tiles per wave
  constexpr unsigned BAD = 0x80000000u;               // a byte offset beyond every descriptor: the load returns zeros
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int qq = L >> 3;
  // integer division runs on the vector ALU: pin the (wave-uniform) quotients back into scalar registers so that
  // everything derived from them -- the chunk walk, the buffer descriptors -- stays scalar
  int tile = __builtin_amdgcn_readfirstlane(qq % n_tiles);
  const int split = __builtin_amdgcn_readfirstlane(qq / n_tiles) * 8 + xcd;
  const int c_tile = __builtin_amdgcn_readfirstlane(tile % c_tiles);
  tile = __builtin_amdgcn_readfirstlane(tile / c_tiles);
  const int ky = __builtin_amdgcn_readfirstlane(tile % KX);
  const int o_tile = __builtin_amdgcn_readfirstlane(tile / KX);
  const int o0 = o_tile * 128, c0 = c_tile * 128;
  const long long total_chunks = ct.chunk_off[d.n_levels];
  const long long ch_begin = (long long)split * chunks_per_split;
  long long ch_end = ch_begin + chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  const bool do_bias = (bias_slab != nullptr) && ky == HALO && c_tile == 0;

  // ---- per-lane byte offsets inside a chunk: constant for the whole kernel
  const int q4 = tid & 31, rr = tid >> 5;
  unsigned offa[NA], offb[NB];
  {
    const int o = o0 + 4 * q4, c = c0 + 4 * q4;
#pragma unroll
    for (int i = 0; i < NA; ++i) offa[i] = (o < Ns) ? (unsigned)(((rr + RG * i) * Ns + o) * 4) : BAD;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      offb[i] = (c < Cs && j < WK + KX - 1) ? (unsigned)((S * j * Cs + c) * 4) : BAD;
    }
  }

  // ---- wave-uniform chunk position: level, image, row, row segment (the divisions run once)
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
    if (i < d.n_levels && ch_begin >= ct.chunk_off[i]) lvl = i;
  int segs = lvl_pick(ct.segs, lvl), H = lvl_pick(d.h, lvl), W = lvl_pick(d.w, lvl);
  long long row0 = lvl_pick64(d.row_off, lvl);
  int seg, n, y;
  {
    const long long r = (ch_begin < ch_end ? ch_begin : 0) - ct.chunk_off[lvl];
    const long long rowl = r / segs;
    // the 64-bit divisions run on the vector ALU: bring the (wave-uniform) results back to scalar registers, or every
    // address and descriptor derived from them stays in VGPRs and each buffer load gets a waterfall loop
    seg = __builtin_amdgcn_readfirstlane((int)(r - rowl * segs));
    n = __builtin_amdgcn_readfirstlane((int)(rowl / H));
    y = __builtin_amdgcn_readfirstlane((int)(rowl - (long long)(rowl / H) * H));
  }

  float4 ra[NA], rb[NB];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue_loads = [&]() {  // the chunk at (lvl, n, y, seg)
    const int x0 = seg * WK;
    const long long rowbase = row0 + ((long long)n * H + y) * W;
    const int kmax = (W - x0 < WK) ? W - x0 : WK;
    {
      const float* base = dy + (rowbase + x0) * Ns;
      const __amdgpu_buffer_rsrc_t ra_src = uniform_rsrc(base, kmax * Ns * 4);
#pragma unroll
      for (int i = 0; i < NA; ++i)
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra_src, (int)offa[i], 0, 0));
    }
    const float* bbase;
    int nrec;
    if (KX == 1) {
      const int Hx = lvl_pick(xd.h, lvl), Wx = lvl_pick(xd.w, lvl);
      const long long xrow = lvl_pick64(xd.row_off, lvl) + ((long long)n * Hx + (long long)S * y) * Wx;
      bbase = x + (xrow + (long long)S * x0) * Cs;
      nrec = ((kmax - 1) * S + 1) * Cs * 4;
    } else {
      const int yy = y + ky - HALO;
      const int jmax = (W - x0 + HALO < WK + KX - 1) ? W - x0 + HALO : WK + KX - 1;
      bbase = x + (rowbase + (long long)(ky - HALO) * W + x0 - HALO) * Cs;  // never dereferenced where it lies outside
      nrec = (yy >= 0 && yy < H) ? jmax * Cs * 4 : 0;
    }
    const __amdgpu_buffer_rsrc_t rb_src = uniform_rsrc(bbase, nrec);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      unsigned off = offb[i];
      if (KX > 1 && i == 0) off = (seg == 0 && rr < HALO) ? BAD : off;  // pixel x0 - HALO + j left of the image
      rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb_src, (int)off, 0, 0));
    }
  };
  auto advance = [&]() {
    if (++seg == segs) {
      seg = 0;
      if (++y == H) {
        y = 0;
        if (++n == d.n_images) {
          n = 0;
          ++lvl;
          segs = lvl_pick(ct.segs, lvl);
          H = lvl_pick(d.h, lvl);
          W = lvl_pick(d.w, lvl);
          row0 = lvl_pick64(d.row_off, lvl);
        }
      }
    }
  };
  auto store_chunk = [&](int stage) {
    __bf16* Ah = sm + stage * W3STAGE(KX);
    __bf16* Al = Ah + WK * W3ROW;
    __bf16* Bh = Al + WK * W3ROW;
    __bf16* Bl = Bh + (WK + KX - 1) * W3ROW;
    bf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int off = wsw3(rr + RG * i, 4 * q4);
      split4(ra[i], hi, lo);
      *reinterpret_cast<bf16x4*>(Ah + off) = hi;
      *reinterpret_cast<bf16x4*>(Al + off) = lo;
      if (do_bias) {
        bsum.x += ra[i].x;
        bsum.y += ra[i].y;
        bsum.z += ra[i].z;
        bsum.w += ra[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int j = rr + RG * i;
      if (j < WK + KX - 1) {
        const int off = wsw3(j, 4 * q4);
        split4(rb[i], hi, lo);
        *reinterpret_cast<bf16x4*>(Bh + off) = hi;
        *reinterpret_cast<bf16x4*>(Bl + off) = lo;
      }
    }
  };

  const int wm = wid % WO, wn = wid / WO;
  const int lr = lane & 15, kg = lane >> 4;
  const int row_lane = 8 * kg + (lr >> 2), col4 = 4 * (lane & 3);
  const int a_col = wm * (16 * TOMAX), b_col = wn * 32;
  const bool c_act = c0 + b_col < Cs;
  const int o_left = Nout - (o0 + a_col);

  f32x4v acc[KX][TOMAX][2];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int b = 0; b < TOMAX; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  auto mma = [&](int stage) {
    const __bf16* Ah = sm + stage * W3STAGE(KX);
    const __bf16* Al = Ah + WK * W3ROW;
    const __bf16* Bh = Al + WK * W3ROW;
    const __bf16* Bl = Bh + (WK + KX - 1) * W3ROW;
    if (TOMAX == 4 && c_act && o_left > 32)
      wgrad_mma_v3<TOMAX, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 16)
      wgrad_mma_v3<2, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
    else if (c_act && o_left > 0)
      wgrad_mma_v3<1, KX, TOMAX>(Ah, Al, Bh, Bl, row_lane, col4, a_col, b_col, acc);
  };
  // convert chunk ch + 1 (in registers since the previous iteration) into `stage`, then fetch chunk ch + 2
  auto stage_next = [&](long long ch, int stage) {
    if (ch + 1 < ch_end) {
      store_chunk(stage);
      if (ch + 2 < ch_end) {
        advance();
        issue_loads();
      }
    }
  };
  const bool late = wid >= NT / 128;  // the SIMD partner of wave w is wave w + 4: the two run the halves in opposite order
  if (ch_begin < ch_end) {
    issue_loads();
    store_chunk(0);
    if (ch_begin + 1 < ch_end) {
      advance();
      issue_loads();
    }
  }
  __syncthreads();
  for (long long ch = ch_begin; ch < ch_end; ++ch) {
    const int cur = (int)(ch - ch_begin) & 1;
    if (!late) {
      stage_next(ch, cur ^ 1);
      mma(cur);
    } else {
      mma(cur);
      stage_next(ch, cur ^ 1);
    }
    __syncthreads();  // stage cur is free for chunk ch + 2, stage cur ^ 1 is complete
  }

  float* out = slab + (long long)split * Nout * T * Cs;
#pragma unroll
  for (int kx = 0; kx < KX; ++kx)
#pragma unroll
    for (int to = 0; to < TOMAX; ++to)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = c0 + b_col + 16 * tc + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + a_col + 16 * to + 4 * kg + r;
          if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][to][tc][r];
        }
      }

  if (do_bias) {
    float* red = reinterpret_cast<float*>(smem_raw);  // [RG][128]
    *reinterpret_cast<float4*>(red + rr * 128 + 4 * q4) = bsum;
    __syncthreads();
    if (tid < 128) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) sum += red[g * 128 + tid];
      if (o0 + tid < Nout) bias_slab[(long long)split * Nout + o0 + tid] = sum;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Sixth generation: producer / consumer wave specialisation.  What is left of a K chunk in the fourth generation beside
// its MFMAs (3.8 us per chunk against 2.9 us for the MFMA phase with its barriers alone, profiles/r03_wgrad_exp.txt) is
// the fp32 -> bf16 hi / lo split and the LDS writes, which all eight waves execute together between two barriers while
// the matrix pipe idles.  Here a workgroup has 12 waves: waves 0..7 ONLY read fragments and issue MFMAs (the 2 x 4 wave
// grid of the earlier generations, two per SIMD), waves 8..11 -- one per SIMD -- ONLY stage: they walk the chunks (scalar
// state, range-checked buffer loads, as in the fourth generation), split and write the NEXT chunk into the other of two
// LDS stages (288-byte rows, see wsw6: 2 x 73 KB) while the consumers multiply the current
// one, and fetch the chunk after that.  One barrier per chunk; the vector ALU work of the split runs beside the matrix
// pipe on every SIMD instead of in front of it.  168 registers per lane (three waves per SIMD): the accumulators live
// in the consumer branch only, the staging registers in the producer branch only.  Same K order: bit-identical slabs.
// ------------------------------------------------------------------------------------------------
// LDS image of the sixth generation: two stages must fit, and the transposed-read addresses must stay "base + immediate"
// (the unpadded image of the third generation XORs four row bits into the column: one address register per (tile, tap,
// row group), which does not fit beside 96 accumulators at three waves per SIMD).  Rows are 288 bytes (256 + 32 pad):
// consecutive rows start 8 banks apart, so the four pixel rows a 16-lane group of ds_read_b64_tr_b16 touches (32 bytes
// each) cover 32 banks; the other lane group of the same half wave reads rows 8 further (64 banks = 0 further) and is
// moved to the other 32 banks by swapping the two 128-byte halves of a row when bit 3 of the row index is set.
#define W6ROW 144  // bf16 elements per LDS row
#define W6STAGE(KX) ((2 * WK + 2 * (WK + (KX) - 1)) * W6ROW)  // bf16 elements per stage: 74,880 B for the 3x3
__device__ __forceinline__ int wsw6(int row, int col) { return row * W6ROW + (col ^ (((row >> 3) & 1) << 6)); }

// Fragment addresses as "lane base + compile-time offset": the half-row swap of wsw6 is applied to the lane's base column
// only (adding 16 t or 16 tc afterwards never carries into bit 6: the base columns are 64 wm + col4 and 32 wn + col4,
// col4 < 16), and a row offset is an immediate wherever it cannot change bit 3 of the row -- true for the rows
// 8 kg + q (+ kx) and 8 kg + q + 4, q = (lane & 15) >> 2; only rows 8 kg + q + 4 + kx, kx = 1, 2 may cross into the next
// group of eight and get bases of their own.  Four address registers instead of one per (tile, tap, row group).
struct W6Lane {
  int a;        // (row 8 kg + q, column a_col + col4): A operand, rows + 4 and tiles + 16 t by immediate
  int b;        // (row 8 kg + q, column b_col + col4): B operand, rows + kx, + 4 (kx = 0) and tiles + 16 tc by immediate
  int b1[2];    // (row 8 kg + q + 4 + kx, same column), kx = 1, 2
};
__device__ __forceinline__ W6Lane w6_lane(int row_lane, int col4, int a_col, int b_col) {
  W6Lane w;
  w.a = wsw6(row_lane, a_col + col4);
  w.b = wsw6(row_lane, b_col + col4);
  w.b1[0] = wsw6(row_lane + 5, b_col + col4);
  w.b1[1] = wsw6(row_lane + 6, b_col + col4);
  return w;
}

template <int TO, int KX, int TOMAX>
__device__ __forceinline__ void wgrad_mma_v6(const __bf16* Ah, const __bf16* Al, const __bf16* Bh, const __bf16* Bl,
                                             const W6Lane& w, f32x4v (&acc)[KX][TOMAX][2]) {
#pragma unroll
  for (int s = 0; s < WK / 32; ++s) {
    bf16x8 ah[TO], al[TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const int o0 = w.a + 32 * s * W6ROW + 16 * t, o1 = o0 + 4 * W6ROW;
      ah[t] = tr_read8_v2(Ah + o0, Ah + o1);
      al[t] = tr_read8_v2(Al + o0, Al + o1);
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        // X row j <-> pixel x0 - HALO + j: tap kx is a row shift
        const int o0 = w.b + (32 * s + kx) * W6ROW + 16 * tc;
        const int o1 = (kx == 0 ? w.b + 4 * W6ROW : w.b1[kx - 1]) + 32 * s * W6ROW + 16 * tc;
        const bf16x8 bh = tr_read8_v2(Bh + o0, Bh + o1);
        const bf16x8 bl = tr_read8_v2(Bl + o0, Bl + o1);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[to], bh, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bl, acc[kx][to][tc], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < TO; ++to) acc[kx][to][tc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[to], bh, acc[kx][to][tc], 0, 0, 0);
      }
    }
  }
}

template <int KX>
__global__ __launch_bounds__(768, 3) void conv_wgrad_bf16x3_v6_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, int dbg, int prio) {
  // dbg (scan_tune "wgrad_exp" 5 / 6 / 7, timing only, wrong results): bit 0 = the consumers skip their MFMA phase, bit 1 =
  // the producers stage the first two chunks only
  constexpr int HALO = KX / 2, T = KX * KX;
  constexpr int TOMAX = 4;                            // 16-row o tiles per consumer wave (64 o x 32 c per wave)
  constexpr int PRG = 8;                              // pixel-row groups of the 256 producer threads
  constexpr int NA = WK / PRG;                        // dY float4 per producer thread per chunk
  constexpr int NB = (WK + KX - 1 + PRG - 1) / PRG;   // X float4 per producer thread per chunk
  constexpr unsigned BAD = 0x80000000u;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* sm = reinterpret_cast<__bf16*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int L = blockIdx.x;
  const int xcd = L & 7;
  const int qq = L >> 3;
  int tile = __builtin_amdgcn_readfirstlane(qq % n_tiles);
  const int split = __builtin_amdgcn_readfirstlane(qq / n_tiles) * 8 + xcd;
  const int c_tile = __builtin_amdgcn_readfirstlane(tile % c_tiles);
  tile = __builtin_amdgcn_readfirstlane(tile / c_tiles);
  const int ky = __builtin_amdgcn_readfirstlane(tile % KX);
  const int o_tile = __builtin_amdgcn_readfirstlane(tile / KX);
  const int o0 = o_tile * 128, c0 = c_tile * 128;
  const long long total_chunks = ct.chunk_off[d.n_levels];
  const long long ch_begin = (long long)split * chunks_per_split;
  long long ch_end = ch_begin + chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;
  const int nch = ch_end > ch_begin ? (int)(ch_end - ch_begin) : 0;
  const bool do_bias = (bias_slab != nullptr) && ky == HALO && c_tile == 0;

  if (wid >= 8) {
    // =============================================================== producers: one wave per SIMD
    // Per-lane offsets: one register per operand, the pixel-row group i of a load is its scalar offset.
    // static priority for the staging wave of a SIMD (scan_tune "wgrad_prio"): its vector instructions are few beside
    // its two partners' MFMA streams, but arbitrated by age it loses the issue slot to them and reaches the barrier last
    if (prio > 0) __builtin_amdgcn_s_setprio(3);
    const int ptid = tid - 512;
    const int q4 = ptid & 31, rr = ptid >> 5;
    const int o = o0 + 4 * q4, c = c0 + 4 * q4;
    const unsigned offa = (o < Ns) ? (unsigned)((rr * Ns + o) * 4) : BAD;
    const unsigned offb = (c < Cs) ? (unsigned)((rr * Cs + c) * 4) : BAD;
    const unsigned offb_last = (rr + PRG * (NB - 1) < WK + KX - 1) ? offb : BAD;  // rows of the last group beyond the halo
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
      if (i < d.n_levels && ch_begin >= ct.chunk_off[i]) lvl = i;
    int segs = lvl_pick(ct.segs, lvl), H = lvl_pick(d.h, lvl), W = lvl_pick(d.w, lvl);
    long long row0 = lvl_pick64(d.row_off, lvl);
    int seg, n, y;
    {
      const long long r = (nch > 0 ? ch_begin : 0) - ct.chunk_off[lvl];
      const long long rowl = r / segs;
      seg = __builtin_amdgcn_readfirstlane((int)(r - rowl * segs));
      n = __builtin_amdgcn_readfirstlane((int)(rowl / H));
      y = __builtin_amdgcn_readfirstlane((int)(rowl - (long long)(rowl / H) * H));
    }
    float4 ra[NA], rb[NB];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    __amdgpu_buffer_rsrc_t ra_src, rb_src;
    bool left_edge = false;
    auto prepare = [&](bool live) {  // descriptors of the chunk at (lvl, n, y, seg); !live: zero records
      const int x0 = seg * WK;
      const long long rowbase = row0 + ((long long)n * H + y) * W;
      const int kmax = (W - x0 < WK) ? W - x0 : WK;
      ra_src = uniform_rsrc(dy + (rowbase + x0) * Ns, live ? kmax * Ns * 4 : 0);
      const int yy = y + ky - HALO;
      const int jmax = (W - x0 + HALO < WK + KX - 1) ? W - x0 + HALO : WK + KX - 1;
      const float* bbase = x + (rowbase + (long long)(ky - HALO) * W + x0 - HALO) * Cs;  // never dereferenced outside
      rb_src = uniform_rsrc(bbase, (live && yy >= 0 && yy < H) ? jmax * Cs * 4 : 0);
      left_edge = seg == 0;
    };
    auto load_a = [&]() {
#pragma unroll
      for (int i = 0; i < NA; ++i)
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra_src, (int)offa, PRG * i * Ns * 4, 0));
    };
    auto load_b = [&]() {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        unsigned off = i == NB - 1 ? offb_last : offb;
        if (KX > 1 && i == 0) off = (left_edge && rr < HALO) ? BAD : off;  // pixel x0 - HALO + j left of the image
        rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb_src, (int)off, PRG * i * Cs * 4, 0));
      }
    };
    auto advance = [&]() {
      if (++seg == segs) {
        seg = 0;
        if (++y == H) {
          y = 0;
          if (++n == d.n_images) {
            n = 0;
            ++lvl;
            segs = lvl_pick(ct.segs, lvl);
            H = lvl_pick(d.h, lvl);
            W = lvl_pick(d.w, lvl);
            row0 = lvl_pick64(d.row_off, lvl);
          }
        }
      }
    };
    auto store_a = [&](int stage) {
      __bf16* Ah = sm + stage * W6STAGE(KX);
      __bf16* Al = Ah + WK * W6ROW;
      bf16x4 hi, lo;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int off = wsw6(rr + PRG * i, 4 * q4);
        split4(ra[i], hi, lo);
        *reinterpret_cast<bf16x4*>(Ah + off) = hi;
        *reinterpret_cast<bf16x4*>(Al + off) = lo;
        if (do_bias) {
          bsum.x += ra[i].x;
          bsum.y += ra[i].y;
          bsum.z += ra[i].z;
          bsum.w += ra[i].w;
        }
      }
    };
    auto store_b = [&](int stage) {
      __bf16* Bh = sm + stage * W6STAGE(KX) + 2 * WK * W6ROW;
      __bf16* Bl = Bh + (WK + KX - 1) * W6ROW;
      bf16x4 hi, lo;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int j = rr + PRG * i;
        if (j < WK + KX - 1) {
          const int off = wsw6(j, 4 * q4);
          split4(rb[i], hi, lo);
          *reinterpret_cast<bf16x4*>(Bh + off) = hi;
          *reinterpret_cast<bf16x4*>(Bl + off) = lo;
        }
      }
    };
    // One register set, refilled as soon as a half of it has been converted: the loads of chunk j + 1 are issued right
    // behind the LDS writes of the same operand of chunk j, i.e. EARLY in an iteration, and have until the same point of
    // the next iteration to land (issued at the end of the iteration -- behind both operands' conversion -- the producers
    // waited a full memory latency in front of every barrier: profiles/r03_wgrad_v6_exp.txt).  Behind the last chunk
    // the loads go through zero-record descriptors: no control flow inside the iteration.
    prepare(nch > 0);
    load_a();
    load_b();
    store_a(0);
    if (nch > 1) advance();
    prepare(nch > 1);
    load_a();
    store_b(0);
    load_b();
    __syncthreads();  // stage 0 is complete
    for (int k = 0; k < nch; ++k) {
      if (!(dbg & 2)) {
        const int stage = (k + 1) & 1;  // chunk k + 1 is in the registers; chunk k + 2 follows it
        const bool more = k + 2 < nch;
        if (more) advance();
        prepare(more);
        store_a(stage);
        load_a();
        store_b(stage);
        load_b();
      }
      __syncthreads();  // the consumers are done with stage k & 1; stage (k + 1) & 1 is complete
    }
    if (do_bias) {  // column sums of this split's dY rows: reduce the 8 pixel-row groups through LDS
      float* red = reinterpret_cast<float*>(smem_raw);  // [PRG][128]; every stage read is behind the last barrier
      *reinterpret_cast<float4*>(red + rr * 128 + 4 * q4) = bsum;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (do_bias && ptid < 128) {
      const float* red = reinterpret_cast<const float*>(smem_raw);
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < PRG; ++g) sum += red[g * 128 + ptid];
      if (o0 + ptid < Nout) bias_slab[(long long)split * Nout + o0 + ptid] = sum;
    }
    return;
  }

  // ================================================================= consumers: 2 (o) x 4 (c) waves, 64 o x 32 c each
  const int wm = wid % 2, wn = wid / 2;
  const int lr = lane & 15, kg = lane >> 4;
  const int row_lane = 8 * kg + (lr >> 2), col4 = 4 * (lane & 3);
  const int a_col = wm * (16 * TOMAX), b_col = wn * 32;
  const bool c_act = c0 + b_col < Cs;
  const int o_left = Nout - (o0 + a_col);

  f32x4v acc[KX][TOMAX][2];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int b = 0; b < TOMAX; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4v{0.f, 0.f, 0.f, 0.f};

  __syncthreads();  // stage 0 is complete
  // one K loop per live-tile count (wave-uniform; dead tiles: third c tile of Cin = 264 / 268, Cout = 8 / 5 / 1 heads):
  // inside one loop the compiler would keep the fragment addresses of all three variants in registers across it, which
  // at 168 registers per lane spills
  const W6Lane wl = w6_lane(row_lane, col4, a_col, b_col);
  auto run = [&](auto to_tag) {
    constexpr int TO = decltype(to_tag)::value;
    for (int k = 0; k < nch; ++k) {
      const int stage = k & 1;
      const __bf16* Ah = sm + stage * W6STAGE(KX);
      const __bf16* Al = Ah + WK * W6ROW;
      const __bf16* Bh = Al + WK * W6ROW;
      const __bf16* Bl = Bh + (WK + KX - 1) * W6ROW;
      if constexpr (TO > 0)
        if (!(dbg & 1)) wgrad_mma_v6<TO, KX, TOMAX>(Ah, Al, Bh, Bl, wl, acc);
      __syncthreads();  // done with this stage; the other one is complete
    }
  };
  if (c_act && o_left > 32)
    run(std::integral_constant<int, 4>{});
  else if (c_act && o_left > 16)
    run(std::integral_constant<int, 2>{});
  else if (c_act && o_left > 0)
    run(std::integral_constant<int, 1>{});
  else
    run(std::integral_constant<int, 0>{});

  float* out = slab + (long long)split * Nout * T * Cs;
#pragma unroll
  for (int kx = 0; kx < KX; ++kx)
#pragma unroll
    for (int to = 0; to < TOMAX; ++to)
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = c0 + b_col + 16 * tc + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + a_col + 16 * to + 4 * kg + r;
          if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][to][tc][r];
        }
      }
  __syncthreads();  // pairs with the producers' bias-reduction barrier
}

// scan_tune "wgrad_v2": 1 = always the 16x16x32 weight-gradient kernel, 0 = always the 32x32x16 one, 2 (default) = by
// shape.  Same-process A/B per layer (profiles/r02_wgrad_ab.txt): +3...11 % where the input channels fill whole
// 128-wide tiles (256 -> 256 towers 758 -> 683 us, conv3_x 1785 -> 1706, conv4_x 1716 -> 1667), -1...3 % on the
// 264 / 268-channel inputs (dis P3, head_out: a third, almost empty channel tile) -- so those stay on the first kernel.
int g_scan_wgrad_v2 = 2;
// scan_tune "wgrad_wg1024": 16 waves per workgroup instead of 8 -- 2 (default) = for the launches that stay on the
// 32x32x16 kernel (input channels not a multiple of 128: dis P3 264 -> 1024 2503 -> 2420 us, head_out 268 -> 256
// 912 -> 864; the padded image needs no per-read address arithmetic, 128 registers, no spills); 1 = for the 16x16x32
// kernel, measured 33 % SLOWER on every layer (conv3_x 1784 -> 2654 us: at 128 registers its swizzled transposed-read
// addresses spill inside the MFMA loop); 0 = 8 waves everywhere.  profiles/r02_wgrad_ab.txt.
int g_scan_wgrad_wg1024 = 2;
// scan_tune "wgrad_v3": 1 = the double-buffered, staggered 16x16x32 weight-gradient kernel for the 3x3 convs
int g_scan_wgrad_v3 = 0;
// scan_tune "wgrad_exp": timing ablations of the 16x16x32 weight-gradient kernel (wrong results; see the kernel)
int g_scan_wgrad_exp = 0;
// scan_tune "wgrad_v4": 2 (default) = every bf16x3 weight-gradient launch takes the fourth generation (scalar chunk walk +
// buffer loads); 1 = only the launches that took the 16x16x32 kernel (bit-identical results; the 264 / 268-channel inputs
// and the 64-channel layers stay on the 32x32x16 kernel); 0 = off.  Same-process A/B: profiles/r03_wgrad_v4_ab.txt
int g_scan_wgrad_v4 = 2;
// scan_tune "wgrad_v5": 1 = the 3x3 launches of the fourth generation take the double-buffered, staggered fifth
int g_scan_wgrad_v5 = 0;
// scan_tune "wgrad_v6": 1 = the 3x3 launches of the fourth generation take the producer / consumer sixth
int g_scan_wgrad_v6 = 1;
// scan_tune "wgrad_prio": 1 = the producer waves of the sixth generation run at s_setprio 3
int g_scan_wgrad_prio = 1;
// scan_tune "wgrad_il": n > 0 = the fourth-generation 3x3 kernel issues the next chunk's loads n at a time between the MFMA
// blocks of the current chunk instead of together before the barrier (0)
int g_scan_wgrad_il = 1;
static inline bool wgrad_use_v2(int Cs) { return g_scan_wgrad_v2 == 1 || (g_scan_wgrad_v2 == 2 && Cs % 128 == 0); }
// 2 = the 16x16x32 weight-gradient kernel, 1 = the 32x32x16 one, for an input channel stride Cs (bench.py labels)
extern "C" int scan_conv_wgrad_bf16x3_generation(int32_t Cs) { return wgrad_use_v2(Cs) ? 2 : 1; }

// weight-slab reduction (float4 columns, splits summed in order) + bias-slab reduction in the extra last block
__global__ __launch_bounds__(256) void slab_bias_reduce_kernel(const float* __restrict__ slab, int splits, int64_t n,
                                                               float* __restrict__ dw, const float* __restrict__ bs,
                                                               int nb, float* __restrict__ db, int accumulate) {
  const int wblocks = gridDim.x - (db ? 1 : 0);
  if ((int)blockIdx.x < wblocks) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)wblocks * blockDim.x) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(slab + (int64_t)k * n)[i];
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
      }
      float4* d = reinterpret_cast<float4*>(dw) + i;
      if (accumulate) {
        const float4 o = *d;
        s.x += o.x;
        s.y += o.y;
        s.z += o.z;
        s.w += o.w;
      }
      *d = s;
    }
  } else {
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
      float s = 0.f;
      for (int k = 0; k < splits; ++k) s += bs[(int64_t)k * nb + i];
      db[i] = accumulate ? db[i] + s : s;
    }
  }
}

__global__ void bias_slab_reduce_kernel(const float* __restrict__ bs, int splits, int n, float* __restrict__ db,
                                        int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += bs[(long long)k * n + i];
  db[i] = accumulate ? db[i] + s : s;
}

extern "C" void scan_slab_reduce_launch(const float* slab, int splits, int64_t n, float* dw, int accumulate,
                                        hipStream_t st);

// scan_tune "wgrad_wgs": workgroups a weight-gradient launch aims at (tiles x splits), see the sweep quoted in wgrad3_plan
int g_scan_wgrad_wgs = 768;
static void wgrad3_plan(const scan_pyramid_t* d, int Cs, int Cout, ChunkTab* ct, int* n_tiles, int* c_tiles,
                        int* splits, int* cps, int KX = 3) {
  ct->chunk_off[0] = 0;
  for (int l = 0; l < SCAN_MAX_LEVELS; ++l) {
    if (l < d->n_levels) {
      ct->segs[l] = (d->w[l] + WK - 1) / WK;
      ct->chunk_off[l + 1] = ct->chunk_off[l] + (long long)d->n_images * d->h[l] * ct->segs[l];
    } else {
      ct->segs[l] = 1;
      ct->chunk_off[l + 1] = ct->chunk_off[l];
    }
  }
  const long long chunks = ct->chunk_off[d->n_levels];
  *c_tiles = (Cs + 127) / 128;
  *n_tiles = ((Cout + 127) / 128) * KX * *c_tiles;
  // ~3 workgroups per CU in total.  Swept on the device (tower layer, us): 256 -> 499, 512 -> 428, 768 -> 355,
  // 1024 -> 414, 1536 -> 411, 2304 -> 486: fewer splits lengthen each workgroup's serial chunk chain, more splits
  // cost slab traffic and leave partial rounds
  long long s = g_scan_wgrad_wgs / *n_tiles;
  if (s < 1) s = 1;
  const long long smax = (chunks + 7) / 8;
  if (s > smax) s = smax;
  s = (s + 7) / 8 * 8;  // groups of 8 splits, one per XCD
  *cps = (int)((chunks + s - 1) / s);
  if (*cps < 1) *cps = 1;
  *splits = (int)s;
}

extern "C" int64_t scan_conv3x3_wgrad_bf16x3_ws_floats(const scan_pyramid_t* d, int32_t Cs, int32_t Cout) {
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad3_plan(d, Cs, Cout, &ct, &nt, &ctl, &sp, &cps);
  return (int64_t)sp * Cout * 9 * Cs + (int64_t)sp * Cout;
}

extern "C" int scan_conv3x3_wgrad_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const float* dy,
                                         int32_t Cout, int32_t Cout_s, float* dw, float* db, int32_t accumulate,
                                         float* ws, void* stream) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1,
                 "conv3x3_wgrad_bf16x3: bad pyramid");
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "conv3x3_wgrad_bf16x3: Cs=%d must be a positive multiple of 4", Cs);
  SCAN_CHECK_ARG(Cout > 0 && Cout_s >= Cout, "conv3x3_wgrad_bf16x3: Cout=%d Cout_s=%d", Cout, Cout_s);
  SCAN_CHECK_ARG(x && dy && dw && ws, "conv3x3_wgrad_bf16x3: null pointer");
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad3_plan(d, Cs, Cout, &ct, &nt, &ctl, &sp, &cps);
  hipStream_t st = as_stream(stream);
  const size_t sh = (size_t)WBUF(3) * sizeof(__bf16);
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_bf16x3_kernel<3, 1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    done = true;
  }
  float* bias_slab = db ? ws + (int64_t)sp * Cout * 9 * Cs : nullptr;
  if (g_scan_wgrad_v3 == 1 || (g_scan_wgrad_v3 == 2 && Cs % 128 == 0)) {
    const size_t sh3 = (size_t)2 * W3STAGE(3) * sizeof(__bf16);
    static bool done3 = false;
    if (!done3) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v3_kernel<3, 1, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh3);
      done3 = true;
    }
    hipLaunchKernelGGL((conv_wgrad_bf16x3_v3_kernel<3, 1, 512>), dim3(nt * sp), dim3(512), sh3, st, x, *d, Cs, dy, Cout,
                       Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
  } else if (g_scan_wgrad_v6 && g_scan_wgrad_v4 && Cs % 4 == 0 && Cout_s % 4 == 0 &&
             (g_scan_wgrad_exp == 0 || g_scan_wgrad_exp >= 5) && (g_scan_wgrad_v4 == 2 || wgrad_use_v2(Cs))) {
    const size_t sh6 = (size_t)2 * W6STAGE(3) * sizeof(__bf16);
    static bool done7 = false;
    if (!done7) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v6_kernel<3>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh6);
      done7 = true;
    }
    hipLaunchKernelGGL((conv_wgrad_bf16x3_v6_kernel<3>), dim3(nt * sp), dim3(768), sh6, st, x, *d, Cs, dy, Cout, Cout_s, ws,
                       bias_slab, ct, nt, ctl, cps, sp, g_scan_wgrad_exp >= 5 ? g_scan_wgrad_exp - 4 : 0, g_scan_wgrad_prio);
  } else if (g_scan_wgrad_v5 && g_scan_wgrad_v4 && Cs % 4 == 0 && Cout_s % 4 == 0 && g_scan_wgrad_exp == 0 &&
             (g_scan_wgrad_v4 == 2 || wgrad_use_v2(Cs))) {
    const size_t sh5 = (size_t)2 * W3STAGE(3) * sizeof(__bf16);
    static bool done6 = false;
    if (!done6) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v5_kernel<3, 1, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh5);
      done6 = true;
    }
    hipLaunchKernelGGL((conv_wgrad_bf16x3_v5_kernel<3, 1, 512>), dim3(nt * sp), dim3(512), sh5, st, x, *d, Cs, dy, Cout,
                       Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
  } else if (g_scan_wgrad_v4 && Cs % 4 == 0 && Cout_s % 4 == 0 && (g_scan_wgrad_v4 == 2 || wgrad_use_v2(Cs)) &&
             g_scan_wgrad_exp == 0) {
    static bool done5 = false;
    if (!done5) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v4_kernel<3, 1, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v4_kernel<3, 1, 512, 1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v4_kernel<3, 1, 512, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v4_kernel<3, 1, 512, 3>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      done5 = true;
    }
    if (g_scan_wgrad_il == 1)
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v4_kernel<3, 1, 512, 1>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy,
                         Cout, Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
    else if (g_scan_wgrad_il == 2)
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v4_kernel<3, 1, 512, 2>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy,
                         Cout, Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
    else if (g_scan_wgrad_il == 3)
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v4_kernel<3, 1, 512, 3>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy,
                         Cout, Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
    else
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v4_kernel<3, 1, 512>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy, Cout,
                         Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
  } else if (wgrad_use_v2(Cs)) {
    static bool done2 = false;
    if (!done2) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<3, 1, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<3, 1, 1024>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      done2 = true;
    }
    if (g_scan_wgrad_exp >= 1 && g_scan_wgrad_exp <= 4) {
      static bool donex = false;
      if (!donex) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<3, 1, 512, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<3, 1, 512, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<3, 1, 512, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<3, 1, 512, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        donex = true;
      }
#define SCAN_WG_EXP(E) hipLaunchKernelGGL((conv_wgrad_bf16x3_v2_kernel<3, 1, 512, E>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy, Cout, Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d)
      if (g_scan_wgrad_exp == 1) SCAN_WG_EXP(1);
      else if (g_scan_wgrad_exp == 2) SCAN_WG_EXP(2);
      else if (g_scan_wgrad_exp == 3) SCAN_WG_EXP(3);
      else SCAN_WG_EXP(4);
#undef SCAN_WG_EXP
    } else if (g_scan_wgrad_wg1024 == 1)
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v2_kernel<3, 1, 1024>), dim3(nt * sp), dim3(1024), sh, st, x, *d, Cs, dy, Cout,
                         Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
    else
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v2_kernel<3, 1, 512>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy, Cout,
                         Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
  } else if (g_scan_wgrad_wg1024 == 2) {
    static bool done4 = false;
    if (!done4) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_bf16x3_kernel<3, 1, 1024>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      done4 = true;
    }
    hipLaunchKernelGGL((conv3x3_wgrad_bf16x3_kernel<3, 1, 1024>), dim3(nt * sp), dim3(1024), sh, st, x, *d, Cs, dy, Cout,
                       Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
  } else {
    hipLaunchKernelGGL((conv3x3_wgrad_bf16x3_kernel<3, 1>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy, Cout,
                       Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *d);
  }
  SCAN_LAUNCH_CHECK("conv3x3_wgrad_bf16x3");
  // one launch reduces the weight slabs and (last block) the bias slabs
  const int64_t n = (int64_t)Cout * 9 * Cs;
  hipLaunchKernelGGL(slab_bias_reduce_kernel, dim3(grid_for(n / 4, 256) + (db ? 1 : 0)), dim3(256), 0, st, ws, sp, n, dw,
                     bias_slab, Cout, db, accumulate);
  SCAN_LAUNCH_CHECK("slab_bias_reduce");
  return 0;
}

// ---- 1x1 weight gradient (stride 1 or 2): dw[Cout][1][Cs] = sum_pixels dY^T X, same kernel with one tap.
extern "C" int64_t scan_conv1x1_wgrad_bf16x3_ws_floats(const scan_pyramid_t* yd, int32_t Cs, int32_t Cout) {
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad3_plan(yd, Cs, Cout, &ct, &nt, &ctl, &sp, &cps, 1);
  return (int64_t)sp * Cout * Cs + (int64_t)sp * Cout;
}

extern "C" int scan_conv1x1_wgrad_bf16x3(const float* x, const scan_pyramid_t* xd, int32_t Cs, const float* dy,
                                         const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t stride,
                                         float* dw, float* db, int32_t accumulate, float* ws, void* stream) {
  SCAN_CHECK_ARG(xd && yd && yd->n_levels >= 1 && yd->n_levels <= SCAN_MAX_LEVELS && yd->n_images >= 1 &&
                     xd->n_levels == yd->n_levels && xd->n_images == yd->n_images,
                 "conv1x1_wgrad_bf16x3: bad pyramids");
  SCAN_CHECK_ARG(stride == 1 || stride == 2, "conv1x1_wgrad_bf16x3: stride must be 1 or 2, got %d", stride);
  for (int l = 0; l < yd->n_levels; ++l)
    SCAN_CHECK_ARG((xd->h[l] - 1) / stride + 1 == yd->h[l] && (xd->w[l] - 1) / stride + 1 == yd->w[l],
                   "conv1x1_wgrad_bf16x3: level %d sizes do not match stride %d", l, stride);
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "conv1x1_wgrad_bf16x3: Cs=%d must be a positive multiple of 4", Cs);
  SCAN_CHECK_ARG(Cout > 0 && Cout_s >= Cout, "conv1x1_wgrad_bf16x3: Cout=%d Cout_s=%d", Cout, Cout_s);
  SCAN_CHECK_ARG(x && dy && dw && ws, "conv1x1_wgrad_bf16x3: null pointer");
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad3_plan(yd, Cs, Cout, &ct, &nt, &ctl, &sp, &cps, 1);
  hipStream_t st = as_stream(stream);
  const size_t sh = (size_t)WBUF(1) * sizeof(__bf16);
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_bf16x3_kernel<1, 1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_bf16x3_kernel<1, 2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    done = true;
  }
  float* bias_slab = db ? ws + (int64_t)sp * Cout * Cs : nullptr;
  if (g_scan_wgrad_v4 && Cs % 4 == 0 && Cout_s % 4 == 0 && (g_scan_wgrad_v4 == 2 || wgrad_use_v2(Cs))) {
    static bool done5 = false;
    if (!done5) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v4_kernel<1, 1, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v4_kernel<1, 2, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      done5 = true;
    }
    if (stride == 1)
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v4_kernel<1, 1, 512>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout,
                         Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *xd);
    else
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v4_kernel<1, 2, 512>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout,
                         Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *xd);
  } else if (wgrad_use_v2(Cs)) {
    static bool done2 = false;
    if (!done2) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<1, 1, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16x3_v2_kernel<1, 2, 512>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      done2 = true;
    }
    if (stride == 1)
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v2_kernel<1, 1, 512>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout,
                         Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *xd);
    else
      hipLaunchKernelGGL((conv_wgrad_bf16x3_v2_kernel<1, 2, 512>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout,
                         Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *xd);
  } else if (stride == 1)
    hipLaunchKernelGGL((conv3x3_wgrad_bf16x3_kernel<1, 1>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout,
                       Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *xd);
  else
    hipLaunchKernelGGL((conv3x3_wgrad_bf16x3_kernel<1, 2>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout,
                       Cout_s, ws, bias_slab, ct, nt, ctl, cps, sp, *xd);
  SCAN_LAUNCH_CHECK("conv1x1_wgrad_bf16x3");
  const int64_t n = (int64_t)Cout * Cs;
  hipLaunchKernelGGL(slab_bias_reduce_kernel, dim3(grid_for(n / 4, 256) + (db ? 1 : 0)), dim3(256), 0, st, ws, sp, n, dw,
                     bias_slab, Cout, db, accumulate);
  SCAN_LAUNCH_CHECK("slab_bias_reduce");
  return 0;
}
