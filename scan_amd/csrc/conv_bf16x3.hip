// Public entry points of the split-operand convolutions (forward / data gradient) and the weight split, plus the FIRST
// generation of the forward / data-gradient kernel.
//
// Arithmetic: conv_split.h.  "bf16x3" = every fp32 operand cut into two bf16 pieces, three piece products (16 significand
// bits per operand: ~2^-16 per product, 2e-6 relative on the losses of a full DA iteration); "bf16x6" = three pieces, six
// piece products (all 24 significand bits: the reference's fp32 multiply, fp32 accumulate).  The launches go to
// conv_fwd.hip (v_mfma_f32_16x16x32_bf16).
//
// The first-generation kernel below (v_mfma_f32_32x32x16_bf16, two pieces, another tile / fragment / LDS layout) is kept as
// an INDEPENDENT implementation the tests compare the production kernel against (scan_tune("conv_v2", 0) routes the
// two-piece launches here).  Structure (per 512-thread workgroup, one workgroup per CU: 95 KB of LDS):
//   output tile   16 x 16 pixels (one image, one pyramid level) x BN = 128 or 256 output channels
//                 (8 x 16 pixels x 64 channels with 256 threads for Cout <= 64)
//   K loop        input channels in chunks of 32; per chunk the 18 x 18 x 32 input HALO patch is read from HBM/L2 ONCE
//                 as fp32, split to bf16 hi/lo while being written to LDS, and then reused by all 9 taps
//   weights       pre-split once per step into bf16 planes [O][9][Csw] (scan_weight_split), staged per (chunk, tap)
//                 through a double-buffered LDS tile
//   waves         4 x 2, each 64 pixels x 64 (BN = 256: 128) channels = 2 x 2 (2 x 4) MFMA tiles of 32x32
// LDS pixel rows are 80 B (64 B of data + 16 B pad) and patch rows 1536 B so the 16-byte fragment reads of consecutive
// pixels / channels fall on distinct bank groups.
//
// dgrad reuses the forward kernels: dX = conv3x3(dY, W') with W'[c][t][o] = W[o][8-t][c]
// (scan_weight_split mode 1 writes the flipped + transposed copy).
#include "conv_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SCAN_MMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

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


// w [O][T][Cs] fp32 -> NP bf16 planes (conv_split.h).
//   mode 0: out[o][t][c]            (O rows, row length Csw >= Cs, zero padded)       -- forward
//   mode 1: out[c][T-1-t][o]        (Cs rows, row length Csw >= O, zero padded)       -- dgrad (flip + transpose)
template <int NP>
__global__ void weight_split_kernel(const float* __restrict__ w, int O, int T, int Cs, int mode, int rows, int Csw,
                                    __bf16* __restrict__ w0, __bf16* __restrict__ w1, __bf16* __restrict__ w2) {
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
    __bf16 q[NP];
    split1_np<NP>(v, q);
    w0[i] = q[0];
    w1[i] = q[1];
    if constexpr (NP == 3) w2[i] = q[2];
  }
}

// 1 (default): two-piece forward / data-gradient launches go to conv_fwd.hip; 0: the 32x32x16 kernel of this file (the
// independent implementation the tests compare against)
int g_scan_conv_v2 = 1;
extern int g_scan_conv_bn256;
int conv3x3_split_launch(int np, const float* x, const scan_pyramid_t* d, int32_t Cs, const void* w0, const void* w1,
                         const void* w2, int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout,
                         int32_t Ns, int32_t relu, void* stream, double* gn_ws);
int conv1x1_split_launch(int np, const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* w0, const void* w1,
                         const void* w2, int32_t Csw, const float* bias, const float* mask, float* y,
                         const scan_pyramid_t* yd, int32_t Nout, int32_t Ns, int32_t relu, int32_t map, void* stream);
// the production kernel stores float4 and loads the ReLU mask as float4
static inline bool v2_ok(const void* y, const void* mask, int32_t Ns) {
  return (Ns & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
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

static int weight_split_launch(int np, const float* w, int32_t O, int32_t T, int32_t Cs, int32_t mode, void* w0, void* w1,
                               void* w2, int32_t Csw, void* stream) {
  SCAN_CHECK_ARG(w && w0 && w1 && (np == 2 || w2) && O > 0 && T > 0 && Cs > 0, "weight_split: bad arguments");
  SCAN_CHECK_ARG(mode == 0 || mode == 1, "weight_split: mode must be 0 or 1");
  SCAN_CHECK_ARG(Csw % 8 == 0 && Csw >= (mode == 0 ? Cs : O), "weight_split: Csw=%d must be a multiple of 8 and cover the row",
                 Csw);
  const int rows = mode == 0 ? O : Cs;
  const int64_t total = (int64_t)rows * T * Csw;
  __bf16 *p0 = reinterpret_cast<__bf16*>(w0), *p1 = reinterpret_cast<__bf16*>(w1), *p2 = reinterpret_cast<__bf16*>(w2);
  if (np == 3)
    hipLaunchKernelGGL(weight_split_kernel<3>, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), w, O, T, Cs, mode,
                       rows, Csw, p0, p1, p2);
  else
    hipLaunchKernelGGL(weight_split_kernel<2>, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), w, O, T, Cs, mode,
                       rows, Csw, p0, p1, p2);
  SCAN_LAUNCH_CHECK("weight_split");
  return 0;
}

extern "C" int scan_weight_split(const float* w, int32_t O, int32_t T, int32_t Cs, int32_t mode, void* wh, void* wl,
                                 int32_t Csw, void* stream) {
  return weight_split_launch(2, w, O, T, Cs, mode, wh, wl, nullptr, Csw, stream);
}
extern "C" int scan_weight_split3(const float* w, int32_t O, int32_t T, int32_t Cs, int32_t mode, void* wh, void* wm,
                                  void* wl, int32_t Csw, void* stream) {
  return weight_split_launch(3, w, O, T, Cs, mode, wh, wm, wl, Csw, stream);
}

// y[M][Ns] = conv3x3_s1(x[M][Cs]) with pre-split weight planes [Nout][9][Csw]; same pyramid in and out.
static int conv3x3_launch(int np, const float* x, const scan_pyramid_t* d, int32_t Cs, const void* w0, const void* w1,
                          const void* w2, int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout,
                          int32_t Ns, int32_t relu, void* stream, double* gn_ws = nullptr) {
  const char* name = np == 2 ? "conv3x3_bf16x3" : "conv3x3_bf16x6";
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1, "%s: bad pyramid", name);
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "%s: Cs=%d must be a positive multiple of 4", name, Cs);
  SCAN_CHECK_ARG(Csw % 8 == 0 && Csw >= Cs, "%s: Csw=%d must be a multiple of 8 and >= Cs", name, Csw);
  SCAN_CHECK_ARG(Nout > 0 && Ns >= Nout, "%s: Nout=%d Ns=%d", name, Nout, Ns);
  SCAN_CHECK_ARG(x && w0 && w1 && (np == 2 || w2) && y, "%s: null pointer", name);
  if (np == 3) {
    SCAN_CHECK_ARG(v2_ok(y, mask, Ns), "%s: y / mask must be 16-byte aligned and Ns a multiple of 4", name);
    return conv3x3_split_launch(3, x, d, Cs, w0, w1, w2, Csw, bias, mask, y, Nout, Ns, relu, stream, gn_ws);
  }
  if (g_scan_conv_v2 && v2_ok(y, mask, Ns))
    return conv3x3_split_launch(2, x, d, Cs, w0, w1, nullptr, Csw, bias, mask, y, Nout, Ns, relu, stream, gn_ws);
  TileTab tt;
  hipStream_t st = as_stream(stream);
  const __bf16* h = reinterpret_cast<const __bf16*>(w0);
  const __bf16* l = reinterpret_cast<const __bf16*>(w1);
  if (Nout > 64) {
    // 16 x 16 pixel tiles, 512 threads (8 waves = 4 x 2).  BN = 128: each wave 64 px x 64 ch; BN = 256: 64 px x 128 ch,
    // used when the output channels fill 256-wide tiles and the launch still has >= 2 workgroups per CU.
    make_tiles(d, &tt, 16);
    const int tiles = tt.tile_off[d->n_levels];
    const bool wide = g_scan_conv_bn256 && Nout % 256 == 0 && (int64_t)tiles * (Nout / 256) >= 512;
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
  return conv3x3_launch(2, x, d, Cs, wh, wl, nullptr, Csw, bias, mask, y, Nout, Ns, relu ? 1 : 0, stream);
}
extern "C" int scan_conv3x3_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wm,
                                   const void* wl, int32_t Csw, const float* bias, const float* mask, float* y,
                                   int32_t Nout, int32_t Ns, int32_t relu, void* stream) {
  return conv3x3_launch(3, x, d, Cs, wh, wm, wl, Csw, bias, mask, y, Nout, Ns, relu ? 1 : 0, stream);
}

// conv3x3 + bias whose output feeds GroupNorm(32, 256): the epilogue also accumulates the per-(level, image, group)
// sum and sum of squares into gn_ws (fp64, n_levels * n_images * 32 * 2 values), which scan_groupnorm_stats_from_sums turns
// into (mean, rstd) -- the separate statistics pass over y disappears.  clear != 0: gn_ws is zeroed here; 0: the sums are
// ADDED to gn_ws as it is (scan_amd/ops.py hands out slices of one buffer it clears with one memset per training iteration
// instead of one memset launch per call).
static int conv3x3_gn_launch(int np, int clear, const float* x, const scan_pyramid_t* d, int32_t Cs, const void* w0,
                             const void* w1, const void* w2, int32_t Csw, const float* bias, float* y, int32_t Nout,
                             int32_t Ns, float* gn_ws, void* stream) {
  SCAN_CHECK_ARG(Nout == 256 && gn_ws, "conv3x3_gn: needs Nout == 256 (GroupNorm(32, 256)) and a workspace");
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1, "conv3x3_gn: bad pyramid");
  if (clear) {
    const size_t bytes = sizeof(double) * 2 * 32 * (size_t)d->n_levels * d->n_images;
    if (hipMemsetAsync(gn_ws, 0, bytes, as_stream(stream)) != hipSuccess) {
      scan_set_error("conv3x3_gn: memset failed");
      return -2;
    }
  }
  return conv3x3_launch(np, x, d, Cs, w0, w1, w2, Csw, bias, nullptr, y, Nout, Ns, 0, stream, reinterpret_cast<double*>(gn_ws));
}
extern "C" int scan_conv3x3_gn_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                                      int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns, float* gn_ws,
                                      void* stream) {
  return conv3x3_gn_launch(2, 1, x, d, Cs, wh, wl, nullptr, Csw, bias, y, Nout, Ns, gn_ws, stream);
}
extern "C" int scan_conv3x3_gn_acc_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                                          int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns, float* gn_ws,
                                          void* stream) {
  return conv3x3_gn_launch(2, 0, x, d, Cs, wh, wl, nullptr, Csw, bias, y, Nout, Ns, gn_ws, stream);
}
extern "C" int scan_conv3x3_gn_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wm,
                                      const void* wl, int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns,
                                      float* gn_ws, int32_t clear, void* stream) {
  return conv3x3_gn_launch(3, clear, x, d, Cs, wh, wm, wl, Csw, bias, y, Nout, Ns, gn_ws, stream);
}

// conv3x3 + bias (+ ReLU) + 2x2 / stride-2 max-pool in one launch: y [N, H/2, W/2, Ns] (forward only; single-level
// pyramid with even H, W).  max and the monotone bias / ReLU commute, so the result equals pooling the conv output.
extern "C" int scan_conv3x3_pool2_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh,
                                         const void* wl, int32_t Csw, const float* bias, float* y, int32_t Nout,
                                         int32_t Ns, int32_t relu, void* stream) {
  SCAN_CHECK_ARG(d && d->n_levels == 1 && (d->h[0] & 1) == 0 && (d->w[0] & 1) == 0,
                 "conv3x3_pool2_bf16x3: needs a single-level pyramid with even H and W");
  return conv3x3_launch(2, x, d, Cs, wh, wl, nullptr, Csw, bias, nullptr, y, Nout, Ns, (relu ? 1 : 0) | 2, stream);
}
extern "C" int scan_conv3x3_pool2_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh,
                                         const void* wm, const void* wl, int32_t Csw, const float* bias, float* y,
                                         int32_t Nout, int32_t Ns, int32_t relu, void* stream) {
  SCAN_CHECK_ARG(d && d->n_levels == 1 && (d->h[0] & 1) == 0 && (d->w[0] & 1) == 0,
                 "conv3x3_pool2_bf16x6: needs a single-level pyramid with even H and W");
  return conv3x3_launch(3, x, d, Cs, wh, wm, wl, Csw, bias, nullptr, y, Nout, Ns, (relu ? 1 : 0) | 2, stream);
}

// y[Mo][Ns] = conv1x1(x[Mi][Cs]) with pre-split weight planes [Nout][1][Csw].  map 0: stride 1 (xd == yd);
// map 1: stride 2 forward (yd = xd.conv_out(1, 2)); map 2: data gradient of a stride-2 1x1 conv (x = dY on the coarse
// pyramid xd, y = dX on the fine pyramid yd, zero where a coordinate is odd).
static int conv1x1_launch(int np, const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* w0, const void* w1,
                          const void* w2, int32_t Csw, const float* bias, const float* mask, float* y,
                          const scan_pyramid_t* yd, int32_t Nout, int32_t Ns, int32_t relu, int32_t map, void* stream) {
  const char* name = np == 2 ? "conv1x1_bf16x3" : "conv1x1_bf16x6";
  SCAN_CHECK_ARG(xd && yd && yd->n_levels >= 1 && yd->n_levels <= SCAN_MAX_LEVELS && yd->n_images >= 1 &&
                     xd->n_levels == yd->n_levels && xd->n_images == yd->n_images,
                 "%s: bad pyramids", name);
  SCAN_CHECK_ARG(map >= 0 && map <= 2, "%s: map=%d must be 0, 1 or 2", name, map);
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "%s: Cs=%d must be a positive multiple of 4", name, Cs);
  SCAN_CHECK_ARG(Csw % 8 == 0 && Csw >= Cs, "%s: Csw=%d must be a multiple of 8 and >= Cs", name, Csw);
  SCAN_CHECK_ARG(Nout > 0 && Ns >= Nout, "%s: Nout=%d Ns=%d", name, Nout, Ns);
  SCAN_CHECK_ARG(x && w0 && w1 && (np == 2 || w2) && y, "%s: null pointer", name);
  for (int l = 0; l < yd->n_levels; ++l) {
    const int eh = map == 0 ? xd->h[l] : map == 1 ? (xd->h[l] - 1) / 2 + 1 : yd->h[l];
    const int ew = map == 0 ? xd->w[l] : map == 1 ? (xd->w[l] - 1) / 2 + 1 : yd->w[l];
    SCAN_CHECK_ARG(eh == yd->h[l] && ew == yd->w[l] &&
                       (map != 2 || ((yd->h[l] - 1) / 2 + 1 == xd->h[l] && (yd->w[l] - 1) / 2 + 1 == xd->w[l])),
                   "%s: level %d sizes do not match map %d", name, l, map);
  }
  if (np == 3) {
    SCAN_CHECK_ARG(v2_ok(y, mask, Ns), "%s: y / mask must be 16-byte aligned and Ns a multiple of 4", name);
    return conv1x1_split_launch(3, x, xd, Cs, w0, w1, w2, Csw, bias, mask, y, yd, Nout, Ns, relu, map, stream);
  }
  if (g_scan_conv_v2 && v2_ok(y, mask, Ns))
    return conv1x1_split_launch(2, x, xd, Cs, w0, w1, nullptr, Csw, bias, mask, y, yd, Nout, Ns, relu, map, stream);
  TileTab tt;
  hipStream_t st = as_stream(stream);
  const __bf16* h = reinterpret_cast<const __bf16*>(w0);
  const __bf16* l = reinterpret_cast<const __bf16*>(w1);
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

extern "C" int scan_conv1x1_bf16x3(const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* wh, const void* wl,
                                   int32_t Csw, const float* bias, const float* mask, float* y,
                                   const scan_pyramid_t* yd, int32_t Nout, int32_t Ns, int32_t relu, int32_t map,
                                   void* stream) {
  return conv1x1_launch(2, x, xd, Cs, wh, wl, nullptr, Csw, bias, mask, y, yd, Nout, Ns, relu, map, stream);
}
extern "C" int scan_conv1x1_bf16x6(const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* wh, const void* wm,
                                   const void* wl, int32_t Csw, const float* bias, const float* mask, float* y,
                                   const scan_pyramid_t* yd, int32_t Nout, int32_t Ns, int32_t relu, int32_t map,
                                   void* stream) {
  return conv1x1_launch(3, x, xd, Cs, wh, wm, wl, Csw, bias, mask, y, yd, Nout, Ns, relu, map, stream);
}
