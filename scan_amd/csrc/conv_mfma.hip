// fp32-MFMA implicit-GEMM convolution for gfx950 (MI355X): forward, data-gradient
// (transposed gather) and weight-gradient (deterministic split-K) on "pyramid"
// [M, C] activations.  Written for CDNA4 directly: 64-lane wavefronts,
// v_mfma_f32_32x32x2_f32 tiles, LDS-staged operands, XCD-aware tile order.
//
// Replaces nn.Conv2d/F.conv2d at the reference call sites listed in
// include/scan_hip.h (VGG body, FPN, condgraph/FCOS/CKA towers).
//
// GEMM view (forward / dgrad):  Y[m][o] = sum_k A[m][k] * B[o][k]
//   m = output pixel (pyramid row), k = (tap, channel), A gathered on the fly from
//   the input pyramid (zero outside the image), B = weights [O][T][Cs].
// GEMM view (wgrad):            dW[o][tap][c] = sum_m dY[m][o] * X[g(m,tap)][c]
//
// fp32 MFMA on gfx950 runs at the fp32 vector rate (64 cycles per 32x32x2 per SIMD),
// so one MFMA hides many LDS reads; the kernels below therefore favour simple,
// conflict-light LDS layouts over elaborate swizzles.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BK 32
#define LDA (BK + 4)  // padded row stride (floats) of the K-contiguous LDS tiles

// bijective XCD-aware remap (blocks b and b+8 share an XCD's L2): give each XCD a
// contiguous range of tiles so neighbouring tiles (shared weights / halos) share L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / 8;
}

struct RowInfo {
  int64_t rowbase;  // first row (pixel (0,0)) of this row's image in the GATHER-SOURCE pyramid
  int y, x;         // output coordinates
  int hs, ws;       // source level height/width (0 => row out of range)
};

// MODE 0: forward  (src = x, iy = y*stride - pad + ky)
// MODE 1: dgrad    (src = dy, iy = (y + pad - ky)/stride when divisible)
template <int MODE>
__device__ __forceinline__ int64_t gather_row(const RowInfo& ri, int ky, int kx, int stride, int pad) {
  int iy, ix;
  if (MODE == 0) {
    iy = ri.y * stride - pad + ky;
    ix = ri.x * stride - pad + kx;
  } else {
    int ny = ri.y + pad - ky, nx = ri.x + pad - kx;
    if (stride == 2) {
      if ((ny | nx) & 1) return -1;
      iy = ny >> 1;
      ix = nx >> 1;
    } else {
      iy = ny;
      ix = nx;
    }
  }
  if (iy < 0 || ix < 0 || iy >= ri.hs || ix >= ri.ws) return -1;
  return ri.rowbase + (int64_t)iy * ri.ws + ix;
}

// ------------------------------------------------------------------------------------------
// forward / dgrad kernel.  Block = 256 threads = 4 waves.
//   NT = 4: block tile 128 x 128, waves 2x2, each 64x64 (2x2 MFMA tiles)
//   NT = 1: block tile 128 x 32,  waves 4x1, each 32x32 (skinny outputs: Cout <= 32)
// ------------------------------------------------------------------------------------------
template <int MODE, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const float* __restrict__ src, scan_pyramid_t sd, int Cs,  // gather source, its pyramid, row stride
    const float* __restrict__ wgt,                              // [Nout][T][Cs]
    const float* __restrict__ bias,                             // [Nout] or null
    const float* __restrict__ mask,                             // [M][Ns] or null (dgrad relu mask)
    float* __restrict__ dst, scan_pyramid_t dd, int Nout, int Ns, int ksize, int stride, int relu, int m_tiles,
    int n_tiles) {
  constexpr int BN = (NT == 4) ? 128 : 32;
  constexpr int WM = (NT == 4) ? 2 : 1;  // MFMA tiles per wave along M
  constexpr int WN = (NT == 4) ? 2 : 1;  // and along N
  constexpr int BROWS = BN / 32;         // B float4 per thread

  extern __shared__ float smem[];
  float* As = smem;                 // [2][BM][LDA]
  float* Bs = smem + 2 * BM * LDA;  // [2][BN][LDA]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int n_tile = bid % n_tiles, m_tile = bid / n_tiles;
  const int64_t m0 = (int64_t)m_tile * BM;
  const int n0 = n_tile * BN;
  const int64_t M = dd.row_off[dd.n_levels];
  const int T = ksize * ksize, pad = ksize / 2;
  const int cchunks = (Cs + BK - 1) / BK;
  const int nchunks = T * cchunks;

  // staging roles: thread -> (row r0 + 32*i, float4 column c4)
  const int c4 = tid & 7, r0 = tid >> 3;
  RowInfo ri[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + r0 + 32 * i;
    if (m < M) {
      RowCoord rc = decode_row(dd, m);
      ri[i].y = rc.y;
      ri[i].x = rc.x;
      ri[i].hs = sd.h[rc.lvl];
      ri[i].ws = sd.w[rc.lvl];
      ri[i].rowbase = sd.row_off[rc.lvl] + (int64_t)rc.n * sd.h[rc.lvl] * sd.w[rc.lvl];
    } else {
      ri[i].y = ri[i].x = 0;
      ri[i].hs = ri[i].ws = 0;
      ri[i].rowbase = 0;
    }
  }

  float4 ra[4], rb[BROWS];
  auto load_chunk = [&](int ch) {
    const int tap = ch / cchunks, c0 = (ch - tap * cchunks) * BK;
    const int ky = tap / ksize, kx = tap - ky * ksize;
    const int c = c0 + 4 * c4;
    const bool cok = c < Cs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = gather_row<MODE>(ri[i], ky, kx, stride, pad);
      ra[i] = (row >= 0 && cok) ? *reinterpret_cast<const float4*>(src + row * Cs + c) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const int o = n0 + r0 + 32 * i;
      rb[i] = (o < Nout && cok) ? *reinterpret_cast<const float4*>(wgt + ((int64_t)o * T + tap) * Cs + c)
                                : make_float4(0, 0, 0, 0);
    }
  };
  auto store_chunk = [&](int buf) {
    float* a = As + buf * BM * LDA;
    float* b = Bs + buf * BN * LDA;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(a + (r0 + 32 * i) * LDA + 4 * c4) = ra[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4*>(b + (r0 + 32 * i) * LDA + 4 * c4) = rb[i];
  };

  // wave -> sub-tile
  const int wm = (NT == 4) ? (wid >> 1) : wid;  // wave row
  const int wn = (NT == 4) ? (wid & 1) : 0;     // wave col
  const int lrow = lane & 31, lh = lane >> 5;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunks) load_chunk(ch + 1);
    const float* a = As + buf * BM * LDA + (wm * WM * 32 + lrow) * LDA + 4 * lh;
    const float* b = Bs + buf * BN * LDA + (wn * WN * 32 + lrow) * LDA + 4 * lh;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
      float4 av[WM], bv[WN];
#pragma unroll
      for (int t = 0; t < WM; ++t) av[t] = *reinterpret_cast<const float4*>(a + t * 32 * LDA + 8 * j);
#pragma unroll
      for (int t = 0; t < WN; ++t) bv[t] = *reinterpret_cast<const float4*>(b + t * 32 * LDA + 8 * j);
      // lane half h supplies k = 8j + 4h + e for element e: A and B use the same k permutation
#pragma unroll
      for (int tm = 0; tm < WM; ++tm)
#pragma unroll
        for (int tn = 0; tn < WN; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].z, bv[tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].w, bv[tn].w, acc[tm][tn], 0, 0, 0);
        }
    }
    if (ch + 1 < nchunks) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
  for (int tn = 0; tn < WN; ++tn) {
    const int o = n0 + (wn * WN + tn) * 32 + lrow;
    const float bv = (bias != nullptr && o < Nout) ? bias[o] : 0.f;
#pragma unroll
    for (int tm = 0; tm < WM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + (wm * WM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && o < Nout) {
          float v = acc[tm][tn][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          if (mask != nullptr) v = (mask[m * Ns + o] > 0.f) ? v : 0.f;
          dst[m * Ns + o] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad kernel: block tile 128 (o) x 128 (c) for one tap; K' = pixel rows, split over gridDim.y.
// LDS tiles are [k][128 + 4] (M/N contiguous), lanes read one float per operand per k.
// ------------------------------------------------------------------------------------------
#define LDW (128 + 4)
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, scan_pyramid_t xd, int Cs,
                                                         const float* __restrict__ dy, scan_pyramid_t yd, int Nout,
                                                         int Ns, int ksize, int stride, float* __restrict__ slab,
                                                         int o_tiles, int c_tiles, int chunks_per_split) {
  extern __shared__ float smem[];
  typedef float (*tile_t)[BK][LDW];
  tile_t As = reinterpret_cast<tile_t>(smem);                 // [2] dY  [k = row][o]
  tile_t Bs = reinterpret_cast<tile_t>(smem + 2 * BK * LDW);  // [2] X   [k = row][c]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int T = ksize * ksize, pad = ksize / 2;
  int bid = blockIdx.x;
  const int c_tile = bid % c_tiles;
  bid /= c_tiles;
  const int tap = bid % T;
  const int o_tile = bid / T;
  const int o0 = o_tile * 128, c0 = c_tile * 128;
  const int ky = tap / ksize, kx = tap - ky * ksize;
  const int64_t M = yd.row_off[yd.n_levels];
  const int64_t total_chunks = (M + BK - 1) / BK;
  const int64_t ch_begin = (int64_t)blockIdx.y * chunks_per_split;
  int64_t ch_end = ch_begin + chunks_per_split;
  if (ch_end > total_chunks) ch_end = total_chunks;

  const int q4 = tid & 31, rr = tid >> 5;  // float4 column, row (rows rr + 8*i)
  float4 ra[4], rb[4];
  auto load_chunk = [&](int64_t ch) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = ch * BK + rr + 8 * i;
      ra[i] = make_float4(0, 0, 0, 0);
      rb[i] = make_float4(0, 0, 0, 0);
      if (m < M) {
        const int o = o0 + 4 * q4;
        if ((Ns & 3) == 0 && o + 3 < Ns) {
          ra[i] = *reinterpret_cast<const float4*>(dy + m * Ns + o);
        } else {
          float t[4] = {0, 0, 0, 0};
          for (int e = 0; e < 4; ++e)
            if (o + e < Ns) t[e] = dy[m * Ns + o + e];
          ra[i] = make_float4(t[0], t[1], t[2], t[3]);
        }
        const int c = c0 + 4 * q4;
        if (c < Cs) {
          RowCoord rc = decode_row(yd, m);
          RowInfo ri;
          ri.y = rc.y;
          ri.x = rc.x;
          ri.hs = xd.h[rc.lvl];
          ri.ws = xd.w[rc.lvl];
          ri.rowbase = xd.row_off[rc.lvl] + (int64_t)rc.n * ri.hs * ri.ws;
          const int64_t row = gather_row<0>(ri, ky, kx, stride, pad);
          if (row >= 0) rb[i] = *reinterpret_cast<const float4*>(x + row * Cs + c);
        }
      }
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&As[buf][rr + 8 * i][4 * q4]) = ra[i];
      *reinterpret_cast<float4*>(&Bs[buf][rr + 8 * i][4 * q4]) = rb[i];
    }
  };

  const int wm = wid >> 1, wn = wid & 1;
  const int lrow = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if (ch_begin < ch_end) {
    load_chunk(ch_begin);
    store_chunk(0);
  }
  __syncthreads();
  for (int64_t ch = ch_begin; ch < ch_end; ++ch) {
    const int buf = (int)((ch - ch_begin) & 1);
    if (ch + 1 < ch_end) load_chunk(ch + 1);
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const int k = 2 * s + lh;
      float av[2], bv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        av[t] = As[buf][k][(wm * 2 + t) * 32 + lrow];
        bv[t] = Bs[buf][k][(wn * 2 + t) * 32 + lrow];
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
    }
    if (ch + 1 < ch_end) store_chunk(buf ^ 1);
    __syncthreads();
  }

  // slab[split][o][tap][c]
  float* out = slab + (int64_t)blockIdx.y * Nout * T * Cs;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int c = c0 + (wn * 2 + tn) * 32 + lrow;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + (wm * 2 + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (o < Nout && c < Cs) out[((int64_t)o * T + tap) * Cs + c] = acc[tm][tn][r];
      }
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, int splits, int64_t n, float* __restrict__ dw,
                                   int accumulate) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 s = make_float4(0, 0, 0, 0);
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
}

extern "C" void scan_slab_reduce_launch(const float* slab, int splits, int64_t n, float* dw, int accumulate,
                                        hipStream_t st) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, st, slab, splits, n, dw, accumulate);
}

__global__ void weight_transpose_kernel(const float* __restrict__ w, int Cout, int T, int Cs, float* __restrict__ wt,
                                        int Os) {
  // wt[c][t][o] = w[o][t][c]; one block per (t, 32x32 tile of (o,c))
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int o = o0 + r, c = c0 + tx;
    tile[r][tx] = (o < Cout && c < Cs) ? w[((int64_t)o * T + t) * Cs + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, o = o0 + tx;
    if (c < Cs && o < Os) wt[((int64_t)c * T + t) * Os + o] = tile[tx][r];
  }
}

// column sums, two stages (deterministic): partial[b][c] over row ranges, then in-order reduce
__global__ void colsum_partial_kernel(const float* __restrict__ dy, int64_t M, int C, int ld, int64_t rows_per_block,
                                      float* __restrict__ part) {
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int64_t r = r_begin; r < r_end; ++r) s += dy[r * ld + c];
    part[(int64_t)blockIdx.x * C + c] = s;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int nb, int C, float* __restrict__ db,
                                    int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < nb; ++b) s += part[(int64_t)b * C + c];
  db[c] = accumulate ? db[c] + s : s;
}

__global__ void relu_bwd_kernel(const float* dy, const float* __restrict__ y, float* out, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    g.x = v.x > 0.f ? g.x : 0.f;
    g.y = v.y > 0.f ? g.y : 0.f;
    g.z = v.z > 0.f ? g.z : 0.f;
    g.w = v.w > 0.f ? g.w : 0.f;
    reinterpret_cast<float4*>(out)[i] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 << 2; i < n; ++i) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// ------------------------------------------------------------------------------------------ C ABI
static int check_pyr(const scan_pyramid_t* d, const char* who) {
  SCAN_CHECK_ARG(d != nullptr, "%s: null pyramid", who);
  SCAN_CHECK_ARG(d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS, "%s: n_levels=%d out of range", who, d->n_levels);
  SCAN_CHECK_ARG(d->n_images >= 1, "%s: n_images=%d", who, d->n_images);
  for (int l = 0; l < d->n_levels; ++l) {
    SCAN_CHECK_ARG(d->h[l] >= 1 && d->w[l] >= 1, "%s: level %d has empty size", who, l);
    SCAN_CHECK_ARG(d->row_off[l + 1] - d->row_off[l] == (int64_t)d->n_images * d->h[l] * d->w[l],
                   "%s: row_off inconsistent at level %d", who, l);
  }
  return 0;
}

template <int MODE>
static int launch_igemm(const float* src, const scan_pyramid_t* sd, int Cs, const float* w, const float* bias,
                        const float* mask, float* dst, const scan_pyramid_t* dd, int Nout, int Ns, int ksize,
                        int stride, int relu, hipStream_t st) {
  const int64_t M = dd->row_off[dd->n_levels];
  const int m_tiles = (int)((M + BM - 1) / BM);
  // few-row problems (the stride-2 P6 / P7 convs: 8 and 2 row tiles) are latency bound on the serial K loop of a
  // handful of workgroups: the 128 x 32 tile gives 4x the workgroups for the same K depth
  const bool small = (int64_t)m_tiles * ((Nout + 127) / 128) < 256;
  if (Nout > 32 && !small) {
    const int n_tiles = (Nout + 127) / 128;
    const size_t sh = (size_t)(2 * BM * LDA + 2 * 128 * LDA) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<MODE, 4>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<MODE, 4>), dim3(m_tiles * n_tiles), dim3(256), sh, st, src, *sd, Cs, w, bias,
                       mask, dst, *dd, Nout, Ns, ksize, stride, relu, m_tiles, n_tiles);
  } else {
    const int n_tiles = (Nout + 31) / 32;
    const size_t sh = (size_t)(2 * BM * LDA + 2 * 32 * LDA) * sizeof(float);
    hipLaunchKernelGGL((conv_igemm_kernel<MODE, 1>), dim3(m_tiles * n_tiles), dim3(256), sh, st, src, *sd, Cs, w, bias,
                       mask, dst, *dd, Nout, Ns, ksize, stride, relu, m_tiles, n_tiles);
  }
  SCAN_LAUNCH_CHECK("conv_igemm");
  return 0;
}

static int check_geometry(const scan_pyramid_t* xd, const scan_pyramid_t* yd, int ksize, int stride, const char* who) {
  SCAN_CHECK_ARG(ksize == 1 || ksize == 3 || ksize == 5 || ksize == 7, "%s: ksize must be 1, 3, 5 or 7, got %d", who, ksize);
  SCAN_CHECK_ARG(stride == 1 || stride == 2, "%s: stride must be 1 or 2, got %d", who, stride);
  SCAN_CHECK_ARG(xd->n_levels == yd->n_levels && xd->n_images == yd->n_images, "%s: pyramid mismatch", who);
  const int pad = ksize / 2;
  for (int l = 0; l < xd->n_levels; ++l) {
    const int eh = (xd->h[l] + 2 * pad - ksize) / stride + 1, ew = (xd->w[l] + 2 * pad - ksize) / stride + 1;
    SCAN_CHECK_ARG(eh == yd->h[l] && ew == yd->w[l], "%s: level %d output size %dx%d, expected %dx%d", who, l, yd->h[l],
                   yd->w[l], eh, ew);
  }
  return 0;
}

extern "C" int scan_conv2d_forward(const float* x, const scan_pyramid_t* xd, int32_t Cin_s, const float* w,
                                   const float* bias, float* y, const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s,
                                   int32_t ksize, int32_t stride, int32_t relu, void* stream) {
  if (check_pyr(xd, "conv2d_forward(x)") || check_pyr(yd, "conv2d_forward(y)")) return -1;
  if (check_geometry(xd, yd, ksize, stride, "conv2d_forward")) return -1;
  SCAN_CHECK_ARG(Cin_s > 0 && Cin_s % 4 == 0, "conv2d_forward: Cin_s=%d must be a positive multiple of 4", Cin_s);
  SCAN_CHECK_ARG(Cout > 0 && Cout_s >= Cout, "conv2d_forward: Cout=%d Cout_s=%d", Cout, Cout_s);
  SCAN_CHECK_ARG(x && w && y, "conv2d_forward: null pointer");
  return launch_igemm<0>(x, xd, Cin_s, w, bias, nullptr, y, yd, Cout, Cout_s, ksize, stride, relu, as_stream(stream));
}

extern "C" int scan_conv2d_dgrad(const float* dy, const scan_pyramid_t* yd, int32_t Cout_s, const float* wt, float* dx,
                                 const scan_pyramid_t* xd, int32_t Cin, int32_t Cin_s, int32_t ksize, int32_t stride,
                                 const float* mask, void* stream) {
  if (check_pyr(xd, "conv2d_dgrad(dx)") || check_pyr(yd, "conv2d_dgrad(dy)")) return -1;
  if (check_geometry(xd, yd, ksize, stride, "conv2d_dgrad")) return -1;
  SCAN_CHECK_ARG(Cout_s > 0 && Cout_s % 4 == 0, "conv2d_dgrad: Cout_s=%d must be a positive multiple of 4", Cout_s);
  SCAN_CHECK_ARG(Cin > 0 && Cin_s >= Cin, "conv2d_dgrad: Cin=%d Cin_s=%d", Cin, Cin_s);
  SCAN_CHECK_ARG(dy && wt && dx, "conv2d_dgrad: null pointer");
  return launch_igemm<1>(dy, yd, Cout_s, wt, nullptr, mask, dx, xd, Cin, Cin_s, ksize, stride, 0, as_stream(stream));
}

static void wgrad_plan(const scan_pyramid_t* yd, int Cin_s, int Cout, int ksize, int* o_tiles, int* c_tiles, int* splits,
                       int* cps) {
  const int64_t M = yd->row_off[yd->n_levels];
  const int64_t chunks = (M + BK - 1) / BK;
  *o_tiles = (Cout + 127) / 128;
  *c_tiles = (Cin_s + 127) / 128;
  const int tiles = *o_tiles * *c_tiles * ksize * ksize;
  int64_t s = 2048 / tiles;
  if (s < 1) s = 1;
  const int64_t smax = (chunks + 7) / 8;
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  *cps = (int)((chunks + s - 1) / s);
  *splits = (int)((chunks + *cps - 1) / *cps);
}

extern "C" int64_t scan_conv2d_wgrad_ws_floats(const scan_pyramid_t* yd, int32_t Cin_s, int32_t Cout, int32_t ksize) {
  int ot, ct, sp, cps;
  wgrad_plan(yd, Cin_s, Cout, ksize, &ot, &ct, &sp, &cps);
  return (int64_t)sp * Cout * ksize * ksize * Cin_s;
}

extern "C" int scan_conv2d_wgrad(const float* x, const scan_pyramid_t* xd, int32_t Cin_s, const float* dy,
                                 const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t ksize, int32_t stride,
                                 float* dw, int32_t accumulate, float* ws, void* stream) {
  if (check_pyr(xd, "conv2d_wgrad(x)") || check_pyr(yd, "conv2d_wgrad(dy)")) return -1;
  if (check_geometry(xd, yd, ksize, stride, "conv2d_wgrad")) return -1;
  SCAN_CHECK_ARG(Cin_s > 0 && Cin_s % 4 == 0, "conv2d_wgrad: Cin_s=%d must be a positive multiple of 4", Cin_s);
  SCAN_CHECK_ARG(Cout > 0 && Cout_s >= Cout, "conv2d_wgrad: Cout=%d Cout_s=%d", Cout, Cout_s);
  SCAN_CHECK_ARG(x && dy && dw && ws, "conv2d_wgrad: null pointer");
  int ot, ct, sp, cps;
  wgrad_plan(yd, Cin_s, Cout, ksize, &ot, &ct, &sp, &cps);
  hipStream_t st = as_stream(stream);
  const size_t sh = (size_t)4 * BK * LDW * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)sh);
    attr_done = true;
  }
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(ot * ct * ksize * ksize, sp), dim3(256), sh, st, x, *xd, Cin_s, dy, *yd,
                     Cout, Cout_s, ksize, stride, ws, ot, ct, cps);
  SCAN_LAUNCH_CHECK("conv_wgrad");
  const int64_t n = (int64_t)Cout * ksize * ksize * Cin_s;  // multiple of 4 because Cin_s is
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, st, ws, sp, n, dw, accumulate);
  SCAN_LAUNCH_CHECK("slab_reduce");
  return 0;
}

extern "C" int scan_weight_transpose(const float* w, int32_t Cout, int32_t T, int32_t Cin_s, float* wt, int32_t Cout_s,
                                     void* stream) {
  SCAN_CHECK_ARG(w && wt && Cout > 0 && T > 0 && Cin_s > 0 && Cout_s >= Cout, "weight_transpose: bad arguments");
  dim3 grid((Cin_s + 31) / 32, (Cout_s + 31) / 32, T);
  hipLaunchKernelGGL(weight_transpose_kernel, grid, dim3(256), 0, as_stream(stream), w, Cout, T, Cin_s, wt, Cout_s);
  SCAN_LAUNCH_CHECK("weight_transpose");
  return 0;
}

static int colsum_blocks(int64_t M) {
  int64_t b = (M + 255) / 256;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}
extern "C" int64_t scan_colsum_ws_floats(int64_t M, int32_t C) { return (int64_t)colsum_blocks(M) * C; }

extern "C" int scan_colsum(const float* dy, int64_t M, int32_t C, int32_t ld, float* db, int32_t accumulate, float* ws,
                           void* stream) {
  SCAN_CHECK_ARG(dy && db && ws && M >= 0 && C > 0 && ld >= C, "colsum: bad arguments");
  const int nb = colsum_blocks(M);
  const int64_t rpb = (M + nb - 1) / nb;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, dy, M, C, ld, rpb > 0 ? rpb : 1, ws);
  SCAN_LAUNCH_CHECK("colsum_partial");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, nb, C, db, accumulate);
  SCAN_LAUNCH_CHECK("colsum_final");
  return 0;
}

extern "C" int scan_relu_backward(const float* dy, const float* y, float* out, int64_t n, void* stream) {
  SCAN_CHECK_ARG(dy && y && out && n >= 0, "relu_backward: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, as_stream(stream), dy, y, out, n);
  SCAN_LAUNCH_CHECK("relu_bwd");
  return 0;
}
