// First-layer convolutions (3 input channels, stored as 4): the ResNet stem 7x7 / stride 2 (reference
// backbone/resnet.py:316-336) and VGG conv1_1 3x3 / stride 1 (backbone/mmdetection/vgg.py:8-33), forward only (both
// sit in frozen stages).  With Cin = 4 the generic kernels spend one 32-channel K chunk per tap on 4 live channels;
// here K = taps x 4 is one dense GEMM dimension (7x7: 196 -> 13 MFMA steps of 16, 3x3: 36 -> 3 steps):
//   workgroup   256 threads = 4 waves; output tile 8 x 32 pixels x 64 channels, wave w = pixel rows 2w, 2w+1
//   A operand   the (8-1)*S+KS x (32-1)*S+KS input patch is read once (one float4 per pixel), split fp32 -> bf16 hi/lo
//               and kept in LDS as [pixel][4]; an MFMA lane's 8 consecutive k are two taps x 4 channels = two 8-byte
//               LDS reads at (tap -> patch offset) addresses: im2col never exists in memory
//   B operand   the whole weight matrix [64][K] is split once per workgroup into LDS (row pitch K+8: 16-byte reads of
//               32 consecutive rows fall on distinct banks); workgroups are persistent over tiles to amortise it
//   product     split operands as in conv_split.h: NPC = 2 pieces ("bf16x3") or 3 ("bf16x6"), fp32 accumulate
// HBM-bound by construction: 7x7/2 at 2 x 1024x2048 reads 67 MB and writes 268 MB; 3x3/1 writes 1.07 GB.
#include "conv_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SC_TH 8
#define SC_TW 32

template <int NPC, int KS, int S>
__global__ __launch_bounds__(256) void conv_smallcin_kernel(const float* __restrict__ x, int N, int H, int W,
                                                            const float* __restrict__ wgt,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int Ho, int Wo, int Nout, int Ns, int relu, int tiles_x,
                                                            int tiles_y, int total_tiles) {
  constexpr int T = KS * KS, K = T * 4, KSTEPS = (K + 15) / 16, KP = KSTEPS * 16, BROW = KP + 8;
  constexpr int PH = (SC_TH - 1) * S + KS, PWD = (SC_TW - 1) * S + KS, NP = PH * PWD, PAD = KS / 2;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* Bs = reinterpret_cast<__bf16*>(smem_raw);  // [NPC piece][64][BROW]
  __bf16* As = Bs + NPC * 64 * BROW;                 // [NPC piece][NP][4]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;

  for (int i = tid; i < 64 * KP; i += 256) {
    const int o = i / KP, k = i - o * KP;
    const float v = (k < K && o < Nout) ? wgt[(int64_t)o * K + k] : 0.f;
    __bf16 q[NPC];
    split1_np<NPC>(v, q);
#pragma unroll
    for (int p = 0; p < NPC; ++p) Bs[(p * 64 + o) * BROW + k] = q[p];
  }

  // per-step tap offsets of this lane's two taps (k = 16 step + 8 lh -> taps 4 step + 2 lh, +1), clamped into the
  // kernel window for the zero-padded tail of K (the weights there are zero)
  int toff[KSTEPS][2];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int t = 4 * s + 2 * lh + e;
      t = t < T ? t : T - 1;
      toff[s][e] = ((t / KS) * PWD + (t % KS)) * 4;
    }
  float bv[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) bv[tn] = (bias != nullptr && tn * 32 + lr < Nout) ? bias[tn * 32 + lr] : 0.f;

  // the next tile's patch is fetched into registers while the current one computes (one LDS stage)
  constexpr int NPR = (NP + 255) / 256;
  float4 pre[NPR];
  auto fetch = [&](int tile) {
    const int per_img = tiles_x * tiles_y;
    const int img = tile / per_img;
    const int t2 = tile - img * per_img;
    const int iy0 = (t2 / tiles_x) * SC_TH * S - PAD, ix0 = (t2 % tiles_x) * SC_TW * S - PAD;
#pragma unroll
    for (int i = 0; i < NPR; ++i) {
      const int p = tid + 256 * i;
      const int py = p / PWD, px = p - py * PWD;
      const int iy = iy0 + py, ix = ix0 + px;
      pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < NP && iy >= 0 && iy < H && ix >= 0 && ix < W)
        pre[i] = *reinterpret_cast<const float4*>(x + (((int64_t)img * H + iy) * W + ix) * 4);
    }
  };
  if ((int)blockIdx.x < total_tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int per_img = tiles_x * tiles_y;
    const int img = tile / per_img;
    const int t2 = tile - img * per_img;
    const int y0 = (t2 / tiles_x) * SC_TH, x0 = (t2 % tiles_x) * SC_TW;
    __syncthreads();  // the previous tile's fragment reads (and, first time, the weight staging) are complete
#pragma unroll
    for (int i = 0; i < NPR; ++i) {
      const int p = tid + 256 * i;
      if (p < NP) {
        bf16x4 pc[NPC];
        split4_np<NPC>(pre[i], pc);
#pragma unroll
        for (int q = 0; q < NPC; ++q) *reinterpret_cast<bf16x4*>(As + (q * NP + p) * 4) = pc[q];
      }
    }
    if (tile + (int)gridDim.x < total_tiles) fetch(tile + gridDim.x);
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    int abase[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) abase[tm] = ((2 * wid + tm) * S * PWD + lr * S) * 4;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      bf16x8 af[NPC][2], bf[NPC][2];
#pragma unroll
      for (int q = 0; q < NPC; ++q)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const bf16x4 h0 = *reinterpret_cast<const bf16x4*>(As + q * NP * 4 + abase[tm] + toff[s][0]);
          const bf16x4 h1 = *reinterpret_cast<const bf16x4*>(As + q * NP * 4 + abase[tm] + toff[s][1]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            af[q][tm][e] = h0[e];
            af[q][tm][4 + e] = h1[e];
          }
        }
#pragma unroll
      for (int q = 0; q < NPC; ++q)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          bf[q][tn] = *reinterpret_cast<const bf16x8*>(Bs + (q * 64 + tn * 32 + lr) * BROW + 16 * s + 8 * lh);
      // piece products, smallest first; within a magnitude class the pixel piece index descends (lo * hi, hi * lo, hi * hi)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int d = NPC - 1; d >= 0; --d)
#pragma unroll
            for (int i = d; i >= 0; --i)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][tm], bf[d - i][tn], acc[tm][tn], 0, 0, 0);
    }
    // C/D map of 32x32: col = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel)
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int oy = y0 + 2 * wid + tm;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int o = tn * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ox = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (oy < Ho && ox < Wo && o < Nout) {
            float v = acc[tm][tn][r] + bv[tn];
            if (relu) v = fmaxf(v, 0.f);
            y[(((int64_t)img * Ho + oy) * Wo + ox) * Ns + o] = v;
          }
        }
      }
    }
  }
}

template <int NPC, int KS, int S>
static int launch_smallcin(const float* x, int N, int H, int W, const float* w, const float* bias, float* y, int Nout,
                           int Ns, int relu, hipStream_t st) {
  constexpr int T = KS * KS, K = T * 4, KP = ((K + 15) / 16) * 16, BROW = KP + 8;
  constexpr int NP = ((SC_TH - 1) * S + KS) * ((SC_TW - 1) * S + KS);
  const int Ho = (H + 2 * (KS / 2) - KS) / S + 1, Wo = (W + 2 * (KS / 2) - KS) / S + 1;
  const int tiles_x = (Wo + SC_TW - 1) / SC_TW, tiles_y = (Ho + SC_TH - 1) / SC_TH;
  const int total = N * tiles_x * tiles_y;
  const size_t sh = (size_t)(NPC * 64 * BROW + NPC * NP * 4) * sizeof(__bf16);
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_smallcin_kernel<NPC, KS, S>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    done = true;
  }
  const int grid = total < 1536 ? total : 1536;  // persistent: 6 workgroups' worth of tiles per CU
  hipLaunchKernelGGL((conv_smallcin_kernel<NPC, KS, S>), dim3(grid), dim3(256), sh, st, x, N, H, W, w, bias, y, Ho, Wo, Nout,
                     Ns, relu, tiles_x, tiles_y, total);
  SCAN_LAUNCH_CHECK("conv_smallcin");
  return 0;
}

template <int NPC>
static int smallcin_entry(const float* x, int32_t N, int32_t H, int32_t W, const float* w, const float* bias, float* y,
                          int32_t Cout, int32_t Cout_s, int32_t ksize, int32_t stride, int32_t relu, void* stream) {
  SCAN_CHECK_ARG(x && w && y && N > 0 && H > 0 && W > 0, "conv_smallcin: bad arguments");
  SCAN_CHECK_ARG(Cout > 0 && Cout <= 64 && Cout_s >= Cout, "conv_smallcin: Cout=%d must be in 1..64 (Cout_s=%d)", Cout,
                 Cout_s);
  hipStream_t st = as_stream(stream);
  if (ksize == 3 && stride == 1) return launch_smallcin<NPC, 3, 1>(x, N, H, W, w, bias, y, Cout, Cout_s, relu, st);
  if (ksize == 7 && stride == 2) return launch_smallcin<NPC, 7, 2>(x, N, H, W, w, bias, y, Cout, Cout_s, relu, st);
  scan_set_error("conv_smallcin: only 3x3/1 and 7x7/2 are built (got %dx%d/%d)", ksize, ksize, stride);
  return -1;
}

extern "C" int scan_conv_smallcin_bf16x3(const float* x, int32_t N, int32_t H, int32_t W, const float* w,
                                         const float* bias, float* y, int32_t Cout, int32_t Cout_s, int32_t ksize,
                                         int32_t stride, int32_t relu, void* stream) {
  return smallcin_entry<2>(x, N, H, W, w, bias, y, Cout, Cout_s, ksize, stride, relu, stream);
}
extern "C" int scan_conv_smallcin_bf16x6(const float* x, int32_t N, int32_t H, int32_t W, const float* w,
                                         const float* bias, float* y, int32_t Cout, int32_t Cout_s, int32_t ksize,
                                         int32_t stride, int32_t relu, void* stream) {
  return smallcin_entry<3>(x, N, H, W, w, bias, y, Cout, Cout_s, ksize, stride, relu, stream);
}
