// Conditioned-kernel generator of the graph middle head in two launches (forward, backward):
//   paradigm P [K, 256, T]  ->  nn.RNN(256, 512, num_layers=2, tanh), h0 = 0, over the T paradigm slots (batch = the K
//   classes)  ->  Conv2d(512, 256, (T, 1)) over the [K, 512, T, 1] sequence = one linear map over (channel, slot)  ->
//   kernels [K, 256]                                          (reference rpn/fcos/condgraph.py:313-319, 237-246)
// Written with torch ops this is T x 2 layers x (2 linear + add + tanh) + stack / permute glue forward and about twice that
// backward: ~120 launches in a strictly dependent chain.  Here: one launch per link of the chain -- 7
// forward (T x 2 recurrent steps + the output map), 8 backward (output map, T x 2 steps, all weight gradients) -- each spread
// over 8-32 workgroups.  (A first version walked the whole chain in ONE workgroup: 350 us forward, bound by what a single
// CU can keep in flight of the 5 MB of weights -- slower inside the step than the torch launches it replaced.)
//   forward link   out[k][j] = act(b[j] + sum_i inA[k][i] WA[j][i] + sum_i inB[k][i] WB[j][i]): a workgroup per 16 units j,
//                  16 lanes per unit split the input dimension (16-byte reads of W[j][:]), butterfly over the 16 lanes.
//   backward link  dA[k][i] = (pre[k][i] + sum_j dX[k][j] WX[j][i] + sum_j dY[k][j] WY[j][i]) (1 - h[k][i]^2): a workgroup
//                  per 64 units i (W[j][i] coalesced over the lanes), its four waves split j, partial sums meet in LDS.
//   weight grads   dW[j][i] = sum_t sum_k dA_t[k][j] in_t[k][i], a thread per element; biases: sum_t sum_k dA_t[k][j].
// fp32 FMAs in a fixed order (no atomics): deterministic; differs from the GEMM library's summation order by rounding only.
#include "common.h"

#define CR_I 256   // input size (paradigm channels)
#define CR_H 512   // hidden size
#define CR_O 256   // generated kernel length
#define CR_KMAX 9  // classes incl. background (C2F: 9, Sim10k / KITTI: 2)
#define CR_TMAX 3  // paradigm slots (MODEL.MIDDLE_HEAD.PROTO_ITER)

struct CondRnnW {
  const float* wih0;  // [512, 256]
  const float* whh0;  // [512, 512]
  const float* bih0;  // [512]
  const float* bhh0;
  const float* wih1;  // [512, 512]
  const float* whh1;
  const float* bih1;
  const float* bhh1;
  const float* wc;    // [256, 512, T]  (Conv2d(512, 256, (T, 1)).weight)
  const float* bc;    // [256]
};

struct CondRnnG {  // gradients, same shapes
  float* wih0;
  float* whh0;
  float* bih0;
  float* bhh0;
  float* wih1;
  float* whh1;
  float* bih1;
  float* bhh1;
  float* wc;
  float* bc;
};


// ---- forward link
// in rows are staged in LDS: [KMAX][NIA] then [KMAX][NIB] (rows k >= K zero).  interleave > 0: inA is h1 [T][K][512] read as
// rows of 512 * T values in (c, t) order (the output map's view of the sequence), interleave = T.
__global__ __launch_bounds__(256) void cr_link_fwd_kernel(const float* __restrict__ inA, int NIA, const float* __restrict__ WA,
                                                          const float* __restrict__ inB, int NIB, const float* __restrict__ WB,
                                                          const float* __restrict__ b1, const float* __restrict__ b2, int K,
                                                          int NJ, int interleave, int act, float* __restrict__ out) {
  extern __shared__ __align__(16) float sm[];
  float* a = sm;
  float* b = sm + CR_KMAX * NIA;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int e = tid; e < CR_KMAX * (NIA + NIB); e += 256) sm[e] = 0.f;
  __syncthreads();
  if (interleave > 0) {
    const int T = interleave;
    for (int e = tid; e < T * K * CR_H; e += 256) {  // h1[t][k][c] -> a[k][c * T + t]
      const int c = e % CR_H, tk = e / CR_H;
      a[(tk % K) * NIA + c * T + tk / K] = inA[e];
    }
  } else {
    for (int e = tid; e < K * NIA; e += 256) a[e] = inA[e];
  }
  for (int e = tid; e < K * NIB; e += 256) b[e] = inB[e];
  __syncthreads();
  const int j = blockIdx.x * 16 + wid * 4 + (lane >> 4), g = lane & 15;
  float acc[CR_KMAX];
#pragma unroll
  for (int k = 0; k < CR_KMAX; ++k) acc[k] = 0.f;
  {
    const float* wrow = WA + (size_t)j * NIA;
    for (int i = g * 4; i < NIA; i += 64) {
      const float4 w = *reinterpret_cast<const float4*>(wrow + i);
#pragma unroll
      for (int k = 0; k < CR_KMAX; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(a + k * NIA + i);
        acc[k] = fmaf(w.x, v.x, fmaf(w.y, v.y, fmaf(w.z, v.z, fmaf(w.w, v.w, acc[k]))));
      }
    }
  }
  if (NIB > 0) {
    const float* wrow = WB + (size_t)j * NIB;
    for (int i = g * 4; i < NIB; i += 64) {
      const float4 w = *reinterpret_cast<const float4*>(wrow + i);
#pragma unroll
      for (int k = 0; k < CR_KMAX; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(b + k * NIB + i);
        acc[k] = fmaf(w.x, v.x, fmaf(w.y, v.y, fmaf(w.z, v.z, fmaf(w.w, v.w, acc[k]))));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CR_KMAX; ++k)
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
  if (g < K) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < CR_KMAX; ++k) v = (g == k) ? acc[k] : v;
    v += b1[j] + (b2 ? b2[j] : 0.f);
    out[(size_t)g * NJ + j] = act ? tanhf(v) : v;
  }
}

// ---- backward link.  grid NI / 64, 256 threads.  pre (may be null): pre[k * pre_ks + i * pre_is]; WX [NJX][NI], WY [NJY][NI];
// h (may be null: no activation behind this link): h[k * NI + i]; out[k * NI + i]
__global__ __launch_bounds__(256) void cr_link_bwd_kernel(const float* __restrict__ pre, int pre_ks, int pre_is,
                                                          const float* __restrict__ dX, int NJX, const float* __restrict__ WX,
                                                          const float* __restrict__ dY, int NJY, const float* __restrict__ WY,
                                                          const float* __restrict__ h, int K, int NI, float* __restrict__ out) {
  extern __shared__ __align__(16) float sm[];
  float* x = sm;                       // [KMAX][NJX]
  float* y = sm + CR_KMAX * NJX;       // [KMAX][NJY]
  float* red = y + CR_KMAX * NJY;      // [4][KMAX][64]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int e = tid; e < CR_KMAX * (NJX + NJY); e += 256) sm[e] = 0.f;
  __syncthreads();
  for (int e = tid; e < K * NJX; e += 256) x[e] = dX[e];
  for (int e = tid; e < K * NJY; e += 256) y[e] = dY[e];
  __syncthreads();
  const int i = blockIdx.x * 64 + lane;
  float acc[CR_KMAX];
#pragma unroll
  for (int k = 0; k < CR_KMAX; ++k) acc[k] = 0.f;
  for (int j = wid; j < NJX; j += 4) {
    const float wv = WX[(size_t)j * NI + i];
#pragma unroll
    for (int k = 0; k < CR_KMAX; ++k) acc[k] = fmaf(wv, x[k * NJX + j], acc[k]);
  }
  for (int j = wid; j < NJY; j += 4) {
    const float wv = WY[(size_t)j * NI + i];
#pragma unroll
    for (int k = 0; k < CR_KMAX; ++k) acc[k] = fmaf(wv, y[k * NJY + j], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < CR_KMAX; ++k) red[(wid * CR_KMAX + k) * 64 + lane] = acc[k];
  __syncthreads();
  for (int e = tid; e < K * 64; e += 256) {
    const int k = e >> 6, l = e & 63, ii = blockIdx.x * 64 + l;
    float v = (red[(0 * CR_KMAX + k) * 64 + l] + red[(1 * CR_KMAX + k) * 64 + l]) +
              (red[(2 * CR_KMAX + k) * 64 + l] + red[(3 * CR_KMAX + k) * 64 + l]);
    if (pre) v += pre[(size_t)k * pre_ks + (size_t)ii * pre_is];
    if (h) {
      const float hv = h[(size_t)k * NI + ii];
      v *= 1.f - hv * hv;
    }
    out[(size_t)k * NI + ii] = v;
  }
}

// ---- weight gradients.  grid (64, 6), 256 threads: blockIdx.y = 0 wih0 (x_t, dA0_t) | 1 whh0 (h0_{t-1}, dA0_t) | 2 wih1
// (h0_t, dA1_t) | 3 whh1 (h1_{t-1}, dA1_t) | 4 wc: dwc[o][c][t] = sum_k dker[k][o] h1_t[k][c] | 5 the biases.
// x [T][K][256], h0 / h1 / dA0 / dA1 [T][K][512].  The input rows of all slots sit in LDS, a row's coefficients are uniform.
__global__ __launch_bounds__(256) void cr_wgrad_kernel(const float* __restrict__ x, int K, int T, const float* __restrict__ h0,
                                                       const float* __restrict__ h1, const float* __restrict__ dker,
                                                       const float* __restrict__ dA0, const float* __restrict__ dA1, CondRnnG g) {
  extern __shared__ __align__(16) float in[];  // [T][KMAX][NI]
  const int tid = threadIdx.x, mat = blockIdx.y;
  if (mat == 5) {  // bias gradients: 512 units per layer, 8 per workgroup
    if (tid < 16) {
      const int j = blockIdx.x * 8 + (tid & 7);
      const float* dA = (tid >> 3) ? dA1 : dA0;
      float s = 0.f;
      for (int tk = 0; tk < T * K; ++tk) s += dA[(size_t)tk * CR_H + j];
      if (tid >> 3) {
        g.bih1[j] = s;
        g.bhh1[j] = s;
      } else {
        g.bih0[j] = s;
        g.bhh0[j] = s;
      }
    }
    if (blockIdx.x < 32 && tid >= 64 && tid < 72) {  // dbc: 256 values, 8 per workgroup of the first 32
      const int o = blockIdx.x * 8 + (tid - 64);
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += dker[k * CR_O + o];
      g.bc[o] = s;
    }
    return;
  }
  const int NI = mat == 0 ? CR_I : CR_H;
  const int shift = (mat == 1 || mat == 3) ? 1 : 0;  // recurrent weights pair dA_t with the state of slot t - 1
  for (int e = tid; e < CR_TMAX * CR_KMAX * NI; e += 256) in[e] = 0.f;
  __syncthreads();
  {
    const float* src = mat == 0 ? x : (mat == 1 || mat == 2) ? h0 : h1;
    for (int e = tid; e < T * K * NI; e += 256) {
      const int i = e % NI, tk = e / NI;
      in[((tk / K) * CR_KMAX + (tk % K)) * NI + i] = src[e];
    }
  }
  __syncthreads();
  if (mat == 4) {
    for (int r = 0; r < 4; ++r) {
      const int o = blockIdx.x * 4 + r;
      float cf[CR_KMAX];
#pragma unroll
      for (int k = 0; k < CR_KMAX; ++k) cf[k] = k < K ? dker[k * CR_O + o] : 0.f;
      for (int e = tid; e < CR_H * T; e += 256) {
        const int t = e % T, c = e / T;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < CR_KMAX; ++k) s = fmaf(cf[k], in[(t * CR_KMAX + k) * CR_H + c], s);
        g.wc[(size_t)o * CR_H * T + e] = s;
      }
    }
    return;
  }
  const float* dA = mat >= 2 ? dA1 : dA0;
  float* out = mat == 0 ? g.wih0 : mat == 1 ? g.whh0 : mat == 2 ? g.wih1 : g.whh1;
  for (int r = 0; r < 8; ++r) {
    const int j = blockIdx.x * 8 + r;
    float cf[CR_TMAX][CR_KMAX];
#pragma unroll
    for (int t = 0; t < CR_TMAX; ++t)
#pragma unroll
      for (int k = 0; k < CR_KMAX; ++k) cf[t][k] = (t < T && t >= shift && k < K) ? dA[((size_t)t * K + k) * CR_H + j] : 0.f;
    for (int i = tid; i < NI; i += 256) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < CR_TMAX; ++t)
#pragma unroll
        for (int k = 0; k < CR_KMAX; ++k)
          if (t >= shift) s = fmaf(cf[t][k], in[((t - shift) * CR_KMAX + k) * NI + i], s);
      out[(size_t)j * NI + i] = s;
    }
  }
}

static int cr_check(int K, int T, const char* who) {
  SCAN_CHECK_ARG(K >= 1 && K <= CR_KMAX && T >= 1 && T <= CR_TMAX, "%s: K=%d (<= %d), T=%d (<= %d)", who, K, CR_KMAX, T, CR_TMAX);
  return 0;
}

static void cr_attr() {
  static bool done = false;
  if (done) return;
  hipFuncSetAttribute(reinterpret_cast<const void*>(cr_link_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)(sizeof(float) * CR_KMAX * (CR_H * CR_TMAX + CR_H)));
  hipFuncSetAttribute(reinterpret_cast<const void*>(cr_link_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)(sizeof(float) * (CR_KMAX * 2 * CR_H + 4 * CR_KMAX * 64)));
  hipFuncSetAttribute(reinterpret_cast<const void*>(cr_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                      (int)(sizeof(float) * CR_TMAX * CR_KMAX * CR_H));
  done = true;
}

// x: the paradigm as [T, K, 256] (slot-major); weights: ten device pointers in CondRnnW order (wih0, whh0, bih0, bhh0, wih1,
// whh1, bih1, bhh1, wc, bc), all fp32 contiguous
extern "C" int scan_cond_rnn_forward(const float* x, int32_t K, int32_t T, const float* const* weights, float* h0, float* h1,
                                     float* kernels, void* stream) {
  if (cr_check(K, T, "cond_rnn_forward")) return -1;
  SCAN_CHECK_ARG(x && weights && h0 && h1 && kernels, "cond_rnn_forward: null pointer");
  const CondRnnW w = {weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6], weights[7], weights[8], weights[9]};
  cr_attr();
  hipStream_t st = as_stream(stream);
  const size_t KH = (size_t)K * CR_H;
  for (int layer = 0; layer < 2; ++layer) {
    const float* wih = layer ? w.wih1 : w.wih0;
    const float* whh = layer ? w.whh1 : w.whh0;
    const float* bih = layer ? w.bih1 : w.bih0;
    const float* bhh = layer ? w.bhh1 : w.bhh0;
    float* hs = layer ? h1 : h0;
    for (int t = 0; t < T; ++t) {
      const float* inA = layer ? h0 + t * KH : x + (size_t)t * K * CR_I;
      const int NIA = layer ? CR_H : CR_I, NIB = t > 0 ? CR_H : 0;
      hipLaunchKernelGGL(cr_link_fwd_kernel, dim3(CR_H / 16), dim3(256), sizeof(float) * CR_KMAX * (NIA + NIB), st, inA, NIA,
                         wih, t > 0 ? hs + (t - 1) * KH : nullptr, NIB, whh, bih, bhh, K, CR_H, 0, 1, hs + t * KH);
    }
  }
  hipLaunchKernelGGL(cr_link_fwd_kernel, dim3(CR_O / 16), dim3(256), sizeof(float) * CR_KMAX * CR_H * T, st, h1, CR_H * T, w.wc,
                     nullptr, 0, nullptr, w.bc, nullptr, K, CR_O, T, 0, kernels);
  SCAN_LAUNCH_CHECK("cond_rnn_forward");
  return 0;
}

// ws: dS [K][512 * T] | dA0 [T][K][512] | dA1 [T][K][512]
extern "C" int64_t scan_cond_rnn_ws_floats(void) { return 3 * CR_TMAX * CR_KMAX * CR_H; }

// grads: ten device pointers in the same order, every one overwritten; ws: scan_cond_rnn_ws_floats() floats
extern "C" int scan_cond_rnn_backward(const float* x, int32_t K, int32_t T, const float* const* weights, const float* h0,
                                      const float* h1, const float* dkernels, float* const* grads, float* ws, void* stream) {
  if (cr_check(K, T, "cond_rnn_backward")) return -1;
  SCAN_CHECK_ARG(x && weights && h0 && h1 && dkernels && grads && ws, "cond_rnn_backward: null pointer");
  const CondRnnW w = {weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6], weights[7], weights[8], weights[9]};
  const CondRnnG g = {grads[0], grads[1], grads[2], grads[3], grads[4], grads[5], grads[6], grads[7], grads[8], grads[9]};
  cr_attr();
  hipStream_t st = as_stream(stream);
  const size_t KH = (size_t)K * CR_H;
  float* dS = ws;
  float* dA0 = dS + (size_t)CR_TMAX * CR_KMAX * CR_H;
  float* dA1 = dA0 + (size_t)CR_TMAX * CR_KMAX * CR_H;
  const size_t shb = sizeof(float) * (CR_KMAX * 2 * CR_H + 4 * CR_KMAX * 64);
  // the output map: dS[k][(c, t)] = sum_o dker[k][o] wc[o][(c, t)]
  hipLaunchKernelGGL(cr_link_bwd_kernel, dim3(CR_H * T / 64), dim3(256), shb, st, nullptr, 0, 0, dkernels, CR_O, w.wc, nullptr, 0,
                     nullptr, nullptr, K, CR_H * T, dS);
  for (int t = T - 1; t >= 0; --t) {
    const bool more = t + 1 < T;
    // dA1_t = (dS_t + dA1_{t+1} . whh1) (1 - h1_t^2)
    hipLaunchKernelGGL(cr_link_bwd_kernel, dim3(CR_H / 64), dim3(256), shb, st, dS + t, CR_H * T, T, more ? dA1 + (t + 1) * KH : nullptr,
                       more ? CR_H : 0, w.whh1, nullptr, 0, nullptr, h1 + t * KH, K, CR_H, dA1 + t * KH);
    // dA0_t = (dA1_t . wih1 + dA0_{t+1} . whh0) (1 - h0_t^2)
    hipLaunchKernelGGL(cr_link_bwd_kernel, dim3(CR_H / 64), dim3(256), shb, st, nullptr, 0, 0, dA1 + t * KH, CR_H, w.wih1,
                       more ? dA0 + (t + 1) * KH : nullptr, more ? CR_H : 0, w.whh0, h0 + t * KH, K, CR_H, dA0 + t * KH);
  }
  hipLaunchKernelGGL(cr_wgrad_kernel, dim3(64, 6), dim3(256), sizeof(float) * CR_TMAX * CR_KMAX * CR_H, st, x, K, T, h0, h1,
                     dkernels, dA0, dA1, g);
  SCAN_LAUNCH_CHECK("cond_rnn_backward");
  return 0;
}
