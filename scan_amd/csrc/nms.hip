// Greedy NMS / label-aware NMS entirely on the device for gfx950.
// Replaces _C.nms / _C.ml_nms (reference csrc/nms.h:10-28, csrc/cpu/nms_cpu.cpp:5-65,
// csrc/cuda/nms.cu:23-131, csrc/ml_nms.h:10-27, csrc/cuda/ml_nms.cu:13-136).
//
// Three launches on the caller's stream, no host round trip (the reference copies the
// n x n/64 mask to the host and scans there, nms.cu:100-123):
//   1. one-workgroup LDS bitonic sort of (score desc, index asc) keys  -> order, sorted boxes
//   2. 64x64 IoU tiles: one wavefront per tile, one u64 suppression word per (row, col-tile)
//   3. one-workgroup scan: wave 0 resolves each 64-box diagonal tile with scalar
//      readlane chains and ORs the kept rows' words into the running mask; then the whole
//      workgroup compacts the kept ORIGINAL indices in ascending order.
//
// n <= SCAN_NMS_PANEL (8192) takes the single-workgroup kernels above; a larger n (up to SCAN_NMS_MAX) takes the panel
// path at the end of this file: the sort as 8192-key LDS blocks + global compare-exchange steps, the scan with its chain
// state for one 8192-candidate panel at a time in LDS.  Same keep list.
//
// Compiled with -ffp-contract=off: every product/sum is a separately rounded fp32 op in
// the oracle's order, so keep indices are bit-exact against oracle/scan_oracle.c.
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ unsigned int f2sortable_desc(float f) {
  unsigned int u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending order-preserving
  return ~u;                                       // descending
}

// n_pad = power of two >= n, <= 8192.  1024 threads.
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ dets,
                                                        const float* __restrict__ scores,
                                                        const float* __restrict__ labels, int n, int n_pad,
                                                        int* __restrict__ order, float4* __restrict__ boxes_sorted,
                                                        float* __restrict__ labels_sorted,
                                                        float* __restrict__ areas_sorted) {
  extern __shared__ u64 keys[];
  for (int i = threadIdx.x; i < n_pad; i += blockDim.x)
    keys[i] = (i < n) ? (((u64)f2sortable_desc(scores[i]) << 32) | (unsigned int)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= n_pad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int o = (int)(keys[i] & 0xffffffffu);
    order[i] = o;
    const float4 b = reinterpret_cast<const float4*>(dets)[o];
    boxes_sorted[i] = b;
    areas_sorted[i] = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
    labels_sorted[i] = labels ? labels[o] : 0.f;
  }
}

// grid (nb, nb), 64 threads: block (cb, rb) computes mask[row][cb] for rows of tile rb
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes,
                                                      const float* __restrict__ areas,
                                                      const float* __restrict__ labels, int n, float thr, int rule_ge,
                                                      int use_labels, u64* __restrict__ mask, int nb) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (cb < rb) return;  // only later (lower-score) boxes can be suppressed
  __shared__ float4 cbox[64];
  __shared__ float carea[64];
  __shared__ float clab[64];
  const int t = threadIdx.x;
  const int cj = cb * 64 + t;
  if (cj < n) {
    cbox[t] = boxes[cj];
    carea[t] = areas[cj];
    clab[t] = labels[cj];
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= n) return;
  const float4 a = boxes[i];
  const float iarea = areas[i];
  const float ilab = labels[i];
  const int csize = (n - cb * 64) < 64 ? (n - cb * 64) : 64;
  u64 bits = 0;
  const int start = (rb == cb) ? t + 1 : 0;
  for (int j = start; j < csize; ++j) {
    if (use_labels && ilab != clab[j]) continue;
    const float4 b = cbox[j];
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.0f, xx2 - xx1 + 1.0f);
    const float h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
    const float inter = w * h;
    const float ovr = inter / (iarea + carea[j] - inter);
    const bool sup = rule_ge ? (ovr >= thr) : (ovr > thr);
    if (sup) bits |= 1ull << j;
  }
  mask[(int64_t)i * nb + cb] = bits;
}

__device__ __forceinline__ u64 readlane64(u64 v, int lane_const) {
  const unsigned int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffu), lane_const);
  const unsigned int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane_const);
  return ((u64)hi << 32) | lo;
}

// one workgroup of 1024 threads
__global__ __launch_bounds__(1024) void nms_scan_kernel(const u64* __restrict__ mask, const int* __restrict__ order,
                                                        int n, int nb, int64_t* __restrict__ keep_out,
                                                        int* __restrict__ num_keep) {
  __shared__ u64 remv[SCAN_NMS_PANEL / 64];
  __shared__ u64 kept_s[SCAN_NMS_PANEL / 64];     // kept candidates of every chunk, by SORTED position
  __shared__ unsigned char flag[SCAN_NMS_PANEL];  // by ORIGINAL index
  __shared__ int wave_tot[16];
  extern __shared__ u64 diag_s[];               // [nb * 64] the diagonal word of every sorted row
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < SCAN_NMS_PANEL / 64; i += blockDim.x) remv[i] = 0;
  for (int i = tid; i < SCAN_NMS_PANEL; i += blockDim.x) flag[i] = 0;
  // the chain below must not wait on vector memory (a wait there drains every load in flight, the look-ahead included):
  // the diagonal words go to LDS up front, the original indices are looked up after the chain
  for (int i = tid; i < nb * 64; i += blockDim.x) diag_s[i] = i < n ? mask[(int64_t)i * nb + (i >> 6)] : 0ull;
  __syncthreads();
  // The greedy pass is a chain over the 64-candidate chunks: wave 0 settles chunk c from the suppression bits collected so
  // far (64 register steps on the chunk's diagonal words), then ALL 16 waves OR the kept rows' words into the running
  // mask of the later chunks -- four rows per wave, fetched ahead of the decision.  (Until round 3 wave 0 did that
  // gather alone, one dependent load per kept row: 1.2 ms for 3.6 k candidates, a sixth of an inference batch.)
  __shared__ u64 keep_s;
  // a wave's four rows of chunk c, words c + 1 + lane (+ 64): loaded BEFORE the chunk is settled (which rows are kept
  // only decides whether a word is used), so the loads run under wave 0's 64 register steps
  u64 pre[2][4];
  auto prefetch = [&](int c) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int w = c + 1 + 64 * g + lane, row = c * 64 + wid * 4 + q;
        pre[g][q] = (w < nb && row < n) ? mask[(int64_t)row * nb + w] : 0ull;
      }
  };
  prefetch(0);
  for (int c = 0; c < nb; ++c) {
    if (wid == 0) {
      const u64 diag = diag_s[c * 64 + lane];
      // the running word and the kept set live in scalar registers: 64 dependent steps of scalar bit tests
      const u64 cur_v = remv[c];
      u64 cur = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(cur_v >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)(cur_v & 0xffffffffu));
      const int valid = (n - c * 64) < 64 ? (n - c * 64) : 64;
      if (valid < 64) cur |= ~0ull << valid;  // positions past the end count as suppressed
      // a diagonal word only holds bits of LATER candidates (nms_mask_kernel), so bit j is final once step j - 1 is
      // done and the kept set is the complement of the word the chain ends with
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const u64 dj = readlane64(diag, j);
        u64 tmp;  // cur |= bit j of cur ? 0 : dj -- three scalar instructions per candidate
        asm volatile("s_bitcmp0_b64 %0, %2\n\ts_cselect_b64 %1, %3, 0\n\ts_or_b64 %0, %0, %1"
                     : "+s"(cur), "=&s"(tmp)
                     : "n"(j), "s"(dj)
                     : "scc");
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // keep the lane reads next to their step (no SGPR spills)
      }
      const u64 keep = ~cur;
      if (lane == 0) {
        keep_s = keep;
        kept_s[c] = keep;
      }
    }
    __syncthreads();
    const u64 keep = keep_s;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      u64 acc = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((keep >> (wid * 4 + q)) & 1ull) acc |= pre[g][q];
      const int w = c + 1 + 64 * g + lane;
      if (acc && w < nb) atomicOr(reinterpret_cast<unsigned long long*>(&remv[w]), (unsigned long long)acc);
    }
    if (c + 1 < nb) prefetch(c + 1);
    __syncthreads();  // remv complete for chunk c + 1; keep_s free
  }
  // kept candidates by original index
  for (int i = tid; i < n; i += blockDim.x)
    if ((kept_s[i >> 6] >> (i & 63)) & 1ull) flag[order[i]] = 1;
  __syncthreads();
  // compaction: thread t owns original indices [8t, 8t+8)
  int cnt = 0;
  unsigned int bits = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = tid * 8 + e;
    if (i < n && flag[i]) {
      bits |= 1u << e;
      ++cnt;
    }
  }
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wave_tot[wid] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wid; ++w) base += wave_tot[w];
  int pos = base + incl - cnt;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (bits & (1u << e)) keep_out[pos++] = (int64_t)(tid * 8 + e);
  if (tid == blockDim.x - 1) num_keep[0] = base + incl;
}

// ------------------------------------------------------------------------------------------------------------------
// n > SCAN_NMS_PANEL.  The reference's CUDA path has no size limit (csrc/cuda/nms.cu:70-131: an n x ceil(n/64) mask and a
// host scan); here the three stages stay on the device.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void nms_keys_kernel(const float* __restrict__ scores, int n, int n_pad,
                                                        u64* __restrict__ keys) {
  const int i = blockIdx.x * 1024 + threadIdx.x;
  if (i < n_pad) keys[i] = (i < n) ? (((u64)f2sortable_desc(scores[i]) << 32) | (unsigned int)i) : ~0ull;
}

// one workgroup = SCAN_NMS_PANEL consecutive keys in LDS.  FULL: every stage k = 2 .. SCAN_NMS_PANEL of the bitonic network
// (directions from the GLOBAL position); else the steps j = SCAN_NMS_PANEL/2 .. 1 of stage k_merge.
template <bool FULL>
__global__ __launch_bounds__(1024) void nms_sort_local_kernel(u64* __restrict__ keys_g, int k_merge) {
  extern __shared__ u64 keys[];
  const int base = blockIdx.x * SCAN_NMS_PANEL;
  for (int i = threadIdx.x; i < SCAN_NMS_PANEL; i += 1024) keys[i] = keys_g[base + i];
  __syncthreads();
  for (int k = FULL ? 2 : k_merge; k <= (FULL ? SCAN_NMS_PANEL : k_merge); k <<= 1) {
    for (int j = (k >> 1) < SCAN_NMS_PANEL / 2 ? (k >> 1) : SCAN_NMS_PANEL / 2; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < SCAN_NMS_PANEL; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 a = keys[i], b = keys[ixj];
          const bool up = ((base + i) & k) == 0;
          if ((a > b) == up) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < SCAN_NMS_PANEL; i += 1024) keys_g[base + i] = keys[i];
}

// one compare-exchange step (k, j) with j >= SCAN_NMS_PANEL: partners sit in different LDS blocks
__global__ __launch_bounds__(1024) void nms_sort_global_kernel(u64* __restrict__ keys, int k, int j, int n_pad) {
  const int t = blockIdx.x * 1024 + threadIdx.x;
  if (t >= n_pad / 2) return;
  const int lo = t & (j - 1);
  const int i = ((t - lo) << 1) | lo, ixj = i | j;
  const u64 a = keys[i], b = keys[ixj];
  const bool up = (i & k) == 0;
  if ((a > b) == up) {
    keys[i] = b;
    keys[ixj] = a;
  }
}

__global__ __launch_bounds__(1024) void nms_gather_kernel(const u64* __restrict__ keys, const float* __restrict__ dets,
                                                          const float* __restrict__ labels, int n, int* __restrict__ order,
                                                          float4* __restrict__ boxes_sorted, float* __restrict__ labels_sorted,
                                                          float* __restrict__ areas_sorted) {
  const int i = blockIdx.x * 1024 + threadIdx.x;
  if (i >= n) return;
  const int o = (int)(keys[i] & 0xffffffffu);
  order[i] = o;
  const float4 b = reinterpret_cast<const float4*>(dets)[o];
  boxes_sorted[i] = b;
  areas_sorted[i] = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
  labels_sorted[i] = labels ? labels[o] : 0.f;
}

// one workgroup of 1024 threads; dynamic LDS: remv[nbp] | kept[nbp] | diag[SCAN_NMS_PANEL]
__global__ __launch_bounds__(1024) void nms_scan_panel_kernel(const u64* __restrict__ mask, const int* __restrict__ order,
                                                              int n, int nb, int nbp, unsigned char* __restrict__ flag,
                                                              int64_t* __restrict__ keep_out, int* __restrict__ num_keep) {
  extern __shared__ u64 sm[];
  u64* remv = sm;
  u64* kept_s = sm + nbp;
  u64* diag_s = sm + 2 * nbp;
  __shared__ u64 keep_s;
  __shared__ int wave_tot[16];
  __shared__ int run_base;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < nbp; i += 1024) remv[i] = 0;
  for (int i = tid; i < n; i += 1024) flag[i] = 0;
  if (tid == 0) run_base = 0;
  for (int c = 0; c < nb; ++c) {
    if ((c & (SCAN_NMS_PANEL / 64 - 1)) == 0) {
      // the diagonal words of the next panel of candidates (the previous panel's last reader passed the barrier that
      // ends iteration c - 1)
      for (int i = tid; i < SCAN_NMS_PANEL; i += 1024) {
        const int row = c * 64 + i;
        diag_s[i] = row < n ? mask[(int64_t)row * nb + (row >> 6)] : 0ull;
      }
      __syncthreads();
    }
    if (wid == 0) {  // the same 64 scalar steps as nms_scan_kernel
      const u64 diag = diag_s[(c & (SCAN_NMS_PANEL / 64 - 1)) * 64 + lane];
      const u64 cur_v = remv[c];
      u64 cur = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(cur_v >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)(cur_v & 0xffffffffu));
      const int valid = (n - c * 64) < 64 ? (n - c * 64) : 64;
      if (valid < 64) cur |= ~0ull << valid;
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const u64 dj = readlane64(diag, j);
        u64 tmp;
        asm volatile("s_bitcmp0_b64 %0, %2\n\ts_cselect_b64 %1, %3, 0\n\ts_or_b64 %0, %0, %1"
                     : "+s"(cur), "=&s"(tmp)
                     : "n"(j), "s"(dj)
                     : "scc");
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
      const u64 keep = ~cur;
      if (lane == 0) {
        keep_s = keep;
        kept_s[c] = keep;
      }
    }
    __syncthreads();
    const u64 keep = keep_s;
    // every wave ORs its four rows of chunk c into the running words of ALL later chunks, 64 words at a time
    if ((keep >> (wid * 4)) & 15ull) {
      for (int w0 = c + 1; w0 < nb; w0 += 64) {
        const int w = w0 + lane;
        u64 acc = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = c * 64 + wid * 4 + q;
          if (((keep >> (wid * 4 + q)) & 1ull) && w < nb && row < n) acc |= mask[(int64_t)row * nb + w];
        }
        if (acc) atomicOr(reinterpret_cast<unsigned long long*>(&remv[w]), (unsigned long long)acc);
      }
    }
    __syncthreads();
  }
  // kept candidates by original index (global bytes: written and read by this workgroup only, barrier in between)
  for (int i = tid; i < n; i += 1024)
    if ((kept_s[i >> 6] >> (i & 63)) & 1ull) flag[order[i]] = 1;
  __syncthreads();
  for (int p0 = 0; p0 < n; p0 += SCAN_NMS_PANEL) {  // compaction, 8 consecutive original indices per thread and pass
    int cnt = 0;
    unsigned int bits = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = p0 + tid * 8 + e;
      if (i < n && flag[i]) {
        bits |= 1u << e;
        ++cnt;
      }
    }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    int base = run_base;
    for (int w = 0; w < wid; ++w) base += wave_tot[w];
    int pos = base + incl - cnt;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (bits & (1u << e)) keep_out[pos++] = (int64_t)(p0 + tid * 8 + e);
    __syncthreads();
    if (tid == 1023) run_base = base + incl;
    __syncthreads();
  }
  if (tid == 0) num_keep[0] = run_base;
}

static int next_pow2(int n) {
  int p = 64;
  while (p < n) p <<= 1;
  return p;
}

struct NmsWs {
  int* order;
  float4* boxes;
  float* labels;
  float* areas;
  u64* mask;
  u64* keys;            // panel path only
  unsigned char* flag;  // panel path only
  size_t bytes;
};
static NmsWs nms_layout(void* base, int64_t n) {
  NmsWs w;
  const int n_pad = next_pow2((int)n);
  const int nb = (int)((n + 63) / 64);
  const bool panel = n > SCAN_NMS_PANEL;
  char* p = reinterpret_cast<char*>(base);
  size_t off = 0;
  w.boxes = reinterpret_cast<float4*>(p + off);
  off += sizeof(float4) * n_pad;
  w.mask = reinterpret_cast<u64*>(p + off);
  // one word per (row, 64-column tile); the panel path sizes it by the rows that exist, not by the padded sort length
  off += sizeof(u64) * (size_t)(panel ? (size_t)nb * 64 : (size_t)n_pad) * (nb > 0 ? nb : 1);
  w.order = reinterpret_cast<int*>(p + off);
  off += sizeof(int) * n_pad;
  w.labels = reinterpret_cast<float*>(p + off);
  off += sizeof(float) * n_pad;
  w.areas = reinterpret_cast<float*>(p + off);
  off += sizeof(float) * n_pad;
  w.keys = reinterpret_cast<u64*>(p + off);
  w.flag = nullptr;
  if (panel) {
    off += sizeof(u64) * n_pad;
    w.flag = reinterpret_cast<unsigned char*>(p + off);
    off += (size_t)n_pad;
  }
  w.bytes = off;
  return w;
}

extern "C" int64_t scan_nms_ws_bytes(int64_t n) {
  if (n <= 0) return 16;
  if (n > SCAN_NMS_MAX) return -1;
  return (int64_t)nms_layout(nullptr, n).bytes;
}

extern "C" int scan_nms(const float* dets, const float* scores, const float* labels, int64_t n, float thr,
                        int32_t rule_ge, int64_t* keep_out, int32_t* num_keep_out, void* workspace, void* stream) {
  SCAN_CHECK_ARG(n >= 0, "nms: negative n");
  SCAN_CHECK_ARG(n <= SCAN_NMS_MAX, "nms: n=%lld exceeds SCAN_NMS_MAX=%d", (long long)n, SCAN_NMS_MAX);
  SCAN_CHECK_ARG(num_keep_out, "nms: null num_keep_out");
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(num_keep_out, 0, sizeof(int), st) != hipSuccess) {
      scan_set_error("nms: memset failed");
      return -2;
    }
    return 0;
  }
  SCAN_CHECK_ARG(dets && scores && keep_out && workspace, "nms: null pointer");
  SCAN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(dets) & 15) == 0,
                 "nms: dets/workspace must be 16-byte aligned");
  const NmsWs w = nms_layout(workspace, n);
  const int n_pad = next_pow2((int)n);
  const int nb = (int)((n + 63) / 64);
  static bool scan_attr = false;
  if (!scan_attr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        SCAN_NMS_PANEL * (int)sizeof(u64));
    hipFuncSetAttribute(reinterpret_cast<const void*>(nms_scan_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (SCAN_NMS_PANEL + 2 * (SCAN_NMS_MAX / 64)) * (int)sizeof(u64));
    hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_local_kernel<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_NMS_PANEL * (int)sizeof(u64));
    hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_local_kernel<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_NMS_PANEL * (int)sizeof(u64));
    scan_attr = true;
  }
  if (n <= SCAN_NMS_PANEL) {
    hipLaunchKernelGGL(nms_sort_kernel, dim3(1), dim3(1024), sizeof(u64) * n_pad, st, dets, scores, labels, (int)n, n_pad,
                       w.order, w.boxes, w.labels, w.areas);
    SCAN_LAUNCH_CHECK("nms_sort");
  } else {
    // bitonic network over n_pad keys: stages k <= 8192 inside LDS blocks, then per stage the steps with partners in
    // different blocks one launch each and the rest of the stage in LDS again (n = 20,000: 6 launches)
    const size_t lds = sizeof(u64) * SCAN_NMS_PANEL;
    hipLaunchKernelGGL(nms_keys_kernel, dim3(n_pad / 1024), dim3(1024), 0, st, scores, (int)n, n_pad, w.keys);
    hipLaunchKernelGGL(nms_sort_local_kernel<true>, dim3(n_pad / SCAN_NMS_PANEL), dim3(1024), lds, st, w.keys, 0);
    for (int k = 2 * SCAN_NMS_PANEL; k <= n_pad; k <<= 1) {
      for (int j = k >> 1; j >= SCAN_NMS_PANEL; j >>= 1)
        hipLaunchKernelGGL(nms_sort_global_kernel, dim3(n_pad / 2048), dim3(1024), 0, st, w.keys, k, j, n_pad);
      hipLaunchKernelGGL(nms_sort_local_kernel<false>, dim3(n_pad / SCAN_NMS_PANEL), dim3(1024), lds, st, w.keys, k);
    }
    hipLaunchKernelGGL(nms_gather_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, st, w.keys, dets, labels, (int)n,
                       w.order, w.boxes, w.labels, w.areas);
    SCAN_LAUNCH_CHECK("nms_sort (panel path)");
  }
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb), dim3(64), 0, st, w.boxes, w.areas, w.labels, (int)n, thr, rule_ge,
                     labels != nullptr ? 1 : 0, w.mask, nb);
  SCAN_LAUNCH_CHECK("nms_mask");
  if (n <= SCAN_NMS_PANEL) {
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), sizeof(u64) * 64 * nb, st, w.mask, w.order, (int)n, nb, keep_out,
                       num_keep_out);
  } else {
    const int nbp = (nb + 63) / 64 * 64;
    hipLaunchKernelGGL(nms_scan_panel_kernel, dim3(1), dim3(1024), sizeof(u64) * (SCAN_NMS_PANEL + 2 * nbp), st, w.mask,
                       w.order, (int)n, nb, nbp, w.flag, keep_out, num_keep_out);
  }
  SCAN_LAUNCH_CHECK("nms_scan");
  return 0;
}
