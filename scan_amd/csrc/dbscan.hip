// Density clustering of the target-domain graph nodes on the device: what the selection in
// PrototypeComputation.DBSCAN_batch_cpu (reference rpn/fcos/loss.py:397-423) needs from
// sklearn.cluster.DBSCAN(eps, min_samples=5).fit_predict(points): "noise -> 1, cluster 0 -> 0" and a pixel is
// selected when any of its class entries is non-zero, i.e. a point is selected iff it is NOT in cluster 0.
// sklearn's label rule (sklearn/cluster/_dbscan_inner.pyx): points are visited in index order, the first unlabeled
// CORE point opens the next cluster, expansion runs through core points, a border point takes the label of the first
// cluster that reaches it.  Hence cluster 0 = the connected component (over core-core eps links) of the lowest-index
// core point, plus every non-core point within eps of one of its core points -- no other cluster matters.
//
//   1. neighbours   tiled P P^T on the matrix cores -- bf16x3 (operands split hi + lo bf16, three v_mfma_f32_32x32x16_bf16
//                   per product, the convolutions' arithmetic; scan_tune "dbscan_bf16x3", default) or exact fp32
//                   (v_mfma_f32_32x32x2_f32) --, d2 = |p_i|^2 + |p_j|^2 - 2 p_i.p_j
//                   thresholded at eps^2; pairs whose fp32 value lies within the rounding band of the threshold are
//                   recomputed as sum (a - b)^2 in fp64 (sklearn's brute-force radius query decides in fp64).  The
//                   adjacency is kept as a bit matrix (n^2 / 8 bytes in HBM) built with wave ballots; row popcounts
//                   give the neighbour counts (self included, like sklearn).
//   2. core points  count >= min_samples -> core bit mask, lowest core index.
//   3. cluster 0    breadth-first over the bit matrix: one launch per level (the host loop reads one flag), a wave per
//                   frontier row ORs row & core & ~visited into the visited / next-frontier masks.
//   4. finish       core point: in cluster 0 iff visited; border / noise: iff adjacent to a visited core point.
// O(n^2) like the host algorithm, but the n^2 part is a GEMM: 31 k points take ~10 ms instead of 3.2 s of sklearn.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DB_T 128
#define DB_K 32
#define DB_LD (DB_K + 4)

__global__ __launch_bounds__(256) void dbscan_sqnorm_kernel(const float* __restrict__ p, int64_t n, int D,
                                                            double* __restrict__ sq) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  double s = 0.0;
  for (int k = lane; k < D; k += 64) {
    const double v = (double)p[row * D + k];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) sq[row] = s;
}

typedef __bf16 db_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 db_bf16x4 __attribute__((ext_vector_type(4)));
#define DB_LDH 40  // bf16 row pitch of the split planes: 32 + 8, rows stay 16-byte aligned

// hi = bf16(x), lo = bf16(x - hi): the split of the convolution kernels (conv_bf16x3_v2.hip)
__device__ __forceinline__ void db_split4(const float4 v, db_bf16x4& hi, db_bf16x4& lo) {
  hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
  lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
  lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
}

template <bool BF>
__global__ __launch_bounds__(256) void dbscan_neighbors_kernel(const float* __restrict__ p, int64_t n, int D,
                                                               const double* __restrict__ sq, double eps2,
                                                               uint32_t* __restrict__ bits, int64_t nw) {
  // fp32: As / Bs [128][36] floats; bf16x3: four planes (A hi, A lo, B hi, B lo) [128][40] bf16 in the same bytes
  constexpr int F32_FLOATS = 2 * DB_T * DB_LD, BF_FLOATS = 4 * DB_T * DB_LDH / 2;
  __shared__ __align__(16) float smem[BF && BF_FLOATS > F32_FLOATS ? BF_FLOATS : F32_FLOATS];
  float* As = smem;
  float* Bs = smem + DB_T * DB_LD;
  __bf16* Ah = reinterpret_cast<__bf16*>(smem);
  __bf16* Al = Ah + DB_T * DB_LDH;
  __bf16* Bh = Al + DB_T * DB_LDH;
  __bf16* Bl = Bh + DB_T * DB_LDH;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // the relation is symmetric: only tiles on or above the diagonal are computed, each writes its bits in both
  // orientations (row words by wave ballot, column words from the lane's own 16 accumulator rows)
  if (blockIdx.x < blockIdx.y) return;
  const bool off_diag = blockIdx.x != blockIdx.y;
  const int64_t i0 = (int64_t)blockIdx.y * DB_T, j0 = (int64_t)blockIdx.x * DB_T;
  const int c4 = tid & 7, r0 = tid >> 3;
  const int wm = wid >> 1, wn = wid & 1, lrow = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // the next K chunk is fetched into registers while the matrix cores work on the current one
  float4 ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + 32 * i, k = k0 + 4 * c4;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      rb[i] = ra[i];
      if (k < D) {
        if (i0 + r < n) ra[i] = *reinterpret_cast<const float4*>(p + (i0 + r) * D + k);
        if (j0 + r < n) rb[i] = *reinterpret_cast<const float4*>(p + (j0 + r) * D + k);
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < D; k0 += DB_K) {
    __syncthreads();
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = r0 + 32 * i;
        db_bf16x4 h, l;
        db_split4(ra[i], h, l);
        *reinterpret_cast<db_bf16x4*>(Ah + r * DB_LDH + 4 * c4) = h;
        *reinterpret_cast<db_bf16x4*>(Al + r * DB_LDH + 4 * c4) = l;
        db_split4(rb[i], h, l);
        *reinterpret_cast<db_bf16x4*>(Bh + r * DB_LDH + 4 * c4) = h;
        *reinterpret_cast<db_bf16x4*>(Bl + r * DB_LDH + 4 * c4) = l;
      }
      if (k0 + DB_K < D) fetch(k0 + DB_K);
      __syncthreads();
      // 32x32x16: lane l holds row l & 31, k = 8 (l >> 5) .. + 7 of a 16-wide k step
      const int fo = (wm * 64 + lrow) * DB_LDH + 8 * lh, go = (wn * 64 + lrow) * DB_LDH + 8 * lh;
#pragma unroll
      for (int ks = 0; ks < DB_K / 16; ++ks) {
        db_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ah[t] = *reinterpret_cast<const db_bf16x8*>(Ah + fo + t * 32 * DB_LDH + 16 * ks);
          al[t] = *reinterpret_cast<const db_bf16x8*>(Al + fo + t * 32 * DB_LDH + 16 * ks);
          bh[t] = *reinterpret_cast<const db_bf16x8*>(Bh + go + t * 32 * DB_LDH + 16 * ks);
          bl[t] = *reinterpret_cast<const db_bf16x8*>(Bl + go + t * 32 * DB_LDH + 16 * ks);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + 32 * i;
      *reinterpret_cast<float4*>(As + r * DB_LD + 4 * c4) = ra[i];
      *reinterpret_cast<float4*>(Bs + r * DB_LD + 4 * c4) = rb[i];
    }
    if (k0 + DB_K < D) fetch(k0 + DB_K);
    __syncthreads();
    const float* a = As + (wm * 64 + lrow) * DB_LD + 4 * lh;
    const float* b = Bs + (wn * 64 + lrow) * DB_LD + 4 * lh;
#pragma unroll
    for (int j = 0; j < DB_K / 8; ++j) {
      float4 av[2], bv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        av[t] = *reinterpret_cast<const float4*>(a + t * 32 * DB_LD + 8 * j);
        bv[t] = *reinterpret_cast<const float4*>(b + t * 32 * DB_LD + 8 * j);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].z, bv[tn].z, acc[tm][tn], 0, 0, 0);
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].w, bv[tn].w, acc[tm][tn], 0, 0, 0);
        }
    }
  }

  // C/D map of 32x32: col = lane & 31 (j), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (i)
  // The threshold test runs in fp32 (one subtract, two FMAs, two compares per pair -- in fp64 this epilogue cost more than
  // the bf16x3 GEMM in front of it); what fp32 adds to the rounding band is 3 * 2^-24 (si + sj) + 2^-24 eps2, far inside
  // the margin the band constants carry.  Squared norms of the tile's rows / columns as floats, through LDS.
  __syncthreads();
  float* sqa = smem;            // [128] rows i0 ..
  float* sqb = smem + DB_T;     // [128] columns j0 ..
  if (tid < DB_T) {
    sqa[tid] = i0 + tid < n ? (float)sq[i0 + tid] : 0.f;
    sqb[tid] = j0 + tid < n ? (float)sq[j0 + tid] : 0.f;
  }
  __syncthreads();
  const float eps2f = (float)eps2;
  const float bandc = BF ? 5.9e-5f : 1.7e-5f;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int64_t j = j0 + (wn * 2 + tn) * 32 + lrow;
    const float sjf = sqb[(wn * 2 + tn) * 32 + lrow];
    const int64_t jw = (j0 + (wn * 2 + tn) * 32) >> 5;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      uint32_t colbits = 0u;  // this lane's column j over the 16 rows it holds
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int64_t i = i0 + (wm * 2 + tm) * 32 + rr;
        const bool valid = i < n && j < n;
        const float ssum = sqa[(wm * 2 + tm) * 32 + rr] + sjf;
        // d2 - eps2 in fp32.  Band: worst-case fp32 accumulation error of the 256-term dot product, (K - 1) 2^-24
        // sum |a_k b_k| <= 1.53e-5 (si + sj) / 2, doubled by the factor 2 in front of it.  bf16x3: with a = ah + al + da,
        // |da| <= 2^-18 |a| (two round-to-nearest bf16 steps), the dropped part of a product is al bl + da b + a db,
        // <= 3 * 2^-18 |a b| = 1.15e-5 |a b|; the products kept are exact in fp32 and there are three times as many
        // of them to add up, (3 K - 1) 2^-24 = 4.6e-5 in the same worst-case count: 5.8e-5 (si + sj) on d2
        const float t = __builtin_fmaf(-2.0f, acc[tm][tn][r], ssum - eps2f);
        const float band = __builtin_fmaf(bandc, ssum, 1e-6f * eps2f + 1e-9f);
        const bool inband = valid && fabsf(t) <= band;
        bool pred = valid && t <= 0.f;
        // pairs inside the rounding band are decided in fp64 from the points themselves -- by the WHOLE wave, one
        // pair at a time (coalesced reads of both rows, butterfly sum): a lane looping over D on its own while 63
        // others wait made dense point sets (many pairs near eps) ~100x slower than the GEMM itself
        unsigned long long todo = __ballot(inband);
        while (todo) {
          const int src = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const int64_t ii = (int64_t)__shfl((int)(i - i0), src, 64) + i0;
          const int64_t jj = (int64_t)__shfl((int)(j - j0), src, 64) + j0;
          double e = 0.0;
          for (int k = lane; k < D; k += 64) {
            const double df = (double)p[ii * D + k] - (double)p[jj * D + k];
            e += df * df;
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
          if (lane == src) pred = e <= eps2;
        }
        colbits |= pred ? (1u << rr) : 0u;
        const unsigned long long m = __ballot(pred);
        // lanes 0..31 hold row i(lh = 0), lanes 32..63 row i + 4
        if (lrow == 0 && i < n) {
          const uint32_t half = lh ? (uint32_t)(m >> 32) : (uint32_t)m;
          bits[i * nw + jw] = half;
        }
      }
      if (off_diag) {  // mirrored tile: row j, the word over rows i
        colbits |= (uint32_t)__shfl_xor((int)colbits, 32, 64);
        if (lh == 0 && j < n) {
          bits[j * nw + ((i0 + (wm * 2 + tm) * 32) >> 5)] = colbits;
        }
      }
    }
  }
}

// neighbour count of a point = popcount of its row (self included); core: count >= min_samples.  A wave per row.
__global__ __launch_bounds__(256) void dbscan_core_kernel(const uint32_t* __restrict__ bits, int64_t n, int64_t nw,
                                                          int min_samples, int* __restrict__ counts,
                                                          uint32_t* __restrict__ core, int* __restrict__ first_core) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  int c = 0;
  for (int64_t w = lane; w < nw; w += 64) c += __popc(bits[i * nw + w]);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) {
    counts[i] = c;
    if (c >= min_samples) {
      atomicOr(&core[i >> 5], 1u << (i & 31));
      atomicMin(first_core, (int)i);
    }
  }
}

__global__ void dbscan_init_kernel(int* __restrict__ first_core, int n) { *first_core = n; }

__global__ void dbscan_seed_kernel(const int* __restrict__ first_core, int64_t n, uint32_t* __restrict__ visited,
                                   uint32_t* __restrict__ frontier) {
  const int f = *first_core;
  if (f >= 0 && f < n) {
    visited[f >> 5] = 1u << (f & 31);
    frontier[f >> 5] = 1u << (f & 31);
  }
}

// one breadth-first level: a wave per row; rows outside the frontier return at once
__global__ __launch_bounds__(256) void dbscan_bfs_kernel(const uint32_t* __restrict__ bits, int64_t n, int64_t nw,
                                                         const uint32_t* __restrict__ core,
                                                         uint32_t* __restrict__ visited,
                                                         const uint32_t* __restrict__ frontier,
                                                         uint32_t* __restrict__ next, int* __restrict__ changed) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n || !((frontier[i >> 5] >> (i & 31)) & 1u)) return;
  bool any = false;
  for (int64_t w = lane; w < nw; w += 64) {
    const uint32_t cand = bits[i * nw + w] & core[w] & ~visited[w];
    if (cand) {
      const uint32_t old = atomicOr(&visited[w], cand);
      const uint32_t fresh = cand & ~old;
      if (fresh) {
        atomicOr(&next[w], fresh);
        any = true;
      }
    }
  }
  if (__ballot(any) && lane == 0) *changed = 1;
}

__global__ __launch_bounds__(256) void dbscan_finish_kernel(const uint32_t* __restrict__ bits, int64_t n, int64_t nw,
                                                            const uint32_t* __restrict__ core,
                                                            const uint32_t* __restrict__ visited,
                                                            uint8_t* __restrict__ in0) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const bool is_core = (core[i >> 5] >> (i & 31)) & 1u;
  bool hit;
  if (is_core) {
    hit = (visited[i >> 5] >> (i & 31)) & 1u;
  } else {
    bool any = false;
    for (int64_t w = lane; w < nw; w += 64) any |= (bits[i * nw + w] & visited[w]) != 0u;  // visited holds core points only
    hit = __ballot(any) != 0ull;
  }
  if (lane == 0) in0[i] = hit ? 1 : 0;
}

static int64_t db_words(int64_t n) { return (n + DB_T - 1) / DB_T * (DB_T / 32); }

// workspace layout (bytes): bits [n][nw] u32 | sq [n] f64 | counts [n] i32 | core, visited, frontier A, frontier B [nw] u32
// each | first_core, changed i32
extern "C" int64_t scan_dbscan_ws_bytes(int64_t n) {
  if (n <= 0 || n > SCAN_DBSCAN_MAX) return -1;
  const int64_t nw = db_words(n);
  return n * nw * 4 + n * 8 + n * 4 + 4 * nw * 4 + 64;
}

struct DbWs {
  uint32_t* bits;
  double* sq;
  int* counts;
  uint32_t *core, *visited, *fa, *fb;
  int *first_core, *changed;
  int64_t nw;
};

static DbWs db_split(void* ws, int64_t n) {
  DbWs w;
  w.nw = db_words(n);
  char* p = reinterpret_cast<char*>(ws);
  w.bits = reinterpret_cast<uint32_t*>(p);
  p += n * w.nw * 4;
  w.sq = reinterpret_cast<double*>(p);
  p += n * 8;
  w.counts = reinterpret_cast<int*>(p);
  p += n * 4;
  w.core = reinterpret_cast<uint32_t*>(p);
  w.visited = w.core + w.nw;
  w.fa = w.visited + w.nw;
  w.fb = w.fa + w.nw;
  w.first_core = reinterpret_cast<int*>(w.fb + w.nw);
  w.changed = w.first_core + 1;
  return w;
}

// steps 1-2 and the seed of step 3.  info (device, int32[2]) <- {lowest core index or n if there is none, 0}
// scan_tune "dbscan_bf16x3": 1 (default) = the pairwise-distance GEMM runs as bf16x3 (three bf16 MFMAs per product) with the
// wider exact re-check band; 0 = exact fp32 MFMA.  Same neighbour bits either way (the band decides in fp64).
int g_scan_dbscan_bf16x3 = 1;

extern "C" int scan_dbscan_prepare(const float* pts, int64_t n, int32_t D, float eps, int32_t min_samples, void* ws,
                                   int32_t* info, void* stream) {
  SCAN_CHECK_ARG(pts && ws && info, "dbscan_prepare: null pointer");
  SCAN_CHECK_ARG(n > 0 && n <= SCAN_DBSCAN_MAX, "dbscan_prepare: n=%lld out of range (SCAN_DBSCAN_MAX)", (long long)n);
  SCAN_CHECK_ARG(D > 0 && D % 4 == 0, "dbscan_prepare: D=%d must be a positive multiple of 4", D);
  SCAN_CHECK_ARG(eps > 0.f && min_samples >= 1, "dbscan_prepare: eps / min_samples");
  hipStream_t st = as_stream(stream);
  DbWs w = db_split(ws, n);
  // everything after the bit matrix and the norms starts at zero; first_core starts at n
  if (hipMemsetAsync(w.counts, 0, (size_t)(n * 4 + 4 * w.nw * 4 + 64), st) != hipSuccess) {
    scan_set_error("dbscan_prepare: memset failed");
    return -2;
  }
  hipLaunchKernelGGL(dbscan_init_kernel, dim3(1), dim3(1), 0, st, w.first_core, (int)n);
  hipLaunchKernelGGL(dbscan_sqnorm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, pts, n, D, w.sq);
  const unsigned tiles = (unsigned)((n + DB_T - 1) / DB_T);
  if (g_scan_dbscan_bf16x3)
    hipLaunchKernelGGL(dbscan_neighbors_kernel<true>, dim3(tiles, tiles), dim3(256), 0, st, pts, n, D, w.sq,
                       (double)eps * (double)eps, w.bits, w.nw);
  else
    hipLaunchKernelGGL(dbscan_neighbors_kernel<false>, dim3(tiles, tiles), dim3(256), 0, st, pts, n, D, w.sq,
                       (double)eps * (double)eps, w.bits, w.nw);
  hipLaunchKernelGGL(dbscan_core_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, w.bits, n, w.nw, min_samples,
                     w.counts, w.core, w.first_core);
  hipLaunchKernelGGL(dbscan_seed_kernel, dim3(1), dim3(1), 0, st, w.first_core, n, w.visited, w.fa);
  if (hipMemcpyAsync(info, w.first_core, 8, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    scan_set_error("dbscan_prepare: info copy failed");
    return -2;
  }
  SCAN_LAUNCH_CHECK("dbscan_prepare");
  return 0;
}

// one breadth-first level; parity = 0, 1, 0, ... selects which frontier buffer is read.  changed (device int32) is
// cleared, then set to 1 if the level reached new core points: the caller loops while it reads 1.
extern "C" int scan_dbscan_bfs_step(int64_t n, void* ws, int32_t parity, int32_t* changed, void* stream) {
  SCAN_CHECK_ARG(ws && changed && n > 0 && n <= SCAN_DBSCAN_MAX, "dbscan_bfs_step: bad arguments");
  hipStream_t st = as_stream(stream);
  DbWs w = db_split(ws, n);
  uint32_t* cur = parity ? w.fb : w.fa;
  uint32_t* nxt = parity ? w.fa : w.fb;
  if (hipMemsetAsync(nxt, 0, (size_t)w.nw * 4, st) != hipSuccess || hipMemsetAsync(changed, 0, 4, st) != hipSuccess) {
    scan_set_error("dbscan_bfs_step: memset failed");
    return -2;
  }
  hipLaunchKernelGGL(dbscan_bfs_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, w.bits, n, w.nw, w.core,
                     w.visited, cur, nxt, changed);
  SCAN_LAUNCH_CHECK("dbscan_bfs_step");
  return 0;
}

// in_cluster0 [n] (uint8): 1 where sklearn's label would be 0
extern "C" int scan_dbscan_finish(int64_t n, void* ws, uint8_t* in_cluster0, void* stream) {
  SCAN_CHECK_ARG(ws && in_cluster0 && n > 0 && n <= SCAN_DBSCAN_MAX, "dbscan_finish: bad arguments");
  DbWs w = db_split(ws, n);
  hipLaunchKernelGGL(dbscan_finish_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, as_stream(stream), w.bits, n,
                     w.nw, w.core, w.visited, in_cluster0);
  SCAN_LAUNCH_CHECK("dbscan_finish");
  return 0;
}
