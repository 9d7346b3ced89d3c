// Everything the source pass derives from the ground truth alone, on the device in three launches:
//   scan_fcos_assign   FCOS location -> GT assignment (reference rpn/fcos/loss.py:40-126; PrototypeComputation holds an
//                      identical copy at :262-343): labels [M], regression targets [M,4], positives per level
//   scan_fcos_compact  per level the rows with label > 0 and the rows with label == 0, each in row order
//   scan_fcos_nodes    graph-node sampling index of the source branch (loss.py:428-463: per level the positives in row
//                      order and floor(linspace(0, n_neg - 2, n_pos)) of the background rows, order [all neg, all pos]),
//                      the positive rows with their regression / centerness targets (loss.py:128-133, 197-222)
// The torch spelling of the same plan (modeling/fcos.py: assign_targets, source_node_index, centerness_targets -- pinned
// against the reference's label maps and node lists) took ~115 launches and a dozen host round trips (nonzero) per
// iteration; here the host reads five counters once.  Integer / index work and single fp32 operations in the reference's
// order: bit-identical to the torch spelling (tests/test_gpu_kernels.py::test_target_plan_kernels_equal_torch_plan).
#include "common.h"

#define FCOS_INF 100000000.0f  // reference rpn/fcos/loss.py:22

struct FcosLevels {
  int stride[SCAN_MAX_LEVELS];
  float lo[SCAN_MAX_LEVELS], hi[SCAN_MAX_LEVELS];  // sizes of interest per level (loss.py:41-47)
};

__global__ __launch_bounds__(256) void fcos_assign_kernel(scan_pyramid_t d, FcosLevels lv, const float* __restrict__ boxes,
                                                          const int64_t* __restrict__ glabels,
                                                          const int32_t* __restrict__ ng, int G,
                                                          int64_t* __restrict__ labels, int32_t* __restrict__ labels_i32,
                                                          float* __restrict__ reg, int32_t* __restrict__ level_pos) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= d.row_off[d.n_levels]) return;
  const RowCoord rc = decode_row(d, m);
  int s = lv.stride[0];
  float lo = lv.lo[0], hi = lv.hi[0];
#pragma unroll
  for (int i = 1; i < SCAN_MAX_LEVELS; ++i)
    if (rc.lvl == i) {
      s = lv.stride[i];
      lo = lv.lo[i];
      hi = lv.hi[i];
    }
  // compute_locations (fcos.py:234-258): arange(0, w * s, s) + s // 2
  const float xs = (float)(rc.x * s) + (float)(s / 2), ys = (float)(rc.y * s) + (float)(s / 2);
  const float* bx = boxes + (int64_t)rc.n * G * 4;
  const int n_gt = ng[rc.n];
  float best = FCOS_INF;
  int gi = 0;
  float4 rg = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = 0; g < n_gt; ++g) {
    const float4 b = *reinterpret_cast<const float4*>(bx + 4 * g);
    const float l = xs - b.x, t = ys - b.y, r = b.z - xs, bb = b.w - ys;
    const float mn = fminf(fminf(l, t), fminf(r, bb)), mx = fmaxf(fmaxf(l, t), fmaxf(r, bb));
    const bool ok = mn > 0.f && mx >= lo && mx <= hi;
    const float area = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);  // BoxList.area(): TO_REMOVE = 1
    const float a = ok ? area : FCOS_INF;
    if (g == 0 || a < best) {  // first minimum, like the reference's CPU min
      best = a;
      gi = g;
      rg = make_float4(l, t, r, bb);
    }
  }
  const int64_t lab = (n_gt > 0 && best != FCOS_INF) ? glabels[(int64_t)rc.n * G + gi] : 0;
  labels[m] = lab;
  labels_i32[m] = (int32_t)lab;
  *reinterpret_cast<float4*>(reg + 4 * m) = rg;
  if (lab > 0) atomicAdd(&level_pos[rc.lvl], 1);
}

// one workgroup per level: positives and negatives of the level in row order, at pos_list / neg_list [row_off[l] ...]
__global__ __launch_bounds__(1024) void fcos_compact_kernel(scan_pyramid_t d, const int64_t* __restrict__ labels,
                                                            int32_t* __restrict__ pos_list,
                                                            int32_t* __restrict__ neg_list) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int l = blockIdx.x;
  const int64_t r0 = d.row_off[l], r1 = d.row_off[l + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t base = r0; base < r1; base += 1024) {
    const int64_t r = base + tid;
    const bool in = r < r1;
    const bool pos = in && labels[r] > 0;
    const unsigned long long bal = __ballot(pos);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wid] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int v = wsum[w];
      if (w < wid) woff += v;
      tot += v;
    }
    const int pb = carry + woff + before;  // positives of this level before row r
    if (pos)
      pos_list[r0 + pb] = (int32_t)r;
    else if (in)
      neg_list[r0 + ((r - r0) - pb)] = (int32_t)r;
    __syncthreads();
    if (tid == 0) carry += tot;
    __syncthreads();
  }
}

struct FcosNodeTab {
  int n_pos[SCAN_MAX_LEVELS], n_neg[SCAN_MAX_LEVELS];
  int pick_off[SCAN_MAX_LEVELS + 1];  // negatives picked per level (n_pos, or all n_neg when n_pos > n_neg), prefix
  int pos_off[SCAN_MAX_LEVELS + 1];
};

__global__ __launch_bounds__(256) void fcos_nodes_kernel(scan_pyramid_t d, FcosNodeTab t, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ reg,
                                                         const int32_t* __restrict__ pos_list,
                                                         const int32_t* __restrict__ neg_list,
                                                         int64_t* __restrict__ node_index,
                                                         int64_t* __restrict__ node_labels, int64_t* __restrict__ pos_inds,
                                                         float* __restrict__ reg_pos, float* __restrict__ ctr_pos) {
  const int picks = t.pick_off[d.n_levels], npos = t.pos_off[d.n_levels];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= picks + npos) return;
  if (i < picks) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < SCAN_MAX_LEVELS; ++k)
      if (k < d.n_levels && i >= t.pick_off[k]) l = k;
    int np = t.n_pos[0], nn = t.n_neg[0];
#pragma unroll
    for (int k = 1; k < SCAN_MAX_LEVELS; ++k)
      if (l == k) {
        np = t.n_pos[k];
        nn = t.n_neg[k];
      }
    const int j = i - t.pick_off[l];
    long long idx = j;  // n_pos > n_neg: every background row
    if (np <= nn) {
      // numpy.linspace(0, nn - 2, np) in float64: arange(np) * step, last element = stop; then floor
      const double stop = (double)(nn - 2);
      if (np == 1)
        idx = 0;
      else if (j == np - 1)
        idx = (long long)floor(stop);
      else {
        const double step = stop / (double)(np - 1);
        idx = (long long)floor((double)j * step);
      }
    }
    node_index[i] = neg_list[d.row_off[l] + idx];
    node_labels[i] = 0;
    return;
  }
  const int p = i - picks;
  int l = 0;
#pragma unroll
  for (int k = 1; k < SCAN_MAX_LEVELS; ++k)
    if (k < d.n_levels && p >= t.pos_off[k]) l = k;
  const int64_t row = pos_list[d.row_off[l] + (p - t.pos_off[l])];
  node_index[i] = row;
  node_labels[i] = labels[row];
  pos_inds[p] = row;
  const float4 r = *reinterpret_cast<const float4*>(reg + 4 * row);
  *reinterpret_cast<float4*>(reg_pos + 4 * (int64_t)p) = r;
  // centerness target (loss.py:128-133): sqrt((min(l, r) / max(l, r)) * (min(t, b) / max(t, b)))
  const float a = fminf(r.x, r.z) / fmaxf(r.x, r.z), b = fminf(r.y, r.w) / fmaxf(r.y, r.w);
  ctr_pos[p] = sqrtf(a * b);
}

extern "C" int scan_fcos_assign(const scan_pyramid_t* d, const int32_t* strides, const float* soi, const float* boxes,
                                const int64_t* glabels, const int32_t* ng, int32_t G, int64_t* labels,
                                int32_t* labels_i32, float* reg, int32_t* level_pos, void* stream) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1, "fcos_assign: bad pyramid");
  SCAN_CHECK_ARG(strides && soi && boxes && glabels && ng && labels && labels_i32 && reg && level_pos && G >= 1,
                 "fcos_assign: null pointer or G < 1");
  FcosLevels lv;
  for (int l = 0; l < SCAN_MAX_LEVELS; ++l) {
    lv.stride[l] = l < d->n_levels ? strides[l] : 1;
    lv.lo[l] = l < d->n_levels ? soi[2 * l] : 0.f;
    lv.hi[l] = l < d->n_levels ? soi[2 * l + 1] : 0.f;
  }
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(level_pos, 0, sizeof(int32_t) * SCAN_MAX_LEVELS, st) != hipSuccess) {
    scan_set_error("fcos_assign: memset failed");
    return -2;
  }
  const int64_t M = d->row_off[d->n_levels];
  hipLaunchKernelGGL(fcos_assign_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, *d, lv, boxes, glabels, ng, G,
                     labels, labels_i32, reg, level_pos);
  SCAN_LAUNCH_CHECK("fcos_assign");
  return 0;
}

extern "C" int scan_fcos_compact(const scan_pyramid_t* d, const int64_t* labels, int32_t* pos_list, int32_t* neg_list,
                                 void* stream) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS, "fcos_compact: bad pyramid");
  SCAN_CHECK_ARG(labels && pos_list && neg_list, "fcos_compact: null pointer");
  SCAN_CHECK_ARG(d->row_off[d->n_levels] < (1ll << 31), "fcos_compact: more than 2^31 rows");
  hipLaunchKernelGGL(fcos_compact_kernel, dim3(d->n_levels), dim3(1024), 0, as_stream(stream), *d, labels, pos_list,
                     neg_list);
  SCAN_LAUNCH_CHECK("fcos_compact");
  return 0;
}

extern "C" int64_t scan_fcos_nodes_count(const scan_pyramid_t* d, const int32_t* level_pos) {
  if (!d || !level_pos) return -1;
  int64_t n = 0;
  for (int l = 0; l < d->n_levels; ++l) {
    const int64_t rows = d->row_off[l + 1] - d->row_off[l], np = level_pos[l], nn = rows - np;
    n += np + (np <= nn ? np : nn);
  }
  return n;
}

extern "C" int scan_fcos_nodes(const scan_pyramid_t* d, const int32_t* level_pos, const int64_t* labels, const float* reg,
                               const int32_t* pos_list, const int32_t* neg_list, int64_t* node_index,
                               int64_t* node_labels, int64_t* pos_inds, float* reg_pos, float* ctr_pos, void* stream) {
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS, "fcos_nodes: bad pyramid");
  SCAN_CHECK_ARG(level_pos && labels && reg && pos_list && neg_list, "fcos_nodes: null pointer");
  FcosNodeTab t;
  t.pick_off[0] = t.pos_off[0] = 0;
  for (int l = 0; l < SCAN_MAX_LEVELS; ++l) {
    if (l < d->n_levels) {
      const int64_t rows = d->row_off[l + 1] - d->row_off[l];
      SCAN_CHECK_ARG(level_pos[l] >= 0 && level_pos[l] <= rows, "fcos_nodes: level %d has %d positives of %lld rows", l,
                     level_pos[l], (long long)rows);
      t.n_pos[l] = level_pos[l];
      t.n_neg[l] = (int)(rows - level_pos[l]);
    } else {
      t.n_pos[l] = t.n_neg[l] = 0;
    }
    t.pick_off[l + 1] = t.pick_off[l] + (t.n_pos[l] <= t.n_neg[l] ? t.n_pos[l] : t.n_neg[l]);
    t.pos_off[l + 1] = t.pos_off[l] + t.n_pos[l];
  }
  const int total = t.pick_off[d->n_levels] + t.pos_off[d->n_levels];
  if (total == 0) return 0;
  SCAN_CHECK_ARG(node_index && node_labels && pos_inds && reg_pos && ctr_pos, "fcos_nodes: null output");
  hipLaunchKernelGGL(fcos_nodes_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), *d, t, labels, reg, pos_list,
                     neg_list, node_index, node_labels, pos_inds, reg_pos, ctr_pos);
  SCAN_LAUNCH_CHECK("fcos_nodes");
  return 0;
}
