// Many-small-tensor work of one training iteration folded into single launches.
//
// A DA iteration touches ~120 conv weights (bf16 hi / lo planes for the forward and the data gradient), 16 optimizer
// segments (weights / biases of 8 sub-models) and, per CKA discriminator, 8 per-class classifier branches whose
// weights are stacked into two fat convolutions.  Launched one tensor at a time these are ~400 kernels of 3-8 us that
// sit on the critical path between the MFMA kernels (tools/gpu_idle.py: "exposed small-kernel time"); here each group
// is ONE grid whose workgroups look their tensor up in a table.
#include "conv_split.h"

// ------------------------------------------------------------------ bf16 planes of many weights
// job layout (int64 x SCAN_SPLIT_JOB_WORDS, device memory): w, plane 0 (hi), plane 1, O, T, Cs, mode, rows, Csw, first
// block, plane 2 (0: a two-piece split -- conv_split.h)
//
// HBM-bound byte work: 4 B read + 4 B written per plane element.  Both modes move 16 bytes per lane on the plane side
// (8 bf16) and float4 on the weight side:
//   mode 0 (forward planes, out[o][t][c]): a thread owns 8 consecutive plane elements of one row -- two float4 reads
//          (the row is contiguous in the weight), one 16-byte store per plane.
//   mode 1 (data-gradient planes, out[c][T-1-t][o]: flip + transpose): per tap a [O][Cs] -> [Cs][O] transpose.  Read
//          along c and written along o directly, consecutive lanes were T*Cs floats apart on the read side and stored
//          2-byte scalars: PMC showed 16x the algorithmic bytes fetched (profiles/r02_pmc_traffic.json).  Now a
//          workgroup moves a 64 (o) x 64 (c) tile of one tap through LDS: float4 reads along c (256 B per weight row),
//          a conflict-free transposed LDS read (pitch 65), 16-byte stores along o.
// Same element arithmetic as weight_split_kernel (conv_bf16x3.hip): bit-identical planes.
#define SPLIT_ELEMS_PER_BLOCK 2048
#define SPLIT_TILE 64

__device__ __forceinline__ void split8_store(const float (&v)[8], __bf16* __restrict__ w0, __bf16* __restrict__ w1,
                                             __bf16* __restrict__ w2, int64_t i) {
  bf16x8 q0, q1, q2;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    __bf16 q[3];
    split1_np<3>(v[k], q);  // the first two pieces of a three-piece split ARE the two-piece split
    q0[k] = q[0];
    q1[k] = q[1];
    q2[k] = q[2];
  }
  *reinterpret_cast<bf16x8*>(w0 + i) = q0;
  *reinterpret_cast<bf16x8*>(w1 + i) = q1;
  if (w2 != nullptr) *reinterpret_cast<bf16x8*>(w2 + i) = q2;
}

__global__ __launch_bounds__(256) void weight_split_batched_kernel(const int64_t* __restrict__ jobs, int n_jobs) {
  __shared__ float tile[SPLIT_TILE][SPLIT_TILE + 1];
  // the job of this block: last job whose first block <= blockIdx.x (jobs are few: a short scan by every thread)
  int j = 0;
  for (int i = 1; i < n_jobs; ++i)
    if ((int64_t)blockIdx.x >= jobs[(int64_t)i * SCAN_SPLIT_JOB_WORDS + 9]) j = i;
  const int64_t* job = jobs + (int64_t)j * SCAN_SPLIT_JOB_WORDS;
  const float* __restrict__ w = reinterpret_cast<const float*>(job[0]);
  __bf16* __restrict__ wh = reinterpret_cast<__bf16*>(job[1]);
  __bf16* __restrict__ wl = reinterpret_cast<__bf16*>(job[2]);
  __bf16* __restrict__ w2 = reinterpret_cast<__bf16*>(job[10]);
  const int O = (int)job[3], T = (int)job[4], Cs = (int)job[5], mode = (int)job[6], rows = (int)job[7],
            Csw = (int)job[8];
  const int64_t blk = (int64_t)blockIdx.x - job[9];
  const int tid = threadIdx.x;
  if (mode == 0) {
    // plane rows are Csw (a multiple of 8) long, so an 8-element group never straddles two rows
    const int64_t total = (int64_t)rows * T * Csw;
    const int64_t i = blk * SPLIT_ELEMS_PER_BLOCK + (int64_t)tid * 8;
    if (i >= total) return;
    const int col = (int)(i % Csw);
    const int64_t rt = i / Csw;  // row * T + tt
    const float* src = w + rt * Cs + col;
    float v[8];
    if (col + 8 <= Cs && (Cs & 3) == 0) {
      const float4 a = *reinterpret_cast<const float4*>(src);
      const float4 b = *reinterpret_cast<const float4*>(src + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
      v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (col + k < Cs) ? src[k] : 0.f;
    }
    split8_store(v, wh, wl, w2, i);
    return;
  }
  // mode 1: block -> (tap tt of the OUTPUT, c tile, o tile)
  const int tiles_o = (Csw + SPLIT_TILE - 1) / SPLIT_TILE, tiles_c = (rows + SPLIT_TILE - 1) / SPLIT_TILE;
  const int to = (int)(blk % tiles_o);
  const int tc = (int)((blk / tiles_o) % tiles_c);
  const int tt = (int)(blk / ((int64_t)tiles_o * tiles_c));
  const int o0 = to * SPLIT_TILE, c0 = tc * SPLIT_TILE;
  {  // read: 16 lanes x float4 along c per weight row, 16 rows per pass
    const int cl = (tid & 15) * 4, ol = tid >> 4;
#pragma unroll
    for (int p = 0; p < SPLIT_TILE / 16; ++p) {
      const int o = o0 + ol + 16 * p, c = c0 + cl;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o < O && c < Cs) {
        const float* src = w + ((int64_t)o * T + (T - 1 - tt)) * Cs + c;
        if (c + 4 <= Cs && (Cs & 3) == 0) {
          a = *reinterpret_cast<const float4*>(src);
        } else {
          a.x = src[0];
          if (c + 1 < Cs) a.y = src[1];
          if (c + 2 < Cs) a.z = src[2];
          if (c + 3 < Cs) a.w = src[3];
        }
      }
      float* t = &tile[ol + 16 * p][cl];
      t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w;
    }
  }
  __syncthreads();
  {  // write: 8 lanes x 8 bf16 along o per plane row, 32 rows per pass; LDS address (o8 + k) * 65 + c: conflict-free
    const int o8 = (tid & 7) * 8, cl = tid >> 3;
#pragma unroll
    for (int p = 0; p < SPLIT_TILE / 32; ++p) {
      const int c = c0 + cl + 32 * p, o = o0 + o8;
      if (c < rows && o < Csw) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = tile[o8 + k][cl + 32 * p];
        split8_store(v, wh, wl, w2, ((int64_t)c * T + tt) * Csw + o);
      }
    }
  }
}

extern "C" int64_t scan_weight_split_job_blocks(int32_t O, int32_t T, int32_t Cs, int32_t mode, int32_t Csw) {
  if (mode == 0) return ((int64_t)O * T * Csw + SPLIT_ELEMS_PER_BLOCK - 1) / SPLIT_ELEMS_PER_BLOCK;
  const int64_t tiles_o = (Csw + SPLIT_TILE - 1) / SPLIT_TILE, tiles_c = (Cs + SPLIT_TILE - 1) / SPLIT_TILE;
  return tiles_o * tiles_c * T;
}

extern "C" int scan_weight_split_batched(const int64_t* jobs, int32_t n_jobs, int32_t job_words, int64_t total_blocks,
                                         void* stream) {
  // the table lives in device memory and cannot be inspected here: the caller states its record length, so that a table
  // built for another version of the layout is refused instead of being read past its records
  SCAN_CHECK_ARG(job_words == SCAN_SPLIT_JOB_WORDS, "weight_split_batched: records of %d words, this library reads %d",
                 job_words, SCAN_SPLIT_JOB_WORDS);
  SCAN_CHECK_ARG(jobs && n_jobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31),
                 "weight_split_batched: bad arguments (n_jobs=%d)", n_jobs);
  hipLaunchKernelGGL(weight_split_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), jobs,
                     n_jobs);
  SCAN_LAUNCH_CHECK("weight_split_batched");
  return 0;
}

// ------------------------------------------------------------------ SGD over many (p, g, buf) segments
struct SgdSegs {
  scan_sgd_segment_t s[SCAN_SGD_MAX_SEGMENTS];
  int64_t first_block[SCAN_SGD_MAX_SEGMENTS + 1];
  int n;
};
#define SGD_ELEMS_PER_BLOCK 4096

__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdSegs t, float momentum) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < SCAN_SGD_MAX_SEGMENTS; ++i)
    if (i < t.n && (int64_t)blockIdx.x >= t.first_block[i]) j = i;
  const scan_sgd_segment_t sg = t.s[j];
  const int64_t base = ((int64_t)blockIdx.x - t.first_block[j]) * SGD_ELEMS_PER_BLOCK;
  float* __restrict__ p = sg.p;
  const float* __restrict__ g = sg.g;
  float* __restrict__ buf = sg.buf;
#pragma unroll 4
  for (int k = 0; k < SGD_ELEMS_PER_BLOCK / 256; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i >= sg.n) break;
    // the arithmetic of sgd_kernel (pointwise.hip), operation for operation
    const float w = p[i];
    const float gg = g[i] + sg.wd * w;
    const float b = sg.first_step ? gg : momentum * buf[i] + gg;
    buf[i] = b;
    p[i] = w - sg.lr * b;
  }
}

extern "C" int scan_sgd_momentum_multi(const scan_sgd_segment_t* segs, int32_t n_segs, float momentum, void* stream) {
  SCAN_CHECK_ARG(segs && n_segs > 0 && n_segs <= SCAN_SGD_MAX_SEGMENTS, "sgd_momentum_multi: n_segs=%d out of 1..%d", n_segs,
                 SCAN_SGD_MAX_SEGMENTS);
  SgdSegs t;
  t.n = 0;
  int64_t blocks = 0;
  for (int i = 0; i < n_segs; ++i) {
    SCAN_CHECK_ARG(segs[i].n >= 0, "sgd_momentum_multi: segment %d has n < 0", i);
    if (segs[i].n == 0) continue;
    SCAN_CHECK_ARG(segs[i].p && segs[i].g && segs[i].buf, "sgd_momentum_multi: null pointer in segment %d", i);
    t.s[t.n] = segs[i];
    t.first_block[t.n] = blocks;
    blocks += (segs[i].n + SGD_ELEMS_PER_BLOCK - 1) / SGD_ELEMS_PER_BLOCK;
    ++t.n;
  }
  if (t.n == 0) return 0;
  t.first_block[t.n] = blocks;
  SCAN_CHECK_ARG(blocks < (1ll << 31), "sgd_momentum_multi: too many elements");
  hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), t, momentum);
  SCAN_LAUNCH_CHECK("sgd_momentum_multi");
  return 0;
}

// ------------------------------------------------------------------ CKA discriminator: stacked class-branch weights
// Per foreground class c the reference has conv3x3 (C+1 -> H) -> ReLU -> conv3x3 (H -> 1) on cat(x, act[c+1])
// (discriminator/fcos_head_discriminator_con.py:44-62,104-121).  All classes read the same x, so the Cf first convs
// are one conv over cat(x, act[1:]) with Cf*H output channels whose act-map input columns are block diagonal, and the
// Cf second convs are one conv Cf*H -> Cf with a block-diagonal weight.  stack writes those two weights in the
// kernels' [O][T][Cs] layout straight from the Cf parameter tensors; unstack accumulates the gradient of the stacked
// weights back into the Cf parameter gradients (element strides given, so NCHW- and channels-last-stored parameters
// both work).
struct CkaPtrs {
  scan_cka_branch_t b[SCAN_CKA_MAX_CLASSES];
};

__global__ __launch_bounds__(256) void cka_stack_kernel(CkaPtrs P, int Cf, int C, int H, int64_t s1o, int64_t s1c,
                                                        int64_t s1k, int64_t s2c, int64_t s2k, int Cs1, int Cs2,
                                                        float* __restrict__ w1, float* __restrict__ b1,
                                                        float* __restrict__ w2, float* __restrict__ b2) {
  const int64_t n1 = (int64_t)Cf * H * 9 * Cs1;   // w1 [Cf*H][9][Cs1]
  const int64_t n2 = (int64_t)Cf * 9 * Cs2;       // w2 [Cf][9][Cs2], Cs2 >= Cf*H
  const int64_t nb = (int64_t)Cf * H + Cf;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2 + nb;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int col = (int)(i % Cs1);
      const int64_t r = i / Cs1;
      const int k = (int)(r % 9);
      const int row = (int)(r / 9);
      const int c = row / H, o = row - c * H;
      float v = 0.f;
      if (col < C) v = P.b[c].w0[o * s1o + col * s1c + k * s1k];
      else if (col == C + c) v = P.b[c].w0[o * s1o + C * s1c + k * s1k];
      w1[i] = v;
    } else if (i < n1 + n2) {
      const int64_t q = i - n1;
      const int col = (int)(q % Cs2);
      const int64_t r = q / Cs2;
      const int k = (int)(r % 9);
      const int c = (int)(r / 9);
      float v = 0.f;
      if (col >= c * H && col < (c + 1) * H) v = P.b[c].w2[(col - c * H) * s2c + k * s2k];
      w2[q] = v;
    } else {
      const int q = (int)(i - n1 - n2);
      if (q < Cf * H) b1[q] = P.b[q / H].b0[q % H];
      else b2[q - Cf * H] = P.b[q - Cf * H].b2[0];
    }
  }
}

__global__ __launch_bounds__(256) void cka_unstack_kernel(CkaPtrs G, int Cf, int C, int H, int64_t s1o, int64_t s1c,
                                                          int64_t s1k, int64_t s2c, int64_t s2k, int Cs1, int Cs2,
                                                          const float* __restrict__ dw1, const float* __restrict__ db1,
                                                          const float* __restrict__ dw2, const float* __restrict__ db2,
                                                          int accumulate) {
  const int64_t n1 = (int64_t)Cf * H * 9 * (C + 1);  // one element of every w0 gradient
  const int64_t n2 = (int64_t)Cf * 9 * H;
  const int64_t nb = (int64_t)Cf * H + Cf;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2 + nb;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n1) {
      const int col = (int)(i % (C + 1));
      const int64_t r = i / (C + 1);
      const int k = (int)(r % 9);
      const int row = (int)(r / 9);
      const int c = row / H, o = row - c * H;
      if (dw1 != nullptr) {
        const float v = dw1[((int64_t)row * 9 + k) * Cs1 + (col < C ? col : C + c)];
        float* dst = const_cast<float*>(G.b[c].w0) + o * s1o + col * s1c + k * s1k;
        *dst = accumulate ? *dst + v : v;
      }
    } else if (i < n1 + n2) {
      const int64_t q = i - n1;
      const int h = (int)(q % H);
      const int64_t r = q / H;
      const int k = (int)(r % 9);
      const int c = (int)(r / 9);
      if (dw2 != nullptr) {
        const float v = dw2[((int64_t)c * 9 + k) * Cs2 + c * H + h];
        float* dst = const_cast<float*>(G.b[c].w2) + h * s2c + k * s2k;
        *dst = accumulate ? *dst + v : v;
      }
    } else {
      const int q = (int)(i - n1 - n2);
      if (q < Cf * H) {
        if (db1 != nullptr) {
          float* dst = const_cast<float*>(G.b[q / H].b0) + q % H;
          *dst = accumulate ? *dst + db1[q] : db1[q];
        }
      } else if (db2 != nullptr) {
        float* dst = const_cast<float*>(G.b[q - Cf * H].b2);
        *dst = accumulate ? *dst + db2[q - Cf * H] : db2[q - Cf * H];
      }
    }
  }
}

static int cka_check(const scan_cka_branch_t* br, int Cf, int C, int H, int Cs1, int Cs2, const char* who) {
  SCAN_CHECK_ARG(br && Cf >= 1 && Cf <= SCAN_CKA_MAX_CLASSES, "%s: Cf=%d out of 1..%d", who, Cf, SCAN_CKA_MAX_CLASSES);
  SCAN_CHECK_ARG(C > 0 && H > 0 && Cs1 >= C + Cf && Cs2 >= Cf * H, "%s: C=%d H=%d Cs1=%d Cs2=%d", who, C, H, Cs1, Cs2);
  for (int c = 0; c < Cf; ++c)
    SCAN_CHECK_ARG(br[c].w0 && br[c].b0 && br[c].w2 && br[c].b2, "%s: null pointer in branch %d", who, c);
  return 0;
}

extern "C" int scan_cka_stack_weights(const scan_cka_branch_t* branches, int32_t Cf, int32_t C, int32_t H,
                                      int64_t s1o, int64_t s1c, int64_t s1k, int64_t s2c, int64_t s2k, int32_t Cs1,
                                      int32_t Cs2, float* w1, float* b1, float* w2, float* b2, void* stream) {
  if (cka_check(branches, Cf, C, H, Cs1, Cs2, "cka_stack_weights")) return -1;
  SCAN_CHECK_ARG(w1 && b1 && w2 && b2, "cka_stack_weights: null output");
  CkaPtrs P;
  for (int c = 0; c < Cf; ++c) P.b[c] = branches[c];
  const int64_t n = (int64_t)Cf * H * 9 * Cs1 + (int64_t)Cf * 9 * Cs2 + (int64_t)Cf * H + Cf;
  hipLaunchKernelGGL(cka_stack_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), P, Cf, C, H, s1o, s1c,
                     s1k, s2c, s2k, Cs1, Cs2, w1, b1, w2, b2);
  SCAN_LAUNCH_CHECK("cka_stack_weights");
  return 0;
}

extern "C" int scan_cka_unstack_grads(const scan_cka_branch_t* grads, int32_t Cf, int32_t C, int32_t H, int64_t s1o,
                                      int64_t s1c, int64_t s1k, int64_t s2c, int64_t s2k, int32_t Cs1, int32_t Cs2,
                                      const float* dw1, const float* db1, const float* dw2, const float* db2,
                                      int32_t accumulate, void* stream) {
  if (cka_check(grads, Cf, C, H, Cs1, Cs2, "cka_unstack_grads")) return -1;
  CkaPtrs G;
  for (int c = 0; c < Cf; ++c) G.b[c] = grads[c];
  const int64_t n = (int64_t)Cf * H * 9 * (C + 1) + (int64_t)Cf * 9 * H + (int64_t)Cf * H + Cf;
  hipLaunchKernelGGL(cka_unstack_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), G, Cf, C, H, s1o, s1c,
                     s1k, s2c, s2k, Cs1, Cs2, dw1, db1, dw2, db2, accumulate);
  SCAN_LAUNCH_CHECK("cka_unstack_grads");
  return 0;
}
