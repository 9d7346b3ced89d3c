// Weight gradient of the 3x3 / stride-1 and 1x1 / stride-1|2 convolutions on v_mfma_f32_16x16x32_bf16 with split fp32
// operands (conv_split.h: NP = 2 pieces = "bf16x3", NP = 3 pieces = "bf16x6"):
//     dW[o][ky][kx][c] = sum_m dY[m][o] * X[m + (ky-1, kx-1)][c]
// A GEMM with K = pixels.  Both operands live in memory pixel-major ([pixel][channel]), but an MFMA lane needs 8
// consecutive k (pixels) of one channel: the tiles are staged in LDS in their natural [pixel][channel] layout and read with
// ds_read_b64_tr_b16, gfx950's transposing LDS read (each 16-lane group fetches a 4-pixel x 16-channel block and receives
// it channel-major), so no transpose pass exists.  A k-step is 32 pixels: lane (col = l & 15, kg = l >> 4) of the A operand
// (dY^T, rows = 16 output channels) and of the B operand (X, columns = 16 input channels) needs pixels 8 kg .. 8 kg + 7 of
// its column, i.e. two transposed reads on rows 8 kg + q and 8 kg + 4 + q.
//
// A workgroup computes the 128 (o) x 128 (c) tile for the THREE kx taps of one ky from a single staged dY chunk and one
// staged X row segment: K chunks are WK consecutive pixels of one image row, the X segment carries one halo pixel each
// side ([x0-1, x0+WK], zero outside the row), and tap kx is just "LDS row + kx" for the transposed reads.  Deterministic
// split-K over chunk ranges into fp32 slabs reduced in order (slab_bias_reduce_kernel); the splits of one tile group are
// placed on the same XCD so the re-reads of a chunk hit that XCD's L2.  The bias gradient (column sums of dY) rides along in
// the ky == 1, c-tile 0 workgroups, which already stream dY.
//
// Two kernels (the generations that survived; numbers and the ablations behind them in DESIGN.md section 3.1b):
//   conv_wgrad_v4_kernel   all eight waves stage and multiply in turn, one LDS stage (320-byte rows), scalar chunk walk,
//                          range-checked buffer loads issued one per MFMA block.  Serves the 1x1 convs (stride 1 and 2)
//                          and is the independent cross-check of the other kernel (bit-identical slabs for NP = 2).
//   conv_wgrad_v6_kernel   producer / consumer wave specialisation, two LDS stages (288-byte rows), one barrier per chunk.
//                          Every 3x3 launch.
#include "conv_split.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct ChunkTab {
  long long chunk_off[SCAN_MAX_LEVELS + 1];
  int segs[SCAN_MAX_LEVELS];
};

__device__ __forceinline__ bf16x8 tr_read8(const __bf16* p0, const __bf16* p1) {
  auto q0 = (__attribute__((address_space(3))) s16x4*)(p0);
  auto q1 = (__attribute__((address_space(3))) s16x4*)(p1);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(q1);
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8, r);
}

// the same from LDS byte addresses (kernels that keep "lane base + immediate" addressing in their own hands)
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ bf16x8 tr_read8_at(unsigned a0, unsigned a1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a1);
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return __builtin_bit_cast(bf16x8, r);
}

// the piece products of one (dY fragment set, X fragment), smallest terms first; within a magnitude class the dY piece
// index descends (the order the two-piece kernels have always used: lo * hi, hi * lo, hi * hi)
template <int NP, int TO, int TOMAX>
__device__ __forceinline__ void wgrad_pieces(const bf16x8 (&a)[NP][TO], const bf16x8 (&b)[NP], f32x4v (&acc)[TOMAX]) {
#pragma unroll
  for (int s = NP - 1; s >= 0; --s)
#pragma unroll
    for (int i = s; i >= 0; --i)
#pragma unroll
      for (int to = 0; to < TO; ++to) acc[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][to], b[s - i], acc[to], 0, 0, 0);
}

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

// ------------------------------------------------------------------------------------------------
// conv_wgrad_v4_kernel.  LDS rows are 320 B (256 B data + 64 B pad); the two 16-lane groups of a half wave read rows 8
// apart in the same 16 columns, which with that pitch fall on the same banks, so the 32-byte column group of a row is
// XOR-ed with bit 3 of the row index (applied by the staging writes and by every lane's read address).
//   * the chunk position (level, image, row, segment) is wave-uniform state advanced by a few scalar instructions per
//     chunk (the divisions run once per workgroup, quotients pinned to SGPRs with readfirstlane);
//   * both operands are fetched with buffer loads whose descriptor (base = first pixel of the chunk's row segment,
//     num_records = bytes up to its last valid pixel) is rebuilt per chunk from scalars: the hardware range check
//     returns zeros beyond the row end / for rows outside the image (num_records = 0), so no load is predicated;
//   * a lane's byte offsets inside a chunk are the same for every chunk and live in registers;
//   * the loads of the next chunk are issued one per MFMA block inside the MFMA phase (issued together before the
//     barrier, the 72 wave-instructions of a workgroup queue up in the texture-address path and the MFMA phase starts
//     late: profiles/r03_wgrad_v4_ab.txt).
// Needs Ns % 4 == 0, Cs % 4 == 0 and row segments below 2 GiB.
// ------------------------------------------------------------------------------------------------
#define WROW 160  // bf16 elements per LDS row
#define WK 64     // pixels per K chunk
#define WBUF(NP, KX) ((WK + WK + (KX) - 1) * (NP) * WROW)  // bf16 elements of the stage: dY planes [NP][WK], X planes [NP][WK+KX-1]

__device__ __forceinline__ int wsw(int row, int col) {  // bf16 element offset of (row, col) in a swizzled stage plane
  return row * WROW + ((((col >> 4) ^ ((row >> 3) & 1)) << 4) | (col & 15));
}

// MFMA work of one staged chunk for one wave, with a hook after every (k-step, tap, column tile) block of MFMAs.
// TO: live 16-row o tiles of this wave (4, 2 or 1)
template <int NP, int TO, int KX, int TOMAX, typename F>
__device__ __forceinline__ void wgrad_mma_v4(const __bf16* A, const __bf16* B, int row_lane, int col4, int a_col, int b_col,
                                             f32x4v (&acc)[KX][2][TOMAX], F&& hook) {
  constexpr int APL = WK * WROW, BPL = (WK + KX - 1) * WROW;
  int blk = 0;
#pragma unroll
  for (int s = 0; s < WK / 32; ++s) {
    // dY rows of this lane: r0 = 32 s + 8 kg + q and r0 + 4 (bit 3 of both = kg & 1: one swizzle per lane)
    const int ra0 = 32 * s + row_lane, ra1 = ra0 + 4;
    bf16x8 a[NP][TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const int c = a_col + 16 * t + col4;
      const int o0 = wsw(ra0, c), o1 = wsw(ra1, c);
#pragma unroll
      for (int p = 0; p < NP; ++p) a[p][t] = tr_read8(A + p * APL + o0, A + p * APL + o1);
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
      const int rb0 = ra0 + kx, rb1 = ra1 + kx;  // X row j <-> pixel x0 - HALO + j: tap kx is a row shift
#pragma unroll
      for (int tc = 0; tc < 2; ++tc) {
        const int c = b_col + 16 * tc + col4;
        const int o0 = wsw(rb0, c), o1 = wsw(rb1, c);
        bf16x8 b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = tr_read8(B + p * BPL + o0, B + p * BPL + o1);
        wgrad_pieces<NP, TO, TOMAX>(a, b, acc[kx][tc]);
        hook(blk);
        ++blk;
      }
    }
  }
}

template <int NP, int KX, int S>
__global__ __launch_bounds__(512, 2) void conv_wgrad_v4_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, scan_pyramid_t xd) {
  constexpr int NT = 512;
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
    seg = __builtin_amdgcn_readfirstlane((int)(r - rowl * segs));
    n = __builtin_amdgcn_readfirstlane((int)(rowl / H));
    y = __builtin_amdgcn_readfirstlane((int)(rowl - (long long)(rowl / H) * H));
  }

  float4 ra[NA], rb[NB];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  __amdgpu_buffer_rsrc_t ra_src, rb_src;
  bool left_edge = false;
  // descriptors of the chunk at (lvl, n, y, seg): scalar work only.  live = false: zero records, every load of the
  // "chunk" returns zeros without touching memory (the loads are issued unconditionally, also behind the last chunk,
  // so that the MFMA phase has no control flow in it)
  auto prepare_loads = [&](bool live) {
    const int x0 = seg * WK;
    const long long rowbase = row0 + ((long long)n * H + y) * W;
    const int kmax = (W - x0 < WK) ? W - x0 : WK;
    ra_src = uniform_rsrc_b(dy + (rowbase + x0) * Ns, live ? kmax * Ns * 4 : 0);
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
    rb_src = uniform_rsrc_b(bbase, live ? nrec : 0);
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
  __bf16* const As = sm;                      // [NP][WK] rows
  __bf16* const Bs = sm + NP * WK * WROW;     // [NP][WK + KX - 1] rows
  auto store_chunk = [&]() {
    bf16x4 pc[NP];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int off = wsw(rr + RG * i, 4 * q4);
      split4_np<NP>(ra[i], pc);
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(As + p * WK * WROW + off) = pc[p];
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
        split4_np<NP>(rb[i], pc);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(Bs + p * (WK + KX - 1) * WROW + off) = pc[p];
      }
    }
  };

  const int wm = wid % WO, wn = wid / WO;
  const int lr = lane & 15, kg = lane >> 4;
  const int row_lane = 8 * kg + (lr >> 2), col4 = 4 * (lane & 3);
  const int a_col = wm * (16 * TOMAX), b_col = wn * 32;
  const bool c_act = c0 + b_col < Cs;
  const int o_left = Nout - (o0 + a_col);

  f32x4v acc[KX][2][TOMAX];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int b = 0; b < TOMAX; ++b) acc[a][c][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (ch_begin < ch_end) {
    prepare_loads(true);
#pragma unroll
    for (int k = 0; k < NA + NB; ++k) issue_one(k);
  }
  for (long long ch = ch_begin; ch < ch_end; ++ch) {
    store_chunk();
    const bool more = ch + 1 < ch_end;
    if (more) advance();
    prepare_loads(more);
    __syncthreads();
    // one load after each of the first NA + NB MFMA blocks (12 blocks for the 3x3, 4 for the 1x1: the rest follow the
    // last block); the sched_barrier keeps the compiler from gathering them at either end of the phase
    constexpr int NBLK = (WK / 32) * KX * 2;
    auto hook = [&](int blk) {
#pragma unroll
      for (int k = 0; k < NA + NB; ++k)
        if (k == blk || (blk == NBLK - 1 && k >= NBLK)) issue_one(k);
      __builtin_amdgcn_sched_barrier(0);
    };
    if (c_act && o_left > 32) {
      wgrad_mma_v4<NP, TOMAX, KX, TOMAX>(As, Bs, row_lane, col4, a_col, b_col, acc, hook);
    } else if (c_act && o_left > 16) {
      wgrad_mma_v4<NP, 2, KX, TOMAX>(As, Bs, row_lane, col4, a_col, b_col, acc, hook);
    } else if (c_act && o_left > 0) {
      wgrad_mma_v4<NP, 1, KX, TOMAX>(As, Bs, row_lane, col4, a_col, b_col, acc, hook);
    } else {
#pragma unroll
      for (int k = 0; k < NA + NB; ++k) issue_one(k);
    }
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
          if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][tc][to][r];
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
// conv_wgrad_v6_kernel: producer / consumer wave specialisation.  What is left of a K chunk in the kernel above beside its
// MFMAs is the fp32 -> bf16 split and the LDS writes, which all eight waves execute together between two barriers while the
// matrix pipe idles.  Here a workgroup has 12 waves: waves 0..7 ONLY read fragments and issue MFMAs (2 x 4 wave grid, two
// per SIMD), waves 8..11 -- one per SIMD, at s_setprio 3 -- ONLY stage: they walk the chunks (scalar state, range-checked
// buffer loads), split and write the NEXT chunk into the other of two LDS stages while the consumers multiply the current
// one, and refill each half of their single register set right behind its LDS writes.  One barrier per chunk.  <= 168
// registers per lane (three waves per SIMD): the accumulators live in the consumer branch only, the staging registers in
// the producer branch only.
//
// LDS image: two stages must fit, and the transposed-read addresses must stay "base + immediate".  Rows are 288 bytes
// (256 + 32 pad): consecutive rows start 8 banks apart, so the four pixel rows a 16-lane group of ds_read_b64_tr_b16 touches
// (32 bytes each) cover 32 banks; the other lane group of the same half wave reads rows 8 further (64 banks = 0 further)
// and is moved to the other 32 banks by swapping the two 128-byte halves of a row when bit 3 of the row index is set.
//
// K chunk: WKC = 64 pixels with two pieces (2 x 74,880 B of LDS), 32 pixels with three (2 x 57,024 B; three planes of a
// 64-pixel chunk would need 2 x 112 KB) -- the MFMA work per chunk and barrier is the same (twice the products on half
// the pixels), the producers convert half the elements into 1.5x the planes.
// ------------------------------------------------------------------------------------------------
#ifndef SCAN_WG_TCHAIN
#define SCAN_WG_TCHAIN 1  // three pieces, 32 x 64 wave tile: per-step temporary accumulator (wgrad_mma_v6)
#endif
#ifndef SCAN_WG_PIPE
#define SCAN_WG_PIPE 1  // three pieces, 32 x 64 wave tile: hand-pipelined consumer loop (wgrad_mma_v6_pipe)
#endif
#ifndef SCAN_WG_TG
#define SCAN_WG_TG 0  // > 0: column tiles per block of the temporary-accumulator form (fewer temporaries)
#endif
#ifndef SCAN_WG_SB
#define SCAN_WG_SB 1  // scheduling fence behind every (tap, column tile) block of the consumers' MFMAs, see wgrad_mma_v6
#endif
#define W6ROW 144  // bf16 elements per LDS row
#define W6STAGE(NP, WKC, KX) (((WKC) + (WKC) + (KX) - 1) * (NP) * W6ROW)  // bf16 elements per stage
__device__ __forceinline__ int wsw6(int row, int col) { return row * W6ROW + (col ^ (((row >> 3) & 1) << 6)); }

// Fragment addresses as "lane base + compile-time offset": the half-row swap of wsw6 is applied to the lane's base column
// only (adding 16 t or 16 tc afterwards never carries into bit 6: the base columns are 64 wm + col4 and 32 wn + col4,
// col4 < 16), and a row offset is an immediate wherever it cannot change bit 3 of the row -- true for the rows
// 8 kg + q (+ kx) and 8 kg + q + 4, q = (lane & 15) >> 2; only rows 8 kg + q + 4 + kx, kx = 1, 2 may cross into the next
// group of eight and get bases of their own.  Four address registers instead of one per (tile, tap, row group).
struct W6Lane {  // LDS BYTE addresses of stage 0 (the consumer loop flips them between the stages in place)
  unsigned a;      // (row 8 kg + q, column a_col + col4) of dY plane 0: rows + 4, tiles + 16 t and planes by immediate
  unsigned b;      // (row 8 kg + q, column b_col + col4) of X plane 0: rows + kx, + 4 (kx = 0), tiles + 16 tc, planes by immediate
  unsigned b1[2];  // (row 8 kg + q + 4 + kx, same column), kx = 1, 2
};
template <int NP, int WKC>
__device__ __forceinline__ W6Lane w6_lane(unsigned lds_base, int row_lane, int col4, int a_col, int b_col) {
  W6Lane w;
  const unsigned xb = lds_base + 2u * (NP * WKC * W6ROW);
  w.a = lds_base + 2u * wsw6(row_lane, a_col + col4);
  w.b = xb + 2u * wsw6(row_lane, b_col + col4);
  w.b1[0] = xb + 2u * wsw6(row_lane + 5, b_col + col4);
  w.b1[1] = xb + 2u * wsw6(row_lane + 6, b_col + col4);
  return w;
}

// TO x TC live 16 x 16 tiles of this wave (of TOMAX x TCMAX).  The X fragments of G column tiles are read together so that
// G * TO >= 4 independent accumulators separate two MFMAs on the same one.
template <int NP, int WKC, int TO, int TC, int KX, int TOMAX, int TCMAX, bool TCHAIN>
__device__ __forceinline__ void wgrad_mma_v6(const W6Lane& w, f32x4v (&acc)[KX][TCMAX][TOMAX]) {
  constexpr unsigned APL = 2 * WKC * W6ROW, BPL = 2 * (WKC + KX - 1) * W6ROW, ROWB = 2 * W6ROW;  // bytes
  constexpr int G0 = (TO >= 4 || TC == 1) ? 1 : (TO == 2 ? (TC >= 2 ? 2 : 1) : (TC >= 4 ? 4 : TC));
  constexpr int G = (TCHAIN && SCAN_WG_TG > 0 && SCAN_WG_TG < G0) ? SCAN_WG_TG : G0;
  static_assert(TC % G == 0, "column tiles per block");
#pragma unroll
  for (int s = 0; s < WKC / 32; ++s) {
    bf16x8 a[NP][TO];
#pragma unroll
    for (int t = 0; t < TO; ++t) {
      const unsigned o0 = w.a + 32 * s * ROWB + 32 * t, o1 = o0 + 4 * ROWB;
#pragma unroll
      for (int p = 0; p < NP; ++p) a[p][t] = tr_read8_at(o0 + p * APL, o1 + p * APL);
    }
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
#pragma unroll
      for (int tc0 = 0; tc0 < TC; tc0 += G) {
        bf16x8 b[G][NP];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          // X row j <-> pixel x0 - HALO + j: tap kx is a row shift
          const unsigned o0 = w.b + (32 * s + kx) * ROWB + 32 * (tc0 + g);
          const unsigned o1 = (kx == 0 ? w.b + 4 * ROWB : w.b1[kx - 1]) + 32 * s * ROWB + 32 * (tc0 + g);
#pragma unroll
          for (int p = 0; p < NP; ++p) b[g][p] = tr_read8_at(o0 + p * BPL, o1 + p * BPL);
        }
        // piece products, smallest first; within a magnitude class the dY piece index descends (wgrad_pieces)
        if constexpr (TCHAIN) {
          // the six products of a 32-pixel step are summed in a temporary that starts at zero and is added to the running
          // accumulator ONCE: one rounding at the accumulator's magnitude per step instead of six (the temporary's own
          // roundings are relative to a 32-term sum).  Over the 8,192-pixel chains of a split-K slab that is what separates
          // the weight gradient's distance from fp64 from the exact fp32-MFMA kernel's.
          f32x4v tmp[G][TO];
#pragma unroll
          for (int g = 0; g < G; ++g)
#pragma unroll
            for (int to = 0; to < TO; ++to) tmp[g][to] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int d = NP - 1; d >= 0; --d)
#pragma unroll
            for (int i = d; i >= 0; --i)
#pragma unroll
              for (int g = 0; g < G; ++g)
#pragma unroll
                for (int to = 0; to < TO; ++to)
                  tmp[g][to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][to], b[g][d - i], tmp[g][to], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < G; ++g)
#pragma unroll
            for (int to = 0; to < TO; ++to) acc[kx][tc0 + g][to] += tmp[g][to];
        } else {
#pragma unroll
          for (int d = NP - 1; d >= 0; --d)
#pragma unroll
            for (int i = d; i >= 0; --i)
#pragma unroll
              for (int g = 0; g < G; ++g)
#pragma unroll
                for (int to = 0; to < TO; ++to)
                  acc[kx][tc0 + g][to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][to], b[g][d - i], acc[kx][tc0 + g][to], 0, 0, 0);
        }
#if SCAN_WG_SB
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
  }
}

// The same chunk, software-pipelined by hand (three pieces, 32 x 64 wave tile): a block is ONE column tile of one tap
// (2 x 6 MFMAs on a temporary pair); the X fragments of block i + 1 are requested before the MFMAs of block i are issued
// and the scheduling fence keeps that order -- the fence-per-block form above waited for its fragments in front of every
// block with the other wave of the SIMD mostly in the same state.  12 more registers for the second fragment set, 8 for the
// temporaries: 162 at the 168-register cap, no scratch.  The resident dY fragments are requested tile by tile so that the
// first block can start after 12 of the 24 transposed reads behind the barrier.
// Round 5, measured and NOT adopted (profiles/r05_wgrad_pipe2_ab.txt): the block as two back-to-back six-product chains (tile
// 0, then tile 1) with each temporary's final add in the shadow of the other tile's MFMAs -- no "s_nop 6 + adds" stall at the
// end of a block in the ISA, four registers fewer, bit-identical -- runs 3-4 % SLOWER on every shape (conv3_x 2,510 -> 2,600 us,
// conv4_x 2,430 -> 2,535, same box, A B A B): the two interleaved chains of this form keep the pipe busier than one dependent
// chain at a time, and the stall at the block end is covered by the SIMD's other consumer wave.
// Second form (profiles/r05_wgrad_c256_probe.txt): the interleaved chains kept, two temporary sets, the fold of block i - 1 issued
// behind the first four MFMAs of block i (no s_nop in front of the adds): 168 registers + one 8-byte spill in the loop,
// 3.5 % SLOWER (conv3_x 2,500 -> 2,590 us); on eight consumer waves alone with 256 registers it changes nothing (2,304 us
// either way): the fold's wait is not what the temporary accumulator costs.
template <int NP, int WKC, int TO, int TC, int KX, int TOMAX, int TCMAX>
__device__ __forceinline__ void wgrad_mma_v6_pipe(const W6Lane& w, f32x4v (&acc)[KX][TCMAX][TOMAX]) {
  static_assert(WKC == 32, "one 32-pixel step per chunk");
  constexpr unsigned APL = 2 * WKC * W6ROW, BPL = 2 * (WKC + KX - 1) * W6ROW, ROWB = 2 * W6ROW;  // bytes
  constexpr int NBLK = KX * TC;
  bf16x8 a[NP][TO];
  bf16x8 b[2][NP];
  auto read_b = [&](int blk, bf16x8 (&dst)[NP]) {
    const int kx = blk / TC, tc = blk % TC;
    const unsigned o0 = w.b + kx * ROWB + 32 * tc;
    const unsigned o1 = (kx == 0 ? w.b + 4 * ROWB : w.b1[kx - 1]) + 32 * tc;
#pragma unroll
    for (int p = 0; p < NP; ++p) dst[p] = tr_read8_at(o0 + p * BPL, o1 + p * BPL);
  };
#pragma unroll
  for (int p = 0; p < NP; ++p) a[p][0] = tr_read8_at(w.a + p * APL, w.a + 4 * ROWB + p * APL);
  read_b(0, b[0]);
#pragma unroll
  for (int t = 1; t < TO; ++t)
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p][t] = tr_read8_at(w.a + 32 * t + p * APL, w.a + 32 * t + 4 * ROWB + p * APL);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const int kx = blk / TC, tc = blk % TC;
    if (blk + 1 < NBLK) read_b(blk + 1, b[(blk + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4v tmp[TO];
#pragma unroll
    for (int to = 0; to < TO; ++to) tmp[to] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = NP - 1; d >= 0; --d)
#pragma unroll
      for (int i = d; i >= 0; --i)
#pragma unroll
        for (int to = 0; to < TO; ++to)
          tmp[to] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][to], b[blk & 1][d - i], tmp[to], 0, 0, 0);
#pragma unroll
    for (int to = 0; to < TO; ++to) acc[kx][tc][to] += tmp[to];
    __builtin_amdgcn_sched_barrier(0);
  }
}

// TOM x TCW: 16 x 16 tiles of a consumer wave along o and c (8 waves cover the 128 x 128 tile: 4 x 2 -> 2 (o) x 4 (c)
// waves, 2 x 4 -> 4 (o) x 2 (c) waves).  Either way 96 accumulator registers; the resident dY fragments are NP * TOM * 4
// registers, the X fragments are re-read per tap: 4 x 2 reads fewer fragments per MFMA (30 transposed reads per 144
// MFMAs with three pieces), 2 x 4 keeps 24 instead of 48 registers resident (42 reads) and leaves room to prefetch.
#if defined(SCAN_EXP_WGRAD_C256) && SCAN_EXP_WGRAD_C256 == 1
// TIMING EXPERIMENT (make exp_wgrad_c256 M=1|2, WRONG results): the consumers alone on two LDS stages they fill once --
// 1: eight waves with 256 registers, temporary accumulator + pipelined loop on either wave tile; 2: the shipped loops beside
// four idle producer waves (168 registers)
#define W6_BOUNDS __launch_bounds__(512, 2)
#define W6_THREADS 512
#else
#define W6_BOUNDS __launch_bounds__(768, 3)
#define W6_THREADS 768
#endif
template <int NP, int WKC, int KX, int TOM, int TCW>
__global__ W6_BOUNDS void conv_wgrad_v6_kernel(
    const float* __restrict__ x, scan_pyramid_t d, int Cs, const float* __restrict__ dy, int Nout, int Ns,
    float* __restrict__ slab, float* __restrict__ bias_slab, ChunkTab ct, int n_tiles, int c_tiles,
    int chunks_per_split, int splits, int prio) {
  constexpr int HALO = KX / 2, T = KX * KX;
  constexpr int TOMAX = TOM, TCMAX = TCW;             // 16 x 16 tiles per consumer wave along o and c
  constexpr int WOC = 128 / (16 * TOM);               // consumer waves along o (x 128 / (16 * TCW) along c = 8)
  static_assert(WOC * (128 / (16 * TCW)) == 8, "eight consumer waves cover the 128 x 128 tile");
  constexpr int PRG = 8;                              // pixel-row groups of the 256 producer threads
  constexpr int NA = WKC / PRG;                       // dY float4 per producer thread per chunk
  constexpr int NB = (WKC + KX - 1 + PRG - 1) / PRG;  // X float4 per producer thread per chunk
  constexpr int STAGE = W6STAGE(NP, WKC, KX);
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
    const unsigned offb_last = (rr + PRG * (NB - 1) < WKC + KX - 1) ? offb : BAD;  // rows of the last group beyond the halo
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
      const int x0 = seg * WKC;
      const long long rowbase = row0 + ((long long)n * H + y) * W;
      const int kmax = (W - x0 < WKC) ? W - x0 : WKC;
      ra_src = uniform_rsrc_b(dy + (rowbase + x0) * Ns, live ? kmax * Ns * 4 : 0);
      const int yy = y + ky - HALO;
      const int jmax = (W - x0 + HALO < WKC + KX - 1) ? W - x0 + HALO : WKC + KX - 1;
      const float* bbase = x + (rowbase + (long long)(ky - HALO) * W + x0 - HALO) * Cs;  // never dereferenced outside
      rb_src = uniform_rsrc_b(bbase, (live && yy >= 0 && yy < H) ? jmax * Cs * 4 : 0);
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
      __bf16* As = sm + stage * STAGE;
      bf16x4 pc[NP];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int off = wsw6(rr + PRG * i, 4 * q4);
#ifdef SCAN_EXP_WGRAD_NOSPLIT  // TIMING EXPERIMENT (make exp_wgrad_nosplit, never in libscan_hip.so): no conversion work
        for (int p = 0; p < NP; ++p) pc[p] = __builtin_bit_cast(bf16x4, make_float2(ra[i].x, ra[i].y));
#else
        split4_np<NP>(ra[i], pc);
#endif
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(As + p * WKC * W6ROW + off) = pc[p];
        if (do_bias) {
          bsum.x += ra[i].x;
          bsum.y += ra[i].y;
          bsum.z += ra[i].z;
          bsum.w += ra[i].w;
        }
      }
    };
    auto store_b = [&](int stage) {
      __bf16* Bs = sm + stage * STAGE + NP * WKC * W6ROW;
      bf16x4 pc[NP];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int j = rr + PRG * i;
        if (j < WKC + KX - 1) {
          const int off = wsw6(j, 4 * q4);
#ifdef SCAN_EXP_WGRAD_NOSPLIT
          for (int p = 0; p < NP; ++p) pc[p] = __builtin_bit_cast(bf16x4, make_float2(rb[i].x, rb[i].y));
#else
          split4_np<NP>(rb[i], pc);
#endif
#pragma unroll
          for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(Bs + p * (WKC + KX - 1) * W6ROW + off) = pc[p];
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
#if defined(SCAN_EXP_WGRAD_PROD)
      // TIMING EXPERIMENT (make exp_wgrad_prod_<mask>, WRONG results): which part of the producers' work costs the consumers
      // their issue slots -- bit 0: the loads, bit 1: the LDS writes (of unconverted bits unless the split is compiled in),
      // bit 3: the chunk walk and descriptors
      const int stage = (k + 1) & 1;
      const bool more = k + 2 < nch;
      auto keep_a = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w));
      };
      auto keep_b = [&]() {
#pragma unroll
        for (int i = 0; i < NB; ++i) asm volatile("" ::"v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
      };
      if (SCAN_EXP_WGRAD_PROD & 8) {
        if (more) advance();
        prepare(more);
      }
      if (SCAN_EXP_WGRAD_PROD & 2) store_a(stage); else keep_a();
      if (SCAN_EXP_WGRAD_PROD & 1) load_a();
      if (SCAN_EXP_WGRAD_PROD & 2) store_b(stage); else keep_b();
      if (SCAN_EXP_WGRAD_PROD & 1) load_b();
#elif !defined(SCAN_EXP_WGRAD_NOPROD) && !defined(SCAN_EXP_WGRAD_C256)  // TIMING EXPERIMENT (make exp_wgrad_noprod): the producers only attend the barriers
      const int stage = (k + 1) & 1;  // chunk k + 1 is in the registers; chunk k + 2 follows it
      const bool more = k + 2 < nch;
      if (more) advance();
      prepare(more);
      store_a(stage);
      load_a();
      store_b(stage);
      load_b();
#endif
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

  // ================================================================= consumers: WOC (o) x 8 / WOC (c) waves
  const int wm = wid % WOC, wn = wid / WOC;
  const int lr = lane & 15, kg = lane >> 4;
  const int row_lane = 8 * kg + (lr >> 2), col4 = 4 * (lane & 3);
  const int a_col = wm * (16 * TOMAX), b_col = wn * (16 * TCMAX);
  const int c_left = Cs - (c0 + b_col);
  const int o_left = Nout - (o0 + a_col);

  f32x4v acc[KX][TCMAX][TOMAX];
#pragma unroll
  for (int a = 0; a < KX; ++a)
#pragma unroll
    for (int c = 0; c < TCMAX; ++c)
#pragma unroll
      for (int b = 0; b < TOMAX; ++b) acc[a][c][b] = f32x4v{0.f, 0.f, 0.f, 0.f};

#ifdef SCAN_EXP_WGRAD_C256
  {  // both stages filled once with bf16 values of the operands' magnitude
    const long long nx = (long long)ct.chunk_off[d.n_levels] * 16 * Cs;
    for (int i = tid; i < STAGE; i += 512) {
      const float2 v = *reinterpret_cast<const float2*>(x + (((long long)i + (long long)blockIdx.x * STAGE) % (nx / 2)) * 2);
      bf16x2 h;
      h[0] = (__bf16)v.x;
      h[1] = (__bf16)v.y;
      *reinterpret_cast<bf16x2*>(sm + 2 * i) = h;
    }
  }
#endif
  __syncthreads();  // stage 0 is complete
  // one K loop per live-tile count (wave-uniform; dead tiles: third c tile of Cin = 264 / 268, Cout = 8 / 5 / 1 heads):
  // inside one loop the compiler would keep the fragment addresses of all variants in registers across it, which at 168
  // registers per lane spills.
  // The lane's four fragment addresses are LDS byte addresses that carry the stage offset themselves and flip between the
  // two stages by +-STAGE at the end of an iteration (four in-place adds): formed as "stage base + lane offset" inside the
  // loop they were four MORE live registers.
  auto run = [&](auto to_tag, auto tc_tag) {
    constexpr int TO = decltype(to_tag)::value, TC = decltype(tc_tag)::value;
    W6Lane wl = w6_lane<NP, WKC>((unsigned)(uintptr_t)(lds_ptr_t)sm, row_lane, col4, a_col, b_col);
    unsigned flip = 2u * STAGE;  // bytes; +-: unsigned wrap-around is the subtraction
    for (int k = 0; k < nch; ++k) {
      if constexpr (TO > 0) {
#if defined(SCAN_EXP_WGRAD_C256) && SCAN_EXP_WGRAD_C256 == 1
        if constexpr (NP == 3)
#else
        if constexpr (SCAN_WG_PIPE && SCAN_WG_TCHAIN && NP == 3 && TOM == 2)
#endif
          wgrad_mma_v6_pipe<NP, WKC, TO, TC, KX, TOMAX, TCMAX>(wl, acc);
        else
          wgrad_mma_v6<NP, WKC, TO, TC, KX, TOMAX, TCMAX, SCAN_WG_TCHAIN && NP == 3 && TOM == 2>(wl, acc);
      }
      wl.a += flip;
      wl.b += flip;
      wl.b1[0] += flip;
      wl.b1[1] += flip;
      flip = 0u - flip;
      __syncthreads();  // done with this stage; the other one is complete
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using ITO = std::integral_constant<int, TOMAX>;
  using ITC = std::integral_constant<int, TCMAX>;
  // live column tiles: all of them, or one when at most 16 channels are left (the 8 / 12 channels of a 264- / 268-channel
  // input beyond its two full 128-wide tiles)
  auto run_to = [&](auto tc_tag) {
    if (o_left > 16 * (TOMAX / 2) && TOMAX > 2)
      run(ITO{}, tc_tag);
    else if (o_left > 16)
      run(I2{}, tc_tag);
    else if (o_left > 0)
      run(I1{}, tc_tag);
    else
      run(I0{}, tc_tag);
  };
  if (c_left <= 0)
    run(I0{}, I1{});
  else if (c_left <= 16)
    run_to(I1{});
  else
    run_to(ITC{});

  float* out = slab + (long long)split * Nout * T * Cs;
#pragma unroll
  for (int kx = 0; kx < KX; ++kx)
#pragma unroll
    for (int to = 0; to < TOMAX; ++to)
#pragma unroll
      for (int tc = 0; tc < TCMAX; ++tc) {
        const int c = c0 + b_col + 16 * tc + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + a_col + 16 * to + 4 * kg + r;
          if (o < Nout && c < Cs) out[((long long)o * T + ky * KX + kx) * Cs + c] = acc[kx][tc][to][r];
        }
      }
  __syncthreads();  // pairs with the producers' bias-reduction barrier
}

// scan_tune "wgrad_tile": consumer wave tile of the producer / consumer kernel: 0 = 64 (o) x 32 (c), 1 = 32 (o) x 64 (c),
// 2 (default; any other value too) = by piece count: three pieces 32 x 64 with a temporary accumulator per 32-pixel step (the six
// products of a step are summed from zero and added once: 0.23-0.5x the fp32-MFMA kernel's distance from an fp64 weight gradient),
// two pieces 64 x 32 (within noise of the other; same results bit for bit).  Three pieces on tile 0 add every piece product straight
// into the running accumulator (1.0-2.1x that distance; tests/test_gpu_kernels.py::test_wgrad_full_size_elementwise measures both).
// Round 6 tried tile 0 as the default for three pieces: the whole golden suite (273 tests) is green on it and on Gaussian operands it
// is 4-5 % faster (conv3_x 2576 -> 2442 us, tools/conv_bench.py) -- but in the training step, on post-ReLU activations and sparse
// gradients, the SAME-BOX A/B shows nothing: 0.5406 / 0.5376 / 0.5396 ms per launch with the temporary against 0.5436 / 0.5396 /
// 0.5443 without, 92.2 ms per step either way (profiles/r06_wgrad_tile_ab.txt).  No speed for the accuracy: the temporary stays.
int g_scan_wgrad_tile = 2;  // 2 = by piece count
// scan_tune "wgrad_v6": 1 (default) = the 3x3 launches take the producer / consumer kernel, 0 = conv_wgrad_v4_kernel
int g_scan_wgrad_v6 = 1;
// scan_tune "wgrad_prio": 1 = the producer waves run at s_setprio 3, 0 (default) = at the consumers' priority.  Round 3 ran them
// raised (the staging wave otherwise reached the barrier last); with the packed conversion and the loads issued early the
// raised priority only takes issue slots from the MFMA waves -- three pieces: conv3_x 2630 -> 2532 us, conv4_x 2532 -> 2468,
// towers 464 -> 453, class branches 3266 -> 3204; two pieces: 0-2 % (conv3_x 1338 -> 1309).  Same results bit for bit.
int g_scan_wgrad_prio = 0;
// scan_tune "wgrad_wgs": workgroups a weight-gradient launch aims at (tiles x splits), see the sweep quoted in wgrad_plan
int g_scan_wgrad_wgs = 768;

// weight-slab reduction (float4 columns, splits summed in order, in fp64: the kernel is bound by the slab reads, the wider
// adds are free and take the reduction's own rounding out of the result) + bias-slab reduction in the extra last block
__global__ __launch_bounds__(256) void slab_bias_reduce_kernel(const float* __restrict__ slab, int splits, int64_t n,
                                                               float* __restrict__ dw, const float* __restrict__ bs,
                                                               int nb, float* __restrict__ db, int accumulate) {
  const int wblocks = gridDim.x - (db ? 1 : 0);
  if ((int)blockIdx.x < wblocks) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)wblocks * blockDim.x) {
      double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
      for (int k = 0; k < splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(slab + (int64_t)k * n)[i];
        sx += (double)v.x;
        sy += (double)v.y;
        sz += (double)v.z;
        sw += (double)v.w;
      }
      float4* d = reinterpret_cast<float4*>(dw) + i;
      float4 s = make_float4((float)sx, (float)sy, (float)sz, (float)sw);
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
      double s = 0.0;
      for (int k = 0; k < splits; ++k) s += (double)bs[(int64_t)k * nb + i];
      db[i] = accumulate ? db[i] + (float)s : (float)s;
    }
  }
}

// chunk table and split-K plan of a launch; wk = pixels per K chunk of the kernel that will run
static void wgrad_plan(const scan_pyramid_t* d, int Cs, int Cout, int KX, int wk, ChunkTab* ct, int* n_tiles, int* c_tiles,
                       int* splits, int* cps) {
  ct->chunk_off[0] = 0;
  for (int l = 0; l < SCAN_MAX_LEVELS; ++l) {
    if (l < d->n_levels) {
      ct->segs[l] = (d->w[l] + wk - 1) / wk;
      ct->chunk_off[l + 1] = ct->chunk_off[l] + (long long)d->n_images * d->h[l] * ct->segs[l];
    } else {
      ct->segs[l] = 1;
      ct->chunk_off[l + 1] = ct->chunk_off[l];
    }
  }
  const long long chunks = ct->chunk_off[d->n_levels];
  *c_tiles = (Cs + 127) / 128;
  *n_tiles = ((Cout + 127) / 128) * KX * *c_tiles;
  // ~3 workgroups per CU in total.  Swept on the device (tower layer, two pieces, us): 256 -> 499, 512 -> 428, 768 -> 355,
  // 1024 -> 414, 1536 -> 411, 2304 -> 486: fewer splits lengthen each workgroup's serial chunk chain, more splits
  // cost slab traffic and leave partial rounds
  // a thin last channel tile (264 / 268 input channels: 8 / 12 live of 128) leaves a third of the workgroups with one column
  // tile of MFMAs: more, shorter workgroups balance that (class branches 264 -> 1024 3390 -> 3304 us, head_out 1181 -> 1104 at 1280)
  const int target = (Cs > 128 && Cs % 128 != 0 && Cs % 128 <= 16) ? g_scan_wgrad_wgs * 5 / 3 : g_scan_wgrad_wgs;
  long long s = target / *n_tiles;
  if (s < 1) s = 1;
  const long long smax = (chunks + 7) / 8;
  if (s > smax) s = smax;
  s = (s + 7) / 8 * 8;  // groups of 8 splits, one per XCD
  *cps = (int)((chunks + s - 1) / s);
  if (*cps < 1) *cps = 1;
  *splits = (int)s;
}

// pixels per K chunk of the kernel a 3x3 launch with np pieces takes
static inline int wgrad3_wk(int np) { return (g_scan_wgrad_v6 && np == 3) ? 32 : WK; }

static int64_t wgrad3_ws_floats(int np, const scan_pyramid_t* d, int32_t Cs, int32_t Cout) {
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad_plan(d, Cs, Cout, 3, wgrad3_wk(np), &ct, &nt, &ctl, &sp, &cps);
  return (int64_t)sp * Cout * 9 * Cs + (int64_t)sp * Cout;
}
extern "C" int64_t scan_conv3x3_wgrad_bf16x3_ws_floats(const scan_pyramid_t* d, int32_t Cs, int32_t Cout) {
  return wgrad3_ws_floats(2, d, Cs, Cout);
}
extern "C" int64_t scan_conv3x3_wgrad_bf16x6_ws_floats(const scan_pyramid_t* d, int32_t Cs, int32_t Cout) {
  return wgrad3_ws_floats(3, d, Cs, Cout);
}

template <typename K>
static void set_lds(K kernel, size_t bytes) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int NP>
static int wgrad3_launch(const float* x, const scan_pyramid_t* d, int32_t Cs, const float* dy, int32_t Cout, int32_t Cout_s,
                         float* dw, float* db, int32_t accumulate, float* ws, void* stream) {
  const char* name = NP == 2 ? "conv3x3_wgrad_bf16x3" : "conv3x3_wgrad_bf16x6";
  SCAN_CHECK_ARG(d && d->n_levels >= 1 && d->n_levels <= SCAN_MAX_LEVELS && d->n_images >= 1, "%s: bad pyramid", name);
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "%s: Cs=%d must be a positive multiple of 4", name, Cs);
  SCAN_CHECK_ARG(Cout > 0 && Cout_s >= Cout && Cout_s % 4 == 0, "%s: Cout=%d Cout_s=%d (Cout_s a multiple of 4)", name, Cout, Cout_s);
  SCAN_CHECK_ARG(x && dy && dw && ws, "%s: null pointer", name);
  constexpr int WK6 = NP == 3 ? 32 : WK;
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad_plan(d, Cs, Cout, 3, wgrad3_wk(NP), &ct, &nt, &ctl, &sp, &cps);
  hipStream_t st = as_stream(stream);
  float* bias_slab = db ? ws + (int64_t)sp * Cout * 9 * Cs : nullptr;
  if (g_scan_wgrad_v6) {
    constexpr size_t sh6 = (size_t)2 * W6STAGE(NP, WK6, 3) * sizeof(__bf16);
    static_assert(sh6 <= 160 * 1024, "LDS: 160 KB per CU");
    static bool done = false;
    if (!done) {
      set_lds(conv_wgrad_v6_kernel<NP, WK6, 3, 4, 2>, sh6);
      set_lds(conv_wgrad_v6_kernel<NP, WK6, 3, 2, 4>, sh6);
      done = true;
    }
    if ((g_scan_wgrad_tile == 0 || g_scan_wgrad_tile == 1) ? g_scan_wgrad_tile == 1 : NP == 3)
      hipLaunchKernelGGL((conv_wgrad_v6_kernel<NP, WK6, 3, 2, 4>), dim3(nt * sp), dim3(W6_THREADS), sh6, st, x, *d, Cs, dy, Cout, Cout_s,
                         ws, bias_slab, ct, nt, ctl, cps, sp, g_scan_wgrad_prio);
    else
      hipLaunchKernelGGL((conv_wgrad_v6_kernel<NP, WK6, 3, 4, 2>), dim3(nt * sp), dim3(W6_THREADS), sh6, st, x, *d, Cs, dy, Cout, Cout_s,
                         ws, bias_slab, ct, nt, ctl, cps, sp, g_scan_wgrad_prio);
  } else {
    constexpr size_t sh = (size_t)WBUF(NP, 3) * sizeof(__bf16);
    static_assert(sh <= 160 * 1024, "LDS: 160 KB per CU");
    static bool done = false;
    if (!done) {
      set_lds(conv_wgrad_v4_kernel<NP, 3, 1>, sh);
      done = true;
    }
    hipLaunchKernelGGL((conv_wgrad_v4_kernel<NP, 3, 1>), dim3(nt * sp), dim3(512), sh, st, x, *d, Cs, dy, Cout, Cout_s, ws,
                       bias_slab, ct, nt, ctl, cps, sp, *d);
  }
  SCAN_LAUNCH_CHECK(name);
  // one launch reduces the weight slabs and (last block) the bias slabs
  const int64_t n = (int64_t)Cout * 9 * Cs;
  hipLaunchKernelGGL(slab_bias_reduce_kernel, dim3(grid_for(n / 4, 256) + (db ? 1 : 0)), dim3(256), 0, st, ws, sp, n, dw,
                     bias_slab, Cout, db, accumulate);
  SCAN_LAUNCH_CHECK("slab_bias_reduce");
  return 0;
}

extern "C" int scan_conv3x3_wgrad_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const float* dy,
                                         int32_t Cout, int32_t Cout_s, float* dw, float* db, int32_t accumulate,
                                         float* ws, void* stream) {
  return wgrad3_launch<2>(x, d, Cs, dy, Cout, Cout_s, dw, db, accumulate, ws, stream);
}
extern "C" int scan_conv3x3_wgrad_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const float* dy,
                                         int32_t Cout, int32_t Cout_s, float* dw, float* db, int32_t accumulate,
                                         float* ws, void* stream) {
  return wgrad3_launch<3>(x, d, Cs, dy, Cout, Cout_s, dw, db, accumulate, ws, stream);
}

// ---- 1x1 weight gradient (stride 1 or 2): dw[Cout][1][Cs] = sum_pixels dY^T X, conv_wgrad_v4_kernel with one tap.
extern "C" int64_t scan_conv1x1_wgrad_bf16x3_ws_floats(const scan_pyramid_t* yd, int32_t Cs, int32_t Cout) {
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad_plan(yd, Cs, Cout, 1, WK, &ct, &nt, &ctl, &sp, &cps);
  return (int64_t)sp * Cout * Cs + (int64_t)sp * Cout;
}
extern "C" int64_t scan_conv1x1_wgrad_bf16x6_ws_floats(const scan_pyramid_t* yd, int32_t Cs, int32_t Cout) {
  return scan_conv1x1_wgrad_bf16x3_ws_floats(yd, Cs, Cout);
}

template <int NP>
static int wgrad1_launch(const float* x, const scan_pyramid_t* xd, int32_t Cs, const float* dy, const scan_pyramid_t* yd,
                         int32_t Cout, int32_t Cout_s, int32_t stride, float* dw, float* db, int32_t accumulate, float* ws,
                         void* stream) {
  const char* name = NP == 2 ? "conv1x1_wgrad_bf16x3" : "conv1x1_wgrad_bf16x6";
  SCAN_CHECK_ARG(xd && yd && yd->n_levels >= 1 && yd->n_levels <= SCAN_MAX_LEVELS && yd->n_images >= 1 &&
                     xd->n_levels == yd->n_levels && xd->n_images == yd->n_images,
                 "%s: bad pyramids", name);
  SCAN_CHECK_ARG(stride == 1 || stride == 2, "%s: stride must be 1 or 2, got %d", name, stride);
  for (int l = 0; l < yd->n_levels; ++l)
    SCAN_CHECK_ARG((xd->h[l] - 1) / stride + 1 == yd->h[l] && (xd->w[l] - 1) / stride + 1 == yd->w[l],
                   "%s: level %d sizes do not match stride %d", name, l, stride);
  SCAN_CHECK_ARG(Cs > 0 && Cs % 4 == 0, "%s: Cs=%d must be a positive multiple of 4", name, Cs);
  SCAN_CHECK_ARG(Cout > 0 && Cout_s >= Cout && Cout_s % 4 == 0, "%s: Cout=%d Cout_s=%d (Cout_s a multiple of 4)", name, Cout, Cout_s);
  SCAN_CHECK_ARG(x && dy && dw && ws, "%s: null pointer", name);
  ChunkTab ct;
  int nt, ctl, sp, cps;
  wgrad_plan(yd, Cs, Cout, 1, WK, &ct, &nt, &ctl, &sp, &cps);
  hipStream_t st = as_stream(stream);
  constexpr size_t sh = (size_t)WBUF(NP, 1) * sizeof(__bf16);
  static bool done = false;
  if (!done) {
    set_lds(conv_wgrad_v4_kernel<NP, 1, 1>, sh);
    set_lds(conv_wgrad_v4_kernel<NP, 1, 2>, sh);
    done = true;
  }
  float* bias_slab = db ? ws + (int64_t)sp * Cout * Cs : nullptr;
  if (stride == 1)
    hipLaunchKernelGGL((conv_wgrad_v4_kernel<NP, 1, 1>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout, Cout_s, ws,
                       bias_slab, ct, nt, ctl, cps, sp, *xd);
  else
    hipLaunchKernelGGL((conv_wgrad_v4_kernel<NP, 1, 2>), dim3(nt * sp), dim3(512), sh, st, x, *yd, Cs, dy, Cout, Cout_s, ws,
                       bias_slab, ct, nt, ctl, cps, sp, *xd);
  SCAN_LAUNCH_CHECK(name);
  const int64_t n = (int64_t)Cout * Cs;
  hipLaunchKernelGGL(slab_bias_reduce_kernel, dim3(grid_for(n / 4, 256) + (db ? 1 : 0)), dim3(256), 0, st, ws, sp, n, dw,
                     bias_slab, Cout, db, accumulate);
  SCAN_LAUNCH_CHECK("slab_bias_reduce");
  return 0;
}

extern "C" int scan_conv1x1_wgrad_bf16x3(const float* x, const scan_pyramid_t* xd, int32_t Cs, const float* dy,
                                         const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t stride,
                                         float* dw, float* db, int32_t accumulate, float* ws, void* stream) {
  return wgrad1_launch<2>(x, xd, Cs, dy, yd, Cout, Cout_s, stride, dw, db, accumulate, ws, stream);
}
extern "C" int scan_conv1x1_wgrad_bf16x6(const float* x, const scan_pyramid_t* xd, int32_t Cs, const float* dy,
                                         const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t stride,
                                         float* dw, float* db, int32_t accumulate, float* ws, void* stream) {
  return wgrad1_launch<3>(x, xd, Cs, dy, yd, Cout, Cout_s, stride, dw, db, accumulate, ws, stream);
}
