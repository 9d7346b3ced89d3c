// 3x3 / stride-1 (and 1x1) convolution, forward and data gradient, on v_mfma_f32_16x16x32_bf16 with split fp32 operands
// (conv_split.h: NP = 2 pieces = "bf16x3", NP = 3 pieces = "bf16x6", the full fp32 significand).
//
//   MFMA roles      A operand = weights (16 output channels x 32 input channels), B operand = pixels (32 channels x 16
//                   pixels of ONE tile row), so a lane's four accumulator registers are four CONSECUTIVE output
//                   channels of one pixel: the epilogue stores (and the dgrad ReLU-mask loads) are 16-byte accesses
//   workgroup       16 x 16 output pixels x BN channels (BN = 128: 64 accumulator registers per lane at 8 waves, 256:
//                   128), waves = (pixel rows) x 2 (channels); 8 x 16 or 16 x 16 pixels x 64 channels with 4 waves for
//                   Cout <= 64
//   K loop          32-channel chunks = exactly one MFMA k-step; per chunk the (TH+2) x 18 x 32 halo patch is read
//                   once as fp32, split into NP bf16 planes in LDS and reused by all nine taps
//   LDS image       64 bytes per pixel (weight row) and plane, NO padding: the four 16-byte k-groups of a row are
//                   stored at slot kg ^ (2 * ((idx >> 2) & 1)), idx = patch column (weight row).  Checked exhaustively
//                   for the 16-lane groups of ds_read_b128 and every tap shift kx: conflict-free.
//   LDS bytes       patch NP x 20.3 KB (16-row tile) + weight double buffer 2 x NP x BN x 64 B:
//                   NP = 2: 106 KB (BN = 256), 73 KB (128);  NP = 3: 160.5 KB (256: the whole CU), 111 KB (128)
//
// dgrad = the same kernel on dY with flipped / transposed weight planes (scan_weight_split mode 1).  Epilogue variants:
// bias, ReLU, ReLU mask of a deferred ReLU (dgrad), fused 2x2 max-pool (frozen stages), GroupNorm sums.
//
// History of the structure with its measurements: DESIGN.md section 3.1 (16x16x32 against 32x32x16: profiles/
// r02_mfma_shape_experiment.txt; LDS-DMA weight tiles: r02_conv_instances.txt; buffer-load staging and the ablations that
// located the remaining time: r03_conv_fwd.txt, r03_conv_exp.txt).
#include "conv_split.h"

#define V2_TW 16
#ifndef SCAN_CONV_MID
#define SCAN_CONV_MID 0  // channel tile of a barrier interval behind whose MFMAs the LDS-DMA path feeds the next tile; -1: right behind the barrier
#endif
#ifndef SCAN_CONV_PFA
#define SCAN_CONV_PFA 1  // 8-wave LDS-DMA instance, NP = 2: patch fragments of the next tap prefetched in front of the barrier
#endif
#define V2_CK 32  // channels per K chunk = one k-step of v_mfma_f32_16x16x32_bf16

struct TileTab2 {
  int tile_off[SCAN_MAX_LEVELS + 1];
  int tiles_x[SCAN_MAX_LEVELS];
  int tiles_y[SCAN_MAX_LEVELS];
};

// 16-byte slot swizzle of a 64-byte row: k-group kg of row idx lives at slot kg ^ swz(idx)
__device__ __forceinline__ int swz(int idx) { return (idx >> 1) & 2; }

// The weight plane slot I of a thread reads from, WITHOUT a run-time table: slot = tid + NT * I lies in plane slot / PL, and
// for every instance that is one compile-time plane or one of two neighbours.  (Written as a nested select on the run-time
// plane index the compiler builds {w0, w1, w2} on the stack and indexes it: scratch in every kernel.)
template <int K>
__device__ __forceinline__ const __bf16* pick_plane(const __bf16* w0, const __bf16* w1, const __bf16* w2) {
  if constexpr (K == 0)
    return w0;
  else if constexpr (K == 1)
    return w1;
  else
    return w2;
}
template <int I, int NT, int PL, int NPL>
__device__ __forceinline__ const __bf16* slot_plane(int tid, const __bf16* w0, const __bf16* w1, const __bf16* w2) {
  constexpr int lo = (NT * I) / PL, hi = (NT * I + NT - 1) / PL;
  constexpr int lo_c = lo < NPL ? lo : NPL - 1, hi_c = hi < NPL ? hi : NPL - 1;  // slots past the last plane are never used
  static_assert(hi - lo <= 1, "a thread's slot spans at most two planes");
  if constexpr (lo_c == hi_c)
    return pick_plane<lo_c>(w0, w1, w2);
  else
    return (tid + NT * I) / PL == lo ? pick_plane<lo_c>(w0, w1, w2) : pick_plane<hi_c>(w0, w1, w2);
}
template <int NT, int PL, int NPL>
__device__ __forceinline__ const __bf16* slot_plane_i(int i, int tid, const __bf16* w0, const __bf16* w1, const __bf16* w2) {
  // i is a constant after unrolling: the chain folds to one call
  return i == 0   ? slot_plane<0, NT, PL, NPL>(tid, w0, w1, w2)
         : i == 1 ? slot_plane<1, NT, PL, NPL>(tid, w0, w1, w2)
         : i == 2 ? slot_plane<2, NT, PL, NPL>(tid, w0, w1, w2)
         : i == 3 ? slot_plane<3, NT, PL, NPL>(tid, w0, w1, w2)
         : i == 4 ? slot_plane<4, NT, PL, NPL>(tid, w0, w1, w2)
                  : slot_plane<5, NT, PL, NPL>(tid, w0, w1, w2);
}

// the piece products of one (weight fragment set, pixel fragment set), smallest terms first (conv_split.h)
template <int NP, int TM>
__device__ __forceinline__ void mma_pieces(const bf16x8 (&w)[NP], const bf16x8 (&p)[NP][TM], f32x4v (&acc)[TM]) {
#pragma unroll
  for (int s = NP - 1; s >= 0; --s)
#pragma unroll
    for (int i = 0; i <= s; ++i)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) acc[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i], p[s - i][tm], acc[tm], 0, 0, 0);
}

// NP: pieces per operand; BN: output channels per workgroup; TH: tile rows (tile = TH x 16 pixels); NT: threads; KS: 3 or 1.
// KS = 1 also serves the stride-2 1x1 convs through MAP (0: same pyramid; 1: source = 2 * output, forward of a stride-2
// conv; 2: source = output / 2 on even coordinates, zero elsewhere: its data gradient).
// TPB: taps staged per barrier (1, or 3 = one ky row of the 3x3: the weight tiles of three taps share one barrier
// interval, 4 barriers per 32-channel chunk instead of 10 -- NP = 2, BN <= 128 only: LDS)
// GL: the weight tiles go global -> LDS by LDS-DMA (buffer_load ... lds) instead of through registers: no staging
// registers, no ds_write pass, and the next tap's tile is in flight while the matrix cores work on the current one (the
// slot swizzle moves to the per-lane SOURCE address: the DMA writes a wave's 64 x 16 bytes contiguously).  Needs whole
// K chunks (Csw % 32 == 0).  Rows beyond Nout in the last channel tile lie beyond the plane, i.e. beyond the buffer
// descriptor's num_records: the range check returns zeros for them, which is what the DMA writes.
template <int NP, int BN, int TH, int NT, int KS, int TPB = 1, bool GL = false>
__global__ __launch_bounds__(NT, NT == 1024 ? 4 : 2) void conv_split_kernel(
    const float* __restrict__ src, scan_pyramid_t d, int Cs, const __bf16* __restrict__ w0, const __bf16* __restrict__ w1,
    const __bf16* __restrict__ w2, int Csw, const float* __restrict__ bias, const float* __restrict__ mask,
    float* __restrict__ dst, int Nout, int Ns, int relu, TileTab2 tt, int n_tiles, scan_pyramid_t sd, int map,
    double* __restrict__ gn_ws) {
  constexpr int HALO = KS / 2, NTAPS = KS * KS;
  constexpr int PH = TH + 2 * HALO;
  constexpr int PWK = V2_TW + 2 * HALO;
  constexpr int NPATCH = PH * PWK;
  constexpr int WAVES = NT / 64;
  constexpr int WN_WAVES = BN >= 128 ? 2 : 1;
  constexpr int WM_WAVES = WAVES / WN_WAVES;
  constexpr int TM = TH / WM_WAVES;             // 16-pixel tile rows per wave
  constexpr int TN = BN / (16 * WN_WAVES);      // 16-channel tiles per wave (4, or 8 for BN = 256)
  constexpr int ASLOTS = (NPATCH * 8 + NT - 1) / NT;  // float4 of the halo patch per thread per chunk
  constexpr int BSLOTS = BN * 4 * NP;                 // 16-byte weight segments per (chunk, tap)
  constexpr int BSEG = (BSLOTS + NT - 1) / NT;        // ... per thread
  constexpr int NGRP = NTAPS / TPB;                   // barrier intervals per chunk
  static_assert(NP == 2 || NP == 3, "two or three pieces per operand");
  static_assert(TM * WM_WAVES == TH && (TM % 2) == 0, "tile rows must split evenly (and pair up for the fused pool)");
  static_assert(NTAPS % TPB == 0, "taps per barrier must divide the tap count");
  static_assert(BSLOTS % 64 == 0, "a wave's 64 weight segments lie in one plane (and past the end only as a whole wave)");

  extern __shared__ __align__(16) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);  // [NP plane][NPATCH][32]
  __bf16* Bs = As + NP * NPATCH * 32;                // [2 buf][TPB taps][NP plane][BN][32]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
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
  const int ty0 = (t / tt.tiles_x[lvl]) * TH, tx0 = (t % tt.tiles_x[lvl]) * V2_TW;
  const int64_t rowbase = d.row_off[lvl] + (int64_t)img * H * W;
  const int n0 = n_tile * BN;
  const int nchunks = (Cs + V2_CK - 1) / V2_CK;

  // ---- halo patch staging: NPATCH pixels x 8 float4, ASLOTS per thread, prefetched one chunk ahead in registers.
  // Range-checked buffer loads: a lane's byte offset from the tile's first patch pixel is the same for every K chunk
  // (the chunk is the scalar offset of the load), pixels outside the image carry an offset beyond the descriptor and
  // read zeros -- no predication, no per-chunk address arithmetic, no 64-bit per-lane pointers held across the K loop.
  constexpr unsigned BAD = 0x80000000u;
  unsigned voff[ASLOTS];
  const int c_tail = Cs - (nchunks - 1) * V2_CK;  // channels of the last chunk (1..32)
  // a lane's float4 column inside a chunk is the same for all its slots (NT is a multiple of 8): one predicate says
  // whether it lies beyond the channel count in the last chunk
  const bool tail_bad = 4 * (tid & 7) >= c_tail;
  __amdgpu_buffer_rsrc_t a_src;
  {
    const bool mapped = KS == 1 && map != 0;
    const int Hs = mapped ? sd.h[lvl] : H, Ws = mapped ? sd.w[lvl] : W;
    int64_t base_row;
    if (!mapped)
      base_row = rowbase + (int64_t)(ty0 - HALO) * W + (tx0 - HALO);
    else if (map == 1)
      base_row = sd.row_off[lvl] + ((int64_t)img * Hs + 2 * ty0) * Ws + 2 * tx0;
    else
      base_row = sd.row_off[lvl] + ((int64_t)img * Hs + (ty0 >> 1)) * Ws + (tx0 >> 1);
    a_src = uniform_rsrc_b(src + base_row * Cs, 0x7ffffff0);
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
      const int slot = tid + NT * i;
      const int q = slot >> 3, c4 = slot & 7;
      const int py = q / PWK, px = q - py * PWK;
      const int y = ty0 - HALO + py, x = tx0 - HALO + px;
      bool ok = (slot < NPATCH * 8) && y >= 0 && y < H && x >= 0 && x < W;
      int pix = py * W + px;
      if (mapped) {
        if (map == 1) {
          ok = ok && 2 * y < Hs && 2 * x < Ws;
          pix = 2 * py * Ws + 2 * px;
        } else {
          ok = ok && ((y | x) & 1) == 0 && (y >> 1) < Hs && (x >> 1) < Ws;
          pix = (py >> 1) * Ws + (px >> 1);
        }
      }
      voff[i] = ok ? (unsigned)((pix * Cs + 4 * c4) * 4) : BAD;
    }
  }
  float4 ra[ASLOTS];
  auto load_a = [&](int cc) {
    const int soff = cc * (V2_CK * 4);
    const bool kill = tail_bad && cc == nchunks - 1;
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i)
      ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_src, (int)(kill ? BAD : voff[i]), soff, 0));
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
      const int slot = tid + NT * i;
      if (slot < NPATCH * 8) {
        const int q = slot >> 3, c4 = slot & 7;
        const int px = q % PWK;
        bf16x4 pc[NP];
        split4_np<NP>(ra[i], pc);
        const int off = q * 32 + (((c4 >> 1) ^ swz(px)) << 3) + ((c4 & 1) << 2);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(As + p * NPATCH * 32 + off) = pc[p];
      }
    }
  };
  // ---- weight tile staging: BN rows x 4 segments x NP planes per (chunk, tap), double-buffered in LDS
  uint4 rb[TPB][BSEG];
  auto load_b = [&](int cc, int grp) {
#pragma unroll
    for (int tt = 0; tt < TPB; ++tt) {
      const int tap = grp * TPB + tt;
#pragma unroll
      for (int i = 0; i < BSEG; ++i) {
        const int slot = tid + NT * i;
        const int plane = slot / (BN * 4);
        const int rem = slot - plane * BN * 4;
        const int row = rem >> 2, seg = rem & 3;
        const int o = n0 + row, c = cc * V2_CK + 8 * seg;
        const __bf16* base = slot_plane_i<NT, BN * 4, NP>(i, tid, w0, w1, w2);
        rb[tt][i] = (slot < BSLOTS && o < Nout && c < Csw)
                        ? *reinterpret_cast<const uint4*>(base + ((int64_t)o * NTAPS + tap) * Csw + c)
                        : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int tt = 0; tt < TPB; ++tt) {
#pragma unroll
      for (int i = 0; i < BSEG; ++i) {
        const int slot = tid + NT * i;
        const int plane = slot / (BN * 4);
        const int rem = slot - plane * BN * 4;
        const int row = rem >> 2, seg = rem & 3;
        if (slot < BSLOTS)
          *reinterpret_cast<uint4*>(Bs + (((buf * TPB + tt) * NP + plane) * BN + row) * 32 + ((seg ^ swz(row)) << 3)) = rb[tt][i];
      }
    }
  };

  // LDS-DMA of the weight tiles as buffer loads: a lane's byte offset inside its plane (row n0 + row, source k-group of
  // its destination slot) is constant, the (chunk, tap) position is the scalar offset -- no 64-bit per-lane pointers,
  // no per-tap address arithmetic.  One descriptor per slot: the plane a slot reads from is wave-uniform (BN * 4 is a
  // multiple of 64).
  static_assert(!GL || BSEG <= 6, "the LDS-DMA path keeps at most six descriptors");
  unsigned boff[BSEG];
#pragma unroll
  for (int i = 0; i < BSEG; ++i) {
    const int slot = tid + NT * i;
    const int plane = slot / (BN * 4);
    const int rem = slot - plane * BN * 4;   // destination slot inside the plane: row * 4 + dslot
    const int row = rem >> 2, seg = (rem & 3) ^ swz(row);  // source k-group of that slot (swz is an involution)
    boff[i] = (unsigned)((((n0 + row) * NTAPS) * Csw + 8 * seg) * 2);
  }
  // (separate variables, not an array: an array of __amdgpu_buffer_rsrc_t silently drops the kernel's host stub)
  const int plane_bytes = Nout * NTAPS * Csw * 2;  // rows >= Nout are out of range: they read (and the DMA writes) zeros
  auto plane_rsrc = [&](int i) { return uniform_rsrc_b(slot_plane_i<NT, BN * 4, NP>(i, tid, w0, w1, w2), plane_bytes); };
  const __amdgpu_buffer_rsrc_t b_src0 = plane_rsrc(0), b_src1 = plane_rsrc(1), b_src2 = plane_rsrc(2),
                               b_src3 = plane_rsrc(3), b_src4 = plane_rsrc(4), b_src5 = plane_rsrc(5);
  auto issue_b = [&](int cc, int grp, int buf) {
#pragma unroll
    for (int tt = 0; tt < TPB; ++tt) {
      const int tap = grp * TPB + tt;
      const int soff = (tap * Csw + cc * V2_CK) * 2;
#pragma unroll
      for (int i = 0; i < BSEG; ++i) {
        const int slot = tid + NT * i;
        if (BSEG * NT != BSLOTS && slot - lane >= BSLOTS) continue;  // (wave-uniform) no such segments
        const int plane = slot / (BN * 4);
        const int rem = slot - plane * BN * 4;
        __bf16* dst_l = Bs + (((buf * TPB + tt) * NP + plane) * BN) * 32 + (rem - lane) * 8;  // the wave's first slot
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            i == 0 ? b_src0 : i == 1 ? b_src1 : i == 2 ? b_src2 : i == 3 ? b_src3 : i == 4 ? b_src4 : b_src5, (lds_ptr_t)dst_l,
            16, (int)boff[i], soff, 0, 0);
      }
    }
  };

  // ---- MFMA roles
  const int wm = wid / WN_WAVES, wn = wid % WN_WAVES;
  const int lr = lane & 15, kg = lane >> 4;
  const int w_row0 = wn * 16 * TN + lr;                                   // this lane's weight row in channel tile 0
  const int w_off = w_row0 * 32 + ((kg ^ swz(w_row0)) << 3);             // + tn * 16 * 32 (swz(row) has period 8)

  f32x4v acc[TN][TM];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int a = 0; a < TM; ++a) acc[b][a] = f32x4v{0.f, 0.f, 0.f, 0.f};

  auto patch_off = [&](int tap) {
    const int ky = tap / KS, kx = tap - KS * ky;
    const int pxs = lr + kx;  // patch column of this lane's pixel
    return ((wm * TM + ky) * PWK + pxs) * 32 + ((kg ^ swz(pxs)) << 3);
  };
  auto read_patch = [&](int tap, bf16x8 (&pf)[NP][TM]) {
    const int p_off = patch_off(tap);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        pf[p][tm] = *reinterpret_cast<const bf16x8*>(As + p * NPATCH * 32 + p_off + tm * PWK * 32);
  };
  auto read_w = [&](const __bf16* bt, int tn, bf16x8 (&wf)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) wf[p] = *reinterpret_cast<const bf16x8*>(bt + p * BN * 32 + tn * 16 * 32);
  };
  // mid(): called once per barrier interval after the MFMAs of the first channel tile -- the LDS-DMA path issues the next
  // tap's weight tile (and the next chunk's patch loads) THERE instead of right behind the barrier: a DMA piece costs its
  // wave 60-185 cycles of issue time (MI355X_MICROARCH.md), and with all waves of a SIMD paying that at the start of the
  // interval the matrix pipe idled for it (profiles/r03_conv_exp.txt: the memory side costs the kernel ~15 %)
  auto taps_mma = [&](int grp, int buf, auto&& mid) {
#pragma unroll
    for (int tt = 0; tt < TPB; ++tt) {
      const int tap = grp * TPB + tt;
      const __bf16* bt = Bs + ((buf * TPB + tt) * NP) * BN * 32 + w_off;
      bf16x8 pf[NP][TM];
      read_patch(tap, pf);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        bf16x8 wf[NP];
        read_w(bt, tn, wf);
        mma_pieces<NP, TM>(wf, pf, acc[tn]);
#ifdef SCAN_EXP_MID_STAGGER  // timing experiment (make exp_mid_stagger): the second wave of a SIMD feeds half an interval later
        if (tt == 0 && tn == ((wid >= WAVES / 2 && TN >= 8) ? SCAN_EXP_MID_STAGGER : SCAN_CONV_MID)) {
#else
        if (tt == 0 && tn == SCAN_CONV_MID) {
#endif
          __builtin_amdgcn_sched_barrier(0);
          mid();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  auto no_mid = []() {};

  // 8-wave LDS-DMA instance, two pieces (two waves per SIMD, 64 px x 128 ch per wave): the patch fragments of the NEXT tap
  // are read behind the last channel tile's MFMAs, in front of the barrier -- the patch does not change inside a chunk,
  // only the weight fragments have to wait for the barrier.  With two waves per SIMD nothing else hides those reads.
  // (Three pieces: 48 more registers than the 256-channel tile has left; the doubled MFMA work per fragment hides them.)
  constexpr bool PFA = GL && NT == 512 && TPB == 1 && NP == 2 && SCAN_CONV_PFA;
  bf16x8 npf[NP][TM];
  auto taps_mma_pf = [&](int grp, int buf, auto&& mid) {
    if constexpr (PFA) {
      const __bf16* bt = Bs + (buf * NP) * BN * 32 + w_off;
      if (grp == 0) read_patch(0, npf);  // first tap of a chunk: the patch was stored just now
#pragma unroll
      for (int tn = 0; tn < TN - 1; ++tn) {
        bf16x8 wf[NP];
        read_w(bt, tn, wf);
        mma_pieces<NP, TM>(wf, npf, acc[tn]);
        if (tn == SCAN_CONV_MID) {
          __builtin_amdgcn_sched_barrier(0);
          mid();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      {
        constexpr int tn = TN - 1;
        bf16x8 wf[NP];
        read_w(bt, tn, wf);
        const bool more = grp < NGRP - 1;
        const int p_off = patch_off(more ? grp + 1 : grp);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], npf[1][tm], acc[tn][tm], 0, 0, 0);
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], npf[0][tm], acc[tn][tm], 0, 0, 0);
          acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], npf[0][tm], acc[tn][tm], 0, 0, 0);
          // unconditional (the last tap of a chunk re-reads its own fragments, unused): no branch between the MFMAs
          npf[0][tm] = *reinterpret_cast<const bf16x8*>(As + p_off + tm * PWK * 32);
          npf[1][tm] = *reinterpret_cast<const bf16x8*>(As + NPATCH * 32 + p_off + tm * PWK * 32);
        }
      }
    }
  };

  load_a(0);
  if constexpr (GL) {
    issue_b(0, 0, 0);
    for (int cc = 0; cc < nchunks; ++cc) {
      __syncthreads();  // every wave is done reading the previous chunk's patch
#ifdef SCAN_EXP_FWD_NOFEED
      if (cc == 0 || !(SCAN_EXP_FWD_NOFEED & 2))
#endif
      store_a();
      if (NGRP == 1 && cc + 1 < nchunks) load_a(cc + 1);
#pragma unroll 1
      for (int grp = 0; grp < NGRP; ++grp) {
        const int buf = (cc * NGRP + grp) & 1;
        // an LDS-DMA counts on vmcnt and the compiler does not wait for it on its own: this wave's pieces of the tile
        // (issued one tap ago) must have landed before the barrier publishes the tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // this tap's weight tile is complete; the patch is visible
        auto feed = [&]() {
#ifdef SCAN_EXP_FWD_NOFEED  // TIMING EXPERIMENT (make exp_fwd_nofeed [M=1|2|3], WRONG results): bit 0 = no weight-tile DMA, bit 1 = no
          // patch loads / conversion / LDS writes after the first chunk
          if ((SCAN_EXP_FWD_NOFEED & 1) && !(SCAN_EXP_FWD_NOFEED & 2)) {
            if (NGRP > 1 && grp == NGRP - 2 && cc + 1 < nchunks) load_a(cc + 1);
            return;
          }
          if ((SCAN_EXP_FWD_NOFEED & 3) == 3) return;
#endif
          // the other buffer was last read one tap ago: the next tap's tile goes there while this one is multiplied
          if (grp < NGRP - 1)
            issue_b(cc, grp + 1, buf ^ 1);
          else if (cc + 1 < nchunks)
            issue_b(cc + 1, 0, buf ^ 1);
          // the next chunk's patch: fetched one tap before it is needed (the barrier above drains every outstanding
          // load, so an earlier prefetch would only stall an earlier tap)
          if (NGRP > 1 && grp == NGRP - 2 && cc + 1 < nchunks) load_a(cc + 1);
        };
        if constexpr (PFA) {
          taps_mma_pf(grp, buf, feed);
        } else if (SCAN_CONV_MID >= 0) {
          taps_mma(grp, buf, feed);
        } else {
          feed();
          taps_mma(grp, buf, no_mid);
        }
      }
    }
  } else {
    load_b(0, 0);
    for (int cc = 0; cc < nchunks; ++cc) {
      __syncthreads();  // every wave is done reading the previous chunk's patch
      store_a();
      if (cc + 1 < nchunks) load_a(cc + 1);
#pragma unroll 1
      for (int grp = 0; grp < NGRP; ++grp) {
        const int buf = (cc * NGRP + grp) & 1;
        store_b(buf);
        if (grp < NGRP - 1)
          load_b(cc, grp + 1);
        else if (cc + 1 < nchunks)
          load_b(cc + 1, 0);
        __syncthreads();
        taps_mma(grp, buf, no_mid);
      }
    }
  }

  // ---- epilogue.  C/D map of 16x16: column = lane & 15 = pixel x of tile row tm, row = 4 * (lane >> 4) + reg = output
  // channel inside channel tile tn: one lane owns four consecutive channels of one pixel
  const int x = tx0 + lr;
  // GroupNorm sums leave the workgroup as ONE fp64 atomic pair per group (not one per wave and group): every tile of a frame
  // adds to the same 64 doubles, and same-address device-scope atomics serialise -- with a pair per wave the towers' forward
  // ran 20 % behind their data gradient (same kernel, same shape, no sums).  The waves' partials meet in the LDS the main
  // loop is done with.
  double* gn_red = reinterpret_cast<double*>(smem_raw);  // [WM_WAVES][BN / 8][2]
  if (gn_ws != nullptr) __syncthreads();
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int o4 = n0 + wn * 16 * TN + tn * 16 + 4 * kg;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) {
      bv.x = o4 + 0 < Nout ? bias[o4 + 0] : 0.f;
      bv.y = o4 + 1 < Nout ? bias[o4 + 1] : 0.f;
      bv.z = o4 + 2 < Nout ? bias[o4 + 2] : 0.f;
      bv.w = o4 + 3 < Nout ? bias[o4 + 3] : 0.f;
    }
    if (relu & 2) {
      // fused 2x2 / stride-2 max-pool (frozen VGG stages): rows y, y+1 are accumulator tiles tm, tm+1 of this lane,
      // columns x, x+1 are lanes l, l^1 -- one DPP exchange, then the even lanes write the pooled pixel
      const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
      for (int tm = 0; tm < TM; tm += 2) {
        const int y = ty0 + wm * TM + tm;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float m = fmaxf(acc[tn][tm][r], acc[tn][tm + 1][r]);
          v[r] = fmaxf(m, __shfl_xor(m, 1, 64));
        }
        if ((lr & 1) == 0 && y < H && x < W && o4 < Nout) {
          float4 o = make_float4(v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w);
          if (relu & 1) {
            o.x = fmaxf(o.x, 0.f);
            o.y = fmaxf(o.y, 0.f);
            o.z = fmaxf(o.z, 0.f);
            o.w = fmaxf(o.w, 0.f);
          }
          *reinterpret_cast<float4*>(dst + ((int64_t)img * Hp * Wp + (int64_t)(y >> 1) * Wp + (x >> 1)) * Ns + o4) = o;
        }
      }
      continue;
    }
    // the ReLU mask of a data gradient: fetch this channel tile's masks first so the loads overlap
    float4 mk[TM];
    if (mask != nullptr) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int y = ty0 + wm * TM + tm;
        const bool ok = y < H && x < W && o4 < Nout;
        mk[tm] = ok ? *reinterpret_cast<const float4*>(mask + (rowbase + (int64_t)y * W + x) * Ns + o4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    double ds = 0.0, dq = 0.0;  // fp32 over a pixel's four channels, fp64 from there on: the same for every instance
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int y = ty0 + wm * TM + tm;
      if (y < H && x < W && o4 < Nout) {
        float4 o = make_float4(acc[tn][tm][0] + bv.x, acc[tn][tm][1] + bv.y, acc[tn][tm][2] + bv.z,
                               acc[tn][tm][3] + bv.w);
        if (relu & 1) {
          o.x = fmaxf(o.x, 0.f);
          o.y = fmaxf(o.y, 0.f);
          o.z = fmaxf(o.z, 0.f);
          o.w = fmaxf(o.w, 0.f);
        }
        if (mask != nullptr) {
          o.x = (mk[tm].x > 0.f) ? o.x : 0.f;
          o.y = (mk[tm].y > 0.f) ? o.y : 0.f;
          o.z = (mk[tm].z > 0.f) ? o.z : 0.f;
          o.w = (mk[tm].w > 0.f) ? o.w : 0.f;
        }
#ifdef SCAN_EXP_FWD_NOSTORE  // TIMING EXPERIMENT (make exp_fwd_nostore, never in libscan_hip.so): the epilogue without its stores
        if (o.x == 1.2345e30f)
#endif
        *reinterpret_cast<float4*>(dst + (rowbase + (int64_t)y * W + x) * Ns + o4) = o;
        ds += (double)((o.x + o.y) + (o.z + o.w));
        dq += (double)((o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w));
      }
    }
    if (gn_ws != nullptr) {
      // GroupNorm(32) sums of the 256-channel output: a group = 8 channels = the lane-group pair kg, kg ^ 1; reduce
      // over the 16 pixels (lanes) and that pair, one fp64 atomic pair per group and wave
#pragma unroll
      for (int sh = 1; sh <= 16; sh <<= 1) {
        ds += __shfl_xor(ds, sh, 64);
        dq += __shfl_xor(dq, sh, 64);
      }
      if (lr == 0 && (kg & 1) == 0) {
        double* r = gn_red + (wm * (BN / 8) + ((o4 - n0) >> 3)) * 2;
        r[0] = ds;
        r[1] = dq;
      }
    }
  }
  if (gn_ws != nullptr) {
    __syncthreads();
    if (tid < BN / 8 * 2) {
      const int g = n0 / 8 + (tid >> 1);
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < WM_WAVES; ++w) v += gn_red[w * (BN / 8) * 2 + tid];
      if (g * 8 < Nout) atomicAdd(&gn_ws[((int64_t)(lvl * d.n_images + img) * 32 + g) * 2 + (tid & 1)], v);
    }
  }
}

static void make_tiles_v2(const scan_pyramid_t* d, TileTab2* tt, int TH) {
  tt->tile_off[0] = 0;
  for (int l = 0; l < SCAN_MAX_LEVELS; ++l) {
    if (l < d->n_levels) {
      tt->tiles_x[l] = (d->w[l] + V2_TW - 1) / V2_TW;
      tt->tiles_y[l] = (d->h[l] + TH - 1) / TH;
      tt->tile_off[l + 1] = tt->tile_off[l] + d->n_images * tt->tiles_x[l] * tt->tiles_y[l];
    } else {
      tt->tiles_x[l] = tt->tiles_y[l] = 1;
      tt->tile_off[l + 1] = tt->tile_off[l];
    }
  }
}

struct ConvArgs {
  const float* x;
  const scan_pyramid_t* od;  // output pyramid (tiles are enumerated over it)
  const scan_pyramid_t* sd;  // source pyramid (== od unless a stride-2 1x1 map is in play)
  int32_t Cs;
  const __bf16* w[3];
  int32_t Csw;
  const float* bias;
  const float* mask;
  float* y;
  int32_t Nout, Ns, relu, map;
  hipStream_t st;
  double* gn_ws;
};

template <int NP, int BN, int TH, int NT, int KS, int TPB = 1, bool GL = false>
static void launch_v2(const ConvArgs& a) {
  constexpr int HALO = KS / 2;
  TileTab2 tt;
  make_tiles_v2(a.od, &tt, TH);
  const int tiles = tt.tile_off[a.od->n_levels];
  const int n_tiles = (a.Nout + BN - 1) / BN;
  constexpr size_t sh = (size_t)(NP * (TH + 2 * HALO) * (V2_TW + 2 * HALO) * 32 + 2 * NP * TPB * BN * 32) * sizeof(__bf16);
  static_assert(sh <= 160 * 1024, "LDS: 160 KB per CU");
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_kernel<NP, BN, TH, NT, KS, TPB, GL>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    done = true;
  }
  hipLaunchKernelGGL((conv_split_kernel<NP, BN, TH, NT, KS, TPB, GL>), dim3(tiles * n_tiles), dim3(NT), sh, a.st, a.x, *a.od,
                     a.Cs, a.w[0], a.w[1], a.w[2], a.Csw, a.bias, a.mask, a.y, a.Nout, a.Ns, a.relu, tt, n_tiles, *a.sd,
                     a.map, a.gn_ws);
}

// Instance choice for an output pyramid and channel count: 64 (8x16- or 16x16-pixel tiles, 256 threads), 128 or 256
// (16x16-pixel tiles; 256 when the channels fill 256-wide tiles and the launch keeps >= 2 workgroups per CU).
int g_scan_conv_bn256 = 2;  // scan_tune "conv_bn256": 0 keeps every launch on the 128-channel instance, 1 = 256 when >= 512 workgroups result, 2 (default) = by rounds (below)
// scan_tune "conv_wg1024" (two pieces only): 1 (default) = the 128- and 256-channel 3x3 instances run with 16 waves per
// workgroup (each wave 32 px x 64 / 128 ch, <= 128 registers, four waves per SIMD) instead of 8 (64 px per wave, two per
// SIMD); 0 = 8 waves; 2 = 16 waves only for the 256-channel tile on multi-level pyramids.  Same-process A/B per layer
// (profiles/r02_conv_instances.txt): tower layer over P3..P7 603 -> 554 us (+8 %: the small levels' partial tiles
// leave 8-wave workgroups short of work to hide latency), single-level layers +1...3 %, none slower.
// With three pieces the fragments of a tap (NP x (TM + 1) x 4 registers) do not fit the 128-register cap of 16 waves:
// the three-piece instances always run 8 waves.
int g_scan_conv_wg1024 = 1;
// scan_tune "conv_w8": 1 = the two-piece 256-channel LDS-DMA instance runs as 8 waves (two per SIMD, 64 px x 128 ch per
// wave, up to 256 registers: 24 fragment reads per 96 MFMAs instead of 20 per 48) instead of 16; profiles/r03_conv_exp.txt
int g_scan_conv_w8 = 1;
// scan_tune "conv_tpb3" (two pieces only): stage the three taps of a ky row per barrier (4 barriers per 32-channel chunk
// instead of 10) -- bit 0 (default on): the 128-channel 3x3 instance, bit 1: the 64-channel one (196 instead of 140
// registers: two instead of three workgroups per CU).  Register-staged it measured +-1 % on every layer
// (profiles/r02_conv_instances.txt); with the weight tiles on LDS-DMA (no staging registers for three taps) the
// 128-channel instance gains 2 % (FCOS towers 285 -> 278 us, conv2_2 1679 -> 1640, conv5_x 378 -> 372).
int g_scan_conv_tpb3 = 1;
// scan_tune "conv_glds": 1 = the 128- / 256-channel 3x3 instances stage their weight tiles by LDS-DMA when the launch has
// whole tiles (Nout % tile == 0, Csw % 32 == 0); 0 = through registers.  Same-process A/B per layer (two pieces,
// tools/conv_bench.py, us): towers over P3..P7 608 -> 556, conv3_x 1530 -> 1431, conv4_x 1431 -> 1349; neutral on the
// 128-channel instance; the 64-channel instance gets slower (1906 -> 2040) and stays on registers.
int g_scan_conv_glds = 1;
// scan_tune "conv_bn64_th16": pixel tile of the <= 64-channel 3x3 instance on single-level pyramids -- 0: always 8x16 (4 waves);
// 1: 16x16 when H, W are multiples of 16 (8 waves x 32 px x 64 ch: conv1_2, 64 -> 64 at 1024x2048, 4 frames, two pieces 2047 ->
// 1940 us); 2 (default): three pieces additionally take 32x16 tiles when H is a multiple of 32 (8 waves x 64 px x 64 ch: 96
// MFMAs per wave and barrier instead of 48, half the weight-tile traffic per MFMA, halo 1.20 instead of 1.27; 142 KB of LDS,
// one workgroup per CU as before: conv1_2 3257 -> 2918 us, tools/conv_bench.py).  Same results bit for bit.
int g_scan_conv_bn64_th16 = 2;
static int v2_instance(const scan_pyramid_t* od, int32_t Nout) {
  if (Nout <= 64) return 64;
  TileTab2 tt;
  make_tiles_v2(od, &tt, 16);
  const int64_t tiles = tt.tile_off[od->n_levels];
  if (g_scan_conv_bn256 == 1 && Nout % 256 == 0 && tiles * (Nout / 256) >= 512) return 256;  // the rule of rounds 2-4
  if (g_scan_conv_bn256 == 2 && Nout % 256 == 0) {
    // rounds of 256 CUs (one workgroup per CU either way; a 256-channel workgroup runs twice as long as a 128-channel one and
    // reads half the fragments per MFMA): the wider tile whenever it does not cost a round -- conv5_x on 4 frames (128 pixel
    // tiles x 2: one full round of 256-channel workgroups instead of two of 128-channel ones), conv4_x on the 2 frames of inference
    const int64_t r256 = (tiles * (Nout / 256) + 255) / 256 * 2, r128 = (tiles * (Nout / 128) + 255) / 256;
    if (r256 * 97 <= r128 * 100) return 256;  // (the wider tile is ~3 % faster per unit of work: a tie of many rounds goes to it)
  }
  return 128;
}

// entry points used by the public functions of conv_bf16x3.hip.  np = pieces per operand (2: planes w0, w1; 3: w0, w1, w2)
int conv3x3_split_launch(int np, const float* x, const scan_pyramid_t* d, int32_t Cs, const void* w0, const void* w1,
                         const void* w2, int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout,
                         int32_t Ns, int32_t relu, void* stream, double* gn_ws) {
  ConvArgs a{x, d, d, Cs, {reinterpret_cast<const __bf16*>(w0), reinterpret_cast<const __bf16*>(w1),
                           reinterpret_cast<const __bf16*>(w2)}, Csw, bias, mask, y, Nout, Ns, relu, 0, as_stream(stream), gn_ws};
  const bool whole = g_scan_conv_glds && Csw % 32 == 0;
  const bool th16 = g_scan_conv_bn64_th16 && d->n_levels == 1 && d->h[0] % 16 == 0 && d->w[0] % 16 == 0;
  if (np == 3) {
    // three pieces: every 3x3 instance stages its weight tiles by LDS-DMA when the planes have whole K chunks (register
    // staging of three planes costs the 8-wave instances 24 registers and spills: 95-103 TFLOP/s against 215-246,
    // tools/conv_bench.py --variants conv_glds=0)
    switch (v2_instance(d, Nout)) {
      case 64:
        if (whole && th16 && g_scan_conv_bn64_th16 == 2 && d->h[0] % 32 == 0)
          launch_v2<3, 64, 32, 512, 3, 1, true>(a);
        else if (whole && th16)
          launch_v2<3, 64, 16, 512, 3, 1, true>(a);
        else if (whole)
          launch_v2<3, 64, 8, 256, 3, 1, true>(a);
        else if (th16)
          launch_v2<3, 64, 16, 512, 3>(a);
        else
          launch_v2<3, 64, 8, 256, 3>(a);
        break;
      case 256:
        if (whole)
          launch_v2<3, 256, 16, 512, 3, 1, true>(a);
        else
          launch_v2<3, 256, 16, 512, 3>(a);
        break;
      default:
        if (whole)
          launch_v2<3, 128, 16, 512, 3, 1, true>(a);
        else
          launch_v2<3, 128, 16, 512, 3>(a);
        break;
    }
    SCAN_LAUNCH_CHECK("conv3x3_bf16x6");
    return 0;
  }
  switch (v2_instance(d, Nout)) {
    case 64:
      if (th16)
        launch_v2<2, 64, 16, 256, 3>(a);
      else if (g_scan_conv_tpb3 & 2)
        launch_v2<2, 64, 8, 256, 3, 3>(a);
      else
        launch_v2<2, 64, 8, 256, 3>(a);
      break;
    case 256:
      // (two pieces: LDS-DMA only on whole channel tiles, as measured in rounds 2 and 3; v2_instance returns 256 only then)
      if (g_scan_conv_w8 && g_scan_conv_wg1024 == 1 && whole)
        launch_v2<2, 256, 16, 512, 3, 1, true>(a);
      else if ((g_scan_conv_wg1024 == 1 || (g_scan_conv_wg1024 == 2 && d->n_levels > 1)) && whole)
        launch_v2<2, 256, 16, 1024, 3, 1, true>(a);
      else if (g_scan_conv_wg1024 == 1 || (g_scan_conv_wg1024 == 2 && d->n_levels > 1))
        launch_v2<2, 256, 16, 1024, 3>(a);
      else
        launch_v2<2, 256, 16, 512, 3>(a);
      break;
    default:
      if (g_scan_conv_wg1024 == 1 && (g_scan_conv_tpb3 & 1) && whole && Nout % 128 == 0)
        launch_v2<2, 128, 16, 1024, 3, 3, true>(a);
      else if (g_scan_conv_wg1024 == 1 && (g_scan_conv_tpb3 & 1))
        launch_v2<2, 128, 16, 1024, 3, 3>(a);
      else if (g_scan_conv_wg1024 == 1 && whole && Nout % 128 == 0)
        launch_v2<2, 128, 16, 1024, 3, 1, true>(a);
      else if (g_scan_conv_wg1024 == 1)
        launch_v2<2, 128, 16, 1024, 3>(a);
      else if (g_scan_conv_tpb3 & 1)
        launch_v2<2, 128, 16, 512, 3, 3>(a);
      else
        launch_v2<2, 128, 16, 512, 3>(a);
      break;
  }
  SCAN_LAUNCH_CHECK("conv3x3_bf16x3");
  return 0;
}

// instance of a three-piece 1x1 launch: 64 / 128 = the register-staged tiles, 1128 / 1256 = the 128- / 256-channel tile with LDS-DMA weights
static int conv1x1_instance3(const scan_pyramid_t* yd, int32_t Nout, int32_t Csw);
// scan_tune "conv1x1" (three pieces): bit 0 = the 1x1 instances stage their weight tiles by LDS-DMA (whole K chunks), bit 1 = the
// 256-channel tile when the channels fill it and it does not cost a round of 256 CUs.  Default 3 (round 6).  Same results bit for bit
// (tools/conv_bench.py --ksize 1 --variants conv1x1=0,conv1x1=1,conv1x1=3, ResNet-50 body + FPN laterals at the K2C bench shape, us):
// forward 64 -> 256 @256x512 259 -> 217, 128 -> 512 172 -> 146, 256 -> 1024 120 -> 108, 512 -> 2048 111 -> 95, lateral 512 -> 256 225 -> 181;
// data gradient 256 -> 64 267 -> 213, 512 -> 128 166 -> 136, 1024 -> 256 111 -> 96, 2048 -> 512 98 -> 84, lateral 259 -> 202; Cout <= 128: +- 2 %.
int g_scan_conv1x1 = 3;
static int conv1x1_instance3(const scan_pyramid_t* yd, int32_t Nout, int32_t Csw) {
  if (Nout <= 64) return 64;
  if (!((g_scan_conv1x1 & 1) && Csw % 32 == 0)) return 128;
  if ((g_scan_conv1x1 & 2) && Nout % 256 == 0) {
    TileTab2 tt;
    make_tiles_v2(yd, &tt, 16);
    const int64_t tiles = tt.tile_off[yd->n_levels];
    const int64_t r256 = (tiles * (Nout / 256) + 255) / 256 * 2, r128 = (tiles * (Nout / 128) + 255) / 256;
    if (r256 * 97 <= r128 * 100) return 1256;
  }
  return 1128;
}
extern "C" int scan_conv1x1_bf16x6_instance(const scan_pyramid_t* yd, int32_t Nout, int32_t Csw) {
  return yd ? conv1x1_instance3(yd, Nout, Csw) : -1;
}

int conv1x1_split_launch(int np, const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* w0, const void* w1,
                         const void* w2, int32_t Csw, const float* bias, const float* mask, float* y,
                         const scan_pyramid_t* yd, int32_t Nout, int32_t Ns, int32_t relu, int32_t map, void* stream) {
  ConvArgs a{x, yd, xd, Cs, {reinterpret_cast<const __bf16*>(w0), reinterpret_cast<const __bf16*>(w1),
                             reinterpret_cast<const __bf16*>(w2)}, Csw, bias, mask, y, Nout, Ns, relu, map, as_stream(stream), nullptr};
  // scan_tune "conv1x1": bit 0 = weight tiles by LDS-DMA (whole K chunks: Csw % 32 == 0), bit 1 = the 256-channel tile when the
  // channels fill it and it does not cost a round of 256 CUs (the 3x3 rule, v2_instance).  1x1 convs are a sliver of the VGG step (FPN
  // laterals) but a fifth of the ResNet-50 body's (BASELINE.json configs[3]: bottleneck conv1 / conv3 / downsample).
  const int inst = np == 3 ? conv1x1_instance3(yd, Nout, Csw) : (Nout <= 64 ? 64 : 128);
  const bool wide = inst == 1256, gl = inst >= 1000;
  if (np == 3) {
    if (Nout <= 64)
      launch_v2<3, 64, 8, 256, 1>(a);
    else if (wide)
      launch_v2<3, 256, 16, 512, 1, 1, true>(a);
    else if (gl)
      launch_v2<3, 128, 16, 512, 1, 1, true>(a);
    else
      launch_v2<3, 128, 16, 512, 1>(a);
  } else {
    if (Nout <= 64)
      launch_v2<2, 64, 8, 256, 1>(a);
    else
      launch_v2<2, 128, 16, 512, 1>(a);
  }
  SCAN_LAUNCH_CHECK("conv1x1_split");
  return 0;
}

// which instance a 3x3 launch on pyramid d with Nout output channels takes (bench.py labels its timings with it):
// two pieces: 64 / 128 / 256, or 1128 / 1256 for the 128- / 256-channel tile run by 16-wave (1024-thread) workgroups, 2256
// for the 256-channel tile on the 8-wave LDS-DMA instance (Csw % 32 == 0 assumed: true for every 3x3 plane ops.py splits)
extern "C" int scan_conv3x3_bf16x3_instance(const scan_pyramid_t* d, int32_t Nout) {
  if (!d) return -1;
  const int bn = v2_instance(d, Nout);
  if (bn == 256 && g_scan_conv_w8 && g_scan_conv_wg1024 == 1 && g_scan_conv_glds) return 2256;
  if (bn == 256 && (g_scan_conv_wg1024 == 1 || (g_scan_conv_wg1024 == 2 && d->n_levels > 1))) return 1256;
  if (bn == 128 && g_scan_conv_wg1024 == 1) return 1128;
  return bn;
}

// three pieces: the output-channel tile (64 / 128 / 256; 8 waves, LDS-DMA weight tiles), 1064 = the 64-channel tile on
// 16x16-pixel tiles with 8 waves (single-level pyramids with sizes that are multiples of 16), 2064 = on 32x16-pixel tiles
// (H a multiple of 32), 64 = on 8x16-pixel tiles, 4 waves
extern "C" int scan_conv3x3_bf16x6_instance(const scan_pyramid_t* d, int32_t Nout) {
  if (!d) return -1;
  const int bn = v2_instance(d, Nout);
  if (bn == 64 && g_scan_conv_bn64_th16 && d->n_levels == 1 && d->h[0] % 16 == 0 && d->w[0] % 16 == 0)
    return (g_scan_conv_bn64_th16 == 2 && d->h[0] % 32 == 0) ? 2064 : 1064;  // (Csw % 32 == 0 assumed, as for the others)
  return bn;
}
