// Input pipeline of the hot path's caller on gfx950: the reference's per-image transforms
// (fcos_core/data/transforms/transforms.py:27-90 -- Resize, RandomHorizontalFlip, ToTensor, Normalize) and the batch
// collation (data/collate_batch.py:5-20 -> structures/image_list.py:29-72) as byte / float streaming kernels.
//
// Resize: torchvision's F.resize on a PIL image is Image.resize(size, BILINEAR), i.e. Pillow's two-pass separable
// resampler (src/libImaging/Resample.c: ImagingResampleHorizontal_8bpc then ImagingResampleVertical_8bpc) with a
// triangle filter whose support grows with the down-scale factor, fixed-point coefficients (PRECISION_BITS = 22) and a
// uint8-rounded intermediate image.  The coefficient tables are computed by the host in double precision exactly as
// precompute_coeffs does (scan_amd/data.py) and passed in; the kernels do the integer arithmetic:
//      acc = 1 << 21;  acc += pixel * coeff ...;  out = clip8(acc >> 22)
// so the result is bit-exact against PIL for any size.  HBM-bound byte work: one output byte per lane, consecutive
// lanes on consecutive bytes of a row (channels interleaved), the taps of a pixel re-read through L1/L2.
//
// Normalise + collate: uint8 HWC RGB -> fp32, (x / 255) [ToTensor] -> channel swap and * 255 [TO_BGR255] -> (x - mean) /
// std [Normalize], optional horizontal flip, written either as the reference's CHW tensor (parity checks) or straight
// into the zero-padded NHWC4 batch the first convolution reads (image slot of a [N, Hp, Wp, 4] buffer: the batch
// collation costs no extra pass).  Every float op is a separate IEEE operation in the reference's order (this file
// is compiled with -ffp-contract=off), so the tensor is bit-identical to the torch-CPU result.
#include "common.h"

#define RS_PRECISION_BITS 22

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= RS_PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [H][W][3] -> dst [H][OW][3];  bounds [OW][2] = (xmin, count), coef [OW][ksize]
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ src, int H, int W,
                                                       unsigned char* __restrict__ dst, int OW,
                                                       const int* __restrict__ bounds, const int* __restrict__ coef,
                                                       int ksize) {
  const int64_t total = (int64_t)H * OW * 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % 3);
    const int64_t p = i / 3;
    const int ox = (int)(p % OW);
    const int64_t y = p / OW;
    const int xmin = bounds[2 * ox], cnt = bounds[2 * ox + 1];
    const int* k = coef + (int64_t)ox * ksize;
    const unsigned char* row = src + (y * W + xmin) * 3 + ch;
    int acc = 1 << (RS_PRECISION_BITS - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)row[3 * x] * k[x];
    dst[i] = clip8(acc);
  }
}

// vertical pass: src [H][OW][3] -> dst [OH][OW][3];  bounds [OH][2] = (ymin, count), coef [OH][ksize]
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* __restrict__ src, int H, int OW,
                                                       unsigned char* __restrict__ dst, int OH,
                                                       const int* __restrict__ bounds, const int* __restrict__ coef,
                                                       int ksize) {
  const int64_t rowlen = (int64_t)OW * 3;
  const int64_t total = (int64_t)OH * rowlen;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t oy = i / rowlen, col = i - oy * rowlen;
    const int ymin = bounds[2 * oy], cnt = bounds[2 * oy + 1];
    const int* k = coef + oy * ksize;
    const unsigned char* p = src + (int64_t)ymin * rowlen + col;
    int acc = 1 << (RS_PRECISION_BITS - 1);
    for (int y = 0; y < cnt; ++y) acc += (int)p[(int64_t)y * rowlen] * k[y];
    dst[i] = clip8(acc);
  }
}

extern "C" int scan_resize_bilinear_u8(const uint8_t* src, int32_t H, int32_t W, uint8_t* tmp, uint8_t* dst, int32_t OH,
                                       int32_t OW, const int32_t* xbounds, const int32_t* xcoef, int32_t kx,
                                       const int32_t* ybounds, const int32_t* ycoef, int32_t ky, void* stream) {
  SCAN_CHECK_ARG(H > 0 && W > 0 && OH > 0 && OW > 0, "resize_bilinear_u8: bad sizes %dx%d -> %dx%d", H, W, OH, OW);
  SCAN_CHECK_ARG(src && dst, "resize_bilinear_u8: null image");
  const bool need_h = OW != W, need_v = OH != H;
  SCAN_CHECK_ARG(!need_h || (xbounds && xcoef && kx > 0), "resize_bilinear_u8: horizontal tables missing");
  SCAN_CHECK_ARG(!need_v || (ybounds && ycoef && ky > 0), "resize_bilinear_u8: vertical tables missing");
  SCAN_CHECK_ARG(!(need_h && need_v) || tmp, "resize_bilinear_u8: two passes need the [H, OW, 3] intermediate");
  hipStream_t st = as_stream(stream);
  if (!need_h && !need_v) {  // Image.resize to the same size returns a copy
    if (hipMemcpyAsync(dst, src, (size_t)H * W * 3, hipMemcpyDeviceToDevice, st) != hipSuccess) {
      scan_set_error("resize_bilinear_u8: copy failed");
      return -2;
    }
    return 0;
  }
  const uint8_t* vin = src;
  if (need_h) {
    uint8_t* hout = need_v ? tmp : dst;
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((int64_t)H * OW * 3, 256)), dim3(256), 0, st, src, H, W, hout, OW,
                       xbounds, xcoef, kx);
    SCAN_LAUNCH_CHECK("resize_h");
    vin = hout;
  }
  if (need_v) {
    hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((int64_t)OH * OW * 3, 256)), dim3(256), 0, st, vin, H, OW, dst, OH,
                       ybounds, ycoef, ky);
    SCAN_LAUNCH_CHECK("resize_v");
  }
  return 0;
}

// one thread per DESTINATION pixel of the padded slot (Hp x Wp); pixels outside the image are the collator's zeros
__global__ __launch_bounds__(256) void normalize_kernel(const unsigned char* __restrict__ src, int H, int W, int flip,
                                                        int to_bgr255, float m0, float m1, float m2, float s0, float s1,
                                                        float s2, float* __restrict__ dst, int Hp, int Wp, int layout) {
  const int64_t total = (int64_t)Hp * Wp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / Wp), x = (int)(i - (int64_t)y * Wp);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (y < H && x < W) {
      const unsigned char* p = src + ((int64_t)y * W + (flip ? W - 1 - x : x)) * 3;
      // ToTensor: uint8 -> float, / 255
      float r = (float)p[0] / 255.0f, g = (float)p[1] / 255.0f, b = (float)p[2] / 255.0f;
      float c0 = r, c1 = g, c2 = b;
      if (to_bgr255) {  // image[[2, 1, 0]] * 255
        c0 = b * 255.0f;
        c1 = g * 255.0f;
        c2 = r * 255.0f;
      }
      v0 = (c0 - m0) / s0;  // F.normalize: (t - mean) / std
      v1 = (c1 - m1) / s1;
      v2 = (c2 - m2) / s2;
    }
    if (layout == 0) {  // CHW planes of the slot
      dst[i] = v0;
      dst[total + i] = v1;
      dst[2 * total + i] = v2;
    } else {  // NHWC4 rows: what the first convolution reads
      reinterpret_cast<float4*>(dst)[i] = make_float4(v0, v1, v2, 0.f);
    }
  }
}

extern "C" int scan_normalize_image_u8(const uint8_t* src, int32_t H, int32_t W, int32_t flip, int32_t to_bgr255,
                                       const float* mean3, const float* std3, float* dst, int32_t Hp, int32_t Wp,
                                       int32_t layout, void* stream) {
  SCAN_CHECK_ARG(H > 0 && W > 0 && Hp >= H && Wp >= W, "normalize_image_u8: bad sizes %dx%d in a %dx%d slot", H, W, Hp, Wp);
  SCAN_CHECK_ARG(src && dst && mean3 && std3, "normalize_image_u8: null pointer");
  SCAN_CHECK_ARG(layout == 0 || layout == 1, "normalize_image_u8: layout must be 0 (CHW) or 1 (NHWC4 rows)");
  SCAN_CHECK_ARG(layout == 0 || (reinterpret_cast<uintptr_t>(dst) & 15) == 0, "normalize_image_u8: NHWC4 rows need a 16-byte aligned slot");
  hipLaunchKernelGGL(normalize_kernel, dim3(grid_for((int64_t)Hp * Wp, 256)), dim3(256), 0, as_stream(stream), src, H, W,
                     flip, to_bgr255, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], dst, Hp, Wp, layout);
  SCAN_LAUNCH_CHECK("normalize_image");
  return 0;
}
