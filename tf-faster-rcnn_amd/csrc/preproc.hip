// Image preprocessing on device: the step right before the path -- lib/model/test.py:26-58
// (_get_image_blob) / lib/utils/blob.py:33-47 (prep_im_for_blob): BGR uint8 -> float32, minus PIXEL_MEANS, bilinear
// resize by im_scale with cv2.resize(..., fx, fy, INTER_LINEAR) semantics, written straight into the staged NHWC
// buffer the stem reads (C_out = 4 with a zero 4th channel, or 3).  One thread per output pixel, 3 B/pixel read
// (4 taps, L2-resident) and 16 B written: HBM-bound, 7.2 MB + 9.6 MB per 600x1000 image.
//
// cv2.resize is third-party code absent from the reference tree (OpenCV, unpinned): restated from OpenCV 3.x
// modules/imgproc/src/resize.cpp (resizeGeneric_ / HResizeLinear / VResizeLinear, 32f path):
//   scale = 1/fx (NOT src/dst);  dst size = cvRound(src * fx) (round half to even);
//   fx_d = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx_d); fx_d -= sx;  sx < 0 -> (0, 0);  sx >= w-1 -> (w-1, 0)
//   rows: sy likewise but the WEIGHT is kept and the two row indices are clamped to [0, h-1]
//   value = (S[y0][x0]*(1-fx) + S[y0][x1]*fx) * (1-fy) + (S[y1][x0]*(1-fx) + S[y1][x1]*fx) * fy, every op rounded to f32
//   (columns right of the last interpolable one: S[y][x0] * 1.0f).
#include "common.h"

// np.round / cvRound: round half to even
static inline long long round_half_even(double v) { return (long long)nearbyint(v); }

extern "C" int frcnn_prep_image_shape(int h, int w, int target_size, int max_size, double* im_scale, int* out_h, int* out_w) {
  if (h <= 0 || w <= 0 || target_size <= 0 || max_size <= 0 || !im_scale || !out_h || !out_w) return FRCNN_E_ARG;
  const int smin = h < w ? h : w, smax = h < w ? w : h;
  double s = (double)target_size / (double)smin;                                       // test.py:45
  if ((double)round_half_even(s * (double)smax) > (double)max_size) s = (double)max_size / (double)smax;   // :47-48 (np.round)
  *im_scale = s;
  *out_h = (int)round_half_even((double)h * s);                                        // cv2: saturate_cast<int>(src * fx)
  *out_w = (int)round_half_even((double)w * s);
  return FRCNN_OK;
}

struct Mean3 { double b, g, r; };

template <typename SRC>
__global__ void k_prep_image(const SRC* __restrict__ src, int h, int w, Mean3 mean, double scale_inv, int OH, int OW, int OC,
                             float* __restrict__ out) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (ox >= OW) return;
  float fx = (float)(((double)ox + 0.5) * scale_inv - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  bool last = false;                         // dx >= xmax: the single-tap tail of HResizeLinear
  if (sx + 1 >= w) {
    last = true;
    if (sx >= w - 1) { fx = 0.f; sx = w - 1; }
  }
  float fy = (float)(((double)oy + 0.5) * scale_inv - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  const int y0 = min(max(sy, 0), h - 1), y1 = min(max(sy + 1, 0), h - 1);
  const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
  const int x1 = last ? sx : sx + 1;
  const double m[3] = {mean.b, mean.g, mean.r};
  float* o = out + ((size_t)oy * OW + ox) * OC;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    // im.astype(float32) - PIXEL_MEANS (float64) assigned in place to float32: float64 subtraction, one rounding
    const float p00 = (float)((double)src[((size_t)y0 * w + sx) * 3 + c] - m[c]);
    const float p01 = (float)((double)src[((size_t)y0 * w + x1) * 3 + c] - m[c]);
    const float p10 = (float)((double)src[((size_t)y1 * w + sx) * 3 + c] - m[c]);
    const float p11 = (float)((double)src[((size_t)y1 * w + x1) * 3 + c] - m[c]);
    const float r0 = last ? p00 * 1.0f : p00 * a0 + p01 * a1;
    const float r1 = last ? p10 * 1.0f : p10 * a0 + p11 * a1;
    o[c] = r0 * b0 + r1 * b1;
  }
  if (OC == 4) o[3] = 0.f;
}

// src_d: BGR [h][w][3], uint8 (src_is_float = 0) or float32 (1).  pixel_means: HOST double[3] (B,G,R; config.py PIXEL_MEANS).
// out_d: float32 [OH][OW][out_c], out_c = 3 or 4 (4th channel zero), OH/OW from frcnn_prep_image_shape.
extern "C" int frcnn_prep_image(const void* src_d, int src_is_float, int h, int w, const double* pixel_means, double im_scale,
                                float* out_d, int OH, int OW, int out_c, void* stream) {
  if (!src_d || !pixel_means || !out_d || h <= 0 || w <= 0 || OH <= 0 || OW <= 0 || !(im_scale > 0)) return FRCNN_E_ARG;
  if (out_c != 3 && out_c != 4) return FRCNN_E_UNSUPPORTED;
  const Mean3 mean = {pixel_means[0], pixel_means[1], pixel_means[2]};
  const dim3 grid((OW + 255) / 256, OH), block(256);
  if (src_is_float)
    hipLaunchKernelGGL(k_prep_image<float>, grid, block, 0, (hipStream_t)stream, (const float*)src_d, h, w, mean, 1.0 / im_scale, OH, OW,
                       out_c, out_d);
  else
    hipLaunchKernelGGL(k_prep_image<unsigned char>, grid, block, 0, (hipStream_t)stream, (const unsigned char*)src_d, h, w, mean,
                       1.0 / im_scale, OH, OW, out_c, out_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
