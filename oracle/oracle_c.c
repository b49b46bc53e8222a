/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's native CPU kernels.
 * Never linked into the product library; only loaded by oracle/frcnn_oracle.py.
 * Citations are relative to /root/reference/lib.  Build: see oracle/Makefile
 * (-O2 -ffp-contract=off: every float op keeps its own rounding, like the x86 Cython build). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float fmax_(float a, float b) { return a >= b ? a : b; }   /* cpu_nms.pyx:11-12 */
static inline float fmin_(float a, float b) { return a <= b ? a : b; }   /* cpu_nms.pyx:14-15 */

/* nms/cpu_nms.pyx:17-68.  dets [k,5] f32 (x1,y1,x2,y2,score); `order` = indices by descending
 * score (computed by the caller: the argsort tie policy lives in frcnn_oracle.order_desc).
 * keep_out receives kept ORIGINAL indices in score order.  Suppression test is done in double
 * (Cython compares the float32 ovr against a Python float: cpu_nms.c:2239-2241). */
int oracle_cpu_nms(const float* dets, int k, const int64_t* order, double thresh, int64_t* keep_out) {
  float* areas = (float*)malloc(sizeof(float) * (size_t)k);
  unsigned char* sup = (unsigned char*)calloc((size_t)k, 1);
  for (int i = 0; i < k; ++i) {
    const float* d = dets + 5 * (size_t)i;
    areas[i] = ((d[2] - d[0]) + 1.0f) * ((d[3] - d[1]) + 1.0f);         /* :24 */
  }
  int nkeep = 0;
  for (int _i = 0; _i < k; ++_i) {
    const int64_t i = order[_i];
    if (sup[i]) continue;
    keep_out[nkeep++] = i;
    const float ix1 = dets[5 * i], iy1 = dets[5 * i + 1], ix2 = dets[5 * i + 2], iy2 = dets[5 * i + 3];
    const float iarea = areas[i];
    for (int _j = _i + 1; _j < k; ++_j) {
      const int64_t j = order[_j];
      if (sup[j]) continue;
      const float xx1 = fmax_(ix1, dets[5 * j]), yy1 = fmax_(iy1, dets[5 * j + 1]);
      const float xx2 = fmin_(ix2, dets[5 * j + 2]), yy2 = fmin_(iy2, dets[5 * j + 3]);
      const float w = fmax_(0.0f, (xx2 - xx1) + 1.0f);
      const float h = fmax_(0.0f, (yy2 - yy1) + 1.0f);
      const float inter = w * h;
      const float ovr = inter / ((iarea + areas[j]) - inter);            /* :64 */
      if ((double)ovr >= thresh) sup[j] = 1;                              /* :65 */
    }
  }
  free(areas);
  free(sup);
  return nkeep;
}

/* utils/bbox.pyx:15-55.  boxes [n,4] f64, query [k,qstride] f64 (cols 0..3 used) -> out [n,k] f64 */
void oracle_bbox_overlaps(const double* boxes, int n, const double* query, int k, int qstride, double* out) {
  for (int kk = 0; kk < k; ++kk) {
    const double* q = query + (size_t)kk * qstride;
    const double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
    for (int nn = 0; nn < n; ++nn) {
      const double* b = boxes + 4 * (size_t)nn;
      double o = 0.0;
      const double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
      if (iw > 0) {
        const double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
        if (ih > 0) {
          const double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
          o = iw * ih / ua;
        }
      }
      out[(size_t)nn * k + kk] = o;
    }
  }
}

/* tf.image.crop_and_resize (bilinear, extrapolation_value 0) as called from
 * nets/resnet_v1.py:55-76 and nets/network.py:141-157.  THIRD-PARTY semantics (TensorFlow r1.2
 * CropAndResize CPU functor), restated from its published definition -- parity unpinned.
 * Box normalisation follows network.py:146-151: x/((W-1)*stride), y/((H-1)*stride), in f32.
 * feat [H,W,C] f32, rois [R,5] (batch,x1,y1,x2,y2) image coords, out [R,P,P,C]. */
void oracle_crop_and_resize(const float* feat, int H, int W, int C, const float* rois, int R,
                            float stride, int P, float* out) {
  const float height = ((float)H - 1.0f) * stride, width = ((float)W - 1.0f) * stride;
  for (int r = 0; r < R; ++r) {
    const float x1 = rois[5 * r + 1] / width, y1 = rois[5 * r + 2] / height;
    const float x2 = rois[5 * r + 3] / width, y2 = rois[5 * r + 4] / height;
    const float hs = (P > 1) ? (y2 - y1) * (float)(H - 1) / (float)(P - 1) : 0.0f;
    const float ws = (P > 1) ? (x2 - x1) * (float)(W - 1) / (float)(P - 1) : 0.0f;
    for (int y = 0; y < P; ++y) {
      float* orow = out + (((size_t)r * P + y) * P) * C;
      const float in_y = (P > 1) ? y1 * (float)(H - 1) + (float)y * hs : 0.5f * (y1 + y2) * (float)(H - 1);
      if (in_y < 0 || in_y > (float)(H - 1)) { memset(orow, 0, sizeof(float) * (size_t)P * C); continue; }
      const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
      const float ly = in_y - (float)top;
      for (int x = 0; x < P; ++x) {
        float* o = orow + (size_t)x * C;
        const float in_x = (P > 1) ? x1 * (float)(W - 1) + (float)x * ws : 0.5f * (x1 + x2) * (float)(W - 1);
        if (in_x < 0 || in_x > (float)(W - 1)) { memset(o, 0, sizeof(float) * (size_t)C); continue; }
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float lx = in_x - (float)left;
        const float* tl = feat + ((size_t)top * W + left) * C;
        const float* tr = feat + ((size_t)top * W + right) * C;
        const float* bl = feat + ((size_t)bot * W + left) * C;
        const float* br = feat + ((size_t)bot * W + right) * C;
        for (int c = 0; c < C; ++c) {
          const float t = tl[c] + (tr[c] - tl[c]) * lx;
          const float b = bl[c] + (br[c] - bl[c]) * lx;
          o[c] = t + (b - t) * ly;
        }
      }
    }
  }
}
