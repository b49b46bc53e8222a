// Bandwidth-bound dense helpers (NHWC f32, float4 along the channel axis) + stream-capture glue.
// Citations relative to /root/reference/lib.
#include "h2_common.h"

extern "C" int frcnn_abi_version(void) { return FRCNN_ABI_VERSION; }
extern "C" const char* frcnn_build_info(void) { return "libfrcnn_hip gfx950 (CDNA4, wave64, f32 / f16 / bf16 MFMA) abi 5"; }

// ---- max pool (nets/resnet_v1.py:83-84 pool1: pad 1 + 3x3/2 VALID; nets/vgg16.py:30-39 2x2/2 SAME;
//      nets/network.py:157).  Out-of-image taps are skipped, which equals TF's SAME behaviour and,
//      for the post-ReLU ResNet stem, the explicit zero pad.  ZERO_PAD makes the zero explicit.
template <bool ZERO_PAD>
__global__ void k_maxpool(const float4* __restrict__ x, int N, int H, int W, int C4, int k, int stride, int pt, int pl,
                          float4* __restrict__ y, int OH, int OW) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)N * OH * OW * C4;
  if (t >= tot) return;
  const int c4 = (int)(t % C4);
  long long pix = t / C4;
  const int ow = (int)(pix % OW); pix /= OW;
  const int oh = (int)(pix % OH);
  const int img = (int)(pix / OH);
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  bool any_oob = false;
  for (int dy = 0; dy < k; ++dy) {
    const int ih = oh * stride - pt + dy;
    for (int dx = 0; dx < k; ++dx) {
      const int iw = ow * stride - pl + dx;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
        const float4 v = x[((size_t)(img * H + ih) * W + iw) * C4 + c4];
        m = make_float4(fmaxf(m.x, v.x), fmaxf(m.y, v.y), fmaxf(m.z, v.z), fmaxf(m.w, v.w));
      } else {
        any_oob = true;
      }
    }
  }
  if (ZERO_PAD && any_oob) m = make_float4(fmaxf(m.x, 0.f), fmaxf(m.y, 0.f), fmaxf(m.z, 0.f), fmaxf(m.w, 0.f));
  y[t] = m;
}

extern "C" int frcnn_maxpool_nhwc(const float* x_d, int N, int H, int W, int C, int k, int stride, int pad_top,
                                  int pad_left, float* y_d, int OH, int OW, void* stream) {
  if (!x_d || !y_d || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || OH <= 0 || OW <= 0) return FRCNN_E_ARG;
  if (C % 4) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)N * OH * OW * (C / 4);
  const bool zero_pad = (pad_top > 0 || pad_left > 0);   // explicit tf.pad zeros (resnet_v1.py:83)
  if (zero_pad)
    hipLaunchKernelGGL(k_maxpool<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x_d, N, H, W, C / 4, k, stride, pad_top, pad_left, (float4*)y_d, OH, OW);
  else
    hipLaunchKernelGGL(k_maxpool<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x_d, N, H, W, C / 4, k, stride, pad_top, pad_left, (float4*)y_d, OH, OW);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- depthwise 3x3 (nets/mobilenet_v1.py:21-49): VALU / bandwidth bound by nature.
// H2: the result also / only leaves as frcnn_gemm_h2 operand planes (the pointwise convolution that follows reads nothing else): the
// 32 consecutive threads of a pixel's 128 channels are one half-wave (C4 % 32 == 0) and reduce the block maximum with DPP moves.
template <bool H2>
__global__ void k_dwconv3x3(const float4* __restrict__ x, int N, int H, int W, int C4, const float4* __restrict__ w,
                            const float4* __restrict__ bias, float4* __restrict__ y, unsigned short* __restrict__ planes,
                            float* __restrict__ inv, int OH, int OW, int stride, int pt, int pl, int act) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)N * OH * OW * C4;
  if (t >= tot) return;
  const int c4 = (int)(t % C4);
  long long pix = t / C4;
  const size_t row = (size_t)pix;
  const int ow = (int)(pix % OW); pix /= OW;
  const int oh = (int)(pix % OH);
  const int img = (int)(pix / OH);
  float4 a = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int dy = 0; dy < 3; ++dy) {
    const int ih = oh * stride - pt + dy;
    if ((unsigned)ih >= (unsigned)H) continue;
    for (int dx = 0; dx < 3; ++dx) {
      const int iw = ow * stride - pl + dx;
      if ((unsigned)iw >= (unsigned)W) continue;
      const float4 v = x[((size_t)(img * H + ih) * W + iw) * C4 + c4];
      const float4 f = w[(dy * 3 + dx) * C4 + c4];
      a.x = fmaf(v.x, f.x, a.x); a.y = fmaf(v.y, f.y, a.y); a.z = fmaf(v.z, f.z, a.z); a.w = fmaf(v.w, f.w, a.w);
    }
  }
  if (act == FRCNN_ACT_RELU) a = act_relu(a);
  else if (act == FRCNN_ACT_RELU6)
    a = act_relu6(a);
  if (!H2 || y) y[t] = a;
  if (H2) {
    const size_t rows = (size_t)N * OH * OW, e = (row * C4 + c4) * 4;
    float* slot = inv + (size_t)(c4 >> 5) * rows + row;
    h2_emit_rows32<1>(&a, &e, &slot, 1u, planes, rows * (size_t)C4 * 4, c4 & 31);
  }
}

extern "C" int frcnn_dwconv3x3_nhwc(const float* x_d, int N, int H, int W, int C, const float* w_d, const float* bias_d,
                                    float* y_d, int OH, int OW, int stride, int pad_top, int pad_left, int act, void* stream) {
  if (!x_d || !w_d || !y_d || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || stride <= 0) return FRCNN_E_ARG;
  if (C % 4) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)N * OH * OW * (C / 4);
  hipLaunchKernelGGL(k_dwconv3x3<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x_d,
                     N, H, W, C / 4, (const float4*)w_d, (const float4*)bias_d, (float4*)y_d, (unsigned short*)nullptr, (float*)nullptr, OH, OW,
                     stride, pad_top, pad_left, act);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ... with the result as operand planes [2][N*OH*OW][C] + y_inv [C/128][N*OH*OW] for the pointwise convolution that follows
// (mobilenet_v1.py:21-49: depthwise -> BN -> ReLU6 -> pointwise); y_d NULL = planes only.  C % 128 == 0.
extern "C" int frcnn_dwconv3x3_nhwc_h2(const float* x_d, int N, int H, int W, int C, const float* w_d, const float* bias_d,
                                       float* y_d, void* y_planes_d, float* y_inv_d, int OH, int OW, int stride, int pad_top,
                                       int pad_left, int act, void* stream) {
  if (!x_d || !w_d || !y_planes_d || !y_inv_d || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || stride <= 0) return FRCNN_E_ARG;
  if (C % H2_KB) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)N * OH * OW * (C / 4);
  hipLaunchKernelGGL(k_dwconv3x3<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x_d,
                     N, H, W, C / 4, (const float4*)w_d, (const float4*)bias_d, (float4*)y_d, (unsigned short*)y_planes_d, y_inv_d, OH, OW,
                     stride, pad_top, pad_left, act);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- spatial mean [N,HW,C] -> [N,C] (nets/resnet_v1.py:124).  One thread per (n, c4): the HW
//      loop strides by C so a wave reads 1 KiB contiguous runs.
__global__ void k_spatial_mean(const float4* __restrict__ x, int N, int HW, int C4, float4* __restrict__ y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * C4) return;
  const int n = t / C4, c4 = t % C4;
  const float4* p = x + (size_t)n * HW * C4 + c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < HW; ++i) {
    const float4 v = p[(size_t)i * C4];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float inv = 1.0f / (float)HW;
  y[t] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
}

extern "C" int frcnn_spatial_mean(const float* x_d, int N, int HW, int C, float* y_d, void* stream) {
  if (!x_d || !y_d || N <= 0 || HW <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % 4) return FRCNN_E_UNSUPPORTED;
  hipLaunchKernelGGL(k_spatial_mean, dim3(cdiv(N * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x_d, N,
                     HW, C / 4, (float4*)y_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- row softmax [R,C] with leading dimension ld (nets/network.py:80-86): one wave per row.
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ x, int R, int C, int ld, float* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* p = x + (size_t)row * ld;
  float m = -INFINITY;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, p[c]);
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += expf(p[c] - m);
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = expf(p[c] - m) / s;
}

extern "C" int frcnn_softmax_rows(const float* x_d, int R, int C, int ld, float* y_d, void* stream) {
  if (!x_d || !y_d || R <= 0 || C <= 0 || ld < C) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_softmax_rows, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x_d, R, C, ld, y_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- RPN pairwise softmax (nets/network.py:68-86,331-334): the reshape-to-2-channels trick pairs
//      channel a (bg) with channel A+a (fg) at every position.
__global__ void k_rpn_softmax(const float* __restrict__ score, int HW, int A, int ld, float* __restrict__ prob) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= HW * A) return;
  const int pix = t / A, a = t % A;
  const float bg = score[(size_t)pix * ld + a], fg = score[(size_t)pix * ld + A + a];
  const float m = fmaxf(bg, fg);
  const float e0 = expf(bg - m), e1 = expf(fg - m);
  const float s = e0 + e1;
  prob[(size_t)pix * 2 * A + a] = e0 / s;
  prob[(size_t)pix * 2 * A + A + a] = e1 / s;
}

extern "C" int frcnn_rpn_softmax(const float* score_d, int HW, int A, int ld, float* prob_d, void* stream) {
  if (!score_d || !prob_d || HW <= 0 || A <= 0 || ld < 2 * A) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_rpn_softmax, dim3(cdiv(HW * A, 256)), dim3(256), 0, (hipStream_t)stream, score_d, HW, A, ld, prob_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

__global__ void k_copy_cols(const float* __restrict__ src, int R, int ld_src, int col0, int cols, float* __restrict__ dst, int ld_dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * cols) return;
  const int r = t / cols, c = t % cols;
  dst[(size_t)r * ld_dst + c] = src[(size_t)r * ld_src + col0 + c];
}

extern "C" int frcnn_copy_cols(const float* src_d, int R, int ld_src, int col0, int cols, float* dst_d, int ld_dst, void* stream) {
  if (!src_d || !dst_d || R <= 0 || cols <= 0 || col0 < 0 || ld_src < col0 + cols || ld_dst < cols) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_copy_cols, dim3(cdiv(R * cols, 256)), dim3(256), 0, (hipStream_t)stream, src_d, R, ld_src, col0, cols, dst_d, ld_dst);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- stream capture: the whole per-image chain (about 150 launches) becomes one hipGraph, the
//      MI355X replacement for the reference's one-sess.run-per-image (nets/network.py:470-479).
extern "C" int frcnn_graph_begin(void* stream) {
  if (!stream) return FRCNN_E_ARG;       // the legacy default stream cannot be captured
  HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return FRCNN_OK;
}
extern "C" int frcnn_graph_end(void* stream, void** graph_exec_out) {
  if (!stream || !graph_exec_out) return FRCNN_E_ARG;
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return FRCNN_E_HIP(e);
  *graph_exec_out = (void*)ge;
  return FRCNN_OK;
}
extern "C" int frcnn_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return FRCNN_E_ARG;
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return FRCNN_OK;
}
extern "C" int frcnn_graph_destroy(void* graph_exec) {
  if (!graph_exec) return FRCNN_E_ARG;
  HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------------
// HOST: CRC-32C (Castagnoli, reflected 0x82F63B78), slicing by 8 -- the checksum of TensorFlow checkpoint data and index
// blocks (frcnn_hip/tensor_bundle.py; tensorflow/core/lib/hash/crc32c.cc).  crc = 0 to start, or a previous result to extend.
// ------------------------------------------------------------------------------------------------
extern "C" unsigned int frcnn_crc32c(const void* data, size_t n, unsigned int crc) {
  static unsigned int T[8][256];
  static bool ready = false;
  if (!ready) {
    for (unsigned int i = 0; i < 256; ++i) {
      unsigned int c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      T[0][i] = c;
    }
    for (unsigned int i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFFu];
    ready = true;
  }
  const unsigned char* p = (const unsigned char*)data;
  unsigned int c = crc ^ 0xFFFFFFFFu;
  while (n && ((size_t)p & 7)) { c = T[0][(c ^ *p++) & 0xFFu] ^ (c >> 8); --n; }
  while (n >= 8) {
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);
    v ^= (unsigned long long)c;
    c = T[7][v & 0xFF] ^ T[6][(v >> 8) & 0xFF] ^ T[5][(v >> 16) & 0xFF] ^ T[4][(v >> 24) & 0xFF] ^ T[3][(v >> 32) & 0xFF] ^
        T[2][(v >> 40) & 0xFF] ^ T[1][(v >> 48) & 0xFF] ^ T[0][(v >> 56) & 0xFF];
    p += 8;
    n -= 8;
  }
  while (n--) c = T[0][(c ^ *p++) & 0xFFu] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------------
// HOST: Snappy raw-format decompression (the block compression of LevelDB tables; V1 TensorFlow checkpoints written by
// early TensorFlow releases -- the slim ImageNet .ckpt files -- use it, which is what lib/model/train_val.py:108-113 warns
// about).  Format (google/snappy format_description.txt): varint32 uncompressed length, then elements tagged by the low two
// bits of the first byte: 00 literal (length-1 in the upper six bits, 60..63 = 1..4 length bytes follow), 01 copy with
// 11-bit offset (length 4..11), 10 copy with 16-bit offset, 11 copy with 32-bit offset (length-1 in the upper six bits).
// Returns the number of bytes written, or -1 on malformed input / insufficient capacity.
// ------------------------------------------------------------------------------------------------
extern "C" long long frcnn_snappy_uncompress(const unsigned char* src, size_t n, unsigned char* dst, size_t cap) {
  if (!src || (!dst && cap)) return -1;
  size_t ip = 0;
  unsigned long long ulen = 0;
  int shift = 0;
  for (;;) {
    if (ip >= n || shift > 28) return -1;
    const unsigned char c = src[ip++];
    ulen |= (unsigned long long)(c & 0x7F) << shift;
    if (!(c & 0x80)) break;
    shift += 7;
  }
  if (ulen > cap) return -1;
  size_t op = 0;
  while (ip < n) {
    const unsigned tag = src[ip++];
    size_t len, off;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2) + 1;
        if (len > 60) {
          const unsigned nb = (unsigned)len - 60;
          if (ip + nb > n) return -1;
          len = 0;
          for (unsigned b = 0; b < nb; ++b) len |= (size_t)src[ip + b] << (8 * b);
          len += 1;
          ip += nb;
        }
        if (ip + len > n || op + len > ulen) return -1;
        __builtin_memcpy(dst + op, src + ip, len);
        ip += len;
        op += len;
        continue;
      }
      case 1:
        if (ip + 1 > n) return -1;
        len = ((tag >> 2) & 7) + 4;
        off = ((size_t)(tag >> 5) << 8) | src[ip];
        ip += 1;
        break;
      case 2:
        if (ip + 2 > n) return -1;
        len = (tag >> 2) + 1;
        off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        break;
      default:
        if (ip + 4 > n) return -1;
        len = (tag >> 2) + 1;
        off = (size_t)src[ip] | ((size_t)src[ip + 1] << 8) | ((size_t)src[ip + 2] << 16) | ((size_t)src[ip + 3] << 24);
        ip += 4;
        break;
    }
    if (off == 0 || off > op || op + len > ulen) return -1;
    for (size_t k = 0; k < len; ++k) dst[op + k] = dst[op + k - off];     // byte-wise on purpose: copies may overlap (runs)
    op += len;
  }
  return op == ulen ? (long long)op : -1;
}
