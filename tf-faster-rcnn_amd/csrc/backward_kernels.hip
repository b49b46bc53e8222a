// Backward-pass helpers and the solver update (lib/model/train_val.py:116-153).
//
// Convolution gradients reuse the forward implicit-GEMM kernel (conv_igemm.hip) on re-laid-out
// operands -- both are "B^T" GEMMs with a contiguous reduction axis once the operands are transposed:
//   dgrad (stride 1):  dX = conv(dY, W')       W'[c][kh'][kw'][n] = W[n][KH-1-kh'][KW-1-kw'][c], pad' = K-1-pad
//   wgrad:             dW[n][(kh,kw,c)] = sum_m dY^T[n][m] * Xcol^T[(kh,kw,c)][m]
//                      = a 1x1 "conv" whose pixels are the Cout rows of dY^T, whose input channels are the M
//                      output pixels (padded to a multiple of 32) and whose filter bank is the transposed
//                      im2col of X: the result lands directly in the packed [Cout][KH][KW][Cin] layout.
// The kernels here produce those operands (tiled LDS transposes, float4 traffic) and the small
// elementwise / scatter pieces of the chain rule.  All bandwidth-bound.
#include "common.h"

// ---- out[c][m] = in[m][c]  (m < M, c < C), rows of `out` padded with zeros to Mp ----------------------
__global__ __launch_bounds__(256) void k_transpose_pad(const float* __restrict__ in, int M, int C, float* __restrict__ out, int Mp) {
  __shared__ float tile[32][33];
  const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, c = c0 + tx;
    tile[r][tx] = (m < M && c < C) ? in[(size_t)m * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, m = m0 + tx;
    if (c < C && m < Mp) out[(size_t)c * Mp + m] = tile[tx][r];
  }
}

extern "C" int frcnn_transpose_pad(const float* in_d, int M, int C, float* out_d, int Mp, void* stream) {
  if (!in_d || !out_d || M <= 0 || C <= 0 || Mp < M) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_transpose_pad, dim3(cdiv(Mp, 32), cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, in_d, M, C, out_d, Mp);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- transposed im2col: out[(kh*KW+kw)*Cin + c][m] = x[img, oh*s-pt+kh, ow*s-pl+kw, c] (0 outside) ----
__global__ __launch_bounds__(256) void k_im2col_t(const float* __restrict__ x, int N, int H, int W, int Cin, int OH, int OW,
                                                  int KH, int KW, int stride, int pt, int pl, float* __restrict__ out, int Mp) {
  __shared__ float tile[32][33];
  const int M = N * OH * OW;
  const int m0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tap = blockIdx.z;
  const int kh = tap / KW, kw = tap % KW;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, c = c0 + tx;
    float v = 0.f;
    if (m < M && c < Cin) {
      const int img = m / (OH * OW), rem = m % (OH * OW), oh = rem / OW, ow = rem % OW;
      const int ih = oh * stride - pt + kh, iw = ow * stride - pl + kw;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[((size_t)(img * H + ih) * W + iw) * Cin + c];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, m = m0 + tx;
    if (c < Cin && m < Mp) out[((size_t)tap * Cin + c) * Mp + m] = tile[tx][r];
  }
}

extern "C" int frcnn_im2col_t(const float* x_d, int N, int H, int W, int Cin, int OH, int OW, int KH, int KW, int stride,
                              int pad_top, int pad_left, float* out_d, int Mp, void* stream) {
  if (!x_d || !out_d || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return FRCNN_E_ARG;
  if (Mp < N * OH * OW) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_im2col_t, dim3(cdiv(Mp, 32), cdiv(Cin, 32), KH * KW), dim3(256), 0, (hipStream_t)stream, x_d, N, H, W, Cin,
                     OH, OW, KH, KW, stride, pad_top, pad_left, out_d, Mp);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- dgrad filter: out[c][KH-1-kh][KW-1-kw][n] = w[n][kh][kw][c] ------------------------------------
__global__ void k_flip_transpose(const float* __restrict__ w, int Cout, int KH, int KW, int Cin, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)Cout * KH * KW * Cin;
  if (t >= tot) return;
  const int n = (int)(t % Cout);                       // consecutive threads write consecutive n
  long long r = t / Cout;
  const int kw = (int)(r % KW); r /= KW;
  const int kh = (int)(r % KH);
  const int c = (int)(r / KH);
  out[t] = w[(((size_t)n * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Cin + c];
}

extern "C" int frcnn_flip_transpose_filter(const float* w_d, int Cout, int KH, int KW, int Cin, float* out_d, void* stream) {
  if (!w_d || !out_d || Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0) return FRCNN_E_ARG;
  const long long tot = (long long)Cout * KH * KW * Cin;
  hipLaunchKernelGGL(k_flip_transpose, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_d, Cout, KH, KW, Cin, out_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- dgrad for strided convs (only the two stride-2 3x3 convs of the ResNet head need it): gather form
__global__ void k_dgrad_strided(const float* __restrict__ dy, int N, int OH, int OW, int Cout, const float* __restrict__ w,
                                int KH, int KW, int Cin, int stride, int pt, int pl, float* __restrict__ dx, int H, int W,
                                int accumulate) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)N * H * W * Cin;
  if (t >= tot) return;
  const int c = (int)(t % Cin);
  long long pix = t / Cin;
  const int iw = (int)(pix % W); pix /= W;
  const int ih = (int)(pix % H);
  const int img = (int)(pix / H);
  float acc = 0.f;
  for (int kh = 0; kh < KH; ++kh) {
    const int oy = ih + pt - kh;
    if (oy < 0 || oy % stride) continue;
    const int oh = oy / stride;
    if (oh >= OH) continue;
    for (int kw = 0; kw < KW; ++kw) {
      const int ox = iw + pl - kw;
      if (ox < 0 || ox % stride) continue;
      const int ow = ox / stride;
      if (ow >= OW) continue;
      const float* g = dy + ((size_t)(img * OH + oh) * OW + ow) * Cout;
      const float* wf = w + ((size_t)kh * KW + kw) * Cin + c;          // w[n][kh][kw][c], stride over n = KH*KW*Cin
      for (int n = 0; n < Cout; ++n) acc = fmaf(g[n], wf[(size_t)n * KH * KW * Cin], acc);
    }
  }
  dx[t] = accumulate ? dx[t] + acc : acc;
}

extern "C" int frcnn_conv2d_dgrad_strided(const float* dy_d, int N, int OH, int OW, int Cout, const float* w_d, int KH, int KW,
                                          int Cin, int stride, int pad_top, int pad_left, float* dx_d, int H, int W,
                                          int accumulate, void* stream) {
  if (!dy_d || !w_d || !dx_d || N <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0 || stride <= 0) return FRCNN_E_ARG;
  const long long tot = (long long)N * H * W * Cin;
  hipLaunchKernelGGL(k_dgrad_strided, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy_d, N, OH, OW, Cout,
                     w_d, KH, KW, Cin, stride, pad_top, pad_left, dx_d, H, W, accumulate);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- elementwise chain-rule pieces ------------------------------------------------------------------
__global__ void k_relu_bwd(float4* __restrict__ g, const float4* __restrict__ y, long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = g[i];
  const float4 b = y[i];
  a.x = b.x > 0.f ? a.x : 0.f; a.y = b.y > 0.f ? a.y : 0.f; a.z = b.z > 0.f ? a.z : 0.f; a.w = b.w > 0.f ? a.w : 0.f;
  g[i] = a;
}
extern "C" int frcnn_relu_bwd(float* grad_d, const float* y_d, long long n, void* stream) {
  if (!grad_d || !y_d || n <= 0 || (n & 3)) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_relu_bwd, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)grad_d, (const float4*)y_d, n / 4);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// dst[img, oh*s, ow*s, :] (+)= src[img, oh, ow, :]   (gradient of slim `subsample` / identity skip)
__global__ void k_add_strided(const float4* __restrict__ src, int N, int OH, int OW, int C4, float4* __restrict__ dst, int H, int W,
                              int stride, int accumulate) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)N * OH * OW * C4;
  if (t >= tot) return;
  const int c4 = (int)(t % C4);
  long long pix = t / C4;
  const int ow = (int)(pix % OW); pix /= OW;
  const int oh = (int)(pix % OH);
  const int img = (int)(pix / OH);
  float4* d = dst + ((size_t)(img * H + oh * stride) * W + ow * stride) * C4 + c4;
  const float4 s = src[t];
  if (accumulate) { float4 v = *d; v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; *d = v; }
  else *d = s;
}
extern "C" int frcnn_add_strided(const float* src_d, int N, int OH, int OW, int C, float* dst_d, int H, int W, int stride,
                                 int accumulate, void* stream) {
  if (!src_d || !dst_d || N <= 0 || OH <= 0 || OW <= 0 || C <= 0 || (C & 3) || stride < 1) return FRCNN_E_ARG;
  if (H < (OH - 1) * stride + 1 || W < (OW - 1) * stride + 1) return FRCNN_E_ARG;
  const long long tot = (long long)N * OH * OW * (C / 4);
  hipLaunchKernelGGL(k_add_strided, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)src_d, N, OH,
                     OW, C / 4, (float4*)dst_d, H, W, stride, accumulate);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// gradient of the spatial mean: dx[n, i, c] = dy[n, c] / HW
__global__ void k_spatial_mean_bwd(const float4* __restrict__ dy, int N, int HW, int C4, float4* __restrict__ dx) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * HW * C4) return;
  const int c4 = (int)(t % C4);
  const int n = (int)(t / ((long long)HW * C4));
  const float inv = 1.0f / (float)HW;
  const float4 g = dy[(size_t)n * C4 + c4];
  dx[t] = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
}
extern "C" int frcnn_spatial_mean_bwd(const float* dy_d, int N, int HW, int C, float* dx_d, void* stream) {
  if (!dy_d || !dx_d || N <= 0 || HW <= 0 || C <= 0 || (C & 3)) return FRCNN_E_ARG;
  const long long tot = (long long)N * HW * (C / 4);
  hipLaunchKernelGGL(k_spatial_mean_bwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)dy_d, N,
                     HW, C / 4, (float4*)dx_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// bias gradient: db[c] = sum_m dy[m][c]  (one wave-column sweep per 64 channels, f64 accumulation, deterministic)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dy, int M, int C, float* __restrict__ db) {
  __shared__ double sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  double s = 0.0;
  if (c < C) for (int m = part; m < M; m += 4) s += (double)dy[(size_t)m * C + c];
  sh[part][threadIdx.x & 63] = s;
  __syncthreads();
  if (part == 0 && c < C) db[c] = (float)(sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}
extern "C" int frcnn_colsum(const float* dy_d, int M, int C, float* db_d, void* stream) {
  if (!dy_d || !db_d || M <= 0 || C <= 0) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_colsum, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, dy_d, M, C, db_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- crop_and_resize backward w.r.t. the feature map (TF CropAndResizeGradImage semantics): every
//      output sample scatters its gradient to its four bilinear taps; out-of-range samples contribute 0.
__global__ __launch_bounds__(256) void k_crop_bwd(const float* __restrict__ dout, int H, int W, int C, const float* __restrict__ rois,
                                                  float stride, int pool, float* __restrict__ dfeat) {
  const int r = blockIdx.x / pool, py = blockIdx.x % pool;
  const float height = ((float)H - 1.0f) * stride, width = ((float)W - 1.0f) * stride;
  const float* roi = rois + 5 * (size_t)r;
  const float x1 = roi[1] / width, y1 = roi[2] / height, x2 = roi[3] / width, y2 = roi[4] / height;
  const float hs = (y2 - y1) * (float)(H - 1) / (float)(pool - 1);
  const float ws = (x2 - x1) * (float)(W - 1) / (float)(pool - 1);
  const float in_y = y1 * (float)(H - 1) + (float)py * hs;
  if (in_y < 0 || in_y > (float)(H - 1)) return;
  const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
  const float ly = in_y - (float)top;
  const float* drow = dout + ((size_t)r * pool + py) * pool * C;
  for (int t = threadIdx.x; t < pool * C; t += 256) {
    const int px = t / C, c = t % C;
    const float in_x = x1 * (float)(W - 1) + (float)px * ws;
    if (in_x < 0 || in_x > (float)(W - 1)) continue;
    const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
    const float lx = in_x - (float)left;
    const float g = drow[(size_t)px * C + c];
    atomicAdd(dfeat + ((size_t)top * W + left) * C + c, g * (1.f - ly) * (1.f - lx));
    atomicAdd(dfeat + ((size_t)top * W + right) * C + c, g * (1.f - ly) * lx);
    atomicAdd(dfeat + ((size_t)bot * W + left) * C + c, g * ly * (1.f - lx));
    atomicAdd(dfeat + ((size_t)bot * W + right) * C + c, g * ly * lx);
  }
}
extern "C" int frcnn_crop_and_resize_bwd(const float* dout_d, int H, int W, int C, const float* rois_d, int R, float feat_stride,
                                         int pool, float* dfeat_d, void* stream) {
  if (!dout_d || !rois_d || !dfeat_d || H < 2 || W < 2 || C <= 0 || R < 0 || pool < 2) return FRCNN_E_ARG;
  if (R == 0) return FRCNN_OK;
  hipLaunchKernelGGL(k_crop_bwd, dim3(R * pool), dim3(256), 0, (hipStream_t)stream, dout_d, H, W, C, rois_d, feat_stride, pool, dfeat_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- momentum SGD (lib/model/train_val.py:128-145: tf.train.MomentumOptimizer(lr, 0.9)) on the packed
//      master filter [Cout][K]: g = grad_scale * dWf * s[n] + wd * W ; acc = mom*acc + g ; W -= lr*acc ;
//      Wf = W * s[n]   (s = folded frozen-BN scale or NULL; dWf is the gradient w.r.t. the folded filter)
__global__ void k_sgd(float* __restrict__ w, float* __restrict__ acc, float* __restrict__ wf, const float* __restrict__ dwf,
                      const float* __restrict__ scale, long long n, int K, float lr, float mom, float wd, float grad_scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = scale ? scale[i / K] : 1.f;
  const float g = grad_scale * dwf[i] * s + wd * w[i];
  const float a = mom * acc[i] + g;
  const float nw = w[i] - lr * a;
  acc[i] = a; w[i] = nw;
  if (wf) wf[i] = nw * s;
}
extern "C" int frcnn_sgd_momentum(float* w_d, float* acc_d, float* w_folded_d, const float* grad_d, const float* scale_d,
                                  long long n, int K, float lr, float momentum, float weight_decay, float grad_scale, void* stream) {
  if (!w_d || !acc_d || !grad_d || n <= 0 || K <= 0) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_sgd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_d, acc_d, w_folded_d, grad_d, scale_d,
                     n, K, lr, momentum, weight_decay, grad_scale);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// The same update for EVERY parameter tensor in one launch (the per-tensor form is ~150 launches per ResNet-152 step): a device
// table of descriptors, grid = (SGD_BLOCKS, count), block (b, t) walks tensor t with a grid stride.  lr and the mean-over-replicas
// factor are launch arguments (they change per step); per-tensor lr multiplier (DOUBLE_BIAS) and weight decay sit in the table.
struct SgdDesc { float* w; float* acc; float* wf; const float* grad; const float* scale; long long n; int K; float lr_mult; float wd; int pad; };
#define SGD_BLOCKS 32
__global__ __launch_bounds__(256) void k_sgd_multi(const SgdDesc* __restrict__ table, float lr, float mom, float grad_scale) {
  const SgdDesc d = table[blockIdx.y];
  const float lr_t = lr * d.lr_mult;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long long)SGD_BLOCKS * 256) {
    const float s = d.scale ? d.scale[i / d.K] : 1.f;
    const float g = grad_scale * d.grad[i] * s + d.wd * d.w[i];
    const float a = mom * d.acc[i] + g;
    const float nw = d.w[i] - lr_t * a;
    d.acc[i] = a; d.w[i] = nw;
    if (d.wf) d.wf[i] = nw * s;
  }
}
extern "C" size_t frcnn_sgd_desc_bytes(void) { return sizeof(SgdDesc); }
extern "C" int frcnn_sgd_momentum_multi(const void* desc_table_d, int count, float lr, float momentum, float grad_scale, void* stream) {
  if (!desc_table_d || count <= 0 || count > 65535) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_sgd_multi, dim3(SGD_BLOCKS, count), dim3(256), 0, (hipStream_t)stream, (const SgdDesc*)desc_table_d, lr, momentum,
                     grad_scale);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// sum of squares (L2 regulariser value, slim l2_regularizer = wd * sum(w^2) / 2): deterministic two-stage
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ w, long long n, double* __restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)w[i] * (double)w[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void k_sumsq_finish(const double* __restrict__ partial, int nb, double scale, float* __restrict__ out, int accumulate) {
  double s = 0.0;
  for (int i = 0; i < nb; ++i) s += partial[i];
  *out = (accumulate ? *out : 0.f) + (float)(s * scale);
}
// All regularised tensors in two launches (the per-tensor form costs two launches per filter: 320 per ResNet-152 step).
// grid = (SSQ_BLOCKS, count): block (b, t) reduces a strided part of tensor t; one block then adds the partials in index order.
#define SSQ_BLOCKS 64
__global__ __launch_bounds__(256) void k_sumsq_multi(const float* const* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                     double* __restrict__ partial) {
  __shared__ double sh[4];
  const float* w = ptrs[blockIdx.y];
  const long long n = sizes[blockIdx.y];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)SSQ_BLOCKS * 256) s += (double)w[i] * (double)w[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * SSQ_BLOCKS + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void k_sumsq_multi_finish(const double* __restrict__ partial, int np, double scale,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (accumulate ? *out : 0.f) + (float)((sh[0] + sh[1] + sh[2] + sh[3]) * scale);
}
extern "C" int frcnn_sumsq_multi(const void* ptr_table_d, const long long* sizes_d, int count, double scale, float* out_d,
                                 int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (!ptr_table_d || !sizes_d || !out_d || !ws || count <= 0 || count > 65535) return FRCNN_E_ARG;
  if (ws_bytes < sizeof(double) * (size_t)count * SSQ_BLOCKS) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_sumsq_multi, dim3(SSQ_BLOCKS, count), dim3(256), 0, (hipStream_t)stream, (const float* const*)ptr_table_d, sizes_d,
                     (double*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sumsq_multi_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws, count * SSQ_BLOCKS, scale, out_d,
                     accumulate);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_sumsq(const float* w_d, long long n, double scale, float* out_d, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (!w_d || !out_d || !ws || n <= 0) return FRCNN_E_ARG;
  const int nb = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  if (ws_bytes < sizeof(double) * 256) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(256), 0, (hipStream_t)stream, w_d, n, (double*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sumsq_finish, dim3(1), dim3(1), 0, (hipStream_t)stream, (const double*)ws, nb, scale, out_d, accumulate);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
