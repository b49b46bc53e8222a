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

// bias gradient: db[c] = sum_m dy[m][c]  (f64 accumulation, deterministic: a fixed partition of the rows and a fixed order of the adds).
// 1024 threads = 16 row parts x 64 channels, every part walks its rows four at a time with four independent accumulators (round 3: 4
// parts, one dependent load + add per iteration -- 226 us for the 2 394 rows of an RPN head, profiles/r03_aj_train_kernels_by_shape.txt).
__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ dy, int M, int C, float* __restrict__ db) {
  __shared__ double sh[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (c < C) {
    int m = part;
    for (; m + 48 < M; m += 64) {
      s0 += (double)dy[(size_t)m * C + c];
      s1 += (double)dy[(size_t)(m + 16) * C + c];
      s2 += (double)dy[(size_t)(m + 32) * C + c];
      s3 += (double)dy[(size_t)(m + 48) * C + c];
    }
    for (; m < M; m += 16) s0 += (double)dy[(size_t)m * C + c];
  }
  sh[part][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][threadIdx.x];
    db[c] = (float)t;
  }
}
extern "C" int frcnn_colsum(const float* dy_d, int M, int C, float* db_d, void* stream) {
  if (!dy_d || !db_d || M <= 0 || C <= 0) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_colsum, dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, dy_d, M, C, db_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- crop_and_resize backward w.r.t. the feature map (TF CropAndResizeGradImage semantics): every output sample hands its
//      gradient to its four bilinear taps; out-of-range samples contribute 0.  GATHER form, deterministic: TensorFlow's kernel (and
//      rounds 1-4 here) scatters with float atomics, so the sum a feature pixel receives depended on the order the hardware happened to
//      retire ~100 overlapping RoIs in, and two runs of the same training step differed in the last bits of every gradient upstream of
//      the crop.  Here one workgroup owns ONE feature row h and 256 channels: (1) all threads scan the R * pool sample rows and compact,
//      in ascending (roi, py) order, those whose top or bottom tap row is h (wave ballots + a running prefix: the list order is a
//      function of the inputs only); (2) thread = channel walks the list and adds the samples' taps into its own column of an LDS row
//      buffer [W][256] in (roi, py, px, tap) order; (3) the row buffer is added to dfeat.  No atomics, a fixed order per element, and
//      the 2 x 7 x 256 RoI rows that touch a feature row are read once, coalesced over the channels.
//      Sizes (round 6): NT = 256 / 128 / 64 channels per workgroup so that the row buffer [W][NT] fits the CU's LDS at any W up
//      to ~600 feature columns, and the hit list is a window of `lcap` entries that is walked and refilled in ascending order when R * pool
//      exceeds it -- the same (roi, py, px, tap) order per element in every shape of the launch, hence the same bits.
template <int NT>
__global__ __launch_bounds__(NT) void k_crop_bwd_rows(const float* __restrict__ dout, int H, int W, int C, const float* __restrict__ rois,
                                                      int R, float stride, int pool, int lcap, float* __restrict__ dfeat) {
  extern __shared__ float crop_lds[];
  float* acc = crop_lds;                                   // [W][NT]
  int* list = (int*)(crop_lds + (size_t)W * NT);           // [lcap] hits, ascending
  __shared__ int wcnt[NT / 64];
  __shared__ int total_s;
  const int h = blockIdx.x, tid = threadIdx.x, c = blockIdx.y * NT + tid;
  const int lane = tid & 63, wave = tid >> 6;
  const float height = ((float)H - 1.0f) * stride, width = ((float)W - 1.0f) * stride;
  for (int w = 0; w < W; ++w) acc[w * NT + tid] = 0.f;
  if (tid == 0) total_s = 0;
  __syncthreads();
  const int n = R * pool;
  const bool live = c < C;
  auto walk = [&](int hits) {                              // thread = channel: the listed sample rows into this thread's column of acc
    for (int k = 0; k < hits; ++k) {
      const int i = list[k];
      const int r = i / pool, py = i - r * pool;
      const float* roi = rois + 5 * (size_t)r;
      const float x1 = roi[1] / width, y1 = roi[2] / height, x2 = roi[3] / width, y2 = roi[4] / height;
      const float hs = (y2 - y1) * (float)(H - 1) / (float)(pool - 1);
      const float ws = (x2 - x1) * (float)(W - 1) / (float)(pool - 1);
      const float in_y = y1 * (float)(H - 1) + (float)py * hs;
      const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
      const float ly = in_y - (float)top;
      const float* drow = dout + ((size_t)r * pool + py) * pool * C;
      for (int px = 0; px < pool; ++px) {
        const float in_x = x1 * (float)(W - 1) + (float)px * ws;
        if (in_x < 0 || in_x > (float)(W - 1)) continue;
        const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
        const float lx = in_x - (float)left;
        const float g = live ? drow[(size_t)px * C + c] : 0.f;
        if (top == h) {
          acc[left * NT + tid] += g * (1.f - ly) * (1.f - lx);
          acc[right * NT + tid] += g * (1.f - ly) * lx;
        }
        if (bot == h) {
          acc[left * NT + tid] += g * ly * (1.f - lx);
          acc[right * NT + tid] += g * ly * lx;
        }
      }
    }
  };
  for (int base = 0; base < n; base += NT) {
    if (total_s + NT > lcap) {                             // (uniform) the window is full: walk it, then refill from here
      walk(total_s);
      __syncthreads();
      if (tid == 0) total_s = 0;
      __syncthreads();
    }
    const int i = base + tid;
    bool hit = false;
    if (i < n) {
      const int r = i / pool, py = i - r * pool;
      const float* roi = rois + 5 * (size_t)r;
      const float y1 = roi[2] / height, y2 = roi[4] / height;
      const float hs = (y2 - y1) * (float)(H - 1) / (float)(pool - 1);
      const float in_y = y1 * (float)(H - 1) + (float)py * hs;
      if (!(in_y < 0 || in_y > (float)(H - 1))) hit = ((int)floorf(in_y) == h) || ((int)ceilf(in_y) == h);
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = total_s, all = 0;
    for (int q = 0; q < NT / 64; ++q) {
      if (q < wave) off += wcnt[q];
      all += wcnt[q];
    }
    if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) total_s += all;
    __syncthreads();
  }
  walk(total_s);
  if (live) {
    float* drow = dfeat + (size_t)h * W * C + c;
    for (int w = 0; w < W; ++w) drow[(size_t)w * C] += acc[w * NT + tid];
  }
}
template <int NT>
static int launch_crop_bwd(const float* dout_d, int H, int W, int C, const float* rois_d, int R, float feat_stride, int pool, int lcap,
                           float* dfeat_d, hipStream_t st) {
  static KernelOnce once;
  HIP_TRY(kernel_once(once, (const void*)k_crop_bwd_rows<NT>, NT, 160 * 1024 - 64));
  hipLaunchKernelGGL(k_crop_bwd_rows<NT>, dim3(H, cdiv(C, NT)), dim3(NT), (size_t)W * NT * 4 + (size_t)lcap * 4, st, dout_d, H, W, C, rois_d, R,
                     feat_stride, pool, lcap, dfeat_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
// max_lds_bytes > 0: the LDS budget of the launch plan (tests force the narrow / windowed forms on small inputs); 0 = the CU's 160 KB
extern "C" int frcnn_crop_and_resize_bwd_plan(const float* dout_d, int H, int W, int C, const float* rois_d, int R, float feat_stride,
                                              int pool, float* dfeat_d, size_t max_lds_bytes, void* stream) {
  if (!dout_d || !rois_d || !dfeat_d || H < 2 || W < 2 || C <= 0 || R < 0 || pool < 2) return FRCNN_E_ARG;
  if (R == 0) return FRCNN_OK;
  const size_t budget = (max_lds_bytes > 0 && max_lds_bytes < 160 * 1024 - 64) ? max_lds_bytes : 160 * 1024 - 64;
  const long long n = (long long)R * pool;
  for (int nt = 256; nt >= 64; nt >>= 1) {
    // widest channel slab whose row buffer leaves room for a hit window of at least 4 x NT entries (or the whole list)
    const size_t accb = (size_t)W * nt * 4;
    const long long want = (n + nt - 1) / nt * nt;
    if (accb + (size_t)min(want, (long long)4 * nt) * 4 > budget) continue;
    const int lcap = (int)min(want, (long long)((budget - accb) / 4 / nt * nt));
    hipStream_t st = (hipStream_t)stream;
    if (nt == 256) return launch_crop_bwd<256>(dout_d, H, W, C, rois_d, R, feat_stride, pool, lcap, dfeat_d, st);
    if (nt == 128) return launch_crop_bwd<128>(dout_d, H, W, C, rois_d, R, feat_stride, pool, lcap, dfeat_d, st);
    return launch_crop_bwd<64>(dout_d, H, W, C, rois_d, R, feat_stride, pool, lcap, dfeat_d, st);
  }
  return FRCNN_E_UNSUPPORTED;             // W > ~600 feature columns (an image side of ~10 000 pixels at stride 16)
}
extern "C" int frcnn_crop_and_resize_bwd(const float* dout_d, int H, int W, int C, const float* rois_d, int R, float feat_stride,
                                         int pool, float* dfeat_d, void* stream) {
  return frcnn_crop_and_resize_bwd_plan(dout_d, H, W, C, rois_d, R, feat_stride, pool, dfeat_d, 0, stream);
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
// (128 blocks per tensor: with 32 the launch took as long as the 32 workgroups of the largest filter -- 2.4 M elements, 290 dependent
// iterations each -- needed, 440 of the 487 us; blocks past a small tensor's end leave at once)
#define SGD_BLOCKS 128
__global__ __launch_bounds__(256) void k_sgd_multi(const SgdDesc* __restrict__ table, float lr, float mom, float grad_scale) {
  const SgdDesc d = table[blockIdx.y];
  if ((long long)blockIdx.x * 256 >= d.n) return;
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
// `first`: the launch covers table[first .. first + count) -- the solver updates the tensors of a finished part of the reverse sweep while
// the sweep goes on (frcnn_hip/train.py)
extern "C" int frcnn_sgd_momentum_range(const void* desc_table_d, int first, int count, float lr, float momentum, float grad_scale, void* stream) {
  if (!desc_table_d || first < 0 || count <= 0 || count > 65535) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_sgd_multi, dim3(SGD_BLOCKS, count), dim3(256), 0, (hipStream_t)stream, (const SgdDesc*)desc_table_d + first, lr, momentum,
                     grad_scale);
  LAUNCH_CHECK();
  return FRCNN_OK;
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

// =====================================================================================================================
// Reverse-sweep pieces of the VGG16 and MobileNet-v1 TRAIN graphs (lib/nets/vgg16.py:26-60, mobilenet_v1.py:114-172).
// All bandwidth-bound, float4 over channels.
// =====================================================================================================================

// ---- tf.nn.relu6 gradient: grad *= (0 < y < 6) --------------------------------------------------------------------
__global__ void k_relu6_bwd(float4* __restrict__ g, const float4* __restrict__ y, long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = g[i];
  const float4 b = y[i];
  a.x = (b.x > 0.f && b.x < 6.f) ? a.x : 0.f; a.y = (b.y > 0.f && b.y < 6.f) ? a.y : 0.f;
  a.z = (b.z > 0.f && b.z < 6.f) ? a.z : 0.f; a.w = (b.w > 0.f && b.w < 6.f) ? a.w : 0.f;
  g[i] = a;
}
extern "C" int frcnn_relu6_bwd(float* grad_d, const float* y_d, long long n, void* stream) {
  if (!grad_d || !y_d || n <= 0 || (n & 3)) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_relu6_bwd, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)grad_d, (const float4*)y_d, n / 4);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- max pool gradient (slim.max_pool2d 'SAME' with bottom/right padding only, vgg16.py:29-41; network.py:156) ---------------
// Gather form, one thread per input element quad: for every window that covers the element, the window's gradient goes to
// its FIRST maximum in (row, column) scan order (the arg-max the forward max would report); windows may overlap (k > stride).
__global__ void k_maxpool_bwd(const float4* __restrict__ x, int N, int H, int W, int C4, int k, int stride, const float4* __restrict__ y,
                              const float4* __restrict__ dy, int OH, int OW, float4* __restrict__ dx) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)N * H * W * C4;
  if (t >= tot) return;
  const int c4 = (int)(t % C4);
  long long pix = t / C4;
  const int iw = (int)(pix % W); pix /= W;
  const int ih = (int)(pix % H);
  const int img = (int)(pix / H);
  const float4 xv = x[t];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int oh_lo = max(0, (ih - k + stride) / stride), oh_hi = min(OH - 1, ih / stride);     // ceil((ih-k+1)/stride) for ih-k+1 > 0
  const int ow_lo = max(0, (iw - k + stride) / stride), ow_hi = min(OW - 1, iw / stride);
  for (int oh = oh_lo; oh <= oh_hi; ++oh)
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const size_t o = ((size_t)(img * OH + oh) * OW + ow) * C4 + c4;
      const float4 m = y[o], g = dy[o];
      bool mx = xv.x == m.x, my = xv.y == m.y, mz = xv.z == m.z, mw = xv.w == m.w;
      // an earlier element of the same window with the same value takes the gradient instead
      const int ry = ih - oh * stride, rx = iw - ow * stride;
      for (int a = 0; a <= ry; ++a)
        for (int b = 0; b < (a == ry ? rx : k); ++b) {
          const int jh = oh * stride + a, jw = ow * stride + b;
          if (jh >= H || jw >= W) continue;
          const float4 e = x[((size_t)(img * H + jh) * W + jw) * C4 + c4];
          mx = mx && !(e.x == m.x); my = my && !(e.y == m.y); mz = mz && !(e.z == m.z); mw = mw && !(e.w == m.w);
        }
      acc.x += mx ? g.x : 0.f; acc.y += my ? g.y : 0.f; acc.z += mz ? g.z : 0.f; acc.w += mw ? g.w : 0.f;
    }
  dx[t] = acc;
}
extern "C" int frcnn_maxpool_bwd(const float* x_d, int N, int H, int W, int C, int k, int stride, const float* y_d, const float* dy_d,
                                 int OH, int OW, float* dx_d, void* stream) {
  if (!x_d || !y_d || !dy_d || !dx_d || N <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || stride <= 0 || OH <= 0 || OW <= 0) return FRCNN_E_ARG;
  if (C % 4 || stride > k) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x_d, N, H, W, C / 4,
                     k, stride, (const float4*)y_d, (const float4*)dy_d, OH, OW, (float4*)dx_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- dropout (slim.dropout(keep_prob = 0.5, is_training = True), vgg16.py:52-58 = tf.nn.dropout: x / keep_prob * floor(keep_prob + u)) ----
// u = a counter-based uniform of (seed, element index): the mask is recomputed in the backward pass, never stored.  TF's own
// random stream is not reproducible outside TF; tests feed the device mask to the float64 reference as a constant.
__device__ __forceinline__ float dropout_keep(unsigned long long seed, unsigned long long idx, float keep_prob) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);            // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);                         // [0, 1)
  return floorf(keep_prob + u);                                                    // 1 with probability keep_prob
}
__global__ void k_dropout(const float* __restrict__ x, long long n, unsigned long long seed, float keep_prob, float* __restrict__ y) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] / keep_prob * dropout_keep(seed, (unsigned long long)i, keep_prob);
}
// forward and backward are the same map (y = x * mask / keep_prob, dx = dy * mask / keep_prob); in-place allowed
extern "C" int frcnn_dropout(const float* x_d, long long n, unsigned long long seed, float keep_prob, float* y_d, void* stream) {
  if (!x_d || !y_d || n <= 0 || !(keep_prob > 0.f) || keep_prob > 1.f) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_dropout, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_d, n, seed, keep_prob, y_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- depthwise 3x3 gradients (mobilenet_v1.py:21-49, 146-160) -----------------------------------------------------------------
// data gradient: dx[n,ih,iw,c] = sum over taps (dy,dx) with oh*s - pt + dy == ih: g[n,oh,ow,c] * w[dy][dx][c]
__global__ void k_dwconv3x3_dgrad(const float4* __restrict__ g, int N, int OH, int OW, int C4, const float4* __restrict__ w,
                                  float4* __restrict__ dx, int H, int W, int stride, int pt, int pl, int accumulate) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = (long long)N * H * W * C4;
  if (t >= tot) return;
  const int c4 = (int)(t % C4);
  long long pix = t / C4;
  const int iw = (int)(pix % W); pix /= W;
  const int ih = (int)(pix % H);
  const int img = (int)(pix / H);
  float4 a = accumulate ? dx[t] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int dy = 0; dy < 3; ++dy) {
    const int th = ih + pt - dy;
    if (th < 0 || th % stride) continue;
    const int oh = th / stride;
    if (oh >= OH) continue;
    for (int dxk = 0; dxk < 3; ++dxk) {
      const int tw = iw + pl - dxk;
      if (tw < 0 || tw % stride) continue;
      const int ow = tw / stride;
      if (ow >= OW) continue;
      const float4 v = g[((size_t)(img * OH + oh) * OW + ow) * C4 + c4];
      const float4 f = w[(dy * 3 + dxk) * C4 + c4];
      a.x = fmaf(v.x, f.x, a.x); a.y = fmaf(v.y, f.y, a.y); a.z = fmaf(v.z, f.z, a.z); a.w = fmaf(v.w, f.w, a.w);
    }
  }
  dx[t] = a;
}
extern "C" int frcnn_dwconv3x3_dgrad(const float* g_d, int N, int OH, int OW, int C, const float* w_d, float* dx_d, int H, int W, int stride,
                                     int pad_top, int pad_left, int accumulate, void* stream) {
  if (!g_d || !w_d || !dx_d || N <= 0 || OH <= 0 || OW <= 0 || C <= 0 || H <= 0 || W <= 0 || stride <= 0) return FRCNN_E_ARG;
  if (C % 4) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(k_dwconv3x3_dgrad, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)g_d, N, OH, OW,
                     C / 4, (const float4*)w_d, (float4*)dx_d, H, W, stride, pad_top, pad_left, accumulate);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// filter gradient: dW[tap][c] = scale[c] * sum_{n,oh,ow} g[n,oh,ow,c] * x[n, oh*s-pt+dy, ow*s-pl+dx, c]   (scale: chain rule through
// the frozen-BN fold, may be null).  Two deterministic stages: workgroup (64-channel group, pixel chunk) -> partial[chunk][9][C],
// then an in-order sum over the chunks.  256 threads = 16 channel quads x 16 pixel lanes (a wave reads 256-byte runs).
#define DWG_CHUNK 4096
__global__ __launch_bounds__(256) void k_dwconv3x3_wgrad(const float4* __restrict__ g, const float4* __restrict__ x, int N, int H, int W,
                                                         int C4, int OH, int OW, int stride, int pt, int pl, float4* __restrict__ partial) {
  __shared__ float4 red[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c4 = blockIdx.x * 16 + tx;
  const long long M = (long long)N * OH * OW;
  const long long m0 = (long long)blockIdx.y * DWG_CHUNK, m1 = min(M, m0 + DWG_CHUNK);
  float4 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < C4)
    for (long long m = m0 + ty; m < m1; m += 16) {
      const int ow = (int)(m % OW), oh = (int)((m / OW) % OH), img = (int)(m / ((long long)OW * OH));
      const float4 gv = g[(size_t)m * C4 + c4];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ih = oh * stride - pt + dy;
#pragma unroll
        for (int dxk = 0; dxk < 3; ++dxk) {
          const int iw = ow * stride - pl + dxk;
          if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
            const float4 v = x[((size_t)(img * H + ih) * W + iw) * C4 + c4];
            float4& a = acc[dy * 3 + dxk];
            a.x = fmaf(gv.x, v.x, a.x); a.y = fmaf(gv.y, v.y, a.y); a.z = fmaf(gv.z, v.z, a.z); a.w = fmaf(gv.w, v.w, a.w);
          }
        }
      }
    }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    red[ty][tx] = acc[q];
    __syncthreads();
    if (ty == 0 && c4 < C4) {
      float4 s = red[0][tx];
      for (int r = 1; r < 16; ++r) { const float4 v = red[r][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
      partial[((size_t)blockIdx.y * 9 + q) * C4 + c4] = s;
    }
    __syncthreads();
  }
}
__global__ void k_dwconv3x3_wgrad_finish(const float4* __restrict__ partial, int chunks, int C4, const float4* __restrict__ scale,
                                         float4* __restrict__ dw) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 9 * C4) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < chunks; ++c) { const float4 v = partial[(size_t)c * 9 * C4 + t]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  if (scale) { const float4 k = scale[t % C4]; s.x *= k.x; s.y *= k.y; s.z *= k.z; s.w *= k.w; }
  dw[t] = s;
}
extern "C" size_t frcnn_dwconv3x3_wgrad_workspace_bytes(int N, int OH, int OW, int C) {
  if (N <= 0 || OH <= 0 || OW <= 0 || C <= 0) return 256;
  return (size_t)cdiv((long long)N * OH * OW, DWG_CHUNK) * 9 * (size_t)C * sizeof(float);
}
extern "C" int frcnn_dwconv3x3_wgrad(const float* g_d, const float* x_d, int N, int H, int W, int C, int OH, int OW, int stride, int pad_top,
                                     int pad_left, const float* scale_d, float* dw_d, void* ws, size_t ws_bytes, void* stream) {
  if (!g_d || !x_d || !dw_d || !ws || N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || stride <= 0) return FRCNN_E_ARG;
  if (C % 4) return FRCNN_E_UNSUPPORTED;
  if (frcnn_dwconv3x3_wgrad_workspace_bytes(N, OH, OW, C) > ws_bytes) return FRCNN_E_WS;
  const int chunks = (int)cdiv((long long)N * OH * OW, DWG_CHUNK), C4 = C / 4;
  hipLaunchKernelGGL(k_dwconv3x3_wgrad, dim3(cdiv(C4, 16), chunks), dim3(256), 0, (hipStream_t)stream, (const float4*)g_d, (const float4*)x_d,
                     N, H, W, C4, OH, OW, stride, pad_top, pad_left, (float4*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dwconv3x3_wgrad_finish, dim3(cdiv(9 * C4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)ws, chunks, C4,
                     (const float4*)scale_d, (float4*)dw_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// forward filter of a depthwise layer from its master copy: wf[tap][c] = w[tap][c] * scale[c] (after the solver step)
__global__ void k_dw_refold(const float* __restrict__ w, const float* __restrict__ scale, int C, float* __restrict__ wf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 9 * C) wf[t] = w[t] * scale[t % C];
}
extern "C" int frcnn_dwconv3x3_refold(const float* w_d, const float* scale_d, int C, float* wf_d, void* stream) {
  if (!w_d || !scale_d || !wf_d || C <= 0) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_dw_refold, dim3(cdiv(9 * C, 256)), dim3(256), 0, (hipStream_t)stream, w_d, scale_d, C, wf_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
