// Winograd for 7x7 feature maps (the per-RoI crops of block4: lib/nets/resnet_v1.py:115-125 on POOLING_SIZE 7): the seven
// outputs of a row are produced as F(4,3) (outputs 0..3 from padded inputs 0..5) + F(3,3) (outputs 4..6 from padded inputs
// 4..8), i.e. 6 + 5 = 11 transform points per dimension and 11 x 11 = 121 products per channel pair and RoI -- F(4x4,3x3)
// tiles need 2 x 2 x 36 = 144 (they cover 8x8), the direct convolution 441.  Every (xi, nu) point has exactly one row per RoI,
// so the products are 121 independent [R x Cin] x [Cin x Cout] GEMMs in one launch of frcnn_gemm_batched_nt.
//
// Matrices: Toom-Cook with points {0, 1, -1, 2, -2, inf} (F(4,3), Lavin & Gray) and {0, 1, -1, 2, inf} (F(3,3)), generated in
// exact rationals and checked by scratch/wino_matrices.py; block-structured 1-D forms B^T (11 x 9 padded inputs),
// A^T (7 x 11), G (11 x 3).  The unrolled loops skip zero coefficients at compile time (the tables are constexpr), so the
// transforms are add/multiply chains like the hand-written F(4,3) ones; both are HBM-bound.
#include "h2_common.h"

namespace w7 {
constexpr float BT[11][9] = {
    {4.f, 0.f, -5.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f},  {0.f, -4.f, -4.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 4.f, -4.f, -1.f, 1.f, 0.f, 0.f, 0.f, 0.f},
    {0.f, -2.f, -1.f, 2.f, 1.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 2.f, -1.f, -2.f, 1.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 4.f, 0.f, -5.f, 0.f, 1.f, 0.f, 0.f, 0.f},
    {0.f, 0.f, 0.f, 0.f, 2.f, -1.f, -2.f, 1.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, -2.f, -1.f, 1.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 2.f, -3.f, 1.f, 0.f},
    {0.f, 0.f, 0.f, 0.f, 0.f, -1.f, 0.f, 1.f, 0.f},  {0.f, 0.f, 0.f, 0.f, 0.f, 2.f, -1.f, -2.f, 1.f}};
constexpr float AT[7][11] = {
    {1.f, 1.f, 1.f, 1.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},  {0.f, 1.f, -1.f, 2.f, -2.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.f, 1.f, 1.f, 4.f, 4.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},  {0.f, 1.f, -1.f, 8.f, -8.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, 1.f, 0.f},  {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, -1.f, 2.f, 0.f},
    {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 4.f, 1.f}};
constexpr double G[11][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6},
                             {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1},                {1.0 / 2, 0, 0},               {-1.0 / 2, -1.0 / 2, -1.0 / 2},
                             {-1.0 / 6, 1.0 / 6, -1.0 / 6},  {1.0 / 6, 1.0 / 3, 2.0 / 3}, {0, 0, 1}};
}  // namespace w7

__device__ __forceinline__ float2 f2mad(float a, float2 x, float2 y) { return make_float2(a * x.x + y.x, a * x.y + y.y); }

// acc += coef * x with coef known at compile time after unrolling: zero terms vanish, +-1 become add / subtract
__device__ __forceinline__ void acc_term(float2& acc, bool& first, float coef, const float2 x) {
  if (coef == 0.f) return;
  if (first) {
    acc = coef == 1.f ? x : (coef == -1.f ? make_float2(-x.x, -x.y) : make_float2(coef * x.x, coef * x.y));
    first = false;
  } else if (coef == 1.f) {
    acc = make_float2(acc.x + x.x, acc.y + x.y);
  } else if (coef == -1.f) {
    acc = make_float2(acc.x - x.x, acc.y - x.y);
  } else {
    acc = f2mad(coef, x, acc);
  }
}

// Where a transform's result goes (float2 words: a thread owns two channels).  H2 = false: float32 tensor `f`.  H2 = true: operand
// planes of frcnn_gemm_h2 -- fp16 pieces [2][rows][C] + inv [C/128][rows]; the NR rows a thread writes together share the scale of
// their common maximum (h2_emit_rows64); the 64 consecutive threads that hold the rows' 128 channels are one wave (C2 % 64 == 0) and
// reduce it with DPP / permlane-swap moves, once per group.  `f` may be given as well.
template <bool H2>
struct Wino7Sink {
  float2* f; unsigned short* planes; float* inv; size_t rows;
  template <int NR>
  __device__ __forceinline__ void putn(const size_t* row, const float2* v, int c2, int C2) const {
    if (!H2 || f) {
#pragma unroll
      for (int n = 0; n < NR; ++n) f[row[n] * C2 + c2] = v[n];
    }
    if (H2) {
      size_t e[NR];
      float* slot[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        e[n] = (row[n] * C2 + c2) * 2;
        slot[n] = inv + (size_t)(c2 >> 6) * rows + row[n];
      }
      h2_emit_rows64<NR>(v, e, slot, (1u << NR) - 1u, planes, rows * (size_t)C2 * 2, c2 & 63);
    }
  }
};

// V[(xi*11+nu)][r][c] = (B^T d B)[xi][nu], d = the 7x7 map of RoI r, channel c, zero padded by one pixel
template <bool H2>
__global__ void __launch_bounds__(256) k_wino7_input(const float2* __restrict__ x, int R, int C2, const Wino7Sink<H2> V) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long long)R * C2) return;
  const int c2 = (int)(id % C2);
  const int r = (int)(id / C2);
  float2 d[7][7];
  const float2* src = x + (size_t)r * 49 * C2 + c2;
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) d[i][j] = src[(size_t)(i * 7 + j) * C2];
#pragma unroll
  for (int xi = 0; xi < 11; ++xi) {
    float2 t[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      bool first = true;
      t[j] = make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 7; ++i) acc_term(t[j], first, w7::BT[xi][i + 1], d[i][j]);      // padded row index = i + 1
    }
    float2 o[11];
    size_t rw[11];
#pragma unroll
    for (int nu = 0; nu < 11; ++nu) {
      bool first = true;
      float2 v = make_float2(0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 7; ++j) acc_term(v, first, w7::BT[nu][j + 1], t[j]);
      o[nu] = v;
      rw[nu] = (size_t)(xi * 11 + nu) * R + r;
    }
    V.template putn<11>(rw, o, c2, C2);
  }
}

// y[r][i][j][o] = act( (A^T M A)[i][j] + bias[o] ), M[(xi*11+nu)][r][o]; the two row segments (xi 0..5 -> rows 0..3, xi 6..10 ->
// rows 4..6) are processed one after the other to bound the register footprint
template <int XI0, int NXI, int I0, int NI, bool H2, bool MASK>
__device__ __forceinline__ void wino7_out_segment(const float2* __restrict__ in, size_t plane, float2 bv, int act, const Wino7Sink<H2>& y,
                                                  size_t row0, int c2, int C2, const float2* __restrict__ mask) {
  float2 m[NXI][11];
#pragma unroll
  for (int a = 0; a < NXI; ++a)
#pragma unroll
    for (int nu = 0; nu < 11; ++nu) m[a][nu] = in[(size_t)((XI0 + a) * 11 + nu) * plane];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float2 s[11];
#pragma unroll
    for (int nu = 0; nu < 11; ++nu) {
      bool first = true;
      s[nu] = make_float2(0.f, 0.f);
#pragma unroll
      for (int a = 0; a < NXI; ++a) acc_term(s[nu], first, w7::AT[I0 + i][XI0 + a], m[a][nu]);
    }
    float2 o[7];
    size_t rw[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      bool first = true;
      float2 v = make_float2(0.f, 0.f);
#pragma unroll
      for (int nu = 0; nu < 11; ++nu) acc_term(v, first, w7::AT[j][nu], s[nu]);
      v = make_float2(v.x + bv.x, v.y + bv.y);
      if (act == FRCNN_ACT_RELU) v = act_relu(v);
      rw[j] = row0 + (I0 + i) * 7 + j;
      if (MASK) {
        const float2 k = mask[rw[j] * C2 + c2];
        v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f;
      }
      o[j] = v;
    }
    y.template putn<7>(rw, o, c2, C2);
  }
}

// MASK (training, frcnn_winograd7_output_transform_masked): y = mask > 0 ? y : 0
template <bool H2, bool MASK = false>
__global__ void __launch_bounds__(256) k_wino7_output(const float2* __restrict__ Mx, int R, int C2, const float2* __restrict__ bias, int act,
                                                       const float2* __restrict__ mask, const Wino7Sink<H2> y) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long long)R * C2) return;
  const int c2 = (int)(id % C2);
  const int r = (int)(id / C2);
  const size_t plane = (size_t)R * C2;
  const float2* in = Mx + (size_t)r * C2 + c2;
  const float2 bv = bias ? bias[c2] : make_float2(0.f, 0.f);
  wino7_out_segment<0, 6, 0, 4, H2, MASK>(in, plane, bv, act, y, (size_t)r * 49, c2, C2, mask);
  wino7_out_segment<6, 5, 4, 3, H2, MASK>(in, plane, bv, act, y, (size_t)r * 49, c2, C2, mask);
}

// U[(xi*11+nu)][o][c] = (G g G^T)[xi][nu] on the device from the packed filter [Cout][3][3][Cin] (training); transpose_flip as in
// frcnn_winograd_filter_transform_device
__global__ void k_wino7_filter(const float* __restrict__ w, int O, int C, int transpose_flip, float* __restrict__ U) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long long)O * C) return;
  int o, c;
  if (transpose_flip) { o = (int)(id % O); c = (int)(id / O); }
  else { c = (int)(id % C); o = (int)(id / C); }
  float g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kh = transpose_flip ? 2 - i : i, kw = transpose_flip ? 2 - j : j;
      g[i][j] = w[(((size_t)o * 3 + kh) * 3 + kw) * C + c];
    }
  float t[11][3];
#pragma unroll
  for (int a = 0; a < 11; ++a)
#pragma unroll
    for (int j = 0; j < 3; ++j) t[a][j] = (float)w7::G[a][0] * g[0][j] + (float)w7::G[a][1] * g[1][j] + (float)w7::G[a][2] * g[2][j];
  const size_t plane = (size_t)O * C;
  float* out = U + (transpose_flip ? (size_t)c * O + o : (size_t)o * C + c);
#pragma unroll
  for (int a = 0; a < 11; ++a)
#pragma unroll
    for (int b = 0; b < 11; ++b)
      out[(size_t)(a * 11 + b) * plane] = t[a][0] * (float)w7::G[b][0] + t[a][1] * (float)w7::G[b][1] + t[a][2] * (float)w7::G[b][2];
}

// HOST: HWIO [3][3][Cin][Cout] (optionally * scale[o]) -> U [121][Cout][Cin], float64 arithmetic
extern "C" int frcnn_winograd7_filter_transform(const float* w_hwio, int Cin, int Cout, const float* scale, float* u_out) {
  if (!w_hwio || !u_out || Cin <= 0 || Cout <= 0) return FRCNN_E_ARG;
  for (int c = 0; c < Cin; ++c)
    for (int o = 0; o < Cout; ++o) {
      double g[3][3], t[11][3];
      const double sc = scale ? (double)scale[o] : 1.0;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g[i][j] = (double)w_hwio[((size_t)(i * 3 + j) * Cin + c) * Cout + o] * sc;
      for (int a = 0; a < 11; ++a)
        for (int j = 0; j < 3; ++j) t[a][j] = w7::G[a][0] * g[0][j] + w7::G[a][1] * g[1][j] + w7::G[a][2] * g[2][j];
      for (int a = 0; a < 11; ++a)
        for (int b = 0; b < 11; ++b)
          u_out[((size_t)(a * 11 + b) * Cout + o) * Cin + c] = (float)(t[a][0] * w7::G[b][0] + t[a][1] * w7::G[b][1] + t[a][2] * w7::G[b][2]);
    }
  return FRCNN_OK;
}

extern "C" int frcnn_winograd7_filter_transform_device(const float* w_packed_d, int Cout, int Cin, int transpose_flip, float* u_d,
                                                       void* stream) {
  if (!w_packed_d || !u_d || Cout <= 0 || Cin <= 0) return FRCNN_E_ARG;
  const long long tot = (long long)Cout * Cin;
  hipLaunchKernelGGL(k_wino7_filter, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_packed_d, Cout, Cin,
                     transpose_flip ? 1 : 0, u_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// x_d [R][7][7][C] -> v_d [121][R][C]
extern "C" int frcnn_winograd7_input_transform(const float* x_d, int R, int C, float* v_d, void* stream) {
  if (!x_d || !v_d || R <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % 2) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)R * (C / 2);
  const Wino7Sink<false> sink{(float2*)v_d, nullptr, nullptr, 0};
  hipLaunchKernelGGL(k_wino7_input<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)x_d, R, C / 2, sink);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ... -> operand planes [2][121 * R][C] + v_inv [C/128][121 * R] of frcnn_gemm_h2; C % 128 == 0
extern "C" int frcnn_winograd7_input_transform_h2(const float* x_d, int R, int C, void* v_planes_d, float* v_inv_d, void* stream) {
  if (!x_d || !v_planes_d || !v_inv_d || R <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % H2_KB) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)R * (C / 2);
  const Wino7Sink<true> sink{nullptr, (unsigned short*)v_planes_d, v_inv_d, (size_t)121 * R};
  hipLaunchKernelGGL(k_wino7_input<true>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)x_d, R, C / 2, sink);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// m_d [121][R][C] (+ bias, ReLU) -> y_d [R][7][7][C]
extern "C" int frcnn_winograd7_output_transform(const float* m_d, int R, int C, const float* bias_d, int act, float* y_d, void* stream) {
  if (!m_d || !y_d || R <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % 2 || (act != FRCNN_ACT_NONE && act != FRCNN_ACT_RELU)) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)R * (C / 2);
  const Wino7Sink<false> sink{(float2*)y_d, nullptr, nullptr, 0};
  hipLaunchKernelGGL((k_wino7_output<false, false>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)m_d, R,
                     C / 2, (const float2*)bias_d, act, (const float2*)nullptr, sink);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ... -> operand planes [2][R * 49][C] + y_inv [C/128][R * 49] (y_d: NULL or the float32 result as well); C % 128 == 0
extern "C" int frcnn_winograd7_output_transform_h2(const float* m_d, int R, int C, const float* bias_d, int act, float* y_d, void* y_planes_d,
                                                   float* y_inv_d, void* stream) {
  if (!m_d || !y_planes_d || !y_inv_d || R <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % H2_KB || (act != FRCNN_ACT_NONE && act != FRCNN_ACT_RELU)) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)R * (C / 2);
  const Wino7Sink<true> sink{(float2*)y_d, (unsigned short*)y_planes_d, y_inv_d, (size_t)R * 49};
  hipLaunchKernelGGL((k_wino7_output<true, false>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)m_d, R,
                     C / 2, (const float2*)bias_d, act, (const float2*)nullptr, sink);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// Training: the output transform followed by the ReLU gradient of the tensor the result is the gradient OF (see
// frcnn_winograd_output_transform_masked): y = mask > 0 ? A^T M A : 0, mask [R][7][7][C] float32; float32 (y_d) and / or operand planes.
extern "C" int frcnn_winograd7_output_transform_masked(const float* m_d, int R, int C, const float* mask_d, float* y_d, void* y_planes_d,
                                                       float* y_inv_d, void* stream) {
  if (!m_d || !mask_d || (!y_d && !y_planes_d) || (y_planes_d && !y_inv_d) || R <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % 2 || (y_planes_d && C % H2_KB)) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)R * (C / 2);
  if (y_planes_d) {
    const Wino7Sink<true> sink{(float2*)y_d, (unsigned short*)y_planes_d, y_inv_d, (size_t)R * 49};
    hipLaunchKernelGGL((k_wino7_output<true, true>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)m_d, R,
                       C / 2, (const float2*)nullptr, FRCNN_ACT_NONE, (const float2*)mask_d, sink);
  } else {
    const Wino7Sink<false> sink{(float2*)y_d, nullptr, nullptr, 0};
    hipLaunchKernelGGL((k_wino7_output<false, true>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)m_d, R,
                       C / 2, (const float2*)nullptr, FRCNN_ACT_NONE, (const float2*)mask_d, sink);
  }
  LAUNCH_CHECK();
  return FRCNN_OK;
}
