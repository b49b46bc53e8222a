// Winograd F(2x2, 3x3) for the stride-1, pad-1 3x3 convolutions of the path (block conv2's, RPN 3x3;
// lib/nets/resnet_v1.py bottleneck conv2, lib/nets/network.py:324): exact algebra, f32 throughout,
// 2.25x fewer multiplications (16 per 2x2 outputs instead of 36).
//
//   U = G g G^T (per filter, host, once)    V = B^T d B (4x4 input tile)    M_xn = sum_c U_xn[o][c] V_xn[t][c]
//   Y = A^T M A (2x2 outputs per tile), + bias, activation
// The 16 element-wise products are 16 independent [T x Cin] x [Cin x Cout] GEMMs -> frcnn_gemm_batched_nt
// (the same f32-MFMA kernel, grid.y = 16).  The two transforms below are bandwidth-bound float4 kernels.
#include "h2_common.h"

//
// m = 2: F(2x2,3x3), 4x4 input tiles, 16 GEMMs.   m = 4: F(4x4,3x3) (Lavin & Gray points 0,+-1,+-2,inf), 6x6 input tiles,
// 36 GEMMs, 4x fewer multiplications than direct; its f32 rounding error is ~10x the direct kernel's, so the network
// only uses it where few such layers follow one another (block4's three conv2 on the 7x7 crops).

// HOST: HWIO [3][3][Cin][Cout] (optionally * scale[o]) -> U [(m+2)^2][Cout][Cin]
extern "C" int frcnn_winograd_filter_transform(const float* w_hwio, int Cin, int Cout, const float* scale, int m, float* u_out) {
  if (!w_hwio || !u_out || Cin <= 0 || Cout <= 0) return FRCNN_E_ARG;
  if (m != 2 && m != 4) return FRCNN_E_UNSUPPORTED;
  static const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  static const double G4[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  const int a = m + 2;
  const double(*G)[3] = m == 2 ? G2 : G4;
  for (int c = 0; c < Cin; ++c)
    for (int o = 0; o < Cout; ++o) {
      double g[3][3], t[6][3];
      const double sc = scale ? (double)scale[o] : 1.0;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g[i][j] = (double)w_hwio[((size_t)(i * 3 + j) * Cin + c) * Cout + o] * sc;
      for (int i = 0; i < a; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
      for (int i = 0; i < a; ++i)
        for (int j = 0; j < a; ++j) {
          const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
          u_out[((size_t)(i * a + j) * Cout + o) * Cin + c] = (float)u;
        }
    }
  return FRCNN_OK;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Where a transform's result goes.  H2 = false: float32 tensor `f` (rows x C, float4 words).  H2 = true: operand planes of
// frcnn_gemm_h2 (csrc/gemm_h2.hip) -- fp16 pieces [2][rows][C] + one power-of-two scale per (row, 128 channels), inv [C/128][rows].
// A thread writes the NR rows of one transform row together and they SHARE the scale of their common maximum (h2_emit_rows32): the
// 32 consecutive threads that hold the rows' 128 channels (C4 % 32 == 0, so they are one half-wave) reduce it with DPP /
// permlane-swap moves, once per group.  `f` may be given as well (both are written).
template <bool H2>
struct WinoSink {
  float4* f; unsigned short* planes; float* inv; size_t rows;      // rows: total rows of the result (plane stride = rows * C)
  template <int NR>
  __device__ __forceinline__ void putn(const size_t* row, const float4* v, unsigned valid, int c4, int C4) const {
    if (!H2 || f) {
#pragma unroll
      for (int n = 0; n < NR; ++n)
        if ((valid >> n) & 1u) f[row[n] * C4 + c4] = v[n];
    }
    if (H2) {
      size_t e[NR];
      float* slot[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        e[n] = (row[n] * C4 + c4) * 4;
        slot[n] = inv + (size_t)(c4 >> 5) * rows + row[n];
      }
      h2_emit_rows32<NR>(v, e, slot, valid, planes, rows * (size_t)C4 * 4, c4 & 31);
    }
  }
};

// V[xn][t][c] = (B^T d B)[xi][nu], tile t = (img, ty, tx), d[i][j] = x[img, 2ty-1+i, 2tx-1+j, c]
template <bool H2>
__global__ void k_wino_input(const float4* __restrict__ x, int N, int H, int W, int C4, int TH, int TW, const WinoSink<H2> V) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long T = (long long)N * TH * TW;
  if (id >= T * C4) return;
  const int c4 = (int)(id % C4);
  const long long t = id / C4;
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), img = (int)(t / ((long long)TW * TH));
  float4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ih = 2 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iw = 2 * tx - 1 + j;
      d[i][j] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) ? x[((size_t)(img * H + ih) * W + iw) * C4 + c4]
                                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 b[4][4];       // B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    b[0][j] = f4sub(d[0][j], d[2][j]);
    b[1][j] = f4add(d[1][j], d[2][j]);
    b[2][j] = f4sub(d[2][j], d[1][j]);
    b[3][j] = f4sub(d[1][j], d[3][j]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {     // (.. B): columns (b0-b2, b1+b2, b2-b1, b1-b3)
    const float4 o[4] = {f4sub(b[i][0], b[i][2]), f4add(b[i][1], b[i][2]), f4sub(b[i][2], b[i][1]), f4sub(b[i][1], b[i][3])};
    const size_t r[4] = {(size_t)(i * 4 + 0) * T + t, (size_t)(i * 4 + 1) * T + t, (size_t)(i * 4 + 2) * T + t, (size_t)(i * 4 + 3) * T + t};
    V.template putn<4>(r, o, 0xfu, c4, C4);
  }
}

// ---------------------------------------------------------------------------------------------------- F(4x4, 3x3)
__device__ __forceinline__ float4 f4mad(float a, float4 x, float4 y) { return make_float4(a * x.x + y.x, a * x.y + y.y, a * x.z + y.z, a * x.w + y.w); }
__device__ __forceinline__ float4 f4mul(float a, float4 x) { return make_float4(a * x.x, a * x.y, a * x.z, a * x.w); }

// one application of B^T (6 -> 6) on a strided 6-vector held in registers
#define WINO4_BT(v0, v1, v2, v3, v4, v5, o0, o1, o2, o3, o4, o5)                       \
  {                                                                                    \
    const float4 t0 = f4mad(-4.f, v2, v4), t1 = f4mad(-4.f, v1, v3);                   \
    const float4 t2 = f4sub(v4, v2), t3 = f4mul(2.f, f4sub(v3, v1));                   \
    const float4 r0 = f4add(f4mad(4.f, v0, f4mul(-5.f, v2)), v4);                      \
    const float4 r5 = f4add(f4mad(4.f, v1, f4mul(-5.f, v3)), v5);                      \
    o0 = r0;               /* outputs may alias the inputs: all reads are done */      \
    o1 = f4add(t0, t1);                                                                \
    o2 = f4sub(t0, t1);                                                                \
    o3 = f4add(t2, t3);                                                                \
    o4 = f4sub(t2, t3);                                                                \
    o5 = r5;                                                                           \
  }

template <bool H2>
__global__ void __launch_bounds__(256) k_wino4_input(const float4* __restrict__ x, int N, int H, int W, int C4, int TH, int TW,
                                                      const WinoSink<H2> V) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long T = (long long)N * TH * TW;
  if (id >= T * C4) return;
  const int c4 = (int)(id % C4);
  const long long t = id / C4;
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), img = (int)(t / ((long long)TW * TH));
  float4 d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int ih = 4 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int iw = 4 * tx - 1 + j;
      d[i][j] = ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) ? x[((size_t)(img * H + ih) * W + iw) * C4 + c4]
                                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) WINO4_BT(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float4 o[6];
    WINO4_BT(d[i][0], d[i][1], d[i][2], d[i][3], d[i][4], d[i][5], o[0], o[1], o[2], o[3], o[4], o[5]);
    size_t r[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) r[j] = (size_t)(i * 6 + j) * T + t;
    V.template putn<6>(r, o, 0x3fu, c4, C4);
  }
}

// The same transform with ONE row xi of V per thread (6 x the threads) for launches that do not fill the chip in the tile-per-thread
// form: a single 38 x 63 image is 160 tiles x C/4 threads = 40 workgroups of 36 dependent loads + 36 stores per thread, i.e. a
// latency chain on a sixth of the CUs (12-19 us per launch, 2 x 36 launches per ResNet-152 training step).  Row xi needs the input rows
// B^T[xi] touches (3 or 4 of the 6); every value is formed by the SAME macro expressions as in k_wino4_input -- unused results of the
// column pass are dead code -- so both forms give the same bits (tests/test_dense_gpu.py: small launch == slot of a big one).
template <int I, bool H2>
__device__ __forceinline__ void wino4_input_row(const float4* __restrict__ x, int H, int W, int C4, int img, int ty, int tx, int c4, long long t,
                                                long long T, const WinoSink<H2>& V) {
  // rows of d that B^T row I reads: 0 -> (0, 2, 4); 5 -> (1, 3, 5); 1..4 -> (1, 2, 3, 4)
  constexpr unsigned need = I == 0 ? 0x15u : I == 5 ? 0x2au : 0x1eu;
  float4 d[6][6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int ih = 4 * ty - 1 + r;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int iw = 4 * tx - 1 + j;
      d[r][j] = (((need >> r) & 1u) && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) ? x[((size_t)(img * H + ih) * W + iw) * C4 + c4]
                                                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float4 b[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float4 c[6];
    WINO4_BT(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], c[0], c[1], c[2], c[3], c[4], c[5]);
    b[j] = c[I];
  }
  float4 o[6];
  WINO4_BT(b[0], b[1], b[2], b[3], b[4], b[5], o[0], o[1], o[2], o[3], o[4], o[5]);
  size_t r[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) r[j] = (size_t)(I * 6 + j) * T + t;
  V.template putn<6>(r, o, 0x3fu, c4, C4);
}

template <bool H2>
__global__ void __launch_bounds__(256) k_wino4_input_rows(const float4* __restrict__ x, int N, int H, int W, int C4, int TH, int TW,
                                                           const WinoSink<H2> V) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long T = (long long)N * TH * TW;
  if (id >= T * C4 * 6) return;
  const int c4 = (int)(id % C4);
  const int i = (int)((id / C4) % 6);              // (wave-uniform for C >= 256; with H2 the 32 lanes of a scale group always share it)
  const long long t = id / ((long long)C4 * 6);
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), img = (int)(t / ((long long)TW * TH));
  switch (i) {
    case 0: wino4_input_row<0, H2>(x, H, W, C4, img, ty, tx, c4, t, T, V); break;
    case 1: wino4_input_row<1, H2>(x, H, W, C4, img, ty, tx, c4, t, T, V); break;
    case 2: wino4_input_row<2, H2>(x, H, W, C4, img, ty, tx, c4, t, T, V); break;
    case 3: wino4_input_row<3, H2>(x, H, W, C4, img, ty, tx, c4, t, T, V); break;
    case 4: wino4_input_row<4, H2>(x, H, W, C4, img, ty, tx, c4, t, T, V); break;
    default: wino4_input_row<5, H2>(x, H, W, C4, img, ty, tx, c4, t, T, V); break;
  }
}

// A^T (6 -> 4): rows (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1)
#define WINO4_AT(v0, v1, v2, v3, v4, v5, o0, o1, o2, o3)                               \
  {                                                                                    \
    const float4 p = f4add(v1, v2), q = f4sub(v1, v2), r = f4add(v3, v4), u = f4sub(v3, v4); \
    o0 = f4add(f4add(v0, p), r);                                                       \
    o1 = f4mad(2.f, u, q);                                                             \
    o2 = f4mad(4.f, r, p);                                                             \
    o3 = f4add(f4mad(8.f, u, q), v5);                                                  \
  }

// MASK (training, frcnn_winograd_output_transform_masked): y = mask > 0 ? y : 0 -- the ReLU gradient of the tensor y is the gradient of
template <bool H2, bool MASK = false>
__global__ void __launch_bounds__(256) k_wino4_output(const float4* __restrict__ Mx, int N, int H, int W, int C4, int TH, int TW,
                                                       const float4* __restrict__ bias, int act, const float4* __restrict__ mask,
                                                       const WinoSink<H2> y) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long T = (long long)N * TH * TW;
  if (id >= T * C4) return;
  const int c4 = (int)(id % C4);
  const long long t = id / C4;
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), img = (int)(t / ((long long)TW * TH));
  const size_t plane = (size_t)T * C4;
  const float4* in = Mx + (size_t)t * C4 + c4;
  float4 s[4][6];       // A^T m, one column j at a time
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float4 m0 = in[(size_t)(0 * 6 + j) * plane], m1 = in[(size_t)(1 * 6 + j) * plane], m2 = in[(size_t)(2 * 6 + j) * plane];
    const float4 m3 = in[(size_t)(3 * 6 + j) * plane], m4 = in[(size_t)(4 * 6 + j) * plane], m5 = in[(size_t)(5 * 6 + j) * plane];
    WINO4_AT(m0, m1, m2, m3, m4, m5, s[0][j], s[1][j], s[2][j], s[3][j]);
  }
  const float4 bv = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int oh = 4 * ty + a;
    if (oh >= H) continue;
    float4 o[4];
    WINO4_AT(s[a][0], s[a][1], s[a][2], s[a][3], s[a][4], s[a][5], o[0], o[1], o[2], o[3]);
    const size_t row = (size_t)(img * H + oh) * W;
    size_t r[4];
    unsigned valid = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ow = 4 * tx + b;
      r[b] = row + ow;
      valid |= (ow < W ? 1u : 0u) << b;
      float4 v = f4add(o[b], bv);
      if (act == FRCNN_ACT_RELU) v = act_relu(v);
      if (MASK && ow < W) {
        const float4 k = mask[r[b] * C4 + c4];
        v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
      }
      o[b] = v;
    }
    y.template putn<4>(r, o, valid, c4, C4);
  }
}

// One output row a (of the tile's 4) per thread, for launches that do not fill the chip tile-per-thread (see k_wino4_input_rows): the same
// macro expressions, the same 4 pixels written together (one shared plane scale per group of rows, as in k_wino4_output) -> the same bits.
template <int A, bool H2, bool MASK>
__device__ __forceinline__ void wino4_output_row(const float4* __restrict__ in, size_t plane, int H, int W, int C4, int img, int ty, int tx, int c4,
                                                 const float4* __restrict__ bias, int act, const float4* __restrict__ mask,
                                                 const WinoSink<H2>& y) {
  const int oh = 4 * ty + A;
  if (oh >= H) return;                 // (the 32 lanes of a plane-scale group share (t, A): they leave together)
  float4 sa[6];                        // row A of A^T m, one column j at a time; A^T row 0 reads m0..m4, rows 1 / 2 m1..m4, row 3 m1..m5
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 m0 = A == 0 ? in[(size_t)(0 * 6 + j) * plane] : z, m1 = in[(size_t)(1 * 6 + j) * plane], m2 = in[(size_t)(2 * 6 + j) * plane];
    const float4 m3 = in[(size_t)(3 * 6 + j) * plane], m4 = in[(size_t)(4 * 6 + j) * plane], m5 = A == 3 ? in[(size_t)(5 * 6 + j) * plane] : z;
    float4 sv[4];
    WINO4_AT(m0, m1, m2, m3, m4, m5, sv[0], sv[1], sv[2], sv[3]);
    sa[j] = sv[A];
  }
  const float4 bv = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 o[4];
  WINO4_AT(sa[0], sa[1], sa[2], sa[3], sa[4], sa[5], o[0], o[1], o[2], o[3]);
  const size_t row = (size_t)(img * H + oh) * W;
  size_t r[4];
  unsigned valid = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int ow = 4 * tx + b;
    r[b] = row + ow;
    valid |= (ow < W ? 1u : 0u) << b;
    float4 v = f4add(o[b], bv);
    if (act == FRCNN_ACT_RELU) v = act_relu(v);
    if (MASK && ow < W) {
      const float4 k = mask[r[b] * C4 + c4];
      v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
    }
    o[b] = v;
  }
  y.template putn<4>(r, o, valid, c4, C4);
}

template <bool H2, bool MASK = false>
__global__ void __launch_bounds__(256) k_wino4_output_rows(const float4* __restrict__ Mx, int N, int H, int W, int C4, int TH, int TW,
                                                            const float4* __restrict__ bias, int act, const float4* __restrict__ mask,
                                                            const WinoSink<H2> y) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long T = (long long)N * TH * TW;
  if (id >= T * C4 * 4) return;
  const int c4 = (int)(id % C4);
  const int a = (int)((id / C4) & 3);
  const long long t = id / ((long long)C4 * 4);
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), img = (int)(t / ((long long)TW * TH));
  const size_t plane = (size_t)T * C4;
  const float4* in = Mx + (size_t)t * C4 + c4;
  switch (a) {
    case 0: wino4_output_row<0, H2, MASK>(in, plane, H, W, C4, img, ty, tx, c4, bias, act, mask, y); break;
    case 1: wino4_output_row<1, H2, MASK>(in, plane, H, W, C4, img, ty, tx, c4, bias, act, mask, y); break;
    case 2: wino4_output_row<2, H2, MASK>(in, plane, H, W, C4, img, ty, tx, c4, bias, act, mask, y); break;
    default: wino4_output_row<3, H2, MASK>(in, plane, H, W, C4, img, ty, tx, c4, bias, act, mask, y); break;
  }
}

// launches below this many workgroups of the tile-per-thread form run the row-per-thread form (256 CUs).  Both forms evaluate the same
// expressions (same bits); frcnn_set_tuning(9, n) moves the threshold for the calling thread (A/B runs, scratch/wino_bench.py).
thread_local int g_wino_rows_below = 256;
#define WINO_ROWS_BELOW g_wino_rows_below

template <bool H2>
static int wino_input_launch(const float* x_d, int N, int H, int W, int C, int m, const WinoSink<H2>& sink, hipStream_t st) {
  const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
  const long long tot = (long long)N * TH * TW * (C / 4);
  if (m == 4 && (tot + 255) / 256 < WINO_ROWS_BELOW)
    hipLaunchKernelGGL(k_wino4_input_rows<H2>, dim3((unsigned)((tot * 6 + 255) / 256)), dim3(256), 0, st, (const float4*)x_d, N, H, W, C / 4, TH, TW,
                       sink);
  else if (m == 4)
    hipLaunchKernelGGL(k_wino4_input<H2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float4*)x_d, N, H, W, C / 4, TH, TW, sink);
  else
    hipLaunchKernelGGL(k_wino_input<H2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float4*)x_d, N, H, W, C / 4, TH, TW, sink);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_winograd_input_transform(const float* x_d, int N, int H, int W, int C, int m, float* v_d, void* stream) {
  if (!x_d || !v_d || N <= 0 || H <= 0 || W <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % 4 || (m != 2 && m != 4)) return FRCNN_E_UNSUPPORTED;
  const WinoSink<false> sink{(float4*)v_d, nullptr, nullptr, 0};
  return wino_input_launch<false>(x_d, N, H, W, C, m, sink, (hipStream_t)stream);
}

// The same transform with V emitted as frcnn_gemm_h2 operand planes [2][(m+2)^2 * T][C] + v_inv [C/128][(m+2)^2 * T]; C % 128 == 0
extern "C" int frcnn_winograd_input_transform_h2(const float* x_d, int N, int H, int W, int C, int m, void* v_planes_d, float* v_inv_d,
                                                 void* stream) {
  if (!x_d || !v_planes_d || !v_inv_d || N <= 0 || H <= 0 || W <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % H2_KB || (m != 2 && m != 4)) return FRCNN_E_UNSUPPORTED;
  const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
  const WinoSink<true> sink{nullptr, (unsigned short*)v_planes_d, v_inv_d, (size_t)(m + 2) * (m + 2) * N * TH * TW};
  return wino_input_launch<true>(x_d, N, H, W, C, m, sink, (hipStream_t)stream);
}

// y[img, 2ty+a, 2tx+b, o] = act( (A^T M A)[a][b] + bias[o] ),  M[xn][t][o]
template <bool H2>
__global__ void k_wino_output(const float4* __restrict__ Mx, int N, int H, int W, int C4, int TH, int TW,
                              const float4* __restrict__ bias, int act, const WinoSink<H2> y) {
  const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long T = (long long)N * TH * TW;
  if (id >= T * C4) return;
  const int c4 = (int)(id % C4);
  const long long t = id / C4;
  const int tx = (int)(t % TW), ty = (int)((t / TW) % TH), img = (int)(t / ((long long)TW * TH));
  const size_t plane = (size_t)T * C4;
  const float4* in = Mx + (size_t)t * C4 + c4;
  float4 m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = in[(size_t)(i * 4 + j) * plane];
  float4 s[2][4];       // A^T m : rows (m0+m1+m2, m1-m2-m3)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[0][j] = f4add(f4add(m[0][j], m[1][j]), m[2][j]);
    s[1][j] = f4sub(f4sub(m[1][j], m[2][j]), m[3][j]);
  }
  const float4 bv = bias ? bias[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oh = 2 * ty + a;
    if (oh >= H) continue;
    float4 o0 = f4add(f4add(f4add(s[a][0], s[a][1]), s[a][2]), bv);
    float4 o1 = f4add(f4sub(f4sub(s[a][1], s[a][2]), s[a][3]), bv);
    if (act == FRCNN_ACT_RELU) {
      o0 = act_relu(o0);
      o1 = act_relu(o1);
    }
    const size_t row = (size_t)(img * H + oh) * W;
    const float4 o[2] = {o0, o1};
    const size_t r[2] = {row + 2 * tx, row + 2 * tx + 1};
    y.template putn<2>(r, o, (2 * tx < W ? 1u : 0u) | (2 * tx + 1 < W ? 2u : 0u), c4, C4);
  }
}

template <bool H2>
static int wino_output_launch(const float* m_d, int N, int H, int W, int C, int m, const float* bias_d, int act, const WinoSink<H2>& sink,
                              hipStream_t st, const float* mask_d = nullptr) {
  const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
  const long long tot = (long long)N * TH * TW * (C / 4);
  // (measured, profiles/r06_g_wino_bench.txt: twice the threshold would take block3's output transform at 8 images from 16.5 to 14.2 us
  // but the row form loses 2 x on wide layers -- RPN, C = 512: 48 -> 89 us -- so the threshold stays one chip's worth of workgroups)
  const bool rows = (tot + 255) / 256 < WINO_ROWS_BELOW;
  if (m == 4 && mask_d && rows)
    hipLaunchKernelGGL((k_wino4_output_rows<H2, true>), dim3((unsigned)((tot * 4 + 255) / 256)), dim3(256), 0, st, (const float4*)m_d, N, H, W,
                       C / 4, TH, TW, (const float4*)bias_d, act, (const float4*)mask_d, sink);
  else if (m == 4 && mask_d)
    hipLaunchKernelGGL((k_wino4_output<H2, true>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float4*)m_d, N, H, W, C / 4, TH,
                       TW, (const float4*)bias_d, act, (const float4*)mask_d, sink);
  else if (mask_d)
    return FRCNN_E_UNSUPPORTED;
  else if (m == 4 && rows)
    hipLaunchKernelGGL((k_wino4_output_rows<H2, false>), dim3((unsigned)((tot * 4 + 255) / 256)), dim3(256), 0, st, (const float4*)m_d, N, H, W,
                       C / 4, TH, TW, (const float4*)bias_d, act, (const float4*)nullptr, sink);
  else if (m == 4)
    hipLaunchKernelGGL((k_wino4_output<H2, false>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float4*)m_d, N, H, W, C / 4, TH,
                       TW, (const float4*)bias_d, act, (const float4*)nullptr, sink);
  else
    hipLaunchKernelGGL(k_wino_output<H2>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float4*)m_d, N, H, W, C / 4, TH, TW,
                       (const float4*)bias_d, act, sink);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_winograd_output_transform(const float* m_d, int N, int H, int W, int C, int m, const float* bias_d, int act,
                                               float* y_d, void* stream) {
  if (!m_d || !y_d || N <= 0 || H <= 0 || W <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % 4 || (m != 2 && m != 4) || (act != FRCNN_ACT_NONE && act != FRCNN_ACT_RELU)) return FRCNN_E_UNSUPPORTED;
  const WinoSink<false> sink{(float4*)y_d, nullptr, nullptr, 0};
  return wino_output_launch<false>(m_d, N, H, W, C, m, bias_d, act, sink, (hipStream_t)stream);
}

// The same transform with the result emitted as operand planes [2][N*H*W][C] + y_inv [C/128][N*H*W] for a following frcnn_gemm_h2
// (the bottleneck's conv3); y_d may be NULL (planes only) or receives the float32 result as well.  C % 128 == 0.
extern "C" int frcnn_winograd_output_transform_h2(const float* m_d, int N, int H, int W, int C, int m, const float* bias_d, int act,
                                                  float* y_d, void* y_planes_d, float* y_inv_d, void* stream) {
  if (!m_d || !y_planes_d || !y_inv_d || N <= 0 || H <= 0 || W <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % H2_KB || (m != 2 && m != 4) || (act != FRCNN_ACT_NONE && act != FRCNN_ACT_RELU)) return FRCNN_E_UNSUPPORTED;
  const WinoSink<true> sink{(float4*)y_d, (unsigned short*)y_planes_d, y_inv_d, (size_t)N * H * W};
  return wino_output_launch<true>(m_d, N, H, W, C, m, bias_d, act, sink, (hipStream_t)stream);
}

// Training (the data-gradient chain of a 3x3 convolution): the output transform followed by the ReLU gradient of the tensor the result
// is the gradient OF, y = mask > 0 ? A^T M A : 0 (mask [N,H,W,C] float32, the forward activation), as float32 (y_d) and / or as the
// operand planes of the next data-gradient GEMM (y_planes_d + y_inv_d, C % 128 == 0) -- instead of frcnn_relu_bwd (+ frcnn_h2_split)
// passes over the result.  m = 4 only.  The float32 result is bit for bit the unfused sequence (exact select).
extern "C" int frcnn_winograd_output_transform_masked(const float* m_d, int N, int H, int W, int C, int m, const float* mask_d, float* y_d,
                                                      void* y_planes_d, float* y_inv_d, void* stream) {
  if (!m_d || !mask_d || (!y_d && !y_planes_d) || (y_planes_d && !y_inv_d) || N <= 0 || H <= 0 || W <= 0 || C <= 0) return FRCNN_E_ARG;
  if (C % 4 || m != 4 || (y_planes_d && C % H2_KB)) return FRCNN_E_UNSUPPORTED;
  if (y_planes_d) {
    const WinoSink<true> sink{(float4*)y_d, (unsigned short*)y_planes_d, y_inv_d, (size_t)N * H * W};
    return wino_output_launch<true>(m_d, N, H, W, C, m, nullptr, FRCNN_ACT_NONE, sink, (hipStream_t)stream, mask_d);
  }
  const WinoSink<false> sink{(float4*)y_d, nullptr, nullptr, 0};
  return wino_output_launch<false>(m_d, N, H, W, C, m, nullptr, FRCNN_ACT_NONE, sink, (hipStream_t)stream, mask_d);
}

// ---------------------------------------------------------------------------------------------------- filter transform on device
// Training changes the filters every step, so U = G g G^T is recomputed on the device from the PACKED filter
// [Cout][3][3][Cin] the convolution kernels use.  transpose_flip = 0: U[xi][o][c] for the forward convolution.
// transpose_flip = 1: the data-gradient convolution's filter g'[kh][kw] = w[o][2-kh][2-kw][c] with the channel roles swapped,
// U'[xi][c][o] (csrc/backward_kernels.hip k_flip_transpose composed with the transform).  f32 arithmetic.
template <int M>
__global__ void k_wino_filter(const float* __restrict__ w, int O, int C, int transpose_flip, float* __restrict__ U) {
  // coalesce the 36 / 16 writes: consecutive threads run along the FAST axis of the output (c forward, o for transpose_flip)
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
  constexpr int A = M + 2;
  float t[A][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float g0 = g[0][j], g1 = g[1][j], g2 = g[2][j];
    if (M == 2) {
      t[0][j] = g0;
      t[1][j] = 0.5f * (g0 + g1 + g2);
      t[2][j] = 0.5f * (g0 - g1 + g2);
      t[3][j] = g2;
    } else {
      t[0][j] = 0.25f * g0;
      t[1][j] = (-1.0f / 6.0f) * (g0 + g1 + g2);
      t[2][j] = (-1.0f / 6.0f) * (g0 - g1 + g2);
      t[3][j] = (1.0f / 24.0f) * g0 + (1.0f / 12.0f) * g1 + (1.0f / 6.0f) * g2;
      t[4][j] = (1.0f / 24.0f) * g0 - (1.0f / 12.0f) * g1 + (1.0f / 6.0f) * g2;
      t[A - 1][j] = g2;
    }
  }
  const size_t plane = (size_t)O * C;
  float* out = U + (transpose_flip ? (size_t)c * O + o : (size_t)o * C + c);
#pragma unroll
  for (int i = 0; i < A; ++i) {
    const float a0 = t[i][0], a1 = t[i][1], a2 = t[i][2];
    float r[A];
    if (M == 2) {
      r[0] = a0; r[1] = 0.5f * (a0 + a1 + a2); r[2] = 0.5f * (a0 - a1 + a2); r[3] = a2;
    } else {
      r[0] = 0.25f * a0;
      r[1] = (-1.0f / 6.0f) * (a0 + a1 + a2);
      r[2] = (-1.0f / 6.0f) * (a0 - a1 + a2);
      r[3] = (1.0f / 24.0f) * a0 + (1.0f / 12.0f) * a1 + (1.0f / 6.0f) * a2;
      r[4] = (1.0f / 24.0f) * a0 - (1.0f / 12.0f) * a1 + (1.0f / 6.0f) * a2;
      r[A - 1] = a2;
    }
#pragma unroll
    for (int j = 0; j < A; ++j) out[(size_t)(i * A + j) * plane] = r[j];
  }
}

extern "C" int frcnn_winograd_filter_transform_device(const float* w_packed_d, int Cout, int Cin, int m, int transpose_flip,
                                                      float* u_d, void* stream) {
  if (!w_packed_d || !u_d || Cout <= 0 || Cin <= 0) return FRCNN_E_ARG;
  if (m != 2 && m != 4) return FRCNN_E_UNSUPPORTED;
  const long long tot = (long long)Cout * Cin;
  const dim3 grid((unsigned)((tot + 255) / 256)), block(256);
  if (m == 2) hipLaunchKernelGGL(k_wino_filter<2>, grid, block, 0, (hipStream_t)stream, w_packed_d, Cout, Cin, transpose_flip ? 1 : 0, u_d);
  else hipLaunchKernelGGL(k_wino_filter<4>, grid, block, 0, (hipStream_t)stream, w_packed_d, Cout, Cin, transpose_flip ? 1 : 0, u_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
