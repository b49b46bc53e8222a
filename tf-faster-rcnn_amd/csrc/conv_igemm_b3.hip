// EXPERIMENTAL (opt-in, frcnn_set_tuning(2, 1)): f32 convolution on the bf16 matrix pipe with exactly split
// operands ("bf16x3").
//
// Every f32 operand is split EXACTLY into three bf16 pieces by truncation, x = h + m + l (8 + 8 + 8 significand
// bits), and a product a*b is evaluated as the six leading cross terms
//     ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)
// on v_mfma_f32_32x32x16_bf16 with f32 accumulation; the three dropped terms are < 2^-23 relative, i.e. below
// f32 rounding.  Six bf16 MFMAs (32 cycles each, 16 k-values) replace eight f32 MFMAs (64 cycles each, 2 k-values
// each): 192 vs 512 matrix-pipe cycles per 16 k -> 2.67x the f32-MFMA rate at f32-class accuracy.  The leading term
// and the five small terms go to SEPARATE accumulators so the small terms do not lose bits against a large sum.
//
// Structure: register-staged double buffer.  A thread loads its float4 chunks of the next activation / filter slab
// (same implicit-GEMM gather as conv_igemm.hip), the MFMAs of the current slab run, then the thread splits its
// chunks (v_and / v_sub / v_perm: ~5.5 VALU per element) and writes three bf16 planes into the other LDS buffer.
// LDS row = 3 planes x 64 B + 16 B pad = 208 B: 13 16-byte slots per row (odd), so the ds_read_b128 fragment reads
// (lane&31 = row) are conflict free.  Same tile order, epilogue and ABI as the f32 kernel.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct ConvParamsB3 {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad_top, pad_left, act;
  int RH, RW, res_stride;
  int M, Ktot, nsteps;
  int mtiles, ntiles;
};

__device__ float4 g_zero_page_b3[4];

constexpr int ROWB = 208;            // bytes per LDS row: planes h | m | l (64 B each) + pad

// exact truncation split of 4 consecutive k values into three packed-bf16 pairs
__device__ __forceinline__ void split4(const float4 v, uint2& h, uint2& m, uint2& l) {
  const u32 x0 = __float_as_uint(v.x), x1 = __float_as_uint(v.y), x2 = __float_as_uint(v.z), x3 = __float_as_uint(v.w);
  h = make_uint2(__builtin_amdgcn_perm(x1, x0, 0x07060302), __builtin_amdgcn_perm(x3, x2, 0x07060302));
  const float r0 = v.x - __uint_as_float(x0 & 0xffff0000u), r1 = v.y - __uint_as_float(x1 & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(x2 & 0xffff0000u), r3 = v.w - __uint_as_float(x3 & 0xffff0000u);
  const u32 y0 = __float_as_uint(r0), y1 = __float_as_uint(r1), y2 = __float_as_uint(r2), y3 = __float_as_uint(r3);
  m = make_uint2(__builtin_amdgcn_perm(y1, y0, 0x07060302), __builtin_amdgcn_perm(y3, y2, 0x07060302));
  const float s0 = r0 - __uint_as_float(y0 & 0xffff0000u), s1 = r1 - __uint_as_float(y1 & 0xffff0000u);
  const float s2 = r2 - __uint_as_float(y2 & 0xffff0000u), s3 = r3 - __uint_as_float(y3 & 0xffff0000u);
  l = make_uint2(__builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302),
                 __builtin_amdgcn_perm(__float_as_uint(s3), __float_as_uint(s2), 0x07060302));
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void k_conv_igemm_b3(const ConvParamsB3 p) {
  constexpr int NW = (BM / WM) * (BN / WN);
  constexpr int NT = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = (BM * 8) / NT, LB = (BN * 8) / NT;
  constexpr int SLABB = (BM + BN) * ROWB;
  static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem_b3[];        // [2][BM+BN][208 B]; reused by the epilogue

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = p.mtiles * p.ntiles;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = bid / p.ntiles, nt = bid % p.ntiles;
  const int bm0 = mt * BM, bn0 = nt * BN;
  const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;

  const float* zero = (const float*)g_zero_page_b3;
  const int q4 = (tid & 7) * 4;
  int a_base[LA], a_ih0[LA], a_iw0[LA];
#pragma unroll
  for (int l = 0; l < LA; ++l) {
    const int m = bm0 + ((tid + l * NT) >> 3);
    if (m < p.M) {
      const int img = m / (p.OH * p.OW), rem = m % (p.OH * p.OW);
      const int oh = rem / p.OW, ow = rem % p.OW;
      a_base[l] = img * p.H * p.W * p.Cin;
      a_ih0[l] = oh * p.stride - p.pad_top;
      a_iw0[l] = ow * p.stride - p.pad_left;
    } else {
      a_base[l] = 0; a_ih0[l] = -(1 << 28); a_iw0[l] = -(1 << 28);
    }
  }
  const float* a_ptr[LA]; int a_inc[LA];
  const float* b_ptr[LB]; int b_inc[LB];
#pragma unroll
  for (int l = 0; l < LB; ++l) {
    const int n = bn0 + ((tid + l * NT) >> 3);
    const bool ok = n < p.Cout;
    b_ptr[l] = ok ? p.w + (size_t)n * p.Ktot + q4 : zero;
    b_inc[l] = ok ? 32 : 0;
  }
  int kh = 0, kw = 0, c0 = 0;
  float4 ra[1][LA], rb[1][LB];          // one staged slab (a second set was measured: the extra VGPRs cost more
                                        // occupancy than the deeper prefetch buys, scratch/conv_sweep.py)

  auto set_tap = [&]() {
#pragma unroll
    for (int l = 0; l < LA; ++l) {
      const int ih = a_ih0[l] + kh, iw = a_iw0[l] + kw;
      const bool ok = ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
      a_ptr[l] = ok ? p.x + (size_t)(a_base[l] + (ih * p.W + iw) * p.Cin + q4) : zero;
      a_inc[l] = ok ? 32 : 0;
    }
  };
  auto advance_k = [&]() {
    c0 += 32;
    if (c0 == p.Cin) { c0 = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
  };
#define B3_LOAD(S)                                                                           \
  {                                                                                          \
    if (c0 == 0) set_tap();                                                                  \
    _Pragma("unroll") for (int l = 0; l < LA; ++l) { ra[S][l] = *(const float4*)a_ptr[l]; a_ptr[l] += a_inc[l]; } \
    _Pragma("unroll") for (int l = 0; l < LB; ++l) { rb[S][l] = *(const float4*)b_ptr[l]; b_ptr[l] += b_inc[l]; } \
    advance_k();                                                                             \
  }
#define B3_STORE(S, BUF)                                                                     \
  {                                                                                          \
    char* sb_ = smem_b3 + (BUF) * SLABB;                                                     \
    _Pragma("unroll") for (int l = 0; l < LA; ++l) {                                         \
      uint2 h_, m_, l_;                                                                      \
      split4(ra[S][l], h_, m_, l_);                                                          \
      char* row_ = sb_ + ((tid + l * NT) >> 3) * ROWB + (tid & 7) * 8;                       \
      *(uint2*)(row_) = h_; *(uint2*)(row_ + 64) = m_; *(uint2*)(row_ + 128) = l_;           \
    }                                                                                        \
    _Pragma("unroll") for (int l = 0; l < LB; ++l) {                                         \
      uint2 h_, m_, l_;                                                                      \
      split4(rb[S][l], h_, m_, l_);                                                          \
      char* row_ = sb_ + (BM + ((tid + l * NT) >> 3)) * ROWB + (tid & 7) * 8;                \
      *(uint2*)(row_) = h_; *(uint2*)(row_ + 64) = m_; *(uint2*)(row_ + 128) = l_;           \
    }                                                                                        \
  }

  f32x16 acc[TM][TN], acs[TM][TN];                  // leading term / sum of the five small cross terms
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acs[i][j][r] = 0.f; }

  B3_LOAD(0);
  B3_STORE(0, 0);
  __syncthreads();

  const int frow = lane & 31, khalf = lane >> 5;
  auto compute = [&](int buf) {
    const char* Ab = smem_b3 + buf * SLABB + (wm0 + frow) * ROWB + khalf * 16;
    const char* Bb = smem_b3 + buf * SLABB + (BM + wn0 + frow) * ROWB + khalf * 16;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                     // two groups of 16 k-values per 32-wide slab
      bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* q = Ab + i * 32 * ROWB + t * 32;
        ah[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(q));
        am[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(q + 64));
        al[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(q + 128));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const char* q = Bb + j * 32 * ROWB + t * 32;
        bh[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(q));
        bm[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(q + 64));
        bl[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(q + 128));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acs[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], acs[i][j], 0, 0, 0);
          acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], acs[i][j], 0, 0, 0);
        }
    }
  };
  int buf = 0;
  for (int step = 0; step < p.nsteps; ++step) {
    const bool more = step + 1 < p.nsteps;
    if (more) B3_LOAD(0);                      // global loads in flight while the MFMAs of this slab run
    compute(buf);
    if (more) B3_STORE(0, buf ^ 1);            // split + three bf16 planes into the other buffer
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue (same as the f32 kernel): accumulators -> LDS tile [BM][BN] -> float4 rows
  float* tile = (float*)smem_b3;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        tile[ml * BN + wn0 + j * 32 + frow] = acc[i][j][r] + acs[i][j][r];
      }
  __syncthreads();
  const int ohow = p.OH * p.OW;
  const bool vec = (p.Cout & 3) == 0;
  constexpr int C4 = BN / 4;
  for (int t = tid; t < BM * C4; t += NT) {
    const int ml = t / C4, nl = (t % C4) * 4;
    const int m = bm0 + ml, n = bn0 + nl;
    if (m >= p.M || n >= p.Cout) continue;
    float v[4];
    *(float4*)v = *(const float4*)(tile + ml * BN + nl);
    size_t ro = 0;
    if (p.res) {
      if (p.res_stride == 1) ro = (size_t)m * p.Cout + n;
      else {
        const int img = m / ohow, rem = m % ohow, oh = rem / p.OW, ow = rem % p.OW;
        ro = ((size_t)(img * p.RH + oh * p.res_stride) * p.RW + ow * p.res_stride) * p.Cout + n;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e >= p.Cout) break;
      float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
      if (p.res) x += p.res[ro + e];
      if (p.act == FRCNN_ACT_RELU) x = fmaxf(x, 0.f);
      else if (p.act == FRCNN_ACT_RELU6) x = fminf(fmaxf(x, 0.f), 6.f);
      v[e] = x;
    }
    if (vec) *(float4*)(p.y + (size_t)m * p.Cout + n) = *(float4*)v;
    else
      for (int e = 0; e < 4 && n + e < p.Cout; ++e) p.y[(size_t)m * p.Cout + n + e] = v[e];
  }
}

template <int BM, int BN, int WM, int WN>
static int launch_b3(ConvParamsB3 p, hipStream_t st) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr size_t ring = (size_t)2 * (BM + BN) * ROWB, epi = sizeof(float) * BM * BN;
  constexpr size_t lds = ring > epi ? ring : epi;
  static bool attr_set = false;
  auto kern = k_conv_igemm_b3<BM, BN, WM, WN>;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  p.mtiles = cdiv(p.M, BM);
  p.ntiles = cdiv(p.Cout, BN);
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles), dim3(NT), lds, st, p);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// called from frcnn_conv2d_nhwc when the experimental path is enabled (Cin % 32 == 0, no fold_w)
int frcnn_conv2d_b3_dispatch(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                             const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW, int Cout,
                             int KH, int KW, int stride, int pad_top, int pad_left, int act, int cfg, hipStream_t st) {
  ConvParamsB3 p;
  p.x = x_d; p.w = w_d; p.bias = bias_d; p.res = residual_d; p.y = y_d;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad_top = pad_top; p.pad_left = pad_left; p.act = act;
  p.RH = RH; p.RW = RW; p.res_stride = residual_d ? res_stride : 1;
  p.M = N * OH * OW;
  p.Ktot = KH * KW * Cin;
  p.nsteps = KH * KW * (Cin / 32);
  p.mtiles = p.ntiles = 0;
  if (cfg < 0) {
    const long long big = (long long)cdiv(p.M, 128) * cdiv(Cout, 128);
    // measured (scratch/conv_sweep.py): 8-wave 128x128 / 128x64 tiles for the per-RoI tail, 64x64 for the feature maps
    cfg = (Cout >= 96 && big >= 384 && p.nsteps >= 8) ? (Cout >= 1024 ? 5 : 4) : (Cout > 32 ? 2 : 3);
  }
  switch (cfg) {
    case 0: return launch_b3<128, 128, 64, 64>(p, st);      // 104 KB LDS, 1 workgroup / CU
    case 1: return launch_b3<128, 64, 64, 32>(p, st);       // 78 KB, 2 / CU
    case 2: return launch_b3<64, 64, 32, 32>(p, st);        // 52 KB, 3 / CU
    case 3: return launch_b3<64, 32, 32, 32>(p, st);
    case 4: return launch_b3<128, 128, 32, 64>(p, st);      // 8 waves
    case 5: return launch_b3<128, 64, 32, 32>(p, st);       // 8 waves, 2 / CU
    default: return FRCNN_E_ARG;
  }
}
