// Filter gradient of an NHWC convolution on the fp16 matrix pipe with block-scaled two-piece operands (the "h2" arithmetic of
// gemm_h2.hip), read straight from the tensors the reverse sweep holds:
//
//   dW[n, (kh, kw, c)] = sum_m  dY[m, n] * X[pix(m) + (kh, kw), c]          m = output pixel (img, oh, ow), n = output channel
//
// The reduction runs over PIXELS, the slow axis of both NHWC operands, while v_mfma_f32_32x32x16_f16 wants 8 consecutive k per lane.
// The f32 values have to pass through registers anyway to be split into fp16 pieces, so the transposition costs nothing extra:
// a thread loads float4s of 4 channels for RPT consecutive pixels, and writes, per channel, those RPT pixels as ONE 8- / 16-byte
// piece of a [channel][64 pixel] plane row in LDS -- the layout a fragment read (ds_read_b128, chunk ^ ((row >> 1) & 7) swizzle as
// in conv_igemm.hip) expects.  No transposed copy of dY, no im2col matrix and no operand planes ever exist in HBM.
//
// Operand format: a slab is 64 pixels; every (channel, slab) column segment of dY and of the X tap gets its own exact power-of-two
// scale 2^e with max |v| 2^e in [2^14, 2^15) (h2_block_scale: the maximum is reduced over the thread's rows, the lanes sharing the
// channels (ds_bpermute), then the four waves through LDS ds_max_u32), v 2^e = h + l in two fp16 roundings, and the slab's product
// sum_64 (h + l)(h' + l') without the l l' term (three MFMAs per 16 pixels, <= 3 * 2^-22 relative) lands in a scratch accumulator
// that is folded into the running f32 sum by  2^-e[n] * 2^-e'[c]  -- one multiply + one fma per element and slab.
//
// Work split as in wgrad_tn.hip: grid.x = output tiles (BT x BT, a column tile inside one tap: Cin % BT == 0), grid.z = S slices
// of the pixel range writing raw partials, k_wgrad_h2_finish adds them in a fixed order (deterministic, no atomics in HBM).
// Pipeline per slab: [regs hold the slab's f32 values, loaded under the previous slab's MFMAs] column maxima -> barrier -> scales,
// split, plane rows + inverse scales to LDS -> barrier -> refill the register set with a later slab -> 12 T^2 MFMAs + fold (two barriers
// per slab: the maxima buffers alternate, so the next slab's ds_max needs no barrier after this slab's fragment reads).
// 32 KB (BT = 64) / 64 KB (BT = 128) of LDS per workgroup, >= 2 workgroups per CU: the others' MFMAs cover this one's split phase.
#include "h2_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct WgradH2Params {
  const float* gy; const float* x; float* out;      // out: [S][Cout][Kf] partials (or the gradient itself when S == 1)
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad_top, pad_left;
  int M, Kf, ntn;
  int nslabs, chunk;                                // ceil(M / 64); slabs per slice
  int direct;                                       // 1x1, stride 1, no padding
  long long gz;
};

template <int BT>
__global__ __launch_bounds__(256, 2) void k_wgrad_h2(const WgradH2Params p) {
  constexpr int T = BT / 64;                 // 32x32 MFMA tiles per wave and direction (waves 2 x 2)
  constexpr int NSET = BT == 64 ? 2 : 1;     // register sets of slab values in flight (64: two slabs ahead; 128: the VGPR file allows one)
  constexpr int KB = 64;                     // pixels per slab = per scale block
  constexpr int CG = BT / 4;                 // float4 channel groups per tile row: 16 / 32
  constexpr int RG = 256 / CG;               // pixel groups: 16 / 8
  constexpr int RPT = KB / RG;               // consecutive pixels per thread: 4 / 8
  constexpr int PLANE = BT * KB * 2;         // bytes of one fp16 plane [BT][64]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sAh = smem;                    // dY planes h, l then X planes h, l
  char* const sAl = smem + PLANE;
  char* const sBh = smem + 2 * PLANE;
  char* const sBl = smem + 3 * PLANE;
  unsigned* const smax = (unsigned*)(smem + 4 * PLANE);              // [2 buffers][2 operands][BT] column maxima (bit patterns)
  float* const sinv = (float*)(smem + 4 * PLANE) + 4 * BT;           // [2 operands][BT] inverse scales of the slab

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tn = blockIdx.x % p.ntn, tm = blockIdx.x / p.ntn;
  const int co0 = tm * BT, kf0 = tn * BT;
  const int tap = kf0 / p.Cin, c0 = kf0 - tap * p.Cin;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int s_begin = (int)blockIdx.z * p.chunk;
  const int nloc = min(p.nslabs - s_begin, p.chunk);
  float* const out = p.out + (size_t)blockIdx.z * p.gz;

  const int cg = tid % CG, rg = tid / CG;
  const int ohow = p.OH * p.OW;

  float4 rav[NSET][RPT], rbv[NSET][RPT];
  auto load_slab = [&](float4* ra, float4* rb, int slab) {
    const int m0 = (s_begin + slab) * KB + rg * RPT;
#pragma unroll
    for (int e = 0; e < RPT; ++e) {
      const int m = m0 + e;
      ra[e] = m < p.M ? *(const float4*)(p.gy + (size_t)m * p.Cout + co0 + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.direct) {
#pragma unroll
      for (int e = 0; e < RPT; ++e) {
        const int m = m0 + e;
        rb[e] = m < p.M ? *(const float4*)(p.x + (size_t)m * p.Cin + c0 + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      int img = m0 / ohow, rem = m0 - img * ohow;
      int oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
      for (int e = 0; e < RPT; ++e) {
        const int ih = oh * p.stride - p.pad_top + kh, iw = ow * p.stride - p.pad_left + kw;
        const bool ok = (m0 + e < p.M) && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        rb[e] = ok ? *(const float4*)(p.x + ((size_t)(img * p.H + ih) * p.W + iw) * p.Cin + c0 + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (++ow == p.OW) { ow = 0; if (++oh == p.OH) { oh = 0; ++img; } }
      }
    }
  };

  f32x16 tot[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;

  // fragment rows / swizzle of this lane, plane-row write position of this thread
  const int frow = lane & 31, khalf = lane >> 5;
  const int wi0 = (wave >> 1) * (BT / 2), wj0 = (wave & 1) * (BT / 2);
  for (int i = tid; i < 4 * BT; i += 256) smax[i] = 0u;
#pragma unroll
  for (int u = 0; u < NSET; ++u)
    if (u < nloc) load_slab(rav[u], rbv[u], u);
  __syncthreads();

  // one slab: `ra` / `rb` hold its values; once they are split the set is refilled with slab s + NSET
  auto slab_body = [&](float4* ra, float4* rb, int s) {
    unsigned* const mxbuf = smax + (s & 1) * 2 * BT;
    // ---- column maxima of the slab: rows of this thread, lanes with the same channels, then the four waves -----------------------
    {
      unsigned ma[4] = {0u, 0u, 0u, 0u}, mb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < RPT; ++e) {
        ma[0] = max(ma[0], h2_abs_bits(ra[e].x)); ma[1] = max(ma[1], h2_abs_bits(ra[e].y));
        ma[2] = max(ma[2], h2_abs_bits(ra[e].z)); ma[3] = max(ma[3], h2_abs_bits(ra[e].w));
        mb[0] = max(mb[0], h2_abs_bits(rb[e].x)); mb[1] = max(mb[1], h2_abs_bits(rb[e].y));
        mb[2] = max(mb[2], h2_abs_bits(rb[e].z)); mb[3] = max(mb[3], h2_abs_bits(rb[e].w));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int d = CG; d < 64; d <<= 1) {
          ma[q] = max(ma[q], (unsigned)__shfl_xor((int)ma[q], d, 64));
          mb[q] = max(mb[q], (unsigned)__shfl_xor((int)mb[q], d, 64));
        }
      }
      if (lane < CG) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          atomicMax(&mxbuf[4 * cg + q], ma[q]);
          atomicMax(&mxbuf[BT + 4 * cg + q], mb[q]);
        }
      }
    }
    __syncthreads();
    // ---- scales, split, plane rows -------------------------------------------------------------------------------------------------
    {
      const uint4 xa = *(const uint4*)(mxbuf + 4 * cg), xb = *(const uint4*)(mxbuf + BT + 4 * cg);
      const unsigned mxa[4] = {xa.x, xa.y, xa.z, xa.w}, mxb[4] = {xb.x, xb.y, xb.z, xb.w};
      float inva[4], invb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 4 * cg + q;
        float sc;
        h2_block_scale_bits(mxa[q], sc, inva[q]);
        _Float16 hh[RPT], ll[RPT];
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          const float v = q == 0 ? ra[e].x : q == 1 ? ra[e].y : q == 2 ? ra[e].z : ra[e].w;
          h2_split1(v, sc, hh[e], ll[e]);
        }
        const int pos = ch * (KB * 2) + ((((rg * RPT) >> 3) ^ ((ch >> 1) & 7)) << 4) + ((rg * RPT) & 7) * 2;
        if constexpr (RPT == 8) {
          *(h8*)(sAh + pos) = h8{hh[0], hh[1], hh[2], hh[3], hh[4 % RPT], hh[5 % RPT], hh[6 % RPT], hh[7 % RPT]};
          *(h8*)(sAl + pos) = h8{ll[0], ll[1], ll[2], ll[3], ll[4 % RPT], ll[5 % RPT], ll[6 % RPT], ll[7 % RPT]};
        } else {
          *(h4*)(sAh + pos) = h4{hh[0], hh[1], hh[2], hh[3]};
          *(h4*)(sAl + pos) = h4{ll[0], ll[1], ll[2], ll[3]};
        }
        h2_block_scale_bits(mxb[q], sc, invb[q]);
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
          const float v = q == 0 ? rb[e].x : q == 1 ? rb[e].y : q == 2 ? rb[e].z : rb[e].w;
          h2_split1(v, sc, hh[e], ll[e]);
        }
        if constexpr (RPT == 8) {
          *(h8*)(sBh + pos) = h8{hh[0], hh[1], hh[2], hh[3], hh[4 % RPT], hh[5 % RPT], hh[6 % RPT], hh[7 % RPT]};
          *(h8*)(sBl + pos) = h8{ll[0], ll[1], ll[2], ll[3], ll[4 % RPT], ll[5 % RPT], ll[6 % RPT], ll[7 % RPT]};
        } else {
          *(h4*)(sBh + pos) = h4{hh[0], hh[1], hh[2], hh[3]};
          *(h4*)(sBl + pos) = h4{ll[0], ll[1], ll[2], ll[3]};
        }
      }
      if (rg == 0) {
        *(float4*)(sinv + 4 * cg) = make_float4(inva[0], inva[1], inva[2], inva[3]);
        *(float4*)(sinv + BT + 4 * cg) = make_float4(invb[0], invb[1], invb[2], invb[3]);
      }
      // the OTHER maxima buffer is free (read last in the previous slab): clear it for the next slab's ds_max
      unsigned* const other = smax + ((s + 1) & 1) * 2 * BT;
      if (tid < 2 * BT) other[tid] = 0u;
    }
    __syncthreads();
    if (s + NSET < nloc) load_slab(ra, rb, s + NSET);         // lands under the MFMAs below (and, with two sets, the next slab's phases)
    // ---- the slab's products in a scratch accumulator, folded by the two scales ------------------------------------------------------
    {
      f32x16 tmp[T][T];
#pragma unroll
      for (int kk = 0; kk < KB / 16; ++kk) {
        h8 ah[T], al[T], bh[T], bl[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
          const int row = wi0 + i * 32 + frow;
          const int off = row * (KB * 2) + (((2 * kk + khalf) ^ ((row >> 1) & 7)) << 4);
          ah[i] = *(const h8*)(sAh + off);
          al[i] = *(const h8*)(sAl + off);
        }
#pragma unroll
        for (int j = 0; j < T; ++j) {
          const int row = wj0 + j * 32 + frow;
          const int off = row * (KB * 2) + (((2 * kk + khalf) ^ ((row >> 1) & 7)) << 4);
          bh[j] = *(const h8*)(sBh + off);
          bl[j] = *(const h8*)(sBl + off);
        }
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
          for (int j = 0; j < T; ++j) {
            if (kk == 0) {
              f32x16 z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], z, 0, 0, 0);
            } else {
              tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], tmp[i][j], 0, 0, 0);
            }
          }
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
          for (int j = 0; j < T; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], tmp[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
          for (int j = 0; j < T; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], tmp[i][j], 0, 0, 0);
      }
      // accumulator register r of lane l: dY channel (row) 8 (r >> 2) + 4 (l >> 5) + (r & 3), X channel (column) l & 31
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const float ib = sinv[BT + wj0 + j * 32 + frow];
#pragma unroll
        for (int i = 0; i < T; ++i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 ia = *(const float4*)(sinv + wi0 + i * 32 + g * 8 + khalf * 4);
            tot[i][j][4 * g + 0] = fmaf(tmp[i][j][4 * g + 0], ia.x * ib, tot[i][j][4 * g + 0]);
            tot[i][j][4 * g + 1] = fmaf(tmp[i][j][4 * g + 1], ia.y * ib, tot[i][j][4 * g + 1]);
            tot[i][j][4 * g + 2] = fmaf(tmp[i][j][4 * g + 2], ia.z * ib, tot[i][j][4 * g + 2]);
            tot[i][j][4 * g + 3] = fmaf(tmp[i][j][4 * g + 3], ia.w * ib, tot[i][j][4 * g + 3]);
          }
        }
      }
    }
    // no barrier here: the next slab touches only the OTHER maxima buffer before its first barrier, and writes planes / scales after it
  };
  for (int s = 0; s < nloc; s += NSET) {
#pragma unroll
    for (int u = 0; u < NSET; ++u)
      if (s + u < nloc) slab_body(rav[u], rbv[u], s + u);
  }

  const int orow0 = co0 + wi0 + khalf * 4, ocol0 = kf0 + wj0 + frow;
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[(size_t)(orow0 + i * 32 + (r >> 2) * 8 + (r & 3)) * p.Kf + ocol0 + j * 32] = tot[i][j][r];
}

__global__ void k_wgrad_h2_finish(const float4* __restrict__ part, int S, long long n4, float4* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = part[i];
    for (int s = 1; s < S; ++s) {
      const float4 u = part[(size_t)s * n4 + i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    out[i] = v;
  }
}

static thread_local int t_wh2_bt = 0, t_wh2_wgs = 0;       // tuning overrides of the calling thread (0 = the plan below)
extern "C" void frcnn_conv2d_wgrad_h2_set_plan(int tile, int min_workgroups) { t_wh2_bt = tile; t_wh2_wgs = min_workgroups; }

// 128-wide tiles from 32 of them on (measured on the ResNet-152 shapes, profiles/r03_ae_wgrad_bench.txt: the RoI tail's 1x1 layers
// 186 vs 266 us, block3's 3x3 43 vs 46 us; block3's 1x1 layers with 16 such tiles 31 vs 23 us), else 64; slices so that the launch has
// >= 256 workgroups (one per CU) while a slice keeps >= 2 slabs of 64 pixels.  More slices fill the chip better in isolation but write
// and re-read more partial tiles than the operands themselves, and the sweep runs these launches BESIDE the data-gradient chain:
// the ResNet-152 step takes 19.65 ms at 128 / 256, 20.4 at 512, 20.75 at 1024, 22.8 at 64, 27.2 with one slice
// (profiles/r03_an_wgrad_slices.txt)
static void wgrad_h2_plan(int M, int Cout, int Kf, int Cin, int& BT, int& S, int& chunk) {
  const int nslabs = cdiv(M, 64);
  BT = 64;
  const bool ok128 = Cout % 128 == 0 && Cin % 128 == 0;
  if (ok128 && (Cout / 128) * (Kf / 128) >= 32) BT = 128;
  if (t_wh2_bt == 64 || (t_wh2_bt == 128 && ok128)) BT = t_wh2_bt;
  const int tiles = (Cout / BT) * (Kf / BT);
  int want = cdiv(t_wh2_wgs > 0 ? t_wh2_wgs : 256, tiles);
  want = max(1, min(want, nslabs / 2));
  chunk = cdiv(nslabs, max(want, 1));
  S = cdiv(nslabs, chunk);
}

extern "C" size_t frcnn_conv2d_wgrad_h2_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW) {
  if (Cin % 64 || Cout % 64 || Cin <= 0 || Cout <= 0) return 0;
  const long long M = (long long)N * OH * OW;
  if (M <= 0 || M >= (1ll << 30)) return 0;
  int BT, S, chunk;
  wgrad_h2_plan((int)M, Cout, KH * KW * Cin, Cin, BT, S, chunk);
  return S > 1 ? (size_t)S * (size_t)Cout * (size_t)(KH * KW * Cin) * sizeof(float) : 0;
}

extern "C" int frcnn_conv2d_wgrad_h2(const float* gy_d, const float* x_d, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH,
                                     int KW, int stride, int pad_top, int pad_left, float* dw_d, void* ws, size_t ws_bytes, void* stream) {
  if (!gy_d || !x_d || !dw_d || N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return FRCNN_E_ARG;
  if (Cin % 64 || Cout % 64 || Cin <= 0 || Cout <= 0) return FRCNN_E_ARG;
  const long long M = (long long)N * OH * OW, Kf = (long long)KH * KW * Cin;
  if (M >= (1ll << 30) || (long long)N * H * W * Cin >= (1ll << 31) || M * Cout >= (1ll << 31) || Kf * Cout >= (1ll << 31)) return FRCNN_E_ARG;
  int BT, S, chunk;
  wgrad_h2_plan((int)M, Cout, (int)Kf, Cin, BT, S, chunk);
  if (S > 1 && (!ws || ws_bytes < (size_t)S * Cout * Kf * sizeof(float))) return FRCNN_E_WS;
  WgradH2Params p;
  p.gy = gy_d; p.x = x_d; p.out = S > 1 ? (float*)ws : dw_d;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad_top = pad_top; p.pad_left = pad_left;
  p.M = (int)M; p.Kf = (int)Kf; p.ntn = (int)Kf / BT;
  p.nslabs = cdiv((int)M, 64); p.chunk = chunk;
  p.direct = (KH == 1 && KW == 1 && stride == 1 && pad_top == 0 && pad_left == 0 && OH == H && OW == W) ? 1 : 0;
  p.gz = (long long)Cout * Kf;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((Cout / BT) * (Kf / BT)), 1, (unsigned)S);
  const size_t lds = (size_t)4 * BT * 64 * 2 + (size_t)6 * BT * 4;
  if (BT == 128) {
    static KernelOnce once;
    HIP_TRY(kernel_once(once, (const void*)k_wgrad_h2<128>, 256, lds));
    hipLaunchKernelGGL((k_wgrad_h2<128>), grid, dim3(256), lds, st, p);
  } else {
    hipLaunchKernelGGL((k_wgrad_h2<64>), grid, dim3(256), lds, st, p);
  }
  LAUNCH_CHECK();
  if (S > 1) {
    const long long n4 = (long long)Cout * Kf / 4;
    hipLaunchKernelGGL(k_wgrad_h2_finish, dim3((unsigned)min((long long)2048, (n4 + 255) / 256)), dim3(256), 0, st, (const float4*)ws, S, n4,
                       (float4*)dw_d);
    LAUNCH_CHECK();
  }
  return FRCNN_OK;
}
