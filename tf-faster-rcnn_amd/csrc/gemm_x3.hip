// f32 "NT" GEMM on the bf16 matrix pipe with EXACTLY split operands ("x3"):  Y[m][n] = act(sum_k A[m][k] * W[n][k] + bias + res).
//
// Every f32 value is split into three bf16 pieces rounded to nearest, x = h + m + l exactly (|m| <= 2^-8 |x|, |l| <= 2^-17 |x|), and a
// product a*w is evaluated as the six leading cross terms  ah*wh + (ah*wm + am*wh) + (ah*wl + al*wh + am*wm)  on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation; the three dropped terms are <= 2^-24 |a*w|, the size of ONE f32 rounding of the product.
// The leading term and the five small terms go to separate accumulators.  Six bf16 MFMAs (8 passes, 16 k) replace eight f32 MFMAs
// (16 passes, 2 k each): 192 instead of 512 matrix-pipe cycles per 16 k.  NOT the f32 MFMA: results agree with k_conv_igemm to
// f32 rounding, not bit for bit -- cfg.HIP.MFMA_X3 (default on, TEST mode; bench.py --mfma x3 / f32), labelled wherever a number is reported.
// Domain: finite operands below the bf16 maximum (3.39e38).  x = +-inf or |x| >= 2^128 - 2^119 gives h = +-inf and x - h = NaN: where the
// f32 MFMA would return +-inf for an overflowing activation, this kernel returns NaN for the whole output row (both mean overflow).
//
// Compared with round 1's experiment (both operands staged through registers, split once, three planes in LDS -- LDS-read bound with
// single-tile waves, one slab of prefetch; removed in round 3):
//   * W is static: it is split ONCE on the device into three bf16 planes [3][N][K] (frcnn_gemm_x3_pack) and travels HBM/L2 -> LDS
//     by direct-to-LDS loads like an f32 slab -- no VALU work, no registers;
//   * A stays f32 in HBM and in LDS (direct-to-LDS slab ring of k_conv_igemm unchanged); each wave splits the fragments it has
//     just read (two ds_read_b128 = 8 consecutive k per lane, the operand layout of the 32x32x16 MFMA) in registers: 5.5 VALU
//     per element, 176 per slab for a 64x64 wave tile against 48 MFMAs x 32 cycles -- hidden under the matrix pipe;
//   * resident workgroups, (tile, slab) stream, register epilogue through range-checked buffer stores: k_gemm_stream's skeleton.
// LDS per stage: 128 A rows x 128 B + 3 planes x 128 W rows x 64 B = 40 KB; two stages = 80 KB -> 2 workgroups per CU.
// W plane rows are 64 B (4 chunks of 8 bf16); chunk c of row n sits at position c ^ ((n >> 2) & 3): a ds_read_b128 is served in the
// 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), four 64-byte rows share a 256-byte bank row, and the four rows of a group
// with equal n mod 4 differ in (n >> 2) & 3 -- conflict-free (round 2 used (n >> 1) & 3: two-way conflicts, SQ_LDS_BANK_CONFLICT =
// half of SQ_LDS_IDX_ACTIVE, profiles/r03_i_pmc_gemm_h2.txt).
#include "common.h"
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct GemmX3Params {
  const float* x; const unsigned short* wp; const float* bias; const float* res; float* y;
  int M, N, K, nsteps, mtiles, ntiles, batch, act;
  long long gx, gwp, gy;               // per batch entry: elements of x / bf16 elements of wp (= 3*N*K) / elements of y
};

#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void x3_glds16(const void* gsrc, unsigned lds_base) {       // see conv_igemm.hip: glds16
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}

// exact split of 8 consecutive k values (two float4) into three bf16x8 operands, every piece rounded to nearest-even
// (v_cvt_pk_bf16_f32): h = bf16(x), m = bf16(x - h), l = x - h - m.  Both subtractions are exact in f32 and l has at most 8
// significant bits left, so h + m + l == x; |m| <= 2^-8 |x|, |l| <= 2^-17 |x|, errors of either sign (truncation would bias every
// product towards zero and is 8x looser: tests/test_x3_math_cpu.py).
__device__ __forceinline__ void x3_split8(const float4 a, const float4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const __bf16 hq = (__bf16)v[q];
    const float r = v[q] - (float)hq;
    const __bf16 mq = (__bf16)r;
    const float s = r - (float)mq;
    h[q] = hq; m[q] = mq; l[q] = (__bf16)s;
  }
}

template <int BM, int BN, int WM, int WN, int ABL = 0, int TERMS = 6>   // ABL: compile-time ablations for measurements only (1: no operand split, 2: no slab
                                                                     // loads after the first); TERMS = 9: also am*wl, al*wm, al*wl -> every product exact
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) __attribute__((amdgpu_waves_per_eu(2))) void k_gemm_x3(const GemmX3Params p) {
  constexpr int NW = (BM / WM) * (BN / WN);
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = BM / 8 / NW;                 // A: 1 KiB = 8 rows x 128 B per direct-to-LDS instruction
  constexpr int LB = 3 * BN / 16 / NW;            // W planes: 1 KiB = 16 rows x 64 B
  constexpr int G = LA + LB;
  constexpr int A_BYTES = BM * 128, P_BYTES = BN * 64, STAGE = A_BYTES + 3 * P_BYTES;
  static_assert((BM / 8) % NW == 0 && (3 * BN / 16) % NW == 0, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][STAGE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = p.mtiles * p.ntiles, T = per * p.batch;
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3, W8 = gridDim.x >> 3;
  const int tq = T / 8, tr = T % 8, tn = tq + (xcd < tr ? 1 : 0);
  const int t_end = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + tn;
  int tile = __builtin_amdgcn_readfirstlane(t_end - tn + wx);
  if (tile >= t_end) return;

  const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
  // per-lane byte offsets of the slab sources
  int a_lane[LA], b_lane[LB];
#pragma unroll
  for (int t = 0; t < LB; ++t) {
    const int u = wave * LB + t;                       // instruction index: plane u / (BN/16), 16-row block u % (BN/16)
    const int plane = u / (BN / 16), row = (u % (BN / 16)) * 16 + (lane >> 2), pos = lane & 3;
    b_lane[t] = ((plane * p.N + row) * p.K) * 2 + ((pos ^ ((row >> 2) & 3)) * 16);
  }
  const float* ia = p.x; const char* ib = (const char*)p.wp;
  int i_bm0 = 0, i_bn0 = 0, i_g = 0;
  auto set_tile = [&](int tl) {
    const int g = tl / per, rem = tl - g * per;
    const int mt = rem / p.ntiles, nt = rem - mt * p.ntiles;
    i_bm0 = mt * BM; i_bn0 = nt * BN; i_g = g;
    ia = p.x + (size_t)g * p.gx + (size_t)i_bm0 * p.K;
    ib = (const char*)(p.wp + (size_t)g * p.gwp + (size_t)i_bn0 * p.K);
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int row = (wave * LA + t) * 8 + (lane >> 3);
      a_lane[t] = (min(row, p.M - 1 - i_bm0) * p.K + ((lane & 7) ^ ((row >> 1) & 7)) * 4) * 4;      // bytes; rows past M re-read row M-1
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
  auto issue_one = [&](int buf, int t) {
    const unsigned sb = lds0 + (unsigned)(buf * STAGE);
    if (t < LA) x3_glds16((const char*)ia + a_lane[t], __builtin_amdgcn_readfirstlane(sb + (wave * LA + t) * 1024));
    else x3_glds16(ib + b_lane[t - LA], __builtin_amdgcn_readfirstlane(sb + A_BYTES + (wave * LB + (t - LA)) * 1024));
  };
  auto advance_k = [&]() { ia += 32; ib += 64; };
  auto issue_slab = [&](int buf) {
#pragma unroll
    for (int t = 0; t < G; ++t) issue_one(buf, t);
    advance_k();
  };

  f32x16 acc[TM][TN], acs[TM][TN];                   // leading term / the five small cross terms
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acs[i][j][r] = 0.f; }
  };
  zero_acc();
  const int frow = lane & 31, khalf = lane >> 5;
  // fragment byte offsets inside a stage: A row (wm0 + i*32 + frow), 16-byte chunk c at c ^ ((row>>1)&7); W plane row, chunk c at c ^ ((row>>1)&3)
  const int a_sw = (frow >> 1) & 7, b_sw = (frow >> 2) & 3;
  const int a_row = (wm0 + frow) * 128, b_row = A_BYTES + (wn0 + frow) * 64;

  int c_bm0, c_bn0, c_g;
  const int lane_off = (4 * khalf * p.N + frow) * 4;
  auto sub_base = [&](int i, int j) { return (long long)(c_bm0 + wm0 + i * 32) * p.N + (long long)(c_bn0 + wn0 + j * 32); };
  auto row_soff = [&](int r) { return ((r & 3) + 8 * (r >> 2)) * p.N * 4; };
  auto rsrc_of = [&](const float* tensor, long long sbase) {
    const long long left = ((long long)p.M * p.N - sbase) * 4;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(tensor + sbase), 0, (int)max(0ll, min(left, 0x7fffffffll)), 0x00020000);
  };

  auto slab_body = [&](int cur) {
    const int nbuf = cur ^ 1;
    const char* sb = smem + cur * STAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                      // two groups of 16 k per 32-wide slab
      bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const char* q = sb + b_row + j * 32 * 64 + (((2 * t + khalf) ^ b_sw) * 16);
        bh[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(q));
        bm[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(q + P_BYTES));
        bl[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(q + 2 * P_BYTES));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* q = sb + a_row + i * 32 * 128;
        const float4 f0 = *(const float4*)(q + (((4 * t + 2 * khalf) ^ a_sw) * 16));
        const float4 f1 = *(const float4*)(q + (((4 * t + 2 * khalf + 1) ^ a_sw) * 16));
        if (ABL & 1) { ah[i] = __builtin_bit_cast(bf16x8, f0); am[i] = __builtin_bit_cast(bf16x8, f1); al[i] = ah[i]; }
        else x3_split8(f0, f1, ah[i], am[i], al[i]);
      }
      // the next slab's direct-to-LDS loads are spread over the two k groups
#pragma unroll
      for (int u = (t * G) / 2; u < ((t + 1) * G) / 2; ++u) if (!(ABL & 2)) issue_one(nbuf, u);
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
          if (TERMS == 9) {
            acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bl[j], acs[i][j], 0, 0, 0);
            acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bm[j], acs[i][j], 0, 0, 0);
            acs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bl[j], acs[i][j], 0, 0, 0);
          }
        }
    }
    advance_k();
  };

  const float act_lo = p.act == FRCNN_ACT_NONE ? -__builtin_inff() : 0.f;
  const float act_hi = p.act == FRCNN_ACT_RELU6 ? 6.f : __builtin_inff();
  auto finish = [&](auto res_c) {
    constexpr bool RES = decltype(res_c)::value;
    float* const py = p.y + (size_t)c_g * p.gy;
    int lo = lane_off;
    asm volatile("" : "+v"(lo));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float bv = p.bias ? p.bias[c_bn0 + wn0 + j * 32 + frow] : 0.f;
        const long long sbase = sub_base(i, j);
        float v[16];
        if (RES) {
          const auto rr = rsrc_of(p.res + (size_t)c_g * p.gy, sbase);
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo + row_soff(r), 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = (acc[i][j][r] + acs[i][j][r]) + bv;
          if (RES) t += v[r];
          v[r] = act_clamp(t, act_lo, act_hi);
        }
        const auto ry = rsrc_of(py, sbase);
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ry, lo + row_soff(r), 0, 0);
      }
  };

  set_tile(tile);
  c_bm0 = i_bm0; c_bn0 = i_bn0; c_g = i_g;
  issue_slab(0);
  int cur = 0, step = 0;
  for (;;) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool last = step + 1 == p.nsteps;
    int next = tile;
    if (last) {
      next = __builtin_amdgcn_readfirstlane(tile + W8);
      set_tile(next < t_end ? next : tile);
    }
    slab_body(cur);
    cur ^= 1;
    if (!last) { ++step; continue; }
    if (p.res) finish(std::true_type{}); else finish(std::false_type{});
    if (next >= t_end) break;
    zero_acc();
    step = 0; tile = next; c_bm0 = i_bm0; c_bn0 = i_bn0; c_g = i_g;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// W [G][N][K] f32 -> planes [G][3][N][K] bf16 (h, m, l rounded to nearest): once per filter
__global__ void k_x3_pack(const float* __restrict__ w, long long nk, int G, unsigned short* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk * G) return;
  const long long g = i / nk, e = i - g * nk;
  const float x = w[i];
  const __bf16 h = (__bf16)x;                      // round to nearest-even, like x3_split8
  const float r = x - (float)h;
  const __bf16 m = (__bf16)r;
  const __bf16 l = (__bf16)(r - (float)m);
  unsigned short* o = out + (size_t)g * 3 * nk + e;
  o[0] = __builtin_bit_cast(unsigned short, h); o[nk] = __builtin_bit_cast(unsigned short, m); o[2 * nk] = __builtin_bit_cast(unsigned short, l);
}

extern "C" size_t frcnn_gemm_x3_pack_bytes(int G, int N, int K) {
  if (G <= 0 || N <= 0 || K <= 0) return 0;
  return (size_t)G * 3 * (size_t)N * (size_t)K * sizeof(unsigned short);
}

extern "C" int frcnn_gemm_x3_pack(const float* w_d, int G, int N, int K, void* planes_d, void* stream) {
  if (!w_d || !planes_d || G <= 0 || N <= 0 || K <= 0) return FRCNN_E_ARG;
  const long long nk = (long long)N * K, tot = nk * G;
  hipLaunchKernelGGL(k_x3_pack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_d, nk, G, (unsigned short*)planes_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

template <int BM, int BN, int WM, int WN, int ABL = 0, int TERMS = 6>
static int launch_x3(const GemmX3Params& q, hipStream_t st) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr size_t lds = 2 * (size_t)(BM * 128 + 3 * BN * 64);
  auto kern = k_gemm_x3<BM, BN, WM, WN, ABL, TERMS>;
  static KernelOnce once;
  int slots = 0;                            // resident workgroups on the CURRENT device
  HIP_TRY(kernel_once(once, (const void*)kern, NT, lds, &slots));
  if (slots < 8 || q.N % BN) return FRCNN_E_UNSUPPORTED;
  GemmX3Params p = q;
  p.mtiles = cdiv(p.M, BM); p.ntiles = p.N / BN; p.nsteps = p.K / 32;
  const long long T = (long long)p.mtiles * p.ntiles * p.batch;
  if (T >= (1ll << 30)) return FRCNN_E_UNSUPPORTED;
  const int grid = (int)min((long long)(slots / 8) * 8, ((T + 7) / 8) * 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, p);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// y[g] = act(x[g] W[g]^T + bias + res[g]) for g < G; x [G][M][K] f32, planes = frcnn_gemm_x3_pack(W [G][N][K]); res / y [G][M][N].
// cfg: -1 = tile configuration by shape, else a configuration id (A/B runs); terms: 6 (cross terms below 2^-24 dropped) or 9 (every
// f32 product exact).  Both are per call: the library keeps no tuning state.
extern "C" int frcnn_gemm_x3(const float* x_d, const void* planes_d, const float* bias_d, const float* res_d, float* y_d, int G, int M,
                             int N, int K, int act, int cfg, int terms, void* stream) {
  if (!x_d || !planes_d || !y_d || G <= 0 || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2 || (terms != 6 && terms != 9)) return FRCNN_E_ARG;
  if (K % 32 || N % 64 || (long long)M * N >= (1ll << 31) || (long long)N * K >= (1ll << 28) || K >= (1 << 22))     // 32-bit per-lane byte offsets
    return FRCNN_E_UNSUPPORTED;
  GemmX3Params p;
  p.x = x_d; p.wp = (const unsigned short*)planes_d; p.bias = bias_d; p.res = res_d; p.y = y_d;
  p.M = M; p.N = N; p.K = K; p.batch = G; p.act = act;
  p.gx = (long long)M * K; p.gwp = 3ll * N * K; p.gy = (long long)M * N;
  p.nsteps = p.mtiles = p.ntiles = 0;
  hipStream_t st = (hipStream_t)stream;
  if (cfg >= 0 && cfg < 10 && (N % 128) && cfg != 6) cfg = -1;       // a forced A/B configuration that cannot tile this N: by shape
  if (cfg < 0)       // 64x64 wave tiles (fewest LDS reads and operand splits per MFMA) from 256 tiles up: alone, a 150..1000-tile launch is
                     // 5-10 % faster with 8 x (32x64) waves (profiles/r02_p_x3_sweep.txt), but in the power-limited pipeline the 64x64
                     // tiles win (profiles/r02_t_x3_config_in_pipeline.txt: 427 vs 424 images/s); N = 64 (block1): 128x64 tiles
    cfg = (N % 128) ? 6 : ((long long)cdiv(M, 128) * (N / 128) * G >= 256) ? 0 : 1;
  if (terms == 9) {
    switch (cfg) {
      case 0: return launch_x3<128, 128, 64, 64, 0, 9>(p, st);
      case 1: return launch_x3<128, 128, 32, 64, 0, 9>(p, st);
      case 6: return launch_x3<128, 64, 64, 32, 0, 9>(p, st);
      default: return FRCNN_E_ARG;
    }
  }
  switch (cfg) {
    case 0: return launch_x3<128, 128, 64, 64>(p, st);
    case 1: return launch_x3<128, 128, 32, 64>(p, st);
    case 2: return launch_x3<64, 128, 32, 64>(p, st);
    case 3: return launch_x3<128, 128, 32, 128>(p, st);       // 4 waves, each 32 rows x the whole tile width: every A row is split once
    case 4: return launch_x3<256, 128, 64, 64>(p, st);        // 8 waves, 112 KB: 1 workgroup / CU, W slab shared by twice the rows
    case 5: return launch_x3<128, 256, 64, 64>(p, st);        // 8 waves, 128 KB
    case 6: return launch_x3<128, 64, 64, 32>(p, st);         // N % 64 == 0: 4 waves of 64x32, 56 KB
#ifdef FRCNN_ABLATION                                          // measurement builds only (wrong results by construction): not in the shipped library
    case 10: return launch_x3<128, 128, 64, 64, 1>(p, st);
    case 11: return launch_x3<128, 128, 64, 64, 2>(p, st);
    case 12: return launch_x3<128, 128, 64, 64, 3>(p, st);
    case 13: return launch_x3<128, 128, 32, 64, 1>(p, st);
    case 14: return launch_x3<128, 128, 32, 64, 2>(p, st);
    case 15: return launch_x3<128, 128, 32, 64, 3>(p, st);
#endif
    default: return FRCNN_E_ARG;
  }
}
