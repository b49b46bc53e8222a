// Filter gradient of an NHWC convolution straight from the tensors the reverse sweep already holds (gfx950, f32 matrix pipe):
//
//   dW[n, (kh, kw, c)] = sum_m  dY[m, n] * X[pix(m) + (kh, kw), c]          m = output pixel (img, oh, ow), n = output channel
//
// i.e. C = A^T B with BOTH operands stored reduction-major: dY is [M][Cout], a tap of X is [M][Cin] (a contiguous Cin run per pixel).
// v_mfma_f32_32x32x2_f32 takes, per lane, ONE element A[i = lane % 32][k = lane / 32] -- with the slab in LDS as [k][i] (exactly how
// it lies in HBM) a fragment is one conflict-free ds_read_b32, 32 consecutive dwords per k row.  So no operand is transposed and no
// im2col matrix is built: round 2 ran this product as the forward NT kernel on transpose_pad(dY) and transpose_pad / im2col_t(X),
// 3 ms of re-layout kernels per ResNet-152 step and twice the HBM traffic of the product itself.
//
// Work split: grid.x = (Cout / BT) * (Kf / BT) output tiles (a BT-wide column tile lies inside one tap: Cin % BT == 0), grid.z = S slices
// of the pixel range (single-image maps have 16 ... 600 tiles; S makes it >= 2 per CU); each slice writes its raw partial tile,
// k_wgrad_finish adds the S partials in a fixed order (deterministic, no atomics).
//
// Pipeline: 32-pixel slabs (BT*4-byte rows of dY and of the X tap) travel HBM -> LDS with direct-to-LDS dwordx4 loads into an NS-deep
// ring, counted vmcnt + one barrier per slab, as in conv_igemm.hip.  Rows past M and padding taps read a zero page.  LDS column index
// is XOR-ed with 32 on odd k rows (applied to the global source chunk and to the fragment column) so that the two k rows of one
// fragment read sit in different bank halves.
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float4 g_wgrad_zero[32];     // 512 zero bytes: the source of padding taps and of rows past M

struct WgradParams {
  const float* gy; const float* x; float* out;      // out: [S][Cout][Kf] partials (or the gradient itself when S == 1)
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad_top, pad_left;
  int M, Kf, ntn;                                   // ntn = Kf / BT column tiles
  int nslabs, chunk;                                // ceil(M / 32); slabs per slice
  int direct;                                       // 1x1, stride 1, no padding: X row of pixel m is x + m * Cin
  long long gz;                                     // Cout * Kf
};

#define LDS_AS __attribute__((address_space(3)))

template <int N>
__device__ __forceinline__ void wg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 64 lanes x 16 B, per-lane global source -> LDS [lds_base, +1 KiB) lane-linear (see conv_igemm.hip glds16 for why this is asm)
__device__ __forceinline__ void wg_glds16(const float* gsrc, unsigned lds_base) {
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

template <int BT, int NS>
__global__ __launch_bounds__(256) void k_wgrad_tn(const WgradParams p) {
  constexpr int T = BT / 64;                 // 32x32 MFMA tiles per wave and direction (waves 2 x 2)
  constexpr int RPI = 256 / BT;              // slab rows one direct-to-LDS instruction covers (1 KiB / row bytes)
  constexpr int CPR = BT / 4;                // 16-byte chunks per row
  constexpr int LA = 32 / RPI / 4;           // load instructions per wave and operand per slab
  constexpr int G = 2 * LA;
  constexpr int P = NS - 1;
  constexpr int SLAB = 2 * 32 * BT;          // floats: dY part [32][BT], then X part [32][BT]
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tn = blockIdx.x % p.ntn, tm = blockIdx.x / p.ntn;       // consecutive workgroups share the dY tile columns' pixel rows
  const int co0 = tm * BT, kf0 = tn * BT;
  const int tap = kf0 / p.Cin, c0 = kf0 - tap * p.Cin;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int s_begin = (int)blockIdx.z * p.chunk;
  const int nloc = min(p.nslabs - s_begin, p.chunk);
  float* const out = p.out + (size_t)blockIdx.z * p.gz;

  const float* zero = (const float*)g_wgrad_zero;
  // rows this lane stages: slab row r = (wave * LA + t) * RPI + lane / CPR, source chunk swizzled by the row's parity
  int a_row[LA], q4[LA];
#pragma unroll
  for (int t = 0; t < LA; ++t) {
    a_row[t] = (wave * LA + t) * RPI + lane / CPR;
    q4[t] = ((lane % CPR) ^ ((a_row[t] & 1) << 3)) * 4;
  }
  int m_next = s_begin * 32;                                          // first pixel of the NEXT slab to issue

  const unsigned lds0 = (unsigned)(size_t)(LDS_AS float*)smem;
  auto issue_slab = [&](int buf) {
    const unsigned sb = lds0 + (unsigned)(buf * SLAB * 4);
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int m = m_next + a_row[t];
      const float* src = m < p.M ? p.gy + (size_t)m * p.Cout + co0 + q4[t] : zero;
      wg_glds16(src, __builtin_amdgcn_readfirstlane(sb + (wave * LA + t) * 1024));
    }
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int m = m_next + a_row[t];
      const float* src = zero;
      if (m < p.M) {
        if (p.direct) src = p.x + (size_t)m * p.Cin + c0 + q4[t];
        else {
          const int ohow = p.OH * p.OW;
          const int img = m / ohow, rem = m - img * ohow;
          const int oh = rem / p.OW, ow = rem - oh * p.OW;
          const int ih = oh * p.stride - p.pad_top + kh, iw = ow * p.stride - p.pad_left + kw;
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            src = p.x + ((size_t)(img * p.H + ih) * p.W + iw) * p.Cin + c0 + q4[t];
        }
      }
      wg_glds16(src, __builtin_amdgcn_readfirstlane(sb + 32 * BT * 4 + (wave * LA + t) * 1024));
    }
    m_next += 32;
  };

  constexpr bool KSPLIT = (T == 1);          // single-tile waves: even / odd k pairs on two accumulators
  f32x16 acc[T][T], acc2[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

  // fragment: row k = 2 s + (lane >> 5) of the slab, column (tile column + lane & 31) ^ (32 on odd k rows = for lanes 32..63)
  const int khalf = lane >> 5;
  int a_off[T], b_off[T];                                                            // + s * 2 * BT
#pragma unroll
  for (int i = 0; i < T; ++i) {
    a_off[i] = khalf * BT + (((wave >> 1) * (BT / 2) + i * 32 + (lane & 31)) ^ (khalf << 5));
    b_off[i] = 32 * BT + khalf * BT + (((wave & 1) * (BT / 2) + i * 32 + (lane & 31)) ^ (khalf << 5));
  }

#pragma unroll
  for (int s = 0; s < P; ++s)
    if (s < nloc) issue_slab(s);

  auto slab = [&](auto more_c, int step) {
    constexpr bool MORE = decltype(more_c)::value;
    if (MORE || step + P <= nloc) wg_wait_vmcnt<(P - 1) * G>();
    else wg_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    const float* sb = smem + (step % NS) * SLAB;
    if (MORE) issue_slab((step + P) % NS);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float a[T], b[T];
#pragma unroll
      for (int i = 0; i < T; ++i) a[i] = sb[a_off[i] + s * 2 * BT];
#pragma unroll
      for (int j = 0; j < T; ++j) b[j] = sb[b_off[j] + s * 2 * BT];
#pragma unroll
      for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) {
          if (KSPLIT && (s & 1)) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc2[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  {
    int step = 0;
    for (; step + P < nloc; ++step) slab(std::true_type{}, step);
    for (; step < nloc; ++step) slab(std::false_type{}, step);
  }

  // accumulator element r of lane l: row (r / 4) * 8 + (l >> 5) * 4 + (r % 4), column l & 31 of the 32x32 tile -> 128-byte row pieces
  const int orow0 = co0 + (wave >> 1) * (BT / 2) + khalf * 4, ocol0 = kf0 + (wave & 1) * (BT / 2) + (lane & 31);
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = KSPLIT ? acc[i][j][r] + acc2[i][j][r] : acc[i][j][r];
        out[(size_t)(orow0 + i * 32 + (r >> 2) * 8 + (r & 3)) * p.Kf + ocol0 + j * 32] = v;
      }
}

__global__ void k_wgrad_finish(const float4* __restrict__ part, int S, long long n4, float4* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = part[i];
    for (int s = 1; s < S; ++s) {
      const float4 u = part[(size_t)s * n4 + i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    out[i] = v;
  }
}

// tile edge and slice count of one problem: 128-wide tiles where they alone give every CU a workgroup, else 64; slices so that the
// launch has >= 512 workgroups while a slice keeps >= 4 slabs
static thread_local int t_wgrad_bt = 0, t_wgrad_wgs = 0;       // tuning overrides of the calling thread (0 = the plan below)
extern "C" void frcnn_conv2d_wgrad_set_plan(int tile, int min_workgroups) { t_wgrad_bt = tile; t_wgrad_wgs = min_workgroups; }

static void wgrad_plan(int M, int Cout, int Kf, int Cin, int& BT, int& S, int& chunk) {
  const int nslabs = cdiv(M, 32);
  BT = 64;
  const bool ok128 = Cout % 128 == 0 && Cin % 128 == 0;
  if (ok128 && (Cout / 128) * (Kf / 128) >= 128) BT = 128;
  if (t_wgrad_bt == 64 || (t_wgrad_bt == 128 && ok128)) BT = t_wgrad_bt;
  const int tiles = (Cout / BT) * (Kf / BT);
  int want = cdiv(t_wgrad_wgs > 0 ? t_wgrad_wgs : 512, tiles);
  want = max(1, min(want, nslabs / 4));
  chunk = cdiv(nslabs, max(want, 1));
  S = cdiv(nslabs, chunk);
}

extern "C" int frcnn_conv2d_wgrad_supported(int Cin, int Cout) { return (Cin % 64 == 0 && Cout % 64 == 0) ? 1 : 0; }

extern "C" size_t frcnn_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW) {
  if (!frcnn_conv2d_wgrad_supported(Cin, Cout)) return 0;
  const long long M = (long long)N * OH * OW;
  if (M <= 0 || M >= (1ll << 30)) return 0;
  int BT, S, chunk;
  wgrad_plan((int)M, Cout, KH * KW * Cin, Cin, BT, S, chunk);
  return S > 1 ? (size_t)S * (size_t)Cout * (size_t)(KH * KW * Cin) * sizeof(float) : 0;
}

extern "C" int frcnn_conv2d_wgrad(const float* gy_d, const float* x_d, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH,
                                  int KW, int stride, int pad_top, int pad_left, float* dw_d, void* ws, size_t ws_bytes, void* stream) {
  if (!gy_d || !x_d || !dw_d || N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return FRCNN_E_ARG;
  if (!frcnn_conv2d_wgrad_supported(Cin, Cout)) return FRCNN_E_ARG;
  const long long M = (long long)N * OH * OW, Kf = (long long)KH * KW * Cin;
  if (M >= (1ll << 30) || (long long)N * H * W * Cin >= (1ll << 31) || M * Cout >= (1ll << 31) || Kf * Cout >= (1ll << 31)) return FRCNN_E_ARG;
  int BT, S, chunk;
  wgrad_plan((int)M, Cout, (int)Kf, Cin, BT, S, chunk);
  if (S > 1 && (!ws || ws_bytes < (size_t)S * Cout * Kf * sizeof(float))) return FRCNN_E_WS;
  WgradParams p;
  p.gy = gy_d; p.x = x_d; p.out = S > 1 ? (float*)ws : dw_d;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = KW;
  p.stride = stride; p.pad_top = pad_top; p.pad_left = pad_left;
  p.M = (int)M; p.Kf = (int)Kf; p.ntn = (int)Kf / BT;
  p.nslabs = cdiv((int)M, 32); p.chunk = chunk;
  p.direct = (KH == 1 && KW == 1 && stride == 1 && pad_top == 0 && pad_left == 0 && OH == H && OW == W) ? 1 : 0;
  p.gz = (long long)Cout * Kf;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((Cout / BT) * (Kf / BT)), 1, (unsigned)S);
  if (BT == 128) {
    constexpr int NS = 3;
    const size_t lds = (size_t)NS * 2 * 32 * 128 * sizeof(float);
    static KernelOnce once;
    HIP_TRY(kernel_once(once, (const void*)k_wgrad_tn<128, NS>, 256, lds));
    hipLaunchKernelGGL((k_wgrad_tn<128, NS>), grid, dim3(256), lds, st, p);
  } else {
    constexpr int NS = 3;
    const size_t lds = (size_t)NS * 2 * 32 * 64 * sizeof(float);
    hipLaunchKernelGGL((k_wgrad_tn<64, NS>), grid, dim3(256), lds, st, p);
  }
  LAUNCH_CHECK();
  if (S > 1) {
    const long long n4 = (long long)Cout * Kf / 4;
    hipLaunchKernelGGL(k_wgrad_finish, dim3((unsigned)min((long long)2048, (n4 + 255) / 256)), dim3(256), 0, st, (const float4*)ws, S, n4,
                       (float4*)dw_d);
    LAUNCH_CHECK();
  }
  return FRCNN_OK;
}
