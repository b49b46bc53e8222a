// Implicit-GEMM NHWC convolution on the CDNA4 f32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
//   Y[m, n] = act( sum_k A[m, k] * Wt[n, k] + bias[n] (+ residual) )
//   m = output pixel (img, oh, ow) flattened, n = output channel, k = (kh, kw, c) flattened.
//
// A is never materialised: a (kh,kw) tap of an NHWC tensor is a contiguous Cin run per pixel, so a
// BM x 32 activation slab is BM 128-byte rows.  Slabs (activation rows + filter rows) travel
// HBM/L2 -> LDS with direct-to-LDS loads (global_load_lds_dwordx4: no VGPR round trip, no
// ds_write pass) into an NS-deep ring; NS-1 slabs are in flight while the MFMAs of the current
// one run (counted s_waitcnt vmcnt + one raw s_barrier per slab).  Padding taps / tile tails read a
// 16-byte zero page instead of being predicated (direct-to-LDS loads cannot write zeros).
//
// LDS image: a slab row is 8 chunks of 16 B; chunk q of row r is stored at position
// q ^ ((r >> 1) & 7).  Direct-to-LDS loads write lane-linear (wave base + lane*16), so the swizzle is
// applied to the per-lane GLOBAL source chunk and again on the fragment read; the ds_read_b128
// fragment reads (lane&31 = row, lane>>5 = k half) then hit 16 distinct 16-byte slots per 16-lane
// group: conflict free without padding.
//
// Why f32 MFMA: BASELINE.json asks for 1e-4 on scores/boxes through a 100-layer backbone; the
// f32-input MFMA is an exact-product f32 fmaf chain and runs at the 157.3 TFLOP/s matrix peak.  One
// accumulator per wave already saturates a SIMD's matrix pipe (64-cycle issue == dependent latency).
//
// Epilogue: accumulators -> LDS tile -> float4 rows: bias + residual (`subsample` stride) + ReLU
// fused, 16-byte coalesced global stores.
//
// Covers the slim call sites of lib/nets/network.py:323-378 (RPN 3x3/1x1, fc heads as 1x1),
// lib/nets/resnet_v1.py:80-125 (7x7/2 stem via fold_w, bottleneck 1x1 / 3x3 / conv2d_same
// stride 2, projection and subsample shortcuts), vgg16.py:26-60.
#include "common.h"
#include <mutex>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float4 g_zero_page[4];     // 64 zero bytes (static device memory, zero-initialised)

struct ConvParams {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  const float* mask;               // training: y = mask[m][n] > 0 ? y : 0 after the activation (the ReLU gradient of the tensor y is the gradient OF)
  int N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad_top, pad_left, act;
  int RH, RW, res_stride;
  int M, Ktot, nsteps;             // M = N*OH*OW, Ktot = KH*KW*Cin (fold_w: KH*32)
  int mtiles, ntiles;
  int batch;                       // grid.y (1 for convolutions)
  long long gx, gw, gy;            // batched GEMM use (grid.y = batch index): element strides of x / w / y per batch
  int dbg;                         // unused by the kernels (kept for the tuning ABI); run-time switches in the hot loop cost MFMA issue slots
  int splits, kchunk;              // split-K: grid.z = splits, each covers kchunk slabs and writes raw partial sums to y + z*gz
  long long gz;
  int stagger, stagger_slots;      // > 0: workgroup in residency slot s of its CU (first dispatch round) starts s * stagger shader cycles late
};

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One direct-to-LDS load: 64 lanes x 16 B from per-lane global addresses to LDS [lds_base, +1 KiB),
// lane-linear.  Issued from inline asm on purpose: hipcc tracks builtin LDS-DMA conservatively and
// puts `s_waitcnt vmcnt(0)` in front of the first ds_read of every k-step (it cannot prove the ring
// slots disjoint), which would drain the NS-deep pipeline; asm loads are invisible to its scoreboard,
// so the counted waits below are the only ones.  M0 (LDS base) is saved/restored inside the statement.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_base) {
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

template <int BM, int BN, int WM, int WN, int NS, bool FOLDW, bool ILV = false, bool RF = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void k_conv_igemm(const ConvParams p) {
  constexpr int NW = (BM / WM) * (BN / WN);
  constexpr int NT = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = BM / 8 / NW, LB = BN / 8 / NW;       // direct-to-LDS wave-instructions per wave per slab
  constexpr int G = LA + LB;
  constexpr int P = NS - 1;                               // slabs in flight ahead of the MFMAs
  constexpr int SLAB = (BM + BN) * 32;                    // floats per slab
  constexpr int EPI_LD = BN;                              // epilogue tile row stride (floats): b32 writes are per
                                                          // 32-lane half, float4 reads are lane-linear -> no padding needed
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile/wave mismatch");
  static_assert((P - 1) * G <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [NS][BM+BN][32]; reused by the epilogue

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.stagger > 0) {
    // Phase-shift the co-resident workgroups of a CU.  Every workgroup of a launch starts at the same time and does the same
    // amount of work, so the (LDS-limited) workgroups sharing a CU reach their prologue / epilogue -- the phases without MFMA
    // issue -- together and the matrix pipe idles; an initial delay of slot * (tile time / slots) for the workgroups of the
    // first dispatch round (observed placement: block b -> XCD b % 8, CU (b / 8) % 32, slot (b / 8) / 32) makes one's
    // epilogue overlap the others' main loops for the rest of the launch.
    const unsigned slot = blockIdx.x >> 8;               // 256 CUs: blocks [256 s, 256 s + 256) fill residency slot s
    if (blockIdx.y == 0 && blockIdx.z == 0 && slot > 0 && slot < (unsigned)p.stagger_slots) {
      const long long t0 = clock64(), wait = (long long)p.stagger * slot;
      while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
  }
  // XCD-aware tile order (observed placement: block b -> XCD b % 8): consecutive tiles of one XCD
  // share an activation slab; bijective for any grid size.
  const int nwg = p.mtiles * p.ntiles;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = bid / p.ntiles, nt = bid % p.ntiles;
  const int bm0 = mt * BM, bn0 = nt * BN;
  const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
  const float* const px = p.x + (size_t)blockIdx.y * p.gx;        // batched GEMM: independent problems along grid.y
  const float* const pw = p.w + (size_t)blockIdx.y * p.gw;
  float* const py = p.y + (size_t)blockIdx.y * p.gy + (size_t)blockIdx.z * p.gz;
  // split-K (under-filled launches: single images, weight gradients): this workgroup reduces slabs [s_begin, s_begin + nloc)
  const int s_begin = (int)blockIdx.z * p.kchunk;
  const int nloc = p.splits > 1 ? min(p.nsteps - s_begin, p.kchunk) : p.nsteps;

  // ---- per-lane source descriptors of the rows this lane stages -------------------------------
  const int lrow = lane >> 3, lpos = lane & 7;
  int a_base[LA], a_ih0[LA], a_iw0[LA], a_q4[LA];
#pragma unroll
  for (int t = 0; t < LA; ++t) {
    const int row = (wave * LA + t) * 8 + lrow;                 // slab row in [0, BM)
    a_q4[t] = (lpos ^ ((row >> 1) & 7)) * 4;                    // swizzled source chunk (float offset)
    const int m = bm0 + row;
    if (m < p.M) {
      const int img = m / (p.OH * p.OW), rem = m % (p.OH * p.OW);
      const int oh = rem / p.OW, ow = rem % p.OW;
      a_base[t] = img * p.H * p.W * p.Cin;
      a_ih0[t] = oh * p.stride - p.pad_top;
      a_iw0[t] = ow * p.stride - p.pad_left;
    } else {
      a_base[t] = 0; a_ih0[t] = -(1 << 28); a_iw0[t] = -(1 << 28);
    }
  }
  // Per-lane running source pointers.  Within one (kh,kw) tap consecutive slabs are 128 B apart, so a
  // slab issue is one 64-bit add per load; pointers are rebuilt only at tap boundaries (wave-uniform
  // branch).  Padding taps / tile-tail rows point at the zero page with increment 0.
  const float* zero = (const float*)g_zero_page;
  const float* a_ptr[LA]; int a_inc[LA];
  const float* b_ptr[LB]; int b_inc[LB];
#pragma unroll
  for (int t = 0; t < LB; ++t) {
    const int row = BM + (wave * LB + t) * 8 + lrow;
    const int n = bn0 + row - BM;
    const int q4 = (lpos ^ ((row >> 1) & 7)) * 4;
    const bool ok = n < p.Cout;
    b_ptr[t] = ok ? pw + (size_t)n * p.Ktot + q4 + (size_t)s_begin * 32 : zero;
    b_inc[t] = ok ? 32 : 0;
  }
  int kh = 0, kw = 0, c0 = 0;                                   // k position of the NEXT slab to issue
  if (!FOLDW && s_begin) {
    const int k0 = s_begin * 32, tap = k0 / p.Cin;
    c0 = k0 - tap * p.Cin;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
  }

  auto set_tap = [&]() {
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int ih = a_ih0[t] + kh, iw = a_iw0[t] + kw;
      bool ok = (unsigned)ih < (unsigned)p.H;
      if (FOLDW) ok = ok && ((unsigned)(iw + (a_q4[t] >> 2)) < (unsigned)p.W);
      else ok = ok && ((unsigned)iw < (unsigned)p.W);
      a_ptr[t] = ok ? px + (size_t)(a_base[t] + (ih * p.W + iw) * p.Cin + c0 + a_q4[t]) : zero;   // c0 != 0 only at a split's start
      a_inc[t] = ok ? 32 : 0;
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(LDS_AS float*)smem;       // LDS byte address of the ring
  // one direct-to-LDS load (t-th of the G = LA + LB a wave owns) of the slab going into ring slot `buf`
  auto issue_one = [&](int buf, int t) {
    const unsigned sb = lds0 + (unsigned)(buf * SLAB * 4);
    if (t < LA) {
      glds16(a_ptr[t], __builtin_amdgcn_readfirstlane(sb + (wave * LA + t) * 1024));
      a_ptr[t] += a_inc[t];
    } else {
      const int u = t - LA;
      glds16(b_ptr[u], __builtin_amdgcn_readfirstlane(sb + BM * 128 + (wave * LB + u) * 1024));
      b_ptr[u] += b_inc[u];
    }
  };
  auto advance_k = [&]() {
    if (FOLDW) { ++kh; }
    else {
      c0 += 32;
      if (c0 == p.Cin) { c0 = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    }
  };
  auto issue_slab = [&](int buf) {
    if (c0 == 0) set_tap();
#pragma unroll
    for (int t = 0; t < G; ++t) issue_one(buf, t);
    advance_k();
  };

  constexpr bool KSPLIT = (TM * TN == 1);                 // second accumulator for odd k of single-tile waves
  f32x16 acc[TM][TN], acc2[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

  // fragment read offsets: row (lane&31) of the wave's 32-row groups, k chunk 2s + (lane>>5), swizzled
  const int frow = lane & 31, khalf = lane >> 5;
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = ((2 * s + khalf) ^ ((frow >> 1) & 7)) * 4;
  const int a_row0 = (wm0 + frow) * 32, b_row0 = (BM + wn0 + frow) * 32;

  // ---- prologue: P slabs in flight ---------------------------------------------------------------
  if (c0 != 0) set_tap();                                        // a split that starts inside a tap
#pragma unroll
  for (int s = 0; s < P; ++s)
    if (s < nloc) issue_slab(s);

  // One slab of the main loop.  MORE (compile time): a further slab is issued into the ring slot that just became free.  The
  // loop is peeled (MORE for steps [0, nloc-P), !MORE for the last P) so that the unrolled body has NO run-time branch: any
  // branch in here splits it into basic blocks and the compiler then serialises fragment reads and MFMAs (s_waitcnt lgkmcnt(0)
  // for all sixteen reads before the first MFMA) instead of pipelining them with counted waits.
  auto slab = [&](auto more_c, int step) {
    constexpr bool MORE = decltype(more_c)::value;
    // slab `step` must have landed: at most the P-1 younger slabs may still be in flight
    if (MORE || step + P <= nloc) wait_vmcnt<(P - 1) * G>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();          // every wave finished slab step-1 and sees slab step
    const int nbuf = (step + P) % NS;
    const float* sb = smem + (step % NS) * SLAB;
    float4 a[4][TM], b[4][TN];
    auto read_frag = [&](int q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[q][i] = *(const float4*)(sb + a_row0 + i * 1024 + koff[q]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[q][j] = *(const float4*)(sb + b_row0 + j * 1024 + koff[q]);
    };
    // MFMA order: consecutive instructions always target DIFFERENT accumulators; single-tile waves split k over two
    // accumulators (KSPLIT)
    auto mfma_group = [&](int q) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float av = e == 0 ? a[q][i].x : e == 1 ? a[q][i].y : e == 2 ? a[q][i].z : a[q][i].w;
            const float bv = e == 0 ? b[q][j].x : e == 1 ? b[q][j].y : e == 2 ? b[q][j].z : b[q][j].w;
            if (KSPLIT && (e & 1)) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
    };
    if (ILV) {
      // the wave's G slab loads are spread between the four k-groups so that their issue slots hide behind MFMAs; group q+1's
      // fragment reads are written BEFORE group q's loads (the asm loads carry a memory clobber, reads cannot cross them)
      if (MORE && c0 == 0) set_tap();
      read_frag(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) read_frag(q + 1);
        if (MORE) {
#pragma unroll
          for (int t = (q * G) / 4; t < ((q + 1) * G) / 4; ++t) issue_one(nbuf, t);
        }
        mfma_group(q);
      }
      if (MORE) advance_k();
    } else {
      // RF ("reads first"): the fragment reads the first MFMAs need are issued BEFORE the next slab's direct-to-LDS loads,
      // so their LDS latency overlaps the ~100 instructions of load issue instead of following it
      constexpr int RF_FIRST = !RF ? 0 : (TM * TN == 1 ? 4 : 1);
#pragma unroll
      for (int q = 0; q < RF_FIRST; ++q) read_frag(q);
      if (MORE) issue_slab(nbuf);
#pragma unroll
      for (int q = RF_FIRST; q < 4; ++q) read_frag(q);
      // single-tile waves have only 4 MFMAs (256 cycles) per k-group -- too few to cover the next group's fragment reads, so
      // keep all eight reads ahead of the MFMAs; multi-tile waves are left to the scheduler, which pipelines group q+1's reads
      // under group q's 16 MFMAs
      if (TM * TN == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) mfma_group(q);
    }
  };
  {
    int step = 0;
    for (; step + P < nloc; ++step) slab(std::true_type{}, step);
    for (; step < nloc; ++step) slab(std::false_type{}, step);
  }
  if (KSPLIT) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += acc2[i][j][r];
  }

  // ---- epilogue: accumulators -> LDS tile [BM][BN] -> float4 rows ---------------------------------
  __syncthreads();                         // all slab reads done; the ring is free
  // D[i][j] of a 32x32 MFMA tile: lane holds column j = lane&31, register r -> row (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        smem[ml * EPI_LD + wn0 + j * 32 + frow] = acc[i][j][r];
      }
  __syncthreads();
  const int ohow = p.OH * p.OW;
  if ((p.Cout & 3) == 0) {
    constexpr int C4 = BN / 4;
    for (int t = tid; t < BM * C4; t += NT) {
      const int ml = t / C4, nl = (t % C4) * 4;
      const int m = bm0 + ml, n = bn0 + nl;
      if (m >= p.M || n >= p.Cout) continue;
      float4 v = *(const float4*)(smem + ml * EPI_LD + nl);
      if (p.bias) {
        const float4 bv = *(const float4*)(p.bias + n);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      }
      if (p.res) {
        size_t ro;
        if (p.res_stride == 1) ro = (size_t)m * p.Cout + n;
        else {
          const int img = m / ohow, rem = m % ohow, oh = rem / p.OW, ow = rem % p.OW;
          ro = ((size_t)(img * p.RH + oh * p.res_stride) * p.RW + ow * p.res_stride) * p.Cout + n;
        }
        const float4 rv = *(const float4*)(p.res + ro);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (p.act == FRCNN_ACT_RELU) v = act_relu(v);
      else if (p.act == FRCNN_ACT_RELU6)
        v = act_relu6(v);
      if (p.mask) {
        const float4 k = *(const float4*)(p.mask + (size_t)m * p.Cout + n);
        v.x = k.x > 0.f ? v.x : 0.f; v.y = k.y > 0.f ? v.y : 0.f; v.z = k.z > 0.f ? v.z : 0.f; v.w = k.w > 0.f ? v.w : 0.f;
      }
      *(float4*)(py + (size_t)m * p.Cout + n) = v;
    }
  } else {          // Cout not a multiple of 4 (RPN / fc heads): scalar path
    for (int t = tid; t < BM * BN; t += NT) {
      const int ml = t / BN, nl = t % BN;
      const int m = bm0 + ml, n = bn0 + nl;
      if (m >= p.M || n >= p.Cout) continue;
      float v = smem[ml * EPI_LD + nl] + (p.bias ? p.bias[n] : 0.f);
      if (p.res) {
        size_t ro;
        if (p.res_stride == 1) ro = (size_t)m * p.Cout + n;
        else {
          const int img = m / ohow, rem = m % ohow, oh = rem / p.OW, ow = rem % p.OW;
          ro = ((size_t)(img * p.RH + oh * p.res_stride) * p.RW + ow * p.res_stride) * p.Cout + n;
        }
        v += p.res[ro];
      }
      if (p.act == FRCNN_ACT_RELU) v = act_relu(v);
      else if (p.act == FRCNN_ACT_RELU6) v = act_relu6(v);
      if (p.mask) v = p.mask[(size_t)m * p.Cout + n] > 0.f ? v : 0.f;
      py[(size_t)m * p.Cout + n] = v;
    }
  }
}

template <int BM, int BN, int WM, int WN, int NS, bool FOLDW, bool ILV = false, bool RF = false>
static int launch_conv(ConvParams p, hipStream_t st) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr size_t ring = sizeof(float) * NS * (BM + BN) * 32, epi = sizeof(float) * BM * BN;
  constexpr size_t lds = ring > epi ? ring : epi;     // the epilogue tile reuses the ring memory
  auto kern = k_conv_igemm<BM, BN, WM, WN, NS, FOLDW, ILV, RF>;
  // once per instantiation, safe when several host threads drive distinct streams ("distinct streams are thread-safe")
  static KernelOnce once;
  HIP_TRY(kernel_once(once, (const void*)kern, NT, lds));
  p.mtiles = cdiv(p.M, BM);
  p.ntiles = cdiv(p.Cout, BN);
  hipLaunchKernelGGL(kern, dim3(p.mtiles * p.ntiles, p.batch, p.splits), dim3(NT), lds, st, p);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ---- streaming (persistent) GEMM ---------------------------------------------------------------------------------------
// For the 1x1 / stride-1 convolutions and the batched Winograd products (plain "NT" GEMMs: A[M][K] activations, W[N][K] filter,
// Y[M][N]).  Same slab ring, fragment layout and MFMA order as k_conv_igemm, but a workgroup is a RESIDENT worker that walks a
// static list of output tiles and treats (tile, k-slab) as ONE stream:
//   * the ring never drains: slab 0 of the next tile is issued while the last slab of the current tile is being multiplied, so the
//     load-latency prologue is paid once per workgroup instead of once per tile (a block3 tile is only 8 slabs = 128 MFMAs/wave);
//   * the epilogue writes straight from the accumulator registers (lane = output column: each store instruction covers two
//     128-byte row segments), so it needs neither the LDS tile nor its two barriers, and the waves of a workgroup finish a tile
//     independently while the next tile's slab is already landing;
//   * with RESPF the residual rows of the tile are requested before the MFMAs of its last slab, not after them;
//   * no workgroup launch / LDS allocation per tile.
// The host requires N % BN == 0; the M tail is handled without predicates (clamped source rows, range-checked buffer stores).
// Schedule: workgroup b lives on XCD b % 8 (observed placement); XCD x owns a contiguous range of the m-major tile list and its
// workgroups take that range round-robin, so concurrently running workgroups of an XCD share activation rows in its L2.
// The per-element summation order is the one of k_conv_igemm (results are bit-identical to it).
struct GemmParams {
  const float* x; const float* w; const float* bias; const float* res; float* y;
  int M, N, K, nsteps, mtiles, ntiles, batch, act;
  long long gx, gw, gy;                                 // element strides per batch entry
};

template <int BM, int BN, int WM, int WN, bool ILV, bool RF, bool RESPF>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void k_gemm_stream(const GemmParams p) {
  constexpr int NW = (BM / WM) * (BN / WN);
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = BM / 8 / NW, LB = BN / 8 / NW;
  constexpr int G = LA + LB;
  constexpr int SLAB = (BM + BN) * 32;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(16))) float smem[];     // [2][BM+BN][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = p.mtiles * p.ntiles, T = per * p.batch;
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3, W8 = gridDim.x >> 3;       // host: gridDim.x % 8 == 0
  const int tq = T / 8, tr = T % 8, tn = tq + (xcd < tr ? 1 : 0);
  const int t_end = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + tn;
  int tile = __builtin_amdgcn_readfirstlane(t_end - tn + wx);
  if (tile >= t_end) return;                                                       // uniform per workgroup

  const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
  const int lrow = lane >> 3, lpos = lane & 7;
  // per-lane slab sources: row (wave*L + t)*8 + lrow of the tile, swizzled 16-byte chunk; the tile / slab position is added on top
  int a_lane[LA], b_lane[LB];
#pragma unroll
  for (int t = 0; t < LB; ++t) {
    const int row = (wave * LB + t) * 8 + lrow;
    b_lane[t] = row * p.K + (lpos ^ (((BM + row) >> 1) & 7)) * 4;
  }
  const float* ia = p.x; const float* ib = p.w;      // issue side (wave-uniform): next slab of the tile whose slabs are being issued
  int i_bm0 = 0, i_bn0 = 0, i_g = 0;
  auto set_tile = [&](int tl) {
    const int g = tl / per, rem = tl - g * per;
    const int mt = rem / p.ntiles, nt = rem - mt * p.ntiles;
    i_bm0 = mt * BM; i_bn0 = nt * BN; i_g = g;
    ia = p.x + (size_t)g * p.gx + (size_t)i_bm0 * p.K;
    ib = p.w + (size_t)g * p.gw + (size_t)i_bn0 * p.K;
    // rows past M (last m-tile only) re-read row M-1: direct-to-LDS loads cannot be predicated; their products are never stored
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int row = (wave * LA + t) * 8 + lrow;
      a_lane[t] = (min(row, p.M - 1 - i_bm0)) * p.K + (lpos ^ ((row >> 1) & 7)) * 4;
    }
  };
  const unsigned lds0 = (unsigned)(size_t)(LDS_AS float*)smem;
  auto issue_one = [&](int buf, int t) {
    const unsigned sb = lds0 + (unsigned)(buf * SLAB * 4);
    if (t < LA) glds16(ia + a_lane[t], __builtin_amdgcn_readfirstlane(sb + (wave * LA + t) * 1024));
    else glds16(ib + b_lane[t - LA], __builtin_amdgcn_readfirstlane(sb + BM * 128 + (wave * LB + (t - LA)) * 1024));
  };
  auto advance_k = [&]() { ia += 32; ib += 32; };
  auto issue_slab = [&](int buf) {
#pragma unroll
    for (int t = 0; t < G; ++t) issue_one(buf, t);
    advance_k();
  };

  constexpr bool KSPLIT = (TM * TN == 1);
  f32x16 acc[TM][TN], acc2[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }
  };
  zero_acc();
  const int frow = lane & 31, khalf = lane >> 5;
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = ((2 * s + khalf) ^ ((frow >> 1) & 7)) * 4;
  const int a_row0 = (wm0 + frow) * 32, b_row0 = (BM + wn0 + frow) * 32;

  // compute side: the tile whose slabs are being multiplied (the issue side runs one slab ahead and crosses into the next tile)
  int c_bm0, c_bn0, c_g;
  float rv[RESPF ? TM : 1][RESPF ? TN : 1][16];
  // Output / residual element of accumulator register r of sub-tile (i, j): row (r&3) + 8*(r>>2) + 4*khalf, column frow.  Addressed
  // with raw buffer instructions: a wave-uniform descriptor per sub-tile whose range ends at the end of the tensor -- rows past M
  // are dropped (stores) / read as zero (loads) by the hardware range check, so the M tail needs no predicate -- and a per-lane
  // byte offset = (tile-invariant lane part) + (scalar row part).
  const int lane_off = (4 * khalf * p.N + frow) * 4;
  auto sub_base = [&](int i, int j) { return (long long)(c_bm0 + wm0 + i * 32) * p.N + (long long)(c_bn0 + wn0 + j * 32); };
  auto row_soff = [&](int r) { return ((r & 3) + 8 * (r >> 2)) * p.N * 4; };
  auto rsrc_of = [&](const float* tensor, long long sbase) {
    const long long left = ((long long)p.M * p.N - sbase) * 4;                       // bytes from the sub-tile origin to the tensor end
    return __builtin_amdgcn_make_buffer_rsrc((void*)(tensor + sbase), 0, (int)max(0ll, min(left, 0x7fffffffll)), 0x00020000);
  };
  auto prefetch_res = [&]() {
    if (!RESPF || !p.res) return;
    int lo = lane_off;
    asm volatile("" : "+v"(lo));             // opaque per tile: keeps the 16 row offsets out of the registers that live through the main loop
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const auto rr = rsrc_of(p.res + (size_t)c_g * p.gy, sub_base(i, j));
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[RESPF ? i : 0][RESPF ? j : 0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo + row_soff(r), 0, 0));
      }
  };

  // The multiply part of one slab: fragment reads of slab `cur`, the direct-to-LDS loads of the stream's next slab, 64 * TM * TN
  // MFMAs.  ONE code instance for every slab of every tile (no branch inside: a branch would split the block and the compiler
  // would then serialise fragment reads and MFMAs).
  auto slab_body = [&](int cur) {
    const int nbuf = cur ^ 1;
    const float* sb = smem + cur * SLAB;
    float4 a[4][TM], b[4][TN];
    auto read_frag = [&](int q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[q][i] = *(const float4*)(sb + a_row0 + i * 1024 + koff[q]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[q][j] = *(const float4*)(sb + b_row0 + j * 1024 + koff[q]);
    };
    auto mfma_group = [&](int q) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float av = e == 0 ? a[q][i].x : e == 1 ? a[q][i].y : e == 2 ? a[q][i].z : a[q][i].w;
            const float bv = e == 0 ? b[q][j].x : e == 1 ? b[q][j].y : e == 2 ? b[q][j].z : b[q][j].w;
            if (KSPLIT && (e & 1)) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
    };
    if (ILV) {
      read_frag(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) read_frag(q + 1);
#pragma unroll
        for (int t = (q * G) / 4; t < ((q + 1) * G) / 4; ++t) issue_one(nbuf, t);
        mfma_group(q);
      }
      advance_k();
    } else {
      constexpr int RF_FIRST = !RF ? 0 : (TM * TN == 1 ? 4 : 1);
#pragma unroll
      for (int q = 0; q < RF_FIRST; ++q) read_frag(q);
      issue_slab(nbuf);
#pragma unroll
      for (int q = RF_FIRST; q < 4; ++q) read_frag(q);
      if (TM * TN == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) mfma_group(q);
    }
  };

  // epilogue from registers: D of a 32x32 MFMA tile -- lane holds column lane&31, register r holds row (r&3) + 8*(r>>2) + 4*(lane>>5).
  // No predicates; the activation is a clamp to [lo, hi] (ReLU: [0, inf), ReLU6: [0, 6]); the residual case is
  // a separate straight-line instantiation.
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
        if (RES && !RESPF) {
          const auto rr = rsrc_of(p.res + (size_t)c_g * p.gy, sbase);
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lo + row_soff(r), 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float t = (KSPLIT ? acc[i][j][r] + acc2[i][j][r] : acc[i][j][r]) + bv;
          if (RES) t += RESPF ? rv[RESPF ? i : 0][RESPF ? j : 0][r] : v[r];
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
    wait_vmcnt<0>();                         // slab `step` of the current tile has landed (this wave's share)
    __builtin_amdgcn_s_barrier();            // ... everybody's share; and every wave is done reading the other ring slot
    const bool last = step + 1 == p.nsteps;  // wave-uniform
    int next = tile;
    if (last) {
      // the stream crosses into the next tile: its slab 0 is issued under this slab's MFMAs (the final tile re-requests its own
      // slab 0 -- in-bounds, never read -- so that the multiply part stays one branch-free code instance)
      next = __builtin_amdgcn_readfirstlane(tile + W8);
      set_tile(next < t_end ? next : tile);
      prefetch_res();
    }
    slab_body(cur);
    cur ^= 1;
    if (!last) { ++step; continue; }
    if (p.res) finish(std::true_type{}); else finish(std::false_type{});
    if (next >= t_end) break;
    zero_acc();
    step = 0; tile = next; c_bm0 = i_bm0; c_bn0 = i_bn0; c_g = i_g;
  }
  wait_vmcnt<0>();                           // the dummy prefetch must not land in a later workgroup's LDS
}

template <int BM, int BN, int WM, int WN, bool ILV, bool RF, bool RESPF>
static int launch_stream(const ConvParams& c, hipStream_t st) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr size_t lds = sizeof(float) * 2 * (BM + BN) * 32;
  auto kern = k_gemm_stream<BM, BN, WM, WN, ILV, RF, RESPF>;
  static KernelOnce once;
  int slots = 0;                            // resident workgroups on the CURRENT device
  HIP_TRY(kernel_once(once, (const void*)kern, NT, lds, &slots));
  const bool plain = c.KH == 1 && c.KW == 1 && c.stride == 1 && c.pad_top == 0 && c.pad_left == 0 && c.OH == c.H && c.OW == c.W;
  if (slots < 8 || !plain || c.mask || c.Cout % BN || c.splits != 1 || (c.res && c.res_stride != 1) || (long long)c.M * c.Cout >= (1ll << 31) ||
      (long long)c.M * c.Cin >= (1ll << 31) || (long long)c.Cout * c.Cin >= (1ll << 31))
    return FRCNN_E_UNSUPPORTED;
  GemmParams p;
  p.x = c.x; p.w = c.w; p.bias = c.bias; p.res = c.res; p.y = c.y;
  p.mtiles = cdiv(c.M, BM); p.M = c.M; p.N = c.Cout; p.K = c.Cin; p.nsteps = c.Cin / 32;
  p.ntiles = c.Cout / BN; p.batch = c.batch; p.act = c.act;
  p.gx = c.gx; p.gw = c.gw; p.gy = c.gy;
  const long long T = (long long)p.mtiles * p.ntiles * p.batch;
  if (T >= (1ll << 30)) return FRCNN_E_UNSUPPORTED;
  const int grid = (int)min((long long)(slots / 8) * 8, ((T + 7) / 8) * 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, p);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// streaming configurations (ids 100+ of tuning key 0)
static int launch_stream_cfg(int id, const ConvParams& p, hipStream_t st) {
  switch (id) {
    case 100: return launch_stream<64, 64, 32, 32, true, false, false>(p, st);      // cfg 15
    case 101: return launch_stream<64, 64, 32, 32, true, false, true>(p, st);
    case 102: return launch_stream<128, 128, 64, 64, false, true, false>(p, st);    // cfg 20
    case 103: return launch_stream<128, 128, 64, 64, false, true, true>(p, st);
    case 104: return launch_stream<128, 128, 32, 64, false, true, false>(p, st);    // cfg 21
    case 105: return launch_stream<128, 128, 32, 64, false, true, true>(p, st);
    case 106: return launch_stream<128, 64, 32, 32, true, false, false>(p, st);     // 8 single-tile waves, 48 KB
    case 107: return launch_stream<128, 64, 32, 32, true, false, true>(p, st);
    case 108: return launch_stream<64, 128, 32, 32, true, false, false>(p, st);
    case 109: return launch_stream<64, 128, 32, 32, true, false, true>(p, st);
    case 110: return launch_stream<128, 64, 64, 32, false, true, true>(p, st);      // 4 waves x 2 accumulators
    case 111: return launch_stream<64, 128, 32, 64, false, true, true>(p, st);
    case 112: return launch_stream<32, 64, 32, 32, true, false, false>(p, st);       // 2 single-tile waves, 24 KB
    case 113: return launch_stream<64, 32, 32, 32, true, false, false>(p, st);
    case 114: return launch_stream<32, 128, 32, 32, true, false, false>(p, st);      // 4 single-tile waves, 40 KB
    case 115: return launch_stream<32, 64, 32, 32, false, true, false>(p, st);
    default: return FRCNN_E_ARG;
  }
}

// tuning knob (experiments / A-B runs): key 0 = force a tile configuration id for every non-stem conv
// (-1 = automatic choice).
static thread_local int g_force_cfg = -1, g_dbg = 0, g_stagger = 0, g_stream = 1, g_split_target = 640;      // per calling thread: no state shared between threads
// key 8: the launches that follow carry this many independent images (a TEST-mode batch).  Split-K changes the order in which a sum is
// formed, so its plan must not depend on how many images share a launch: it is made for the launch as it would look in a batch of
// PLAN_IMAGES images (per-image rows x 4: the plan of the 4-image batches the throughput configuration runs), whatever the actual
// batch -- the same image then gives the same bits at batch 1, 4 or 8 (lib/model/test.py:88 is strictly batch-1).  0 = plan by the launch.
static thread_local int g_plan_images = 0;
static constexpr int PLAN_IMAGES = 4;
extern thread_local int g_wino_rows_below;              // csrc/winograd.hip: workgroup count below which the F(4,3) transforms run row-per-thread
extern "C" int frcnn_set_tuning(int key, int value) {
  if (key == 9) { g_wino_rows_below = value > 0 ? value : 256; return FRCNN_OK; }
  if (key == 0) { g_force_cfg = value; return FRCNN_OK; }
  if (key == 1) { g_dbg = value; return FRCNN_OK; }
  if (key == 5) { g_stagger = value; return FRCNN_OK; }        // 0 off; n > 0: second-slot workgroups start n/8 of a tile late
  if (key == 6) { g_stream = value; return FRCNN_OK; }         // 0: never dispatch to k_gemm_stream (A/B runs)
  if (key == 7) { g_split_target = value > 0 ? value : 640; return FRCNN_OK; }   // workgroups a split-K launch aims at (plan_splits)
  if (key == 8) { g_plan_images = value > 0 ? value : 0; return FRCNN_OK; }      // images sharing the next launches (batch-invariant split plan)
  return FRCNN_E_ARG;
}

static int launch_cfg(int id, const ConvParams& p, hipStream_t st) {
  switch (id) {
    case 0: return launch_conv<128, 128, 64, 64, 2, false>(p, st);   // 64 KB LDS: 2 workgroups / CU
    case 1: return launch_conv<128, 128, 64, 64, 3, false>(p, st);   // 96 KB: 1 workgroup / CU, 2 slabs ahead
    case 2: return launch_conv<64, 64, 32, 32, 4, false>(p, st);
    case 3: return launch_conv<32, 64, 32, 32, 4, false>(p, st);
    case 4: return launch_conv<64, 32, 32, 32, 4, false>(p, st);
    case 5: return launch_conv<128, 64, 64, 32, 3, false>(p, st);
    case 6: return launch_conv<64, 128, 32, 64, 3, false>(p, st);
    case 7: return launch_conv<64, 64, 32, 32, 2, false>(p, st);
    case 8: return launch_conv<32, 64, 32, 32, 2, false>(p, st);
    case 9: return launch_conv<32, 32, 32, 32, 4, false>(p, st);
    case 10: return launch_conv<128, 128, 32, 64, 2, false>(p, st);  // 8 waves, 2 accumulators each
    case 11: return launch_conv<128, 128, 32, 64, 3, false, true>(p, st);   // 96 KB ring, loads interleaved with MFMAs
    case 12: return launch_conv<128, 128, 32, 64, 2, false, true>(p, st);
    case 13: return launch_conv<128, 256, 64, 64, 2, false>(p, st);         // 8 waves x 64x64, 96 KB, 1 workgroup / CU
    case 14: return launch_conv<128, 256, 64, 64, 2, false, true>(p, st);
    case 15: return launch_conv<64, 64, 32, 32, 2, false, true>(p, st);
    case 16: return launch_conv<128, 128, 64, 64, 2, false, true>(p, st);
    case 17: return launch_conv<128, 64, 64, 32, 2, false>(p, st);          // 48 KB: 3 workgroups / CU
    case 18: return launch_conv<64, 128, 32, 64, 2, false>(p, st);
    case 19: return launch_conv<128, 64, 32, 32, 2, false>(p, st);          // 8 single-tile waves
    case 20: return launch_conv<128, 128, 64, 64, 2, false, false, true>(p, st);   // reads-first variants of 0 / 10 / 7 / 4
    case 21: return launch_conv<128, 128, 32, 64, 2, false, false, true>(p, st);
    case 22: return launch_conv<64, 64, 32, 32, 2, false, false, true>(p, st);
    case 23: return launch_conv<64, 32, 32, 32, 4, false, false, true>(p, st);
    default: return FRCNN_E_ARG;
  }
}

// ---- split-K for under-filled launches ------------------------------------------------------------------------------
// A 38x63 feature map of ONE image gives 38 x Cout/64 tiles (152 for Cout = 256) and a weight-gradient GEMM Cout/64 x K/64
// tiles for 256 CUs x 2..5 resident workgroups.  Such launches are cut along K: grid.z = S workgroup sets, each reduces
// kchunk slabs into its own [M][Cout] partial (plain stores, no atomics -> deterministic), then k_splitk_finish adds the S
// partials in a fixed order and applies bias / residual / activation.
// rows of a launch as every size-dependent rule sees them (key 8): per-image rows x PLAN_IMAGES
static long long plan_rows(long long M) {
  return (g_plan_images > 0 && M % g_plan_images == 0) ? M / g_plan_images * PLAN_IMAGES : M;
}

static int plan_splits(int M, int Cout, int nsteps) {
  if (g_force_cfg >= 0) return 1;
  M = (int)min(plan_rows(M), (long long)0x7fffffff);
  const long long big = (long long)cdiv(M, 128) * cdiv(Cout, 128);
  if (Cout >= 96 && big >= 384 && nsteps >= 8) return 1;
  const long long tiles = (long long)cdiv(M, 64) * cdiv(Cout, Cout > 32 ? 64 : 32);
  if (tiles >= 384 || nsteps < 16) return 1;
  int S = (int)min((long long)8, (g_split_target + tiles - 1) / tiles);
  S = min(S, nsteps / 8);
  if (S < 2) return 1;
  const int kchunk = cdiv(nsteps, S);
  return cdiv(nsteps, kchunk);
}

extern "C" size_t frcnn_conv2d_workspace_bytes(int N, int OH, int OW, int Cout, int KH, int KW, int Cin, int fold_w) {
  if (fold_w || N <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || Cin <= 0 || Cin % 32) return 0;
  const long long M = (long long)N * OH * OW;
  if (M >= (1ll << 29)) return 0;
  const int S = plan_splits((int)M, Cout, KH * KW * (Cin / 32));
  return S > 1 ? (size_t)S * (size_t)M * (size_t)Cout * sizeof(float) : 0;
}

__global__ void k_splitk_finish(const float* __restrict__ part, int S, int M, int Cout, const float* __restrict__ bias,
                                const float* res, int res_stride, int OH, int OW, int RH, int RW, int act, const float* __restrict__ mask,
                                float* y) {      // res may alias y (in-place gradient accumulation): no __restrict__ on the pair
  const long long total = (long long)M * Cout;
  const int ohow = OH * OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float v = part[i];
    for (int s = 1; s < S; ++s) v += part[(size_t)s * total + i];
    const int m = (int)(i / Cout), n = (int)(i - (long long)m * Cout);
    if (bias) v += bias[n];
    if (res) {
      size_t ro;
      if (res_stride == 1) ro = (size_t)i;
      else {
        const int img = m / ohow, rem = m % ohow, oh = rem / OW, ow = rem % OW;
        ro = ((size_t)(img * RH + oh * res_stride) * RW + ow * res_stride) * Cout + n;
      }
      v += res[ro];
    }
    if (act == FRCNN_ACT_RELU) v = act_relu(v);
    else if (act == FRCNN_ACT_RELU6) v = act_relu6(v);
    if (mask) v = mask[i] > 0.f ? v : 0.f;
    y[i] = v;
  }
}

static int conv2d_impl(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                       const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW, int Cout, int KH, int KW,
                       int stride, int pad_top, int pad_left, int act, int fold_w, void* ws, size_t ws_bytes, const float* mask_d,
                       void* stream);

extern "C" int frcnn_conv2d_nhwc(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                                 const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW,
                                 int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int act, int fold_w,
                                 void* stream) {
  return conv2d_impl(x_d, N, H, W, Cin, w_d, bias_d, residual_d, RH, RW, res_stride, y_d, OH, OW, Cout, KH, KW, stride, pad_top,
                     pad_left, act, fold_w, nullptr, 0, nullptr, stream);
}

// Same convolution with a scratch buffer of frcnn_conv2d_workspace_bytes(...) bytes: under-filled launches run split-K.
extern "C" int frcnn_conv2d_nhwc_ws(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                                    const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW,
                                    int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int act, int fold_w,
                                    void* ws, size_t ws_bytes, void* stream) {
  return conv2d_impl(x_d, N, H, W, Cin, w_d, bias_d, residual_d, RH, RW, res_stride, y_d, OH, OW, Cout, KH, KW, stride, pad_top,
                     pad_left, act, fold_w, ws, ws_bytes, nullptr, stream);
}

// Training (the data-gradient chain): the same convolution followed by the ReLU gradient of the tensor the result is the gradient OF,
//   y = mask > 0 ? y : 0   (mask: float32 [N, OH, OW, Cout], the forward activation; lib/nets/resnet_v1.py's bottleneck: every
// convolution input of the trunk is a ReLU output) -- in the kernel's / the split-K finish's epilogue instead of a frcnn_relu_bwd pass
// over the result.  The select is exact, so the result equals frcnn_conv2d_nhwc_ws + frcnn_relu_bwd bit for bit.
extern "C" int frcnn_conv2d_nhwc_masked_ws(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                                           const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW,
                                           int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int act, const float* mask_d,
                                           void* ws, size_t ws_bytes, void* stream) {
  if (!mask_d) return FRCNN_E_ARG;
  return conv2d_impl(x_d, N, H, W, Cin, w_d, bias_d, residual_d, RH, RW, res_stride, y_d, OH, OW, Cout, KH, KW, stride, pad_top,
                     pad_left, act, 0, ws, ws_bytes, mask_d, stream);
}

static int conv2d_impl(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                       const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW, int Cout, int KH, int KW,
                       int stride, int pad_top, int pad_left, int act, int fold_w, void* ws, size_t ws_bytes, const float* mask_d,
                       void* stream) {
  if (!x_d || !w_d || !y_d) return FRCNN_E_ARG;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0)
    return FRCNN_E_ARG;
  if (act < 0 || act > 2) return FRCNN_E_ARG;
  if (residual_d && (res_stride < 1 || RH < (OH - 1) * res_stride + 1 || RW < (OW - 1) * res_stride + 1)) return FRCNN_E_ARG;
  if (fold_w) {
    if (Cin != 4 || KW > 8) return FRCNN_E_UNSUPPORTED;
  } else if (Cin % 32) {
    return FRCNN_E_UNSUPPORTED;
  }
  if ((long long)N * H * W * Cin >= (1ll << 31) || (long long)N * OH * OW >= (1ll << 31) / 4) return FRCNN_E_UNSUPPORTED;
  ConvParams p;
  p.x = x_d; p.w = w_d; p.bias = bias_d; p.res = residual_d; p.y = y_d; p.mask = mask_d;
  p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.OH = OH; p.OW = OW; p.Cout = Cout; p.KH = KH; p.KW = fold_w ? 1 : KW;
  p.stride = stride; p.pad_top = pad_top; p.pad_left = pad_left; p.act = act;
  p.RH = RH; p.RW = RW; p.res_stride = residual_d ? res_stride : 1;
  p.M = N * OH * OW;
  p.Ktot = fold_w ? KH * 32 : KH * KW * Cin;
  p.nsteps = fold_w ? KH : KH * KW * (Cin / 32);
  p.mtiles = p.ntiles = 0;
  p.gx = p.gw = p.gy = p.gz = 0;
  p.batch = 1;
  p.splits = 1; p.kchunk = p.nsteps;
  p.stagger = 0; p.stagger_slots = 0;
  p.dbg = g_dbg;
  hipStream_t st = (hipStream_t)stream;
  if (fold_w) return launch_conv<128, 64, 32, 64, 3, true>(p, st);
  if (g_force_cfg >= 100) return launch_stream_cfg(g_force_cfg, p, st);
  if (g_force_cfg >= 0) return launch_cfg(g_force_cfg, p, st);
  if (ws) {
    const int S = plan_splits(p.M, Cout, p.nsteps);
    if (S > 1 && (size_t)S * p.M * Cout * sizeof(float) <= ws_bytes) {
      ConvParams q = p;
      q.bias = nullptr; q.res = nullptr; q.act = FRCNN_ACT_NONE; q.y = (float*)ws; q.mask = nullptr;
      q.kchunk = cdiv(p.nsteps, S); q.splits = S; q.gz = (long long)p.M * Cout;
      const int rc = launch_cfg(Cout > 32 ? 15 : 4, q, st);
      if (rc) return rc;
      const long long total = (long long)p.M * Cout;
      hipLaunchKernelGGL(k_splitk_finish, dim3((unsigned)min((long long)2048, (total + 255) / 256)), dim3(256), 0, st, (const float*)ws, S,
                         p.M, Cout, bias_d, residual_d, p.res_stride, OH, OW, RH, RW, act, mask_d, y_d);
      LAUNCH_CHECK();
      return FRCNN_OK;
    }
  }
  // Tile choice, from the measured sweep (profiles/r01_conv_tile_sweep.txt).  f32 MFMA needs few
  // bytes per FLOP, so the limiter is never LDS or HBM but (a) how many of the 1024 SIMDs get a wave
  // and (b) the per-slab barrier/ds_read overhead: 128x128 tiles with 8 waves (2 accumulators each)
  // for the per-RoI tail (M = 14700), 64x64 tiles with a shallow ring (32 KB LDS -> up to 5
  // workgroups per CU) for the 38x63 / 75x125 / 150x250 feature maps.
  // Short-K pointwise convolutions (bottleneck conv3: K = Cin <= 256 into a 4x wider output + residual): a tile is only <= 8 slabs
  // (<= 128 MFMAs per wave), so per-tile launch / prologue / LDS-staged epilogue cost as much as the multiply -> resident
  // streaming workers with register epilogues (profiles/r02_h_stream_sweep.txt: 64.5 -> 51 us on 9576 x 1024 x 256).
  // Which kernel / tile configuration: the configurations do not all add in the same order (8-wave tiles keep two accumulators per
  // sub-tile), so like the split-K plan the choice follows the PLANNED rows (key 8), not the rows of this particular launch.
  const long long Mp = plan_rows(p.M);
  if (g_stream && p.nsteps <= 8 && stride == 1 && KH == 1 && KW == 1 && (!residual_d || p.res_stride == 1)) {
    if (Cout % 128 == 0 && (Mp + 63) / 64 * (Cout / 128) >= 512) {
      const int rc = launch_stream_cfg(108, p, st);
      if (rc != FRCNN_E_UNSUPPORTED) return rc;
    } else if (Cout == 64 && (Mp + 127) / 128 >= 512) {
      const int rc = launch_stream_cfg(106, p, st);
      if (rc != FRCNN_E_UNSUPPORTED) return rc;
    }
  }
  const long long big = (Mp + 127) / 128 * cdiv(Cout, 128);
  if (Cout >= 96 && big >= 384 && p.nsteps >= 8) {
    // a 128x128 tile keeps each SIMD's matrix pipe busy for nsteps * 64 MFMAs * 64 cycles; two workgroups share a CU
    if (g_stagger > 0 && big >= 1024) {
      p.stagger = (int)min((long long)p.nsteps * 8192 * g_stagger / 8, (long long)1 << 30);
      p.stagger_slots = 2;
    }
    return launch_cfg(Cout >= 1024 ? 21 : 20, p, st);   // reads-first; 8 waves help the residual epilogue
  }
  if (g_stagger > 0 && (g_stagger & 16) && Cout > 32 && (long long)cdiv(p.M, 64) * cdiv(Cout, 64) >= 1280) {
    p.stagger = p.nsteps * 1024;                         // 64x64 tiles: 5 workgroups per CU (32 KB LDS), one fifth of a tile apart
    p.stagger_slots = 5;
  }
  if (Cout > 32) return launch_cfg(15, p, st);
  return launch_cfg(4, p, st);
}

// G independent "NT" GEMMs in one launch: y[g][m][n] = sum_k x[g][m][k] * w[g][n][k]  (f32 MFMA, same kernel, grid.y = g).
// Used by the Winograd path (16 transformed positions).  K % 32 == 0.
extern "C" int frcnn_gemm_batched_nt(const float* x_d, const float* w_d, float* y_d, int G, int M, int N, int K, void* stream) {
  if (!x_d || !w_d || !y_d || G <= 0 || M <= 0 || N <= 0 || K <= 0) return FRCNN_E_ARG;
  if (K % 32 || G > 65535) return FRCNN_E_UNSUPPORTED;
  ConvParams p;
  p.x = x_d; p.w = w_d; p.bias = nullptr; p.res = nullptr; p.y = y_d; p.mask = nullptr;
  p.N = 1; p.H = 1; p.W = M; p.Cin = K; p.OH = 1; p.OW = M; p.Cout = N; p.KH = 1; p.KW = 1;
  p.stride = 1; p.pad_top = 0; p.pad_left = 0; p.act = FRCNN_ACT_NONE;
  p.RH = p.RW = 0; p.res_stride = 1;
  p.M = M; p.Ktot = K; p.nsteps = K / 32; p.mtiles = p.ntiles = 0;
  p.gx = (long long)M * K; p.gw = (long long)N * K; p.gy = (long long)M * N; p.gz = 0;
  p.dbg = 0;
  p.batch = G;
  p.splits = 1; p.kchunk = p.nsteps;
  p.stagger = 0; p.stagger_slots = 0;
  const long long Mp = plan_rows(M);                            // key 8: configuration by the planned rows (see conv2d_impl)
  const long long big = (Mp + 127) / 128 * cdiv(N, 128) * G;
  if (g_force_cfg >= 100) return launch_stream_cfg(g_force_cfg, p, (hipStream_t)stream);
  if (g_force_cfg >= 0) return launch_cfg(g_force_cfg, p, (hipStream_t)stream);
  if (g_stream && p.nsteps <= 8 && N % 128 == 0 && (Mp + 63) / 64 * (N / 128) * G >= 512) {      // short-K batched products
    const int rc = launch_stream_cfg(108, p, (hipStream_t)stream);
    if (rc != FRCNN_E_UNSUPPORTED) return rc;
  }
  // 128-row tiles only when they waste at most a quarter of their rows: 160 Winograd tiles of one 38 x 63 image would be 2 x 128 (the RPN
  // 3x3's data gradient, N = 1024: 350 us in 128 x 128 tiles, profiles/r04_ak_*), 64-row tiles cover them with 3 x 64
  const bool waste = (Mp + 127) / 128 * 128 * 4 > Mp * 5;
  return (N >= 96 && big >= 384 && p.nsteps >= 8 && !waste) ? launch_cfg(N >= 1024 ? 21 : 20, p, (hipStream_t)stream)
                                                           : launch_cfg(N > 32 ? 15 : 4, p, (hipStream_t)stream);
}

// HOST: HWIO -> [Cout][KH][KW][Cin] with optional per-output-channel scale (folded frozen BN).
extern "C" int frcnn_pack_filter_hwio(const float* w_hwio, int KH, int KW, int Cin, int Cout, const float* scale, float* out) {
  if (!w_hwio || !out || KH <= 0 || KW <= 0 || Cin <= 0 || Cout <= 0) return FRCNN_E_ARG;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw)
      for (int c = 0; c < Cin; ++c) {
        const float* src = w_hwio + (((size_t)kh * KW + kw) * Cin + c) * Cout;
        for (int o = 0; o < Cout; ++o)
          out[(((size_t)o * KH + kh) * KW + kw) * Cin + c] = scale ? src[o] * scale[o] : src[o];
      }
  return FRCNN_OK;
}
