// f32 "NT" GEMM on the 16-bit matrix pipe with BLOCK-SCALED two-piece fp16 operands ("h2"):
//
//     Y[m][n] = act( sum_k X[m][k] * W[n][k] + bias[n] + res[m][n] )            X, W, Y float32
//
// Operand format.  A float32 row segment of 128 consecutive k (a "scale block") is stored as
//     x = 2^-e * (h + l) + r,     h = fp16(x * 2^e),  l = fp16(x * 2^e - h),   e chosen so that max|x * 2^e| is in [2^14, 2^15)
// i.e. two fp16 planes H, L and ONE exact power-of-two scale per (row, 128-k block).  Both roundings are to nearest, the scaling is
// exact, x * 2^e - h is exact in f32; |r| <= 2^-22 |x| (and never more than 2^-39 of the block's largest element, where l leaves the
// fp16 normal range).  A product is evaluated as the three leading cross terms  xh*wh + xh*wl + xl*wh  on v_mfma_f32_32x32x16_f16 with
// f32 accumulation -- each term an exact product of 11-bit significands; what is dropped (xl*wl and the two residuals) is <= 3 * 2^-22
// |x w|, unbiased (round to nearest) and measured at 1e-7 of the output scale on the GEMM shapes of the path, below the f32
// accumulation noise of ANY f32 kernel (tests/test_h2_math_cpu.py states this in numpy; tests/test_dense_gpu.py measures the kernel
// against float64 beside the f32-MFMA kernel).  Three 16-bit MFMAs (8 passes, 16 k) replace eight f32 MFMAs (16 passes, 2 k each) and
// the six of csrc/gemm_x3.hip: 96 instead of 512 / 192 matrix-pipe cycles per 16 k -> ceiling 2500 / 3 = 833 TFLOP/s f32-equivalent.
//
// The weights' scale is per output row n over all of K (static, split once: frcnn_h2_pack_w); the activations' scale is per
// (row m, 128-k block), so every producer can compute it locally: a GEMM workgroup owns 128 output columns of its rows and emits
// the NEXT layer's operand planes straight from its register epilogue (`yp`, `y_inv`); the Winograd transforms and the generic
// splitter (frcnn_h2_split) do the same.  Inside the GEMM a 128-k block is accumulated in a scratch accumulator (first MFMA of the
// block takes C = 0) and folded into the running f32 sum with ONE fma per element by the block's exact scale -- 128 VALU per 96 MFMAs
// per wave, instead of the 176 per 48 of the in-register bf16 split of gemm_x3.
//
// Domain: finite operands whose 128-k blocks span less than ~2^20 between the block maximum and the elements that matter: an element
// e below the maximum m of ITS block carries an absolute error of max(2^-23 |e|, 2^-38 m), so a 1e6 outlier next to O(1) values
// still gives 6e-7 at GEMM level, a 1e8 outlier 8e-5 (tests/test_h2_math_cpu.py::test_domain_outliers_inside_a_scale_block; the exact
// split of csrc/gemm_x3.hip has no such limit).  Post-ReLU / batch-normalised activations span a few decades per pixel.  An inf / nan element makes its block's scale tiny and its own pieces inf / nan (l = inf - inf = nan): the
// output rows that read the block come out nan, where an f32 kernel would return +-inf for a plain overflow (both mean overflow).
//
// Orientation.  The MFMA computes D = Wfrag x Xfrag^T: accumulator lane l holds output ROW m = l & 31 (+ sub-tile), registers walk
// the columns n = 8 (r >> 2) + 4 (l >> 5) + (r & 3).  The activation scale is then ONE value per lane, residual / result / planes move
// as 16- / 8-byte accesses of 4 consecutive n, and the row maximum for the output planes is a per-lane reduction over registers.
//
// Pipeline: k_gemm_x3's skeleton -- resident workgroups walk a static XCD-aware tile list, (tile, slab) is one stream of 32-k slabs
// through an NS-deep LDS ring filled by direct-to-LDS loads (global_load_lds_dwordx4, scalar base + 32-bit lane offset: no address
// VALU), counted vmcnt (NS - 2 slabs stay in flight across the per-slab barrier), 64-byte plane rows with chunk c of row r at position c ^ ((r >> 2) & 3) -> conflict-free
// ds_read_b128 fragments (a ds_read_b128 is served in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): four 64-byte
// rows share a 256-byte bank row, and the four rows of a group with the same r mod 4 differ in (r >> 2) & 3).  A stage = X planes 2 x BM x 64 B + W planes 2 x BN x 64 B + the BM block scales (1 KB):
// 33 KB for 128 x 128 -> two stages = 66 KB = 2 workgroups per CU, or 3-4 stages with one.
#include "h2_common.h"
#include <mutex>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define LDS_AS __attribute__((address_space(3)))

struct GemmH2Params {
  const unsigned short* x; const float* x_inv; const unsigned short* w; const float* w_inv;
  const float* bias; const float* res; const unsigned short* resp; const float* resp_inv; float* y; unsigned short* yp; float* y_inv;
  int M, N, K, batch, act, nsteps, mtiles, ntiles;
  long long Mtot;                     // rows of x / res / y / yp over all batch entries (= batch * M)
  int wshare;                         // 1: every batch entry multiplies by W[0] (frcnn_gemm_h2_mean: entries = images); 0: entry g by W[g]
  const float* mask;                  // frcnn_gemm_h2_masked (training): result = mask > 0 ? result : 0, mask [batch * M][N] float32
  float* mean_part; int mean_rows;    // frcnn_gemm_h2_mean: the result is not stored; column sums of row groups go to mean_part [batch][ceil(M / 32)][2][N]
#ifdef FRCNN_H2_TRACE
  unsigned long long* trace;          // measurement builds only (scratch/h2_trace.py): s_memtime stamps of the first slabs of a few workgroups
#endif
};

// one direct-to-LDS load: 64 lanes x 16 B from (scalar base + per-lane 32-bit byte offset) to LDS [lds_base, +1 KiB), lane-linear.
// Inline asm on purpose (see conv_igemm.hip::glds16): the compiler's scoreboard must not see these loads, the counted waits below are
// the only ones.  M0 is saved / restored inside the statement.
__device__ __forceinline__ void h2_glds16(unsigned voff, const void* sbase, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}

// the 4-byte form: 64 lanes x 4 B -> LDS [lds_base, +256 B), lane-linear.  Fetches the block scales one row per lane, so a tile's
// last rows never read past the tensor (any M; the 16-byte form would fetch four rows per lane)
__device__ __forceinline__ void h2_glds4(unsigned voff, const void* sbase, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}

// the same two loads with M0 declared clobbered instead of saved / restored (the ping-pong schedule: 3 scalar instructions per load)
__device__ __forceinline__ void h2_glds16c(unsigned voff, const void* sbase, unsigned lds_base) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}
__device__ __forceinline__ void h2_glds4c(unsigned voff, const void* sbase, unsigned lds_base) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}

template <int N>
__device__ __forceinline__ void h2_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TUNE bits (A/B measurements, all produce identical results): 1 = the slab's loads are issued in two halves around the first k group
// instead of in one burst after the barrier; 2 = the block scales travel only with the first slab of a 128-k block (NS == 2 only)
template <int BM, int BN, int WM, int WN, int NS, int WPE = 2, int TUNE = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_gemm_h2(const GemmH2Params p) {
  constexpr int NW = (BM / WM) * (BN / WN);
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LA = 2 * BM / 16 / NW;            // X planes: 1 KiB = 16 rows x 64 B per direct-to-LDS instruction
  constexpr int LB = 2 * BN / 16 / NW;            // W planes
  constexpr int G = LA + LB;                      // per wave per slab (wave 0 issues SL more: the block scales, 64 rows each)
  constexpr int SL = BM / 64;
  constexpr int XP = BM * 64, WP = BN * 64;       // bytes of one plane of a stage
  constexpr bool PP = (TUNE & 32) != 0;           // the ping-pong schedule (below): 8 waves in two groups, one barrier apart
  constexpr bool LTB = (TUNE & 256) != 0;         // the light tile boundary (round 5, below): no dependent memory round trip, no store drain
  constexpr bool W21 = (TUNE & 512) != 0;         // plane stores widened to 16 B per lane by v_permlane32_swap pairs (half the instructions)
  constexpr bool DE = (TUNE & 1024) != 0;         // the deferred epilogue (round 6, below): tile t drains under the first 128-k block of tile t + 1
  constexpr int NTA = (TUNE & 16384) ? 2 : 0;     // `nt` on the streams a tile touches once (residual loads, result stores): cache-policy experiment (cfg 42)
  constexpr int S_OFF = 2 * XP + 2 * WP, STAGE = S_OFF + (PP ? 0 : 1024);
  constexpr int RED_OFF = NS * STAGE;             // row-maximum exchange of the plane-emitting epilogue: [BN / WN][BM] floats
  constexpr int S2_OFF = RED_OFF + (BN / WN) * BM * 4;      // TUNE & 2: two 1 KB block-scale regions, alternating per 128-k block
                                                            // PP: [wave][parity] 256 B: each wave's own 64 row scales
  constexpr int CB_OFF = S2_OFF + (PP ? NW * 512 : (TUNE & 2) ? 2048 : 0);   // LTB: [tile parity][filter scales BN | bias BN] floats
  static_assert(!LTB || (!PP && NS == 2 && NW >= 3 && BN % 64 == 0), "light boundary: the two-slot one-barrier-per-slab schedule");
  static_assert(!DE || (LTB && W21 && BN == H2_KB), "deferred epilogue: built on the light boundary's LDS constants and counted waits");
  static_assert(!PP || (NW == 8 && NS == 3 && (TUNE & 2) && ((BM == 256 && WM == 64) || (BM == 128 && WM == 32)) && BN == 128 && WN == 64),
                "ping-pong geometry: 8 waves as 4 (M) x 2 (N), waves 0-3 = the upper half of the rows");
  static_assert((2 * BM / 16) % NW == 0 && (2 * BN / 16) % NW == 0, "tile/wave mismatch");
  static_assert(BM <= 256 && NS >= 2 && NS <= 4, "stage layout");
  static_assert((NS - 2) * (G + SL) <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = p.mtiles * p.ntiles, T = per * p.batch;
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3, W8 = gridDim.x >> 3;
  const int tq = T / 8, tr = T % 8, tn = tq + (xcd < tr ? 1 : 0);
  const int t_end = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + tn;
  const int tile0 = t_end - tn + wx;
  if (tile0 >= t_end) return;
  const int my_tiles = (t_end - tile0 + W8 - 1) / W8;
  int left = __builtin_amdgcn_readfirstlane(my_tiles * p.nsteps);      // slabs of this workgroup's stream not yet issued

  const int wmi = wave / (BN / WN), wni = wave % (BN / WN);
  const int wm0 = wmi * WM, wn0 = wni * WN;
  const int frow = lane & 31, khalf = lane >> 5;

  // ---- issue side: the slab stream --------------------------------------------------------------------------------------------
  const size_t xplane = (size_t)p.Mtot * p.K * 2, wplane = (size_t)p.N * p.K * 2;     // bytes
  unsigned a_off[LA], b_off[LB], s_off[SL];
  const char* i_xb = nullptr; const char* i_wb = nullptr; const char* i_sb = nullptr;  // wave-uniform bases, advanced per slab
  int i_tile = tile0, i_step = 0, i_par = 0, c_par = 0;       // *_par: parity of the running 128-k block count (issue / compute side)
  int i_g = 0, i_bn0 = 0, i_cpar = 0, c_cpar = 0;             // LTB: batch entry / first column of the tile being issued; tile parities
  int pend = 0;                                               // LTB: vector-memory instructions this wave issued AFTER its last slab load
#pragma unroll
  for (int t = 0; t < LB; ++t) {
    const int u = wave * LB + t, plane = u / (BN / 16), row = (u % (BN / 16)) * 16 + (lane >> 2), pos = lane & 3;
    b_off[t] = (unsigned)((size_t)plane * wplane + (size_t)row * p.K * 2 + ((pos ^ ((row >> ((TUNE & 8) ? 1 : 2)) & 3)) * 16));
#ifdef FRCNN_ABLATION     // TUNE & 16: the same bytes as 8 rows x one full 128-byte line per instruction (timing only: wrong LDS image)
    if (TUNE & 16) b_off[t] = (unsigned)((size_t)(u * 8 + (lane >> 3)) * p.K * 2 + (lane & 7) * 16);
#endif
  }
  // tile id -> (batch entry, row tile, column tile).  TUNE & 32768 (experiment, cfg 43): inside a batch entry the ids walk PANELS of 8 row
  // tiles x 8 column tiles, the column groups of odd panels in reverse -- the 64 tiles an XCD has resident then touch 2 + 2 MB of operands
  // per round (K = 512) instead of 1 + 4 MB in the row-major order (the filter planes re-fetched every round, profiles/r06_counters_conv3.json)
  constexpr bool PANEL = (TUNE & 32768) != 0;
  auto tile_coords = [&](int tl, int& g, int& mt, int& nt) {
    g = tl / per;
    const int rem = tl - g * per;
    if (PANEL && (p.ntiles & 7) == 0) {
      const int per_panel = 8 * p.ntiles, panel = rem / per_panel, r2 = rem - panel * per_panel;
      const int pm = min(8, p.mtiles - panel * 8), per_grp = pm * 8;
      int grp = r2 / per_grp;
      const int r3 = r2 - grp * per_grp;
      if (panel & 1) grp = (p.ntiles >> 3) - 1 - grp;
      mt = panel * 8 + (r3 >> 3);
      nt = grp * 8 + (r3 & 7);
      return;
    }
    mt = rem / p.ntiles;
    nt = rem - mt * p.ntiles;
  };
  auto set_tile = [&](int tl) {
    int g, mt, nt;
    tile_coords(tl, g, mt, nt);
    const int bm0 = mt * BM, bn0 = nt * BN;
    const size_t row0 = (size_t)g * p.M + bm0;
    i_g = g; i_bn0 = bn0;
    i_xb = (const char*)p.x + row0 * p.K * 2;
#ifdef FRCNN_ABLATION     // TUNE & 64: every tile reads the FIRST tile's X rows (cache-resident operands: what does the HBM latency cost?)
    if (TUNE & 64) i_xb = (const char*)p.x;
#endif
    i_wb = (const char*)p.w + ((size_t)(p.wshare ? 0 : g) * 2 * p.N + bn0) * p.K * 2;
    i_sb = (const char*)(p.x_inv + row0);
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int u = wave * LA + t, plane = u / (BM / 16), row = (u % (BM / 16)) * 16 + (lane >> 2), pos = lane & 3;
      const int rr = min(row, p.M - 1 - bm0);                                           // rows past M re-read row M - 1
      a_off[t] = (unsigned)((size_t)plane * xplane + (size_t)rr * p.K * 2 + ((pos ^ ((row >> ((TUNE & 8) ? 1 : 2)) & 3)) * 16));
#ifdef FRCNN_ABLATION
      if (TUNE & 16) a_off[t] = (unsigned)((size_t)min(u * 8 + (lane >> 3), p.M - 1 - bm0) * p.K * 2 + (lane & 7) * 16);
#endif
    }
    // block scales of the tile's rows: one float per lane and 64-row group; rows past the tensor re-read its last row (unused)
    const long long last = p.Mtot - 1 - (long long)row0;
#pragma unroll
    for (int j = 0; j < SL; ++j) s_off[j] = (unsigned)(min((long long)((PP ? wm0 : j * 64) + lane), last) * 4);
  };
  const unsigned lds0 = (unsigned)(size_t)(LDS_AS char*)smem;
  auto uniform_ptr = [](const char* q) {
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };
  auto issue_one = [&](int buf, int t) {
    const unsigned sb = lds0 + (unsigned)(buf * STAGE);
    if constexpr (PP) {       // t < G: the wave's share of the slab; t == G: its own 64 block scales, with the first slab of a 128-k block
      if (t < LA) h2_glds16c(a_off[t], uniform_ptr(i_xb), __builtin_amdgcn_readfirstlane(sb + (wave * LA + t) * 1024));
      else if (t < G) h2_glds16c(b_off[t - LA], uniform_ptr(i_wb), __builtin_amdgcn_readfirstlane(sb + 2 * XP + (wave * LB + (t - LA)) * 1024));
      else if (t == G && (i_step & 3) == 0) {
        h2_glds4c(s_off[0], uniform_ptr(i_sb), __builtin_amdgcn_readfirstlane(lds0 + S2_OFF + (wave * 2 + i_par) * 256));
        i_par ^= 1;
      }
      return;
    }
    if (t < LA) h2_glds16(a_off[t], uniform_ptr(i_xb), __builtin_amdgcn_readfirstlane(sb + (wave * LA + t) * 1024));
    else if (t < G) h2_glds16(b_off[t - LA], uniform_ptr(i_wb), __builtin_amdgcn_readfirstlane(sb + 2 * XP + (wave * LB + (t - LA)) * 1024));
    else if (wave == 0) {
      const int j = t - G;
      if (!(TUNE & 2)) h2_glds4(s_off[j], uniform_ptr(i_sb), __builtin_amdgcn_readfirstlane(sb + S_OFF + j * 256));
      else if ((i_step & 3) == 0) {              // once per 128-k block, into the parity region of that block (not a ring slot)
        h2_glds4(s_off[j], uniform_ptr(i_sb), __builtin_amdgcn_readfirstlane(lds0 + S2_OFF + i_par * 1024 + j * 256));
        if (j == SL - 1) i_par ^= 1;
      }
    }
  };
  static_assert(!(TUNE & 2) || NS == 2 || PP, "scales-once needs the uncounted wait of the two-stage ring");
  auto issue_advance = [&]() {                      // after the last piece of a slab
    --left;
    if (++i_step == p.nsteps) {
      i_step = 0; i_tile += W8;
      if (left > 0) set_tile(i_tile);
    } else {
#ifdef FRCNN_ABLATION
      if (TUNE & 16) {          // line s of plane (s & 1): every 128-byte line of both planes is fetched exactly once
        if (i_step & 1) { i_xb += xplane - 64; i_wb += wplane - 64; } else { i_xb += 128 - xplane - 64; i_wb += 128 - wplane - 64; }
      }
#endif
      i_xb += 64; i_wb += 64;
      if ((i_step & 3) == 0) i_sb += (size_t)p.Mtot * 4;       // next 128-k block: next row of x_inv [K/128][Mtot]
    }
  };

  // LTB: the column-block constants of the tile whose FIRST slab has just been issued -- the filter scales and the bias of its BN columns --
  // travel to LDS with that slab (waves 1 and 2, BN / 64 four-byte direct-to-LDS loads each), into the parity region of the tile.  The
  // tile start and the epilogue then read them with ds_read: no global load whose wait would drain the stores / slabs in flight.
  auto issue_cb = [&]() {
    const unsigned cb = lds0 + CB_OFF + i_cpar * (2 * BN * 4);
    if (wave == 1) {
#pragma unroll
      for (int j = 0; j < BN / 64; ++j)          // (j in the scalar base: one per-lane offset for all the loads)
        h2_glds4((unsigned)(lane * 4), uniform_ptr((const char*)(p.w_inv + (size_t)(p.wshare ? 0 : i_g) * p.N + i_bn0 + j * 64)),
                 __builtin_amdgcn_readfirstlane(cb + j * 256));
    }
    if (wave == 2 && p.bias) {
#pragma unroll
      for (int j = 0; j < BN / 64; ++j)
        h2_glds4((unsigned)(lane * 4), uniform_ptr((const char*)(p.bias + i_bn0 + j * 64)), __builtin_amdgcn_readfirstlane(cb + BN * 4 + j * 256));
    }
    i_cpar ^= 1;
  };
  // LTB: wait until at most `n` (rounded down to a template value) of this wave's vector-memory instructions are outstanding.  vmcnt
  // retires in issue order, loads and stores alike (gfx9), so with n = the instructions issued AFTER the slab loads the wave needs, the
  // epilogue's stores and the next tile's residual loads stay in flight across the wait.
  auto wait_pending = [&](int n) {
    if constexpr (DE) {           // the drained tile's stores and the new tile's residual come a few per slab: steps of 8, as a depth-3 tree
      if (n >= 32) {
        if (n >= 48) { if (n >= 56) h2_wait_vmcnt<56>(); else h2_wait_vmcnt<48>(); }
        else { if (n >= 40) h2_wait_vmcnt<40>(); else h2_wait_vmcnt<32>(); }
      } else {
        if (n >= 16) { if (n >= 24) h2_wait_vmcnt<24>(); else h2_wait_vmcnt<16>(); }
        else { if (n >= 8) h2_wait_vmcnt<8>(); else h2_wait_vmcnt<0>(); }
      }
      return;
    }
    if (n >= 63) h2_wait_vmcnt<63>();
    else if (n >= 48) h2_wait_vmcnt<48>();
    else if (n >= 32) h2_wait_vmcnt<32>();
    else if (n >= 16) h2_wait_vmcnt<16>();
    else h2_wait_vmcnt<0>();
  };

  // ---- compute side -------------------------------------------------------------------------------------------------------------
  f32x16 tot[TM][TN], tmp[TM][TN];
  const int sw = (frow >> ((TUNE & 8) ? 1 : 2)) & 3;        // TUNE & 8: round-2 swizzle (two-way bank conflicts), kept for A/B runs
  const int x_row = (wm0 + frow) * 64, w_row = 2 * XP + (wn0 + frow) * 64;

  // TUNE & 4: the next slab's G + SL direct-to-LDS loads are issued ONE AT A TIME between the MFMAs of the current slab (evenly spread
  // over its NM MFMAs, the first after MFMA 1) instead of in a burst after the barrier: a load's issue stall (~60-150 cycles) then falls
  // under the matrix-pipe time of the MFMAs already issued, not in front of the slab's first fragment reads.
  constexpr int NM = 2 * 3 * TM * TN, ND = G + SL;
  // FRCNN_ABLATION builds (the energy ledger, scratch/energy_ledger.py; wrong results by construction): TUNE & 2048 = no MFMAs (the
  // fragments are still read), TUNE & 4096 = no fragment reads (the MFMAs run on whatever the registers hold), TUNE & 8192 = no slab
  // loads after the prologue (the ring is never refilled)
#ifdef FRCNN_ABLATION
  constexpr bool NO_MFMA = (TUNE & 2048) != 0, NO_FRAG = (TUNE & 4096) != 0, NO_LOAD = (TUNE & 8192) != 0;
#else
  constexpr bool NO_MFMA = false, NO_FRAG = false, NO_LOAD = false;
#endif
  h8 kxh[NO_FRAG ? TM : 1], kxl[NO_FRAG ? TM : 1], kwh[NO_FRAG ? TN : 1], kwl[NO_FRAG ? TN : 1];     // (ablation: the kept fragments)
  (void)kxh; (void)kxl; (void)kwh; (void)kwl;
  auto slab_mfma = [&](int cur, auto first_c, auto&& between, auto&& after) {
    constexpr bool FIRST = decltype(first_c)::value;
    const char* sb = smem + cur * STAGE;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                    // two groups of 16 k per 32-wide slab
      if (t == 1) between();
      h8 xh[TM], xl[TM], wh[TN], wl[TN];
      if constexpr (NO_FRAG) {          // (ablation) the first 16-k group of every 128-k block is read and serves the whole block: real operand
        if (FIRST && t == 0) {          // values (the matrix pipe's power follows its data), 1 / 8 of the LDS fragment reads
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const char* q = sb + w_row + j * 32 * 64 + ((khalf ^ sw) * 16);
            kwh[j] = __builtin_bit_cast(h8, *(const uint4*)(q));
            kwl[j] = __builtin_bit_cast(h8, *(const uint4*)(q + WP));
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const char* q = sb + x_row + i * 32 * 64 + ((khalf ^ sw) * 16);
            kxh[i] = __builtin_bit_cast(h8, *(const uint4*)(q));
            kxl[i] = __builtin_bit_cast(h8, *(const uint4*)(q + XP));
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) { wh[j] = kwh[j]; wl[j] = kwl[j]; asm volatile("" : "+v"(wh[j]), "+v"(wl[j])); }
#pragma unroll
        for (int i = 0; i < TM; ++i) { xh[i] = kxh[i]; xl[i] = kxl[i]; asm volatile("" : "+v"(xh[i]), "+v"(xl[i])); }
      } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const char* q = sb + w_row + j * 32 * 64 + (((2 * t + khalf) ^ sw) * 16);
        wh[j] = __builtin_bit_cast(h8, *(const uint4*)(q));
        wl[j] = __builtin_bit_cast(h8, *(const uint4*)(q + WP));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* q = sb + x_row + i * 32 * 64 + (((2 * t + khalf) ^ sw) * 16);
        xh[i] = __builtin_bit_cast(h8, *(const uint4*)(q));
        xl[i] = __builtin_bit_cast(h8, *(const uint4*)(q + XP));
      }
      }
      if constexpr (NO_MFMA) {          // (ablation) the fragments are consumed by an empty asm, the accumulators stay as they are
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(wh[j]), "v"(wl[j]));
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(xh[i]), "v"(xl[i]));
        if (FIRST && t == 0) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) tmp[i][j][r] = 0.f;
        }
        continue;
      }
      // term-major order: consecutive MFMAs write different accumulators
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (FIRST && t == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xh[i], z, 0, 0, 0);
          } else {
            tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xh[i], tmp[i][j], 0, 0, 0);
          }
          after(t * 3 * TM * TN + i * TN + j);
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], xh[i], tmp[i][j], 0, 0, 0);
          after(t * 3 * TM * TN + TM * TN + i * TN + j);
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xl[i], tmp[i][j], 0, 0, 0);
          after(t * 3 * TM * TN + 2 * TM * TN + i * TN + j);
        }
    }
  };

  int c_tile = tile0, c_step = 0, cur = 0, nxt = 0;          // nxt: ring slot of the next slab to issue
  int c_bm0 = 0, c_bn0 = 0, c_g = 0;
  auto set_ctile = [&](int tl) {
    int g, mt, nt;
    tile_coords(tl, g, mt, nt);
    c_bm0 = mt * BM; c_bn0 = nt * BN; c_g = g;
  };
  const float act_lo = p.act == FRCNN_ACT_NONE ? -__builtin_inff() : 0.f;
  const float act_hi = p.act == FRCNN_ACT_RELU6 ? 6.f : __builtin_inff();

  auto rsrc_f = [&](const void* base, long long elems_left, int esize) {
    const long long bytes = elems_left * esize;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)max(0ll, min(bytes, 0x7fffffffll)), 0x00020000);
  };

  struct TileRef { int bm0, bn0, g, cpar; };             // a tile by value: the epilogue parts work on the tile they are handed, not on the c_* cursor
  // ONE descriptor per tensor and batch entry: base = the entry's first row, range = its M * N elements, so the rows of a tile past M fall
  // out of range (stores dropped, loads return 0) whichever sub-tile they belong to; the position inside the entry travels in the per-lane
  // offset = a lane constant (lo4 / lo2 / lo2w below) + a wave-uniform sub-tile offset.  (Rounds 3-5 built a descriptor per 32 x 32
  // sub-tile: four 64-bit address computations and 16 SGPRs per sub-tile and tensor, which the deferred epilogue's pieces cannot afford
  // inside the slab loop.)  M * N < 2^29 elements (checked by the callers): every offset fits 31 bits.
  auto ent_rsrc = [&](const void* base, int g, int esize) {
    const char* b = (const char*)base + (size_t)g * p.M * p.N * esize;
    return __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, p.M * p.N * esize, 0x00020000);
  };
  const int lo4 = (frow * p.N + 4 * khalf) * 4;          // float32 tensors: 16-byte accesses of the accumulator layout
  // planes: 8-byte accesses (lo2) / 16-byte accesses after the v_permlane32_swap pairing (lo2w, W21) -- derived from lo4 where they are
  // used (the empty asm keeps the compiler from hoisting two more lane constants into registers that live through the whole K loop)
  auto lo2_of = [&]() { int v = lo4; asm volatile("" : "+v"(v)); return v >> 1; };
  auto lo2w_of = [&]() { int v = lo4; asm volatile("" : "+v"(v)); return (v >> 1) + 8 * (int)(threadIdx.x >> 5 & 1); };
  auto sub_off = [&](const auto& T, int i, int j) {      // elements from the entry's first row to sub-tile (i, j) of tile T: wave-uniform
    return __builtin_amdgcn_readfirstlane((T.bm0 + wm0 + i * 32) * p.N + T.bn0 + wn0 + j * 32);
  };

#ifdef FRCNN_H2_TRACE
  int tr_slab = 0;
  auto stamp = [&](int point) {       // [workgroup < 16][wave][slab < 64][point < 8]
    if (p.trace && blockIdx.x < 16 && tr_slab < 64 && lane == 0)
      p.trace[(((size_t)blockIdx.x * NW + wave) * 64 + tr_slab) * 8 + point] = __builtin_amdgcn_s_memtime();
  };
  // tile-boundary stamps (scratch/h2_trace_boundary.py): a second region behind the slab stamps, [workgroup < 16][wave][tile < 32][point < 8]
  int tr_tile = 0;
  auto bstamp = [&](int point) {
    if (p.trace && blockIdx.x < 16 && tr_tile < 32 && lane == 0)
      p.trace[(size_t)16 * 8 * 64 * 8 + (((size_t)blockIdx.x * NW + wave) * 32 + tr_tile) * 8 + point] = __builtin_amdgcn_s_memtime();
  };
#else
  auto stamp = [](int) {};
  auto bstamp = [](int) {};
#endif
  float rri[TM];                                                // DE: the residual planes' block scales of the tile's rows (raw residual: first fold)
  (void)rri;
  // The tile's residual (tile under the c_* cursor), sub-tile row i, RAW into tot[i][*]: always four 16-byte loads per sub-tile through ONE instruction stream --
  // float32 residual: quad q <- columns 8 q + 4 khalf .. + 3; residual as operand planes: the inverse of the W21 store pairing -- load 2 p
  // <- H words of columns 16 p + 8 khalf .. + 7, load 2 p + 1 <- the L words of the same columns (un-paired by v_permlane32_swap in the first
  // fold); no residual: the same loads through a zero-length descriptor return 0.  The three cases differ in descriptors and offsets
  // (wave-uniform selects), not in control flow: a branch per case made the register allocator keep a second copy of the accumulators.
  auto load_res_raw = [&](int i) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const TileRef C{c_bm0, c_bn0, c_g, c_cpar};
    const bool planes = !p.res && p.resp;
    const auto none = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000);
    const auto r_even = p.res ? ent_rsrc(p.res, c_g, 4) : planes ? ent_rsrc(p.resp, c_g, 2) : none;
    const auto r_odd = p.res ? r_even : planes ? ent_rsrc(p.resp + (size_t)p.Mtot * p.N, c_g, 2) : none;
    const int step_q = planes ? 0 : 32, step_p = planes ? 32 : 0;
    int base = lo4;
    asm volatile("" : "+v"(base));
    if (planes) base = (base >> 1) + 8 * (int)(threadIdx.x >> 5 & 1);          // lo2w: (frow * N + 8 khalf) * 2
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int vo = base + sub_off(C, i, j) * (planes ? 2 : 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4 ld = __builtin_amdgcn_raw_buffer_load_b128((q & 1) ? r_odd : r_even, vo, q * step_q + (q >> 1) * step_p, NTA);
        tot[i][j][4 * q + 0] = __uint_as_float(ld[0]); tot[i][j][4 * q + 1] = __uint_as_float(ld[1]);
        tot[i][j][4 * q + 2] = __uint_as_float(ld[2]); tot[i][j][4 * q + 3] = __uint_as_float(ld[3]);
      }
      pend += 4;
    }
    if (p.resp) rri[i] = p.resp_inv[(size_t)(c_bn0 / H2_KB) * p.Mtot + (size_t)c_g * p.M + min(c_bm0 + wm0 + i * 32 + frow, p.M - 1)];
  };
  // The accumulators of a tile START at (bias + res) * 2^e_w: the filter row's scale w_inv = 2^-e_w is an exact power of two, so the
  // final  tot * w_inv  = products + bias + res  is one f32 sum evaluated in the scaled domain -- and the residual is fetched when the
  // tile starts (its latency hides under the first 128-k block; it is first touched by that block's fold) instead of after the last MFMA.
  auto init_tot = [&]() {
    if constexpr (DE) return;          // deferred epilogue: the residual arrives with the drain pieces (load_res_raw), sub-tile row by row
    // Every load of the tile start is issued before the first one is used: the filter scales and the bias of the tile's columns
    // (TN x 4 float4 each) and the residual straight into `tot`.  (Round 3 interleaved load, wait and arithmetic per 4 columns inside
    // runtime `if (p.res)` branches: ~30 dependent memory round trips per tile, as long as the whole K loop of a K = 256 tile.)
    const size_t row_base = (size_t)c_g * p.M;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if (p.res) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int m0 = c_bm0 + wm0 + i * 32, nc = c_bn0 + wn0 + j * 32;
          const auto rr = rsrc_f(p.res + (long long)(row_base + m0) * p.N + nc, (long long)(p.M - m0) * p.N - nc, 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 ld = __builtin_amdgcn_raw_buffer_load_b128(rr, (frow * p.N + 4 * khalf) * 4 + 32 * q, 0, 0);
            tot[i][j][4 * q + 0] = __uint_as_float(ld[0]); tot[i][j][4 * q + 1] = __uint_as_float(ld[1]);
            tot[i][j][4 * q + 2] = __uint_as_float(ld[2]); tot[i][j][4 * q + 3] = __uint_as_float(ld[3]);
          }
        }
      pend += TM * TN * 4;
    } else if (p.resp) {
      // the residual as operand planes (the trunk of a bottleneck chain kept as planes only): (h + l) is exact in f32 (<= 23
      // significant bits), times the block's power-of-two scale; the tile's 128 columns are one scale block (BN == 128)
      float ri[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        ri[i] = p.resp_inv[(size_t)(c_bn0 / H2_KB) * p.Mtot + row_base + min(c_bm0 + wm0 + i * 32 + frow, p.M - 1)];
      h4 hv[TM][TN][4], lv[TM][TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int m0 = c_bm0 + wm0 + i * 32, nc = c_bn0 + wn0 + j * 32;
          const long long sbase = (long long)(row_base + m0) * p.N + nc, left_e = (long long)(p.M - m0) * p.N - nc;
          const auto rh = rsrc_f(p.resp + sbase, left_e, 2), rl = rsrc_f(p.resp + (size_t)p.Mtot * p.N + sbase, left_e, 2);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int lo2 = (frow * p.N + 4 * khalf) * 2 + 16 * q;
            hv[i][j][q] = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rh, lo2, 0, 0));
            lv[i][j][q] = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rl, lo2, 0, 0));
          }
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[i][j][4 * q + e] = ((float)hv[i][j][q][e] + (float)lv[i][j][q][e]) * ri[i];
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    }
    // LTB: that is all -- the residual stays RAW in `tot` (its loads in flight behind the previous tile's stores) until the first fold of the
    // tile, four slabs from here, turns it into the scaled start value with the filter scales and the bias read from LDS (fold_first).
    if constexpr (LTB) return;
    // the filter scales and the bias of the tile's columns (issued behind the residual's loads: all of them are in flight together)
    float4 wi[TN][4], bv[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        wi[j][q] = *(const float4*)(p.w_inv + (size_t)(p.wshare ? 0 : c_g) * p.N + c_bn0 + wn0 + j * 32 + 4 * khalf + 8 * q);
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[j][q] = *(const float4*)(p.bias + c_bn0 + wn0 + j * 32 + 4 * khalf + 8 * q);
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // The accumulators of a tile START at (bias + res) * 2^e_w: the filter row's scale w_inv = 2^-e_w is an exact power of two, so the
    // final  tot * w_inv  = products + bias + res  is one f32 sum evaluated in the scaled domain (frcnn_h2_pack_w keeps e_w <= 54, so
    // (bias + res) * 2^e_w cannot overflow; 1 / w_inv is formed exactly from its exponent field).
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float s0 = __uint_as_float(0x7f000000u - __float_as_uint(wi[j][q].x)), s1 = __uint_as_float(0x7f000000u - __float_as_uint(wi[j][q].y));
        const float s2 = __uint_as_float(0x7f000000u - __float_as_uint(wi[j][q].z)), s3 = __uint_as_float(0x7f000000u - __float_as_uint(wi[j][q].w));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          tot[i][j][4 * q + 0] = (tot[i][j][4 * q + 0] + bv[j][q].x) * s0;
          tot[i][j][4 * q + 1] = (tot[i][j][4 * q + 1] + bv[j][q].y) * s1;
          tot[i][j][4 * q + 2] = (tot[i][j][4 * q + 2] + bv[j][q].z) * s2;
          tot[i][j][4 * q + 3] = (tot[i][j][4 * q + 3] + bv[j][q].w) * s3;
        }
      }
  };

  // ---- the epilogue in parts: the standalone form runs them back to back after a tile's last fold; the deferred form (DE) runs them as
  //      pieces under the next tile's first slabs.  A part reads the tile it works on from a TileRef, not from the c_* cursor.
  // v = act(tot * w_inv), kept in tot (bias and residual went in with the start value)
  auto ep_scale_act = [&](const TileRef& T) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n0 = T.bn0 + wn0 + j * 32 + 4 * khalf;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 wi = LTB ? *(const float4*)(smem + CB_OFF + T.cpar * (2 * BN * 4) + (wn0 + j * 32 + 4 * khalf + 8 * q) * 4)
                             : *(const float4*)(p.w_inv + (size_t)(p.wshare ? 0 : T.g) * p.N + n0 + 8 * q);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          tot[i][j][4 * q + 0] = act_clamp(tot[i][j][4 * q + 0] * wi.x, act_lo, act_hi);
          tot[i][j][4 * q + 1] = act_clamp(tot[i][j][4 * q + 1] * wi.y, act_lo, act_hi);
          tot[i][j][4 * q + 2] = act_clamp(tot[i][j][4 * q + 2] * wi.z, act_lo, act_hi);
          tot[i][j][4 * q + 3] = act_clamp(tot[i][j][4 * q + 3] * wi.w, act_lo, act_hi);
        }
      }
    }
  };
  // frcnn_gemm_h2_masked: the ReLU gradient of the tensor this result is the gradient of (an exact select; NaN / inf of the result pass
  // where the mask is positive, like frcnn_relu_bwd).  TUNE & 128: the training instantiations (the inference kernels do not carry this code)
  auto ep_mask = [&](const TileRef& T) {
    if ((TUNE & 128) && p.mask) {
      const auto rk = ent_rsrc(p.mask, T.g, 4);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int vo = lo4 + sub_off(T, i, j) * 4;
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 k[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) k[q] = __builtin_amdgcn_raw_buffer_load_b128(rk, vo + 32 * q, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[i][j][4 * q + e] = __uint_as_float(k[q][e]) > 0.f ? tot[i][j][4 * q + e] : 0.f;
        }
      pend += TM * TN * 4;              // (vector-memory instructions issued after the last slab load: the light boundary's counted wait)
    }
  };
  // frcnn_gemm_h2_mean: reduce_mean over row groups instead of a result tensor (the tail's last convolution feeds only the spatial
  // mean, lib/nets/resnet_v1.py:115-125).  A 32-row accumulator block (lanes = rows) meets at most two groups (mean_rows >= 32):
  // the rows of the group its first row belongs to, and of the next one, are added over the 32 lanes by a fixed xor butterfly
  // and written as two partial rows; k_h2_mean_finish adds a group's 2-3 blocks in ascending order.  Which rows meet in which
  // block depends only on the row index inside the batch entry, so with one batch entry per image the same RoI gives the same
  // bits in every batch slot and at every batch size.
  auto ep_mean = [&](const TileRef& T) {
    const int nblk = (p.M + 31) >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mb = T.bm0 + wm0 + i * 32;
      if (mb >= p.M) continue;                                            // wave-uniform: the block lies past the entry's rows
      const int m = mb + frow, g0 = mb / p.mean_rows, gid = m / p.mean_rows;
      const bool in_a = m < p.M && gid == g0, in_b = m < p.M && gid == g0 + 1;
      float* dst = p.mean_part + ((size_t)T.g * nblk + (mb >> 5)) * 2 * p.N;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n0 = T.bn0 + wn0 + j * 32 + 4 * khalf;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a[4], b[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = tot[i][j][4 * q + e];
            a[e] = in_a ? v : 0.f;
            b[e] = in_b ? v : 0.f;
          }
#pragma unroll
          for (int o = 1; o < 32; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a[e] += __shfl_xor(a[e], o, 64);
              b[e] += __shfl_xor(b[e], o, 64);
            }
          if (frow == 0) {
            *(float4*)(dst + n0 + 8 * q) = make_float4(a[0], a[1], a[2], a[3]);
            *(float4*)(dst + p.N + n0 + 8 * q) = make_float4(b[0], b[1], b[2], b[3]);
          }
          pend += 2;
        }
      }
    }
  };
  // the float32 result of sub-tile row i (16-byte stores in the accumulator layout).
  // (Measured and not kept, profiles/r04_y_*: the same stores ROW-MAJOR through a wave-private LDS block -- 8 rows x one full 128-byte
  // line per instruction instead of 32 rows x 32 bytes -- double the tile boundary (30.8 -> 57.8 thousand cycles for residual + float32 +
  // planes): the extra LDS round trips and the registers they hold cost more than the request count saves.)
  auto ep_store_f32 = [&](const TileRef& T, int i) {
    if (!p.y) return;
    const auto ry = ent_rsrc(p.y, T.g, 4);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int lo = lo4 + sub_off(T, i, j) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 o;
        // (__float_as_uint, not __builtin_bit_cast: bit_cast of an ext-vector ELEMENT lvalue reads element 0 -- clang 19 / ROCm 7.2)
        o[0] = __float_as_uint(tot[i][j][4 * q + 0]); o[1] = __float_as_uint(tot[i][j][4 * q + 1]);
        o[2] = __float_as_uint(tot[i][j][4 * q + 2]); o[3] = __float_as_uint(tot[i][j][4 * q + 3]);
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, lo + 32 * q, 0, NTA);
      }
      pend += 4;
      if constexpr (DE) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the next layer's operand planes: block scale over this workgroup's 128 columns, per row.  Row maxima: per lane over the registers,
  // then between the column waves through LDS (`red`); the caller puts a workgroup barrier between the two halves.
  auto ep_rowmax_write = [&](float (&mx)[TM]) {
    float* red = (float*)(smem + RED_OFF);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(tot[i][j][r]));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      mx[i] = m;
      if (BN / WN > 1 && khalf == 0) red[wni * BM + wm0 + i * 32 + frow] = m;
    }
  };
  auto ep_rowmax_merge = [&](float (&mx)[TM]) {
    const float* red = (const float*)(smem + RED_OFF);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int o = 0; o < BN / WN; ++o) mx[i] = fmaxf(mx[i], red[o * BM + wm0 + i * 32 + frow]);
  };
  auto ep_store_planes = [&](const TileRef& T, int i, float mxi) {
    const size_t row_base = (size_t)T.g * p.M;
    const size_t yplane = (size_t)p.Mtot * p.N;                                 // elements
    float scale, inv;
    h2_block_scale(mxi, scale, inv);
    const int m0 = T.bm0 + wm0 + i * 32;
    if (wni == 0 && khalf == 0 && m0 + frow < p.M)
      p.y_inv[(size_t)(T.bn0 / H2_KB) * p.Mtot + row_base + m0 + frow] = inv;
    const auto rh = ent_rsrc(p.yp, T.g, 2), rl = ent_rsrc(p.yp + yplane, T.g, 2);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int so = sub_off(T, i, j) * 2;
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      if constexpr (W21) {
        // The accumulator layout gives a lane 4 consecutive columns (8 bytes of a plane) per register quad q and puts the next 4 in lane + 32.
        // v_permlane32_swap(vdst = quad 2p, src = quad 2p + 1) exchanges the upper half of vdst with the lower half of src: afterwards a
        // lower lane holds columns 16p .. 16p+7 of its row as [own 2p | upper's 2p], an upper lane columns 16p+8 .. 16p+15 as
        // [lower's 2p+1 | own 2p+1] -- ONE 16-byte store per pair and plane instead of two 8-byte stores: the same bytes at the same
        // addresses from half the store instructions (the epilogue is bound by store ISSUE, not by bytes).
        const int lo = lo2w_of() + so;
#pragma unroll
        for (int pq = 0; pq < 2; ++pq) {
          h4 ha, la, hb, lb;
          h2_split4(make_float4(tot[i][j][8 * pq + 0], tot[i][j][8 * pq + 1], tot[i][j][8 * pq + 2], tot[i][j][8 * pq + 3]), scale, ha, la);
          h2_split4(make_float4(tot[i][j][8 * pq + 4], tot[i][j][8 * pq + 5], tot[i][j][8 * pq + 6], tot[i][j][8 * pq + 7]), scale, hb, lb);
          const u32x2 uha = __builtin_bit_cast(u32x2, ha), uhb = __builtin_bit_cast(u32x2, hb);
          const u32x2 ula = __builtin_bit_cast(u32x2, la), ulb = __builtin_bit_cast(u32x2, lb);
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const auto h0 = __builtin_amdgcn_permlane32_swap(uha[0], uhb[0], false, false), h1 = __builtin_amdgcn_permlane32_swap(uha[1], uhb[1], false, false);
          const auto l0 = __builtin_amdgcn_permlane32_swap(ula[0], ulb[0], false, false), l1 = __builtin_amdgcn_permlane32_swap(ula[1], ulb[1], false, false);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{h0[0], h1[0], h0[1], h1[1]}, rh, lo + 32 * pq, 0, NTA);
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{l0[0], l1[0], l0[1], l1[1]}, rl, lo + 32 * pq, 0, NTA);
        }
        pend += 4;
        if constexpr (DE) __builtin_amdgcn_sched_barrier(0);        // (one sub-tile's split temporaries at a time: the pieces run beside 128 live accumulators)
      } else {
        const int lo = lo2_of() + so;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h4 hh, ll;
          h2_split4(make_float4(tot[i][j][4 * q + 0], tot[i][j][4 * q + 1], tot[i][j][4 * q + 2], tot[i][j][4 * q + 3]), scale, hh, ll);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hh), rh, lo + 16 * q, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ll), rl, lo + 16 * q, 0, 0);
        }
        pend += 8;
      }
    }
  };

  // the standalone epilogue of the tile under the c_* cursor: every part back to back, after the tile's last fold
  auto epilogue = [&]() {
    const TileRef T{c_bm0, c_bn0, c_g, c_cpar};
    bstamp(0);
    ep_scale_act(T);
    bstamp(1);
    ep_mask(T);
    if (p.mean_part) {
      ep_mean(T);
      c_cpar ^= 1;
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) ep_store_f32(T, i);
    bstamp(2);
    if (p.yp) {
      float mx[TM];
      ep_rowmax_write(mx);
      if (BN / WN > 1) {
        // raw barrier + lgkmcnt(0): the exchange goes through LDS only.  __syncthreads() would add vmcnt(0), i.e. wait until the float32
        // stores just issued (and the slabs in flight) have COMPLETED -- 7-9 thousand cycles per drain under a write burst
        // (profiles/r04_w_h2_tile_boundary.txt)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ep_rowmax_merge(mx);
      }
      bstamp(3);
#pragma unroll
      for (int i = 0; i < TM; ++i) ep_store_planes(T, i, mx[i]);
      bstamp(4);
      if (BN / WN > 1) {                                                         // red[] is reused by the next tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    bstamp(5);
    c_cpar ^= 1;
  };

  // ---- the deferred epilogue (TUNE & 1024, round 6) -----------------------------------------------------------------------------------
  // The counters of the conv3-class launches (profiles/r05_counters_conv3.json: waves on s_waitcnt 0.50 of their cycles, matrix pipe busy
  // 0.28) say one workgroup's serial chain binds: K loop -> scale / row maxima / barrier / split / stores -> residual round trip -> K
  // loop.  A second accumulator set would let tile t + 1 multiply while tile t drains -- and the registers for it exist without a third
  // set: the running sum `tot` is first WRITTEN by a tile's first fold, at the end of its fourth slab; until then only the block
  // accumulator `tmp` is live.  So tile t's results stay in `tot` and leave in PIECES under tile t + 1's first three slabs
  //     slab 0:  v = act(tot * w_inv); row maxima over the registers -> LDS          (the slab barrier of slab 1 orders the exchange)
  //     slab 1:  maxima of the other column wave; sub-tile rows [0, HALF): f32 / plane stores, then tile t + 1's residual for those
  //              rows is loaded RAW into the registers the stores have just read
  //     slab 2:  the same for sub-tile rows [HALF, TM)
  // and the first fold (end of slab 3) turns the raw residual into the start value as the light boundary does.  Every element goes
  // through the same operations in the same order as in the standalone epilogue: bit-identical to cfg 9.  The stores are a few per slab
  // (counted in `pend`, waited for in steps of 4), no wait ever names them; the only tile without cover is a workgroup's last one.
  constexpr int HALF = (TM + 1) / 2;
  bool have_prev = false;
  TileRef P{0, 0, 0, 0};                                        // the tile whose results are still in `tot`
  float dmx[TM];                                                // its merged row maxima (slab 1 -> slab 2)
  (void)dmx;
  auto de_piece = [&](auto pos_c) {
    constexpr int POS = decltype(pos_c)::value;
    if constexpr (POS == 0) {
      if (have_prev) {
        ep_scale_act(P);
        ep_mask(P);
        if (p.yp) ep_rowmax_write(dmx);
      }
    } else if constexpr (POS == 1 || POS == 2) {
      constexpr int I0 = POS == 1 ? 0 : HALF, I1 = POS == 1 ? HALF : TM;
      if (have_prev) {
        if (POS == 1 && p.yp && BN / WN > 1) ep_rowmax_merge(dmx);
#pragma unroll
        for (int i = I0; i < I1; ++i) {
          ep_store_f32(P, i);
          if (p.yp) ep_store_planes(P, i, dmx[i]);
        }
      }
#pragma unroll
      for (int i = I0; i < I1; ++i) load_res_raw(i);
    }
  };

  // one slab of the stream: wait for it, let the ring slot it frees be refilled (loads spread over nothing here: they are issued
  // right after the barrier, G + SL instructions, and land under the 24 * TM * TN / 4 MFMAs of this slab and the next NS - 2)
  auto slab = [&](auto pos_c, bool fold, bool first_block) {
    constexpr int POS = decltype(pos_c)::value;          // position inside the 128-k block: 0 .. 3 (0: the block's first MFMAs take C = 0)
    const std::integral_constant<bool, POS == 0> first_c{};
    stamp(0);
    if constexpr (DE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (a drain piece's row maxima are in LDS before the barrier)
    if constexpr (LTB) {
      wait_pending(pend);                              // the slab's loads are older than the `pend` instructions that may stay in flight
    } else if (left >= 1) {                            // steady state: NS - 2 younger slabs may stay in flight
      if (wave == 0) h2_wait_vmcnt<(NS - 2) * (G + SL)>(); else h2_wait_vmcnt<(NS - 2) * G>();
    } else {
      h2_wait_vmcnt<0>();                              // tail of the stream: nothing more will be issued
    }
    stamp(1);
    __builtin_amdgcn_s_barrier();
    stamp(2);
    const bool more = left > 0;
    constexpr int HALF = NO_LOAD ? 0 : (TUNE & 4) ? 0 : (TUNE & 1) ? (G + SL) / 2 : G + SL;
    static_assert(!LTB || NO_LOAD || HALF == G + SL, "light boundary: the slab's loads are issued in one burst");
    if (more) {
#pragma unroll
      for (int t = 0; t < HALF; ++t) issue_one(nxt, t);
      if constexpr (LTB) {
        if (i_step == 0) issue_cb();                   // first slab of a tile: its filter scales and bias travel with it
        pend = 0;
      }
    }
    stamp(3);
    if constexpr (DE) {
      // (scheduling fences: the piece's temporaries die before the slab's fragments are read -- without them the scheduler interleaves the
      // two and the kernel spills)
      __builtin_amdgcn_sched_barrier(0);
      if (first_block) de_piece(pos_c);                // the previous tile drains / this tile's residual arrives, under this slab's MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    float ainv[TM];
    if (fold) {
      const int soff = (TUNE & 2) ? S2_OFF + c_par * 1024 : cur * STAGE + S_OFF;
#pragma unroll
      for (int i = 0; i < TM; ++i) ainv[i] = *(const float*)(smem + soff + (wm0 + i * 32 + frow) * 4);
      c_par ^= 1;
    }
    if (!(TUNE & 4)) {
      slab_mfma(cur, first_c, [&]() {
        stamp(4);
        if (more && !NO_LOAD) {
#pragma unroll
          for (int t = HALF; t < G + SL; ++t) issue_one(nxt, t);
        }
      }, [](int) {});
    } else if (more) {                                 // two code instances: the loads sit between the MFMAs without a branch each
      slab_mfma(cur, first_c, []() {}, [&](int m) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
          if (1 + (d * (NM - 1)) / ND == m) issue_one(nxt, d);
      });
    } else {
      slab_mfma(cur, first_c, []() {}, [](int) {});
    }
    stamp(5);
    if (more) {
      issue_advance();
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
    if (fold) {
      if (LTB && first_block) {
        // The tile's first fold: `tot` holds the raw residual (or zeros); the start value (bias + res) * 2^e_w is formed here, from the
        // filter scales and the bias in LDS -- the same add, multiply and fma as init_tot's, bit for bit.
        const float* cbp = (const float*)(smem + CB_OFF + c_cpar * (2 * BN * 4));
        if (DE && !p.res && p.resp) {
          // the residual arrived as RAW plane words in the accumulator's own registers (load_res_raw): words 8 p .. 8 p + 3 = the H
          // words of columns 16 p + 8 khalf .. + 7, words 8 p + 4 .. + 7 the L words.  v_permlane32_swap of (first pair, second pair)
          // hands every lane its own two quads 2 p, 2 p + 1 (the inverse of the W21 store pairing); (h + l) is exact in f32, times the
          // block's power-of-two scale -- init_tot's expression, bit for bit.
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
              for (int pq = 0; pq < 2; ++pq) {
                unsigned w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(tot[i][j][8 * pq + e]);
                const auto h0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false), h1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
                const auto l0 = __builtin_amdgcn_permlane32_swap(w[4], w[6], false, false), l1 = __builtin_amdgcn_permlane32_swap(w[5], w[7], false, false);
                const h4 ha = __builtin_bit_cast(h4, u32x2{h0[0], h1[0]}), hb = __builtin_bit_cast(h4, u32x2{h0[1], h1[1]});
                const h4 la = __builtin_bit_cast(h4, u32x2{l0[0], l1[0]}), lb = __builtin_bit_cast(h4, u32x2{l0[1], l1[1]});
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  tot[i][j][8 * pq + e] = ((float)ha[e] + (float)la[e]) * rri[i];
                  tot[i][j][8 * pq + 4 + e] = ((float)hb[e] + (float)lb[e]) * rri[i];
                }
              }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 wi = *(const float4*)(cbp + wn0 + j * 32 + 4 * khalf + 8 * q);
            const float4 bv = p.bias ? *(const float4*)(cbp + BN + wn0 + j * 32 + 4 * khalf + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float s0 = __uint_as_float(0x7f000000u - __float_as_uint(wi.x)), s1 = __uint_as_float(0x7f000000u - __float_as_uint(wi.y));
            const float s2 = __uint_as_float(0x7f000000u - __float_as_uint(wi.z)), s3 = __uint_as_float(0x7f000000u - __float_as_uint(wi.w));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              tot[i][j][4 * q + 0] = __builtin_fmaf(tmp[i][j][4 * q + 0], ainv[i], (tot[i][j][4 * q + 0] + bv.x) * s0);
              tot[i][j][4 * q + 1] = __builtin_fmaf(tmp[i][j][4 * q + 1], ainv[i], (tot[i][j][4 * q + 1] + bv.y) * s1);
              tot[i][j][4 * q + 2] = __builtin_fmaf(tmp[i][j][4 * q + 2], ainv[i], (tot[i][j][4 * q + 2] + bv.z) * s2);
              tot[i][j][4 * q + 3] = __builtin_fmaf(tmp[i][j][4 * q + 3], ainv[i], (tot[i][j][4 * q + 3] + bv.w) * s3);
            }
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] = __builtin_fmaf(tmp[i][j][r], ainv[i], tot[i][j][r]);
      }
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
    stamp(6);
#ifdef FRCNN_H2_TRACE
    ++tr_slab;
#endif
  };

  set_tile(i_tile);
  set_ctile(c_tile);
  if constexpr (PP) {
    // ---- the ping-pong schedule ----------------------------------------------------------------------------------------------------
    // 8 waves = two groups of 4 (rows 0..127 / 128..255 of the 256 x 128 tile).  A wave's slab q is two SEGMENTS, each closed by the
    // workgroup barrier:   MEM(q):  all 16 fragments of slab q LDS -> registers, the wave's share of slab q + 2 issued into the ring
    //                               slot that slab q - 1 just left, the fold of the previous 128-k block;
    //                      MFMA(q): the 24 MFMAs, nothing else (s_setprio 1).
    // Group 1 executes one extra barrier before its first segment, so it runs one segment behind: while one group multiplies, the other
    // reads LDS and issues loads -- one wave per SIMD is in its MFMA segment at any time, and the loads of a slab are issued three to
    // four segments before its first read (the one-barrier-per-slab schedule above: one slab = its own duration, which the loaded
    // latency of ~2 500 cycles exceeds, profiles/r03_t_h2_trace.txt).  Same arithmetic, same order: bit-identical to the other configs.
    //   RAW  slab q + 1 is first read in group 0's MEM(q + 1).  The barrier in front of it closes group 0's MFMA(q) and group 1's
    //        MEM(q); each wave waits for ITS loads of slab q + 1 (vmcnt counted: the slab q + 2 loads stay in flight) before it.
    //   WAR  slot (q + 2) % 3 = (q - 1) % 3 was last read in group 1's MEM(q - 1), closed by lgkmcnt(0) + the barrier that precedes
    //        group 0's MEM(q), the first segment to issue into it.
    // The epilogue's barriers (two when planes are emitted) are executed by both groups once per tile, so the one-barrier offset holds.
    const int grp = wave >> 2;
    h8 fxh[2][TM], fxl[2][TM], fwh[2][TN], fwl[2][TN];
    auto seg_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto fold = [&]() {
      float ainv[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) ainv[i] = *(const float*)(smem + S2_OFF + (wave * 2 + c_par) * 256 + (i * 32 + frow) * 4);
      c_par ^= 1;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[i][j][r] = __builtin_fmaf(tmp[i][j][r], ainv[i], tot[i][j][r]);
    };
    // returns whether slab q + 2 was issued (the group-0 wait at the end of MFMA(q) needs to know)
    auto pp_mem = [&](auto pos_c, bool fold_prev) -> bool {
      constexpr int POS = decltype(pos_c)::value;
      const char* sb = smem + cur * STAGE;
      stamp(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const char* q = sb + w_row + j * 32 * 64 + (((2 * t + khalf) ^ sw) * 16);
          fwh[t][j] = __builtin_bit_cast(h8, *(const uint4*)(q));
          fwl[t][j] = __builtin_bit_cast(h8, *(const uint4*)(q + WP));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const char* q = sb + x_row + i * 32 * 64 + (((2 * t + khalf) ^ sw) * 16);
          fxh[t][i] = __builtin_bit_cast(h8, *(const uint4*)(q));
          fxl[t][i] = __builtin_bit_cast(h8, *(const uint4*)(q + XP));
        }
      }
      const bool more = left > 0;
      if (more) {
#pragma unroll
        for (int t = 0; t <= G; ++t) issue_one(nxt, t);
        issue_advance();
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
      }
      if (fold_prev) fold();
      stamp(1);
      if (grp == 1) {                        // slab q + 1 of THIS wave has landed (the slab q + 2 loads, G or G + 1 of them, stay in flight)
        if (more) h2_wait_vmcnt<(POS == 2 ? G + 1 : G)>(); else h2_wait_vmcnt<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(2);
      seg_barrier();
      stamp(3);
      return more;
    };
    auto pp_mfma = [&](auto pos_c, bool issued) {
      constexpr int POS = decltype(pos_c)::value;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (POS == 0 && t == 0) {
              f32x16 z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[t][j], fxh[t][i], z, 0, 0, 0);
            } else {
              tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[t][j], fxh[t][i], tmp[i][j], 0, 0, 0);
            }
          }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl[t][j], fxh[t][i], tmp[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) tmp[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[t][j], fxl[t][i], tmp[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      stamp(4);
      if (grp == 0) {
        if (issued) h2_wait_vmcnt<(POS == 2 ? G + 1 : G)>(); else h2_wait_vmcnt<0>();
      }
      cur = cur + 1 == NS ? 0 : cur + 1;
      stamp(5);
      seg_barrier();
      stamp(6);
#ifdef FRCNN_H2_TRACE
      ++tr_slab;
#endif
    };
    // prologue: slabs 0 and 1 issued, slab 0 landed everywhere
    bool two = false;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (left > 0) {
#pragma unroll
        for (int t = 0; t <= G; ++t) issue_one(nxt, t);
        issue_advance();
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
        two = s2 == 1;
      }
    }
    if (two) h2_wait_vmcnt<G>(); else h2_wait_vmcnt<0>();       // slab 1 carries no scales (nsteps % 4 == 0)
    seg_barrier();
    if (grp == 1) seg_barrier();                                // one segment behind from here on
    const int nkb = p.nsteps >> 2;
    for (int tl = 0; tl < my_tiles; ++tl) {
      init_tot();
      for (int kb = 0; kb < nkb; ++kb) {
        bool is;
        is = pp_mem(std::integral_constant<int, 0>{}, kb > 0); pp_mfma(std::integral_constant<int, 0>{}, is);
        is = pp_mem(std::integral_constant<int, 1>{}, false);  pp_mfma(std::integral_constant<int, 1>{}, is);
        is = pp_mem(std::integral_constant<int, 2>{}, false);  pp_mfma(std::integral_constant<int, 2>{}, is);
        is = pp_mem(std::integral_constant<int, 3>{}, false);  pp_mfma(std::integral_constant<int, 3>{}, is);
      }
      fold();
      epilogue();
      c_tile += W8;
      if (tl + 1 < my_tiles) set_ctile(c_tile);
    }
    if (grp == 0) seg_barrier();                                // the barrier group 1 spent in front
    h2_wait_vmcnt<0>();
    return;
  }
  // ---- prologue: NS - 1 slabs ahead ---------------------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    if (left > 0) {
#pragma unroll
      for (int t = 0; t < G + SL; ++t) issue_one(nxt, t);
      if constexpr (LTB) {
        if (i_step == 0) issue_cb();
        pend = 0;
      }
      issue_advance();
      nxt = nxt + 1 == NS ? 0 : nxt + 1;
    }
  }
  const int nkb = p.nsteps >> 2;
  for (int tl = 0; tl < my_tiles; ++tl) {
    init_tot();
    bstamp(6);
#ifdef FRCNN_H2_TRACE
    ++tr_tile;                                         // (stamps 0-5 of tile t's epilogue and stamp 6 of tile t + 1's start share an index)
#endif
    for (int kb = 0; kb < nkb; ++kb) {
      slab(std::integral_constant<int, 0>{}, false, kb == 0);
      slab(std::integral_constant<int, 1>{}, false, kb == 0);
      slab(std::integral_constant<int, 2>{}, false, kb == 0);
      slab(std::integral_constant<int, 3>{}, true, kb == 0);
    }
    if constexpr (DE) {
      if (tl + 1 < my_tiles) {                           // the results stay in `tot` and drain under the next tile's first slabs
        P = TileRef{c_bm0, c_bn0, c_g, c_cpar};
        have_prev = true;
        c_cpar ^= 1;
        c_tile += W8;
        set_ctile(c_tile);
        continue;
      }
    }
    epilogue();
    c_tile += W8;
    if (tl + 1 < my_tiles) set_ctile(c_tile);
  }
  h2_wait_vmcnt<0>();
}

// ---- splitters -------------------------------------------------------------------------------------------------------------------
// x [M][K] f32 -> planes [2][M][K] fp16 + inv [K/128][M]: one half-wave (32 lanes x 4 k) per (row, 128-k block)
__global__ __launch_bounds__(256) void k_h2_split(const float* __restrict__ x, long long M, int K, unsigned short* __restrict__ planes,
                                                  float* __restrict__ inv_out) {
  const int nkb = K / H2_KB;
  const long long pairs = M * nkb;
  const int sub = threadIdx.x >> 5, l = threadIdx.x & 31;
  for (long long pr = (long long)blockIdx.x * 8 + sub; pr < pairs; pr += (long long)gridDim.x * 8) {
    const long long m = pr / nkb;
    const int kb = (int)(pr - m * nkb);
    const size_t e = (size_t)m * K + (size_t)kb * H2_KB + 4 * l;
    const float4 v = *(const float4*)(x + e);
    float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float scale, inv;
    h2_block_scale(mx, scale, inv);
    h4 hh, ll;
    h2_split4(v, scale, hh, ll);
    *(h4*)(planes + e) = hh;
    *(h4*)(planes + (size_t)M * K + e) = ll;
    if (l == 0) inv_out[(size_t)kb * M + m] = inv;
  }
}

// W [G][N][K] f32 -> planes [G][2][N][K] fp16 + w_inv [G][N]: ONE scale per output row (over all of K), one wave per row
__global__ __launch_bounds__(256) void k_h2_pack_w(const float* __restrict__ w, int rows, int N, int K, unsigned short* __restrict__ planes,
                                                   float* __restrict__ inv_out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (row >= rows) return;
  const float* src = w + (size_t)row * K;
  float mx = 0.f;
  for (int k = 4 * l; k < K; k += 256) {
    const float4 v = *(const float4*)(src + k);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float scale, inv;
  // filter rows: the scale stops at 2^54 (rows whose largest weight is below 2^-40 are zero for every purpose): the GEMM starts its
  // accumulators at (bias + residual) * 2^e_w, which must stay finite
  h2_block_scale(fmaxf(mx, 0x1p-40f), scale, inv);
  const int g = row / N, n = row - g * N;
  unsigned short* ph = planes + ((size_t)g * 2 * N + n) * K;
  unsigned short* pl = ph + (size_t)N * K;
  for (int k = 4 * l; k < K; k += 256) {
    const float4 v = *(const float4*)(src + k);
    h4 hh, ll;
    _Float16 a, b;
    h2_split1(v.x, scale, a, b); hh[0] = a; ll[0] = b;
    h2_split1(v.y, scale, a, b); hh[1] = a; ll[1] = b;
    h2_split1(v.z, scale, a, b); hh[2] = a; ll[2] = b;
    h2_split1(v.w, scale, a, b); hh[3] = a; ll[3] = b;
    *(h4*)(ph + k) = hh;
    *(h4*)(pl + k) = ll;
  }
  if (l == 0) inv_out[row] = inv;
}

#ifdef FRCNN_H2_TRACE
unsigned long long* g_h2_trace = nullptr;
extern "C" void frcnn_h2_set_trace(unsigned long long* buf) { g_h2_trace = buf; }
#endif

extern "C" size_t frcnn_h2_planes_bytes(long long rows, int K) {
  if (rows <= 0 || K <= 0) return 0;
  return (size_t)2 * (size_t)rows * (size_t)K * sizeof(unsigned short);
}

extern "C" int frcnn_h2_pack_w(const float* w_d, int G, int N, int K, void* planes_d, float* w_inv_d, void* stream) {
  if (!w_d || !planes_d || !w_inv_d || G <= 0 || N <= 0 || K <= 0) return FRCNN_E_ARG;
  if (K % 4) return FRCNN_E_UNSUPPORTED;
  const int rows = G * N;
  hipLaunchKernelGGL(k_h2_pack_w, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, w_d, rows, N, K, (unsigned short*)planes_d, w_inv_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_h2_split(const float* x_d, long long M, int K, void* planes_d, float* inv_d, void* stream) {
  if (!x_d || !planes_d || !inv_d || M <= 0 || K <= 0) return FRCNN_E_ARG;
  if (K % H2_KB) return FRCNN_E_UNSUPPORTED;
  const long long pairs = M * (K / H2_KB);
  const unsigned grid = (unsigned)min((pairs + 7) / 8, (long long)256 * 64);
  hipLaunchKernelGGL(k_h2_split, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_d, M, K, (unsigned short*)planes_d, inv_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

template <int BM, int BN, int WM, int WN, int NS, int WPE = 2, int TUNE = 0>
static int launch_h2(const GemmH2Params& q, hipStream_t st) {
  constexpr int NT = (BM / WM) * (BN / WN) * 64;
  constexpr size_t lds = ((TUNE & 32) ? (size_t)NS * (2 * BM * 64 + 2 * BN * 64) + (size_t)(BN / WN) * BM * 4 + (size_t)(NT / 64) * 512
                                      : (size_t)NS * (2 * BM * 64 + 2 * BN * 64 + 1024) + (size_t)(BN / WN) * BM * 4 + ((TUNE & 2) ? 2048 : 0)) +
                         ((TUNE & 256) ? (size_t)2 * 2 * BN * 4 : 0);
  auto kern = k_gemm_h2<BM, BN, WM, WN, NS, WPE, TUNE>;
  static KernelOnce once;
  int slots = 0;                            // resident workgroups on the CURRENT device
  HIP_TRY(kernel_once(once, (const void*)kern, NT, lds, &slots));
  if (slots < 8 || q.N % BN) return FRCNN_E_UNSUPPORTED;
  if (q.yp && BN != H2_KB) return FRCNN_E_UNSUPPORTED;            // the emitted block scale spans exactly the workgroup's columns
  GemmH2Params p = q;
  p.mtiles = cdiv(p.M, BM); p.ntiles = p.N / BN; p.nsteps = p.K / 32;
  const long long T = (long long)p.mtiles * p.ntiles * p.batch;
  if (T >= (1ll << 30)) return FRCNN_E_UNSUPPORTED;
  const int grid = (int)min((long long)(slots / 8) * 8, ((T + 7) / 8) * 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, p);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

static int run_h2(GemmH2Params& p, int cfg, hipStream_t st);

// y[g] = act(x[g] W[g]^T + bias + res[g]), g < G.  x: planes [2][G*M][K] + x_inv [K/128][G*M] (frcnn_h2_split or a producer's `yp`
// output); W: frcnn_h2_pack_w(W [G][N][K]); res / y [G*M][N] f32 (y may be null when only planes are wanted); the residual may be
// given as operand planes instead (res_planes [2][G*M][N] + res_inv [N/128][G*M], e.g. an earlier launch's `yp`);
// yp / y_inv: planes [2][G*M][N] + [N/128][G*M] of the result for the next GEMM (null: not emitted).
// cfg: -1 by shape, else a tile configuration id (A/B runs; per call, no process-wide state).
extern "C" int frcnn_gemm_h2(const void* x_planes_d, const float* x_inv_d, const void* w_planes_d, const float* w_inv_d, const float* bias_d,
                             const float* res_d, const void* res_planes_d, const float* res_inv_d, float* y_d, void* y_planes_d, float* y_inv_d,
                             int G, int M, int N, int K, int act, int cfg, void* stream) {
  if (!x_planes_d || !x_inv_d || !w_planes_d || !w_inv_d || (!y_d && !y_planes_d) || (y_planes_d && !y_inv_d) || G <= 0 || M <= 0 || N <= 0 ||
      K <= 0 || act < 0 || act > 2 || (res_d && res_planes_d) || (res_planes_d && !res_inv_d))
    return FRCNN_E_ARG;
  const long long Mtot = (long long)G * M;
  // 32-bit per-lane byte offsets: both x planes / both W planes of a batch entry / one result row block
  if (K % H2_KB || N % 128 || 4ll * Mtot * K >= (1ll << 32) || 4ll * N * K >= (1ll << 32) ||
      (long long)M * N >= (1ll << 29))
    return FRCNN_E_UNSUPPORTED;
  GemmH2Params p;
  p.x = (const unsigned short*)x_planes_d; p.x_inv = x_inv_d; p.w = (const unsigned short*)w_planes_d; p.w_inv = w_inv_d;
  p.bias = bias_d; p.res = res_d; p.resp = (const unsigned short*)res_planes_d; p.resp_inv = res_inv_d; p.y = y_d; p.yp = (unsigned short*)y_planes_d; p.y_inv = y_inv_d;
  p.M = M; p.N = N; p.K = K; p.batch = G; p.act = act; p.Mtot = Mtot;
  p.nsteps = p.mtiles = p.ntiles = 0;
#ifdef FRCNN_H2_TRACE
  p.trace = g_h2_trace;
#endif
  p.mean_part = nullptr; p.mean_rows = 0; p.wshare = 0; p.mask = nullptr;
  return run_h2(p, cfg, (hipStream_t)stream);
}

// Training (the data-gradient chain, dX = dY W): frcnn_gemm_h2 followed by the ReLU gradient of the tensor the result is the gradient OF --
//   y = mask > 0 ? act(x W^T + bias + res) : 0,  mask [G*M][N] float32 (the forward activation X) -- inside the tile epilogue instead of a
// frcnn_relu_bwd pass over y.  The select is exact: bit for bit frcnn_gemm_h2 + frcnn_relu_bwd.  The operand planes of the result
// (y_planes) are those of the masked tensor.
extern "C" int frcnn_gemm_h2_masked(const void* x_planes_d, const float* x_inv_d, const void* w_planes_d, const float* w_inv_d, const float* bias_d,
                                    const float* res_d, const float* mask_d, float* y_d, void* y_planes_d, float* y_inv_d, int G, int M, int N, int K,
                                    int act, int cfg, void* stream) {
  if (!x_planes_d || !x_inv_d || !w_planes_d || !w_inv_d || !mask_d || (!y_d && !y_planes_d) || (y_planes_d && !y_inv_d) || G <= 0 || M <= 0 ||
      N <= 0 || K <= 0 || act < 0 || act > 2)
    return FRCNN_E_ARG;
  const long long Mtot = (long long)G * M;
  if (K % H2_KB || N % 128 || 4ll * Mtot * K >= (1ll << 32) || 4ll * N * K >= (1ll << 32) || (long long)M * N >= (1ll << 29)) return FRCNN_E_UNSUPPORTED;
  GemmH2Params p;
  p.x = (const unsigned short*)x_planes_d; p.x_inv = x_inv_d; p.w = (const unsigned short*)w_planes_d; p.w_inv = w_inv_d;
  p.bias = bias_d; p.res = res_d; p.resp = nullptr; p.resp_inv = nullptr; p.y = y_d; p.yp = (unsigned short*)y_planes_d; p.y_inv = y_inv_d;
  p.M = M; p.N = N; p.K = K; p.batch = G; p.act = act; p.Mtot = Mtot;
  p.nsteps = p.mtiles = p.ntiles = 0;
#ifdef FRCNN_H2_TRACE
  p.trace = g_h2_trace;
#endif
  p.mean_part = nullptr; p.mean_rows = 0; p.wshare = 0; p.mask = mask_d;
  return run_h2(p, cfg, (hipStream_t)stream);
}

static int run_h2(GemmH2Params& p, int cfg, hipStream_t st) {
  const int G = p.batch, M = p.M, N = p.N, K = p.K;
  if (cfg < 0) {
    // By shape.  Every configuration multiplies and folds in the same order (bit-identical results), so this is a speed choice only.
    // The ping-pong schedule (cfg 21) keeps the matrix pipe busier per joule (profiles/r04_e_h2_power.txt: every configuration sits at
    // the 1 400 W socket limit, so time follows energy) but owns a whole CU per 256 x 128 tile: it pays on long-K launches whose tile
    // epilogues are a small part (K >= 1024: 64 slabs per tile), with at least one tile per CU and 256-row tiles that waste < 4 % of M.
    const long long mt256 = (M + 255) / 256;
#ifndef FRCNN_H2_PP_MIN_K
#define FRCNN_H2_PP_MIN_K 1024          // (scratch/bench_ablation.py rebuilds with other values: 512 measured in the pipeline, profiles/r04_u)
#endif
    // cfg -3 / -4 / -5 (A/B runs, round 5): the ping-pong tiles with 16-byte plane stores (cfg 34) and / or the conv3 class (K = 512,
    // result tensor written: not the fused-mean form, which loses on one workgroup per CU -- profiles/r05_b) on the ping-pong schedule too
    // -6: cfg 34 for the conv3 class only, the long-K launches stay on cfg 21
    const int pp_min_k = ((cfg == -4 || cfg == -5 || cfg == -6) && !p.mean_part) ? 512 : FRCNN_H2_PP_MIN_K;
    const int pp_cfg = (cfg == -3 || cfg == -4 || (cfg == -6 && K < FRCNN_H2_PP_MIN_K)) ? 34 : 21;
    const bool pp = K >= pp_min_k && mt256 * 256 * 100 <= (long long)M * 104 && mt256 * (N / 128) * G >= 256;
    // fewer than 150 tiles of 128 x 128 (a single image's launches: batch-1 latency mode): 64-row tiles, three workgroups per CU
    // (profiles/r03_g_h2_sweep.txt: 21.8 vs 30.9 us on one image's block3 conv1); in the 4-image pipeline these lose (r03_l_ab.txt)
    const bool tiny = (long long)((M + 127) / 128) * (N / 128) * G < 150;
    // cfg == -2: round 4's choice (A/B runs).  Round 5: the same tiles with the light tile boundary and 16-byte plane stores (31, 33):
    // bit-identical, 5-16 % faster on the short-K launches, indifferent elsewhere (profiles/r05_b_h2_conv3_light_boundary.txt)
    // Round 6: the deferred epilogue (cfgs 40 / 41: tile t drains under tile t + 1's first slabs) wherever a result tensor is written --
    // not the fused-mean form (its epilogue is a reduction, kept standalone).  cfg == -7: round 5's choice (A/B runs).
    // Measured (profiles/r06_d_h2_de.txt, isolated 8-image launches): it pays where the boundary is longest -- residual read as planes,
    // planes only out, the identity units of a trunk kept as planes: block3 conv3 65.6 -> 60.0 us, block4 conv3 1 190 -> 1 172 us -- and
    // costs 1-5 % where the standalone epilogue was short (float32 out without residual: its stores then sit in the next tile's slabs
    // instead of beside the co-resident workgroup's K loop); 64-row tiles (cfg 41) lose everywhere.  cfg == -8: DE wherever it exists.
    const bool de = cfg != -2 && cfg != -7 && !p.mean_part && !p.mask && ((p.resp && p.yp && !p.y) || cfg == -8);
    cfg = pp ? pp_cfg : tiny ? (cfg == -2 ? 12 : (de && cfg == -8) ? 41 : 33) : (cfg == -2 ? 9 : de ? 40 : 31);
  }
  if (p.mean_part && (cfg == 40 || cfg == 41)) cfg = cfg == 40 ? 31 : 33;     // (the fused-mean form keeps the standalone epilogue: same tiles)
  if (p.mask) switch (cfg) {            // frcnn_gemm_h2_masked (TUNE & 128: the ReLU-gradient select in the epilogue): every shipped
    case 9: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 128>(p, st);           // configuration has its masked twin since round 6 -- the
    case 12: return launch_h2<64, 128, 32, 64, 2, 2, 128>(p, st);               // data-gradient chain of the training step takes the light
    case 21: return launch_h2<256, 128, 64, 64, 3, 2, 34 + 128>(p, st);         // tile boundary and the 16-byte plane stores by shape, like
    case 30: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 128>(p, st);    // the forward pass (the mask's loads are counted in `pend`)
    case 31: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 128>(p, st);
    case 32: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 512 + 128>(p, st);
    case 33: return launch_h2<64, 128, 32, 64, 2, 2, 256 + 512 + 128>(p, st);
    case 34: return launch_h2<256, 128, 64, 64, 3, 2, 34 + 512 + 128>(p, st);
    case 40: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 1024 + 128>(p, st);
    case 41: return launch_h2<64, 128, 32, 64, 2, 2, 256 + 512 + 1024 + 128>(p, st);
    default: return FRCNN_E_ARG;
  }
  switch (cfg) {
    case 9: return launch_h2<128, 128, 64, 64, 2, 2, 2>(p, st);  // 67 KB: 2 workgroups / CU, one barrier per slab, scales once per 128-k block
    case 12: return launch_h2<64, 128, 32, 64, 2>(p, st);        // 64-row tiles, 4 waves of 32 x 64, 51 KB: 3 workgroups / CU (single-image launches)
    case 21: return launch_h2<256, 128, 64, 64, 3, 2, 34>(p, st); // ping-pong: 256 x 128, 8 waves in two groups a segment apart, 3-slot ring
    case 30: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256>(p, st);        // cfg 9 with the light tile boundary
    case 31: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512>(p, st);  // ... and 16-byte plane stores
    case 32: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 512>(p, st);        // cfg 9 with 16-byte plane stores only
    case 33: return launch_h2<64, 128, 32, 64, 2, 2, 256 + 512>(p, st);       // cfg 12 with both
    case 34: return launch_h2<256, 128, 64, 64, 3, 2, 34 + 512>(p, st);       // cfg 21 (ping-pong) with 16-byte plane stores
    case 40: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 1024>(p, st);   // cfg 31 with the deferred epilogue (round 6)
    case 41: return launch_h2<64, 128, 32, 64, 2, 2, 256 + 512 + 1024>(p, st);        // cfg 33 with the deferred epilogue
#ifdef FRCNN_ABLATION
    // measurement builds only (scratch/ablation_lib.py): the configurations the sweeps under profiles/r03_*, r04_* compare.  All of
    // them multiply and fold in the same order as the three above (bit-identical results; measured, not shipped).
    case 0: return launch_h2<128, 128, 64, 64, 2>(p, st);        // cfg 9 with the block scales in every stage
    case 1: return launch_h2<128, 128, 64, 64, 3>(p, st);        // 100 KB: 1 workgroup / CU, 2 slabs in flight
    case 2: return launch_h2<128, 128, 64, 64, 4>(p, st);        // 134 KB: 3 slabs in flight
    case 3: return launch_h2<256, 128, 64, 64, 2>(p, st);        // 8 waves in lockstep, 100 KB
    case 8: return launch_h2<128, 128, 64, 64, 2, 2, 1>(p, st);  // loads issued in two halves
    case 13: return launch_h2<64, 128, 32, 64, 3>(p, st);        // 64-row tiles, 3-slot ring
    case 15: return launch_h2<64, 128, 32, 64, 4>(p, st);        // ... 4-slot ring
    case 14: return launch_h2<128, 128, 64, 64, 2, 2, 4>(p, st); // loads spread between the MFMAs (spills)
    case 18: return launch_h2<128, 128, 64, 64, 2, 2, 10>(p, st); // round 2's (r >> 1) & 3 swizzle (two-way LDS bank conflicts)
    case 24: return launch_h2<128, 128, 32, 64, 3, 2, 34>(p, st); // ping-pong on 128 x 128 tiles (8 waves of 32 x 64)
    case 22: return launch_h2<256, 128, 64, 64, 3, 2, 34 + 64>(p, st);   // ping-pong with cache-resident X (wrong results by construction)
    case 23: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 64>(p, st);    // cfg 9 with cache-resident X
    case 20: return launch_h2<128, 128, 64, 64, 2, 2, 18>(p, st); // cfg 9's byte count as full-line loads (wrong results by construction)
    // cache-policy experiments on the conv3 class (profiles/r06_h_*; all bit-identical, none shipped): `nt` on the once-touched streams
    // (residual loads, result stores) 1 152 -> 1 472 us on block4 conv3 with MORE L2 misses; the 8 x 8 panel order 1 152 -> 1 138 us (-1.2 %,
    // L2 misses -2 %: the residual / result streams dominate them) and +2 % on block3 conv3
    case 42: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 1024 + 16384>(p, st);
    case 43: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 1024 + 32768>(p, st);
    case 44: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 1024 + 16384 + 32768>(p, st);
    // the energy ledger (scratch/energy_ledger.py): cfg 31 with one ingredient of the slab loop taken out (wrong results by construction)
    case 50: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 2048>(p, st);          // no MFMAs
    case 51: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 4096>(p, st);          // no fragment reads (LDS -> registers)
    case 52: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 8192>(p, st);          // no slab loads (L2 -> LDS) after the prologue
    case 53: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 2048 + 4096>(p, st);   // loads + epilogue only
    case 54: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 4096 + 8192>(p, st);   // MFMAs + epilogue only
    case 55: return launch_h2<128, 128, 64, 64, 2, 2, 2 + 256 + 512 + 2048 + 4096 + 8192>(p, st);   // the epilogue (and the loop skeleton) only
#endif
    default: return FRCNN_E_ARG;
  }
}

// ---- frcnn_gemm_h2_mean: out[g * (M / rows) + r][n] = mean over the `rows` consecutive rows of group r of act(x[g] W^T + bias + res[g]) ------
__global__ __launch_bounds__(256) void k_h2_mean_finish(const float* __restrict__ part, int G, int M, int N4, int rows, float4* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = M / rows;
  if (t >= (long long)G * per * N4) return;
  const int n4 = (int)(t % N4);
  const long long gr = t / N4;
  const int g = (int)(gr / per), r = (int)(gr % per);
  const int nblk = (M + 31) >> 5;
  const int row0 = r * rows, row1 = row0 + rows - 1;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = row0 >> 5; b <= (row1 >> 5); ++b) {                 // the group's blocks in ascending order
    const int which = ((b << 5) / rows == r) ? 0 : 1;              // the block's first row belongs to this group, or to the one before
    const float4 v = *(const float4*)(part + (((size_t)g * nblk + b) * 2 + which) * (size_t)N4 * 4 + (size_t)n4 * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float inv = 1.0f / (float)rows;                            // like k_spatial_mean
  out[t] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
}

extern "C" size_t frcnn_gemm_h2_mean_workspace_bytes(int G, int M, int N) {
  if (G <= 0 || M <= 0 || N <= 0) return 256;
  return (size_t)G * (size_t)((M + 31) / 32) * 2 * (size_t)N * sizeof(float);
}

// The tail's last 1x1 convolution + reduce_mean (lib/nets/resnet_v1.py:115-125) without the [G*M, N] tensor in between: G batch entries
// (one per IMAGE: the reduction order then depends on the RoI's index inside its image only), M rows each, groups of `rows` consecutive
// rows (M % rows == 0, rows >= 32).  x / W / bias / residual as in frcnn_gemm_h2; mean_out [G * M / rows][N]; ws: frcnn_gemm_h2_mean_workspace_bytes.
extern "C" int frcnn_gemm_h2_mean(const void* x_planes_d, const float* x_inv_d, const void* w_planes_d, const float* w_inv_d, const float* bias_d,
                                  const float* res_d, const void* res_planes_d, const float* res_inv_d, int G, int M, int N, int K, int act, int rows,
                                  float* mean_out_d, void* ws, size_t ws_bytes, int cfg, void* stream) {
  if (!x_planes_d || !x_inv_d || !w_planes_d || !w_inv_d || !mean_out_d || !ws || G <= 0 || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2 ||
      rows <= 0 || (res_d && res_planes_d) || (res_planes_d && !res_inv_d))
    return FRCNN_E_ARG;
  const long long Mtot = (long long)G * M;
  if (K % H2_KB || N % 128 || M % rows || rows < 32 || 4ll * Mtot * K >= (1ll << 32) || 4ll * N * K >= (1ll << 32) || (long long)M * N >= (1ll << 29))
    return FRCNN_E_UNSUPPORTED;
  if (frcnn_gemm_h2_mean_workspace_bytes(G, M, N) > ws_bytes) return FRCNN_E_WS;
  GemmH2Params p;
  p.x = (const unsigned short*)x_planes_d; p.x_inv = x_inv_d; p.w = (const unsigned short*)w_planes_d; p.w_inv = w_inv_d;
  p.bias = bias_d; p.res = res_d; p.resp = (const unsigned short*)res_planes_d; p.resp_inv = res_inv_d; p.y = nullptr; p.yp = nullptr; p.y_inv = nullptr;
  p.M = M; p.N = N; p.K = K; p.batch = G; p.act = act; p.Mtot = Mtot;
  p.nsteps = p.mtiles = p.ntiles = 0;
#ifdef FRCNN_H2_TRACE
  p.trace = g_h2_trace;
#endif
  p.mean_part = (float*)ws; p.mean_rows = rows; p.wshare = 1; p.mask = nullptr;
  hipStream_t st = (hipStream_t)stream;
  const int rc = run_h2(p, cfg, st);
  if (rc) return rc;
  const long long tot = (long long)G * (M / rows) * (N / 4);
  hipLaunchKernelGGL(k_h2_mean_finish, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (const float*)ws, G, M, N / 4, rows, (float4*)mean_out_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
