// The "h2" operand format of csrc/gemm_h2.hip, shared by every kernel that PRODUCES operand planes (the GEMM epilogue, the splitter,
// the Winograd transforms): block scale rule, two-piece split, and the 32- / 64-lane maxima the producers need.  gfx950 only.
#pragma once
#include "common.h"

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

#define H2_KB 128                     // k per scale block

// The block scale of a 128-k block whose largest magnitude is `mx` (>= 0): (2^e, 2^-e) with mx * 2^e in [2^14, 2^15).  Pure exponent
// arithmetic, so every producer (GEMM epilogue, splitter, Winograd transforms, host reference) derives the identical pair.
// Blocks of zeros / denormals-only clamp at 2^126; inf / nan blocks get a tiny scale and stay inf / nan.
__device__ __forceinline__ void h2_block_scale_bits(unsigned mx_bits, float& scale, float& inv) {
  int ex = (int)((mx_bits >> 23) & 0xffu);
  ex = ex < 15 ? 15 : ex;
  scale = __uint_as_float((unsigned)(268 - ex) << 23);
  inv = __uint_as_float((unsigned)(ex - 14) << 23);
}
__device__ __forceinline__ void h2_block_scale(float mx, float& scale, float& inv) { h2_block_scale_bits(__float_as_uint(mx), scale, inv); }

__device__ __forceinline__ void h2_split1(float v, float scale, _Float16& h, _Float16& l) {
  const float vs = v * scale;
  h = (_Float16)vs;
  l = (_Float16)(vs - (float)h);
}

// |v| as its bit pattern: for non-negative floats the unsigned order IS the float order (NaN patterns sort above inf, so a NaN in the
// block still yields the "inf / nan" scale), and the scale only reads the exponent field -- the reductions below run on v_max_u32.
__device__ __forceinline__ unsigned h2_abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned h2_abs_bits4(float4 v) {
  return max(max(h2_abs_bits(v.x), h2_abs_bits(v.y)), max(h2_abs_bits(v.z), h2_abs_bits(v.w)));
}

template <int CTRL>
__device__ __forceinline__ unsigned h2_dpp(unsigned v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
// maximum over the 16 lanes of a DPP row, in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ unsigned h2_max16(unsigned v) {
  v = max(v, h2_dpp<0xB1>(v));
  v = max(v, h2_dpp<0x4E>(v));
  v = max(v, h2_dpp<0x141>(v));
  v = max(v, h2_dpp<0x140>(v));
  return v;
}
// ... over each half of the wave (lanes 0-31 / 32-63), in every lane: v_permlane16_swap exchanges the odd rows of one copy with the
// even rows of the other
__device__ __forceinline__ unsigned h2_max32(unsigned v) {
  v = h2_max16(v);
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return max(r[0], r[1]);
}
// ... over the wave
__device__ __forceinline__ unsigned h2_max64(unsigned v) {
  v = h2_max32(v);
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return max(r[0], r[1]);
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// 4 consecutive k of one row -> the two fp16 pieces, as 8-byte words.  Vector form so that the compiler issues the packed
// instructions (v_pk_mul_f32, v_cvt_pk_f16_f32: round to nearest even like the scalar conversions of h2_split1 -- same bits)
__device__ __forceinline__ void h2_split4(float4 v, float scale, h4& hh, h4& ll) {
  const f32x4v vs = f32x4v{v.x, v.y, v.z, v.w} * scale;
  hh = __builtin_convertvector(vs, h4);
  ll = __builtin_convertvector(vs - __builtin_convertvector(hh, f32x4v), h4);
}
__device__ __forceinline__ void h2_split2(float2 v, float scale, h2v& hh, h2v& ll) {
  const f32x2v vs = f32x2v{v.x, v.y} * scale;
  hh = __builtin_convertvector(vs, h2v);
  ll = __builtin_convertvector(vs - __builtin_convertvector(hh, f32x2v), h2v);
}

// NR rows of an operand written together by one half-wave with ONE shared scale per 128-k block: lane `l32` (0..31) holds the 4
// consecutive k  4*l32 .. 4*l32+3  of the block for each of the rows; the scale comes from the largest magnitude over all valid rows
// (one cross-lane reduction instead of NR; a producer may choose any power of two that keeps max |x| 2^e below 2^15 -- sharing it
// between a few rows of similar magnitude costs nothing until elements drop below 2^-38 of the group maximum).  `valid`: bit n set =
// row n is written (wave-uniform).  e = row * K + block start + 4*l32 per row.  All 32 lanes must call.
template <int NR>
__device__ __forceinline__ void h2_emit_rows32(const float4* v, const size_t* e, float* const* inv_slot, unsigned valid,
                                               unsigned short* __restrict__ planes, size_t plane, int l32) {
  unsigned mx = 0;
#pragma unroll
  for (int n = 0; n < NR; ++n)
    if ((valid >> n) & 1u) mx = max(mx, h2_abs_bits4(v[n]));
  mx = h2_max32(mx);
  float scale, inv;
  h2_block_scale_bits(mx, scale, inv);
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    if (!((valid >> n) & 1u)) continue;
    h4 hh, ll;
    h2_split4(v[n], scale, hh, ll);
    *(h4*)(planes + e[n]) = hh;
    *(h4*)(planes + plane + e[n]) = ll;
    if (l32 == 0) *inv_slot[n] = inv;
  }
}

// The same for a whole wave holding 2 consecutive k per lane (lane l: k 2l, 2l+1 of the block)
template <int NR>
__device__ __forceinline__ void h2_emit_rows64(const float2* v, const size_t* e, float* const* inv_slot, unsigned valid,
                                               unsigned short* __restrict__ planes, size_t plane, int l64) {
  unsigned mx = 0;
#pragma unroll
  for (int n = 0; n < NR; ++n)
    if ((valid >> n) & 1u) mx = max(mx, max(h2_abs_bits(v[n].x), h2_abs_bits(v[n].y)));
  mx = h2_max64(mx);
  float scale, inv;
  h2_block_scale_bits(mx, scale, inv);
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    if (!((valid >> n) & 1u)) continue;
    h2v hh, ll;
    h2_split2(v[n], scale, hh, ll);
    *(h2v*)(planes + e[n]) = hh;
    *(h2v*)(planes + plane + e[n]) = ll;
    if (l64 == 0) *inv_slot[n] = inv;
  }
}
