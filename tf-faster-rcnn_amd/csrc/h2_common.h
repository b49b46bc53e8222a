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

// 4 consecutive k of one row -> the two fp16 pieces, as 8-byte words
__device__ __forceinline__ void h2_split4(float4 v, float scale, h4& hh, h4& ll) {
  _Float16 a, b;
  h2_split1(v.x, scale, a, b); hh[0] = a; ll[0] = b;
  h2_split1(v.y, scale, a, b); hh[1] = a; ll[1] = b;
  h2_split1(v.z, scale, a, b); hh[2] = a; ll[2] = b;
  h2_split1(v.w, scale, a, b); hh[3] = a; ll[3] = b;
}

// One (row, 128-k block) of an operand written by one half-wave: lane `l32` (0..31) holds the 4 consecutive k  4*l32 .. 4*l32+3  of the
// block.  `e` = element offset of (row, block start + 4*l32) inside a plane, `plane` = elements per plane, `inv_slot` = where the block's
// 2^-e goes.  All 32 lanes of the half-wave must call (the maximum is a cross-lane reduction).
__device__ __forceinline__ void h2_emit_block32(float4 v, unsigned short* __restrict__ planes, size_t plane, size_t e, float* inv_slot, int l32) {
  const unsigned mx = h2_max32(h2_abs_bits4(v));
  float scale, inv;
  h2_block_scale_bits(mx, scale, inv);
  h4 hh, ll;
  h2_split4(v, scale, hh, ll);
  *(h4*)(planes + e) = hh;
  *(h4*)(planes + plane + e) = ll;
  if (l32 == 0) *inv_slot = inv;
}

// The same for a whole wave holding 2 consecutive k per lane (lane l: k 2l, 2l+1 of the block)
__device__ __forceinline__ void h2_emit_block64(float2 v, unsigned short* __restrict__ planes, size_t plane, size_t e, float* inv_slot, int l64) {
  const unsigned mx = h2_max64(max(h2_abs_bits(v.x), h2_abs_bits(v.y)));
  float scale, inv;
  h2_block_scale_bits(mx, scale, inv);
  _Float16 a, b;
  h2v hh, ll;
  h2_split1(v.x, scale, a, b); hh[0] = a; ll[0] = b;
  h2_split1(v.y, scale, a, b); hh[1] = a; ll[1] = b;
  *(h2v*)(planes + e) = hh;
  *(h2v*)(planes + plane + e) = ll;
  if (l64 == 0) *inv_slot = inv;
}
