// Shared helpers for the libfrcnn_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "frcnn_hip.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return FRCNN_E_HIP(_e);      \
  } while (0)

#define LAUNCH_CHECK()                                 \
  do {                                                 \
    hipError_t _e = hipGetLastError();                 \
    if (_e != hipSuccess) return FRCNN_E_HIP(_e);      \
  } while (0)

// One-time setup of a kernel instance PER DEVICE -- the dynamic-LDS attribute and, for the resident-workgroup kernels, how many workgroups
// the device holds -- safe when several host threads drive distinct streams or distinct devices of one process (include/frcnn_hip.h:
// "distinct streams are thread-safe").  One static KernelOnce per kernel instantiation.
struct KernelOnce {
  static constexpr int MAXDEV = 64;
  std::once_flag flag[MAXDEV];
  hipError_t rc[MAXDEV];
  int slots[MAXDEV];
};
static inline hipError_t kernel_once(KernelOnce& k, const void* kern, int threads, size_t lds, int* slots_out = nullptr) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= KernelOnce::MAXDEV) return hipErrorInvalidDevice;
  std::call_once(k.flag[dev], [&] {
    hipError_t r = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu = 0, cus = 0;
    if (r == hipSuccess && slots_out) r = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds);
    if (r == hipSuccess && slots_out) r = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    k.rc[dev] = r;
    k.slots[dev] = per_cu * cus;
  });
  if (slots_out) *slots_out = k.slots[dev];
  return k.rc[dev];
}

// Activations that KEEP NaN.  TensorFlow's Relu is Eigen's max(x, 0) evaluated as (x < 0) ? 0 : x, so a NaN feature stays NaN; fmaxf(NaN, 0)
// (v_max_f32 in IEEE mode) would return 0 and turn an overflowed / poisoned activation into a plausible finite network output.
__device__ __forceinline__ float act_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float act_relu(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float act_relu6(float v) { return v < 0.f ? 0.f : (v > 6.f ? 6.f : v); }
__device__ __forceinline__ float4 act_relu(float4 v) { return make_float4(act_relu(v.x), act_relu(v.y), act_relu(v.z), act_relu(v.w)); }
__device__ __forceinline__ float4 act_relu6(float4 v) { return make_float4(act_relu6(v.x), act_relu6(v.y), act_relu6(v.z), act_relu6(v.w)); }
__device__ __forceinline__ float2 act_relu(float2 v) { return make_float2(act_relu(v.x), act_relu(v.y)); }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Smallest float32 f with (double)f >= thresh: `(double)ovr >= thresh` (lib/nms/cpu_nms.c:2239-2241)
// is then equivalent to the all-f32 test `ovr >= f`.
static inline float thresh_to_f32(double thresh) {
  float f = (float)thresh;
  if ((double)f < thresh) f = nextafterf(f, INFINITY);
  return f;
}

// Monotone float -> uint32 map (larger float <=> larger uint).
__device__ __forceinline__ u32 sortable_u32(float f) {
  u32 b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// 64-bit sort key: (score descending, index ascending) <=> key descending.  Keys are unique.
__device__ __forceinline__ u64 make_key(float score, u32 index) {
  return ((u64)sortable_u32(score) << 32) | (u64)(0xffffffffu - index);
}

// a if a >= b else b  /  a if a <= b else b   (lib/nms/cpu_nms.pyx:11-15)
__device__ __forceinline__ float rmax(float a, float b) { return a >= b ? a : b; }
__device__ __forceinline__ float rmin(float a, float b) { return a <= b ? a : b; }

__device__ __forceinline__ float box_area(const float4 b) {       // cpu_nms.pyx:24
  return ((b.z - b.x) + 1.0f) * ((b.w - b.y) + 1.0f);
}
// cpu_nms.pyx:57-65 with the threshold pre-rounded by thresh_to_f32().  Separate roundings
// (this TU is compiled with -ffp-contract=off), IEEE division.
__device__ __forceinline__ bool iou_suppresses(const float4 a, float aa, const float4 b, float ab, float thr) {
  const float xx1 = rmax(a.x, b.x), yy1 = rmax(a.y, b.y);
  const float xx2 = rmin(a.z, b.z), yy2 = rmin(a.w, b.w);
  const float w = rmax(0.0f, (xx2 - xx1) + 1.0f);
  const float h = rmax(0.0f, (yy2 - yy1) + 1.0f);
  const float inter = w * h;
  const float ovr = inter / ((aa + ab) - inter);
  return ovr >= thr;
}

// The reference's CUDA kernel rule (lib/nms/nms_kernel.cu:24-33,71 devIoU(...) > nms_overlap_thresh; the same test as
// lib/nms/py_cpu_nms.py:35 `ovr <= thresh` kept): identical f32 IoU arithmetic with the +1 pixel convention, but strict
// `>` against the threshold rounded to f32.
__device__ __forceinline__ bool iou_suppresses_gt(const float4 a, float aa, const float4 b, float ab, float thr) {
  const float xx1 = rmax(a.x, b.x), yy1 = rmax(a.y, b.y);
  const float xx2 = rmin(a.z, b.z), yy2 = rmin(a.w, b.w);
  const float w = rmax(0.0f, (xx2 - xx1) + 1.0f);
  const float h = rmax(0.0f, (yy2 - yy1) + 1.0f);
  const float inter = w * h;
  const float ovr = inter / ((aa + ab) - inter);
  return ovr > thr;
}

// tf.image.non_max_suppression's overlap test (TensorFlow r1.2 core/kernels/non_max_suppression_op.cc, ComputeIOU + the
// `> iou_threshold` test): corner order normalised with min/max, NO +1 on widths, degenerate boxes never overlap, f32.
__device__ __forceinline__ float box_area_tf(const float4 b) {
  return (rmax(b.y, b.w) - rmin(b.y, b.w)) * (rmax(b.x, b.z) - rmin(b.x, b.z));
}
__device__ __forceinline__ bool iou_suppresses_tf(const float4 a, float aa, const float4 b, float ab, float thr) {
  if (aa <= 0.0f || ab <= 0.0f) return false;
  const float ymin = rmax(rmin(a.y, a.w), rmin(b.y, b.w)), xmin = rmax(rmin(a.x, a.z), rmin(b.x, b.z));
  const float ymax = rmin(rmax(a.y, a.w), rmax(b.y, b.w)), xmax = rmin(rmax(a.x, a.z), rmax(b.x, b.z));
  const float inter = rmax(ymax - ymin, 0.0f) * rmax(xmax - xmin, 0.0f);
  const float iou = inter / ((aa + ab) - inter);
  return iou > thr;
}

// bbox_transform_inv for one box / one delta quadruple (lib/model/bbox_transform.py:35-65), f32.
__device__ __forceinline__ float4 decode_box(const float4 b, const float4 d) {
  const float w = (b.z - b.x) + 1.0f;
  const float h = (b.w - b.y) + 1.0f;
  const float cx = b.x + 0.5f * w;
  const float cy = b.y + 0.5f * h;
  const float pcx = d.x * w + cx;
  const float pcy = d.y * h + cy;
  const float pw = expf(d.z) * w;
  const float ph = expf(d.w) * h;
  return make_float4(pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph);
}

// inverse of sortable_u32 (exact: the score comes back bit for bit out of a sort key)
__device__ __forceinline__ float unsortable_f32(u32 k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// per-image slice of a batched scratch / tensor: base + b * stride_bytes
template <typename T>
__device__ __forceinline__ T* img_ptr(T* p, size_t stride_bytes, int b) {
  return (T*)((char*)p + stride_bytes * (size_t)b);
}

__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
  const u32 lo = __shfl((u32)v, src, 64), hi = __shfl((u32)(v >> 32), src, 64);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int lane_const) {
  const u32 lo = __builtin_amdgcn_readlane((u32)v, lane_const);
  const u32 hi = __builtin_amdgcn_readlane((u32)(v >> 32), lane_const);
  return ((u64)hi << 32) | lo;
}
