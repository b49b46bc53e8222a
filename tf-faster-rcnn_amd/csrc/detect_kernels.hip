// Detection-side kernels of libfrcnn_hip.so (gfx950, wave64): anchors, box codec, (score,index)
// rank sort, bitmask NMS with an on-device greedy reduce, proposal layers, per-class
// post-processing, crop_and_resize, IoU matrix.
//
// These are HBM/latency-bound integer+f32 kernels; they are NOT reshaped into GEMMs.  The TU is
// compiled with -ffp-contract=off so every f32 op keeps its own rounding, as in the reference's
// numpy / Cython code (citations relative to /root/reference/lib).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// anchors  (layer_utils/generate_anchors.py:41-105, layer_utils/snippets.py:14-30)
// ------------------------------------------------------------------------------------------------
extern "C" int frcnn_generate_anchors(int base_size, const double* ratios, int n_ratios, const double* scales,
                                      int n_scales, double* out_base) {
  if (!ratios || !scales || !out_base || n_ratios <= 0 || n_scales <= 0 || base_size <= 0) return FRCNN_E_ARG;
  const double w = (double)base_size, h = (double)base_size;     // base window (0,0,bs-1,bs-1)
  const double cx = 0.5 * (w - 1), cy = 0.5 * (h - 1);
  for (int r = 0; r < n_ratios; ++r) {
    const double ws = nearbyint(sqrt(w * h / ratios[r]));          // np.round: half to even (:90)
    const double hs = nearbyint(ws * ratios[r]);                   // (:91)
    const double x1 = cx - 0.5 * (ws - 1), y1 = cy - 0.5 * (hs - 1);
    const double x2 = cx + 0.5 * (ws - 1), y2 = cy + 0.5 * (hs - 1);
    const double aw = x2 - x1 + 1, ah = y2 - y1 + 1;               // _whctrs of the ratio anchor
    const double acx = x1 + 0.5 * (aw - 1), acy = y1 + 0.5 * (ah - 1);
    for (int s = 0; s < n_scales; ++s) {
      const double sw = aw * scales[s], sh = ah * scales[s];
      double* o = out_base + 4 * ((size_t)r * n_scales + s);
      o[0] = acx - 0.5 * (sw - 1);
      o[1] = acy - 0.5 * (sh - 1);
      o[2] = acx + 0.5 * (sw - 1);
      o[3] = acy + 0.5 * (sh - 1);
    }
  }
  return FRCNN_OK;
}

// anchor n = base[n % A] + stride * (x, y, x, y), (y*W + x) = n / A; float64 add then f32 cast
// exactly like `anchors.reshape(..) + shifts.reshape(..)` -> astype(float32) (snippets.py:26-27).
__device__ __forceinline__ float4 anchor_at(const double* __restrict__ base, int n, int A, int W, int stride) {
  const int a = n % A, pix = n / A;
  const double sx = (double)((pix % W) * stride), sy = (double)((pix / W) * stride);
  const double* b = base + 4 * a;
  return make_float4((float)(b[0] + sx), (float)(b[1] + sy), (float)(b[2] + sx), (float)(b[3] + sy));
}

__global__ void k_anchors(const double* __restrict__ base, int A, int W, int stride, int N, float4* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) out[n] = anchor_at(base, n, A, W, stride);
}

extern "C" int frcnn_generate_anchors_pre(int height, int width, int feat_stride, const double* base_d, int A,
                                          float* anchors_d, void* stream) {
  if (!base_d || !anchors_d || height <= 0 || width <= 0 || A <= 0) return FRCNN_E_ARG;
  const int N = height * width * A;
  hipLaunchKernelGGL(k_anchors, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, base_d, A, width,
                     feat_stride, N, (float4*)anchors_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------------
// stage 1 of proposal_layer: fused anchor generation + bbox_transform_inv + clip_boxes + sort key
// (proposal_layer.py:27-31, bbox_transform.py:35-81).  One thread per anchor; algorithmic traffic
// 4N (fg score) + 16N (deltas) read, 16N (boxes) + 8N (keys) + 4N (rank=0) written.
// ------------------------------------------------------------------------------------------------
__global__ void k_decode_clip_key(const float* __restrict__ prob, const float4* __restrict__ deltas,
                                  const double* __restrict__ base, int A, int W, int stride, int N, float hi_x,
                                  float hi_y, float4* __restrict__ boxes, u64* __restrict__ keys,
                                  u32* __restrict__ rank) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int a = n % A, pix = n / A;
  const float score = prob[(size_t)pix * 2 * A + A + a];                  // fg = channels [A,2A)  (:27)
  float4 b = decode_box(anchor_at(base, n, A, W, stride), deltas[n]);
  b.x = rmax(rmin(b.x, hi_x), 0.0f);                                      // np.maximum(np.minimum(v, dim-1), 0)
  b.y = rmax(rmin(b.y, hi_y), 0.0f);
  b.z = rmax(rmin(b.z, hi_x), 0.0f);
  b.w = rmax(rmin(b.w, hi_y), 0.0f);
  boxes[n] = b;
  keys[n] = make_key(score, (u32)n);
  rank[n] = 0u;
}

// keys for an arbitrary dets [k,5] array (frcnn_nms)
__global__ void k_dets_key(const float* __restrict__ dets, int k, float4* __restrict__ boxes, u64* __restrict__ keys,
                           u32* __restrict__ rank) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= k) return;
  const float* d = dets + 5 * (size_t)n;
  boxes[n] = make_float4(d[0], d[1], d[2], d[3]);
  keys[n] = make_key(d[4], (u32)n);
  rank[n] = 0u;
}

// ------------------------------------------------------------------------------------------------
// stage 2: rank[i] = #{ j : key[j] > key[i] }  -- the full `argsort()[::-1]` of proposal_layer.py:34
// as a counting sort over unique 64-bit keys.  grid = (ceil(N/256), JS): each block owns 256 keys
// and one slice of the j range; the j loop index is wave-uniform, so the compared key comes through
// the scalar cache (s_load) and the body is v_cmp_gt_u64 + add-with-carry.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank(const u64* __restrict__ keys, int N, int jchunk, u32* __restrict__ rank) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const u64 mine = (i < N) ? keys[i] : ~0ull;
  const int j0 = blockIdx.y * jchunk;
  const int j1 = min(N, j0 + jchunk);
  u32 cnt = 0;
  int j = j0;
  for (; j + 8 <= j1; j += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) cnt += (keys[j + u] > mine) ? 1u : 0u;
  }
  for (; j < j1; ++j) cnt += (keys[j] > mine) ? 1u : 0u;
  if (i < N && cnt) atomicAdd(&rank[i], cnt);
}

// stage 3: scatter the top-K (rank < K) into score order.
__global__ void k_scatter_topk(const float4* __restrict__ boxes, const u64* __restrict__ keys,
                               const u32* __restrict__ rank, int N, int K, float4* __restrict__ sboxes,
                               float* __restrict__ sscores, int* __restrict__ sidx, const float* __restrict__ prob,
                               int A, const float* __restrict__ dets, const float* __restrict__ flat) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const u32 r = rank[n];
  if (r >= (u32)K) return;
  sboxes[r] = boxes[n];
  sidx[r] = n;
  float s;
  if (prob) {
    const int a = n % A, pix = n / A;
    s = prob[(size_t)pix * 2 * A + A + a];
  } else if (dets) {
    s = dets[5 * (size_t)n + 4];
  } else {
    s = flat[n];
  }
  sscores[r] = s;
}

// ------------------------------------------------------------------------------------------------
// stage 4: suppression bitmask.  Replaces nms_kernel (nms/nms_kernel.cu:34-78) for wave64: one wave
// = one 64-box row tile x one 64-box column tile; lane i builds its own 64-bit word, no ballot
// needed.  Only the upper triangle (column tile >= row tile) is computed and ever read.
// ------------------------------------------------------------------------------------------------
template <bool TF>
__global__ __launch_bounds__(256) void k_nms_mask(const float4* __restrict__ boxes, int K, int cb, float thr,
                                                  u64* __restrict__ mask) {
  __shared__ float4 cbox[64];
  __shared__ float carea[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = blockIdx.x;                 // column tile
  const int rt = blockIdx.y * 4 + wave;      // row tile of this wave
  if (threadIdx.x < 64) {
    const int j = ct * 64 + threadIdx.x;
    const float4 b = (j < K) ? boxes[j] : make_float4(0, 0, 0, 0);
    cbox[threadIdx.x] = b;
    carea[threadIdx.x] = TF ? box_area_tf(b) : box_area(b);
  }
  __syncthreads();
  if (rt >= cb || ct < rt) return;
  const int i = rt * 64 + lane;
  if (i >= K) return;
  const float4 bi = boxes[i];
  const float ai = TF ? box_area_tf(bi) : box_area(bi);
  const int nj = min(64, K - ct * 64);
  u64 bits = 0;
  for (int j = 0; j < nj; ++j) {
    const int gj = ct * 64 + j;
    const bool sup = TF ? iou_suppresses_tf(bi, ai, cbox[j], carea[j], thr) : iou_suppresses(bi, ai, cbox[j], carea[j], thr);
    if (gj > i && sup) bits |= (1ull << j);
  }
  mask[(size_t)i * cb + ct] = bits;
}

// ------------------------------------------------------------------------------------------------
// stage 5: greedy reduce on device (the reference does this on the host after a D2H copy of the
// whole mask, nms_kernel.cu:118-140).  One wave walks the boxes in 64-box chunks: the in-chunk
// decisions are resolved with scalar bit operations on the diagonal words, then the mask rows of
// the kept boxes are OR-ed into the per-lane `removed` words with up to 8 rows in flight.  The scan
// stops as soon as max_keep boxes are kept (== truncating the keep list, proposal_layer.py:44-45).
// ------------------------------------------------------------------------------------------------
template <typename Emit>
__device__ __forceinline__ int greedy_reduce_wave(const u64* __restrict__ mask, int K, int cb, int max_keep, Emit emit) {
  const int lane = threadIdx.x & 63;
  u64 remv0 = 0, remv1 = 0, remv2 = 0, remv3 = 0;      // removed-bit words lane, lane+64, lane+128, lane+192
  int total = 0;
  for (int c = 0; c < cb && total < max_keep; ++c) {
    const int i = c * 64 + lane;
    const u64 d = (i < K) ? mask[(size_t)i * cb + c] : 0ull;
    const int slot = c >> 6;
    u64 sel = remv0;
    if (slot == 1) sel = remv1;
    if (slot == 2) sel = remv2;
    if (slot == 3) sel = remv3;
    sel = shfl_u64(sel, c & 63);
    // readfirstlane returns a (signed) int: go through u32 or the low word sign-extends into the high one
    const u32 cur_lo = (u32)__builtin_amdgcn_readfirstlane((u32)sel);
    const u32 cur_hi = (u32)__builtin_amdgcn_readfirstlane((u32)(sel >> 32));
    u64 cur = ((u64)cur_hi << 32) | (u64)cur_lo;
    const int nvalid = min(64, K - c * 64);
    if (nvalid < 64) cur |= (~0ull) << nvalid;
    u64 kept = 0;
#pragma unroll
    for (int b = 0; b < 64; ++b) {
      const u64 db = readlane_u64(d, b);
      if (!((cur >> b) & 1ull)) {
        kept |= (1ull << b);
        cur |= db;
      }
    }
    if ((kept >> lane) & 1ull) {
      const int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < max_keep) emit(pos, i);
    }
    total += __popcll(kept);
    if (total >= max_keep) break;
    // OR the mask rows of the kept boxes into the removed words (words > c only)
    u64 kk = kept;
    const int w0 = lane, w1 = lane + 64, w2 = lane + 128, w3 = lane + 192;
    while (kk) {
      int bq[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        bq[q] = kk ? (__ffsll((long long)kk) - 1) : bq[q ? q - 1 : 0];
        if (kk) kk &= kk - 1;
      }
      u64 v0[8], v1[8], v2[8], v3[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const u64* row = mask + (size_t)(c * 64 + bq[q]) * cb;
        v0[q] = (w0 > c && w0 < cb) ? row[w0] : 0ull;
        v1[q] = (w1 > c && w1 < cb) ? row[w1] : 0ull;
        v2[q] = (w2 > c && w2 < cb) ? row[w2] : 0ull;
        v3[q] = (w3 > c && w3 < cb) ? row[w3] : 0ull;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        remv0 |= v0[q];
        remv1 |= v1[q];
        remv2 |= v2[q];
        remv3 |= v3[q];
      }
    }
  }
  return min(total, max_keep);
}

// keep list only (frcnn_nms / frcnn_nms_sorted)
__global__ __launch_bounds__(64) void k_nms_reduce_keep(const u64* __restrict__ mask, int K, int cb, int max_keep,
                                                        const int* __restrict__ sidx, int* __restrict__ keep,
                                                        int* __restrict__ num) {
  const int n = greedy_reduce_wave(mask, K, cb, max_keep, [&](int pos, int i) { keep[pos] = sidx ? sidx[i] : i; });
  if (threadIdx.x == 0) *num = n;
}

// keep + gather into the proposal blob (proposal_layer.py:44-51): rois [post,5], scores [post]
__global__ __launch_bounds__(64) void k_nms_reduce_rois(const u64* __restrict__ mask, int K, int cb, int max_keep,
                                                        const float4* __restrict__ sboxes,
                                                        const float* __restrict__ sscores, float* __restrict__ rois,
                                                        float* __restrict__ scores, int* __restrict__ num) {
  const int n = greedy_reduce_wave(mask, K, cb, max_keep, [&](int pos, int i) {
    const float4 b = sboxes[i];
    float* r = rois + 5 * (size_t)pos;
    r[0] = 0.0f; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
    scores[pos] = sscores[i];
  });
  for (int p = n + (int)threadIdx.x; p < max_keep; p += 64) {
    float* r = rois + 5 * (size_t)p;
    r[0] = r[1] = r[2] = r[3] = r[4] = 0.0f;
    scores[p] = 0.0f;
  }
  if (threadIdx.x == 0) *num = n;
}

// Same scan for K up to 65536 boxes (tf.image.non_max_suppression sees ALL H*W*A anchors, proposal_layer.py:56-72): the
// removed-bit words live in LDS (word w is owned by lane w & 63, so there is no cross-lane hazard beyond the chunk read).
#define NMS_WIDE_WORDS 1024
template <typename Emit>
__device__ __forceinline__ int greedy_reduce_wave_wide(const u64* __restrict__ mask, int K, int cb, int max_keep, Emit emit) {
  __shared__ u64 remv[NMS_WIDE_WORDS];
  const int lane = threadIdx.x & 63;
  for (int w = lane; w < cb; w += 64) remv[w] = 0ull;
  __syncthreads();
  int total = 0;
  for (int c = 0; c < cb && total < max_keep; ++c) {
    const int i = c * 64 + lane;
    const u64 d = (i < K) ? mask[(size_t)i * cb + c] : 0ull;
    const u64 sel = remv[c];
    const u32 cur_lo = (u32)__builtin_amdgcn_readfirstlane((u32)sel);
    const u32 cur_hi = (u32)__builtin_amdgcn_readfirstlane((u32)(sel >> 32));
    u64 cur = ((u64)cur_hi << 32) | (u64)cur_lo;
    const int nvalid = min(64, K - c * 64);
    if (nvalid < 64) cur |= (~0ull) << nvalid;
    u64 kept = 0;
#pragma unroll
    for (int b = 0; b < 64; ++b) {
      const u64 db = readlane_u64(d, b);
      if (!((cur >> b) & 1ull)) {
        kept |= (1ull << b);
        cur |= db;
      }
    }
    if ((kept >> lane) & 1ull) {
      const int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < max_keep) emit(pos, i);
    }
    total += __popcll(kept);
    if (total >= max_keep) break;
    for (int w = (c + 1) - ((c + 1) & 63) + lane; w < cb; w += 64) {     // words > c, lane-owned
      if (w <= c) continue;
      u64 acc = remv[w];
      u64 kk = kept;
      while (kk) {
        const int b = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        acc |= mask[(size_t)(c * 64 + b) * cb + w];
      }
      remv[w] = acc;
    }
    __syncthreads();          // single wave: orders the LDS writes before the next chunk's broadcast read
  }
  return min(total, max_keep);
}

__global__ __launch_bounds__(64) void k_nms_reduce_keep_wide(const u64* __restrict__ mask, int K, int cb, int max_keep,
                                                             const int* __restrict__ sidx, int* __restrict__ keep,
                                                             int* __restrict__ num) {
  const int n = greedy_reduce_wave_wide(mask, K, cb, max_keep, [&](int pos, int i) { keep[pos] = sidx ? sidx[i] : i; });
  if (threadIdx.x == 0) *num = n;
}

__global__ __launch_bounds__(64) void k_nms_reduce_rois_wide(const u64* __restrict__ mask, int K, int cb, int max_keep,
                                                             const float4* __restrict__ sboxes,
                                                             const float* __restrict__ sscores, float* __restrict__ rois,
                                                             float* __restrict__ scores, int* __restrict__ num) {
  const int n = greedy_reduce_wave_wide(mask, K, cb, max_keep, [&](int pos, int i) {
    const float4 b = sboxes[i];
    float* r = rois + 5 * (size_t)pos;
    r[0] = 0.0f; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
    scores[pos] = sscores[i];
  });
  for (int p = n + (int)threadIdx.x; p < max_keep; p += 64) {
    float* r = rois + 5 * (size_t)p;
    r[0] = r[1] = r[2] = r[3] = r[4] = 0.0f;
    scores[p] = 0.0f;
  }
  if (threadIdx.x == 0) *num = n;
}

// keys for separate boxes [k,4] / scores [k] arrays (tf.image.non_max_suppression's inputs)
__global__ void k_boxes_scores_key(const float4* __restrict__ in_boxes, const float* __restrict__ in_scores, int k,
                                   float4* __restrict__ boxes, u64* __restrict__ keys, u32* __restrict__ rank) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= k) return;
  boxes[n] = in_boxes[n];
  keys[n] = make_key(in_scores[n], (u32)n);
  rank[n] = 0u;
}

// proposal_top_layer gather (proposal_top_layer.py:46-55)
__global__ void k_top_rois(const float4* __restrict__ sboxes, const float* __restrict__ sscores, int K,
                           float* __restrict__ rois, float* __restrict__ scores) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= K) return;
  const float4 b = sboxes[p];
  float* r = rois + 5 * (size_t)p;
  r[0] = 0.0f; r[1] = b.x; r[2] = b.y; r[3] = b.z; r[4] = b.w;
  scores[p] = sscores[p];
}

// ------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------
struct SortWs {
  float4* boxes; u64* keys; u32* rank; float4* sboxes; float* sscores; int* sidx; u64* mask; size_t bytes;
};
static SortWs carve(void* ws, int N, int K) {
  SortWs s;
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? (void*)(p + o) : (void*)nullptr; };
  s.boxes = (float4*)take(sizeof(float4) * (size_t)N);
  s.keys = (u64*)take(sizeof(u64) * (size_t)N);
  s.rank = (u32*)take(sizeof(u32) * (size_t)N);
  s.sboxes = (float4*)take(sizeof(float4) * (size_t)K);
  s.sscores = (float*)take(sizeof(float) * (size_t)K);
  s.sidx = (int*)take(sizeof(int) * (size_t)K);
  s.mask = (u64*)take(sizeof(u64) * (size_t)K * (size_t)cdiv(K, 64));
  s.bytes = off;
  return s;
}

static int launch_rank_scatter(const SortWs& s, int N, int K, const float* prob, int A, const float* dets, hipStream_t st,
                               const float* flat = nullptr) {
  // enough (i-block, j-slice) pairs to fill 256 CUs x 8 waves/SIMD, at least 2048 keys per slice
  const int iblocks = cdiv(N, 256);
  int js = max(1, min(cdiv(N, 2048), cdiv(4096, iblocks)));
  const int jchunk = align_up((size_t)cdiv(N, js), 8);
  js = cdiv(N, jchunk);
  hipLaunchKernelGGL(k_rank, dim3(iblocks, js), dim3(256), 0, st, s.keys, N, jchunk, s.rank);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scatter_topk, dim3(cdiv(N, 256)), dim3(256), 0, st, s.boxes, s.keys, s.rank, N, K, s.sboxes,
                     s.sscores, s.sidx, prob, A, dets, flat);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

static int launch_mask(const float4* boxes, int K, float thr, u64* mask, hipStream_t st, bool tf = false) {
  const int cb = cdiv(K, 64);
  if (tf) hipLaunchKernelGGL(k_nms_mask<true>, dim3(cb, cdiv(cb, 4)), dim3(256), 0, st, boxes, K, cb, thr, mask);
  else hipLaunchKernelGGL(k_nms_mask<false>, dim3(cb, cdiv(cb, 4)), dim3(256), 0, st, boxes, K, cb, thr, mask);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" size_t frcnn_nms_workspace_bytes(int max_boxes) {
  if (max_boxes <= 0) return 256;
  return carve(nullptr, max_boxes, max_boxes).bytes;
}

extern "C" int frcnn_nms(const float* dets_d, int k, double thresh, int max_keep, int* keep_d, int* num_keep_d,
                         void* ws, size_t ws_bytes, void* stream) {
  if (!num_keep_d || k < 0 || max_keep < 0) return FRCNN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k == 0 || max_keep == 0) {                       // nms_wrapper.py:18-19: empty in, empty out
    HIP_TRY(hipMemsetAsync(num_keep_d, 0, sizeof(int), st));
    return FRCNN_OK;
  }
  if (!dets_d || !keep_d || !ws) return FRCNN_E_ARG;
  if (k > 16384) return FRCNN_E_UNSUPPORTED;
  SortWs s = carve(ws, k, k);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_dets_key, dim3(cdiv(k, 256)), dim3(256), 0, st, dets_d, k, s.boxes, s.keys, s.rank);
  LAUNCH_CHECK();
  int rc = launch_rank_scatter(s, k, k, nullptr, 0, dets_d, st);
  if (rc) return rc;
  rc = launch_mask(s.sboxes, k, thresh_to_f32(thresh), s.mask, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nms_reduce_keep, dim3(1), dim3(64), 0, st, s.mask, k, cdiv(k, 64), min(max_keep, k), s.sidx,
                     keep_d, num_keep_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

__global__ void k_strided_boxes(const float* __restrict__ src, int k, int stride, float4* __restrict__ dst) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= k) return;
  const float* d = src + (size_t)stride * n;
  dst[n] = make_float4(d[0], d[1], d[2], d[3]);
}

extern "C" int frcnn_nms_sorted(const float* boxes_d, int k, int stride, double thresh, int max_keep, int* keep_d,
                                int* num_keep_d, void* ws, size_t ws_bytes, void* stream) {
  if (!num_keep_d || k < 0 || max_keep < 0 || stride < 4) return FRCNN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k == 0 || max_keep == 0) {
    HIP_TRY(hipMemsetAsync(num_keep_d, 0, sizeof(int), st));
    return FRCNN_OK;
  }
  if (!boxes_d || !keep_d || !ws) return FRCNN_E_ARG;
  if (k > 16384) return FRCNN_E_UNSUPPORTED;
  SortWs s = carve(ws, k, k);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_strided_boxes, dim3(cdiv(k, 256)), dim3(256), 0, st, boxes_d, k, stride, s.sboxes);
  LAUNCH_CHECK();
  int rc = launch_mask(s.sboxes, k, thresh_to_f32(thresh), s.mask, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nms_reduce_keep, dim3(1), dim3(64), 0, st, s.mask, k, cdiv(k, 64), min(max_keep, k),
                     (const int*)nullptr, keep_d, num_keep_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// Drop-in for the reference's host-pointer `_nms` (nms/gpu_nms.hpp:1-2).  Like the original
// (nms_kernel.cu:12-19,100-107,142-143) it allocates per call, blocks, and cannot report errors
// through its signature; unlike the original it leaves num_out = 0 on failure instead of garbage.
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
          int device_id) {
  if (num_out) *num_out = 0;
  if (!keep_out || !num_out || !boxes_host || boxes_num <= 0 || boxes_dim < 4 || boxes_num > 16384) return;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return;
  if (cur != device_id && hipSetDevice(device_id) != hipSuccess) return;       // nms_kernel.cu:80-89
  float* d_boxes = nullptr; int* d_keep = nullptr; void* d_ws = nullptr;
  const size_t wsb = frcnn_nms_workspace_bytes(boxes_num);
  bool ok = hipMalloc(&d_boxes, sizeof(float) * (size_t)boxes_num * boxes_dim) == hipSuccess &&
            hipMalloc(&d_keep, sizeof(int) * ((size_t)boxes_num + 1)) == hipSuccess &&
            hipMalloc(&d_ws, wsb) == hipSuccess;
  if (ok) ok = hipMemcpy(d_boxes, boxes_host, sizeof(float) * (size_t)boxes_num * boxes_dim, hipMemcpyHostToDevice) == hipSuccess;
  // the float threshold is widened exactly: (double)thresh_f reproduces `ovr >= thresh_f`
  if (ok) ok = frcnn_nms_sorted(d_boxes, boxes_num, boxes_dim, (double)nms_overlap_thresh, boxes_num, d_keep + 1, d_keep,
                                d_ws, wsb, nullptr) == FRCNN_OK;
  int n = 0;
  if (ok) ok = hipMemcpy(&n, d_keep, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  if (ok && n > 0) ok = hipMemcpy(keep_out, d_keep + 1, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
  if (ok) *num_out = n;
  (void)hipFree(d_boxes); (void)hipFree(d_keep); (void)hipFree(d_ws);
}

// ------------------------------------------------------------------------------------------------
// proposal layers
// ------------------------------------------------------------------------------------------------
extern "C" size_t frcnn_proposal_workspace_bytes(int H, int W, int A, int pre_nms_topn) {
  const long long N = (long long)H * W * A;
  if (N <= 0) return 256;
  const int K = (pre_nms_topn > 0 && pre_nms_topn < N) ? pre_nms_topn : (int)N;
  return carve(nullptr, (int)N, K).bytes;
}

extern "C" int frcnn_proposal_layer(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h, float im_w,
                                    int H, int W, int A, int feat_stride, const double* base_d, int pre_nms_topn,
                                    int post_nms_topn, double nms_thresh, float* rois_d, float* scores_d, int* num_d,
                                    void* ws, size_t ws_bytes, void* stream) {
  if (!rpn_cls_prob_d || !rpn_bbox_pred_d || !base_d || !rois_d || !scores_d || !num_d || !ws) return FRCNN_E_ARG;
  if (H <= 0 || W <= 0 || A <= 0 || post_nms_topn <= 0) return FRCNN_E_ARG;
  const int N = H * W * A;
  const int K = (pre_nms_topn > 0 && pre_nms_topn < N) ? pre_nms_topn : N;
  if (K > 16384) return FRCNN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  SortWs s = carve(ws, N, K);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_decode_clip_key, dim3(cdiv(N, 256)), dim3(256), 0, st, rpn_cls_prob_d,
                     (const float4*)rpn_bbox_pred_d, base_d, A, W, feat_stride, N, im_w - 1.0f, im_h - 1.0f, s.boxes,
                     s.keys, s.rank);
  LAUNCH_CHECK();
  int rc = launch_rank_scatter(s, N, K, rpn_cls_prob_d, A, nullptr, st);
  if (rc) return rc;
  rc = launch_mask(s.sboxes, K, thresh_to_f32(nms_thresh), s.mask, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nms_reduce_rois, dim3(1), dim3(64), 0, st, s.mask, K, cdiv(K, 64), post_nms_topn, s.sboxes,
                     s.sscores, rois_d, scores_d, num_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_proposal_top_layer(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h,
                                        float im_w, int H, int W, int A, int feat_stride, const double* base_d,
                                        int rpn_top_n, float* rois_d, float* scores_d, void* ws, size_t ws_bytes,
                                        void* stream) {
  if (!rpn_cls_prob_d || !rpn_bbox_pred_d || !base_d || !rois_d || !scores_d || !ws) return FRCNN_E_ARG;
  if (H <= 0 || W <= 0 || A <= 0 || rpn_top_n <= 0) return FRCNN_E_ARG;
  const int N = H * W * A;
  if (N < rpn_top_n) return FRCNN_E_UNSUPPORTED;   // the reference fills randomly here (proposal_top_layer.py:30-33)
  hipStream_t st = (hipStream_t)stream;
  SortWs s = carve(ws, N, rpn_top_n);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  // decode+clip every anchor, then keep the rpn_top_n best: identical values to decode-after-select
  hipLaunchKernelGGL(k_decode_clip_key, dim3(cdiv(N, 256)), dim3(256), 0, st, rpn_cls_prob_d,
                     (const float4*)rpn_bbox_pred_d, base_d, A, W, feat_stride, N, im_w - 1.0f, im_h - 1.0f, s.boxes,
                     s.keys, s.rank);
  LAUNCH_CHECK();
  int rc = launch_rank_scatter(s, N, rpn_top_n, rpn_cls_prob_d, A, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_top_rois, dim3(cdiv(rpn_top_n, 256)), dim3(256), 0, st, s.sboxes, s.sscores, rpn_top_n, rois_d,
                     scores_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------------
// USE_E2E_TF graph (the reference's default, lib/model/config.py:275): tf.image.non_max_suppression semantics
// ------------------------------------------------------------------------------------------------
extern "C" int frcnn_non_max_suppression(const float* boxes_d, const float* scores_d, int k, int max_output_size,
                                         float iou_threshold, int* selected_d, int* num_d, void* ws, size_t ws_bytes,
                                         void* stream) {
  if (!num_d || k < 0 || max_output_size < 0) return FRCNN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k == 0 || max_output_size == 0) {
    HIP_TRY(hipMemsetAsync(num_d, 0, sizeof(int), st));
    return FRCNN_OK;
  }
  if (!boxes_d || !scores_d || !selected_d || !ws) return FRCNN_E_ARG;
  if (k > 64 * NMS_WIDE_WORDS) return FRCNN_E_UNSUPPORTED;
  SortWs s = carve(ws, k, k);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_boxes_scores_key, dim3(cdiv(k, 256)), dim3(256), 0, st, (const float4*)boxes_d, scores_d, k, s.boxes,
                     s.keys, s.rank);
  LAUNCH_CHECK();
  int rc = launch_rank_scatter(s, k, k, nullptr, 0, nullptr, st, scores_d);
  if (rc) return rc;
  rc = launch_mask(s.sboxes, k, iou_threshold, s.mask, st, true);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nms_reduce_keep_wide, dim3(1), dim3(64), 0, st, s.mask, k, cdiv(k, 64), min(max_output_size, k), s.sidx,
                     selected_d, num_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_proposal_layer_tf(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h, float im_w,
                                       int H, int W, int A, int feat_stride, const double* base_d, int post_nms_topn,
                                       float nms_thresh, float* rois_d, float* scores_d, int* num_d, void* ws,
                                       size_t ws_bytes, void* stream) {
  if (!rpn_cls_prob_d || !rpn_bbox_pred_d || !base_d || !rois_d || !scores_d || !num_d || !ws) return FRCNN_E_ARG;
  if (H <= 0 || W <= 0 || A <= 0 || post_nms_topn <= 0) return FRCNN_E_ARG;
  const long long NN = (long long)H * W * A;
  if (NN > 64 * NMS_WIDE_WORDS) return FRCNN_E_UNSUPPORTED;
  const int N = (int)NN;
  hipStream_t st = (hipStream_t)stream;
  SortWs s = carve(ws, N, N);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_decode_clip_key, dim3(cdiv(N, 256)), dim3(256), 0, st, rpn_cls_prob_d,
                     (const float4*)rpn_bbox_pred_d, base_d, A, W, feat_stride, N, im_w - 1.0f, im_h - 1.0f, s.boxes,
                     s.keys, s.rank);
  LAUNCH_CHECK();
  int rc = launch_rank_scatter(s, N, N, rpn_cls_prob_d, A, nullptr, st);
  if (rc) return rc;
  rc = launch_mask(s.sboxes, N, nms_thresh, s.mask, st, true);
  if (rc) return rc;
  hipLaunchKernelGGL(k_nms_reduce_rois_wide, dim3(1), dim3(64), 0, st, s.mask, N, cdiv(N, 64), post_nms_topn, s.sboxes,
                     s.sscores, rois_d, scores_d, num_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------------
// crop_and_resize (TF semantics: SURVEY.md A.2; call sites nets/resnet_v1.py:55-76, network.py:141-157)
// One workgroup per (roi, output row); lanes run along the contiguous NHWC channel axis with
// float4 loads, so every bilinear tap is a coalesced C*4-byte run.  Algorithmic traffic: feature
// map read once (it stays L2-resident) + R*P*P*C*4 written.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 lerp4(const float4 a, const float4 b, float t) {
  return make_float4(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t, a.w + (b.w - a.w) * t);
}
__device__ __forceinline__ float4 max4(const float4 a, const float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// one bilinear sample (all channels of group c4) at crop-grid position (gy, gx) of a P-grid
__device__ __forceinline__ float4 crop_sample(const float4* __restrict__ feat, int H, int W, int C4, int c4, float y1,
                                              float x1, float hs, float ws, int gy, int gx) {
  const float in_y = y1 * (float)(H - 1) + (float)gy * hs;
  const float in_x = x1 * (float)(W - 1) + (float)gx * ws;
  if (in_y < 0 || in_y > (float)(H - 1) || in_x < 0 || in_x > (float)(W - 1)) return make_float4(0, 0, 0, 0);
  const int top = (int)floorf(in_y), bot = (int)ceilf(in_y);
  const int left = (int)floorf(in_x), right = (int)ceilf(in_x);
  const float ly = in_y - (float)top, lx = in_x - (float)left;
  const float4 tl = feat[((size_t)top * W + left) * C4 + c4], tr = feat[((size_t)top * W + right) * C4 + c4];
  const float4 bl = feat[((size_t)bot * W + left) * C4 + c4], br = feat[((size_t)bot * W + right) * C4 + c4];
  return lerp4(lerp4(tl, tr, lx), lerp4(bl, br, lx), ly);
}

template <bool MAX2>
__global__ __launch_bounds__(256) void k_crop_and_resize(const float4* __restrict__ feat, int H, int W, int C4,
                                                         const float* __restrict__ rois, float stride, int pool,
                                                         const float4* __restrict__ bias, int act, float4* __restrict__ out) {
  const int r = blockIdx.x / pool, py = blockIdx.x % pool;
  const float height = ((float)H - 1.0f) * stride, width = ((float)W - 1.0f) * stride;   // network.py:146-147
  const float* roi = rois + 5 * (size_t)r;
  const float x1 = roi[1] / width, y1 = roi[2] / height, x2 = roi[3] / width, y2 = roi[4] / height;
  const int P = MAX2 ? 2 * pool : pool;
  const float hs = (y2 - y1) * (float)(H - 1) / (float)(P - 1);
  const float ws = (x2 - x1) * (float)(W - 1) / (float)(P - 1);
  float4* orow = out + ((size_t)r * pool + py) * pool * C4;
  for (int t = threadIdx.x; t < pool * C4; t += 256) {
    const int px = t / C4, c4 = t % C4;
    float4 v;
    if (MAX2) {
      v = crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py, 2 * px);
      v = max4(v, crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py, 2 * px + 1));
      v = max4(v, crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py + 1, 2 * px));
      v = max4(v, crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py + 1, 2 * px + 1));
    } else {
      v = crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, py, px);
    }
    if (bias) {                       // optional fused epilogue (see frcnn_crop_and_resize_bias_act)
      const float4 b = bias[c4];
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (act == FRCNN_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    orow[(size_t)px * C4 + c4] = v;
  }
}

static int launch_crop(const float* feat_d, int H, int W, int C, const float* rois_d, int R, float feat_stride, int pool,
                       int fuse_max2x2, const float* bias_d, int act, float* out_d, void* stream) {
  if (R == 0) return FRCNN_OK;                          // empty in, empty out (pointers may be null)
  if (!feat_d || !rois_d || !out_d || H < 2 || W < 2 || C <= 0 || R < 0 || pool < 2) return FRCNN_E_ARG;
  if (C % 4) return FRCNN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (fuse_max2x2)
    hipLaunchKernelGGL(k_crop_and_resize<true>, dim3(R * pool), dim3(256), 0, st, (const float4*)feat_d, H, W, C / 4,
                       rois_d, feat_stride, pool, (const float4*)bias_d, act, (float4*)out_d);
  else
    hipLaunchKernelGGL(k_crop_and_resize<false>, dim3(R * pool), dim3(256), 0, st, (const float4*)feat_d, H, W, C / 4,
                       rois_d, feat_stride, pool, (const float4*)bias_d, act, (float4*)out_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_crop_and_resize(const float* feat_d, int H, int W, int C, const float* rois_d, int R,
                                     float feat_stride, int pool, int fuse_max2x2, float* out_d, void* stream) {
  return launch_crop(feat_d, H, W, C, rois_d, R, feat_stride, pool, fuse_max2x2, nullptr, FRCNN_ACT_NONE, out_d, stream);
}

// crop_and_resize followed by (+ bias[c], activation).  Lets a 1x1 convolution that consumes a RoI crop
// run on the H x W feature map instead of on the R x pool x pool crops: conv1x1 and the bilinear crop are
// both linear, so  conv1x1(crop(F)) + b == crop(conv1x1(F)) + b  (the bias must be added AFTER the crop
// because out-of-range samples are zeros).
extern "C" int frcnn_crop_and_resize_bias_act(const float* feat_d, int H, int W, int C, const float* rois_d, int R,
                                              float feat_stride, int pool, const float* bias_d, int act, float* out_d,
                                              void* stream) {
  if (act != FRCNN_ACT_NONE && act != FRCNN_ACT_RELU) return FRCNN_E_ARG;
  return launch_crop(feat_d, H, W, C, rois_d, R, feat_stride, pool, 0, bias_d, act, out_d, stream);
}

// ------------------------------------------------------------------------------------------------
// IoU matrix, float64 (utils/bbox.pyx:15-55)
// ------------------------------------------------------------------------------------------------
__global__ void k_bbox_overlaps(const double* __restrict__ boxes, int n, const double* __restrict__ query, int k,
                                double* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * k) return;
  const int nn = (int)(t / k), kk = (int)(t % k);
  const double* b = boxes + 4 * (size_t)nn;
  const double* q = query + 4 * (size_t)kk;
  const double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
  double o = 0.0;
  const double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
  if (iw > 0) {
    const double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
    if (ih > 0) {
      const double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
      o = iw * ih / ua;
    }
  }
  out[t] = o;
}

extern "C" int frcnn_bbox_overlaps(const double* boxes_d, int n, const double* query_d, int k, double* out_d, void* stream) {
  if (n < 0 || k < 0) return FRCNN_E_ARG;
  if (n == 0 || k == 0) return FRCNN_OK;
  if (!boxes_d || !query_d || !out_d) return FRCNN_E_ARG;
  const long long tot = (long long)n * k;
  hipLaunchKernelGGL(k_bbox_overlaps, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes_d, n,
                     query_d, k, out_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------------
// test-time post-processing (model/test.py:95-102 and :162-180)
// kernel A: one workgroup per foreground class: score filter, rois/scale, decode, final clip,
//           rank sort in LDS, suppression bitmask (global scratch), greedy reduce by wave 0.
// kernel B: one workgroup: exact max_per_image-th score by 4-pass radix select, then an
//           order-preserving compaction (wave ballot + popcount prefix) into the record list.
// ------------------------------------------------------------------------------------------------
// im_detect's box stage alone (model/test.py:95-102): rois/scale, decode for EVERY class, final clip.
__global__ void k_im_detect_boxes(const float* __restrict__ rois, const float* __restrict__ bbox_pred, int R, int C,
                                  double im_scale, float hi_x, float hi_y, float4* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= R * C) return;
  const int r = t / C;
  const float* ro = rois + 5 * (size_t)r;
  const float4 box = make_float4((float)((double)ro[1] / im_scale), (float)((double)ro[2] / im_scale),
                                 (float)((double)ro[3] / im_scale), (float)((double)ro[4] / im_scale));
  float4 b = decode_box(box, ((const float4*)bbox_pred)[t]);
  b.x = rmax(b.x, 0.0f); b.y = rmax(b.y, 0.0f);
  b.z = rmin(b.z, hi_x); b.w = rmin(b.w, hi_y);
  out[t] = b;
}

extern "C" int frcnn_im_detect_boxes(const float* rois_d, const float* bbox_pred_d, int R, int C, double im_scale, int im_h,
                                     int im_w, float* boxes_d, void* stream) {
  if (R < 0 || C <= 0 || !(im_scale > 0)) return FRCNN_E_ARG;
  if (R == 0) return FRCNN_OK;
  if (!rois_d || !bbox_pred_d || !boxes_d) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_im_detect_boxes, dim3(cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, rois_d, bbox_pred_d, R, C,
                     im_scale, (float)(im_w - 1), (float)(im_h - 1), (float4*)boxes_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

#define PC_MAXR 1024
#define PC_WORDS (PC_MAXR / 64)

__global__ __launch_bounds__(256) void k_perclass_nms(const float* __restrict__ prob, const float* __restrict__ bbox_pred,
                                                      const float* __restrict__ rois, const int* __restrict__ num_rois,
                                                      int R, int C, double im_scale, float hi_x, float hi_y, float thr,
                                                      float score_thresh, u64* __restrict__ mask_ws,
                                                      float* __restrict__ cls_dets, int* __restrict__ cls_count) {
  __shared__ u64 keys[PC_MAXR];
  __shared__ float4 boxes[PC_MAXR];
  __shared__ float4 sboxes[PC_MAXR];
  __shared__ float sscores[PC_MAXR];
  __shared__ int nvalid_s;
  const int j = blockIdx.x + 1;                         // class; 0 is background (test.py:162)
  const int tid = threadIdx.x;
  const int nr = num_rois ? min(*num_rois, R) : R;
  if (tid == 0) nvalid_s = 0;
  __syncthreads();
  for (int r = tid; r < R; r += 256) {
    const float s = prob[(size_t)r * C + j];
    const bool valid = (r < nr) && (s > score_thresh);                                   // test.py:163
    float4 b = make_float4(0, 0, 0, 0);
    if (valid) {
      const float* ro = rois + 5 * (size_t)r;
      const float4 box = make_float4((float)((double)ro[1] / im_scale), (float)((double)ro[2] / im_scale),
                                     (float)((double)ro[3] / im_scale), (float)((double)ro[4] / im_scale));  // test.py:95
      const float4 d = *(const float4*)(bbox_pred + (size_t)r * 4 * C + 4 * j);
      b = decode_box(box, d);                                                              // test.py:101
      b.x = rmax(b.x, 0.0f); b.y = rmax(b.y, 0.0f);                                        // test.py:67-77
      b.z = rmin(b.z, hi_x); b.w = rmin(b.w, hi_y);
      atomicAdd(&nvalid_s, 1);
    }
    boxes[r] = b;
    keys[r] = valid ? make_key(s, (u32)r) : (u64)(0xffffffffu - (u32)r);                  // invalid rows sort last
  }
  __syncthreads();
  const int nv = nvalid_s;
  for (int r = tid; r < R; r += 256) {
    const u64 mine = keys[r];
    int rk = 0;
    for (int q = 0; q < R; ++q) rk += (keys[q] > mine) ? 1 : 0;
    if (rk < nv) {
      sboxes[rk] = boxes[r];
      sscores[rk] = prob[(size_t)r * C + j];
    }
  }
  __syncthreads();
  const int words = (nv + 63) / 64;
  u64* mask = mask_ws + (size_t)blockIdx.x * PC_MAXR * PC_WORDS;
  for (int t = tid; t < nv * words; t += 256) {
    const int i = t / words, w = t % words;
    if (w < (i >> 6)) continue;
    const float4 bi = sboxes[i];
    const float ai = box_area(bi);
    const int nj = min(64, nv - w * 64);
    u64 bits = 0;
    for (int q = 0; q < nj; ++q) {
      const int gj = w * 64 + q;
      if (gj > i) {
        const float4 bj = sboxes[gj];
        if (iou_suppresses(bi, ai, bj, box_area(bj), thr)) bits |= (1ull << q);
      }
    }
    mask[(size_t)i * words + w] = bits;
  }
  __threadfence_block();
  __syncthreads();
  if (tid < 64) {
    float* out = cls_dets + (size_t)blockIdx.x * PC_MAXR * 5;
    const int n = (nv > 0) ? greedy_reduce_wave(mask, nv, words, nv, [&](int pos, int i) {
      const float4 b = sboxes[i];
      float* o = out + 5 * (size_t)pos;
      o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = sscores[i];
    }) : 0;
    if (tid == 0) cls_count[blockIdx.x] = n;
  }
}

__global__ __launch_bounds__(1024) void k_final_select(const float* __restrict__ cls_dets, const int* __restrict__ cls_count,
                                                       int nfg, int max_per_image, float* __restrict__ out_dets,
                                                       int* __restrict__ out_count, int max_out) {
  __shared__ int hist[256];
  __shared__ int wave_off[16];
  __shared__ u32 sel_prefix, sel_mask;
  __shared__ int sel_k, total_s, running_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int span = nfg * PC_MAXR;
  if (tid == 0) { total_s = 0; running_s = 0; }
  __syncthreads();
  if (tid < nfg) atomicAdd(&total_s, cls_count[tid]);
  for (int t = tid + 1024; t < nfg; t += 1024) atomicAdd(&total_s, cls_count[t]);
  __syncthreads();
  const int total = total_s;
  u32 cut = 0;                                              // keep everything (test.py:175: only if len > max)
  if (max_per_image > 0 && total > max_per_image) {
    if (tid == 0) { sel_prefix = 0; sel_mask = 0; sel_k = max_per_image; }
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const u32 pf = sel_prefix, mk = sel_mask;
      for (int t = tid; t < span; t += 1024) {
        const int c = t / PC_MAXR, p = t % PC_MAXR;
        if (p < cls_count[c]) {
          const u32 key = sortable_u32(cls_dets[(size_t)t * 5 + 4]);
          if ((key & mk) == pf) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0, b = 255;
        for (; b > 0; --b) {
          if (acc + hist[b] >= sel_k) break;
          acc += hist[b];
        }
        sel_k -= acc;
        sel_prefix |= ((u32)b << shift);
        sel_mask |= (255u << shift);
      }
      __syncthreads();
    }
    cut = sel_prefix;                                       // key of np.sort(scores)[-max_per_image]
  }
  // order-preserving compaction, class-major
  for (int base = 0; base < span; base += 1024) {
    const int t = base + tid;
    bool f = false;
    if (t < span) {
      const int c = t / PC_MAXR, p = t % PC_MAXR;
      f = (p < cls_count[c]) && (sortable_u32(cls_dets[(size_t)t * 5 + 4]) >= cut);     // test.py:178 `>=`
    }
    const u64 bal = __ballot(f);
    if (lane == 0) wave_off[wave] = __popcll(bal);
    __syncthreads();
    int off = running_s;
    for (int w = 0; w < wave; ++w) off += wave_off[w];
    if (f) {
      const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (pos < max_out) {
        const float* d = cls_dets + (size_t)t * 5;
        float* o = out_dets + (size_t)pos * 6;
        o[0] = d[0]; o[1] = d[1]; o[2] = d[2]; o[3] = d[3]; o[4] = d[4];
        o[5] = (float)(t / PC_MAXR + 1);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int s = 0;
      for (int w = 0; w < 16; ++w) s += wave_off[w];
      running_s += s;
    }
    __syncthreads();
  }
  if (tid == 0) *out_count = running_s;
  for (int p = running_s + tid; p < max_out; p += 1024) {
    float* o = out_dets + (size_t)p * 6;
    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0f;
  }
}

extern "C" size_t frcnn_detect_post_workspace_bytes(int R, int C) {
  (void)R;
  const size_t nfg = (size_t)(C > 1 ? C - 1 : 1);
  return align_up(nfg * PC_MAXR * PC_WORDS * sizeof(u64), 256) + align_up(nfg * PC_MAXR * 5 * sizeof(float), 256) +
         align_up(nfg * sizeof(int), 256);
}

extern "C" int frcnn_detect_post(const float* cls_prob_d, const float* bbox_pred_d, const float* rois_d,
                                 const int* num_rois_d, int R, int C, double im_scale, int im_h, int im_w,
                                 double nms_thresh, float score_thresh, int max_per_image, float* out_dets_d,
                                 int* out_count_d, int max_out, void* ws, size_t ws_bytes, void* stream) {
  if (!cls_prob_d || !bbox_pred_d || !rois_d || !out_dets_d || !out_count_d || !ws) return FRCNN_E_ARG;
  if (R <= 0 || C < 2 || max_out <= 0 || !(im_scale > 0)) return FRCNN_E_ARG;
  if (R > PC_MAXR) return FRCNN_E_UNSUPPORTED;
  if (frcnn_detect_post_workspace_bytes(R, C) > ws_bytes) return FRCNN_E_WS;
  const int nfg = C - 1;
  char* p = (char*)ws;
  u64* mask = (u64*)p;
  p += align_up((size_t)nfg * PC_MAXR * PC_WORDS * sizeof(u64), 256);
  float* cls_dets = (float*)p;
  p += align_up((size_t)nfg * PC_MAXR * 5 * sizeof(float), 256);
  int* cls_count = (int*)p;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_perclass_nms, dim3(nfg), dim3(256), 0, st, cls_prob_d, bbox_pred_d, rois_d, num_rois_d, R, C,
                     im_scale, (float)(im_w - 1), (float)(im_h - 1), thresh_to_f32(nms_thresh), score_thresh, mask,
                     cls_dets, cls_count);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_final_select, dim3(1), dim3(1024), 0, st, cls_dets, cls_count, nfg, max_per_image, out_dets_d,
                     out_count_d, max_out);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
