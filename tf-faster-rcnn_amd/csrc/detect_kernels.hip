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
// Batched layout: every stage below runs for B images in ONE launch (one grid dimension = image).  Scratch is carved per
// image (`SortWs`, stride `img` bytes); tensors of image b start at base + b * per-image size.  The reference graph is
// batch-1 (lib/nets/network.py:388); a batch here is B independent images whose launches are shared.
// ------------------------------------------------------------------------------------------------
#define NMS_RULE_CPU FRCNN_NMS_RULE_CPU    // (double)ovr >= thresh   lib/nms/cpu_nms.pyx:65
#define NMS_RULE_GPU FRCNN_NMS_RULE_GPU    // ovr > (float)thresh     lib/nms/nms_kernel.cu:71, lib/nms/py_cpu_nms.py:35
#define NMS_RULE_TF 2                      // tf.image.non_max_suppression (internal: frcnn_non_max_suppression / *_tf)

// ------------------------------------------------------------------------------------------------
// stage 1 of proposal_layer: fused anchor generation + bbox_transform_inv + clip_boxes + sort key
// (proposal_layer.py:27-31, bbox_transform.py:35-81).  One thread per anchor; algorithmic traffic
// 4N (fg score) + 16N (deltas) read, 16N (boxes) + 8N (keys) written.  grid = (ceil(N/256), B).
// ------------------------------------------------------------------------------------------------
__global__ void k_decode_clip_key(const float* __restrict__ prob, const float4* __restrict__ deltas,
                                  const double* __restrict__ base, int A, int W, int stride, int N, float hi_x,
                                  float hi_y, float4* __restrict__ boxes, u64* __restrict__ keys, size_t img) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int b = blockIdx.y;
  prob += (size_t)b * 2 * N;                                               // [B][H*W][2A]: 2N floats per image
  deltas += (size_t)b * N;
  const int a = n % A, pix = n / A;
  const float score = prob[(size_t)pix * 2 * A + A + a];                  // fg = channels [A,2A)  (:27)
  float4 bx = decode_box(anchor_at(base, n, A, W, stride), deltas[n]);
  bx.x = rmax(rmin(bx.x, hi_x), 0.0f);                                    // np.maximum(np.minimum(v, dim-1), 0)
  bx.y = rmax(rmin(bx.y, hi_y), 0.0f);
  bx.z = rmax(rmin(bx.z, hi_x), 0.0f);
  bx.w = rmax(rmin(bx.w, hi_y), 0.0f);
  img_ptr(boxes, img, b)[n] = bx;
  img_ptr(keys, img, b)[n] = make_key(score, (u32)n);
}

// keys for an arbitrary dets [k,5] array (frcnn_nms)
__global__ void k_dets_key(const float* __restrict__ dets, int k, float4* __restrict__ boxes, u64* __restrict__ keys) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= k) return;
  const float* d = dets + 5 * (size_t)n;
  boxes[n] = make_float4(d[0], d[1], d[2], d[3]);
  keys[n] = make_key(d[4], (u32)n);
}

// keys for separate boxes [k,4] / scores [k] arrays (tf.image.non_max_suppression's inputs)
__global__ void k_boxes_scores_key(const float4* __restrict__ in_boxes, const float* __restrict__ in_scores, int k,
                                   float4* __restrict__ boxes, u64* __restrict__ keys) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= k) return;
  boxes[n] = in_boxes[n];
  keys[n] = make_key(in_scores[n], (u32)n);
}

// ------------------------------------------------------------------------------------------------
// stage 2a: the K best of N keys.  `order = scores.ravel().argsort()[::-1]; order = order[:pre_nms_topN]`
// (proposal_layer.py:34-38) needs the K = 6000 best candidates in order, not a sort of all N = 21 546 (63 000 at
// 800x1333, A = 15).  One 1024-thread workgroup per image finds the K-th largest 64-bit key exactly by an 8-pass
// radix select (keys are unique, so exactly K keys are >= it), then compacts those K keys (any order; stage 2b sorts
// them) and zeroes their rank counters.  Reads 8N bytes nine times out of L2; replaces an O(N^2) counting sort
// over all anchors.
// ------------------------------------------------------------------------------------------------
// One radix-select step over a 256-bin histogram: the bin that holds the k-th largest key and the number of keys in the bins
// above it.  Parallel suffix sum (shuffles inside the four 64-bin waves, then the wave totals); a serial walk by one thread
// costs ~10 us of dependent LDS round trips per pass.  Called by EVERY thread of the workgroup (barriers inside); k must not
// exceed the histogram total.
__device__ __forceinline__ void pick_bin(const int* hist, int k, int* wtot, int* out_bin, int* out_above) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int h = 0, x = 0;
  if (tid < 256) {
    h = hist[tid];
    x = h;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_down(x, off, 64);
      if (lane + off < 64) x += y;                       // inclusive suffix sum inside the wave
    }
    if (lane == 0) wtot[wave] = x;
  }
  __syncthreads();
  if (tid < 256) {
    int above = 0;
    for (int w = wave + 1; w < 4; ++w) above += wtot[w];
    const int excl = x - h + above;                      // keys in bins > tid
    if (excl < k && k <= excl + h) { *out_bin = tid; *out_above = excl; }
  }
  __syncthreads();
}

#define SEL_LIST 4096
__global__ __launch_bounds__(1024) void k_select_topk(const u64* __restrict__ keys_all, int N, int K,
                                                      u64* __restrict__ ckeys_all, u32* __restrict__ rank_all, size_t img) {
  // per-wave histograms (row stride 257 words: the 16 rows start in different LDS banks), summed per pass
  __shared__ int whist[16 * 257];
  __shared__ int hist[256];
  __shared__ u64 list[SEL_LIST];             // the keys sharing the selected 16-bit prefix: the last six passes run on these
  __shared__ u64 sel_prefix, sel_mask;
  __shared__ int sel_k, fill, list_n, gather_flag, pick_b, pick_a;
  __shared__ int wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u64* keys = img_ptr(keys_all, img, blockIdx.x);
  u64* ckeys = img_ptr(ckeys_all, img, blockIdx.x);
  u32* rank = img_ptr(rank_all, img, blockIdx.x);
  int* myhist = whist + wave * 257;
  u64 cut = 0;                                                 // K >= N: every key is selected
  if (K < N) {
    if (tid == 0) { sel_prefix = 0; sel_mask = 0; sel_k = K; list_n = -1; }
    for (int shift = 56; shift >= 0; shift -= 8) {
      for (int t = tid; t < 16 * 257; t += 1024) whist[t] = 0;
      __syncthreads();
      const u64 pf = sel_prefix, mk = sel_mask;
      const int ln = list_n;
      if (ln >= 0) {                                           // candidates already gathered into LDS
        for (int t = tid; t < ln; t += 1024) {
          const u64 key = list[t];
          if ((key & mk) == pf) atomicAdd(&myhist[(int)((key >> shift) & 255)], 1);
        }
      } else {
        // 8 independent loads in flight per thread before the (LDS-atomic) histogram updates: a scan of N keys by ONE
        // workgroup is latency-bound otherwise (N / 1024 dependent round trips to L2 per pass)
        for (int base = tid - lane; base < N; base += 8 * 1024) {          // wave-uniform trip count (ballots inside)
          u64 kq[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) kq[q] = (base + q * 1024 + lane < N) ? keys[base + q * 1024 + lane] : 0ull;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const bool on = (base + q * 1024 + lane < N) && (kq[q] & mk) == pf;
            const int bin = (int)((kq[q] >> shift) & 255);
            // the high-byte passes put (nearly) every lane of a wave into ONE bin (probabilities share sign + exponent): 64
            // same-address LDS atomics serialise, so a wave whose active lanes agree sends a single add
            const u64 act = __ballot(on);
            if (!act) continue;
            const int first = __ffsll((long long)act) - 1;
            const int v0 = __shfl(bin, first, 64);
            if (__ballot(on && bin == v0) == act) {
              if (lane == first) atomicAdd(&myhist[v0], __popcll(act));
            } else if (on) {
              atomicAdd(&myhist[bin], 1);
            }
          }
        }
      }
      __syncthreads();
      if (tid < 256) {
        int sum = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) sum += whist[w * 257 + tid];
        hist[tid] = sum;
      }
      __syncthreads();
      pick_bin(hist, sel_k, wtot, &pick_b, &pick_a);
      if (tid == 0) {
        const int bsel = pick_b;
        sel_k -= pick_a;
        sel_prefix |= ((u64)bsel << shift);
        sel_mask |= (255ull << shift);
        // after the second pass: if the selected 16-bit bucket is small, finish on an LDS copy of it
        gather_flag = (shift == 48 && hist[bsel] <= SEL_LIST) ? 1 : 0;
        if (gather_flag) list_n = 0;
      }
      __syncthreads();
      if (gather_flag) {                                       // workgroup-uniform: written once per pass by thread 0
        const u64 pf2 = sel_prefix, mk2 = sel_mask;
        for (int base = tid; base < N; base += 8 * 1024) {
          u64 kq[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) kq[q] = (base + q * 1024 < N) ? keys[base + q * 1024] : 0ull;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (base + q * 1024 < N && (kq[q] & mk2) == pf2) list[atomicAdd(&list_n, 1)] = kq[q];
        }
        __syncthreads();
      }
    }
    cut = sel_prefix;                                          // the K-th largest key itself
  }
  __syncthreads();
  if (tid == 0) fill = 0;
  __syncthreads();
  const int Kc = min(K, N);
  for (int base = 0; base < N; base += 8 * 1024) {
    u64 kq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) kq[q] = (base + q * 1024 + tid < N) ? keys[base + q * 1024 + tid] : 0ull;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool f = (base + q * 1024 + tid < N) && (kq[q] >= cut);
      const u64 bal = __ballot(f);
      int wbase = 0;
      if (lane == 0 && bal) wbase = atomicAdd(&fill, __popcll(bal));
      wbase = __shfl(wbase, 0, 64);
      if (f) {
        const int pos = wbase + __popcll(bal & ((1ull << lane) - 1ull));
        if (pos < Kc) { ckeys[pos] = kq[q]; rank[pos] = 0u; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// stage 2b: rank[i] = #{ j : key[j] > key[i] } among the K selected keys -- the `argsort()[::-1]` order as a counting
// sort over unique keys.  grid = (ceil(K/256), JS, B): each block owns 256 keys and one slice of the j range; the j
// loop index is wave-uniform, so the compared key comes through the scalar cache (s_load) and the body is
// v_cmp_gt_u64 + add-with-carry.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank(const u64* __restrict__ keys_all, int N, int jchunk, u32* __restrict__ rank_all,
                                              size_t img) {
  const u64* keys = img_ptr(keys_all, img, blockIdx.z);
  u32* rank = img_ptr(rank_all, img, blockIdx.z);
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

// stage 3: scatter the K selected candidates into score order.  The score is recovered bit for bit from the key.
__global__ void k_scatter_sorted(const float4* __restrict__ boxes_all, const u64* __restrict__ ckeys_all,
                                 const u32* __restrict__ rank_all, int K, float4* __restrict__ sboxes_all,
                                 float* __restrict__ sscores_all, int* __restrict__ sidx_all, size_t img) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= K) return;
  const int b = blockIdx.y;
  const u64 key = img_ptr(ckeys_all, img, b)[t];
  const u32 r = img_ptr(rank_all, img, b)[t];
  const u32 n = 0xffffffffu - (u32)key;
  img_ptr(sboxes_all, img, b)[r] = img_ptr(boxes_all, img, b)[n];
  img_ptr(sscores_all, img, b)[r] = unsortable_f32((u32)(key >> 32));
  img_ptr(sidx_all, img, b)[r] = (int)n;
}

// ------------------------------------------------------------------------------------------------
// stage 4: suppression bitmask.  Replaces nms_kernel (nms/nms_kernel.cu:34-78) for wave64: one wave
// = one 64-box row tile x one 64-box column tile; lane i builds its own 64-bit word, no ballot
// needed.  Only the upper triangle (column tile >= row tile) is computed and ever read.  grid = (cb, ceil(cb/4), B).
// ------------------------------------------------------------------------------------------------
template <int RULE>
__device__ __forceinline__ float rule_area(const float4 b) { return RULE == NMS_RULE_TF ? box_area_tf(b) : box_area(b); }
template <int RULE>
__device__ __forceinline__ bool rule_suppresses(const float4 a, float aa, const float4 b, float ab, float thr) {
  if (RULE == NMS_RULE_TF) return iou_suppresses_tf(a, aa, b, ab, thr);
  if (RULE == NMS_RULE_GPU) return iou_suppresses_gt(a, aa, b, ab, thr);
  return iou_suppresses(a, aa, b, ab, thr);
}

template <int RULE>
__global__ __launch_bounds__(256) void k_nms_mask(const float4* __restrict__ boxes_all, int K, int cb, float thr,
                                                  u64* __restrict__ mask_all, size_t img) {
  __shared__ float4 cbox[64];
  __shared__ float carea[64];
  const float4* boxes = img_ptr(boxes_all, img, blockIdx.z);
  u64* mask = img_ptr(mask_all, img, blockIdx.z);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = blockIdx.x;                 // column tile
  const int rt = blockIdx.y * 4 + wave;      // row tile of this wave
  if (threadIdx.x < 64) {
    const int j = ct * 64 + threadIdx.x;
    const float4 b = (j < K) ? boxes[j] : make_float4(0, 0, 0, 0);
    cbox[threadIdx.x] = b;
    carea[threadIdx.x] = rule_area<RULE>(b);
  }
  __syncthreads();
  if (rt >= cb || ct < rt) return;
  const int i = rt * 64 + lane;
  if (i >= K) return;
  const float4 bi = boxes[i];
  const float ai = rule_area<RULE>(bi);
  const int nj = min(64, K - ct * 64);
  u64 bits = 0;
  for (int j = 0; j < nj; ++j) {
    const int gj = ct * 64 + j;
    if (gj > i && rule_suppresses<RULE>(bi, ai, cbox[j], carea[j], thr)) bits |= (1ull << j);
  }
  mask[(size_t)i * cb + ct] = bits;
}

// ------------------------------------------------------------------------------------------------
// stage 5: greedy reduce on device (the reference does this on the host after a D2H copy of the
// whole mask, nms_kernel.cu:118-140).
//
// resolve_chunk: the 64 in-chunk decisions from the chunk's diagonal words, scalar bit operations.
// k_nms_reduce<WPL>: one 1024-thread workgroup per image walks the boxes in 64-box chunks.  The 64 mask rows of chunk c+1
// and c+2 (every word >= c+1: 4 rows per wave, WPL words per lane) are loaded into registers WHILE chunk c is being decided, so no
// decision waits for HBM/L2 latency: per chunk the cost is one diagonal resolve on wave 0 plus an LDS atomic-OR of the kept
// rows into the removed-bit words.  The scan stops as soon as max_keep boxes are kept (== truncating the keep list,
// proposal_layer.py:44-45).  K <= 4096 * WPL.
// ------------------------------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only: __syncthreads() would also drain vmcnt and with it the prefetched mask rows.
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// `cur` (wave-uniform): bit b set = box b of the chunk is already removed (or past the end).  d: lane b holds the diagonal
// mask word of box b (bits > b only).  One iteration per KEPT box (not per box): next survivor = lowest clear bit.
__device__ __forceinline__ u64 resolve_chunk(u64 d, u64 cur) {
  u64 kept = 0;
  u64 avail = ~cur;
  while (avail) {
    const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)avail) - 1);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)d, b);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(d >> 32), b);
    kept |= (1ull << b);
    cur |= (((u64)hi << 32) | (u64)lo);
    avail = (b == 63) ? 0ull : (~cur & ((~0ull) << (b + 1)));
  }
  return kept;
}

struct ReduceOut {
  const float4* sboxes; const float* sscores; const int* sidx;     // per-image scratch (stride img)
  float* rois; float* scores; int* keep; int* num;                 // outputs: rois [B][max_keep][5], scores [B][max_keep], keep [max_keep], num [B]
  int write_rois;
};

__device__ __forceinline__ void reduce_emit(const ReduceOut& o, size_t img, int b, int max_keep, int pos, int i) {
  if (o.write_rois) {
    const float4 bx = img_ptr(o.sboxes, img, b)[i];
    float* r = o.rois + 5 * ((size_t)b * max_keep + pos);
    r[0] = (float)b; r[1] = bx.x; r[2] = bx.y; r[3] = bx.z; r[4] = bx.w;     // rois[:, 0] = batch index (proposal_layer.py:49-51)
    o.scores[(size_t)b * max_keep + pos] = img_ptr(o.sscores, img, b)[i];
  } else {
    o.keep[pos] = o.sidx ? img_ptr(o.sidx, img, b)[i] : i;
  }
}

__device__ __forceinline__ void reduce_finish(const ReduceOut& o, int b, int max_keep, int n, int nthreads) {
  if (o.write_rois) {
    for (int p = n + (int)threadIdx.x; p < max_keep; p += nthreads) {
      float* r = o.rois + 5 * ((size_t)b * max_keep + p);
      r[0] = (float)b;                                    // padding rows stay inside their own image (box_ind of the crop)
      r[1] = r[2] = r[3] = r[4] = 0.0f;
      o.scores[(size_t)b * max_keep + p] = 0.0f;
    }
  }
  if (threadIdx.x == 0) o.num[b] = n;
}

#define NMS_KEEP_LDS 4096
template <int WPL>
__global__ __launch_bounds__(1024) void k_nms_reduce(const u64* __restrict__ mask_all, int K, int cb, int max_keep,
                                                     const ReduceOut o, size_t img) {
  __shared__ u64 remv[64 * WPL];
  __shared__ u64 diag[64];
  __shared__ u64 kept_s;
  __shared__ int total_s;
  // The kept boxes' indices wait in LDS and the output rows (box gather -> roi / keep entry: a dependent global load per kept box) are
  // written by all 1024 threads AFTER the scan: inside the scan that load sat on wave 0's in-order vmcnt queue behind the prefetched mask
  // rows, one memory round trip per chunk of the serial loop (650 us for 12 000 -> 2 000 boxes, profiles/r04_ac_train_kernels_by_shape.txt).
  __shared__ int kidx[NMS_KEEP_LDS];
  const bool defer = max_keep <= NMS_KEEP_LDS;
  const int b = blockIdx.x;
  const u64* mask = img_ptr(mask_all, img, b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;       // 16 waves x 4 rows = the 64 rows of a chunk
  for (int w = tid; w < 64 * WPL; w += 1024) remv[w] = 0ull;
  if (tid == 0) total_s = 0;
  int total = 0;                                                        // wave 0's running count (uniform)

  // the rows of chunks c+1 and (WPL <= 3: at most 24 registers per chunk) c+2 are in flight while chunk c is decided -- deciding a chunk
  // takes less than a memory round trip, so with one chunk in flight every chunk of the serial scan waited for its rows (3.2 us each).
  // 1024 threads leave 128 registers each: a third buffer of the 4-word form spills, so 12 000 boxes (188 words) run the 3-word form
  constexpr bool DEEP = WPL <= 3;
  u64 bufA[4][WPL], bufB[4][WPL], bufC[DEEP ? 4 : 1][DEEP ? WPL : 1];
  auto load_chunk = [&](int c, u64 (&buf)[4][WPL]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = c * 64 + wave * 4 + q;
#pragma unroll
      for (int s = 0; s < WPL; ++s) {
        // words >= c only (upper triangle).  Unconditional loads from clamped addresses (out-of-range words are never used:
        // `process` tests w < cb, rows >= K are pre-marked removed): straight-line code lets the compiler wait with a COUNTED
        // vmcnt for this chunk while the next chunk's loads stay in flight.
        const int w = min(c + lane + 64 * s, cb - 1);
        buf[q][s] = mask[(size_t)min(row, K - 1) * cb + w];
      }
    }
  };
  // returns true when the scan is complete
  auto process = [&](int c, u64 (&buf)[4][WPL]) -> bool {
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) diag[wave * 4 + q] = buf[q][0];        // word c of rows wave*4+q
    }
    barrier_lds();                         // diag visible; the previous chunk's ORs into remv are complete
    if (wave == 0) {
      const u64 d = diag[lane];
      u64 cur = remv[c];
      const int nvalid = min(64, K - c * 64);
      if (nvalid < 64) cur |= (~0ull) << nvalid;
      const u32 cur_lo = (u32)__builtin_amdgcn_readfirstlane((u32)cur);
      const u32 cur_hi = (u32)__builtin_amdgcn_readfirstlane((u32)(cur >> 32));
      const u64 kept = resolve_chunk(d, ((u64)cur_hi << 32) | (u64)cur_lo);
      if ((kept >> lane) & 1ull) {
        const int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
        if (pos < max_keep) {
          if (defer) kidx[pos] = c * 64 + lane;
          else reduce_emit(o, img, b, max_keep, pos, c * 64 + lane);
        }
      }
      total += __popcll(kept);
      if (lane == 0) { kept_s = kept; total_s = total; }
    }
    barrier_lds();
    const u64 kept = kept_s;
    if (total_s >= max_keep) return true;
#pragma unroll
    for (int s = 0; s < WPL; ++s) {
      u64 acc = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((kept >> (wave * 4 + q)) & 1ull) acc |= buf[q][s];
      const int w = c + lane + 64 * s;
      if (acc && w > c && w < cb) atomicOr((unsigned long long*)&remv[w], (unsigned long long)acc);
    }
    return false;
  };
  __syncthreads();
  load_chunk(0, bufA);
  if constexpr (DEEP) {
    if (1 < cb) load_chunk(1, bufB);
    for (int c = 0; c < cb; c += 3) {
      if (c + 2 < cb) load_chunk(c + 2, bufC);
      if (process(c, bufA)) break;
      if (c + 1 >= cb) break;
      if (c + 3 < cb) load_chunk(c + 3, bufA);
      if (process(c + 1, bufB)) break;
      if (c + 2 >= cb) break;
      if (c + 4 < cb) load_chunk(c + 4, bufB);
      if (process(c + 2, bufC)) break;
    }
  } else {
    for (int c = 0; c < cb; c += 2) {
      if (c + 1 < cb) load_chunk(c + 1, bufB);
      if (process(c, bufA)) break;
      if (c + 1 >= cb) break;
      if (c + 2 < cb) load_chunk(c + 2, bufA);
      if (process(c + 1, bufB)) break;
    }
  }
  __syncthreads();
  const int n = min(total_s, max_keep);
  if (defer)
    for (int p = tid; p < n; p += 1024) reduce_emit(o, img, b, max_keep, p, kidx[p]);
  reduce_finish(o, b, max_keep, n, 1024);
}

// Same scan for K up to 65536 boxes (tf.image.non_max_suppression sees ALL H*W*A anchors, proposal_layer.py:56-72; also
// pre_nms_topN <= 0 = "all" on a full-size map): one wave per image, the removed-bit words live in LDS (word w is owned by
// lane w & 63, so there is no cross-lane hazard beyond the chunk read), mask rows are fetched on demand.
#define NMS_WIDE_WORDS 1024
__global__ __launch_bounds__(64) void k_nms_reduce_wide(const u64* __restrict__ mask_all, int K, int cb, int max_keep,
                                                        const ReduceOut o, size_t img) {
  __shared__ u64 remv[NMS_WIDE_WORDS];
  const int b = blockIdx.x;
  const u64* mask = img_ptr(mask_all, img, b);
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
    const u64 kept = resolve_chunk(d, cur);
    if ((kept >> lane) & 1ull) {
      const int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
      if (pos < max_keep) reduce_emit(o, img, b, max_keep, pos, i);
    }
    total += __popcll(kept);
    if (total >= max_keep) break;
    for (int w = (c + 1) - ((c + 1) & 63) + lane; w < cb; w += 64) {     // words > c, lane-owned
      if (w <= c) continue;
      u64 acc = remv[w];
      u64 kk = kept;
      while (kk) {
        const int bq = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        acc |= mask[(size_t)(c * 64 + bq) * cb + w];
      }
      remv[w] = acc;
    }
    __syncthreads();          // single wave: orders the LDS writes before the next chunk's broadcast read
  }
  reduce_finish(o, b, max_keep, min(total, max_keep), 64);
}

// Greedy scan of a mask held in LDS (the per-class stage): one wave, mask rows `words` (<= 64) u64 wide; lane w owns the
// removed-bit word w.
template <typename Emit>
__device__ __forceinline__ int greedy_reduce_lds(const u64* mask, int K, int words, Emit emit) {
  const int lane = threadIdx.x & 63;
  u64 remv = 0ull;
  int total = 0;
  for (int c = 0; c < words; ++c) {
    const int i = c * 64 + lane;
    const u64 d = (i < K) ? mask[(size_t)i * words + c] : 0ull;
    const u64 sel = shfl_u64(remv, c);
    const u32 cur_lo = (u32)__builtin_amdgcn_readfirstlane((u32)sel);
    const u32 cur_hi = (u32)__builtin_amdgcn_readfirstlane((u32)(sel >> 32));
    u64 cur = ((u64)cur_hi << 32) | (u64)cur_lo;
    const int nvalid = min(64, K - c * 64);
    if (nvalid < 64) cur |= (~0ull) << nvalid;
    const u64 kept = resolve_chunk(d, cur);
    if ((kept >> lane) & 1ull) emit(total + __popcll(kept & ((1ull << lane) - 1ull)), i);
    total += __popcll(kept);
    if (lane > c && lane < words) {
      u64 acc = remv;
      u64 kk = kept;
      while (kk) {
        const int bq = __ffsll((long long)kk) - 1;
        kk &= kk - 1;
        acc |= mask[(size_t)(c * 64 + bq) * words + lane];
      }
      remv = acc;
    }
  }
  return total;
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

// proposal_top_layer with the indices given by the caller (proposal_top_layer.py:30-33: fewer anchors than rpn_top_n ->
// `npr.choice(length, size=rpn_top_n, replace=True)`, drawn on the host from numpy's global stream like the reference):
// decode + clip exactly those anchors, in the given order.
__global__ void k_decode_gather(const float* __restrict__ prob, const float4* __restrict__ deltas, const double* __restrict__ base,
                                int A, int W, int stride, int N, float hi_x, float hi_y, const int* __restrict__ inds, int n_out,
                                float* __restrict__ rois, float* __restrict__ scores) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_out) return;
  int n = inds[p];
  n = n < 0 ? 0 : (n >= N ? N - 1 : n);
  const int a = n % A, pix = n / A;
  float4 bx = decode_box(anchor_at(base, n, A, W, stride), deltas[n]);
  bx.x = rmax(rmin(bx.x, hi_x), 0.0f);
  bx.y = rmax(rmin(bx.y, hi_y), 0.0f);
  bx.z = rmax(rmin(bx.z, hi_x), 0.0f);
  bx.w = rmax(rmin(bx.w, hi_y), 0.0f);
  float* r = rois + 5 * (size_t)p;
  r[0] = 0.0f; r[1] = bx.x; r[2] = bx.y; r[3] = bx.z; r[4] = bx.w;
  scores[p] = prob[(size_t)pix * 2 * A + A + a];
}
// ------------------------------------------------------------------------------------------------
// workspace carving (per image; a batch uses B consecutive copies)
// ------------------------------------------------------------------------------------------------
struct SortWs {
  float4* boxes; u64* keys; u64* ckeys; u32* rank; float4* sboxes; float* sscores; int* sidx; u64* mask;
  size_t bytes;                 // per image, 256-byte aligned
};
static SortWs carve(void* ws, int N, int K) {
  SortWs s;
  size_t off = 0;
  char* p = (char*)ws;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return p ? (void*)(p + o) : (void*)nullptr; };
  s.boxes = (float4*)take(sizeof(float4) * (size_t)N);
  s.keys = (u64*)take(sizeof(u64) * (size_t)N);
  s.ckeys = (u64*)take(sizeof(u64) * (size_t)K);
  s.rank = (u32*)take(sizeof(u32) * (size_t)K);
  s.sboxes = (float4*)take(sizeof(float4) * (size_t)K);
  s.sscores = (float*)take(sizeof(float) * (size_t)K);
  s.sidx = (int*)take(sizeof(int) * (size_t)K);
  s.mask = (u64*)take(sizeof(u64) * (size_t)K * (size_t)cdiv(K, 64));
  s.bytes = off;
  return s;
}

// keys [N] (already written) -> the K best in score order: sboxes / sscores / sidx [K], for B images
static int launch_topk_sort(const SortWs& s, int N, int K, int B, hipStream_t st) {
  hipLaunchKernelGGL(k_select_topk, dim3(B), dim3(1024), 0, st, s.keys, N, K, s.ckeys, s.rank, s.bytes);
  LAUNCH_CHECK();
  // enough (i-block, j-slice, image) triples to fill 256 CUs x 8 waves/SIMD, at least 256 keys per slice
  const int iblocks = cdiv(K, 256);
  int js = max(1, min(cdiv(K, 256), cdiv(2048, iblocks * B)));
  const int jchunk = (int)align_up((size_t)cdiv(K, js), 8);
  js = cdiv(K, jchunk);
  hipLaunchKernelGGL(k_rank, dim3(iblocks, js, B), dim3(256), 0, st, s.ckeys, K, jchunk, s.rank, s.bytes);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scatter_sorted, dim3(cdiv(K, 256), B), dim3(256), 0, st, s.boxes, s.ckeys, s.rank, K, s.sboxes,
                     s.sscores, s.sidx, s.bytes);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

static int launch_mask(const SortWs& s, int K, int B, float thr, int rule, hipStream_t st) {
  const int cb = cdiv(K, 64);
  const dim3 grid(cb, cdiv(cb, 4), B), block(256);
  if (rule == NMS_RULE_TF) hipLaunchKernelGGL(k_nms_mask<NMS_RULE_TF>, grid, block, 0, st, s.sboxes, K, cb, thr, s.mask, s.bytes);
  else if (rule == NMS_RULE_GPU) hipLaunchKernelGGL(k_nms_mask<NMS_RULE_GPU>, grid, block, 0, st, s.sboxes, K, cb, thr, s.mask, s.bytes);
  else hipLaunchKernelGGL(k_nms_mask<NMS_RULE_CPU>, grid, block, 0, st, s.sboxes, K, cb, thr, s.mask, s.bytes);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

static int launch_reduce(const SortWs& s, int K, int B, int max_keep, const ReduceOut& o, hipStream_t st) {
  const int cb = cdiv(K, 64);
  if (cb <= 128) hipLaunchKernelGGL(k_nms_reduce<2>, dim3(B), dim3(1024), 0, st, s.mask, K, cb, max_keep, o, s.bytes);
  else if (cb <= 192) hipLaunchKernelGGL(k_nms_reduce<3>, dim3(B), dim3(1024), 0, st, s.mask, K, cb, max_keep, o, s.bytes);
  else if (cb <= 256) hipLaunchKernelGGL(k_nms_reduce<4>, dim3(B), dim3(1024), 0, st, s.mask, K, cb, max_keep, o, s.bytes);
  else if (cb <= NMS_WIDE_WORDS) hipLaunchKernelGGL(k_nms_reduce_wide, dim3(B), dim3(64), 0, st, s.mask, K, cb, max_keep, o, s.bytes);
  else return FRCNN_E_UNSUPPORTED;
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// threshold as the kernels compare it (f32): CPU rule `(double)ovr >= thresh` <=> ovr >= smallest f32 >= thresh;
// GPU rule `ovr > (float)thresh` (the reference passes a float, nms/gpu_nms.pyx:30); TF rule `iou > iou_threshold` (float attr).
static float rule_threshold(double thresh, int rule) { return rule == NMS_RULE_CPU ? thresh_to_f32(thresh) : (float)thresh; }
static bool rule_ok(int rule) { return rule == NMS_RULE_CPU || rule == NMS_RULE_GPU; }
#define NMS_MAX_BOXES (64 * NMS_WIDE_WORDS)

extern "C" size_t frcnn_nms_workspace_bytes(int max_boxes) {
  if (max_boxes <= 0) return 256;
  return carve(nullptr, max_boxes, max_boxes).bytes;
}

extern "C" int frcnn_nms_rule(const float* dets_d, int k, double thresh, int rule, int max_keep, int* keep_d, int* num_keep_d,
                              void* ws, size_t ws_bytes, void* stream) {
  if (!num_keep_d || k < 0 || max_keep < 0 || !rule_ok(rule)) return FRCNN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k == 0 || max_keep == 0) {                       // nms_wrapper.py:18-19: empty in, empty out
    HIP_TRY(hipMemsetAsync(num_keep_d, 0, sizeof(int), st));
    return FRCNN_OK;
  }
  if (!dets_d || !keep_d || !ws) return FRCNN_E_ARG;
  if (k > NMS_MAX_BOXES) return FRCNN_E_UNSUPPORTED;
  SortWs s = carve(ws, k, k);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_dets_key, dim3(cdiv(k, 256)), dim3(256), 0, st, dets_d, k, s.boxes, s.keys);
  LAUNCH_CHECK();
  int rc = launch_topk_sort(s, k, k, 1, st);
  if (rc) return rc;
  rc = launch_mask(s, k, 1, rule_threshold(thresh, rule), rule, st);
  if (rc) return rc;
  ReduceOut o = {s.sboxes, s.sscores, s.sidx, nullptr, nullptr, keep_d, num_keep_d, 0};
  return launch_reduce(s, k, 1, min(max_keep, k), o, st);
}

extern "C" int frcnn_nms(const float* dets_d, int k, double thresh, int max_keep, int* keep_d, int* num_keep_d,
                         void* ws, size_t ws_bytes, void* stream) {
  return frcnn_nms_rule(dets_d, k, thresh, NMS_RULE_CPU, max_keep, keep_d, num_keep_d, ws, ws_bytes, stream);
}

__global__ void k_strided_boxes(const float* __restrict__ src, int k, int stride, float4* __restrict__ dst) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= k) return;
  const float* d = src + (size_t)stride * n;
  dst[n] = make_float4(d[0], d[1], d[2], d[3]);
}

extern "C" int frcnn_nms_sorted_rule(const float* boxes_d, int k, int stride, double thresh, int rule, int max_keep, int* keep_d,
                                     int* num_keep_d, void* ws, size_t ws_bytes, void* stream) {
  if (!num_keep_d || k < 0 || max_keep < 0 || stride < 4 || !rule_ok(rule)) return FRCNN_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k == 0 || max_keep == 0) {
    HIP_TRY(hipMemsetAsync(num_keep_d, 0, sizeof(int), st));
    return FRCNN_OK;
  }
  if (!boxes_d || !keep_d || !ws) return FRCNN_E_ARG;
  if (k > NMS_MAX_BOXES) return FRCNN_E_UNSUPPORTED;
  SortWs s = carve(ws, k, k);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_strided_boxes, dim3(cdiv(k, 256)), dim3(256), 0, st, boxes_d, k, stride, s.sboxes);
  LAUNCH_CHECK();
  int rc = launch_mask(s, k, 1, rule_threshold(thresh, rule), rule, st);
  if (rc) return rc;
  ReduceOut o = {s.sboxes, s.sscores, nullptr, nullptr, nullptr, keep_d, num_keep_d, 0};
  return launch_reduce(s, k, 1, min(max_keep, k), o, st);
}

extern "C" int frcnn_nms_sorted(const float* boxes_d, int k, int stride, double thresh, int max_keep, int* keep_d,
                                int* num_keep_d, void* ws, size_t ws_bytes, void* stream) {
  return frcnn_nms_sorted_rule(boxes_d, k, stride, thresh, NMS_RULE_CPU, max_keep, keep_d, num_keep_d, ws, ws_bytes, stream);
}

// Drop-in for the reference's host-pointer `_nms` (nms/gpu_nms.hpp:1-2).  Like the original
// (nms_kernel.cu:12-19,100-107,142-143) it allocates per call, blocks, cannot report errors through its signature and
// applies the CUDA kernel's rule (devIoU > nms_overlap_thresh in f32, nms_kernel.cu:71); unlike the original it leaves
// num_out = 0 on failure instead of garbage.
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim, float nms_overlap_thresh,
          int device_id) {
  if (num_out) *num_out = 0;
  if (!keep_out || !num_out || !boxes_host || boxes_num <= 0 || boxes_dim < 4 || boxes_num > NMS_MAX_BOXES) return;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return;
  if (cur != device_id && hipSetDevice(device_id) != hipSuccess) return;       // nms_kernel.cu:80-89
  float* d_boxes = nullptr; int* d_keep = nullptr; void* d_ws = nullptr;
  const size_t wsb = frcnn_nms_workspace_bytes(boxes_num);
  bool ok = hipMalloc(&d_boxes, sizeof(float) * (size_t)boxes_num * boxes_dim) == hipSuccess &&
            hipMalloc(&d_keep, sizeof(int) * ((size_t)boxes_num + 1)) == hipSuccess &&
            hipMalloc(&d_ws, wsb) == hipSuccess;
  if (ok) ok = hipMemcpy(d_boxes, boxes_host, sizeof(float) * (size_t)boxes_num * boxes_dim, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) ok = frcnn_nms_sorted_rule(d_boxes, boxes_num, boxes_dim, (double)nms_overlap_thresh, NMS_RULE_GPU, boxes_num, d_keep + 1,
                                     d_keep, d_ws, wsb, nullptr) == FRCNN_OK;
  int n = 0;
  if (ok) ok = hipMemcpy(&n, d_keep, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  if (ok && n > 0) ok = hipMemcpy(keep_out, d_keep + 1, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess;
  if (ok) *num_out = n;
  (void)hipFree(d_boxes); (void)hipFree(d_keep); (void)hipFree(d_ws);
}

// The reference declares `_nms` WITHOUT extern "C" (nms/gpu_nms.hpp:1-2; nms/gpu_nms.pyx:13-14 binds it as `cdef extern`), so an
// already-built gpu_nms extension of the reference imports the Itanium-mangled name `_Z4_nmsPiS_PKfiifi`.  The same entry under that
// name: the prebuilt extension relinks against libfrcnn_hip.so unchanged (binary compatible, not only source compatible).
extern "C" void frcnn_nms_cxx_name(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                                   float nms_overlap_thresh, int device_id) __asm__("_Z4_nmsPiS_PKfiifi");
extern "C" void frcnn_nms_cxx_name(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
                                   float nms_overlap_thresh, int device_id) {
  _nms(keep_out, num_out, boxes_host, boxes_num, boxes_dim, nms_overlap_thresh, device_id);
}

// ------------------------------------------------------------------------------------------------
// proposal layers
// ------------------------------------------------------------------------------------------------
extern "C" size_t frcnn_proposal_batched_workspace_bytes(int B, int H, int W, int A, int pre_nms_topn) {
  const long long N = (long long)H * W * A;
  if (N <= 0 || B <= 0) return 256;
  const int K = (pre_nms_topn > 0 && pre_nms_topn < N) ? pre_nms_topn : (int)N;
  return carve(nullptr, (int)N, K).bytes * (size_t)B;
}
extern "C" size_t frcnn_proposal_workspace_bytes(int H, int W, int A, int pre_nms_topn) {
  return frcnn_proposal_batched_workspace_bytes(1, H, W, A, pre_nms_topn);
}

static int launch_decode(const SortWs& s, const float* prob, const float* deltas, int B, float im_h, float im_w, int W, int A,
                         int feat_stride, const double* base_d, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_decode_clip_key, dim3(cdiv(N, 256), B), dim3(256), 0, st, prob, (const float4*)deltas, base_d, A, W,
                     feat_stride, N, im_w - 1.0f, im_h - 1.0f, s.boxes, s.keys, s.bytes);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_proposal_layer_batched(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, int B, float im_h,
                                            float im_w, int H, int W, int A, int feat_stride, const double* base_d,
                                            int pre_nms_topn, int post_nms_topn, double nms_thresh, int rule, float* rois_d,
                                            float* scores_d, int* num_d, void* ws, size_t ws_bytes, void* stream) {
  if (!rpn_cls_prob_d || !rpn_bbox_pred_d || !base_d || !rois_d || !scores_d || !num_d || !ws) return FRCNN_E_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || A <= 0 || post_nms_topn <= 0 || !rule_ok(rule)) return FRCNN_E_ARG;
  const long long NN = (long long)H * W * A;
  if (NN >= (1ll << 28)) return FRCNN_E_UNSUPPORTED;
  const int N = (int)NN;
  const int K = (pre_nms_topn > 0 && pre_nms_topn < N) ? pre_nms_topn : N;
  if (K > NMS_MAX_BOXES) return FRCNN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  SortWs s = carve(ws, N, K);
  if (s.bytes * (size_t)B > ws_bytes) return FRCNN_E_WS;
  int rc = launch_decode(s, rpn_cls_prob_d, rpn_bbox_pred_d, B, im_h, im_w, W, A, feat_stride, base_d, N, st);
  if (rc) return rc;
  rc = launch_topk_sort(s, N, K, B, st);
  if (rc) return rc;
  rc = launch_mask(s, K, B, rule_threshold(nms_thresh, rule), rule, st);
  if (rc) return rc;
  ReduceOut o = {s.sboxes, s.sscores, s.sidx, rois_d, scores_d, nullptr, num_d, 1};
  return launch_reduce(s, K, B, post_nms_topn, o, st);
}

extern "C" int frcnn_proposal_layer(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h, float im_w,
                                    int H, int W, int A, int feat_stride, const double* base_d, int pre_nms_topn,
                                    int post_nms_topn, double nms_thresh, float* rois_d, float* scores_d, int* num_d,
                                    void* ws, size_t ws_bytes, void* stream) {
  return frcnn_proposal_layer_batched(rpn_cls_prob_d, rpn_bbox_pred_d, 1, im_h, im_w, H, W, A, feat_stride, base_d, pre_nms_topn,
                                      post_nms_topn, nms_thresh, NMS_RULE_CPU, rois_d, scores_d, num_d, ws, ws_bytes, stream);
}

extern "C" int frcnn_proposal_top_layer(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h,
                                        float im_w, int H, int W, int A, int feat_stride, const double* base_d,
                                        int rpn_top_n, float* rois_d, float* scores_d, void* ws, size_t ws_bytes,
                                        void* stream) {
  if (!rpn_cls_prob_d || !rpn_bbox_pred_d || !base_d || !rois_d || !scores_d || !ws) return FRCNN_E_ARG;
  if (H <= 0 || W <= 0 || A <= 0 || rpn_top_n <= 0) return FRCNN_E_ARG;
  const int N = H * W * A;
  if (N < rpn_top_n) return FRCNN_E_UNSUPPORTED;   // the reference draws random indices here: frcnn_proposal_top_layer_inds
  hipStream_t st = (hipStream_t)stream;
  SortWs s = carve(ws, N, rpn_top_n);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  // decode+clip every anchor, then keep the rpn_top_n best: identical values to decode-after-select
  int rc = launch_decode(s, rpn_cls_prob_d, rpn_bbox_pred_d, 1, im_h, im_w, W, A, feat_stride, base_d, N, st);
  if (rc) return rc;
  rc = launch_topk_sort(s, N, rpn_top_n, 1, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_top_rois, dim3(cdiv(rpn_top_n, 256)), dim3(256), 0, st, s.sboxes, s.sscores, rpn_top_n, rois_d,
                     scores_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_proposal_top_layer_inds(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h, float im_w,
                                             int H, int W, int A, int feat_stride, const double* base_d, const int* top_inds_d,
                                             int n_inds, float* rois_d, float* scores_d, void* stream) {
  if (!rpn_cls_prob_d || !rpn_bbox_pred_d || !base_d || !rois_d || !scores_d || !top_inds_d) return FRCNN_E_ARG;
  if (H <= 0 || W <= 0 || A <= 0 || n_inds <= 0) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_decode_gather, dim3(cdiv(n_inds, 256)), dim3(256), 0, (hipStream_t)stream, rpn_cls_prob_d,
                     (const float4*)rpn_bbox_pred_d, base_d, A, W, feat_stride, H * W * A, im_w - 1.0f, im_h - 1.0f, top_inds_d, n_inds,
                     rois_d, scores_d);
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
  if (k > NMS_MAX_BOXES) return FRCNN_E_UNSUPPORTED;
  SortWs s = carve(ws, k, k);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipLaunchKernelGGL(k_boxes_scores_key, dim3(cdiv(k, 256)), dim3(256), 0, st, (const float4*)boxes_d, scores_d, k, s.boxes,
                     s.keys);
  LAUNCH_CHECK();
  int rc = launch_topk_sort(s, k, k, 1, st);
  if (rc) return rc;
  rc = launch_mask(s, k, 1, iou_threshold, NMS_RULE_TF, st);
  if (rc) return rc;
  ReduceOut o = {s.sboxes, s.sscores, s.sidx, nullptr, nullptr, selected_d, num_d, 0};
  return launch_reduce(s, k, 1, min(max_output_size, k), o, st);
}

extern "C" int frcnn_proposal_layer_tf_batched(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, int B, float im_h,
                                               float im_w, int H, int W, int A, int feat_stride, const double* base_d,
                                               int post_nms_topn, float nms_thresh, float* rois_d, float* scores_d, int* num_d,
                                               void* ws, size_t ws_bytes, void* stream) {
  if (!rpn_cls_prob_d || !rpn_bbox_pred_d || !base_d || !rois_d || !scores_d || !num_d || !ws) return FRCNN_E_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || A <= 0 || post_nms_topn <= 0) return FRCNN_E_ARG;
  const long long NN = (long long)H * W * A;
  if (NN > NMS_MAX_BOXES) return FRCNN_E_UNSUPPORTED;
  const int N = (int)NN;
  hipStream_t st = (hipStream_t)stream;
  SortWs s = carve(ws, N, N);
  if (s.bytes * (size_t)B > ws_bytes) return FRCNN_E_WS;
  int rc = launch_decode(s, rpn_cls_prob_d, rpn_bbox_pred_d, B, im_h, im_w, W, A, feat_stride, base_d, N, st);
  if (rc) return rc;
  rc = launch_topk_sort(s, N, N, B, st);
  if (rc) return rc;
  rc = launch_mask(s, N, B, nms_thresh, NMS_RULE_TF, st);
  if (rc) return rc;
  ReduceOut o = {s.sboxes, s.sscores, s.sidx, rois_d, scores_d, nullptr, num_d, 1};
  return launch_reduce(s, N, B, post_nms_topn, o, st);
}

extern "C" int frcnn_proposal_layer_tf(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h, float im_w,
                                       int H, int W, int A, int feat_stride, const double* base_d, int post_nms_topn,
                                       float nms_thresh, float* rois_d, float* scores_d, int* num_d, void* ws,
                                       size_t ws_bytes, void* stream) {
  return frcnn_proposal_layer_tf_batched(rpn_cls_prob_d, rpn_bbox_pred_d, 1, im_h, im_w, H, W, A, feat_stride, base_d, post_nms_topn,
                                         nms_thresh, rois_d, scores_d, num_d, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// crop_and_resize (TF semantics: SURVEY.md A.2; call sites nets/resnet_v1.py:55-76, network.py:141-157:
// tf.image.crop_and_resize(bottom, bboxes, tf.to_int32(batch_ids), ...) with batch_ids = rois[:, 0]).
//
// Work split: one workgroup per (roi, output row, channel slab); lanes run along the contiguous NHWC channel axis with
// float4 loads, so every bilinear tap is a coalesced run.  The channel axis is cut into 8 slabs and the slab index is the
// FASTEST block index: workgroup b lands on XCD b % 8, so XCD x only ever touches channel slab x of the feature map --
// 1/8 of it (1.2 MB of a 38x63x1024 map) instead of all of it -- and the taps of every RoI are served by that XCD's own
// 4 MB L2 instead of missing to HBM / Infinity Cache once per XCD.  Algorithmic traffic: feature map read once +
// R*P*P*C*4 written.
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
__global__ __launch_bounds__(256) void k_crop_and_resize(const float4* __restrict__ feat_all, int NIMG, int H, int W, int C4,
                                                         int nslab, const float* __restrict__ rois, float stride, int pool,
                                                         const float4* __restrict__ bias, int act, float4* __restrict__ out) {
  const int slab = blockIdx.x % nslab, rp = blockIdx.x / nslab;
  const int r = rp / pool, py = rp % pool;
  const int SC4 = C4 / nslab, c40 = slab * SC4;                       // this workgroup's float4 channel range
  const float height = ((float)H - 1.0f) * stride, width = ((float)W - 1.0f) * stride;   // network.py:146-147
  const float* roi = rois + 5 * (size_t)r;
  const int img = (int)roi[0];                                         // box_ind = rois[:, 0] (network.py:143)
  float4* orow = out + ((size_t)r * pool + py) * pool * C4;
  if (img < 0 || img >= NIMG) {                                        // not a valid image index: defined (zero) output
    for (int t = threadIdx.x; t < pool * SC4; t += 256) orow[(size_t)(t / SC4) * C4 + c40 + t % SC4] = make_float4(0, 0, 0, 0);
    return;
  }
  const float4* feat = feat_all + (size_t)img * H * W * C4;
  const float x1 = roi[1] / width, y1 = roi[2] / height, x2 = roi[3] / width, y2 = roi[4] / height;
  const int P = MAX2 ? 2 * pool : pool;
  const float hs = (y2 - y1) * (float)(H - 1) / (float)(P - 1);
  const float ws = (x2 - x1) * (float)(W - 1) / (float)(P - 1);
  for (int t = threadIdx.x; t < pool * SC4; t += 256) {
    const int px = t / SC4, c4 = c40 + t % SC4;
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
      const float4 bv = bias[c4];
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (act == FRCNN_ACT_RELU) v = act_relu(v);
    orow[(size_t)px * C4 + c4] = v;
  }
}

// Round 5: one workgroup per (roi, channel slab) -- all pool x pool output pixels of the RoI -- instead of one per (roi, output row, slab).
// The row form launches R * pool * 8 workgroups of 112-448 float4 results each (C = 512: 144 of 256 threads idle) and every one of them
// re-derives the RoI's geometry; the launch was bound by workgroup dispatch (134 400 workgroups in 249 us), not by bytes.  Same
// crop_sample per element: the same bits.  ITEMS results per thread are in flight together (their 4 taps each are independent loads).
template <bool MAX2, int ITEMS>
__global__ __launch_bounds__(256) void k_crop_and_resize_roi(const float4* __restrict__ feat_all, int NIMG, int H, int W, int C4,
                                                             int nslab, const float* __restrict__ rois, float stride, int pool,
                                                             const float4* __restrict__ bias, int act, float4* __restrict__ out) {
  const int slab = blockIdx.x % nslab, r = blockIdx.x / nslab;
  const int SC4 = C4 / nslab, c40 = slab * SC4;
  const float height = ((float)H - 1.0f) * stride, width = ((float)W - 1.0f) * stride;   // network.py:146-147
  const float* roi = rois + 5 * (size_t)r;
  const int img = (int)roi[0];
  float4* obase = out + (size_t)r * pool * pool * C4;
  const int total = pool * pool * SC4;
  if (img < 0 || img >= NIMG) {
    for (int t = threadIdx.x; t < total; t += 256) obase[(size_t)(t / SC4) * C4 + c40 + t % SC4] = make_float4(0, 0, 0, 0);
    return;
  }
  const float4* feat = feat_all + (size_t)img * H * W * C4;
  const float x1 = roi[1] / width, y1 = roi[2] / height, x2 = roi[3] / width, y2 = roi[4] / height;
  const int P = MAX2 ? 2 * pool : pool;
  const float hs = (y2 - y1) * (float)(H - 1) / (float)(P - 1);
  const float ws = (x2 - x1) * (float)(W - 1) / (float)(P - 1);
  for (int t0 = threadIdx.x; t0 < total; t0 += 256 * ITEMS) {
    float4 v[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int t = t0 + k * 256;
      if (t >= total) { v[k] = make_float4(0, 0, 0, 0); continue; }
      const int pix = t / SC4, c4 = c40 + t % SC4;
      const int py = pix / pool, px = pix - py * pool;
      if (MAX2) {
        v[k] = crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py, 2 * px);
        v[k] = max4(v[k], crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py, 2 * px + 1));
        v[k] = max4(v[k], crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py + 1, 2 * px));
        v[k] = max4(v[k], crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, 2 * py + 1, 2 * px + 1));
      } else {
        v[k] = crop_sample(feat, H, W, C4, c4, y1, x1, hs, ws, py, px);
      }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int t = t0 + k * 256;
      if (t >= total) continue;
      const int pix = t / SC4, c4 = c40 + t % SC4;
      float4 o = v[k];
      if (bias) {
        const float4 bv = bias[c4];
        o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
      }
      if (act == FRCNN_ACT_RELU) o = act_relu(o);
      obase[(size_t)pix * C4 + c4] = o;
    }
  }
}

static thread_local int g_crop_slabs = -1;      // tuning (frcnn_set_tuning key 4): -1 automatic, else the channel-slab count
static thread_local int g_crop_form = 0;        // tuning key 5 (A/B runs): 0 = one workgroup per (roi, slab) [round 5]; 1 = per (roi, output row, slab)

static int launch_crop(const float* feat_d, int NIMG, int H, int W, int C, const float* rois_d, int R, float feat_stride, int pool,
                       int fuse_max2x2, const float* bias_d, int act, float* out_d, void* stream) {
  if (R == 0) return FRCNN_OK;                          // empty in, empty out (pointers may be null)
  if (!feat_d || !rois_d || !out_d || NIMG <= 0 || H < 2 || W < 2 || C <= 0 || R < 0 || pool < 2) return FRCNN_E_ARG;
  if (C % 4) return FRCNN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int C4 = C / 4;
  int nslab = (C4 % 8 == 0 && C4 / 8 >= 16) ? 8 : 1;    // >= 256-byte runs per slab
  // the per-RoI form wants >= 512-byte runs (profiles/r05_o_crop_forms.txt: C = 512 on 4 slabs 74 us, on 8 slabs 83, on 16 104; C = 2048 on 8
  // slabs 349 us, on 4 slabs 400): 4 slabs put slab s on XCDs s and s + 4
  if (g_crop_form == 0 && nslab == 8 && C4 / 8 < 32) nslab = 4;
  if (g_crop_slabs > 0 && C4 % g_crop_slabs == 0) nslab = g_crop_slabs;
  const long long blocks = (long long)R * pool * nslab;
  if (blocks >= (1ll << 31)) return FRCNN_E_UNSUPPORTED;
  if (g_crop_form == 0) {
    const unsigned grid = (unsigned)((long long)R * nslab);
    if (fuse_max2x2)
      hipLaunchKernelGGL((k_crop_and_resize_roi<true, 2>), dim3(grid), dim3(256), 0, st, (const float4*)feat_d, NIMG, H, W, C4, nslab, rois_d,
                         feat_stride, pool, (const float4*)bias_d, act, (float4*)out_d);
    else
      hipLaunchKernelGGL((k_crop_and_resize_roi<false, 4>), dim3(grid), dim3(256), 0, st, (const float4*)feat_d, NIMG, H, W, C4, nslab, rois_d,
                         feat_stride, pool, (const float4*)bias_d, act, (float4*)out_d);
    LAUNCH_CHECK();
    return FRCNN_OK;
  }
  if (fuse_max2x2)
    hipLaunchKernelGGL(k_crop_and_resize<true>, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)feat_d, NIMG, H, W, C4, nslab,
                       rois_d, feat_stride, pool, (const float4*)bias_d, act, (float4*)out_d);
  else
    hipLaunchKernelGGL(k_crop_and_resize<false>, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)feat_d, NIMG, H, W, C4, nslab,
                       rois_d, feat_stride, pool, (const float4*)bias_d, act, (float4*)out_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_detect_set_tuning(int key, int value) {
  if (key == 4) { g_crop_slabs = value; return FRCNN_OK; }
  if (key == 5) { g_crop_form = value; return FRCNN_OK; }
  return FRCNN_E_ARG;
}

extern "C" int frcnn_crop_and_resize(const float* feat_d, int H, int W, int C, const float* rois_d, int R,
                                     float feat_stride, int pool, int fuse_max2x2, float* out_d, void* stream) {
  return launch_crop(feat_d, 1, H, W, C, rois_d, R, feat_stride, pool, fuse_max2x2, nullptr, FRCNN_ACT_NONE, out_d, stream);
}

extern "C" int frcnn_crop_and_resize_batched(const float* feat_d, int N, int H, int W, int C, const float* rois_d, int R,
                                             float feat_stride, int pool, int fuse_max2x2, float* out_d, void* stream) {
  return launch_crop(feat_d, N, H, W, C, rois_d, R, feat_stride, pool, fuse_max2x2, nullptr, FRCNN_ACT_NONE, out_d, stream);
}

// crop_and_resize followed by (+ bias[c], activation).  Lets a 1x1 convolution that consumes a RoI crop
// run on the H x W feature map instead of on the R x pool x pool crops: conv1x1 and the bilinear crop are
// both linear, so  conv1x1(crop(F)) + b == crop(conv1x1(F)) + b  (the bias must be added AFTER the crop
// because out-of-range samples are zeros).  feat_d [N,H,W,C], image index = rois[:, 0].
extern "C" int frcnn_crop_and_resize_bias_act(const float* feat_d, int N, int H, int W, int C, const float* rois_d, int R,
                                              float feat_stride, int pool, const float* bias_d, int act, float* out_d,
                                              void* stream) {
  if (act != FRCNN_ACT_NONE && act != FRCNN_ACT_RELU) return FRCNN_E_ARG;
  return launch_crop(feat_d, N, H, W, C, rois_d, R, feat_stride, pool, 0, bias_d, act, out_d, stream);
}

// ------------------------------------------------------------------------------------------------
// box codec as stand-alone entries (model/bbox_transform.py:14-81): the numpy-in / numpy-out functions other reference code
// calls directly (lib/model/test.py:101, layer_utils/*_target_layer.py).  One thread per (row, class) quadruple.
// ------------------------------------------------------------------------------------------------
__global__ void k_bbox_transform_inv(const float4* __restrict__ boxes, const float4* __restrict__ deltas, int N, int k,
                                     float4* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * k) return;
  out[t] = decode_box(boxes[t / k], deltas[t]);                          // deltas[:, 0::4] ... strided per class (:46-49)
}
__global__ void k_clip_boxes(float4* __restrict__ boxes, long long n, float hi_x, float hi_y) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float4 b = boxes[t];
  b.x = rmax(rmin(b.x, hi_x), 0.0f); b.y = rmax(rmin(b.y, hi_y), 0.0f);  // :74-80
  b.z = rmax(rmin(b.z, hi_x), 0.0f); b.w = rmax(rmin(b.w, hi_y), 0.0f);
  boxes[t] = b;
}
// bbox_transform (:14-32) in float32: `x2 - x1 + 1.0` etc. on float32 arrays, np.log in float32
__global__ void k_bbox_transform(const float4* __restrict__ ex, const float4* __restrict__ gt, int N, float4* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const float4 e = ex[t], g = gt[t];
  const float ew = (e.z - e.x) + 1.0f, eh = (e.w - e.y) + 1.0f;
  const float ecx = e.x + 0.5f * ew, ecy = e.y + 0.5f * eh;
  const float gw = (g.z - g.x) + 1.0f, gh = (g.w - g.y) + 1.0f;
  const float gcx = g.x + 0.5f * gw, gcy = g.y + 0.5f * gh;
  out[t] = make_float4((gcx - ecx) / ew, (gcy - ecy) / eh, logf(gw / ew), logf(gh / eh));
}

extern "C" int frcnn_bbox_transform_inv(const float* boxes_d, const float* deltas_d, int N, int k, float* out_d, void* stream) {
  if (N < 0 || k <= 0) return FRCNN_E_ARG;
  if (N == 0) return FRCNN_OK;
  if (!boxes_d || !deltas_d || !out_d) return FRCNN_E_ARG;
  const long long tot = (long long)N * k;
  hipLaunchKernelGGL(k_bbox_transform_inv, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)boxes_d, (const float4*)deltas_d, N, k, (float4*)out_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
extern "C" int frcnn_clip_boxes(float* boxes_d, int N, int k, float im_h, float im_w, void* stream) {
  if (N < 0 || k <= 0) return FRCNN_E_ARG;
  if (N == 0) return FRCNN_OK;
  if (!boxes_d) return FRCNN_E_ARG;
  const long long tot = (long long)N * k;
  hipLaunchKernelGGL(k_clip_boxes, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)boxes_d, tot,
                     im_w - 1.0f, im_h - 1.0f);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
extern "C" int frcnn_bbox_transform(const float* ex_rois_d, const float* gt_rois_d, int N, float* targets_d, void* stream) {
  if (N < 0) return FRCNN_E_ARG;
  if (N == 0) return FRCNN_OK;
  if (!ex_rois_d || !gt_rois_d || !targets_d) return FRCNN_E_ARG;
  hipLaunchKernelGGL(k_bbox_transform, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)ex_rois_d,
                     (const float4*)gt_rois_d, N, (float4*)targets_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
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
// test-time post-processing (model/test.py:95-102 and :162-180), batched over B images
// kernel A: one workgroup per (foreground class, image): score filter, rois/scale, decode, final clip, rank sort,
//           suppression bitmask and greedy reduce -- all in LDS (the mask of 1024 x 16 words reuses the sort scratch).
// kernel B: one workgroup per image: exact max_per_image-th score by 4-pass radix select, then an
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
  float4 b = box;                                                          // TEST.BBOX_REG False (test.py:103-105): np.tile(boxes), no clip
  if (bbox_pred) {
    b = decode_box(box, ((const float4*)bbox_pred)[t]);
    b.x = rmax(b.x, 0.0f); b.y = rmax(b.y, 0.0f);
    b.z = rmin(b.z, hi_x); b.w = rmin(b.w, hi_y);
  }
  out[t] = b;
}

extern "C" int frcnn_im_detect_boxes(const float* rois_d, const float* bbox_pred_d, int R, int C, double im_scale, int im_h,
                                     int im_w, float* boxes_d, void* stream) {
  if (R < 0 || C <= 0 || !(im_scale > 0)) return FRCNN_E_ARG;
  if (R == 0) return FRCNN_OK;
  if (!rois_d || !boxes_d) return FRCNN_E_ARG;          // bbox_pred_d NULL: cfg.TEST.BBOX_REG False
  hipLaunchKernelGGL(k_im_detect_boxes, dim3(cdiv(R * C, 256)), dim3(256), 0, (hipStream_t)stream, rois_d, bbox_pred_d, R, C,
                     im_scale, (float)(im_w - 1), (float)(im_h - 1), (float4*)boxes_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

#define PC_MAXR 1024
#define PC_THREADS 1024          // the 64x64-IoU mask rows of a class are the bulk of the work: spread them over 16 waves

// dynamic LDS: sboxes float4[Rp] | sscores float[Rp] | union { keys u64[Rp] + boxes float4[Rp] ; mask u64[Rp * Rp/64] }
static size_t perclass_lds_bytes(int R) {
  const size_t Rp = (size_t)align_up((size_t)R, 64);
  const size_t sort_b = Rp * (sizeof(u64) + sizeof(float4)), mask_b = Rp * (Rp / 64) * sizeof(u64);
  return Rp * (sizeof(float4) + sizeof(float)) + (sort_b > mask_b ? sort_b : mask_b);
}

template <int RULE>
__global__ __launch_bounds__(PC_THREADS) void k_perclass_nms(const float* __restrict__ prob_all, const float* __restrict__ bbox_pred_all,
                                                      const float* __restrict__ rois_all, const int* __restrict__ num_rois,
                                                      int R, int C, double im_scale, float hi_x, float hi_y, float thr,
                                                      float score_thresh, float* __restrict__ cls_dets_all,
                                                      int* __restrict__ cls_count_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
  __shared__ int nvalid_s;
  const int Rp = (R + 63) & ~63;
  float4* sboxes = (float4*)pc_smem;
  float* sscores = (float*)(sboxes + Rp);
  unsigned char* un = (unsigned char*)(sscores + Rp);
  u64* keys = (u64*)un;
  float4* boxes = (float4*)(keys + Rp);
  u64* mask = (u64*)un;                                 // valid after the sort: keys / boxes are dead by then
  const int j = blockIdx.x + 1;                         // class; 0 is background (test.py:162)
  const int img = blockIdx.y, nfg = C - 1;
  const float* prob = prob_all + (size_t)img * R * C;
  const float* bbox_pred = bbox_pred_all + (size_t)img * R * 4 * C;
  const float* rois = rois_all + (size_t)img * R * 5;
  float* out = cls_dets_all + ((size_t)img * nfg + blockIdx.x) * (size_t)R * 5;
  const int tid = threadIdx.x;
  const int nr = num_rois ? min(num_rois[img], R) : R;
  if (tid == 0) nvalid_s = 0;
  __syncthreads();
  for (int r = tid; r < R; r += PC_THREADS) {
    const float s = prob[(size_t)r * C + j];
    const bool valid = (r < nr) && (s > score_thresh);                                   // test.py:163
    float4 b = make_float4(0, 0, 0, 0);
    if (valid) {
      const float* ro = rois + 5 * (size_t)r;
      const float4 box = make_float4((float)((double)ro[1] / im_scale), (float)((double)ro[2] / im_scale),
                                     (float)((double)ro[3] / im_scale), (float)((double)ro[4] / im_scale));  // test.py:95
      b = box;                                                                             // TEST.BBOX_REG False: test.py:103-105
      if (bbox_pred_all) {
        const float4 d = *(const float4*)(bbox_pred + (size_t)r * 4 * C + 4 * j);
        b = decode_box(box, d);                                                            // test.py:101
        b.x = rmax(b.x, 0.0f); b.y = rmax(b.y, 0.0f);                                      // test.py:67-77
        b.z = rmin(b.z, hi_x); b.w = rmin(b.w, hi_y);
      }
      atomicAdd(&nvalid_s, 1);
    }
    boxes[r] = b;
    keys[r] = valid ? make_key(s, (u32)r) : (u64)(0xffffffffu - (u32)r);                  // invalid rows sort last
  }
  __syncthreads();
  const int nv = nvalid_s;
  for (int r = tid; r < R; r += PC_THREADS) {
    const u64 mine = keys[r];
    int rk = 0;
    for (int q = 0; q < R; ++q) rk += (keys[q] > mine) ? 1 : 0;
    if (rk < nv) {
      sboxes[rk] = boxes[r];
      sscores[rk] = unsortable_f32((u32)(mine >> 32));
    }
  }
  __syncthreads();                                      // keys / boxes dead from here: the mask takes their place
  const int words = (nv + 63) / 64;
  for (int t = tid; t < nv * words; t += PC_THREADS) {
    const int i = t / words, w = t % words;
    u64 bits = 0;
    if (w >= (i >> 6)) {
      const float4 bi = sboxes[i];
      const float ai = box_area(bi);
      const int nj = min(64, nv - w * 64);
      for (int q = 0; q < nj; ++q) {
        const int gj = w * 64 + q;
        if (gj > i) {
          const float4 bj = sboxes[gj];
          if (rule_suppresses<RULE>(bi, ai, bj, box_area(bj), thr)) bits |= (1ull << q);
        }
      }
    }
    mask[(size_t)i * words + w] = bits;
  }
  __syncthreads();
  if (tid < 64) {
    const int n = (nv > 0) ? greedy_reduce_lds(mask, nv, words, [&](int pos, int i) {
      const float4 b = sboxes[i];
      float* o = out + 5 * (size_t)pos;
      o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = sscores[i];
    }) : 0;
    if (tid == 0) cls_count_all[(size_t)img * nfg + blockIdx.x] = n;
  }
}

__global__ __launch_bounds__(1024) void k_final_select(const float* __restrict__ cls_dets_all, const int* __restrict__ cls_count_all,
                                                       int nfg, int R, int max_per_image, float* __restrict__ out_dets_all,
                                                       int* __restrict__ out_count, int max_out, long long out_stride) {
  __shared__ int hist[256];
  __shared__ int wave_off[16];
  __shared__ u32 sel_prefix, sel_mask;
  __shared__ int sel_k, total_s, running_s, pick_b, pick_a;
  __shared__ int wtot[4];
  const int img = blockIdx.x;
  const float* cls_dets = cls_dets_all + (size_t)img * nfg * R * 5;
  const int* cls_count = cls_count_all + (size_t)img * nfg;
  float* out_dets = out_dets_all + (size_t)img * (size_t)out_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int span = nfg * R;
  if (tid == 0) { total_s = 0; running_s = 0; }
  __syncthreads();
  for (int t = tid; t < nfg; t += 1024) atomicAdd(&total_s, cls_count[t]);
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
        const int c = t / R, p = t % R;
        if (p < cls_count[c]) {
          const u32 key = sortable_u32(cls_dets[(size_t)t * 5 + 4]);
          if ((key & mk) == pf) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
      }
      __syncthreads();
      pick_bin(hist, sel_k, wtot, &pick_b, &pick_a);
      if (tid == 0) {
        sel_k -= pick_a;
        sel_prefix |= ((u32)pick_b << shift);
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
      const int c = t / R, p = t % R;
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
        o[5] = (float)(t / R + 1);
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
  if (tid == 0) out_count[img] = running_s;
  for (int p = running_s + tid; p < max_out; p += 1024) {
    float* o = out_dets + (size_t)p * 6;
    o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0f;
  }
}

extern "C" size_t frcnn_detect_post_batched_workspace_bytes(int B, int R, int C) {
  const size_t nfg = (size_t)(C > 1 ? C - 1 : 1), b = (size_t)(B > 0 ? B : 1), r = (size_t)(R > 0 ? R : 1);
  return align_up(b * nfg * r * 5 * sizeof(float), 256) + align_up(b * nfg * sizeof(int), 256);
}
extern "C" size_t frcnn_detect_post_workspace_bytes(int R, int C) { return frcnn_detect_post_batched_workspace_bytes(1, R, C); }

extern "C" int frcnn_detect_post_batched(const float* cls_prob_d, const float* bbox_pred_d, const float* rois_d,
                                         const int* num_rois_d, int B, int R, int C, double im_scale, int im_h, int im_w,
                                         double nms_thresh, int rule, float score_thresh, int max_per_image, float* out_dets_d,
                                         int* out_count_d, int max_out, long long out_stride, void* ws, size_t ws_bytes,
                                         void* stream) {
  if (!cls_prob_d || !rois_d || !out_dets_d || !out_count_d || !ws) return FRCNN_E_ARG;          // bbox_pred_d NULL: TEST.BBOX_REG False
  if (B <= 0 || R <= 0 || C < 2 || max_out <= 0 || !(im_scale > 0) || !rule_ok(rule)) return FRCNN_E_ARG;
  if (out_stride == 0) out_stride = (long long)max_out * 6;
  if (out_stride < (long long)max_out * 6) return FRCNN_E_ARG;
  if (R > PC_MAXR) return FRCNN_E_UNSUPPORTED;
  if (frcnn_detect_post_batched_workspace_bytes(B, R, C) > ws_bytes) return FRCNN_E_WS;
  const int nfg = C - 1;
  char* p = (char*)ws;
  float* cls_dets = (float*)p;
  p += align_up((size_t)B * nfg * R * 5 * sizeof(float), 256);
  int* cls_count = (int*)p;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = perclass_lds_bytes(R);
  const float thr = rule_threshold(nms_thresh, rule);
  auto kern = rule == NMS_RULE_GPU ? k_perclass_nms<NMS_RULE_GPU> : k_perclass_nms<NMS_RULE_CPU>;
  if (lds > 48 * 1024)      // beyond the default dynamic-LDS limit (R > ~512); not a stream operation, legal during capture
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nfg, B), dim3(PC_THREADS), lds, st, cls_prob_d, bbox_pred_d, rois_d, num_rois_d, R, C, im_scale,
                     (float)(im_w - 1), (float)(im_h - 1), thr, score_thresh, cls_dets, cls_count);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_final_select, dim3(B), dim3(1024), 0, st, cls_dets, cls_count, nfg, R, max_per_image, out_dets_d,
                     out_count_d, max_out, out_stride);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_detect_post(const float* cls_prob_d, const float* bbox_pred_d, const float* rois_d,
                                 const int* num_rois_d, int R, int C, double im_scale, int im_h, int im_w,
                                 double nms_thresh, float score_thresh, int max_per_image, float* out_dets_d,
                                 int* out_count_d, int max_out, void* ws, size_t ws_bytes, void* stream) {
  return frcnn_detect_post_batched(cls_prob_d, bbox_pred_d, rois_d, num_rois_d, 1, R, C, im_scale, im_h, im_w, nms_thresh,
                                   FRCNN_NMS_RULE_CPU, score_thresh, max_per_image, out_dets_d, out_count_d, max_out, 0, ws, ws_bytes,
                                   stream);
}
