// Training-side kernels of libfrcnn_hip.so: anchor_target_layer,
// proposal_target_layer, softmax cross-entropy and SmoothL1 (forward + backward).
// Citations relative to /root/reference/lib.  Compiled with -ffp-contract=off.
//
// Random subsampling: the reference draws with numpy's global MT19937 stream
// (anchor_target_layer.py:73-86, proposal_target_layer.py:119-135), which cannot be reproduced on a
// GPU without a host round trip.  Here every candidate gets a counter-based hash key (seed, index)
// and the `k` smallest keys are kept: the same distribution (uniform k-subsets), deterministic for a
// given seed, order-independent.  Everything that is NOT random (labels before sampling, argmax
// assignments, regression targets, weights) is bit-/1e-6-comparable with the oracle, and tests check
// exactly that plus the counts.
#include "common.h"

__device__ __forceinline__ u64 hash_key(u64 seed, u32 idx) {       // splitmix64 finaliser
  u64 z = seed + 0x9E3779B97F4A7C15ull * ((u64)idx + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (z & ~0xFFFFFFFFull) | idx;                               // unique per index; never ~0
}

__device__ __forceinline__ double iou_f64(const float4 b, const float* __restrict__ q) {   // utils/bbox.pyx:28-54
  const double b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
  const double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
  const double box_area = (q2 - q0 + 1) * (q3 - q1 + 1);
  const double iw = (b2 < q2 ? b2 : q2) - (b0 > q0 ? b0 : q0) + 1;
  if (iw > 0) {
    const double ih = (b3 < q3 ? b3 : q3) - (b1 > q1 ? b1 : q1) + 1;
    if (ih > 0) {
      const double ua = (b2 - b0 + 1) * (b3 - b1 + 1) + box_area - iw * ih;
      return iw * ih / ua;
    }
  }
  return 0.0;
}

// bbox_transform for one pair (model/bbox_transform.py:14-32), f32 like the numpy call with f32 inputs
__device__ __forceinline__ float4 encode_box(const float4 ex, const float* __restrict__ gt) {
  const float ew = (ex.z - ex.x) + 1.0f, eh = (ex.w - ex.y) + 1.0f;
  const float ecx = ex.x + 0.5f * ew, ecy = ex.y + 0.5f * eh;
  const float gw = (gt[2] - gt[0]) + 1.0f, gh = (gt[3] - gt[1]) + 1.0f;
  const float gcx = gt[0] + 0.5f * gw, gcy = gt[1] + 0.5f * gh;
  return make_float4((gcx - ecx) / ew, (gcy - ecy) / eh, logf(gw / ew), logf(gh / eh));
}

__device__ __forceinline__ float4 anchor_f32(const double* __restrict__ base, int n, int A, int W, int stride) {
  const int a = n % A, pix = n / A;
  const double sx = (double)((pix % W) * stride), sy = (double)((pix / W) * stride);
  const double* b = base + 4 * a;
  return make_float4((float)(b[0] + sx), (float)(b[1] + sy), (float)(b[2] + sx), (float)(b[3] + sy));
}

// ------------------------------------------------------------------------------------------------
// anchor_target_layer (layer_utils/anchor_target_layer.py:18-138)
// ------------------------------------------------------------------------------------------------
struct AtWs { double* maxov; int* argmax; u64* gtmax; u64* fgkey; u64* bgkey; u32* fgrank; u32* bgrank; int* counts; size_t bytes; };
static AtWs at_carve(void* ws, int N, int G) {
  AtWs s; size_t off = 0; char* p = (char*)ws;
  auto take = [&](size_t b) { size_t o = off; off = align_up(off + b, 256); return p ? (void*)(p + o) : (void*)nullptr; };
  s.maxov = (double*)take(sizeof(double) * (size_t)N);
  s.argmax = (int*)take(sizeof(int) * (size_t)N);
  s.gtmax = (u64*)take(sizeof(u64) * (size_t)(G > 0 ? G : 1));
  s.fgkey = (u64*)take(sizeof(u64) * (size_t)N);
  s.bgkey = (u64*)take(sizeof(u64) * (size_t)N);
  s.fgrank = (u32*)take(sizeof(u32) * (size_t)N);
  s.bgrank = (u32*)take(sizeof(u32) * (size_t)N);
  s.counts = (int*)take(sizeof(int) * 8);
  s.bytes = off;
  return s;
}

// pass 1: inside test, IoU row (f64), argmax/max per anchor, max per gt (atomicMax on the f64 bits:
// overlaps are >= 0 so the bit pattern is monotone)
__global__ void k_at_overlaps(const double* __restrict__ base, int A, int W, int stride, int N, const float* __restrict__ gt,
                              int G, float im_h, float im_w, double* __restrict__ maxov, int* __restrict__ argmax,
                              u64* __restrict__ gtmax) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float4 an = anchor_f32(base, n, A, W, stride);
  const bool inside = (an.x >= 0.0f) && (an.y >= 0.0f) && (an.z < im_w) && (an.w < im_h);       // :31-36
  if (!inside) { maxov[n] = -1.0; argmax[n] = -1; return; }
  double best = -1.0; int bi = 0;
  for (int g = 0; g < G; ++g) {
    const double o = iou_f64(an, gt + 5 * (size_t)g);
    if (o > best) { best = o; bi = g; }                                                          // np.argmax: first maximum
    atomicMax(&gtmax[g], (u64)__double_as_longlong(o));
  }
  maxov[n] = best; argmax[n] = bi;
}

// pass 2: labels before sampling (:57-66) + sampling keys
__global__ void k_at_labels(const double* __restrict__ base, int A, int W, int stride, int N, const float* __restrict__ gt,
                            int G, const double* __restrict__ maxov, const u64* __restrict__ gtmax, double neg_ov,
                            double pos_ov, int clobber, u64 seed, u64* __restrict__ fgkey, u64* __restrict__ bgkey,
                            u32* __restrict__ fgrank, u32* __restrict__ bgrank, int* __restrict__ counts) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int label = -1;
  const double mo = maxov[n];
  if (mo >= 0.0) {                                   // inside the image
    if (!clobber && mo < neg_ov) label = 0;          // :57-60 negatives first, so that positives can clobber them
    const float4 an = anchor_f32(base, n, A, W, stride);
    bool is_gt_argmax = false;                       // `overlaps == gt_max_overlaps` (:55): ALL ties, zeros included
    for (int g = 0; g < G; ++g)
      is_gt_argmax = is_gt_argmax || ((u64)__double_as_longlong(iou_f64(an, gt + 5 * (size_t)g)) == gtmax[g]);
    if (is_gt_argmax) label = 1;
    if (mo >= pos_ov) label = 1;
    if (clobber && mo < neg_ov) label = 0;           // :68-70 TRAIN.RPN_CLOBBER_POSITIVES: negatives last, they clobber positives
  }
  fgkey[n] = (label == 1) ? hash_key(seed, (u32)n) : ~0ull;
  bgkey[n] = (label == 0) ? hash_key(seed ^ 0xA5A5A5A5DEADBEEFull, (u32)n) : ~0ull;
  fgrank[n] = 0u; bgrank[n] = 0u;
  if (label == 1) atomicAdd(&counts[0], 1);
  if (label == 0) atomicAdd(&counts[1], 1);
}

// rank among candidates by key (same scheme as k_rank of the proposal sort)
__global__ __launch_bounds__(256) void k_key_rank(const u64* __restrict__ keys, int N, int jchunk, u32* __restrict__ rank) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const u64 mine = (i < N) ? keys[i] : 0ull;
  if (__all(mine == ~0ull || i >= N)) return;
  const int j0 = blockIdx.y * jchunk, j1 = min(N, j0 + jchunk);
  u32 cnt = 0;
  for (int j = j0; j < j1; ++j) cnt += (keys[j] < mine) ? 1u : 0u;
  if (i < N && mine != ~0ull && cnt) atomicAdd(&rank[i], cnt);
}

// pass 3: subsample, targets, weights, layouts (:72-135)
__global__ void k_at_finish(const double* __restrict__ base, int A, int H, int W, int stride, int N,
                            const float* __restrict__ gt, const int* __restrict__ argmax, const u64* __restrict__ fgkey,
                            const u64* __restrict__ bgkey, const u32* __restrict__ fgrank, const u32* __restrict__ bgrank,
                            const int* __restrict__ counts, int batchsize, int num_fg, int do_sample, double pos_weight,
                            float4 inside_v, float* __restrict__ labels, float4* __restrict__ targets, float4* __restrict__ inside_w,
                            float4* __restrict__ outside_w) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int nfg = counts[0], nbg = counts[1];
  const int fg_keep = do_sample ? min(nfg, num_fg) : nfg;
  const int bg_keep = do_sample ? min(nbg, batchsize - fg_keep) : nbg;
  float label = -1.0f;
  if (fgkey[n] != ~0ull) label = ((int)fgrank[n] < fg_keep) ? 1.0f : -1.0f;
  else if (bgkey[n] != ~0ull) label = ((int)bgrank[n] < bg_keep) ? 0.0f : -1.0f;
  const int a = n % A, pix = n / A, h = pix / W, w = pix % W;
  labels[((size_t)a * H + h) * W + w] = label;                          // (1,H,W,A)->(1,A,H,W)->(1,1,A*H,W)  (:118-119)
  float4 t = make_float4(0, 0, 0, 0), iw = make_float4(0, 0, 0, 0), ow = make_float4(0, 0, 0, 0);
  const int am = argmax[n];
  if (am >= 0) {                                                         // inside anchors get targets (:88-89)
    t = encode_box(anchor_f32(base, n, A, W, stride), gt + 5 * (size_t)am);
    if (label == 1.0f) iw = inside_v;                                    // :91-93 TRAIN.RPN_BBOX_INSIDE_WEIGHTS
    if (label >= 0.0f) {
      // :96-109.  RPN_POSITIVE_WEIGHT < 0: uniform 1 / num_examples; else p / #positives for the positives and (1 - p) / #negatives
      // for the negatives (counts AFTER the subsampling), float64 quotients stored as float32 like the reference's array assignment
      double v = 1.0 / (double)(fg_keep + bg_keep);
      if (pos_weight >= 0.0) v = label == 1.0f ? pos_weight / (double)fg_keep : (1.0 - pos_weight) / (double)bg_keep;
      ow = make_float4((float)v, (float)v, (float)v, (float)v);
    }
  }
  targets[n] = t; inside_w[n] = iw; outside_w[n] = ow;                   // (1,H,W,4A): index n*4 (:123-135)
}

// host-oracle sampling mode: the caller (the py_func-compatible mirror lib/layer_utils/anchor_target_layer.py) drew the
// `disable_inds` of anchor_target_layer.py:72-86 from numpy's global stream exactly like the reference; those anchors lose
// their label here and the candidate counts follow, so k_at_finish (no sampling) sees the reference's label set.
__global__ void k_at_disable(const int* __restrict__ disable, int n_disable, int N, u64* __restrict__ fgkey,
                             u64* __restrict__ bgkey, int* __restrict__ counts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_disable) return;
  const int n = disable[t];
  if (n < 0 || n >= N) return;
  if (fgkey[n] != ~0ull) { fgkey[n] = ~0ull; atomicSub(&counts[0], 1); }
  else if (bgkey[n] != ~0ull) { bgkey[n] = ~0ull; atomicSub(&counts[1], 1); }
}

extern "C" size_t frcnn_anchor_target_workspace_bytes(int H, int W, int A, int max_gt) {
  const long long N = (long long)H * W * A;
  if (N <= 0) return 256;
  return at_carve(nullptr, (int)N, max_gt).bytes;
}

static int launch_key_rank(const u64* keys, int N, u32* rank, hipStream_t st) {
  const int iblocks = cdiv(N, 256);
  int js = max(1, min(cdiv(N, 2048), cdiv(4096, iblocks)));
  const int jchunk = cdiv(N, js);
  js = cdiv(N, jchunk);
  hipLaunchKernelGGL(k_key_rank, dim3(iblocks, js), dim3(256), 0, st, keys, N, jchunk, rank);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

static int anchor_target_impl(const float* gt_boxes_d, int G, float im_h, float im_w, int H, int W, int A, int feat_stride,
                              const double* base_d, int rpn_batchsize, double fg_fraction, double pos_overlap,
                              double neg_overlap, long long seed, const int* disable_d, int n_disable, const double* opts,
                              float* labels_d, float* bbox_targets_d, float* inside_w_d, float* outside_w_d, void* ws,
                              size_t ws_bytes, void* stream) {
  if (!gt_boxes_d || !base_d || !labels_d || !bbox_targets_d || !inside_w_d || !outside_w_d || !ws) return FRCNN_E_ARG;
  if (G <= 0 || H <= 0 || W <= 0 || A <= 0 || rpn_batchsize <= 0 || n_disable < 0 || (n_disable > 0 && !disable_d)) return FRCNN_E_ARG;
  // opts (host, may be null = the reference defaults): {TRAIN.RPN_CLOBBER_POSITIVES, TRAIN.RPN_POSITIVE_WEIGHT, RPN_BBOX_INSIDE_WEIGHTS[4]}
  const int clobber = opts ? (opts[0] != 0.0) : 0;
  const double pos_weight = opts ? opts[1] : -1.0;
  if (pos_weight >= 0.0 && !(pos_weight > 0.0 && pos_weight < 1.0)) return FRCNN_E_ARG;          // the reference asserts 0 < p < 1 (:103-104)
  const float4 inside_v = opts ? make_float4((float)opts[2], (float)opts[3], (float)opts[4], (float)opts[5]) : make_float4(1.f, 1.f, 1.f, 1.f);
  const int N = H * W * A;
  AtWs s = at_carve(ws, N, G);
  if (s.bytes > ws_bytes) return FRCNN_E_WS;
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(s.gtmax, 0, sizeof(u64) * (size_t)G, st));
  HIP_TRY(hipMemsetAsync(s.counts, 0, sizeof(int) * 8, st));
  const int nb = cdiv(N, 256);
  hipLaunchKernelGGL(k_at_overlaps, dim3(nb), dim3(256), 0, st, base_d, A, W, feat_stride, N, gt_boxes_d, G, im_h, im_w,
                     s.maxov, s.argmax, s.gtmax);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_at_labels, dim3(nb), dim3(256), 0, st, base_d, A, W, feat_stride, N, gt_boxes_d, G, s.maxov, s.gtmax,
                     neg_overlap, pos_overlap, clobber, (u64)seed, s.fgkey, s.bgkey, s.fgrank, s.bgrank, s.counts);
  LAUNCH_CHECK();
  const int do_sample = (seed >= 0 && !disable_d) ? 1 : 0;
  if (do_sample) {
    int rc = launch_key_rank(s.fgkey, N, s.fgrank, st);
    if (rc) return rc;
    rc = launch_key_rank(s.bgkey, N, s.bgrank, st);
    if (rc) return rc;
  }
  if (n_disable > 0) {
    hipLaunchKernelGGL(k_at_disable, dim3(cdiv(n_disable, 256)), dim3(256), 0, st, disable_d, n_disable, N, s.fgkey, s.bgkey, s.counts);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_at_finish, dim3(nb), dim3(256), 0, st, base_d, A, H, W, feat_stride, N, gt_boxes_d, s.argmax, s.fgkey,
                     s.bgkey, s.fgrank, s.bgrank, s.counts, rpn_batchsize, (int)(fg_fraction * rpn_batchsize), do_sample, pos_weight,
                     inside_v, labels_d, (float4*)bbox_targets_d, (float4*)inside_w_d, (float4*)outside_w_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_anchor_target_layer(const float* gt_boxes_d, int G, float im_h, float im_w, int H, int W, int A,
                                         int feat_stride, const double* base_d, int rpn_batchsize, double fg_fraction,
                                         double pos_overlap, double neg_overlap, long long seed, const double* opts, float* labels_d,
                                         float* bbox_targets_d, float* inside_w_d, float* outside_w_d, void* ws,
                                         size_t ws_bytes, void* stream) {
  return anchor_target_impl(gt_boxes_d, G, im_h, im_w, H, W, A, feat_stride, base_d, rpn_batchsize, fg_fraction, pos_overlap,
                            neg_overlap, seed, nullptr, 0, opts, labels_d, bbox_targets_d, inside_w_d, outside_w_d, ws, ws_bytes, stream);
}

extern "C" int frcnn_anchor_target_layer_inject(const float* gt_boxes_d, int G, float im_h, float im_w, int H, int W, int A,
                                                int feat_stride, const double* base_d, int rpn_batchsize, double fg_fraction,
                                                double pos_overlap, double neg_overlap, const int* disable_d, int n_disable,
                                                const double* opts, float* labels_d, float* bbox_targets_d, float* inside_w_d,
                                                float* outside_w_d, void* ws, size_t ws_bytes, void* stream) {
  static const int none = 0;      // n_disable == 0 still means "no device sampling": pass a non-null list
  return anchor_target_impl(gt_boxes_d, G, im_h, im_w, H, W, A, feat_stride, base_d, rpn_batchsize, fg_fraction, pos_overlap,
                            neg_overlap, -1, n_disable > 0 ? disable_d : &none, n_disable, opts, labels_d, bbox_targets_d, inside_w_d,
                            outside_w_d, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// proposal_target_layer (layer_utils/proposal_target_layer.py:18-152), one workgroup (N <= 4096 rois)
// ------------------------------------------------------------------------------------------------
#define PT_MAXN 3072
// The candidate set of _sample_rois: the N proposal rows and, with TRAIN.USE_GT (:30-36), the G ground-truth boxes appended as
// rows (0, x1, y1, x2, y2) with score 0.
struct PtCand {
  const float* rois; const float* scores; const float* gt; int N, G, use_gt;
  __device__ __forceinline__ int count() const { return N + (use_gt ? G : 0); }
  __device__ __forceinline__ float4 box(int i) const {
    const float* r = i < N ? rois + 5 * (size_t)i + 1 : gt + 5 * (size_t)(i - N);
    return make_float4(r[0], r[1], r[2], r[3]);
  }
  __device__ __forceinline__ float image(int i) const { return i < N ? rois[5 * (size_t)i] : 0.f; }
  __device__ __forceinline__ float score(int i) const { return i < N ? scores[i] : 0.f; }
};
struct PtNorm { double mean[4], std[4]; float inside[4]; };     // TRAIN.BBOX_NORMALIZE_MEANS / _STDS (float64 like np.array(cfg...)), BBOX_INSIDE_WEIGHTS

// one output row of _sample_rois (:137-152): the sampled roi, its score, label (bg rows clamped to 0, :142), class-expanded
// normalised regression targets and weights (:58-96; outside = inside > 0, :53)
__device__ __forceinline__ void pt_emit_row(int s, int src, bool is_fg, int assigned, const PtCand& cand, int C, const PtNorm& nm,
                                            float* __restrict__ out_rois, float* __restrict__ out_scores,
                                            float* __restrict__ out_labels, float* __restrict__ out_targets,
                                            float* __restrict__ out_inside, float* __restrict__ out_outside) {
  float* orow = out_rois + 5 * (size_t)s;
  float* trow = out_targets + (size_t)s * 4 * C;
  float* irow = out_inside + (size_t)s * 4 * C;
  float* urow = out_outside + (size_t)s * 4 * C;
  for (int c = 0; c < 4 * C; ++c) { trow[c] = 0.f; irow[c] = 0.f; urow[c] = 0.f; }
  if (src < 0) { orow[0] = orow[1] = orow[2] = orow[3] = orow[4] = 0.f; out_scores[s] = 0.f; out_labels[s] = 0.f; return; }
  const float4 b = cand.box(src);
  orow[0] = cand.image(src); orow[1] = b.x; orow[2] = b.y; orow[3] = b.z; orow[4] = b.w;
  out_scores[s] = cand.score(src);
  const float* g = cand.gt + 5 * (size_t)assigned;
  const float label = is_fg ? g[4] : 0.0f;                         // bg labels clamped to 0 (:142)
  out_labels[s] = label;
  if (label > 0.f) {                                               // :58-80, targets :83-96
    const float4 t = encode_box(b, g);
    const float tv[4] = {t.x, t.y, t.z, t.w};
    const int c4 = 4 * (int)label;
    for (int q = 0; q < 4; ++q) {
      trow[c4 + q] = (float)(((double)tv[q] - nm.mean[q]) / nm.std[q]);
      irow[c4 + q] = nm.inside[q];
      urow[c4 + q] = nm.inside[q] > 0.f ? 1.f : 0.f;
    }
  }
}
__global__ __launch_bounds__(1024) void k_proposal_target(const PtCand cand, int Nmax, const int* __restrict__ num_d, int C, int batch,
                                                          int fg_per_image, double fg_thresh, double bg_hi, double bg_lo,
                                                          u64 seed, const PtNorm nm, float* __restrict__ out_rois,
                                                          float* __restrict__ out_scores, float* __restrict__ out_labels,
                                                          float* __restrict__ out_targets, float* __restrict__ out_inside,
                                                          float* __restrict__ out_outside, int* __restrict__ out_counts) {
  __shared__ u64 key[PT_MAXN];
  __shared__ short kind[PT_MAXN];          // 1 fg, 0 bg, -1 neither
  __shared__ short assign[PT_MAXN];
  __shared__ short fg_list[PT_MAXN], bg_list[PT_MAXN];   // candidates in random (key) order
  __shared__ int nfg_s, nbg_s;
  const int tid = threadIdx.x;
  PtCand cd = cand;
  cd.N = num_d ? max(0, min(*num_d, Nmax)) : Nmax;             // device-resident proposal count: no host round trip
  const int N = cd.count(), G = cd.G;
  const float* gt = cd.gt;
  if (tid == 0) { nfg_s = 0; nbg_s = 0; }
  __syncthreads();
  for (int i = tid; i < N; i += 1024) {
    const float4 b = cd.box(i);
    double best = -1.0; int bi = 0;
    for (int g = 0; g < G; ++g) {
      const double o = iou_f64(b, gt + 5 * (size_t)g);
      if (o > best) { best = o; bi = g; }
    }
    short k = -1;
    if (best >= fg_thresh) k = 1;                                   // :111
    else if (best < bg_hi && best >= bg_lo) k = 0;                  // :114-115
    kind[i] = k; assign[i] = (short)bi;
    key[i] = (k >= 0) ? hash_key(seed, (u32)i) : ~0ull;
    if (k == 1) atomicAdd(&nfg_s, 1);
    if (k == 0) atomicAdd(&nbg_s, 1);
  }
  __syncthreads();
  const int nfg = nfg_s, nbg = nbg_s;
  for (int i = tid; i < N; i += 1024) {
    if (kind[i] < 0) continue;
    int rk = 0;
    for (int j = 0; j < N; ++j) rk += (kind[j] == kind[i] && key[j] < key[i]) ? 1 : 0;
    if (kind[i] == 1) fg_list[rk] = (short)i; else bg_list[rk] = (short)i;
  }
  __syncthreads();
  // sampling plan (:119-135)
  int n_fg_out, n_bg_out; bool fg_repl = false, bg_repl = false;
  if (nfg > 0 && nbg > 0) { n_fg_out = min(fg_per_image, nfg); n_bg_out = batch - n_fg_out; bg_repl = nbg < n_bg_out; }
  else if (nfg > 0) { n_fg_out = batch; n_bg_out = 0; fg_repl = nfg < batch; }
  else if (nbg > 0) { n_fg_out = 0; n_bg_out = batch; bg_repl = nbg < batch; }
  else { n_fg_out = 0; n_bg_out = 0; }                               // the reference drops into pdb here (:133-135)
  if (tid == 0) { out_counts[0] = n_fg_out; out_counts[1] = n_bg_out; out_counts[2] = nfg; out_counts[3] = nbg; }
  for (int s = tid; s < batch; s += 1024) {
    int src = -1; bool is_fg = s < n_fg_out;
    if (is_fg) src = fg_list[fg_repl ? (int)((u32)(hash_key(seed ^ 0x1234567ull, (u32)s) >> 32) % (u32)nfg) : s];
    else if (s < n_fg_out + n_bg_out) {
      const int t = s - n_fg_out;
      src = bg_list[bg_repl ? (int)((u32)(hash_key(seed ^ 0x7654321ull, (u32)s) >> 32) % (u32)nbg) : t];
    }
    pt_emit_row(s, src, is_fg, src >= 0 ? (int)assign[src] : 0, cd, C, nm, out_rois, out_scores, out_labels, out_targets, out_inside,
                out_outside);
  }
}

// host-oracle sampling mode: keep_inds [batch] = np.append(fg_inds, bg_inds) as drawn by the caller from numpy's global
// stream (proposal_target_layer.py:119-138), indices into the candidate set (proposals, then the gt boxes with TRAIN.USE_GT); the
// first n_fg rows are foreground.  gt assignment (argmax IoU, f64) is recomputed here for the selected rows.
__global__ __launch_bounds__(256) void k_proposal_target_inject(const PtCand cand, int C, int batch, const int* __restrict__ keep_inds,
                                                                int n_fg, const PtNorm nm, float* __restrict__ out_rois,
                                                                float* __restrict__ out_scores, float* __restrict__ out_labels,
                                                                float* __restrict__ out_targets, float* __restrict__ out_inside,
                                                                float* __restrict__ out_outside) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= batch) return;
  int src = keep_inds[s];
  if (src < 0 || src >= cand.count()) src = -1;
  int bi = 0;
  if (src >= 0) {
    const float4 b = cand.box(src);
    double best = -1.0;
    for (int g = 0; g < cand.G; ++g) {
      const double o = iou_f64(b, cand.gt + 5 * (size_t)g);
      if (o > best) { best = o; bi = g; }
    }
  }
  pt_emit_row(s, src, s < n_fg, bi, cand, C, nm, out_rois, out_scores, out_labels, out_targets, out_inside, out_outside);
}

// opts (host, may be null = the reference defaults): {TRAIN.USE_GT, TRAIN.BBOX_INSIDE_WEIGHTS[4]}
static PtNorm pt_norm(const double* means4, const double* stds4, const double* opts) {
  PtNorm nm;
  for (int q = 0; q < 4; ++q) { nm.mean[q] = means4[q]; nm.std[q] = stds4[q]; nm.inside[q] = opts ? (float)opts[1 + q] : 1.f; }
  return nm;
}

static int proposal_target_impl(const float* rpn_rois_d, const float* rpn_scores_d, int N, const int* num_d, const float* gt_boxes_d,
                                int G, int num_classes, int batch_size, double fg_fraction, double fg_thresh,
                                double bg_thresh_hi, double bg_thresh_lo, const double* means4, const double* stds4,
                                long long seed, const double* opts, float* rois_d, float* roi_scores_d, float* labels_d,
                                float* bbox_targets_d, float* inside_w_d, float* outside_w_d, int* counts_d,
                                void* stream) {
  if (!rpn_rois_d || !rpn_scores_d || !gt_boxes_d || !means4 || !stds4 || !rois_d || !roi_scores_d || !labels_d ||
      !bbox_targets_d || !inside_w_d || !outside_w_d || !counts_d)
    return FRCNN_E_ARG;
  if (N <= 0 || G <= 0 || num_classes < 2 || batch_size <= 0) return FRCNN_E_ARG;
  const int use_gt = opts ? (opts[0] != 0.0) : 0;
  if (N + (use_gt ? G : 0) > PT_MAXN || G > 32767) return FRCNN_E_UNSUPPORTED;
  const int fg_per_image = (int)nearbyint(fg_fraction * batch_size);     // np.round (:40)
  const PtCand cand{rpn_rois_d, rpn_scores_d, gt_boxes_d, N, G, use_gt};
  hipLaunchKernelGGL(k_proposal_target, dim3(1), dim3(1024), 0, (hipStream_t)stream, cand, N, num_d, num_classes, batch_size,
                     fg_per_image, fg_thresh, bg_thresh_hi, bg_thresh_lo, (u64)seed, pt_norm(means4, stds4, opts), rois_d, roi_scores_d,
                     labels_d, bbox_targets_d, inside_w_d, outside_w_d, counts_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

extern "C" int frcnn_proposal_target_layer(const float* rpn_rois_d, const float* rpn_scores_d, int N, const float* gt_boxes_d,
                                           int G, int num_classes, int batch_size, double fg_fraction, double fg_thresh,
                                           double bg_thresh_hi, double bg_thresh_lo, const double* means4, const double* stds4,
                                           long long seed, const double* opts, float* rois_d, float* roi_scores_d, float* labels_d,
                                           float* bbox_targets_d, float* inside_w_d, float* outside_w_d, int* counts_d,
                                           void* stream) {
  return proposal_target_impl(rpn_rois_d, rpn_scores_d, N, nullptr, gt_boxes_d, G, num_classes, batch_size, fg_fraction, fg_thresh,
                              bg_thresh_hi, bg_thresh_lo, means4, stds4, seed, opts, rois_d, roi_scores_d, labels_d, bbox_targets_d,
                              inside_w_d, outside_w_d, counts_d, stream);
}

// Same, with the number of valid proposal rows read on the device (*num_rois_d, the proposal layer's own output): the
// padded [max_rois,5] buffer goes straight in and the training step needs no host synchronisation here.
extern "C" int frcnn_proposal_target_layer_dn(const float* rpn_rois_d, const float* rpn_scores_d, int max_rois, const int* num_rois_d,
                                              const float* gt_boxes_d, int G, int num_classes, int batch_size, double fg_fraction,
                                              double fg_thresh, double bg_thresh_hi, double bg_thresh_lo, const double* means4,
                                              const double* stds4, long long seed, const double* opts, float* rois_d, float* roi_scores_d,
                                              float* labels_d, float* bbox_targets_d, float* inside_w_d, float* outside_w_d,
                                              int* counts_d, void* stream) {
  if (!num_rois_d) return FRCNN_E_ARG;
  return proposal_target_impl(rpn_rois_d, rpn_scores_d, max_rois, num_rois_d, gt_boxes_d, G, num_classes, batch_size, fg_fraction,
                              fg_thresh, bg_thresh_hi, bg_thresh_lo, means4, stds4, seed, opts, rois_d, roi_scores_d, labels_d,
                              bbox_targets_d, inside_w_d, outside_w_d, counts_d, stream);
}

extern "C" int frcnn_proposal_target_layer_inject(const float* rpn_rois_d, const float* rpn_scores_d, int N, const float* gt_boxes_d,
                                                  int G, int num_classes, int batch_size, const int* keep_inds_d, int n_fg,
                                                  const double* means4, const double* stds4, const double* opts, float* rois_d,
                                                  float* roi_scores_d, float* labels_d, float* bbox_targets_d, float* inside_w_d,
                                                  float* outside_w_d, void* stream) {
  if (!rpn_rois_d || !rpn_scores_d || !gt_boxes_d || !means4 || !stds4 || !rois_d || !roi_scores_d || !labels_d ||
      !bbox_targets_d || !inside_w_d || !outside_w_d || !keep_inds_d)
    return FRCNN_E_ARG;
  if (N <= 0 || G <= 0 || num_classes < 2 || batch_size <= 0 || n_fg < 0 || n_fg > batch_size) return FRCNN_E_ARG;
  const PtCand cand{rpn_rois_d, rpn_scores_d, gt_boxes_d, N, G, opts ? (opts[0] != 0.0) : 0};
  hipLaunchKernelGGL(k_proposal_target_inject, dim3(cdiv(batch_size, 256)), dim3(256), 0, (hipStream_t)stream, cand, num_classes,
                     batch_size, keep_inds_d, n_fg, pt_norm(means4, stds4, opts), rois_d, roi_scores_d, labels_d, bbox_targets_d,
                     inside_w_d, outside_w_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// ------------------------------------------------------------------------------------------------
// losses (nets/network.py:264-321): forward value + gradient w.r.t. the network output, one pass.
// Partial sums per workgroup, then a single-thread f64 reduction: deterministic.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* sh) {
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;          // valid in thread 0
}

// sparse softmax CE.  rows: R logits rows of C classes with row stride ld and class stride cs
// (RPN pairs: C = 2, class stride A inside a [HW, 2A] row -> see host wrapper); label < 0 = ignored.
__global__ __launch_bounds__(256) void k_softmax_ce(const float* __restrict__ logits, const float* __restrict__ labels, int R,
                                                    int C, int rpn_A, int rpn_H, int rpn_W, float* __restrict__ dlogits,
                                                    double* __restrict__ partial, int* __restrict__ nsel_partial) {
  __shared__ double sh[4];
  __shared__ int shn[4];
  const int r = blockIdx.x * 256 + threadIdx.x;
  double loss = 0.0; int sel = 0;
  if (r < R) {
    // addressing: plain rows, or RPN element r = (a*H + h)*W + w -> logits[(h*W+w)*2A + {a, A+a}]
    size_t base; int cs;
    if (rpn_A > 0) {
      const int w = r % rpn_W, t = r / rpn_W, h = t % rpn_H, a = t / rpn_H;
      base = ((size_t)h * rpn_W + w) * 2 * rpn_A + a; cs = rpn_A;
    } else { base = (size_t)r * C; cs = 1; }
    const float lab = labels[r];
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, logits[base + (size_t)c * cs]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(logits[base + (size_t)c * cs] - m);
    const bool on = lab >= 0.f;
    sel = on ? 1 : 0;
    const int li = (int)lab;
    if (on) loss = (double)(logf(s) + m - logits[base + (size_t)li * cs]);
    for (int c = 0; c < C; ++c) {      // un-normalised gradient; the 1/nsel factor is applied by k_scale_grad
      const float p = expf(logits[base + (size_t)c * cs] - m) / s;
      dlogits[base + (size_t)c * cs] = on ? (p - (c == li ? 1.f : 0.f)) : 0.f;
    }
  }
  const double bs = block_sum(loss, sh);
  int n = sel;
  for (int o = 32; o; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) shn[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) { partial[blockIdx.x] = bs; nsel_partial[blockIdx.x] = shn[0] + shn[1] + shn[2] + shn[3]; }
}

__global__ void k_ce_finish(const double* __restrict__ partial, const int* __restrict__ nsel_partial, int nblocks,
                            float* __restrict__ loss_out, float* __restrict__ inv_n) {
  double s = 0.0; long long n = 0;
  for (int i = 0; i < nblocks; ++i) { s += partial[i]; n += nsel_partial[i]; }
  *loss_out = n > 0 ? (float)(s / (double)n) : 0.f;          // reduce_mean over the selected rows
  *inv_n = n > 0 ? (float)(1.0 / (double)n) : 0.f;
}

__global__ void k_scale_grad(float* __restrict__ g, long long n, const float* __restrict__ scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] *= *scale;
}

extern "C" size_t frcnn_loss_workspace_bytes(long long elements) {
  const long long nb = (elements + 255) / 256 + 1;
  return (size_t)(align_up((size_t)nb * sizeof(double), 256) + align_up((size_t)nb * sizeof(int), 256) + 256);
}

extern "C" int frcnn_softmax_ce_loss(const float* logits_d, const float* labels_d, int R, int C, int rpn_A, int rpn_H, int rpn_W,
                                     float* loss_d, float* dlogits_d, void* ws, size_t ws_bytes, void* stream) {
  if (!logits_d || !labels_d || !loss_d || !dlogits_d || !ws || R <= 0 || C < 2) return FRCNN_E_ARG;
  if (rpn_A > 0 && (C != 2 || (long long)rpn_A * rpn_H * rpn_W != R)) return FRCNN_E_ARG;
  if (frcnn_loss_workspace_bytes(R) > ws_bytes) return FRCNN_E_WS;
  const int nb = cdiv(R, 256);
  char* p = (char*)ws;
  double* partial = (double*)p; p += align_up((size_t)(nb + 1) * sizeof(double), 256);
  int* nsel = (int*)p; p += align_up((size_t)(nb + 1) * sizeof(int), 256);
  float* inv_n = (float*)p;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_softmax_ce, dim3(nb), dim3(256), 0, st, logits_d, labels_d, R, C, rpn_A, rpn_H, rpn_W, dlogits_d, partial, nsel);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_ce_finish, dim3(1), dim3(1), 0, st, partial, nsel, nb, loss_d, inv_n);
  LAUNCH_CHECK();
  const long long tot = (long long)R * C;
  hipLaunchKernelGGL(k_scale_grad, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dlogits_d, tot, inv_n);
  LAUNCH_CHECK();
  return FRCNN_OK;
}

// SmoothL1 (network.py:264-277): loss = scale * sum( out_w * f(in_w * (pred - tgt)) ),
// f(d) = 0.5*sigma^2*d^2 if |d| < 1/sigma^2 else |d| - 0.5/sigma^2;  scale = 1/(mean divisor).
__global__ __launch_bounds__(256) void k_smooth_l1(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                   const float* __restrict__ in_w, const float* __restrict__ out_w,
                                                   long long n, float sigma2, float scale, float* __restrict__ dpred,
                                                   double* __restrict__ partial) {
  __shared__ double sh[4];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  double l = 0.0;
  if (i < n) {
    const float d = in_w[i] * (pred[i] - tgt[i]);
    const float ad = fabsf(d);
    const bool quad = ad < 1.0f / sigma2;
    const float f = quad ? (d * d) * (sigma2 / 2.0f) : ad - (0.5f / sigma2);
    l = (double)(out_w[i] * f);
    const float df = quad ? sigma2 * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    dpred[i] = scale * out_w[i] * in_w[i] * df;
  }
  const double bs = block_sum(l, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = bs;
}

__global__ void k_sum_finish(const double* __restrict__ partial, int nblocks, double scale, float* __restrict__ loss_out) {
  double s = 0.0;
  for (int i = 0; i < nblocks; ++i) s += partial[i];
  *loss_out = (float)(s * scale);
}

extern "C" int frcnn_smooth_l1_loss(const float* pred_d, const float* targets_d, const float* inside_w_d,
                                    const float* outside_w_d, long long n, float sigma, float mean_divisor, float* loss_d,
                                    float* dpred_d, void* ws, size_t ws_bytes, void* stream) {
  if (!pred_d || !targets_d || !inside_w_d || !outside_w_d || !loss_d || !dpred_d || !ws || n <= 0 || !(sigma > 0) ||
      !(mean_divisor > 0))
    return FRCNN_E_ARG;
  if (frcnn_loss_workspace_bytes(n) > ws_bytes) return FRCNN_E_WS;
  const int nb = (int)((n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_smooth_l1, dim3(nb), dim3(256), 0, st, pred_d, targets_d, inside_w_d, outside_w_d, n, sigma * sigma,
                     1.0f / mean_divisor, dpred_d, (double*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(1), 0, st, (const double*)ws, nb, 1.0 / (double)mean_divisor, loss_d);
  LAUNCH_CHECK();
  return FRCNN_OK;
}
