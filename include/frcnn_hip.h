/* frcnn_hip.h -- C ABI of libfrcnn_hip.so: the MI355X (gfx950) hot path of Faster R-CNN behind the
 * interfaces of endernewton/tf-faster-rcnn.  Plain pointers and sizes only; no torch / TF types.
 *
 * Conventions:
 *   - every pointer named *_d / documented "dev" is a DEVICE pointer (HBM); everything else is host.
 *   - `stream` is a hipStream_t passed as void*; the library never synchronises, never allocates,
 *     never calls hipSetDevice and never prints.  Scratch memory comes from the caller
 *     (`ws`, sized by the matching *_workspace_bytes query).  Distinct streams are thread-safe.
 *   - return value: 0 = ok, FRCNN_E_ARG (-1) bad argument, FRCNN_E_WS (-2) workspace too small,
 *     FRCNN_E_UNSUPPORTED (-3) shape outside the kernels' limits, -(1000 + hipError_t) HIP failure.
 *   - tensors are NHWC float32, exactly the layouts the reference's TF graph uses.
 * Reference citations are relative to /root/reference.
 */
#ifndef FRCNN_HIP_H_
#define FRCNN_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK 0
#define FRCNN_E_ARG (-1)
#define FRCNN_E_WS (-2)
#define FRCNN_E_UNSUPPORTED (-3)
#define FRCNN_E_HIP(e) (-(1000 + (int)(e)))

/* NMS suppression rule.  CPU: suppress iff (double)ovr >= thresh (lib/nms/cpu_nms.pyx:65, lib/nms/cpu_nms.c:2239-2241).
 * GPU: suppress iff ovr > (float)thresh (lib/nms/nms_kernel.cu:71; same test as lib/nms/py_cpu_nms.py:35).  Both use the
 * +1 pixel convention and f32 IoU arithmetic with one rounding per operation. */
#define FRCNN_NMS_RULE_CPU 0
#define FRCNN_NMS_RULE_GPU 1

#define FRCNN_ACT_NONE 0
#define FRCNN_ACT_RELU 1
#define FRCNN_ACT_RELU6 2

/* ---- library ------------------------------------------------------------------------------- */
/* Bumps when an exported signature changes.  2: batched detection stages, NMS rule.  3: `opts` argument of the target-layer entries,
 * per-call cfg / terms of frcnn_gemm_x3 (frcnn_gemm_x3_set_* removed), frcnn_gemm_h2 + operand planes.  4: frcnn_gemm_h2_mean replaces
 * frcnn_conv1x1_mean (whose reduction order depended on the batch slot).  5: the *_masked entries of the training step's data-gradient
 * chain (a library without them must not be loaded by a caller that expects them).  6: frcnn_crop_and_resize_bwd_plan, the
 * reference-mangled `_Z4_nmsPiS_PKfiifi` export beside `_nms`, frcnn_gemm_h2 configuration ids 40 / 41.  A caller compiled against
 * this header compares frcnn_abi_version() with FRCNN_ABI_VERSION before its first call (the ctypes binding does, on load). */
#define FRCNN_ABI_VERSION 6
int frcnn_abi_version(void);
const char* frcnn_build_info(void);          /* "gfx950 ..." */

/* ---- NMS: replaces lib/nms/gpu_nms.hpp:1-2 (`_nms`), lib/nms/cpu_nms.pyx:17-68 -------------- */
/* Source-compatible with the reference's extern "C++" `_nms` (lib/nms/gpu_nms.hpp:1-2, driver
 * lib/nms/nms_kernel.cu:91-144): HOST pointers, blocking, boxes [n, boxes_dim>=4] already sorted
 * by descending score, keep_out capacity n.  Applies the CUDA kernel's rule FRCNN_NMS_RULE_GPU (`devIoU > thresh`,
 * nms_kernel.cu:71), so it returns what the reference `_nms` returns, also at IoU == thresh.  Allocates/frees its own
 * device scratch like the original did.  n <= 65536. */
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
          float nms_overlap_thresh, int device_id);

size_t frcnn_nms_workspace_bytes(int max_boxes);
/* cpu_nms(dets, thresh) on device: dets_d [k,5] f32 (x1,y1,x2,y2,score) in ANY order.
 * keep_d [max_keep] int32 receives kept ORIGINAL indices in descending-score order
 * (ties: lower index first), num_keep_d the count (<= max_keep; the greedy scan stops there, which
 * equals truncating the full keep list, lib/layer_utils/proposal_layer.py:44-45).
 * frcnn_nms: rule FRCNN_NMS_RULE_CPU -- suppress iff (double)ovr >= thresh, exactly as lib/nms/cpu_nms.c:2239-2241.
 * frcnn_nms_rule: the rule is an argument (FRCNN_NMS_RULE_GPU = what gpu_nms / the CUDA kernel compute).  k <= 65536. */
int frcnn_nms(const float* dets_d, int k, double thresh, int max_keep, int* keep_d, int* num_keep_d,
              void* ws, size_t ws_bytes, void* stream);
int frcnn_nms_rule(const float* dets_d, int k, double thresh, int rule, int max_keep, int* keep_d, int* num_keep_d,
                   void* ws, size_t ws_bytes, void* stream);
/* Same, for input already sorted by descending score (boxes_d rows of `stride` floats, the first
 * four are x1,y1,x2,y2): the device-pointer form of `_nms`. */
int frcnn_nms_sorted(const float* boxes_d, int k, int stride, double thresh, int max_keep, int* keep_d,
                     int* num_keep_d, void* ws, size_t ws_bytes, void* stream);
int frcnn_nms_sorted_rule(const float* boxes_d, int k, int stride, double thresh, int rule, int max_keep, int* keep_d,
                          int* num_keep_d, void* ws, size_t ws_bytes, void* stream);

/* ---- anchors: replaces lib/layer_utils/generate_anchors.py:41-105, snippets.py:14-30 --------- */
/* HOST: base anchors, float64 [n_ratios*n_scales, 4], ratio-major; np.round half-to-even. */
int frcnn_generate_anchors(int base_size, const double* ratios, int n_ratios, const double* scales,
                           int n_scales, double* out_base);
/* DEVICE: all shifted anchors f32 [H*W*A,4], index (y*W+x)*A+a.  base_d = device copy of the
 * float64 base anchors [A,4]. */
int frcnn_generate_anchors_pre(int height, int width, int feat_stride, const double* base_d, int A,
                               float* anchors_d, void* stream);

/* ---- proposal layers: replaces lib/layer_utils/proposal_layer.py:16-53, proposal_top_layer.py:17-55
 * (the tf.py_func seams at lib/nets/network.py:100-103,123-126) ------------------------------ */
size_t frcnn_proposal_workspace_bytes(int H, int W, int A, int pre_nms_topn);
/* rpn_cls_prob_d [1,H,W,2A] (fg = channels [A,2A)), rpn_bbox_pred_d [1,H,W,4A], im_h/im_w =
 * im_info[0], im_info[1] (scaled image), base_d = float64 base anchors [A,4] (anchors are
 * re-generated in-kernel from the anchor index; they are never read from HBM).
 * rois_d [post_nms_topn,5] = (0,x1,y1,x2,y2), scores_d [post_nms_topn]; rows >= *num_d are zero.
 * pre_nms_topn <= 0 means "all" (proposal_layer.py:35); min(pre, H*W*A) <= 65536.  Rule FRCNN_NMS_RULE_CPU. */
int frcnn_proposal_layer(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h,
                         float im_w, int H, int W, int A, int feat_stride, const double* base_d,
                         int pre_nms_topn, int post_nms_topn, double nms_thresh, float* rois_d,
                         float* scores_d, int* num_d, void* ws, size_t ws_bytes, void* stream);
/* The same layer for B same-size images in the same launches (the reference graph is batch-1, lib/nets/network.py:388; a
 * batch is B independent images): rpn_cls_prob_d [B,H,W,2A], rpn_bbox_pred_d [B,H,W,4A] -> rois_d [B*post_nms_topn,5] with
 * rois[:,0] = image index (the batch_inds column of proposal_layer.py:49-51), scores_d [B*post_nms_topn], num_d [B];
 * `rule` = FRCNN_NMS_RULE_CPU / _GPU (what lib/model/nms_wrapper.py:15-23 picks by cfg.USE_GPU_NMS).
 * Stages per launch: decode+clip+key, exact radix select of the pre_nms_topn best keys, counting sort of those, 64x64-tile
 * suppression bitmask, greedy reduce with prefetched mask rows. */
size_t frcnn_proposal_batched_workspace_bytes(int B, int H, int W, int A, int pre_nms_topn);
int frcnn_proposal_layer_batched(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, int B, float im_h, float im_w,
                                 int H, int W, int A, int feat_stride, const double* base_d, int pre_nms_topn,
                                 int post_nms_topn, double nms_thresh, int rule, float* rois_d, float* scores_d, int* num_d,
                                 void* ws, size_t ws_bytes, void* stream);
/* TEST.MODE 'top': top rpn_top_n by score, decode+clip, no NMS; requires H*W*A >= rpn_top_n. */
int frcnn_proposal_top_layer(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h,
                             float im_w, int H, int W, int A, int feat_stride, const double* base_d,
                             int rpn_top_n, float* rois_d, float* scores_d, void* ws, size_t ws_bytes,
                             void* stream);
/* proposal_top_layer for H*W*A < rpn_top_n (lib/layer_utils/proposal_top_layer.py:30-33): the reference then draws
 * `npr.choice(length, size=rpn_top_n, replace=True)` from numpy's global stream; the caller draws the same indices on the
 * host and this entry decodes + clips exactly those anchors in the given order.  top_inds_d [n_inds] int32 (device). */
int frcnn_proposal_top_layer_inds(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h, float im_w, int H,
                                  int W, int A, int feat_stride, const double* base_d, const int* top_inds_d, int n_inds,
                                  float* rois_d, float* scores_d, void* stream);

/* USE_E2E_TF graph (the reference's default, lib/model/config.py:275).
 * frcnn_non_max_suppression: tf.image.non_max_suppression(boxes [k,4], scores [k], max_output_size, iou_threshold) as called
 * at lib/layer_utils/proposal_layer.py:72 -- order (score desc, index asc), overlap WITHOUT the +1 pixel convention, corner
 * order normalised, degenerate boxes never overlap, suppress iff iou > iou_threshold (f32); k <= 65536.
 * Workspace: frcnn_nms_workspace_bytes(k).
 * frcnn_proposal_layer_tf: proposal_layer_tf (proposal_layer.py:56-84): decode + clip ALL H*W*A anchors, that NMS over all of
 * them (no pre-NMS top-N), first post_nms_topn survivors -> rois [post,5] (zero padded), scores [post], *num_d.
 * Workspace: frcnn_proposal_workspace_bytes(H, W, A, 0). */
int frcnn_non_max_suppression(const float* boxes_d, const float* scores_d, int k, int max_output_size, float iou_threshold,
                              int* selected_d, int* num_d, void* ws, size_t ws_bytes, void* stream);
int frcnn_proposal_layer_tf(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, float im_h, float im_w, int H, int W,
                            int A, int feat_stride, const double* base_d, int post_nms_topn, float nms_thresh, float* rois_d,
                            float* scores_d, int* num_d, void* ws, size_t ws_bytes, void* stream);
/* B images per launch (layouts as frcnn_proposal_layer_batched); workspace frcnn_proposal_batched_workspace_bytes(B,H,W,A,0). */
int frcnn_proposal_layer_tf_batched(const float* rpn_cls_prob_d, const float* rpn_bbox_pred_d, int B, float im_h, float im_w,
                                    int H, int W, int A, int feat_stride, const double* base_d, int post_nms_topn,
                                    float nms_thresh, float* rois_d, float* scores_d, int* num_d, void* ws, size_t ws_bytes,
                                    void* stream);

/* ---- RoI pooling: replaces tf.image.crop_and_resize as called from lib/nets/resnet_v1.py:55-76
 * and lib/nets/network.py:141-157 ----------------------------------------------------------- */
/* feat_d [H,W,C] (batch 1), rois_d [R,5] image coords, out_d [R,pool,pool,C].  fuse_max2x2 != 0:
 * sample a (2*pool)^2 grid and take the 2x2/2 max (network.py:152-157).  C % 4 == 0.  rois[:,0] must be 0 (rows with
 * another value produce zeros). */
int frcnn_crop_and_resize(const float* feat_d, int H, int W, int C, const float* rois_d, int R,
                          float feat_stride, int pool, int fuse_max2x2, float* out_d, void* stream);
/* feat_d [N,H,W,C]: the box_ind argument of tf.image.crop_and_resize is rois[:,0], as at network.py:143 / resnet_v1.py:57
 * (`batch_ids = rois[:, 0]`); rows whose index is outside [0,N) produce zeros.  Launch shape: one workgroup per
 * (roi, output row, 1/8 channel slab) with the slab as the fastest block index, so each XCD's L2 holds one slab of the map. */
int frcnn_crop_and_resize_batched(const float* feat_d, int N, int H, int W, int C, const float* rois_d, int R,
                                  float feat_stride, int pool, int fuse_max2x2, float* out_d, void* stream);

/* out = act(crop_and_resize(feat) + bias[c]), feat_d [N,H,W,C], image index = rois[:,0].  A 1x1 convolution commutes
 * with the (linear) bilinear crop, so the two 1x1 convs that consume the RoI crops (block4/unit_1 shortcut and conv1,
 * lib/nets/resnet_v1.py:115-125) can run once on the H x W map (2 394 pixels) instead of on R x 7 x 7 crops (14 700
 * pixels); bias and ReLU are applied here, after the crop, because out-of-range samples are zeros. */
int frcnn_crop_and_resize_bias_act(const float* feat_d, int N, int H, int W, int C, const float* rois_d, int R,
                                   float feat_stride, int pool, const float* bias_d, int act, float* out_d,
                                   void* stream);
/* tuning knobs for A/B runs (thread-local): key 4 = channel-slab count of the crop kernels (-1 automatic); key 5 = 1: the crop as one workgroup
 * per (roi, output row, slab) (rounds 2-4) instead of one per (roi, slab) -- the same bits. */
int frcnn_detect_set_tuning(int key, int value);

/* ---- test-time post-processing: replaces lib/model/test.py:95-102 (im_detect) and :162-180
 * (test_net per-class NMS + max_per_image cut) ----------------------------------------------- */
size_t frcnn_detect_post_workspace_bytes(int R, int C);
/* cls_prob_d [R,C], bbox_pred_d [R,4C] (already *stds+means; NULL = cfg.TEST.BBOX_REG False: every class takes the un-regressed,
 * un-clipped rois / scale, test.py:103-105 -- also in frcnn_detect_post_batched and frcnn_im_detect_boxes), rois_d [R,5] (scaled image coords),
 * num_rois_d: device int (rows >= it are ignored) or NULL.  im_scale: float64 like im_scales[0];
 * im_h, im_w: ORIGINAL image size.  out_dets_d [max_out,6] = x1,y1,x2,y2,score,class (class-major,
 * score-descending inside a class == all_boxes[j][i] order), *out_count_d = number of detections
 * (may exceed max_out only when ties straddle the max_per_image cut; excess rows are dropped).
 * R <= 1024.  Rule FRCNN_NMS_RULE_CPU. */
int frcnn_detect_post(const float* cls_prob_d, const float* bbox_pred_d, const float* rois_d,
                      const int* num_rois_d, int R, int C, double im_scale, int im_h, int im_w,
                      double nms_thresh, float score_thresh, int max_per_image, float* out_dets_d,
                      int* out_count_d, int max_out, void* ws, size_t ws_bytes, void* stream);
/* B images per launch: cls_prob_d [B*R,C], bbox_pred_d [B*R,4C], rois_d [B*R,5] (R rows per image), num_rois_d [B] or
 * NULL -> out_dets_d [B,max_out,6] with `out_stride` floats between images (0 = dense, max_out*6), out_count_d [B].  One workgroup per (class, image); sort, bitmask and greedy reduce of a
 * class live entirely in LDS.  `rule` as in frcnn_nms_rule. */
size_t frcnn_detect_post_batched_workspace_bytes(int B, int R, int C);
int frcnn_detect_post_batched(const float* cls_prob_d, const float* bbox_pred_d, const float* rois_d, const int* num_rois_d,
                              int B, int R, int C, double im_scale, int im_h, int im_w, double nms_thresh, int rule,
                              float score_thresh, int max_per_image, float* out_dets_d, int* out_count_d, int max_out,
                              long long out_stride, void* ws, size_t ws_bytes, void* stream);

/* The box stage of im_detect alone (lib/model/test.py:95-102): pred_boxes [R,4C] = clip(decode(rois/scale,
 * bbox_pred)) for every class, for callers that want the reference's (scores, pred_boxes) pair. */
int frcnn_im_detect_boxes(const float* rois_d, const float* bbox_pred_d, int R, int C, double im_scale,
                          int im_h, int im_w, float* boxes_d, void* stream);

/* ---- box codec: replaces lib/model/bbox_transform.py:14-81 (numpy in / numpy out helpers) -------------------- */
/* bbox_transform_inv(boxes [N,4], deltas [N,4k]) -> out [N,4k] (class quadruple j of row i decoded against boxes[i]);
 * clip_boxes(boxes [N,4k], im_shape) in place: every coordinate clamped to [0, dim-1];
 * bbox_transform(ex_rois [N,4], gt_rois [N,4]) -> targets [N,4] (dx, dy, dw, dh) in float32. */
int frcnn_bbox_transform_inv(const float* boxes_d, const float* deltas_d, int N, int k, float* out_d, void* stream);
int frcnn_clip_boxes(float* boxes_d, int N, int k, float im_h, float im_w, void* stream);
int frcnn_bbox_transform(const float* ex_rois_d, const float* gt_rois_d, int N, float* targets_d, void* stream);

/* ---- IoU matrix: replaces lib/utils/bbox.pyx:15-55 ------------------------------------------ */
/* boxes_d [n,4] f64, query_d [k,4] f64 -> out_d [n,k] f64. */
int frcnn_bbox_overlaps(const double* boxes_d, int n, const double* query_d, int k, double* out_d, void* stream);

/* ---- dense ops (the TF/slim call sites of lib/nets/network.py:323-378, resnet_v1.py:80-125,
 * vgg16.py:26-60, mobilenet_v1.py:114-172) ---------------------------------------------------- */
/* Implicit-GEMM convolution on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32).
 *   x_d  [N,H,W,Cin] NHWC;   w_d packed [Cout][KH][KW][Cin] (see frcnn_pack_filter_hwio);
 *   y_d  [N,OH,OW,Cout] = act( conv(x,w) + bias [+ residual] )
 *   ih = oh*stride - pad_top + kh, iw = ow*stride - pad_left + kw, zero outside the image
 *   (covers slim `SAME` for stride 1, resnet_utils.conv2d_same and explicit-pad + VALID).
 *   bias_d [Cout] or NULL.  residual_d NULL or an NHWC tensor [N,RH,RW,Cout] read at
 *   (oh*res_stride, ow*res_stride): res_stride 1 = plain skip, 2 = slim `subsample` shortcut.
 *   Requires Cin % 32 == 0, or Cin == 4 with fold_w (7x7 stem: the kw taps are folded into the
 *   channel run, w_d packed [Cout][KH][8][4]). */
int frcnn_conv2d_nhwc(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                      const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW,
                      int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int act, int fold_w,
                      void* stream);
/* frcnn_conv2d_nhwc with a scratch buffer: launches that would leave most of the 256 CUs idle (a single 38x63 image,
 * weight-gradient GEMMs) are cut along K into S workgroup sets writing separate partial sums (no atomics, deterministic),
 * then reduced with bias / residual / activation.  frcnn_conv2d_workspace_bytes: bytes that make the split possible for
 * this shape (0: the plain launch is used anyway).  Same arguments and results (to f32 summation order) otherwise. */
size_t frcnn_conv2d_workspace_bytes(int N, int OH, int OW, int Cout, int KH, int KW, int Cin, int fold_w);
int frcnn_conv2d_nhwc_ws(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                         const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW, int Cout, int KH,
                         int KW, int stride, int pad_top, int pad_left, int act, int fold_w, void* ws, size_t ws_bytes,
                         void* stream);

/* Tuning overrides for A/B measurements and for tests that must reach every tile configuration: THREAD-LOCAL (they affect only the
 * launches the calling thread makes afterwards, so the thread-safety contract above holds); no product path sets keys 0-7.  key 0 = force a
 * conv tile configuration id (-1 = automatic); key 1 = ablation bits; key 5 = phase stagger of co-resident workgroups; key 6 = 0 keeps
 * the short-K GEMMs off k_gemm_stream; key 7 = the workgroup count a split-K launch aims at (0 = the default 640; 160 ... 640 move the
 * ResNet-152 training step by +-1 %); key 9 = the workgroup count below which the F(4x4,3x3) Winograd transforms run in their
 * row-per-thread form (0 = the default 256; both forms give the same bits).  frcnn_gemm_x3 / frcnn_gemm_h2 take their configuration per call instead.
 * key 8 is NOT a measurement knob but launch context, set by the TEST-mode graph builder around its launches (lib/nets/network.py
 * _build_network): the number of independent images that share the following launches.  Split-K changes a sum's order, so
 * frcnn_conv2d_nhwc_ws plans it for the launch as it would look in a 4-image batch (per-image rows x 4) whatever the batch is -- the
 * same image gives the same bits at batch 1, 4 or 8 (the reference is batch-1: lib/model/test.py:88).  0 (default): plan by the launch. */
int frcnn_set_tuning(int key, int value);
/* HOST: CRC-32C (Castagnoli) of n bytes, crc = 0 to start or a previous result to extend: the checksum of TensorFlow
 * checkpoint shards / index blocks (frcnn_hip/tensor_bundle.py replaces pywrap_tensorflow.NewCheckpointReader,
 * lib/model/train_val.py:105-114). */
unsigned int frcnn_crc32c(const void* data, size_t n, unsigned int crc);

/* HOST: Snappy raw-format decompression (compressed LevelDB-table blocks of V1 TensorFlow checkpoints, the case
 * lib/model/train_val.py:108-113 warns about).  Returns bytes written (== the length header) or -1 on malformed input or
 * cap too small. */
long long frcnn_snappy_uncompress(const unsigned char* src, size_t n, unsigned char* dst, size_t cap);

/* Image preprocessing on device (lib/model/test.py:26-58 _get_image_blob, lib/utils/blob.py:33-47 prep_im_for_blob).
 * frcnn_prep_image_shape (HOST): the scale rule -- target_size / min side, capped so that round(scale * max side) <=
 * max_size -- and cv2.resize's output size cvRound(src * scale).
 * frcnn_prep_image: BGR [h][w][3] uint8 (src_is_float 0) or float32 (1) on device -> (pixel - pixel_means[c]) resized
 * with cv2.INTER_LINEAR semantics into float32 [OH][OW][out_c]; out_c = 4 adds the zero 4th channel of the staged stem
 * input.  pixel_means: HOST double[3]. */
int frcnn_prep_image_shape(int h, int w, int target_size, int max_size, double* im_scale, int* out_h, int* out_w);
int frcnn_prep_image(const void* src_d, int src_is_float, int h, int w, const double* pixel_means, double im_scale, float* out_d,
                     int OH, int OW, int out_c, void* stream);

/* G independent NT GEMMs in one launch (f32 MFMA): y[g][m][n] = sum_k x[g][m][k] * w[g][n][k];  K % 32 == 0. */
int frcnn_gemm_batched_nt(const float* x_d, const float* w_d, float* y_d, int G, int M, int N, int K, void* stream);
/* f32 "NT" GEMM on the bf16 matrix pipe with exactly split operands (csrc/gemm_x3.hip; cfg.HIP.MFMA_X3, default on in TEST mode; finite operands below the bf16 maximum -- an inf operand yields NaN, not inf): every f32 value
 * = h + m + l (three bf16 pieces, exact), a product = the six leading cross terms on v_mfma_f32_32x32x16_bf16 with f32 accumulation
 * (dropped terms <= 2^-24 relative).  Results agree with frcnn_gemm_batched_nt / frcnn_conv2d_nhwc to f32 rounding, NOT bit for bit.
 *   frcnn_gemm_x3_pack: W [G][N][K] f32 (device) -> planes [G][3][N][K] bf16 (frcnn_gemm_x3_pack_bytes bytes), once per filter.
 *   frcnn_gemm_x3:      y[g] = act(x[g] W[g]^T + bias + res[g]);  x [G][M][K], res / y [G][M][N];  K % 32 == 0, N % 64 == 0. */
size_t frcnn_gemm_x3_pack_bytes(int G, int N, int K);
int frcnn_gemm_x3_pack(const float* w_d, int G, int N, int K, void* planes_d, void* stream);
int frcnn_gemm_x3(const float* x_d, const void* planes_d, const float* bias_d, const float* res_d, float* y_d, int G, int M, int N,
                  int K, int act, int cfg, int terms, void* stream);   /* cfg: -1 = tiles by shape (else a configuration id, A/B runs); terms: 6
                                                                          (default: am*wl, al*wm, al*wl dropped, <= 2^-24 relative) or 9 (every f32
                                                                          product exact); per call -- no process-wide tuning state */
/* f32 "NT" GEMM on the 16-bit matrix pipe with block-scaled two-piece fp16 operands (csrc/gemm_h2.hip; cfg.HIP.MFMA_H2): a float32
 * value x of a 128-k block is h + l (two fp16 pieces, rounded to nearest) times ONE exact power-of-two block scale; a product = the three
 * leading cross terms on v_mfma_f32_32x32x16_f16, f32 accumulation, each block folded into the f32 sum by its exact scale.  Dropped terms
 * <= 3 * 2^-22 |x w| (measured: 1e-7 of the output scale, below the accumulation noise of any f32 kernel).  These replace the slim conv /
 * fc call sites of lib/nets/network.py:323-378, resnet_v1.py:80-125 for plain GEMM shapes, like frcnn_gemm_x3.
 *   frcnn_h2_planes_bytes: bytes of the planes [2][rows][K] fp16 of a tensor.
 *   frcnn_h2_pack_w:  W [G][N][K] f32 (device) -> planes [G][2][N][K] + w_inv [G][N] (one scale per output row), once per filter; K % 4 == 0.
 *   frcnn_h2_split:   x [M][K] f32 -> planes [2][M][K] + x_inv [K/128][M] (one scale per row and 128-k block); K % 128 == 0.
 *   frcnn_gemm_h2:    y[g] = act(x[g] W[g]^T + bias + res[g]), g < G; x as planes [2][G*M][K] + x_inv [K/128][G*M]; res / y [G*M][N] f32 (y may be
 *                     NULL); res_planes / res_inv (instead of res, may be NULL): the residual as operand planes [2][G*M][N] + [N/128][G*M],
 *                     read as (h + l) * 2^-e -- the trunk of a bottleneck chain kept as planes only (cfg.HIP.H2_TRUNK_PLANES); y_planes / y_inv (may be NULL): the result as operand planes [2][G*M][N] + [N/128][G*M] for the next GEMM, emitted
 *                     from the register epilogue (bit-identical to frcnn_h2_split of y).  K % 128 == 0, N % 128 == 0, any M.  cfg: -1 = by shape (-2 ... -5: other by-shape rules, A/B runs), else a tile configuration id (per call: no process-wide state; every configuration gives the same bits). */
size_t frcnn_h2_planes_bytes(long long rows, int K);
int frcnn_h2_pack_w(const float* w_d, int G, int N, int K, void* planes_d, float* w_inv_d, void* stream);
int frcnn_h2_split(const float* x_d, long long M, int K, void* planes_d, float* inv_d, void* stream);
int frcnn_gemm_h2(const void* x_planes_d, const float* x_inv_d, const void* w_planes_d, const float* w_inv_d, const float* bias_d,
                  const float* res_d, const void* res_planes_d, const float* res_inv_d, float* y_d, void* y_planes_d, float* y_inv_d,
                  int G, int M, int N, int K, int act, int cfg, void* stream);
/* frcnn_gemm_h2_mean: the RoI tail's last 1x1 convolution + reduce_mean over each RoI's P*P positions (lib/nets/resnet_v1.py:115-125,
 * `fc7 = tf.reduce_mean(fc7, axis=[1, 2])`) without the [G*M, N] tensor in between:
 *   mean_out[g * (M / rows) + r][n] = mean over rows [r * rows, (r + 1) * rows) of act(x[g] W^T + bias + res[g]).
 * G batch entries of M rows -- ONE ENTRY PER IMAGE, so that the (fixed) order in which a RoI's rows are added depends on the RoI's index
 * inside its image only: the same RoI gives the same bits in every batch slot and at every batch size.  M % rows == 0, rows >= 32,
 * K % 128 == 0, N % 128 == 0; operands as in frcnn_gemm_h2; ws of frcnn_gemm_h2_mean_workspace_bytes(G, M, N) bytes. */
size_t frcnn_gemm_h2_mean_workspace_bytes(int G, int M, int N);
int frcnn_gemm_h2_mean(const void* x_planes_d, const float* x_inv_d, const void* w_planes_d, const float* w_inv_d, const float* bias_d,
                       const float* res_d, const void* res_planes_d, const float* res_inv_d, int G, int M, int N, int K, int act, int rows,
                       float* mean_out_d, void* ws, size_t ws_bytes, int cfg, void* stream);
/* Winograd F(m x m, 3x3), m = 2 or 4, for stride-1 pad-1 3x3 convolutions (exact algebra, f32): filter transform on the
 * host (U [(m+2)^2][Cout][Cin], optional folded BN scale), input transform V [(m+2)^2][T][C] with
 * T = N*ceil(H/m)*ceil(W/m), the (m+2)^2 GEMMs via frcnn_gemm_batched_nt, output transform (+bias, ReLU) back to NHWC
 * [N,H,W,Cout].  m = 4 does 4x fewer multiplications than direct but rounds ~10x worse (still f32-class). */
int frcnn_winograd_filter_transform(const float* w_hwio, int Cin, int Cout, const float* scale, int m, float* u_out);
/* The filter transform on the DEVICE from the packed filter [Cout][3][3][Cin] (training: filters change every step).
 * transpose_flip = 0 -> U [(m+2)^2][Cout][Cin] (forward); 1 -> the data-gradient filter (taps flipped, channel roles swapped)
 * U' [(m+2)^2][Cin][Cout]. */
int frcnn_winograd_filter_transform_device(const float* w_packed_d, int Cout, int Cin, int m, int transpose_flip, float* u_d,
                                           void* stream);
int frcnn_winograd_input_transform(const float* x_d, int N, int H, int W, int C, int m, float* v_d, void* stream);
int frcnn_winograd_output_transform(const float* m_d, int N, int H, int W, int C, int m, const float* bias_d, int act, float* y_d,
                                    void* stream);
/* Winograd for 7x7 maps (the per-RoI crops of block4): each row of 7 outputs = F(4,3) + F(3,3), 11 transform points per
 * dimension, 121 GEMMs of [R x Cin] x [Cin x Cout] (one row per RoI) instead of the 144 of 2x2 F(4x4,3x3) tiles.  Same roles as
 * the frcnn_winograd_* functions: U [121][Cout][Cin] (host float64, or on the device from the packed filter, optionally the
 * flipped / transposed data-gradient form), V [121][R][C], M [121][R][Cout] -> y [R][7][7][Cout] (+ bias, ReLU). */
int frcnn_winograd7_filter_transform(const float* w_hwio, int Cin, int Cout, const float* scale, float* u_out);
int frcnn_winograd7_filter_transform_device(const float* w_packed_d, int Cout, int Cin, int transpose_flip, float* u_d, void* stream);
int frcnn_winograd7_input_transform(const float* x_d, int R, int C, float* v_d, void* stream);
int frcnn_winograd7_output_transform(const float* m_d, int R, int C, const float* bias_d, int act, float* y_d, void* stream);
/* The Winograd transforms as PRODUCERS of frcnn_gemm_h2 operand planes (csrc/gemm_h2.hip): the input transforms emit V as fp16 pieces
 * [2][G*T][C] + v_inv [C/128][G*T] (G = (m+2)^2 or 121 points, T tiles / RoIs) instead of float32; the output transforms emit
 * act(A^T M A + bias) as planes [2][pixels][C] + y_inv [C/128][pixels] for the bottleneck's next 1x1 convolution (resnet_v1.py:80-125:
 * conv2 -> conv3) and, when y_d is not NULL, the float32 tensor as well.  The rows one thread writes together (the points of one
 * transform row of a tile / the pixels of one output row of a tile) SHARE the scale of their common maximum -- one cross-lane
 * reduction per group; any power of two that keeps the block maximum below 2^15 is a valid scale for frcnn_gemm_h2 (numpy statement:
 * oracle/h2_ref.py split_grouped).  C % 128 == 0. */
int frcnn_winograd_input_transform_h2(const float* x_d, int N, int H, int W, int C, int m, void* v_planes_d, float* v_inv_d, void* stream);
int frcnn_winograd_output_transform_h2(const float* m_d, int N, int H, int W, int C, int m, const float* bias_d, int act, float* y_d,
                                       void* y_planes_d, float* y_inv_d, void* stream);
int frcnn_winograd7_input_transform_h2(const float* x_d, int R, int C, void* v_planes_d, float* v_inv_d, void* stream);
int frcnn_winograd7_output_transform_h2(const float* m_d, int R, int C, const float* bias_d, int act, float* y_d, void* y_planes_d,
                                        float* y_inv_d, void* stream);
/* HOST helper: HWIO (TF layout, [KH][KW][Cin][Cout]) -> packed [Cout][KH][KW][Cin], optionally
 * multiplying output channel o by scale[o] (folded frozen batch-norm gamma/sqrt(var+eps)). */
int frcnn_pack_filter_hwio(const float* w_hwio, int KH, int KW, int Cin, int Cout, const float* scale, float* out);
/* max-pool NHWC, window k, stride s, zero/neg-inf-free: out-of-image taps are ignored; (pad_top,
 * pad_left) shift the window (ResNet pool1: k3 s2 pad 1; VGG: k2 s2 SAME pad 0). */
int frcnn_maxpool_nhwc(const float* x_d, int N, int H, int W, int C, int k, int stride, int pad_top,
                       int pad_left, float* y_d, int OH, int OW, void* stream);
/* depthwise 3x3 (MobileNet, lib/nets/mobilenet_v1.py:21-49): w_d [3][3][C], bias, act. */
int frcnn_dwconv3x3_nhwc(const float* x_d, int N, int H, int W, int C, const float* w_d, const float* bias_d,
                         float* y_d, int OH, int OW, int stride, int pad_top, int pad_left, int act, void* stream);
/* ... with the result as frcnn_gemm_h2 operand planes [2][N*OH*OW][C] + y_inv [C/128][N*OH*OW] for the pointwise convolution that
 * follows (mobilenet_v1.py:21-49); y_d NULL = planes only.  C % 128 == 0. */
int frcnn_dwconv3x3_nhwc_h2(const float* x_d, int N, int H, int W, int C, const float* w_d, const float* bias_d, float* y_d,
                            void* y_planes_d, float* y_inv_d, int OH, int OW, int stride, int pad_top, int pad_left, int act, void* stream);
/* mean over the HW positions of [N,HW,C] -> [N,C]  (resnet_v1.py:124, mobilenet_v1.py:249). */
int frcnn_spatial_mean(const float* x_d, int N, int HW, int C, float* y_d, void* stream);
/* row softmax [R,C] (network.py:80-86, cls_prob). */
int frcnn_softmax_rows(const float* x_d, int R, int C, int ld, float* y_d, void* stream);
/* RPN pairwise softmax (network.py:68-86,331-334): score_d [HW, ld] channels (a, A+a) ->
 * prob_d [HW,2A]. */
int frcnn_rpn_softmax(const float* score_d, int HW, int A, int ld, float* prob_d, void* stream);
/* strided 2-D copy  dst[r, 0:cols] = src[r, col0:col0+cols]  (splits fused head outputs). */
int frcnn_copy_cols(const float* src_d, int R, int ld_src, int col0, int cols, float* dst_d, int ld_dst, void* stream);

/* ---- training targets + losses ---------------------------------------- */
/* anchor_target_layer (lib/layer_utils/anchor_target_layer.py:18-138).  gt_boxes_d [G,5] f32; anchors are
 * regenerated from base_d (float64 [A,4]).  Outputs in the reference layouts: labels_d [1,1,A*H,W],
 * bbox_targets_d / inside_w_d / outside_w_d [1,H,W,4A].  seed >= 0: fg/bg subsampling to rpn_batchsize with a
 * counter-based hash (same distribution as npr.choice without replacement, not the same stream);
 * seed < 0: no subsampling (every fg/bg anchor keeps its label) -- the deterministic part, used for parity.
 * opts (HOST, may be NULL = the reference's defaults): double[6] = {TRAIN.RPN_CLOBBER_POSITIVES (0 / 1, :57-70), TRAIN.RPN_POSITIVE_WEIGHT
 * (< 0: uniform 1 / num_examples; 0 < p < 1: p / #positives and (1 - p) / #negatives, :96-109), TRAIN.RPN_BBOX_INSIDE_WEIGHTS[4] (:91-93)}. */
size_t frcnn_anchor_target_workspace_bytes(int H, int W, int A, int max_gt);
int frcnn_anchor_target_layer(const float* gt_boxes_d, int G, float im_h, float im_w, int H, int W, int A,
                              int feat_stride, const double* base_d, int rpn_batchsize, double fg_fraction,
                              double pos_overlap, double neg_overlap, long long seed, const double* opts, float* labels_d,
                              float* bbox_targets_d, float* inside_w_d, float* outside_w_d, void* ws,
                              size_t ws_bytes, void* stream);
/* Host-oracle sampling mode: the reference draws its fg/bg subsamples from numpy's GLOBAL
 * MT19937 stream (anchor_target_layer.py:72-86); the py_func-compatible mirror lib/layer_utils/anchor_target_layer.py makes
 * the same npr.choice calls on the host and hands the chosen `disable_inds` (as indices into ALL H*W*A anchors, int32 device
 * array, fg and bg lists concatenated, no duplicates) to this entry, which then reproduces the reference's outputs bit for bit.
 * n_disable = 0: nothing is disabled and nothing is sampled. */
int frcnn_anchor_target_layer_inject(const float* gt_boxes_d, int G, float im_h, float im_w, int H, int W, int A, int feat_stride,
                                     const double* base_d, int rpn_batchsize, double fg_fraction, double pos_overlap,
                                     double neg_overlap, const int* disable_d, int n_disable, const double* opts, float* labels_d,
                                     float* bbox_targets_d, float* inside_w_d, float* outside_w_d, void* ws, size_t ws_bytes,
                                     void* stream);
/* proposal_target_layer (lib/layer_utils/proposal_target_layer.py:18-152).  rpn_rois_d [N,5], rpn_scores_d [N], N (+ G with
 * TRAIN.USE_GT) <= 3072.  means4 / stds4: HOST double[4] (np.array(cfg.TRAIN.BBOX_NORMALIZE_*), float64 arithmetic like the reference).
 * opts (HOST, may be NULL = defaults): double[5] = {TRAIN.USE_GT (0 / 1: the G gt boxes join the candidates as rows (0, box), score 0,
 * :30-36), TRAIN.BBOX_INSIDE_WEIGHTS[4] (:78; outside weight = inside > 0, :53)}.  Outputs: rois_d [B,5], roi_scores_d [B], labels_d [B], bbox_targets_d /
 * inside_w_d / outside_w_d [B,4*num_classes], counts_d[4] = {fg sampled, bg sampled, fg candidates, bg candidates}.
 * fg rows first, then bg rows (np.append(fg_inds, bg_inds), :138). */
int frcnn_proposal_target_layer(const float* rpn_rois_d, const float* rpn_scores_d, int N, const float* gt_boxes_d,
                                int G, int num_classes, int batch_size, double fg_fraction, double fg_thresh,
                                double bg_thresh_hi, double bg_thresh_lo, const double* means4, const double* stds4,
                                long long seed, const double* opts, float* rois_d, float* roi_scores_d, float* labels_d,
                                float* bbox_targets_d, float* inside_w_d, float* outside_w_d, int* counts_d,
                                void* stream);
/* Same with the valid row count on the device (*num_rois_d = the proposal layer's num output; rows beyond it are ignored):
 * no host round trip between the proposal layer and the target layer. */
int frcnn_proposal_target_layer_dn(const float* rpn_rois_d, const float* rpn_scores_d, int max_rois, const int* num_rois_d,
                                   const float* gt_boxes_d, int G, int num_classes, int batch_size, double fg_fraction,
                                   double fg_thresh, double bg_thresh_hi, double bg_thresh_lo, const double* means4,
                                   const double* stds4, long long seed, const double* opts, float* rois_d, float* roi_scores_d,
                                   float* labels_d, float* bbox_targets_d, float* inside_w_d, float* outside_w_d, int* counts_d,
                                   void* stream);
/* Host-oracle sampling mode: keep_inds_d [batch_size] int32 = np.append(fg_inds, bg_inds) as the caller drew them with
 * npr.choice (proposal_target_layer.py:119-138) -- indices into the candidate set (the N proposals, then the gt boxes with
 * TRAIN.USE_GT) --, the first n_fg rows are foreground; outputs as above, in that row order. */
int frcnn_proposal_target_layer_inject(const float* rpn_rois_d, const float* rpn_scores_d, int N, const float* gt_boxes_d, int G,
                                       int num_classes, int batch_size, const int* keep_inds_d, int n_fg, const double* means4,
                                       const double* stds4, const double* opts, float* rois_d, float* roi_scores_d, float* labels_d,
                                       float* bbox_targets_d, float* inside_w_d, float* outside_w_d, void* stream);
/* Losses of lib/nets/network.py:264-321, value + gradient w.r.t. the logits / predictions in one call.
 * softmax CE: logits [R,C] rows (rpn_A = 0) or the RPN pair layout (rpn_A = A: logits [H*W,2A], element
 * r = (a*H+h)*W+w pairs channels (a, A+a), labels_d in the [1,1,A*H,W] layout); label < 0 is ignored;
 * loss = mean over the selected rows.  SmoothL1: loss = sum(out_w * f(in_w*(pred-tgt))) / mean_divisor. */
size_t frcnn_loss_workspace_bytes(long long elements);
int frcnn_softmax_ce_loss(const float* logits_d, const float* labels_d, int R, int C, int rpn_A, int rpn_H, int rpn_W,
                          float* loss_d, float* dlogits_d, void* ws, size_t ws_bytes, void* stream);
int frcnn_smooth_l1_loss(const float* pred_d, const float* targets_d, const float* inside_w_d, const float* outside_w_d,
                         long long n, float sigma, float mean_divisor, float* loss_d, float* dpred_d, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- backward pass + solver (lib/model/train_val.py:116-153) ------------------ */
/* Operand re-layouts that let frcnn_conv2d_nhwc compute conv gradients (see csrc/backward_kernels.hip):
 *   frcnn_transpose_pad:         out[c][m] = in[m][c], rows zero-padded to Mp (Mp % 32 == 0 for use as a GEMM K axis)
 *   frcnn_im2col_t:              out[(kh*KW+kw)*Cin + c][m] = x[img, oh*s-pt+kh, ow*s-pl+kw, c]  (transposed im2col)
 *   frcnn_flip_transpose_filter: out[c][KH-1-kh][KW-1-kw][n] = w[n][kh][kw][c]                    (dgrad filter) */
int frcnn_transpose_pad(const float* in_d, int M, int C, float* out_d, int Mp, void* stream);
int frcnn_im2col_t(const float* x_d, int N, int H, int W, int Cin, int OH, int OW, int KH, int KW, int stride,
                   int pad_top, int pad_left, float* out_d, int Mp, void* stream);
int frcnn_flip_transpose_filter(const float* w_d, int Cout, int KH, int KW, int Cin, float* out_d, void* stream);
/* dX of a strided convolution (gather form; the ResNet head has two stride-2 3x3 convs). */
int frcnn_conv2d_dgrad_strided(const float* dy_d, int N, int OH, int OW, int Cout, const float* w_d, int KH, int KW,
                               int Cin, int stride, int pad_top, int pad_left, float* dx_d, int H, int W,
                               int accumulate, void* stream);
/* dW of a convolution straight from dY [N,OH,OW,Cout] and X [N,H,W,Cin] (csrc/wgrad_tn.hip: C = A^T B on the f32 matrix pipe, both
 * operands read as they lie -- no transposed copies, no im2col matrix; slices of the pixel range reduced in a fixed order):
 * dw_d [Cout][KH][KW][Cin], the layout of the packed forward filter (what the autograd of slim.conv2d hands the optimizer,
 * train_val.py:128-145).  Needs Cin % 64 == 0 and Cout % 64 == 0 (frcnn_conv2d_wgrad_supported; the other layers keep the
 * transpose_pad / im2col_t + frcnn_conv2d_nhwc_ws route above).  ws: frcnn_conv2d_wgrad_workspace_bytes(...) bytes (0 = none needed). */
int frcnn_conv2d_wgrad_supported(int Cin, int Cout);
size_t frcnn_conv2d_wgrad_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW);
int frcnn_conv2d_wgrad(const float* dy_d, const float* x_d, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                       int stride, int pad_top, int pad_left, float* dw_d, void* ws, size_t ws_bytes, void* stream);
/* tuning hook of the calling thread: tile edge (64 / 128, 0 = by shape) and the workgroup count the slicing aims at (0 = 512) */
void frcnn_conv2d_wgrad_set_plan(int tile, int min_workgroups);
/* The same gradient on the fp16 matrix pipe in the arithmetic of frcnn_gemm_h2 (csrc/wgrad_h2.hip): every 64-pixel column segment of dY
 * and of the X tap is split in registers into two fp16 pieces with its own exact power-of-two scale, three MFMAs per product, f32
 * accumulation -- nothing but dY and X is read from HBM.  Same arguments, same limits (Cin, Cout % 64 == 0), its own workspace size. */
size_t frcnn_conv2d_wgrad_h2_workspace_bytes(int N, int OH, int OW, int Cin, int Cout, int KH, int KW);
int frcnn_conv2d_wgrad_h2(const float* dy_d, const float* x_d, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
                          int stride, int pad_top, int pad_left, float* dw_d, void* ws, size_t ws_bytes, void* stream);
void frcnn_conv2d_wgrad_h2_set_plan(int tile, int min_workgroups);      /* as above; 0 = 256 workgroups */
int frcnn_relu_bwd(float* grad_d, const float* y_d, long long n, void* stream);              /* grad *= (y > 0) */
/* The data-gradient chain with that ReLU gradient INSIDE the producing launch (the reverse sweep of lib/nets/resnet_v1.py's bottlenecks:
 * every convolution input of the trunk is a ReLU output, so the gradient a data-gradient launch produces is masked by the launch's own
 * forward input): y = mask > 0 ? y : 0, mask = the forward activation, float32, same shape as y.  The select is exact -- the float32
 * result of each entry equals its unmasked counterpart followed by frcnn_relu_bwd bit for bit; operand planes, where emitted, are those
 * of the MASKED tensor (the GEMM's: frcnn_h2_split's bits; the Winograd transforms': one scale per group of rows written together, like
 * their unmasked _h2 forms) -- tests/test_chain_fusion_gpu.py.
 *   frcnn_conv2d_nhwc_masked_ws:            frcnn_conv2d_nhwc_ws (f32 matrix pipe; split-K plans mask in their finishing pass)
 *   frcnn_gemm_h2_masked:                   frcnn_gemm_h2 (float32 residual only)
 *   frcnn_winograd_output_transform_masked: F(4x4,3x3) output transform without bias / activation; float32 (y_d) and / or operand planes
 *   frcnn_winograd7_output_transform_masked: the 7x7-map transform likewise */
int frcnn_conv2d_nhwc_masked_ws(const float* x_d, int N, int H, int W, int Cin, const float* w_d, const float* bias_d,
                                const float* residual_d, int RH, int RW, int res_stride, float* y_d, int OH, int OW,
                                int Cout, int KH, int KW, int stride, int pad_top, int pad_left, int act, const float* mask_d,
                                void* ws, size_t ws_bytes, void* stream);
int frcnn_gemm_h2_masked(const void* x_planes_d, const float* x_inv_d, const void* w_planes_d, const float* w_inv_d, const float* bias_d,
                         const float* res_d, const float* mask_d, float* y_d, void* y_planes_d, float* y_inv_d, int G, int M, int N, int K,
                         int act, int cfg, void* stream);
int frcnn_winograd_output_transform_masked(const float* m_d, int N, int H, int W, int C, int m, const float* mask_d, float* y_d,
                                           void* y_planes_d, float* y_inv_d, void* stream);
int frcnn_winograd7_output_transform_masked(const float* m_d, int R, int C, const float* mask_d, float* y_d, void* y_planes_d,
                                            float* y_inv_d, void* stream);
int frcnn_relu6_bwd(float* grad_d, const float* y_d, long long n, void* stream);             /* grad *= (0 < y < 6) */
/* Reverse-sweep pieces of the VGG16 / MobileNet-v1 TRAIN graphs (lib/nets/vgg16.py:26-60, mobilenet_v1.py:114-172):
 *   frcnn_maxpool_bwd:      gradient of slim.max_pool2d (padding at the bottom / right only): a window's gradient goes to its first
 *                           maximum in (row, column) order; dx is written, overlapping windows (k > stride) are summed.
 *   frcnn_dropout:          tf.nn.dropout, y = x / keep_prob * floor(keep_prob + u(seed, i)); the same call on dy is the backward
 *                           pass (the mask is a counter-based function of (seed, element index), never stored).  In-place allowed.
 *   frcnn_dwconv3x3_dgrad:  data gradient of frcnn_dwconv3x3_nhwc (w = the forward filter [3][3][C]).
 *   frcnn_dwconv3x3_wgrad:  filter gradient [3][3][C], optionally times scale[c] (chain rule through a frozen-BN fold); two
 *                           deterministic stages through a workspace of frcnn_dwconv3x3_wgrad_workspace_bytes bytes.
 *   frcnn_dwconv3x3_refold: wf[tap][c] = w[tap][c] * scale[c] (forward filter from the master copy after a solver step). */
int frcnn_maxpool_bwd(const float* x_d, int N, int H, int W, int C, int k, int stride, const float* y_d, const float* dy_d,
                      int OH, int OW, float* dx_d, void* stream);
int frcnn_dropout(const float* x_d, long long n, unsigned long long seed, float keep_prob, float* y_d, void* stream);
int frcnn_dwconv3x3_dgrad(const float* g_d, int N, int OH, int OW, int C, const float* w_d, float* dx_d, int H, int W, int stride,
                          int pad_top, int pad_left, int accumulate, void* stream);
size_t frcnn_dwconv3x3_wgrad_workspace_bytes(int N, int OH, int OW, int C);
int frcnn_dwconv3x3_wgrad(const float* g_d, const float* x_d, int N, int H, int W, int C, int OH, int OW, int stride, int pad_top,
                          int pad_left, const float* scale_d, float* dw_d, void* ws, size_t ws_bytes, void* stream);
int frcnn_dwconv3x3_refold(const float* w_d, const float* scale_d, int C, float* wf_d, void* stream);
int frcnn_add_strided(const float* src_d, int N, int OH, int OW, int C, float* dst_d, int H, int W, int stride,
                      int accumulate, void* stream);                                        /* skip / subsample gradient */
int frcnn_spatial_mean_bwd(const float* dy_d, int N, int HW, int C, float* dx_d, void* stream);
int frcnn_colsum(const float* dy_d, int M, int C, float* db_d, void* stream);                 /* bias gradient */
/* Gradient of tf.image.crop_and_resize w.r.t. the feature map: dfeat_d += sum over the output samples of their four bilinear taps
 * (dfeat_d must be zeroed or hold the gradient to accumulate into).  DETERMINISTIC since round 5: a gather per feature row in ascending
 * (roi, sample row, sample column, tap) order, no float atomics (TensorFlow's CropAndResizeGradImage scatters with atomics; its sum
 * order, hence the last bits, vary from run to run).  Any R; W up to ~600 feature columns (FRCNN_E_UNSUPPORTED beyond): the launch plan
 * picks 256 / 128 / 64 channels per workgroup so that the row buffer fits the LDS and walks the hit list in windows -- the same bits under
 * every plan.  _plan: the same with an explicit LDS budget in bytes (0 = the CU's 160 KB; tests force the narrow / windowed plans).
 * No max-pool variant (ResNet crops 7x7 directly). */
int frcnn_crop_and_resize_bwd(const float* dout_d, int H, int W, int C, const float* rois_d, int R, float feat_stride,
                              int pool, float* dfeat_d, void* stream);
int frcnn_crop_and_resize_bwd_plan(const float* dout_d, int H, int W, int C, const float* rois_d, int R, float feat_stride,
                                   int pool, float* dfeat_d, size_t max_lds_bytes, void* stream);
/* tf.train.MomentumOptimizer step on a packed master filter [Cout][K] (+ refresh of the BN-folded copy):
 * g = grad_scale*grad*scale[n] + weight_decay*w ; acc = momentum*acc + g ; w -= lr*acc ; w_folded = w*scale[n].
 * scale_d / w_folded_d may be NULL (biases, fc). */
int frcnn_sgd_momentum(float* w_d, float* acc_d, float* w_folded_d, const float* grad_d, const float* scale_d,
                       long long n, int K, float lr, float momentum, float weight_decay, float grad_scale, void* stream);
/* The same update for `count` tensors in ONE launch.  desc_table_d: device array of `count` descriptors of frcnn_sgd_desc_bytes()
 * bytes each, laid out as { float* w; float* acc; float* w_folded (or NULL); const float* grad; const float* scale (or NULL);
 * long long n; int K; float lr_mult; float weight_decay; int pad; } -- lr_mult = 2 for biases under TRAIN.DOUBLE_BIAS
 * (lib/model/train_val.py:132-141), 1 otherwise.  Element-wise identical to frcnn_sgd_momentum. */
size_t frcnn_sgd_desc_bytes(void);
int frcnn_sgd_momentum_multi(const void* desc_table_d, int count, float lr, float momentum, float grad_scale, void* stream);
/* ... for the descriptors [first, first + count) of the table: the solver updates the tensors of a finished part of the reverse sweep
 * on its own stream while the sweep goes on (single-GPU runs; a data-parallel step updates after the all-reduce, in one launch). */
int frcnn_sgd_momentum_range(const void* desc_table_d, int first, int count, float lr, float momentum, float grad_scale, void* stream);
/* out (+)= scale * sum(w^2)   (slim l2_regularizer value); ws >= 2 KiB. */
int frcnn_sumsq(const float* w_d, long long n, double scale, float* out_d, int accumulate, void* ws, size_t ws_bytes, void* stream);
/* The same over `count` tensors in two launches: ptr_table_d = device array of `count` float pointers, sizes_d = their
 * element counts; *out_d (+)= scale * sum over all tensors of sum(w^2); ws >= 64 * count doubles.  Deterministic. */
int frcnn_sumsq_multi(const void* ptr_table_d, const long long* sizes_d, int count, double scale, float* out_d, int accumulate,
                      void* ws, size_t ws_bytes, void* stream);

/* ---- stream capture (one hipGraph per image-shape; replaces the per-image sess.run) ---------- */
int frcnn_graph_begin(void* stream);
int frcnn_graph_end(void* stream, void** graph_exec_out);
int frcnn_graph_launch(void* graph_exec, void* stream);
int frcnn_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H_ */
