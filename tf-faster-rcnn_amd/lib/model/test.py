"""model.test -- the test loop of the reference (lib/model/test.py) on the device chain.

`im_detect` / `test_net` keep the reference's names and return conventions.  `im_detect` / `detect` take the ALREADY
SCALED, mean-subtracted blob plus its scale; `_get_image_blob` (test.py:26-58) is provided ON DEVICE
(`frcnn_prep_image`: uint8 BGR in HBM -> mean-subtracted, cv2.INTER_LINEAR-resized, staged stem input), and
`im_detect_bgr` / `detect_bgr` are the raw-image forms (the reference's `im_detect(sess, net, im)` signature)."""
import numpy as np
import torch

from frcnn_hip import ops
from frcnn_hip.runtime import Timer
from model.config import cfg


def im_detect(sess, net, blob, im_scale, im_shape):
    """blob [1,H,W,3] f32 (BGR - PIXEL_MEANS, scaled by im_scale), im_shape = original (h, w[, c]).
    Returns (scores [R,C], pred_boxes [R,4C]) like lib/model/test.py:86-107 -- decode + clip of the
    per-class boxes included -- computed from the device tensors."""
    im_info = np.array([blob.shape[1], blob.shape[2], im_scale], dtype=np.float32)
    img = net._stage_image(sess, blob, im_info)
    p = net.forward_device(sess, img, im_info)
    n = p["rois"].shape[0] if net._num_rois is None else int(net._num_rois.item())
    rois, bbox_pred = p["rois"][:n].contiguous(), (p["bbox_pred"][:n].contiguous() if cfg.TEST.BBOX_REG else None)
    pred_boxes = ops.im_detect_boxes(rois, bbox_pred, im_scale, im_shape[0], im_shape[1], net._num_classes)      # test.py:95-105
    return p["cls_prob"][:n].cpu().numpy(), pred_boxes.cpu().numpy()


def _get_image_blob(sess, net, im):
    """test.py:26-58 on device for cfg.TEST.SCALES[0]: im = BGR uint8 (or float32) [h,w,3], numpy or device tensor.
    Returns (staged image [1,H,W,4] on device -- zero 4th channel, what forward_device takes --, im_scale)."""
    if isinstance(im, np.ndarray):
        im = torch.from_numpy(np.ascontiguousarray(im)).to(sess.device, non_blocking=True)     # 3 B/pixel over PCIe
    im_scale, OH, OW = ops.prep_image_shape(im.shape[0], im.shape[1], cfg.TEST.SCALES[0], cfg.TEST.MAX_SIZE)
    with net.shape_scope(sess, (1, OH, OW, 4), (OH, OW)):        # the staged image belongs to its shape's scope (freed with the shape's graph)
        out = sess.buf(net._tag + "/image", (1, OH, OW, 4))
    ops.prep_image(im, cfg.PIXEL_MEANS, im_scale, (OH, OW), out=out, out_c=4)
    return out, im_scale


def im_detect_bgr(sess, net, im):
    """The reference's `im_detect(sess, net, im)` (test.py:86-107): raw BGR image in, (scores, pred_boxes) out."""
    img, im_scale = _get_image_blob(sess, net, im)
    im_info = np.array([img.shape[1], img.shape[2], im_scale], dtype=np.float32)
    p = net.forward_device(sess, img, im_info)
    n = p["rois"].shape[0] if net._num_rois is None else int(net._num_rois.item())
    pred_boxes = ops.im_detect_boxes(p["rois"][:n].contiguous(), p["bbox_pred"][:n].contiguous() if cfg.TEST.BBOX_REG else None, im_scale,
                                     im.shape[0], im.shape[1], net._num_classes)
    return p["cls_prob"][:n].cpu().numpy(), pred_boxes.cpu().numpy()


def detect_bgr(sess, net, im, max_per_image=100, thresh=0.):
    """Raw BGR image -> per-class detections, everything after the (optional) H2D copy on the GPU."""
    img, im_scale = _get_image_blob(sess, net, im)
    im_info = np.array([img.shape[1], img.shape[2], im_scale], dtype=np.float32)
    dets, cnt = net.detect_device(sess, img, im_info, im.shape[:2], max_per_image=max_per_image, thresh=thresh)
    n = min(int(cnt.item()), dets.shape[0])
    rec = dets[:n].cpu().numpy()
    return [np.zeros((0, 5), dtype=np.float32)] + [rec[rec[:, 5] == j, :5] for j in range(1, net._num_classes)]


def detect(sess, net, blob, im_scale, im_shape, max_per_image=100, thresh=0.):
    """Whole per-image body of test_net (test.py:156-180) on the GPU -> all_boxes-style list over
    classes of [n,5] arrays (x1,y1,x2,y2,score)."""
    im_info = np.array([blob.shape[1], blob.shape[2], im_scale], dtype=np.float32)
    img = net._stage_image(sess, blob, im_info)
    dets, cnt = net.detect_device(sess, img, im_info, im_shape, max_per_image=max_per_image, thresh=thresh)
    n = min(int(cnt.item()), dets.shape[0])
    rec = dets[:n].cpu().numpy()
    out = [np.zeros((0, 5), dtype=np.float32) for _ in range(net._num_classes)]
    for j in range(1, net._num_classes):
        out[j] = rec[rec[:, 5] == j, :5]
    return out


def apply_nms(all_boxes, thresh):
    """test.py:109-135: non-maximum suppression (the device NMS behind model.nms_wrapper.nms) on every all_boxes[cls][image]
    of a finished test_net run; degenerate boxes (x2 <= x1 or y2 <= y1) are dropped first.  Returns a new nested list."""
    from model.nms_wrapper import nms
    num_classes, num_images = len(all_boxes), len(all_boxes[0])
    nms_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    for c in range(num_classes):
        for i in range(num_images):
            dets = all_boxes[c][i]
            if isinstance(dets, list) or len(dets) == 0:
                continue
            dets = np.asarray(dets, dtype=np.float32)
            dets = dets[(dets[:, 2] > dets[:, 0]) & (dets[:, 3] > dets[:, 1])]
            if dets.shape[0] == 0:
                continue
            keep = nms(dets, thresh)
            if len(keep) == 0:
                continue
            nms_boxes[c][i] = dets[keep, :].copy()
    return nms_boxes


def imdb_images(imdb):
    """BGR uint8 images of an imdb (datasets.pascal_voc), decoded with PIL (cv2 is not available here; both wrap libjpeg, the
    decoded pixels may differ in the last bit from cv2.imread's)."""
    from PIL import Image
    for i in range(imdb.num_images):
        yield np.ascontiguousarray(np.asarray(Image.open(imdb.image_path_at(i)).convert("RGB"))[:, :, ::-1])


def test_net_imdb(sess, net, imdb, output_dir, max_per_image=100, thresh=0.):
    """The reference's test_net(sess, net, imdb, weights_filename) (test.py:139-192): every image of the imdb through the
    raw-image device path, then detections.pkl + imdb.evaluate_detections."""
    import os
    import pickle
    all_boxes = [[[] for _ in range(imdb.num_images)] for _ in range(imdb.num_classes)]
    _t = {'im_detect': Timer(), 'misc': Timer()}
    for i, im in enumerate(imdb_images(imdb)):
        _t['im_detect'].tic()
        per_class = detect_bgr(sess, net, im, max_per_image, thresh)
        _t['im_detect'].toc()
        for j in range(1, imdb.num_classes):
            all_boxes[j][i] = per_class[j]
        print('im_detect: {:d}/{:d} {:.3f}s {:.3f}s'.format(i + 1, imdb.num_images, _t['im_detect'].average_time, _t['misc'].average_time))
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, 'detections.pkl'), 'wb') as f:
        pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)
    print('Evaluating detections')
    imdb.evaluate_detections(all_boxes, output_dir)
    return all_boxes


def test_net(sess, net, images, num_classes=None, max_per_image=100, thresh=0., imdb=None, output_dir=None):
    """images: iterable of (blob, im_scale, im_shape).  Returns all_boxes[cls][image] like the
    reference; prints the same per-image timing line (test.py:183-185).  With an imdb (datasets.pascal_voc) the
    detections are written and evaluated as at test.py:187-192: `detections.pkl` + imdb.evaluate_detections."""
    images = list(images)
    num_classes = net._num_classes if num_classes is None else num_classes
    all_boxes = [[[] for _ in range(len(images))] for _ in range(num_classes)]
    _t = {'im_detect': Timer(), 'misc': Timer()}
    for i, (blob, im_scale, im_shape) in enumerate(images):
        _t['im_detect'].tic()
        per_class = detect(sess, net, blob, im_scale, im_shape, max_per_image, thresh)
        torch.cuda.synchronize()
        _t['im_detect'].toc()
        _t['misc'].tic()
        for j in range(1, num_classes):
            all_boxes[j][i] = per_class[j]
        _t['misc'].toc()
        print('im_detect: {:d}/{:d} {:.3f}s {:.3f}s'.format(i + 1, len(images), _t['im_detect'].average_time,
                                                            _t['misc'].average_time))
    if imdb is not None and output_dir is not None:
        import os
        import pickle
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, 'detections.pkl'), 'wb') as f:
            pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)
        print('Evaluating detections')
        imdb.evaluate_detections(all_boxes, output_dir)
    return all_boxes
