"""CPU restatement of the validation side of the path (scope row f-2): eval decode, NMS, mAP.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parity status
  * decode_library / CalculateMAP / the candidate logic around NMS: PINNED by tests/golden/eval.npz, captured from the
    reference itself (oracle/make_golden.py eval).
  * nms(): the arithmetic lives in torchvision.ops.nms (torchvision 0.11.2 pinned by the reference's requirements; the
    module is absent from this image and from /root/reference) -> PARITY UNPINNED for the greedy suppression core.  It
    restates torchvision's published CPU kernel (torchvision/csrc/ops/cpu/nms_kernel.cpp, v0.11.2): sort by score
    descending, walk the order, suppress j when inter / (area_i + area_j - inter) > iou_threshold, return the kept
    indices in score order.  Ties in score are broken by original index (stable sort); torchvision leaves them
    unspecified.  The golden NMS vectors ran the reference's own wrappers with this function standing in for
    torchvision.ops.nms.

Reference files restated: detection/models/yolov3.py:35-53, detection/tools/NMS.py:5-23,
demos/yolov3_u/utils/nms.py:5-98, demos/yolov3_u/inference.py:58-120, metrics/map.py:6-141, utils/fit.py:73-105.
"""
import numpy as np
import torch


# ------------------------------------------------------------------------------------------------ decode
def decode_library(head_out, strides, anchors_px):
    """detection/models/yolov3.py:35-53.  head_out: list of [B,A,H,W,5+C]; anchors_px: list of [A,2] pixel anchors.
    Rows of a level are ordered (a, y, x); the missing ``offset`` helper is the (x, y) cell grid."""
    res = []
    for out, stride, anc in zip(head_out, strides, anchors_px):
        bs, A, H, W, K = out.shape
        ys = torch.arange(H).view(H, 1).expand(H, W)
        xs = torch.arange(W).view(1, W).expand(H, W)
        cell = torch.stack([xs, ys], dim=2).to(out)
        xy = (out[..., 0:2].sigmoid() + cell) * stride
        wh = torch.exp(out[..., 2:4]) * anc.view(A, 1, 1, 2).to(out)
        res.append(torch.cat((xy, wh, out[..., 4:].sigmoid()), -1).reshape(bs, -1, K))
    return torch.cat(res, 1)


def xywh2xyxy(b):
    hw, hh = b[:, 2] / 2, b[:, 3] / 2
    return torch.stack([b[:, 0] - hw, b[:, 1] - hh, b[:, 0] + hw, b[:, 1] + hh], dim=1)


def decode_demo(predict_layers, strides, anchors_feat, resize_ratio, padding_left, padding_top, ori_width, ori_height):
    """demos/yolov3_u/inference.py:58-106 up to the NMS call: per level [bs, A*(5+C), h, w] -> rows (y, x, a) with
    xy = (sigmoid*2 - 0.5 + cell) * stride, wh = (sigmoid*2)^2 * anchor * stride, un-letterboxed, clamped, boxes with
    w <= 5 or h <= 5 dropped, converted to clamped xyxy."""
    rows = []
    for predict, stride, anchor in zip(predict_layers, strides, anchors_feat):
        A = anchor.size(0)
        bs, c, h, w = predict.shape
        K = c // A
        p = predict.permute(0, 2, 3, 1).reshape(bs, h, w, A, K).clone()
        ys = torch.arange(h).view(h, 1).expand(h, w)
        xs = torch.arange(w).view(1, w).expand(h, w)
        cell = torch.stack([xs, ys], dim=2).view(1, h, w, 1, 2).to(p)
        p[..., 0:2] = (torch.sigmoid(p[..., 0:2]) * 2 - 0.5 + cell) * stride
        p[..., 2:4] = (torch.sigmoid(p[..., 2:4]) * 2) ** 2 * anchor.view(1, 1, 1, A, 2).to(p) * stride
        p[..., 4:] = torch.sigmoid(p[..., 4:])
        p = p.reshape(-1, K)
        p[:, 0] = ((p[:, 0] - padding_left) / resize_ratio).clamp(0, ori_width - 1)
        p[:, 1] = ((p[:, 1] - padding_top) / resize_ratio).clamp(0, ori_height - 1)
        p[:, 2] = (p[:, 2] / resize_ratio).clamp(0, ori_width)
        p[:, 3] = (p[:, 3] / resize_ratio).clamp(0, ori_height)
        p = p[(p[:, 2] > 5) & (p[:, 3] > 5)]
        box = xywh2xyxy(p[:, 0:4])
        p[:, 0] = box[:, 0].clamp(0, ori_width - 1)
        p[:, 1] = box[:, 1].clamp(0, ori_height - 1)
        p[:, 2] = box[:, 2].clamp(0, ori_width - 1)
        p[:, 3] = box[:, 3].clamp(0, ori_height - 1)
        rows.append(p)
    return torch.cat(rows, dim=0)


# ------------------------------------------------------------------------------------------------ NMS
def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms restated (see the module docstring): kept indices, highest score first."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    n = b.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = np.argsort(-s, kind='stable')
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    thr = np.float32(iou_threshold)
    keep = []
    for pos in range(n):
        i = order[pos]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        xx1, yy1 = np.maximum(x1[i], x1[rest]), np.maximum(y1[i], y1[rest])
        xx2, yy2 = np.minimum(x2[i], x2[rest]), np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def nms_library(prediction, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """detection/tools/NMS.py:5-23 -- class-agnostic, score = max_c(cls_c * obj)."""
    prediction = prediction[prediction[..., 4] > conf_thres].clone()
    if not prediction.size(0):
        return torch.zeros(0, 1), torch.zeros(0, 1), torch.zeros(0, 4)
    prediction[:, 5:] *= prediction[:, 4:5]
    boxes = xywh2xyxy(prediction[:, :4])
    scores, categories = torch.max(prediction[:, 5:], dim=1)
    keep = nms(boxes, scores, iou_thres)
    return scores[keep][:max_det].view(-1, 1), categories[keep][:max_det].view(-1, 1), boxes[keep][:max_det].view(-1, 4)


def nms_demo(prediction, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """demos/yolov3_u/utils/nms.py:5-52 -- input already xyxy; NMS score = objectness (the obj*cls overwrite is commented
    out in the reference); per-class via a 4096 * category coordinate gap.  Returns [n,6] = xyxy, obj, category."""
    max_wh, max_nms = 4096, 30000
    prediction = prediction[prediction[:, 4] > conf_thres].clone()
    if len(prediction) == 0:
        return torch.zeros((0, 6))
    prediction[:, 5:] *= prediction[:, 4:5]
    _, categories = prediction[:, 5:].max(1, keepdim=True)
    prediction = torch.cat([prediction[:, :5], categories.view(-1, 1).to(prediction)], dim=1)
    if prediction.size(0) > max_nms:
        prediction = prediction[prediction[:, 4].argsort(descending=True, stable=True)[:max_nms]]
    boxes, scores = prediction[:, :4] + prediction[:, 5:6] * max_wh, prediction[:, 4]
    keep = nms(boxes, scores, iou_thres)[:max_det]
    return prediction[keep]


def nms_demo_batch(prediction_batch, conf_thres=0.25, iou_thres=0.45, max_det=300):
    """demos/yolov3_u/utils/nms.py:54-98 -- xywh input, score = max_c(cls_c*obj) re-thresholded, per-class gap."""
    max_wh, max_nms = 4096, 30000
    out = []
    for prediction in prediction_batch:
        prediction = prediction[prediction[:, 4] > conf_thres].clone()
        prediction[:, 5:] *= prediction[:, 4:5]
        boxes = xywh2xyxy(prediction[:, :4])
        scores, categories = prediction[:, 5:].max(1, keepdim=True)
        prediction = torch.cat((boxes, scores, categories.float()), 1)[scores.view(-1) > conf_thres]
        if prediction.size(0) > max_nms:
            prediction = prediction[prediction[:, 4].argsort(descending=True, stable=True)[:max_nms]]
        boxes, scores = prediction[:, :4] + prediction[:, 5:6] * max_wh, prediction[:, 4]
        keep = nms(boxes, scores, iou_thres)[:max_det]
        out.append(prediction[keep])
    return out


# ------------------------------------------------------------------------------------------------ mAP
def iou_xyxy_batch(a, b):
    """cal_iou_batch(mode='xyxy') of detection/tools/IOU.py: [N,4] x [M,4] -> [N,M] (same form as oracle.boxes)."""
    from .boxes import cal_iou_batch
    return cal_iou_batch(a, b, mode='xyxy')


class CalculateMAP:
    """metrics/map.py:6-141 (np.float / np.long spelled float / int64: the aliases numpy < 1.24 gave them)."""

    def __init__(self, map_iou_values):
        self.map_iou_values = map_iou_values
        self.correct_all_images = []
        self.seen_all_targets_cls = []

    def process_one(self, y_pred, y_true):
        correct = np.zeros([y_pred.size(0), 2 + len(self.map_iou_values)], dtype=float)
        predict_cls, predict_conf, predict_xyxy = y_pred[:, 0], y_pred[:, 1], y_pred[:, 2:]
        target_cls, target_xyxy = y_true[:, 0], y_true[:, 1:]
        if target_cls.size(0) != 0:
            self.seen_all_targets_cls.append(target_cls.detach().cpu().numpy())
        if y_pred.size(0) == 0:
            return
        iou = iou_xyxy_batch(target_xyxy, predict_xyxy)
        matched = ((iou > self.map_iou_values[0]) & (target_cls[:, None] == predict_cls)).cpu().numpy()
        ti, pi = np.where(matched)
        m = np.stack([ti.astype(np.float32), pi.astype(np.float32), iou.cpu().numpy()[ti, pi].astype(np.float32),
                      target_cls.cpu().numpy()[ti].astype(np.float32), predict_conf.cpu().numpy()[pi].astype(np.float32)],
                     axis=1).reshape(-1, 5)
        m = m[np.argsort(-m[:, 2]), ...]
        m = m[np.unique(m[:, 1], return_index=True)[1], ...]
        m = m[np.unique(m[:, 0], return_index=True)[1], ...]
        correct[:, 0] = predict_conf.detach().cpu().numpy()
        correct[:, 1] = predict_cls.detach().cpu().numpy()
        correct[m[:, 1].astype(np.int64), 2:] = m[:, 2:3] > self.map_iou_values
        self.correct_all_images.append(correct)

    def compute_ap(self, recall, precision, method='coco'):
        m_recall = np.concatenate(([0.0], recall, [1.0]))
        m_precision = np.concatenate(([1.0], precision, [0.0]))
        env = np.flip(np.maximum.accumulate(m_precision[::-1]))
        if method == 'coco':
            x = np.linspace(0, 1, 101)
            return np.trapezoid(np.interp(x, m_recall, env), x)
        i = np.where(m_recall[1:] != m_recall[:-1])[0]
        return np.sum((m_recall[i + 1] - m_recall[i]) * env[i + 1])

    def _ap_per_class(self, total_positive, correct):
        ap = np.zeros((len(self.map_iou_values),), dtype=float)
        tp = np.cumsum(correct, axis=0)
        fn = total_positive - tp
        fp = np.cumsum(1 - correct, axis=0)
        recall = tp / (tp + fn + 1e-16)
        precision = tp / (tp + fp + 1e-16)
        for k in range(correct.shape[1]):
            ap[k] = self.compute_ap(recall[:, k], precision[:, k])
        return ap

    def fetch(self):
        correct = np.concatenate(self.correct_all_images, axis=0)
        seen = np.concatenate(self.seen_all_targets_cls, axis=0)
        uniq = np.unique(seen).tolist()
        ap = np.zeros((len(uniq), len(self.map_iou_values)), dtype=float)
        for c in uniq:
            cur = correct[correct[:, 1] == c, ...]
            cur = cur[np.argsort(-cur[:, 0]), ...]
            ap[uniq.index(c)] = self._ap_per_class(np.sum(seen == c), cur[:, 2:])
        return np.mean(ap, axis=0), np.mean(ap, axis=1), [int(c) for c in uniq]
