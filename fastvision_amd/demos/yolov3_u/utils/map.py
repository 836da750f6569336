"""``mean_average_precision`` of the demo (demos/yolov3_u/utils/map.py:8-170): per image ``process_one(y_pred [M,6] = class, conf,
xyxy; y_true [N,5] = class, xyxy)`` greedily pairs predictions and targets of equal class by IoU (one prediction per target, one
target per prediction, best IoU first) and records, for each of the ``map_iou_values`` thresholds, whether the prediction counts as
a true positive; ``fetch()`` integrates precision over recall per class (101-point COCO interpolation) and returns
(mAP per threshold, the classes seen, AP per class and threshold).

Host-side evaluation utility (numpy); the pairwise IoU matrix comes from the device kernel (fva_iou_batch) when the boxes live on
the GPU.  The demo's fit loop receives this object but never calls it (cfg/_fit.py:6); inference.py does."""
import numpy as np
import torch

__all__ = ['mean_average_precision']


def _iou_matrix(a, b):
    """[N,4] x [M,4] xyxy -> [N,M]"""
    if a.is_cuda:
        from ....detection.tools.IOU import xyxy_iou_batch
        return xyxy_iou_batch(a.float().contiguous(), b.float().contiguous()).cpu().numpy()
    a, b = a.double().numpy(), b.double().numpy()
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (area_a[:, None] + area_b[None, :] - inter + 1e-7)


class mean_average_precision:
    def __init__(self, map_iou_values):
        self.map_iou_values = np.asarray(map_iou_values, dtype=np.float64)
        self.correct_all_images = []
        self.seen_all_targets_cls = []

    def process_one(self, y_pred, y_true):
        nt = len(self.map_iou_values)
        correct = np.zeros([y_pred.size(0), 2 + nt], dtype=np.float64)
        if y_true.size(0):
            self.seen_all_targets_cls.append(y_true[:, 0].detach().cpu().numpy())
        if y_pred.size(0) == 0:
            return
        correct[:, 0] = y_pred[:, 1].detach().cpu().numpy()
        correct[:, 1] = y_pred[:, 0].detach().cpu().numpy()
        if y_true.size(0):
            iou = _iou_matrix(y_true[:, 1:].detach(), y_pred[:, 2:].detach())                     # [N, M]
            same = y_true[:, 0].detach().cpu().numpy()[:, None] == correct[None, :, 1]
            ti, pi = np.where((iou > self.map_iou_values[0]) & same)
            if len(ti):
                pairs = np.stack([ti, pi, iou[ti, pi]], axis=1)
                pairs = pairs[np.argsort(-pairs[:, 2], kind='stable')]                              # best IoU first
                pairs = pairs[np.unique(pairs[:, 1], return_index=True)[1]]                         # one target per prediction
                pairs = pairs[np.unique(pairs[:, 0], return_index=True)[1]]                         # one prediction per target
                correct[pairs[:, 1].astype(np.int64), 2:] = pairs[:, 2:3] > self.map_iou_values
        self.correct_all_images.append(correct)

    @staticmethod
    def compute_ap(recall, precision, method='coco'):
        r = np.concatenate(([0.0], recall, [1.0]))
        p = np.concatenate(([1.0], precision, [0.0]))
        env = np.flip(np.maximum.accumulate(p[::-1]))
        if method == 'coco':
            x = np.linspace(0, 1, 101)
            y = np.interp(x, r, env)
            return float(np.sum((y[1:] + y[:-1]) * 0.5 * np.diff(x)))
        i = np.where(r[1:] != r[:-1])[0]
        return float(np.sum((r[i + 1] - r[i]) * env[i + 1]))

    def fetch(self):
        nt = len(self.map_iou_values)
        if not self.correct_all_images:
            return np.zeros(nt), np.array([0]), [0]
        correct = np.concatenate(self.correct_all_images, axis=0)
        seen = np.concatenate(self.seen_all_targets_cls, axis=0) if self.seen_all_targets_cls else np.zeros(0)
        classes = np.unique(seen).tolist()
        ap = np.zeros((len(classes), nt))
        for ci, c in enumerate(classes):
            cur = correct[correct[:, 1] == c]
            cur = cur[np.argsort(-cur[:, 0], kind='stable')][:, 2:]
            total = float((seen == c).sum())
            if len(cur) == 0 or total == 0:
                continue
            tp = np.cumsum(cur, axis=0)
            fp = np.cumsum(1 - cur, axis=0)
            recall = tp / (total + 1e-16)
            precision = tp / (tp + fp + 1e-16)
            for t in range(nt):
                ap[ci, t] = self.compute_ap(recall[:, t], precision[:, t])
        return ap.mean(axis=0) if len(classes) else np.zeros(nt), np.array(classes), ap
