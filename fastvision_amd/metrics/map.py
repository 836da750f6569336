"""mean average precision -- API mirror of the reference's metrics/map.py (class CalculateMAP).

Host-side bookkeeping like the reference's (numpy), with the target x prediction IoU matrix taken from the device
kernel behind ``cal_iou_batch``.  ``np.float`` / ``np.long`` of the reference (map.py:34,81,106,126) are spelled
``float`` / ``np.int64``: the aliases were removed from numpy 1.24.
"""
import numpy as np
import torch

from ..detection.tools import cal_iou_batch

__all__ = ['CalculateMAP']


class CalculateMAP:
    def __init__(self, map_iou_values):
        """map_iou_values: e.g. np.linspace(0.5, 0.95, 10)"""
        self.map_iou_values = map_iou_values
        self.correct_all_images = []
        self.seen_all_targets_cls = []

    def process_one(self, y_pred, y_true):
        """y_pred [M,6] = category, confidence, xmin, ymin, xmax, ymax; y_true [N,5] = category, xmin, ymin, xmax, ymax.
        Appends the image's [M, 2 + n_iou] matrix (conf, class, correct-at-each-threshold) -- map.py:17-83."""
        n_iou = len(self.map_iou_values)
        correct = np.zeros([y_pred.size(0), 2 + n_iou], dtype=float)
        predict_cls, predict_conf, predict_xyxy = y_pred[:, 0], y_pred[:, 1], y_pred[:, 2:]
        target_cls, target_xyxy = y_true[:, 0], y_true[:, 1:]
        if target_cls.size(0) != 0:
            self.seen_all_targets_cls.append(target_cls.detach().cpu().numpy())
        if y_pred.size(0) == 0:
            return
        if target_cls.size(0) != 0:
            iou = cal_iou_batch(target_xyxy.contiguous(), predict_xyxy.contiguous(), mode='xyxy')   # [N,M]
            matched = (iou > self.map_iou_values[0]) & (target_cls[:, None] == predict_cls)
            iou_h, matched_h = iou.detach().cpu().numpy(), matched.detach().cpu().numpy()
            tcls_h, pconf_h = target_cls.detach().cpu().numpy(), predict_conf.detach().cpu().numpy()
            ti, pi = np.where(matched_h)
            m = np.stack([ti.astype(np.float32), pi.astype(np.float32), iou_h[ti, pi].astype(np.float32),
                          tcls_h[ti].astype(np.float32), pconf_h[pi].astype(np.float32)], axis=1).reshape(-1, 5)
            m = m[np.argsort(-m[:, 2]), ...]                            # best IoU first
            m = m[np.unique(m[:, 1], return_index=True)[1], ...]        # one target per prediction
            m = m[np.unique(m[:, 0], return_index=True)[1], ...]        # one prediction per target
            correct[m[:, 1].astype(np.int64), 2:] = m[:, 2:3] > self.map_iou_values
        correct[:, 0] = y_pred[:, 1].detach().cpu().numpy()
        correct[:, 1] = y_pred[:, 0].detach().cpu().numpy()
        self.correct_all_images.append(correct)

    def compute_ap(self, recall, precision, method='coco'):
        m_recall = np.concatenate(([0.0], recall, [1.0]))
        m_precision = np.concatenate(([1.0], precision, [0.0]))
        envelope = np.flip(np.maximum.accumulate(m_precision[::-1]))
        if method == 'coco':                                            # 101-point interpolation
            x = np.linspace(0, 1, 101)
            return np.trapezoid(np.interp(x, m_recall, envelope), x)
        if method == 'voc2009':
            i = np.where(m_recall[1:] != m_recall[:-1])[0]
            return np.sum((m_recall[i + 1] - m_recall[i]) * envelope[i + 1])
        raise Exception('Not complete')                                 # voc2007, as in the reference

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
        """(mAP per IoU threshold, mAP per class, class ids) over everything processed so far (map.py:121-141)."""
        correct = np.concatenate(self.correct_all_images, axis=0)
        seen = np.concatenate(self.seen_all_targets_cls, axis=0)
        uniq = np.unique(seen).tolist()
        ap = np.zeros((len(uniq), len(self.map_iou_values)), dtype=float)
        for c in uniq:
            cur = correct[correct[:, 1] == c, ...]
            cur = cur[np.argsort(-cur[:, 0]), ...]
            ap[uniq.index(c)] = self._ap_per_class(np.sum(seen == c), cur[:, 2:])
        return np.mean(ap, axis=0), np.mean(ap, axis=1), [int(c) for c in uniq]
