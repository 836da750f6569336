"""Anchor k-means on the label widths / heights -- host-side utility with the reference's names (detection/tools/ANCHOR.py:7-110,
demos/yolov3_u/utils/anchor_generator.py): ``KMeans(xs, k).fit(iters)`` with distance 1 - wh_iou and ``AnchorGenerator(loaders,
k, iters, ...).get_anchors()`` returning the k centres in input pixels, largest area first.  Runs once before training, on the
CPU (numpy); not part of the per-step hot path.  Deterministic given numpy's global seed, like the reference (it shuffles the
samples with np.random.shuffle and takes the first k as initial centres)."""
import os

import numpy as np

__all__ = ['KMeans', 'AnchorGenerator']


def _wh_iou(a, b):
    """[N,2] x [K,2] -> [N,K] IoU of boxes that share a corner"""
    inter = np.minimum(a[:, None, 0], b[None, :, 0]) * np.minimum(a[:, None, 1], b[None, :, 1])
    return inter / (a[:, None, 0] * a[:, None, 1] + b[None, :, 0] * b[None, :, 1] - inter)


class KMeans:
    def __init__(self, xs, k=9):
        self.samples = np.array(xs, dtype=np.float64).reshape(-1, 2)
        self.num_samples, self.k = len(self.samples), k
        np.random.shuffle(self.samples)
        self.centers = self.samples[:k].copy()
        self.categories = None

    def cal_distance(self, xs, centers):
        return 1.0 - _wh_iou(xs, centers)

    def _fit(self):
        self.categories = np.argmin(self.cal_distance(self.samples, self.centers), axis=1) + 1
        for c in range(self.k):
            members = self.samples[self.categories == c + 1]
            if len(members):                       # an empty cluster keeps its centre
                self.centers[c] = members.mean(axis=0)

    def fit(self, iters):
        for _ in range(iters):
            self._fit()
        return self.centers, self.categories


class AnchorGenerator:
    def __init__(self, data_loaders, k=9, iters=100, num_workers=1, plot=False, cache='./cache', use_cache=False, save_dir=None):
        self.data_loaders, self.k, self.iters, self.num_workers = data_loaders, k, iters, num_workers
        self.cache = os.path.join(save_dir if save_dir is not None else cache, 'anchor.txt')
        self.use_cache, self.plot = use_cache, plot
        self.input_height = self.input_width = None

    def load_data(self):
        wh = []
        for loader in self.data_loaders:
            for images, labels in loader:
                self.input_height, self.input_width = images.shape[2:]
                wh.append(labels[:, 4:].detach().cpu().numpy())
        return np.concatenate(wh, axis=0)

    def load_cache(self):
        with open(self.cache) as f:
            return [[float(v) for v in line.split()] for line in f if line.strip()]

    def get_anchors(self):
        if self.use_cache:
            return np.array(self.load_cache(), dtype=np.float64).reshape(-1, 2)
        wh = self.load_data().astype(np.float32).reshape(-1, 2)
        centers, _ = KMeans(wh, self.k).fit(self.iters)
        centers = np.array(sorted(centers.tolist(), key=lambda c: -c[0] * c[1]), dtype=np.float64).reshape(-1, 2)
        centers[:, 0] *= self.input_width                # labels are normalised to the network input
        centers[:, 1] *= self.input_height
        os.makedirs(os.path.dirname(self.cache) or '.', exist_ok=True)
        with open(self.cache, 'w') as f:
            for w, h in centers:
                f.write(f'{w} {h}\n')
        return centers
