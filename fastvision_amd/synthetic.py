"""Synthetic COCO-shaped batches: identical on the CPU oracle and the GPU path.

Label contract follows the reference's collate functions (datasets/detection_dataloader.py:98-103,
demos/yolov3_u/data_gen.py:366-371): targets [T,6] = [image_idx, class_idx, xc, yc, w, h], boxes
normalised to [0,1], rows sorted by image.  Every image gets >= 1 box and centres stay strictly
inside the image so the demo loss (unclamped grid indices, lossv3.py:56,94) is defined.
"""
import math

import torch

COCO_ANCHORS_PX = [[116, 90, 156, 198, 373, 326], [30, 61, 62, 45, 59, 119], [10, 13, 16, 30, 33, 23]]
LEVEL_STRIDES = [32, 16, 8]


def coco_anchors_px():
    """[9,2] pixel anchors, largest level first (demos/yolov3_u/train.py:60-62 numerators)."""
    return torch.tensor(COCO_ANCHORS_PX, dtype=torch.float32).view(-1, 2)


def coco_anchors_feature():
    """Three [3,2] feature-scale anchor tensors (/32, /16, /8) as demos/yolov3_u/train.py:60-62 builds them."""
    a = coco_anchors_px().view(3, 3, 2)
    return tuple(a[i] / s for i, s in enumerate(LEVEL_STRIDES))


def synthetic_batch(batch, size, num_classes=80, seed=1234, rank=0, max_boxes=40):
    """images [B,3,S,S] fp32 in [0,1) and targets [T,6] fp32, both on CPU, from one seeded CPU generator."""
    g = torch.Generator('cpu').manual_seed(seed + rank)
    images = torch.rand(batch, 3, size, size, generator=g)
    rows = []
    for img in range(batch):
        n = int(torch.poisson(torch.tensor([7.0]), generator=g).clamp(1, max_boxes).item())
        cls = torch.randint(0, num_classes, (n,), generator=g).float()
        lo, hi = math.log(0.02), math.log(0.8)
        wh = torch.exp(lo + (hi - lo) * torch.rand(n, 2, generator=g))
        u = torch.rand(n, 2, generator=g)
        xy = wh / 2 + u * (1 - wh)
        xy = xy.clamp(max=1.0 - 1e-4)
        rows.append(torch.cat([torch.full((n, 1), float(img)), cls.view(n, 1), xy, wh], dim=1))
    return images, torch.cat(rows, 0)
