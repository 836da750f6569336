"""Image path of the reference's demos/faster_rcnn/data_gen.py on the GPU: a sample there is ResizeByMax(input_size), in training a
HorizontalFlip with probability 0.5, Padding to the square canvas with 128, `/ 255` (data_gen.py:284-335) -- the yolov3_u demo's
pipeline without the mosaic, vertical flip and colour extras.  One kernel launch per batch (``fva_paste_resize_normalize``); the box
arithmetic stays in numpy float32 on the host, written as the reference writes it.
"""
import numpy as np

from ...pipeline_ops import PasteJob, pack_images, paste_batch
from ..yolov3_u.data_gen import DeviceAugmenter as _Base
from ..yolov3_u.data_gen import _hflip_boxes, resize_by_max_shape

__all__ = ['DeviceAugmenter']


class DeviceAugmenter(_Base):
    def batch(self, samples):
        """samples: list of (rgb uint8 [h,w,3], xyxy float32 [n,4], category [n], hflip) -> (images [B,3,S,S] float32 on the device,
        labels [T,6] = image, class, normalised xywh)."""
        S = self.input_size
        buf, offsets, shapes = pack_images([s[0] for s in samples])
        jobs, labels = [], []
        for i, ((h, w), (_, xyxy, cat, hflip)) in enumerate(zip(shapes, samples)):
            ratio, rh, rw = resize_by_max_shape(h, w, S)
            top, left = int((S - rh) // 2), int((S - rw) // 2)
            jobs.append(PasteJob(i, i, rh, rw, top, left, 1 if hflip else 0, 0))
            lab = np.asarray(xyxy, dtype=np.float32) * ratio
            if hflip:
                lab = _hflip_boxes(lab, rw)
            lab[:, [1, 3]] = lab[:, [1, 3]] + top
            lab[:, [0, 2]] = lab[:, [0, 2]] + left
            labels.append(self._labels(lab, cat))
        images = paste_batch(buf, offsets, shapes, jobs, len(samples), S, S, self.fill_value, self.table, self.device)
        return self._collate(images, labels)
