"""API mirror of the image path of the reference's demos/yolov3_u/data_gen.py, on the GPU.

The reference resizes, flips, pads and tiles (Mosaic01) decoded images one by one with cv2 on the CPU
(data_gen.py:42-131,171-216) and divides by 255 after ToTensor (:352-360).  ``DeviceAugmenter`` does the same steps for a
whole batch with two kernel launches (``fva_paste_resize_u8`` for the per-image ResizeByMax + flips,
``fva_paste_resize_normalize`` for Padding or Mosaic01 + ``/ 255``); the box arithmetic stays in numpy float32 on the
host, written as the reference writes it.  The probability-0.5 extras of the training path run on the device too
(csrc/colour.hip): Jitter (a further resize pass), HistEqualize (CLAHE on the luma) and HueSaturationValue (byte tables in HSV)
on the resized images, then on the mosaic OneOf[Blur, MedianBlur, GaussianBlur] (3x3), ChannelShuffle and ``/ 255`` in one launch.
Only image decoding stays on the host (DataLoader workers).
"""
import numpy as np
import torch

from ...pipeline_ops import (PasteJob, blur_shuffle_normalize, canvas_sources, clahe_hsv_, hsv_tables, pack_images, paste_batch,
                             paste_batch_u8, value_table)
from .utils.box import xyxy2xywhn

__all__ = ['DeviceAugmenter', 'resize_by_max_shape', 'HostImageBatch', 'BaseDataset', 'create_dataset']


def resize_by_max_shape(h, w, max_size):
    """(resize_ratio, resized_h, resized_w) of ResizeByMax (data_gen.py:50-55)"""
    ratio = max_size / max(h, w)
    return ratio, int(h * ratio), int(w * ratio)


def _hflip_boxes(xyxy, width):
    """HorizontalFlip's label half (data_gen.py:104-106): via xywh, centre mirrored"""
    xywh = np.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2, xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]], axis=1)
    xywh[:, 0] = width - xywh[:, 0]
    return np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], axis=1)


def _jitter_boxes(xyxy, h, w, new_hw):
    """Jitter's label half (data_gen.py:152-170): x by new_w / w, y by new_h / h; returns (boxes, new_h, new_w)"""
    nh, nw = int(new_hw[0]), int(new_hw[1])
    out = np.array(xyxy, dtype=np.float32, copy=True)
    out[:, [0, 2]] = out[:, [0, 2]] * (nw / w)
    out[:, [1, 3]] = out[:, [1, 3]] * (nh / h)
    return out, nh, nw


def _vflip_boxes(xyxy, height):
    xywh = np.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2, xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]], axis=1)
    xywh[:, 1] = height - xywh[:, 1]
    return np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2, xywh[:, 1] + xywh[:, 3] / 2], axis=1)


class DeviceAugmenter:
    def __init__(self, input_size, device='cuda', fill_value=128):
        self.input_size, self.device, self.fill_value = int(input_size), device, int(fill_value)
        self.table = value_table(single=True)

    def _labels(self, xyxy, cat):
        """xyxy (pixels of the input canvas) -> [n,6] rows 0, class, xc, yc, w, h (data_gen.py:362-367)"""
        xywhn = xyxy2xywhn(xyxy, self.input_size, self.input_size)
        rows = np.zeros([xywhn.shape[0], 6], dtype=np.float32)
        rows[:, 1] = cat
        rows[:, 2:] = xywhn
        return torch.tensor(rows, dtype=torch.float32)

    def _collate(self, images, labels):
        for i, l in enumerate(labels):
            l[:, 0] = i
        return images, torch.cat(labels, 0).to(self.device, non_blocking=True)

    def val_labels(self, samples):
        """the label half of val_batch alone (host arithmetic; used by the DataLoader-side collate of BaseDataset)"""
        S, labels = self.input_size, []
        for img, xyxy, cat in samples:
            h, w = img.shape[:2]
            ratio, rh, rw = resize_by_max_shape(h, w, S)
            top, left = int((S - rh) // 2), int((S - rw) // 2)
            lab = np.asarray(xyxy, dtype=np.float32) * ratio
            lab[:, [1, 3]] = lab[:, [1, 3]] + top
            lab[:, [0, 2]] = lab[:, [0, 2]] + left
            labels.append(self._labels(lab, cat))
        return labels

    def val_batch(self, samples, packed=None):
        """samples: list of (rgb uint8 [h,w,3], xyxy float32 [n,4], category [n]).  ResizeByMax(input_size) -> Padding(128) ->
        / 255 (data_gen.py:42-92,356-360) for the whole batch in one launch."""
        S = self.input_size
        buf, offsets, shapes = packed if packed is not None else pack_images([s[0] for s in samples])
        jobs, labels = [], []
        for i, ((h, w), (_, xyxy, cat)) in enumerate(zip(shapes, samples)):
            ratio, rh, rw = resize_by_max_shape(h, w, S)
            top, left = int((S - rh) // 2), int((S - rw) // 2)
            jobs.append(PasteJob(i, i, rh, rw, top, left))
            lab = np.asarray(xyxy, dtype=np.float32) * ratio
            lab[:, [1, 3]] = lab[:, [1, 3]] + top
            lab[:, [0, 2]] = lab[:, [0, 2]] + left
            labels.append(self._labels(lab, cat))
        images = paste_batch(buf, offsets, shapes, jobs, len(samples), S, S, self.fill_value, self.table, self.device)
        return self._collate(images, labels)

    def train_labels(self, groups):
        """the label half of train_batch alone (host arithmetic, same expressions in the same order)"""
        S, labels = self.input_size, []
        cx = cy = S // 2
        for g in groups:
            boxes, cats = [], []
            for idx, t in enumerate(g):
                img, xyxy, cat, hf, vf = t[:5]
                h, w = img.shape[:2]
                xyxy = np.asarray(xyxy, dtype=np.float32)
                jit = (t[5] or {}).get('jitter') if len(t) > 5 else None
                if jit is not None:
                    xyxy, h, w = _jitter_boxes(xyxy, h, w, jit)
                ratio, rh, rw = resize_by_max_shape(h, w, S)
                lab = xyxy * ratio
                if hf:
                    lab = _hflip_boxes(lab, rw)
                if vf:
                    lab = _vflip_boxes(lab, rh)
                ratio2, th, tw = resize_by_max_shape(rh, rw, S // 2)
                x0 = cx - tw if idx in (0, 2) else cx
                y0 = cy - th if idx in (0, 1) else cy
                lab = lab * ratio2
                lab[:, [0, 2]] = lab[:, [0, 2]] + x0
                lab[:, [1, 3]] = lab[:, [1, 3]] + y0
                boxes.append(lab)
                cats.append(np.asarray(cat))
            xyxy = np.clip(np.concatenate(boxes, axis=0), 0, S - 1)
            labels.append(self._labels(xyxy, np.concatenate(cats, axis=0).reshape(-1)))
        return labels

    def train_batch(self, groups, packed=None, post=None):
        """groups: per output image a list of FOUR (rgb, xyxy, category, hflip, vflip[, extras]) -- the sample and its three random
        companions (data_gen.py:338-345); extras = {'jitter': (new_h, new_w), 'clahe': bool, 'hsv_gains': [3]} (the drawn
        probability-0.5 transforms of preprocess_image_label); post: per output image {'blur': 0..3, 'perm': (c0, c1, c2)}.  Pass 1 (uint8): ResizeByMax(input_size) + flips of all 4*B images; pass 2:
        Mosaic01 (each tile resized again to input_size // 2 on its longer side, the four meet at the centre) + / 255."""
        S = self.input_size
        flat = [t for g in groups for t in g]
        buf, offsets, shapes = packed if packed is not None else pack_images([t[0] for t in flat])
        extras = [(t[5] if len(t) > 5 and t[5] else {}) for t in flat]
        pitches0 = None
        if any(e.get('jitter') is not None for e in extras):
            # Jitter (before ResizeByMax): one more resize pass into canvases of the largest jittered size; images that were not
            # drawn pass through at their own size (an identity resize copies bytes exactly)
            sizes = [tuple(int(v) for v in e['jitter']) if e.get('jitter') is not None else (h, w) for e, (h, w) in zip(extras, shapes)]
            JH, JW = max(s[0] for s in sizes), max(s[1] for s in sizes)
            jobs0 = [PasteJob(i, i, sizes[i][0], sizes[i][1], 0, 0) for i in range(len(flat))]
            jit = paste_batch_u8(buf, offsets, shapes, jobs0, len(flat), JH, JW, self.fill_value, self.device)
            buf = jit.view(-1)
            offsets, shapes, pitches0 = canvas_sources(len(flat), JH, JW, sizes)
        jobs1, mid_shapes, mid_boxes = [], [], []
        for i, ((h, w), t) in enumerate(zip(shapes, flat)):
            xyxy, cat, hf, vf = t[1:5]
            xyxy = np.asarray(xyxy, dtype=np.float32)
            if extras[i].get('jitter') is not None:
                oh, ow = t[0].shape[:2]
                xyxy, _, _ = _jitter_boxes(xyxy, oh, ow, extras[i]['jitter'])
            ratio, rh, rw = resize_by_max_shape(h, w, S)
            jobs1.append(PasteJob(i, i, rh, rw, 0, 0, hf, vf))
            lab = xyxy * ratio
            if hf:
                lab = _hflip_boxes(lab, rw)
            if vf:
                lab = _vflip_boxes(lab, rh)
            mid_shapes.append((rh, rw))
            mid_boxes.append(lab)
        mid = paste_batch_u8(buf, offsets, shapes, jobs1, len(flat), S, S, self.fill_value, self.device, pitches=pitches0)
        # HistEqualize, then HueSaturationValue, on the resized + flipped images (data_gen.py:304-310), in place
        clahe_hsv_(mid, mid_shapes, [bool(e.get('clahe')) for e in extras],
                   [hsv_tables(e['hsv_gains']) if e.get('hsv_gains') is not None else None for e in extras])
        m_off, m_shapes, m_pitch = canvas_sources(len(flat), S, S, mid_shapes)
        jobs2, labels = [], []
        cx = cy = S // 2
        for b, g in enumerate(groups):
            boxes, cats = [], []
            for idx in range(4):
                i = b * 4 + idx
                rh, rw = mid_shapes[i]
                ratio, th, tw = resize_by_max_shape(rh, rw, S // 2)
                x0 = cx - tw if idx in (0, 2) else cx
                y0 = cy - th if idx in (0, 1) else cy
                jobs2.append(PasteJob(i, b, th, tw, y0, x0))
                lab = mid_boxes[i] * ratio
                lab[:, [0, 2]] = lab[:, [0, 2]] + x0
                lab[:, [1, 3]] = lab[:, [1, 3]] + y0
                boxes.append(lab)
                cats.append(np.asarray(g[idx][2]))
            xyxy = np.clip(np.concatenate(boxes, axis=0), 0, S - 1)
            labels.append(self._labels(xyxy, np.concatenate(cats, axis=0).reshape(-1)))
        if post is not None and any(p.get('blur', 0) or tuple(p.get('perm', (0, 1, 2))) != (0, 1, 2) for p in post):
            # the albumentations stage on the mosaic: OneOf[Blur, MedianBlur, GaussianBlur] 3x3, ChannelShuffle, then / 255
            mosaic = paste_batch_u8(mid.view(-1), m_off, m_shapes, jobs2, len(groups), S, S, self.fill_value, self.device, pitches=m_pitch)
            images = blur_shuffle_normalize(mosaic, [p.get('blur', 0) for p in post], [tuple(p.get('perm', (0, 1, 2))) for p in post],
                                            self.table)
        else:
            images = paste_batch(mid.view(-1), m_off, m_shapes, jobs2, len(groups), S, S, self.fill_value, self.table, self.device,
                                 pitches=m_pitch)
        return self._collate(images, labels)


# ---------------------------------------------------------------------------------------------------------------------
# The reference's dataset surface (data_gen.py:247-394: BaseDataset, collate_fn, create_dataset), so that its train.py builds its
# loaders unchanged:  DataLoader(dataset=create_dataset(dir, size, mode), collate_fn=dataset.collate_fn, pin_memory=True,
# num_workers=N)  ->  for images, target in loader: images = images.cuda(non_blocking=True) ...
# Workers only decode files and draw the random numbers (no GPU context in a forked worker); collate_fn packs the decoded bytes
# and computes the labels (host arithmetic of DeviceAugmenter's plan); the image batch it returns is a ``HostImageBatch`` whose
# ``.cuda()`` / ``.to(device)`` uploads the bytes and runs the resize / flip / pad / mosaic / "/255" kernels -- the train loop's
# own ``images.cuda(non_blocking=True)`` is what launches them.
class HostImageBatch:
    """Decoded images of one batch (packed uint8 bytes) + the geometric plan; becomes the [B,3,S,S] float tensor on the device."""

    def __init__(self, mode, input_size, fill_value, samples, post=None):
        self.mode, self.input_size, self.fill_value, self.samples, self.post = mode, int(input_size), int(fill_value), samples, post
        self.buf, self.offsets, self.shapes = pack_images([s[0] for s in (samples if mode != 'train' else [t for g in samples for t in g])], pin=False)

    def __len__(self):
        return len(self.samples)

    def size(self, dim=None):
        shape = (len(self.samples), 3, self.input_size, self.input_size)
        return shape if dim is None else shape[dim]

    shape = property(lambda self: self.size())

    def pin_memory(self):
        self.buf = self.buf.pin_memory()
        return self

    def to(self, device, non_blocking=False):
        aug = DeviceAugmenter(self.input_size, device, self.fill_value)
        packed = (self.buf, self.offsets, self.shapes)
        if self.mode == 'train':
            return aug.train_batch(self.samples, packed=packed, post=self.post)[0]
        return aug.val_batch(self.samples, packed=packed)[0]

    def cuda(self, device=None, non_blocking=False):
        return self.to(torch.device('cuda', torch.cuda.current_device()) if device is None else device, non_blocking)


class BaseDataset(torch.utils.data.Dataset):
    """samples: array of (image path, label path); label files hold ``class xmin ymin xmax ymax`` per line (data_gen.py:269-282)."""

    def __init__(self, samples, input_size, mode, fill_value=128):
        self.samples, self.input_size, self.mode, self.fill_value = samples, int(input_size), mode, fill_value
        self._planner = DeviceAugmenter(self.input_size, 'cpu', fill_value)       # label arithmetic only

    def __len__(self):
        return len(self.samples)

    def load_image(self, img_path):
        from ...datasets.detection_dataloader import _decode_rgb
        return _decode_rgb(img_path)

    def load_label(self, label_path):
        rows = []
        with open(label_path, 'r') as f:
            for line in f:
                if line.strip():
                    rows.append(line.split())
        return np.array(rows, dtype=np.float32).reshape([-1, 5])

    def _one(self, img_path, label_path):
        import random
        image, label = self.load_image(img_path), self.load_label(label_path)
        if self.mode != 'train':
            return image, label[:, 1:], label[:, 0], False, False
        # the draws of preprocess_image_label, in its order (data_gen.py:293-310)
        extras = {}
        if random.random() <= 0.5:
            h, w = image.shape[:2]
            rnd = lambda a, b: np.random.rand() * (b - a) + a
            extras['jitter'] = (int(h * rnd(0.7, 1.3)), int(w * rnd(0.7, 1.3)))
        hf = random.random() <= 0.5
        vf = random.random() <= 0.5
        if random.random() <= 0.5:
            extras['clahe'] = True
        if random.random() <= 0.5:
            extras['hsv_gains'] = np.random.uniform(-1, 1, 3) * [0.015, 0.7, 0.4] + 1
        return image, label[:, 1:], label[:, 0], hf, vf, extras

    def __getitem__(self, idx):
        import random
        first = self._one(self.samples[idx][0], self.samples[idx][1])
        if self.mode != 'train':
            return first[:3]
        group = [first]
        for _ in range(3):
            r = random.choice(self.samples)
            group.append(self._one(r[0], r[1]))
        # the albumentations Compose on the mosaic (data_gen.py:26-33): OneOf of three blurs with p = 0.5, ChannelShuffle with p = 0.5
        post = {'blur': random.randint(1, 3) if random.random() < 0.5 else 0,
                'perm': tuple(random.sample(range(3), 3)) if random.random() < 0.5 else (0, 1, 2)}
        return group, post

    def collate_fn(self, batch):
        """-> (HostImageBatch, targets [T,6] float32 on the host): the reference's (images, labels) pair (data_gen.py:366-371)"""
        post = None
        if self.mode == 'train':
            batch, post = [b[0] for b in batch], [b[1] for b in batch]
        images = HostImageBatch(self.mode, self.input_size, self.fill_value, list(batch), post)
        plan = self._planner
        labels = plan.train_labels(batch) if self.mode == 'train' else plan.val_labels(batch)
        for i, l in enumerate(labels):
            l[:, 0] = i
        return images, torch.cat(labels, 0)


def create_dataset(base_dir, input_size, mode):
    """<base_dir>/images/* with <base_dir>/labels/<name>.txt (data_gen.py:373-394); sample order shuffled with numpy's generator"""
    import os
    label_dir, image_dir = os.path.join(base_dir, 'labels'), os.path.join(base_dir, 'images')
    samples = []
    for img_name in sorted(os.listdir(image_dir)):
        samples.append((os.path.join(image_dir, img_name), os.path.join(label_dir, f"{img_name.split('.')[0]}.txt")))
    samples = np.array(samples).reshape([-1, 2])
    np.random.shuffle(samples)
    print(f'total samples: {len(samples)}')
    return BaseDataset(samples, input_size, mode)
