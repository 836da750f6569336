"""Detection data loader -- API mirror of the reference's datasets/detection_dataloader.py with the per-sample image
work moved to the GPU.

The reference's ``BaseDataset.__getitem__`` decodes one image and then resizes, pads, flips, normalises and transposes
it on the CPU (:44-96); ``collate_fn`` stacks the float32 results (:98-103).  Here ``__getitem__`` only decodes (and
draws the same three random numbers per sample, in the same order); ``collate_fn`` packs the decoded bytes of the whole
batch and ONE kernel launch (``fva_paste_resize_normalize``) produces the [B,3,S,S] float32 batch on the device.  The
label arithmetic stays on the host in numpy float32, operation for operation as the reference does it.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from ..detection.tools import xyxy2xywhn
from ..pipeline_ops import PasteJob, pack_images, paste_batch, value_table

__all__ = ['BaseDataset', 'DeviceLoader', 'create_dataloader', 'load_samples', 'letterbox_geometry', 'show_dataset']

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
PAD_VALUE = 114


def letterbox_geometry(ori_height, ori_width, input_size):
    """(resized_h, resized_w), (ratio_h, ratio_w), (top, left, bottom, right) of load_image + Padding
    (detection_dataloader.py:44-58, datasets/common/padding.py:3-19: note its round(x -/+ 0.1))."""
    if isinstance(input_size, int):
        ratio = input_size / max(ori_height, ori_width)
        ratio_height, ratio_width = ratio, ratio
        target_h, target_w = input_size, input_size
    else:
        target_h, target_w = input_size[0], input_size[1]
        ratio_height, ratio_width = target_h / ori_height, target_w / ori_width
    rw, rh = int(ori_width * ratio_width), int(ori_height * ratio_height)
    ph, pw = (target_h - rh) / 2, (target_w - rw) / 2
    top, bottom = int(round(ph - 0.1)), int(round(ph + 0.1))
    left, right = int(round(pw - 0.1)), int(round(pw + 0.1))
    return (rh, rw), (ratio_height, ratio_width), (top, left, bottom, right)


def _decode_rgb(path):
    try:
        import cv2
        return cv2.cvtColor(cv2.imread(path), cv2.COLOR_BGR2RGB)
    except ImportError:
        from PIL import Image
        return np.array(Image.open(path).convert('RGB'))


class BaseDataset(Dataset):
    def __init__(self, samples, input_size, max_det, flip_p=(0.5, 0.1)):
        self.samples, self.max_det, self.input_size = samples, max_det, input_size
        self.input_height, self.input_width = (input_size, input_size) if isinstance(input_size, int) else (input_size[0], input_size[1])
        self.flip_p = flip_p                                   # HorizontalFlip(p=0.5), VerticalFlip(p=0.1) (:33-37)

    def __len__(self):
        return len(self.samples)

    def load_image(self, img_path, mode='rgb'):
        img = _decode_rgb(img_path)
        return img if mode == 'rgb' else img[:, :, ::-1]

    def labels_for(self, annotations, ori_hw, hflip, vflip):
        """[n,5] = class, xc, yc, w, h (normalised) after resize, padding and flips -- :60-66,85-87 and the label halves of
        HorizontalFlip / VerticalFlip (augmentation.py:316-318,342-344), in float32 numpy like the reference."""
        _, (ratio_h, ratio_w), pad = letterbox_geometry(ori_hw[0], ori_hw[1], self.input_size)
        label = np.array(annotations, dtype=np.float32).reshape([-1, 5])
        label[:, 1] = label[:, 1] * ratio_w + pad[1]
        label[:, 2] = label[:, 2] * ratio_h + pad[0]
        label[:, 3] = label[:, 3] * ratio_w + pad[1]
        label[:, 4] = label[:, 4] * ratio_h + pad[0]
        label[:, 1:] = xyxy2xywhn(label[:, 1:], heigth=self.input_height, width=self.input_width)
        box = label[:, 1:]
        if len(box):            # the reference indexes label[0] and fails on an image without boxes; nothing to mirror then
            if hflip:
                box[:, 0] = (1 if (box[0, 2] < 1 and box[0, 3] < 1) else self.input_width) - box[:, 0]
            if vflip:
                box[:, 1] = (1 if (box[0, 2] < 1 and box[0, 3] < 1) else self.input_height) - box[:, 1]
        return label

    def __getitem__(self, idx):
        """Host half of a sample: decoded RGB bytes, its label rows and the flip decisions (three draws, as the reference's
        HorizontalFlip, VerticalFlip and Normalization(p=1.0) make them)."""
        img_path, annotations = self.samples[idx][0], self.samples[idx][1]
        rgb = self.load_image(os.path.join(img_path), mode='rgb')
        hflip = random.random() <= self.flip_p[0]
        vflip = random.random() <= self.flip_p[1]
        random.random()
        label = self.labels_for(annotations, rgb.shape[:2], hflip, vflip)
        labels_out = torch.zeros([len(label), 6], dtype=torch.float32)
        labels_out[:, 1:] = torch.from_numpy(label)
        return rgb, labels_out, (hflip, vflip)

    def jobs_for(self, shapes, flips):
        jobs = []
        for i, ((h, w), (hf, vf)) in enumerate(zip(shapes, flips)):
            (rh, rw), _, (top, left, bottom, right) = letterbox_geometry(h, w, self.input_size)
            # flipping the padded canvas = flipping the image and mirroring its position (the border is constant)
            jobs.append(PasteJob(i, i, rh, rw, bottom if vf else top, right if hf else left, hf, vf))
        return jobs

    def collate_host(self, batch):
        """Worker-side half of collate_fn: one byte buffer + the label table.  No GPU work and NO pinning here: this runs in
        forked DataLoader workers, where pinning would need a GPU context per worker (and the pin would be lost anyway when the
        tensor crosses the worker queue into shared memory).  The main process pins: DataLoader(pin_memory=True)."""
        rgbs, labels, flips = zip(*batch)
        for i, l in enumerate(labels):
            l[:, 0] = i
        buf, offsets, shapes = pack_images(rgbs, pin=False)
        return buf, offsets, shapes, list(flips), torch.cat(labels, 0)

    def to_device(self, host_batch, device):
        """Main-process half: upload the bytes, one launch -> ([B,3,S,S] float32, [T,6] float32) on ``device``."""
        buf, offsets, shapes, flips, labels = host_batch
        table = value_table(IMAGENET_MEAN, IMAGENET_STD)
        images = paste_batch(buf, offsets, shapes, self.jobs_for(shapes, flips), len(shapes), self.input_height, self.input_width,
                             PAD_VALUE, table, device)
        return images, labels.to(device, non_blocking=True)

    def collate_fn(self, batch, device='cuda'):
        return self.to_device(self.collate_host(batch), device)


class DeviceLoader:
    """Iterates a host DataLoader (workers decode and pack) and finishes every batch on the GPU."""

    def __init__(self, loader, dataset, device):
        self.loader, self.dataset, self.device = loader, dataset, device

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for host_batch in self.loader:
            yield self.dataset.to_device(host_batch, self.device)


def _load_samples(img_name, images_dir, labels_dir, samples):
    """one image + its label file 'class xmin ymin xmax ymax' per line (:105-118)"""
    img_id = img_name.split('.')[0]
    labels = []
    with open(os.path.join(labels_dir, f'{img_id}.txt'), 'r') as f:
        for line in f.readlines():
            category_id, xmin, ymin, xmax, ymax = line.strip().split()
            labels.append((float(category_id), float(xmin), float(ymin), float(xmax), float(ymax)))
    samples.append((os.path.join(images_dir, img_name), labels))


def load_samples(data_dir, prefix, num_workers=0, cache=None, use_cache=False):
    """(:120-158) without the multiprocessing pool: parsing label files is not the bottleneck of this path."""
    if use_cache:
        with open(os.path.join(cache, f'{prefix}.txt'), 'r') as f:
            return eval(f.read())
    images_dir, labels_dir = os.path.join(data_dir, 'images'), os.path.join(data_dir, 'labels')
    samples = []
    for img_name in sorted(os.listdir(images_dir)):
        _load_samples(img_name, images_dir, labels_dir, samples)
    if cache:
        os.makedirs(cache, exist_ok=True)
        with open(os.path.join(cache, f'{prefix}.txt'), 'w') as f:
            f.write(str(samples))
    return samples


def create_dataloader(prefix, data_dir, batch_size, input_size, device, num_workers=0, cache='./cache', use_cache=False, shuffle=True,
                      pin_memory=True, drop_last=False, max_det=200):
    """Same signature as the reference (:160-178); yields device tensors."""
    samples = load_samples(data_dir, prefix, num_workers, cache, use_cache)
    dataset = BaseDataset(samples, input_size, max_det)
    device = torch.device(device) if not isinstance(device, torch.device) else device
    loader = DataLoader(dataset=dataset, batch_size=batch_size, shuffle=shuffle, pin_memory=bool(pin_memory) and device.type == 'cuda',
                        drop_last=drop_last, num_workers=num_workers, collate_fn=dataset.collate_host)
    return DeviceLoader(loader, dataset, device)


def show_dataset(prefix, data_dir, category_names, num_workers=0, cache='./cache', use_cache=False, out_dir=None, limit=None):
    """The reference's label viewer (detection_dataloader.py:176-190) without a display: every sample's boxes are drawn into the
    decoded image and written as a binary PPM under ``out_dir`` (default <cache>/show_<prefix>).  Returns the paths written."""
    from ..detection.plot import draw_box_label
    samples = load_samples(data_dir, prefix, num_workers, cache, use_cache)
    out_dir = out_dir or os.path.join(cache, f'show_{prefix}')
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for img_path, labels in samples[:limit]:
        img = np.ascontiguousarray(_decode_rgb(img_path))
        for category_idx, xmin, ymin, xmax, ymax in labels:
            draw_box_label(img, (xmin, ymin, xmax, ymax), text=str(category_names[int(category_idx)]), line_color=int(category_idx), bgr=False)
        path = os.path.join(out_dir, os.path.splitext(os.path.basename(img_path))[0] + '.ppm')
        with open(path, 'wb') as f:
            f.write(f'P6 {img.shape[1]} {img.shape[0]} 255\n'.encode())
            f.write(img.tobytes())
        written.append(path)
    return written
