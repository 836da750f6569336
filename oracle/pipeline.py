"""CPU restatement of the input side of the path (scope row f-3): decoded image -> network input batch.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Parity status
  * everything AROUND the resampler -- resize geometry, Padding offsets (its round(x -/+ 0.1) rule), label scaling /
    shifting / xyxy2xywhn, flips of image and labels, Normalization (float64 arithmetic rounded to float32), HWC->CHW,
    collate -- is PINNED by tests/golden/pipeline.npz, captured from the reference's own classes
    (datasets/detection_dataloader.py BaseDataset, datasets/common/{augmentation,padding}.py; demos/yolov3_u/data_gen.py
    ResizeByMax / Padding / flips / Mosaic01) run in the build container (oracle/make_golden.py pipeline).
  * cv2.resize / cv2.copyMakeBorder / cv2.flip are OpenCV's (opencv-python pinned 4.5.5.62 by the reference's
    requirements; the module is absent from this image and from /root/reference) -> PARITY UNPINNED for the resampler.
    resize_linear_u8() restates OpenCV's published 8-bit INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp, 4.5.x):
      - pixel-centre mapping fx = (dx + 0.5) * scale - 0.5 in float32, scale = 1 / (dst / src) in float64,
        sx = floor(fx), taps clamped to the image (fx = 0 beyond the last column);
      - 11-bit fixed-point weights saturate_cast<short>(w * 2048) (round half to even);
      - horizontal pass in int32, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
        (the uchar specialisation of VResizeLinear, identical in its SIMD and scalar forms);
      - an exact 2x decimation in both axes is rerouted to INTER_AREA: (s00 + s01 + s10 + s11 + 2) >> 2.
    The golden vectors ran the reference classes with these three functions standing in for cv2's.

Reference files restated: datasets/detection_dataloader.py:19-103, datasets/common/padding.py:3-22,
datasets/common/augmentation.py:298-376, detection/tools/BOX.py; demos/yolov3_u/data_gen.py:42-131,171-216,332-371.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------ OpenCV stand-ins
def _taps(dst, src):
    """per destination index: first source tap, int16 weights (w0, w1) in units of 1/2048"""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    return s, np.minimum(s + 1, src - 1), w0, w1


def resize_linear_u8(img, dsize):
    """cv2.resize(img, (width, height), interpolation=cv2.INTER_LINEAR) for uint8 HWC images (see module docstring)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    if dw <= 0 or dh <= 0:
        raise ValueError('resize: empty destination')
    if (dw, dh) == (sw, sh):
        return img.copy()
    src = img.astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:                       # INTER_LINEAR -> INTER_AREA fast path
        out = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8)
    x0, x1, a0, a1 = _taps(dw, sw)
    y0, y1, b0, b1 = _taps(dh, sh)
    shape = (1, dw) + (1,) * (img.ndim - 2)
    rows = src[:, x0] * a0.reshape(shape) + src[:, x1] * a1.reshape(shape)          # [sh, dw, (c)] int, <= 255 * 2048
    vshape = (dh, 1) + (1,) * (img.ndim - 2)
    out = (((b0.reshape(vshape) * (rows[y0] >> 4)) >> 16) + ((b1.reshape(vshape) * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def copy_make_border_constant(img, top, bottom, left, right, value):
    """cv2.copyMakeBorder(..., cv2.BORDER_CONSTANT, value=value)"""
    h, w = img.shape[:2]
    out = np.empty((h + top + bottom, w + left + right) + img.shape[2:], dtype=img.dtype)
    out[...] = np.asarray(value, dtype=img.dtype)[:img.shape[2]] if img.ndim == 3 else value
    out[top:top + h, left:left + w] = img
    return out


def flip(img, code):
    """cv2.flip: code 1 = horizontal, 0 = vertical"""
    return img[:, ::-1].copy() if code == 1 else img[::-1].copy()


# ------------------------------------------------------------------------------------------------ library pipeline
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(1, 1, 3)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(1, 1, 3)


def library_geometry(ori_h, ori_w, input_size):
    """detection_dataloader.py:44-58 + padding.py:3-19 for an int input_size: resized (h, w), ratios, (top, left)."""
    ratio = input_size / max(ori_h, ori_w)
    rw, rh = int(ori_w * ratio), int(ori_h * ratio)
    ph, pw = (input_size - rh) / 2, (input_size - rw) / 2
    top, bottom = int(round(ph - 0.1)), int(round(ph + 0.1))
    left, right = int(round(pw - 0.1)), int(round(pw + 0.1))
    return (rh, rw), (ratio, ratio), (top, left, bottom, right)


def library_sample(img_rgb, annotations, input_size, hflip, vflip):
    """BaseDataset.__getitem__ (detection_dataloader.py:68-96) with the two random draws given: returns
    (image float32 [3,S,S], labels float32 [n,6] with column 0 left at zero)."""
    ori_h, ori_w = img_rgb.shape[:2]
    (rh, rw), (ratio_h, ratio_w), pad = library_geometry(ori_h, ori_w, input_size)
    resized = resize_linear_u8(img_rgb, (rw, rh))
    img = copy_make_border_constant(resized, pad[0], pad[2], pad[1], pad[3], (114, 114, 114))
    label = np.array(annotations, dtype=np.float32).reshape([-1, 5])
    label[:, 1] = label[:, 1] * ratio_w + pad[1]
    label[:, 2] = label[:, 2] * ratio_h + pad[0]
    label[:, 3] = label[:, 3] * ratio_w + pad[1]
    label[:, 4] = label[:, 4] * ratio_h + pad[0]
    label[:, 1:] = xyxy2xywhn(label[:, 1:], input_size, input_size)
    box = label[:, 1:]
    if hflip:
        img = np.fliplr(img)
        base = 1 if (box[0, 2] < 1 and box[0, 3] < 1) else img.shape[1]
        box[:, 0] = base - box[:, 0]
    if vflip:
        img = np.flipud(img)
        base = 1 if (box[0, 2] < 1 and box[0, 3] < 1) else img.shape[0]
        box[:, 1] = base - box[:, 1]
    img = img / 255.
    img = (img - IMAGENET_MEAN) / IMAGENET_STD
    out = np.zeros([len(label), 6], dtype=np.float32)
    out[:, 1:] = label
    return np.ascontiguousarray(img.transpose([2, 0, 1])).astype(np.float32), out


def xyxy2xywhn(xyxy, height, width):
    """detection/tools/BOX.py"""
    return np.stack([((xyxy[:, 0] + xyxy[:, 2]) / 2) / width, ((xyxy[:, 1] + xyxy[:, 3]) / 2) / height,
                     (xyxy[:, 2] - xyxy[:, 0]) / width, (xyxy[:, 3] - xyxy[:, 1]) / height], axis=1)


def collate(samples):
    """BaseDataset.collate_fn: stack images, concatenate labels with the image index in column 0."""
    imgs, labels = zip(*samples)
    for i, l in enumerate(labels):
        l[:, 0] = i
    return np.stack(imgs, 0), np.concatenate(labels, 0)


# ------------------------------------------------------------------------------------------------ demo pipeline
def demo_resize_by_max(image, labels, max_size):
    """data_gen.py:42-62"""
    h, w = image.shape[:2]
    ratio = max_size / max(h, w)
    return resize_linear_u8(image, (int(w * ratio), int(h * ratio))), labels * ratio


def demo_padding(image, label, size, fill_value=128):
    """data_gen.py:64-92 (centre, // 2 offsets)"""
    h, w, c = image.shape
    top, left = int((size - h) // 2), int((size - w) // 2)
    out = np.full((size, size, c), fill_value, dtype=image.dtype)
    out[top:top + h, left:left + w] = image
    label = label.copy()
    label[:, [1, 3]] += top
    label[:, [0, 2]] += left
    return out, label


def demo_hflip(image, label):
    """data_gen.py:94-108: xyxy -> xywh, mirror the centre, back to xyxy"""
    image = flip(image, 1)
    w = image.shape[1]
    xywh = np.stack([(label[:, 0] + label[:, 2]) / 2, (label[:, 1] + label[:, 3]) / 2, label[:, 2] - label[:, 0],
                     label[:, 3] - label[:, 1]], axis=1)
    xywh[:, 0] = w - xywh[:, 0]
    return image, np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2,
                            xywh[:, 1] + xywh[:, 3] / 2], axis=1)


def demo_vflip(image, label):
    """data_gen.py:110-124"""
    image = flip(image, 0)
    h = image.shape[0]
    xywh = np.stack([(label[:, 0] + label[:, 2]) / 2, (label[:, 1] + label[:, 3]) / 2, label[:, 2] - label[:, 0],
                     label[:, 3] - label[:, 1]], axis=1)
    xywh[:, 1] = h - xywh[:, 1]
    return image, np.stack([xywh[:, 0] - xywh[:, 2] / 2, xywh[:, 1] - xywh[:, 3] / 2, xywh[:, 0] + xywh[:, 2] / 2,
                            xywh[:, 1] + xywh[:, 3] / 2], axis=1)


def demo_mosaic(images_labels, input_size, fill_value=128):
    """data_gen.py:171-216 (Mosaic01): four images, each resized to input_size // 2 on its longer side, meet at the centre"""
    merged = np.full((input_size, input_size, 3), fill_value, dtype=np.uint8)
    cx = cy = input_size // 2
    boxes, cats = [], []
    for idx, (image, xyxy, cat) in enumerate(images_labels):
        image, xyxy = demo_resize_by_max(image, xyxy, input_size // 2)
        h, w = image.shape[:2]
        x0 = cx - w if idx in (0, 2) else cx
        y0 = cy - h if idx in (0, 1) else cy
        merged[y0:y0 + h, x0:x0 + w] = image
        xyxy = xyxy.copy()
        xyxy[:, [0, 2]] += x0
        xyxy[:, [1, 3]] += y0
        boxes.append(xyxy)
        cats.append(cat)
    return merged, np.clip(np.concatenate(boxes, 0), 0, input_size - 1), np.concatenate(cats, 0).reshape(-1)
