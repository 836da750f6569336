"""CPU restatement of the colour / blur extras of the demo's training image path (scope row f-3, remainder):
demos/yolov3_u/data_gen.py:120-150 (HueSaturationValue, HistEqualize), :28-33 (the albumentations OneOf[Blur, MedianBlur,
GaussianBlur] at 3x3 and ChannelShuffle) and Jitter (:152-170, a second cv2.resize -- oracle/pipeline.resize_linear_u8).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED.  Every function here restates an algorithm of OpenCV 4.5.x (opencv-python 4.5.5.62 and albumentations are pinned
by the reference's requirements; both are absent from this image and from /root/reference, and the reference holds no fixture of
their output).  What is restated, per function, from the published sources (modules/imgproc/src/{color_yuv.simd.hpp,
color_hsv.simd.hpp, clahe.cpp, box_filter.simd.hpp, median_blur.simd.hpp, smooth.simd.hpp}):
  * RGB <-> YUV, 8-bit: 14-bit fixed point, Y = (R*4899 + G*9617 + B*1868 + 2^13) >> 14, U = ((B - Y)*8061 + (128 << 14) + 2^13) >> 14,
    V = ((R - Y)*14369 + ...) >> 14; back: R = Y + ((V-128)*18678 + 2^13 >> 14), G = Y + (((U-128)*-6472 + (V-128)*-9519 + 2^13) >> 14),
    B = Y + (((U-128)*33292 + 2^13) >> 14), saturated to a byte;
  * CLAHE (clip limit 2.0, 8 x 8 tiles): unless both sides divide by 8 the image is extended by reflection (BORDER_REFLECT_101)
    by 8 - size % 8 on each axis (a full 8 on an axis that does divide: clahe.cpp's own rule), per-tile histogram
    clipped at max(int(2.0 * tile_area / 256), 1), the excess redistributed (equal share, then one count every max(256 / rest, 1)
    bins), LUT = saturate(round(cumulative * 255 / tile_area)); pixels interpolate the four neighbouring tiles' LUTs bilinearly in
    float32 (tile coordinate x / tile_w - 0.5), rounded half to even;
  * RGB -> HSV, 8-bit, H in [0, 180): the integer division tables ((255 << 12) / v, (180 << 12) / (6 * diff)), 12-bit rounding;
    HSV -> RGB, 8-bit: through float32 (h * 6/180, sector tables), * 255 rounded half to even;
  * blur 3x3: box sum * float32(1/9) rounded half to even, borders BORDER_REFLECT_101; median 3x3: per channel, borders replicated;
    Gaussian 3x3, sigma 0: kernel (1, 2, 1)/4 separable in 8.8 fixed point = (sum of (1,2,1)x(1,2,1) weights + 8) >> 4, REFLECT_101.
The HIP kernels (csrc/colour.hip) are checked bit for bit against these functions; properties that hold for ANY correct
implementation (constant images are fixed points of every transform except the LUT stage, identity LUTs give back the HSV round trip,
blurs preserve constants, ...) are tested on top.
"""
import numpy as np

from .pipeline import resize_linear_u8


def _descale(x, n=14):
    return (x + (1 << (n - 1))) >> n


def rgb2yuv_u8(rgb):
    p = rgb.astype(np.int64)
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    y = _descale(r * 4899 + g * 9617 + b * 1868)
    u = _descale((b - y) * 8061 + (128 << 14))
    v = _descale((r - y) * 14369 + (128 << 14))
    return np.clip(np.stack([y, u, v], -1), 0, 255).astype(np.uint8)


def yuv2rgb_u8(yuv):
    p = yuv.astype(np.int64)
    y, u, v = p[..., 0], p[..., 1] - 128, p[..., 2] - 128
    r = y + _descale(v * 18678)
    g = y + _descale(u * -6472 + v * -9519)
    b = y + _descale(u * 33292)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def _reflect101(idx, n):
    idx = np.abs(idx)
    return np.where(idx >= n, 2 * (n - 1) - idx, idx)


def clahe_u8(img, clip_limit=2.0, tiles=8):
    """cv2.createCLAHE(clipLimit, (tiles, tiles)).apply(img) for a uint8 [H, W] plane"""
    h, w = img.shape
    if h % tiles == 0 and w % tiles == 0:
        ph = pw = 0
    else:                                   # clahe.cpp pads BOTH axes by tiles - (size % tiles): a full 8 on an axis that divides
        ph, pw = tiles - h % tiles, tiles - w % tiles
    src = img[_reflect101(np.arange(h + ph), h)][:, _reflect101(np.arange(w + pw), w)] if (ph or pw) else img
    th, tw = src.shape[0] // tiles, src.shape[1] // tiles
    area = th * tw
    clip = max(int(clip_limit * area / 256), 1)
    scale = np.float32(255.0) / np.float32(area)
    luts = np.zeros((tiles, tiles, 256), dtype=np.uint8)
    for ty in range(tiles):
        for tx in range(tiles):
            hist = np.bincount(src[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].reshape(-1), minlength=256).astype(np.int64)
            excess = int(np.maximum(hist - clip, 0).sum())
            hist = np.minimum(hist, clip)
            batch, rest = excess // 256, excess % 256
            hist += batch
            if rest:
                step = max(256 // rest, 1)
                i = 0
                while i < 256 and rest > 0:
                    hist[i] += 1
                    i += step
                    rest -= 1
            cum = np.cumsum(hist).astype(np.float32) * scale
            luts[ty, tx] = np.clip(np.rint(cum), 0, 255).astype(np.uint8)
    inv_tw, inv_th = np.float32(1.0) / np.float32(tw), np.float32(1.0) / np.float32(th)
    xf = np.arange(w, dtype=np.float32) * inv_tw - np.float32(0.5)
    yf = np.arange(h, dtype=np.float32) * inv_th - np.float32(0.5)
    tx1, ty1 = np.floor(xf).astype(np.int64), np.floor(yf).astype(np.int64)
    xa, ya = (xf - tx1.astype(np.float32)).astype(np.float32), (yf - ty1.astype(np.float32)).astype(np.float32)
    xa1, ya1 = np.float32(1.0) - xa, np.float32(1.0) - ya
    tx2, ty2 = np.minimum(tx1 + 1, tiles - 1), np.minimum(ty1 + 1, tiles - 1)
    tx1, ty1 = np.maximum(tx1, 0), np.maximum(ty1, 0)
    v = img.astype(np.int64)
    l11 = luts[ty1[:, None], tx1[None, :], v].astype(np.float32)
    l12 = luts[ty1[:, None], tx2[None, :], v].astype(np.float32)
    l21 = luts[ty2[:, None], tx1[None, :], v].astype(np.float32)
    l22 = luts[ty2[:, None], tx2[None, :], v].astype(np.float32)
    res = (l11 * xa1[None, :] + l12 * xa[None, :]) * ya1[:, None] + (l21 * xa1[None, :] + l22 * xa[None, :]) * ya[:, None]
    return np.clip(np.rint(res.astype(np.float32)), 0, 255).astype(np.uint8)


def hist_equalize(rgb):
    """HistEqualize(image, adaptive=True), data_gen.py:137-146"""
    yuv = rgb2yuv_u8(rgb)
    yuv[..., 0] = clahe_u8(yuv[..., 0])
    return yuv2rgb_u8(yuv)


_SDIV = np.array([0] + [int(round((255 << 12) / float(i))) for i in range(1, 256)], dtype=np.int64)
_HDIV180 = np.array([0] + [int(round((180 << 12) / (6.0 * i))) for i in range(1, 256)], dtype=np.int64)


def rgb2hsv_u8(rgb):
    p = rgb.astype(np.int64)
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    v = np.maximum(np.maximum(r, g), b)
    diff = v - np.minimum(np.minimum(r, g), b)
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV180[diff] + (1 << 11)) >> 12
    h = np.where(h < 0, h + 180, h)
    return np.clip(np.stack([h, s, v], -1), 0, 255).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def hsv2rgb_u8(hsv):
    f = np.float32
    h = hsv[..., 0].astype(np.float32) * f(6.0 / 180.0)
    s = hsv[..., 1].astype(np.float32) * f(1.0 / 255.0)
    v = hsv[..., 2].astype(np.float32) * f(1.0 / 255.0)
    sector = np.floor(h).astype(np.int64)
    fr = (h - sector.astype(np.float32)).astype(np.float32)
    over = sector >= 6                                           # cannot happen for H < 180; OpenCV's guard restated
    sector, fr = np.where(over, 0, sector), np.where(over, np.float32(0), fr).astype(np.float32)
    tab = np.stack([v, (v * (f(1.0) - s)).astype(np.float32), (v * (f(1.0) - (s * fr).astype(np.float32))).astype(np.float32),
                    (v * (f(1.0) - (s * (f(1.0) - fr)).astype(np.float32))).astype(np.float32)], -1)
    idx = _SECTOR[sector]                                        # [..., 3] = (b, g, r) table slots
    b = np.take_along_axis(tab, idx[..., 0:1], -1)[..., 0]
    g = np.take_along_axis(tab, idx[..., 1:2], -1)[..., 0]
    r = np.take_along_axis(tab, idx[..., 2:3], -1)[..., 0]
    grey = hsv[..., 1] == 0
    r, g, b = np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)
    out = np.stack([r, g, b], -1).astype(np.float32) * f(255.0)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def hsv_luts(gains):
    """the three byte tables of HueSaturationValue (data_gen.py:125-131) for r = gains (float64 [3])"""
    x = np.arange(0, 256, dtype=np.float64)
    return np.stack([((x * gains[0]) % 180).astype(np.uint8), np.clip(x * gains[1], 0, 255).astype(np.uint8),
                     np.clip(x * gains[2], 0, 255).astype(np.uint8)])


def hue_saturation_value(rgb, luts):
    hsv = rgb2hsv_u8(rgb)
    out = np.stack([luts[0][hsv[..., 0]], luts[1][hsv[..., 1]], luts[2][hsv[..., 2]]], -1)
    return hsv2rgb_u8(out)


def _neigh(img, border):
    h, w = img.shape[:2]
    if border == 'reflect101':
        ys, xs = _reflect101(np.arange(-1, h + 1), h), _reflect101(np.arange(-1, w + 1), w)
    else:
        ys, xs = np.clip(np.arange(-1, h + 1), 0, h - 1), np.clip(np.arange(-1, w + 1), 0, w - 1)
    p = img[ys][:, xs].astype(np.int64)
    return [p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)]


def blur3(img):
    """cv2.blur(img, (3, 3))"""
    s = sum(_neigh(img, 'reflect101')).astype(np.float32) * np.float32(1.0 / 9.0)
    return np.clip(np.rint(s), 0, 255).astype(np.uint8)


def median3(img):
    """cv2.medianBlur(img, 3)"""
    return np.sort(np.stack(_neigh(img, 'replicate'), 0), axis=0)[4].astype(np.uint8)


def gauss3(img):
    """cv2.GaussianBlur(img, (3, 3), 0)"""
    n = _neigh(img, 'reflect101')
    wts = [1, 2, 1, 2, 4, 2, 1, 2, 1]
    return ((sum(w * p for w, p in zip(wts, n)) + 8) >> 4).astype(np.uint8)


def jitter(image, new_h, new_w):
    """Jitter's image half (data_gen.py:152-170): cv2.resize to the drawn size"""
    return resize_linear_u8(image, (new_w, new_h))
