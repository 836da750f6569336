"""Properties of the CPU restatement of the colour / blur extras (oracle/colour.py; parity UNPINNED -- OpenCV and albumentations are
absent here and the reference holds no fixture of their output).  These hold for any correct implementation of the published
algorithms, and anchor the oracle that the device kernels are compared with bit for bit (tests/test_gpu_colour.py)."""
import numpy as np

from oracle import colour as C


def _img(h, w, seed=0):
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)


def test_yuv_roundtrip_and_known_values():
    grey = np.repeat(np.arange(256, dtype=np.uint8).reshape(16, 16, 1), 3, axis=2)
    yuv = C.rgb2yuv_u8(grey)
    assert np.array_equal(yuv[..., 0], grey[..., 0]) and np.all(yuv[..., 1] == 128) and np.all(yuv[..., 2] == 128)   # greys: Y = v, U = V = 128
    assert np.array_equal(C.yuv2rgb_u8(yuv), grey)
    rgb = _img(32, 40)
    yuv = C.rgb2yuv_u8(rgb)
    back = C.yuv2rgb_u8(yuv).astype(int)
    inside = (yuv[..., 1] > 0) & (yuv[..., 1] < 255) & (yuv[..., 2] > 0) & (yuv[..., 2] < 255)      # 8-bit V saturates on strong reds
    assert inside.mean() > 0.8 and np.abs(back - rgb)[inside].max() <= 3     # elsewhere the round trip loses a few counts, never more
    assert tuple(C.rgb2yuv_u8(np.array([[[255, 0, 0]]], np.uint8))[0, 0]) == (76, 91, 255)      # pure red: 0.299*255 = 76, 128 - 0.492*76 = 90.6, 128 + 0.877*179 > 255


def test_hsv_known_values_and_roundtrip():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128]]], np.uint8)
    hsv = C.rgb2hsv_u8(px)[0]
    assert hsv.tolist() == [[0, 255, 255], [60, 255, 255], [120, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128]]
    assert np.array_equal(C.hsv2rgb_u8(C.rgb2hsv_u8(px)), px)
    rgb = _img(24, 24, 1)
    back = C.hsv2rgb_u8(C.rgb2hsv_u8(rgb)).astype(int)
    assert np.abs(back - rgb).max() <= 6 and C.rgb2hsv_u8(rgb)[..., 0].max() < 180
    ident = C.hsv_luts(np.ones(3))
    assert np.array_equal(ident[1], np.arange(256)) and np.array_equal(ident[0][:180], np.arange(180))
    assert np.array_equal(C.hue_saturation_value(rgb, ident), C.hsv2rgb_u8(C.rgb2hsv_u8(rgb)))


def test_clahe_properties():
    flat = np.full((64, 72), 97, np.uint8)
    out = C.clahe_u8(flat)
    assert np.all(out == out[0, 0])                             # a constant plane stays constant
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 4, (64, 1))
    eq = C.clahe_u8(ramp)
    assert eq.shape == ramp.shape and eq.dtype == np.uint8
    assert np.all(np.diff(eq[32].astype(int)) >= -2)            # monotone input stays (nearly) monotone inside a row
    odd = np.random.RandomState(2).randint(0, 256, (61, 83)).astype(np.uint8)     # sizes that need the reflected extension
    assert C.clahe_u8(odd).shape == odd.shape
    half = np.random.RandomState(3).randint(0, 256, (64, 83)).astype(np.uint8)    # one axis divides, the other does not
    assert C.clahe_u8(half).shape == half.shape
    rgb = _img(40, 56, 4)
    he = C.hist_equalize(rgb)
    assert he.shape == rgb.shape and not np.array_equal(he, rgb)


def test_blurs_preserve_constants_and_match_direct_sums():
    flat = np.full((9, 11, 3), 200, np.uint8)
    for f in (C.blur3, C.median3, C.gauss3):
        assert np.array_equal(f(flat), flat)
    img = _img(7, 9, 5)
    p = img.astype(int)
    y, x = 3, 4
    win = p[y - 1:y + 2, x - 1:x + 2]
    assert np.array_equal(C.blur3(img)[y, x], np.rint(win.sum((0, 1)) / 9.0).astype(int))
    assert np.array_equal(C.median3(img)[y, x], np.sort(win.reshape(9, 3), axis=0)[4])
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]])
    assert np.array_equal(C.gauss3(img)[y, x], ((win * k[:, :, None]).sum((0, 1)) + 8) >> 4)
    # corner pixel: reflect-101 takes rows/cols (1, 0, 1); the median replicates (0, 0, 1)
    w101 = p[[1, 0, 1]][:, [1, 0, 1]]
    assert np.array_equal(C.gauss3(img)[0, 0], ((w101 * k[:, :, None]).sum((0, 1)) + 8) >> 4)
    wrep = p[[0, 0, 1]][:, [0, 0, 1]]
    assert np.array_equal(C.median3(img)[0, 0], np.sort(wrep.reshape(9, 3), axis=0)[4])
