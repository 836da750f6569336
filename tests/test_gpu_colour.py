"""Colour / blur extras of the demo's training image path on the device (csrc/colour.hip; scope row f-3) against the CPU restatement
(oracle/colour.py), BIT FOR BIT: HistEqualize (RGB<->YUV + CLAHE), HueSaturationValue (RGB<->HSV + byte tables), the three 3x3 blurs,
ChannelShuffle + / 255, Jitter -- and the whole DeviceAugmenter.train_batch with every extra switched on against the same steps
composed from oracle functions.  Reference: demos/yolov3_u/data_gen.py:26-33,120-170,293-353 (OpenCV / albumentations underneath:
parity unpinned, see the oracle's header)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _img(h, w, seed):
    r = np.random.RandomState(seed)
    base = r.randint(0, 256, (h, w, 3)).astype(np.uint8)
    base[: h // 3] = (base[: h // 3] // 4 + 90).astype(np.uint8)          # a low-contrast band: CLAHE clips there
    return base


def _canvases(images, H, W):
    c = np.full((len(images), H, W, 3), 128, np.uint8)
    for i, im in enumerate(images):
        c[i, :im.shape[0], :im.shape[1]] = im
    return torch.from_numpy(c).to(DEV)


@pytest.mark.parametrize('sizes', [[(64, 72), (61, 83), (64, 83), (40, 56)], [(96, 96), (17, 130)]])
def test_clahe_and_hsv_bit_exact(sizes):
    from oracle import colour as OC
    from fastvision_amd.pipeline_ops import clahe_hsv_, hsv_tables
    images = [_img(h, w, 10 + i) for i, (h, w) in enumerate(sizes)]
    H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
    rng = np.random.RandomState(1)
    gains = [rng.uniform(-1, 1, 3) * [0.015, 0.7, 0.4] + 1 for _ in images]
    for clahe, hsv in ((True, False), (False, True), (True, True)):
        c = _canvases(images, H, W)
        flags = [clahe and i % 3 != 2 for i in range(len(images))]
        luts = [hsv_tables(g) if (hsv and i != 1) else None for i, g in enumerate(gains)]
        clahe_hsv_(c, sizes, flags, luts)
        got = c.cpu().numpy()
        for i, im in enumerate(images):
            want = im
            if flags[i]:
                want = OC.hist_equalize(want)
            if luts[i] is not None:
                assert np.array_equal(luts[i], OC.hsv_luts(gains[i]))
                want = OC.hue_saturation_value(want, luts[i])
            h, w = sizes[i]
            assert np.array_equal(got[i, :h, :w], want), (i, clahe, hsv, np.abs(got[i, :h, :w].astype(int) - want).max())
            assert np.all(got[i, h:] == 128) and np.all(got[i, :, w:] == 128)          # outside the valid region: untouched


def test_blur_shuffle_normalize_bit_exact():
    from oracle import colour as OC
    from fastvision_amd.pipeline_ops import blur_shuffle_normalize, value_table
    images = [_img(48, 48, 20 + i) for i in range(5)]
    c = _canvases(images, 48, 48)
    blur = [0, 1, 2, 3, 3]
    perms = [(0, 1, 2), (2, 0, 1), (1, 0, 2), (0, 1, 2), (2, 1, 0)]
    out = blur_shuffle_normalize(c, blur, perms, value_table(single=True)).cpu().numpy()
    fns = {0: lambda x: x, 1: OC.blur3, 2: OC.median3, 3: OC.gauss3}
    for i, im in enumerate(images):
        want = fns[blur[i]](im)[..., list(perms[i])]
        want = (np.transpose(want, (2, 0, 1)).astype(np.float32) / np.float32(255.0))
        assert np.array_equal(out[i], want), (i, np.abs(out[i] - want).max())


def test_train_batch_with_every_extra_vs_oracle():
    """DeviceAugmenter.train_batch: Jitter -> ResizeByMax -> flips -> HistEqualize -> HueSaturationValue -> Mosaic01 -> blur ->
    ChannelShuffle -> / 255 for a batch of two mosaics, against the same chain built from the oracle's functions; labels too."""
    from oracle import colour as OC, pipeline as OP
    from fastvision_amd.demos.yolov3_u.data_gen import DeviceAugmenter
    from fastvision_amd.pipeline_ops import hsv_tables
    S = 96
    rng = np.random.RandomState(7)
    groups, post = [], [{'blur': 3, 'perm': (2, 0, 1)}, {'blur': 2, 'perm': (0, 1, 2)}]
    for b in range(2):
        g = []
        for k in range(4):
            h, w = int(rng.randint(50, 120)), int(rng.randint(50, 120))
            img = _img(h, w, 100 + 4 * b + k)
            n = int(rng.randint(1, 4))
            x0, y0 = rng.randint(0, w // 2, n), rng.randint(0, h // 2, n)
            xyxy = np.stack([x0, y0, x0 + rng.randint(5, w // 2, n), y0 + rng.randint(5, h // 2, n)], 1).astype(np.float32)
            extras = {}
            if (b + k) % 2 == 0:
                extras['jitter'] = (int(h * rng.uniform(0.7, 1.3)), int(w * rng.uniform(0.7, 1.3)))
            if k % 2 == 1:
                extras['clahe'] = True
            if k >= 1:
                extras['hsv_gains'] = rng.uniform(-1, 1, 3) * [0.015, 0.7, 0.4] + 1
            g.append((img, xyxy, rng.randint(0, 5, n).astype(np.float32), bool(k & 1), bool(k & 2), extras))
        groups.append(g)
    aug = DeviceAugmenter(S, DEV)
    images, labels = aug.train_batch(groups, post=post)
    assert tuple(images.shape) == (2, 3, S, S)
    got = images.cpu().numpy()
    plan_labels = DeviceAugmenter(S, 'cpu').train_labels(groups)
    for i, l in enumerate(plan_labels):
        l[:, 0] = i
    assert torch.equal(torch.cat(plan_labels, 0), labels.cpu())
    for b, g in enumerate(groups):
        tiles = []
        for img, xyxy, cat, hf, vf, ex in g:
            lab = xyxy.copy()
            if ex.get('jitter') is not None:
                nh, nw = ex['jitter']
                lab[:, [0, 2]] *= nw / img.shape[1]
                lab[:, [1, 3]] *= nh / img.shape[0]
                img = OC.jitter(img, nh, nw)
            img, lab = OP.demo_resize_by_max(img, lab, S)
            if hf:
                img, lab = OP.demo_hflip(img, lab)
            if vf:
                img, lab = OP.demo_vflip(img, lab)
            if ex.get('clahe'):
                img = OC.hist_equalize(img)
            if ex.get('hsv_gains') is not None:
                img = OC.hue_saturation_value(img, hsv_tables(ex['hsv_gains']))
            tiles.append((img, lab, cat))
        mosaic, mlab, mcat = OP.demo_mosaic(tiles, S, fill_value=128)
        fn = {0: lambda x: x, 1: OC.blur3, 2: OC.median3, 3: OC.gauss3}[post[b]['blur']]
        want = fn(mosaic)[..., list(post[b]['perm'])]
        want = np.transpose(want, (2, 0, 1)).astype(np.float32) / np.float32(255.0)
        assert np.array_equal(got[b], want), (b, np.abs(got[b] - want).max())
