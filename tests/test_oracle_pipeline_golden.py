"""The input-side oracle (oracle/pipeline.py) against vectors captured from the reference's own dataset classes
(tests/golden/pipeline*.npz, made by oracle/make_golden.py pipeline / pipeline_demo).  CPU only."""
import os

import numpy as np
import pytest

from oracle import pipeline as P

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def lib():
    return np.load(os.path.join(GOLD, 'pipeline.npz'))


@pytest.fixture(scope='module')
def demo():
    return np.load(os.path.join(GOLD, 'pipeline_demo.npz'))


def _ann(z, i):
    return [tuple(r) for r in z[f'p_{i}_ann'].tolist()]


def test_library_sample_matches_reference(lib):
    for i, (h, w, S, n) in enumerate(lib['p_cases']):
        for hf in (0, 1):
            for vf in (0, 1):
                img, lab = P.library_sample(lib[f'p_{i}_rgb'], _ann(lib, i), int(S), hf, vf)
                np.testing.assert_array_equal(lab, lib[f'l1_{i}_{hf}{vf}_lab'])
                if S <= 128:
                    np.testing.assert_array_equal(img, lib[f'l1_{i}_{hf}{vf}_img'])
                else:
                    np.testing.assert_array_equal(img[:, ::7, ::5], lib[f'l1_{i}_{hf}{vf}_img_sub'])
                    sums = np.array([img.astype(np.float64).sum(), (img.astype(np.float64) ** 2).sum()])
                    np.testing.assert_allclose(sums, lib[f'l1_{i}_{hf}{vf}_img_sums'], rtol=1e-12)


def test_library_collate_matches_reference(lib):
    idx = lib['l2_index']
    batch = [P.library_sample(lib[f'p_{i}_rgb'], _ann(lib, i), 128, 0, 0) for i in idx]
    imgs, labs = P.collate(batch)
    np.testing.assert_array_equal(labs, lib['l2_labels'])
    np.testing.assert_allclose(imgs.astype(np.float64).sum(axis=(1, 2, 3)), lib['l2_image_sums'], rtol=1e-12)


def test_padding_offsets_match_reference(lib):
    for rh, rw, top, left, bottom, right in lib['l3_padding']:
        ph, pw = (40 - rh) / 2, (40 - rw) / 2
        assert (int(round(ph - 0.1)), int(round(pw - 0.1)), int(round(ph + 0.1)), int(round(pw + 0.1))) == (top, left, bottom, right)


def test_demo_steps_match_reference(demo):
    imgs = []
    for i in range(4):
        rgb, ann = demo[f'q_{i}_rgb'], demo[f'q_{i}_ann']
        xyxy, cat = ann[:, 1:].copy(), ann[:, 0].copy()
        imgs.append((rgb, xyxy, cat))
        im, lb = P.demo_resize_by_max(rgb, xyxy.copy(), 96)
        np.testing.assert_array_equal(im, demo[f'd1_{i}_resized'])
        np.testing.assert_array_equal(lb, demo[f'd1_{i}_resized_lab'])
        imh, lbh = P.demo_hflip(im, lb)
        imv, lbv = P.demo_vflip(imh, lbh)
        np.testing.assert_array_equal(imv, demo[f'd1_{i}_hv'])
        np.testing.assert_array_equal(lbv, demo[f'd1_{i}_hv_lab'])
        pim, plb = P.demo_padding(imv, lbv, 96)
        np.testing.assert_array_equal(pim, demo[f'd1_{i}_padded'])
        np.testing.assert_array_equal(plb, demo[f'd1_{i}_padded_lab'])
    for case, S in enumerate((96, 160)):
        pre = [P.demo_resize_by_max(a, b.copy(), S) + (c,) for a, b, c in imgs]
        m, xyxy, cat = P.demo_mosaic(pre, S)
        np.testing.assert_array_equal(m, demo[f'd2_{case}_image'])
        np.testing.assert_array_equal(xyxy, demo[f'd2_{case}_xyxy'])
        np.testing.assert_array_equal(cat, demo[f'd2_{case}_cat'])


def test_resize_properties():
    """the restated resampler: constants stay constant, identity size copies, exact 2x decimation averages 2x2 blocks,
    a horizontal ramp stays monotone and inside the source range"""
    const = np.full((37, 53, 3), 201, np.uint8)
    assert (P.resize_linear_u8(const, (91, 17)) == 201).all()
    g = np.random.default_rng(0)
    img = g.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    assert (P.resize_linear_u8(img, (60, 40)) == img).all()
    half = P.resize_linear_u8(img, (30, 20))
    ref = (img.astype(np.int64).reshape(20, 2, 30, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    assert (half == ref).all()
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8)[None, :, None], (8, 1, 3))
    up = P.resize_linear_u8(ramp, (333, 8)).astype(np.int64)
    assert (np.diff(up[0, :, 0]) >= 0).all() and up.min() >= 0 and up.max() <= 198
