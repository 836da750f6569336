"""The validation-side oracle (oracle/detect.py) against vectors captured from the reference (tests/golden/eval_*.npz,
made by oracle/make_golden.py eval_lib / eval_demo).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import detect as D
from fastvision_amd.synthetic import coco_anchors_px, coco_anchors_feature, LEVEL_STRIDES

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def lib():
    return np.load(os.path.join(GOLD, 'eval_lib.npz'))


@pytest.fixture(scope='module')
def demo():
    return np.load(os.path.join(GOLD, 'eval_demo.npz'))


def test_decode_library_matches_reference(lib):
    anchors = list(coco_anchors_px().view(3, 3, 2))
    for case in range(2):
        heads = [torch.from_numpy(lib[f'e1_{case}_head{l}']) for l in range(3)]
        got = D.decode_library(heads, LEVEL_STRIDES, anchors)
        np.testing.assert_allclose(got.numpy(), lib[f'e1_{case}_results'], rtol=1e-6, atol=1e-6)


def test_nms_library_matches_reference_wrapper(lib):
    for case in range(6):
        ct, it, md = lib[f'e2_{case}_cfg']
        sc, cat, box = D.nms_library(torch.from_numpy(lib[f'e2_{case}_pred']), ct, it, int(md))
        np.testing.assert_array_equal(sc.numpy().reshape(-1), lib[f'e2_{case}_scores'])
        np.testing.assert_array_equal(cat.numpy().reshape(-1), lib[f'e2_{case}_cats'])
        np.testing.assert_array_equal(box.numpy().reshape(-1, 4), lib[f'e2_{case}_boxes'])


def test_map_matches_reference(lib):
    est = D.CalculateMAP(np.linspace(0.5, 0.95, 10))
    n_img, n_correct = lib['e3_n']
    for i in range(n_img):
        est.process_one(torch.from_numpy(lib[f'e3_{i}_pred']), torch.from_numpy(lib[f'e3_{i}_target']))
    assert len(est.correct_all_images) == n_correct
    for i, c in enumerate(est.correct_all_images):
        np.testing.assert_array_equal(c, lib[f'e3_correct{i}'])
    map_iou, map_cls, idx = est.fetch()
    np.testing.assert_allclose(map_iou, lib['e3_map_each_iou'], rtol=1e-12)
    np.testing.assert_allclose(map_cls, lib['e3_map_each_cls'], rtol=1e-12)
    assert idx == lib['e3_cls_idx'].tolist()


def test_demo_postprocess_matches_reference(demo):
    anchors = [a.view(-1, 2) for a in coco_anchors_feature()]
    for case in range(2):
        S, rr, pl, pt, ow, oh, ct, it = demo[f'd1_{case}_cfg']
        layers = [torch.from_numpy(demo[f'd1_{case}_layer{l}']) for l in range(3)]
        rows = D.decode_demo(layers, [32, 16, 8], anchors, rr, int(pl), int(pt), int(ow), int(oh))
        res = D.nms_demo(rows, ct, it, 300)
        np.testing.assert_allclose(res[:, 4].numpy(), demo[f'd1_{case}_scores'], rtol=1e-6)
        np.testing.assert_array_equal(res[:, 5].numpy(), demo[f'd1_{case}_cats'])
        np.testing.assert_allclose(res[:, :4].numpy(), demo[f'd1_{case}_boxes'], rtol=1e-6, atol=1e-4)


def test_demo_nms_wrappers_match_reference(demo):
    for case in range(3):
        ct, it, md = demo[f'd2_{case}_cfg']
        pred = torch.from_numpy(demo[f'd2_{case}_pred'])
        xyxy = pred.clone()
        xyxy[:, 2:4] = xyxy[:, 0:2] + pred[:, 2:4]
        np.testing.assert_array_equal(D.nms_demo(xyxy, ct, it, int(md)).numpy().reshape(-1, 6), demo[f'd2_{case}_single'])
        resb = D.nms_demo_batch([pred, pred.flip(0)], ct, it, int(md))
        np.testing.assert_array_equal(resb[0].numpy().reshape(-1, 6), demo[f'd2_{case}_batch0'])
        np.testing.assert_array_equal(resb[1].numpy().reshape(-1, 6), demo[f'd2_{case}_batch1'])


def test_nms_core_properties():
    """the restated torchvision.ops.nms: kept boxes are mutually <= thr, every dropped box overlaps a better kept one"""
    g = torch.Generator().manual_seed(3)
    xy = torch.rand(400, 2, generator=g) * 100
    boxes = torch.cat([xy, xy + 5 + torch.rand(400, 2, generator=g) * 40], dim=1)
    scores = torch.rand(400, generator=g)
    keep = D.nms(boxes, scores, 0.5)
    iou = D.iou_xyxy_batch(boxes, boxes)
    assert (scores[keep][:-1] >= scores[keep][1:]).all()
    sub = iou[keep][:, keep] - torch.eye(len(keep))
    assert (sub <= 0.5 + 1e-6).all()
    dropped = sorted(set(range(400)) - set(keep.tolist()))
    for j in dropped:
        better = keep[scores[keep] >= scores[j]]
        assert (iou[j, better] > 0.5 - 1e-6).any()
