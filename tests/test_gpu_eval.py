"""Validation side on the GPU (scope row f-2): eval decode, NMS and mAP through the C ABI, against the oracle and the
vectors captured from the reference (tests/golden/eval_*.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def lib():
    return np.load(os.path.join(GOLD, 'eval_lib.npz'))


@pytest.fixture(scope='module')
def demo():
    return np.load(os.path.join(GOLD, 'eval_demo.npz'))


def _anchor_lists(px=True):
    from fastvision_amd.synthetic import coco_anchors_px, coco_anchors_feature
    sets = coco_anchors_px().view(3, 3, 2) if px else coco_anchors_feature()
    return [[(float(a[0]), float(a[1])) for a in s] for s in sets]


def test_decode_kernel_matches_reference(lib):
    from fastvision_amd.detect_ops import yolo_decode
    from fastvision_amd.synthetic import LEVEL_STRIDES
    for case in range(2):
        heads = [torch.from_numpy(lib[f'e1_{case}_head{l}']).to(DEV) for l in range(3)]
        got = yolo_decode(heads, _anchor_lists(), LEVEL_STRIDES, variant=0)
        np.testing.assert_allclose(got.cpu().numpy(), lib[f'e1_{case}_results'], rtol=2e-6, atol=1e-6)
        # strided views of an NHWC buffer (what the head kernels produce) decode identically
        views = []
        for h in heads:
            B, A, H, W, K = h.shape
            buf = h.permute(0, 2, 3, 1, 4).reshape(B, H, W, A * K).contiguous()
            views.append(buf.view(B, H, W, A, K).permute(0, 3, 1, 2, 4))
        got2 = yolo_decode(views, _anchor_lists(), LEVEL_STRIDES, variant=0)
        assert torch.equal(got, got2)


def test_eval_forward_with_decode_matches_reference(lib):
    """whole model in eval mode (running statistics) + decode, fp32, against the reference's output"""
    import fastvision_amd
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.synthetic import coco_anchors_px
    with fastvision_amd.compute_dtype(torch.float32):
        for case in range(2):
            S, B, seed = lib[f'e1_{case}_cfg']
            torch.manual_seed(int(seed))
            m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(),
                       num_anchors_per_level=[3, 3, 3], in_channels=3, num_classes=80, training=False)
            m.eval()
            g = torch.Generator().manual_seed(int(seed))
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                    mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
            images = torch.rand(int(B), 3, int(S), int(S), generator=g)
            np.testing.assert_array_equal(images.numpy(), lib[f'e1_{case}_images'])
            m.to(DEV)
            with torch.no_grad():
                head_out, results = m(images.to(DEV))
            for l in range(3):
                np.testing.assert_allclose(head_out[l].cpu().numpy(), lib[f'e1_{case}_head{l}'], rtol=1e-3, atol=1e-4)
            ref = lib[f'e1_{case}_results']
            np.testing.assert_allclose(results.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


def test_nms_library_matches_reference_wrapper(lib):
    from fastvision_amd.detection.tools import non_max_suppression
    for case in range(6):
        ct, it, md = lib[f'e2_{case}_cfg']
        pred = torch.from_numpy(lib[f'e2_{case}_pred']).to(DEV)
        keep = pred.clone()
        sc, cat, box = non_max_suppression(pred, ct, it, int(md))
        assert torch.equal(pred, keep)
        assert sc.shape[1:] == (1,) and cat.shape[1:] == (1,) and box.shape[1:] == (4,) and cat.dtype == torch.int64
        np.testing.assert_array_equal(sc.cpu().numpy().reshape(-1), lib[f'e2_{case}_scores'])
        np.testing.assert_array_equal(cat.cpu().numpy().reshape(-1), lib[f'e2_{case}_cats'])
        np.testing.assert_array_equal(box.cpu().numpy().reshape(-1, 4), lib[f'e2_{case}_boxes'])


def test_nms_batched_equals_per_image_and_oracle():
    from fastvision_amd.detect_ops import nms_batch, NMS_LIBRARY, NMS_DEMO_BATCH
    from oracle import detect as D
    from oracle.make_golden import synth_predictions
    preds = torch.stack([synth_predictions(torch.Generator().manual_seed(40 + i), 3000, [4, 0, 9, 2][i], 640) for i in range(4)])
    for mode, fn in ((NMS_LIBRARY, None), (NMS_DEMO_BATCH, D.nms_demo_batch)):
        res = nms_batch(preds.to(DEV), 0.25, 0.45, 50, mode)
        for b in range(4):
            det, rows = res[b]
            if fn is None:
                sc, cat, box = D.nms_library(preds[b], 0.25, 0.45, 50)
                ref = torch.cat([box, sc, cat.float()], dim=1)
            else:
                ref = fn([preds[b]], 0.25, 0.45, 50)[0]
            np.testing.assert_array_equal(det.cpu().numpy(), ref.numpy().reshape(-1, 6))
            if len(rows):
                assert (preds[b][rows.cpu(), 4] > 0.25).all()


def test_nms_every_row_a_candidate():
    """an untrained model passes every row: 25200 candidates per image at 640x640 (the per-image path of nms_batch)"""
    from fastvision_amd import detect_ops
    from oracle import detect as D
    g = torch.Generator().manual_seed(5)
    R = 6000
    pred = torch.rand(2, R, 85, generator=g)
    pred[..., 0:2] *= 640
    pred[..., 2:4] = 10 + pred[..., 2:4] * 80
    pred[..., 4] = 0.5 + 0.5 * pred[..., 4]
    old = detect_ops.MASK_BYTES_PER_CALL
    try:
        for limit in (old, 1 << 20):                       # second round forces the one-image-at-a-time path
            detect_ops.MASK_BYTES_PER_CALL = limit
            res = detect_ops.nms_batch(pred.to(DEV), 0.25, 0.45, 300, detect_ops.NMS_LIBRARY)
            for b in range(2):
                sc, cat, box = D.nms_library(pred[b], 0.25, 0.45, 300)
                np.testing.assert_array_equal(res[b][0].cpu().numpy(), torch.cat([box, sc, cat.float()], 1).numpy())
    finally:
        detect_ops.MASK_BYTES_PER_CALL = old


def test_map_matches_reference(lib):
    from fastvision_amd.metrics import CalculateMAP
    est = CalculateMAP(np.linspace(0.5, 0.95, 10))
    n_img, n_correct = lib['e3_n']
    for i in range(n_img):
        est.process_one(torch.from_numpy(lib[f'e3_{i}_pred']).to(DEV), torch.from_numpy(lib[f'e3_{i}_target']).to(DEV))
    assert len(est.correct_all_images) == n_correct
    for i, c in enumerate(est.correct_all_images):
        np.testing.assert_array_equal(c, lib[f'e3_correct{i}'])
    map_iou, map_cls, idx = est.fetch()
    np.testing.assert_allclose(map_iou, lib['e3_map_each_iou'], rtol=1e-12)
    np.testing.assert_allclose(map_cls, lib['e3_map_each_cls'], rtol=1e-12)
    assert idx == lib['e3_cls_idx'].tolist()


def test_demo_postprocess_matches_reference(demo):
    from fastvision_amd.demos.yolov3_u.inference import postProcess
    from fastvision_amd.synthetic import coco_anchors_feature
    from oracle import detect as D
    anchors = [a.view(-1, 2).to(DEV) for a in coco_anchors_feature()]
    for case in range(2):
        S, rr, pl, pt, ow, oh, ct, it = demo[f'd1_{case}_cfg']
        layers = [torch.from_numpy(demo[f'd1_{case}_layer{l}']).to(DEV) for l in range(3)]
        # the model hands out NCHW views of NHWC buffers: feed the same kind of view
        layers = [l.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2) for l in layers]
        sc, cat, box = postProcess(layers, [32, 16, 8], anchors, ct, it, rr, int(pl), int(pt), int(ow), int(oh))
        n = len(demo[f'd1_{case}_scores'])
        assert sc.shape == (n, 1) and cat.shape == (n, 1) and box.shape == (n, 4)
        np.testing.assert_allclose(sc.cpu().numpy().reshape(-1), demo[f'd1_{case}_scores'], rtol=1e-5)
        np.testing.assert_array_equal(cat.cpu().numpy().reshape(-1), demo[f'd1_{case}_cats'])
        np.testing.assert_allclose(box.cpu().numpy(), demo[f'd1_{case}_boxes'], rtol=1e-5, atol=1e-3)


def test_demo_nms_wrappers_match_reference(demo):
    from fastvision_amd.demos.yolov3_u.utils import non_max_suppression, non_max_suppression_batch
    for case in range(3):
        ct, it, md = demo[f'd2_{case}_cfg']
        pred = torch.from_numpy(demo[f'd2_{case}_pred'])
        xyxy = pred.clone()
        xyxy[:, 2:4] = xyxy[:, 0:2] + pred[:, 2:4]
        res = non_max_suppression(xyxy.to(DEV), ct, it, int(md))
        np.testing.assert_array_equal(res.cpu().numpy().reshape(-1, 6), demo[f'd2_{case}_single'])
        resb = non_max_suppression_batch([pred.to(DEV), pred.flip(0).to(DEV)], ct, it, int(md))
        assert all(not r.is_cuda for r in resb)
        np.testing.assert_array_equal(resb[0].numpy().reshape(-1, 6), demo[f'd2_{case}_batch0'])
        np.testing.assert_array_equal(resb[1].numpy().reshape(-1, 6), demo[f'd2_{case}_batch1'])


def test_fit_val_reports_loss_and_map():
    """Fit._val end to end on a tiny synthetic loader: eval forward + decode + NMS + mAP run and agree with the same
    steps done through the oracle on the device's decoded rows."""
    import fastvision_amd
    from fastvision_amd.classfication.models import darknet53
    from fastvision_amd.detection.neck import yolov3neck
    from fastvision_amd.detection.head import yolov3head
    from fastvision_amd.detection.models import yolov3
    from fastvision_amd.loss import Yolov3Loss
    from fastvision_amd.utils import Fit
    from fastvision_amd.synthetic import coco_anchors_px, synthetic_batch
    from oracle import detect as D
    with fastvision_amd.compute_dtype(torch.float32):
        torch.manual_seed(3)
        m = yolov3(backbone=darknet53, neck=yolov3neck, head=yolov3head, anchors=coco_anchors_px(),
                   num_anchors_per_level=[3, 3, 3], in_channels=3, num_classes=80, training=True).to(DEV)
        # push the objectness bias down so that only some rows pass the 0.25 threshold
        with torch.no_grad():
            for conv in m.head.heads:
                conv.bias.view(3, 85)[:, 4] -= 1.5
        crit = Yolov3Loss(m, 0.5, 0.05, 1.0, 0.5)
        images, targets = synthetic_batch(2, 64, seed=9)
        fit = Fit(m, torch.device(DEV), None, None, crit, end_epoch=1, train_loader=[(images, targets)], val_loader=[(images, targets)])
        out = fit._val()
        assert len(out['loss']) == 1 and np.isfinite(out['loss'][0])
        m.eval()
        with torch.no_grad():
            _, results = m(images.to(DEV), val=True)
        est = D.CalculateMAP(np.linspace(0.5, 0.95, 10))
        for b in range(2):
            sc, cat, box = D.nms_library(results[b].cpu(), 0.25, 0.45, 300)
            tg = targets[targets[:, 0] == b, 1:].clone()
            tg[:, 1:] = D.xywh2xyxy(tg[:, 1:]) * 64
            est.process_one(torch.cat([cat.float(), sc, box], dim=1), tg)
        if est.correct_all_images:
            ref_iou, ref_cls, ref_idx = est.fetch()
            np.testing.assert_allclose(out['map_each_iou'], ref_iou, rtol=1e-9, atol=1e-12)
            assert out['map_each_cls_idx'] == ref_idx
