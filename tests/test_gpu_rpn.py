"""RPN proposal layer on the HIP kernels (scope row f-4) against the reference's vectors (tests/golden/rpn_proposals.npz) and
the CPU restatement, through the C ABI (fastvision_amd.rpn_ops)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rpn_proposals.npz'))
CASES = sorted({k.split('_')[0] for k in GOLD.files})


@pytest.mark.parametrize('c', CASES)
def test_rows_and_proposals_match_reference(c):
    from fastvision_amd.rpn_ops import filter_proposals, rpn_proposal_rows
    B, H, W, A, pre, post = (int(v) for v in GOLD[f'{c}_shape'])
    cls = torch.from_numpy(GOLD[f'{c}_cls']).to(DEV)
    d = torch.from_numpy(GOLD[f'{c}_d']).to(DEV)
    rows = rpn_proposal_rows(cls, d, GOLD[f'{c}_base_wh']).cpu()
    assert tuple(rows.shape) == (B, H * W * A, 6)
    un = torch.from_numpy(GOLD[f'{c}_xyxy_unclamped']).view(B, -1, 4)
    lim = torch.tensor([W - 1, H - 1, W - 1, H - 1], dtype=torch.float32)
    want_box = torch.minimum(un.clamp(min=0), lim)
    # exp / softmax run in the device's libm: boxes to 1e-5 relative, scores to 1e-6 absolute; the clamp bounds exactly
    assert torch.allclose(rows[..., :4], want_box, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rows[..., 4], torch.from_numpy(GOLD[f'{c}_score']).view(B, -1), rtol=0, atol=1e-6)
    assert torch.all(rows[..., 5] == 1)
    props = filter_proposals(cls, d, GOLD[f'{c}_base_wh'], pre, post, 0.7)
    assert len(props) == B
    for b, p in enumerate(props):
        want = torch.from_numpy(GOLD[f'{c}_prop{b}'])
        assert p.shape == want.shape, (p.shape, want.shape)
        assert torch.allclose(p.cpu(), want, rtol=1e-5, atol=1e-4)


def test_reference_size_properties():
    """VGG16 stride-16 map of an 800x608 image (38 x 50 cells, 9 anchors = 17100 rows per image), 4 images, the reference's
    defaults (2000 / 2000 / 0.7): sizes, ordering and the NMS invariant (no kept pair overlaps by more than the threshold)."""
    from fastvision_amd.rpn_ops import filter_proposals
    from oracle.detect import iou_xyxy_batch
    g = torch.Generator().manual_seed(7)
    B, H, W, A = 4, 38, 50, 9
    cls = (torch.randn(B, H, W, A, 2, generator=g) * 2).to(DEV)
    d = (torch.randn(B, H, W, A, 4, generator=g) * 0.5).to(DEV)
    base = torch.tensor([[11.3, 5.7], [22.6, 11.3], [45.3, 22.6], [8, 8], [16, 16], [32, 32], [5.7, 11.3], [11.3, 22.6], [22.6, 45.3]])
    props = filter_proposals(cls, d, base, 2000, 2000, 0.7)
    assert len(props) == B
    for p in props:
        n = p.size(0)
        assert 0 < n <= 2000 and p.size(1) == 4
        assert torch.all(p[:, 2] >= 0) and torch.all(p[:, 3] >= 0)
        xyxy = torch.stack([p[:, 0] - p[:, 2] / 2, p[:, 1] - p[:, 3] / 2, p[:, 0] + p[:, 2] / 2, p[:, 1] + p[:, 3] / 2], 1).cpu()
        assert xyxy.min() >= -1e-4 and xyxy[:, [0, 2]].max() <= W - 1 + 1e-4 and xyxy[:, [1, 3]].max() <= H - 1 + 1e-4
        iou = torch.as_tensor(iou_xyxy_batch(xyxy[:400], xyxy[:400]))
        iou = torch.nan_to_num(iou, nan=0.0)
        iou.fill_diagonal_(0)
        assert iou.max() <= 0.7 + 1e-5


# ------------------------------------------------------------------------------------------------ matcher (index work: bit-exact)
from tests.test_oracle_rpn_golden import MATCH, MCASES, check_labels_against_reference  # noqa: E402


@pytest.mark.parametrize('c', MCASES)
def test_matcher_labels_equal_reference(c):
    from fastvision_amd.rpn_ops import rpn_match
    from oracle import rpn as R
    B, H, W, A, T = (int(v) for v in MATCH[f'{c}_shape'])
    anchors = R.make_anchors_xywh(MATCH[f'{c}_base_wh'], H, W)
    targets = torch.from_numpy(MATCH[f'{c}_targets'])
    labels = rpn_match(anchors.to(DEV), targets.to(DEV), B, H, W).cpu()
    assert labels.dtype == torch.int64 and tuple(labels.shape) == (B, H * W * A)
    check_labels_against_reference(c, labels)
    assert torch.equal(labels, R.rpn_match(anchors, targets, B, H, W))


def test_matcher_full_size_equals_oracle_and_edge_cases():
    """4 images of a 38 x 50 map with 9 anchors (17100 per image), 57 boxes incl. duplicates and boxes that overlap nothing."""
    from fastvision_amd.rpn_ops import rpn_match, rpn_sample
    from oracle import rpn as R
    g = torch.Generator().manual_seed(11)
    B, H, W = 4, 38, 50
    base = torch.tensor([[11.3, 5.7], [22.6, 11.3], [45.3, 22.6], [8, 8], [16, 16], [32, 32], [5.7, 11.3], [11.3, 22.6], [22.6, 45.3]])
    anchors = R.make_anchors_xywh(base, H, W)
    T = 57
    tb = torch.sort(torch.randint(0, B, (T,), generator=g))[0].float()
    wh = torch.exp(np.log(0.02) + (np.log(0.95) - np.log(0.02)) * torch.rand(T, 2, generator=g))
    xy = wh / 2 + (1 - wh) * torch.rand(T, 2, generator=g)
    targets = torch.cat([tb[:, None], torch.zeros(T, 1), xy, wh], 1)
    targets[5] = targets[4]                                   # a duplicated box: the later one claims the shared best anchor
    targets[9, 2:] = torch.tensor([0.5, 0.5, 1e-4, 1e-4])    # overlaps (almost) nothing: still claims an anchor
    want = R.rpn_match(anchors, targets, B, H, W)
    got = rpn_match(anchors.to(DEV), targets.to(DEV), B, H, W)
    assert torch.equal(got.cpu(), want)
    # an image without boxes (the reference raises there): ignored everywhere; and no boxes at all
    only0 = targets[targets[:, 0] == 0]
    lab = rpn_match(anchors.to(DEV), only0.to(DEV), 2, H, W).cpu()
    assert torch.equal(lab[0], want[0]) and torch.all(lab[1] == -2)
    assert torch.all(rpn_match(anchors.to(DEV), torch.zeros(0, 6, device=DEV), 1, H, W) == -2)
    pos, neg = rpn_sample(got[0], 128, 128)
    assert pos.numel() == min(int((want[0] >= 0).sum()), 128) and pos.numel() + neg.numel() == 256
    assert torch.all(got[0][pos] >= 0) and torch.all(got[0][neg] == -1)


# ------------------------------------------------------------------------------------------------ Fast head samples
from tests.test_oracle_rpn_golden import FCASES, check_fast_samples, fast_case  # noqa: E402


@pytest.mark.parametrize('c', FCASES)
def test_fast_samples_equal_reference(c):
    from fastvision_amd.rpn_ops import fast_match, fast_select_samples
    from oracle import rpn as R
    proposals, targets, B = fast_case(c)
    for b, p in enumerate(proposals):                          # index work: the labels themselves, bit-exact with the restatement
        want = R.fast_match(p, targets[targets[:, 0] == b][:, 2:])
        assert torch.equal(fast_match(p.to(DEV), targets.to(DEV), b).cpu(), want)
    perms = [(torch.arange(p.size(0), device=DEV), torch.arange(p.size(0), device=DEV)) for p in proposals]
    pos, neg = fast_select_samples([p.to(DEV) for p in proposals], targets.to(DEV), 0.5, 0.5, 10 ** 7, 10 ** 7, perms=perms)
    check_fast_samples(c, pos.cpu().numpy(), neg.cpu().numpy())
    # the reference's sample sizes (16 + 48 per image): 64 rows per image in total when there are enough candidates
    pos2, neg2 = fast_select_samples([p.to(DEV) for p in proposals], targets.to(DEV))
    for b in range(B):
        nb = int((pos2[:, 0] == b).sum()) + int((neg2[:, 0] == b).sum())
        assert nb <= 64 and int((pos2[:, 0] == b).sum()) <= 16
    assert fast_match(torch.zeros(0, 4, device=DEV), targets.to(DEV), 0).numel() == 0


# ------------------------------------------------------------------------------------------------ the whole RPN module, one training step
def test_rpn_module_training_step_vs_reference():
    """fastvision_amd.demos.faster_rcnn.models.RPN (conv3x3 + ReLU, the two 1x1 heads, proposals, matcher, sampler, focal and
    smooth-L1 losses, backward through the heads and the conv) against one forward + backward of the reference's RPN class on the
    same weights, inputs and randperm draws (tests/golden/rpn_step.npz), fp32."""
    import fastvision_amd
    from fastvision_amd.demos.faster_rcnn.models import RPN
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rpn_step.npz'))
    B, C, H, W, T = (int(v) for v in G['shape'])
    rpn = RPN(training=True, base_anchors=torch.from_numpy(G['base_anchors_px']), backbone_stride=16, in_channels=C,
              rpn_positives_per_image=16, rpn_negatives_per_image=48).to(DEV)
    rpn.load_state_dict({k[2:]: torch.from_numpy(G[k]) for k in G.files if k.startswith('w_')})
    feature = torch.from_numpy(G['feature']).to(DEV).requires_grad_(True)
    targets = torch.from_numpy(G['targets']).to(DEV)
    perms = [(torch.from_numpy(G[f'perm{2 * b}']).to(DEV), torch.from_numpy(G[f'perm{2 * b + 1}']).to(DEV)) for b in range(B)]
    with fastvision_amd.compute_dtype(torch.float32):
        proposals, loss_cls, loss_box = rpn(feature, targets, perms=perms)
        (loss_cls + loss_box).backward()
    np.testing.assert_allclose(loss_cls.item(), G['loss_cls'], rtol=1e-4)
    np.testing.assert_allclose(loss_box.item(), G['loss_box'], rtol=1e-4)
    for b, p in enumerate(proposals):
        assert p.shape == G[f'prop{b}'].shape
        assert np.allclose(p.detach().cpu().numpy(), G[f'prop{b}'], rtol=1e-4, atol=1e-3)

    def close(got, want, what):
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-12))
        assert err < 1e-3, (what, err)
    close(feature.grad.cpu().numpy(), G['grad_feature'], 'feature')
    for k, p in rpn.named_parameters():
        close(p.grad.cpu().numpy(), G['g_' + k], k)
