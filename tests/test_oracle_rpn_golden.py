"""RPN proposal layer (scope row f-4): the CPU restatement (oracle/rpn.py) against vectors captured from the reference's own
RPN class (tests/golden/rpn_proposals.npz, oracle/make_golden.py rpn)."""
import os

import numpy as np
import torch

from oracle import rpn as R

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rpn_proposals.npz'))
CASES = sorted({k.split('_')[0] for k in GOLD.files})


def test_anchor_grid_and_rows_match_reference():
    for c in CASES:
        B, H, W, A, pre, post = GOLD[f'{c}_shape']
        anchors = R.make_anchors_xywh(GOLD[f'{c}_base_wh'], H, W)
        assert np.array_equal(anchors.numpy(), GOLD[f'{c}_anchors'])
        rows = R.proposal_rows(torch.from_numpy(GOLD[f'{c}_cls']), torch.from_numpy(GOLD[f'{c}_d']), anchors, H, W)
        assert np.array_equal(rows[..., 0].view(B, H, W, A).numpy(), GOLD[f'{c}_score'])
        un = torch.from_numpy(GOLD[f'{c}_xyxy_unclamped']).view(B, -1, 4)
        lim = torch.tensor([W - 1, H - 1, W - 1, H - 1], dtype=torch.float32)
        assert torch.equal(rows[..., 1:], torch.minimum(un.clamp(min=0), lim))


def test_filter_proposals_matches_reference():
    for c in CASES:
        B, H, W, A, pre, post = GOLD[f'{c}_shape']
        props = R.filter_proposals(torch.from_numpy(GOLD[f'{c}_cls']), torch.from_numpy(GOLD[f'{c}_d']), GOLD[f'{c}_base_wh'], int(pre), int(post), 0.7)
        assert len(props) == B
        for b, p in enumerate(props):
            want = GOLD[f'{c}_prop{b}']
            assert p.shape == want.shape and want.shape[0] <= post
            assert np.array_equal(p.numpy(), want)


# ------------------------------------------------------------------------------------------------ matcher
MATCH = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rpn_match.npz'))
MCASES = sorted({k.split('_')[0] for k in MATCH.files if k.startswith('m')})


def check_labels_against_reference(c, labels):
    """labels [B, Na] int64 vs what the reference's computet_loss revealed (tests/golden/rpn_match.npz): per image the exact
    sets of negative and positive anchors, and for every positive anchor the regression target of the box it was matched to."""
    B, H, W, A, T = (int(v) for v in MATCH[f'{c}_shape'])
    anchors = R.make_anchors_xywh(MATCH[f'{c}_base_wh'], H, W).view(-1, 4)
    targets = torch.from_numpy(MATCH[f'{c}_targets'])
    scale = torch.tensor([W, H, W, H], dtype=torch.float32)
    pos_rows = iter(range(len(MATCH[f'{c}_pos_anchor'])))
    for b in range(B):
        sel = MATCH[f'{c}_image'] == b
        ref_pos = MATCH[f'{c}_anchor'][sel & (MATCH[f'{c}_is_pos'] == 1)]
        ref_neg = MATCH[f'{c}_anchor'][sel & (MATCH[f'{c}_is_pos'] == 0)]
        lab = labels[b]
        assert np.array_equal(torch.nonzero(lab >= 0).flatten().numpy(), ref_pos)           # ascending anchor index, like the reference
        assert np.array_equal(torch.nonzero(lab == -1).flatten().numpy(), ref_neg)
        assert int((lab == -2).sum()) == lab.numel() - len(ref_pos) - len(ref_neg)
        boxes = targets[targets[:, 0] == b][:, 2:] * scale
        for a in ref_pos:
            row = next(pos_rows)
            assert MATCH[f'{c}_pos_anchor'][row] == a
            want = MATCH[f'{c}_pos_dxdydwdh'][row]
            got = R.xywh2dxdydwdh(boxes[lab[a]][None], anchors[a][None])[0].numpy()
            # identifies the matched box; log() differs in the last bit between host CPUs (vectorised libm), so not array_equal
            assert np.allclose(got, want, rtol=2e-6, atol=1e-6), (c, b, a, int(lab[a]))


def test_matcher_labels_match_reference():
    for c in MCASES:
        B, H, W, A, T = (int(v) for v in MATCH[f'{c}_shape'])
        anchors = R.make_anchors_xywh(MATCH[f'{c}_base_wh'], H, W)
        labels = R.rpn_match(anchors, torch.from_numpy(MATCH[f'{c}_targets']), B, H, W)
        assert (labels >= 0).sum() > 0 and (labels == -1).sum() > 0
        check_labels_against_reference(c, labels)


def test_sampler_sizes_follow_reference_rule():
    lab = torch.tensor([-1] * 300 + [0, 1, 2] + [-2] * 50)
    pos, neg = R.rpn_sample(lab, 128, 128, perm_pos=torch.arange(3), perm_neg=torch.arange(300))
    assert pos.tolist() == [300, 301, 302] and neg.numel() == 253 and neg[0] == 0          # 128 + 128 - 3 negatives fill the batch


# ------------------------------------------------------------------------------------------------ Fast head samples
FCASES = sorted({k.split('_')[0] for k in MATCH.files if k.startswith('f')})


def fast_case(c):
    B, n, T, H, W = (int(v) for v in MATCH[f'{c}_shape'])
    return [torch.from_numpy(MATCH[f'{c}_prop{b}']) for b in range(B)], torch.from_numpy(MATCH[f'{c}_targets']), B


def check_fast_samples(c, pos, neg):
    """(positives [P,10], negatives [Q,5]) vs the reference's select_positive_negative_samples with randperm = identity and
    unbounded sample sizes: same rows in the same order; the regression targets go through log() (last-bit differences
    between CPUs), everything else is copied data and must be equal."""
    want_pos, want_neg = MATCH[f'{c}_pos'], MATCH[f'{c}_neg']
    assert pos.shape == want_pos.shape and neg.shape == want_neg.shape and len(want_pos) > 0 and len(want_neg) > 0
    assert np.array_equal(neg, want_neg)
    assert np.array_equal(pos[:, :5], want_pos[:, :5]) and np.array_equal(pos[:, 9], want_pos[:, 9])
    assert np.allclose(pos[:, 5:9], want_pos[:, 5:9], rtol=2e-6, atol=1e-6)


def test_fast_samples_match_reference():
    for c in FCASES:
        proposals, targets, B = fast_case(c)
        big = 10 ** 7
        perms = [(torch.arange(p.size(0)), torch.arange(p.size(0))) for p in proposals]
        pos, neg = R.fast_select_samples(proposals, targets, 0.5, 0.5, big, big, perms=perms)
        check_fast_samples(c, pos.numpy(), neg.numpy())
