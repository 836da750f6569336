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
