"""The CPU restatement of the Faster R-CNN training forward (oracle/faster.py, scope row f-4) against the reference's own model
(tests/golden/faster_step.npz): losses and the gradient of every parameter.  The parameters live in the product's mirror classes
(pure torch containers at construction: nothing here touches a GPU or the HIP library)."""
import os

import numpy as np
import torch

from oracle import faster as OF

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'faster_step.npz'))


def build_cpu():
    from fastvision_amd.demos.faster_rcnn.models import Faster_Rcnn
    seed, B, H, W, T, NC = (int(v) for v in G['meta'])
    torch.manual_seed(seed)
    model = Faster_Rcnn(training=True, num_classes=NC, base_anchors=torch.from_numpy(G['base_anchors_px']), rpn_positives_per_image=16,
                        rpn_negatives_per_image=48, fast_positives_per_image=8, fast_negatives_per_image=24)
    for m in model.backbone.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data *= float(G['conv_scale'][0])
    return model, B


def test_oracle_losses_and_gradients_match_reference():
    torch.set_num_threads(max(1, min(8, len(os.sched_getaffinity(0)))))
    model, B = build_cpu()
    perms = [(torch.from_numpy(G[f'perm{2 * i}']), torch.from_numpy(G[f'perm{2 * i + 1}'])) for i in range(2 * B)]
    out = OF.training_losses(model, torch.from_numpy(G['images']), torch.from_numpy(G['targets']), perms)
    losses = torch.stack([l.reshape(()) for l in out[1:]])
    np.testing.assert_allclose(losses.detach().numpy(), G['losses'], rtol=1e-5)
    losses.sum().backward()
    for k, p in model.named_parameters():
        want = G['gstat_' + k]
        gr = p.grad.double()
        assert abs(gr.norm().item() - want[2]) <= 1e-4 * max(want[2], 1e-12), k
        assert abs(gr.sum().item() - want[0]) <= 1e-3 * max(want[1], 1e-12), k
