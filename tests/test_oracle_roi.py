"""RoIAlign restatement (oracle/roi_align.py, scope row f-4): known answers and properties on the CPU.  The arithmetic is
torchvision's, absent here: parity unpinned (the oracle's header says so); these tests pin the restatement to cases whose
answers follow from the published algorithm by hand."""
import numpy as np

from oracle import roi_align as R


def test_constant_map_gives_constant_bins():
    f = np.full((2, 3, 9, 11), 2.5, dtype=np.float32)
    out = R.roi_align(f, [[1, 1.2, 0.7, 8.9, 6.3], [0, 0, 0, 11, 9]], (7, 7))
    assert out.shape == (2, 3, 7, 7) and np.allclose(out, 2.5, atol=1e-6)


def test_linear_ramp_is_reproduced_at_bin_centres():
    """bilinear interpolation of a plane is exact, and the mean of a bin's symmetric samples is the plane at the bin centre"""
    H, W = 16, 20
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    f = (0.5 * xx + 2.0 * yy + 1.0)[None, None]
    box = [0, 2.0, 3.0, 16.0, 10.0]                                     # 14 x 7 cells, well inside
    out = R.roi_align(f, [box], (7, 7))[0, 0]
    cy = 3.0 + (np.arange(7) + 0.5) * (7.0 / 7)
    cx = 2.0 + (np.arange(7) + 0.5) * (14.0 / 7)
    want = 0.5 * cx[None, :] + 2.0 * cy[:, None] + 1.0
    assert np.allclose(out, want, atol=1e-4)


def test_tiny_and_outside_boxes():
    f = np.arange(2 * 5 * 5, dtype=np.float32).reshape(1, 2, 5, 5)
    # a degenerate box is widened to 1 x 1 (aligned=False): one sample per bin inside the cell [2,3] x [2,3]
    out = R.roi_align(f, [[0, 2.0, 2.0, 2.0, 2.0]], (2, 2))
    assert out.shape == (1, 2, 2, 2)
    assert np.allclose(out[0, 0, 0, 0], 0.25 * 0 + f[0, 0, 2, 2] + 0.25 * 5 + 0.25 * 1)      # sample at (2.25, 2.25): + .25 row, + .25 col
    # entirely outside the map: every sample is beyond size -> zeros
    assert np.all(R.roi_align(f, [[0, 7.0, 7.0, 9.0, 9.0]], (3, 3)) == 0)
    # empty box list
    assert R.roi_align(f, np.zeros((0, 5), np.float32), (7, 7)).shape == (0, 2, 7, 7)


def test_backward_is_the_adjoint_of_forward():
    rng = np.random.default_rng(0)
    f = rng.standard_normal((2, 3, 8, 9)).astype(np.float32)
    boxes = np.array([[0, 0.5, 1.0, 7.5, 6.0], [1, -2.0, -1.0, 4.0, 9.5], [1, 3.0, 3.0, 3.2, 3.1]], dtype=np.float32)
    g = rng.standard_normal((3, 3, 4, 5)).astype(np.float32)
    out = R.roi_align(f, boxes, (4, 5))
    df = R.roi_align_backward(g, boxes, f.shape)
    assert abs(float((out.astype(np.float64) * g).sum()) - float((df.astype(np.float64) * f).sum())) < 1e-3
