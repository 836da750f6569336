"""demos/yolov3_u/utils/anchor_generator.py surface: ``AnchorGenerator(data_loaders, k, iters, plot, save_dir).get_anchors()`` and
``KMeans`` -- the library's k-means (detection/tools/ANCHOR.py) with the demo's constructor signature."""
from ....detection.tools.ANCHOR import AnchorGenerator as _Base, KMeans  # noqa: F401

__all__ = ['AnchorGenerator', 'KMeans']


class AnchorGenerator(_Base):
    def __init__(self, data_loaders, k=9, iters=100, plot=False, save_dir='./'):
        super().__init__(data_loaders, k=k, iters=iters, plot=plot, save_dir=save_dir)
