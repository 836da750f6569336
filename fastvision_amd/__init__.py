"""fastvision_amd -- the YOLOv3 detection training hot path of ielym/fastvision on MI355X (gfx950).

The package mirrors the reference's Python surface for that path and nothing else:

    fastvision_amd.classfication.models.darknet53          (classfication/models/darknet53.py)
    fastvision_amd.detection.neck / .head / .models        (detection/{neck,head,models})
    fastvision_amd.detection.tools                         (IOU.py, BOX.py, GRID.py, NMS.py)
    fastvision_amd.metrics                                 (map.py: CalculateMAP)
    fastvision_amd.datasets                                (detection_dataloader.py: decode on host, the rest on device)
    fastvision_amd.loss                                    (yolov3_loss.py, iou_loss.py, classification_loss.py)
    fastvision_amd.utils.Fit, .utils.sheduler, .utils.checkpoints   (utils/fit.py step contract; f-1 helpers)
    fastvision_amd.demos.yolov3_u.{models,utils,cfg}       (demos/yolov3_u: YoloV3, ComputeLoss, Fit/_Train, nms, postProcess)

plus ``FusedAdam``, ``parallel`` (one process per GPU, RCCL gradient all-reduce) and ``graphs`` (HIP-graph replay of the
shape-static inference forward).  All device arithmetic
runs in hand-written HIP kernels behind the C ABI of include/fastvision_amd.h (csrc/); importing the package
is cheap, the shared library is loaded on first use and its absence is an error (no CPU fallback).
"""
from .ops import compute_dtype, get_compute_dtype, set_compute_dtype  # noqa: F401
from .optim import FusedAdam  # noqa: F401

__version__ = '0.1.0'
