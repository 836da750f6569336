"""CPU oracle for the YOLOv3 training hot path -- TEST INFRASTRUCTURE ONLY.

This package is an independent, pure-PyTorch (CPU, fp32) restatement of the
algorithms on the reference's detection training path.  It exists to *check*
the HIP product path in ``fastvision_amd`` and to act as the timed
``cpu_baseline`` leg of ``bench.py``.  It is never shipped and never on the
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.

Parity status: PINNED.  Every function here is checked against golden vectors
captured from the reference itself, imported and run in the build container
(see ``oracle/make_golden.py`` and ``tests/golden/``; the checks are
``tests/test_oracle_golden.py``).  Third-party arithmetic underneath the
reference (``nn.Conv2d``/``nn.BatchNorm2d``/``nn.SiLU``/``nn.Upsample``/
``torch.optim.Adam``) is PyTorch's; the reference holds no tests of its own,
so those are pinned by the same captured vectors (torch 2.10 CPU kernels).

Reference files restated (paths relative to the reference root):
  classfication/models/darknet53.py, detection/neck/yolov3neck.py,
  detection/head/yolov3head.py, detection/models/yolov3.py,
  loss/yolov3_loss.py, loss/iou_loss.py, loss/classification_loss.py,
  detection/tools/{IOU,BOX,GRID}.py, datasets/common/id_2_onehot.py,
  demos/yolov3_u/models/{darknet,yolov3}.py, demos/yolov3_u/utils/{lossv3,iou,box}.py,
  utils/fit.py:47-71 and demos/yolov3_u/cfg/_fit.py:36-58 (step contract).
"""
