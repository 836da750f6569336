"""YOLOv3 assembly -- API mirror of the reference's detection/models/yolov3.py (Yolov3, yolov3).

``forward(images, val=False)``: train mode returns the raw head list (yolov3.py:54); eval / ``val=True`` also
returns decoded boxes [B, sum 3*g*g, 5+C] (yolov3.py:35-53), produced for all three levels by one HIP kernel
(``fva_yolo_decode``); the reference's missing ``offset`` helper is the [H,W,(x,y)] cell grid.
"""
import torch
import torch.nn as nn

from ...detect_ops import yolo_decode
from ...parallel import refuse_dataparallel_replica

__all__ = ['Yolov3', 'yolov3']


class Yolov3(nn.Module):
    def __init__(self, backbone, neck, head, anchors, num_anchors_per_level, in_channels=3, num_classes=80, training=False):
        super().__init__()
        self.training = training
        anchors = anchors.view(-1, 2)
        self.anchors_per_level = []
        start = 0
        for n in num_anchors_per_level:
            self.anchors_per_level.append(anchors[start:start + n].view(n, 1, 1, 2))
            start += n
        self._anchor_lists = [[(float(a[0]), float(a[1])) for a in lvl.view(-1, 2)] for lvl in self.anchors_per_level]
        self.num_classes = num_classes
        self.backbone = backbone(in_channels=in_channels, including_top=False)
        self.backbone_strides_per_level = self.backbone.backbone_strides_per_level()
        self.backbone_channels_per_level = self.backbone.backbone_channels_per_level()
        self.neck = neck(feature_channels=self.backbone_channels_per_level)
        self.head = head(feature_channels=self.backbone_channels_per_level, num_levels=len(self.backbone_channels_per_level),
                         num_anchors_per_level=num_anchors_per_level, num_classes=num_classes)

    def forward(self, images, val=False):
        refuse_dataparallel_replica(self)        # nn.DataParallel over > 1 device: one process per GPU instead (parallel.py)
        head_out = self.head(self.neck(self.backbone(images)))
        if self.training and not val:
            return head_out
        return head_out, yolo_decode(list(head_out), self._anchor_lists, self.backbone_strides_per_level, variant=0)


def yolov3(backbone=None, neck=None, head=None, anchors=None, num_anchors_per_level=None, in_channels=3, num_classes=80,
           training=False):
    # the reference only fills neck/head defaults when ``backbone`` is None (yolov3.py:62,65); here each default is
    # filled on its own, which is a superset of that behaviour
    if backbone is None:
        from ...classfication.models import darknet53 as backbone
    if neck is None:
        from ..neck import yolov3neck as neck
    if head is None:
        from ..head import yolov3head as head
    return Yolov3(backbone=backbone, neck=neck, head=head, anchors=anchors, num_anchors_per_level=num_anchors_per_level,
                  in_channels=in_channels, num_classes=num_classes, training=training)
