"""Oracle: the reference's per-batch training step on CPU, and a small driver around it.

Step contract (utils/fit.py:52-66; demos/yolov3_u/cfg/_fit.py:41-56):
    pred = model(images); optimizer.zero_grad(); loss = criterion(pred, labels); loss.backward();
    optimizer.step(); loss.item()
Optimizer as the demo builds it (demos/yolov3_u/train.py:66-70): Adam(betas=(0.937, 0.999), weight_decay=5e-4).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import time

import torch

from . import losses, model as omodel


def make_library(seed, num_classes=80, in_channels=3):
    torch.manual_seed(seed)
    net = omodel.LibYolov3(omodel.coco_anchors_px(), (3, 3, 3), in_channels, num_classes, training=True)
    net.train()

    def criterion(pred, labels, ratios=(0.05, 1.0, 0.5)):
        return losses.yolov3_loss(pred, labels, net.anchors_per_level, net.backbone_strides_per_level, *ratios)
    return net, criterion


def make_demo(seed, num_classes=80, in_channels=3):
    torch.manual_seed(seed)
    anchors = omodel.coco_anchors_feature()
    net = omodel.DemoYoloV3(in_channels, num_classes, anchors)
    net.train()

    def criterion(pred, labels):
        return losses.demo_loss(pred, labels, anchors)
    return net, criterion


def make_adam(net, lr=1e-4, weight_decay=5e-4):
    return torch.optim.Adam(net.parameters(), lr=lr, betas=(0.937, 0.999), weight_decay=weight_decay)


def train_steps(net, criterion, optimizer, images, labels, steps):
    """Run ``steps`` reference-shaped steps on one fixed batch; returns (losses, seconds per step list)."""
    out, times = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        pred = net(images)
        optimizer.zero_grad()
        loss = criterion(pred, labels)
        loss.backward()
        optimizer.step()
        out.append(float(loss.item()))
        times.append(time.perf_counter() - t0)
    return out, times
