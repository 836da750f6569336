"""API mirror of the reference's demos/faster_rcnn/inference.py on the device ops: ``preProcess`` (ResizeByMax + Padding(128) + / 255
in one kernel launch), ``postProcess`` (feature cells -> input pixels -> original image, clamps, the 5-pixel filter, per-class NMS),
``anchor_fn``, ``model_fn`` and the ``Inference`` loop (detections are returned / printed; the reference's cv2.imshow viewer is not
part of this package)."""
import os
from glob import glob

import numpy as np
import torch

from .data_gen import DeviceAugmenter
from .models.faster import Faster_Rcnn
from .utils import get_base_anchor, non_max_suppression, xywh2xyxy

__all__ = ['preProcess', 'postProcess', 'anchor_fn', 'model_fn', 'Inference']


def preProcess(img_path, input_size, device='cuda'):
    """-> (image [1,3,S,S] float32 on the device, original RGB image, resize_ratio, padding_left, padding_top, ori_height, ori_width)
    (inference.py:55-85; the image argument may also be a decoded uint8 RGB array)"""
    from ...datasets.detection_dataloader import _decode_rgb
    ori_img = _decode_rgb(img_path) if isinstance(img_path, (str, os.PathLike)) else np.asarray(img_path)
    ori_height, ori_width = ori_img.shape[:2]
    resize_ratio = input_size / max(ori_height, ori_width)
    resize_width, resize_height = int(ori_width * resize_ratio), int(ori_height * resize_ratio)
    padding_left, padding_top = (input_size - resize_width) // 2, (input_size - resize_height) // 2
    image, _ = DeviceAugmenter(input_size, device).val_batch([(ori_img, np.zeros((0, 4), np.float32), np.zeros((0,), np.float32))])
    return image, ori_img, resize_ratio, padding_left, padding_top, ori_height, ori_width


def postProcess(proposals, args, resize_ratio, padding_left, padding_top, ori_width, ori_height):
    """proposals [n, 6] = xywh in feature cells, class, score (the model's eval output for one image) -> (scores [m,1],
    categories [m,1], boxes [m,4] xyxy in original-image pixels) (inference.py:87-118)"""
    p = proposals.clone()
    p[:, 0:4] = p[:, 0:4] * args.backbone_stride
    p[:, 0] = ((p[:, 0] - padding_left) / resize_ratio).clamp(0, ori_width - 1)
    p[:, 1] = ((p[:, 1] - padding_top) / resize_ratio).clamp(0, ori_height - 1)
    p[:, 2] = (p[:, 2] / resize_ratio).clamp(0, ori_width)
    p[:, 3] = (p[:, 3] / resize_ratio).clamp(0, ori_height)
    p = p[(p[:, 2] > 5) & (p[:, 3] > 5)]
    p[:, 0:4] = xywh2xyxy(p[:, 0:4])
    p[:, 0], p[:, 2] = p[:, 0].clamp(0, ori_width - 1), p[:, 2].clamp(0, ori_width - 1)
    p[:, 1], p[:, 3] = p[:, 1].clamp(0, ori_height - 1), p[:, 3].clamp(0, ori_height - 1)
    results = non_max_suppression(p, conf_thres=args.inference_conf_thres, iou_thres=args.inference_iou_thres, max_det=300)
    return results[:, 5:6], results[:, 4:5], results[:, 0:4]


def anchor_fn(scales, ratios):
    return torch.from_numpy(get_base_anchor(scales=scales, ratios=ratios))


def model_fn(args, base_anchors):
    model = Faster_Rcnn(
        training=args.training, in_channels=args.in_channels, num_classes=args.num_classes, base_anchors=base_anchors,
        backbone_stride=args.backbone_stride, backbone_output_channels=args.backbone_output_channels, backbone_weights=args.backbone_weights,
        rpn_positive_iou_thres=args.rpn_positive_iou_thres, rpn_negative_iou_thres=args.rpn_negative_iou_thres,
        rpn_positives_per_image=args.rpn_positives_per_image, rpn_negatives_per_image=args.rpn_negatives_per_image,
        rpn_pre_nms_top_n=args.rpn_pre_nms_top_n, rpn_post_nms_top_n=args.rpn_post_nms_top_n, rpn_nms_thresh=args.rpn_nms_thresh,
        fast_multi_reg_head=args.fast_multi_reg_head, fast_positive_iou_thres=args.fast_positive_iou_thres,
        fast_negative_iou_thres=args.fast_negative_iou_thres, fast_positives_per_image=args.fast_positives_per_image,
        fast_negatives_per_image=args.fast_negatives_per_image, fast_roi_pool=args.fast_roi_pool)
    if getattr(args, 'inference_weights', None):
        model.load_state_dict(torch.load(args.inference_weights), True)
    model.eval()
    return model.to(getattr(args, 'device', 'cuda'))


@torch.no_grad()
def Inference(args, image_dir, log=print):
    """The reference's loop (inference.py:158-224) over ``image_dir``/*.jpg|png: returns {file: (scores, categories, boxes)}."""
    base_anchors = anchor_fn(args.scales, args.ratios)
    model = model_fn(args=args, base_anchors=base_anchors)
    out = {}
    files = sorted(glob(os.path.join(image_dir, '*.jpg')) + glob(os.path.join(image_dir, '*.png')))
    for file in files:
        image, ori_img, resize_ratio, padding_left, padding_top, ori_height, ori_width = preProcess(file, args.input_size, getattr(args, 'device', 'cuda'))
        predicts = model(image)
        scores, categories, boxes = postProcess(predicts[0], args, resize_ratio, padding_left, padding_top, ori_width, ori_height)
        out[file] = (scores, categories, boxes)
        log(f'{os.path.basename(file)}: {boxes.size(0)} detections')
    return out
