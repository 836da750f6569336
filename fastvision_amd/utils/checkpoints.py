"""Checkpoint helpers -- API mirror of the reference's utils/checkpoints.py (host-side; "next" row f-1).

state_dict keys of the fastvision_amd modules equal the reference's, so checkpoints move both ways unchanged."""
import os
from collections import OrderedDict
from copy import deepcopy
from datetime import datetime

import torch
import torch.nn as nn

__all__ = ['is_parallel', 'SqueezeModel', 'LoadStatedict', 'LoadFromParrel', 'SaveModel']


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def SqueezeModel(model, params, squeeze: bool):
    """Set requires_grad = ``squeeze`` on all parameters ('all') or on those whose name contains one of ``params``."""
    for name, value in model.named_parameters():
        if params == 'all' or any(p in name for p in params):
            value.requires_grad = squeeze
    return model


def _read(weights, device):
    try:
        blob = torch.load(weights, map_location=device)
    except Exception:                                  # the reference falls back to '<name>-nonzip.<ext>' (:32-38)
        head, tail = os.path.split(weights)
        parts = tail.split('.')
        blob = torch.load(os.path.join(head, ''.join(parts[:-1]) + '-nonzip.' + parts[-1]), map_location=device)
    return blob['model'] if 'model' in blob.keys() else blob


def _load(model, weights, device, strict, strip):
    own = model.state_dict()
    keep, skipped = OrderedDict(), []
    for k, v in _read(weights, device).items():
        name = k[strip:]
        if name in own and v.size() == own[name].size():
            keep[name] = v
        else:
            skipped.append(k)
    model.load_state_dict(keep, strict=strict)
    print('Load state_dict not load keys : ', skipped)
    return model


def LoadStatedict(model, weights, device, strict=False):
    """Load the entries whose key and shape match (checkpoints.py:28-56); accepts {'model': state_dict} blobs."""
    return _load(model, weights, device, strict, 0)


def LoadFromParrel(model, weights, device, strict=False):
    """Same for a checkpoint saved from nn.DataParallel: strips the 7-character 'module.' prefix (:58-88)."""
    return _load(model, weights, device, strict, 7)


def SaveModel(ckpt, filename, weights_only=True):
    """Stamp the date, unwrap (Distributed)DataParallel, deep-copy, optionally keep only the state_dict (:90-100)."""
    ckpt['date'] = datetime.now().isoformat()
    model = ckpt['model']
    ckpt['model'] = deepcopy(model.module if is_parallel(model) else model)
    if weights_only:
        ckpt['model'] = ckpt['model'].state_dict()
    torch.save(ckpt, filename)
