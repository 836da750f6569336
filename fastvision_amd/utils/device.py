"""set_device (reference utils/device.py:4-27): restrict the visible GPUs and return the torch device.  One process drives ONE
MI355X here (data parallelism is one process per GPU, fastvision_amd.parallel), so with several ids only the visibility mask is
set; HIP honours CUDA_VISIBLE_DEVICES (and HIP_VISIBLE_DEVICES) the way the reference expects."""
import os

import torch

__all__ = ['set_device']


def set_device(devices):
    if len(devices) == 0:
        os.environ['CUDA_VISIBLE_DEVICES'] = '-1'
    else:
        os.environ['CUDA_VISIBLE_DEVICES'] = ','.join(str(d) for d in devices)
    cuda = bool(len(devices)) and torch.cuda.is_available()
    device = torch.device('cuda' if cuda else 'cpu')
    msg = f'Device : {device.type} \t'
    if device.type == 'cuda':
        msg += f"CUDA_VISIBLE_DEVICES : {os.environ['CUDA_VISIBLE_DEVICES']}\t"
    print(msg)
    return device
