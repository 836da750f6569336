"""LR schedules -- API mirror of the reference's utils/sheduler.py (the file name keeps the reference's spelling).

Host-side only: three LambdaLR factories whose lambda returns the *absolute* LR (the reference builds its optimizers
with lr=1, generate/template-yolov3/train.py:51-62) and a warm-up + cosine-restart scheduler.  "Next" row f-1 of
the scope table (caller of the hot path); values are checked against ones captured from the reference
(tests/golden/sched.npz)."""
import math
from bisect import bisect_right

from torch.optim import lr_scheduler

__all__ = ['CosineLR', 'LinearLR', 'ExponentialLR', 'WarmupCosineLR']


def CosineLR(optimizer, steps, initial_lr, last_lr):
    """initial_lr -> last_lr along half a cosine over ``steps`` (sheduler.py:6-19)."""
    return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda s: (1 - math.cos(s * math.pi / steps)) / 2 * (last_lr - initial_lr) + initial_lr)


def LinearLR(optimizer, steps, initial_lr, last_lr):
    """straight line reaching last_lr at step ``steps - 1`` (sheduler.py:21-35)."""
    return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda s: (1 - s / (steps - 1)) * (initial_lr - last_lr) + last_lr)


def ExponentialLR(optimizer, steps, initial_lr, last_lr):
    """geometric interpolation (sheduler.py:37-43)."""
    ratio = (last_lr / initial_lr) ** (1 / steps)
    return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda s: initial_lr * ratio ** s)


class WarmupCosineLR(lr_scheduler._LRScheduler):
    """Linear warm-up for ``warmup_iters`` steps, then cosine cycles between consecutive milestones; each cycle starts
    at base_lr * cycle_decay**(cycle-1) and decays to base_lr * min_ratio (sheduler.py:46-74)."""

    def __init__(self, optimizer, milestones, min_ratio=0., cycle_decay=1., warmup_iters=1000, warmup_factor=1. / 10, last_epoch=-1):
        if list(milestones) != sorted(milestones):
            raise ValueError('Milestones should be a list of increasing integers. Got {}'.format(milestones))
        self.milestones = [warmup_iters] + list(milestones)
        self.min_ratio, self.cycle_decay = min_ratio, cycle_decay
        self.warmup_iters, self.warmup_factor = warmup_iters, warmup_factor
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        e = self.last_epoch
        if e < self.warmup_iters:
            a = float(e) / self.warmup_iters
            f = self.warmup_factor * (1 - a) + a
            return [b * f for b in self.base_lrs]
        cycle = min(bisect_right(self.milestones, e), len(self.milestones) - 1)
        lo, hi = self.milestones[cycle - 1], self.milestones[cycle]
        frac = min((e - lo) / (hi - lo), 1.)
        return [b * self.min_ratio + (b * self.cycle_decay ** (cycle - 1) - b * self.min_ratio) * (1 + math.cos(math.pi * frac)) / 2
                for b in self.base_lrs]
