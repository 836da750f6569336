"""Training loop -- API mirror of the reference's utils/fit.py (class Fit): the CALLER of the accelerated path.

``_train`` keeps the reference's per-batch contract (fit.py:52-66): pred = model(images); optimizer.zero_grad();
loss = self.loss(pred, labels); loss.backward(); optimizer.step(); scheduler.step() per epoch.  ``_val`` follows
fit.py:73-105: eval-mode forward with decode, validation loss, NMS (conf 0.25, IoU 0.45, 300 detections) and
CalculateMAP over IoU 0.5:0.95 -- decode and NMS run in the HIP kernels, one NMS pass per batch instead of one per image.
"""
import numpy as np
import torch

from ..detection.tools import non_max_suppression_images, xywh2xyxy
from ..metrics import CalculateMAP
from .checkpoints import SaveModel

__all__ = ['Fit', 'stall_watch']


def _freeze_garbage_collector():
    """After the first step everything long-lived exists (modules, optimizer state, the library): collect once and move the
    survivors to the permanent generation.  A later generation-2 pass over all of it stops the launching thread for
    0.2-0.3 s -- several steps' worth of queued GPU work -- every few hundred steps (measured in bench.py)."""
    import gc
    gc.collect()
    gc.freeze()


class stall_watch:
    """A rare host stall is on record (4 of ~120 eager benchmark runs of rounds 2 and 3: the HOST stopped issuing launches for
    180-770 ms inside one step, cause unknown, the GPU ran dry).  Re-running does not find such a thing; naming the blocking call when
    it next happens does: around every training step a watchdog (faulthandler's own thread: no cost while nothing hangs) dumps every
    thread's Python stack to stderr once when the step has not returned after ``seconds``.  bench.py arms the same watchdog in its
    timed loop.  FVA_STALL_WATCH=0 switches it off; FVA_STALL_WATCH=<seconds> changes the threshold (default 0.5 s here: a step may
    legitimately wait for its data loader)."""

    def __init__(self, seconds=None):
        import os
        env = os.environ.get('FVA_STALL_WATCH')
        self.seconds = float(env) if env not in (None, '') else (0.5 if seconds is None else seconds)

    def __enter__(self):
        if self.seconds > 0:
            import faulthandler
            import sys
            faulthandler.dump_traceback_later(self.seconds, repeat=False, file=sys.stderr, exit=False)
        return self

    def __exit__(self, *a):
        if self.seconds > 0:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()


class Fit:
    def __init__(self, model, device, optimizer, scheduler, loss, end_epoch, start_epoch=0, train_loader=None, val_loader=None,
                 test_loader=None, data_dict=None, log_every=0, save_last='last.pth'):
        self.model, self.device, self.optimizer, self.scheduler, self.loss = model, device, optimizer, scheduler, loss
        self.start_epoch, self.end_epoch = start_epoch, end_epoch
        self.train_loader, self.val_loader, self.test_loader = train_loader, val_loader, test_loader
        self.category_names = {k: v for k, v in enumerate((data_dict or {}).get('categories', []))}
        self.log_every = log_every
        self.save_last = save_last                  # None / '' skips the per-epoch checkpoint the reference always writes
        self.history = []

    def run_epoches(self):
        for epoch in range(self.start_epoch, self.end_epoch):
            self._train(epoch)
            if self.val_loader:
                self._val()
            if self.save_last:                      # fit.py:36-41: the whole model + optimizer state after every epoch
                SaveModel({'model': self.model, 'optimizer': self.optimizer.state_dict()}, self.save_last, weights_only=True)
        if self.test_loader:
            self._test()

    def _to_device(self, images, labels):
        if getattr(self.device, 'type', str(self.device)) == 'cuda':
            images = images.cuda(non_blocking=True)
            labels = labels.cuda(non_blocking=True)
        return images, labels

    def _train(self, epoch):
        assert self.train_loader, 'train_loader can not be None'
        self.model.train()
        losses = []
        for batch_idx, (images, labels) in enumerate(self.train_loader):
            images, labels = self._to_device(images, labels)
            with stall_watch():
                pred = self.model(images)
                self.optimizer.zero_grad()
                loss = self.loss(pred, labels)
                loss.backward()
                self.optimizer.step()
            losses.append(loss.detach())          # device tensor: no per-step host sync (the reference's tqdm .item() does one)
            del pred, loss                        # the graph of this step must not outlive it (it would sit in memory beside the next step's)
            if epoch == self.start_epoch and batch_idx == 0:
                _freeze_garbage_collector()
            if self.log_every and (batch_idx + 1) % self.log_every == 0:
                print(f'Epoch {epoch + 1} batch {batch_idx + 1} loss {float(losses[-1])}')
        if self.scheduler is not None:
            self.scheduler.step()
        self.history.append(torch.stack([l.reshape(()) for l in losses]).float().cpu().tolist())
        return self.history[-1]

    def _val(self):
        """Returns {'loss': [...per batch], 'map_each_iou', 'map_each_cls', 'map_each_cls_idx'}.  Like the reference
        (fit.py:79) this iterates ``train_loader`` when no ``val_loader`` was given."""
        loader = self.val_loader or self.train_loader
        map_est = CalculateMAP(map_iou_values=np.linspace(0.5, 0.95, 10))
        self.model.eval()
        losses = []
        with torch.no_grad():
            for images, labels in loader:
                images, labels = self._to_device(images, labels)
                head_out, results = self.model(images, val=True)
                losses.append(float(self.loss(head_out, labels)))
                scale = torch.tensor([images.size(3), images.size(2), images.size(3), images.size(2)]).to(labels)
                dets = non_max_suppression_images(results, conf_thres=0.25, iou_thres=0.45, max_det=300)
                for img_idx, (conf, cls, xyxy) in enumerate(dets):
                    predict = torch.cat([cls.float(), conf, xyxy], dim=1)
                    target = labels[labels[:, 0] == img_idx, 1:].clone()
                    target[:, 1:] = xywh2xyxy(target[:, 1:]) * scale
                    map_est.process_one(predict, target)
        out = {'loss': losses}
        if map_est.seen_all_targets_cls and map_est.correct_all_images:
            out['map_each_iou'], out['map_each_cls'], out['map_each_cls_idx'] = map_est.fetch()
        self.last_val = out
        return out

    def _test(self):
        pass
