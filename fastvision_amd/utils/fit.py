"""Training loop -- API mirror of the reference's utils/fit.py (class Fit): the CALLER of the accelerated path.

``_train`` keeps the reference's per-batch contract (fit.py:52-66): pred = model(images); optimizer.zero_grad();
loss = self.loss(pred, labels); loss.backward(); optimizer.step(); scheduler.step() per epoch.  Validation in the
reference also runs NMS + mAP (fit.py:73-105: torchvision / metrics, the "next" rows f-2 of the scope table); here
``_val`` reports the validation loss only.
"""
import torch

__all__ = ['Fit']


class Fit:
    def __init__(self, model, device, optimizer, scheduler, loss, end_epoch, start_epoch=0, train_loader=None, val_loader=None,
                 test_loader=None, data_dict=None, log_every=0):
        self.model, self.device, self.optimizer, self.scheduler, self.loss = model, device, optimizer, scheduler, loss
        self.start_epoch, self.end_epoch = start_epoch, end_epoch
        self.train_loader, self.val_loader, self.test_loader = train_loader, val_loader, test_loader
        self.category_names = {k: v for k, v in enumerate((data_dict or {}).get('categories', []))}
        self.log_every = log_every
        self.history = []

    def run_epoches(self):
        for epoch in range(self.start_epoch, self.end_epoch):
            self._train(epoch)
            if self.val_loader:
                self._val()
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
            pred = self.model(images)
            self.optimizer.zero_grad()
            loss = self.loss(pred, labels)
            loss.backward()
            self.optimizer.step()
            losses.append(loss.detach())          # device tensor: no per-step host sync (the reference's tqdm .item() does one)
            if self.log_every and (batch_idx + 1) % self.log_every == 0:
                print(f'Epoch {epoch + 1} batch {batch_idx + 1} loss {float(losses[-1])}')
        if self.scheduler is not None:
            self.scheduler.step()
        self.history.append(torch.stack([l.reshape(()) for l in losses]).float().cpu().tolist())
        return self.history[-1]

    def _val(self):
        self.model.eval()
        out = []
        with torch.no_grad():
            for images, labels in self.val_loader:
                images, labels = self._to_device(images, labels)
                head_out, _ = self.model(images, val=True)
                out.append(float(self.loss(head_out, labels)))
        return out

    def _test(self):
        pass
