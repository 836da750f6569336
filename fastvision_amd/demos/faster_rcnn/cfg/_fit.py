"""Fit loop of the reference's Faster R-CNN demo -- API mirror of demos/faster_rcnn/cfg/_fit.py (clip_gradient, Fit, _Train).

Step contract (_fit.py:35-50): losses = model(images, targets); optimizer.zero_grad(); loss = the four losses summed;
loss.backward(); clip_gradient(model, 10.) ("for vgg only"); optimizer.step(); the learning rate is divided by ten at every ninth
epoch (:22-24) and the model's state_dict is saved after each epoch (:30).  Differences: the batch goes to the GPU here (the
reference leaves its .cuda() calls commented out and relies on DataParallel), the global norm is computed on the device with one
read-back instead of one per parameter, and the per-batch print reads the five loss values back in one transfer.
"""
import torch

CLIP_NORM = 10.


def clip_gradient(model, clip_norm):
    """Scale every gradient by clip_norm / max(global L2 norm, clip_norm) (_fit.py:6-17).  Returns the norm."""
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    if not grads:
        return 0.0
    total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)).float()).item()     # sqrt(sum of squared norms), one read-back
    scale = clip_norm / max(total, clip_norm)
    if scale != 1.0:
        torch._foreach_mul_(grads, scale)
    return total


def _Train(model, train_loader, optimizer, log=print):
    model.train()
    for images, targets in train_loader:
        if torch.cuda.is_available():
            images, targets = images.cuda(non_blocking=True), targets.cuda(non_blocking=True)
        _, loss_rpn_cls, loss_rpn_box, loss_fast_cls, loss_fast_box = model(images, targets)
        optimizer.zero_grad()
        parts = torch.stack([l.reshape(()) for l in (loss_rpn_cls, loss_rpn_box, loss_fast_cls, loss_fast_box)])
        loss = parts.sum()
        loss.backward()
        clip_gradient(model, CLIP_NORM)
        optimizer.step()
        if log:
            log(*torch.cat([loss.detach().reshape(1), parts.detach()]).tolist())


def Fit(model, args, optimizer, train_loader, validation_loader=None):
    for epoch in range(args.start_epoch, args.total_epoch):
        if (epoch + 1) % 9 == 0:
            for group in optimizer.param_groups:
                group['lr'] = group['lr'] * 0.1
        print('\nEpoch {} learning_rate : {}'.format(epoch + 1, optimizer.param_groups[0]['lr']))
        _Train(model, train_loader, optimizer)
        state = (model.module if hasattr(model, 'module') else model).state_dict()
        torch.save(state, f'./{epoch + 1}.pth')
