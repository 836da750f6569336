"""Demo fit loop -- API mirror of the reference's demos/yolov3_u/cfg/_fit.py (Fit, _Train, _Validate).

Step contract (cfg/_fit.py:41-56): predict = model(images); optimizer.zero_grad(); loss = criterion(predict, target,
model); loss.backward(); optimizer.step().  Training and validation share one epoch runner here; the per-batch log
line, the best-validation checkpoint and the divide-by-ten-every-fourth-epoch-without-improvement rule are the
reference's (:10-34).  The loss is read back once per batch (the reference reads it twice), for that log line.
"""
import time

import torch

LR_FLOOR = 1e-8
PATIENCE = 3
_BATCH_LINE = 'epoch : {} batch : {} / {} loss : {:.3f} time : {:.3f}'


def _run_epoch(model, loader, criterion, epoch, optimizer=None, log=print):
    """One pass over ``loader``; with an optimizer it trains, without one it evaluates under no_grad.  Mean batch loss."""
    training = optimizer is not None
    model.train(training)
    total, count = 0.0, 0
    with torch.set_grad_enabled(training):
        for count, (images, target) in enumerate(loader, start=1):
            tick = time.time()
            predict = model(images.cuda(non_blocking=True))
            target = target.cuda(non_blocking=True)
            if training:
                optimizer.zero_grad()
            loss = criterion(predict, target, model)
            if training:
                loss.backward()
                optimizer.step()
            value = loss.item()
            total += value
            if log:
                log(_BATCH_LINE.format(epoch + 1, count, len(loader), value, time.time() - tick))
    return total / count


def _Train(model, train_loader, optimizer, criterion, epoch, log=print):
    return _run_epoch(model, train_loader, criterion, epoch, optimizer=optimizer, log=log)


def _Validate(model, val_loader, criterion, epoch, log=print):
    return _run_epoch(model, val_loader, criterion, epoch, optimizer=None, log=log)


def _decay_lr(optimizer, factor=0.1):
    for group in optimizer.param_groups:
        group['lr'] = max(group['lr'] * factor, LR_FLOOR)


def Fit(model, args, optimizer, criterion, metric, train_loader, validation_loader):
    best, since_best = float('inf'), 0
    for epoch in range(args.start_epoch, args.max_epochs):
        started = time.time()
        print('\nEpoch {} learning_rate : {}'.format(epoch + 1, optimizer.param_groups[0]['lr']))
        results = (('train_loss', _Train(model, train_loader, optimizer, criterion, epoch)),
                   ('val_loss', _Validate(model, validation_loader, criterion, epoch)))
        val = results[1][1]
        if val < best:
            best, since_best = val, 0
            torch.save(model.state_dict(), f'./epoch_{epoch+1}_loss_{val}.pth')
        if since_best >= PATIENCE:                 # checked before the increment, as the reference does
            _decay_lr(optimizer)
            since_best = 0
        since_best += 1
        for name, value in results:
            print('epoch : {} {} : {:.3f} time : {:.3f}'.format(epoch + 1, name, value, time.time() - started))
