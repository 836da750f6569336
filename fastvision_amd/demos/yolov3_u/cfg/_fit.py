"""Demo fit loop -- API mirror of the reference's demos/yolov3_u/cfg/_fit.py (Fit, _Train, _Validate).

Same step contract (cfg/_fit.py:41-56): predict = model(images); optimizer.zero_grad(); loss = criterion(predict,
target, model); loss.backward(); optimizer.step().  The reference calls ``loss.item()`` twice per batch; here the
loss is read back once per batch (still one sync, kept for the printed log).
"""
import time

import torch


def Fit(model, args, optimizer, criterion, metric, train_loader, validation_loader):
    best_val_loss = float('inf')
    patient = 0
    for epoch in range(args.start_epoch, args.max_epochs):
        s_time = time.time()
        print('\nEpoch {} learning_rate : {}'.format(epoch + 1, optimizer.param_groups[0]['lr']))
        loss_train = _Train(model, train_loader, optimizer, criterion, epoch)
        loss_val = _Validate(model, validation_loader, criterion, epoch)
        if loss_val < best_val_loss:
            torch.save(model.state_dict(), f'./epoch_{epoch+1}_loss_{loss_val}.pth')
            best_val_loss = loss_val
            patient = 0
        if patient >= 3:
            for group in optimizer.param_groups:
                group['lr'] = group['lr'] * 0.1 if group['lr'] * 0.1 > 1e-8 else 1e-8
            patient = 0
        patient += 1
        print('epoch : {} train_loss : {:.3f} time : {:.3f}'.format(epoch + 1, loss_train, time.time() - s_time))
        print('epoch : {} val_loss : {:.3f} time : {:.3f}'.format(epoch + 1, loss_val, time.time() - s_time))


def _Train(model, train_loader, optimizer, criterion, epoch, log=print):
    model.train()
    loss_batch = []
    for batch, (images, target) in enumerate(train_loader):
        s_time = time.time()
        images = images.cuda(non_blocking=True)
        target = target.cuda(non_blocking=True)
        predict = model(images)
        optimizer.zero_grad()
        loss = criterion(predict, target, model)
        loss.backward()
        optimizer.step()
        value = loss.item()
        loss_batch.append(value)
        if log:
            log('epoch : {} batch : {} / {} loss : {:.3f} time : {:.3f}'.format(epoch + 1, batch + 1, len(train_loader), value,
                                                                                 time.time() - s_time))
    return sum(loss_batch) / len(loss_batch)


def _Validate(model, val_loader, criterion, epoch, log=print):
    model.eval()
    loss_batch = []
    with torch.no_grad():
        for batch, (images, target) in enumerate(val_loader):
            s_time = time.time()
            images = images.cuda(non_blocking=True)
            target = target.cuda(non_blocking=True)
            predict = model(images)
            loss = criterion(predict, target, model)
            value = loss.item()
            loss_batch.append(value)
            if log:
                log('epoch : {} batch : {} / {} loss : {:.3f} time : {:.3f}'.format(epoch + 1, batch + 1, len(val_loader), value,
                                                                                     time.time() - s_time))
    return sum(loss_batch) / len(loss_batch)
