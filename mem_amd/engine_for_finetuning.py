"""Finetuning loop -- drop-in for /root/reference/mem/engine_for_finetuning.py (train_class_batch :30-33,
train_one_epoch :41-212, evaluate :215-244), SURVEY section 8 row f3.

Same signatures and returned meter dicts.  Differences, all forced by the fused engine and stated here:
bf16 needs no loss scaling (``loss_scaler`` is utils.NativeScalerWithGradNormCount: backward -> fused clip ->
grouped AdamW); the deepspeed branch (``loss_scaler is None``) and the wandb image logging are not carried;
gradient accumulation is not supported by the flat gradient buffer (``update_freq`` must be 1); ``model_ema``
is updated through its own ``update(model)`` if one is passed."""
import math
import sys
from typing import Iterable, Optional

import torch

from . import utils


def accuracy(output, target, topk=(1,)):
    """timm.utils.accuracy (un-vendored; restated from its published definition): top-k accuracies in percent."""
    maxk = min(max(topk), output.size(1))
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / batch_size for k in topk]


def train_class_batch(model, samples, target, criterion):
    outputs = model(samples)
    loss = criterion(outputs, target)
    return loss, outputs


def train_one_epoch(args, model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable, optimizer,
                    device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, model_ema=None,
                    mixup_fn=None, log_writer=None, start_steps=None, lr_schedule_values=None,
                    wd_schedule_values=None, num_training_steps_per_epoch=None, update_freq=None):
    update_freq = update_freq or 1
    if update_freq != 1:
        raise NotImplementedError("update_freq > 1: the fused engine's flat gradient buffer is rewritten every step")
    if loss_scaler is None:
        raise NotImplementedError("the deepspeed branch (loss_scaler=None) is not part of the fused path")
    start_steps = start_steps or 0
    model.train(True)
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    metric_logger.add_meter("min_lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    optimizer.zero_grad()
    for data_iter_step, (samples, targets) in enumerate(metric_logger.log_every(data_loader, 10, header)):
        step = data_iter_step // update_freq
        if num_training_steps_per_epoch is not None and step >= num_training_steps_per_epoch:
            continue
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[it] * param_group.get("lr_scale", 1.0)
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[it]
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        loss, output = train_class_batch(model, samples, targets, criterion)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=False,
                                update_grad=True)
        optimizer.zero_grad()
        if model_ema is not None:
            model_ema.update(model)
        loss_scale_value = loss_scaler.state_dict()["scale"]
        torch.cuda.synchronize()
        class_acc = (output.max(-1)[-1] == targets).float().mean() if mixup_fn is None else None
        metric_logger.update(loss=loss_value)
        metric_logger.update(class_acc=class_acc)
        metric_logger.update(loss_scale=loss_scale_value)
        min_lr, max_lr = 10.0, 0.0
        for group in optimizer.param_groups:
            min_lr = min(min_lr, group["lr"])
            max_lr = max(max_lr, group["lr"])
        metric_logger.update(lr=max_lr)
        metric_logger.update(min_lr=min_lr)
        weight_decay_value = None
        for group in optimizer.param_groups:
            if group["weight_decay"] > 0:
                weight_decay_value = group["weight_decay"]
        metric_logger.update(weight_decay=weight_decay_value)
        metric_logger.update(grad_norm=grad_norm)
        if log_writer is not None:
            log_writer.update(loss=loss_value, head="loss")
            log_writer.update(class_acc=class_acc, head="loss")
            log_writer.update(loss_scale=loss_scale_value, head="opt")
            log_writer.update(lr=max_lr, head="opt")
            log_writer.update(min_lr=min_lr, head="opt")
            log_writer.update(weight_decay=weight_decay_value, head="opt")
            log_writer.update(grad_norm=grad_norm, head="opt")
            log_writer.set_step()
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(data_loader, model, device):
    criterion = torch.nn.CrossEntropyLoss()
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = "Test:"
    model.eval()
    for batch in metric_logger.log_every(data_loader, 10, header):
        images = batch[0].to(device, non_blocking=True)
        target = batch[-1].to(device, non_blocking=True)
        output = model(images)
        loss = criterion(output.float(), target)
        n = min(5, len(output[0]))
        acc1, acc5 = accuracy(output, target, topk=(1, n))
        batch_size = images.shape[0]
        metric_logger.update(loss=loss.item())
        metric_logger.meters["acc1"].update(acc1.item(), n=batch_size)
        metric_logger.meters["acc5"].update(acc5.item(), n=batch_size)
    metric_logger.synchronize_between_processes()
    print("* Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f} loss {losses.global_avg:.3f}"
          .format(top1=metric_logger.acc1, top5=metric_logger.acc5, losses=metric_logger.loss))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
