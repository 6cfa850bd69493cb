"""Finetuning loop -- drop-in for /root/reference/mem/engine_for_finetuning.py (train_class_batch :30-33,
train_one_epoch :41-212, evaluate :215-244), SURVEY section 8 row f3.

Same signatures and returned meter dicts.  Differences, all forced by the fused engine and stated here:
bf16 needs no loss scaling (``loss_scaler`` is utils.NativeScalerWithGradNormCount: backward -> fused clip ->
grouped AdamW); the deepspeed branch (``loss_scaler is None``) and the wandb image logging are not carried;
``update_freq`` > 1 accumulates the micro-batch gradients in the flat gradient buffer (``engine.accumulate_grads``;
``optimizer.zero_grad()`` clears it after the update); ``model_ema`` is updated through its own ``update(model)`` if
one is passed."""
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


def _apply_schedules(optimizer, it, lr_values, wd_values):
    """Per-step lr (x the group's layer-decay lr_scale) and weight decay (groups that decay only)."""
    for group in optimizer.param_groups:
        if lr_values is not None:
            group["lr"] = lr_values[it] * group.get("lr_scale", 1.0)
        if wd_values is not None and group["weight_decay"] > 0:
            group["weight_decay"] = wd_values[it]


def _optimizer_stats(optimizer):
    lrs = [g["lr"] for g in optimizer.param_groups]
    wds = [g["weight_decay"] for g in optimizer.param_groups if g["weight_decay"] > 0]
    return min([10.0] + lrs), max([0.0] + lrs), (wds[-1] if wds else None)


def train_one_epoch(args, model: torch.nn.Module, criterion: torch.nn.Module, data_loader: Iterable, optimizer,
                    device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, model_ema=None,
                    mixup_fn=None, log_writer=None, start_steps=None, lr_schedule_values=None,
                    wd_schedule_values=None, num_training_steps_per_epoch=None, update_freq=None):
    update_freq = update_freq or 1
    if loss_scaler is None:
        raise NotImplementedError("the deepspeed branch (loss_scaler=None) is not part of the fused path")
    first_it = start_steps or 0
    model.train(True)
    meters = utils.MetricLogger(delimiter="  ")
    for name in ("lr", "min_lr"):
        meters.add_meter(name, utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    eng = getattr(model, "engine", None)
    if eng is not None:
        # update_freq > 1 (engine_for_finetuning.py:78,117-124): micro-batch gradients ADD into the flat buffer; the
        # optimizer.zero_grad() below is what clears it
        eng.accumulate_grads = update_freq > 1
    optimizer.zero_grad()
    for data_iter_step, (samples, targets) in enumerate(meters.log_every(data_loader, 10, "Epoch: [{}]".format(epoch))):
        step = data_iter_step // update_freq
        if num_training_steps_per_epoch is not None and step >= num_training_steps_per_epoch:
            continue
        _apply_schedules(optimizer, first_it + step, lr_schedule_values, wd_schedule_values)
        samples, targets = samples.to(device, non_blocking=True), targets.to(device, non_blocking=True)
        if mixup_fn is not None:
            samples, targets = mixup_fn(samples, targets)
        loss, output = train_class_batch(model, samples, targets, criterion)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        # bf16: no loss scaling -- backward, fused clip, (grouped) AdamW; returns the pre-clip gradient norm
        do_update = (data_iter_step + 1) % update_freq == 0
        loss = loss / update_freq
        grad_norm = loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=False,
                                update_grad=do_update)
        if do_update:
            optimizer.zero_grad()
            if model_ema is not None:
                model_ema.update(model)
        torch.cuda.synchronize()
        min_lr, max_lr, wd_now = _optimizer_stats(optimizer)
        stats = {"loss": loss_value,
                 "class_acc": (output.max(-1)[-1] == targets).float().mean() if mixup_fn is None else None,
                 "loss_scale": loss_scaler.state_dict()["scale"], "lr": max_lr, "min_lr": min_lr, "weight_decay": wd_now,
                 "grad_norm": grad_norm}
        for k, v in stats.items():
            meters.update(**{k: v})
        if log_writer is not None:
            for k, v in stats.items():
                log_writer.update(head="loss" if k in ("loss", "class_acc") else "opt", **{k: v})
            log_writer.set_step()
    if eng is not None:
        eng.accumulate_grads = False
    meters.synchronize_between_processes()
    print("Averaged stats:", meters)
    return {k: meter.global_avg for k, meter in meters.meters.items()}


@torch.no_grad()
def evaluate(data_loader, model, device):
    """engine_for_finetuning.py:215-244: mean CE, top-1 and top-min(5, classes) accuracy in percent."""
    criterion = torch.nn.CrossEntropyLoss()
    meters = utils.MetricLogger(delimiter="  ")
    model.eval()
    for batch in meters.log_every(data_loader, 10, "Test:"):
        images, target = batch[0].to(device, non_blocking=True), batch[-1].to(device, non_blocking=True)
        output = model(images)
        acc1, acc5 = accuracy(output, target, topk=(1, min(5, len(output[0]))))
        meters.update(loss=criterion(output.float(), target).item())
        for name, val in (("acc1", acc1), ("acc5", acc5)):
            meters.meters[name].update(val.item(), n=images.shape[0])
    meters.synchronize_between_processes()
    print("* Acc@1 {top1.global_avg:.3f} Acc@5 {top5.global_avg:.3f} loss {losses.global_avg:.3f}"
          .format(top1=meters.acc1, top5=meters.acc5, losses=meters.loss))
    return {k: meter.global_avg for k, meter in meters.meters.items()}
