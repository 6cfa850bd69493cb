"""Pretraining loop -- mirror of /root/reference/mem/engine_for_pretraining.py
(train_one_epoch :108-287, evaluate :289-366; the plotting / wandb-image helpers :28-105 are out
of scope).

Arithmetic per step is the reference's (:123-162): schedule assignment, tokenizer labels,
masked-token cross-entropy, backward, clip, AdamW -- but the loss and its gradient are fused into
the HIP pipeline (``model.forward_loss`` / ``model.backward``), the data-parallel all-reduce is
overlapped with backward (parallel.GradReducer) and the four per-step host synchronisations of
the reference (:156,:165,:230,:233) are replaced by ONE read-back every ``print_freq`` steps: the
logged values are the same, they just reach the meters in batches.  Works for any device the
model supports; the shipped loop's CUDA-only crash sites (:163,:165) have no equivalent here.
"""
import math
import sys
from typing import Iterable

import torch

from . import utils


_TOK_STREAM = {}
_COPY_STREAM = {}
_BAD_SAMPLES = {}          # device -> i64 [1]: samples the augmentation chain flagged since the last check


def _note_status(device, status):
    """Per-sample status words of augment.BatchAugPipeline (empty after the filter / canvas beyond the bound / events
    outside the canvas -- the reference raises ValueError or IndexError in the DataLoader worker for those; here the
    sample became an all-zero image).  Accumulated on the device, read with the meters' deferred transfer."""
    key = str(device)
    if key not in _BAD_SAMPLES:
        _BAD_SAMPLES[key] = torch.zeros(1, dtype=torch.int64, device=device)
    _BAD_SAMPLES[key] += (status != 0).sum()


def check_bad_samples(device=None):
    """Raise if the augmentation chain flagged a sample since the last call (one host read-back)."""
    for key, t in _BAD_SAMPLES.items():
        if device is not None and key != str(device):
            continue
        n = int(t.item())
        if n:
            t.zero_()
            raise ValueError(f"{n} sample(s) were empty after the event filter, needed a canvas beyond the sensor bound, or had "
                             f"events outside their canvas (the reference raises ValueError / IndexError in its transform "
                             f"chain for these; here they would train as all-zero images)")


def _copy_stream(device):
    if device not in _COPY_STREAM:
        _COPY_STREAM[device] = torch.cuda.Stream(device=device)
    return _COPY_STREAM[device]


def _tokenizer_stream(device):
    if device not in _TOK_STREAM:
        _TOK_STREAM[device] = torch.cuda.Stream(device=device)
    return _TOK_STREAM[device]


def _prep_batch(batch, device, model, d_vae, MAE=False):
    """-> (samples, images, bool_masked_pos, labels, extra) ; extra = dict(rows, mask_u8, labels_event) for raw batches
    (empty for the reference-style tuple batches)."""
    if isinstance(batch, dict):
        # raw batch (datasets.RawEventDataset.collate): one upload of the events, then the whole transform chain of
        # build_transformNPY + ColorJitter on the GPU (augment.BatchAugPipeline); patches IS visual_tokens for
        # discrete_vae_type == "event" (datasets.py:49-51)
        # the events (246 MB per 256 samples of 30 000 events) go up on a COPY stream: the host runs a step ahead of the
        # GPU, so this transfer overlaps the previous step's kernels instead of queueing behind them on the launch stream
        main0 = torch.cuda.current_stream()
        cs = _copy_stream(device)
        with torch.cuda.stream(cs):
            ev = batch["events"].to(device, non_blocking=True)
            ev_up = torch.cuda.Event()
            ev_up.record(cs)
        main0.wait_event(ev_up)
        ev.record_stream(main0)
        samples, stages = batch["pipe"](ev, batch["offsets"], batch["draws"], return_stages=True)
        _note_status(ev.device, stages["status"])
        images = samples
        masks = batch["masks"]
        if MAE:
            return samples, images, torch.from_numpy(masks).to(device).flatten(1).to(torch.bool), None, {}
        # masked rows on the HOST (the masks come from the host): no device nonzero(), no synchronisation
        import numpy as np
        B = masks.shape[0]
        m2 = masks.reshape(B, -1)
        L = m2.shape[1]
        bi, pi = np.nonzero(m2)
        rows = torch.from_numpy((bi * (L + 1) + 1 + pi).astype(np.int32)).to(device, non_blocking=True)
        flat = torch.from_numpy((bi * L + pi).astype(np.int64)).to(device, non_blocking=True)
        mask_u8 = torch.from_numpy(m2.astype(np.uint8).reshape(-1)).to(device, non_blocking=True)
        bool_masked_pos = torch.from_numpy(m2.astype(bool)).to(device, non_blocking=True)
        # the frozen tokenizer (engine_for_pretraining.py:140-145) runs on its own stream BESIDE the ViT trunk: the labels
        # are only needed by the loss at the end of the forward pass (ViTEngine.forward waits on labels_event there)
        main = torch.cuda.current_stream()
        side = _tokenizer_stream(device)
        e0 = torch.cuda.Event()
        e0.record(main)
        with torch.no_grad(), torch.cuda.stream(side):
            side.wait_event(e0)
            input_ids = d_vae.get_codebook_indices(images).flatten(1)       # (B, 14*14)
            labels = input_ids.reshape(-1).index_select(0, flat)             # == input_ids[bool_masked_pos]
            ev_done = torch.cuda.Event()
            ev_done.record(side)
        # tensors allocated on one stream and used on the other: keep the caching allocator from recycling them early
        images.record_stream(side)
        flat.record_stream(side)
        labels.record_stream(main)
        in_chans = model.patch_embed.proj.weight.shape[1]
        if samples.shape[1] == 3 and in_chans == 2:
            samples = samples[:, 0::2].contiguous()
        return samples, images, bool_masked_pos, labels, dict(rows=rows, mask_u8=mask_u8, labels_event=ev_done)
    samples, images, bool_masked_pos = batch
    images = images.to(device, non_blocking=True)
    samples = samples.to(device, non_blocking=True)
    bool_masked_pos = bool_masked_pos.to(device, non_blocking=True)
    if MAE:                                          # engine_for_pretraining.py:141-142: no tokenizer, 3-channel images
        return samples, images, bool_masked_pos.flatten(1).to(torch.bool), None, {}
    with torch.no_grad():
        bool_masked_pos = bool_masked_pos.flatten(1).to(torch.bool)
        input_ids = d_vae.get_codebook_indices(images).flatten(1)       # (B, 14*14)
        labels = input_ids[bool_masked_pos]                              # (numMasked)
    in_chans = model.patch_embed.proj.weight.shape[1]
    if samples.shape[1] == 3 and in_chans == 2:
        samples = samples[:, 0::2].contiguous()      # the 2-bin voxel view (engine_for_finetuning.py:228)
    return samples, images, bool_masked_pos, labels, {}


def _flush(pending, metric_logger, log_writer, optimizer, run):
    """One device->host transfer for all steps since the last flush."""
    if not pending:
        return
    check_bad_samples()
    vals = torch.stack([torch.cat([la, gn]) for la, gn, _ in pending]).tolist()     # [[loss, acc, gnorm], ...]
    for (loss_value, mlm_acc, grad_norm), (_, _, meta) in zip(vals, pending):
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        metric_logger.update(mlm_acc=mlm_acc)
        metric_logger.update(loss=loss_value)
        metric_logger.update(grad_norm=grad_norm)
        if log_writer is not None:
            log_writer.update(mlm_acc=mlm_acc, head="loss")
            log_writer.update(loss=loss_value, head="loss")
            log_writer.update(loss_scale=meta["loss_scale"], head="opt")
            log_writer.update(lr=meta["max_lr"], head="opt")
            log_writer.update(min_lr=meta["min_lr"], head="opt")
            log_writer.update(weight_decay=meta["wd"], head="opt")
            log_writer.update(grad_norm=grad_norm, head="opt")
            log_writer.set_step()
    pending.clear()


def train_one_epoch(model: torch.nn.Module, d_vae: torch.nn.Module, data_loader: Iterable, optimizer,
                    device: torch.device, epoch: int, loss_scaler, max_norm: float = 0, log_writer=None,
                    lr_scheduler=None, start_steps=None, lr_schedule_values=None, wd_schedule_values=None,
                    run=None, args=None, plotting=False, MAE=False):
    model.train()
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter("lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    metric_logger.add_meter("min_lr", utils.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    print_freq = 10
    start_steps = start_steps or 0
    reducer = getattr(model, "_reducer", None)
    pending = []
    try:
        return _train_steps(model, d_vae, data_loader, optimizer, device, loss_scaler, max_norm, log_writer, lr_scheduler,
                            start_steps, lr_schedule_values, wd_schedule_values, run, MAE, metric_logger, header, print_freq,
                            reducer, pending)
    finally:
        if reducer is not None and hasattr(reducer, "release"):
            reducer.release()                       # no CU reservation outlives the epoch, whatever ended it


def _train_steps(model, d_vae, data_loader, optimizer, device, loss_scaler, max_norm, log_writer, lr_scheduler, start_steps,
                 lr_schedule_values, wd_schedule_values, run, MAE, metric_logger, header, print_freq, reducer, pending):
    for step, (batch, _) in enumerate(metric_logger.log_every(data_loader, print_freq, header)):
        it = start_steps + step
        if lr_schedule_values is not None or wd_schedule_values is not None:
            for param_group in optimizer.param_groups:
                if lr_schedule_values is not None:
                    param_group["lr"] = lr_schedule_values[it] * param_group["lr_scale"]
                if wd_schedule_values is not None and param_group["weight_decay"] > 0:
                    param_group["weight_decay"] = wd_schedule_values[it]
        samples, images, bool_masked_pos, labels, extra = _prep_batch(batch, device, model, d_vae, MAE)
        if MAE:
            loss_acc = model.forward_loss(samples)                  # loss, pred, mask = model(samples) (:149); mlm_acc = 0
        else:
            loss_acc = model.forward_loss(samples, bool_masked_pos, labels, **extra)
        model._fused_loss_pending = True
        # samples the transform chain flagged (empty / outside the canvas) since the last check: this step's loss reads NaN
        # and its update is skipped on the device; the deferred check below raises the reference's error (no host sync here)
        bad = _BAD_SAMPLES.get(str(loss_acc.device))
        if bad is not None:
            loss_acc[0:1].add_(torch.where(bad > 0, float("nan"), 0.0).to(loss_acc.dtype))
        grad_norm = loss_scaler(loss_acc, optimizer, clip_grad=max_norm, parameters=model.parameters(),
                                model=model, reducer=reducer, poison=bad)
        lrs = [g["lr"] for g in optimizer.param_groups]
        wds = [g["weight_decay"] for g in optimizer.param_groups if g["weight_decay"] > 0]
        pending.append((loss_acc.clone(), grad_norm.clone(),
                        dict(loss_scale=loss_scaler.state_dict()["scale"], max_lr=max(lrs), min_lr=min(lrs),
                             wd=wds[-1] if wds else None)))
        # host-side values go to the meters right away (the log line of this step shows them); the device
        # values (loss, accuracy, gradient norm) follow at the next flush -- one host sync per print_freq steps
        metric_logger.update(loss_scale=loss_scaler.state_dict()["scale"])
        metric_logger.update(lr=max(lrs))
        metric_logger.update(min_lr=min(lrs))
        metric_logger.update(weight_decay=wds[-1] if wds else None)
        if (step + 1) % print_freq == 0:
            _flush(pending, metric_logger, log_writer, optimizer, run)
        if lr_scheduler is not None:
            lr_scheduler.step_update(start_steps + step)
    _flush(pending, metric_logger, log_writer, optimizer, run)
    metric_logger.synchronize_between_processes()
    print("Averaged stats:", metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(data_loader, model, d_vae, device, args, plotting=False, MAE=False):
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = "Test:"
    model.eval()
    reducer = getattr(model, "_reducer", None)
    if reducer is not None and hasattr(reducer, "release"):
        reducer.release()
    for batch in metric_logger.log_every(data_loader, 10, header):
        samples, images, bool_masked_pos, labels, extra = _prep_batch(batch[0], device, model, d_vae, MAE)
        la = (model.forward_loss(samples) if MAE else model.forward_loss(samples, bool_masked_pos, labels, **extra)).tolist()
        metric_logger.update(loss=la[0])
        metric_logger.meters["mlm_acc"].update(la[1])
    check_bad_samples()
    metric_logger.synchronize_between_processes()
    print("* mlm_acc {mlm_acc.global_avg:.3f} loss {losses.global_avg:.3f}".format(
        mlm_acc=metric_logger.mlm_acc, losses=metric_logger.loss))
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
