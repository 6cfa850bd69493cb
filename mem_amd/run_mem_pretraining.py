"""Pretraining entrypoint -- mirror of /root/reference/mem/run_mem_pretraining.py
(get_args :32-170, get_model :173-223, main :226-440).

Same flags (incl. the ``pt_`` aliases and ``--config <key = value file>``; unknown keys are
ignored like parse_known_args does), plus the seven flags the reference reads but never declares
(--num_layers --transformer_depth --transformer_heads --transformer_mlp_ratio
--transformer_emb --num_tokens --voxel) and ``--model`` defaulting to ``pt_vit``.  Data-parallel
runs use one process per GPU (torchrun / RANK, WORLD_SIZE, LOCAL_RANK) with RCCL gradient
all-reduce overlapped with backward (parallel.GradReducer) instead of torch DDP.
    python -m mem_amd.run_mem_pretraining --expweek 2026-10 --data_path synthetic --input_H 224 --input_W 224 ...
"""
import argparse
import datetime
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

from . import utils
from .datasets import build_pretraining_dataset
from .engine_for_pretraining import evaluate, train_one_epoch
from .modeling_pretrain import create_model
from .optim_factory import create_optimizer
from .parallel import GradReducer
from .utils import NativeScalerWithGradNormCount as NativeScaler


def _config_file_args(argv):
    """configargparse stand-in: ``--config file`` with flat ``key = value`` lines becomes
    ``--key value`` pairs placed BEFORE the command line (so explicit flags win)."""
    if "--config" not in argv:
        return argv
    i = argv.index("--config")
    path = argv[i + 1]
    rest = argv[:i] + argv[i + 2:]
    extra = []
    for line in open(path):
        line = line.split("#", 1)[0].strip()
        if "=" not in line:
            continue
        k, v = (s.strip() for s in line.split("=", 1))
        if v != "":
            extra += ["--" + k, v]
    return extra + rest


def ops_mod():
    from . import ops
    return ops


def get_args(argv=None):
    p = argparse.ArgumentParser("Pretraining script", add_help=False, allow_abbrev=False)
    p.add_argument("--expweek", type=str, required=True)
    p.add_argument("--expname", default=None, type=str)
    p.add_argument("--batch_size", "--pt_batch_size", default=64, type=int)
    p.add_argument("--epochs", "--pt_epochs", default=300, type=int)
    p.add_argument("--save_ckpt_freq", "--pt_save_ckpt_freq", default=20, type=int)
    p.add_argument("--discrete_vae_weight_path", type=str)
    p.add_argument("--discrete_vae_type", type=str, default="event")
    p.add_argument("--tokenizer_impl", type=str, default="hip_fp16x2", choices=["hip", "hip_fp16x2", "hip_bf16", "torch"],
                   help="hip_fp16x2 (default): two-plane fp16 operands (22 significant bits), three fp16 MFMAs per product, fp32 "
                        "accumulation, CERTIFIED since round 5: a label is kept only where its top-2 logit gap exceeds a stated "
                        "multiple of the fp16x2 error bound; every sample with a token below that margin is recomputed by the fp32 "
                        "kernels on the device, so the labels equal the fp32 mode's wherever the measured margin holds -- calibrated per model, audited at run time (vae_model.HipTokenizer; "
                        "tests/test_tokenizer_gpu.py plants near-ties), ~2x faster than fp32; hip: the fp32 implicit-GEMM forward "
                        "(csrc/conv_f32.hip; fp32 operands and accumulation like the reference); hip_bf16: the bf16-operand kernels "
                        "of csrc/conv.hip (~6x faster, 1-3 %% of the labels differ at near ties); torch: the fp32 module on stock "
                        "PyTorch-ROCm convolutions")
    p.add_argument("--timesurface", type=int, default=0)
    p.add_argument("--hotpixfilter", type=int, default=1)
    p.add_argument("--hotpix_num_stds", type=float, default=10)
    p.add_argument("--logtrafo", type=int, default=0)
    p.add_argument("--gammatrafo", type=int, default=0)
    p.add_argument("--gamma", type=float, default=0.5)
    p.add_argument("--normalize_events", type=int, default=0)
    p.add_argument("--slice_max_evs", type=int, default=30000)
    p.add_argument("--max_random_shift_evs", type=int, default=15)
    p.add_argument("--rand_aug", type=int, default=1)
    p.add_argument("--model", default="pt_vit", type=str, metavar="MODEL")
    p.add_argument("--pretrained", default=0, type=int)
    p.add_argument("--rel_pos_bias", action="store_true")
    p.add_argument("--disable_rel_pos_bias", action="store_false", dest="rel_pos_bias")
    p.set_defaults(rel_pos_bias=True)
    p.add_argument("--abs_pos_emb", action="store_true")
    p.set_defaults(abs_pos_emb=False)
    p.add_argument("--layer_scale_init_value", default=0.1, type=float)
    p.add_argument("--masking", default="block", type=str)
    p.add_argument("--num_mask_patches", default=75, type=int)
    p.add_argument("--max_mask_patches_per_block", type=int, default=None)
    p.add_argument("--min_mask_patches_per_block", type=int, default=16)
    p.add_argument("--MAE", "--mae", default=0, type=int)
    p.add_argument("--input_size", default=224, type=int)
    p.add_argument("--input_H", default=128, type=int)
    p.add_argument("--input_W", default=128, type=int)
    p.add_argument("--input_H2", default=128, type=int)
    p.add_argument("--input_W2", default=128, type=int)
    p.add_argument("--drop_path", "--pt_dropout", type=float, default=0.1, metavar="PCT")
    p.add_argument("--disable_eval_during_pretraining", action="store_true", default=False)
    p.add_argument("--opt", default="adamw", type=str, metavar="OPTIMIZER")
    p.add_argument("--opt_eps", default=1e-8, type=float, metavar="EPSILON")
    p.add_argument("--opt_betas", default=[0.9, 0.999], type=float, nargs="+", metavar="BETA")
    p.add_argument("--clip_grad", "--pt_grad_clip", type=float, default=None, metavar="NORM")
    p.add_argument("--momentum", type=float, default=0.9, metavar="M")
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--weight_decay_end", type=float, default=None)
    p.add_argument("--lr", "--pt_lr", type=float, default=5e-4, metavar="LR")
    p.add_argument("--warmup_lr", type=float, default=1e-6, metavar="LR")
    p.add_argument("--min_lr", type=float, default=1e-5, metavar="LR")
    p.add_argument("--warmup_epochs", type=int, default=5, metavar="N")
    p.add_argument("--warmup_steps", "--pt_warmup_steps", type=int, default=-1, metavar="N")
    p.add_argument("--train_interpolation", type=str, default="bicubic")
    p.add_argument("--second_interpolation", type=str, default="lanczos")
    p.add_argument("--data_path", default="synthetic", type=str)
    p.add_argument("--data_set", default="npy", choices=["CIFAR", "IMNET", "image_folder", "npy", "dsec_semseg"], type=str)
    p.add_argument("--imagenet_default_mean_and_std", default=False, action="store_true")
    p.add_argument("--resize", action="store_true", default=False)
    p.add_argument("--color_jitter", "--pt_color_jitter", type=float, default=0.2, metavar="PCT")
    p.add_argument("--output_dir", default="")
    p.add_argument("--log_dir", default=None)
    p.add_argument("--dist_eval", action="store_true", default=False)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="")
    p.add_argument("--auto_resume", action="store_true")
    p.add_argument("--no_auto_resume", action="store_false", dest="auto_resume")
    p.set_defaults(auto_resume=True)
    p.add_argument("--start_epoch", default=0, type=int, metavar="N")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"],
                   help="bf16: the reference's autocast placement (bf16 GEMM operands, fp32 accumulate / residual / softmax); "
                        "fp32: parity mode, fp32 operands everywhere like the reference without autocast (slow; for loss-curve "
                        "parity against the reference's fp32 run)")
    p.add_argument("--num_workers", default=10, type=int)        # run_mem_pretraining.py:152
    p.add_argument("--synthetic_if_missing", default=0, type=int,
                   help="0 (default): a --data_path that does not exist fails like the reference's assert (only the literal "
                        "'synthetic' selects synthetic streams); 1: opt in to replacing a missing path (loudly) by seeded "
                        "synthetic event streams of the sensor geometry its name implies")
    p.add_argument("--canvas_max_H", default=0, type=int, help="bound of data-dependent canvases (0: 480)")
    p.add_argument("--canvas_max_W", default=0, type=int, help="bound of data-dependent canvases (0: 640)")
    p.add_argument("--pin_mem", action="store_true")
    p.add_argument("--no_pin_mem", action="store_false", dest="pin_mem")
    p.set_defaults(pin_mem=True)                                      # run_mem_pretraining.py:155-157
    p.add_argument("--world_size", default=1, type=int)
    p.add_argument("--local_rank", default=-1, type=int)
    p.add_argument("--dist_on_itp", action="store_true")
    p.add_argument("--dist_url", default="env://")
    p.add_argument("--wandb", type=int, default=0)
    p.add_argument("--wandb_group", default="pt")
    # read by get_model (run_mem_pretraining.py:184-191) but never declared by the reference
    p.add_argument("--num_layers", default=4, type=int)
    p.add_argument("--transformer_depth", default=12, type=int)
    p.add_argument("--transformer_heads", default=12, type=int)
    p.add_argument("--transformer_mlp_ratio", default=4, type=int)
    p.add_argument("--transformer_emb", default=768, type=int)
    p.add_argument("--num_tokens", default=8192, type=int)
    p.add_argument("--voxel", default=0, type=int)
    # synthetic stand-in for the dataset folder
    p.add_argument("--synthetic_samples", default=64, type=int)
    argv = _config_file_args(list(sys.argv[1:] if argv is None else argv))
    return p.parse_known_args(argv)[0]


def get_model(args):
    print(f"Creating model: {args.model}")
    return create_model(
        args.model, pretrained=False, drop_path_rate=args.drop_path, drop_block_rate=None,
        use_shared_rel_pos_bias=args.rel_pos_bias, use_abs_pos_emb=args.abs_pos_emb,
        init_values=args.layer_scale_init_value, in_chans=2 if args.voxel == 0 else args.voxel,
        img_size=(args.input_H, args.input_W), patch_size=(2 ** args.num_layers, 2 ** args.num_layers),
        embed_dim=args.transformer_emb, depth=args.transformer_depth, num_heads=args.transformer_heads,
        mlp_ratio=args.transformer_mlp_ratio, vocab_size=args.num_tokens, precision=args.precision)


def main(args):
    utils.init_distributed_mode(args)
    # the launch thread shares the container's CPU quota with num_workers loader processes: keep torch's intra-op pool small
    utils.cap_host_threads(4)
    print("Running", f"{args.expweek}_{args.expname}")
    print(args)
    if bool(args.pretrained):
        raise NotImplementedError("--pretrained 1 downloads timm ImageNet weights: no network here")
    device = torch.device(args.device)
    seed = args.seed + utils.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    input_size = (args.input_H, args.input_W)
    second_input_size = (args.input_H2, args.input_W2)
    model = get_model(args)
    if args.MAE:                                         # run_mem_pretraining.py:231-232,275-276
        from .modeling_mae import mae_vit_base_patch16_dec512d8b
        assert args.input_H == args.input_W, "the MAE variant patchifies square images (modeling_mae.py:169)"
        model = mae_vit_base_patch16_dec512d8b(norm_pix_loss=0, LOSS_ONLY_MASKED_MAE=True, img_size=args.input_H,
                                               precision=args.precision)
        print(f"Using MAE loss ({args.precision} engine: mem_amd/modeling_mae.py)")
    patch_size = model.patch_embed.patch_size
    print("Patch size = %s" % str(patch_size))
    args.window_size = (input_size[0] // patch_size[0], input_size[1] // patch_size[1])
    print("Window size = %s" % str(args.window_size))
    args.patch_size = patch_size
    dataset_train = build_pretraining_dataset(args)
    dataset_val = None if args.disable_eval_during_pretraining else build_pretraining_dataset(is_train=False, args=args)
    if args.discrete_vae_weight_path:
        d_vae = utils.create_d_vae(weight_path=args.discrete_vae_weight_path, d_vae_type=args.discrete_vae_type,
                                   device=device, image_size=second_input_size)
    else:
        from .vae_model import DiscreteVAE
        print("WARNING: no --discrete_vae_weight_path: using a randomly initialised tokenizer (synthetic labels)")
        d_vae = DiscreteVAE(input_H=args.input_H, input_W=args.input_W, num_layers=args.num_layers,
                            num_tokens=args.num_tokens, codebook_dim=32, hidden_dim=64, num_resnet_blocks=0).to(device)
    if args.tokenizer_impl in ("hip", "hip_fp16x2", "hip_bf16") and device.type == "cuda":
        from .vae_model import HipTokenizer
        try:
            d_vae = HipTokenizer(d_vae.eval(), max_batch=args.batch_size,
                                 precision={"hip": "fp32", "hip_fp16x2": "fp16x2", "hip_bf16": "bf16"}[args.tokenizer_impl])
        except AssertionError as e:                     # shapes the HIP convolutions do not provide
            print(f"tokenizer: staying on the torch module ({e})")
    num_tasks, global_rank = utils.get_world_size(), utils.get_rank()
    num_training_steps_per_epoch = len(dataset_train) // args.batch_size // num_tasks
    sampler_train = torch.utils.data.DistributedSampler(dataset_train, num_replicas=num_tasks, rank=global_rank, shuffle=True)
    sampler_val = None
    if dataset_val is not None:
        sampler_val = (torch.utils.data.DistributedSampler(dataset_val, num_replicas=num_tasks, rank=global_rank, shuffle=False)
                       if args.dist_eval else torch.utils.data.SequentialSampler(dataset_val))
    # datasets hand out raw events + draw records (CPU only: DataLoader workers are safe); collate builds one CSR batch
    # and train_one_epoch runs the whole transform chain on the GPU once per batch (augment.BatchAugPipeline)
    data_loader_train = torch.utils.data.DataLoader(dataset_train, sampler=sampler_train, batch_size=args.batch_size,
                                                    num_workers=args.num_workers, pin_memory=args.pin_mem, drop_last=True,
                                                    collate_fn=getattr(dataset_train, "collate", None))
    data_loader_val = None
    if dataset_val is not None:
        data_loader_val = torch.utils.data.DataLoader(dataset_val, sampler=sampler_val, batch_size=int(1.5 * args.batch_size),
                                                      num_workers=args.num_workers, pin_memory=args.pin_mem, drop_last=False,
                                                      collate_fn=getattr(dataset_val, "collate", None))
    model.to(device)
    model_without_ddp = model
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print("number of params:", n_parameters)
    total_batch_size = args.batch_size * utils.get_world_size()
    print("LR = %.8f" % args.lr)
    print("Batch size = %d" % total_batch_size)
    print("Number of training steps = %d" % num_training_steps_per_epoch)
    eng = model.engine                                   # packs parameters into the flat buffers
    # the model's drop-path generator is seeded from the run seed + rank HERE (not from whatever torch's seed is at the first
    # draw), and the numerics switches of the run travel with args and the checkpoints (utils.save_model: "numerics")
    if hasattr(model, "_dp_uniform"):
        from .utils import DropPathStream
        model._dp_stream = DropPathStream()
        model._dp_stream.seed(args.seed + utils.get_rank())
    args.numerics = {"precision": getattr(args, "precision", "bf16"),
                     "tokenizer_impl": args.tokenizer_impl if not isinstance(d_vae, torch.nn.Module) else "torch",
                     "tokenizer_certified": bool(getattr(d_vae, "certify", False)),
                     "gelu_dg": int(getattr(eng, "epi_gelu", None) == getattr(ops_mod(), "EPI_BIAS_GELU_DG", -1)),
                     "dp_skip": bool(getattr(eng, "dp_skip", False))}
    print("numerics:", args.numerics)
    if args.distributed:
        # (parallel.py: reserve_cus > 0 leaves CUs to RCCL's channel kernels while buckets are in flight; no measurement on
        # more than one GPU exists yet, so the default is 0 -- MEMHIP_RESERVE_CUS=16 tries it)
        model._reducer = GradReducer(eng.flat_g, eng.buckets, flat_p=eng.flat_p,
                                     reserve_cus=int(os.environ.get("MEMHIP_RESERVE_CUS", "0")),
                                     streams=lambda: [torch.cuda.current_stream(), eng._side])
        eng.grad_hook = model._reducer
        eng.weights_dirty = True
    optimizer = create_optimizer(args, model_without_ddp)
    loss_scaler = NativeScaler()
    print("Use step level LR & WD scheduler!")
    lr_schedule_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, num_training_steps_per_epoch,
                                                warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
    if args.weight_decay_end is None:
        args.weight_decay_end = args.weight_decay
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs,
                                                num_training_steps_per_epoch)
    print("Max WD = %.7f, Min WD = %.7f" % (max(wd_schedule_values), min(wd_schedule_values)))
    if args.output_dir:
        utils.auto_load_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer,
                              loss_scaler=loss_scaler)
    print(f"Start training for {args.epochs} epochs")
    start_time = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        if args.distributed:
            data_loader_train.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, d_vae, data_loader_train, optimizer, device, epoch, loss_scaler,
                                      args.clip_grad, log_writer=None,
                                      start_steps=epoch * num_training_steps_per_epoch,
                                      lr_schedule_values=lr_schedule_values, wd_schedule_values=wd_schedule_values,
                                      run=None, args=args, MAE=args.MAE)
        if args.output_dir and ((epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs):
            utils.save_model(args=args, model=model, model_without_ddp=model_without_ddp, optimizer=optimizer,
                             loss_scaler=loss_scaler, epoch=epoch)
        if data_loader_val is not None:
            test_stats = evaluate(data_loader_val, model, d_vae, device, args, MAE=args.MAE)
            print(f"test_stats: {test_stats}")
        log_stats = {**{f"train_{k}": v for k, v in train_stats.items()}, "epoch": epoch, "n_parameters": n_parameters}
        if args.output_dir and utils.is_main_process():
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(log_stats) + "\n")
    print("Training time {}".format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    utils.cleanup_distributed_mode()


if __name__ == "__main__":
    opts = get_args()
    if opts.output_dir:
        Path(opts.output_dir).mkdir(parents=True, exist_ok=True)
    main(opts)
