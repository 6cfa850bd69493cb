"""-m gpu: optimisation-step and loss-curve parity of the HIP path against the reference goldens
(tests/golden/vit_tiny_train100.npz: reference model + reference create_optimizer, 100 steps, and
vit_base_c3.npz cfg1: BASELINE config #1 = ViT-B, B=2, 10 steps).

Stated tolerances: the product trains in bf16 (GEMM operands) with fp32 accumulate/residual --
the reference's autocast policy.  Against the reference's bf16-autocast curve the per-step loss
differs by accumulation-order noise that compounds slowly over steps:
   |loss_hip - loss_ref_bf16| <= 0.02 over 100 steps, <= 5e-3 over the first 10;
   vs the fp32 reference curve <= 0.05;  grad-norm within 5 %.
(The north-star 1e-4 figure needs an fp32-operand mode that this round does not have: DESIGN.md.)
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _run(cfg, n_steps, batch_fn, lr, wd, clip, seed_w, overlap_optimizer=None):
    from mem_amd.modeling_pretrain import pt_vit
    from mem_amd.optim_factory import create_optimizer
    from mem_amd.utils import NativeScalerWithGradNormCount
    from oracle.vit_ref import fill_by_name
    import contextlib, io

    class A:
        opt = "adamw"; weight_decay = 0.05; lr = 5e-4; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9
    m = pt_vit(**cfg)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=seed_w))
    m = m.cuda().train()
    if overlap_optimizer is not None:
        m.engine.overlap_optimizer = overlap_optimizer
    with contextlib.redirect_stdout(io.StringIO()):
        opt = create_optimizer(A(), m)
    scaler = NativeScalerWithGradNormCount()
    rec = []
    for it in range(n_steps):
        for g in opt.param_groups:                       # engine_for_pretraining.py:124-130
            g["lr"] = lr[it] * g["lr_scale"]
            if g["weight_decay"] > 0:
                g["weight_decay"] = wd[it]
        x, mask, labels = batch_fn(it)
        la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
        m._fused_loss_pending = True
        gn = scaler(la, opt, clip_grad=clip, parameters=m.parameters(), model=m)
        rec.append((la[0].item(), gn.item(), la[1].item()))
    return m, opt, np.array(rec)


def test_tiny_100_steps_vs_reference_curves():
    from oracle.gen_golden import TINY, vit_inputs
    g = np.load(os.path.join(GOLDEN, "vit_tiny_train100.npz"))
    m, opt, rec = _run(TINY, 100, lambda it: vit_inputs(TINY, 4, 1000 + it % 8, 6), g["lr"], g["wd"], 30.0, 0)
    d16 = np.abs(rec[:, 0] - g["bf16__loss"])
    d32 = np.abs(rec[:, 0] - g["fp32__loss"])
    print("max |dloss| vs bf16 ref: first10 %.2e, all %.2e; vs fp32 ref %.2e; step-100 %.2e" %
          (d16[:10].max(), d16.max(), d32.max(), d16[-1]))
    assert d16[:10].max() <= 5e-3 and d16.max() <= 2e-2 and d32.max() <= 5e-2
    assert rec[-1, 0] < rec[0, 0] - 0.5                                    # it actually learns
    assert np.abs(rec[:, 1] / g["bf16__gnorm"] - 1).max() <= 0.05
    # checkpoint round trip of the optimizer state in torch's layout
    sd = opt.state_dict()
    assert len(sd["state"]) == len(list(m.parameters())) and sd["param_groups"][0]["betas"] == (0.9, 0.95)
    opt.load_state_dict(sd)


def test_pipelined_optimizer_equals_the_blocking_update():
    """ViTEngine.overlap_optimizer (round 4): AdamW / cast / transposes per layer bucket on their own stream with the next
    forward waiting per bucket -- the same kernels on sub-ranges of the flat buffers, so parameters, moments and the loss
    curve equal those of the one-launch update in front of the next forward up to the run-to-run noise of the fp32 gradient
    atomics (12 steps, gradient clipping active; a forward that read a bucket one step early would be off by the step size,
    ~1e-2 relative)."""
    from oracle.gen_golden import TINY, vit_inputs
    g = np.load(os.path.join(GOLDEN, "vit_tiny_train100.npz"))
    out = []
    for ov in (True, False):
        m, opt, rec = _run(TINY, 12, lambda it: vit_inputs(TINY, 4, 1000 + it % 8, 6), g["lr"], g["wd"], 0.5, 0, overlap_optimizer=ov)
        sd = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
        torch.cuda.synchronize()
        out.append((rec, sd, opt.exp_avg.cpu().clone(), opt.exp_avg_sq.cpu().clone()))
    # (two runs of the SAME mode differ by up to ~1e-5 relative after 12 steps -- fp32 gradient atomics, amplified by the
    # clipped updates; a bucket read one step early is off by lr / |w| ~ 2e-2)
    assert np.allclose(out[0][0], out[1][0], rtol=1e-3, atol=1e-4)
    for k in out[0][1]:
        a, b = out[0][1][k], out[1][1][k]
        assert ((a - b).norm() <= 2e-3 * b.norm() + 1e-6), (k, ((a - b).norm() / (b.norm() + 1e-12)).item())
    for a, b in ((out[0][2], out[1][2]), (out[0][3], out[1][3])):
        assert (a - b).norm() <= 1e-2 * b.norm()


def test_vit_base_config1_10_steps():
    """BASELINE config #1 arithmetic (ViT-B C=3, B=2, 10 steps, ncaltech.conf hyper-parameters) on the
    GPU against the reference's fp32 CPU curve."""
    from oracle.gen_golden import BASE, vit_inputs
    g = np.load(os.path.join(GOLDEN, "vit_base_c3.npz"))
    cfg = dict(BASE, in_chans=3)
    from oracle.vit_ref import cosine_scheduler
    wd10 = cosine_scheduler(0.05, 0.05, 1, 10)
    _, _, rec = _run(cfg, 10, lambda it: vit_inputs(cfg, 2, 500 + it, 98), g["cfg1__lr"], wd10, 30.0, 1)
    d = np.abs(rec[:, 0] - g["cfg1__loss"])
    print("config#1 max |dloss| vs fp32 reference: %.3e" % d.max(), rec[:, 0], g["cfg1__loss"])
    assert d.max() <= 3e-2
    assert np.abs(rec[:, 1] / g["cfg1__gnorm"] - 1).max() <= 0.08


def test_cli_entrypoint_two_epochs_synthetic(tmp_path):
    """The reference's entrypoint surface end to end (run_mem_pretraining.py:226-440): synthetic event
    streams -> augment/rasterize -> HIP tokenizer labels -> masked pretraining -> eval -> checkpoint + log."""
    import json
    from mem_amd.run_mem_pretraining import get_args, main
    out = tmp_path / "run"
    out.mkdir()
    args = get_args(["--expweek", "t", "--data_path", "synthetic", "--input_H", "112", "--input_W", "112",
                     "--batch_size", "8", "--epochs", "2", "--warmup_epochs", "0", "--synthetic_samples", "32",
                     "--num_workers", "0", "--output_dir", str(out), "--num_tokens", "512", "--color_jitter", "0",
                     "--rand_aug", "0", "--transformer_depth", "2", "--transformer_emb", "128",
                     "--transformer_heads", "2", "--num_mask_patches", "20", "--min_mask_patches_per_block", "4",
                     "--slice_max_evs", "5000"])
    main(args)
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert len(log) == 2 and all("train_loss" in e and "train_mlm_acc" in e and e["epoch"] in (0, 1) for e in log)
    assert all(e["train_loss"] == e["train_loss"] for e in log)                 # finite
    assert any(p.name.startswith("checkpoint-") for p in out.iterdir())


def test_pretrain_checkpoint_to_finetune_loop(tmp_path, capsys):
    """f3 end to end: a pretraining checkpoint ({'model': ...}, shared relative-position table, mask token, lm head)
    initialises the finetuning model through utils.finetune (table expanded to every block, head / mask token
    reported), then engine_for_finetuning.train_one_epoch with layer-wise lr decay fits a small synthetic
    classification set and evaluate() reports the reference's meters."""
    import contextlib
    import io
    from mem_amd import engine_for_finetuning as EF
    from mem_amd import optim_factory as OF
    from mem_amd import utils as U
    from mem_amd.modeling_finetune import ft_vit
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.vit_ref import fill_by_name
    geo = dict(img_size=(64, 96), patch_size=(16, 16), in_chans=3, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4)
    pre = pt_vit(vocab_size=512, drop_path_rate=0.0, use_shared_rel_pos_bias=True, use_abs_pos_emb=False, init_values=0.1,
                 **geo)
    sd = fill_by_name(pre.state_dict(), seed=2)
    ckpt = os.path.join(tmp_path, "checkpoint-9.pth")
    torch.save({"model": sd, "epoch": 9}, ckpt)
    torch.manual_seed(0)
    m = ft_vit(num_classes=4, drop_path_rate=0.1, init_values=0.1, use_abs_pos_emb=False, use_rel_pos_bias=True,
               use_mean_pooling=True, **geo)

    class Args:
        finetune = ckpt; model_key = "model|module"; model_prefix = ""
    U.finetune(Args(), m)
    out = capsys.readouterr().out
    assert "Expand the shared relative position embedding" in out
    assert "lm_head.weight" in out and "mask_token" in out and "head.weight" in out     # unexpected / missing report
    for i in range(3):
        assert torch.equal(m.state_dict()[f"blocks.{i}.attn.relative_position_bias_table"],
                           sd["rel_pos_bias.relative_position_bias_table"])
    assert torch.equal(m.state_dict()["patch_embed.proj.weight"], sd["patch_embed.proj.weight"])
    assert torch.equal(m.state_dict()["blocks.2.mlp.fc2.weight"], sd["blocks.2.mlp.fc2.weight"])
    m = m.cuda()
    # synthetic 4-class set: the class decides which quadrant of the canvas carries events
    g = torch.Generator().manual_seed(3)
    data = []
    for _ in range(6):
        y = torch.randint(0, 4, (16,), generator=g)
        x = torch.zeros(16, 3, 64, 96)
        for b in range(16):
            r, c = divmod(int(y[b]), 2)
            x[b, :, r * 32:(r + 1) * 32, c * 48:(c + 1) * 48] = (torch.rand(3, 32, 48, generator=g) < 0.3).float()
        data.append((x, y))
    depth = m.get_num_layers()
    assigner = OF.LayerDecayValueAssigner(list(0.75 ** (depth + 1 - i) for i in range(depth + 2)))

    class OA:
        opt = "adamw"; weight_decay = 0.05; lr = 2e-3; opt_eps = 1e-8
    with contextlib.redirect_stdout(io.StringIO()):
        opt = OF.create_optimizer(OA(), m, skip_list=m.no_weight_decay(), get_num_layer=assigner.get_layer_id,
                                  get_layer_scale=assigner.get_scale)
    assert len(opt.param_groups) > 2 and len({gr["lr_scale"] for gr in opt.param_groups}) == depth + 2
    scaler = U.NativeScalerWithGradNormCount()
    lr_sched = U.cosine_scheduler(2e-3, 1e-5, 8, len(data), warmup_epochs=1)
    crit = torch.nn.CrossEntropyLoss()
    first = last = None
    with contextlib.redirect_stdout(io.StringIO()):
        for ep in range(8):
            st = EF.train_one_epoch(None, m, crit, data, opt, torch.device("cuda"), ep, scaler, max_norm=5.0,
                                    start_steps=ep * len(data), lr_schedule_values=lr_sched,
                                    num_training_steps_per_epoch=len(data), update_freq=1)
            first = st if first is None else first
            last = st
        ev = EF.evaluate(data, m, torch.device("cuda"))
    assert {"loss", "class_acc", "lr", "min_lr", "grad_norm", "weight_decay", "loss_scale"} <= set(last)
    assert last["min_lr"] < last["lr"]                                  # layer decay: different lrs per group
    assert last["loss"] < 0.5 * first["loss"], (first["loss"], last["loss"])
    assert set(ev) == {"loss", "acc1", "acc5"} and ev["acc1"] >= 90.0, ev


def test_bench_contract_and_rccl_dry_run():
    """bench.py prints ONE JSON line and it is the LAST line of stdout, also when RCCL is initialised (its version
    banner goes through C stdio).  MEMHIP_BENCH_FORCE_DIST=1 runs the N > 1 code path -- process group, parameter
    broadcast, per-bucket async all-reduce hooked into backward, join before AdamW -- in a one-rank RCCL group."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MEMHIP_BENCH_FORCE_DIST="1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "16", "--steps", "3", "--warmup", "1",
                        "--no-tokenizer-figure", "--no-raster-figure", "--no-cpu-baseline", "--no-entrypoint-figure",
                        "--no-config4-figure", "--bucket-dtype", "bf16", "--reserve-cus", "16"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    out = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["scaling"] == "weak" and out["dtype"] == "bf16"
    assert out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1
    assert sum(ln.lstrip().startswith("{") for ln in lines) == 1
    # the N > 1 evidence block: what RCCL reports, who took part, bytes on the wire, exposed exchange time
    rc = out["rccl"]
    assert rc["world_size"] == 1 and rc["backend"] == "nccl" and len(rc["ranks"]) == 1 and rc["ranks"][0]["rank"] == 0
    assert rc["bucket_dtype"] == "bf16" and rc["reserve_cus"] == 16 and rc["buckets_per_step"] == 14
    assert 180e6 < rc["bytes_per_step"] < 190e6                      # 91.8 M parameters (padded) x 2 bytes
    assert isinstance(rc["allreduce_exposed_ms"], float)
    from mem_amd import _lib
    assert _lib.get_option("reserve_cus") == 0                      # (this process: the option is per process, default off)
