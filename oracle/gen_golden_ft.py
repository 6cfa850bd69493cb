"""Generates tests/golden/vit_ft.npz + vit_ft_meta.json from the REFERENCE finetuning model (build container only;
the reference never travels): modeling_finetune.ft_vit in the default finetuning configuration of
run_class_finetuning.py (per-block relative-position bias, mean pooling + fc_norm, layer scale 0.1, in_chans 3) and
in the cls-token / shared-bias / abs-pos-embed variant, on a tiny geometry.  Asserts first that the oracle
restatement (oracle/vit_ref.py::RefFtViT) reproduces the reference bit for bit -- same-seed init, logits, loss,
every gradient under bf16 autocast -- and that oracle.layer_decay_groups equals the reference's
get_parameter_groups(LayerDecayValueAssigner) assignment."""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _refimport as R                                # noqa: E402
from oracle import vit_ref as V                                   # noqa: E402

FT_A = dict(img_size=(64, 96), patch_size=(16, 16), in_chans=3, num_classes=11, embed_dim=128, depth=3, num_heads=2,
            mlp_ratio=4, drop_path_rate=0.0, init_values=0.1, use_abs_pos_emb=False, use_rel_pos_bias=True,
            use_shared_rel_pos_bias=False, use_mean_pooling=True, init_scale=0.001)
FT_B = dict(FT_A, use_abs_pos_emb=True, use_rel_pos_bias=False, use_shared_rel_pos_bias=True, use_mean_pooling=False,
            init_values=None)
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def ft_inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    C, (H, W) = cfg["in_chans"], cfg["img_size"]
    x = torch.rand((B, C, H, W), generator=g) * (torch.rand((B, C, H, W), generator=g) < 0.3)
    y = torch.randint(0, cfg["num_classes"], (B,), generator=g)
    return x, y


def main():
    assert R.install(), "/root/reference is not available"
    import modeling_finetune as MF
    import optim_factory as OF
    torch.set_num_threads(1)
    gold, meta = {}, {}
    for tag, cfg in (("a", FT_A), ("b", FT_B)):
        torch.manual_seed(0); ref = MF.ft_vit(**cfg)
        torch.manual_seed(0); ora = V.RefFtViT(**cfg)
        sd_r, sd_o = ref.state_dict(), ora.state_dict()
        assert list(sd_r.keys()) == list(sd_o.keys()), (list(sd_r.keys())[:12], list(sd_o.keys())[:12])
        for k in sd_r:
            assert torch.equal(sd_r[k], sd_o[k]), k
        meta[f"{tag}_state_keys"] = list(sd_r.keys())
        w = V.fill_by_name(sd_r, seed=5)
        ref.load_state_dict(w); ora.load_state_dict(w)
        x, y = ft_inputs(cfg, 5, 31)
        outs = []
        for m in (ref, ora):
            m.zero_grad()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                lo = m(x)
                loss = torch.nn.CrossEntropyLoss()(lo, y)
            loss.backward()
            outs.append((lo.detach().float(), loss.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), tag
        for k in outs[0][2]:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), (tag, k)
        gold[f"{tag}__logits"] = outs[0][0].numpy()
        gold[f"{tag}__loss"] = outs[0][1].numpy()
        for k, g in outs[0][2].items():
            gold[f"{tag}__grad__{k}"] = g.numpy()
        with torch.no_grad():
            gold[f"{tag}__logits_fp32"] = ref(x).numpy()
        # layer-decay parameter groups (run_class_finetuning.py:549-552 + optim_factory.py:56-100)
        depth = cfg["depth"]
        assigner = OF.LayerDecayValueAssigner(list(0.75 ** (depth + 1 - i) for i in range(depth + 2)))
        name_of = {id(p): n for n, p in ref.named_parameters()}
        with contextlib.redirect_stdout(io.StringIO()):
            groups = OF.get_parameter_groups(ref, 0.05, ref.no_weight_decay(), assigner.get_layer_id, assigner.get_scale)
        rg = {}
        for g in groups:
            names = [name_of[id(p)] for p in g["params"]]
            rg[json.dumps(names)] = (g["weight_decay"], g["lr_scale"])
        og = V.layer_decay_groups(ora, 0.05, 0.75)
        assert {json.dumps(v["params"]): (v["weight_decay"], v["lr_scale"]) for v in og.values()} == rg, tag
        meta[f"{tag}_layer_decay_groups"] = og
    np.savez_compressed(os.path.join(OUT, "vit_ft.npz"), **gold)
    json.dump(meta, open(os.path.join(OUT, "vit_ft_meta.json"), "w"), indent=1)
    print("wrote vit_ft.npz / vit_ft_meta.json", {k: v.shape for k, v in gold.items() if "grad" not in k})


if __name__ == "__main__":
    main()
