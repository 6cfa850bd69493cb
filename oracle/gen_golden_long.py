"""Generates tests/golden/vit_long.npz from the REFERENCE model (build container only; the reference never
travels): a small ViT on a NON-SQUARE 256 x 320 canvas = 16 x 20 + 1 = 321 tokens, i.e. more than the 256
tokens one workgroup holds, so the product takes its streaming attention kernels (BASELINE configs[4] is
the same path at 30 x 40 + 1 = 1201 tokens).  Asserts first that the oracle restatement (oracle/vit_ref.py)
reproduces the reference bit for bit on this geometry -- logits, loss and every gradient, in fp32 and under
bf16 autocast -- then commits inputs-by-seed + reference outputs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _refimport as R                                # noqa: E402
from oracle import vit_ref as V                                   # noqa: E402
from oracle.gen_golden import vit_inputs                          # noqa: E402

LONG = dict(img_size=(256, 320), patch_size=(16, 16), in_chans=2, vocab_size=512, embed_dim=128, depth=2,
            num_heads=2, mlp_ratio=4, drop_path_rate=0.0, use_shared_rel_pos_bias=True,
            use_abs_pos_emb=False, init_values=0.1)
LONG_INPUTS = (3, 21, 120)                                        # vit_inputs(LONG, B, seed, nmask)


def main():
    assert R.install(), "/root/reference is not available"
    import modeling_pretrain as MPre
    torch.set_num_threads(1)
    torch.manual_seed(0); ref = MPre.pt_vit(**LONG)
    torch.manual_seed(0); ora = V.RefViT(**LONG)
    sd = ref.state_dict()
    assert list(sd.keys()) == list(ora.state_dict().keys())
    for k, v in ora.state_dict().items():
        assert torch.equal(sd[k], v), k
    w = V.fill_by_name(sd, seed=3)
    ref.load_state_dict(w); ora.load_state_dict(w)
    x, mask, labels = vit_inputs(LONG, *LONG_INPUTS)
    gold = {}
    for mode, dt in (("fp32", None), ("bf16", torch.bfloat16)):
        outs = []
        for m in (ref, ora):
            m.zero_grad()
            if dt is None:
                lo = m(x, mask); loss = torch.nn.CrossEntropyLoss()(lo, labels)
            else:
                with torch.autocast("cpu", dtype=dt):
                    lo = m(x, mask); loss = torch.nn.CrossEntropyLoss()(lo, labels)
            loss.backward()
            outs.append((lo.detach().float(), loss.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), mode
        for k in outs[0][2]:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), (mode, k)
        gold[f"{mode}__logits"] = outs[0][0].numpy()
        gold[f"{mode}__loss"] = outs[0][1].numpy()
        if mode == "bf16":                                        # the gradients the product is compared with
            for k, g in outs[0][2].items():
                gold[f"{mode}__grad__{k}"] = g.numpy()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vit_long.npz")
    np.savez_compressed(out, **gold)
    print("wrote", out, {k: v.shape for k, v in gold.items() if "grad" not in k})


if __name__ == "__main__":
    main()
