"""Generates tests/golden/vit_c5.npz from the REFERENCE model (build container only; the reference never travels):
BASELINE configs[4] GEOMETRY -- ViT-L width (D = 1024, 16 heads of 64, mlp 4096, layer scale 1e-5) on a 480 x 640
2-bin voxel canvas = 30 x 40 + 1 = 1201 tokens with the 4664 x 16 relative-position table, 600 masked patches, B = 1 --
at depth 2 and a 1024-entry vocabulary so that the CPU run takes seconds and the fixture stays small.  Asserts first that
the oracle restatement reproduces the reference bit for bit (fp32 and bf16 autocast: logits, loss, every gradient), then
commits the reference's outputs: loss, a block of logits, every small gradient tensor whole (incl. the [4664, 16] table
gradient), the big matrices as every 64th row, and every tensor's L2 norm."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _refimport as R                                # noqa: E402
from oracle import vit_ref as V                                   # noqa: E402
from oracle.gen_golden import vit_inputs                          # noqa: E402

C5 = dict(img_size=(480, 640), patch_size=(16, 16), in_chans=2, vocab_size=1024, embed_dim=1024, depth=2,
          num_heads=16, mlp_ratio=4, drop_path_rate=0.0, use_shared_rel_pos_bias=True,
          use_abs_pos_emb=False, init_values=1e-5)
C5_INPUTS = (1, 55, 600)                                          # vit_inputs(C5, B, seed, nmask)
ROW_STRIDE = 64                                                   # big matrices: rows 0, 64, 128, ...
SMALL = 1 << 17                                                   # tensors up to this many elements are stored whole


def sample(t):
    """What the fixture keeps of a gradient tensor (the test applies the same function to the product's gradient)."""
    return t if t.numel() <= SMALL else t[::ROW_STRIDE]


def main():
    assert R.install(), "/root/reference is not available"
    import modeling_pretrain as MPre
    torch.set_num_threads(8)
    torch.manual_seed(0); ref = MPre.pt_vit(**C5)
    torch.manual_seed(0); ora = V.RefViT(**C5)
    sd = ref.state_dict()
    assert list(sd.keys()) == list(ora.state_dict().keys())
    for k, v in ora.state_dict().items():
        assert torch.equal(sd[k], v), k
    assert tuple(sd["rel_pos_bias.relative_position_bias_table"].shape) == (4664, 16)
    w = V.fill_by_name(sd, seed=9)
    ref.load_state_dict(w); ora.load_state_dict(w)
    x, mask, labels = vit_inputs(C5, *C5_INPUTS)
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
        # (8 threads: fp32 CPU GEMMs are bit-stable across thread counts here -- SURVEY section 8c -- and both models run
        # in this same process)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), mode
        for k in outs[0][2]:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), (mode, k)
        gold[f"{mode}__logits_head"] = outs[0][0][:96].numpy()
        gold[f"{mode}__loss"] = outs[0][1].numpy()
        for k, g in outs[0][2].items():
            gold[f"{mode}__gnorm__{k}"] = np.float64(g.double().norm().item())
            gold[f"{mode}__grad__{k}"] = sample(g).numpy()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vit_c5.npz")
    np.savez_compressed(out, **gold)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB", {k: v.shape for k, v in gold.items() if "logits" in k or "loss" in k})


if __name__ == "__main__":
    main()
