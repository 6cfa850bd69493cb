"""config-5 geometry fixture: eval logits of the product vs the reference's bf16 / fp32 runs, for attn_win = 0 and 1."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import _lib
from mem_amd.modeling_pretrain import pt_vit
from oracle.gen_golden import vit_inputs
from oracle.gen_golden_c5 import C5, C5_INPUTS
from oracle.vit_ref import fill_by_name
g = np.load("tests/golden/vit_c5.npz")
for mode in (0, 1):
    _lib.set_option("attn_win", mode)
    m = pt_vit(**C5)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=9))
    m = m.cuda().eval()
    x, mask, labels = vit_inputs(C5, *C5_INPUTS)
    with torch.no_grad():
        lo = m(x.cuda(), mask.cuda())[:96].float().cpu().numpy()
    for ref in ("bf16__logits_head", "fp32__logits_head"):
        d = np.abs(lo - g[ref])
        print(f"attn_win={mode} vs {ref}: max {d.max():.5f} mean {d.mean():.6f} p99.9 {np.quantile(d, 0.999):.5f} count>0.02 {(d > 0.02).sum()} of {d.size}")
d = np.abs(g["bf16__logits_head"] - g["fp32__logits_head"])
print(f"reference bf16 vs reference fp32: max {d.max():.5f} mean {d.mean():.6f}")
