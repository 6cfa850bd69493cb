"""Time of the certified tokenizer's fp32 recompute (HipTokenizer._exact._forward_dyn) by the number of flagged samples (the count is a
device word: every launch covers the capacity and returns behind the live rows).  usage: [MEMHIP_LIB=...] r06_recompute_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mem_amd.vae_model import DiscreteVAE, HipTokenizer
B = 256
torch.manual_seed(20251)
vae = DiscreteVAE(input_H=224, input_W=224, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3, hidden_dim=384, channels=3).cuda().eval()
img = torch.rand(B, 3, 224, 224, device="cuda")
tok = HipTokenizer(vae, max_batch=B, precision="fp16x2")
tok.get_codebook_indices(img)
ex = tok._exact
lst = torch.arange(0, B, dtype=torch.int32, device="cuda")
res = []
for n in (0, 1, 2, 4, 8, 16):
    cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
    f = lambda: ex._forward_dyn(img, tok.norm, lst, cnt, 0, tok.n_round)
    f(); f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    res.append(f"{n}: {a.elapsed_time(b) / 10 * 1e3:.0f} us")
    ids = ex.ids[: n * 196].clone() if n else None
print("flagged samples -> recompute time:", " | ".join(res), flush=True)
print("ids checksum (8 samples):", int(ex.ids[: 8 * 196].sum()))
