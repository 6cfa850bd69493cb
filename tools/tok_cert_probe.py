"""Certified fp16x2 tokenizer on the bench shape (ViT-B tokenizer, 256 x 224^2): measured logit deviation relative to the row
rms, flagged samples and time for several margins."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import datasets as D
from mem_amd.vae_model import DiscreteVAE, HipTokenizer
torch.manual_seed(3)
B, NE, H, W = 256, 30000, 224, 224
vae = DiscreteVAE(input_H=H, input_W=W, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3,
                  hidden_dim=384, channels=3).cuda().eval()
g = np.random.default_rng(1234)
ev = np.empty((B * NE, 4), dtype=np.float64)
ev[:, 0] = g.integers(0, W, B * NE); ev[:, 1] = g.integers(0, H, B * NE)
ev[:, 2] = np.sort(g.integers(0, 300000, (B, NE)), axis=1).reshape(-1); ev[:, 3] = g.integers(0, 2, B * NE) * 2 - 1
ev = torch.from_numpy(ev).cuda(); off = (torch.arange(B + 1, dtype=torch.int64) * NE).cuda()
sets = {"rasterised": D.rasterize(ev, off, H, W, True, strict=False).float() / 255.0, "uniform": torch.rand(B, 3, H, W, device="cuda")}
def tm(f, n=4):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t32 = HipTokenizer(vae, max_batch=B)
raw = HipTokenizer(vae, max_batch=B, precision="fp16x2", certify=False)
for name, im in sets.items():
    ids32 = t32.get_codebook_indices(im).clone(); l32 = t32.logits.clone()
    rms = l32.pow(2).mean(1, keepdim=True).sqrt()
    relgap = (t32.last_top2_gap(B).flatten() / rms.flatten())
    ids16 = raw.get_codebook_indices(im)
    dev = ((raw.logits - l32).abs() / rms).max().item()
    print(f"{name}: rms {rms.mean().item():.4f}  max deviation / rms {dev:.3e}  raw mismatches {int((ids16 != ids32).sum())}  "
          f"tokens with gap/rms < 1.4e-4: {int((relgap < 1.4e-4).sum())}, < 7e-5: {int((relgap < 7e-5).sum())}, < 3.5e-5: {int((relgap < 3.5e-5).sum())}", flush=True)
    del l32
print(f"raw fp16x2 {tm(lambda: raw.get_codebook_indices(sets['uniform'])):.2f} ms, fp32 {tm(lambda: t32.get_codebook_indices(sets['uniform']), 2):.2f} ms", flush=True)
del raw, t32
for kappa in (1.4e-4, 7e-5):
    for cap in (256, 64, 32):
        HipTokenizer.CERT_KAPPA = kappa
        c = HipTokenizer(vae, max_batch=B, precision="fp16x2", exact_capacity=cap)
        for name, im in sets.items():
            s0 = c.certification_stats()["flagged_samples"]
            t = tm(lambda: c.get_codebook_indices(im))
            s1 = c.certification_stats()
            print(f"kappa {kappa:.1e} capacity {cap}: {name}: {t:.2f} ms, flagged samples per call {(s1['flagged_samples'] - s0) / 5:.1f}", flush=True)
        del c
