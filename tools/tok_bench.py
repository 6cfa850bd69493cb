import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd.vae_model import DiscreteVAE
B=256
vae = DiscreteVAE(input_H=224, input_W=224, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3, hidden_dim=384, channels=3).cuda().eval()
img = torch.rand(B, 3, 224, 224, device="cuda")
def t(f, n=3):
    f(); f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("fp32 nchw", t(lambda: vae.get_codebook_indices(img)))
with torch.autocast("cuda", dtype=torch.bfloat16):
    print("bf16 autocast nchw", t(lambda: vae.get_codebook_indices(img)))
vae2 = vae.to(memory_format=torch.channels_last); img2 = img.contiguous(memory_format=torch.channels_last)
with torch.autocast("cuda", dtype=torch.bfloat16):
    print("bf16 autocast nhwc", t(lambda: vae2.get_codebook_indices(img2)))
ids32 = vae.get_codebook_indices(img[:32])
with torch.autocast("cuda", dtype=torch.bfloat16):
    ids16 = vae.get_codebook_indices(img[:32])
print("agreement fp32 vs bf16 (random weights):", (ids32 == ids16).float().mean().item())
from mem_amd.vae_model import HipTokenizer
tok = HipTokenizer(vae, max_batch=B, precision=sys.argv[1] if len(sys.argv) > 1 else "fp32")
print("HIP tokenizer ms", t(lambda: tok.get_codebook_indices(img)))
ids = tok.get_codebook_indices(img[:32]); ref = vae.get_codebook_indices(img[:32])
print("agreement HIP vs fp32:", (ids == ref).float().mean().item())
