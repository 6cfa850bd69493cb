"""fp16x2 tokenizer forward with 4 waves (one per SIMD) vs 8 waves (two per SIMD) per convolution workgroup: interleaved timing of
the raw forward at B = 256, 224^2, and bit-equality of the logits (same sum order).  usage: r06_conv_waves.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mem_amd import _lib
from mem_amd.vae_model import DiscreteVAE, HipTokenizer
B = 256
WAVES = [int(v) for v in os.environ.get('WAVES', '4,8').split(',')]
torch.manual_seed(20251)
vae = DiscreteVAE(input_H=224, input_W=224, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3, hidden_dim=384, channels=3).cuda().eval()
img = torch.rand(B, 3, 224, 224, device="cuda")
tok = HipTokenizer(vae, max_batch=B, precision="fp16x2", certify=False)
def t(n=5):
    tok.get_codebook_indices(img); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tok.get_codebook_indices(img)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
res = {}
for rep in range(3):
    for w in WAVES:
        _lib.set_option("conv_waves", w)
        ms = t()
        res[w] = tok.logits.clone()
        print(f"rep {rep} conv_waves {w}: {ms:.2f} ms per 256 samples = {B * 24.4e9 * 3 / (ms * 1e-3) / 1e15:.3f} PFLOP/s of fp16 MFMA work", flush=True)
print("logits bit-equal:", all(bool(torch.equal(res[WAVES[0]], res[w])) for w in WAVES))
