"""The raw fp16x2 tokenizer forward on the bench shape (256 x 224^2), a few repetitions: run under rocprofv3 --kernel-trace --stats for
the per-layer kernel times (tools/exp/r05_run24.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd.vae_model import DiscreteVAE, HipTokenizer
torch.manual_seed(3)
B, H, W = 256, 224, 224
vae = DiscreteVAE(input_H=H, input_W=W, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3,
                  hidden_dim=384, channels=3).cuda().eval()
im = torch.rand(B, 3, H, W, device="cuda")
raw = HipTokenizer(vae, max_batch=B, precision="fp16x2", certify=False)
for _ in range(4): raw.get_codebook_indices(im)
torch.cuda.synchronize()
