"""A/B: the real training step with the tokenizer on the main stream vs on its own stream (same box, interleaved)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, contextlib, io
from mem_amd import datasets as D
from mem_amd.masking_generator import MaskingGenerator
from mem_amd.modeling_pretrain import pt_vit
from mem_amd.optim_factory import FlatAdamW, get_parameter_groups
from mem_amd.utils import HostStager
from mem_amd.vae_model import DiscreteVAE, HipTokenizer
B, NE, H, W, C = 256, 30000, 224, 224, 2
torch.manual_seed(0)
model = pt_vit(img_size=(H, W), patch_size=(16, 16), in_chans=C, vocab_size=8192, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
               drop_path_rate=0.1, use_shared_rel_pos_bias=True, use_abs_pos_emb=False, init_values=0.1).cuda().train()
eng = model.engine
with contextlib.redirect_stdout(io.StringIO()):
    opt = FlatAdamW(model, get_parameter_groups(model, 0.05, model.no_weight_decay()), lr=5e-4)
opt.max_norm = 30.0
g = np.random.default_rng(1)
ev = np.empty((B * NE, 4)); ev[:, 0] = g.integers(0, W, B * NE); ev[:, 1] = g.integers(0, H, B * NE)
ev[:, 2] = np.sort(g.integers(0, 300000, (B, NE)), axis=1).reshape(-1); ev[:, 3] = g.integers(0, 2, B * NE) * 2 - 1
ev_dev = torch.from_numpy(ev).cuda(); offsets = (torch.arange(B + 1, dtype=torch.int64) * NE).cuda()
pipe = D.EventBatchPipeline(H, W, out_chans=C, time_surface=False, train_augs=False)
masker = MaskingGenerator((14, 14), 98, min_num_patches=16, seed=1)
vae = DiscreteVAE(input_H=H, input_W=W, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3, hidden_dim=384, channels=3).cuda().eval()
img = torch.rand(B, 3, H, W, device="cuda")
st_rows, st_mask, st_flat = HostStager(B * 98 * 4, "cuda"), HostStager(B * 196, "cuda"), HostStager(B * 98 * 8, "cuda")
side = torch.cuda.Stream(); T = eng.T
def run(tok, overlap, n=12):
    def step():
        x = pipe(ev_dev, offsets)
        m = masker.batch_u8(B).reshape(B, -1); bi, pi = np.nonzero(m)
        rows = st_rows.put((bi * T + 1 + pi).astype(np.int32)); flat = st_flat.put((bi * 196 + pi).astype(np.int64)); mask_u8 = st_mask.put(m.reshape(-1))
        if overlap:
            e0 = torch.cuda.Event(); e0.record()
            with torch.cuda.stream(side):
                side.wait_event(e0)
                labels = tok.get_codebook_indices(img).reshape(-1).index_select(0, flat)
                e1 = torch.cuda.Event(); e1.record(side)
            labels.record_stream(torch.cuda.current_stream())
        else:
            labels = tok.get_codebook_indices(img).reshape(-1).index_select(0, flat); e1 = None
        model.forward_loss(x, None, labels, rows=rows, mask_u8=mask_u8, labels_event=e1); model.backward(); eng.grad_norm(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for prec in ("fp32", "fp16x2"):
    tok = HipTokenizer(vae, max_batch=B, precision=prec)
    for rep in range(2):
        print(prec, "sequential %.2f ms   own stream %.2f ms" % (run(tok, False), run(tok, True)), flush=True)
    del tok
