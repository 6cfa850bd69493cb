"""BASELINE configs[4] on ONE GPU: ViT-L/16 (D=1024, depth 24, 16 heads, layer scale 1e-5) on 480 x 640 2-bin
voxels = 1201 tokens, 600 masked patches per sample (block-wise masks), bf16, AdamW.  A step = masks +
forward + CE + backward + clip + AdamW on a batch resident in HBM.  FLOPs: 2635.5 GFLOP per sample fwd+bwd
(GEMMs only).  python tools/bench_vitl.py [B] [steps]"""
import contextlib, io, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd.masking_generator import MaskingGenerator
from mem_amd.modeling_pretrain import pt_vit
from mem_amd.optim_factory import FlatAdamW, get_parameter_groups
from mem_amd import _lib
for kv in filter(None, os.environ.get("MEMHIP_OPTS", "").split(",")):      # A/B: MEMHIP_OPTS=attn_win=0
    _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W = 480, 640
torch.manual_seed(0)
model = pt_vit(img_size=(H, W), patch_size=(16, 16), in_chans=2, vocab_size=8192, embed_dim=1024, depth=24, num_heads=16,
               mlp_ratio=4, drop_path_rate=0.1, use_shared_rel_pos_bias=True, use_abs_pos_emb=False, init_values=1e-5).cuda().train()
eng = model.engine
if os.environ.get("MEMHIP_NO_SIDE") == "1":                                  # per-kernel accounting: weight gradients on the launch stream
    eng.wgrad_side_stream = False
if os.environ.get("MEMHIP_DS_WS") in ("0", "1"):                             # A/B: the dS-storing attention backward (the default; 3 GB workspace at B = 64) / the recomputing one
    eng.attn_ds_workspace = os.environ["MEMHIP_DS_WS"] == "1"
if os.environ.get("MEMHIP_WGRAD_GROUP"):                                     # A/B: engine.wgrad_group = 0 / 1 / 2
    eng.wgrad_group = int(os.environ["MEMHIP_WGRAD_GROUP"])
with contextlib.redirect_stdout(io.StringIO()):
    groups = get_parameter_groups(model, 0.05, model.no_weight_decay())
opt = FlatAdamW(model, groups, lr=1e-4)
opt.max_norm = 30.0
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, 2, H, W), generator=g, device="cuda") * (torch.rand((B, 2, H, W), generator=g, device="cuda") < 0.3)
masker = MaskingGenerator((30, 40), 600, min_num_patches=16, seed=1)
pool = torch.randint(0, 8192, (B * 600,), device="cuda")
def step():
    m = torch.from_numpy(masker.batch_u8(B).reshape(B, -1).astype(bool)).cuda()
    la = model.forward_loss(x, m, pool[: int(m.sum())])
    model.backward()
    eng.grad_norm()
    opt.step()
    return la
for _ in range(2): la = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): la = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(json.dumps({"workload": "ViT-L/16 480x640 C=2, 1201 tokens, 600 masked, bf16, 1 GPU", "batch": B, "ms_per_step": round(dt * 1e3, 2),
                  "samples_per_sec": round(B / dt, 2), "model_flops_frac_of_peak": round(B / dt * 2635.5e9 / 2.5e15, 4),
                  "loss": round(float(la[0].item()), 4), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
