"""`--mae 1` step at the benchmark batch (ViT-B MAE factory, 224^2, B = 256): forward + loss + backward + clip + AdamW on the
bf16 engine (and the fp32 parity engine with --fp32), against the pt_vit step's time per algorithmic FLOP.
FLOPs per sample (GEMMs only, 2MNK, fwd x 3): encoder 12 x 768-d on 99 tokens, decoder 8 x 512-d on 197 tokens."""
import contextlib, io, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd.modeling_mae import mae_vit_base_patch16_dec512d8b
from mem_amd.optim_factory import FlatAdamW, get_parameter_groups

from mem_amd import _lib
from mem_amd.utils import cap_host_threads
cap_host_threads(4)                      # the launch thread must not be throttled by the container's CPU quota
for kv in os.environ.get("OPTS", "").split(","):
    if kv: _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
B = int(os.environ.get("B", 256)); prec = "fp32" if "--fp32" in sys.argv else "bf16"
def lin(tok, d, hid): return tok * 2 * (d * 3 * d + d * d + 2 * d * hid)
def att(tok, d): return 4 * tok * tok * d
enc = 12 * (lin(99, 768, 3072) + att(99, 768)); dec = 8 * (lin(197, 512, 2048) + att(197, 512))
misc = 196 * 2 * 768 * 768 + 99 * 2 * 768 * 512 + 197 * 2 * 512 * 768
flop = 3.0 * (enc + dec + misc)
with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(0)
    m = mae_vit_base_patch16_dec512d8b(norm_pix_loss=0, LOSS_ONLY_MASKED_MAE=True, img_size=224, precision=prec).cuda().train()
    opt = FlatAdamW(m, get_parameter_groups(m, 0.05, m.no_weight_decay()), lr=1e-4)
opt.max_norm = 3.0
x = torch.rand(B, 3, 224, 224, device="cuda") * (torch.rand(B, 3, 224, 224, device="cuda") < 0.3)
def step():
    la = m.forward_loss(x); m.backward(); m.engine.grad_norm(); opt.step(); return la
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 3 if prec == "fp32" else 10
for _ in range(n): la = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
ref_ms_per_gflop = float(os.environ.get("PTVIT_MS", 37.8)) / 108.85      # pt_vit step: profiles/r03_final_bench.json (round 2: 38.9)
print(json.dumps({"mae_precision": prec, "batch": B, "ms_per_step": round(dt * 1e3, 2), "samples_per_sec": round(B / dt, 1),
                  "gflop_per_sample": round(flop / 1e9, 2), "tflops": round(B * flop / dt / 1e12, 1),
                  "ms_per_gflop_sample": round(dt * 1e3 / (flop / 1e9), 4), "pt_vit_ms_per_gflop_sample": round(ref_ms_per_gflop, 4),
                  "ratio_to_pt_vit_time_per_flop": round(dt * 1e3 / (flop / 1e9) / ref_ms_per_gflop, 3), "loss": float(la[0])}))
