"""Time the streaming attention kernels at the ViT-L / 480x640 geometry (1201 tokens, 30x40 window, 16 heads), library = MEMHIP_LIB."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
from oracle.vit_ref import rel_pos_index
B, T, H, win = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1201, 16, (30, 40)
D = 64 * H; TP = ops.attn_tokens_padded(T)
g = torch.Generator(device="cuda").manual_seed(3)
qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.5).bfloat16()
idx, nrd = rel_pos_index(win)
table = torch.randn(nrd, H, generator=g, device="cuda") * 0.3
dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
def t(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse); ops.attn_delta(dout, out, B * T, H, delta)
print(os.environ.get("MEMHIP_LIB", "default"),
      f"fwd {t(lambda: ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)):.1f} us  "
      f"delta+bwd (dtable) {t(lambda: (ops.attn_delta(dout, out, B * T, H, delta), ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dqb, None))):.1f} us  "
      f"delta+bwd (no dtable) {t(lambda: (ops.attn_delta(dout, out, B * T, H, delta), ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, None, dqb, None))):.1f} us")
