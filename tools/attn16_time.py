"""Time the 14x14 attention kernels (B = 256, 12 heads) of the library selected with MEMHIP_LIB."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index
B, H, T = 256, 12, 197
D = 64 * H; TP = ops.attn_tokens_padded(T)
g = torch.Generator(device="cuda").manual_seed(5)
qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.5).bfloat16()
idx, nrd = rel_pos_index((14, 14))
table = torch.randn(nrd, H, generator=g, device="cuda") * 0.3
dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse); ops.attn_delta(dout, out, B * T, H, delta)
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for rep in range(2):
    print(os.environ.get("MEMHIP_LIB", "default"),
          f"fwd {t(lambda: ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)):.1f} us  "
          f"bwd {t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None)):.1f} us  "
          f"bwd(no dtable) {t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, None, dqb, None)):.1f} us  "
          f"bwd(fused delta, as in the step) {t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None, out=out)):.1f} us")
