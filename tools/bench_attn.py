import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
from oracle.vit_ref import rel_pos_index
B, T, H = 256, 197, 12
D = 64 * H
TP = ops.attn_tokens_padded(T)
qkv = (torch.randn(B * T, 3 * D, device="cuda") * 0.5).bfloat16()
idx, nrd = rel_pos_index((14, 14))
table = torch.randn(nrd, H, device="cuda") * 0.3
out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
dout = torch.randn(B * T, D, device="cuda").bfloat16()
dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda"); dvb = torch.zeros(D, device="cuda")
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("fwd us", t(lambda: ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)))
print("delta us", t(lambda: ops.attn_delta(dout, out, B * T, H, delta)))
print("bwd us (dtable+bias)", t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, dvb)))
print("bwd us (dtable, no dv_bias: engine path)", t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None)))
print("bwd us (no dtable)", t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, None)))
