"""Run the attention kernels of one ViT-B layer (B = 256, 197 tokens, 12 heads) n times: the profiling target of
tools/attn_pmc.sh.  argv: [attn16 0|1] [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
assert _lib.lib.memhip_set_option(b"attn16", mode) == 0
B, T, H = 256, 197, 12
D = 64 * H
TP = ops.attn_tokens_padded(T)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.5).bfloat16()
idx, nrd = rel_pos_index((14, 14))
table = torch.randn(nrd, H, generator=g, device="cuda") * 0.3
out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
for _ in range(n):
    ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)
    ops.attn_delta(dout, out, B * T, H, delta)
    ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None)
torch.cuda.synchronize()
