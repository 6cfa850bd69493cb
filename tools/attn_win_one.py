"""One forward launch family of the window attention at the config-#5 size (for rocprofv3 counter passes): attn_win_one.py [mode] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
assert _lib.lib.memhip_set_option(b"attn_win", mode) == 0
B, H, win = 64, 16, (30, 40)
T, D = win[0] * win[1] + 1, 64 * H
TP = ops.attn_tokens_padded(T)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.7).bfloat16()
idx, nrd = rel_pos_index(win)
table = torch.randn(nrd, H, generator=g, device="cuda") * 0.5
out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
for _ in range(reps): ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
torch.cuda.synchronize()
