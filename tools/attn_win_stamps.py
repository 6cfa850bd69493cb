"""Section cycles of the forward window-attention kernel (stamp build: MEMHIP_LIB=variants/winstamp.so)."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index
B, H, win = 64, 16, (30, 40)
T, D = win[0] * win[1] + 1, 64 * H
TP = ops.attn_tokens_padded(T)
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.7).bfloat16()
idx, nrd = rel_pos_index(win)
table = torch.randn(nrd, H, generator=g, device="cuda") * 0.5
out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
for _ in range(3): ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
torch.cuda.synchronize()
a = np.zeros(1024 * 8, dtype=np.uint64)
assert _lib.lib.memhip_debug_win_stamps(a.ctypes.data_as(C.c_void_p)) == 0
t = a.reshape(1024, 2, 4).astype(np.float64)
for w in (0, 1):
    n = t[:, w, 3]; ok = n > 0
    print(f"wave {4*w}: per chunk: wait+barrier+stage {np.median(t[ok, w, 0] / n[ok]):.0f}  phase A {np.median(t[ok, w, 1] / n[ok]):.0f}  "
          f"phase B {np.median(t[ok, w, 2] / n[ok]):.0f}  (chunks per workgroup {np.median(n[ok]):.0f}, workgroups {ok.sum()})")
