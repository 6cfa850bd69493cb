"""Per-section cycle totals of the fused attention backward (a -DATTN16_TIMING build: tools/build_variant.sh timing
-DATTN16_TIMING; run with MEMHIP_LIB=variants/timing.so)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index
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
ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)
ops.attn_delta(dout, out, B * T, H, delta)
names = ["top wait+barrier", "bound + zero", "S/dP mfma", "softmax+atomics", "dQ mfma", "barrier B", "exch write+barrier A",
         "consume", "end barrier", "epilogue/next-iter", "", ""]
buf = (ctypes.c_ulonglong * 32)()
f = _lib.lib.memhip_attn16_prof
f.argtypes = [ctypes.c_void_p, ctypes.c_int]
f(None, 1)
n = 5
for _ in range(n): ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)
torch.cuda.synchronize(); f(buf, 1)
fn = ["top wait+barrier", "stage issue + S mfma", "bias + max", "exp + sum", "PV", "stores / next"]
print("forward (cycles per workgroup-kernel)")
for w in (0, 1):
    print(f"  wave {4 * w}: total {sum(buf[w * 16 + 6 + i] for i in range(6)) / (n * 252):9.0f}  " + "  ".join(f"{fn[i]} {buf[w * 16 + 6 + i] / (n * 252):.0f}" for i in range(6)))
for dt, fo in ((dtable, None), (dtable, out), (None, None)):
    ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dt, dqb, None, out=fo)
    torch.cuda.synchronize(); f(None, 1)
    n = 5
    for _ in range(n):
        ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dt, dqb, None, out=fo)
    torch.cuda.synchronize(); f(buf, 1)
    wgs = 240
    print("dtable" if dt is not None else "no dtable", "fused delta" if fo is not None else "delta from the workspace", "(cycles per workgroup-kernel, wave 0 | wave 4)")
    for w in (0, 1):
        tot = sum(buf[w * 16 + i] for i in range(12)) / (n * wgs)
        print(f"  wave {4 * w}: total {tot:9.0f}  " + "  ".join(f"{names[i]} {buf[w * 16 + i] / (n * wgs):.0f}" for i in range(10)))
