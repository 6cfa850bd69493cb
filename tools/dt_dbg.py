import os, sys, torch
sys.path.insert(0, "/root/repo")
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index
B, T, H = 8, 197, 2
D = 64 * H; TP = ops.attn_tokens_padded(T)
g = torch.Generator(device="cuda").manual_seed(5)
qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.5).bfloat16()
idx, nrd = rel_pos_index((14, 14))
table = torch.randn(nrd, H, generator=g, device="cuda") * 0.3
dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
res = []
for mode in (0, 1):
    _lib.set_option("attn16", mode)
    out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
    dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
    delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
    ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse); ops.attn_delta(dout, out, B * T, H, delta)
    ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None)
    torch.cuda.synchronize(); res.append(dtable.clone())
d = (res[1] - res[0])[:, 0]
ref = res[0][:, 0]
print("rel", (d.norm() / ref.norm()).item(), "sum new", res[1][:, 0].sum().item(), "sum ref", ref.sum().item())
top = d.abs().topk(12).indices.tolist()
for i in top:
    if i < 729: print(i, "dy", i // 27 - 13, "dx", i % 27 - 13, "ref %.4f new %.4f" % (ref[i].item(), res[1][i, 0].item()))
    else: print(i, "special", "ref %.4f new %.4f" % (ref[i].item(), res[1][i, 0].item()))
