import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0], "1", "0", "0"]
exec(open(os.path.join(os.path.dirname(__file__), "..", "stress_attn_win.py")).read().split("ok = True")[0])
for (B, H, win) in ((10, 2, (24, 40)), (9, 4, (26, 40)), (4, 3, (26, 20)), (11, 1, (31, 20)), (9, 4, (23, 40)), (11, 2, (19, 40))):
    T, D = win[0] * win[1] + 1, 64 * H
    ref = run(B, H, win, 103, 1)
    for rep in range(4):
        cur = run(B, H, win, 103, 1)
        msg = []
        for nm, x, y in (("out", ref[0], cur[0]), ("lse", ref[1], cur[1])):
            if not torch.equal(x, y): msg.append(nm + " differs")
        for name, lo in (("dQ", 0), ("dK", D), ("dV", 2 * D)):
            x, y = ref[2][:, lo:lo + D].float().view(B, T, H, 64), cur[2][:, lo:lo + D].float().view(B, T, H, 64)
            bad = ~torch.isfinite(y)
            diff = (x != y)
            if bad.any() or diff.any():
                w = (diff | bad).any(3).nonzero()
                msg.append(f"{name}: nan {int(bad.sum())} diff {int(diff.sum())} maxabs {float((torch.nan_to_num(x)-torch.nan_to_num(y)).abs().max()):.3g} at (b,t,h) {w[:6].tolist()} .. tokens {sorted(set(w[:,1].tolist()))[:12]}")
        print(B, H, win, "rep", rep, msg or "identical", flush=True)
