"""layerscale_grad: result against torch and time per launch, ViT-B and ViT-L shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
for N, K in ((768, 768), (768, 3072), (1024, 1024), (1024, 4096), (1023, 772)):
    g = torch.Generator(device="cuda").manual_seed(N + K)
    W = torch.randn(N, K, generator=g, device="cuda"); W16 = W.bfloat16()
    dW = torch.randn(N, K, generator=g, device="cuda"); b = torch.randn(N, generator=g, device="cuda"); db = torch.randn(N, generator=g, device="cuda")
    gamma = torch.randn(N, generator=g, device="cuda") * 0.1; gamma[3] = 0
    out = torch.empty(N, device="cuda")
    ops.layerscale_grad(W16, dW, b, db, gamma, N, K, out)
    ref = ((W16.float() * dW).sum(1) + b * db) / gamma; ref[3] = 0
    err = ((out - ref).abs() / (ref.abs() + 1e-3)).max().item()
    for _ in range(3): ops.layerscale_grad(W16, dW, b, db, gamma, N, K, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ops.layerscale_grad(W16, dW, b, db, gamma, N, K, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50 * 1e6
    print(f"N={N} K={K}: max rel err {err:.2e}  {dt:.1f} us  ({N * K * 6 / dt / 1e6:.2f} TB/s)")
    assert err < 5e-3
