import os, sys, json, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mem_amd.modeling_pretrain import pt_vit
from oracle.gen_golden import BASE, vit_inputs
from oracle.vit_ref import fill_by_name
GOLDEN = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
for C in (2, 3):
    g = np.load(os.path.join(GOLDEN, f"vit_base_c{C}.npz"))
    cfg = dict(BASE, in_chans=C)
    m = pt_vit(**cfg); m.load_state_dict(fill_by_name(m.state_dict(), seed=1)); m = m.cuda().eval()
    x, mask, labels = vit_inputs(cfg, 2, 77, 98)
    with torch.no_grad(): lo = m(x.cuda(), mask.cuda())
    samp = lo.float().cpu()[::7, ::97].numpy(); ref = g["bf16__logits_sample"]; r32 = g["fp32__logits_sample"] if "fp32__logits_sample" in g else None
    d = np.abs(samp - ref)
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-30))) - 7)
    print(f"C={C}: n={d.size} max|d|={d.max():.4f} mean|d|={d.mean():.5f} rel-L2={np.linalg.norm(samp-ref)/np.linalg.norm(ref):.5f} max|ref|={np.abs(ref).max():.2f} "
          f"max d/ulp={np.max(d/ulp):.2f} frac<=1ulp={np.mean(d<=ulp):.4f} frac exact={np.mean(d==0):.4f} frac<=2ulp={np.mean(d<=2*ulp):.4f}")
    if r32 is not None:
        d2 = np.abs(ref - r32); d3 = np.abs(samp - r32)
        print(f"      reference bf16 vs its own fp32: max {d2.max():.4f} mean {d2.mean():.5f}; ours vs fp32: max {d3.max():.4f} mean {d3.mean():.5f}")
