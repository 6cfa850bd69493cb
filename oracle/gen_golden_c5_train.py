"""tests/golden/vit_c5_train20.npz: 20 optimizer steps at the GEOMETRY of BASELINE configs[4] (ViT-L width, 480 x 640 -> 30 x 40 + 1 = 1201
tokens, 4664 x 16 relative-position table, depth 2, vocabulary 1024, B = 1, 600 masked patches) run by the REFERENCE in the build container
(TEST INFRASTRUCTURE; needs /root/reference).  Reference model + reference optimizer + the loop of engine_for_pretraining.py:123-162 restated
(oracle.vit_ref.train_step), clip 30, lr = cosine(5e-4 -> 1e-5, 4 warm-up steps), wd 0.05, two recurring batches; fp32 and bf16 autocast.  The oracle
restatement runs beside the reference for the first 3 fp32 steps and must equal it bit for bit.  The long-window attention kernels (attn_win.hip) are
the only attention path this geometry reaches: the product's bf16 engine is compared with these curves in tests/test_model_gpu.py.

    python -m oracle.gen_golden_c5_train        # ~3 min on 8 threads
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refimport as R                                # noqa: E402
from oracle import vit_ref as V                                   # noqa: E402
from oracle.gen_golden import vit_inputs                          # noqa: E402
from oracle.gen_golden_c5 import C5                               # noqa: E402

STEPS = 20


def batch(it):
    return vit_inputs(C5, 1, 3000 + it % 2, 600)


class OptArgs:
    opt = "adamw"; weight_decay = 0.05; lr = 5e-4; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9


def main():
    if not R.install():
        print("no /root/reference here: nothing generated")
        return
    import modeling_pretrain as MPre
    import optim_factory as OF
    torch.set_num_threads(8)
    with contextlib.redirect_stdout(io.StringIO()):
        lr_s = V.cosine_scheduler(5e-4, 1e-5, 1, STEPS, warmup_epochs=5, warmup_steps=4)
        wd_s = V.cosine_scheduler(0.05, 0.05, 1, STEPS)
    out = {"lr": lr_s, "wd": wd_s}
    t0 = time.time()
    for mode, dt in (("fp32", None), ("bf16", torch.bfloat16)):
        torch.manual_seed(0)
        ref = MPre.pt_vit(**C5)
        w = V.fill_by_name(ref.state_dict(), seed=9)
        ref.load_state_dict(w)
        with contextlib.redirect_stdout(io.StringIO()):
            ropt = OF.create_optimizer(OptArgs(), ref)
        ora = oopt = None
        if dt is None:
            ora = V.RefViT(**C5); ora.load_state_dict(w)
            oopt = V.make_optimizer(ora, lr=5e-4, weight_decay=0.05)
        rec = []
        for it in range(STEPS):
            x, m, l = batch(it)
            rec.append(V.train_step(ref, ropt, x, m, l, it, lr_s, wd_s, clip_grad=30.0, autocast_dtype=dt))
            if ora is not None and it < 3:
                assert V.train_step(ora, oopt, x, m, l, it, lr_s, wd_s, clip_grad=30.0) == rec[-1], (mode, it)
            print(f"{mode} step {it}: loss {rec[-1][0]:.6f} gnorm {rec[-1][1]:.4f}  ({time.time() - t0:.0f} s)", flush=True)
        out[f"{mode}__loss"] = np.array([r[0] for r in rec])
        out[f"{mode}__gnorm"] = np.array([r[1] for r in rec])
        if dt is None:
            out["fp32__final__table_rows"] = ref.rel_pos_bias.relative_position_bias_table.detach()[::64].numpy().copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "vit_c5_train20.npz"), **out)
    print("wrote vit_c5_train20.npz in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
