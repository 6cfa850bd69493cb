"""Import the reference (``/root/reference``) in the build container.

TEST INFRASTRUCTURE (golden generation only).  Holds no reference source: it
registers the stub modules the reference's imports need (timm / torchvision /
wandb / tensorboardX / torch._six are not installed here) and puts the reference
on ``sys.path``.  ``available()`` is False on the GPU box, where
``/root/reference`` does not exist; everything that uses this module must skip.
"""
import ast
import math
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "mem"))


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def _timm_drop_path(x, drop_prob=0.0, training=False):
    # timm==0.4.12 semantics (requirements.txt:4); un-vendored, parity unpinned.
    if drop_prob == 0.0 or not training:
        return x
    keep = 1 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    r = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
    r.floor_()
    return x.div(keep) * r


_done = False


def install():
    """Register stubs + sys.path once.  Returns False if no reference here."""
    global _done
    if not available():
        return False
    if _done:
        return True
    if not hasattr(np, "int"):
        np.int = int          # reference uses np.int / np.float (numpy<1.24)
    if not hasattr(np, "float"):
        np.float = float
    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", drop_path=_timm_drop_path,
         trunc_normal_=torch.nn.init.trunc_normal_,
         to_2tuple=lambda v: v if isinstance(v, tuple) else (v, v))
    _mod("timm.models.registry", register_model=lambda f: f)
    _mod("timm.utils", get_state_dict=lambda m: m.state_dict())
    _mod("timm.optim")
    for f, c in [("adafactor", "Adafactor"), ("adahessian", "Adahessian"), ("adamp", "AdamP"),
                 ("lookahead", "Lookahead"), ("nadam", "Nadam"), ("novograd", "NovoGrad"),
                 ("nvnovograd", "NvNovoGrad"), ("radam", "RAdam"), ("rmsprop_tf", "RMSpropTF"),
                 ("sgdp", "SGDP")]:
        _mod("timm.optim." + f, **{c: None})
    _mod("torch._six", inf=math.inf)
    _mod("tensorboardX", SummaryWriter=object)
    _mod("wandb", log=lambda *a, **k: None, Image=lambda *a, **k: None,
         Histogram=lambda *a, **k: None)
    _mod("torchvision")
    _mod("torchvision.utils", make_grid=None, save_image=None)

    class _IM:
        NEAREST = "nearest"; BILINEAR = "bilinear"; BICUBIC = "bicubic"
        LANCZOS = "lanczos"; HAMMING = "hamming"; BOX = "box"

    _mod("torchvision.transforms", InterpolationMode=_IM)
    _mod("torchvision.transforms.functional")
    for p in (REF + "/mem", REF, REF + "/eventvae"):
        if p not in sys.path:
            sys.path.insert(0, p)
    _done = True
    return True


def dataset_classes():
    """mem/datasets.py cannot be imported (torchvision.datasets, mmcv,
    timm.data): lift the event-level classes by AST and exec them."""
    install()
    wanted = {"EventArrToImg", "SliceRandomMaxEvs", "RandomTimeFlip", "Aug_FlipEvsAlongX",
              "Aug_RandomShiftEvs", "ReshapeScaleXandY"}
    ns = {"np": np, "random": random, "print": lambda *a, **k: None}
    src = open(REF + "/mem/datasets.py").read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name in wanted:
            exec(compile(ast.Module([node], []), "datasets.py", "exec"), ns)
    return ns
