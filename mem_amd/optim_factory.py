"""Optimizer factory -- mirror of /root/reference/mem/optim_factory.py:56-133.

``get_parameter_groups`` keeps the reference's decay / no_decay rule (ndim==1, ``.bias``, the
model's ``no_weight_decay()`` skip list) and ``create_optimizer`` keeps its AdamW with betas
hard-set to (0.9, 0.95) (:121).  Only ``--opt adamw`` -- the one optimizer every config uses -- is
implemented; the update itself is one streaming HIP kernel over the model's flat parameter /
gradient / moment buffers (csrc/optim.hip), with gradient clipping folded in.
"""
import json

import torch


def get_num_layer_for_vit(var_name, num_max_layer):
    """optim_factory.py:31-43: the layer id that decides a parameter's lr scale under layer-wise lr decay."""
    if var_name in ("cls_token", "mask_token", "pos_embed"):
        return 0
    elif var_name.startswith("patch_embed"):
        return 0
    elif var_name.startswith("rel_pos_bias"):
        return num_max_layer - 1
    elif var_name.startswith("blocks"):
        return int(var_name.split(".")[1]) + 1
    else:
        return num_max_layer - 1


class LayerDecayValueAssigner(object):
    """optim_factory.py:46-53."""

    def __init__(self, values):
        self.values = values

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        return get_num_layer_for_vit(var_name, len(self.values))


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None):
    """optim_factory.py:56-100: decay / no_decay, split per layer id (``layer_%d_decay`` ...) with an ``lr_scale``
    when the finetuning entry point passes a LayerDecayValueAssigner's methods."""
    names = {}
    groups = {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or name in skip_list:
            gname, wd = "no_decay", 0.0
        else:
            gname, wd = "decay", weight_decay
        if get_num_layer is not None:
            layer_id = get_num_layer(name)
            gname = "layer_%d_%s" % (layer_id, gname)
        else:
            layer_id = None
        if gname not in groups:
            scale = get_layer_scale(layer_id) if get_layer_scale is not None else 1.0
            names[gname] = {"weight_decay": wd, "params": [], "lr_scale": scale}
            groups[gname] = {"weight_decay": wd, "params": [], "lr_scale": scale}
        groups[gname]["params"].append(param)
        names[gname]["params"].append(name)
    print("Param groups = %s" % json.dumps(names, indent=2))
    return list(groups.values())


class FlatAdamW:
    """torch.optim.AdamW-shaped optimizer over the engine's flat buffers.

    ``param_groups`` (lr / weight_decay / lr_scale, as engine_for_pretraining.py:126-130 mutates
    them every step), ``zero_grad``, ``step``, ``state_dict`` / ``load_state_dict`` (torch's
    per-parameter layout, so reference checkpoints interchange)."""

    def __init__(self, model, param_groups, lr, betas=(0.9, 0.95), eps=1e-8):
        self.engine = model.engine
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0)
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("betas", tuple(betas))
            g.setdefault("eps", eps)
            self.param_groups.append(g)
        e = self.engine
        # chunk -> group map for the grouped update (layer-wise lr decay); built lazily on the first step that needs it
        self._group_of_chunk = None
        self._group_table = None
        self.exp_avg = torch.zeros(e.nflat, dtype=torch.float32, device=e.dev)
        self.exp_avg_sq = torch.zeros(e.nflat, dtype=torch.float32, device=e.dev)
        self.steps = 0
        self.is_second_order = False
        self.max_norm = 0.0           # set by the scaler (clip folded into the update kernel)

    def zero_grad(self, set_to_none=False):
        from . import ops
        if hasattr(self.engine, "wait_optimizer"):
            self.engine.wait_optimizer()          # a pipelined update may still be reading the gradients
        ops.zero_(self.engine.flat_g)

    def step(self):
        lrs = {g["lr"] for g in self.param_groups}
        wdset = {g["weight_decay"] for g in self.param_groups if g["weight_decay"] > 0}
        g0 = self.param_groups[0]
        self.steps += 1
        if len(lrs) == 1 and len(wdset) <= 1 and self._covers_flag_layout():
            wd = wdset.pop() if wdset else 0.0
            self.engine.adamw_step(self.exp_avg, self.exp_avg_sq, lrs.pop(), wd, self.steps, betas=g0["betas"],
                                   eps=g0["eps"], max_norm=self.max_norm)
        else:
            self._grouped_step(g0)

    def _covers_flag_layout(self):
        """True when the groups' decay / no_decay split is the engine's built-in one (pretraining)."""
        if not hasattr(self, "_flag_ok"):
            e = self.engine
            name_of = {id(p): n for n, p in e.named.items()}
            dec = {name_of[id(p)] for g in self.param_groups if g["weight_decay"] > 0 for p in g["params"]}
            allp = {name_of[id(p)] for g in self.param_groups for p in g["params"]}
            self._flag_ok = dec == set(e.decay_names) and allp == set(e.named)
        return self._flag_ok

    def _grouped_step(self, g0):
        """Per-group lr / weight decay (finetuning with layer decay, frozen parameters): memhip_adamw_groups.  Chunks
        that belong to no group (frozen parameters, padding) map to a group with lr 0 and no decay."""
        import numpy as np
        from . import ops
        e = self.engine
        ng = len(self.param_groups)
        assert ng < 255, "grouped AdamW: at most 254 parameter groups"
        if self._group_of_chunk is None:
            name_of = {id(p): n for n, p in e.named.items()}
            goc = np.full(e.nflat // 1024, ng, dtype=np.uint8)             # ng = the "frozen" group
            for gi, g in enumerate(self.param_groups):
                for p in g["params"]:
                    o, k = e.segs[name_of[id(p)]]
                    goc[o // 1024:(o + k + 1023) // 1024] = gi
            # (q_bias and v_bias share one padded [q_bias | 0 | v_bias] segment and one group: same layer, no decay)
            self._group_of_chunk = torch.from_numpy(goc).to(e.dev)
            self._group_table = torch.zeros((ng + 1, 2), dtype=torch.float32, device=e.dev)
        b1 = g0["betas"][0]
        bc1 = 1.0 - b1 ** self.steps
        tab = np.empty((ng + 1, 2), dtype=np.float32)
        for gi, g in enumerate(self.param_groups):
            tab[gi, 0] = np.float32(1.0 - g["lr"] * g["weight_decay"])
            tab[gi, 1] = np.float32(g["lr"] / bc1)
        tab[ng] = (1.0, 0.0)
        self._group_table.copy_(torch.from_numpy(tab), non_blocking=True)
        if hasattr(e, "wait_optimizer"):
            e.wait_optimizer()
        ops.adamw_groups(e.flat_p, e.flat_g, self.exp_avg, self.exp_avg_sq, e.nflat, self._group_of_chunk, self._group_table,
                         ng + 1, g0["betas"][0], g0["betas"][1], g0["eps"], self.steps, gnorm=e.gnorm,
                         max_norm=self.max_norm or 0.0)
        e.weights_dirty = True

    # ---- torch-format state for checkpoints (utils.save_model / auto_load_model)
    def _param_list(self):
        return [p for g in self.param_groups for p in g["params"]]

    def state_dict(self):
        e = self.engine
        if hasattr(e, "wait_optimizer"):
            e.wait_optimizer()                    # a pipelined update may still be writing the moments
        name_of = {id(p): n for n, p in e.named.items()}
        state, idx = {}, 0
        groups = []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                o, k = e.segs[name_of[id(p)]]
                if self.steps:
                    state[idx] = {"step": torch.tensor(float(self.steps)),
                                  "exp_avg": self.exp_avg[o:o + k].view(p.shape).clone(),
                                  "exp_avg_sq": self.exp_avg_sq[o:o + k].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        e = self.engine
        if hasattr(e, "wait_optimizer"):
            e.wait_optimizer()
        name_of = {id(p): n for n, p in e.named.items()}
        params = self._param_list()
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for k, v in sg.items():
                if k != "params":
                    g[k] = v
        for idx, st in sd["state"].items():
            p = params[int(idx)]
            o, k = e.segs[name_of[id(p)]]
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.steps = int(float(st["step"]))


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    opt_lower = args.opt.lower()
    if opt_lower.split("_")[-1] != "adamw":
        raise NotImplementedError(f"--opt {args.opt}: the pretraining path uses adamw (every reference config)")
    weight_decay = args.weight_decay
    if weight_decay and filter_bias_and_bn:
        skip = skip_list if skip_list is not None else (model.no_weight_decay() if hasattr(model, "no_weight_decay") else {})
        parameters = get_parameter_groups(model, weight_decay, skip, get_num_layer, get_layer_scale)
    else:
        parameters = [{"params": list(model.parameters()), "weight_decay": weight_decay, "lr_scale": 1.0}]
    eps = args.opt_eps if getattr(args, "opt_eps", None) is not None else 1e-8
    return FlatAdamW(model, parameters, lr=args.lr, betas=(0.9, 0.95), eps=eps)   # betas: optim_factory.py:121
