"""Plain-PyTorch CPU restatement of the MEM pretraining model + step.  TEST INFRASTRUCTURE.

Restates (own structure, same arithmetic and the same state-dict keys):
  /root/reference/mem/modeling_finetune.py:56-247  Mlp / Attention / Block /
                                                   PatchEmbed / RelativePositionBias
  /root/reference/mem/modeling_pretrain.py:22-140  VisionTransformerForMaskedImageModeling, pt_vit
  /root/reference/mem/optim_factory.py:56-133      parameter groups + AdamW(0.9, 0.95)
  /root/reference/mem/utils.py:395-412             cosine_scheduler
  /root/reference/mem/engine_for_pretraining.py:123-162  one optimisation step
Pinned against the imported reference by oracle/gen_golden.py (forward logits,
loss, every gradient and the post-step weights are compared there, fp32,
bit-for-bit on CPU) and by tests/golden/vit_*.npz.

This is also what bench.py times as ``cpu_baseline`` (kind "port").
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def rel_pos_index(window):
    """modeling_finetune.py:224-240: pairwise relative-position bucket ids for a
    (Wh, Ww) token grid plus a class token at position 0.  Buckets
    [0, (2Wh-1)(2Ww-1)) are (dy, dx) offsets; the last three are
    cls->token, token->cls and cls->cls."""
    Wh, Ww = window
    nrd = (2 * Wh - 1) * (2 * Ww - 1) + 3
    ys, xs = torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + (Wh - 1)
    dx = xs[:, None] - xs[None, :] + (Ww - 1)
    n = Wh * Ww + 1
    idx = torch.zeros((n, n), dtype=torch.int64)
    idx[1:, 1:] = dy * (2 * Ww - 1) + dx
    idx[0, :] = nrd - 3
    idx[:, 0] = nrd - 2
    idx[0, 0] = nrd - 1
    return idx, nrd


class _Attn(nn.Module):
    def __init__(self, dim, heads, window=None):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.window = window
        if window:                               # use_rel_pos_bias: the block's own table (modeling_finetune.py:96-120)
            idx, nrd = rel_pos_index(window)
            self.relative_position_bias_table = nn.Parameter(torch.zeros(nrd, heads))
            self.register_buffer("relative_position_index", idx)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, bias):
        # modeling_finetune.py:128-157 (shared rel-pos bias branch, dropouts p=0)
        B, N, C = x.shape
        b3 = torch.cat((self.q_bias, torch.zeros_like(self.v_bias, requires_grad=False), self.v_bias))
        qkv = F.linear(input=x, weight=self.qkv.weight, bias=b3)
        qkv = qkv.reshape(B, N, 3, self.heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * self.scale
        a = q @ k.transpose(-2, -1)
        if self.window:                          # modeling_finetune.py:140-146
            n = self.window[0] * self.window[1] + 1
            own = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
            a = a + own.permute(2, 0, 1).contiguous().unsqueeze(0)
        if bias is not None:
            a = a + bias
        a = a.softmax(dim=-1)
        x = (a @ v).transpose(1, 2).reshape(B, N, -1)
        return self.proj(x)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))   # modeling_finetune.py:66-71


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, drop_path, init_values, eps, window=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attn(dim, heads, window)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.drop_prob = float(drop_path)
        if init_values is not None and init_values > 0:
            self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
            self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        else:
            self.gamma_1 = self.gamma_2 = None

    def _dp(self, y, keep):
        """Stochastic depth (timm 0.4.12 drop_path, un-vendored): y / keep_prob *
        bernoulli mask per sample.  ``keep`` is the per-sample 0/1 mask supplied
        by the caller (None = identity) so that parity runs can feed the mask."""
        if keep is None or self.drop_prob == 0.0:
            return y
        kp = 1.0 - self.drop_prob
        return y.div(kp) * keep.view(-1, 1, 1).to(y.dtype)

    def forward(self, x, bias, keep1=None, keep2=None):
        # modeling_finetune.py:182-189
        a = self.attn(self.norm1(x), bias)
        if self.gamma_1 is None:
            x = x + self._dp(a, keep1)
            x = x + self._dp(self.mlp(self.norm2(x)), keep2)
        else:
            x = x + self._dp(self.gamma_1 * a, keep1)
            x = x + self._dp(self.gamma_2 * self.mlp(self.norm2(x)), keep2)
        return x


class _PatchEmbed(nn.Module):
    def __init__(self, img, patch, cin, dim):
        super().__init__()
        self.img_size, self.patch_size = img, patch
        self.patch_shape = (img[0] // patch[0], img[1] // patch[1])
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.proj = nn.Conv2d(cin, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        assert x.shape[2] == self.img_size[0] and x.shape[3] == self.img_size[1]
        return self.proj(x).flatten(2).transpose(1, 2)   # modeling_finetune.py:205-210


class _RelPos(nn.Module):
    def __init__(self, window, heads):
        super().__init__()
        idx, nrd = rel_pos_index(window)
        self.window = window
        self.relative_position_bias_table = nn.Parameter(torch.zeros(nrd, heads))
        self.register_buffer("relative_position_index", idx)

    def forward(self):
        n = self.window[0] * self.window[1] + 1
        t = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
        return t.permute(2, 0, 1).contiguous()       # modeling_finetune.py:242-247


class RefViT(nn.Module):
    """== VisionTransformerForMaskedImageModeling as built by pt_vit
    (modeling_pretrain.py:128-133: qkv_bias=True, LayerNorm eps=1e-6)."""

    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, vocab_size=8192,
                 embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, drop_path_rate=0.0,
                 init_values=None, use_abs_pos_emb=True, use_shared_rel_pos_bias=False,
                 init_std=0.02):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = (nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
                          if use_abs_pos_emb else None)
        self.rel_pos_bias = (_RelPos(self.patch_embed.patch_shape, num_heads)
                             if use_shared_rel_pos_bias else None)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, dpr[i], init_values, 1e-6)
                                     for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.lm_head = nn.Linear(embed_dim, vocab_size)
        self.init_std = init_std
        self._init()

    @staticmethod
    def _tn(t, std):
        nn.init.trunc_normal_(t, mean=0.0, std=std, a=-std, b=std)   # modeling_pretrain.py:19-20

    def _init(self):
        """modeling_pretrain.py:61-88 -- same RNG consumption order as the
        reference: pos_embed, cls, mask, lm_head.weight, then apply() in module
        traversal order, then the depth rescale of proj / fc2."""
        s = self.init_std
        if self.pos_embed is not None:
            self._tn(self.pos_embed, s)
        self._tn(self.cls_token, s)
        self._tn(self.mask_token, s)
        self._tn(self.lm_head.weight, s)
        def visit(m):                     # nn.Module.apply order: children first
            for c in m.children():
                visit(c)
            if isinstance(m, nn.Linear):
                self._tn(m.weight, s)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
            elif isinstance(m, nn.Conv2d):
                self._tn(m.weight, s)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        visit(self)
        with torch.no_grad():
            for i, blk in enumerate(self.blocks):
                blk.attn.proj.weight.div_(math.sqrt(2.0 * (i + 1)))
                blk.mlp.fc2.weight.div_(math.sqrt(2.0 * (i + 1)))

    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def forward_features(self, x, bool_masked_pos, keep=None):
        # modeling_pretrain.py:97-117
        x = self.patch_embed(x)
        B, L, _ = x.shape
        cls = self.cls_token.expand(B, -1, -1)
        mt = self.mask_token.expand(B, L, -1)
        w = bool_masked_pos.unsqueeze(-1).type_as(mt)
        x = x * (1 - w) + mt * w
        x = torch.cat((cls, x), dim=1)
        if self.pos_embed is not None:
            x = x + self.pos_embed
        bias = self.rel_pos_bias() if self.rel_pos_bias is not None else None
        for i, blk in enumerate(self.blocks):
            k1 = k2 = None
            if keep is not None:
                k1, k2 = keep[i]
            x = blk(x, bias, k1, k2)
        return self.norm(x)

    def forward(self, x, bool_masked_pos, return_all_tokens=False, keep=None):
        # modeling_pretrain.py:119-126
        x = self.forward_features(x, bool_masked_pos, keep)[:, 1:]
        if return_all_tokens:
            return self.lm_head(x)
        return self.lm_head(x[bool_masked_pos])


class RefFtViT(nn.Module):
    """== VisionTransformer as built by ft_vit (modeling_finetune.py:250-378: qkv_bias=True, LayerNorm eps=1e-6,
    dropouts 0): trunk + mean pooling over the patch tokens + fc_norm + head (or cls token + norm)."""

    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, drop_path_rate=0.0, init_values=None, use_abs_pos_emb=True,
                 use_rel_pos_bias=False, use_shared_rel_pos_bias=False, use_mean_pooling=True, init_scale=0.001):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = (nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
                          if use_abs_pos_emb else None)
        self.rel_pos_bias = (_RelPos(self.patch_embed.patch_shape, num_heads) if use_shared_rel_pos_bias else None)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, dpr[i], init_values, 1e-6,
                                            self.patch_embed.patch_shape if use_rel_pos_bias else None)
                                     for i in range(depth)])
        self.norm = nn.Identity() if use_mean_pooling else nn.LayerNorm(embed_dim, eps=1e-6)
        self.fc_norm = nn.LayerNorm(embed_dim, eps=1e-6) if use_mean_pooling else None
        self.head = nn.Linear(embed_dim, num_classes)
        def tn(t, std):          # timm 0.4.12 trunc_normal_(tensor, std=std): ABSOLUTE cut-offs a=-2, b=2 (un-vendored)
            nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)
        if self.pos_embed is not None:
            tn(self.pos_embed, 0.02)
        tn(self.cls_token, 0.02)
        tn(self.head.weight, 0.02)

        def visit(m):                     # nn.Module.apply order: children first (modeling_finetune.py:311-318)
            for c in m.children():
                visit(c)
            if isinstance(m, nn.Linear):
                tn(m.weight, 0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        visit(self)
        with torch.no_grad():
            for i, blk in enumerate(self.blocks):
                blk.attn.proj.weight.div_(math.sqrt(2.0 * (i + 1)))
                blk.mlp.fc2.weight.div_(math.sqrt(2.0 * (i + 1)))
            self.head.weight.mul_(init_scale)
            self.head.bias.mul_(init_scale)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def forward(self, x, keep=None):
        # modeling_finetune.py:334-356
        x = self.patch_embed(x)
        B = x.shape[0]
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        if self.pos_embed is not None:
            x = x + self.pos_embed
        bias = self.rel_pos_bias() if self.rel_pos_bias is not None else None
        for i, blk in enumerate(self.blocks):
            k1 = k2 = None
            if keep is not None:
                k1, k2 = keep[i]
            x = blk(x, bias, k1, k2)
        x = self.norm(x)
        x = self.fc_norm(x[:, 1:, :].mean(1)) if self.fc_norm is not None else x[:, 0]
        return self.head(x)


def layer_decay_groups(model, weight_decay, layer_decay):
    """optim_factory.py:31-100 + run_class_finetuning.py:550-552: group name -> (weight_decay, lr_scale, [names])."""
    depth = len(model.blocks)
    values = [layer_decay ** (depth + 1 - i) for i in range(depth + 2)]

    def layer_id(n):
        if n in ("cls_token", "mask_token", "pos_embed") or n.startswith("patch_embed"):
            return 0
        if n.startswith("rel_pos_bias"):
            return len(values) - 1
        if n.startswith("blocks"):
            return int(n.split(".")[1]) + 1
        return len(values) - 1
    skip = model.no_weight_decay()
    out = {}
    for n, p in model.named_parameters():
        nd = p.ndim == 1 or n.endswith(".bias") or n in skip
        lid = layer_id(n)
        g = "layer_%d_%s" % (lid, "no_decay" if nd else "decay")
        out.setdefault(g, {"weight_decay": 0.0 if nd else weight_decay, "lr_scale": values[lid], "params": []})["params"].append(n)
    return out


def param_groups(model, weight_decay=0.05):
    """optim_factory.py:56-95 with skip = model.no_weight_decay()."""
    skip = model.no_weight_decay()
    decay, no_decay, dn, nn_ = [], [], [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim == 1 or name.endswith(".bias") or name in skip:
            no_decay.append(p); nn_.append(name)
        else:
            decay.append(p); dn.append(name)
    # group order = first-seen order in named_parameters (cls_token first => no_decay first)
    groups = [{"weight_decay": 0.0, "params": no_decay, "lr_scale": 1.0},
              {"weight_decay": weight_decay, "params": decay, "lr_scale": 1.0}]
    return groups, {"no_decay": nn_, "decay": dn}


def make_optimizer(model, lr=5e-4, weight_decay=0.05, eps=1e-8):
    """optim_factory.py:98-133: AdamW, betas hard-set to (0.9, 0.95) (:121)."""
    groups, _ = param_groups(model, weight_decay)
    return torch.optim.AdamW(groups, lr=lr, weight_decay=0.0, eps=eps, betas=(0.9, 0.95))


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0,
                     start_warmup_value=0, warmup_steps=-1):
    """utils.py:395-412."""
    warm = np.array([])
    wi = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        wi = warmup_steps
    if warmup_epochs > 0:
        warm = np.linspace(start_warmup_value, base_value, wi)
    n = epochs * niter_per_ep - wi
    it = np.arange(n)
    sched = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / n))
                      for i in it])
    sched = np.concatenate((warm, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


def train_step(model, opt, samples, bool_masked_pos, labels, it, lr_sched=None, wd_sched=None,
               clip_grad=None, autocast_dtype=None, keep=None):
    """engine_for_pretraining.py:123-162 + utils.py:357-371 (scaler disabled: bf16 /
    fp32 need no loss scaling).  Returns (loss, grad_norm, mlm_acc)."""
    for g in opt.param_groups:
        if lr_sched is not None:
            g["lr"] = lr_sched[it] * g["lr_scale"]
        if wd_sched is not None and g["weight_decay"] > 0:
            g["weight_decay"] = wd_sched[it]
    kw = {} if keep is None else {"keep": keep}
    if autocast_dtype is not None:
        with torch.autocast("cpu", dtype=autocast_dtype):
            out = model(samples, bool_masked_pos, **kw)
            loss = nn.CrossEntropyLoss()(input=out, target=labels)
    else:
        out = model(samples, bool_masked_pos, **kw)
        loss = nn.CrossEntropyLoss()(input=out, target=labels)
    opt.zero_grad()
    loss.backward()
    params = [p for g in opt.param_groups for p in g["params"]]
    if clip_grad is not None:
        norm = torch.nn.utils.clip_grad_norm_(params, clip_grad)
    else:
        norm = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in params
                                       if p.grad is not None]), 2.0)
    opt.step()
    acc = (out.max(-1)[1] == labels).float().mean().item()
    return loss.item(), float(norm), acc


def fill_by_name(state_dict, seed=0, scale=None):
    """Deterministic, reference-independent weight recipe used for goldens: every
    tensor is filled from a hash of its NAME, so the same recipe reproduces the
    same weights in the reference (here), in this oracle and in the product on
    the GPU box, without any reference bytes travelling."""
    import zlib
    out = {}
    for name, t in state_dict.items():
        if not torch.is_floating_point(t):
            out[name] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        elif "gamma_" in name:
            v = 0.1 + 0.02 * torch.randn(t.shape, generator=g)
        elif t.ndim == 1 or name.endswith(".bias"):
            v = 0.02 * torch.randn(t.shape, generator=g)
        elif "relative_position_bias_table" in name:
            v = 0.2 * torch.randn(t.shape, generator=g)
        else:
            fan_in = t[0].numel() if t.ndim > 1 else t.numel()
            s = scale if scale is not None else min(0.05, 1.0 / math.sqrt(fan_in))
            v = s * torch.randn(t.shape, generator=g)
        out[name] = v.to(t.dtype)
    return out
