"""Pretraining model -- drop-in for /root/reference/mem/modeling_pretrain.py
(VisionTransformerForMaskedImageModeling :22-126, factory pt_vit :128-140).

Same constructor keywords, state-dict keys / shapes, init (same torch-RNG consumption order:
:61-88), ``forward(x, bool_masked_pos, return_all_tokens=False)``, ``no_weight_decay()``,
``get_num_layers()``, ``patch_embed.patch_size / patch_shape``.  Execution is the fused HIP
pipeline of ``vit_engine.ViTEngine`` (bf16 MFMA operands, fp32 accumulate / residual / softmax),
i.e. the reference's autocast policy; there is no eager fallback.

Two ways in:
  * ``model(x, bool_masked_pos)`` -> logits with autograd support (one custom Function around the
    whole network), for callers that own the loss;
  * ``model.forward_loss(x, bool_masked_pos, labels)`` + ``model.backward()``: the loss
    (nn.CrossEntropyLoss mean) and its gradient are fused into the pipeline -- what
    engine_for_pretraining uses.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from ._lib import require_gpu
from .modeling_finetune import Block, PatchEmbed, RelativePositionBias
from .vit_engine import ViTEngine

_REGISTRY = {}


def register_model(fn):
    _REGISTRY[fn.__name__] = fn
    try:                                     # also expose through timm's registry when timm exists
        from timm.models.registry import register_model as _timm_register
        return _timm_register(fn)
    except Exception:
        return fn


def create_model(name, pretrained=False, **kwargs):
    """timm.create_model stand-in (timm 0.4.12 drops kwargs whose value is None before calling
    the factory -- that is how drop_block_rate=None never reaches pt_vit,
    run_mem_pretraining.py:176-192)."""
    if name not in _REGISTRY:
        raise RuntimeError("Unknown model (%s)" % name)
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _REGISTRY[name](pretrained=pretrained, **kwargs)


def trunc_normal_(tensor, mean=0.0, std=1.0):
    nn.init.trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)     # modeling_pretrain.py:19-20


class _ViTFunction(torch.autograd.Function):
    """Autograd boundary around the fused pipeline: parameter gradients are written straight into
    the flat gradient buffer (each p.grad is a view of it)."""

    @staticmethod
    def forward(ctx, model, x, mask_u8, rows_idx, dp_masks, anchor):
        ctx.model = model
        logits = model.engine.forward(x, mask_u8, rows_idx, labels=None, dp_masks=dp_masks)
        return logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.model
        m.engine.attach_grads()
        dt = torch.float32 if getattr(m, "precision", "bf16") == "fp32" else torch.bfloat16
        m.engine.backward(dlogits.to(dt).contiguous())
        return None, None, None, None, None, None


class VisionTransformerForMaskedImageModeling(nn.Module):
    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, vocab_size=8192, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=None, init_values=None, attn_head_dim=None,
                 use_abs_pos_emb=True, use_rel_pos_bias=False, use_shared_rel_pos_bias=False, init_std=0.02,
                 **kwargs):
        super().__init__()
        assert qkv_bias and drop_rate == 0.0 and attn_drop_rate == 0.0 and not use_rel_pos_bias, \
            "fused path covers the pretraining configuration (qkv_bias, no dropout, shared rel-pos bias)"
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        if use_abs_pos_emb:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        else:
            self.pos_embed = None
        self.pos_drop = nn.Dropout(p=drop_rate)
        if use_shared_rel_pos_bias:
            self.rel_pos_bias = RelativePositionBias(window_size=self.patch_embed.patch_shape, num_heads=num_heads)
        else:
            self.rel_pos_bias = None
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                  init_values=init_values, window_size=None, attn_head_dim=attn_head_dim)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.init_std = init_std
        self.lm_head = nn.Linear(embed_dim, vocab_size)
        if self.pos_embed is not None:
            trunc_normal_(self.pos_embed, std=self.init_std)
        trunc_normal_(self.cls_token, std=self.init_std)
        trunc_normal_(self.mask_token, std=self.init_std)
        trunc_normal_(self.lm_head.weight, std=self.init_std)
        self.apply(self._init_weights)
        self.fix_init_weight()
        self._engine = None
        self._anchor = None

    def fix_init_weight(self):
        for layer_id, layer in enumerate(self.blocks):
            layer.attn.proj.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))
            layer.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=self.init_std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, nn.Conv2d):
            trunc_normal_(m.weight, std=self.init_std)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def get_num_layers(self):
        return len(self.blocks)

    # ------------------------------------------------------------------ fused execution
    @property
    def engine(self):
        if self._engine is None:
            require_gpu()
            if getattr(self, "precision", "bf16") == "fp32":     # --precision fp32: the parity mode (vit_engine_f32.py)
                from .vit_engine_f32 import ViTEngineF32
                self._engine = ViTEngineF32(self)
            else:
                self._engine = ViTEngine(self)
        return self._engine

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.weights_dirty = True
        return r

    def _prep(self, x, bool_masked_pos, return_all_tokens):
        eng = self.engine
        B = x.shape[0]
        L = eng.L
        m2 = bool_masked_pos.reshape(B, L)
        mask_u8 = m2.to(device=eng.dev, dtype=torch.uint8).contiguous().view(-1)
        if return_all_tokens:
            sel = torch.ones((B, L), dtype=torch.bool, device=eng.dev)
        else:
            sel = m2.to(eng.dev).bool()
        bi, pi = torch.nonzero(sel, as_tuple=True)          # row-major order == x[:,1:][mask] order
        rows = (bi * eng.T + 1 + pi).to(torch.int32).contiguous()
        x = x.to(device=eng.dev, dtype=torch.float32).contiguous()
        return x, mask_u8, rows

    def draw_drop_path(self, B):
        """Per-sample stochastic-depth keep masks for one step: f32 [2*depth, B] of 0/1 with
        P(1) = 1 - drop_prob of the block (timm drop_path: floor(keep_prob + U[0,1)))."""
        eng = self.engine
        probs = [float(b.drop_prob) for b in self.blocks for _ in range(2)]
        if max(probs) == 0.0:                         # decided on the host: no device read-back per step
            return None
        keep = getattr(self, "_dp_keep", None)
        if keep is None or keep[0] != probs:          # the static (1 - p) column lives on the device once
            keep = self._dp_keep = (probs, (1.0 - torch.tensor(probs, device=eng.dev)).view(-1, 1))
        if getattr(eng, "dp_skip", False):
            # work-skipping stochastic depth (ViTEngine.dp_skip): the engine sizes its launches by the number of kept samples,
            # so the masks are drawn on the HOST (torch's CPU generator; timm draws on the device: another stream, same law)
            # The stream is the model's OWN generator (seeded once from torch's seed of the moment, i.e. from the run seed +
            # rank that the entrypoint sets): sampler seeds, num_workers = 0 augmentation draws and mask draws on the global
            # CPU generator do not shift it, and the masked form below (the A/B) draws the same uniforms.
            return torch.floor((1.0 - torch.tensor(probs)).view(-1, 1) + self._dp_uniform(2 * eng.depth, B))
        return torch.floor(keep[1] + self._dp_uniform(2 * eng.depth, B, eng.dev)).contiguous()

    def _dp_uniform(self, rows, B, device=None):
        """U[0,1) draws of the model's own generator (utils.DropPathStream: explicit seed, checkpointed state, pinned upload)."""
        st = getattr(self, "_dp_stream", None)
        if st is None:
            from .utils import DropPathStream
            st = self._dp_stream = DropPathStream()
        return st.uniform(rows, B, device)


    def forward(self, x, bool_masked_pos, return_all_tokens=False, drop_path_masks=None):
        x, mask_u8, rows = self._prep(x, bool_masked_pos, return_all_tokens)
        if self.training and drop_path_masks is None:
            drop_path_masks = self.draw_drop_path(x.shape[0])
        if not self.training:
            drop_path_masks = None
        if torch.is_grad_enabled() and self.training:
            if self._anchor is None:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            out = _ViTFunction.apply(self, x, mask_u8, rows, drop_path_masks, self._anchor)
        else:
            out = self.engine.forward(x, mask_u8, rows, labels=None, dp_masks=drop_path_masks).clone()
        if return_all_tokens:
            return out.view(x.shape[0], self.engine.L, -1)
        return out

    def forward_loss(self, x, bool_masked_pos, labels, drop_path_masks=None, rows=None, mask_u8=None, labels_event=None):
        """Fused forward + mean cross-entropy (+ dlogits).  Returns a device tensor [2] =
        (loss, mlm_acc) without synchronising.  Call ``backward()`` next for the gradients.
        labels_event: HIP event that marks `labels` ready (tokenizer running on another stream beside the trunk)."""
        if rows is None:
            x, mask_u8, rows = self._prep(x, bool_masked_pos, False)
        if self.training and drop_path_masks is None:
            drop_path_masks = self.draw_drop_path(x.shape[0])
        if not self.training:
            drop_path_masks = None
        self.engine.forward(x, mask_u8, rows, labels=labels, dp_masks=drop_path_masks, labels_event=labels_event)
        return self.engine.loss_acc

    def backward(self):
        self.engine.attach_grads()
        self.engine.backward()


from .modeling_finetune import ft_vit as _ft_vit   # noqa: E402
ft_vit = register_model(_ft_vit)


@register_model
def pt_vit(pretrained=False, **kwargs):
    init_ckpt = kwargs.pop("init_ckpt", None)
    precision = kwargs.pop("precision", "bf16")
    assert precision in ("bf16", "fp32")
    model = VisionTransformerForMaskedImageModeling(qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                                    **kwargs)
    model.precision = precision      # "fp32": parity mode (fp32 operands, no bf16 rounding points; slow)
    model.default_cfg = {"url": "", "num_classes": 2, "input_size": (3, 128, 128), "pool_size": None, "crop_pct": 1,
                         "interpolation": "bicubic", "mean": (0.5, 0, 0.5), "std": (0.5, 0, 0.5)}
    if pretrained:
        checkpoint = torch.load(init_ckpt, map_location="cpu")
        model.load_state_dict(checkpoint["model"])
    return model
