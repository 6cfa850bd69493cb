"""ViT building blocks -- mirror of /root/reference/mem/modeling_finetune.py:42-247
(DropPath, Mlp, Attention, Block, PatchEmbed, RelativePositionBias) -- and the finetuning model
``VisionTransformer`` / ``ft_vit`` (:250-385).

These classes keep the reference's constructor arguments, attribute names and therefore its
state-dict keys and its torch-RNG initialisation order, but they are PARAMETER CONTAINERS: the
arithmetic of a block runs in the fused HIP pipeline of ``vit_engine.ViTEngine`` (LayerNorm ->
MFMA GEMM with fused epilogues -> fused attention ...), driven by
``modeling_pretrain.VisionTransformerForMaskedImageModeling``.  Calling a container's forward on
its own is not part of the pretraining path and raises.
"""
import torch
import torch.nn as nn


def _fused(self, *a, **k):
    raise NotImplementedError(f"{type(self).__name__} is executed inside mem_amd's fused HIP ViT engine; "
                              "call the VisionTransformerForMaskedImageModeling model instead")


class DropPath(nn.Module):
    """modeling_finetune.py:42-53 (per-sample stochastic depth; applied in the GEMM epilogue)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    forward = _fused

    def extra_repr(self):
        return "p={}".format(self.drop_prob)


class Mlp(nn.Module):
    """modeling_finetune.py:56-71."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        assert act_layer is nn.GELU and drop == 0.0, "fused path: exact-erf GELU, dropout 0 (all reference configs)"
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    forward = _fused


class Attention(nn.Module):
    """modeling_finetune.py:74-157 (window_size=None: the shared rel-pos bias comes from the model)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0,
                 window_size=None, attn_head_dim=None):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if attn_head_dim is not None:
            head_dim = attn_head_dim
        all_head_dim = head_dim * num_heads
        assert head_dim == 64 and all_head_dim == dim, "fused attention kernel: head_dim 64 (ViT-B / ViT-L)"
        assert attn_drop == 0.0 and proj_drop == 0.0
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, all_head_dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(all_head_dim))
            self.v_bias = nn.Parameter(torch.zeros(all_head_dim))
        else:
            self.q_bias = None
            self.v_bias = None
        if window_size:
            # use_rel_pos_bias: this block's own bucket table + index buffer (modeling_finetune.py:96-124), same
            # parameter / buffer names; the fused attention kernel gathers from the table directly
            self.window_size = window_size
            rp = RelativePositionBias(window_size, num_heads)
            self.num_relative_distance = rp.num_relative_distance
            self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
            self.register_buffer("relative_position_index", rp.relative_position_index)
        else:
            self.window_size = None
            self.relative_position_bias_table = None
            self.relative_position_index = None
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(all_head_dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    forward = _fused


class Block(nn.Module):
    """modeling_finetune.py:160-189."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, init_values=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm, window_size=None,
                 attn_head_dim=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, window_size=window_size, attn_head_dim=attn_head_dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.drop_prob = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        if init_values is not None and init_values > 0:
            self.gamma_1 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
            self.gamma_2 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
        else:
            self.gamma_1, self.gamma_2 = None, None

    forward = _fused


class PatchEmbed(nn.Module):
    """modeling_finetune.py:192-210 (k = s = patch conv == im2col + MFMA GEMM in the engine)."""

    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, embed_dim=768):
        super().__init__()
        self.patch_shape = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.img_size = img_size
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    forward = _fused


class RelativePositionBias(nn.Module):
    """modeling_finetune.py:213-247: the bucket table + the index buffer (same values / dtype)."""

    def __init__(self, window_size, num_heads):
        super().__init__()
        self.window_size = window_size
        Wh, Ww = window_size
        self.num_relative_distance = (2 * Wh - 1) * (2 * Ww - 1) + 3
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
        ys, xs = torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        n = Wh * Ww + 1
        idx = torch.zeros((n, n), dtype=torch.int64)
        idx[1:, 1:] = (ys[:, None] - ys[None, :] + Wh - 1) * (2 * Ww - 1) + (xs[:, None] - xs[None, :] + Ww - 1)
        idx[0, 0:] = self.num_relative_distance - 3      # cls -> token
        idx[0:, 0] = self.num_relative_distance - 2      # token -> cls
        idx[0, 0] = self.num_relative_distance - 1       # cls -> cls
        self.register_buffer("relative_position_index", idx)

    forward = _fused


# ---------------------------------------------------------------------------------------------------------
# Finetuning model (modeling_finetune.py:250-385).  The trunk (patch embedding, position embedding, blocks with
# per-block or shared relative-position bias, layer scale, stochastic depth) runs in the fused HIP engine; the
# O(B x D) tail -- mean pooling over the patch tokens, fc_norm, head -- is ordinary torch modules under bf16
# autocast, connected to the engine by one autograd Function.
import math                                   # noqa: E402
from functools import partial                 # noqa: E402


def _trunc_normal_(tensor, mean=0.0, std=1.0):
    # modeling_finetune.py:17 imports timm's trunc_normal_ itself (absolute cut-offs a=-2, b=2), unlike
    # modeling_pretrain.py:19-20 which wraps it with a=-std, b=std
    nn.init.trunc_normal_(tensor, mean=mean, std=std, a=-2.0, b=2.0)


class _TrunkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, dp_masks, anchor):
        ctx.model = model
        eng = model.engine
        xl = eng.forward_trunk(x, None, dp_masks)
        return xl[: x.shape[0] * eng.T].view(x.shape[0], eng.T, eng.D).clone()

    @staticmethod
    def backward(ctx, dxl):
        eng = ctx.model.engine
        eng.attach_grads()
        eng.backward_trunk(dxl.float().contiguous())
        return None, None, None, None


class VisionTransformer(nn.Module):
    """modeling_finetune.py:250-369: same constructor keywords, state-dict keys, init order and forward()."""

    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=nn.LayerNorm, init_values=None, use_abs_pos_emb=True,
                 use_rel_pos_bias=False, use_shared_rel_pos_bias=False, use_mean_pooling=True, init_scale=0.001,
                 use_batch_norm=False, **kwargs):
        super().__init__()
        assert qkv_bias and drop_rate == 0.0 and attn_drop_rate == 0.0, "fused path: qkv_bias, no dropout"
        assert not use_batch_norm, "linear-probe BatchNorm head: not in the fused path"
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim)) if use_abs_pos_emb else None
        self.pos_drop = nn.Dropout(p=drop_rate)
        if use_shared_rel_pos_bias:
            self.rel_pos_bias = RelativePositionBias(window_size=self.patch_embed.patch_shape, num_heads=num_heads)
        else:
            self.rel_pos_bias = None
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.use_rel_pos_bias = use_rel_pos_bias
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                  init_values=init_values, window_size=self.patch_embed.patch_shape if use_rel_pos_bias else None)
            for i in range(depth)])
        self.norm = nn.Identity() if use_mean_pooling else norm_layer(embed_dim)
        self.fc_norm = norm_layer(embed_dim) if use_mean_pooling else None
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        if self.pos_embed is not None:
            _trunc_normal_(self.pos_embed, std=0.02)
        _trunc_normal_(self.cls_token, std=0.02)
        if isinstance(self.head, nn.Linear):
            _trunc_normal_(self.head.weight, std=0.02)
        self.apply(self._init_weights)
        self.fix_init_weight()
        if isinstance(self.head, nn.Linear):
            self.head.weight.data.mul_(init_scale)
            self.head.bias.data.mul_(init_scale)
        self._engine = None
        self._anchor = None

    def fix_init_weight(self):
        for layer_id, layer in enumerate(self.blocks):
            layer.attn.proj.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))
            layer.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token"}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=""):
        raise NotImplementedError("reset_classifier after the engine is built: construct the model with num_classes")

    @property
    def engine(self):
        if self._engine is None:
            from ._lib import require_gpu
            from .vit_engine import ViTEngine
            require_gpu()
            self._engine = ViTEngine(self)
        return self._engine

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.weights_dirty = True
        return r

    def draw_drop_path(self, B):
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


    def _trunk(self, x, drop_path_masks=None):
        eng = self.engine
        x = x.to(device=eng.dev, dtype=torch.float32).contiguous()
        if self.training and drop_path_masks is None:
            drop_path_masks = self.draw_drop_path(x.shape[0])
        if not self.training:
            drop_path_masks = None
        if torch.is_grad_enabled() and self.training:
            if self._anchor is None:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            return _TrunkFunction.apply(self, x, drop_path_masks, self._anchor)
        B = x.shape[0]
        return eng.forward_trunk(x, None, drop_path_masks)[: B * eng.T].view(B, eng.T, eng.D).clone()

    def forward_features(self, x, drop_path_masks=None):
        t = self._trunk(x, drop_path_masks)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            t = self.norm(t)
            if self.fc_norm is not None:
                return self.fc_norm(t[:, 1:, :].mean(1))
            return t[:, 0]

    def forward(self, x, drop_path_masks=None):
        x = self.forward_features(x, drop_path_masks)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return self.head(x)

    def get_intermediate_layers(self, x):
        raise NotImplementedError("per-block features are not exported by the fused engine")


def ft_vit(pretrained=False, **kwargs):
    """modeling_finetune.py:372-378."""
    model = VisionTransformer(qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = {"url": "", "num_classes": 2, "input_size": (3, 128, 128), "pool_size": None, "crop_pct": 1,
                         "interpolation": "bicubic", "mean": (0.5, 0, 0.5), "std": (0.5, 0, 0.5)}
    return model
