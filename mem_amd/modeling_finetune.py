"""ViT building blocks -- mirror of /root/reference/mem/modeling_finetune.py:42-247
(DropPath, Mlp, Attention, Block, PatchEmbed, RelativePositionBias).

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
        assert window_size is None, "per-block rel-pos tables (use_rel_pos_bias) are unused by the pretraining configs"
        assert attn_drop == 0.0 and proj_drop == 0.0
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, all_head_dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(all_head_dim))
            self.v_bias = nn.Parameter(torch.zeros(all_head_dim))
        else:
            self.q_bias = None
            self.v_bias = None
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
