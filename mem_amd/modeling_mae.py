"""MAE variant of the pretraining model (`--mae 1`) -- mirror of /root/reference/mem/modeling_mae.py
(MaskedAutoencoderViT :101-302, factory mae_vit_base_patch16_dec512d8b :304-313; selected at
mem/run_mem_pretraining.py:231-232,275-276, loop branch mem/engine_for_pretraining.py:141-149).

Same constructor, parameter names / shapes (timm 0.4.12 PatchEmbed / Block / Attention / Mlp attribute names, so
reference checkpoints load), the same initialisation order (same torch seed -> same weights), fixed 2-D sin-cos position
embeddings, per-sample random masking by argsort of uniform noise, decoder with mask tokens, per-patch MSE loss.

Execution (``precision``):
  * "bf16" (default, what `--mae 1` trains with): `MaeEngineBF16` -- the reference's autocast placement on the bf16 MFMA
    kernels of the pretraining model (gemm_p8 / gemm_tn_p8 with fused bias / GELU / GELU' / residual epilogues, the
    LDS-resident attention kernels, LayerNorm, AdamW): bf16 GEMM / attention operands, fp32 accumulate, fp32 residual
    stream, LayerNorm statistics, softmax and loss.  The decoder's 32-wide heads run on the 64-wide attention kernels
    through zero-padded head slots (padded shadow weights: the padding columns are exact zeros end to end).
  * "fp32": `MaeEngineF32` on csrc/fp32_path.hip (fp32 MFMA GEMMs, generic attention with 64- and 32-wide heads) -- the
    parity mode (loss 2e-6 from the reference's fp32 run).
No CPU / eager fallback.
"""
import math
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import require_gpu
from .vit_engine import ALIGN, _pad

MASK_RATIO = 0.5


# ---- fixed sin-cos position embedding (modeling_mae.py:21-99)
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=float)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    assert embed_dim % 2 == 0
    return np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                           get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])
    pos = get_2d_sincos_pos_embed_from_grid(embed_dim, grid)
    if cls_token:
        pos = np.concatenate([np.zeros([1, embed_dim]), pos], axis=0)
    return pos


# ---- parameter containers with timm 0.4.12's attribute names (construction order = the reference's RNG order)
class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop = nn.Dropout(0.0)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()


class MaskedAutoencoderViT(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0, norm_layer=nn.LayerNorm,
                 norm_pix_loss=False, LOSS_ONLY_MASKED_MAE=False):
        super().__init__()
        if norm_pix_loss:
            raise NotImplementedError("norm_pix_loss: the entrypoint builds the model with norm_pix_loss=0")
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([_Block(decoder_embed_dim, decoder_num_heads, mlp_ratio, norm_layer)
                                             for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self.LOSS_ONLY_MASKED_MAE = LOSS_ONLY_MASKED_MAE
        self.in_chans, self.embed_dim = in_chans, embed_dim
        print(f"LOSS_ONLY_MASKED_MAE = {self.LOSS_ONLY_MASKED_MAE}")
        self.initialize_weights()
        self._engine = None

    def initialize_weights(self):
        g = int(self.patch_embed.num_patches ** 0.5)
        self.pos_embed.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.pos_embed.shape[-1], g, True)).float().unsqueeze(0))
        self.decoder_pos_embed.data.copy_(
            torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g, True)).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        torch.nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        torch.nn.init.normal_(self.cls_token, std=0.02)
        torch.nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        return set()

    # ------------------------------------------------------------------ helpers of the reference surface
    def patchify(self, imgs):
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        c = imgs.shape[1]
        x = imgs.reshape(imgs.shape[0], c, h, p, w, p)
        return torch.einsum("nchpwq->nhwpqc", x).reshape(imgs.shape[0], h * w, p * p * c)

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        c = x.shape[2] // (p * p)
        x = x.reshape(x.shape[0], h, w, p, p, c)
        return torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], c, h * p, h * p)

    @staticmethod
    def masking_indices(noise, mask_ratio):
        """random_masking (modeling_mae.py:204-231) without the gather: ids_keep, mask (0 keep / 1 remove), ids_restore."""
        N, L = noise.shape
        len_keep = int(L * (1 - mask_ratio))
        ids_shuffle = torch.argsort(noise, dim=1)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        ids_keep = ids_shuffle[:, :len_keep].contiguous()
        mask = torch.ones([N, L], device=noise.device)
        mask[:, :len_keep] = 0
        mask = torch.gather(mask, dim=1, index=ids_restore)
        return ids_keep, mask, ids_restore.contiguous()

    # ------------------------------------------------------------------ fused execution
    @property
    def engine(self):
        if self._engine is None:
            require_gpu()
            self._engine = MaeEngineF32(self) if getattr(self, "precision", "bf16") == "fp32" else MaeEngineBF16(self)
        return self._engine

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        if self._engine is not None:
            self._engine.weights_dirty = True
        return r

    def forward_loss(self, imgs, mask_ratio=MASK_RATIO, noise=None):
        """loss (device scalar tensor [1]) with everything kept for `backward()`; `noise` [N, L] overrides the draw."""
        eng = self.engine
        imgs = imgs.to(device=eng.dev, dtype=torch.float32).contiguous()
        if noise is None:
            noise = torch.rand(imgs.shape[0], eng.L, device=eng.dev)        # modeling_mae.py:213
        ids_keep, mask, ids_restore = self.masking_indices(noise.to(eng.dev), MASK_RATIO)   # the reference ignores mask_ratio (:295)
        eng.forward(imgs, ids_keep, ids_restore, mask.contiguous())
        self._last_mask = mask
        return eng.loss_acc

    def backward(self):
        self.engine.backward()

    def forward(self, imgs, mask_ratio=MASK_RATIO, noise=None):
        """-> (loss, unpatchify(pred), mask) like the reference (:294-298); inference surface (no autograd graph: training
        goes through forward_loss / backward)."""
        la = self.forward_loss(imgs, mask_ratio, noise)
        eng = self.engine
        B = imgs.shape[0]
        pred = eng.pred[: B * eng.T].view(B, eng.T, -1)[:, 1:, :].float()
        return la[0].clone(), self.unpatchify(pred.clone()), self._last_mask


def mae_vit_base_patch16_dec512d8b(norm_pix_loss=False, LOSS_ONLY_MASKED_MAE=False, precision="bf16", **kwargs):
    m = MaskedAutoencoderViT(patch_size=16, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                             decoder_depth=8, decoder_num_heads=16, mlp_ratio=4,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), norm_pix_loss=norm_pix_loss,
                             LOSS_ONLY_MASKED_MAE=LOSS_ONLY_MASKED_MAE, **kwargs)
    assert precision in ("bf16", "fp32")
    m.precision = precision
    return m


class MaeEngineF32:
    """Flat fp32 parameter / gradient buffers (same contract as ViTEngine: FlatAdamW, GradReducer, checkpoints work
    unchanged) + the explicit forward / backward of the MAE model on the fp32 kernels."""
    precision = "fp32"

    def __init__(self, model):
        self.model = model
        p0 = next(model.parameters())
        assert p0.is_cuda, "mem_amd runs on the GPU only: move the model to cuda first (no CPU fallback)"
        self.dev = p0.device
        pe = model.patch_embed
        self.C = pe.proj.weight.shape[1]
        self.ph, self.pw = pe.patch_size
        self.H, self.W = pe.img_size
        self.L = pe.num_patches
        self.T = self.L + 1
        self.D = model.embed_dim
        self.Dd = model.decoder_embed.weight.shape[0]
        self.Pp = model.decoder_pred.weight.shape[0]
        self.Kpe = self.C * self.ph * self.pw
        self.enc = dict(pre="blocks.", depth=len(model.blocks), D=self.D, heads=model.blocks[0].attn.num_heads,
                        hidden=model.blocks[0].mlp.fc1.weight.shape[0])
        self.dec = dict(pre="decoder_blocks.", depth=len(model.decoder_blocks), D=self.Dd,
                        heads=model.decoder_blocks[0].attn.num_heads, hidden=model.decoder_blocks[0].mlp.fc1.weight.shape[0])
        self.eps = float(model.norm.eps)
        named = {n: p for n, p in model.named_parameters() if p.requires_grad}
        skip = model.no_weight_decay()
        segs, off, flags = {}, 0, []
        names = list(named)
        dec_names = [n for n in names if n.startswith("decoder") or n == "mask_token"]
        enc_names = [n for n in names if n not in set(dec_names)]
        buckets = []
        for bname, group in (("decoder", dec_names), ("encoder", enc_names)):      # backward produces the decoder first
            b0 = off
            for n in group:
                p = named[n]
                segs[n] = (off, p.numel())
                size = _pad(p.numel(), ALIGN)
                decay = not (p.ndim == 1 or n.endswith(".bias") or n in skip)
                flags += [1 if decay else 0] * (size // ALIGN)
                off += size
            buckets.append((bname, b0, off))
        self.nflat, self.segs, self.buckets, self.named = off, segs, buckets, named
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.wd_flags = torch.tensor(flags, dtype=torch.uint8, device=self.dev)
        self.decay_names = [n for n, p in named.items() if not (p.ndim == 1 or n.endswith(".bias") or n in skip)]
        for n, p in named.items():
            o, k = segs[n]
            view = self.flat_p[o:o + k].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + k].view(p.shape)
        self.gn_ws = torch.zeros(1024, dtype=torch.float64, device=self.dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.loss_acc = torch.zeros(2, dtype=torch.float32, device=self.dev)       # [loss, 0] (mlm_acc is 0 for MAE)
        self.scratch2 = torch.zeros(2, dtype=torch.float32, device=self.dev)
        self.grad_hook = None
        self.weights_dirty = True
        self.B = 0
        self.wT = {}

    def P(self, name):
        o, k = self.segs[name]
        return self.flat_p[o:o + k]

    def G(self, name):
        o, k = self.segs[name]
        return self.flat_g[o:o + k]

    def Wm(self, name):
        p = self.named[name]
        o, k = self.segs[name]
        return self.flat_p[o:o + k].view(p.shape[0], -1)

    def attach_grads(self):
        for n, p in self.named.items():
            o, k = self.segs[n]
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + k].view(p.shape)

    def _lin_names(self):
        out = ["decoder_embed.weight", "decoder_pred.weight"]
        for spec in (self.enc, self.dec):
            for i in range(spec["depth"]):
                out += [f"{spec['pre']}{i}.{k}.weight" for k in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")]
        return out

    def sync_weights(self):
        for n in self._lin_names():
            w = self.Wm(n)
            if n not in self.wT:
                self.wT[n] = torch.empty((w.shape[1], w.shape[0]), dtype=torch.float32, device=self.dev)
            ops.f32_transpose(w, w.shape[0], w.shape[1], self.wT[n])
        self.weights_dirty = False

    def ensure_batch(self, B, K):
        if B <= self.B and K == getattr(self, "K", None):
            return
        dev, f = self.dev, torch.float32
        e = lambda *s: torch.empty(s, dtype=f, device=dev)   # noqa: E731
        L, T, D, Dd = self.L, self.T, self.D, self.Dd
        Me, Md = B * (K + 1), B * T
        self.patches, self.xe = e(B * L, self.Kpe), e(B * L, D)

        def acts(spec, M):
            Dm, Hd = spec["D"], spec["hidden"]
            return dict(x=[torch.zeros((M, Dm), dtype=f, device=dev) for _ in range(2 * spec["depth"] + 1)],
                        a=[dict(h1=e(M, Dm), qkv=e(M, 3 * Dm), ao=e(M, Dm), h2=e(M, Dm), hpre=e(M, Hd), a=e(M, Hd),
                                mean1=e(M), rstd1=e(M), mean2=e(M), rstd2=e(M)) for _ in range(spec["depth"])],
                        dx=torch.zeros((M, Dm), dtype=f, device=dev), dh=e(M, Dm), dbig=e(M, Hd), dqkv=e(M, 3 * Dm), dao=e(M, Dm))
        self.ea, self.da = acts(self.enc, Me), acts(self.dec, Md)
        self.latent, self.meanE, self.rstdE = e(Me, D), e(Me), e(Me)
        self.yd, self.dyd = e(Me, Dd), e(Me, Dd)
        self.hdn, self.meanD, self.rstdD = e(Md, Dd), e(Md), e(Md)
        self.pred, self.dpred = e(Md, self.Pp), e(Md, self.Pp)
        self.row_loss = e(B * L)
        self.dlat, self.dxe = e(Me, D), e(B * L, D)
        Rp = _pad(max(Md, B * L), 32)
        wide = max(3 * D, self.enc["hidden"], 3 * Dd, self.dec["hidden"], self.Pp, self.Kpe)
        self.tA, self.tB = e(wide, Rp), e(wide, Rp)
        self.B, self.K = B, K

    # ---- generic timm Block (x = x + attn(norm1(x)); x = x + mlp(norm2(x)))
    def _blk_fwd(self, spec, acts, i, B, T):
        P, G = self.P, ops.f32_gemm_nt
        D, Hd, heads = spec["D"], spec["hidden"], spec["heads"]
        M = B * T
        pre = f"{spec['pre']}{i}."
        a = acts["a"][i]
        xin, xmid, xout = acts["x"][2 * i], acts["x"][2 * i + 1], acts["x"][2 * i + 2]
        scale = (D // heads) ** -0.5
        ops.f32_layernorm_fwd(xin, P(pre + "norm1.weight"), P(pre + "norm1.bias"), a["h1"], a["mean1"], a["rstd1"], M, D, eps=self.eps)
        G(a["h1"], self.Wm(pre + "attn.qkv.weight"), M, 3 * D, D, ops.EPI_BIAS_BF16, out0=a["qkv"], bias=P(pre + "attn.qkv.bias"),
          colscale=scale, colscale_n=D)
        ops.f32_attn_fwd(a["qkv"], B, T, D, heads, None, None, a["ao"])
        G(a["ao"], self.Wm(pre + "attn.proj.weight"), M, D, D, ops.EPI_RESIDUAL, bias=P(pre + "attn.proj.bias"), resid=xmid,
          aux=xin, ldaux=D, rows_per_sample=T)
        ops.f32_layernorm_fwd(xmid, P(pre + "norm2.weight"), P(pre + "norm2.bias"), a["h2"], a["mean2"], a["rstd2"], M, D, eps=self.eps)
        G(a["h2"], self.Wm(pre + "mlp.fc1.weight"), M, Hd, D, ops.EPI_BIAS_GELU, out0=a["hpre"], out1=a["a"], bias=P(pre + "mlp.fc1.bias"))
        G(a["a"], self.Wm(pre + "mlp.fc2.weight"), M, D, Hd, ops.EPI_RESIDUAL, bias=P(pre + "mlp.fc2.bias"), resid=xout, aux=xmid,
          ldaux=D, rows_per_sample=T)

    def _wgrad(self, dY, X, R, n_out, n_in, gname):
        Rp = _pad(R, 32)
        tA, tB = self.tA[:n_out, :Rp], self.tB[:n_in, :Rp]
        ops.f32_transpose(dY, R, n_out, tA)
        ops.f32_transpose(X, R, n_in, tB)
        ops.f32_gemm_nt(tA, tB, n_out, n_in, Rp, ops.EPI_F32, out0=self.G(gname).view(n_out, n_in), accumulate=True)

    def _blk_bwd(self, spec, acts, i, B, T):
        P, Gr, G = self.P, self.G, ops.f32_gemm_nt
        D, Hd, heads = spec["D"], spec["hidden"], spec["heads"]
        M = B * T
        pre = f"{spec['pre']}{i}."
        a = acts["a"][i]
        xin, xmid = acts["x"][2 * i], acts["x"][2 * i + 1]
        dx, dh, dbig, dqkv, dao = acts["dx"], acts["dh"], acts["dbig"], acts["dqkv"], acts["dao"]
        scale = (D // heads) ** -0.5
        # MLP branch: the branch output gradient IS dx (no layer scale, no drop path)
        ops.f32_colsum(dx, M, D, Gr(pre + "mlp.fc2.bias"))
        G(dx, self.wT[pre + "mlp.fc2.weight"], M, Hd, D, ops.EPI_DGELU, out0=dbig, aux=a["hpre"], colsum=Gr(pre + "mlp.fc1.bias"))
        self._wgrad(dx, a["a"], M, D, Hd, pre + "mlp.fc2.weight")
        self._wgrad(dbig, a["h2"], M, Hd, D, pre + "mlp.fc1.weight")
        G(dbig, self.wT[pre + "mlp.fc1.weight"], M, D, Hd, ops.EPI_BIAS_BF16, out0=dh)
        ops.f32_layernorm_bwd(dh, xmid, P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx, Gr(pre + "norm2.weight"),
                              Gr(pre + "norm2.bias"), M, D, accumulate=True)
        # attention branch
        ops.f32_colsum(dx, M, D, Gr(pre + "attn.proj.bias"))
        G(dx, self.wT[pre + "attn.proj.weight"], M, D, D, ops.EPI_BIAS_BF16, out0=dao)
        self._wgrad(dx, a["ao"], M, D, D, pre + "attn.proj.weight")
        ops.f32_attn_bwd(a["qkv"], dao, B, T, D, heads, scale, None, None, dqkv, None)
        ops.f32_colsum(dqkv, M, 3 * D, Gr(pre + "attn.qkv.bias"))
        self._wgrad(dqkv, a["h1"], M, 3 * D, D, pre + "attn.qkv.weight")
        G(dqkv, self.wT[pre + "attn.qkv.weight"], M, D, 3 * D, ops.EPI_BIAS_BF16, out0=dh)
        ops.f32_layernorm_bwd(dh, xin, P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx, Gr(pre + "norm1.weight"),
                              Gr(pre + "norm1.bias"), M, D, accumulate=True)

    # ------------------------------------------------------------------ forward / backward
    def forward(self, imgs, ids_keep, ids_restore, mask):
        assert imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous()
        B = imgs.shape[0]
        assert tuple(imgs.shape[1:]) == (self.C, self.H, self.W), f"Input image size {tuple(imgs.shape)} doesn't match the model"
        K = ids_keep.shape[1]
        self.ensure_batch(B, K)
        if self.weights_dirty:
            self.sync_weights()
        P, G = self.P, ops.f32_gemm_nt
        m = self.model
        L, T, D, Dd = self.L, self.T, self.D, self.Dd
        self.cur = dict(B=B, K=K, ids_keep=ids_keep, ids_restore=ids_restore, mask=mask, imgs=imgs)
        ops.f32_im2col(imgs, B, self.C, self.H, self.W, self.ph, self.pw, self.patches)
        G(self.patches, self.Wm("patch_embed.proj.weight"), B * L, D, self.Kpe, ops.EPI_BIAS_BF16, out0=self.xe,
          bias=P("patch_embed.proj.bias"))
        ops.mae_enc_assemble(self.xe, m.pos_embed.data.view(T, D), P("cls_token"), ids_keep, B, L, K, D, self.ea["x"][0])
        for i in range(self.enc["depth"]):
            self._blk_fwd(self.enc, self.ea, i, B, K + 1)
        Me, Md = B * (K + 1), B * T
        ops.f32_layernorm_fwd(self.ea["x"][-1], P("norm.weight"), P("norm.bias"), self.latent, self.meanE, self.rstdE, Me, D, eps=self.eps)
        G(self.latent, self.Wm("decoder_embed.weight"), Me, Dd, D, ops.EPI_BIAS_BF16, out0=self.yd, bias=P("decoder_embed.bias"))
        ops.mae_dec_assemble(self.yd, P("mask_token"), m.decoder_pos_embed.data.view(T, Dd), ids_restore, B, L, K, Dd, self.da["x"][0])
        for i in range(self.dec["depth"]):
            self._blk_fwd(self.dec, self.da, i, B, T)
        ops.f32_layernorm_fwd(self.da["x"][-1], P("decoder_norm.weight"), P("decoder_norm.bias"), self.hdn, self.meanD, self.rstdD,
                              Md, Dd, eps=self.eps)
        G(self.hdn, self.Wm("decoder_pred.weight"), Md, self.Pp, Dd, ops.EPI_BIAS_BF16, out0=self.pred, bias=P("decoder_pred.bias"))
        ops.mae_loss(self.pred, imgs, mask, B, self.C, self.H, self.W, self.ph, m.LOSS_ONLY_MASKED_MAE, self.row_loss, self.dpred,
                     self.scratch2)
        self.loss_acc[0:1].copy_(self.scratch2[1:2])
        return self.loss_acc

    def backward(self):
        c = self.cur
        B, K = c["B"], c["K"]
        P, Gr, G = self.P, self.G, ops.f32_gemm_nt
        L, T, D, Dd = self.L, self.T, self.D, self.Dd
        Me, Md = B * (K + 1), B * T
        self.attach_grads()
        self.flat_g.zero_()
        # decoder head
        self._wgrad(self.dpred, self.hdn, Md, self.Pp, Dd, "decoder_pred.weight")
        ops.f32_colsum(self.dpred, Md, self.Pp, Gr("decoder_pred.bias"))
        G(self.dpred, self.wT["decoder_pred.weight"], Md, Dd, self.Pp, ops.EPI_BIAS_BF16, out0=self.da["dh"])
        ops.f32_layernorm_bwd(self.da["dh"], self.da["x"][-1], P("decoder_norm.weight"), self.meanD, self.rstdD, self.da["dx"],
                              Gr("decoder_norm.weight"), Gr("decoder_norm.bias"), Md, Dd, accumulate=False)
        for i in reversed(range(self.dec["depth"])):
            self._blk_bwd(self.dec, self.da, i, B, T)
        ops.mae_dec_assemble_bwd(self.da["dx"], c["ids_restore"], B, L, K, Dd, self.dyd, Gr("mask_token"))
        self._wgrad(self.dyd, self.latent, Me, Dd, D, "decoder_embed.weight")
        ops.f32_colsum(self.dyd, Me, Dd, Gr("decoder_embed.bias"))
        if self.grad_hook:
            self.grad_hook(0)
        G(self.dyd, self.wT["decoder_embed.weight"], Me, D, Dd, ops.EPI_BIAS_BF16, out0=self.dlat)
        ops.f32_layernorm_bwd(self.dlat, self.ea["x"][-1], P("norm.weight"), self.meanE, self.rstdE, self.ea["dx"],
                              Gr("norm.weight"), Gr("norm.bias"), Me, D, accumulate=False)
        for i in reversed(range(self.enc["depth"])):
            self._blk_bwd(self.enc, self.ea, i, B, K + 1)
        ops.mae_enc_assemble_bwd(self.ea["dx"], c["ids_keep"], B, L, K, D, self.dxe, Gr("cls_token"))
        self._wgrad(self.dxe, self.patches, B * L, D, self.Kpe, "patch_embed.proj.weight")
        ops.f32_colsum(self.dxe, B * L, D, Gr("patch_embed.proj.bias"))
        if self.grad_hook:
            self.grad_hook(1)

    # ------------------------------------------------------------------ optimizer primitives (ViTEngine contract)
    def grad_norm(self):
        ops.grad_norm(self.flat_g, self.nflat, self.gnorm, self.gn_ws)
        return self.gnorm

    def adamw_step(self, m, v, lr, wd, step, betas=(0.9, 0.95), eps=1e-8, max_norm=0.0):
        ops.adamw(self.flat_p, self.flat_g, m, v, self.nflat, self.wd_flags, lr, betas[0], betas[1], eps, wd, step,
                  gnorm=self.gnorm, max_norm=max_norm or 0.0)
        self.weights_dirty = True



class MaeEngineBF16(MaeEngineF32):
    """The MAE model on the bf16 MFMA kernels (same flat fp32 master buffers, optimizer and reducer contract as
    MaeEngineF32).  Rounding points = the reference under autocast (mem/engine_for_pretraining.py:141-149): every Linear /
    Conv output is bf16 (fp32 accumulate + fp32 bias), the residual stream, LayerNorm, softmax and the loss are fp32.

    32-wide heads (the decoder: 512 / 16): the attention kernels are built for 64-wide heads, so q / k / v of head h live
    in columns [64 h, 64 h + 32) of a padded [*, 3 * 64 * heads] qkv matrix and the other 32 columns are zero.  The qkv
    and proj weights have padded bf16 shadows (zero rows / columns), so the GEMMs produce and consume the padded layout
    directly; q k^T, the softmax and the real output columns are unchanged, the padded columns of every activation and
    gradient are exact zeros, and the weight gradients of the padded rows / columns (zero) are dropped when the real rows
    are copied back."""
    precision = "bf16"

    def __init__(self, model):
        super().__init__(model)
        dev = self.dev
        for spec in (self.enc, self.dec):
            hd = spec["D"] // spec["heads"]
            assert hd in (32, 64) and spec["D"] % 64 == 0 and spec["hidden"] % 64 == 0, \
                "bf16 MAE engine: head_dim 32 or 64, widths multiples of 64 (use precision='fp32' otherwise)"
            spec["hd"], spec["Dp"] = hd, spec["heads"] * 64
        assert self.Kpe % 64 == 0 and self.Pp % 8 == 0
        self.w16, self.wT16, self.bpad = {}, {}, {}
        self.tn_ws = None
        self.set_gelu_dg(True)

    def set_gelu_dg(self, on):
        """fc1 keeps gelu'(h) (fp16) for the backward instead of the pre-activation (vit_engine.ViTEngine.set_gelu_dg)."""
        self.epi_gelu, self.epi_dgelu = (ops.EPI_BIAS_GELU_DG, ops.EPI_MUL_AUX) if on else (ops.EPI_BIAS_GELU, ops.EPI_DGELU)

    # ---- bf16 shadows of the Linear weights ([out,in] for forward, [in,out] for dgrad), padded where heads are 32 wide
    def _padded(self, spec, kind, W):
        """fp32 [out, in] -> the padded fp32 matrix the shadows are cast from (None: no padding for this spec)."""
        if spec["hd"] == 64:
            return None
        H, D, Dp = spec["heads"], spec["D"], spec["Dp"]
        if kind == "qkv":
            Wp = torch.zeros((3 * Dp, D), dtype=torch.float32, device=self.dev)
            Wp.view(3, H, 64, D)[:, :, :32, :] = W.view(3, H, 32, D)
        else:                                                       # proj: padded INPUT columns
            Wp = torch.zeros((D, Dp), dtype=torch.float32, device=self.dev)
            Wp.view(D, H, 64)[:, :, :32] = W.view(D, H, 32)
        return Wp

    def _build_shadows(self):
        """Once: the bf16 twin of the whole flat master buffer ([out,in] shadows of unpadded weights are views of it), the
        padded fp32 staging matrices (zero outside the real rows / columns, which every sync overwrites), their bf16
        shadows, and the descriptors of ONE batched transpose launch for all [in,out]-major copies."""
        import numpy as np
        dev = self.dev
        self.flat_w16 = torch.zeros(self.nflat, dtype=torch.bfloat16, device=dev)
        self.pad_src, self.pad_cast, items = [], [], []   # pad_src: (staging fp32, view of its real part, fp32 master view)

        def add(name, src32, o, i, w16):
            self.w16[name] = w16
            self.wT16[name] = torch.empty((i, o), dtype=torch.bfloat16, device=dev)
            items.append((src32, o, i, self.wT16[name]))

        def plain(name):
            W = self.Wm(name)
            o, i = W.shape
            off, k = self.segs[name]
            add(name, W, o, i, self.flat_w16[off:off + k].view(o, i))
        plain("patch_embed.proj.weight"); plain("decoder_embed.weight"); plain("decoder_pred.weight")
        for spec in (self.enc, self.dec):
            H, D, Dp = spec["heads"], spec["D"], spec["Dp"]
            for i in range(spec["depth"]):
                pre = f"{spec['pre']}{i}."
                plain(pre + "mlp.fc1.weight"); plain(pre + "mlp.fc2.weight")
                if spec["hd"] == 64:
                    plain(pre + "attn.qkv.weight"); plain(pre + "attn.proj.weight")
                    continue
                Wq = torch.zeros((3 * Dp, D), dtype=torch.float32, device=dev)
                Wp = torch.zeros((D, Dp), dtype=torch.float32, device=dev)
                b = torch.zeros(3 * Dp, dtype=torch.float32, device=dev)
                self.pad_src += [(Wq, Wq.view(3, H, 64, D)[:, :, :32, :], self.Wm(pre + "attn.qkv.weight").view(3, H, 32, D)),
                                 (Wp, Wp.view(D, H, 64)[:, :, :32], self.Wm(pre + "attn.proj.weight").view(D, H, 32)),
                                 (None, b.view(3, H, 64)[:, :, :32], self.P(pre + "attn.qkv.bias").view(3, H, 32))]
                self.bpad[pre] = b
                add(pre + "attn.qkv.weight", Wq, 3 * Dp, D, torch.empty((3 * Dp, D), dtype=torch.bfloat16, device=dev))
                add(pre + "attn.proj.weight", Wp, D, Dp, torch.empty((D, Dp), dtype=torch.bfloat16, device=dev))
                self.pad_cast += [(Wq, self.w16[pre + "attn.qkv.weight"]), (Wp, self.w16[pre + "attn.proj.weight"])]
        desc = np.zeros((len(items), 6), dtype=np.int64)
        prefix = np.zeros(len(items) + 1, dtype=np.int32)
        for k, (src, R, Cc, dst) in enumerate(items):
            desc[k] = (src.data_ptr(), src.stride(0), R, Cc, dst.data_ptr(), dst.stride(0))
            prefix[k + 1] = prefix[k] + ((R + 63) // 64) * ((Cc + 63) // 64)
        self._tdesc, self._tprefix = torch.from_numpy(desc).to(dev), torch.from_numpy(prefix).to(dev)
        self._tn, self._ttiles = len(items), int(prefix[-1])

    def sync_weights(self):
        """fp32 masters -> bf16 shadows: one flat cast, the padded matrices refreshed (copy of the real part + cast), one
        batched transpose launch for the [in,out]-major copies."""
        if not hasattr(self, "flat_w16"):
            self._build_shadows()
        ops.cast_f32_bf16(self.flat_p, self.flat_w16, self.nflat)
        for stage, real, master in self.pad_src:
            real.copy_(master)
        for stage, w in self.pad_cast:
            ops.cast_f32_bf16(stage, w, stage.numel())
        ops.transpose_cast_batched(self._tdesc, self._tprefix, self._tn, self._ttiles)
        self.weights_dirty = False

    CS_COPIES = 8
    FUSE_LN_BRANCH = True      # LayerNorm backward fused with the following branch backward (the ViT engine's ln_bwd_branch kernel)

    def ensure_batch(self, B, K):
        if B <= self.B and K == getattr(self, "K", None):
            return
        dev, f, h = self.dev, torch.float32, torch.bfloat16
        e = lambda *s: torch.empty(s, dtype=f, device=dev)      # noqa: E731
        e16 = lambda *s: torch.empty(s, dtype=h, device=dev)    # noqa: E731
        L, T, D, Dd = self.L, self.T, self.D, self.Dd
        Me, Md = B * (K + 1), B * T
        self.patches, self.xe16, self.xe = e16(B * L, self.Kpe), e16(B * L, D), e(B * L, D)

        def acts(spec, M, Tt):
            Dm, Hd, Dp, heads = spec["D"], spec["hidden"], spec["Dp"], spec["heads"]
            TP = ops.attn_tokens_padded(Tt)
            window = (14, 14) if Tt == 197 else (1, Tt - 1)      # no position bias: any window with Tt - 1 cells (zero table)
            nrd = (2 * window[0] - 1) * (2 * window[1] - 1) + 3
            return dict(x=[torch.zeros((M, Dm), dtype=f, device=dev) for _ in range(2 * spec["depth"] + 1)],
                        a=[dict(h1=e16(M, Dm), qkv=torch.zeros((M, 3 * Dp), dtype=h, device=dev), ao=e16(M, Dp), h2=e16(M, Dm),
                                hpre=e16(M, Hd), a=e16(M, Hd), mean1=e(M), rstd1=e(M), mean2=e(M), rstd2=e(M),
                                lse=e(B, heads, TP)) for _ in range(spec["depth"])],
                        dx=torch.zeros((M, Dm), dtype=f, device=dev), dy16=e16(M, Dm), dh16=e16(M, Dm), dbig16=e16(M, Hd),
                        dqkv16=e16(M, 3 * Dp), dao16=e16(M, Dp), delta=e(2 * M + 4, heads),
                        window=window, table=torch.zeros((nrd, heads), dtype=f, device=dev),
                        gq=torch.zeros((3 * Dp, Dm), dtype=f, device=dev) if spec["hd"] == 32 else None,
                        gp=torch.zeros((Dm, Dp), dtype=f, device=dev) if spec["hd"] == 32 else None,
                        gb=torch.zeros(3 * Dp, dtype=f, device=dev) if spec["hd"] == 32 else None)
        self.ea, self.da = acts(self.enc, Me, K + 1), acts(self.dec, Md, T)
        self.cs_ws = torch.zeros(self.CS_COPIES * max(self.enc["hidden"], self.dec["hidden"], self.enc["Dp"], self.dec["Dp"]),
                                 dtype=f, device=dev)         # column-sum accumulator copies of the fused GEMM epilogues
        self.latent16, self.meanE, self.rstdE = e16(Me, D), e(Me), e(Me)
        self.yd16, self.yd, self.dyd, self.dyd16 = e16(Me, Dd), e(Me, Dd), e(Me, Dd), e16(Me, Dd)
        self.hdn16, self.meanD, self.rstdD = e16(Md, Dd), e(Md), e(Md)
        self.pred16, self.pred, self.dpred, self.dpred16 = e16(Md, self.Pp), e(Md, self.Pp), e(Md, self.Pp), e16(Md, self.Pp)
        self.row_loss = e(B * L)
        self.dlat16, self.dxe, self.dxe16 = e16(Me, D), e(B * L, D), e16(B * L, D)
        need = 0
        for spec, M in ((self.enc, Me), (self.dec, Md)):
            for n_out, n_in in ((3 * spec["Dp"], spec["D"]), (spec["D"], spec["Dp"]), (spec["hidden"], spec["D"]),
                                (spec["D"], spec["hidden"])):
                need = max(need, ops.gemm_tn_workspace(M, n_out, n_in))
        need = max(need, ops.gemm_tn_workspace(Md, self.Pp, Dd), ops.gemm_tn_workspace(Me, Dd, D),
                   ops.gemm_tn_workspace(B * L, D, self.Kpe))
        self.tn_ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        self.B, self.K = B, K

    # ---- timm Block on the bf16 kernels
    def _blk_fwd(self, spec, acts, i, B, T):
        P = self.P
        D, Hd, heads, Dp = spec["D"], spec["hidden"], spec["heads"], spec["Dp"]
        M = B * T
        pre = f"{spec['pre']}{i}."
        a = acts["a"][i]
        xin, xmid, xout = acts["x"][2 * i], acts["x"][2 * i + 1], acts["x"][2 * i + 2]
        scale = spec["hd"] ** -0.5
        qb = self.bpad[pre] if spec["hd"] == 32 else P(pre + "attn.qkv.bias")
        ops.layernorm_fwd(xin, P(pre + "norm1.weight"), P(pre + "norm1.bias"), a["h1"], a["mean1"], a["rstd1"], M, D, eps=self.eps)
        ops.gemm_nt(a["h1"], self.w16[pre + "attn.qkv.weight"], M, 3 * Dp, D, ops.EPI_BIAS_BF16, out0=a["qkv"], bias=qb,
                    colscale=scale, colscale_n=Dp)
        ops.attn_fwd(a["qkv"], B, T, Dp, heads, acts["table"], acts["window"], a["ao"], a["lse"])
        ops.gemm_nt(a["ao"], self.w16[pre + "attn.proj.weight"], M, D, Dp, ops.EPI_RESIDUAL, bias=P(pre + "attn.proj.bias"),
                    resid=xmid, aux=xin, ldaux=D, rows_per_sample=T)
        ops.layernorm_fwd(xmid, P(pre + "norm2.weight"), P(pre + "norm2.bias"), a["h2"], a["mean2"], a["rstd2"], M, D, eps=self.eps)
        # (round 4: fc1 stores gelu'(h) as fp16 instead of the pre-activation, the backward multiplies: ViTEngine.set_gelu_dg)
        ops.gemm_nt(a["h2"], self.w16[pre + "mlp.fc1.weight"], M, Hd, D, self.epi_gelu, out0=a["hpre"], out1=a["a"],
                    bias=P(pre + "mlp.fc1.bias"))
        ops.gemm_nt(a["a"], self.w16[pre + "mlp.fc2.weight"], M, D, Hd, ops.EPI_RESIDUAL, bias=P(pre + "mlp.fc2.bias"),
                    resid=xout, aux=xmid, ldaux=D, rows_per_sample=T)

    def _wgrad16(self, dY, X, R, n_out, n_in, out):
        ops.gemm_tn(dY, X, R, n_out, n_in, out, accumulate=True, workspace=self.tn_ws)

    def _blk_bwd(self, spec, acts, i, B, T):
        P, Gr = self.P, self.G
        D, Hd, heads, Dp = spec["D"], spec["hidden"], spec["heads"], spec["Dp"]
        M = B * T
        pre = f"{spec['pre']}{i}."
        a = acts["a"][i]
        xin, xmid = acts["x"][2 * i], acts["x"][2 * i + 1]
        dx, dy, dh, dbig, dqkv, dao = acts["dx"], acts["dy16"], acts["dh16"], acts["dbig16"], acts["dqkv16"], acts["dao16"]
        scale = spec["hd"] ** -0.5
        pad = spec["hd"] == 32
        Gw = lambda n: Gr(n).view(self.named[n].shape[0], -1)       # noqa: E731
        # MLP branch: the branch output gradient IS dx (no layer scale, no drop path); Linear grad_outputs are bf16.
        # dy = bf16(dx) + its column sums: for every block but the last of a stack this already came out of the fused norm1
        # backward of block i + 1 (below); FUSE_LN_BRANCH = False keeps the two-kernel form (A/B, tests)
        fuse = self.FUSE_LN_BRANCH and D <= 1024
        if i == spec["depth"] - 1 or not fuse:
            ops.branch_bwd(dx, None, None, dy, None, Gr(pre + "mlp.fc2.bias"), M, D)
        # (fused column sums go to CS_COPIES accumulator copies, folded by a tiny kernel: atomics on one address serialise)
        ops.gemm_nt(dy, self.wT16[pre + "mlp.fc2.weight"], M, Hd, D, self.epi_dgelu, out0=dbig, aux=a["hpre"],
                    colsum=self.cs_ws, colsum_copies=self.CS_COPIES)
        ops.colsum_fold(self.cs_ws, self.CS_COPIES, Hd, Gr(pre + "mlp.fc1.bias"))
        self._wgrad16(dy, a["a"], M, D, Hd, Gw(pre + "mlp.fc2.weight"))
        self._wgrad16(dbig, a["h2"], M, Hd, D, Gw(pre + "mlp.fc1.weight"))
        ops.gemm_nt(dbig, self.wT16[pre + "mlp.fc1.weight"], M, D, Hd, ops.EPI_BIAS_BF16, out0=dh)
        # norm2 backward into dx + the attention branch's dy = bf16(dx) and proj-bias column sums: one pass over dx
        if fuse:
            ops.layernorm_bwd_branch(dh, xmid, P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx, Gr(pre + "norm2.weight"),
                                     Gr(pre + "norm2.bias"), M, D, None, None, dy, None, Gr(pre + "attn.proj.bias"))
        else:
            ops.layernorm_bwd(dh, xmid, P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx, Gr(pre + "norm2.weight"),
                              Gr(pre + "norm2.bias"), M, D, accumulate=True)
            ops.branch_bwd(dx, None, None, dy, None, Gr(pre + "attn.proj.bias"), M, D)
        # qkv.bias gradient without a pass over dqkv: the v part is colsum(dao) (sum_k dV_k = sum_q dO_q: softmax rows sum to
        # one), fused into the GEMM that produces dao; the q part comes out of the attention backward kernel; the k part is
        # zero in real arithmetic (sum_k dS_qk = 0 for every query row) and is left at zero
        gb = acts["gb"] if pad else Gr(pre + "attn.qkv.bias")
        if pad:
            gb.zero_()
        ops.gemm_nt(dy, self.wT16[pre + "attn.proj.weight"], M, Dp, D, ops.EPI_BIAS_BF16, out0=dao, colsum=self.cs_ws,
                    colsum_copies=self.CS_COPIES)
        ops.colsum_fold(self.cs_ws, self.CS_COPIES, Dp, gb[2 * Dp:3 * Dp])
        if pad:
            acts["gp"].zero_()
            self._wgrad16(dy, a["ao"], M, D, Dp, acts["gp"])
            Gw(pre + "attn.proj.weight").view(D, heads, 32).add_(acts["gp"].view(D, heads, 64)[:, :, :32])
        else:
            self._wgrad16(dy, a["ao"], M, D, Dp, Gw(pre + "attn.proj.weight"))
        ops.attn_bwd(a["qkv"], dao, a["lse"], acts["delta"], acts["table"], acts["window"], B, T, Dp, heads, scale, dqkv, None,
                     dq_bias=gb[0:Dp], out=a["ao"])
        if pad:
            acts["gq"].zero_()
            self._wgrad16(dqkv, a["h1"], M, 3 * Dp, D, acts["gq"])
            Gr(pre + "attn.qkv.bias").view(3, heads, 32).add_(acts["gb"].view(3, heads, 64)[:, :, :32])
            Gw(pre + "attn.qkv.weight").view(3, heads, 32, D).add_(acts["gq"].view(3, heads, 64, D)[:, :, :32, :])
        else:
            self._wgrad16(dqkv, a["h1"], M, 3 * D, D, Gw(pre + "attn.qkv.weight"))
        ops.gemm_nt(dqkv, self.wT16[pre + "attn.qkv.weight"], M, D, 3 * Dp, ops.EPI_BIAS_BF16, out0=dh)
        if fuse and i > 0:        # norm1 backward of block i + the MLP branch's dy / fc2-bias column sums of block i - 1
            pb = f"{spec['pre']}{i - 1}."
            ops.layernorm_bwd_branch(dh, xin, P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx, Gr(pre + "norm1.weight"),
                                     Gr(pre + "norm1.bias"), M, D, None, None, dy, None, Gr(pb + "mlp.fc2.bias"))
        else:
            ops.layernorm_bwd(dh, xin, P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx, Gr(pre + "norm1.weight"),
                              Gr(pre + "norm1.bias"), M, D, accumulate=True)

    def forward(self, imgs, ids_keep, ids_restore, mask):
        assert imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous()
        B = imgs.shape[0]
        assert tuple(imgs.shape[1:]) == (self.C, self.H, self.W), f"Input image size {tuple(imgs.shape)} doesn't match the model"
        K = ids_keep.shape[1]
        self.ensure_batch(B, K)
        if self.weights_dirty:
            self.sync_weights()
        P, m = self.P, self.model
        L, T, D, Dd = self.L, self.T, self.D, self.Dd
        Me, Md = B * (K + 1), B * T
        self.cur = dict(B=B, K=K, ids_keep=ids_keep, ids_restore=ids_restore, mask=mask, imgs=imgs)
        ops.im2col(imgs, B, self.C, self.H, self.W, self.ph, self.pw, self.patches)
        ops.gemm_nt(self.patches, self.w16["patch_embed.proj.weight"], B * L, D, self.Kpe, ops.EPI_BIAS_BF16, out0=self.xe16,
                    bias=P("patch_embed.proj.bias"))
        self.xe[: B * L].copy_(self.xe16[: B * L])                   # bf16 conv output + fp32 pos_embed -> fp32 (type promotion)
        ops.mae_enc_assemble(self.xe, m.pos_embed.data.view(T, D), P("cls_token"), ids_keep, B, L, K, D, self.ea["x"][0])
        for i in range(self.enc["depth"]):
            self._blk_fwd(self.enc, self.ea, i, B, K + 1)
        ops.layernorm_fwd(self.ea["x"][-1], P("norm.weight"), P("norm.bias"), self.latent16, self.meanE, self.rstdE, Me, D, eps=self.eps)
        ops.gemm_nt(self.latent16, self.w16["decoder_embed.weight"], Me, Dd, D, ops.EPI_BIAS_BF16, out0=self.yd16,
                    bias=P("decoder_embed.bias"))
        self.yd[:Me].copy_(self.yd16[:Me])
        ops.mae_dec_assemble(self.yd, P("mask_token"), m.decoder_pos_embed.data.view(T, Dd), ids_restore, B, L, K, Dd, self.da["x"][0])
        for i in range(self.dec["depth"]):
            self._blk_fwd(self.dec, self.da, i, B, T)
        ops.layernorm_fwd(self.da["x"][-1], P("decoder_norm.weight"), P("decoder_norm.bias"), self.hdn16, self.meanD, self.rstdD,
                          Md, Dd, eps=self.eps)
        ops.gemm_nt(self.hdn16, self.w16["decoder_pred.weight"], Md, self.Pp, Dd, ops.EPI_BIAS_BF16, out0=self.pred16,
                    bias=P("decoder_pred.bias"))
        self.pred[:Md].copy_(self.pred16[:Md])                       # (pred - target) ** 2 runs in fp32 on the bf16 prediction
        ops.mae_loss(self.pred, imgs, mask, B, self.C, self.H, self.W, self.ph, m.LOSS_ONLY_MASKED_MAE, self.row_loss, self.dpred,
                     self.scratch2)
        self.loss_acc[0:1].copy_(self.scratch2[1:2])
        return self.loss_acc

    def backward(self):
        c = self.cur
        B, K = c["B"], c["K"]
        P, Gr = self.P, self.G
        L, T, D, Dd = self.L, self.T, self.D, self.Dd
        Me, Md = B * (K + 1), B * T
        Gw = lambda n: Gr(n).view(self.named[n].shape[0], -1)       # noqa: E731
        self.attach_grads()
        self.flat_g.zero_()
        ops.branch_bwd(self.dpred, None, None, self.dpred16, None, Gr("decoder_pred.bias"), Md, self.Pp)
        self._wgrad16(self.dpred16, self.hdn16, Md, self.Pp, Dd, Gw("decoder_pred.weight"))
        ops.gemm_nt(self.dpred16, self.wT16["decoder_pred.weight"], Md, Dd, self.Pp, ops.EPI_BIAS_BF16, out0=self.da["dh16"])
        ops.layernorm_bwd(self.da["dh16"], self.da["x"][-1], P("decoder_norm.weight"), self.meanD, self.rstdD, self.da["dx"],
                          Gr("decoder_norm.weight"), Gr("decoder_norm.bias"), Md, Dd, accumulate=False)
        for i in reversed(range(self.dec["depth"])):
            self._blk_bwd(self.dec, self.da, i, B, T)
        ops.mae_dec_assemble_bwd(self.da["dx"], c["ids_restore"], B, L, K, Dd, self.dyd, Gr("mask_token"))
        ops.branch_bwd(self.dyd, None, None, self.dyd16, None, Gr("decoder_embed.bias"), Me, Dd)
        self._wgrad16(self.dyd16, self.latent16, Me, Dd, D, Gw("decoder_embed.weight"))
        if self.grad_hook:
            self.grad_hook(0)
        ops.gemm_nt(self.dyd16, self.wT16["decoder_embed.weight"], Me, D, Dd, ops.EPI_BIAS_BF16, out0=self.dlat16)
        ops.layernorm_bwd(self.dlat16, self.ea["x"][-1], P("norm.weight"), self.meanE, self.rstdE, self.ea["dx"],
                          Gr("norm.weight"), Gr("norm.bias"), Me, D, accumulate=False)
        for i in reversed(range(self.enc["depth"])):
            self._blk_bwd(self.enc, self.ea, i, B, K + 1)
        ops.mae_enc_assemble_bwd(self.ea["dx"], c["ids_keep"], B, L, K, D, self.dxe, Gr("cls_token"))
        ops.branch_bwd(self.dxe, None, None, self.dxe16, None, Gr("patch_embed.proj.bias"), B * L, D)
        self._wgrad16(self.dxe16, self.patches, B * L, D, self.Kpe, Gw("patch_embed.proj.weight"))
        if self.grad_hook:
            self.grad_hook(1)
