"""ViTEngineF32 -- the `--precision fp32` parity mode of the fused ViT pipeline (csrc/fp32_path.hip).

Same parameter packing (flat fp32 buffers, reverse-layer buckets, nn.Parameter views), same entry points
(``forward`` / ``backward`` / ``grad_norm`` / ``adamw_step``) and the same reference arithmetic as ``ViTEngine``, but with fp32
operands and accumulation everywhere and NO bf16 rounding point: the reference model WITHOUT autocast
(mem/modeling_pretrain.py:97-126, mem/modeling_finetune.py:56-189, nn.CrossEntropyLoss at mem/engine_for_pretraining.py:152).
It exists for parity evidence -- loss curves against the reference's fp32 CPU run (north star: step-100 loss within 1e-4;
SURVEY.md 8d config #1: <= 1e-5 rel) -- not for speed: GEMMs run on v_mfma_f32_16x16x4_f32 at 1/16 of the bf16 rate and
none of the fast path's derived-gradient shortcuts are used (layer-scale gradient from the stored branch output, v_bias /
q_bias gradients as column sums of dqkv, delta = sum P dP).
"""
import torch

from . import ops
from .vit_engine import ViTEngine, _pad


class ViTEngineF32(ViTEngine):
    precision = "fp32"

    def __init__(self, model):
        super().__init__(model)
        assert self.head_kind == "mlm", "fp32 parity mode covers the pretraining model"
        m = model
        idx = None
        if self.rel == "shared":
            idx = m.rel_pos_bias.relative_position_index
        elif self.rel == "block":
            idx = m.blocks[0].attn.relative_position_index
        self.rel_index = None if idx is None else idx.to(self.dev, torch.int32).contiguous().view(-1)
        D, Hd, V, dev = self.D, self.hidden, self.V, self.dev
        f = torch.float32
        self.wT32 = {i: dict(qkv=torch.empty((D, 3 * D), dtype=f, device=dev), proj=torch.empty((D, D), dtype=f, device=dev),
                             fc1=torch.empty((D, Hd), dtype=f, device=dev), fc2=torch.empty((Hd, D), dtype=f, device=dev))
                     for i in range(self.depth)}
        self.wT32_lm = torch.empty((D, V), dtype=f, device=dev)

    def W32(self, name, rows, cols):
        o, k = self.segs[name]
        return self.flat_p[o:o + k].view(rows, cols)

    def sync_weights(self):
        D, Hd = self.D, self.hidden
        for i in range(self.depth):
            pre = f"blocks.{i}."
            ops.f32_transpose(self.W32(pre + "attn.qkv.weight", 3 * D, D), 3 * D, D, self.wT32[i]["qkv"])
            ops.f32_transpose(self.W32(pre + "attn.proj.weight", D, D), D, D, self.wT32[i]["proj"])
            ops.f32_transpose(self.W32(pre + "mlp.fc1.weight", Hd, D), Hd, D, self.wT32[i]["fc1"])
            ops.f32_transpose(self.W32(pre + "mlp.fc2.weight", D, Hd), D, Hd, self.wT32[i]["fc2"])
        ops.f32_transpose(self.W32("lm_head.weight", self.V, D), self.V, D, self.wT32_lm)
        self.weights_dirty = False

    def ensure_batch(self, B, Mm_max):
        if B <= self.B and Mm_max <= getattr(self, "Mm_cap", 0):
            return
        B = max(B, self.B)
        # (every patch of the batch: the masked-row count differs from step to step, and a new maximum must not re-allocate the
        # whole buffer set -- the bf16 engine's round-4 finding)
        Mm_cap = max(Mm_max, getattr(self, "Mm_cap", 0), B * self.L)
        dev, f = self.dev, torch.float32
        D, Hd, T, V = self.D, self.hidden, self.T, self.V
        M = B * T
        e = lambda *s, dt=f: torch.empty(s, dtype=dt, device=dev)   # noqa: E731
        self.patches = e(B * self.L, self.Kpe)
        self.x = [torch.zeros((M, D), dtype=f, device=dev) for _ in range(2 * self.depth + 1)]
        self.act = [dict(h1=e(M, D), qkv=e(M, 3 * D), ao=e(M, D), y1=e(M, D), h2=e(M, D), hpre=e(M, Hd), a=e(M, Hd), y2=e(M, D),
                         mean1=e(M), rstd1=e(M), mean2=e(M), rstd2=e(M)) for _ in range(self.depth)]
        self.hN, self.meanN, self.rstdN = e(Mm_cap, D), e(Mm_cap), e(Mm_cap)
        self.logits = e(Mm_cap, V)
        self.row_loss, self.row_ok = e(Mm_cap), torch.empty(Mm_cap, dtype=torch.int32, device=dev)
        self.dhN = e(Mm_cap, D)
        self.dx = torch.zeros((M, D), dtype=f, device=dev)
        self.dY, self.dh_small, self.dbig, self.dqkv, self.dao = e(M, D), e(M, D), e(M, Hd), e(M, 3 * D), e(M, D)
        self.dYpe = e(B * self.L, D)
        Rp = _pad(max(M, Mm_cap, B * self.L), 32)
        wide = max(3 * D, Hd, V, self.Kpe)
        self.tA, self.tB = e(wide, Rp), e(wide, Rp)              # transposed operands of the weight-gradient products
        self.B, self.Mm_cap = B, Mm_cap

    # ------------------------------------------------------------------ forward
    def forward(self, x, mask_u8, rows_idx, labels=None, dp_masks=None, all_tokens=False, labels_event=None):
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (self.C, self.H, self.W), \
            f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model ({self.H}*{self.W})."
        Mm = rows_idx.numel()
        self.ensure_batch(B, Mm)
        if self.weights_dirty:
            self.sync_weights()
        D, Hd, T, L, V = self.D, self.hidden, self.T, self.L, self.V
        M = B * T
        self.cur = dict(B=B, M=M, Mm=Mm, mask=mask_u8, rows=rows_idx, dp=dp_masks, labels=labels)
        P, G = self.P, ops.f32_gemm_nt
        ops.f32_im2col(x, B, self.C, self.H, self.W, self.ph, self.pw, self.patches)
        x0 = self.x[0]
        ops.fill_cls(x0, B, T, D, P("cls_token"))
        G(self.patches, self.W32("patch_embed.proj.weight", D, self.Kpe), B * L, D, self.Kpe, ops.EPI_PATCH_EMBED,
          bias=P("patch_embed.proj.bias"), vec1=P("mask_token"), resid=x0, aux=mask_u8, rows_per_sample=L, ldaux=0)
        if self.has_pos:
            x0[:M].view(B, T, D).add_(P("pos_embed").view(1, T, D))
        for i in range(self.depth):
            pre = f"blocks.{i}."
            a = self.act[i]
            blk = self.model.blocks[i]
            keep = 1.0 - blk.drop_prob
            use_dp = dp_masks is not None and blk.drop_prob > 0.0
            xin, xmid, xout = self.x[2 * i], self.x[2 * i + 1], self.x[2 * i + 2]
            g1 = P(pre + "gamma_1") if (pre + "gamma_1") in self.segs else None
            g2 = P(pre + "gamma_2") if (pre + "gamma_2") in self.segs else None
            ops.f32_layernorm_fwd(xin, P(pre + "norm1.weight"), P(pre + "norm1.bias"), a["h1"], a["mean1"], a["rstd1"], M, D)
            G(a["h1"], self.W32(pre + "attn.qkv.weight", 3 * D, D), M, 3 * D, D, ops.EPI_BIAS_BF16, out0=a["qkv"],
              bias=P(pre + "attn.qkvbias3"), colscale=self.scale, colscale_n=D)
            ops.f32_attn_fwd(a["qkv"], B, T, D, self.heads, self.table(i) if self.rel_index is not None else None,
                             self.rel_index, a["ao"])
            G(a["ao"], self.W32(pre + "attn.proj.weight", D, D), M, D, D, ops.EPI_RESIDUAL, out0=a["y1"],
              bias=P(pre + "attn.proj.bias"), vec1=g1, resid=xmid, aux=xin, ldaux=D,
              rowmask=dp_masks[2 * i] if use_dp else None, keep_prob=keep, rows_per_sample=T)
            ops.f32_layernorm_fwd(xmid, P(pre + "norm2.weight"), P(pre + "norm2.bias"), a["h2"], a["mean2"], a["rstd2"], M, D)
            G(a["h2"], self.W32(pre + "mlp.fc1.weight", Hd, D), M, Hd, D, ops.EPI_BIAS_GELU, out0=a["hpre"], out1=a["a"],
              bias=P(pre + "mlp.fc1.bias"))
            G(a["a"], self.W32(pre + "mlp.fc2.weight", D, Hd), M, D, Hd, ops.EPI_RESIDUAL, out0=a["y2"],
              bias=P(pre + "mlp.fc2.bias"), vec1=g2, resid=xout, aux=xmid, ldaux=D,
              rowmask=dp_masks[2 * i + 1] if use_dp else None, keep_prob=keep, rows_per_sample=T)
        xl = self.x[2 * self.depth]
        ops.f32_layernorm_fwd(xl, P("norm.weight"), P("norm.bias"), self.hN, self.meanN, self.rstdN, Mm, D, row_idx=rows_idx)
        G(self.hN, self.W32("lm_head.weight", V, D), Mm, V, D, ops.EPI_BIAS_BF16, out0=self.logits, bias=P("lm_head.bias"))
        if labels is not None:
            if labels_event is not None:
                torch.cuda.current_stream().wait_event(labels_event)
            ops.f32_cross_entropy(self.logits, labels, Mm, V, 1.0 / Mm, self.row_loss, self.row_ok, self.loss_acc)
        return self.logits[:Mm]

    # ------------------------------------------------------------------ backward
    def _wgrad32(self, dY, X, R, n_out, n_in, gname):
        Rp = _pad(R, 32)
        tA, tB = self.tA[:n_out, :Rp], self.tB[:n_in, :Rp]
        ops.f32_transpose(dY, R, n_out, tA)
        ops.f32_transpose(X, R, n_in, tB)
        ops.f32_gemm_nt(tA, tB, n_out, n_in, Rp, ops.EPI_F32, out0=self.G(gname).view(n_out, n_in), accumulate=True)

    def backward(self, dlogits=None):
        c = self.cur
        B, M, Mm = c["B"], c["M"], c["Mm"]
        D, Hd, T, L, V = self.D, self.hidden, self.T, self.L, self.V
        dp_masks = c["dp"]
        P, Gr, G = self.P, self.G, ops.f32_gemm_nt
        if dlogits is not None:
            self.logits[:Mm].copy_(dlogits.float())
        self.flat_g.zero_()
        dx = self.dx
        dx[:M].zero_()
        dl = self.logits
        G(dl, self.wT32_lm, Mm, D, V, ops.EPI_BIAS_BF16, out0=self.dhN)
        self._wgrad32(dl, self.hN, Mm, V, D, "lm_head.weight")
        ops.f32_colsum(dl, Mm, V, Gr("lm_head.bias"))
        ops.f32_layernorm_bwd(self.dhN, self.x[2 * self.depth], P("norm.weight"), self.meanN, self.rstdN, dx,
                              Gr("norm.weight"), Gr("norm.bias"), Mm, D, accumulate=False, row_idx=c["rows"])
        if self.grad_hook:
            self.grad_hook(0)
        for i in reversed(range(self.depth)):
            pre = f"blocks.{i}."
            a = self.act[i]
            blk = self.model.blocks[i]
            keep = 1.0 - blk.drop_prob
            use_dp = dp_masks is not None and blk.drop_prob > 0.0
            xin, xmid = self.x[2 * i], self.x[2 * i + 1]
            has_g = (pre + "gamma_1") in self.segs
            # -- MLP branch
            ops.f32_branch_bwd(dx, a["y2"], P(pre + "gamma_2") if has_g else None, self.dY,
                               Gr(pre + "gamma_2") if has_g else None, Gr(pre + "mlp.fc2.bias"), M, D,
                               rowmask=dp_masks[2 * i + 1] if use_dp else None, keep_prob=keep, rows_per_sample=T)
            G(self.dY, self.wT32[i]["fc2"], M, Hd, D, ops.EPI_DGELU, out0=self.dbig, aux=a["hpre"], colsum=Gr(pre + "mlp.fc1.bias"))
            self._wgrad32(self.dY, a["a"], M, D, Hd, pre + "mlp.fc2.weight")
            self._wgrad32(self.dbig, a["h2"], M, Hd, D, pre + "mlp.fc1.weight")
            G(self.dbig, self.wT32[i]["fc1"], M, D, Hd, ops.EPI_BIAS_BF16, out0=self.dh_small)
            ops.f32_layernorm_bwd(self.dh_small, xmid, P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx,
                                  Gr(pre + "norm2.weight"), Gr(pre + "norm2.bias"), M, D, accumulate=True)
            # -- attention branch
            ops.f32_branch_bwd(dx, a["y1"], P(pre + "gamma_1") if has_g else None, self.dY,
                               Gr(pre + "gamma_1") if has_g else None, Gr(pre + "attn.proj.bias"), M, D,
                               rowmask=dp_masks[2 * i] if use_dp else None, keep_prob=keep, rows_per_sample=T)
            G(self.dY, self.wT32[i]["proj"], M, D, D, ops.EPI_BIAS_BF16, out0=self.dao)
            self._wgrad32(self.dY, a["ao"], M, D, D, pre + "attn.proj.weight")
            ops.f32_attn_bwd(a["qkv"], self.dao, B, T, D, self.heads, self.scale,
                             self.table(i) if self.rel_index is not None else None, self.rel_index, self.dqkv, self.dtable(i))
            ops.f32_colsum(self.dqkv[:, :D], M, D, Gr(pre + "attn.q_bias"))
            ops.f32_colsum(self.dqkv[:, 2 * D:], M, D, Gr(pre + "attn.v_bias"))
            self._wgrad32(self.dqkv, a["h1"], M, 3 * D, D, pre + "attn.qkv.weight")
            G(self.dqkv, self.wT32[i]["qkv"], M, D, 3 * D, ops.EPI_BIAS_BF16, out0=self.dh_small)
            ops.f32_layernorm_bwd(self.dh_small, xin, P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx,
                                  Gr(pre + "norm1.weight"), Gr(pre + "norm1.bias"), M, D, accumulate=True)
            if self.grad_hook:
                self.grad_hook(self.depth - i)
        if self.has_pos:
            Gr("pos_embed").view(T, D).copy_(dx[:M].view(B, T, D).sum(0))
        ops.f32_embed_bwd(dx, c["mask"], B, L, D, self.dYpe, Gr("cls_token"), Gr("mask_token"))
        self._wgrad32(self.dYpe, self.patches, B * L, D, self.Kpe, "patch_embed.proj.weight")
        ops.f32_colsum(self.dYpe, B * L, D, Gr("patch_embed.proj.bias"))
        if self.grad_hook:
            self.grad_hook(self.depth + 1)

    def forward_trunk(self, *a, **k):
        raise NotImplementedError("fp32 parity mode covers the pretraining model (forward / backward)")

    backward_trunk = forward_trunk
