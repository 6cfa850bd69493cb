"""ViTEngine -- the fused HIP forward/backward pipeline behind
VisionTransformerForMaskedImageModeling (reference arithmetic: mem/modeling_pretrain.py:97-126 +
mem/modeling_finetune.py:128-189 + nn.CrossEntropyLoss, mem/engine_for_pretraining.py:152).

MI355X-first structure (not an autograd graph):
  * all parameters live in ONE flat fp32 buffer (tensors padded to 1024 elements, laid out in
    reverse-layer order so gradient buckets become ready -- and can be all-reduced over RCCL --
    in the order backward produces them); gradients and both Adam moments are flat twins.
    nn.Parameter objects are views into it, so state_dict()/checkpoints keep the reference keys.
  * bf16 shadow weights (plus [in,out]-major copies for dgrad) are re-derived from the fp32
    masters once per step by a streaming cast.
  * activations are pre-allocated per batch size (288 GB of HBM: ~1.6 GB per ViT-B block at
    B=256), every op is a hand-written HIP kernel from libmemhip.so enqueued on the current
    stream; there is no host synchronisation inside forward/backward.
  * precision policy == the reference under autocast: bf16 GEMM operands with fp32 accumulate,
    fp32 residual stream / LayerNorm statistics / softmax / loss / optimizer.
"""
import math

import torch

from . import ops

ALIGN = 1024


def _pad(n, a):
    return (n + a - 1) // a * a


class ViTEngine:
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
        self.depth = len(model.blocks)
        self.heads = model.blocks[0].attn.num_heads
        self.scale = float(model.blocks[0].attn.scale)
        self.hidden = model.blocks[0].mlp.fc1.weight.shape[0]
        # "mlm": masked-token head of the pretraining model (lm_head on the masked rows);  "cls": the finetuning
        # model (modeling_finetune.VisionTransformer) -- the engine runs the trunk, the O(B*D) pooling / fc_norm / head
        # tail stays in torch autograd
        self.head_kind = "mlm" if hasattr(model, "lm_head") else "cls"
        self.V = model.lm_head.weight.shape[0] if self.head_kind == "mlm" else 0
        self.Kpe = self.C * self.ph * self.pw
        assert self.D % 64 == 0 and self.hidden % 64 == 0 and self.Kpe % 64 == 0, "GEMM K dims must be multiples of 64"
        self.has_pos = model.pos_embed is not None
        self.window = tuple(pe.patch_shape)
        self.nrd = (2 * self.window[0] - 1) * (2 * self.window[1] - 1) + 3
        if model.rel_pos_bias is not None:
            self.rel = "shared"                      # one table for all blocks (pretraining; --disable_rel_pos_bias off)
        elif getattr(model.blocks[0].attn, "relative_position_bias_table", None) is not None:
            self.rel = "block"                       # use_rel_pos_bias: one table per block (finetuning default)
        else:
            self.rel = "none"
        self.TP = ops.attn_tokens_padded(self.T)
        self._pack_parameters()
        self._build_static()
        if hasattr(model, "register_state_dict_pre_hook"):
            model.register_state_dict_pre_hook(lambda *_a, **_k: self.wait_optimizer())
        self.B = 0
        self.step_masks = None
        self.weights_dirty = True
        self._zero_plans = {}
        self.grad_hook = None            # called as grad_hook(bucket_index) when a bucket's grads are final
        self.hook_on_side = True         # the hook runs on the weight-gradient stream: no per-layer join (_bucket_ready)
        # Weight-gradient products on a second HIP stream (backward only): every dgrad GEMM has an independent wgrad
        # GEMM beside it (both only read dY), and both are persistent one-workgroup-per-CU launches, so the CUs that a
        # launch leaves idle in its last partial round of tiles (N = 768: 591 tiles on 256 CUs) pick up workgroups of
        # the other stream's launch instead of waiting.  Results are identical (same kernels, same operands).
        # gradient accumulation (finetuning with update_freq > 1): backward ADDS into the flat gradient buffer and the
        # caller's optimizer.zero_grad() clears it (every gradient kernel accumulates; the layer-scale gradient is a
        # linear function of the accumulated weight gradient)
        self.accumulate_grads = False
        # Optimizer pipelined with the NEXT forward (round 4, OFF by default): AdamW, the bf16 cast and the transposed copies
        # run per layer bucket on their own stream in FORWARD order (embedding, block 0, ...), and the next forward waits per
        # bucket, so the ~0.7 ms of HBM-bound update work run beside the next step's first blocks instead of in front of them.
        # Bit-equal to the blocking update (tests/test_train_gpu.py) -- and SLOWER on MI355X: interleaved A/B on one box,
        # p50 35.86 / 35.73 / 35.75 ms blocking against 36.08 / 36.00 / 36.18 pipelined (tools/exp/r04_run22.sh): the update's
        # HBM traffic beside the power-limited GEMMs costs them more than the 0.7 ms it hides.  Kept as an option.
        # Every other reader of the parameters / writer of the gradients on the launch stream goes through
        # wait_optimizer() (backward, zero_grad, state_dict, sync_weights).  Pretraining engine only.
        self.overlap_optimizer = False
        self._opt_stream = None
        self._opt_ev = None               # bucket name -> event (parameters of that bucket are updated, cast)
        self._opt_done = None             # everything incl. the transposed copies
        self.wgrad_side_stream = True
        self.wgrad_group = 2             # 1: proj + qkv weight gradients of a block as one launch (_wgrad_group); 2: fc2 + fc1 too; 0: off
        self.fwd_two_streams = False      # forward: uneven two-stream split (see forward_trunk / _split_point).  It paid
                                          # -0.24 ms at B = 256 while the GEMM epilogues stalled on their own stores (the
                                          # second part filled those stalls); with the branch-free epilogues of the 256-row
                                          # kernel (whole tiles only: a split batch leaves ragged row counts to the slower
                                          # kernels) one stream is 0.2-0.5 ms faster.  Kept as an option (bench --fwd-split).
        self._side = None
        self._ev_pool, self._ev_i = [], 0
        # Stochastic depth as WORK SKIPPING (timm drop_path zeroes a dropped sample's branch output: nothing of that branch
        # has to be computed, forward or backward).  Per block and branch the kept samples are processed as a compact batch:
        # LayerNorm gathers their rows, the GEMMs / attention run on kept * T rows, the residual epilogue and the backward
        # row kernels address the residual stream through a sample map.  ~5 % of all block work at drop_path 0.1.
        self.dp_skip = True
        self.tail_rows = True             # last block: the MLP branch runs only on the rows that reach the head (forward())
        self._dp_stage = None

    # ------------------------------------------------------------------ parameter packing
    def _pack_parameters(self):
        m = self.model
        named = dict(m.named_parameters())
        skip = m.no_weight_decay()
        order = []                                    # (bucket, [names]) in reverse-layer order
        order.append(("head", ["lm_head.weight", "lm_head.bias", "head.weight", "head.bias", "fc_norm.weight", "fc_norm.bias",
                               "norm.weight", "norm.bias"]))
        for i in reversed(range(self.depth)):
            pre = f"blocks.{i}."
            names = [pre + n for n in ("mlp.fc2.weight", "mlp.fc2.bias", "gamma_2", "mlp.fc1.weight", "mlp.fc1.bias",
                                       "norm2.weight", "norm2.bias", "attn.proj.weight", "attn.proj.bias", "gamma_1",
                                       "attn.qkv.weight", "QKVBIAS", "attn.relative_position_bias_table", "norm1.weight",
                                       "norm1.bias")]
            order.append((f"block{i}", names))
        order.append(("embed", ["rel_pos_bias.relative_position_bias_table", "patch_embed.proj.weight",
                                "patch_embed.proj.bias", "mask_token", "cls_token", "pos_embed"]))
        segs, off = {}, 0
        buckets, flags = [], []
        for bname, names in order:
            b0 = off
            for n in names:
                if n.endswith("QKVBIAS"):
                    pre = n[: -len("QKVBIAS")]
                    D = self.D
                    segs[pre + "attn.q_bias"] = (off, D)
                    segs[pre + "attn.v_bias"] = (off + 2 * D, D)
                    segs[pre + "attn.qkvbias3"] = (off, 3 * D)          # [q_bias | 0 | v_bias]
                    size = _pad(3 * D, ALIGN)
                    flags += [0] * (size // ALIGN)
                    off += size
                    continue
                if n not in named:
                    continue                                           # gamma_* absent when layer scale is off
                p = named[n]
                segs[n] = (off, p.numel())
                size = _pad(p.numel(), ALIGN)
                decay = not (p.ndim == 1 or n.endswith(".bias") or n in skip)   # optim_factory.py:63
                flags += [1 if decay else 0] * (size // ALIGN)
                off += size
            buckets.append((bname, b0, off))
        missing = [n for n in named if n not in segs]
        assert not missing, f"parameters not placed in the flat buffer: {missing}"
        self.nflat = off
        self.segs, self.buckets = segs, buckets
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.flat_w16 = torch.zeros(off, dtype=torch.bfloat16, device=self.dev)
        self.wd_flags = torch.tensor(flags, dtype=torch.uint8, device=self.dev)
        self.decay_names = [n for n, p in named.items() if not (p.ndim == 1 or n.endswith(".bias") or n in skip)]
        self.no_decay_names = [n for n in named if n not in set(self.decay_names)]
        for n, p in named.items():
            o, k = segs[n]
            view = self.flat_p[o:o + k].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + k].view(p.shape)
        self.named = named

    def attach_grads(self):
        """Re-point p.grad at the flat gradient buffer (zero_grad(set_to_none=True) drops them)."""
        for n, p in self.named.items():
            o, k = self.segs[n]
            g = p.grad
            if g is None or g.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + k].view(p.shape)

    def P(self, name):           # fp32 master view
        o, k = self.segs[name]
        return self.flat_p[o:o + k]

    def G(self, name):           # fp32 gradient view
        o, k = self.segs[name]
        return self.flat_g[o:o + k]

    def W16(self, name, rows, cols):   # bf16 shadow as a [rows, cols] matrix
        o, k = self.segs[name]
        return self.flat_w16[o:o + k].view(rows, cols)

    def _build_static(self):
        D, Hd, V, dev = self.D, self.hidden, self.V, self.dev
        bf = torch.bfloat16
        self.wT = {}
        for i in range(self.depth):
            self.wT[i] = dict(qkv=torch.empty((D, 3 * D), dtype=bf, device=dev),
                              proj=torch.empty((D, D), dtype=bf, device=dev),
                              fc1=torch.empty((D, Hd), dtype=bf, device=dev),
                              fc2=torch.empty((Hd, D), dtype=bf, device=dev))
        self.wT_lm = torch.empty((D, V), dtype=bf, device=dev) if self.head_kind == "mlm" else None
        self.zero_table = torch.zeros((self.nrd, self.heads), dtype=torch.float32, device=dev)   # rel == "none"
        self.zero_vec = torch.zeros(D, dtype=torch.float32, device=dev)                           # no mask_token
        self.head_end = self.buckets[0][2]           # flat offset where the head bucket ends
        self._tdesc = None
        # gelu_dg (default since round 4): fc1 keeps gelu'(h) for the backward instead of h (EPI_BIAS_GELU_DG / EPI_MUL_AUX):
        # erf / exp are evaluated once, in the forward epilogue, and the GELU backward is a plain product.  gelu' is stored as
        # FP16 (same 16 bits per value; gelu' lies in [-0.13, 1.13], so fp16's 11 significant bits apply): the reference
        # evaluates gelu' in fp32 from the stored bf16 pre-activation, so the product differs by a relative 2^-11 in front of
        # its bf16 rounding (~6 % of the elements move by one bf16 ulp, unbiased; with a bf16 gelu' -- round 2's form, left
        # off for that reason -- it was half of them).  -0.5 ms per step; set_gelu_dg(False) = the reference's placement.
        self.set_gelu_dg(True)
        self.fuse_ln_branch = True       # LayerNorm backward fused with the following branch backward (D <= 1024)
        self.gn_ws = torch.zeros(1024, dtype=torch.float64, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_acc = torch.zeros(2, dtype=torch.float32, device=dev)

    # ------------------------------------------------------------------ zero fills (no torch fill kernels inside a step)
    def _matrix_grad_names(self):
        names = [f"blocks.{i}.{n}" for i in range(self.depth) for n in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight",
                                                                       "mlp.fc2.weight")]
        names.append("patch_embed.proj.weight")
        if self.head_kind == "mlm":
            names.append("lm_head.weight")
        return names

    def _zero_small_grads(self, start=0):
        """Zero every gradient of flat_g[start:] that is ACCUMULATED (atomics: biases, LayerNorm affines, bucket tables,
        tokens, layer scales) -- everything but the weight matrices, which their weight-gradient GEMM writes.  One launch
        over a static range table: 1.7 of 344 MB for ViT-B (the reference: optimizer.zero_grad() on every tensor)."""
        key = ("zr", start)
        if key not in self._zero_plans:
            import numpy as np
            mats = sorted(self.segs[n] for n in self._matrix_grad_names())
            ranges, pos = [], start
            for o, k in mats:
                end = o + _pad(k, ALIGN)
                if end <= start:
                    continue
                if o > pos:
                    ranges.append((pos * 4, (o - pos) * 4))
                pos = max(pos, end)
            if pos < self.nflat:
                ranges.append((pos * 4, (self.nflat - pos) * 4))
            arr = np.asarray(ranges, dtype=np.int64).reshape(-1, 2)
            self._zero_plans[key] = (torch.from_numpy(arr).to(self.dev), len(ranges), int(arr[:, 1].sum()) if len(ranges) else 0)
        dev, n, total = self._zero_plans[key]
        ops.zero_ranges(self.flat_g, dev, n, total)

    def _zero_grad_of(self, name):
        o, k = self.segs[name]
        ops.zero_(self.flat_g[o:o + _pad(k, ALIGN)])

    def set_gelu_dg(self, on):
        self.epi_gelu, self.epi_dgelu = (ops.EPI_BIAS_GELU_DG, ops.EPI_MUL_AUX) if on else (ops.EPI_BIAS_GELU, ops.EPI_DGELU)

    # ------------------------------------------------------------------ buffers per batch size
    def ensure_batch(self, B, Mm_max):
        """Buffers for a batch of B samples with Mm_max rows reaching the head.  The two grow independently: the number of
        masked rows differs from step to step (block-wise masks), and a new maximum must not re-allocate (and zero-fill) the
        ~100 per-block buffers of the batch -- found in the round-4 step trace as ~80 torch fill kernels inside a step."""
        if B > self.B:
            self._alloc_batch(B)
        if Mm_max > getattr(self, "Mm_cap", 0) or (self.head_kind == "mlm" and not hasattr(self, "hN")):
            # headroom: a few per cent above the largest count seen, whole 256-row tiles, never more than every patch
            self._alloc_head(min(self.B * self.L, _pad(Mm_max + Mm_max // 16 + 1, 256)))

    def _alloc_head(self, Mm_cap):
        dev, bf, f32 = self.dev, torch.bfloat16, torch.float32
        D, V = self.D, self.V
        e = lambda *s, dt=bf: torch.empty(s, dtype=dt, device=dev)   # noqa: E731
        if self.head_kind == "mlm":
            self.hN = e(Mm_cap, D)
            self.meanN, self.rstdN = e(Mm_cap, dt=f32), e(Mm_cap, dt=f32)
            self.logits = e(Mm_cap, V)
            self.row_loss, self.row_ok = e(Mm_cap, dt=f32), torch.empty(Mm_cap, dtype=torch.int32, device=dev)
            self.dhN = e(Mm_cap, D)
            # dead-row elimination in the last block (tail rows): compact residual rows that reach the head, their
            # gradient, and the bf16 fc2 output of the compact MLP branch (whole 256-row tiles)
            self.x_tail, self.dxc = e(max(Mm_cap, 1), D, dt=f32), e(max(Mm_cap, 1), D, dt=f32)
            self.y_tail = torch.zeros((_pad(max(Mm_cap, 1), 256), D), dtype=bf, device=dev)
        self.Mm_cap = Mm_cap

    def _alloc_batch(self, B):
        dev, bf, f32 = self.dev, torch.bfloat16, torch.float32
        D, Hd, T = self.D, self.hidden, self.T
        M = B * T
        # Work-skipping stochastic depth runs the GEMMs of a branch on its kept samples' rows ROUNDED UP to whole 256-row
        # tiles (a ragged row count costs every product an extra launch on the 128-row kernel): the token-major buffers
        # carry 256 rows of slack (zero-initialised, only ever finite), and the residual snapshots one dummy sample that the
        # padded rows' residual updates go to (sample index B in the sample map)
        Ma = M + 256
        e = lambda *s, dt=bf: torch.empty(s, dtype=dt, device=dev)   # noqa: E731
        z = lambda *s, dt=bf: torch.zeros(s, dtype=dt, device=dev)   # noqa: E731
        self.patches = e(B * self.L, self.Kpe)
        self.x = [torch.zeros((M + T, D), dtype=f32, device=dev) for _ in range(2 * self.depth + 1)]
        self.act = []
        for _ in range(self.depth):
            self.act.append(dict(h1=z(Ma, D), qkv=z(Ma, 3 * D), ao=z(Ma, D), h2=z(Ma, D), hpre=z(Ma, Hd),
                                 a=z(Ma, Hd), lse=e(B, self.heads, self.TP, dt=f32),
                                 mean1=e(M, dt=f32), rstd1=e(M, dt=f32), mean2=e(M, dt=f32), rstd2=e(M, dt=f32)))
        if self.head_kind != "mlm":
            self.zero_mask = torch.zeros(B * self.L, dtype=torch.uint8, device=dev)
        # backward temporaries (shared by all blocks)
        self.dx = torch.zeros((M, D), dtype=f32, device=dev)
        self.dY, self.dh_small = z(Ma, D), z(Ma, D)
        self.dY2 = z(Ma, D)                                   # attention-branch twin of dY (the side stream reads both)
        self.dbig = z(Ma, Hd)
        self.dqkv = z(Ma, 3 * D)
        self.dao = z(Ma, D)
        self.delta_ws = e(2 * M + 4, self.heads, dt=f32)   # rowsum(dO*O), |dO|^2, 4 rows of per-head bounds
        self._tn_ws = torch.empty(0, dtype=torch.uint8, device=dev)     # partial tiles of the weight-gradient GEMMs
        self.bias_scr = torch.zeros(2, D, dtype=f32, device=dev)   # ping-pong colsum(dY) of the proj branch
        self.cs_ws = torch.zeros(self.CS_COPIES, self.hidden, dtype=f32, device=dev)   # column-sum accumulator copies (zero between uses)
        self.dYpe = e(B * self.L, D)
        self.B = B

    # ------------------------------------------------------------------ weights
    def sync_weights(self):
        """fp32 masters -> bf16 shadows (+ [in,out]-major copies for the dgrad GEMMs)."""
        self.wait_optimizer()
        ops.cast_f32_bf16(self.flat_p, self.flat_w16, self.nflat)
        if self._tdesc is None:
            self._build_transpose_descs()
        ops.transpose_cast_batched(self._tdesc, self._tprefix, self._tn, self._ttiles)
        self.weights_dirty = False

    def _build_transpose_descs(self):
        """Descriptors of the [in,out]-major bf16 weight copies (one batched launch per step, memhip_transpose_cast_batched)."""
        import numpy as np
        D, Hd = self.D, self.hidden
        items = []
        for i in range(self.depth):
            pre = f"blocks.{i}."
            items += [(self.P(pre + "attn.qkv.weight"), 3 * D, D, self.wT[i]["qkv"]),
                      (self.P(pre + "attn.proj.weight"), D, D, self.wT[i]["proj"]),
                      (self.P(pre + "mlp.fc1.weight"), Hd, D, self.wT[i]["fc1"]),
                      (self.P(pre + "mlp.fc2.weight"), D, Hd, self.wT[i]["fc2"])]
        if self.head_kind == "mlm":
            items.append((self.P("lm_head.weight"), self.V, D, self.wT_lm))
        desc = np.zeros((len(items), 6), dtype=np.int64)
        prefix = np.zeros(len(items) + 1, dtype=np.int32)
        for k, (src, R, Cc, dst) in enumerate(items):
            desc[k] = (src.data_ptr(), Cc, R, Cc, dst.data_ptr(), dst.stride(0))
            prefix[k + 1] = prefix[k] + ((R + 63) // 64) * ((Cc + 63) // 64)
        self._tdesc = torch.from_numpy(desc).to(self.dev)
        self._tprefix = torch.from_numpy(prefix).to(self.dev)
        self._tn, self._ttiles = len(items), int(prefix[-1])
        # the same per layer bucket (the pipelined optimizer transposes a bucket's matrices as soon as they are updated):
        # bucket -> (first item, count, local prefix, tiles)
        self._tbucket = {}
        locs = []
        for i in range(self.depth):
            locs.append((f"block{i}", 4 * i, 4))
        if self.head_kind == "mlm":
            locs.append(("head", 4 * self.depth, 1))
        loc_prefix = []
        for name, k0, n in locs:
            pl = (prefix[k0:k0 + n + 1] - prefix[k0]).astype(np.int32)
            self._tbucket[name] = (k0, n, len(loc_prefix), int(pl[-1]))
            loc_prefix += list(pl)
        self._tprefix_loc = torch.from_numpy(np.asarray(loc_prefix, dtype=np.int32)).to(self.dev)

    def table(self, i):
        """Relative-position bucket table of block i ([nrd, heads] fp32 master)."""
        if self.rel == "shared":
            return self.P("rel_pos_bias.relative_position_bias_table")
        if self.rel == "block":
            return self.P(f"blocks.{i}.attn.relative_position_bias_table")
        return self.zero_table

    def dtable(self, i):
        if self.rel == "shared":
            return self.G("rel_pos_bias.relative_position_bias_table")
        if self.rel == "block":
            return self.G(f"blocks.{i}.attn.relative_position_bias_table")
        return None

    # ------------------------------------------------------------------ stochastic depth: per-step work plan
    CS_COPIES = 8        # accumulator copies of a fused GEMM column sum (ops.gemm_nt colsum_copies): one per XCD

    def _dp_plan(self, dp_masks, B):
        """dp_masks [2*depth, B] (0/1; host tensor / ndarray preferred, a device tensor costs one sync) -> plan dict or
        None.  Per row j of the mask with a drop probability: the kept samples (count on the host: it sizes the launches),
        on the device kidx[j] (compact -> sample, padded with zeros for the GEMM epilogue's read-ahead), cmap[j]
        (sample -> compact or -1), the dropped samples, and ridx[j] (compact row -> row of the residual stream)."""
        import numpy as np
        if dp_masks is None or not self.dp_skip:
            return None
        probs = [float(b.drop_prob) for b in self.model.blocks for _ in range(2)]
        if max(probs) == 0.0:
            return None
        mk = dp_masks.detach().cpu().numpy() if isinstance(dp_masks, torch.Tensor) else np.asarray(dp_masks)
        mk = mk.reshape(2 * self.depth, B) != 0
        J, T = 2 * self.depth, self.T
        W = 3 * B + 256                                      # per row: kidx [B + 256] | cmap [B] | dropped [B]
        host = np.zeros((J, W), dtype=np.int32)
        host[:, B + 256:2 * B + 256] = -1
        kept_n = []
        for j in range(J):
            if probs[j] == 0.0:
                kept_n.append(None)                          # branch never drops: plain path
                continue
            k = np.flatnonzero(mk[j]).astype(np.int32)
            d = np.flatnonzero(~mk[j]).astype(np.int32)
            host[j, :len(k)] = k
            host[j, len(k):B + 256] = B                       # rows of the padded tile rows: the dummy sample
            host[j, B + 256 + k] = np.arange(len(k), dtype=np.int32)
            host[j, 2 * B + 256:2 * B + 256 + len(d)] = d
            kept_n.append(len(k))
        if self._dp_stage is None or self._dp_stage_w != J * W:
            from .utils import HostStager
            self._dp_stage, self._dp_stage_w = HostStager(J * W * 4, self.dev), J * W
            self._dp_ar = torch.arange(T, dtype=torch.int32, device=self.dev)
        dev = self._dp_stage.put(host).view(torch.int32).view(J, W)
        # compact row -> residual-stream row, all branches in one launch (rows past kept * T are never read)
        ridx = (dev[:, :B, None] * T + self._dp_ar[None, None, :]).view(J, B * T)
        return dict(n=kept_n, kidx=dev[:, :B + 256], cmap=dev[:, B + 256:2 * B + 256], drop=dev[:, 2 * B + 256:], ridx=ridx, B=B)

    def _copy_dropped(self, plan, j, src, dst, B):
        """Rows of the samples branch j dropped pass through unchanged: dst[sample] = src[sample]."""
        nd = B - plan["n"][j]
        if nd > 0:
            ops.copy_samples(src, dst, plan["drop"][j], nd, self.T * self.D)

    # ------------------------------------------------------------------ forward
    def forward(self, x, mask_u8, rows_idx, labels=None, dp_masks=None, all_tokens=False, labels_event=None):
        """x f32 [B,C,H,W]; mask_u8 u8 [B*L]; rows_idx i32 [Mm] (token rows b*T+1+p of the masked
        patches, or of ALL patches when all_tokens); labels i64 [Mm] or None (labels_event: a HIP event after which
        `labels` is valid -- the tokenizer may still be running on another stream while the trunk executes).
        dp_masks: f32 [2*depth, B] stochastic-depth keep masks (0/1) or None.
        Leaves logits (bf16 [Mm,V]) in self.logits; with labels also loss/acc in self.loss_acc and
        dlogits (in place of the logits)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (self.C, self.H, self.W), \
            f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model ({self.H}*{self.W})."
        Mm = rows_idx.numel()
        self.ensure_batch(B, Mm)
        # Dead-row elimination (tail rows): nothing behind the last block reads a row that is not in rows_idx, and no
        # gradient flows into one, so the last block's MLP branch -- token-wise work -- runs on those rows only (98 of 197
        # tokens per sample in pretraining: half of 64 % of a block).  The reference computes every row
        # (mem/modeling_pretrain.py:110-126); loss and gradients are the same (logits bit-identical).
        use_tail = (self.tail_rows and not all_tokens and self.depth >= 1 and 0 < Mm and 4 * Mm <= 3 * B * self.T)
        xl = self.forward_trunk(x, mask_u8, dp_masks, tail_rows=rows_idx if use_tail else None)
        D, Hd, T, L, V = self.D, self.hidden, self.T, self.L, self.V
        M = B * T
        self.cur.update(Mm=Mm, rows=rows_idx, labels=labels)
        self._wait_params("head")
        # final norm on exactly the rows that reach the head (x[:,1:][bool_masked_pos])
        if self.cur.get("tail") is not None:
            ops.layernorm_fwd(self.x_tail, self.P("norm.weight"), self.P("norm.bias"), self.hN, self.meanN, self.rstdN, Mm, D)
        else:
            ops.layernorm_fwd(xl, self.P("norm.weight"), self.P("norm.bias"), self.hN, self.meanN, self.rstdN, Mm, D,
                              row_idx=rows_idx)
        ops.gemm_nt(self.hN, self.W16("lm_head.weight", V, D), Mm, V, D, ops.EPI_BIAS_BF16, out0=self.logits,
                    bias=self.P("lm_head.bias"))
        if labels is not None:
            if labels_event is not None:
                torch.cuda.current_stream().wait_event(labels_event)
            ops.cross_entropy(self.logits, labels, Mm, V, 1.0 / Mm, self.row_loss, self.row_ok, self.loss_acc,
                              write_grad=True)
        return self.logits[:Mm]

    def _split_point(self, B, num_cu=256):
        """Sample count of the BIG part of an uneven two-stream split: the largest part whose N = D products (D / 256
        tile columns, the coarsest launches of a block) fill whole rounds of 256-row tiles on the CUs; the remaining
        samples run beside it on the second stream and fill what is left.  ViT-B at B = 256: 220 + 36 samples
        (170 tile rows x {3, 9, 12} tile columns = 510 / 1530 / 2040 tiles = 1.99 / 5.98 / 7.97 rounds)."""
        ntn = max(1, self.D // 256)
        tile_rows = -(-B * self.T // 256)
        rounds = tile_rows * ntn // num_cu
        if rounds < 1:
            return 0
        bs = (rounds * num_cu // ntn) * 256 // self.T
        return bs if 0 < bs < B and (B - bs) * self.T >= 4096 else 0

    def _block_fwd(self, i, b0, b1, dp_masks):
        """Block i on the samples [b0, b1) (mem/modeling_finetune.py:160-189)."""
        D, Hd, T = self.D, self.hidden, self.T
        r0, r1 = b0 * T, b1 * T
        M, Bs = r1 - r0, b1 - b0
        pre = f"blocks.{i}."
        a = self.act[i]
        table = self.table(i)
        blk = self.model.blocks[i]
        keep = 1.0 - blk.drop_prob
        use_dp = dp_masks is not None and blk.drop_prob > 0.0
        xin, xmid, xout = self.x[2 * i][r0:r1], self.x[2 * i + 1][r0:r1], self.x[2 * i + 2][r0:r1]
        g1 = self.P(pre + "gamma_1") if (pre + "gamma_1") in self.segs else None
        g2 = self.P(pre + "gamma_2") if (pre + "gamma_2") in self.segs else None
        h1, qkv, ao, h2, hpre, aa = (a[k][r0:r1] for k in ("h1", "qkv", "ao", "h2", "hpre", "a"))
        plan = self.cur.get("plan")
        tail = self.cur.get("tail") if (i == self.depth - 1 and b0 == 0 and b1 == self.cur["B"]) else None
        if plan is not None and (plan["n"][2 * i] is not None or plan["n"][2 * i + 1] is not None):
            assert b0 == 0 and b1 == plan["B"], "stochastic-depth work skipping runs the whole batch on one stream"
            self._block_fwd_skip(i, plan, a, xin, xmid, xout, g1, g2, table, keep, tail=tail)
            return
        ops.layernorm_fwd(xin, self.P(pre + "norm1.weight"), self.P(pre + "norm1.bias"), h1, a["mean1"][r0:r1],
                          a["rstd1"][r0:r1], M, D)
        ops.gemm_nt(h1, self.W16(pre + "attn.qkv.weight", 3 * D, D), M, 3 * D, D, ops.EPI_BIAS_BF16,
                    out0=qkv, bias=self.P(pre + "attn.qkvbias3"), colscale=self.scale, colscale_n=D)
        ops.attn_fwd(qkv, Bs, T, D, self.heads, table, self.window, ao, a["lse"][b0:b1])
        ops.gemm_nt(ao, self.W16(pre + "attn.proj.weight", D, D), M, D, D, ops.EPI_RESIDUAL, out0=None,
                    bias=self.P(pre + "attn.proj.bias"), vec1=g1, resid=xmid, aux=xin, ldaux=D,
                    rowmask=dp_masks[2 * i][b0:b1] if use_dp else None, keep_prob=keep, rows_per_sample=T)
        if tail is not None:
            self._mlp_fwd_tail(i, a, self.x[2 * i + 1], g2, keep, tail)
            return
        ops.layernorm_fwd(xmid, self.P(pre + "norm2.weight"), self.P(pre + "norm2.bias"), h2, a["mean2"][r0:r1],
                          a["rstd2"][r0:r1], M, D)
        ops.gemm_nt(h2, self.W16(pre + "mlp.fc1.weight", Hd, D), M, Hd, D, self.epi_gelu, out0=hpre,
                    out1=aa, bias=self.P(pre + "mlp.fc1.bias"))
        ops.gemm_nt(aa, self.W16(pre + "mlp.fc2.weight", D, Hd), M, D, Hd, ops.EPI_RESIDUAL, out0=None,
                    bias=self.P(pre + "mlp.fc2.bias"), vec1=g2, resid=xout, aux=xmid, ldaux=D,
                    rowmask=dp_masks[2 * i + 1][b0:b1] if use_dp else None, keep_prob=keep, rows_per_sample=T)

    def _mlp_fwd_tail(self, i, a, xmid, g2, keep, tail):
        """MLP branch of the LAST block on the rows that reach the head only (forward(): tail rows): norm2 on the gathered
        rows, fc1 / fc2 on the compact rows (whole 256-row tiles; rows past Mm hold finite leftovers and are never used),
        then x_tail[i] = x_mid[rows[i]] + drop_path(gamma_2 * y[i]) -- the residual epilogue's arithmetic, per compact row."""
        D, Hd = self.D, self.hidden
        pre = f"blocks.{i}."
        rows, Mm = tail["rows"], tail["Mm"]
        Mp = _pad(Mm, 256)
        ops.layernorm_fwd(xmid, self.P(pre + "norm2.weight"), self.P(pre + "norm2.bias"), a["h2"], a["mean2"], a["rstd2"],
                          Mm, D, row_idx=rows)
        ops.gemm_nt(a["h2"], self.W16(pre + "mlp.fc1.weight", Hd, D), Mp, Hd, D, self.epi_gelu, out0=a["hpre"],
                    out1=a["a"], bias=self.P(pre + "mlp.fc1.bias"))
        ops.gemm_nt(a["a"], self.W16(pre + "mlp.fc2.weight", D, Hd), Mp, D, Hd, ops.EPI_BIAS_BF16, out0=self.y_tail,
                    bias=self.P(pre + "mlp.fc2.bias"))
        ops.residual_rows(xmid, rows, self.y_tail, g2, tail["rowkeep"], keep, Mm, D, self.x_tail)

    def _block_fwd_skip(self, i, plan, a, xin, xmid, xout, g1, g2, table, keep, tail=None):
        """Block i with the dropped samples of each branch skipped (see dp_skip): compact activations."""
        D, Hd, T, B = self.D, self.hidden, self.T, plan["B"]
        pre = f"blocks.{i}."
        ja, jm = 2 * i, 2 * i + 1
        na = B if plan["n"][ja] is None else plan["n"][ja]
        nm = B if plan["n"][jm] is None else plan["n"][jm]
        # -- attention branch on the na kept samples
        if plan["n"][ja] is not None:
            self._copy_dropped(plan, ja, xin, xmid, B)
        if na > 0:
            M1 = na * T
            ridx = plan["ridx"][ja] if plan["n"][ja] is not None else None
            smap = plan["kidx"][ja] if plan["n"][ja] is not None else None
            M1p = _pad(M1, 256) if smap is not None else M1       # GEMM rows: whole 256-row tiles (see ensure_batch)
            ops.layernorm_fwd(xin, self.P(pre + "norm1.weight"), self.P(pre + "norm1.bias"), a["h1"], a["mean1"], a["rstd1"],
                              M1, D, row_idx=ridx)
            ops.gemm_nt(a["h1"], self.W16(pre + "attn.qkv.weight", 3 * D, D), M1p, 3 * D, D, ops.EPI_BIAS_BF16,
                        out0=a["qkv"], bias=self.P(pre + "attn.qkvbias3"), colscale=self.scale, colscale_n=D)
            ops.attn_fwd(a["qkv"], na, T, D, self.heads, table, self.window, a["ao"], a["lse"])
            ops.gemm_nt(a["ao"], self.W16(pre + "attn.proj.weight", D, D), M1p, D, D, ops.EPI_RESIDUAL, out0=None,
                        bias=self.P(pre + "attn.proj.bias"), vec1=g1, resid=xmid, aux=xin, ldaux=D,
                        keep_prob=keep if smap is not None else 1.0, rows_per_sample=T, sample_map=smap)
        if tail is not None:                                  # last block: the rows that reach the head, every sample
            self._mlp_fwd_tail(i, a, self.x[2 * i + 1], g2, keep, tail)      # (dropped samples: keep flag 0 per row)
            return
        # -- MLP branch on the nm kept samples
        if plan["n"][jm] is not None:
            self._copy_dropped(plan, jm, xmid, xout, B)
        if nm > 0:
            M2 = nm * T
            ridx = plan["ridx"][jm] if plan["n"][jm] is not None else None
            smap = plan["kidx"][jm] if plan["n"][jm] is not None else None
            M2p = _pad(M2, 256) if smap is not None else M2
            ops.layernorm_fwd(xmid, self.P(pre + "norm2.weight"), self.P(pre + "norm2.bias"), a["h2"], a["mean2"], a["rstd2"],
                              M2, D, row_idx=ridx)
            ops.gemm_nt(a["h2"], self.W16(pre + "mlp.fc1.weight", Hd, D), M2p, Hd, D, self.epi_gelu, out0=a["hpre"],
                        out1=a["a"], bias=self.P(pre + "mlp.fc1.bias"))
            ops.gemm_nt(a["a"], self.W16(pre + "mlp.fc2.weight", D, Hd), M2p, D, Hd, ops.EPI_RESIDUAL, out0=None,
                        bias=self.P(pre + "mlp.fc2.bias"), vec1=g2, resid=xout, aux=xmid, ldaux=D,
                        keep_prob=keep if smap is not None else 1.0, rows_per_sample=T, sample_map=smap)

    def forward_trunk(self, x, mask_u8=None, dp_masks=None, tail_rows=None):
        """Patch embedding (+ mask-token blend, + abs. position embedding) and the blocks: x f32 [B,C,H,W] ->
        the fp32 residual stream after the last block, [B*T, D] (engine-owned, valid until the next forward)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (self.C, self.H, self.W), \
            f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model ({self.H}*{self.W})."
        self.ensure_batch(B, getattr(self, "Mm_cap", 0))
        if self.weights_dirty:
            self.sync_weights()
        if self.fwd_two_streams:
            self.wait_optimizer()
        self._wait_params("embed")
        if mask_u8 is None:
            mask_u8 = self.zero_mask[: B * self.L]
        D, Hd, T, L = self.D, self.hidden, self.T, self.L
        M = B * T
        plan = self._dp_plan(dp_masks, B)
        if plan is not None:
            dp_masks = None                                  # the plan replaces the keep masks everywhere below
        self.cur = dict(B=B, M=M, Mm=0, mask=mask_u8, rows=None, dp=dp_masks, labels=None, plan=plan, tail=None)
        if tail_rows is not None and not self.fwd_two_streams:
            # per compact row: did the last block's MLP branch keep the row's sample?  (stochastic depth of that branch is
            # applied per row in the tail form, in both the masked and the work-skipping mode)
            jm, lastb = 2 * self.depth - 1, self.model.blocks[self.depth - 1]
            keepvec = None
            if lastb.drop_prob > 0.0:
                if plan is not None and plan["n"][jm] is not None:
                    keepvec = (plan["cmap"][jm] >= 0).to(torch.float32)
                elif dp_masks is not None:
                    keepvec = dp_masks[jm]
            rowkeep = None
            if keepvec is not None:
                rowkeep = keepvec.index_select(0, torch.div(tail_rows, T, rounding_mode="floor").long()).contiguous()
            self.cur["tail"] = dict(rows=tail_rows, Mm=tail_rows.numel(), rowkeep=rowkeep)
        if self.head_kind == "cls" and not self.accumulate_grads:
            ops.zero_(self.flat_g[: self.head_end])     # the torch tail accumulates its gradients here before backward_trunk
        ops.im2col(x, B, self.C, self.H, self.W, self.ph, self.pw, self.patches)
        x0 = self.x[0]
        ops.fill_cls(x0, B, T, D, self.P("cls_token"))
        ops.gemm_nt(self.patches, self.W16("patch_embed.proj.weight", D, self.Kpe), B * L, D, self.Kpe,
                    ops.EPI_PATCH_EMBED, bias=self.P("patch_embed.proj.bias"),
                    vec1=self.P("mask_token") if "mask_token" in self.segs else self.zero_vec,
                    resid=x0, aux=mask_u8, rows_per_sample=L, ldaux=0)
        if self.has_pos:
            x0[:M].view(B, T, D).add_(self.P("pos_embed").view(1, T, D))
        # Two sample-halves on two HIP streams (forward ops are independent per sample): the halves' persistent GEMM
        # launches interleave on the CUs, so the workgroups of one launch fill the partial last round of the other and
        # the HBM-bound epilogue phase of one half runs beside the MFMA-bound main loop of the other.  Same kernels on
        # the same rows: results are identical to the single-stream order.
        bs = self._split_point(B) if (self.fwd_two_streams and ops.GEMM_TIMER is None and plan is None) else 0
        split = 0 < bs < B
        if split:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            e0 = torch.cuda.Event()
            e0.record()
            self._side.wait_event(e0)
            for i in range(self.depth):
                self._block_fwd(i, 0, bs, dp_masks)
                with torch.cuda.stream(self._side):
                    self._block_fwd(i, bs, B, dp_masks)
            e1 = torch.cuda.Event()
            e1.record(self._side)
            torch.cuda.current_stream().wait_event(e1)
        else:
            for i in range(self.depth):
                self._wait_params(f"block{i}")
                self._block_fwd(i, 0, B, dp_masks)
        return self.x[2 * self.depth][:M]

    # ------------------------------------------------------------------ backward
    def _wgrad(self, dY, X, R, n_out, n_in, gname, bias_grads=()):
        """grad[gname] [n_out, n_in] += dY[R, n_out]^T @ X[R, n_in]; bias_grads = ((grad_view, c0, c1), ...)
        column sums of dY[:, c0:c1] (the Linear bias gradients)."""
        need = ops.gemm_tn_workspace(R, n_out, n_in)
        if need > self._tn_ws.numel():                      # grows to the largest product once
            self._tn_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        # (the weight matrices are WRITTEN by their one weight-gradient product per backward: they are not part of the
        # zero fill in front of backward, see _zero_small_grads; gradient accumulation adds instead)
        ops.gemm_tn(dY, X, R, n_out, n_in, self.G(gname).view(n_out, n_in), accumulate=self.accumulate_grads,
                    workspace=self._tn_ws)
        for gv, c0, c1 in bias_grads:
            ops.colsum_bf16(dY[:, c0:c1], R, c1 - c0, gv)

    def _wg_mlp(self, pre, a, has_g, R, dY):
        """fc2 and fc1 weight gradients of a block (+ the layer-scale gradient, a linear function of the fc2 one)."""
        D, Hd = self.D, self.hidden
        both = int(self.wgrad_group) >= 2
        if both:
            self._wgrad_group([(dY, a["a"], R, D, Hd, pre + "mlp.fc2.weight"), (self.dbig, a["h2"], R, Hd, D, pre + "mlp.fc1.weight")])
        else:
            self._wgrad(dY, a["a"], R, D, Hd, pre + "mlp.fc2.weight")
        self._side_read_done("dY")
        if has_g:
            # layer-scale gradient from the weight gradient (no branch output y kept): memhip_layerscale_grad
            ops.layerscale_grad(self.W16(pre + "mlp.fc2.weight", D, Hd), self.G(pre + "mlp.fc2.weight").view(D, Hd),
                                self.P(pre + "mlp.fc2.bias"), self.G(pre + "mlp.fc2.bias"),
                                self.P(pre + "gamma_2"), D, Hd, self.G(pre + "gamma_2"))
        if not both:
            self._wgrad(self.dbig, a["h2"], R, Hd, D, pre + "mlp.fc1.weight")
        self._side_read_done("dbig")

    def _wgrad_group(self, items):
        """items = [(dY, X, R, n_out, n_in, gname), ...]: the weight gradients of layers whose operands are ready at the
        same time as ONE launch (ops.gemm_tn_group): the 768 x 768 proj gradient rides with the 7 row slices of the qkv
        gradient instead of the 28 it needs alone to fill the chip (proj + qkv: 241 -> 212 us per block); fc2 + fc1 run as
        ONE round of 216 workgroups with 3 row slices each instead of two launches with 7 (as two rounds of 504 they were
        no faster than two launches).  ViT-B step: 34.8 (single launches) -> 34.3 (proj + qkv) -> 33.9 ms (both pairs)."""
        need = ops.gemm_tn_group_workspace([(R, n_out, n_in) for _, _, R, n_out, n_in, _ in items])
        if need > self._tn_ws.numel():
            self._tn_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        ops.gemm_tn_group([(dY, X, R, n_out, n_in, self.G(gname).view(n_out, n_in)) for dY, X, R, n_out, n_in, gname in items],
                          accumulate=self.accumulate_grads, workspace=self._tn_ws)

    # ---- second stream for the weight-gradient products
    def _event(self):
        if self._ev_i == len(self._ev_pool):
            self._ev_pool.append(torch.cuda.Event())
        e = self._ev_pool[self._ev_i]
        self._ev_i += 1
        return e

    def _side_begin(self):
        """Start of a backward: decide whether the side stream is used (not under the per-launch GEMM timer of
        bench.py, whose HIP events must see one kernel at a time)."""
        self._use_side = bool(self.wgrad_side_stream) and ops.GEMM_TIMER is None
        self._ev_i = 0
        self._side_reads = {}
        if self._use_side and self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        if self._use_side and getattr(self, "_ws_for", None) != (self.B, getattr(self, "Mm_cap", 0)):
            self._ws_for = (self.B, getattr(self, "Mm_cap", 0))
            # the wgrad workspace is sized once so that it is never reallocated while the side stream uses it
            D, Hd = self.D, self.hidden
            M = self.B * self.T
            need = max(ops.gemm_tn_group_workspace([(M, D, D), (M, 3 * D, D)]), ops.gemm_tn_group_workspace([(M, D, Hd), (M, Hd, D)]),
                       ops.gemm_tn_workspace(M, 3 * D, D), ops.gemm_tn_workspace(M, Hd, D), ops.gemm_tn_workspace(M, D, Hd),
                       ops.gemm_tn_workspace(M, D, D), ops.gemm_tn_workspace(getattr(self, "Mm_cap", 0) or M, max(self.V, 1), D),
                       ops.gemm_tn_workspace(self.B * self.L, D, self.Kpe))
            if need > self._tn_ws.numel():
                self._tn_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)

    def _on_side(self, fn):
        """Run fn() (weight-gradient work that only READS what the main stream has produced so far) on the side stream."""
        if not self._use_side:
            fn()
            return
        e = self._event()
        e.record()                                           # everything enqueued on the main stream so far
        with torch.cuda.stream(self._side):
            self._side.wait_event(e)
            fn()

    def _side_read_done(self, name):
        """Mark the side stream's position after its (last) read of buffer `name`."""
        if self._use_side:
            e = self._event()
            e.record(self._side)
            self._side_reads[name] = e

    def _before_overwrite(self, name):
        """Main stream: wait until the side stream has finished reading buffer `name`."""
        if self._use_side:
            e = self._side_reads.pop(name, None)
            if e is not None:
                torch.cuda.current_stream().wait_event(e)

    def _side_join(self):
        """Main stream waits for the side stream (before a buffer the side stream reads is overwritten, before a
        gradient bucket is handed to the reducer, at the end of backward)."""
        if not self._use_side:
            return
        e = self._event()
        e.record(self._side)
        torch.cuda.current_stream().wait_event(e)

    def _bucket_ready(self, k):
        """Backward has left layer bucket k: hand it to the reducer (grad_hook).  The bucket's last writers are the main
        stream (bias / LayerNorm / table gradients, all enqueued by now) AND the side stream (the weight-gradient GEMMs).
        The MAIN stream never waits here: the side stream waits for the main stream's position and the hook runs with the
        side stream current, so the collective is ordered behind both while the main stream goes on with the next layer's
        dgrad chain (round 5; `hook_on_side = False` restores the per-layer join of the two streams, the A/B switch)."""
        if not self.grad_hook:
            return
        if not self._use_side:
            self.grad_hook(k)
            return
        if not self.hook_on_side:
            self._side_join()
            self.grad_hook(k)
            return
        e = self._event()
        e.record()
        with torch.cuda.stream(self._side):
            self._side.wait_event(e)
            self.grad_hook(k)

    # The dS-storing backward of the long-window attention kernels (memhip_attn_bwd_ws; round 6).  Kernels alone it measured equal to the
    # recomputing form on the first boxes (2 480-2 510 vs 2 490 us per layer at B = 64 x 16 heads x 1 201 tokens, profiles/r06_attn_win_ab.txt)
    # and 2-3 % faster on the later ones (2 484 vs 2 526-2 541, 2 397 vs 2 482); INSIDE the ViT-L step (tools/bench_vitl.py, MEMHIP_DS_WS=1,
    # interleaved on two boxes) it is 3 % faster: 222.1-224.2 vs 229.7-235.3 ms and 216.1-221.9 vs 227.0-227.7 ms -- its dQ kernel streams dS from
    # HBM while the weight-gradient GEMMs of the other stream hold the matrix pipes.  3 GB of scratch at B = 64 (46 MB per sample); windows the
    # kernels do not serve (14 x 14) return a zero workspace size and are not affected.
    attn_ds_workspace = True

    def _attn_ws(self, nb):
        if not self.attn_ds_workspace:
            return None
        need = ops.attn_bwd_workspace(nb, self.T, self.heads, self.window)
        if need == 0:
            return None
        ws = getattr(self, "_attn_ws_buf", None)
        if ws is None or ws.numel() < need:
            ws = self._attn_ws_buf = torch.empty(ops.attn_bwd_workspace(self.cur["B"], self.T, self.heads, self.window),
                                                 dtype=torch.uint8, device=self.dev)
        return ws

    def backward(self, dlogits=None):
        """Gradients of mean-CE (dlogits already in self.logits after forward(labels=...)) or of a
        caller-supplied dlogits (bf16 [Mm,V]) w.r.t. every parameter, into the flat grad buffer."""
        c = self.cur
        B, M, Mm = c["B"], c["M"], c["Mm"]
        D, Hd, T, L, V = self.D, self.hidden, self.T, self.L, self.V
        dp_masks = c["dp"]
        self.wait_optimizer()                # (the transposed copies; the gradient buffer is free again)
        if dlogits is not None:
            self.logits[:Mm].copy_(dlogits)
        if not self.accumulate_grads:
            self._zero_small_grads()
        dx = self.dx
        ops.zero_(dx[:M])
        dl = self.logits
        # ---- head
        self._side_begin()
        ops.gemm_nt(dl, self.wT_lm, Mm, D, V, ops.EPI_BIAS_BF16, out0=self.dhN)
        self._on_side(lambda: self._wgrad(dl, self.hN, Mm, V, D, "lm_head.weight", bias_grads=((self.G("lm_head.bias"), 0, V),)))
        if c.get("tail") is not None:
            # the head's rows in compact form (the last block's MLP backward reads them), then into the residual gradient
            ops.layernorm_bwd(self.dhN, self.x_tail, self.P("norm.weight"), self.meanN, self.rstdN, self.dxc,
                              self.G("norm.weight"), self.G("norm.bias"), Mm, D, accumulate=False)
            ops.scatter_rows(self.dxc, c["rows"], Mm, D, dx)
        else:
            ops.layernorm_bwd(self.dhN, self.x[2 * self.depth], self.P("norm.weight"), self.meanN, self.rstdN, dx,
                              self.G("norm.weight"), self.G("norm.bias"), Mm, D, accumulate=False, row_idx=c["rows"])
        self._backward_trunk()

    def backward_trunk(self, dxl):
        """Finetuning: gradient of the loss w.r.t. the trunk output (f32 [B*T, D] or [B,T,D]) -> every trunk parameter."""
        c = self.cur
        M, D = c["M"], self.D
        if not self.accumulate_grads:
            self._zero_small_grads(self.head_end)
        self.dx[:M].copy_(dxl.reshape(M, D))
        self._side_begin()
        self._backward_trunk()

    def _mlp_bwd_tail(self, i, a, pre, has_g, keep, tail, attn_branch_bwd):
        """Backward of _mlp_fwd_tail: the MLP branch of the last block on the compact rows that reached the head (their
        gradient is self.dxc; every other row's is zero), norm2 backward scattered into dx, then the attention branch's
        branch backward over all of dx (attn_branch_bwd: the caller's masked / work-skipping form)."""
        D, Hd = self.D, self.hidden
        rows, Mm, rk = tail["rows"], tail["Mm"], tail["rowkeep"]
        Mp = _pad(Mm, 256)
        dx, dY = self.dx, self.dY
        self._before_overwrite("dY")
        ops.branch_bwd(self.dxc, None, self.P(pre + "gamma_2") if has_g else None, dY, None, self.G(pre + "mlp.fc2.bias"),
                       Mm, D, rowmask=rk, keep_prob=keep, rows_per_sample=1)
        if Mp > Mm:
            ops.zero_(dY[Mm:Mp])
        self._before_overwrite("dbig")
        ops.gemm_nt(dY, self.wT[i]["fc2"], Mp, Hd, D, self.epi_dgelu, out0=self.dbig, aux=a["hpre"],
                    colsum=self.cs_ws, colsum_copies=self.CS_COPIES)
        ops.colsum_fold(self.cs_ws, self.CS_COPIES, Hd, self.G(pre + "mlp.fc1.bias"))

        def wg_mlp():
            self._wg_mlp(pre, a, has_g, Mm, dY)
        self._on_side(wg_mlp)
        ops.gemm_nt(self.dbig, self.wT[i]["fc1"], Mp, D, Hd, ops.EPI_BIAS_BF16, out0=self.dh_small)
        ops.layernorm_bwd(self.dh_small, self.x[2 * i + 1], self.P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx,
                          self.G(pre + "norm2.weight"), self.G(pre + "norm2.bias"), Mm, D, accumulate=True, row_idx=rows)
        attn_branch_bwd()

    def _backward_trunk_skip(self, plan):
        """_backward_trunk with the dropped samples of every branch skipped (dp_skip): the branch gradients dY / dY2, the
        dgrad chain and the weight gradients run on the kept samples' rows only; the row kernels move between the compact
        row sets and the residual-stream gradient dx through sample maps (in_map: who the LayerNorm'ed branch kept,
        out_map: who the branch whose output gradient is produced kept)."""
        c = self.cur
        B, M = c["B"], c["M"]
        D, Hd, T, L = self.D, self.hidden, self.T, self.L
        dx = self.dx
        self._bucket_ready(0)
        ops.zero_(self.bias_scr)
        fuse = D <= 1024 and self.fuse_ln_branch
        nk = lambda j: B if plan["n"][j] is None else plan["n"][j]               # noqa: E731  kept samples of branch j
        cmap = lambda j: None if plan["n"][j] is None else plan["cmap"][j]       # noqa: E731
        ridx = lambda j: None if plan["n"][j] is None else plan["ridx"][j]       # noqa: E731
        kp = lambda j, i: (1.0 - self.model.blocks[i].drop_prob) if plan["n"][j] is not None else 1.0   # noqa: E731
        for i in reversed(range(self.depth)):
            pre = f"blocks.{i}."
            a = self.act[i]
            ja, jm = 2 * i, 2 * i + 1
            na, nm = nk(ja), nk(jm)
            M1, M2 = na * T, nm * T
            # dgrad GEMM rows: whole 256-row tiles (ensure_batch); the weight gradients take the exact row counts
            M1p = _pad(M1, 256) if plan["n"][ja] is not None else M1
            M2p = _pad(M2, 256) if plan["n"][jm] is not None else M2
            xin, xmid = self.x[2 * i], self.x[2 * i + 1]
            has_g = (pre + "gamma_1") in self.segs
            table, dtable = self.table(i), self.dtable(i)
            dY, dY2 = self.dY, self.dY2
            scr = self.bias_scr[i & 1]
            tail = c.get("tail") if i == self.depth - 1 else None
            if tail is not None:
                # last block, MLP branch on the rows that reached the head only (forward(): tail rows); the helper ends with
                # the attention branch's branch backward in this loop's form
                def attn_bb(i=i, pre=pre, has_g=has_g, ja=ja, scr=scr):
                    self._before_overwrite("dY2")
                    ops.branch_bwd(dx, None, self.P(pre + "gamma_1") if has_g else None, dY2, None, scr, M, D,
                                   keep_prob=kp(ja, i), rows_per_sample=T, out_map=cmap(ja))
                self._mlp_bwd_tail(i, a, pre, has_g, 1.0 - self.model.blocks[i].drop_prob, tail, attn_bb)
            # -- MLP branch (for every block but the last dY already came out of the fused norm1 backward of block i + 1)
            if tail is None and (i == self.depth - 1 or not fuse):
                self._before_overwrite("dY")
                ops.branch_bwd(dx, None, self.P(pre + "gamma_2") if has_g else None, dY, None, self.G(pre + "mlp.fc2.bias"),
                               M, D, keep_prob=kp(jm, i), rows_per_sample=T, out_map=cmap(jm))
            if tail is None and nm == 0 and not self.accumulate_grads:
                self._zero_grad_of(pre + "mlp.fc1.weight")      # every sample dropped this branch: zero, not last step's
                self._zero_grad_of(pre + "mlp.fc2.weight")
            if tail is None and nm > 0:
                self._before_overwrite("dbig")
                if M2p > M2:
                    ops.zero_(dY[M2:M2p])   # rows of the padding: zero in, zero out (the epilogue's column sums see them)
                ops.gemm_nt(dY, self.wT[i]["fc2"], M2p, Hd, D, self.epi_dgelu, out0=self.dbig, aux=a["hpre"],
                            colsum=self.cs_ws, colsum_copies=self.CS_COPIES)
                ops.colsum_fold(self.cs_ws, self.CS_COPIES, Hd, self.G(pre + "mlp.fc1.bias"))

                def wg_mlp(i=i, pre=pre, a=a, has_g=has_g, M2=M2):
                    self._wg_mlp(pre, a, has_g, M2, dY)
                self._on_side(wg_mlp)
                ops.gemm_nt(self.dbig, self.wT[i]["fc1"], M2p, D, Hd, ops.EPI_BIAS_BF16, out0=self.dh_small)
            if tail is not None:
                pass
            elif fuse:
                # norm2 backward (rows the MLP branch kept) + attention-branch backward (rows it kept) in one pass over dx
                self._before_overwrite("dY2")
                ops.layernorm_bwd_branch(self.dh_small, xmid, self.P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx,
                                         self.G(pre + "norm2.weight"), self.G(pre + "norm2.bias"), M, D, None,
                                         self.P(pre + "gamma_1") if has_g else None, dY2, None, scr,
                                         keep_prob=kp(ja, i), rows_per_sample=T,
                                         in_map=cmap(jm) if cmap(jm) is not None else None, out_map=cmap(ja))
            else:
                if nm > 0:
                    ops.layernorm_bwd(self.dh_small, xmid, self.P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx,
                                      self.G(pre + "norm2.weight"), self.G(pre + "norm2.bias"), M2, D, accumulate=True,
                                      row_idx=ridx(jm))
                self._before_overwrite("dY2")
                ops.branch_bwd(dx, None, self.P(pre + "gamma_1") if has_g else None, dY2, None, scr, M, D,
                               keep_prob=kp(ja, i), rows_per_sample=T, out_map=cmap(ja))
            # -- attention branch
            if na == 0 and not self.accumulate_grads:
                self._zero_grad_of(pre + "attn.proj.weight")
                self._zero_grad_of(pre + "attn.qkv.weight")
            if na > 0:
                ops.gemm_nt(dY2, self.wT[i]["proj"], M1p, D, D, ops.EPI_BIAS_BF16, out0=self.dao)
            ops.gemv_acc(self.wT[i]["proj"], D, D, scr, self.G(pre + "attn.v_bias"),
                         x_acc=self.G(pre + "attn.proj.bias"), zero=self.bias_scr[(i & 1) ^ 1])
            if na > 0:
                def wg_proj(pre=pre, a=a, has_g=has_g, M1=M1, grouped=False):
                    if not grouped:
                        self._wgrad(dY2, a["ao"], M1, D, D, pre + "attn.proj.weight")
                    self._side_read_done("dY2")
                    if has_g:
                        ops.layerscale_grad(self.W16(pre + "attn.proj.weight", D, D), self.G(pre + "attn.proj.weight").view(D, D),
                                            self.P(pre + "attn.proj.bias"), self.G(pre + "attn.proj.bias"),
                                            self.P(pre + "gamma_1"), D, D, self.G(pre + "gamma_1"))
                if not self.wgrad_group:
                    self._on_side(wg_proj)
                self._before_overwrite("dqkv")
                # (rowsum(dO * O) is computed inside the fused 14 x 14 backward; other windows: a delta pass in the library)
                ops.attn_bwd(a["qkv"], self.dao, a["lse"], self.delta_ws, table, self.window, na, T, D, self.heads,
                             self.scale, self.dqkv, dtable, dq_bias=self.G(pre + "attn.q_bias"), out=a["ao"], ws=self._attn_ws(na))

                def wg_qkv(pre=pre, a=a, M1=M1, wg_proj=wg_proj):
                    if self.wgrad_group:
                        self._wgrad_group([(dY2, a["ao"], M1, D, D, pre + "attn.proj.weight"),
                                           (self.dqkv, a["h1"], M1, 3 * D, D, pre + "attn.qkv.weight")])
                        wg_proj(grouped=True)
                    else:
                        self._wgrad(self.dqkv, a["h1"], M1, 3 * D, D, pre + "attn.qkv.weight")
                    self._side_read_done("dqkv")
                self._on_side(wg_qkv)
                ops.gemm_nt(self.dqkv, self.wT[i]["qkv"], M1p, D, 3 * D, ops.EPI_BIAS_BF16, out0=self.dh_small)
            if fuse and i > 0:
                # norm1 backward of block i (rows its attention branch kept) + MLP-branch backward of block i - 1
                pb = f"blocks.{i - 1}."
                has_gb = (pb + "gamma_1") in self.segs
                jb = 2 * (i - 1) + 1
                self._before_overwrite("dY")
                ops.layernorm_bwd_branch(self.dh_small, xin, self.P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx,
                                         self.G(pre + "norm1.weight"), self.G(pre + "norm1.bias"), M, D, None,
                                         self.P(pb + "gamma_2") if has_gb else None, dY, None, self.G(pb + "mlp.fc2.bias"),
                                         keep_prob=kp(jb, i - 1), rows_per_sample=T, in_map=cmap(ja), out_map=cmap(jb))
            elif na > 0:
                ops.layernorm_bwd(self.dh_small, xin, self.P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx,
                                  self.G(pre + "norm1.weight"), self.G(pre + "norm1.bias"), M1, D, accumulate=True,
                                  row_idx=ridx(ja))
            self._bucket_ready(self.depth - i)
        self._backward_embed(c, B, M)

    def _backward_trunk(self):
        c = self.cur
        if c.get("plan") is not None:
            return self._backward_trunk_skip(c["plan"])
        B, M = c["B"], c["M"]
        D, Hd, T, L = self.D, self.hidden, self.T, self.L
        dp_masks = c["dp"]
        dx = self.dx
        self._bucket_ready(0)
        # both ping-pong rows of the proj-bias scratch start clean: block i accumulates into row i&1 and clears the
        # other one, which leaves row (depth-1)&1 dirty for the next backward when depth is odd
        ops.zero_(self.bias_scr)
        fuse = D <= 1024 and self.fuse_ln_branch
        for i in reversed(range(self.depth)):
            pre = f"blocks.{i}."
            a = self.act[i]
            blk = self.model.blocks[i]
            keep = 1.0 - blk.drop_prob
            use_dp = dp_masks is not None and blk.drop_prob > 0.0
            xin, xmid = self.x[2 * i], self.x[2 * i + 1]
            has_g = (pre + "gamma_1") in self.segs
            table, dtable = self.table(i), self.dtable(i)
            # -- MLP branch (for every block but the last this already ran fused into the norm1 backward of
            # block i+1, see below).  dY = gradient of the MLP branch output, dY2 = of the attention branch output.
            dY, dY2 = self.dY, self.dY2
            tail = c.get("tail") if i == self.depth - 1 else None
            if tail is not None:
                # last block, MLP branch on the rows that reached the head only (forward(): tail rows)
                scr = self.bias_scr[i & 1]

                def attn_bb(i=i, pre=pre, has_g=has_g, scr=scr, use_dp=use_dp, keep=keep):
                    self._before_overwrite("dY2")
                    ops.branch_bwd(dx, None, self.P(pre + "gamma_1") if has_g else None, dY2, None, scr, M, D,
                                   rowmask=dp_masks[2 * i] if use_dp else None, keep_prob=keep, rows_per_sample=T)
                self._mlp_bwd_tail(i, a, pre, has_g, keep, tail, attn_bb)
            else:
                if i == self.depth - 1 or not fuse:
                    self._before_overwrite("dY")
                    ops.branch_bwd(dx, None, self.P(pre + "gamma_2") if has_g else None, dY,
                                   None, self.G(pre + "mlp.fc2.bias"), M, D,
                                   rowmask=dp_masks[2 * i + 1] if use_dp else None, keep_prob=keep, rows_per_sample=T)
                self._before_overwrite("dbig")
                # fc1 bias grad = column sums of dh, accumulated in CS_COPIES copies (one per XCD: atomics on one address
                # serialise and would sit in front of the GEMM's operand stream) and folded by a 3 us kernel
                ops.gemm_nt(dY, self.wT[i]["fc2"], M, Hd, D, self.epi_dgelu, out0=self.dbig, aux=a["hpre"],
                            colsum=self.cs_ws, colsum_copies=self.CS_COPIES)
                ops.colsum_fold(self.cs_ws, self.CS_COPIES, Hd, self.G(pre + "mlp.fc1.bias"))

                def wg_mlp(i=i, pre=pre, a=a, has_g=has_g):
                    self._wg_mlp(pre, a, has_g, M, dY)
                self._on_side(wg_mlp)
                ops.gemm_nt(self.dbig, self.wT[i]["fc1"], M, D, Hd, ops.EPI_BIAS_BF16, out0=self.dh_small)
                scr = self.bias_scr[i & 1]
                if fuse:
                    # norm2 backward + attention-branch backward in one pass over dx (proj.bias column sums -> scr)
                    self._before_overwrite("dY2")
                    ops.layernorm_bwd_branch(self.dh_small, xmid, self.P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx,
                                             self.G(pre + "norm2.weight"), self.G(pre + "norm2.bias"), M, D, None,
                                             self.P(pre + "gamma_1") if has_g else None, dY2, None, scr,
                                             rowmask=dp_masks[2 * i] if use_dp else None, keep_prob=keep, rows_per_sample=T)
                else:
                    ops.layernorm_bwd(self.dh_small, xmid, self.P(pre + "norm2.weight"), a["mean2"], a["rstd2"], dx,
                                      self.G(pre + "norm2.weight"), self.G(pre + "norm2.bias"), M, D, accumulate=True)
                # -- attention branch
                # proj.bias gradient (column sums of dY) goes to a scratch vector first: the v_bias gradient is
                # derived from it.  sum_k dV[k] = sum_q dO[q] * sum_k P[q,k] and the softmax rows sum to one, so
                # v_bias.grad = colsum(d attn_out) = colsum(dY) @ W_proj: one 768x768 GEMV instead of column
                # sums inside the attention kernel (which cost it 32 VGPRs and its occupancy).
                if not fuse:
                    self._before_overwrite("dY2")
                    ops.branch_bwd(dx, None, self.P(pre + "gamma_1") if has_g else None, dY2, None, scr, M, D,
                                   rowmask=dp_masks[2 * i] if use_dp else None, keep_prob=keep, rows_per_sample=T)
            ops.gemm_nt(dY2, self.wT[i]["proj"], M, D, D, ops.EPI_BIAS_BF16, out0=self.dao)
            ops.gemv_acc(self.wT[i]["proj"], D, D, scr, self.G(pre + "attn.v_bias"),
                         x_acc=self.G(pre + "attn.proj.bias"), zero=self.bias_scr[(i & 1) ^ 1])

            def wg_proj(pre=pre, a=a, has_g=has_g, grouped=False):
                if not grouped:
                    self._wgrad(dY2, a["ao"], M, D, D, pre + "attn.proj.weight")
                self._side_read_done("dY2")
                if has_g:
                    ops.layerscale_grad(self.W16(pre + "attn.proj.weight", D, D), self.G(pre + "attn.proj.weight").view(D, D),
                                        self.P(pre + "attn.proj.bias"), self.G(pre + "attn.proj.bias"),
                                        self.P(pre + "gamma_1"), D, D, self.G(pre + "gamma_1"))
            if not self.wgrad_group:
                self._on_side(wg_proj)
            self._before_overwrite("dqkv")
            # (rowsum(dO * O) is computed inside the fused 14 x 14 backward; other windows: a delta pass in the library)
            ops.attn_bwd(a["qkv"], self.dao, a["lse"], self.delta_ws, table, self.window, B, T, D, self.heads,
                         self.scale, self.dqkv, dtable, dq_bias=self.G(pre + "attn.q_bias"), out=a["ao"], ws=self._attn_ws(B))

            def wg_qkv(pre=pre, a=a, wg_proj=wg_proj):
                if self.wgrad_group:
                    self._wgrad_group([(dY2, a["ao"], M, D, D, pre + "attn.proj.weight"),
                                       (self.dqkv, a["h1"], M, 3 * D, D, pre + "attn.qkv.weight")])
                    wg_proj(grouped=True)
                else:
                    self._wgrad(self.dqkv, a["h1"], M, 3 * D, D, pre + "attn.qkv.weight")
                self._side_read_done("dqkv")
            self._on_side(wg_qkv)
            ops.gemm_nt(self.dqkv, self.wT[i]["qkv"], M, D, 3 * D, ops.EPI_BIAS_BF16, out0=self.dh_small)
            if fuse and i > 0:
                # norm1 backward of block i + MLP-branch backward of block i-1 in one pass over dx
                pb, ab_, bb_ = f"blocks.{i - 1}.", self.act[i - 1], self.model.blocks[i - 1]
                has_gb = (pb + "gamma_1") in self.segs
                use_dpb = dp_masks is not None and bb_.drop_prob > 0.0
                self._before_overwrite("dY")
                ops.layernorm_bwd_branch(self.dh_small, xin, self.P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx,
                                         self.G(pre + "norm1.weight"), self.G(pre + "norm1.bias"), M, D, None,
                                         self.P(pb + "gamma_2") if has_gb else None, dY, None,
                                         self.G(pb + "mlp.fc2.bias"),
                                         rowmask=dp_masks[2 * (i - 1) + 1] if use_dpb else None,
                                         keep_prob=1.0 - bb_.drop_prob, rows_per_sample=T)
            else:
                ops.layernorm_bwd(self.dh_small, xin, self.P(pre + "norm1.weight"), a["mean1"], a["rstd1"], dx,
                                  self.G(pre + "norm1.weight"), self.G(pre + "norm1.bias"), M, D, accumulate=True)
            self._bucket_ready(self.depth - i)
        self._backward_embed(c, B, M)

    def _backward_embed(self, c, B, M):
        D, T, L = self.D, self.T, self.L
        dx = self.dx
        # ---- embedding
        if self.has_pos:
            if self.accumulate_grads:
                self.G("pos_embed").view(T, D).add_(dx[:M].view(B, T, D).sum(0))
            else:
                self.G("pos_embed").view(T, D).copy_(dx[:M].view(B, T, D).sum(0))
        ops.embed_bwd(dx, c["mask"], B, L, D, self.dYpe, self.G("cls_token"),
                      self.G("mask_token") if "mask_token" in self.segs else self.zero_vec)
        self._on_side(lambda: self._wgrad(self.dYpe, self.patches, B * L, D, self.Kpe, "patch_embed.proj.weight",
                                          bias_grads=((self.G("patch_embed.proj.bias"), 0, D),)))
        self._side_join()                                     # every gradient is final on the main stream from here on
        if self.grad_hook:
            self.grad_hook(self.depth + 1)

    # ------------------------------------------------------------------ optimizer primitives
    def grad_norm(self):
        ops.grad_norm(self.flat_g, self.nflat, self.gnorm, self.gn_ws)
        return self.gnorm

    def adamw_step(self, m, v, lr, wd, step, betas=(0.9, 0.95), eps=1e-8, max_norm=0.0):
        if not (self.overlap_optimizer and self.head_kind == "mlm" and not self.fwd_two_streams):
            self.wait_optimizer()
            ops.adamw(self.flat_p, self.flat_g, m, v, self.nflat, self.wd_flags, lr, betas[0], betas[1], eps, wd, step,
                      gnorm=self.gnorm, max_norm=max_norm or 0.0)
            self.weights_dirty = True
            return
        # ---- pipelined with the next forward (see __init__): bucket by bucket in forward order on the optimizer stream
        self.wait_optimizer()
        if self._tdesc is None:
            self._build_transpose_descs()
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.dev)
            self._opt_events = {b[0]: torch.cuda.Event() for b in self.buckets}
            self._opt_events["__done__"] = torch.cuda.Event()
            self._opt_start = torch.cuda.Event()
        main, st = torch.cuda.current_stream(), self._opt_stream
        self._opt_start.record(main)                          # gradients and their norm are final
        with torch.cuda.stream(st):
            st.wait_event(self._opt_start)
            for name, b0, b1 in reversed(self.buckets):       # embed, block 0 .. block depth-1, head
                if b1 == b0:
                    self._opt_events[name].record(st)
                    continue
                ops.adamw(self.flat_p[b0:b1], self.flat_g[b0:b1], m[b0:b1], v[b0:b1], b1 - b0,
                          self.wd_flags[b0 // ALIGN: b1 // ALIGN], lr, betas[0], betas[1], eps, wd, step,
                          gnorm=self.gnorm, max_norm=max_norm or 0.0)
                ops.cast_f32_bf16(self.flat_p[b0:b1], self.flat_w16[b0:b1], b1 - b0)
                self._opt_events[name].record(st)
            for name, (k0, n, p0, tiles) in self._tbucket.items():      # [in,out]-major copies: only backward reads them
                ops.transpose_cast_batched(self._tdesc[k0:k0 + n], self._tprefix_loc[p0:p0 + n + 1], n, tiles)
            self._opt_events["__done__"].record(st)
        self._opt_ev = self._opt_events
        self._opt_done = self._opt_events["__done__"]
        self.weights_dirty = False

    def _wait_params(self, bucket):
        """The launch stream waits until the pipelined optimizer has updated (and cast) the parameters of `bucket`."""
        if self._opt_ev is not None:
            torch.cuda.current_stream().wait_event(self._opt_ev[bucket])

    def wait_optimizer(self):
        """The launch stream waits for everything the pipelined optimizer has in flight (parameters, bf16 shadows, transposed
        copies; it has then also finished reading the gradient buffer)."""
        if self._opt_done is not None:
            torch.cuda.current_stream().wait_event(self._opt_done)
            self._opt_ev = None
            self._opt_done = None
