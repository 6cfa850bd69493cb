"""ORACLE (test infrastructure, not product): CPU restatement of the frozen dVAE tokenizer forward.

Follows /root/reference/eventvae/vae/vae_model.py:
  ResBlock            :29-42     net = conv3x3, ReLU, conv3x3, ReLU, conv1x1 ; out = net(x) + x
  encoder stack       :76-101    num_layers x [conv4x4 stride 2 pad 1, ReLU], num_resnet_blocks x ResBlock,
                                 conv1x1 -> num_tokens
  norm                :132-140   (x - mean) / std per channel when `normalization` is given
  get_codebook_indices:153-158   argmax over the token logits, flattened to [B, h*w]
written functionally on a state dict (keys 'encoder.<i>...' as in the reference module tree), fp32.
Pinned by oracle/gen_golden_vae.py, which runs the reference class itself on the same seeded weights and
inputs and asserts bit-equality before writing tests/golden/vae_tiny.npz."""
import math

import torch
import torch.nn.functional as F


def encoder_logits(sd, img, num_layers, num_resnet_blocks, normalization=None):
    x = img
    if normalization is not None:
        mean, std = (torch.as_tensor(t, dtype=img.dtype).view(1, -1, 1, 1) for t in normalization)
        x = (x - mean) / std
    i = 0
    for _ in range(num_layers):                                   # Sequential(Conv2d(4, s2, p1), ReLU)
        x = F.relu(F.conv2d(x, sd[f"encoder.{i}.0.weight"], sd[f"encoder.{i}.0.bias"], stride=2, padding=1))
        i += 1
    for _ in range(num_resnet_blocks):                            # ResBlock
        p = f"encoder.{i}.net."
        y = F.relu(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1))
        y = F.relu(F.conv2d(y, sd[p + "2.weight"], sd[p + "2.bias"], padding=1))
        y = F.conv2d(y, sd[p + "4.weight"], sd[p + "4.bias"])
        x = y + x
        i += 1
    return F.conv2d(x, sd[f"encoder.{i}.weight"], sd[f"encoder.{i}.bias"])


def get_codebook_indices(sd, img, num_layers, num_resnet_blocks, normalization=None):
    return encoder_logits(sd, img, num_layers, num_resnet_blocks, normalization).argmax(dim=1).flatten(1)


def fill_vae_by_name(state_dict, seed=0):
    """Deterministic weight recipe keyed by tensor NAME (no reference bytes travel): conv weights
    ~ N(0, 1/fan_in) scaled so that activations keep O(1) variance through the ReLU stack, biases small."""
    import zlib
    out = {}
    for name, t in state_dict.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 104729 * seed) & 0x7FFFFFFF)
        if name.endswith(".bias"):
            v = 0.05 * torch.randn(t.shape, generator=g)
        elif t.ndim == 4:
            v = math.sqrt(2.0 / t[0].numel()) * torch.randn(t.shape, generator=g)
        else:
            v = 0.1 * torch.randn(t.shape, generator=g)
        out[name] = v.to(t.dtype)
    return out


TINY_VAE = dict(input_H=64, input_W=64, num_tokens=512, codebook_dim=32, num_layers=4, num_resnet_blocks=2,
                hidden_dim=64, channels=3)


# the MEM tokenizer at ViT-B pretraining shape (run_mem_pretraining.py:183-186 defaults + configs/ncaltech.conf):
# 4 stride-2 layers, hidden 384, 3 ResBlocks, 8192 tokens, 224 x 224 -> 14 x 14 ids
BASE_VAE = dict(input_H=224, input_W=224, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3,
                hidden_dim=384, channels=3)


def vae_inputs(cfg, batch, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, cfg["channels"], cfg["input_H"], cfg["input_W"], generator=g)
