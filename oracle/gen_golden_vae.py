"""Generates tests/golden/vae_tiny.npz from the REFERENCE DiscreteVAE (build container only; the
reference never travels): seeded weights (recipe by name), seeded images -> logits and token ids, after
asserting that the oracle restatement (oracle/vae_ref.py) reproduces the reference bit for bit."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _refimport                                     # noqa: E402
from oracle.vae_ref import BASE_VAE, TINY_VAE, encoder_logits, fill_vae_by_name, get_codebook_indices, vae_inputs   # noqa: E402


def main():
    assert _refimport.install(), "/root/reference is not available"
    torch.set_num_threads(1)
    from vae.vae_model import DiscreteVAE                         # the reference class
    ref = DiscreteVAE(**TINY_VAE).eval()
    sd = fill_vae_by_name(ref.state_dict(), seed=0)
    ref.load_state_dict(sd)
    img = vae_inputs(TINY_VAE, 6, 11)
    with torch.no_grad():
        ref_logits = ref(img, return_logits=True)
        ref_ids = ref.get_codebook_indices(img)
        ora_logits = encoder_logits(sd, img, TINY_VAE["num_layers"], TINY_VAE["num_resnet_blocks"])
        ora_ids = get_codebook_indices(sd, img, TINY_VAE["num_layers"], TINY_VAE["num_resnet_blocks"])
    assert torch.equal(ref_logits, ora_logits), "oracle logits != reference"
    assert torch.equal(ref_ids, ora_ids), "oracle ids != reference"
    srt = ref_logits.flatten(2).transpose(1, 2).sort(dim=-1, descending=True).values
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vae_tiny.npz")
    np.savez_compressed(out, ids=ref_ids.numpy().astype(np.int32),
                        top2_gap=(srt[..., 0] - srt[..., 1]).numpy().astype(np.float32),
                        logits_b0=ref_logits[0].numpy().astype(np.float32),
                        keys=np.array(sorted(sd.keys())))
    print("wrote", out, "ids", tuple(ref_ids.shape), "state-dict tensors", len(sd))
    # ---- the ViT-B tokenizer shape (hidden 384, 8192 tokens, 224^2), 2 samples, event-like sparse inputs
    torch.set_num_threads(8)
    ref = DiscreteVAE(**BASE_VAE).eval()
    sd = fill_vae_by_name(ref.state_dict(), seed=1)
    ref.load_state_dict(sd)
    img = vae_inputs(BASE_VAE, 2, 12)
    img = img * (vae_inputs(BASE_VAE, 2, 13) < 0.3)
    with torch.no_grad():
        ref_logits = ref(img, return_logits=True)
        ref_ids = ref.get_codebook_indices(img)
        ora_ids = get_codebook_indices(sd, img, BASE_VAE["num_layers"], BASE_VAE["num_resnet_blocks"])
    assert torch.equal(ref_ids, ora_ids), "oracle ids != reference"
    srt = ref_logits.flatten(2).transpose(1, 2).topk(2, dim=-1).values
    out = os.path.join(os.path.dirname(out), "vae_base.npz")
    np.savez_compressed(out, ids=ref_ids.numpy().astype(np.int32),
                        top2_gap=(srt[..., 0] - srt[..., 1]).numpy().astype(np.float32),
                        logit_std=np.float32(ref_logits.std().item()))
    print("wrote", out, "ids", tuple(ref_ids.shape), "min top-2 gap %.3e, logit std %.3f" %
          (float((srt[..., 0] - srt[..., 1]).min()), ref_logits.std().item()))


if __name__ == "__main__":
    main()
