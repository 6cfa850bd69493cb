"""Event dVAE tokenizer, inference side -- adjacent dependency of the pretraining loop
(/root/reference/eventvae/vae/vae_model.py:29-113,153-189; called every step at
mem/engine_for_pretraining.py:144).  SURVEY.md section 8 row a22 / f1: v1 runs the frozen tokenizer
through stock PyTorch-ROCm (MIOpen convs); a hand-written implicit-GEMM version is the next row.
Same module tree => same state-dict keys as the reference checkpoints ('hparams' / 'weights').
Training the dVAE (gumbel-softmax path, losses) is a different stage and not provided."""
import torch
from torch import nn


class ResBlock(nn.Module):
    def __init__(self, chan):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(chan, chan, 3, padding=1), nn.ReLU(), nn.Conv2d(chan, chan, 3, padding=1),
                                 nn.ReLU(), nn.Conv2d(chan, chan, 1))

    def forward(self, x):
        return self.net(x) + x


class DiscreteVAE(nn.Module):
    def __init__(self, input_H=256, input_W=256, num_tokens=512, codebook_dim=512, num_layers=3,
                 num_resnet_blocks=0, hidden_dim=64, channels=3, loss="mse", temperature=0.9,
                 straight_through=False, kl_div_loss_weight=0.0, normalization=None):
        super().__init__()
        assert input_H % (2 ** num_layers) == 0 and input_W % (2 ** num_layers) == 0 and num_layers >= 1
        self.input_H, self.input_W, self.input_size = input_H, input_W, (input_H, input_W)
        self.num_tokens, self.num_layers = num_tokens, num_layers
        self.normalization = normalization
        self.codebook = nn.Embedding(num_tokens, codebook_dim)
        has_res = num_resnet_blocks > 0
        enc_chans = [channels] + [hidden_dim] * num_layers
        dec_chans = [codebook_dim if not has_res else hidden_dim] + [hidden_dim] * num_layers
        enc, dec = [], []
        for (ei, eo), (di, do) in zip(zip(enc_chans[:-1], enc_chans[1:]), zip(dec_chans[:-1], dec_chans[1:])):
            enc.append(nn.Sequential(nn.Conv2d(ei, eo, 4, stride=2, padding=1), nn.ReLU()))
            dec.append(nn.Sequential(nn.ConvTranspose2d(di, do, 4, stride=2, padding=1), nn.ReLU()))
        for _ in range(num_resnet_blocks):
            dec.insert(0, ResBlock(dec_chans[1]))
            enc.append(ResBlock(enc_chans[-1]))
        if has_res:
            dec.insert(0, nn.Conv2d(codebook_dim, dec_chans[1], 1))
        enc.append(nn.Conv2d(enc_chans[-1], num_tokens, 1))
        dec.append(nn.Conv2d(dec_chans[-1], channels, 1))
        self.encoder, self.decoder = nn.Sequential(*enc), nn.Sequential(*dec)

    def norm(self, images):
        if self.normalization is None:
            return images
        means, stds = (torch.as_tensor(t).to(images).view(1, -1, 1, 1) for t in self.normalization)
        return (images - means) / stds

    @torch.no_grad()
    def get_codebook_indices(self, images):
        """vae_model.py:153-158: argmax over the token logits, eval mode, no grad -> i64 [B, h*w]."""
        was = self.training
        self.eval()
        logits = self.encoder(self.norm(images))
        self.train(was)
        return logits.argmax(dim=1).flatten(1)

    def forward(self, img, return_logits=False, **kw):
        assert img.shape[-1] == self.input_W and img.shape[-2] == self.input_H
        if not return_logits:
            raise NotImplementedError("dVAE training / reconstruction is a separate stage (out of scope)")
        return self.encoder(self.norm(img))

    def decode(self, img_seq):
        emb = self.codebook(img_seq)
        b, n, d = emb.shape
        h, w = self.input_H // 2 ** self.num_layers, self.input_W // 2 ** self.num_layers
        return self.decoder(emb.transpose(1, 2).reshape(b, d, h, w))
