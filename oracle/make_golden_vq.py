"""Golden fixture of the VQGAN first stage, produced by the REAL reference (/root/reference/model/VQGAN).

TEST INFRASTRUCTURE.  Run in the build container only (the reference is not on the GPU box):

    python oracle/make_golden_vq.py            ->  tests/golden/vq_f4_small.pt

The reference's ``VQModel`` (model/VQGAN/vqgan.py:31-93; Encoder / Decoder model.py:342-537; VectorQuantizer2
quantize.py:271-313) is imported unmodified (pytorch_lightning is stubbed: LightningModule = nn.Module) at the VQ-f4 geometry of
configs/Template-LBBDM-f4.yaml:55-72 -- ch 128, ch_mult (1, 2, 4), 2 ResBlocks per level, 8192 x 3 codebook -- shrunk only in
the image size (64x64 instead of 256x256: a 16x16 latent) and with attention switched on at the 16x16 level, so that the
single-head 512-channel AttnBlock path (model.py:140-192) runs.  Weights come from tests/fixture_weights.synth_weights (frozen
numpy streams: regenerated bit-identically on the GPU box, the fixture does not carry the 55 M parameters); the codebook is
scaled to the spread of the encoder's latents so that the nearest-code search is non-trivial.

Stored: the input images, the pre-quantisation latent, the codebook, the code indices, the quantised latent and the decoded
images -- plus a NEAR-TIE case for the codebook search: latents on the midpoint between a code and its nearest other code
(pushed 1e-6 of their distance towards the first -- below the fp32 resolution of the distance expression, so WHICH of the two
wins is rounding noise of the host's matmul) with the two candidates and the indices the reference returned here.
"""
from __future__ import annotations

import argparse
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(os.path.dirname(HERE), "tests")]
from fixture_weights import synth_weights  # noqa: E402
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "vq_f4_small.pt")

DDCONFIG = dict(double_z=False, z_channels=3, resolution=64, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 4),
                num_res_blocks=2, attn_resolutions=[16], dropout=0.0)
N_EMBED, EMBED_DIM, WEIGHT_SEED = 8192, 3, 2718


def main():
    pl = types.ModuleType("pytorch_lightning")
    pl.__spec__ = importlib.machinery.ModuleSpec("pytorch_lightning", None)
    pl.LightningModule = nn.Module
    sys.modules["pytorch_lightning"] = pl
    sys.path.insert(0, REF)
    from model.VQGAN.vqgan import VQModel as RefVQ

    torch.manual_seed(0)
    # (the reference takes its sub-configs as namespaces: vqgan.py:47 does Encoder(**vars(ddconfig)))
    ref = RefVQ(ddconfig=argparse.Namespace(**DDCONFIG), lossconfig=argparse.Namespace(target="torch.nn.Identity"), n_embed=N_EMBED,
                embed_dim=EMBED_DIM).eval()
    shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    sd = synth_weights(shapes, WEIGHT_SEED)
    ref.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(1618)
    x = torch.randn(2, 3, 64, 64, generator=g).clamp(-1, 1)
    with torch.no_grad():
        z = ref.quant_conv(ref.encoder(x))
        # codebook with the spread of the latents (frozen numpy stream), then the reference's own quantiser and decoder
        cb = torch.from_numpy(np.random.RandomState(31337).standard_normal((N_EMBED, EMBED_DIM)).astype(np.float32)) * float(z.std())
        ref.quantize.embedding.weight.copy_(cb)
        zq, _, (_, _, idx) = ref.quantize(z)
        img = ref.decode(zq)
        # near ties: the midpoint between a code and its NEAREST other code (those two are the closest codes to it), nudged
        # towards the first by 1e-6 of their distance
        first = torch.from_numpy(np.random.RandomState(7).randint(0, N_EMBED, size=256))
        dist = torch.cdist(cb[first].double(), cb.double())
        dist[torch.arange(256), first] = float("inf")
        pairs = torch.stack([first, dist.argmin(dim=1)], dim=1)
        a, b = cb[pairs[:, 0]], cb[pairs[:, 1]]
        z_tie = (0.5 * (a + b) + 1e-6 * (a - b)).t().reshape(1, EMBED_DIM, 16, 16).contiguous()
        _, _, (_, _, idx_tie) = ref.quantize(z_tie)
    rec = {"ddconfig": DDCONFIG, "n_embed": N_EMBED, "embed_dim": EMBED_DIM, "weight_seed": WEIGHT_SEED, "shapes": shapes,
           "x": x, "z": z, "codebook": cb, "indices": idx.reshape(-1).clone(), "zq": zq, "img": img,
           "z_tie": z_tie, "indices_tie": idx_tie.reshape(-1).clone(), "tie_pairs": pairs.clone()}
    torch.save(rec, OUT)
    nparam = sum(int(np.prod(s)) for _, s in shapes)
    print(f"vq_f4_small: {nparam / 1e6:.1f} M parameters, latent {tuple(z.shape)}, {len(set(idx.reshape(-1).tolist()))} distinct codes of "
          f"{idx.numel()}, near-tie case: {int((idx_tie.reshape(-1) == pairs[:, 0]).sum())}/256 resolved to the nudged side "
          f"-> {OUT} ({os.path.getsize(OUT) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
