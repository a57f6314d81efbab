"""Shared bodies of the first-stage (VQGAN on HIP) tests: bbdm_amd.first_stage_hip against the plain
PyTorch first stage it derives from (itself checked against the real reference VQModel in tests/test_first_stage.py)."""
import torch

from fixtures import rel_err

DD = dict(double_z=False, z_channels=4, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=(1, 2, 2), num_res_blocks=1,
          attn_resolutions=[16], dropout=0.0)


def make(dev, seed=0, **over):
    from bbdm_amd.first_stage_hip import VQModel
    torch.manual_seed(seed)
    m = VQModel(ddconfig=dict(DD, **over), n_embed=64, embed_dim=4)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for k, p in m.named_parameters():          # non-trivial norms / biases / codebook
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1 + (1.0 if "norm" in k and k.endswith("weight") else 0.0))
            elif "embedding" in k:
                p.copy_(torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (p[0].numel() ** 0.5)))
    return m.eval().to(dev)


def encode_decode_parity(dev, N=2, tol=2e-4, **over):
    m = make(dev, **over)
    ref = make(torch.device("cpu"), **over)
    g = torch.Generator().manual_seed(7)
    S = DD["resolution"]
    x = torch.randn(N, 3, S, S, generator=g).clamp(-1, 1)
    with torch.no_grad():
        z_ref = ref.quant_conv(ref.encoder(x))
        z_pre = ref.encoder(x)
    z = m.encode_latent(x.to(dev), quant_conv=True)
    assert rel_err(z.cpu(), z_ref) < tol
    assert rel_err(m.encode_latent(x.to(dev), quant_conv=False).cpu(), z_pre) < tol
    # decode: the SAME latent into both (so the codebook lookup sees identical inputs): indices must be bit-equal
    zl = z_ref + 0.05 * torch.randn(z_ref.shape, generator=g)
    with torch.no_grad():
        zq_ref, _, (_, _, idx_ref) = ref.quantize(zl)
        img_ref = ref.decode(zq_ref)
    img, idx = m.decode_latent(zl.to(dev), return_indices=True)
    assert torch.equal(idx.cpu().flatten(), idx_ref.flatten())
    assert rel_err(img.cpu(), img_ref) < tol
    with torch.no_grad():
        img_ref2 = ref.decode(ref.quantize(ref.quant_conv(z_pre))[0])
    assert rel_err(m.decode_latent(z_pre.to(dev), quant_conv_first=True).cpu(), img_ref2) < tol
    return m


def vq_indices_bit_exact(dev):
    """bbdm_vq_nearest_f32 against the reference expression (quantize.py:280-285) on hard inputs: many near-ties."""
    import kernel_ops as ops
    g = torch.Generator().manual_seed(11)
    for e_dim, n_e in ((3, 8192), (4, 256), (8, 16384)):
        cb = torch.randn(n_e, e_dim, generator=g) * 0.5
        z = torch.randn(5000, e_dim, generator=g) * 0.6
        z[:1000] = cb[torch.randint(0, n_e, (1000,), generator=g)] + 1e-4 * torch.randn(1000, e_dim, generator=g)
        z[1000:1100] = 0.5 * (cb[:100] + cb[100:200])                    # midpoints of code pairs: near-ties
        d = torch.sum(z ** 2, dim=1, keepdim=True) + torch.sum(cb ** 2, dim=1) - 2 * torch.einsum('bd,dn->bn', z, cb.t())
        ref = torch.argmin(d, dim=1)
        idx, zq = ops.vq_nearest(z.to(dev), cb.to(dev))
        same = int((idx.cpu() == ref).sum())
        assert same == z.shape[0], (e_dim, n_e, z.shape[0] - same)
        assert torch.equal(zq.cpu(), cb[ref])


def golden_vq_f4(dev, hip: bool, tol=1e-3):
    """tests/golden/vq_f4_small.pt = encode / code indices / decode of the REAL reference VQModel (oracle/make_golden_vq.py) at the
    VQ-f4 geometry (ch 128, ch_mult (1, 2, 4), 8192 x 3 codebook, single-head 512-channel AttnBlock at the 16x16 level), 64x64
    images.  ``hip``: bbdm_amd.first_stage_hip.VQModel on the HIP kernels, else the in-package PyTorch first stage."""
    import os
    from fixture_weights import synth_weights
    rec = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vq_f4_small.pt"))
    if hip:
        from bbdm_amd.first_stage_hip import VQModel
    else:
        from bbdm_amd.first_stage import VQModel
    m = VQModel(ddconfig=dict(rec["ddconfig"]), n_embed=rec["n_embed"], embed_dim=rec["embed_dim"]).eval()
    sd = synth_weights(rec["shapes"], rec["weight_seed"])
    sd["quantize.embedding.weight"] = rec["codebook"]
    m.load_state_dict(sd, strict=True)
    m = m.to(dev)
    x, z_ref = rec["x"].to(dev), rec["z"]
    with torch.no_grad():
        if hip:
            z = m.encode_latent(x, quant_conv=True)
            img, idx = m.decode_latent(z_ref.to(dev), return_indices=True)          # the golden latent: indices must be bit-equal
            _, idx_tie = m.decode_latent(rec["z_tie"].to(dev), return_indices=True)
        else:
            z = m.quant_conv(m.encoder(x))
            zq, _, (_, _, idx) = m.quantize(z_ref.to(dev))
            img = m.decode(zq)
            _, _, (_, _, idx_tie) = m.quantize(rec["z_tie"].to(dev))
    e_z, e_img = rel_err(z.cpu(), z_ref), rel_err(img.cpu(), rec["img"])
    print(f"VQ-f4 golden ({'HIP' if hip else 'PyTorch'} first stage): encode rel err {e_z:.2e}, decode rel err {e_img:.2e}")
    assert e_z < tol and e_img < tol
    assert torch.equal(idx.cpu().reshape(-1), rec["indices"])
    # near ties (latents on the midpoint between a code and its nearest other code, nudged by 1e-6 of their distance: below the fp32
    # resolution of d = z^2 + e^2 - 2 z.e, so the reference's own pick depends on its host's matmul rounding): the search must
    # return one of the two candidates, never a third code; how often it agrees with the pick recorded in the fixture is printed
    got, pairs = idx_tie.cpu().reshape(-1), rec["tie_pairs"]
    assert bool(((got == pairs[:, 0]) | (got == pairs[:, 1])).all())
    print(f"  near-tie latents: {int((got == rec['indices_tie']).sum())}/{got.numel()} picks equal to the reference run of the fixture")
