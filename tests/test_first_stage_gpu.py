"""VQGAN first stage on the HIP kernels, GPU: parity with the PyTorch first stage (CPU, fp32) on a small
config and on the real VQ-f4 geometry (Template-LBBDM-f4.yaml: ch 128, ch_mult (1,2,4), 256x256 -> 3x64x64, single-head
512-channel attention over 4096 tokens in the middle block); codebook indices bit-equal; timing against the same modules
run by PyTorch-ROCm."""
import pytest
import torch

import first_stage_cases as C
from fixtures import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_vq_indices_bit_exact(dev):
    C.vq_indices_bit_exact(dev)


def test_golden_of_the_real_reference_vqmodel(dev):
    """Encode / indices / decode against outputs of the reference's own VQModel (tests/golden/vq_f4_small.pt)."""
    C.golden_vq_f4(dev, hip=True)


def test_encode_decode_match_the_pytorch_first_stage(dev):
    C.encode_decode_parity(dev, N=3)


def test_vq_f4_geometry_and_rate(dev):
    over = dict(z_channels=3, resolution=256, ch=128, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=[])
    old = dict(C.DD)
    C.DD.update(over)
    try:
        import bbdm_amd.first_stage_hip as H
        torch.manual_seed(0)
        m = H.VQModel(ddconfig=dict(C.DD), n_embed=8192, embed_dim=3).eval()
        ref_sd = {k: v.clone() for k, v in m.state_dict().items()}
        ref = H.FS.VQModel(ddconfig=dict(C.DD), n_embed=8192, embed_dim=3).eval()
        ref.load_state_dict(ref_sd, strict=True)
        m = m.to(dev)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 3, 256, 256, generator=g).clamp(-1, 1)
        with torch.no_grad():
            z_ref = ref.quant_conv(ref.encoder(x))
            zq_ref, _, (_, _, idx_ref) = ref.quantize(z_ref)
            img_ref = ref.decode(zq_ref)
        z = m.encode_latent(x.to(dev))
        e_z = rel_err(z.cpu(), z_ref)
        img, idx = m.decode_latent(z_ref.to(dev), return_indices=True)
        e_img = rel_err(img.cpu(), img_ref)
        same = float((idx.cpu().flatten() == idx_ref.flatten()).float().mean())
        print(f"VQ-f4 geometry 256x256: encode rel err {e_z:.2e}, decode rel err {e_img:.2e}, codebook indices equal {same:.6f}")
        assert e_z < 1e-3 and e_img < 1e-3 and same == 1.0
        # rate: batch 16, HIP pipeline vs the same modules on PyTorch-ROCm
        xb = torch.randn(16, 3, 256, 256, device=dev).clamp(-1, 1)
        zb = m.encode_latent(xb)

        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        t_enc, t_dec = timed(lambda: m.encode_latent(xb)), timed(lambda: m.decode_latent(zb))
        with torch.no_grad():
            t_enc_t = timed(lambda: m.quant_conv(m.encoder(xb)))
            t_dec_t = timed(lambda: m.decode(m.quantize(zb)[0]))
        print(f"VQ-f4 batch 16 at 256x256: encode {t_enc:.1f} ms (PyTorch-ROCm {t_enc_t:.1f}), decode {t_dec:.1f} ms "
              f"(PyTorch-ROCm {t_dec_t:.1f})")
    finally:
        C.DD.clear()
        C.DD.update(old)
