"""The kernel parity tests of test_kernels_gpu.py / test_backward_kernels_gpu.py, run on the CPU (no GPU needed).

The SAME kernel sources (bbdm_amd/csrc/*.hip) are compiled for the host against tools/hipemu -- a fiber-based emulation
of the HIP device model (64-lane wavefronts, __syncthreads, LDS, __shfl, the v_mfma_f32_32x32x2_f32 lane maps,
atomics) -- and driven through the same C-ABI by the same test bodies, on the smaller parameter sets.  This is how a
kernel change gets its first correctness check in the build container before GPU minutes are spent; it says nothing
about performance, and the `-m gpu` tests remain the parity tests proper."""
import pytest
import torch

import test_backward_kernels_gpu as BK
import test_kernels_gpu as K
from emu_backend import emulated_backend

CPU = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator():
    with emulated_backend() as emu:
        yield emu


SMALL_CONV = [c for c in K.CONV_CASES if c[0] * c[1] * c[2] * c[3] * c[4] * c[5] ** 2 <= 3e7]


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,res", SMALL_CONV)
def test_conv2d(N, H, W, Cin, Cout, ks, res):
    K.test_conv2d(CPU, N, H, W, Cin, Cout, ks, res)


SMALL_WINO = [c for c in K.WINO_CASES if c[1] * c[2] * c[3] * c[4] * c[5] <= 3e6]


@pytest.mark.parametrize("m,N,H,W,Cin,Cout,res", SMALL_WINO + [(6, 2, 12, 12, 64, 128, 0), (6, 1, 6, 6, 16, 8, 0),
                                                                (6, 3, 16, 20, 48, 72, 1)])
def test_conv3x3_winograd(m, N, H, W, Cin, Cout, res):
    K.test_conv3x3_winograd(CPU, m, N, H, W, Cin, Cout, res)


@pytest.mark.parametrize("m", [2, 4, 6])
def test_conv3x3_winograd_dgrad_and_slices(m):
    K.test_conv3x3_winograd_dgrad_and_slices(CPU, m)


@pytest.mark.parametrize("m,up,silu", [(2, 1, 1), (4, 0, 1), (4, 1, 0), (6, 1, 1)])
def test_winograd_stages_fused_producer(m, up, silu):
    K.test_winograd_stages_fused_producer(CPU, m, up, silu)


@pytest.mark.parametrize("batch,T,Cin,Cout", [(2, 256, 48, 72), (1, 512, 64, 132)])
def test_gemm_bf3_accuracy(batch, T, Cin, Cout):
    K.test_gemm_bf3_accuracy(CPU, batch, T, Cin, Cout)


@pytest.mark.parametrize("batch,T,Cin,Cout,extra", [(2, 256, 48, 72, 0), (1, 300, 32, 132, 2), (8, 256, 32, 40, 1)])
def test_gemm_bf3p_matches_bf3_bitwise(batch, T, Cin, Cout, extra):
    K.test_gemm_bf3p_matches_bf3_bitwise(CPU, batch, T, Cin, Cout, extra)


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", [(6, 2, 12, 20, 32, 40), (4, 1, 8, 16, 16, 24), (2, 3, 4, 6, 16, 8), (6, 1, 14, 10, 16, 128)])
def test_winograd_output_adds_upsampled_residual(m, N, H, W, Cin, Cout):
    K.test_winograd_output_adds_upsampled_residual(CPU, m, N, H, W, Cin, Cout)


@pytest.mark.parametrize("kernel", [4, 5, 7])
@pytest.mark.parametrize("batch,T,Cin,Cout", [(1, 256, 16, 256), (1, 256, 32, 256), (8, 256, 80, 260), (2, 512, 64, 256),
                                              (1, 768, 48, 128)])
def test_gemm_bf3p_kernel_variants(kernel, batch, T, Cin, Cout):
    K.test_gemm_bf3p_kernel_variants(CPU, kernel, batch, T, Cin, Cout)


@pytest.mark.parametrize("m,N,H,W,Cin,Cout,pre", [(6, 1, 7, 11, 16, 40, 1), (4, 1, 8, 8, 16, 32, 0), (2, 2, 4, 6, 16, 8, 1),
                                                  (6, 1, 7, 11, 16, 128, 1), (7, 1, 7, 11, 16, 128, 1), (7, 2, 6, 13, 16, 128, 0),
                                                  (7, 1, 14, 8, 32, 128, 1)])
def test_upsample_conv_as_phase_filters(m, N, H, W, Cin, Cout, pre):
    K.test_upsample_conv_as_phase_filters(CPU, m, N, H, W, Cin, Cout, pre)


@pytest.mark.parametrize("batch,T,rows,Cin,Cout,splits", [(2, 256, 96, 64, 72, 2), (1, 256, 256, 48, 132, 3), (8, 256, 32, 32, 40, 1)])
def test_gemm_bf3p_splitk(batch, T, rows, Cin, Cout, splits):
    K.test_gemm_bf3p_splitk(CPU, batch, T, rows, Cin, Cout, splits)


@pytest.mark.parametrize("m,N,H,W,Cin,Cout,splits", [(2, 2, 4, 4, 32, 40, 2), (4, 1, 8, 8, 48, 32, 3)])
def test_winograd_splitk_stages(m, N, H, W, Cin, Cout, splits):
    K.test_winograd_splitk_stages(CPU, m, N, H, W, Cin, Cout, splits)


@pytest.mark.parametrize("m,up,silu,N,H,W,Cin,Cout", [(6, 1, 1, 2, 12, 12, 32, 40), (4, 0, 1, 1, 8, 8, 16, 8),
                                                      (2, 1, 0, 1, 8, 8, 16, 24)])
def test_winograd_bf3p_stages(m, up, silu, N, H, W, Cin, Cout):
    K.test_winograd_bf3p_stages(CPU, m, up, silu, N, H, W, Cin, Cout)


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(256, 32, 40, False), (512, 64, 132, True)])
def test_conv1x1_bf3(pixels, Cin, Cout, res):
    K.test_conv1x1_bf3(CPU, pixels, Cin, Cout, res)


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", [(6, 2, 14, 20, 32, 64), (4, 3, 8, 12, 16, 128), (6, 9, 6, 6, 16, 32), (6, 2, 14, 20, 16, 128),
                                              (6, 3, 7, 9, 16, 256)])
def test_winograd_output_accumulates_groupnorm_statistics(m, N, H, W, Cin, Cout):
    K.test_winograd_output_accumulates_groupnorm_statistics(CPU, m, N, H, W, Cin, Cout)


def test_conv2d_accumulates_groupnorm_statistics():
    K.test_conv2d_accumulates_groupnorm_statistics(CPU)


@pytest.mark.parametrize("Cin", [8, 4])
def test_stem_conv_accumulates_groupnorm_statistics(Cin):
    K.test_stem_conv_accumulates_groupnorm_statistics(CPU, Cin)


def test_conv_rejections_and_slices():
    K.test_conv3x3_winograd_rejects_bad_shapes(CPU)
    K.test_conv2d_channel_slices(CPU)


@pytest.mark.parametrize("N,H,W,C", [(2, 8, 8, 96), (1, 4, 4, 192), (3, 4, 4, 1536), (1, 32, 32, 32)])
@pytest.mark.parametrize("mode", ["plain", "film_silu", "silu_pool", "silu_up"])
def test_groupnorm(N, H, W, C, mode):
    K.test_groupnorm(CPU, N, H, W, C, mode)


def test_resample_and_layout():
    K.test_resample_only(CPU)
    K.test_layout_roundtrip(CPU)


@pytest.mark.parametrize("N,T,heads,ch", [(2, 16, 4, 64), (2, 100, 3, 32), (3, 16, 2, 16), (2, 37, 1, 64), (1, 256, 2, 64)])
@pytest.mark.parametrize("new_order", [False, True])
def test_attention(N, T, heads, ch, new_order):
    K.test_attention(CPU, N, T, heads, ch, new_order)


@pytest.mark.parametrize("N,Tq,Tk,heads,ch", [(2, 64, 64, 2, 16), (1, 16, 64, 4, 16), (2, 100, 37, 3, 32)])
def test_cross_attention(N, Tq, Tk, heads, ch):
    K.test_cross_attention(CPU, N, Tq, Tk, heads, ch)


@pytest.mark.parametrize("rows,C", [(7, 64), (130, 32), (3, 260)])
def test_layernorm_and_geglu(rows, C):
    K.test_layernorm_and_geglu(CPU, rows, C)


@pytest.mark.parametrize("N,Tq,Tk,heads,ch", [(1, 100, 37, 2, 32), (1, 64, 160, 1, 64), (1, 33, 31, 1, 64)])
def test_attention_interleaved_loop_is_bit_equal(N, Tq, Tk, heads, ch):
    K.test_attention_interleaved_loop_is_bit_equal(CPU, N, Tq, Tk, heads, ch)


@pytest.mark.parametrize("N,T,heads,ch,new_order", [(1, 128, 2, 64, False), (1, 256, 1, 32, True)])
def test_attention_presplit_form_is_bit_equal(N, T, heads, ch, new_order):
    K.test_attention_presplit_form_is_bit_equal(CPU, N, T, heads, ch, new_order)


@pytest.mark.parametrize("N,T,heads,ch,new_order,slack", [(1, 128, 2, 64, False, 1.0), (1, 256, 1, 32, True, 4096.0)])
def test_attention_h2(N, T, heads, ch, new_order, slack):
    K.test_attention_h2(CPU, N, T, heads, ch, new_order, slack)


def test_h2_projection_bound():
    K.test_h2_projection_bound(CPU)


@pytest.mark.parametrize("N,H,W,Cin,Cout,pre,up,res,f32v", [(1, 13, 11, 32, 128, 1, 0, 1, 0), (1, 8, 16, 32, 128, 1, 1, 2, 0),
                                                             (1, 9, 8, 32, 128, 0, 0, 0, 1), (1, 13, 11, 32, 128, 1, 0, 1, 2),
                                                             (1, 8, 16, 32, 128, 1, 1, 2, 2)])
def test_winograd_f8_forward(N, H, W, Cin, Cout, pre, up, res, f32v):
    K.test_winograd_f8_forward(CPU, N, H, W, Cin, Cout, pre, up, res, f32v)


# ---- the fp16-pair planes (csrc/h2_split.h) ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("batch,T,Cin,Cout,extra,decades", [(2, 256, 48, 72, 0, 3), (1, 300, 32, 132, 2, 3), (9, 300, 16, 72, 1, 6),
                                                            (12, 512, 16, 132, 2, 0)])
def test_gemm_h2p_shapes_and_dynamic_range(batch, T, Cin, Cout, extra, decades):
    K.test_gemm_h2p_shapes_and_dynamic_range(CPU, batch, T, Cin, Cout, extra, decades)


@pytest.mark.parametrize("kernel,Cin", [(4, 16), (4, 32), (4, 48), (4, 64), (4, 80), (4, 112), (5, 48), (7, 48)])
def test_gemm_h2p_kernel_variants(kernel, Cin):
    K.test_gemm_h2p_kernel_variants(CPU, kernel, Cin)


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(256, 32, 40, False), (300, 48, 256, True), (96, 16, 8, False)])
def test_conv1x1_h2q(pixels, Cin, Cout, res):
    K.test_conv1x1_h2q(CPU, pixels, Cin, Cout, res)


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(512, 64, 132, True), (96, 128, 8, False), (100, 192, 72, True), (64, 64, 64, False)])
def test_conv1x1_h2s(pixels, Cin, Cout, res):
    K.test_conv1x1_h2s(CPU, pixels, Cin, Cout, res)


def test_h2_stats_bound():
    K.test_h2_stats_bound(CPU)


def test_gemm_h2p_bound_is_respected():
    K.test_gemm_h2p_bound_is_respected(CPU)


@pytest.mark.parametrize("m,up,silu,film,N,H,W,C", [(2, 0, 1, True, 3, 4, 4, 64), (4, 1, 1, False, 2, 16, 8, 64), (6, 0, 0, True, 1, 14, 8, 64),
                                                    (8, 0, 1, True, 1, 16, 8, 64)])
def test_h2_bounds_and_planes_of_a_groupnorm_fed_layer(m, up, silu, film, N, H, W, C):
    K.test_h2_bounds_and_planes_of_a_groupnorm_fed_layer(CPU, m, up, silu, film, N, H, W, C)


@pytest.mark.parametrize("batch,T,Cin,Cout,extra", [(12, 512, 16, 132, 2), (9, 300, 16, 72, 1), (10, 256, 16, 40, 0), (11, 256, 16, 40, 0)])
def test_gemm_bf3p_shared_last_group(batch, T, Cin, Cout, extra):
    K.test_gemm_bf3p_matches_bf3_bitwise(CPU, batch, T, Cin, Cout, extra)


@pytest.mark.parametrize("kernel", [4, 6])
def test_gemm_bf3p_ragged_rows_read_the_padding(kernel):
    K.test_gemm_bf3p_ragged_rows_read_the_padding(CPU, kernel)


def test_attention_forces_rescale():
    K.test_attention_forces_rescale(CPU)


@pytest.mark.parametrize("N", [1, 4, 33])
def test_embedding_path(N):
    K.test_embedding_path(CPU, N)


@pytest.mark.parametrize("objective", ["grad", "noise", "ysubx"])
def test_bridge_arithmetic(objective):
    K.test_bridge_arithmetic(CPU, objective)


@pytest.mark.parametrize("N,H,W,C,Cout,ks,film,silu", [(2, 16, 16, 128, 64, 3, True, True), (3, 8, 8, 96, 128, 3, False, True),
                                                        (1, 64, 64, 64, 3, 3, False, True), (2, 72, 88, 32, 3, 3, True, True)])
def test_conv_with_fused_groupnorm_producer(N, H, W, C, Cout, ks, film, silu):
    K.test_conv_with_fused_groupnorm_producer(CPU, N, H, W, C, Cout, ks, film, silu)


# ---- backward kernels -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Cin,Cout,ks", [(2, 16, 16, 32, 64, 3), (3, 8, 8, 128, 96, 3), (2, 16, 16, 64, 192, 1),
                                               (1, 13, 9, 8, 3, 3), (2, 8, 8, 4, 32, 3), (1, 16, 16, 96, 68, 1), (1, 72, 60, 64, 68, 1)])
def test_conv_backward(N, H, W, Cin, Cout, ks):
    BK.test_conv_backward(CPU, N, H, W, Cin, Cout, ks)


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", [c for c in BK.WINO_WGRAD if c[1] * c[2] * c[3] * c[4] * c[5] <= 6e6])
def test_winograd_wgrad(m, N, H, W, Cin, Cout):
    BK.test_winograd_wgrad(CPU, m, N, H, W, Cin, Cout)


@pytest.mark.parametrize("batch,K,M,N", [(3, 40, 64, 36), (2, 1000, 256, 128), (4, 512, 132, 260)])
def test_gemm_tn_batched(batch, K, M, N):
    BK.test_gemm_tn_batched(CPU, batch, K, M, N)


@pytest.mark.parametrize("N,H,W,C", [(2, 4, 4, 1536), (2, 8, 8, 96), (3, 16, 16, 32)])
@pytest.mark.parametrize("mode", ["plain", "film_silu", "silu_pool", "silu_up", "silu_add_acc"])
def test_groupnorm_backward(N, H, W, C, mode):
    BK.test_groupnorm_backward(CPU, N, H, W, C, mode)


def test_resample_only_backward():
    BK.test_resample_only_backward(CPU)


@pytest.mark.parametrize("N,T,heads,ch", [(2, 16, 4, 64), (2, 100, 3, 32), (3, 16, 2, 16)])
@pytest.mark.parametrize("new_order", [False, True])
def test_attention_backward(N, T, heads, ch, new_order):
    BK.test_attention_backward(CPU, N, T, heads, ch, new_order)


@pytest.mark.parametrize("N,In,Out,act", [(4, 128, 512, False), (33, 96, 70, True)])
def test_linear_backward(N, In, Out, act):
    BK.test_linear_backward(CPU, N, In, Out, act)


@pytest.mark.parametrize("N,Tq,Tk,heads,ch", [(2, 64, 64, 2, 32), (1, 200, 77, 4, 64), (2, 130, 260, 1, 16)])
def test_cross_attention_backward(N, Tq, Tk, heads, ch):
    BK.test_cross_attention_backward(CPU, N, Tq, Tk, heads, ch)


@pytest.mark.parametrize("rows,C,with_add", [(37, 64, False), (512, 320, True)])
def test_layernorm_backward(rows, C, with_add):
    BK.test_layernorm_backward(CPU, rows, C, with_add)


def test_geglu_backward():
    BK.test_geglu_backward(CPU, 33, 32)


@pytest.mark.parametrize("m,N,H,W,Cin,Cout", [(2, 2, 8, 8, 32, 24), (4, 1, 8, 12, 32, 64), (6, 1, 7, 10, 32, 8), (6, 2, 12, 12, 64, 132),
                                              (8, 1, 8, 16, 32, 8), (8, 2, 13, 18, 64, 132)])
def test_winograd_wgrad_bf3p(m, N, H, W, Cin, Cout):
    BK.test_winograd_wgrad_bf3p(CPU, m, N, H, W, Cin, Cout)


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks", [(4, 32, 32, 64, 72, 1), (1, 72, 60, 96, 256, 1)])
def test_conv1x1_wgrad_planes_path(N, H, W, Cin, Cout, ks):
    """Option "wgrad1x1_bf3" = 2 / 0: the 1x1 weight gradients on the transposing split pass + bbdm_gemm_bf3p_tn_f32, and always on
    gemm_tn_f32, on the emulator."""
    BK.test_conv1x1_wgrad_planes_path(CPU, N, H, W, Cin, Cout, ks)


def test_conv_wgrad_checks_its_workspace():
    """A short workspace: fall back from the plane path / refuse, on the emulator (ABI 21)."""
    import kernel_ops as ops
    from bbdm_amd import _lib
    N, H, W, Cin, Cout = 4, 32, 32, 64, 72
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g)
    dy = torch.randn(N, H, W, Cout, generator=g)
    lib = _lib.load()
    with _lib.option("wgrad1x1_bf3", 0):
        small = lib.bbdm_conv_wgrad_workspace_floats(N, H, W, Cin, Cout, 1)
        want, wantb = ops.conv_wgrad(x, dy, Cin, Cout, 1, with_bias=True)
    with _lib.option("wgrad1x1_bf3", 2):
        full = lib.bbdm_conv_wgrad_workspace_floats(N, H, W, Cin, Cout, 1)
        assert small < full
        got, gotb = ops.conv_wgrad(x, dy, Cin, Cout, 1, with_bias=True, ws_floats=small)
        assert torch.equal(got, want) and torch.equal(gotb, wantb)
        with pytest.raises(_lib.BBDMHipError, match="workspace"):
            ops.conv_wgrad(x, dy, Cin, Cout, 1, ws_floats=64)
    with pytest.raises(_lib.BBDMHipError, match="workspace"):
        ops.conv_wgrad(x[:1, :8, :8].contiguous(), dy[:1, :8, :8, :32].contiguous(), 64, 32, 3, ws_floats=16)


@pytest.mark.parametrize("m,up,silu,N,H,W,Cin,Cout", [(6, 1, 1, 2, 20, 12, 32, 136), (4, 1, 0, 1, 8, 8, 16, 8)])
def test_winograd_input_64bit_index_variant(m, up, silu, N, H, W, Cin, Cout):
    """Option "wino_idx64": the 64-bit row-address instantiation of the input transform on the emulator."""
    K.test_winograd_input_64bit_index_variant(CPU, m, up, silu, N, H, W, Cin, Cout)


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(256, 32, 40, False), (512, 64, 132, True), (320, 48, 256, True), (96, 16, 8, False)])
def test_conv1x1_bf3q_bitwise(pixels, Cin, Cout, res):
    K.test_conv1x1_bf3q_bitwise(CPU, pixels, Cin, Cout, res)


@pytest.mark.parametrize("m,up,silu,film,N,H,W,C", [(2, 0, 1, True, 3, 4, 4, 64), (4, 1, 1, False, 2, 16, 8, 64), (6, 0, 0, True, 2, 14, 20, 64)])
def test_winograd_input_forms_groupnorm_coefficients_bitwise(m, up, silu, film, N, H, W, C):
    K.test_winograd_input_forms_groupnorm_coefficients_bitwise(CPU, m, up, silu, film, N, H, W, C)


@pytest.mark.parametrize("pixels,Cin,Cout,res", [(128, 256, 132, True), (96, 128, 8, False), (100, 192, 520, True), (64, 64, 64, False)])
def test_conv1x1_bf3s_bitwise(pixels, Cin, Cout, res):
    K.test_conv1x1_bf3s_bitwise(CPU, pixels, Cin, Cout, res)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(1, 8, 16, 128, 16), (1, 13, 10, 128, 32)])
def test_winograd_f8_dgrad(N, H, W, Cin, Cout):
    K.test_winograd_f8_dgrad(CPU, N, H, W, Cin, Cout)


@pytest.mark.parametrize("m,Cout,Cin,in_pad,dgrad", [(4, 96, 40, 48, False), (2, 24, 16, 32, True), (6, 40, 130, 144, False)])
def test_winograd_weight_planes_fused_bitwise(m, Cout, Cin, in_pad, dgrad):
    K.test_winograd_weight_planes_fused_bitwise(CPU, m, Cout, Cin, in_pad, dgrad)


@pytest.mark.parametrize("N,In,Out,act_in,act_out", [(5, 128, 2090, True, False), (33, 64, 2056, True, True), (16, 128, 40, False, False)])
def test_linear_on_the_matrix_core(N, In, Out, act_in, act_out):
    K.test_linear_on_the_matrix_core(CPU, N, In, Out, act_in, act_out)


@pytest.mark.parametrize("N,In,Out,act_in,act_out", [(4, 512, 200, True, False), (17, 64, 40, False, True), (32, 96, 33, True, True),
                                                     (1, 32, 32, False, False)])
def test_linear_on_packed_weights_bitwise(N, In, Out, act_in, act_out):
    K.test_linear_on_packed_weights_bitwise(CPU, N, In, Out, act_in, act_out)
