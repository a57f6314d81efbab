"""Host logic of the execution plan (no GPU): the per-layer Winograd / direct choice, the consistency of the emitted
op list and the FLOP accounting bench.py reports.  Plans are built on CPU tensors -- construction only queries the
C-ABI's host-side size functions and allocates buffers; nothing is launched."""
import collections

import pytest
import torch

import bench
from bbdm_amd import _lib, unet


def _plan(workload, batch=None, training=False, winograd=4):
    desc, up, ch, size, n, *_ = bench.WORKLOADS[workload]
    m = unet.UNetModel(**up)
    m.winograd = winograd
    x = torch.zeros(batch or n, up["in_channels"], size, size)
    return m, m._plan_for(x, training)


def lib_tiles(m, N, H, W):
    return _lib.load().bbdm_winograd_tiles(m, N, H, W)


def test_winograd_tile_choice_follows_the_measured_crossovers():
    wt = unet.winograd_tile
    # profiles/r01_wino_bench.txt (MI355X): F(4x4) wins on every wide layer with >= 256 tiles ...
    assert wt(16, 64, 64, 1024, 1024, 4) == 4
    assert wt(16, 256, 256, 128, 128, 4) == 4          # 1.27x
    assert wt(16, 128, 128, 128, 512, 4) == 4          # 1.46x
    assert wt(4, 32, 32, 512, 512) == 4                # 256 tiles: 1.50x
    # ... loses below that on the f32-MFMA tile GEMM (padded GEMM tiles, launch-bound), where F(2x2) did not pay either (round 2) ...
    assert wt(4, 16, 16, 1024, 1024, small=False) == 0              # 0.55x / 0.98x
    assert wt(32, 4, 4, 1024, 1024, small=False) == 0
    assert wt(32, 8, 8, 512, 512, small=False) == 0
    # ... and takes F(2x2) on the bf16x3 pipe GEMM with 128-row tiles + split-K (profiles/r03_small_conv_bench.txt: direct vs F2)
    assert wt(4, 16, 16, 1024, 1024) == 2              # 0.183 vs 0.115 ms
    assert wt(32, 4, 4, 1024, 1024) == 2               # 0.151 vs 0.115 ms (128 tiles: the floor of the rule)
    assert wt(32, 8, 8, 512, 512) == 2                 # 0.111 vs 0.072 ms
    assert wt(32, 8, 8, 2048, 1024) == 2               # 0.701 vs 0.250 ms
    assert wt(32, 2, 2, 1024, 1024) == 0 and wt(32, 8, 8, 520, 512) == 0       # 32 tiles / K not whole 16-channel chunks
    # the stem / head / narrow outputs stay on the direct kernel
    assert wt(16, 256, 256, 8, 128) == 0 and wt(16, 256, 256, 128, 3) == 0 and wt(16, 64, 64, 64, 64) == 0
    # H, W not multiples of 4 -> F(2x2) if the layer is wide and large enough, else direct
    assert wt(16, 66, 66, 512, 512, 4) == 2 and wt(16, 66, 66, 128, 128, 4) == 0 and wt(16, 33, 33, 512, 512) == 0
    # profiles/r02_wino_bench.txt: F(6x6) beats F(4x4) by 1.1-1.25x with >= ~900 8x8 tiles and <= 10 % edge waste ...
    assert wt(16, 64, 64, 1024, 1024) == 6             # 2.63 vs 3.28 ms
    assert wt(16, 256, 256, 128, 128) == 6             # 1.23 vs 1.50 ms
    assert wt(8, 64, 64, 256, 256) == 6                # 968 tiles: 0.124 vs 0.166 ms
    assert wt(32, 64, 64, 128, 128) == 6
    assert wt(16, 66, 66, 512, 512) == 6               # H, W need not be multiples of 6
    # ... and loses on 32x32 / 16x16 latents (27 % of the 8x8 tiles' outputs fall outside the image)
    assert wt(32, 32, 32, 512, 512) == 4               # 0.475 vs 0.472 ms
    assert wt(32, 16, 16, 1024, 1024) == 4             # 0.640 vs 0.486 ms
    assert wt(8, 32, 32, 512, 512) == 4
    # the cap (UNetModel.winograd / BBDM_WINOGRAD)
    assert wt(16, 64, 64, 1024, 1024, 2) == 2 and wt(16, 64, 64, 1024, 1024, 0) == 0
    assert wt(16, 8, 8, 1024, 1024, 6, small=False) == 0
    tiny = unet.UNetModel(image_size=8, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1,
                          attention_resolutions=(), channel_mult=(1,), num_head_channels=32, condition_key="nocond")
    assert tiny.winograd == 8 and tiny.winograd_train8 == 2 and tiny.winograd_wgrad == 8
    # the default cap: F(8x8, 3x3) where a plan on the bf16x3 pipeline has the tiles for it (allow8) ...
    assert wt(16, 64, 64, 1024, 1024, 8, allow8=True) == 8 and wt(16, 256, 256, 128, 128, 8, allow8=True) == 8
    assert wt(16, 64, 64, 1024, 1024, 8) == 6 and wt(16, 64, 64, 1024, 1024, 6, allow8=True) == 6      # ... never without the pipeline / under a cap of 6
    assert wt(16, 64, 64, 1024, 1000, 8, allow8=True) == 6 and wt(16, 64, 64, 136, 128, 8, allow8=True) == 6     # whole 128-channel output blocks, 16-channel chunks
    assert wt(32, 32, 32, 512, 512, 8, allow8=True) == 8 and wt(4, 64, 64, 512, 512, 8, allow8=True) == 4     # 512 tiles / 256: too few
    assert lib_tiles(6, 16, 64, 64) == 2048 and lib_tiles(4, 16, 64, 64) == 4096 and lib_tiles(2, 3, 8, 12) == 256


def test_winograd_wgrad_tile_choice_follows_the_measured_crossovers():
    """profiles/r02_wgrad_bench.txt (MI355X, batch 32; direct / m = 4 / m = 6 in ms)."""
    wt = unet.winograd_wgrad_tile
    assert wt(32, 64, 64, 512, 512) == 6               # 4.82 / 2.17 / 1.69
    assert wt(32, 64, 64, 128, 128) == 6               # 0.51 / 0.46 / 0.38
    assert wt(32, 64, 64, 640, 128) == 6               # 1.69 / 0.89 / 0.77
    assert wt(32, 32, 32, 1536, 512) == 4              # 4.78 / 1.60 / 1.46: m = 6 only ties here (27 % edge waste) ...
    assert wt(32, 16, 16, 2048, 1024) == 4             # 2.67 / 1.31 / 1.40: ... and loses on 16x16
    assert wt(2, 64, 64, 128, 128) == 4                # 512 4x4 tiles but only 242 8x8 tiles
    assert wt(2, 16, 16, 1024, 1024) == 0              # too few tiles for the TN GEMMs' K: direct kernel
    assert wt(32, 64, 64, 8, 128) == 0 and wt(32, 64, 64, 128, 3) == 0      # stem / head
    assert wt(32, 64, 64, 512, 512, 4) == 4 and wt(32, 64, 64, 512, 512, 0) == 0     # the cap (BBDM_WINOGRAD_WGRAD)
    # m = 8 (allow8: the forward takes it and keeps the transposed planes): the forward's rule + whole 32-channel row groups of V^T
    assert wt(32, 64, 64, 512, 512, 8, allow8=True) == 8 and wt(32, 32, 32, 1536, 512, 8, allow8=True) == 8
    assert wt(32, 64, 64, 512, 512, 8) == 6 and wt(32, 64, 64, 512, 512, 6, allow8=True) == 6
    assert wt(32, 16, 16, 2048, 1024, 8, allow8=True) == 4 and wt(32, 64, 64, 144, 128, 8, allow8=True) == 6      # 128 tiles / cin % 32


@pytest.mark.parametrize("workload,batch,training,cap", [("c1", 4, False, 4), ("c1", 16, False, 4), ("c1", 16, True, 4),
                                                         ("c5", 32, False, 4), ("c1", 16, False, 6), ("c2", 2, False, 6),
                                                         ("c3", 32, True, 6), ("c2", 2, False, 8), ("c3", 32, False, 8),
                                                         ("c3", 32, True, 8)])
def test_plan_ops_are_consistent(workload, batch, training, cap):
    lib = _lib.load()
    m, plan = _plan(workload, batch, training, winograd=cap)
    ops = list(plan.ops) + (list(plan.bops) if training else [])
    names = collections.Counter(n for n, _ in ops)
    assert names["bbdm_winograd_input_f32"] == names["bbdm_winograd_gemm_f32"] == names["bbdm_winograd_output_f32"]
    for k, (name, a) in enumerate(ops):
        if name != "bbdm_winograd_input_f32":
            continue
        (n1, i), (n2, g), (n3, o) = ops[k], ops[k + 1], ops[k + 2]
        assert (n2, n3) == ("bbdm_winograd_gemm_f32", "bbdm_winograd_output_f32")       # emitted as a triple
        wm = i[0]
        N, H, W, cin = i[9:13]
        assert wm in (2, 4, 6, 7, 8) and g[0] == wm and o[0] == wm
        # m = 8 = F(8x8, 3x3): >= 512 tiles, whole 16-channel chunks in, whole 128-channel blocks out, under a cap of 8 (training plans:
        # UNetModel.winograd_train8)
        assert wm != 8 or (cap == 8 and (not training or m.winograd_train8) and cin % 16 == 0 and g[8] % 128 == 0
                           and unet.wino_tiles(8, N, H, W) >= 512
                           and not getattr(ops[k][0], "entry", "").endswith("_gn_f32"))
        assert (wm >= 6 or (H % wm == 0 and W % wm == 0)) and cin % 4 == 0
        phases = bool(o[7] & 8)          # conv3x3(nearest x2 (x)) as four phase filters on x: the GEMMs produce 4 Cout channels
        # m = 7 = F(7x7, 2x2): only the phase filters of an up-sampling conv, inference, on the pre-split planes, whole 128-channel
        # blocks per phase, and only where the coarser tile grid still saves GEMM work
        assert wm != 7 or (phases and not training and (g[8] // 4) % 128 == 0 and cin % 16 == 0 and
                           ("bf3p" in getattr(ops[k + 1][0], "entry", "") or "h2p" in getattr(ops[k + 1][0], "entry", "")) and
                           unet.wino_tiles(7, N, H, W) <= 0.9 * unet.wino_tiles(6, N, H, W))
        assert tuple(g[4:8]) == (N, H, W, cin) and tuple(o[8:11]) == (N, H, W) and (4 if phases else 1) * o[11] == g[8]
        cout = g[8]
        up = i[8]
        src = i[1]                                            # the view being convolved
        assert (src.H, src.W) == ((H // 2, W // 2) if up else (H, W)) and src.C == cin
        if phases:
            dst = o[5]
            assert not up and not training and (dst.H, dst.W, dst.C) == (2 * H, 2 * W, cout // 4) and o[3] is None
            assert unet.winograd_tile(N, H, W, cin, cout, m.winograd) >= min(6, unet.winograd_tile(N, 2 * H, 2 * W, cin, cout // 4, m.winograd,
                                                                                                   allow8=not training))
        tiles = lib.bbdm_winograd_tiles(wm, N, H, W)
        P = unet.wino_planes(wm)
        assert tiles % 256 == 0 and tiles >= unet.wino_tiles(wm, N, H, W)
        assert plan._wino_v.t.numel() >= P * tiles * cin and plan._wino_m.t.numel() >= P * tiles * cout
        entry = getattr(ops[k + 1][0], "entry", "")
        if "h2p" in entry:                      # two fp16 planes per operand under a bound (csrc/h2_split.h): inference only, 4 B per element
            in_entry = getattr(ops[k][0], "entry", "")
            assert m.gemm_h2 and (not training or m.gemm_h2_train >= (2 if k >= len(plan.ops) else 1))
            assert in_entry in ("bbdm_winograd_input_h2p_f32", "bbdm_winograd_input_h2p_gn_f32", "bbdm_winograd_input_h2p_tr_f32",
                                "bbdm_winograd_input_h2p_tr2_f32")
            if in_entry.endswith("_tr2_f32"):   # ... the weight gradient on the fp16 pair as well (gemm_h2_train = 3): V^T as two planes
                assert training and m.gemm_h2_train >= 3 and i[8] == 0 and i[13].t.numel() == lib.bbdm_gemm_h2p_tn_at_bytes(P, tiles, cin)
            vb = i[-1]                          # the bound of the transformed tensor: a slot of the plan, the same one in both launches
            assert isinstance(vb, unet._Plan._H2Ref) and g[-2] is vb
            if k >= len(plan.ops):              # data gradient: dY under its measured maximum -- the launch before the transform takes it
                assert training and vb.kind == "dy" and 0 <= vb.k < plan._h2_dy_slots and i[4] is None
                # ... measured by ONE pass earlier in the same layer's backward (right before, or ahead of its weight-gradient chain)
                prev = [pa for pn, pa in ops[max(0, k - 8):k] if pn == "bbdm_absmax_rows_f32" and pa[-1] is vb]
                assert len(prev) == 1 and prev[0][0] is i[1] and prev[0][3] == cin
            else:                               # forward: the GroupNorm bound of this layer's input
                assert vb.kind == "gn" and 0 <= vb.k < len(plan._h2_layers)
                gam, bet, fo, C, z = plan._h2_layers[vb.k]
                assert C == cin and z >= 1.0 and gam.numel() == cin and (fo == -1 or 0 <= fo <= plan.film_total - 2 * cin)
            if in_entry.endswith("_tr_f32"):    # training forward: the transposed copy stays bf16x3 (the weight gradient's operand)
                assert training and i[8] == 0 and i[13].t.numel() == lib.bbdm_gemm_bf3p_tn_at_bytes(P, tiles, cin)
            assert g[-1].t.numel() == 1 and g[-1].t.dtype == torch.float32          # max |U|, a device float owned by the packed weights
            if in_entry.endswith("_gn_f32"):
                assert tiles <= m.gn_in_transform and i[5] is None and i[6] == cin and len(i) == 21
            assert 4 * plan._wino_v.t.numel() >= lib.bbdm_gemm_h2p_a_bytes(P, tiles, cin) and cin % 16 == 0
            assert g[2].t.dtype == torch.uint8 and g[2].t.numel() == lib.bbdm_gemm_h2p_b_bytes(P, cin, cout)
            assert lib.bbdm_gemm_bf3p_supported(tiles, cin, cout)
        elif "bf3p" in entry:                   # V pre-split by the input transform (csrc/gemm_bf3p.hip): both ops, 6 B per element
            in_entry = getattr(ops[k][0], "entry", "")
            assert in_entry in ("bbdm_winograd_input_bf3p_f32", "bbdm_winograd_input_bf3p_tr_f32", "bbdm_winograd_input_bf3p_gn_f32")
            if in_entry.endswith("_gn_f32"):        # small inference layer: the transform forms the GroupNorm coefficients itself
                assert not training and tiles <= m.gn_in_transform and i[5] is None and i[6] == cin and len(i) == 20
                assert i[17] == src.H * src.W and i[18] == 32 and cin % 64 == 0            # HW of the normalised tensor, groups
            if in_entry.endswith("_tr_f32"):        # training forward of a layer whose weight gradient contracts the transposed planes
                assert training and i[8] == 0 and i[13].t.numel() == lib.bbdm_gemm_bf3p_tn_at_bytes(P, tiles, cin)
                assert lib.bbdm_gemm_bf3p_tn_supported(tiles, cin, cout)
            assert 4 * plan._wino_v.t.numel() >= lib.bbdm_gemm_bf3p_a_bytes(P, tiles, cin) and cin % 16 == 0
            assert g[2].t.dtype == torch.uint8 and g[2].t.numel() == lib.bbdm_gemm_bf3p_b_bytes(P, cin, cout)
            assert lib.bbdm_gemm_bf3p_supported(tiles, cin, cout)
            # a kept V is the transposed copy (an extra argument of the training forward's input transform), never the planes
            # the forward GEMM reads (the shared scratch)
            assert not any(v[0] is i[3] for v in plan._saved_V.values())
        elif entry.endswith("bf3_f32"):
            assert getattr(ops[k][0], "entry", "bbdm_winograd_input_f32") == "bbdm_winograd_input_f32"
            assert g[2].t.dtype == torch.int16 and g[2].t.numel() == lib.bbdm_gemm_bf3_packed_halfs(P, cin, cout)
            assert lib.bbdm_gemm_bf3_supported(tiles, cin, cout)
        else:
            assert g[2].t.numel() == lib.bbdm_winograd_packed_floats(wm, cout, cin)
        small = bool(m.gemm_bf3 and m.gemm_bf3p and m.winograd_small)
        assert (unet.phase_filter_tile(N, H, W, cin, cout, m.winograd, small) if phases else
                unet.winograd_tile(N, H, W, cin, cout, m.winograd, small=small, allow8=plan._allow8(1 if k >= len(plan.ops) else 2))) == wm
        if entry.endswith("splitk_f32"):        # small layer: split-K partials, added by the output transform of the same count
            ks = g[9]
            assert ks == lib.bbdm_winograd_gemm_bf3p_splits(wm, N, H, W, cin, cout) > 1
            assert getattr(ops[k + 2][0], "entry", "") == "bbdm_winograd_output_splitk_stats_f32" and ops[k + 2][1][-1] == ks
            assert plan._wino_m.t.numel() >= ks * (wm + 2) ** 2 * tiles * cout
    fused_gn = sum(1 for n, a in ops if n == "bbdm_winograd_input_f32" and a[4] is not None)
    folded = sum(1 for n, a in ops if getattr(n, "entry", "") in ("bbdm_winograd_input_bf3p_gn_f32", "bbdm_winograd_input_h2p_gn_f32"))
    # every fused GroupNorm has its coefficients from exactly one place: a bbdm_groupnorm_coeffs_f32 launch or its consumer's transform
    assert names["bbdm_groupnorm_coeffs_f32"] + folded >= fused_gn and (training or folded > 0 or workload in ("c2", "c3"))
    if (workload, batch, training) == ("c5", 32, False):
        assert folded == fused_gn > 0 and names["bbdm_groupnorm_coeffs_f32"] == 1       # (the head's direct kernel keeps its launch)
    if training:
        assert names["bbdm_conv_wgrad_f32"] > 0
        fwd_wino = sum(n == "bbdm_winograd_gemm_f32" for n, _ in plan.ops)
        bwd_wino = sum(n == "bbdm_winograd_gemm_f32" for n, _ in plan.bops)
        assert fwd_wino > 0 and bwd_wino > 0                  # forward and data-gradient convolutions both take it


def test_1x1_layers_take_the_kernel_their_size_asks_for():
    """1x1 convolutions / Linears of the forward plan: from BBDM_BF3_MIN_TILES output tiles of 256 x 128 on the wide bf16x3 kernels
    (bbdm_conv1x1_bf3_f32 / _bf3q_f32), below it on the small-problem kernel (bbdm_conv1x1_bf3s_f32: K a multiple of 64, plain NHWC
    output, no fused producer) -- same planes, same bits -- and with conv1x1_small off on the split-K f32 kernel as in round 3."""
    def one_by_ones(plan):
        out = []
        for name, a in plan.ops:
            if name == "bbdm_conv1x1_bf3_f32":
                out.append((getattr(name, "entry", name), a[8], a[9], a[10]))               # entry, pixels, cin, cout
            elif name == "bbdm_conv2d_nhwc_f32" and a[20] == 1:
                out.append((name, a[15] * a[16] * a[17], a[18], a[19]))
        return out
    m, plan = _plan("c5", 32)
    layers = one_by_ones(plan)
    assert len(layers) >= 20
    for entry, pixels, cin, cout in layers:
        tiles = (pixels // 256) * -(-cout // 128)
        if tiles >= m.bf3_min_tiles:
            assert entry in ("bbdm_conv1x1_bf3_f32", "bbdm_conv1x1_bf3q_f32", "bbdm_conv1x1_h2q_f32"), (entry, pixels, cin, cout)
        else:       # (round 6: on the fp16 pair where the input carries a bound -- skip projections, qkv, proj_out)
            assert entry in ("bbdm_conv1x1_bf3s_f32", "bbdm_conv1x1_h2s_f32") and cin % 64 == 0 and cout % 4 == 0, (entry, pixels, cin, cout)
    assert any(e == "bbdm_conv1x1_h2s_f32" and (px, ci, co) == (512, 1024, 3072) for e, px, ci, co in layers)     # the qkv projections
    # ... proj_out: the attention's output is a convex combination of value rows, bounded by bound(qkv) = GroupNorm bound x max row L1 of
    # the qkv weight + max |bias| -- one bbdm_h2_affine_bound_f32 launch per attention block, before its proj_out
    assert any(e == "bbdm_conv1x1_h2s_f32" and ci == co for e, px, ci, co in layers)
    names = [str(n) for n, _ in plan.ops]
    assert names.count("bbdm_h2_affine_bound_f32") == names.count("bbdm_attention_f32") > 0
    for k, n in enumerate(names):
        if n == "bbdm_attention_f32":
            aff = max(j for j in range(k) if names[j] == "bbdm_h2_affine_bound_f32")
            proj = next(j for j in range(k + 1, len(names)) if names[j] == "bbdm_conv1x1_bf3_f32")
            assert plan.ops[proj][1][-2] is plan.ops[aff][1][-1]            # proj_out's xbound is the slot that launch wrote
    m1 = unet.UNetModel(**bench.WORKLOADS["c5"][1])
    m1.winograd, m1.conv1x1_h2 = 4, False
    layers1 = one_by_ones(m1._plan_for(torch.zeros(32, bench.WORKLOADS["c5"][1]["in_channels"], 16, 16), False))
    assert [l[1:] for l in layers1] == [l[1:] for l in layers] and not any("h2" in str(e) for e, *_ in layers1)
    m0 = unet.UNetModel(**bench.WORKLOADS["c5"][1])
    m0.winograd, m0.conv1x1_small = 4, False
    plan0 = m0._plan_for(torch.zeros(32, bench.WORKLOADS["c5"][1]["in_channels"], 16, 16), False)
    layers0 = one_by_ones(plan0)
    assert len(layers0) == len(layers) and not any(e in ("bbdm_conv1x1_bf3s_f32", "bbdm_conv1x1_h2s_f32") for e, *_ in layers0)
    assert [l[1:] for l in layers0] == [l[1:] for l in layers]


def test_flop_accounting_evaluates_on_every_op_of_a_training_plan():
    """bench.py's per-launch pass calls _Plan._algorithmic_flops on every forward and gradient-plan op with the op's UNRESOLVED argument
    list: an entry point whose signature grew (bbdm_conv_wgrad_f32 gained ws_floats in ABI 21) shifts the indices it reads -- found on
    the GPU box in round 5, caught here since."""
    m, plan = _plan("c3", 32, True, winograd=6)
    for name, args in list(plan.ops) + list(plan.bops):
        fl = plan._algorithmic_flops(str(name), args)
        assert isinstance(fl, (int, float)) and fl >= 0, (name, fl)
    assert any(str(n) == "bbdm_conv_wgrad_f32" for n, _ in plan.bops)


def test_side_stream_band_admits_the_latent_f4_projections_only():
    """UNetModel.side_stream_min_macs / _max_macs (measured: C3 -2 %, C1 / C5 +3 %, C2 +1 %): with the defaults only the LBBDM-f4
    plans (sampling, and training: UNetModel.side_stream_train) fork their 1x1 skip projections; every fork is joined before the launch
    that adds it; in the gradient plan the forked launches are the projection's weight / data gradient on their own workspace."""
    for workload, batch, training, want in (("c3", 32, False, True), ("c5", 32, False, False), ("c1", 4, False, False),
                                            ("c3", 32, True, True)):
        m, plan = _plan(workload, batch, training, winograd=8)
        sides = plan._side_ranges
        assert bool(sides) == want, (workload, training, len(sides))
        last = 0
        for k0, k1, kj in sides:
            assert last <= k0 < k1 <= kj < len(plan.ops) and all(str(n) == "bbdm_conv1x1_bf3_f32" for n, _ in plan.ops[k0:k1])
            dest = plan.ops[k0][1][6]                                   # the projection's destination view ...
            assert any(a is dest for a in plan.ops[kj][1])               # ... is the residual the joining launch adds
            last = kj
        if training:
            proj = [r for r in plan._bside_ranges if str(plan.bops[r[0]][0]) == "bbdm_conv_wgrad_f32"]
            chains = [r for r in plan._bside_ranges if str(plan.bops[r[0]][0]) == "bbdm_winograd_dy_transform_bf3p_f32"]
            assert 0 < len(proj) <= len(sides) and len(proj) + len(chains) == len(plan._bside_ranges)
            for k0, k1, kj in proj:       # (a data gradient on the direct kernel keeps its block's projection on one stream)
                names = [str(n) for n, _ in plan.bops[k0:k1]]
                assert names == ["bbdm_conv_wgrad_f32", "bbdm_conv1x1_bf3_f32"] and str(plan.bops[kj][0]) == "bbdm_groupnorm_bwd_f32"
                assert plan.bops[k0][1][6] is plan._ws_f_side                   # its own workspace, never the shared one
                dxr = plan.bops[k1 - 1][1][6]                                    # the data gradient's destination ...
                assert any(a is dxr for a in plan.bops[kj][1])                   # ... is what the joining GroupNorm backward adds
            # the Winograd-domain weight gradients of the ResBlocks' 3x3 layers (UNetModel.side_stream_wgrad): dY transform -> TN GEMM ->
            # finish on the side workspace; every other user of a weight-gradient workspace keeps the shared one on stream one
            assert len(chains) >= 30
            for k0, k1, kj in chains:
                names = [str(n) for n, _ in plan.bops[k0:k1]]
                assert names[:2] == ["bbdm_winograd_dy_transform_bf3p_f32", "bbdm_gemm_bf3p_tn_f32"] and names[2].startswith("bbdm_winograd_wgrad_finish")
                assert str(plan.bops[kj][0]) == "bbdm_groupnorm_bwd_f32"
                assert all(a.t is plan._ws_f_side for a in plan.bops[k0][1][3:5]) and plan.bops[k0 + 1][1][1].t is plan._ws_f_side
            # every second-stream range of the gradient plan forks AND joins inside ONE backward segment (round-5 verdict, item 8): a
            # segment's autograd node hands its parameter gradients to autograd -- and DDP's reducer hooks -- when its last launch has
            # been enqueued, so a range that joined in a later segment would let the all-reduce read unfinished gradients
            segs = [(lo, hi) for lo, hi, _ in plan.bsegs]
            assert len(segs) >= 2 and segs[0][0] == 0 and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
            for k0, k1, kj in plan._bside_ranges:
                inside = [(lo, hi) for lo, hi in segs if lo <= k0 and kj < hi]
                assert len(inside) == 1 and k0 < k1 <= kj, (k0, k1, kj, segs)
            on_side = {j for k0, k1, _ in plan._bside_ranges for j in range(k0, k1)}
            for j, (n, a) in enumerate(plan.bops):
                if j not in on_side:
                    assert not any(getattr(v, "t", None) is plan._ws_f_side or v is plan._ws_f_side for v in a), (j, str(n))
            m.side_stream_train = False
            p1 = m._plan_for(torch.zeros(batch, 3, 64, 64), True)
            assert not p1._side_ranges and not p1._bside_ranges
    m, plan = _plan("c2", 16, False, winograd=8)                         # the benchmarked C2 batch: above the band
    assert not plan._side_ranges


def test_first_stage_flags_cover_every_switch_the_plan_emitters_read():
    """The VQGAN plans reuse _Plan's emitters with first_stage_hip._Flags standing in for the UNetModel: every switch an emitter reads
    (``self.m.<name>`` / ``m.<name>`` in unet.py's _Plan) must exist there, or the first plan of a first stage dies on a GPU box only."""
    import inspect
    import re
    from bbdm_amd import first_stage_hip
    src = inspect.getsource(unet._Plan)
    read = set(re.findall(r"self\.m\.([a-z_0-9]+)", src)) | set(re.findall(r"(?<![\w.])m\.([a-z_0-9]+)", src))
    switches = {n for n in read if hasattr(unet.UNetModel(**bench.WORKLOADS["c5"][1]), n)
                and not callable(getattr(unet.UNetModel, n, None)) and not isinstance(getattr(unet.UNetModel, n, None), property)}
    flags = first_stage_hip._Flags()
    model_only = {"out", "input_blocks", "middle_block", "output_blocks", "time_embed", "model_channels", "out_channels", "in_channels",
                  "context_dim", "training", "dtype", "num_heads", "num_head_channels"}
    missing = sorted(n for n in switches - model_only if not hasattr(flags, n) and not isinstance(getattr(unet.UNetModel(**bench.WORKLOADS["c5"][1]), n), torch.nn.Module))
    assert not missing, missing


def test_flop_accounting_direct_equivalent_matches_the_direct_plan():
    """bench.py reports `tflops_executed` (what the MFMA runs) and `tflops_algorithmic` (SURVEY.md §8d's direct
    count): a Winograd GEMM executing F FLOP stands for F * 9 m^2 / (m+2)^2 FLOP of direct convolution."""
    _, direct = _plan("c1", 16, winograd=0)
    assert not any(n.startswith("bbdm_winograd") for n, _ in direct.ops)
    want = sum(direct.op_flops)
    for cap in (2, 4, 6):                                     # 6: masked edge tiles (64 is not a multiple of 6)
        _, plan = _plan("c1", 16, winograd=cap)
        executed = sum(plan.op_flops)
        equiv = sum(18.0 * a[4] * a[5] * a[6] * getattr(a[2].t, "cin_true", a[7]) * a[8] if n == "bbdm_winograd_gemm_f32" else f
                    for (n, a), f in zip(plan.ops, plan.op_flops))
        assert executed < want and abs(equiv - want) < 1e-6 * want


def test_fused_producers_in_training_plans_only_where_the_forward_keeps_V():
    """Inference plans fold GN -> FiLM -> SiLU into the Winograd input transforms.  A training plan does so only for the layers
    whose weight gradient contracts the transformed input the forward keeps (same Winograd tile both ways); every other layer
    materialises the activation for its weight gradient."""
    _, inf = _plan("c1", 16)
    m, trn = _plan("c1", 16, training=True)
    fused = lambda p: sum(n == "bbdm_winograd_input_f32" and a[4] is not None for n, a in p.ops)
    assert fused(inf) > 0
    assert fused(trn) == len(trn._fused_train) == sum(n == "bbdm_groupnorm_coeffs_f32" for n, _ in trn.ops)
    assert trn._fused_train <= set(trn._saved_V)                      # fused => its V is kept
    n_staged = sum(n in ("bbdm_gemm_tn_batched_f32", "bbdm_gemm_bf3p_tn_f32") for n, _ in trn.bops)
    assert n_staged == len(trn._saved_V)                              # ... and the gradient plan uses it
    # with direct weight gradients nothing may be fused: the activation is what conv_wgrad_f32 reads
    desc, up, ch, size, n, *_ = bench.WORKLOADS["c1"]
    m0 = unet.UNetModel(**up)
    m0.winograd, m0.winograd_wgrad = 4, 0
    trn0 = m0._plan_for(torch.zeros(16, up["in_channels"], size, size), True)
    assert fused(trn0) == 0 and not trn0._fused_train and not trn0._saved_V
    assert sum(n == "bbdm_groupnorm_coeffs_f32" for n, _ in trn0.ops) == 0


def test_unsupported_options_fail_loudly_and_only_where_they_matter():
    """`num_classes`, `dims != 2`, `n_embed` raise at construction; `dropout > 0` only in train() mode (nn.Dropout is the identity
    in eval(): a checkpoint trained with dropout must sample)."""
    base = dict(image_size=8, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=(),
                channel_mult=(1,), num_head_channels=32, condition_key="nocond")
    for bad in (dict(num_classes=10), dict(dims=3), dict(n_embed=16)):
        with pytest.raises(NotImplementedError):
            unet.UNetModel(**base, **bad)
    m = unet.UNetModel(**base, dropout=0.1)
    m.eval()
    plan = m._plan_for(torch.zeros(2, 4, 8, 8), False)          # an inference plan builds: dropout is the identity there
    assert len(plan.ops) > 0
    m.train()
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 4, 8, 8), timesteps=torch.zeros(2, dtype=torch.int64))
