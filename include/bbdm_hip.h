/*
 * bbdm_hip.h  --  C-ABI of libbbdm_hip.so: the MI355X (gfx950) hot path of xuekt98/BBDM.
 *
 * The reference has no FFI / operator registry (it is pure PyTorch); the seam is the Python model object the
 * runner constructs (runners/DiffusionBasedModelRunners/BBDMRunner.py:21-29).  bbdm_amd/ mirrors that object
 * and calls the entry points below through ctypes.  Every entry point replaces one ATen call site (or a fused
 * run of them) of the reference; the file:line it replaces is cited on each declaration (paths relative to the
 * reference root; "openaimodel.py" = model/BrownianBridge/base/modules/diffusionmodules/openaimodel.py,
 * "util.py" = .../diffusionmodules/util.py, "BBM.py" = model/BrownianBridge/BrownianBridgeModel.py).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, floats, a hipStream_t passed as void*.  No torch types.
 *   - the library never allocates, frees or synchronises; kernels are enqueued on `stream`; all outputs and
 *     workspaces are caller-owned (torch.empty on the Python side).  Stateless and re-entrant.
 *   - return value: 0 = enqueued; <0 = BBDM_E_* (nothing was launched).  The Python shim raises RuntimeError.
 *   - activations are fp32 NHWC: element (n, h, w, c) of a tensor with row pitch `ld` (floats per pixel,
 *     ld >= C) lives at  base[((n*H + h)*W + w)*ld + c].  A pitch larger than C addresses a channel slice of a
 *     wider buffer, which is how th.cat((h, skip), dim=1) (openaimodel.py:742,752) is made copy-free.
 *   - all arithmetic is fp32 (the reference's precision; SURVEY.md §5 "Mixed precision: absent").
 */
#ifndef BBDM_HIP_H
#define BBDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBDM_OK 0
#define BBDM_E_BADARG (-1)   /* unsupported shape / misaligned pointer / bad enum */
#define BBDM_E_LAUNCH (-2)   /* hipLaunchKernel reported an error             */

/* GroupNorm statistics accumulator (layout and rationale: the GroupNorm section below, csrc/stats_acc.h) */
typedef void bbdm_stats_t;

/* ---- library ---------------------------------------------------------------------------------------- */
int bbdm_version(void);                       /* ABI version, bumped on any signature change           */
const char* bbdm_last_error(void);            /* text of the last error on the calling thread          */

int bbdm_device_cus(void);                    /* compute units of the current device (256): persistent grids */

/* ---- layout ------------------------------------------------------------------------------------------ */
/* NCHW [N,Ca,H,W] (+ optional second NCHW [N,Cb,H,W]) -> NHWC [N,H,W,ldo], channels >= Ca+Cb zero-filled up to
 * Cpad.  Replaces th.cat([x, context], dim=1) + the implicit NCHW read of the first conv
 * (openaimodel.py:741-745).  b may be NULL (Cb = 0). */
int bbdm_nchw_to_nhwc_f32(const float* a, int Ca, const float* b, int Cb, float* out, int ldo, int Cpad,
                          int N, int H, int W, void* stream);
/* NHWC (pitch ldx) -> NCHW contiguous.  The UNet output returns to the reference's layout (openaimodel.py:759). */
int bbdm_nhwc_to_nchw_f32(const float* x, int ldx, float* out, int N, int H, int W, int C, void* stream);

/* ---- convolution (openaimodel.py:207,233,244,524,690; conv1d k=1 at :307,315 is the ks=1 case) ------- */
/* Number of floats of the packed weight buffer for a [Cout, Cin, ks, ks] filter whose input tensor carries
 * CinPad >= Cin channels (CinPad % 4 == 0). */
size_t bbdm_conv_packed_floats(int Cout, int CinPad, int ks);
/* OIHW fp32 (the reference's state_dict layout) -> packed [tap][Cin-chunk][CoutPad][16] with zero padding.
 * Runs on `stream`; call again whenever the weight changes (optimizer step, EMA swap, load_state_dict). */
int bbdm_conv_pack_weight_f32(const float* w_oihw, float* packed, int Cout, int Cin, int CinPad, int ks,
                              void* stream);
/* out = conv_ks(x) + bias (+ residual).  ks in {1,3}; stride 1, padding ks/2 (F.conv2d semantics).
 * x: NHWC pitch ldx with CinPad channels; residual (may be NULL): NHWC pitch ldr, Cout channels, may alias out;
 * out: NHWC pitch ldo.  flags: BBDM_CONV_OUT_NCHW = write NCHW contiguous instead (the UNet head);
 * BBDM_CONV_RES_PER_IMAGE = residual is [N][ldr], one row per image broadcast over its pixels (h + emb_out[..., None,
 * None] of the use_scale_shift_norm=False ResBlock, openaimodel.py:275).  Implicit GEMM on v_mfma_f32_32x32x2_f32. */
#define BBDM_CONV_OUT_NCHW 1
#define BBDM_CONV_RES_PER_IMAGE 2
#define BBDM_CONV_RES_UPSAMPLE 4   /* Winograd output transform only: residual is [N][H/2][W/2][ldr], added nearest-upsampled x2 (the
                                      skip path x_upd(x) of an up-sampling ResBlock, openaimodel.py:259-264) */
#define BBDM_CONV_OUT_PHASES 8     /* Winograd output transform only: the layer is conv3x3(nearest x2 (x)) (Upsample.forward,
                                      openaimodel.py:111-121) run as FOUR phase filters on the low-resolution x: H, W are x's; M carries
                                      4 Cout channels, channel (2a + b) Cout + co at (i, j) is output pixel (2i + a, 2j + b), channel co, of
                                      the [N][2H][2W][ldo] result -- the input transform and the GEMM's A operand are 4x smaller than on
                                      the upsampled tensor, the GEMM's work is the same.  The phase filters: bbdm_upsample_phase_weights_f32 */
/* ws (may be NULL) / ws_floats: scratch for split-K.  When the output tiles alone cannot fill the 256 CUs (small
 * latents: LBBDM-f16 runs the 1024-channel layers on 4x4 images) the Cin reduction is spread over extra workgroups
 * whose partial sums are added in a fixed order by a second kernel (deterministic).  Size it with
 * bbdm_conv_splitk_workspace_floats() (0 = this shape never splits). */
size_t bbdm_conv_splitk_workspace_floats(int N, int H, int W, int CinPad, int Cout, int ks);
/* pre_scale / pre_bias (both NULL or both set): fp32 [N][pre_ld] per-image, per-input-channel coefficients of a fused
 * producer: the kernel convolves act(x * pre_scale[n][c] + pre_bias[n][c]) (act = SiLU when pre_silu) instead of x,
 * applying the transform while the input patch is staged into LDS.  With the coefficients from
 * bbdm_groupnorm_coeffs_f32 this is GroupNorm32 -> [FiLM] -> [SiLU] -> conv (openaimodel.py:205-207,229-233,306-307,
 * 688-690) without writing the normalised tensor to HBM; zero padding applies to the activated tensor, as in F.conv2d. */
int bbdm_conv2d_nhwc_f32(const float* x, int ldx, const float* packed_w, const float* bias,
                         const float* residual, int ldr, float* out, int ldo, int flags, float* ws, size_t ws_floats,
                         const float* pre_scale, const float* pre_bias, int pre_ld, int pre_silu,
                         int N, int H, int W, int CinPad, int Cout, int ks, void* stream);

/* The same convolution with the GroupNorm statistics of its OUTPUT accumulated in the epilogue (every conv output of the
 * UNet is normalised next: openaimodel.py:205,229,306,688) instead of by a separate bbdm_groupnorm_stats_f32 pass that
 * re-reads the tensor.  stats0 / stats1 (either may be NULL): accumulators (bbdm_stats_t, G = 32: sum, sum of squares per image
 * and 32-group index) of up to two consumers of `out` -- the next block's GroupNorm and, through the copy-free concat, an
 * output block's; cpg = channels per group of that consumer, coff = channel offset of `out` inside the consumer's tensor.
 * The caller zeroes them; the kernel adds.  Only where bbdm_conv_stats_fusable() says so (no split-K, one image per tile). */
int bbdm_conv_stats_fusable(int N, int H, int W, int CinPad, int Cout, int ks);
int bbdm_conv2d_nhwc_stats_f32(const float* x, int ldx, const float* packed_w, const float* bias,
                               const float* residual, int ldr, float* out, int ldo, int flags, float* ws, size_t ws_floats,
                               const float* pre_scale, const float* pre_bias, int pre_ld, int pre_silu,
                               int N, int H, int W, int CinPad, int Cout, int ks, bbdm_stats_t* stats0, int cpg0, int coff0,
                               bbdm_stats_t* stats1, int cpg1, int coff1, void* stream);

/* ---- 3x3 convolution through Winograd F(m x m, 3x3), m = 2, 4, 6 or 8 (same call sites, wide layers) */
/* Y = A^T[(G g G^T) (.) (B^T d B)]A: (m+2)^2 multiplies per m^2 outputs instead of 9 m^2 -- 2.25x (m = 2) or 4x (m = 4)
 * fewer MFMA FLOP; the choice cuDNN / MIOpen make for the reference's wide 3x3 layers (openaimodel.py:207,233,524;
 * their fp32 "Winograd non-fused" is m = 4).  stride 1, padding 1, H and W multiples of m, CinPad % 4 == 0,
 * Cout % 4 == 0.  Three launches: input transform -> (m+2)^2 batched GEMMs on the fp32 MFMA (conv_igemm_f32 in 1x1
 * mode) -> output transform (+ bias, + residual).  fp32 throughout; rounding error vs an fp64 convolution (rms / max,
 * Cin = 512): direct 2e-7 / 3e-7, m = 2: 5e-7 / 6e-7, m = 4: 3e-6 / 1e-5.
 * packed_wino: bbdm_winograd_packed_floats() floats filled by bbdm_winograd_pack_weight_f32 (dgrad != 0 packs the
 * data-gradient convolution Cout -> Cin of the same filter: transposed + flipped; then InPad is the channel count of
 * dY and the forward entry is called with CinPad = InPad, Cout = Cin).
 * ws: bbdm_winograd_workspace_floats() floats (transformed input V[(m+2)^2][tiles][CinPad] + products
 * M[(m+2)^2][tiles][Cout]).  flags: only BBDM_CONV_RES_PER_IMAGE.
 * m = 6 (8x8 tiles, 64 transform points; H, W arbitrary -- edge tiles are masked; CinPad, Cout multiples of 4) is accepted
 * by every entry below (measured in round 2: 1.1-1.25x faster than m = 4 on layers with >= ~1000 tiles and H, W >= 64).
 * m = 8 (round 5, ABI 22: 10x10 tiles, 100 transform points {0, +-1/2, +-3/4, +-4/3, +-2, inf}; H, W arbitrary) is the
 * tile of the large layers: 1.56 multiplies per output instead of m = 6's 1.78, at ~7x its rounding error (rms 4e-5 against
 * 6e-6 at Cin = 256 with every stage in fp32).  B^T and A^T are exact in fp32 by construction; the weight transform runs in fp64.
 * Accepted by pack_weight, pack_weight_bf3p (ABI 23: also dgrad = 1), tiles, workspace_floats, input (CinPad % 32 == 0), input_bf3p
 * (CinPad % 16 == 0; no coefficient folding), the three tile-GEMM entries and output (Cout % 128 == 0, no split-K partials).
 * ABI 23, the gradient side on the pre-split bf16x3 pipeline: bbdm_winograd_input_bf3p_tr_f32 (the transposed planes of V),
 * bbdm_winograd_dy_transform_bf3p_f32 (A dY A^T on ten points; dm11 then holds the tile sums of dY directly -- the point set has no
 * x = 1) and bbdm_winograd_wgrad_finish[_bias]_f32 (G^T dU G in fp64, rounded once) take m = 8; the fp32 gradient entry points
 * (bbdm_winograd_dy_transform_f32, bbdm_conv3x3_winograd_wgrad_f32) keep m <= 6. */
size_t bbdm_winograd_packed_floats(int m, int Cout, int CinPad);
int bbdm_winograd_pack_weight_f32(int m, const float* w_oihw, float* packed, int Cout, int Cin, int InPad, int dgrad,
                                  void* stream);
/* The same transform written straight into gemm_bf3p's B planes (bbdm_winograd_pack_weight_f32 followed by bbdm_gemm_bf3p_pack_b_f32
 * without the fp32 U tensor in between; the same values up to the FMA contraction of G g G^T, <= 1 ulp): the per-optimizer-step weight preparation of the training path
 * (nn.Conv2d weights, openaimodel.py:207,233,244).  b_planes: bbdm_gemm_bf3p_b_bytes((m+2)^2, InPad, dgrad ? Cin : Cout); InPad % 16 == 0. */
int bbdm_winograd_pack_weight_bf3p_f32(int m, const float* w_oihw, void* b_planes, int Cout, int Cin, int InPad, int dgrad,
                                       void* stream);
size_t bbdm_winograd_workspace_floats(int m, int N, int H, int W, int CinPad, int Cout);
int bbdm_conv3x3_winograd_f32(int m, const float* x, int ldx, const float* packed_wino, const float* bias,
                              const float* residual, int ldr, float* out, int ldo, int flags, float* ws,
                              int N, int H, int W, int CinPad, int Cout, void* stream);
/* The three stages on their own (what bbdm_conv3x3_winograd_f32 chains; a static plan calls them directly so that it
 * can share V / M across layers, time each stage and fold the producer of the convolved tensor into stage 1).
 * tiles = bbdm_winograd_tiles(m, N, H, W) = N ceil(H/m) ceil(W/m) rounded up to whole 256-row GEMM tiles.
 *   input : x -> V[(m+2)^2][tiles][CinPad].  pre_scale / pre_bias / pre_ld / pre_silu: same fused GroupNorm [-> FiLM]
 *           [-> SiLU] producer as bbdm_conv2d_nhwc_f32.  upsample != 0: x is [N, H/2, W/2, ldx] and the convolved tensor
 *           is its nearest x2 upsampling (Upsample.forward, openaimodel.py:111-121) -- never materialised.
 *   gemm  : M[xi] = V[xi] . U[xi] for the (m+2)^2 transform points, one launch.
 *   output: M[(m+2)^2][tiles][Cout] -> out NHWC (+ bias, + residual; flags: BBDM_CONV_RES_PER_IMAGE or BBDM_CONV_RES_UPSAMPLE). */
size_t bbdm_winograd_tiles(int m, int N, int H, int W);
/* The four phase filters of conv3x3(nearest x2 (x)) (Upsample.forward / the up-sampling ResBlock, openaimodel.py:111-121,259-264) as one
 * 3x3 convolution Cin -> 4 Cout on the low-resolution x (see BBDM_CONV_OUT_PHASES): w4 [4 Cout][Cin][3][3] from w [Cout][Cin][3][3].
 * Per axis the taps collapse to [w0, w1 + w2, 0] (phase 0) / [0, w0 + w1, w2] (phase 1): one fp32 addition per collapsed tap.
 * m = 7 (round 5) runs those four filters as the 2 x 2 filters they are, F(7x7, 2x2) on the eight transform points of F(6x6, 3x3): 49
 * outputs per 64-point tile instead of 36.  bbdm_winograd_pack_weight_f32(7, w4, packed, 4 Cout, Cin, InPad, 0),
 * bbdm_winograd_tiles(7, N, H, W) = N ceil((H + 1) / 7) ceil((W + 1) / 7) (tile t reads x rows 7 t - 1 .. 7 t + 6; phase pa writes rows
 * 7 t + a - pa, a = 0 .. 6), bbdm_winograd_input_bf3p_f32(7, ..., upsample = 0, ...), bbdm_winograd_gemm_bf3p_f32(7, ...) (64 GEMMs) and
 * bbdm_winograd_output_stats_f32(7, ..., flags = BBDM_CONV_OUT_PHASES, ...) with Cout % 128 == 0 -- no other entry point takes m = 7. */
int bbdm_upsample_phase_weights_f32(const float* w_oihw, float* w4, int Cout, int Cin, void* stream);
int bbdm_winograd_input_f32(int m, const float* x, int ldx, float* V, const float* pre_scale, const float* pre_bias,
                            int pre_ld, int pre_silu, int upsample, int N, int H, int W, int CinPad, void* stream);
int bbdm_winograd_gemm_f32(int m, const float* V, const float* packed_wino, float* M, int N, int H, int W, int CinPad,
                           int Cout, void* stream);
/* stage (3) with the output's GroupNorm statistics accumulated (see bbdm_conv2d_nhwc_stats_f32; cpg % 4 == 0) */
int bbdm_winograd_output_stats_f32(int m, const float* M, const float* bias, const float* residual, int ldr, float* out,
                                   int ldo, int flags, int N, int H, int W, int Cout, bbdm_stats_t* stats0, int cpg0, int coff0,
                                   bbdm_stats_t* stats1, int cpg1, int coff1, void* stream);
int bbdm_winograd_output_f32(int m, const float* M, const float* bias, const float* residual, int ldr, float* out,
                             int ldo, int flags, int N, int H, int W, int Cout, void* stream);

/* ---- convolution backward (training: autograd of the call sites above; the reference uses ATen's) -------- */
/* Data gradient = the forward kernel run on dY with transposed + spatially flipped weights: pack them with this
 * (dY carries CoutIn >= Cout channels, CoutIn % 4 == 0), then call bbdm_conv2d_nhwc_f32(dY, ..., CinPad = CoutIn,
 * Cout = Cin, ks).  The packed size is bbdm_conv_packed_dgrad_floats(). */
size_t bbdm_conv_packed_dgrad_floats(int Cout, int Cin, int CoutIn, int ks);
int bbdm_conv_pack_weight_dgrad_f32(const float* w_oihw, float* packed, int Cout, int Cin, int CoutIn, int ks,
                                    void* stream);
/* Weight gradient dW[co][ci][r][s] = sum_{n,h,w} dY[n,h,w,co] X[n,h+r-p,w+s-p,ci], written in OIHW (overwrite), and
 * (dbias != NULL) the bias gradient dbias[co] = sum_{n,h,w} dY[n,h,w,co] from the same pass over dY.
 * x: NHWC pitch ldx (Cin % 4 == 0); dy: NHWC pitch ldy; ws: ws_floats floats of scratch (split-K partials, reduced in a fixed
 * order: deterministic).  bbdm_conv_wgrad_workspace_floats() covers every path the launcher can take for these dimensions; the
 * launcher CHECKS ws_floats against the path it takes (ABI 21): the bf16-plane path of wide 1x1 layers falls back to the TN GEMM when
 * ws is too small for it, every other shortfall returns BBDM_E_BADARG without launching anything. */
size_t bbdm_conv_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int ks);
int bbdm_conv_wgrad_f32(const float* x, int ldx, const float* dy, int ldy, float* dw_oihw, float* dbias, float* ws,
                        size_t ws_floats, int N, int H, int W, int Cin, int Cout, int ks, void* stream);
/* The same weight gradient for ks = 3, stride 1 in the Winograd domain (csrc/winograd_wgrad.hip): dU_xi = V_xi^T dM_xi over
 * the tiles with V = B^T d B (bbdm_winograd_input_f32) and dM = A dY A^T, then dW = G^T dU G -- (m+2)^2 / (9 m^2) of the
 * direct FLOPs.  m = 2, 4 (H, W multiples of m) or 6 (any H, W); Cin % 4 == 0, Cout % 4 == 0.  Deterministic (K splits are
 * added in a fixed order).  ws: bbdm_winograd_wgrad_workspace_floats() floats, 16-byte aligned.
 *   bbdm_conv3x3_winograd_wgrad_f32 chains the stages below (+ bbdm_colsum_f32 for dbias != NULL):
 *   bbdm_winograd_dy_transform_f32 : dy NHWC (pitch ld) -> dM[(m+2)^2][bbdm_winograd_tiles()][Cout]
 *   bbdm_gemm_tn_batched_f32       : C[z][b][M][N] = sum_{k in K-range z} A[b][k][M] B[b][k][N] (fp32 MFMA; z <
 *                                    bbdm_gemm_tn_splits(batch, K, M, N) partial sums for the consumer to add in order)
 *   bbdm_winograd_wgrad_finish_f32 : dU[splits][(m+2)^2][Cin][Cout] -> dW OIHW.
 *   bbdm_winograd_wgrad_finish_bias_f32 : the same + dbias[c] = sum_{t < T} dm11[t][c] in the same launch (dm11: the fp32 plane
 *                                    (1, 1) of dM = the tile sums of dY, pitch Cout, from bbdm_winograd_dy_transform_bf3p_f32;
 *                                    Cout % 4 == 0) -- the bias gradient autograd derives for nn.Conv2d (openaimodel.py:233,244). */
size_t bbdm_winograd_wgrad_workspace_floats(int m, int N, int H, int W, int Cin, int Cout);
int bbdm_conv3x3_winograd_wgrad_f32(int m, const float* x, int ldx, const float* dy, int ldy, float* dw_oihw, float* dbias,
                                    float* ws, int N, int H, int W, int Cin, int Cout, void* stream);
int bbdm_winograd_dy_transform_f32(int m, const float* dy, int ld, float* dM, int N, int H, int W, int Cout, void* stream);
int bbdm_gemm_tn_splits(int batch, long long K, int M, int N);
int bbdm_gemm_tn_batched_f32(const float* A, int lda, size_t a_stride, const float* B, int ldb, size_t b_stride, float* C,
                             int batch, long long K, int M, int N, void* stream);
int bbdm_winograd_wgrad_finish_f32(int m, const float* dU, int splits, float* dw_oihw, int Cin, int Cout, void* stream);
int bbdm_winograd_wgrad_finish_bias_f32(int m, const float* dU, int splits, float* dw_oihw, int Cin, int Cout, const float* dm11,
                                        long long T, float* dbias, void* stream);
/* Column sums out[c] = sum_m dy[m][c] (bias gradients, per-channel reductions).  acc: 4 * C 8-byte words of scratch (exact
 * integer-limb cells, csrc/stats_acc.h: the sums do not depend on the order in which workgroups finish -- ABI 21). */
int bbdm_colsum_f32(const float* dy, int ld, double* acc, float* out, long long M, int C, void* stream);
/* Per-image column sums out[n*ldo + c] = sum_{m < M} dy[(n*M + m)*ld + c] (gradient of a per-image broadcast add).
 * acc: 4 * N * C 8-byte words of scratch (limb cells, as above). */
int bbdm_colsum_batched_f32(const float* dy, int ld, double* acc, float* out, int ldo, int N, long long M, int C,
                            void* stream);

/* ---- GroupNorm (util.py:199-216; sites openaimodel.py:205,229,306,688) -------------------------------- */
/* GroupNorm statistics live in an EXACT, order-independent accumulator (csrc/stats_acc.h): per (image, group) the sum and the
 * sum of squares as integer limbs, [N][G][2][4] 64-bit words = bbdm_groupnorm_stats_bytes(N, G) bytes, zeroed by the caller
 * (one memset per forward for all GroupNorms).  Producers add with integer atomics, so the value does not depend on the
 * order in which their workgroups finish (the reference pins its kernels with cudnn.deterministic, main.py:57-65);
 * consumers fold the limbs back into fp64.  `stats` arguments below are pointers to such accumulators (`bbdm_stats_t`). */
size_t bbdm_groupnorm_stats_bytes(int N, int G);
/* sums_out[N][G][2] (fp64: sum, sum of squares) = the accumulated values (tests, diagnostics). */
int bbdm_groupnorm_stats_read_f64(const bbdm_stats_t* stats, double* sums_out, int N, int G, void* stream);
/* Accumulate per-(n, group) sum and sum-of-squares of x into stats. */
int bbdm_groupnorm_stats_f32(const float* x, int ldx, bbdm_stats_t* stats, int N, int HW, int C, int G, void* stream);
/* scale_out / bias_out [N][ld]: GN(x)[n,:,:,c] [* (1 + film scale) + film shift] == x * scale_out[n][c] + bias_out[n][c],
 * from the statistics above -- the coefficients bbdm_conv2d_nhwc_f32 applies on the fly (pre_scale / pre_bias). */
int bbdm_groupnorm_coeffs_f32(const bbdm_stats_t* stats, const float* gamma, const float* beta, const float* film, int film_ld,
                              float* scale_out, float* bias_out, int ld, int N, int HW, int C, int G, float eps,
                              void* stream);
/* y = resample( act( GN(x) [* (1 + scale) + shift] ) )  --  the fused run
 *   GroupNorm32 -> [FiLM: openaimodel.py:270-273] -> SiLU -> [avg-pool 2x2 :159 | nearest x2 :118].
 * stats: as produced above (may be NULL with gamma == NULL: pure resample of x, the x_upd path :263).
 * film: [N][film_ld] with scale at [n][c] and shift at [n][C + c] (the emb_layers output :267-272), or NULL.
 * silu: 0/1.  resample: 0 none, 1 avg-pool 2x2 (H, W even), 2 nearest x2, 3 keep the even positions (turns a stride-1
 * conv output into the stride-2, pad-1 conv of Downsample(use_conv=True), openaimodel.py:153-156), 4 keep the odd positions
 * (the stride-2 conv on a (0,1,0,1)-padded input of the VQGAN's Downsample, model/VQGAN/model.py:68-72).  H, W are INPUT dims. */
int bbdm_groupnorm_apply_f32(const float* x, int ldx, const bbdm_stats_t* stats, const float* gamma, const float* beta,
                             const float* film, int film_ld, float* y, int ldy, int N, int H, int W, int C, int G,
                             float eps, int silu, int resample, void* stream);

/* ---- attention (openaimodel.py:359-375 legacy order, :398-413 new order) ------------------------------ */
/* qkv: NHWC [N, T, 3*heads*ch] pitch ldq.  Channel of (head h, part p in {q,k,v}, c):
 *   legacy (new_order = 0): h*3*ch + p*ch + c        new order: p*heads*ch + h*ch + c
 * out: NHWC [N, T, heads*ch] pitch ldo, channel h*ch + c.  softmax((q*s)^T (k*s)) v with s = ch^-1/4,
 * streamed over keys (no T x T tensor is ever materialised).  ch in {16, 32, 64}.
 * lse (may be NULL): fp32 [N][heads][T] log-sum-exp of every query's score row, kept for the backward pass. */
int bbdm_attention_f32(const float* qkv, int ldq, float* out, int ldo, float* lse, int N, int T, int heads, int ch,
                       int new_order, void* stream);
/* The same attention in two launches for long sequences (ABI 22): bbdm_attention_kv_planes_f32 writes the bf16x3 operand planes of
 * every (image, head)'s keys and values ONCE (K scaled; 6 bytes per element in the MFMA fragment order, csrc/attention.hip) and
 * bbdm_attention_planes_f32 copies them tile by tile into LDS (LDS-DMA) instead of splitting K / V again in each of the T / 128
 * workgroups that walk them.  bbdm_attention_kv_planes_bytes: size of `planes`, or 0 where the form does not apply (ch not in
 * {32, 64}, T % 128 != 0, T < 1024, option "attn_pipe" < 2) -- callers then use bbdm_attention_f32.  Bit-equal to it. */
size_t bbdm_attention_kv_planes_bytes(int N, int T, int heads, int ch);
int bbdm_attention_kv_planes_f32(const float* qkv, int ldq, void* planes, size_t planes_bytes, int N, int T, int heads, int ch,
                                 int new_order, void* stream);
int bbdm_attention_planes_f32(const float* qkv, int ldq, float* out, int ldo, float* lse, int N, int T, int heads, int ch, int new_order,
                              const void* planes, void* stream);        /* (bbdm_attention_f32's arguments, then the planes) */
/* ... on the fp16-pair planes (ABI 25; csrc/h2_split.h): Q, K, V under ONE power-of-two scale derived from `bound`, a device float >=
 * max |qkv| (both launches read the same one); the softmax weights P in (0, 1] under their exact bound 1.  Three f16 MFMA terms per
 * product instead of six, 4 bytes per K / V element instead of 6; measured against an fp64 attention at least as accurate as the bf16x3
 * form (tests/test_kernels_gpu.py::test_attention_h2).  Same shapes as the bf16x3 pair (bbdm_attention_kv_planes_h2_bytes is 0 otherwise).
 * The bound of a qkv PROJECTION y = W x + b needs no pass over y: max_row sum |W| x bound(x) + max |b| --
 *   bbdm_h2_rowl1_f32       : out2[0] = max over rows of sum_k |w[row][k]| (rounded up), out2[1] = max |bias| (0 without one); w [rows][cols]
 *   bbdm_h2_affine_bound_f32: *out_bound = *in_bound * gain2[0] + gain2[1]  (gain2 = bbdm_h2_rowl1_f32's output; one thread, every forward) */
size_t bbdm_attention_kv_planes_h2_bytes(int N, int T, int heads, int ch);
int bbdm_attention_kv_planes_h2_f32(const float* qkv, int ldq, void* planes, size_t planes_bytes, int N, int T, int heads, int ch,
                                    int new_order, const float* bound, void* stream);
int bbdm_attention_planes_h2_f32(const float* qkv, int ldq, float* out, int ldo, float* lse, int N, int T, int heads, int ch, int new_order,
                                 const void* planes, const float* bound, void* stream);
int bbdm_h2_rowl1_f32(const float* w, const float* bias, int rows, int cols, float* out2, void* stream);
int bbdm_h2_affine_bound_f32(const float* in_bound, const float* gain2, float* out_bound, void* stream);
/* Backward of the above (training; the reference re-runs the block under CheckpointFunction, util.py:119-148):
 * dqkv (same layout / pitch convention as qkv, pitch lddq) from dout [N,T,heads*ch] (pitch lddo), the forward's
 * qkv, out and lse.  Two streaming kernels (dQ per query block; dK,dV per key block), no T x T tensor. */
int bbdm_attention_bwd_f32(const float* qkv, int ldq, const float* out, int ldo, const float* dout, int lddo,
                           const float* lse, float* dwork /* N*heads*T floats */, float* dqkv, int lddq,
                           int N, int T, int heads, int ch, int new_order, void* stream);

/* ---- timestep embedding + small dense layers (util.py:151-171; openaimodel.py:511-516,735; :221-227,267) */
/* emb[n][:] = [cos(t_n f_0..f_{half-1}), sin(t_n f_0..)] (+ one zero column if dim is odd).  t: int64[N];
 * freqs: fp32[dim/2] = exp(-ln(1e4) i / half), computed by the HOST exactly as the reference does (util.py:160-163
 * builds it on the CPU and moves it to the device). */
int bbdm_timestep_embedding_f32(const int64_t* t, const float* freqs, float* emb, int N, int dim, void* stream);
/* y[n][o] = act_out( sum_i act_in(x[n][i]) * w[o][i] + b[o] ), w row-major [Out][In] (nn.Linear layout).
 * act_*: 0 none, 1 SiLU.  For the M = batch GEMMs of the embedding path (N <= 64 rows). */
int bbdm_linear_f32(const float* x, const float* w, const float* b, float* y, int N, int In, int Out,
                    int act_in, int act_out, void* stream);

/* bbdm_linear_f32 on weights packed ONCE (inference plans: static weights; the 51 MB FiLM projection of the UNets moves at 1 TB/s
 * when every lane streams its own 2 KB row).  packed = bbdm_linear_pack_f32(w [Out][In]): bbdm_linear_packed_bytes(Out, In) bytes,
 * 4 KB blocks [32 outputs x 32 k] in the order the B fragments are read from LDS, streamed sequentially per output tile
 * (csrc/embed.hip: linear_packed_kernel).  N <= 32 rows, In a multiple of 32 (bbdm_linear_packed_supported); results identical to
 * bbdm_linear_f32 bit for bit. */
size_t bbdm_linear_packed_bytes(int Out, int In);
int bbdm_linear_packed_supported(int N, int In, int Out);
int bbdm_linear_pack_f32(const float* w, void* packed, int Out, int In, void* stream);
int bbdm_linear_packed_f32(const float* x, const void* packed, const float* b, float* y, int N, int In, int Out, int act_in,
                           int act_out, void* stream);
/* Backward of bbdm_linear_f32 (training): dw[o][i] = sum_n dy[n][o] act_in(x[n][i]); db[o] = sum_n dy[n][o] (db may be
 * NULL); dx[n][i] = act_in'(x[n][i]) sum_o dy[n][o] w[o][i] (dx may be NULL).  x is the PRE-activation input.
 * ws: bbdm_linear_bwd_workspace_floats() floats.  N <= 64. */
size_t bbdm_linear_bwd_workspace_floats(int N, int In, int Out);
int bbdm_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                        int N, int In, int Out, int act_in, void* stream);

/* ---- Brownian-Bridge scheduler arithmetic (BBM.py) ---------------------------------------------------- */
/* objective ids: 0 'grad', 1 'noise', 2 'ysubx' (BBM.py:134-141,148-160). */
/* q_sample (BBM.py:128-146): x_t = (1-m)x0 + m y + sqrt(var) eps ; target per objective.  t: int64[N];
 * m_t / variance_t: the registered fp32 buffers [T].  per_sample = C*H*W. */
int bbdm_bb_q_sample_f32(const float* x0, const float* y, const float* noise, const int64_t* t,
                         const float* m_t, const float* variance_t, float* x_t, float* target,
                         int N, int per_sample, int objective, void* stream);
/* One reverse step after the UNet call (BBM.py:186-201, and the steps[i]==0 branch :174-180):
 * x0_recon = predict_x0(x_t, y, t, pred) [clamped to +-1 if clip]; if last: x_next = x0_recon, else the
 * posterior mean + sigma_t * noise.  t / t_next are the scalar table indices steps[i], steps[i+1].
 * x_next_alias (may be NULL): a second destination for x_next -- the sampling loop feeds x_next straight back as the next
 * step's x_t (BBM.py:218-220), so the step writes it into the UNet plan's input buffer as well and the next call skips its
 * input copy (one launch less outside the replayed graph per step). */
int bbdm_bb_p_sample_step_f32(const float* x_t, const float* y, const float* pred, const float* noise,
                              const float* m_t, const float* variance_t, int t, int t_next, int is_last,
                              float eta, int clip, int objective, float* x_next, float* x0_recon, float* x_next_alias,
                              int N, int per_sample, void* stream);
/* predict_x0_from_objective alone (BBM.py:148-160), per-sample t (used by p_losses :121). */
int bbdm_bb_predict_x0_f32(const float* x_t, const float* y, const float* pred, const int64_t* t,
                           const float* m_t, const float* variance_t, float* x0_recon,
                           int N, int per_sample, int objective, void* stream);
/* loss (BBM.py:114-117): loss_type 0 'l1' mean|a-b|, 1 'l2' mean (a-b)^2.  partial: 4 x 8 bytes zeroed by the caller (one exact
 * integer-limb cell, csrc/stats_acc.h: bitwise reproducible whatever the block order -- ABI 21);
 * out[0] = float(partial / count) is written by a tail kernel on the same stream. */
int bbdm_bb_loss_f32(const float* a, const float* b, double* partial, float* out, size_t count, int loss_type,
                     void* stream);

/* d loss / d pred of bbdm_bb_loss_f32 (BBM.py:114-117 under autograd), same layout as pred, scaled by the upstream
 * scalar gradient gscale[0] (device memory: no host sync).  pred is the second argument of the loss (a - b = target - pred
 * in the reference; the sign convention here is d/d pred). */
int bbdm_bb_loss_bwd_f32(const float* pred, const float* target, const float* gscale, float* dpred, size_t count,
                         int loss_type, void* stream);

/* ---- GroupNorm backward (training) -------------------------------------------------------------------- */
/* Backward of bbdm_groupnorm_apply_f32 (same x / stats / gamma / beta / film / silu / resample as the forward):
 *   da   : gradient of the forward's output y (at the OUTPUT resolution), pitch ldda
 *   dadd : optional second gradient at the output resolution, passed through the same resampling transpose and
 *          added (the skip path h_upd/x_upd share the resampler, openaimodel.py:262-263); with gamma == NULL only
 *          this term is produced (pure resample backward)
 *   dx   : result, pitch lddx; accumulate != 0 adds to what is there (a tensor feeding two consumers)
 *   dgamma / dbeta [C] are overwritten; dfilm (may be NULL) receives [N][dfilm_ld] d scale at [c], d shift at [C+c]
 *   ws   : bbdm_groupnorm_bwd_workspace_doubles() fp64 elements of scratch. */
size_t bbdm_groupnorm_bwd_workspace_doubles(int N, int C, int G);
int bbdm_groupnorm_bwd_f32(const float* x, int ldx, const bbdm_stats_t* stats, const float* gamma, const float* beta,
                           const float* film, int film_ld, const float* da, int ldda, const float* dadd, int ldadd,
                           float* dx, int lddx, int accumulate, float* dgamma, float* dbeta, float* dfilm,
                           int dfilm_ld, double* ws, int N, int H, int W, int C, int G, float eps, int silu,
                           int resample, void* stream);

/* ---- VQGAN first stage: the ops model/VQGAN/model.py + quantize.py need beyond the UNet's --------- */
/* Batched GEMM with ACTIVATION operands (AttnBlock's bmm(q, k) and bmm(v, w_), model.py:166-185, one head over all channels):
 * bbdm_gemm_pack_b_f32 packs, per batch element, a row-major matrix [R x K] (transposed = 0; out = A B^T) or [K x R]
 * (transposed != 0; out = A B) into the matrix-core kernel's operand layout (bbdm_gemm_packed_b_floats floats each);
 * bbdm_gemm_batched_f32 then computes out_b [rows x R] = A_b [rows x K] . B_b on v_mfma_f32_32x32x2_f32. */
size_t bbdm_gemm_packed_b_floats(int R, int K);
int bbdm_gemm_pack_b_f32(const float* b, int ldb, size_t b_stride, float* packed, int batch, int R, int K, int transposed,
                         void* stream);
int bbdm_gemm_batched_f32(const float* a, int lda, size_t a_stride, const float* packed_b, float* out, int ldo,
                          size_t out_stride, int batch, int rows, int K, int R, void* stream);
/* In-place softmax(scale * s) over each row of s [rows][T] (pitch ld): w_ = softmax(w_ * c^-0.5, dim=2), model.py:175-176. */
int bbdm_softmax_rows_f32(float* s, int ld, long long rows, int T, float scale, void* stream);
/* VectorQuantizer2.forward (quantize.py:271-286): per latent pixel z [e_dim] the index of the nearest codebook row under
 * d = sum(z^2) + sum(e^2) - 2 z.e (fp32, the reference's term order; first minimum wins) and/or that row itself.
 * z: [pixels][ldz] (NHWC latent), codebook: [n_e][e_dim] (embedding.weight), indices: int64 [pixels] or NULL,
 * zq: [pixels][ldq] or NULL.  e_dim <= 8. */
int bbdm_vq_nearest_f32(const float* z, int ldz, const float* codebook, long long* indices, float* zq, int ldq,
                        long long pixels, int n_e, int e_dim, void* stream);

/* ---- SpatialTransformer / cross-attention conditioning (attention.py = model/BrownianBridge/base/
 *      modules/attention.py) -------------------------------------------------------------------------------------------- */
/* CrossAttention.forward (attention.py:170-194): out[n, i, h*ch + d] = sum_j softmax_j(q_i . k_j * ch^-1/2) v_j per head.
 * q: [N][Tq][ldq], k / v: [N][Tk][ldkv] token-major ('b n (h d)': head h at channel h*ch), out: [N][Tq][ldo].
 * ch in {16, 32, 64}.  Streaming softmax on the f32 matrix core, same kernel as bbdm_attention_f32.
 * lse (may be NULL): [N][heads][Tq] log-sum-exp of the scaled scores, kept for the backward pass.
 * bbdm_cross_attention_bwd_f32 (autograd of the above; the reference re-runs the block under CheckpointFunction,
 * attention.py:212-213): dq [N][Tq] (pitch lddq), dk / dv [N][Tk] (pitch lddkv), overwritten; dwork: N*heads*Tq floats. */
int bbdm_cross_attention_f32(const float* q, int ldq, const float* k, const float* v, int ldkv, float* out, int ldo, float* lse,
                             int N, int Tq, int Tk, int heads, int ch, void* stream);
int bbdm_cross_attention_bwd_f32(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* out, int ldo,
                                 const float* dout, int lddo, const float* lse, float* dwork, float* dq, int lddq, float* dk,
                                 float* dv, int lddkv, int N, int Tq, int Tk, int heads, int ch, void* stream);
/* nn.LayerNorm(C) over the channel axis of `rows` tokens (attention.py:204-206): y = (x - mean) / sqrt(var + eps) * gamma
 * + beta, biased variance, fp32. */
int bbdm_layernorm_f32(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, long long rows,
                       int C, float eps, void* stream);
/* GEGLU (attention.py:38-46): y[r][c] = a[r][c] * gelu(a[r][inner + c]), exact (erf) GELU; a = proj(x) with 2*inner columns. */
int bbdm_geglu_f32(const float* a, int lda, float* y, int ldy, long long rows, int inner, void* stream);
/* Their gradients (training through SpatialTransformer blocks).  layernorm_bwd: dx = d/dx of the LayerNorm (+ dadd when given:
 * the residual branch of BasicTransformerBlock, attention.py:215-218), dgamma / dbeta overwritten; ws: 8*C 8-byte words ([2][C] limb cells, C <= 2048).
 * geglu_bwd: da[rows][2*inner] from the forward input a and dy[rows][inner]. */
int bbdm_layernorm_bwd_f32(const float* x, int ldx, const float* gamma, const float* dy, int lddy, const float* dadd, int ldadd,
                           float* dx, int lddx, float* dgamma, float* dbeta, double* ws, long long rows, int C, float eps,
                           void* stream);
int bbdm_geglu_bwd_f32(const float* a, int lda, const float* dy, int lddy, float* da, int ldda, long long rows, int inner,
                       void* stream);

/* ---- fp32-accurate batched GEMM on the BF16 matrix core (the Winograd tile GEMMs; csrc/gemm_bf3.hip) ----------- */
/* v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate on gfx950.  Each fp32 operand is split exactly into three
 * bf16 numbers (x = x1 + x2 + x3, round-to-nearest residuals) and the six leading bf16 x bf16 products -- everything
 * above 2^-23 of |a b| -- are accumulated in fp32 by v_mfma_f32_32x32x16_bf16: an fp32-accurate product at 2.67x the
 * matrix throughput of the f32 MFMA (error against an fp64 GEMM measured next to the f32-MFMA kernel in
 * tests/test_kernels_gpu.py::test_gemm_bf3_accuracy).
 *   bbdm_gemm_bf3_packed_halfs : number of 16-bit elements of the split weight buffer
 *   bbdm_gemm_bf3_pack_f32     : fp32 packed [batch][CinPad/16][CoutPad128][16] (what bbdm_winograd_pack_weight_f32 /
 *                                bbdm_conv_pack_weight_f32(ks = 1) produce) -> [batch][CinPad/16][3][CoutPad128][16] bf16
 *   bbdm_gemm_bf3_supported    : shape gate (T % 256 == 0, CinPad % 16 == 0, Cout % 4 == 0)
 *   bbdm_gemm_bf3_f32          : M[b] = V[b] ([T x CinPad] fp32, split while staged) . U[b]   for b < batch
 *   bbdm_winograd_gemm_bf3_f32 : stage (2) of the Winograd path (same arguments as bbdm_winograd_gemm_f32). */
size_t bbdm_gemm_bf3_packed_halfs(int batch, int CinPad, int Cout);
int bbdm_gemm_bf3_pack_f32(const float* packed_f32, void* packed_bf3, int batch, int CinPad, int Cout, void* stream);
int bbdm_gemm_bf3_supported(long long T, int CinPad, int Cout);
int bbdm_gemm_bf3_f32(const float* V, const void* packed_bf3, float* M, int batch, long long T, int CinPad, int Cout,
                      void* stream);
/* 1x1 convolution / nn.Linear on NHWC activations on the same kernel: out = x . W^T + bias (+ residual), operand pitches
 * ldx / ldo / ldr address channel slices; packed_bf3 = bbdm_gemm_bf3_pack_f32(batch 1) of bbdm_conv_pack_weight_f32(ks 1).
 * pixels % 256 == 0, CinPad % 16 == 0 (bbdm_gemm_bf3_supported).  residual may alias out. */
int bbdm_conv1x1_bf3_f32(const float* x, int ldx, const void* packed_bf3, const float* bias, const float* residual, int ldr,
                         float* out, int ldo, long long pixels, int CinPad, int Cout, void* stream);
int bbdm_winograd_gemm_bf3_f32(int m, const float* V, const void* packed_bf3, float* M, int N, int H, int W, int CinPad,
                               int Cout, void* stream);

/* ---- the same fp32-accurate GEMM with BOTH operands pre-split by their producers (csrc/gemm_bf3p.hip) ----------- */
/* Replaces the contraction of nn.Conv2d 3x3 on the wide layers (openaimodel.py:207,233,524) like bbdm_gemm_bf3_f32, arithmetic
 * identical to it bit for bit; the difference is WHERE the exact three-way bf16 split happens: the producer of the A operand
 * (the Winograd input transform, or bbdm_gemm_bf3p_split_rows_f32 for an ordinary fp32 matrix) writes the three planes, in the
 * order the matrix core consumes them, so the GEMM's main loop is LDS-DMA copies + MFMAs with no VALU arithmetic.
 * Plane layout: 1 KB "fragment units" of 32 rows x 16 k of one plane, element (r, k) at byte (k>>3)*512 + r*16 + (k&7)*2;
 *   A planes: [batch][T/32][CinPad/16][3][1 KB],  B planes: [batch][CoutPad128/32][CinPad/16][3][1 KB]  (6 B per element).
 *   bbdm_gemm_bf3p_a_bytes / _b_bytes : buffer sizes (T rounded up to 256 rows, Cout to 128 columns)
 *   bbdm_gemm_bf3p_supported          : shape gate (T % 256 == 0, CinPad % 16 == 0, Cout % 4 == 0)
 *   bbdm_gemm_bf3p_pack_b_f32         : fp32 packed [batch][CinPad/16][CoutPad128][16] -> B planes
 *   bbdm_gemm_bf3p_split_rows_f32     : fp32 rows [batch][T][ldx] -> A planes (rows up to the next multiple of 256 zeroed)
 *   bbdm_gemm_bf3p_f32                : M[b][T][ldo] = A_b . B_b (+ bias) (+ residual; may alias M)   for b < batch
 *   bbdm_winograd_input_bf3p_f32      : stage (1) of the Winograd path writing A planes (arguments of bbdm_winograd_input_f32)
 *   bbdm_winograd_gemm_bf3p_f32       : stage (2) on them (arguments of bbdm_winograd_gemm_f32) */
size_t bbdm_gemm_bf3p_a_bytes(int batch, long long T, int CinPad);
size_t bbdm_gemm_bf3p_b_bytes(int batch, int CinPad, int Cout);
int bbdm_gemm_bf3p_supported(long long T, int CinPad, int Cout);
int bbdm_gemm_bf3p_pack_b_f32(const float* packed_f32, void* b_planes, int batch, int CinPad, int Cout, void* stream);
int bbdm_gemm_bf3p_split_rows_f32(const float* x, int ldx, void* a_planes, int batch, long long T, int CinPad, void* stream);
int bbdm_gemm_bf3p_f32(const void* a_planes, const void* b_planes, const float* bias, const float* residual, int ldr, float* M,
                       int ldo, int batch, long long T, int CinPad, int Cout, void* stream);
/* bbdm_conv1x1_bf3_f32 (1x1 convolutions / Linears: openaimodel.py:244,307,315; attention.py:162-169) on the PIPELINED kernel
 * with an fp32 A operand (csrc/gemm_bf3p.hip: gemm_bf3q_pipe_kernel): x is read as it lies in HBM, every wave splits its 16 B of the
 * next-but-one chunk between its MFMAs; b_planes = bbdm_gemm_bf3p_pack_b_f32(batch = 1) of bbdm_conv_pack_weight_f32(ks = 1)'s
 * buffer.  Same arguments and results (bit for bit) as bbdm_conv1x1_bf3_f32; pixels need not be a multiple of 256. */
int bbdm_conv1x1_bf3q_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr,
                          float* out, int ldo, long long pixels, int CinPad, int Cout, void* stream);
/* The same product for SMALL problems -- the 1x1 convolutions / Linears of the latent and 64^2-pixel configurations (a few hundred
 * to a few thousand pixels: qkv / proj_out, skip projections), which are bound by the length of a workgroup's chain of K steps, not
 * by the matrix pipe (csrc/gemm_bf3p.hip: gemm_bf3s_kernel -- 64 x 64 tiles, 64 channels per step, x and the weight planes by
 * LDS-DMA two steps ahead, x split when a wave reads its fragment; one launch, no split-K workspace).  Same arguments, same b_planes,
 * same results bit for bit as bbdm_conv1x1_bf3q_f32; CinPad a multiple of 64, any pixel count. */
int bbdm_conv1x1_bf3s_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr,
                          float* out, int ldo, long long pixels, int CinPad, int Cout, void* stream);
int bbdm_winograd_input_bf3p_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias,
                                 int pre_ld, int pre_silu, int upsample, int N, int H, int W, int CinPad, void* stream);
/* bbdm_winograd_input_bf3p_f32 with GroupNorm [-> FiLM] [-> SiLU] folded in FROM THE STATISTICS (util.py:214-216,
 * openaimodel.py:258-278): the kernel forms x * sc[n][c] + bi[n][c] itself from `stats` (this GroupNorm's accumulator slot,
 * bbdm_groupnorm_stats_bytes layout), gamma / beta [C] and the FiLM vector (film[n][c] scale, film[n][C + c] shift; null: none) --
 * the expressions, hence the bits, of bbdm_groupnorm_coeffs_f32 followed by bbdm_winograd_input_bf3p_f32, in one launch.  For SMALL
 * problems (every thread repeats the fp64 fold of its channel pair).  Same argument order with (stats, unused) for (pre_scale,
 * pre_bias) and C for pre_ld; C == CinPad, C / G even, HW = pixels per image of the normalised tensor (x's own H * W, also with
 * upsample = 1). */
int bbdm_winograd_input_bf3p_gn_f32(int m, const float* x, int ldx, void* Vp, const void* stats, const void* unused, int C,
                                    int pre_silu, int upsample, int N, int H, int W, int CinPad, const float* gamma,
                                    const float* beta, const float* film, int film_ld, int HW, int G, float eps, void* stream);
int bbdm_winograd_gemm_bf3p_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W, int CinPad,
                                int Cout, void* stream);

/* ---- the SMALL 3x3 layers on the same GEMM (latent configurations: Template-LBBDM-f16.yaml:107-130 runs every conv of
 * openaimodel.py:207,233 on 16x16 / 8x8 / 4x4 maps; the 64^2-pixel model its inner levels) -------------------------------------------
 * A few hundred rows and K up to 2048: 256-row tiles would give a fraction of the 256 CUs a workgroup and stream the weights -- read
 * exactly once -- through a handful of prefetch queues.  The forward GEMM therefore (1) computes only the row tiles that hold real
 * rows (`rows` <= T, a multiple of 32), (2) drops to 128 x 128 tiles when larger ones leave CUs idle (inside bbdm_gemm_bf3p_f32 too),
 * and (3) splits K: split z < splits writes its partial sums to M[z][batch][T][ldo] (no bias / residual) and the consumer adds the
 * partials in order z = 0, 1, ... -- deterministic, no atomics.
 *   bbdm_gemm_bf3p_fwd_splits            : the measured split count for a problem (1 = the problem fills the chip unsplit)
 *   bbdm_gemm_bf3p_splitk_f32            : the GEMM (any 1 <= splits <= CinPad / 16 that leaves no split empty)
 *   bbdm_winograd_gemm_bf3p_splits       : split count of a Winograd layer's tile GEMMs (m = 2, 4; always 1 for m = 6)
 *   bbdm_winograd_gemm_bf3p_splitk_f32   : stage (2) writing M[splits][(m+2)^2][tiles][Cout]
 *   bbdm_winograd_output_splitk_stats_f32: stage (3) adding the partials while it loads them (arguments of
 *                                          bbdm_winograd_output_stats_f32, `splits` appended) */
int bbdm_gemm_bf3p_fwd_splits(int batch, long long rows, int CinPad, int Cout);
int bbdm_gemm_bf3p_splitk_f32(const void* a_planes, const void* b_planes, float* M, int ldo, int batch, long long T, long long rows,
                              int CinPad, int Cout, int splits, void* stream);
int bbdm_winograd_gemm_bf3p_splits(int m, int N, int H, int W, int CinPad, int Cout);
int bbdm_winograd_gemm_bf3p_splitk_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W, int CinPad,
                                       int Cout, int splits, void* stream);
int bbdm_winograd_output_splitk_stats_f32(int m, const float* M, const float* bias, const float* residual, int ldr, float* out,
                                          int ldo, int flags, int N, int H, int W, int Cout, bbdm_stats_t* stats0, int cpg0, int coff0,
                                          bbdm_stats_t* stats1, int cpg1, int coff1, int splits, void* stream);

/* ---- the same plane GEMM on TWO fp16 planes per operand ("h2"; round 6, ABI 24: csrc/h2_split.h, gemm_bf3p.hip, winograd.hip) ------- */
/* Replaces, for the inference forward of the Winograd layers whose input is GroupNorm -> [FiLM] -> SiLU of a tensor (every 3x3
 * convolution inside a ResBlock: openaimodel.py:205-207,229-233), the six-term bf16x3 product by x 2^e = h1 + h2 (fp16, round to
 * nearest) and the three terms h1 k1 + h1 k2 + h2 k1 on v_mfma_f32_32x32x16_f16: half the matrix instructions, 4 B per operand element
 * instead of 6, and HALF the roundings of the fp32 accumulator -- which, not the 2^-23 of the operands, is what bounds the accuracy of an
 * fp32 GEMM on the matrix core (against fp64 the h2 product is more accurate than bf16x3 AND than v_mfma_f32_32x32x2_f32:
 * tests/test_kernels_gpu.py::test_gemm_h2p_accuracy).  fp16 has 5 exponent bits, so every operand carries a BOUND -- a device float
 * >= max |x| over the whole operand -- from which producer and consumer derive the same power-of-two scale (h2_exp_of_bound): the scaled
 * bound lies in [2^14, 2^15), nothing can reach fp16's 65504.  A bound is measured (bbdm_absmax_f32: weights) or proved
 * (bbdm_h2_gn_bounds_f32: a z-score over n values is at most sqrt(n - 1)), never guessed.
 *   bbdm_gemm_h2p_a_bytes / _b_bytes   : plane buffer sizes (4 B per element; T to 256 rows, Cout to 128 columns)
 *   bbdm_absmax_f32                    : *bound = max(*bound, max |x[i]|)  (order-independent; the caller zeroes *bound)
 *   bbdm_gemm_h2p_pack_b_f32           : fp32 packed [batch][CinPad/16][CoutPad128][16] -> B planes scaled by `bound` x `gain` (gain: 1, or
 *                                        the factor between the bound's tensor and the packed one -- bbdm_winograd_g_gain(m))
 *   bbdm_gemm_h2p_split_rows_f32       : fp32 rows -> A planes scaled by `bound`
 *   bbdm_gemm_h2p_f32 / _splitk_f32    : the GEMMs of bbdm_gemm_bf3p_f32 / _splitk_f32 (same shapes, tiles and split rule)
 *   bbdm_h2_gn_bounds_f32              : bounds[l] for every GroupNorm-fed layer l of a plan in ONE launch; `table`: device array of
 *                                        {const float* gamma, *beta; int film_off, C; float zmax, gain} (32 B each)
 *   bbdm_winograd_input_gain           : max |B^T d B| / max |d| of tile m
 *   bbdm_winograd_input_h2p_f32 / _gn_f32, bbdm_winograd_gemm_h2p_f32 / _splitk_f32 : stages (1) and (2) of the Winograd path on
 *                                        these planes: the arguments of the bf3p forms, then vbound >= max |d| of the TRANSFORMED
 *                                        tensor (x after the fused producer; both stages apply the gain themselves) and, for the
 *                                        GEMMs, ubound >= the filter's largest tap (bbdm_absmax_f32 of the weights; the packers and
 *                                        the GEMMs multiply it by bbdm_winograd_g_gain(m)) */
size_t bbdm_gemm_h2p_a_bytes(int batch, long long T, int CinPad);
size_t bbdm_gemm_h2p_b_bytes(int batch, int CinPad, int Cout);
int bbdm_absmax_f32(const float* x, long long n, float* bound, void* stream);
/* ... of a [rows][C] view with pitch ldx (a channel slice of an NHWC buffer: the dY of a data-gradient convolution); C % 4 == 0 */
int bbdm_absmax_rows_f32(const float* x, int ldx, long long rows, int C, float* bound, void* stream);
int bbdm_gemm_h2p_pack_b_f32(const float* packed_f32, void* b_planes, const float* bound, float gain, int batch, int CinPad, int Cout,
                             void* stream);
/* G g G^T straight into the fp16-pair planes (bbdm_winograd_pack_weight_bf3p_f32's arguments + wbound: a device float >= the filter's
 * largest tap; planes scaled by wbound x bbdm_winograd_g_gain(m), the factor bbdm_winograd_gemm_h2p_f32 applies to its `ubound`) */
int bbdm_winograd_pack_weight_h2p_f32(int m, const float* w_oihw, void* b_planes, int Cout, int Cin, int InPad, int dgrad,
                                      const float* wbound, void* stream);
float bbdm_winograd_g_gain(int m);
int bbdm_gemm_h2p_split_rows_f32(const float* x, int ldx, void* a_planes, const float* bound, int batch, long long T, int CinPad,
                                 void* stream);
int bbdm_gemm_h2p_f32(const void* a_planes, const void* b_planes, const float* bound_a, const float* bound_b, const float* bias,
                      const float* residual, int ldr, float* M, int ldo, int batch, long long T, int CinPad, int Cout, void* stream);
int bbdm_gemm_h2p_splitk_f32(const void* a_planes, const void* b_planes, const float* bound_a, const float* bound_b, float* M, int ldo,
                             int batch, long long T, long long rows, int CinPad, int Cout, int splits, void* stream);
int bbdm_h2_gn_bounds_f32(const void* table, int nlayers, const float* film, int film_ld, int N, float* bounds, void* stream);
/* *bound = max over the N x G cells of `stats` of sqrt(sum of squares): >= max |x| of the tensor the statistics were taken of (its raw
 * values: the skip projections read the block input itself).  Launch it after the producers that fill `stats`. */
int bbdm_h2_stats_bound_f32(const bbdm_stats_t* stats, int N, int G, float* bound, void* stream);
/* The wide 1x1 convolutions / Linears of bbdm_conv1x1_bf3q_f32 on the fp16-pair planes (openaimodel.py:244,307; same arguments):
 * x is read as it lies in HBM and split into two fp16 halves under xbound's scale by the waves between their MFMAs;
 * b_planes = bbdm_gemm_h2p_pack_b_f32(batch = 1) of bbdm_conv_pack_weight_f32(ks = 1)'s buffer under wbound = its bbdm_absmax_f32. */
int bbdm_conv1x1_h2q_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr, float* out,
                         int ldo, long long pixels, int CinPad, int Cout, const float* xbound, const float* wbound, void* stream);
/* ... and of bbdm_conv1x1_bf3s_f32 (the small-problem kernel: CinPad a multiple of 64), same planes and bounds */
int bbdm_conv1x1_h2s_f32(const float* x, int ldx, const void* b_planes, const float* bias, const float* residual, int ldr, float* out,
                         int ldo, long long pixels, int CinPad, int Cout, const float* xbound, const float* wbound, void* stream);
float bbdm_winograd_input_gain(int m);
int bbdm_winograd_input_h2p_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias, int pre_ld,
                                int pre_silu, int upsample, int N, int H, int W, int CinPad, const float* vbound, void* stream);
/* training forward: Vp as above + the TRANSPOSED planes Vt of the weight gradient, which stay the exact bf16 split (layout and size of
 * bbdm_winograd_input_bf3p_tr_f32's: dM, their partner in bbdm_gemm_bf3p_tn_f32, has no bounded range) */
int bbdm_winograd_input_h2p_tr_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias, int pre_ld,
                                   int pre_silu, int upsample /* 0 */, int N, int H, int W, int CinPad, void* Vt, const float* vbound,
                                   void* stream);
/* ... and the Winograd-domain WEIGHT gradient on these planes (UNetModel.gemm_h2_train = 3): the transposed copy Vt as fp16 pairs
 * (bbdm_winograd_input_h2p_tr2_f32; bbdm_gemm_h2p_tn_at_bytes bytes), dM^T = (A dY A^T)^T under the measured maximum of dY x
 * bbdm_winograd_dy_gain(m) (bbdm_winograd_dy_transform_h2p_f32; bbdm_gemm_h2p_tn_bt_bytes bytes), and their TN product
 * (bbdm_gemm_h2p_tn_f32: the shapes, split rule and C layout of bbdm_gemm_bf3p_tn_f32) */
int bbdm_winograd_input_h2p_tr2_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias, int pre_ld,
                                    int pre_silu, int upsample /* 0 */, int N, int H, int W, int CinPad, void* Vt, const float* vbound,
                                    void* stream);
int bbdm_winograd_dy_transform_h2p_f32(int m, const float* dy, int ld, void* dMt, float* dm11, int N, int H, int W, int Cout,
                                       const float* dybound, void* stream);
float bbdm_winograd_dy_gain(int m);
size_t bbdm_gemm_h2p_tn_at_bytes(int batch, long long K, int M);
size_t bbdm_gemm_h2p_tn_bt_bytes(int batch, long long K, int N);
int bbdm_gemm_h2p_tn_f32(const void* at_planes, const void* bt_planes, float* C, int batch, long long K, int M, int N,
                         const float* bound_a, float gain_a, const float* bound_b, float gain_b, void* stream);
int bbdm_winograd_input_h2p_gn_f32(int m, const float* x, int ldx, void* Vp, const bbdm_stats_t* stats, const void* unused, int C,
                                   int pre_silu, int upsample, int N, int H, int W, int CinPad, const float* gamma, const float* beta,
                                   const float* film, int film_ld, int HW, int G, float eps, const float* vbound, void* stream);
int bbdm_winograd_gemm_h2p_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W, int CinPad, int Cout,
                               const float* vbound, const float* ubound, void* stream);
int bbdm_winograd_gemm_h2p_splitk_f32(int m, const void* Vp, const void* b_planes, float* M, int N, int H, int W, int CinPad, int Cout,
                                      int splits, const float* vbound, const float* ubound, void* stream);

/* ---- Winograd-domain weight gradient on the same bf16x3 GEMM (training; csrc/gemm_bf3p.hip, csrc/winograd.hip) --------- */
/* Replaces bbdm_gemm_tn_batched_f32 (f32 MFMA) in dW = G^T [ sum_tiles V_xi^T dM_xi ] G (autograd of nn.Conv2d 3x3 at
 * openaimodel.py:207,233; BaseRunner.py:412) where both operands come as bf16 planes written TRANSPOSED by their producers -- rows =
 * channels, contraction index = tiles -- so that the contraction over the tiles is the ordinary K loop of bbdm_gemm_bf3p_f32's kernel:
 *   bbdm_winograd_input_bf3p_tr_f32     : the input transform of the TRAINING forward: the planes Vp for the forward GEMM and their
 *                                         transposed copy Vt [xi][CinPad32 / 32][tiles / 16][3][1 KB] for the weight gradient
 *   bbdm_winograd_dy_transform_bf3p_f32 : dY -> dMt [xi][CoutPad128 / 32][tiles / 16][3][1 KB] (A dY A^T, transposed planes) and
 *                                         dm11 [tiles][Cout] fp32 = its plane (1, 1) = the tile sums of dY (m = 8: the sums themselves),
 *                                         whose column sums are the bias gradient
 *   bbdm_gemm_bf3p_tn_at_bytes / _bt_bytes / _supported / _splits : buffer sizes, shape gate (K % 256, M % 32, N % 4), K splits
 *   bbdm_gemm_bf3p_tn_f32               : C[z][b][M][N] = sum over the K range of split z of At_b^T-as-stored . Bt_b, z < splits
 *                                         (the consumer adds the splits in order: bbdm_winograd_wgrad_finish_f32) */
int bbdm_winograd_input_bf3p_tr_f32(int m, const float* x, int ldx, void* Vp, const float* pre_scale, const float* pre_bias,
                                    int pre_ld, int pre_silu, int upsample /* 0 */, int N, int H, int W, int CinPad, void* Vt,
                                    void* stream);
int bbdm_winograd_dy_transform_bf3p_f32(int m, const float* dy, int ld, void* dMt, float* dm11, int N, int H, int W, int Cout,
                                        void* stream);
size_t bbdm_gemm_bf3p_tn_at_bytes(int batch, long long K, int M);
size_t bbdm_gemm_bf3p_tn_bt_bytes(int batch, long long K, int N);
int bbdm_gemm_bf3p_tn_supported(long long K, int M, int N);
int bbdm_gemm_bf3p_tn_splits(int batch, long long K, int M, int N);
int bbdm_gemm_bf3p_tn_f32(const void* at_planes, const void* bt_planes, float* C, int batch, long long K, int M, int N, void* stream);

/* ---- optimizer + EMA in one pass (training) ------------------------------------------------ */
/* Replaces torch.optim.Adam.step() (runners/utils.py:48-51; called at runners/BaseRunner.py:413-415) and
 * EMA.update() (runners/base/EMA.py:21-29; called at BaseRunner.py:173-178,422-423) for all parameters with ONE launch.
 * `table` is DEVICE memory: one entry per <= bbdm_opt_chunk_elems() consecutive elements of one parameter tensor; the
 * pointers address that chunk inside the parameter, its .grad, its Adam moments and its EMA shadow (grad / shadow may be
 * NULL: that chunk skips the Adam / the EMA part).  do_adam: apply the Adam update with `step` = the 1-based count of
 * optimizer steps (bias corrections 1 - beta^step); ema_mode: 0 = no EMA, 1 = shadow = (1 - decay) * p + decay * shadow
 * (the UPDATED p when do_adam), 2 = shadow = p (EMA.update(with_decay=False), BaseRunner.py:174).  Arithmetic: the
 * single-tensor formulas of torch.optim.Adam (lerp / mul+addcmul / sqrt / div / addcdiv), fp32, no FMA contraction. */
typedef struct BbdmOptChunk {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    float* shadow;
    int n;
    int pad_;
} BbdmOptChunk;
int bbdm_opt_chunk_elems(void);
int bbdm_adam_ema_step_f32(const BbdmOptChunk* table, int nchunks, int do_adam, double lr, double beta1, double beta2,
                           double eps, double weight_decay, long long step, int ema_mode, double ema_decay,
                           void* stream);

/* ---- sample egress ------------------------------------------------------------------------ */
/* fp32 NCHW [N,C,H,W] -> uint8 NHWC [N,H,W,C] with the arithmetic of save_single_image (runners/utils.py:67-74):
 * to_normal != 0: v = clamp(v * 0.5 + 0.5, 0, 1); then u8 = (uint8) clamp(v * 255 + 0.5, 0, 255) (truncation) -- each
 * operation rounded separately in fp32, so the bytes equal the reference's.  One launch for the whole batch. */
int bbdm_images_to_u8_f32(const float* x_nchw, unsigned char* out_nhwc, int N, int C, int H, int W, int to_normal,
                          void* stream);

/* ---- options (no reference counterpart): the library's few integer switches, for tests and tools/ A-B runs ----------------- */
/* Nothing in the library reads the environment, and no launcher latches a setting: an option is read at every call, so one
 * process can run both sides of an A/B.  Process-wide and unsynchronised.  Names (default):
 *   "wgrad1x1_bf3" (1)  1x1 weight gradients: 1 = the bf16x3 planes path from ~100 FLOP per split byte, 0 = always the f32-MFMA
 *                       TN GEMM, 2 = every shape the plane layout accepts (tests);
 *   "wino_idx64"   (0)  1 = force the 64-bit row-address variant of the Winograd input transform (tests);
 *   "bf3p_kernel"  (6)  tile shape of the pre-split bf16x3 GEMM: 6 = the library's choice, 4 = 256 x 256, 5 = 256 x 128,
 *                       7 = 128 x 128 workgroup tiles (the parity tests reach every instantiation on small problems);
 *   "attn_bf3"     (1)  attention forward: 1 = Q K^T and P V on the bf16x3 path, 2 = only Q K^T, 0 = both on the f32 MFMA;
 *   "attn_pipe"    (2)  ... its main loop with the operand splits dealt between the MFMAs (1, 2) or in phases of their own (0); 2 also
 *                       enables the pre-split entry points (bbdm_attention_kv_planes_bytes > 0) from T = 1024 (3: for every T % 128 == 0, tests);
 *                       same bits in every setting;
 *   "bf3p_pad_rows" (1) pre-split GEMM: the idle 32-row blocks of a ragged last row tile read the zero rows the producer of the A planes
 *                       wrote behind the real ones (1) or the last real rows again (0); the stored result is the same.
 * Unknown names return BBDM_E_BADARG. */
int bbdm_set_option(const char* name, int value);
int bbdm_get_option(const char* name, int* value);

#ifdef __cplusplus
}
#endif
#endif /* BBDM_HIP_H */
