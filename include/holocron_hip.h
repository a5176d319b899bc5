/* holocron_hip.h — C ABI of libholocron_hip.so (MI355X / gfx950).
 *
 * The reference (frgfm/Holocron) is pure Python on top of torch ops; it has no FFI of its
 * own.  These entry points are what a binding for its hot path would call:
 * each one replaces the aten / torchvision kernels that the cited reference lines launch.
 * All pointers are DEVICE pointers unless said otherwise; no torch types cross this
 * boundary.  Every function enqueues work on `stream` and returns immediately:
 * 0 = ok, 1 = bad argument, 2 = launch failure.  Buffers are owned by the caller.
 *
 * Tensor layout: activations are NHWC bf16 ("channels-last" in HBM), weights are packed
 * bf16 [rows][taps][k] by hc_pack_conv_weight, parameters/gradients/optimizer state are fp32
 * in the reference's own layouts (OIHW for conv weights).
 */
#ifndef HOLOCRON_HIP_H
#define HOLOCRON_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* hc_stream_t; /* hipStream_t */

#define HC_MAX_TAPS 12
#define HC_STAT_REPLICAS 128
/* Run-to-run determinism.  Per-channel statistics (BatchNorm sums, their backward sums, depthwise weight gradients) are
 * accumulated with fp32 atomics into `hc_get_stat_replicas()` replicas ([replicas][k][C]; workgroup b -> replica b % replicas)
 * that a finalize kernel adds in a fixed order.  With the default 128 replicas several workgroups share a replica and the order of
 * their atomics moves the last bit of a sum; hc_set_deterministic(1) raises the count to HC_STAT_REPLICAS_DETERMINISTIC, above the
 * grid of every statistics-producing launch (checked: launches with more workgroups return HC_ERR_ARG), so that every workgroup
 * owns its slot and two runs of the same step are bit-identical.  It also makes the split reduction of hc_conv_wgrad and the
 * global average pool single-writer.  Callers size every statistics buffer with hc_get_stat_replicas() (zero-filled as before). */
#define HC_STAT_REPLICAS_DETERMINISTIC 32768
int hc_set_deterministic(int on);
int hc_get_deterministic(void);
int hc_get_stat_replicas(void);

/* One "parity class" of a gather-conv: output sub-grid (i,j) -> output pixel
 * (i*ostep+oy0, j*ostep+ox0); tap t reads source pixel (i*istep+dy[t], j*istep+dx[t]) of
 * source tensor src[t] and multiplies with weight tap wt[t]. */
typedef struct {
    int32_t OHg, OWg, oy0, ox0, ostep, istep, ntaps;
    int32_t tap[HC_MAX_TAPS]; /* packed: (dy & 0xff) | (dx & 0xff) << 8 | src << 16 | wt << 24 */
} hc_conv_class;
#define HC_TAP(dy, dx, src, wt) \
    ((int32_t)(((uint32_t)(dy) & 0xffu) | (((uint32_t)(dx) & 0xffu) << 8) | ((uint32_t)(src) << 16) | ((uint32_t)(wt) << 24)))

/* Implicit-GEMM convolution on MFMA: forward conv (1 class) and data-gradient
 * (1 class for stride 1, 4 parity classes for stride 2; two sources fuse the 3x3 and 1x1
 * branches of a RepBlock).  Replaces aten::convolution / convolution_backward(input) as
 * launched by nn.Conv2d in holocron/models/utils.py:73 (conv_sequence) and
 * holocron/models/classification/repvgg.py:71-73 (RepBlock.forward). */
typedef struct {
    const void* src0;   /* NHWC bf16 [N][IH][IW][srcC] */
    const void* src1;   /* second source (same geometry) or NULL */
    const void* wpk;    /* packed bf16 weights [Cout][T][srcC] */
    void* dst;          /* NHWC bf16 [N][OH][OW][Cout] */
    const void* resid;  /* optional NHWC bf16 added to the result (same shape as dst) */
    float* stats;       /* optional fp32 [HC_STAT_REPLICAS][2][Cout]: += sum(y), += sum(y*y); the
                           replicas spread atomic contention, hc_rep_bn_finalize sums them */
    const float* bias;  /* optional fp32 [Cout] */
    int32_t act;        /* 0 none, 1 relu, 2 hard_mish, 3 leaky(0.1), 4 mish, 5 silu */
    int32_t N, IH, IW, srcC;
    int32_t OH, OW, Cout;
    int32_t T;          /* taps stored in wpk */
    int32_t nclass;
    hc_conv_class cls[4];
    /* optional patch normalisation of NormConv2d (holocron/nn/functional.py:345-349), applied to the fp32 accumulator
     * before bias: v = pix_scale[pixel] * (v - pix_shift[pixel] * ch_coef[channel]); all three NULL otherwise */
    const float* pix_scale;  /* fp32 [N*OH*OW]: rsqrt(var + eps) of each unfolded patch */
    const float* pix_shift;  /* fp32 [N*OH*OW]: mean of each unfolded patch */
    const float* ch_coef;    /* fp32 [Cout]: sum over (ci, kh, kw) of the (bf16-rounded) weights */
    /* fp8 inference (BASELINE config C5, reparametrised RepVGG: repvgg.py:75-107 + one conv + ReLU per block): when
     * ch_mult != NULL the sources, packed weights and dst hold OCP e4m3 bytes (srcC % 64 == 0, Cout % 4 == 0), the MFMA is
     * v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales and dst = fp8(act(acc * ch_mult[co] + bias[co])), i.e.
     * ch_mult = weight_scale[co] * input_scale / output_scale and bias is pre-divided by the output scale. */
    const float* ch_mult;    /* fp32 [Cout] */
    /* two convolutions of the same source in one launch (stacked weight rows): when co_split > 0, output channels
     * [co_split, Cout) go to dst2 (NHWC with Cout - co_split channels) and their statistics to stats2, channels [0, co_split) to
     * dst / stats with co_split channels per pixel.  co_split % 4 == 0; no resid / bias / act / normalisation with it.  Used for the
     * 3x3 + 1x1 branches of the small-channel stride-2 RepBlocks and the stem (the 1x1 kernel sits at the centre tap of a second
     * set of rows): the source is gathered once instead of twice. */
    void* dst2;
    float* stats2;
    int32_t co_split;
    /* INFERENCE epilogue of a conv_sequence unit (holocron/models/utils.py:73-84 in eval mode: conv -> BatchNorm2d with running
     * statistics -> activation): when ch_scale != NULL the fp32 accumulator becomes v * ch_scale[co] + bias[co] (bias = the
     * BatchNorm shift; required with ch_scale) BEFORE the activation - the whole unit is one launch and the conv output is never
     * stored unnormalised.  act_slope: negative slope of act == 3 (0 = the 0.1 of the other callers).  resid_after_act != 0 adds
     * `resid` AFTER the activation (DarkNet's ResBlock: x + act(bn(conv)), darknetv3.py:59-61) instead of before it.  bf16 path
     * only, not with co_split / pix_scale / ch_mult. */
    const float* ch_scale;   /* fp32 [Cout] */
    float act_slope;
    int32_t resid_after_act;
} hc_conv_desc;
int hc_conv_gather(const hc_conv_desc* d, hc_stream_t stream);
/* Streaming form for 1 x 1 stride-1 convolutions over at most 128 input channels (the expansion convolutions of rexnet.py:97-103 and
 * the data gradients of the projections, conv_sequence(..., kernel_size=1) in holocron/models/utils.py:73): weights stationary in
 * registers, 32-pixel tiles, optional BatchNorm statistics; no bias / residual / activation / second source.  hc_conv_gather routes the
 * launches this covers here by itself (HC_CONV_PW=0: never, =2: also when the output is narrower than twice the input);
 * hc_conv_pointwise returns HC_ERR_ARG for a descriptor hc_conv_pointwise_supported rejects. */
int hc_conv_pointwise_supported(const hc_conv_desc* d);
int hc_conv_pointwise(const hc_conv_desc* d, hc_stream_t stream);

/* Stride-1 3x3 (+1x1) convolution of a RepBlock (holocron/models/classification/repvgg.py:71-73 and its data gradient) for
 * small channel counts: C <= 48 on large images (persistent software-pipelined workgroups, weights resident in registers, DMA'd
 * row window with halo, XCD-grouped tiles) and 64 <= C <= 256 on maps that fit one image into LDS (image-resident kernel).
 * Environment (experiments / tests): HC_CONV_SMALL_PIPE=0 non-pipelined kernel, HC_CONV_SMALL_GRID=n grid cap.
 * mode 0: out3 = W3 (*) srcA, out1 = W1 . srcA (+ optional BN statistics, replicated like hc_conv_desc);
 * mode 1: out3 = W3 (*) srcA + W1 . srcB + resid  (RepBlock data gradient; srcA = dy3, srcB = dy1).
 * w3/w1 are packed bf16 rows [out channel][tap][C] with the given row strides (elements).
 * mode | HC_CONV_SMALL_ROWS_IMAGE: w3 is the row-unit image of BOTH kernels (hc_pack_conv_weight modes 3 / 4; w1 and the strides
 * are ignored) - the format of the row-unit kernels for 192 channels @ 14x14, 96 @ 28x28 (conv_rows.hip) and 48 @ 112x112 / 56x56
 * (conv_rows48.hip); supported()
 * says whether a shape has it. */
#define HC_CONV_SMALL_ROWS_IMAGE 4
typedef struct {
    const void* srcA;
    const void* srcB;
    const void* w3;
    const void* w1;
    void* out3;
    void* out1;
    const void* resid;
    float* stats3;
    float* stats1;
    int32_t w3_rstride, w1_rstride;
    int32_t N, H, W, C, Cout, mode;
} hc_conv_small_desc;
int hc_conv_small(const hc_conv_small_desc* d, hc_stream_t stream);
int hc_conv_small_supported(const hc_conv_small_desc* d);

/* Stride-2 RepBlock forward of the HBM-bound front layers (csrc/conv_s2.hip): y3 = conv3x3(x, stride 2, pad 1),
 * y1 = conv1x1(x, stride 2) and the per-channel sum / sum of squares of both fp32 results - the two nn.Conv2d of a stride-2
 * RepBlock (holocron/models/classification/repvgg.py:57-60) and the batch statistics of the nn.BatchNorm2d behind each
 * (models/utils.py:76) - in ONE launch that reads the input once.  Replaces aten::convolution x 2 (+ the explicit im2col of the
 * 3-channel stem).  x: NHWC bf16 [N][H][W][Cin], or with x_nchw_f32 the image batch itself, NCHW fp32 [N][3][H][W];
 * w3img / w1img: the fragment images of hc_pack_conv_weights_multi modes 5 (6 for the stem); y3 / y1: NHWC bf16
 * [N][H/2][W/2][Cout]; stats3 / stats1: [replicas][2][Cout] accumulators (+=) or both NULL.
 * hc_conv_s2_supported: 1 for the shapes the kernel is built for (3 -> 48 @ 224, 48 -> 48 @ 112, 48 -> 96 @ 56), else 0. */
typedef struct {
    const void* x;
    const void* w3img;
    const void* w1img;
    void* y3;
    void* y1;
    float* stats3;
    float* stats1;
    int32_t N, H, W, Cin, Cout, x_nchw_f32;
} hc_conv_s2_desc;
int hc_conv_s2_supported(const hc_conv_s2_desc* d);
int hc_conv_s2_fwd(const hc_conv_s2_desc* d, hc_stream_t stream);
/* ... and its data gradient dx = conv3x3^T(dy3) + conv1x1^T(dy1) (aten::convolution_backward(input) of both convs, summed by
 * autograd in the reference): dy3 / dy1 NHWC bf16 [N][H/2][W/2][Cout], dx NHWC bf16 [N][H][W][Cin], wimg = the fragment image of
 * hc_pack_conv_weights_multi mode 7 (both kernels, taps ordered by output parity).  Shapes: 48 <- 48 @ 112, 48 <- 96 @ 56. */
typedef struct {
    const void* dy3;
    const void* dy1;
    const void* wimg;
    void* dx;
    int32_t N, H, W, Cin, Cout;
} hc_conv_s2_dgrad_desc;
int hc_conv_s2_dgrad_supported(const hc_conv_s2_dgrad_desc* d);
int hc_conv_s2_dgrad(const hc_conv_s2_dgrad_desc* d, hc_stream_t stream);
/* ... and the weight gradients of the stem block (aten::convolution_backward(weight) of its 3x3 and 1x1 conv, repvgg.py:57-60 with
 * in_channels = 3) straight from the image batch: x NCHW fp32 [N][3][224][224], dy3 / dy1 NHWC bf16 [N][112][112][48],
 * dw3 fp32 [48][3][3][3], dw1 fp32 [48][3][1][1] (= or += with `accumulate`), ws: hc_conv_s2_stem_wgrad_ws_bytes() of scratch
 * (per-wave partial slabs, added in a fixed order: bit-reproducible).  Returns an argument error for any other geometry. */
int64_t hc_conv_s2_stem_wgrad_ws_bytes(void);
int hc_conv_s2_stem_wgrad(const float* x, const void* dy3, const void* dy1, float* dw3, float* dw1, void* ws, int32_t N, int32_t H,
                          int32_t W, int32_t accumulate, hc_stream_t stream);

/* The stem block (in_channels = 3) FUSED with its BatchNorm passes: the 3x3 and the 1x1 conv are recomputed from the image batch in
 * every pass instead of being stored (each output is twice the input's size), so y3 / y1 / dy3 / dy1 never exist in HBM.  Together with
 * hc_rep_bn_finalize these three launches replace, for the first RepBlock of RepVGG (repvgg.py:57-60,71-73: two nn.Conv2d, two
 * training-mode nn.BatchNorm2d, the python sum and the ReLU), aten::convolution x 2, aten::native_batch_norm x 2, add, relu and their
 * backward ops (aten::threshold_backward, native_batch_norm_backward x 2, convolution_backward(weight) x 2).
 * x: NCHW fp32 [N][3][224][224]; w3img / w1img: hc_pack_conv_weights_multi mode 6 images; out / g: NHWC bf16 [N][112][112][48].
 *   hc_stem_stats: stats3 / stats1 [replicas][2][48] += per-channel sum / sum of squares of the fp32 conv results
 *   hc_stem_apply: out = act(coef[0] c3 + coef[1] c1 + coef[3])            (coef [4][48] of hc_rep_bn_finalize; act 1 = ReLU);
 *                  out_stats (or NULL) [replicas][2][48] += sum / sum of squares of the bf16-rounded `out` (the statistics of the
 *                  next block's identity BatchNorm, like hc_rep_apply)
 *   hc_stem_bwd:   the whole backward from ONE pass over (x, g): dgamma / dbeta of both BatchNorm layers and dw3 [48][3][3][3],
 *                  dw1 [48][3][1][1] (= or += with `accumulate`), through G = dz^T X and the Gram matrix X^T X of the 27-wide conv
 *                  windows (csrc/conv_s2.hip has the algebra); save [6][48] = hc_rep_bn_finalize's means / inverse deviations, w3 / w1
 *                  the fp32 master weights, `frozen` = eval-mode BatchNorm (dy = a dz), ws: hc_stem_bwd_ws_bytes() of scratch
 *                  (per-workgroup slabs, added in a fixed order: bit-reproducible). */
typedef struct {
    const float* x;
    const void* w3img;
    const void* w1img;
    int32_t N, H, W;
} hc_stem_desc;
typedef struct {
    const float* coef;
    const void* g;
    const float* save;
    const float* gamma3;
    const float* gamma1;
    const float* w3;
    const float* w1;
    float* dgamma3;
    float* dbeta3;
    float* dgamma1;
    float* dbeta1;
    float* dw3;
    float* dw1;
    void* ws;
    int32_t act, frozen, accumulate;
} hc_stem_bwd_desc;
int hc_stem_fused_supported(const hc_stem_desc* d);
int hc_stem_stats(const hc_stem_desc* d, float* stats3, float* stats1, hc_stream_t stream);
int hc_stem_apply(const hc_stem_desc* d, const float* coef, int32_t act, void* out, float* out_stats, hc_stream_t stream);
int64_t hc_stem_bwd_ws_bytes(void);
int hc_stem_bwd(const hc_stem_desc* d, const hc_stem_bwd_desc* b, hc_stream_t stream);

/* Weight gradient dW[co][ci][kh][kw] = sum_m dy[m][co] * x[m + tap][ci].
 * Replaces aten::convolution_backward(weight).  Split-K over output pixels: partial fp32
 * slabs go to `ws` (at least hc_conv_wgrad_ws_bytes bytes), then a reduce kernel writes
 * (beta=0) or accumulates (beta=1) the OIHW fp32 gradient. */
typedef struct {
    const void* x;      /* NHWC bf16 [N][IH][IW][Cin] */
    const void* dy;     /* NHWC bf16 [N][OH][OW][Cout] */
    float* dw;          /* fp32 [Cout][Cin][KH][KW] */
    void* ws;
    int32_t N, IH, IW, Cin, OH, OW, Cout;
    int32_t KH, KW, stride, pad;
    int32_t beta;
    int32_t co_valid, ci_valid; /* channel-padded operands (ReXNet's 27, 38, 50 ... carried as multiples of 16): when > 0, dw is the
                                 * UNPADDED gradient [co_valid][ci_valid][KH][KW] of the parameter itself - the padding rows / columns
                                 * of the slabs are dropped by the reduce kernel instead of by a slicing copy afterwards; 0 = Cout / Cin */
} hc_wgrad_desc;
int64_t hc_conv_wgrad_ws_bytes(const hc_wgrad_desc* d);
int hc_conv_wgrad(const hc_wgrad_desc* d, hc_stream_t stream);

/* The weight gradients of up to HC_WGRAD_MAX_JOBS SAME-SHAPED conv layers in one launch pair (aten::convolution_backward(weight) of
 * each; the conv_sequence units of the DarkNet / CSP / YOLO stacks, holocron/models/utils.py:73 under darknetv4.py:38-115 and
 * yolov4.py:31-229, repeat a handful of shapes 4-9 times per step).  One layer of those stacks has 3-6 output tiles, so on its own it
 * needs 40-85 split-K slabs to fill 256 CUs and its fp32 partial sums outweigh its operands several times; a group fills the chip
 * with tiles x jobs and the split factor drops by the group size.  x[j] NHWC bf16 [N][IH][IW][Cin], dy[j] NHWC bf16 [N][OH][OW][Cout],
 * dw[j] fp32 OIHW (= when beta == 0, += when 1); ws: hc_conv_wgrad_group_ws_bytes().  Slabs are reduced in a fixed order.
 * hc_conv_wgrad_group_supported: 0 for shapes outside the DMA kernel's plan (Cin, Cout multiples of 64 from 128 on, 3x3 or 1x1) - call
 * hc_conv_wgrad per layer then. */
#define HC_WGRAD_MAX_JOBS 16
typedef struct {
    const void* x[HC_WGRAD_MAX_JOBS];
    const void* dy[HC_WGRAD_MAX_JOBS];
    float* dw[HC_WGRAD_MAX_JOBS];
    void* ws;
    int32_t njobs, N, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, beta;
} hc_wgrad_group_desc;
int hc_conv_wgrad_group_supported(const hc_wgrad_group_desc* d);
int64_t hc_conv_wgrad_group_ws_bytes(const hc_wgrad_group_desc* d);
int hc_conv_wgrad_group(const hc_wgrad_group_desc* d, hc_stream_t stream);

/* Both weight gradients of a RepBlock (repvgg.py:57-60: a 3x3 / pad 1 and a 1x1 / pad 0 conv on the same input with the same
 * stride; aten::convolution_backward(weight) of both) in ONE launch from ONE staging of x, for up to HC_WREP_MAX_JOBS blocks of
 * the same shape at a time (csrc/conv_wgrad_rep.hip).  x[j] NHWC bf16 [N][IH][IW][Cin], dy3[j] / dy1[j] NHWC bf16
 * [N][OH][OW][Cout], dw3[j] fp32 OIHW [Cout][Cin][3][3], dw1[j] fp32 [Cout][Cin][1][1] (= or += when `accumulate`); ws: scratch of
 * hc_rep_wgrad_ws_bytes().  The split-K partial sums are reduced in a fixed order: results are bit-reproducible.
 * hc_rep_wgrad_supported: shapes outside the kernel's plan (channel counts, LDS) -> 0, use hc_conv_wgrad twice instead.
 * hc_rep_wgrad_plan: diagnostic, out8 = {MR, NR, rows per step, steps in flight, nsplit, grid, LDS bytes, ring slots}. */
#define HC_WREP_MAX_JOBS 16
typedef struct {
    const void* x[HC_WREP_MAX_JOBS];
    const void* dy3[HC_WREP_MAX_JOBS];
    const void* dy1[HC_WREP_MAX_JOBS];
    float* dw3[HC_WREP_MAX_JOBS];
    float* dw1[HC_WREP_MAX_JOBS];
    void* ws;
    int32_t njobs, N, IH, IW, Cin, OH, OW, Cout, stride, accumulate;
} hc_rep_wgrad_desc;
int hc_rep_wgrad_supported(const hc_rep_wgrad_desc* d);
int64_t hc_rep_wgrad_ws_bytes(const hc_rep_wgrad_desc* d);
int hc_rep_wgrad_plan(const hc_rep_wgrad_desc* d, int32_t* out8);
int hc_rep_wgrad(const hc_rep_wgrad_desc* d, hc_stream_t stream);

/* fp32 OIHW master weights -> packed bf16.  mode 0: forward  [Cout][KH*KW][Cin];
 * mode 1: data-gradient [Cin][KH*KW (spatially flipped)][Cout]; mode 2: im2col order (see below).  `tap0`/`T` let several
 * kernels (3x3 + 1x1) share one packed tensor: taps are written at [tap0, tap0+KH*KW).
 * modes 3 / 4: the row-unit image read by hc_conv_small with HC_CONV_SMALL_ROWS_IMAGE (forward / data gradient; Cout == Cin == C,
 * C % 48 == 0): bf16 [10 * ceil(C/32) k32-steps][C rows][32], zero where C = 48 pads a tap to two steps; 3x3 taps at tap0 = 0, the 1x1 at tap0 = 9; rows and
 * k values are permuted so that an MFMA A fragment of a wave is 1 KB of consecutive bytes (layout: rows_image_index, rep_bn.hip). */
int hc_pack_conv_weight(const float* w, void* wpk, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                        int32_t mode, int32_t tap0, int32_t T, hc_stream_t stream);

/* Same for many tensors in ONE launch (`items` is a DEVICE array).  mode 2: im2col order
 * wpk[co][tap0 + (kh*KW+kw)*Cin + ci] with row length T (= padded K); untouched entries keep
 * their previous contents (zero them once). */
typedef struct {
    const float* w;
    void* dst;
    int32_t Cout, Cin, KH, KW, mode, tap0, T;
    int32_t ld; /* elements per packed row when the destination is channel padded (mode 0: >= Cin, mode 1: >= Cout); 0 = dense.
                 * modes 5 / 6 (fragment images of hc_conv_s2_fwd, bf16 [Cout/16][T steps][64][8], zero-initialised by the caller):
                 * ld = first k32 step of this image, tap0 = first 16-byte piece (mode 5) / first tap (mode 6) - csrc/rep_bn.hip
                 * s2_image_index.  mode 7: the data-gradient image of hc_conv_s2_dgrad, bf16 [Cin/16][T = 10 Cout/32 steps][64][8];
                 * the 3x3 kernel with tap0 = 0, the 1x1 with tap0 = 1 (its position in the parity-ordered tap list) */
} hc_pack_item;
int hc_pack_conv_weights_multi(const hc_pack_item* items, int32_t nitems, int64_t max_elems, hc_stream_t stream);

/* NCHW fp32 -> NHWC bf16 with channel padding to Cpad (zero filled) and back. */
int hc_nchw_to_nhwc_bf16(const float* x, void* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                         hc_stream_t stream);
int hc_nhwc_bf16_to_nchw(const void* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                         hc_stream_t stream);

/* ---- RepBlock training-mode BatchNorm fusion (repvgg.py:71-73 + torch BatchNorm2d) ---- */
/* From the conv-epilogue statistics compute the per-channel affine of every branch and update
 * running_mean/var (momentum, unbiased var) and num_batches_tracked like nn.BatchNorm2d.
 * coef out: fp32 [4][C] = a3, a1, a0, b ; save out: fp32 [6][C] = mean/invstd per branch. */
typedef struct {
    const float* stats[3];     /* [HC_STAT_REPLICAS][2][C] sums per branch (3x3, 1x1, identity) */
    const float* gamma[3];
    const float* beta[3];
    float* running_mean[3];
    float* running_var[3];
    int64_t* num_batches_tracked[3];
    float* coef;
    float* save;
    int32_t C;
    int64_t count;             /* N*H*W */
    float eps, momentum;
    int32_t training;          /* 0: use running stats (eval) */
    int32_t c_valid;           /* > 0: channels >= c_valid are layout padding: a = shift = 0, parameter / running
                                  statistic arrays hold only c_valid entries */
} hc_rep_bn_desc;
int hc_rep_bn_finalize(const hc_rep_bn_desc* d, hc_stream_t stream);

/* out = act(a3*y3 + a1*y1 + a0*x + b); optionally accumulates sum/sumsq of `out` into
 * out_stats [HC_STAT_REPLICAS][2][C] (the next block's identity-BN statistics). */
int hc_rep_apply(const void* y3, const void* y1, const void* x, const float* coef, void* out, float* out_stats,
                 int64_t npix, int32_t C, int32_t act, hc_stream_t stream);

/* stats[HC_STAT_REPLICAS][2][C] += per-channel sum / sum of squares of an NHWC bf16 tensor. */
int hc_channel_stats(const void* x, float* stats, int64_t npix, int32_t C, hc_stream_t stream);

/* Backward of the fused BN+sum+ReLU.  Pass 1: per-channel sums of dz = g*(out>0), dz*y3,
 * dz*y1, dz*x into red[HC_STAT_REPLICAS][4][C] (replicas spread the atomics).  Pass 2 (after hc_rep_bn_bwd_finalize): dy3, dy1, dx_id. */
int hc_rep_bwd_reduce(const void* g, const void* out, const void* y3, const void* y1, const void* x, float* red,
                      int64_t npix, int32_t C, hc_stream_t stream);
/* The same two passes without reading `out`: the mask is (z > 0) of the pre-activation z = a3*y3 + a1*y1 + a0*x + b recomputed from
 * the forward's `coef` (hc_rep_bn_finalize) with the very fma chain hc_rep_apply uses - bit-identical results, one tensor less per
 * pass.  act: 1 = ReLU, 0 = no activation (dz = g). */
int hc_rep_bwd_reduce_z(const void* g, const float* coef, int32_t act, const void* y3, const void* y1, const void* x, float* red,
                        int64_t npix, int32_t C, hc_stream_t stream);
int hc_rep_bwd_apply_z(const void* g, const float* coef, int32_t act, const void* y3, const void* y1, const void* x,
                       const float* bcoef, void* dy3, void* dy1, void* dxid, int64_t npix, int32_t C, hc_stream_t stream);
typedef struct {
    const float* red;          /* [HC_STAT_REPLICAS][4][C] */
    const float* save;         /* [6][C] from forward */
    const float* gamma[3];
    float* dgamma[3];
    float* dbeta[3];
    float* bcoef;              /* out fp32 [9][C]: A,B,Cc per branch */
    int32_t C;
    int64_t count;
    int32_t has_identity;
    int32_t accumulate;        /* 1: dgamma/dbeta += */
    int32_t c_valid;           /* > 0: channels >= c_valid are layout padding (A = B = C = 0, nothing written) */
    int32_t frozen;            /* 1: the forward normalised with the RUNNING statistics (eval mode / freeze_bn, trainer/utils.py:14-30): they
                                * do not depend on the batch, so dy = a * dz (B = C = 0); dgamma / dbeta as in training mode */
} hc_rep_bn_bwd_desc;
int hc_rep_bn_bwd_finalize(const hc_rep_bn_bwd_desc* d, hc_stream_t stream);
int hc_rep_bwd_apply(const void* g, const void* out, const void* y3, const void* y1, const void* x,
                     const float* bcoef, void* dy3, void* dy1, void* dxid, int64_t npix, int32_t C,
                     hc_stream_t stream);

/* ---- generic conv -> BatchNorm(train) -> activation [+ residual]  (conv_sequence, models/utils.py:61-84;
 * DarkNet ResBlock, darknetv3.py:23-70) ----
 * y = conv(x) comes from hc_conv_gather with the statistics epilogue; hc_rep_bn_finalize with only
 * branch 0 set gives coef (a = coef[0][C], shift = coef[3][C]).  act codes as in hc_conv_desc; `slope`
 * is the LeakyReLU negative slope.  Backward recomputes z = a*y + shift:
 *   reduce: red[replica][0][C] += sum g*act'(z), red[replica][1][C] += sum g*act'(z)*y  (layout of hc_rep_bwd_reduce)
 *   hc_rep_bn_bwd_finalize (has_identity = 0) -> bcoef rows 0..2 = A, B, C;  dy = A*g*act'(z) + B*y + C.
 * DropBlock placed after the activation (conv -> BN -> act -> DropBlock, models/utils.py:75-84) rides in the same
 * pass when keep/count (outputs of hc_dropblock_mask) are given: out = act(z)*keep*scale [+ res]; pass NULL for none.
 * `out_ld` / `g_ld`: channels per pixel of the buffer the output / incoming gradient lives in (>= C, % 8 == 0; the
 * pointer already includes the channel offset) so that a channel concat (darknetv4.py:115, yolov4.py:137) is
 * written in place by its producers and its gradient is read in place.
 * `res` (optional) has res_C <= C channels per pixel and is added to the first res_C channels (ReXBlock shortcut,
 * rexnet.py:140-141; res_C == C for the DarkNet residual). */
int hc_bn_act_apply(const void* y, const float* coef, const void* res, int32_t res_C, const float* keep, const float* count,
                    void* out, int32_t out_ld, int64_t npix, int32_t C, int32_t act, float slope, hc_stream_t stream);
int hc_bn_act_bwd_reduce(const void* g, int32_t g_ld, const void* y, const float* coef, const float* keep,
                         const float* count, float* red, int64_t npix, int32_t C, int32_t act, float slope,
                         hc_stream_t stream);
int hc_bn_act_bwd_apply(const void* g, int32_t g_ld, const void* y, const float* coef, const float* bcoef,
                        const float* keep, const float* count, void* dy, int64_t npix, int32_t C, int32_t act,
                        float slope, hc_stream_t stream);
/* ... with a SECOND DropBlock behind the residual add riding in the same passes: DarkNet's ResBlock is
 * dropblock(x + conv_sequence(x)) (holocron/models/classification/darknetv3.py:44-61), i.e. out = (act(z)*keep*scale + res)*keep2*scale2.
 * Backward: the incoming gradient is multiplied by keep2*scale2 first; `gres` (dense [npix][C] bf16, optional) receives that product -
 * the gradient of the residual input - so that no separate DropBlock pass (one read + one write of the block output, forward and
 * backward) is left.  keep2 / count2 NULL: the plain forms above. */
int hc_bn_act_apply_post(const void* y, const float* coef, const void* res, int32_t res_C, const float* keep, const float* count,
                         const float* keep2, const float* count2, void* out, int32_t out_ld, int64_t npix, int32_t C, int32_t act,
                         float slope, hc_stream_t stream);
int hc_bn_act_bwd_reduce_post(const void* g, int32_t g_ld, const void* y, const float* coef, const float* keep, const float* count,
                              const float* keep2, const float* count2, float* red, int64_t npix, int32_t C, int32_t act, float slope,
                              hc_stream_t stream);
int hc_bn_act_bwd_apply_post(const void* g, int32_t g_ld, const void* y, const float* coef, const float* bcoef, const float* keep,
                             const float* count, const float* keep2, const float* count2, void* gres, void* dy, int64_t npix,
                             int32_t C, int32_t act, float slope, hc_stream_t stream);

/* ---- NHWC bf16 data movement of the CSP / PAN / SPP stacks (darknetv4.py:112-115, yolov4.py:134-139,
 * nn/modules/downsample.py:154-167).  `*_ld` = channels per pixel of the buffer, `*_c0` = first channel; all % 8. ----
 * copy: dst[p][dst_c0 + c] = src[p][src_c0 + c], c < C  (chunk / cat and their gradients).
 * upsample2x: nearest-neighbour x2 (nn.Upsample(scale_factor=2), yolov4.py:64); bwd sums the four gradients.
 * spp: out [N][H][W][4C] = [x | maxpool5 | maxpool9 | maxpool13] (stride 1, pad k/2); idx uint8 [3][N][H][W][C]
 *      holds each window's argmax ((dy+6)*13 + dx+6; ties -> first in row-major order like torch's max_pool2d);
 *      bwd gathers the gradients whose argmax is the pixel. */
int hc_nhwc_copy(const void* src, int32_t src_ld, int32_t src_c0, void* dst, int32_t dst_ld, int32_t dst_c0, int64_t npix,
                 int32_t C, hc_stream_t stream);
int hc_upsample2x_fwd(const void* src, int32_t src_ld, int32_t src_c0, void* dst, int32_t dst_ld, int32_t dst_c0, int32_t N,
                      int32_t H, int32_t W, int32_t C, hc_stream_t stream);
int hc_upsample2x_bwd(const void* g, int32_t g_ld, int32_t g_c0, void* dx, int32_t dx_ld, int32_t dx_c0, int32_t N,
                      int32_t H, int32_t W, int32_t C, hc_stream_t stream);
int hc_spp_fwd(const void* x, void* out, void* idx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream);
int hc_spp_bwd(const void* g, const void* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream);

/* Global average pool over H*W (holocron/nn/modules/downsample.py:70-74) on NHWC bf16. */
int hc_gap_fwd(const void* x, float* y, int32_t N, int32_t HW, int32_t C, hc_stream_t stream);
int hc_gap_bwd(const float* dy, void* dx, int32_t N, int32_t HW, int32_t C, hc_stream_t stream);

/* Stem / tiny-Cin convs go through an explicit im2col: x NCHW fp32 -> col NHWC bf16
 * [N][OH][OW][Kpad] with k = (kh*KW+kw)*Cin+ci (zero padded to Kpad, Kpad % 16 == 0); the conv is
 * then a 1x1 hc_conv_gather on `col` with weights packed in mode 2, and its weight gradient a 1x1
 * hc_conv_wgrad unpacked by hc_unpack_im2col_grad. */
int hc_im2col_small(const float* x, void* col, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t OH, int32_t OW,
                    int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t Kpad, hc_stream_t stream);
int hc_unpack_im2col_grad(const float* dwcol, float* dw, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t Kpad,
                          int32_t beta, hc_stream_t stream);

/* ---- optimizers (multi-tensor; holocron/optim/adabelief.py:121-167, lars.py:90-135) ---- */
/* One entry per CHUNK of a parameter tensor (the host splits tensors into chunks of at most
 * HC_MT_CHUNK elements so that one workgroup handles one entry). */
#define HC_MT_CHUNK 8192
typedef struct {
    float* p;
    float* g;          /* LARS adds weight decay into g in place (lars.py:113-114) */
    float* m;          /* exp_avg / momentum_buffer (may be NULL for LARS with momentum 0) */
    float* s;          /* exp_avg_sq (AdaBelief) */
    float* smax;       /* max_exp_avg_sq (amsgrad) or NULL */
    int32_t n;         /* elements in this chunk */
    int32_t group;     /* index into the group array */
    int32_t tensor;    /* index of the owning tensor (LARS norms) */
    int32_t flags;     /* bit0: LARS momentum buffer is initialised from the gradient */
} hc_mt_chunk;
typedef struct {
    double lr, beta1, beta2, eps, weight_decay;
    int32_t step;      /* step count used by the update (>=1 when the update runs) */
    int32_t amsgrad;
} hc_adabelief_group;
/* `chunks` and `groups` are DEVICE arrays (the caller keeps them alive).  advance != 0 first
 * increments every group's `step` ON THE DEVICE (so a step captured in a hipGraph keeps counting
 * across replays), then applies the update. */
int hc_adabelief_step(const hc_mt_chunk* chunks, int32_t nchunks, hc_adabelief_group* groups, int32_t ngroups,
                      int32_t advance, hc_stream_t stream);

typedef struct {
    double lr, momentum, dampening, weight_decay;
    int32_t nesterov;
    int32_t pad_;
} hc_lars_group;
/* norms: fp32 [ntensors][2] scratch (sum of squares of p and g), zeroed by the callee. */
int hc_lars_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_lars_group* groups, float* norms,
                 int32_t ntensors, hc_stream_t stream);

/* AdamP / AdEMAMix (holocron/optim/adamp.py:142-200, holocron/optim/ademamix.py:158-200; the optimizers
 * references/classification/train.py:33,206-213 imports next to AdaBelief).  `step` is the count after the increment.
 * AdEMAMix: chunk.m = exp_avg, chunk.s = exp_avg_sq, chunk.smax = exp_avg_slow.
 * AdamP: two passes (moments + per-tensor sums, then the update with the tangent-space projection decided on the device
 * instead of the reference's host-side `if cosine_similarity(...) < delta / sqrt(numel)`); sums fp32 [ntensors][4] scratch,
 * numel int32 [ntensors]. */
typedef struct {
    double lr, beta1, beta2, beta3, alpha, eps, weight_decay, delta;
    int32_t step, amsgrad;
} hc_adamx_group;
int hc_ademamix_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_adamx_group* groups, hc_stream_t stream);
int hc_adamp_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_adamx_group* groups, float* sums, const int32_t* numel,
                  int32_t ntensors, hc_stream_t stream);

/* LAMB / RaLars (holocron/optim/lamb.py:84-137, holocron/optim/ralars.py:66-140): Adam-style moments + a per-tensor trust
 * ratio.  mode 0: LAMB, u = m / (sqrt(v) + eps); 1: rectified adaptive momentum, u = rect * (m / bc1) / (sqrt(v / bc2) + eps);
 * 2: the same without rectification (force_adaptive_momentum); 3: u = m / bc1.  u += weight_decay * p;
 * local_lr = 1 if clamp(|p|, clip_lo, clip_hi) == 0 or |u| == 0 else clamp(|p|) / |u|;  p -= lr * local_lr * u.
 * norms: fp32 [ntensors][2] scratch (zeroed by the callee); local_lr: fp32 [ntensors] out. */
typedef struct {
    double lr, beta1, beta2, eps, weight_decay, clip_lo, clip_hi, rect;
    int32_t step, mode;
} hc_lamb_group;
int hc_lamb_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_lamb_group* groups, float* norms, float* local_lr,
                 int32_t ntensors, hc_stream_t stream);
/* TAdam (holocron/optim/tadam.py:157-212): per tensor w_t = (dof + numel) / (sum((g - m)^2 / (v + eps)) + dof),
 * m = m W/(W + w_t) + w_t g/(W + w_t), W <- W (2 beta1 - 1)/beta1 + w_t (W_t: device array of pointers to the per-parameter
 * one-element state tensors), then the Adam step.  scratch: fp32 [3 * ntensors]; dof fp32 [ntensors] (numel when the
 * group's dof is None); numel / tensor_group int32 [ntensors]. */
int hc_tadam_step(const hc_mt_chunk* chunks, int32_t nchunks, const hc_adamx_group* groups, float* scratch, const float* dof,
                  const int32_t* numel, const int32_t* tensor_group, float* const* W_t, int32_t ntensors, hc_stream_t stream);
/* Adan (holocron/optim/adan.py:146-199): chunk.m = exp_avg, chunk.s = exp_avg_sq, chunk.smax = exp_avg_delta;
 * extra[i].m = max_exp_avg_delta (amsgrad) and extra[i].s = prev_grad (read only, like the reference) of the same chunk. */
int hc_adan_step(const hc_mt_chunk* chunks, const hc_mt_chunk* extra, int32_t nchunks, const hc_adamx_group* groups,
                 hc_stream_t stream);
/* Lookahead / Scout synchronisation (holocron/optim/wrapper.py:121-134): chunk.p = fast, chunk.m = slow weights;
 * slow += sync_rate * (fast - slow) when sync_rate > 0, then fast = slow. */
int hc_lookahead_sync(const hc_mt_chunk* chunks, int32_t nchunks, float sync_rate, hc_stream_t stream);

/* ---- gradient-bucket pack / unpack (no reference counterpart: the reference is single-device, holocron/trainer/core.py:90-104;
 * this is the cast-copy on either side of the data-parallel all-reduce of SURVEY.md §8e, holocron_amd/parallel.py) ----
 * dst[i][0..n[i]) = scale * src[i][0..n[i]) for nitems <= HC_MULTI_COPY_MAX pieces in ONE launch; element types fp32 or bf16
 * (src_bf16 / dst_bf16).  The table is passed by value to the kernel, so the descriptor may live on the caller's stack and a launch
 * captured in a hipGraph holds no reference to host memory.  The caller splits large tensors into pieces of a few hundred
 * thousand elements (HC_MULTI_COPY_PIECE) so that every piece is swept by the same 16 workgroups. */
#define HC_MULTI_COPY_MAX 64
#define HC_MULTI_COPY_PIECE 262144
typedef struct {
    const void* src[HC_MULTI_COPY_MAX];
    void* dst[HC_MULTI_COPY_MAX];
    int64_t n[HC_MULTI_COPY_MAX];
    int32_t nitems;
    int32_t src_bf16, dst_bf16;   /* 0: fp32, 1: bf16 */
    float scale;
} hc_multi_copy_desc;
int hc_multi_copy(const hc_multi_copy_desc* d, hc_stream_t stream);

/* ---- pointwise / losses / boxes ---- */
/* hard_mish: 0.5*x*clamp(x+2,0,2) (holocron/nn/functional.py:30-41), fp32, y may alias x. */
int hc_hard_mish_fwd(const float* x, float* y, int64_t n, hc_stream_t stream);
int hc_hard_mish_bwd(const float* x, const float* dy, float* dx, int64_t n, hc_stream_t stream);

/* pairwise box ops (holocron/ops/boxes.py:16-211), boxes fp32 xyxy; out [M][N].
 * kind: 0 iou, 1 giou, 2 diou_loss, 3 ciou_loss (== diou_loss), 4 iou_penalty,
 * 5 aspect_ratio_consistency */
int hc_box_pairwise(const float* b1, const float* b2, float* out, int32_t M, int32_t N, int32_t kind,
                    hc_stream_t stream);
/* Gradient of the same (the reference's functions are plain differentiable torch expressions, boxes.py:16-211; ciou_loss is the
 * training loss at yolov4.py:403): db1 [M][4] += sum_j g[i][j] d out[i][j] / d b1[i], db2 [N][4] likewise; the caller zeroes db1 / db2.
 * Ties of max / min split the gradient evenly, clamp(min=0) passes it at 0 - torch autograd's conventions. */
int hc_box_pairwise_bwd(const float* b1, const float* b2, const float* g, float* db1, float* db2, int32_t M, int32_t N, int32_t kind,
                        hc_stream_t stream);

/* Greedy NMS with torchvision.ops.nms semantics (yolov4.py:329): boxes [n][4] fp32 already
 * sorted by descending score (stable); ws scratch of hc_nms_ws_bytes(n); keep out: int32 [n]
 * indices into the sorted order (ascending), nkeep out: int32[1]. */
int64_t hc_nms_ws_bytes(int32_t n);
int hc_nms_sorted(const float* boxes, int32_t n, float iou_thr, void* ws, int32_t* keep, int32_t* nkeep,
                  hc_stream_t stream);
/* The same greedy NMS for `nprob` INDEPENDENT problems in one launch pair - the (image, scale) problems of a detector's eval batch
 * (holocron/models/detection/yolov4.py:302-336 runs torchvision.ops.nms once per image inside each of the three YoloLayers).  Problem
 * p owns the score-sorted boxes [off[p], off[p + 1]) of `boxes` (off: DEVICE int32 [nprob + 1]), its suppression bitmap lives at
 * ws + ws_off[p] 64-bit words (DEVICE int64 [nprob]; n_p * ceil(n_p / 64) words each), its kept indices - local to the problem, in
 * score order - are written to keep + off[p] and counted in nkeep[p].  nmax = the largest n_p (host value: grid and LDS size).
 * Decisions are those of hc_nms_sorted box for box. */
int hc_nms_sorted_batched(const float* boxes, const int32_t* off, int32_t nprob, int32_t nmax, float iou_thr, void* ws,
                          const int64_t* ws_off, int32_t* keep, int32_t* nkeep, hc_stream_t stream);

/* focal loss forward/backward (holocron/nn/functional.py:59-113); x [N][K][S] fp32 (S = product
 * of trailing dims), target int64 [N][S]; loss_el out [N*S] (unreduced), valid out uint8 [N*S]
 * (0 where target == ignore_index and 0 <= ignore_index < K). */
int hc_focal_loss_fwd(const float* x, const int64_t* target, const float* weight, float* loss_el, uint8_t* valid,
                      int32_t N, int32_t K, int64_t S, int32_t ignore_index, float gamma, hc_stream_t stream);
int hc_focal_loss_bwd(const float* x, const int64_t* target, const float* weight, const float* dloss_el, float* dx,
                      int32_t N, int32_t K, int64_t S, float gamma, hc_stream_t stream);

/* softmax cross-entropy with label smoothing on [N][K] fp32 logits (criterion of
 * references/classification/train.py:194); per-sample loss and dlogits of the MEAN loss. */
int hc_ce_fwd_bwd(const float* logits, const int64_t* target, float* loss_el, float* dlogits, int32_t N, int32_t K,
                  float label_smoothing, hc_stream_t stream);

/* The same criterion as the training loop calls it - nn.CrossEntropyLoss(label_smoothing=ls), mean over the rows whose target is
 * not ignore_index (references/classification/train.py:194; torch composes it from ~25 aten launches): forward writes the scalar
 * loss and aux[0] = number of valid rows (fixed-order sums: one single-workgroup launch for K <= 64, a wave-per-row launch plus a
 * single-workgroup sum for wider heads); backward writes dlogits = dloss[0] * d(mean loss)/dlogits with dloss a DEVICE scalar (the
 * upstream gradient autograd hands over).  aux holds hc_ce_mean_aux_floats(N) = 1 + 3 N floats (the rows' log-sum-exp and partial
 * sums behind aux[0]) and goes unchanged from forward to backward. */
int64_t hc_ce_mean_aux_floats(int32_t N);
int hc_ce_mean_fwd(const float* logits, const int64_t* target, float* loss, float* aux, int32_t N, int32_t K, float label_smoothing,
                   int64_t ignore_index, hc_stream_t stream);
int hc_ce_mean_bwd(const float* logits, const int64_t* target, const float* dloss, const float* aux, float* dlogits, int32_t N,
                   int32_t K, float label_smoothing, int64_t ignore_index, hc_stream_t stream);

/* Poly-1 loss (holocron/nn/functional.py:540-613).  Hard labels: target int64 [N][S], loss_el [N*S] =
 * w[t]*(-logp_t + eps*(1 - p_t)), valid as for focal.  Soft labels: target fp32 [N][K][S], loss_pos [N*S] =
 * sum over classes k != ignore_index (when 0 <= ignore_index < K) of w[k]*(-l_k + eps*(1 - exp l_k)),
 * l_k = log_softmax(x)_k * target_k (functional.py:578-610). */
int hc_poly_loss_hard_fwd(const float* x, const int64_t* target, const float* weight, float* loss_el, uint8_t* valid,
                          int32_t N, int32_t K, int64_t S, int32_t ignore_index, float eps, hc_stream_t stream);
int hc_poly_loss_hard_bwd(const float* x, const int64_t* target, const float* weight, const float* dloss_el, float* dx,
                          int32_t N, int32_t K, int64_t S, float eps, hc_stream_t stream);
int hc_poly_loss_soft_fwd(const float* x, const float* target, const float* weight, float* loss_pos, int32_t N,
                          int32_t K, int64_t S, int32_t ignore_index, float eps, hc_stream_t stream);
int hc_poly_loss_soft_bwd(const float* x, const float* target, const float* weight, const float* dloss_pos, float* dx,
                          int32_t N, int32_t K, int64_t S, int32_t ignore_index, float eps, hc_stream_t stream);

/* Dice loss reductions (holocron/nn/functional.py:523-524): sums fp32 [3][K] = per class sum over (n, s) of
 * x*t, x, t for x, target fp32 [N][K][S]; the K-sized rational expression stays on the host side.
 * Backward: dx = dsums[0][k]*t + dsums[1][k]. */
int hc_dice_sums(const float* x, const float* target, float* sums, int32_t N, int32_t K, int64_t S, hc_stream_t stream);
int hc_dice_bwd(const float* target, const float* dsums, float* dx, int32_t N, int32_t K, int64_t S, hc_stream_t stream);

/* DropBlock (holocron/nn/functional.py:465-500).  noise fp32 [N][H][W] uniform samples; keep out fp32
 * [N][H][W] = 1 - maxpool_{bs x bs, stride 1, pad bs/2}(noise <= gamma); count out fp32[1] = sum(keep)
 * (kept on the device: the reference's `if one_count > 0` host sync becomes a device-side select).
 * apply: y = (x * keep) * (count > 0 ? N*HW / count : 1); dtype 0 fp32 / 1 bf16, nhwc 0: [N][C][HW], 1: [N][HW][C];
 * y may alias x (in-place).  block_size must be odd (the reference's shapes only broadcast for odd sizes). */
int hc_dropblock_mask(const float* noise, float* keep, float* count, int32_t N, int32_t H, int32_t W, int32_t block_size,
                      float gamma, hc_stream_t stream);
int hc_dropblock_apply(const void* x, const float* keep, const float* count, void* y, int64_t N, int32_t C, int64_t HW,
                       int32_t dtype, int32_t nhwc, hc_stream_t stream);
/* every DropBlock mask of a training step (YOLOv4 has 123) in one launch: `items` is a DEVICE array, noise / keep are
 * arenas indexed by item.off, counts fp32 [nitems] (zeroed here), max_pixels = largest N*H*W (sizes the grid). */
typedef struct {
    int64_t off;
    int32_t N, H, W, block_size;
    float gamma;
    int32_t pad_;
} hc_drop_item;
int hc_dropblock_mask_batched(const hc_drop_item* items, int32_t nitems, int64_t max_pixels, const float* noise, float* keep,
                              float* counts, hc_stream_t stream);

/* ---- YOLOv4 detection layer (holocron/models/detection/yolov4.py:269-420) ----
 * The logits of one scale are read in place: dtype 0 = fp32, 1 = bf16; element strides sn / sc / sp for image,
 * channel and pixel (NHWC bf16 with padded channels: sc = 1, sp = ld; NCHW fp32: sc = H*W, sp = 1); channel =
 * anchor * (5 + num_classes) + k.  anchors fp32 [A][2] (fractions of the image), all outputs in (h, w, anchor) order.
 * decode (_format_outputs, :269-300): boxes fp32 [N][H][W][A][4] = xyxy from
 *   b_xy = (scale_xy*sigmoid(t_xy) - 0.5*(scale_xy - 1) + cell) / (W, H), b_wh = clamp(exp(t_wh)*anchor, 0, 2);
 *   optional obj = sigmoid(objectness); optional score = max_c sigmoid(cls_c) * obj and label = argmax (post_process,
 *   :302-336); clamp01 clamps the boxes to [0, 1] like post_process does.
 * assign (_build_targets, :350-373): obj_mask uint8 [N][H][W][A] = 1 at the cell of each ground-truth centre and the
 *   anchor with the best origin-centred IoU; cell_gt uint8 [N][H][W] = 1 where noobj_mask is cleared.  gt_img int32 [G].
 * loss (_compute_losses, :394-420): sums fp32 [4] = raw sums of (sigmoid(o) - max_k IoU)^2, sigmoid(o)^2 over cells
 *   without ground truth, min_k (1 - IoU_k + penalty_k) (ciou_loss == diou_loss in the reference), mean_c BCE; the host
 *   applies lambda / N.  gt_off int32 [N+1] = per-image ranges into gt_boxes / gt_labels.
 * loss_bwd: dlogits (same dtype / strides as logits) = sum_i gcoef[i] * d sums[i] / d logits, including the gradient
 *   that reaches the boxes through the IoU objectness target (the reference does not detach it).  dlogits must ARRIVE ZEROED: the
 *   kernel writes the objectness gradient of every predictor and the box / class gradients of the assigned ones only. */
int hc_yolo_decode(const void* logits, int32_t dtype, int64_t sn, int64_t sc, int64_t sp, int32_t N, int32_t H, int32_t W,
                   int32_t A, int32_t num_classes, const float* anchors, float scale_xy, float* boxes, float* obj,
                   float* score, int64_t* label, int32_t clamp01, hc_stream_t stream);
int hc_yolo_assign(const float* gt_boxes, const int32_t* gt_img, int32_t G, const float* anchors, int32_t N, int32_t H,
                   int32_t W, int32_t A, uint8_t* obj_mask, uint8_t* cell_gt, hc_stream_t stream);
int hc_yolo_loss_fwd(const void* logits, int32_t dtype, int64_t sn, int64_t sc, int64_t sp, int32_t N, int32_t H, int32_t W,
                     int32_t A, int32_t num_classes, const float* anchors, float scale_xy, const float* gt_boxes,
                     const int64_t* gt_labels, const int32_t* gt_off, const uint8_t* obj_mask, const uint8_t* cell_gt,
                     float* sums, hc_stream_t stream);
int hc_yolo_loss_bwd(const void* logits, int32_t dtype, int64_t sn, int64_t sc, int64_t sp, int32_t N, int32_t H, int32_t W,
                     int32_t A, int32_t num_classes, const float* anchors, float scale_xy, const float* gt_boxes,
                     const int64_t* gt_labels, const int32_t* gt_off, const uint8_t* obj_mask, const uint8_t* cell_gt,
                     const float* gcoef, void* dlogits, hc_stream_t stream);

/* ---- YOLOv1 / YOLOv2 (holocron/models/detection/yolo.py:48-215): _YOLO._compute_losses, to_isoboxes, post_process on the
 * formatted predictions (fp32, contiguous): pred_boxes [N][H][W][A][4] (xc, yc, w, h), pred_o [N][H][W][A], pred_scores
 * [N][H][W][As][nc] with As = 1 (YOLOv1) or A.  cell_rel != 0: box centres are relative to their cell (YOLOv1.to_isoboxes,
 * yolo.py:140-163), else absolute (YOLOv2.to_isoboxes, yolov2.py:157-173).  Targets: gt_boxes [G][4] xyxy in [0, 1],
 * gt_labels int64 [G], gt_img int32 [G] (image of each box), gt_off int32 [N + 1].
 * loss_fwd: sums[4] = obj, noobj, bbox, clf (before the lambda / N scaling); assign: int32 [2 * G] scratch kept for the
 * backward pass, mark: uint8 [N*H*W*A] (1 where a box was assigned).  The objectness target is the differentiable IoU and the
 * sqrt(w), sqrt(h) term runs over every box of the image, like the reference (yolo.py:111-120).
 * loss_bwd: grad_sums[4] (device) -> d_boxes, d_o, d_scores (same shapes as the predictions, overwritten).
 * decode: boxes [N*H*W*A][4] xyxy (clamped to [0, 1] when clamp01); optional score = max_c p_c * objectness and its label
 * (b_scores [N*H*W*A][nc], already repeated per anchor as in YOLOv1.forward, yolo.py:366-367). ---- */
int hc_yolo1_loss_fwd(const float* pred_boxes, const float* pred_o, const float* pred_scores, int32_t N, int32_t H, int32_t W,
                      int32_t A, int32_t As, int32_t nc, int32_t cell_rel, int32_t ignore_high_iou, const float* gt_boxes,
                      const int64_t* gt_labels, const int32_t* gt_img, const int32_t* gt_off, int32_t G, int32_t* assign,
                      uint8_t* mark, float* sums, hc_stream_t stream);
int hc_yolo1_loss_bwd(const float* pred_boxes, const float* pred_o, const float* pred_scores, int32_t N, int32_t H, int32_t W,
                      int32_t A, int32_t As, int32_t nc, int32_t cell_rel, int32_t ignore_high_iou, const float* gt_boxes,
                      const int64_t* gt_labels, const int32_t* gt_img, const int32_t* gt_off, int32_t G, const int32_t* assign,
                      const uint8_t* mark, const float* grad_sums, float* d_boxes, float* d_o, float* d_scores,
                      hc_stream_t stream);
int hc_yolo1_decode(const float* b_coords, const float* b_o, const float* b_scores, int32_t N, int32_t H, int32_t W, int32_t A,
                    int32_t nc, int32_t cell_rel, int32_t clamp01, float* boxes, float* score, int64_t* label,
                    hc_stream_t stream);

/* ---- YOLOv1._format_outputs (holocron/models/detection/yolo.py:314-334) and YOLOv2._format_outputs
 * (holocron/models/detection/yolov2.py:175-200): raw head output (fp32, any strides) -> boxes [N][H][W][A][4], objectness
 * [N][H][W][A], class distribution [N][H][W][As][nc] (fp32, contiguous; As = 1: one softmax per cell, YOLOv1; As = A: YOLOv2).
 * Replaces the reshape / sigmoid / exp / softmax / stack chain of those lines and its autograd.
 * layout int64 [8], HOST memory, element strides of the logits: {sn, si, sj, sa, sk, cls0, ca, cc} - box / objectness logit k of
 * predictor (n, i, j, a) sits at n*sn + i*si + j*sj + a*sa + k*sk, class logit c at cls0 + n*sn + i*si + j*sj + a*ca + c*cc.
 * v2 == 0: boxes = sigmoid of the four logits (YOLOv1).  v2 != 0: ((sigmoid(tx) + j) / W, (sigmoid(ty) + i) / H,
 * anchors[a][0] * exp(tw), anchors[a][1] * exp(th)) with anchors fp32 [A][2] on the device.
 * format_bwd: cotangents g_boxes / g_obj / g_scores (contiguous, nullptr = zero) and the forward's `scores` -> dx in the layout
 * dx_layout (every logit is written exactly once: no zero fill needed). ---- */
int hc_yolo_format_fwd(const float* x, const int64_t* layout, int32_t N, int32_t H, int32_t W, int32_t A, int32_t As, int32_t nc,
                       int32_t v2, const float* anchors, float* boxes, float* obj, float* scores, hc_stream_t stream);
int hc_yolo_format_bwd(const float* x, const int64_t* layout, const int64_t* dx_layout, int32_t N, int32_t H, int32_t W, int32_t A,
                       int32_t As, int32_t nc, int32_t v2, const float* anchors, const float* scores, const float* g_boxes,
                       const float* g_obj, const float* g_scores, float* dx, hc_stream_t stream);

/* ---- DarkNet-19 / 24 bodies and the YOLOv2 passthrough (darknet.py:83, darknetv2.py:94, nn/functional.py:116-136), NHWC bf16,
 * C % 8 == 0.
 * maxpool2: nn.MaxPool2d(2) (floor mode); idx uint16 [N][H/2][W/2][C/8]: 2 bits per channel, the first maximum in row-major
 *           order; bwd routes each gradient to that position (dx dense [N][H][W][C], zero elsewhere).
 * space_to_depth: concat_downsample2d.  forward: src dense [N][OH*s][OW*s][C] -> dst[n][oh][ow][c0 + (a*s + b)*C + c] with `ld`
 *           channels per pixel (a concat buffer); backward != 0: src is that [N][OH][OW][ld] gradient, dst the dense one.
 * leaky_bwd: dy = g * (out > 0 ? 1 : slope), the gradient of ReLU (slope 0) / LeakyReLU from the stored output. ---- */
int hc_maxpool2_fwd(const void* x, void* out, void* idx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream);
int hc_maxpool2_bwd(const void* g, const void* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, hc_stream_t stream);
int hc_space_to_depth(const void* src, void* dst, int32_t ld, int32_t c0, int32_t N, int32_t OH, int32_t OW, int32_t C,
                      int32_t scale, int32_t backward, hc_stream_t stream);
int hc_leaky_bwd(const void* g, int32_t g_ld, const void* out, void* dy, int64_t npix, int32_t C, float slope,
                 hc_stream_t stream);

/* ---- depthwise 3x3 convolution, pad 1, stride 1 | 2, NHWC bf16 (ReXBlock, rexnet.py:111-124; FReLU,
 * nn/modules/activation.py:58-82).  C % 8 == 0 (pad channels carry zero weights).  wpk: fp32 tap-major [9][C] from
 * hc_dw3x3_pack (flip = 1 gives the taps of the stride-1 data gradient).  fwd optionally accumulates the BatchNorm
 * statistics sum / sum of squares of the fp32 results into stats [HC_STAT_REPLICAS][2][C] (zeroed by the caller).
 * wgrad: dw fp32 OIHW [Creal][1][3][3] (= or +=); ws scratch of hc_dw3x3_wgrad_ws_bytes(C). ---- */
int hc_dw3x3_pack(const float* w, float* out, int32_t C, int32_t Cpad, int32_t flip, hc_stream_t stream);
/* ... and for every depthwise kernel of a model in one launch: `items` is a DEVICE array (uploaded once; the pointers do not change
 * between steps), max_cpad the largest Cpad among them. */
typedef struct {
    const float* w;
    float* out;
    int32_t C, Cpad, flip, pad_;
} hc_dwpack_item;
int hc_dw3x3_pack_multi(const hc_dwpack_item* items, int32_t nitems, int32_t max_cpad, hc_stream_t stream);
int hc_dw3x3_fwd(const void* x, const float* wpk, void* y, float* stats, int32_t N, int32_t H, int32_t W, int32_t C,
                 int32_t stride, hc_stream_t stream);
int hc_dw3x3_dgrad(const void* dy, const float* wpk, const float* wpk_flipped, void* dx, int32_t N, int32_t H, int32_t W,
                   int32_t C, int32_t stride, hc_stream_t stream);
int64_t hc_dw3x3_wgrad_ws_bytes(int32_t C);
int hc_dw3x3_wgrad(const void* x, const void* dy, void* ws, float* dw, int32_t N, int32_t H, int32_t W, int32_t C,
                   int32_t Creal, int32_t stride, int32_t accumulate, hc_stream_t stream);

/* elementwise max of two NHWC bf16 tensors and its gradient (FReLU: max(x, bn(conv(x))), activation.py:79-82; ties
 * split the gradient evenly like torch.max). nelem % 8 == 0. */
int hc_max_fwd(const void* a, const void* b, void* out, int64_t nelem, hc_stream_t stream);
int hc_max_bwd(const void* a, const void* b, const void* g, void* da, void* db, int64_t nelem, hc_stream_t stream);

/* The MLP of ReXNet's squeeze-excite block on the pooled vectors (holocron/models/classification/rexnet.py:38-66: the nn.Sequential
 * of conv_sequence(C, C / r, act, BatchNorm2d, kernel_size=1, bias=False) and conv_sequence(C / r, C, Sigmoid, None, kernel_size=1)
 * - the sigmoid itself is applied by hc_se_scale_fwd), training-mode BatchNorm over the N rows; the six GEMMs run on the matrix cores
 * with operands rounded to bf16 (as every convolution of this library), fp32 accumulation, statistics and stored intermediates:
 *     h1 = pooled W1^T ; h = act(gamma (h1 - mean) rstd + beta) ; logits = h W2^T + b2          (act 0 none | 1 ReLU | 6 ReLU6)
 * hc_se_mlp_fwd: two launches; writes logits (bf16 [N][Cp], channels [C, Cp) zero), h1 [N][R], the row-tile partial sums `part`
 * (hc_se_mlp_part_floats(N, R) floats), stat = {mean[R], rstd[R]}, and updates running_mean / running_var (momentum, unbiased variance)
 * and num_batches_tracked when they are given.  hc_se_mlp_bwd: four launches; from dl (bf16 [N][Cp], the gradient of the logits) and the
 * tensors the forward saved it writes dpool (fp32 [N][Cp], pad channels zero), dw1 [R][C], dgamma / dbeta [R], dw2 [C][R], db2 [C]
 * (plain stores, not accumulated); g [N][R] and part2 (same size as part) are scratch.  R <= 128, Cp >= C rounded up to 16 and
 * Cp % 8 == 0, pooled and dl 16-byte aligned with finite (zero) pad channels.  All sums run in a fixed order. */
typedef struct {
    const float* pooled;             /* fp32 [N][Cp]: global average pool of the block input (hc_gap_fwd) */
    const float* w1;                 /* fp32 [R][C]: first 1x1 conv, no bias */
    const float* gamma;              /* fp32 [R] BatchNorm weight */
    const float* beta;               /* fp32 [R] BatchNorm bias */
    float* running_mean;             /* fp32 [R] or NULL (forward) */
    float* running_var;              /* fp32 [R] or NULL (forward) */
    int64_t* num_batches_tracked;    /* or NULL (forward) */
    const float* w2;                 /* fp32 [C][R]: second 1x1 conv */
    const float* b2;                 /* fp32 [C] or NULL */
    float* h1;                       /* fp32 [N][R]: forward output, backward input */
    float* part;                     /* forward output */
    float* stat;                     /* fp32 [2][R]: forward output, backward input */
    void* logits;                    /* bf16 [N][Cp]: forward output */
    const void* dl;                  /* bf16 [N][Cp]: backward input */
    float* g;                        /* fp32 [N][R]: backward scratch */
    float* part2;                    /* backward scratch */
    float* dpool;                    /* fp32 [N][Cp]: backward output */
    float* dw1;                      /* fp32 [R][C] */
    float* dgamma;                   /* fp32 [R] or NULL (both) */
    float* dbeta;                    /* fp32 [R] */
    float* dw2;                      /* fp32 [C][R] */
    float* db2;                      /* fp32 [C] or NULL */
    int32_t N, C, Cp, R, act;
    float eps, momentum;
} hc_se_mlp_desc;
int64_t hc_se_mlp_part_floats(int32_t N, int32_t R);
int hc_se_mlp_fwd(const hc_se_mlp_desc* d, hc_stream_t stream);
int hc_se_mlp_bwd(const hc_se_mlp_desc* d, hc_stream_t stream);

/* Squeeze-excite gate of ReXNet (rexnet.py:63-66) fused with the ReLU6 that follows it (rexnet.py:129):
 * out = act(z * sigmoid(l[n][c])), z NHWC bf16 [N][HW][C], l bf16 [N][C] gate logits, act 0 | 6 (ReLU6).
 * bwd_gate: dgate fp32 [N][C] = sum_hw g*mask*z and dlogits bf16 [N][C] = dgate * s (1 - s);
 * bwd_apply: dz = g*mask*s + dpool[n][c] / HW with dpool fp32 [N][C] the gradient of the pooled input. */
int hc_se_scale_fwd(const void* z, const void* gate_logits, void* out, int64_t N, int64_t HW, int32_t C, int32_t act,
                    hc_stream_t stream);
int hc_se_scale_bwd_gate(const void* g, const void* z, const void* gate_logits, float* dgate, void* dlogits, int64_t N,
                         int64_t HW, int32_t C, int32_t act, hc_stream_t stream);
int hc_se_scale_bwd_apply(const void* g, const void* z, const void* gate_logits, const float* dpool, void* dz, int64_t N,
                          int64_t HW, int32_t C, int32_t act, hc_stream_t stream);

/* SlimConv2d channel gate + fold (holocron/nn/modules/conv.py:352-364): s = sigmoid(gate_logits),
 * top[j] = x[j] s[j] + x[j+C/2] s[j+C/2], bot[j] = x[j] s[C-1-j] + x[j+C/2] s[C-1-j-C/2], j < C/2.
 * x NHWC bf16 with x_ld >= C channels per pixel, gate_logits bf16 [N][l_ld], top/bot NHWC bf16 with out_ld >= C/2
 * channels per pixel (pad channels zeroed).  bwd_gate: dlogits bf16 [N][l_ld]; bwd_apply: dx (x_ld wide) including
 * dpool[n][c] / HW, the gradient that reaches x through the global average pool of the gate. */
int hc_slim_fold_fwd(const void* x, int32_t x_ld, const void* gate_logits, int32_t l_ld, void* top, void* bot, int32_t out_ld,
                     int64_t N, int64_t HW, int32_t C, hc_stream_t stream);
int hc_slim_fold_bwd_gate(const void* x, int32_t x_ld, const void* gate_logits, int32_t l_ld, const void* gtop, const void* gbot,
                          int32_t out_ld, void* dlogits, int64_t N, int64_t HW, int32_t C, hc_stream_t stream);
int hc_slim_fold_bwd_apply(const void* gate_logits, int32_t l_ld, const void* gtop, const void* gbot, int32_t out_ld,
                           const float* dpool, int32_t dpool_ld, void* dx, int32_t x_ld, int64_t N, int64_t HW, int32_t C,
                           hc_stream_t stream);

/* NormConv2d helpers (holocron/nn/modules/conv.py:55-147, holocron/nn/functional.py:322-413).
 * patch_stats: mean / rsqrt(biased var + eps) over the Cin*KH*KW entries of every unfolded patch (zero padding
 *   counts as zeros) of x NHWC bf16 [N][H][W][x_ld] with C real channels -> fp32 [N*OH*OW] each.
 * bwd_scale: gs = g * rstd[pixel] (bf16, the dy of the weight-gradient GEMM) and red fp32 [HC_STAT_REPLICAS][2][C] +=
 *   per-channel sums of g (bias gradient) and g * rstd * mean (the correction dW[co][k] -= sum_p g r m). */
int hc_patch_stats(const void* x, int32_t x_ld, float* mean, float* rstd, int32_t N, int32_t H, int32_t W, int32_t C, int32_t KH,
                   int32_t KW, int32_t stride, int32_t pad, float eps, hc_stream_t stream);
int hc_normconv_bwd_scale(const void* g, const float* mean, const float* rstd, void* gs, float* red, int64_t npix, int32_t C,
                          hc_stream_t stream);

/* fp8 helpers of the C5 inference path: quantize NHWC bf16 -> e4m3 bytes (out = fp8(clamp(x * inv_scale))), with optional
 * channel padding dst_ld >= C (zeros); global average pool of an fp8 NHWC tensor -> fp32 [N][C] (times `scale`). */
/* im2col of a tiny-Cin input straight to fp8: col uint8 [N][OH][OW][Kpad] = fp8(clamp(x * inv_scale)) (fp8 inference stem). */
int hc_im2col_small_fp8(const float* x, void* col, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t OH, int32_t OW, int32_t KH,
                        int32_t KW, int32_t stride, int32_t pad, int32_t Kpad, float inv_scale, hc_stream_t stream);
int hc_quantize_fp8(const void* src_bf16, int32_t src_ld, void* dst_fp8, int32_t dst_ld, int64_t npix, int32_t C, float inv_scale,
                    hc_stream_t stream);
int hc_gap_fp8(const void* x_fp8, float* y, int32_t N, int32_t HW, int32_t ld, int32_t C, float scale, hc_stream_t stream);

/* ---- MobileOne over-parameterised blocks, training form (holocron/models/classification/mobileone.py:31-176):
 * out = act(sum_b BN_b(y_b)) over B <= HC_MSBN_MAX_BRANCHES parallel branch outputs y_b (NHWC bf16, y_b[p][c] at
 * y[b] + p * ld[b] + c), every branch with its own BatchNorm2d.  DepthConvBlock.forward (:66-67): the branches are K depthwise
 * 3x3 + one depthwise 1x1 (hc_dw3x3_fwd planes) + the input; PointConvBlock.forward (:120-121): K dense 1x1 (one stacked
 * hc_conv_gather) + the input.  act: 0 none | 1 ReLU.
 *   hc_msbn_finalize      per-branch statistics ([HC_STAT_REPLICAS][2][stats_ld] sums) -> coef [B][2][C] (scale, shift),
 *                         save [B][2][C] (mean, rstd); updates running statistics like nn.BatchNorm2d (training) or uses
 *                         them (eval).  Channels >= c_valid are layout padding (scale = shift = 0).
 *   hc_msbn_apply         out[p][c] = act(sum_b scale_b y_b + shift_b); out dense [npix][C]; optional out_stats
 *                         [HC_STAT_REPLICAS][2][C] (zeroed by the caller) += sum / sum of squares of the stored output.
 *   hc_msbn_bwd_reduce    red [HC_STAT_REPLICAS][B + 1][C] (zeroed by the caller) += sum gz, sum gz y_b with
 *                         gz = g * act'(out)
 *   hc_msbn_bwd_finalize  red -> dgamma_b, dbeta_b (= or +=), bcoef [B][3][C]: dy_b = k1 gz + k2 y_b + k3
 *   hc_msbn_bwd_apply     writes dy_b to dy[b] + p * dld[b] + c for every branch with dy[b] != NULL
 *   hc_dwrep_dgrad        DepthConvBlock data gradient: dx = sum_b dwconv3x3^T(dy_b, wpk_b) (+ extra, stride 1 only: the
 *                         identity branch's gradient); wpk_b forward tap-major fp32 [9][C] from hc_dw3x3_pack ---- */
#define HC_MSBN_MAX_BRANCHES 6
typedef struct {
    const float* stats;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    int64_t* num_batches_tracked;
    float* dgamma;
    float* dbeta;
    int32_t stats_ld;
    float eps, momentum;
    int32_t pad_;
} hc_msbn_branch;
typedef struct {
    hc_msbn_branch br[HC_MSBN_MAX_BRANCHES];
    float* coef;
    float* save;
    const float* red;
    float* bcoef;
    int64_t count;             /* N*H*W */
    int32_t B, C, c_valid;
    int32_t training;          /* 0: running statistics (eval) */
    int32_t accumulate;        /* bwd_finalize: dgamma / dbeta += */
    int32_t pad_;
} hc_msbn_desc;
typedef struct {
    const void* y[HC_MSBN_MAX_BRANCHES];
    void* dy[HC_MSBN_MAX_BRANCHES];
    int32_t ld[HC_MSBN_MAX_BRANCHES];
    int32_t dld[HC_MSBN_MAX_BRANCHES];
    int64_t npix;
    int32_t B, C;
} hc_msbn_io;
int hc_msbn_finalize(const hc_msbn_desc* d, hc_stream_t stream);
int hc_msbn_bwd_finalize(const hc_msbn_desc* d, hc_stream_t stream);
int hc_msbn_apply(const hc_msbn_io* io, const float* coef, void* out, float* out_stats, int32_t act, hc_stream_t stream);
int hc_msbn_bwd_reduce(const hc_msbn_io* io, const void* g, int32_t g_ld, const void* out, float* red, int32_t act,
                       hc_stream_t stream);
int hc_msbn_bwd_apply(const hc_msbn_io* io, const void* g, int32_t g_ld, const void* out, const float* bcoef, int32_t act,
                      hc_stream_t stream);
int hc_dwrep_dgrad(const void* const* dy, const float* const* wpk, int32_t nplanes, const void* extra, void* dx, int32_t N,
                   int32_t H, int32_t W, int32_t C, int32_t stride, hc_stream_t stream);

/* ---- the steps either side of the training step.
 * mixup (holocron/utils/data/collate.py:39-64): out[i] = lam x[i] + (1 - lam) x[perm[i]] over N rows of D elements (dtype 0 fp32,
 *   1 bf16; out must not alias x); mixup_onehot: class indices -> out fp32 [N][C] = lam onehot(t[i]) + (1 - lam) onehot(t[perm[i]]).
 * topk_hits (holocron/trainer/classification.py:60-66 without the per-batch .item() synchronisations): counters[0] += rows whose
 *   target has the highest logit, counters[1] += rows whose target is among the k highest (k > 1), counters[2] += N. ---- */
int hc_mixup(const void* x, const int64_t* perm, void* out, int64_t N, int64_t D, int32_t dtype, float lam, hc_stream_t stream);
int hc_mixup_onehot(const int64_t* target, const int64_t* perm, float* out, int64_t N, int32_t C, float lam, hc_stream_t stream);
int hc_topk_hits(const float* logits, const int64_t* target, int64_t N, int32_t C, int32_t k, float* counters, hc_stream_t stream);

const char* hc_version(void);

#ifdef __cplusplus
}
#endif
#endif
