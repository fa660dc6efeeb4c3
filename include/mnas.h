/*
 * mnas.h -- C ABI of libmnas_hip.so: the MI355X (gfx950) kernels behind the MNASNet training hot path.
 *
 * The reference (snakers4/mnasnet-pytorch) has no FFI of its own: its hot path is
 *     ConvBlock.forward      src/models/mnasnet.py:58-62   (conv -> BatchNorm2d -> ReLU)
 *     MBConv_block.forward   src/models/mnasnet.py:131-137 (x + seq(x))
 *     loss.backward()        src/train.py:439              (autograd mirror of the above)
 * executed by PyTorch ATen.  This library is what a maintainer binds instead of ATen for that path
 * (INTEGRATION.md shows the ctypes stub).  Every entry point:
 *   - takes plain pointers + sizes (no torch types), device pointers unless stated otherwise;
 *   - launches on the HIP stream passed as `stream` (a hipStream_t cast to void*), never synchronises,
 *     never allocates or frees, keeps no pointer after returning, has no global mutable state;
 *   - returns 0 (MNAS_OK) or a non-zero hipError_t / MNAS_E* code; never throws across the ABI;
 *   - is bit-reproducible: no float atomics, every partial-sum table is reduced in a fixed order.
 *
 * Data layout in HBM
 *   activations / activation gradients : NHWC, bf16, dense, C % 8 == 0 (16-byte channel groups)
 *   network input                      : NCHW fp32 (what train.py:427 hands over), read by the stem kernel
 *   network output / its gradient      : NCHW fp32 (what classifiers.py:109 consumes)
 *   BatchNorm is never applied in place: a conv kernel writes the RAW conv output y (bias included) and
 *   per-workgroup partial (sum, sum of squares); mnas_bn_fwd_finalize turns them into a per-channel
 *   (scale, shift); every CONSUMER applies relu(scale*y+shift) while loading ("act-on-load").
 *   In backward the consumer of a gradient computes dy = c1*(g*[scale*y+shift>0]) + c2*y + c3 while
 *   loading ("dy-on-load"); c1..c3 come from mnas_bn_bwd_finalize.
 *
 * BN coefficient block ("bnbuf"): float[8][C] per ConvBlock application
 *   row 0 scale  s = gamma*invstd         row 4 c3
 *   row 1 shift  t = beta - mean*s        row 5 batch mean
 *   row 2 c1                              row 6 invstd
 *   row 3 c2                              row 7 (reserved)
 */
#ifndef MNAS_H
#define MNAS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNAS_OK 0
#define MNAS_EINVAL 10001   /* unsupported shape / argument */
#define MNAS_BN_ROWS 8

/* "act-on-load" input: value = scale ? relu(scale[c]*data+shift[c]) : data */
typedef struct MnasActIn {
    const void*  data;    /* bf16 NHWC */
    const float* scale;   /* [C] or NULL */
    const float* shift;   /* [C] or NULL */
} MnasActIn;

/* "dy-on-load" input: dy = c1*(g*[s*y+t>0]) + c2*y + c3, coef = bnbuf rows 0..4.  y == NULL && coef == NULL: g IS dy. */
typedef struct MnasGradIn {
    const void*  g;       /* bf16 NHWC: dL/d(activated output) */
    const void*  y;       /* bf16 NHWC: saved raw conv output */
    const float* coef;    /* float[5][C] */
} MnasGradIn;

int mnas_version(void);                 /* ABI version: 8 (round 6: + mnas_probe_copy4 / _read; 7 = round 5: + mnas_se_fc_*; 6 = struct layouts changed in rounds 2, 3, twice in round 4 -- 4 = the tiled-block forms,
                                         * 5 = squeeze-excite on load: MnasConvGemm.gate, MnasPwBwd.seg_px -- and in round 5: 6 = the opt-in
                                         * forms that lost their A/B are gone (MnasPwBwd.dy_out / red4, MnasDwBwd.src_* / g_gate / g_bias,
                                         * mnas_dw_exp_*, mnas_irb_*, mnas_gram*, mnas_se_bn_assemble); their opcode numbers stay retired) */
const char* mnas_arch(void);            /* "gfx950" */

/* ---- 1x1 / dense kxk convolution as an implicit GEMM on MFMA (bf16 in, fp32 accumulate) -------------
 * Replaces ATen conv2d forward / conv2d input-gradient for ConvBlock's groups==1 convs
 * (mnasnet.py:48-54 with kernel_size 1 or 3).
 * mode 0 (forward):  out[n,ho,wo,co] = bias[co] + sum_{kh,kw,ci} act(in)[n,ho*s+kh-p,wo*s+kw-p,ci] * W
 *                    in  = (N,Hi,Wi,Ci) read through `act`;  out = (N,Ho,Wo,Co)
 * mode 1 (dgrad):    out[n,hi,wi,ci] = resid + sum_{kh,kw,co} dy[n,(hi+p-kh)/s,(wi+p-kw)/s,co] * W
 *                    in  = dy (N,Hi,Wi,Ci) read through `grad` (Hi,Wi,Ci are the FORWARD OUTPUT dims and
 *                    channels), out = (N,Ho,Wo,Co) are the FORWARD INPUT dims/channels.
 * w: packed bf16 [Co_pad16][Kpad32], K = kh*kw*Ci ordered (tap, ci) -- see mnas_pack_weights.
 * stats (optional, forward): float[2][Co][nparts] partial (sum, sumsq) of the fp32 output, one column per
 * pixel-workgroup (channel-major so the finalize kernel reads one channel contiguously); fully overwritten
 * (no memset needed). */
typedef struct MnasConvGemm {
    int32_t mode;
    int32_t N, Hi, Wi, Ci;
    int32_t Ho, Wo, Co;
    int32_t kh, kw, stride, pad;
    int32_t nparts;          /* pixel-workgroups (grid.x); 1..4096 */
    int32_t reserved;
    MnasActIn  act;
    MnasGradIn grad;
    const void*  w;
    const float* bias;       /* [Co] or NULL */
    const void*  resid;      /* bf16 (N,Ho,Wo,Co) or NULL */
    void*        out;        /* bf16 (N,Ho,Wo,Co) */
    float*       stats;      /* or NULL */
    /* mode 1 only, optional: fuse the NEXT BatchNorm-backward reduction into this kernel's epilogue.  `out` is the
     * gradient g of some ConvBlock's activated output; red_y is that ConvBlock's saved raw output (same shape as
     * out), red_bn its bnbuf.  stats then receives float[2][Co][nparts] partial (sum dz, sum dz*xhat), exactly what
     * mnas_bn_bwd_reduce would produce from (out, red_y, red_bn). */
    const void*  red_y;
    const float* red_bn;
    /* ABI 5, mode 0, 1x1 only, optional: per-(image, input channel) multiplier float[N][Ci] applied AFTER the activation on
     * load, value = relu(scale*x+shift) * gate[n][c] -- the squeeze-excite excitation folded into the project conv's load
     * (mnas_se_gate writes the table).  Requires act.scale; supported shapes: mnas_conv_gemm_gate_ok. */
    const float* gate;
} MnasConvGemm;
int mnas_conv_gemm_gate_ok(int N, int HW, int Ci, int Co);
int mnas_conv_gemm(const MnasConvGemm* a, void* stream);
/* Pixels per tile (64 or 128) mnas_conv_gemm uses for a problem with M output pixels, Co output channels and reduction
 * length K (= kh*kw*Ci of that mode): callers size nparts in whole tiles with it (host-side, no launch). */
int mnas_conv_gemm_tile_pixels(int M, int Co, int K);
/* Preferred nparts for a launch (host-side, no launch): > 0 where the kernel sizes its own persistent grid (the 1x1
 * forward: DMA-pipelined kernel, csrc/mnas_pwf.hip), -1 = caller's choice in whole tiles (mnas_conv_gemm_tile_pixels).
 * taps = kh*kw. */
int mnas_conv_gemm_parts(int mode, int M, int Ci, int Co, int taps);
/* Same question for the dense k x k convs, which depend on the image geometry: > 0 (= N) where mnas_conv_gemm runs the
 * whole-image kernel (csrc/mnas_dimg.hip: 3x3, pad 1, output plane <= 256 pixels, N >= 32; mode 1: stride 1 and a
 * materialised dy), -1 otherwise.  mode 1 arguments as in MnasConvGemm (Hi,Wi,Ci = dy dims, Ho,Wo,Co = result dims). */
int mnas_conv_img_parts(int mode, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int k, int stride, int pad);

/* ---- input gradient of the stride-2 dense 3x3 convs as a transposed convolution (csrc/mnas_tconv.hip): one GEMM per 2x2
 * output block over the four dy pixels it depends on; no per-element gather.  dy: bf16 (N,Ho,Wo,Co), MATERIALISED
 * (mnas_dy_materialize); w: mnas_pack_weights(MNAS_PACK_TCONV, Co, Ci, 3, 3); out: bf16 (N,2Ho,2Wo,Ci); optional fused
 * BatchNorm-backward reduce as in mnas_conv_gemm mode 1 (stats: float[2][Ci][nparts]).  Supported: the conv's input plane is
 * exactly 2Ho x 2Wo, 4*Co <= 256, 4*Ci <= 96 (mnas_tconv_supported); otherwise use mnas_conv_gemm(mode 1). */
typedef struct MnasTconvDgrad {
    int32_t N, Ho, Wo, Co, Ci, nparts;
    const void* dy;
    const void* w;
    void* out;
    float* stats;
    const void* red_y;
    const float* red_bn;
} MnasTconvDgrad;
int mnas_tconv_dgrad(const MnasTconvDgrad* a, void* stream);
int mnas_tconv_supported(int Ho, int Wo, int Co, int Ci);
int mnas_tconv_parts(int N, int Ho, int Wo, int Co, int Ci);     /* preferred nparts (persistent workgroups) */

/* ---- weight gradient of the same convs: dW[co][tap][ci] = sum_pix dy[pix][co] * act(x)[src(pix,tap)][ci]
 * Replaces ATen conv2d weight-gradient.  x = (N,Hi,Wi,Ci) forward input, dy = (N,Ho,Wo,Co).
 * partial: float[nsplit][Co][K] (K = kh*kw*Ci), one slab per pixel split, fully overwritten;
 * reduce + relayout with mnas_wgrad_finalize. */
typedef struct MnasConvWgrad {
    int32_t N, Hi, Wi, Ci;
    int32_t Ho, Wo, Co;
    int32_t kh, kw, stride, pad;
    int32_t nsplit;
    MnasActIn  x;
    MnasGradIn dy;
    float* partial;
} MnasConvWgrad;
int mnas_conv_wgrad(const MnasConvWgrad* a, void* stream);
/* Number of (cout x k) slabs mnas_conv_wgrad splits dW[Co][taps*Ci] into (the slab shape is chosen per (Co, K)): a launch
 * runs nsplit x slabs workgroups, so callers size nsplit as (workgroup budget) / slabs. */
int mnas_conv_wgrad_slabs(int Co, int Ci, int taps);

/* grad[co][ci][kh][kw] (reference layout, fp32) (+)= sum_s partial[s][co][tap*Ci+ci].  Deterministic (fixed summation
 * order, no atomics).  `partial` is scratch: with more than 256 splits the first row of every 128-row chunk is
 * overwritten by the chunk's sum (two-level reduction). */
int mnas_wgrad_finalize(float* partial, int nsplit, int Co, int Ci, int taps,
                        float* grad, int accumulate, void* stream);

/* ---- fused backward of a 1x1 conv: input gradient + weight-gradient partials + (optionally) the BatchNorm-backward
 * reduce of the ConvBlock that produced x, in one sweep over the pixels.  Replaces the pair
 * mnas_conv_gemm(mode 1) + mnas_conv_wgrad for the large-pixel-count layers (autograd mirror of mnasnet.py:48-62).
 * Supported channel pairs: mnas_pw_bwd_supported(Ci, Co) != 0 (the 112^2/56^2/28^2 pointwise convs of MNASNet-1.0).
 * wpartial: float[nparts][Co][Ci], one slab per workgroup, fully overwritten -> mnas_wgrad_finalize(wpartial, nparts,
 * Co, Ci, 1, grad, ...).  red_partial (or NULL): float[2][Ci][nparts] = (sum dz, sum dz*xhat) of (gin, red_y) under
 * red_bn -- the fused BatchNorm-backward reduce of the ConvBlock whose activated output gin is the gradient of (red_y =
 * its raw output; equals x.data when x is that block's virtual activation). */
typedef struct MnasPwBwd {
    int32_t M, Ci, Co, nparts;   /* pixels, conv input / output channels, persistent workgroups (<= 65535) */
    MnasActIn  x;                /* forward input (M,Ci): raw + producer's (scale,shift), or a materialised activation */
    MnasGradIn dy;               /* g, y (M,Co) and coef[5][Co] of this ConvBlock's BatchNorm */
    const void* w;               /* MNAS_PACK_DGRAD weights */
    const void* resid;           /* bf16 (M,Ci) added to gin, or NULL */
    void*  gin;                  /* bf16 (M,Ci) */
    float* wpartial;
    float* red_partial;
    const void*  red_y;
    const float* red_bn;
    /* round 4, "RECOMP" (the EXPAND conv's backward; mnas_pw_bwd_forms bit 1): dy.y == NULL && w_fwd != NULL -- the raw forward
     * output y of dy-on-load is not read but recomputed per tile as bf16(W act(x) + b_fwd) from the staged x tile, bit-identical
     * to what mnas_conv_gemm(mode 0) stored (same MFMA, same k order).  w_fwd: MNAS_PACK_FWD [Co_pad16][Ci_pad32]; Ci <= 32. */
    const void*  w_fwd;
    const float* b_fwd;
    /* round 4: store gin MASKED, dz = gin*[s*x+t>0] under red_bn (the mask its fused reduce computes anyway), for a consumer
     * that would otherwise re-derive it per element and window column (mnas_dw_bwd g_masked).  Out-stage forms with the fused
     * reduce only (mnas_pw_bwd_forms bit 2); MNAS_EINVAL otherwise. */
    int32_t gin_masked;
    /* ABI 5: > 0 = "segment mode": workgroup b owns the contiguous pixels [b*seg_px, (b+1)*seg_px) (nparts*seg_px >= M >
     * (nparts-1)*seg_px) instead of striding over the tiles of the whole tensor, so that wpartial[b] is the weight-gradient sum
     * over a KNOWN pixel range -- with seg_px dividing H*W, a fraction of one image: what mnas_se_proj_finalize needs. */
    int32_t seg_px;
} MnasPwBwd;
int mnas_pw_bwd(const MnasPwBwd* a, void* stream);
int mnas_pw_bwd_supported(int Ci, int Co);
/* ABI 5: pixels per tile (64 / 128) and channel slices of the launch for a supported channel pair (-1 otherwise): segment mode
 * wastes ceil(seg_px / tile) * tile - seg_px pixel slots per workgroup, a caller picks seg_px with that in view. */
int mnas_pw_bwd_tile_pixels(int Ci, int Co);
int mnas_pw_bwd_slices(int Ci, int Co);
int mnas_pw_bwd_forms(int Ci, int Co);      /* bit 1: RECOMP available for this channel pair, bit 2: out-stage form (gin_masked) */

/* ---- depthwise kxk (k in {3,5}, stride 1, pad k/2), LDS-tiled direct conv on the vector ALU ----------
 * Replaces ATen conv2d fwd/bwd for ConvBlock's groups==C convs (mnasnet.py:122-125, 76-81). */
typedef struct MnasDwFwd {
    int32_t N, H, W, C, k;
    int32_t nparts;          /* upper bound on workgroups; must be >= C/64 (channel blocks) */
    MnasActIn in;
    const float* w;          /* fp32 [k*k][C] (tap-major) */
    const float* bias;       /* [C] or NULL */
    void*  out;              /* bf16 (N,Ho,Wo,C) raw output */
    float* stats;            /* float[2][C][rows] or NULL, rows = mnas_dw_rows(N,H,W,C,k,nparts,0) (stride 2: which = 4) */
    /* ABI 6: 0 / 1 = stride 1 (the LDS row-ring sweeps); 2 = the depthwise conv of SepConv(reduce=True) (mnasnet.py:73-81):
     * Ho = (H + 2*(k/2) - k)/2 + 1, plain direct kernels (csrc/mnas_dw2.hip). */
    int32_t stride, reserved;
} MnasDwFwd;
int mnas_dw_fwd(const MnasDwFwd* a, void* stream);
/* Number of partial rows/columns a depthwise launch writes for this shape and nparts (host-side, no launch).
 * which = 0: forward statistics float[2][C][rows];
 * which = 1: both tables of a phase-0 (fused) backward launch: wpartial float[rows][k*k][C], reduce float[2][C][rows];
 * which = 2: the fused-reduce table of a phase-1 (input-gradient-only) launch;
 * which = 3: wpartial of a phase-2 (weight-gradient-only) launch.  Returns < 0 for unsupported shapes.
 * which = 4 / 7: the same as 0 / 3 for the STRIDE-2 kernels (MnasDwFwd.stride / MnasDwBwd.stride == 2). */
/* Diagnostics: geometry picked for a launch form (`which` as in mnas_dw_rows).  out[7] = {channel pairs per workgroup,
 * 4-column strips per workgroup, threads, strips per image row, channel blocks, LDS bytes, rows per DMA group}. */
int mnas_dw_geometry(int N, int H, int W, int C, int k, int which, int* out);
int mnas_dw_rows(int N, int H, int W, int C, int k, int nparts, int which);

typedef struct MnasDwBwd {
    int32_t N, H, W, C, k;
    int32_t nparts;
    MnasActIn  x;            /* forward input (act-on-load) */
    MnasGradIn dy;           /* gradient of the forward output (dy-on-load) */
    const float* w;          /* fp32 [k*k][C] */
    void*  gin;              /* bf16 (N,H,W,C): dL/d act(x) */
    float* wpartial;         /* float[rows1][k*k][C], rows1 = mnas_dw_rows(...,1), fully overwritten */
    /* optional fused BatchNorm-backward reduction for the producer of x (x.data = its raw output, red_bn = its
     * bnbuf): red_partial receives float[2][C][r] (sum dz, sum dz*xhat) of (gin, x.data), r = mnas_dw_rows(...,1) for
     * phase 0 and mnas_dw_rows(...,2) for phase 1; wpartial rows: which = 1 (phase 0) / 3 (phase 2) */
    const float* red_bn;
    float* red_partial;
    int32_t phase;           /* 0: one fused sweep (input gradient + weight gradient + reduce); 1: input gradient (+reduce)
                                only; 2: weight gradient only (lets the caller put the two on different streams) */
    int32_t stride;          /* ABI 6: 0 / 1 = stride 1; 2 = SepConv(reduce=True)'s depthwise conv (N,H,W = the conv's INPUT dims): only the
                                two-launch form -- phase 1 (input gradient, no fused reduce, no g_masked), phase 2 (weight gradient,
                                wpartial float[mnas_dw_rows(...,7)][k*k][C]) */
    /* round 4: dy.g already holds dz = g*[s*y+t>0] (written by mnas_pw_bwd with gin_masked): dy-on-read skips the mask.
     * Phase 0 with the fused reduce only; results are bit-identical to the plain form on the unmasked g. */
    int32_t g_masked, reserved;
} MnasDwBwd;
int mnas_dw_bwd(const MnasDwBwd* a, void* stream);
/* grad[c][0][kh][kw] (+)= sum_{p<nparts} wpartial[p][tap][c]   (pass nparts = rows1); wpartial is scratch like above */
int mnas_dw_wgrad_finalize(float* wpartial, int nparts, int C, int k, float* grad, int accumulate,
                           void* stream);

/* ---- stem: dense 3x3 stride 2 pad 1 on the fp32 NCHW network input (mnasnet.py:179) ------------------ */
typedef struct MnasStemFwd {
    int32_t N, H, W, Ho, Wo, Co;      /* Ci == 3 */
    int32_t nparts;
    const float* x;          /* fp32 NCHW (N,3,H,W) */
    const void*  w;          /* packed bf16 [Co_pad16][32]: mnas_pack_weights(MNAS_PACK_FWD, Co, 27, 1, 1) of
                                the reference [Co][3][3][3] tensor viewed as [Co][27] (k = ci*9+kh*3+kw) */
    const float* bias;
    void*  out;              /* bf16 (N,Ho,Wo,Co) */
    float* stats;            /* float[2][Co][nparts] */
    /* optional fused input pipeline (the normalisation the reference's dataset applies on the CPU: datasets.py:474-516
     * transforms.Normalize with the mean / std of classifiers.py:91-92): the conv reads in_affine[0][c] * x + in_affine[1][c];
     * in_u8 != 0: x is a uint8 NCHW image (a quarter of the PCIe / HBM bytes).  Band-kernel shapes only (Co == 32,
     * W % 4 == 0), otherwise MNAS_EINVAL. */
    const float* in_affine;  /* float[2][3] (scale, shift per input plane) or NULL */
    int32_t in_u8, reserved;
} MnasStemFwd;
int mnas_stem_fwd(const MnasStemFwd* a, void* stream);
typedef struct MnasStemWgrad {
    int32_t N, H, W, Ho, Wo, Co;
    int32_t nparts;
    const float* x;
    MnasGradIn dy;
    float* partial;          /* float[nparts][Co][27], fully overwritten */
    const float* in_affine;  /* as in MnasStemFwd: the weight gradient must see the same transformed input */
    int32_t in_u8, reserved;
} MnasStemWgrad;
int mnas_stem_wgrad(const MnasStemWgrad* a, void* stream);
/* Preferred nparts (persistent workgroups) for the stem launches: which = 0 forward, 1 weight gradient; -1 = caller's choice
 * (host-side, no launch).  Co == 32, W % 4 == 0 (and Wo % 8 == 0 for the weight gradient) run as band kernels
 * (csrc/mnas_stem.hip: input rows staged once per band in LDS); other shapes use the im2col staging of the GEMM kernels. */
int mnas_stem_parts(int which, int N, int H, int W, int Co);
/* Input gradient of the stem conv = dL/d image, fp32 NCHW (N,3,H,W): ATen conv2d input gradient of ConvBlock(3, Co, 3, stride 2,
 * pad 1) with dy-on-load; w = the reference fp32 [Co][3][3][3] weight (rounded to bf16 inside, like the forward's packed copy);
 * in_affine as in MnasStemFwd (the result is the gradient w.r.t. the RAW float image: scaled by in_affine[0][c]) or NULL.
 * Not on the training path (train.py:427) -- a plain gather kernel for autograd completeness (saliency, adversarial inputs). */
int mnas_stem_dgrad(const MnasGradIn* dy, const float* w, int N, int H, int W, int Ho, int Wo, int Co, const float* in_affine,
                    float* dx, void* stream);

/* ---- BatchNorm2d bookkeeping (replaces ATen native_batch_norm / native_batch_norm_backward) ---------- */
/* partial: float[2][C][nparts] (sum, sumsq over `count` elements per channel).
 * training=1: batch stats -> bnbuf rows 0,1,5,6; running_mean/var momentum update (unbiased var),
 *             num_batches_tracked (int64 device scalar, may be NULL) += 1.
 * training=0: bnbuf rows 0,1 from the running stats; nothing else is touched (partial may be NULL). */
int mnas_bn_fwd_finalize(const float* partial, int nparts, int C, double count,
                         const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         float momentum, float eps, int training, float* bnbuf, void* stream);
/* partial[0][c][p] = sum dz, partial[1][c][p] = sum dz*xhat over the rows of workgroup p, where
 * dz = g*[s*y+t>0], xhat = (y-mean)*invstd.  g,y: bf16 [rows][C]. */
int mnas_bn_bwd_reduce(const void* g, const void* y, const float* bnbuf, int64_t rows, int C,
                       int nparts, float* partial, void* stream);
/* out[rows][C] (bf16) = dy-on-load of (g, y, coef), materialised once.  mnas_conv_gemm(mode 1) and mnas_conv_wgrad accept the
 * result as a PLAIN gradient: grad.g = out, grad.y = grad.coef = NULL (no transform on load).  Used for the dense 3x3 convs,
 * whose kernels gather every dy element 2.25-10 times. */
int mnas_dy_materialize(const MnasGradIn* d, int64_t rows, int C, void* out_bf16, void* stream);
/* dgamma (+)= sum dz*xhat ; dbeta (+)= sum dz ; bnbuf rows 2..4 = c1,c2,c3 */
int mnas_bn_bwd_finalize(const float* partial, int nparts, int C, double count,
                         float* bnbuf, float* dgamma, float* dbeta, int accumulate, void* stream);

/* ---- one launch for the bookkeeping between two dependent backward kernels: mnas_bn_bwd_finalize (accumulate = 1) of the next
 * layer + up to two weight-gradient reductions (mnas_wgrad_finalize / mnas_dw_wgrad_finalize semantics, accumulate = 1).
 * level: 0 = none, 1 = whole reduction in one level (nsplit <= 256), 2 = first level of two (chunks of 128 rows folded into
 * each chunk's first row, in place), 3 = second level of the same table (call it in a LATER launch than its level 2).
 * bn_C == 0: no BatchNorm part. */
typedef struct MnasPostWgrad {
    float* partial;          /* float[nsplit][Co][taps*Ci]   (dw: float[nsplit][taps][Co]) */
    float* grad;             /* reference layout, accumulated into */
    int32_t nsplit, Co, Ci, taps, dw, level;
} MnasPostWgrad;
typedef struct MnasBwdPost {
    const float* bn_partial; /* float[2][C][nparts] */
    float* bnbuf;
    float* dgamma;
    float* dbeta;
    double count;
    int32_t bn_nparts, bn_C;
    MnasPostWgrad w1, w2;
} MnasBwdPost;
int mnas_bwd_post(const MnasBwdPost* p, void* stream);

/* ---- element-wise glue ------------------------------------------------------------------------------ */
/* out = act(a) + act(b)   (b.data may be NULL -> out = act(a)); rows x C bf16.  The MBConv_block residual
 * (mnasnet.py:133).  If out_nchw_f32 != NULL the result is ALSO/INSTEAD written as fp32 NCHW (N,C,H,W)
 * with rows = N*HW (the features output handed to AdaptiveAvgPool2d, classifiers.py:109). */
int mnas_add_act(const MnasActIn* a, const MnasActIn* b, int64_t rows, int C, void* out_bf16,
                 float* out_nchw_f32, int HW, void* stream);
/* Global average pool fused with the last BatchNorm+ReLU: out[n][c] (fp32) = mean over the HW pixels of act(a)[n][.][c]
 * (classifiers.py:49,109: AdaptiveAvgPool2d(1) on the features output, which is then never materialised), and its backward:
 * g[n][hw][c] (bf16 NHWC) = gpool[n][c] / HW. */
int mnas_pool_act(const MnasActIn* a, int N, int HW, int C, float* out, void* stream);
int mnas_pool_bwd(const float* gpool, int N, int HW, int C, void* g_bf16, void* stream);
/* bf16 NHWC <- fp32 NCHW  (incoming gradient of the features output) */
int mnas_nchw_f32_to_nhwc_bf16(const float* src, void* dst, int N, int C, int HW, void* stream);

/* ---- classifier head + loss (csrc/mnas_head.hip) -------------------------------------------------------
 * Replaces, for FineTuneModelPool.classifier (classifiers.py:56-89: nn.Sequential of Dropout / Linear / ReLU) and
 * nn.CrossEntropyLoss (train.py:277), ATen's dropout / addmm / relu / log_softmax / nll_loss forward and backward.
 * One MnasHeadLinear = the Dropout in front of a Linear + the Linear + the optional ReLU behind it, all fp32:
 *   fwd    y[N][O]  = act( (x * keep/(1-p)) W^T + b )                 act = relu if `relu`
 *   bwd_w  dw[O][I] (+)= dz^T (x * keep/(1-p)),  db[O] (+)= sum_n dz   (+= if `accumulate`)
 *   bwd_x  dx[N][I] = (dz W) * keep/(1-p) * [relu_mask > 0]            relu_mask = x when the layer in FRONT ends in a
 *                                                                      ReLU (then dx is that layer's dz), else NULL
 * keep(seed, n*I+i) is a counter-based hash (splitmix64), never stored: the three calls of one layer and step take the same
 * (drop_p, seed); drop_p == 0 (or eval mode) -> no dropout.  mnas_head_dropout_mask writes the keep bytes of a layer
 * (tests).  Unused pointers may be NULL. */
typedef struct MnasHeadLinear {
    int32_t N, I, O;
    int32_t relu;            /* fwd: ReLU after the Linear */
    int32_t accumulate;      /* bwd_w */
    float   drop_p;          /* dropout probability on the layer's input, 0 <= p < 1 */
    uint64_t seed;
    const void* x;           /* [N][I] layer input (before dropout) */
    const void* w;           /* [O][I] */
    const void* b;           /* [O] or NULL */
    void*       y;           /* [N][O] */
    const void* dz;          /* [N][O] gradient of the Linear's output (after the ReLU mask) */
    void*       dw;          /* [O][I] */
    void*       db;          /* [O] or NULL */
    void*       dx;          /* [N][I] */
    const void* relu_mask;   /* [N][I] or NULL */
} MnasHeadLinear;
int mnas_head_linear_fwd(const MnasHeadLinear* a, void* stream);
int mnas_head_linear_bwd_w(const MnasHeadLinear* a, void* stream);
int mnas_head_linear_bwd_x(const MnasHeadLinear* a, void* stream);
int mnas_head_dropout_mask(void* out_u8, int64_t n, float p, uint64_t seed, void* stream);
/* nn.CrossEntropyLoss(reduction='mean', ignore_index): logits fp32 [N][C], target int64 [N].
 * loss_rows[N] (scratch), *loss = mean over the non-ignored rows, dlogits[N][C] = dloss/dlogits (NULL: forward only).
 * A target outside [0, C) that is not ignore_index sets *bad_flag (int32, device) and poisons the loss with NaN (ATen
 * asserts on the device). */
int mnas_head_cross_entropy(const void* logits, const void* target, int N, int C, int64_t ignore_index,
                            void* loss_rows, void* loss, void* dlogits, void* bad_flag, void* stream);

/* ---- squeeze-and-excitation of the SE variant of MBConv_block (BASELINE config 4; build-defined -- the reference has no SE
 * block; csrc/mnas_se.hip, restated in oracle.se_apply).  a = the activated depthwise output (act-on-load of y2), u = the
 * excite logits fp32 [N][C] (mnas_pool_act -> mnas_head_linear_fwd x2 produce them).
 *   mnas_se_scale      : out = act(a) * sigmoid(u)[n][c]            (bf16 (N,HW,C): what the project conv then reads)
 *   mnas_se_bwd_reduce : du[n][c] = (sum_hw gs * act(a)) * s (1 - s), s = sigmoid(u)   (gs = dL/d out, bf16); scratch =
 *                        mnas_se_scratch_bytes(N, HW, C) bytes of partial sums (pixel splits, added in a fixed order)
 *   mnas_se_bwd_apply  : out = gs * sigmoid(u) + dz[n][c] / HW      (dz = dL/d(pooled a) from the MLP backward; out = dL/d act(a));
 *                        optional fused BatchNorm-backward reduce of the ConvBlock whose activated output `out` is the gradient of:
 *                        red_partial float[2][C][mnas_se_bwd_apply_cols(N,HW,C)] from (out, red_y, red_bn), as mnas_bn_bwd_reduce */
int mnas_se_scale(const MnasActIn* a, const float* u, int N, int HW, int C, void* out_bf16, void* stream);
int mnas_se_bwd_reduce(const void* gs, const MnasActIn* a, const float* u, int N, int HW, int C, float* du, float* scratch,
                       void* stream);
int64_t mnas_se_scratch_bytes(int N, int HW, int C);
int mnas_se_bwd_apply(const void* gs, const float* u, const float* dz, int N, int HW, int C, void* out_bf16,
                      const void* red_y, const float* red_bn, float* red_partial, void* stream);
int mnas_se_bwd_apply_cols(int N, int HW, int C);
/* ABI 5 -- the excitation applied ON LOAD instead of through a materialised a*s (csrc/mnas_se.hip):
 *   mnas_se_gate          : gate[n][c] = sigmoid(u[n][c]), fp32 -- the table MnasConvGemm.gate takes.
 *   mnas_se_proj_finalize : after mnas_pw_bwd of the PROJECT conv run in segment mode (MnasPwBwd.seg_px = HW / kseg, nparts =
 *                           N * kseg) on the UNGATED activation act(a): wpartial float[N*kseg][Co][Ci] holds per-image(-fraction)
 *                           sums of dy[pix][o] * act(a)[pix][c].  Writes du[n][c] = (sum_o W[o][c] * P_n[o][c]) * s (1 - s)
 *                           (= what mnas_se_bwd_reduce computes from gs = dy . W, without the pass over gs and a) and the conv's
 *                           weight gradient dW[o][c] (+)= sum_n s[n][c] * P_n[o][c].  W: the conv's fp32 MASTER weight [Co][Ci]
 *                           (gs itself was formed with the bf16-rounded weight: du differs from mnas_se_bwd_reduce's by that
 *                           rounding, <= 4e-3 relative per term, inside the gradient tolerances of tests/test_gpu_se.py).
 *                           wpartial is overwritten (scratch).  Deterministic (fixed summation order). */
int mnas_se_gate(const float* u, int N, int C, float* gate, void* stream);
int mnas_se_proj_finalize(float* wpartial, int N, int kseg, int Co, int Ci, const float* u, const float* W, float* dW,
                          int accumulate, float* du, void* stream);
/* The excitation MLP in one launch per direction (ABI 7).  z [N][E] fp32 = the pooled activation, W1 [R][E] / b1 [R] = fc1, W2 [E][R] /
 * b2 [E] = fc2 (torch.nn.Linear layouts, 4-byte alignment is enough), shapes: mnas_se_fc_supported:
 *   mnas_se_fc_fwd : hb = relu(W1 z + b1) [N][R];  u = W2 hb + b2 [N][E];  gate = sigmoid(u) [N][E] when gate != NULL
 *                    (replaces two mnas_head_linear_fwd calls and mnas_se_gate)
 *   mnas_se_fc_bwd : from du = dL/du [N][E]:  dh = (du W2) * [hb > 0] [N][R] (caller-provided scratch);  dz = dh W1 [N][E];
 *                    dW2 (+)= du^T hb, db2 (+)= sum_n du, dW1 (+)= dh^T z, db1 (+)= sum_n dh   (accumulate != 0: add to the buffers)
 *                    (replaces two mnas_head_linear_bwd_w and two mnas_head_linear_bwd_x calls; two kernels, fixed summation order) */
int mnas_se_fc_supported(int E, int R);     /* 1: the two calls below take this shape (R <= 48, E <= 1280, 64 KB of LDS); else use mnas_head_linear_* */
int mnas_se_fc_fwd(const float* z, const float* W1, const float* b1, const float* W2, const float* b2, int N, int E, int R,
                   float* hb, float* u, float* gate, void* stream);
int mnas_se_fc_bwd(const float* du, const float* z, const float* hb, const float* W1, const float* W2, int N, int E, int R,
                   float* dh, float* dz, float* dW1, float* db1, float* dW2, float* db2, int accumulate, void* stream);


/* ---- weight packing (fp32 reference layout [Co][Ci/g][kh][kw] -> kernel layouts) -------------------- */
#define MNAS_PACK_FWD   0   /* bf16 [Co_pad16][Kpad32], k = tap*Ci+ci            (mnas_conv_gemm mode 0) */
#define MNAS_PACK_DGRAD 1   /* bf16 [Ci_pad16][Kpad32], k = tap*Co+co            (mnas_conv_gemm mode 1) */
#define MNAS_PACK_DW    2   /* fp32 [k*k][C]                                      (mnas_dw_*)            */
#define MNAS_PACK_TCONV 3   /* bf16 [round16(4*Ci)][round32(4*Co)]: block matrix of a 3x3 stride-2 conv's taps by output parity
                               class (rows) and dy neighbour (columns)            (mnas_tconv_dgrad)     */
int mnas_pack_weights(const float* w, int kind, int Co, int Ci, int kh, int kw, void* dst, void* stream);
/* The same for many tensors in one launch: `descs` is a DEVICE array of n descriptors (taps = kh*kw; for MNAS_PACK_DW
 * Ci is ignored).  Used once per forward for all layers of the network. */
typedef struct MnasPackDesc {
    const float* w;
    void*   dst;
    int32_t kind, Co, Ci, taps;
} MnasPackDesc;
int mnas_pack_weights_batch(const MnasPackDesc* descs, int n, void* stream);
/* sizes in BYTES of the packed buffers */
int64_t mnas_packed_bytes(int kind, int Co, int Ci, int kh, int kw);

/* ---- fused Adam over a flat fp32 parameter/gradient buffer (train.py:219-221: Adam(lr)) -------------- */
int mnas_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* The other two optimizers train.py offers (:222-228), same flat buffers and grad_scale convention:
 * torch.optim.RMSprop (centered = False; momentum_buf may be NULL when momentum == 0) and torch.optim.SGD (step = 1 initialises
 * the momentum buffer with the gradient, as torch does; momentum_buf may be NULL when momentum == 0). */
int mnas_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buf, int64_t n, float lr, float alpha,
                      float eps, float weight_decay, float momentum, float grad_scale, void* stream);
int mnas_sgd_step(float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                  float weight_decay, int nesterov, int step, float grad_scale, void* stream);

/* ---- batched launch: run a pre-built list of the calls above with ONE host->library transition ------- */
#define MNAS_OP_CONV_GEMM 1
#define MNAS_OP_CONV_WGRAD 2
#define MNAS_OP_WGRAD_FINALIZE 3
#define MNAS_OP_DW_FWD 4
#define MNAS_OP_DW_BWD 5
#define MNAS_OP_DW_WGRAD_FINALIZE 6
#define MNAS_OP_STEM_FWD 7
#define MNAS_OP_STEM_WGRAD 8
#define MNAS_OP_BN_FWD_FINALIZE 9
#define MNAS_OP_BN_BWD_REDUCE 10
#define MNAS_OP_BN_BWD_FINALIZE 11
#define MNAS_OP_ADD_ACT 12
#define MNAS_OP_NCHW_TO_NHWC 13
#define MNAS_OP_PACK_WEIGHTS 14
#define MNAS_OP_EVENT_RECORD 15    /* p[0] = event handle from mnas_event_create: hipEventRecord on the op's stream;
                                      p[1] = NULL or a HOST int*: the record is skipped while *p[1] == 0 (measurement gating) */
#define MNAS_OP_EVENT_WAIT 16      /* p[0] = event handle: hipStreamWaitEvent(op's stream, event) */
#define MNAS_OP_PW_BWD 17
#define MNAS_OP_PACK_BATCH 18
#define MNAS_OP_POOL_ACT 22
#define MNAS_OP_POOL_BWD 23
#define MNAS_OP_DY_MAT 24
#define MNAS_OP_BWD_POST 25
#define MNAS_OP_TCONV_DGRAD 26
/* 19-21, 27-29, 36: retired in ABI 6 (Gram-statistics / fused-block / affine-on-read forms; DESIGN_HISTORY.md) */
#define MNAS_OP_HEAD_LINEAR 30  /* i[5]: 0 forward, 1 weight gradient, 2 input gradient */
#define MNAS_OP_SE_SCALE 31
#define MNAS_OP_SE_BWD_REDUCE 32
#define MNAS_OP_SE_BWD_APPLY 33
#define MNAS_OP_SE_GATE 34          /* ABI 5 */
#define MNAS_OP_SE_PROJ_FIN 35      /* ABI 5 */
#define MNAS_OP_SE_FC_FWD 37        /* ABI 7: i = {N, E, R}; p = {z, W1, b1, W2, b2, hb, u, gate or NULL} */
#define MNAS_OP_STEM_DGRAD 39       /* ABI 7: i = {N, H, W, Ho, Wo, Co}; p = {g, y, coef, w fp32, in_affine or NULL, dx} */
#define MNAS_OP_SE_FC_BWD 38        /* ABI 7: i = {N, E, R, accumulate}; p = {du, z, hb, W1, W2, dh, dz, dW1, db1, dW2, db2} */
typedef struct MnasOp {
    int32_t opcode;
    int32_t i[15];
    double  d[4];
    void*   p[16];
} MnasOp;
/* Field use per opcode is documented next to mnas_run_ops in csrc/mnas_abi.hip. Stops at the first error
 * and returns it (index of the failing op in *failed_at if non-NULL). */
int mnas_run_ops(const MnasOp* ops, int n, void* stream, int* failed_at);
/* Same, over several streams: op.i[14] selects streams[op.i[14]] (0 <= i[14] < nstreams).  Ordering between
 * streams is expressed with MNAS_OP_EVENT_RECORD / MNAS_OP_EVENT_WAIT ops.  Used to run the weight-gradient
 * kernels of a layer concurrently with the input-gradient chain (they only share read-only inputs). */
int mnas_run_ops_multi(const MnasOp* ops, int n, void* const* streams, int nstreams, int* failed_at);
/* The same launch list captured once into a hipGraph and replayed (ABI 7).  mnas_graph_create runs the list under stream capture on
 * streams[0] (nothing executes) and instantiates it; mnas_graph_launch replays it on `stream`.  Everything in the list is baked in
 * -- pointers, integers, and whether a gated EVENT_RECORD was live at capture time: re-create when the list changes. */
int mnas_graph_create(const MnasOp* ops, int n, void* const* streams, int nstreams, void** exec_out, int* failed_at);
int mnas_graph_launch(void* exec, void* stream);
int mnas_graph_destroy(void* exec);

/* ---- scratch sizes (bytes) of the partial tables the launches above write; the library never allocates.
 * kind MNAS_WS_CONV_STATS:  float[2][c][n]      (c = Co, n = nparts; also the fused-reduce tables)
 * kind MNAS_WS_CONV_WGRAD:  float[n][c][k]      (n = nsplit, c = Co, k = kh*kw*Ci)
 * kind MNAS_WS_PW_BWD:      float[n][c][k]      (n = nparts, c = Co, k = Ci)
 * kind MNAS_WS_DW_WGRAD:    float[n][k][c]      (n = mnas_dw_rows(...), k = taps, c = C) */
#define MNAS_WS_CONV_STATS 0
#define MNAS_WS_CONV_WGRAD 1
#define MNAS_WS_PW_BWD 2
#define MNAS_WS_DW_WGRAD 3
int64_t mnas_workspace_bytes(int kind, int n, int c, int k);

/* ---- HIP events on the launch stream (measurement only: bench.py brackets single kernel launches inside
 * the timed region; torch.cuda.Event cannot be recorded from inside mnas_run_ops) ------------------------ */
int mnas_event_create(void** event);
int mnas_event_destroy(void* event);
int mnas_event_record(void* event, void* stream);
int mnas_event_elapsed_ms(void* start, void* stop, float* ms);   /* non-zero if either is not complete */

/* ---- box calibration (measurement only, csrc/mnas_probe.hip): what THIS GPU sustains, taken by bench.py right before and after
 * its timed windows -- boxes of one pool differ in shader clock / power state, and a step rate means nothing without them.
 * mnas_probe_copy: dst = src as 16-byte-per-lane loads and stores (bytes % 16 == 0); rate = 2*bytes / time.
 * mnas_probe_valu: blocks x 256 threads, iters x 8 independent v_pk_fma_f32 per lane, nothing else:
 *                  flop = blocks*256*iters*8*4;  TFLOP/s / 65.536 = sustained shader clock in GHz (256 CUs x 4 SIMDs x 16 lanes). */
int mnas_probe_copy(const void* src, void* dst, int64_t bytes, void* stream);
int mnas_probe_valu(float* out, int blocks, int iters, void* stream);
/* round 6 (ABI 8): the probes a kernel of the step may be compared with -- four independent 16-byte nontemporal loads in flight
 * per lane (k_probe_copy keeps one).  mnas_probe_copy4: rate = 2*bytes / time.  mnas_probe_read: read-only stream, `sink`
 * (>= 4 * workgroups bytes) is never written; rate = bytes / time.  Both: bytes % 16384 == 0.  blocks = 0: one 16 KB trip per
 * workgroup up to 65536 workgroups, > 0: that many persistent workgroups. */
int mnas_probe_copy4(const void* src, void* dst, int64_t bytes, int blocks, void* stream);
int mnas_probe_read(const void* src, void* sink, int64_t bytes, int blocks, void* stream);
int mnas_probe_empty(int blocks, int threads, void* stream);      /* a kernel that does nothing: the cost of one launch in a stream */

#ifdef __cplusplus
}
#endif
#endif /* MNAS_H */
