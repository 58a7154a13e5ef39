/*
 * animateportrait_amd.h -- C ABI of the MI355X (gfx950) kernels behind the
 * Module2 generator / discriminator hot path of AnimatePortrait.
 *
 * The reference has no FFI of its own (it is pure Python on stock PyTorch
 * ops); the seam is the set of ATen ops its nn.Modules
 * launch.  Each entry point below names the reference call it stands in for
 * (paths relative to /root/reference/).  The Python host side
 * (animateportrait_amd/_capi.py, ops.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to contiguous fp32 NCHW tensors unless
 *     stated otherwise; the CALLER allocates and owns every buffer, including
 *     packed weights, statistics partials and workspaces;
 *   - every function only enqueues work on `stream` (a hipStream_t passed as
 *     void*); nothing synchronises, nothing allocates;
 *   - return value: 0 = ok, negative = error (message: ap_last_error(),
 *     thread-local); no C++ exception crosses the boundary;
 *   - no global mutable state apart from one-time kernel attribute setup,
 *     so the HOST side of calls on distinct streams is thread-safe;
 *   - but keep all work of this library on ONE stream per device at a time (and
 *     let no other kernel share the device with it): a kernel that shares
 *     compute units with the library's matrix kernels was measured to read
 *     wrong data in lanes 48-63 on MI355X ("Concurrent
 *     streams"; tools/conc_warp.py / tools/hazard/run_hazard.py reproduce it);
 *   - inputs are FINITE fp32 values.  Zero padding and out-of-frame sampling taps
 *     are evaluated as (some in-range element) x (weight 0) and the consumer-side
 *     activations as max(v, slope * v): identical to the reference for finite data,
 *     but a tensor that already holds Inf / NaN (a diverged training run) spreads
 *     it to every padded / out-of-frame position, where ATen would write 0.
 */
#ifndef ANIMATEPORTRAIT_AMD_H
#define ANIMATEPORTRAIT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ap_stream_t; /* hipStream_t */

enum { AP_OK = 0, AP_ERR_INVALID = -1, AP_ERR_UNSUPPORTED = -2, AP_ERR_LAUNCH = -3 };
enum { AP_ACT_NONE = 0, AP_ACT_RELU = 1, AP_ACT_LRELU = 2 /* slope 0.2 */, AP_ACT_TANH = 3 };
enum { AP_PAD_ZERO = 0, AP_PAD_REFLECT = 1 };
enum { AP_W_OIHW = 0 /* nn.Conv2d weight */, AP_W_IOHW = 1 /* nn.ConvTranspose2d weight */ };
enum { AP_PRECISION_FP32 = 0, AP_PRECISION_BF16X3 = 1, AP_PRECISION_BF16 = 2 };

/* One channel segment of a (virtually concatenated) convolution input.  The
 * loader applies, per element, x := act((x - mean[n,c]) * rstd[n,c]) when
 * mean/rstd are given: this is how InstanceNorm2d+ReLU/LeakyReLU of the
 * PRODUCER layer is fused into the CONSUMER convolution, and how torch.cat
 * (Module2/models/networks.py:1330,1335) is executed without a copy. */
typedef struct ap_src {
    const float* data;  /* N x C x H x W */
    const float* mean;  /* N*C, or NULL */
    const float* rstd;  /* N*C, or NULL */
    int32_t C;
    int32_t act;        /* AP_ACT_NONE / RELU / LRELU, applied after the normalisation */
} ap_src;

/* Descriptor of one convolution-like operator.
 *   transposed = 0: nn.Conv2d(Cin, Cout, (KH,KW), stride, pad) with zero or
 *                   reflection padding (nn.ReflectionPad2d fused);
 *   transposed = 1: nn.ConvTranspose2d(Cin, Cout, k, stride=2, pad, output_padding).
 * Cin is the sum of the segments' C.  w_layout / w_flip describe how the
 * operator's taps index the caller's weight tensor, which lets the same kernel
 * run the data-gradient of a convolution (swap roles, flip taps). */
typedef struct ap_conv_desc {
    int32_t N, H, W;      /* input batch / height / width */
    int32_t Cout;
    int32_t KH, KW;
    int32_t stride;       /* 1 or 2 */
    int32_t pad;
    int32_t pad_mode;     /* AP_PAD_* (reflect only for transposed = 0) */
    int32_t transposed;
    int32_t output_padding;
    int32_t w_layout;     /* AP_W_* : memory layout of the weight passed to ap_conv2d_pack_weights */
    int32_t w_flip;       /* 1: use tap (KH-1-ky, KW-1-kx) */
    int32_t act;          /* epilogue activation after bias: AP_ACT_NONE / LRELU / TANH / RELU */
    int32_t nsrc;         /* 1..3 */
    int32_t precision;    /* AP_PRECISION_FP32: exact fp32 MFMA.  AP_PRECISION_BF16X3: the caller accepts fp32-class
                           * (~1e-4 relative) results; wide 3x3 layers then run on the bf16 matrix pipe with operands
                           * split into bf16 head + tail (three MFMAs per tile, fp32 accumulation); other layers are
                           * unaffected.  AP_PRECISION_BF16: plain bf16 arithmetic on the same layers (head parts only:
                           * ONE MFMA per tile, fp32 accumulation, half the LDS image -- two workgroups per CU); the
                           * training configurations (BASELINE configs[2-3], "bf16"): ~4e-3 relative per product, as
                           * torch autocast(bf16) of the reference would give; weights, activations and gradients in
                           * HBM stay fp32 / split tensors (fp32 master weights).  The same value must be used for
                           * pack_weights and fwd. */
    int32_t presplit;     /* 1: src[s].data are split tensors written by ap_split_prepass (mean/rstd/act already applied
                           * there and ignored here).  Required iff ap_conv2d_wants_presplit(d) == 1. */
    int32_t s2d_k;        /* 0, or 3: this 2x2 stride-1 layer over 4 C channels is the space-to-depth form of a 3x3
                           * stride-2 pad-1 layer (networks.py:1221-1228 encoder): tap (ty, tx) of input phase (ry, rx) is
                           * all-zero when 2 ty + ry > 2 or 2 tx + rx > 2, and the split-bf16 kernel skips it */
    ap_src src[3];
} ap_conv_desc;

/* Bumped whenever a signature or a descriptor field of this header changes meaning (round 2 inserted an argument in
 * ap_instnorm_finalize and gave ap_conv_desc.reserved a meaning as s2d_k without one).  A binding compares
 * ap_abi_version() with the AP_ABI_VERSION it was written against at load time and refuses a mismatch
 * (animateportrait_amd/_capi.py does). */
#define AP_ABI_VERSION 13
int32_t ap_abi_version(void);
const char* ap_version(void);
const char* ap_last_error(void);

/* ---- convolution + InstanceNorm2d (+ ReLU, + residual add) in ONE launch (round 4, inference):
 * replaces  x + IN(conv(pad(.)))  /  relu(IN(conv(pad(.))))  of ResnetBlock / ResnetBlock2
 * (Module2/models/networks.py:2329-2360, :2389-2420) for the wide stride-1 3x3 layers.  The workgroups that hold the tiles of
 * one (image, 64-channel tile) exchange per-channel sums through `partials` and `counters` (two rounds: mean, then squares
 * centred on it) while the accumulators stay in registers, so neither the raw convolution output nor a separate
 * normalisation pass touches HBM.  Outputs: the split-bf16 copy `xs` the next convolution stages (ap_split_prepass layout)
 * and / or the fp32 values in the channel-octet layout `y_oct` [N][Cout/8][OH*OW][8] (the residual stream between blocks).
 * Only where ap_conv2d_fused_norm_ok(d) == 1 (split-bf16 arithmetic, 3x3 stride 1, whole 64 x 16 x 32 tiles, and a tile
 * list whose per-image groups fall into one round of the persistent grid -- the waiting workgroups must all be resident:
 * keep the device free of other work, as the single-stream rule above says).  `counters` (ap_conv2d_fused_norm_counters
 * uint32) must be ZERO at launch; its LAST element is an error flag the kernel sets (and the results are then invalid) when a
 * workgroup waited ~0.2 s in vain for its group; `partials` has ap_conv2d_stat_tiles(d) float pairs per (n, c). */
typedef struct ap_fused_norm {
    int32_t act;            /* AP_ACT_NONE / RELU / LRELU applied after the normalisation */
    float eps;              /* nn.InstanceNorm2d eps (1e-5) */
    const float* res_oct;   /* residual added after normalisation + activation: channel-octet fp32, or */
    const float* res_nchw;  /* ... NCHW fp32, or neither (both NULL) */
    float* y_oct;           /* fp32 result, channel-octet layout, or NULL */
    void* xs;               /* split-bf16 copy of the result, or NULL */
    float* mean;            /* [N * Cout] finished statistics (written) */
    float* rstd;
    float* partials;        /* [N * Cout][stat_tiles][2] exchange buffer */
    uint32_t* counters;     /* zero-initialised by the caller */
} ap_fused_norm;
int32_t ap_conv2d_fused_norm_ok(const ap_conv_desc* d);
int32_t ap_conv2d_fused_norm_counters(const ap_conv_desc* d);
int ap_conv2d_fwd_norm(const ap_conv_desc* d, const float* packed, const ap_fused_norm* fn, ap_stream_t stream);

/* ---- convolution: nn.Conv2d / nn.ConvTranspose2d (+ fused pad, bias, act, IN statistics)
 * replaces F.conv2d / F.conv_transpose2d / F.pad(reflect) launched by
 *   Module2/models/networks.py:1218-1282 (generator layers),
 *   :2329-2361, :2390-2421 (ResnetBlock, ResnetBlock2),
 *   :2620-2643 (NLayerDiscriminator). */
int ap_conv2d_out_size(const ap_conv_desc* d, int32_t* Hout, int32_t* Wout);
/* number of floats of the packed-weight buffer the caller must provide */
int64_t ap_conv2d_packed_floats(const ap_conv_desc* d);
/* number of partial (sum, sumsq) tiles per (n, cout) written by ap_conv2d_fwd: stats buffer
 * is N * Cout * tiles * 2 floats */
int32_t ap_conv2d_stat_tiles(const ap_conv_desc* d);
/* re-lay the weight (layout d->w_layout, Cout/Cin/KH/KW from d) for the kernel's LDS image */
int ap_conv2d_pack_weights(const ap_conv_desc* d, const float* weight, float* packed, ap_stream_t stream);

/* ---- batched packing (training: every optimiser step invalidates every packed image of a network).
 * A weight operand as a strided / derived view of a parameter, in OPERATOR terms: element (cout, cin, ky, kx) of the
 * operator described by the conv descriptor sits at w[cout * s_co + cin * s_ci + ky * s_ky + kx * s_kx] -- a channel slice
 * or a transposed-tap view of the layer's parameter for its data-gradient operators needs no contiguous copy.
 *   s2d_c  > 0: the operator is the 2x2 space-to-depth form (4 * s2d_c input channels) of a ksrc x ksrc stride-2 layer whose
 *               (cout, c, ky, kx) element the strides address;
 *   rows_c > 0: the operator is the 1xK row form (ap_split_prepass_rows) of a ksrc x ksrc stem over rows_c channels. */
typedef struct ap_weight_view {
    const float* w;
    int64_t s_co, s_ci, s_ky, s_kx;
    int32_t s2d_c, rows_c, ksrc, reserved;
} ap_weight_view;
/* bytes of one packer entry */
int32_t ap_conv2d_pack_entry_bytes(void);
/* Fill `entries` (HOST memory, max_entries * ap_conv2d_pack_entry_bytes()) with the packer entries of this operator: what
 * ap_conv2d_pack_weights would launch, with `v` as the source and `packed` as the destination.  Returns the number of entries
 * written, 0 when the plan is not a split-bf16 plan (then use ap_conv2d_pack_weights), negative on error.  The caller
 * concatenates the entries of all layers of a network, uploads the table once and calls ap_conv2d_pack_run after every
 * optimiser step: one launch instead of one per (layer, operator). */
int32_t ap_conv2d_pack_entries(const ap_conv_desc* d, const ap_weight_view* v, float* packed, void* entries, int32_t max_entries);
/* run `count` entries of a DEVICE-resident table */
int ap_conv2d_pack_run(const void* entries_dev, int32_t count, ap_stream_t stream);
/* y = act(conv(src...) + bias); bias may be NULL; stat_partials may be NULL.
 * When stat_partials != NULL the epilogue also writes per-tile sum / sum of squares of the
 * pre-activation output for every (n, cout) (InstanceNorm2d statistics, networks.py:33-34). */
int ap_conv2d_fwd(const ap_conv_desc* d, const float* packed, const float* bias, float* y,
                  float* stat_partials, ap_stream_t stream);
/* ap_conv2d_fwd into a window: only output pixels oy < OH, ox < OW are computed, and y[n][co][oy][ox] is stored at
 *     y + n*nstride + co*cstride + (oy + y_off)*rstride + (ox + x_off)*xstride          (elements).
 * Split-bf16 single-launch plans only (stride-1 / stride-2 convolutions, not transposed), no statistics epilogue.
 * Used for the data gradient of the reflection-padded 3x3 layers (networks.py:2329-2421), whose padded-coordinate
 * output is 66 = 2 x 32 + 2 columns wide: the main launch computes columns 0..63 (two tile columns instead of
 * three), a second launch on the transposed last gradient columns writes the 2-column rest (rstride = 1,
 * xstride = row length). */
typedef struct ap_out_view {
    int64_t nstride, cstride;
    int32_t rstride, xstride, y_off, x_off, OH, OW;
} ap_out_view;
int ap_conv2d_fwd_view(const ap_conv_desc* d, const ap_out_view* view, const float* packed, const float* bias, float* y,
                       ap_stream_t stream);
/* ap_conv2d_fwd with the output in the CHANNEL-OCTET layout y[n][Cout/8][Hout*Wout][8] (8 consecutive channels of a pixel
 * are 32 contiguous bytes) -- what the warp kernel gathers best (ap_warp_concat_fwd_ex flags bit 1): the three encoder
 * layers in front of double_feature_warping (networks.py:1317-1328: model_tri00, model_tri11, model_tri22) write it in
 * inference, so that a bilinear tap of 8 channels is one 32-byte read instead of 8 reads from 8 planes.  Statistics as
 * ap_conv2d_fwd (finalise with ap_instnorm_finalize_octet).  Only where ap_conv2d_octet_ok(d) == 1: single-launch
 * split-bf16 layers of the run-time-tap (2x2 space-to-depth) and row (7x7 stem) kernel families and (round 6) the dense 3x3
 * stride-1 kernel in AP_PRECISION_BF16X3 -- the ResNet trunk's convolutions in inference (networks.py:2329-2361), whose raw output
 * is then read by ap_norm_apply_split_ex with flags bit 4; Cout % 8 == 0. */
int32_t ap_conv2d_octet_ok(const ap_conv_desc* d);
int ap_conv2d_fwd_octet(const ap_conv_desc* d, const float* packed, const float* bias, float* y, float* stat_partials,
                        ap_stream_t stream);
/* ap_conv2d_fwd / ap_conv2d_fwd_view with the OUTPUT stored as bf16 (round to nearest even; InstanceNorm partial sums still from the
 * fp32 accumulators): the plain-bf16 train step (precision = AP_PRECISION_BF16, BASELINE configs[2-3]) stores the raw outputs of the
 * ResNet trunk's convolutions (networks.py:2329-2421) and the gradients that leave their data-gradient convolutions this way --
 * every reader of those tensors rounds them to bf16 anyway.  Only where ap_conv2d_bf16out_ok(d) == 1 (dense 3x3 stride-1 layers on
 * the bf16 matrix path).  Readers that take such a tensor: ap_norm_apply_split_ex (flags bit 3), ap_instnorm_bwd_split (heads_only
 * bit 1: y, bit 2: g1), the padded-row operand pass of ap_conv2d_wgrad (ap_src.act bit 8).  Element offsets / strides are those of
 * the fp32 forms (in elements). */
int32_t ap_conv2d_bf16out_ok(const ap_conv_desc* d);
int ap_conv2d_fwd_bf16out(const ap_conv_desc* d, const float* packed, const float* bias, void* y_bf16, float* stat_partials,
                          ap_stream_t stream);
int ap_conv2d_fwd_view_bf16out(const ap_conv_desc* d, const ap_out_view* view, const float* packed, const float* bias, void* y_bf16,
                               ap_stream_t stream);

/* Split-bf16 path (precision = AP_PRECISION_BF16X3, wide 3x3 layers): the convolution consumes its sources as
 * split tensors XS[n][head|tail][C/8][H*W + 1][8 x bf16] (the last 16-byte slot of every plane is all-zero and
 * feeds the zero-padding taps).  ap_split_prepass applies the producer's
 * InstanceNorm + activation (src->mean/rstd/act) once and writes that tensor; it can be shared by every consumer of
 * the same activation.  `out` must hold ap_split_prepass_bytes() bytes, 16-byte aligned. */
int32_t ap_conv2d_wants_presplit(const ap_conv_desc* d);
int64_t ap_split_prepass_bytes(int32_t N, int32_t C, int32_t H, int32_t W);
int ap_split_prepass(const ap_src* src, int32_t N, int32_t H, int32_t W, void* out, ap_stream_t stream);
/* 7x7 stems with 1..4 input channels (networks.py:1218, 1231, 1244: ReflectionPad2d(3) + Conv2d(input_nc, ., 7)) on
 * the split-bf16 path: the input is expanded to 32 "row channels" R[ky*C + c][y][x] = src[c][y + ky - pad][x]
 * (vertical padding applied here; `out`: ap_split_prepass_bytes(N, 32, H, W) bytes), and the stem becomes a 1 x 7
 * convolution over them: ap_conv2d_* with KH = 1, KW = 7, pad 3, one 32-channel source, presplit = 1 and weights
 * W'[co][ky*C + c][0][kx] = W[co][c][ky][kx] (zero for the unused channels).  One expansion serves all three stems. */
int ap_split_prepass_rows(const ap_src* src, int32_t N, int32_t H, int32_t W, int32_t K, int32_t pad, int32_t pad_mode,
                          void* out, ap_stream_t stream);
/* 4x4 stride-2 pad-1 layers (PatchGAN body, networks.py:2620-2636) on the split-bf16 path, as space-to-depth:
 *     X'[(ry*2 + rx)*C + c][qy][qx] = pad1(act(IN(src)))[c][2 qy + ry][2 qx + rx]      ((H/2 + 1) x (W/2 + 1), 4C channels)
 * (`out`: ap_split_prepass_bytes(N, 4C, H/2 + 1, W/2 + 1) bytes; C % 8 == 0, H and W even), and the layer becomes
 * ap_conv2d_* with KH = KW = 2, stride 1, pad 0, one 4C-channel source, presplit = 1 and weights
 * W'[co][(ry*2 + rx)*C + c][ty][tx] = W[co][c][2 ty + ry][2 tx + rx]. */
int ap_split_prepass_s2d(const ap_src* src, int32_t N, int32_t H, int32_t W, void* out, ap_stream_t stream);
/* The whole step between two convolutions in one streaming pass (ResnetBlock: networks.py:2329-2360):
 *     v = act(IN(src)) [+ IN(residual)]      y = v as fp32 (NULL: skip)      xs = split-bf16 copy of v (NULL: skip)
 * The InstanceNorm statistics of `src` come finished (src->mean / rstd) or -- stat_partials != NULL, src->mean NULL --
 * as the producing convolution's partial tiles (ap_conv2d_fwd), which this call finalises exactly as
 * ap_instnorm_finalize does and stores to mean_out / rstd_out ([N*C]) for later consumers.  `residual` (nullable)
 * may carry finished statistics but no activation.  With y == xs == NULL it is a plain finalize. */
int ap_norm_apply_split(const ap_src* src, const float* stat_partials, int32_t tiles, float eps, float* mean_out,
                        float* rstd_out, const ap_src* residual, int32_t N, int32_t H, int32_t W, float* y, void* xs,
                        ap_stream_t stream);
/* ... with flags: bit 0 = write the head planes of xs only (every consumer runs in AP_PRECISION_BF16 and never reads the
 * tails: a quarter of the pass's HBM traffic); bit 1 = xs holds relu(v) while y holds v (the `activation -> conv` input of the
 * next layer of a pre-activation residual stream: ResidualBlock of intrinsic_flow_models/networks.py:26-60); bit 2 =
 * residual->data is the residual's SPLIT COPY (head + tail planes, ap_split_prepass layout; mean / rstd must be NULL): in
 * inference the residual stream of the ResNet trunk (x + conv_block(x), networks.py:2358-2360) then lives only in the form the
 * next convolution stages, and the fp32 output y is not needed (pass NULL); bit 3 = src->data holds bf16 values
 * (ap_conv2d_fwd_bf16out); bit 4 = src->data is the CHANNEL-OCTET raw output of ap_conv2d_fwd_octet ([N][C/8][H*W][8] fp32): the
 * only output is then the split copy (y must be NULL), the residual absent or a split copy (bit 2) */
int ap_norm_apply_split_ex(const ap_src* src, const float* stat_partials, int32_t tiles, float eps, float* mean_out,
                           float* rstd_out, const ap_src* residual, int32_t N, int32_t H, int32_t W, float* y, void* xs,
                           int32_t flags, ap_stream_t stream);

/* name of the conv_igemm_f32 instantiation the plan selects for `d` (as it appears, demangled, in a
 * rocprofv3 kernel trace), e.g. "ConvCfg<4, 1, 3, 2, 2, 2, 2>"; used by bench.py to attribute time per kernel */
int ap_conv2d_kernel_name(const ap_conv_desc* d, char* buf, int32_t buflen);

/* ---- InstanceNorm2d(affine=False, eps) : networks.py:33-34 (F.instance_norm)
 * finalize: partial tiles -> mean[n,c], rstd[n,c] (biased variance); count = Hout*Wout.  y (nullable): the convolution
 * output the tiles were summed over ([NC][count]); when given, planes whose |mean| is many standard deviations -- where
 * E[x^2] - E[x]^2 of fp32 sums is rounding noise -- are recomputed from it with the shifted two-pass formula. */
int ap_instnorm_finalize(const float* stat_partials, const float* y, int32_t NC, int32_t tiles, int32_t count,
                         float eps, float* mean, float* rstd, ap_stream_t stream);
/* ... for y in the channel-octet layout of ap_conv2d_fwd_octet ([N][C/8][count][8]); mean / rstd stay [N*C]. */
int ap_instnorm_finalize_octet(const float* stat_partials, const float* y, int32_t N, int32_t C, int32_t tiles, int32_t count,
                               float eps, float* mean, float* rstd, ap_stream_t stream);
/* out = act((x - mean) * rstd) + residual, where residual is
 *   NULL, a plain tensor (res_mean == NULL), or itself normalised: (res - res_mean) * res_rstd.
 * Covers `x + conv_block(x)` (networks.py:2358-2360) and `shortcut(x) + conv_block(x)` (:2418-2420).
 * HW = H*W, NC = N*C. */
int ap_instnorm_apply(const float* x, const float* mean, const float* rstd, int32_t act,
                      const float* res, const float* res_mean, const float* res_rstd,
                      float* out, int32_t NC, int32_t HW, ap_stream_t stream);

/* ---- double_feature_warping : networks.py:1298-1313 + intrinsic_flow_models/modules.py:596-625
 * out[:, 0:C]  = grid_sample(x, motion_L)                         (bilinear, zeros, align_corners=False)
 * out[:, C:2C] = where(mask_L > 0.5, grid_sample(x, grid(flow_L)), -1)
 * with motion_L / flow_L / mask_L the align_corners=True bilinear resizes of the full-resolution
 * motion (N,S,S,2) / flow/2^level (N,2,S,S) / ifmask (N,1,S,S) to x's H x W, computed on the fly.
 * x may be a raw conv output with (mean, rstd, act) applied per tap (act(IN(x)) is what is sampled). */
int ap_warp_concat_fwd(const float* x, const float* x_mean, const float* x_rstd, int32_t x_act,
                       const float* motion, const float* flow, const float* ifmask,
                       float* out, int32_t N, int32_t C, int32_t H, int32_t W, int32_t S,
                       float flow_scale, ap_stream_t stream);
/* Motion grid of the data layer (cal_motion256: Module2/data/umlvd_ifw_dataset.py:60-74, umlvdfw_test_dataset.py:67-81 =
 * scipy.interpolate.griddata(destination, source, 256x256 grid, method='linear')): rasterises the piecewise-linear map
 * defined by a triangulation.  pts / val: [N][P][2] destination / source points as (row, col); tri: [N][T][3] indices
 * into the P points (Delaunay simplices from the host; -1 entries are ignored); out: [N][S][S][2] in grid_sample
 * coordinates ((col, row) / ((S-1)/2) - 1), i.e. the `warp_motion` input of the generator. */
int ap_motion_grid(const float* pts, const float* val, const int32_t* tri, int32_t N, int32_t P, int32_t T, int32_t S,
                   float* out, ap_stream_t stream);
/* Image-level helpers of the streaming-inference model (geomcgt_ifw_test_model.py:282-285, 294):
 * y = F.interpolate(x, (OH, OW), mode='bilinear', align_corners=False) over NC planes, and
 * y = F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=...) with grid (N, OH, OW, 2). */
int ap_resize_bilinear(const float* x, int32_t NC, int32_t H, int32_t W, int32_t OH, int32_t OW, float* y,
                       ap_stream_t stream);
int ap_grid_sample(const float* x, const float* grid, int32_t N, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                   int32_t align_corners, float* y, ap_stream_t stream);
/* Same operator with both outputs optional: out (fp32 [N, 2C, H, W]) and / or xs, the split-bf16 copy of that
 * tensor (ap_split_prepass layout, ap_split_prepass_bytes(N, 2C, H, W) bytes; needs C % 8 == 0) that the next
 * split-bf16 convolution stages -- an inference pass then never writes or re-reads the fp32 concat. */
int ap_warp_concat_fwd_split(const float* x, const float* x_mean, const float* x_rstd, int32_t x_act,
                             const float* motion, const float* flow, const float* ifmask,
                             float* out, void* xs, int32_t N, int32_t C, int32_t H, int32_t W, int32_t S,
                             float flow_scale, ap_stream_t stream);
/* ... with flags.  bit 0: xs takes the SPACE-TO-DEPTH split layout of the 2C-channel concat -- what ap_split_prepass_s2d
 * would make of it: (N, 4 * 2C, H/2 + 1, W/2 + 1) with the zero padding ring written -- for a stride-2 3x3 consumer
 * (model_tri01 / model_tri12, networks.py:1318-1324) that runs as a 2x2 stride-1 layer (ap_conv_desc.s2d_k = 3); H, W even.
 * bit 1: x is in the channel-octet layout [N][C/8][H*W][8] written by ap_conv2d_fwd_octet (C % 8 == 0); x_mean / x_rstd
 * stay [N*C]. */
int ap_warp_concat_fwd_ex(const float* x, const float* x_mean, const float* x_rstd, int32_t x_act,
                          const float* motion, const float* flow, const float* ifmask, float* out, void* xs,
                          int32_t N, int32_t C, int32_t H, int32_t W, int32_t S, float flow_scale, int32_t flags,
                          ap_stream_t stream);

/* ======================================================================= backward pass
 * What torch.autograd launches for the layers above (loss.backward() at
 * Module2/models/geomgm_ifw_fore_model.py:586,610,634,780).
 *
 * Data gradient of a convolution = ap_conv2d_fwd with a descriptor whose roles are swapped
 * (w_layout / w_flip / transposed), so it has no entry point of its own:
 *   Conv2d(s=1, pad p, k)          -> conv with IOHW-read weights, w_flip=1, zero pad k-1 (reflect-padded
 *                                     layers: gradient in PADDED coordinates, folded by the readers below)
 *   Conv2d(s=2, pad p, k)          -> transposed conv (output_padding chosen to restore the input size)
 *   ConvTranspose2d(s=2)           -> stride-2 conv with OIHW-read weights
 */

/* Weight gradient.  dW[m][ci*K*K + ky*K + kx] = sum_{n,oy,ox} G[n,m,oy,ox] * A[n,ci,oy*s+ky-pad,ox*s+kx-pad].
 *   nn.Conv2d:          g = gradient w.r.t. the conv output (M = Cout, grid GHxGW = output size), src = the
 *                       conv's input segments (virtual inputs allowed) -> dW is the OIHW tensor;
 *   nn.ConvTranspose2d: g = the layer's INPUT (M = Cin, virtual allowed), src = gradient w.r.t. its output,
 *                       stride 2 -> dW is the IOHW tensor. */
typedef struct ap_wgrad_desc {
    int32_t N, M;
    int32_t GH, GW;       /* spatial size of g (the grid that is iterated) */
    int32_t H, W;         /* spatial size of the src tensors */
    int32_t K, stride, pad, pad_mode;
    int32_t nsrc;
    int32_t precision;    /* AP_PRECISION_*: BF16X3 lets wide stride-1 3x3 / 4x4 layers run on the bf16 matrix pipe
                             with split operands (fp32-class results, ~4x the exact-fp32 MFMA rate); BF16: the head x head
                             product only (plain bf16 operands, fp32 accumulation) */
    ap_src g;             /* g.C is ignored (M is used) */
    ap_src src[3];
    /* Optional (ABI 10): the split-bf16 copies the FORWARD pass staged of the same sources, still alive in the backward pass.
     * When every segment has one (channels a multiple of 8) and the layer is on the bf16 matrix plan, the shifted operand is
     * re-tiled from them (16-byte slot transposition, no second normalisation pass over src[].data, which is then not read):
     *   src_xs[i]   ap_split_prepass layout of segment i (stride-1 3x3 / 4x4 layers);
     *   src_xs_s2d  ap_split_prepass_s2d layout of the single source (stride-2 layers in their space-to-depth form; without it
 *               the view is gathered from src_xs[0]);
     *   xs_parts    planes those copies hold: 2 = head + tail, 1 = heads only (copies written in AP_PRECISION_BF16 mode:
     *               usable by a BF16 weight gradient only).  All null / 0: the operand is prepared from src[].data. */
    const void* src_xs[3];
    const void* src_xs_s2d;
    int32_t xs_parts;
} ap_wgrad_desc;
/* Weight gradient of a one-output-channel 4x4 stride-1 layer on a small map (the PatchGAN output layer,
 * networks.py:2643): dw[0][c][ky][kx] = sum_{n,oy,ox} g[n,0,oy,ox] * act(IN(src))[n,c,oy+ky-pad,ox+kx-pad].  One workgroup
 * per input channel, fixed summation order, no workspace.  (Its forward is a plan of ap_conv2d_fwd.) */
int ap_conv_head_wgrad(const ap_src* src, const float* g, int32_t N, int32_t H, int32_t W, int32_t K, int32_t pad,
                       float* dw, ap_stream_t stream);
/* Weight gradient of a one-output-channel 7x7 pad-3 layer (the generator's last layer, networks.py:1277-1279):
 * dw[0][c][ky][kx] = sum_{n,y,x} g[n,0,y,x] * pad3(act(IN(src)))[n,c,y+ky,x+kx]  (reflection or zero padding), on the vector
 * ALUs with the input window staged through LDS; workspace: ap_conv_final_wgrad_workspace_floats() floats. */
int64_t ap_conv_final_wgrad_workspace_floats(int32_t N, int32_t C, int32_t H, int32_t W);
int ap_conv_final_wgrad(const ap_src* src, const float* g, int32_t N, int32_t H, int32_t W, int32_t K, int32_t pad,
                        int32_t pad_mode, float* workspace, float* dw, ap_stream_t stream);
/* Weight gradients of the 7x7 pad-3 (reflection) edge layers at full resolution on the bf16 matrix pipe, plain-bf16 arithmetic
 * (operands rounded to bf16, fp32 accumulation; csrc/wgrad_k7.h).  A WIDE tensor (C = 32 or 64 channels, read once) against the 49
 * shifted views of a NARROW one:
 *   final_form = 0 (the stems, networks.py:1251-1260): wide = the layer's output gradient [N][M][H][W] (plain; fp32, or bf16
 *       values with wide->act bit 8 set: what ap_instnorm_bwd stores on request), narrow = its
 *       input [N][C][H][W] (C = 1 or 3, plain):  dw[m][c][ky][kx] = sum g[n,m,y,x] * reflpad3(in)[n,c,y+ky,x+kx];
 *   final_form = 1 (the last layer, networks.py:1277-1279; replaces ap_conv_final_wgrad in this arithmetic): wide = the layer's
 *       input (InstanceNorm + activation of wide->mean / rstd / act applied on the fly), narrow = the one-channel output gradient:
 *       dw[0][c][ky][kx] = sum g[n,0,y,x] * reflpad3(act(IN(src)))[n,c,y+ky,x+kx].
 * Served shapes: ap_wgrad_k7_bf16_ok() == 1 (W a multiple of 16 in 16..256, H >= 4); everything else AP_ERR_UNSUPPORTED.
 * workspace: ap_wgrad_k7_bf16_workspace_floats() floats.  Fixed summation order. */
int32_t ap_wgrad_k7_bf16_ok(int32_t N, int32_t wide_C, int32_t narrow_C, int32_t H, int32_t W, int32_t final_form);
int64_t ap_wgrad_k7_bf16_workspace_floats(int32_t N, int32_t wide_C, int32_t narrow_C, int32_t H, int32_t W, int32_t final_form);
int ap_wgrad_k7_bf16(const ap_src* wide, const ap_src* narrow, int32_t N, int32_t H, int32_t W, int32_t final_form,
                     float* workspace, float* dw, ap_stream_t stream);
/* Data gradient of the same last layer (Conv2d(C, 1, 7) behind ReflectionPad2d(3), C = 32 or 64) in plain-bf16 arithmetic on the bf16
 * matrix pipe (csrc/dgrad_k7.h): the gradient w.r.t. the PADDED input,
 *   gp[n][c][py][px] = sum_{ky,kx} w[0][c][ky][kx] * g[n][0][py-ky][px-kx],  py < H+6, px < W+6 (g zero outside),
 * to be folded over the reflection by its consumer (ap_instnorm_bwd / ap_fold_add with g1_pad = 3).  w: the layer's weight
 * [1][C][7][7].  Served: ap_conv_final_dgrad_bf16_ok() == 1 (W a multiple of 16 in 16..256). */
int32_t ap_conv_final_dgrad_bf16_ok(int32_t N, int32_t C, int32_t H, int32_t W);
int64_t ap_conv_final_dgrad_bf16_workspace_floats(int32_t N, int32_t C, int32_t H, int32_t W);
int ap_conv_final_dgrad_bf16(const float* g, const float* w, int32_t N, int32_t C, int32_t H, int32_t W, float* workspace, float* gp,
                             ap_stream_t stream);
/* Data gradient of the PatchGAN's output layer Conv2d(C, 1, 4, stride 1, pad 1) (networks.py:2643) in plain-bf16 arithmetic on the
 * bf16 matrix pipe (csrc/dgrad_k7.h):  gx[n][c][iy][ix] = sum_{ky,kx} w[0][c][ky][kx] * g[n][0][iy-ky+1][ix-kx+1]  (g: [N][1][H-1][W-1],
 * zero outside; w: the layer's weight [1][C][4][4]; gx: [N][C][H][W]).  Served: ap_conv_head_dgrad_bf16_ok() == 1 (C a multiple of 32,
 * H W <= 1156). */
int32_t ap_conv_head_dgrad_bf16_ok(int32_t N, int32_t C, int32_t H, int32_t W);
int ap_conv_head_dgrad_bf16(const float* g, const float* w, int32_t N, int32_t C, int32_t H, int32_t W, float* gx, ap_stream_t stream);
/* The PatchGAN's first layer, y = act(Conv2d(Cin = 1 | 2, 64, 4, stride 2, pad 1)(x) + bias) (networks.py:2620-2623), in plain-bf16
 * arithmetic on the bf16 matrix pipe as an output stream (csrc/conv_d0.h).  x: plain [N][Cin][H][W], w: the layer's weight
 * [64][Cin][4][4], bias [64] or NULL, act AP_ACT_NONE / RELU / LRELU, y [N][64][H/2][W/2].
 * Served: ap_conv_d0_fwd_bf16_ok() == 1 (even H, W a multiple of 4 up to 256). */
int32_t ap_conv_d0_fwd_bf16_ok(int32_t N, int32_t Cin, int32_t Cout, int32_t H, int32_t W);
int ap_conv_d0_fwd_bf16(const float* x, const float* w, const float* bias, int32_t N, int32_t Cin, int32_t Cout, int32_t H, int32_t W,
                        int32_t act, float* y, ap_stream_t stream);
/* ... and its weight gradient, dw[m][c][ky][kx] = sum g[n,m,oy,ox] * zeropad1(x)[n,c,2oy+ky,2ox+kx] (g: [N][64][H/2][W/2] plain, x: the
 * layer's input), on the same kernel as ap_wgrad_k7_bf16 (form 2 of csrc/wgrad_k7.h: the gradient is read once).  Served:
 * ap_wgrad_d0_bf16_ok() == 1 (M = 64, Cin 1 | 2, even H, W a multiple of 32 up to 512 for one input channel, up to 448 for two). */
int32_t ap_wgrad_d0_bf16_ok(int32_t N, int32_t M, int32_t Cin, int32_t H, int32_t W);
int64_t ap_wgrad_d0_bf16_workspace_floats(int32_t N, int32_t M, int32_t Cin, int32_t H, int32_t W);
int ap_wgrad_d0_bf16(const float* g, const float* x, int32_t N, int32_t M, int32_t Cin, int32_t H, int32_t W, float* workspace, float* dw,
                     ap_stream_t stream);
/* workspace = padded copies of the operands (normalisation / activation / concat / padding applied once, streaming)
 * + per-split partial sums */
int64_t ap_conv2d_wgrad_workspace_floats(const ap_wgrad_desc* d);
/* out[n][c][y][x], y < Hp, x < Wp: the padded view of act(InstanceNorm(concat(src))) -- (y,x) maps to input
 * (y-pad, x-pad) with zero or reflection padding (F.pad / nn.ReflectionPad2d), zeros beyond H+2pad / W+2pad */
int ap_pad_materialize(const ap_src* src, int32_t nsrc, int32_t N, int32_t H, int32_t W, int32_t pad, int32_t pad_mode,
                       int32_t Hp, int32_t Wp, float* out, ap_stream_t stream);
int ap_conv2d_wgrad(const ap_wgrad_desc* d, float* workspace, float* dw, ap_stream_t stream);

/* backward of a = act(InstanceNorm(y)) w.r.t. y.  The incoming gradient is fold(g1) + g2 where g1 has spatial
 * (H+2*g1_pad) x (W+2*g1_pad) (gradient of a reflection-padded consumer, nn.ReflectionPad2d backward fused) and g2
 * (optional) is plain.  sums_ws: N*C*2 floats of scratch.
 * act bit 8 (0x100): dy is STORED as bf16 values (N*C*H*W of them) -- for a gradient whose one reader is ap_wgrad_k7_bf16's stem
 * form; served where the big-plane kernel runs (g1_pad == 0, 16384 < H*W <= 65536, W % 4 == 0), AP_ERR_UNSUPPORTED elsewhere. */
int ap_instnorm_bwd(const float* g1, int32_t g1_pad, const float* g2, const float* y, const float* mean,
                    const float* rstd, int32_t act, int32_t NC, int32_t H, int32_t W, float* sums_ws, float* dy,
                    ap_stream_t stream);
/* The same backward for a layer whose gradient is only ever read by the bf16 matrix kernels (round 4): instead of the fp32 dy the
 * kernel writes the operands those kernels stage -- what ap_split_prepass and the weight gradient's own transposition would make
 * of dy in two more passes over it (autograd of networks.py:2329-2421 through nn.InstanceNorm2d / nn.ReLU):
 *   xs    the split-bf16 copy (ap_split_prepass layout and size; what the data-gradient ap_conv2d_fwd takes with desc.presplit), or NULL;
 *   gt    the M-role operand of ap_conv2d_wgrad_pre, gt_dims = {GHp, GX8, Mp} from ap_conv2d_wgrad_gt_dims; slots outside the
 *         H x W x C gradient must be zero on entry (they are not written), or NULL;
 *   strip fp32 [N][C][2][H]: dy[:, :, :, W-2:] transposed (the small second operand of a reflection-padded layer's data gradient), or NULL;
 *   dy    the fp32 gradient itself, or NULL.
 * heads_only: the consumers multiply head parts only (AP_PRECISION_BF16): tail planes are not written.
 * ap_instnorm_bwd_split_ok: 1 when the shape is served (C % 8 == 0, W % 8 == 0, H >= 3, H * W <= 4096, fold 0 or 1). */
int ap_instnorm_bwd_split_ok(int32_t C, int32_t H, int32_t W, int32_t g1_pad);
int ap_instnorm_bwd_split(const float* g1, int32_t g1_pad, const float* g2, const float* y, const float* mean, const float* rstd,
                          int32_t act, int32_t N, int32_t C, int32_t H, int32_t W, void* xs, void* gt, const int32_t* gt_dims,
                          float* strip, float* dy, int32_t heads_only, ap_stream_t stream);
/* ap_conv2d_wgrad with its M-role operand g prepared by the producer (ap_instnorm_bwd_split): ap_conv2d_wgrad_gt_dims returns 1
 * and the operand's {rows, pixel octets per row, padded channels} when the descriptor runs on the bf16 matrix kernel (the
 * operand is N * 2 * rows * octets * channels 16-byte slots), 0 when it does not (then only ap_conv2d_wgrad serves it);
 * d->g.data is not read by ap_conv2d_wgrad_pre. */
int ap_conv2d_wgrad_gt_dims(const ap_wgrad_desc* d, int32_t* dims);
int ap_conv2d_wgrad_pre(const ap_wgrad_desc* d, const void* g_t, float* workspace, float* dw, ap_stream_t stream);
/* ap_conv2d_wgrad with BOTH operands taken as the split copies the convolutions stage (ABI 11; stride-1 3x3 layers on the bf16 matrix
 * plan): the shifted operand from d->src_xs[] (the forward pass's copies, see ap_wgrad_desc), the M-role operand from g_xs = the
 * split copy of dy in the ap_split_prepass layout (what ap_instnorm_bwd_split writes as `xs` for the data-gradient convolution).
 * No operand is prepared: the kernel reads 8-channel slots with transposing LDS reads.  d->g / d->src[].data are not read.
 * ap_conv2d_wgrad_xs_ok: 1 when the descriptor (with its src_xs / xs_parts) is served.  Workspace: ap_conv2d_wgrad_workspace_floats. */
int32_t ap_conv2d_wgrad_xs_ok(const ap_wgrad_desc* d);
int ap_conv2d_wgrad_xs(const ap_wgrad_desc* d, const void* g_xs, float* workspace, float* dw, ap_stream_t stream);
/* dy = (fold(g1) + g2) * act'(out) for layers without normalisation; act = NONE makes it a fold-and-add
 * (out may then be NULL).  AP_ACT_TANH: 1 - out^2; RELU / LRELU: from the sign of the activated output. */
int ap_act_bwd(const float* g1, int32_t g1_pad, const float* g2, const float* out, int32_t act, int32_t NC,
               int32_t H, int32_t W, float* dy, ap_stream_t stream);
/* ap_act_bwd of a layer WITHOUT InstanceNorm together with its bias gradient db[c] = sum_{n,y,x} dy[n][c][y][x] (fixed summation order):
 * the block sums of dy are formed while it is written, no second pass over it.  workspace: ap_act_bwd_bias_workspace_floats() floats. */
int64_t ap_act_bwd_bias_workspace_floats(int32_t N, int32_t C, int32_t H, int32_t W);
int ap_act_bwd_bias(const float* g1, int32_t g1_pad, const float* g2, const float* out, int32_t act, int32_t N, int32_t C,
                    int32_t H, int32_t W, float* dy, float* workspace, float* db, ap_stream_t stream);
/* db[c] = sum over n and pixels of dy (layers whose bias is live) */
int ap_bias_grad(const float* dy, int32_t N, int32_t C, int32_t HW, float* db, ap_stream_t stream);
/* the same sum in two stages (parallel over n and slices of the plane, then a fixed-order add): the form to use when
 * C is small (the 1-channel output layers); workspace: ap_bias_grad_workspace_floats() floats */
int64_t ap_bias_grad_workspace_floats(int32_t N, int32_t C, int32_t HW);
int ap_bias_grad_ws(const float* dy, int32_t N, int32_t C, int32_t HW, float* workspace, float* db, ap_stream_t stream);
/* backward of ap_warp_concat_fwd w.r.t. x (gout: N x 2C x H x W -> dx: N x C x H x W, zeroed inside) */
int ap_warp_concat_bwd(const float* gout, const float* motion, const float* flow, const float* ifmask,
                       float* dx, int32_t N, int32_t C, int32_t H, int32_t W, int32_t S, float flow_scale,
                       ap_stream_t stream);

/* ======================================================================= train-step helpers
 * sparse_image_warp (Module2/models/sparse_image_warp.py:35-58): order-2 polyharmonic spline through n control
 * points (src -> dst, (row, col) order), dense flow, bilinear warp with edge clamping.  Batched (the reference is
 * b=1 only).  ap_tps_solve: coef = B x (n+3) x 2 (spline weights w then affine v); *status (optional device int) is
 * set to 1 if a system is singular (the reference drops into pdb there, :124-128).  ap_tps_warp: img / out are
 * B x C x H x W (the reference passes NHWC; for its C=1 use both layouts coincide); flow_out (optional) B x H x W x 2. */
int ap_tps_solve(const float* src, const float* dst, int32_t B, int32_t n, float* coef, int32_t* status,
                 ap_stream_t stream);
int ap_tps_warp(const float* img, const float* dst, const float* coef, int32_t B, int32_t n, int32_t C, int32_t H,
                int32_t W, float* out, float* flow_out, ap_stream_t stream);
/* torch.optim.Adam(lr, betas=(beta1, beta2), eps) single step over a flat buffer
 * (Module2/models/geomgm_ifw_fore_model.py:346-360); step counts from 1. */
int ap_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, int32_t step, ap_stream_t stream);

/* ======================================================================= losses, compositing, aux-net glue, rasterisers
 * (csrc/losses.hip).  Reductions are deterministic (fixed-order two-stage sums); `workspace` holds
 * ap_reduce_workspace_floats() floats.
 *
 * ap_reduce_mean: out[0] = weight * mean(term_i), i < n, with
 *   op 0  term = (a_i - c)^2            GANLoss('lsgan'), c = 1.0 / 0.0 (Module2/models/networks.py:429-430, 455-473)
 *   op 1  term = |a_i - b_i|            nn.L1Loss (warp / coherence losses, geomgm_ifw_fore_model.py:734-739)
 *   op 2  term = (a_i + c) * b_i        lip-line loss mean((fake + 1) * mask) (:715-719)
 * ap_reduce_mean_bwd: ga_i = gout[0] * weight / n * d term_i / d a_i (gout: device scalar, no host sync). */
int64_t ap_reduce_workspace_floats(void);
int ap_reduce_mean(int32_t op, const float* a, const float* b, float c, int64_t n, float weight, float* workspace,
                   float* out, ap_stream_t stream);
int ap_reduce_mean_bwd(int32_t op, const float* a, const float* b, float c, int64_t n, float weight, const float* gout,
                       float* ga, ap_stream_t stream);
/* Mask compositing in the reference's operation order (bit-identical to the ATen chain).  a: N x C x HW, m: N x 1 x HW,
 * s: N x Cs x HW (Cs = 1 or C, mode 2 only); out: N x (C + append_mask) x HW, the appended channel is m.
 *   mode 0  (a/2+.5)*m*2-1                         BaseModel.masked type 0 (base_model.py:240-241)
 *   mode 1  ((a/2+.5)*m + 1 - m)*2-1               masked type 1 / 3 (:242-247), foreground on white (geomgm_ifw_fore_model.py:523-527)
 *   mode 2  ((a/2+.5)*m + (s/2+.5)*(1-m))*2-1      background blend with the static drawing (:541, 543)
 *   mode 3  a                                      masked type 2 (append_mask = 1)
 * ap_mask_composite_bwd: ga = g[:, :C] * m (modes 0-2) / g[:, :C] (mode 3); g has GC >= C channels. */
int ap_mask_composite(const float* a, const float* m, const float* s, int32_t N, int32_t C, int32_t Cs, int32_t HW,
                      int32_t mode, int32_t append_mask, float* out, ap_stream_t stream);
int ap_mask_composite_bwd(const float* g, const float* m, int32_t N, int32_t C, int32_t GC, int32_t HW, int32_t mode,
                          float* ga, ap_stream_t stream);
/* dst += alpha * src: accumulation of a network's gradient block into the optimiser's flat gradient buffer
 * (what autograd's AccumulateGrad does per parameter in the reference, loss.backward() at :586, 610, 634, 780). */
int ap_axpy(float* dst, const float* src, int64_t n, float alpha, ap_stream_t stream);
/* Window crop into a ones-filled (x2-x1)^2 box + resize, fused: the glue in front of the frozen auxiliary nets.
 *   get_lm (geomgm_ifw_fore_model.py:390-410): BGR / x3 channel map, bicubic align_corners=False to 112^2, (v+1)*0.5
 *   FaceLoss.crop_head_bbox (networks.py:2946-2966): bilinear align_corners=True to 112 x 96
 * x: N x C x H x W; win: N x 4 int32 ON THE DEVICE, [x1, x2, y1, y2] per sample; out: N x OC x OH x OW (OC <= 4),
 * out[:, k] = scale * resize(box of channel (chmap >> 8k) & 255) + shift.  mode 0 bilinear/align_corners=True,
 * mode 1 bicubic (A = -0.75)/align_corners=False.  ap_crop_resize_bwd: gx (N x C x H x W, zeroed inside) = d/dx. */
int ap_crop_resize_fwd(const float* x, const int32_t* win, int32_t N, int32_t C, int32_t H, int32_t W, int32_t OC,
                       int32_t chmap, int32_t OH, int32_t OW, int32_t mode, float scale, float shift, float* out,
                       ap_stream_t stream);
int ap_crop_resize_bwd(const float* gout, const int32_t* win, int32_t N, int32_t C, int32_t H, int32_t W, int32_t OC,
                       int32_t chmap, int32_t OH, int32_t OW, int32_t mode, float scale, float* gx, ap_stream_t stream);
/* kp_to_map(mode='binary') (geomgm_ifw_fore_model.py:19-44): lm N x P x 2 (x, y) -> out N x P x S x S in {0, 1},
 * out = ((x - lm_x*num/den)^2 + (y - lm_y*num/den)^2 <= radius^2); the reference calls it with S=224, num/den = 7/8,
 * radius 4 on the CPU with numpy (:71-73) -- a GPU -> CPU -> GPU round trip per frame there. */
int ap_kp_to_map(const float* lm, int32_t N, int32_t P, int32_t S, float num, float den, float radius, float* out,
                 ap_stream_t stream);
/* Tail of flow_network_warp (geomgm_ifw_fore_model.py:75-83): mask = (argmax_c vis < 2), flow' = flow * gain * mask / den
 * * num, both resized S -> OS bilinear align_corners=True.  flow N x 2 x S x S, vis N x VC x S x S ->
 * flow_out N x 2 x OS x OS, mask_out N x 1 x OS x OS.  Reference values: gain 20, num/den = 8/7, S 224, OS 256. */
int ap_flow_post(const float* flow, const float* vis, int32_t N, int32_t VC, int32_t S, int32_t OS, float gain, float num,
                 float den, float* flow_out, float* mask_out, ap_stream_t stream);
/* nn.PixelShuffle(2): x (N, 4 C, H, W) -> y (N, C, 2 H, 2 W), y[n][c][2h+i][2w+j] = x[n][4c+2i+j][h][w]
 * (decoder upsampling of FlowUnet_v2, Module2/intrinsic_flow_models/networks.py:693-698). */
int ap_pixel_shuffle2(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float* y, ap_stream_t stream);

/* getlipline (Module2/models/geomgm_ifw_fore_model.py:507-515): out[n] (size x size, {0, 1}) = union over the segments
 * (seg_a[i], seg_b[i]) of cv2.line(mask, lands[n][a], lands[n][b], 255, thickness) -- OpenCV 4.2 ThickLine (quad through
 * FillConvexPoly at 16.16 fixed point + a filled circle at both ends), float coordinates truncated to int as the Python
 * binding does.  lands: device (N, P, 2) as (x, y); seg_a / seg_b: HOST arrays of nseg <= 32 landmark indices;
 * thickness 2..16.  replaces the numpy + cv2 loop of getlipline (one launch + one memset). */
int ap_lip_line_mask(const float* lands, int32_t N, int32_t P, const int32_t* seg_a, const int32_t* seg_b, int32_t nseg,
                     int32_t size, int32_t thickness, float* out, ap_stream_t stream);
/* draw2(op=0) (Module2/data/umlvdfw_test_dataset.py:34-41): filled cv2.circle(radius) at np.round(lm) for every
 * landmark; out N x 1 x H x W = hi inside / lo outside (the reference: +1 / -1).  The disc is OpenCV's octant-walk
 * fill (drawing.cpp Circle(), opencv-python 4.2 pinned by requirements.txt:2), whose row half-widths
 * ap_circle_rows returns (hw[0..radius]); radius <= 31. */
int ap_landmark_discs(const float* lm, int32_t N, int32_t P, int32_t H, int32_t W, int32_t radius, float lo, float hi,
                      float* out, ap_stream_t stream);
int ap_circle_rows(int32_t radius, int32_t* hw);

/* The recurrence of one direction of one nn.LSTM layer over a whole sequence, time loop inside the kernel (lstm.hip): replaces the
 * per-time-step library kernels behind nn.LSTM in the AutoVC content converter (Module1/src/autovc/retrain_version/model_vc_37_1.py:
 * 75 `self.lstm = nn.LSTM(dim_enc, dim_neck, 2, batch_first=True, bidirectional=True)` and :98 `nn.LSTM(..., dim_dec, 3)`).
 * xproj [B][T][4H] = W_ih x_t + b_ih + b_hh for every t (gate order i, f, g, o; a library GEMM on the caller's side), whh [4H][H],
 * h0 / c0 [B][H] or null (zeros), out [B][T][out_stride] with this direction's H values at column out_off, hn / cn the final state
 * or null.  H <= 64 at any B, or H in {256, 512} at B == 1 (workspace of ap_lstm_workspace_bytes(H) bytes); otherwise
 * AP_ERR_UNSUPPORTED.  ap_lstm_timed_out(workspace, H): 1 when the last H > 64 launch gave up waiting for its peer workgroups. */
int64_t ap_lstm_workspace_bytes(int32_t H);
int ap_lstm_recurrence(const float* xproj, const float* whh, const float* h0, const float* c0, float* out, float* hn, float* cn,
                       int32_t B, int32_t T, int32_t H, int32_t reverse, int32_t out_stride, int32_t out_off, void* workspace,
                       ap_stream_t stream);
int ap_lstm_timed_out(const void* workspace, int32_t H);

#ifdef __cplusplus
}
#endif
#endif
