/*
 * e3dge_hip.h -- C-ABI of libe3dge_hip.so: the MI355X (gfx950) volume-rendering hot path of E3DGE.
 *
 * Every entry point is `extern "C"`, takes plain device pointers + sizes + a HIP stream, allocates
 * nothing, never synchronises, and returns 0 on success or a negative E3DGE_ERR_* code (the message is
 * available from e3dge_last_error()).  All tensors are fp32, contiguous, resident in HBM.
 *
 * Each function names the reference interface it replaces (paths relative to the reference checkout).
 * The Python host side (cvpr23-e3dge_amd/_lib.py) binds these with ctypes; INTEGRATION.md shows the
 * stub a reference maintainer would add.
 */
#ifndef E3DGE_HIP_H
#define E3DGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* e3dge_stream_t; /* a hipStream_t (NULL = the legacy default stream) */

enum {
    E3DGE_OK = 0,
    E3DGE_ERR_INVALID_ARG = -1, /* bad size / unsupported configuration / null pointer */
    E3DGE_ERR_LAUNCH = -2,      /* hipLaunchKernel / hipFuncSetAttribute failed          */
    E3DGE_ERR_UNSUPPORTED = -3  /* valid for the reference but outside this build's coverage */
};

/* ABI version; bumped whenever a signature below changes. */
int e3dge_abi_version(void);
/* bit 0: built with -DE3DGE_EXPERIMENTAL (the A/B-only precisions of include/e3dge_hip_experimental.h exist) */
int e3dge_build_flags(void);
/* 0 when `stream` is not being captured into a HIP graph, otherwise a positive id unique to the capture session (-1: query failed).
 * Host-side caches that hand device buffers from one launch to a later one (the backbone record of E3dgeRenderArgs) use it to
 * keep producer and consumer inside the same capture -- a graph that holds only the consumer would replay against a stale buffer. */
int64_t e3dge_stream_capture_id(e3dge_stream_t stream);
/* Thread-local message of the most recent failing call on this thread ("" if none). */
const char* e3dge_last_error(void);

/* --------------------------------------------------------------------------------------------
 * StyleGAN2 custom ops (reference: project/models/op/)
 * ------------------------------------------------------------------------------------------ */

/*
 * Replaces fused_bias_act_op / fused_bias_act_kernel
 * (project/models/op/fused_bias_act_kernel.cu:19-98, binding fused_bias_act.cpp:11-20).
 *   y[i] = act(x[i] + bias[(i / step_b) % size_b]) * scale
 *   act*10+grad: 10/11 linear, 12 -> 0, 30 lrelu(x>0 ? x : alpha x), 31 lrelu-grad gated by sign of
 *   ref[i] (the saved forward OUTPUT), 32 -> 0.
 * bias may be NULL (size_b == 0), ref may be NULL (treated as 0, as the reference does).
 * n < 2^31 (the reference indexes with int32, :66).
 */
int e3dge_fused_bias_act(float* y, const float* x, const float* bias, const float* ref, int act,
                         int grad, float alpha, float scale, int64_t n, int64_t step_b,
                         int64_t size_b, e3dge_stream_t stream);

/*
 * StyledConv tail fused into one pass: NoiseInjection + FusedLeakyReLU
 * (project/models/stylesdf_model.py:459-466 and :500-507 followed by fused_act.py:55-118):
 *   y[b,c,h,w] = lrelu(x[b,c,h,w] + noise_weight[0] * noise[b % noise_batch, 0, h, w] + bias[c], alpha) * scale
 * x,y: (batch, channels, hw) ; noise: (noise_batch, hw) with noise_batch in {1, batch};
 * noise_weight: device pointer to the 1-element NoiseInjection.weight.
 */
int e3dge_noise_bias_act(float* y, const float* x, const float* noise, const float* noise_weight,
                         const float* bias, float alpha, float scale, int64_t batch,
                         int64_t channels, int64_t hw, int64_t noise_batch,
                         e3dge_stream_t stream);

/*
 * Replaces upfirdn2d_op / upfirdn2d_kernel{,_large}
 * (project/models/op/upfirdn2d_kernel.cu:49-369, binding upfirdn2d.cpp:12-23) for minor_dim == 1,
 * which is the only way the Python wrappers call it (upfirdn2d.py:27,78,96).
 * x: (major, in_h, in_w); k: (kh, kw) FIR, correlated FLIPPED (upfirdn2d_kernel.cu:137);
 * y: (major, out_h, out_w) with out = (in*up + pad0 + pad1 - k + down) / down (:237-240).
 * Negative pads crop.  kh, kw <= 32.
 */
int e3dge_upfirdn2d(float* y, const float* x, const float* k, int64_t major, int in_h, int in_w,
                    int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                    int pad_x1, int pad_y0, int pad_y1, e3dge_stream_t stream);
/* Half-precision forms of the two ops (ABI 11; the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF:
 * fused_bias_act_kernel.cu:79, upfirdn2d_kernel.cu:311).  x, y, bias, ref are IEEE fp16 tensors; the FIR taps stay fp32.  Arithmetic
 * is fp32 with ONE rounding (RNE) on store -- at least as accurate as the reference's scalar_t = half arithmetic, which rounds after
 * every operation; parity is stated against the fp32 op on the widened inputs, rounded to fp16 (tests/test_gpu_selftest_and_ops.py).
 * What the half forms are FOR: e3dge_fused_bias_act_f16 is a stream-rate kernel (half the bytes, 0.8 of HBM like the fp32 form).
 * e3dge_upfirdn2d_f16 exists for DISPATCH PARITY with upfirdn2d_kernel.cu:311 only: it stages the same fp32 LDS patch as the fp32 form
 * and is bound by that stage, so a half Blur takes the wall time of the fp32 Blur (0.31 of HBM on half the bytes, bench `stream_ops`);
 * the decoder itself never calls it -- its blurs are fused into the packed fp32-accurate convolutions (e3dge_dec2_forward). */
int e3dge_fused_bias_act_f16(void* y, const void* x, const void* bias, const void* ref, int act, int grad, float alpha, float scale,
                             int64_t n, int64_t step_b, int64_t size_b, e3dge_stream_t stream);
int e3dge_upfirdn2d_f16(void* y, const void* x, const float* k, int64_t major, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                        int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, e3dge_stream_t stream);
/* Double-precision forms of the two ops (ABI 12): the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF (fused_bias_act_kernel.cu:79,
 * upfirdn2d_kernel.cu:311) includes double, which is what torch.autograd.gradcheck / gradgradcheck feed an op.  Arithmetic is double
 * throughout (scalar_t = double in the reference); alpha / scale are widened from float as fused_bias_act.cpp:11-20 passes them; the FIR
 * taps are double (the reference reads kernel.data_ptr<scalar_t>()).  Plain element kernels: correctness tools, not stream-rate code. */
int e3dge_fused_bias_act_f64(double* y, const double* x, const double* bias, const double* ref, int act, int grad, float alpha, float scale,
                             int64_t n, int64_t step_b, int64_t size_b, e3dge_stream_t stream);
int e3dge_upfirdn2d_f64(double* y, const double* x, const double* k, int64_t major, int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                        int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, e3dge_stream_t stream);
/* Output extent helper (same formula as above); returns <0 if the result would be empty. */
int e3dge_upfirdn2d_out_size(int in, int up, int down, int pad0, int pad1, int k);

/*
 * ModulatedConv2d weight preparation (project/models/stylesdf_model.py:317-338) in one pass:
 *   w'[b,o,i,:] = scale * weight[o,i,:] * style[b,i];  if demod: w' *= rsqrt(sum_{i,k} w'^2 + 1e-8)
 * weight: (co, ci, kk) ; style: (batch, ci) (already through `modulation`) ;
 * transpose == 0 -> out (batch*co, ci, kk)   [F.conv2d, groups=batch]
 * transpose == 1 -> out (batch*ci, co, kk)   [F.conv_transpose2d, groups=batch]
 */
int e3dge_modconv_weights(float* out, const float* weight, const float* style, float scale,
                          int demodulate, int transpose, int batch, int co, int ci, int kk,
                          e3dge_stream_t stream);

/*
 * Fused modulated 3x3 convolution.  Replaces ModulatedConv2d.forward for kernel_size 3
 * (project/models/stylesdf_model.py:317-362: weight modulation + demodulation + F.conv2d(padding=1, groups=B) or
 * F.conv_transpose2d(stride=2, groups=B)) and, for the stride-1 layers, StyledConv's NoiseInjection + FusedLeakyReLU
 * tail (:459-466, :500-507) in the epilogue.  No per-sample weight tensor exists: the style scales the input patch
 * while it is staged, demod[b,co] scales the accumulators; the contraction runs as split-f16 (hi+lo, 3 products, fp32
 * accumulate) MFMAs against a weight image packed once per layer.
 *   e3dge_modconv_packed_words(co, ci): 32-bit words of the image.
 *   e3dge_modconv_pack_weights: weight (co, ci, 3, 3) [ModulatedConv2d.weight[0]], scale = 1/sqrt(ci*9) (:302) ->
 *       image, and wsq (co, ci) = sum_k (scale w)^2 for the demodulation.
 *   e3dge_modconv_demod: style (batch, ci) [= modulation(style), :319] -> demod (batch, co) = rsqrt(sum_ci s^2 wsq + 1e-8)
 *       (:323-324; skipped when demodulate == 0) and s_amax (batch) = max_ci |s|.
 *   amax buffers: max|tensor| tracked by the producer of an activation, so the next conv can choose its operand scale
 *       without re-reading the data: E3DGE_AMAX_FLOATS floats, zero-initialised by the caller; producers atomically max
 *       into one of E3DGE_AMAX_SLOTS slots (E3DGE_AMAX_STRIDE floats apart), the consumer takes the maximum over the slots.
 *   e3dge_amax: max|x| of a tensor into such a buffer (for inputs whose producer did not track it).
 *   e3dge_modconv3x3: x (batch, ci, H, W) -> y (batch, co, H, W), or with upsample (batch, co, 2H+1, 2W+1) = the
 *       transposed convolution BEFORE the FIR blur.  in_amax: amax buffer with max over slots >= max|x| (operand scaling;
 *       any upper bound works, a tight one keeps full precision).  act != 0 (stride-1 only):
 *       y = lrelu(conv * demod + noise_w[0] * noise[b % noise_batch] + bias[co], negative_slope) * act_scale.
 *       out_amax (optional amax buffer) receives max|y|.  ci %% 16 == 0, co %% 32 == 0.
 */
#define E3DGE_AMAX_SLOTS 64
#define E3DGE_AMAX_STRIDE 32
#define E3DGE_AMAX_FLOATS (E3DGE_AMAX_SLOTS * E3DGE_AMAX_STRIDE)
typedef struct E3dgeModconvArgs {
    const float* x; const uint32_t* wimg; const float* style; const float* demod; const float* in_amax;
    const float* s_amax; const float* noise; const float* noise_w; const float* bias;
    float* y; float* out_amax;
    float negative_slope, act_scale;
    int act, upsample, batch, ci, co, height, width, noise_batch;
} E3dgeModconvArgs;
/* One row of the decoder's modulation table (device-resident array) for e3dge_decoder_styles: all layers' modulation
 * vectors s = conv.modulation(latent[:, latent_index]) (EqualLinear, stylesdf_model.py:234-244 as used at :319), their
 * demodulation factors and max|s| in two launches instead of ~4 tiny kernels per layer.  row_start / co_start are the prefix
 * sums of ci / co over the table (co counts 0 for rows without demod_out). */
typedef struct E3dgeModLayer {
    const float* mod_weight;   /* (ci, style_dim)  conv.modulation.weight                                  */
    const float* mod_bias;     /* (ci)             conv.modulation.bias                                    */
    const float* wsq;          /* (co, ci) from e3dge_modconv_pack_weights, or NULL                         */
    float* style_out;          /* (batch, ci) out                                                          */
    float* demod_out;          /* (batch, co) out, or NULL (ToRGB: no demodulation)                         */
    float* s_amax_out;         /* (batch) out, or NULL                                                     */
    int ci, co, latent_index, row_start, co_start;
    float lin_scale, lr_mul;   /* EqualLinear.scale = lr_mul / sqrt(style_dim), lr_mul                      */
} E3dgeModLayer;
int e3dge_decoder_styles(const E3dgeModLayer* table, int n_layers, int total_rows, int total_co, const float* latent,
                         int n_latent, int style_dim, int batch, e3dge_stream_t stream);
int64_t e3dge_modconv_packed_words(int co, int ci);
int e3dge_modconv_pack_weights(uint32_t* image, float* wsq, const float* weight, float scale, int co, int ci,
                               e3dge_stream_t stream);
int e3dge_modconv_demod(float* demod, float* s_amax, const float* style, const float* wsq, int batch, int co, int ci,
                        int demodulate, e3dge_stream_t stream);
int e3dge_amax(float* out, const float* x, int64_t n, e3dge_stream_t stream);
/* ABI 14: the same over the first `width` columns of `n_rows` rows of pitch `ld` floats (4-byte aligned): a column block of a wider row tensor,
 * e.g. the first 256 of the 301 gradient columns torch.cat's backward hands to Fuse_sft_MLP (sft.py:84-110 behind e3dge_full_runner.py:185-317). */
int e3dge_amax_rows(float* out, const float* x, int64_t n_rows, int width, int64_t ld, e3dge_stream_t stream);
int e3dge_modconv3x3(const E3dgeModconvArgs* args, e3dge_stream_t stream);

/*
 * Blur + StyledConv tail of the up-sampling layers in one pass: y = lrelu(upfirdn2d(x, k, pad=(pad0,pad1)) +
 * noise_weight[0] * noise + bias[c], alpha) * scale  (stylesdf_model.py:346 then :459-466, :500-507), k a 4x4 FIR.
 * x (batch, channels, in_h, in_w) -> y (batch, channels, in_h+pad0+pad1-3, ...); noise (noise_batch, out_h*out_w) or NULL;
 * out_amax: optional amax buffer (see e3dge_modconv3x3) receiving max|y|.
 */
int e3dge_blur_noise_bias_act(float* y, const float* x, const float* k, const float* noise, const float* noise_weight,
                              const float* bias, float alpha, float scale, int64_t batch, int64_t channels, int in_h,
                              int in_w, int pad0, int pad1, int64_t noise_batch, float* out_amax, e3dge_stream_t stream);

/*
 * ToRGB.forward (stylesdf_model.py:531-541) in one pass: 1x1 modulated conv without demodulation (weight (3, ci) =
 * conv.weight[0,:,:,0,0], style (batch, ci) = conv.modulation(style), scale = 1/sqrt(ci)) + bias (3) + the skip image
 * (batch, 3, H/2, W/2) up-sampled with upfirdn2d(up=2, pad=(2,1)) by the 4x4 FIR `fir` (Upsample.kernel), or NULL.
 * x (batch, ci, H, W) -> y (batch, 3, H, W).  W %% 4 == 0.
 */
int e3dge_torgb(float* y, const float* x, const float* weight, const float* style, const float* bias, const float* skip,
                const float* fir, float scale, int batch, int ci, int height, int width, e3dge_stream_t stream);

/*
 * ---- Packed decoder pipeline ("dec2"): Decoder.forward (project/models/stylesdf_model.py:741-797) as ONE native call ----
 *
 * The up-sampler's activations never exist as fp32 planes between its kernels.  Every producer writes its output already
 * split for the f16 matrix pipe, in the layout the next convolution's LDS-DMA consumes verbatim:
 *
 *   packed tensor (batch, C/8, 2, H+2, W+2) of 16-byte entries:  entry (b, g, hl, y, x) = the eight f16 values
 *       hl = 0: hi = f16_rtz(v * 2^(141 - eb)),   hl = 1: lo = f16(v * 2^(141 - eb) - hi)       of channels 8g .. 8g+7
 *   at pixel (y-1, x-1); the one-entry border is zero (= the convolutions' zero padding) and is never written by a
 *   producer, so the caller zero-fills a packed buffer ONCE (workspace) and reuses it.  eb (one int per tensor, in
 *   `meta`) is the biased exponent of an a-priori bound on max|v| that the producer computes before its first store:
 *   stride-1 conv: act_scale * (amax_in * sqrt(9 ci) + |noise_w| amax_noise + max|bias|) (Cauchy-Schwarz: demodulated
 *   filters have unit norm); blur: act_scale * (amax_T + ...).  The bound may be 2^12 loose before the representation
 *   (22 bits relative to the tensor's maximum) degrades to fp32's 24.
 *
 * Weights: per sample, w'' = ((scale W) s) demod exactly as the reference rounds them (:319-326), times 128, split into
 * f16 hi/lo MFMA A-fragments (e3dge_modconv_packed_words words per sample), rebuilt by one launch per forward from the
 * pre-arranged fp32 image `wpre` (e3dge_dec2_prepack_weights, once per weight update).
 *
 * Kernels of one forward, all launched by e3dge_dec2_forward on `stream` (nothing else, no allocation, no sync unless
 * kernel_ms is given): styles (2), amax + pack of the features, per-sample weights, conv1, ToRGB, then per level: transposed
 * conv by output phase -> fp32 T, blur + noise + bias + lrelu -> packed, stride-1 conv (+ noise + bias + lrelu) -> packed,
 * ToRGB (+ FIR-up-sampled skip).  The convolutions stage BOTH operands by LDS-DMA (no VGPR staging, no conversion).
 */
#define E3DGE_DEC2_MAX_UP 6
typedef struct E3dgeDec2Conv {
    const float* wpre;        /* e3dge_dec2_prepack_weights image of ModulatedConv2d.weight (co*ci*9 floats)            */
    const float* style;       /* (batch, ci)  conv.modulation(latent) -- written by the styles launch of this call       */
    const float* demod;       /* (batch, co)  demodulation factors    -- idem                                            */
    uint32_t* wimg;           /* workspace: batch * e3dge_modconv_packed_words(co, ci) words                              */
    const float* noise;       /* (noise_batch, 1, OH, OW) or NULL                                                        */
    const float* noise_w;     /* NoiseInjection.weight (device scalar), required with noise                              */
    const float* noise_amax;  /* amax buffer holding max|noise| (e3dge_amax), required with noise                        */
    const float* bias;        /* (co) FusedLeakyReLU.bias                                                                */
    float bias_amax;          /* max|bias| (host value, computed when the weights are packed)                            */
    int32_t ci, co, noise_batch;
} E3dgeDec2Conv;
typedef struct E3dgeDec2Rgb {
    const float* weight;      /* (3, ci) ToRGB.conv.weight                                                               */
    const float* style;       /* (batch, ci)                                                                             */
    const float* bias;        /* (3)                                                                                     */
    float* wm;                /* workspace (batch, 3, ci): (scale W) s                                                   */
    float* out;               /* (batch, 3, res, res): this level's image (the last one is Decoder.forward's result)     */
    float scale;              /* 1 / sqrt(ci)                                                                            */
    int32_t ci;
} E3dgeDec2Rgb;
typedef struct E3dgeDec2Plan {
    int32_t batch, n_up, in_res, in_ch;
    const float* features;                         /* (batch, in_ch, in_res, in_res) fp32                                 */
    const float* skip_in;                          /* rgbd_in of Decoder.forward (batch, 3, in_res, in_res) or NULL        */
    const E3dgeModLayer* mod_table;                /* device table for the styles launch (see e3dge_decoder_styles)        */
    const float* latent;                           /* (batch, n_latent, style_dim)                                        */
    int32_t n_mod, mod_rows, mod_co, n_latent, style_dim, reserved0;
    E3dgeDec2Conv conv1;
    E3dgeDec2Rgb rgb1;
    E3dgeDec2Conv up[E3DGE_DEC2_MAX_UP];           /* up-sampling StyledConv of level u (convs[2u])                        */
    E3dgeDec2Conv conv[E3DGE_DEC2_MAX_UP];         /* stride-1 StyledConv of level u (convs[2u+1])                         */
    E3dgeDec2Rgb rgb[E3DGE_DEC2_MAX_UP];
    uint32_t* act[2 * E3DGE_DEC2_MAX_UP + 2];      /* packed workspaces: [0] features, [1] conv1 out, [2+2u] blur out, [3+2u] conv out */
    float* tbuf[E3DGE_DEC2_MAX_UP];                /* (batch, co, 2H+3, 2W+4) fp32, zero-filled once: transposed-conv outputs */
    float* amax;                                   /* (3 n_up + 2) amax buffers (zeroed by the call): [0] features, [1] conv1 out, [2+3u] T, [3+3u] blur out, [4+3u] conv out */
    int32_t* meta;                                 /* (2 n_up + 2) ints: eb of act[i]                                      */
    const float* fir_blur;                         /* 4x4 taps of the up-sampling convs' Blur (make_kernel * 4)            */
    const float* fir_up;                           /* 4x4 taps of ToRGB's Upsample                                        */
    float negative_slope, act_scale;
    float* kernel_ms;                              /* host array, n_kernel_ms floats, or NULL: HIP-event time of every launch (makes the call synchronous) */
    int32_t n_kernel_ms, reserved1;
    /* ABI 11: when the blur kernel is rank one with a SYMMETRIC factor (make_kernel([1,3,3,1]) is), fir_blur = outer(fir_blur_1d,
     * fir_blur_1d), fir_blur_1d = (g0, g1, g1, g0) and fir_blur_separable != 0: the fused up-sampling kernel then applies the two 1-D passes (horizontal in registers,
     * vertical through LDS).  With fir_blur_separable == 0 the 4x4 taps are applied as they are (first-generation kernel). */
    float fir_blur_1d[4];
    int32_t fir_blur_separable;
    /* ABI 12: != 0 keeps the packed activation of the LAST convolution too (normally it only exists inside the fused ToRGB epilogue):
     * e3dge_dec2_backward reads the sign of every stored activation for lrelu'. */
    int32_t save_for_backward;
} E3dgeDec2Plan;
/* 32-bit words of a packed tensor / floats of a T buffer / floats of a wpre image */
int64_t e3dge_dec2_act_words(int batch, int channels, int res);
int64_t e3dge_dec2_tbuf_floats(int batch, int co, int in_res);
/* weight (co, ci, 3, 3) -> wpre[t][c][tap][lane][j] = scale * weight[32t + (lane & 31)][16c + 8 (lane >> 5) + j][tap] */
int e3dge_dec2_prepack_weights(float* wpre, const float* weight, float scale, int co, int ci, e3dge_stream_t stream);
/* launches per forward with n_up levels (= number of kernel_ms entries written): 6 + 4 n_up */
int e3dge_dec2_num_launches(int n_up);
int e3dge_dec2_forward(const E3dgeDec2Plan* plan, e3dge_stream_t stream);
/*
 * ---- Backward of the packed pipeline (ABI 12): d image -> d features, generator frozen ----------------------------------------------
 * The data gradient of Decoder.forward (project/models/stylesdf_model.py:742-797; ModulatedConv2d :317-362, StyledConv :469-507, ToRGB
 * :531-541, Blur / Upsample -> op/upfirdn2d.py:18-142, FusedLeakyReLU -> op/fused_act.py:19-84) that train_ae.py's stage-1 step takes
 * through the pixel loss on pool_256(gen_imgs) (trainers/trainer.py:1017-1031, :728) -- in the reference: autograd through F.conv2d /
 * F.conv_transpose2d on the modulated weights and the two custom ops' backward classes.  Here: the same MFMA convolutions on TRANSPOSED
 * per-sample weight images, gradients packed like the activations, lrelu' from the sign of the stored packed activations (csrc/
 * decoder2_bwd.h).  Call it after e3dge_dec2_forward(plan) ran with plan->save_for_backward != 0 on the same workspace and BEFORE
 * anything else overwrites that workspace (act[], style / demod / wm buffers).  d latent and parameter gradients are NOT produced.
 * Launches: norms (+ clearing the amax block), transposed weights, amax(d img), ToRGB^T + mask at the top, then per level
 * Upsample^T of d rgb, conv^T, Blur^T + phase split, convT^T (+ ToRGB^T of the level below), and conv1^T: 5 + 4 n_up.
 */
typedef struct E3dgeDec2BwdConv {
    const float* wpre_t;      /* e3dge_dec2_prepack_weights_t image of ModulatedConv2d.weight (co*ci*9 floats)                        */
    const float* wcol;        /* (ci) column sums over co of the (co, ci) table sum_taps (scale W)^2 that e3dge_modconv_pack_weights writes */
    uint32_t* wimg_t;         /* workspace: batch * e3dge_modconv_packed_words(ci, co) words                                          */
} E3dgeDec2BwdConv;
typedef struct E3dgeDec2BwdPlan {
    const float* d_img;                            /* (batch, 3, R, R), R = in_res << n_up: gradient of Decoder.forward's image       */
    float* d_features;                             /* (batch, in_ch, in_res, in_res): result                                          */
    E3dgeDec2BwdConv conv1;
    E3dgeDec2BwdConv up[E3DGE_DEC2_MAX_UP];
    E3dgeDec2BwdConv conv[E3DGE_DEC2_MAX_UP];
    uint32_t* gact[2 * E3DGE_DEC2_MAX_UP + 2];     /* packed gradient workspaces, shapes of plan->act[i], zero-filled ONCE; [0] unused */
    uint32_t* pbuf;                                /* phase planes of Blur^T: e3dge_dec2_pbuf_words(batch, co, res) of the largest up level */
    float* drgb[E3DGE_DEC2_MAX_UP];                /* [i]: (batch, 3, r, r) gradient of the ToRGB image of level i - 1 (r = in_res << i) */
    float* amax;                                   /* (4 n_up + 2) amax buffers (cleared by the call)                                 */
    int32_t* meta;                                 /* (3 n_up + 1) ints                                                               */
    float* bounds;                                 /* (3 n_up + 2) floats: operator norms of the transposed images / ToRGB tables     */
    float* kernel_ms;                              /* host array or NULL: HIP-event time of every launch of the d-features chain (makes the call synchronous) */
    int32_t n_kernel_ms, reserved;
    /* optional: the gradient to the decoder's W+ latent, (batch, n_latent, style_dim), or NULL.  Formed from per-channel dot products of
     * the tensors the chain leaves in its workspace -- dL/ds[ci] = (1/s) sum_p x dx - s sum_co demod^2 wsq sum_p g y (3x3 layers),
     * (1/s) sum_p act (ToRGB^T d rgb) (ToRGB), then modulation^T -- no weight-gradient contraction (csrc/decoder2_bwd.h).  2 n_up + 3 more
     * launches.  Needs plan->features / mod_table as the forward had them and ds_part = e3dge_dec2_dlatent_ws_floats(plan) floats. */
    float* d_latent;
    float* ds_part;
    int64_t ds_part_floats;
} E3dgeDec2BwdPlan;
/* weight (co, ci, 3, 3) -> wpre_t[t][c][tap][lane][j] = scale * weight[16c + 8 (lane >> 5) + j][32t + (lane & 31)][flip ? 8 - tap : tap];
 * flip = 1 for the stride-1 convolutions, 0 for the up-sampling (transposed-stride) ones.  ci %% 32 == 0, co %% 16 == 0. */
int e3dge_dec2_prepack_weights_t(float* wpre_t, const float* weight, float scale, int co, int ci, int flip, e3dge_stream_t stream);
int64_t e3dge_dec2_pbuf_words(int batch, int channels, int res);
int e3dge_dec2_bwd_num_launches(int n_up);
int64_t e3dge_dec2_dlatent_ws_floats(const E3dgeDec2Plan* plan);
int e3dge_dec2_backward(const E3dgeDec2Plan* plan, const E3dgeDec2BwdPlan* bwd, e3dge_stream_t stream);
/* stand-alone pieces (tests, tools): fp32 (batch, c, res, res) <-> packed; both use meta[0] / amax as e3dge_dec2_forward does */
int e3dge_dec2_pack(uint32_t* packed, int32_t* meta, const float* x, const float* amax, int batch, int channels, int res,
                    e3dge_stream_t stream);
int e3dge_dec2_unpack(float* x, const uint32_t* packed, const int32_t* meta, int batch, int channels, int res,
                      e3dge_stream_t stream);

/*
 * Reference-view hit probability (VolumeFeatureRenderer.query_hitting_probability_fixed_interval,
 * project/utils/volume_renderer.py:1326-1495; compositing with no_force_stop :826-837, :884).
 *   e3dge_hitprob_points: pts (batch, rays, s_pts, 3) world-space points of the query view, poses / extrinsics (batch, 3, 4) of the
 *       reference view (c2w / w2c), near / far (batch, rays), t_vals (n_samples) -> q (batch, rays, s_pts, n_samples, 3): the samples
 *       of the reference camera's ray through every point, and aux (batch, rays, s_pts, 4) = (lo, hi, frac, index) of the interpolation.
 *   e3dge_hitprob_composite: sdf (batch, rays, s_pts, n_samples) at q [e3dge_siren_points_fwd] -> out (batch, rays, s_pts): the
 *       compositing weight (visibility == 0) or transmittance (visibility != 0) interpolated at the point.
 */
int e3dge_hitprob_points(float* q, float* aux, const float* pts, const float* poses, const float* extrinsics, const float* near,
                         const float* far, const float* t_vals, int batch, int64_t rays, int s_pts, int n_samples, e3dge_stream_t stream);
int e3dge_hitprob_composite(float* out, const float* sdf, const float* aux, const float* near, const float* far, const float* t_vals,
                            float sigmoid_beta, int visibility, int batch, int64_t rays, int s_pts, int n_samples, e3dge_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * FiLM-SIREN volume renderer (reference: project/utils/volume_renderer.py)
 * ------------------------------------------------------------------------------------------ */

#define E3DGE_SIREN_WIDTH 256     /* W (options.py:841-844); the kernels are specialised for 256 */
#define E3DGE_SIREN_DEPTH 8       /* D backbone layers (options.py:837-840)                       */
#define E3DGE_SIREN_NLAYERS 9     /* D + views_linears                                           */

/* How the 256-wide contractions are evaluated.  Both accumulate in fp32 and meet the same parity bounds. */
#define E3DGE_PREC_F32 0          /* v_mfma_f32_32x32x2_f32 on fp32 operands                                     */
#define E3DGE_PREC_F16X3 1        /* operands split as f16 hi+lo, 3 f16 MFMA products per fp32 product, fp32 accumulate.  Forward
                                     launches: 8 waves x 16 points on v_mfma_f32_16x16x32_f16; backward-type launches: 4 waves x 32
                                     points on v_mfma_f32_32x32x16_f16 with per-point block scaling */
#define E3DGE_PREC_F16X3_V1 2     /* forward launches: the first-generation split-f16 kernel (4 waves x 32 points,
                                     v_mfma_f32_32x32x16_f16), kept for A/B measurements; backward-type launches: same as
                                     E3DGE_PREC_F16X3 */
#define E3DGE_PREC_F16X3_G2 3     /* The training configuration of E3DGE_PREC_F16X3 (round 6, csrc/siren16_bwd.h):
                                     * backward-type launches (e3dge_siren_bwd / _render_bwd / _sdf_grad / _tangent / _tangent_tr): the 8-wave x
                                       16-point layout of the forward kernel, saved-state streams by LDS-DMA through a per-wave ring.  Same
                                       block-scaled split-f16 arithmetic as E3DGE_PREC_F16X3; the second-order inputs come as the products
                                       ta_l r_l (e3dge_siren_tangent_tr);
                                     * forward launches (e3dge_siren_render_fwd / _points_fwd): the E3DGE_PREC_F16X3 kernel; `save_args` is
                                       written SLAB-MAJOR, which is what the launches above read (and write: rsave, tang): 16 consecutive
                                       rows of an image form a slab [L layers][16 tiles][lane 16 q + (row & 15)][4 floats] (feature =
                                       16 tile + 4 q + j; L = 9 for save_args, 8 for rsave / tang), so that a wave's tile is one contiguous
                                       KiB.  Rows per image are padded to a multiple of 16: every saved-state buffer of this precision
                                       holds batch * ceil16(n_pts) * L * 256 floats.  Point-major (batch, n_pts, L, 256) everywhere else. */

/* Number of floats of the packed weight image produced by e3dge_siren_pack_weights. */
int64_t e3dge_siren_packed_floats(void);

/*
 * Re-lays the SirenGenerator parameters (volume_renderer.py:158-166; FiLMSiren.weight/bias :91-104,
 * rgb_linear / sigma_linear :165-166) into the MFMA fragment-major image the render kernels stream
 * through LDS.  Call once per weight update.  All inputs row-major (out, in) as in the state dict:
 *   w_first (256,3)  b_first (256)            pts_linears.0
 *   w_hidden (7,256,256)  b_hidden (7,256)    pts_linears.1..7
 *   w_view (256,259)  b_view (256)            views_linears  (input = [features(256), viewdir(3)])
 *   w_rgb (3,256) b_rgb (3) ; w_sigma (1,256) b_sigma (1)
 * packed: e3dge_siren_packed_floats() floats.
 */
int e3dge_siren_pack_weights(float* packed, const float* w_first, const float* b_first,
                             const float* w_hidden, const float* b_hidden, const float* w_view,
                             const float* b_view, const float* w_rgb, const float* b_rgb,
                             const float* w_sigma, const float* b_sigma, e3dge_stream_t stream);

/*
 * FiLM parameters of all 9 layers for a batch of W+ codes
 * (FiLMSiren.gamma / .beta = LinearLayer(std_init=15,bias_init=30) / (std_init=0.25),
 *  volume_renderer.py:76-80,107-120):
 *   film[b,l,0,:] = 15  * (Wg_l styles[b,l] + bg_l) + 30
 *   film[b,l,1,:] = 0.25 * (Wb_l styles[b,l] + bb_l)
 * styles (batch, 9, 256); wg, wb (9,256,256); bg, bb (9,256); film (batch, 9, 2, 256).
 */
int e3dge_film_params(float* film, const float* styles, const float* wg, const float* bg,
                      const float* wb, const float* bb, int batch, e3dge_stream_t stream);

/* Inputs/outputs of one fused render launch; unused outputs may be NULL. */
typedef struct E3dgeRenderArgs {
    /* ---- inputs ---- */
    const float* packed;   /* from e3dge_siren_pack_weights                                      */
    const float* film;     /* (batch, 9, 2, 256) from e3dge_film_params                           */
    const float* c2w;      /* (batch, 3, 4) camera-to-world (cam_poses)                            */
    const float* focal;    /* (batch)                                                              */
    const float* near;     /* (batch)                                                              */
    const float* far;      /* (batch)                                                              */
    const float* t_vals;   /* (n_samples) -- VolumeFeatureRenderer.t_vals (:690-698)                */
    const float* tex_alpha;/* optional (batch,H,W,S,256) per-point texture FiLM alpha (:217-220)     */
    const float* tex_beta; /* optional, same shape                                                  */
    float sigmoid_beta;    /* renderer.sigmoid_beta (:663)                                         */
    float box_scale;       /* grid_warper scale = 2 / (2*dist_radius) (:720)                        */
    float mask_depth_thresh; /* 1.08 (:910)                                                        */
    int batch, height, width, n_samples;
    int res;               /* out_im_res used for the pixel-centre offset (:773-774)                */
    int force_background;  /* (:884-886)                                                           */
    int precision;         /* E3DGE_PREC_F32 or E3DGE_PREC_F16X3                                      */
    /* ---- outputs ---- */
    float* rgb;            /* (batch, 3, H, W)    gen_thumb_imgs (:888-890, permuted :1964)          */
    float* features;       /* (batch, 256, H, W)  (:894, :1967)                                     */
    float* xyz;            /* (batch, 3, H, W)    (:905, :1958)                                     */
    float* depth;          /* (batch, H, W)       (:907)                                            */
    float* mask;           /* (batch, H, W)       (:910)                                            */
    float* sdf;            /* (batch, H, W, S)    (:880)                                            */
    float* weights;        /* (batch, H, W, S)    hit_prob (:877,:885)                              */
    float* points;         /* (batch, H, W, S, 3) (:1231)                                           */
    float* rays_d;         /* (batch, H, W, 3)    (:782)                                            */
    float* viewdirs;       /* (batch, H, W, 3)    normalised (:1679)                                */
    float* dists;          /* (batch, H, W, S)    (:826-837)                                        */
    float* save_args;      /* training only, else NULL: (batch, H, W, S, 9, 256) pre-sine arguments of the 9 FiLM
                              layers, consumed by e3dge_siren_bwd                                        */
    /* ---- backbone hand-over between the two renders of one evaluated image (precision f16x3, inference; all NULL otherwise) ----
     * que_render_given_ref (e3dge_full_runner.py:185-317) renders the same rays twice: without and with the texture FiLM, which
     * enters behind the sdf head (volume_renderer.py:217-220).  Layers 0..7, the sdf head, alpha and the transmittance scan are
     * identical in both.  backbone_out: the first launch also writes its layer-7 output (e3dge_siren_backbone_bytes bytes).
     * backbone_in + weights_in (that launch's `weights` output): the second launch, same sizes / poses / film, reads them and
     * runs only texture FiLM -> view layer -> compositing; it produces rgb and features (+ ray geometry): sdf, weights, xyz,
     * depth, mask must be NULL there (the first launch's are the values).  Results are bit-identical to a full launch. */
    void* backbone_out;
    const void* backbone_in;
    const float* weights_in;
} E3dgeRenderArgs;

/*
 * Replaces VolumeFeatureRenderer.render -> render_rays -> run_network -> SirenGenerator.forward ->
 * volume_integration (volume_renderer.py:1666-1701, 1183-1287, 1052-1128, 168-264, 809-943) for the
 * inference configuration (offset_sampling, static_viewdirs, perturb=0, with_sdf, return_xyz):
 * ray generation, sample placement, the 9 FiLM-SIREN layers, both heads, SDF->alpha, transmittance
 * scan and all composites in ONE kernel; the (B,H,W,S,260) raw tensor never exists.
 * Requires n_samples >= 16 (or use e3dge_siren_points_fwd for un-composited queries).
 */
int e3dge_siren_render_fwd(const E3dgeRenderArgs* args, e3dge_stream_t stream);
/* bytes of the backbone record of a render launch with these sizes (0 if the sizes are not renderable) */
int64_t e3dge_siren_backbone_bytes(int batch, int height, int width, int n_samples);

/*
 * Replaces VolumeFeatureRenderer.run_network on an arbitrary point set
 * (volume_renderer.py:1052-1128; callers :925-930, :1916-1949, sample_uniform_grid :945-).
 * pts (batch, n_pts, 3) world-space (box warp applied inside); viewdirs (batch, n_pts, 3) or NULL (= 0).
 * Outputs (any may be NULL): sdf (batch, n_pts); raw (batch, n_pts, 260) = [rgb3, sdf1, feat256]
 * exactly as SirenGenerator.forward concatenates them (:259-261); save_args (batch, n_pts, 9, 256) as in
 * E3dgeRenderArgs (training only).
 */
int e3dge_siren_points_fwd(const float* packed, const float* film, const float* pts,
                           const float* viewdirs, float box_scale, int batch, int64_t n_pts,
                           float* sdf, float* raw, float* save_args, int precision, e3dge_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training direction: gradient of a loss w.r.t. the renderer's styles (W+ latents), generator weights frozen.
 * Replaces autograd through SirenGenerator.forward (project/utils/volume_renderer.py:168-264) as the encoder
 * trainer drives it (project/trainer.py:728 loss.backward(); the generator is frozen at :1568).
 *
 *   args     (batch, n_pts, 9, 256)  pre-sine arguments saved by the forward launch (save_args)
 *   d_feat   (batch, n_pts, 256)     dL/d(view-layer features) per point, or NULL (= 0)
 *   d_rgb    (batch, n_pts, 3)       dL/d(rgb head output, pre-sigmoid), or NULL
 *   d_sdf    (batch, n_pts)          dL/d(sdf head output), or NULL
 *   wg, wb   (9, 256, 256)           the gamma / beta style-linear weights as given to e3dge_film_params
 *   partials  e3dge_siren_bwd_partial_floats(batch, n_pts) floats of scratch (need not be initialised)
 *   dfilm    (batch, 9, 2, 256) out  dL/d(gamma, beta)
 *   dstyles  (batch, 9, 256)    out  dL/d(styles)
 * precision: E3DGE_PREC_F32 (fp32 MFMA) or E3DGE_PREC_F16X3: every gradient operand is scaled per point by a power of
 * two into [1, 2), split into f16 hi + lo, three f16 MFMA products accumulated in fp32 -- same error as fp32.
 * Optional outputs / inputs (NULL = absent):
 *   d_pts (batch, n_pts, 3) out: dL/d(query points) = box_scale * W_0^T (gamma_0 * adj(a_0)) -- with tang/rsave this is
 *     the Hessian-vector product the reference obtains by keeping xyz in the graph of the surface normals
 *     (volume_renderer.py:921-930); box_scale as given to the forward launch.
 *   tex_alpha (batch, n_pts, 256): the forward pass applied the texture FiLM h8' = (alpha+1) h8 + beta (:217-220);
 *     d_tex_alpha, d_tex_beta (batch, n_pts, 256) out receive dL/dalpha, dL/dbeta (stage-2 training differentiates the
 *     second pass, e3dge_full_runner.py:185-317).  Not combinable with tang/rsave.
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t e3dge_siren_bwd_partial_floats(int batch, int64_t n_pts);
typedef struct E3dgeSirenBwdArgs {
    const float* packed; const float* film; const float* args;
    const float* d_feat; const float* d_rgb; const float* d_sdf;
    const float* tang; const float* rsave;
    const float* wg; const float* wb;
    const float* tex_alpha;
    int batch; int precision;
    int64_t n_pts;
    float box_scale;
    float* partials; float* dfilm; float* dstyles;
    float* d_pts; float* d_tex_alpha; float* d_tex_beta;
} E3dgeSirenBwdArgs;
int e3dge_siren_bwd(const E3dgeSirenBwdArgs* args, e3dge_stream_t stream);

/* Eikonal term e = d sdf / d x (get_eikonal_term, volume_renderer.py:796-802) and its double backward.
 *   e3dge_siren_sdf_grad : from the saved arguments, e (batch, n_pts, 3) [world-space x: includes box_scale] and
 *                          rsave (batch, n_pts, 8, 256) = d sdf / d h_l, needed again if a loss on e is differentiated.
 *                          seed (batch, n_pts) scales r_7 per point, NULL = 1 (grad_outputs = ones, :799).
 *   e3dge_siren_tangent  : given v = dL/de (batch, n_pts, 3), the tangent arguments tang (batch, n_pts, 8, 256).
 * Passing tang + rsave to e3dge_siren_bwd / E3dgeRenderBwdArgs adds dL/d(styles) of the loss on e (the reference's
 * create_graph=True path) to the first-order gradient; NULL, NULL = no such loss. */
int e3dge_siren_sdf_grad(const float* packed, const float* film, const float* args, const float* seed,
                         float box_scale, int batch, int64_t n_pts, float* rsave, float* eik, int precision,
                         e3dge_stream_t stream);
int e3dge_siren_tangent(const float* packed, const float* film, const float* args, const float* v,
                        float box_scale, int batch, int64_t n_pts, float* tang, int precision, e3dge_stream_t stream);
/* ABI 13, precision E3DGE_PREC_F16X3_G2 only: the tangent pass in product form.  ta_l and r_l enter the second-order backward only as
 * ta_l * r_l (adj(a_l) -= sin(a_l) ta_l r_l, adj(gamma_l) += ta_l r_l cos(a_l) / gamma_l), so this launch reads rsave (the sdf
 * chain's r_l) beside the arguments and stores tr (batch, n_pts, 8, 256) = ta_l r_l; e3dge_siren_bwd / e3dge_siren_render_bwd in the
 * same precision take it as `tang` with `rsave` = NULL -- one 8-KB-per-point stream less in the longest kernel of the training step. */
int e3dge_siren_tangent_tr(const float* packed, const float* film, const float* args, const float* v, const float* rsave,
                           float box_scale, int batch, int64_t n_pts, float* tr, int precision, e3dge_stream_t stream);

/* Backward of e3dge_siren_render_fwd: volume_integration (volume_renderer.py:809-943) back to the per-point outputs
 * (one wave per ray), then the MLP chain above.  Gradient maps are ROW-MAJOR PER RAY (ray = (b*H + y)*W + x):
 *   d_rgb_map (rays,3)  d_feat_map (rays,256)  d_xyz_map (rays,3)  d_depth_map (rays)  d_sdf (rays,S); any may be NULL.
 * args/sdf/dists/points/weights are the forward launch's outputs (save_args, sdf, dists, points, weights).
 * d_rgb_pts (rays,S,3) and d_sdf_pts (rays,S) are scratch the caller provides; partials as for e3dge_siren_bwd with
 * n_pts = H*W*S.  sigmoid_beta and the generator weights get no gradient (frozen in encoder training).
 * d_weights (rays,S): gradient arriving at the compositing weights (`hit_prob`, read by cycle_runner.py:134), or NULL.
 * tex_alpha / d_tex_alpha / d_tex_beta (rays,S,256): as in E3dgeSirenBwdArgs, for the second (texture-FiLM) pass. */
typedef struct E3dgeRenderBwdArgs {
    const float* packed; const float* film; const float* args; const float* sdf; const float* dists;
    const float* points; const float* weights; const float* t_vals; const float* near; const float* far;
    const float* wg; const float* wb;
    const float* d_rgb_map; const float* d_feat_map; const float* d_xyz_map; const float* d_depth_map; const float* d_sdf;
    const float* tang; const float* rsave;
    const float* d_weights; const float* tex_alpha;
    float sigmoid_beta;
    int batch, height, width, n_samples, force_background;
    int precision;           /* E3DGE_PREC_F32 or E3DGE_PREC_F16X3 (block-scaled split-f16 GEMMs, fp32 accumulate) */
    float* d_rgb_pts; float* d_sdf_pts; float* partials;
    float* dfilm; float* dstyles;
    float* d_tex_alpha; float* d_tex_beta;
    int phase;               /* ABI 14: 0 = both launches; 1 = only the backward of the compositing (reads the d_*_map / d_sdf / d_weights inputs, writes
                                d_rgb_pts / d_sdf_pts; needs neither tang nor rsave); 2 = only the network backward on what phase 1 left in those buffers.
                                Lets a caller run phase 1 beside e3dge_siren_tangent(_tr) on another stream and wait for the tangent before phase 2. */
} E3dgeRenderBwdArgs;
int e3dge_siren_render_bwd(const E3dgeRenderBwdArgs* args, e3dge_stream_t stream);



/* ------------------------------------------------------------------------------------------------------------------
 * Local-feature -> texture-FiLM head (second renderer pass): out = W_s x + W_1 relu(W_0 relu(x) + b_0) + b_1,
 * (alpha, beta) = split(out, 256).  Replaces ResnetBlockFC.forward (project/models/helper_modules/resnetfc.py:49-58) as
 * netLocal.local_feat_to_tex_modulations_linear (vendor/pifu/lib/model/HGPIFuGANNetResidualInputResnetFC.py:84-93), called
 * in SirenLocalGlobal.forward_backbone (project/utils/volume_renderer.py:327-336).
 *   w0 (cin, cin), b0 (cin)   fc_0        w1 (512, cin), b1 (512)   fc_1        ws (512, cin)   shortcut (no bias)
 *   feats (n_pts, cin) row-major, cin <= 320        alpha, beta (n_pts, 256) out -- the tex_alpha / tex_beta inputs of
 *   e3dge_siren_render_fwd.  Split-f16 contraction with per-point block scaling, fp32 accumulate.
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t e3dge_resblock_packed_floats(void);
int e3dge_resblock_pack_weights(float* packed, const float* w0, const float* b0, const float* w1, const float* b1,
                                const float* ws, int cin, e3dge_stream_t stream);
int e3dge_tex_modulations_fwd(const float* packed, const float* feats, int cin, int64_t n_pts,
                              float* alpha, float* beta, e3dge_stream_t stream);
/* Data gradient of the head (round 5): d feats = W_s^T d out + (W_0^T ((W_1^T d out) [net > 0])) [x > 0], d out = [d alpha | d beta] -- what
 * autograd runs through ResnetBlockFC (helper_modules/resnetfc.py:49-58) for the stage-2 losses (e3dge_full_runner.py:185-317); the reference
 * has no hand-written backward.  One launch, split-f16 contractions like the forward; net is recomputed (signs only).
 *   packed_bwd   e3dge_resblock_bwd_packed_floats() floats from e3dge_resblock_bwd_pack_weights (W_0, W_1^T, W_s^T, W_0^T images + b_0)
 *   d_alpha, d_beta (n_pts, 256), 16-byte aligned     d_feats (n_pts, cin) out
 *   ws           e3dge_tex_modulations_bwd_ws_floats(n_pts) floats, 16-byte aligned; on return ws[0 .. n_pts * 320) holds d net = d L / d (fc_0
 *                output) as (n_pts, 320) rows (columns >= cin are zero) -- the operand of the parameter gradients (e3dge_wgrad)
 *   net_out      NULL, or (n_pts, 320) rows receiving net = fc_0(relu(x)) as recomputed (the input of fc_1's parameter gradient, before its relu)
 *   amax4        (ABI 14) NULL, or 4 x E3DGE_AMAX_FLOATS floats, zeroed by the caller: amax buffers of feats, [d alpha | d beta], d net and net (the
 *                last only with net_out) -- the operand scales e3dge_wgrad needs, from values this launch holds anyway */
int64_t e3dge_resblock_bwd_packed_floats(void);
int e3dge_resblock_bwd_pack_weights(float* packed_bwd, const float* w0, const float* b0, const float* w1, const float* ws, int cin,
                                    e3dge_stream_t stream);
int64_t e3dge_tex_modulations_bwd_ws_floats(int64_t n_pts);
int e3dge_tex_modulations_bwd(const float* packed_bwd, const float* feats, int cin, int64_t n_pts, const float* d_alpha, const float* d_beta,
                              float* d_feats, float* ws, float* net_out, float* amax4, e3dge_stream_t stream);
/* The head INSIDE the second render pass's data flow (ABI 11; SURVEY.md 8 f1 as specified: (alpha, beta) never materialise):
 * feats (batch, height, width, n_samples, cin) -> h' = (alpha + 1) h8 + beta, where h8 is the layer-7 output that render
 * pass #1 left in `backbone_in` (e3dge_siren_render_fwd backbone_out, e3dge_siren_backbone_bytes bytes) -- the FiLM step of
 * volume_renderer.py:217-220 in siren16_kernel's operation order, bit-identical to applying tex_alpha / tex_beta there.
 * `backbone_out` (same size and layout, zero-filled once by the caller: padding slabs are never written) is what pass #2 then
 * takes as its backbone_in, WITHOUT tex_alpha / tex_beta.  Records are limited to 4 GiB (32-bit offsets). */
int e3dge_tex_film_fwd(const float* packed, const float* feats, int cin, int batch, int height, int width, int n_samples,
                       const void* backbone_in, void* backbone_out, e3dge_stream_t stream);


/* ------------------------------------------------------------------------------------------------------------------
 * Per-point query of the local branch's feature maps (project/trainers/E3DGE/e3dge_full_runner.py:185-317).
 *   e3dge_local_query: replaces HGPIFuNetGAN.query(return_feat_only / return_projection_only / im_feat=...)
 *       (vendor/pifu/lib/model/HGPIFuGANNet.py:85-151) = perspective projection (vendor/pifu/lib/geometry.py:101-129; the
 *       sign of the depth is decided by the first point of the first sample, as there) + y flip + in-image mask + bilinear
 *       gather index() (geometry.py:64-80: grid_sample, zeros padding, align_corners=False).
 *       pts (batch, n_pts, 3) world space; calibs (batch, 3, 4); fmap_nhwc (batch, fh, fw, channels) CHANNEL-LAST, or NULL
 *       for projection / mask only.  Features go to out[(b*n_pts + n) * ld + col_off + c]; in_img (1.0 / 0.0) to
 *       in_img[(b*n_pts + n) * mask_ld + mask_off] (NULL = not wanted; may point into the same rows as `out`);
 *       proj (batch, n_pts, 3) = (x, flipped y, depth) or NULL.  channels %% 4 == 0.
 *   e3dge_pos_encoding: PosEncoding.forward (project/utils/misc_utils.py:148-185, logscale): pts (n_pts, 3) ->
 *       out[n * ld + col_off + ...] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(F-1) x), cos(2^(F-1) x)], 3 (2F+1) columns.
 * ---------------------------------------------------------------------------------------------------------------- */
int e3dge_local_query(float* out, int ld, int col_off, float* in_img, int mask_ld, int mask_off, float* proj,
                      const float* pts, const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts, int channels,
                      int fh, int fw, e3dge_stream_t stream);
/* Backward of the gather (ABI 11; reference: project/models/op/grid_sample_gradfix.py:52-89 behind HGPIFuNetGAN.query's index()):
 * d_fmap_nhwc (batch, fh, fw, channels), ZERO-FILLED by the caller, receives the bilinear scatter of d_out[..., col_off:col_off+C]
 * (atomic adds); d_pts (batch, n_pts, 3) the gradient through the sampling position (projection included).  Either may be NULL. */
int e3dge_local_query_bwd(float* d_fmap_nhwc, float* d_pts, const float* d_out, int ld, int col_off, const float* pts,
                          const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts, int channels, int fh, int fw,
                          e3dge_stream_t stream);
/* ABI 14 (round 6): the same with the points walked in the order of a counting sort by (image, pixel of the top-left corner) -- for gathers
 * whose consecutive points do NOT share pixels (another view's rays: every sample of the reference-view gather lands on its own pixel, and the
 * scatter is then one device-scope atomic instruction per corner, channel group and point).  ws: e3dge_local_query_sort_ws_ints(...) ints of
 * scratch (keys, order, ranks, bin counters); five launches (zero, key + count + rank, scan, scatter, the gather's backward on `order`).  Sums within a pixel
 * arrive in an order that varies from run to run, like the atomics' of the unsorted form. */
int64_t e3dge_local_query_sort_ws_ints(int batch, int64_t n_pts, int fh, int fw);
int e3dge_local_query_bwd_sorted(float* d_fmap_nhwc, float* d_pts, const float* d_out, int ld, int col_off, const float* pts,
                                 const float* calibs, const float* fmap_nhwc, int batch, int64_t n_pts, int channels, int fh, int fw,
                                 int32_t* ws, int64_t ws_ints, e3dge_stream_t stream);
int e3dge_pos_encoding(float* out, int ld, int col_off, const float* pts, int64_t n_pts, int n_freqs, e3dge_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Per-image reconstruction metrics in one pass (the scalars the sharded evaluation all-gathers):
 * replaces the MSE / L1 / PSNR / SSIM part of Loss.calc_2d_rec_loss (project/losses/builder.py:130-184; SSIM = kornia
 * ssim_loss with a 5x5 Gaussian window, sigma 1.5, reflect padding, max_val as given -- the reference passes its [-1,1]
 * images with max_val = 1).  pred, gt (batch, channels, H, W); scratch e3dge_image_metrics_scratch_floats() floats;
 * sums (batch, 4) = [sum (p-g)^2, sum |p-g|, sum clamp((1-ssim)/2, 0, 1), element count] per image.
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t e3dge_image_metrics_scratch_floats(int batch, int channels, int height, int width);
int e3dge_image_metrics(float* sums, float* scratch, const float* pred, const float* gt, int batch, int channels,
                        int height, int width, float max_val, e3dge_stream_t stream);
/* The eight columns builder.py:174-184 reports, from those sums (means over the whole batch tensor, as the reference's
 * losses are): row (8) = [loss_l2 = MSE, loss_id = 0, loss_lpips = 0, loss = l2_lambda * MSE, mae, PSNR of the images
 * rescaled to [0,1], SSIM = 1 - ssim_loss, ID_SIM = 1] -- the identity / LPIPS networks are outside the path and their
 * columns are what the reference reports with those lambdas at 0 (:145, :158-163). */
int e3dge_image_metric_row(float* row, const float* sums, int batch, float l2_lambda, e3dge_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Surface extraction, device half: replaces align_volume (project/utils/mesh_utils.py:17-44; called on the rendered
 * 128^3 SDF volume at volume_renderer.py:1706, before the CPU marching cubes).  volume, out (batch, height, width, depth,
 * channels), not aliased; xs (width), ys (height), zs (depth) = linspace(-1, 1, n) and coef (depth) = linspace(far / near,
 * 1, depth) as the caller's torch.linspace produced them.  out[b, y, x, z] = trilinear border-clamped lookup
 * (grid_sample, align_corners) at (xs[x] coef[z], ys[y] coef[z], zs[z]), or 1 where that point leaves [-1, 1]^3.
 * ---------------------------------------------------------------------------------------------------------------- */
int e3dge_align_volume(float* out, const float* volume, const float* xs, const float* ys, const float* zs, const float* coef,
                       int batch, int height, int width, int depth, int channels, e3dge_stream_t stream);

/* Layout self-test: runs a 32x32xK fp32-MFMA product with the fragment conventions the render kernel
 * relies on and writes it to c (32*32 floats, row-major) for the caller to compare with a @ b^T.
 * a: (32, k) row-major, b: (32, k) row-major, k multiple of 8, k <= 256. */
int e3dge_selftest_mfma(float* c, const float* a, const float* b, int k, e3dge_stream_t stream);
/* The same with one f16 MFMA (v_mfma_f32_32x32x16_f16) per 16 k on the hi halves of a, b; k multiple of 16. */
int e3dge_selftest_mfma16(float* c, const float* a, const float* b, int k, e3dge_stream_t stream);
/* The same for v_mfma_f32_16x16x32_f16 (c: 16x16, a, b: (16, k) row-major, k multiple of 32). */
int e3dge_selftest_mfma16x16(float* c, const float* a, const float* b, int k, e3dge_stream_t stream);
/* Weight-stationary split-f16 layers (csrc/siren_ws.hip): the weights of a 256 x 256 layer live in registers, activations in LDS.
 *   wimg   : e3dge_ws_image_bytes(n_layers) bytes, written by e3dge_ws_pack from fp32 weights (n_layers, 256, 256) [out][in]
 * (The round-3 study chain e3dge_ws_chain -- DESIGN.md 4.1d -- was removed in round 5.) */
int64_t e3dge_ws_image_bytes(int n_layers);
int e3dge_ws_pack(void* wimg, const float* weights, int n_layers, e3dge_stream_t stream);
/* One 256 x 256 linear layer over the rows of a matrix, weight-stationary split-f16 (the layers of Fuse_sft_MLP,
 * project/models/helper_modules/sft.py:84-109 and resnetfc.py:49-58 -- local_query.py chains nine of these):
 *   y[row, off_y + f] = post( sum_k W[f][k] pre(x[row, off_x + k]) + bias[f] + colw[f] pre(m[row, off_m]) + r1[row, off_r1 + f] + r2[row, off_r2 + f] )
 * pre = relu (pre_relu != 0) or identity; post: 0 identity, 1 leaky relu (slope), 2 the SFT fuse  D + w_fuse (D S + v)  with
 * D = r1, S = r2 and v the bracket without r1, r2.  wimg = e3dge_ws_pack(W, 1).  amax_in: amax buffer (E3DGE_AMAX_FLOATS) holding
 * max |x| over the tensor x comes from (operand scale; NULL: values of order 1), amax_out: NULL or the buffer that receives max |y|
 * (zero it first).  Pointers other than wimg, x, y may be NULL (colw and m together); row pitches in floats, any alignment.
 * ABI 12 -- what the BACKWARD of those layers needs (d input = d output @ W is the same layer on the image of W^T):
 *   input side:  x <- x (.) xmul[row, off_xmul + k] * x_scale  (xmul NULL: x * x_scale; x_scale 0 = unset = 1; amax_xmul = amax buffer of xmul)
 *   post 3:      y = v * (r1 > 0 ? 1 : slope) + r2     the backward through lrelu (slope) / relu (slope 0) whose output / input is r1
 *   post 4:      y = v + r1 (1 + w_fuse r2)            d dec of the SFT fuse: v + g (1 + w scale)
 * y may alias r2 (posts 0, 1, 3) -- every element is read by the thread that writes it. */
typedef struct E3dgeWsLinear {
    const void* wimg; const float* x; const float* amax_in; const float* bias; const float* colw; const float* m;
    const float* r1; const float* r2; float* y; float* amax_out;
    int64_t n_rows;
    int32_t ld_x, off_x, ld_m, off_m, ld_r1, off_r1, ld_r2, off_r2, ld_y, off_y;
    int32_t pre_relu, post;
    float slope, w_fuse;
    const float* xmul; const float* amax_xmul;
    int32_t ld_xmul, off_xmul;
    float x_scale;
    int32_t reserved;
} E3dgeWsLinear;
int e3dge_ws_linear(const E3dgeWsLinear* args, e3dge_stream_t stream);
/* out[row * ld_out + off_out] = a[row, :] . u + [gate[row * ld_gate + off_gate] > 0] (b[row, :] . v)   (gate NULL: 1); a, b (n_rows, 256), u, v (256).
 * The data gradient of ONE extra input column shared by two 256-wide layers -- the visibility-mask column of Fuse_sft_MLP's 513-wide input
 * (sft.py:103-109 / resnetfc.py:49-58: d x[:, 256] = de Ws[:, 256] + (dnet W0[:, 256]) [x[:, 256] > 0]). */
int e3dge_ws_rowdot2(float* out, int ld_out, int off_out, const float* a, const float* u, const float* b, const float* v, const float* gate,
                     int ld_gate, int off_gate, int64_t n_rows, e3dge_stream_t stream);
/* Parameter gradient of a fully connected layer of the local branch (round 5):  c (m, n) = sum over rows p of a[p, off_a : off_a + m]^T f(b[p, off_b : off_b + n]),
 * f = relu when relu_b != 0 -- autograd's `grad_output.t() @ input` for the nn.Linear layers of ResnetBlockFC (helper_modules/resnetfc.py:49-58)
 * and Fuse_sft_MLP (helper_modules/sft.py:84-110) in the stage-2 step (e3dge_full_runner.py:185-317), with the relu of the layer's input folded in.
 * a, b: fp32 rows of lda / ldb floats (4-byte aligned; any row pitch), amax_a / amax_b: amax buffers (e3dge_amax) bounding |a| / |b|;
 * ws: e3dge_wgrad_ws_floats(m, n, n_rows) floats (split-K partial blocks, folded in fixed order: bit-reproducible).  Split-f16 x 3, fp32 accumulate.
 * ABI 14 (round 6) -- what rides with the same pass over the rows (all optional, NULL / 0 = off):
 *   colsum (m):          sum_p a[p, off_a + i]                       autograd's `grad_output.sum(0)`, the layer's bias gradient
 *   xcol, ccol:          ccol[i * ld_ccol] = sum_p a[p, off_a + i] f(xcol[p * ld_xcol])   one more column of b that the block grid leaves out
 *   b_gap_at, b_gap:     columns [b_gap_at, b_gap_at + b_gap) of b's rows (and of c's) are skipped: c[:, j] for j >= b_gap_at comes from
 *                        b[:, off_b + j + b_gap] and lands in c[:, j + b_gap]; n counts the columns that ARE contracted.  b_gap_at: a
 *                        multiple of 256.  (Fuse_sft_MLP's 513-wide input: 256 features | visibility mask | 256 features -- n = 512,
 *                        b_gap_at = 256, b_gap = 1, xcol = b + off_b + 256, ccol = c + 256.) */
typedef struct E3dgeWgrad {
    const float* a; const float* amax_a; const float* b; const float* amax_b;
    float* c; float* ws;
    int64_t ws_floats, n_rows;
    int32_t lda, off_a, m, ldb, off_b, n, ldc, relu_b;
    const float* xcol; float* colsum; float* ccol;
    int32_t ld_xcol, ld_ccol, b_gap_at, b_gap;
} E3dgeWgrad;
int64_t e3dge_wgrad_ws_floats(int m, int n, int64_t n_rows);
int e3dge_wgrad(const E3dgeWgrad* args, e3dge_stream_t stream);
/* Accuracy self-test of the kernel's sine: y[i] = sin(x[i]) with the device routine the SIREN layers use. */
int e3dge_selftest_sin(float* y, const float* x, int n, e3dge_stream_t stream);
/* The alternative 13-op polynomial sine (kernels built with -DE3DGE_POLY_SINE use it). */
int e3dge_selftest_sin_poly(float* y, const float* x, int n, e3dge_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* E3DGE_HIP_H */
