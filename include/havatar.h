/*
 * havatar.h -- C ABI of libhavatar_hip.so: the MI355X (gfx950) implementation of HAvatar's
 * volumetric-rendering hot path and of its two StyleGAN2 custom ops.
 *
 * Every entry point takes plain device pointers + sizes + a HIP stream (void*, may be NULL for
 * the default stream).  Nothing here allocates, frees or synchronises: the caller owns all
 * memory and the launch is asynchronous on `stream` (reference convention: kernels go to
 * at::cuda::getCurrentCUDAStream() with no sync and no post-launch check,
 * model/op/fused_bias_act_kernel.cu:73,98; model/op/upfirdn2d_kernel.cu:215).
 * Return value: 0 on success, a hipError_t (>0) from the launch, or a negative HAV_E* code for
 * arguments this library refuses.
 *
 * Citations `path:line` are into the reference repository (XChenZ/havatar @ 2024_08_07).
 */
#ifndef HAVATAR_H
#define HAVATAR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAV_ABI_VERSION 6

#define HAV_EINVAL   (-1) /* bad size / null pointer / inconsistent arguments            */
#define HAV_EUNSUP   (-2) /* valid for the reference, not supported by this build        */

/* element types of the op entry points (reference dispatch: AT_DISPATCH_FLOATING_TYPES_AND_HALF,
 * fused_bias_act_kernel.cu:96, upfirdn2d_kernel.cu:311; bf16 is an addition) */
#define HAV_F32  0
#define HAV_F16  1
#define HAV_BF16 2
#define HAV_F64  3

int hav_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * fused_bias_act -- replaces `fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)`
 * (pybind: model/op/fused_bias_act.cpp:18-31; kernel: model/op/fused_bias_act_kernel.cu:18-105).
 *   out[i] = scale * f(x[i] + b[(i / step_b) % size_b]),  i in [0, size_x)
 *   act=1 linear, act=3 leaky-relu(alpha); grad=0 forward, grad=1 first derivative gated by the
 *   sign of `ref`, grad=2 -> 0.  `b`/`ref` may be NULL (= the reference's empty tensors).
 *   step_b = prod(x.shape[2:]), size_b = bias.numel().
 * ------------------------------------------------------------------------------------------ */
int hav_fused_bias_act(void* out, const void* x, const void* b, const void* ref,
                       int dtype, int act, int grad, float alpha, float scale,
                       int64_t size_x, int64_t step_b, int64_t size_b, void* stream);

/* ------------------------------------------------------------------------------------------
 * upfirdn2d -- replaces `upfirdn2d.upfirdn2d(input[major,in_h,in_w,minor], kernel[kh,kw],
 * up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)`
 * (pybind: model/op/upfirdn2d.cpp:17-31; kernels: model/op/upfirdn2d_kernel.cu:49-369).
 *   zero-stuff by `up`, pad (negative pads crop), correlate with the FLIPPED FIR, decimate by
 *   `down`.  out is [major,out_h,out_w,minor] with
 *   out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y   (upfirdn2d_kernel.cu:237-240).
 *   The FIR is always float32 in this ABI (the reference casts it to the input dtype; taps are
 *   accumulated in float32 either way, upfirdn2d_kernel.cu:115-116).
 * ------------------------------------------------------------------------------------------ */
int hav_upfirdn2d(void* out, const void* in, const float* kernel, int dtype,
                  int64_t major, int in_h, int in_w, int minor, int kh, int kw,
                  int up_x, int up_y, int down_x, int down_y,
                  int pad_x0, int pad_x1, int pad_y0, int pad_y1, void* stream);
/* out_h/out_w for the arguments above (host-side helper, no device work) */
int hav_upfirdn2d_out_size(int in_h, int in_w, int kh, int kw, int up_x, int up_y,
                           int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                           int pad_y1, int* out_h, int* out_w);

/* The style network of a generator, `self.style(z)` = PixelNorm -> n x (EqualLinear + fused leaky-ReLU) (model/styleUnet.py:53-55,
 * EqualLinear.forward, _style_mlp), for widths <= 64 in one launch.  blob: per layer [Din][Dout] scale * W^T then [Dout] lr_mul * b. */
int hav_style_mlp(float* out, const float* z, const float* blob, int n_layers, int B, int D0, int D, float slope, float gain, void* stream);

/* ------------------------------------------------------------------------------------------
 * StyleGAN2 block glue -- the tiny-op chains around every modulated convolution of the tri-plane encoders / the upsampler
 * (model/styleUnet.py: ModulatedConv2d.forward :196-254 non-fused branch, NoiseInjection :306-310, FusedLeakyReLU;
 * model/op/fused_act.py).  Not separate native ops in the reference (it runs them as ~10 ATen launches per layer); here each
 * chain is one launch.  float32 only.
 *
 * hav_style_demod:   s[b,i] = <style[b,:], mod_w[i,:]> + mod_b[i]                      (EqualLinear, scale folded into mod_w/mod_b)
 *                    d[b,o] = rsqrt(sum_i s[b,i]^2 * wsq[i,o] + eps)   if d_out != NULL  (wsq[i,o] = sum_k (scale W[o,i,k])^2)
 * hav_styled_epilogue: y = leaky_relu((x * d[b,c] + (*noise_weight) * noise[b?,p]) + bias[c], slope) * gain, every product and
 *                    sum rounded separately in that order (== the unfused ATen sequence); d, noise, bias nullable;
 *                    noise is [B,HW] if noise_batched else [HW].
 * ------------------------------------------------------------------------------------------ */
int hav_style_demod(float* s_out /*[B,Cin]*/, float* d_out /*[B,Cout] or NULL*/, const float* style /*[B,D]*/,
                    const float* mod_w /*[Cin,D]*/, const float* mod_b /*[Cin] or NULL*/, const float* wsq /*[Cin,Cout] or NULL*/,
                    float eps, int B, int D, int Cin, int Cout, void* stream);
/* The same for every modulated convolution of a generator in ONE launch (their style vectors depend only on the latent, which is
 * known before the first convolution runs): `layers_dev` is a device-resident table, layer l reads styles[b, style_index, :] and
 * owns workgroups [first_block, first_block + hav_style_demod_blocks(Cout, d_out != NULL)); total_blocks = the sum; max_cin = the
 * largest Cin of the table (sizes the LDS). */
typedef struct HavStyleDemodLayer {
    const float* mod_w;   /* [Cin, D] */
    const float* mod_b;   /* [Cin] or NULL */
    const float* wsq;     /* [Cin, Cout] or NULL (no demodulation) */
    float* s_out;         /* [B, Cin] */
    float* d_out;         /* [B, Cout] or NULL */
    int32_t Cin, Cout;
    int32_t style_index;
    int32_t first_block;
} HavStyleDemodLayer;
int hav_style_demod_blocks(int Cout, int demodulate);
int hav_style_demod_batched(const HavStyleDemodLayer* layers_dev, int n_layers, int total_blocks, int max_cin,
                            const float* styles /*[B,n_styles,D]*/, float eps, int B, int n_styles, int D, void* stream);
int hav_styled_epilogue(float* out, const float* x /*[B,C,HW]*/, const float* d /*[B,C] or NULL*/, const float* noise,
                        const float* noise_weight /*device scalar or NULL*/, const float* bias /*[C] or NULL*/, float slope,
                        float gain, int B, int C, int64_t HW, int noise_batched, void* stream);

/* ToRGB of a StyleGAN2 generator level (reference model/styleUnet.py:602-628: ModulatedConv2d(kernel_size=1, demodulate=False) :165-297,
 * + bias + skip) in one pass over the activations:
 *   out[b,o,p] = sum_i ((scale * W[o,i]) * s[b,i]) * x[b,i,p] + bias[o] + skip[b,o,p]
 * x [B,Cin,HW], W [Cout,Cin] (the 1x1 parameter), s [B,Cin] or NULL (no modulation), bias [Cout] or NULL, skip [B,Cout,HW] or NULL (the
 * caller's dwt(upsample(iwt(skip))) or upsample(skip)); out [B,Cout,HW] may not alias x.  fp32 FMA chains; Cout in {3, 12} (the two ToRGB
 * widths), Cin <= 1024, HW % 4 == 0; anything else returns HAV_EUNSUP (the caller keeps its ATen route). */
int hav_torgb(float* out, const float* x, const float* W, const float* s, const float* bias, const float* skip, float scale,
              int B, int Cout, int Cin, int64_t HW, void* stream);

/* Demodulation factors of a modulated convolution under autograd (training path):
 *   q[i,o] = scale^2 sum_k W[o,i,k]^2 (written to `q`, [Cin,Cout], kept for the backward);  d[b,o] = rsqrt(sum_i s[b,i]^2 q[i,o] + eps)
 * (model/styleUnet.py:214-227, factored form) and its backward: gs [B,Cin] = d loss / d s through d only (the caller adds the direct
 * term), gW [Cout,Cin,KK] = d loss / d W through d only; gq_scratch [Cin,Cout].  W is the raw parameter [Cout,Cin,k,k] (KK = k*k). */
int hav_demod_fwd(float* d, float* q, const float* s, const float* W, float scale, float eps, int B, int Cin, int Cout, int KK, void* stream);
int hav_demod_bwd(float* gs, float* gW, float* gq_scratch, const float* gd, const float* s, const float* d, const float* q, const float* W,
                  float scale, int B, int Cin, int Cout, int KK, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tri-plane gather with gradients (training path) -- sample_from_triplane_new + its autograd (utils/util.py:359-406:
 * two F.grid_sample(bilinear, zeros, align_corners=True) + stack): plane 0 at (q.x,q.y), plane 1 at (q.z,q.y),
 * feat[i, 2c+p].  Planes are CHANNELS-LAST [2,B,H,W,C] here (the caller permutes; autograd carries the permutation).
 * q [n,3] are the box-warped coordinates, query i belongs to frame i / n_per_b.
 * bwd: dplanes_cl += scatter (caller zero-fills), dq [n,3] (nullable) = d loss / d q.
 * ------------------------------------------------------------------------------------------ */
int hav_triplane_gather_fwd(float* feat /*[n,2C]*/, const float* planes_cl, const float* q, int64_t n, int64_t n_per_b,
                            int B, int H, int W, int C, void* stream);
int hav_triplane_gather_bwd(float* dplanes_cl, float* dq, const float* dfeat /*[n,2C]*/, const float* planes_cl, const float* q,
                            int64_t n, int64_t n_per_b, int B, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-path field ops (autograd statement of the march).  The radiance MLP between them stays on
 * rocBLAS; these replace what surrounds it, forward and backward, with one launch each.
 *
 * hav_field_inputs_*: pts [n,3] -> X [n, 2C+48] = cat(tri-plane features, positional encoding) of the DEFORMED point:
 *   Deformation_Field_new.forward (model/Skinning_Field.py:70-98: two bones {identity, inv_T}, trilinear border sampling of the
 *   skinning volume vol [2,D,D,D] at the skin-box-warped p_i, normalised blend) -> UniformBoxWarp_new of the NeRF box
 *   (utils/util.py:232-236) -> sample_from_triplane_new (:359-406; planes channels-last [2,B,H,W,C], feature 2c+plane) and
 *   Embedder.embed (model/network/embedder.py:32-61; 8 octaves, cos as sin(x + pi/2)) -> cat (model/nerf_model.py:104).
 *   Query i belongs to frame i / n_per_b (inv_T [B,4,3]).
 *   bwd: dplanes_cl += scatter of dX (nullable; caller zero-fills), dvol [2,D,D,D] += d loss / d volume through the blend
 *   weights (nullable; caller zero-fills).  pts and inv_T are data (no gradient), as in the reference's training step.
 * hav_composite_*: volume_render_radiance_field(act_feat=False) + cumprod_exclusive (utils/nerf_util.py:4-73):
 *   rf [n_rays,S,CH+1] (density last; the first n_sigmoid channels pass through a sigmoid, :45-46), z [n_rays,S], rd [n_rays,3],
 *   noise [n_rays,S] (already scaled by radiance_field_noise_std; nullable), bg [n_rays,3] (nullable) ->
 *   rgb [n_rays,CH], acc [n_rays], weights [n_rays,S], depth [n_rays].  S <= 64 (HAV_EUNSUP above).
 *   bwd: d_rf [n_rays,S,CH+1] from d_rgb (required), d_acc / d_weights / d_depth (nullable = zero).
 * ------------------------------------------------------------------------------------------ */
typedef struct HavFieldParams {
    int64_t n;            /* queries                                                            */
    int64_t n_per_b;      /* queries per frame                                                  */
    int32_t B, H, W, C;   /* planes [2,B,H,W,C]                                                 */
    int32_t D;            /* skinning volume resolution                                         */
    float   nerf_scale[3], nerf_trans[3];
    float   skin_scale[3], skin_trans[3];
} HavFieldParams;

int hav_field_inputs_fwd(float* X, const HavFieldParams* p, const float* pts, const float* inv_T, const float* vol,
                         const float* planes_cl, void* stream);
/* The same rows as bf16 [n, 2C+48] (round to nearest even) for hav_mlp_train_fwd_xbf16 / _bwd_xbf16: the radiance MLP's bf16 kernels round
 * their fp32 input rows exactly so, hence identical results with half the bytes between the two kernels.  C <= 64, C even. */
int hav_field_inputs_fwd_bf16(void* Xb, const HavFieldParams* p, const float* pts, const float* inv_T, const float* vol,
                              const float* planes_cl, void* stream);
int hav_field_inputs_bwd(float* dplanes_cl, float* dvol, const float* dX, const HavFieldParams* p, const float* pts,
                         const float* inv_T, const float* vol, const float* planes_cl, void* stream);
/* the same for queries laid out [B][rays][samples_per_ray] whose neighbouring rays are neighbouring pixels of an image row (the training patch,
 * dataloader/dataloader.py:93-121): the scatter merges the taps of 16 neighbouring rays per depth.  Falls back to hav_field_inputs_bwd's order
 * when the rays per batch element are not a multiple of 16. */
int hav_field_inputs_bwd_rows(float* dplanes_cl, float* dvol, const float* dX, const HavFieldParams* p, const float* pts,
                              const float* inv_T, const float* vol, const float* planes_cl, int samples_per_ray, void* stream);
/* Bit-reproducible form of hav_field_inputs_bwd (ABI 6): the same gradients, summed as 64-bit fixed-point integers with integer atomics
 * (integer addition is associative: the result does not depend on the order the atomics land in; float atomics do).  dx_amax: the
 * HAV_ABSMAX_WORDS words of hav_absmax(dX) (the planes' scale; the volume taps are buffered and scaled by their own maximum).  scratch:
 * hav_field_inputs_bwd_fixed_scratch_bytes(p) bytes (accumulators + buffered taps; zeroed by the call).  plane_ch <= 64. */
int64_t hav_field_inputs_bwd_fixed_scratch_bytes(const HavFieldParams* p);
int hav_field_inputs_bwd_fixed(float* dplanes_cl, float* dvol, const float* dX, const void* dx_amax, void* scratch, const HavFieldParams* p,
                               const float* pts, const float* inv_T, const float* vol, const float* planes_cl, void* stream);
int hav_composite_fwd(float* rgb, float* acc, float* weights, float* depth, const float* rf, const float* z, const float* rd,
                      const float* noise, const float* bg, int64_t n_rays, int S, int CH, int n_sigmoid, void* stream);
int hav_composite_bwd(float* d_rf, const float* d_rgb, const float* d_acc, const float* d_weights, const float* d_depth,
                      const float* rf, const float* z, const float* rd, const float* noise, const float* bg, int64_t n_rays,
                      int S, int CH, int n_sigmoid, void* stream);
/* Importance resampling between the two passes under autograd (added within ABI 6: additive) -- replaces the ATen chain of
 * model/nerf_trainer.py:166-170 (z_vals_mid, sample_pdf, z_samples.detach(), cat with z_vals[::2], sort) and utils/nerf_util.py:76-117
 * (sample_pdf: +1e-5, sum, cumsum, stratified or linspace u, searchsorted(right), clamped gathers, the 1e-5 denominator floor):
 *   z2 [n, ceil(S_c/2) + S_f] = sort(cat(z[:, ::2], z_samples)),  z_samples [n, S_f] (nullable output)
 * z, weights [n, S_c]: the coarse pass' depths and compositing weights (weights[:, 1:-1] are the ones used).  zeta [n, S_f] = the
 * raw torch.rand draw of utils/nerf_util.py:95 (u = k * (1/S_f) + zeta * (1/S_f - 1e-6), the library never draws random numbers
 * here), NULL = det=True (u = linspace(0, 1, S_f)).  Sum / cumulative sum run in index order (what the oracle restates); every
 * other operation is rounded separately like the ATen statement: bit-exact against orc_resample_depths_f32.  No gradient (the
 * reference detaches).  S_c >= 3, S_c <= 128, ceil(S_c/2) + S_f <= 128 (else HAV_EUNSUP). */
int hav_resample_depths(float* z2, float* z_samples, const float* z, const float* weights, const float* zeta, int64_t n_rays,
                        int S_c, int S_f, void* stream);
/* EqualLinear without activation under autograd (added within ABI 6: additive) -- replaces, per modulation layer of a ModulatedConv2d and
 * per optimisation step, the ATen launches of model/styleUnet.py:128-162 (`F.linear(input, self.weight * self.scale, bias=self.bias *
 * self.lr_mul)`: two scalar products over the parameters + addmm forward; their adjoints, two GEMMs and a column sum backward):
 *   fwd: y [B,out] = x [B,in] . fl(W [out,in] * scale)^T + fl(bias [out] * lr_mul)      (bias nullable)
 *   bwd: dx [B,in] = dy . fl(W * scale);  dW [out,in] = scale * dy^T . x;  dbias [out] = lr_mul * sum_b dy      (each output nullable)
 * fp32 throughout, sums in a fixed order (bit-reproducible).  B <= 8, in <= 4096, in * out <= 2^24 (else HAV_EUNSUP). */
int hav_equal_linear_fwd(float* y, const float* x, const float* W, const float* bias, float scale, float lr_mul, int B, int in_dim,
                         int out_dim, void* stream);
int hav_equal_linear_bwd(float* dx, float* dW, float* dbias, const float* dy, const float* x, const float* W, float scale, float lr_mul,
                         int B, int in_dim, int out_dim, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3x3, stride-1, zero-padded convolution of the StyleGAN blocks with the block's glue fused in -- replaces, at
 * inference, ModulatedConv2d + NoiseInjection + FusedLeakyReLU of a StyledConv (model/styleUnet.py:165-297,300-310,565-599) and
 * EqualConv2d + FusedLeakyReLU of a ConvLayer (:90-125,326-368), i.e. one MIOpen convolution plus 1-3 elementwise launches:
 *   y[b,o,p] = act( d[b,o] * sum_{i,ky,kx} W[o,i,ky,kx] * (s[b,i] * x[b,i,p+(ky-1,kx-1)]) + (*noise_weight) * noise[p] + bias[o] ) * gain
 * x, y NCHW float32; s [B,Cin], d [B,Cout], noise [H*W] or [B,H*W] (noise_batched), noise_weight (device scalar), bias [Cout] are
 * all nullable; act != 0 applies leaky-ReLU(slope) * gain, act == 0 leaves the sum as is (gain ignored).
 * Arithmetic: implicit GEMM on v_mfma_f32_32x32x16_f16 with split operands (x and W as hi + lo fp16, three products, fp32
 * accumulation): fp32-class results; the fp16 range limit of the split (|s x| < 65504) is lifted by `in_amax` below.
 * `packed` = hav_conv3x3_pack(W [Cout,Cin,3,3], wmul), hav_conv3x3_packed_bytes(Cout, Cin) bytes; wmul is folded into the weights
 * (EqualConv2d / ModulatedConv2d scale 1/sqrt(9 Cin)).  Needs Cin % 16 == 0, Cout % 64 == 0, H % 4 == 0, W % 32 == 0
 * (HAV_EUNSUP otherwise: the caller keeps its MIOpen route).
 * ------------------------------------------------------------------------------------------ */
int64_t hav_conv3x3_packed_bytes(int Cout, int Cin);
int hav_conv3x3_pack(void* packed, const float* w, int Cout, int Cin, float wmul, void* stream);
/* hav_conv3x3_pack_t: the blob of the DATA GRADIENT of the convolution with filters w [Cout_w,Cin_w,3,3] -- the convolution with
 * Cin_w output and Cout_w input channels and filters W'[i][o][t] = w[o][i][8 - t] -- packed straight from w (hav_conv3x3_packed_bytes(Cin_w,
 * Cout_w) bytes; Cout_w % 16 == 0, Cin_w % 32 == 0). */
int hav_conv3x3_pack_t(void* packed, const float* w, int Cout_w, int Cin_w, float wmul, void* stream);
/* Maps with too few 64 x 128 output tiles to fill the GPU (32^2) split the channel range over 2-4 workgroups and add the slices up in a
 * second, deterministic pass: hav_conv3x3_scratch_bytes() bytes of caller scratch (0: not needed; NULL: never split). */
int64_t hav_conv3x3_scratch_bytes(int B, int Cin, int Cout, int H, int W);
int hav_conv3x3_split(float* y, const float* x, const void* packed, const float* s, const float* d, const float* noise,
                      const float* noise_weight, const float* bias, float slope, float gain, int act, int noise_batched, int B,
                      int Cin, int Cout, int H, int W, void* scratch, const void* in_amax, void* stream);
/* The same with stride 2 (the down-sampling ConvLayer / ConvBlock of the encoders after its Blur: EqualConv2d(stride 2, padding 0),
 * model/styleUnet.py:326-368): x [B,Cin,Hin,Win] -> y [B,Cout,Hout,Wout], Hout = (Hin + 2 pad - 3) / 2 + 1, pad 0 or 1 (zeros); same packed
 * weights, same fused terms (noise / d / bias indexed by the OUTPUT map), same arithmetic and range control.  Needs Cin % 16 == 0,
 * Cout % 64 == 0, Hout % 4 == 0, Wout % 32 == 0 (HAV_EUNSUP otherwise).  Maps with too few output tiles to fill the GPU split the channel
 * range over 2-8 workgroups per tile (deterministic second pass): hav_conv3x3s2_scratch_bytes() bytes of caller scratch (0: not needed;
 * NULL: never split).  ABI 4: the `scratch` argument. */
int64_t hav_conv3x3s2_scratch_bytes(int B, int Cin, int Cout, int Hin, int Win, int pad);
int hav_conv3x3s2_split(float* y, const float* x, const void* packed, const float* s, const float* d, const float* noise,
                        const float* noise_weight, const float* bias, float slope, float gain, int act, int noise_batched, int B,
                        int Cin, int Cout, int Hin, int Win, int pad, void* scratch, const void* in_amax, void* stream);
/* Range control for inputs far from 1 (gradients, 1e5-sized activations): hav_absmax leaves HAV_ABSMAX_WORDS partial maxima of |x| (bit patterns of
 * non-negative floats, one per slice of x; no atomics, nothing to initialise) in a caller buffer of HAV_ABSMAX_WORDS * 4 bytes,
 * 16-byte aligned; passed as `in_amax` (NULL: off) the convolution folds them and scales its input by the power of two that brings
 * max |x| * max_i |s[b,i]| (a bound on what is split; max |x| without a modulation) to [512, 1024) before the fp16 split and scales
 * the result back -- exact, no host round trip, no overflow whatever the sizes of x and s.  Without it inputs below 2^-3 gradually
 * lose the low part of the split to fp16 subnormals (absolute operand error 2^-25), inputs below 6e-8 vanish and |s x| >= 65504
 * becomes Inf.  hav_gemm_split (`in_amax`) and both operands of hav_conv3x3_wgrad (`g_amax`, `x_amax`) take the same words. */
#define HAV_ABSMAX_WORDS 256
int hav_absmax(void* out_bits, const float* x, int64_t n, void* stream);
/* Development aid of the graphed training step (havatar_amd/native/conv.py::_trace, HAVATAR_NAN_TRACE=2): *flag |= 1 iff one of the n
 * words of x, read as float32, is Inf or NaN.  One launch, no allocation, no temporary: the caller owns `flag` (4 bytes, zeroed by the
 * caller before the launch or the graph replay), so the check adds nothing to a captured graph's memory pool.  ABI 5. */
int hav_debug_nonfinite(void* flag, const void* x, int64_t n, void* stream);

/* Weight gradient of the same convolution (training): gw[o,i,ky,kx] = sum_{b,y,x} g[b,o,y,x] * x[b,i,y+ky-1,x+kx-1] on the split-fp16
 * matrix path (Cin % 32 == 0, Cout % 64 == 0, W % 16 == 0).  scratch: hav_conv3x3_wgrad_scratch_bytes() bytes (K-split partial sums);
 * g_amax, x_amax: HAV_ABSMAX_WORDS words from hav_absmax(g) / hav_absmax(x) or NULL (range control of either operand: g is
 * gradient-sized, x may be anything). */
int64_t hav_conv3x3_wgrad_scratch_bytes(int B, int Cin, int Cout, int H, int W);
int hav_conv3x3_wgrad(float* gw /*[Cout,Cin,3,3]*/, const float* g /*[B,Cout,H,W]*/, const float* x /*[B,Cin,H,W]*/, void* scratch,
                      const void* g_amax, const void* x_amax, int B, int Cin, int Cout, int H, int W, void* stream);
/* The same with the x operand modulated (xs [B,Cin] * x, NULL: plain) and the result multiplied by out_mul: the gradient of the raw
 * ModulatedConv2d / EqualConv2d weight PARAMETER (out_mul = the layer's 1/sqrt(9 Cin) scale) given g = dL/d(conv output). */
int hav_conv3x3_wgrad_mod(float* gw, const float* g, const float* x, const float* xs, float out_mul, void* scratch, const void* g_amax,
                          const void* x_amax, int B, int Cin, int Cout, int H, int W, void* stream);
/* Weight gradient of the STRIDE-2 3x3 layers (training; ABI 6):  out[m,n,ky,kx] = out_mul * sum_{b,y,x} ss[b,m] * S[b,m,y,x] * L[b,n,2y+ky,2x+kx]
 * with S [B,M,H,W], L [B,N,2H+1,2W+1] (M % 64 == 0, N % 32 == 0, W % 16 == 0), on the split-fp16 matrix path.  It is the weight gradient of
 *   - the down-sampling ConvLayer (Blur -> EqualConv2d stride 2 padding 0 -> FusedLeakyReLU, reference model/styleUnet.py:326-368 under
 *     autograd):  S = dL/d(conv output) [B,Cout,H,W], L = the blurred input [B,Cin,2H+1,2W+1]  ->  gw [Cout,Cin,3,3];
 *   - the up-sampling StyledConv's transposed convolution (F.conv_transpose2d stride 2, model/styleUnet.py:214-231):  S = the layer's input
 *     x [B,Cin,H,W] with ss = its modulation s [B,Cin], L = dL/d(conv_transpose2d output) [B,Cout,2H+1,2W+1], transpose_out = 1  ->
 *     the parameter's layout [Cout,Cin,3,3].
 * (the reference runs both through ATen's convolution_backward: MIOpen igemm_wrw + NHWC transposes).  ss NULL: plain S.  scratch:
 * hav_conv3x3s2_wgrad_scratch_bytes() bytes; s_amax / l_amax: hav_absmax words of S / L or NULL (power-of-two range control). */
int64_t hav_conv3x3s2_wgrad_scratch_bytes(int B, int M, int N, int H, int W);
int hav_conv3x3s2_wgrad(float* gw, const float* S, const float* L, const float* ss, float out_mul, int transpose_out, void* scratch,
                        const void* s_amax, const void* l_amax, int B, int M, int N, int H, int W, void* stream);
/* Backward glue of one fused convolution block y = act(d * conv(s * x, W) + noise_weight * noise + bias) * gain under autograd (what
 * ATen runs as ~20 small launches per layer: reference model/styleUnet.py:165-310 + model/op/fused_act.py:20-52 under autograd):
 *   hav_conv_block_bwd   g = dL/dy, y = the block's output ->
 *                        gc [B,Cout,HW] = d * g_pre  (g_pre = g * gain * act'(y): the gradient of the raw convolution output),
 *                        gd [B,Cout] = sum_p g_pre * conv_raw,  gbias [Cout] = sum_{b,p} g_pre,  gnw [1] = sum g_pre * noise
 *                        (each output nullable; d / noise / bias nullable as in the forward; the pre-activation is recovered from y,
 *                        so act != 0 needs slope > 0 and gain > 0); sums_scratch: B * Cout * 3 floats; HW % 4 == 0
 *   hav_mod_input_bwd    gxs = dL/d(s * x) in gx_inout ->  gs [B,Cin] = sum_p x * gxs,  gx_inout = s * gxs */
int hav_conv_block_bwd(float* gc, float* gd, float* gbias, float* gnw, float* sums_scratch, const float* g, const float* y, const float* d,
                       const float* noise, const float* noise_weight, const float* bias, float slope, float gain, int act, int noise_batched,
                       int B, int Cout, int64_t HW, void* stream);
int hav_mod_input_bwd(float* gx_inout, float* gs, const float* x, const float* s, int B, int Cin, int64_t HW, void* stream);
/* The up-sampling StyledConv of the StyleGAN blocks (model/styleUnet.py:236-243: conv_transpose2d(x * s, W, stride 2) -> 4x4 FIR with
 * padding (1,1) -> demodulation -> noise -> bias -> leaky-ReLU) in two launches:
 *   hav_gemm_split     y[b, m, n] = sum_k A[m, k] * (s[b, k] * x[b, k, n])    split-fp16 matrix path, fp32-class (K % 32 == 0, N % 128 == 0;
 *                      A packed by hav_gemm_pack from a row-major [M, K] matrix with `wmul` folded in; s nullable; in_amax: words of
 *                      hav_absmax(x) or NULL, range control as in hav_conv3x3_split)
 *   hav_upconv_finish  col [B, Cout*9, H*W] (row 9 o + 3 ky + kx: the product above with A[9 o + t, i] = W[i, o, t]) -> y [B, Cout, 2H, 2W]:
 *                      stride-2 scatter, FIR (`fir4x4`: the 16 taps as upfirdn2d takes them) and the epilogue
 *                      act(d[b,o] * . + noise_weight * noise + bias[o]) * gain, every term nullable as in hav_conv3x3_split. */
int64_t hav_gemm_packed_bytes(int M, int K);
int hav_gemm_pack(void* packed, const float* w /*[M,K]*/, int M, int K, float wmul, void* stream);
int hav_gemm_split(float* y /*[B,M,N]*/, const float* x /*[B,K,N]*/, const void* packed, const float* s /*[B,K] or NULL*/,
                   const void* in_amax, int B, int M, int K, int N, void* stream);
int hav_upconv_finish(float* y, const float* col, const float* fir4x4, const float* d, const float* noise, const float* noise_weight,
                      const float* bias, float slope, float gain, int act, int noise_batched, int B, int Cout, int H, int W,
                      void* stream);

/* Haar analysis / synthesis of SWGAN_unet's wavelet-domain skip path (model/styleUnet.py HaarTransform / InverseHaarTransform: four
 * upfirdn2d calls each, + cat / + three adds) as one pass each, bit-identical to the four-call sequence.
 *   hav_haar_dwt : in [B,C,H,W] -> out [B,4C,H/2,W/2], channel = band*C + c, bands ll, lh, hl, hh   (H even, W % 8 == 0)
 *   hav_haar_idwt: in [B,4C,H,W] -> out [B,C,2H,2W]                                                  (W % 4 == 0)
 * k4x2x2: the four 2x2 kernels exactly as the upfirdn2d calls receive them (the synthesis takes ll, -lh, -hl, hh). */
int hav_haar_dwt(float* out, const float* in, const float* k4x2x2, int B, int C, int H, int W, void* stream);
int hav_haar_idwt(float* out, const float* in, const float* k4x2x2, int B, int C, int H, int W, void* stream);
/* The skip path of ToRGB, `skip = dwt(upsample(iwt(skip)))` (model/styleUnet.py:476-480: InverseHaarTransform -> Upsample (4x4 FIR, up 2,
 * pad (2, 1)) -> HaarTransform), as one pass: in [B,4C,H,W] -> out [B,4C,2H,2W], bit-identical to hav_haar_idwt -> hav_upfirdn2d -> hav_haar_dwt.
 * ki4x2x2 / kd4x2x2: the synthesis / analysis kernels as the two calls above receive them; fir4x4: Upsample's kernel (gain folded in).  W even. */
int hav_haar_up2(float* out, const float* in, const float* ki4x2x2, const float* fir4x4, const float* kd4x2x2, int B, int C, int H, int W,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Trilinear x2 up-sampling of a [N,C,D,H,W] float32 volume and its adjoint -- nn.Upsample(scale_factor=2, mode='trilinear',
 * align_corners=False), the first stage of every UpConv3DBlock of the skinning-volume decoder
 * (model/network/voxel_encoder.py:183-210).  NC = N*C; out / dout are [NC, 2D, 2H, 2W].  Gather form both ways (no atomics).
 * ------------------------------------------------------------------------------------------ */
int hav_upsample3d_2x_fwd(float* out, const float* in, int64_t NC, int D, int H, int W, void* stream);
int hav_upsample3d_2x_bwd(float* din, const float* dout, int64_t NC, int D, int H, int W, void* stream);
/* Patch matrix of a 3^3 / padding-1 Conv3d on a small cubic volume and its adjoint (the first layers of VolumeDecoder, model/network/voxel_encoder.py:
 * 183-210, as matrix products): col [27 C, R^3], col[(i, t)][p] = x[i][p + off(t)] or 0; dx[i][q] = sum_t dcol[(i, t)][q - off(t)].  x / dx [C, R, R, R]. */
int hav_im2col3d(float* col, const float* x, int C, int R, void* stream);
int hav_col2im3d(float* dx, const float* dcol, int C, int R, void* stream);

/* ------------------------------------------------------------------------------------------
 * Radiance MLP of the training path on the bf16 matrix cores (BASELINE config 5) -- replaces, under
 * autograd, the five nn.Linear calls of ConditionalTriplaneNeRFModel_multiRender_split_view.forward (model/nerf_model.py:104-117):
 *   X [n,176] = cat(tri-plane features, positional encoding)  ->  rf [n,68] = [rgb(3) | feature(64) | alpha(1)]
 * and their backward with the activations RECOMPUTED from X (nothing is saved between forward and backward but X itself):
 *   d_rf [n,68]  ->  dX [n,176] (nullable), gradients of all ten parameter tensors in nn.Linear layout.
 * Operands (weights and activations) are rounded to bf16, accumulation / inputs / outputs / master weights are fp32.
 * The weight gradients are summed over fixed query slices in a fixed order: bit-reproducible, no atomics.
 * Scratch (caller-allocated, contents irrelevant): `ops` hav_mlp_train_ops_bytes(n) bytes, `partial` hav_mlp_train_partial_bytes(n).
 * `blob` = hav_mlp_train_pack(weights), hav_mlp_train_blob_bytes() bytes; re-pack whenever the weights change.
 * ------------------------------------------------------------------------------------------ */
typedef struct HavMlpGrads {           /* device pointers, float32, nn.Linear layouts (same order as HavMlpWeights) */
    float* W1; float* b1; float* W2; float* b2; float* Wa; float* ba; float* Wf; float* bf; float* Wc; float* bc;
} HavMlpGrads;
struct HavMlpWeights;
int64_t hav_mlp_train_blob_bytes(void);
int64_t hav_mlp_train_ops_bytes(int64_t n);
int64_t hav_mlp_train_partial_bytes(int64_t n);
int hav_mlp_train_pack(void* blob, const struct HavMlpWeights* w, void* stream);
int hav_mlp_train_fwd(float* rf, const float* X, const void* blob, int64_t n, void* stream);
/* accumulate != 0: grads += (autograd's accumulation into .grad); 0: grads = */
int hav_mlp_train_bwd(float* dX, const HavMlpGrads* grads, int accumulate, const float* X, const float* d_rf, const void* blob,
                      void* ops, void* partial, int64_t n, void* stream);
/* the same with X given as bf16 rows [n,176] (hav_field_inputs_fwd_bf16); dX stays fp32 */
int hav_mlp_train_fwd_xbf16(float* rf, const void* Xb, const void* blob, int64_t n, void* stream);
int hav_mlp_train_bwd_xbf16(float* dX, const HavMlpGrads* grads, int accumulate, const void* Xb, const float* d_rf, const void* blob,
                            void* ops, void* partial, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray march -- replaces Trainer.predict_and_render_radiance (model/nerf_trainer.py:120-201) and
 * everything it calls: ray sampling (:129-141), Deformation_Field_new.forward
 * (model/Skinning_Field.py:70-98), sample_pts_triplane_feat (model/nerf_model.py:88-99),
 * Embedder.embed (model/network/embedder.py:32-61), the radiance MLP (model/nerf_model.py:101-117),
 * volume_render_radiance_field / cumprod_exclusive (utils/nerf_util.py:4-73), sample_pdf
 * (utils/nerf_util.py:76-117) and the sort/merge of the fine z list (model/nerf_trainer.py:170).
 * One launch covers ALL rays of the call: the reference's 4096-ray chunk loop
 * (model/nerf_trainer.py:66-71) only bounds activation memory and is not needed here.
 * ------------------------------------------------------------------------------------------ */
typedef struct HavRenderParams {
    int32_t B;            /* frames in this call (per-frame inv_T and tri-plane)               */
    int32_t R;            /* rays per frame                                                    */
    int32_t ray_stride;   /* floats per ray in `rays`; [0:3]=o [3:6]=d [6]=near [7]=far        */
    int32_t S_c;          /* cfg.nerf.<mode>.num_coarse                                        */
    int32_t S_f;          /* cfg.nerf.<mode>.num_fine (0: coarse pass only)                    */
    int32_t perturb;      /* cfg.nerf.<mode>.perturb != 0                                      */
    float   noise_std;    /* cfg.nerf.<mode>.radiance_field_noise_std                          */
    int32_t plane_res;    /* tri-plane H = W (128)                                             */
    int32_t plane_ch;     /* tri-plane channels per plane (64)                                 */
    int32_t vol_res;      /* skinning volume D = H = W (64)                                    */
    float   nerf_scale[3], nerf_trans[3];   /* UniformBoxWarp_new of the NeRF box (util.py:232) */
    float   skin_scale[3], skin_trans[3];   /* ... of the skinning box (nerf_trainer.py:29-34)  */
    uint64_t seed;        /* key of the on-device xi/zeta/eps streams used when the rand pointers are NULL */
    uint64_t rng_offset;  /* counter base of the on-device streams (advance per call)          */
    int32_t mlp_mode;     /* HAV_MLP_SPLIT_BF16 (0), HAV_MLP_F32 (1), HAV_MLP_SPLIT_F16 (2) or HAV_MLP_SPLIT_F16_MX (3) */
    int32_t flags;        /* HAV_FLAG_* bit mask (0 = library defaults); unknown bits -> HAV_EINVAL */
    uint64_t* rng_counter; /* optional DEVICE counter: the call uses rng_offset + *rng_counter and increments the counter
                            * on the stream afterwards, so a hipGraph replay of a captured call draws fresh jitter    */
    void*    workspace;    /* optional DEVICE scratch, hav_render_workspace_bytes() bytes: the fine pass then re-uses the
                            * radiance-field values of the even coarse samples that the merged list repeats
                            * (model/nerf_trainer.py:170) instead of evaluating them again: 80 evaluations per ray, not 112.
                            * NULL / too small: every merged sample is evaluated, like the reference does (so does
                            * HAV_FLAG_FINE_RECOMPUTE).  Used by both split modes, with and without jitter; in the fp16
                            * mode by calls that decline the coarse outputs (HavRenderOut) only.                          */
    uint64_t workspace_bytes;
    float*   dbg_zfine;    /* optional DEVICE [B*R, S_fp] (tests): the call also dumps the merged, sorted fine depths here  */
    uint32_t* status;      /* optional DEVICE word; the call ORs HAV_STATUS_* bits into it on the stream (the caller zeroes it) */
    int32_t  grid_blocks;  /* ABI 6.  0 = one persistent workgroup per compute unit (the default).  > 0: at most this many (rounded up to
                            * a multiple of 8, one share per XCD), which leaves compute units to kernels of another stream.  Results do
                            * not depend on it (the blocks of 32 rays are dealt to fewer workgroups, nothing else changes).  Measured
                            * (tools/pipeline_probe.py, profiles/r05_pipeline_probe.txt): 208 units 7.81 ms vs 256 units 7.01 ms -- the
                            * kernel is power-limited -- and the next frame's encoders beside it are no faster than behind it.   */
    int32_t  reserved0;    /* 0 */
} HavRenderParams;

/* HavRenderParams.flags */
#define HAV_FLAG_PAIR_KERNEL     1   /* force the ray-pair kernel (otherwise only used when num_coarse > 67); A/B runs and tests */
#define HAV_FLAG_FINE_CACHE      2   /* the fine-pass cache where a workspace is offered: now the default, kept for callers that set it */
#define HAV_FLAG_FINE_RECOMPUTE  4   /* never use the fine-pass cache                                                               */
#define HAV_FLAG_NO_FP16_GUARD   8   /* HAV_MLP_SPLIT_F16 only: skip the range guard below (the caller vouches for the range)       */
#define HAV_FLAGS_ALL           15

/* HavRenderParams.status bits */
#define HAV_STATUS_FP16_FALLBACK 1u  /* the fp16-split kernel declined (range guard) and the bf16-split kernel rendered the call  */

/* How the two dense layers run on the matrix cores.  All three pass every parity test at the path's tolerance:
 *  HAV_MLP_SPLIT_BF16: each fp32 operand is split EXACTLY into 3 bf16 parts (24 significant bits: not narrower than the reference's
 *                      fp32) and the 6 leading bf16 x bf16 products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (dropped
 *                      terms ~2^-23 relative per product; measured worst case 2^-21.7, see HAV_MLP_SPLIT_F16_MX); the Python layer's
 *                      default in round 4, the fp16 range guard's stand-in since;
 *  HAV_MLP_F32:        v_mfma_f32_32x32x2_f32, bit-for-bit an fp32 fmaf chain. */
#define HAV_MLP_SPLIT_BF16 0
#define HAV_MLP_F32        1
#define HAV_MLP_SPLIT_F16_MX 3    /* ABI 5.  hi + lo fp16 as in HAV_MLP_SPLIT_F16 for the three leading partial products, PLUS the three terms of
                                  * order 2^-22 that mode drops (hi.tail, tail.hi, lo.lo; tail = v - hi - lo, exact) from one block-scaled
                                  * 4- / 6-bit matrix instruction each per 64 k (v_mfma_scale_f32_32x32x64_f8f6f4; factors rounded to 2-4
                                  * significant bits against per-(row, 32 k) / per-(query, 32 k) power-of-two scales: <= 2^-25 of the
                                  * product).  Operands carry hi + lo + tail = the fp32 value itself.  MEASURED per-product error against fp64
                                  * (one-hot inputs, tests/test_render_gpu.py::test_arithmetic_modes_of_the_dense_layers_against_fp64_products):
                                  * <= 2^-21.3 at worst with the SAME rms as the bf16 triple split (whose worst case is 2^-21.7): six
                                  * roundings of partial products, not the operands, set both -- the class of the bf16 triple split at
                                  * 0.65 of its matrix time.  Same fp16 RANGE guard and bf16 stand-in
                                  * as HAV_MLP_SPLIT_F16.  The Python layer's default since round 5 (HAVATAR_MLP=mx). */
#define HAV_MLP_SPLIT_F16  2     /* each fp32 operand = hi + lo fp16 (both rounded to nearest: <= 2^-22 relative -- the size of the
                                  * fp32 accumulation error of a 128-term dot product), 3 products on v_mfma_f32_32x32x16_f16:
                                  * half the matrix time and two thirds of the LDS of the bf16 triple split -- 22-bit operands, i.e.
                                  * NARROWER than fp32: opt-in (HAVATAR_MLP=half in the Python layer), ~25 % faster.
                                  * RANGE: fp16 tops out at 65504.  hav_triplane_prepare derives a rigorous bound on
                                  * every value this mode converts to fp16 (weights, relu(h1), relu(h2)) from the weights and the
                                  * projected planes -- |h1_u| <= |b1_u| + sum_k |W1pe_uk| + max_texel |P0_u| + max_texel |P1_u|,
                                  * |h2_v| <= |b2_v| + sum_u |W2_vu| bound(h1_u) -- and stores the verdict next to the planes; when
                                  * the bound reaches 60000 (or is not finite) the fp16 kernel declines ON THE DEVICE and the
                                  * bf16-split kernel (fp32 range) renders the call instead: same launch sequence, no host sync,
                                  * hipGraph-safe; HavRenderParams.status reports it.  Values below 2^-14 use fp16 subnormals for
                                  * their low part: absolute operand error <= 2^-25 there instead of 2^-22 relative. */

/* Bytes of scratch with which hav_render_rays can skip the repeated even coarse samples for these parameters
 * (0: not applicable -- no fine pass, exact-f32 mode, or num_coarse > 67). */
int64_t hav_render_workspace_bytes(const HavRenderParams* p);

/* Radiance MLP parameters in nn.Linear layout (model/nerf_model.py:46-51), device pointers. */
typedef struct HavMlpWeights {
    const float* W1; const float* b1;   /* layers_xyz.0  [128,176],[128]   */
    const float* W2; const float* b2;   /* layers_xyz.1  [128,128],[128]   */
    const float* Wa; const float* ba;   /* fc_alpha      [1,128],[1]       */
    const float* Wf; const float* bf;   /* fc_rgbFeat    [64,128],[64]     */
    const float* Wc; const float* bc;   /* fc_rgb        [3,64],[3]        */
} HavMlpWeights;

/* Bytes of the packed, MFMA-fragment-ordered weight blob the ray march consumes. */
int64_t hav_mlp_blob_bytes(void);
/* Pack nn.Linear-layout weights into `blob` (device, hav_mlp_blob_bytes() bytes). Re-run whenever
 * the weights change (training step / load_state_dict). */
int hav_mlp_pack(void* blob, const HavMlpWeights* w, void* stream);

/* Per-frame tri-plane preparation.  Input: NCHW [2,B,64,H,W] (Trainer.model_coarse.triPlane_embeddings,
 * model/nerf_model.py:85-86).  Output (device, hav_triplane_prepared_bytes(B,H,W) bytes: the planes + a 2-KB trailer holding the
 * per-unit maxima and the fp16 range verdict described at HAV_MLP_SPLIT_F16): 128 floats per texel of
 * [2,B,H,W] in which every texel has already been multiplied by the 128x64 block of layers_xyz.0 that reads this
 * plane's channels (bilinear interpolation and the first linear layer commute), stored in the ray-march kernel's
 * accumulator order.  Needs the packed blob (hav_mlp_pack) of the CURRENT weights: re-run when planes OR weights change.
 * The buffer's internal arrangement (groups of 4 x-adjacent texels, 16-byte pieces interleaved across the group so that
 * neighbouring rays read one 64-byte segment) is private to the library: only hav_render_rays reads it.  W % 4 == 0. */
int64_t hav_triplane_prepared_bytes(int B, int H, int W);
int hav_triplane_prepare(float* dst, const float* src_nchw, const void* mlp_blob, int B, int C, int H, int W,
                         void* stream);

typedef struct HavRenderOut {      /* all device pointers, float32; fine pointers unused if S_f==0.  With a fine pass the three
                                    * *_coarse pointers may ALL be NULL ("not wanted": Trainer.forward only uses the fine maps then) */
    float* rgb_coarse;   /* [B,R,67]  rgb(3, sigmoid, + background) | feature(64)                 */
    float* depth_coarse; /* [B,R]                                                                 */
    float* acc_coarse;   /* [B,R]                                                                 */
    float* weights_max;  /* [B,R]  max_i w_i of the LAST pass run (model/nerf_trainer.py:195,200) */
    float* rgb_fine;     /* [B,R,67]                                                              */
    float* depth_fine;   /* [B,R]                                                                 */
    float* acc_fine;     /* [B,R]                                                                 */
} HavRenderOut;

/*
 * rays      [B,R,ray_stride]                    (ray_batch; viewdirs, if present, are ignored:
 *                                                the only consumer is dead code, nerf_trainer.py:146-150)
 * bg        [B,R,3] or NULL                     (background_prior)
 * inv_T     [B,4,3]                             (inv_head_T: rows 0-2 = M, row 3 = tau)
 * planes    prepared planes, hav_triplane_prepared_bytes() bytes (hav_triplane_prepare; layout private to the library)
 * skin_vol  [2,D,H,W]                           (canonical_W[0], shared by the batch)
 * mlp_blob  hav_mlp_pack output
 * t_rand    [B,R,S_c] or NULL                   (xi  = torch.rand at nerf_trainer.py:138)
 * u_rand    [B*R,S_f] or NULL                   (zeta= torch.rand at utils/nerf_util.py:95)
 * noise_c   [B*R,S_c] / noise_f [B*R,S_fp] or NULL (eps = torch.randn at utils/nerf_util.py:49-56,
 *                                                UNSCALED standard normals; multiplied by noise_std here)
 * With perturb!=0 and a NULL rand pointer the values come from on-device counter-based streams (a murmur-finalizer hash for
 * the stratified jitter, Philox4x32-10 + Box-Muller for the density noise).
 */
int hav_render_rays(const HavRenderParams* p, const float* rays, const float* bg,
                    const float* inv_T, const float* planes_prepared, const float* skin_vol,
                    const void* mlp_blob, const float* t_rand, const float* u_rand,
                    const float* noise_c, const float* noise_f, const HavRenderOut* out,
                    void* stream);

/* Test hook (ABI 5): one dense layer of the radiance MLP without its activation, evaluated by the matrix routine hav_render_rays uses
 * in `mlp_mode` -- y[n,128] = W . x + b with layer 1: x [n,48] = the positional-encoding inputs of layers_xyz.0 (its columns 128..175,
 * |x| <= 1), layer 2: x [n,128] >= 0 (layers_xyz.1 sees relu outputs; model/nerf_model.py:104-109).  Lets a test hold each arithmetic
 * mode against an fp64 product directly.  mlp_blob: hav_mlp_pack output. */
int hav_debug_mlp_layer(float* y, const float* x, const void* mlp_blob, int mlp_mode, int layer, int64_t n, void* stream);

/* Name of the ray-march kernel variant hav_render_rays launches for these parameters ("hav_march_blk_kernel<RNG, PREC, CACHE>",
 * the same decision function the launch uses), written NUL-terminated into buf[len].  coarse_outputs = 0 if the call declines
 * the coarse pass's composited outputs (rgb/depth/acc_coarse NULL in HavRenderOut); injected_rand != 0 if any of
 * t_rand / u_rand / noise_c / noise_f is given.  Returns 0, or HAV_EINVAL (NULL arguments / buffer too small). */
int hav_render_variant_name(const HavRenderParams* p, int coarse_outputs, int injected_rand, char* buf, int len);

/* ------------------------------------------------------------------------------------------
 * get_rays on device (next-1; reference: dataloader/data_util.py:28-56 +
 * dataloader/dataloader.py:174-177): rays[H*W,8] = (o3, d3 normalised, near, far) from
 * intr=(fx,fy,cx/W,cy/H), c2w[3,4] row-major, for pixel rows [y0,y1).
 * ------------------------------------------------------------------------------------------ */
int hav_gen_rays(float* rays, int H, int W, const float intr[4], const float c2w[12],
                 float near, float far, int y0, int y1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HAVATAR_H */
