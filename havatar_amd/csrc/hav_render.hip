// hav_render.hip -- the fused ray march for gfx950 (MI355X): ray sampling -> skinning-field lookup ->
// tri-plane gather -> positional encoding -> radiance MLP on the matrix cores -> alpha compositing ->
// inverse-CDF resampling -> second pass, in ONE kernel, with nothing but the ray inputs and the
// composited outputs touching HBM.
//
// Replaces Trainer.predict_and_render_radiance and its callees (reference file:line in include/havatar.h
// and next to each step below).  Design (DESIGN.md has the long form and the measurements behind each choice):
//
//  * Two mappings of rays to a wave64 share one per-tile evaluator (sample_eval): a "tile" is 32 radiance-field queries, lanes
//    j = lane&31 are the queries, the two half-waves h = lane>>5 split each query's hidden units, its tri-plane gather (64 of the
//    128 projected channels each), its PE octaves (4h..4h+3) and its two skinning bones.
//      - BLOCK kernel (hav_march_blk_kernel, the default): a wave owns 32 consecutive rays and walks them through the sample
//        index together -- neighbouring rays hit the same texels, and compositing is a per-lane recurrence (see its header).
//      - PAIR kernel (hav_march_f32_kernel, num_coarse > 67 or HAV_MARCH=pair): a wave owns two rays; the 16 lanes of a DPP row
//        are consecutive samples of one ray, the transmittance product is a DPP row scan + a scalar carry.
//  * MLP in "transposed" form  H^T[unit][query] = W[unit][k] . X^T[k][query].  With this orientation the accumulator registers
//    of layer l ARE the B operands of layer l+1 (lane = its query's column; register r of a 32-row tile holds rows
//    (r&3)+8(r>>2)+4h, exactly the k-group an MFMA step consumes from the two half-waves), so activations never leave
//    registers and never cross lanes between layers.  Only weights move: pre-permuted once into that k order ("fragment
//    order", hav_mlp_pack) and LDS-resident.  Three arithmetic modes (HavRenderParams.mlp_mode; PREC template parameter):
//      - split-operand fp16 (HAV_MLP_SPLIT_F16, what Trainer / bench.py run): every fp32 operand = hi + lo fp16 (22 bits), three
//        partial products on v_mfma_f32_32x32x16_f16 with fp32 accumulation (mfma_split2h); operands beyond fp16's range are
//        caught by the range guard on the device, and the bf16 split renders the call instead (status HAV_STATUS_FP16_FALLBACK);
//      - split-operand bf16 (HAV_MLP_SPLIT_BF16, mode 0 of the ABI and the guard's fallback): hi + mid + lo bf16 exactly, six
//        partial products on v_mfma_f32_32x32x16_bf16 (mfma_split3) -- fp32-sgemm-class results over fp32's whole range;
//      - exact fp32: v_mfma_f32_32x32x2_f32 (an fmaf chain).
//  * Measured on gfx950 (tools/ubench/mfma_overlap2.hip, round 4): plain VALU and LDS instructions DO issue beside the matrix pipe --
//    a partner wave's v_fma_f32 / v_max / v_cvt stream keeps 95 % of its pace beside a saturating MFMA stream, ~6 fillers fit behind
//    each v_mfma_f32_32x32x16 of the same wave -- and only PACKED fp32 VALU (v_pk_fma/mul/add_f32) serialises with it.  (Rounds 1-3
//    read "VALU does not overlap with MFMA" off microbenchmarks whose C fillers hipcc had packed into v_pk_fma_f32.)  What bounds the
//    kernel is the socket power cap: it runs at 2.0-2.2 of 2.4 GHz, and cycle savings come back as lower clock (docs/history/DESIGN_r1-r4.md 3.13).  The
//    matrix work -- the most power-hungry part -- was cut algebraically:
//      - Layer 1's 128 tri-plane columns are folded into the planes once per frame: bilinear interpolation is linear,
//        W1f (sum_tap w_tap texel_tap) = sum_tap w_tap (W1f texel_tap).  hav_triplane_prepare projects every texel
//        through W1f (a 128x64 GEMM over 32768 texels = 0.5 GFLOP per frame, vs 2.8 TFLOP for the march) into
//        128-channel planes stored in accumulator order; the kernel then accumulates 8 taps straight into the layer-1
//        accumulators (packed FMAs) and only the 48 PE columns go through the matrix cores.
//      - fc_rgb o fc_rgbFeat has no activation in between (model/nerf_model.py:110-111): the 3 rgb rows fold into 3
//        rows over the 128 hidden units; with alpha that is 4 dot products per query on the VALU.
//      - the 64 feature channels are LINEAR in h2 and only consumed through the compositing sum:
//        sum_s w_s (Wf h2_s + bf) = Wf (sum_s w_s h2_s) + bf sum_s w_s.  The kernel composites the 128 hidden units and
//        applies fc_rgbFeat once per RAY.
//    Per tile that leaves 352 f32 MFMAs (or 264 bf16 ones) instead of 736 for the literal network.
//  * One persistent workgroup (8 waves, 2 per SIMD) per CU; XCD-aware ray assignment keeps each XCD's L2 on one
//    horizontal band of the image / projected planes.
//  * Experiment knobs (results are WRONG or timing-only when set): HAV_ABLATE bit mask, HAV_STAGGER, -DHAV_PROFILE (phase
//    timers, tools/phase_profile.sh), HAV_MARCH=pair|blk, HAV_MLP=f32|split.
#include "hav_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

// ------------------------------------------------------------------------------------------------
// packed weight blob (float offsets).  [0, LDS_FLOATS) is copied verbatim into LDS by every workgroup.
// ------------------------------------------------------------------------------------------------
#define HAV_HID 128
#define HAV_PC 64                      // channels per plane
#define HAV_IN (2 * HAV_PC + 48)       // 176
#define KPE_STEPS 24                   // MFMA k-steps of the PE part of layer 1 (48 columns)
#define K2_STEPS (HAV_HID / 2)         // 64
#define OFF_W1PE 0                                  // [24][4][64]  layer-1 PE columns, fragment order
#define OFF_W2 (OFF_W1PE + KPE_STEPS * 4 * 64)      // [64][4][64]  layer 2, fragment order
#define OFF_W4 (OFF_W2 + K2_STEPS * 4 * 64)         // [64][2][4]   head rows rgb0 rgb1 rgb2 alpha per (k-step, half-wave)
#define OFF_WFT (OFF_W4 + K2_STEPS * 2 * 4)         // [128][64]    fc_rgbFeat weight transposed
#define LDS_FLOATS (OFF_WFT + HAV_HID * 64)         // 31232 floats = 122 KB
#define OFF_B1 LDS_FLOATS                           // [128]
#define OFF_B2 (OFF_B1 + 128)                       // [128]
#define OFF_B4 (OFF_B2 + 128)                       // [4]  folded rgb biases, alpha bias
#define OFF_BF (OFF_B4 + 4)                         // [64] fc_rgbFeat bias
#define OFF_W1F (OFF_BF + 64)                       // [2][128][64] layer-1 plane columns per plane, rows in accumulator order
#define OFF_WFF (OFF_W1F + 2 * 128 * 64)             // [64 k-steps][2 row tiles][64 lanes] fc_rgbFeat, fragment order (block kernel epilogue)
// split-operand (3 x bf16) fragments for v_mfma_f32_32x32x16_bf16: per (16-wide k chunk, row tile, part hi/mid/lo, lane) 8 bf16
#define OFF_A1S (OFF_WFF + K2_STEPS * 2 * 64)       // [3 chunks][4][3][64][4 dwords]  layer-1 PE columns
#define OFF_A2S (OFF_A1S + 3 * 4 * 3 * 64 * 4)      // [8 chunks][4][3][64][4 dwords]  layer 2
#define OFF_W4S (OFF_A2S + 8 * 4 * 3 * 64 * 4)      // copy of W4 so that [A1S | A2S | W4S] is one contiguous LDS image
#define LDS3_FLOATS (3 * 4 * 3 * 64 * 4 + 8 * 4 * 3 * 64 * 4 + K2_STEPS * 2 * 4)    // 34304 dwords = 134 KB
// split-operand (2 x fp16) fragments for v_mfma_f32_32x32x16_f16: per (16-wide k chunk, row tile, part hi/lo, lane) 8 halves
#define OFF_A1H (OFF_W4S + K2_STEPS * 2 * 4)        // [3 chunks][4][2][64][4 dwords]
#define OFF_A2H (OFF_A1H + 3 * 4 * 2 * 64 * 4)      // [8 chunks][4][2][64][4 dwords]
#define OFF_W4H (OFF_A2H + 8 * 4 * 2 * 64 * 4)      // copy of W4: [A1H | A2H | W4H] is one contiguous LDS image
#define OFF_AFH (OFF_W4H + K2_STEPS * 2 * 4)        // [8 chunks][2 row tiles][2][64][4 dwords]  fc_rgbFeat, same k order as layer 2
#define HAV_TAPPAIR 1      // the two x-adjacent taps of a row are loaded interleaved, piece by piece
#define HAV_TG 2           // log2 of the texel group of the prepared-plane layout (4 x-adjacent texels) of plane 0 (x,y) ...
#define HAV_TG1 2          // ... and of plane 1 (z,y)
#define LDSH_FLOATS (3 * 4 * 2 * 64 * 4 + 8 * 4 * 2 * 64 * 4 + K2_STEPS * 2 * 4 + 8 * 2 * 2 * 64 * 4)    // 31232 dwords = 122 KB
// fp16 range guard (include/havatar.h, HAV_MLP_SPLIT_F16): [128] |b1_u| + sum_k |W1pe_uk| per hidden unit, in the accumulator
// order hu of the prepared planes | [1] max |w| over every weight the fp16 mode converts (inf if one is not finite) | pad
#define OFF_RNG (OFF_AFH + 8 * 2 * 2 * 64 * 4)
// "fp16 x 2 + MX" mode (HAV_MLP_SPLIT_F16_MX, PREC 3): the fp16 hi / lo fragments above ([A1H | A2H | W4H]) evaluate the three leading partial
// products of every fp32 product; what they leave out -- lo.lo, hi.tail, tail.hi, all of order 2^-22 of the product -- comes from ONE
// block-scaled 4- / 6-bit matrix instruction per term and 64 k (v_mfma_scale_f32_32x32x64_f8f6f4).  Per (k group of 4 chunks = 32 slots per
// lane, row tile) one record of MX_G_DWORDS: AH4 [64 lanes][4 dwords] fp4(e2m1) of the weights' hi parts | AT4 [64][4] fp4 of their tails
// w - hi - lo | AL6 [64][4] + [64][2] fp6(e2m3) of their lo parts | SC [64] E8M0 block scales, bytes (hi, tail, lo, 0).  Slot s = 8 c + e of a
// lane is element e of chunk c of the group in the fp16 fragments' k order; layer 1 has one group of 3 chunks (slots 24-31 are zero).
#define MX_G_DWORDS 960
#define MX_GROUPS 12                                  // layer 1: row tiles 0-3 | layer 2: k group 0, row tiles 0-3 | k group 1, row tiles 0-3
#define OFF_MX (OFF_RNG + 132)
#define LDSX_FP16 (OFF_AFH - OFF_A1H)                // 23040 dwords: [A1H | A2H | W4H]
#define LDSX_FLOATS (LDSX_FP16 + MX_GROUPS * MX_G_DWORDS)      // 34560 dwords = 135 KB
#define BLOB_FLOATS (OFF_MX + MX_GROUPS * MX_G_DWORDS)
#define HAV_FP16_LIMIT 60000.0f
#ifndef HAV_PRIO_DENSE
#define HAV_PRIO_DENSE 3          // issue priority of a wave behind its gather (sample_eval); -DHAV_PRIO_DENSE=0: A/B builds without it
#endif

extern "C" int64_t hav_mlp_blob_bytes(void) { return (int64_t)BLOB_FLOATS * 4; }

// hidden unit held by half-wave h in accumulator register r of row-tile mp
__host__ __device__ inline int acc_row(int mp, int r, int h) { return 32 * mp + (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ void __launch_bounds__(256) mlp_pack_kernel(float* __restrict__ blob, HavMlpWeights w)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= OFF_MX) return;                // (the MX records behind it are written by mlp_pack_mx_kernel)
    float v = 0.f;
    if (e < OFF_W2) {                   // PE column consumed by half-wave h at k-step t: 128 + 24h + t
        const int l = e & 63, m = (e >> 6) & 3, t = e >> 8;
        v = w.W1[(32 * m + (l & 31)) * HAV_IN + 2 * HAV_PC + 24 * (l >> 5) + t];
    } else if (e < OFF_W4) {
        const int q = e - OFF_W2;
        const int l = q & 63, m = (q >> 6) & 3, ks = q >> 8;
        v = w.W2[(32 * m + (l & 31)) * HAV_HID + acc_row(ks >> 4, ks & 15, l >> 5)];
    } else if (e < OFF_WFT) {
        const int q = e - OFF_W4;
        const int c = q & 3, hh = (q >> 2) & 1, ks = q >> 3;
        const int col = acc_row(ks >> 4, ks & 15, hh);
        if (c < 3) {               // fc_rgb o fc_rgbFeat folded (no activation in between): (Wc Wf)[c][col]
            double s = 0.0;
            for (int k = 0; k < 64; ++k) s += (double)w.Wc[c * 64 + k] * (double)w.Wf[k * HAV_HID + col];
            v = (float)s;
        } else v = w.Wa[col];
    } else if (e < OFF_B1) {
        const int q = e - OFF_WFT;
        v = w.Wf[(q & 63) * HAV_HID + (q >> 6)];
    } else if (e < OFF_B2) v = w.b1[e - OFF_B1];
    else if (e < OFF_B4) v = w.b2[e - OFF_B2];
    else if (e < OFF_BF) {
        const int c = e - OFF_B4;
        if (c < 3) {
            double s = (double)w.bc[c];
            for (int k = 0; k < 64; ++k) s += (double)w.Wc[c * 64 + k] * (double)w.bf[k];
            v = (float)s;
        } else v = w.ba[0];
    } else if (e < OFF_W1F) v = w.bf[e - OFF_BF];
    else if (e >= OFF_RNG) {
        const int hu = e - OFF_RNG;
        if (hu < 128) {                 // |PE value| <= 1 + 1e-7 (the second value of a pair is cos x - eps sin x)
            const int u = acc_row((hu & 63) >> 4, hu & 15, hu >> 6);
            float sacc = fabsf(w.b1[u]);
            for (int k = 0; k < 48; ++k) sacc += 1.0001f * fabsf(w.W1[u * HAV_IN + 2 * HAV_PC + k]);
            v = sacc;
        } else v = 0.f;                 // OFF_RNG + 128 (max |w|) is written by mlp_wmax_kernel
    } else if (e >= OFF_AFH) {
        const int q = e - OFF_AFH;
        const int d = q & 3, l = (q >> 2) & 63, part = (q >> 8) & 1, m = (q >> 9) & 1, ch = q >> 10;
        const int row = 32 * m + (l & 31), hh = l >> 5;
        uint32_t word = 0;
        for (int t = 0; t < 2; ++t) {
            const float wv = w.Wf[row * HAV_HID + acc_row(ch >> 1, 8 * (ch & 1) + 2 * d + t, hh)];
            const __half hi = __float2half_rn(wv);
            const __half lo = __float2half_rn(wv - __half2float(hi));
            word |= (uint32_t)__half_as_ushort(part ? lo : hi) << (16 * t);
        }
        v = __uint_as_float(word);
    } else if (e >= OFF_W4H) {
        const int q = e - OFF_W4H;
        const int c = q & 3, hh = (q >> 2) & 1, ks = q >> 3;
        const int col = acc_row(ks >> 4, ks & 15, hh);
        if (c < 3) {
            double s = 0.0;
            for (int k = 0; k < 64; ++k) s += (double)w.Wc[c * 64 + k] * (double)w.Wf[k * HAV_HID + col];
            v = (float)s;
        } else v = w.Wa[col];
    } else if (e >= OFF_A1H) {
        // same element order as the bf16 fragments; 2-way fp16 split, round-to-nearest at both levels: wv = hi + lo (+ <= 2^-22 |wv|)
        const bool l2 = e >= OFF_A2H;
        const int q = e - (l2 ? OFF_A2H : OFF_A1H);
        const int d = q & 3, l = (q >> 2) & 63, part = (q >> 8) & 1, m = (q >> 9) & 3, ch = q >> 11;
        const int row = 32 * m + (l & 31), hh = l >> 5;
        uint32_t word = 0;
        for (int t = 0; t < 2; ++t) {
            const int el = 2 * d + t;
            const float wv = l2 ? w.W2[row * HAV_HID + acc_row(ch >> 1, 8 * (ch & 1) + el, hh)]
                                : w.W1[row * HAV_IN + 2 * HAV_PC + 24 * hh + 8 * ch + el];
            const __half hi = __float2half_rn(wv);
            const __half lo = __float2half_rn(wv - __half2float(hi));
            word |= (uint32_t)__half_as_ushort(part ? lo : hi) << (16 * t);
        }
        v = __uint_as_float(word);
    } else if (e >= OFF_W4S) {
        const int q = e - OFF_W4S;
        const int c = q & 3, hh = (q >> 2) & 1, ks = q >> 3;
        const int col = acc_row(ks >> 4, ks & 15, hh);
        if (c < 3) {
            double s = 0.0;
            for (int k = 0; k < 64; ++k) s += (double)w.Wc[c * 64 + k] * (double)w.Wf[k * HAV_HID + col];
            v = (float)s;
        } else v = w.Wa[col];
    } else if (e >= OFF_A1S) {
        // dword d of lane l holds elements e0 = 2d, e1 = 2d+1 of the lane's 8 k-values; k-value e of half-wave hh in chunk:
        //   layer 1: PE column 128 + 24 hh + 8 chunk + e ;  layer 2: hidden unit acc_row(chunk>>1, 8 (chunk&1) + e, hh)
        const bool l2 = e >= OFF_A2S;
        const int q = e - (l2 ? OFF_A2S : OFF_A1S);
        const int d = q & 3, l = (q >> 2) & 63, part = (q >> 8) % 3, m = ((q >> 8) / 3) & 3, ch = (q >> 8) / 12;
        const int row = 32 * m + (l & 31), hh = l >> 5;
        uint32_t word = 0;
        for (int t = 0; t < 2; ++t) {
            const int el = 2 * d + t;
            const float wv = l2 ? w.W2[row * HAV_HID + acc_row(ch >> 1, 8 * (ch & 1) + el, hh)]
                                : w.W1[row * HAV_IN + 2 * HAV_PC + 24 * hh + 8 * ch + el];
            // exact 3-way split, round-to-nearest-even at each level: wv = hi + mid + lo
            float rem = wv;
            uint32_t bits = 0;
            for (int pp = 0; pp <= part; ++pp) {
                const uint32_t u = __float_as_uint(rem);
                const uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
                bits = r >> 16;
                rem = rem - __uint_as_float(bits << 16);
            }
            word |= bits << (16 * t);
        }
        v = __uint_as_float(word);
    } else if (e >= OFF_WFF) {
        const int q = e - OFF_WFF;
        const int l = q & 63, m = (q >> 6) & 1, ks = q >> 7;
        v = w.Wf[(32 * m + (l & 31)) * HAV_HID + acc_row(ks >> 4, ks & 15, l >> 5)];
    } else {                            // W1F[p][hu][c] = W1[unit(hu)][2c + p]; hu = h*64 + (m*16 + r)
        const int q = e - OFF_W1F;
        const int c = q & 63, hu = (q >> 6) & 127, p = q >> 13;
        const int u = hu & 63;
        v = w.W1[acc_row(u >> 4, u & 15, hu >> 6) * HAV_IN + 2 * c + p];    // feature index = 2*channel + plane (nerf_model.py:99)
    }
    blob[e] = v;
}

// max |w| over the weights the fp16 split converts: layer-1 PE columns, layer 2, fc_rgbFeat (the head rows stay fp32 on the VALU).
// A weight that is not finite, or not below the limit, makes the result +inf (fmaxf would drop a NaN).
__global__ void __launch_bounds__(256) mlp_wmax_kernel(float* __restrict__ blob, HavMlpWeights w)
{
    __shared__ float red[256];
    float m = 0.f;
    auto see = [&](float x) { const float ax = fabsf(x); m = (ax < HAV_FP16_LIMIT) ? fmaxf(m, ax) : __builtin_inff(); };
    for (int i = threadIdx.x; i < HAV_HID * 48; i += 256) see(w.W1[(i / 48) * HAV_IN + 2 * HAV_PC + i % 48]);
    for (int i = threadIdx.x; i < HAV_HID * HAV_HID; i += 256) see(w.W2[i]);
    for (int i = threadIdx.x; i < 64 * HAV_HID; i += 256) see(w.Wf[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) blob[OFF_RNG + 128] = red[0];
}

// OCP MX element formats (no Inf / NaN encodings): round to nearest even, saturating.  EB / MB exponent / mantissa bits, BIAS.
template <int EB, int MB, int BIAS>
__device__ unsigned int mx_quant(float v)
{
    const unsigned int sign = v < 0.f ? 1u : 0u;
    const float a = fabsf(v);
    if (!(a > 0.f)) return sign << (EB + MB);
    constexpr int emin = 1 - BIAS, emax = (1 << EB) - 1 - BIAS;
    int e;
    (void)frexpf(a, &e);
    e -= 1;                                  // a = 1.x * 2^e
    if (e < emin) e = emin;                  // the subnormal grid has the spacing of the smallest normal binade
    const float ulp = ldexpf(1.0f, e - MB);
    float q = rintf(a / ulp) * ulp;          // (both scalings are exact; rintf rounds half to even)
    const float maxv = ldexpf((float)((2 << MB) - 1), emax - MB);
    if (q > maxv) q = maxv;
    unsigned int ef, m;
    if (q < ldexpf(1.0f, emin)) { ef = 0u; m = (unsigned int)(q / ldexpf(1.0f, emin - MB)); }
    else {
        int e2;
        (void)frexpf(q, &e2);
        e2 -= 1;
        ef = (unsigned int)(e2 + BIAS);
        m = (unsigned int)(q / ldexpf(1.0f, e2 - MB)) - (1u << MB);
    }
    return (sign << (EB + MB)) | (ef << MB) | m;
}
// block exponent: the smallest s with maxabs / 2^s <= top (the format's largest value); E8M0 byte = s + 127
__device__ int mx_block_exp(float maxabs, float top)
{
    if (!(maxabs > 0.f)) return -126;
    int k;
    const float f = frexpf(maxabs / top, &k);          // maxabs / top = f * 2^k, f in [0.5, 1)
    int s = (f == 0.5f) ? k - 1 : k;
    return s < -126 ? -126 : (s > 127 ? 127 : s);
}
// One thread per (record, lane): the lane's 32 weights of the record's k group in fragment order, split exactly as the fp16 fragments are
// (hi = rn16(w), lo = rn16(w - hi)), tail = w - hi - lo (exact), each kind quantised against its own block scale.
__global__ void __launch_bounds__(64) mlp_pack_mx_kernel(float* __restrict__ blob, HavMlpWeights w)
{
    const int G = blockIdx.x, l = threadIdx.x;
    const int m = G & 3, hh = l >> 5, row = 32 * m + (l & 31);
    const bool l2 = G >= 4;
    const int g = l2 ? (G - 4) >> 2 : 0;
    float hi[32], lo[32], tl[32];
    float mh = 0.f, ml = 0.f, mt = 0.f;
    for (int sl = 0; sl < 32; ++sl) {
        const int c = sl >> 3, el = sl & 7;
        float wv = 0.f;
        if (l2) { const int ch = 4 * g + c; wv = w.W2[row * HAV_HID + acc_row(ch >> 1, 8 * (ch & 1) + el, hh)]; }
        else if (c < 3) wv = w.W1[row * HAV_IN + 2 * HAV_PC + 24 * hh + 8 * c + el];
        const float h_ = __half2float(__float2half_rn(wv));
        const float r_ = wv - h_;
        const float l_ = __half2float(__float2half_rn(r_));
        hi[sl] = h_; lo[sl] = l_; tl[sl] = r_ - l_;
        if (!(fabsf(wv) < HAV_FP16_LIMIT)) { hi[sl] = lo[sl] = tl[sl] = 0.f; }          // (out of the fp16 range: the range guard hands the call to the bf16 kernel)
        mh = fmaxf(mh, fabsf(hi[sl])); ml = fmaxf(ml, fabsf(lo[sl])); mt = fmaxf(mt, fabsf(tl[sl]));
    }
    const int sh = mx_block_exp(mh, 6.0f), st = mx_block_exp(mt, 6.0f), sl_ = mx_block_exp(ml, 7.5f);
    unsigned int ah[4] = {0, 0, 0, 0}, at[4] = {0, 0, 0, 0}, al[6] = {0, 0, 0, 0, 0, 0};
    for (int sl = 0; sl < 32; ++sl) {
        const unsigned int qh = mx_quant<2, 1, 1>(ldexpf(hi[sl], -sh)), qt = mx_quant<2, 1, 1>(ldexpf(tl[sl], -st));
        const unsigned int ql = mx_quant<2, 3, 1>(ldexpf(lo[sl], -sl_));
        ah[sl >> 3] |= qh << (4 * (sl & 7));
        at[sl >> 3] |= qt << (4 * (sl & 7));
        const int bit = 6 * sl, d = bit >> 5, o = bit & 31;
        al[d] |= ql << o;
        if (o + 6 > 32) al[d + 1] |= ql >> (32 - o);
    }
    unsigned int* rec = reinterpret_cast<unsigned int*>(blob) + OFF_MX + G * MX_G_DWORDS;
    for (int d = 0; d < 4; ++d) { rec[l * 4 + d] = ah[d]; rec[256 + l * 4 + d] = at[d]; rec[512 + l * 4 + d] = al[d]; }
    rec[768 + l * 2 + 0] = al[4]; rec[768 + l * 2 + 1] = al[5];
    rec[896 + l] = (unsigned int)(sh + 127) | ((unsigned int)(st + 127) << 8) | ((unsigned int)(sl_ + 127) << 16);
}

extern "C" int hav_mlp_pack(void* blob, const HavMlpWeights* w, void* stream)
{
    if (!blob || !w || !w->W1 || !w->b1 || !w->W2 || !w->b2 || !w->Wa || !w->ba || !w->Wf || !w->bf || !w->Wc || !w->bc)
        return HAV_EINVAL;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((BLOB_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float*)blob, *w);
    HAV_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_wmax_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (float*)blob, *w);
    HAV_LAUNCH_CHECK();
    hipLaunchKernelGGL(mlp_pack_mx_kernel, dim3(MX_GROUPS), dim3(64), 0, (hipStream_t)stream, (float*)blob, *w);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Per-frame plane preparation: P[p][b][texel][hu] = sum_c W1F[p][hu][c] * plane[p][b][c][texel]
// (NCHW in, channels-last 128-wide out, channel order = accumulator order of the march kernel).
// One workgroup = 64 texels x 128 outputs; plane tile, weights and the output tile go through LDS so that both the
// NCHW reads and the channels-last writes are coalesced.
// ------------------------------------------------------------------------------------------------
// trailer of the prepared-plane buffer (uint32 words): [2][128] bit patterns of max_texel |P_p[hu]| (atomicMax on the bits of a
// non-negative float is a float max; a NaN sorts above +inf) | [256] verdict: 0 = the fp16 split is safe, 1 = not | [257..258] the
// two bounds (diagnostics)
#define PREP_TRAILER_WORDS 512
__global__ void __launch_bounds__(256) plane_project_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                            const float* __restrict__ blob, int HW, unsigned int* __restrict__ trailer)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sX = sm;                 // [64 c][64 texels]
    float* sW = sm + 64 * 64;       // [128 hu][65]
    float* sO = sW + 128 * 65;      // [64 texels][129]
    const int pb = blockIdx.y;      // p * B + b
    const int p = pb / (gridDim.y / 2);
    const int t0 = blockIdx.x * 64;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const float* plane = src + (size_t)pb * HAV_PC * HW;
    for (int c = ty; c < HAV_PC; c += 4) sX[c * 64 + tx] = (t0 + tx < HW) ? plane[(size_t)c * HW + t0 + tx] : 0.f;
    const float* wsrc = blob + OFF_W1F + (size_t)p * 128 * 64;
    for (int i = tid; i < 128 * 64; i += 256) sW[(i >> 6) * 65 + (i & 63)] = wsrc[i];
    __syncthreads();
    // thread: texel tx, 32 outputs hu = ty*32 .. +31 (weights broadcast across the wave, texel column conflict-free)
    float acc[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) acc[o] = 0.f;
    for (int c = 0; c < HAV_PC; ++c) {
        const float xv = sX[c * 64 + tx];
#pragma unroll
        for (int o = 0; o < 32; ++o) acc[o] = fmaf(sW[(ty * 32 + o) * 65 + c], xv, acc[o]);
    }
#pragma unroll
    for (int o = 0; o < 32; ++o) sO[tx * 129 + ty * 32 + o] = acc[o];
    __syncthreads();
    // Layout: groups of 4 x-adjacent texels; inside a group the 16-byte piece (h, c) of the four texels is contiguous
    // ([group][h*16 + c][x & 3][4 floats], 512 floats per group).  The four lanes of a quad are x-adjacent rays at the same sample
    // index and mostly need the same piece of neighbouring texels: they now hit one 64-byte segment instead of four lines 512 bytes
    // apart (the texture path serves a quad per cycle only when it stays inside one line).  W % 4 == 0.
    float* out = dst + (size_t)pb * HW * 128;
    const int tg = p ? HAV_TG1 : HAV_TG;
    if (t0 + 64 <= HW && (64 >> tg) << tg == 64) {
        // the workgroup's 64 texels are 64 >> tg whole groups = ONE contiguous block of 64 x 128 floats in the prepared layout: walk it
        // in address order, 16 bytes per thread (a wave writes 1 KB contiguous; indexed by (texel, unit) the same stores were 16-byte
        // fragments at a 64-byte stride: four times the store transactions)
        float4* blk = reinterpret_cast<float4*>(out + (size_t)(t0 >> tg) * (128 << tg));
        for (int q = tid; q < 64 * 32; q += 256) {          // float4 index inside the block
            const int g = q / (32 << tg), rem = q - g * (32 << tg);          // group, float4 inside the group: [piece 0..31][texel 0..(1<<tg)-1]
            const int piece = rem >> tg, tx4 = rem & ((1 << tg) - 1);
            const float* so = sO + ((g << tg) + tx4) * 129 + 4 * piece;
            blk[q] = make_float4(so[0], so[1], so[2], so[3]);
        }
    } else {
        for (int i = tid; i < 64 * 128; i += 256) {
            const int t = i >> 7, hu = i & 127;
            const int tt = t0 + t;                       // y * W + x with W % 4 == 0: tt >> 2 = group, tt & 3 = x & 3
            if (tt < HW) out[(size_t)(tt >> tg) * (128 << tg) + (hu >> 2) * (4 << tg) + (tt & ((1 << tg) - 1)) * 4 + (hu & 3)] = sO[t * 129 + hu];
        }
    }
    if (tid < 128) {                // fp16 range guard: this tile's max |P[hu]| (texels past the end hold exact zeros)
        float m = 0.f;
        unsigned int bad = 0;
        for (int t = 0; t < 64; ++t) {
            const float x = fabsf(sO[t * 129 + tid]);
            bad |= (x != x);
            m = fmaxf(m, x);
        }
        atomicMax(&trailer[p * 128 + tid], bad ? 0x7FC00000u : __float_as_uint(m));
    }
}

// |h1_u| <= |b1_u| + sum_k |W1pe_uk| + max |P0_u| + max |P1_u|;  |h2_v| <= |b2_v| + sum_u |W2_vu| bound(h1_u)  (relu only shrinks)
__global__ void __launch_bounds__(128) range_verdict_kernel(unsigned int* __restrict__ trailer, const float* __restrict__ blob)
{
    __shared__ float b1u[128], red[128];
    const int t = threadIdx.x;
    {
        const float bound = blob[OFF_RNG + t] + __uint_as_float(trailer[t]) + __uint_as_float(trailer[128 + t]);     // t = hu
        b1u[acc_row((t & 63) >> 4, t & 15, t >> 6)] = bound;
        red[t] = (bound < HAV_FP16_LIMIT) ? bound : __builtin_inff();
    }
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) { if (t < st) red[t] = fmaxf(red[t], red[t + st]); __syncthreads(); }
    const float m1 = red[0];
    __syncthreads();
    {
        const int m = t >> 5, l31 = t & 31;             // row v = t of W2 in the fp32 fragment section: [ks][m][lane]
        float bound = fabsf(blob[OFF_B2 + t]);
        for (int ks = 0; ks < K2_STEPS; ++ks)
            for (int hh = 0; hh < 2; ++hh)
                bound += fabsf(blob[OFF_W2 + (ks * 4 + m) * 64 + hh * 32 + l31]) * b1u[acc_row(ks >> 4, ks & 15, hh)];
        red[t] = (bound < HAV_FP16_LIMIT) ? bound : __builtin_inff();
    }
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) { if (t < st) red[t] = fmaxf(red[t], red[t + st]); __syncthreads(); }
    if (t == 0) {
        const float m2 = red[0], wmax = blob[OFF_RNG + 128];
        const bool safe = (m1 < HAV_FP16_LIMIT) && (m2 < HAV_FP16_LIMIT) && (wmax < HAV_FP16_LIMIT);
        trailer[256] = safe ? 0u : 1u;
        trailer[257] = __float_as_uint(m1);
        trailer[258] = __float_as_uint(m2);
    }
}

extern "C" int64_t hav_triplane_prepared_bytes(int B, int H, int W) { return ((int64_t)2 * B * H * W * 128 + PREP_TRAILER_WORDS) * 4; }

// one bit per device id: which devices already have the dynamic-LDS attribute of this file's kernels raised
static bool attr_done(std::atomic<unsigned long long>& mask, int& dev)
{
    dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    return (mask.load(std::memory_order_acquire) >> dev) & 1ull;
}

extern "C" int hav_triplane_prepare(float* dst, const float* src_nchw, const void* mlp_blob, int B, int C, int H, int W,
                                    void* stream)
{
    if (!dst || !src_nchw || !mlp_blob || B < 1 || H < 1 || W < 1) return HAV_EINVAL;
    if (C != HAV_PC || (W & ((1 << HAV_TG) - 1)) || (W & ((1 << HAV_TG1) - 1))) return HAV_EUNSUP;
    const int HW = H * W;
    const size_t lds = (64 * 64 + 128 * 65 + 64 * 129) * sizeof(float);
    static std::atomic<unsigned long long> attr_mask{0};
    int dev;
    if (!attr_done(attr_mask, dev)) {
        hipError_t e = hipFuncSetAttribute((const void*)plane_project_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_mask.fetch_or(1ull << dev, std::memory_order_release);
    }
    unsigned int* trailer = reinterpret_cast<unsigned int*>(dst + (size_t)2 * B * HW * 128);
    hipError_t me = hipMemsetAsync(trailer, 0, PREP_TRAILER_WORDS * 4, (hipStream_t)stream);
    if (me != hipSuccess) return (int)me;
    hipLaunchKernelGGL(plane_project_kernel, dim3((HW + 63) / 64, 2 * B), dim3(256), lds, (hipStream_t)stream, dst, src_nchw,
                       (const float*)mlp_blob, HW, trailer);
    HAV_LAUNCH_CHECK();
    hipLaunchKernelGGL(range_verdict_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, trailer, (const float*)mlp_blob);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#define DPP_QUAD_XOR1 0xB1
#define DPP_QUAD_XOR2 0x4E
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_HALF_MIRROR 0x141

template <int CTRL> __device__ __forceinline__ float dpp_mov(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<DPP_QUAD_XOR1>(0.f, v);
    v += dpp_mov<DPP_QUAD_XOR2>(0.f, v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(0.f, v);
    v += dpp_mov<DPP_ROW_MIRROR>(0.f, v);
    return v;
}
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR1>(0.f, v));
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR2>(0.f, v));
    v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(0.f, v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(0.f, v));
    return v;
}
// inclusive product scan along the 16 lanes of a DPP row (Kogge-Stone, identity 1)
__device__ __forceinline__ float row16_scan_mul(float v)
{
    v *= dpp_mov<DPP_ROW_SHR(1)>(1.f, v);
    v *= dpp_mov<DPP_ROW_SHR(2)>(1.f, v);
    v *= dpp_mov<DPP_ROW_SHR(4)>(1.f, v);
    v *= dpp_mov<DPP_ROW_SHR(8)>(1.f, v);
    return v;
}
__device__ __forceinline__ float read_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// value of the partner lane in the other half-wave (lane ^ 32): ds_bpermute_b32 (__shfl_xor).  gfx950's VALU instruction for it,
// v_permlane32_swap_b32, is NOT usable here with ROCm 7.2: results are right on small runs and differ from launch to launch on a full
// frame (8 % of the launches of a 512^2 frame; docs/history/DESIGN_r1-r4.md 3.12, tools/ubench/permlane32.hip).
__device__ __forceinline__ float half_swap(float v, int h)
{
    return __shfl_xor(v, 32, 64);
}
// sigmoid with the reciprocal instruction (1 ulp) instead of the compiler's IEEE division sequence -- three of those interleaved per
// sample is the pattern docs/history/DESIGN_r1-r4.md 3.12 is about; rcp(inf) = 0 covers exp overflow (no Newton step: inf * 0 would poison it)
__device__ __forceinline__ float sigmoid_rcp(float x) { return __builtin_amdgcn_rcpf(1.0f + expf(-x)); }
// orders this wave's LDS traffic (the per-wave scratch is private to a wave: no workgroup barrier needed)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Philox4x32-10 (Salmon et al. 2011), counter-based: one call per (ray, sample, stream)
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Positional encoding pair for one angle x = p * 2^k (model/network/embedder.py:42-56): the reference evaluates
// sin(x) and sin(fl(x + pi/2)) in fp32.  One branch-free Cody-Waite reduction by pi/2 (3 FMA terms) + degree-9/8
// minimax polynomials gives sin x and cos x (max abs error 9.3e-8 for |x| <= 260, i.e. |p| <= 2 at the top octave; the
// box is +-1.6; accuracy degrades gracefully up to |x| ~ 1e5).  The reference's second value is NOT cos x: the fp32
// rounding of x + pi/2 moves the angle by up to half an ulp of x (1e-5 at the top octave).  TwoSum recovers that
// rounding error exactly, and sin(fl(x+pi/2)) = cos(x + eps) = cos x - eps sin x reproduces the reference's value to
// 1.2e-7 (validated against float64 in DESIGN.md) while sharing the range reduction between the two outputs.
__device__ __forceinline__ void pe_pair(float x, float& s_out, float& c_out)
{
    const float n = rintf(x * 0.63661977236758134f);
    float r = fmaf(-n, 1.5707964f, x);
    r = fmaf(-n, -4.371139e-08f, r);
    r = fmaf(-n, -1.7763568e-15f, r);
    const float s = r * r;
    float p = fmaf(s, 2.7158228022017283e-06f, -0.00019839018932543695f);
    p = fmaf(s, p, 0.008333328180015087f);
    p = fmaf(s, p, -0.1666666716337204f);
    const float sn = fmaf(r * s, p, r);
    float q = fmaf(s, -2.7208204755879706e-07f, 2.479949216649402e-05f);
    q = fmaf(s, q, -0.0013888883404433727f);
    q = fmaf(s, q, 0.0416666679084301f);
    const float cs = fmaf(s * s, q, fmaf(s, -0.5f, 1.0f));
    const int k = (int)n;
    const float a = (k & 1) ? cs : sn, b = (k & 1) ? sn : cs;
    const float sinx = (k & 2) ? -a : a;
    const float cosx = ((k + 1) & 2) ? -b : b;
    const float hp = 1.57079632679489661923f;
    const float t = x + hp;                       // the reference's fp32 angle
    const float bb = t - x;
    const float err = (x - (t - bb)) + (hp - bb); // x + hp == t + err exactly (TwoSum)
    const float eps = 4.371139e-08f - err;        // t == x + pi/2 + eps   (hp - pi/2 = +4.371139e-8)
    s_out = sinx;
    c_out = fmaf(-eps, sinx, cosx);
}

struct MarchArgs {
    HavRenderParams p;
    const float* rays; const float* bg; const float* inv_T; const float* planes; const float* vol; const float* blob;
    const float* t_rand; const float* u_rand; const float* noise_c; const float* noise_f;
    HavRenderOut out;
    unsigned long long rng_base;   // p.rng_offset (+ *p.rng_counter, read on the device)
    int ablate;             // HAV_ABLATE bit mask (timing experiments only; results are wrong when set)
    int stagger;            // start-up delay per wave index, in units of 64 cycles (phase de-synchronisation of the CU's 8 waves)
    float* ws;              // fine-pass cache (CACHE kernels): one slot of ws_slot floats per wave of the grid
    long long ws_slot;
    float* dbg_zfine;       // optional [B*R, S_fp] dump of the merged fine depths (tests)
    long long NR;           // B*R
    int S_fp;               // ceil(S_c/2) + S_f, 0 if no fine pass
    int scr_floats;         // per-wave LDS scratch
    int o_w, o_cdf, o_cand, o_zf, o_racc, s_pad_c, s_pad_f;
    const float* pplanes;   // projected tri-planes [2,B,H,W,128] (hav_triplane_prepare)
    const unsigned int* guard;   // fp16 range verdict word in the prepared planes' trailer (nullptr: no guard on this launch)
    int guard_run_if;       // the launch proceeds iff (*guard != 0) == (guard_run_if != 0): fp16 kernel 0, its bf16 fallback 1
    unsigned int* status;   // optional HavRenderParams.status
    // wave-uniform constants the host computes once (the same correctly-rounded fp32 / fp64 divisions the kernels used to do: on the
    // device they sat in VGPRs as loop invariants and were spilled)
    float step_c;           // 1 / (S_c - 1)
    float u_st, u_sN, u_w;  // 1 / (S_f - 1) | (float)(1.0 / S_f) | (float)(1.0 / S_f - 1e-6)   (utils/nerf_util.py:93-99)
    float plim, vlim;       // plane_res - 1 | vol_res - 1
};

enum { STREAM_XI = 0, STREAM_ZETA = 1, STREAM_EPS_C = 2, STREAM_EPS_F = 3 };

__device__ __forceinline__ unsigned long long rng_off(const MarchArgs& a)
{
    return a.p.rng_counter ? a.p.rng_offset + *a.p.rng_counter : a.p.rng_offset;
}

__global__ void rng_advance_kernel(unsigned long long* c) { *c += 1ull; }

// Uniform [0,1) for the stratified jitter (xi, zeta): counter-based -- a pure function of (seed, call offset, ray, sample,
// stream): murmur3's finaliser (a bijective avalanche mixer) over [hashed ray key] ^ [sample/stream/call counter word].  The
// jitter only needs equidistribution inside a bin (tests/test_host_logic.py checks moments, bin counts and lag correlations of
// this formula), and one draw is ~10 integer ops where Philox4x32-10 (kept for the Gaussian density noise below) is ~120; the
// block kernel evaluates one such number per sample on the critical VALU path.
__device__ __forceinline__ uint32_t fmix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// per-ray part of the key: two multiplies + one finaliser, hoisted out of the sample loops by the callers
struct RKey { uint32_t ray, call, off; float step; };   // hashed ray key | wave-uniform call key | call counter (read from memory ONCE per kernel) | 1 / (S_c - 1)
__device__ __forceinline__ uint32_t rng_call_off(const MarchArgs& a) { return (uint32_t)rng_off(a); }
__device__ __forceinline__ RKey rng_ray_key(const MarchArgs& a, long long gr, uint32_t call_off)
{
    RKey k;
    k.ray = fmix32((uint32_t)gr * 0x9E3779B1u + (uint32_t)a.p.seed) ^ ((uint32_t)((unsigned long long)gr >> 32) * 0x7FEB352Du);
    k.call = (uint32_t)(a.p.seed >> 32) + call_off * 0x68E31DA4u;
    k.off = call_off;
    k.step = a.step_c;
    return k;
}
__device__ __forceinline__ float rng_uniform(const MarchArgs& a, const RKey& k, int s, int stream)
{
    const uint32_t x = fmix32(k.ray ^ ((uint32_t)s * 0x846CA68Bu + (uint32_t)stream * 0x632BE5ABu + k.call));
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float rng_normal(const MarchArgs& a, const RKey& k, long long gr, int s, int stream)
{
    uint32_t o[4];
    philox4x32((uint32_t)gr, (uint32_t)((unsigned long long)gr >> 32), (uint32_t)s,
               (uint32_t)stream + 16u * k.off, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32), o);
    const float u1 = ((float)(o[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// coarse depth of sample i (model/nerf_trainer.py:129-139): torch.linspace + stratified jitter
__device__ __forceinline__ float lin_t(int k, int S, float step) { return (k < S / 2) ? step * (float)k : 1.0f - step * (float)(S - 1 - k); }
// RANDOM: 0 = deterministic, 1 = device RNG only (production), 2 = injected tensors where given (parity tests), else device RNG
template <int RANDOM>
__device__ __forceinline__ float z_coarse(const MarchArgs& a, long long gr, const RKey& rkey, int i, float near, float far)
{
    const int S = a.p.S_c;
    const float step = rkey.step;      // (through the key so that the block kernel can fence it: see RAY_FENCE)
    const float t = lin_t(i, S, step);
    float z = near * (1.0f - t) + far * t;
    if (RANDOM && a.p.perturb) {
        const int il = i > 0 ? i - 1 : 0, ih = i < S - 1 ? i + 1 : S - 1;
        const float tl = lin_t(il, S, step), th = lin_t(ih, S, step);
        const float zl = near * (1.0f - tl) + far * tl, zh = near * (1.0f - th) + far * th;
        const float lo = i > 0 ? 0.5f * (z + zl) : z;
        const float up = i < S - 1 ? 0.5f * (zh + z) : z;
        const float xi = (RANDOM == 2 && a.t_rand) ? a.t_rand[gr * S + i] : rng_uniform(a, rkey, i, STREAM_XI);
        z = lo + (up - lo) * xi;
    }
    return z;
}

// ------------------------------------------------------------------------------------------------
// One sample per lane pair: geometry + skinning field, 8-tap gather of the prepared planes into the layer-1
// accumulators, PE columns + layer 2 on the matrix cores, head rows on the VALU.
// Outputs: acc2 = relu(h2) (this half-wave's 64 hidden units), hd0..2 = raw rgb, hd3 = raw density.
// ------------------------------------------------------------------------------------------------
struct LaneCtx {
    const float* sW1; const float* sW2; const float4* sW4;
    const uint4* sA1; const uint4* sA2;         // split-bf16 fragments (PREC == 1 kernels)
    const uint4* sAF;                           // fc_rgbFeat fragments (fp16 split; feature parking, PREC == 2)
    const unsigned int* sMX;                    // MX correction records (PREC == 3): layer 1 at 0, layer 2 at 4 * MX_G_DWORDS
    const float4* sB;                           // LDS copy of b1 | b2 (block kernel; the pair kernel reads them through the buffer path)
    __amdgpu_buffer_rsrc_t wrs;
    int lane, h, hoff;
};

#define LDB4(off_floats) __builtin_amdgcn_raw_buffer_load_b128(L.wrs, h * 16, (off_floats) * 4, 0)

// GQ = float4 loads per pipeline stage of the gather (two stages in flight): 16 = a whole 256-B tap per stage (128 VGPRs
// of loads in flight, pair kernel), 8 = half a tap (64 VGPRs; block kernel, which also keeps 64 accumulators alive).
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

// 3-way split of two fp32 values into packed bf16 pairs: v = hi + mid + lo exactly (truncation at each level keeps every
// remainder representable).  v_perm_b32 picks the upper halves of the two words, so no masking is needed for the packing.
__device__ __forceinline__ void split3(float v0, float v1, uint32_t& ph, uint32_t& pm, uint32_t& pl)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t u0 = __float_as_uint(v0), u1 = __float_as_uint(v1);
    ph = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const f2 v = {v0, v1};
    const f2 t = {__uint_as_float(u0 & 0xFFFF0000u), __uint_as_float(u1 & 0xFFFF0000u)};
    const f2 r = v - t;                                     // one v_pk_add_f32 (exact)
    const uint32_t a0 = __float_as_uint(r.x), a1 = __float_as_uint(r.y);
    pm = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    const f2 t2 = {__uint_as_float(a0 & 0xFFFF0000u), __uint_as_float(a1 & 0xFFFF0000u)};
    const f2 q = r - t2;
    pl = __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), 0x07060302u);
}

// relu of the 4 accumulator tiles of a layer, one instruction per value.  fmaxf(x, 0) on an MFMA result costs two (the
// compiler first canonicalises the operand with v_max x,x; v_med3(x,0,inf) is folded back to the same pair; the integer form
// bitcast(max(bitcast<int>(x), 0)) is "simplified" into wrong code by this compiler), so the v_max is issued from inline asm, IN
// PLACE (a separate output operand made the compiler rebuild each accumulator tuple with ~10 v_mov_b64).  The compiler pads
// NO wait states between an MFMA and an asm statement that reads its result (measured: ~1 % of rays corrupted, run to run,
// with a bare asm v_max), so the first statement carries the XDL-write -> VALU-read wait states itself (24 >= the 16-pass
// requirement) and ties all four tiles to it.
// (In the variants that also carry the composited-hidden-unit accumulators the in-place form costs more in spills than it
// saves in moves: they keep the two-operand form, INPLACE = false.)
#define HAV_RELU_NOPS "s_nop 15\n\ts_nop 7"
template <bool INPLACE>
__device__ __forceinline__ void relu_tiles(f32x16 (&acc)[4])
{
    if (!INPLACE) {
        asm volatile(HAV_RELU_NOPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[m][r], y;
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
                acc[m][r] = y;
            }
        return;
    }
    asm volatile(HAV_RELU_NOPS : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = acc[m][r];
            asm volatile("v_max_f32 %0, 0, %0" : "+v"(x));
            acc[m][r] = x;
        }
}

// acc[0..3] += W . x over NCH 16-wide k chunks with both operands split hi/mid/lo: the six products whose magnitude is
// >= 2^-16 of the leading one (hh, hm, mh, hl, lh, mm), smallest first; every bf16 x bf16 product is exact in fp32 and the
// MFMA accumulates in fp32, so the dropped terms (ml, lm, ll) bound the error at ~2^-23 relative per product -- fp32-sgemm
// class.  The A fragments of group g+1 (one row tile of one chunk: 3 x ds_read_b128) are requested before the six MFMAs of
// group g are issued (explicit register double buffer: under this kernel's register pressure the compiler otherwise
// re-uses one buffer and every group waits out a full LDS round trip).
// (gfx950 does NOT read matrix operands after issue -- tools/ubench/mfma_war.hip: an operand register overwritten by the very next
// instruction, 0 of 10^10 results change; rounds 1-2 believed otherwise and padded these sequences.  What is left of that period in this
// routine -- 32 wait states behind a chunk's last group and a scheduling barrier per group -- is part of the binary that went through
// the determinism stress runs (41 000 launches, docs/history/DESIGN_r1-r4.md 3.12) and is kept as it is.)
template <int NCH, typename GetV>
__device__ __forceinline__ void mfma_split3(f32x16 (&acc)[4], const uint4* frag /* [NCH][4 m][3 parts][64 lanes] */, int lane, GetV getv)
{
    uint4 A[2][3];
    uint4 bh, bm, bl;
    bf16x8_t pa, pb;                  // operands of the previous group's last MFMA (tied to the wait states below)
    // (Do NOT launder `frag` through an empty asm to stop address hoisting: the pointer then loses its LDS address space and every
    // fragment read becomes a FLAT load -- +0.8 ms per frame, measured.)
#pragma unroll
    for (int q = 0; q < 3; ++q) A[0][q] = frag[q * 64 + lane];
#pragma unroll
    for (int g = 0; g < NCH * 4; ++g) {
        const int m = g & 3, ch = g >> 2;
        if (m == 0) {
            float v[8];
            getv(ch, v);
            split3(v[0], v[1], bh.x, bm.x, bl.x); split3(v[2], v[3], bh.y, bm.y, bl.y);
            split3(v[4], v[5], bh.z, bm.z, bl.z); split3(v[6], v[7], bh.w, bm.w, bl.w);
        }
        const bf16x8_t xh = __builtin_bit_cast(bf16x8_t, bh), xm = __builtin_bit_cast(bf16x8_t, bm), xl = __builtin_bit_cast(bf16x8_t, bl);
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, A[g & 1][0]), am = __builtin_bit_cast(bf16x8_t, A[g & 1][1]);
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, A[g & 1][2]);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, acc[m], 0, 0, 0);
        if (g + 1 < NCH * 4) {        // next group's fragments go into the buffer whose last reader (group g-1) is long done
#pragma unroll
            for (int q = 0; q < 3; ++q) A[(g + 1) & 1][q] = frag[((g + 1) * 3 + q) * 64 + lane];
        }
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xm, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xh, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xm, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, acc[m], 0, 0, 0);
        pa = ah; pb = xh;
        if (m == 3 && g + 1 < NCH * 4) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[m]) : "v"(pa), "v"(pb));
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[3]) : "v"(pa), "v"(pb));      // the last MFMA's operands outlive it by 32 wait states
}
// Two-part fp16 variant: x = hi + lo with both parts rounded to nearest (representation error <= 2^-22 |x|, the order of the fp32
// accumulation error of a 128-term dot product), three products (hi.lo, lo.hi, hi.hi) on v_mfma_f32_32x32x16_f16: half the
// matrix time and two thirds of the LDS of the bf16 triple split.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(float v0, float v1, uint32_t& ph, uint32_t& pl)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const f2 v = {v0, v1};
    const h2 hi = __builtin_convertvector(v, h2);
    ph = __builtin_bit_cast(uint32_t, hi);
    // lo = fp16(v - (float)hi), the subtraction exact in fp32: the mixed-precision FMAs read the fp16 half directly and write the rounded
    // fp16 result into one half of the destination -- 3 instructions per operand pair instead of 5 (two cvt_f32_f16, pk_add, cvt_pk)
    uint32_t lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(ph), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(ph), "v"(v1));
    pl = lo;
}
template <int NCH, int NM, typename GetV>
__device__ __forceinline__ void mfma_split2h(f32x16 (&acc)[NM], const uint4* frag /* [NCH][NM row tiles][2 parts][64 lanes] */, int lane, GetV getv)
{
    constexpr int FD = 1, NBUF = FD + 1, NG = NCH * NM;      // FD: groups of MFMAs a fragment read runs ahead of its use (2 and 3 measured the same)
    uint4 A[NBUF][2];
    uint4 bh, bl;
#pragma unroll
    for (int d = 0; d < FD; ++d)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (d < NG) A[d][q] = frag[(d * 2 + q) * 64 + lane];
#pragma unroll
    for (int g = 0; g < NCH * NM; ++g) {
        const int m = g % NM, ch = g / NM;
        if (m == 0) {
            float v[8];
            getv(ch, v);
            split2h(v[0], v[1], bh.x, bl.x); split2h(v[2], v[3], bh.y, bl.y);
            split2h(v[4], v[5], bh.z, bl.z); split2h(v[6], v[7], bh.w, bl.w);
        }
        const f16x8_t xh = __builtin_bit_cast(f16x8_t, bh), xl = __builtin_bit_cast(f16x8_t, bl);
        const f16x8_t ah = __builtin_bit_cast(f16x8_t, A[g % NBUF][0]), al = __builtin_bit_cast(f16x8_t, A[g % NBUF][1]);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc[m], 0, 0, 0);
        if (g + FD < NG) {           // into the buffer whose last reader (group g - 1) is long done
#pragma unroll
            for (int q = 0; q < 2; ++q) A[(g + FD) % NBUF][q] = frag[((g + FD) * 2 + q) * 64 + lane];
        }
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc[m], 0, 0, 0);
    }
}


// "fp16 x 2 + MX" (PREC 3): acc[0..3] += W . x with every operand = hi + lo + tail, hi = rn16(v), lo = rn16(v - hi), tail = v - hi - lo
// (exact: 11 + 11 + sign bits leave <= 2^-23 |v| for the tail).  The three leading partial products (lo.hi, hi.lo, hi.hi) run on
// v_mfma_f32_32x32x16_f16 exactly as in mfma_split2h; the three terms of order 2^-22 that the fp16 mode drops -- hi.tail, tail.hi, lo.lo --
// are added by ONE block-scaled 4- / 6-bit matrix instruction each per 64 k (v_mfma_scale_f32_32x32x64_f8f6f4, K = 64: 4 chunks x 16):
//     acc += A[fp4  hi  ] . B[bf6 tail]      acc += A[fp4 tail] . B[bf6  hi  ]      acc += A[fp6  lo  ] . B[fp6  lo  ]
// Their factors need 2-4 significant bits: a relative error of 2^-3 on a term of 2^-22 |a b| is 2^-25 |a b|, below the rounding of the fp32
// product itself, so the mode carries the full 24-bit operands at 3 + 3 x 1.25 / 4 = 3.9 sixteen-bit-product equivalents per k chunk instead
// of the bf16 triple split's 6 (tools/ubench/mx6_probe.hip: operand layouts, scale semantics and the instruction's issue rate, measured).
// Weights: pre-split, pre-quantised records in LDS (hav_mlp_pack, MX_G_DWORDS), block scale per (row, 32 k).  Activations: per lane and group
// the 32 values are split here, the three kinds converted with one v_cvt_scalef32_pk32 / 2xpk16 instruction each against a per-lane power of
// two taken from the largest value (DYN; the positional encoding is bounded by 1: static).
// NCH = 3 (layer 1: one group, slots 24-31 zero) or 8 (layer 2: two groups).  getv(ch, v[8]) as in mfma_split2h.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x6_t __attribute__((ext_vector_type(6)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x32_t __attribute__((ext_vector_type(32)));
__device__ __forceinline__ i32x8_t mx_op4(const u32x4_t a) { return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], 0, 0, 0, 0}; }
__device__ __forceinline__ i32x8_t mx_op6(const u32x4_t a, const u32x2_t b) { return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], 0, 0}; }
__device__ __forceinline__ i32x8_t mx_op6(const u32x6_t a) { return i32x8_t{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)a[4], (int)a[5], 0, 0}; }
// (tools/ubench/mx6_probe.hip T4: like v_mfma_f32_32x32x16_f16, the scaled instruction does NOT read its A / B / scale registers after issue.)
#ifndef HAV_MX_TERMS
#define HAV_MX_TERMS 7          // lab: bit 0 hi.tail, bit 1 tail.hi, bit 2 lo.lo
#endif
template <int NCH, bool DYN, typename GetV>
__device__ __forceinline__ void mfma_split2x(f32x16 (&acc)[4], const uint4* frag /* fp16: [NCH][4 row tiles][2 parts][64 lanes] */,
                                             const unsigned int* mx /* this layer's records: [k group][4 row tiles][MX_G_DWORDS] */, int lane, GetV getv)
{
    constexpr int NGRP = (NCH + 3) / 4;
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
        const int nc = (NCH - 4 * g) < 4 ? (NCH - 4 * g) : 4;
        // ---- activations of the group: hi, lo (packed fp16: the B operands of the fp16 products) and the tails ----
        uint4 xh4[4], xl4[4];          // per chunk: 8 packed fp16 = the B operand of the chunk's fp16 products
        // tails as packed bf16 (the upper halves of the exact fp32 tails: 8 significant bits, of which the 6-bit format keeps 3).  NOT through
        // v_cvt_scalef32_2xpk16_bf6_f32, the conversion that takes fp32 sources: in this kernel its results differed from launch to launch
        // on two thirds of the rays (profiles/r05_mx_bisect.txt: the term that uses them alone breaks the bitwise repeatability, the other
        // two conversions never do), while the bf16 / fp16-source forms are stable -- and the bf16 form needs half the registers.
        typedef unsigned int u32x16_t __attribute__((ext_vector_type(16)));
        typedef __bf16 bf16x32_t __attribute__((ext_vector_type(32)));
        u32x16_t TB;
        float vmax = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v[8];
            if (c < nc) getv(4 * g + c, v);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const float v0 = v[2 * q], v1 = v[2 * q + 1];
                const h2 hi = __builtin_convertvector(f2{v0, v1}, h2);
                const uint32_t ph = __builtin_bit_cast(uint32_t, hi);
                float d0, d1;          // v - hi, exact
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(ph), "v"(v0));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(ph), "v"(v1));
                const h2 lo = __builtin_convertvector(f2{d0, d1}, h2);
                const uint32_t pl = __builtin_bit_cast(uint32_t, lo);
                float t0, t1;          // v - hi - lo, exact
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(pl), "v"(d0));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(pl), "v"(d1));
                if (q == 0) { xh4[c].x = ph; xl4[c].x = pl; } else if (q == 1) { xh4[c].y = ph; xl4[c].y = pl; }
                else if (q == 2) { xh4[c].z = ph; xl4[c].z = pl; } else { xh4[c].w = ph; xl4[c].w = pl; }
                TB[4 * c + q] = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
                if (DYN) vmax = fmaxf(vmax, fmaxf(fabsf(v0), fabsf(v1)));
            }
        }
        // per-lane block exponent E (biased) of the group: values < 2^(E - 126).  hi / 2^(E-130) < 16 (bf6: 28), |lo| / 2^(E-140) <= 4 (fp6: 7.5),
        // |tail| / 2^(E-153) <= 8 (bf6) -- where lo is a NORMAL fp16; below 2^-14 lo is rounded on the subnormal grid and the tail can reach
        // 2^-25 whatever E is: its scale never goes below 2^-29 (tail / scale <= 16).  An all-zero group keeps the bytes valid the same way.
        unsigned int E = DYN ? (__float_as_uint(vmax) >> 23) : 127u;          // (PE inputs: |v| <= 1 + 1e-7 < 2)
        if (DYN) E = E < 40u ? 40u : (E > 254u ? 254u : E);
        const unsigned int bh = E - 3u, bl = E - 13u, bt = (E - 26u) < 98u ? 98u : (E - 26u);
        const float sch = __uint_as_float(bh << 23), scl = __uint_as_float(bl << 23), sct = __uint_as_float(bt << 23);
        const int sbv = (int)(bh | (bl << 8) | (bt << 16));
        typedef _Float16 f16x16_t __attribute__((ext_vector_type(16)));
        auto cat32 = [](const uint4 (&x)[4]) {
            const f16x16_t a = __builtin_shufflevector(__builtin_bit_cast(f16x8_t, x[0]), __builtin_bit_cast(f16x8_t, x[1]), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            const f16x16_t b = __builtin_shufflevector(__builtin_bit_cast(f16x8_t, x[2]), __builtin_bit_cast(f16x8_t, x[3]), 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
            return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
        };
        const u32x6_t H6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(cat32(xh4), sch);
        const u32x6_t L6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(cat32(xl4), scl);
        const u32x6_t T6 = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(__builtin_bit_cast(bf16x32_t, TB), sct);
        // ---- the three correction terms, row tile by row tile ----
        const unsigned int* rec = mx + g * 4 * MX_G_DWORDS;
        u32x4_t ah[2], at[2], ala[2]; u32x2_t alb[2]; int sc[2];
        auto ldrec = [&](int slot, int m) {
            const unsigned int* r = rec + m * MX_G_DWORDS;
            ah[slot] = reinterpret_cast<const u32x4_t*>(r)[lane];
            at[slot] = reinterpret_cast<const u32x4_t*>(r + 256)[lane];
            ala[slot] = reinterpret_cast<const u32x4_t*>(r + 512)[lane];
            alb[slot] = reinterpret_cast<const u32x2_t*>(r + 768)[lane];
            sc[slot] = (int)r[896 + lane];
        };
        ldrec(0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m + 1 < 4) ldrec((m + 1) & 1, m + 1);
            const int k = m & 1;
            if (HAV_MX_TERMS & 1) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op4(ah[k]), mx_op6(T6), acc[m], 4, 3, 0, sc[k], 2, sbv);          // hi . tail
            if (HAV_MX_TERMS & 2) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op4(at[k]), mx_op6(H6), acc[m], 4, 3, 1, sc[k], 0, sbv);          // tail . hi
            if (HAV_MX_TERMS & 4) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(mx_op6(ala[k], alb[k]), mx_op6(L6), acc[m], 2, 2, 2, sc[k], 1, sbv);  // lo . lo
        }
        // ---- the three leading products of the group's chunks on the fp16 instruction (as mfma_split2h) ----
        uint4 A[2][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) A[0][q] = frag[((4 * g * 4) * 2 + q) * 64 + lane];
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) {
            const int c = gg >> 2, m = gg & 3;
            if (c < nc) {
                const f16x8_t xh = __builtin_bit_cast(f16x8_t, xh4[c]), xl = __builtin_bit_cast(f16x8_t, xl4[c]);
                const f16x8_t ahh = __builtin_bit_cast(f16x8_t, A[gg & 1][0]), all_ = __builtin_bit_cast(f16x8_t, A[gg & 1][1]);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(all_, xh, acc[m], 0, 0, 0);
                if (gg + 1 < 4 * nc) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) A[(gg + 1) & 1][q] = frag[(((4 * g) * 4 + gg + 1) * 2 + q) * 64 + lane];
                }
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahh, xl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahh, xh, acc[m], 0, 0, 0);
            }
        }
    }
}

#include "hav_render_lab.h"      // ABL / TICK / PROF_* / DBG_*: diagnostic builds only, all empty in the shipped library

// PREC = 0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  PREC = 1: split-operand bf16 MFMA (3 x bf16 per operand, 6 products).
// The evaluation ends by calling cont(acc2, hd0, hd1, hd2, hd3): the caller's per-tile epilogue (compositing, parking).
template <int GQ, int PREC, bool BLK, bool LEAN = false, typename Cont>
__device__ __forceinline__ void sample_eval(const MarchArgs& a_in, const LaneCtx& L, int b, float ox, float oy, float oz, float dx,
                                            float dy, float dz, float z, Cont&& cont PROF_ARG DBG_ARG)
{
    const MarchArgs& a = a_in;
    // the lane id is opaque per tile: what derives from it (half-wave, fragment columns, strip slots) is then re-derived in one or two
    // ALU ops where it is used instead of being hoisted out of the sample loop as dozens of loop invariants, spilled and RELOADED
    int lane = L.lane;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5;
    const float* sW1 = L.sW1;
    const float* sW2 = L.sW2;
    const float4* sW4 = L.sW4;
    // the bias reads are loop-invariant; an opaque base per tile keeps the compiler from hoisting 128 values out of the sample
    // loop (and spilling them)
    // (the OFFSET is made opaque, not the pointer: a pointer that has been through an asm statement loses its LDS address space and
    // every read through it becomes a FLAT load -- slower, and FLAT accesses complete out of order with the other memory counters)
    int sb_off = 0;
    asm volatile("" : "+v"(sb_off));          // (in an SGPR instead, the surrounding code was allocated differently and the hazard of docs/history/DESIGN_r1-r4.md 3.12 was 100x more frequent)
    const float4* sBt = L.sB + sb_off;
    const int PR = a.p.plane_res, VR = a.p.vol_res;
    // ---- pts = o + d z; skinning field (model/Skinning_Field.py:77-95): half-wave h evaluates bone h ---------
    const float px = ox + dx * z, py = oy + dy * z, pz = oz + dz * z;
    const float* iT = a.inv_T + (size_t)b * 12;
    const float tx_ = px + iT[9], ty_ = py + iT[10], tz_ = pz + iT[11];
    const float p1x = tx_ * iT[0] + ty_ * iT[3] + tz_ * iT[6];
    const float p1y = tx_ * iT[1] + ty_ * iT[4] + tz_ * iT[7];
    const float p1z = tx_ * iT[2] + ty_ * iT[5] + tz_ * iT[8];
    float wmine;
    {
        const float gx = (h ? p1x : px) * a.p.skin_scale[0] + a.p.skin_trans[0];
        const float gy = (h ? p1y : py) * a.p.skin_scale[1] + a.p.skin_trans[1];
        const float gz = (h ? p1z : pz) * a.p.skin_scale[2] + a.p.skin_trans[2];
        // grid_sample 3-D, border padding, align_corners=True (utils/util.py:409-418)
        const float lim = a.vlim;
        float ix = ((gx + 1.0f) * 0.5f) * lim, iy = ((gy + 1.0f) * 0.5f) * lim, iz = ((gz + 1.0f) * 0.5f) * lim;
        ix = fminf(fmaxf(ix, 0.f), lim); iy = fminf(fmaxf(iy, 0.f), lim); iz = fminf(fmaxf(iz, 0.f), lim);
        const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
        const float fx = ix - x0f, fy = iy - y0f, fz = iz - z0f;
        const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
        const int x1 = min(x0 + 1, VR - 1), y1 = min(y0 + 1, VR - 1), z1 = min(z0 + 1, VR - 1);  // weight is 0 when clamped
        const float* v = a.vol + (size_t)h * VR * VR * VR;
        float acc = 0.f;
#pragma unroll
        for (int cz = 0; cz < 2; ++cz)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                for (int cx = 0; cx < 2; ++cx) {
                    const float wgt = (cx ? fx : 1.0f - fx) * (cy ? fy : 1.0f - fy) * (cz ? fz : 1.0f - fz);
                    acc += v[((size_t)(cz ? z1 : z0) * VR + (cy ? y1 : y0)) * VR + (cx ? x1 : x0)] * wgt;
                }
        wmine = acc;
    }
    const float wother = half_swap(wmine, h);
    const float w0 = h ? wother : wmine, w1 = h ? wmine : wother;
    const float den = (w0 + w1) + 1e-8f;
    // one reciprocal + one Newton step (<= 1 ulp from the IEEE quotient), NOT w / den: with ROCm 7.2 on gfx950 the compiler's IEEE expansion
    // (two interleaved v_div_scale / v_div_fmas / v_div_fixup sequences, VCC and SGPR-pair traffic in between) is where a rare run-to-run
    // difference came from -- identical inputs, a different warped point in lanes 48-63 of one tile; 22 % of the launches of a 512^2 frame on
    // one box with the IEEE form, 0 of 41 000 with this one (docs/history/DESIGN_r1-r4.md 3.12)
    float rden = __builtin_amdgcn_rcpf(den);
    rden = rden * (2.0f - den * rden);
    float n0 = w0 * rden, n1 = w1 * rden;
    const float qx_ = n0 * px + n1 * p1x, qy_ = n0 * py + n1 * p1y, qz_ = n0 * pz + n1 * p1z;   // p'
    DBG_PUT(4, z); DBG_PUT(5, wmine); DBG_PUT(6, wother); DBG_PUT(7, (qx_ + qy_) + qz_);
    DBG_PUT(8, den); DBG_PUT(9, n0); DBG_PUT(10, n1); DBG_PUT(11, (px + py + pz) + (p1x + p1y + p1z));
    TICK(1);

    // ---- layer 1 (model/nerf_model.py:104-108): bias + 8 projected tri-plane taps + PE columns on the MFMA ----
    auto bias_init = [&](f32x16 (&acc1)[4]) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (BLK) {           // block kernel: biases are LDS-resident (keep the texture path for the taps)
                    const float4 bl = sBt[8 * m + 2 * q + h];
                    acc1[m][4 * q + 0] = bl.x; acc1[m][4 * q + 1] = bl.y; acc1[m][4 * q + 2] = bl.z; acc1[m][4 * q + 3] = bl.w;
                    continue;
                }
                const auto bb = LDB4(OFF_B1 + 32 * m + 8 * q);
                acc1[m][4 * q + 0] = __uint_as_float(bb[0]); acc1[m][4 * q + 1] = __uint_as_float(bb[1]);
                acc1[m][4 * q + 2] = __uint_as_float(bb[2]); acc1[m][4 * q + 3] = __uint_as_float(bb[3]);
            }
    };
    auto gather = [&](f32x16 (&acc1)[4]) {
        if (!ABL(1)) {
            // sample_from_triplane_new (utils/util.py:359-392): plane0 at (x,y), plane1 at (z,y); zeros padding.
            // Each texel of the prepared planes already carries W1f . texel for this half-wave's 64 hidden units.
            const float lim = a.plim;
            const float gy = qy_ * a.p.nerf_scale[1] + a.p.nerf_trans[1];
            const float iy = ((gy + 1.0f) * 0.5f) * lim;
            float y0f = floorf(iy);
            const float wy1 = iy - y0f, wy0 = 1.0f - wy1;
            y0f = fminf(fmaxf(y0f, -2.f), lim + 2.f);
            const int y0 = (int)y0f, y1 = y0 + 1;
            const bool vy0 = y0 >= 0 && y0 < PR, vy1 = y1 >= 0 && y1 < PR;
            const int cy0 = min(max(y0, 0), PR - 1), cy1 = min(max(y1, 0), PR - 1);
            float tw[8];
            const float4* tp[8];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const float gx = pl ? qz_ * a.p.nerf_scale[2] + a.p.nerf_trans[2] : qx_ * a.p.nerf_scale[0] + a.p.nerf_trans[0];
                const float ix = ((gx + 1.0f) * 0.5f) * lim;
                float x0f = floorf(ix);
                const float wx1 = ix - x0f, wx0 = 1.0f - wx1;
                x0f = fminf(fmaxf(x0f, -2.f), lim + 2.f);
                const int x0 = (int)x0f, x1 = x0 + 1;
                const bool vx0 = x0 >= 0 && x0 < PR, vx1 = x1 >= 0 && x1 < PR;
                const int cx0 = min(max(x0, 0), PR - 1), cx1 = min(max(x1, 0), PR - 1);
                tw[4 * pl + 0] = (vx0 && vy0) ? wx0 * wy0 : 0.f; tw[4 * pl + 1] = (vx1 && vy0) ? wx1 * wy0 : 0.f;
                tw[4 * pl + 2] = (vx0 && vy1) ? wx0 * wy1 : 0.f; tw[4 * pl + 3] = (vx1 && vy1) ? wx1 * wy1 : 0.f;
                // prepared layout: [group of 4 x-adjacent texels][piece h*16 + c][x & 3][4 floats]; piece c of this half is float4 4*c
                const int tg = pl ? HAV_TG1 : HAV_TG;
                const float* plb = a.pplanes + ((size_t)pl * a.p.B + b) * PR * PR * 128 + h * (64 << tg);
                auto texel = [&](int cy, int cx) { return reinterpret_cast<const float4*>(plb + ((size_t)((cy * PR + cx) >> tg)) * (128 << tg) + (cx & ((1 << tg) - 1)) * 4); };
                tp[4 * pl + 0] = texel(cy0, cx0);
                tp[4 * pl + 1] = texel(cy0, cx1);
                tp[4 * pl + 2] = texel(cy1, cx0);
                tp[4 * pl + 3] = texel(cy1, cx1);
            }
            if constexpr (LAB_GATHER_MODEL != 0 && PREC == 3) {          // lab build only (hav_render_lab.h): timing model, wrong results
                lab_gather_model(acc1, tp, tw, lane, a.pplanes, (long long)2 * a.p.B * PR * PR * 128);
                return;
            }
            // 8 taps x 256 B per lane, two taps (32 x 16 B) in flight.  The empty asm pins each tap's FMAs before the
            // loads that recycle its registers: left alone, the compiler hoists all 128 loads and spills them.
            constexpr int NST = 8 * 16 / GQ;          // pipeline stages: stage g covers float4s [(g*GQ)%16, +GQ) of tap (g*GQ)/16 ...
            // ... or, PAIRED (16-load stages): half the pieces of the two x-adjacent taps of a row, alternating between them -- in the
            // grouped layout those two pieces are 16 bytes apart three times out of four, so the second load hits the line the first
            // one just brought in instead of re-walking the group a stage later.  Per accumulator the taps still arrive in order 0..7.
            constexpr bool PAIRED = (GQ == 16) && HAV_TAPPAIR;
            auto tap_of = [](int g, int c) constexpr { return PAIRED ? 2 * (g >> 1) + (c & 1) : (g * GQ) / 16; };
            auto piece_of = [](int g, int c) constexpr { return PAIRED ? 8 * (g & 1) + (c >> 1) : (g * GQ) % 16 + c; };
            auto pstep = [](int tap) constexpr { return 1 << (tap >= 4 ? HAV_TG1 : HAV_TG); };      // float4s between consecutive pieces of a texel
            float4 tv[2][GQ];
#pragma unroll
            for (int c = 0; c < GQ; ++c) { tv[0][c] = tp[tap_of(0, c)][pstep(tap_of(0, c)) * piece_of(0, c)]; tv[1][c] = tp[tap_of(1, c)][pstep(tap_of(1, c)) * piece_of(1, c)]; }
#pragma unroll
            for (int g = 0; g < NST; ++g) {
#pragma unroll
                for (int c = 0; c < GQ; ++c) {     // slot u = 4*c4+e  <->  accumulator (m = u>>4, r = u&15)
                    const int c4 = piece_of(g, c);
                    const float wt = tw[tap_of(g, c)];
                    const float4 t4 = tv[g & 1][c];
                    acc1[c4 >> 2][4 * (c4 & 3) + 0] = fmaf(t4.x, wt, acc1[c4 >> 2][4 * (c4 & 3) + 0]);
                    acc1[c4 >> 2][4 * (c4 & 3) + 1] = fmaf(t4.y, wt, acc1[c4 >> 2][4 * (c4 & 3) + 1]);
                    acc1[c4 >> 2][4 * (c4 & 3) + 2] = fmaf(t4.z, wt, acc1[c4 >> 2][4 * (c4 & 3) + 2]);
                    acc1[c4 >> 2][4 * (c4 & 3) + 3] = fmaf(t4.w, wt, acc1[c4 >> 2][4 * (c4 & 3) + 3]);
                }
                asm volatile("" : "+v"(acc1[0]), "+v"(acc1[1]), "+v"(acc1[2]), "+v"(acc1[3]) : : "memory");
                if (g + 2 < NST) {
#pragma unroll
                    for (int c = 0; c < GQ; ++c) tv[g & 1][c] = tp[tap_of(g + 2, c)][pstep(tap_of(g + 2, c)) * piece_of(g + 2, c)];
                }
            }
        }
    };
    // everything behind the gather: PE, the two dense layers, the head rows, then the caller's epilogue
    auto finish = [&](f32x16 (&acc1)[4]) {
        f32x16 acc2[4];
        float hd0, hd1, hd2, hd3;
        DBG_STAGE(0, acc1);
        TICK(2);
        // ---- PE octaves 4h..4h+3 of this half-wave (model/network/embedder.py:32-61) ---------------------
        float pe[KPE_STEPS];
        if (ABL(2)) {          // timing experiment: no sin / cos
#pragma unroll
            for (int kk = 0; kk < KPE_STEPS; ++kk) pe[kk] = qx_ * (float)(kk + 1);
        } else
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float f = h ? (float)(16 << kk) : (float)(1 << kk);
            pe_pair(qx_ * f, pe[6 * kk + 0], pe[6 * kk + 3]);
            pe_pair(qy_ * f, pe[6 * kk + 1], pe[6 * kk + 4]);
            pe_pair(qz_ * f, pe[6 * kk + 2], pe[6 * kk + 5]);
        }
        __builtin_amdgcn_sched_barrier(0);
        TICK(3);

        if (PREC == 3) {
            if (!ABL(16))
            mfma_split2x<3, false>(acc1, L.sA1, L.sMX, lane, [&](int c, float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = pe[8 * c + e];
            });
        } else if (PREC == 2) {
            if (!ABL(16))
            mfma_split2h<3, 4>(acc1, L.sA1, lane, [&](int c, float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = pe[8 * c + e];
            });
        } else if (PREC == 1) {
            if (!ABL(16))
            mfma_split3<3>(acc1, L.sA1, lane, [&](int c, float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = pe[8 * c + e];
            });
        }
        else if (!ABL(16)) {
            float af[2][4];               // explicit double buffer: the A fragments of k-step t+1 are requested before the MFMAs of step t
#pragma unroll
            for (int m = 0; m < 4; ++m) af[0][m] = sW1[m * 64 + lane];
#pragma unroll
            for (int t = 0; t < KPE_STEPS; ++t) {
                if (t + 1 < KPE_STEPS) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) af[(t + 1) & 1][m] = sW1[((t + 1) * 4 + m) * 64 + lane];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t & 1][m], pe[t], acc1[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        relu_tiles<LEAN>(acc1);
        DBG_STAGE(1, acc1);
        TICK(4);

        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 2: 128 -> 128, relu; B operands are layer-1 accumulator registers, A fragments from LDS ------
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (BLK) {
                    const float4 bl = sBt[32 + 8 * m + 2 * q + h];
                    acc2[m][4 * q + 0] = bl.x; acc2[m][4 * q + 1] = bl.y; acc2[m][4 * q + 2] = bl.z; acc2[m][4 * q + 3] = bl.w;
                    continue;
                }
                const auto bb = LDB4(OFF_B2 + 32 * m + 8 * q);
                acc2[m][4 * q + 0] = __uint_as_float(bb[0]); acc2[m][4 * q + 1] = __uint_as_float(bb[1]);
                acc2[m][4 * q + 2] = __uint_as_float(bb[2]); acc2[m][4 * q + 3] = __uint_as_float(bb[3]);
            }
        if (PREC == 3) {
            if (!ABL(32))
            mfma_split2x<8, true>(acc2, L.sA2, L.sMX + 4 * MX_G_DWORDS, lane, [&](int ch, float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc1[ch >> 1][8 * (ch & 1) + e];
            });
        } else if (PREC == 2) {
            if (!ABL(32))
            mfma_split2h<8, 4>(acc2, L.sA2, lane, [&](int ch, float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc1[ch >> 1][8 * (ch & 1) + e];
            });
        } else if (PREC == 1) {
            if (!ABL(32))
            mfma_split3<8>(acc2, L.sA2, lane, [&](int ch, float (&v)[8]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc1[ch >> 1][8 * (ch & 1) + e];
            });
        } else if (!ABL(32)) {
            float af[2][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[0][m] = sW2[m * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < K2_STEPS; ++ks) {
                if (ks + 1 < K2_STEPS) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) af[(ks + 1) & 1][m] = sW2[((ks + 1) * 4 + m) * 64 + lane];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) acc2[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ks & 1][m], acc1[ks >> 4][ks & 15], acc2[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        relu_tiles<LEAN>(acc2);
        DBG_STAGE(2, acc2);
        TICK(5);

        __builtin_amdgcn_sched_barrier(0);
        // ---- head rows rgb(3, folded fc_rgb o fc_rgbFeat) + alpha: 4 dot products over the 128 hidden units.
        // Each lane holds 64 of its sample's hidden units; the 4 weights per unit are one broadcast ds_read_b128.
        {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 hA = {0.f, 0.f}, hB = {0.f, 0.f};                 // packed FMAs: (rgb0, rgb1) and (rgb2, alpha)
            if (!ABL(64)) {
                if (LEAN) {
#pragma unroll
                for (int mp = 0; mp < 4; ++mp)
#pragma unroll
                    for (int rb = 0; rb < 16; rb += 8) {
                        // 8 broadcast ds_read_b128 in flight, then their 16 packed FMAs: left alone the compiler issues read -> wait -> FMA
                        // 64 times in a row (one full LDS round trip per hidden unit, 6-7 K cycles per tile)
                        typedef float f4v __attribute__((ext_vector_type(4)));
                        f4v w4[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) w4[q] = *reinterpret_cast<const f4v*>(&sW4[(mp * 16 + rb + q) * 2 + h]);
                        // all eight reads must be issued before the first value is consumed
                        asm volatile("" : "+v"(w4[0]), "+v"(w4[1]), "+v"(w4[2]), "+v"(w4[3]), "+v"(w4[4]), "+v"(w4[5]), "+v"(w4[6]), "+v"(w4[7]));
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float v = acc2[mp][rb + q];
                            const f2 vv = {v, v}, wA = {w4[q].x, w4[q].y}, wB = {w4[q].z, w4[q].w};
                            hA = __builtin_elementwise_fma(vv, wA, hA);
                            hB = __builtin_elementwise_fma(vv, wB, hB);
                        }
                    }
                } else {
#pragma unroll
                    for (int mp = 0; mp < 4; ++mp)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float4 w4 = sW4[(mp * 16 + r) * 2 + h];
                            const float v = acc2[mp][r];
                            const f2 vv = {v, v}, wA = {w4.x, w4.y}, wB = {w4.z, w4.w};
                            hA = __builtin_elementwise_fma(vv, wA, hA);
                            hB = __builtin_elementwise_fma(vv, wB, hB);
                        }
                }
            }
            hd0 = hA.x; hd1 = hA.y; hd2 = hB.x; hd3 = hB.y;
        }
        hd0 += half_swap(hd0, h); hd1 += half_swap(hd1, h);
        hd2 += half_swap(hd2, h); hd3 += half_swap(hd3, h);
        if (BLK) {          // block kernel: LDS copy (a global load here queued behind the other waves' tap loads in every tile: ~1 K cycles)
            int b4_off = 64;          // its own opaque offset: through sBt the base of the bias reads would stay live across both layers
            asm volatile("" : "+v"(b4_off));
            const float4 b4 = L.sB[b4_off];
            hd0 += b4.x; hd1 += b4.y; hd2 += b4.z; hd3 += b4.w;
        } else {
            const auto b4 = __builtin_amdgcn_raw_buffer_load_b128(L.wrs, 0, OFF_B4 * 4, 0);
            hd0 += __uint_as_float(b4[0]); hd1 += __uint_as_float(b4[1]); hd2 += __uint_as_float(b4[2]); hd3 += __uint_as_float(b4[3]);
        }
        TICK(6);
        DBG_PUT(3, (hd0 + hd1) + (hd2 + hd3));
        cont(acc2, hd0, hd1, hd2, hd3);
    };
    f32x16 acc1[4];
    bias_init(acc1);
    gather(acc1);
    // Wave priority (round 6): the two waves of a SIMD are mostly in different phases; the one that is past its gather -- positional encoding, the dense
    // layers, the head rows, the caller's compositing / parking -- goes first at the issue port, the one still collecting taps (its FMAs wait for loads
    // anyway) yields.  Same instructions, same results bit for bit; -1.2 ... -1.8 % kernel time over interleaved same-box rounds of 40 launches
    // (profiles/r06_prio_ab.txt: levels 1, 2 and 3, the dense layers alone or everything behind the gather measure the same; priority on the gather or
    // geometry side, and a static level for the second-dispatched half of the workgroup alone, measure nothing).
    __builtin_amdgcn_s_setprio(HAV_PRIO_DENSE);
    finish(acc1);
    __builtin_amdgcn_s_setprio(0);
}
#undef LDB4

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
#define MARCH_THREADS 512
#define MARCH_WAVES (MARCH_THREADS / 64)
#define RACC_N 136  // 0..127 composited hidden units | 128..130 rgb | 131 depth | 132 acc | 133 wmax
#define R_RGB 128
#define R_DEPTH 131
#define R_ACC 132
#define R_WMAX 133

typedef float f32x2 __attribute__((ext_vector_type(2)));

// exchange with lane^4 inside a DPP row: banks {0,2} read from lane+4, banks {1,3} from lane-4
__device__ __forceinline__ float dpp_xor4(float v)
{
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104 /*row_shl:4*/, 0xf, 0x5, false);
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(v), 0x114 /*row_shr:4*/, 0xf, 0xA, false);
    return __int_as_float(r);
}

// RANDOM=false: perturb == 0 and noise_std == 0 (deterministic rendering, the parity configuration); the random-number
// paths (injected tensors or Philox) are compiled out.
template <bool RANDOM>
__global__ void __launch_bounds__(MARCH_THREADS, MARCH_THREADS / 256) hav_march_f32_kernel(const MarchArgs a)
{
    constexpr int RM = RANDOM ? 2 : 0;
    const uint32_t call_off = RANDOM ? rng_call_off(a) : 0u;      // one memory read per kernel, not per draw
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const float* sW1 = smem + OFF_W1PE;
    const float* sW2 = smem + OFF_W2;
    const float4* sW4 = reinterpret_cast<const float4*>(smem + OFF_W4);
    const float* sWFT = smem + OFF_WFT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* scr = smem + LDS_FLOATS + wave * a.scr_floats;
    float* s_w = scr + a.o_w;       // [2][s_pad_c]  coarse weights
    float* s_cdf = scr + a.o_cdf;   // [2][s_pad_c]
    float* s_cand = scr + a.o_cand; // [2][s_pad_f]  unsorted fine depths
    float* s_zf = scr + a.o_zf;     // [2][s_pad_f]  sorted fine depths
    float* s_racc = scr + a.o_racc; // [2][RACC_N]

    {   // every weight the tile loop touches becomes LDS-resident (fragment order, straight copy)
        const float4* src = reinterpret_cast<const float4*>(a.blob);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = tid; i < LDS_FLOATS / 4; i += MARCH_THREADS) dst[i] = src[i];
    }
    __syncthreads();

    const int j = lane & 31, h = lane >> 5, col = lane & 15, rowt = (lane >> 4) & 1;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.blob), 0, BLOB_FLOATS * 4, 0x00020000);
    LaneCtx L;
    L.sW1 = sW1; L.sW2 = sW2; L.sW4 = sW4; L.wrs = wrs; L.lane = lane; L.h = h; L.hoff = h * 16;

    const int S_c = a.p.S_c, S_fp = a.S_fp;
    const long long NR = a.NR;
    const long long npairs = (NR + 1) >> 1;

    // XCD-aware distribution: workgroup b runs on XCD b%8 (observed, used for L2 locality only); each XCD
    // owns one contiguous eighth of the ray pairs, its workgroups' waves sweep it in lockstep.
    long long chunk, base, span;
    int lb, nbx;
    if ((gridDim.x & 7) == 0) {
        chunk = (npairs + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        lb = blockIdx.x >> 3; nbx = gridDim.x >> 3;
        base = chunk * xcd;
        span = npairs - base; if (span > chunk) span = chunk; if (span < 0) span = 0;
    } else { base = 0; span = npairs; lb = blockIdx.x; nbx = gridDim.x; }

    for (long long local = (long long)lb * MARCH_WAVES + wave; local < span; local += (long long)nbx * MARCH_WAVES) {
        const long long pair = base + local;
        const long long ray0 = pair * 2;
        const bool has1 = (ray0 + 1) < NR;
        if (!has1) {     // odd ray count: slot 1 idles on finite dummy data
            for (int i = lane; i < a.s_pad_c; i += 64) s_w[a.s_pad_c + i] = 0.f;
        }

        for (int pass = 0; pass < (S_fp > 0 ? 2 : 1); ++pass) {
            const int S = pass == 0 ? S_c : S_fp;
            const int nr = (S + 15) >> 4;         // DPP rows per ray
            const int ntiles = nr;                // 2 rays * nr rows / 2 rows per tile
            for (int i = lane; i < 2 * RACC_N; i += 64) s_racc[i] = 0.f;
            float carry = 1.0f;
            wave_lds_sync();

            for (int tile = 0; tile < ntiles; ++tile) {
                // ---- which sample does this lane evaluate -------------------------------------------
                const int rowg = 2 * tile + rowt;
                const int slot = rowg >= nr ? 1 : 0;
                const int srow = rowg - slot * nr;
                int s = srow * 16 + col;
                const bool rayok = slot == 0 || has1;
                const bool valid = (s < S) && rayok;
                if (s > S - 1) s = S - 1;
                const long long gr = rayok ? ray0 + slot : ray0;
                const RKey rkey = RANDOM ? rng_ray_key(a, gr, call_off) : RKey{0u, 0u, 0u, a.step_c};
                const int b = (int)(gr / a.p.R);
                const float* ray = a.rays + gr * a.p.ray_stride;
                const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
                const float near = ray[6], far = ray[7];

                float z, znb;      // depth of this sample and of the neighbour that defines `dists`
                const int snb = (s + 1 < S) ? s + 1 : S - 2;       // utils/nerf_util.py:36-37 (last dist repeated)
                if (pass == 0) {
                    z = z_coarse<RM>(a, gr, rkey, s, near, far);
                    znb = z_coarse<RM>(a, gr, rkey, snb, near, far);
                } else {
                    z = s_zf[slot * a.s_pad_f + s];
                    znb = s_zf[slot * a.s_pad_f + snb];
                }
                const float dist = (s + 1 < S) ? (znb - z) : (z - znb);

                PROF_DECL();
                sample_eval<16, 0, false>(a, L, b, ox, oy, oz, dx, dy, dz, z, [&](f32x16 (&acc2)[4], float hd0, float hd1, float hd2, float hd3) {
                __builtin_amdgcn_sched_barrier(0);
                // ---- volume_render_radiance_field (utils/nerf_util.py:28-73) -------------------------------
                float sg = hd3;
                if (RANDOM && a.p.noise_std > 0.f) {
                    const float* nz = pass == 0 ? a.noise_c : a.noise_f;
                    const float e = nz ? nz[gr * S + s] : rng_normal(a, rkey, gr, s, pass == 0 ? STREAM_EPS_C : STREAM_EPS_F);
                    sg += e * a.p.noise_std;
                }
                sg = fmaxf(sg, 0.f);
                const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
                float alpha = 1.0f - expf(-sg * (dist * dn));
                if (!valid) alpha = 0.f;
                const float om = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
                const float pin = row16_scan_mul(om);                        // inclusive product within the row
                const float pex = dpp_mov<DPP_ROW_SHR(1)>(1.0f, pin);        // exclusive
                const float tot0 = read_lane(pin, 15), tot1 = read_lane(pin, 31);
                const bool first0 = (2 * tile) % nr == 0, first1 = (2 * tile + 1) % nr == 0;
                const float cin0 = first0 ? 1.0f : carry;
                const float cin1 = first1 ? 1.0f : cin0 * tot0;
                carry = cin1 * tot1;
                const float wgt = alpha * ((rowt ? cin1 : cin0) * pex);         // :60
                if (pass == 0 && h == 0 && valid) s_w[slot * a.s_pad_c + s] = wgt;

                // ---- weighted sums over the samples of each row: reduce-scatter across the 16 lanes of the DPP row.
                // 64 products per lane -> 4 per lane after xor-8/4/2/1 exchanges; lane `col` ends up with hidden units
                // [4*(2*col + h) .. +3] (64*b3 + 32*b2 + 16*b1 + 8*b0 + 4h), i.e. one aligned float4 of the LDS accumulator.
                const bool same = (2 * tile) / nr == (2 * tile + 1) / nr;    // both rows of this tile belong to one ray
                float* racc = s_racc + slot * RACC_N;
                if (!ABL(8)) {
                    const bool b3 = (col & 8) != 0, b2 = (col & 4) != 0, b1 = (col & 2) != 0, b0 = (col & 1) != 0;
                    float v1[32];
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        const float lo = acc2[i >> 4][i & 15] * wgt, hi = acc2[(i + 32) >> 4][i & 15] * wgt;
                        v1[i] = (b3 ? hi : lo) + dpp_mov<0x128 /*row_ror:8*/>(0.f, b3 ? lo : hi);
                    }
                    float v2[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v2[i] = (b2 ? v1[i + 16] : v1[i]) + dpp_xor4(b2 ? v1[i] : v1[i + 16]);
                    float v3[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v3[i] = (b1 ? v2[i + 8] : v2[i]) + dpp_mov<DPP_QUAD_XOR2>(0.f, b1 ? v2[i] : v2[i + 8]);
                    float v4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v4[i] = (b0 ? v3[i + 4] : v3[i]) + dpp_mov<DPP_QUAD_XOR1>(0.f, b0 ? v3[i] : v3[i + 4]);
                    // rows are always added in increasing row order, one at a time, so a ray's sums do not depend on which
                    // slot of the pair (or which call) it was rendered in: results are bit-reproducible
                    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
                    if (same) { o0 = __shfl_xor(v4[0], 16, 64); o1 = __shfl_xor(v4[1], 16, 64); o2 = __shfl_xor(v4[2], 16, 64); o3 = __shfl_xor(v4[3], 16, 64); }
                    if (!same || rowt == 0) {
                        float4* dst = reinterpret_cast<float4*>(racc + 8 * col + 4 * h);
                        float4 cur = *dst;
                        cur.x = (cur.x + v4[0]) + o0; cur.y = (cur.y + v4[1]) + o1; cur.z = (cur.z + v4[2]) + o2; cur.w = (cur.w + v4[3]) + o3;
                        *dst = cur;
                    }
                }
                {
                    const bool writer = (col == 0) && (!same || rowt == 0);
                    float e0, e1, e2, e3, e4;
                    if (h == 0) {
                        e0 = sigmoid_rcp(hd0);              // sigmoid on rgb only (:45-46)
                        e1 = sigmoid_rcp(hd1);
                        e2 = sigmoid_rcp(hd2);
                        e3 = z; e4 = 1.0f;
                    } else { e0 = e1 = e2 = e3 = e4 = 0.f; }
                    const float v0 = row16_sum(e0 * wgt), v1 = row16_sum(e1 * wgt), v2 = row16_sum(e2 * wgt);
                    const float v3 = row16_sum(e3 * wgt), v4 = row16_sum(e4 * wgt);
                    float v5 = row16_max(wgt);
                    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f, u4 = 0.f;
                    if (same) {
                        u0 = __shfl_xor(v0, 16, 64); u1 = __shfl_xor(v1, 16, 64); u2 = __shfl_xor(v2, 16, 64);
                        u3 = __shfl_xor(v3, 16, 64); u4 = __shfl_xor(v4, 16, 64); v5 = fmaxf(v5, __shfl_xor(v5, 16, 64));
                    }
                    if (writer && h == 0) {
                        racc[R_RGB + 0] = (racc[R_RGB + 0] + v0) + u0; racc[R_RGB + 1] = (racc[R_RGB + 1] + v1) + u1;
                        racc[R_RGB + 2] = (racc[R_RGB + 2] + v2) + u2;
                        racc[R_DEPTH] = (racc[R_DEPTH] + v3) + u3; racc[R_ACC] = (racc[R_ACC] + v4) + u4;
                        racc[R_WMAX] = fmaxf(racc[R_WMAX], v5);
                    }
                }
                } PROF_PASS DBG_PASS(DBG_NONE()));
                wave_lds_sync();
            }   // tiles

            // ---- write this pass's outputs: rgb | fc_rgbFeat applied ONCE per ray to the composited hidden units ----
            {
                const float* ra = s_racc;
                const float* rb = s_racc + RACC_N;
                const float bfv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, OFF_BF * 4, 0));
                float ga = bfv * ra[R_ACC], gb = bfv * rb[R_ACC];         // lane f: bf[f] * sum_s w_s
#pragma unroll 4
                for (int k4 = 0; k4 < HAV_HID / 4; ++k4) {
                    const float4 ha = *reinterpret_cast<const float4*>(ra + 4 * k4), hb = *reinterpret_cast<const float4*>(rb + 4 * k4);
                    const float wq0 = sWFT[(4 * k4 + 0) * 64 + lane], wq1 = sWFT[(4 * k4 + 1) * 64 + lane];
                    const float wq2 = sWFT[(4 * k4 + 2) * 64 + lane], wq3 = sWFT[(4 * k4 + 3) * 64 + lane];
                    ga = fmaf(wq0, ha.x, ga); ga = fmaf(wq1, ha.y, ga); ga = fmaf(wq2, ha.z, ga); ga = fmaf(wq3, ha.w, ga);
                    gb = fmaf(wq0, hb.x, gb); gb = fmaf(wq1, hb.y, gb); gb = fmaf(wq2, hb.z, gb); gb = fmaf(wq3, hb.w, gb);
                }
                for (int slot = 0; slot < (has1 ? 2 : 1); ++slot) {
                    const long long gr = ray0 + slot;
                    const RKey rkey = RANDOM ? rng_ray_key(a, gr, call_off) : RKey{0u, 0u, 0u, a.step_c};
                    const float* racc = s_racc + slot * RACC_N;
                    const float accv = racc[R_ACC];
                    float* rgb = (pass == 0 ? a.out.rgb_coarse : a.out.rgb_fine) + gr * 67;
                    rgb[3 + lane] = slot ? gb : ga;
                    if (lane < 3) {
                        float v = racc[R_RGB + lane];
                        if (a.bg) v = v + (1.0f - accv) * a.bg[gr * 3 + lane];     // :70-71
                        rgb[lane] = v;
                    }
                    if (lane == 0) {
                        (pass == 0 ? a.out.depth_coarse : a.out.depth_fine)[gr] = racc[R_DEPTH];
                        (pass == 0 ? a.out.acc_coarse : a.out.acc_fine)[gr] = accv;
                        if (pass == 1 || S_fp == 0) a.out.weights_max[gr] = racc[R_WMAX];    // model/nerf_trainer.py:195,200
                    }
                }
            }

            // ---- inverse-CDF resampling + merge (utils/nerf_util.py:76-117, model/nerf_trainer.py:166-170) ---
            if (pass == 0 && S_fp > 0 && !ABL(128)) {
                const int slot = h, li = j;                       // half-wave h prepares ray slot h
                const long long gr = (slot == 0 || has1) ? ray0 + slot : ray0;
                const RKey rkey = RANDOM ? rng_ray_key(a, gr, call_off) : RKey{0u, 0u, 0u, a.step_c};
                const float* ray = a.rays + gr * a.p.ray_stride;
                const float near = ray[6], far = ray[7];
                const int nb = S_c - 1, nw = S_c - 2, S_half = (S_c + 1) >> 1, S_f = a.p.S_f;
                float* w = s_w + slot * a.s_pad_c;
                float* cdf = s_cdf + slot * a.s_pad_c;
                float* cand = s_cand + slot * a.s_pad_f;
                float* zf = s_zf + slot * a.s_pad_f;
                if (li == 0) {      // sequential sum / cumsum in the reference's order (SURVEY B-11)
                    float sum = 0.f;
                    for (int i = 0; i < nw; ++i) sum += (w[1 + i] + 1e-5f);
                    float run = 0.f;
                    cdf[0] = 0.f;
                    for (int i = 0; i < nw; ++i) { run += (w[1 + i] + 1e-5f) / sum; cdf[i + 1] = run; }
                }
                wave_lds_sync();
                for (int k = li; k < S_f; k += 32) {
                    float u;
                    if (!RANDOM || !a.p.perturb) {       // det: torch.linspace(0,1,S_f)
                        const float st = a.u_st;
                        u = (S_f == 1) ? 0.f : ((k < S_f / 2) ? st * (float)k : 1.0f - st * (float)(S_f - 1 - k));
                    } else {
                        const float zeta = a.u_rand ? a.u_rand[gr * S_f + k] : rng_uniform(a, rkey, k, STREAM_ZETA);
                        u = (float)k * a.u_sN + zeta * a.u_w;
                    }
                    int inds = 0;                                  // searchsorted(cdf, u, right=True)
                    for (int i = 0; i < nb; ++i) inds += (cdf[i] <= u) ? 1 : 0;
                    const int below = max(inds - 1, 0), above = min(inds, nb - 1);
                    float dnm = cdf[above] - cdf[below];
                    if (dnm < 1e-5f) dnm = 1.0f;
                    const float tt = (u - cdf[below]) / dnm;
                    const float bl = 0.5f * (z_coarse<RM>(a, gr, rkey, below + 1, near, far) + z_coarse<RM>(a, gr, rkey, below, near, far));
                    const float ba = 0.5f * (z_coarse<RM>(a, gr, rkey, above + 1, near, far) + z_coarse<RM>(a, gr, rkey, above, near, far));
                    cand[S_half + k] = bl + tt * (ba - bl);
                }
                for (int i = li; i < S_half; i += 32) cand[i] = z_coarse<RM>(a, gr, rkey, 2 * i, near, far);   // z_vals[:, ::2]
                wave_lds_sync();
                for (int e = li; e < S_fp; e += 32) {              // rank sort == torch.sort on 48 values
                    const float v = cand[e];
                    int rank = 0;
                    for (int q = 0; q < S_fp; ++q) {
                        const float o = cand[q];
                        rank += ((o == o) ? ((v != v) || o < v || (o == v && q < e)) : ((v != v) && q < e)) ? 1 : 0;      // total order, NaN last (torch.sort)
                    }
                    zf[rank] = v;
                    if (a.dbg_zfine && (slot == 0 || has1)) a.dbg_zfine[gr * S_fp + rank] = v;
                }
                wave_lds_sync();
            }
        }   // pass
    }       // pairs
}

// ------------------------------------------------------------------------------------------------
// Block kernel: a wave owns 32 CONSECUTIVE rays (neighbouring pixels) and walks them through the sample index together:
// tile = sample s of 32 rays.  Neighbouring rays hit the same texels, so the 64 lanes of every gather load touch a handful
// of cache lines instead of 64 (the pair kernel's gather is texture-addresser bound), and compositing needs no cross-lane
// work at all: transmittance, the composited hidden units, colour, depth and opacity are per-lane recurrences over s, summed
// in the reference's sequential order.  fc_rgbFeat is applied to the composited hidden units on the matrix cores once per
// 32 rays.  The coarse weights a ray needs for its inverse CDF are parked in that ray's own (not yet written) rgb_fine
// row; the 16 importance samples per ray live in LDS and are merged with the even coarse depths on the fly.
// Requires S_c <= 67 when a fine pass is requested (host falls back to the pair kernel otherwise).
// ------------------------------------------------------------------------------------------------
// CACHE: the merged fine list of a ray contains its even coarse samples (z_vals[::2], model/nerf_trainer.py:170), whose radiance
// field values the coarse pass has already computed.  With a workspace, the coarse pass parks relu(h2) (64 values per lane)
// and the 4 raw head values of every even sample, the fine pass evaluates only the S_f NEW samples (parking them too), then
//   A. sweeps the merged list in depth order per ray with the parked head values only (transmittance, colour, depth, opacity:
//      the reference's sequential order, unchanged) and records each entry's compositing weight,
//   B. accumulates sum_e w_e h2_e over the parked entries in ENTRY order -- the same entry for all 32 rays, so every load is a
//      coalesced 1-KB row (the feature sum is order-independent up to fp32 rounding; the depth-ordered scalars are not touched).
// 80 instead of 112 field evaluations per ray.  Slot layout per wave (floats): H2 [S_fp][16][64 lanes][4] | RAW [S_fp][32][4] |
// WV [S_fp][32].  The slot is re-used block after block by the same wave; agent-scope fences order its stores and loads and
// drop stale L1 lines.
// streaming accesses to the parked hidden units: non-temporal so that 13 GB per frame do not evict the tri-plane texels from L2
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4* p, float4 v)
{
    nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
}
__device__ __forceinline__ float4 nt_load4(const float4* p)
{
    const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
#define HAV_GQ2 16               // float4 loads per gather stage in the fine-maps-only variant (its register budget allows a whole tap: -2 %)
#define HAV_RESAMPLE_PF 8        // coarse weights in flight in the resampling sweep
#define HAV_LDS_BIAS 264         // LDS copy of b1 | b2 | b4 (folded rgb biases, alpha bias) | pad, behind the weight image
#define WS_H2_FLOATS 4096
#define WS_ENTRY_FLOATS (WS_H2_FLOATS + 128 + 32)
// CACHE = 2: additionally, the caller does not want the coarse pass's composited outputs (Trainer.forward with a fine pass only
// uses the fine ones): the coarse pass then carries no composited-hidden-unit accumulators at all (64 VGPRs less in its loop).
template <int RANDOM, int PREC, int CACHE>
__global__ void __launch_bounds__(MARCH_THREADS, MARCH_THREADS / 256) hav_march_blk_kernel(const MarchArgs a)
{
    constexpr int RM = RANDOM;
    constexpr bool COUT = CACHE != 2;
    // fp16 mode: the fc_rgbFeat fragments fit in LDS next to the layer weights, so what is parked per sample is its 64 FEATURES
    // (8 rows of 1 KB per wave) instead of its 128 hidden units (16 rows): half the parking traffic for 48 more MFMAs per parked tile
    constexpr bool FEATPARK = (PREC == 2) && (CACHE != 0);
    constexpr int H2F = FEATPARK ? WS_H2_FLOATS / 2 : WS_H2_FLOATS, ENTF = H2F + 128 + 32;      // floats of one parked entry
    if (a.guard) {          // fp16 range guard: the fp16 kernel and its bf16 fallback are both launched, one of them proceeds
        const unsigned int unsafe = __builtin_amdgcn_readfirstlane(*a.guard);
        if ((unsafe != 0) != (a.guard_run_if != 0)) return;
        if (a.guard_run_if && a.status && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, HAV_STATUS_FP16_FALLBACK);
    }
    const uint32_t call_off = RANDOM ? rng_call_off(a) : 0u;      // one memory read per kernel, not per draw
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WLDS = PREC == 3 ? LDSX_FLOATS : (PREC == 2 ? LDSH_FLOATS : (PREC == 1 ? LDS3_FLOATS : LDS_FLOATS));     // LDS image: fp32 | split-bf16 | split-fp16 fragments | fp16 fragments + MX records
    const float* sWFF = smem + OFF_WFT;           // PREC 0: fc_rgbFeat fragments live in the WFT slot (PREC 1 streams them from L2)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* s_n = smem + WLDS + HAV_LDS_BIAS + wave * a.scr_floats;      // [S_f][32] importance samples of this wave's rays (after the b1|b2|b4 copy)
    if (PREC == 3) {          // [A1H | A2H | W4H] as in the fp16 mode, then the MX records (stored behind the range-guard words in the blob)
        const float4* src = reinterpret_cast<const float4*>(a.blob + OFF_A1H);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = tid; i < LDSX_FP16 / 4; i += MARCH_THREADS) dst[i] = src[i];
        const float4* srcx = reinterpret_cast<const float4*>(a.blob + OFF_MX);
        float4* dstx = reinterpret_cast<float4*>(smem + LDSX_FP16);
        for (int i = tid; i < MX_GROUPS * MX_G_DWORDS / 4; i += MARCH_THREADS) dstx[i] = srcx[i];
    } else if (PREC >= 1) {
        const float4* src = reinterpret_cast<const float4*>(a.blob + (PREC == 2 ? OFF_A1H : OFF_A1S));
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = tid; i < WLDS / 4; i += MARCH_THREADS) dst[i] = src[i];
    } else {
        const float4* src = reinterpret_cast<const float4*>(a.blob);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = tid; i < OFF_WFT / 4; i += MARCH_THREADS) dst[i] = src[i];
        const float4* srcf = reinterpret_cast<const float4*>(a.blob + OFF_WFF);
        float4* dstf = reinterpret_cast<float4*>(smem + OFF_WFT);
        for (int i = tid; i < K2_STEPS * 2 * 64 / 4; i += MARCH_THREADS) dstf[i] = srcf[i];
    }
    if (tid < 65) reinterpret_cast<float4*>(smem + WLDS)[tid] = reinterpret_cast<const float4*>(a.blob + OFF_B1)[tid];   // b1 | b2 | b4 (contiguous in the blob)
    __syncthreads();

    PROF_DECL();
    PROF_BEGIN();
    const int j = lane & 31, h = lane >> 5;
    LaneCtx L;
    L.sW1 = smem + OFF_W1PE; L.sW2 = smem + OFF_W2;
    L.sW4 = reinterpret_cast<const float4*>(smem + (PREC >= 2 ? (OFF_W4H - OFF_A1H) : (PREC == 1 ? (OFF_W4S - OFF_A1S) : OFF_W4)));
    L.sA1 = reinterpret_cast<const uint4*>(smem);
    L.sA2 = reinterpret_cast<const uint4*>(smem + (PREC >= 2 ? (OFF_A2H - OFF_A1H) : (OFF_A2S - OFF_A1S)));
    L.sAF = reinterpret_cast<const uint4*>(smem + (OFF_AFH - OFF_A1H));
    L.sMX = reinterpret_cast<const unsigned int*>(smem + LDSX_FP16);
    L.sB = reinterpret_cast<const float4*>(smem + WLDS);
    L.wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.blob), 0, BLOB_FLOATS * 4, 0x00020000);
    L.lane = lane; L.h = h; L.hoff = h * 16;

    const int S_c = a.p.S_c, S_f = a.p.S_f, S_fp = a.S_fp, S_half = (S_c + 1) >> 1;
    const int R = a.p.R;
    const int bpf = (R + 31) >> 5;                       // blocks per frame: a block never straddles two frames
    const long long nblk = (long long)bpf * a.p.B;

    LAB_STAGGER(wave);
    long long chunk, base, span;
    int lb, nbx;
    if ((gridDim.x & 7) == 0) {
        chunk = (nblk + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        lb = blockIdx.x >> 3; nbx = gridDim.x >> 3;
        base = chunk * xcd;
        span = nblk - base; if (span > chunk) span = chunk; if (span < 0) span = 0;
    } else { base = 0; span = nblk; lb = blockIdx.x; nbx = gridDim.x; }

    for (long long local = (long long)lb * MARCH_WAVES + wave; local < span; local += (long long)nbx * MARCH_WAVES) {
        const long long blk = base + local;
        const int b = (int)(blk / bpf);
        const int r0 = (int)(blk - (long long)b * bpf) * 32 + j;
        const bool rayok = r0 < R;
        long long gr = (long long)b * R + (rayok ? r0 : R - 1);
        const float* ray = a.rays + gr * a.p.ray_stride;
        const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
        float near = ray[6], far = ray[7];
        const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
        float* wpark = a.out.rgb_fine ? a.out.rgb_fine + gr * 67 : nullptr;   // this ray's parking row for w[0..S_c)
        RKey rkey = RANDOM ? rng_ray_key(a, gr, call_off) : RKey{0u, 0u, 0u, a.step_c};
        // Register hygiene: the phases of a block (sample loop | resampling | stage 1 | A | B | stores) all start from these four
        // per-ray values.  Left transparent, the compiler shares sub-expressions BETWEEN phases (coarse depths of fixed indices,
        // output addresses, hash pieces): dozens of values that then live -- spilled -- through the sample loops.  An opaque fence
        // at each phase boundary makes every phase recompute its own (a handful of ALU ops per block).
#define RAY_FENCE() asm volatile("" : "+v"(near), "+v"(far), "+v"(gr), "+v"(rkey.ray), "+v"(rkey.step))
        RAY_FENCE();
        float* slot = CACHE ? a.ws + (ABL(1024) ? 0 : a.ws_slot * ((long long)blockIdx.x * MARCH_WAVES + wave)) : nullptr;   // 1024: timing experiment, all waves share one L2-resident slot
        // coarse weights w[0..S_c) of the block's rays for the inverse CDF: with a workspace, one coalesced 128-byte row per sample in
        // this wave's own slot (wrow: written and read by the same wave, plain cached loads -- a wave's stores are coherent with its own
        // CU's L1); without (CACHE == 0, not the production path), the ray's (not yet written) rgb_fine row (wpark).  rgb_fine rows of
        // neighbouring rays share cache lines ACROSS waves, and an L1 line fetched by a neighbour before this wave's stores could be
        // stale on another CU only -- the row is private to the ray, so it is the same-CU case too, but the wpark reads stay
        // L1-bypassing (non-temporal): that path is outside the determinism stress runs.
        float* wrow = CACHE ? slot + (size_t)a.S_fp * ENTF : nullptr;
        auto park = [&](int e, const f32x16 (&v)[4], float r0, float r1, float r2, float r3) {       // entry e of this block's slot
            float4* H2 = reinterpret_cast<float4*>(slot + (size_t)e * ENTF);
            if (FEATPARK) {
                f32x16 ft[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ft[m][r] = 0.f;
                if (!ABL(4))
                mfma_split2h<8, 2>(ft, L.sAF, lane, [&](int ch, float (&x)[8]) {
#pragma unroll
                    for (int el = 0; el < 8; ++el) x[el] = v[ch >> 1][8 * (ch & 1) + el];
                });
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        nt_store4(&H2[(m * 4 + rg) * 64 + lane], make_float4(ft[m][4 * rg + 0], ft[m][4 * rg + 1], ft[m][4 * rg + 2], ft[m][4 * rg + 3]));
            } else
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    nt_store4(&H2[(m * 4 + rg) * 64 + lane], make_float4(v[m][4 * rg + 0], v[m][4 * rg + 1], v[m][4 * rg + 2], v[m][4 * rg + 3]));
            if (h == 0) reinterpret_cast<float4*>(slot + (size_t)e * ENTF + H2F)[j] = make_float4(r0, r1, r2, r3);
        };
        // fc_rgbFeat on the composited hidden units ([64 x 128] . [128 x 32 rays] on the matrix cores) + the per-ray output stores
        auto emit = [&](int pass, const f32x16 (&hs)[4], float c0, float c1, float c2, float dep, float accw, float wmax,
                        const f32x16* fs = nullptr) {          // fs: composited features (feature parking) instead of hidden units
            // ---- fc_rgbFeat on the composited hidden units: [64 x 128] . [128 x 32 rays] on the matrix cores -------------
            f32x16 og[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const auto bb = __builtin_amdgcn_raw_buffer_load_b128(L.wrs, L.hoff, (OFF_BF + 32 * m + 8 * q) * 4, 0);
                    og[m][4 * q + 0] = __uint_as_float(bb[0]) * accw; og[m][4 * q + 1] = __uint_as_float(bb[1]) * accw;   // bf * sum_s w_s
                    og[m][4 * q + 2] = __uint_as_float(bb[2]) * accw; og[m][4 * q + 3] = __uint_as_float(bb[3]) * accw;
                }
            if (fs) {
#pragma unroll
                for (int m = 0; m < 2; ++m) og[m] += fs[m];
            } else if (PREC >= 1) {          // fragments stream from L2 (once per 32 rays per pass), 8 k-steps in flight
                float wf[2][16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wf[0][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(L.wrs, lane * 4, (OFF_WFF + u * 64) * 4, 0));
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    if (g + 1 < 8) {
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            wf[(g + 1) & 1][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(L.wrs, lane * 4, (OFF_WFF + ((g + 1) * 16 + u) * 64) * 4, 0));
                    }
#pragma unroll
                    for (int kq = 0; kq < 8; ++kq) {
                        const int ks = g * 8 + kq;
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            og[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[g & 1][kq * 2 + m], hs[ks >> 4][ks & 15], og[m], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < K2_STEPS; ++ks) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        og[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(sWFF[(ks * 2 + m) * 64 + lane], hs[ks >> 4][ks & 15], og[m], 0, 0, 0);
                }
            }
            if (rayok && (pass == 1 || a.out.rgb_coarse)) {          // coarse outputs are optional when there is a fine pass
                float* rgb = (pass == 0 ? a.out.rgb_coarse : a.out.rgb_fine) + gr * 67;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) rgb[3 + acc_row(m, r, h)] = og[m][r];
                if (h == 0) {
                    const float bgx = a.bg ? a.bg[gr * 3 + 0] : 0.f, bgy = a.bg ? a.bg[gr * 3 + 1] : 0.f, bgz = a.bg ? a.bg[gr * 3 + 2] : 0.f;
                    rgb[0] = a.bg ? c0 + (1.0f - accw) * bgx : c0;              // :70-71
                    rgb[1] = a.bg ? c1 + (1.0f - accw) * bgy : c1;
                    rgb[2] = a.bg ? c2 + (1.0f - accw) * bgz : c2;
                    (pass == 0 ? a.out.depth_coarse : a.out.depth_fine)[gr] = dep;
                    (pass == 0 ? a.out.acc_coarse : a.out.acc_fine)[gr] = accw;
                    if (pass == 1 || S_fp == 0) a.out.weights_max[gr] = wmax;  // model/nerf_trainer.py:195,200
                }
            }
        };

        for (int pass = 0; pass < (S_fp > 0 ? 2 : 1); ++pass) {
            const int S = pass == 0 ? S_c : S_fp;
            f32x16 hsum[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) hsum[m][r] = 0.f;
            float T = 1.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f, dep = 0.f, accw = 0.f, wmax = 0.f;
            // merged fine depths are produced on the fly: even coarse depths and the LDS-resident importance samples
            int ie = 0, ik = 0;
            float ze = 0.f, nk = 0.f;
            auto next_fine = [&]() -> float {
                const bool take_e = (ie < S_half) && (ik >= S_f || ze <= nk);
                const float v = take_e ? ze : nk;
                if (take_e) { ++ie; ze = (ie < S_half) ? z_coarse<RM>(a, gr, rkey, 2 * ie, near, far) : 3.0e38f; }
                else { ++ik; nk = (ik < S_f) ? s_n[ik * 32 + j] : 3.0e38f; }
                return v;
            };
            float z, znext;
            if (CACHE && pass == 1) {
                // ---- stage 1: the S_f new samples (same k for all 32 rays), parked behind the even coarse entries ----
                for (int k = 0; k < S_f; ++k) {
                    TICK(0);
                    const float zk = s_n[k * 32 + j];
                    sample_eval<(CACHE == 2 ? HAV_GQ2 : 8), PREC, true, CACHE == 2>(a, L, b, ox, oy, oz, dx, dy, dz, zk,
                                                                 [&](f32x16 (&acc2)[4], float hd0, float hd1, float hd2, float hd3) {
                        __builtin_amdgcn_sched_barrier(0);
                        park(S_half + k, acc2, hd0, hd1, hd2, hd3);
                    } PROF_PASS DBG_PASS(DBG_TILE(S_c + k)));
                    TICK(7);
                }
                RAY_FENCE();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                // ---- stage A: depth-ordered sweep over the merged list with the parked head values -------------------
                // The entry order of a ray depends only on depths, not on loaded data: a FIFO of 8 (depth, entry, head values)
                // runs ahead of the consumer, so 8 of the lane-divergent 16-byte reads are in flight instead of one.
                auto next_entry = [&](int& eid) -> float {
                    const bool take_e = (ie < S_half) && (ik >= S_f || ze <= nk);
                    const float v = take_e ? ze : nk;
                    eid = take_e ? ie : S_half + ik;
                    if (take_e) { ++ie; ze = (ie < S_half) ? z_coarse<RM>(a, gr, rkey, 2 * ie, near, far) : 3.0e38f; }
                    else { ++ik; nk = (ik < S_f) ? s_n[ik * 32 + j] : 3.0e38f; }
                    return v;
                };
                ze = z_coarse<RM>(a, gr, rkey, 0, near, far);
                nk = s_n[j];
                constexpr int FQ = 8;
                float zq[FQ]; int eq[FQ]; float4 rq[FQ];
                const float4* RAWp = reinterpret_cast<const float4*>(slot + H2F);          // + e * (ENTF / 4)
                int produced = 0;
#pragma unroll
                for (int u = 0; u < FQ; ++u) {
                    zq[u] = 0.f; eq[u] = 0;
                    if (produced < S) { zq[u] = next_entry(eq[u]); ++produced; }
                    rq[u] = *(&RAWp[(size_t)eq[u] * (ENTF / 4) + j]);
                }
                float dist = 0.f;
                for (int s0 = 0; s0 < S; s0 += FQ) {
#pragma unroll
                    for (int u = 0; u < FQ; ++u) {
                        const int sidx = s0 + u;
                        if (sidx < S) {
                            const float zc = zq[u];
                            const int ec = eq[u];
                            const float4 raw = rq[u];
                            // refill this slot with merged sample sidx + FQ (if any) before using its neighbour's depth
                            if (produced < S) {
                                zq[u] = next_entry(eq[u]); ++produced;
                                rq[u] = *(&RAWp[(size_t)eq[u] * (ENTF / 4) + j]);
                            }
                            if (sidx + 1 < S) dist = zq[(u + 1) % FQ] - zc;          // dists[-1] repeats dists[-2] (:36-37)
                            float sg = raw.w;
                            if (RANDOM && a.p.noise_std > 0.f) {
                                const float e = (RANDOM == 2 && a.noise_f) ? a.noise_f[gr * S + sidx] : rng_normal(a, rkey, gr, sidx, STREAM_EPS_F);
                                sg += e * a.p.noise_std;
                            }
                            sg = fmaxf(sg, 0.f);
                            const float alpha = 1.0f - expf(-sg * (dist * dn));
                            const float wgt = alpha * T;
                            T = T * ((1.0f - alpha) + 1e-10f);
                            c0 = fmaf(wgt, sigmoid_rcp(raw.x), c0);
                            c1 = fmaf(wgt, sigmoid_rcp(raw.y), c1);
                            c2 = fmaf(wgt, sigmoid_rcp(raw.z), c2);
                            dep = fmaf(wgt, zc, dep);
                            accw += wgt;
                            wmax = fmaxf(wmax, wgt);
                            if (h == 0) slot[(size_t)ec * ENTF + H2F + 128 + j] = wgt;      // one writer per address
                            if (a.dbg_zfine && h == 0 && rayok) a.dbg_zfine[gr * S_fp + sidx] = zc;
                        }
                    }
                }
                TICK(0);            // (profile build: stage A is booked under "loop top", stage B under "positional encoding")
                RAY_FENCE();
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                // ---- stage B: composited hidden units (or features), entry order (coalesced 1-KB rows) ---------------------------
                f32x16 hsumB[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hsumB[m][r] = 0.f;
                constexpr int NROW = FEATPARK ? 8 : 16;
                // one entry at a time: two / three / four in flight measured +2 / +8 / +12 % kernel time (profiles/r04_ab_lat.txt) and cost registers
                for (int e = 0; e < S; ++e) {
                    const float4* H2 = reinterpret_cast<const float4*>(slot + (size_t)e * ENTF);
                    const float wgt = *(&slot[(size_t)e * ENTF + H2F + 128 + j]);
#pragma unroll
                    for (int q = 0; q < NROW; ++q) {
                        const float4 v = nt_load4(&H2[q * 64 + lane]);
                        hsumB[q >> 2][4 * (q & 3) + 0] = fmaf(wgt, v.x, hsumB[q >> 2][4 * (q & 3) + 0]);
                        hsumB[q >> 2][4 * (q & 3) + 1] = fmaf(wgt, v.y, hsumB[q >> 2][4 * (q & 3) + 1]);
                        hsumB[q >> 2][4 * (q & 3) + 2] = fmaf(wgt, v.z, hsumB[q >> 2][4 * (q & 3) + 2]);
                        hsumB[q >> 2][4 * (q & 3) + 3] = fmaf(wgt, v.w, hsumB[q >> 2][4 * (q & 3) + 3]);
                    }
                }
                TICK(3);
                emit(1, hsumB, c0, c1, c2, dep, accw, wmax, FEATPARK ? hsumB : nullptr);       // (feature parking: hsumB[0..1] are the features)
            }
            if (pass == 0) { z = z_coarse<RM>(a, gr, rkey, 0, near, far); znext = z_coarse<RM>(a, gr, rkey, 1, near, far); }
            else if (!CACHE) {
                ze = z_coarse<RM>(a, gr, rkey, 0, near, far);
                nk = s_n[j];
                z = next_fine(); znext = next_fine();
            }
            float dist = (CACHE && pass == 1) ? 0.f : znext - z;

            for (int s = 0; s < ((CACHE && pass == 1) ? 0 : S); ++s) {
                TICK(0);
                sample_eval<(CACHE == 2 ? HAV_GQ2 : 8), PREC, true, CACHE == 2>(a, L, b, ox, oy, oz, dx, dy, dz, z,
                                                             [&](f32x16 (&acc2)[4], float hd0, float hd1, float hd2, float hd3) {
                __builtin_amdgcn_sched_barrier(0);
                // volume_render_radiance_field (utils/nerf_util.py:28-73), one ray per lane, sequential in s
                float sg = hd3;
                if (RANDOM && a.p.noise_std > 0.f) {
                    const float* nz = pass == 0 ? a.noise_c : a.noise_f;
                    const float e = (RANDOM == 2 && nz) ? nz[gr * S + s] : rng_normal(a, rkey, gr, s, pass == 0 ? STREAM_EPS_C : STREAM_EPS_F);
                    sg += e * a.p.noise_std;
                }
                sg = fmaxf(sg, 0.f);
                const float alpha = 1.0f - expf(-sg * (dist * dn));
                const float wgt = alpha * T;                                   // :60, exclusive product
                T = T * ((1.0f - alpha) + 1e-10f);
                if (COUT) {          // (CACHE == 2: this loop only ever runs the coarse pass, whose composited outputs nobody wants)
                    const f32x16 wv = {wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt, wgt};
#pragma unroll
                    for (int m = 0; m < 4; ++m) hsum[m] = __builtin_elementwise_fma(wv, acc2[m], hsum[m]);   // v_pk_fma_f32
                }
                c0 = fmaf(wgt, sigmoid_rcp(hd0), c0);               // sigmoid on rgb only (:45-46)
                c1 = fmaf(wgt, sigmoid_rcp(hd1), c1);
                c2 = fmaf(wgt, sigmoid_rcp(hd2), c2);
                dep = fmaf(wgt, z, dep);
                accw += wgt;
                wmax = fmaxf(wmax, wgt);
                if (pass == 0 && S_fp > 0 && h == 0) { if (CACHE) wrow[s * 32 + j] = wgt; else if (rayok) wpark[s] = wgt; }
                if (CACHE && pass == 0 && S_fp > 0 && !(s & 1)) park(s >> 1, acc2, hd0, hd1, hd2, hd3);
                if (pass == 1 && a.dbg_zfine && h == 0 && rayok) a.dbg_zfine[gr * S_fp + s] = z;
                // advance: dists[-1] repeats dists[-2] (:36-37)
                z = znext;
                if (s + 2 < S) {
                    znext = pass == 0 ? z_coarse<RM>(a, gr, rkey, s + 2, near, far) : next_fine();
                    dist = znext - z;
                }
                } PROF_PASS DBG_PASS(DBG_TILE(s)));
                __builtin_amdgcn_sched_barrier(0);
                TICK(7);
            }
            RAY_FENCE();

            if constexpr (CACHE != 2) {
                if (!(CACHE && pass == 1)) emit(pass, hsum, c0, c1, c2, dep, accw, wmax);
            }

            TICK(8);
            // ---- inverse-CDF resampling (utils/nerf_util.py:76-117): one ray per lane, one sequential sweep over the CDF ----
            if (pass == 0 && S_fp > 0) {
                const int nw = S_c - 2, nb = S_c - 1;
                float sum = 0.f;
                constexpr int SPF = HAV_RESAMPLE_PF;
                for (int i0 = 0; i0 < nw; i0 += SPF) {          // SPF loads in flight per round trip, summed in index order
                    float t8[SPF];
#pragma unroll
                    for (int q = 0; q < SPF; ++q) t8[q] = (i0 + q < nw) ? (CACHE ? *(&wrow[(1 + i0 + q) * 32 + j]) : __builtin_nontemporal_load(&wpark[1 + i0 + q])) : 0.f;
#pragma unroll
                    for (int q = 0; q < SPF; ++q) if (i0 + q < nw) sum += (t8[q] + 1e-5f);
                }
                // the weights come back through an 8-deep rotation of registers: a lone load in this phase queues behind the other waves' tap
                // loads in the texture path (~1 K cycles each, measured: the two passes over 62 weights were 170 K cycles per block)
                auto wload = [&](int i) -> float { return (i < nw) ? (CACHE ? *(&wrow[(1 + i) * 32 + j]) : __builtin_nontemporal_load(&wpark[1 + i])) : 0.f; };
                constexpr int WPF = HAV_RESAMPLE_PF;
                float wq[WPF];
#pragma unroll
                for (int q = 0; q < WPF; ++q) wq[q] = wload(q);
                float run = 0.f, cdf_lo = 0.f;
                int k = 0;
                auto u_of = [&](int kk) -> float {
                    if (!RANDOM || !a.p.perturb) {
                        const float st = a.u_st;
                        return (S_f == 1) ? 0.f : ((kk < S_f / 2) ? st * (float)kk : 1.0f - st * (float)(S_f - 1 - kk));
                    }
                    const float zeta = (RANDOM == 2 && a.u_rand) ? a.u_rand[gr * S_f + kk] : rng_uniform(a, rkey, kk, STREAM_ZETA);
                    return (float)kk * a.u_sN + zeta * a.u_w;
                };
                float u = u_of(0);
                // bin centres are carried along the sweep (one coarse depth per bin) so the emission loop, which runs
                // whenever ANY of the 32 rays emits, stays a handful of instructions
                float zi = z_coarse<RM>(a, gr, rkey, 0, near, far), zi1 = z_coarse<RM>(a, gr, rkey, 1, near, far);
                float zi2 = z_coarse<RM>(a, gr, rkey, 2, near, far);
                for (int i = 0; i < nw; ++i) {
                    const float wi = wq[0];
#pragma unroll
                    for (int q = 0; q + 1 < WPF; ++q) wq[q] = wq[q + 1];
                    wq[WPF - 1] = wload(i + WPF);
                    run += (wi + 1e-5f) / sum;
                    const float cdf_hi = run;
                    const float bl = 0.5f * (zi1 + zi), ba = 0.5f * (zi2 + zi1);
                    float dnm = cdf_hi - cdf_lo;
                    if (dnm < 1e-5f) dnm = 1.0f;
                    while (k < S_f && u < cdf_hi) {          // cdf[i] <= u < cdf[i+1]: below = i, above = i+1
                        const float tt = (u - cdf_lo) / dnm;
                        if (h == 0) s_n[k * 32 + j] = bl + tt * (ba - bl);
                        ++k;
                        u = (k < S_f) ? u_of(k) : 0.f;
                    }
                    cdf_lo = cdf_hi;
                    zi = zi1; zi1 = zi2;
                    if (i + 3 < S_c) zi2 = z_coarse<RM>(a, gr, rkey, i + 3, near, far);
                }
                if (k < S_f) {                               // u >= cdf[nb-1]: below = above = nb-1 -> the last bin centre
                    const float zl = 0.5f * (z_coarse<RM>(a, gr, rkey, nb, near, far) + z_coarse<RM>(a, gr, rkey, nb - 1, near, far));
                    for (; k < S_f; ++k) if (h == 0) s_n[k * 32 + j] = zl;
                }
                wave_lds_sync();
            }
            RAY_FENCE();
            TICK(9);
        }   // pass
        wave_lds_sync();
    }       // blocks
#undef RAY_FENCE
    PROF_END(lane);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Lab builds (-DHAV_LAB, hav_render_lab.h) read the timing-experiment knobs HAV_ABLATE / HAV_STAGGER from the environment once; the shipped
// library reads no environment variable at all: everything that selects behaviour per call travels in HavRenderParams.
struct EnvKnobs { int ablate, stagger; };
static const EnvKnobs& env_knobs()
{
    static const EnvKnobs k{LAB_ENV_INT("HAV_ABLATE"), LAB_ENV_INT("HAV_STAGGER")};
    return k;
}

static bool use_block_kernel(const HavRenderParams* p)
{
    if (p->flags & HAV_FLAG_PAIR_KERNEL) return false;    // A/B runs and the pair kernel's own parity tests
    return p->S_f == 0 || p->S_c <= 67;                   // coarse weights are parked in the 67-float rgb_fine row
}

static int march_grid_blocks(const HavRenderParams* p)
{
    const long long nblk = (long long)((p->R + 31) / 32) * p->B;
    int gridb = hav_num_cus();
    const long long needb = (nblk + MARCH_WAVES - 1) / MARCH_WAVES;
    if (needb < gridb) gridb = (int)((needb + 7) / 8 * 8);
    if (p->grid_blocks > 0 && p->grid_blocks < gridb) gridb = (p->grid_blocks + 7) / 8 * 8;      // the caller leaves CUs to a concurrent stream (ABI 6)
    return gridb;
}
// 0 = exact f32 MFMA, 1 = bf16 triple split, 2 = fp16 double split, 3 = fp16 double split + MX correction terms
static int mlp_prec(const HavRenderParams* p)
{
    return p->mlp_mode == HAV_MLP_F32 ? 0 : (p->mlp_mode == HAV_MLP_SPLIT_F16 ? 2 : (p->mlp_mode == HAV_MLP_SPLIT_F16_MX ? 3 : 1));
}
static long long fine_cache_slot_floats(const HavRenderParams* p)
{
    const long long S_fp = (p->S_c + 1) / 2 + p->S_f;
    return S_fp * (((mlp_prec(p) == 2) ? WS_H2_FLOATS / 2 : WS_H2_FLOATS) + 128 + 32)       // features (fp16 mode) or hidden units
           + (long long)((p->S_c + 3) & ~3) * 32;                                                // + the coarse weights of the block, [S_c][32]
}
// Would a workspace be used at all?  Measured on MI355X (docs/history/DESIGN_r1-r4.md 3.7): with stratified jitter on (the production setting)
// skipping the repeated samples is worth 10-15 % of the kernel; with deterministic depths the coarse tiles are so coherent (all 32
// rays at the same depth) that in bf16 mode the 13 GB of parking traffic per frame cancel the gain.  With feature parking (fp16
// mode) the parking stream is half as large and the cache pays with deterministic depths as well.
// Default: fp16 -> always, bf16 -> iff perturb; HAV_FLAG_FINE_CACHE / HAV_FLAG_FINE_RECOMPUTE force either.
static bool fine_cache_applies(const HavRenderParams* p)
{
    if (p->S_f <= 0 || p->S_c < 2 || !use_block_kernel(p) || mlp_prec(p) == 0) return false;
    if (p->flags & HAV_FLAG_FINE_RECOMPUTE) return false;
    // With a workspace the cache is the default in both split modes, with and without jitter.  (Round 1 had it off for deterministic
    // depths in the bf16 mode, where it was neutral at the time; on the present kernels it is worth 7 % with all seven maps and 14 %
    // with the fine maps only -- 9.0 -> 7.75 ms at 512 x 512, profiles/r04_ab_cache_det.txt.)  HAV_FLAG_FINE_CACHE is kept for callers
    // that set it; HAV_FLAG_FINE_RECOMPUTE is the way to get every merged sample evaluated.
    return true;
}
static bool use_fine_cache(const HavRenderParams* p)
{
    if (!fine_cache_applies(p) || !p->workspace) return false;
    return p->workspace_bytes >= (uint64_t)march_grid_blocks(p) * MARCH_WAVES * fine_cache_slot_floats(p) * sizeof(float);
}
extern "C" int64_t hav_render_workspace_bytes(const HavRenderParams* p)
{
    if (!p || p->R < 0 || p->B < 1 || !fine_cache_applies(p)) return 0;
    return (int64_t)march_grid_blocks(p) * MARCH_WAVES * fine_cache_slot_floats(p) * (int64_t)sizeof(float);
}

// THE decision function: which kernel instantiation renders a call.  hav_render_rays launches what this returns and
// hav_render_variant_name prints it, so a test that asserts a name asserts what ran.
//   rm   0: no random numbers, 1: device streams only (production), 2: injected tensors where given (parity tests), else device
//   prec 0: exact fp32 MFMA, 1: bf16 triple split, 2: fp16 double split
//   cm   0: every merged fine sample is evaluated, 1: fine-pass cache, 2: cache + fine maps only (coarse outputs declined)
// fp16 + cache + coarse outputs (<., 2, 1>) does not exist: that combination carries 64 more live accumulators through the
// feature projection of parked tiles and ran into the MFMA operand hazard of docs/history/DESIGN_r1-r4.md 3.5 under 240-280 spilled VGPRs; a
// caller that wants the coarse maps in fp16 mode gets every merged sample evaluated instead.
struct MarchVariant { bool blk; int rm, prec, cm; bool guard; };
static MarchVariant pick_variant(const HavRenderParams* p, bool no_coarse_out, bool injected)
{
    MarchVariant v{};
    const bool random = p->perturb != 0 || p->noise_std > 0.f;
    v.blk = use_block_kernel(p);
    if (!v.blk) { v.rm = random ? 2 : 0; return v; }                       // ray-pair kernel: exact fp32 only
    v.prec = mlp_prec(p);
    v.rm = !random ? 0 : ((injected || v.prec == 0) ? 2 : 1);              // (the f32 mode only instantiates the injected-tensor RNG variant)
    const bool cache = use_fine_cache(p);
    v.cm = !cache ? 0 : (no_coarse_out ? 2 : (v.prec == 2 ? 0 : 1));
    v.guard = v.prec >= 2 && !(p->flags & HAV_FLAG_NO_FP16_GUARD);          // both fp16-based modes convert weights and activations to fp16
    return v;
}

extern "C" int hav_render_variant_name(const HavRenderParams* p, int coarse_outputs, int injected_rand, char* buf, int len)
{
    if (!p || !buf || len < 40) return HAV_EINVAL;
    const MarchVariant v = pick_variant(p, !coarse_outputs && p->S_f > 0, injected_rand != 0);
    if (!v.blk) snprintf(buf, (size_t)len, "hav_march_f32_kernel<%s>", v.rm ? "true" : "false");
    else snprintf(buf, (size_t)len, "hav_march_blk_kernel<%d, %d, %d>", v.rm, v.prec, v.cm);
    return 0;
}

template <int R_, int P_, int C_>
static void launch_blk(int gridb, size_t ldsb, hipStream_t s, const MarchArgs& a)
{
    hipLaunchKernelGGL((hav_march_blk_kernel<R_, P_, C_>), dim3(gridb), dim3(MARCH_THREADS), ldsb, s, a);
}
struct BlkEntry { int rm, prec, cm; const void* fn; void (*launch)(int, size_t, hipStream_t, const MarchArgs&); };
#define BLK(R_, P_, C_) {R_, P_, C_, (const void*)hav_march_blk_kernel<R_, P_, C_>, launch_blk<R_, P_, C_>}
static const BlkEntry kBlk[] = {
#ifdef HAV_FAST_BUILD      // development builds: the production kernel only (seconds instead of a minute per compile)
    BLK(0, 2, 2), BLK(1, 2, 2), BLK(0, 1, 0), BLK(1, 1, 0),       // + the fp16 guard's bf16 stand-ins
    BLK(0, 1, 2), BLK(1, 1, 2),                                   // + the bf16 production pair
    BLK(0, 3, 2), BLK(1, 3, 2)};                                  // + the fp16 + MX production pair
#else
    BLK(0, 0, 0), BLK(2, 0, 0),
    BLK(0, 1, 0), BLK(1, 1, 0), BLK(2, 1, 0), BLK(0, 1, 1), BLK(1, 1, 1), BLK(2, 1, 1), BLK(0, 1, 2), BLK(1, 1, 2), BLK(2, 1, 2),
    BLK(0, 2, 0), BLK(1, 2, 0), BLK(2, 2, 0), BLK(0, 2, 2), BLK(1, 2, 2), BLK(2, 2, 2),
    BLK(0, 3, 0), BLK(1, 3, 0), BLK(2, 3, 0), BLK(0, 3, 1), BLK(1, 3, 1), BLK(2, 3, 1), BLK(0, 3, 2), BLK(1, 3, 2), BLK(2, 3, 2)};
#endif
#undef BLK
static const BlkEntry* find_blk(int rm, int prec, int cm)
{
    for (const BlkEntry& e : kBlk) if (e.rm == rm && e.prec == prec && e.cm == cm) return &e;
    return nullptr;
}

extern "C" int hav_render_rays(const HavRenderParams* p, const float* rays, const float* bg, const float* inv_T,
                               const float* planes_prepared, const float* skin_vol, const void* mlp_blob, const float* t_rand,
                               const float* u_rand, const float* noise_c, const float* noise_f, const HavRenderOut* out,
                               void* stream)
{
    if (!p || !rays || !inv_T || !planes_prepared || !skin_vol || !mlp_blob || !out) return HAV_EINVAL;
    if (p->B < 1 || p->R < 0 || p->ray_stride < 8 || p->S_c < 2 || p->S_f < 0) return HAV_EINVAL;
    if (p->plane_ch != HAV_PC) return HAV_EUNSUP;                 // Trainer hard-codes triPlane_feat_dim=64 (nerf_trainer.py:22)
    if (p->plane_res < 2 || p->vol_res < 2) return HAV_EINVAL;
    if (p->S_c > 256 || p->S_f > 128) return HAV_EUNSUP;
    if ((p->flags & ~HAV_FLAGS_ALL) || ((p->flags & HAV_FLAG_FINE_CACHE) && (p->flags & HAV_FLAG_FINE_RECOMPUTE))) return HAV_EINVAL;
    if (p->grid_blocks < 0 || p->reserved0 != 0) return HAV_EINVAL;
    if (p->mlp_mode != HAV_MLP_SPLIT_BF16 && p->mlp_mode != HAV_MLP_F32 && p->mlp_mode != HAV_MLP_SPLIT_F16 && p->mlp_mode != HAV_MLP_SPLIT_F16_MX)
        return HAV_EINVAL;
    // the coarse pass's composited outputs may be declined (all three NULL) when there is a fine pass: Trainer.forward then only
    // uses the fine ones, and the kernel drops the accumulators that exist for them
    const bool no_coarse_out = !out->rgb_coarse && !out->depth_coarse && !out->acc_coarse;
    if (no_coarse_out ? (p->S_f <= 0) : (!out->rgb_coarse || !out->depth_coarse || !out->acc_coarse)) return HAV_EINVAL;
    if (!out->weights_max) return HAV_EINVAL;
    if (no_coarse_out && !use_block_kernel(p)) return HAV_EUNSUP;          // the ray-pair kernel always writes them
    if (p->S_f > 0 && (!out->rgb_fine || !out->depth_fine || !out->acc_fine)) return HAV_EINVAL;
    if (p->R == 0) return 0;
    hipStream_t st = (hipStream_t)stream;

    MarchArgs a;
    a.p = *p;
    a.rays = rays; a.bg = bg; a.inv_T = inv_T; a.planes = nullptr; a.pplanes = planes_prepared; a.vol = skin_vol;
    a.blob = (const float*)mlp_blob;
    a.t_rand = t_rand; a.u_rand = u_rand; a.noise_c = noise_c; a.noise_f = noise_f;
    a.out = *out;
    a.dbg_zfine = p->dbg_zfine;
    a.ablate = env_knobs().ablate;
    a.stagger = env_knobs().stagger;
    a.guard = nullptr; a.guard_run_if = 0; a.status = p->status;
    a.ws = nullptr; a.ws_slot = 0; a.rng_base = 0;
    a.step_c = 1.0f / (float)(p->S_c - 1);
    a.u_st = p->S_f > 1 ? 1.0f / (float)(p->S_f - 1) : 0.f;
    a.u_sN = p->S_f > 0 ? (float)(1.0 / (double)p->S_f) : 0.f;
    a.u_w = p->S_f > 0 ? (float)(1.0 / (double)p->S_f - 1e-6) : 0.f;
    a.plim = (float)(p->plane_res - 1);
    a.vlim = (float)(p->vol_res - 1);
    a.NR = (long long)p->B * p->R;
    a.S_fp = p->S_f > 0 ? (p->S_c + 1) / 2 + p->S_f : 0;
    a.s_pad_c = (p->S_c + 15) & ~15;
    a.s_pad_f = ((a.S_fp > 0 ? a.S_fp : 1) + 15) & ~15;
    a.o_racc = 0;                                   // first: keeps the float4 accumulators 16-byte aligned
    a.o_w = a.o_racc + 2 * RACC_N;
    a.o_cdf = a.o_w + 2 * a.s_pad_c;
    a.o_cand = a.o_cdf + 2 * a.s_pad_c;
    a.o_zf = a.o_cand + 2 * a.s_pad_f;
    a.scr_floats = (a.o_zf + 2 * a.s_pad_f + 3) & ~3;

    const bool random = p->perturb != 0 || p->noise_std > 0.f;
    static std::atomic<unsigned long long> attr_mask{0};
    int dev;
    if (!attr_done(attr_mask, dev)) {               // per device: a second GPU driven from the same process needs its own attributes
        for (const BlkEntry& e : kBlk) {
            hipError_t er = hipFuncSetAttribute(e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (er != hipSuccess) return (int)er;
        }
#ifndef HAV_FAST_BUILD
        const void* pk[2] = {(const void*)hav_march_f32_kernel<false>, (const void*)hav_march_f32_kernel<true>};
        for (int i = 0; i < 2; ++i) {
            hipError_t er = hipFuncSetAttribute(pk[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (er != hipSuccess) return (int)er;
        }
#endif
        attr_mask.fetch_or(1ull << dev, std::memory_order_release);
    }
    const bool injected = t_rand || u_rand || noise_c || noise_f;     // parity tests; production draws everything on the device
    const MarchVariant v = pick_variant(p, no_coarse_out, injected);
    if (v.blk) {
        a.scr_floats = ((p->S_f > 0 ? p->S_f : 1) * 32 + 3) & ~3;
        auto lds_of = [&](int prec) {
            return ((size_t)(prec == 3 ? LDSX_FLOATS : (prec == 2 ? LDSH_FLOATS : (prec == 1 ? LDS3_FLOATS : LDS_FLOATS))) + HAV_LDS_BIAS + (size_t)MARCH_WAVES * a.scr_floats) * sizeof(float);
        };
        if (lds_of(v.prec) > 160 * 1024 || (v.guard && lds_of(1) > 160 * 1024)) return HAV_EUNSUP;
        const int gridb = march_grid_blocks(p);
        const BlkEntry* main_k = find_blk(v.rm, v.prec, v.cm);
        if (!main_k) return HAV_EUNSUP;
        a.ws = v.cm ? (float*)p->workspace : nullptr;
        a.ws_slot = fine_cache_slot_floats(p);
        if (v.guard) {
            // fp16 range guard (include/havatar.h): the verdict hav_triplane_prepare left in the planes' trailer decides ON THE DEVICE
            // whether the fp16 kernel or its bf16 stand-in (every merged sample evaluated, no workspace) does the work; the other
            // one returns at once.  No host round trip, so a captured hipGraph keeps working when weights or planes change.
            a.guard = reinterpret_cast<const unsigned int*>(planes_prepared + (size_t)2 * p->B * p->plane_res * p->plane_res * 128) + 256;
            a.guard_run_if = 0;
        }
        main_k->launch(gridb, lds_of(v.prec), st, a);
        HAV_LAUNCH_CHECK();
        if (v.guard) {
            const BlkEntry* fb = find_blk(v.rm, 1, 0);
            if (!fb) return HAV_EUNSUP;
            a.guard_run_if = 1;
            a.ws = nullptr;
            fb->launch(gridb, lds_of(1), st, a);
            HAV_LAUNCH_CHECK();
        }
    } else {
        const size_t lds = ((size_t)LDS_FLOATS + (size_t)MARCH_WAVES * a.scr_floats) * sizeof(float);
        if (lds > 160 * 1024) return HAV_EUNSUP;
        const long long npairs = (a.NR + 1) / 2;
        int grid = hav_num_cus();
        const long long need = (npairs + MARCH_WAVES - 1) / MARCH_WAVES;
        if (need < grid) grid = (int)((need + 7) / 8 * 8);
#ifndef HAV_FAST_BUILD
        if (random) hipLaunchKernelGGL(hav_march_f32_kernel<true>, dim3(grid), dim3(MARCH_THREADS), lds, st, a);
        else hipLaunchKernelGGL(hav_march_f32_kernel<false>, dim3(grid), dim3(MARCH_THREADS), lds, st, a);
#else
        (void)grid; (void)lds;
        return HAV_EUNSUP;          // development build (tools/build_variant.sh): the ray-pair kernel is not compiled in
#endif
        HAV_LAUNCH_CHECK();
    }
    if (p->rng_counter && random) { hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, st, (unsigned long long*)p->rng_counter); HAV_LAUNCH_CHECK(); }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Test hook (include/havatar.h: hav_debug_mlp_layer): one dense layer of the radiance MLP, WITHOUT its activation, evaluated by the very
// matrix routine the march kernel uses in `mlp_mode` -- so that the arithmetic modes can be held against an fp64 product directly
// (tests/test_render_gpu.py, tools/err_modes.py) instead of through 80 compositing steps that hide the last three bits.
// ------------------------------------------------------------------------------------------------
template <int PREC, int LAYER>
__global__ void __launch_bounds__(64) debug_mlp_layer_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ blob, long long n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    {
        const int base = PREC == 0 ? 0 : (PREC == 1 ? OFF_A1S : OFF_A1H);
        const int cnt = PREC == 0 ? OFF_W4 : (PREC == 1 ? LDS3_FLOATS : LDSX_FP16);
        for (int i = lane; i < cnt / 4; i += 64) reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(blob + base)[i];
        if (PREC == 3)
            for (int i = lane; i < MX_GROUPS * MX_G_DWORDS / 4; i += 64) reinterpret_cast<float4*>(smem + LDSX_FP16)[i] = reinterpret_cast<const float4*>(blob + OFF_MX)[i];
    }
    __syncthreads();
    const uint4* sA1 = reinterpret_cast<const uint4*>(smem);
    const uint4* sA2 = reinterpret_cast<const uint4*>(smem + (PREC >= 2 ? (OFF_A2H - OFF_A1H) : (OFF_A2S - OFF_A1S)));
    const unsigned int* sMX = reinterpret_cast<const unsigned int*>(smem + LDSX_FP16);
    constexpr int KIN = LAYER == 1 ? 48 : 128;
    for (long long tile = blockIdx.x; tile * 32 < n; tile += gridDim.x) {
        const long long q = tile * 32 + j;
        const long long qc = q < n ? q : n - 1;
        f32x16 acc[4], in[4];
        float pe[KPE_STEPS];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][r] = blob[(LAYER == 1 ? OFF_B1 : OFF_B2) + acc_row(m, r, h)];
                in[m][r] = LAYER == 2 ? x[qc * KIN + acc_row(m, r, h)] : 0.f;
            }
#pragma unroll
        for (int k = 0; k < KPE_STEPS; ++k) pe[k] = LAYER == 1 ? x[qc * KIN + 24 * h + k] : 0.f;
        auto get1 = [&](int c, float (&v)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = pe[8 * c + e];
        };
        auto get2 = [&](int ch, float (&v)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = in[ch >> 1][8 * (ch & 1) + e];
        };
        if (LAYER == 1) {
            if (PREC == 3) mfma_split2x<3, false>(acc, sA1, sMX, lane, get1);
            else if (PREC == 2) mfma_split2h<3, 4>(acc, sA1, lane, get1);
            else if (PREC == 1) mfma_split3<3>(acc, sA1, lane, get1);
            else {
#pragma unroll
                for (int t = 0; t < KPE_STEPS; ++t)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(smem[OFF_W1PE + (t * 4 + m) * 64 + lane], pe[t], acc[m], 0, 0, 0);
            }
        } else {
            if (PREC == 3) mfma_split2x<8, true>(acc, sA2, sMX + 4 * MX_G_DWORDS, lane, get2);
            else if (PREC == 2) mfma_split2h<8, 4>(acc, sA2, lane, get2);
            else if (PREC == 1) mfma_split3<8>(acc, sA2, lane, get2);
            else {
#pragma unroll
                for (int ks = 0; ks < K2_STEPS; ++ks)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(smem[OFF_W2 + (ks * 4 + m) * 64 + lane], in[ks >> 4][ks & 15], acc[m], 0, 0, 0);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        if (q < n) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[q * 128 + acc_row(m, r, h)] = acc[m][r];
        }
    }
}

extern "C" int hav_debug_mlp_layer(float* y, const float* x, const void* mlp_blob, int mlp_mode, int layer, int64_t n, void* stream)
{
    if (!y || !x || !mlp_blob || n < 0 || (layer != 1 && layer != 2)) return HAV_EINVAL;
    if (n == 0) return 0;
    const int prec = mlp_mode == HAV_MLP_F32 ? 0 : (mlp_mode == HAV_MLP_SPLIT_BF16 ? 1 : (mlp_mode == HAV_MLP_SPLIT_F16 ? 2 : (mlp_mode == HAV_MLP_SPLIT_F16_MX ? 3 : -1)));
    if (prec < 0) return HAV_EINVAL;
    const size_t lds = (size_t)(prec == 0 ? OFF_W4 : (prec == 1 ? LDS3_FLOATS : (prec == 2 ? LDSX_FP16 : LDSX_FLOATS))) * sizeof(float);
    const void* fn[4][2] = {{(const void*)debug_mlp_layer_kernel<0, 1>, (const void*)debug_mlp_layer_kernel<0, 2>},
                            {(const void*)debug_mlp_layer_kernel<1, 1>, (const void*)debug_mlp_layer_kernel<1, 2>},
                            {(const void*)debug_mlp_layer_kernel<2, 1>, (const void*)debug_mlp_layer_kernel<2, 2>},
                            {(const void*)debug_mlp_layer_kernel<3, 1>, (const void*)debug_mlp_layer_kernel<3, 2>}};
    hipError_t er = hipFuncSetAttribute(fn[prec][layer - 1], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (er != hipSuccess) return (int)er;
    const long long tiles = (n + 31) / 32;
    const int grid = (int)(tiles < hav_num_cus() ? tiles : hav_num_cus());
    float* yy = y; const float* xx = x; const float* bb = (const float*)mlp_blob; long long nn = n;
    void* args[] = {&yy, &xx, &bb, &nn};
    er = hipLaunchKernel(fn[prec][layer - 1], dim3(grid), dim3(64), args, lds, (hipStream_t)stream);
    if (er != hipSuccess) return (int)er;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// get_rays on device (dataloader/data_util.py:28-56 + dataloader/dataloader.py:174-177)
// ------------------------------------------------------------------------------------------------
struct GenRaysArgs { float intr[4]; float c2w[12]; float near, far; int H, W, y0, y1; };

__global__ void __launch_bounds__(256) gen_rays_kernel(float* __restrict__ rays, GenRaysArgs g)
{
    const int n = (g.y1 - g.y0) * g.W;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int i = idx % g.W, jj = g.y0 + idx / g.W;
    const float fx = g.intr[0], fy = g.intr[1], cx = g.intr[2] * (float)g.W, cy = g.intr[3] * (float)g.H;
    const float dc0 = ((float)i - cx) / fx, dc1 = ((float)jj - cy) / fy, dc2 = 1.0f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = g.c2w[r * 4 + 0] * dc0 + g.c2w[r * 4 + 1] * dc1 + g.c2w[r * 4 + 2] * dc2;
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float4* o = reinterpret_cast<float4*>(rays + (size_t)idx * 8);
    o[0] = make_float4(g.c2w[3], g.c2w[7], g.c2w[11], d[0] / nrm);
    o[1] = make_float4(d[1] / nrm, d[2] / nrm, g.near, g.far);
}

extern "C" int hav_gen_rays(float* rays, int H, int W, const float intr[4], const float c2w[12], float near, float far,
                            int y0, int y1, void* stream)
{
    if (!rays || !intr || !c2w || H < 1 || W < 1 || y0 < 0 || y1 > H || y1 <= y0) return HAV_EINVAL;
    GenRaysArgs g;
    for (int i = 0; i < 4; ++i) g.intr[i] = intr[i];
    for (int i = 0; i < 12; ++i) g.c2w[i] = c2w[i];
    g.near = near; g.far = far; g.H = H; g.W = W; g.y0 = y0; g.y1 = y1;
    const int n = (y1 - y0) * W;
    hipLaunchKernelGGL(gen_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays, g);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_abi_version(void) { return HAV_ABI_VERSION; }
