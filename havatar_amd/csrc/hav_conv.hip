// hav_conv.hip -- the convolutions of the StyleGAN blocks on the fp16 matrix cores with split operands (fp32-class results):
//   conv3x3_split_kernel / conv3x3_il_kernel   3x3 stride 1 as an implicit GEMM, the block's modulation / demodulation / noise / bias /
//                                              leaky-ReLU fused in (plain 64 x 128 tiles; interleaved 128 x 128 tiles for Cout % 128 == 0)
//   gemm_split_kernel + upconv_finish_kernel   the up-sampling StyledConv: transposed convolution as a matrix product, then stride-2
//                                              scatter + 4x4 FIR + epilogue in one pass
//   conv3x3_wgrad_kernel (+ reduce)            weight gradient of the 3x3 convolution (training)
//   absmax_kernel                              range control for gradient-sized inputs
// The first section describes the 3x3 forward kernel; the others carry their own headers further down.
//
// 3x3 stride-1 convolution of the StyleGAN blocks as an implicit GEMM on the fp16 matrix cores with split
// operands (fp32-class results), with the modulation, demodulation, noise, bias and leaky-ReLU of the block fused in.
// SURVEY 8(f) next-4; reference: ModulatedConv2d / StyledConv / ConvLayer of model/styleUnet.py:165-297,326-368,565-599 in their
// scale-input / shared-weight / scale-output form (the reference's own non-fused algebra, :200-227):
//
//   y[b,o,p] = act( d[b,o] * sum_{i,ky,kx} W[o,i,ky,kx] * (s[b,i] * x[b,i,p + (ky-1, kx-1)])  + nw * noise[p] + bias[o] ) * gain
//
// Every term but the convolution is optional (plain ConvLayer: bias + leaky-ReLU only).  Zero padding, NCHW fp32 in and out.
//
// Why not MIOpen: its fastest fp32 solver for these shapes (Winograd F(2,3)) runs at ~86 TFLOP/s effective; the 16-bit matrix pipe of
// gfx950 is 16x faster per k than the fp32 one, and three fp16 products with fp32 accumulation (x = xh + xl, w = wh + wl:
// wl.xh + wh.xl + wh.xh) reproduce the fp32 product to ~2^-22, the size of the fp32 accumulation error of a 4608-term dot product.
//
// GEMM view: M = Cout, N = pixels, K = 9 Cin.  Workgroup = 4 waves = a 64 (Cout) x [4 rows x 32 columns] output tile; wave (wm, wn)
// owns one 32-row M tile and two 32-pixel rows.  K runs over 16-channel chunks: the chunk's input patch (6 x 34 pixels with halo,
// modulated by s, split into fp16 hi / lo) is staged in LDS once and serves all 9 taps -- a tap is just a pixel offset into the patch
// -- so the B operand of tap (ky, kx) is one conflict-free ds_read_b128 per part (pixel records are 80 bytes apart: 16 lanes hit 16
// distinct bank groups).  The A operands (weights) are pre-split, pre-scaled by 2^8 (keeps the low parts out of the fp16 subnormals;
// undone exactly in the epilogue) and pre-arranged per (chunk, tap, M tile, part) by hav_conv3x3_pack, and stream from L2 with the
// 18 fragments of a chunk in flight at once.  Global loads of chunk c+1 are issued before the MFMAs of chunk c, converted and written
// to the other LDS buffer after them: one barrier per chunk.
#include "hav_common.h"
#include <atomic>
#include <type_traits>

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float fl2_t __attribute__((ext_vector_type(2)));

#define CV_ROWS 4
#define CV_COLS 32
#define CV_PR (CV_ROWS + 2)
#define CV_PC (CV_COLS + 2)
#define CV_PIX (CV_PR * CV_PC)           // 204 pixels of the staged patch
#define CV_REC 20                        // dwords per pixel record: 8 (hi, 16 ch) + 8 (lo) + 4 pad = 80 bytes
#define CV_TASKS (CV_PIX * 8)            // (pixel, channel pair) staging tasks per chunk
#define CV_TPT ((CV_TASKS + 255) / 256)  // per thread
#define CV_WSHIFT 256.0f
// Rounds 1-3 put 32 wait states behind the matrix instructions of a chunk, before the staging code of the next one, against an operand
// hazard that does not exist (docs/history/DESIGN_r1-r4.md 3.5; tools/ubench/mfma_war.hip).  Round 4 re-validated the kernels without them (every parity test
// of tests/test_ops_gpu.py incl. test_conv3x3_full_occupancy_runs_are_bitwise_identical: profiles/r04_conv_nopad.txt) and dropped them; the
// asm statement stays as a scheduling fence (it ties the accumulators), empty.  -DCV_MFMA_PAD='"s_nop 15\n\ts_nop 15"' brings them back.
#ifndef CV_MFMA_PAD
#define CV_MFMA_PAD ""
#endif

extern "C" int64_t hav_conv3x3_packed_bytes(int Cout, int Cin) { return (int64_t)(Cin / 16) * 9 * (Cout / 32) * 2 * 64 * 16; }

// fragment (chunk cc, tap t, M tile m, part): lane (i, h) holds W[32m + i][16cc + 8h + e][t] * wmul * 2^8, e = 0..7, as fp16 hi or lo
// transposed != 0: the filters of the DATA GRADIENT, W'[o' = i][i' = o][t] = W[o][i][8 - t] (Cout, Cin are those of W'), read straight
// from W [Cin, Cout, 3, 3] -- no flip / transpose / contiguous passes in front of the pack
// One workgroup per (16-channel chunk cc, 32-row tile m): the tile's 32 x 16 x 9 weights are read as whole runs (144 consecutive floats per
// filter row; transposed: 288 per input channel) into LDS and leave as the 18 fragments (9 taps x hi / lo) of the tile, 16 bytes per lane.
// (Rounds 1-4 gathered every fragment element straight from memory: 8 loads per thread that each touched 64 cache lines -- 12.9 us per
// 512 x 512 layer, 36 packs per training step.)
__global__ void __launch_bounds__(256) conv3x3_pack_kernel(uint4* __restrict__ blob, const float* __restrict__ w, int Cout, int Cin, float wmul,
                                                           int transposed)
{
    __shared__ float tile[16 * 289 > 32 * 145 ? 16 * 289 : 32 * 145];
    const int cc = blockIdx.x, m = blockIdx.y, MT = Cout / 32, tid = threadIdx.x;
    if (!transposed) {
        // rows = filters 32 m .. + 31, 144 floats each: W[o][16 cc .. + 15][0..8]
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const int e = tid + 256 * j, row = e / 144, col = e - 144 * row;
            tile[row * 145 + col] = w[((int64_t)(32 * m + row) * Cin + 16 * cc) * 9 + col];
        }
    } else {
        // rows = input channels 16 cc .. + 15 of W' = rows of W, 288 floats each: W[ci][32 m .. + 31][0..8]
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            const int e = tid + 256 * j, row = e / 288, col = e - 288 * row;
            tile[row * 289 + col] = w[((int64_t)(16 * cc + row) * Cout + 32 * m) * 9 + col];
        }
    }
    __syncthreads();
    for (int q = tid; q < 9 * 2 * 64; q += 256) {
        const int lane = q & 63, part = (q >> 6) & 1, t = q >> 7, i = lane & 31, h = lane >> 5;
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cl = 8 * h + 2 * d + u;
                v[u] = (transposed ? tile[cl * 289 + i * 9 + (8 - t)] : tile[i * 145 + cl * 9 + t]) * (wmul * CV_WSHIFT);
            }
            const fl2_t f = {v[0], v[1]};
            const h2_t hi = __builtin_convertvector(f, h2_t);
            const h2_t lo = __builtin_convertvector(f - __builtin_convertvector(hi, fl2_t), h2_t);
            o[d] = __builtin_bit_cast(uint32_t, part ? lo : hi);
        }
        blob[((((int64_t)cc * 9 + t) * MT + m) * 2 + part) * 64 + lane] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int hav_conv3x3_pack(void* blob, const float* w, int Cout, int Cin, float wmul, void* stream)
{
    if (!blob || !w || Cout < 32 || Cin < 16) return HAV_EINVAL;
    if ((Cout % 32) || (Cin % 16)) return HAV_EUNSUP;
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((unsigned)(Cin / 16), (unsigned)(Cout / 32)), dim3(256), 0, (hipStream_t)stream, (uint4*)blob, w, Cout, Cin,
                       wmul, 0);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_conv3x3_pack_t(void* blob, const float* w, int Cout_w, int Cin_w, float wmul, void* stream)
{
    // w [Cout_w, Cin_w, 3, 3] -> the blob of the convolution with Cin_w output and Cout_w input channels (flipped taps)
    if (!blob || !w || Cout_w < 16 || Cin_w < 32) return HAV_EINVAL;
    if ((Cout_w % 16) || (Cin_w % 32)) return HAV_EUNSUP;
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((unsigned)(Cout_w / 16), (unsigned)(Cin_w / 32)), dim3(256), 0, (hipStream_t)stream, (uint4*)blob, w, Cin_w, Cout_w,
                       wmul, 1);
    HAV_LAUNCH_CHECK();
    return 0;
}

// Range control of the fp16 split (all kernels below): e with 2^e * (max |x| * extra) in [512, 1024).  `words` = hav_absmax's 256
// partial maxima (bit patterns of non-negative floats: unsigned order = float order), folded here by every wave itself; `extra` = a
// bound on what multiplies x before the split (max |s| of the modulation; 1 otherwise).  The exponent is clamped so that 2^e and
// 2^-e stay normal floats; a zero, subnormal, infinite or NaN maximum gives 0 (scale 1).
__device__ __forceinline__ int amax_pow2(const unsigned int* words, int lane, float extra)
{
    static_assert(HAV_ABSMAX_WORDS == 256, "four partial maxima per lane");
    const uint4 w4 = reinterpret_cast<const uint4*>(words)[lane];
    unsigned int mb = w4.x > w4.y ? w4.x : w4.y;
    mb = w4.z > mb ? w4.z : mb;
    mb = w4.w > mb ? w4.w : mb;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned int t = (unsigned int)__shfl_xor((int)mb, o, 64); mb = t > mb ? t : mb; }
    mb = __float_as_uint(__uint_as_float(mb) * extra);
    const int be = (int)((mb >> 23) & 0xFFu);          // biased exponent of the bound
    if (be < 1 || be > 254) return 0;
    const int e = 9 - (be - 127);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned int)(127 + e) << 23); }
// max |s[0..n)| over the wave (the modulation of one sample)
__device__ __forceinline__ float wave_absmax(const float* s, int n, int lane)
{
    float m = 0.f;
    for (int c = lane; c < n; c += 64) m = fmaxf(m, fabsf(s[c]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    return m;
}

struct ConvArgs {
    float* y; const float* x; const uint4* blob;
    float* partial;          // K-split: [ksplit][B,Cout,H,W] raw accumulators (already scaled back), reduced by conv3x3_finish_kernel
    int ksplit;
    const unsigned int* in_amax;   // optional: bits of max |x| (hav_absmax); the input is then scaled by the power of two that brings it to
                                   // [512, 1024) before the fp16 split and the result scaled back -- exact, and it keeps tiny inputs
                                   // (gradients: 1e-6 and below) out of the fp16 subnormals, where the hi + lo split loses its low part
    const float* s; const float* d; const float* noise; const float* noise_weight; const float* bias;
    float slope, gain;
    int act, noise_batched;
    int B, Cin, Cout, H, W;
};

#ifdef CV_PROFILE
// tools/conv_phase.sh: per-wave cycle sums of the main loop's phases (alternative build only)
__device__ unsigned long long* g_cv_prof = nullptr;
extern "C" int hav_conv_profile_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cv_prof), &p, sizeof(p)); }
#define CV_T(v) do { v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define CV_T(v) do { } while (0)
#endif

// NARROW: maps 16 columns wide (the 16^2 layers of the generators; any W % 16 == 0 that is not a multiple of 32, H % 8 == 0): the tile is
// 8 rows x 16 columns instead of 4 x 32 -- lane j of a 32-pixel MFMA column block is pixel (row j >> 4, column j & 15) of a row pair.
template <bool HAS_S, bool NARROW = false>      // modulated (s given) or plain: compile-time, so that neither variant carries the other's loads / multiplies
__global__ void __launch_bounds__(256, 2) conv3x3_split_kernel(ConvArgs a)
{
    constexpr int ROWS = NARROW ? 8 : CV_ROWS, COLS = NARROW ? 16 : CV_COLS, PC = COLS + 2, PIX = (ROWS + 2) * PC, TASKS = PIX * 8;
    static_assert(PIX <= CV_PIX, "the narrow patch fits the wide one's LDS");
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][CV_PIX * CV_REC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int bw = a.W / COLS;
    const int px = blockIdx.x % bw, py = blockIdx.x / bw;
    const int x0 = px * COLS, y0 = py * ROWS;
    const int mt = blockIdx.y * 2 + wm;          // this wave's 32-row tile of output channels
    const int b = blockIdx.z / a.ksplit, ks = blockIdx.z - b * a.ksplit;
    const int H = a.H, W = a.W, Cin = a.Cin, NCT = Cin / 16, MT = a.Cout / 32;
    // K-split (small maps: too few output tiles to fill the GPU): this workgroup covers the channel chunks [c_lo, c_hi)
    const int c_lo = (NCT * ks) / a.ksplit, c_hi = (NCT * (ks + 1)) / a.ksplit, NC = c_hi - c_lo;
    const float* xb = a.x + (int64_t)b * Cin * H * W;
    const float* sb = HAS_S ? a.s + (int64_t)b * Cin : nullptr;
    float in_sc = 1.0f, out_sc = 1.0f / CV_WSHIFT;
    if (a.in_amax) {
        // power of two that brings max |s x| into [512, 1024) (with a modulation: bounded by max |x| * max |s_b|, all Cin of this sample
        // whatever the K-split chunk); exact, undone in the epilogue
        const int e = amax_pow2(a.in_amax, lane, HAS_S ? wave_absmax(sb, Cin, lane) : 1.0f);
        in_sc = pow2f(e);
        out_sc = pow2f(-e - 8);          // 2^-e / CV_WSHIFT
    }

    // staging tasks of this thread: (pixel of the patch, channel pair) -> one hi dword + one lo dword
    int t_off[CV_TPT], t_lds[CV_TPT], t_cp[CV_TPT];
    bool t_ok[CV_TPT];
#pragma unroll
    for (int q = 0; q < CV_TPT; ++q) {
        const int task = tid + 256 * q;
        const int cp = task / PIX, p = task - cp * PIX;
        const int pr = p / PC, pc = p - pr * PC;
        const int gy = y0 + pr - 1, gx = x0 + pc - 1;
        t_ok[q] = task < TASKS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        t_off[q] = (2 * cp) * H * W + gy * W + gx;
        t_lds[q] = task < TASKS ? p * CV_REC + cp : -1;
        t_cp[q] = task < TASKS ? 2 * cp : 0;          // idle slots of the last round must not index past s[Cin]
    }
    // fetch() only ISSUES the global loads of a chunk (raw values; the modulation factors ride along); every use of them -- scaling,
    // splitting, the LDS writes -- happens in stash(), after the chunk's MFMAs.  A multiply inside fetch() would put the load latency
    // (s_waitcnt vmcnt(0)) in front of the matrix work of every chunk: measured 90 -> 183 us for 512 -> 512 @ 64^2.
    float sv[CV_TPT][2], ssc[CV_TPT][2];
    auto fetch = [&](int cc, float (&v)[CV_TPT][2], float (&sc)[CV_TPT][2]) {
        const float* src = xb + (int64_t)(16 * cc) * H * W;
#pragma unroll
        for (int q = 0; q < CV_TPT; ++q) {
            v[q][0] = t_ok[q] ? src[t_off[q]] : 0.f;
            v[q][1] = t_ok[q] ? src[t_off[q] + H * W] : 0.f;
            if (HAS_S) { sc[q][0] = sb[16 * cc + t_cp[q]]; sc[q][1] = sb[16 * cc + t_cp[q] + 1]; }
        }
    };
    auto stash = [&](int buf, const float (&v)[CV_TPT][2], const float (&sc)[CV_TPT][2]) {
#pragma unroll
        for (int q = 0; q < CV_TPT; ++q) {
            if (t_lds[q] < 0) continue;
            const fl2_t f = {HAS_S ? v[q][0] * (sc[q][0] * in_sc) : v[q][0] * in_sc, HAS_S ? v[q][1] * (sc[q][1] * in_sc) : v[q][1] * in_sc};
            const h2_t hi = __builtin_convertvector(f, h2_t);
            const h2_t lo = __builtin_convertvector(f - __builtin_convertvector(hi, fl2_t), h2_t);
            lds[buf][t_lds[q]] = __builtin_bit_cast(uint32_t, hi);
            lds[buf][t_lds[q] + 8] = __builtin_bit_cast(uint32_t, lo);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }

    fetch(c_lo, sv, ssc);
    stash(0, sv, ssc);
    __syncthreads();
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0, pt4 = 0, ps[4] = {0, 0, 0, 0};
    for (int ci = 0; ci < NC; ++ci) {
        const int cc = c_lo + ci, buf = ci & 1;
        CV_T(pt0);
        // weights of this chunk: 9 taps x (hi, lo), all in flight
        const uint4* ab = a.blob + ((int64_t)(cc * 9) * MT + mt) * 128 + lane;
        uint4 A[9][2];
#pragma unroll
        for (int t = 0; t < 9; ++t) { A[t][0] = ab[(int64_t)t * MT * 128]; A[t][1] = ab[(int64_t)t * MT * 128 + 64]; }
        if (ci + 1 < NC) fetch(cc + 1, sv, ssc);
        CV_T(pt1);
        const uint32_t* L = lds[buf];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            const f16x8_t ah = __builtin_bit_cast(f16x8_t, A[t][0]), al = __builtin_bit_cast(f16x8_t, A[t][1]);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int p = NARROW ? (4 * wn + 2 * rr + (j >> 4) + ky) * PC + (j & 15) + kx : (2 * wn + rr + ky) * PC + j + kx;
                const uint4 bh = *reinterpret_cast<const uint4*>(L + p * CV_REC + 4 * h);
                const uint4 bl = *reinterpret_cast<const uint4*>(L + p * CV_REC + 8 + 4 * h);
                const f16x8_t xh = __builtin_bit_cast(f16x8_t, bh), xl = __builtin_bit_cast(f16x8_t, bl);
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc[rr], 0, 0, 0);
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc[rr], 0, 0, 0);
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc[rr], 0, 0, 0);
            }
        }
        // the matrix instructions keep reading their operand registers for a while after issue (docs/history/DESIGN_r1-r4.md 3.5): wait them out before
        // the conversion code below may recycle registers
        CV_T(pt2);
        asm volatile(CV_MFMA_PAD : "+v"(acc[0]), "+v"(acc[1]));
        if (ci + 1 < NC) stash(buf ^ 1, sv, ssc);
        CV_T(pt3);
        __syncthreads();
        CV_T(pt4);
#ifdef CV_PROFILE
        ps[0] += pt1 - pt0; ps[1] += pt2 - pt1; ps[2] += pt3 - pt2; ps[3] += pt4 - pt3;
#endif
    }
#ifdef CV_PROFILE
    if (g_cv_prof && lane == 0) {
        const int w = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
        for (int q = 0; q < 4; ++q) g_cv_prof[(size_t)w * 4 + q] = ps[q];
    }
#endif
    if (a.partial) {            // K-split: raw sums out, the epilogue runs in conv3x3_finish_kernel after the slices are added up
        float* pp = a.partial + (int64_t)ks * a.B * a.Cout * H * W;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int gy = NARROW ? y0 + 4 * wn + 2 * rr + (j >> 4) : y0 + 2 * wn + rr, gx = NARROW ? x0 + (j & 15) : x0 + j;
                pp[(((int64_t)b * a.Cout + co) * H + gy) * W + gx] = acc[rr][r] * out_sc;
            }
        return;
    }

    // epilogue: demodulate, inject noise, bias, leaky-ReLU, gain -- in the order of the unfused statement (hav_styled_epilogue)
    const float nw = (a.noise && a.noise_weight) ? *a.noise_weight : 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int gy = NARROW ? y0 + 4 * wn + 2 * rr + (j >> 4) : y0 + 2 * wn + rr, gx = NARROW ? x0 + (j & 15) : x0 + j;
        const float nz = a.noise ? a.noise[(a.noise_batched ? (int64_t)b * H * W : 0) + (int64_t)gy * W + gx] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[rr][r] * out_sc;
            if (a.d) v = v * a.d[(int64_t)b * a.Cout + co];
            if (a.noise) v = v + nw * nz;
            if (a.bias) v = v + a.bias[co];
            if (a.act) v = (v > 0.f ? v : v * a.slope) * a.gain;
            a.y[(((int64_t)b * a.Cout + co) * H + gy) * W + gx] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Interleaved variant (the default where it applies: Cout % 128 == 0).  What the measurements above say: an in-order wave cannot run
// its MFMAs beside its own memory / conversion work unless that work sits BETWEEN the MFMAs in program order, and a second wave on
// the SIMD does not help (a wave with an MFMA waiting for the pipe keeps the other wave's vector instructions from issuing: a
// 512-thread ping-pong variant -- one half of the workgroup loading / converting chunk c+1 while the other half runs the MFMAs of
// chunk c -- measured exactly the sum again, 70.8 vs 73.4 us, and was dropped).  A 32x32x16 MFMA occupies the matrix pipe for 32 cycles but the issue port
// only for ~4: the ~7 slots behind it are free for the same wave.  So this kernel
//   * takes 128 (M) x 128 (N) tiles -- each wave owns one 32-row M tile and all four pixel rows: the 18 weight fragments of a chunk
//     feed 108 MFMAs instead of 54, which brings the vector-memory path from 80 % to 40 % of its throughput at MFMA speed;
//   * cuts the chunk into 36 segments of three MFMAs (one tap, one row) and hangs a fixed slice of the side work on every segment:
//     the two ds_read_b128 of the segment two ahead (ring of three B operands), one weight-fragment load of the NEXT chunk (second
//     register set), the input loads of the chunk AFTER next, and the convert / split / ds_write of one staging task of the next
//     chunk (its loads were issued a whole chunk earlier).  sched_barrier(0) pins the segments; nothing waits inside the loop.
// One wave per SIMD (the two weight sets + 64 accumulators need ~330 registers of the 512).
#define CV_KEEP4(u) asm volatile("" : : "v"((u).x), "v"((u).y), "v"((u).z), "v"((u).w))
template <bool HAS_S>
__global__ void __launch_bounds__(256, 1) conv3x3_il_kernel(ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][CV_PIX * CV_REC];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, h = lane >> 5;
    const int bw = a.W / CV_COLS;
    const int px = blockIdx.x % bw, py = blockIdx.x / bw;
    const int x0 = px * CV_COLS, y0 = py * CV_ROWS;
    const int mt = blockIdx.y * 4 + wave;          // this wave's 32-row tile of output channels (all four rows of the pixel tile)
    const int b = blockIdx.z / a.ksplit, ks = blockIdx.z - b * a.ksplit;
    const int H = a.H, W = a.W, Cin = a.Cin, NCT = Cin / 16, MT = a.Cout / 32;
    const int c_lo = (NCT * ks) / a.ksplit, c_hi = (NCT * (ks + 1)) / a.ksplit, NC = c_hi - c_lo;
    const float* xb = a.x + (int64_t)b * Cin * H * W;
    const float* sb = HAS_S ? a.s + (int64_t)b * Cin : nullptr;
    float in_sc = 1.0f, out_sc = 1.0f / CV_WSHIFT;
    if (a.in_amax) {          // see conv3x3_split_kernel
        const int e = amax_pow2(a.in_amax, lane, HAS_S ? wave_absmax(sb, Cin, lane) : 1.0f);
        in_sc = pow2f(e);
        out_sc = pow2f(-e - 8);
    }
    // staging tasks (pixel of the patch, channel pair); padding / idle slots load a valid address and are zeroed by their factor
    int t_off[CV_TPT], t_lds[CV_TPT], t_cp[CV_TPT];
    float t_m[CV_TPT];
#pragma unroll
    for (int q = 0; q < CV_TPT; ++q) {
        const int task = tid + 256 * q;
        const int cp = task / CV_PIX, p = task - cp * CV_PIX;
        const int pr = p / CV_PC, pc = p - pr * CV_PC;
        const int gy = y0 + pr - 1, gx = x0 + pc - 1;
        const bool ok = task < CV_TASKS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        t_off[q] = ok ? (2 * cp) * H * W + gy * W + gx : 0;
        t_m[q] = ok ? in_sc : 0.f;
        t_lds[q] = task < CV_TASKS ? p * CV_REC + cp : CV_PIX * CV_REC;          // idle slots of the last round: caught below
        t_cp[q] = task < CV_TASKS ? 2 * cp : 0;
    }
    auto fetch1 = [&](int cc, int q, float (&v)[CV_TPT][2], float (&sc)[CV_TPT][2]) {
        const float* src = xb + (int64_t)(16 * cc) * H * W;
        v[q][0] = src[t_off[q]];
        v[q][1] = src[t_off[q] + H * W];
        if (HAS_S) { sc[q][0] = sb[16 * cc + t_cp[q]]; sc[q][1] = sb[16 * cc + t_cp[q] + 1]; }
    };
    auto stash1 = [&](uint32_t* L, int q, const float (&v)[CV_TPT][2], const float (&sc)[CV_TPT][2]) {
        if (q == CV_TPT - 1 && t_lds[q] >= CV_PIX * CV_REC) return;
        const fl2_t f = {HAS_S ? v[q][0] * (sc[q][0] * t_m[q]) : v[q][0] * t_m[q], HAS_S ? v[q][1] * (sc[q][1] * t_m[q]) : v[q][1] * t_m[q]};
        const h2_t hi = __builtin_convertvector(f, h2_t);
        const h2_t lo = __builtin_convertvector(f - __builtin_convertvector(hi, fl2_t), h2_t);
        L[t_lds[q]] = __builtin_bit_cast(uint32_t, hi);
        L[t_lds[q] + 8] = __builtin_bit_cast(uint32_t, lo);
    };
    auto aload = [&](int cc, int idx, uint4 (&A)[9][2]) {          // fragment idx = 2 t + part of chunk cc
        const uint4* ab = a.blob + ((int64_t)(cc * 9) * MT + mt) * 128 + lane;
        A[idx >> 1][idx & 1] = ab[(int64_t)(idx >> 1) * MT * 128 + 64 * (idx & 1)];
    };
    f32x16 acc[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rr][r] = 0.f;

    uint4 A0[9][2], A1[9][2];
    float v0[CV_TPT][2], v1[CV_TPT][2], s0[CV_TPT][2], s1[CV_TPT][2];
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, ps[4] = {0, 0, 0, 0};
    // prologue: chunk c_lo's patch goes to buffer 0; its fragments to A0; chunk c_lo+1's inputs are in flight in v0
#pragma unroll
    for (int q = 0; q < CV_TPT; ++q) fetch1(c_lo, q, v1, s1);
#pragma unroll
    for (int idx = 0; idx < 18; ++idx) aload(c_lo, idx, A0);
#pragma unroll
    for (int q = 0; q < CV_TPT; ++q) stash1(lds[0], q, v1, s1);
    {
        const int c1 = NC > 1 ? c_lo + 1 : c_lo;
#pragma unroll
        for (int q = 0; q < CV_TPT; ++q) fetch1(c1, q, v0, s0);
    }
    __syncthreads();

    // one chunk: MFMAs on (Ac, lds[buf]); side work: An <- fragments of chunk cn1, vn <- inputs of chunk cn2, vc -> lds[buf ^ 1]
    auto chunk = [&](int buf, int cn1, int cn2, uint4 (&Ac)[9][2], uint4 (&An)[9][2], float (&vc)[CV_TPT][2], float (&sc_)[CV_TPT][2],
                     float (&vn)[CV_TPT][2], float (&sn)[CV_TPT][2]) {
        const uint32_t* L = lds[buf] + j * CV_REC + 4 * h;
        uint32_t* Lw = lds[buf ^ 1];
        CV_T(pt0);
        uint4 Bq[3][2];
        auto bload = [&](int k, uint4 (&d)[2]) {          // segment k = 4 t + rr
            const int t = k >> 2, rr = k & 3, ky = t / 3, kx = t - 3 * ky;
            d[0] = *reinterpret_cast<const uint4*>(L + ((rr + ky) * CV_PC + kx) * CV_REC);
            d[1] = *reinterpret_cast<const uint4*>(L + ((rr + ky) * CV_PC + kx) * CV_REC + 8);
        };
        bload(0, Bq[0]);
        bload(1, Bq[1]);
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            const int t = k >> 2, rr = k & 3;
            const f16x8_t ah = __builtin_bit_cast(f16x8_t, Ac[t][0]), al = __builtin_bit_cast(f16x8_t, Ac[t][1]);
            const f16x8_t xh = __builtin_bit_cast(f16x8_t, Bq[k % 3][0]), xl = __builtin_bit_cast(f16x8_t, Bq[k % 3][1]);
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc[rr], 0, 0, 0);
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc[rr], 0, 0, 0);
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc[rr], 0, 0, 0);
            if (k + 2 < 36) bload(k + 2, Bq[(k + 2) % 3]);          // slot last read by segment k-1: its MFMAs issued >= 96 cycles ago
            if ((k & 1) == 0) aload(cn1, k >> 1, An);               // 18 fragments of the next chunk, one per even segment
            else if (k < 2 * CV_TPT) fetch1(cn2, k >> 1, vn, sn);   // inputs of the chunk after next, one task per odd segment
            if (k >= 15 && (k - 15) % 3 == 0 && (k - 15) / 3 < CV_TPT) stash1(Lw, (k - 15) / 3, vc, sc_);          // k = 15, 18, .., 33
            __builtin_amdgcn_sched_barrier(0);
        }
        // the matrix instructions keep reading their operand registers for a while after issue (docs/history/DESIGN_r1-r4.md 3.5): the fragment
        // registers stay allocated until here (no temporary may land in them), and the pipe drains before the next chunk's first writes
#pragma unroll
        for (int t = 0; t < 9; ++t) { CV_KEEP4(Ac[t][0]); CV_KEEP4(Ac[t][1]); }
        asm volatile(CV_MFMA_PAD : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        CV_T(pt1);
        __syncthreads();
        CV_T(pt2);
#ifdef CV_PROFILE
        ps[1] += pt1 - pt0; ps[3] += pt2 - pt1;
#endif
    };
    for (int ci = 0; ci < NC; ci += 2) {
        const int cc = c_lo + ci;
        const int n1 = ci + 1 < NC ? cc + 1 : cc, n2 = ci + 2 < NC ? cc + 2 : cc, n3 = ci + 3 < NC ? cc + 3 : cc;
        chunk(0, n1, n2, A0, A1, v0, s0, v1, s1);
        if (ci + 1 < NC) chunk(1, n2, n3, A1, A0, v1, s1, v0, s0);
    }
#ifdef CV_PROFILE
    if (g_cv_prof && lane == 0) {
        const int w = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
        for (int q = 0; q < 4; ++q) g_cv_prof[(size_t)w * 4 + q] = ps[q];
    }
#endif

    if (a.partial) {
        float* pp = a.partial + (int64_t)ks * a.B * a.Cout * H * W;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                pp[(((int64_t)b * a.Cout + co) * H + y0 + rr) * W + x0 + j] = acc[rr][r] * out_sc;
            }
        return;
    }
    const float nw = (a.noise && a.noise_weight) ? *a.noise_weight : 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int gy = y0 + rr, gx = x0 + j;
        const float nz = a.noise ? a.noise[(a.noise_batched ? (int64_t)b * H * W : 0) + (int64_t)gy * W + gx] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[rr][r] * out_sc;
            if (a.d) v = v * a.d[(int64_t)b * a.Cout + co];
            if (a.noise) v = v + nw * nz;
            if (a.bias) v = v + a.bias[co];
            if (a.act) v = (v > 0.f ? v : v * a.slope) * a.gain;
            a.y[(((int64_t)b * a.Cout + co) * H + gy) * W + gx] = v;
        }
    }
}

// max |x| over a tensor as HAV_ABSMAX_WORDS partial maxima (bit patterns of non-negative floats; NaNs are skipped by fmaxf): block k
// owns the k-th slice and stores its maximum -- no atomics, nothing to zero beforehand, so the whole range control is ONE launch; the
// consumer (conv3x3_split_kernel) folds the words with one load per thread and a wave reduction.
__global__ void __launch_bounds__(256) absmax_kernel(unsigned int* __restrict__ out, const float* __restrict__ x, int64_t n)
{
    __shared__ float red[4];
    float m = 0.f;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = per * blockIdx.x, hi = (lo + per < n4) ? lo + per : n4;
    int64_t i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {          // four independent 16-byte loads in flight per thread
        const float4 v0 = x4[i], v1 = x4[i + 256], v2 = x4[i + 512], v3 = x4[i + 768];
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))),
                           fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(v2.x), fabsf(v2.y)), fmaxf(fabsf(v2.z), fabsf(v2.w))),
                           fmaxf(fmaxf(fabsf(v3.x), fabsf(v3.y)), fmaxf(fabsf(v3.z), fabsf(v3.w)))));
    }
    for (; i < hi; i += 256) {
        const float4 v = x4[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

extern "C" int hav_absmax(void* out_bits, const float* x, int64_t n, void* stream)
{
    if (!out_bits || !x || n < 0) return HAV_EINVAL;
    if (((uintptr_t)x & 15) != 0) return HAV_EUNSUP;          // float4 loads (every tensor this library is handed is 16-byte aligned)
    hipLaunchKernelGGL(absmax_kernel, dim3(HAV_ABSMAX_WORDS), dim3(256), 0, (hipStream_t)stream, (unsigned int*)out_bits, x, n);
    HAV_LAUNCH_CHECK();
    return 0;
}

// is-finite probe of the graphed training step's tracer (include/havatar.h: hav_debug_nonfinite): exponent field all ones = Inf / NaN
__global__ void __launch_bounds__(256) nonfinite_kernel(unsigned int* __restrict__ flag, const unsigned int* __restrict__ x, int64_t n)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= ((x[i] >> 23) & 0xFFu) == 0xFFu;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

extern "C" int hav_debug_nonfinite(void* flag, const void* x, int64_t n, void* stream)
{
    if (!flag || !x || n < 0) return HAV_EINVAL;
    if (n == 0) return 0;
    const int64_t need = (n + 255) / 256;
    hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)(need < 1024 ? need : 1024)), dim3(256), 0, (hipStream_t)stream, (unsigned int*)flag,
                       (const unsigned int*)x, n);
    HAV_LAUNCH_CHECK();
    return 0;
}

// K-split epilogue: y = act(d * (sum of the slices, in slice order) + nw * noise + bias) * gain
__global__ void __launch_bounds__(256) conv3x3_finish_kernel(ConvArgs a, int64_t total)
{
    const float nw = (a.noise && a.noise_weight) ? *a.noise_weight : 0.f;
    const int64_t HW = (int64_t)a.H * a.W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t plane = e / HW, p = e - plane * HW;
        const int co = (int)(plane % a.Cout);
        const int64_t b = plane / a.Cout;
        float v = 0.f;
        for (int k = 0; k < a.ksplit; ++k) v += a.partial[(int64_t)k * total + e];
        if (a.d) v = v * a.d[plane];
        if (a.noise) v = v + nw * a.noise[(a.noise_batched ? b * HW : 0) + p];
        if (a.bias) v = v + a.bias[co];
        if (a.act) v = (v > 0.f ? v : v * a.slope) * a.gain;
        a.y[e] = v;
    }
}

// scratch the K-split path needs for these sizes (0: no K-split, pass NULL)
// which kernel: the interleaved 128 x 128 one wherever Cout allows it; HAVATAR_CONV_KERNEL=0 forces the plain 64 x 128 kernel (A/B
// runs; read once)
static bool conv_interleaved(int Cout)
{
    static const int forced = [] { const char* e = getenv("HAVATAR_CONV_KERNEL"); return e ? atoi(e) : -1; }();
    return forced != 0 && (Cout % 128) == 0;
}
static bool conv_narrow(int H, int W) { return (W % CV_COLS) != 0 && (W % 16) == 0 && (H % 8) == 0; }          // 8 x 16 tiles (plain kernel only)
static bool conv_shape_ok(int H, int W) { return ((H % CV_ROWS) == 0 && (W % CV_COLS) == 0) || conv_narrow(H, W); }
static int conv_ksplit(int B, int Cin, int Cout, int H, int W)
{
    const bool il = conv_interleaved(Cout) && !conv_narrow(H, W);
    const int64_t tiles = (int64_t)B * (Cout / (il ? 128 : 64)) * ((int64_t)H * W / (CV_ROWS * CV_COLS));
    // workgroups wanted before the K dimension is split (A/B: HAVATAR_CONV_KSPLIT_CUS, read once; default = the number of compute units)
    static const int want = [] { const char* e = getenv("HAVATAR_CONV_KSPLIT_CUS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : hav_num_cus(); }();
    int ks = 1;
    if (il) { while (tiles * ks < want && ks < 8 && (Cin / 16) / (ks * 2) >= 2) ks *= 2; }
    else { while (tiles * ks < want && ks < 4 && (Cin / 16) / (ks * 2) >= 4) ks *= 2; }
    return ks;
}
extern "C" int64_t hav_conv3x3_scratch_bytes(int B, int Cin, int Cout, int H, int W)
{
    if (B < 1 || Cin < 16 || Cout < 64 || H < 1 || W < 1 || (Cin % 16) || (Cout % 64) || !conv_shape_ok(H, W)) return 0;
    const int ks = conv_ksplit(B, Cin, Cout, H, W);
    return ks > 1 ? (int64_t)ks * B * Cout * H * W * 4 : 0;
}

extern "C" int hav_conv3x3_split(float* y, const float* x, const void* packed, const float* s, const float* d, const float* noise,
                                 const float* noise_weight, const float* bias, float slope, float gain, int act, int noise_batched, int B,
                                 int Cin, int Cout, int H, int W, void* scratch, const void* in_amax, void* stream)
{
    if (!y || !x || !packed || B < 1 || Cin < 16 || Cout < 64 || H < 1 || W < 1) return HAV_EINVAL;
    if ((Cin % 16) || (Cout % 64) || !conv_shape_ok(H, W)) return HAV_EUNSUP;
    ConvArgs a;
    a.in_amax = (const unsigned int*)in_amax;
    a.ksplit = scratch ? conv_ksplit(B, Cin, Cout, H, W) : 1;
    a.partial = a.ksplit > 1 ? (float*)scratch : nullptr;
    a.y = y; a.x = x; a.blob = (const uint4*)packed; a.s = s; a.d = d; a.noise = noise; a.noise_weight = noise_weight; a.bias = bias;
    a.slope = slope; a.gain = gain; a.act = act; a.noise_batched = noise_batched;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    const bool narrow = conv_narrow(H, W), il = conv_interleaved(Cout) && !narrow;
    const dim3 grid((unsigned)((int64_t)H * W / (CV_ROWS * CV_COLS)), (unsigned)(Cout / (il ? 128 : 64)), (unsigned)(B * a.ksplit));
    if (narrow) {
        if (s) hipLaunchKernelGGL((conv3x3_split_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv3x3_split_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else if (il) {
        if (s) hipLaunchKernelGGL(conv3x3_il_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(conv3x3_il_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
        if (s) hipLaunchKernelGGL((conv3x3_split_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((conv3x3_split_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    HAV_LAUNCH_CHECK();
    if (a.partial) {
        const int64_t total = (int64_t)B * Cout * H * W;
        int64_t blocks = (total + 255) / 256;
        if (blocks > (int64_t)hav_num_cus() * 16) blocks = (int64_t)hav_num_cus() * 16;
        hipLaunchKernelGGL(conv3x3_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, total);
        HAV_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Stride-2 3x3 convolution (the down-sampling ConvLayer / ConvBlock of the encoders: Blur -> EqualConv2d(stride 2, padding 0) ->
// FusedLeakyReLU, model/styleUnet.py:326-368; MIOpen runs it as Im2d2Col + an fp32 GEMM + the activation launch) on the same split-fp16
// path and the same packed weights as the stride-1 kernel.  Workgroup = 64 (Cout) x [4 rows x 32 columns] of OUTPUT, so the chunk's
// input patch is 9 x 65 pixels; it is staged with even and odd columns apart ([row][column parity][column / 2]): tap (ky, kx) of output
// column j reads patch column 2 j + kx = entry j + (kx >> 1) of parity kx & 1, i.e. consecutive lanes read consecutive 80-byte records
// as in the stride-1 kernel (records two apart would put 16 lanes on 8 bank groups).  A stride-2 patch feeds 2.25 taps per staged pixel
// where a stride-1 patch feeds 9, so the staging is what has to be hidden.  Second version (round 4): 512 threads.  The 18 (tap, output
// row) pairs of a wave's former work are dealt to TWO waves (9 each: 27 MFMAs per chunk and wave), which halves the staging tasks and the
// weight-fragment loads per thread and puts two waves on every SIMD -- one stages / waits on LDS while the other one's MFMAs run
// (the first version had one wave per SIMD: its loads, conversions, LDS round trips and MFMAs simply added up: 96.7 us on the 256 -> 512
// layer at 129^2 -> 64^2).  The two halves of a sum meet in LDS after the last chunk; the weight fragments of chunk c + 1 and the
// B operands of pair p + 1 are requested before the MFMAs of chunk c / pair p.  Maps with too few tiles for 256 CUs split the channel range
// over 2-8 workgroups per tile; conv3x3_finish_kernel adds the slices in slice order and applies the epilogue.
#define S2_PR (2 * CV_ROWS + 1)            // 9 patch rows
#define S2_HC (CV_COLS + 1)                // 33 entries per (row, parity)
#define S2_RECS (S2_PR * 2 * S2_HC)        // 594 records (9 of them -- odd column 65 -- never read)
#define S2_TASKS (S2_RECS * 2)            // (record, half of its 8 channel pairs): 8 values in, two 16-byte LDS writes out
#define S2_THREADS 512
#define S2_TPT ((S2_TASKS + S2_THREADS - 1) / S2_THREADS)    // 3
struct ConvS2Args {
    float* y; const float* x; const uint4* blob;
    float* partial; int ksplit;          // K-split: [ksplit][B,Cout,Hout,Wout] raw sums, reduced by conv3x3_finish_kernel
    const unsigned int* in_amax;
    const float* s; const float* d; const float* noise; const float* noise_weight; const float* bias;
    float slope, gain;
    int act, noise_batched;
    int B, Cin, Cout, Hin, Win, Hout, Wout, pad;
};

template <bool HAS_S>
__global__ void __launch_bounds__(S2_THREADS, 1) conv3x3s2_split_kernel(ConvS2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s2_lds[];          // [2][S2_RECS * CV_REC]
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = (wave >> 1) & 1, kh = wave >> 2;
    const int bw = a.Wout / CV_COLS;
    const int px = blockIdx.x % bw, py = blockIdx.x / bw;
    const int x0 = px * CV_COLS, y0 = py * CV_ROWS;          // output tile origin
    const int mt = blockIdx.y * 2 + wm;
    const int b = blockIdx.z / a.ksplit, ks = blockIdx.z - b * a.ksplit;
    const int Hin = a.Hin, Win = a.Win, Cin = a.Cin, NCT = Cin / 16, MT = a.Cout / 32;
    const int c_lo = (NCT * ks) / a.ksplit, c_hi = (NCT * (ks + 1)) / a.ksplit;
    const int64_t HWin = (int64_t)Hin * Win;
    const float* xb = a.x + (int64_t)b * Cin * HWin;
    const float* sb = HAS_S ? a.s + (int64_t)b * Cin : nullptr;
    float in_sc = 1.0f, out_sc = 1.0f / CV_WSHIFT;
    if (a.in_amax) {          // see conv3x3_split_kernel
        const int e = amax_pow2(a.in_amax, lane, HAS_S ? wave_absmax(sb, Cin, lane) : 1.0f);
        in_sc = pow2f(e);
        out_sc = pow2f(-e - 8);
    }
    // staging tasks: (record of the patch, half of the chunk's 16 channels).  The 8 split values of a task leave as one 16-byte LDS write for the
    // high and one for the low parts, at the records' 80-byte stride -- conflict-free (channel pairs written with 4-byte stores at that stride
    // were 4-way bank conflicts, 20 writes per thread and chunk)
    int t_off[S2_TPT], t_lds[S2_TPT];
    bool t_ok[S2_TPT];
#pragma unroll
    for (int q = 0; q < S2_TPT; ++q) {
        const int task = tid + S2_THREADS * q;
        const int qd = task / S2_RECS, rec = task - qd * S2_RECS;
        const int prow = rec / (2 * S2_HC), rem = rec - prow * (2 * S2_HC);
        const int par = rem / S2_HC, half = rem - par * S2_HC;
        const int pcol = 2 * half + par;
        const int gy = 2 * y0 + prow - a.pad, gx = 2 * x0 + pcol - a.pad;
        t_ok[q] = task < S2_TASKS && pcol <= 2 * CV_COLS && gy >= 0 && gy < Hin && gx >= 0 && gx < Win;
        t_off[q] = t_ok[q] ? (int)((8 * qd) * HWin + (int64_t)gy * Win + gx) : 0;
        t_lds[q] = task < S2_TASKS ? rec * CV_REC + 4 * qd : -1;
    }
    float sv[S2_TPT][8];
    auto fetch = [&](int cc, float (&v)[S2_TPT][8]) {
        const float* src = xb + (int64_t)(16 * cc) * HWin;
#pragma unroll
        for (int q = 0; q < S2_TPT; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[q][e] = t_ok[q] ? src[t_off[q] + e * HWin] : 0.f;
    };
    auto stash = [&](int buf, int cc, const float (&v)[S2_TPT][8]) {
        uint32_t* L = s2_lds + buf * (S2_RECS * CV_REC);
#pragma unroll
        for (int q = 0; q < S2_TPT; ++q) {
            if (t_lds[q] < 0) continue;
            const int qd = (tid + S2_THREADS * q) / S2_RECS;
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m0 = in_sc, m1 = in_sc;
                if (HAS_S) { m0 *= sb[16 * cc + 8 * qd + 2 * e]; m1 *= sb[16 * cc + 8 * qd + 2 * e + 1]; }          // L1 hits, issued behind the matrix work
                const fl2_t f = {v[q][2 * e] * m0, v[q][2 * e + 1] * m1};
                const h2_t hi = __builtin_convertvector(f, h2_t);
                const h2_t lo = __builtin_convertvector(f - __builtin_convertvector(hi, fl2_t), h2_t);
                hw[e] = __builtin_bit_cast(uint32_t, hi);
                lw[e] = __builtin_bit_cast(uint32_t, lo);
            }
            *reinterpret_cast<uint4*>(L + t_lds[q]) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *reinterpret_cast<uint4*>(L + t_lds[q] + 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    // this wave's nine (tap, row) pairs: pair p = 9 kh + i  ->  tap p >> 1, row p & 1; its five taps are 4 kh .. 4 kh + 4.  kh is a
    // template value inside `run` (register arrays want compile-time indices), and the chunk loop is unrolled by two so that the
    // two fragment sets swap roles by name.
    auto run = [&](auto KH) {
        constexpr int khc = decltype(KH)::value;
        uint4 A0[5][2], A1[5][2];
        auto load_a = [&](int cc, uint4 (&F)[5][2]) {
            const uint4* ab = a.blob + ((int64_t)(cc * 9 + 4 * khc) * MT + mt) * 128 + lane;
#pragma unroll
            for (int t = 0; t < 5; ++t) { F[t][0] = ab[(int64_t)t * MT * 128]; F[t][1] = ab[(int64_t)t * MT * 128 + 64]; }
        };
        int rec_off[9];          // LDS offset (in uint32) of pair i's B operand, lane part included
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int p = 9 * khc + i, t = p >> 1, rr = p & 1, ky = t / 3, kx = t - 3 * ky;
            rec_off[i] = (((2 * (2 * wn + rr) + ky) * 2 + (kx & 1)) * S2_HC + j + (kx >> 1)) * CV_REC + 4 * h;
        }
        // input prefetch runs TWO chunks ahead (the maps come from L2 / HBM: one chunk of 27 MFMAs per wave is shorter than that
        // round trip), the weight fragments (L2-resident, shared by every workgroup) one chunk ahead
        auto step = [&](int cc, int buf, const uint4 (&F)[5][2], uint4 (&Fn)[5][2], float (&Sc)[S2_TPT][8], float (&So)[S2_TPT][8]) {
            if (cc + 2 < c_hi) fetch(cc + 2, So);
            if (cc + 1 < c_hi) load_a(cc + 1, Fn);
            const uint32_t* L = s2_lds + buf * (S2_RECS * CV_REC);
            uint4 bh[2], bl[2];
            bh[0] = *reinterpret_cast<const uint4*>(L + rec_off[0]);
            bl[0] = *reinterpret_cast<const uint4*>(L + rec_off[0] + 8);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i + 1 < 9) {
                    bh[(i + 1) & 1] = *reinterpret_cast<const uint4*>(L + rec_off[i + 1]);
                    bl[(i + 1) & 1] = *reinterpret_cast<const uint4*>(L + rec_off[i + 1] + 8);
                }
                const int tl = ((9 * khc + i) >> 1) - 4 * khc, rr = (9 * khc + i) & 1;          // compile-time after unrolling
                const f16x8_t ah = __builtin_bit_cast(f16x8_t, F[tl][0]), al = __builtin_bit_cast(f16x8_t, F[tl][1]);
                const f16x8_t xh = __builtin_bit_cast(f16x8_t, bh[i & 1]), xl = __builtin_bit_cast(f16x8_t, bl[i & 1]);
#ifdef S2_ABL_NOMFMA
                acc[rr][0] += (xh[0] + xl[1]) + (ah[0] + al[0]);
#else
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc[rr], 0, 0, 0);
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc[rr], 0, 0, 0);
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc[rr], 0, 0, 0);
#endif
            }
            asm volatile(CV_MFMA_PAD : "+v"(acc[0]), "+v"(acc[1]));
#ifndef S2_ABL_NOSTASH
            if (cc + 1 < c_hi) stash(buf ^ 1, cc + 1, Sc);
#endif
            __syncthreads();
        };
        float sv2[S2_TPT][8];
        fetch(c_lo, sv);
        load_a(c_lo, A0);
        stash(0, c_lo, sv);
        if (c_lo + 1 < c_hi) fetch(c_lo + 1, sv);
        __syncthreads();
        for (int cc = c_lo; cc < c_hi; cc += 2) {
            step(cc, 0, A0, A1, sv, sv2);
            if (cc + 1 < c_hi) step(cc + 1, 1, A1, A0, sv2, sv);
        }
    };
    if (kh) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
    // the two tap halves of a tile meet in LDS (the staging buffers are free after the loop's last barrier)
    float* red = reinterpret_cast<float*>(s2_lds);
    if (kh) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave & 3) * 2 + rr) * 16 + r) * 64 + lane] = acc[rr][r];
    }
    __syncthreads();
    if (kh) return;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rr][r] += red[((wave * 2 + rr) * 16 + r) * 64 + lane];

    const int Ho = a.Hout, Wo = a.Wout;
    if (a.partial) {            // K-split: raw sums out, the epilogue runs in conv3x3_finish_kernel after the slices are added up
        float* pp = a.partial + (int64_t)ks * a.B * a.Cout * Ho * Wo;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                pp[(((int64_t)b * a.Cout + co) * Ho + y0 + 2 * wn + rr) * Wo + x0 + j] = acc[rr][r] * out_sc;
            }
        return;
    }
    const float nw = (a.noise && a.noise_weight) ? *a.noise_weight : 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int gy = y0 + 2 * wn + rr, gx = x0 + j;
        const float nz = a.noise ? a.noise[(a.noise_batched ? (int64_t)b * Ho * Wo : 0) + (int64_t)gy * Wo + gx] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[rr][r] * out_sc;
            if (a.d) v = v * a.d[(int64_t)b * a.Cout + co];
            if (a.noise) v = v + nw * nz;
            if (a.bias) v = v + a.bias[co];
            if (a.act) v = (v > 0.f ? v : v * a.slope) * a.gain;
            a.y[(((int64_t)b * a.Cout + co) * Ho + gy) * Wo + gx] = v;
        }
    }
}

static int conv_s2_ksplit(int B, int Cin, int Cout, int Hout, int Wout)
{
    const int64_t tiles = (int64_t)B * (Cout / 64) * (Hout / CV_ROWS) * (Wout / CV_COLS);
    int ks = 1;
    while (tiles * ks < hav_num_cus() && ks < 8 && (Cin / 16) / (ks * 2) >= 2) ks *= 2;
    return ks;
}
extern "C" int64_t hav_conv3x3s2_scratch_bytes(int B, int Cin, int Cout, int Hin, int Win, int pad)
{
    if (B < 1 || Cin < 16 || Cout < 64 || Hin < 3 || Win < 3 || pad < 0 || pad > 1 || (Cin % 16) || (Cout % 64)) return 0;
    const int Hout = (Hin + 2 * pad - 3) / 2 + 1, Wout = (Win + 2 * pad - 3) / 2 + 1;
    if ((Hout % CV_ROWS) || (Wout % CV_COLS)) return 0;
    const int ks = conv_s2_ksplit(B, Cin, Cout, Hout, Wout);
    return ks > 1 ? (int64_t)ks * B * Cout * Hout * Wout * 4 : 0;
}

extern "C" int hav_conv3x3s2_split(float* y, const float* x, const void* packed, const float* s, const float* d, const float* noise,
                                   const float* noise_weight, const float* bias, float slope, float gain, int act, int noise_batched, int B,
                                   int Cin, int Cout, int Hin, int Win, int pad, void* scratch, const void* in_amax, void* stream)
{
    if (!y || !x || !packed || B < 1 || Cin < 16 || Cout < 64 || Hin < 3 || Win < 3 || pad < 0 || pad > 1) return HAV_EINVAL;
    const int Hout = (Hin + 2 * pad - 3) / 2 + 1, Wout = (Win + 2 * pad - 3) / 2 + 1;
    if ((Cin % 16) || (Cout % 64) || (Hout % CV_ROWS) || (Wout % CV_COLS)) return HAV_EUNSUP;
    if ((int64_t)Cin * Hin * Win > 0x7fffffffLL) return HAV_EUNSUP;          // 32-bit element offsets inside one sample
    ConvS2Args a;
    a.y = y; a.x = x; a.blob = (const uint4*)packed; a.in_amax = (const unsigned int*)in_amax;
    a.ksplit = scratch ? conv_s2_ksplit(B, Cin, Cout, Hout, Wout) : 1;
    a.partial = a.ksplit > 1 ? (float*)scratch : nullptr;
    a.s = s; a.d = d; a.noise = noise; a.noise_weight = noise_weight; a.bias = bias;
    a.slope = slope; a.gain = gain; a.act = act; a.noise_batched = noise_batched;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.pad = pad;
    const size_t lds = 2 * (size_t)S2_RECS * CV_REC * sizeof(uint32_t);          // 95 KB: dynamic (above the 64 KB static limit)
    static std::atomic<unsigned long long> attr_mask{0};          // per device: the attribute belongs to the device's copy of the kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (!((attr_mask.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t e1 = hipFuncSetAttribute((const void*)conv3x3s2_split_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute((const void*)conv3x3s2_split_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess || e2 != hipSuccess) return (int)(e1 != hipSuccess ? e1 : e2);
        attr_mask.fetch_or(1ull << dev, std::memory_order_release);
    }
    const dim3 grid((unsigned)((Wout / CV_COLS) * (Hout / CV_ROWS)), (unsigned)(Cout / 64), (unsigned)(B * a.ksplit));
    if (s) hipLaunchKernelGGL(conv3x3s2_split_kernel<true>, grid, dim3(S2_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(conv3x3s2_split_kernel<false>, grid, dim3(S2_THREADS), lds, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    if (a.partial) {
        ConvArgs f;
        f.y = y; f.x = nullptr; f.blob = nullptr; f.partial = a.partial; f.ksplit = a.ksplit; f.in_amax = nullptr;
        f.s = nullptr; f.d = d; f.noise = noise; f.noise_weight = noise_weight; f.bias = bias;
        f.slope = slope; f.gain = gain; f.act = act; f.noise_batched = noise_batched;
        f.B = B; f.Cin = Cin; f.Cout = Cout; f.H = Hout; f.W = Wout;
        const int64_t total = (int64_t)B * Cout * Hout * Wout;
        int64_t blocks = (total + 255) / 256;
        if (blocks > (int64_t)hav_num_cus() * 16) blocks = (int64_t)hav_num_cus() * 16;
        hipLaunchKernelGGL(conv3x3_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f, total);
        HAV_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The up-sampling StyledConv (model/styleUnet.py:236-243: conv_transpose2d(x * s, W^T, stride 2) -> Blur(4x4, pad (1,1)) -> demodulate ->
// noise -> bias -> leaky-ReLU) in two launches:
//   1. hav_gemm_split:   col[b, 9 o + t, p] = sum_i W[i, o, t] * (s[b,i] * x[b, i, p])        -- a [9 Cout x Cin] x [Cin x HW] product on the
//      split-fp16 matrix path (three products per k step, fp32 accumulate, as above).  That IS the transposed convolution before its
//      scatter: MIOpen runs the same product in fp32 (85 TFLOP/s), then Col2Im, and ATen / our upfirdn2d the blur and the epilogue.
//   2. hav_upconv_finish: the stride-2 scatter (col2im), the 4x4 FIR and the block's epilogue in one pass: a workgroup builds a tile of
//      the (2H+1) x (2W+1) transposed-convolution output in LDS -- every element gathers its <= 4 contributing taps, no atomics --
//      and filters it from there.
//
// GEMM tiling: workgroup = 4 waves = 128 (M) x 128 (N); wave (wm, wn) owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator VGPRs): per
// 16-wide k step a wave reads 4 A fragments (L2, pre-packed) and 4 B fragments (LDS) for 12 MFMAs.  K advances in chunks of 32; the
// chunk's [32 x 128] slab of x is modulated, split and written to LDS as 80-byte (pixel, 16-k group) records, double-buffered, with
// the next chunk's global loads in flight across the MFMAs (consumed only after them, cf. conv3x3_split_kernel).
#define GM_KC 32
#define GM_NT 128
#define GM_TPT ((GM_NT * (GM_KC / 2)) / 256)          // (pixel, channel pair) staging tasks per thread and chunk

extern "C" int64_t hav_gemm_packed_bytes(int M, int K) { return (int64_t)(K / 16) * (((M + 127) / 128) * 4) * 2 * 64 * 16; }

// fragment (k step kk, M tile m, part): lane (i, h) holds W[32m + i][16kk + 8h + e] * wmul * 2^8, e = 0..7 (rows >= M: zero)
__global__ void __launch_bounds__(256) gemm_pack_kernel(uint4* __restrict__ blob, const float* __restrict__ w, int M, int K, float wmul,
                                                        int64_t total)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), i = lane & 31, h = lane >> 5;
    int64_t q = idx >> 6;
    const int part = (int)(q & 1); q >>= 1;
    const int MT = ((M + 127) / 128) * 4;
    const int m = (int)(q % MT);
    const int kk = (int)(q / MT);
    const int row = 32 * m + i;
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) v[u] = row < M ? w[(int64_t)row * K + 16 * kk + 8 * h + 2 * d + u] * (wmul * CV_WSHIFT) : 0.f;
        const fl2_t f = {v[0], v[1]};
        const h2_t hi = __builtin_convertvector(f, h2_t);
        const h2_t lo = __builtin_convertvector(f - __builtin_convertvector(hi, fl2_t), h2_t);
        o[d] = __builtin_bit_cast(uint32_t, part ? lo : hi);
    }
    blob[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}

extern "C" int hav_gemm_pack(void* blob, const float* w, int M, int K, float wmul, void* stream)
{
    if (!blob || !w || M < 1 || K < 16) return HAV_EINVAL;
    if (K % GM_KC) return HAV_EUNSUP;
    const int64_t total = hav_gemm_packed_bytes(M, K) / 16;
    hipLaunchKernelGGL(gemm_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint4*)blob, w, M, K, wmul,
                       total);
    HAV_LAUNCH_CHECK();
    return 0;
}

struct GemmArgs {
    float* y; const float* x; const uint4* blob; const float* s;
    const unsigned int* in_amax;          // optional, as ConvArgs::in_amax: max |x| words -> power-of-two range control of s * x
    int B, M, K, N;
};

template <bool HAS_S>
__global__ void __launch_bounds__(256, 2) gemm_split_kernel(GemmArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][2 * GM_NT * CV_REC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int n0 = blockIdx.x * GM_NT;
    const int b = blockIdx.z;
    const int K = a.K, N = a.N, NC = K / GM_KC;
    const int MT = ((a.M + 127) / 128) * 4;
    const int mt0 = blockIdx.y * 4 + wm * 2;          // this wave's two 32-row M tiles
    const float* xb = a.x + (int64_t)b * K * N + n0;
    const float* sb = HAS_S ? a.s + (int64_t)b * K : nullptr;
    float in_sc = 1.0f, out_sc = 1.0f / CV_WSHIFT;
    if (a.in_amax) {          // see conv3x3_split_kernel
        const int e = amax_pow2(a.in_amax, lane, HAS_S ? wave_absmax(sb, K, lane) : 1.0f);
        in_sc = pow2f(e);
        out_sc = pow2f(-e - 8);
    }

    // staging task q (0, 1) of this thread: pixel p = tid & 127, octet o = (tid >> 7) + 2 q (0..3) = 8 consecutive channels (k group o >> 1, half
    // o & 1): the 8 split values leave as ONE 16-byte LDS write for the high and one for the low parts, at the lane stride (80 bytes) of the
    // reads -- conflict-free.  (Rounds 3-4 wrote channel PAIRS with 4-byte stores at that stride: 4-way bank conflicts, 16 writes per thread.)
    const int t_p = tid & (GM_NT - 1), t_c0 = tid >> 7;
    float sv[GM_TPT][2], ssc[GM_TPT][2];          // [2 q + (pair >> 2)... ] flattened: task q holds pairs 4 q .. 4 q + 3
    auto fetch = [&](int cc, float (&v)[GM_TPT][2], float (&sc)[GM_TPT][2]) {
        const float* src = xb + (int64_t)(GM_KC * cc) * N + t_p;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * (t_c0 + 2 * q) + 2 * e;
                v[4 * q + e][0] = src[(int64_t)k * N];
                v[4 * q + e][1] = src[(int64_t)(k + 1) * N];
                if (HAS_S) { sc[4 * q + e][0] = sb[GM_KC * cc + k] * in_sc; sc[4 * q + e][1] = sb[GM_KC * cc + k + 1] * in_sc; }
            }
    };
    auto stash = [&](int buf, const float (&v)[GM_TPT][2], const float (&sc)[GM_TPT][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = t_c0 + 2 * q, g = o >> 1, hf = o & 1;
            uint32_t hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const fl2_t f = {v[4 * q + e][0] * (HAS_S ? sc[4 * q + e][0] : in_sc), v[4 * q + e][1] * (HAS_S ? sc[4 * q + e][1] : in_sc)};
                const h2_t hi = __builtin_convertvector(f, h2_t);
                const h2_t lo = __builtin_convertvector(f - __builtin_convertvector(hi, fl2_t), h2_t);
                hw[e] = __builtin_bit_cast(uint32_t, hi);
                lw[e] = __builtin_bit_cast(uint32_t, lo);
            }
            uint32_t* at = &lds[buf][(g * GM_NT + t_p) * CV_REC + 4 * hf];
            *reinterpret_cast<uint4*>(at) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            *reinterpret_cast<uint4*>(at + 8) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.f; acc[0][1][r] = 0.f; acc[1][0][r] = 0.f; acc[1][1][r] = 0.f; }

    fetch(0, sv, ssc);
    stash(0, sv, ssc);
    __syncthreads();
    for (int cc = 0; cc < NC; ++cc) {
        const int buf = cc & 1;
        const uint4* ab = a.blob + ((int64_t)(2 * cc) * MT + mt0) * 128 + lane;
        uint4 A[2][2][2];          // [k step][M tile][part]
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                A[g][mi][0] = ab[((int64_t)g * MT + mi) * 128];
                A[g][mi][1] = ab[((int64_t)g * MT + mi) * 128 + 64];
            }
        if (cc + 1 < NC) fetch(cc + 1, sv, ssc);
        const uint32_t* L = lds[buf];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f16x8_t xh[2], xl[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int p = 64 * wn + 32 * ni + j;
                xh[ni] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(L + (g * GM_NT + p) * CV_REC + 4 * h));
                xl[ni] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(L + (g * GM_NT + p) * CV_REC + 8 + 4 * h));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const f16x8_t ah = __builtin_bit_cast(f16x8_t, A[g][mi][0]), al = __builtin_bit_cast(f16x8_t, A[g][mi][1]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ni], acc[mi][ni], 0, 0, 0);
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ni], acc[mi][ni], 0, 0, 0);
                }
            }
        }
        // the matrix instructions keep reading their operand registers after issue (docs/history/DESIGN_r1-r4.md 3.5): wait them out before stash() may
        // recycle registers
        asm volatile(CV_MFMA_PAD : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        if (cc + 1 < NC) stash(buf ^ 1, sv, ssc);
        __syncthreads();
    }
    float* yb = a.y + (int64_t)b * a.M * N + n0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * (mt0 + mi) + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < a.M) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) yb[(int64_t)row * N + 64 * wn + 32 * ni + j] = acc[mi][ni][r] * out_sc;
            }
        }
}

extern "C" int hav_gemm_split(float* y, const float* x, const void* packed, const float* s, const void* in_amax, int B, int M, int K, int N,
                              void* stream)
{
    if (!y || !x || !packed || B < 1 || M < 1 || K < GM_KC || N < GM_NT) return HAV_EINVAL;
    if ((K % GM_KC) || (N % GM_NT)) return HAV_EUNSUP;
    GemmArgs a;
    a.y = y; a.x = x; a.blob = (const uint4*)packed; a.s = s; a.in_amax = (const unsigned int*)in_amax; a.B = B; a.M = M; a.K = K; a.N = N;
    const dim3 grid((unsigned)(N / GM_NT), (unsigned)((M + 127) / 128), (unsigned)B);
    if (s) hipLaunchKernelGGL(gemm_split_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_split_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

// col2im (stride 2, 3x3, no padding) + 4x4 FIR with padding (1,1) + demodulation / noise / bias / leaky-ReLU, one pass.
//   z[P,Q]  = sum over taps (ky,kx) with P-ky, Q-kx even and inside the input:  col[o*9 + 3ky+kx][(P-ky)/2, (Q-kx)/2]     (2H+1 x 2W+1)
//   y[Y,X]  = sum_{t,u} fir[3-t][3-u] * zpad[Y+t, X+u],  zpad = z with a one-pixel zero border                           (2H x 2W)
// Workgroup = one (b, o) plane x UF_TR output rows: the z rows Y0-1 .. Y0+UF_TR+1 go to LDS first.
#define UF_TR 16
struct UpFinishArgs {
    float* y; const float* col; const float* fir; const float* d; const float* noise; const float* noise_weight; const float* bias;
    float slope, gain;
    int act, noise_batched, B, Cout, H, W;
};
__global__ void __launch_bounds__(256) upconv_finish_kernel(UpFinishArgs a)
{
    extern __shared__ float zt[];          // [UF_TR + 3][ZW], ZW = 2W + 4 (z columns -1 .. 2W+2; column q of z at index q + 1)
    const int H = a.H, W = a.W, OH = 2 * H, OW = 2 * W, ZW = OW + 4;
    const int o = blockIdx.y, b = blockIdx.z, Y0 = blockIdx.x * UF_TR;
    const float* cb = a.col + ((int64_t)b * a.Cout + o) * 9 * H * W;
    // z tile: the nine tap planes of this channel are streamed row by row with 16-byte loads (fully coalesced) and added into LDS tap by
    // tap -- within one tap every z element is touched by at most one thread, taps are separated by barriers, so the sum order is the
    // fixed tap order (ky asc, kx asc) of the definition.  (A gather per z element -- up to four strided 4-byte loads each -- ran at
    // 1.7 TB/s on the 256^2 -> 512^2 layer.)
    for (int e = threadIdx.x; e < (UF_TR + 3) * ZW; e += 256) zt[e] = 0.f;
    const int i_lo = max(Y0 / 2 - 1, 0), i_hi = min(Y0 / 2 + UF_TR / 2, H - 1), nrow = i_hi - i_lo + 1;
    const bool vec = (W & 3) == 0 && (((uintptr_t)cb) & 15) == 0;
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
        const float* tp = cb + (int64_t)t * H * W;
        if (vec) {
            const int w4 = W >> 2;
            for (int e = threadIdx.x; e < nrow * w4; e += 256) {
                const int r = e / w4, j4 = e - r * w4, i = i_lo + r;
                const int zr = 2 * i + ky - (Y0 - 1);
                if (zr < 0 || zr >= UF_TR + 3) continue;
                const float4 v = *reinterpret_cast<const float4*>(tp + (int64_t)i * W + 4 * j4);
                float* zrow = zt + zr * ZW + 1 + kx + 8 * j4;          // column Q = 2 j + kx at index Q + 1
                zrow[0] += v.x; zrow[2] += v.y; zrow[4] += v.z; zrow[6] += v.w;
            }
        } else {
            for (int e = threadIdx.x; e < nrow * W; e += 256) {
                const int r = e / W, jj = e - r * W, i = i_lo + r;
                const int zr = 2 * i + ky - (Y0 - 1);
                if (zr < 0 || zr >= UF_TR + 3) continue;
                zt[zr * ZW + 1 + kx + 2 * jj] += tp[(int64_t)i * W + jj];
            }
        }
        __syncthreads();
    }
    float f[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) f[q] = a.fir[15 - q];          // flipped: upfirdn2d is a true convolution
    const float nw = (a.noise && a.noise_weight) ? *a.noise_weight : 0.f;
    const float dd = a.d ? a.d[(int64_t)b * a.Cout + o] : 1.f, bb = a.bias ? a.bias[o] : 0.f;
    float* yb = a.y + ((int64_t)b * a.Cout + o) * OH * OW;
    for (int e = threadIdx.x; e < UF_TR * OW; e += 256) {
        const int r = e / OW, X = e - r * OW, Y = Y0 + r;
        if (Y >= OH) break;
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) v = fmaf(f[4 * t + u], zt[(r + t) * ZW + X + u], v);
        if (a.d) v = v * dd;
        if (a.noise) v = v + nw * a.noise[(a.noise_batched ? (int64_t)b * OH * OW : 0) + (int64_t)Y * OW + X];
        if (a.bias) v = v + bb;
        if (a.act) v = (v > 0.f ? v : v * a.slope) * a.gain;
        yb[(int64_t)Y * OW + X] = v;
    }
}

extern "C" int hav_upconv_finish(float* y, const float* col, const float* fir4x4, const float* d, const float* noise, const float* noise_weight,
                                 const float* bias, float slope, float gain, int act, int noise_batched, int B, int Cout, int H, int W,
                                 void* stream)
{
    if (!y || !col || !fir4x4 || B < 1 || Cout < 1 || H < 1 || W < 1) return HAV_EINVAL;
    const size_t lds = (size_t)(UF_TR + 3) * (2 * W + 4) * sizeof(float);
    if (lds > 64 * 1024) return HAV_EUNSUP;
    UpFinishArgs a;
    a.y = y; a.col = col; a.fir = fir4x4; a.d = d; a.noise = noise; a.noise_weight = noise_weight; a.bias = bias;
    a.slope = slope; a.gain = gain; a.act = act; a.noise_batched = noise_batched; a.B = B; a.Cout = Cout; a.H = H; a.W = W;
    hipLaunchKernelGGL(upconv_finish_kernel, dim3((unsigned)((2 * H + UF_TR - 1) / UF_TR), (unsigned)Cout, (unsigned)B), dim3(256), lds,
                       (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride 1 / padding 1 convolution on the same split-fp16 matrix path (training; reference: autograd of
// conv2d_gradfix.conv2d in model/styleUnet.py -- MIOpen's fp32 igemm_wrw + its NHWC transposes are 6.4 ms of a 31 ms step):
//
//   gw[o, i, ky, kx] = sum_{b, y, x} g[b, o, y, x] * xin[b, i, y + ky - 1, x + kx - 1]
//
// GEMM view: M = Cout, N = Cin, K = pixels -- per tap a different shift of the N operand.  Both operands are activations (nothing to
// pre-pack): both go through LDS as fp16 hi / lo.  Workgroup = 64 (o) x 32 (i) x 9 taps; wave (mo, tg) owns one 32-row M tile and taps
// 0-4 or 5-8 (5 / 4 accumulator tiles).  K advances one image-row segment of 16 pixels per step down a column strip (b, x0): the step
// needs g[.., y, x0..x0+15] and input rows y-1, y, y+1; rows live in a ring of four LDS slots, each row stored THREE times, shifted by
// kx - 1 = -1, 0, +1 pixels (an MFMA operand is 8 consecutive fp16 = one 16-byte LDS read, and a one-pixel shift is 2 bytes), so a
// step stages one new input row (x3) and one g row.  Column strips are dealt to gridDim.z workgroups per output block (K split);
// partial sums are laid out [z][tap][o][i] (lane = i: coalesced) and added up by conv3x3_wgrad_reduce_kernel into [o][i][3][3].
// g is gradient-sized (1e-6): it takes the same power-of-two range control as the forward kernels (in_amax of g).
#define WG_XI 52            // dwords per input channel in a row slot: 3 shifts x 16 (8 hi + 8 lo) + 4 pad (16 lanes -> 16 bank groups)
#define WG_GO 20            // dwords per output channel in a g buffer: 8 hi + 8 lo + 4 pad
struct WgradArgs {
    float* partial; const float* g; const float* x; const unsigned int* g_amax; const unsigned int* x_amax;
    const float* xs;          // optional [B, Cin]: the x operand is xs[b, i] * x (the modulated input of a ModulatedConv2d)
    int B, Cin, Cout, H, W, strips;
};

__global__ void __launch_bounds__(256, 2) conv3x3_wgrad_kernel(WgradArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t xs[4][32 * WG_XI];
    __shared__ __attribute__((aligned(16))) uint32_t gs[2][64 * WG_GO];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, h = lane >> 5;
    const int mo = wave & 1, tg = wave >> 1;
    const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 64;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
    const int sw = W / 16;          // strips per image
    // both operands are activations: each gets its own power of two (max |g|, max |x| -> [512, 1024)), undone together on the way out
    const int eg = a.g_amax ? amax_pow2(a.g_amax, lane, 1.0f) : 0;
    const int ex = a.x_amax ? amax_pow2(a.x_amax, lane, a.xs ? wave_absmax(a.xs, a.B * Cin, lane) : 1.0f) : 0;
    const float g_sc = pow2f(eg), x_sc = pow2f(ex);
    const float out_g = pow2f(-eg), out_x = pow2f(-ex);          // applied one after the other: |eg + ex| may pass 127
    // staging roles.  g: thread = (o = tid >> 2, 4 pixels q4 = tid & 3): one float4.  x: thread = (i = tid >> 3, pixel pair p8 = tid & 7)
    const int g_o = tid >> 2, g_q = tid & 3;
    const int x_i = tid >> 3, x_p = tid & 7;
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto split2 = [](float v0, float v1, uint32_t& hi, uint32_t& lo) {
        const fl2_t f = {v0, v1};
        const h2_t hh = __builtin_convertvector(f, h2_t);
        const h2_t ll = __builtin_convertvector(f - __builtin_convertvector(hh, fl2_t), h2_t);
        hi = __builtin_bit_cast(uint32_t, hh); lo = __builtin_bit_cast(uint32_t, ll);
    };
    // input row `row` of strip (b, x0) -> ring slot (row + 1) & 3, three shifted copies; rows outside the image are zeros.  fetch_*
    // only issue the loads; stash_* (conversion, LDS writes) run after the step's MFMAs so that the loads fly under them
    struct XR { float a0, a1, el, er, sc; };
    auto fetch_x = [&](int b, int x0, int row) {
        XR r = {0.f, 0.f, 0.f, 0.f, x_sc};
        if (a.xs) r.sc = x_sc * a.xs[(int64_t)b * Cin + i0 + x_i];
        if (row >= 0 && row < H) {
            const float* src = a.x + (((int64_t)b * Cin + i0 + x_i) * H + row) * W + x0;
            const float2 v = *reinterpret_cast<const float2*>(src + 2 * x_p);
            r.a0 = v.x; r.a1 = v.y;
            if (x_p == 0 && x0 > 0) r.el = src[-1];
            if (x_p == 7 && x0 + 16 < W) r.er = src[16];
        }
        return r;
    };
    auto stash_x = [&](int row, const XR& r) {
        uint32_t* dst = xs[(row + 1) & 3] + x_i * WG_XI;
        // neighbours inside the 8-thread group of this channel (lanes are consecutive): previous pair's second / next pair's first
        float pa1 = __shfl_up(r.a1, 1, 64), na0 = __shfl_down(r.a0, 1, 64);
        if (x_p == 0) pa1 = r.el;
        if (x_p == 7) na0 = r.er;
        const float a0 = r.a0 * r.sc, a1 = r.a1 * r.sc;
        pa1 *= r.sc; na0 *= r.sc;
        uint32_t hi, lo;
        split2(pa1, a0, hi, lo);  dst[0 * 16 + x_p] = hi; dst[0 * 16 + 8 + x_p] = lo;          // kx = 0: element j = x[x0 + j - 1]
        split2(a0, a1, hi, lo); dst[1 * 16 + x_p] = hi; dst[1 * 16 + 8 + x_p] = lo;          // kx = 1
        split2(a1, na0, hi, lo);  dst[2 * 16 + x_p] = hi; dst[2 * 16 + 8 + x_p] = lo;          // kx = 2: element j = x[x0 + j + 1]
    };
    auto fetch_g = [&](int b, int x0, int row) {
        return *reinterpret_cast<const float4*>(a.g + (((int64_t)b * Cout + o0 + g_o) * H + row) * W + x0 + 4 * g_q);
    };
    auto stash_g = [&](int buf, const float4& v) {
        uint32_t* dst = gs[buf] + g_o * WG_GO + 2 * g_q;
        uint32_t h0, l0, h1, l1;
        split2(v.x * g_sc, v.y * g_sc, h0, l0);
        split2(v.z * g_sc, v.w * g_sc, h1, l1);
        dst[0] = h0; dst[1] = h1; dst[8] = l0; dst[9] = l1;
    };

    for (int s = blockIdx.z; s < a.strips; s += gridDim.z) {
        const int b = s / sw, x0 = (s - b * sw) * 16;
        __syncthreads();          // the previous strip's last step is done with the ring
        stash_x(-1, fetch_x(b, x0, -1));
        stash_x(0, fetch_x(b, x0, 0));
        stash_x(1, fetch_x(b, x0, 1));
        stash_g(0, fetch_g(b, x0, 0));
        __syncthreads();
        for (int y = 0; y < H; ++y) {
            // next step's operands: input row y + 2 (its slot held row y - 2) and g row y + 1 (other buffer): loads now, LDS after the MFMAs
            const bool more = y + 1 < H;
            XR nx = {0.f, 0.f, 0.f, 0.f, 0.f};
            float4 ng = make_float4(0.f, 0.f, 0.f, 0.f);
            if (more) { nx = fetch_x(b, x0, y + 2); ng = fetch_g(b, x0, y + 1); }
            const uint32_t* G = gs[y & 1] + (32 * mo + j) * WG_GO + 4 * h;
            const f16x8_t gh = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(G));
            const f16x8_t gl = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(G + 8));
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int t = tg * 5 + q;          // tap (wave-uniform); tg = 1 has four (t = 5..8)
                if (t < 9) {
                    const int ky = t / 3, kx = t - 3 * ky;
                    const uint32_t* X = xs[(y + ky) & 3] + j * WG_XI + kx * 16 + 4 * h;          // row y + ky - 1 -> slot (y + ky) & 3
                    const f16x8_t xh = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(X));
                    const f16x8_t xl = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(X + 8));
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, xh, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xl, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xh, acc[q], 0, 0, 0);
                }
            }
            asm volatile(CV_MFMA_PAD : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));
            if (more) { stash_x(y + 2, nx); stash_g((y + 1) & 1, ng); }
            __syncthreads();
        }
    }
    // D[m = o][n = i]: lane j = i, registers = o rows.  partial[z][t][o][i]
    float* pp = a.partial + (int64_t)blockIdx.z * 9 * Cout * Cin;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int t = tg * 5 + q;
        if (t < 9) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * h;
                pp[((int64_t)t * Cout + o) * Cin + i0 + j] = acc[q][r] * out_g * out_x;
            }
        }
    }
}

// transpose: the partials are [o][i]; write gw as [i][o][3][3] (the stride-2 kernel below, called with the operands of a transposed convolution)
__global__ void __launch_bounds__(256) conv3x3_wgrad_reduce_kernel(float* __restrict__ gw, const float* __restrict__ partial, int ks, int Cout, int Cin,
                                                                   float out_mul, int transpose)
{
    const int64_t n = (int64_t)Cout * Cin;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float s = 0.f;
            for (int k = 0; k < ks; ++k) s += partial[((int64_t)k * 9 + t) * n + e];
            v[t] = s;
        }
        const int64_t d = transpose ? ((e % Cin) * Cout + e / Cin) : e;
#pragma unroll
        for (int t = 0; t < 9; ++t) gw[d * 9 + t] = v[t] * out_mul;
    }
}

#ifndef WGRAD_FILL
#define WGRAD_FILL 2
#endif
static int wgrad_ksplit(int B, int Cin, int Cout, int H, int W)
{
    const int blocks = (Cin / 32) * (Cout / 64), strips = B * (W / 16);
    int ks = 1;
    while (blocks * ks < WGRAD_FILL * hav_num_cus() && ks * 2 <= strips) ks *= 2;
    return ks;
}
extern "C" int64_t hav_conv3x3_wgrad_scratch_bytes(int B, int Cin, int Cout, int H, int W)
{
    if (B < 1 || Cin < 32 || Cout < 64 || H < 1 || W < 16 || (Cin % 32) || (Cout % 64) || (W % 16)) return 0;
    return (int64_t)wgrad_ksplit(B, Cin, Cout, H, W) * 9 * Cout * Cin * 4;
}
static int wgrad_launch(float* gw, const float* g, const float* x, const float* xs, float out_mul, void* scratch, const void* g_amax,
                        const void* x_amax, int B, int Cin, int Cout, int H, int W, void* stream);
extern "C" int hav_conv3x3_wgrad(float* gw, const float* g, const float* x, void* scratch, const void* g_amax, const void* x_amax, int B, int Cin,
                                 int Cout, int H, int W, void* stream)
{
    return wgrad_launch(gw, g, x, nullptr, 1.0f, scratch, g_amax, x_amax, B, Cin, Cout, H, W, stream);
}
// the same with the x operand modulated (xs [B, Cin] * x) and the result scaled by out_mul: the weight gradient of a ModulatedConv2d /
// EqualConv2d PARAMETER (out_mul = its 1 / sqrt(9 Cin) scale) in one go
extern "C" int hav_conv3x3_wgrad_mod(float* gw, const float* g, const float* x, const float* xs, float out_mul, void* scratch, const void* g_amax,
                                     const void* x_amax, int B, int Cin, int Cout, int H, int W, void* stream)
{
    return wgrad_launch(gw, g, x, xs, out_mul, scratch, g_amax, x_amax, B, Cin, Cout, H, W, stream);
}
static int wgrad_launch(float* gw, const float* g, const float* x, const float* xs, float out_mul, void* scratch, const void* g_amax,
                        const void* x_amax, int B, int Cin, int Cout, int H, int W, void* stream)
{
    if (!gw || !g || !x || !scratch || B < 1 || H < 1) return HAV_EINVAL;
    if (Cin < 32 || Cout < 64 || W < 16 || (Cin % 32) || (Cout % 64) || (W % 16)) return HAV_EUNSUP;
    WgradArgs a;
    a.partial = (float*)scratch; a.g = g; a.x = x; a.g_amax = (const unsigned int*)g_amax; a.x_amax = (const unsigned int*)x_amax; a.xs = xs;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.strips = B * (W / 16);
    const int ks = wgrad_ksplit(B, Cin, Cout, H, W);
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3((unsigned)(Cin / 32), (unsigned)(Cout / 64), (unsigned)ks), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    const int64_t n = (int64_t)Cout * Cin;
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 8) blocks = (int64_t)hav_num_cus() * 8;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gw, (const float*)scratch, ks, Cout, Cin, out_mul, 0);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the STRIDE-2 3x3 layers (training): one contraction serves both re-sampling layers of the StyleGAN blocks --
//
//   out[m, n, ky, kx] = sum_{b, y, x} S[b, m, y, x] * L[b, n, 2 y + ky, 2 x + kx]          S: [B,M,H,W]   L: [B,N,2H+1,2W+1]
//
//   * the down-sampling ConvLayer (Blur -> conv 3x3 stride 2 padding 0, reference model/styleUnet.py:326-368): S = the gradient at the
//     layer's output, L = the blurred input -> gw[Cout][Cin][3][3];
//   * the up-sampling StyledConv (conv_transpose2d stride 2, model/styleUnet.py:214-231): S = the modulated input s * x (ss = s), L = the
//     gradient at the transposed convolution's output -> [Cin][Cout][3][3], written transposed as the parameter's [Cout][Cin][3][3].
// MIOpen's route for these was igemm_wrw / CK fp32 kernels + NHWC transposes (3.7 ms of a 24 ms step).  Same scheme as conv3x3_wgrad_kernel
// above: M = S channels (64 per workgroup), N = L channels (32), K = pixels of S, one 16-pixel row segment per step; what changes is the N
// operand's staging -- a step needs L rows 2y .. 2y+2 (two NEW rows per step: a ring of eight slots) and, per row, the 33 columns
// 2 x0 .. 2 x0 + 32 de-interleaved into even / odd / even-shifted copies (kx = 0 / 1 / 2), again so that every MFMA operand is one 16-byte
// LDS read.  Both operands under the power-of-two range control (S or L is gradient-sized).
struct WgradS2Args {
    float* partial; const float* s; const float* l; const unsigned int* s_amax; const unsigned int* l_amax;
    const float* ss;          // optional [B, M]: the S operand is ss[b, m] * S
    int B, M, N, H, W, strips;
};

__global__ void __launch_bounds__(256, 2) conv3x3s2_wgrad_kernel(WgradS2Args a)
{
    __shared__ __attribute__((aligned(16))) uint32_t xs[8][32 * WG_XI];
    __shared__ __attribute__((aligned(16))) uint32_t gs[2][64 * WG_GO];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, h = lane >> 5;
    const int mo = wave & 1, tg = wave >> 1;
    const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 64;
    const int H = a.H, W = a.W, M = a.M, N = a.N, LH = 2 * H + 1, LW = 2 * W + 1;
    const int sw = W / 16;
    const int eg = a.s_amax ? amax_pow2(a.s_amax, lane, a.ss ? wave_absmax(a.ss, a.B * M, lane) : 1.0f) : 0;
    const int ex = a.l_amax ? amax_pow2(a.l_amax, lane, 1.0f) : 0;
    const float g_sc = pow2f(eg), x_sc = pow2f(ex);
    const float out_g = pow2f(-eg), out_x = pow2f(-ex);
    const int g_o = tid >> 2, g_q = tid & 3;          // S staging: thread = (channel, 4 pixels)
    const int x_i = tid >> 3, x_p = tid & 7;          // L staging: thread = (channel, S-pixel pair = 4 L columns + 1)
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto split2 = [](float v0, float v1, uint32_t& hi, uint32_t& lo) {
        const fl2_t f = {v0, v1};
        const h2_t hh = __builtin_convertvector(f, h2_t);
        const h2_t ll = __builtin_convertvector(f - __builtin_convertvector(hh, fl2_t), h2_t);
        hi = __builtin_bit_cast(uint32_t, hh); lo = __builtin_bit_cast(uint32_t, ll);
    };
    // L row `row` (always inside the map: rows 0 .. 2H) of strip (b, x0), columns 2 x0 + 4 x_p .. + 4: e0 o0 e1 o1 (+ e2 = the next thread's e0)
    struct LR { float e0, o0, e1, o1, er; };
    auto fetch_l = [&](int b, int x0, int row) {
        const float* src = a.l + (((int64_t)b * N + i0 + x_i) * LH + row) * LW + 2 * x0 + 4 * x_p;
        LR r;
        r.e0 = src[0]; r.o0 = src[1]; r.e1 = src[2]; r.o1 = src[3];
        r.er = (x_p == 7) ? src[4] : 0.f;          // column 2 x0 + 32 <= 2 W: always there
        return r;
    };
    auto stash_l = [&](int row, const LR& r) {
        uint32_t* dst = xs[row & 7] + x_i * WG_XI;
        float e2 = __shfl_down(r.e0, 1, 64);
        if (x_p == 7) e2 = r.er;
        uint32_t hi, lo;
        split2(r.e0 * x_sc, r.e1 * x_sc, hi, lo); dst[0 * 16 + x_p] = hi; dst[0 * 16 + 8 + x_p] = lo;          // kx = 0: element j = L[2 (x0 + j)]
        split2(r.o0 * x_sc, r.o1 * x_sc, hi, lo); dst[1 * 16 + x_p] = hi; dst[1 * 16 + 8 + x_p] = lo;          // kx = 1: L[2 (x0 + j) + 1]
        split2(r.e1 * x_sc, e2 * x_sc, hi, lo);   dst[2 * 16 + x_p] = hi; dst[2 * 16 + 8 + x_p] = lo;          // kx = 2: L[2 (x0 + j) + 2]
    };
    auto fetch_g = [&](int b, int x0, int row) {
        return *reinterpret_cast<const float4*>(a.s + (((int64_t)b * M + o0 + g_o) * H + row) * W + x0 + 4 * g_q);
    };
    auto stash_g = [&](int buf, const float4& v, float sc) {
        uint32_t* dst = gs[buf] + g_o * WG_GO + 2 * g_q;
        uint32_t h0, l0, h1, l1;
        split2(v.x * sc, v.y * sc, h0, l0);
        split2(v.z * sc, v.w * sc, h1, l1);
        dst[0] = h0; dst[1] = h1; dst[8] = l0; dst[9] = l1;
    };

    for (int s = blockIdx.z; s < a.strips; s += gridDim.z) {
        const int b = s / sw, x0 = (s - b * sw) * 16;
        const float sc = a.ss ? g_sc * a.ss[(int64_t)b * M + o0 + g_o] : g_sc;
        __syncthreads();          // the previous strip's last step is done with the ring
        stash_l(0, fetch_l(b, x0, 0));
        stash_l(1, fetch_l(b, x0, 1));
        stash_l(2, fetch_l(b, x0, 2));
        stash_g(0, fetch_g(b, x0, 0), sc);
        __syncthreads();
        for (int y = 0; y < H; ++y) {
            const bool more = y + 1 < H;
            LR n1 = {0.f, 0.f, 0.f, 0.f, 0.f}, n2 = {0.f, 0.f, 0.f, 0.f, 0.f};
            float4 ng = make_float4(0.f, 0.f, 0.f, 0.f);
            if (more) { n1 = fetch_l(b, x0, 2 * y + 3); n2 = fetch_l(b, x0, 2 * y + 4); ng = fetch_g(b, x0, y + 1); }
            const uint32_t* G = gs[y & 1] + (32 * mo + j) * WG_GO + 4 * h;
            const f16x8_t gh = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(G));
            const f16x8_t gl = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(G + 8));
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int t = tg * 5 + q;          // tap (wave-uniform); tg = 1 has four (t = 5..8)
                if (t < 9) {
                    const int ky = t / 3, kx = t - 3 * ky;
                    const uint32_t* X = xs[(2 * y + ky) & 7] + j * WG_XI + kx * 16 + 4 * h;
                    const f16x8_t xh = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(X));
                    const f16x8_t xl = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(X + 8));
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, xh, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xl, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, xh, acc[q], 0, 0, 0);
                }
            }
            asm volatile(CV_MFMA_PAD : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));
            if (more) { stash_l(2 * y + 3, n1); stash_l(2 * y + 4, n2); stash_g((y + 1) & 1, ng, sc); }          // slots of rows 2y-5 .. 2y-4: long done
            __syncthreads();
        }
    }
    float* pp = a.partial + (int64_t)blockIdx.z * 9 * M * N;          // partial[z][t][m][n], lane j = n
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int t = tg * 5 + q;
        if (t < 9) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * mo + (r & 3) + 8 * (r >> 2) + 4 * h;
                pp[((int64_t)t * M + o) * N + i0 + j] = acc[q][r] * out_g * out_x;
            }
        }
    }
}

// The same contraction for SMALL maps (the 4^2 -> 8^2 and 8^2 -> 16^2 up-sampling layers: W < 16, K = B H W <= 128 pixels, 512 x 512 x 9
// outputs): nothing for the matrix cores to chew on -- MIOpen's route was a 173 us CK kernel per layer -- so plain fp32 FMAs: a workgroup
// owns 16 x 16 (m, n) pairs, stages its S rows (modulation folded in) and L rows in LDS once, thread (m, n) accumulates the nine taps over
// all pixels in a fixed order (bit-reproducible, fp32-exact products: no range control needed).
struct WgradS2SmallArgs { float* gw; const float* s; const float* l; const float* ss; float out_mul; int transpose, B, M, N, H, W, lstride; };
__global__ void __launch_bounds__(256) conv3x3s2_wgrad_small_kernel(WgradS2SmallArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    const int B = a.B, M = a.M, N = a.N, H = a.H, W = a.W, HW = H * W, LH = 2 * H + 1, LW = 2 * W + 1, LHW = LH * LW;
    float* sS = wsm;                          // [16][B * HW]
    float* sL = wsm + 16 * B * HW;            // [16][lstride]  (lstride = B * LHW made odd: the 16 n of a wave hit 16 banks)
    const int tid = threadIdx.x, m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    for (int i = tid; i < 16 * B * HW; i += 256) {
        const int mi = i / (B * HW), r = i - mi * (B * HW), b = r / HW, p = r - b * HW;
        float v = a.s[((int64_t)b * M + m0 + mi) * HW + p];
        if (a.ss) v *= a.ss[(int64_t)b * M + m0 + mi];
        sS[i] = v;
    }
    for (int i = tid; i < 16 * B * LHW; i += 256) {
        const int ni = i / (B * LHW), r = i - ni * (B * LHW), b = r / LHW, p = r - b * LHW;
        sL[ni * a.lstride + r] = a.l[((int64_t)b * N + n0 + ni) * LHW + p];
    }
    __syncthreads();
    const int mi = tid >> 4, ni = tid & 15;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            const float* srow = sS + mi * (B * HW) + b * HW + y * W;
            const float* lrow = sL + ni * a.lstride + b * LHW + (2 * y) * LW;
            for (int x = 0; x < W; ++x) {
                const float sv = srow[x];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = fmaf(sv, lrow[ky * LW + 2 * x + kx], acc[ky * 3 + kx]);
            }
        }
    const int64_t d = a.transpose ? ((int64_t)(n0 + ni) * M + m0 + mi) : ((int64_t)(m0 + mi) * N + n0 + ni);
#pragma unroll
    for (int t = 0; t < 9; ++t) a.gw[d * 9 + t] = acc[t] * a.out_mul;
}

static bool wgrad_s2_shape_ok(int B, int M, int N, int H, int W)
{
    return B >= 1 && H >= 1 && M >= 64 && N >= 32 && W >= 16 && !(M % 64) && !(N % 32) && !(W % 16);
}
static int wgrad_s2_small_lstride(int B, int H, int W) { return (B * (2 * H + 1) * (2 * W + 1)) | 1; }
static bool wgrad_s2_small_ok(int B, int M, int N, int H, int W)
{
    if (B < 1 || H < 1 || W < 1 || W >= 16 || (M % 16) || (N % 16)) return false;
    return (int64_t)(16 * B * H * W + 16 * wgrad_s2_small_lstride(B, H, W)) * 4 <= 64 * 1024;
}
extern "C" int64_t hav_conv3x3s2_wgrad_scratch_bytes(int B, int M, int N, int H, int W)
{
    if (!wgrad_s2_shape_ok(B, M, N, H, W)) return 0;          // (0 as well for the small-map kernel, which needs none, and for unsupported shapes)
    return (int64_t)wgrad_ksplit(B, N, M, H, W) * 9 * M * N * 4;
}
extern "C" int hav_conv3x3s2_wgrad(float* gw, const float* S, const float* L, const float* ss, float out_mul, int transpose_out, void* scratch,
                                   const void* s_amax, const void* l_amax, int B, int M, int N, int H, int W, void* stream)
{
    if (!gw || !S || !L || B < 1 || H < 1) return HAV_EINVAL;
    if (wgrad_s2_small_ok(B, M, N, H, W)) {
        WgradS2SmallArgs q;
        q.gw = gw; q.s = S; q.l = L; q.ss = ss; q.out_mul = out_mul; q.transpose = transpose_out ? 1 : 0;
        q.B = B; q.M = M; q.N = N; q.H = H; q.W = W; q.lstride = wgrad_s2_small_lstride(B, H, W);
        const size_t lds = (size_t)(16 * B * H * W + 16 * q.lstride) * 4;
        hipLaunchKernelGGL(conv3x3s2_wgrad_small_kernel, dim3((unsigned)(N / 16), (unsigned)(M / 16)), dim3(256), lds, (hipStream_t)stream, q);
        HAV_LAUNCH_CHECK();
        return 0;
    }
    if (!scratch) return HAV_EINVAL;
    if (!wgrad_s2_shape_ok(B, M, N, H, W)) return HAV_EUNSUP;
    if (((uintptr_t)S & 15) != 0) return HAV_EUNSUP;          // float4 loads of S rows (W % 16 == 0 keeps every row aligned)
    WgradS2Args a;
    a.partial = (float*)scratch; a.s = S; a.l = L; a.s_amax = (const unsigned int*)s_amax; a.l_amax = (const unsigned int*)l_amax; a.ss = ss;
    a.B = B; a.M = M; a.N = N; a.H = H; a.W = W; a.strips = B * (W / 16);
    const int ks = wgrad_ksplit(B, N, M, H, W);
    hipLaunchKernelGGL(conv3x3s2_wgrad_kernel, dim3((unsigned)(N / 32), (unsigned)(M / 64), (unsigned)ks), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    const int64_t n = (int64_t)M * N;
    int64_t blocks = (n + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 8) blocks = (int64_t)hav_num_cus() * 8;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gw, (const float*)scratch, ks, M, N, out_mul,
                       transpose_out ? 1 : 0);
    HAV_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// Backward glue of a fused convolution block under autograd (native/conv.py::_FusedConvBlock): what ATen ran as ~20 small launches
// per layer (activation gradient, three reductions, two broadcasts, the d / s scalings) in three:
//   block_bwd_pre       gc = d[b,o] * g_pre,  g_pre = g * gain * act'(y);  per (b, o): S0 = sum g_pre, S1 = sum g_pre * noise,
//                       S2 = sum g_pre * pre   (pre = the pre-activation, recovered from the block's output y: the leaky-ReLU is
//                       invertible -- sign(y), then y / gain (/ slope))
//   block_bwd_finalize  gd[b,o] = (S2 - nw S1 - bias[o] S0) / d[b,o]   (= sum g_pre * conv_raw),  gbias[o] = sum_b S0,  gnw = sum S1
//   mod_input_bwd       gs[b,i] = sum_p x * gxs,  gx = s[b,i] * gxs   (in place)
struct BlockBwdArgs {
    float* gc; float* sums;          // [B,Cout,H,W], [B*Cout][3]
    const float* g; const float* y; const float* d; const float* noise;
    float slope, gain;
    int act, noise_batched, Cout;
    int64_t HW;
};
__device__ __forceinline__ float block_sum256(float v, float* red)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(256) block_bwd_pre_kernel(BlockBwdArgs a)
{
    __shared__ float red[4];
    const int bo = blockIdx.x, b = bo / a.Cout;
    const float dd = a.d ? a.d[bo] : 1.0f;
    const float* g = a.g + (int64_t)bo * a.HW;
    const float* y = a.y + (int64_t)bo * a.HW;
    const float* nz = a.noise ? a.noise + (a.noise_batched ? (int64_t)b * a.HW : 0) : nullptr;
    float* gc = a.gc + (int64_t)bo * a.HW;
    const float inv_gain = 1.0f / a.gain, inv_slope = 1.0f / a.slope;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t p = 4 * (int64_t)threadIdx.x; p < a.HW; p += 1024) {          // HW % 4 == 0 (the convolution kernels need W % 32 == 0)
        const float4 gv = *reinterpret_cast<const float4*>(g + p), yv = *reinterpret_cast<const float4*>(y + p);
        const float4 nv = nz ? *reinterpret_cast<const float4*>(nz + p) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float ge[4] = {gv.x, gv.y, gv.z, gv.w}, ye[4] = {yv.x, yv.y, yv.z, yv.w}, ne[4] = {nv.x, nv.y, nv.z, nv.w};
        float oe[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gp = ge[k], pre = ye[k];
            if (a.act) {
                const bool pos = ye[k] > 0.f;
                gp = ge[k] * (pos ? a.gain : a.gain * a.slope);
                pre = ye[k] * (pos ? inv_gain : inv_gain * inv_slope);
            }
            s0 += gp; s1 = fmaf(gp, ne[k], s1); s2 = fmaf(gp, pre, s2);
            oe[k] = gp * dd;
        }
        *reinterpret_cast<float4*>(gc + p) = make_float4(oe[0], oe[1], oe[2], oe[3]);
    }
    s0 = block_sum256(s0, red); s1 = block_sum256(s1, red); s2 = block_sum256(s2, red);
    if (threadIdx.x == 0) { a.sums[3 * (int64_t)bo] = s0; a.sums[3 * (int64_t)bo + 1] = s1; a.sums[3 * (int64_t)bo + 2] = s2; }
}
__global__ void __launch_bounds__(256) block_bwd_finalize_kernel(float* __restrict__ gd, float* __restrict__ gbias, float* __restrict__ gnw,
                                                                const float* __restrict__ sums, const float* __restrict__ d,
                                                                const float* __restrict__ bias, const float* __restrict__ noise_weight, int B, int Cout)
{
    __shared__ float red[4];
    const float nw = noise_weight ? *noise_weight : 0.f;
    float tot1 = 0.f;
    for (int o = threadIdx.x; o < Cout; o += 256) {
        float sb = 0.f;
        for (int b = 0; b < B; ++b) {
            const float* s = sums + 3 * ((int64_t)b * Cout + o);
            sb += s[0]; tot1 += s[1];
            if (gd) gd[(int64_t)b * Cout + o] = ((s[2] - nw * s[1]) - (bias ? bias[o] : 0.f) * s[0]) / d[(int64_t)b * Cout + o];
        }
        if (gbias) gbias[o] = sb;
    }
    tot1 = block_sum256(tot1, red);
    if (gnw && threadIdx.x == 0) *gnw = tot1;
}
extern "C" int hav_conv_block_bwd(float* gc, float* gd, float* gbias, float* gnw, float* sums_scratch /*[B*Cout*3]*/, const float* g, const float* y,
                                  const float* d, const float* noise, const float* noise_weight, const float* bias, float slope, float gain, int act,
                                  int noise_batched, int B, int Cout, int64_t HW, void* stream)
{
    if (!gc || !sums_scratch || !g || !y || B < 1 || Cout < 1 || HW < 4 || (gd && !d) || (gnw && !noise)) return HAV_EINVAL;
    if (HW % 4) return HAV_EUNSUP;
    if (act && (!(slope > 0.f) || !(gain > 0.f))) return HAV_EUNSUP;          // the activation must be invertible
    BlockBwdArgs a;
    a.gc = gc; a.sums = sums_scratch; a.g = g; a.y = y; a.d = d; a.noise = noise; a.slope = slope; a.gain = gain; a.act = act;
    a.noise_batched = noise_batched; a.Cout = Cout; a.HW = HW;
    hipLaunchKernelGGL(block_bwd_pre_kernel, dim3((unsigned)(B * Cout)), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    if (gd || gbias || gnw) {
        hipLaunchKernelGGL(block_bwd_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gd, gbias, gnw, (const float*)sums_scratch, d, bias,
                           noise_weight, B, Cout);
        HAV_LAUNCH_CHECK();
    }
    return 0;
}
__global__ void __launch_bounds__(256) mod_input_bwd_kernel(float* __restrict__ gx, float* __restrict__ gs, const float* __restrict__ x,
                                                           const float* __restrict__ s, int64_t HW)
{
    __shared__ float red[4];
    const int bi = blockIdx.x;
    const float sv = s[bi];
    float* gp = gx + (int64_t)bi * HW;
    const float* xp = x + (int64_t)bi * HW;
    float acc = 0.f;
    for (int64_t p = 4 * (int64_t)threadIdx.x; p < HW; p += 1024) {
        float4 gv = *reinterpret_cast<const float4*>(gp + p);
        const float4 xv = *reinterpret_cast<const float4*>(xp + p);
        acc = fmaf(gv.x, xv.x, acc); acc = fmaf(gv.y, xv.y, acc); acc = fmaf(gv.z, xv.z, acc); acc = fmaf(gv.w, xv.w, acc);
        gv.x *= sv; gv.y *= sv; gv.z *= sv; gv.w *= sv;
        *reinterpret_cast<float4*>(gp + p) = gv;
    }
    acc = block_sum256(acc, red);
    if (threadIdx.x == 0) gs[bi] = acc;
}
extern "C" int hav_mod_input_bwd(float* gx_inout, float* gs, const float* x, const float* s, int B, int Cin, int64_t HW, void* stream)
{
    if (!gx_inout || !gs || !x || !s || B < 1 || Cin < 1 || HW < 4) return HAV_EINVAL;
    if (HW % 4) return HAV_EUNSUP;
    hipLaunchKernelGGL(mod_input_bwd_kernel, dim3((unsigned)(B * Cin)), dim3(256), 0, (hipStream_t)stream, gx_inout, gs, x, s, HW);
    HAV_LAUNCH_CHECK();
    return 0;
}
