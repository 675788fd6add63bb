// hav_ops.hip -- fused_bias_act and upfirdn2d for gfx950 (MI355X).
//
// Both ops are pure HBM streaming work (no contraction): they are written for coalesced 16-byte
// per-lane accesses, 64-wide wavefronts and enough resident waves to cover HBM latency; upfirdn2d
// stages its input tile through LDS so every input element is fetched from memory once per tile.
//
// Reference behaviour restated (not translated) from:
//   model/op/fused_bias_act_kernel.cu:18-105, model/op/fused_bias_act.cpp:18-31
//   model/op/upfirdn2d_kernel.cu:49-369,      model/op/upfirdn2d.cpp:17-31
#include "hav_common.h"
#include <atomic>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

// ================================================================================================
// fused_bias_act
// ================================================================================================
template <typename T> struct CompT { typedef float type; };
template <> struct CompT<double> { typedef double type; };

template <typename CT>
__device__ __forceinline__ CT fba_apply(CT x, CT ref, int mode, CT alpha, CT scale)
{
    // mode = act*10+grad  (fused_bias_act_kernel.cu:40-63)
    CT y;
    switch (mode) {
    case 12: case 32: y = (CT)0; break;
    case 30: y = (x > (CT)0) ? x : x * alpha; break;
    case 31: y = (ref > (CT)0) ? x : x * alpha; break;
    default: y = x; break;      // 10, 11 and any unknown act: linear
    }
    return y * scale;
}

template <typename T, int VEC> struct __attribute__((aligned(16))) VecT { T v[VEC]; };

template <typename T, typename CT> __device__ __forceinline__ CT cvt_in(T v)
{
    if constexpr (sizeof(T) == 8) return (CT)v; else return (CT)to_f<T>(v);
}

// One 16-byte vector per lane per iteration; requires step_b % VEC == 0 so a vector never straddles
// two bias channels.  Grid-stride over vectors.
template <typename T, int VEC>
__global__ void __launch_bounds__(256) fba_vec_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                      const T* __restrict__ b, const T* __restrict__ ref, int mode,
                                                      float alpha_f, float scale_f, int64_t nvec, int64_t step_vec,
                                                      int64_t size_b)
{
    typedef typename CompT<T>::type CT;
    typedef VecT<T, VEC> V;
    const CT alpha = (CT)alpha_f, scale = (CT)scale_f;
    const V* xv = reinterpret_cast<const V*>(x);
    const V* rv = reinterpret_cast<const V*>(ref);
    V* ov = reinterpret_cast<V*>(out);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        V a = xv[i];
        V r;
        if (ref) r = rv[i];
        CT bias = (CT)0;
        if (b) bias = cvt_in<T, CT>(b[(i / step_vec) % size_b]);
        V o;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            CT xv_ = cvt_in<T, CT>(a.v[k]) + bias;
            CT rr = ref ? cvt_in<T, CT>(r.v[k]) : (CT)0;
            CT y = fba_apply<CT>(xv_, rr, mode, alpha, scale);
            if constexpr (sizeof(T) == 8) o.v[k] = (T)y; else o.v[k] = from_f<T>((float)y);
        }
        ov[i] = o;
    }
}

// scalar kernel: tails, unaligned pointers, step_b not a multiple of the vector width
template <typename T>
__global__ void __launch_bounds__(256) fba_scalar_kernel(T* __restrict__ out, const T* __restrict__ x,
                                                         const T* __restrict__ b, const T* __restrict__ ref, int mode,
                                                         float alpha_f, float scale_f, int64_t begin, int64_t size_x,
                                                         int64_t step_b, int64_t size_b)
{
    typedef typename CompT<T>::type CT;
    const CT alpha = (CT)alpha_f, scale = (CT)scale_f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < size_x; i += stride) {
        CT v;
        if constexpr (sizeof(T) == 8) v = (CT)x[i]; else v = (CT)to_f<T>(x[i]);
        if (b) {
            T bb = b[(i / step_b) % size_b];
            if constexpr (sizeof(T) == 8) v += (CT)bb; else v += (CT)to_f<T>(bb);
        }
        CT rr = (CT)0;
        if (ref) { if constexpr (sizeof(T) == 8) rr = (CT)ref[i]; else rr = (CT)to_f<T>(ref[i]); }
        CT y = fba_apply<CT>(v, rr, mode, alpha, scale);
        if constexpr (sizeof(T) == 8) out[i] = (T)y; else out[i] = from_f<T>((float)y);
    }
}

template <typename T>
static int fba_launch(void* out_, const void* x_, const void* b_, const void* ref_, int act, int grad, float alpha,
                      float scale, int64_t size_x, int64_t step_b, int64_t size_b, hipStream_t st)
{
    T* out = (T*)out_;
    const T* x = (const T*)x_;
    const T* b = (const T*)b_;
    const T* ref = (const T*)ref_;
    constexpr int VEC = 16 / sizeof(T);
    const int mode = act * 10 + grad;
    if (size_x == 0) return 0;
    const bool aligned = (((uintptr_t)out | (uintptr_t)x | (uintptr_t)(ref ? ref : x)) & 15) == 0;
    int64_t done = 0;
    if (aligned && (!b || step_b % VEC == 0) && size_x >= VEC) {
        const int64_t nvec = size_x / VEC;
        int64_t blocks = (nvec + 255) / 256;
        const int64_t cap = (int64_t)hav_num_cus() * 16;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((fba_vec_kernel<T, VEC>), dim3((unsigned)blocks), dim3(256), 0, st, out, x, b, ref, mode, alpha,
                           scale, nvec, b ? step_b / VEC : (int64_t)1, b ? size_b : (int64_t)1);
        HAV_LAUNCH_CHECK();
        done = nvec * VEC;
    }
    if (done < size_x) {
        const int64_t rem = size_x - done;
        int64_t blocks = (rem + 255) / 256;
        const int64_t cap = (int64_t)hav_num_cus() * 16;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL((fba_scalar_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, out, x, b, ref, mode, alpha,
                           scale, done, size_x, b ? step_b : (int64_t)1, b ? size_b : (int64_t)1);
        HAV_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int hav_fused_bias_act(void* out, const void* x, const void* b, const void* ref, int dtype, int act, int grad,
                                  float alpha, float scale, int64_t size_x, int64_t step_b, int64_t size_b, void* stream)
{
    if (size_x < 0) return HAV_EINVAL;
    if (size_x > 0 && (!out || !x)) return HAV_EINVAL;
    if (b && (step_b <= 0 || size_b <= 0)) return HAV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case HAV_F32: return fba_launch<float>(out, x, b, ref, act, grad, alpha, scale, size_x, step_b, size_b, st);
    case HAV_F16: return fba_launch<__half>(out, x, b, ref, act, grad, alpha, scale, size_x, step_b, size_b, st);
    case HAV_BF16: return fba_launch<__hip_bfloat16>(out, x, b, ref, act, grad, alpha, scale, size_x, step_b, size_b, st);
    case HAV_F64: return fba_launch<double>(out, x, b, ref, act, grad, alpha, scale, size_x, step_b, size_b, st);
    default: return HAV_EUNSUP;
    }
}

// ================================================================================================
// StyleGAN2 block glue (model/styleUnet.py: ModulatedConv2d / NoiseInjection / FusedLeakyReLU).  In the reference these are
// ~10 tiny ATen launches per styled convolution (EqualLinear, bias, square, matmul, +eps, rsqrt, 3 broadcasts, bias-act); at
// ~4.5 us per launch on this GPU they cost as much as the convolution itself at 32^2..64^2.  Two kernels replace them.
// ================================================================================================
// (a) style + demodulation vectors of one modulated convolution:
//     s[b,i] = <style[b,:], mod_w[i,:]> + mod_b[i]                  (EqualLinear with its scale / lr_mul folded into mod_w, mod_b)
//     d[b,o] = rsqrt(sum_i s[b,i]^2 * wsq[i,o] + eps)               (wsq[i,o] = sum_k (scale*W[o,i,k])^2), optional
// Grid = (ceil(Cout/16), B): every workgroup recomputes s (32 K MACs, L2-resident weights) into LDS, then produces 16 outputs with
// the i range split over 64 thread slices (all loads of a thread independent and in flight together: one memory round trip
// instead of Cin/4), reduced through LDS.  Workgroup x = 0 also writes s.
#define SD_OT 16
#define SD_SL 64
// PF / PCF: pointer types of the outputs / read-only tables -- plain pointers for kernel arguments, address_space(1) pointers for those
// read from the batched kernel's device table (generic to the compiler otherwise: FLAT loads and stores)
template <typename PF, typename PCF>
__device__ __forceinline__ void style_demod_body(PF s_out, PF d_out, const float* __restrict__ st, PCF mod_w, PCF mod_b, PCF wsq, float eps, int D,
                                                 int Cin, int Cout, int bx, float* s_sq)
{
    float* s_part = s_sq + Cin;
    const int tid = threadIdx.x;
    // s[i] = <style, mod_w[i,:]> + mod_b[i]: a group of G = pow2 >= D lanes (<= 64) per row, so that a wave reads whole rows of mod_w
    // coalesced (a thread per row walked D floats at a stride of D: 13 us for Cin = 512, D = 32); in-group tree sum, fixed order
    {
        int G = 1;
        while (G < D && G < 64) G <<= 1;
        const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
        const int gl = lane & (G - 1), grp = lane / G, gpw = 64 / G;
        constexpr int UB = 8;              // rows per group in flight: the loads of a batch are all issued before the first reduction
        for (int i0 = wave * gpw; i0 < Cin; i0 += nw * gpw * UB) {
            float acc[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * nw * gpw + grp;
                acc[u] = 0.f;
                if (i < Cin)
                    for (int k = gl; k < D; k += G) acc[u] = fmaf(st[k], mod_w[(size_t)i * D + k], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * nw * gpw + grp;
                float t = acc[u];
                for (int o = G >> 1; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
                if (i < Cin && gl == 0) {
                    const float v = t + (mod_b ? mod_b[i] : 0.f);
                    if (bx == 0) s_out[i] = v;
                    s_sq[i] = v * v;
                }
            }
        }
    }
    if (!d_out) return;
    __syncthreads();
    const int oi = tid & (SD_OT - 1), sl = tid / SD_OT;
    const int o = bx * SD_OT + oi;
    float acc = 0.f;
    if (o < Cout) {
#pragma unroll 8
        for (int i = sl; i < Cin; i += SD_SL) acc = fmaf(s_sq[i], wsq[(size_t)i * Cout + o], acc);
    }
    s_part[sl * SD_OT + oi] = acc;
    __syncthreads();
    if (tid < SD_OT && o < Cout) {
        float t = 0.f;
        for (int q = 0; q < SD_SL; ++q) t += s_part[q * SD_OT + tid];
        d_out[o] = rsqrtf(t + eps);
    }
}

__global__ void __launch_bounds__(SD_OT * SD_SL) style_demod_kernel(float* __restrict__ s_out, float* __restrict__ d_out,
                                                                    const float* __restrict__ style, const float* __restrict__ mod_w,
                                                                    const float* __restrict__ mod_b, const float* __restrict__ wsq,
                                                                    float eps, int D, int Cin, int Cout)
{
    extern __shared__ float s_sq[];                 // [Cin] s^2, then [SD_SL][SD_OT] partial sums
    const int b = blockIdx.y;
    style_demod_body<float*, const float*>(s_out + (size_t)b * Cin, d_out ? d_out + (size_t)b * Cout : nullptr, style + (size_t)b * D, mod_w, mod_b, wsq, eps, D,
                     Cin, Cout, blockIdx.x, s_sq);
}

// All modulated convolutions of a generator in ONE launch: their style vectors only depend on the latent, which is known before the
// first convolution runs, and one style_demod launch is latency-bound (13-20 us for 1 MB of wsq).  The layer table lives in device
// memory; workgroup x belongs to the layer whose [first_block, first_block + blocks) range contains it.
__global__ void __launch_bounds__(SD_OT * SD_SL) style_demod_batched_kernel(const HavStyleDemodLayer* __restrict__ layers, int n_layers,
                                                                            const float* __restrict__ styles, float eps, int n_styles, int D)
{
    extern __shared__ float s_sq[];
    int l = 0;
    for (int q = 1; q < n_layers; ++q)
        if ((int)blockIdx.x >= layers[q].first_block) l = q;
    const HavStyleDemodLayer L = layers[l];
    const int b = blockIdx.y;
    // pointers read from the table are generic to the compiler (FLAT loads / stores); they are device-memory pointers by contract
    typedef __attribute__((address_space(1))) float gfloat;
    typedef const __attribute__((address_space(1))) float cgfloat;
    gfloat* s_out = (gfloat*)L.s_out;
    gfloat* d_out = (gfloat*)L.d_out;
    cgfloat* mod_w = (cgfloat*)L.mod_w;
    cgfloat* mod_b = (cgfloat*)L.mod_b;
    cgfloat* wsq = (cgfloat*)L.wsq;
    style_demod_body<gfloat*, cgfloat*>(s_out + (size_t)b * L.Cin, d_out ? d_out + (size_t)b * L.Cout : (gfloat*)nullptr,
                     styles + ((size_t)b * n_styles + L.style_index) * D, mod_w, mod_b, wsq, eps, D, L.Cin, L.Cout,
                     (int)blockIdx.x - L.first_block, s_sq);
}

// The style network of a generator -- PixelNorm, then n x (EqualLinear + fused leaky-ReLU) on a [B, <= 64] vector (model/styleUnet.py:
// _style_mlp; the tri-plane generators run 32 -> 32 x 4) -- as ONE wave per batch row instead of 5 + 2 n launch-bound ATen / rocBLAS
// launches at the head of every generator's chain (13 launches, ~75 us of the frame's critical path).  lane = output unit; the input
// vector lives one value per lane and is broadcast with v_readlane.  blob: per layer the scaled weight TRANSPOSED [Din][Dout] (coalesced
// over the lanes) followed by the scaled bias [Dout], as the EqualLinear caches hold them (scale * W, lr_mul * b).
__global__ void __launch_bounds__(64) style_mlp_kernel(float* __restrict__ out, const float* __restrict__ z, const float* __restrict__ blob,
                                                       int n_layers, int D0, int D, float slope, float gain)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    float x = lane < D0 ? z[(int64_t)b * D0 + lane] : 0.f;
    float ss = x * x;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    x = x * rsqrtf(ss / (float)D0 + 1e-8f);          // PixelNorm (:53-55)
    const float* w = blob;
    for (int l = 0; l < n_layers; ++l) {
        const int Din = l == 0 ? D0 : D;
        float acc = 0.f;
        const int oc = lane < D ? lane : D - 1;
        // the layer's weight column of this lane in ONE round trip (<= 64 loads in flight + the bias), then the products in index order.  (Round 5
        // read w[k] inside the k loop: 32 dependent L2 round trips per layer -- 55 us for the 32 -> 32 x 4 network at the head of each generator's
        // chain; same sums, same order.)
        float wv[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) wv[k] = k < Din ? w[k * D + oc] : 0.f;
        const float bias = w[Din * D + oc];
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (k < Din) {          // (wave-uniform)
                const float xk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), k));
                acc = fmaf(xk, wv[k], acc);
            }
        }
        const float y = acc + bias;
        x = lane < D ? (y > 0.f ? y : y * slope) * gain : 0.f;          // fused_leaky_relu (fused_bias_act_kernel.cu:40-63)
        w += (Din + 1) * D;
    }
    if (lane < D) out[(int64_t)b * D + lane] = x;
}

extern "C" int hav_style_mlp(float* out, const float* z, const float* blob, int n_layers, int B, int D0, int D, float slope, float gain, void* stream)
{
    if (!out || !z || !blob || n_layers < 1 || B < 1 || D0 < 1 || D < 1) return HAV_EINVAL;
    if (D0 > 64 || D > 64) return HAV_EUNSUP;
    hipLaunchKernelGGL(style_mlp_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, out, z, blob, n_layers, D0, D, slope, gain);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_style_demod(float* s_out, float* d_out, const float* style, const float* mod_w, const float* mod_b,
                               const float* wsq, float eps, int B, int D, int Cin, int Cout, void* stream)
{
    if (!s_out || !style || !mod_w || B < 1 || D < 1 || Cin < 1) return HAV_EINVAL;
    if (d_out && (!wsq || Cout < 1)) return HAV_EINVAL;
    const size_t lds = ((size_t)Cin + SD_OT * SD_SL) * sizeof(float);
    if (lds > 64 * 1024) return HAV_EUNSUP;
    const int gx = d_out ? (Cout + SD_OT - 1) / SD_OT : 1;
    hipLaunchKernelGGL(style_demod_kernel, dim3(gx, B), dim3(SD_OT * SD_SL), lds, (hipStream_t)stream, s_out, d_out, style, mod_w, mod_b,
                       wsq, eps, D, Cin, Cout);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_style_demod_blocks(int Cout, int demodulate) { return demodulate ? (Cout + SD_OT - 1) / SD_OT : 1; }

extern "C" int hav_style_demod_batched(const HavStyleDemodLayer* layers_dev, int n_layers, int total_blocks, int max_cin,
                                       const float* styles, float eps, int B, int n_styles, int D, void* stream)
{
    if (!layers_dev || !styles || n_layers < 1 || total_blocks < 1 || B < 1 || n_styles < 1 || D < 1 || max_cin < 1) return HAV_EINVAL;
    const size_t lds = ((size_t)max_cin + SD_OT * SD_SL) * sizeof(float);
    if (lds > 64 * 1024) return HAV_EUNSUP;
    hipLaunchKernelGGL(style_demod_batched_kernel, dim3(total_blocks, B), dim3(SD_OT * SD_SL), lds, (hipStream_t)stream, layers_dev,
                       n_layers, styles, eps, n_styles, D);
    HAV_LAUNCH_CHECK();
    return 0;
}

// (b) epilogue of a styled convolution, one pass over the activation:
//     y[b,c,p] = lrelu_{0.2}( (x[b,c,p] * d[b,c]  +  nw * noise[b?,p])  +  bias[c] ) * sqrt(2)
// with the reference's operation order and roundings (product, product, sum, sum: no FMA contraction), so the result equals the
// unfused ATen sequence bit for bit.  d / noise / bias are optional.  nw is read from device memory (it is a parameter).
// individually rounded product / sum: opaque to the compiler's FMA contraction (hipcc contracts x*y+z across __fmul_rn/__fadd_rn)
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int VEC>
__global__ void __launch_bounds__(256) styled_epilogue_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                              const float* __restrict__ d, const float* __restrict__ noise,
                                                              const float* __restrict__ nw_ptr, const float* __restrict__ bias,
                                                              float slope, float gain, int64_t total_v, int C, int64_t HW,
                                                              int noise_batched)
{
    const float nw = (noise && nw_ptr) ? *nw_ptr : 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total_v; v += stride) {
        const int64_t e = v * VEC;
        const int64_t plane = e / HW, p = e - plane * HW;          // plane = b*C + c  (HW % VEC == 0: a vector never straddles planes)
        const int c = (int)(plane % C);
        const int64_t b = plane / C;
        float xv[VEC], nv[VEC];
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + e);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
            if (noise) {
                const float4 n4 = *reinterpret_cast<const float4*>(noise + (noise_batched ? b * HW : 0) + p);
                nv[0] = n4.x; nv[1] = n4.y; nv[2] = n4.z; nv[3] = n4.w;
            }
        } else {
            xv[0] = x[e];
            if (noise) nv[0] = noise[(noise_batched ? b * HW : 0) + p];
        }
        const float dd = d ? d[plane] : 1.f, bb = bias ? bias[c] : 0.f;
        float yv[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float t = d ? mul_rn(xv[q], dd) : xv[q];
            if (noise) t = add_rn(t, mul_rn(nw, nv[q]));
            t = add_rn(t, bb);
            yv[q] = mul_rn(t > 0.f ? t : mul_rn(t, slope), gain);
        }
        if (VEC == 4) *reinterpret_cast<float4*>(out + e) = make_float4(yv[0], yv[1], yv[2], yv[3]);
        else out[e] = yv[0];
    }
}

extern "C" int hav_styled_epilogue(float* out, const float* x, const float* d, const float* noise, const float* noise_weight,
                                   const float* bias, float slope, float gain, int B, int C, int64_t HW, int noise_batched,
                                   void* stream)
{
    if (!out || !x || B < 1 || C < 1 || HW < 1) return HAV_EINVAL;
    const int64_t total = (int64_t)B * C * HW;
    const bool vec = (HW % 4 == 0) && (((uintptr_t)out | (uintptr_t)x | (uintptr_t)noise) % 16 == 0);
    const int64_t total_v = vec ? total / 4 : total;
    int64_t blocks = (total_v + 255) / 256;
    const int64_t cap = (int64_t)hav_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (vec) hipLaunchKernelGGL(styled_epilogue_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, x, d, noise,
                                noise_weight, bias, slope, gain, total_v, C, HW, noise_batched);
    else hipLaunchKernelGGL(styled_epilogue_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, x, d, noise,
                            noise_weight, bias, slope, gain, total_v, C, HW, noise_batched);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// Tri-plane gather with gradients (training path).  Same sampling as sample_from_triplane_new
// (utils/util.py:359-392: plane 0 at (x,y), plane 1 at (z,y); F.grid_sample bilinear, zeros padding, align_corners=True) on
// CHANNELS-LAST planes [2,B,H,W,C]: one wave per query, lane = channel, so every tap is one coalesced 4*C-byte row and the
// backward scatter is one coalesced row of float atomics per tap.  (ATen's grid_sampler_2d_backward on the NCHW planes takes
// 18 ms per training step on MI355X: its lanes are C*H*W*4 bytes apart.)   feat[n, 2c+p] layout = the reference's stack/reshape.
// ================================================================================================
__device__ __forceinline__ void tri_taps(float u, float v, int H, int W, int (&idx)[4], float (&w)[4], float& wx0, float& wx1,
                                         float& wy0, float& wy1, bool (&valid)[4])
{
    const float ix = ((u + 1.0f) * 0.5f) * (float)(W - 1), iy = ((v + 1.0f) * 0.5f) * (float)(H - 1);
    float x0f = floorf(ix), y0f = floorf(iy);
    wx1 = ix - x0f; wx0 = 1.0f - wx1; wy1 = iy - y0f; wy0 = 1.0f - wy1;
    x0f = fminf(fmaxf(x0f, -2.f), (float)W + 1.f); y0f = fminf(fmaxf(y0f, -2.f), (float)H + 1.f);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1), cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    idx[0] = cy0 * W + cx0; idx[1] = cy0 * W + cx1; idx[2] = cy1 * W + cx0; idx[3] = cy1 * W + cx1;
    valid[0] = vx0 && vy0; valid[1] = vx1 && vy0; valid[2] = vx0 && vy1; valid[3] = vx1 && vy1;
    w[0] = valid[0] ? wx0 * wy0 : 0.f; w[1] = valid[1] ? wx1 * wy0 : 0.f; w[2] = valid[2] ? wx0 * wy1 : 0.f; w[3] = valid[3] ? wx1 * wy1 : 0.f;
}

__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// MODE 0: forward (feat out).  MODE 1: backward (dplanes += scatter, dq out).
template <int MODE>
__global__ void __launch_bounds__(256) triplane_gather_kernel(float* __restrict__ feat, float* __restrict__ dplanes, float* __restrict__ dq,
                                                              const float* __restrict__ dfeat, const float* __restrict__ planes,
                                                              const float* __restrict__ q, int64_t n, int64_t n_per_b, int B, int H,
                                                              int W, int C)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const size_t plane_sz = (size_t)B * H * W * C;
    for (int64_t i = wave0; i < n; i += nwaves) {
        const int b = (int)(i / n_per_b);
        const float qx = q[i * 3 + 0], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
        float dqx = 0.f, dqy = 0.f, dqz = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int idx[4]; float w[4]; bool valid[4]; float wx0, wx1, wy0, wy1;
            tri_taps(p ? qz : qx, qy, H, W, idx, w, wx0, wx1, wy0, wy1, valid);
            const float* pl = planes + p * plane_sz + (size_t)b * H * W * C;
            for (int c = lane; c < C; c += 64) {
                float t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = valid[k] ? pl[(size_t)idx[k] * C + c] : 0.f;
                if (MODE == 0) {
                    // the reference's sum order: ((nw*v00 + ne*v01) + sw*v10) + se*v11
                    feat[i * (2 * C) + 2 * c + p] = ((t[0] * w[0] + t[1] * w[1]) + t[2] * w[2]) + t[3] * w[3];
                } else {
                    const float g = dfeat[i * (2 * C) + 2 * c + p];
                    float* dpl = dplanes + p * plane_sz + (size_t)b * H * W * C;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (valid[k]) atomicAdd(dpl + (size_t)idx[k] * C + c, w[k] * g);
                    // d/d(ix) = (t01 - t00) wy0 + (t11 - t10) wy1 ;  d/d(iy) = (t10 - t00) wx0 + (t11 - t01) wx1
                    const float gx = g * ((t[1] - t[0]) * wy0 + (t[3] - t[2]) * wy1);
                    const float gy = g * ((t[2] - t[0]) * wx0 + (t[3] - t[1]) * wx1);
                    if (p == 0) dqx += gx; else dqz += gx;
                    dqy += gy;
                }
            }
        }
        if (MODE == 1 && dq) {
            dqx = wave_sum64(dqx); dqy = wave_sum64(dqy); dqz = wave_sum64(dqz);
            if (lane == 0) {
                dq[i * 3 + 0] = dqx * (0.5f * (float)(W - 1));
                dq[i * 3 + 1] = dqy * (0.5f * (float)(H - 1));
                dq[i * 3 + 2] = dqz * (0.5f * (float)(W - 1));
            }
        }
    }
}

extern "C" int hav_triplane_gather_fwd(float* feat, const float* planes_cl, const float* q, int64_t n, int64_t n_per_b, int B, int H,
                                       int W, int C, void* stream)
{
    if (!feat || !planes_cl || !q || n < 0 || n_per_b < 1 || B < 1 || H < 2 || W < 2 || C < 1) return HAV_EINVAL;
    if (n == 0) return 0;
    int64_t blocks = (n + 3) / 4;
    const int64_t cap = (int64_t)hav_num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(triplane_gather_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, feat, nullptr, nullptr, nullptr,
                       planes_cl, q, n, n_per_b, B, H, W, C);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_triplane_gather_bwd(float* dplanes_cl, float* dq, const float* dfeat, const float* planes_cl, const float* q, int64_t n,
                                       int64_t n_per_b, int B, int H, int W, int C, void* stream)
{
    if (!dplanes_cl || !dfeat || !planes_cl || !q || n < 0 || n_per_b < 1 || B < 1 || H < 2 || W < 2 || C < 1) return HAV_EINVAL;
    if (n == 0) return 0;
    int64_t blocks = (n + 3) / 4;
    const int64_t cap = (int64_t)hav_num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(triplane_gather_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, nullptr, dplanes_cl, dq, dfeat,
                       planes_cl, q, n, n_per_b, B, H, W, C);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// upfirdn2d
// ================================================================================================
struct UfdArgs {
    int64_t major;
    int in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, px0, py0, out_h, out_w;
};

__device__ __forceinline__ float fma_ct(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_ct(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ int floor_div(int a, int b) { int q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }
__device__ __forceinline__ int ceil_div(int a, int b) { return -floor_div(-a, b); }

template <typename T, typename CT> __device__ __forceinline__ CT ld_as(const T* p)
{
    if constexpr (sizeof(T) == 8) return (CT)*p; else return (CT)to_f<T>(*p);
}
template <typename T, typename CT> __device__ __forceinline__ void st_as(T* p, CT v)
{
    if constexpr (sizeof(T) == 8) *p = (T)v; else *p = from_f<T>((float)v);
}

// Generic kernel: any up/down/pad/FIR size and minor; one output element per thread, taps walked
// directly (only the taps that land on a real input sample are visited).
template <typename T>
__global__ void __launch_bounds__(256) ufd_generic_kernel(T* __restrict__ out, const T* __restrict__ in,
                                                          const float* __restrict__ k, UfdArgs a)
{
    typedef typename CompT<T>::type CT;
    const int64_t total = a.major * a.out_h * a.out_w * a.minor;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int mi = (int)(idx % a.minor);
        int64_t t = idx / a.minor;
        int ox = (int)(t % a.out_w);
        t /= a.out_w;
        int oy = (int)(t % a.out_h);
        int64_t m = t / a.out_h;
        // zero-stuffed coordinate of tap (i,j): Y = oy*down_y + i - py0, must be a multiple of up_y
        const int Y0 = oy * a.down_y - a.py0, X0 = ox * a.down_x - a.px0;
        int iy_lo = ceil_div(Y0, a.up_y);
        if (iy_lo < 0) iy_lo = 0;
        int iy_hi = floor_div(Y0 + a.kh - 1, a.up_y);
        if (iy_hi > a.in_h - 1) iy_hi = a.in_h - 1;
        int ix_lo = ceil_div(X0, a.up_x);
        if (ix_lo < 0) ix_lo = 0;
        int ix_hi = floor_div(X0 + a.kw - 1, a.up_x);
        if (ix_hi > a.in_w - 1) ix_hi = a.in_w - 1;
        CT v = (CT)0;
        for (int iy = iy_lo; iy <= iy_hi; ++iy) {
            const int i = iy * a.up_y - Y0;          // tap row; FIR is applied flipped (upfirdn2d_kernel.cu:137)
            const T* row = in + ((m * a.in_h + iy) * a.in_w) * a.minor + mi;
            const float* krow = k + (a.kh - 1 - i) * a.kw;
            for (int ix = ix_lo; ix <= ix_hi; ++ix) {
                const int j = ix * a.up_x - X0;
                v = fma_ct(ld_as<T, CT>(row + (int64_t)ix * a.minor), (CT)krow[a.kw - 1 - j], v);
            }
        }
        st_as<T, CT>(out + idx, v);
    }
}

// LDS-tiled kernel for minor == 1 and compile-time (UP, DOWN, KH, KW), square up/down factors.
// A 256-thread workgroup (4 wave64) produces a 16 x 64 output tile of one [in_h,in_w] plane: the
// input footprint is staged once in LDS (coalesced rows), each wave then writes 64-wide row
// segments (256 B per wave-store for f32).
template <int UP, int DOWN, int KH, int KW> struct UfdTile {
    static constexpr int TH = (DOWN == 1) ? 32 : 16, TW = 64, SR = TH / 4;   // SR output rows per thread
    static constexpr int IH = ((TH - 1) * DOWN + KH - 1) / UP + 2;
    static constexpr int IW = ((TW - 1) * DOWN + KW - 1) / UP + 2;
    static constexpr int IWP = IW | 1;    // odd leading dimension: column walks spread over LDS banks
};

template <typename T, int UP, int DOWN, int KH, int KW>
__global__ void __launch_bounds__(256) ufd_tiled_kernel(T* __restrict__ out, const T* __restrict__ in,
                                                        const float* __restrict__ k, UfdArgs a, int tiles_x, int tiles_y)
{
    typedef typename CompT<T>::type CT;
    typedef UfdTile<UP, DOWN, KH, KW> TL;
    __shared__ CT s_in[TL::IH * TL::IWP];
    __shared__ CT s_k[KH * KW];

    // workgroup b runs on XCD b % 8, each XCD has its own L2: every XCD gets a CONTIGUOUS eighth of the tile list, so that the lines two
    // neighbouring tiles share (halo rows; the partial 128-byte lines at the ends of 513-float rows) come from the L2 that already
    // holds them instead of from HBM a second time (profiles/r03_v5_pmc_ops.txt: 1.37x the input bytes fetched for the decimating
    // 4x4 case when tiles are dealt round-robin).  The grid is padded to a multiple of 8; the surplus workgroups leave at once.
    int64_t bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (bid >= (int64_t)tiles_x * tiles_y * a.major) return;
    const int tx = (int)(bid % tiles_x);
    bid /= tiles_x;
    const int ty = (int)(bid % tiles_y);
    const int64_t m = bid / tiles_y;
    const int oy0 = ty * TL::TH, ox0 = tx * TL::TW;
    const int tid = threadIdx.x;

    // runtime FIR may be smaller than the compiled one: zero-extend at the high index side, which
    // after the flip keeps tap (i,j) aligned exactly as in the direct definition
    // (same trick as the reference's mode 6, upfirdn2d_kernel.cu:136,353-357).
    if (tid < KH * KW) {
        const int i = tid / KW, j = tid % KW;
        const int ki = a.kh - 1 - i, kj = a.kw - 1 - j;
        s_k[tid] = (i < a.kh && j < a.kw) ? (CT)k[ki * a.kw + kj] : (CT)0;
    }
    const int Y0 = oy0 * DOWN - a.py0, X0 = ox0 * DOWN - a.px0;
    const int iy_min = floor_div(Y0, UP), ix_min = floor_div(X0, UP);
    const T* plane = in + m * (int64_t)a.in_h * a.in_w;
    // staging: one tile row per wave per step, lane = column (no index division, row test is wave-uniform)
    const int lx = tid & 63, sy = tid >> 6;
    // every global load of the thread is issued before the first LDS write (the staging is latency-bound otherwise:
    // one round trip per tile row instead of one per tile)
    constexpr int RPW = (TL::IH + 3) / 4, NCH = (TL::IW + 63) / 64;
    CT stg[RPW][NCH];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = sy + 4 * q, iy = iy_min + r;
        const bool row_ok = r < TL::IH && iy >= 0 && iy < a.in_h;
        const T* src = plane + (int64_t)iy * a.in_w + ix_min;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            const int c = h * 64 + lx, ix = ix_min + c;
            stg[q][h] = (CT)0;
            if (row_ok && c < TL::IW && ix >= 0 && ix < a.in_w) stg[q][h] = ld_as<T, CT>(src + c);
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = sy + 4 * q;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            const int c = h * 64 + lx;
            if (r < TL::IH && c < TL::IW) s_in[r * TL::IWP + c] = stg[q][h];
        }
    }
    __syncthreads();

    // Each thread produces a vertical strip of SR outputs (rows SR*sy .. SR*sy+SR-1 of column lx): a staged input value is
    // read from LDS once and feeds every output row it contributes to, the FIR lives in registers, lanes walk consecutive
    // columns (conflict-free ds_read_b32, 256-B row-segment stores). Per output the tap order stays (i asc, j asc).
    constexpr int SR = TL::SR;
    const int ox = ox0 + lx;
    const int Xb = lx * DOWN + X0 - ix_min * UP;        // zero-stuffed x of tap j=0, relative to the tile origin (>= 0)
    const int Yb = (SR * sy) * DOWN + Y0 - iy_min * UP;  // zero-stuffed y of tap i=0 of this strip's first row
    CT kreg[KH * KW];
#pragma unroll
    for (int q = 0; q < KH * KW; ++q) kreg[q] = s_k[q];
    CT acc[SR];
#pragma unroll
    for (int q = 0; q < SR; ++q) acc[q] = (CT)0;
    constexpr int NR = (SR - 1) * DOWN + KH;             // zero-stuffed rows touched by the strip
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int Y = Yb + rr;
        if (UP > 1 && (Y % UP) != 0) continue;
        const CT* row = s_in + (Y / UP) * TL::IWP;
#pragma unroll
        for (int j = 0; j < KW; ++j) {
            const int X = Xb + j;
            if (UP > 1 && (X % UP) != 0) continue;
            const CT v = row[X / UP];
#pragma unroll
            for (int q = 0; q < SR; ++q) {
                const int i = rr - q * DOWN;            // tap row of output q that sees zero-stuffed row rr
                if (i >= 0 && i < KH) acc[q] = fma_ct(v, kreg[i * KW + j], acc[q]);
            }
        }
    }
    T* oplane = out + m * (int64_t)a.out_h * a.out_w;
#pragma unroll
    for (int q = 0; q < SR; ++q) {
        const int oy = oy0 + SR * sy + q;
        if (oy < a.out_h && ox < a.out_w) st_as<T, CT>(oplane + (int64_t)oy * a.out_w + ox, acc[q]);
    }
}

template <typename T, int UP, int DOWN, int KH, int KW>
static int ufd_launch_tiled(T* out, const T* in, const float* k, const UfdArgs& a, hipStream_t st)
{
    typedef UfdTile<UP, DOWN, KH, KW> TL;
    const int tiles_x = (a.out_w + TL::TW - 1) / TL::TW, tiles_y = (a.out_h + TL::TH - 1) / TL::TH;
    const int64_t tiles = a.major * tiles_x * tiles_y;
    const int64_t blocks = (tiles + 7) / 8 * 8;          // XCD-contiguous tile list (see the kernel)
    if (tiles <= 0 || blocks > 0x7fffffffLL) return HAV_EUNSUP;
    hipLaunchKernelGGL((ufd_tiled_kernel<T, UP, DOWN, KH, KW>), dim3((unsigned)blocks), dim3(256), 0, st, out, in, k, a,
                       tiles_x, tiles_y);
    HAV_LAUNCH_CHECK();
    return 0;
}


// Direct (LDS-free) f32 kernel for up == 1 and minor == 1: a thread produces a strip of 4 consecutive output columns x SR output rows
// of one plane.  Its input footprint -- NR rows x (3*DOWN + KW) columns -- is fetched with dword-aligned 16-byte global loads, ALL
// issued before the first FMA (one round trip per strip); neighbouring lanes' footprints overlap, which the L1/L2 absorb, so HBM sees
// every input byte about once; outputs leave as one 16-byte store per row.  No workgroup barrier, no LDS round trip, no half-empty
// staging chunk on the odd 513-wide rows (the tiled kernel's second 64-column chunk had 3 live lanes), and the work is dealt out
// as a flat list of strips, so only the last wave of the launch has idle lanes.  Strips that touch the plane's border (or whose
// 16-byte loads would run past the end of a row) take a per-element path with zero fill.  Per output the taps are accumulated in
// the same order as the tiled and generic kernels (i ascending, then j): results are bit-identical to theirs.
template <int DOWN, int KH, int KW, int SR>
__global__ void __launch_bounds__(256) ufd_direct_f32_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ k,
                                                             UfdArgs a, int strips_x, int blocks_y, int64_t total)
{
    constexpr int NR = (SR - 1) * DOWN + KH, NCOL = 3 * DOWN + KW, NL = (NCOL + 3) / 4;
    // workgroup b runs on XCD b % 8, and each XCD has its own L2: hand every XCD a CONTIGUOUS eighth of the strip list, so that the
    // KH - DOWN input rows two vertically adjacent row blocks share are found in the L2 that fetched them (dealt round-robin, the
    // neighbour sits on another XCD and the shared rows come from HBM a second time: +37 % input traffic for the 4x4 blur)
    int64_t lb = blockIdx.x;
    if ((gridDim.x & 7) == 0) lb = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int64_t gid = lb * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int sx = (int)(gid % strips_x);
    int64_t t = gid / strips_x;
    const int rb = (int)(t % blocks_y);
    const int64_t m = t / blocks_y;
    const int ox0 = 4 * sx, oy0 = rb * SR;
    const int ix0 = ox0 * DOWN - a.px0, iy0 = oy0 * DOWN - a.py0;
    float kreg[KH * KW];          // flipped FIR, zero-extended to the compiled size (upfirdn2d_kernel.cu:136-137)
#pragma unroll
    for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int j = 0; j < KW; ++j) kreg[i * KW + j] = (i < a.kh && j < a.kw) ? k[(a.kh - 1 - i) * a.kw + (a.kw - 1 - j)] : 0.f;
    const float* plane = in + m * (int64_t)a.in_h * a.in_w;
    float v[NR][4 * NL];
    const bool interior = iy0 >= 0 && iy0 + NR <= a.in_h && ix0 >= 0 && ix0 + 4 * NL <= a.in_w;
    if (interior) {
        const float* src = plane + (int64_t)iy0 * a.in_w + ix0;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                const f4u q = *reinterpret_cast<const f4u*>(src + (int64_t)r * a.in_w + 4 * l);
                v[r][4 * l] = q.x; v[r][4 * l + 1] = q.y; v[r][4 * l + 2] = q.z; v[r][4 * l + 3] = q.w;
            }
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int iy = iy0 + r;
            const bool rok = iy >= 0 && iy < a.in_h;
#pragma unroll
            for (int c = 0; c < 4 * NL; ++c) {
                const int ix = ix0 + c;
                v[r][c] = (c < NCOL && rok && ix >= 0 && ix < a.in_w) ? plane[(int64_t)iy * a.in_w + ix] : 0.f;
            }
        }
    }
    float acc[SR][4];
#pragma unroll
    for (int q = 0; q < SR; ++q) acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int j = 0; j < KW; ++j)
#pragma unroll
            for (int q = 0; q < SR; ++q) {
                const int i = rr - q * DOWN;                    // tap row of output row q that sees input row rr
                if (i >= 0 && i < KH) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(v[rr][c * DOWN + j], kreg[i * KW + j], acc[q][c]);
                }
            }
    float* oplane = out + m * (int64_t)a.out_h * a.out_w;
#pragma unroll
    for (int q = 0; q < SR; ++q) {
        const int oy = oy0 + q;
        if (oy >= a.out_h) break;
        float* dst = oplane + (int64_t)oy * a.out_w + ox0;
        if (ox0 + 3 < a.out_w) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const f4u o = {acc[q][0], acc[q][1], acc[q][2], acc[q][3]};
            *reinterpret_cast<f4u*>(dst) = o;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (ox0 + c < a.out_w) dst[c] = acc[q][c];
        }
    }
}

template <int DOWN, int KH, int KW, int SR>
static int ufd_launch_direct(float* out, const float* in, const float* k, const UfdArgs& a, hipStream_t st)
{
    const int strips_x = (a.out_w + 3) / 4, blocks_y = (a.out_h + SR - 1) / SR;
    const int64_t total = a.major * strips_x * blocks_y;
    const int64_t blocks = ((total + 255) / 256 + 7) / 8 * 8;          // a multiple of 8: the XCD-contiguous mapping needs it
    if (blocks <= 0 || blocks > 0x7fffffffLL) return HAV_EUNSUP;
    hipLaunchKernelGGL((ufd_direct_f32_kernel<DOWN, KH, KW, SR>), dim3((unsigned)blocks), dim3(256), 0, st, out, in, k, a, strips_x, blocks_y, total);
    HAV_LAUNCH_CHECK();
    return 0;
}

// Decimating FIR (up = 1, down = 2, FIR <= 4x4, minor = 1: `Downsample` / the Blur-free half of the discriminator and of FromRGB,
// model/styleUnet.py:78-88) without LDS: a thread owns 4 consecutive output columns x SR output rows.  Its footprint is 10 input columns
// per row, 8 of which are its own -- two 16-byte loads, disjoint between lanes, so a wave reads each input row segment exactly once --
// and the last two are the first two of the next lane's segment, fetched from that lane's registers (ds_bpermute) instead of from memory
// (the 3-loads-per-row version of this kernel was slower than the tiled one).  Lanes without a usable neighbour (last lane of a wave,
// last strip of a row) load their two halo values themselves.  Out-of-image columns are masked after the load (the vector may run over
// the end of a row into the next one: still inside the tensor), rows outside the image are not loaded at all; only windows that would
// leave the tensor take the per-element path.  Tap order per output as everywhere (i ascending, then j): bit-identical results.
template <int KH, int KW, int SR>
__global__ void __launch_bounds__(256) ufd_down2_direct_f32_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ k,
                                                                   UfdArgs a, int strips_x, int blocks_y, int64_t total, int64_t in_total)
{
    constexpr int NR = (SR - 1) * 2 + KH;
    int64_t lb = blockIdx.x;
    if ((gridDim.x & 7) == 0) lb = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);          // XCD-contiguous strip list
    const int64_t gid0 = lb * blockDim.x + threadIdx.x;
    const bool live = gid0 < total;
    const int64_t gid = live ? gid0 : total - 1;          // idle lanes of the last wave shadow the last strip (they take part in the shuffles)
    const int lane = threadIdx.x & 63;
    const int sx = (int)(gid % strips_x);
    int64_t t = gid / strips_x;
    const int rb = (int)(t % blocks_y);
    const int64_t m = t / blocks_y;
    const int ox0 = 4 * sx, oy0 = rb * SR;
    const int ix0 = 2 * ox0 - a.px0, iy0 = 2 * oy0 - a.py0;
    float kreg[KH * KW];
#pragma unroll
    for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int j = 0; j < KW; ++j) kreg[i * KW + j] = (i < a.kh && j < a.kw) ? k[(a.kh - 1 - i) * a.kw + (a.kw - 1 - j)] : 0.f;
    const int64_t pbase = m * (int64_t)a.in_h * a.in_w;
    // the halo comes from lane + 1 when that lane holds the next strip of the same row block
    const bool nb_ok = lane < 63 && sx + 1 < strips_x && gid0 + 1 < total;
    // 16-byte loads stay inside the tensor for every image row of the window?
    const int ry0 = iy0 < 0 ? 0 : iy0, ry1 = iy0 + NR - 1 >= a.in_h ? a.in_h - 1 : iy0 + NR - 1;
    const bool vec_ok = pbase + (int64_t)ry0 * a.in_w + ix0 >= 0 && pbase + (int64_t)ry1 * a.in_w + ix0 + 8 <= in_total;
    bool cok[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) cok[c] = ix0 + c >= 0 && ix0 + c < a.in_w;
    float v[NR][10], own[NR][2];
    // every load of the thread -- the two vectors per row and, for lanes without a neighbour, their own halo -- is issued before the first use
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int iy = iy0 + r;
        const bool rok = iy >= 0 && iy < a.in_h;
        const float* src = in + pbase + (int64_t)iy * a.in_w + ix0;
        if (rok && vec_ok) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const f4u q0 = *reinterpret_cast<const f4u*>(src), q1 = *reinterpret_cast<const f4u*>(src + 4);
            v[r][0] = q0.x; v[r][1] = q0.y; v[r][2] = q0.z; v[r][3] = q0.w; v[r][4] = q1.x; v[r][5] = q1.y; v[r][6] = q1.z; v[r][7] = q1.w;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[r][c] = (rok && cok[c]) ? src[c] : 0.f;
        }
        own[r][0] = (!nb_ok && rok && cok[8]) ? src[8] : 0.f;
        own[r][1] = (!nb_ok && rok && cok[9]) ? src[9] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[r][c] = cok[c] ? v[r][c] : 0.f;
        const float h0 = __shfl_down(v[r][0], 1, 64), h1 = __shfl_down(v[r][1], 1, 64);          // already masked by their owner
        v[r][8] = nb_ok ? h0 : own[r][0];
        v[r][9] = nb_ok ? h1 : own[r][1];
    }
    float acc[SR][4];
#pragma unroll
    for (int q = 0; q < SR; ++q) acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int j = 0; j < KW; ++j)
#pragma unroll
            for (int q = 0; q < SR; ++q) {
                const int i = rr - q * 2;
                if (i >= 0 && i < KH) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[q][c] = fmaf(v[rr][c * 2 + j], kreg[i * KW + j], acc[q][c]);
                }
            }
    if (!live) return;
    float* oplane = out + m * (int64_t)a.out_h * a.out_w;
#pragma unroll
    for (int q = 0; q < SR; ++q) {
        const int oy = oy0 + q;
        if (oy >= a.out_h) break;
        float* dst = oplane + (int64_t)oy * a.out_w + ox0;
        if (ox0 + 3 < a.out_w) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const f4u o = {acc[q][0], acc[q][1], acc[q][2], acc[q][3]};
            *reinterpret_cast<f4u*>(dst) = o;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (ox0 + c < a.out_w) dst[c] = acc[q][c];
        }
    }
}

template <int KH, int KW, int SR>
static int ufd_launch_down2_direct(float* out, const float* in, const float* k, const UfdArgs& a, hipStream_t st)
{
    const int strips_x = (a.out_w + 3) / 4, blocks_y = (a.out_h + SR - 1) / SR;
    const int64_t total = a.major * strips_x * blocks_y;
    const int64_t blocks = ((total + 255) / 256 + 7) / 8 * 8;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return HAV_EUNSUP;
    hipLaunchKernelGGL((ufd_down2_direct_f32_kernel<KH, KW, SR>), dim3((unsigned)blocks), dim3(256), 0, st, out, in, k, a, strips_x, blocks_y, total,
                       a.major * (int64_t)a.in_h * a.in_w);
    HAV_LAUNCH_CHECK();
    return 0;
}

// x2 up-sampling FIR (up = 2, down = 1, FIR <= 4x4: `Upsample` of the ToRGB skip path, model/styleUnet.py:65-75) without LDS: a thread owns
// an 8 x 8 output block = the 2x2-tap polyphase filters applied to a 6 x 6 input window held in registers (12 loads, 16 16-byte stores).
// EY / EX = parity of the padding: with output rows Y0 = 8 rb the zero-stuffed coordinate of tap i of row Y0 + q is 2 a0 + EY + q + i,
// so which taps land on real samples -- and on which row of the window -- is known at compile time.  Same tap order (i asc, j asc over
// the real taps, FMA chain from 0) as the tiled kernel: bit-identical results.
template <int EY, int EX>
__global__ void __launch_bounds__(256) ufd_up2_direct_f32_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ k,
                                                                 UfdArgs a, int strips_x, int blocks_y, int64_t total)
{
    int64_t lb = blockIdx.x;
    if ((gridDim.x & 7) == 0) lb = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);          // XCD-contiguous strip list
    const int64_t gid = lb * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int sx = (int)(gid % strips_x);
    int64_t t = gid / strips_x;
    const int rb = (int)(t % blocks_y);
    const int64_t m = t / blocks_y;
    const int ox0 = 8 * sx, oy0 = 8 * rb;
    const int a0f = (oy0 - a.py0 - EY) / 2, b0f = (ox0 - a.px0 - EX) / 2;          // exact divisions (even numerators), negative at the top / left edge
    float kreg[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) kreg[i * 4 + j] = (i < a.kh && j < a.kw) ? k[(a.kh - 1 - i) * a.kw + (a.kw - 1 - j)] : 0.f;
    const float* plane = in + m * (int64_t)a.in_h * a.in_w;
    float v[6][6];
    const bool interior = a0f >= 0 && a0f + 6 <= a.in_h && b0f >= 0 && b0f + 6 <= a.in_w;
    if (interior) {
        const float* src = plane + (int64_t)a0f * a.in_w + b0f;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
            const f4u q = *reinterpret_cast<const f4u*>(src + (int64_t)r * a.in_w);
            const f2u w = *reinterpret_cast<const f2u*>(src + (int64_t)r * a.in_w + 4);
            v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z; v[r][3] = q.w; v[r][4] = w.x; v[r][5] = w.y;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = a0f + r;
            const bool rok = iy >= 0 && iy < a.in_h;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int ix = b0f + c;
                v[r][c] = (rok && ix >= 0 && ix < a.in_w) ? plane[(int64_t)iy * a.in_w + ix] : 0.f;
            }
        }
    }
    float* oplane = out + m * (int64_t)a.out_h * a.out_w;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (((q + i + EY) & 1) != 0) continue;          // tap row i of output row q sits between two samples
            const int r = (q + i + EY) >> 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (((c + j + EX) & 1) != 0) continue;
                    acc[c] = fmaf(v[r][(c + j + EX) >> 1], kreg[i * 4 + j], acc[c]);
                }
        }
        const int oy = oy0 + q;
        if (oy >= a.out_h) break;
        float* dst = oplane + (int64_t)oy * a.out_w + ox0;
        if (ox0 + 7 < a.out_w) {
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
            const f4u o0 = {acc[0], acc[1], acc[2], acc[3]}, o1 = {acc[4], acc[5], acc[6], acc[7]};
            *reinterpret_cast<f4u*>(dst) = o0;
            *reinterpret_cast<f4u*>(dst + 4) = o1;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) if (ox0 + c < a.out_w) dst[c] = acc[c];
        }
    }
}
static int ufd_launch_up2_direct(float* out, const float* in, const float* k, const UfdArgs& a, hipStream_t st)
{
    const int strips_x = (a.out_w + 7) / 8, blocks_y = (a.out_h + 7) / 8;
    const int64_t total = a.major * strips_x * blocks_y;
    const int64_t blocks = ((total + 255) / 256 + 7) / 8 * 8;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return HAV_EUNSUP;
    const int ey = a.py0 & 1, ex = a.px0 & 1;
#define UFD_UP2(EY_, EX_) hipLaunchKernelGGL((ufd_up2_direct_f32_kernel<EY_, EX_>), dim3((unsigned)blocks), dim3(256), 0, st, out, in, k, a, strips_x, blocks_y, total)
    if (ey == 0 && ex == 0) UFD_UP2(0, 0);
    else if (ey == 0) UFD_UP2(0, 1);
    else if (ex == 0) UFD_UP2(1, 0);
    else UFD_UP2(1, 1);
#undef UFD_UP2
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Row-streaming f32 kernels (minor == 1): blur, decimation by 2 and x2 up-sampling with a FIR of up to 4x4 taps as ROLLING WINDOWS.
// A lane owns a strip of 4 output columns of one plane and the input columns its windows start in; a wave takes 64 neighbouring entries
// of the flat (plane, strip) list -- its stores are whole KiB-long runs of a row -- and walks a segment of rows downwards: per input row
// ONE 16-byte buffer load per owned vector (the columns shared with the right neighbour come out of that lane's registers through a DPP
// wave shift; lane 63 and the last strip of a row fetch them with one more load in which every other lane is out of range), the row joins
// a window of <= 4 rows in registers, every completed output row leaves as one 16-byte buffer store per lane.
// * No branch around a memory instruction: rows outside the image are clamped and masked, columns are masked, loads and stores that must
//   not happen get an offset beyond the buffer descriptor's range (the hardware drops them), the end of the tensor is covered by the same
//   range check.  The compiler therefore counts the memory instructions in flight exactly (s_waitcnt vmcnt(2 D - 1)) and a wave computes
//   row r while rows r + 1 .. r + D are on their way.
// * Whole-line stores: with 63 producing lanes per wave (an earlier version) every store began and ended inside a 128-byte line that
//   another wave completed later: 31 us instead of 28 for the 513^2 -> 512^2 blur.
// * Segment borders are shifted per plane: all waves start together and walk at the same pace, with borders at the same rows in every
//   plane they touch addresses that are equal modulo (rows per segment x pitch) at every step -- the same few channels (34 us).
// * The launch is a flat list of single-wave workgroups, (strip group, segment) with the segment running fastest, dealt to the XCDs in
//   contiguous eighths; rows per segment (33 / 17 / 9 / 5) = the longest that still gives the chip >= 9 waves per compute unit.
// Tap order per output as in every other kernel here (i ascending, then j, one FMA chain from 0): bit-identical results.
// Measured (tools/ufd_roll_ab.py, every launch on its own buffers): [64,513,513] -> 512^2 blur 34.2 -> 28.4 us (4.7 TB/s), 3x3 blur
// 32.4 -> 28.0, x2 up-sampling [12,512,512] 18.8 -> 17.4; the decimating class stays on the LDS-tiled kernel (19.3 vs 20.3 us).
typedef unsigned int ufd_u4 __attribute__((ext_vector_type(4)));
#define UFD_OOB ((int)0x80000000)          // beyond every descriptor built here (tensors of < 2 GiB)

__device__ __forceinline__ float ufd_lane_next(float x)          // the value lane + 1 holds (lane 63: unspecified)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

struct UfdRollGeom {
    int sxp, nseg;                    // strips per plane row, segments per plane
    int64_t n_strips, n_waves;
    unsigned in_bytes, out_bytes;
};

__device__ __forceinline__ int64_t ufd_roll_wave_index()
{
    // single-wave workgroups; workgroup b runs on XCD b % 8: every XCD gets a contiguous eighth of the wave list (grid is a multiple of 8)
    return (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
}

// up == 1, DOWN in {1, 2}, FIR <= KH x KW with KW > DOWN, 0 <= pad_x0 < in_w.  SEG output rows per segment, D rows in flight (a multiple of 4).
template <int DOWN, int KH, int KW, int SEG, int D>
__global__ void __launch_bounds__(64) ufd_roll_f32_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ k, UfdArgs a,
                                                          UfdRollGeom g)
{
    constexpr int NV = DOWN, OWN = 4 * DOWN, NCOL = 3 * DOWN + KW, HALO = NCOL - OWN;
    constexpr int NRT = (SEG - 1) * DOWN + KH;          // input rows a segment walks
    constexpr int NIT = (NRT + D - 1) / D;
    static_assert(HALO >= 1 && HALO <= 4 && KH <= 4 && (D & 3) == 0 && D >= 4, "window shape");
    const int64_t wv = ufd_roll_wave_index();
    if (wv >= g.n_waves) return;
    const int lane = threadIdx.x;
    const int sg = (int)(wv % g.nseg);
    const int64_t f0 = (wv / g.nseg) * 64 + lane;
    const bool live = f0 < g.n_strips;
    const int64_t f = f0 < g.n_strips ? f0 : g.n_strips - 1;          // idle lanes shadow the last entry
    const int t = (int)(f % g.sxp);
    const int64_t m = f / g.sxp;
    const int ox0 = 4 * t, ix0 = ox0 * DOWN - a.px0;                   // first output column, first owned input column
    // Segment borders are shifted by a per-group number of rows.  Every wave starts at the same moment and walks at the same pace: with
    // borders at multiples of SEG rows in every plane all of them would touch addresses that are equal modulo SEG x pitch at every
    // step, i.e. the same few memory channels (measured: the launch then runs at 4.0 TB/s whatever SEG, D or the FIR size).
    const int oy_s = sg * SEG - (int)(((((wv / g.nseg) * 64 + 31) / g.sxp) * 11) % SEG), iy_s = oy_s * DOWN - a.py0;
    if (oy_s >= a.out_h || oy_s + SEG <= 0) return;
    const int r_first = oy_s < 0 ? -oy_s * DOWN : 0;          // input rows before this one only feed output rows above the image
    float kreg[KH * KW];          // flipped FIR, zero-extended to the compiled size (upfirdn2d_kernel.cu:136-137)
#pragma unroll
    for (int i = 0; i < KH; ++i)
#pragma unroll
        for (int j = 0; j < KW; ++j) kreg[i * KW + j] = (i < a.kh && j < a.kw) ? k[(a.kh - 1 - i) * a.kw + (a.kw - 1 - j)] : 0.f;
    bool cok[OWN];
#pragma unroll
    for (int c = 0; c < OWN; ++c) cok[c] = ix0 + c >= 0 && ix0 + c < a.in_w;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, g.out_bytes, 0x00020000);
    // A vector that starts left of column 0 reads the tail of the previous row (masked) -- except in row 0 of plane 0, where it would start
    // before the tensor: that one vector is fetched element-wise here and swapped in when the row comes up.
    const bool fixl = m == 0 && ix0 < 0;
    float fix[OWN];
#pragma unroll
    for (int c = 0; c < OWN; ++c) fix[c] = 0.f;
    if (fixl) {
#pragma unroll
        for (int c = 0; c < OWN; ++c) if (cok[c]) fix[c] = in[ix0 + c];
    }
    const int voff0 = (int)((m * (int64_t)a.in_h * a.in_w + ix0) * 4);          // row 0 of the lane's plane, first owned column
    const int row_bytes = a.in_w * 4, orow_bytes = a.out_w * 4;
    const int soff0 = (int)((m * (int64_t)a.out_h * a.out_w + ox0) * 4);
    const bool st_vec = live && ox0 + 3 < a.out_w;
    const int ragged = a.out_w & 3;                                             // the last strip of a row holds this many columns (0: a full one)
    const bool st_rag = live && ragged != 0 && ox0 + 3 >= a.out_w && ox0 < a.out_w;

    // the right neighbour's columns come from lane + 1, except for lane 63 and for the last strip of a plane row: those fetch them
    // themselves (one more load per row in which every other lane is out of range)
    const bool self_halo = lane == 63 || t == g.sxp - 1;
    bool hok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) hok[c] = c < HALO && ix0 + OWN + c >= 0 && ix0 + OWN + c < a.in_w;

    float win[4][NCOL];          // row r of the segment lives in slot r & 3
    ufd_u4 pre[D][NV + 1];       // rows r .. r + D - 1 in flight (+ the self-fetched neighbour columns)
    auto request = [&](int slot, int r) {
        int iy = iy_s + r;
        iy = iy < 0 ? 0 : (iy >= a.in_h ? a.in_h - 1 : iy);          // rows outside the image are masked on arrival
        int vo = voff0 + iy * row_bytes;
        if (r >= NRT || r < r_first || (fixl && iy == 0)) vo = UFD_OOB;              // nothing to fetch: the range check answers with zeros
#pragma unroll
        for (int l = 0; l < NV; ++l) pre[slot][l] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo + 16 * l, 0, 0);
        pre[slot][NV] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (self_halo && vo != UFD_OOB) ? vo + 16 * NV : UFD_OOB, 0, 0);
    };
    // the prologue issues the same sequence of memory instructions as one trip of the loop (its stores are out of range): the compiler's
    // count of what is in flight at the loop header is then the same from both sides, and its waits inside the loop stay exact
#pragma unroll
    for (int p = 0; p < D; ++p) {
        request(p, p);
        if ((p + 4 - (KH - 1)) % DOWN != 0) continue;
        const ufd_u4 z = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(z, wsrc, UFD_OOB, 0, 0);
        if (ragged != 0) {
            __builtin_amdgcn_raw_buffer_store_b32(0u, wsrc, UFD_OOB, 0, 0);
            if (ragged > 1) __builtin_amdgcn_raw_buffer_store_b32(0u, wsrc, UFD_OOB, 0, 0);
            if (ragged > 2) __builtin_amdgcn_raw_buffer_store_b32(0u, wsrc, UFD_OOB, 0, 0);
        }
    }
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        const int r0 = it * D;
#pragma unroll
        for (int p = 0; p < D; ++p) {
            const int r = r0 + p, w = p & 3;
            const int iy = iy_s + r;
            const bool rok = iy >= 0 && iy < a.in_h;          // wave-uniform
            const bool usefix = fixl && iy == 0;
            float raw[OWN];
#pragma unroll
            for (int l = 0; l < NV; ++l) {
                raw[4 * l + 0] = __uint_as_float(pre[p][l].x); raw[4 * l + 1] = __uint_as_float(pre[p][l].y);
                raw[4 * l + 2] = __uint_as_float(pre[p][l].z); raw[4 * l + 3] = __uint_as_float(pre[p][l].w);
            }
#pragma unroll
            for (int c = 0; c < OWN; ++c) {
                const float v = usefix ? fix[c] : raw[c];
                win[w][c] = (rok && cok[c]) ? v : 0.f;
            }
            {
                const float hs[4] = {__uint_as_float(pre[p][NV].x), __uint_as_float(pre[p][NV].y), __uint_as_float(pre[p][NV].z), __uint_as_float(pre[p][NV].w)};
#pragma unroll
                for (int c = 0; c < HALO; ++c) {
                    const float nb = ufd_lane_next(win[w][c]);
                    win[w][OWN + c] = self_halo ? ((rok && hok[c]) ? hs[c] : 0.f) : nb;
                }
            }
            request(p, r + D);
            if ((p + 4 - (KH - 1)) % DOWN != 0) continue;          // r0 is a multiple of 4 and DOWN divides 4: known at compile time
            const int q = (r - (KH - 1)) / DOWN;                   // the output row whose last input row is r (negative in the first rows: dropped)
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KH; ++i) {
                const int sl = (p + 4 - (KH - 1) + i) & 3;          // slot of input row r - (KH - 1) + i
#pragma unroll
                for (int j = 0; j < KW; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(win[sl][c * DOWN + j], kreg[i * KW + j], acc[c]);
            }
            const bool row_ok = r >= KH - 1 && q < SEG && oy_s + q >= 0 && oy_s + q < a.out_h;          // wave-uniform
            const int so = soff0 + (oy_s + q) * orow_bytes;
            const ufd_u4 o = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
            __builtin_amdgcn_raw_buffer_store_b128(o, wsrc, (row_ok && st_vec) ? so : UFD_OOB, 0, 0);
            if (ragged != 0) {                                       // wave-uniform, the same in every row
                const int sr = (row_ok && st_rag) ? so : UFD_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0]), wsrc, sr, 0, 0);
                if (ragged > 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[1]), wsrc, sr == UFD_OOB ? sr : sr + 4, 0, 0);
                if (ragged > 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[2]), wsrc, sr == UFD_OOB ? sr : sr + 8, 0, 0);
            }
        }
    }
}

static bool ufd_roll_geom(UfdRollGeom& g, const UfdArgs& a, int out_cols_per_strip, int nseg)
{
    g.sxp = (a.out_w + out_cols_per_strip - 1) / out_cols_per_strip;
    g.nseg = nseg;
    g.n_strips = a.major * g.sxp;
    g.n_waves = (g.n_strips + 63) / 64 * g.nseg;
    const int64_t ib = a.major * (int64_t)a.in_h * a.in_w * 4, ob = a.major * (int64_t)a.out_h * a.out_w * 4;
    if (ib >= 0x7fff0000LL || ob >= 0x7fff0000LL) return false;
    g.in_bytes = (unsigned)ib;
    g.out_bytes = (unsigned)ob;
    return true;
}

template <int DOWN, int KH, int KW, int SEG, int D>
static int ufd_launch_roll(float* out, const float* in, const float* k, const UfdArgs& a, hipStream_t st)
{
    UfdRollGeom g;
    if (!ufd_roll_geom(g, a, 4, (a.out_h + SEG - 1 + SEG - 1) / SEG)) return HAV_EUNSUP;          // + the rows the shifted borders add
    const int64_t blocks = (g.n_waves + 7) / 8 * 8;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return HAV_EUNSUP;
    hipLaunchKernelGGL((ufd_roll_f32_kernel<DOWN, KH, KW, SEG, D>), dim3((unsigned)blocks), dim3(64), 0, st, out, in, k, a, g);
    HAV_LAUNCH_CHECK();
    return 0;
}

// x2 up-sampling (up = 2, down = 1, FIR <= 4x4, 0 <= pad0 <= 4, EX = pad_x0 & 1): strip t produces the output columns 4 t .. 4 t + 3 (a wave
// stores one contiguous KiB per output row); tap j of output c is real iff c + j - EX is even and then sits on input column
// b + (c + j - EX) / 2, b = 2 t - pad_x0 / 2: 2 owned columns (one 8-byte load) and 2 - EX of the neighbour's.  Rows: input row a
// completes the output rows 2 a + pad_y0 - 3 + q, q = 0, 1, which use rows a - 1 and a (tap i of q is real iff q + 1 + i is even).
// A segment is SEG input rows.
typedef unsigned int ufd_u2 __attribute__((ext_vector_type(2)));
template <int EX, int SEG, int D>
__global__ void __launch_bounds__(64) ufd_roll_up2_f32_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ k, UfdArgs a,
                                                              UfdRollGeom g, int a_min, int a_max)
{
    constexpr int OWN = 2, HALO = 2 - EX, NCOL = OWN + HALO, NRT = SEG + 1, NIT = (NRT + D - 1) / D;
    static_assert((D & 1) == 0, "two window slots");
    const int64_t wv = ufd_roll_wave_index();
    if (wv >= g.n_waves) return;
    const int lane = threadIdx.x;
    const int sg = (int)(wv % g.nseg);
    const int64_t f0 = (wv / g.nseg) * 64 + lane;
    const bool live = f0 < g.n_strips;
    const int64_t f = f0 < g.n_strips ? f0 : g.n_strips - 1;
    const int t = (int)(f % g.sxp);
    const int64_t m = f / g.sxp;
    const int ox0 = 4 * t, ix0 = 2 * t - (a.px0 >> 1);
    const int iy_s = a_min + sg * SEG - (int)(((((wv / g.nseg) * 64 + 31) / g.sxp) * 11) % SEG);          // first input row of the segment's window (borders shifted per plane)
    if (iy_s > a_max || iy_s + SEG <= a_min) return;
    const int r_first = iy_s < a_min ? a_min - iy_s : 0;
    const int oy_s = 2 * iy_s + a.py0 - 1;                   // the output row (rows iy_s, iy_s + 1; q = 0)
    float kreg[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) kreg[i * 4 + j] = (i < a.kh && j < a.kw) ? k[(a.kh - 1 - i) * a.kw + (a.kw - 1 - j)] : 0.f;
    bool cok[OWN], hok[2];
#pragma unroll
    for (int c = 0; c < OWN; ++c) cok[c] = ix0 + c >= 0 && ix0 + c < a.in_w;
#pragma unroll
    for (int c = 0; c < 2; ++c) hok[c] = c < HALO && ix0 + OWN + c >= 0 && ix0 + OWN + c < a.in_w;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, g.out_bytes, 0x00020000);
    const bool fixl = m == 0 && ix0 < 0;          // see ufd_roll_f32_kernel
    float fix[OWN];
#pragma unroll
    for (int c = 0; c < OWN; ++c) fix[c] = 0.f;
    if (fixl) {
#pragma unroll
        for (int c = 0; c < OWN; ++c) if (cok[c]) fix[c] = in[ix0 + c];
    }
    const int voff0 = (int)((m * (int64_t)a.in_h * a.in_w + ix0) * 4);
    const int row_bytes = a.in_w * 4, orow_bytes = a.out_w * 4;
    const int soff0 = (int)((m * (int64_t)a.out_h * a.out_w + ox0) * 4);
    const bool st_vec = live && ox0 + 3 < a.out_w;
    const int ragged = a.out_w & 3;
    const bool st_rag = live && ragged != 0 && ox0 + 3 >= a.out_w && ox0 < a.out_w;
    const bool self_halo = lane == 63 || t == g.sxp - 1;

    float win[2][NCOL];          // row r of the segment lives in slot r & 1
    ufd_u2 pre[D][2];
    auto request = [&](int slot, int r) {
        int iy = iy_s + r;
        iy = iy < 0 ? 0 : (iy >= a.in_h ? a.in_h - 1 : iy);
        int vo = voff0 + iy * row_bytes;
        if (r >= NRT || r < r_first || (fixl && iy == 0)) vo = UFD_OOB;
        pre[slot][0] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, 0, 0);
        pre[slot][1] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (self_halo && vo != UFD_OOB) ? vo + 4 * OWN : UFD_OOB, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < D; ++p) {          // same sequence of memory instructions as one trip of the loop (see ufd_roll_f32_kernel)
        request(p, p);
        const ufd_u4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_raw_buffer_store_b128(z, wsrc, UFD_OOB, 0, 0);
            if (ragged != 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < ragged) __builtin_amdgcn_raw_buffer_store_b32(0u, wsrc, UFD_OOB, 0, 0);
            }
        }
    }
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        const int r0 = it * D;
#pragma unroll
        for (int p = 0; p < D; ++p) {
            const int r = r0 + p, w = p & 1;
            const int iy = iy_s + r;
            const bool rok = iy >= 0 && iy < a.in_h;
            const bool usefix = fixl && iy == 0;
            const float raw[2] = {__uint_as_float(pre[p][0].x), __uint_as_float(pre[p][0].y)};
#pragma unroll
            for (int c = 0; c < OWN; ++c) {
                const float v = usefix ? fix[c] : raw[c];
                win[w][c] = (rok && cok[c]) ? v : 0.f;
            }
            {
                const float hs[2] = {__uint_as_float(pre[p][1].x), __uint_as_float(pre[p][1].y)};
#pragma unroll
                for (int c = 0; c < HALO; ++c) {
                    const float nb = ufd_lane_next(win[w][c]);
                    win[w][OWN + c] = self_halo ? ((rok && hok[c]) ? hs[c] : 0.f) : nb;
                }
            }
            request(p, r + D);
            // rows r - 1 and r complete the output rows oy_s + 2 (r - 1) + q
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oy = oy_s + 2 * (r - 1) + q;
                const bool row_ok = r >= 1 && r < NRT && oy >= 0 && oy < a.out_h;          // wave-uniform
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (((q + 1 + i) & 1) != 0) continue;          // tap row i of this output row sits between two samples
                    const int sl = (w + 1 + ((q + i - 1) >> 1)) & 1;          // q = 0: i = 1 -> row r - 1, i = 3 -> row r; q = 1: i = 0 -> r - 1, i = 2 -> r
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (((c + j + EX) & 1) != 0) continue;
                            acc[c] = fmaf(win[sl][(c + j - EX) >> 1], kreg[i * 4 + j], acc[c]);
                        }
                }
                const int so = soff0 + oy * orow_bytes;
                const ufd_u4 v0 = {__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]), __float_as_uint(acc[3])};
                __builtin_amdgcn_raw_buffer_store_b128(v0, wsrc, (row_ok && st_vec) ? so : UFD_OOB, 0, 0);
                if (ragged != 0) {                                   // wave-uniform
                    const int sr = (row_ok && st_rag) ? so : UFD_OOB;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (c < ragged) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[c]), wsrc, sr == UFD_OOB ? sr : sr + 4 * c, 0, 0);
                }
            }
        }
    }
}

template <int SEG, int D>
static int ufd_launch_roll_up2(float* out, const float* in, const float* k, const UfdArgs& a, hipStream_t st)
{
    UfdRollGeom g;
    const int a_min = -(a.py0 >> 1), a_max = (a.out_h - 1 - a.py0 + 1) >> 1;          // first window rows of output rows 0 and out_h - 1
    if (!ufd_roll_geom(g, a, 4, (a_max - a_min + 1 + SEG - 1 + SEG - 1) / SEG)) return HAV_EUNSUP;
    const int64_t blocks = (g.n_waves + 7) / 8 * 8;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return HAV_EUNSUP;
    if (a.px0 & 1) hipLaunchKernelGGL((ufd_roll_up2_f32_kernel<1, SEG, D>), dim3((unsigned)blocks), dim3(64), 0, st, out, in, k, a, g, a_min, a_max);
    else hipLaunchKernelGGL((ufd_roll_up2_f32_kernel<0, SEG, D>), dim3((unsigned)blocks), dim3(64), 0, st, out, in, k, a, g, a_min, a_max);
    HAV_LAUNCH_CHECK();
    return 0;
}

// kernel choice for A/B runs and the bit-identity tests: HAVATAR_UFD=legacy (read once) or hav_lab_upfirdn2d(mode, seg) at run time
// (mode 1: row-streaming kernels for blur and x2 up-sampling, the default; 2: for decimation too; 0: the strip / tiled kernels; seg > 0
// forces the rows per segment).  Not part of the ABI.
static std::atomic<int> g_ufd_mode{-1}, g_ufd_seg{0};
static int ufd_lab_mode()
{
    int m = g_ufd_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("HAVATAR_UFD");
        m = (e && !strcmp(e, "legacy")) ? 0 : 1;
        const char* sg = getenv("HAVATAR_UFD_SEG");
        if (sg) g_ufd_seg.store(atoi(sg), std::memory_order_relaxed);
        g_ufd_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
extern "C" void hav_lab_upfirdn2d(int mode, int seg)
{
    g_ufd_mode.store(mode, std::memory_order_relaxed);
    g_ufd_seg.store(seg > 0 ? seg : 0, std::memory_order_relaxed);
}

template <typename T>
static int ufd_launch(void* out_, const void* in_, const float* k, const UfdArgs& a, hipStream_t st)
{
    T* out = (T*)out_;
    const T* in = (const T*)in_;
    if (a.minor == 1 && a.up_x == a.up_y && a.down_x == a.down_y) {
        const int up = a.up_x, dn = a.down_x, kh = a.kh, kw = a.kw;
        if constexpr (std::is_same<T, float>::value) {
            // the row-streaming kernels take the three classes of the StyleGAN blocks; HAVATAR_UFD=legacy (read once) keeps the strip / tiled
            // kernels below for A/B runs and for the bit-identity tests, HAVATAR_UFD_SEG forces the rows per segment
            const int roll_mode = ufd_lab_mode(), roll_seg = g_ufd_seg.load(std::memory_order_relaxed);
            // (tensors of 2 GiB and more stay on the kernels below: the buffer descriptors of the row-streaming kernels take 32-bit offsets)
            const bool fits32 = a.major * (int64_t)a.in_h * a.in_w * 4 < 0x7fff0000LL && a.major * (int64_t)a.out_h * a.out_w * 4 < 0x7fff0000LL;
            if (roll_mode && fits32 && a.out_w >= 64 && kh <= 4 && kw <= 4 && a.px0 >= 0 && a.py0 >= 0 && a.px0 <= 4 && a.py0 <= 4 && a.in_w > 8) {
                // rows per segment: the longest segment that still gives the chip 9 waves per compute unit (fewer window rows re-read);
                // small launches take the shortest one
                auto pick = [&](int rows, int shortest) {
                    if (roll_seg) return roll_seg;
                    const int64_t grp = (a.major * ((a.out_w + 3) / 4) + 63) / 64, want = (int64_t)hav_num_cus() * 9;
                    for (int sgm = 32; sgm > shortest; sgm >>= 1) if (grp * ((rows + sgm - 1) / sgm) >= want) return sgm;
                    return shortest;
                };
#define UFD_ROLL(DN_, KH_, KW_)                                                                                                       \
    do {                                                                                                                              \
        const int sgm = pick(a.out_h, 8);                                                                                             \
        if (sgm >= 32) return ufd_launch_roll<DN_, KH_, KW_, 33, 4>((float*)out, (const float*)in, k, a, st);                         \
        if (sgm >= 16) return ufd_launch_roll<DN_, KH_, KW_, 17, 4>((float*)out, (const float*)in, k, a, st);                         \
        return ufd_launch_roll<DN_, KH_, KW_, 9, 4>((float*)out, (const float*)in, k, a, st);                                         \
    } while (0)
                if (up == 1 && dn == 1 && (kh > 3 || kw > 3)) UFD_ROLL(1, 4, 4);
                if (up == 1 && dn == 1 && kw >= 2) UFD_ROLL(1, 3, 3);
                // decimation: the LDS-tiled kernel below is faster (19.3 vs 20.3 us at [64,513,513]); mode 2 takes the window kernel (tests)
                if (up == 1 && dn == 2 && kw > 2 && roll_mode == 2) UFD_ROLL(2, 4, 4);
#undef UFD_ROLL
                if (up == 2 && dn == 1) {
                    const int sgm = pick((a.out_h + 1) / 2, 4);
                    if (sgm >= 32) return ufd_launch_roll_up2<33, 4>((float*)out, (const float*)in, k, a, st);
                    if (sgm >= 16) return ufd_launch_roll_up2<17, 4>((float*)out, (const float*)in, k, a, st);
                    if (sgm >= 8) return ufd_launch_roll_up2<9, 4>((float*)out, (const float*)in, k, a, st);
                    return ufd_launch_roll_up2<5, 4>((float*)out, (const float*)in, k, a, st);
                }
            }
            // f32 blur (no re-sampling) on planes wide enough for 16-byte strips: the direct kernel.  Measured on MI355X with every launch
            // on its own buffers (tools/bench_ops.py): [64,513,513] 4x4 blur 37.4 -> 34.4 us = 3.9 TB/s, which is 82 % of what a plain
            // device copy of the same 135 MB reaches (28.3 us) and 71 % of this library's best streaming kernel at that size.  The
            // decimating cases stay on the LDS-tiled kernel: their strips need 3 loads per input row and were slower (25.7 vs 21.3 us).
            if (up == 1 && dn == 1 && a.out_w >= 64 && a.in_w >= 64) {
                if (kh <= 4 && kw <= 4 && (kh > 3 || kw > 3)) return ufd_launch_direct<1, 4, 4, 8>((float*)out, (const float*)in, k, a, st);
                if (kh <= 3 && kw <= 3) return ufd_launch_direct<1, 3, 3, 8>((float*)out, (const float*)in, k, a, st);
            }
            // f32 decimation by 2 with a FIR of up to 4x4 taps stays on the LDS-tiled kernel: the direct kernel with the lane-to-lane halo
            // (round 3) is bit-identical but no faster -- [64,513,513]: 22.3 us (4 rows per thread) / 23.5 us (8 rows) against 21.3 us,
            // tools/ufd_ab.sh -- so it only runs on request (HAVATAR_UFD_DOWN2=direct|sr8, read once; the parity test drives all three)
            if (up == 1 && dn == 2 && kh <= 4 && kw <= 4 && (kh > 2 || kw > 2) && a.out_w >= 64 && a.px0 >= 0 && a.py0 >= 0) {
                static const int mode = [] { const char* e = getenv("HAVATAR_UFD_DOWN2"); return !e ? 0 : (!strcmp(e, "direct") ? 1 : (!strcmp(e, "sr8") ? 2 : 0)); }();
                if (mode == 1) return ufd_launch_down2_direct<4, 4, 4>((float*)out, (const float*)in, k, a, st);
                if (mode == 2) return ufd_launch_down2_direct<4, 4, 8>((float*)out, (const float*)in, k, a, st);
            }
            // f32 x2 up-sampling with a FIR of up to 4x4 taps (non-negative padding): the register-window kernel
            if (up == 2 && dn == 1 && kh <= 4 && kw <= 4 && a.px0 >= 0 && a.py0 >= 0 && a.out_w >= 64)
                return ufd_launch_up2_direct((float*)out, (const float*)in, k, a, st);
        }
        // the six parameter classes the StyleGAN blocks use (modes 1-6) + up2/down2
        if (up == 1 && dn == 1 && kh <= 3 && kw <= 3) return ufd_launch_tiled<T, 1, 1, 3, 3>(out, in, k, a, st);
        if (up == 1 && dn == 1 && kh <= 4 && kw <= 4) return ufd_launch_tiled<T, 1, 1, 4, 4>(out, in, k, a, st);
        if (up == 2 && dn == 1 && kh <= 2 && kw <= 2) return ufd_launch_tiled<T, 2, 1, 2, 2>(out, in, k, a, st);
        if (up == 2 && dn == 1 && kh <= 4 && kw <= 4) return ufd_launch_tiled<T, 2, 1, 4, 4>(out, in, k, a, st);
        if (up == 1 && dn == 2 && kh <= 2 && kw <= 2) return ufd_launch_tiled<T, 1, 2, 2, 2>(out, in, k, a, st);
        if (up == 1 && dn == 2 && kh <= 4 && kw <= 4) return ufd_launch_tiled<T, 1, 2, 4, 4>(out, in, k, a, st);
    }
    const int64_t total = a.major * a.out_h * a.out_w * a.minor;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)hav_num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((ufd_generic_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, out, in, k, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_upfirdn2d_out_size(int in_h, int in_w, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                      int pad_x0, int pad_x1, int pad_y0, int pad_y1, int* out_h, int* out_w)
{
    if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kh < 1 || kw < 1 || in_h < 1 || in_w < 1) return HAV_EINVAL;
    // upfirdn2d_kernel.cu:237-240
    const int oh = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    const int ow = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    if (out_h) *out_h = oh;
    if (out_w) *out_w = ow;
    return (oh >= 1 && ow >= 1) ? 0 : HAV_EINVAL;
}

extern "C" int hav_upfirdn2d(void* out, const void* in, const float* kernel, int dtype, int64_t major, int in_h, int in_w,
                             int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                             int pad_y0, int pad_y1, void* stream)
{
    int oh = 0, ow = 0;
    if (major < 0 || minor < 1) return HAV_EINVAL;
    int rc = hav_upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, &oh, &ow);
    if (rc) return rc;
    if (major == 0) return 0;
    if (!out || !in || !kernel) return HAV_EINVAL;
    UfdArgs a;
    a.major = major; a.in_h = in_h; a.in_w = in_w; a.minor = minor; a.kh = kh; a.kw = kw;
    a.up_x = up_x; a.up_y = up_y; a.down_x = down_x; a.down_y = down_y; a.px0 = pad_x0; a.py0 = pad_y0;
    a.out_h = oh; a.out_w = ow;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
    case HAV_F32: return ufd_launch<float>(out, in, kernel, a, st);
    case HAV_F16: return ufd_launch<__half>(out, in, kernel, a, st);
    case HAV_BF16: return ufd_launch<__hip_bfloat16>(out, in, kernel, a, st);
    case HAV_F64: return ufd_launch<double>(out, in, kernel, a, st);
    default: return HAV_EUNSUP;
    }
}

// ================================================================================================
// Haar analysis / synthesis of the wavelet-domain skip path of SWGAN_unet (model/styleUnet.py:510-560 HaarTransform /
// InverseHaarTransform: four upfirdn2d calls with the 2x2 kernels ll, lh, hl, hh + a cat, resp. four up-sampling calls + three adds)
// as ONE pass each.  Arithmetic per output element is the unfused sequence's, operation for operation (tap order i asc, j asc with
// the flipped kernel and fused multiply-adds for the analysis; one rounded product per band summed ((ll + lh) + hl) + hh for the
// synthesis), so the results are bit-identical to the four-call path -- asserted in tests/test_ops_gpu.py.
//   dwt : in [N, H, W]  -> out [B, 4, C, H/2, W/2] viewed as [B, 4C, H/2, W/2]  (N = B*C planes; band-major channel order of the cat)
//   idwt: in [B, 4, C, H, W] -> out [N, 2H, 2W]
// k: [4][2][2] floats (the four 2x2 kernels as upfirdn2d receives them, unflipped).
// One thread = 4 output columns of the analysis (2 x 8 inputs) / 4 input columns of the synthesis (2 x 8 outputs): 16-byte accesses.
__global__ void __launch_bounds__(256) haar_dwt_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ k,
                                                       int B, int C, int H, int W)
{
    const int OH = H >> 1, OW = W >> 1, OW4 = OW >> 2;
    const int64_t total = (int64_t)B * C * OH * OW4;
    float kk[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) kk[q] = k[q];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(idx % OW4);
        int64_t t = idx / OW4;
        const int oy = (int)(t % OH);
        const int64_t n = t / OH;          // plane b*C + c
        const int64_t b = n / C, c = n - b * C;
        const float* r0 = in + (n * H + 2 * oy) * (int64_t)W + 8 * x4;
        const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(r0 + W), b1 = *reinterpret_cast<const float4*>(r0 + W + 4);
        const float top[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float bot[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int band = 0; band < 4; ++band) {
            const float* kb = kk + 4 * band;          // taps (i, j) use the flipped kernel k[1-i][1-j]
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = fmaf(top[2 * q], kb[3], 0.f);
                v = fmaf(top[2 * q + 1], kb[2], v);
                v = fmaf(bot[2 * q], kb[1], v);
                v = fmaf(bot[2 * q + 1], kb[0], v);
                o[q] = v;
            }
            float* dst = out + ((((b * 4 + band) * C + c) * OH + oy) * (int64_t)OW) + 4 * x4;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

__global__ void __launch_bounds__(256) haar_idwt_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ k,
                                                        int B, int C, int H, int W)
{
    const int W4 = W >> 2;
    const int64_t total = (int64_t)B * C * H * W4;
    float kk[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) kk[q] = k[q];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(idx % W4);
        int64_t t = idx / W4;
        const int y = (int)(t % H);
        const int64_t n = t / H;
        const int64_t b = n / C, c = n - b * C;
        float v[4][4];
#pragma unroll
        for (int band = 0; band < 4; ++band) {
            const float4 q4 = *reinterpret_cast<const float4*>(in + ((((b * 4 + band) * C + c) * H + y) * (int64_t)W) + 4 * x4);
            v[band][0] = q4.x; v[band][1] = q4.y; v[band][2] = q4.z; v[band][3] = q4.w;
        }
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
            float o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rx = 0; rx < 2; ++rx) {
                    // out[2y+ry, 2x+rx] = in[y, x] * k[ry][rx] per band (the one tap that lands on a real sample), bands summed in order
                    // (individually rounded products and sums: the compiler would contract r + v*k into an fma)
                    float r = mul_rn(v[0][q], kk[0 + 2 * ry + rx]);
                    r = add_rn(r, mul_rn(v[1][q], kk[4 + 2 * ry + rx]));
                    r = add_rn(r, mul_rn(v[2][q], kk[8 + 2 * ry + rx]));
                    r = add_rn(r, mul_rn(v[3][q], kk[12 + 2 * ry + rx]));
                    o[2 * q + rx] = r;
                }
            float* dst = out + (n * 2 * H + 2 * y + ry) * (int64_t)(2 * W) + 8 * x4;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// The wavelet-domain skip path of ToRGB (model/styleUnet.py:476-480: skip = dwt(upsample(iwt(skip)))) as ONE pass: synthesis, x2 up-sampling with
// the 4x4 FIR (pad (2, 1)) and analysis, [B,4C,H,W] -> [B,4C,2H,2W].  A thread produces 4 output columns of one (plane, row) in all four bands:
// it rebuilds the 3 x 6 synthesised pixels around them from 2 x 4 input pixels of each band (hav_haar_idwt's arithmetic: one rounded product
// per band, summed ((ll + lh) + hl) + hh), the 2 x 8 up-sampled pixels from those (the up-sampling kernels' FMA chain over the real taps, i then
// j ascending, zeros outside the image) and the four band values of each output from their 2 x 2 block (hav_haar_dwt's FMA chain): the three
// stages' results, operation for operation -- bit-identical to the three-launch sequence (16 -> 5 launches per SWGAN_unet forward).
__global__ void __launch_bounds__(256) haar_up2_kernel(float* __restrict__ out, const float* __restrict__ in, const float* __restrict__ ki,
                                                       const float* __restrict__ fir, const float* __restrict__ kd, int B, int C, int H, int W)
{
    const int OH = 2 * H, OW = 2 * W, OW4 = OW >> 2, IH = 2 * H, IW = 2 * W;          // output and synthesised-image sizes (both 2H x 2W)
    const int64_t total = (int64_t)B * C * OH * OW4;
    float ks[16], kf[16], ka[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { ks[q] = ki[q]; ka[q] = kd[q]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[i * 4 + j] = fir[(3 - i) * 4 + (3 - j)];          // flipped FIR (upfirdn2d_kernel.cu:136-137)
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int x4 = (int)(idx % OW4);
        int64_t t = idx / OW4;
        const int Y = (int)(t % OH);
        const int64_t n = t / OH;
        const int64_t b = n / C, c = n - b * C;
        const int X0 = 4 * x4;
        // input pixels: rows (Y - 1) >> 1 and + 1, columns X0 / 2 - 1 .. X0 / 2 + 2, four bands
        const int ra = (Y - 1) >> 1, ca = (X0 >> 1) - 1;
        float v[4][2][4];
#pragma unroll
        for (int band = 0; band < 4; ++band) {
            const float* pl = in + (((b * 4 + band) * C + c) * (int64_t)H) * W;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int iy = ra + r, ix = ca + q;
                    v[band][r][q] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? pl[(int64_t)iy * W + ix] : 0.f;
                }
        }
        // synthesised image I[Y - 1 + a][X0 - 1 + e], a = 0..2, e = 0..5 (zero outside)
        float I[3][6];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int y = Y - 1 + a, x = X0 - 1 + e;
                const int r = (y >> 1) - ra, q = (x >> 1) - ca, ry = y & 1, rx = x & 1;
                float s;
                // (r, q, ry, rx depend on the parities of Y only through y: resolved per thread; the selects below keep the indexing static)
                const float v0 = r == 0 ? (q == 0 ? v[0][0][0] : q == 1 ? v[0][0][1] : q == 2 ? v[0][0][2] : v[0][0][3]) : (q == 0 ? v[0][1][0] : q == 1 ? v[0][1][1] : q == 2 ? v[0][1][2] : v[0][1][3]);
                const float v1 = r == 0 ? (q == 0 ? v[1][0][0] : q == 1 ? v[1][0][1] : q == 2 ? v[1][0][2] : v[1][0][3]) : (q == 0 ? v[1][1][0] : q == 1 ? v[1][1][1] : q == 2 ? v[1][1][2] : v[1][1][3]);
                const float v2 = r == 0 ? (q == 0 ? v[2][0][0] : q == 1 ? v[2][0][1] : q == 2 ? v[2][0][2] : v[2][0][3]) : (q == 0 ? v[2][1][0] : q == 1 ? v[2][1][1] : q == 2 ? v[2][1][2] : v[2][1][3]);
                const float v3 = r == 0 ? (q == 0 ? v[3][0][0] : q == 1 ? v[3][0][1] : q == 2 ? v[3][0][2] : v[3][0][3]) : (q == 0 ? v[3][1][0] : q == 1 ? v[3][1][1] : q == 2 ? v[3][1][2] : v[3][1][3]);
                const int kq = 2 * ry + rx;
                const float k0 = kq == 0 ? ks[0] : kq == 1 ? ks[1] : kq == 2 ? ks[2] : ks[3];
                const float k1 = kq == 0 ? ks[4] : kq == 1 ? ks[5] : kq == 2 ? ks[6] : ks[7];
                const float k2 = kq == 0 ? ks[8] : kq == 1 ? ks[9] : kq == 2 ? ks[10] : ks[11];
                const float k3 = kq == 0 ? ks[12] : kq == 1 ? ks[13] : kq == 2 ? ks[14] : ks[15];
                s = mul_rn(v0, k0);
                s = add_rn(s, mul_rn(v1, k1));
                s = add_rn(s, mul_rn(v2, k2));
                s = add_rn(s, mul_rn(v3, k3));
                I[a][e] = (y >= 0 && y < IH && x >= 0 && x < IW) ? s : 0.f;
            }
        // up-sampled pixels U[2Y + q][2 X0 + p]: row taps i with q + i even on I row Y - 1 + (q + i) / 2, column taps j with p + j even on
        // I column X0 - 1 + (p + j) / 2 (pad 2: the zero-stuffed coordinate of tap i of row 2Y + q is 2Y + q + i - 2)
        float U[2][8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if ((q + i) & 1) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if ((p + j) & 1) continue;
                        acc = fmaf(I[(q + i) >> 1][(p + j) >> 1], kf[i * 4 + j], acc);
                    }
                }
                U[q][p] = acc;
            }
        // analysis: band value of output (Y, X0 + t) from U[0..1][2t..2t+1], taps (i, j) with the flipped 2x2 kernel
#pragma unroll
        for (int band = 0; band < 4; ++band) {
            const float* kb = ka + 4 * band;
            float o[4];
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                float w = fmaf(U[0][2 * tq], kb[3], 0.f);
                w = fmaf(U[0][2 * tq + 1], kb[2], w);
                w = fmaf(U[1][2 * tq], kb[1], w);
                w = fmaf(U[1][2 * tq + 1], kb[0], w);
                o[tq] = w;
            }
            float* dst = out + ((((b * 4 + band) * C + c) * OH + Y) * (int64_t)OW) + X0;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

extern "C" int hav_haar_up2(float* out, const float* in, const float* ki4x2x2, const float* fir4x4, const float* kd4x2x2, int B, int C, int H, int W,
                            void* stream)
{
    if (!out || !in || !ki4x2x2 || !fir4x4 || !kd4x2x2 || B < 1 || C < 1 || H < 1 || W < 1) return HAV_EINVAL;
    if ((W & 1) || (((uintptr_t)out) & 15)) return HAV_EUNSUP;
    const int64_t total = (int64_t)B * C * (2 * H) * (W / 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 32) blocks = (int64_t)hav_num_cus() * 32;
    hipLaunchKernelGGL(haar_up2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, in, ki4x2x2, fir4x4, kd4x2x2, B, C, H, W);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_haar_dwt(float* out, const float* in, const float* k4x2x2, int B, int C, int H, int W, void* stream)
{
    if (!out || !in || !k4x2x2 || B < 1 || C < 1 || H < 2 || W < 2) return HAV_EINVAL;
    if ((H & 1) || (W % 8) || (((uintptr_t)out | (uintptr_t)in) & 15)) return HAV_EUNSUP;
    const int64_t total = (int64_t)B * C * (H / 2) * (W / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 16) blocks = (int64_t)hav_num_cus() * 16;
    hipLaunchKernelGGL(haar_dwt_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, in, k4x2x2, B, C, H, W);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_haar_idwt(float* out, const float* in, const float* k4x2x2, int B, int C, int H, int W, void* stream)
{
    if (!out || !in || !k4x2x2 || B < 1 || C < 1 || H < 1 || W < 1) return HAV_EINVAL;
    if ((W % 4) || (((uintptr_t)out | (uintptr_t)in) & 15)) return HAV_EUNSUP;
    const int64_t total = (int64_t)B * C * H * (W / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 16) blocks = (int64_t)hav_num_cus() * 16;
    hipLaunchKernelGGL(haar_idwt_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, in, k4x2x2, B, C, H, W);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// Demodulation of a modulated convolution under autograd (training path):
//   q[i,o] = c^2 sum_k W[o,i,k]^2 ,   d[b,o] = rsqrt( sum_i s[b,i]^2 q[i,o] + eps )
// (reference model/styleUnet.py:214-227 in its factored form) and its backward
//   gt[b,o] = -1/2 gd[b,o] d[b,o]^3 ,  gs[b,i] = 2 s[b,i] sum_o gt[b,o] q[i,o] ,  gW[o,i,k] = 2 c^2 W[o,i,k] sum_b gt[b,o] s[b,i]^2 .
// Two launches forward and two backward instead of the ~8 + ~14 ATen launches of the elementwise statement (pow, sum, transpose,
// square, matmul, add, rsqrt and their gradients), three of which stream the whole weight tensor.
// ================================================================================================
__global__ void __launch_bounds__(256) demod_wsq_kernel(float* __restrict__ q, const float* __restrict__ W, float c2, int Cout, int Cin, int KK)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // = o * Cin + i : coalesced over W
    if (idx >= (int64_t)Cout * Cin) return;
    const int o = (int)(idx / Cin), i = (int)(idx - (int64_t)o * Cin);
    const float* w = W + idx * KK;
    float s = 0.f;
    for (int k = 0; k < KK; ++k) s = fmaf(w[k], w[k], s);
    q[(int64_t)i * Cout + o] = c2 * s;
}

// grid (ceil(Cout / 64), B); 256 threads = 64 outputs x 4 slices of the input channels
__global__ void __launch_bounds__(256) demod_fwd_kernel(float* __restrict__ d, const float* __restrict__ s, const float* __restrict__ q, float eps,
                                                        int Cin, int Cout)
{
    __shared__ float part[4][64];
    const int b = blockIdx.y, ol = threadIdx.x & 63, sl = threadIdx.x >> 6, o = blockIdx.x * 64 + ol;
    float acc = 0.f;
    if (o < Cout) {
        // 8 (s, q) pairs requested before the first is used: with 16 workgroups in the whole launch nothing else hides a memory round trip per
        // input channel (30 us at 512 x 512 before; hav_demod_fwd's two kernels together take 14 us now); the sum keeps its order
        int i = sl;
        for (; i + 28 < Cin; i += 32) {
            float sv[8], qv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { sv[u] = s[(int64_t)b * Cin + i + 4 * u]; qv[u] = q[(int64_t)(i + 4 * u) * Cout + o]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(sv[u] * sv[u], qv[u], acc);
        }
        for (; i < Cin; i += 4) { const float v = s[(int64_t)b * Cin + i]; acc = fmaf(v * v, q[(int64_t)i * Cout + o], acc); }
    }
    part[sl][ol] = acc;
    __syncthreads();
    if (sl == 0 && o < Cout) d[(int64_t)b * Cout + o] = rsqrtf(((part[0][ol] + part[1][ol]) + part[2][ol]) + part[3][ol] + eps);
}

// one workgroup per input channel i: gs[b,i] and gq[i,:]
__global__ void __launch_bounds__(256) demod_bwd_kernel(float* __restrict__ gs, float* __restrict__ gq, const float* __restrict__ gd,
                                                        const float* __restrict__ s, const float* __restrict__ d, const float* __restrict__ q,
                                                        int B, int Cin, int Cout)
{
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int b = 0; b < B; ++b) {
        float acc = 0.f;
        for (int o = tid; o < Cout; o += 256) {
            const float dd = d[(int64_t)b * Cout + o];
            acc = fmaf(-0.5f * gd[(int64_t)b * Cout + o] * dd * dd * dd, q[(int64_t)i * Cout + o], acc);
        }
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k, 64);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0) gs[(int64_t)b * Cin + i] = 2.0f * s[(int64_t)b * Cin + i] * (((red[0] + red[1]) + red[2]) + red[3]);
        __syncthreads();
    }
    for (int o = tid; o < Cout; o += 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const float dd = d[(int64_t)b * Cout + o], sv = s[(int64_t)b * Cin + i];
            acc = fmaf(-0.5f * gd[(int64_t)b * Cout + o] * dd * dd * dd, sv * sv, acc);
        }
        gq[(int64_t)i * Cout + o] = acc;
    }
}

__global__ void __launch_bounds__(256) demod_gw_kernel(float* __restrict__ gW, const float* __restrict__ W, const float* __restrict__ gq, float c2x2,
                                                       int Cout, int Cin, int KK, int64_t total)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t oi = e / KK;
        const int o = (int)(oi / Cin), i = (int)(oi - (int64_t)o * Cin);
        gW[e] = c2x2 * W[e] * gq[(int64_t)i * Cout + o];
    }
}

extern "C" int hav_demod_fwd(float* d, float* q, const float* s, const float* W, float scale, float eps, int B, int Cin, int Cout, int KK, void* stream)
{
    if (!d || !q || !s || !W || B < 1 || Cin < 1 || Cout < 1 || KK < 1) return HAV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(demod_wsq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, W, scale * scale, Cout, Cin, KK);
    HAV_LAUNCH_CHECK();
    hipLaunchKernelGGL(demod_fwd_kernel, dim3((Cout + 63) / 64, B), dim3(256), 0, st, d, s, q, eps, Cin, Cout);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_demod_bwd(float* gs, float* gW, float* gq_scratch, const float* gd, const float* s, const float* d, const float* q, const float* W,
                             float scale, int B, int Cin, int Cout, int KK, void* stream)
{
    if (!gs || !gW || !gq_scratch || !gd || !s || !d || !q || !W || B < 1 || Cin < 1 || Cout < 1 || KK < 1) return HAV_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(demod_bwd_kernel, dim3(Cin), dim3(256), 0, st, gs, gq_scratch, gd, s, d, q, B, Cin, Cout);
    HAV_LAUNCH_CHECK();
    const int64_t total = (int64_t)Cout * Cin * KK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 16) blocks = (int64_t)hav_num_cus() * 16;
    hipLaunchKernelGGL(demod_gw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gW, W, gq_scratch, 2.0f * scale * scale, Cout, Cin, KK, total);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// ToRGB: the modulated 1x1 convolution (demodulate=False) + bias + skip add of a StyleGAN2 generator level
// (reference model/styleUnet.py:602-628, ModulatedConv2d :165-297 with kernel_size 1) as ONE pass over the activations:
//   y[b,o,p] = sum_i (scale * W[o,i] * s[b,i]) * x[b,i,p] + bias[o] (+ skip[b,o,p])
// The reference's route is x * s (a full read + write of x), an fp32 GEMM with 3 or 12 output rows (MIOpen: NHWC transposes around it),
// a bias add and the skip add: four to five passes over x for 2 * Cout FLOP per element.  Here a workgroup owns 64 pixel quads; its
// four waves split the input channels, every lane streams float4s of x (1 KB per wave and load, coalesced), the Cout x Cin weight
// matrix with the modulation folded in sits in LDS ([i][Cout] rows: three broadcast ds_read_b128 per channel), fp32 FMA chains in
// channel order inside a slice, the four slices added in a fixed order (bit-reproducible).  HBM-bound: 4 B per element of x.
// ================================================================================================
typedef float nt_f4v __attribute__((ext_vector_type(4)));
template <int COUT, int KS>          // KS = waves of a workgroup that split the input channels (4: small maps / many channels; 1: large maps)
__global__ void __launch_bounds__(256) torgb_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ W,
                                                    const float* __restrict__ s, const float* __restrict__ bias, const float* __restrict__ skip,
                                                    float scale, int Cin, int64_t HW)
{
    constexpr int CP = (COUT + 3) & ~3;                      // row pitch of the LDS weight matrix (floats)
    constexpr int QB = 256 / KS;                             // pixel quads per workgroup
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wm = sm;                                          // [Cin][CP]
    const int b = blockIdx.y, tid = threadIdx.x, q = tid % QB, k = tid / QB;
    for (int i = tid; i < Cin; i += 256) {                   // one input channel per thread: COUT coalesced weight loads in flight
        const float sv = s ? s[(size_t)b * Cin + i] : 1.0f;
        float wv[CP];
#pragma unroll
        for (int o = 0; o < CP; ++o) wv[o] = o < COUT ? W[(size_t)o * Cin + i] : 0.f;
#pragma unroll
        for (int o = 0; o < CP; ++o) wm[(size_t)i * CP + o] = o < COUT ? (scale * wv[o]) * sv : 0.f;          // (scale * W) * s: ModulatedConv2d's order (:250)
    }
    __syncthreads();
    const int64_t quad = (int64_t)blockIdx.x * QB + q, nquad = HW >> 2;
    const bool live = quad < nquad;
    const int per = (Cin + KS - 1) / KS, i0 = k * per, i1 = min(Cin, i0 + per);
    float4 acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* xb = reinterpret_cast<const float4*>(x + (size_t)b * Cin * HW) + (live ? quad : 0);
    const int64_t cs = HW >> 2;                              // float4s between consecutive channels
    constexpr int U = KS == 1 ? 8 : 4;                       // channel loads in flight per lane
    int i = i0;
    for (; i + U <= i1; i += U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const nt_f4v t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4v*>(xb + (int64_t)(i + u) * cs)); xv[u] = make_float4(t.x, t.y, t.z, t.w); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4* wr = reinterpret_cast<const float4*>(wm + (size_t)(i + u) * CP);
#pragma unroll
            for (int o4 = 0; o4 < CP / 4; ++o4) {
                const float4 w4 = wr[o4];
                const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = 4 * o4 + e;
                    if (o < COUT) {
                        acc[o].x = fmaf(wv[e], xv[u].x, acc[o].x); acc[o].y = fmaf(wv[e], xv[u].y, acc[o].y);
                        acc[o].z = fmaf(wv[e], xv[u].z, acc[o].z); acc[o].w = fmaf(wv[e], xv[u].w, acc[o].w);
                    }
                }
            }
        }
    }
    for (; i < i1; ++i) {
        const float4 xv = xb[(int64_t)i * cs];
        const float* wr = wm + (size_t)i * CP;
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float w = wr[o];
            acc[o].x = fmaf(w, xv.x, acc[o].x); acc[o].y = fmaf(w, xv.y, acc[o].y);
            acc[o].z = fmaf(w, xv.z, acc[o].z); acc[o].w = fmaf(w, xv.w, acc[o].w);
        }
    }
    if (KS > 1) {
        float4* r4 = reinterpret_cast<float4*>(sm + (size_t)Cin * CP);          // [KS - 1][COUT][QB]
        if (k > 0) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) r4[((k - 1) * COUT + o) * QB + q] = acc[o];
        }
        __syncthreads();
        if (k == 0) {
#pragma unroll
            for (int o = 0; o < COUT; ++o)
#pragma unroll
                for (int kk = 0; kk < KS - 1; ++kk) {                              // slices 1, 2, ... in this order
                    const float4 u = r4[(kk * COUT + o) * QB + q];
                    acc[o].x += u.x; acc[o].y += u.y; acc[o].z += u.z; acc[o].w += u.w;
                }
        }
    }
    if (k == 0 && live) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            float4 t = acc[o];
            const float bb = bias ? bias[o] : 0.f;
            t.x += bb; t.y += bb; t.z += bb; t.w += bb;
            const size_t off = ((size_t)b * COUT + o) * HW + (size_t)quad * 4;
            if (skip) {
                const float4 sk = *reinterpret_cast<const float4*>(skip + off);
                t.x += sk.x; t.y += sk.y; t.z += sk.z; t.w += sk.w;
            }
            *reinterpret_cast<float4*>(out + off) = t;
        }
    }
}

template <int COUT, int KS>
static int torgb_launch(float* out, const float* x, const float* W, const float* s, const float* bias, const float* skip, float scale,
                        int B, int Cin, int64_t HW, hipStream_t st)
{
    constexpr int CP = (COUT + 3) & ~3, QB = 256 / KS;
    const size_t lds = ((size_t)Cin * CP + (size_t)(KS - 1) * COUT * QB * 4) * sizeof(float);
    if (lds > 160 * 1024) return HAV_EUNSUP;
    // the dynamic-LDS attribute is per device: one bit per device id (a second GPU driven from the same process needs its own call;
    // a race here only repeats an idempotent call)
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (!((attr_mask.load(std::memory_order_acquire) >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute((const void*)torgb_kernel<COUT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_mask.fetch_or(1ull << dev, std::memory_order_release);
    }
    const dim3 grid((unsigned)(((HW >> 2) + QB - 1) / QB), (unsigned)B);
    hipLaunchKernelGGL((torgb_kernel<COUT, KS>), grid, dim3(256), lds, st, out, x, W, s, bias, skip, scale, Cin, HW);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_torgb(float* out, const float* x, const float* W, const float* s, const float* bias, const float* skip, float scale,
                         int B, int Cout, int Cin, int64_t HW, void* stream)
{
    if (!out || !x || !W || B < 1 || Cin < 1 || HW < 4) return HAV_EINVAL;
    if ((Cout != 3 && Cout != 12) || Cin > 1024 || (HW & 3)) return HAV_EUNSUP;
    // large maps: one lane per pixel quad walks all channels (no reduction); smaller maps: 4 or 16 channel slices per pixel quad, so that the
    // chain of dependent load rounds stays short (a 16^2 map with 512 channels is 8 rounds of 4 loads per lane instead of 128)
    const int64_t work = (HW >> 2) * B;
    const int ks = work >= 64 * 1024 && Cin <= 128 ? 1 : (work >= 4 * 1024 ? 4 : 16);
    hipStream_t st = (hipStream_t)stream;
#define HAV_TORGB(CO) (ks == 1 ? torgb_launch<CO, 1>(out, x, W, s, bias, skip, scale, B, Cin, HW, st) \
                               : (ks == 4 ? torgb_launch<CO, 4>(out, x, W, s, bias, skip, scale, B, Cin, HW, st) \
                                          : torgb_launch<CO, 16>(out, x, W, s, bias, skip, scale, B, Cin, HW, st)))
    return Cout == 3 ? HAV_TORGB(3) : HAV_TORGB(12);
#undef HAV_TORGB
}
