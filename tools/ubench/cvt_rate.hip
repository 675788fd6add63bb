// Issue rate of gfx950's packed block-scale conversions (the VALU side of the march kernel's "fp16 x 2 + MX" mode): how many cycles does a wave
// spend per v_cvt_scalef32_pk32_{fp6,bf6}_{f16,bf16} / v_cvt_scalef32_pk_fp4_f16 instruction, alone and next to a stream of matrix instructions
// of another wave on the same SIMD?  Decides whether the 6-bit A operands of the correction terms can be derived from the fp16 fragments on
// the fly (which frees 30 KB of LDS for the feature-parking weights) or must stay pre-converted in LDS.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/cvt_rate tools/ubench/cvt_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef _Float16 f16x32_t __attribute__((ext_vector_type(32)));
typedef __bf16 bf16x32_t __attribute__((ext_vector_type(32)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x6_t __attribute__((ext_vector_type(6)));
typedef unsigned int u32x16_t __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

// MODE 0: pk32_fp6_f16 | 1: pk32_bf6_f16 | 2: pk32_bf6_bf16 | 3: 16 x pk_fp4_f16 (= 32 values) | 4: v_fma_f32 reference (one per "instruction")
// MIX: waves with odd index run matrix instructions instead (2 waves per SIMD: 512 threads per workgroup)
template <int MODE, bool MIX>
__global__ void __launch_bounds__(512) cvt_kernel(unsigned int* out, int iters)
{
    u32x16_t src;
    for (int i = 0; i < 16; ++i) src[i] = 0x3c003800u + 0x00010001u * (threadIdx.x + 7 * i);
    const int wave = threadIdx.x >> 6;
    unsigned int acc = 0;
    if (MIX && (wave & 4)) {          // waves 4-7 share the SIMDs of waves 0-3
        v16f c0 = {}, c1 = {};
        v8h ha, hb;
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i); hb[i] = (_Float16)(0.02f * i); }
        for (int it = 0; it < iters; ++it) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
        out[blockIdx.x * 512 + threadIdx.x] = __float_as_uint(s);
        return;
    }
    float scale = 1.0f;
    float f = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) { const u32x6_t r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32_t, src), scale); acc ^= r[0] ^ r[5]; src[u] += r[3] & 1u; }
            else if (MODE == 1) { const u32x6_t r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(__builtin_bit_cast(f16x32_t, src), scale); acc ^= r[0] ^ r[5]; src[u] += r[3] & 1u; }
            else if (MODE == 2) { const u32x6_t r = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(__builtin_bit_cast(bf16x32_t, src), scale); acc ^= r[0] ^ r[5]; src[u] += r[3] & 1u; }
            else if (MODE == 3) {
                unsigned int r = 0, r2 = 0, r3 = 0, r4 = 0;
#define Q4(q)                                                                                                        \
                r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, __builtin_bit_cast(f16x2_t, src[q]), scale, q);          \
                r2 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r2, __builtin_bit_cast(f16x2_t, src[4 + q]), scale, q);    \
                r3 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r3, __builtin_bit_cast(f16x2_t, src[8 + q]), scale, q);    \
                r4 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r4, __builtin_bit_cast(f16x2_t, src[12 + q]), scale, q);
                Q4(0) Q4(1) Q4(2) Q4(3)
#undef Q4
                acc ^= r ^ r2 ^ r3 ^ r4; src[u] += r & 1u;
            } else { f = __builtin_fmaf(f, 1.0000001f, 1e-9f); }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc ^ __float_as_uint(f);
}

template <int MODE, bool MIX>
static int run(const char* name)
{
    const int iters = 20000, blocks = 256;
    unsigned int* dO;
    CK(hipMalloc(&dO, blocks * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((cvt_kernel<MODE, MIX>), dim3(blocks), dim3(512), 0, 0, dO, 500);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((cvt_kernel<MODE, MIX>), dim3(blocks), dim3(512), 0, 0, dO, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * 4;
    printf("  %-34s %8.3f ms: %6.1f ns per conversion of 32 values per wave (%s) = %5.1f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / n,
           MIX ? "1 converting wave + 1 matrix wave per SIMD" : "2 converting waves per SIMD", ms * 1e6 / n * 2.4 / (MIX ? 1 : 2));
    CK(hipFree(dO));
    return 0;
}

int main()
{
    printf("== conversion issue rate (256 workgroups x 512 threads; a dependent chain of 4 per iteration) ==\n");
    if (run<4, false>("v_fma_f32 (reference)")) return 1;
    if (run<0, false>("cvt_scalef32_pk32_fp6_f16")) return 1;
    if (run<1, false>("cvt_scalef32_pk32_bf6_f16")) return 1;
    if (run<2, false>("cvt_scalef32_pk32_bf6_bf16")) return 1;
    if (run<3, false>("16 x cvt_scalef32_pk_fp4_f16")) return 1;
    if (run<0, true>("pk32_fp6_f16 beside MFMA")) return 1;
    if (run<3, true>("16 x pk_fp4_f16 beside MFMA")) return 1;
    if (run<4, true>("v_fma_f32 beside MFMA")) return 1;
    return 0;
}
