// Does VALU work of one wave overlap with MFMA work of the OTHER wave on the same SIMD -- per VALU instruction class?
// Round 1's mfma_valu_overlap.hip wrote its VALU stream as C (x = fmaf(x, c, d) on 8 chains); hipcc SLP-packs that into
// v_pk_fma_f32, and packed-f32 VALU is the one class that does NOT run beside the matrix pipe on gfx950 (MI355X_MICROARCH.md,
// "price of one filler beside MFMAs").  This version pins the instruction with inline asm.
// 512-thread workgroups (waves w and w+4 share a SIMD), one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

// VK: 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_pk_mul_f32, 3 = v_max_f32, 4 = v_cvt_pk_f16_f32, 5 = v_exp_f32
template <int VK>
__global__ void __launch_bounds__(512, 2) k(float* out, int mode, int n_mfma, int n_valu)
{
    const int wave = threadIdx.x >> 6;
    float res = 0.f;
    const long long t0 = clock64();
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            f16x8 ba, bb;
            for (int i = 0; i < 8; ++i) { ba[i] = (_Float16)(1.0f + i); bb[i] = (_Float16)(0.5f + threadIdx.x * 1e-3f); }
            for (int i = 0; i < n_mfma; i += 4) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ba, bb, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ba, bb, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ba, bb, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ba, bb, a3, 0, 0, 0);
            }
            res = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (mode & 2) {
        f2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, 1.0f + i};
        const f2 c = {1.0000001f, 1.0000001f}, d = {1e-7f, 1e-7f};
        for (int i = 0; i < n_valu; i += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q].x) : "v"(c.x), "v"(d.x));
                else if (VK == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(c), "v"(d));
                else if (VK == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[q]) : "v"(c));
                else if (VK == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[q].x) : "v"(d.x));
                else if (VK == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(x[q].y) : "v"(x[q].x), "v"(c.x));
                else asm volatile("v_exp_f32 %0, %1" : "+v"(x[q].y) : "v"(x[q].x));
            }
        }
        for (int i = 0; i < 8; ++i) res += x[i].x + x[i].y;
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ((long long*)(out + 256 * 512))[wave] = t1 - t0;
}

// same wave: FILL instructions of class VK between consecutive MFMAs (4 accumulators round-robin)
template <int VK, int FILL>
__global__ void __launch_bounds__(256, 1) kf(float* out, int n)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    f16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (_Float16)(1.0f + i); bb[i] = (_Float16)(0.5f + threadIdx.x * 1e-3f); }
    f2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, 1.0f + i};
    const f2 c = {1.0000001f, 1.0000001f}, d = {1e-7f, 1e-7f};
    const long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#define STEP(acc)                                                                                        \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ba, bb, acc, 0, 0, 0);                              \
        _Pragma("unroll") for (int q = 0; q < FILL; ++q) {                                               \
            if (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q % 8].x) : "v"(c.x), "v"(d.x)); \
            else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[q % 8]) : "v"(c), "v"(d));          \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3)
    }
    const long long t1 = clock64();
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 8; ++i) r += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) ((long long*)(out + 256 * 512))[0] = t1 - t0;
}

template <int VK> void run(float* d, const char* nm)
{
    const int nm_ = 80000, nv = 640000;      // 80000 MFMAs x 32 cyc = 2.56 M cycles; 640 k VALU x ~4.4 = 2.8 M
    float ms[3]; long long cyc[3][8];
    for (int mode = 1; mode <= 3; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<VK>, dim3(256), dim3(512), 0, 0, d, mode, nm_, nv);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<VK>, dim3(256), dim3(512), 0, 0, d, mode, nm_, nv);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms[mode - 1], a, b);
        hipMemcpy(cyc[mode - 1], d + 256 * 512, 64, hipMemcpyDeviceToHost);
    }
    printf("%-18s MFMA alone %.3f ms (%lld cyc) | VALU alone %.3f ms (%lld cyc) | together %.3f ms (MFMA wave %lld, VALU wave %lld cyc)\n", nm,
           ms[0], cyc[0][0], ms[1], cyc[1][4], ms[2], cyc[2][0], cyc[2][4]);
}
template <int VK, int FILL> void runf(float* d, const char* nm)
{
    const int n = 5000;
    hipLaunchKernelGGL((kf<VK, FILL>), dim3(256), dim3(256), 0, 0, d, n);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, d + 256 * 512, 8, hipMemcpyDeviceToHost);
    printf("same wave, %-13s fill=%2d : %.1f cycles per MFMA\n", nm, FILL, (double)c / (4.0 * n));
}
int main()
{
    float* d; hipMalloc(&d, 256 * 512 * 4 + 64);
    printf("== two waves per SIMD: waves 0-3 f16 32x32x16 MFMA stream, waves 4-7 VALU stream (inline asm)\n");
    run<0>(d, "v_fma_f32"); run<1>(d, "v_pk_fma_f32"); run<2>(d, "v_pk_mul_f32"); run<3>(d, "v_max_f32");
    run<4>(d, "v_cvt_pk_f16_f32"); run<5>(d, "v_exp_f32");
    printf("== one wave per SIMD: fillers between MFMAs of the same wave\n");
    runf<0, 0>(d, "v_fma_f32"); runf<0, 2>(d, "v_fma_f32"); runf<0, 4>(d, "v_fma_f32"); runf<0, 6>(d, "v_fma_f32"); runf<0, 8>(d, "v_fma_f32");
    runf<0, 12>(d, "v_fma_f32");
    runf<1, 1>(d, "v_pk_fma_f32"); runf<1, 2>(d, "v_pk_fma_f32"); runf<1, 4>(d, "v_pk_fma_f32"); runf<1, 6>(d, "v_pk_fma_f32");
    return 0;
}
