// Does VALU work of one wave overlap with MFMA work of the other wave on the same SIMD?
// 512-thread workgroups (waves w and w+4 share a SIMD), one workgroup per CU.
// mode bit0: waves 0-3 run MFMAs; bit1: waves 4-7 run VALU fma chains.  kind: 0 = f32 32x32x2, 1 = bf16 32x32x16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(512, 2) k(float* out, int mode, int n_mfma, int n_valu)
{
    int wave = threadIdx.x >> 6;
    if (mode & 4) wave ^= 4;                 // swap roles: VALU on the older waves 0-3
    if ((mode & 8) && wave >= 4) __builtin_amdgcn_s_setprio(3);   // VALU waves get priority
    if ((mode & 16) && wave < 4) __builtin_amdgcn_s_setprio(3);   // MFMA waves get priority
    float res = 0.f;
    const long long t0 = clock64();
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            float fa = threadIdx.x * 1e-3f, fb = 1.0f + threadIdx.x * 1e-4f;
            bf16x8 ba, bb;
            for (int i = 0; i < 8; ++i) { ba[i] = (short)(0x3f80 + i); bb[i] = (short)(0x3f00 + threadIdx.x); }
            for (int i = 0; i < n_mfma; i += 4) {
                if (KIND == 0) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a3, 0, 0, 0);
                }
            }
            res = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else {
        if (mode & 2) {
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
            const float c = 1.0000001f, d = 1e-7f;
            for (int i = 0; i < n_valu; i += 8) {
                x0 = fmaf(x0, c, d); x1 = fmaf(x1, c, d); x2 = fmaf(x2, c, d); x3 = fmaf(x3, c, d);
                x4 = fmaf(x4, c, d); x5 = fmaf(x5, c, d); x6 = fmaf(x6, c, d); x7 = fmaf(x7, c, d);
            }
            res = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ((long long*)(out + 256 * 512))[wave] = t1 - t0;
}

template <int KIND> float run(float* d, int mode, int nm, int nv)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, mode, nm, nv);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, mode, nm, nv);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[8]; hipMemcpy(h, d + 256 * 512, 64, hipMemcpyDeviceToHost);
    printf("   per-wave cycles:"); for (int i = 0; i < 8; ++i) printf(" %lld", h[i]); printf("\n");
    return ms;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 512 * 4 + 64);
    const int nm = 40000, nv = 1280000;   // 40000 f32 MFMAs x 64 cyc = 2.56M cyc ; 1.28M fma x 2 cyc = 2.56M cyc
    printf("f32 MFMA only      : %.3f ms\n", run<0>(d, 1, nm, nv));
    printf("VALU only          : %.3f ms\n", run<0>(d, 2, nm, nv));
    printf("f32 MFMA + VALU    : %.3f ms\n", run<0>(d, 3, nm, nv));
    printf("f32 swap roles     : %.3f ms\n", run<0>(d, 3 + 4, nm, nv));
    printf("f32 VALU prio      : %.3f ms\n", run<0>(d, 3 + 8, nm, nv));
    printf("f32 MFMA prio      : %.3f ms\n", run<0>(d, 3 + 16, nm, nv));
    printf("f32 swap+VALU prio : %.3f ms\n", run<0>(d, 3 + 4 + 8, nm, nv));
    const int nb = nm * 2;                // bf16 32x32x16: 32 cyc each
    printf("bf16 MFMA only     : %.3f ms\n", run<1>(d, 1, nb, nv));
    printf("bf16 MFMA + VALU   : %.3f ms\n", run<1>(d, 3, nb, nv));
    printf("bf16 VALU prio     : %.3f ms\n", run<1>(d, 3 + 8, nb, nv));
    printf("bf16 swap roles    : %.3f ms\n", run<1>(d, 3 + 4, nb, nv));
    return 0;
}
