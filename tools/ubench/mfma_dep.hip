// Issue interval of v_mfma_f32_32x32x16_bf16 when consecutive MFMAs accumulate into the SAME registers (dependent chain)
// vs 2 / 4 independent accumulators; one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void __launch_bounds__(256, 1) k(float* out, int n)
{
    f32x16 a[4] = {{0}, {0}, {0}, {0}};
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (short)(0x3f80 + i); bb[i] = (short)(0x3f00 + threadIdx.x); }
    const long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int q = 0; q < 12; ++q) a[q % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a[q % NACC], 0, 0, 0);
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    if (blockIdx.x == 0 && threadIdx.x == 0) ((long long*)(out + 256 * 256))[0] = t1 - t0;
}
template <int NACC> void run(float* d)
{
    const int n = 4000;
    hipLaunchKernelGGL((k<NACC>), dim3(256), dim3(256), 0, 0, d, n);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, d + 256 * 256, 8, hipMemcpyDeviceToHost);
    printf("bf16 32x32x16, %d accumulator(s) in rotation: %.1f cycles per MFMA\n", NACC, (double)c / (12.0 * n));
}
int main() { float* d; hipMalloc(&d, 256 * 256 * 4 + 64); run<1>(d); run<2>(d); run<3>(d); run<4>(d); return 0; }
