// How many independent VALU instructions fit in the shadow of one MFMA when interleaved IN THE SAME WAVE?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int FILL>
__global__ void __launch_bounds__(256, 1) k(float* out, int n)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float fa = threadIdx.x * 1e-3f, fb = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (short)(0x3f80 + i); bb[i] = (short)(0x3f00 + threadIdx.x); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
    const float c = 1.0000001f, d = 1e-7f;
    const long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#define STEP(acc)                                                                              \
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);       \
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc, 0, 0, 0);              \
        _Pragma("unroll") for (int q = 0; q < FILL; ++q) x[q % 16] = fmaf(x[q % 16], c, d);    \
        __builtin_amdgcn_sched_barrier(0);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3)
    }
    const long long t1 = clock64();
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 16; ++i) r += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) ((long long*)(out + 256 * 256))[0] = t1 - t0;
}
template <int KIND, int FILL> void run(float* d, const char* nm)
{
    const int n = 5000;
    hipLaunchKernelGGL((k<KIND, FILL>), dim3(256), dim3(256), 0, 0, d, n);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, d + 256 * 256, 8, hipMemcpyDeviceToHost);
    printf("%s fill=%2d : %.1f cycles per MFMA\n", nm, FILL, (double)c / (4.0 * n));
}
int main()
{
    float* d; hipMalloc(&d, 256 * 256 * 4 + 64);
    run<0, 0>(d, "f32 32x32x2 "); run<0, 4>(d, "f32 32x32x2 "); run<0, 8>(d, "f32 32x32x2 "); run<0, 12>(d, "f32 32x32x2 ");
    run<0, 16>(d, "f32 32x32x2 "); run<0, 24>(d, "f32 32x32x2 "); run<0, 32>(d, "f32 32x32x2 ");
    run<1, 0>(d, "bf16 32x32x16"); run<1, 2>(d, "bf16 32x32x16"); run<1, 4>(d, "bf16 32x32x16"); run<1, 6>(d, "bf16 32x32x16");
    run<1, 8>(d, "bf16 32x32x16"); run<1, 12>(d, "bf16 32x32x16"); run<1, 16>(d, "bf16 32x32x16");
    return 0;
}
