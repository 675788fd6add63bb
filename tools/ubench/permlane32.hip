// What v_permlane32_swap_b32 does on gfx950, and whether a reader right behind it needs wait states.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out, int mode)
{
    const unsigned lane = threadIdx.x;
    unsigned a = 1000 + lane, b = 2000 + lane;
    if (mode == 0) {
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        out[lane] = r[0]; out[64 + lane] = r[1];
    } else {
        asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "+v"(b));
        out[lane] = a; out[64 + lane] = b;
    }
}
int main()
{
    unsigned* d; hipMalloc(&d, 128 * 4);
    unsigned h[128];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (a = 1000 + lane, b = 2000 + lane)\n  a': lane0 %u lane31 %u lane32 %u lane63 %u\n  b': lane0 %u lane31 %u lane32 %u lane63 %u\n", mode,
               h[0], h[31], h[32], h[63], h[64], h[95], h[96], h[127]);
    }
    return 0;
}
