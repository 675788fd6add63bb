// Which SIMD does wave w of a 512-thread workgroup run on?  (The march kernel's matrix-core lock assumes waves w and w + 4 share one.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(unsigned* out)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, all 32 bits
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 256; ++b) {
        for (int w = 0; w < 4; ++w) {
            const unsigned s0 = (h[b * 8 + w] >> 4) & 3, s1 = (h[b * 8 + w + 4] >> 4) & 3;
            if (s0 != s1) ++bad;
        }
        for (int w = 0; w < 8; ++w) for (int v = w + 1; v < 8; ++v)
            if (v != w + 4 && ((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + v] >> 4) & 3)) ++bad;
    }
    for (int b = 0; b < 3; ++b) { printf("block %d: SIMD_ID (HW_ID bits 5:4) of waves 0..7:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("   raw %08x\n", h[b * 8]); }
    printf("workgroups where waves w and w+4 do NOT share a SIMD id, or two other waves do: %d violations in 256 workgroups\n", bad);
    return 0;
}
