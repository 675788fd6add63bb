// Does v_mfma_f32_32x32x16_f16 on gfx950 still read its A / B operand registers after it has issued?  (docs/history/DESIGN_r1-r4.md 3.5 inferred it from a
// run-to-run difference whose signature -- columns 16-31 of a tile -- round 3 traced to something else, docs/history/DESIGN_r1-r4.md 3.12.)
// Every wave runs ITER matrix instructions; right behind each one (same asm block, no wait states) a VALU instruction overwrites one
// register of the B operand (mode 1), of the A operand (mode 2), or nothing (mode 0); the operands are restored behind a long wait before
// the next round.  If the matrix instruction reads its operands late, the accumulators of modes 1 / 2 differ from mode 0.
// 512 threads per workgroup = two waves per SIMD, all issuing matrix instructions: the pipe is contended as in the march kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.25f + 0.01f * ((lane * 8 + e) % 17)); b[e] = (_Float16)(0.5f - 0.02f * ((lane * 5 + e) % 13)); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const f16x8 a0 = a, b0 = b;
    const unsigned* au = reinterpret_cast<const unsigned*>(&a0);
    const unsigned* bu = reinterpret_cast<const unsigned*>(&b0);
    // the operands live in fixed physical registers v[100:103] / v[104:107] inside ONE asm block, so that the overwrite really hits the
    // register the matrix instruction was given
#define MW_BODY(extra)                                                                                                             \
    asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\t"                        \
                 "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v106, %7\n\tv_mov_b32 v107, %8\n\ts_nop 7\n\t"               \
                 "v_mfma_f32_32x32x16_f16 %0, v[100:103], v[104:107], %0\n\t" extra "s_nop 15\n\ts_nop 15"                             \
                 : "+v"(acc) : "v"(au[0]), "v"(au[1]), "v"(au[2]), "v"(au[3]), "v"(bu[0]), "v"(bu[1]), "v"(bu[2]), "v"(bu[3])            \
                 : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107")
    // modes 3 / 4 / 5: THREE dependent matrix instructions back to back (same accumulator, as the hi.lo / lo.hi / hi.hi products of the
    // split kernels): the third one is issued while the second still runs.  3 = reference, 4 = B of the chain overwritten right behind the
    // third issue, 5 = A.
#define MW_CHAIN(extra)                                                                                                            \
    asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\t"                        \
                 "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v106, %7\n\tv_mov_b32 v107, %8\n\ts_nop 7\n\t"               \
                 "v_mfma_f32_32x32x16_f16 %0, v[100:103], v[104:107], %0\n\t"                                                         \
                 "v_mfma_f32_32x32x16_f16 %0, v[100:103], v[104:107], %0\n\t"                                                         \
                 "v_mfma_f32_32x32x16_f16 %0, v[100:103], v[104:107], %0\n\t" extra "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"       \
                 : "+v"(acc) : "v"(au[0]), "v"(au[1]), "v"(au[2]), "v"(au[3]), "v"(bu[0]), "v"(bu[1]), "v"(bu[2]), "v"(bu[3])            \
                 : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107")
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) { MW_CHAIN(""); continue; }
        if (MODE == 4) { MW_CHAIN("v_mov_b32 v107, 0x7bff7bff\n\tv_mov_b32 v104, 0x7bff7bff\n\t"); continue; }
        if (MODE == 5) { MW_CHAIN("v_mov_b32 v103, 0x7bff7bff\n\tv_mov_b32 v100, 0x7bff7bff\n\t"); continue; }
        if (MODE == 0) MW_BODY("");
        else if (MODE == 1) MW_BODY("v_mov_b32 v107, 0x7bff7bff\n\tv_mov_b32 v104, 0x7bff7bff\n\t");          // B, immediately behind the issue
        else MW_BODY("v_mov_b32 v103, 0x7bff7bff\n\tv_mov_b32 v100, 0x7bff7bff\n\t");                       // A
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}
int main()
{
    const int blocks = 1024, iters = 4000;
    float* d[3]; float* h[3];
    for (int m = 0; m < 3; ++m) { hipMalloc(&d[m], blocks * 512 * 4); h[m] = (float*)malloc(blocks * 512 * 4); }
    for (int rep = 0; rep < 3; ++rep) {          // dependent chains of three
        hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, d[0], iters);
        hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, d[1], iters);
        hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(512), 0, 0, d[2], iters);
        for (int m = 0; m < 3; ++m) hipMemcpy(h[m], d[m], blocks * 512 * 4, hipMemcpyDeviceToHost);
        long bad1 = 0, bad2 = 0; int l1[4] = {0, 0, 0, 0}, l2[4] = {0, 0, 0, 0};
        for (long i = 0; i < (long)blocks * 512; ++i) {
            if (h[1][i] != h[0][i]) { ++bad1; ++l1[(i & 63) >> 4]; }
            if (h[2][i] != h[0][i]) { ++bad2; ++l2[(i & 63) >> 4]; }
        }
        printf("chain rep %d: B overwritten behind the third issue: %ld of %ld lanes differ (by quarter %d %d %d %d);  A: %ld (by quarter %d %d %d %d)\n", rep,
               bad1, (long)blocks * 512, l1[0], l1[1], l1[2], l1[3], bad2, l2[0], l2[1], l2[2], l2[3]);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d[0], iters);
        hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d[1], iters);
        hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, d[2], iters);
        for (int m = 0; m < 3; ++m) hipMemcpy(h[m], d[m], blocks * 512 * 4, hipMemcpyDeviceToHost);
        long bad1 = 0, bad2 = 0; int l1[4] = {0, 0, 0, 0}, l2[4] = {0, 0, 0, 0};
        for (long i = 0; i < (long)blocks * 512; ++i) {
            if (h[1][i] != h[0][i]) { ++bad1; ++l1[(i & 63) >> 4]; }
            if (h[2][i] != h[0][i]) { ++bad2; ++l2[(i & 63) >> 4]; }
        }
        printf("rep %d: B overwritten behind the issue: %ld of %ld lanes differ (by quarter %d %d %d %d);  A: %ld (by quarter %d %d %d %d)\n", rep,
               bad1, (long)blocks * 512, l1[0], l1[1], l1[2], l1[3], bad2, l2[0], l2[1], l2[2], l2[3]);
    }
    return 0;
}
