// Issue cost of the march kernel's instruction mix on gfx950: cycles per instruction for one wave per SIMD (256 threads) and two
// (512 threads), per instruction kind.  Evidence for DESIGN.md 3.3 (the kernel is bound by instruction issue, not by bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters)
{
    __shared__ float4 lds[1024];
    lds[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    __syncthreads();
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    f2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f2{x[2 * i], x[2 * i + 1]};
    const f2 w = {1.0000001f, 1.0000001f};
    const float c = 1.0000001f, d = 1e-7f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 16; ++r) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], c, d);
            } else if (KIND == 1) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], w, p[i]);
            } else if (KIND == 2) {       // cvt_pk_f16_f32 + back
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    h2 hh = __builtin_convertvector(p[i], h2);
                    p[i] = p[i] - __builtin_convertvector(hh, f2);
                }
            } else if (KIND == 3) {       // v_max_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x[i]));
            } else if (KIND == 4) {       // ds_read_b128 broadcast + 1 add to consume
#pragma unroll
                for (int i = 0; i < 16; ++i) { float4 v = lds[(i * 7 + it) & 1023]; x[i] += v.x; }
            } else if (KIND == 5) {       // v_mov
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 15]));
            } else if (KIND == 7) {       // ONE dependent chain of v_fma_f32 (asm: the compiler must not re-associate)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c), "v"(d));
            } else if (KIND == 8) {       // TWO interleaved dependent chains
#pragma unroll
                for (int i = 0; i < 8; ++i) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(c), "v"(d)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(c), "v"(d)); }
            } else if (KIND == 9) {       // v_cmp -> v_cndmask pairs (VCC round trip), 8 independent
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = (x[i] > d) ? x[i + 8] : x[i] + c;
            } else if (KIND == 10) {      // 16 independent v_fma_f32 via asm (no SLP packing)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
            } else if (KIND == 6) {       // v_exp_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]);
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ((long long*)(out + 256 * 512))[0] = t1 - t0;
}
template <int KIND> void run(const char* name, float* d, int instr_per_rep)
{
    for (int threads = 256; threads <= 512; threads += 256) {
        const int iters = 2000;
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        long long h; hipMemcpy(&h, d + 256 * 512, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * (REP / 16) * instr_per_rep;
        printf("%-34s %d wave(s)/SIMD: %.2f cycles per instruction per wave  (%.2f per SIMD)\n", name, threads / 256, h / n, h / n / (threads / 256));
    }
}
int main()
{
    float* d; hipMalloc(&d, 256 * 512 * 4 + 64);
    run<0>("v_fma_f32 (16 indep.)", d, 16);
    run<1>("v_pk_fma_f32 (8 indep. x2)", d, 16);
    run<2>("fp16 split pair (cvt_pk,2cvt,sub)", d, 8 * 4);
    run<3>("v_max_f32", d, 16);
    run<4>("ds_read_b128 + v_add", d, 32);
    run<5>("v_mov_b32", d, 16);
    run<6>("v_exp_f32", d, 16);
    run<10>("v_fma_f32 asm (16 indep.)", d, 16);
    run<7>("v_fma_f32 ONE dependent chain", d, 16);
    run<8>("v_fma_f32 two dependent chains", d, 16);
    run<9>("cmp+add+cndmask (8 indep.)", d, 24);
    return 0;
}
