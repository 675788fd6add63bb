// Probe of gfx950's block-scaled 6-/4-bit matrix path, which the march kernel's "fp16 x 2 + MX correction" arithmetic mode rests on
// (DESIGN.md 3.3):
//   T1  v_cvt_scalef32_pk32_{fp6,bf6}_f16 / v_cvt_scalef32_2xpk16_{fp6,bf6}_f32: element order inside the 6 result dwords, rounding,
//       saturation, and the direction of the scale (divide or multiply);
//   T2  v_mfma_scale_f32_32x32x64_f8f6f4 with A in {fp4, fp6, bf6} and B in {fp6, bf6}: which (lane, slot) of an operand is which
//       (row / column, k), how 4- and 6-bit elements are packed, per-lane E8M0 scales and the byte select (op_sel);
//   T3  cycles per instruction of the format pairs (4x6, 6x6, 8x8) against v_mfma_f32_32x32x16_f16.
// Every test prints what it found next to what the kernel assumes ("OK" / "MISMATCH"); the host re-computes the expected values from the
// raw operand bits, so a wrong assumption shows up as a count of differing elements plus a few examples.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mx6_probe tools/ubench/mx6_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned int v6u __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- host-side number formats
// fp6 e2m3 (bias 1), bf6 e3m2 (bias 3), fp4 e2m1 (bias 1): sign | exponent | mantissa, no inf / NaN (OCP MX)
static float dec_small(unsigned bits, int ebits, int mbits, int bias)
{
    const int w = 1 + ebits + mbits;
    const int s = (bits >> (w - 1)) & 1, e = (bits >> mbits) & ((1 << ebits) - 1), m = bits & ((1 << mbits) - 1);
    const float v = e == 0 ? ldexpf((float)m, 1 - bias - mbits) : ldexpf((float)((1 << mbits) + m), e - bias - mbits);
    return s ? -v : v;
}
static unsigned enc_small(float x, int ebits, int mbits, int bias)      // round to nearest even, saturating
{
    const int w = 1 + ebits + mbits;
    const unsigned sign = (x < 0.f || (x == 0.f && signbit(x))) ? 1u : 0u;
    const float ax = fabsf(x);
    unsigned best = 0; float bd = INFINITY;
    for (unsigned c = 0; c < (1u << (w - 1)); ++c) {
        const float d = fabsf(dec_small(c, ebits, mbits, bias) - ax);
        if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; }
    }
    return (sign << (w - 1)) | best;
}
struct Fmt { const char* name; int code, w, eb, mb, bias; };
static const Fmt FP6{"fp6(e2m3)", 2, 6, 2, 3, 1}, BF6{"bf6(e3m2)", 3, 6, 3, 2, 3}, FP4{"fp4(e2m1)", 4, 4, 2, 1, 1};
static unsigned get_bits(const uint32_t* regs, int elem, int w)
{
    const int bit = elem * w, d = bit >> 5, o = bit & 31;
    uint64_t two = regs[d];
    if (o + w > 32) two |= (uint64_t)regs[d + 1] << 32;
    return (unsigned)((two >> o) & ((1u << w) - 1));
}
static void put_bits(uint32_t* regs, int elem, int w, unsigned v)
{
    const int bit = elem * w, d = bit >> 5, o = bit & 31;
    regs[d] |= v << o;
    if (o + w > 32) regs[d + 1] |= v >> (32 - o);
}

// ---------------------------------------------------------------- T1: conversions
__global__ void cvt_kernel(uint32_t* out, const _Float16* src16, const float* src32, float scale)
{
    const int l = threadIdx.x;
    v32h hv; v16f f0, f1;
    for (int i = 0; i < 32; ++i) hv[i] = src16[l * 32 + i];
    for (int i = 0; i < 16; ++i) { f0[i] = src32[l * 32 + i]; f1[i] = src32[l * 32 + 16 + i]; }
    const v6u a = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hv, scale);
    const v6u b = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(hv, scale);
    const v6u c = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(f0, f1, scale);
    const v6u d = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(f0, f1, scale);
    for (int i = 0; i < 6; ++i) { out[(l * 4 + 0) * 6 + i] = a[i]; out[(l * 4 + 1) * 6 + i] = b[i]; out[(l * 4 + 2) * 6 + i] = c[i]; out[(l * 4 + 3) * 6 + i] = d[i]; }
}

static int test_cvt()
{
    printf("== T1: conversions ==\n");
    std::vector<_Float16> s16(64 * 32); std::vector<float> s32(64 * 32);
    srand(7);
    for (int i = 0; i < 64 * 32; ++i) {
        float v;
        const int kind = i % 8;
        if (kind == 0) v = 0.f;
        else if (kind == 1) v = ldexpf(1.f, (rand() % 9) - 5) * ((rand() & 1) ? 1.f : -1.f);
        else v = ((float)rand() / RAND_MAX * 2.f - 1.f) * ldexpf(1.f, (rand() % 8) - 3);
        if (i == 5) v = 100.f;             // saturation
        if (i == 6) v = -100.f;
        if (i == 7) v = 0.0625f * 1.5f;    // tie cases of the subnormal grid
        s16[i] = (_Float16)v; s32[i] = (float)s16[i];
    }
    _Float16* d16; float* d32; uint32_t* dout;
    CK(hipMalloc(&d16, s16.size() * 2)); CK(hipMalloc(&d32, s32.size() * 4)); CK(hipMalloc(&dout, 64 * 4 * 6 * 4));
    CK(hipMemcpy(d16, s16.data(), s16.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d32, s32.data(), s32.size() * 4, hipMemcpyHostToDevice));
    int bad_total = 0;
    const float scales[3] = {1.f, 4.f, 0.25f};
    for (int si = 0; si < 3; ++si) {
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dout, d16, d32, scales[si]);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> o(64 * 4 * 6);
        CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
        const char* names[4] = {"pk32_fp6_f16", "pk32_bf6_f16", "2xpk16_fp6_f32", "2xpk16_bf6_f32"};
        for (int v = 0; v < 4; ++v) {
            const Fmt& F = (v & 1) ? BF6 : FP6;
            // hypotheses: value = src / scale  |  value = src * scale ; element e at bits [6e, 6e+6); the f32 form takes its 32 inputs
            // INTERLEAVED: result element 2i = first vector's element i, element 2i+1 = second vector's element i
            int bad_div = 0, bad_mul = 0, shown = 0;
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 32; ++e) {
                    const float x = s32[l * 32 + (v >= 2 ? (e >> 1) + 16 * (e & 1) : e)];
                    const unsigned got = get_bits(&o[(l * 4 + v) * 6], e, 6);
                    const unsigned ed = enc_small(x / scales[si], F.eb, F.mb, F.bias), em = enc_small(x * scales[si], F.eb, F.mb, F.bias);
                    // -0 and +0 are both fine
                    auto same = [&](unsigned a, unsigned b) { return a == b || ((a & 31) == 0 && (b & 31) == 0); };
                    if (!same(got, ed)) { ++bad_div; if (si == 1 && shown < 4 && !same(got, ed) && !same(got, em)) { printf("   %s lane %d elem %d x=%g got %02x (%g) exp/ %02x (%g) exp* %02x\n", names[v], l, e, x, got, dec_small(got, F.eb, F.mb, F.bias), ed, dec_small(ed, F.eb, F.mb, F.bias), em); ++shown; } }
                    if (!same(got, em)) ++bad_mul;
                }
            printf("  %-16s scale %-5g : differs from cvt(x / scale) in %4d of 2048, from cvt(x * scale) in %4d   %s\n", names[v], scales[si], bad_div, bad_mul,
                   bad_div == 0 ? (v >= 2 ? "OK (divides, element 2i / 2i+1 = i-th of the first / second vector)" : "OK (divides, element e at bits 6e)") : (bad_mul == 0 ? "MULTIPLIES" : "MISMATCH"));
            if (bad_div) ++bad_total;
        }
    }
    return bad_total;
}

// ---------------------------------------------------------------- T2: the scaled matrix instruction
// one wave per workgroup, one instruction; operands and scales come from memory as the host laid them out
template <int CBSZ, int BLGP>
__global__ void mfma_kernel(float* out, const uint32_t* A, const uint32_t* B, const uint32_t* SA, const uint32_t* SB)
{
    const int l = threadIdx.x, w = blockIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)A[(w * 64 + l) * 8 + i]; b[i] = (int)B[(w * 64 + l) * 8 + i]; }
    const int sa = (int)SA[w * 64 + l], sb = (int)SB[w * 64 + l];
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, BLGP, 1 /* op_sel A: byte 1 */, sa, 2 /* op_sel B: byte 2 */, sb);
    for (int i = 0; i < 16; ++i) out[(w * 64 + l) * 16 + i] = c[i];
}

template <int CBSZ, int BLGP>
static int test_mfma(const Fmt& FA, const Fmt& FB)
{
    const int NW = 8;
    std::vector<float> Am(NW * 32 * 64), Bm(NW * 64 * 32);
    std::vector<uint32_t> Ar(NW * 64 * 8, 0), Br(NW * 64 * 8, 0), SA(NW * 64), SB(NW * 64);
    std::vector<int> ea(NW * 64), eb(NW * 64);
    srand(11 + CBSZ * 7 + BLGP);
    const float vals[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, -1.f, -0.5f};         // exactly representable in all three formats
    for (int w = 0; w < NW; ++w)
        for (int l = 0; l < 64; ++l) {
            const int i = l & 31, kb = l >> 5;
            ea[w * 64 + l] = (rand() % 5) - 2; eb[w * 64 + l] = (rand() % 5) - 2;      // per-lane block exponents
            // scale VGPR: four different bytes; the instruction is told to use byte 1 (A) / byte 2 (B)
            SA[w * 64 + l] = 0x7F | ((uint32_t)(127 + ea[w * 64 + l]) << 8) | (0x85u << 16) | (0x70u << 24);
            SB[w * 64 + l] = 0x7E | (0x83u << 8) | ((uint32_t)(127 + eb[w * 64 + l]) << 16) | (0x72u << 24);
            for (int s = 0; s < 32; ++s) {
                const float va = vals[rand() % 8], vb = vals[rand() % 8];
                const int k = 32 * kb + s;
                Am[(w * 32 + i) * 64 + k] = va * ldexpf(1.f, ea[w * 64 + l]);
                Bm[(w * 64 + k) * 32 + i] = vb * ldexpf(1.f, eb[w * 64 + l]);
                put_bits(&Ar[(w * 64 + l) * 8], s, FA.w, enc_small(va, FA.eb, FA.mb, FA.bias));
                put_bits(&Br[(w * 64 + l) * 8], s, FB.w, enc_small(vb, FB.eb, FB.mb, FB.bias));
            }
        }
    uint32_t *dA, *dB, *dSA, *dSB; float* dO;
    CK(hipMalloc(&dA, Ar.size() * 4)); CK(hipMalloc(&dB, Br.size() * 4)); CK(hipMalloc(&dSA, SA.size() * 4)); CK(hipMalloc(&dSB, SB.size() * 4));
    CK(hipMalloc(&dO, NW * 64 * 16 * 4));
    CK(hipMemcpy(dA, Ar.data(), Ar.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Br.data(), Br.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dSA, SA.data(), SA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dSB, SB.data(), SB.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((mfma_kernel<CBSZ, BLGP>), dim3(NW), dim3(64), 0, 0, dO, dA, dB, dSA, dSB);
    CK(hipDeviceSynchronize());
    std::vector<float> O(NW * 64 * 16);
    CK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, shown = 0;
    for (int w = 0; w < NW; ++w)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double s = 0.0;
                for (int k = 0; k < 64; ++k) s += (double)Am[(w * 32 + row) * 64 + k] * (double)Bm[(w * 64 + k) * 32 + col];
                const float got = O[(w * 64 + l) * 16 + r];
                if (fabs((double)got - s) > 1e-4 * (1.0 + fabs(s))) { ++bad; if (shown < 6) { printf("   wave %d lane %d reg %d (row %d col %d): got %g expected %g\n", w, l, r, row, col, got, s); ++shown; } }
            }
    printf("  A %-10s x B %-10s: %5d of %d elements differ   %s\n", FA.name, FB.name, bad, NW * 64 * 16,
           bad == 0 ? "OK (lane = row|col + 32 k-block, slot s = k & 31 at bits w*s, E8M0 byte per lane via op_sel)" : "MISMATCH");
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dSA)); CK(hipFree(dSB)); CK(hipFree(dO));
    return bad ? 1 : 0;
}

// one-hot scan, only printed when a layout test failed: where does element (lane la, slot sa) of A meet element (lane lb, slot sb) of B?
template <int CBSZ, int BLGP>
static void scan_layout(const Fmt& FA, const Fmt& FB)
{
    printf("  one-hot scan A %s x B %s (A one-hot at (lane, slot); list of B (lane, slot) that produce a non-zero, and where)\n", FA.name, FB.name);
    const int NW = 64 * 32;
    const int probes[6][2] = {{0, 0}, {0, 1}, {0, 5}, {1, 0}, {32, 0}, {33, 7}};
    std::vector<uint32_t> Ar(NW * 64 * 8), Br(NW * 64 * 8, 0), S(NW * 64, 0x7F7F7F7Fu);
    uint32_t *dA, *dB, *dS; float* dO;
    CK(hipMalloc(&dA, Ar.size() * 4)); CK(hipMalloc(&dB, Br.size() * 4)); CK(hipMalloc(&dS, S.size() * 4)); CK(hipMalloc(&dO, (size_t)NW * 64 * 16 * 4));
    for (int w = 0; w < NW; ++w) put_bits(&Br[(w * 64 + (w >> 5)) * 8], w & 31, FB.w, enc_small(1.f, FB.eb, FB.mb, FB.bias));
    CK(hipMemcpy(dB, Br.data(), Br.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dS, S.data(), S.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> O((size_t)NW * 64 * 16);
    for (int p = 0; p < 6; ++p) {
        std::fill(Ar.begin(), Ar.end(), 0u);
        for (int w = 0; w < NW; ++w) put_bits(&Ar[(w * 64 + probes[p][0]) * 8], probes[p][1], FA.w, enc_small(1.f, FA.eb, FA.mb, FA.bias));
        CK(hipMemcpy(dA, Ar.data(), Ar.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((mfma_kernel<CBSZ, BLGP>), dim3(NW), dim3(64), 0, 0, dO, dA, dB, dS, dS);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
        printf("   A(lane %d, slot %d):", probes[p][0], probes[p][1]);
        int n = 0;
        for (int w = 0; w < NW && n < 40; ++w)
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r)
                    if (O[((size_t)w * 64 + l) * 16 + r] != 0.f && n < 40) { printf(" B(%d,%d)->out(lane %d,reg %d)=%g", w >> 5, w & 31, l, r, O[((size_t)w * 64 + l) * 16 + r]); ++n; }
        printf("\n");
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dS)); CK(hipFree(dO));
}

// ---------------------------------------------------------------- T3: issue rate
template <int MODE>      // 0: f16 32x32x16 | 1: fp4 x fp6 | 2: fp6 x fp6 | 3: bf6 x bf6 | 4: fp8 x fp8 | 5: fp4 x bf6 | 6: fp4 x fp4
__global__ void __launch_bounds__(256) rate_kernel(float* out, unsigned long long* cyc, int iters)
{
    v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x11111111 * (i + 1) + threadIdx.x; b[i] = 0x01010101 * (i + 3) ^ threadIdx.x; }
    v8h ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * i); hb[i] = (_Float16)(0.02f * i); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STEP(cx)                                                                                                         \
        if (MODE == 0) cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, cx, 0, 0, 0);                                     \
        else if (MODE == 1) cx = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cx, 4, 2, 0, 127, 0, 127);          \
        else if (MODE == 2) cx = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cx, 2, 2, 0, 127, 0, 127);          \
        else if (MODE == 3) cx = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cx, 3, 3, 0, 127, 0, 127);          \
        else if (MODE == 4) cx = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cx, 0, 0, 0, 127, 0, 127);          \
        else if (MODE == 5) cx = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cx, 4, 3, 0, 127, 0, 127);          \
        else cx = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, cx, 4, 4, 0, 127, 0, 127);
        STEP(c0) STEP(c1) STEP(c2) STEP(c3)
#undef STEP
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void test_rate(const char* name)
{
    const int iters = 20000, blocks = 256;
    float* dO; unsigned long long* dC;
    CK(hipMalloc(&dO, blocks * 256 * 4)); CK(hipMalloc(&dC, blocks * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, dO, dC, 1000);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_kernel<MODE>), dim3(blocks), dim3(256), 0, 0, dO, dC, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * 4;            // instructions per wave; one wave per SIMD (256 threads, one workgroup per CU)
    const double flop = (MODE == 0 ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 64 * 2) * n * 4 * blocks;
    printf("  %-12s %8.3f ms for %d x 4 instructions per wave, one wave per SIMD: %6.1f ns per instruction = %5.1f cycles at 2.4 GHz; %7.1f TFLOP/s\n",
           name, ms, iters, ms * 1e6 / n, ms * 1e6 / n * 2.4, flop / (ms * 1e-3) * 1e-12);
    CK(hipFree(dO)); CK(hipFree(dC));
}

// ---------------------------------------------------------------- T4: does the scaled instruction read its operands after issue?
// (tools/ubench/mfma_war.hip found that v_mfma_f32_32x32x16_f16 does not; the compiler assumes the same here and lets the very next VALU
// instruction recycle an operand register.)  Every wave runs `iters` rounds of: operands into FIXED registers, the matrix instruction, NOPS
// wait states, a v_mov that overwrites one operand register (WHICH: 0 nothing | 1 A's first | 2 A's last | 3 B's first | 4 B's last |
// 5 the scale register), a long drain, restore.  A result that differs from WHICH = 0 means the instruction was still reading.
template <int WHICH, int NOPS>
__global__ void __launch_bounds__(512) war_kernel(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    unsigned a[6], b[6];
    for (int i = 0; i < 6; ++i) { a[i] = 0x11111111u * (i + 1) ^ (lane * 0x01010101u); b[i] = 0x0f0f0f0fu * (i + 2) ^ (lane * 0x00110011u); }
    a[0] &= 0x1f7df7dfu; b[0] &= 0x1f7df7dfu;
    const unsigned sc = 0x7f7f7f7fu;
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#define WAR_NOP(n) (n == 0 ? "" : n == 1 ? "s_nop 0\n\t" : n == 2 ? "s_nop 1\n\t" : n == 4 ? "s_nop 3\n\t" : n == 8 ? "s_nop 7\n\t" : "s_nop 15\n\t")
    for (int it = 0; it < iters; ++it) {
#define WAR_BODY(nops, clobber)                                                                                                                \
        asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\tv_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\t"   \
                     "v_mov_b32 v108, %7\n\tv_mov_b32 v109, %8\n\tv_mov_b32 v110, %9\n\tv_mov_b32 v111, %10\n\tv_mov_b32 v112, %11\n\tv_mov_b32 v113, %12\n\t" \
                     "v_mov_b32 v116, %13\n\ts_nop 7\n\t"                                                                                        \
                     "v_mfma_scale_f32_32x32x64_f8f6f4 %0, v[100:105], v[108:113], %0, v116, v116 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"           \
                     nops clobber "s_nop 15\n\ts_nop 15\n\ts_nop 15"                                                                            \
                     : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),  \
                       "v"(b[4]), "v"(b[5]), "v"(sc)                                                                                            \
                     : "v100", "v101", "v102", "v103", "v104", "v105", "v108", "v109", "v110", "v111", "v112", "v113", "v116")
#define WAR_N(clobber) do { if (NOPS == 0) WAR_BODY("", clobber); else if (NOPS == 1) WAR_BODY("s_nop 0\n\t", clobber); else if (NOPS == 2) WAR_BODY("s_nop 1\n\t", clobber); \
                            else if (NOPS == 4) WAR_BODY("s_nop 3\n\t", clobber); else if (NOPS == 8) WAR_BODY("s_nop 7\n\t", clobber); else WAR_BODY("s_nop 15\n\t", clobber); } while (0)
        if (WHICH == 0) WAR_N("");
        else if (WHICH == 1) WAR_N("v_mov_b32 v100, 0x12345678\n\t");
        else if (WHICH == 2) WAR_N("v_mov_b32 v105, 0x12345678\n\t");
        else if (WHICH == 3) WAR_N("v_mov_b32 v108, 0x12345678\n\t");
        else if (WHICH == 4) WAR_N("v_mov_b32 v113, 0x12345678\n\t");
        else WAR_N("v_mov_b32 v116, 0x75757575\n\t");
        for (int r = 0; r < 16; ++r) acc[r] *= 0.5f;
    }
    for (int r = 0; r < 16; ++r) out[((size_t)blockIdx.x * 512 + threadIdx.x) * 16 + r] = acc[r];
}
template <int WHICH, int NOPS>
static long war_run(std::vector<float>& ref, const char* what)
{
    const int blocks = 256, iters = 400;
    float* d; CK(hipMalloc(&d, (size_t)blocks * 512 * 16 * 4));
    hipLaunchKernelGGL((war_kernel<WHICH, NOPS>), dim3(blocks), dim3(512), 0, 0, d, iters);
    CK(hipDeviceSynchronize());
    std::vector<float> o((size_t)blocks * 512 * 16);
    CK(hipMemcpy(o.data(), d, o.size() * 4, hipMemcpyDeviceToHost));
    CK(hipFree(d));
    long bad = 0;
    if (WHICH == 0 && ref.empty()) ref = o;
    else for (size_t i = 0; i < o.size(); ++i) bad += memcmp(&o[i], &ref[i], 4) != 0;
    printf("  overwrite %-22s after %2d wait state(s): %8ld of %zu results differ%s\n", what, NOPS, bad, o.size(), WHICH == 0 ? " (reference run / repeat)" : "");
    return bad;
}
static void test_war()
{
    printf("== T4: operand registers of v_mfma_scale_f32_32x32x64_f8f6f4 overwritten behind the instruction (2 waves per SIMD, 256 workgroups x 400 rounds) ==\n");
    std::vector<float> ref;
    war_run<0, 0>(ref, "nothing"); war_run<0, 0>(ref, "nothing");
    war_run<1, 0>(ref, "A, first register"); war_run<1, 1>(ref, "A, first register"); war_run<1, 2>(ref, "A, first register"); war_run<1, 4>(ref, "A, first register");
    war_run<1, 8>(ref, "A, first register"); war_run<1, 16>(ref, "A, first register");
    war_run<2, 0>(ref, "A, last register"); war_run<2, 1>(ref, "A, last register"); war_run<2, 2>(ref, "A, last register"); war_run<2, 4>(ref, "A, last register");
    war_run<2, 8>(ref, "A, last register"); war_run<2, 16>(ref, "A, last register");
    war_run<3, 0>(ref, "B, first register"); war_run<3, 2>(ref, "B, first register"); war_run<3, 4>(ref, "B, first register"); war_run<3, 8>(ref, "B, first register");
    war_run<4, 0>(ref, "B, last register"); war_run<4, 2>(ref, "B, last register"); war_run<4, 4>(ref, "B, last register"); war_run<4, 8>(ref, "B, last register");
    war_run<4, 16>(ref, "B, last register");
    war_run<5, 0>(ref, "the scale register"); war_run<5, 2>(ref, "the scale register"); war_run<5, 8>(ref, "the scale register");
}

int main()
{
    int bad = test_cvt();
    printf("== T2: v_mfma_scale_f32_32x32x64_f8f6f4 operand layout ==\n");
    int b;
    b = test_mfma<4, 2>(FP4, FP6); if (b) scan_layout<4, 2>(FP4, FP6); bad += b;
    b = test_mfma<2, 2>(FP6, FP6); if (b) scan_layout<2, 2>(FP6, FP6); bad += b;
    b = test_mfma<4, 3>(FP4, BF6); bad += b;
    b = test_mfma<3, 3>(BF6, BF6); bad += b;
    b = test_mfma<2, 3>(FP6, BF6); bad += b;
    printf("== T3: issue rate ==\n");
    test_rate<0>("f16 x16"); test_rate<1>("fp4 x fp6"); test_rate<2>("fp6 x fp6"); test_rate<3>("bf6 x bf6"); test_rate<5>("fp4 x bf6"); test_rate<6>("fp4 x fp4"); test_rate<4>("fp8 x fp8");
    test_war();
    printf(bad ? "RESULT: %d tests disagree with the assumed semantics\n" : "RESULT: all assumptions hold\n", bad);
    return 0;
}
