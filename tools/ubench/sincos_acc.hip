// Accuracy of v_sin_f32 / v_cos_f32 (inputs in revolutions) after an exact-ish Cody-Waite reduction, against the polynomial
// pe_pair of hav_render.hip, for the positional-encoding angles x = p * 2^k, |p| <= 1.6, k = 0..7.  Reference: double sin/cos of the
// fp32 angle (what torch.sin rounds from).  Decides whether the hardware transcendentals can replace the polynomials (docs/history/DESIGN_r1-r4.md 3.4).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ void pe_poly(float x, float& s_out, float& c_out)
{
    const float n = rintf(x * 0.63661977236758134f);
    float r = fmaf(-n, 1.5707964f, x);
    r = fmaf(-n, -4.371139e-08f, r);
    r = fmaf(-n, -1.7763568e-15f, r);
    const float s = r * r;
    float p = fmaf(s, 2.7158228022017283e-06f, -0.00019839018932543695f);
    p = fmaf(s, p, 0.008333328180015087f);
    p = fmaf(s, p, -0.1666666716337204f);
    const float sn = fmaf(r * s, p, r);
    float q = fmaf(s, -2.7208204755879706e-07f, 2.479949216649402e-05f);
    q = fmaf(s, q, -0.0013888883404433727f);
    q = fmaf(s, q, 0.0416666679084301f);
    const float cs = fmaf(s * s, q, fmaf(s, -0.5f, 1.0f));
    const int k = (int)n;
    const float a = (k & 1) ? cs : sn, b = (k & 1) ? sn : cs;
    s_out = (k & 2) ? -a : a;
    c_out = ((k + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void pe_hw(float x, float& s_out, float& c_out)
{
    // n = nearest integer number of turns; r = x - n * 2pi in two exact-product steps (n <= 64: 2pi_hi has 17 significant bits)
    const float n = rintf(x * 0.15915494309189535f);
    float r = fmaf(-n, 6.28314208984375f, x);           // 2pi truncated to 17 significant bits (0x40C90F80): n * hi is exact for n < 128
    r = fmaf(-n, 4.321733649703674e-05f, r);            // 2pi - hi (residual 6.6e-13)
    const float t = r * 0.15915494309189535f;           // |t| <= 0.5 revolutions
    s_out = __builtin_amdgcn_sinf(t);
    c_out = __builtin_amdgcn_cosf(t);
}
__global__ void k(const float* x, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pe_poly(x[i], out[4 * i + 0], out[4 * i + 1]);
    pe_hw(x[i], out[4 * i + 2], out[4 * i + 3]);
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> hx(n), ho(4 * (size_t)n);
    srand(1);
    for (int i = 0; i < n; ++i) { const float p = 3.2f * ((float)rand() / RAND_MAX) - 1.6f; hx[i] = p * (float)(1 << (i & 7)); }
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 16 * (size_t)n);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho.data(), dout, 16 * (size_t)n, hipMemcpyDeviceToHost);
    double e[4] = {0, 0, 0, 0}, eo[8][2] = {};
    for (int i = 0; i < n; ++i) {
        const double s = sin((double)hx[i]), c = cos((double)hx[i]);
        const double d[4] = {fabs(ho[4 * i] - s), fabs(ho[4 * i + 1] - c), fabs(ho[4 * i + 2] - s), fabs(ho[4 * i + 3] - c)};
        for (int q = 0; q < 4; ++q) if (d[q] > e[q]) e[q] = d[q];
        if (d[2] > eo[i & 7][0]) eo[i & 7][0] = d[2];
        if (d[3] > eo[i & 7][1]) eo[i & 7][1] = d[3];
    }
    printf("max abs error vs double sin/cos of the fp32 angle, %d angles:\n  polynomial  sin %.3e  cos %.3e\n  v_sin/v_cos sin %.3e  cos %.3e\n", n, e[0], e[1], e[2], e[3]);
    for (int o = 0; o < 8; ++o) printf("  octave %d (|x| <= %6.1f): hw sin %.3e cos %.3e\n", o, 1.6 * (1 << o), eo[o][0], eo[o][1]);
    return 0;
}
