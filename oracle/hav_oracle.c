/*
 * hav_oracle.c -- CPU restatement of the reference's algorithm for the hot path.
 *
 * *** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library,
 * and only as the checker / the timed CPU baseline.  Nothing under havatar_amd/ imports it; the
 * product path fails loudly when libhavatar_hip.so is missing instead of falling back to this.
 *
 * What it restates (XChenZ/havatar @ 2024_08_07, file:line):
 *   - fused_bias_act            model/op/fused_bias_act_kernel.cu:18-65  (CPU twin: model/op/fused_act.py:108-119)
 *   - upfirdn2d                 model/op/upfirdn2d_kernel.cu:49-105      (CPU twin: model/op/upfirdn2d.py:172-213)
 *   - predict_and_render_radiance and callees: see hav_oracle_impl.h
 *   - eval_sh                   utils/sh_util.py:55-107  (inactive on the path: sh_deg=0, kept for API completeness)
 *   - get_rays                  dataloader/data_util.py:28-56
 *
 * Parity pin: the reference ships NO tests, golden vectors or known-answer fixtures for this path
 * (SURVEY.md section 4).  The oracle is therefore pinned against outputs of the reference itself,
 * imported in the build container by oracle/gen_golden.py, which writes the .npz vectors under tests/golden;
 * tests/test_oracle_golden.py checks this file against those vectors.
 *
 * Build: make -C oracle   (gcc -O3 -mavx2 -mfma -fopenmp -shared -> oracle/_build/libhav_oracle.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/havatar.h"

typedef struct OrcDebug {          /* optional per-sample dumps (element type = REAL of the variant called) */
    void* z_coarse;   /* [B*R,S_c]      */
    void* w_coarse;   /* [B*R,S_c]      */
    void* z_fine;     /* [B*R,S_fp]     */
    void* w_fine;     /* [B*R,S_fp]     */
    void* raw_coarse; /* [B*R,S_c,68]   MLP output before sigmoid: rgb3 | feat64 | alpha */
    void* raw_fine;   /* [B*R,S_fp,68]  */
} OrcDebug;

#define REAL float
#define SUF(x) x##_f32
#include "hav_oracle_impl.h"
#undef REAL
#undef SUF
#define REAL double
#define SUF(x) x##_f64
#include "hav_oracle_impl.h"
#undef REAL
#undef SUF

/* ---- fused_bias_act (model/op/fused_bias_act_kernel.cu:33-63) ------------------------------- */
#define DEF_FBA(NAME, T)                                                                                    \
    int NAME(T* out, const T* x, const T* b, const T* ref, int act, int grad, T alpha, T scale,             \
             int64_t size_x, int64_t step_b, int64_t size_b)                                                \
    {                                                                                                       \
        if (size_x < 0 || (b && (step_b <= 0 || size_b <= 0))) return HAV_EINVAL;                           \
        for (int64_t i = 0; i < size_x; ++i) {                                                              \
            T v = x[i];                                                                                     \
            if (b) v += b[(i / step_b) % size_b];                                                           \
            T r = ref ? ref[i] : (T)0;                                                                      \
            T y;                                                                                            \
            switch (act * 10 + grad) {                                                                      \
            default: case 10: case 11: y = v; break;                                                        \
            case 12: y = 0; break;                                                                          \
            case 30: y = (v > 0) ? v : v * alpha; break;                                                    \
            case 31: y = (r > 0) ? v : v * alpha; break;                                                    \
            case 32: y = 0; break;                                                                          \
            }                                                                                               \
            out[i] = y * scale;                                                                             \
        }                                                                                                   \
        return 0;                                                                                           \
    }
DEF_FBA(orc_fused_bias_act_f32, float)
DEF_FBA(orc_fused_bias_act_f64, double)

/* ---- upfirdn2d (generic kernel, model/op/upfirdn2d_kernel.cu:49-105; definition SURVEY A-9) ------ */
#define DEF_UFD(NAME, T, ACC)                                                                               \
    int NAME(T* out, const T* in, const float* k, int64_t major, int in_h, int in_w, int minor, int kh,     \
             int kw, int up_x, int up_y, int down_x, int down_y, int px0, int px1, int py0, int py1)        \
    {                                                                                                       \
        if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kh < 1 || kw < 1) return HAV_EINVAL;        \
        int out_h = (in_h * up_y + py0 + py1 - kh + down_y) / down_y;                                       \
        int out_w = (in_w * up_x + px0 + px1 - kw + down_x) / down_x;                                       \
        if (out_h < 1 || out_w < 1) return HAV_EINVAL;                                                      \
        for (int64_t m = 0; m < major; ++m)                                                                 \
            for (int oy = 0; oy < out_h; ++oy)                                                              \
                for (int ox = 0; ox < out_w; ++ox)                                                          \
                    for (int mi = 0; mi < minor; ++mi) {                                                    \
                        ACC v = 0;                                                                          \
                        for (int i = 0; i < kh; ++i) {                                                      \
                            int yy = oy * down_y + i - py0; /* row in the zero-stuffed image */             \
                            if (yy < 0 || yy % up_y) continue;                                              \
                            int iy = yy / up_y;                                                             \
                            if (iy >= in_h) continue;                                                       \
                            for (int j = 0; j < kw; ++j) {                                                  \
                                int xx = ox * down_x + j - px0;                                             \
                                if (xx < 0 || xx % up_x) continue;                                          \
                                int ix = xx / up_x;                                                         \
                                if (ix >= in_w) continue;                                                   \
                                v += (ACC)in[((m * in_h + iy) * in_w + ix) * minor + mi] *                  \
                                     (ACC)k[(kh - 1 - i) * kw + (kw - 1 - j)];                              \
                            }                                                                               \
                        }                                                                                   \
                        out[((m * out_h + oy) * out_w + ox) * minor + mi] = (T)v;                           \
                    }                                                                                       \
        return 0;                                                                                           \
    }
DEF_UFD(orc_upfirdn2d_f32, float, float)
DEF_UFD(orc_upfirdn2d_f64, double, double)

/* ---- eval_sh (utils/sh_util.py:55-107), deg 0..4, sh [n,C,(deg+1)^2], dirs [n,3] -> [n,C] ---------- */
int orc_eval_sh_f32(int deg, const float* sh, const float* dirs, int64_t n, int C, float* out)
{
    static const double C0 = 0.28209479177387814, C1 = 0.4886025119029199;
    static const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
    static const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                 -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
    static const double C4[9] = {2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
                                 -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761};
    if (deg < 0 || deg > 4) return HAV_EINVAL;
    int K = (deg + 1) * (deg + 1);
    for (int64_t i = 0; i < n; ++i) {
        float x = dirs[i * 3], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
        for (int c = 0; c < C; ++c) {
            const float* s = sh + ((size_t)i * C + c) * K;
            float r = (float)C0 * s[0];
            if (deg > 0) {
                r = r - (float)C1 * y * s[1] + (float)C1 * z * s[2] - (float)C1 * x * s[3];
                if (deg > 1) {
                    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    r = r + (float)C2[0] * xy * s[4] + (float)C2[1] * yz * s[5] + (float)C2[2] * (2.0f * zz - xx - yy) * s[6] +
                        (float)C2[3] * xz * s[7] + (float)C2[4] * (xx - yy) * s[8];
                    if (deg > 2) {
                        r = r + (float)C3[0] * y * (3 * xx - yy) * s[9] + (float)C3[1] * xy * z * s[10] +
                            (float)C3[2] * y * (4 * zz - xx - yy) * s[11] + (float)C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * s[12] +
                            (float)C3[4] * x * (4 * zz - xx - yy) * s[13] + (float)C3[5] * z * (xx - yy) * s[14] +
                            (float)C3[6] * x * (xx - 3 * yy) * s[15];
                        if (deg > 3) {
                            r = r + (float)C4[0] * xy * (xx - yy) * s[16] + (float)C4[1] * yz * (3 * xx - yy) * s[17] +
                                (float)C4[2] * xy * (7 * zz - 1) * s[18] + (float)C4[3] * yz * (7 * zz - 3) * s[19] +
                                (float)C4[4] * (zz * (35 * zz - 30) + 3) * s[20] + (float)C4[5] * xz * (7 * zz - 3) * s[21] +
                                (float)C4[6] * (xx - yy) * (7 * zz - 1) * s[22] + (float)C4[7] * xz * (xx - 3 * yy) * s[23] +
                                (float)C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)) * s[24];
                        }
                    }
                }
            }
            out[(size_t)i * C + c] = r;
        }
    }
    return 0;
}

/* ---- get_rays (dataloader/data_util.py:28-56) + near/far (dataloader/dataloader.py:174-177) -------- */
int orc_gen_rays_f32(float* rays, int H, int W, const float intr[4], const float c2w[12], float near_, float far_,
                     int y0, int y1)
{
    /* K = [[fx,0,cx*W],[0,fy,cy*H],[0,0,1]];  K^-1 [i,j,1] = ((i-cx*W)/fx, (j-cy*H)/fy, 1) */
    float fx = intr[0], fy = intr[1], cx = intr[2] * (float)W, cy = intr[3] * (float)H;
    for (int j = y0; j < y1; ++j)
        for (int i = 0; i < W; ++i) {
            float dc[3] = {((float)i - cx) / fx, ((float)j - cy) / fy, 1.0f};
            float d[3];
            for (int r = 0; r < 3; ++r) d[r] = c2w[r * 4 + 0] * dc[0] + c2w[r * 4 + 1] * dc[1] + c2w[r * 4 + 2] * dc[2];
            float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            float* o = rays + ((size_t)(j - y0) * W + i) * 8;
            o[0] = c2w[3]; o[1] = c2w[7]; o[2] = c2w[11];
            o[3] = d[0] / nrm; o[4] = d[1] / nrm; o[5] = d[2] / nrm;
            o[6] = near_; o[7] = far_;
        }
    return 0;
}

/* ---- importance resampling between the two passes of the training forward --------------------------------
 * model/nerf_trainer.py:166-170 (z_vals_mid, sample_pdf, cat with z_vals[::2], sort) + utils/nerf_util.py:76-117 (sample_pdf),
 * one ray at a time.  Statement for statement like the PyTorch code: every product / sum rounded on its own (the function is
 * compiled without FMA contraction -- ATen's element-wise ops do not fuse across statements); sum and cumsum in index order
 * (the CPU cumsum's order; SURVEY B-11).  z, w [n,S_c]; zeta [n,S_f] raw torch.rand values or NULL (det=True);
 * zs [n,S_f] (nullable), z2 [n, ceil(S_c/2)+S_f]. */
#define DEF_RESAMPLE(NAME, T)                                                                                       \
    static int NAME##_cmp(const void* a, const void* b)                                                             \
    {                                                                                                               \
        T x = *(const T*)a, y = *(const T*)b;                                                                       \
        return (x > y) - (x < y);                                                                                   \
    }                                                                                                               \
    __attribute__((optimize("fp-contract=off"))) int NAME(const T* z, const T* w, int64_t n, int S_c, int S_f,      \
                                                          const float* zeta, T* zs, T* z2)                          \
    {                                                                                                               \
        if (!z || !w || !z2 || S_c < 3 || S_f < 1 || S_c > 4096 || S_f > 4096) return HAV_EINVAL;                   \
        const int nb = S_c - 1, nw = S_c - 2, S_half = (S_c + 1) / 2, S_fp = S_half + S_f;                          \
        T* cdf = (T*)malloc(sizeof(T) * (size_t)(nb + S_c + S_fp));                                                 \
        T* wp = cdf + nb;                                                                                           \
        T* cand = wp + S_c;                                                                                         \
        for (int64_t r = 0; r < n; ++r) {                                                                           \
            const T* zr = z + (size_t)r * S_c;                                                                      \
            T sum = 0;                                                                                              \
            for (int i = 0; i < nw; ++i) { wp[i] = w[(size_t)r * S_c + 1 + i] + (T)1e-5; sum = sum + wp[i]; } /* :79-80 */ \
            T run = 0;                                                                                              \
            cdf[0] = 0;                                                                                             \
            for (int i = 0; i < nw; ++i) { T pdf = wp[i] / sum; run = run + pdf; cdf[i + 1] = run; }   /* :80-84 */ \
            for (int k = 0; k < S_f; ++k) {                                                                         \
                T u;                                                                                                \
                if (!zeta) {                                            /* torch.linspace(0, 1, S_f), :87-90 */     \
                    T step = (T)1 / (T)(S_f - 1), lo_ = step * (T)k, hi_ = step * (T)(S_f - 1 - k);                 \
                    u = (S_f == 1) ? (T)0 : ((k < S_f / 2) ? lo_ : (T)1 - hi_);                                     \
                } else {                                                /* arange * s (a FLOAT32 tensor whatever the weights' dtype) + rand * (s - 1e-6), :93-95 */ \
                    T a_ = (T)((float)k * (float)(1.0 / (double)S_f)), b_ = (T)zeta[(size_t)r * S_f + k] * (T)(1.0 / (double)S_f - 1e-6); \
                    u = a_ + b_;                                                                                    \
                }                                                                                                   \
                int inds = 0;                                           /* searchsorted(right=True), :102 */        \
                while (inds < nb && cdf[inds] <= u) ++inds;                                                         \
                int below = inds - 1 < 0 ? 0 : inds - 1, above = inds > nb - 1 ? nb - 1 : inds;   /* :103-104 */    \
                T den = cdf[above] - cdf[below];                                                                    \
                if (den < (T)1e-5) den = 1;                             /* :112-113 */                              \
                T num = u - cdf[below];                                                                             \
                T t = num / den;                                                                                    \
                T sb = zr[below + 1] + zr[below], sa = zr[above + 1] + zr[above];                                   \
                T bl = (T)0.5 * sb, ba = (T)0.5 * sa;                   /* z_vals_mid, nerf_trainer.py:166 */       \
                T d_ = ba - bl, m_ = t * d_;                                                                        \
                T v = bl + m_;                                          /* :114-115 */                              \
                cand[S_half + k] = v;                                                                               \
                if (zs) zs[(size_t)r * S_f + k] = v;                                                                \
            }                                                                                                       \
            for (int i = 0; i < S_half; ++i) cand[i] = zr[2 * i];       /* z_vals[:, ::2], nerf_trainer.py:170 */   \
            qsort(cand, (size_t)S_fp, sizeof(T), NAME##_cmp);                                                       \
            for (int i = 0; i < S_fp; ++i) z2[(size_t)r * S_fp + i] = cand[i];                                      \
        }                                                                                                           \
        free(cdf);                                                                                                  \
        return 0;                                                                                                   \
    }
DEF_RESAMPLE(orc_resample_depths_f32, float)
DEF_RESAMPLE(orc_resample_depths_f64, double)

double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
