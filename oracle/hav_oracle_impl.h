/*
 * hav_oracle_impl.h -- body of the CPU oracle, included twice by hav_oracle.c with
 *   REAL = float  / SUF(x) = x##_f32   (the parity oracle: same precision as the reference)
 *   REAL = double / SUF(x) = x##_f64   (noise-floor oracle: reference math in fp64, SURVEY B-11)
 *
 * TEST INFRASTRUCTURE ONLY -- see the header comment of hav_oracle.c.
 * Each function cites the reference file:line it restates (XChenZ/havatar @ 2024_08_07).
 */

/* ---- helpers ------------------------------------------------------------------------------ */
static inline REAL SUF(r_floor)(REAL x) { return (REAL)floor((double)x); }
static inline REAL SUF(r_exp)(REAL x) { return sizeof(REAL) == 4 ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static inline REAL SUF(r_sin)(REAL x) { return sizeof(REAL) == 4 ? (REAL)sinf((float)x) : (REAL)sin((double)x); }
static inline REAL SUF(r_sqrt)(REAL x) { return sizeof(REAL) == 4 ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }

/* F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) for one point and all
 * C channels of one NCHW plane [C,H,W]  (utils/util.py:395-406). gx -> W axis, gy -> H axis. */
static void SUF(bilinear_zeros)(const float* plane, int C, int H, int W, REAL gx, REAL gy, REAL* out, int out_stride)
{
    REAL ix = ((gx + (REAL)1) / (REAL)2) * (REAL)(W - 1);
    REAL iy = ((gy + (REAL)1) / (REAL)2) * (REAL)(H - 1);
    REAL x0f = SUF(r_floor)(ix), y0f = SUF(r_floor)(iy);
    REAL wx1 = ix - x0f, wx0 = (REAL)1 - wx1;
    REAL wy1 = iy - y0f, wy0 = (REAL)1 - wy1;
    /* the clamp only protects the float->int cast for absurd coordinates; in-range taps are unaffected */
    REAL xc = x0f < (REAL)-4 ? (REAL)-4 : (x0f > (REAL)(W + 4) ? (REAL)(W + 4) : x0f);
    REAL yc = y0f < (REAL)-4 ? (REAL)-4 : (y0f > (REAL)(H + 4) ? (REAL)(H + 4) : y0f);
    int x0 = (int)xc, y0 = (int)yc, x1 = x0 + 1, y1 = y0 + 1;
    int vx0 = (x0 >= 0 && x0 < W), vx1 = (x1 >= 0 && x1 < W);
    int vy0 = (y0 >= 0 && y0 < H), vy1 = (y1 >= 0 && y1 < H);
    REAL wnw = wx0 * wy0, wne = wx1 * wy0, wsw = wx0 * wy1, wse = wx1 * wy1;
    for (int c = 0; c < C; ++c) {
        const float* pc = plane + (size_t)c * H * W;
        REAL v = 0;
        if (vx0 && vy0) v += (REAL)pc[y0 * W + x0] * wnw;
        if (vx1 && vy0) v += (REAL)pc[y0 * W + x1] * wne;
        if (vx0 && vy1) v += (REAL)pc[y1 * W + x0] * wsw;
        if (vx1 && vy1) v += (REAL)pc[y1 * W + x1] * wse;
        out[(size_t)c * out_stride] = v;
    }
}

/* F.grid_sample 3-D, mode='bilinear' (trilinear), padding_mode='border', align_corners=True on a
 * single-channel volume [D,H,W]; gx -> W, gy -> H, gz -> D  (utils/util.py:409-418). */
static REAL SUF(trilinear_border)(const float* vol, int D, int H, int W, REAL gx, REAL gy, REAL gz)
{
    REAL ix = ((gx + (REAL)1) / (REAL)2) * (REAL)(W - 1);
    REAL iy = ((gy + (REAL)1) / (REAL)2) * (REAL)(H - 1);
    REAL iz = ((gz + (REAL)1) / (REAL)2) * (REAL)(D - 1);
    ix = ix < 0 ? 0 : (ix > (REAL)(W - 1) ? (REAL)(W - 1) : ix);
    iy = iy < 0 ? 0 : (iy > (REAL)(H - 1) ? (REAL)(H - 1) : iy);
    iz = iz < 0 ? 0 : (iz > (REAL)(D - 1) ? (REAL)(D - 1) : iz);
    REAL x0f = SUF(r_floor)(ix), y0f = SUF(r_floor)(iy), z0f = SUF(r_floor)(iz);
    REAL fx = ix - x0f, fy = iy - y0f, fz = iz - z0f;
    int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
    REAL acc = 0;
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                int x = x0 + dx, y = y0 + dy, z = z0 + dz;
                if (x < 0 || x >= W || y < 0 || y >= H || z < 0 || z >= D) continue; /* weight is 0 there */
                REAL w = (dx ? fx : (REAL)1 - fx) * (dy ? fy : (REAL)1 - fy) * (dz ? fz : (REAL)1 - fz);
                acc += (REAL)vol[((size_t)z * H + y) * W + x] * w;
            }
    return acc;
}

/* One MLP + skin + gather evaluation of S samples of one ray, followed by compositing.
 * Steps 2-7 of SURVEY Appendix A. `zs` are the S depths; results are written to the pass outputs. */
typedef struct SUF(PassCtx) {
    const HavRenderParams* p;
    const float* planes;   /* [2,B,C,H,W] */
    const float* vol;      /* [2,D,H,W]   */
    const REAL *W1T, *W2T, *WfT;  /* transposed copies [in][out] */
    const HavMlpWeights* w;
} SUF(PassCtx);

static void SUF(ray_pass)(const SUF(PassCtx)* cx, int b, const float* ray, const float* bg3, const float* invT,
                          const REAL* zs, int S, const float* noise /* [S] or NULL */,
                          REAL* rgb67, REAL* depth, REAL* acc, REAL* wts /* [S] */, REAL* raw68 /* [S*68] or NULL */)
{
    const HavRenderParams* p = cx->p;
    const int C = p->plane_ch, PR = p->plane_res, VR = p->vol_res;
    const int IN = 2 * C + 48;
    REAL o[3] = {ray[0], ray[1], ray[2]}, d[3] = {ray[3], ray[4], ray[5]};
    REAL* X = (REAL*)malloc(sizeof(REAL) * (size_t)S * IN);
    REAL* H1 = (REAL*)malloc(sizeof(REAL) * (size_t)S * 128);
    REAL* H2 = (REAL*)malloc(sizeof(REAL) * (size_t)S * 128);
    REAL* RF = (REAL*)malloc(sizeof(REAL) * (size_t)S * 68);
    const float* P0 = cx->planes + ((size_t)0 * p->B + b) * C * PR * PR;
    const float* P1 = cx->planes + ((size_t)1 * p->B + b) * C * PR * PR;
    const float* V0 = cx->vol;
    const float* V1 = cx->vol + (size_t)VR * VR * VR;

    for (int s = 0; s < S; ++s) {
        /* pts = ro + rd * z  (model/nerf_trainer.py:141) */
        REAL pt[3];
        for (int a = 0; a < 3; ++a) pt[a] = o[a] + d[a] * zs[s];
        /* Deformation_Field_new.forward (model/Skinning_Field.py:77-95):
         * T0 = identity -> p0 = p;  T1 = inv_head_T -> p1 = (p + tau) @ M (row vector x matrix) */
        REAL p1[3], q[3];
        for (int c = 0; c < 3; ++c) {
            REAL sacc = 0;
            for (int r = 0; r < 3; ++r) sacc += (pt[r] + (REAL)invT[9 + r]) * (REAL)invT[r * 3 + c];
            p1[c] = sacc;
        }
        REAL g0[3], g1[3];
        for (int a = 0; a < 3; ++a) {
            g0[a] = pt[a] * (REAL)p->skin_scale[a] + (REAL)p->skin_trans[a]; /* UniformBoxWarp_new, utils/util.py:232-236 */
            g1[a] = p1[a] * (REAL)p->skin_scale[a] + (REAL)p->skin_trans[a];
        }
        REAL w0 = SUF(trilinear_border)(V0, VR, VR, VR, g0[0], g0[1], g0[2]);
        REAL w1 = SUF(trilinear_border)(V1, VR, VR, VR, g1[0], g1[1], g1[2]);
        REAL den = (w0 + w1) + (REAL)1e-8;                /* Skinning_Field.py:87 */
        REAL n0 = w0 / den, n1 = w1 / den;
        REAL pp[3];
        for (int a = 0; a < 3; ++a) pp[a] = n0 * pt[a] + n1 * p1[a];   /* :90,95 */
        /* sample_pts_triplane_feat (model/nerf_model.py:88-99) -> sample_from_triplane_new
         * (utils/util.py:359-392): plane0 at (x,y), plane1 at (z,y); stack(dim=-1) -> feature 2c+plane */
        for (int a = 0; a < 3; ++a) q[a] = pp[a] * (REAL)p->nerf_scale[a] + (REAL)p->nerf_trans[a];
        REAL* x = X + (size_t)s * IN;
        SUF(bilinear_zeros)(P0, C, PR, PR, q[0], q[1], x + 0, 2);
        SUF(bilinear_zeros)(P1, C, PR, PR, q[2], q[1], x + 1, 2);
        /* Embedder.embed (model/network/embedder.py:32-61), multires=8, include_input=False:
         * [f][sin(x f) xyz][sin(x f + pi/2) xyz]; cos is sin(theta + pi/2) in working precision */
        REAL* e = x + 2 * C;
        const REAL halfpi = (REAL)(3.14159265358979323846 / 2.0);
        for (int k = 0; k < 8; ++k) {
            REAL f = (REAL)(1 << k);
            for (int a = 0; a < 3; ++a) {
                REAL ang = pp[a] * f;
                e[6 * k + a] = SUF(r_sin)(ang);
                e[6 * k + 3 + a] = SUF(r_sin)(ang + halfpi);
            }
        }
    }
    /* radiance MLP (model/nerf_model.py:101-117): 176->128 relu ->128 relu -> {alpha 1, feat 64 (linear) -> rgb 3} */
    for (int s = 0; s < S; ++s) {
        REAL* h = H1 + (size_t)s * 128;
        for (int oo = 0; oo < 128; ++oo) h[oo] = (REAL)cx->w->b1[oo];
        const REAL* x = X + (size_t)s * IN;
        for (int k = 0; k < IN; ++k) {
            REAL xk = x[k];
            const REAL* wr = cx->W1T + (size_t)k * 128;
            for (int oo = 0; oo < 128; ++oo) h[oo] += xk * wr[oo];
        }
        for (int oo = 0; oo < 128; ++oo) h[oo] = h[oo] > 0 ? h[oo] : 0;
        REAL* h2 = H2 + (size_t)s * 128;
        for (int oo = 0; oo < 128; ++oo) h2[oo] = (REAL)cx->w->b2[oo];
        for (int k = 0; k < 128; ++k) {
            REAL xk = h[k];
            const REAL* wr = cx->W2T + (size_t)k * 128;
            for (int oo = 0; oo < 128; ++oo) h2[oo] += xk * wr[oo];
        }
        for (int oo = 0; oo < 128; ++oo) h2[oo] = h2[oo] > 0 ? h2[oo] : 0;
        REAL* rf = RF + (size_t)s * 68;   /* cat[rgb3, feat64, alpha1] (nerf_model.py:116-117) */
        REAL a = (REAL)cx->w->ba[0];
        for (int k = 0; k < 128; ++k) a += h2[k] * (REAL)cx->w->Wa[k];
        REAL* g = rf + 3;
        for (int oo = 0; oo < 64; ++oo) g[oo] = (REAL)cx->w->bf[oo];
        for (int k = 0; k < 128; ++k) {
            REAL xk = h2[k];
            const REAL* wr = cx->WfT + (size_t)k * 64;
            for (int oo = 0; oo < 64; ++oo) g[oo] += xk * wr[oo];
        }
        for (int c = 0; c < 3; ++c) {
            REAL v = (REAL)cx->w->bc[c];
            for (int k = 0; k < 64; ++k) v += g[k] * (REAL)cx->w->Wc[c * 64 + k];
            rf[c] = v;
        }
        rf[67] = a;
    }
    if (raw68) memcpy(raw68, RF, sizeof(REAL) * (size_t)S * 68);
    /* volume_render_radiance_field (utils/nerf_util.py:28-73) */
    REAL dn = SUF(r_sqrt)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    REAL T = 1, A = 0, Dp = 0;
    for (int c = 0; c < 67; ++c) rgb67[c] = 0;
    for (int s = 0; s < S; ++s) {
        REAL dist = (s + 1 < S) ? (zs[s + 1] - zs[s]) : (zs[S - 1] - zs[S - 2]);   /* :36-37 */
        dist = dist * dn;                                                        /* :38 */
        REAL* rf = RF + (size_t)s * 68;
        for (int c = 0; c < 3; ++c) rf[c] = (REAL)1 / ((REAL)1 + SUF(r_exp)(-rf[c]));  /* sigmoid on rgb only, :45-46 */
        REAL sg = rf[67] + (noise ? (REAL)noise[s] * (REAL)p->noise_std : (REAL)0);   /* :49-58 */
        sg = sg > 0 ? sg : 0;
        REAL alpha = (REAL)1 - SUF(r_exp)(-sg * dist);                           /* :59 */
        REAL wgt = alpha * T;                                                    /* :60 (exclusive product) */
        T = T * (((REAL)1 - alpha) + (REAL)1e-10);
        wts[s] = wgt;
        for (int c = 0; c < 67; ++c) rgb67[c] += wgt * rf[c];                    /* :62-63 */
        Dp += wgt * zs[s];                                                       /* :64-65 */
        A += wgt;                                                                /* :67 */
    }
    if (bg3) for (int c = 0; c < 3; ++c) rgb67[c] = rgb67[c] + ((REAL)1 - A) * (REAL)bg3[c];   /* :70-71 */
    *depth = Dp;
    *acc = A;
    free(X); free(H1); free(H2); free(RF);
}

/* sample_pdf (utils/nerf_util.py:76-117): bins [nb], weights [nb-1], -> ns samples.
 * `zeta` are the raw torch.rand values (NULL => det=True => u = linspace(0,1,ns)). */
static void SUF(sample_pdf)(const REAL* bins, const REAL* weights, int nb, int ns, const float* zeta, REAL* samples)
{
    int nw = nb - 1;
    REAL* cdf = (REAL*)malloc(sizeof(REAL) * (size_t)nb);
    REAL sum = 0;
    for (int i = 0; i < nw; ++i) sum += (weights[i] + (REAL)1e-5);              /* :79-80 */
    cdf[0] = 0;
    REAL run = 0;
    for (int i = 0; i < nw; ++i) { run += (weights[i] + (REAL)1e-5) / sum; cdf[i + 1] = run; }  /* :81-84 */
    for (int k = 0; k < ns; ++k) {
        REAL u;
        if (!zeta) {
            /* torch.linspace(0,1,ns): start + step*k for the first half, end - step*(ns-1-k) for the second */
            REAL step = (REAL)1 / (REAL)(ns - 1);
            u = (k < ns / 2) ? (REAL)0 + step * (REAL)k : (REAL)1 - step * (REAL)(ns - 1 - k);
            if (ns == 1) u = 0;
        } else {
            /* s = 1/ns is a Python double; arange*s and rand*(s-1e-6) are float32 ops on the casted scalars (:93-95) */
            /* torch.arange(ns) * s is a float32 tensor whatever the weights' dtype (int64 tensor x Python scalar) */
            REAL kN = (REAL)((float)k * (float)(1.0 / (double)ns));
            u = kN + (REAL)zeta[k] * (REAL)(1.0 / (double)ns - 1e-6);
        }
        int inds = 0;                                                            /* searchsorted(right=True), :102 */
        while (inds < nb && cdf[inds] <= u) ++inds;
        int below = inds - 1 < 0 ? 0 : inds - 1;                                 /* :103 */
        int above = inds > nb - 1 ? nb - 1 : inds;                               /* :104 */
        REAL den = cdf[above] - cdf[below];
        if (den < (REAL)1e-5) den = 1;                                           /* :112-113 */
        REAL t = (u - cdf[below]) / den;
        samples[k] = bins[below] + t * (bins[above] - bins[below]);              /* :114-115 */
    }
    free(cdf);
}

static int SUF(cmp_real)(const void* a, const void* b)
{
    REAL x = *(const REAL*)a, y = *(const REAL*)b;
    return (x > y) - (x < y);
}

/* Trainer.predict_and_render_radiance (model/nerf_trainer.py:120-201) for all B*R rays. */
int SUF(orc_render_rays)(const HavRenderParams* p, const float* rays, const float* bg, const float* inv_T,
                         const float* planes_nchw, const float* skin_vol, const HavMlpWeights* w,
                         const float* t_rand, const float* u_rand, const float* noise_c, const float* noise_f,
                         REAL* rgb_c, REAL* depth_c, REAL* acc_c, REAL* wmax,
                         REAL* rgb_f, REAL* depth_f, REAL* acc_f,
                         const OrcDebug* dbg, int nthreads)
{
    if (!p || p->S_c < 2 || p->plane_ch != 64) return HAV_EINVAL;
    if (p->perturb && !t_rand) return HAV_EINVAL;            /* the oracle never draws random numbers itself */
    if (p->perturb && p->S_f > 0 && !u_rand) return HAV_EINVAL;
    if (p->noise_std > 0 && (!noise_c || (p->S_f > 0 && !noise_f))) return HAV_EINVAL;
    if (p->S_c > 512 || p->S_f > 256) return HAV_EUNSUP;
    const int S_c = p->S_c, S_f = p->S_f;
    const int S_half = (S_c + 1) / 2;                        /* z_vals[:, ::2] */
    const int S_fp = S_f > 0 ? S_half + S_f : 0;
    const int IN = 2 * p->plane_ch + 48;
    REAL* W1T = (REAL*)malloc(sizeof(REAL) * (size_t)IN * 128);
    REAL* W2T = (REAL*)malloc(sizeof(REAL) * 128 * 128);
    REAL* WfT = (REAL*)malloc(sizeof(REAL) * 128 * 64);
    for (int oo = 0; oo < 128; ++oo) for (int k = 0; k < IN; ++k) W1T[(size_t)k * 128 + oo] = (REAL)w->W1[(size_t)oo * IN + k];
    for (int oo = 0; oo < 128; ++oo) for (int k = 0; k < 128; ++k) W2T[(size_t)k * 128 + oo] = (REAL)w->W2[(size_t)oo * 128 + k];
    for (int oo = 0; oo < 64; ++oo) for (int k = 0; k < 128; ++k) WfT[(size_t)k * 64 + oo] = (REAL)w->Wf[(size_t)oo * 128 + k];
    SUF(PassCtx) cx = {p, planes_nchw, skin_vol, W1T, W2T, WfT, w};
    const long NR = (long)p->B * p->R;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads)
    for (long gr = 0; gr < NR; ++gr) {
        int b = (int)(gr / p->R);
        const float* ray = rays + (size_t)gr * p->ray_stride;
        const float* bg3 = bg ? bg + (size_t)gr * 3 : NULL;
        const float* iT = inv_T + (size_t)b * 12;
        REAL zc[512], wc[512], zf[768], wf[768], mids[512], zs_new[256];
        REAL near = ray[6], far = ray[7];
        /* t_vals = linspace(0,1,S_c); z = near*(1-t) + far*t  (:129-130) */
        for (int i = 0; i < S_c; ++i) {
            REAL step = (REAL)1 / (REAL)(S_c - 1);
            REAL t = (i < S_c / 2) ? step * (REAL)i : (REAL)1 - step * (REAL)(S_c - 1 - i);
            zc[i] = near * ((REAL)1 - t) + far * t;
        }
        if (p->perturb) {                                     /* :132-139 */
            REAL lo[512], up[512];
            for (int i = 0; i < S_c; ++i) {
                REAL m_hi = (i + 1 < S_c) ? (REAL)0.5 * (zc[i + 1] + zc[i]) : zc[S_c - 1];
                REAL m_lo = (i > 0) ? (REAL)0.5 * (zc[i] + zc[i - 1]) : zc[0];
                up[i] = m_hi; lo[i] = m_lo;
            }
            for (int i = 0; i < S_c; ++i) zc[i] = lo[i] + (up[i] - lo[i]) * (REAL)t_rand[(size_t)gr * S_c + i];
        }
        REAL rgbt[67], dpt, act;
        SUF(ray_pass)(&cx, b, ray, bg3, iT, zc, S_c, noise_c ? noise_c + (size_t)gr * S_c : NULL, rgbt, &dpt, &act, wc,
                      (dbg && dbg->raw_coarse) ? (REAL*)dbg->raw_coarse + (size_t)gr * S_c * 68 : NULL);
        for (int c = 0; c < 67; ++c) rgb_c[(size_t)gr * 67 + c] = rgbt[c];
        depth_c[gr] = dpt; acc_c[gr] = act;
        REAL mx = wc[0];
        for (int i = 1; i < S_c; ++i) mx = wc[i] > mx ? wc[i] : mx;
        if (dbg && dbg->z_coarse) for (int i = 0; i < S_c; ++i) ((REAL*)dbg->z_coarse)[(size_t)gr * S_c + i] = zc[i];
        if (dbg && dbg->w_coarse) for (int i = 0; i < S_c; ++i) ((REAL*)dbg->w_coarse)[(size_t)gr * S_c + i] = wc[i];
        if (S_f > 0) {
            /* z_vals_mid; sample_pdf(mid, weights[1:-1]); sort(cat(z[::2], z_samples))  (:166-170) */
            for (int i = 0; i + 1 < S_c; ++i) mids[i] = (REAL)0.5 * (zc[i + 1] + zc[i]);
            SUF(sample_pdf)(mids, wc + 1, S_c - 1, S_f, (p->perturb ? (u_rand ? u_rand + (size_t)gr * S_f : NULL) : NULL), zs_new);
            int n = 0;
            for (int i = 0; i < S_c; i += 2) zf[n++] = zc[i];
            for (int i = 0; i < S_f; ++i) zf[n++] = zs_new[i];
            qsort(zf, (size_t)n, sizeof(REAL), SUF(cmp_real));
            SUF(ray_pass)(&cx, b, ray, bg3, iT, zf, S_fp, noise_f ? noise_f + (size_t)gr * S_fp : NULL, rgbt, &dpt, &act, wf,
                          (dbg && dbg->raw_fine) ? (REAL*)dbg->raw_fine + (size_t)gr * S_fp * 68 : NULL);
            for (int c = 0; c < 67; ++c) rgb_f[(size_t)gr * 67 + c] = rgbt[c];
            depth_f[gr] = dpt; acc_f[gr] = act;
            mx = wf[0];
            for (int i = 1; i < S_fp; ++i) mx = wf[i] > mx ? wf[i] : mx;   /* weights.max of the LAST pass (:195) */
            if (dbg && dbg->z_fine) for (int i = 0; i < S_fp; ++i) ((REAL*)dbg->z_fine)[(size_t)gr * S_fp + i] = zf[i];
            if (dbg && dbg->w_fine) for (int i = 0; i < S_fp; ++i) ((REAL*)dbg->w_fine)[(size_t)gr * S_fp + i] = wf[i];
        }
        wmax[gr] = mx;
    }
    free(W1T); free(W2T); free(WfT);
    return 0;
}
