// hav_train.hip -- the training-path statement of the march as two kernel pairs with gradients.
//
// The inference kernel (hav_render.hip) fuses the whole of predict_and_render_radiance; under autograd the radiance MLP stays on
// rocBLAS (its weight gradients are K = 0.9 M GEMMs) and what surrounds it is fused here:
//   field inputs  = Deformation_Field_new.forward (model/Skinning_Field.py:70-98) -> UniformBoxWarp_new (utils/util.py:232-236)
//                   -> sample_from_triplane_new (:359-406) + Embedder.embed (model/network/embedder.py:32-61) -> cat (nerf_model.py:104)
//                   : pts [n,3] -> X [n, 2C+48];   gradients to the planes (channels-last) and to the skinning volume
//   compositing   = volume_render_radiance_field + cumprod_exclusive (utils/nerf_util.py:4-73)
//                   : rf [rays,S,CH+1] -> rgb/acc/weights/depth;   gradient to rf
// Both are one wave per unit (query / ray) with lanes along channels or samples, so every plane row and every radiance-field row
// is read and written coalesced; scatters are float atomics on rows (planes) or on single taps (volume).
#include "hav_common.h"
#include <stdlib.h>
#include <string.h>

#define PE_FREQS 8
#define PE_DIM (6 * PE_FREQS)

__device__ __forceinline__ float wsum64(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// the same sum over the 64 lanes on the vector pipe: quad swaps, half-row and row mirrors (DPP), then the four row sums read out of lanes
// 0 / 16 / 32 / 48 -- 4 + 7 vector instructions instead of 6 trips through the LDS crossbar; every lane gets the same value
// (a different association of the 64 terms than wsum64's butterfly)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wsum64_dpp(float v)
{
    v = dpp_add<0xB1>(v);           // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);           // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);          // row_half_mirror
    v = dpp_add<0x140>(v);          // row_mirror
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16)))
         + (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

struct FieldArgs {
    float* X;
    unsigned short* Xb;          // MODE 0 of the run kernel: bf16 rows instead of X (hav_field_inputs_fwd_bf16)
    const float* dX;
    float* dplanes;
    float* dvol;
    const float* pts;
    const float* invT;
    const float* vol;
    const float* planes;
    float ss[3], st[3], bs[3], bt[3];
    int64_t n, n_per_b;
    int B, H, W, C, D;
    int S;          // MODE 2: > 0 = samples per ray: a run takes 16 neighbouring rays at one depth instead of 16 depths of one ray (see below)
    // MODE 4 (bit-reproducible scatter): 64-bit fixed-point accumulators, the words of hav_absmax(dX), the buffered volume taps
    long long* fplanes; const unsigned int* dx_amax; float* vval; int* vidx32; unsigned int* vmax; int fixbits;
};

// Bit-reproducible scatter (MODE 4, hav_field_inputs_bwd_fixed): float atomics make the sums depend on the order the atomics land in; integer
// addition is associative, so every contribution v is added as llrint(v * 2^(fixbits - e)) into a 64-bit accumulator with 2^e > max |v|
// (planes: |w g| <= max |dX|, known from hav_absmax(dX); volume: the taps are buffered, their maximum found on the way, and scattered by a
// second small kernel).  fixbits = min(40, 61 - log2(number of contributions)): no sum can leave the 63 bits; a contribution keeps >= 2^-40
// of the largest one, finer than the fp32 sum it replaces.  fixed_to_float_kernel turns the accumulators into the fp32 gradients.
__device__ __forceinline__ int fold_max_exp(const unsigned int* words, int nwords, int lane)
{
    unsigned int mb = 0;
    for (int i = lane; i < nwords; i += 64) mb = words[i] > mb ? words[i] : mb;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned int t = (unsigned int)__shfl_xor((int)mb, o, 64); mb = t > mb ? t : mb; }
    const int be = (int)((mb >> 23) & 0xFFu);          // biased exponent of the maximum: max < 2^(be - 126)
    const int e = be - 126;
    return e < -80 ? -80 : (e > 100 ? 100 : e);        // (an all-zero gradient, or one beyond 2^100: the scales below stay normal floats)
}
__device__ __forceinline__ long long to_fixed(float v, float scale) { return __float2ll_rn(v * scale); }

// bilinear taps of F.grid_sample(align_corners=True, padding_mode='zeros'); weights of out-of-range taps are zeroed
__device__ __forceinline__ void plane_taps(float u, float v, int H, int W, int (&idx)[4], float (&w)[4], float& wx0, float& wx1,
                                           float& wy0, float& wy1, bool (&valid)[4])
{
    const float ix = ((u + 1.0f) * 0.5f) * (float)(W - 1), iy = ((v + 1.0f) * 0.5f) * (float)(H - 1);
    float x0f = floorf(ix), y0f = floorf(iy);
    wx1 = ix - x0f; wx0 = 1.0f - wx1; wy1 = iy - y0f; wy0 = 1.0f - wy1;
    x0f = fminf(fmaxf(x0f, -2.f), (float)W + 1.f); y0f = fminf(fmaxf(y0f, -2.f), (float)H + 1.f);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1), cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    idx[0] = cy0 * W + cx0; idx[1] = cy0 * W + cx1; idx[2] = cy1 * W + cx0; idx[3] = cy1 * W + cx1;
    valid[0] = vx0 && vy0; valid[1] = vx1 && vy0; valid[2] = vx0 && vy1; valid[3] = vx1 && vy1;
    w[0] = valid[0] ? wx0 * wy0 : 0.f; w[1] = valid[1] ? wx1 * wy0 : 0.f; w[2] = valid[2] ? wx0 * wy1 : 0.f; w[3] = valid[3] ? wx1 * wy1 : 0.f;
}

// MODE 0: X out.  MODE 1: dplanes / dvol scatter from dX, one row of float atomics per tap.  MODE 2 (C <= 64): the same scatter with
// the taps of FI_RUN consecutive queries merged in registers first.  Queries arrive ray-major, sample-minor, so a wave that walks a run
// of consecutive queries walks along a ray: in the plane the ray is (nearly) normal to, every sample lands on the same four texels; in
// the other one consecutive bilinear cells share an edge.  Everything about a tap but its channel is wave-uniform (lane = channel), so
// the texel ids live in SGPRs (readfirstlane), the match of old against new taps is scalar code with uniform branches, and a row of
// atomics is only issued when a texel leaves the 2 x 2 window (or the run ends).  Sums are re-associated, not changed otherwise.
// MODE 4: MODE 2's walk with 64-bit fixed-point sums and integer atomics (see above): the result does not depend on the order of anything.
#define FI_RUN 16
template <int MODE>
__global__ void __launch_bounds__(256) field_inputs_kernel(FieldArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int C = a.C, H = a.H, W = a.W, D = a.D, XW = 2 * C + PE_DIM;
    const size_t plane_sz = (size_t)a.B * H * W * C, vol_sz = (size_t)D * D * D;
    constexpr int RUN = (MODE == 2 || MODE == 4) ? FI_RUN : 1;
    constexpr bool WIN = MODE == 2 || MODE == 4;          // tap windows in registers
    const float pscale = MODE == 4 ? __uint_as_float((unsigned int)(127 + a.fixbits - fold_max_exp(a.dx_amax, 256, lane)) << 23) : 0.f;
    float vmax_w = 0.f;
    const int64_t nruns = (a.n + RUN - 1) / RUN;
    for (int64_t run = wave0; run < nruns; run += nwaves) {
    // MODE 2: the window of each plane -- row ids (texel row of dplanes in units of C floats, -1 = empty) and this lane's pending sums
    int wkey[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}};
    float wacc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    long long facc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};          // (MODE 4)
    // MODE 2: this lane's pending volume tap -- consecutive samples of a ray fall into the same cell of the 64^3 volume about twice in a
    // row (the same 8 corners in the same lanes): their contributions are summed here and leave as one atomic when the corner changes
    long long vkey = -1; float vacc = 0.f;
    // Which 16 queries make a run.  Ray-major (the default): 16 consecutive depths of one ray -- the (x, y) plane's taps repeat, the (z, y)
    // plane's change at every sample.  With a.S (samples per ray, hav_field_inputs_bwd_rows) for rays that come in image rows (the training
    // patch: 64 x 64 neighbouring pixels): 16 NEIGHBOURING RAYS at one depth -- the (z, y) plane's taps are the same for all of them, the
    // (x, y) plane's advance by half a texel per ray, the volume's cell is shared by ~8 rays: 1.91 -> 1.44 ms per call at config 5's size.
    int64_t i_base = run * RUN, i_step = 1;
    if (WIN && a.S > 0) {
        const int64_t runs_b = (a.n_per_b / a.S / RUN) * a.S, bb = run / runs_b, u = run - bb * runs_b;
        const int64_t rb = u / a.S, s = u - rb * a.S;
        i_base = bb * a.n_per_b + rb * RUN * a.S + s;
        i_step = a.S;
    }
    for (int qi = 0; qi < RUN; ++qi) {
        const int64_t i = i_base + qi * i_step;
        if (i >= a.n) break;
        const int b = (int)(i / a.n_per_b);
        const float px = a.pts[i * 3 + 0], py = a.pts[i * 3 + 1], pz = a.pts[i * 3 + 2];
        // ---- Deformation_Field_new: p_0 = p (identity), p_1 = (p + tau) M       (Skinning_Field.py:77-83)
        const float* T = a.invT + (size_t)b * 12;
        const float tx = px + T[9], ty = py + T[10], tz = pz + T[11];
        const float p1x = tx * T[0] + ty * T[3] + tz * T[6], p1y = tx * T[1] + ty * T[4] + tz * T[7], p1z = tx * T[2] + ty * T[5] + tz * T[8];
        // skinning weights: lanes 0-7 hold the 8 taps of bone 0 at boxwarp(p_0), lanes 8-15 those of bone 1 at boxwarp(p_1)
        // (grid_sample 3-D, align_corners=True, padding_mode='border', :85)
        const int bone = (lane >> 3) & 1, tap = lane & 7;
        const float sx = bone ? p1x : px, sy = bone ? p1y : py, sz = bone ? p1z : pz;
        float gx = ((sx * a.ss[0] + a.st[0]) + 1.f) * 0.5f * (float)(D - 1), gy = ((sy * a.ss[1] + a.st[1]) + 1.f) * 0.5f * (float)(D - 1),
              gz = ((sz * a.ss[2] + a.st[2]) + 1.f) * 0.5f * (float)(D - 1);
        gx = fminf(fmaxf(gx, 0.f), (float)(D - 1)); gy = fminf(fmaxf(gy, 0.f), (float)(D - 1)); gz = fminf(fmaxf(gz, 0.f), (float)(D - 1));
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        const int vx = (int)fx + (tap & 1), vy = (int)fy + ((tap >> 1) & 1), vz = (int)fz + (tap >> 2);
        const float wx = (tap & 1) ? gx - fx : 1.f - (gx - fx), wy = (tap & 2) ? gy - fy : 1.f - (gy - fy), wz = (tap & 4) ? gz - fz : 1.f - (gz - fz);
        const bool vin = lane < 16 && vx < D && vy < D && vz < D;
        const size_t vidx = (size_t)bone * vol_sz + ((size_t)vz * D + vy) * D + vx;
        const float tw = vin ? wx * wy * wz : 0.f;
        float part = vin ? tw * a.vol[vidx] : 0.f;
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
        const float w0 = __shfl(part, 0, 64), w1 = __shfl(part, 8, 64);
        // :87.  The two quotients as one reciprocal + Newton step (<= 1 ulp from w / s), exactly as the inference kernel forms them
        // (hav_render.hip HAV_FAST_DIV, docs/history/DESIGN_r1-r4.md 3.12: the compiler's interleaved IEEE division sequences are not trusted on gfx950)
        const float s = (w0 + w1) + 1e-8f;
        float rs = __builtin_amdgcn_rcpf(s);
        rs = rs * (2.0f - s * rs);
        const float h0 = w0 * rs, h1 = w1 * rs;
        const float rx = h0 * px + h1 * p1x, ry = h0 * py + h1 * p1y, rz = h0 * pz + h1 * p1z;   // :90,95
        // ---- UniformBoxWarp_new of the NeRF box, then the two plane look-ups (nerf_model.py:88-99)
        const float qx = rx * a.bs[0] + a.bt[0], qy = ry * a.bs[1] + a.bt[1], qz = rz * a.bs[2] + a.bt[2];
        float dqx = 0.f, dqy = 0.f, dqz = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int idx[4]; float w[4]; bool valid[4]; float wx0, wx1, wy0, wy1;
            plane_taps(p ? qz : qx, qy, H, W, idx, w, wx0, wx1, wy0, wy1, valid);
            const float* pl = a.planes + p * plane_sz + (size_t)b * H * W * C;
            // MODE 2 (C <= 64): exactly one trip for every lane (idle lanes take channel C - 1 with a zero gradient), so that the window
            // bookkeeping below stays wave-uniform
            for (int c0 = lane; WIN ? c0 == lane : c0 < C; c0 += 64) {
                const int c = WIN ? min(c0, C - 1) : c0;
                float t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = valid[k] ? pl[(size_t)idx[k] * C + c] : 0.f;
                if (MODE == 0) {
                    a.X[i * XW + 2 * c + p] = ((t[0] * w[0] + t[1] * w[1]) + t[2] * w[2]) + t[3] * w[3];
                } else {
                    const float g = (WIN && lane >= C) ? 0.f : a.dX[i * XW + 2 * c + p];
                    if (a.dplanes && MODE == 1) {
                        float* dpl = a.dplanes + p * plane_sz + (size_t)b * H * W * C;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (valid[k]) atomicAdd(dpl + (size_t)idx[k] * C + c, w[k] * g);
                    }
                    if (a.dplanes && MODE == 2) {
                        // new window: the four taps of this query; old sums move to the slot of the same texel or are flushed
                        int nkey[4]; float nacc[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            nkey[k] = __builtin_amdgcn_readfirstlane(valid[k] ? b * H * W + idx[k] : -1);
                            nacc[k] = w[k] * g;
                        }
                        float* dp0 = a.dplanes + p * plane_sz;
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const int ok = wkey[p][o];
                            if (ok < 0) continue;
                            if (ok == nkey[0]) nacc[0] += wacc[p][o];
                            else if (ok == nkey[1]) nacc[1] += wacc[p][o];
                            else if (ok == nkey[2]) nacc[2] += wacc[p][o];
                            else if (ok == nkey[3]) nacc[3] += wacc[p][o];
                            else if (lane < C) atomicAdd(dp0 + (size_t)ok * C + lane, wacc[p][o]);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) { wkey[p][k] = nkey[k]; wacc[p][k] = nacc[k]; }
                    }
                    if (a.fplanes && MODE == 4) {          // the same window walk on integers
                        int nkey[4]; long long nacc[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            nkey[k] = __builtin_amdgcn_readfirstlane(valid[k] ? b * H * W + idx[k] : -1);
                            nacc[k] = to_fixed(w[k] * g, pscale);
                        }
                        long long* dp0 = a.fplanes + p * plane_sz;
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const int ok = wkey[p][o];
                            if (ok < 0) continue;
                            if (ok == nkey[0]) nacc[0] += facc[p][o];
                            else if (ok == nkey[1]) nacc[1] += facc[p][o];
                            else if (ok == nkey[2]) nacc[2] += facc[p][o];
                            else if (ok == nkey[3]) nacc[3] += facc[p][o];
                            else if (lane < C && facc[p][o] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(dp0 + (size_t)ok * C + lane), (unsigned long long)facc[p][o]);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) { wkey[p][k] = nkey[k]; facc[p][k] = nacc[k]; }
                    }
                    const float ggx = g * ((t[1] - t[0]) * wy0 + (t[3] - t[2]) * wy1);
                    const float ggy = g * ((t[2] - t[0]) * wx0 + (t[3] - t[1]) * wx1);
                    if (p == 0) dqx += ggx; else dqz += ggx;
                    dqy += ggy;
                }
            }
        }
        // ---- Embedder.embed: [f][sin(x f), sin(y f), sin(z f), sin(x f + pi/2), ...], f = 2^0..2^7 (embedder.py:42-56)
        const int f = lane / 6, r6 = lane - 6 * f, j = r6 >= 3 ? r6 - 3 : r6;
        const float freq = (float)(1 << (f & 7));
        const float coord = j == 0 ? rx : (j == 1 ? ry : rz);
        const float arg = r6 >= 3 ? coord * freq + 1.57079632679489661923f : coord * freq;
        if (MODE == 0) {
            if (lane < PE_DIM) a.X[i * XW + 2 * C + lane] = sinf(arg);
        } else if (a.dvol || (MODE == 4 && a.vval)) {
            // d loss / d p' : plane coordinates (through the box warp) + the encoding
            float dx = dqx * (0.5f * (float)(W - 1)) * a.bs[0], dy = dqy * (0.5f * (float)(H - 1)) * a.bs[1], dz = dqz * (0.5f * (float)(W - 1)) * a.bs[2];
            if (lane < PE_DIM) {
                const float ge = a.dX[i * XW + 2 * C + lane] * freq * cosf(arg);
                if (j == 0) dx += ge; else if (j == 1) dy += ge; else dz += ge;
            }
            dx = wsum64_dpp(dx); dy = wsum64_dpp(dy); dz = wsum64_dpp(dz);
            // p' = h0 p0 + h1 p1, h_i = w_i / s: d/dw_i = (d/dh_i - sum_j d/dh_j h_j) / s
            const float dh0 = dx * px + dy * py + dz * pz, dh1 = dx * p1x + dy * p1y + dz * p1z;
            const float mix = dh0 * h0 + dh1 * h1;
            const float dw = ((bone ? dh1 : dh0) - mix) * rs;
            if (MODE == 4) {          // buffered: [n][16] values + voxel ids; the scatter (and its scale) come afterwards
                if (lane < 16) {
                    const float cv = vin ? tw * dw : 0.f;
                    a.vval[i * 16 + lane] = cv;
                    a.vidx32[i * 16 + lane] = vin ? (int)vidx : -1;
                    vmax_w = fmaxf(vmax_w, fabsf(cv));
                }
            } else if (MODE == 2) {
                const float cv = vin ? tw * dw : 0.f;
                const long long key = vin ? (long long)vidx : -1;
                if (key == vkey) vacc += cv;
                else {
                    if (vkey >= 0 && vacc != 0.f) atomicAdd(a.dvol + vkey, vacc);
                    vkey = key; vacc = cv;
                }
            } else if (vin && tw * dw != 0.f) atomicAdd(a.dvol + vidx, tw * dw);   // clamped (border) coordinates: half the taps weigh 0
        }
    }
    if (MODE == 2 && a.dvol && vkey >= 0 && vacc != 0.f) atomicAdd(a.dvol + vkey, vacc);
    if (MODE == 2 && a.dplanes) {          // end of the run: whatever is still in the windows
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (wkey[p][o] >= 0 && lane < C) atomicAdd(a.dplanes + p * plane_sz + (size_t)wkey[p][o] * C + lane, wacc[p][o]);
    }
    if (MODE == 4 && a.fplanes) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (wkey[p][o] >= 0 && lane < C && facc[p][o] != 0)
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.fplanes + p * plane_sz + (size_t)wkey[p][o] * C + lane), (unsigned long long)facc[p][o]);
    }
    }
    if (MODE == 4 && a.vmax) {          // max |volume tap| of this wave (a maximum does not depend on the order either)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax_w = fmaxf(vmax_w, __shfl_xor(vmax_w, o, 64));
        if (lane == 0 && vmax_w > 0.f) atomicMax(a.vmax, __float_as_uint(vmax_w));
    }
}

// ---- the same field inputs, 16 queries of a wave at a time --------------------------------------------------------------------------
// field_inputs_kernel walks its queries one after the other: the point -> the volume taps -> the plane taps -> (backward) the gradient
// row are four dependent round trips to memory per query, 16 queries deep, and nothing else hides them but the other waves of the
// compute unit.  Here a wave takes
// a run of 16 consecutive queries through two stages:
//   1. geometry, 4 lanes per query: the point, both bones' volume taps (4 taps per lane, all 16 queries' loads in flight together), the
//      blended point, the bilinear taps of both planes -- per-query values that stay in the lanes 4 q .. 4 q + 3;
//   2. channels, lane = channel: for 4 queries at a time every plane row (and gradient row) is requested before the first one is used --
//      the tap ids, weights and validity bits of a query are wave-uniform and come out of lane 4 q with v_readlane --, then the queries
//      are finished in order: X rows out (MODE 0), or the tap windows merged and flushed as rows of atomics exactly as field_inputs_kernel<2>
//      does (MODE 2); the volume gradients leave at the end of the run from the geometry lanes (4 taps per lane).
// The arithmetic per query is the other kernel's, statement for statement (same association of every sum).
// Measured at config 5's size (0.92 M queries, tools/bench_field_inputs.py): forward 0.58 -> 0.27 ms per call; backward see hav_field_inputs_bwd.
#define FR_RUN 16
template <int MODE>
__global__ void __launch_bounds__(256) field_inputs_run_kernel(FieldArgs a)
{
    const int lane = threadIdx.x & 63, sub = lane & 3, ql = lane >> 2;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int C = a.C, H = a.H, W = a.W, D = a.D, XW = 2 * C + PE_DIM;
    const size_t plane_sz = (size_t)a.B * H * W * C;
    const int vol_sz = D * D * D;
    const int cc = lane < C ? lane : C - 1;
    const bool con = lane < C;
    const int f = lane / 6, r6 = lane - 6 * f, j = r6 >= 3 ? r6 - 3 : r6;
    const float freq = (float)(1 << (f & 7));
    const bool pe_on = lane < PE_DIM;
    const int64_t nruns = (a.n + FR_RUN - 1) / FR_RUN;
    for (int64_t run = wave0; run < nruns; run += nwaves) {
        const int64_t i0 = run * FR_RUN;
        // ---------------- stage 1: geometry of query i0 + ql in lanes 4 ql .. 4 ql + 3
        const bool qok = i0 + ql < a.n;
        const int64_t iq = qok ? i0 + ql : a.n - 1;
        const int b = (int)(iq / a.n_per_b);
        const float px = a.pts[iq * 3 + 0], py = a.pts[iq * 3 + 1], pz = a.pts[iq * 3 + 2];
        const float* T = a.invT + (size_t)b * 12;
        const float tx = px + T[9], ty = py + T[10], tz = pz + T[11];
        const float p1x = tx * T[0] + ty * T[3] + tz * T[6], p1y = tx * T[1] + ty * T[4] + tz * T[7], p1z = tx * T[2] + ty * T[5] + tz * T[8];
        const int bone = sub >> 1;
        const float sx = bone ? p1x : px, sy = bone ? p1y : py, sz = bone ? p1z : pz;
        float gx = ((sx * a.ss[0] + a.st[0]) + 1.f) * 0.5f * (float)(D - 1), gy = ((sy * a.ss[1] + a.st[1]) + 1.f) * 0.5f * (float)(D - 1),
              gz = ((sz * a.ss[2] + a.st[2]) + 1.f) * 0.5f * (float)(D - 1);
        gx = fminf(fmaxf(gx, 0.f), (float)(D - 1)); gy = fminf(fmaxf(gy, 0.f), (float)(D - 1)); gz = fminf(fmaxf(gz, 0.f), (float)(D - 1));
        const float fx = floorf(gx), fy = floorf(gy), fz = floorf(gz);
        float tw[4]; int vidx[4]; bool vin[4]; float pv[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int tap = 4 * (sub & 1) + tt;
            const int vx = (int)fx + (tap & 1), vy = (int)fy + ((tap >> 1) & 1), vz = (int)fz + (tap >> 2);
            const float wx = (tap & 1) ? gx - fx : 1.f - (gx - fx), wy = (tap & 2) ? gy - fy : 1.f - (gy - fy), wz = (tap & 4) ? gz - fz : 1.f - (gz - fz);
            vin[tt] = vx < D && vy < D && vz < D;
            vidx[tt] = bone * vol_sz + (min(vz, D - 1) * D + min(vy, D - 1)) * D + min(vx, D - 1);
            tw[tt] = vin[tt] ? wx * wy * wz : 0.f;
            pv[tt] = a.vol[vidx[tt]];
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) pv[tt] = vin[tt] ? tw[tt] * pv[tt] : 0.f;
        float part = (pv[0] + pv[1]) + (pv[2] + pv[3]);          // the butterfly of the other kernel: xor 1, xor 2 inside the lane, xor 4 across
        part += __shfl_xor(part, 1, 64);
        const float wo = __shfl_xor(part, 2, 64);
        const float w0 = bone ? wo : part, w1 = bone ? part : wo;
        const float s = (w0 + w1) + 1e-8f;
        float rs = __builtin_amdgcn_rcpf(s);
        rs = rs * (2.0f - s * rs);
        const float h0 = w0 * rs, h1 = w1 * rs;
        const float rx = h0 * px + h1 * p1x, ry = h0 * py + h1 * p1y, rz = h0 * pz + h1 * p1z;
        const float qx = rx * a.bs[0] + a.bt[0], qy = ry * a.bs[1] + a.bt[1], qz = rz * a.bs[2] + a.bt[2];
        int trow[2][4]; float twt[2][4], twx0[2], twx1[2], twy0[2], twy1[2];
        int vmask = 0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            int idx[4]; bool valid[4];
            plane_taps(p ? qz : qx, qy, H, W, idx, twt[p], twx0[p], twx1[p], twy0[p], twy1[p], valid);
#pragma unroll
            for (int k = 0; k < 4; ++k) { trow[p][k] = b * H * W + idx[k]; vmask |= valid[k] ? 1 << (4 * p + k) : 0; }
        }
        float dxs = 0.f, dys = 0.f, dzs = 0.f;          // (MODE 2) d loss / d p' of this lane's query, filled in stage 2

        // ---------------- stage 2: lane = channel
        int wkey[2][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}};
        float wacc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto bcast_i = [&](int v, int q) { return __builtin_amdgcn_readlane(v, 4 * q); };
        auto bcast_f = [&](float v, int q) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 4 * q)); };
#pragma unroll 1
        for (int qg = 0; qg < FR_RUN / 4; ++qg) {
            float t[4][2][4]; float2 g2[4]; float gpe[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int q = 4 * qg + qq;
                const int64_t i = i0 + q < a.n ? i0 + q : a.n - 1;          // (queries past the end repeat the last one; nothing of them leaves)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[qq][p][k] = a.planes[p * plane_sz + (size_t)bcast_i(trow[p][k], q) * C + cc];
                if (MODE != 0) {
                    g2[qq] = *reinterpret_cast<const float2*>(a.dX + i * XW + 2 * cc);
                    gpe[qq] = a.dX[i * XW + 2 * C + (pe_on ? lane : 0)];
                }
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int q = 4 * qg + qq;
                const int64_t i = i0 + q;
                if (i >= a.n) break;          // wave-uniform
                const int vm = bcast_i(vmask, q);
                float dqx = 0.f, dqy = 0.f, dqz = 0.f;
                float xo[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    float w[4], tt4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { w[k] = bcast_f(twt[p][k], q); tt4[k] = ((vm >> (4 * p + k)) & 1) ? t[qq][p][k] : 0.f; }
                    if (MODE == 0) {
                        xo[p] = ((tt4[0] * w[0] + tt4[1] * w[1]) + tt4[2] * w[2]) + tt4[3] * w[3];
                    } else {
                        const float g = con ? (p ? g2[qq].y : g2[qq].x) : 0.f;
                        const float wx0 = bcast_f(twx0[p], q), wx1 = bcast_f(twx1[p], q), wy0 = bcast_f(twy0[p], q), wy1 = bcast_f(twy1[p], q);
                        if (a.dplanes) {
                            // new window: the four taps of this query; old sums move to the slot of the same texel or are flushed
                            int nkey[4]; float nacc[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                nkey[k] = ((vm >> (4 * p + k)) & 1) ? bcast_i(trow[p][k], q) : -1;
                                nacc[k] = w[k] * g;
                            }
                            float* dp0 = a.dplanes + p * plane_sz;
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                const int ok = wkey[p][o];
                                if (ok < 0) continue;
                                if (ok == nkey[0]) nacc[0] += wacc[p][o];
                                else if (ok == nkey[1]) nacc[1] += wacc[p][o];
                                else if (ok == nkey[2]) nacc[2] += wacc[p][o];
                                else if (ok == nkey[3]) nacc[3] += wacc[p][o];
                                else if (con) atomicAdd(dp0 + (size_t)ok * C + lane, wacc[p][o]);
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) { wkey[p][k] = nkey[k]; wacc[p][k] = nacc[k]; }
                        }
                        const float ggx = g * ((tt4[1] - tt4[0]) * wy0 + (tt4[3] - tt4[2]) * wy1);
                        const float ggy = g * ((tt4[2] - tt4[0]) * wx0 + (tt4[3] - tt4[1]) * wx1);
                        if (p == 0) dqx += ggx; else dqz += ggx;
                        dqy += ggy;
                    }
                }
                const float rxq = bcast_f(rx, q), ryq = bcast_f(ry, q), rzq = bcast_f(rz, q);
                const float coord = j == 0 ? rxq : (j == 1 ? ryq : rzq);
                const float arg = r6 >= 3 ? coord * freq + 1.57079632679489661923f : coord * freq;
                if (MODE == 0) {
                    if (a.Xb) {          // bf16 rows, rounded as the MLP kernels round fp32 rows (v_cvt_pk_bf16_f32: nearest even)
                        typedef float f2v __attribute__((ext_vector_type(2)));
                        typedef __bf16 b2v __attribute__((ext_vector_type(2)));
                        const f2v pr = {xo[0], xo[1]};
                        if (con) reinterpret_cast<unsigned int*>(a.Xb + i * XW)[lane] = __builtin_bit_cast(unsigned int, __builtin_convertvector(pr, b2v));
                        const f2v ps = {sinf(arg), 0.f};
                        if (pe_on) a.Xb[i * XW + 2 * C + lane] = (unsigned short)(__builtin_bit_cast(unsigned int, __builtin_convertvector(ps, b2v)) & 0xFFFFu);
                    } else {
                        if (con) *reinterpret_cast<float2*>(a.X + i * XW + 2 * lane) = make_float2(xo[0], xo[1]);
                        if (pe_on) a.X[i * XW + 2 * C + lane] = sinf(arg);
                    }
                } else if (a.dvol) {
                    float dx = dqx * (0.5f * (float)(W - 1)) * a.bs[0], dy = dqy * (0.5f * (float)(H - 1)) * a.bs[1], dz = dqz * (0.5f * (float)(W - 1)) * a.bs[2];
                    if (pe_on) {
                        const float ge = gpe[qq] * freq * cosf(arg);
                        if (j == 0) dx += ge; else if (j == 1) dy += ge; else dz += ge;
                    }
                    dx = wsum64_dpp(dx); dy = wsum64_dpp(dy); dz = wsum64_dpp(dz);
                    if (ql == q) { dxs = dx; dys = dy; dzs = dz; }
                }
            }
        }
        if (MODE != 0 && a.dplanes) {          // end of the run: whatever is still in the windows
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (wkey[p][o] >= 0 && con) atomicAdd(a.dplanes + p * plane_sz + (size_t)wkey[p][o] * C + lane, wacc[p][o]);
        }
        if (MODE != 0 && a.dvol && qok) {
            // p' = h0 p0 + h1 p1, h_i = w_i / s: d/dw_i = (d/dh_i - sum_j d/dh_j h_j) / s
            const float dh0 = dxs * px + dys * py + dzs * pz, dh1 = dxs * p1x + dys * p1y + dzs * p1z;
            const float mix = dh0 * h0 + dh1 * h1;
            const float dw = ((bone ? dh1 : dh0) - mix) * rs;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                if (vin[tt] && tw[tt] * dw != 0.f) atomicAdd(a.dvol + vidx[tt], tw[tt] * dw);   // clamped (border) coordinates: half the taps weigh 0
        }
    }
}

// volume taps of MODE 4 -> 64-bit fixed-point accumulators (scale from the maximum the first kernel left in *vmax)
__global__ void __launch_bounds__(256) vol_fixed_scatter_kernel(long long* __restrict__ fvol, const float* __restrict__ vval, const int* __restrict__ vidx,
                                                                int64_t n16, const unsigned int* __restrict__ vmax, int fixbits)
{
    const int e = fold_max_exp(vmax, 1, threadIdx.x & 63);
    const float scale = __uint_as_float((unsigned int)(127 + fixbits - e) << 23);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        const int id = vidx[i];
        const float v = vval[i];
        if (id >= 0 && v != 0.f) atomicAdd(reinterpret_cast<unsigned long long*>(fvol + id), (unsigned long long)to_fixed(v, scale));
    }
}
// out = acc * 2^(e - fixbits): e from the words of hav_absmax (nwords = 256) or from one maximum word (nwords = 1)
__global__ void __launch_bounds__(256) fixed_to_float_kernel(float* __restrict__ out, const long long* __restrict__ acc, int64_t n,
                                                             const unsigned int* __restrict__ words, int nwords, int fixbits)
{
    const int e = fold_max_exp(words, nwords, threadIdx.x & 63);
    // two exact power-of-two factors (their product may not be a normal float)
    const int sh = e - fixbits;
    const int s1 = sh / 2, s2 = sh - s1;
    const float f1 = __uint_as_float((unsigned int)(127 + s1) << 23), f2 = __uint_as_float((unsigned int)(127 + s2) << 23);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = ((float)acc[i] * f1) * f2;
}

static int field_check(const HavFieldParams* p, const void* pts, const void* invT, const void* vol, const void* planes)
{
    if (!p || !pts || !invT || !vol || !planes) return HAV_EINVAL;
    if (p->n < 0 || p->n_per_b < 1 || p->B < 1 || p->H < 2 || p->W < 2 || p->C < 1 || p->D < 2) return HAV_EINVAL;
    if (p->n > p->n_per_b * p->B) return HAV_EINVAL;
    return 0;
}

static FieldArgs field_args(const HavFieldParams* p, const float* pts, const float* invT, const float* vol, const float* planes)
{
    FieldArgs a{};
    a.pts = pts; a.invT = invT; a.vol = vol; a.planes = planes;
    for (int k = 0; k < 3; ++k) { a.ss[k] = p->skin_scale[k]; a.st[k] = p->skin_trans[k]; a.bs[k] = p->nerf_scale[k]; a.bt[k] = p->nerf_trans[k]; }
    a.n = p->n; a.n_per_b = p->n_per_b; a.B = p->B; a.H = p->H; a.W = p->W; a.C = p->C; a.D = p->D;
    return a;
}

static unsigned field_blocks(int64_t n)
{
    int64_t blocks = (n + 3) / 4;
    const int64_t cap = (int64_t)hav_num_cus() * 32;
    return (unsigned)(blocks > cap ? cap : blocks);
}

extern "C" int hav_field_inputs_fwd(float* X, const HavFieldParams* p, const float* pts, const float* inv_T, const float* vol,
                                    const float* planes_cl, void* stream)
{
    int rc = field_check(p, pts, inv_T, vol, planes_cl);
    if (rc || !X) return rc ? rc : HAV_EINVAL;
    if (p->n == 0) return 0;
    FieldArgs a = field_args(p, pts, inv_T, vol, planes_cl);
    a.X = X;
    // HAVATAR_FIELD_BWD=walk|taps keeps the one-query-at-a-time kernels (A/B runs); read once
    static const bool serial = [] { const char* e = getenv("HAVATAR_FIELD_BWD"); return e && (!strcmp(e, "taps") || !strcmp(e, "walk")); }();
    if (p->C <= 64 && !serial)
        hipLaunchKernelGGL(field_inputs_run_kernel<0>, dim3(field_blocks((p->n + FR_RUN - 1) / FR_RUN)), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(field_inputs_kernel<0>, dim3(field_blocks(p->n)), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

// X as bf16 rows [n, 2C+48] for hav_mlp_train_*_xbf16 (the MLP rounds fp32 rows to bf16 itself: same operands, half the bytes written here
// and read there twice).  C <= 64 (the run kernel).
extern "C" int hav_field_inputs_fwd_bf16(void* Xb, const HavFieldParams* p, const float* pts, const float* inv_T, const float* vol,
                                         const float* planes_cl, void* stream)
{
    int rc = field_check(p, pts, inv_T, vol, planes_cl);
    if (rc || !Xb) return rc ? rc : HAV_EINVAL;
    if (p->C > 64 || (p->C & 1)) return HAV_EUNSUP;
    if (p->n == 0) return 0;
    FieldArgs a = field_args(p, pts, inv_T, vol, planes_cl);
    a.Xb = (unsigned short*)Xb;
    hipLaunchKernelGGL(field_inputs_run_kernel<0>, dim3(field_blocks((p->n + FR_RUN - 1) / FR_RUN)), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

static int field_inputs_bwd_any(float* dplanes_cl, float* dvol, const float* dX, const HavFieldParams* p, const float* pts,
                                const float* inv_T, const float* vol, const float* planes_cl, int samples_per_ray, void* stream)
{
    int rc = field_check(p, pts, inv_T, vol, planes_cl);
    if (rc || !dX || (!dplanes_cl && !dvol)) return rc ? rc : HAV_EINVAL;
    if (p->n == 0) return 0;
    FieldArgs a = field_args(p, pts, inv_T, vol, planes_cl);
    a.dX = dX; a.dplanes = dplanes_cl; a.dvol = dvol;
    // rows of neighbouring rays: only when every batch element is whole rays of S samples in groups of 16 rays
    if (samples_per_ray > 0 && p->n == (int64_t)p->B * p->n_per_b && p->n_per_b % samples_per_ray == 0 && (p->n_per_b / samples_per_ray) % FI_RUN == 0)
        a.S = samples_per_ray;
    // HAVATAR_FIELD_BWD=taps keeps the one-row-of-atomics-per-tap kernel (A/B runs); read once
    static const bool per_tap = [] { const char* e = getenv("HAVATAR_FIELD_BWD"); return e && !strcmp(e, "taps"); }();
    // the 16-queries-at-a-time kernel is the forward's default; backward it is bound by its atomics and wave sums, not by latency, and runs
    // at 2 waves per SIMD: 1.73 vs 1.48 ms per call at config 5's size (tools/bench_field_inputs.py), so it only runs on request
    static const bool runs = [] { const char* e = getenv("HAVATAR_FIELD_BWD"); return e && !strcmp(e, "runs"); }();
    if (p->C <= 64 && !per_tap && runs)
        hipLaunchKernelGGL(field_inputs_run_kernel<2>, dim3(field_blocks((p->n + FR_RUN - 1) / FR_RUN)), dim3(256), 0, (hipStream_t)stream, a);
    else if (p->C <= 64 && !per_tap)
        hipLaunchKernelGGL(field_inputs_kernel<2>, dim3(field_blocks((p->n + FI_RUN - 1) / FI_RUN)), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(field_inputs_kernel<1>, dim3(field_blocks(p->n)), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}
extern "C" int hav_field_inputs_bwd(float* dplanes_cl, float* dvol, const float* dX, const HavFieldParams* p, const float* pts,
                                    const float* inv_T, const float* vol, const float* planes_cl, void* stream)
{
    return field_inputs_bwd_any(dplanes_cl, dvol, dX, p, pts, inv_T, vol, planes_cl, 0, stream);
}
// The same gradients for queries laid out [B][rays][samples_per_ray] whose rays come in image rows (neighbouring pixels are neighbouring rays:
// the training patch): the scatter merges the taps of 16 neighbouring rays per depth instead of 16 depths per ray.  Same sums, other order.
extern "C" int hav_field_inputs_bwd_rows(float* dplanes_cl, float* dvol, const float* dX, const HavFieldParams* p, const float* pts,
                                         const float* inv_T, const float* vol, const float* planes_cl, int samples_per_ray, void* stream)
{
    return field_inputs_bwd_any(dplanes_cl, dvol, dX, p, pts, inv_T, vol, planes_cl, samples_per_ray, stream);
}

static int fixed_bits(int64_t n)
{
    int lg = 0;
    while (((int64_t)1 << lg) < n * 4) ++lg;          // <= 4 n contributions can meet in one accumulator (merged runs count once each)
    const int fb = 61 - lg;
    return fb > 40 ? 40 : (fb < 8 ? 8 : fb);
}
extern "C" int64_t hav_field_inputs_bwd_fixed_scratch_bytes(const HavFieldParams* p)
{
    if (!p || p->n < 0) return 0;
    const int64_t planes = 2LL * p->B * p->H * p->W * p->C, vol = 2LL * p->D * p->D * p->D;
    // [planes] + [vol] 64-bit accumulators | [n][16] tap values | [n][16] voxel ids | 1 maximum word (+ pad)
    return (planes + vol) * 8 + p->n * 16 * 8 + 256;
}
// Bit-reproducible form of hav_field_inputs_bwd (ABI 6): same gradients, summed as 64-bit fixed-point integers (see fold_max_exp above), so two
// runs on the same inputs give the same bits.  dx_amax: the words of hav_absmax(dX).  scratch: hav_field_inputs_bwd_fixed_scratch_bytes()
// bytes, zeroed by this call.  C <= 64 (the tap-merging walk).
extern "C" int hav_field_inputs_bwd_fixed(float* dplanes_cl, float* dvol, const float* dX, const void* dx_amax, void* scratch, const HavFieldParams* p,
                                          const float* pts, const float* inv_T, const float* vol, const float* planes_cl, void* stream)
{
    int rc = field_check(p, pts, inv_T, vol, planes_cl);
    if (rc || !dX || !dx_amax || !scratch || (!dplanes_cl && !dvol)) return rc ? rc : HAV_EINVAL;
    if (p->C > 64) return HAV_EUNSUP;
    if (p->n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_pl = 2LL * p->B * p->H * p->W * p->C, n_vol = 2LL * p->D * p->D * p->D;
    long long* fpl = (long long*)scratch;
    long long* fvol = fpl + n_pl;
    float* vval = (float*)(fvol + n_vol);
    int* vidx = (int*)(vval + p->n * 16);
    unsigned int* vmax = (unsigned int*)(vidx + p->n * 16);
    hipError_t me = hipMemsetAsync(fpl, 0, (size_t)(n_pl + n_vol) * 8, st);
    if (me == hipSuccess) me = hipMemsetAsync(vmax, 0, 256, st);
    if (me != hipSuccess) return (int)me;
    FieldArgs a = field_args(p, pts, inv_T, vol, planes_cl);
    a.dX = dX; a.fplanes = dplanes_cl ? fpl : nullptr; a.dx_amax = (const unsigned int*)dx_amax; a.fixbits = fixed_bits(p->n);
    a.vval = dvol ? vval : nullptr; a.vidx32 = vidx; a.vmax = dvol ? vmax : nullptr;
    hipLaunchKernelGGL(field_inputs_kernel<4>, dim3(field_blocks((p->n + FI_RUN - 1) / FI_RUN)), dim3(256), 0, st, a);
    HAV_LAUNCH_CHECK();
    const unsigned cb = (unsigned)hav_num_cus() * 8;
    if (dvol) {
        hipLaunchKernelGGL(vol_fixed_scatter_kernel, dim3(cb), dim3(256), 0, st, fvol, (const float*)vval, (const int*)vidx, p->n * 16, (const unsigned int*)vmax, a.fixbits);
        HAV_LAUNCH_CHECK();
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3(cb), dim3(256), 0, st, dvol, (const long long*)fvol, n_vol, (const unsigned int*)vmax, 1, a.fixbits);
        HAV_LAUNCH_CHECK();
    }
    if (dplanes_cl) {
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3(cb), dim3(256), 0, st, dplanes_cl, (const long long*)fpl, n_pl, (const unsigned int*)dx_amax, 256, a.fixbits);
        HAV_LAUNCH_CHECK();
    }
    return 0;
}

// ================================================================================================
// Compositing: volume_render_radiance_field(act_feat=False) with gradients (utils/nerf_util.py:28-73)
// ================================================================================================
struct CompArgs {
    float* rgb; float* acc; float* weights; float* depth;                  // fwd out
    float* d_rf;                                                          // bwd out
    const float* d_rgb; const float* d_acc; const float* d_w; const float* d_depth;
    const float* rf; const float* z; const float* rd; const float* noise; const float* bg;
    int64_t n_rays;
    int S, CH, nsig;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int MODE>
__global__ void __launch_bounds__(256) composite_kernel(CompArgs a)
{
    __shared__ float sw_[4][64], sd_[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* sw = sw_[wv]; float* sd = sd_[wv];
    const int S = a.S, CH = a.CH, RW = a.CH + 1;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    extern __shared__ __attribute__((aligned(16))) float srf_all[];          // MODE 2: [waves][S][RW]
    for (int64_t r = wave0; r < a.n_rays; r += nwaves) {
        const float* rf = a.rf + (size_t)r * S * RW;
        if (MODE == 2) {
            // backward: the lane = sample phase below walks a row of CH + 1 values per lane -- straight from memory that is 64 cache lines per
            // load instruction, the same ones 17 times over, with every resident wave's 17 KB block competing for the 32 KB L1 (0.9 TB/s).
            // The block is copied once, coalesced, into LDS (row pitch CH + 1 = 69 words: odd, conflict-free for both walks) and read there.
            float* srf = srf_all + (size_t)wv * S * RW;
            const int nel = S * RW;
            __builtin_amdgcn_wave_barrier();
            if ((nel & 3) == 0 && ((reinterpret_cast<uintptr_t>(rf) & 15) == 0)) {
                // 6 vectors per lane in flight (a serial load -> store loop is one memory round trip per 1 KiB)
                const int n4 = nel >> 2;
                for (int e0 = 0; e0 < n4; e0 += 6 * 64) {
                    float4 v[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) { const int e = e0 + 64 * u + lane; v[u] = e < n4 ? reinterpret_cast<const float4*>(rf)[e] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
                    for (int u = 0; u < 6; ++u) { const int e = e0 + 64 * u + lane; if (e < n4) reinterpret_cast<float4*>(srf)[e] = v[u]; }
                }
            } else {
                for (int e = lane; e < nel; e += 64) srf[e] = rf[e];
            }
            __builtin_amdgcn_wave_barrier();
            rf = srf;
        }
        // ---- lane = sample: alpha, transmittance, weight (:36-60)
        const bool on = lane < S;
        const int li = on ? lane : S - 1;
        const float zi = a.z[r * S + li];
        const int i0 = li < S - 1 ? li : S - 2;
        const float dz = S > 1 ? a.z[r * S + i0 + 1] - a.z[r * S + i0] : 0.f;       // the last distance is repeated (:37)
        const float dx = a.rd[r * 3], dy = a.rd[r * 3 + 1], dzz = a.rd[r * 3 + 2];
        const float dist = dz * sqrtf(dx * dx + dy * dy + dzz * dzz);
        const float raw = rf[(size_t)li * RW + CH] + (a.noise ? a.noise[r * S + li] : 0.f);
        const float sigma = fmaxf(raw, 0.f);
        const float alpha = 1.0f - expf(-sigma * dist);
        const float tt = on ? (1.0f - alpha) + 1e-10f : 1.0f;
        float incl = tt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float u = __shfl_up(incl, o, 64);
            if (lane >= o) incl *= u;
        }
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float w = on ? alpha * excl : 0.f;
        const float acc = wsum64(w);
        __builtin_amdgcn_wave_barrier();      // sw/sd are private to the wave: LDS executes a wave's accesses in order
        sw[lane] = w;
        if (MODE == 0) {
            const float depth = wsum64(w * zi);
            if (on) a.weights[r * S + lane] = w;
            if (lane == 0) { a.acc[r] = acc; a.depth[r] = depth; }
            __builtin_amdgcn_wave_barrier();
            // ---- lane = channel: rgb_map = sum_i w_i c_i (:62-63), white-background term on the first three (:70-71)
            for (int c = lane; c < CH; c += 64) {
                float s = 0.f;
                for (int i = 0; i < S; ++i) {
                    const float v = rf[(size_t)i * RW + c];
                    s += sw[i] * (c < a.nsig ? sigmoidf_(v) : v);
                }
                if (a.bg && c < 3) s += (1.0f - acc) * a.bg[r * 3 + c];
                a.rgb[r * CH + c] = s;
            }
        } else {
            // ---- G_i = d loss / d w_i
            const float* drgb = a.d_rgb + (size_t)r * CH;
            float dacc = a.d_acc ? a.d_acc[r] : 0.f;
            if (a.bg)
                for (int c = 0; c < 3 && c < CH; ++c) dacc -= drgb[c] * a.bg[r * 3 + c];
            float G = dacc + (a.d_depth ? a.d_depth[r] * zi : 0.f) + (a.d_w && on ? a.d_w[r * S + li] : 0.f);
            {
                const float* row = rf + (size_t)li * RW;
                float dot = 0.f;
                for (int c = 0; c < CH; ++c) {
                    const float v = row[c];
                    dot += drgb[c] * (c < a.nsig ? sigmoidf_(v) : v);
                }
                G += dot;
            }
            if (!on) G = 0.f;
            // suffix sum of G_j w_j over j > i
            float suf = G * w;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const float u = __shfl_down(suf, o, 64);
                if (lane + o < 64) suf += u;
            }
            suf -= G * w;
            const float dalpha = G * excl - suf / tt;
            const float dsig = dalpha * dist * (1.0f - alpha);
            sd[lane] = (on && raw > 0.f) ? dsig : 0.f;
            __builtin_amdgcn_wave_barrier();
            // ---- lane = channel: d c_i = w_i d_rgb (sigmoid' on the first nsig), d raw_i in the last column
            float* srf_w = MODE == 2 ? srf_all + (size_t)wv * S * RW : nullptr;          // MODE 2: the gradient block replaces the staged one in place
            for (int c = lane; c < RW; c += 64) {
                const float g = c < CH ? drgb[c] : 0.f;
                for (int i = 0; i < S; ++i) {
                    float o;
                    if (c == CH) o = sd[i];
                    else if (c < a.nsig) { const float sg = sigmoidf_(rf[(size_t)i * RW + c]); o = sw[i] * g * sg * (1.0f - sg); }
                    else o = sw[i] * g;
                    if (MODE == 2) srf_w[i * RW + c] = o;
                    else a.d_rf[((size_t)r * S + i) * RW + c] = o;
                }
            }
            if (MODE == 2) {          // ... and leaves as whole 16-byte vectors, a contiguous KiB per wave and store
                float* dst = a.d_rf + (size_t)r * S * RW;
                const int nel = S * RW;
                __builtin_amdgcn_wave_barrier();
                if ((nel & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                    for (int e = lane; e < (nel >> 2); e += 64) reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(srf_w)[e];
                } else {
                    for (int e = lane; e < nel; e += 64) dst[e] = srf_w[e];
                }
            }
        }
    }
}

static unsigned comp_blocks(int64_t n_rays)
{
    int64_t blocks = (n_rays + 3) / 4;
    const int64_t cap = (int64_t)hav_num_cus() * 16;
    return (unsigned)(blocks > cap ? cap : blocks);
}

extern "C" int hav_composite_fwd(float* rgb, float* acc, float* weights, float* depth, const float* rf, const float* z, const float* rd,
                                 const float* noise, const float* bg, int64_t n_rays, int S, int CH, int n_sigmoid, void* stream)
{
    if (!rgb || !acc || !weights || !depth || !rf || !z || !rd || n_rays < 0 || S < 1 || CH < 1 || n_sigmoid < 0) return HAV_EINVAL;
    if (S > 64) return HAV_EUNSUP;
    if (n_rays == 0) return 0;
    CompArgs a{};
    a.rgb = rgb; a.acc = acc; a.weights = weights; a.depth = depth; a.rf = rf; a.z = z; a.rd = rd; a.noise = noise; a.bg = bg;
    a.n_rays = n_rays; a.S = S; a.CH = CH; a.nsig = n_sigmoid;
    hipLaunchKernelGGL(composite_kernel<0>, dim3(comp_blocks(n_rays)), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_composite_bwd(float* d_rf, const float* d_rgb, const float* d_acc, const float* d_weights, const float* d_depth,
                                 const float* rf, const float* z, const float* rd, const float* noise, const float* bg, int64_t n_rays,
                                 int S, int CH, int n_sigmoid, void* stream)
{
    if (!d_rf || !d_rgb || !rf || !z || !rd || n_rays < 0 || S < 1 || CH < 1 || n_sigmoid < 0) return HAV_EINVAL;
    if (S > 64) return HAV_EUNSUP;
    if (n_rays == 0) return 0;
    CompArgs a{};
    a.d_rf = d_rf; a.d_rgb = d_rgb; a.d_acc = d_acc; a.d_w = d_weights; a.d_depth = d_depth;
    a.rf = rf; a.z = z; a.rd = rd; a.noise = noise; a.bg = bg;
    a.n_rays = n_rays; a.S = S; a.CH = CH; a.nsig = n_sigmoid;
    // the staged form: 2 waves per workgroup, S x (CH + 1) floats of LDS per wave (35 KB per workgroup at config 5's 64 x 69); larger blocks and
    // HAVATAR_COMPOSITE_BWD=direct take the form that reads the rows from memory
    static const bool direct = [] { const char* e = getenv("HAVATAR_COMPOSITE_BWD"); return e && !strcmp(e, "direct"); }();
    const size_t lds = (size_t)2 * S * (CH + 1) * sizeof(float);
    if (!direct && lds <= 48 * 1024) {
        int64_t blocks = (n_rays + 1) / 2;
        const int64_t cap = (int64_t)hav_num_cus() * 32;
        hipLaunchKernelGGL(composite_kernel<2>, dim3((unsigned)(blocks > cap ? cap : blocks)), dim3(128), lds, (hipStream_t)stream, a);
    } else
        hipLaunchKernelGGL(composite_kernel<1>, dim3(comp_blocks(n_rays)), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// Trilinear x2 up-sampling of a [N,C,D,H,W] volume, forward and adjoint -- the parameter-free first stage of every
// UpConv3DBlock of the skinning-volume decoder (reference model/network/voxel_encoder.py:183-210: nn.Upsample(scale_factor=2,
// mode='trilinear'), align_corners=False).  Per axis  out[2i] = 0.25 x[i-1] + 0.75 x[i],  out[2i+1] = 0.75 x[i] + 0.25 x[i+1]
// with indices clamped at the borders; the 3-D operator is the tensor product.  One thread per output element, gather form both
// ways (no atomics): the adjoint collects, per axis, the <= 4 outputs an input voxel feeds (2i-1, 2i, 2i+1, 2i+2 with the border
// terms folded back).  Replaces ~30 ATen launches per block forward and ~60 backward (six blocks, evaluated per training step).
// ================================================================================================
__device__ __forceinline__ void up2_taps(int o, int n, int& i0, int& i1, float& w0, float& w1)
{
    const int i = o >> 1;
    if (o & 1) { i0 = i; i1 = min(i + 1, n - 1); w0 = 0.75f; w1 = 0.25f; }
    else { i0 = max(i - 1, 0); i1 = i; w0 = 0.25f; w1 = 0.75f; }
}

__global__ void __launch_bounds__(256) up3d_fwd_kernel(float* __restrict__ out, const float* __restrict__ in, int64_t NC, int D, int H, int W)
{
    const int64_t total = NC * 8 * D * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = idx;
        const int ox = (int)(t % (2 * W)); t /= 2 * W;
        const int oy = (int)(t % (2 * H)); t /= 2 * H;
        const int oz = (int)(t % (2 * D));
        const int64_t nc = t / (2 * D);
        int x0, x1, y0, y1, z0, z1; float wx0, wx1, wy0, wy1, wz0, wz1;
        up2_taps(ox, W, x0, x1, wx0, wx1); up2_taps(oy, H, y0, y1, wy0, wy1); up2_taps(oz, D, z0, z1, wz0, wz1);
        const float* p = in + nc * D * H * W;
        auto at = [&](int z, int y, int x) { return p[((int64_t)z * H + y) * W + x]; };
        // same association as three 1-D passes along D, then H, then W
        const float a00 = wz0 * at(z0, y0, x0) + wz1 * at(z1, y0, x0), a01 = wz0 * at(z0, y0, x1) + wz1 * at(z1, y0, x1);
        const float a10 = wz0 * at(z0, y1, x0) + wz1 * at(z1, y1, x0), a11 = wz0 * at(z0, y1, x1) + wz1 * at(z1, y1, x1);
        const float b0 = wy0 * a00 + wy1 * a10, b1 = wy0 * a01 + wy1 * a11;
        out[idx] = wx0 * b0 + wx1 * b1;
    }
}

// adjoint along one axis: input index i collects  sum_o w(o -> i) g[o]  over the outputs that read it
__device__ __forceinline__ int up2_adj(int i, int n, int (&o)[4], float (&w)[4])
{
    int k = 0;
    // even output 2j reads (max(j-1,0): 0.25, j: 0.75); odd output 2j+1 reads (j: 0.75, min(j+1,n-1): 0.25)
    o[k] = 2 * i; w[k++] = 0.75f;                           // out[2i]   <- 0.75 x[i]
    o[k] = 2 * i + 1; w[k++] = 0.75f;                       // out[2i+1] <- 0.75 x[i]
    if (i + 1 < n) { o[k] = 2 * (i + 1); w[k++] = 0.25f; }  // out[2(i+1)] <- 0.25 x[i]
    else { o[k] = 2 * i + 1; w[k++] = 0.25f; }              // border: out[2n-1] <- 0.25 x[min(n, n-1)]
    if (i > 0) { o[k] = 2 * (i - 1) + 1; w[k++] = 0.25f; }  // out[2(i-1)+1] <- 0.25 x[i]
    else { o[k] = 0; w[k++] = 0.25f; }                      // border: out[0] <- 0.25 x[max(-1, 0)]
    return k;
}

__global__ void __launch_bounds__(256) up3d_bwd_kernel(float* __restrict__ din, const float* __restrict__ dout, int64_t NC, int D, int H, int W)
{
    const int64_t total = NC * D * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = idx;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H); t /= H;
        const int z = (int)(t % D);
        const int64_t nc = t / D;
        int ox[4], oy[4], oz[4]; float wx[4], wy[4], wz[4];
        up2_adj(x, W, ox, wx); up2_adj(y, H, oy, wy); up2_adj(z, D, oz, wz);
        const float* g = dout + nc * 8 * D * H * W;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float sy = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float sx = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) sx += wx[c] * g[((int64_t)oz[a] * (2 * H) + oy[b]) * (2 * W) + ox[c]];
                sy += wy[b] * sx;
            }
            s += wz[a] * sy;
        }
        din[idx] = s;
    }
}

extern "C" int hav_upsample3d_2x_fwd(float* out, const float* in, int64_t NC, int D, int H, int W, void* stream)
{
    if (!out || !in || NC < 0 || D < 1 || H < 1 || W < 1) return HAV_EINVAL;
    const int64_t total = NC * 8 * D * H * W;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)hav_num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(up3d_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, in, NC, D, H, W);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_upsample3d_2x_bwd(float* din, const float* dout, int64_t NC, int D, int H, int W, void* stream)
{
    if (!din || !dout || NC < 0 || D < 1 || H < 1 || W < 1) return HAV_EINVAL;
    const int64_t total = NC * D * H * W;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)hav_num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(up3d_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, din, dout, NC, D, H, W);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// Conv3d 3^3 / padding 1 on SMALL volumes (the first layers of the skinning-volume decoder, reference model/network/voxel_encoder.py:183-210:
// 1024 -> 512 channels on 2^3 voxels, 512 -> 256 on 4^3, 256 -> 128 on 8^3) as matrix products over an explicit patch matrix:
//   col[(i, t)][p] = x[i][p + off(t)]  (0 outside)   K = 27 Cin rows, P = R^3 columns: 0.9 / 3.5 / 14 MB
//   y = W[Cout, K] . col,  dW = dy . col^T,  dcol = W^T . dy,  dx[i][q] = sum_t dcol[(i, t)][q - off(t)]
// The three products are plain GEMMs (rocBLAS through ATen: W is streamed once each, the layers are weight-bound: 56 / 14 / 3.5 MB);
// MIOpen / CK take 350 / 200 / 115 us for the forward and 120 us for the gradients of each of these layers.  These two kernels build the
// patch matrix and fold its gradient back (gather form both ways: no atomics).
// ================================================================================================
__global__ void __launch_bounds__(256) im2col3d_kernel(float* __restrict__ col, const float* __restrict__ x, int C, int R)
{
    const int P = R * R * R;
    const int64_t total = (int64_t)C * 27 * P;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(e % P);
        const int64_t k = e / P;
        const int t = (int)(k % 27), i = (int)(k / 27);
        const int pz = p / (R * R), py = (p / R) % R, px = p % R;
        const int z = pz + t / 9 - 1, y = py + (t / 3) % 3 - 1, xx = px + t % 3 - 1;
        col[e] = (z >= 0 && z < R && y >= 0 && y < R && xx >= 0 && xx < R) ? x[(int64_t)i * P + (z * R + y) * R + xx] : 0.f;
    }
}
__global__ void __launch_bounds__(256) col2im3d_kernel(float* __restrict__ dx, const float* __restrict__ dcol, int C, int R)
{
    const int P = R * R * R;
    const int64_t total = (int64_t)C * P;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % P), i = (int)(e / P);
        const int qz = q / (R * R), qy = (q / R) % R, qx = q % R;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            // x[q] fed col[(i, t)][p] with p = q - off(t)
            const int z = qz - (t / 9 - 1), y = qy - ((t / 3) % 3 - 1), xx = qx - (t % 3 - 1);
            if (z >= 0 && z < R && y >= 0 && y < R && xx >= 0 && xx < R) s += dcol[((int64_t)i * 27 + t) * P + (z * R + y) * R + xx];
        }
        dx[e] = s;
    }
}
extern "C" int hav_im2col3d(float* col, const float* x, int C, int R, void* stream)
{
    if (!col || !x || C < 1 || R < 1) return HAV_EINVAL;
    const int64_t total = (int64_t)C * 27 * R * R * R;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 32) blocks = (int64_t)hav_num_cus() * 32;
    hipLaunchKernelGGL(im2col3d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, col, x, C, R);
    HAV_LAUNCH_CHECK();
    return 0;
}
extern "C" int hav_col2im3d(float* dx, const float* dcol, int C, int R, void* stream)
{
    if (!dx || !dcol || C < 1 || R < 1) return HAV_EINVAL;
    const int64_t total = (int64_t)C * R * R * R;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)hav_num_cus() * 32) blocks = (int64_t)hav_num_cus() * 32;
    hipLaunchKernelGGL(col2im3d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dx, dcol, C, R);
    HAV_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Importance resampling of the training pass (model/nerf_trainer.py:166-170 + utils/nerf_util.py:76-117 of the reference):
//   z_mid = .5 (z[1:] + z[:-1]);  z_s = sample_pdf(z_mid, weights[1:-1], S_f, det);  z2 = sort(cat(z[::2], z_s))
// as ONE launch instead of ~30 ATen launches (add, sum, div, cumsum, cat, arange / rand arithmetic, searchsorted, clamps, stack,
// 2 x gather, where, sub / div / mul / add, slice, cat, sort).  No gradient flows through it (the reference detaches z_samples).
// One wave per ray; the ray's depths, weights, CDF and the merged candidate list live in the wave's own LDS rows.
//   * sum and cumulative sum: sequential on lane 0 in index order -- the order of the CPU statement the oracle restates (SURVEY B-11);
//     ATen's device reductions use a tree whose shape depends on the launch geometry, there is no canonical order to match
//   * every other product / sum is rounded separately like the ATen element-wise ops (no contraction into FMAs)
//   * the pdf's divisions run one per lane between the two sequential passes; searchsorted(right=True) is a binary search (the CDF of
//     non-negative weights is non-decreasing in fp32 too); the sort is a rank sort (stable on ties: the output equals torch.sort's values)
// ------------------------------------------------------------------------------------------------
#define RS_MAX 128
struct ResampleArgs {
    float* z2; float* zs; const float* z; const float* w; const float* zeta;
    int64_t n; int S_c, S_f; float u_sN, u_w, u_st;
};
__global__ void __launch_bounds__(256) resample_depths_kernel(ResampleArgs a)
{
#pragma clang fp contract(off)      // plain operators below, none fused (the __f*_rn intrinsics are inline operators compiled with contraction on)
    __shared__ float s_z[4][RS_MAX], s_w[4][RS_MAX], s_cdf[4][RS_MAX], s_cand[4][RS_MAX];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int S_c = a.S_c, S_f = a.S_f, nb = S_c - 1, nw = S_c - 2, S_half = (S_c + 1) >> 1, S_fp = S_half + S_f;
    float* zc = s_z[wv]; float* w = s_w[wv]; float* cdf = s_cdf[wv]; float* cand = s_cand[wv];
    for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < a.n; r += (int64_t)gridDim.x * 4) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < S_c; i += 64) { zc[i] = a.z[r * S_c + i]; w[i] = a.w[r * S_c + i] + 1e-5f; }   // :79
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {                                   // sum in index order; w[0] (not part of weights[1:-1]) carries it to the other lanes
            float sum = 0.f;
            for (int i = 0; i < nw; ++i) sum = sum + w[1 + i];                                  // :80
            w[0] = sum;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            const float sum = w[0];
            for (int i = lane; i < nw; i += 64) w[1 + i] = w[1 + i] / sum;                      // pdf, :80 (one division per lane)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {                                   // cumulative sum in index order, :81-84
            float run = 0.f;
            cdf[0] = 0.f;
            for (int i = 0; i < nw; ++i) { run = run + w[1 + i]; cdf[i + 1] = run; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < S_f; k += 64) {
            float u;
            if (!a.zeta)         // det: torch.linspace(0, 1, S_f) -- start + step k below the middle, end - step (S_f-1-k) above it
                { const float lo = a.u_st * (float)k, hi = a.u_st * (float)(S_f - 1 - k); u = (S_f == 1) ? 0.f : ((k < S_f / 2) ? lo : 1.0f - hi); }
            else                 // arange * s + rand * (s - 1e-6)   (:93-95)
                { const float ks = (float)k * a.u_sN, zw = a.zeta[r * S_f + k] * a.u_w; u = ks + zw; }
            int inds = 0, hi_ = nb;                        // :102 searchsorted(right=True): the CDF is non-decreasing (pdf >= 0)
            while (inds < hi_) { const int mid = (inds + hi_) >> 1; if (cdf[mid] <= u) inds = mid + 1; else hi_ = mid; }
            const int below = max(inds - 1, 0), above = min(inds, nb - 1);                      // :103-104
            const float c0 = cdf[below], c1 = cdf[above];
            float dnm = c1 - c0;                                                      // :112
            if (dnm < 1e-5f) dnm = 1.0f;                                                        // :113
            const float num = u - c0, tt = num / dnm;                                  // :114
            const float sb = zc[below + 1] + zc[below], bl = 0.5f * sb;              // z_vals_mid (nerf_trainer.py:166)
            const float sa = zc[above + 1] + zc[above], ba = 0.5f * sa;
            const float dd = ba - bl, md = tt * dd, v = bl + md;                    // :115
            cand[S_half + k] = v;
            if (a.zs) a.zs[r * S_f + k] = v;
        }
        for (int i = lane; i < S_half; i += 64) cand[i] = zc[2 * i];                            // z_vals[:, ::2]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < S_fp; e += 64) {
            const float v = cand[e];
            int rank = 0;
            for (int q = 0; q < S_fp; ++q) {
                const float o = cand[q];
                // a TOTAL order, NaN last with the index as tie-break (what torch.sort does): every output slot is written exactly once even
                // when a diverged step hands NaN depths / weights in (with `o < v || (o == v && q < e)` alone all NaNs ranked 0 and collided)
                const bool before = (o == o) ? ((v != v) || o < v || (o == v && q < e)) : ((v != v) && q < e);
                rank += before ? 1 : 0;
            }
            a.z2[r * S_fp + rank] = v;
        }
    }
}
extern "C" int hav_resample_depths(float* z2, float* z_samples, const float* z, const float* weights, const float* zeta, int64_t n_rays,
                                   int S_c, int S_f, void* stream)
{
    if (n_rays < 0 || S_c < 3 || S_f < 1) return HAV_EINVAL;
    if (S_c > RS_MAX || ((S_c + 1) >> 1) + S_f > RS_MAX) return HAV_EUNSUP;
    if (n_rays == 0) return 0;                           // (an empty tensor's pointer is null)
    if (!z2 || !z || !weights) return HAV_EINVAL;
    ResampleArgs a{};
    a.z2 = z2; a.zs = z_samples; a.z = z; a.w = weights; a.zeta = zeta; a.n = n_rays; a.S_c = S_c; a.S_f = S_f;
    a.u_sN = (float)(1.0 / (double)S_f);                 // s = 1 / num_samples is a Python double, cast at the multiplication
    a.u_w = (float)(1.0 / (double)S_f - 1e-6);
    a.u_st = S_f > 1 ? 1.0f / (float)(S_f - 1) : 0.f;    // torch.linspace's step, computed in float
    int64_t blocks = (n_rays + 3) / 4;
    const int64_t cap = (int64_t)hav_num_cus() * 32;
    hipLaunchKernelGGL(resample_depths_kernel, dim3((unsigned)(blocks > cap ? cap : blocks)), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// EqualLinear without activation under autograd (model/styleUnet.py:128-162 of the reference: F.linear(x, weight * scale, bias * lr_mul)) --
// the modulation layer of every ModulatedConv2d: x [B, in] (B <= 8, in = the style width, 32 on this path), W [out, in].  The ATen
// statement is 3 launches forward (two scalar products over the parameters, addmm) and 6-7 backward (two products, two GEMMs, a sum,
// accumulation) per layer, 26 layers per optimisation step: one launch each way here.
//   forward : y[b,o] = sum_i x[b,i] * fl(W[o,i] * scale) + fl(bias[o] * lr_mul)          (products of the scaled weight, like the reference)
//   backward: dW[o,i] = scale * sum_b dy[b,o] x[b,i];  db[o] = lr_mul * sum_b dy[b,o];  dx[b,i] = sum_o dy[b,o] * fl(W[o,i] * scale)
// All sums run in a fixed order (bit-reproducible); dx is a wave per (b, i) with a fixed butterfly.
// ------------------------------------------------------------------------------------------------
#define EQL_MAXB 8
struct EqLinArgs {
    float* y; float* dx; float* dW; float* db;
    const float* x; const float* W; const float* b; const float* dy;
    float scale, lr_mul; int B, in, out, nA;
};
// x goes through LDS in chunks of `chunk` columns (B * chunk <= 2048 floats, chunk a multiple of 32; the path's style width of 32 is one
// chunk); the weight row is read 8 x 16 bytes at a time when rows are 16-byte aligned (in % 4 == 0), so that one round trip to memory
// covers 32 weights.  The sum runs over i in ascending order either way.
__global__ void __launch_bounds__(64) equal_linear_fwd_kernel(EqLinArgs a)
{
    __shared__ float sx[2048];
    const int o = blockIdx.x * 64 + threadIdx.x;
    const bool active = o < a.out;
    const int chunk = (2048 / a.B) & ~31;
    float acc[EQL_MAXB];
#pragma unroll
    for (int b = 0; b < EQL_MAXB; ++b) acc[b] = 0.f;
    const float* wr = a.W + (size_t)(active ? o : 0) * a.in;
    const bool vec = (a.in & 3) == 0 && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0;
    for (int c0 = 0; c0 < a.in; c0 += chunk) {
        const int cn = min(chunk, a.in - c0);
        if (c0) __syncthreads();
        for (int e = threadIdx.x; e < a.B * cn; e += 64) { const int b = e / cn, i = e - b * cn; sx[b * chunk + i] = a.x[b * a.in + c0 + i]; }
        __syncthreads();
        if (!active) continue;
        int i0 = 0;
        if (vec) {
            const float4* wr4 = reinterpret_cast<const float4*>(wr + c0);
            const int n4 = cn >> 2;
            for (int c = 0; c + 8 <= n4; c += 8) {
                float4 wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = wr4[c + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float w4[4] = {wv[j].x * a.scale, wv[j].y * a.scale, wv[j].z * a.scale, wv[j].w * a.scale};
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int b = 0; b < EQL_MAXB; ++b)
                            if (b < a.B) acc[b] = fmaf(sx[b * chunk + (c + j) * 4 + t], w4[t], acc[b]);
                }
            }
            i0 = (n4 & ~7) << 2;
        }
        for (int i = i0; i < cn; ++i) {
            const float w = wr[c0 + i] * a.scale;
#pragma unroll
            for (int b = 0; b < EQL_MAXB; ++b)
                if (b < a.B) acc[b] = fmaf(sx[b * chunk + i], w, acc[b]);
        }
    }
    if (!active) return;
    const float bias = a.b ? a.b[o] * a.lr_mul : 0.f;
#pragma unroll
    for (int b = 0; b < EQL_MAXB; ++b)
        if (b < a.B) a.y[(size_t)b * a.out + o] = acc[b] + bias;
}
// wide inputs (in >= 128; not on this path, whose style width is 32): a wave per output, lanes across the row (coalesced 256-byte reads of W
// and x), one butterfly per batch row.  A thread per output would make every load instruction touch 64 different rows.
__global__ void __launch_bounds__(256) equal_linear_fwd_wave_kernel(EqLinArgs a)
{
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= a.out) return;
    float acc[EQL_MAXB];
#pragma unroll
    for (int b = 0; b < EQL_MAXB; ++b) acc[b] = 0.f;
    const float* wr = a.W + (size_t)o * a.in;
    for (int i = lane; i < a.in; i += 64) {
        const float w = wr[i] * a.scale;
#pragma unroll
        for (int b = 0; b < EQL_MAXB; ++b)
            if (b < a.B) acc[b] = fmaf(a.x[b * a.in + i], w, acc[b]);
    }
    const float bias = a.b ? a.b[o] * a.lr_mul : 0.f;
#pragma unroll
    for (int b = 0; b < EQL_MAXB; ++b)
        if (b < a.B) {
            const float t = wsum64(acc[b]);
            if (lane == 0) a.y[(size_t)b * a.out + o] = t + bias;
        }
}
__global__ void __launch_bounds__(256) equal_linear_bwd_kernel(EqLinArgs a)
{
    if ((int)blockIdx.x < a.nA) {                                  // dW, db: one thread per weight
        const int e = blockIdx.x * 256 + threadIdx.x;
        if (e >= a.out * a.in) return;
        const int o = e / a.in, i = e - o * a.in;
        float g = 0.f, gb = 0.f;
        for (int b = 0; b < a.B; ++b) { const float d = a.dy[(size_t)b * a.out + o]; g = fmaf(d, a.x[b * a.in + i], g); gb += d; }
        if (a.dW) a.dW[e] = g * a.scale;
        if (a.db && i == 0) a.db[o] = gb * a.lr_mul;
        return;
    }
    if (!a.dx) return;
    const int q = ((int)blockIdx.x - a.nA) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;     // dx: one wave per (b, i)
    if (q >= a.B * a.in) return;
    const int b = q / a.in, i = q - b * a.in;
    float acc = 0.f;
    for (int o = lane; o < a.out; o += 64) acc = fmaf(a.dy[(size_t)b * a.out + o], a.W[(size_t)o * a.in + i] * a.scale, acc);
    acc = wsum64(acc);
    if (lane == 0) a.dx[q] = acc;
}
extern "C" int hav_equal_linear_fwd(float* y, const float* x, const float* W, const float* bias, float scale, float lr_mul, int B, int in_dim,
                                    int out_dim, void* stream)
{
    if (!y || !x || !W || B < 1 || in_dim < 1 || out_dim < 1) return HAV_EINVAL;
    if (B > EQL_MAXB || in_dim > 4096 || (int64_t)in_dim * out_dim > (1 << 24)) return HAV_EUNSUP;
    EqLinArgs a{};
    a.y = y; a.x = x; a.W = W; a.b = bias; a.scale = scale; a.lr_mul = lr_mul; a.B = B; a.in = in_dim; a.out = out_dim;
    if (in_dim >= 128)
        hipLaunchKernelGGL(equal_linear_fwd_wave_kernel, dim3((out_dim + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(equal_linear_fwd_kernel, dim3((out_dim + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}
extern "C" int hav_equal_linear_bwd(float* dx, float* dW, float* dbias, const float* dy, const float* x, const float* W, float scale, float lr_mul,
                                    int B, int in_dim, int out_dim, void* stream)
{
    if (!dy || !x || !W || B < 1 || in_dim < 1 || out_dim < 1) return HAV_EINVAL;
    if (B > EQL_MAXB || in_dim > 4096 || (int64_t)in_dim * out_dim > (1 << 24)) return HAV_EUNSUP;
    if (!dx && !dW && !dbias) return 0;
    EqLinArgs a{};
    a.dx = dx; a.dW = dW; a.db = dbias; a.dy = dy; a.x = x; a.W = W; a.scale = scale; a.lr_mul = lr_mul; a.B = B; a.in = in_dim; a.out = out_dim;
    a.nA = (dW || dbias) ? (in_dim * out_dim + 255) / 256 : 0;
    const int nB = dx ? (B * in_dim + 3) / 4 : 0;
    hipLaunchKernelGGL(equal_linear_bwd_kernel, dim3(a.nA + nB), dim3(256), 0, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}
