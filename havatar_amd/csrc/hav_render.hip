// hav_render.hip -- the fused ray march for gfx950 (MI355X): ray sampling -> skinning-field lookup ->
// tri-plane gather -> positional encoding -> radiance MLP on the matrix cores -> alpha compositing ->
// inverse-CDF resampling -> second pass, in ONE kernel, with nothing but the ray inputs and the
// composited outputs touching HBM.
//
// Replaces Trainer.predict_and_render_radiance and its callees (reference file:line in include/havatar.h
// and next to each step below).  Design notes (DESIGN.md has the long form):
//
//  * Work unit = a PAIR of rays per wave64.  A wave evaluates 32 samples per MLP pass ("tile"): lanes
//    j = lane&31 are the samples, the two half-waves h = lane>>5 split each sample's 176 MLP inputs
//    (plane h's 64 channels + PE octaves 4h..4h+3), its hidden units and its gather work.
//    Rows of 16 lanes (= one DPP row) always belong to one ray, so the transmittance product is a
//    DPP row scan + a scalar carry; 64 coarse samples = 4 rows, 48 fine samples = 3 rows, and the odd
//    fine row of ray A shares a tile with the first row of ray B: no matrix-core slot is wasted.
//  * MLP in "transposed" form  H^T[feature][sample] = W[feature][k] . X^T[k][sample]  on
//    v_mfma_f32_32x32x2_f32 (exact fp32).  With this orientation the accumulator registers of layer
//    l ARE the B operands of layer l+1 (lane holds column = its sample; register r of a 32-row tile
//    holds rows (r&3)+8(r>>2)+4h, i.e. exactly the k-pair an MFMA step consumes from the two
//    half-waves), so activations never leave registers and never cross lanes between layers.  Only the
//    weights move: they are pre-permuted once into that k order ("fragment order", hav_mlp_pack) so
//    every A operand is one conflict-free 256-byte ds_read_b32 / global_load_dword per wave.
//  * fc_rgb o fc_rgbFeat has no activation in between (model/nerf_model.py:110-111), so the 3 rgb rows
//    are folded into the head: one 128 -> {64 feat, 3 rgb, 1 alpha} layer.
//  * Layer-1 weights (88 KB) are LDS-resident per workgroup; layer-2/head fragments stream from L2
//    (coalesced, 8.75 B/clk/CU).  One persistent workgroup (8 waves) per CU; XCD-aware ray assignment
//    keeps each XCD's L2 on one horizontal band of the image / tri-planes.
#include "hav_common.h"

// ------------------------------------------------------------------------------------------------
// packed weight blob (float offsets)
// ------------------------------------------------------------------------------------------------
#define HAV_HID 128
#define HAV_PC 64                      // channels per plane
#define HAV_IN (2 * HAV_PC + 48)       // 176
#define K1_STEPS (HAV_IN / 2)          // 88 MFMA k-steps in layer 1
#define K2_STEPS (HAV_HID / 2)         // 64
#define HEAD_TILES 3                   // 96 rows: 64 feat | 3 rgb | alpha | 28 zero rows
#define OFF_W1 0
#define OFF_W2 (OFF_W1 + K1_STEPS * 4 * 64)
#define OFF_WH (OFF_W2 + K2_STEPS * 4 * 64)
#define OFF_B1 (OFF_WH + K2_STEPS * HEAD_TILES * 64)
#define OFF_B2 (OFF_B1 + 128)
#define OFF_BH (OFF_B2 + 128)
#define BLOB_FLOATS (OFF_BH + 32 * HEAD_TILES)

extern "C" int64_t hav_mlp_blob_bytes(void) { return (int64_t)BLOB_FLOATS * 4; }

// input column consumed by half-wave h at layer-1 k-step t
__host__ __device__ inline int k1_col(int t, int h) { return t < HAV_PC ? 2 * t + h : 2 * HAV_PC + 24 * h + (t - HAV_PC); }
// hidden unit held by half-wave h in accumulator register r of row-tile mp
__host__ __device__ inline int acc_row(int mp, int r, int h) { return 32 * mp + (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ void __launch_bounds__(256) mlp_pack_kernel(float* __restrict__ blob, HavMlpWeights w)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= BLOB_FLOATS) return;
    float v = 0.f;
    if (e < OFF_W2) {
        const int l = e & 63, m = (e >> 6) & 3, t = e >> 8;
        v = w.W1[(32 * m + (l & 31)) * HAV_IN + k1_col(t, l >> 5)];
    } else if (e < OFF_WH) {
        const int q = e - OFF_W2;
        const int l = q & 63, m = (q >> 6) & 3, ks = q >> 8;
        v = w.W2[(32 * m + (l & 31)) * HAV_HID + acc_row(ks >> 4, ks & 15, l >> 5)];
    } else if (e < OFF_B1) {
        const int q = e - OFF_WH;
        const int l = q & 63, m = (q >> 6) % HEAD_TILES, ks = (q >> 6) / HEAD_TILES;
        const int row = 32 * m + (l & 31), col = acc_row(ks >> 4, ks & 15, l >> 5);
        if (row < 64) v = w.Wf[row * HAV_HID + col];
        else if (row < 67) {       // fc_rgb o fc_rgbFeat folded: (Wc Wf)[c][col]
            double s = 0.0;
            for (int k = 0; k < 64; ++k) s += (double)w.Wc[(row - 64) * 64 + k] * (double)w.Wf[k * HAV_HID + col];
            v = (float)s;
        } else if (row == 67) v = w.Wa[col];
    } else if (e < OFF_B2) v = w.b1[e - OFF_B1];
    else if (e < OFF_BH) v = w.b2[e - OFF_B2];
    else {
        const int row = e - OFF_BH;
        if (row < 64) v = w.bf[row];
        else if (row < 67) {
            double s = (double)w.bc[row - 64];
            for (int k = 0; k < 64; ++k) s += (double)w.Wc[(row - 64) * 64 + k] * (double)w.bf[k];
            v = (float)s;
        } else if (row == 67) v = w.ba[0];
    }
    blob[e] = v;
}

extern "C" int hav_mlp_pack(void* blob, const HavMlpWeights* w, void* stream)
{
    if (!blob || !w || !w->W1 || !w->b1 || !w->W2 || !w->b2 || !w->Wa || !w->ba || !w->Wf || !w->bf || !w->Wc || !w->bc)
        return HAV_EINVAL;
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((BLOB_FLOATS + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float*)blob, *w);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// NCHW -> channels-last tri-plane: dst[(n*H*W + p)*C + c] = src[(n*C + c)*H*W + p]; LDS transpose tile
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(float* __restrict__ dst, const float* __restrict__ src, int C,
                                                           int HW)
{
    __shared__ float tile[64][65];
    const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int p = p0 + i, c = c0 + tx;
        if (c < C && p < HW) dst[((size_t)n * HW + p) * C + c] = tile[tx][i];
    }
}

extern "C" int hav_triplane_to_channels_last(float* dst, const float* src, int B, int C, int H, int W, void* stream)
{
    if (!dst || !src || B < 1 || C < 1 || H < 1 || W < 1) return HAV_EINVAL;
    const int HW = H * W;
    dim3 grid((HW + 63) / 64, (C + 63) / 64, 2 * B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, dst, src, C, HW);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#define DPP_QUAD_XOR1 0xB1
#define DPP_QUAD_XOR2 0x4E
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_HALF_MIRROR 0x141

template <int CTRL> __device__ __forceinline__ float dpp_mov(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<DPP_QUAD_XOR1>(0.f, v);
    v += dpp_mov<DPP_QUAD_XOR2>(0.f, v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(0.f, v);
    v += dpp_mov<DPP_ROW_MIRROR>(0.f, v);
    return v;
}
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR1>(0.f, v));
    v = fmaxf(v, dpp_mov<DPP_QUAD_XOR2>(0.f, v));
    v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(0.f, v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(0.f, v));
    return v;
}
// inclusive product scan along the 16 lanes of a DPP row (Kogge-Stone, identity 1)
__device__ __forceinline__ float row16_scan_mul(float v)
{
    v *= dpp_mov<DPP_ROW_SHR(1)>(1.f, v);
    v *= dpp_mov<DPP_ROW_SHR(2)>(1.f, v);
    v *= dpp_mov<DPP_ROW_SHR(4)>(1.f, v);
    v *= dpp_mov<DPP_ROW_SHR(8)>(1.f, v);
    return v;
}
__device__ __forceinline__ float read_lane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// orders this wave's LDS traffic (the per-wave scratch is private to a wave: no workgroup barrier needed)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Philox4x32-10 (Salmon et al. 2011), counter-based: one call per (ray, sample, stream)
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4])
{
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// sin for the positional encoding: branch-free Cody-Waite reduction by pi/2 (3 FMA terms) + degree-9/8 minimax
// polynomials, max abs error 9.3e-8 (<= 1.6 ulp) for |x| <= 260, i.e. |p| <= 2 at the top octave 2^7 (the NeRF box is
// +-1.6); accuracy degrades gracefully (not catastrophically) up to |x| ~ 1e5.  The libm-style sinf carries a
// Payne-Hanek slow path whose control flow costs registers in a kernel that has none to spare.
__device__ __forceinline__ float pe_sin(float x)
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
    const float v = (k & 1) ? cs : sn;
    return (k & 2) ? -v : v;
}

struct MarchArgs {
    HavRenderParams p;
    const float* rays; const float* bg; const float* inv_T; const float* planes; const float* vol; const float* blob;
    const float* t_rand; const float* u_rand; const float* noise_c; const float* noise_f;
    HavRenderOut out;
    float* dbg_zfine;       // optional [B*R, S_fp] dump of the merged fine depths (tests)
    long long NR;           // B*R
    int S_fp;               // ceil(S_c/2) + S_f, 0 if no fine pass
    int scr_floats;         // per-wave LDS scratch
    int o_w, o_cdf, o_cand, o_zf, o_racc, s_pad_c, s_pad_f;
};

enum { STREAM_XI = 0, STREAM_ZETA = 1, STREAM_EPS_C = 2, STREAM_EPS_F = 3 };

__device__ __forceinline__ float rng_uniform(const MarchArgs& a, long long gr, int s, int stream)
{
    uint32_t o[4];
    philox4x32((uint32_t)gr, (uint32_t)((unsigned long long)gr >> 32), (uint32_t)s,
               (uint32_t)stream + 16u * (uint32_t)a.p.rng_offset, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32), o);
    return (float)(o[0] >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float rng_normal(const MarchArgs& a, long long gr, int s, int stream)
{
    uint32_t o[4];
    philox4x32((uint32_t)gr, (uint32_t)((unsigned long long)gr >> 32), (uint32_t)s,
               (uint32_t)stream + 16u * (uint32_t)a.p.rng_offset, (uint32_t)a.p.seed, (uint32_t)(a.p.seed >> 32), o);
    const float u1 = ((float)(o[0] >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(o[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// coarse depth of sample i (model/nerf_trainer.py:129-139): torch.linspace + stratified jitter
__device__ __forceinline__ float lin_t(int k, int S, float step) { return (k < S / 2) ? step * (float)k : 1.0f - step * (float)(S - 1 - k); }
template <bool RANDOM>
__device__ __forceinline__ float z_coarse(const MarchArgs& a, long long gr, int i, float near, float far)
{
    const int S = a.p.S_c;
    const float step = 1.0f / (float)(S - 1);
    const float t = lin_t(i, S, step);
    float z = near * (1.0f - t) + far * t;
    if (RANDOM && a.p.perturb) {
        const int il = i > 0 ? i - 1 : 0, ih = i < S - 1 ? i + 1 : S - 1;
        const float tl = lin_t(il, S, step), th = lin_t(ih, S, step);
        const float zl = near * (1.0f - tl) + far * tl, zh = near * (1.0f - th) + far * th;
        const float lo = i > 0 ? 0.5f * (z + zl) : z;
        const float up = i < S - 1 ? 0.5f * (zh + z) : z;
        const float xi = a.t_rand ? a.t_rand[gr * S + i] : rng_uniform(a, gr, i, STREAM_XI);
        z = lo + (up - lo) * xi;
    }
    return z;
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
#define MARCH_THREADS 512
#define MARCH_WAVES (MARCH_THREADS / 64)
#define RACC_N 72   // 67 colour/feature sums | 67 depth | 68 acc | 69 wmax

// RANDOM=false: perturb == 0 and noise_std == 0 (deterministic rendering, the parity configuration); the random-number
// paths (injected tensors or Philox) are compiled out.
template <bool RANDOM>
__global__ void __launch_bounds__(MARCH_THREADS, 2) hav_march_f32_kernel(const MarchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sW1 = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* scr = smem + K1_STEPS * 4 * 64 + wave * a.scr_floats;
    float* s_w = scr + a.o_w;       // [2][s_pad_c]  coarse weights
    float* s_cdf = scr + a.o_cdf;   // [2][s_pad_c]
    float* s_cand = scr + a.o_cand; // [2][s_pad_f]  unsorted fine depths
    float* s_zf = scr + a.o_zf;     // [2][s_pad_f]  sorted fine depths
    float* s_racc = scr + a.o_racc; // [2][RACC_N]

    {   // stage the layer-1 fragments (already in fragment order) into LDS once per workgroup
        const float4* src = reinterpret_cast<const float4*>(a.blob + OFF_W1);
        float4* dst = reinterpret_cast<float4*>(sW1);
        for (int i = tid; i < K1_STEPS * 4 * 64 / 4; i += MARCH_THREADS) dst[i] = src[i];
    }
    __syncthreads();

    const int j = lane & 31, h = lane >> 5, col = lane & 15, rowt = (lane >> 4) & 1;
    // streamed weights/biases go through ONE buffer descriptor: per-load address = SGPR constant + lane*4, so the
    // compiler has no 64-bit per-load pointers to hoist out of the loops (it spilled ~900 of them otherwise)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.blob), 0, BLOB_FLOATS * 4, 0x00020000);
    const int voff = lane * 4, hoff = h * 16;
#define LDW(off_floats) __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, voff, (off_floats) * 4, 0))
#define LDB4(off_floats) __builtin_amdgcn_raw_buffer_load_b128(wrs, hoff, (off_floats) * 4, 0)

    const int S_c = a.p.S_c, S_fp = a.S_fp;
    const int PR = a.p.plane_res, VR = a.p.vol_res;
    const long long NR = a.NR;
    const long long npairs = (NR + 1) >> 1;

    // XCD-aware distribution: workgroup b runs on XCD b%8 (observed, used for L2 locality only); each XCD
    // owns one contiguous eighth of the ray pairs, its workgroups' waves sweep it in lockstep.
    long long chunk, base, span;
    int lb, nbx;
    if ((gridDim.x & 7) == 0) {
        chunk = (npairs + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        lb = blockIdx.x >> 3; nbx = gridDim.x >> 3;
        base = chunk * xcd;
        span = npairs - base; if (span > chunk) span = chunk; if (span < 0) span = 0;
    } else { base = 0; span = npairs; lb = blockIdx.x; nbx = gridDim.x; }

    for (long long local = (long long)lb * MARCH_WAVES + wave; local < span; local += (long long)nbx * MARCH_WAVES) {
        const long long pair = base + local;
        const long long ray0 = pair * 2;
        const bool has1 = (ray0 + 1) < NR;
        if (!has1) {     // odd ray count: slot 1 idles on finite dummy data
            for (int i = lane; i < a.s_pad_c; i += 64) s_w[a.s_pad_c + i] = 0.f;
        }

        for (int pass = 0; pass < (S_fp > 0 ? 2 : 1); ++pass) {
            const int S = pass == 0 ? S_c : S_fp;
            const int nr = (S + 15) >> 4;         // DPP rows per ray
            const int ntiles = nr;                // 2 rays * nr rows / 2 rows per tile
            for (int i = lane; i < 2 * RACC_N; i += 64) s_racc[i] = 0.f;
            float carry = 1.0f;
            wave_lds_sync();

            for (int tile = 0; tile < ntiles; ++tile) {
                // ---- which sample does this lane evaluate -------------------------------------------
                const int rowg = 2 * tile + rowt;
                const int slot = rowg >= nr ? 1 : 0;
                const int srow = rowg - slot * nr;
                int s = srow * 16 + col;
                const bool rayok = slot == 0 || has1;
                const bool valid = (s < S) && rayok;
                if (s > S - 1) s = S - 1;
                const long long gr = rayok ? ray0 + slot : ray0;
                const int b = (int)(gr / a.p.R);
                const float* ray = a.rays + gr * a.p.ray_stride;
                const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
                const float near = ray[6], far = ray[7];

                float z, znb;      // depth of this sample and of the neighbour that defines `dists`
                const int snb = (s + 1 < S) ? s + 1 : S - 2;       // utils/nerf_util.py:36-37 (last dist repeated)
                if (pass == 0) {
                    z = z_coarse<RANDOM>(a, gr, s, near, far);
                    znb = z_coarse<RANDOM>(a, gr, snb, near, far);
                } else {
                    z = s_zf[slot * a.s_pad_f + s];
                    znb = s_zf[slot * a.s_pad_f + snb];
                }
                const float dist = (s + 1 < S) ? (znb - z) : (z - znb);

                // ---- pts = o + d z; skinning field (model/Skinning_Field.py:77-95) -----------------------
                const float px = ox + dx * z, py = oy + dy * z, pz = oz + dz * z;
                const float* iT = a.inv_T + (size_t)b * 12;
                const float tx_ = px + iT[9], ty_ = py + iT[10], tz_ = pz + iT[11];
                const float p1x = tx_ * iT[0] + ty_ * iT[3] + tz_ * iT[6];
                const float p1y = tx_ * iT[1] + ty_ * iT[4] + tz_ * iT[7];
                const float p1z = tx_ * iT[2] + ty_ * iT[5] + tz_ * iT[8];
                float wsk[2];
#pragma unroll
                for (int bone = 0; bone < 2; ++bone) {
                    const float gx = (bone ? p1x : px) * a.p.skin_scale[0] + a.p.skin_trans[0];
                    const float gy = (bone ? p1y : py) * a.p.skin_scale[1] + a.p.skin_trans[1];
                    const float gz = (bone ? p1z : pz) * a.p.skin_scale[2] + a.p.skin_trans[2];
                    // grid_sample 3-D, border padding, align_corners=True (utils/util.py:409-418)
                    const float lim = (float)(VR - 1);
                    float ix = ((gx + 1.0f) * 0.5f) * lim, iy = ((gy + 1.0f) * 0.5f) * lim, iz = ((gz + 1.0f) * 0.5f) * lim;
                    ix = fminf(fmaxf(ix, 0.f), lim); iy = fminf(fmaxf(iy, 0.f), lim); iz = fminf(fmaxf(iz, 0.f), lim);
                    const float x0f = floorf(ix), y0f = floorf(iy), z0f = floorf(iz);
                    const float fx = ix - x0f, fy = iy - y0f, fz = iz - z0f;
                    const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
                    const int x1 = min(x0 + 1, VR - 1), y1 = min(y0 + 1, VR - 1), z1 = min(z0 + 1, VR - 1);  // weight is 0 when clamped
                    const float* v = a.vol + (size_t)bone * VR * VR * VR;
                    float acc = 0.f;
#pragma unroll
                    for (int cz = 0; cz < 2; ++cz)
#pragma unroll
                        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                            for (int cx = 0; cx < 2; ++cx) {
                                const float wgt = (cx ? fx : 1.0f - fx) * (cy ? fy : 1.0f - fy) * (cz ? fz : 1.0f - fz);
                                acc += v[((size_t)(cz ? z1 : z0) * VR + (cy ? y1 : y0)) * VR + (cx ? x1 : x0)] * wgt;
                            }
                    wsk[bone] = acc;
                }
                const float den = (wsk[0] + wsk[1]) + 1e-8f;
                const float n0 = wsk[0] / den, n1 = wsk[1] / den;
                const float qx_ = n0 * px + n1 * p1x, qy_ = n0 * py + n1 * p1y, qz_ = n0 * pz + n1 * p1z;   // p'

                __builtin_amdgcn_sched_barrier(0);
                // ---- MLP inputs of this half-wave: plane h (64 ch) + PE octaves 4h..4h+3 (24) ----------
                float x[K1_STEPS];
                {
                    // Embedder.embed (model/network/embedder.py:32-61): sin(p f), sin(p f + pi/2), f = 2^k
                    const float halfpi = 1.57079632679489661923f;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const float f = h ? (float)(16 << kk) : (float)(1 << kk);
                        const float ax = qx_ * f, ay = qy_ * f, az = qz_ * f;
                        x[HAV_PC + 6 * kk + 0] = pe_sin(ax);
                        x[HAV_PC + 6 * kk + 1] = pe_sin(ay);
                        x[HAV_PC + 6 * kk + 2] = pe_sin(az);
                        x[HAV_PC + 6 * kk + 3] = pe_sin(ax + halfpi);
                        x[HAV_PC + 6 * kk + 4] = pe_sin(ay + halfpi);
                        x[HAV_PC + 6 * kk + 5] = pe_sin(az + halfpi);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // sample_from_triplane_new (utils/util.py:359-392): plane0 at (x,y), plane1 at (z,y); zeros padding
                    const float gx = (h ? qz_ * a.p.nerf_scale[2] + a.p.nerf_trans[2] : qx_ * a.p.nerf_scale[0] + a.p.nerf_trans[0]);
                    const float gy = qy_ * a.p.nerf_scale[1] + a.p.nerf_trans[1];
                    const float lim = (float)(PR - 1);
                    const float ix = ((gx + 1.0f) * 0.5f) * lim, iy = ((gy + 1.0f) * 0.5f) * lim;
                    float x0f = floorf(ix), y0f = floorf(iy);
                    const float wx1 = ix - x0f, wx0 = 1.0f - wx1, wy1 = iy - y0f, wy0 = 1.0f - wy1;
                    x0f = fminf(fmaxf(x0f, -2.f), lim + 2.f); y0f = fminf(fmaxf(y0f, -2.f), lim + 2.f);
                    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
                    const bool vx0 = x0 >= 0 && x0 < PR, vx1 = x1 >= 0 && x1 < PR, vy0 = y0 >= 0 && y0 < PR, vy1 = y1 >= 0 && y1 < PR;
                    const float w00 = (vx0 && vy0) ? wx0 * wy0 : 0.f, w01 = (vx1 && vy0) ? wx1 * wy0 : 0.f;
                    const float w10 = (vx0 && vy1) ? wx0 * wy1 : 0.f, w11 = (vx1 && vy1) ? wx1 * wy1 : 0.f;
                    const int cx0 = min(max(x0, 0), PR - 1), cx1 = min(max(x1, 0), PR - 1);
                    const int cy0 = min(max(y0, 0), PR - 1), cy1 = min(max(y1, 0), PR - 1);
                    const float* pl = a.planes + ((size_t)h * a.p.B + b) * PR * PR * HAV_PC;
                    const float4* t00 = reinterpret_cast<const float4*>(pl + ((size_t)cy0 * PR + cx0) * HAV_PC);
                    const float4* t01 = reinterpret_cast<const float4*>(pl + ((size_t)cy0 * PR + cx1) * HAV_PC);
                    const float4* t10 = reinterpret_cast<const float4*>(pl + ((size_t)cy1 * PR + cx0) * HAV_PC);
                    const float4* t11 = reinterpret_cast<const float4*>(pl + ((size_t)cy1 * PR + cx1) * HAV_PC);
                    // 4 x 256 B per lane; issued 16 loads (4 channel quads x 4 taps) at a time to bound live registers
#pragma unroll
                    for (int cg = 0; cg < HAV_PC / 16; ++cg) {
                        float4 v00[4], v01[4], v10[4], v11[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { v00[u] = t00[4 * cg + u]; v01[u] = t01[4 * cg + u]; v10[u] = t10[4 * cg + u]; v11[u] = t11[4 * cg + u]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c4 = 4 * cg + u;
                            x[4 * c4 + 0] = v00[u].x * w00 + v01[u].x * w01 + v10[u].x * w10 + v11[u].x * w11;
                            x[4 * c4 + 1] = v00[u].y * w00 + v01[u].y * w01 + v10[u].y * w10 + v11[u].y * w11;
                            x[4 * c4 + 2] = v00[u].z * w00 + v01[u].z * w01 + v10[u].z * w10 + v11[u].z * w11;
                            x[4 * c4 + 3] = v00[u].w * w00 + v01[u].w * w01 + v10[u].w * w10 + v11[u].w * w11;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }

                // ---- layer 1: 176 -> 128, relu (model/nerf_model.py:104-108) ---------------------------
                f32x16 acc1[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const auto bb = LDB4(OFF_B1 + 32 * m + 8 * q);
                        acc1[m][4 * q + 0] = __uint_as_float(bb[0]); acc1[m][4 * q + 1] = __uint_as_float(bb[1]);
                        acc1[m][4 * q + 2] = __uint_as_float(bb[2]); acc1[m][4 * q + 3] = __uint_as_float(bb[3]);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < K1_STEPS; ++t) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        acc1[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(sW1[(t * 4 + m) * 64 + lane], x[t], acc1[m], 0, 0, 0);
                    if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[m][r] = fmaxf(acc1[m][r], 0.f);

                __builtin_amdgcn_sched_barrier(0);
                // ---- layer 2: 128 -> 128, relu; B operands are layer-1 accumulator registers ----------
                f32x16 acc2[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const auto bb = LDB4(OFF_B2 + 32 * m + 8 * q);
                        acc2[m][4 * q + 0] = __uint_as_float(bb[0]); acc2[m][4 * q + 1] = __uint_as_float(bb[1]);
                        acc2[m][4 * q + 2] = __uint_as_float(bb[2]); acc2[m][4 * q + 3] = __uint_as_float(bb[3]);
                    }
                {   // weights stream from L2: 4 k-steps (16 fragments) are in flight while the previous 16 MFMAs run
                    float af[2][16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) af[0][u] = LDW(OFF_W2 + u * 64);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        if (g + 1 < 16) {
#pragma unroll
                            for (int u = 0; u < 16; ++u) af[(g + 1) & 1][u] = LDW(OFF_W2 + ((g + 1) * 16 + u) * 64);
                        }
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) {
                            const int ks = g * 4 + kq;
#pragma unroll
                            for (int m = 0; m < 4; ++m)
                                acc2[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][kq * 4 + m], acc1[ks >> 4][ks & 15], acc2[m], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[m][r] = fmaxf(acc2[m][r], 0.f);

                __builtin_amdgcn_sched_barrier(0);
                // ---- head: 128 -> 64 feat | 3 rgb (folded fc_rgb o fc_rgbFeat) | alpha ---------------------
                f32x16 acc3[HEAD_TILES];
#pragma unroll
                for (int m = 0; m < HEAD_TILES; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const auto bb = LDB4(OFF_BH + 32 * m + 8 * q);
                        acc3[m][4 * q + 0] = __uint_as_float(bb[0]); acc3[m][4 * q + 1] = __uint_as_float(bb[1]);
                        acc3[m][4 * q + 2] = __uint_as_float(bb[2]); acc3[m][4 * q + 3] = __uint_as_float(bb[3]);
                    }
                {
                    float af[2][4 * HEAD_TILES];
#pragma unroll
                    for (int u = 0; u < 4 * HEAD_TILES; ++u) af[0][u] = LDW(OFF_WH + u * 64);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        if (g + 1 < 16) {
#pragma unroll
                            for (int u = 0; u < 4 * HEAD_TILES; ++u) af[(g + 1) & 1][u] = LDW(OFF_WH + ((g + 1) * 4 * HEAD_TILES + u) * 64);
                        }
#pragma unroll
                        for (int kq = 0; kq < 4; ++kq) {
                            const int ks = g * 4 + kq;
#pragma unroll
                            for (int m = 0; m < HEAD_TILES; ++m)
                                acc3[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][kq * HEAD_TILES + m], acc2[ks >> 4][ks & 15], acc3[m], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }

                __builtin_amdgcn_sched_barrier(0);
                // ---- volume_render_radiance_field (utils/nerf_util.py:28-73) -------------------------------
                // rows 64..67 of the head (rgb, alpha) live in half-wave 0, registers 0..3 of tile 2
                const float sig_raw = __shfl(acc3[2][3], j, 64);
                float sg = sig_raw;
                if (RANDOM && a.p.noise_std > 0.f) {
                    const float* nz = pass == 0 ? a.noise_c : a.noise_f;
                    const float e = nz ? nz[gr * S + s] : rng_normal(a, gr, s, pass == 0 ? STREAM_EPS_C : STREAM_EPS_F);
                    sg += e * a.p.noise_std;
                }
                sg = fmaxf(sg, 0.f);
                const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
                float alpha = 1.0f - expf(-sg * (dist * dn));
                if (!valid) alpha = 0.f;
                const float om = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
                const float pin = row16_scan_mul(om);                        // inclusive product within the row
                const float pex = dpp_mov<DPP_ROW_SHR(1)>(1.0f, pin);        // exclusive
                const float tot0 = read_lane(pin, 15), tot1 = read_lane(pin, 31);
                const bool first0 = (2 * tile) % nr == 0, first1 = (2 * tile + 1) % nr == 0;
                const float cin0 = first0 ? 1.0f : carry;
                const float cin1 = first1 ? 1.0f : cin0 * tot0;
                carry = cin1 * tot1;
                const float wgt = alpha * ((rowt ? cin1 : cin0) * pex);         // :60
                if (pass == 0 && h == 0 && valid) s_w[slot * a.s_pad_c + s] = wgt;

                // weighted sums: reduce over the 16 lanes of each row, then add into the per-ray LDS accumulators
                const bool same = (2 * tile) / nr == (2 * tile + 1) / nr;    // both rows of this tile belong to one ray
                float* racc = s_racc + slot * RACC_N;
                const bool writer = (col == 0) && (!same || rowt == 0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // rows are always added in increasing row order, one at a time, so a ray's sums do not depend
                        // on which slot of the pair (or which call) it was rendered in: results are bit-reproducible
                        const float v = row16_sum(acc3[m][r] * wgt);
                        const float vo = same ? __shfl_xor(v, 16, 64) : 0.f;
                        if (writer) racc[3 + acc_row(m, r, h)] = (racc[3 + acc_row(m, r, h)] + v) + vo;
                    }
                {
                    float e0, e1, e2, e3, e4;
                    if (h == 0) {
                        e0 = 1.0f / (1.0f + expf(-acc3[2][0]));              // sigmoid on rgb only (:45-46)
                        e1 = 1.0f / (1.0f + expf(-acc3[2][1]));
                        e2 = 1.0f / (1.0f + expf(-acc3[2][2]));
                        e3 = z; e4 = 1.0f;
                    } else { e0 = e1 = e2 = e3 = e4 = 0.f; }
                    const float v0 = row16_sum(e0 * wgt), v1 = row16_sum(e1 * wgt), v2 = row16_sum(e2 * wgt);
                    const float v3 = row16_sum(e3 * wgt), v4 = row16_sum(e4 * wgt);
                    float v5 = row16_max(wgt);
                    float u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f, u4 = 0.f;
                    if (same) {
                        u0 = __shfl_xor(v0, 16, 64); u1 = __shfl_xor(v1, 16, 64); u2 = __shfl_xor(v2, 16, 64);
                        u3 = __shfl_xor(v3, 16, 64); u4 = __shfl_xor(v4, 16, 64); v5 = fmaxf(v5, __shfl_xor(v5, 16, 64));
                    }
                    if (writer && h == 0) {
                        racc[0] = (racc[0] + v0) + u0; racc[1] = (racc[1] + v1) + u1; racc[2] = (racc[2] + v2) + u2;
                        racc[67] = (racc[67] + v3) + u3; racc[68] = (racc[68] + v4) + u4;
                        racc[69] = fmaxf(racc[69], v5);
                    }
                }
                wave_lds_sync();
            }   // tiles

            // ---- write this pass's outputs ----------------------------------------------------------
            for (int slot = 0; slot < (has1 ? 2 : 1); ++slot) {
                const long long gr = ray0 + slot;
                const float* racc = s_racc + slot * RACC_N;
                const float accv = racc[68];
                float* rgb = (pass == 0 ? a.out.rgb_coarse : a.out.rgb_fine) + gr * 67;
                for (int i = lane; i < 67; i += 64) {
                    float v = racc[i];
                    if (i < 3 && a.bg) v = v + (1.0f - accv) * a.bg[gr * 3 + i];     // :70-71
                    rgb[i] = v;
                }
                if (lane == 0) {
                    (pass == 0 ? a.out.depth_coarse : a.out.depth_fine)[gr] = racc[67];
                    (pass == 0 ? a.out.acc_coarse : a.out.acc_fine)[gr] = accv;
                    if (pass == 1 || S_fp == 0) a.out.weights_max[gr] = racc[69];    // model/nerf_trainer.py:195,200
                }
            }

            // ---- inverse-CDF resampling + merge (utils/nerf_util.py:76-117, model/nerf_trainer.py:166-170) ---
            if (pass == 0 && S_fp > 0) {
                const int slot = h, li = j;                       // half-wave h prepares ray slot h
                const long long gr = (slot == 0 || has1) ? ray0 + slot : ray0;
                const float* ray = a.rays + gr * a.p.ray_stride;
                const float near = ray[6], far = ray[7];
                const int nb = S_c - 1, nw = S_c - 2, S_half = (S_c + 1) >> 1, S_f = a.p.S_f;
                float* w = s_w + slot * a.s_pad_c;
                float* cdf = s_cdf + slot * a.s_pad_c;
                float* cand = s_cand + slot * a.s_pad_f;
                float* zf = s_zf + slot * a.s_pad_f;
                if (li == 0) {      // sequential sum / cumsum in the reference's order (SURVEY B-11)
                    float sum = 0.f;
                    for (int i = 0; i < nw; ++i) sum += (w[1 + i] + 1e-5f);
                    float run = 0.f;
                    cdf[0] = 0.f;
                    for (int i = 0; i < nw; ++i) { run += (w[1 + i] + 1e-5f) / sum; cdf[i + 1] = run; }
                }
                wave_lds_sync();
                for (int k = li; k < S_f; k += 32) {
                    float u;
                    if (!RANDOM || !a.p.perturb) {       // det: torch.linspace(0,1,S_f)
                        const float st = 1.0f / (float)(S_f - 1);
                        u = (S_f == 1) ? 0.f : ((k < S_f / 2) ? st * (float)k : 1.0f - st * (float)(S_f - 1 - k));
                    } else {
                        const float zeta = a.u_rand ? a.u_rand[gr * S_f + k] : rng_uniform(a, gr, k, STREAM_ZETA);
                        const float sN = (float)(1.0 / (double)S_f);
                        u = (float)k * sN + zeta * (float)(1.0 / (double)S_f - 1e-6);
                    }
                    int inds = 0;                                  // searchsorted(cdf, u, right=True)
                    for (int i = 0; i < nb; ++i) inds += (cdf[i] <= u) ? 1 : 0;
                    const int below = max(inds - 1, 0), above = min(inds, nb - 1);
                    float dnm = cdf[above] - cdf[below];
                    if (dnm < 1e-5f) dnm = 1.0f;
                    const float tt = (u - cdf[below]) / dnm;
                    const float bl = 0.5f * (z_coarse<RANDOM>(a, gr, below + 1, near, far) + z_coarse<RANDOM>(a, gr, below, near, far));
                    const float ba = 0.5f * (z_coarse<RANDOM>(a, gr, above + 1, near, far) + z_coarse<RANDOM>(a, gr, above, near, far));
                    cand[S_half + k] = bl + tt * (ba - bl);
                }
                for (int i = li; i < S_half; i += 32) cand[i] = z_coarse<RANDOM>(a, gr, 2 * i, near, far);   // z_vals[:, ::2]
                wave_lds_sync();
                for (int e = li; e < S_fp; e += 32) {              // rank sort == torch.sort on 48 values
                    const float v = cand[e];
                    int rank = 0;
                    for (int q = 0; q < S_fp; ++q) {
                        const float o = cand[q];
                        rank += (o < v || (o == v && q < e)) ? 1 : 0;
                    }
                    zf[rank] = v;
                    if (a.dbg_zfine && (slot == 0 || has1)) a.dbg_zfine[gr * S_fp + rank] = v;
                }
                wave_lds_sync();
            }
        }   // pass
    }       // pairs
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static float* g_dbg_zfine = nullptr;
// test hook: next hav_render_rays call also dumps the merged fine depths [B*R,S_fp] to this device buffer
extern "C" void hav_debug_set_zfine(float* dev_ptr) { g_dbg_zfine = dev_ptr; }

extern "C" const char* hav_render_variant(const HavRenderParams* p)
{
    return (p && (p->perturb != 0 || p->noise_std > 0.f)) ? "hav_march_f32_kernel<true>" : "hav_march_f32_kernel<false>";
}

extern "C" int hav_render_rays(const HavRenderParams* p, const float* rays, const float* bg, const float* inv_T,
                               const float* planes_cl, const float* skin_vol, const void* mlp_blob, const float* t_rand,
                               const float* u_rand, const float* noise_c, const float* noise_f, const HavRenderOut* out,
                               void* stream)
{
    if (!p || !rays || !inv_T || !planes_cl || !skin_vol || !mlp_blob || !out) return HAV_EINVAL;
    if (p->B < 1 || p->R < 0 || p->ray_stride < 8 || p->S_c < 2 || p->S_f < 0) return HAV_EINVAL;
    if (p->plane_ch != HAV_PC) return HAV_EUNSUP;                 // Trainer hard-codes triPlane_feat_dim=64 (nerf_trainer.py:22)
    if (p->plane_res < 2 || p->vol_res < 2) return HAV_EINVAL;
    if (p->S_c > 256 || p->S_f > 128) return HAV_EUNSUP;
    if (!out->rgb_coarse || !out->depth_coarse || !out->acc_coarse || !out->weights_max) return HAV_EINVAL;
    if (p->S_f > 0 && (!out->rgb_fine || !out->depth_fine || !out->acc_fine)) return HAV_EINVAL;
    if (p->R == 0) return 0;

    MarchArgs a;
    a.p = *p;
    a.rays = rays; a.bg = bg; a.inv_T = inv_T; a.planes = planes_cl; a.vol = skin_vol; a.blob = (const float*)mlp_blob;
    a.t_rand = t_rand; a.u_rand = u_rand; a.noise_c = noise_c; a.noise_f = noise_f;
    a.out = *out;
    a.dbg_zfine = g_dbg_zfine; g_dbg_zfine = nullptr;
    a.NR = (long long)p->B * p->R;
    a.S_fp = p->S_f > 0 ? (p->S_c + 1) / 2 + p->S_f : 0;
    a.s_pad_c = (p->S_c + 15) & ~15;
    a.s_pad_f = ((a.S_fp > 0 ? a.S_fp : 1) + 15) & ~15;
    a.o_w = 0;
    a.o_cdf = a.o_w + 2 * a.s_pad_c;
    a.o_cand = a.o_cdf + 2 * a.s_pad_c;
    a.o_zf = a.o_cand + 2 * a.s_pad_f;
    a.o_racc = a.o_zf + 2 * a.s_pad_f;
    a.scr_floats = (a.o_racc + 2 * RACC_N + 3) & ~3;
    const size_t lds = ((size_t)K1_STEPS * 4 * 64 + (size_t)MARCH_WAVES * a.scr_floats) * sizeof(float);
    if (lds > 160 * 1024) return HAV_EUNSUP;

    const bool random = p->perturb != 0 || p->noise_std > 0.f;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)hav_march_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)hav_march_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long long npairs = (a.NR + 1) / 2;
    int grid = hav_num_cus();
    const long long need = (npairs + MARCH_WAVES - 1) / MARCH_WAVES;
    if (need < grid) grid = (int)((need + 7) / 8 * 8);
    if (random) hipLaunchKernelGGL(hav_march_f32_kernel<true>, dim3(grid), dim3(MARCH_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(hav_march_f32_kernel<false>, dim3(grid), dim3(MARCH_THREADS), lds, (hipStream_t)stream, a);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// get_rays on device (dataloader/data_util.py:28-56 + dataloader/dataloader.py:174-177)
// ------------------------------------------------------------------------------------------------
struct GenRaysArgs { float intr[4]; float c2w[12]; float near, far; int H, W, y0, y1; };

__global__ void __launch_bounds__(256) gen_rays_kernel(float* __restrict__ rays, GenRaysArgs g)
{
    const int n = (g.y1 - g.y0) * g.W;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int i = idx % g.W, jj = g.y0 + idx / g.W;
    const float fx = g.intr[0], fy = g.intr[1], cx = g.intr[2] * (float)g.W, cy = g.intr[3] * (float)g.H;
    const float dc0 = ((float)i - cx) / fx, dc1 = ((float)jj - cy) / fy, dc2 = 1.0f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = g.c2w[r * 4 + 0] * dc0 + g.c2w[r * 4 + 1] * dc1 + g.c2w[r * 4 + 2] * dc2;
    const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float4* o = reinterpret_cast<float4*>(rays + (size_t)idx * 8);
    o[0] = make_float4(g.c2w[3], g.c2w[7], g.c2w[11], d[0] / nrm);
    o[1] = make_float4(d[1] / nrm, d[2] / nrm, g.near, g.far);
}

extern "C" int hav_gen_rays(float* rays, int H, int W, const float intr[4], const float c2w[12], float near, float far,
                            int y0, int y1, void* stream)
{
    if (!rays || !intr || !c2w || H < 1 || W < 1 || y0 < 0 || y1 > H || y1 <= y0) return HAV_EINVAL;
    GenRaysArgs g;
    for (int i = 0; i < 4; ++i) g.intr[i] = intr[i];
    for (int i = 0; i < 12; ++i) g.c2w[i] = c2w[i];
    g.near = near; g.far = far; g.H = H; g.W = W; g.y0 = y0; g.y1 = y1;
    const int n = (y1 - y0) * W;
    hipLaunchKernelGGL(gen_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays, g);
    HAV_LAUNCH_CHECK();
    return 0;
}

extern "C" int hav_abi_version(void) { return HAV_ABI_VERSION; }
