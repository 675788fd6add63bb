// hav_mlp_train.hip -- the radiance MLP of the TRAINING path (BASELINE config 5) on the bf16 matrix cores of gfx950:
// forward, and a backward that recomputes the activations instead of storing them.
//
// Replaces, under autograd, ConditionalTriplaneNeRFModel_multiRender_split_view.forward's five nn.Linear calls
// (reference model/nerf_model.py:104-117): x[176] -> relu(W1 x + b1)[128] -> relu(W2 . + b2)[128] -> { a = Wa h2 + ba,
// g = Wf h2 + bf (64, no activation), c = Wc g + bc (3) } -> rf = [c | g | a] (68), and their gradients.
//
// Three kernels, all built on one 32-query tile per wave64 (lane = j + 32 h; j = query of the tile, h = which half of every
// 16-wide k chunk the lane feeds -- the operand layout of v_mfma_f32_32x32x16_bf16):
//   mlp_fwd_kernel        X -> rf.  "T" orientation  Z^T[unit][query] = W . X^T : the D registers of one layer are the B operands
//                         of the next (lane = query), weights come as pre-permuted A fragments (hav_mlp_train_pack).
//   mlp_bwd_data_kernel   X, d_rf -> dX.  Recomputes h1, h2 (T), runs the chain  dG -> dH2 -> dZ2 -> dH1 -> dZ1 -> dX  (T), and
//                         ALSO evaluates every quantity a weight gradient contracts over the queries in the "N" orientation
//                         Z[query][unit] -- which on these matrix cores is the SAME instruction with the two operands swapped: the A
//                         and B register layouts of the 32x32x16 MFMA are mirror images, so mfma(A = x-frag, B = w-frag) yields the
//                         transposed tile with lane = unit and registers = queries.  Those registers, rounded to bf16, ARE ready-made
//                         A/B fragments of a product whose k dimension is the query index: they are streamed out as such (1 KB
//                         coalesced per store, no LDS, no shuffles).  Plain copies are transposed the same way, by an identity B operand.
//   mlp_bwd_weights_kernel  dW = sum over queries: each wave owns a strip of output tiles (one A row tile x up to 7 B column tiles)
//                         and a slice of the query tiles, accumulates in registers, and writes one partial per slice;
//                         mlp_reduce_kernel sums the slices in a fixed order (bit-reproducible: no atomics) into the nn.Linear
//                         layouts.  Bias gradients fall out of the same products: the N copy of X carries a row of ones.
// Arithmetic: bf16 operands (weights and activations, round-to-nearest-even), fp32 accumulation, fp32 inputs/outputs and master
// weights -- BASELINE.json's "bf16 MFMA MLP GEMM".  Algorithmic work: 94 848 FLOP/query forward, 3x that with both gradients.
#include "hav_common.h"
#include <atomic>

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

#define T_IN 176
#define T_HID 128
#define T_OUT 68
// fragment section of the packed blob, in uint4 (= one lane's 8 bf16) units: [k chunk][row tile][64 lanes]
#define TF_W1 0                          // [11][4]  A of layer 1 (T) / B of H1' (N):   W1[32m+i][16c+8h+e]
#define TF_W2 (TF_W1 + 11 * 4 * 64)      // [8][4]   W2[32m+i][U(ch,h,e)]
#define TF_WF (TF_W2 + 8 * 4 * 64)       // [8][2]   Wf[32m+i][U(ch,h,e)]
#define TF_WFT (TF_WF + 8 * 2 * 64)      // [5][4]   c<4: Wf[16c+8h+e][32m+i];  c==4: slot 0 = Wa[32m+i] (the alpha row rides along)
#define TF_W2T (TF_WFT + 5 * 4 * 64)     // [8][4]   W2[U(ch,h,e)][32m+i]
#define TF_W1T (TF_W2T + 8 * 4 * 64)     // [8][6]   W1[U(ch,h,e)][32m+i]  (0 for input rows >= 176)
#define TF_END (TF_W1T + 8 * 6 * 64)
// fp32 vector section (float offsets after the fragments)
#define TV_B1 0
#define TV_B2 128
#define TV_BF 256
#define TV_WA 320
#define TV_BA 448
#define TV_WC 452                        // [3][64]
#define TV_BC 644
#define TV_END 648
// N-orientation operand stream: 27 tiles of [2 k chunks][64 lanes] uint4 per 32-query tile
#define OP_XN 0     // 6: inputs 0..191 (row 176 = ones, rows 177.. = 0)
#define OP_H1N 6    // 4
#define OP_H2N 10   // 4
#define OP_GN 14    // 2
#define OP_DZ1N 16  // 4
#define OP_DZ2N 20  // 4
#define OP_DGN 24   // 2
#define OP_D4N 26   // 1: rows d_c0, d_c1, d_c2, d_a
#define OP_TILES 27
#define OUT_TILES 61
#define N_ROLES 11

extern "C" int64_t hav_mlp_train_blob_bytes(void) { return (int64_t)TF_END * 16 + TV_END * 4; }
extern "C" int64_t hav_mlp_train_ops_bytes(int64_t n) { return ((n + 31) / 32) * (int64_t)OP_TILES * 2 * 64 * 16; }
static int weight_slices(int64_t ntiles)
{
    int64_t s = (int64_t)hav_num_cus() * 8 / N_ROLES;        // ~8 single-wave workgroups per CU
    if (s > ntiles) s = ntiles;
    if (s < 1) s = 1;
    return (int)s;
}
extern "C" int64_t hav_mlp_train_partial_bytes(int64_t n) { return (int64_t)weight_slices((n + 31) / 32) * OUT_TILES * 64 * 16 * 4; }

// hidden unit that k slot (h, e) of chunk ch stands for when the B operand is a D-register tile (same map as the inference kernel)
__host__ __device__ inline int unit_of(int ch, int h, int e) { const int r = 8 * (ch & 1) + e; return 32 * (ch >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ uint32_t pk_bf16(float a, float b)
{
    const f2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2_t));      // v_cvt_pk_bf16_f32 (round to nearest even)
}
__device__ __forceinline__ bf16x8_t frag(uint4 u) { return __builtin_bit_cast(bf16x8_t, u); }
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(a), frag(b), c, 0, 0, 0)

// ------------------------------------------------------------------------------------------------------------------------
// pack: nn.Linear-layout fp32 weights -> bf16 fragments + fp32 vectors
// ------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mlp_train_pack_kernel(uint4* __restrict__ fragp, float* __restrict__ vec, HavMlpWeights w)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < TV_END) {
        float v = 0.f;
        if (idx < TV_B2) v = w.b1[idx];
        else if (idx < TV_BF) v = w.b2[idx - TV_B2];
        else if (idx < TV_WA) v = w.bf[idx - TV_BF];
        else if (idx < TV_BA) v = w.Wa[idx - TV_WA];
        else if (idx == TV_BA) v = w.ba[0];
        else if (idx >= TV_WC && idx < TV_BC) v = w.Wc[idx - TV_WC];
        else if (idx >= TV_BC && idx < TV_BC + 3) v = w.bc[idx - TV_BC];
        vec[idx] = v;
    }
    if (idx >= TF_END) return;
    const int l = idx & 63, i = l & 31, h = l >> 5;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float x;
        if (idx < TF_W2) { const int q = idx >> 6, m = q & 3, c = q >> 2; x = w.W1[(32 * m + i) * T_IN + 16 * c + 8 * h + e]; }
        else if (idx < TF_WF) { const int q = (idx - TF_W2) >> 6, m = q & 3, ch = q >> 2; x = w.W2[(32 * m + i) * T_HID + unit_of(ch, h, e)]; }
        else if (idx < TF_WFT) { const int q = (idx - TF_WF) >> 6, m = q & 1, ch = q >> 1; x = w.Wf[(32 * m + i) * T_HID + unit_of(ch, h, e)]; }
        else if (idx < TF_W2T) {
            const int q = (idx - TF_WFT) >> 6, m = q & 3, c = q >> 2;
            x = c < 4 ? w.Wf[(16 * c + 8 * h + e) * T_HID + 32 * m + i] : ((h == 0 && e == 0) ? w.Wa[32 * m + i] : 0.f);
        } else if (idx < TF_W1T) { const int q = (idx - TF_W2T) >> 6, m = q & 3, ch = q >> 2; x = w.W2[unit_of(ch, h, e) * T_HID + 32 * m + i]; }
        else { const int q = (idx - TF_W1T) >> 6, m = q % 6, ch = q / 6; x = (32 * m + i < T_IN) ? w.W1[unit_of(ch, h, e) * T_IN + 32 * m + i] : 0.f; }
        v[e] = x;
    }
    fragp[idx] = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
}

extern "C" int hav_mlp_train_pack(void* blob, const HavMlpWeights* w, void* stream)
{
    if (!blob || !w || !w->W1 || !w->b1 || !w->W2 || !w->b2 || !w->Wa || !w->ba || !w->Wf || !w->bf || !w->Wc || !w->bc) return HAV_EINVAL;
    hipLaunchKernelGGL(mlp_train_pack_kernel, dim3((TF_END + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint4*)blob,
                       (float*)((char*)blob + (size_t)TF_END * 16), *w);
    HAV_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// shared device pieces
// ------------------------------------------------------------------------------------------------------------------------
struct TrainCtx {
    const uint4* fr;     // fragments (+ lane already added)
    const float* vec;
    int lane, j, h;
};

// Rounds 1-2 ended every batch of MFMAs with 32 wait states, on the belief that v_mfma_f32_32x32x16_* keeps reading its A/B registers after
// issue; tools/ubench/mfma_war.hip showed that gfx950 does not (docs/history/DESIGN_r1-r4.md 3.5, retracted in round 3), hav_render.hip and
// hav_conv.hip dropped their pads then.  The statement stays as a scheduling fence that ties the accumulators (nothing that builds the next
// operands is moved in front of the batch); -DMLP_MFMA_PAD='"s_nop 15\n\ts_nop 15"' brings the wait states back.
#ifndef MLP_MFMA_PAD
#define MLP_MFMA_PAD ""
#endif
#define MFMA_FENCE(accv) asm volatile(MLP_MFMA_PAD : "+v"(accv))
#define KEEP_ALIVE(frag_) asm volatile("" : : "v"((frag_).x), "v"((frag_).w))

// acc[m] += sum over NCH k chunks of  (SWAP ? op(c) . ld(c, m) : ld(c, m) . op(c)):  ld = a weight fragment from memory, op = a register
// operand.  The fragments of chunk c + 1 are requested before the MFMAs of chunk c issue (explicit double buffer) and nothing may
// move across a chunk boundary: left alone, the compiler hoists every load of the unrolled loop to the top and spills hundreds of
// registers.  SWAP selects the orientation: false = T (D rows = the fragment's rows, lane = query), true = N (lane = the
// fragment's rows, registers = queries).
template <int NCH, int NM, bool SWAP, typename LD, typename OP>
__device__ __forceinline__ void mma_seq(f32x16 (&acc)[NM], LD ld, OP op)
{
    uint4 buf[2][NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) buf[0][m] = ld(0, m);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int m = 0; m < NM; ++m) buf[(c + 1) & 1][m] = ld(c + 1, m);
        }
        const uint4 o = op(c);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m] = SWAP ? MFMA_BF16(o, buf[c & 1][m], acc[m]) : MFMA_BF16(buf[c & 1][m], o, acc[m]);
        __builtin_amdgcn_sched_barrier(0);
    }
    MFMA_FENCE(acc[NM - 1]);
}
template <int NM> __device__ __forceinline__ void zero_tiles(f32x16 (&acc)[NM])
{
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
}
// N-orientation bias: every register of tile m holds b[32 m + lane's unit]
template <int NM> __device__ __forceinline__ void bias_tiles_n(f32x16 (&acc)[NM], const float* b, int j)
{
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const float v = b[32 * m + j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = v;
    }
}

// D-register tile set (4 row tiles of one layer, lane = query) -> relu -> bf16 B/A fragments, chunk ch = regs 8(ch&1).. of tile ch>>1
__device__ __forceinline__ void tiles_to_frags(const f32x16 (&acc)[4], uint4 (&f)[8], bool relu)
{
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = acc[ch >> 1][8 * (ch & 1) + e]; v[e] = relu ? fmaxf(x, 0.f) : x; }
        f[ch] = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
    }
}
// 0xFFFF per bf16 half that is > 0 (relu outputs are >= +0: nonzero bits <=> positive)
__device__ __forceinline__ uint32_t pos_mask(uint32_t p) { return ((p & 0xFFFFu) ? 0xFFFFu : 0u) | ((p & 0xFFFF0000u) ? 0xFFFF0000u : 0u); }
// dZ = dH * [h > 0] for a D-register tile set, straight to bf16 fragments
__device__ __forceinline__ void masked_frags(const f32x16 (&acc)[4], const uint4 (&hf)[8], uint4 (&f)[8])
{
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const uint32_t hm[4] = {hf[ch].x, hf[ch].y, hf[ch].z, hf[ch].w};
        uint32_t o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float a = acc[ch >> 1][8 * (ch & 1) + 2 * d], b = acc[ch >> 1][8 * (ch & 1) + 2 * d + 1];
            o[d] = pk_bf16(a, b) & pos_mask(hm[d]);
        }
        f[ch] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// bias of a T-orientation tile set: acc[m][r] = b[unit(m, r, h)]
__device__ __forceinline__ void bias_tiles(f32x16 (&acc)[4], const float* b, int h)
{
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(b + 32 * m + 8 * g + 4 * h);
            acc[m][4 * g + 0] = v.x; acc[m][4 * g + 1] = v.y; acc[m][4 * g + 2] = v.z; acc[m][4 * g + 3] = v.w;
        }
}
// this lane's 16-wide chunks of input row q (fp32 [n,176]) as bf16 fragments; rows past n read as zero
__device__ __forceinline__ void load_x_frags(const float* __restrict__ X, long long q, bool valid, int h, uint4 (&xf)[11])
{
    const float* xr = X + q * T_IN + 8 * h;
#pragma unroll
    for (int c = 0; c < 11; ++c) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (valid) { a = *reinterpret_cast<const float4*>(xr + 16 * c); b = *reinterpret_cast<const float4*>(xr + 16 * c + 4); }
        xf[c] = make_uint4(pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w));
    }
}
// the same from rows that already are bf16 ([n,176] bf16 = what hav_field_inputs_fwd_bf16 writes: the fp32 values rounded the way pk_bf16
// rounds them, so the fragments -- and everything computed from them -- have the same bits either way)
__device__ __forceinline__ void load_x_frags_b(const void* __restrict__ X, long long q, bool valid, int h, uint4 (&xf)[11])
{
    const unsigned short* xr = (const unsigned short*)X + q * T_IN + 8 * h;
#pragma unroll
    for (int c = 0; c < 11; ++c) xf[c] = valid ? *reinterpret_cast<const uint4*>(xr + 16 * c) : make_uint4(0u, 0u, 0u, 0u);
}
// layers 1 and 2 in the T orientation: h1f, h2f = bf16 fragments of relu(h1), relu(h2); acc2 = pre-activation of layer 2 (for the heads)
__device__ __forceinline__ void forward_T(const TrainCtx& C, const uint4 (&xf)[11], uint4 (&h1f)[8], uint4 (&h2f)[8], f32x16 (&acc)[4])
{
    bias_tiles(acc, C.vec + TV_B1, C.h);
    mma_seq<11, 4, false>(acc, [&](int c, int m) { return C.fr[TF_W1 + (c * 4 + m) * 64]; }, [&](int c) { return xf[c]; });
    tiles_to_frags(acc, h1f, true);
    bias_tiles(acc, C.vec + TV_B2, C.h);
    mma_seq<8, 4, false>(acc, [&](int ch, int m) { return C.fr[TF_W2 + (ch * 4 + m) * 64]; }, [&](int ch) { return h1f[ch]; });
    tiles_to_frags(acc, h2f, true);
}

// ------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------
template <bool XB>          // XB: X rows are bf16
__global__ void __launch_bounds__(256, 2) mlp_fwd_kernel(float* __restrict__ rf, const float* __restrict__ X, const uint4* __restrict__ fragp,
                                                         const float* __restrict__ vec_in, long long n, int ntiles)
{
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int t = wave; t < ntiles; t += nwaves) {
        // the weights are loop-invariant: opaque per-tile bases keep the compiler from hoisting ~200 fragment loads (800 VGPRs) out of
        // the tile loop and spilling them
        // (the OFFSETS are made opaque, not the pointers: a pointer that has been through an asm statement loses its global address space
        // and every access through it becomes a FLAT instruction -- 248 / 458 of them in these two kernels before round 3)
        int fr_off = lane, vec_off = 0;
        asm volatile("" : "+v"(fr_off), "+s"(vec_off));
        const uint4* fr_ = fragp + fr_off;
        const float* vec = vec_in + vec_off;
        TrainCtx C{fr_, vec, lane, j, h};
        const long long q = (long long)t * 32 + j;
        const bool valid = q < n;
        uint4 xf[11], h1f[8], h2f[8];
        f32x16 acc[4];
        if (XB) load_x_frags_b(X, q, valid, h, xf); else load_x_frags(X, q, valid, h, xf);
        forward_T(C, xf, h1f, h2f, acc);
        // alpha = Wa . relu(h2) + ba on the fp32 accumulators (each lane holds 64 of its query's 128 units)
        float al = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 wv = *reinterpret_cast<const float4*>(vec + TV_WA + 32 * m + 8 * g + 4 * h);
                al = fmaf(wv.x, fmaxf(acc[m][4 * g + 0], 0.f), al); al = fmaf(wv.y, fmaxf(acc[m][4 * g + 1], 0.f), al);
                al = fmaf(wv.z, fmaxf(acc[m][4 * g + 2], 0.f), al); al = fmaf(wv.w, fmaxf(acc[m][4 * g + 3], 0.f), al);
            }
        al += __shfl_xor(al, 32, 64);
        al += vec[TV_BA];
        // g = Wf h2 + bf (two row tiles), c = Wc g + bc
        f32x16 gg[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(vec + TV_BF + 32 * m + 8 * g + 4 * h);
                gg[m][4 * g + 0] = v.x; gg[m][4 * g + 1] = v.y; gg[m][4 * g + 2] = v.z; gg[m][4 * g + 3] = v.w;
            }
        mma_seq<8, 2, false>(gg, [&](int ch, int m) { return C.fr[TF_WF + (ch * 2 + m) * 64]; }, [&](int ch) { return h2f[ch]; });
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        float* row = rf + q * T_OUT;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float gv = gg[m][r];
                c0 = fmaf(vec[TV_WC + ch], gv, c0); c1 = fmaf(vec[TV_WC + 64 + ch], gv, c1); c2 = fmaf(vec[TV_WC + 128 + ch], gv, c2);
                if (valid) row[3 + ch] = gv;
            }
        c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
        if (valid && h == 0) { row[0] = c0 + vec[TV_BC]; row[1] = c1 + vec[TV_BC + 1]; row[2] = c2 + vec[TV_BC + 2]; row[T_OUT - 1] = al; }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, data path + operand stream for the weight gradients
// ------------------------------------------------------------------------------------------------------------------------
// an N-orientation tile (lane = unit / channel, registers = the tile's 32 queries) as two ready-made k-chunk fragments
__device__ __forceinline__ void store_n_tile(uint4* __restrict__ ops, int t, int optile, int lane, const f32x16& d)
{
    uint4* dst = ops + ((size_t)t * OP_TILES + optile) * 128 + lane;
    dst[0] = make_uint4(pk_bf16(d[0], d[1]), pk_bf16(d[2], d[3]), pk_bf16(d[4], d[5]), pk_bf16(d[6], d[7]));
    dst[64] = make_uint4(pk_bf16(d[8], d[9]), pk_bf16(d[10], d[11]), pk_bf16(d[12], d[13]), pk_bf16(d[14], d[15]));
}

template <bool XB>
__global__ void __launch_bounds__(256) mlp_bwd_data_kernel(float* __restrict__ dX, uint4* __restrict__ ops, const float* __restrict__ X,
                                                              const float* __restrict__ d_rf, const uint4* __restrict__ fragp,
                                                              const float* __restrict__ vec_in, long long n, int ntiles)
{
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    // identity B operands: column j of the output picks k slot (h, e) with 8h + e == j (lo) / == j - 16 (hi)
    uint4 idLo = make_uint4(0, 0, 0, 0), idHi = idLo;
    {
        const int eLo = j - 8 * h, eHi = j - 16 - 8 * h;
        uint32_t* lo = reinterpret_cast<uint32_t*>(&idLo);
        uint32_t* hi = reinterpret_cast<uint32_t*>(&idHi);
        if (eLo >= 0 && eLo < 8) lo[eLo >> 1] = 0x3F80u << (16 * (eLo & 1));
        if (eHi >= 0 && eHi < 8) hi[eHi >> 1] = 0x3F80u << (16 * (eHi & 1));
    }
    for (int t = wave; t < ntiles; t += nwaves) {
        int fr_off = lane, vec_off = 0;          // opaque per tile (see mlp_fwd_kernel)
        asm volatile("" : "+v"(fr_off), "+s"(vec_off));
        const uint4* fr_ = fragp + fr_off;
        const float* vec = vec_in + vec_off;
        TrainCtx C{fr_, vec, lane, j, h};
        const long long q = (long long)t * 32 + j;
        const bool valid = q < n;
        uint4 xf[11], h1f[8], h2f[8];
        uint32_t m1n[2], m2n[2];          // relu masks of the N-orientation tiles, 16 bits per tile
        {
            f32x16 acc[4];
            if (XB) load_x_frags_b(X, q, valid, h, xf); else load_x_frags(X, q, valid, h, xf);
            forward_T(C, xf, h1f, h2f, acc);
        }
        // ---- N orientation of the forward quantities: operands swapped, lane = unit ---------------------------------------
        m1n[0] = m1n[1] = m2n[0] = m2n[1] = 0u;
        {
            f32x16 d[4];                        // H1' = relu(X W1^T + b1)
            bias_tiles_n(d, vec + TV_B1, j);
            mma_seq<11, 4, true>(d, [&](int c, int m) { return C.fr[TF_W1 + (c * 4 + m) * 64]; }, [&](int c) { return xf[c]; });
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint32_t bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) { d[m][r] = fmaxf(d[m][r], 0.f); bits |= (d[m][r] > 0.f ? 1u : 0u) << r; }
                m1n[m >> 1] |= bits << (16 * (m & 1));
                store_n_tile(ops, t, OP_H1N + m, lane, d[m]);
            }
            bias_tiles_n(d, vec + TV_B2, j);      // H2' = relu(H1 W2^T + b2)
            mma_seq<8, 4, true>(d, [&](int ch, int m) { return C.fr[TF_W2 + (ch * 4 + m) * 64]; }, [&](int ch) { return h1f[ch]; });
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint32_t bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) { d[m][r] = fmaxf(d[m][r], 0.f); bits |= (d[m][r] > 0.f ? 1u : 0u) << r; }
                m2n[m >> 1] |= bits << (16 * (m & 1));
                store_n_tile(ops, t, OP_H2N + m, lane, d[m]);
            }
        }
        {
            f32x16 d[2];                        // G' = H2 Wf^T + bf
            bias_tiles_n(d, vec + TV_BF, j);
            mma_seq<8, 2, true>(d, [&](int ch, int m) { return C.fr[TF_WF + (ch * 2 + m) * 64]; }, [&](int ch) { return h2f[ch]; });
            store_n_tile(ops, t, OP_GN, lane, d[0]);
            store_n_tile(ops, t, OP_GN + 1, lane, d[1]);
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) {       // X' = X . I (exact: the operand is already bf16); row 176 = ones (bias gradients)
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            d = MFMA_BF16(xf[2 * m], idLo, d);
            if (2 * m + 1 < 11) d = MFMA_BF16(xf[2 * m + 1], idHi, d);
            MFMA_FENCE(d);
            if (m == 5 && j == 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 1.0f;
            }
            store_n_tile(ops, t, OP_XN + m, lane, d);
            __builtin_amdgcn_sched_barrier(0);
        }
        KEEP_ALIVE(xf[10]);
        // ---- upstream gradient: dG = d_rf[3:67] + Wc^T d_c, plus the alpha slot ---------------------------------------
        uint4 dgf[5];
        float dc0 = 0.f, dc1 = 0.f, dc2 = 0.f, da = 0.f;
        {
            const float* gr = d_rf + q * T_OUT;
            if (valid) { dc0 = gr[0]; dc1 = gr[1]; dc2 = gr[2]; da = gr[T_OUT - 1]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int g = 16 * c + 8 * h + e;
                    float x = valid ? gr[3 + g] : 0.f;
                    x = fmaf(vec[TV_WC + g], dc0, x); x = fmaf(vec[TV_WC + 64 + g], dc1, x); x = fmaf(vec[TV_WC + 128 + g], dc2, x);
                    v[e] = x;
                }
                dgf[c] = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
            }
            dgf[4] = make_uint4(h == 0 ? pk_bf16(da, 0.f) : 0u, 0u, 0u, 0u);
        }
        // ---- dH2 (T) -> dZ2 ---------------------------------------------------------------------------------------------
        uint4 dz2f[8];
        {
            f32x16 acc[4];
            zero_tiles(acc);
            mma_seq<5, 4, false>(acc, [&](int c, int m) { return C.fr[TF_WFT + (c * 4 + m) * 64]; }, [&](int c) { return dgf[c]; });
            masked_frags(acc, h2f, dz2f);
        }
        // ---- N: dG', [d_c | d_a]', dZ2' -----------------------------------------------------------------------------------
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            d = MFMA_BF16(dgf[2 * m], idLo, d);
            d = MFMA_BF16(dgf[2 * m + 1], idHi, d);
            MFMA_FENCE(d);
            store_n_tile(ops, t, OP_DGN + m, lane, d);
        }
        {
            const uint4 d4 = make_uint4(h == 0 ? pk_bf16(dc0, dc1) : 0u, h == 0 ? pk_bf16(dc2, da) : 0u, 0u, 0u);
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            d = MFMA_BF16(d4, idLo, d);
            MFMA_FENCE(d);
            KEEP_ALIVE(d4);
            store_n_tile(ops, t, OP_D4N, lane, d);
        }
        {
            f32x16 d[4];
            zero_tiles(d);
            mma_seq<5, 4, true>(d, [&](int c, int m) { return C.fr[TF_WFT + (c * 4 + m) * 64]; }, [&](int c) { return dgf[c]; });
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t bits = (m2n[m >> 1] >> (16 * (m & 1))) & 0xFFFFu;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[m][r] = ((bits >> r) & 1u) ? d[m][r] : 0.f;
                store_n_tile(ops, t, OP_DZ2N + m, lane, d[m]);
            }
        }
        KEEP_ALIVE(dgf[4]);
        // ---- dH1 (T) -> dZ1; dZ1' (N) ------------------------------------------------------------------------------------
        uint4 dz1f[8];
        {
            f32x16 acc[4];
            zero_tiles(acc);
            mma_seq<8, 4, false>(acc, [&](int ch, int m) { return C.fr[TF_W2T + (ch * 4 + m) * 64]; }, [&](int ch) { return dz2f[ch]; });
            masked_frags(acc, h1f, dz1f);
        }
        {
            f32x16 d[4];
            zero_tiles(d);
            mma_seq<8, 4, true>(d, [&](int ch, int m) { return C.fr[TF_W2T + (ch * 4 + m) * 64]; }, [&](int ch) { return dz2f[ch]; });
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t bits = (m1n[m >> 1] >> (16 * (m & 1))) & 0xFFFFu;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[m][r] = ((bits >> r) & 1u) ? d[m][r] : 0.f;
                store_n_tile(ops, t, OP_DZ1N + m, lane, d[m]);
            }
        }
        KEEP_ALIVE(dz2f[7]);
        // ---- dX (T): six row tiles of inputs ---------------------------------------------------------------------------------
        if (dX) {
            float* xr = dX + q * T_IN;
#pragma unroll
            for (int mg = 0; mg < 2; ++mg) {
                f32x16 d[3];
                zero_tiles(d);
                mma_seq<8, 3, false>(d, [&](int ch, int mm) { return C.fr[TF_W1T + (ch * 6 + 3 * mg + mm) * 64]; }, [&](int ch) { return dz1f[ch]; });
#pragma unroll
                for (int mm = 0; mm < 3; ++mm)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = 32 * (3 * mg + mm) + 8 * g + 4 * h;
                        if (valid && col < T_IN)
                            *reinterpret_cast<float4*>(xr + col) = make_float4(d[mm][4 * g], d[mm][4 * g + 1], d[mm][4 * g + 2], d[mm][4 * g + 3]);
                    }
            }
        }
        KEEP_ALIVE(dz1f[7]);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// backward, weight gradients: dW = A'^T-contraction over the queries, slice partials + deterministic reduction
// ------------------------------------------------------------------------------------------------------------------------
struct Role { int a_tile; int nb; int b_tile[7]; int out0; };
__device__ __forceinline__ Role role_of(int r)
{
    Role R;
    if (r < 4) { R.a_tile = OP_DZ1N + r; R.nb = 6; for (int k = 0; k < 6; ++k) R.b_tile[k] = OP_XN + k; R.b_tile[6] = 0; R.out0 = r * 6; }
    else if (r < 8) { R.a_tile = OP_DZ2N + (r - 4); R.nb = 5; for (int k = 0; k < 4; ++k) R.b_tile[k] = OP_H1N + k; R.b_tile[4] = OP_XN + 5; R.b_tile[5] = R.b_tile[6] = 0; R.out0 = 24 + (r - 4) * 5; }
    else if (r < 10) { R.a_tile = OP_DGN + (r - 8); R.nb = 5; for (int k = 0; k < 4; ++k) R.b_tile[k] = OP_H2N + k; R.b_tile[4] = OP_XN + 5; R.b_tile[5] = R.b_tile[6] = 0; R.out0 = 44 + (r - 8) * 5; }
    else { R.a_tile = OP_D4N; R.nb = 7; for (int k = 0; k < 4; ++k) R.b_tile[k] = OP_H2N + k; R.b_tile[4] = OP_GN; R.b_tile[5] = OP_GN + 1; R.b_tile[6] = OP_XN + 5; R.out0 = 54; }
    return R;
}

template <int NB>
__device__ __forceinline__ void weights_strip(const uint4* __restrict__ ops, float* __restrict__ partial, const Role& R, int lane, int t0, int t1, int slice)
{
    f32x16 acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    for (int t = t0; t < t1; ++t) {
        const uint4* base = ops + (size_t)t * OP_TILES * 128 + lane;
        const uint4 a0 = base[R.a_tile * 128], a1 = base[R.a_tile * 128 + 64];
        uint4 b0[NB], b1[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { b0[k] = base[R.b_tile[k] * 128]; b1[k] = base[R.b_tile[k] * 128 + 64]; }
#pragma unroll
        for (int k = 0; k < NB; ++k) { acc[k] = MFMA_BF16(a0, b0[k], acc[k]); acc[k] = MFMA_BF16(a1, b1[k], acc[k]); }
    }
    MFMA_FENCE(acc[NB - 1]);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        float4* dst = reinterpret_cast<float4*>(partial + (((size_t)slice * OUT_TILES + R.out0 + k) * 64 + lane) * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[g] = make_float4(acc[k][4 * g], acc[k][4 * g + 1], acc[k][4 * g + 2], acc[k][4 * g + 3]);
    }
}

__global__ void __launch_bounds__(64) mlp_bwd_weights_kernel(float* __restrict__ partial, const uint4* __restrict__ ops, int ntiles, int slices)
{
    const int lane = threadIdx.x, slice = blockIdx.x, role = blockIdx.y;
    const int per = (ntiles + slices - 1) / slices;
    const int t0 = slice * per, t1 = min(ntiles, t0 + per);
    const Role R = role_of(role);
    if (R.nb == 6) weights_strip<6>(ops, partial, R, lane, t0, t1, slice);
    else if (R.nb == 5) weights_strip<5>(ops, partial, R, lane, t0, t1, slice);
    else weights_strip<7>(ops, partial, R, lane, t0, t1, slice);
}

// element (row rho, column col) of output tile `id`, summed over the slices in slice order
__device__ __forceinline__ float tile_sum(const float* __restrict__ partial, int slices, int id, int rho, int col)
{
    const int h = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    const size_t off = ((size_t)id * 64 + col + 32 * h) * 16 + r;
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += partial[(size_t)k * OUT_TILES * 1024 + off];
    return s;
}

struct MlpGrads { float* W1; float* b1; float* W2; float* b2; float* Wa; float* ba; float* Wf; float* bf; float* Wc; float* bc; };
#define G_W1 0
#define G_B1 (G_W1 + 128 * 176)
#define G_W2 (G_B1 + 128)
#define G_B2 (G_W2 + 128 * 128)
#define G_WF (G_B2 + 128)
#define G_BF (G_WF + 64 * 128)
#define G_WA (G_BF + 64)
#define G_BA (G_WA + 128)
#define G_WC (G_BA + 1)
#define G_BC (G_WC + 192)
#define G_END (G_BC + 3)

__global__ void __launch_bounds__(256) mlp_reduce_kernel(MlpGrads g, const float* __restrict__ partial, int slices, int accumulate)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G_END) return;
    float* dst; float v;
    if (e < G_B1) { const int u = e / 176, i = e % 176; v = tile_sum(partial, slices, (u >> 5) * 6 + (i >> 5), u & 31, i & 31); dst = g.W1 + e; }
    else if (e < G_W2) { const int u = e - G_B1; v = tile_sum(partial, slices, (u >> 5) * 6 + 5, u & 31, 16); dst = g.b1 + u; }
    else if (e < G_B2) { const int q = e - G_W2, vv = q >> 7, u = q & 127; v = tile_sum(partial, slices, 24 + (vv >> 5) * 5 + (u >> 5), vv & 31, u & 31); dst = g.W2 + q; }
    else if (e < G_WF) { const int u = e - G_B2; v = tile_sum(partial, slices, 24 + (u >> 5) * 5 + 4, u & 31, 16); dst = g.b2 + u; }
    else if (e < G_BF) { const int q = e - G_WF, c = q >> 7, u = q & 127; v = tile_sum(partial, slices, 44 + (c >> 5) * 5 + (u >> 5), c & 31, u & 31); dst = g.Wf + q; }
    else if (e < G_WA) { const int c = e - G_BF; v = tile_sum(partial, slices, 44 + (c >> 5) * 5 + 4, c & 31, 16); dst = g.bf + c; }
    else if (e < G_BA) { const int u = e - G_WA; v = tile_sum(partial, slices, 54 + (u >> 5), 3, u & 31); dst = g.Wa + u; }
    else if (e < G_WC) { v = tile_sum(partial, slices, 60, 3, 16); dst = g.ba; }
    else if (e < G_BC) { const int q = e - G_WC, c = q >> 6, ch = q & 63; v = tile_sum(partial, slices, 58 + (ch >> 5), c, ch & 31); dst = g.Wc + q; }
    else { const int c = e - G_BC; v = tile_sum(partial, slices, 60, c, 16); dst = g.bc + c; }
    *dst = accumulate ? *dst + v : v;
}

// ------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------
static int tile_grid(int ntiles, int waves_per_block, int blocks_per_cu)
{
    const int need = (ntiles + waves_per_block - 1) / waves_per_block;
    const int cap = hav_num_cus() * blocks_per_cu;
    return need < cap ? (need < 1 ? 1 : need) : cap;
}

static int mlp_train_fwd_any(float* rf, const void* X, bool xb, const void* blob, int64_t n, void* stream)
{
    if (!rf || !X || !blob || n < 0) return HAV_EINVAL;
    if (n == 0) return 0;
    const int64_t nt = (n + 31) / 32;
    if (nt > 0x7FFFFFFF) return HAV_EUNSUP;
    const float* vec = (const float*)((const char*)blob + (size_t)TF_END * 16);
    if (xb) hipLaunchKernelGGL(mlp_fwd_kernel<true>, dim3(tile_grid((int)nt, 4, 4)), dim3(256), 0, (hipStream_t)stream, rf, (const float*)X, (const uint4*)blob, vec, (long long)n, (int)nt);
    else hipLaunchKernelGGL(mlp_fwd_kernel<false>, dim3(tile_grid((int)nt, 4, 4)), dim3(256), 0, (hipStream_t)stream, rf, (const float*)X, (const uint4*)blob, vec, (long long)n, (int)nt);
    HAV_LAUNCH_CHECK();
    return 0;
}
extern "C" int hav_mlp_train_fwd(float* rf, const float* X, const void* blob, int64_t n, void* stream) { return mlp_train_fwd_any(rf, X, false, blob, n, stream); }
extern "C" int hav_mlp_train_fwd_xbf16(float* rf, const void* Xb, const void* blob, int64_t n, void* stream) { return mlp_train_fwd_any(rf, Xb, true, blob, n, stream); }
static int mlp_train_bwd_any(float* dX, const HavMlpGrads* grads, int accumulate, const void* X_, bool xb, const float* d_rf, const void* blob,
                             void* ops, void* partial, int64_t n, void* stream)
{
    const float* X = (const float*)X_;
    if (!grads || !X || !d_rf || !blob || !ops || !partial || n < 0) return HAV_EINVAL;
    if (!grads->W1 || !grads->b1 || !grads->W2 || !grads->b2 || !grads->Wa || !grads->ba || !grads->Wf || !grads->bf || !grads->Wc || !grads->bc)
        return HAV_EINVAL;
    if (n == 0) return 0;
    const int64_t nt = (n + 31) / 32;
    if (nt > 0x7FFFFFFF) return HAV_EUNSUP;
    hipStream_t st = (hipStream_t)stream;
    const float* vec = (const float*)((const char*)blob + (size_t)TF_END * 16);
    if (xb) hipLaunchKernelGGL(mlp_bwd_data_kernel<true>, dim3(tile_grid((int)nt, 4, 1)), dim3(256), 0, st, dX, (uint4*)ops, X, d_rf, (const uint4*)blob, vec, (long long)n, (int)nt);
    else hipLaunchKernelGGL(mlp_bwd_data_kernel<false>, dim3(tile_grid((int)nt, 4, 1)), dim3(256), 0, st, dX, (uint4*)ops, X, d_rf, (const uint4*)blob, vec, (long long)n, (int)nt);
    HAV_LAUNCH_CHECK();
    const int slices = weight_slices(nt);
    hipLaunchKernelGGL(mlp_bwd_weights_kernel, dim3(slices, N_ROLES), dim3(64), 0, st, (float*)partial, (const uint4*)ops, (int)nt, slices);
    HAV_LAUNCH_CHECK();
    MlpGrads g{(float*)grads->W1, (float*)grads->b1, (float*)grads->W2, (float*)grads->b2, (float*)grads->Wa, (float*)grads->ba,
               (float*)grads->Wf, (float*)grads->bf, (float*)grads->Wc, (float*)grads->bc};
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((G_END + 255) / 256), dim3(256), 0, st, g, (const float*)partial, slices, accumulate);
    HAV_LAUNCH_CHECK();
    return 0;
}
extern "C" int hav_mlp_train_bwd(float* dX, const HavMlpGrads* grads, int accumulate, const float* X, const float* d_rf, const void* blob,
                                 void* ops, void* partial, int64_t n, void* stream)
{
    return mlp_train_bwd_any(dX, grads, accumulate, X, false, d_rf, blob, ops, partial, n, stream);
}
extern "C" int hav_mlp_train_bwd_xbf16(float* dX, const HavMlpGrads* grads, int accumulate, const void* Xb, const float* d_rf, const void* blob,
                                       void* ops, void* partial, int64_t n, void* stream)
{
    return mlp_train_bwd_any(dX, grads, accumulate, Xb, true, d_rf, blob, ops, partial, n, stream);
}
