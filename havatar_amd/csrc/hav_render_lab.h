// hav_render_lab.h -- the diagnostic builds of the ray-march kernel, kept out of hav_render.hip.  The shipped library defines none of the
// symbols below: every macro then expands to nothing (ABL(bit) to `false`) and hav_render.hip contains the production code only.
//   -DHAV_LAB          timing ablations and wave start-up stagger from the environment (HAV_ABLATE bit mask, HAV_STAGGER; tools/ablate.sh,
//                      tools/power_probe.sh, tools/clock_probe.sh).  Results are WRONG when a bit is set, only the time matters:
//                      1 no tri-plane gather | 2 no sin / cos | 4 no feature-parking MFMAs | 8 no composited hidden units (pair kernel) |
//                      16 no layer-1 MFMAs | 32 no layer-2 MFMAs | 64 no head dots | 128 no resampling (pair kernel) |
//                      1024 all waves park in one L2-resident slot
//   -DHAV_PROFILE      per-phase s_memtime sums (tools/phase_profile.sh): wave-uniform deltas summed per phase, added to g_prof at kernel
//                      exit; waits are attributed to the phase in which the s_waitcnt / s_nop sits.  hav_debug_read_prof() reads them.
//   -DHAV_DEBUG_TRACE  twelve per-lane stage values of EVERY tile evaluation go to a trace buffer [12 quantities x 2 half-waves][rays][80
//                      slots] behind the merged-depth dump (tools/stress_diag.py DUMP=41), compared between launches on the host:
//                      0 gather checksum, 1 after layer 1, 2 after layer 2, 3 heads, 4 depth, 5 own bone weight, 6 partner's, 7 warped point,
//                      8 den, 9 n0, 10 n1, 11 p + p1.  This is the build that traced the rare run-to-run difference of rounds 2-3 to the
//                      compiler's IEEE division sequence (docs/history/DESIGN_r1-r4.md 3.12).
#pragma once

#ifdef HAV_LAB
#define ABL(bit) ((a.ablate & (bit)) != 0)
#define LAB_ENV_INT(name) (getenv(name) ? atoi(getenv(name)) : 0)
#define LAB_STAGGER(wave_) for (int q_ = 0; q_ < (wave_) * a.stagger; ++q_) __builtin_amdgcn_s_sleep(1)      // start-up delay per wave index, 64 cycles each
#else
#define ABL(bit) false
#define LAB_ENV_INT(name) 0
#define LAB_STAGGER(wave_) do { } while (0)
#endif

// -DHAV_GATHER_MODEL (round 6, with -DHAV_LAB): TIMING MODEL of a tri-plane gather on the matrix cores -- results are WRONG.  The 32 rays of a
// tile touch ~10 x 2 texels per plane (plane resolution 128 against 512 pixels: neighbouring rays are a quarter texel apart), yet every lane
// fetches its own 8 taps x 256 B through the texture path: 128 KB per tile for ~12 KB of distinct data.  The model replaces the 128 loads +
// 256 packed FMAs by what an interpolation-as-matrix-product would cost: out[unit][query] = sum_t P[unit][t] . Wt[t][query] over a 16-column
// x 2-row footprint per plane (K = 4 chunks of 16), P pre-split into fp16 hi / lo fragments + MX records (read from global memory, coalesced
// 1-KB rows), Wt = the bilinear weights scattered into the footprint's slots (selects + the hi / lo / tail split + 3 conversions on the VALU),
// 48 fp16 MFMAs + 12 block-scaled ones per tile through mfma_split2x itself.  tools/ab_march.py compares it with the shipped kernel.
#ifdef HAV_GATHER_MODEL
#define LAB_GATHER_MODEL 1
template <typename ACC>
__device__ __forceinline__ void lab_gather_model(ACC& acc1, const float4* const (&tp)[8], const float (&tw)[8], int lane, const float* pplanes, long long plane_floats)
{
    const int h = lane >> 5;
    // footprint origin: wave-wide minimum of the lanes' first-tap columns (two DPP reductions + broadcasts, as the real thing would need per plane)
    long long off4 = (long long)(tp[0] - reinterpret_cast<const float4*>(pplanes));
    float fo = (float)(int)(off4 & 0xFFFFF), f1 = (float)(int)((tp[4] - reinterpret_cast<const float4*>(pplanes)) & 0xFFFFF);
    fo = -row16_max(-fo); f1 = -row16_max(-f1);
    fo = fminf(fminf(read_lane(fo, 0), read_lane(fo, 16)), fminf(read_lane(fo, 32), read_lane(fo, 48)));
    f1 = fminf(fminf(read_lane(f1, 0), read_lane(f1, 16)), fminf(read_lane(f1, 32), read_lane(f1, 48)));
    long long base4 = (long long)__builtin_amdgcn_readfirstlane((int)fo) + (off4 & ~0xFFFFFll);
    base4 = __builtin_amdgcn_readfirstlane((int)(base4 & 0x7FFFFFFF));
    const long long lim4 = plane_floats / 4 - 16384;          // the model reads 32 KB of fragments + 15 KB of records behind the origin
    if (base4 > lim4) base4 = lim4;
    if (base4 < 0) base4 = 0;
    const uint4* frag = reinterpret_cast<const uint4*>(pplanes) + base4;
    const unsigned int* rec = reinterpret_cast<const unsigned int*>(pplanes) + 4 * base4 + 8192 + ((int)f1 & 3) * 4;
    const int dx0 = (int)((off4 >> 2) & 7) + 1, dx1 = (int)(((tp[4] - tp[0]) >> 2) & 7) + 1;          // column of this lane's first tap inside the footprint
    mfma_split2x<4, false>(acc1, frag, rec, lane, [&](int c, float (&v)[8]) {          // chunk c = (plane c >> 1, tap row c & 1)
        const int d = (c >> 1) ? dx1 : dx0;
        const float w0 = tw[4 * (c >> 1) + 2 * (c & 1)], w1 = tw[4 * (c >> 1) + 2 * (c & 1) + 1];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int col = 8 * h + e; v[e] = (col == d) ? w0 : ((col == d + 1) ? w1 : 0.f); }
    });
}
#else
#define LAB_GATHER_MODEL 0
template <typename ACC, typename TP, typename TW> __device__ __forceinline__ void lab_gather_model(ACC&, const TP&, const TW&, int, const float*, long long) {}
#endif

#ifdef HAV_PROFILE
#define HAV_NPROF 24      // 0-9 phases | 10 wave lifetime | 11.. free
__device__ unsigned long long g_prof[HAV_NPROF];
struct ProfCtx { unsigned long long t, acc[HAV_NPROF], t0; };
#define PROF_ARG , ProfCtx& P
#define PROF_PASS , P
#define PROF_DECL() ProfCtx P
#define PROF_BEGIN() do { for (int i_ = 0; i_ < HAV_NPROF; ++i_) P.acc[i_] = 0; P.t0 = __builtin_amdgcn_s_memtime(); P.t = P.t0; } while (0)
#define PROF_END(lane_) do { P.acc[10] = __builtin_amdgcn_s_memtime() - P.t0; \
                             if ((lane_) == 0) for (int i_ = 0; i_ < HAV_NPROF; ++i_) atomicAdd(&g_prof[i_], P.acc[i_]); } while (0)
#define TICK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                     P.acc[i] += t_ - P.t; P.t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
// phase-timing build only: copy (and clear) the per-phase cycle sums
extern "C" int hav_debug_read_prof(unsigned long long* out)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * HAV_NPROF);
    unsigned long long z[HAV_NPROF] = {};
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
    return (int)e;
}
#else
#define PROF_ARG
#define PROF_PASS
#define PROF_DECL() do { } while (0)
#define PROF_BEGIN() do { } while (0)
#define PROF_END(lane_) do { } while (0)
#define TICK(i) do { } while (0)
#endif

#ifdef HAV_DEBUG_TRACE
struct DbgTrace { float* p; long long plane; };
#define DBG_ARG , DbgTrace dtr
#define DBG_PASS(x) , x
#define DBG_NONE() DbgTrace{nullptr, 0}
#define DBG_PUT(i, v) do { if (dtr.p) dtr.p[(i) * dtr.plane] = (v); } while (0)
#define DBG_STAGE(i, t4) do { float s_ = 0.f; for (int m_ = 0; m_ < 4; ++m_) for (int r_ = 0; r_ < 16; ++r_) s_ += (t4)[m_][r_]; DBG_PUT(i, s_); } while (0)
// trace slot of (this lane's ray, tile): behind the [rays][S_fp] depth dump; 80 slots per ray = S_c coarse tiles + S_f new samples.  Uses the
// block kernel's local names (a, rayok, CACHE, h, gr).
#define DBG_TILE(slot_) DbgTrace{(a.dbg_zfine && rayok && CACHE == 2 && a.p.S_c + a.p.S_f <= 80) ? a.dbg_zfine + (long long)a.p.B * a.p.R * a.S_fp + (12 * h) * ((long long)a.p.B * a.p.R * 80) + gr * 80 + (slot_) : nullptr, (long long)a.p.B * a.p.R * 80}
#else
#define DBG_ARG
#define DBG_PASS(x)
#define DBG_NONE()
#define DBG_PUT(i, v) do { } while (0)
#define DBG_STAGE(i, t4) do { } while (0)
#define DBG_TILE(slot_)
#endif
