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
