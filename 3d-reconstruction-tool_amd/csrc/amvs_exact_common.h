// amvs_exact_common.h -- device code shared by the translation units of the EXACT arithmetic:
// amvs_kernels.hip (sweep step, plane sweep for the compiled patch sizes) and amvs_generic.hip (the
// run-time patch size fallback): the per-source sampler that reproduces the reference's float32 chain
// (mvs_patchmatch.py:341-377) operation for operation.
#pragma once
#include "amvs_kernel_common.h"

namespace amvs {

// Where pm_step tests the validity of its lean reciprocals / square roots: once per row and stage
// (1) or after every operation / source (0, measured 2 % faster there).  The plane sweep always
// uses the per-row form (+3 %).  AMVS_RELOAD_STRIDE sources share one opaque pointer copy (which
// is also a scheduling barrier), so their arithmetic can interleave; 1 is fastest (registers).
// (Sampling the sources two at a time with packed fp32 arithmetic -- v_pk_fma/mul/add_f32 on pose
// pairs interleaved in the job table, 5 % fewer VALU instructions, bit-identical -- was measured
// 1-2 % slower in three different states of this kernel and is no longer carried in the source.)
#ifndef AMVS_RELOAD_STRIDE
#define AMVS_RELOAD_STRIDE 1
#endif
#ifndef AMVS_PM_ROW_CHECK_SAMPLING
#define AMVS_PM_ROW_CHECK_SAMPLING 0
#endif
#ifndef AMVS_PM_ROW_CHECK_NCC
#define AMVS_PM_ROW_CHECK_NCC 0
#endif

// The scalar operands of one source: one 64-byte record of the job table (SrcEntry), i.e. one
// batch of scalar loads and one wait.
struct SrcScalars { float R[9], t[3]; unsigned long long img; };
AMVS_DEV SrcScalars load_src_scalars(JobCP jr, int s, bool u8)
{
    SrcScalars c;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = jr->src[s].R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] = jr->src[s].t[i];
    c.img = u8 ? jr->src[s].pairs : jr->src[s].gray;
    return c;
}

#ifndef AMVS_STEP_PRIO
#define AMVS_STEP_PRIO true
#endif
// Sample all S sources of one pixel.
// LEAN / `ok`: optimistic lean reciprocal (amvs_device.h).  SRC_CHECK = true tests `ok` after each
// source's geometry and repeats that geometry with IEEE arithmetic (one wave-uniform branch per
// source); SRC_CHECK = false leaves the test to the caller (one branch per row).
// PRIO (the sweep step): raised issue priority from here until the gathers are requested -- see
// fast_sample_sources (amvs_kernels_fast.hip).
template <int S, bool U8, bool LEAN, bool SRC_CHECK, bool PRIO = false>
AMVS_DEV unsigned sample_sources(JobCP job, const StepArgsBase &a, const SampleConsts &sc, const float *lut,
                                 Vec3 Pw, bool live, float (&v)[S], bool &ok)
{
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    unsigned okbits = 0u;
    JobCP jr = job;
    // the shared intrinsics: loaded once per row, with the reference pose (same scalar-load batch)
    float Kc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Kc[i] = jr->K[i];
    // Geometry of every source first, each gather issued as soon as its address exists, then the
    // decodes: the S gather latencies overlap (one exposed wait per row instead of S: +3 %).  The
    // taps' weights wait in registers meanwhile (4 per source) -- affordable since the window-sum
    // stage, not the sampling stage, sets this kernel's register peak.
    TapGeom<U8> tg[S];
    TapRaw<U8> tr[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        bool valid;
        if (s % AMVS_RELOAD_STRIDE == 0) jr = reload(jr);
        const SrcScalars c = load_src_scalars(jr, s, U8);
        if constexpr (SRC_CHECK) {
            bool ok_s = true;
            tg[s] = sample_geom<U8, true>(Kc, c.R, c.t, sc, Pw, live, valid, ok_s);
            if (__builtin_expect(!__all(ok_s), 0)) tg[s] = sample_geom<U8, false>(Kc, c.R, c.t, sc, Pw, live, valid, ok_s);
        } else {
            tg[s] = sample_geom<U8, LEAN>(Kc, c.R, c.t, sc, Pw, live, valid, ok);
        }
        okbits |= valid ? (1u << s) : 0u;
        tr[s] = sample_load<U8>(c.img, tg[s], sc.W + 2 * AMVS_PAIR_BORDER);
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = sample_finish<U8>(tr[s], tg[s], lut, live);
    return okbits;
}

// Optimistic sampling of a row: lean arithmetic first; the IEEE repeat only when some lane's
// projection depth left the range the lean reciprocal is verified for (amvs_device.h).
template <int S, bool U8, bool ROW_CHECK, bool PRIO = false>
AMVS_DEV unsigned sample_sources_checked(JobCP job, const StepArgsBase &a, const SampleConsts &sc, const float *lut,
                                         Vec3 Pw, bool live, float (&v)[S])
{
    bool ok = true;
    if constexpr (!ROW_CHECK) return sample_sources<S, U8, true, true, PRIO>(job, a, sc, lut, Pw, live, v, ok);
    unsigned okbits = sample_sources<S, U8, true, false, PRIO>(job, a, sc, lut, Pw, live, v, ok);
    if (__builtin_expect(!__all(ok), 0))
        okbits = sample_sources<S, U8, false, false, PRIO>(reload(job), a, sc, lut, Pw, live, v, ok);
    return okbits;
}

}  // namespace amvs
