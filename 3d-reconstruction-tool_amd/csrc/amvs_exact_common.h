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

// ------------------------------------------------------------------ window sums --
// k x k window sums of v, v*v and r*v for S sources at once, from the per-lane vertical rings.
//   1. column sums, top -> bottom: plain sum for v, fma chains for v*v and r*v   (registers)
//   2. row sums, right -> left:   sum_{j=K-1..0} c(lane+j)
// Step 2 needs the column sums of the K-1 lanes to the right.  Two implementations with the
// same summation order (and therefore the same bits):
//   DPP  (default): K-1 `v_add_f32_dpp ... wave_shl:1` per sum, no LDS.  A DPP add costs 3x a
//        plain add to issue on gfx950 (tools/shift_rate.hip) but needs no extra registers.
//   LDS  (-DAMVS_HSUM_LDS): every lane stores its 3*S column sums as NV4 float4 at a lane stride
//        of 12 (S<=4) or 20 (S>4) dwords -- conflict-free for ds_write_b128 / ds_read_b128 -- and
//        reads its neighbours' with (K-1)*NV4 ds_read_b128 (one wave per workgroup and in-order
//        LDS: no barrier).  Measured within +-1.5 % of DPP (26.9 vs 26.5 G px-hyp/s before the
//        row pipeline) at ~25 more VGPRs, so it is not the default.
template <int S> struct HSum { static constexpr int NV4 = S <= 4 ? 3 : 5; };

// Where the vertical rings live.  The ref ring and the rings of the first NL sources are kept in
// LDS ([ring][slot][lane], rotating slot, conflict-free one-dword-per-lane accesses) instead of
// registers: 7*(NL+1) fewer VGPRs held across the whole row loop, which is what lets the k=7, S=4
// kernel run five waves per SIMD (the window-sum stage is its register peak today); the other
// sources stay in shifting register rings.
#ifndef AMVS_RING_LDS_SOURCES
#define AMVS_RING_LDS_SOURCES 2
#endif
// (21 x 21 and up: ONE source in LDS -- the rings of a four-wave workgroup must stay below the 64 KB a workgroup
// may take; the registers hold the others, at two waves per SIMD)
constexpr int ring_lds_sources(int K) { return K <= 19 ? AMVS_RING_LDS_SOURCES : 1; }
template <int K, int S> struct Ring {
    static constexpr int NL = ring_lds_sources(K) < S ? ring_lds_sources(K) : S;       // sources in LDS
    static constexpr int NR = S - NL > 0 ? S - NL : 1;                                  // register rings (>=1 for the type)
    static constexpr bool REF_IN_LDS = NL > 0;
};

template <int K, int S>
AMVS_DEV void ring_push(float *lring, int lane, int wslot, float (&ring_r)[K], float (&ring_v)[Ring<K, S>::NR][K],
                        float rv, const float (&v)[S])
{
    constexpr int NL = Ring<K, S>::NL;
    if (Ring<K, S>::REF_IN_LDS) {
        lring[wslot * AMVS_WAVE + lane] = rv;
#pragma unroll
        for (int s = 0; s < NL; ++s) lring[((s + 1) * K + wslot) * AMVS_WAVE + lane] = v[s];
    } else {
#pragma unroll
        for (int i = 0; i < K - 1; ++i) ring_r[i] = ring_r[i + 1];
        ring_r[K - 1] = rv;
    }
#pragma unroll
    for (int s = NL; s < S; ++s) {
#pragma unroll
        for (int i = 0; i < K - 1; ++i) ring_v[s - NL][i] = ring_v[s - NL][i + 1];
        ring_v[s - NL][K - 1] = v[s];
    }
}

// `oldest` = LDS slot of the oldest row (the next write slot once the ring is full)
// REV: the rings were filled walking UP the image (paired-band schedule, bottom-up wave): ring entry i is
// then row (K-1-i) of the window, and the column sums take them newest first -- the same top -> bottom
// order of the same values.
// REFSUMS = false (the plane sweep, whose reference statistics come from the precomputed maps): br / brr are
// not formed.  (The sweep STEP keeps forming them from its ring: loading the two maps instead -- 8 B per pixel and
// launch of coalesced traffic for 2K - 1 adds / FMAs and 2(K - 1) cross-lane adds less per row -- was measured
// slower there in round 4: 37.7 against 40.4-40.7 G px-hyp/s at 7x7, 30.6 against 32.3 at 11x11; the step is not
// bound by its instruction count alone, the plane sweep is.)
template <int K, int S, bool REV = false, bool REFSUMS = true>
AMVS_DEV void window_sums(const float *lring, int oldest, const float (&ring_r)[K],
                          const float (&ring_v)[Ring<K, S>::NR][K], float4 *hbuf, int lane,
                          float (&bv)[S], float (&bvv)[S], float (&brv)[S], float &br, float &brr)
{
    constexpr int NV4 = HSum<S>::NV4;
    constexpr int NL = Ring<K, S>::NL;
    float cs[NV4 * 4];
#pragma unroll
    for (int i = 0; i < NV4 * 4; ++i) cs[i] = 0.0f;
    // ref values of the window, oldest -> newest
    float rr[K];
    int slot[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const int j = REV ? K - 1 - i : i;              // window row i (top -> bottom) = ring age j
        slot[i] = oldest + j >= K ? oldest + j - K : oldest + j;
        rr[i] = Ring<K, S>::REF_IN_LDS ? lring[slot[i] * AMVS_WAVE + lane] : ring_r[j];
    }
    // window sums of the reference image itself (r, r*r): the statistics mean1 / var1 of
    // mvs_patchmatch.py:403,406, recomputed from the ring (same order as box_stats_kernel, so the
    // same bits) instead of streaming two precomputed maps (8 B per pixel and step)
    if constexpr (REFSUMS) {
        float cr = rr[0], crr = rr[0] * rr[0];
#pragma unroll
        for (int i = 1; i < K; ++i) { cr = cr + rr[i]; crr = __builtin_fmaf(rr[i], rr[i], crr); }
        float ar = cr, arr = crr;
#pragma unroll
        for (int j = 1; j < K; ++j) { ar = wave_shl1(ar) + cr; arr = wave_shl1(arr) + crr; }
        br = ar; brr = arr;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        float vv[K];
#pragma unroll
        for (int i = 0; i < K; ++i)
            vv[i] = s < NL ? lring[((s + 1) * K + slot[i]) * AMVS_WAVE + lane] : ring_v[s < NL ? 0 : s - NL][REV ? K - 1 - i : i];
        float cv = vv[0];
        float cvv = vv[0] * vv[0];
        float crv = rr[0] * vv[0];
#pragma unroll
        for (int i = 1; i < K; ++i) {
            cv = cv + vv[i];
            cvv = __builtin_fmaf(vv[i], vv[i], cvv);
            crv = __builtin_fmaf(rr[i], vv[i], crv);
        }
        cs[3 * s] = cv; cs[3 * s + 1] = cvv; cs[3 * s + 2] = crv;
    }
#ifndef AMVS_HSUM_LDS
    (void)hbuf;
    float acc[NV4 * 4];
#pragma unroll
    for (int i = 0; i < 3 * S; ++i) acc[i] = cs[i];
#pragma unroll
    for (int j = 1; j < K; ++j)
#pragma unroll
        for (int i = 0; i < 3 * S; ++i) acc[i] = wave_shl1(acc[i]) + cs[i];
#else
    float4 *mine = hbuf + lane * NV4;
#pragma unroll
    for (int q = 0; q < NV4; ++q) mine[q] = make_float4(cs[4 * q], cs[4 * q + 1], cs[4 * q + 2], cs[4 * q + 3]);
    __builtin_amdgcn_wave_barrier();
    float acc[NV4 * 4];
    // one float4 column group at a time: its K-1 neighbour reads are issued together, then
    // summed right -> left
#pragma unroll
    for (int q = 0; q < NV4; ++q) {
        float4 t[K - 1];
#pragma unroll
        for (int j = 1; j < K; ++j) t[j - 1] = mine[j * NV4 + q];
        float4 r = t[K - 2];
#pragma unroll
        for (int j = K - 2; j >= 1; --j) {
            r.x += t[j - 1].x; r.y += t[j - 1].y; r.z += t[j - 1].z; r.w += t[j - 1].w;
        }
        acc[4 * q] = r.x + cs[4 * q]; acc[4 * q + 1] = r.y + cs[4 * q + 1];
        acc[4 * q + 2] = r.z + cs[4 * q + 2]; acc[4 * q + 3] = r.w + cs[4 * q + 3];
        // pin the sums here (LLVM otherwise sinks the adds to their first use and keeps every
        // neighbour read live across the NCC epilogue)
        asm volatile("" : "+v"(acc[4 * q]), "+v"(acc[4 * q + 1]), "+v"(acc[4 * q + 2]), "+v"(acc[4 * q + 3]));
    }
    __builtin_amdgcn_wave_barrier();
#endif
#pragma unroll
    for (int s = 0; s < S; ++s) { bv[s] = acc[3 * s]; bvv[s] = acc[3 * s + 1]; brv[s] = acc[3 * s + 2]; }
}

// lanes beyond the strip read (K-1) entries past lane 63: keep them defined
template <int K, int S>
AMVS_DEV void window_sums_init(float4 *hbuf, int lane)
{
#ifndef AMVS_HSUM_LDS
    (void)hbuf; (void)lane;
    return;
#endif
    constexpr int NV4 = HSum<S>::NV4;
    if (lane < K - 1)
#pragma unroll
        for (int q = 0; q < NV4; ++q) hbuf[(AMVS_WAVE + lane) * NV4 + q] = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace amvs
