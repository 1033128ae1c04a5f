// amvs_kernel_common.h -- device helpers shared by the exact (amvs_kernels.hip) and the fast
// (amvs_kernels_fast.hip) sweep kernels.
#pragma once
#include "amvs_kernels.h"
#include "amvs_device.h"

#include <type_traits>

namespace amvs {

// The job table is never written while a sweep kernel runs: reading it through the
// constant address space lets the compiler use scalar loads (s_load) for the poses.
typedef const __attribute__((address_space(4))) Job *JobCP;

// Opaque copy of a uniform pointer.  Loads through the result cannot be hoisted above this
// point, so the row loops re-issue their scalar loads (s_load from the scalar cache) every
// iteration instead of keeping ~90 pose / intrinsics values live and spilling SGPRs into VGPR
// lanes (v_writelane / v_readlane), which cost 14 % of the VALU stream before.
AMVS_DEV JobCP reload(JobCP p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// (Non-temporal hints on the streaming state were measured without effect on MI355X -- 31.8 vs 31.7
// G px-hyp/s -- and are not used.)

// contiguous strip ranges per XCD (blocks are dealt round-robin to XCDs); bijective
AMVS_DEV int xcd_remap(int bid, int nblk)
{
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// strip index -> (job, strip row, strip column); see StepArgs::band_major
AMVS_DEV void strip_of(const StepArgs &a, int t, int &job_id, int &ty, int &tx)
{
    if (a.band_major) {
        const int per_band = a.n_jobs * a.tiles_x;
        ty = t / per_band;
        const int rem = t - ty * per_band;
        job_id = rem / a.tiles_x;
        tx = rem - job_id * a.tiles_x;
    } else {
        const int tiles_per_job = a.tiles_x * a.tiles_y;
        job_id = t / tiles_per_job;
        const int rem = t - job_id * tiles_per_job;
        ty = rem / a.tiles_x;
        tx = rem - ty * a.tiles_x;
    }
}

// per-row validity bits of the last K/2+1 rows packed into one (or two) registers
template <int K, int S> struct Hist {
    // (more than 64 bits from 21 x 21 with 6 sources on: a 128-bit integer, two more shifts per row)
    typedef typename std::conditional<(S * (K / 2 + 1) <= 32), uint32_t,
                                      typename std::conditional<(S * (K / 2 + 1) <= 64), unsigned long long, unsigned __int128>::type>::type T;
};

// Depth state maps carry, in the sign bit, the normal buffer that holds the pixel's current normal
// (StepArgs::nbuf); depths themselves are positive.
AMVS_DEV float depth_untag(float d, unsigned mask) { return __uint_as_float(__float_as_uint(d) & mask); }
AMVS_DEV unsigned depth_buffer(float d) { return __float_as_uint(d) >> 31; }
AMVS_DEV float depth_tag(float d, unsigned buffer) { return __uint_as_float(__float_as_uint(d) | (buffer << 31)); }

// Queued winners: one 32-bit entry each (the queue of a wave is 2 x 64 entries = 512 bytes of LDS; round 4 --
// 8-byte entries before -- so that the paired-band exchange of the 11 x 11 patch fits beside the rings at four
// workgroups per CU).  Pixel indices are below 2^29 (amvs_create).
//
// Normal update of `n` queued refinement winners (entries head .. head+n-1 of the ring `nq`), one
// per lane: normal <- normalize(normal + randn * range)   (mvs_patchmatch.py:475-476).  Entry: pixel
// index with the winner's normal buffer in bit 31; the pixel's hash (a pure function of the pixel and the
// launch's stream key) is formed again here.
AMVS_DEV uint32_t refine_entry(int pc, unsigned buf_c) { return (unsigned)pc | (buf_c << 31); }

AMVS_DEV void refine_normals(const uint32_t *nq, int head, int n, int lane, float *nbuf0, float *nbuf1, float normal_range,
                             StreamKey key, int hw)
{
    (void)hw;                                                  // (pixels per map: the index-checked build's extent)
    if (lane < n) {
        const uint32_t e = nq[(head + lane) & (2 * AMVS_WAVE - 1)];
        const uint32_t pc = e & 0x7FFFFFFFu;
        float *np = ((e >> 31) ? nbuf1 : nbuf0) + 3ll * AMVS_IDX((int)pc, hw);
        float g0, g1, g2;
        rng_normals3(pixel_hash(pc, key), g0, g1, g2);
        float cn0 = np[0] + g0 * normal_range;
        float cn1 = np[1] + g1 * normal_range;
        float cn2 = np[2] + g2 * normal_range;
        normalize3(cn0, cn1, cn2);
        np[0] = cn0; np[1] = cn1; np[2] = cn2;
    }
}

// Propagation winners (mvs_patchmatch.py:452-455), queued like the refinement winners and moved 64 at
// a time: pixel pc takes the pre-step normal of its neighbour pn = pc + noff (zero outside the image, F.pad
// :431-442) into the buffer it does not currently use.  Entry: pc | neighbour inside the image << 29 |
// neighbour's buffer << 30 | pc's current buffer << 31.
// Nothing writes a neighbour's CURRENT normal during a propagation launch and nothing reads the
// buffer a winner writes (StepArgs::nbuf), so the move may happen any time before the launch ends.
AMVS_DEV uint32_t propagate_entry(int pc, unsigned buf_c, bool inb_c, unsigned buf_n)
{
    return (unsigned)pc | ((inb_c ? 1u : 0u) << 29) | (buf_n << 30) | (buf_c << 31);
}

AMVS_DEV void propagate_normals(const uint32_t *nq, int head, int n, int lane, float *nbuf0, float *nbuf1, int noff, int hw)
{
    (void)hw;
    if (lane < n) {
        const uint32_t e = nq[(head + lane) & (2 * AMVS_WAVE - 1)];
        const int pc = AMVS_IDX((int)(e & 0x1FFFFFFFu), hw);
        const bool inb_c = (e >> 29) & 1u;
        const int pn = AMVS_IDX(inb_c ? pc + noff : 0, hw);
        // (three consecutive dwords each way: hipcc merges them into one dwordx3 access)
        const float *src = (((e >> 30) & 1u) ? nbuf1 : nbuf0) + 3ll * pn;
        const float t0 = src[0], t1 = src[1], t2 = src[2];
        float *dst = ((e >> 31) ? nbuf0 : nbuf1) + 3ll * pc;
        dst[0] = inb_c ? t0 : 0.0f;
        dst[1] = inb_c ? t1 : 0.0f;
        dst[2] = inb_c ? t2 : 0.0f;
    }
}

// AMVS_WG_WAVES horizontally adjacent strips share one workgroup (one CU, started together) and
// re-align with a barrier every AMVS_WG_SYNC_ROWS rows: x-neighbours sample overlapping epipolar
// bands of the sources, and they only share those lines in L1 / L2 while they work on the same rows.
// Four waves (one per SIMD, so the workgroup granularity costs no occupancy) re-aligned every 8 rows
// measured +1.8 % over single-wave workgroups (39.5 vs 38.85 G px-hyp/s; 16 rows the same, no
// barrier +0.5 %).  With 10 waves the HBM-side traffic halves -- and the launch
// gets slower, because a 5- or 10-wave workgroup fits only twice / once per CU.
#ifndef AMVS_WG_WAVES
#define AMVS_WG_WAVES 4
#endif
#ifndef AMVS_WG_SYNC_ROWS
#define AMVS_WG_SYNC_ROWS 8
#endif
// Paired-band schedule (StepArgs::paired): a workgroup is AMVS_PAIR_COLS strip columns x 2 vertically
// adjacent bands (2 -> 4 waves, 4 -> 8 waves); compiled where the exchange rows fit beside the rings at four
// workgroups per CU
#ifndef AMVS_PAIR_COLS
#define AMVS_PAIR_COLS 2
#endif
constexpr int PAIR_WAVES = 2 * AMVS_PAIR_COLS;
constexpr bool step_pair_supported_ks(int K, int S) { return K >= 5 && K <= 11 && S <= 4; }

#if defined(AMVS_HSUM_LDS) && AMVS_WG_WAVES > 1
#error "the LDS horizontal-sum variant keeps one exchange buffer per workgroup: build it with -DAMVS_WG_WAVES=1"
#endif

}  // namespace amvs
