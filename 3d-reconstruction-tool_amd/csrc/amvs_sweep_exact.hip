// amvs_sweep_exact.hip -- the plane sweep (_plane_sweep_torch, src/core/dense_stereo.py:262-310) in the EXACT
// arithmetic, compiled patch sizes.  Its own translation unit (round 4) so that it can be built with another
// instruction scheduler than the sweep step (csrc/Makefile), as amvs_sweep_fast.hip is.
//
// _plane_sweep_torch: for each of D fronto-parallel planes count the neighbours with NCC > thresh and z > 0.1;
// keep the first plane with the highest count.  A wave keeps the running best of its strip and plane chunk in
// LDS as a 16-bit key ((count << 12) | (4095 - plane index inside the chunk): a plain max implements
// torch.max's first-index rule; 16 bits keep the strip's keys at 4 KB, which is what lets four waves per SIMD
// fit the 160 KB of LDS) and never materialises the (D,H,W) volume the reference allocates (:262).  Chunks are
// merged through atomicMax on the 32-bit key (count << 16) | (65535 - plane).
//
// The kernel is VALU-issue-bound (the sources stay in L2).  Round 4 removed instructions WITHOUT changing
// any value the reference's float32 chain produces:
//   * the reference image's window statistics mean1 / var1 (dense_stereo.py:333-338) do not depend on the plane:
//     they are loaded from the maps box_stats_kernel precomputes (the same sums in the same order, hence the
//     same bits) instead of being re-formed for every plane (2 K - 1 adds / FMAs + 2 (K - 1) DPP adds per row);
//   * the vote ncc > thresh (:303) is decided for almost every pixel by a conservative squared comparison
//     (sure_yes / sure_no below, with a margin that covers every rounding of both evaluations); only when a lane
//     of the wave falls inside the margin does the wave run the square root and the correctly rounded quotient
//     -- the same decisions by construction (~28 -> ~12 instructions per source and row);
//   * the validity test with infinite bounds is u < +inf, v < +inf (sample_geom<..., NOBOUNDS>); the range
//     test of the lean reciprocals runs once per row on the min / max of |z| (TRACK); the column part of the
//     rays (px * Kinv[0], [3], [6]) is formed once per strip.
// Measured on MI355X (BASELINE config 2 in the exact arithmetic, bench.py --workload planesweep --mode exact,
// G px-hyp/s, alternated in one run): round 3's kernel 41.7 (11.30 ms) -> 47.3-47.6 with the four changes
// above -> 48.0-48.1 built with LLVM's occupancy-driven iterative scheduler (csrc/Makefile) -> 48.4 with two
// sources per opaque job-pointer copy (AMVS_RELOAD_STRIDE 2 below: their geometry interleaves).  Not adopted:
// 3 / 4 / 1 source rings in LDS instead of 2: 46.7 / 44.7 / 47.7; max-ilp scheduler 45.8; strips of 24 / 20
// rows 46.8 / 46.6; 16 / 22 planes per wave 45.9 / 44.2 (automatic: 13).
#define AMVS_RELOAD_STRIDE 2
#define AMVS_TU_ID 4
#include "amvs_exact_common.h"

namespace amvs {

// rays = [x,y,1] @ K_inv.T with the column products formed once per strip: the same operations in the same
// order as backproject (amvs_device.h)
struct RayCols { float c0, c1, c2; };

template <class KP, class RP, class TP>
AMVS_DEV Vec3 backproject_cols(KP Kinv, RP Rref, TP tref, const RayCols &rc, int y, float d)
{
    const float py = (float)y;
    float q0 = __builtin_fmaf(1.0f, Kinv[2], __builtin_fmaf(py, Kinv[1], rc.c0)) * d - tref[0];
    float q1 = __builtin_fmaf(1.0f, Kinv[5], __builtin_fmaf(py, Kinv[4], rc.c1)) * d - tref[1];
    float q2 = __builtin_fmaf(1.0f, Kinv[8], __builtin_fmaf(py, Kinv[7], rc.c2)) * d - tref[2];
    Vec3 w;
    w.x = __builtin_fmaf(q2, Rref[6], __builtin_fmaf(q1, Rref[3], q0 * Rref[0]));
    w.y = __builtin_fmaf(q2, Rref[7], __builtin_fmaf(q1, Rref[4], q0 * Rref[1]));
    w.z = __builtin_fmaf(q2, Rref[8], __builtin_fmaf(q1, Rref[5], q0 * Rref[2]));
    return w;
}

// all S sources of one pixel: sample_sources (amvs_exact_common.h) with the plane sweep's validity test and one
// range test of the lean reciprocals per row
template <int S, bool U8, bool LEAN>
AMVS_DEV unsigned sweep_sample_sources(JobCP job, const SampleConsts &sc, const float *lut, Vec3 Pw, bool live,
                                       float (&v)[S], bool &ok)
{
    unsigned okbits = 0u;
    JobCP jr = job;
    float Kc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) Kc[i] = jr->K[i];
    TapGeom<U8> tg[S];
    TapRaw<U8> tr[S];
    float zlo = 1.0f, zhi = 1.0f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        bool valid, unused = true;
        if (s % AMVS_RELOAD_STRIDE == 0) jr = reload(jr);
        const SrcScalars c = load_src_scalars(jr, s, U8);
        tg[s] = sample_geom<U8, LEAN, true, LEAN>(Kc, c.R, c.t, sc, Pw, live, valid, unused, &zlo, &zhi);
        okbits |= valid ? (1u << s) : 0u;
        tr[s] = sample_load<U8>(c.img, tg[s], sc.W + 2 * AMVS_PAIR_BORDER);
    }
    if constexpr (LEAN) ok = (zlo >= 0x1p-95f) & (zhi < 0x1p96f);
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = sample_finish<U8>(tr[s], tg[s], lut, live);
    return okbits;
}

// The vote of one source, ncc = qdiv(cov, sqrt(x)) > t with x = v1 var2 + 1e-8 (dense_stereo.py:344-345, :303),
// decided without the square root and the quotient wherever that is safe.  For t >= 2^-10:
//     ncc > t  <=>  cov > 0  and  cov^2 > t^2 x        (x > 0, real arithmetic)
// The evaluated ncc carries two roundings (square root, correctly rounded quotient: relative 1.2e-7), the
// evaluated cov^2 and t^2 x three (1.8e-7); a margin of 1e-6 on t^2 x separates the sure cases:
//     sure_yes:  cov > 0, 1e-12 <= x < 1e30, 1e30 > cov^2 > t^2 x (1 + 1e-6)   (the upper bound on cov^2 keeps the
//                quotient's intermediate products finite: qdiv of an overflowing cov / den is NaN, not a vote)
//     sure_no :  not (cov > 0): the quotient is <= 0, -inf or NaN (a NaN cov included): never > t > 0
//                or 1e-12 <= x < 1e30, cov^2 < t^2 x (1 - 1e-6)
// (the bounds on x keep t^2 x a normal, finite number for 2^-10 <= t < 2^10; a cov^2 that underflows is far
//  below any such t^2 x; a NaN x fails both bounds and, for cov > 0, stays undecided.)  Everything else -- a measure-zero band around the
// threshold, negative or tiny x, non-finite values -- is undecided and makes the wave evaluate the vote itself.
struct VoteGate { float t2_hi, t2_lo; bool usable; };

AMVS_DEV VoteGate make_vote_gate(float thresh)
{
    VoteGate g;
    const float t2 = thresh * thresh;
    g.t2_hi = t2 * 1.000001f;
    g.t2_lo = t2 * 0.999999f;
    g.usable = thresh >= 0x1p-10f && thresh < 0x1p10f;
    return g;
}

template <int K, int S, bool U8>
__global__ __launch_bounds__(AMVS_WAVE) void plane_sweep_kernel(const SweepArgs a)
{
    constexpr int HALF = K / 2;
    constexpr int OUTW = AMVS_WAVE - 2 * HALF;
    constexpr float INV_AREA = 1.0f / (float)(K * K);
    __shared__ uint16_t best[AMVS_SWEEP_MAX_TH][AMVS_WAVE];
    __shared__ float lut[256];
#ifdef AMVS_HSUM_LDS
    __shared__ float4 hbuf[(AMVS_WAVE + K - 1) * HSum<S>::NV4];
#else
    float4 *hbuf = nullptr;
#endif
    __shared__ float lring[(Ring<K, S>::NL + 1) * K * AMVS_WAVE];

    const int lane = threadIdx.x;
    window_sums_init<K, S>(hbuf, lane);
    if (U8) fill_gray_lut(lut, lane);
    const int t0 = xcd_remap(blockIdx.x, gridDim.x);
    const int cid = t0 % a.n_chunks;          // plane chunk of this wave
    const int t = t0 / a.n_chunks;
    const int tiles_per_job = a.tiles_x * a.tiles_y;
    const int job_id = t / tiles_per_job;
    const int rem = t - job_id * tiles_per_job;
    const int ty = rem / a.tiles_x;
    const int tx = rem - ty * a.tiles_x;
    const int d_begin = cid * a.chunk, d_end = min(a.D, d_begin + a.chunk);

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    const float *__restrict__ ref = a.images + job->ref_img * a.img_stride;
    const float *__restrict__ ref_mean = a.ref_mean + job->ref_img * a.img_stride;
    const float *__restrict__ ref_var = a.ref_var + job->ref_img * a.img_stride;
    const GlobalU16 ref_pairs = U8 ? (GlobalU16)job->ref_pairs : (GlobalU16)a.pairs;   // global, not FLAT, loads
    constexpr int PADW = U8 ? 2 * AMVS_PAIR_BORDER : 0;
    const SampleConsts sc = make_sample_consts(H, W, -__builtin_inff(), __builtin_inff(), __builtin_inff());
    const VoteGate gate = make_vote_gate(a.thresh);

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int trows = min(a.TH, H - y0);
    const int rows = trows + 2 * HALF;
    RayCols rc;
    {
        const float px = (float)xr;
        rc.c0 = px * job->Kinv[0]; rc.c1 = px * job->Kinv[3]; rc.c2 = px * job->Kinv[6];
    }

    // running best per output pixel of the strip: 16-bit keys (votes << 12 | 4095 - plane of the chunk), or -- a.key8,
    // chunks of at most 32 planes -- 8-bit keys (votes << 5 | 31 - plane) in the same array, for strips twice as high
    uint8_t *best8 = (uint8_t *)&best[0][0];
    if (a.key8) { for (int i = 0; i < trows; ++i) best8[i * AMVS_WAVE + lane] = (uint8_t)0; }
    else { for (int i = 0; i < trows; ++i) best[i][lane] = (uint16_t)0; }

    for (int d = d_begin; d < d_end; ++d) {
        const float depth = a.depths[AMVS_IDX(d, a.D)];
        float ring_r[K];
        float ring_v[Ring<K, S>::NR][K];
        typename Hist<K, S>::T hist_ok = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            ring_r[i] = 0.0f;
#pragma unroll
            for (int s = 0; s < Ring<K, S>::NR; ++s) ring_v[s][i] = 0.0f;
        }
        int wslot = 0;

        for (int r = 0; r < rows; ++r) {
            const int yr = y0 - HALF + r;
            const bool live = col_in & ((unsigned)yr < (unsigned)H);
            const int pix = yr * W + xr;
            // ref gray: in the packed path the low byte of the row-pair map decoded through the table (the
            // same float as the float32 map holds, at half the bytes)
            const float rvl = U8 ? lut[ref_pairs[AMVS_IDX_LOHI(live ? pix + PADW * yr : 0, -((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER), (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER) - ((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER))] & 0xFFu]
                                 : ref[AMVS_IDX(live ? pix : 0, HW)];
            const float rv = live ? rvl : 0.0f;
            JobCP jr = reload(job);
            const Vec3 Pw = backproject_cols(jr->Kinv, jr->Rref, jr->tref, rc, yr, depth);
            float v[S];
            bool ok = true;
            unsigned okbits = sweep_sample_sources<S, U8, true>(jr, sc, lut, Pw, live, v, ok);
            if (__builtin_expect(!__all(ok), 0)) okbits = sweep_sample_sources<S, U8, false>(reload(job), sc, lut, Pw, live, v, ok);
            ring_push<K, S>(lring, lane, wslot, ring_r, ring_v, rv, v);
            wslot = wslot + 1 == K ? 0 : wslot + 1;
            hist_ok = (hist_ok >> S) | ((typename Hist<K, S>::T)okbits << (S * HALF));
            if (r < 2 * HALF) continue;

            const int yc = yr - HALF;
            const int xc = xr + HALF;
            const bool outl = (lane < OUTW) & (xc < W);
            const int pc = AMVS_IDX(outl ? yc * W + xc : 0, HW);
            // mean1 / var1 of the reference window (precomputed: box_stats_kernel forms the sums this kernel
            // used to form per plane, in the same order)
            const float m1 = ref_mean[pc], v1 = ref_var[pc];
            const unsigned okc = (unsigned)__shfl_down((int)(unsigned)hist_ok, HALF);
            float bvs[S], bvvs[S], brvs[S], br_unused, brr_unused;
            window_sums<K, S, false, false>(lring, wslot, ring_r, ring_v, hbuf, lane, bvs, bvvs, brvs, br_unused, brr_unused);
            float covs[S], xs[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                // _compute_ncc_torch (dense_stereo.py:333-345)
                const float mean2 = bvs[s] * INV_AREA;
                const float var2 = bvvs[s] * INV_AREA - mean2 * mean2;
                covs[s] = brvs[s] * INV_AREA - m1 * mean2;
                xs[s] = v1 * var2 + 1e-8f;
            }
            // (bit s of yes_bits / open_bits: source s votes for sure / is undecided; an invalid projection casts no
            //  vote whatever the NCC, so both are masked with the validity bits at the end)
            uint32_t yes_bits = 0u, open_bits = 0u;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float c2 = covs[s] * covs[s];
                const bool pos = covs[s] > 0.0f, xok = (xs[s] >= 1e-12f) & (xs[s] < 1e30f);
                const bool yes = pos & xok & (c2 > gate.t2_hi * xs[s]) & (c2 < 1e30f);
                const bool no = !pos | (xok & (c2 < gate.t2_lo * xs[s]));
                yes_bits |= yes ? (1u << s) : 0u;
                open_bits |= (yes | no) ? 0u : (1u << s);
            }
            uint32_t votes = (uint32_t)__builtin_popcount(yes_bits & okc & ((1u << S) - 1u));
            const bool decided = gate.usable & ((open_bits & okc) == 0u);
            if (__builtin_expect(!__all(decided | !outl), 0)) {
                // some lane is inside the margin (or the threshold is outside the gate's range): the vote as
                // the reference forms it, for the whole wave
                auto vote_stage = [&](auto lean, bool &vok) {
                    constexpr bool LEAN = decltype(lean)::value;
                    votes = 0u;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float den = sqrt_t<LEAN>(xs[s], vok);
                        const float ncc = qdiv(covs[s], den, rcp_t<LEAN>(den, vok));
                        if (ncc > a.thresh && ((okc >> s) & 1u)) votes += 1u;   // :303-304
                    }
                };
                bool vok = true;
                vote_stage(std::true_type{}, vok);
                if (__builtin_expect(!__all(vok), 0)) vote_stage(std::false_type{}, vok);
            }
            if (outl) {
                // the chunk's first plane always enters (torch.max over a volume that starts at 0
                // votes): its key (0 << 12) | 4095 -- (0 << 5) | 31 -- beats the initial 0
                if (a.key8) {
                    const uint32_t keyv = (votes << 5) | (uint32_t)(AMVS_SWEEP_MAX_CHUNK8 - 1 - (d - d_begin));
                    const uint32_t cur = best8[(yc - y0) * AMVS_WAVE + lane];
                    if (keyv > cur) best8[(yc - y0) * AMVS_WAVE + lane] = (uint8_t)keyv;
                } else {
                    const uint32_t keyv = (votes << 12) | (uint32_t)(AMVS_SWEEP_MAX_CHUNK - 1 - (d - d_begin));
                    const uint32_t cur = best[yc - y0][lane];
                    if (keyv > cur) best[yc - y0][lane] = (uint16_t)keyv;
                }
            }
        }
    }

    unsigned *__restrict__ keys = a.keys + job->slot * HW;
    const int xc = xr + HALF;
    if (lane < OUTW && xc < W)
        for (int i = 0; i < trows; ++i) {
            const uint32_t b = a.key8 ? (uint32_t)best8[i * AMVS_WAVE + lane] : (uint32_t)best[i][lane];
            const uint32_t votes = a.key8 ? b >> 5 : b >> 12;
            const uint32_t plane = (uint32_t)d_begin + (a.key8 ? AMVS_SWEEP_MAX_CHUNK8 - 1 - (b & (AMVS_SWEEP_MAX_CHUNK8 - 1))
                                                                 : AMVS_SWEEP_MAX_CHUNK - 1 - (b & (AMVS_SWEEP_MAX_CHUNK - 1)));
            atomicMax(&keys[AMVS_IDX((y0 + i) * W + xc, HW)], (votes << 16) | (65535u - plane));
        }
}

// decode the merged keys: depth of the winning plane (dense_stereo.py:310) and its vote count
__global__ __launch_bounds__(256) void plane_sweep_finish_kernel(const unsigned *__restrict__ keys,
                                                                 const float *__restrict__ depths, int n_planes, long long n,
                                                                 float *__restrict__ depth_out,
                                                                 float *__restrict__ conf_out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const unsigned b = keys[i];
        depth_out[i] = depths[AMVS_IDX(65535 - (int)(b & 0xFFFFu), n_planes)];
        conf_out[i] = (float)(b >> 16);
    }
}

// ------------------------------------------------------------------ dispatch -----
template <int K, int S>
static hipError_t launch_sweep_ks(const SweepArgs &a, int nblk, hipStream_t st)
{
    if (!a.ref_mean || !a.ref_var) return hipErrorInvalidValue;
    if (a.pairs)
        hipLaunchKernelGGL((plane_sweep_kernel<K, S, true>), dim3(nblk), dim3(AMVS_WAVE), 0, st, a);
    else
        hipLaunchKernelGGL((plane_sweep_kernel<K, S, false>), dim3(nblk), dim3(AMVS_WAVE), 0, st, a);
    return hipGetLastError();
}

#define AMVS_FOR_S(K, FN, ...)                                      \
    switch (S) {                                                    \
    case 2: return FN<K, 2>(__VA_ARGS__);                           \
    case 3: return FN<K, 3>(__VA_ARGS__);                           \
    case 4: return FN<K, 4>(__VA_ARGS__);                           \
    case 5: return FN<K, 5>(__VA_ARGS__);                           \
    case 6: return FN<K, 6>(__VA_ARGS__);                           \
    default: return decltype(FN<K, 2>(__VA_ARGS__))(1);             \
    }

hipError_t launch_sweep(int K, int S, const SweepArgs &a, hipStream_t st)
{
    if (!patch_compiled(K)) return launch_sweep_generic(K, S, a, st);
    if (a.fast) return launch_sweep_fast(K, S, a, st);
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y * a.n_chunks;
    switch (K) {
    case 3: AMVS_FOR_S(3, launch_sweep_ks, a, nblk, st)
    case 5: AMVS_FOR_S(5, launch_sweep_ks, a, nblk, st)
    case 7: AMVS_FOR_S(7, launch_sweep_ks, a, nblk, st)
    case 9: AMVS_FOR_S(9, launch_sweep_ks, a, nblk, st)
    case 11: AMVS_FOR_S(11, launch_sweep_ks, a, nblk, st)
    case 13: AMVS_FOR_S(13, launch_sweep_ks, a, nblk, st)
    case 15: AMVS_FOR_S(15, launch_sweep_ks, a, nblk, st)
    case 17: AMVS_FOR_S(17, launch_sweep_ks, a, nblk, st)
    case 19: AMVS_FOR_S(19, launch_sweep_ks, a, nblk, st)
    case 21: AMVS_FOR_S(21, launch_sweep_ks, a, nblk, st)
    case 23: AMVS_FOR_S(23, launch_sweep_ks, a, nblk, st)
    case 25: AMVS_FOR_S(25, launch_sweep_ks, a, nblk, st)
    case 27: AMVS_FOR_S(27, launch_sweep_ks, a, nblk, st)
    case 29: AMVS_FOR_S(29, launch_sweep_ks, a, nblk, st)
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sweep_finish(const SweepArgs &a, hipStream_t st)
{
    const long long n = (long long)a.n_jobs * a.H * a.W;
    const int bx = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(plane_sweep_finish_kernel, dim3(bx), dim3(256), 0, st, a.keys, a.depths, a.D, n, a.depth_out,
                       a.conf_out);
    return hipGetLastError();
}

}  // namespace amvs

AMVS_CHECK_TU(sweep_exact)
