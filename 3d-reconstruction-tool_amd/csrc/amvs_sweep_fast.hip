// amvs_sweep_fast.hip -- the plane sweep (_plane_sweep_torch, src/core/dense_stereo.py:262-310) in the FAST
// arithmetic.  Its own translation unit because it is built with another instruction scheduler than the
// sweep step: the kernel is VALU-issue-bound at three to four waves per SIMD, and LLVM's occupancy-driven
// iterative scheduler (-mllvm -amdgpu-sched-strategy=iterative-maxocc, csrc/Makefile) measured +1.7 % on it
// (69.3-69.7 against 67.9-68.7 G px-hyp/s, four alternations in one run), -0.5 % on the sweep step.  The
// instruction order changes, the operations do not (-ffp-contract=off, explicit fmaf): same bits.
#define AMVS_TU_ID 3
#include "amvs_fast_common.h"

namespace amvs {

// ------------------------------------------------------------------ plane sweep --
// _plane_sweep_torch (dense_stereo.py:262-310) in the fast arithmetic; structure (strips, plane
// chunks, 16-bit running-best keys in LDS, atomicMax merge) as plane_sweep_kernel.
template <int K, int S>
__global__ __launch_bounds__(AMVS_WAVE) void plane_sweep_fast_kernel(const SweepArgs a)
{
    constexpr int HALF = K / 2;
    constexpr int OUTW = AMVS_WAVE - 2 * HALF;
    constexpr float C1 = (float)(1.0 / ((double)(K * K) * 255.0));
    constexpr float C2 = (float)(1.0 / ((double)(K * K) * 65025.0));
    constexpr int NL = FRing<K, S>::NL;
    __shared__ uint16_t best[AMVS_SWEEP_MAX_TH][AMVS_WAVE];
    __shared__ float lring[(NL > 0 ? NL : 1) * K * AMVS_WAVE];

    const int lane = threadIdx.x;
    const int t0 = xcd_remap(blockIdx.x, gridDim.x);
    const int cid = t0 % a.n_chunks;
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
    constexpr int PADW = 2 * AMVS_PAIR_BORDER;
    // (global address space: a generic pointer would make these FLAT loads, which force vmcnt(0) and
    // lgkmcnt(0) waits)
    const GlobalU16 ref_pairs = (GlobalU16)job->ref_pairs;
    const GlobalFloat2s ref_stats = (GlobalFloat2s)job->ref_stats;
    const FastConsts fc = make_fast_consts(H, W, 0);

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    const float fx = (float)xr;
    FastCol cols[S];
    fast_columns<S>(job, fx, cols);
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int trows = min(a.TH, H - y0);
    const int rows = trows + 2 * HALF;

    // running best per output pixel of the strip: 16-bit keys (votes << 12 | 4095 - plane of the chunk), or -- a.key8,
    // chunks of at most 32 planes -- 8-bit keys (votes << 5 | 31 - plane) in the same array, for strips twice as high
    uint8_t *best8 = (uint8_t *)&best[0][0];
    if (a.key8) { for (int i = 0; i < trows; ++i) best8[i * AMVS_WAVE + lane] = (uint8_t)0; }
    else { for (int i = 0; i < trows; ++i) best[i][lane] = (uint16_t)0; }

    // The lean reciprocal (v_rcp_f32 + one FMA) equals 1.0f / z wherever 2^-95 <= |z| < 2^96
    // (amvs_device.h).  For a plane, z = fma(depth, fma(M7, y, t2), b2) + 1e-8 is monotonic along a
    // lane's column, so the test is made ONCE per strip and plane at the strip's first and last row
    // (same sign at both ends: no zero crossing inside) instead of in every row; a strip that fails runs
    // its rows with the IEEE quotient -- the same values either way (measured +2 %: 61.7 against 60.5
    // G px-hyp/s in one run).
    const float fy_first = (float)(y0 - HALF), fy_last = (float)(y0 - HALF + rows - 1);

    for (int d = d_begin; d < d_end; ++d) {
        const float depth = a.depths[AMVS_IDX(d, a.D)];
        bool lean_ok = true;
        {
            JobCP jr = reload(job);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float m7 = jr->fsrc[s].M[7], b2 = jr->fsrc[s].b[2];
                const float z0 = __builtin_fmaf(depth, __builtin_fmaf(m7, fy_first, cols[s].t2), b2) + 1e-8f;
                const float z1 = __builtin_fmaf(depth, __builtin_fmaf(m7, fy_last, cols[s].t2), b2) + 1e-8f;
                const float a0 = __builtin_fabsf(z0), a1 = __builtin_fabsf(z1);
                lean_ok &= (a0 >= 0x1p-95f) & (a0 < 0x1p96f) & (a1 >= 0x1p-95f) & (a1 < 0x1p96f) & ((z0 > 0.0f) == (z1 > 0.0f));
            }
        }
        const bool lean_strip = __all(lean_ok);
        uint32_t rb[RefBytes<K>::NB];
        float ring_v[FRing<K, S>::NR][K];
        typename Hist<K, S>::T hist_ok = 0;
#pragma unroll
        for (int i = 0; i < RefBytes<K>::NB; ++i) rb[i] = 0u;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int s = 0; s < FRing<K, S>::NR; ++s) ring_v[s][i] = 0.0f;
        int wslot = 0;

        for (int r = 0; r < rows; ++r) {
            const int yr = y0 - HALF + r;
            const bool live = col_in & ((unsigned)yr < (unsigned)H);
            const int pix = yr * W + xr;
            const uint32_t rc_raw = ref_pairs[AMVS_IDX_LOHI(live ? pix + PADW * yr : 0, -((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER), (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER) - ((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER))];
            const uint32_t rcode = live ? (rc_raw & 0xFFu) : 0u;
            float v[S];
            bool unused_ok = true;
            const unsigned okbits = lean_strip
                ? fast_sample_sources<S, true, false, false>(job, fc, cols, (float)yr, depth, live, v, unused_ok)
                : fast_sample_sources<S, false, false, false>(reload(job), fc, cols, (float)yr, depth, live, v, unused_ok);
            ref_bytes_push<K>(rb, rcode);
            fring_push<K, S>(lring, lane, wslot, ring_v, v);
            wslot = wslot + 1 == K ? 0 : wslot + 1;
            hist_ok = (hist_ok >> S) | ((typename Hist<K, S>::T)okbits << (S * HALF));
            if (r < 2 * HALF) continue;

            const int yc = yr - HALF;
            const int xc = xr + HALF;
            const bool outl = (lane < OUTW) & (xc < W);
            const f32x2_t mv1 = ref_stats[AMVS_IDX(outl ? yc * W + xc : 0, HW)];
            const unsigned okc = (unsigned)__shfl_down((int)(unsigned)hist_ok, HALF);
            float rr[K];
#pragma unroll
            for (int i = 0; i < K; ++i) rr[i] = ref_bytes_get<K>(rb, i);
            float bvs[S], bvvs[S], brvs[S];
            window_sums_fast<K, S>(lring, wslot, rr, ring_v, lane, bvs, bvvs, brvs);
            const float m1 = mv1.x, v1 = mv1.y;
            uint32_t votes = 0u;
            if (a.thresh > 0.0f) {
                // ncc > thresh (dense_stereo.py:303) without square root and division:
                // cov / sqrt(x) > t  <=>  cov > 0, x >= 0 (a negative x is the reference's NaN) and cov^2 > t^2 x
                const float t2 = a.thresh * a.thresh;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float mean2 = bvs[s] * C1;
                    const float var2 = __builtin_fmaf(-mean2, mean2, bvvs[s] * C2);
                    const float cov = __builtin_fmaf(-m1, mean2, brvs[s] * C2);
                    const float x = v1 * var2 + 1e-8f;
                    const bool vote = (cov > 0.0f) & (x >= 0.0f) & (cov * cov > t2 * x) & (((okc >> s) & 1u) != 0u);
                    votes += vote ? 1u : 0u;
                }
            } else {
                auto vote_stage = [&](auto lean, bool &ok) {
                    constexpr bool LEAN = decltype(lean)::value;
                    votes = 0u;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        // _compute_ncc_torch: eps inside the sqrt (dense_stereo.py:344-345)
                        const float mean2 = bvs[s] * C1;
                        const float var2 = __builtin_fmaf(-mean2, mean2, bvvs[s] * C2);
                        const float cov = __builtin_fmaf(-m1, mean2, brvs[s] * C2);
                        const float den = sqrt_t<LEAN>(v1 * var2 + 1e-8f, ok);
                        const float ncc = cov * rcp_t<LEAN>(den, ok);
                        if (ncc > a.thresh && ((okc >> s) & 1u)) votes += 1u;   // :303-304
                    }
                };
                bool ok = true;
                vote_stage(std::true_type{}, ok);
                if (__builtin_expect(!__all(ok), 0)) vote_stage(std::false_type{}, ok);
            }
            if (outl) {
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

// ------------------------------------------------------------------ dispatch -----
template <int K, int S>
static hipError_t launch_sweep_fast_ks(const SweepArgs &a, int nblk, hipStream_t st)
{
    hipLaunchKernelGGL((plane_sweep_fast_kernel<K, S>), dim3(nblk), dim3(AMVS_WAVE), 0, st, a);
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

hipError_t launch_sweep_fast(int K, int S, const SweepArgs &a, hipStream_t st)
{
    if (!a.pairs) return hipErrorInvalidValue;
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y * a.n_chunks;
    switch (K) {
    case 3: AMVS_FOR_S(3, launch_sweep_fast_ks, a, nblk, st)
    case 5: AMVS_FOR_S(5, launch_sweep_fast_ks, a, nblk, st)
    case 7: AMVS_FOR_S(7, launch_sweep_fast_ks, a, nblk, st)
    case 9: AMVS_FOR_S(9, launch_sweep_fast_ks, a, nblk, st)
    case 11: AMVS_FOR_S(11, launch_sweep_fast_ks, a, nblk, st)
    case 13: AMVS_FOR_S(13, launch_sweep_fast_ks, a, nblk, st)
    case 15: AMVS_FOR_S(15, launch_sweep_fast_ks, a, nblk, st)
    case 17: AMVS_FOR_S(17, launch_sweep_fast_ks, a, nblk, st)
    case 19: AMVS_FOR_S(19, launch_sweep_fast_ks, a, nblk, st)
    case 21: AMVS_FOR_S(21, launch_sweep_fast_ks, a, nblk, st)
    case 23: AMVS_FOR_S(23, launch_sweep_fast_ks, a, nblk, st)
    case 25: AMVS_FOR_S(25, launch_sweep_fast_ks, a, nblk, st)
    case 27: AMVS_FOR_S(27, launch_sweep_fast_ks, a, nblk, st)
    case 29: AMVS_FOR_S(29, launch_sweep_fast_ks, a, nblk, st)
    default: return hipErrorInvalidValue;
    }
}

}  // namespace amvs

AMVS_CHECK_TU(sweep_fast)
