// NOT BUILT -- kept as evidence for DESIGN.md section 5.  Two pixels per lane (interleaved columns,
// 128-column strips, halo 1.049 instead of 1.103, half the DPP shifts): bit-identical to the one-pixel
// kernel and the CPU checker (10 parity cases incl. 16 x 1080p), 7.5 % fewer VALU, 25 % fewer SALU and
// 55 % fewer LDS instructions -- and 22-25 % SLOWER in every iteration (1.23 vs 1.02 ms in the first,
// 0.93 vs 0.73 in the last; 8, 12 or 16 waves per CU alike).  The sampling stage alone explains it: 64
// ADJACENT pixels of a row share their epipolar band, so one gather instruction touches ~30 lines, not
// 64; lanes two pixels apart spread an instruction over a 128-pixel span (sampling-only kernel: 0.465 ms
// one pixel per lane, 0.616 interleaved, 0.507 with lane i on columns i and 64+i).  Needs the shared
// helpers of amvs_kernels_fast.hip moved into a header (amvs_fast_common.h) to build.
// amvs_kernels_fast2.hip -- the fast-arithmetic sweep step with TWO pixels per lane.
//
// Same step, same arithmetic and the same order of every sum as pm_step_fast_kernel
// (amvs_kernels_fast.hip; reference: src/core/mvs_patchmatch.py:323-491) -- the results are
// bit-identical -- in a different execution shape:
//   * a wave owns a strip of 128 columns, lane i the columns 2i and 2i+1 (interleaved).  The k x k
//     window sums nest right to left, so with E[i] / O[i] the column sums of a lane's even / odd
//     column the sum of the window that starts at the even column 2i is
//         E[i] + (O[i] + (E[i+1] + (O[i+1] + ... )))
//     i.e. a chain that alternates a plain add with an add whose operand is shifted by one lane
//     (v_add_f32_dpp wave_shl:1): k-1 adds per window as before, only (k-1)/2 of them DPP, and
//     nothing is exchanged between waves;
//   * a strip yields 128 - 2(k/2) output columns (122 of 128 for k = 7 where a 64-column strip
//     yields 58 of 64): the horizontal halo falls from 1.103 to 1.049, and with it the number of
//     gathered 128-byte lines, which is what bounds the launch;
//   * the per-row scalar work (job-table loads, loop control), the LDS ring instructions (one
//     ds_write_b64 / ds_read_b64 per pair of pixels) and the DPP shifts are paid once per two pixels;
//   * half as many waves hold the same number of pixels in flight: two 4-wave workgroups per CU.
#include "amvs_fast_common.h"

namespace amvs {

#ifndef AMVS_FAST2_MAX_WGS_PER_CU
#define AMVS_FAST2_MAX_WGS_PER_CU 2
#endif

// sources whose vertical ring lives in LDS (float2 per lane and row); a workgroup's static LDS must
// stay under 64 KiB
#ifndef AMVS_FAST2_RING_LDS
#define AMVS_FAST2_RING_LDS 3
#endif
template <int K, int S> struct F2Ring {
    static constexpr int WANT = K >= 11 && AMVS_FAST2_RING_LDS > 2 ? 2 : AMVS_FAST2_RING_LDS;
    static constexpr int NL = WANT < S ? WANT : S;
    static constexpr int NR = S - NL > 0 ? S - NL : 1;
};

template <int K, int S> struct Step2Lds {
    static constexpr unsigned STATIC = AMVS_WG_WAVES * (F2Ring<K, S>::NL * K * AMVS_WAVE * 8u + 2u * AMVS_WAVE * 8u);
    // a workgroup may hold at most 64 KiB (static + dynamic) without a function attribute; 64 KiB
    // each already means two workgroups per CU
    static constexpr unsigned WANT = 160u * 1024u / AMVS_FAST2_MAX_WGS_PER_CU;
    static constexpr unsigned SHARE = WANT < 64u * 1024u ? WANT : 64u * 1024u;
    static constexpr unsigned EXTRA = STATIC < SHARE ? SHARE - STATIC : 0u;
};

// lane i <- lane i+N of a 32-bit value (lanes past the end receive 0)
template <int N>
AMVS_DEV uint32_t wave_shl_u32(uint32_t x)
{
#pragma unroll
    for (int i = 0; i < N; ++i) x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, true);
    return x;
}

// both pixels of a lane through the S sources: the pose rows are loaded once per source
template <int S, bool LEAN>
AMVS_DEV void fast2_sample(JobCP job, const FastConsts &fc, const float (&fx)[2], float fy, const float (&d)[2],
                           const bool (&live)[2], float (&v)[2][S], unsigned (&okb)[2], bool &ok)
{
    FastTap tg[2][S];
    uint32_t raw[2][S];
    float zlo = 1.0f, zhi = 1.0f;
    okb[0] = okb[1] = 0u;
    JobCP jr = job;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s % AMVS_FAST_RELOAD_STRIDE == 0) jr = reload(jr);
        float M[9], b[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) M[i] = jr->fsrc[s].M[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = jr->fsrc[s].b[i];
        const unsigned long long img = jr->fsrc[s].pairs;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            bool valid;
            tg[p][s] = fast_geom<LEAN, true>(M, b, fc, fx[p], fy, d[p], valid, zlo, zhi);
            okb[p] |= valid ? (1u << s) : 0u;
            raw[p][s] = fast_load(img, tg[p][s].off);
        }
    }
    if constexpr (LEAN) ok = (zlo >= 0x1p-95f) & (zhi < 0x1p96f);
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) v[p][s] = fast_finish(raw[p][s], tg[p][s], live[p]);
}

template <int K, int S, int MODE_T>
__global__ __launch_bounds__(AMVS_WAVE * AMVS_WG_WAVES, 2) void pm_step_fast2_kernel(const StepArgs a)
{
    constexpr int HALF = K / 2;
    constexpr int OUTL = AMVS_WAVE - HALF;                    // lanes that own two output columns
    constexpr int OUTW = 2 * OUTL;                            // output columns per strip
    constexpr float C1 = (float)(1.0 / ((double)(K * K) * 255.0));      // 1 / (k^2 * 255)
    constexpr float C2 = (float)(1.0 / ((double)(K * K) * 65025.0));    // 1 / (k^2 * 255^2)
    constexpr int NL = F2Ring<K, S>::NL, NR = F2Ring<K, S>::NR;
    __shared__ float2 lring_all[AMVS_WG_WAVES * NL * K * AMVS_WAVE];
    constexpr int NQ = 2 * AMVS_WAVE;
    __shared__ uint2 nq_all[AMVS_WG_WAVES * NQ];

    const int lane = threadIdx.x & (AMVS_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / AMVS_WAVE);
    float2 *lring = lring_all + wv * (NL * K * AMVS_WAVE);
    uint2 *nq = nq_all + wv * NQ;
    int q_head = 0, q_tail = 0;
    const int tiles_per_job = a.tiles_x * a.tiles_y;
    const int t = xcd_remap(blockIdx.x, gridDim.x) * AMVS_WG_WAVES + wv;
    if (t >= a.n_jobs * tiles_per_job) return;                 // last workgroup only
    int job_id, ty, tx;
    strip_of(a, t, job_id, ty, tx);

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W;
    constexpr int mode = MODE_T;
    static_assert(MODE_T == MODE_REFINE || MODE_T == MODE_PROP, "the two-pixel kernel serves the schedule's hot steps");
    const long long HW = (long long)H * W;
    constexpr int PADW = 2 * AMVS_PAIR_BORDER;
    const GlobalU16 ref_pairs = (GlobalU16)job->ref_pairs;
    const GlobalFloat2s ref_stats = (GlobalFloat2s)job->ref_stats;
    const float *__restrict__ d_in = a.d_in + job->slot * HW;
    const float *__restrict__ n_in = a.n_in + job->slot * HW * 3;
    float *__restrict__ d_out = a.d_out + job->slot * HW;
    float *cost_io = a.cost + job->slot * HW;
    float *n_out = a.n_out + job->slot * HW * 3;

    const StreamKey key = stream_key(a.seed, job->stream_view, a.draw);
    const FastConsts fc = make_fast_consts(H, W, HALF);        // patch bounds (mvs_patchmatch.py:362-363)

    const int xout = tx * OUTW + 2 * lane;                     // first of the lane's two output columns
    const int xr0 = xout - HALF;                               // first of the lane's two sampled columns
    const int y0 = ty * a.TH;
    const int rows = min(a.TH, H - y0) + 2 * HALF;
    const float fx[2] = {(float)xr0, (float)(xr0 + 1)};
    const bool col_in[2] = {(unsigned)xr0 < (unsigned)W, (unsigned)(xr0 + 1) < (unsigned)W};

    uint32_t rb[2][RefBytes<K>::NB];
    float ring_v[2][NR][K];
    typename Hist<K, S>::T hist_ok[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < RefBytes<K>::NB; ++i) rb[p][i] = 0u;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int s = 0; s < NR; ++s) ring_v[p][s][i] = 0.0f;
    }
    int wslot = 0;

    const int oy = mode == MODE_PROP ? a.oy : 0, ox = mode == MODE_PROP ? a.ox : 0;
    const int noff = oy * W + ox;

    for (int r = 0; r < rows; ++r) {
#if AMVS_WG_WAVES > 1 && AMVS_WG_SYNC_ROWS > 0
        if (r % AMVS_WG_SYNC_ROWS == 0) __builtin_amdgcn_s_barrier();
#endif
        const int yr = y0 - HALF + r;
        const bool row_in = (unsigned)yr < (unsigned)H;
        bool live[2];
        float dc[2];
        uint32_t rcode[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int xr = xr0 + p;
            live[p] = col_in[p] & row_in;
            const bool inb = live[p] & ((unsigned)(yr + oy) < (unsigned)H) & ((unsigned)(xr + ox) < (unsigned)W);
            const int pix = yr * W + xr;
            const float d_raw = d_in[inb ? pix + noff : 0];
            const uint32_t rc_raw = ref_pairs[live[p] ? pix + PADW * yr : 0];
            const uint32_t h0 = pixel_hash((uint32_t)pix, key);
            dc[p] = candidate_depth(a, mode, inb, d_raw, h0);
            rcode[p] = live[p] ? (rc_raw & 0xFFu) : 0u;
        }

        float v[2][S];
        unsigned okb[2];
        {
            bool ok = true;
            fast2_sample<S, true>(job, fc, fx, (float)yr, dc, live, v, okb, ok);
            if (__builtin_expect(!__all(ok), 0)) fast2_sample<S, false>(reload(job), fc, fx, (float)yr, dc, live, v, okb, ok);
        }

        // ---- push into the vertical rings ----
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ref_bytes_push<K>(rb[p], rcode[p]);
            hist_ok[p] = (hist_ok[p] >> S) | ((typename Hist<K, S>::T)okb[p] << (S * HALF));
#pragma unroll
            for (int s = NL; s < S; ++s) {
#pragma unroll
                for (int i = 0; i < K - 1; ++i) ring_v[p][s - NL][i] = ring_v[p][s - NL][i + 1];
                ring_v[p][s - NL][K - 1] = v[p][s];
            }
        }
#pragma unroll
        for (int s = 0; s < NL; ++s) lring[(s * K + wslot) * AMVS_WAVE + lane] = make_float2(v[0][s], v[1][s]);
        wslot = wslot + 1 == K ? 0 : wslot + 1;

        if (r < 2 * HALF) continue;

        // ---- column sums of both pixels, top -> bottom (order of window_sums_fast) ----
        float rr[2][K];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < K; ++i) rr[p][i] = ref_bytes_get<K>(rb[p], i);
        int slot[K];
#pragma unroll
        for (int i = 0; i < K; ++i) slot[i] = wslot + i >= K ? wslot + i - K : wslot + i;
        float cs[2][3 * S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float vv[2][K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (s < NL) {
                    const float2 w = lring[((s < NL ? s : 0) * K + slot[i]) * AMVS_WAVE + lane];
                    vv[0][i] = w.x; vv[1][i] = w.y;
                } else {
                    vv[0][i] = ring_v[0][s < NL ? 0 : s - NL][i];
                    vv[1][i] = ring_v[1][s < NL ? 0 : s - NL][i];
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float cv = vv[p][0];
                float cvv = vv[p][0] * vv[p][0];
                float crv = rr[p][0] * vv[p][0];
#pragma unroll
                for (int i = 1; i < K; ++i) {
                    cv = cv + vv[p][i];
                    cvv = __builtin_fmaf(vv[p][i], vv[p][i], cvv);
                    crv = __builtin_fmaf(rr[p][i], vv[p][i], crv);
                }
                cs[p][3 * s] = cv; cs[p][3 * s + 1] = cvv; cs[p][3 * s + 2] = crv;
            }
        }
        // ---- row sums right -> left over the interleaved columns (see the file header) ----
        float win[2][3 * S];       // [0]: window starting at the lane's even column, [1]: at its odd column
#pragma unroll
        for (int i = 0; i < 3 * S; ++i) {
            float g = cs[0][i], h = cs[1][i];
#pragma unroll
            for (int m = 1; m < K; ++m) {
                if (m & 1) {
                    g = wave_shl1(g) + cs[1][i];
                    h = h + cs[0][i];
                } else {
                    g = g + cs[0][i];
                    h = wave_shl1(h) + cs[1][i];
                }
            }
            win[0][i] = g; win[1][i] = h;
        }

        // ---- NCC, aggregate, select for the two output pixels (yc, xout + q) ----
        const int yc = yr - HALF;
        // validity bits of the window centres: the centre of output q is sampled column
        // 2 lane + q + HALF of the strip, i.e. pixel (q + HALF) & 1 of lane + (q + HALF) / 2
        unsigned okc[2];
        okc[0] = wave_shl_u32<(HALF) / 2>((uint32_t)hist_ok[HALF & 1]);
        okc[1] = wave_shl_u32<(HALF + 1) / 2>((uint32_t)hist_ok[(HALF + 1) & 1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int xc = xout + q;
            const bool outl = (lane < OUTL) & (xc < W);
            const int pc = outl ? yc * W + xc : 0;
            const float oldd = d_in[pc], oldc = cost_io[pc];
            const f32x2_t mv1 = ref_stats[pc];
            const float m1 = mv1.x, v1 = mv1.y;
            float total = 0.0f, cnt = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                // _ncc_cost (mvs_patchmatch.py:403-411), sums in code units
                const float mean2 = win[q][3 * s] * C1;
                const float var2 = __builtin_fmaf(-mean2, mean2, win[q][3 * s + 1] * C2);
                const float cov = __builtin_fmaf(-m1, mean2, win[q][3 * s + 2] * C2);
                float den, rden;
                ncc_denominator(v1 * var2, den, rden);
                const float cost = 1.0f - cov * rden;
                const bool hit = (okc[q] >> s) & 1u;
                total = hit ? total + cost : total;
                cnt = hit ? cnt + 1.0f : cnt;
            }
            // average over valid sources, +inf when fewer than two (mvs_patchmatch.py:387-388)
            const float cden = cnt + 1e-8f;
            bool cden_ok = true;
            const float avg = total * rcp_t<true>(cden, cden_ok);
            const float newc = cnt >= 2.0f ? avg : __builtin_inff();

            // ---- select (mvs_patchmatch.py:452-455 / :486-489) ----
            const bool better = outl & (newc < oldc);
            if (better) cost_io[pc] = newc;
            if (mode == MODE_PROP) {
                const bool inb_c = ((unsigned)(yc + oy) < (unsigned)H) & ((unsigned)(xc + ox) < (unsigned)W);
                const int pn = inb_c ? pc + noff : 0;
                const int ps = better ? pn : pc;
                const float nb_d = d_in[pn];
                const float t0 = n_in[3 * ps], t1 = n_in[3 * ps + 1], t2 = n_in[3 * ps + 2];
                const bool zero = better & !inb_c;
                if (outl) {
                    d_out[pc] = better ? (inb_c ? nb_d : a.depth_min) : oldd;
                    n_out[3 * pc] = zero ? 0.0f : t0;
                    n_out[3 * pc + 1] = zero ? 0.0f : t1;
                    n_out[3 * pc + 2] = zero ? 0.0f : t2;
                }
            } else {
                const uint32_t h0c = pixel_hash((uint32_t)pc, key);
                const float delta = (rng_uniform(h0c) * 2.0f - 1.0f) * a.depth_range;
                float d = oldd + delta;
                d = d < a.depth_min ? a.depth_min : d;
                d = d > a.depth_max ? a.depth_max : d;
                if (outl) d_out[pc] = better ? d : oldd;
                const unsigned long long won = __ballot(better);
                if (won != 0ull) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(won >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((unsigned)won, 0u));
                    if (better) nq[(q_tail + rank) & (NQ - 1)] = make_uint2((unsigned)pc, h0c);
                    q_tail += __popcll(won);
                    if (q_tail - q_head >= AMVS_WAVE) {
                        refine_normals(nq, q_head, AMVS_WAVE, lane, n_out, a.normal_range);
                        q_head += AMVS_WAVE;
                    }
                }
            }
        }
    }
    if (mode == MODE_REFINE) {
        while (q_tail - q_head > 0) {
            const int n = min(q_tail - q_head, AMVS_WAVE);
            refine_normals(nq, q_head, n, lane, n_out, a.normal_range);
            q_head += n;
        }
    }
}

// ------------------------------------------------------------------ dispatch -----
int strip_out_width2(int K) { return 2 * (AMVS_WAVE - K / 2); }

template <int K, int S, int MODE_T>
static hipError_t launch_fast2_one(const StepArgs &a, const dim3 &grid, const dim3 &block, hipStream_t st)
{
    unsigned XL = Step2Lds<K, S>::EXTRA;
    if (a.px2_wgs > 0) {                               // tuning override: resident workgroups per CU
        unsigned share = 160u * 1024u / (unsigned)a.px2_wgs;
        if (share > 64u * 1024u) share = 64u * 1024u;
        constexpr unsigned ST = Step2Lds<K, S>::STATIC;
        XL = share > ST ? share - ST : 0u;
    }
    hipLaunchKernelGGL((pm_step_fast2_kernel<K, S, MODE_T>), grid, block, XL, st, a);
    return hipGetLastError();
}

template <int K, int S>
static hipError_t launch_step_fast2_ks(const StepArgs &a, int nblk, hipStream_t st)
{
    const dim3 grid((nblk + AMVS_WG_WAVES - 1) / AMVS_WG_WAVES), block(AMVS_WAVE * AMVS_WG_WAVES);
    if (a.mode == MODE_REFINE) return launch_fast2_one<K, S, MODE_REFINE>(a, grid, block, st);
    if (a.mode == MODE_PROP) return launch_fast2_one<K, S, MODE_PROP>(a, grid, block, st);
    return hipErrorInvalidValue;
}

#define AMVS_FOR_S2(K, FN, ...)                                     \
    switch (S) {                                                    \
    case 2: return FN<K, 2>(__VA_ARGS__);                           \
    case 3: return FN<K, 3>(__VA_ARGS__);                           \
    case 4: return FN<K, 4>(__VA_ARGS__);                           \
    case 5: return FN<K, 5>(__VA_ARGS__);                           \
    case 6: return FN<K, 6>(__VA_ARGS__);                           \
    default: return hipErrorInvalidValue;                           \
    }

// StepArgs::tiles_x must have been formed with strip_out_width2(K)
hipError_t launch_step_fast2(int K, int S, const StepArgs &a, hipStream_t st)
{
    if (!a.pairs || a.presampled) return hipErrorInvalidValue;
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y;
    switch (K) {
    case 3: AMVS_FOR_S2(3, launch_step_fast2_ks, a, nblk, st)
    case 5: AMVS_FOR_S2(5, launch_step_fast2_ks, a, nblk, st)
    case 7: AMVS_FOR_S2(7, launch_step_fast2_ks, a, nblk, st)
    case 9: AMVS_FOR_S2(9, launch_step_fast2_ks, a, nblk, st)
    case 11: AMVS_FOR_S2(11, launch_step_fast2_ks, a, nblk, st)
    default: return hipErrorInvalidValue;
    }
}

int step_fast2_waves_per_cu() { return AMVS_FAST2_MAX_WGS_PER_CU * AMVS_WG_WAVES; }

}  // namespace amvs
