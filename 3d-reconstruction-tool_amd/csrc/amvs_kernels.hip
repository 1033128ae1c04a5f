// amvs_kernels.hip -- hand-written gfx950 kernels of the PatchMatch-MVS sweep and the
// plane-sweep stereo (reference: src/core/mvs_patchmatch.py, src/core/dense_stereo.py).
//
// Execution shape (CDNA4, wave64):
//   * one 64-lane wave owns a column strip: lane i <-> image column xbase+i, and walks
//     TH + 2*(k/2) image rows top to bottom.  Every global access of a row is one
//     coalesced 256-byte segment (depth, cost, ref gray) or a 768-byte one (normals).
//   * the k x k NCC window sums are separable: the vertical part lives in a per-lane
//     register ring of the last k sampled values (no LDS, no barriers), the horizontal
//     part is a chain of k-1 `v_add_f32_dpp ... wave_shl:1` whole-wave shifts
//     (one VALU instruction per tap).  A strip therefore yields 64 - 2*(k/2) output
//     columns per row.
//   * reference semantics need the sampled value of EVERY pixel of the window at that
//     pixel's OWN candidate depth (mvs_patchmatch.py:341-380), so halo pixels recompute
//     their candidate: neighbour pull for propagation, counter-hash RNG for refinement.
//   * state is ping-ponged (Jacobi semantics of mvs_patchmatch.py:429-455): a step reads
//     buffers "in" and writes buffers "out", so strips never see half-updated maps.
//   * block index -> strip mapping is XCD-aware: each of the 8 XCDs receives a contiguous
//     range of strips so halos and source-image rows are shared in that XCD's L2.
#define AMVS_TU_ID 1
#include "amvs_exact_common.h"

namespace amvs {

// ------------------------------------------------------------------ sweep step ---
// One cost evaluation + select over a batch of reference views
// (_compute_patch_cost / _spatial_propagation / _random_refinement /
//  _compute_confidence, mvs_patchmatch.py:323-534).
// minimum waves per SIMD the register allocator must leave room for (512 VGPRs per SIMD lane)
#ifndef AMVS_MIN_WAVES_BIAS
#define AMVS_MIN_WAVES_BIAS 0
#endif
constexpr int min_waves(int K, int S)
{
    // the smaller of what the registers allow (the window-sum stage holds (S + 1) K values) and what the LDS rings
    // of the workgroups of a CU allow (13 x 13 and up: three workgroups, 17 x 17 and up: two)
    const int by_regs = (S + 1) * K <= 40 ? 4 : ((S + 1) * K <= 60 ? 3 : 2);
    const int nl = S < ring_lds_sources(K) ? S : ring_lds_sources(K);
    const int by_lds = 163840 / (AMVS_WG_WAVES * (nl + 1) * K * AMVS_WAVE * 4 + 4096);
    return (by_regs < by_lds ? by_regs : (by_lds < 1 ? 1 : by_lds)) + AMVS_MIN_WAVES_BIAS;
}

// MODE_T: MODE_PROP / MODE_REFINE are compiled as their own kernels (99 % of the launches: the mode
// switches, the other modes' code and, for propagation steps, the whole RNG hash fall away at
// compile time); -1 is the generic kernel, used for MODE_EVAL and MODE_CONF.
// PAIR (StepArgs::paired, AMVS_SCHEDULE_PAIRED): 2 strip columns x 2 vertically adjacent bands per
// workgroup; the bands walk towards each other and exchange the samples of their last K/2 rows through LDS
// (K/2 halo rows per band instead of K - 1; same samples, same sums in the same order: bit-identical maps) --
// see pm_step_fast_kernel (amvs_kernels_fast.hip), whose structure this follows.
template <int K, int S, bool U8, int MODE_T, bool PAIR = false>
__global__ __launch_bounds__(AMVS_WAVE * (PAIR ? PAIR_WAVES : AMVS_WG_WAVES), min_waves(K, S)) void pm_step_kernel(const StepArgs a)
{
    constexpr int WGW = PAIR ? PAIR_WAVES : AMVS_WG_WAVES;      // waves of this workgroup
    constexpr int HALF = K / 2;
    constexpr int OUTW = AMVS_WAVE - 2 * HALF;
    constexpr float INV_AREA = 1.0f / (float)(K * K);
    __shared__ float lut[256];
    #ifdef AMVS_HSUM_LDS
    __shared__ float4 hbuf[(AMVS_WAVE + K - 1) * HSum<S>::NV4];
#else
    float4 *hbuf = nullptr;
#endif
    __shared__ float lring_all[WGW * (Ring<K, S>::NL + 1) * K * AMVS_WAVE];
    constexpr int NQ = 2 * AMVS_WAVE;                    // refinement winners waiting for their normal
    __shared__ uint32_t nq_all[WGW * NQ];
    // paired bands: the partner's rows of the LDS-resident sources are read from the partner's ring itself (see
    // pm_step_fast_kernel); exchange rows only for the XS sources with register rings
    constexpr int XS = S - Ring<K, S>::NL > 0 ? S - Ring<K, S>::NL : 0;
    constexpr int XW = HALF * XS * AMVS_WAVE;            // floats of a wave's exchange rows [row][register source][lane]
    __shared__ float xbuf_all[PAIR && XS > 0 ? WGW * XW : 1];

    const int lane = threadIdx.x & (AMVS_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / AMVS_WAVE);
    float *lring = lring_all + wv * ((Ring<K, S>::NL + 1) * K * AMVS_WAVE);
    uint32_t *nq = nq_all + wv * NQ;
    int q_head = 0, q_tail = 0;                          // wave-uniform; at most 63 + 58 entries queued
    window_sums_init<K, S>(hbuf, lane);
    if (U8) fill_gray_lut(lut, lane);
    const int tiles_per_job = a.tiles_x * a.tiles_y;
    int job_id, ty, tx;
    bool paired = false;                       // this wave has a partner band to exchange with
    int up = 0;                                // 1: the wave walks up the image (lower band of a pair)
    if constexpr (PAIR) {
        constexpr int PC = AMVS_PAIR_COLS;
        const int col_pairs = (a.tiles_x + PC - 1) / PC, pair_rows = (a.tiles_y + 1) / 2;
        const int wg = xcd_remap(blockIdx.x, gridDim.x);
        job_id = wg / (col_pairs * pair_rows);
        const int rem = wg - job_id * (col_pairs * pair_rows);
        const int py = rem / col_pairs, px = rem - py * col_pairs;
        tx = PC * px + (wv % PC);
        up = wv / PC;
        ty = 2 * py + up;
        paired = 2 * py + 1 < a.tiles_y;
        // (every wave fills the decode table first; a wave without a strip leaves before the first barrier of
        //  the row loop, as do the partners of an odd last band that does not exist)
        if (job_id >= a.n_jobs || tx >= a.tiles_x || ty >= a.tiles_y) return;
    } else {
        const int t = xcd_remap(blockIdx.x, gridDim.x) * AMVS_WG_WAVES + wv;
        if (t >= a.n_jobs * tiles_per_job) return;                 // last workgroup only
        strip_of(a, t, job_id, ty, tx);
    }

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W, mode = MODE_T >= 0 ? MODE_T : a.mode;
    const long long HW = (long long)H * W;

    const float *__restrict__ ref = a.images + job->ref_img * a.img_stride;
    // ref gray of the packed path: low byte of the padded row-pair map (pitch W+4, origin at pixel (0,0))
    const GlobalU16 ref_pairs = U8 ? (GlobalU16)job->ref_pairs : (GlobalU16)a.pairs;   // global, not FLAT, loads
    constexpr int PADW = U8 ? 2 * AMVS_PAIR_BORDER : 0;
#if AMVS_CODE_BYTES
#define AMVS_REF_CODE(i) (((const __attribute__((address_space(1))) uint8_t *)ref_pairs)[i])
#else
#define AMVS_REF_CODE(i) (ref_pairs[i])
#endif
    const float *__restrict__ d_in = a.d_in + job->slot * HW;
    float *__restrict__ d_out = a.d_out + job->slot * HW;
    float *cost_io = a.cost + job->slot * HW;          // read and (where the candidate wins) written
    float *nbuf0 = a.nbuf[0] + job->slot * HW * 3, *nbuf1 = a.nbuf[1] + job->slot * HW * 3;
    float *__restrict__ aux = a.aux + job->slot * HW;

    const StreamKey key = stream_key(a.seed, job->stream_view, a.draw);

    // validity window of the projection: patch bounds (mvs_patchmatch.py:362-363) or
    // image bounds for the confidence pass (:516-517)
    const SampleConsts sc = make_sample_consts(H, W, mode == MODE_CONF ? 0.0f : (float)HALF,
                                               mode == MODE_CONF ? (float)W : (float)(W - HALF),
                                               mode == MODE_CONF ? (float)H : (float)(H - HALF));

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int th_w = min(a.TH, H - y0);                        // output rows of this strip
    // classic: th_w + K - 1 rows.  PAIR: every wave of the workgroup runs the same a.TH + K - 1 steps (common
    // barriers); a wave whose band is shorter idles first, so that the partners meet at their boundary
    const int rows = PAIR ? a.TH + 2 * HALF : th_w + 2 * HALF;
    const int idle_first = (PAIR && up) ? a.TH - th_w : 0;
    const int n_loc = th_w + 2 * HALF;                         // steps this wave works
    const int n_own = paired ? th_w + HALF : n_loc;            // ... of which it samples itself
    const int y_start = up ? y0 + th_w + HALF - 1 : y0 - HALF, dy = up ? -1 : 1;
    float *xmine = PAIR ? xbuf_all + wv * XW : nullptr;
    const float *xpartner = PAIR ? xbuf_all + (wv ^ AMVS_PAIR_COLS) * XW : nullptr;
    const float *lring_p = PAIR ? lring_all + (wv ^ AMVS_PAIR_COLS) * ((Ring<K, S>::NL + 1) * K * AMVS_WAVE) : nullptr;
    int pslot = 0;                      // ring slot of the partner's own row next to the boundary
    if (PAIR && paired) pslot = (min(a.TH, H - (ty ^ 1) * a.TH) + HALF - 1) % K;

    float ring_r[K];
    float ring_v[Ring<K, S>::NR][K];
    typename Hist<K, S>::T hist_ok = 0;    // S validity bits per row, newest row in the top bits
    uint32_t hist_h0[HALF + 1];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        ring_r[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < Ring<K, S>::NR; ++s) ring_v[s][i] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i <= HALF; ++i) hist_h0[i] = 0u;
    int wslot = 0;                      // LDS ring slot the next row is written to

    // The candidate of pixel (y,x) is read at (y+oy, x+ox): the neighbour for a propagation step
    // (mvs_patchmatch.py:431-441), the pixel itself ((0,0)) for every other mode.  All loads use
    // clamped indices (dead lanes read element 0), so there is no branch in the load path.
    //
    // The launch time follows the instruction stream: software-pipelining the
    // row loop by one row (requests of row r+1 behind the epilogue of row r) measured -2 % at +60
    // VGPRs, requesting the next row's depth / ref gray a row ahead measured nothing.
    const int oy = mode == MODE_PROP ? a.oy : 0, ox = mode == MODE_PROP ? a.ox : 0;
    const int noff = oy * W + ox;

    for (int r = 0; r < rows; ++r) {
#if AMVS_WG_WAVES > 1 && AMVS_WG_SYNC_ROWS > 0
        // re-align the workgroup's strips (waves that have ended do not take part in s_barrier, so
        // strips of different heights in one workgroup are fine)
        if (r % AMVS_WG_SYNC_ROWS == 0) __builtin_amdgcn_s_barrier();
#endif
        if constexpr (PAIR) {
            // the partners have written the samples of their last K/2 own rows (steps TH .. TH + K/2 - 1)
            if (r == a.TH + HALF) __syncthreads();
        }
        const int loc = r - idle_first;                        // this wave's step
        if (PAIR && (loc < 0 || loc >= n_loc)) continue;        // (wave-uniform)
        const int yr = PAIR ? y_start + dy * loc : y0 - HALF + r;
        const bool own = !PAIR || loc < n_own;                  // sampled here, not taken from the partner
        const bool live = col_in & ((unsigned)yr < (unsigned)H);
        const bool inb = live & ((unsigned)(yr + oy) < (unsigned)H) & ((unsigned)(xr + ox) < (unsigned)W);
        const int pix = yr * W + xr;
        const float d_raw = d_in[AMVS_IDX(inb ? pix + noff : 0, HW)];       // re-read by neighbours: cached
        // ref gray: in the packed path the low byte of the row-pair map decoded through the table
        // (the same float as the float32 map holds, at half the bytes)
        const float r_raw = U8 ? lut[AMVS_REF_CODE(AMVS_IDX_LOHI(live ? pix + PADW * yr : 0, -((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER), (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER) - ((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER))) & 0xFFu]
                               : ref[AMVS_IDX(live ? pix : 0, HW)];

        // ---- candidate depth of this (possibly halo) pixel ----
        // outside the image the pulled candidate is depth_min (F.pad value)
        float dc = inb ? depth_untag(d_raw, a.depth_mask) : a.depth_min;
        // depth + (rand*2-1)*range, clamped (mvs_patchmatch.py:471-472)
        const uint32_t h0 = pixel_hash((uint32_t)pix, key);
        {
            float delta = (rng_uniform(h0) * 2.0f - 1.0f) * a.depth_range;
            float d = dc + delta;
            d = d < a.depth_min ? a.depth_min : d;
            d = d > a.depth_max ? a.depth_max : d;
            dc = mode == MODE_REFINE ? d : dc;
        }
        const float rv = live ? r_raw : 0.0f;
        float v[S];
        unsigned okbits = 0u;
        if (PAIR && !own) {
            // a row of the partner band: its samples, taken at its own candidates, from LDS (the partner
            // wrote them walking towards the boundary: the row next to it last)
            constexpr int NLS = Ring<K, S>::NL;
            const float *xp = xpartner + (HALF - 1 - (loc - n_own)) * (XS * AMVS_WAVE);
#pragma unroll
            for (int s = 0; s < S; ++s)
                v[s] = s < NLS ? lring_p[((s + 1) * K + pslot) * AMVS_WAVE + lane] : xp[(s < NLS ? 0 : s - NLS) * AMVS_WAVE + lane];
            pslot = pslot == 0 ? K - 1 : pslot - 1;
        } else {
            JobCP jr = reload(job);
            const Vec3 Pw = backproject(jr->Kinv, jr->Rref, jr->tref, xr, yr, dc);
            okbits = sample_sources_checked<S, U8, AMVS_PM_ROW_CHECK_SAMPLING, AMVS_STEP_PRIO>(jr, a, sc, lut, Pw, live, v);
            if constexpr (PAIR) {
                if (XS > 0 && paired && loc >= n_own - HALF) {  // the last K/2 own rows: for the partner
                    float *xm = xmine + (loc - (n_own - HALF)) * (XS * AMVS_WAVE);
#pragma unroll
                    for (int s = Ring<K, S>::NL; s < S; ++s) xm[(s - Ring<K, S>::NL) * AMVS_WAVE + lane] = v[s];
                }
            }
        }

        // ---- push into the vertical rings ----
        ring_push<K, S>(lring, lane, wslot, ring_r, ring_v, rv, v);
        wslot = wslot + 1 == K ? 0 : wslot + 1;          // now the slot of the oldest row
        hist_ok = (hist_ok >> S) | ((typename Hist<K, S>::T)okbits << (S * HALF));
#pragma unroll
        for (int i = 0; i < HALF; ++i) hist_h0[i] = hist_h0[i + 1];
        hist_h0[HALF] = h0;

        if ((PAIR ? loc : r) < 2 * HALF) continue;

        // ---- window sums, NCC, aggregate for centre row yc and centre column xc ----
        const int yc = PAIR ? yr - dy * HALF : yr - HALF;
        const int xc = xr + HALF;
        const bool outl = (lane < OUTW) & (xc < W);
        const int pc = AMVS_IDX(outl ? yc * W + xc : 0, HW);
        const float oldd_tagged = d_in[pc], oldc = cost_io[pc];
        const float oldd = depth_untag(oldd_tagged, a.depth_mask);
        const unsigned buf_c = depth_buffer(oldd_tagged);     // where this pixel's current normal lives
        // propagation: the neighbour the candidate was pulled from (requested with the other state
        // loads: behind the selection it would expose a memory round trip in every row).  Lanes
        // without an output pixel have pc = 0: they must not form pc + noff, which lies BEFORE the map
        // for the negative offsets of odd iterations.
        const bool inb_c = ((unsigned)(yc + oy) < (unsigned)H) & ((unsigned)(xc + ox) < (unsigned)W);
        const int pn = AMVS_IDX((outl & inb_c) ? pc + noff : 0, HW);
        const float nb_tagged = mode == MODE_PROP ? d_in[pn] : 0.0f;
        // the centre pixel was sampled by lane+HALF, HALF rows ago
        const unsigned okc = (unsigned)__shfl_down((int)(unsigned)hist_ok, HALF);     // low S bits: row r-HALF
        const uint32_t h0c = (uint32_t)__shfl_down((int)hist_h0[0], HALF);

        float bvs[S], bvvs[S], brvs[S], br, brr;
        if (PAIR && up) window_sums<K, S, true>(lring, wslot, ring_r, ring_v, hbuf, lane, bvs, bvvs, brvs, br, brr);
        else window_sums<K, S>(lring, wslot, ring_r, ring_v, hbuf, lane, bvs, bvvs, brvs, br, brr);
        const float m1 = br * INV_AREA;
        const float v1 = brr * INV_AREA - m1 * m1;

#if AMVS_PM_ROW_CHECK_NCC
        // NCC of every source + aggregation; optimistic lean sqrt / reciprocal with one check per
        // row (amvs_device.h), IEEE repeat if a window's variance product left the verified range
        float total, cnt;
        auto ncc_stage = [&](auto lean, bool &ok) {
            constexpr bool LEAN = decltype(lean)::value;
            total = 0.0f; cnt = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float bv = bvs[s], bvv = bvvs[s], brv = brvs[s];
                // _ncc_cost (mvs_patchmatch.py:403-411)
                const float mean2 = bv * INV_AREA;
                const float var2 = bvv * INV_AREA - mean2 * mean2;
                const float cov = brv * INV_AREA - m1 * mean2;
                const float den = sqrt_t<LEAN>(v1 * var2, ok) + 1e-8f;
                const float ncc = qdiv(cov, den, rcp_t<LEAN>(den, ok));
                const float cost = 1.0f - ncc;
                const bool oks = (okc >> s) & 1u;
                // confidence: consistent = valid & (1 - cost > 0.6)   (mvs_patchmatch.py:530-532)
                const float ncc2 = 1.0f - cost;
                const bool hit = mode == MODE_CONF ? (oks & (ncc2 > 0.6f)) : oks;
                // cost: total += where(valid, cost, 0); count += valid   (:383-384)
                total = (hit & (mode != MODE_CONF)) ? total + cost : total;
                cnt = hit ? cnt + 1.0f : cnt;
            }
        };
        {
            bool ok = true;
            ncc_stage(std::true_type{}, ok);
            if (__builtin_expect(!__all(ok), 0)) ncc_stage(std::false_type{}, ok);
        }
#else
        float total = 0.0f, cnt = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float bv = bvs[s], bvv = bvvs[s], brv = brvs[s];
            // _ncc_cost (mvs_patchmatch.py:403-411)
            const float mean2 = bv * INV_AREA;
            const float var2 = bvv * INV_AREA - mean2 * mean2;
            const float cov = brv * INV_AREA - m1 * mean2;
            float den, rden;
            ncc_denominator(v1 * var2, den, rden);
            const float ncc = qdiv(cov, den, rden);
            const float cost = 1.0f - ncc;
            const bool oks = (okc >> s) & 1u;
            // confidence: consistent = valid & (1 - cost > 0.6)   (mvs_patchmatch.py:530-532)
            const float ncc2 = 1.0f - cost;
            const bool hit = mode == MODE_CONF ? (oks & (ncc2 > 0.6f)) : oks;
            // cost: total += where(valid, cost, 0); count += valid   (:383-384)
            total = (hit & (mode != MODE_CONF)) ? total + cost : total;
            cnt = hit ? cnt + 1.0f : cnt;
        }
#endif
        // lanes that own an output pixel; control flow below stays wave-uniform (the refinement
        // queue needs every lane), so the stores are predicated instead of skipped
        const bool act = outl;

        if (mode == MODE_CONF) {
            if (act) aux[pc] = cnt;
            continue;
        }

        // average over valid sources, +inf when fewer than two (mvs_patchmatch.py:387-388)
        const float cden = cnt + 1e-8f;              // 1e-8 ... S: always inside the lean reciprocal's range
        bool cden_ok = true;
        const float avg = qdiv(total, cden, rcp_t<true>(cden, cden_ok));
        const float newc = cnt >= 2.0f ? avg : __builtin_inff();
        if (mode == MODE_EVAL) {
            if (act) aux[pc] = newc;
            continue;
        }

        // ---- select (mvs_patchmatch.py:452-455 / :486-489) ----
        // State is only moved where it has to be: depth goes to the other buffer for every pixel;
        // cost is rewritten in place and only where the candidate wins; a refinement step rewrites
        // the normal in place and only where the candidate wins (4.5 % of the pixels on average).
        const bool better = act & (newc < oldc);
        if (better) cost_io[pc] = newc;
        if (mode == MODE_PROP) {
            // candidate = the neighbour's pre-step state; out-of-image neighbour: depth_min and a
            // zero normal (F.pad, :431-442).  Only the winners' normals move (StepArgs::nbuf).
            // out-of-image neighbour: depth_min and a zero normal (F.pad, :431-442).  Only the winners'
            // normals move (StepArgs::nbuf), queued and moved 64 at a time like the refinement winners'.
            const float nb_d = depth_untag(nb_tagged, a.depth_mask);
            if (act) d_out[pc] = better ? depth_tag(inb_c ? nb_d : a.depth_min, buf_c ^ 1u) : oldd_tagged;
            const unsigned long long won = __ballot(better);
            if (won != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(won >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)won, 0u));
                if (better) nq[(q_tail + rank) & (NQ - 1)] = propagate_entry(pc, buf_c, inb_c, depth_buffer(nb_tagged));
                q_tail += __popcll(won);
                if (q_tail - q_head >= AMVS_WAVE) {
                    propagate_normals(nq, q_head, AMVS_WAVE, lane, nbuf0, nbuf1, noff, (int)HW);
                    q_head += AMVS_WAVE;
                }
            }
        } else {
            float delta = (rng_uniform(h0c) * 2.0f - 1.0f) * a.depth_range;
            float d = oldd + delta;
            d = d < a.depth_min ? a.depth_min : d;
            d = d > a.depth_max ? a.depth_max : d;
            if (act) d_out[pc] = better ? depth_tag(d, buf_c) : oldd_tagged;
            // The winners' normals (normalize(normal + randn*range), mvs_patchmatch.py:475-476) are
            // not updated here: a row has ~3 winners among its 58 pixels, yet the ~150-instruction
            // update would run for the whole wave on almost every row.  Winners are queued in LDS
            // (pixel, hash) and updated 64 at a time -- same arithmetic, 1/20 of the issue slots.
            // Nothing else reads or writes a pixel's normal during a refinement launch.
            const unsigned long long won = __ballot(better);
            if (won != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(won >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)won, 0u));
                if (better) nq[(q_tail + rank) & (NQ - 1)] = refine_entry(pc, buf_c);
                q_tail += __popcll(won);
                if (q_tail - q_head >= AMVS_WAVE) {
                    refine_normals(nq, q_head, AMVS_WAVE, lane, nbuf0, nbuf1, a.normal_range, key, (int)HW);
                    q_head += AMVS_WAVE;
                }
            }
        }
    }
    if (mode == MODE_REFINE || mode == MODE_PROP) {
        while (q_tail - q_head > 0) {
            const int n = min(q_tail - q_head, AMVS_WAVE);
            if (mode == MODE_REFINE) refine_normals(nq, q_head, n, lane, nbuf0, nbuf1, a.normal_range, key, (int)HW);
            else propagate_normals(nq, q_head, n, lane, nbuf0, nbuf1, noff, (int)HW);
            q_head += n;
        }
    }
}

// ------------------------------------------------------------------ split step (experiment) ---
// Sampling stage of a sweep step as a kernel of its own: one thread per PX pixels of the reference
// view (every pixel exactly once, no strip halo, no rings): candidate depth, back-projection, the S
// projections / gathers / bilinear samples.  Writes the samples ([pixel][S] float32), the candidate
// depth and the validity bits for a window / NCC / select kernel to stream.

// ------------------------------------------------------------------ sample dump ---
// Test hook (amvs_sample_sources): the bilinear sample of every pixel in every source at the
// pixel's own depth (mvs_patchmatch.py:341-377) and the validity bits -- the stage before the box
// filter, which the tests' CPU checker reproduces bit for bit against torch-CPU's grid_sample.
template <int S, bool U8>
__global__ __launch_bounds__(AMVS_WAVE) void sample_dump_kernel(const StepArgs a, float *__restrict__ out,
                                                                unsigned char *__restrict__ valid_out)
{
    __shared__ float lut[256];
    const int lane = threadIdx.x;
    if (U8) fill_gray_lut(lut, lane);
    const JobCP job = (JobCP)a.jobs;
    const int H = a.H, W = a.W, half = a.TH;                       // TH carries k/2 here
    const bool conf = a.mode == MODE_CONF, nobounds = a.mode == MODE_EVAL + 100;
    const float inf = __builtin_inff();
    const SampleConsts sc = make_sample_consts(H, W, nobounds ? -inf : (conf ? 0.0f : (float)half),
                                               nobounds ? inf : (conf ? (float)W : (float)(W - half)),
                                               nobounds ? inf : (conf ? (float)H : (float)(H - half)));
    const long long HW = (long long)H * W;
    const int x = blockIdx.x * AMVS_WAVE + lane, y = blockIdx.y;
    const bool live = x < W;
    const float d = a.d_in[AMVS_IDX(live ? y * W + x : 0, HW)];
    JobCP jr = reload(job);
    const Vec3 Pw = backproject(jr->Kinv, jr->Rref, jr->tref, x, y, d);
    float v[S];
    const unsigned okbits = sample_sources_checked<S, U8, true>(jr, a, sc, lut, Pw, live, v);
    if (live) {
#pragma unroll
        for (int s = 0; s < S; ++s) out[s * HW + y * W + x] = v[s];
        valid_out[y * W + x] = (unsigned char)okbits;
    }
}

template <int S>
static hipError_t launch_sample_dump_s(const StepArgs &a, float *out, unsigned char *valid_out, hipStream_t st)
{
    const dim3 grid((a.W + AMVS_WAVE - 1) / AMVS_WAVE, a.H), blk(AMVS_WAVE);
    if (a.pairs) hipLaunchKernelGGL((sample_dump_kernel<S, true>), grid, blk, 0, st, a, out, valid_out);
    else hipLaunchKernelGGL((sample_dump_kernel<S, false>), grid, blk, 0, st, a, out, valid_out);
    return hipGetLastError();
}

hipError_t launch_sample_dump(int S, const StepArgs &a, float *out, unsigned char *valid_out, hipStream_t st)
{
    if (a.fast) return launch_sample_dump_fast(S, a, out, valid_out, st);
    switch (S) {
    case 2: return launch_sample_dump_s<2>(a, out, valid_out, st);
    case 3: return launch_sample_dump_s<3>(a, out, valid_out, st);
    case 4: return launch_sample_dump_s<4>(a, out, valid_out, st);
    case 5: return launch_sample_dump_s<5>(a, out, valid_out, st);
    case 6: return launch_sample_dump_s<6>(a, out, valid_out, st);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------ ref stats ----
// mean / variance of a gray image under the k x k zero-padded box filter
// (mvs_patchmatch.py:403,406): computed once per image and patch size instead of once
// per cost evaluation.
template <int K>
__global__ __launch_bounds__(AMVS_WAVE) void box_stats_kernel(const float *__restrict__ images,
                                                              long long img_stride, int H, int W,
                                                              int TH, int tiles_x, int tiles_y,
                                                              int first_img,
                                                              float *__restrict__ mean_out,
                                                              float *__restrict__ var_out)
{
    constexpr int HALF = K / 2;
    constexpr int OUTW = AMVS_WAVE - 2 * HALF;
    constexpr float INV_AREA = 1.0f / (float)(K * K);
    const int lane = threadIdx.x;
    const int t = blockIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int img_id = first_img + t / tiles;
    const int rem = t % tiles;
    const int ty = rem / tiles_x, tx = rem % tiles_x;
    const float *__restrict__ img = images + img_id * img_stride;
    float *__restrict__ mo = mean_out + img_id * img_stride;
    float *__restrict__ vo = var_out + img_id * img_stride;
    const int xr = tx * OUTW - HALF + lane;
    const int y0 = ty * TH;
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int rows = min(TH, H - y0) + 2 * HALF;
    float ring[K];
#pragma unroll
    for (int i = 0; i < K; ++i) ring[i] = 0.0f;
    for (int r = 0; r < rows; ++r) {
        const int yr = y0 - HALF + r;
        const bool live = col_in & ((unsigned)yr < (unsigned)H);
        const float rv = live ? img[AMVS_IDX(yr * W + xr, (long long)H * W)] : 0.0f;
#pragma unroll
        for (int i = 0; i < K - 1; ++i) ring[i] = ring[i + 1];
        ring[K - 1] = rv;
        if (r < 2 * HALF) continue;
        float cr = ring[0], crr = ring[0] * ring[0];
#pragma unroll
        for (int i = 1; i < K; ++i) { cr = cr + ring[i]; crr = __builtin_fmaf(ring[i], ring[i], crr); }
        float br = cr, brr = crr;
#pragma unroll
        for (int j = 1; j < K; ++j) { br = wave_shl1(br) + cr; brr = wave_shl1(brr) + crr; }
        const int xc = xr + HALF, yc = yr - HALF;
        if (lane < OUTW && xc < W) {
            const float m = br * INV_AREA;
            mo[AMVS_IDX(yc * W + xc, (long long)H * W)] = m;
            vo[AMVS_IDX(yc * W + xc, (long long)H * W)] = brr * INV_AREA - m * m;
        }
    }
}

// ------------------------------------------------------------------ 8-bit pack ---
// Build the packed row-pair map of one view (layout: amvs_device.h, "U8 = true") and test that it
// is lossless: pixel (y,x) gets code = rint(g*255) clamped to [0,255]; `inexact` is raised if any
// pixel differs from (float)code / 255.0f, in which case the sweep keeps sampling the float32 map.
// `pairs` is the padded map's first element; entry (y,x), y in [-2, H+2), x in [-2, W+2), is
// code(y,x) | code(y+1,x) << 8 with code = 0 outside the image.
__global__ __launch_bounds__(256) void pack_pairs_kernel(const float *__restrict__ img, int H, int W,
                                                         uint16_t *__restrict__ pairs,
                                                         int *__restrict__ inexact)
{
    constexpr int B = AMVS_PAIR_BORDER;
    const int PW = W + 2 * B;
    const long long n = (long long)(H + 2 * B) * PW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / PW) - B, x = (int)(i % PW) - B;
        const bool xin = (unsigned)x < (unsigned)W;
        const bool in0 = xin & ((unsigned)y < (unsigned)H), in1 = xin & ((unsigned)(y + 1) < (unsigned)H);
        const float g0 = in0 ? img[AMVS_IDX((long long)y * W + x, (long long)H * W)] : 0.0f;
        const float g1 = in1 ? img[AMVS_IDX((long long)(y + 1) * W + x, (long long)H * W)] : 0.0f;
        const int c0 = min(max((int)__builtin_rintf(g0 * 255.0f), 0), 255);
        const int c1 = min(max((int)__builtin_rintf(g1 * 255.0f), 0), 255);
        if (in0 & !((float)c0 / 255.0f == g0)) atomicOr(inexact, 1);
#if AMVS_CODE_BYTES
        ((uint8_t *)pairs)[i] = (uint8_t)c0;
#else
        pairs[i] = (uint16_t)(c0 | (c1 << 8));
#endif
    }
}

hipError_t launch_pack_pairs(const float *img, int H, int W, uint16_t *pairs, int *inexact,
                             hipStream_t st)
{
    const long long n = (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER);
    const int bx = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_pairs_kernel, dim3(bx), dim3(256), 0, st, img, H, W, pairs, inexact);
    return hipGetLastError();
}

// ushorts one padded map occupies (before alignment)
long long pair_map_elems(int H, int W)
{
    return (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER);
}
// ushort offset of image pixel (0,0) inside a padded map
long long pair_map_origin(int W) { return (long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER; }
int pair_map_texel_bytes() { return AMVS_CODE_BYTES ? 1 : 2; }

// ------------------------------------------------------------------ init ---------
// depth = exp(rand*(ln dmax - ln dmin) + ln dmin); normal = normalize(randn*0.3,
// randn*0.3, -1); best_cost = +inf   (mvs_patchmatch.py:268-284)
__global__ __launch_bounds__(256) void pm_init_kernel(const Job *__restrict__ jobs, long long HW,
                                                      unsigned long long seed, float log_scale,
                                                      float log_min, float *__restrict__ depth,
                                                      float *__restrict__ normal,
                                                      float *__restrict__ cost)
{
    const JobCP job = (JobCP)(jobs + blockIdx.y);
    const StreamKey key = stream_key(seed, job->stream_view, 0u);
    const long long base = job->slot * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW;
         i += (long long)gridDim.x * blockDim.x) {
        const uint32_t h0 = pixel_hash((uint32_t)i, key);
        depth[base + i] = exp_poly(rng_uniform(h0) * log_scale + log_min);
        float g0, g1, g2;
        rng_normals3(h0, g0, g1, g2);
        float nx = g0 * 0.3f, ny = g1 * 0.3f, nz = -1.0f;
        normalize3(nx, ny, nz);
        normal[3 * (base + i)] = nx;
        normal[3 * (base + i) + 1] = ny;
        normal[3 * (base + i) + 2] = nz;
        cost[base + i] = __builtin_inff();      // (depth > 0: the normal just written, buffer 0, is the current one)
    }
}

// tagged state -> plain maps (launch_resolve_state, StepArgs::nbuf)
__global__ __launch_bounds__(256) void resolve_state_kernel(const Job *__restrict__ jobs, long long HW, float *depth,
                                                            float *nbuf0, const float *__restrict__ nbuf1,
                                                            float *depth_out, float *normal_out, long long out_slot0)
{
    const JobCP job = (JobCP)(jobs + blockIdx.y);
    const long long base = job->slot * HW;
    const long long obase = (job->slot - out_slot0) * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW;
         i += (long long)gridDim.x * blockDim.x) {
        const float d = depth[base + i];
        const unsigned buf = depth_buffer(d);
        const float da = depth_untag(d, 0x7FFFFFFFu);
        if (depth_out) {
            const float *src = (buf ? nbuf1 : nbuf0) + 3 * (base + i);
            const float t0 = src[0], t1 = src[1], t2 = src[2];
            depth_out[obase + i] = da;
            normal_out[3 * (obase + i)] = t0; normal_out[3 * (obase + i) + 1] = t1; normal_out[3 * (obase + i) + 2] = t2;
        } else if (buf) {
            depth[base + i] = da;
            const float *src = nbuf1 + 3 * (base + i);
            const float t0 = src[0], t1 = src[1], t2 = src[2];
            float *dst = nbuf0 + 3 * (base + i);
            dst[0] = t0; dst[1] = t1; dst[2] = t2;
        }
    }
}

__global__ __launch_bounds__(256) void rng_fill_kernel(unsigned long long seed, unsigned view,
                                                       unsigned draw, long long n,
                                                       float *__restrict__ u_out,
                                                       float *__restrict__ n_out)
{
    const StreamKey key = stream_key(seed, view, draw);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const uint32_t h0 = pixel_hash((uint32_t)i, key);
        if (u_out) u_out[i] = rng_uniform(h0);
        if (n_out) {
            float g0, g1, g2;
            rng_normals3(h0, g0, g1, g2);
            n_out[3 * i] = g0; n_out[3 * i + 1] = g1; n_out[3 * i + 2] = g2;
        }
    }
}

// ------------------------------------------------------------------ self test ----
// Compare rcp_rn / sqrt_rn with the IEEE expansions on EVERY float bit pattern.
__global__ __launch_bounds__(256) void lean_math_check_kernel(unsigned long long *mismatch)
{
    unsigned long long bad_rcp = 0, bad_sqrt = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32);
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = rcp_rn(x), b = 1.0f / x;
        const float c = sqrt_rn(x), d = __builtin_sqrtf(x);
        // NaN results compare equal when both are NaN
        bad_rcp += (__float_as_uint(a) != __float_as_uint(b)) & !((a != a) & (b != b));
        bad_sqrt += (__float_as_uint(c) != __float_as_uint(d)) & !((c != c) & (d != d));
    }
    if (bad_rcp) atomicAdd(&mismatch[0], bad_rcp);
    if (bad_sqrt) atomicAdd(&mismatch[1], bad_sqrt);
}

hipError_t launch_lean_math_check(unsigned long long *mismatch, hipStream_t st)
{
    hipLaunchKernelGGL(lean_math_check_kernel, dim3(8192), dim3(256), 0, st, mismatch);
    return hipGetLastError();
}

// ------------------------------------------------------------------ dispatch -----
// Resident workgroups per CU of the sweep step are capped through unused dynamic LDS: the launch is
// bound by the CU's L1 line rate for scattered gathers, not by latency, so waves beyond ~16 per CU
// only widen the band of source rows an XCD touches at once (more L2 misses); measured on the fast
// kernel 24 waves 0.897 ms, 20: 0.822, 16: 0.814, 12: 0.856 (amvs_kernels_fast.hip).
// (StepArgs::wg_cap; 0 = AMVS_DEFAULT_WGS_PER_CU)
template <auto Kern>
static unsigned step_extra_lds(int wg_cap)
{
    static const unsigned static_lds = [] {
        hipFuncAttributes at{};
        if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(Kern)) != hipSuccess) return ~0u;
        return (unsigned)at.sharedSizeBytes;
    }();
    const unsigned share = 160u * 1024u / (unsigned)(wg_cap > 0 ? wg_cap : AMVS_DEFAULT_WGS_PER_CU);
    return static_lds < share ? share - static_lds : 0u;
}

template <int K, int S>
static hipError_t launch_step_ks(const StepArgs &a, int nblk, hipStream_t st)
{
    const int nwg = (nblk + AMVS_WG_WAVES - 1) / AMVS_WG_WAVES;
    const dim3 grid(nwg), block(AMVS_WAVE * AMVS_WG_WAVES);
#define AMVS_LAUNCH_STEP(U8, M) \
    hipLaunchKernelGGL((pm_step_kernel<K, S, U8, M>), grid, block, (step_extra_lds<&pm_step_kernel<K, S, U8, M>>(a.wg_cap)), st, a)
    if constexpr (step_pair_supported_ks(K, S)) {
        if (a.pairs && a.paired && (a.mode == MODE_REFINE || a.mode == MODE_PROP)) {
            const int pwg = a.n_jobs * ((a.tiles_x + AMVS_PAIR_COLS - 1) / AMVS_PAIR_COLS) * ((a.tiles_y + 1) / 2);
            const dim3 pgrid(pwg), pblock(AMVS_WAVE * PAIR_WAVES);
            if (a.mode == MODE_REFINE)
                hipLaunchKernelGGL((pm_step_kernel<K, S, true, MODE_REFINE, true>), pgrid, pblock,
                                   (step_extra_lds<&pm_step_kernel<K, S, true, MODE_REFINE, true>>(a.wg_cap)), st, a);
            else
                hipLaunchKernelGGL((pm_step_kernel<K, S, true, MODE_PROP, true>), pgrid, pblock,
                                   (step_extra_lds<&pm_step_kernel<K, S, true, MODE_PROP, true>>(a.wg_cap)), st, a);
            return hipGetLastError();
        }
    }
    if (a.pairs) {
        if (a.mode == MODE_REFINE) AMVS_LAUNCH_STEP(true, MODE_REFINE);
        else if (a.mode == MODE_PROP) AMVS_LAUNCH_STEP(true, MODE_PROP);
        else AMVS_LAUNCH_STEP(true, -1);
    } else {
        if (a.mode == MODE_REFINE) AMVS_LAUNCH_STEP(false, MODE_REFINE);
        else if (a.mode == MODE_PROP) AMVS_LAUNCH_STEP(false, MODE_PROP);
        else AMVS_LAUNCH_STEP(false, -1);
    }
#undef AMVS_LAUNCH_STEP
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

template <int K, int S>
static int step_occupancy_ks(bool u8, int wg_cap)
{
    int n = 0;
    constexpr int TPB = AMVS_WAVE * AMVS_WG_WAVES;
    hipError_t e = u8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pm_step_kernel<K, S, true, MODE_REFINE>, TPB,
                                                                     step_extra_lds<&pm_step_kernel<K, S, true, MODE_REFINE>>(wg_cap))
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pm_step_kernel<K, S, false, MODE_REFINE>, TPB,
                                                                     step_extra_lds<&pm_step_kernel<K, S, false, MODE_REFINE>>(wg_cap));
    return e == hipSuccess && n > 0 ? n * AMVS_WG_WAVES : 8;
}

// resident waves per CU of the sweep kernel (register-limited)
int step_waves_per_cu(int K, int S, bool u8, int wg_cap)
{
    switch (K) {
    case 3: AMVS_FOR_S(3, step_occupancy_ks, u8, wg_cap)
    case 5: AMVS_FOR_S(5, step_occupancy_ks, u8, wg_cap)
    case 7: AMVS_FOR_S(7, step_occupancy_ks, u8, wg_cap)
    case 9: AMVS_FOR_S(9, step_occupancy_ks, u8, wg_cap)
    case 11: AMVS_FOR_S(11, step_occupancy_ks, u8, wg_cap)
    case 13: AMVS_FOR_S(13, step_occupancy_ks, u8, wg_cap)
    case 15: AMVS_FOR_S(15, step_occupancy_ks, u8, wg_cap)
    case 17: AMVS_FOR_S(17, step_occupancy_ks, u8, wg_cap)
    case 19: AMVS_FOR_S(19, step_occupancy_ks, u8, wg_cap)
    case 21: AMVS_FOR_S(21, step_occupancy_ks, u8, wg_cap)
    case 23: AMVS_FOR_S(23, step_occupancy_ks, u8, wg_cap)
    case 25: AMVS_FOR_S(25, step_occupancy_ks, u8, wg_cap)
    case 27: AMVS_FOR_S(27, step_occupancy_ks, u8, wg_cap)
    case 29: AMVS_FOR_S(29, step_occupancy_ks, u8, wg_cap)
    default: return step_generic_waves_per_cu(K, S);
    }
}

bool patch_compiled(int K) { return K >= 3 && K <= 29 && (K & 1) == 1; }
bool patch_supported(int K) { return K >= 3 && K <= AMVS_MAX_PATCH && (K & 1) == 1; }
bool step_pair_supported(int K, int S) { return patch_compiled(K) && S >= 2 && S <= AMVS_KMAX_SRC && step_pair_supported_ks(K, S); }
int strip_out_width(int K) { return AMVS_WAVE - 2 * (K / 2); }

hipError_t launch_step(int K, int S, const StepArgs &a, hipStream_t st)
{
    if (!patch_compiled(K)) return launch_step_generic(K, S, a, st);
    if (a.fast) return launch_step_fast(K, S, a, st);
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y;
    switch (K) {
    case 3: AMVS_FOR_S(3, launch_step_ks, a, nblk, st)
    case 5: AMVS_FOR_S(5, launch_step_ks, a, nblk, st)
    case 7: AMVS_FOR_S(7, launch_step_ks, a, nblk, st)
    case 9: AMVS_FOR_S(9, launch_step_ks, a, nblk, st)
    case 11: AMVS_FOR_S(11, launch_step_ks, a, nblk, st)
    case 13: AMVS_FOR_S(13, launch_step_ks, a, nblk, st)
    case 15: AMVS_FOR_S(15, launch_step_ks, a, nblk, st)
    case 17: AMVS_FOR_S(17, launch_step_ks, a, nblk, st)
    case 19: AMVS_FOR_S(19, launch_step_ks, a, nblk, st)
    case 21: AMVS_FOR_S(21, launch_step_ks, a, nblk, st)
    case 23: AMVS_FOR_S(23, launch_step_ks, a, nblk, st)
    case 25: AMVS_FOR_S(25, launch_step_ks, a, nblk, st)
    case 27: AMVS_FOR_S(27, launch_step_ks, a, nblk, st)
    case 29: AMVS_FOR_S(29, launch_step_ks, a, nblk, st)
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_box_stats(int K, const float *images, long long img_stride, int H, int W,
                            int first_img, int n_img, float *mean_out, float *var_out,
                            hipStream_t st)
{
    if (!patch_compiled(K)) return launch_box_stats_generic(K, images, img_stride, H, W, first_img, n_img, mean_out, var_out, st);
    const int TH = 32;
    const int tiles_x = (W + strip_out_width(K) - 1) / strip_out_width(K);
    const int tiles_y = (H + TH - 1) / TH;
    const dim3 grid(n_img * tiles_x * tiles_y), blk(AMVS_WAVE);
    switch (K) {
    case 3:
        hipLaunchKernelGGL((box_stats_kernel<3>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 9:
        hipLaunchKernelGGL((box_stats_kernel<9>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 5:
        hipLaunchKernelGGL((box_stats_kernel<5>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 7:
        hipLaunchKernelGGL((box_stats_kernel<7>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 11:
        hipLaunchKernelGGL((box_stats_kernel<11>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 13:
        hipLaunchKernelGGL((box_stats_kernel<13>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 15:
        hipLaunchKernelGGL((box_stats_kernel<15>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 17:
        hipLaunchKernelGGL((box_stats_kernel<17>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 19:
        hipLaunchKernelGGL((box_stats_kernel<19>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 21:
        hipLaunchKernelGGL((box_stats_kernel<21>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 23:
        hipLaunchKernelGGL((box_stats_kernel<23>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 25:
        hipLaunchKernelGGL((box_stats_kernel<25>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 27:
        hipLaunchKernelGGL((box_stats_kernel<27>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    case 29:
        hipLaunchKernelGGL((box_stats_kernel<29>), grid, blk, 0, st, images, img_stride, H, W, TH,
                           tiles_x, tiles_y, first_img, mean_out, var_out);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_init(const Job *jobs, int n_jobs, long long HW, unsigned long long seed,
                       float log_scale, float log_min, float *depth, float *normal, float *cost,
                       hipStream_t st)
{
    const int bx = (int)((HW + 255) / 256 < 2048 ? (HW + 255) / 256 : 2048);
    hipLaunchKernelGGL(pm_init_kernel, dim3(bx, n_jobs), dim3(256), 0, st, jobs, HW, seed,
                       log_scale, log_min, depth, normal, cost);
    return hipGetLastError();
}

hipError_t launch_resolve_state(const Job *jobs, int n_jobs, long long HW, float *depth, float *nbuf0, const float *nbuf1,
                                float *depth_out, float *normal_out, long long out_slot0, hipStream_t st)
{
    if ((depth_out == nullptr) != (normal_out == nullptr)) return hipErrorInvalidValue;
    const int bx = (int)((HW + 255) / 256 < 2048 ? (HW + 255) / 256 : 2048);
    hipLaunchKernelGGL(resolve_state_kernel, dim3(bx, n_jobs), dim3(256), 0, st, jobs, HW, depth, nbuf0, nbuf1,
                       depth_out, normal_out, out_slot0);
    return hipGetLastError();
}

hipError_t launch_rng_fill(unsigned long long seed, unsigned view, unsigned draw, long long n,
                           float *u_out, float *n_out, hipStream_t st)
{
    const int bx = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(rng_fill_kernel, dim3(bx > 0 ? bx : 1), dim3(256), 0, st, seed, view, draw, n,
                       u_out, n_out);
    return hipGetLastError();
}

}  // namespace amvs

AMVS_CHECK_TU(kernels)
