// amvs_generic.hip -- the sweep step, the plane sweep and the reference statistics for ANY odd patch size
// up to AMVS_MAX_PATCH (the reference takes any patch_size: mvs_patchmatch.py:45, box kernel built from it
// at :396-397; dense_stereo.py:36, :325-341).  The kernels of amvs_kernels.hip / amvs_kernels_fast.hip /
// amvs_sweep_fast.hip are compiled for k = 3, 5, ..., 29 (rings in registers, unrolled window sums); every
// other odd k runs here, with the patch size a launch argument:
//   * the same strip shape (a wave owns 64 columns and walks its rows top to bottom, 64 - 2 (k/2) output
//     columns), one wave per workgroup;
//   * ALL vertical rings live in LDS, sized at launch: [1 + S][k][64] floats (ring 0 = the reference values),
//     plus the validity bits of the last k/2 + 1 rows;
//   * column sums top -> bottom and row sums right -> left as k - 1 `v_add_f32_dpp wave_shl:1` steps -- the
//     summation order of the compiled kernels and of the tests' CPU checker (whose k is
//     a run-time argument), hence the same bits;
//   * both arithmetic modes (FAST: precomposed projections, 8-bit code sums, precomputed reference statistics);
//   * the classic schedule only (no paired bands, no split schedule), modes selected at run time.
// It is a fallback: correct for every k, tuned for none.
#define AMVS_TU_ID 5
#include "amvs_exact_common.h"
#include "amvs_fast_common.h"

namespace amvs {

struct GenericConsts {
    int K;
    float inv_area;      // 1 / k^2                      (exact: mvs_patchmatch.py:397)
    float c1, c2;        // 1 / (k^2 255), 1 / (k^2 255^2)   (fast: sums in code units)
};

static GenericConsts generic_consts(int K)
{
    GenericConsts g;
    g.K = K;
    g.inv_area = 1.0f / (float)(K * K);
    g.c1 = (float)(1.0 / ((double)(K * K) * 255.0));
    g.c2 = (float)(1.0 / ((double)(K * K) * 65025.0));
    return g;
}

// dynamic LDS of one (single-wave) workgroup, in bytes
static unsigned generic_step_lds(int K, int S) { return 4u * (256u + (unsigned)(S + 1) * K * AMVS_WAVE + (unsigned)(K / 2 + 1) * AMVS_WAVE + 2u * AMVS_WAVE); }
static unsigned generic_sweep_lds(int K, int S)
{
    return 4u * (256u + (unsigned)(S + 1) * K * AMVS_WAVE + (unsigned)(K / 2 + 1) * AMVS_WAVE) + 2u * AMVS_SWEEP_MAX_TH * AMVS_WAVE;
}

// k x k window sums from the LDS rings: column sums top -> bottom (plain sum for v, FMA chains for v*v and
// r*v; for the exact arithmetic also r and r*r of the reference itself), then the row sums right -> left.
// `oldest` = ring slot of the window's top row.  acc[3 s + {0, 1, 2}] = sum v, sum v*v, sum r*v of source s;
// acc[3 S], acc[3 S + 1] = sum r, sum r*r (WITH_REF).
template <int S, bool WITH_REF>
AMVS_DEV void generic_window_sums(const float *ring, int K, int oldest, int lane, float (&acc)[3 * S + 2])
{
    float cs[3 * S + 2];
#pragma unroll
    for (int i = 0; i < 3 * S + 2; ++i) cs[i] = 0.0f;
    int slot = oldest;
    for (int i = 0; i < K; ++i) {
        const float rr = ring[slot * AMVS_WAVE + lane];
        if (i == 0) {
            if (WITH_REF) { cs[3 * S] = rr; cs[3 * S + 1] = rr * rr; }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float vv = ring[((s + 1) * K + slot) * AMVS_WAVE + lane];
                cs[3 * s] = vv; cs[3 * s + 1] = vv * vv; cs[3 * s + 2] = rr * vv;
            }
        } else {
            if (WITH_REF) { cs[3 * S] = cs[3 * S] + rr; cs[3 * S + 1] = __builtin_fmaf(rr, rr, cs[3 * S + 1]); }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float vv = ring[((s + 1) * K + slot) * AMVS_WAVE + lane];
                cs[3 * s] = cs[3 * s] + vv;
                cs[3 * s + 1] = __builtin_fmaf(vv, vv, cs[3 * s + 1]);
                cs[3 * s + 2] = __builtin_fmaf(rr, vv, cs[3 * s + 2]);
            }
        }
        slot = slot + 1 == K ? 0 : slot + 1;
    }
    constexpr int NA = WITH_REF ? 3 * S + 2 : 3 * S;
#pragma unroll
    for (int i = 0; i < 3 * S + 2; ++i) acc[i] = cs[i];
    for (int j = 1; j < K; ++j)
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] = wave_shl1(acc[i]) + cs[i];
}

// ------------------------------------------------------------------ sweep step ---
// One cost evaluation + select (_compute_patch_cost / _spatial_propagation / _random_refinement /
// _compute_confidence, mvs_patchmatch.py:323-534) for a run-time patch size; see pm_step_kernel
// (amvs_kernels.hip) and pm_step_fast_kernel (amvs_kernels_fast.hip), whose arithmetic, candidate / select
// logic and state handling (StepArgs::nbuf) this repeats.
template <int S, bool U8, bool FAST>
__global__ __launch_bounds__(AMVS_WAVE) void pm_step_generic_kernel(const StepArgs a, const GenericConsts gc)
{
    static_assert(!FAST || U8, "the fast arithmetic samples the packed 8-bit maps");
    extern __shared__ float smem[];
    const int K = gc.K, HALF = K / 2, OUTW = AMVS_WAVE - 2 * HALF;
    float *lut = smem;                                                      // [256] code -> gray (exact, packed maps)
    float *ring = smem + 256;                                               // [1 + S][K][64]
    uint32_t *okring = (uint32_t *)(ring + (S + 1) * K * AMVS_WAVE);        // [HALF + 1][64]
    constexpr int NQ = 2 * AMVS_WAVE;
    uint32_t *nq = okring + (HALF + 1) * AMVS_WAVE;                         // [NQ] winners waiting for their normal
    int q_head = 0, q_tail = 0;

    const int lane = threadIdx.x;
    if (U8 && !FAST) fill_gray_lut(lut, lane);
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    int job_id, ty, tx;
    strip_of(a, t, job_id, ty, tx);

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W, mode = a.mode;
    const long long HW = (long long)H * W;
    const float *__restrict__ ref = a.images + job->ref_img * a.img_stride;
    const GlobalU16 ref_pairs = U8 ? (GlobalU16)job->ref_pairs : (GlobalU16)a.pairs;
    const GlobalFloat2s ref_stats = (GlobalFloat2s)job->ref_stats;          // FAST only
    constexpr int PADW = U8 ? 2 * AMVS_PAIR_BORDER : 0;
    const float *__restrict__ d_in = a.d_in + job->slot * HW;
    float *__restrict__ d_out = a.d_out + job->slot * HW;
    float *cost_io = a.cost + job->slot * HW;
    float *nbuf0 = a.nbuf[0] + job->slot * HW * 3, *nbuf1 = a.nbuf[1] + job->slot * HW * 3;
    float *__restrict__ aux = a.aux + job->slot * HW;
    const StreamKey key = stream_key(a.seed, job->stream_view, a.draw);

    // validity window of the projection: patch bounds (mvs_patchmatch.py:362-363) or image bounds for the
    // confidence pass (:516-517)
    const SampleConsts sc = make_sample_consts(H, W, mode == MODE_CONF ? 0.0f : (float)HALF,
                                               mode == MODE_CONF ? (float)W : (float)(W - HALF),
                                               mode == MODE_CONF ? (float)H : (float)(H - HALF));
    const FastConsts fc = make_fast_consts(H, W, mode == MODE_CONF ? 0 : HALF);

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    FastCol cols[S];
    if constexpr (FAST) fast_columns<S>(job, (float)xr, cols);
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int th_w = min(a.TH, H - y0);
    const int rows = th_w + 2 * HALF;
    const int oy = mode == MODE_PROP ? a.oy : 0, ox = mode == MODE_PROP ? a.ox : 0;
    const int noff = oy * W + ox;
    int wslot = 0, okslot = 0;

    for (int r = 0; r < rows; ++r) {
        const int yr = y0 - HALF + r;
        const bool live = col_in & ((unsigned)yr < (unsigned)H);
        const bool inb = live & ((unsigned)(yr + oy) < (unsigned)H) & ((unsigned)(xr + ox) < (unsigned)W);
        const int pix = yr * W + xr;
        const float d_raw = d_in[AMVS_IDX(inb ? pix + noff : 0, HW)];
        // (clamped index, no branch in the load path: dead lanes read element 0)
        float rv;
        if constexpr (U8) {
            const uint32_t code = ref_pairs[AMVS_IDX_LOHI(live ? pix + PADW * yr : 0, -((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER), (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER) - ((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER))] & 0xFFu;
            rv = FAST ? (float)code : lut[code];
        } else {
            rv = ref[AMVS_IDX(live ? pix : 0, HW)];
        }
        rv = live ? rv : 0.0f;

        // ---- candidate depth of this (possibly halo) pixel (mvs_patchmatch.py:430-436, :468-473) ----
        float dc = inb ? depth_untag(d_raw, a.depth_mask) : a.depth_min;
        const uint32_t h0 = pixel_hash((uint32_t)pix, key);
        {
            const float delta = (rng_uniform(h0) * 2.0f - 1.0f) * a.depth_range;
            float d = dc + delta;
            d = d < a.depth_min ? a.depth_min : d;
            d = d > a.depth_max ? a.depth_max : d;
            dc = mode == MODE_REFINE ? d : dc;
        }
        float v[S];
        unsigned okbits;
        if constexpr (FAST) {
            okbits = fast_sample_sources_checked<S, true>(job, fc, cols, (float)yr, dc, live, v);
        } else {
            JobCP jr = reload(job);
            const Vec3 Pw = backproject(jr->Kinv, jr->Rref, jr->tref, xr, yr, dc);
            okbits = sample_sources_checked<S, U8, true>(jr, a, sc, lut, Pw, live, v);
        }

        // ---- push into the vertical rings ----
        ring[wslot * AMVS_WAVE + lane] = rv;
#pragma unroll
        for (int s = 0; s < S; ++s) ring[((s + 1) * K + wslot) * AMVS_WAVE + lane] = v[s];
        okring[okslot * AMVS_WAVE + lane] = okbits;
        wslot = wslot + 1 == K ? 0 : wslot + 1;            // now the slot of the oldest row
        okslot = okslot + 1 == HALF + 1 ? 0 : okslot + 1;  // now the slot of row r - HALF
        __builtin_amdgcn_wave_barrier();
        if (r < 2 * HALF) continue;

        // ---- window sums, NCC, aggregate for centre row yc and centre column xc ----
        const int yc = yr - HALF;
        const int xc = xr + HALF;
        const bool outl = (lane < OUTW) & (xc < W);
        const int pc = AMVS_IDX(outl ? yc * W + xc : 0, HW);
        const float oldd_tagged = d_in[pc], oldc = cost_io[pc];
        const float oldd = depth_untag(oldd_tagged, a.depth_mask);
        const unsigned buf_c = depth_buffer(oldd_tagged);
        // lanes without an output pixel have pc = 0: they must not form pc + noff (it can lie before the map)
        const bool inb_c = ((unsigned)(yc + oy) < (unsigned)H) & ((unsigned)(xc + ox) < (unsigned)W);
        const int pn = AMVS_IDX((outl & inb_c) ? pc + noff : 0, HW);
        const float nb_tagged = mode == MODE_PROP ? d_in[pn] : 0.0f;
        // the centre pixel was sampled by lane + HALF, HALF rows ago; its hash is a pure function of the pixel
        const unsigned okc = okring[okslot * AMVS_WAVE + ((lane + HALF) & (AMVS_WAVE - 1))];
        const uint32_t h0c = pixel_hash((uint32_t)(yc * W + xc), key);

        float acc[3 * S + 2];
        generic_window_sums<S, !FAST>(ring, K, wslot, lane, acc);
        float m1, v1;
        if constexpr (FAST) {
            const f32x2_t mv1 = ref_stats[pc];
            m1 = mv1.x; v1 = mv1.y;
        } else {
            m1 = acc[3 * S] * gc.inv_area;
            v1 = acc[3 * S + 1] * gc.inv_area - m1 * m1;
        }

        float total = 0.0f, cnt = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // _ncc_cost (mvs_patchmatch.py:403-411)
            float cost;
            if constexpr (FAST) {
                const float mean2 = acc[3 * s] * gc.c1;
                const float var2 = __builtin_fmaf(-mean2, mean2, acc[3 * s + 1] * gc.c2);
                const float cov = __builtin_fmaf(-m1, mean2, acc[3 * s + 2] * gc.c2);
                float den, rden;
                ncc_denominator(v1 * var2, den, rden);
                cost = 1.0f - cov * rden;
            } else {
                const float mean2 = acc[3 * s] * gc.inv_area;
                const float var2 = acc[3 * s + 1] * gc.inv_area - mean2 * mean2;
                const float cov = acc[3 * s + 2] * gc.inv_area - m1 * mean2;
                float den, rden;
                ncc_denominator(v1 * var2, den, rden);
                cost = 1.0f - qdiv(cov, den, rden);
            }
            const bool oks = (okc >> s) & 1u;
            const float ncc2 = 1.0f - cost;                       // confidence: valid & (1 - cost > 0.6)   (:530-532)
            const bool hit = mode == MODE_CONF ? (oks & (ncc2 > 0.6f)) : oks;
            total = (hit & (mode != MODE_CONF)) ? total + cost : total;          // (:383-384)
            cnt = hit ? cnt + 1.0f : cnt;
        }
        const bool act = outl;
        if (mode == MODE_CONF) {
            if (act) aux[pc] = cnt;
            continue;
        }
        // average over valid sources, +inf when fewer than two (mvs_patchmatch.py:387-388)
        const float cden = cnt + 1e-8f;
        bool cden_ok = true;
        const float rc = rcp_t<true>(cden, cden_ok);
        const float avg = FAST ? total * rc : qdiv(total, cden, rc);
        const float newc = cnt >= 2.0f ? avg : __builtin_inff();
        if (mode == MODE_EVAL) {
            if (act) aux[pc] = newc;
            continue;
        }

        // ---- select (mvs_patchmatch.py:452-455 / :486-489) ----
        const bool better = act & (newc < oldc);
        if (better) cost_io[pc] = newc;
        if (mode == MODE_PROP) {
            const float nb_d = depth_untag(nb_tagged, a.depth_mask);
            if (act) d_out[pc] = better ? depth_tag(inb_c ? nb_d : a.depth_min, buf_c ^ 1u) : oldd_tagged;
            const unsigned long long won = __ballot(better);
            if (won != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(won >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)won, 0u));
                if (better) nq[(q_tail + rank) & (NQ - 1)] = propagate_entry(pc, buf_c, inb_c, depth_buffer(nb_tagged));
                q_tail += __popcll(won);
                if (q_tail - q_head >= AMVS_WAVE) {
                    propagate_normals(nq, q_head, AMVS_WAVE, lane, nbuf0, nbuf1, noff, (int)HW);
                    q_head += AMVS_WAVE;
                }
            }
        } else {
            const float delta = (rng_uniform(h0c) * 2.0f - 1.0f) * a.depth_range;
            float d = oldd + delta;
            d = d < a.depth_min ? a.depth_min : d;
            d = d > a.depth_max ? a.depth_max : d;
            if (act) d_out[pc] = better ? depth_tag(d, buf_c) : oldd_tagged;
            const unsigned long long won = __ballot(better);
            if (won != 0ull) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(won >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)won, 0u));
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

// ------------------------------------------------------------------ plane sweep --
// _plane_sweep_torch (dense_stereo.py:262-310) for a run-time patch size; structure (strips, plane chunks,
// 16-bit running-best keys in LDS, atomicMax merge) as plane_sweep_kernel / plane_sweep_fast_kernel.
template <int S, bool U8, bool FAST>
__global__ __launch_bounds__(AMVS_WAVE) void plane_sweep_generic_kernel(const SweepArgs a, const GenericConsts gc)
{
    static_assert(!FAST || U8, "the fast arithmetic samples the packed 8-bit maps");
    extern __shared__ float smem[];
    const int K = gc.K, HALF = K / 2, OUTW = AMVS_WAVE - 2 * HALF;
    float *lut = smem;
    float *ring = smem + 256;
    uint32_t *okring = (uint32_t *)(ring + (S + 1) * K * AMVS_WAVE);
    uint16_t *best = (uint16_t *)(okring + (HALF + 1) * AMVS_WAVE);        // [AMVS_SWEEP_MAX_TH][64]

    const int lane = threadIdx.x;
    if (U8 && !FAST) fill_gray_lut(lut, lane);
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
    const float *__restrict__ ref = a.images + job->ref_img * a.img_stride;
    const GlobalU16 ref_pairs = U8 ? (GlobalU16)job->ref_pairs : (GlobalU16)a.pairs;
    const GlobalFloat2s ref_stats = (GlobalFloat2s)job->ref_stats;
    constexpr int PADW = U8 ? 2 * AMVS_PAIR_BORDER : 0;
    const SampleConsts sc = make_sample_consts(H, W, -__builtin_inff(), __builtin_inff(), __builtin_inff());
    const FastConsts fc = make_fast_consts(H, W, 0);

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    FastCol cols[S];
    if constexpr (FAST) fast_columns<S>(job, (float)xr, cols);
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int trows = min(a.TH, H - y0);
    const int rows = trows + 2 * HALF;
    for (int i = 0; i < trows; ++i) best[i * AMVS_WAVE + lane] = (uint16_t)0;

    for (int d = d_begin; d < d_end; ++d) {
        const float depth = a.depths[AMVS_IDX(d, a.D)];
        int wslot = 0, okslot = 0;
        // (the rings need no clearing between planes: a window is read only after its 2 HALF + 1 rows were written)
        for (int r = 0; r < rows; ++r) {
            const int yr = y0 - HALF + r;
            const bool live = col_in & ((unsigned)yr < (unsigned)H);
            const int pix = yr * W + xr;
            // (clamped index, no branch in the load path: dead lanes read element 0)
            float rv;
            if constexpr (U8) {
                const uint32_t code = ref_pairs[AMVS_IDX_LOHI(live ? pix + PADW * yr : 0, -((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER), (long long)(H + 2 * AMVS_PAIR_BORDER) * (W + 2 * AMVS_PAIR_BORDER) - ((long long)AMVS_PAIR_BORDER * (W + 2 * AMVS_PAIR_BORDER) + AMVS_PAIR_BORDER))] & 0xFFu;
                rv = FAST ? (float)code : lut[code];
            } else {
                rv = ref[AMVS_IDX(live ? pix : 0, HW)];
            }
            rv = live ? rv : 0.0f;
            float v[S];
            unsigned okbits;
            if constexpr (FAST) {
                okbits = fast_sample_sources_checked<S, false>(job, fc, cols, (float)yr, depth, live, v);
            } else {
                JobCP jr = reload(job);
                const Vec3 Pw = backproject(jr->Kinv, jr->Rref, jr->tref, xr, yr, depth);
                okbits = sample_sources_checked<S, U8, true>(jr, a, sc, lut, Pw, live, v);
            }
            ring[wslot * AMVS_WAVE + lane] = rv;
#pragma unroll
            for (int s = 0; s < S; ++s) ring[((s + 1) * K + wslot) * AMVS_WAVE + lane] = v[s];
            okring[okslot * AMVS_WAVE + lane] = okbits;
            wslot = wslot + 1 == K ? 0 : wslot + 1;
            okslot = okslot + 1 == HALF + 1 ? 0 : okslot + 1;
            __builtin_amdgcn_wave_barrier();
            if (r < 2 * HALF) continue;

            const int yc = yr - HALF;
            const int xc = xr + HALF;
            const bool outl = (lane < OUTW) & (xc < W);
            const unsigned okc = okring[okslot * AMVS_WAVE + ((lane + HALF) & (AMVS_WAVE - 1))];
            float acc[3 * S + 2];
            generic_window_sums<S, !FAST>(ring, K, wslot, lane, acc);
            uint32_t votes = 0u;
            if constexpr (FAST) {
                const f32x2_t mv1 = ref_stats[AMVS_IDX(outl ? yc * W + xc : 0, HW)];
                const float m1 = mv1.x, v1 = mv1.y;
                if (a.thresh > 0.0f) {
                    // ncc > thresh (dense_stereo.py:303) as cov > 0, x >= 0, cov^2 > t^2 x (plane_sweep_fast_kernel)
                    const float t2 = a.thresh * a.thresh;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float mean2 = acc[3 * s] * gc.c1;
                        const float var2 = __builtin_fmaf(-mean2, mean2, acc[3 * s + 1] * gc.c2);
                        const float cov = __builtin_fmaf(-m1, mean2, acc[3 * s + 2] * gc.c2);
                        const float x = v1 * var2 + 1e-8f;
                        const bool vote = (cov > 0.0f) & (x >= 0.0f) & (cov * cov > t2 * x) & (((okc >> s) & 1u) != 0u);
                        votes += vote ? 1u : 0u;
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float mean2 = acc[3 * s] * gc.c1;
                        const float var2 = __builtin_fmaf(-mean2, mean2, acc[3 * s + 1] * gc.c2);
                        const float cov = __builtin_fmaf(-m1, mean2, acc[3 * s + 2] * gc.c2);
                        const float den = sqrt_rn(v1 * var2 + 1e-8f);
                        const float ncc = cov * rcp_rn(den);
                        if (ncc > a.thresh && ((okc >> s) & 1u)) votes += 1u;
                    }
                }
            } else {
                const float m1 = acc[3 * S] * gc.inv_area;
                const float v1 = acc[3 * S + 1] * gc.inv_area - m1 * m1;
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    // _compute_ncc_torch: eps inside the sqrt (dense_stereo.py:344-345)
                    const float mean2 = acc[3 * s] * gc.inv_area;
                    const float var2 = acc[3 * s + 1] * gc.inv_area - mean2 * mean2;
                    const float cov = acc[3 * s + 2] * gc.inv_area - m1 * mean2;
                    const float den = sqrt_rn(v1 * var2 + 1e-8f);
                    const float ncc = qdiv(cov, den, rcp_rn(den));
                    if (ncc > a.thresh && ((okc >> s) & 1u)) votes += 1u;   // :303-304
                }
            }
            if (outl) {
                const uint32_t keyv = (votes << 12) | (uint32_t)(AMVS_SWEEP_MAX_CHUNK - 1 - (d - d_begin));
                const uint32_t cur = best[(yc - y0) * AMVS_WAVE + lane];
                if (keyv > cur) best[(yc - y0) * AMVS_WAVE + lane] = (uint16_t)keyv;
            }
        }
    }

    unsigned *__restrict__ keys = a.keys + job->slot * HW;
    const int xc = xr + HALF;
    if (lane < OUTW && xc < W)
        for (int i = 0; i < trows; ++i) {
            const uint32_t b = best[i * AMVS_WAVE + lane];
            const uint32_t plane = (uint32_t)d_begin + (AMVS_SWEEP_MAX_CHUNK - 1 - (b & (AMVS_SWEEP_MAX_CHUNK - 1)));
            atomicMax(&keys[AMVS_IDX((y0 + i) * W + xc, HW)], ((b >> 12) << 16) | (65535u - plane));
        }
}

// ------------------------------------------------------------------ ref stats ----
// mean / variance under the k x k zero-padded box filter (mvs_patchmatch.py:403,406), run-time k: the
// column sums are formed from k loads per output row (a one-off per view; no ring), in box_stats_kernel's order.
__global__ __launch_bounds__(AMVS_WAVE) void box_stats_generic_kernel(const float *__restrict__ images, long long img_stride,
                                                                      int H, int W, int TH, int tiles_x, int tiles_y,
                                                                      int first_img, GenericConsts gc,
                                                                      float *__restrict__ mean_out, float *__restrict__ var_out)
{
    const int K = gc.K, HALF = K / 2, OUTW = AMVS_WAVE - 2 * HALF;
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
    const int trows = min(TH, H - y0);
    for (int i = 0; i < trows; ++i) {
        const int yc = y0 + i;
        float cr = 0.0f, crr = 0.0f;
        for (int k = 0; k < K; ++k) {
            const int yy = yc - HALF + k;
            const float rv = (col_in & ((unsigned)yy < (unsigned)H)) ? img[AMVS_IDX(yy * W + xr, (long long)H * W)] : 0.0f;
            if (k == 0) { cr = rv; crr = rv * rv; }
            else { cr = cr + rv; crr = __builtin_fmaf(rv, rv, crr); }
        }
        float br = cr, brr = crr;
        for (int j = 1; j < K; ++j) { br = wave_shl1(br) + cr; brr = wave_shl1(brr) + crr; }
        const int xc = xr + HALF;
        if (lane < OUTW && xc < W) {
            const float m = br * gc.inv_area;
            mo[AMVS_IDX(yc * W + xc, (long long)H * W)] = m;
            vo[AMVS_IDX(yc * W + xc, (long long)H * W)] = brr * gc.inv_area - m * m;
        }
    }
}

// (mean1, var1) from the 8-bit codes for the fast arithmetic: exact integer window sums (k <= 31: k^2 255^2 < 2^26
// -- NOT below 2^24 for k > 15, so the squares' sum is converted from a 32-bit integer, which rounds once, as the
// CPU checker's (float)srr does)
__global__ __launch_bounds__(256) void fast_stats_generic_kernel(const uint16_t *__restrict__ pairs, int H, int W, GenericConsts gc,
                                                                 float2 *__restrict__ out)
{
    constexpr int B = AMVS_PAIR_BORDER;
    const int HALF = gc.K / 2;
    const int PW = W + 2 * B;
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        int sr = 0, srr = 0;
        for (int dy = -HALF; dy <= HALF; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int dx = -HALF; dx <= HALF; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                const int c = pairs[AMVS_IDX((long long)(yy + B) * PW + xx + B, (long long)(H + 2 * B) * PW)] & 0xFF;
                sr += c; srr += c * c;
            }
        }
        const float m1 = (float)sr * gc.c1;
        out[i] = make_float2(m1, __builtin_fmaf(-m1, m1, (float)srr * gc.c2));
    }
}

// ------------------------------------------------------------------ dispatch -----
template <int S>
static hipError_t launch_step_generic_s(int K, const StepArgs &a, hipStream_t st)
{
    const GenericConsts gc = generic_consts(K);
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y;
    const unsigned lds = generic_step_lds(K, S);
    if (a.fast) {
        if (!a.pairs) return hipErrorInvalidValue;
        hipLaunchKernelGGL((pm_step_generic_kernel<S, true, true>), dim3(nblk), dim3(AMVS_WAVE), lds, st, a, gc);
    } else if (a.pairs) {
        hipLaunchKernelGGL((pm_step_generic_kernel<S, true, false>), dim3(nblk), dim3(AMVS_WAVE), lds, st, a, gc);
    } else {
        hipLaunchKernelGGL((pm_step_generic_kernel<S, false, false>), dim3(nblk), dim3(AMVS_WAVE), lds, st, a, gc);
    }
    return hipGetLastError();
}

template <int S>
static hipError_t launch_sweep_generic_s(int K, const SweepArgs &a, hipStream_t st)
{
    const GenericConsts gc = generic_consts(K);
    const int nblk = a.n_jobs * a.tiles_x * a.tiles_y * a.n_chunks;
    const unsigned lds = generic_sweep_lds(K, S);
    if (a.fast) {
        if (!a.pairs) return hipErrorInvalidValue;
        hipLaunchKernelGGL((plane_sweep_generic_kernel<S, true, true>), dim3(nblk), dim3(AMVS_WAVE), lds, st, a, gc);
    } else if (a.pairs) {
        hipLaunchKernelGGL((plane_sweep_generic_kernel<S, true, false>), dim3(nblk), dim3(AMVS_WAVE), lds, st, a, gc);
    } else {
        hipLaunchKernelGGL((plane_sweep_generic_kernel<S, false, false>), dim3(nblk), dim3(AMVS_WAVE), lds, st, a, gc);
    }
    return hipGetLastError();
}

bool generic_patch_ok(int K) { return K >= 3 && K <= AMVS_MAX_PATCH && (K & 1) == 1; }

hipError_t launch_step_generic(int K, int S, const StepArgs &a, hipStream_t st)
{
    if (!generic_patch_ok(K) || a.paired || a.presampled) return hipErrorInvalidValue;
    switch (S) {
    case 2: return launch_step_generic_s<2>(K, a, st);
    case 3: return launch_step_generic_s<3>(K, a, st);
    case 4: return launch_step_generic_s<4>(K, a, st);
    case 5: return launch_step_generic_s<5>(K, a, st);
    case 6: return launch_step_generic_s<6>(K, a, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_sweep_generic(int K, int S, const SweepArgs &a, hipStream_t st)
{
    if (!generic_patch_ok(K)) return hipErrorInvalidValue;
    switch (S) {
    case 2: return launch_sweep_generic_s<2>(K, a, st);
    case 3: return launch_sweep_generic_s<3>(K, a, st);
    case 4: return launch_sweep_generic_s<4>(K, a, st);
    case 5: return launch_sweep_generic_s<5>(K, a, st);
    case 6: return launch_sweep_generic_s<6>(K, a, st);
    default: return hipErrorInvalidValue;
    }
}

// resident waves per CU of the generic step: LDS-limited (its rings), at most 16
int step_generic_waves_per_cu(int K, int S)
{
    const unsigned per = generic_step_lds(K, S);
    const int n = (int)(160u * 1024u / (per ? per : 1u));
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}

hipError_t launch_box_stats_generic(int K, const float *images, long long img_stride, int H, int W, int first_img, int n_img,
                                    float *mean_out, float *var_out, hipStream_t st)
{
    if (!generic_patch_ok(K)) return hipErrorInvalidValue;
    const int TH = 32;
    const int tiles_x = (W + strip_out_width(K) - 1) / strip_out_width(K);
    const int tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL(box_stats_generic_kernel, dim3(n_img * tiles_x * tiles_y), dim3(AMVS_WAVE), 0, st, images, img_stride, H, W,
                       TH, tiles_x, tiles_y, first_img, generic_consts(K), mean_out, var_out);
    return hipGetLastError();
}

hipError_t launch_fast_stats_generic(int K, const uint16_t *pairs_view, int H, int W, float2 *out, hipStream_t st)
{
    if (!generic_patch_ok(K)) return hipErrorInvalidValue;
    const long long n = (long long)H * W;
    const dim3 grid((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), blk(256);
    hipLaunchKernelGGL(fast_stats_generic_kernel, grid, blk, 0, st, pairs_view, H, W, generic_consts(K), out);
    return hipGetLastError();
}

}  // namespace amvs

AMVS_CHECK_TU(generic)
