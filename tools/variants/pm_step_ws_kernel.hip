// NOT BUILT -- kept as evidence for DESIGN.md section 5.  The wave-specialised sweep step: inside one
// 8-wave workgroup four SAMPLER waves (hypothesis, projection, gather, bilinear finish -> a ring of rows
// in LDS) feed four WINDOW waves (column sums from that ring, DPP row sums, NCC, selection), phases of 4
// rows, one barrier per phase.  Bit-identical to pm_step_fast_kernel and to the CPU checker (12 parity
// cases including 16 x 1080p), 69-119 VGPRs, 61 KB LDS per workgroup (two per CU), horizontal halo
// 238 / 232 instead of 64 / 58, half the columns in flight per CU -- and 22-27 % SLOWER: 82.2-87.0 ms per
// step against 65.4 (strip heights 16 ... 64; with the 4 rows of a phase gathered as one batch: 83.6-85.7).
// The roles wait for each other at every phase barrier.  This text goes between pm_step_fast_kernel and
// the plane sweep in csrc/amvs_kernels_fast.hip (it uses that file's helpers); the launcher dispatched on
// StepArgs::ws with tiles_x = ceil(W / (4 * (64 - 2 * (K / 2)))) and one workgroup of 512 threads per tile.
// ------------------------------------------------------------------ wave-specialised step ---
// The same sweep step as pm_step_fast_kernel -- same arithmetic, same order of every sum, bit-identical
// maps -- with the two halves of a row on DIFFERENT WAVES of one workgroup:
//   * WS_SW sampler waves own 64 adjacent columns each (WS_SW * 64 contiguous columns per workgroup):
//     hypothesis, projection into the S sources, gather, bilinear finish -> the encoded sample goes into
//     a ring of rows in LDS.  They do nothing else, so the CU's gather queue stays fed;
//   * WS_WW window waves read their 64 columns (58 outputs for k = 7, overlapping the next wave's by 6)
//     of the last k rows from that ring: column sums, DPP row sums, NCC, selection, stores.  No ring of
//     their own, no geometry;
//   * phases of BR rows: samplers produce rows [BR p, BR p + BR) while the window waves consume the rows
//     of phase p - 1; one barrier per phase; the ring holds k - 1 + 2 BR rows.
// Against the one-wave-does-all strips: the columns shared by adjacent window waves are sampled ONCE
// (horizontal halo 238 / 232 instead of 64 / 58), and a CU holds two workgroups = 464 output columns
// in flight instead of 16 strips = 928, which is what the L2 hit rate depends on.
#ifndef AMVS_WS_BLOCK_ROWS
#define AMVS_WS_BLOCK_ROWS 4
#endif
constexpr int WS_SW = 4, WS_WW = 4;

template <int K, int S> struct WsCfg {
    static constexpr int HALF = K / 2;
    static constexpr int OUT1 = AMVS_WAVE - 2 * HALF;        // outputs per window wave
    static constexpr int OUTW = WS_WW * OUT1;                // outputs per workgroup
    static constexpr int COLS = WS_SW * AMVS_WAVE;           // sample columns of a ring row
    static constexpr int BR = AMVS_WS_BLOCK_ROWS;
    static constexpr int RING = K - 1 + 2 * BR;
    static constexpr unsigned LDS = RING * S * COLS * 4u + WS_WW * 2u * AMVS_WAVE * 8u;
    static constexpr bool FITS = LDS <= 64u * 1024u && OUTW + 2 * HALF <= COLS;
};

// NR rows of one column through the S sources with all NR * S gathers in flight together (the pose rows
// are loaded once per source)
template <int S, int NR, bool LEAN>
AMVS_DEV void ws_sample_rows(JobCP job, const FastConsts &fc, float fx, const float (&fy)[NR], const float (&d)[NR],
                             const bool (&live)[NR], float (&v)[NR][S], unsigned (&okb)[NR], bool &ok)
{
    FastTap tg[NR][S];
    uint32_t raw[NR][S];
    float zlo = 1.0f, zhi = 1.0f;
#pragma unroll
    for (int q = 0; q < NR; ++q) okb[q] = 0u;
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
        for (int q = 0; q < NR; ++q) {
            bool valid;
            tg[q][s] = fast_geom<LEAN, true>(M, b, fc, fx, fy[q], d[q], valid, zlo, zhi);
            okb[q] |= valid ? (1u << s) : 0u;
            raw[q][s] = fast_load(img, tg[q][s].off);
        }
    }
    if constexpr (LEAN) ok = (zlo >= 0x1p-95f) & (zhi < 0x1p96f);
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
        for (int s = 0; s < S; ++s) v[q][s] = fast_finish(raw[q][s], tg[q][s], live[q]);
}

template <int K, int S, int MODE_T>
__global__ __launch_bounds__(AMVS_WAVE * (WS_SW + WS_WW)) void pm_step_ws_kernel(const StepArgs a)
{
    using C = WsCfg<K, S>;
    constexpr int HALF = C::HALF, OUT1 = C::OUT1, COLS = C::COLS, BR = C::BR, RING = C::RING;
    constexpr float C1 = (float)(1.0 / ((double)(K * K) * 255.0));
    constexpr float C2 = (float)(1.0 / ((double)(K * K) * 65025.0));
    constexpr int mode = MODE_T;
    static_assert(MODE_T == MODE_REFINE || MODE_T == MODE_PROP, "hot steps only");
    constexpr int NQ = 2 * AMVS_WAVE;
    __shared__ uint32_t ring[C::FITS ? RING * S * COLS : 1];
    __shared__ uint2 nq_all[WS_WW * NQ];

    const int lane = threadIdx.x & (AMVS_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / AMVS_WAVE);
    const int g = xcd_remap(blockIdx.x, gridDim.x);            // the grid is exact: one workgroup per tile
    int job_id, ty, gx;
    strip_of(a, g, job_id, ty, gx);

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W;
    const long long HW = (long long)H * W;
    constexpr int PADW = 2 * AMVS_PAIR_BORDER;
    const float *__restrict__ d_in = a.d_in + job->slot * HW;
    const StreamKey key = stream_key(a.seed, job->stream_view, a.draw);
    const int x0 = gx * C::OUTW;                               // first output column of the workgroup
    const int xs0 = x0 - HALF;                                 // first sampled column
    const int xs_end = min(x0 + C::OUTW + HALF, W);            // sampled columns of use end here
    const int y0 = ty * a.TH;
    const int rows = min(a.TH, H - y0) + 2 * HALF;
    const int n_ph = (rows + BR - 1) / BR + 1;
    const int oy = mode == MODE_PROP ? a.oy : 0, ox = mode == MODE_PROP ? a.ox : 0;
    const int noff = oy * W + ox;

    if (wv < WS_SW) {
        // ================================ sampler wave ================================
        const int col = AMVS_WAVE * wv + lane;
        const int xr = xs0 + col;
        if (xs0 + AMVS_WAVE * wv >= xs_end) return;            // none of its columns is read by a window wave
        const FastConsts fc = make_fast_consts(H, W, HALF);
        const float fx = (float)xr;
        const bool col_in = (unsigned)xr < (unsigned)W;
        for (int ph = 0; ph < n_ph; ++ph) {
            // the BR rows of the phase as one batch: BR * S gathers in flight per wave (rows past the end
            // of the strip are computed as dead rows and not stored)
            float fy[BR], dc[BR];
            bool live[BR];
#pragma unroll
            for (int q = 0; q < BR; ++q) {
                const int r = ph * BR + q;
                const int yr = y0 - HALF + r;
                live[q] = col_in & ((unsigned)yr < (unsigned)H) & (r < rows);
                const bool inb = live[q] & ((unsigned)(yr + oy) < (unsigned)H) & ((unsigned)(xr + ox) < (unsigned)W);
                const int pix = yr * W + xr;
                const float d_raw = d_in[inb ? pix + noff : 0];
                const uint32_t h0 = pixel_hash((uint32_t)pix, key);
                dc[q] = candidate_depth(a, mode, inb, d_raw, h0);
                fy[q] = (float)yr;
            }
            if (ph * BR < rows) {
                float v[BR][S];
                unsigned okb[BR];
                bool ok = true;
                ws_sample_rows<S, BR, true>(job, fc, fx, fy, dc, live, v, okb, ok);
                if (__builtin_expect(!__all(ok), 0)) ws_sample_rows<S, BR, false>(reload(job), fc, fx, fy, dc, live, v, okb, ok);
#pragma unroll
                for (int q = 0; q < BR; ++q) {
                    const int r = ph * BR + q;
                    if (r < rows) {
                        uint32_t *dst = ring + (r % RING) * (S * COLS) + col;
#pragma unroll
                        for (int s = 0; s < S; ++s) dst[s * COLS] = sample_encode(v[q][s], (okb[q] >> s) & 1u);
                    }
                }
            }
            __syncthreads();
        }
        return;
    }

    // ================================ window wave ================================
    const int w = wv - WS_SW;
    if (x0 + OUT1 * w >= W) return;                            // all of its outputs lie right of the image
    uint2 *nq = nq_all + w * NQ;
    int q_head = 0, q_tail = 0;
    const GlobalU16 ref_pairs = (GlobalU16)job->ref_pairs;
    const GlobalFloat2s ref_stats = (GlobalFloat2s)job->ref_stats;
    const float *__restrict__ n_in = a.n_in + job->slot * HW * 3;
    float *__restrict__ d_out = a.d_out + job->slot * HW;
    float *cost_io = a.cost + job->slot * HW;
    float *n_out = a.n_out + job->slot * HW * 3;
    const int cw = OUT1 * w + lane;                            // the lane's column inside a ring row
    const int xr = xs0 + cw;
    const bool col_in = (unsigned)xr < (unsigned)W;
    uint32_t rb[RefBytes<K>::NB];
    typename Hist<K, S>::T hist_ok = 0;
#pragma unroll
    for (int i = 0; i < RefBytes<K>::NB; ++i) rb[i] = 0u;

    for (int ph = 0; ph < n_ph; ++ph) {
        const int r_begin = (ph - 1) * BR, r_end = min(ph * BR, rows);
        for (int r = r_begin < 0 ? r_end : r_begin; r < r_end; ++r) {
            const int yr = y0 - HALF + r;
            const bool live = col_in & ((unsigned)yr < (unsigned)H);
            const int pix = yr * W + xr;
            const uint32_t rc_raw = ref_pairs[live ? pix + PADW * yr : 0];
            const uint32_t rcode = live ? (rc_raw & 0xFFu) : 0u;
            ref_bytes_push<K>(rb, rcode);
            // validity bits of this row's samples
            const uint32_t *newest = ring + (r % RING) * (S * COLS) + cw;
            unsigned okbits = 0u;
#pragma unroll
            for (int s = 0; s < S; ++s) okbits |= (newest[s * COLS] >> 31) ? 0u : (1u << s);
            hist_ok = (hist_ok >> S) | ((typename Hist<K, S>::T)okbits << (S * HALF));
            if (r < 2 * HALF) continue;

            const int yc = yr - HALF;
            const int xc = xr + HALF;
            const bool outl = (lane < OUT1) & (xc < W);
            const int pc = outl ? yc * W + xc : 0;
            const float oldd = d_in[pc], oldc = cost_io[pc];
            const f32x2_t mv1 = ref_stats[pc];
            const unsigned okc = (unsigned)__shfl_down((int)(unsigned)hist_ok, HALF);

            float rr[K];
#pragma unroll
            for (int i = 0; i < K; ++i) rr[i] = ref_bytes_get<K>(rb, i);
            // column sums top -> bottom from the shared ring (rows r-K+1 .. r), then the DPP row sums
            int slot[K];
            {
                int sl = (r - (K - 1)) % RING;
#pragma unroll
                for (int i = 0; i < K; ++i) { slot[i] = sl; sl = sl + 1 == RING ? 0 : sl + 1; }
            }
            float acc[3 * S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float vv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const uint32_t e = ring[slot[i] * (S * COLS) + s * COLS + cw];
                    vv[i] = col_in ? __uint_as_float(e & 0x7FFFFFFFu) : 0.0f;
                }
                float cv = vv[0];
                float cvv = vv[0] * vv[0];
                float crv = rr[0] * vv[0];
#pragma unroll
                for (int i = 1; i < K; ++i) {
                    cv = cv + vv[i];
                    cvv = __builtin_fmaf(vv[i], vv[i], cvv);
                    crv = __builtin_fmaf(rr[i], vv[i], crv);
                }
                acc[3 * s] = cv; acc[3 * s + 1] = cvv; acc[3 * s + 2] = crv;
            }
            {
                float cs[3 * S];
#pragma unroll
                for (int i = 0; i < 3 * S; ++i) cs[i] = acc[i];
#pragma unroll
                for (int j = 1; j < K; ++j)
#pragma unroll
                    for (int i = 0; i < 3 * S; ++i) acc[i] = wave_shl1(acc[i]) + cs[i];
            }
            const float m1 = mv1.x, v1 = mv1.y;
            float total = 0.0f, cnt = 0.0f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float mean2 = acc[3 * s] * C1;
                const float var2 = __builtin_fmaf(-mean2, mean2, acc[3 * s + 1] * C2);
                const float cov = __builtin_fmaf(-m1, mean2, acc[3 * s + 2] * C2);
                float den, rden;
                ncc_denominator(v1 * var2, den, rden);
                const float cost = 1.0f - cov * rden;
                const bool hit = (okc >> s) & 1u;
                total = hit ? total + cost : total;
                cnt = hit ? cnt + 1.0f : cnt;
            }
            const float cden = cnt + 1e-8f;
            bool cden_ok = true;
            const float avg = total * rcp_t<true>(cden, cden_ok);
            const float newc = cnt >= 2.0f ? avg : __builtin_inff();
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
        __syncthreads();
    }
    if (mode == MODE_REFINE) {
        while (q_tail - q_head > 0) {
            const int n = min(q_tail - q_head, AMVS_WAVE);
            refine_normals(nq, q_head, n, lane, n_out, a.normal_range);
            q_head += n;
        }
    }
}

