// NOT BUILT -- kept as evidence for DESIGN.md section 5.  Two waves per strip, each sampling half of the
// sources; the per-source costs of a row cross through LDS behind one workgroup barrier per row and the
// waves take turns adding all S in source order, selecting and storing.  Bit-identical to
// pm_step_fast_kernel (6 parity cases: k = 3 .. 11, S = 2, 4, 6), 57 VGPRs, 24.5 KB of LDS per 4-wave
// workgroup; a CU holds 8 strips instead of 16 (half the image rows in flight per XCD).  81.7 / 83.8 /
// 81.7 / 80.6 ms per step at automatic / 16 / 24 / 32-row strips against 65.4: slower.  This text goes
// before the plane sweep in csrc/amvs_kernels_fast.hip; the samplers took a source offset `s0`
// (fsrc[s0 + s]) and launch_step_fast dispatched on StepArgs::pair_split with (strips + 1) / 2 workgroups.
// ------------------------------------------------------------------ source-pair step ---
// The same sweep step with TWO WAVES PER STRIP, each sampling half of the sources: wave (strip, h)
// projects / gathers / sums sources [h S/2, (h+1) S/2) and computes their NCC costs; the per-source
// costs of a row (validity in the sign bit) cross through LDS after a workgroup barrier, and the
// wave whose turn it is (alternating rows) adds all S of them IN SOURCE ORDER, selects and stores --
// the same operations in the same order as pm_step_fast_kernel, so the same bits.  A workgroup of four
// waves is two adjacent strips; a CU's 16 waves hold 8 strips instead of 16, i.e. an XCD works on half
// as many image rows at once, which is what its L2 hit rate depends on.
template <int K, int S, int MODE_T>
__global__ __launch_bounds__(AMVS_WAVE * 4, 4) void pm_step_pair_kernel(const StepArgs a)
{
    static_assert(S % 2 == 0, "the sources are split in two halves");
    static_assert(MODE_T == MODE_REFINE || MODE_T == MODE_PROP, "hot steps only");
    constexpr int SH = S / 2;
    constexpr int HALF = K / 2;
    constexpr int OUTW = AMVS_WAVE - 2 * HALF;
    constexpr float C1 = (float)(1.0 / ((double)(K * K) * 255.0));
    constexpr float C2 = (float)(1.0 / ((double)(K * K) * 65025.0));
    constexpr int mode = MODE_T;
    constexpr int NQ = 2 * AMVS_WAVE;
    __shared__ float lring_all[4 * SH * K * AMVS_WAVE];
    __shared__ uint32_t xch_all[4 * 2 * (SH + 1) * AMVS_WAVE]; // [wave][row parity][costs of the half, validity bits][lane]
    __shared__ uint2 nq_all[4 * NQ];

    const int lane = threadIdx.x & (AMVS_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / AMVS_WAVE);
    const int half = wv & 1, partner = wv ^ 1;
    float *lring = lring_all + wv * (SH * K * AMVS_WAVE);
    uint32_t *xch_mine = xch_all + wv * (2 * (SH + 1) * AMVS_WAVE);
    const uint32_t *xch_other = xch_all + partner * (2 * (SH + 1) * AMVS_WAVE);
    uint2 *nq = nq_all + wv * NQ;
    int q_head = 0, q_tail = 0;
    const int tiles_per_job = a.tiles_x * a.tiles_y;
    const int t = xcd_remap(blockIdx.x, gridDim.x) * 2 + (wv >> 1);
    if (t >= a.n_jobs * tiles_per_job) return;                 // last workgroup only: both waves of the strip leave
    int job_id, ty, tx;
    strip_of(a, t, job_id, ty, tx);

    const JobCP job = (JobCP)(a.jobs + job_id);
    const int H = a.H, W = a.W;
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
    const FastConsts fc = make_fast_consts(H, W, HALF);

    const int xbase = tx * OUTW - HALF;
    const int y0 = ty * a.TH;
    const int xr = xbase + lane;
    const float fx = (float)xr;
    const int s0 = half * SH;
    FastCol cols[SH];
    fast_columns<SH>(job, fx, cols, s0);
    const bool col_in = (unsigned)xr < (unsigned)W;
    const int rows = min(a.TH, H - y0) + 2 * HALF;

    uint32_t rb[RefBytes<K>::NB];
    uint32_t hist_ok = 0;                                     // SH bits per row, HALF + 1 rows
#pragma unroll
    for (int i = 0; i < RefBytes<K>::NB; ++i) rb[i] = 0u;
    int wslot = 0;
    const int oy = mode == MODE_PROP ? a.oy : 0, ox = mode == MODE_PROP ? a.ox : 0;
    const int noff = oy * W + ox;

    for (int r = 0; r < rows; ++r) {
        const int yr = y0 - HALF + r;
        const bool live = col_in & ((unsigned)yr < (unsigned)H);
        const bool inb = live & ((unsigned)(yr + oy) < (unsigned)H) & ((unsigned)(xr + ox) < (unsigned)W);
        const int pix = yr * W + xr;
        const uint32_t rc_raw = ref_pairs[live ? pix + PADW * yr : 0];
        const uint32_t h0 = pixel_hash((uint32_t)pix, key);
        const float d_raw = d_in[inb ? pix + noff : 0];
        const float dc = candidate_depth(a, mode, inb, d_raw, h0);
        float v[SH];
        const unsigned okbits = fast_sample_sources_checked<SH, true>(job, fc, cols, (float)yr, dc, live, v, s0);
        const uint32_t rcode = live ? (rc_raw & 0xFFu) : 0u;
        ref_bytes_push<K>(rb, rcode);
#pragma unroll
        for (int s = 0; s < SH; ++s) lring[(s * K + wslot) * AMVS_WAVE + lane] = v[s];
        wslot = wslot + 1 == K ? 0 : wslot + 1;
        hist_ok = (hist_ok >> SH) | (okbits << (SH * HALF));
        if (r < 2 * HALF) continue;                            // (both waves of a strip skip the same rows)

        const int yc = yr - HALF;
        const int xc = xr + HALF;
        const bool outl = (lane < OUTW) & (xc < W);
        const int pc = outl ? yc * W + xc : 0;
        const f32x2_t mv1 = ref_stats[pc];
        const unsigned okc = (unsigned)__shfl_down((int)hist_ok, HALF);
        float rr[K];
#pragma unroll
        for (int i = 0; i < K; ++i) rr[i] = ref_bytes_get<K>(rb, i);
        // column sums top -> bottom, row sums right -> left (window_sums_fast, for this half's sources)
        int slot[K];
#pragma unroll
        for (int i = 0; i < K; ++i) slot[i] = wslot + i >= K ? wslot + i - K : wslot + i;
        float acc[3 * SH];
        {
            float cs[3 * SH];
#pragma unroll
            for (int s = 0; s < SH; ++s) {
                float vv[K];
#pragma unroll
                for (int i = 0; i < K; ++i) vv[i] = lring[(s * K + slot[i]) * AMVS_WAVE + lane];
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
#pragma unroll
            for (int i = 0; i < 3 * SH; ++i) acc[i] = cs[i];
#pragma unroll
            for (int j = 1; j < K; ++j)
#pragma unroll
                for (int i = 0; i < 3 * SH; ++i) acc[i] = wave_shl1(acc[i]) + cs[i];
        }
        const float m1 = mv1.x, v1 = mv1.y;
        // this half's per-source costs; they and the validity bits of the window centre cross to the
        // partner through LDS (double-buffered by row parity: one barrier per row suffices)
        float mine[SH];
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const float mean2 = acc[3 * s] * C1;
            const float var2 = __builtin_fmaf(-mean2, mean2, acc[3 * s + 1] * C2);
            const float cov = __builtin_fmaf(-m1, mean2, acc[3 * s + 2] * C2);
            float den, rden;
            ncc_denominator(v1 * var2, den, rden);
            mine[s] = 1.0f - cov * rden;
        }
        uint32_t *slot_x = xch_mine + (r & 1) * ((SH + 1) * AMVS_WAVE);
#pragma unroll
        for (int s = 0; s < SH; ++s) slot_x[s * AMVS_WAVE + lane] = __float_as_uint(mine[s]);
        slot_x[SH * AMVS_WAVE + lane] = okc;
        __syncthreads();
        if ((r & 1) != half) continue;                         // the partner aggregates this row

        // ---- all S costs in source order ----
        const uint32_t *other = xch_other + (r & 1) * ((SH + 1) * AMVS_WAVE);
        const unsigned okc_other = other[SH * AMVS_WAVE + lane];
        float cst[S];
        bool hitv[S];
        const bool first = half == 0;                          // wave-uniform: constant register indices below
#pragma unroll
        for (int s = 0; s < SH; ++s) {
            const float oc = __uint_as_float(other[s * AMVS_WAVE + lane]);
            const bool mh = (okc >> s) & 1u, oh = (okc_other >> s) & 1u;
            cst[s] = first ? mine[s] : oc;
            hitv[s] = first ? mh : oh;
            cst[SH + s] = first ? oc : mine[s];
            hitv[SH + s] = first ? oh : mh;
        }
        const float oldd = d_in[pc], oldc = cost_io[pc];
        float total = 0.0f, cnt = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            total = hitv[s] ? total + cst[s] : total;
            cnt = hitv[s] ? cnt + 1.0f : cnt;
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
    if (mode == MODE_REFINE) {
        while (q_tail - q_head > 0) {
            const int n = min(q_tail - q_head, AMVS_WAVE);
            refine_normals(nq, q_head, n, lane, n_out, a.normal_range);
            q_head += n;
        }
    }
}

