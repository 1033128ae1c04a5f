// amvs_capi.hip -- the C ABI of include/amvs.h: context, device memory, step scheduling.
//
// Host-side orchestration of PatchMatchMVS._patchmatch_cuda (mvs_patchmatch.py:225-321)
// and DenseStereoReconstructor._plane_sweep_torch (dense_stereo.py:222-316): all views of
// a scene are uploaded once and stay resident; a batch of reference views is swept
// together, one kernel launch per cost-evaluation step over the whole batch.
#include "../../include/amvs.h"
#include "amvs_kernels.h"
#include "amvs_pool.h"

#include <dlfcn.h>
#include <rccl/rccl.h>          // types only: the library is resolved at run time (amvs_comm_*)

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

static_assert(AMVS_MAX_SRC == AMVS_KMAX_SRC, "source-count limits out of sync");

#ifdef AMVS_CHECK_INDICES
// index-checked build (amvs_check.h): one device-side report per kernel translation unit
namespace amvs {
void check_fetch_kernels(unsigned long long out[4], bool reset);
void check_fetch_kernels_fast(unsigned long long out[4], bool reset);
void check_fetch_sweep_fast(unsigned long long out[4], bool reset);
void check_fetch_sweep_exact(unsigned long long out[4], bool reset);
void check_fetch_generic(unsigned long long out[4], bool reset);
void check_fetch_extended(unsigned long long out[4], bool reset);
void check_fetch_fusion(unsigned long long out[4], bool reset);
void check_fetch_knn(unsigned long long out[4], bool reset);
}  // namespace amvs
#endif

namespace {

std::string g_create_error;

// Sum of the violations all kernels have counted since the last reset and the record of the first one found
// (out[1] = translation unit << 32 | source line: 1 amvs_kernels, 2 amvs_kernels_fast, 3 amvs_sweep_fast,
// 4 amvs_sweep_exact, 5 amvs_generic, 6 amvs_extended, 7 amvs_fusion, 8 amvs_knn; out[2] = the index, out[3] = the
// extent it was compared with).  Zeros in the shipped build.
void index_report(uint64_t out[4], bool reset)
{
    out[0] = out[1] = out[2] = out[3] = 0;
#ifdef AMVS_CHECK_INDICES
    (void)hipDeviceSynchronize();
    void (*const fetch[])(unsigned long long[4], bool) = {
        amvs::check_fetch_kernels, amvs::check_fetch_kernels_fast, amvs::check_fetch_sweep_fast, amvs::check_fetch_sweep_exact,
        amvs::check_fetch_generic, amvs::check_fetch_extended, amvs::check_fetch_fusion, amvs::check_fetch_knn};
    for (auto f : fetch) {
        unsigned long long r[4] = {0, 0, 0, 0};
        f(r, reset);
        if (r[0] && !out[0]) { out[1] = r[1]; out[2] = r[2]; out[3] = r[3]; }
        out[0] += r[0];
    }
#else
    (void)reset;
#endif
}

struct Stats {
    float *mean = nullptr, *var = nullptr;
    std::vector<char> done;
};

struct FastStats {
    float2 *maps = nullptr;             // [n_views][H*W]
    std::vector<char> done;
};

}  // namespace

struct amvs_ctx {
    int device = 0, H = 0, W = 0, n_views = 0, n_cu = 256;
    long long stride = 0;   // floats between images (H*W rounded up + tail padding)
    float K[9], Kinv[9];
    std::vector<std::array<float, 9>> R;
    std::vector<std::array<float, 3>> t;
    std::vector<char> have;
    float *d_images = nullptr;
    // packed 8-bit row-pair maps (sampling fast path), valid while every uploaded view is
    // exactly code/255 (n_inexact == 0); otherwise the sweep samples the float32 maps
    uint16_t *d_pairs = nullptr;
    long long pstride = 0;              // ushorts between packed maps
    unsigned char *d_bgr = nullptr;      // [n_views][H*W*3] prepared colour images (amvs_set_view_bgr8), lazily allocated
    unsigned char *d_prep_src = nullptr; // staging of one uploaded source image + the resize tables (amvs_set_view_bgr8):
    size_t cap_prep_src = 0;             // kept across calls -- a hipMalloc / hipFree pair per view cost more than the copy
    int *d_prep_tab = nullptr;
    size_t cap_prep_tab = 0;
    std::vector<char> have_bgr;
    int *d_flag = nullptr;               // [n_views] 1 = the view did not quantise to 8 bits losslessly
    mutable std::vector<char> exact8;    // host copy of !d_flag, refreshed lazily (flags_dirty)
    mutable bool flags_dirty = false;
    bool force_f32 = false;             // amvs_set_sampling: A/B switch for tests
    int mode = AMVS_MODE_EXACT;         // arithmetic of the sweeps (amvs_set_mode)
    int default_band_major = 0;         // schedule of amvs_pm_params.schedule == 0 (view-major measured faster)
    int sweep_tile_rows = 0, sweep_chunk = 0;   // amvs_set_sweep_tuning (0 = automatic)
    int sweep_key8 = 1;                         // strips above 32 rows with 8-bit keys where the plane chunks allow it
    std::map<int, Stats> stats;
    std::map<int, FastStats> fstats;    // fast mode: (mean1, var1) maps per patch size
    int cap_slots = 0;
    float *d_depth[2] = {nullptr, nullptr}, *d_cost[2] = {nullptr, nullptr},
          *d_normal[2] = {nullptr, nullptr}, *d_aux = nullptr;
    amvs::Job *d_jobs = nullptr;
    int cap_jobs = 0;
    float *d_planes = nullptr;
    int cap_planes = 0;
    unsigned *d_keys = nullptr;          // plane-sweep running best, [slot][H*W]
    int cap_keys = 0;
    float *d_xcand_d = nullptr, *d_xcand_n = nullptr;           // extended mode: view-propagation candidates
    int *d_xsrc = nullptr;
    int cap_x = 0, cap_xsrc = 0;
    float *d_sweep_depth = nullptr, *d_sweep_conf = nullptr;   // maps of the last amvs_plane_sweep_batch
    int cap_sweep = 0, n_sweep = 0;
    double *d_cloud_pts = nullptr;       // result of the last amvs_fuse_filter
    unsigned char *d_cloud_rgb = nullptr;
    long long cloud_n = 0;
    // split schedule (amvs_pm_params.schedule == AMVS_SCHEDULE_SPLIT): sample maps, one stream per
    // view group, the token events that serialise the sampling kernels across the groups
    float *d_samples = nullptr;
    size_t cap_samples = 0;
    std::vector<hipStream_t> split_streams;  // [0] sampling kernels, [1] window kernels
    std::vector<hipEvent_t> split_events;    // [0] fork, [1 + g] sampled(g), [9 + g] windowed(g)
    int split_groups = 0, split_sample_rows = 0, split_sample_lds = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int last_tile_rows = 0, last_views_per_launch = 0;
    // state a continuation call (amvs_pm_params.first_iteration > 0) resumes: which depth buffer is
    // current, the next iteration, and a fingerprint of the batch it belongs to
    bool pm_resumable = false;
    int pm_cur = 0, pm_next_iteration = 0;
    uint64_t pm_key = 0;
    // native exchange (amvs_comm_*): RCCL resolved with dlopen, one communicator per context
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    // amvs_set_step_tuning: strip rows / resident workgroups per CU by [iteration][0 = propagation, 1 = refinement]
    // (0 = automatic); iterations beyond the table use its last row
    std::vector<int> tune_rows, tune_cap;
    // amvs_set_step_timing: an event behind every sweep launch of the last PatchMatch call
    bool step_timing = false;
    std::vector<hipEvent_t> ev_steps;
    int n_step_events = 0;
    std::vector<hipEvent_t> ev_groups;   // per view group of the last PatchMatch call: init / steps / confidence
    int timing_groups = 0;
    bool timing_pending = false;
    amvs_timing timing{};
    std::string err;
};

namespace {

int fail(amvs_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(c, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((c), AMVS_EHIP,                                                   \
                        std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)

// end of a synchronising entry point: in the index-checked build a recorded violation turns success into
// AMVS_EINDEX (the report stays until amvs_index_check resets it)
int checked(amvs_ctx *c, int rc)
{
#ifdef AMVS_CHECK_INDICES
    if (rc == AMVS_OK) {
        uint64_t r[4];
        index_report(r, false);
        if (r[0])
            return fail(c, AMVS_EINDEX, "index check: " + std::to_string(r[0]) + " out-of-range accesses; first in translation unit " +
                                            std::to_string(r[1] >> 32) + " line " + std::to_string(r[1] & 0xFFFFFFFFull) + ": index " +
                                            std::to_string((long long)r[2]) + ", extent " + std::to_string((long long)r[3]));
    }
#else
    (void)c;
#endif
    return rc;
}

int bind_device(amvs_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    return AMVS_OK;
}

int check_patch_src(amvs_ctx *c, int patch, int n_src)
{
    if (!amvs::patch_supported(patch))
        return fail(c, AMVS_EUNSUPPORTED,
                    "patch_size " + std::to_string(patch) + " unsupported (odd sizes from 3 to " + std::to_string(AMVS_MAX_PATCH) + ")");
    if (n_src < 2 || n_src > AMVS_MAX_SRC)
        return fail(c, AMVS_EUNSUPPORTED,
                    "n_src " + std::to_string(n_src) + " outside [2, " + std::to_string(AMVS_MAX_SRC) + "]");
    return AMVS_OK;
}

int ensure_slots(amvs_ctx *c, int n)
{
    if (n <= c->cap_slots) return AMVS_OK;
    const size_t hw = (size_t)c->H * c->W;
    for (int i = 0; i < 2; ++i) {
        if (c->d_depth[i]) (void)hipFree(c->d_depth[i]);
        if (c->d_cost[i]) (void)hipFree(c->d_cost[i]);
        if (c->d_normal[i]) (void)hipFree(c->d_normal[i]);
        c->d_depth[i] = c->d_cost[i] = c->d_normal[i] = nullptr;
    }
    if (c->d_aux) (void)hipFree(c->d_aux);
    c->d_aux = nullptr;
    c->cap_slots = 0;
    for (int i = 0; i < 2; ++i) {
        HIPCHK(c, hipMalloc(&c->d_depth[i], sizeof(float) * hw * n));
        if (i == 0) HIPCHK(c, hipMalloc(&c->d_cost[i], sizeof(float) * hw * n));   // cost is updated in place
        HIPCHK(c, hipMalloc(&c->d_normal[i], sizeof(float) * hw * n * 3));
    }
    HIPCHK(c, hipMalloc(&c->d_aux, sizeof(float) * hw * n));
    c->cap_slots = n;
    return AMVS_OK;
}

int ensure_jobs(amvs_ctx *c, int n)
{
    if (n <= c->cap_jobs) return AMVS_OK;
    if (c->d_jobs) (void)hipFree(c->d_jobs);
    c->d_jobs = nullptr; c->cap_jobs = 0;
    HIPCHK(c, hipMalloc(&c->d_jobs, sizeof(amvs::Job) * n));
    c->cap_jobs = n;
    return AMVS_OK;
}

// mean1 / var1 of every uploaded view for this patch size (computed once, kept resident)
int ensure_stats(amvs_ctx *c, int patch)
{
    Stats &s = c->stats[patch];
    if (!s.mean) {
        HIPCHK(c, hipMalloc(&s.mean, sizeof(float) * c->stride * c->n_views));
        HIPCHK(c, hipMalloc(&s.var, sizeof(float) * c->stride * c->n_views));
        s.done.assign(c->n_views, 0);
    }
    for (int v = 0; v < c->n_views; ++v) {
        if (!c->have[v] || s.done[v]) continue;
        HIPCHK(c, amvs::launch_box_stats(patch, c->d_images, c->stride, c->H, c->W, v, 1, s.mean,
                                         s.var, c->stream));
        s.done[v] = 1;
    }
    return AMVS_OK;
}

// fast mode: (mean1, var1) of every uploaded view for this patch size (exact integer window sums
// of the 8-bit codes), computed once and kept resident
int ensure_fast_stats(amvs_ctx *c, int patch)
{
    FastStats &s = c->fstats[patch];
    const size_t hw = (size_t)c->H * c->W;
    if (!s.maps) {
        HIPCHK(c, hipMalloc(&s.maps, sizeof(float2) * hw * c->n_views));
        s.done.assign(c->n_views, 0);
    }
    for (int v = 0; v < c->n_views; ++v) {
        if (!c->have[v] || s.done[v]) continue;
        HIPCHK(c, amvs::launch_fast_stats(patch, c->d_pairs + (long long)v * c->pstride, c->H, c->W,
                                          s.maps + (size_t)v * hw, c->stream));
        s.done[v] = 1;
    }
    return AMVS_OK;
}

// `fast_patch` > 0: also fill the fast-mode records (precomposed projections, ref statistics of
// that patch size)
int upload_jobs(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src, int fast_patch = 0,
                bool compose_only = false)
{
    if (n_ref <= 0 || !ref_ids || !src_ids) return fail(c, AMVS_EINVAL, "empty batch");
    const float2 *fmaps = nullptr;
    if (fast_patch > 0) {
        int rc = ensure_fast_stats(c, fast_patch);
        if (rc) return rc;
        fmaps = c->fstats[fast_patch].maps;
    }
    std::vector<amvs::Job> jobs(n_ref);
    for (int i = 0; i < n_ref; ++i) {
        amvs::Job &j = jobs[i];
        std::memset(&j, 0, sizeof(j));
        const int r = ref_ids[i];
        if (r < 0 || r >= c->n_views || !c->have[r])
            return fail(c, AMVS_EINVAL, "reference view " + std::to_string(r) + " not uploaded");
        std::memcpy(j.K, c->K, 36);
        std::memcpy(j.Kinv, c->Kinv, 36);
        std::memcpy(j.Rref, c->R[r].data(), 36);
        std::memcpy(j.tref, c->t[r].data(), 12);
        j.ref_img = r;
        j.ref_pairs = (unsigned long long)(uintptr_t)(c->d_pairs + (long long)r * c->pstride) +
                      (unsigned long long)(amvs::pair_map_origin(c->W) * amvs::pair_map_texel_bytes());
        j.ref_stats = fmaps ? (unsigned long long)(uintptr_t)(fmaps + (size_t)r * c->H * c->W) : 0ull;
        j.stream_view = (uint32_t)r;
        j.slot = i;
        j.n_src = n_src;
        for (int s = 0; s < n_src; ++s) {
            const int v = src_ids[i * n_src + s];
            if (v < 0 || v >= c->n_views || !c->have[v])
                return fail(c, AMVS_EINVAL, "source view " + std::to_string(v) + " not uploaded");
            j.src[s].pairs = (unsigned long long)(uintptr_t)(c->d_pairs + (long long)v * c->pstride);
            j.src[s].gray = (unsigned long long)(uintptr_t)(c->d_images + (long long)v * c->stride);
            std::memcpy(j.src[s].R, c->R[v].data(), 36);
            std::memcpy(j.src[s].t, c->t[v].data(), 12);
            if (fast_patch > 0 || compose_only) {
                amvs::fast_compose(c->K, c->R[r].data(), c->t[r].data(), c->R[v].data(), c->t[v].data(),
                                   j.fsrc[s].M, j.fsrc[s].b);
                j.fsrc[s].pairs = j.src[s].pairs;
            }
        }
    }
    int rc = ensure_jobs(c, n_ref);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, jobs.data(), sizeof(amvs::Job) * n_ref,
                             hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // `jobs` is a stack-lifetime staging buffer
    return AMVS_OK;
}

// the packed maps can be used when every uploaded view quantised losslessly
const uint16_t *usable_pairs(const amvs_ctx *c)
{
    if (c->force_f32) return nullptr;
    if (c->flags_dirty) {
        // the uploads only queue the losslessness test; its results are read here, once
        std::vector<int> flags(c->n_views, 1);
        if (hipMemcpyAsync(flags.data(), c->d_flag, sizeof(int) * c->n_views, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
            hipStreamSynchronize(c->stream) == hipSuccess) {
            for (int v = 0; v < c->n_views; ++v) c->exact8[v] = flags[v] ? 0 : 1;
            c->flags_dirty = false;
        } else {
            return nullptr;
        }
    }
    for (int v = 0; v < c->n_views; ++v)
        if (c->have[v] && !c->exact8[v]) return nullptr;
    return c->d_pairs;
}

// 1 when the sweeps of this call run in the fast arithmetic; fails when fast was asked for but
// some uploaded view is not 8-bit exact (the fast kernels sample the packed maps only)
int resolve_fast(amvs_ctx *c, int requested, int *fast)
{
    const int m = requested == AMVS_MODE_DEFAULT ? c->mode : requested;
    if (m != AMVS_MODE_EXACT && m != AMVS_MODE_FAST) return fail(c, AMVS_EINVAL, "unknown arithmetic mode");
    *fast = m == AMVS_MODE_FAST;
    if (*fast && !usable_pairs(c))
        return fail(c, AMVS_EUNSUPPORTED,
                    "fast mode needs 8-bit images (every uploaded view exactly code/255) and packed sampling");
    return AMVS_OK;
}

// Views swept together by one launch (the views of a batch are independent, mvs_patchmatch.py:104-123, so
// a batch can be swept in groups, each through the whole schedule).  Groups of FOUR views against the
// whole 16-view batch, measured on MI355X in round 3 (S=4, G px-hyp/s, automatic strip heights, one run
// per pair):
//     fast, 1080p:  k=3  52.0 / 49.3    k=5  46.2 / 45.0    k=7  42.1 / 41.6    k=9  36.4 / 38.3    k=11  33.8 / 35.4
//     fast, k=7:    2560x1440  36.4 / 38.8      3840x2160 (8 views: 4 / 8 per launch)  35.4 / 36.1
//     exact, 1080p, k=7:  37.0 / 38.2           fast, 1080p, k=7, 8-view rank shard:  42.2 / 40.0
// Four 1080p views are two generations of resident waves at 18-row strips and an XCD's L2 then serves the
// sources of 4 views instead of 16 (hit rate 0.81 against 0.69); a launch of two generations also has a
// relatively longer tail than one of six, which is what the wider images, the larger patches and the
// slower exact kernel lose more to than the L2 returns.  Hence groups of four exactly where they were
// measured to win, else the whole batch (capped so that the per-launch state stays in the low GB).
// Paired bands (round 3, later; bench.py, G px-hyp/s, groups of four / whole batch): 16 views 43.3 / 43.2,
// 8 views 44.2 / 42.3, k=5 47.2 / 45.9 -- the same rule holds.
int default_views_per_launch(const amvs_ctx *c, int n_ref, int patch, bool fast)
{
    if (fast && patch <= 7 && c->W <= 2048 && n_ref >= 8 && n_ref % 4 == 0) return 4;
    return n_ref < 32 ? n_ref : 32;
}

// Rows per wave strip.  A strip re-samples 2*(patch/2) halo rows, so tall strips waste less; two
// things pull the other way.  (1) The set of source rows the resident waves touch at once: measured
// on MI355X (S=4, 16 views 1080p; best strip height per patch size) k=3: 10-12 rows, k=5: 14-18, k=7:
// 20-26 (32: -3 %, 40+: -15 %), k=9: 24-32, k=11: 32-40, i.e. about 4k-4, and lower for wider images
// (8 views 4K, k=7: 12-16 rows best, 24: -6 %) -- that is the cap `tall`.  (2) Wave quantisation: the
// launch runs in generations of `slots` resident waves, and a last generation that is nearly empty
// costs as much as a full one.  Measured, k=7, 1080p, G px-hyp/s by (views per launch: strip rows):
// 16: 24 -> 40.2; 8: 24 / 20 / 16 / 12 -> 39.2 / 39.2 / 39.1 / 38.5; 4: 24 / 16 / 12 / 8 -> 36.7 / 38.4 /
// 39.9 / 37.5; 2: 24 / 16 / 12 / 8 -> 39.4 / 34.5 / 38.6 / 35.9; 1: 24 / 16 / 12 / 8 -> 26.5 / 29.0 / 35.3 / 29.1
// -- the winners are the heights whose wave count is just below a whole number of generations
// (or at least 3/4 of one).  Hence: among the heights up to `tall`, the best product of the last
// generation's fill and the strip's useful fraction rows / (rows + patch - 1).
// Paired bands (`paired`): a pair of bands samples 2 th + patch - 1 rows for 2 th output rows (an odd last
// band keeps the classic th + patch - 1), and the locality cap is lower -- measured, 16 views 1080p, k=7, ms
// per launch by band rows 12 / 14 / 16 / 18 / 20 / 22 / 24 / 27 / 30 / 36: 0.761 / 0.755 / 0.750 / 0.749 /
// 0.751 / 0.755 / 0.762 / 0.761 / 0.765 / 0.780 -> 3 * patch - 3; k=5 in groups of 4 views: 12 rows 47.2, 16 rows
// 44.0 G px-hyp/s; 2560x1440: 12 / 16 / 18 rows 40.5 / 41.1 / 41.3; 8 views 3840x2160: 38.5 / 38.4 / 38.2 (the
// reduction for wide images applies beyond 3072 columns only).
int pick_tile_rows(const amvs_ctx *c, int patch, int n_src, int n_jobs, int requested, int cap, bool fast = false,
                   int wg_cap = 0, bool paired = false)
{
    if (requested > 0) return requested < cap ? requested : cap;
    const int tiles_x = (c->W + amvs::strip_out_width(patch) - 1) / amvs::strip_out_width(patch);
    const long long slots = (long long)c->n_cu * (fast ? amvs::step_fast_waves_per_cu(patch, n_src, wg_cap)
                                                       : amvs::step_waves_per_cu(patch, n_src, usable_pairs(c) != nullptr, wg_cap));
    int tall = 4 * patch - 4 > 12 ? 4 * patch - 4 : 12;
    if (paired) tall = 3 * patch - 3 > 8 ? 3 * patch - 3 : 8;
    if (c->W > (paired ? 3072 : 2048)) tall = tall * 2 / 3 > 8 ? tall * 2 / 3 : 8;
    if (tall > cap) tall = cap;
    int best = tall < 8 ? tall : 8;
    double best_score = -1.0;
    for (int th = tall; th >= (tall < 8 ? tall : 8); th -= 2) {
        const double waves = (double)n_jobs * tiles_x * ((c->H + th - 1) / th);
        const double g = waves / (double)slots;
        const double fill = g > 1.0 ? g / std::ceil(g) : (g >= 0.75 ? 1.0 : g / 0.75);
        double useful = (double)th / (double)(th + patch - 1);
        if (paired) {
            const int bands = (c->H + th - 1) / th;
            const double sampled = (double)(bands / 2) * (2 * th + patch - 1) + (double)(bands % 2) * (th + patch - 1);
            useful = (double)c->H / sampled;
        }
        const double score = fill * useful;
        if (score > best_score + 1e-9) { best_score = score; best = th; }
    }
    return best;
}

// Band-major schedule (StepArgs::band_major): strip height such that ONE band of all views of the
// launch about fills an XCD's wave slots, i.e. every XCD walks its own band(s) of all views top to
// bottom in one generation of waves.  Few views: several adjacent bands per XCD.
int pick_band_rows(const amvs_ctx *c, int patch, int n_src, int n_jobs, bool fast)
{
    const int tiles_x = (c->W + amvs::strip_out_width(patch) - 1) / amvs::strip_out_width(patch);
    const long long slots_xcd = (long long)(c->n_cu / 8 > 0 ? c->n_cu / 8 : 1) *
                                (fast ? amvs::step_fast_waves_per_cu(patch, n_src)
                                      : amvs::step_waves_per_cu(patch, n_src, usable_pairs(c) != nullptr));
    const long long per_band = (long long)n_jobs * tiles_x;
    long long g = slots_xcd / per_band;             // bands resident together per XCD
    if (g < 1) g = 1;
    const long long bands = 8 * g;
    long long th = (c->H + bands - 1) / bands;
    const int min_th = 2 * patch > 8 ? 2 * patch : 8;
    if (th < min_th) th = min_th;
    return (int)th;
}

amvs::StepArgs base_args(const amvs_ctx *c, int patch, int n_jobs, int TH)
{
    amvs::StepArgs a{};
    a.H = c->H; a.W = c->W; a.TH = TH;
    a.tiles_x = (c->W + amvs::strip_out_width(patch) - 1) / amvs::strip_out_width(patch);
    a.tiles_y = (c->H + TH - 1) / TH;
    a.n_jobs = n_jobs;
    a.img_stride = c->stride;
    a.images = c->d_images;
    a.pairs = usable_pairs(c);
    a.pair_stride = c->pstride;
    a.jobs = c->d_jobs;
    a.aux = c->d_aux;
    return a;
}

// depth buffer `cur_d` is read and cur_d^1 written on every step; cost lives in d_cost[0] and is
// updated in place; normals: both buffers, the sign bit of the state depths names each pixel's
// current one (StepArgs::nbuf).  `tagged`: d_in is a state map (its depths carry that bit).
void set_io(amvs::StepArgs &a, const amvs_ctx *c, int cur_d, bool tagged = true)
{
    a.d_in = c->d_depth[cur_d];
    a.d_out = c->d_depth[cur_d ^ 1];
    a.cost = c->d_cost[0];
    a.nbuf[0] = c->d_normal[0];
    a.nbuf[1] = c->d_normal[1];
    a.depth_mask = tagged ? 0x7FFFFFFFu : 0xFFFFFFFFu;
}

void resolve_timing(amvs_ctx *c)
{
    if (!c->timing_pending) return;
    c->timing_pending = false;
    if (hipEventSynchronize(c->ev[3]) != hipSuccess) return;
    if (c->timing_groups > 0) {
        // PatchMatch: per view group [start | init | steps | confidence]
        double t_init = 0, t_sweep = 0, t_conf = 0;
        hipEvent_t prev = c->ev[0];
        for (int g = 0; g < c->timing_groups; ++g) {
            float a = 0.f, b = 0.f, d = 0.f;
            (void)hipEventElapsedTime(&a, prev, c->ev_groups[3 * g]);
            (void)hipEventElapsedTime(&b, c->ev_groups[3 * g], c->ev_groups[3 * g + 1]);
            (void)hipEventElapsedTime(&d, c->ev_groups[3 * g + 1], c->ev_groups[3 * g + 2]);
            t_init += a; t_sweep += b; t_conf += d;
            prev = c->ev_groups[3 * g + 2];
        }
        c->timing.init_ms = t_init; c->timing.sweep_ms = t_sweep; c->timing.confidence_ms = t_conf;
    } else {
        float ms0 = 0.f, ms1 = 0.f, ms2 = 0.f;
        (void)hipEventElapsedTime(&ms0, c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&ms1, c->ev[1], c->ev[2]);
        (void)hipEventElapsedTime(&ms2, c->ev[2], c->ev[3]);
        c->timing.init_ms = ms0; c->timing.sweep_ms = ms1; c->timing.confidence_ms = ms2;
    }
}

// Launch shape of one sweep step.  The gathers of an early iteration are scattered over the whole
// depth range (every pixel perturbs its depth by up to depth_range / 2^it), those of a late one are
// coherent, so the best strip height / residency differ by iteration; measured table: DESIGN.md
// section 5 (round 3).  An explicit amvs_pm_params.tile_rows, then amvs_set_step_tuning, override it.
struct StepShape { int rows, wg_cap; };

StepShape step_shape(const amvs_ctx *c, const amvs_pm_params *p, int n_src, int n_jobs, int iter, bool refine, bool fast,
                     int default_rows)
{
    StepShape s{default_rows, 0};
    const size_t idx = (size_t)2 * (size_t)iter + (refine ? 1 : 0);
    const size_t last = c->tune_rows.size() >= 2 ? c->tune_rows.size() - 2 + (refine ? 1 : 0) : 0;
    int rows = 0, cap = 0;
    if (!c->tune_rows.empty()) rows = c->tune_rows[idx < c->tune_rows.size() ? idx : last];
    if (!c->tune_cap.empty()) cap = c->tune_cap[idx < c->tune_cap.size() ? idx : last];
    if (cap > 0) s.wg_cap = cap;
    if (p->tile_rows > 0) return s;                       // the caller fixed the strip height
    if (rows > 0) s.rows = rows;
    else if (cap > 0) s.rows = pick_tile_rows(c, p->patch_size, n_src, n_jobs, 0, 1 << 20, fast, cap);
    return s;
}

// The sweep schedule of one PatchMatch call as a list of launches (the same for every view).
struct SchedStep {
    int mode, oy, ox;
    float depth_range, normal_range;
    unsigned draw;
    int flip_d;                          // depth buffers ping-ponged by the step (set_io)
    int iter;
};

std::vector<SchedStep> build_schedule(const amvs_pm_params *p)
{
    std::vector<SchedStep> v;
    for (int it = p->first_iteration; it < p->first_iteration + p->num_iterations; ++it) {
        // _spatial_propagation (mvs_patchmatch.py:415-457): even iterations pull from
        // (y+1,x) then (y,x+1), odd iterations from (y-1,x) then (y,x-1)
        const int sgn = (it % 2 == 0) ? 1 : -1;
        for (int k = 0; k < 2; ++k)
            v.push_back(SchedStep{amvs::MODE_PROP, k == 0 ? sgn : 0, k == 0 ? 0 : sgn, 0.f, 0.f, 0u, 1, it});
        // _random_refinement (mvs_patchmatch.py:459-491): ranges formed in double, cast once
        const float dr = (float)(((double)p->depth_max - (double)p->depth_min) * std::pow(0.5, it));
        const float nr = (float)(0.5 * std::pow(0.5, it));
        for (int s = 0; s < p->num_samples; ++s)
            v.push_back(SchedStep{amvs::MODE_REFINE, 0, 0, dr, nr, (unsigned)(1 + it * p->num_samples + s), 1, it});
    }
    return v;
}

void apply_step(amvs::StepArgs &a, const SchedStep &st)
{
    a.mode = st.mode; a.oy = st.oy; a.ox = st.ox;
    if (st.mode == amvs::MODE_REFINE) { a.depth_range = st.depth_range; a.normal_range = st.normal_range; a.draw = st.draw; }
}

// One stream: the batch in groups of `vpl` views, each group through the whole schedule with the
// fused kernel (sampling + window sums + selection in one launch).
int run_fused_schedule(amvs_ctx *c, int n_ref, int n_src, const amvs_pm_params *p, uint64_t seed, int fast,
                       const std::vector<SchedStep> &sched, void *conf_dev, int cur0, bool do_init, bool do_conf)
{
    const size_t hw = (size_t)c->H * c->W;
    // Views per launch: the views of a batch are independent, so the batch can be swept in groups of
    // `vpl` views, each group through the whole schedule (see default_views_per_launch).
    const int band_major = p->schedule == 0 ? c->default_band_major : (p->schedule == 2);
    // paired bands: asked for, or the automatic choice where they were measured faster (round 3, fast
    // arithmetic, one run per pair, G px-hyp/s paired / classic: k=7 1080p 43.3 / 41.9, k=5 47.3 / 46.2,
    // k=3 51.2 / 52.0 -- a one-row halo leaves nothing to save --, 2560x1440 41.6 / 39.3, 8 views of
    // 3840x2160 38.7 / 36.4, 32 views 43.0 / 41.7)
    // 9x9 / 11x11 (round 4: compiled, their exchange fits four workgroups per CU, bit-identical -- and measured NOT
    // faster: fast 11x11 36.2 paired at its best height (34 rows) against 36.4 classic, 9x9 38.2 against 39.2, exact
    // 35.4 / 35.7 and 30.7 / 32.3: at these sizes the k - 1 cross-lane adds of the window sums, not the sampled rows,
    // carry the launch): the automatic schedule keeps them classic
    const bool paired = (p->schedule == AMVS_SCHEDULE_PAIRED ||
                         (p->schedule == AMVS_SCHEDULE_AUTO && !band_major && p->patch_size >= 5 && p->patch_size <= 7)) &&
                        (fast ? amvs::step_fast_pair_supported(p->patch_size, n_src)
                              : (usable_pairs(c) != nullptr && amvs::step_pair_supported(p->patch_size, n_src)));
    int vpl = p->views_per_launch > 0 ? p->views_per_launch : default_views_per_launch(c, n_ref, p->patch_size, fast != 0);
    if (vpl > n_ref) vpl = n_ref;
    const int TH = p->tile_rows > 0 || !band_major
                       ? pick_tile_rows(c, p->patch_size, n_src, vpl, p->tile_rows, 1 << 20, fast != 0, 0, paired)
                       : pick_band_rows(c, p->patch_size, n_src, vpl, fast != 0);
    c->last_tile_rows = TH;
    // launch shape of every step: strip rows and resident workgroups per CU (step_shape)
    std::vector<StepShape> shapes(sched.size());
    for (size_t i = 0; i < sched.size(); ++i) {
        shapes[i] = band_major ? StepShape{TH, 0}
                               : step_shape(c, p, n_src, vpl, sched[i].iter, sched[i].mode == amvs::MODE_REFINE, fast != 0, TH);
        c->last_tile_rows = shapes[i].rows;
    }
    const int n_steps_timed = c->step_timing ? (int)sched.size() * ((n_ref + vpl - 1) / vpl) : 0;
    while ((int)c->ev_steps.size() < n_steps_timed + (n_ref + vpl - 1) / vpl) {
        hipEvent_t ev;
        HIPCHK(c, hipEventCreate(&ev));
        c->ev_steps.push_back(ev);
    }
    c->n_step_events = 0;
    const int n_groups = (n_ref + vpl - 1) / vpl;
    // events: [0] start, then per group: after init, after steps, after confidence
    while ((int)c->ev_groups.size() < 3 * n_groups) {
        hipEvent_t ev;
        HIPCHK(c, hipEventCreate(&ev));
        c->ev_groups.push_back(ev);
    }
    c->timing_groups = n_groups;
    int64_t launches = 0;
    for (int g = 0; g < n_groups; ++g) {
        const int j0 = g * vpl, nj = (n_ref - j0) < vpl ? (n_ref - j0) : vpl;
        amvs::StepArgs a = base_args(c, p->patch_size, nj, TH);
        a.fast = fast;
        a.band_major = band_major;
        a.paired = paired ? 1 : 0;
        a.jobs = c->d_jobs + j0;                   // slots stay global: job.slot = index in the batch
        a.depth_min = p->depth_min; a.depth_max = p->depth_max;
        a.seed = seed;
        int cur = cur0;
        // initialisation (mvs_patchmatch.py:268-284); a continuation call resumes the context's state
        if (do_init)
            HIPCHK(c, amvs::launch_init(a.jobs, nj, (long long)hw, seed, p->log_depth_scale, p->log_depth_min,
                                        c->d_depth[cur], c->d_normal[0], c->d_cost[0], c->stream));
        HIPCHK(c, hipEventRecord(c->ev_groups[3 * g], c->stream));
        if (c->step_timing) HIPCHK(c, hipEventRecord(c->ev_steps[c->n_step_events++], c->stream));
        for (size_t i = 0; i < sched.size(); ++i) {
            const SchedStep &st = sched[i];
            apply_step(a, st);
            a.TH = shapes[i].rows;
            a.tiles_y = (c->H + a.TH - 1) / a.TH;
            a.wg_cap = shapes[i].wg_cap;
            set_io(a, c, cur);
            HIPCHK(c, amvs::launch_step(p->patch_size, n_src, a, c->stream));
            if (c->step_timing) HIPCHK(c, hipEventRecord(c->ev_steps[c->n_step_events++], c->stream));
            cur ^= st.flip_d; ++launches;
        }
        a.TH = TH;
        a.tiles_y = (c->H + TH - 1) / TH;
        a.wg_cap = 0;
        HIPCHK(c, hipEventRecord(c->ev_groups[3 * g + 1], c->stream));
        // _compute_confidence (mvs_patchmatch.py:493-534), written straight into the output
        if (do_conf) {
            a.mode = amvs::MODE_CONF;
            set_io(a, c, cur);
            a.aux = conf_dev ? (float *)conf_dev : c->d_aux;
            HIPCHK(c, amvs::launch_step(p->patch_size, n_src, a, c->stream));
        }
        HIPCHK(c, hipEventRecord(c->ev_groups[3 * g + 2], c->stream));
    }
    c->timing.sweep_launches = launches;
    c->last_views_per_launch = vpl;
    return AMVS_OK;
}

// Split schedule (fast mode).  A sweep step is two kernels: the SAMPLING kernel visits every pixel
// once -- no strip halo -- and is bound by the CU's L1 line rate for scattered gathers; the WINDOW
// kernel streams the sample maps (coalesced) through the box sums, NCC and selection.  They stress
// different parts of the CU, so the batch is cut into G view groups and the two kernel kinds run on
// two streams (distinct priorities, so that they land on distinct hardware queues):
//     sampling stream:  sample(g0,n) sample(g1,n) sample(g0,n+1) ...   (never two of them at once)
//     window stream:    window(g,n) after sample(g,n); sample(g,n+1) after window(g,n)   (events)
// so window(g,n) runs under the sampling of the next group.  Views are independent
// (mvs_patchmatch.py:104-123), nothing else orders the groups.  Results are those of the fused kernel
// bit for bit (same arithmetic, same order of every sum).
int run_split_schedule(amvs_ctx *c, int n_ref, int n_src, const amvs_pm_params *p, uint64_t seed,
                       const std::vector<SchedStep> &sched, void *conf_dev)
{
    const size_t hw = (size_t)c->H * c->W;
    constexpr int MAXG = 8;
    int G = p->views_per_launch > 0 ? (n_ref + p->views_per_launch - 1) / p->views_per_launch
                                    : (c->split_groups > 0 ? c->split_groups : 2);
    if (G > n_ref) G = n_ref;
    if (G > MAXG) G = MAXG;
    const int vpl = (n_ref + G - 1) / G;
    G = (n_ref + vpl - 1) / vpl;
    const size_t need = (size_t)n_ref * n_src * hw;
    if (need > c->cap_samples) {
        if (c->d_samples) (void)hipFree(c->d_samples);
        c->d_samples = nullptr; c->cap_samples = 0;
        HIPCHK(c, hipMalloc(&c->d_samples, 4 * need));
        c->cap_samples = need;
    }
    if (c->split_streams.empty()) {
        int lo = 0, hi = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));        // lo = least urgent
        for (int i = 0; i < 2; ++i) {
            hipStream_t st;
            HIPCHK(c, hipStreamCreateWithPriority(&st, hipStreamNonBlocking, i == 0 ? lo : hi));
            c->split_streams.push_back(st);
        }
    }
    while ((int)c->split_events.size() < 1 + 2 * MAXG) {
        hipEvent_t ev;
        HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->split_events.push_back(ev);
    }
    while ((int)c->ev_groups.size() < 3) {
        hipEvent_t ev;
        HIPCHK(c, hipEventCreate(&ev));
        c->ev_groups.push_back(ev);
    }
    c->timing_groups = 1;
    c->n_step_events = 0;                       // (no per-launch events in this schedule: amvs_get_step_times returns none)
    // The window kernel gathers nothing, so its strips can be tall (vertical halo 1.09 at 64 rows);
    // the sampling kernel has no halo at all and wants SHORT strips (the resident waves then touch
    // fewer source rows at once).  Measured on MI355X, 16 views 1080p, k=7, S=4, 2 groups, ms per step:
    // window rows 32 / 48 / 64 / 96: 67.4 / 67.3 / 66.5 / 68.3; sampling rows 16 / 8 / 4 / 2: 70.8 / 68.4 /
    // 65.2-66.8 / 66.5.
    const int TH = p->tile_rows > 0 ? p->tile_rows : (c->H < 64 ? c->H : 64);
    c->last_tile_rows = TH;
    const int s_TH = c->split_sample_rows > 0 ? c->split_sample_rows : 4;
    hipStream_t s_smp = c->split_streams[0], s_win = c->split_streams[1];
    hipEvent_t ev_fork = c->split_events[0];
    hipEvent_t *ev_sampled = &c->split_events[1], *ev_windowed = &c->split_events[1 + MAXG];

    amvs::StepArgs all = base_args(c, p->patch_size, n_ref, TH);
    all.fast = 1;
    all.depth_min = p->depth_min; all.depth_max = p->depth_max;
    all.seed = seed;
    all.samples = c->d_samples;
    all.half = p->patch_size / 2;
    all.s_TH = s_TH;
    all.s_lds = c->split_sample_lds;
    all.s_tiles_x = (c->W + 63) / 64;
    all.s_tiles_y = (c->H + s_TH - 1) / s_TH;
    int cur = 0;
    HIPCHK(c, amvs::launch_init(all.jobs, n_ref, (long long)hw, seed, p->log_depth_scale, p->log_depth_min,
                                c->d_depth[cur], c->d_normal[0], c->d_cost[0], c->stream));
    HIPCHK(c, hipEventRecord(c->ev_groups[0], c->stream));
    HIPCHK(c, hipEventRecord(ev_fork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(s_smp, ev_fork, 0));
    HIPCHK(c, hipStreamWaitEvent(s_win, ev_fork, 0));
    bool first = true;
    for (const SchedStep &st : sched) {
        for (int g = 0; g < G; ++g) {
            const int j0 = g * vpl, nj = (n_ref - j0) < vpl ? (n_ref - j0) : vpl;
            amvs::StepArgs a = all;
            a.n_jobs = nj;
            a.jobs = c->d_jobs + j0;
            apply_step(a, st);
            set_io(a, c, cur);
            if (!first) HIPCHK(c, hipStreamWaitEvent(s_smp, ev_windowed[g], 0));
            HIPCHK(c, amvs::launch_sample_fast(n_src, a, s_smp));
            HIPCHK(c, hipEventRecord(ev_sampled[g], s_smp));
            HIPCHK(c, hipStreamWaitEvent(s_win, ev_sampled[g], 0));
            a.presampled = 1;
            HIPCHK(c, amvs::launch_step(p->patch_size, n_src, a, s_win));
            HIPCHK(c, hipEventRecord(ev_windowed[g], s_win));
        }
        first = false;
        cur ^= st.flip_d;
    }
    // the window stream is in order, so its last event covers every group
    if (!sched.empty()) HIPCHK(c, hipStreamWaitEvent(c->stream, ev_windowed[G - 1], 0));
    HIPCHK(c, hipEventRecord(c->ev_groups[1], c->stream));
    // _compute_confidence (mvs_patchmatch.py:493-534): one fused launch over the whole batch
    if ((p->flags & AMVS_PM_NO_CONFIDENCE) == 0) {
        all.mode = amvs::MODE_CONF;
        set_io(all, c, cur);
        all.aux = conf_dev ? (float *)conf_dev : c->d_aux;
        HIPCHK(c, amvs::launch_step(p->patch_size, n_src, all, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev_groups[2], c->stream));
    c->timing.sweep_launches = (int64_t)sched.size();     // one hypothesis of the whole batch each
    c->last_views_per_launch = n_ref;
    return AMVS_OK;
}

// single-view, single-step helper for the test entry points
struct OneStep {
    amvs_ctx *c;
    amvs::StepArgs a;
    int patch, n_src;
    size_t hw;
};

int one_step_begin(amvs_ctx *c, int ref, const int *src_ids, int n_src, int patch, OneStep &o)
{
    if (!c) return AMVS_EINVAL;
    int rc = bind_device(c);
    if (rc) return rc;
    if ((rc = check_patch_src(c, patch, n_src))) return rc;
    if ((rc = ensure_slots(c, 1))) return rc;
    int fast = 0;
    if ((rc = resolve_fast(c, AMVS_MODE_DEFAULT, &fast))) return rc;
    if ((rc = upload_jobs(c, 1, &ref, src_ids, n_src, fast ? patch : 0))) return rc;
    c->pm_resumable = false;                    // the single-step entry points overwrite the state of slot 0
    o.c = c; o.patch = patch; o.n_src = n_src; o.hw = (size_t)c->H * c->W;
    o.a = base_args(c, patch, 1, pick_tile_rows(c, patch, n_src, 1, 0, 64, fast != 0));
    o.a.fast = fast;
    o.a.mode = amvs::MODE_EVAL;
    set_io(o.a, c, 0, false);                   // caller-supplied depth maps: no tag to strip
    return AMVS_OK;
}

int upload_state(amvs_ctx *c, size_t hw, const float *depth, const float *normal, const float *cost)
{
    if (depth) HIPCHK(c, hipMemcpyAsync(c->d_depth[0], depth, 4 * hw, hipMemcpyHostToDevice, c->stream));
    if (normal) HIPCHK(c, hipMemcpyAsync(c->d_normal[0], normal, 12 * hw, hipMemcpyHostToDevice, c->stream));
    if (cost) HIPCHK(c, hipMemcpyAsync(c->d_cost[0], cost, 4 * hw, hipMemcpyHostToDevice, c->stream));
    return AMVS_OK;
}

// state of slot 0 after a single step (tagged depths in d_depth[dbuf]) -> plain host arrays
int download_state(amvs_ctx *c, size_t hw, int dbuf, float *depth, float *normal, float *cost)
{
    HIPCHK(c, amvs::launch_resolve_state(c->d_jobs, 1, (long long)hw, c->d_depth[dbuf], c->d_normal[0], c->d_normal[1],
                                         nullptr, nullptr, 0, c->stream));
    HIPCHK(c, hipMemcpyAsync(depth, c->d_depth[dbuf], 4 * hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(normal, c->d_normal[0], 12 * hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cost, c->d_cost[0], 4 * hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMVS_OK;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

#ifdef AMVS_CHECK_INDICES
const char *amvs_version(void) { return "amvs 0.1 (gfx950) +index-checks"; }
#else
const char *amvs_version(void) { return "amvs 0.1 (gfx950)"; }
#endif

int amvs_index_check(uint64_t report[4], int reset)
{
    if (!report) return AMVS_EINVAL;
    index_report(report, reset != 0);
    return AMVS_OK;
}

const char *amvs_last_error(const amvs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int amvs_create(int device_id, int H, int W, int n_views, const float K[9], const float K_inv[9],
                amvs_ctx **out)
{
    if (!out) return fail(nullptr, AMVS_EINVAL, "out is NULL");
    *out = nullptr;
    if (H < 2 || W < 2 || n_views < 1 || !K || !K_inv)
        return fail(nullptr, AMVS_EINVAL, "bad image size / view count / intrinsics");
    if ((long long)H * W > (1ll << 29)) return fail(nullptr, AMVS_EINVAL, "image too large (H*W must stay below 2^29: 32-bit pixel indices, 3 per normal)");
    if (H > (1 << 23) || W > (1 << 23)) return fail(nullptr, AMVS_EINVAL, "image side above 2^23 (24-bit row arithmetic)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, AMVS_EHIP, "no HIP device available (this backend has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, AMVS_EINVAL, "device_id out of range");
    amvs_ctx *c = new amvs_ctx();
    c->device = device_id; c->H = H; c->W = W; c->n_views = n_views;
    // rows of an image are W floats; one extra 256-byte line of tail padding per image
    c->stride = (((long long)H * W + 63) / 64) * 64 + 64;
    std::memcpy(c->K, K, 36);
    std::memcpy(c->Kinv, K_inv, 36);
    c->R.resize(n_views); c->t.resize(n_views); c->have.assign(n_views, 0);
    c->exact8.assign(n_views, 0);
    c->have_bgr.assign(n_views, 0);
    c->pstride = ((amvs::pair_map_elems(H, W) + 63) / 64) * 64 + 64;
    auto bail = [&](const char *what, hipError_t e) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        amvs_destroy(c);
        return fail(nullptr, AMVS_EHIP, m);
    };
    hipError_t e;
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && ncu > 0)
            c->n_cu = ncu;
    }
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    c->stream = c->own_stream;
    for (auto &ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc(&c->d_images, sizeof(float) * c->stride * n_views)) != hipSuccess)
        return bail("hipMalloc(images)", e);
    if ((e = hipMemsetAsync(c->d_images, 0, sizeof(float) * c->stride * n_views, c->stream)) != hipSuccess)
        return bail("hipMemset(images)", e);
    if ((e = hipMalloc(&c->d_pairs, sizeof(uint16_t) * c->pstride * n_views)) != hipSuccess)
        return bail("hipMalloc(pairs)", e);
    if ((e = hipMemsetAsync(c->d_pairs, 0, sizeof(uint16_t) * c->pstride * n_views, c->stream)) != hipSuccess)
        return bail("hipMemset(pairs)", e);
    if ((e = hipMalloc(&c->d_flag, sizeof(int) * n_views)) != hipSuccess) return bail("hipMalloc(flag)", e);
    if ((e = hipMemsetAsync(c->d_flag, 0, sizeof(int) * n_views, c->stream)) != hipSuccess) return bail("hipMemset(flag)", e);
    *out = c;
    return AMVS_OK;
}

int amvs_destroy(amvs_ctx *c)
{
    if (!c) return AMVS_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)amvs_comm_destroy(c);
    for (int i = 0; i < 2; ++i) {
        if (c->d_depth[i]) (void)hipFree(c->d_depth[i]);
        if (c->d_cost[i]) (void)hipFree(c->d_cost[i]);
        if (c->d_normal[i]) (void)hipFree(c->d_normal[i]);
    }
    if (c->d_aux) (void)hipFree(c->d_aux);
    if (c->d_jobs) (void)hipFree(c->d_jobs);
    if (c->d_planes) (void)hipFree(c->d_planes);
    if (c->d_keys) (void)hipFree(c->d_keys);
    if (c->d_xcand_d) (void)hipFree(c->d_xcand_d);
    if (c->d_xcand_n) (void)hipFree(c->d_xcand_n);
    if (c->d_xsrc) (void)hipFree(c->d_xsrc);
    if (c->d_samples) (void)hipFree(c->d_samples);
    for (auto &st : c->split_streams) (void)hipStreamDestroy(st);
    for (auto &ev : c->split_events) (void)hipEventDestroy(ev);
    if (c->d_sweep_depth) (void)hipFree(c->d_sweep_depth);
    if (c->d_sweep_conf) (void)hipFree(c->d_sweep_conf);
    if (c->d_cloud_pts) (void)hipFree(c->d_cloud_pts);
    if (c->d_cloud_rgb) (void)hipFree(c->d_cloud_rgb);
    amvs::pool_trim();                  // the post-steps' cached scratch blocks (amvs_pool.hip)
    if (c->d_images) (void)hipFree(c->d_images);
    if (c->d_pairs) (void)hipFree(c->d_pairs);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->d_bgr) (void)hipFree(c->d_bgr);
    if (c->d_prep_src) (void)hipFree(c->d_prep_src);
    if (c->d_prep_tab) (void)hipFree(c->d_prep_tab);
    for (auto &kv : c->stats) {
        if (kv.second.mean) (void)hipFree(kv.second.mean);
        if (kv.second.var) (void)hipFree(kv.second.var);
    }
    for (auto &kv : c->fstats)
        if (kv.second.maps) (void)hipFree(kv.second.maps);
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->ev_groups) (void)hipEventDestroy(ev);
    for (auto &ev : c->ev_steps) (void)hipEventDestroy(ev);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return AMVS_OK;
}

int amvs_set_stream(amvs_ctx *c, void *hip_stream)
{
    if (!c) return AMVS_EINVAL;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return AMVS_OK;
}

int amvs_sync(amvs_ctx *c)
{
    if (!c) return AMVS_EINVAL;
    int rc = bind_device(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_timing(c);
    return checked(c, AMVS_OK);
}

static int set_view_common(amvs_ctx *c, int view, const void *gray, const float R[9], const float t[3],
                           hipMemcpyKind kind)
{
    if (c) c->pm_resumable = false;             // new images / poses: a sweep cannot be continued across them
    if (!c) return AMVS_EINVAL;
    if (view < 0 || view >= c->n_views || !gray || !R || !t) return fail(c, AMVS_EINVAL, "bad view argument");
    int rc = bind_device(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_images + view * c->stride, gray, sizeof(float) * c->H * c->W, kind,
                             c->stream));
    // packed 8-bit map + losslessness test of this view (read back lazily, usable_pairs)
    HIPCHK(c, hipMemsetAsync(c->d_flag + view, 0, sizeof(int), c->stream));
    HIPCHK(c, amvs::launch_pack_pairs(c->d_images + view * c->stride, c->H, c->W,
                                      c->d_pairs + view * c->pstride, c->d_flag + view, c->stream));
    c->flags_dirty = true;
    c->have_bgr[view] = 0;
    // a host buffer is the caller's again on return; a device buffer is only ordered on the stream
    if (kind == hipMemcpyHostToDevice) HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(c->R[view].data(), R, 36);
    std::memcpy(c->t[view].data(), t, 12);
    c->have[view] = 1;
    for (auto &kv : c->stats) if (!kv.second.done.empty()) kv.second.done[view] = 0;
    for (auto &kv : c->fstats) if (!kv.second.done.empty()) kv.second.done[view] = 0;
    return AMVS_OK;
}

// OpenCV's linear-resize tables for one axis (resize.cpp, resizeGeneric_ setup, ksize = 2): float32
// arithmetic as there; cvRound = round half to even
static void resize_axis_tables(int n_dst, int n_src, std::vector<int> &ofs, std::vector<short> &w, bool clamp_ofs)
{
    const double scale = 1.0 / ((double)n_dst / (double)n_src);
    ofs.resize(n_dst); w.resize(2 * (size_t)n_dst);
    for (int d = 0; d < n_dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (clamp_ofs) {                       // columns: taps clamped into the image, weight zeroed
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
        }
        ofs[d] = s;                            // rows: the kernel clamps the two row indices, weights stay
        const float c0 = 1.f - f, c1 = f;
        w[2 * d] = (short)std::nearbyint(c0 * 2048.f);
        w[2 * d + 1] = (short)std::nearbyint(c1 * 2048.f);
    }
}

int amvs_set_view_bgr8(amvs_ctx *c, int view, const uint8_t *bgr_host, int src_h, int src_w, const float R[9],
                       const float t[3], uint8_t *scaled_bgr_out)
{
    if (c) c->pm_resumable = false;
    if (!c) return AMVS_EINVAL;
    if (view < 0 || view >= c->n_views || !bgr_host || !R || !t || src_h < 1 || src_w < 1)
        return fail(c, AMVS_EINVAL, "bad view argument");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t n_src = (size_t)src_h * src_w, n_dst = (size_t)c->H * c->W;
    std::vector<int> xofs, yofs;
    std::vector<short> ialpha, ibeta;
    resize_axis_tables(c->W, src_w, xofs, ialpha, true);
    resize_axis_tables(c->H, src_h, yofs, ibeta, false);
    // the prepared colour image stays on the device (the fusion reads it there: amvs_fuse_filter_views)
    if (!c->d_bgr) HIPCHK(c, hipMalloc(&c->d_bgr, 3 * n_dst * (size_t)c->n_views));
    unsigned char *d_scaled = c->d_bgr + 3 * n_dst * (size_t)view;
    const size_t tab_ints = (size_t)c->W + c->H, tab_shorts = 2 * ((size_t)c->W + c->H);
    hipError_t e = hipSuccess;
    if (3 * n_src > c->cap_prep_src) {
        if (c->d_prep_src) (void)hipFree(c->d_prep_src);
        c->d_prep_src = nullptr; c->cap_prep_src = 0;
        e = hipMalloc(&c->d_prep_src, 3 * n_src);
        if (e == hipSuccess) c->cap_prep_src = 3 * n_src;
    }
    if (e == hipSuccess && 4 * tab_ints + 2 * tab_shorts > c->cap_prep_tab) {
        if (c->d_prep_tab) (void)hipFree(c->d_prep_tab);
        c->d_prep_tab = nullptr; c->cap_prep_tab = 0;
        e = hipMalloc(&c->d_prep_tab, 4 * tab_ints + 2 * tab_shorts);
        if (e == hipSuccess) c->cap_prep_tab = 4 * tab_ints + 2 * tab_shorts;
    }
    unsigned char *d_src = c->d_prep_src;
    int *d_tab = c->d_prep_tab;
    int *d_xofs = d_tab, *d_yofs = d_tab ? d_tab + c->W : nullptr;
    short *d_ialpha = d_tab ? (short *)(d_tab + tab_ints) : nullptr, *d_ibeta = d_ialpha ? d_ialpha + 2 * c->W : nullptr;
    if (e == hipSuccess) e = hipMemcpyAsync(d_src, bgr_host, 3 * n_src, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_xofs, xofs.data(), 4 * (size_t)c->W, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_yofs, yofs.data(), 4 * (size_t)c->H, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ialpha, ialpha.data(), 4 * (size_t)c->W, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ibeta, ibeta.data(), 4 * (size_t)c->H, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = amvs::launch_prep_bgr8(d_src, src_h, src_w, c->H, c->W, d_xofs, d_ialpha, d_yofs, d_ibeta, d_scaled,
                                   c->d_images + view * c->stride, c->stream);
    if (e == hipSuccess && scaled_bgr_out)
        e = hipMemcpyAsync(scaled_bgr_out, d_scaled, 3 * n_dst, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_flag + view, 0, sizeof(int), c->stream);
    if (e == hipSuccess)
        e = amvs::launch_pack_pairs(c->d_images + view * c->stride, c->H, c->W, c->d_pairs + view * c->pstride,
                                    c->d_flag + view, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("set_view_bgr8: ") + hipGetErrorString(e));
    c->flags_dirty = true;
    c->have_bgr[view] = 1;
    std::memcpy(c->R[view].data(), R, 36);
    std::memcpy(c->t[view].data(), t, 12);
    c->have[view] = 1;
    for (auto &kv : c->stats) if (!kv.second.done.empty()) kv.second.done[view] = 0;
    for (auto &kv : c->fstats) if (!kv.second.done.empty()) kv.second.done[view] = 0;
    return AMVS_OK;
}

int amvs_set_view_colors(amvs_ctx *c, int view, const uint8_t *bgr_host)
{
    if (!c) return AMVS_EINVAL;
    if (view < 0 || view >= c->n_views || !bgr_host) return fail(c, AMVS_EINVAL, "bad view argument");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t n = (size_t)c->H * c->W;
    if (!c->d_bgr) HIPCHK(c, hipMalloc(&c->d_bgr, 3 * n * (size_t)c->n_views));
    HIPCHK(c, hipMemcpyAsync(c->d_bgr + 3 * n * (size_t)view, bgr_host, 3 * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_bgr[view] = 1;
    return AMVS_OK;
}

int amvs_set_view(amvs_ctx *c, int view, const float *gray_host, const float R[9], const float t[3])
{
    return set_view_common(c, view, gray_host, R, t, hipMemcpyHostToDevice);
}

int amvs_set_view_device(amvs_ctx *c, int view, const void *gray_device, const float R[9], const float t[3])
{
    return set_view_common(c, view, gray_device, R, t, hipMemcpyDeviceToDevice);
}

// The sweep of a batch: state in the context's buffers (final tagged depth in d_depth[*cur], normals
// in d_normal[0 / 1] as the tags say), confidence into `conf_dev` (NULL: the context's d_aux).
static int patchmatch_core(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                           const amvs_pm_params *p, uint64_t seed, void *conf_dev, int *cur_out)
{
    if (!p || !ref_ids || !src_ids || n_ref <= 0) return fail(c, AMVS_EINVAL, "NULL argument / empty batch");
    if (p->num_iterations < 0 || p->num_samples < 0) return fail(c, AMVS_EINVAL, "negative iteration count");
    // (the reference takes log(depth_min), mvs_patchmatch.py:269; the state maps use the depths' sign bit)
    if (!(p->depth_min > 0.0f) || !(p->depth_max >= p->depth_min)) return fail(c, AMVS_EINVAL, "need 0 < depth_min <= depth_max");
    int rc = bind_device(c);
    if (rc) return rc;
    if ((rc = check_patch_src(c, p->patch_size, n_src))) return rc;
    if ((rc = ensure_slots(c, n_ref))) return rc;
    int fast = 0;
    if ((rc = resolve_fast(c, p->mode, &fast))) return rc;
    if ((rc = upload_jobs(c, n_ref, ref_ids, src_ids, n_src, fast ? p->patch_size : 0))) return rc;

    const size_t hw = (size_t)c->H * c->W;
    if (p->schedule < 0 || p->schedule > AMVS_SCHEDULE_PAIRED) return fail(c, AMVS_EINVAL, "unknown schedule");
    if (p->schedule == AMVS_SCHEDULE_SPLIT && !fast)
        return fail(c, AMVS_EUNSUPPORTED, "the split schedule exists in fast mode only");
    if (p->schedule == AMVS_SCHEDULE_SPLIT && !amvs::patch_compiled(p->patch_size))
        return fail(c, AMVS_EUNSUPPORTED, "the split schedule exists for the compiled patch sizes (3 ... 29) only");
    // Continuation: iterations first_iteration .. of a sweep whose earlier iterations a previous call ran
    // on the same batch; the state maps stay in the context between the calls.
    if (p->first_iteration < 0) return fail(c, AMVS_EINVAL, "negative first_iteration");
    uint64_t key = 1469598103934665603ull;
    auto mix = [&key](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    mix((uint64_t)n_ref); mix((uint64_t)n_src); mix((uint64_t)p->patch_size); mix((uint64_t)p->num_samples); mix(seed);
    mix((uint64_t)fast); mix((uint64_t)__builtin_bit_cast(uint32_t, p->depth_min)); mix((uint64_t)__builtin_bit_cast(uint32_t, p->depth_max));
    for (int i = 0; i < n_ref; ++i) mix((uint64_t)(uint32_t)ref_ids[i]);
    for (int i = 0; i < n_ref * n_src; ++i) mix((uint64_t)(uint32_t)src_ids[i]);
    const bool resume = p->first_iteration > 0;
    if (resume) {
        if (p->schedule == AMVS_SCHEDULE_SPLIT) return fail(c, AMVS_EUNSUPPORTED, "the split schedule cannot resume a sweep");
        if (!c->pm_resumable || c->pm_key != key || c->pm_next_iteration != p->first_iteration)
            return fail(c, AMVS_EINVAL, "first_iteration > 0 continues the previous call: same batch, sources, patch, samples, "
                                        "seed and depth range, and first_iteration = the iterations already run");
    }
    c->pm_resumable = false;
    resolve_timing(c);
    c->timing = amvs_timing{};
    const std::vector<SchedStep> sched = build_schedule(p);
    const int cur0 = resume ? c->pm_cur : 0;
    int cur = cur0;
    for (const SchedStep &st : sched) cur ^= st.flip_d;                           // final depth buffer
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    if (p->schedule == AMVS_SCHEDULE_SPLIT) {
        if ((rc = run_split_schedule(c, n_ref, n_src, p, seed, sched, conf_dev))) return rc;
    } else {
        if ((rc = run_fused_schedule(c, n_ref, n_src, p, seed, fast, sched, conf_dev, cur0, !resume,
                                     (p->flags & AMVS_PM_NO_CONFIDENCE) == 0))) return rc;
        c->pm_resumable = true; c->pm_cur = cur; c->pm_key = key;
        c->pm_next_iteration = p->first_iteration + p->num_iterations;
    }
    // every group ran the same schedule, so the final depth buffer is the same for all
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    *cur_out = cur;
    c->timing.pixel_hypotheses =
        (int64_t)n_ref * (int64_t)hw * p->num_iterations * (2 + p->num_samples);
    c->timing_pending = true;
    return AMVS_OK;
}

int amvs_patchmatch_device(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                           const amvs_pm_params *p, uint64_t seed, void *depth_dev, void *normal_dev,
                           void *conf_dev)
{
    if (!c) return AMVS_EINVAL;
    if (!depth_dev || !normal_dev || !conf_dev) return fail(c, AMVS_EINVAL, "NULL output");
    int cur = 0;
    int rc = patchmatch_core(c, n_ref, ref_ids, src_ids, n_src, p, seed, conf_dev, &cur);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    // untagged depths and the current normal of every pixel straight into the caller's arrays
    HIPCHK(c, amvs::launch_resolve_state(c->d_jobs, n_ref, (long long)hw, c->d_depth[cur], c->d_normal[0], c->d_normal[1],
                                         (float *)depth_dev, (float *)normal_dev, 0, c->stream));
    return AMVS_OK;
}

// Host-buffer entry: the maps go from the context's own state buffers straight to the caller's
// arrays -- no per-call device allocation, one synchronisation.
int amvs_patchmatch(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                    const amvs_pm_params *p, uint64_t seed, float *depth_out, float *normal_out,
                    float *conf_out)
{
    if (!c) return AMVS_EINVAL;
    if (!depth_out || !normal_out || !conf_out || n_ref <= 0) return fail(c, AMVS_EINVAL, "NULL output");
    int cur = 0;
    int rc = patchmatch_core(c, n_ref, ref_ids, src_ids, n_src, p, seed, nullptr, &cur);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    HIPCHK(c, amvs::launch_resolve_state(c->d_jobs, n_ref, (long long)hw, c->d_depth[cur], c->d_normal[0], c->d_normal[1],
                                         nullptr, nullptr, 0, c->stream));
    HIPCHK(c, hipMemcpyAsync(depth_out, c->d_depth[cur], 4 * hw * n_ref, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(normal_out, c->d_normal[0], 12 * hw * n_ref, hipMemcpyDeviceToHost, c->stream));
    if ((p->flags & AMVS_PM_NO_CONFIDENCE) == 0)
        HIPCHK(c, hipMemcpyAsync(conf_out, c->d_aux, 4 * hw * n_ref, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_timing(c);
    return checked(c, AMVS_OK);
}

int amvs_get_timing(const amvs_ctx *c, amvs_timing *out)
{
    if (!c || !out) return AMVS_EINVAL;
    resolve_timing(const_cast<amvs_ctx *>(c));
    *out = c->timing;
    return AMVS_OK;
}

int amvs_sampling_mode(const amvs_ctx *c) { return c && usable_pairs(c) ? 1 : 0; }

int amvs_set_mode(amvs_ctx *c, int mode)
{
    if (!c) return AMVS_EINVAL;
    if (mode != AMVS_MODE_EXACT && mode != AMVS_MODE_FAST) return fail(c, AMVS_EINVAL, "unknown arithmetic mode");
    c->mode = mode;
    return AMVS_OK;
}

int amvs_get_mode(const amvs_ctx *c) { return c ? c->mode : AMVS_EINVAL; }

int amvs_set_sampling(amvs_ctx *c, int force_f32)
{
    if (!c) return AMVS_EINVAL;
    c->force_f32 = force_f32 != 0;
    return AMVS_OK;
}

int amvs_set_sweep_tuning(amvs_ctx *c, int tile_rows, int chunk)
{
    if (!c) return AMVS_EINVAL;
    if (tile_rows < 0 || tile_rows > AMVS_SWEEP_MAX_TH8 || chunk < 0)
        return fail(c, AMVS_EINVAL, "plane-sweep tuning out of range");
    c->sweep_tile_rows = tile_rows; c->sweep_chunk = chunk;
    return AMVS_OK;
}

int amvs_set_split_tuning(amvs_ctx *c, int groups, int sample_rows, int sample_lds_bytes)
{
    if (!c) return AMVS_EINVAL;
    if (groups < 0 || groups > 8 || sample_rows < 0 || sample_lds_bytes < 0 || sample_lds_bytes > 64 * 1024)
        return fail(c, AMVS_EINVAL, "split tuning out of range");
    c->split_groups = groups; c->split_sample_rows = sample_rows; c->split_sample_lds = sample_lds_bytes;
    return AMVS_OK;
}

int amvs_set_step_tuning(amvs_ctx *c, int n_iterations, const int32_t *tile_rows, const int32_t *wgs_per_cu)
{
    if (!c) return AMVS_EINVAL;
    if (n_iterations < 0 || (n_iterations > 0 && !tile_rows && !wgs_per_cu)) return fail(c, AMVS_EINVAL, "bad step tuning table");
    c->tune_rows.clear(); c->tune_cap.clear();
    for (int i = 0; i < 2 * n_iterations; ++i) {
        const int r = tile_rows ? tile_rows[i] : 0, w = wgs_per_cu ? wgs_per_cu[i] : 0;
        if (r < 0 || r > (1 << 20) || w < 0 || w > 8) {
            c->tune_rows.clear(); c->tune_cap.clear();
            return fail(c, AMVS_EINVAL, "step tuning: rows >= 0, workgroups per CU in 0..8");
        }
        c->tune_rows.push_back(r); c->tune_cap.push_back(w);
    }
    return AMVS_OK;
}

int amvs_set_step_timing(amvs_ctx *c, int enable)
{
    if (!c) return AMVS_EINVAL;
    c->step_timing = enable != 0;
    c->n_step_events = 0;
    return AMVS_OK;
}

int amvs_get_step_times(amvs_ctx *c, float *ms_out, int capacity, int *n_out)
{
    if (!c) return AMVS_EINVAL;
    if (!n_out || (capacity > 0 && !ms_out)) return fail(c, AMVS_EINVAL, "NULL argument");
    int rc = bind_device(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // events: per view group one start event followed by one per launch
    const int groups = c->timing_groups > 0 ? c->timing_groups : 1;
    const int per_group = c->n_step_events / groups;          // 1 + launches
    int n = 0;
    for (int g = 0; g < groups && per_group > 1; ++g)
        for (int i = 1; i < per_group; ++i, ++n) {
            if (n >= capacity) continue;
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev_steps[g * per_group + i - 1], c->ev_steps[g * per_group + i]));
            ms_out[n] = ms;
        }
    *n_out = n;
    return AMVS_OK;
}

int amvs_last_tile_rows(const amvs_ctx *c) { return c ? c->last_tile_rows : 0; }

int amvs_last_views_per_launch(const amvs_ctx *c) { return c ? c->last_views_per_launch : 0; }

int amvs_plane_sweep_device(amvs_ctx *c, int n_ref, const int *ref_ids, const int *nbr_ids, int n_nbr,
                            const float *depths, int D, int patch_size, float thresh, void *depth_dev,
                            void *conf_dev)
{
    if (!c) return AMVS_EINVAL;
    if (!depths || D < 1 || D > 65535 || !depth_dev || !conf_dev) return fail(c, AMVS_EINVAL, "bad plane list / outputs");
    int rc = bind_device(c);
    if (rc) return rc;
    if ((rc = check_patch_src(c, patch_size, n_nbr))) return rc;
    int fast = 0;
    if ((rc = resolve_fast(c, AMVS_MODE_DEFAULT, &fast))) return rc;
    if ((rc = upload_jobs(c, n_ref, ref_ids, nbr_ids, n_nbr, fast ? patch_size : 0))) return rc;
    if (D > c->cap_planes) {
        if (c->d_planes) (void)hipFree(c->d_planes);
        c->d_planes = nullptr; c->cap_planes = 0;
        HIPCHK(c, hipMalloc(&c->d_planes, sizeof(float) * D));
        c->cap_planes = D;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_planes, depths, sizeof(float) * D, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t hw = (size_t)c->H * c->W;
    if (n_ref > c->cap_keys) {
        if (c->d_keys) (void)hipFree(c->d_keys);
        c->d_keys = nullptr; c->cap_keys = 0;
        HIPCHK(c, hipMalloc(&c->d_keys, sizeof(unsigned) * hw * n_ref));
        c->cap_keys = n_ref;
    }
    amvs::SweepArgs a{};
    a.H = c->H; a.W = c->W;
    // tall strips (little halo re-sampling); the planes are chunked so that the launch still has
    // about four strips per resident wave slot.  A strip's running best lives in 4 KB of LDS: 16-bit keys for up
    // to AMVS_SWEEP_MAX_TH = 32 rows, or -- compiled patch sizes, chunks of at most 32 planes -- 8-bit keys for up
    // to 64 rows (SweepArgs::key8); the fewest bands of at most that many rows, evenly high.
    a.tiles_x = (c->W + amvs::strip_out_width(patch_size) - 1) / amvs::strip_out_width(patch_size);
    a.n_jobs = n_ref; a.D = D;
    auto shape = [&](int max_rows) {
        const int bands = (c->H + max_rows - 1) / max_rows;
        a.TH = (c->H + bands - 1) / bands;
        if (c->sweep_tile_rows >= 1 && c->sweep_tile_rows <= max_rows && c->sweep_tile_rows < c->H) a.TH = c->sweep_tile_rows;
        a.tiles_y = (c->H + a.TH - 1) / a.TH;
        const long long strips = (long long)n_ref * a.tiles_x * a.tiles_y;
        const long long slots = (long long)c->n_cu * 16;        // four waves per SIMD
        // chunks for ~8 waves per slot, evenly sized (measured on MI355X, config 2, strips of 60 rows, planes per
        // wave 2 / 3 / 4 / 5 / 6 / 8 / 13: exact 51.7 / 50.0 / 51.7 / 50.7 / 49.8 / 49.2 / 46.1, fast 73.7 / 73.8 / 76.8 /
        // 74.5 / 74.0 / 72.5 / 67.8 G px-hyp/s: many short waves fill the tail of the launch, uneven last chunks lose)
        long long want = (8 * slots + strips - 1) / strips;
        if (want < 1) want = 1;
        if (want > (D + 1) / 2) want = (D + 1) / 2;              // (at least two planes per wave: a wave's set-up)
        a.chunk = (int)((D + want - 1) / want);
        a.chunk = (int)((D + (D + a.chunk - 1) / a.chunk - 1) / ((D + a.chunk - 1) / a.chunk));   // even chunks
        if (c->sweep_chunk >= 1) a.chunk = c->sweep_chunk < D ? c->sweep_chunk : D;
        if (a.chunk > AMVS_SWEEP_MAX_CHUNK) a.chunk = AMVS_SWEEP_MAX_CHUNK;
        a.n_chunks = (D + a.chunk - 1) / a.chunk;
    };
    a.key8 = 0;
    if (amvs::patch_compiled(patch_size) && c->sweep_key8 != 0 && (c->sweep_tile_rows == 0 || c->sweep_tile_rows > AMVS_SWEEP_MAX_TH)) {
        shape(AMVS_SWEEP_MAX_TH8);
        a.key8 = a.chunk <= AMVS_SWEEP_MAX_CHUNK8 ? 1 : 0;
    }
    if (!a.key8) shape(AMVS_SWEEP_MAX_TH);
    c->last_tile_rows = a.TH;
    a.img_stride = c->stride;
    a.images = c->d_images;
    a.pairs = usable_pairs(c);
    a.pair_stride = c->pstride;
    a.fast = fast;
    a.depths = c->d_planes;
    a.thresh = thresh;
    if (!fast && amvs::patch_compiled(patch_size)) {
        // the exact sweep loads the reference views' window statistics (plane-invariant) from the resident maps
        if ((rc = ensure_stats(c, patch_size))) return rc;
        a.ref_mean = c->stats[patch_size].mean;
        a.ref_var = c->stats[patch_size].var;
    }
    a.depth_out = (float *)depth_dev; a.conf_out = (float *)conf_dev;
    a.keys = c->d_keys;
    a.jobs = c->d_jobs;
    resolve_timing(c);
    c->timing = amvs_timing{};
    c->timing_groups = 0;
    c->n_step_events = 0;
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_keys, 0, sizeof(unsigned) * hw * n_ref, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, amvs::launch_sweep(patch_size, n_nbr, a, c->stream));
    HIPCHK(c, amvs::launch_sweep_finish(a, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    c->timing.sweep_launches = 1;
    c->timing.pixel_hypotheses = (int64_t)n_ref * c->H * c->W * D;
    c->timing_pending = true;
    return AMVS_OK;
}

int amvs_plane_sweep(amvs_ctx *c, int ref, const int *nbr_ids, int n_nbr, const float *depths, int D,
                     int patch_size, float thresh, float *depth_out, float *conf_out)
{
    if (!c) return AMVS_EINVAL;
    if (!depth_out || !conf_out) return fail(c, AMVS_EINVAL, "NULL output");
    int rc = bind_device(c);
    if (rc) return rc;
    if ((rc = ensure_slots(c, 1))) return rc;
    c->pm_resumable = false;                    // the maps below land in slot 0 of the PatchMatch state
    const size_t hw = (size_t)c->H * c->W;
    rc = amvs_plane_sweep_device(c, 1, &ref, nbr_ids, n_nbr, depths, D, patch_size, thresh,
                                 c->d_depth[0], c->d_aux);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(depth_out, c->d_depth[0], 4 * hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(conf_out, c->d_aux, 4 * hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_timing(c);
    return checked(c, AMVS_OK);
}

int amvs_plane_sweep_batch(amvs_ctx *c, int n_ref, const int *ref_ids, const int *nbr_ids, int n_nbr,
                           const float *depths, int D, int patch_size, float thresh)
{
    if (!c) return AMVS_EINVAL;
    if (n_ref <= 0) return fail(c, AMVS_EINVAL, "empty batch");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    if (n_ref > c->cap_sweep) {
        if (c->d_sweep_depth) (void)hipFree(c->d_sweep_depth);
        if (c->d_sweep_conf) (void)hipFree(c->d_sweep_conf);
        c->d_sweep_depth = c->d_sweep_conf = nullptr; c->cap_sweep = 0;
        HIPCHK(c, hipMalloc(&c->d_sweep_depth, 4 * hw * n_ref));
        HIPCHK(c, hipMalloc(&c->d_sweep_conf, 4 * hw * n_ref));
        c->cap_sweep = n_ref;
    }
    c->n_sweep = 0;
    rc = amvs_plane_sweep_device(c, n_ref, ref_ids, nbr_ids, n_nbr, depths, D, patch_size, thresh,
                                 c->d_sweep_depth, c->d_sweep_conf);
    if (rc) return rc;
    c->n_sweep = n_ref;
    return AMVS_OK;
}

int amvs_fetch_sweep_maps(amvs_ctx *c, int first, int count, float *depth_out, float *conf_out)
{
    if (!c) return AMVS_EINVAL;
    if (first < 0 || count < 0 || first + count > c->n_sweep || !depth_out || !conf_out)
        return fail(c, AMVS_EINVAL, "sweep maps out of range (run amvs_plane_sweep_batch first)");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    HIPCHK(c, hipMemcpyAsync(depth_out, c->d_sweep_depth + first * hw, 4 * hw * count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(conf_out, c->d_sweep_conf + first * hw, 4 * hw * count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_timing(c);
    return AMVS_OK;
}

int amvs_stereo_backproject(amvs_ctx *c, int n_maps, const void *depth, const void *conf, int maps_where,
                            const uint8_t *colors_bgr_host, const double K_inv[9], const double *poses,
                            float min_confidence, int64_t *per_map_counts, int64_t *total)
{
    if (!c) return AMVS_EINVAL;
    if (n_maps < 1 || !colors_bgr_host || !K_inv || !poses || !total || maps_where < 0 || maps_where > 2)
        return fail(c, AMVS_EINVAL, "bad argument");
    if (maps_where == 2 ? n_maps != c->n_sweep : (!depth || !conf))
        return fail(c, AMVS_EINVAL, maps_where == 2 ? "n_maps differs from the resident plane-sweep batch" : "NULL maps");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W, n = hw * (size_t)n_maps;
    if (c->d_cloud_pts) (void)hipFree(c->d_cloud_pts);
    if (c->d_cloud_rgb) (void)hipFree(c->d_cloud_rgb);
    c->d_cloud_pts = nullptr; c->d_cloud_rgb = nullptr; c->cloud_n = 0;
    float *dd = nullptr, *dc = nullptr;
    unsigned char *dbgr = nullptr;
    hipError_t e = hipMalloc(&dbgr, 3 * n);
    if (e == hipSuccess) e = hipMemcpyAsync(dbgr, colors_bgr_host, 3 * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && maps_where == 0) {
        e = hipMalloc(&dd, 4 * n);
        if (e == hipSuccess) e = hipMalloc(&dc, 4 * n);
        if (e == hipSuccess) e = hipMemcpyAsync(dd, depth, 4 * n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dc, conf, 4 * n, hipMemcpyHostToDevice, c->stream);
    } else if (e == hipSuccess) {
        dd = maps_where == 2 ? c->d_sweep_depth : (float *)depth;
        dc = maps_where == 2 ? c->d_sweep_conf : (float *)conf;
    }
    long long tot = 0;
    std::vector<long long> per(n_maps, 0);
    if (e == hipSuccess)
        e = amvs::stereo_backproject(dd, dc, dbgr, n_maps, c->H, c->W, K_inv, poses, min_confidence, &c->d_cloud_pts,
                                     &c->d_cloud_rgb, &tot, per.data(), c->stream);
    (void)hipStreamSynchronize(c->stream);
    if (maps_where == 0) { if (dd) (void)hipFree(dd); if (dc) (void)hipFree(dc); }
    if (dbgr) (void)hipFree(dbgr);
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("stereo_backproject: ") + hipGetErrorString(e));
    c->cloud_n = tot;
    *total = tot;
    if (per_map_counts) for (int j = 0; j < n_maps; ++j) per_map_counts[j] = per[j];
    return AMVS_OK;
}

int amvs_stereo_backproject_views(amvs_ctx *c, int n_maps, const int *view_ids, const double K_inv[9], const double *poses,
                                  float min_confidence, int64_t *per_map_counts, int64_t *total)
{
    if (!c) return AMVS_EINVAL;
    if (n_maps < 1 || !view_ids || !K_inv || !poses || !total) return fail(c, AMVS_EINVAL, "bad argument");
    if (n_maps != c->n_sweep) return fail(c, AMVS_EINVAL, "n_maps differs from the resident plane-sweep batch");
    for (int j = 0; j < n_maps; ++j)
        if (view_ids[j] < 0 || view_ids[j] >= c->n_views || !c->have_bgr[view_ids[j]])
            return fail(c, AMVS_EINVAL, "view " + std::to_string(view_ids[j]) + " has no resident colour image (amvs_set_view_bgr8)");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    if (c->d_cloud_pts) (void)hipFree(c->d_cloud_pts);
    if (c->d_cloud_rgb) (void)hipFree(c->d_cloud_rgb);
    c->d_cloud_pts = nullptr; c->d_cloud_rgb = nullptr; c->cloud_n = 0;
    unsigned char *dbgr = nullptr;
    hipError_t e = hipMalloc(&dbgr, 3 * hw * (size_t)n_maps);
    for (int j = 0; j < n_maps && e == hipSuccess; ++j)
        e = hipMemcpyAsync(dbgr + 3 * hw * (size_t)j, c->d_bgr + 3 * hw * (size_t)view_ids[j], 3 * hw,
                           hipMemcpyDeviceToDevice, c->stream);
    long long tot = 0;
    std::vector<long long> per(n_maps, 0);
    if (e == hipSuccess)
        e = amvs::stereo_backproject(c->d_sweep_depth, c->d_sweep_conf, dbgr, n_maps, c->H, c->W, K_inv, poses, min_confidence,
                                     &c->d_cloud_pts, &c->d_cloud_rgb, &tot, per.data(), c->stream);
    (void)hipStreamSynchronize(c->stream);
    if (dbgr) (void)hipFree(dbgr);
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("stereo_backproject_views: ") + hipGetErrorString(e));
    c->cloud_n = tot;
    *total = tot;
    if (per_map_counts) for (int j = 0; j < n_maps; ++j) per_map_counts[j] = per[j];
    return AMVS_OK;
}

int amvs_cloud_knn_mean_distance(amvs_ctx *c, int k, double *mean_out)
{
    if (!c) return AMVS_EINVAL;
    if (!mean_out || c->cloud_n < 1) return fail(c, AMVS_EINVAL, "no resident cloud / NULL output");
    if (!amvs::knn_supported(k)) return fail(c, AMVS_EUNSUPPORTED, "k not compiled in (8, 10, 16, 20, 32)");
    if (c->cloud_n < k) return fail(c, AMVS_EINVAL, "fewer points than neighbours");
    int rc = bind_device(c);
    if (rc) return rc;
    HIPCHK(c, amvs::knn_mean_distance(c->d_cloud_pts, c->cloud_n, k, mean_out, c->stream, true));
    return AMVS_OK;
}

int amvs_cloud_voxel_downsample(amvs_ctx *c, const uint8_t *keep_mask, double voxel_size, int64_t *count)
{
    if (!c) return AMVS_EINVAL;
    if (!count || !(voxel_size > 0.0)) return fail(c, AMVS_EINVAL, "bad argument");
    int rc = bind_device(c);
    if (rc) return rc;
    double *p2 = nullptr;
    unsigned char *r2 = nullptr;
    long long m = 0;
    if (c->cloud_n > 0) {
        hipError_t e = amvs::voxel_downsample(c->d_cloud_pts, c->d_cloud_rgb, c->cloud_n, keep_mask, voxel_size, &p2, &r2,
                                              &m, c->stream);
        if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("voxel_downsample: ") + hipGetErrorString(e));
        (void)hipFree(c->d_cloud_pts); (void)hipFree(c->d_cloud_rgb);
    }
    c->d_cloud_pts = p2; c->d_cloud_rgb = r2; c->cloud_n = m;
    *count = m;
    return AMVS_OK;
}

int amvs_cloud_take(amvs_ctx *c, const int64_t *indices, int64_t m)
{
    if (!c) return AMVS_EINVAL;
    if (m < 0 || (m > 0 && !indices)) return fail(c, AMVS_EINVAL, "bad argument");
    if (c->cloud_n < 1 && m > 0) return fail(c, AMVS_EINVAL, "no resident cloud");
    int rc = bind_device(c);
    if (rc) return rc;
    static_assert(sizeof(long long) == sizeof(int64_t), "index width");
    double *p2 = nullptr;
    unsigned char *r2 = nullptr;
    if (m > 0) {
        hipError_t e = amvs::cloud_take(c->d_cloud_pts, c->d_cloud_rgb, c->cloud_n, (const long long *)indices, m, &p2, &r2, c->stream);
        if (e == hipErrorInvalidValue) return fail(c, AMVS_EINVAL, "cloud_take: index outside the resident cloud");
        if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("cloud_take: ") + hipGetErrorString(e));
    }
    if (c->d_cloud_pts) (void)hipFree(c->d_cloud_pts);
    if (c->d_cloud_rgb) (void)hipFree(c->d_cloud_rgb);
    c->d_cloud_pts = p2; c->d_cloud_rgb = r2; c->cloud_n = m;
    return AMVS_OK;
}

int amvs_knn_supported(int k) { return amvs::knn_supported(k) ? 1 : 0; }

// ---- extended mode (csrc/amvs_extended.hip) ----
static int xpm_begin(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src, const amvs_xpm_params *p,
                     void *depth_all, void *normal_all, void *cost_all, amvs::XArgs &a)
{
    if (!c) return AMVS_EINVAL;
    if (!p || !depth_all || !normal_all || !cost_all) return fail(c, AMVS_EINVAL, "NULL argument");
    if (p->patch_size < 3 || p->patch_size > 31 || (p->patch_size & 1) == 0 || p->window_stride < 1)
        return fail(c, AMVS_EINVAL, "extended mode: odd patch_size in 3..31 and window_stride >= 1");
    if (n_src < 2 || n_src > AMVS_MAX_SRC) return fail(c, AMVS_EUNSUPPORTED, "n_src outside [2, 6]");
    int rc = bind_device(c);
    if (rc) return rc;
    if ((rc = upload_jobs(c, n_ref, ref_ids, src_ids, n_src, 0, true))) return rc;
    const size_t hw = (size_t)c->H * c->W;
    if (n_ref > c->cap_x) {
        if (c->d_xcand_d) (void)hipFree(c->d_xcand_d);
        if (c->d_xcand_n) (void)hipFree(c->d_xcand_n);
        c->d_xcand_d = c->d_xcand_n = nullptr; c->cap_x = 0;
        HIPCHK(c, hipMalloc(&c->d_xcand_d, 4 * hw * n_ref));
        HIPCHK(c, hipMalloc(&c->d_xcand_n, 12 * hw * n_ref));
        c->cap_x = n_ref;
    }
    if (n_ref * n_src > c->cap_xsrc) {
        if (c->d_xsrc) (void)hipFree(c->d_xsrc);
        c->d_xsrc = nullptr; c->cap_xsrc = 0;
        HIPCHK(c, hipMalloc(&c->d_xsrc, sizeof(int) * n_ref * n_src));
        c->cap_xsrc = n_ref * n_src;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_xsrc, src_ids, sizeof(int) * n_ref * n_src, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    a = amvs::XArgs{};
    a.H = c->H; a.W = c->W; a.n_jobs = n_ref; a.n_src = n_src;
    a.jobs = c->d_jobs; a.images = c->d_images; a.img_stride = c->stride;
    a.pairs = usable_pairs(c); a.pair_stride = c->pstride;
    a.depth = (float *)depth_all; a.normal = (float *)normal_all; a.cost = (float *)cost_all;
    a.snap_depth = a.depth; a.snap_normal = a.normal;
    a.cand_d = c->d_xcand_d; a.cand_n = c->d_xcand_n; a.src_view = c->d_xsrc;
    a.patch = p->patch_size; a.stride = p->window_stride;
    a.depth_min = p->depth_min; a.depth_max = p->depth_max;
    return AMVS_OK;
}

int amvs_xpm_init(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src, const amvs_xpm_params *p,
                  uint64_t seed, void *depth_all, void *normal_all, void *cost_all)
{
    amvs::XArgs a;
    int rc = xpm_begin(c, n_ref, ref_ids, src_ids, n_src, p, depth_all, normal_all, cost_all, a);
    if (rc) return rc;
    a.seed = seed;
    HIPCHK(c, amvs::launch_xpm_init(a, p->log_depth_scale, p->log_depth_min, c->stream));
    return AMVS_OK;
}

// ranges / hypothesis set of one iteration (shared by amvs_xpm_iterate and amvs_xpm_step)
static void xpm_iteration_args(amvs::XArgs &a, const amvs_xpm_params *p, int iteration, uint64_t seed)
{
    a.seed = seed;
    const double shrink = std::pow(0.5, iteration);
    a.rel_range = (float)std::max(0.2 * shrink, 0.004);
    a.nrm_range = (float)std::max(0.4 * shrink, 0.01);
    a.n_refine = p->num_refine < 0 ? 0 : (p->num_refine > 6 ? 6 : p->num_refine);
    a.with_random = iteration < 2;
    a.with_view_cand = p->view_propagation ? 1 : 0;
}

static int xpm_run_phase(amvs_ctx *c, amvs::XArgs a, int n_src, int iteration, int phase, void *cost_out)
{
    if (phase == AMVS_XPM_PHASE_CANDIDATES) {
        // view propagation from a snapshot: the candidates of this iteration come from source
        // (iteration mod n_src) of every view, read from maps no call of this iteration has written
        if (a.with_view_cand) {
            a.colour = iteration % n_src;
            HIPCHK(c, amvs::launch_xpm_view_candidates(a, c->stream));
        }
    } else if (phase == AMVS_XPM_PHASE_RED || phase == AMVS_XPM_PHASE_BLACK) {
        a.colour = phase - AMVS_XPM_PHASE_RED;
        a.draw = (unsigned)(1 + 2 * iteration + a.colour);
        HIPCHK(c, amvs::launch_xpm_sweep(a, c->stream));
    } else {
        if (!cost_out) return fail(c, AMVS_EINVAL, "NULL output");
        HIPCHK(c, amvs::launch_xpm_eval(a, (float *)cost_out, c->stream));
    }
    return AMVS_OK;
}

static int xpm_phases(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src, const amvs_xpm_params *p,
                      int iteration, uint64_t seed, int first_phase, int last_phase, void *depth_all, void *normal_all,
                      void *cost_all, const void *snapshot_depth, const void *snapshot_normal, void *cost_out)
{
    amvs::XArgs a;
    int rc = xpm_begin(c, n_ref, ref_ids, src_ids, n_src, p, depth_all, normal_all, cost_all, a);
    if (rc) return rc;
    if (iteration < 0) return fail(c, AMVS_EINVAL, "negative iteration");
    if ((snapshot_depth == nullptr) != (snapshot_normal == nullptr)) return fail(c, AMVS_EINVAL, "snapshot: both maps or none");
    if (snapshot_depth) { a.snap_depth = (const float *)snapshot_depth; a.snap_normal = (const float *)snapshot_normal; }
    xpm_iteration_args(a, p, iteration, seed);
    for (int phase = first_phase; phase <= last_phase; ++phase)
        if ((rc = xpm_run_phase(c, a, n_src, iteration, phase, cost_out))) return rc;
    return AMVS_OK;
}

int amvs_xpm_step(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src, const amvs_xpm_params *p,
                  int iteration, uint64_t seed, int phase, void *depth_all, void *normal_all, void *cost_all,
                  const void *snapshot_depth, const void *snapshot_normal, void *cost_out)
{
    if (!c) return AMVS_EINVAL;
    if (phase < AMVS_XPM_PHASE_CANDIDATES || phase > AMVS_XPM_PHASE_EVAL) return fail(c, AMVS_EINVAL, "unknown phase");
    return xpm_phases(c, n_ref, ref_ids, src_ids, n_src, p, iteration, seed, phase, phase, depth_all, normal_all, cost_all,
                      snapshot_depth, snapshot_normal, cost_out);
}

int amvs_xpm_iterate(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src, const amvs_xpm_params *p,
                     int iteration, uint64_t seed, void *depth_all, void *normal_all, void *cost_all,
                     const void *snapshot_depth, const void *snapshot_normal)
{
    if (!c) return AMVS_EINVAL;
    return xpm_phases(c, n_ref, ref_ids, src_ids, n_src, p, iteration, seed, AMVS_XPM_PHASE_CANDIDATES, AMVS_XPM_PHASE_BLACK,
                      depth_all, normal_all, cost_all, snapshot_depth, snapshot_normal, nullptr);
}

int amvs_xpm_fetch_candidates(amvs_ctx *c, int n_ref, float *cand_depth_out, float *cand_normal_out)
{
    if (!c) return AMVS_EINVAL;
    if (!cand_depth_out || !cand_normal_out || n_ref < 1 || n_ref > c->cap_x) return fail(c, AMVS_EINVAL, "bad argument / no candidates");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    HIPCHK(c, hipMemcpyAsync(cand_depth_out, c->d_xcand_d, 4 * hw * n_ref, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(cand_normal_out, c->d_xcand_n, 12 * hw * n_ref, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMVS_OK;
}

int amvs_xpm_consistency(amvs_ctx *c, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                         const amvs_xpm_params *p, void *depth_all, void *normal_all, void *cost_all, void *conf_out)
{
    amvs::XArgs a;
    int rc = xpm_begin(c, n_ref, ref_ids, src_ids, n_src, p, depth_all, normal_all, cost_all, a);
    if (rc) return rc;
    if (!conf_out) return fail(c, AMVS_EINVAL, "NULL output");
    HIPCHK(c, amvs::launch_xpm_consistency(a, (float *)conf_out, p->consistency_px, p->consistency_rel, c->stream));
    return AMVS_OK;
}

int amvs_eval_cost(amvs_ctx *c, int ref, const int *src_ids, int n_src, int patch_size,
                   const float *depth_in, float *cost_out)
{
    OneStep o;
    int rc = one_step_begin(c, ref, src_ids, n_src, patch_size, o);
    if (rc) return rc;
    if (!depth_in || !cost_out) return fail(c, AMVS_EINVAL, "NULL argument");
    if ((rc = upload_state(c, o.hw, depth_in, nullptr, nullptr))) return rc;
    o.a.mode = amvs::MODE_EVAL;
    HIPCHK(c, amvs::launch_step(patch_size, n_src, o.a, c->stream));
    HIPCHK(c, hipMemcpyAsync(cost_out, c->d_aux, 4 * o.hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMVS_OK;
}

int amvs_sample_sources(amvs_ctx *c, int ref, const int *src_ids, int n_src, int patch_size, int bounds,
                        const float *depth_in, float *sampled_out, uint8_t *valid_out)
{
    OneStep o;
    int rc = one_step_begin(c, ref, src_ids, n_src, patch_size, o);
    if (rc) return rc;
    if (!depth_in || !sampled_out || !valid_out || bounds < 0 || bounds > 2) return fail(c, AMVS_EINVAL, "bad argument");
    if ((rc = upload_state(c, o.hw, depth_in, nullptr, nullptr))) return rc;
    float *ds = nullptr;
    unsigned char *dv = nullptr;
    HIPCHK(c, hipMalloc(&ds, 4 * o.hw * n_src));
    hipError_t e = hipMalloc(&dv, o.hw);
    o.a.TH = patch_size / 2;
    o.a.mode = bounds == 0 ? amvs::MODE_EVAL : (bounds == 1 ? amvs::MODE_CONF : amvs::MODE_EVAL + 100);
    if (e == hipSuccess) e = amvs::launch_sample_dump(n_src, o.a, ds, dv, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(sampled_out, ds, 4 * o.hw * n_src, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(valid_out, dv, o.hw, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(ds);
    if (dv) (void)hipFree(dv);
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("sample_sources: ") + hipGetErrorString(e));
    return AMVS_OK;
}

int amvs_confidence(amvs_ctx *c, int ref, const int *src_ids, int n_src, int patch_size,
                    const float *depth_in, float *conf_out)
{
    OneStep o;
    int rc = one_step_begin(c, ref, src_ids, n_src, patch_size, o);
    if (rc) return rc;
    if (!depth_in || !conf_out) return fail(c, AMVS_EINVAL, "NULL argument");
    if ((rc = upload_state(c, o.hw, depth_in, nullptr, nullptr))) return rc;
    o.a.mode = amvs::MODE_CONF;
    set_io(o.a, c, 0, false);
    HIPCHK(c, amvs::launch_step(patch_size, n_src, o.a, c->stream));
    HIPCHK(c, hipMemcpyAsync(conf_out, c->d_aux, 4 * o.hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMVS_OK;
}

int amvs_propagate_step(amvs_ctx *c, int ref, const int *src_ids, int n_src, int patch_size, float *depth,
                        float *normal, float *cost, int oy, int ox, float depth_min)
{
    OneStep o;
    int rc = one_step_begin(c, ref, src_ids, n_src, patch_size, o);
    if (rc) return rc;
    if (!depth || !normal || !cost) return fail(c, AMVS_EINVAL, "NULL argument");
    if ((rc = upload_state(c, o.hw, depth, normal, cost))) return rc;
    o.a.mode = amvs::MODE_PROP;
    set_io(o.a, c, 0);
    o.a.oy = oy; o.a.ox = ox; o.a.depth_min = depth_min;
    HIPCHK(c, amvs::launch_step(patch_size, n_src, o.a, c->stream));
    return download_state(c, o.hw, 1, depth, normal, cost);
}

int amvs_refine_step(amvs_ctx *c, int ref, const int *src_ids, int n_src, int patch_size, float *depth,
                     float *normal, float *cost, uint64_t seed, uint32_t stream_view, uint32_t draw,
                     float depth_range, float normal_range, float depth_min, float depth_max)
{
    OneStep o;
    int rc = one_step_begin(c, ref, src_ids, n_src, patch_size, o);
    if (rc) return rc;
    if (!depth || !normal || !cost) return fail(c, AMVS_EINVAL, "NULL argument");
    if ((rc = upload_state(c, o.hw, depth, normal, cost))) return rc;
    // the job's RNG stream defaults to the reference view; tests may address another stream
    HIPCHK(c, hipMemcpyAsync(&c->d_jobs[0].stream_view, &stream_view, sizeof(uint32_t),
                             hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    o.a.mode = amvs::MODE_REFINE;
    set_io(o.a, c, 0);
    o.a.seed = seed; o.a.draw = draw;
    o.a.depth_range = depth_range; o.a.normal_range = normal_range;
    o.a.depth_min = depth_min; o.a.depth_max = depth_max;
    HIPCHK(c, amvs::launch_step(patch_size, n_src, o.a, c->stream));
    return download_state(c, o.hw, 1, depth, normal, cost);
}

int amvs_init_state(amvs_ctx *c, uint64_t seed, uint32_t stream_view, float log_depth_scale,
                    float log_depth_min, float *depth, float *normal, float *cost)
{
    if (!c) return AMVS_EINVAL;
    if (!depth || !normal || !cost) return fail(c, AMVS_EINVAL, "NULL argument");
    int rc = bind_device(c);
    if (rc) return rc;
    c->pm_resumable = false;
    if ((rc = ensure_slots(c, 1))) return rc;
    if ((rc = ensure_jobs(c, 1))) return rc;
    amvs::Job j;
    std::memset(&j, 0, sizeof(j));
    j.stream_view = stream_view;
    HIPCHK(c, hipMemcpyAsync(c->d_jobs, &j, sizeof(j), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t hw = (size_t)c->H * c->W;
    HIPCHK(c, amvs::launch_init(c->d_jobs, 1, (long long)hw, seed, log_depth_scale, log_depth_min,
                                c->d_depth[0], c->d_normal[0], c->d_cost[0], c->stream));
    return download_state(c, hw, 0, depth, normal, cost);
}

int amvs_box_stats(amvs_ctx *c, int view, int patch_size, float *mean_out, float *var_out)
{
    if (!c) return AMVS_EINVAL;
    if (view < 0 || view >= c->n_views || !c->have[view] || !mean_out || !var_out)
        return fail(c, AMVS_EINVAL, "bad view / NULL output");
    int rc = bind_device(c);
    if (rc) return rc;
    if (!amvs::patch_supported(patch_size)) return fail(c, AMVS_EUNSUPPORTED, "patch_size unsupported (odd sizes from 3 to 31)");
    if ((rc = ensure_stats(c, patch_size))) return rc;
    const Stats &s = c->stats.at(patch_size);
    const size_t hw = (size_t)c->H * c->W;
    HIPCHK(c, hipMemcpyAsync(mean_out, s.mean + view * c->stride, 4 * hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(var_out, s.var + view * c->stride, 4 * hw, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AMVS_OK;
}

int amvs_fuse_filter(amvs_ctx *c, int n_maps, const void *depth, const void *conf, int maps_on_device,
                     const uint8_t *colors_bgr_host, const double K_inv[9], const double *poses,
                     float min_views, int do_filter, int64_t counts[2])
{
    if (!c) return AMVS_EINVAL;
    if (n_maps < 1 || !depth || !conf || !colors_bgr_host || !K_inv || !poses || !counts)
        return fail(c, AMVS_EINVAL, "bad argument");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W, n = hw * (size_t)n_maps;
    if (c->d_cloud_pts) (void)hipFree(c->d_cloud_pts);
    if (c->d_cloud_rgb) (void)hipFree(c->d_cloud_rgb);
    c->d_cloud_pts = nullptr; c->d_cloud_rgb = nullptr; c->cloud_n = 0;
    float *dd = nullptr, *dc = nullptr;
    unsigned char *dbgr = nullptr;
    auto cleanup = [&]() {
        if (!maps_on_device) { if (dd) (void)hipFree(dd); if (dc) (void)hipFree(dc); }
        if (dbgr) (void)hipFree(dbgr);
    };
    hipError_t e = hipMalloc(&dbgr, 3 * n);
    if (e == hipSuccess) e = hipMemcpyAsync(dbgr, colors_bgr_host, 3 * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && !maps_on_device) {
        e = hipMalloc(&dd, 4 * n);
        if (e == hipSuccess) e = hipMalloc(&dc, 4 * n);
        if (e == hipSuccess) e = hipMemcpyAsync(dd, depth, 4 * n, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dc, conf, 4 * n, hipMemcpyHostToDevice, c->stream);
    } else if (e == hipSuccess) {
        dd = (float *)depth; dc = (float *)conf;
    }
    long long cnt[2] = {0, 0};
    if (e == hipSuccess)
        e = amvs::fuse_filter(dd, dc, dbgr, n_maps, c->H, c->W, K_inv, poses, min_views, do_filter != 0,
                              &c->d_cloud_pts, &c->d_cloud_rgb, cnt, c->stream);
    (void)hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("fuse_filter: ") + hipGetErrorString(e));
    counts[0] = cnt[0]; counts[1] = cnt[1];
    c->cloud_n = cnt[1];
    return AMVS_OK;
}

int amvs_fuse_filter_views(amvs_ctx *c, int n_maps, const int *view_ids, const void *depth_dev, const void *conf_dev,
                           const double K_inv[9], const double *poses, float min_views, int do_filter,
                           int64_t counts[2])
{
    if (!c) return AMVS_EINVAL;
    if (n_maps < 1 || !view_ids || !depth_dev || !conf_dev || !K_inv || !poses || !counts)
        return fail(c, AMVS_EINVAL, "bad argument");
    for (int j = 0; j < n_maps; ++j)
        if (view_ids[j] < 0 || view_ids[j] >= c->n_views || !c->have_bgr[view_ids[j]])
            return fail(c, AMVS_EINVAL, "view " + std::to_string(view_ids[j]) + " has no resident colour image (amvs_set_view_bgr8)");
    int rc = bind_device(c);
    if (rc) return rc;
    const size_t hw = (size_t)c->H * c->W;
    if (c->d_cloud_pts) (void)hipFree(c->d_cloud_pts);
    if (c->d_cloud_rgb) (void)hipFree(c->d_cloud_rgb);
    c->d_cloud_pts = nullptr; c->d_cloud_rgb = nullptr; c->cloud_n = 0;
    // the maps' colour images in map order (device-to-device; the images of a scene are rarely in
    // that order already)
    unsigned char *dbgr = nullptr;
    hipError_t e = hipMalloc(&dbgr, 3 * hw * (size_t)n_maps);
    for (int j = 0; j < n_maps && e == hipSuccess; ++j)
        e = hipMemcpyAsync(dbgr + 3 * hw * (size_t)j, c->d_bgr + 3 * hw * (size_t)view_ids[j], 3 * hw,
                           hipMemcpyDeviceToDevice, c->stream);
    long long cnt[2] = {0, 0};
    if (e == hipSuccess)
        e = amvs::fuse_filter((const float *)depth_dev, (const float *)conf_dev, dbgr, n_maps, c->H, c->W, K_inv, poses,
                              min_views, do_filter != 0, &c->d_cloud_pts, &c->d_cloud_rgb, cnt, c->stream);
    (void)hipStreamSynchronize(c->stream);
    if (dbgr) (void)hipFree(dbgr);
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("fuse_filter_views: ") + hipGetErrorString(e));
    counts[0] = cnt[0]; counts[1] = cnt[1];
    c->cloud_n = cnt[1];
    return AMVS_OK;
}

int amvs_fetch_cloud(amvs_ctx *c, double *points, uint8_t *colors)
{
    if (!c) return AMVS_EINVAL;
    if (c->cloud_n == 0) return AMVS_OK;
    if (!points || !colors) return fail(c, AMVS_EINVAL, "NULL output");
    int rc = bind_device(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(points, c->d_cloud_pts, sizeof(double) * 3 * c->cloud_n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(colors, c->d_cloud_rgb, 3 * c->cloud_n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return checked(c, AMVS_OK);
}

// utils.save_ply (utils.py:8-37): ASCII PLY, "%.6f %.6f %.6f %d %d %d" per vertex.  Host-only:
// formats into a 1 MiB buffer instead of one Python f.write per point.
int amvs_knn_mean_distance(amvs_ctx *c, const double *points, int64_t n, int k, double *mean_out)
{
    if (!c) return AMVS_EINVAL;
    if (!points || !mean_out || n < 1) return fail(c, AMVS_EINVAL, "NULL argument / empty cloud");
    if (!amvs::knn_supported(k)) return fail(c, AMVS_EUNSUPPORTED, "k not compiled in (8, 10, 16, 20, 32)");
    if (n < k) return fail(c, AMVS_EINVAL, "fewer points than neighbours");
    if (n > (1ll << 30)) return fail(c, AMVS_EINVAL, "cloud too large (32-bit point indices)");
    int rc = bind_device(c);
    if (rc) return rc;
    HIPCHK(c, amvs::knn_mean_distance(points, (long long)n, k, mean_out, c->stream));
    return AMVS_OK;
}

// ---- native exchange: RCCL through dlopen (no link-time dependency; with a PyTorch-ROCm wheel in the
// process the SONAME librccl.so.1 resolves to the copy torch already loaded) ----
extern "C++" {
namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl q;
        // AMVS_RCCL_LIB (read once, here): the library to open instead of the default names -- a site with
        // RCCL elsewhere, and the test of the not-found path
        const char *forced = std::getenv("AMVS_RCCL_LIB");
        std::string last;
        if (forced && *forced) {
            q.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
            if (!q.lib) { const char *e = dlerror(); last = e ? e : ""; }     // (dlerror() clears itself: call it once)
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                q.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (q.lib) break;
                const char *e = dlerror();
                last = e ? e : "";
            }
        }
        if (!q.lib) { q.why = "RCCL not found (dlopen " + std::string(forced && *forced ? forced : "librccl.so.1") + "): " + last; return q; }
        q.GetUniqueId = (decltype(q.GetUniqueId))dlsym(q.lib, "ncclGetUniqueId");
        q.CommInitRank = (decltype(q.CommInitRank))dlsym(q.lib, "ncclCommInitRank");
        q.AllGather = (decltype(q.AllGather))dlsym(q.lib, "ncclAllGather");
        q.CommDestroy = (decltype(q.CommDestroy))dlsym(q.lib, "ncclCommDestroy");
        q.GetErrorString = (decltype(q.GetErrorString))dlsym(q.lib, "ncclGetErrorString");
        if (!q.GetUniqueId || !q.CommInitRank || !q.AllGather || !q.CommDestroy || !q.GetErrorString)
            q.why = "RCCL library lacks an expected symbol";
        return q;
    }();
    return r;
}

int rccl_fail(amvs_ctx *c, const char *what, ncclResult_t e)
{
    return fail(c, AMVS_EHIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error"));
}
}  // namespace
}  // extern "C++"

int amvs_comm_unique_id(uint8_t id_out[AMVS_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == AMVS_COMM_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return fail(nullptr, AMVS_EINVAL, "NULL id");
    if (!rccl().why.empty()) return fail(nullptr, AMVS_EUNSUPPORTED, rccl().why);
    ncclUniqueId id;
    const ncclResult_t e = rccl().GetUniqueId(&id);
    if (e != ncclSuccess) return rccl_fail(nullptr, "ncclGetUniqueId", e);
    std::memcpy(id_out, &id, AMVS_COMM_ID_BYTES);
    return AMVS_OK;
}

int amvs_comm_init(amvs_ctx *c, int rank, int world, const uint8_t id_in[AMVS_COMM_ID_BYTES])
{
    if (!c) return AMVS_EINVAL;
    if (!id_in || world < 1 || rank < 0 || rank >= world) return fail(c, AMVS_EINVAL, "bad rank / world / id");
    if (!rccl().why.empty()) return fail(c, AMVS_EUNSUPPORTED, rccl().why);
    int rc = bind_device(c);
    if (rc) return rc;
    if ((rc = amvs_comm_destroy(c))) return rc;
    ncclUniqueId id;
    std::memcpy(&id, id_in, AMVS_COMM_ID_BYTES);
    const ncclResult_t e = rccl().CommInitRank(&c->comm, world, id, rank);
    if (e != ncclSuccess) { c->comm = nullptr; return rccl_fail(c, "ncclCommInitRank", e); }
    c->comm_rank = rank; c->comm_world = world;
    return AMVS_OK;
}

int amvs_allgather_maps(amvs_ctx *c, const void *local_dev, void *full_dev, int64_t floats_per_rank)
{
    if (!c) return AMVS_EINVAL;
    if (!c->comm) return fail(c, AMVS_EINVAL, "no communicator (amvs_comm_init)");
    if (!local_dev || !full_dev || floats_per_rank < 1) return fail(c, AMVS_EINVAL, "bad buffers / count");
    int rc = bind_device(c);
    if (rc) return rc;
    const ncclResult_t e = rccl().AllGather(local_dev, full_dev, (size_t)floats_per_rank, ncclFloat, c->comm, c->stream);
    if (e != ncclSuccess) return rccl_fail(c, "ncclAllGather", e);
    return AMVS_OK;
}

int amvs_comm_destroy(amvs_ctx *c)
{
    if (!c) return AMVS_EINVAL;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        const ncclResult_t e = rccl().CommDestroy(c->comm);
        c->comm = nullptr; c->comm_world = 0;
        if (e != ncclSuccess) return rccl_fail(c, "ncclCommDestroy", e);
    }
    return AMVS_OK;
}

// "%.6f" of a double, the bytes printf writes (correctly rounded decimal expansion of the exact binary value,
// ties to even -- glibc), without printf for the common case: for |x| < 1e9 the scaled value x * 1e6 splits
// into an integer n (exact as a double: below 2^53) and a residual r = fma(|x|, 1e6, -n), which is exact up to
// one rounding far below the decision margin; the sixth decimal rounds up iff r > 1/2.  A residual within 1e-9
// of 1/2 (true ties exist: 0.0078125 * 1e6 = 7812.5) and everything outside the range goes through snprintf.
// (The per-point printf was 40 % of the CLI-default run's end-to-end time: 24 of 61 ms for 56 k points.)
static inline char *put_u64(char *o, uint64_t v)
{
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *o++ = tmp[--k];
    return o;
}

static inline char *put_f6(char *o, double x)
{
    const double ax = std::fabs(x);
    if (!(ax < 1e9)) return o + std::snprintf(o, 400, "%.6f", x);       // (also NaN / inf)
    uint64_t n = (uint64_t)(ax * 1e6);
    double r = std::fma(ax, 1e6, -(double)n);
    if (r < 0.0) { n -= 1; r += 1.0; }
    if (r >= 1.0) { n += 1; r -= 1.0; }
    if (std::fabs(r - 0.5) < 1e-9 || r < 0.0 || r >= 1.0) return o + std::snprintf(o, 400, "%.6f", x);
    if (r > 0.5) n += 1;
    if (std::signbit(x)) *o++ = '-';
    o = put_u64(o, n / 1000000u);
    *o++ = '.';
    uint32_t f = (uint32_t)(n % 1000000u);
    for (int i = 5; i >= 0; --i) { o[i] = (char)('0' + f % 10); f /= 10; }
    return o + 6;
}

static inline char *put_i64(char *o, long long v)
{
    if (v < 0) { *o++ = '-'; return put_u64(o, (uint64_t)(-(v + 1)) + 1u); }
    return put_u64(o, (uint64_t)v);
}

int amvs_write_ply(const char *path, const double *points, const int64_t *colors, int64_t n)
{
    if (!path || n < 0 || (n > 0 && (!points || !colors))) return fail(nullptr, AMVS_EINVAL, "bad argument");
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(nullptr, AMVS_EINVAL, std::string("cannot open ") + path);
    std::vector<char> buf(1 << 20);
    size_t used = (size_t)std::snprintf(buf.data(), buf.size(),
                                        "ply\nformat ascii 1.0\nelement vertex %lld\nproperty float x\n"
                                        "property float y\nproperty float z\nproperty uchar red\n"
                                        "property uchar green\nproperty uchar blue\nend_header\n",
                                        (long long)n);
    bool ok = true;
    for (int64_t i = 0; i < n && ok; ++i) {
        if (used + 1400 > buf.size()) {              // (a "%.6f" of the largest double is 316 characters)
            ok = std::fwrite(buf.data(), 1, used, f) == used;
            used = 0;
        }
        char *o = buf.data() + used;
        o = put_f6(o, points[3 * i]); *o++ = ' ';
        o = put_f6(o, points[3 * i + 1]); *o++ = ' ';
        o = put_f6(o, points[3 * i + 2]); *o++ = ' ';
        o = put_i64(o, (long long)colors[3 * i]); *o++ = ' ';
        o = put_i64(o, (long long)colors[3 * i + 1]); *o++ = ' ';
        o = put_i64(o, (long long)colors[3 * i + 2]); *o++ = '\n';
        used = (size_t)(o - buf.data());
    }
    if (ok && used) ok = std::fwrite(buf.data(), 1, used, f) == used;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? AMVS_OK : fail(nullptr, AMVS_EINVAL, std::string("write failed: ") + path);
}

int amvs_selftest_lean_math(amvs_ctx *c, uint64_t mismatches[2])
{
    if (!c || !mismatches) return AMVS_EINVAL;
    int rc = bind_device(c);
    if (rc) return rc;
    unsigned long long *d = nullptr;
    HIPCHK(c, hipMalloc(&d, 16));
    hipError_t e = hipMemsetAsync(d, 0, 16, c->stream);
    if (e == hipSuccess) e = amvs::launch_lean_math_check(d, c->stream);
    unsigned long long h[2] = {~0ull, ~0ull};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("selftest: ") + hipGetErrorString(e));
    mismatches[0] = h[0]; mismatches[1] = h[1];
    return AMVS_OK;
}

int amvs_rng_fill(amvs_ctx *c, uint64_t seed, uint32_t stream_view, uint32_t draw, int64_t n, float *u_out,
                  float *n_out)
{
    if (!c) return AMVS_EINVAL;
    if (n < 0 || n > (1ll << 31)) return fail(c, AMVS_EINVAL, "bad n");
    int rc = bind_device(c);
    if (rc) return rc;
    float *du = nullptr, *dn = nullptr;
    if (u_out) HIPCHK(c, hipMalloc(&du, 4 * (size_t)(n ? n : 1)));
    if (n_out && hipMalloc(&dn, 12 * (size_t)(n ? n : 1)) != hipSuccess) {
        if (du) (void)hipFree(du);
        return fail(c, AMVS_EHIP, "hipMalloc(rng) failed");
    }
    hipError_t e = amvs::launch_rng_fill(seed, stream_view, draw, n, du, dn, c->stream);
    if (e == hipSuccess && du) e = hipMemcpyAsync(u_out, du, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && dn) e = hipMemcpyAsync(n_out, dn, 12 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (du) (void)hipFree(du);
    if (dn) (void)hipFree(dn);
    if (e != hipSuccess) return fail(c, AMVS_EHIP, std::string("rng_fill: ") + hipGetErrorString(e));
    return AMVS_OK;
}

}  // extern "C"
#pragma GCC visibility pop
