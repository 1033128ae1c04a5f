// amvs_kernels.h -- host-visible launch interface of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AMVS_KMAX_SRC 6
#define AMVS_SWEEP_MAX_TH 32          // rows per strip with 16-bit running-best keys ...
#define AMVS_SWEEP_MAX_CHUNK 4096     // ... of up to 4096 planes per wave (12-bit plane index in its LDS keys)
#define AMVS_SWEEP_MAX_TH8 64         // rows per strip with 8-bit keys (SweepArgs::key8: the same 4 KB of LDS) ...
#define AMVS_SWEEP_MAX_CHUNK8 32      // ... of up to 32 planes per wave (5-bit plane index, 3-bit vote count)
#define AMVS_MAX_PATCH 31             // largest (odd) patch size: the run-time-k kernels' rings fit 64 KB of LDS up to here

namespace amvs {

// One source view of a job: everything its sampling needs in one 64-byte record (a single
// s_load_dwordx16 and one wait per source instead of three load / wait rounds).
struct alignas(64) SrcEntry {
    float R[9], t[3];
    unsigned long long pairs;   // first element of the zero-bordered packed map
    unsigned long long gray;    // float32 gray map
};

// Fast (tolerance) mode: the chain back-project -> world -> source camera -> pixel of
// mvs_patchmatch.py:341-360 precomposed per (reference, source) pair in double on the host:
//   [u z, v z, z]^T = d * (M [x, y, 1]^T) + b,  M = K R_s R_ref^T K^-1,  b = K (t_s - R_s R_ref^T t_ref)
struct alignas(64) FastSrc {
    float M[9], b[3];
    unsigned long long pairs;   // first element of the zero-bordered packed map
    unsigned long long pad_;
};

// One reference view of a batch: its pose, its source views and where its state lives.
struct alignas(64) Job {
    float K[9], Kinv[9];    // shared intrinsics, copied per job so that the kernels fetch them with
                            // the same in-loop scalar loads as the poses
    float Rref[9], tref[3];
    int ref_img;
    uint32_t stream_view;   // RNG stream id (the reference view's index)
    int slot;               // state / output slot inside the batch buffers
    int n_src;
    unsigned long long ref_pairs;   // image pixel (0,0) of the reference view's packed map
    unsigned long long ref_stats;   // fast mode: float2 (mean1, var1) map of the reference view
    SrcEntry src[AMVS_KMAX_SRC];
    FastSrc fsrc[AMVS_KMAX_SRC];    // fast mode
};

enum Mode { MODE_EVAL = 0, MODE_PROP = 1, MODE_REFINE = 2, MODE_CONF = 3 };

// where the images live (common head of StepArgs and SweepArgs)
struct StepArgsBase {
    long long img_stride;                    // floats between consecutive images
    const float *images;                     // float32 gray maps [n_views][img_stride]
    const uint16_t *pairs;                   // packed 8-bit row-pair maps [n_views][pair_stride], or NULL
    long long pair_stride;
    int fast;                                // 1: fast (tolerance) arithmetic; needs `pairs`
};

struct StepArgs : StepArgsBase {
    int H, W, TH, tiles_x, tiles_y, n_jobs;
    // Strip order.  0 (view-major): strips run view by view, row by row -- an XCD's resident
    // waves cover ~19 consecutive strip rows of one view.  1 (band-major): strips run band by band
    // (a band = one strip row of EVERY view of the launch), so an XCD's resident waves walk the same
    // rows of all views together; the views' source images are each other, so the XCD's L2 holds
    // one thin row front of every image instead of a tall band of a few.
    int band_major;
    // State [slot][H*W] (normals [slot][H*W*3]).  Depth is ping-ponged on every step (halo pixels
    // of other strips read the pre-step map).  Cost is only ever touched at a lane's own pixel and
    // updated in place.  Normals never enter the cost (mvs_patchmatch.py:323-390 ignores them), so
    // they are moved only where a candidate WINS: they live in two buffers nbuf[0], nbuf[1], and the
    // SIGN BIT of a pixel's depth in the state maps names the buffer that holds its current normal
    // (depths are >= depth_min > 0, so the bit is free; it travels with the depth ping-pong at no
    // cost).  A propagation winner p (candidate = the NEIGHBOUR q's pre-step normal, :452-455) reads
    // nbuf[sign d_in[q]][q], writes the buffer p does not currently use and stores its new depth with
    // the flipped sign; a refinement winner updates its current buffer in place; losers touch no
    // normal.  No pixel reads a location another pixel writes in the same launch.  `depth_mask` strips
    // the tag from every depth read: 0x7FFFFFFF for state maps, 0xFFFFFFFF for caller-supplied maps
    // (amvs_eval_cost / amvs_confidence).  launch_resolve_state turns tagged state into plain maps.
    const float *d_in;
    float *d_out;
    float *cost;
    float *nbuf[2];
    unsigned depth_mask;
    float *aux;                              // MODE_EVAL: cost map, MODE_CONF: confidence
    const Job *jobs;
    int mode, oy, ox;
    float depth_min, depth_max, depth_range, normal_range;
    unsigned long long seed;
    unsigned draw;
    // Split schedule (fast mode): sample maps [slot][n_src][H*W] written by launch_sample_fast and
    // read by launch_step when `presampled`; the sampling kernel's own strips (64 columns, no halo).
    float *samples;
    int presampled;
    int half;                                // patch_size / 2
    int s_TH, s_tiles_x, s_tiles_y;
    int s_lds;                               // unused dynamic LDS per sampling workgroup (occupancy cap); 0 = default
    // Resident workgroups per CU of this sweep launch (enforced through unused dynamic LDS: 160 KiB /
    // wg_cap per workgroup); 0 = the default of AMVS_DEFAULT_WGS_PER_CU.  Performance only.
    int wg_cap;
    // Paired-band schedule of the fast step (pm_step_fast_kernel<..., PAIR = true>): 2 x 2 strips per
    // workgroup, vertically adjacent bands exchange their boundary samples through LDS
    int paired;
};

#define AMVS_DEFAULT_WGS_PER_CU 4

struct SweepArgs : StepArgsBase {
    int H, W, TH, tiles_x, tiles_y, n_jobs, D;
    int n_chunks, chunk;                     // planes are split into n_chunks groups of `chunk`
    int key8;                                // 1: 8-bit running-best keys (chunk <= 32): strips of up to 64 rows in the same LDS
    unsigned *keys;                          // [slot][H*W] running best, merged with atomicMax
    const float *depths;                     // [D] device
    float thresh;
    float *depth_out, *conf_out;             // [slot][H*W]
    const Job *jobs;
    // exact arithmetic, compiled patch sizes: mean1 / var1 maps of every view for this patch size
    // ([n_views][img_stride], launch_box_stats) -- plane-invariant, so the sweep loads them instead of
    // re-forming the reference's window sums for every plane
    const float *ref_mean, *ref_var;
};

bool patch_supported(int K);          // any odd patch size in 3 .. AMVS_MAX_PATCH
bool patch_compiled(int K);           // ... with kernels specialised at compile time (3, 5, ..., 29); the others run
                                      // the run-time-k kernels of amvs_generic.hip
// amvs_generic.hip: sweep step / plane sweep / statistics with the patch size as a launch argument (both
// arithmetic modes, classic schedule); launch_step / launch_sweep / launch_box_stats / launch_fast_stats forward
// to these for patch sizes that are not compiled in
hipError_t launch_step_generic(int K, int S, const StepArgs &a, hipStream_t st);
hipError_t launch_sweep_generic(int K, int S, const SweepArgs &a, hipStream_t st);
hipError_t launch_box_stats_generic(int K, const float *images, long long img_stride, int H, int W, int first_img, int n_img,
                                    float *mean_out, float *var_out, hipStream_t st);
hipError_t launch_fast_stats_generic(int K, const uint16_t *pairs_view, int H, int W, float2 *out, hipStream_t st);
int step_generic_waves_per_cu(int K, int S);
bool knn_supported(int k);
hipError_t knn_mean_distance(const double *points, long long n, int k, double *mean_out, hipStream_t st,
                             bool points_on_device = false);
int strip_out_width(int K);
int step_waves_per_cu(int K, int S, bool u8, int wg_cap = 0);      // resident waves per CU under the cap (0 = default)
hipError_t launch_step(int K, int S, const StepArgs &a, hipStream_t st);
hipError_t launch_sweep(int K, int S, const SweepArgs &a, hipStream_t st);
// amvs_kernels_fast.hip: the same steps in the fast arithmetic (a.fast != 0; launch_step /
// launch_sweep forward to these)
hipError_t launch_step_fast(int K, int S, const StepArgs &a, hipStream_t st);
hipError_t launch_sample_fast(int S, const StepArgs &a, hipStream_t st);      // split schedule, first half
hipError_t launch_sweep_fast(int K, int S, const SweepArgs &a, hipStream_t st);
int step_fast_waves_per_cu(int K, int S, int wg_cap = 0);
bool step_fast_pair_supported(int K, int S);
bool step_pair_supported(int K, int S);            // ... of the exact arithmetic (packed 8-bit maps)       // the paired-band schedule is compiled for this patch / source count
// test hook: per-source samples [S][H*W] and validity bits [H*W] of job 0 at the depth map a.d_in;
// a.TH carries k/2, a.mode selects the bounds (MODE_EVAL patch bounds, MODE_CONF image bounds,
// MODE_EVAL + 100 depth test only = plane sweep)
hipError_t launch_sample_dump(int S, const StepArgs &a, float *out, unsigned char *valid_out, hipStream_t st);
hipError_t launch_sample_dump_fast(int S, const StepArgs &a, float *out, unsigned char *valid_out, hipStream_t st);
// (mean1, var1) of one view under the k x k zero-padded box filter from its packed 8-bit map:
// exact integer window sums, statistics in gray units (float2 per pixel)
hipError_t launch_fast_stats(int K, const uint16_t *pairs_view, int H, int W, float2 *out, hipStream_t st);
// the double-precision composition of FastSrc::M / b
void fast_compose(const float K[9], const float Rr[9], const float tr[3], const float Rs[9], const float ts[3],
                  float M[9], float b[3]);
hipError_t launch_sweep_finish(const SweepArgs &a, hipStream_t st);
hipError_t launch_box_stats(int K, const float *images, long long img_stride, int H, int W,
                            int first_img, int n_img, float *mean_out, float *var_out,
                            hipStream_t st);
long long pair_map_elems(int H, int W);      // ushorts of one zero-bordered packed map
long long pair_map_origin(int W);            // texel offset of image pixel (0,0) inside it
int pair_map_texel_bytes();                  // 2 (row-pair layout) or 1 (byte layout)
hipError_t launch_pack_pairs(const float *img, int H, int W, uint16_t *pairs, int *inexact,
                             hipStream_t st);
hipError_t launch_init(const Job *jobs, int n_jobs, long long HW, unsigned long long seed,
                       float log_scale, float log_min, float *depth, float *normal, float *cost,
                       hipStream_t st);
// Tagged state (StepArgs::nbuf) -> plain maps for the slots of `jobs`: depth_out[p] = |depth[p]|,
// normal_out[p] = nbuf[sign depth[p]][p].  Outputs are indexed by (slot - out_slot0); with depth_out ==
// NULL the state is resolved in place (depth untagged, current normals gathered into nbuf0).
hipError_t launch_resolve_state(const Job *jobs, int n_jobs, long long HW, float *depth, float *nbuf0, const float *nbuf1,
                                float *depth_out, float *normal_out, long long out_slot0, hipStream_t st);
hipError_t launch_lean_math_check(unsigned long long *mismatch, hipStream_t st);
hipError_t launch_rng_fill(unsigned long long seed, unsigned view, unsigned draw, long long n,
                           float *u_out, float *n_out, hipStream_t st);

// amvs_fusion.hip: fusion (+ filter) of per-view maps into a cloud; results are hipMalloc'ed
hipError_t fuse_filter(const float *depth, const float *conf, const unsigned char *bgr, int n_maps, int H, int W,
                       const double *Kinv_h, const double *poses_h, float min_views, bool do_filter,
                       double **pts_out, unsigned char **rgb_out, long long counts[2], hipStream_t st);

// amvs_extended.hip: the extended PatchMatch mode (slanted-plane cost, red-black schedule, view
// propagation, geometric consistency); no reference counterpart
struct XArgs {
    int H, W, n_jobs, n_src;
    const Job *jobs;
    const float *images;          // float32 gray maps [n_views][img_stride]
    long long img_stride;
    const uint16_t *pairs;        // packed 8-bit row-pair maps [n_views][pair_stride] (NULL: sample `images`)
    long long pair_stride;
    float *depth, *normal, *cost; // state of ALL views: [n_views][H*W] (normal x3), indexed by Job::ref_img
    const float *snap_depth, *snap_normal;   // what the view candidates read: a snapshot of depth / normal taken
                                             // before any call of the iteration wrote a map (or the live maps)
    float *cand_d, *cand_n;       // view-propagation candidates [n_jobs][H*W] (x3), indexed by Job::slot
    const int *src_view;          // [n_jobs][n_src] view ids of the sources
    int patch, stride;            // window side, sample stride inside it
    float depth_min, depth_max;
    float rel_range, nrm_range;   // refinement ranges of this iteration
    int n_refine, with_random, colour, with_view_cand;
    unsigned long long seed;
    unsigned draw;
};

hipError_t launch_xpm_init(const XArgs &a, float log_scale, float log_min, hipStream_t st);
hipError_t launch_xpm_view_candidates(const XArgs &a, hipStream_t st);
hipError_t launch_xpm_sweep(const XArgs &a, hipStream_t st);
hipError_t launch_xpm_eval(const XArgs &a, float *cost_out, hipStream_t st);   // test hook: cost of every pixel's current plane
hipError_t launch_xpm_consistency(const XArgs &a, float *conf_out, float max_px, float max_rel, hipStream_t st);

// amvs_prep.hip: cv.resize (INTER_LINEAR, 8-bit BGR) + cvtColor(BGR2GRAY) / 255 of one view; the
// weight tables are built on the host (amvs_capi.hip, OpenCV's float32 arithmetic)
hipError_t launch_prep_bgr8(const unsigned char *src, int sh, int sw, int dh, int dw, const int *xofs,
                            const short *ialpha, const int *yofs, const short *ibeta, unsigned char *scaled,
                            float *gray, hipStream_t st);

// amvs_fusion.hip: the stereo path's post-steps (dense_stereo.py:407-437, 475-492)
hipError_t stereo_backproject(const float *depth, const float *conf, const unsigned char *bgr, int n_maps, int H, int W,
                              const double *Kinv_h, const double *poses_h, float min_confidence, double **pts_out,
                              unsigned char **rgb_out, long long *total, long long *per_map_h, hipStream_t st);
hipError_t cloud_take(const double *pts, const unsigned char *rgb, long long n, const long long *idx_h, long long m,
                      double **pts_out, unsigned char **rgb_out, hipStream_t st);
hipError_t voxel_downsample(const double *pts, const unsigned char *rgb, long long m, const unsigned char *keep_h,
                            double voxel, double **pts_out, unsigned char **rgb_out, long long *m_out, hipStream_t st);

}  // namespace amvs
