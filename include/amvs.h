/*
 * amvs.h -- C ABI of the MI355X-native dense-reconstruction backend.
 *
 * The reference (dackey-wav/3d-reconstruction-tool) has no FFI: its dense path is a
 * Python class surface that runs stock torch ops.  These entry points are what a
 * ctypes binding of that surface needs; each cites the reference interface it
 * replaces (file:line under the reference's src/core/).  Plain pointers and
 * sizes only, no C++ or torch types.  Every call returns 0 on success or a
 * negative AMVS_E* code; amvs_last_error() returns a description.
 *
 * Conventions:
 *   - images are float32 gray maps in [0,1], row-major (H,W), one size per context
 *     (mvs_patchmatch.py:183-189 'gray'); all views share one K (camera.py:111-139)
 *   - poses are world->camera, X_c = R X_w + t, R row-major (camera.py:78-103)
 *   - depth/cost/confidence maps are float32 (H,W); normal maps float32 (H,W,3)
 *     interleaved xyz (mvs_patchmatch.py:30-35 DepthNormalMap)
 *   - host buffers are caller-owned and C-contiguous; *_device variants take
 *     device pointers (e.g. torch tensor .data_ptr()) and enqueue on the
 *     context's stream without synchronising.
 */
#ifndef AMVS_H
#define AMVS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMVS_MAX_SRC 6      /* PatchMatch uses 4 (mvs_patchmatch.py:108), plane sweep 6 (dense_stereo.py:109) */

#define AMVS_OK            0
#define AMVS_EINVAL       -1   /* bad argument                                  */
#define AMVS_EHIP         -2   /* HIP runtime error (no device, OOM, launch)    */
#define AMVS_EINDEX       -4   /* index-checked build only: a kernel formed an out-of-range index (amvs_index_check) */
#define AMVS_EUNSUPPORTED -3   /* patch size (even, or above 31) / source count outside [2, 6] / k of the kNN */

typedef struct amvs_ctx amvs_ctx;

/* Arithmetic of the sweep kernels (SURVEY.md section 8b proposed a `mode` parameter).
 *   AMVS_MODE_EXACT  every float32 operation of the reference's torch chain reproduced in order
 *                    (mvs_patchmatch.py:341-411): bit-identical to the exact mode of the tests' CPU checker,
 *                    which matches torch-CPU bit for bit up to the box filter's summation order.
 *   AMVS_MODE_FAST   the same algorithm with the projection precomposed per source, one
 *                    reciprocal per projection and 8-bit code arithmetic:
 *                    bit-identical to the fast mode of the tests' CPU checker, which is pinned against the
 *                    reference's golden vectors within the tolerances DESIGN.md states (cost
 *                    mean 2e-6; >= 98 % of depths within 1e-3 relative end to end).  Needs 8-bit
 *                    images (every view exactly code/255); otherwise the call fails with
 *                    AMVS_EUNSUPPORTED.
 * AMVS_MODE_DEFAULT in amvs_pm_params.mode means "the context's mode" (amvs_set_mode).          */
#define AMVS_MODE_DEFAULT 0
#define AMVS_MODE_EXACT   1
#define AMVS_MODE_FAST    2

/* PatchMatchMVS constructor parameters that reach the device path
 * (mvs_patchmatch.py:43-50) plus the depth range of _estimate_depth_range
 * (:141-165).  log_depth_scale / log_depth_min are (float)(ln dmax - ln dmin)
 * and (float)ln dmin formed in double by the host, as :268-271 does.           */
typedef struct {
    int32_t patch_size;       /* any odd size in 3..31 (mvs_patchmatch.py:45 takes any); 3, 5, ..., 29 run
                                 kernels specialised at compile time, 31 the run-time-k kernels
                                 (csrc/amvs_generic.hip: same results contract, classic schedule, slower)  */
    int32_t num_iterations;
    int32_t num_samples;
    int32_t tile_rows;        /* rows per wave strip; 0 = choose automatically    */
    int32_t views_per_launch; /* views swept together (cache residency); 0 = auto  */
    float   depth_min, depth_max;
    float   log_depth_scale, log_depth_min;
    int32_t mode;             /* AMVS_MODE_DEFAULT / _EXACT / _FAST                    */
    int32_t schedule;         /* AMVS_SCHEDULE_*: 0 = automatic, 1 = view-major strips, 2 = band-major
                                 strips, 3 = split (fast mode only: sampling kernel + window kernel,
                                 pipelined over view groups).  Performance only; results do not
                                 depend on it.                                          */
    int32_t first_iteration;  /* 0: a new sweep (random initialisation, mvs_patchmatch.py:268-284).
                                 k > 0: CONTINUE the sweep of the previous PatchMatch call of this context
                                 -- same batch, sources, patch, samples, seed, depth range -- with
                                 iterations k .. k + num_iterations - 1 of the schedule (:287-308); k must
                                 equal the iterations already run.  A sweep run one iteration per call
                                 returns the maps of one call bit for bit; the outputs of every call are
                                 the maps as they stand (what a per-iteration exchange gathers).   */
    int32_t flags;            /* AMVS_PM_NO_CONFIDENCE: skip the confidence pass (conf output untouched) */
} amvs_pm_params;

#define AMVS_PM_NO_CONFIDENCE 1

#define AMVS_SCHEDULE_AUTO        0
#define AMVS_SCHEDULE_VIEW_MAJOR  1
#define AMVS_SCHEDULE_BAND_MAJOR  2
#define AMVS_SCHEDULE_SPLIT       3
#define AMVS_SCHEDULE_PAIRED      4   /* specialised patch sizes 3 .. 11, up to 4 sources, 8-bit images (the fast mode, and the
                                         exact mode on its packed maps): 2 x 2 strips per workgroup, vertically adjacent bands
                                         walk towards each other and exchange their boundary samples through LDS (K/2 halo
                                         rows per strip instead of K - 1); elsewhere it runs as view-major.
                                         AMVS_SCHEDULE_AUTO chooses it for patches of 5x5 and 7x7 (measured faster there;
                                         9x9 / 11x11: measured equal or slower).  The maps do not depend on the schedule.   */

/* Per-call device timing of the sweep kernels (HIP events on the context
 * stream; used by bench.py for the roofline figure).                            */
typedef struct {
    double  init_ms;          /* initialisation launch                             */
    double  sweep_ms;         /* propagation + refinement launches (pm_step)       */
    double  confidence_ms;    /* confidence launch                                 */
    int64_t sweep_launches;   /* number of pm_step launches inside sweep_ms        */
    int64_t pixel_hypotheses; /* n_ref * H * W * iters * (2 + samples)            */
} amvs_timing;

const char *amvs_version(void);
const char *amvs_last_error(const amvs_ctx *ctx);   /* ctx may be NULL: error of a failed amvs_create */

/* A context owns all device memory for one scene: n_views gray images of H x W
 * with shared intrinsics K (row-major 3x3) and K_inv (the reference forms it with
 * torch.inverse in float32, mvs_patchmatch.py:237-238; the caller passes it).   */
int amvs_create(int device_id, int H, int W, int n_views,
                const float K[9], const float K_inv[9], amvs_ctx **out);
int amvs_destroy(amvs_ctx *ctx);
int amvs_set_stream(amvs_ctx *ctx, void *hip_stream);     /* NULL = context's own stream */
int amvs_sync(amvs_ctx *ctx);

/* Upload one view once per scene (the reference re-uploads every view for every
 * reference view, mvs_patchmatch.py:235-257).  amvs_set_view returns when the host buffer
 * has been consumed.  amvs_set_view_device only ORDERS the copy on the context's stream: the
 * device buffer must be complete before the call (as seen from that stream) and stay unchanged
 * until the stream has passed it (amvs_sync, or any synchronising call).                        */
int amvs_set_view(amvs_ctx *ctx, int view, const float *gray_host,
                  const float R[9], const float t[3]);
int amvs_set_view_device(amvs_ctx *ctx, int view, const void *gray_device,
                         const float R[9], const float t[3]);

/* _prepare_images (mvs_patchmatch.py:167-191, dense_stereo.py:156-176) of one view on the device: the
 * undistorted 8-bit BGR image (src_h x src_w x 3, host; sfm_pipeline.py:114-120) is uploaded as it is
 * (3 B/pixel), resized to the context's H x W with cv.resize's INTER_LINEAR fixed-point arithmetic,
 * converted with cvtColor(BGR2GRAY)'s and divided by 255; the packed 8-bit map the sweeps sample is
 * built from it directly.  scaled_bgr_out (optional, H x W x 3) receives the resized colour image the
 * fusion takes its colours from.  The context must have been created with H = int(src_h * scale),
 * W = int(src_w * scale).  OpenCV is absent from the build container: the arithmetic restates its
 * published source and equals core/imageprep.py bit for bit; parity with cv2 itself is unpinned.
 * STATED TOLERANCE against cv2 (a5): bit equality of the resized colour image and of the gray codes is
 * expected for OpenCV 4.x's generic code path; an OpenCV build that dispatches the 8-bit resize to IPP
 * or another HAL, or OpenCV 3.x's 14-bit gray coefficients, may differ by at most +-1 gray code (1/255)
 * on isolated pixels.  tests/test_host_logic.py::test_image_preparation_equals_cv2_when_present checks
 * the bit equality wherever cv2 is importable; the Python classes use cv2 itself there by default.   */
int amvs_set_view_bgr8(amvs_ctx *ctx, int view, const uint8_t *bgr_host, int src_h, int src_w,
                       const float R[9], const float t[3], uint8_t *scaled_bgr_out);

/* PatchMatchMVS._patchmatch_cuda (mvs_patchmatch.py:225-321) for n_ref reference
 * views in one batch.  src_ids is [n_ref][n_src].  Outputs are [n_ref][H][W]
 * (depth, confidence) and [n_ref][H][W][3] (normal).  The RNG stream of a view is
 * (seed, ref_ids[i]); see amvs_rng_fill.                                          */
int amvs_patchmatch(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *src_ids,
                    int n_src, const amvs_pm_params *p, uint64_t seed,
                    float *depth_out, float *normal_out, float *conf_out);
int amvs_patchmatch_device(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *src_ids,
                           int n_src, const amvs_pm_params *p, uint64_t seed,
                           void *depth_dev, void *normal_dev, void *conf_dev);
int amvs_get_timing(const amvs_ctx *ctx, amvs_timing *out);
/* 1 if the sweeps sample the packed 8-bit row-pair maps (every uploaded view is exactly
 * code/255, as cvtColor(...).astype(float32)/255 yields, mvs_patchmatch.py:177), 0 if they
 * sample the float32 maps.  In exact mode both give bit-identical results;
 * amvs_set_sampling(ctx, 1) selects the float32 path unconditionally (A/B tests).           */
int amvs_sampling_mode(const amvs_ctx *ctx);
int amvs_set_sampling(amvs_ctx *ctx, int force_f32);
/* Arithmetic mode of every later sweep call of this context (PatchMatch with
 * amvs_pm_params.mode == AMVS_MODE_DEFAULT, plane sweep, the single-step entry points).
 * A new context is in AMVS_MODE_EXACT.                                                       */
int amvs_set_mode(amvs_ctx *ctx, int mode);
int amvs_get_mode(const amvs_ctx *ctx);
/* Plane-sweep launch shape: rows per wave strip (1..64; more than 32 only for the compiled patch sizes and at most
 * 32 planes per wave -- the strip's running best then uses 8-bit keys) and planes per wave; 0 = automatic.   */
int amvs_set_sweep_tuning(amvs_ctx *ctx, int tile_rows, int planes_per_wave);
/* Launch shape of the PatchMatch sweep steps by iteration (performance only; the maps do not depend
 * on it -- tests/test_hip_fullsize_parity.py): tile_rows / wgs_per_cu are [n_iterations][2] tables,
 * column 0 = the two propagation launches of an iteration (mvs_patchmatch.py:415-457), column 1 = its
 * refinement launches (:459-491); an entry 0 = automatic, iterations beyond the table use its last
 * row, n_iterations = 0 clears it.  wgs_per_cu caps the resident workgroups (4 waves each) per CU.
 * amvs_pm_params.tile_rows > 0 overrides the row entries.                                      */
int amvs_set_step_tuning(amvs_ctx *ctx, int n_iterations, const int32_t *tile_rows, const int32_t *wgs_per_cu);
/* Per-launch device times of the sweep steps of the LAST PatchMatch call (HIP events on the context
 * stream, recorded when enabled; view groups concatenated, schedule order within a group).      */
int amvs_set_step_timing(amvs_ctx *ctx, int enable);
int amvs_get_step_times(amvs_ctx *ctx, float *ms_out, int capacity, int *n_out);
/* Split schedule (AMVS_SCHEDULE_SPLIT): number of view groups pipelined against each other (1..8),
 * rows per strip of the sampling kernel, and bytes of unused LDS per sampling workgroup (caps how
 * many of them a CU holds, which leaves room for the window kernel); 0 = automatic.              */
int amvs_set_split_tuning(amvs_ctx *ctx, int groups, int sample_rows, int sample_lds_bytes);
/* Rows per wave strip the last sweep used (amvs_pm_params.tile_rows, or the automatic choice). */
int amvs_last_tile_rows(const amvs_ctx *ctx);
/* Views per launch group the last PatchMatch call used (amvs_pm_params.views_per_launch or auto). */
int amvs_last_views_per_launch(const amvs_ctx *ctx);

/* DenseStereoReconstructor._plane_sweep_torch (dense_stereo.py:222-316) for one
 * reference view: D depth planes, votes (ncc > thresh) & (z > 0.1) over n_nbr
 * neighbours, first maximal plane wins.                                          */
int amvs_plane_sweep(amvs_ctx *ctx, int ref, const int *nbr_ids, int n_nbr,
                     const float *depths, int D, int patch_size, float thresh,
                     float *depth_out, float *conf_out);
int amvs_plane_sweep_device(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *nbr_ids,
                            int n_nbr, const float *depths, int D, int patch_size,
                            float thresh, void *depth_dev, void *conf_dev);

/* The reconstruct loop of DenseStereoReconstructor (dense_stereo.py:105-130) as ONE batched sweep
 * whose maps stay in the context: [n_ref][H][W] depth and vote count.  amvs_fetch_sweep_maps copies
 * maps first .. first+count-1 to the host (tests; the multi-rank gather).                        */
int amvs_plane_sweep_batch(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *nbr_ids, int n_nbr,
                           const float *depths, int D, int patch_size, float thresh);
int amvs_fetch_sweep_maps(amvs_ctx *ctx, int first, int count, float *depth_out, float *conf_out);

/* DenseStereoReconstructor._backproject (dense_stereo.py:407-437) for n_maps reference views at once,
 * in float64 and in the reference's order (view by view, row-major): pixels with confidence >=
 * min_confidence and depth > 0 -> world points + RGB colours, kept on the device (amvs_fetch_cloud).
 * maps_where: 0 = depth / conf are host arrays, 1 = device pointers, 2 = the resident maps of the last
 * amvs_plane_sweep_batch (depth / conf ignored).  per_map_counts (optional, n_maps) receives the points
 * of every view -- the numbers of the reference's progress lines (:129).                           */
int amvs_stereo_backproject(amvs_ctx *ctx, int n_maps, const void *depth, const void *conf, int maps_where,
                            const uint8_t *colors_bgr_host, const double K_inv[9], const double *poses,
                            float min_confidence, int64_t *per_map_counts, int64_t *total);
/* The same for the resident plane-sweep batch and resident colour images (amvs_set_view_bgr8): map j
 * belongs to view view_ids[j]; nothing is uploaded.                                              */
int amvs_stereo_backproject_views(amvs_ctx *ctx, int n_maps, const int *view_ids, const double K_inv[9],
                                  const double *poses, float min_confidence, int64_t *per_map_counts,
                                  int64_t *total);
/* amvs_knn_mean_distance on the context's resident cloud (the result of amvs_stereo_backproject /
 * amvs_fuse_filter): the statistic of _filter_outliers without a host round trip of the points.     */
int amvs_cloud_knn_mean_distance(amvs_ctx *ctx, int k, double *mean_out);
/* DenseStereoReconstructor._voxel_down_sample (dense_stereo.py:475-492) of the resident cloud, after an
 * optional keep mask (one byte per point: the outlier filter's selection, computed by the caller as
 * the reference does with numpy): first point of every voxel in key order.  The cloud is replaced.  */
int amvs_cloud_voxel_downsample(amvs_ctx *ctx, const uint8_t *keep_mask, double voxel_size, int64_t *count);
/* The reference's random sub-sample of clouds above 500 000 points, points[chosen] / colors[chosen]
 * (dense_stereo.py:449-455), on the resident cloud: it is replaced by its rows indices[0 .. m) in that order.  The
 * caller draws the indices (the reference uses the unseeded np.random.choice; the Python class does the same), so
 * the cloud need not travel to the host for the outlier statistic and the voxel grid that follow.                */
int amvs_cloud_take(amvs_ctx *ctx, const int64_t *indices_host, int64_t m);
/* 1 if amvs_knn_mean_distance is compiled for this neighbour count.                                */
int amvs_knn_supported(int k);

/* The k-nearest-neighbour statistic of DenseStereoReconstructor._filter_outliers
 * (dense_stereo.py:456-460: NearestNeighbors(n_neighbors=k).fit(p).kneighbors(p), then
 * np.mean(distances[:, 1:], axis=1)) for n points (host, n x 3 float64) on the device: mean_out[i]
 * = mean distance from point i to its k-1 nearest other points, bit-identical to scikit-learn +
 * numpy (same float64 distance expression, same summation order) wherever scikit-learn uses its
 * KD-tree (k < n / 2; for smaller clouds it switches to a brute-force kernel whose rounding differs
 * by a few ulps).  k in {8, 10, 16, 20, 32}, n >= k.  The threshold mean + 2 sigma and the selection stay on the host (numpy), as there.    */
int amvs_knn_mean_distance(amvs_ctx *ctx, const double *points, int64_t n, int k, double *mean_out);

/* PatchMatchMVS._fuse_depth_maps + _filter_points (mvs_patchmatch.py:536-588) on the device, in
 * float64 and in the reference's order: pixels with confidence >= min_views of n_maps maps
 * ([n_maps][H][W] float32, host or device memory) are back-projected with the float64 K_inv and
 * poses (n_maps x 12 doubles: R row-major then t); colours come from BGR uint8 images
 * ([n_maps][H][W][3], host) and leave as RGB.  With do_filter: 95th-percentile radius cut around
 * the per-axis median, then 1 cm voxel de-duplication keeping the first point of each voxel in
 * key order.  counts[0] = fused points, counts[1] = points kept; amvs_fetch_cloud copies the
 * counts[1] x 3 points (float64) and colours (uint8) to the host.                              */
int amvs_fuse_filter(amvs_ctx *ctx, int n_maps, const void *depth, const void *conf, int maps_on_device,
                     const uint8_t *colors_bgr_host, const double K_inv[9], const double *poses,
                     float min_views, int do_filter, int64_t counts[2]);
/* Colour image (H x W x 3 uint8 BGR, at the context's size, host) of a view whose gray map was
 * uploaded with amvs_set_view[_device]: kept on the device for the *_views entry points below.
 * (amvs_set_view_bgr8 leaves the image it prepared there by itself; a later amvs_set_view[_device] of
 * the same view drops it.)                                                                        */
int amvs_set_view_colors(amvs_ctx *ctx, int view, const uint8_t *bgr_host);
/* The same fusion + filter for maps resident on the device whose colour images are resident too:
 * map j belongs to view view_ids[j], whose prepared BGR image amvs_set_view_bgr8 left on the device
 * (no host colour array: 3 B/pixel of upload per map saved).  Same results as amvs_fuse_filter.   */
int amvs_fuse_filter_views(amvs_ctx *ctx, int n_maps, const int *view_ids, const void *depth_dev, const void *conf_dev,
                           const double K_inv[9], const double *poses, float min_views, int do_filter,
                           int64_t counts[2]);
int amvs_fetch_cloud(amvs_ctx *ctx, double *points_out, uint8_t *colors_out);

/* ---- extended mode: what the reference's docstring names but does not implement ----------------
 * (mvs_patchmatch.py:1-13 lists plane hypotheses with normals and VIEW propagation; its code ignores
 * the normal in the cost, :323-390, and has no view propagation.)  Slanted-plane homography cost,
 * red-black in-place propagation, view propagation from a snapshot of the other views' maps, random
 * refinement, geometric-consistency confidence.  No reference counterpart, hence no parity: judged
 * against synthetic ground truth (tests/test_extended_mode.py); off unless a caller asks for it.
 * State arrays are caller-owned DEVICE memory covering ALL views of the context: depth / cost
 * [n_views][H][W], normal [n_views][H][W][3]; a call updates the rows of ref_ids and reads the others
 * (view propagation, consistency) -- on several GPUs the caller all-gathers the rows between
 * iterations.  conf_out of amvs_xpm_consistency is [n_ref][H][W] in ref_ids order.                */
typedef struct {
    int32_t patch_size;       /* odd window side, 3..31                                  */
    int32_t window_stride;    /* sample every window_stride-th pixel of the window       */
    int32_t num_refine;       /* perturbed hypotheses per pixel and half sweep (0..6)    */
    int32_t view_propagation; /* 0 / 1                                                   */
    float   depth_min, depth_max;
    float   log_depth_scale, log_depth_min;     /* as amvs_pm_params                     */
    float   consistency_px, consistency_rel;    /* forward-backward thresholds           */
} amvs_xpm_params;
int amvs_xpm_init(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                  const amvs_xpm_params *p, uint64_t seed, void *depth_all, void *normal_all, void *cost_all);
/* One iteration = view candidates, red half sweep, black half sweep.  snapshot_depth / snapshot_normal
 * (device, same layout as depth_all / normal_all; NULL = the live maps): what the view candidates
 * read.  A caller that splits the views of one iteration over several calls (jobs with different
 * source counts) passes a copy taken before the first call, so that no call sees maps another call
 * of the same iteration already wrote -- the result then does not depend on the grouping.          */
int amvs_xpm_iterate(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                     const amvs_xpm_params *p, int iteration, uint64_t seed,
                     void *depth_all, void *normal_all, void *cost_all,
                     const void *snapshot_depth, const void *snapshot_normal);
/* The phases of an iteration one at a time (tests/test_extended_oracle.py compares each with the CPU
 * restatement oracle/xpm_oracle.py), plus a test hook: AMVS_XPM_PHASE_EVAL writes the cost of every
 * pixel's CURRENT plane to cost_out ([n_ref][H][W], device) and changes nothing.                     */
#define AMVS_XPM_PHASE_CANDIDATES 0
#define AMVS_XPM_PHASE_RED        1
#define AMVS_XPM_PHASE_BLACK      2
#define AMVS_XPM_PHASE_EVAL       3
int amvs_xpm_step(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                  const amvs_xpm_params *p, int iteration, uint64_t seed, int phase,
                  void *depth_all, void *normal_all, void *cost_all,
                  const void *snapshot_depth, const void *snapshot_normal, void *cost_out);
/* The view-propagation candidates of the last AMVS_XPM_PHASE_CANDIDATES call: depth [n_ref][H][W]
 * (0 = none), normal [n_ref][H][W][3], host arrays (test hook).                                       */
int amvs_xpm_fetch_candidates(amvs_ctx *ctx, int n_ref, float *cand_depth_out, float *cand_normal_out);
int amvs_xpm_consistency(amvs_ctx *ctx, int n_ref, const int *ref_ids, const int *src_ids, int n_src,
                         const amvs_xpm_params *p, void *depth_all, void *normal_all, void *cost_all,
                         void *conf_out);

/* ---- single-step entry points (parity tests drive these one reference call at a time) ---- */

/* _compute_patch_cost (mvs_patchmatch.py:323-390): depth map in, averaged cost out. */
int amvs_eval_cost(amvs_ctx *ctx, int ref, const int *src_ids, int n_src, int patch_size,
                   const float *depth_in, float *cost_out);
/* The stage before the box filter (mvs_patchmatch.py:341-377): every pixel projected into every
 * source at its own depth and sampled bilinearly.  sampled_out is [n_src][H][W] -- gray in exact
 * mode, 8-bit code units (gray * 255) in fast mode --, valid_out [H][W] holds bit s = source s
 * valid.  bounds: 0 = patch bounds (:362-363), 1 = image bounds (:516-517), 2 = depth test only
 * (dense_stereo.py:280,303).                                                                    */
int amvs_sample_sources(amvs_ctx *ctx, int ref, const int *src_ids, int n_src, int patch_size, int bounds,
                        const float *depth_in, float *sampled_out, uint8_t *valid_out);
/* _compute_confidence (mvs_patchmatch.py:493-534). */
int amvs_confidence(amvs_ctx *ctx, int ref, const int *src_ids, int n_src, int patch_size,
                    const float *depth_in, float *conf_out);
/* One pull step of _spatial_propagation (mvs_patchmatch.py:427-455): candidate at
 * (y,x) is the state at (y+oy, x+ox); state arrays are updated in place.          */
int amvs_propagate_step(amvs_ctx *ctx, int ref, const int *src_ids, int n_src, int patch_size,
                        float *depth, float *normal, float *cost,
                        int oy, int ox, float depth_min);
/* One sample of _random_refinement (mvs_patchmatch.py:470-489) with draw `draw`
 * of stream (seed, stream_view).                                                  */
int amvs_refine_step(amvs_ctx *ctx, int ref, const int *src_ids, int n_src, int patch_size,
                     float *depth, float *normal, float *cost,
                     uint64_t seed, uint32_t stream_view, uint32_t draw,
                     float depth_range, float normal_range,
                     float depth_min, float depth_max);
/* Initialisation (mvs_patchmatch.py:268-284) from draw 0 of stream (seed, stream_view). */
int amvs_init_state(amvs_ctx *ctx, uint64_t seed, uint32_t stream_view,
                    float log_depth_scale, float log_depth_min,
                    float *depth, float *normal, float *cost);
/* mean / variance maps of view `view` under a k x k zero-padded box filter
 * (mvs_patchmatch.py:403,406).                                                    */
int amvs_box_stats(amvs_ctx *ctx, int view, int patch_size, float *mean_out, float *var_out);
/* The counter-hash RNG that stands in for torch.rand / torch.randn
 * (mvs_patchmatch.py:271,279,280,471,475): per element one uniform (u_out, n
 * floats) and three normals (n_out, n x 3).  Either output may be NULL.           */
int amvs_rng_fill(amvs_ctx *ctx, uint64_t seed, uint32_t stream_view, uint32_t draw,
                  int64_t n, float *u_out, float *n_out);

/* ---- native exchange between ranks (amvs_comm_init / amvs_allgather_maps) ----
 * For a consumer of this ABI WITHOUT torch.distributed (the Python classes use torch's process group,
 * which issues the same RCCL calls).  One process per GPU; the reference has no counterpart (its loop
 * over views is serial, mvs_patchmatch.py:104-123).  RCCL is resolved at run time (dlopen of
 * librccl.so.1 -- the copy already in the process when a PyTorch-ROCm wheel loaded one, the system copy
 * otherwise), so libamvs.so carries no link-time dependency on it.
 *   amvs_comm_unique_id   rank 0 creates the 128-byte id; the caller sends it to the other ranks
 *                         (any transport: a file, MPI, a socket) -- ncclGetUniqueId
 *   amvs_comm_init        every rank, with the same id: ncclCommInitRank on the context's device
 *   amvs_allgather_maps   all-gather of `floats_per_rank` float32 from local_dev into full_dev
 *                         ([world][floats_per_rank], rank order), device pointers, enqueued on the
 *                         context's stream behind the sweep that produced local_dev -- ncclAllGather;
 *                         full_dev + rank * floats_per_rank may be local_dev itself (in place)
 *   amvs_comm_destroy     ncclCommDestroy (also done by amvs_destroy)                                 */
#define AMVS_COMM_ID_BYTES 128
int amvs_comm_unique_id(uint8_t id_out[AMVS_COMM_ID_BYTES]);
int amvs_comm_init(amvs_ctx *ctx, int rank, int world, const uint8_t id[AMVS_COMM_ID_BYTES]);
int amvs_allgather_maps(amvs_ctx *ctx, const void *local_dev, void *full_dev, int64_t floats_per_rank);
int amvs_comm_destroy(amvs_ctx *ctx);

/* utils.save_ply (utils.py:8-37): ASCII PLY with "%.6f %.6f %.6f %d %d %d" per vertex; the same
 * bytes as the reference writes, through one buffered native writer (no GPU involved; ctx-free).
 * points: n x 3 float64, colors: n x 3 int64 (the reference casts with .astype(int)).          */
int amvs_write_ply(const char *path, const double *points, const int64_t *colors, int64_t n);

/* Index-checked build (csrc/amvs_check.h, -DAMVS_CHECK_INDICES; amvs_version() then ends in "+index-checks"): the
 * GPU-side substitute for an address sanitizer, which this pool does not offer for device code.  Every
 * data-dependent global index of the sweep, plane-sweep, extended, fusion and neighbour-search kernels is compared
 * with its buffer's extent before the access; a violation is counted, the first is recorded and the access
 * redirected to a safe index.  report[0] = violations since the last reset, report[1] = translation unit << 32 |
 * source line of the first, report[2] = its index, report[3] = the extent; amvs_sync, amvs_patchmatch,
 * amvs_plane_sweep and amvs_fetch_cloud return AMVS_EINDEX while a violation is on record.  The shipped build
 * compiles the checks away: it reports zeros.  (The reference has no counterpart; test infrastructure of the
 * device code.)                                                                                             */
int amvs_index_check(uint64_t report[4], int reset);

/* Self test: the kernels replace the IEEE divide / sqrt expansions by v_rcp_f32 / v_rsq_f32 with
 * FMA corrections (plus an IEEE path for out-of-range operands).  Compares both against
 * 1.0f/x and sqrtf(x) on ALL 2^32 float bit patterns; mismatches[0] = reciprocal,
 * mismatches[1] = square root (both must be 0 for bit-exact parity with the tests' CPU checker).      */
int amvs_selftest_lean_math(amvs_ctx *ctx, uint64_t mismatches[2]);

#ifdef __cplusplus
}
#endif
#endif /* AMVS_H */
