/*
 * orbslam_amd.h -- C-ABI of the MI355X (gfx950) hot path of Sunny-cpp/orbslam_In_practice.
 *
 * Plain C, POD only: no cv::, STL, Eigen or torch types cross this boundary.  The thin C++ classes in
 * include/orbslam_amd/{ORBextractor,ORBmatcher,Optimizer}.h keep the reference's own call signatures
 * and forward to these entry points.
 *
 * Every function returns 0 on success or a negative ORB_E_* code; all output buffers are caller
 * allocated.  A handle owns its device memory, pinned staging and one HIP stream; use one handle per
 * host thread (the reference's ORBextractor is likewise non re-entrant: include/ORBextractor.h:71).
 *
 * There is NO CPU fallback: if no HIP device is usable, create() fails with ORB_E_NO_DEVICE.
 */
#ifndef ORBSLAM_AMD_H
#define ORBSLAM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORB_OK = 0,
    ORB_E_INVALID = -1,   /* bad argument (null pointer, non-positive size, unsupported config)        */
    ORB_E_CAPACITY = -2,  /* caller's output capacity too small / batch larger than the handle's max  */
    ORB_E_NO_DEVICE = -3, /* no usable HIP device                                                      */
    ORB_E_HIP = -4,       /* a HIP runtime call failed (orb_last_error() has the text)                 */
    ORB_E_ASPECT = -5,    /* image so tall that the reference's quadtree root count rounds to 0 (the
                             reference divides by zero there, src/ORBextractor.cpp:493-495)           */
    ORB_E_STOPPED = -6,   /* ba_solve: *stop became true before the first iteration                    */
    ORB_E_RESERVED_7 = -7, /* retired (was ORB_E_SINGULAR in round 1): never returned, never reused        */
    ORB_E_BUSY = -8       /* ba_submit_batch: all three batch slots in flight (call ba_wait first)     */
};

const char *orb_last_error(void);
/* number of HIP devices visible (0 if none); never fails */
int orb_device_count(void);

/* cv::KeyPoint as a 28-byte POD: pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct OrbKeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} OrbKeyPoint;

/* ctor arguments of ORBSlam::ORBextractor (reference include/ORBextractor.h:35-36) */
typedef struct OrbConfig {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
} OrbConfig;

typedef struct OrbExtractor OrbExtractor; /* opaque */

/* ---- ORBextractor (replaces reference src/ORBextractor.cpp:360-420 ctor) ----------------------
 * max_w/max_h/max_batch size the device buffers; a larger frame grows them on first use, a larger batch
 * is refused with ORB_E_CAPACITY. */
int orb_create(const OrbConfig *cfg, int max_w, int max_h, int max_batch, int device_id, OrbExtractor **out);
void orb_destroy(OrbExtractor *h);

/* the six getters of include/ORBextractor.h:47-69 plus mnFeaturesPerLevel / umax (a1) */
int orb_get_levels(const OrbExtractor *h);
float orb_get_scale_factor(const OrbExtractor *h);
int orb_get_tables(const OrbExtractor *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                   int32_t *features_per_level, int32_t *umax16);
/* pyramid level sizes for a w x h input (src/ORBextractor.cpp:1075-1077) */
int orb_get_level_sizes(const OrbExtractor *h, int w, int hgt, int32_t *ws, int32_t *hs);
/* worst-case number of keypoints one frame can return: per level max(quota + 4, 4 * nIni), nIni = round(W / H) quadtree
 * roots (src/ORBextractor.cpp:493-534 -- a wide frame with a small budget returns more than its budget).  The bound is
 * that of the frame size the handle currently holds: the creation size until a call with another size, after which
 * (also after an ORB_E_CAPACITY from such a call) it reports the new one. */
int orb_max_keypoints(const OrbExtractor *h);

/* ---- ORBextractor::operator() (replaces src/ORBextractor.cpp:1001-1065) -------------------------
 * gray: 8-bit single channel, `stride` bytes per row.  kps_out[cap], desc_out[cap*32].  An empty
 * image (w<=0||h<=0||gray==NULL) returns ORB_OK with *n_out = 0 and leaves the outputs untouched
 * (reference :1004-1005). */
int orb_extract(OrbExtractor *h, const uint8_t *gray, int w, int hgt, int stride, OrbKeyPoint *kps_out,
                uint8_t *desc_out, int cap, int32_t *n_out);

/* n_frames independent frames of one size (BASELINE config 4).  imgs[f] -> frame f (host memory);
 * kps_out[n_frames*cap], desc_out[n_frames*cap*32], n_out[n_frames]. */
int orb_extract_batch(OrbExtractor *h, const uint8_t *const *imgs, int n_frames, int w, int hgt, int stride,
                      OrbKeyPoint *kps_out, uint8_t *desc_out, int cap, int32_t *n_out);

/* Same, but every pointer is DEVICE memory (frames at d_imgs + f*frame_stride) and the work is only
 * enqueued on `stream` (a hipStream_t, NULL = the handle's own stream); nothing is copied to the host.
 * d_kps[n_frames*cap], d_desc[n_frames*cap*32], d_counts[n_frames]. */
int orb_extract_batch_device(OrbExtractor *h, const uint8_t *d_imgs, int n_frames, int w, int hgt, int stride,
                             int64_t frame_stride, OrbKeyPoint *d_kps, uint8_t *d_desc, int cap,
                             int32_t *d_counts, void *stream);

/* mvImagePyramid (public member, include/ORBextractor.h:71): level `level` of frame `frame` of the
 * most recent extract call, WITH its 19-px BORDER_REFLECT_101 apron (src/ORBextractor.cpp:1078-1093).
 * out must hold (w_l+38)*(h_l+38) bytes.  blurred!=0 returns the 7x7 sigma=2 blurred level without
 * apron (w_l*h_l bytes) instead (src/ORBextractor.cpp:1045-1046). */
int orb_download_level(OrbExtractor *h, int frame, int level, int blurred, uint8_t *out);
/* FAST candidates of (frame, level) before the quadtree, in the reference's order
 * (src/ORBextractor.cpp:745-786); returns the count through n_out even if cap is too small. */
int orb_download_candidates(OrbExtractor *h, int frame, int level, OrbKeyPoint *out, int cap, int32_t *n_out);
/* measurement hooks: with profiling enabled every orb_extract_batch_device call records HIP events on
 * the launch stream between its stages; orb_get_stage_ms waits for the last call and returns the five
 * stage durations in ms: {pyramid (copy + resizes), FAST cells, quadtree, blur, describe}. */
int orb_set_profiling(OrbExtractor *h, int enable);
int orb_get_stage_ms(OrbExtractor *h, float *ms5);
/* wait for everything enqueued on the handle's stream */
int orb_synchronize(OrbExtractor *h);
/* the handle's hipStream_t */
void *orb_stream(OrbExtractor *h);

/* ---- ORBmatcher ----------------------------------------------------------------------------------
 * orb_descriptor_distance: ORBmatcher::DescriptorDistance (src/ORBmatcher.cpp:128-144), host-side. */
int orb_descriptor_distance(const uint8_t *a, const uint8_t *b);
/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cpp:147-188) on the 30 bin sizes, host-side. */
void orb_compute_three_maxima(const int32_t *histo_sizes, int L, int32_t *ind1, int32_t *ind2, int32_t *ind3);

typedef struct OrbMatcher OrbMatcher; /* opaque: device buffers + stream for matching */
int orb_matcher_create(int max_keypoints, int max_pairs, int device_id, OrbMatcher **out);
void orb_matcher_destroy(OrbMatcher *m);
void *orb_matcher_stream(OrbMatcher *m);
/* same hook for the matcher: duration in ms of the last orb_match_batch_device kernel */
int orb_matcher_set_profiling(OrbMatcher *m, int enable);
int orb_matcher_get_ms(OrbMatcher *m, float *ms);
/* the same average per kernel: ms2[0] = phase 1 (all-candidates top-8 scan), ms2[1] = phase 2 (greedy resolve) */
int orb_matcher_get_stage_ms(OrbMatcher *m, float *ms2);
int orb_matcher_synchronize(OrbMatcher *m);

/* brute-force Hamming 2-NN (north_star): for every row of d1[n1*32] the best and second
 * best distance over all rows of d2[n2*32]; ties -> lowest index; second counts duplicates; n2==0
 * gives idx=-1, best=second=0xFFFF.  Host pointers. */
int orb_hamming_knn2(OrbMatcher *m, const uint8_t *d1, int n1, const uint8_t *d2, int n2, int32_t *idx,
                     uint16_t *best, uint16_t *second);

/* one frame as the matcher sees it (Frame::GetUnKeyPts / GetDescriptors / image bounds) */
typedef struct OrbFrameView {
    const OrbKeyPoint *kps; /* undistorted keypoints, n entries                  */
    const uint8_t *desc;    /* n x 32 bytes                                       */
    int32_t n;
    float min_x, max_x, min_y, max_y; /* Frame::miMinX.. (src/Frame.cpp:111-141) */
} OrbFrameView;

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cpp:9-126) with Frame::GetFeaturesInArea
 * (src/Frame.cpp:219-271).  prev_matched_xy[2*n1] in/out, matches12[n1] out (-1 = none); returns the
 * number of matches through n_matches.  window_size < 0 selects the brute-force variant (every
 * octave, every keypoint of F2 in ascending index order as candidates). */
int orb_match_init(OrbMatcher *m, const OrbFrameView *f1, const OrbFrameView *f2, float *prev_matched_xy,
                   int32_t *matches12, int window_size, float nnratio, int check_orientation,
                   int32_t *n_matches);

/* Frame::AssignFeaturesToGrid + Frame::GetFeaturesInArea (reference src/Frame.cpp:144-168, 219-271)
 * for a batch of queries against one frame's keypoints, on the device: queries_xyr[3*q] = (x, y, r),
 * query_levels[2*q] = (minLevel, maxLevel) or NULL for (-1, -1) (no level filter, as :245 computes it).
 * out_idx[q*cap + k] receives the indices the reference returns, in its order (grid cell column-major, then
 * insertion order), -1 padded; out_count[q] the full count (entries beyond cap are dropped).  Host pointers;
 * frames of up to 16384 keypoints (ORB_E_CAPACITY beyond). */
int orb_features_in_area(OrbMatcher *m, const OrbFrameView *f, const float *queries_xyr, const int32_t *query_levels,
                         int n_queries, int32_t *out_idx, int cap, int32_t *out_count);


/* ---- ORBmatcher::SearchByProjection / SearchByBoW --------------------------------------------------------
 * Both have EMPTY bodies in the reference (include/ORBmatcher.h:22,24) although Tracking calls them
 * (src/Tracking.cpp:298,344,348): there is no reference behaviour to be identical to.  These entry points implement
 * upstream ORB-SLAM2's monocular semantics on the reference's data model (oracle/match_oracle.cpp states them line
 * by line): PARITY UNPINNED by construction.  The candidate scan (projection, window / level filter, Hamming
 * distances, the closest candidates per feature) runs on the device; for these HOST-POINTER entry points the sequential walk
 * over the features (a feature taken by an earlier one is closed to later ones) and the rotation histogram run on the host
 * (10-20 us there; the device walk of orb_search_by_projection_device needs ~70 single-wavefront rounds, and the lists are
 * coming back with the same copy anyway).  ORBSLAM_AMD_N4_DEVICE_WALK=1 runs the device walk here too (tests).
 *
 * orb_search_by_projection: last_flags[n_last]: bit0 = the feature holds a map point and is not an outlier, bit1 = that
 * map point has observations; last_xw[n_last*3] world positions (read where bit0 is set); Rcw9 / tcw3 the current
 * frame's pose (x_c = Rcw x_w + tcw); scale_factors[nlevels]; cur_taken[n_cur] in/out: the feature holds a map point
 * with observations; cur_match[n_cur] out: index of the last-frame feature whose map point it receives, or -1.
 * TH_HIGH = 100; rotation histogram as in the reference's SearchForInitialization. */
int orb_search_by_projection(OrbMatcher *m, const OrbFrameView *cur, const OrbFrameView *last, const uint8_t *last_flags,
                             const double *last_xw, const double *Rcw9, const double *tcw3, double fx, double fy, double cx,
                             double cy, const float *scale_factors, int nlevels, uint8_t *cur_taken, float th,
                             int check_orientation, int32_t *cur_match, int32_t *n_matches);
/* The same search on DEVICE-RESIDENT frames: feature slabs as orb_extract_batch_device left them (d_cur_kps / d_cur_desc /
 * n_cur of the current frame, d_last_* of the last one), the last frame's flags and world positions, the in/out taken
 * flags and the results all in device memory; Rcw9 / tcw3 / scale_factors are host arrays read before the call returns.
 * Projection, candidate scan, the sequential walk (speculative rounds on one wavefront, k_n4_walk) and the rotation
 * histogram run as three kernels on `stream` (NULL = the matcher's): no copy, no synchronisation.  Identical results to
 * orb_search_by_projection.  ORB_E_CAPACITY beyond ~8000 features per frame (the walk keeps its state in LDS).
 * One search in flight per matcher handle (its scratch holds the candidate lists).  d_n_cur: NULL (n_cur is the exact count)
 * or the current frame's count in DEVICE memory as the extractor wrote it -- n_cur is then only the capacity of the slab,
 * and a caller chaining extraction and search never has to wait for the count (features at and beyond the count keep
 * whatever d_cur_match / d_cur_taken held). */
int orb_search_by_projection_device(OrbMatcher *m, const OrbKeyPoint *d_cur_kps, const uint8_t *d_cur_desc, int n_cur,
                                    const int32_t *d_n_cur, const OrbKeyPoint *d_last_kps, const uint8_t *d_last_desc, int n_last,
                                    const uint8_t *d_last_flags, const double *d_last_xw, const double *Rcw9, const double *tcw3,
                                    double fx, double fy, double cx, double cy, const float *scale_factors, int nlevels,
                                    float min_x, float max_x, float min_y, float max_y, uint8_t *d_cur_taken, float th,
                                    int check_orientation, int32_t *d_cur_match, int32_t *d_n_matches, void *stream);
/* orb_search_by_bow: the reference has no vocabulary (ComputeBoW is empty), i.e. one node holding every feature: for every
 * keyframe feature with kf_valid[i] != 0, in index order, best and second best distance over the frame features not
 * matched yet; accepted if best <= TH_LOW (50) and best < nnratio * second.  f_match[n_f] out: keyframe feature or -1. */
int orb_search_by_bow(OrbMatcher *m, const OrbFrameView *kf, const uint8_t *kf_valid, const OrbFrameView *f, float nnratio,
                      int check_orientation, int32_t *f_match, int32_t *n_matches);

/* Batched, device-resident matching of n_pairs frame pairs (pair p: frame a[p] against frame b[p] of
 * the slabs an extractor produced with capacity `cap`): d_kps[n_frames*cap], d_desc[n_frames*cap*32],
 * d_counts[n_frames].  prev_matched is taken to be F1's own keypoint positions (first call of
 * Tracking::MonocularInitialization, src/Tracking.cpp:170-189).  d_matches12[n_pairs*cap],
 * d_nmatches[n_pairs].  Enqueued on `stream` (NULL = matcher's stream). */
int orb_match_batch_device(OrbMatcher *m, const OrbKeyPoint *d_kps, const uint8_t *d_desc,
                           const int32_t *d_counts, int cap, const int32_t *d_pair_a, const int32_t *d_pair_b,
                           int n_pairs, float min_x, float max_x, float min_y, float max_y, int window_size,
                           float nnratio, int check_orientation, int32_t *d_matches12, int32_t *d_nmatches,
                           void *stream);

/* ---- Optimizer -----------------------------------------------------------------------------------
 * One bundle-adjustment problem in g2o's terms (reference src/Optimizer.cpp:23-129): SE3 pose
 * vertices (qw,qx,qy,qz,tx,ty,tz; world->camera, the reference's "Rwc/twc"), XYZ point vertices
 * (marginalised), one EdgeSE3ProjectXYZ per observation. */
typedef struct BAProblem {
    int32_t n_cams, n_points, n_obs;
    const double *poses;      /* n_cams x 7                                              */
    const uint8_t *fixed;     /* n_cams: 1 = vertex fixed (KF id 0, src/Optimizer.cpp:52) */
    const double *points;     /* n_points x 3                                            */
    const double *obs;        /* n_obs x 2 (undistorted keypoint)                        */
    const int32_t *cam_idx;   /* n_obs                                                   */
    const int32_t *pt_idx;    /* n_obs                                                   */
    const double *inv_sigma2; /* n_obs: information = I * inv_sigma2                      */
    double fx, fy, cx, cy;
    double huber_delta;       /* <= 0: no robust kernel (bRobust=false)                   */
    /* schedule: rounds[i] LM iterations; after round i (if i+1 < n_rounds) edges with chi2 >
     * chi2_threshold or non-positive depth are switched off (level 1) and, if drop_robust_after == i,
     * the robust kernel is removed from the rest.  GlobalBA: n_rounds=1, rounds={nIterations};
     * LocalBA: n_rounds=2, rounds={5,10}, chi2_threshold=5.991, drop_robust_after=0. */
    int32_t n_rounds;
    int32_t rounds[4];
    double chi2_threshold;
    int32_t drop_robust_after;
    volatile const uint8_t *stop; /* optional pbStopFlag, polled between iterations */
} BAProblem;

typedef struct BAResult {
    double *poses;       /* n_cams x 7 out   */
    double *points;      /* n_points x 3 out */
    uint8_t *outlier;    /* n_obs out (1 = switched off), may be NULL */
    double chi2_initial; /* robustified chi2 before / after */
    double chi2_final;
    int32_t iterations;  /* LM iterations actually run */
    int32_t lin_solves;  /* linear systems solved (incl. rejected LM trials) */
} BAResult;

typedef struct BASolver BASolver; /* opaque */
/* max_batch is enforced (ORB_E_CAPACITY beyond it); max_cams / max_points / max_obs are sizing hints only -- every
 * buffer grows on demand and there is no limit on the number of cameras (g2o has none either): a reduced system
 * whose Cholesky panel does not fit the LDS is solved by the in-kernel PCG, with its vectors in HBM if need be. */
int ba_create(int max_cams, int max_points, int max_obs, int max_batch, int device_id, BASolver **out);
void ba_destroy(BASolver *s);
/* Optimizer::BundleAdjustment / LocalBundleAdjustment numerical core.  ONE window (the LocalMapping call pattern)
 * is spread over a team of up to 64 workgroups with device-side barriers between the phases of an iteration. */
int ba_solve(BASolver *s, const BAProblem *p, BAResult *r);
/* n independent windows (BASELINE config 3 / target), solved concurrently on the device */
int ba_solve_batch(BASolver *s, const BAProblem *p, BAResult *r, int n);
/* The same in two halves, for a steady stream of batches: ba_submit_batch stages the batch on the host and enqueues
 * its uploads, kernels and download; ba_wait blocks until the OLDEST submitted batch has finished and fills its
 * BAResult array.  Up to THREE batches may be in flight (each slot has its own pinned and device buffers): the host side of
 * the next batch and the download of the previous one overlap the kernels of the running one, and the device queue always
 * holds a staged batch beyond it.  A fourth ba_submit_batch without a ba_wait returns ORB_E_BUSY.  p / r and everything they point to must stay valid until the matching ba_wait.
 * A batch that carries a stop flag has to be polled between iterations and runs to completion inside submit. */
int ba_submit_batch(BASolver *s, const BAProblem *p, BAResult *r, int n);
int ba_wait(BASolver *s);
/* measurement hooks: with profiling enabled the device work of every optimize() round is bracketed by HIP events on
 * the solver's stream; after a batch has finished, ba_get_iteration_ms returns the average duration of these
 * intervals, their number, and the workgroups per window (team size) the fused iteration kernel ran with -- 0 when
 * the batch ran as phase kernels (six or more windows -- four when none has 16000 observations --, all on the dense Cholesky: an LM trial is then a
 * sequence of kernels over all windows instead of one kernel per window). */
int ba_set_profiling(BASolver *s, int enable);
int ba_get_iteration_ms(BASolver *s, float *avg_ms, int32_t *n_launches, int32_t *team);

/* Optimizer::PoseOptimization numerical core (src/Optimizer.cpp:131-261): one free pose, n unary
 * edges with fixed Xw; 4 rounds x 10 iterations with chi2 relabelling; robust kernel dropped after
 * round index 2.  pose[7] in/out, outlier[n] out; returns inliers through n_inliers. */
int ba_pose_optimization(BASolver *s, double *pose, const double *xw, const double *obs,
                         const double *inv_sigma2, int n, double fx, double fy, double cx, double cy,
                         uint8_t *outlier, int32_t *n_inliers);
/* PoseOptimization of the correspondences a device-resident SearchByProjection produced: the pose problem of
 * Optimizer::PoseOptimization(Frame*) (src/Optimizer.cpp:161-200) built on the device from the current frame's features j
 * with d_cur_match[j] >= 0, in index order -- world point d_last_xw[3 * match], measurement = the keypoint position,
 * information = inv_sigma2_levels[octave] -- and solved by the same kernel as ba_pose_optimization.  d_cur_kps[cap_cur] /
 * d_n_cur as an extractor left them in device memory; d_pose7 in (initial estimate) / out; d_outlier[cap_cur] per CURRENT
 * feature (0 where it has no correspondence); d_n_inliers, d_n_corr (may be NULL) out.  Everything d_* is device memory,
 * inv_sigma2_levels a host array read before the call returns; enqueued on `stream` (NULL = the solver's), no copy, no
 * wait.  Uses the solver's pose buffers: one pose call in flight per handle. */
int ba_pose_optimization_matches_device(BASolver *s, const OrbKeyPoint *d_cur_kps, int cap_cur, const int32_t *d_n_cur,
                                        const int32_t *d_cur_match, const double *d_last_xw, double *d_pose7,
                                        const double *inv_sigma2_levels, int nlevels, double fx, double fy, double cx, double cy,
                                        uint8_t *d_outlier, int32_t *d_n_inliers, int32_t *d_n_corr, void *stream);
/* n_problems independent pose problems in one launch (Tracking calls PoseOptimization up to three times per
 * frame: src/Tracking.cpp:309,355,382).  Problem p owns edges [offsets[p], offsets[p+1]) of xw / obs /
 * inv_sigma2 / outlier; poses[7*n_problems] in/out; n_inliers[n_problems] out. */
int ba_pose_optimization_batch(BASolver *s, double *poses, const double *xw, const double *obs,
                               const double *inv_sigma2, const int32_t *offsets, int n_problems, double fx, double fy,
                               double cx, double cy, uint8_t *outlier, int32_t *n_inliers);

/* ---- One frame of Tracking::TrackWithMotionModel as ONE device chain (round 3) ----------------------------------------
 * The reference's caller makes three calls per frame -- Frame(im, &extractor) (src/Frame.cpp:77), SearchByProjection
 * (src/Tracking.cpp:344) and PoseOptimization (:355) -- each a host round trip through cv:: containers, and the drop-in
 * classes keep that shape.  A caller that can take the three results together uses an OrbTracker: the image and the
 * previous frame's map-point association go up in one copy, the extracted features STAY in device memory (they are the
 * next call's "last frame"), projection / candidate scan / sequential walk / rotation histogram and the pose problem built
 * from the matches follow on one stream without a wait, and keypoints, descriptors, match indices, outlier labels, pose and
 * counts come back in one copy: one synchronisation per frame.  Same kernels, same arithmetic, same results as the three
 * calls.  The tracker borrows the three handles (one frame in flight; do not use them concurrently from elsewhere).
 *
 * orb_tracker_track: gray/w/h/stride as orb_extract.  prev_flags[n_prev] / prev_xw[n_prev * 3]: the map-point association of
 * the PREVIOUS frame's features (bit0 = holds a map point and is not an outlier, bit1 = the point has observations; world
 * position read where bit0 is set); n_prev = 0 on the first frame (or after orb_tracker_reset): extraction only.  pose_pred7:
 * the predicted pose (qw qx qy qz tx ty tz, world -> camera).  Image bounds as Frame::FindimageBound gives them.  Outputs:
 * kps_out / desc_out / n_out as orb_extract (cap >= orb_max_keypoints of the extractor); cur_match[n] = index of the previous
 * frame's feature whose map point the feature receives, or -1; outlier[n] = PoseOptimization's label of that
 * correspondence; pose_out7; n_matches (SearchByProjection's count), n_inliers (PoseOptimization's return value).  The
 * retry with a wider window when fewer than 20 matches are found (src/Tracking.cpp:347-351) is the caller's second call
 * (orb_search_by_projection on the returned features): rare, and its inputs are in the caller's hands already.
 * Keypoint positions are the measurements: frames are taken to be undistorted. */
typedef struct OrbTracker OrbTracker;
int orb_tracker_create(OrbExtractor *ext, OrbMatcher *mat, BASolver *sol, int device_id, OrbTracker **out);
void orb_tracker_destroy(OrbTracker *t);
int orb_tracker_reset(OrbTracker *t);
int orb_tracker_track(OrbTracker *t, const uint8_t *gray, int w, int h, int stride, const uint8_t *prev_flags, const double *prev_xw,
                      int n_prev, const double *pose_pred7, double fx, double fy, double cx, double cy, float min_x, float max_x,
                      float min_y, float max_y, float th, int check_orientation, OrbKeyPoint *kps_out, uint8_t *desc_out, int cap,
                      int32_t *n_out, int32_t *cur_match, uint8_t *outlier, double *pose_out7, int32_t *n_matches,
                      int32_t *n_inliers);
/* The same call with a LOOK-AHEAD for callers that hold the next frame already (a recorded sequence, a camera queue):
 * next_gray (same w / h / stride; NULL = none) is uploaded and run through the extractor on a second stream WHILE this
 * frame's search and pose problem run, and the call that then passes that very buffer as `gray` finds its features in
 * device memory (a call with any other buffer drops the look-ahead).  Extraction does not depend on the pose, so the
 * results are those of orb_tracker_track bit for bit; the buffer behind next_gray must stay unchanged until that call. */
int orb_tracker_track_prefetch(OrbTracker *t, const uint8_t *gray, int w, int h, int stride, const uint8_t *prev_flags,
                               const double *prev_xw, int n_prev, const double *pose_pred7, double fx, double fy, double cx, double cy,
                               float min_x, float max_x, float min_y, float max_y, float th, int check_orientation,
                               OrbKeyPoint *kps_out, uint8_t *desc_out, int cap, int32_t *n_out, int32_t *cur_match, uint8_t *outlier,
                               double *pose_out7, int32_t *n_matches, int32_t *n_inliers, const uint8_t *next_gray);

#ifdef __cplusplus
}
#endif
#endif /* ORBSLAM_AMD_H */
