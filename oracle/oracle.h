/*
 * oracle.h -- C API of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a dependency-free C++17 restatement of the reference's hot
 * path (ORBextractor / ORBmatcher / Optimizer) plus the OpenCV / g2o primitives
 * the reference merely calls out to.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped library
 * (orbslam_in_practice_amd/csrc) never includes, links or calls anything here.
 *
 * PARITY STATUS: "parity unpinned" at the OpenCV / g2o boundaries -- the
 * reference holds no tests, goldens or fixtures and
 * cannot be compiled here (no OpenCV / Eigen / g2o in the image).
 * Functions whose source IS in the reference follow it line by line and cite it.
 */
#ifndef ORBSLAM_ORACLE_H
#define ORBSLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint as a 28-byte POD (pt.x, pt.y, size, angle, response, octave, class_id) */
typedef struct OrcKeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} OrcKeyPoint;

typedef struct OrcOrbCfg {
    int32_t nfeatures;
    float scaleFactor;
    int32_t nlevels, iniThFAST, minThFAST;
} OrcOrbCfg;

/* trig_mode for the descriptor steering (reference src/ORBextractor.cpp:62-63):
 *   0 = libm cosf/sinf  (what the reference calls; platform-libm dependent)
 *   1 = deterministic double-polynomial sincos shared with the device code */
enum { ORC_TRIG_LIBM = 0, ORC_TRIG_DET = 1 };

void *orc_orb_create(const OrcOrbCfg *cfg);
void orc_orb_destroy(void *h);
/* a1: constructor tables. arrays sized nlevels (umax: 16). */
int orc_orb_tables(void *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                   int32_t *features_per_level, int32_t *umax16);
int orc_orb_level_sizes(void *h, int w, int hgt, int32_t *ws, int32_t *hs);
/* full operator() (a2..a9). returns 0, n_out = number of keypoints. */
int orc_orb_extract(void *h, const uint8_t *img, int w, int hgt, int stride, OrcKeyPoint *kps,
                    uint8_t *desc, int cap, int32_t *n_out, int trig_mode);
/* intermediates of the last orc_orb_extract call */
int orc_orb_get_level(void *h, int level, uint8_t *out);      /* un-blurred pyramid level, tight */
int orc_orb_get_blur(void *h, int level, uint8_t *out);       /* 7x7 sigma=2 blurred level       */
int orc_orb_get_candidates(void *h, int level, OrcKeyPoint *out, int cap); /* FAST output, returns count */
int orc_orb_get_level_count(void *h, int level);              /* keypoints kept by the quadtree  */

/* primitives (OpenCV restatements), for unit tests */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                          int dstride);
void orc_border_reflect101_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int border);
void orc_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
int orc_fast9_16(const uint8_t *roi, int w, int h, int stride, int threshold, int nonmax, OrcKeyPoint *out,
                 int cap);
int orc_fast_corner_score(const uint8_t *center, int stride, int threshold);
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t *img, int stride, int x, int y);
int orc_distribute_octtree(const OrcKeyPoint *in, int n, int minX, int maxX, int minY, int maxY, int N,
                           OrcKeyPoint *out, int cap);
void orc_sincos_det(float angle_rad, float *c, float *s);

/* matcher (a10..a12) */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b);
void orc_hamming_knn2(const uint8_t *d1, int n1, const uint8_t *d2, int n2, int32_t *idx, uint16_t *best,
                      uint16_t *second);
typedef struct OrcFrameView {
    const OrcKeyPoint *kps; /* undistorted keypoints */
    const uint8_t *desc;    /* n x 32 */
    int32_t n;
    float min_x, max_x, min_y, max_y; /* image bounds (Frame::miMinX..) */
} OrcFrameView;
/* clone_mode: 1 = literally clone the whole descriptor Mat per GetDescriptors() call
 * (reference include/Frame.h:39-42), 0 = no clone; results identical, timing differs.
 * window_size < 0 means "infinite window, all octaves, ascending i2" (the a12 variant). */
void orc_features_in_area(const OrcFrameView *f, const float *qxyr, const int32_t *qlevels, int nq, int32_t *out_idx, int cap,
                          int32_t *out_count);
int orc_search_for_initialization(const OrcFrameView *f1, const OrcFrameView *f2, float *prev_matched_xy,
                                  int32_t *matches12, int window_size, float nnratio, int check_ori,
                                  int clone_mode);
/* SURVEY N4 (empty bodies in the reference: upstream semantics restated, parity unpinned by construction) */
int orc_search_by_projection(const OrcFrameView *cur, const OrcFrameView *last, const uint8_t *last_flags, const double *last_xw,
                             const double *Rcw, const double *tcw, double fx, double fy, double cx, double cy,
                             const float *scale_factors, int nlevels, uint8_t *cur_taken, float th, int check_ori,
                             int32_t *cur_match);
int orc_search_by_bow(const OrcFrameView *kf, const uint8_t *kf_valid, const OrcFrameView *f, float nnratio, int check_ori,
                      int32_t *f_match);
void orc_compute_three_maxima(const int32_t *histo_sizes, int L, int32_t *ind1, int32_t *ind2,
                              int32_t *ind3);

/* ---- bundle adjustment (a13..a15); same layout as BAProblem / BAResult of include/orbslam_amd.h ---- */
typedef struct OrcBAProblem {
    int32_t n_cams, n_points, n_obs;
    const double *poses;
    const uint8_t *fixed;
    const double *points;
    const double *obs;
    const int32_t *cam_idx;
    const int32_t *pt_idx;
    const double *inv_sigma2;
    double fx, fy, cx, cy;
    double huber_delta;
    int32_t n_rounds;
    int32_t rounds[4];
    double chi2_threshold;
    int32_t drop_robust_after;
    volatile const uint8_t *stop;
} OrcBAProblem;

typedef struct OrcBAResult {
    double *poses;
    double *points;
    uint8_t *outlier;
    double chi2_initial, chi2_final;
    int32_t iterations, lin_solves;
} OrcBAResult;

int orc_ba_solve(const OrcBAProblem *p, OrcBAResult *r);
int orc_pose_optimization(double *pose7, const double *xw, const double *obs, const double *inv_sigma2, int n,
                          double fx, double fy, double cx, double cy, uint8_t *outlier, int32_t *n_inliers);

#ifdef __cplusplus
}
#endif
#endif
