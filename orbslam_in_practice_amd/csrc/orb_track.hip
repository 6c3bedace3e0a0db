/*
 * orb_track.hip -- one frame of Tracking::TrackWithMotionModel (reference src/Tracking.cpp:330-372) as ONE device chain.
 *
 * The reference's caller makes three calls per frame -- Frame(im, &extractor) (src/Frame.cpp:77), SearchByProjection
 * (:344) and PoseOptimization (:355) -- and through host containers each of them is a host round trip: upload, kernels,
 * download, wait.  The drop-in classes keep that shape (they must return cv:: containers after every call).  For a
 * caller that can take the three results together, OrbTracker chains the same kernels on one stream: the image goes up
 * with the previous frame's map-point association in one copy, the extracted features STAY in HBM (they are the next
 * call's "last frame"), projection / candidate scan / walk / rotation histogram (orb_search_by_projection_device) and the
 * pose problem built from the matches (ba_pose_optimization_matches_device) follow without a wait, and everything the
 * caller needs -- keypoints, descriptors, match indices, outlier labels, pose, counts -- comes back in one copy.  One
 * synchronisation per frame instead of three; the same arithmetic, so the same matches and the same trajectory
 * (tests/test_gpu_tracker.py against the oracle's sequence).
 *
 * Built on the public C-ABI of the three handles; owns only its staging blocks and a stream.
 */
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbslam_amd.h"
#include "orb_common.h"

static OrbErrSlot t_err;
#define TCHK(expr)                                                      \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if (_e != hipSuccess) {                                         \
            t_err = std::string(#expr) + ": " + hipGetErrorString(_e);  \
            return ORB_E_HIP;                                           \
        }                                                               \
    } while (0)

struct OrbTracker {
    OrbExtractor *ext = nullptr;
    OrbMatcher *mat = nullptr;
    BASolver *sol = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    int cap = 0;      /* feature slots per frame */
    int nlevels = 0;
    std::vector<float> scale;
    std::vector<double> inv_sigma2;
    uint8_t *d_in = nullptr, *h_in = nullptr; /* [prev flags | prev xw | pose] */
    size_t in_cap = 0;
    uint8_t *d_img[2] = {nullptr, nullptr}, *h_img[2] = {nullptr, nullptr}; /* this frame's image / the frame fetched ahead */
    size_t img_cap = 0;
    int img_i = 0;  /* image buffer of the NEXT upload */
    uint8_t *d_out[3] = {nullptr, nullptr, nullptr}, *h_out = nullptr; /* per frame: see Layout; ring: previous, current, ahead */
    size_t out_bytes = 0;
    int cur = 0;    /* out block of the NEXT frame */
    int n_prev = -1; /* features of the previous frame (-1: none yet) */
    /* look-ahead (orb_tracker_track_prefetch): the next frame's upload and extraction on a second stream, beside this
       frame's search and pose problem */
    hipStream_t pf_stream = nullptr;
    hipEvent_t ev_ext = nullptr, ev_pf = nullptr;
    bool pf_valid = false;
    const uint8_t *pf_ptr = nullptr;
    int pf_w = 0, pf_h = 0, pf_stride = 0;
};

/* one frame's device block: everything the chain writes, contiguous, so that it comes back as one copy */
struct TrackLayout {
    size_t count, kps, desc, match, taken, nm, pose, outlier, ninl, ncorr, total;
    explicit TrackLayout(int cap) {
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        count = 0;
        kps = al(64);
        desc = al(kps + (size_t)cap * sizeof(OrbKeyPoint));
        match = al(desc + (size_t)cap * 32);
        taken = al(match + (size_t)cap * 4);
        nm = al(taken + cap);
        pose = al(nm + 64);
        outlier = al(pose + 64);
        ninl = al(outlier + cap);
        ncorr = ninl + 8;
        total = al(ncorr + 8);
    }
};

extern "C" void orb_tracker_destroy(OrbTracker *t);
extern "C" int orb_tracker_create(OrbExtractor *ext, OrbMatcher *mat, BASolver *sol, int device_id, OrbTracker **out) {
    if (!ext || !mat || !sol || !out) return ORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ORB_E_NO_DEVICE;
    TCHK(hipSetDevice(device_id));
    OrbTracker *t = new OrbTracker();
    t->ext = ext;
    t->mat = mat;
    t->sol = sol;
    t->device = device_id;
    t->nlevels = orb_get_levels(ext);
    t->scale.resize(t->nlevels);
    std::vector<float> inv_s2(t->nlevels), tmp(t->nlevels);
    orb_get_tables(ext, t->scale.data(), tmp.data(), tmp.data(), inv_s2.data(), nullptr, nullptr);
    t->inv_sigma2.assign(inv_s2.begin(), inv_s2.end()); /* (Optimizer::SetLevelSigma2 takes the extractor's float table too) */
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&t->pf_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&t->ev_ext, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t->ev_pf, hipEventDisableTiming) != hipSuccess) {
        orb_tracker_destroy(t);
        return ORB_E_HIP;
    }
    *out = t;
    return ORB_OK;
}

extern "C" void orb_tracker_destroy(OrbTracker *t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->pf_stream) (void)hipStreamSynchronize(t->pf_stream);
    (void)hipFree(t->d_in);
    if (t->h_in) (void)hipHostFree(t->h_in);
    for (int b = 0; b < 2; b++) {
        (void)hipFree(t->d_img[b]);
        if (t->h_img[b]) (void)hipHostFree(t->h_img[b]);
    }
    for (int b = 0; b < 3; b++) (void)hipFree(t->d_out[b]);
    if (t->h_out) (void)hipHostFree(t->h_out);
    if (t->ev_ext) (void)hipEventDestroy(t->ev_ext);
    if (t->ev_pf) (void)hipEventDestroy(t->ev_pf);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    if (t->pf_stream) (void)hipStreamDestroy(t->pf_stream);
    delete t;
}

extern "C" int orb_tracker_reset(OrbTracker *t) {
    if (!t) return ORB_E_INVALID;
    t->n_prev = -1;
    if (t->pf_valid) { /* a frame fetched ahead belongs to the sequence that ends here */
        (void)hipSetDevice(t->device);
        (void)hipStreamSynchronize(t->pf_stream);
        t->pf_valid = false;
    }
    return ORB_OK;
}

static int tracker_buffers(OrbTracker *t, int w, int h, int n_prev) {
    const int need_cap = orb_max_keypoints(t->ext);
    if (need_cap > t->cap || !t->d_out[0]) { /* (the extractor's bound can grow with the frame shape: its geometry is set by the first call) */
        TCHK(hipStreamSynchronize(t->stream));
        TCHK(hipStreamSynchronize(t->pf_stream));
        t->pf_valid = false;
        const int cap = std::max(need_cap, t->cap);
        const TrackLayout L(cap);
        for (int b = 0; b < 3; b++) {
            if (t->d_out[b]) TCHK(hipFree(t->d_out[b]));
            t->d_out[b] = nullptr;
            TCHK(hipMalloc((void **)&t->d_out[b], L.total));
            TCHK(hipMemset(t->d_out[b], 0, L.total));
        }
        if (t->h_out) TCHK(hipHostFree(t->h_out));
        t->h_out = nullptr;
        TCHK(hipHostMalloc((void **)&t->h_out, L.total));
        t->out_bytes = L.total;
        if (t->cap != 0 && cap != t->cap) t->n_prev = -1; /* the previous frame's slab went with the old blocks */
        t->cap = cap;
    }
    const size_t img_need = ((size_t)w * h + 255) & ~(size_t)255;
    if (img_need > t->img_cap) {
        TCHK(hipStreamSynchronize(t->stream));
        TCHK(hipStreamSynchronize(t->pf_stream));
        t->pf_valid = false;
        for (int b = 0; b < 2; b++) {
            if (t->d_img[b]) TCHK(hipFree(t->d_img[b]));
            if (t->h_img[b]) TCHK(hipHostFree(t->h_img[b]));
            t->d_img[b] = t->h_img[b] = nullptr;
        }
        t->img_cap = 0;
        for (int b = 0; b < 2; b++) {
            TCHK(hipMalloc((void **)&t->d_img[b], img_need));
            TCHK(hipHostMalloc((void **)&t->h_img[b], img_need));
        }
        t->img_cap = img_need;
    }
    const size_t np_ = (size_t)std::max(n_prev, 0);
    const size_t in_need = ((np_ + 255) & ~(size_t)255) + ((np_ * 24 + 255) & ~(size_t)255) + 320; /* [flags | xw | pose], each 256-B aligned */
    if (in_need > t->in_cap) {
        TCHK(hipStreamSynchronize(t->stream));
        if (t->d_in) TCHK(hipFree(t->d_in));
        if (t->h_in) TCHK(hipHostFree(t->h_in));
        t->d_in = t->h_in = nullptr;
        t->in_cap = 0;
        const size_t cap_b = in_need + in_need / 4;
        TCHK(hipMalloc((void **)&t->d_in, cap_b));
        TCHK(hipHostMalloc((void **)&t->h_in, cap_b));
        t->in_cap = cap_b;
    }
    return ORB_OK;
}

static void stage_image(uint8_t *dst, const uint8_t *gray, int w, int h, int stride) {
    if (stride == w)
        memcpy(dst, gray, (size_t)w * h);
    else
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, gray + (size_t)y * stride, w);
}

extern "C" int orb_tracker_track_prefetch(OrbTracker *t, const uint8_t *gray, int w, int h, int stride, const uint8_t *prev_flags,
                                          const double *prev_xw, int n_prev, const double *pose_pred7, double fx, double fy, double cx,
                                          double cy, float min_x, float max_x, float min_y, float max_y, float th, int check_orientation,
                                          OrbKeyPoint *kps_out, uint8_t *desc_out, int cap, int32_t *n_out, int32_t *cur_match,
                                          uint8_t *outlier, double *pose_out7, int32_t *n_matches, int32_t *n_inliers,
                                          const uint8_t *next_gray) {
    if (!t || !gray || w <= 0 || h <= 0 || stride < w || !kps_out || !desc_out || !n_out || !pose_pred7 || !pose_out7 || !n_matches ||
        !n_inliers || n_prev < 0 || (n_prev > 0 && (!prev_flags || !prev_xw || !cur_match || !outlier)))
        return ORB_E_INVALID;
    const bool have_prev = t->n_prev >= 0 && n_prev > 0;
    if (n_prev > 0 && n_prev != t->n_prev) {
        t_err = "orb_tracker_track: n_prev is not the feature count of the previous frame of this tracker";
        return ORB_E_INVALID;
    }
    TCHK(hipSetDevice(t->device));
    int rc;
    /* the extractor's geometry (and with it its keypoint bound) belongs to the frame shape: a first frame of a new shape
       goes through the host-pointer call once so that the bound is known before the blocks are sized */
    if (orb_max_keypoints(t->ext) <= 0 || t->cap == 0) {
        std::vector<OrbKeyPoint> k0(std::max(cap, 1));
        std::vector<uint8_t> d0((size_t)std::max(cap, 1) * 32);
        int32_t n0 = 0;
        rc = orb_extract(t->ext, gray, w, h, stride, k0.data(), d0.data(), cap, &n0);
        if (rc != ORB_OK && rc != ORB_E_CAPACITY) return rc;
    }
    /* is this the frame that the previous call fetched ahead? (same buffer, same shape: the caller's promise) */
    bool have_pf = t->pf_valid && t->pf_ptr == gray && t->pf_w == w && t->pf_h == h && t->pf_stride == stride;
    if (t->pf_valid && !have_pf) { /* a different frame came: the look-ahead is dropped once its work has drained */
        TCHK(hipStreamSynchronize(t->pf_stream));
        t->pf_valid = false;
    }
    if ((rc = tracker_buffers(t, w, h, have_prev ? n_prev : 0))) return rc;
    have_pf = have_pf && t->pf_valid; /* (a reallocation drops it) */
    if (cap < t->cap) return ORB_E_CAPACITY;
    const TrackLayout L(t->cap);
    uint8_t *O = t->d_out[t->cur], *P = t->d_out[(t->cur + 2) % 3], *N = t->d_out[(t->cur + 1) % 3]; /* this frame's block, the previous frame's, the next one's */
    if (have_pf) {
        /* ---- uploaded and extracted beside the previous frame's chain, into this block ---- */
        TCHK(hipStreamWaitEvent(t->stream, t->ev_pf, 0));
        t->pf_valid = false;
    } else {
        uint8_t *hi = t->h_img[t->img_i], *di = t->d_img[t->img_i];
        t->img_i ^= 1;
        stage_image(hi, gray, w, h, stride);
        TCHK(hipMemcpyAsync(di, hi, (size_t)w * h, hipMemcpyHostToDevice, t->stream));
        /* ---- extraction: features stay in this frame's block ---- */
        if ((rc = orb_extract_batch_device(t->ext, di, 1, w, h, w, (int64_t)w * h, (OrbKeyPoint *)(O + L.kps), O + L.desc, t->cap,
                                           (int32_t *)(O + L.count), t->stream)))
            return rc;
        TCHK(hipEventRecord(t->ev_ext, t->stream)); /* the extractor's own buffers are free again behind this point */
        TCHK(hipStreamWaitEvent(t->pf_stream, t->ev_ext, 0));
    }
    /* ---- one small upload: [flags | xw | pose] ---- */
    const size_t o_fl = 0, o_xw = ((size_t)(have_prev ? n_prev : 0) + 255) & ~(size_t)255,
                 o_pose = (o_xw + (size_t)(have_prev ? n_prev : 0) * 24 + 255) & ~(size_t)255, in_bytes = o_pose + 64;
    if (have_prev) {
        memcpy(t->h_in + o_fl, prev_flags, n_prev);
        memcpy(t->h_in + o_xw, prev_xw, (size_t)n_prev * 24);
    }
    memcpy(t->h_in + o_pose, pose_pred7, 56);
    TCHK(hipMemcpyAsync(t->d_in, t->h_in, in_bytes, hipMemcpyHostToDevice, t->stream));
    TCHK(hipMemcpyAsync(O + L.pose, t->d_in + o_pose, 56, hipMemcpyDeviceToDevice, t->stream));
    if (have_prev) {
        TCHK(hipMemsetAsync(O + L.taken, 0, t->cap, t->stream)); /* a new frame holds no map points yet (src/Frame.cpp:66); cur_match is
                                                                     written for every feature by the walk itself */
        /* pose (quaternion) -> R, t for the projection */
        const double *q = pose_pred7;
        const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
        const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw),     2 * (qx * qz + qy * qw),
                             2 * (qx * qy + qz * qw),     1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                             2 * (qx * qz - qy * qw),     2 * (qy * qz + qx * qw),     1 - 2 * (qx * qx + qy * qy)};
        if ((rc = orb_search_by_projection_device(t->mat, (const OrbKeyPoint *)(O + L.kps), O + L.desc, t->cap, (const int32_t *)(O + L.count),
                                                  (const OrbKeyPoint *)(P + L.kps), P + L.desc, n_prev, t->d_in + o_fl,
                                                  (const double *)(t->d_in + o_xw), R, q + 4, fx, fy, cx, cy, t->scale.data(), t->nlevels,
                                                  min_x, max_x, min_y, max_y, O + L.taken, th, check_orientation,
                                                  (int32_t *)(O + L.match), (int32_t *)(O + L.nm), t->stream)))
            return rc;
        if ((rc = ba_pose_optimization_matches_device(t->sol, (const OrbKeyPoint *)(O + L.kps), t->cap, (const int32_t *)(O + L.count),
                                                      (const int32_t *)(O + L.match), (const double *)(t->d_in + o_xw),
                                                      (double *)(O + L.pose), t->inv_sigma2.data(), t->nlevels, fx, fy, cx, cy,
                                                      O + L.outlier, (int32_t *)(O + L.ninl), (int32_t *)(O + L.ncorr), t->stream)))
            return rc;
    }
    if (next_gray) {
        /* ---- look-ahead: the next frame goes up and through the extractor on the second stream while the search and
           the pose problem of this one run; its features land in the next block of the ring (whose last readers -- the
           search of the call before this one -- finished before that call returned) ---- */
        uint8_t *hi = t->h_img[t->img_i], *di = t->d_img[t->img_i];
        t->img_i ^= 1;
        stage_image(hi, next_gray, w, h, stride);
        TCHK(hipMemcpyAsync(di, hi, (size_t)w * h, hipMemcpyHostToDevice, t->pf_stream));
        if ((rc = orb_extract_batch_device(t->ext, di, 1, w, h, w, (int64_t)w * h, (OrbKeyPoint *)(N + L.kps), N + L.desc, t->cap,
                                           (int32_t *)(N + L.count), t->pf_stream)))
            return rc;
        TCHK(hipEventRecord(t->ev_pf, t->pf_stream));
        t->pf_valid = true;
        t->pf_ptr = next_gray;
        t->pf_w = w;
        t->pf_h = h;
        t->pf_stride = stride;
    }
    /* ---- one download, one wait ---- */
    TCHK(hipMemcpyAsync(t->h_out, O, L.total, hipMemcpyDeviceToHost, t->stream));
    TCHK(hipStreamSynchronize(t->stream));
    const int n = *(const int32_t *)(t->h_out + L.count);
    if (n < 0 || n > t->cap) {
        t_err = "orb_tracker_track: corrupt feature count";
        return ORB_E_HIP;
    }
    *n_out = n;
    memcpy(kps_out, t->h_out + L.kps, (size_t)n * sizeof(OrbKeyPoint));
    memcpy(desc_out, t->h_out + L.desc, (size_t)n * 32);
    if (have_prev) {
        memcpy(cur_match, t->h_out + L.match, (size_t)n * 4);
        memcpy(outlier, t->h_out + L.outlier, n);
        memcpy(pose_out7, t->h_out + L.pose, 56);
        *n_matches = *(const int32_t *)(t->h_out + L.nm);
        *n_inliers = *(const int32_t *)(t->h_out + L.ninl);
    } else {
        if (cur_match)
            for (int j = 0; j < n; j++) cur_match[j] = -1;
        if (outlier) memset(outlier, 0, n);
        memcpy(pose_out7, pose_pred7, 56);
        *n_matches = 0;
        *n_inliers = 0;
    }
    t->n_prev = n;
    t->cur = (t->cur + 1) % 3;
    return ORB_OK;
}

extern "C" int orb_tracker_track(OrbTracker *t, const uint8_t *gray, int w, int h, int stride, const uint8_t *prev_flags,
                                 const double *prev_xw, int n_prev, const double *pose_pred7, double fx, double fy, double cx,
                                 double cy, float min_x, float max_x, float min_y, float max_y, float th, int check_orientation,
                                 OrbKeyPoint *kps_out, uint8_t *desc_out, int cap, int32_t *n_out, int32_t *cur_match,
                                 uint8_t *outlier, double *pose_out7, int32_t *n_matches, int32_t *n_inliers) {
    return orb_tracker_track_prefetch(t, gray, w, h, stride, prev_flags, prev_xw, n_prev, pose_pred7, fx, fy, cx, cy, min_x, max_x, min_y,
                                      max_y, th, check_orientation, kps_out, desc_out, cap, n_out, cur_match, outlier, pose_out7, n_matches,
                                      n_inliers, nullptr);
}
