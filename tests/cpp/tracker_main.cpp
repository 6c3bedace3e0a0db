/*
 * tracker_main.cpp -- the per-frame chain of Tracking::TrackWithMotionModel (reference src/Tracking.cpp:330-372) through
 * the ONE-round-trip entry point of the C-ABI (orb_tracker_track, include/orbslam_amd.h): image up, extraction,
 * SearchByProjection (walk + rotation histogram on the device), PoseOptimization of the matches, results down, one
 * synchronisation.  The counterpart of tracking_main.cpp's "tracking alone" figure, which makes the same three steps as
 * three calls on the drop-in classes; no local BA here.  Same synthetic plane sequence, same stand-ins for what the
 * out-of-scope callers compute (a keypoint becomes a map point by intersecting its ray with the plane Z = 0; frame 0's
 * pose is given; constant-velocity prediction).  Plain C++ on the C-ABI: no cv:: types, no HIP in this file.
 *
 * usage: tracker_main <w> <h> <n> <frames.raw> <gt_poses.bin> <fx> <fy> <cx> <cy> <out.bin>
 * out.bin: int32 n, then per frame 7 doubles (pose), int32 matches, int32 inliers; then double loop_seconds
 */
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbslam_amd.h"

static void quat_R(const double *q, double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static void R_quat(const double R[9], double *q) {
    const double t = R[0] + R[4] + R[8];
    if (t > 0) {
        const double s = std::sqrt(t + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        const double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0) * 2;
        q[1 + i] = 0.25 * s; q[0] = (R[k * 3 + j] - R[j * 3 + k]) / s; q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) / s;
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) / s;
    }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int a = 0; a < 4; a++) q[a] /= n;
    if (q[0] < 0)
        for (int a = 0; a < 4; a++) q[a] = -q[a];
}
static void mul33(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
/* the world point (Z = 0) seen at pixel (u, v) from the pose (R, t) */
static void plane_point(const double R[9], const double t[3], double u, double v, double fx, double fy, double cx, double cy, double X[3]) {
    double C[3], r[3];
    const double rc[3] = {(u - cx) / fx, (v - cy) / fy, 1.0};
    for (int i = 0; i < 3; i++) {
        C[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
        r[i] = R[i] * rc[0] + R[3 + i] * rc[1] + R[6 + i] * rc[2];
    }
    const double lam = -C[2] / r[2];
    for (int i = 0; i < 3; i++) X[i] = C[i] + lam * r[i];
}

int main(int argc, char **argv) {
    if (argc < 11) return 2;
    const int w = atoi(argv[1]), h = atoi(argv[2]), n = atoi(argv[3]);
    const double fx = atof(argv[6]), fy = atof(argv[7]), cx = atof(argv[8]), cy = atof(argv[9]);
    std::vector<unsigned char> raw((size_t)n * w * h);
    std::vector<double> gt((size_t)n * 7);
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) return 3;
    fclose(f);
    f = fopen(argv[5], "rb");
    if (!f || fread(gt.data(), 8, gt.size(), f) != gt.size()) return 3;
    fclose(f);
    OrbConfig cfg = {1000, 1.2f, 8, 20, 7};
    OrbExtractor *ext = nullptr;
    OrbMatcher *mat = nullptr;
    BASolver *sol = nullptr;
    OrbTracker *trk = nullptr;
    if (orb_create(&cfg, w, h, 1, 0, &ext) != ORB_OK) return 4;
    /* (the extractor's keypoint bound belongs to the frame shape: one host-pointer call sets the geometry) */
    int cap = orb_max_keypoints(ext);
    std::vector<OrbKeyPoint> kps(cap > 0 ? cap : 4096), kps_last;
    std::vector<uint8_t> desc((size_t)kps.size() * 32);
    int32_t n0 = 0;
    if (orb_extract(ext, raw.data(), w, h, w, kps.data(), desc.data(), (int)kps.size(), &n0) != ORB_OK) return 4;
    cap = orb_max_keypoints(ext);
    kps.resize(cap);
    desc.resize((size_t)cap * 32);
    if (orb_matcher_create(cap, 1, 0, &mat) != ORB_OK) return 4;
    if (ba_create(8, 64, 64, 1, 0, &sol) != ORB_OK) return 4;
    if (orb_tracker_create(ext, mat, sol, 0, &trk) != ORB_OK) return 4;
    const int kf_every = 4;
    const float th = 15.f;
    const bool prefetch = argc > 11 && std::strcmp(argv[11], "prefetch") == 0;
    std::vector<int32_t> match(cap), out_nm(n, 0), out_inl(n, 0);
    std::vector<uint8_t> outl(cap), flags, flags_cur;
    std::vector<double> xw, xw_cur, out_pose((size_t)n * 7);
    int n_last = 0;
    double pose_last[7], Rv[9], tv[3] = {0, 0, 0};
    bool have_velocity = false;
    double t_track = 0.0;
    const auto t_begin = std::chrono::steady_clock::now();
    for (int k = 0; k < n; k++) {
        double pred[7];
        if (k == 0) {
            memcpy(pred, &gt[0], 56);
        } else {
            double Rl[9], Rp[9];
            quat_R(pose_last, Rl);
            if (have_velocity) {
                mul33(Rv, Rl, Rp);
                for (int i = 0; i < 3; i++) pred[4 + i] = Rv[i * 3] * pose_last[4] + Rv[i * 3 + 1] * pose_last[5] + Rv[i * 3 + 2] * pose_last[6] + tv[i];
            } else {
                memcpy(Rp, Rl, sizeof Rp);
                for (int i = 0; i < 3; i++) pred[4 + i] = pose_last[4 + i];
            }
            R_quat(Rp, pred);
        }
        int32_t nk = 0, nm = 0, ninl = 0;
        double pose[7];
        const auto tq = std::chrono::steady_clock::now();
        /* look-ahead (argv[11] = "prefetch"): the whole sequence is in memory, so frame k + 1 is handed over with frame k */
        const unsigned char *next = (prefetch && k + 1 < n) ? raw.data() + (size_t)(k + 1) * w * h : nullptr;
        const int rc = orb_tracker_track_prefetch(trk, raw.data() + (size_t)k * w * h, w, h, w, k ? flags.data() : nullptr,
                                                  k ? xw.data() : nullptr, k ? n_last : 0, pred, fx, fy, cx, cy, 0.f, (float)w, 0.f, (float)h, th,
                                                  1, kps.data(), desc.data(), cap, &nk, match.data(), outl.data(), pose, &nm, &ninl, next);
        if (k) t_track += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq).count();
        if (rc != ORB_OK) {
            std::fprintf(stderr, "orb_tracker_track: %d %s\n", rc, orb_last_error());
            return 5;
        }
        if (k == 0) memcpy(pose, &gt[0], 56);
        out_nm[k] = nm;
        out_inl[k] = ninl;
        memcpy(&out_pose[(size_t)k * 7], pose, 56);
        /* the frame's map-point association: matched inliers inherit the point; keyframes give every free keypoint one */
        flags_cur.assign(nk, 0);
        xw_cur.assign((size_t)nk * 3, 0.0);
        double R[9];
        quat_R(pose, R);
        for (int j = 0; j < nk; j++) {
            if (k > 0 && match[j] >= 0 && !outl[j]) { /* src/Tracking.cpp:357-370: outlier associations are dropped */
                flags_cur[j] = 3;
                memcpy(&xw_cur[(size_t)j * 3], &xw[(size_t)match[j] * 3], 24);
            } else if (k % kf_every == 0) { /* LocalMapping::CreateNewMapPoints stand-in: the plane */
                flags_cur[j] = 3;
                plane_point(R, pose + 4, kps[j].x, kps[j].y, fx, fy, cx, cy, &xw_cur[(size_t)j * 3]);
            }
        }
        if (k > 0) {
            double Rl[9];
            quat_R(pose_last, Rl);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) Rv[i * 3 + j] = R[i * 3] * Rl[j * 3] + R[i * 3 + 1] * Rl[j * 3 + 1] + R[i * 3 + 2] * Rl[j * 3 + 2];
            for (int i = 0; i < 3; i++) tv[i] = pose[4 + i] - (Rv[i * 3] * pose_last[4] + Rv[i * 3 + 1] * pose_last[5] + Rv[i * 3 + 2] * pose_last[6]);
            have_velocity = true;
        }
        memcpy(pose_last, pose, 56);
        flags.swap(flags_cur);
        xw.swap(xw_cur);
        n_last = nk;
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    f = fopen(argv[10], "wb");
    if (!f) return 6;
    fwrite(&n, 4, 1, f);
    for (int k = 0; k < n; k++) {
        fwrite(&out_pose[(size_t)k * 7], 8, 7, f);
        fwrite(&out_nm[k], 4, 1, f);
        fwrite(&out_inl[k], 4, 1, f);
    }
    fwrite(&secs, 8, 1, f);
    fclose(f);
    orb_tracker_destroy(trk);
    ba_destroy(sol);
    orb_matcher_destroy(mat);
    orb_destroy(ext);
    std::printf("tracker_main: %d frames, %.3f ms per tracked frame through orb_tracker_track = %.1f frames/s (whole loop %.3f s)\n", n,
                t_track / (n - 1) * 1e3, (n - 1) / t_track, secs);
    std::printf("{\"frames\": %d, \"secs\": %.6f, \"track_secs\": %.6f}\n", n, secs, t_track);
    return 0;
}
