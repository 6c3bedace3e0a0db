/*
 * Optimizer.h -- drop-in replacement of the reference's include/Optimizer.h:8-20 on top of the MI355X
 * C-ABI (ba_solve / ba_pose_optimization).  Like the reference header (include/Optimizer.h:4) it includes
 * "KeyFrame.h" and defines the all-static `class ORBSlam::Optimizer` with the reference's exact signatures
 *     void static BundleAdjustment(const std::vector<KeyFrame*>&, const std::vector<MapPoint*>&, int, bool*, unsigned long, bool);
 *     void static GlobalBundleAdjustemnt(Map*, int, bool*, unsigned long, bool);        // sic, src/Tracking.cpp:254
 *     int  static PoseOptimization(Frame*);                                             // src/Tracking.cpp:309,355,382
 * plus the one north_star adds in the reference's style:
 *     void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap);
 * The bodies are the templates of OptimizerT, which only need the accessors the reference's own Optimizer uses
 * (src/Optimizer.cpp:43-122, :131-261):
 *   KeyFrame: IsBad(), GetR()(i,j), Gettrans()(i), SetR(M), Settrans(v), GetMapPoints(), mlnId, mvUnKeypts,
 *             static fx,fy,cx,cy
 *   Frame:    GetR(), GetT(), GetPose()(i,j), SetPose(M), GetUnKeyPts(), static GetCameraPara(cv::Mat&), mvpMappts,
 *             mvbOutlier
 *   MapPoint: IsBad(), Getpos()(i), SetPos(v), GetObservations() -> std::map<KeyFrame*,int>, mlnId
 *   Map:      GetKeyFrames(), GetMapPoints()
 * Graph construction follows the reference line by line; the indeterminacies it leaves (uninitialised
 * invSigma2 / maxKFid, no write-back, cx/cy read from the wrong cells of the camera matrix) are resolved as SURVEY.md
 * 8c lists: information = 1/sigma^2 of the keypoint's octave (1.2^-2*octave unless SetLevelSigma2 is called),
 * maxKFid = max(mlnId), estimates ARE written back with SetR / Settrans / SetPos, cx = K(0,2), cy = K(1,2).
 *
 * Threading: the reference's static methods build a fresh g2o optimizer per call and upstream runs BA on the
 * LocalMapping thread while Tracking optimises poses.  Here every host thread gets its OWN device solver
 * (thread_local handle: stream, staging, device buffers), created on first use and destroyed at thread exit.
 * Errors: none on the reference's paths (no exceptions, no codes); a dead device is fatal (orbslam_amd_fatal).
 */
#ifndef ORBSLAM_AMD_OPTIMIZER_H
#define ORBSLAM_AMD_OPTIMIZER_H

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <set>
#include <type_traits>
#include <vector>

#include "../orbslam_amd.h"
#include "ORBextractor.h" /* orbslam_amd_fatal */
#include "cv_shim.h"

#if !defined(ORBSLAM_AMD_NO_FRAME_INCLUDE) && defined(__has_include)
#if __has_include("KeyFrame.h")
#include "KeyFrame.h" /* reference include/Optimizer.h:4 (brings Frame, MapPoint, Map) */
#define ORBSLAM_AMD_HAVE_KEYFRAME_H 1
#endif
#endif

namespace ORBSlam {

class OptimizerT {
  public:
    /* per-octave inverse variances (ORBextractor::GetInverseScaleSigmaSquares()); default 1.2^(-2 l) */
    static std::vector<float> &InvLevelSigma2() {
        static std::vector<float> v = [] {
            std::vector<float> t(8);
            float s = 1.f;
            for (int i = 0; i < 8; i++) {
                t[i] = 1.f / (s * s);
                s *= 1.2f;
            }
            return t;
        }();
        return v;
    }
    static void SetLevelSigma2(const std::vector<float> &inv_sigma2) { InvLevelSigma2() = inv_sigma2; }

    /* reference src/Optimizer.cpp:23-129 */
    template <class KeyFrameT, class MapPointT>
    static void BundleAdjustment(const std::vector<KeyFrameT *> &vpKF, const std::vector<MapPointT *> &vpMP,
                                 int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                 const bool bRobust = true) {
        (void)nLoopKF;
        int rounds[1] = {nIterations};
        Run(vpKF, std::vector<KeyFrameT *>(), vpMP, rounds, 1, pbStopFlag, bRobust, -1);
    }

    /* reference src/Optimizer.cpp:14-21 (sic) */
    template <class MapT>
    static void GlobalBundleAdjustemnt(MapT *pMap, int nIterations = 5, bool *pbStopFlag = NULL,
                                       const unsigned long nLoopKF = 0, const bool bRobust = true) {
        auto vpKeyfs = pMap->GetKeyFrames();
        auto vpMappts = pMap->GetMapPoints();
        BundleAdjustment(vpKeyfs, vpMappts, nIterations, pbStopFlag, nLoopKF, bRobust);
    }

    /* a14, window-explicit form: vpKF are optimised (the one with mlnId == 0 stays fixed, as everywhere in the
     * reference), vpFixedKF only constrain (upstream's lFixedCameras), vpMP are optimised.  Upstream ORB-SLAM2
     * schedule: 5 its, switch off chi2 > 5.991 / non-positive-depth edges and the robust kernel, 10 its. */
    template <class KeyFrameT, class MapPointT>
    static void LocalBundleAdjustment(const std::vector<KeyFrameT *> &vpKF, const std::vector<KeyFrameT *> &vpFixedKF,
                                      const std::vector<MapPointT *> &vpMP, bool *pbStopFlag = NULL) {
        int rounds[2] = {5, 10};
        Run(vpKF, vpFixedKF, vpMP, rounds, 2, pbStopFlag, true, 0);
    }
    template <class KeyFrameT, class MapPointT>
    static void LocalBundleAdjustment(const std::vector<KeyFrameT *> &vpKF, const std::vector<MapPointT *> &vpMP,
                                      bool *pbStopFlag = NULL) {
        LocalBundleAdjustment(vpKF, std::vector<KeyFrameT *>(), vpMP, pbStopFlag);
    }

    /* the covisibility window of a keyframe, from the accessors the reference exposes
     * (`mvpOrderedConnectedKeyFrames` is private, include/KeyFrame.h:86-88, and has no getter).  Same rule as
     * KeyFrame::UpdateConnections (src/KeyFrame.cpp:54-113): a keyframe is connected to pKF when they share MORE
     * than 15 map points, or -- if none does -- the one sharing the most.  Local keyframes = pKF + connected ones;
     * local points = the good points any of them holds; fixed keyframes = every other observer of a local point
     * (upstream LocalBundleAdjustment).  Orders are deterministic: keyframes by mlnId, points by mlnId. */
    template <class KeyFrameT, class MapPointT>
    static void LocalWindow(KeyFrameT *pKF, std::vector<KeyFrameT *> &vpLocalKF, std::vector<KeyFrameT *> &vpFixedKF,
                            std::vector<MapPointT *> &vpLocalMP) {
        vpLocalKF.clear();
        vpFixedKF.clear();
        vpLocalMP.clear();
        if (!pKF || pKF->IsBad()) return;
        std::map<KeyFrameT *, int> counts; /* src/KeyFrame.cpp:57-74 */
        {
            const auto mps = pKF->GetMapPoints();
            for (size_t i = 0; i < mps.size(); i++) {
                MapPointT *mp = mps[i];
                if (!mp || mp->IsBad()) continue;
                const auto &obs = mp->GetObservations();
                for (auto it = obs.begin(); it != obs.end(); ++it)
                    if (it->first->mlnId != pKF->mlnId) counts[it->first]++;
            }
        }
        const int th = 15; /* src/KeyFrame.cpp:82 */
        KeyFrameT *best = nullptr;
        int nbest = -1;
        std::set<KeyFrameT *> local;
        local.insert(pKF);
        for (auto it = counts.begin(); it != counts.end(); ++it) {
            if (it->first->IsBad()) continue;
            if (it->second > nbest || (it->second == nbest && best && it->first->mlnId < best->mlnId)) {
                nbest = it->second;
                best = it->first;
            }
            if (it->second > th) local.insert(it->first);
        }
        if (local.size() == 1 && best) local.insert(best); /* :99-103 */
        auto by_id = [](const KeyFrameT *a, const KeyFrameT *b) { return a->mlnId < b->mlnId; };
        vpLocalKF.assign(local.begin(), local.end());
        std::sort(vpLocalKF.begin(), vpLocalKF.end(), by_id);
        std::set<MapPointT *> pts;
        for (KeyFrameT *kf : vpLocalKF) {
            const auto mps = kf->GetMapPoints();
            for (size_t i = 0; i < mps.size(); i++)
                if (mps[i] && !mps[i]->IsBad()) pts.insert(mps[i]);
        }
        vpLocalMP.assign(pts.begin(), pts.end());
        std::sort(vpLocalMP.begin(), vpLocalMP.end(), [](const MapPointT *a, const MapPointT *b) { return a->mlnId < b->mlnId; });
        std::set<KeyFrameT *> fixed;
        for (MapPointT *mp : vpLocalMP) {
            const auto &obs = mp->GetObservations();
            for (auto it = obs.begin(); it != obs.end(); ++it)
                if (!local.count(it->first) && !it->first->IsBad()) fixed.insert(it->first);
        }
        vpFixedKF.assign(fixed.begin(), fixed.end());
        std::sort(vpFixedKF.begin(), vpFixedKF.end(), by_id);
    }

    /* a14 / N1 in the reference's style: the call LocalMapping::Run would make after ProcessNewKeyFrame
     * (src/LocalMapping.cpp:11-20).  pMap is not read (upstream takes its mutex there; the reference has none). */
    template <class KeyFrameT, class MapT> static void LocalBundleAdjustment(KeyFrameT *pKF, bool *pbStopFlag, MapT *pMap) {
        (void)pMap;
        typedef typename std::remove_pointer<typename decltype(pKF->GetMapPoints())::value_type>::type MapPointT;
        std::vector<KeyFrameT *> vpLocalKF, vpFixedKF;
        std::vector<MapPointT *> vpLocalMP;
        LocalWindow(pKF, vpLocalKF, vpFixedKF, vpLocalMP);
        if (vpLocalKF.empty() || vpLocalMP.empty()) return;
        LocalBundleAdjustment(vpLocalKF, vpFixedKF, vpLocalMP, pbStopFlag);
    }

    /* reference src/Optimizer.cpp:131-261, intrinsics given explicitly */
    template <class FrameT> static int PoseOptimization(FrameT *pFrame, double fx, double fy, double cx, double cy) {
        auto &vpmappts = pFrame->mvpMappts;
        const auto &vkeyPts = pFrame->GetUnKeyPts();
        const int N = (int)vpmappts.size();
        std::vector<double> xw, obs, info;
        std::vector<int> idx;
        for (int i = 0; i < N; i++)
            if (vpmappts[i]) {
                pFrame->mvbOutlier[i] = false; /* :176 */
                auto p = vpmappts[i]->Getpos();
                xw.push_back(p(0));
                xw.push_back(p(1));
                xw.push_back(p(2));
                obs.push_back(vkeyPts[i].pt.x);
                obs.push_back(vkeyPts[i].pt.y);
                info.push_back(Info(vkeyPts[i].octave));
                idx.push_back(i);
            }
        const int n = (int)idx.size();
        if (n < 3) return 0; /* :206-207 */
        double pose[7];
        RtToPose(pFrame->GetR(), pFrame->GetT(), pose);
        std::vector<uint8_t> outlier(n);
        int32_t ninl = 0;
        if (ba_pose_optimization(Solver(), pose, xw.data(), obs.data(), info.data(), n, fx, fy, cx, cy, outlier.data(),
                                 &ninl) != ORB_OK) {
            std::fprintf(stderr, "orbslam_amd: Optimizer::PoseOptimization: %s\n", orb_last_error());
            return 0;
        }
        for (int k = 0; k < n; k++) pFrame->mvbOutlier[idx[k]] = outlier[k] != 0;
        double R[9];
        PoseToR(pose, R);
        auto T = pFrame->GetPose(); /* Eigen::Matrix4d in the reference; filled through operator()(i,j) */
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) T(i, j) = R[i * 3 + j];
            T(i, 3) = pose[4 + i];
            T(3, i) = 0.0;
        }
        T(3, 3) = 1.0;
        pFrame->SetPose(T); /* :249-258 */
        return ninl;
    }

    /* reference signature, src/Optimizer.cpp:131-139: intrinsics from Frame::GetCameraPara -- fx = K(0,0),
     * fy = K(1,1), cx = K(0,2), cy = K(1,2) (the reference reads (0,1) and (0,2) for cx, cy: SURVEY 8c (v)) */
    template <class FrameT> static int PoseOptimization(FrameT *pFrame) {
        cv::Mat camk;
        pFrame->GetCameraPara(camk);
        return PoseOptimization(pFrame, (double)camk.at<float>(0, 0), (double)camk.at<float>(1, 1), (double)camk.at<float>(0, 2),
                                (double)camk.at<float>(1, 2));
    }

  protected:
    static double Info(int octave) {
        const std::vector<float> &t = InvLevelSigma2();
        return octave >= 0 && octave < (int)t.size() ? (double)t[octave] : 1.0;
    }
    /* one device solver per host thread */
    static BASolver *Solver() {
        struct Holder {
            BASolver *s = nullptr;
            ~Holder() { ba_destroy(s); }
        };
        static thread_local Holder h;
        if (!h.s && ba_create(256, 1 << 20, 1 << 22, 1, 0, &h.s) != ORB_OK) orbslam_amd_fatal("Optimizer");
        return h.s;
    }
    template <class M, class V> static void RtToPose(const M &R, const V &t, double pose[7]) {
        double r[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) r[i * 3 + j] = R(i, j);
        double tr = r[0] + r[4] + r[8], q[4];
        if (tr > 0) {
            double s = std::sqrt(tr + 1.0);
            q[0] = 0.5 * s;
            s = 0.5 / s;
            q[1] = (r[7] - r[5]) * s;
            q[2] = (r[2] - r[6]) * s;
            q[3] = (r[3] - r[1]) * s;
        } else {
            int i = 0;
            if (r[4] > r[0]) i = 1;
            if (r[8] > r[i * 4]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            double s = std::sqrt(r[i * 4] - r[j * 4] - r[k * 4] + 1.0), v[3];
            v[i] = 0.5 * s;
            s = 0.5 / s;
            q[0] = (r[k * 3 + j] - r[j * 3 + k]) * s;
            v[j] = (r[j * 3 + i] + r[i * 3 + j]) * s;
            v[k] = (r[k * 3 + i] + r[i * 3 + k]) * s;
            q[1] = v[0];
            q[2] = v[1];
            q[3] = v[2];
        }
        for (int i = 0; i < 4; i++) pose[i] = q[i];
        for (int i = 0; i < 3; i++) pose[4 + i] = t(i);
    }
    static void PoseToR(const double *q, double R[9]) {
        const double w = q[0], x = q[1], y = q[2], z = q[3];
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
        R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
        R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
    }

    template <class KeyFrameT, class MapPointT>
    static void Run(const std::vector<KeyFrameT *> &vpKF, const std::vector<KeyFrameT *> &vpFixedKF,
                    const std::vector<MapPointT *> &vpMP, const int *rounds, int n_rounds, bool *pbStopFlag, bool bRobust,
                    int drop_robust_after) {
        /* pose vertices (:43-58); the fixed observers of a local window follow the free ones */
        std::map<KeyFrameT *, int> camOf;
        std::vector<KeyFrameT *> cams;
        std::vector<double> poses;
        std::vector<uint8_t> fixed;
        unsigned long maxKFid = 0; /* uninitialised in the reference (:28) */
        size_t nfree = 0;
        for (int pass = 0; pass < 2; pass++) {
            const std::vector<KeyFrameT *> &v = pass == 0 ? vpKF : vpFixedKF;
            for (size_t i = 0; i < v.size(); i++) {
                if (!v[i] || v[i]->IsBad() || camOf.count(v[i])) continue;
                double p[7];
                RtToPose(v[i]->GetR(), v[i]->Gettrans(), p);
                camOf[v[i]] = (int)cams.size();
                cams.push_back(v[i]);
                poses.insert(poses.end(), p, p + 7);
                fixed.push_back((pass == 1 || v[i]->mlnId == 0) ? 1 : 0); /* :52-53 */
                if (v[i]->mlnId > maxKFid) maxKFid = v[i]->mlnId;
            }
            if (pass == 0) nfree = cams.size();
        }
        if (cams.empty()) return;
        /* point vertices and edges (:64-122) */
        std::vector<MapPointT *> pts;
        std::vector<double> points, obs, info;
        std::vector<int32_t> cam_idx, pt_idx;
        for (size_t i = 0; i < vpMP.size(); i++) {
            MapPointT *mp = vpMP[i];
            if (!mp || mp->IsBad()) continue;
            int nedge = 0;
            const int pid = (int)pts.size();
            const auto &observations = mp->GetObservations();
            for (auto it = observations.begin(); it != observations.end(); ++it) {
                KeyFrameT *kf = it->first;
                if (kf->IsBad() || kf->mlnId > maxKFid) continue; /* :84-85 */
                auto f = camOf.find(kf);
                if (f == camOf.end()) continue; /* vertex not in the graph */
                nedge++;
                const cv::KeyPoint &kp = kf->mvUnKeypts[it->second];
                obs.push_back(kp.pt.x);
                obs.push_back(kp.pt.y);
                info.push_back(Info(kp.octave));
                cam_idx.push_back(f->second);
                pt_idx.push_back(pid);
            }
            if (nedge == 0) continue; /* vertex removed (:115-119) */
            auto p = mp->Getpos();
            points.push_back(p(0));
            points.push_back(p(1));
            points.push_back(p(2));
            pts.push_back(mp);
        }
        BAProblem P;
        P.n_cams = (int)cams.size();
        P.n_points = (int)pts.size();
        P.n_obs = (int)cam_idx.size();
        P.poses = poses.data();
        P.fixed = fixed.data();
        P.points = points.data();
        P.obs = obs.data();
        P.cam_idx = cam_idx.data();
        P.pt_idx = pt_idx.data();
        P.inv_sigma2 = info.data();
        P.fx = KeyFrameT::fx; /* :107-110 */
        P.fy = KeyFrameT::fy;
        P.cx = KeyFrameT::cx;
        P.cy = KeyFrameT::cy;
        P.huber_delta = bRobust ? std::sqrt(5.99) : 0.0; /* :61 */
        P.n_rounds = n_rounds;
        for (int i = 0; i < 4; i++) P.rounds[i] = i < n_rounds ? rounds[i] : 0;
        P.chi2_threshold = 5.991;
        P.drop_robust_after = drop_robust_after;
        P.stop = (volatile const uint8_t *)pbStopFlag; /* :39-40 */
        std::vector<double> out_poses(poses.size()), out_points(points.size());
        BAResult Rr;
        Rr.poses = out_poses.data();
        Rr.points = out_points.data();
        Rr.outlier = nullptr;
        const int rc = ba_solve(Solver(), &P, &Rr);
        if (rc == ORB_E_STOPPED) return; /* stop flag already set: g2o's optimize() returns without iterating */
        if (rc != ORB_OK) {              /* estimates stay as they were */
            std::fprintf(stderr, "orbslam_amd: Optimizer::BundleAdjustment: %s\n", orb_last_error());
            return;
        }
        /* write-back (the reference's TODO at :127; upstream semantics: free vertices only) */
        for (size_t c = 0; c < nfree; c++) {
            if (fixed[c]) continue;
            double R[9];
            PoseToR(&out_poses[c * 7], R);
            auto Rm = cams[c]->GetR(); /* include/KeyFrame.h:34-52 */
            auto tv = cams[c]->Gettrans();
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) Rm(i, j) = R[i * 3 + j];
                tv(i) = out_poses[c * 7 + 4 + i];
            }
            cams[c]->SetR(Rm);
            cams[c]->Settrans(tv);
        }
        for (size_t p = 0; p < pts.size(); p++) {
            auto pv = pts[p]->Getpos(); /* include/MapPoint.h:27-37 */
            for (int i = 0; i < 3; i++) pv(i) = out_points[p * 3 + i];
            pts[p]->SetPos(pv);
        }
    }
};

#ifdef ORBSLAM_AMD_HAVE_KEYFRAME_H
/* the reference's class by its own name with the reference's non-template signatures (include/Optimizer.h:8-20);
 * the templates of OptimizerT stay reachable (window-explicit LocalBundleAdjustment, explicit intrinsics) */
class Optimizer : public OptimizerT {
  public:
    using OptimizerT::BundleAdjustment;
    using OptimizerT::GlobalBundleAdjustemnt;
    using OptimizerT::LocalBundleAdjustment;
    using OptimizerT::PoseOptimization;
    void static BundleAdjustment(const std::vector<KeyFrame *> &vpKF, const std::vector<MapPoint *> &vpMP, int nIterations = 5,
                                 bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true) {
        OptimizerT::BundleAdjustment<KeyFrame, MapPoint>(vpKF, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
    }
    void static GlobalBundleAdjustemnt(Map *pMap, int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                       const bool bRobust = true) {
        OptimizerT::GlobalBundleAdjustemnt<Map>(pMap, nIterations, pbStopFlag, nLoopKF, bRobust);
    }
    int static PoseOptimization(Frame *pFrame) { return OptimizerT::PoseOptimization<Frame>(pFrame); }
    void static LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap) {
        OptimizerT::LocalBundleAdjustment<KeyFrame, Map>(pKF, pbStopFlag, pMap);
    }
};
#else
typedef OptimizerT Optimizer;
#endif

} // namespace ORBSlam
#endif
