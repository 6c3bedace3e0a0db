/*
 * Optimizer.h -- drop-in replacement of the reference's include/Optimizer.h:8-20 on top of the MI355X
 * C-ABI (ba_solve / ba_pose_optimization).  All-static class like the reference; the methods are
 * templates over the KeyFrame / MapPoint / Map / Frame types, which only need the accessors the
 * reference's own Optimizer uses (src/Optimizer.cpp:43-122, :131-261):
 *   KeyFrame: IsBad(), GetR()(i,j), Gettrans()(i), SetR(M), Settrans(v), mlnId, mvUnKeypts, static fx,fy,cx,cy
 *   Frame:    GetR(), GetT(), GetPose()(i,j), SetPose(M), GetUnKeyPts(), mvpMappts, mvbOutlier
 *   MapPoint: IsBad(), Getpos()(i), SetPos(v), GetObservations() -> std::map<KeyFrame*,int>, mlnId
 *   Map:      GetKeyFrames(), GetMapPoints()
 * Graph construction follows the reference line by line; the indeterminacies it leaves (uninitialised
 * invSigma2 / maxKFid, no write-back) are resolved as SURVEY.md 8c lists: information = 1/sigma^2 of the
 * keypoint's octave (1.2^-2*octave unless SetLevelSigma2 is called), maxKFid = max(mlnId), estimates
 * ARE written back with SetR / Settrans / SetPos.
 */
#ifndef ORBSLAM_AMD_OPTIMIZER_H
#define ORBSLAM_AMD_OPTIMIZER_H

#include <cmath>
#include <map>
#include <stdexcept>
#include <vector>

#include "../orbslam_amd.h"
#include "cv_shim.h"

namespace ORBSlam {

class Optimizer {
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

    /* reference src/Optimizer.cpp:23-129 */
    template <class KeyFrameT, class MapPointT>
    static void BundleAdjustment(const std::vector<KeyFrameT *> &vpKF, const std::vector<MapPointT *> &vpMP,
                                 int nIterations = 5, bool *pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                 const bool bRobust = true) {
        (void)nLoopKF;
        int rounds[1] = {nIterations};
        Run(vpKF, vpMP, rounds, 1, pbStopFlag, bRobust, -1);
    }

    /* reference src/Optimizer.cpp:14-21 (sic) */
    template <class MapT>
    static void GlobalBundleAdjustemnt(MapT *pMap, int nIterations = 5, bool *pbStopFlag = NULL,
                                       const unsigned long nLoopKF = 0, const bool bRobust = true) {
        auto vpKeyfs = pMap->GetKeyFrames();
        auto vpMappts = pMap->GetMapPoints();
        BundleAdjustment(vpKeyfs, vpMappts, nIterations, pbStopFlag, nLoopKF, bRobust);
    }

    /* a14 (absent from the reference): window-explicit local BA, upstream ORB-SLAM2 schedule
     * (5 its, switch off chi2 > 5.991 / negative depth edges and the robust kernel, 10 its). */
    template <class KeyFrameT, class MapPointT>
    static void LocalBundleAdjustment(const std::vector<KeyFrameT *> &vpKF, const std::vector<MapPointT *> &vpMP,
                                      bool *pbStopFlag = NULL) {
        int rounds[2] = {5, 10};
        Run(vpKF, vpMP, rounds, 2, pbStopFlag, true, 0);
    }

    /* reference src/Optimizer.cpp:131-261.  FrameT needs GetR(), GetT(), SetPose(M4), GetUnKeyPts(),
     * mvpMappts, mvbOutlier and a static GetCameraPara-equivalent passed here as fx,fy,cx,cy. */
    template <class FrameT>
    static int PoseOptimization(FrameT *pFrame, double fx, double fy, double cx, double cy) {
        auto &vpmappts = pFrame->mvpMappts;
        const auto &vkeyPts = pFrame->GetUnKeyPts();
        const int N = (int)vpmappts.size();
        std::vector<double> xw, obs, info;
        std::vector<int> idx;
        for (int i = 0; i < N; i++)
            if (vpmappts[i]) {
                pFrame->mvbOutlier[i] = false;
                auto p = vpmappts[i]->Getpos();
                xw.push_back(p(0));
                xw.push_back(p(1));
                xw.push_back(p(2));
                obs.push_back(vkeyPts[i].pt.x);
                obs.push_back(vkeyPts[i].pt.y);
                const int o = vkeyPts[i].octave;
                info.push_back(o >= 0 && o < (int)InvLevelSigma2().size() ? InvLevelSigma2()[o] : 1.0);
                idx.push_back(i);
            }
        const int n = (int)idx.size();
        if (n < 3) return 0; /* :206-207 */
        double pose[7];
        RtToPose(pFrame->GetR(), pFrame->GetT(), pose);
        std::vector<uint8_t> outlier(n);
        int32_t ninl = 0;
        if (ba_pose_optimization(Solver(), pose, xw.data(), obs.data(), info.data(), n, fx, fy, cx, cy, outlier.data(),
                                 &ninl) != ORB_OK)
            throw std::runtime_error("Optimizer::PoseOptimization failed");
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

  private:
    static BASolver *Solver() {
        static BASolver *s = nullptr;
        if (!s && ba_create(256, 1 << 20, 1 << 22, 1, 0, &s) != ORB_OK) throw std::runtime_error("Optimizer: no HIP device");
        return s;
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
    static void Run(const std::vector<KeyFrameT *> &vpKF, const std::vector<MapPointT *> &vpMP, const int *rounds,
                    int n_rounds, bool *pbStopFlag, bool bRobust, int drop_robust_after) {
        /* pose vertices (:43-58) */
        std::map<KeyFrameT *, int> camOf;
        std::vector<KeyFrameT *> cams;
        std::vector<double> poses;
        std::vector<uint8_t> fixed;
        unsigned long maxKFid = 0; /* uninitialised in the reference (:28) */
        for (size_t i = 0; i < vpKF.size(); i++) {
            if (vpKF[i]->IsBad()) continue;
            double p[7];
            RtToPose(vpKF[i]->GetR(), vpKF[i]->Gettrans(), p);
            camOf[vpKF[i]] = (int)cams.size();
            cams.push_back(vpKF[i]);
            poses.insert(poses.end(), p, p + 7);
            fixed.push_back(vpKF[i]->mlnId == 0 ? 1 : 0); /* :52-53 */
            if (vpKF[i]->mlnId > maxKFid) maxKFid = vpKF[i]->mlnId;
        }
        if (cams.empty()) return;
        /* point vertices and edges (:64-122) */
        std::vector<MapPointT *> pts;
        std::vector<double> points, obs, info;
        std::vector<int32_t> cam_idx, pt_idx;
        for (size_t i = 0; i < vpMP.size(); i++) {
            MapPointT *mp = vpMP[i];
            if (mp->IsBad()) continue;
            int nedge = 0;
            const int pid = (int)pts.size();
            auto &observations = mp->GetObservations();
            for (auto it = observations.begin(); it != observations.end(); ++it) {
                KeyFrameT *kf = it->first;
                if (kf->IsBad() || kf->mlnId > maxKFid) continue;
                auto f = camOf.find(kf);
                if (f == camOf.end()) continue; /* vertex not in the graph */
                nedge++;
                const cv::KeyPoint &kp = kf->mvUnKeypts[it->second];
                obs.push_back(kp.pt.x);
                obs.push_back(kp.pt.y);
                info.push_back(kp.octave >= 0 && kp.octave < (int)InvLevelSigma2().size() ? InvLevelSigma2()[kp.octave] : 1.0);
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
        int rc = ba_solve(Solver(), &P, &Rr);
        if (rc == ORB_E_STOPPED) return;
        if (rc != ORB_OK) throw std::runtime_error("Optimizer::BundleAdjustment failed");
        /* write-back (the reference's TODO at :127; upstream semantics) */
        for (size_t c = 0; c < cams.size(); c++) {
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

} // namespace ORBSlam
#endif
