/*
 * refmodel.h -- TEST stand-ins for the reference's data-model classes (the unchanged CALLERS of the hot path,
 * out of scope per SURVEY 8): just the accessor surface that ORBextractor / ORBmatcher / Optimizer touch, with the
 * reference's names and signatures (include/Frame.h:17-105, include/KeyFrame.h:14-92, include/MapPoint.h:17-77,
 * include/Map.h:15-33), and a 3-type stand-in for the Eigen fixed-size matrices they return.  No reference code:
 * the bodies are trivial storage.  Lets tests/cpp/dropin_main.cpp call the drop-in headers exactly as
 * src/Frame.cpp / src/Tracking.cpp do, by the reference's own class names.
 */
#ifndef ORBSLAM_AMD_TEST_REFMODEL_H
#define ORBSLAM_AMD_TEST_REFMODEL_H

#include <map>
#include <set>
#include <vector>

#include "orbslam_amd/ORBextractor.h"

namespace Eigen {
struct Matrix3d {
    double v[9];
    Matrix3d() : v{1, 0, 0, 0, 1, 0, 0, 0, 1} {}
    double &operator()(int i, int j) { return v[i * 3 + j]; }
    double operator()(int i, int j) const { return v[i * 3 + j]; }
};
struct Matrix4d {
    double v[16];
    Matrix4d() : v{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1} {}
    double &operator()(int i, int j) { return v[i * 4 + j]; }
    double operator()(int i, int j) const { return v[i * 4 + j]; }
};
struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double a, double b, double c) : v{a, b, c} {}
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
} // namespace Eigen

namespace ORBSlam {
class MapPoint;
class KeyFrame;
class Map;

class Frame {
  public:
    Frame() : mpReferenceKF(nullptr), mpOrbextractor(nullptr) {}
    /* src/Frame.cpp:40-78 without the undistortion (zero distortion: mvUnKeypts = mvKeypts, :82-86) */
    Frame(const cv::Mat &img, ORBextractor *porbextractor) : mpReferenceKF(nullptr), mpOrbextractor(porbextractor) {
        (*mpOrbextractor)(img, cv::Mat(), mvKeypts, mcvDescriptors); /* src/Frame.cpp:77 */
        mvUnKeypts = mvKeypts;
        mvpMappts.assign(mvKeypts.size(), nullptr);
        mvbOutlier.assign(mvKeypts.size(), false);
    }
    std::vector<cv::KeyPoint> GetUnKeyPts() const { return mvUnKeypts; }
    std::vector<cv::KeyPoint> &GetUnKeyPts() { return mvUnKeypts; }
    cv::Mat GetDescriptors() const { return mcvDescriptors.clone(); }
    Eigen::Matrix3d GetR() const { return mRwc; }
    Eigen::Matrix4d GetPose() const { return mTwc; }
    Eigen::Vector3d GetT() const { return mtwc; }
    void SetPose(const Eigen::Matrix4d &pose) { /* src/Frame.cpp:212-217 */
        mTwc = pose;
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) mRwc(i, j) = pose(i, j);
            mtwc(i) = pose(i, 3);
        }
    }
    static void GetCameraPara(cv::Mat &camk) { camk = mcamk.clone(); } /* src/Frame.cpp:207-210 */
    static void SetCameraPara(float fx, float fy, float cx, float cy) {
        mcamk.create(3, 3, CV_32F);
        for (int i = 0; i < 9; i++) mcamk.ptr<float>()[i] = 0.f;
        mcamk.at<float>(0, 0) = fx;
        mcamk.at<float>(1, 1) = fy;
        mcamk.at<float>(0, 2) = cx;
        mcamk.at<float>(1, 2) = cy;
        mcamk.at<float>(2, 2) = 1.f;
    }
    std::vector<MapPoint *> mvpMappts;
    std::vector<bool> mvbOutlier;
    KeyFrame *mpReferenceKF;
    std::vector<cv::KeyPoint> mvKeypts, mvUnKeypts; /* private in the reference; the test fills them */
    cv::Mat mcvDescriptors;

  private:
    ORBextractor *mpOrbextractor;
    static cv::Mat mcamk;
    Eigen::Matrix3d mRwc;
    Eigen::Vector3d mtwc;
    Eigen::Matrix4d mTwc;
};

class KeyFrame {
  public:
    KeyFrame() : mlnId(0), mnFrameId(0), mbBad(false) {}
    long unsigned int mlnId, mnFrameId;
    bool IsBad() const { return mbBad; }
    Eigen::Matrix3d GetR() const { return mKfRwc; }
    void SetR(const Eigen::Matrix3d &rot) { mKfRwc = rot; }
    Eigen::Vector3d Gettrans() const { return mKftwc; }
    void Settrans(const Eigen::Vector3d &trans) { mKftwc = trans; }
    std::vector<MapPoint *> GetMapPoints() const { return mvpMappts; }
    void AddMapPoint(MapPoint *p, int id) {
        if ((int)mvpMappts.size() <= id) mvpMappts.resize(id + 1, nullptr);
        mvpMappts[id] = p;
    }
    std::vector<cv::KeyPoint> mvUnKeypts;
    cv::Mat mcvDescriptors; /* include/KeyFrame.h:64 */
    static float fx, fy, cx, cy;

  private:
    Eigen::Matrix3d mKfRwc;
    Eigen::Vector3d mKftwc;
    std::vector<MapPoint *> mvpMappts;
    bool mbBad;
};

class MapPoint {
  public:
    MapPoint() : mlnId(0), mbBad(false) {}
    bool IsBad() const { return mbBad; }
    Eigen::Vector3d Getpos() const { return mWorldPos; }
    void SetPos(const Eigen::Vector3d &pt) { mWorldPos = pt; }
    std::map<KeyFrame *, int> GetObservations() const { return mObservations; }
    std::map<KeyFrame *, int> &GetObservations() { return mObservations; }
    void AddObservation(KeyFrame *pf, int id) { mObservations[pf] = id; }
    int Obsnums() const { return (int)mObservations.size(); } /* include/MapPoint.h:39 (nObs) */
    long unsigned int mlnId;

  private:
    Eigen::Vector3d mWorldPos;
    std::map<KeyFrame *, int> mObservations;
    bool mbBad;
};

class Map {
  public:
    void AddKeyFrame(KeyFrame *k) { mspKeyframes.insert(k); }
    void AddMapPoint(MapPoint *p) { mspMapPts.insert(p); }
    std::vector<KeyFrame *> GetKeyFrames() const { return std::vector<KeyFrame *>(mspKeyframes.begin(), mspKeyframes.end()); }
    std::vector<MapPoint *> GetMapPoints() const { return std::vector<MapPoint *>(mspMapPts.begin(), mspMapPts.end()); }

  private:
    std::set<KeyFrame *> mspKeyframes;
    std::set<MapPoint *> mspMapPts;
};
} // namespace ORBSlam
#endif
