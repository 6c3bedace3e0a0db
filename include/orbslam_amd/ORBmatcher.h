/*
 * ORBmatcher.h -- drop-in replacement of the reference's include/ORBmatcher.h:8-32 on top of the MI355X
 * C-ABI.  Same member functions; the class is a template over the Frame type so that it compiles both
 * inside the reference tree (where `Frame` is the reference's class) and in this repo's tests.
 *
 * The reference's Frame keeps its image bounds in PRIVATE statics (include/Frame.h:75) and exposes no
 * getter; the matcher needs them to reproduce GetFeaturesInArea's grid.  INTEGRATION.md shows the
 * one-line accessor to add; until then the bounds default to (0, cols, 0, rows) through
 * ORBmatcher::SetImageBounds (what Frame::FindimageBound computes for an undistorted camera,
 * src/Frame.cpp:113-119).
 */
#ifndef ORBSLAM_AMD_ORBMATCHER_H
#define ORBSLAM_AMD_ORBMATCHER_H

#include <stdexcept>
#include <string>
#include <vector>

#include "../orbslam_amd.h"
#include "cv_shim.h"

namespace ORBSlam {

template <class FrameT, class KeyFrameT = void, class MapPointT = void> class ORBmatcherT {
  public:
    ORBmatcherT(float nnratio = 0.6, bool checkOri = true)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), m_(nullptr), cap_(0) {}
    ~ORBmatcherT() { orb_matcher_destroy(m_); }
    ORBmatcherT(const ORBmatcherT &) = delete;
    ORBmatcherT &operator=(const ORBmatcherT &) = delete;

    static void SetImageBounds(float minX, float maxX, float minY, float maxY) {
        bounds()[0] = minX;
        bounds()[1] = maxX;
        bounds()[2] = minY;
        bounds()[3] = maxY;
    }

    /* reference src/ORBmatcher.cpp:9-126 */
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched,
                                std::vector<int> &vnMatches12, int windowSize) {
        const std::vector<cv::KeyPoint> &k1 = F1.GetUnKeyPts();
        const std::vector<cv::KeyPoint> &k2 = F2.GetUnKeyPts();
        const cv::Mat d1 = F1.GetDescriptors(), d2 = F2.GetDescriptors(); /* ONE clone each, not one per use */
        vnMatches12.assign(k1.size(), -1);
        if (k1.empty()) return 0;
        ensure((int)std::max(k1.size(), k2.size()));
        std::vector<unsigned char> b1(k1.size() * 32), b2(k2.size() * 32);
        for (size_t i = 0; i < k1.size(); i++) memcpy(&b1[i * 32], d1.ptr((int)i), 32);
        for (size_t i = 0; i < k2.size(); i++) memcpy(&b2[i * 32], d2.ptr((int)i), 32);
        OrbFrameView v1 = {(const OrbKeyPoint *)k1.data(), b1.data(), (int32_t)k1.size(), bounds()[0], bounds()[1],
                           bounds()[2], bounds()[3]};
        OrbFrameView v2 = {(const OrbKeyPoint *)k2.data(), b2.data(), (int32_t)k2.size(), bounds()[0], bounds()[1],
                           bounds()[2], bounds()[3]};
        static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f must be two floats");
        int32_t n = 0;
        int rc = orb_match_init(m_, &v1, &v2, (float *)vbPrevMatched.data(), vnMatches12.data(), windowSize, mfNNratio,
                                mbCheckOrientation ? 1 : 0, &n);
        if (rc != ORB_OK) throw std::runtime_error("ORBmatcher::SearchForInitialization failed");
        return n;
    }

    /* reference src/ORBmatcher.cpp:128-144 */
    int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) { return orb_descriptor_distance(a.ptr(), b.ptr()); }

    /* reference src/ORBmatcher.cpp:147-188 */
    void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
        std::vector<int32_t> sizes(L);
        for (int i = 0; i < L; i++) sizes[i] = (int32_t)histo[i].size();
        int32_t a = ind1, b = ind2, c = ind3;
        orb_compute_three_maxima(sizes.data(), L, &a, &b, &c);
        ind1 = a;
        ind2 = b;
        ind3 = c;
    }

    /* empty-bodied stubs in the reference (include/ORBmatcher.h:22,24): kept as explicit "no matches" */
    template <class KF, class MP> int SearchByBoW(KF *, FrameT, std::vector<MP *> &) { return 0; }
    int SearchByProjection(FrameT &, const FrameT &, const float, const bool) { return 0; }

  private:
    static float *bounds() {
        static float b[4] = {0.f, 640.f, 0.f, 480.f};
        return b;
    }
    void ensure(int n) {
        if (m_ && n <= cap_) return;
        orb_matcher_destroy(m_);
        m_ = nullptr;
        cap_ = n < 4096 ? 4096 : n;
        if (orb_matcher_create(cap_, 1, 0, &m_) != ORB_OK) throw std::runtime_error("ORBmatcher: no HIP device");
    }
    static const int HISTO_LENGTH = 30;
    static const int TH_LOW = 50;
    float mfNNratio;
    bool mbCheckOrientation;
    OrbMatcher *m_;
    int cap_;
};

} // namespace ORBSlam
#endif
