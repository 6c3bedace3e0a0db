/*
 * ORBmatcher.h -- drop-in replacement of the reference's include/ORBmatcher.h:8-32 on top of the MI355X
 * C-ABI.  Like the reference header (include/ORBmatcher.h:4) it includes "Frame.h" and then defines
 * `class ORBSlam::ORBmatcher` with the reference's member functions, so src/Tracking.cpp:181-189
 *     ORBmatcher matcher(0.9, true);
 *     matcher.SearchForInitialization(minitialRefFrame, mCurframe, vbPrevMatched, mvIniMatches, 100);
 * compiles unchanged.  The implementation is the class template ORBmatcherT<FrameT> (usable on its own
 * where no "Frame.h" is on the include path: define ORBSLAM_AMD_NO_FRAME_INCLUDE or simply do not have one).
 *
 * Image bounds of Frame::GetFeaturesInArea's grid: the reference keeps them in PRIVATE statics
 * (include/Frame.h:75) with no getter.  Resolution order: (1) ORBmatcher::SetImageBounds, if called;
 * (2) `static void FrameT::GetImageBounds(float&, float&, float&, float&)` if the Frame type has it (the one-line
 * accessor INTEGRATION.md shows, needed for DISTORTED cameras only); (3) (0, cols, 0, rows) of the image most
 * recently given to an ORBextractor -- what Frame::FindimageBound computes for an undistorted camera
 * (src/Frame.cpp:113-119).  So with the reference as it is, nothing has to be called by hand.
 */
#ifndef ORBSLAM_AMD_ORBMATCHER_H
#define ORBSLAM_AMD_ORBMATCHER_H

#include <algorithm>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../orbslam_amd.h"
#include "ORBextractor.h"
#include "cv_shim.h"

#if !defined(ORBSLAM_AMD_NO_FRAME_INCLUDE) && defined(__has_include)
#if __has_include("Frame.h")
#include "Frame.h" /* reference include/ORBmatcher.h:4 */
#define ORBSLAM_AMD_HAVE_FRAME_H 1
#endif
#endif

namespace ORBSlam {

namespace detail {
template <class F, class = void> struct has_image_bounds : std::false_type {};
template <class F>
struct has_image_bounds<F, decltype(F::GetImageBounds(std::declval<float &>(), std::declval<float &>(), std::declval<float &>(),
                                                      std::declval<float &>()))> : std::true_type {};
template <class F> typename std::enable_if<has_image_bounds<F>::value, bool>::type frame_bounds(float *b) {
    F::GetImageBounds(b[0], b[1], b[2], b[3]);
    return b[1] > b[0] && b[3] > b[2];
}
template <class F> typename std::enable_if<!has_image_bounds<F>::value, bool>::type frame_bounds(float *) { return false; }
} // namespace detail

template <class FrameT> class ORBmatcherT {
  public:
    ORBmatcherT(float nnratio = 0.6, bool checkOri = true)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), m_(nullptr), cap_(0) {}
    ~ORBmatcherT() {} /* (the device handle belongs to the host thread, see ensure()) */
    ORBmatcherT(const ORBmatcherT &) = delete;
    ORBmatcherT &operator=(const ORBmatcherT &) = delete;

    /* optional override of the image bounds (see the header comment) */
    static void SetImageBounds(float minX, float maxX, float minY, float maxY) {
        float *b = explicit_bounds();
        b[0] = minX;
        b[1] = maxX;
        b[2] = minY;
        b[3] = maxY;
        b[4] = 1.f;
    }

    /* reference src/ORBmatcher.cpp:9-126 */
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<cv::Point2f> &vbPrevMatched,
                                std::vector<int> &vnMatches12, int windowSize) {
        const std::vector<cv::KeyPoint> &k1 = F1.GetUnKeyPts();
        const std::vector<cv::KeyPoint> &k2 = F2.GetUnKeyPts();
        const cv::Mat d1 = F1.GetDescriptors(), d2 = F2.GetDescriptors(); /* ONE clone each, not one per use */
        vnMatches12.assign(k1.size(), -1);                                  /* :13 */
        if (k1.empty()) return 0;
        ensure((int)std::max(k1.size(), k2.size()));
        std::vector<unsigned char> b1(k1.size() * 32), b2(k2.size() * 32);
        for (size_t i = 0; i < k1.size(); i++) memcpy(&b1[i * 32], d1.ptr((int)i), 32);
        for (size_t i = 0; i < k2.size(); i++) memcpy(&b2[i * 32], d2.ptr((int)i), 32);
        float b[4];
        bounds(b);
        OrbFrameView v1 = {(const OrbKeyPoint *)k1.data(), b1.data(), (int32_t)k1.size(), b[0], b[1], b[2], b[3]};
        OrbFrameView v2 = {(const OrbKeyPoint *)k2.data(), b2.data(), (int32_t)k2.size(), b[0], b[1], b[2], b[3]};
        static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f must be two floats");
        int32_t n = 0;
        if (orb_match_init(m_, &v1, &v2, (float *)vbPrevMatched.data(), vnMatches12.data(), windowSize, mfNNratio,
                           mbCheckOrientation ? 1 : 0, &n) != ORB_OK)
            orbslam_amd_fatal("ORBmatcher::SearchForInitialization");
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

    /* per-level scale factors for SearchByProjection's search radius (ORBextractor::GetScaleFactors()); default 1.2^l */
    static std::vector<float> &ScaleFactors() {
        static std::vector<float> v = [] {
            std::vector<float> t(8);
            float s = 1.f;
            for (int i = 0; i < 8; i++) {
                t[i] = s;
                s *= 1.2f;
            }
            return t;
        }();
        return v;
    }

    /* SURVEY N4.  Both functions have EMPTY bodies in the reference (include/ORBmatcher.h:22,24 -- `{}` with an int
     * return type) although Tracking calls them (src/Tracking.cpp:298,344,348): nothing in the reference to be identical
     * to.  They implement upstream ORB-SLAM2's monocular semantics on the reference's data model (stated line by line in
     * oracle/match_oracle.cpp; PARITY UNPINNED by construction) over orb_search_by_projection / orb_search_by_bow.
     *
     * SearchByProjection: every map point of the last frame (not an outlier) is projected with the current frame's
     * pose, the closest free feature inside th * scaleFactor[octave] on the neighbouring pyramid levels takes it if the
     * Hamming distance is <= 100; rotation-consistency histogram; CurrentFrame.mvpMappts is filled.  Needs of FrameT,
     * beyond SearchForInitialization's: GetR(), GetT(), static GetCameraPara(cv::Mat&), mvpMappts, mvbOutlier; of the
     * map point type: Getpos(), Obsnums(). */
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono) {
        (void)bMono; /* the reference is monocular throughout */
        const std::vector<cv::KeyPoint> kc = CurrentFrame.GetUnKeyPts(), kl = LastFrame.GetUnKeyPts();
        const cv::Mat dc = CurrentFrame.GetDescriptors(), dl = LastFrame.GetDescriptors();
        const int nc = (int)kc.size(), nl = (int)kl.size();
        if (nc == 0 || nl == 0) return 0;
        ensure(std::max(nc, nl));
        std::vector<unsigned char> bc((size_t)nc * 32), bl((size_t)nl * 32), flags(nl, 0), taken(nc, 0);
        for (int i = 0; i < nc; i++) memcpy(&bc[(size_t)i * 32], dc.ptr(i), 32);
        for (int i = 0; i < nl; i++) memcpy(&bl[(size_t)i * 32], dl.ptr(i), 32);
        std::vector<double> xw((size_t)nl * 3, 0.0);
        for (int i = 0; i < nl && i < (int)LastFrame.mvpMappts.size(); i++) {
            auto *mp = LastFrame.mvpMappts[i];
            if (!mp || (i < (int)LastFrame.mvbOutlier.size() && LastFrame.mvbOutlier[i])) continue;
            const auto X = mp->Getpos();
            for (int k = 0; k < 3; k++) xw[(size_t)i * 3 + k] = X(k);
            flags[i] = (unsigned char)(1 | (mp->Obsnums() > 0 ? 2 : 0));
        }
        if ((int)CurrentFrame.mvpMappts.size() < nc) CurrentFrame.mvpMappts.resize(nc, nullptr);
        for (int j = 0; j < nc; j++) {
            auto *mp = CurrentFrame.mvpMappts[j];
            taken[j] = (mp && mp->Obsnums() > 0) ? 1 : 0;
        }
        const auto R = CurrentFrame.GetR();
        const auto t = CurrentFrame.GetT();
        double Rcw[9], tcw[3];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) Rcw[i * 3 + j] = R(i, j);
            tcw[i] = t(i);
        }
        cv::Mat camk;
        CurrentFrame.GetCameraPara(camk);
        float b[4];
        bounds(b);
        OrbFrameView vc = {(const OrbKeyPoint *)kc.data(), bc.data(), nc, b[0], b[1], b[2], b[3]};
        OrbFrameView vl = {(const OrbKeyPoint *)kl.data(), bl.data(), nl, b[0], b[1], b[2], b[3]};
        std::vector<int32_t> match(nc, -1);
        int32_t n = 0;
        if (orb_search_by_projection(m_, &vc, &vl, flags.data(), xw.data(), Rcw, tcw, (double)camk.template at<float>(0, 0),
                                     (double)camk.template at<float>(1, 1), (double)camk.template at<float>(0, 2),
                                     (double)camk.template at<float>(1, 2), ScaleFactors().data(), (int)ScaleFactors().size(),
                                     taken.data(), th, mbCheckOrientation ? 1 : 0, match.data(), &n) != ORB_OK)
            orbslam_amd_fatal("ORBmatcher::SearchByProjection");
        for (int j = 0; j < nc; j++)
            if (match[j] >= 0) CurrentFrame.mvpMappts[j] = LastFrame.mvpMappts[match[j]];
        return n;
    }

    /* SearchByBoW: the reference has no vocabulary (KeyFrame::ComputeBoW / Frame::ComputeBOW are empty), i.e. a single
     * node: every keyframe feature with a good map point against every frame feature not matched yet, TH_LOW and the
     * nearest-neighbour ratio, rotation histogram.  vpMatches12[j] = the map point frame feature j receives.  KF needs
     * GetMapPoints(), mvUnKeypts, mcvDescriptors; the frame is taken by value as in the reference's signature. */
    template <class KF, class MP> int SearchByBoW(KF *pKF, FrameT F, std::vector<MP *> &vpMatches12) {
        const std::vector<cv::KeyPoint> &kk = pKF->mvUnKeypts;
        const std::vector<cv::KeyPoint> kf = F.GetUnKeyPts();
        const cv::Mat dk = pKF->mcvDescriptors, df = F.GetDescriptors();
        const int nk = (int)kk.size(), nf = (int)kf.size();
        vpMatches12.assign(nf, static_cast<MP *>(nullptr));
        if (nk == 0 || nf == 0) return 0;
        ensure(std::max(nk, nf));
        const std::vector<MP *> mps = pKF->GetMapPoints();
        std::vector<unsigned char> bk((size_t)nk * 32), bf((size_t)nf * 32), valid(nk, 0);
        for (int i = 0; i < nk; i++) memcpy(&bk[(size_t)i * 32], dk.ptr(i), 32);
        for (int i = 0; i < nf; i++) memcpy(&bf[(size_t)i * 32], df.ptr(i), 32);
        for (int i = 0; i < nk && i < (int)mps.size(); i++) valid[i] = (mps[i] && !mps[i]->IsBad()) ? 1 : 0;
        float b[4];
        bounds(b);
        OrbFrameView vk = {(const OrbKeyPoint *)kk.data(), bk.data(), nk, b[0], b[1], b[2], b[3]};
        OrbFrameView vf = {(const OrbKeyPoint *)kf.data(), bf.data(), nf, b[0], b[1], b[2], b[3]};
        std::vector<int32_t> match(nf, -1);
        int32_t n = 0;
        if (orb_search_by_bow(m_, &vk, valid.data(), &vf, mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), &n) != ORB_OK)
            orbslam_amd_fatal("ORBmatcher::SearchByBoW");
        for (int j = 0; j < nf; j++)
            if (match[j] >= 0) vpMatches12[j] = mps[match[j]];
        return n;
    }

  protected:
    static float *explicit_bounds() {
        static float b[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        return b;
    }
    static void bounds(float *b) {
        const float *e = explicit_bounds();
        if (e[4] != 0.f) {
            for (int i = 0; i < 4; i++) b[i] = e[i];
            return;
        }
        if (detail::frame_bounds<FrameT>(b)) return;
        b[0] = 0.f;
        b[1] = (float)LastExtractedImageSize()[0];
        b[2] = 0.f;
        b[3] = (float)LastExtractedImageSize()[1];
        if (!(b[1] > 0.f && b[3] > 0.f)) { /* no extractor ran in this process: VGA, the reference's test camera */
            b[1] = 640.f;
            b[3] = 480.f;
        }
    }
    /* The reference constructs an ORBmatcher wherever it needs one (src/Tracking.cpp:293, 333: once per frame), which
       costs it nothing; a device handle (streams, device and pinned buffers) per object would cost milliseconds per
       frame.  The handle therefore belongs to the HOST THREAD -- one per thread, grown on demand, released when the
       thread ends -- and every ORBmatcher object of that thread borrows it (nnratio / checkOri travel with each call). */
    struct ThreadHandle {
        OrbMatcher *m = nullptr;
        int cap = 0;
        ~ThreadHandle() { orb_matcher_destroy(m); }
    };
    static ThreadHandle &thread_handle() {
        static thread_local ThreadHandle h;
        return h;
    }
    void ensure(int n) {
        ThreadHandle &h = thread_handle();
        if (!(h.m && n <= h.cap)) {
            orb_matcher_destroy(h.m);
            h.m = nullptr;
            h.cap = n < 4096 ? 4096 : n;
            if (orb_matcher_create(h.cap, 1, 0, &h.m) != ORB_OK) orbslam_amd_fatal("ORBmatcher");
        }
        m_ = h.m;
        cap_ = h.cap;
    }
    static const int HISTO_LENGTH = 30;
    static const int TH_LOW = 50;
    float mfNNratio;
    bool mbCheckOrientation;
    OrbMatcher *m_;
    int cap_;
};

#ifdef ORBSLAM_AMD_HAVE_FRAME_H
/* the reference's class, by its own name (include/ORBmatcher.h:8) */
class ORBmatcher : public ORBmatcherT<Frame> {
  public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : ORBmatcherT<Frame>(nnratio, checkOri) {}
};
#endif

} // namespace ORBSlam
#endif
