/*
 * ORBextractor.h -- drop-in replacement of the reference's include/ORBextractor.h:29-97 on top of the
 * MI355X C-ABI (include/orbslam_amd.h).  Same namespace, class name, constructor, call operator,
 * getters and public mvImagePyramid member, so src/Frame.cpp:77
 *     (*mpOrbextractor)(curImg, cv::Mat(), mvKeypts, mcvDescriptors);
 * and src/Tracking.cpp:47-48 (`new ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST)`)
 * compile and behave unchanged.  Header-only; link with -lorbslam_amd.
 *
 * Error model = the reference's (SURVEY 8b "Errors"): no exceptions and no status codes on any path the
 * reference has.  Empty image -> silent return; wrong type -> assert.  The one situation the reference cannot
 * be in -- no usable HIP device / a HIP runtime failure -- is fatal: message on stderr, abort() (there is no
 * CPU fallback to continue on).
 */
#ifndef ORBSLAM_AMD_ORBEXTRACTOR_H
#define ORBSLAM_AMD_ORBEXTRACTOR_H

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <string>
#include <vector>

#include "../orbslam_amd.h"
#include "cv_shim.h"

namespace ORBSlam {

/* loud failure of the device path (never reached on the reference's own paths) */
[[noreturn]] inline void orbslam_amd_fatal(const char *where) {
    std::fprintf(stderr, "orbslam_amd: %s: %s\n", where, orb_last_error());
    std::abort();
}

/* size of the frame most recently handed to any ORBextractor of this process: Frame::FindimageBound
 * (src/Frame.cpp:111-119) computes (0, cols, 0, rows) from exactly this image for an undistorted camera, and the
 * reference keeps the result in private statics -- ORBmatcher reads it here when no bounds were set */
inline int *LastExtractedImageSize() {
    static int wh[2] = {0, 0};
    return wh;
}

/* public in the reference header (include/ORBextractor.h:16-27); the quadtree now lives on the device
 * (orbslam_in_practice_amd/csrc/orb_quadtree.hip), the type is kept for source compatibility */
class ExtractorNode {
  public:
    ExtractorNode() : bNoMore(false) {}
    std::vector<cv::KeyPoint> vKeys;
    cv::Point2i UL, UR, BL, BR;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore;
};

class ORBextractor {
  public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    /* the reference's public `std::vector<cv::Mat> mvImagePyramid` (include/ORBextractor.h:71).  The levels live in
     * HBM; a reader that touches the member (operator[], size(), iteration, conversion to the vector) gets them
     * downloaded at that moment -- each level the reference's ROI at (19,19) inside its BORDER_REFLECT_101 padded
     * image (src/ORBextractor.cpp:1078-1093), apron included and reachable through negative offsets as there. */
    class Pyramid {
      public:
        cv::Mat &operator[](size_t l) { return fill()[l]; }
        size_t size() const { return v_.size(); }
        bool empty() const { return v_.empty(); }
        std::vector<cv::Mat>::iterator begin() { return fill().begin(); }
        std::vector<cv::Mat>::iterator end() { return fill().end(); }
        cv::Mat &at(size_t l) { return fill().at(l); }
        cv::Mat &front() { return fill().front(); }
        cv::Mat &back() { return fill().back(); }
        operator std::vector<cv::Mat> &() { return fill(); }
        void resize(size_t n) { v_.resize(n); } /* src/ORBextractor.cpp:383 */

      private:
        friend class ORBextractor;
        std::vector<cv::Mat> &fill() {
            if (owner_) owner_->DownloadImagePyramid();
            return v_;
        }
        ORBextractor *owner_ = nullptr;
        std::vector<cv::Mat> v_;
    };

    /* max_w / max_h only size the device buffers; larger frames grow them on first use */
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int max_w = 1280,
                 int max_h = 720, int device = 0)
        : h_(nullptr), nlevels_(nlevels), last_w_(0), last_h_(0), pyramid_valid_(true) {
        OrbConfig cfg = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
        if (orb_create(&cfg, max_w, max_h, 1, device, &h_) != ORB_OK) orbslam_amd_fatal("ORBextractor");
        scale_.resize(nlevels);
        inv_scale_.resize(nlevels);
        sigma2_.resize(nlevels);
        inv_sigma2_.resize(nlevels);
        orb_get_tables(h_, scale_.data(), inv_scale_.data(), sigma2_.data(), inv_sigma2_.data(), nullptr, nullptr);
        mvImagePyramid.owner_ = this;
        mvImagePyramid.resize(nlevels);
        cap_ = orb_max_keypoints(h_);
    }
    ~ORBextractor() { orb_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    /* reference src/ORBextractor.cpp:1001-1065.  Mask is ignored, as in the reference. */
    void operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors) {
        if (image.empty()) return; /* :1004-1005 */
        cv::Mat img = image.getMat();
        assert(img.type() == CV_8UC1); /* :1008 */
        static_assert(sizeof(cv::KeyPoint) == sizeof(OrbKeyPoint), "cv::KeyPoint must be the 28-byte POD");
        int32_t n = 0;
        int rc = ORB_OK;
        for (int attempt = 0; attempt < 2; attempt++) {
            kp_buf_.resize(cap_);
            desc_buf_.resize((size_t)cap_ * 32);
            rc = orb_extract(h_, img.data, img.cols, img.rows, (int)img.step, (OrbKeyPoint *)kp_buf_.data(),
                             desc_buf_.data(), cap_, &n);
            /* a frame shape with a larger bound than the creation size (wide frames: 4 * nIni keypoints per level):
               the handle now holds its geometry, ask again and retry once */
            if (rc == ORB_E_CAPACITY && attempt == 0 && orb_max_keypoints(h_) > cap_) {
                cap_ = orb_max_keypoints(h_);
                continue;
            }
            break;
        }
        if (rc == ORB_E_ASPECT) { /* the reference divides by zero here (src/ORBextractor.cpp:493-495): no features */
            keypoints.clear();
            descriptors.release();
            return;
        }
        if (rc != ORB_OK) orbslam_amd_fatal("ORBextractor::operator()");
        last_w_ = img.cols;
        last_h_ = img.rows;
        LastExtractedImageSize()[0] = img.cols;
        LastExtractedImageSize()[1] = img.rows;
        pyramid_valid_ = false; /* mvImagePyramid is refreshed when somebody reads it */
        if (n == 0)
            descriptors.release(); /* :1024-1025 */
        else {
            descriptors.create(n, 32, CV_8U);
            for (int i = 0; i < n; i++) memcpy(descriptors.ptr(i), &desc_buf_[(size_t)i * 32], 32);
        }
        keypoints.assign(kp_buf_.begin(), kp_buf_.begin() + n);
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return orb_get_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return scale_; }
    std::vector<float> inline GetInverseScaleFactors() { return inv_scale_; }
    std::vector<float> inline GetScaleSigmaSquares() { return sigma2_; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return inv_sigma2_; }

    Pyramid mvImagePyramid;

    /* what reading mvImagePyramid triggers; callable directly as well */
    void DownloadImagePyramid() {
        if (pyramid_valid_ || last_w_ == 0) return;
        std::vector<int32_t> ws(nlevels_), hs(nlevels_);
        orb_get_level_sizes(h_, last_w_, last_h_, ws.data(), hs.data());
        for (int l = 0; l < nlevels_; l++) {
            cv::Mat padded(hs[l] + 38, ws[l] + 38, CV_8UC1);
            if (orb_download_level(h_, 0, l, 0, padded.data) != ORB_OK) orbslam_amd_fatal("ORBextractor::mvImagePyramid");
            mvImagePyramid.v_[l] = padded(cv::Rect(19, 19, ws[l], hs[l])); /* :1081 */
        }
        pyramid_valid_ = true;
    }
    OrbExtractor *handle() { return h_; }

  protected:
    OrbExtractor *h_;
    int nlevels_, cap_, last_w_, last_h_;
    bool pyramid_valid_;
    std::vector<float> scale_, inv_scale_, sigma2_, inv_sigma2_;
    std::vector<cv::KeyPoint> kp_buf_;
    std::vector<unsigned char> desc_buf_;
};

} // namespace ORBSlam
#endif
