/*
 * ORBextractor.h -- drop-in replacement of the reference's include/ORBextractor.h:29-97 on top of the
 * MI355X C-ABI (include/orbslam_amd.h).  Same namespace, class name, constructor, call operator,
 * getters and public mvImagePyramid member, so src/Frame.cpp:77
 *     (*mpOrbextractor)(curImg, cv::Mat(), mvKeypts, mcvDescriptors);
 * and src/Tracking.cpp:47-48 (`new ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST)`)
 * compile and behave unchanged.  Header-only; link with -lorbslam_amd.
 */
#ifndef ORBSLAM_AMD_ORBEXTRACTOR_H
#define ORBSLAM_AMD_ORBEXTRACTOR_H

#include <cassert>
#include <list>
#include <stdexcept>
#include <string>
#include <vector>

#include "../orbslam_amd.h"
#include "cv_shim.h"

namespace ORBSlam {

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

    /* max_w / max_h only size the device buffers; larger frames grow them on first use */
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int max_w = 1280,
                 int max_h = 720, int device = 0)
        : h_(nullptr), nlevels_(nlevels), last_w_(0), last_h_(0), pyramid_valid_(false) {
        OrbConfig cfg = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
        int rc = orb_create(&cfg, max_w, max_h, 1, device, &h_);
        if (rc != ORB_OK) throw std::runtime_error(std::string("ORBextractor: ") + orb_last_error());
        scale_.resize(nlevels);
        inv_scale_.resize(nlevels);
        sigma2_.resize(nlevels);
        inv_sigma2_.resize(nlevels);
        orb_get_tables(h_, scale_.data(), inv_scale_.data(), sigma2_.data(), inv_sigma2_.data(), nullptr, nullptr);
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
        if (rc != ORB_OK) throw std::runtime_error(std::string("ORBextractor: ") + orb_last_error());
        last_w_ = img.cols;
        last_h_ = img.rows;
        pyramid_valid_ = false;
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

    /* public member of the reference (include/ORBextractor.h:71).  The pyramid lives in HBM; call
     * DownloadImagePyramid() to materialise the levels (each an ROI-equivalent copy WITHOUT the apron;
     * the apron-carrying backing image is available through orb_download_level). */
    std::vector<cv::Mat> mvImagePyramid;
    void DownloadImagePyramid() {
        if (pyramid_valid_ || last_w_ == 0) return;
        std::vector<int32_t> ws(nlevels_), hs(nlevels_);
        orb_get_level_sizes(h_, last_w_, last_h_, ws.data(), hs.data());
        std::vector<unsigned char> padded;
        for (int l = 0; l < nlevels_; l++) {
            const int W = ws[l] + 38, H = hs[l] + 38;
            padded.resize((size_t)W * H);
            if (orb_download_level(h_, 0, l, 0, padded.data()) != ORB_OK) return;
            mvImagePyramid[l].create(hs[l], ws[l], CV_8UC1);
            for (int y = 0; y < hs[l]; y++)
                memcpy(mvImagePyramid[l].ptr(y), &padded[(size_t)(y + 19) * W + 19], ws[l]);
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
