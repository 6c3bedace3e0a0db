/*
 * cv_shim.h -- the small subset of OpenCV types that the reference's hot-path signatures mention
 * (cv::Mat, cv::KeyPoint, cv::Point2f, cv::InputArray / cv::OutputArray, CV_8UC1), for builds where
 * OpenCV is not installed (this image has none).  With real OpenCV available define
 * ORBSLAM_AMD_USE_OPENCV before including the drop-in headers and this file forwards to it instead.
 */
#ifndef ORBSLAM_AMD_CV_SHIM_H
#define ORBSLAM_AMD_CV_SHIM_H

#ifdef ORBSLAM_AMD_USE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
    Point_ &operator*=(T s) {
        x *= s;
        y *= s;
        return *this;
    }
};
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point2i Point;

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};

/* 28-byte POD, layout-compatible with OrbKeyPoint of the C-ABI */
struct KeyPoint {
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
        : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

/* reference-counted 2-D 8-bit matrix (single channel), enough for images and N x 32 descriptors */
class Mat {
  public:
    int rows, cols;
    size_t step;
    unsigned char *data;
    Mat() : rows(0), cols(0), step(0), data(nullptr) {}
    Mat(int r, int c, int type) : rows(0), cols(0), step(0), data(nullptr) { create(r, c, type); }
    Mat(int r, int c, int /*type*/, void *ext, size_t stp = 0)
        : rows(r), cols(c), step(stp ? stp : (size_t)c), data((unsigned char *)ext) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && buf_) return;
        rows = r;
        cols = c;
        step = (size_t)c;
        buf_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)r * c > 0 ? (size_t)r * c : 1],
                                              std::default_delete<unsigned char[]>());
        data = buf_.get();
    }
    void release() {
        buf_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    Mat clone() const {
        Mat m;
        if (!empty()) {
            m.create(rows, cols, CV_8UC1);
            for (int y = 0; y < rows; y++) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, cols);
        }
        return m;
    }
    Mat row(int y) const {
        Mat m(*this);
        m.rows = 1;
        m.data = data + (size_t)y * step;
        return m;
    }
    template <typename T> T *ptr(int y = 0) { return (T *)(data + (size_t)y * step); }
    template <typename T> const T *ptr(int y = 0) const { return (const T *)(data + (size_t)y * step); }
    unsigned char *ptr(int y = 0) { return data + (size_t)y * step; }
    const unsigned char *ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T &at(int y, int x) { return ((T *)(data + (size_t)y * step))[x]; }
    template <typename T> const T &at(int y, int x) const { return ((const T *)(data + (size_t)y * step))[x]; }
    Mat getMat() const { return *this; }

  private:
    std::shared_ptr<unsigned char> buf_;
};

typedef const Mat &InputArray;
typedef Mat &OutputArray;

} // namespace cv
#endif /* ORBSLAM_AMD_USE_OPENCV */
#endif
