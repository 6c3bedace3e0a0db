/*
 * cv_shim.h -- the small subset of OpenCV types that the reference's hot-path signatures mention
 * (cv::Mat, cv::KeyPoint, cv::Point2f, cv::Rect, cv::InputArray / cv::OutputArray, CV_8UC1, CV_32F), for builds
 * where OpenCV is not installed (this image has none).  With real OpenCV available define
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
#define CV_32FC1 5
#define CV_64F 6
#define CV_64FC1 6

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

struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
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

/* reference-counted 2-D single-channel matrix (8U / 32F / 64F), enough for images, N x 32 descriptors and the
 * 3 x 3 camera matrix of Frame::GetCameraPara; an ROI shares its parent's buffer like cv::Mat(m, rect) */
class Mat {
  public:
    int rows, cols;
    size_t step;
    unsigned char *data;
    Mat() : rows(0), cols(0), step(0), data(nullptr), type_(CV_8UC1) {}
    Mat(int r, int c, int type) : rows(0), cols(0), step(0), data(nullptr), type_(CV_8UC1) { create(r, c, type); }
    Mat(int r, int c, int type, void *ext, size_t stp = 0)
        : rows(r), cols(c), step(stp ? stp : (size_t)c * esz(type)), data((unsigned char *)ext), type_(type) {}
    Mat(const Mat &m, const Rect &roi)
        : rows(roi.height), cols(roi.width), step(m.step), data(m.data + (size_t)roi.y * m.step + (size_t)roi.x * esz(m.type_)),
          type_(m.type_), buf_(m.buf_) {}
    Mat operator()(const Rect &roi) const { return Mat(*this, roi); }
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && buf_ && step == (size_t)c * esz(type)) return;
        rows = r;
        cols = c;
        type_ = type;
        step = (size_t)c * esz(type);
        const size_t n = (size_t)r * step;
        buf_ = std::shared_ptr<unsigned char>(new unsigned char[n > 0 ? n : 1], std::default_delete<unsigned char[]>());
        data = buf_.get();
    }
    void release() {
        buf_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    Mat clone() const {
        Mat m;
        if (!empty()) {
            m.create(rows, cols, type_);
            for (int y = 0; y < rows; y++) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * esz(type_));
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
    static size_t esz(int type) { return type == CV_32F ? 4 : (type == CV_64F ? 8 : 1); }
    int type_;
    std::shared_ptr<unsigned char> buf_;
};

typedef const Mat &InputArray;
typedef Mat &OutputArray;

} // namespace cv
#endif /* ORBSLAM_AMD_USE_OPENCV */
#endif
