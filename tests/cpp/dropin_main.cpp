/*
 * dropin_main.cpp -- exercises the C++ drop-in classes (include/orbslam_amd/*.h) exactly the way the
 * reference's Frame / Tracking call them, with minimal stand-ins for the reference's data-model classes
 * (Frame, KeyFrame, MapPoint, Map expose the same accessor names).  Reads a raw gray frame pair,
 * writes keypoints / descriptors / matches / BA results to stdout files for the Python test to compare
 * with the C-ABI path and the oracle.
 *
 *   dropin_main <w> <h> <frame0.raw> <frame1.raw> <out_prefix>
 */
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#include "orbslam_amd/ORBextractor.h"
#include "orbslam_amd/ORBmatcher.h"
#include "orbslam_amd/Optimizer.h"

struct M3 {
    double v[9];
    double &operator()(int i, int j) { return v[i * 3 + j]; }
    double operator()(int i, int j) const { return v[i * 3 + j]; }
};
struct M4 {
    double v[16];
    double &operator()(int i, int j) { return v[i * 4 + j]; }
    double operator()(int i, int j) const { return v[i * 4 + j]; }
};
struct V3 {
    double v[3];
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};

namespace ORBSlam {
class MapPoint;
/* the accessors of reference include/Frame.h that the hot path touches */
class Frame {
  public:
    Frame() {}
    Frame(const cv::Mat &img, ORBextractor *ex) : mpOrbextractor(ex) {
        (*mpOrbextractor)(img, cv::Mat(), mvKeypts, mcvDescriptors); /* src/Frame.cpp:75-78 */
        mvUnKeypts = mvKeypts;                                        /* no distortion: src/Frame.cpp:82-86 */
    }
    std::vector<cv::KeyPoint> &GetUnKeyPts() { return mvUnKeypts; }
    cv::Mat GetDescriptors() const { return mcvDescriptors.clone(); }
    ORBextractor *mpOrbextractor;
    std::vector<cv::KeyPoint> mvKeypts, mvUnKeypts;
    cv::Mat mcvDescriptors;
};
class KeyFrame {
  public:
    static float fx, fy, cx, cy;
    unsigned long mlnId;
    std::vector<cv::KeyPoint> mvUnKeypts;
    bool IsBad() const { return false; }
    M3 GetR() const { return R; }
    V3 Gettrans() const { return t; }
    void SetR(const M3 &r) { R = r; }
    void Settrans(const V3 &v) { t = v; }
    M3 R;
    V3 t;
};
float KeyFrame::fx, KeyFrame::fy, KeyFrame::cx, KeyFrame::cy;
class MapPoint {
  public:
    unsigned long mlnId;
    bool IsBad() const { return false; }
    V3 Getpos() const { return p; }
    void SetPos(const V3 &v) { p = v; }
    std::map<KeyFrame *, int> &GetObservations() { return obs; }
    V3 p;
    std::map<KeyFrame *, int> obs;
};
class Map {
  public:
    std::vector<KeyFrame *> GetKeyFrames() const { return std::vector<KeyFrame *>(kfs.begin(), kfs.end()); }
    std::vector<MapPoint *> GetMapPoints() const { return std::vector<MapPoint *>(mps.begin(), mps.end()); }
    std::vector<KeyFrame *> kfs;
    std::vector<MapPoint *> mps;
};
typedef ORBmatcherT<Frame> ORBmatcher;
} // namespace ORBSlam

static std::vector<unsigned char> read_file(const char *p, size_t n) {
    std::vector<unsigned char> b(n);
    FILE *f = fopen(p, "rb");
    if (!f || fread(b.data(), 1, n, f) != n) {
        fprintf(stderr, "cannot read %s\n", p);
        exit(2);
    }
    fclose(f);
    return b;
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    using namespace ORBSlam;
    const int w = atoi(argv[1]), h = atoi(argv[2]);
    std::vector<unsigned char> b0 = read_file(argv[3], (size_t)w * h), b1 = read_file(argv[4], (size_t)w * h);
    cv::Mat im0(h, w, CV_8UC1, b0.data()), im1(h, w, CV_8UC1, b1.data());
    /* src/Tracking.cpp:47-48: two extractors, the initialisation one with 2*nFeatures */
    ORBextractor *mpIniOrbextractor = new ORBextractor(2 * 1000, 1.2f, 8, 20, 7, w, h);
    Frame F0(im0, mpIniOrbextractor), F1(im1, mpIniOrbextractor);
    /* src/Tracking.cpp:170-189 */
    std::vector<cv::Point2f> vbPrevMatched(F0.GetUnKeyPts().size());
    for (size_t i = 0; i < vbPrevMatched.size(); i++) vbPrevMatched[i] = F0.GetUnKeyPts()[i].pt;
    std::vector<int> vnMatches12;
    ORBmatcher::SetImageBounds(0.f, (float)w, 0.f, (float)h);
    ORBmatcher matcher(0.9f, true);
    int nmatches = matcher.SearchForInitialization(F0, F1, vbPrevMatched, vnMatches12, 100);

    std::string pre(argv[5]);
    FILE *f = fopen((pre + "_f0.bin").c_str(), "wb");
    int n0 = (int)F0.mvKeypts.size();
    fwrite(&n0, 4, 1, f);
    fwrite(F0.mvKeypts.data(), sizeof(cv::KeyPoint), n0, f);
    for (int i = 0; i < n0; i++) fwrite(F0.mcvDescriptors.ptr(i), 1, 32, f);
    fclose(f);
    f = fopen((pre + "_match.bin").c_str(), "wb");
    fwrite(&nmatches, 4, 1, f);
    fwrite(vnMatches12.data(), 4, vnMatches12.size(), f);
    fclose(f);

    /* a tiny map through Optimizer::GlobalBundleAdjustemnt (src/Tracking.cpp:254): 2 KFs looking down +z at
       a plane of points, second KF translated; observations are exact projections + the initial second
       pose is perturbed, so BA must pull it back */
    KeyFrame::fx = KeyFrame::fy = 525.f;
    KeyFrame::cx = 319.5f;
    KeyFrame::cy = 239.5f;
    Map map;
    KeyFrame k0, k1;
    k0.mlnId = 0;
    k1.mlnId = 1;
    const M3 I3 = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    k0.R = I3;
    k1.R = I3;
    k0.t = V3{{0, 0, 0}};
    const V3 t1_true = {{-0.5, 0.02, 0.01}};
    k1.t = V3{{-0.48, 0.05, -0.02}};
    std::vector<MapPoint> mps(60);
    for (int i = 0; i < 60; i++) {
        V3 X = {{-1.0 + 0.25 * (i % 10), -0.6 + 0.25 * (i / 10), 3.0 + 0.07 * ((i * 7) % 11)}};
        mps[i].mlnId = i;
        mps[i].p = V3{{X(0) + 0.01 * ((i % 3) - 1), X(1), X(2) + 0.02 * ((i % 5) - 2)}};
        cv::KeyPoint a((float)(525 * X(0) / X(2) + 319.5), (float)(525 * X(1) / X(2) + 239.5), 31.f);
        const double xc = X(0) + t1_true(0), yc = X(1) + t1_true(1), zc = X(2) + t1_true(2);
        cv::KeyPoint b((float)(525 * xc / zc + 319.5), (float)(525 * yc / zc + 239.5), 31.f);
        k0.mvUnKeypts.push_back(a);
        k1.mvUnKeypts.push_back(b);
        mps[i].obs[&k0] = i;
        mps[i].obs[&k1] = i;
        map.mps.push_back(&mps[i]);
    }
    map.kfs.push_back(&k0);
    map.kfs.push_back(&k1);
    Optimizer::GlobalBundleAdjustemnt(&map, 20);
    f = fopen((pre + "_ba.txt").c_str(), "w");
    /* scale gauge is free with one fixed KF: report the direction of t1 and the reprojection error */
    const double nt = std::sqrt(k1.t(0) * k1.t(0) + k1.t(1) * k1.t(1) + k1.t(2) * k1.t(2));
    const double ntt = std::sqrt(t1_true(0) * t1_true(0) + t1_true(1) * t1_true(1) + t1_true(2) * t1_true(2));
    double cosang = (k1.t(0) * t1_true(0) + k1.t(1) * t1_true(1) + k1.t(2) * t1_true(2)) / (nt * ntt);
    double err = 0;
    for (int i = 0; i < 60; i++) {
        const V3 &X = mps[i].p;
        const double xc = k1.R(0, 0) * X(0) + k1.R(0, 1) * X(1) + k1.R(0, 2) * X(2) + k1.t(0);
        const double yc = k1.R(1, 0) * X(0) + k1.R(1, 1) * X(1) + k1.R(1, 2) * X(2) + k1.t(1);
        const double zc = k1.R(2, 0) * X(0) + k1.R(2, 1) * X(1) + k1.R(2, 2) * X(2) + k1.t(2);
        const double du = 525 * xc / zc + 319.5 - k1.mvUnKeypts[i].pt.x, dv = 525 * yc / zc + 239.5 - k1.mvUnKeypts[i].pt.y;
        err += du * du + dv * dv;
    }
    fprintf(f, "%.12g %.12g\n", cosang, std::sqrt(err / 60));
    fclose(f);
    delete mpIniOrbextractor;
    printf("dropin ok: %d keypoints, %d matches\n", n0, nmatches);
    return 0;
}
