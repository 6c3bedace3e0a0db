/*
 * orb_oracle.cpp -- CPU ORACLE for the ORB front-end (test infrastructure only).
 *
 * Restates reference src/ORBextractor.cpp (file:line cited per function) and the
 * OpenCV 2.4/3.x primitives it calls (cv::resize INTER_LINEAR, copyMakeBorder
 * REFLECT_101, cv::FAST TYPE_9_16, cv::GaussianBlur 7x7 sigma 2 on CV_8U,
 * cv::fastAtan2, cvRound).  OpenCV is NOT present in this image, so those
 * primitives are restated from their documented 2.4/3.x behaviour:
 * **parity unpinned** at that boundary (no reference tests/goldens exist).
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off  (no FMA contraction: the float
 * expressions must round exactly as the reference's SSE2 scalar code would).
 */
#include "oracle.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

namespace {

const int PATCH_SIZE = 31;      /* reference src/ORBextractor.cpp:22 */
const int HALF_PATCH_SIZE = 15; /* :23 */
const int EDGE_THRESHOLD = 19;  /* :24 */

static const int bit_pattern_31_[256 * 4] = {
#include "orb_pattern_data.inc"
};

/* cvRound(double): lrint, round-half-to-even (SSE2 cvtsd2si). */
inline int cvRound(double v) { return (int)std::lrint(v); }
inline int cvFloor(double v) {
    int i = (int)v;
    return i - (i > v);
}
inline int cvCeil(double v) {
    int i = (int)v;
    return i + (i < v);
}
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline short sat_s16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    void create(int w_, int h_) {
        w = w_;
        h = h_;
        d.assign((size_t)w * h, 0);
    }
    const uint8_t *row(int y) const { return d.data() + (size_t)y * w; }
    uint8_t *row(int y) { return d.data() + (size_t)y * w; }
};

/* ---- cv::resize(INTER_LINEAR) for CV_8UC1, OpenCV 2.4/3.x generic path --------------------
 * imgproc/src/imgwarp.cpp (resizeGeneric_ / HResizeLinear / VResizeLinear, fixed point with
 * INTER_RESIZE_COEF_BITS = 11).  Called by reference src/ORBextractor.cpp:1084. */
void resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                      int dstride) {
    const int COEF_SCALE = 2048;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx >= sw - 1) {
            fx = 0;
            sx = sw - 1;
        }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[dx * 2] = sat_s16(cvRound(c0 * COEF_SCALE));
        ialpha[dx * 2 + 1] = sat_s16(cvRound(c1 * COEF_SCALE));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[dy * 2] = sat_s16(cvRound(c0 * COEF_SCALE));
        ibeta[dy * 2 + 1] = sat_s16(cvRound(c1 * COEF_SCALE));
    }
    std::vector<int> r0(dw), r1(dw);
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    auto hresize = [&](int sy, std::vector<int> &D) {
        const uint8_t *S = src + (size_t)sy * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            D[dx] = S[sx] * ialpha[dx * 2] + S[sx1] * ialpha[dx * 2 + 1];
        }
    };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
        hresize(sy0, r0);
        hresize(sy1, r1);
        short b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

/* cv::borderInterpolate(BORDER_REFLECT_101) */
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (len - 1) - p;
    }
    return p;
}

/* copyMakeBorder(src, dst, b,b,b,b, BORDER_REFLECT_101) -- reference :1086,:1091 */
void border_reflect101_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int border) {
    int W = w + 2 * border;
    for (int y = -border; y < h + border; y++) {
        int sy = reflect101(y, h);
        for (int x = -border; x < w + border; x++)
            dst[(size_t)(y + border) * W + (x + border)] = src[(size_t)sy * sstride + reflect101(x, w)];
    }
}

/* ---- cv::GaussianBlur(Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8U, OpenCV 2.4/3.x path ----
 * getGaussianKernel(7, 2, CV_32F) -> float taps; createSeparableLinearFilter promotes an 8-bit
 * smooth symmetric kernel to fixed point: taps*256 rounded to int (bits=8 per pass), i32
 * accumulate, column pass FixedPtCastEx<int,uchar>(16): (v + 2^15) >> 16, saturate.
 * Called by reference src/ORBextractor.cpp:1045-1046 on a CLONE of the level (so the border
 * reflects at the image edge, not into the 19-px apron). */
void gaussian_taps7(int taps[7]) {
    const int n = 7;
    const double sigma = 2.0;
    float cf[7];
    double scale2X = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = (float)(cf[i] * sum);
        taps[i] = cvRound((double)cf[i] * 256.0); /* convertTo(CV_32S, 1<<8) */
    }
}

void gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    int k[7];
    gaussian_taps7(k);
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < 7; t++) s += k[t] * S[reflect101(x + t - 3, w)];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int t = 0; t < 7; t++) s += k[t] * tmp[(size_t)reflect101(y + t - 3, h) * w + x];
            D[x] = sat_u8((s + (1 << 15)) >> 16);
        }
    }
}

/* ---- cv::FAST(img, kps, threshold, nonmax) == FAST_t<16> (TYPE_9_16), features2d/src/fast.cpp ----
 * circle offsets (OpenCV makeOffsets, patternSize 16) */
static const int fast_off16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                      {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* cornerScore<16>: largest threshold for which the pixel is still a 9/16 corner (fast_score.cpp).
 * Written in closed form (max over the 16 nine-arcs of the min / max of v - p[k]); the early-outs of
 * the OpenCV loop only prune, they do not change the result. */
int fast_corner_score(const uint8_t *ptr, int stride, int threshold) {
    int d[25];
    int v = ptr[0];
    for (int k = 0; k < 16; k++) d[k] = v - ptr[fast_off16[k][1] * stride + fast_off16[k][0]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

/* is the pixel a 9/16 corner at `threshold` (>= 9 contiguous ring pixels all > v+t or all < v-t) */
inline bool fast_is_corner(const uint8_t *ptr, int stride, int threshold) {
    int v = ptr[0];
    { /* the usual high-speed rejection (OpenCV's FAST does the same through its threshold table): every 9-arc
         holds two adjacent compass pixels, so without such a pair on one side there is no corner */
        const int hi = v + threshold, lo = v - threshold;
        const int p0 = ptr[3 * stride], p4 = ptr[3], p8 = ptr[-3 * stride], p12 = ptr[-3];
        const bool b0 = p0 > hi, b4 = p4 > hi, b8 = p8 > hi, b12 = p12 > hi;
        const bool d0 = p0 < lo, d4 = p4 < lo, d8 = p8 < lo, d12 = p12 < lo;
        if (!((b0 && b4) || (b4 && b8) || (b8 && b12) || (b12 && b0) || (d0 && d4) || (d4 && d8) || (d8 && d12) ||
              (d12 && d0)))
            return false;
    }
    int ring[25];
    for (int k = 0; k < 16; k++) ring[k] = ptr[fast_off16[k][1] * stride + fast_off16[k][0]];
    for (int k = 16; k < 25; k++) ring[k] = ring[k - 16];
    int vt = v - threshold, count = 0;
    for (int k = 0; k < 25; k++) {
        if (ring[k] < vt) {
            if (++count > 8) return true;
        } else
            count = 0;
    }
    vt = v + threshold;
    count = 0;
    for (int k = 0; k < 25; k++) {
        if (ring[k] > vt) {
            if (++count > 8) return true;
        } else
            count = 0;
    }
    return false;
}

/* FAST on one ROI: scans rows [3,h-3), cols [3,w-3); score buffers are zero outside that zone, NMS
 * keeps a pixel iff its score is strictly greater than all 8 neighbours; output row-major.
 * KeyPoint(x, y, 7.f, -1, score). */
void fast9_16(const uint8_t *roi, int w, int h, int stride, int threshold, bool nonmax,
              std::vector<OrcKeyPoint> &out) {
    out.clear();
    if (w < 7 || h < 7) return;
    static thread_local std::vector<uint8_t> score, is_c;
    score.assign((size_t)w * h, 0);
    is_c.assign((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *p = roi + (size_t)y * stride + x;
            if (fast_is_corner(p, stride, threshold)) {
                is_c[(size_t)y * w + x] = 1;
                if (nonmax) score[(size_t)y * w + x] = (uint8_t)fast_corner_score(p, stride, threshold);
            }
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            if (!is_c[(size_t)y * w + x]) continue;
            if (nonmax) {
                int s = score[(size_t)y * w + x];
                bool keep = true;
                for (int dy = -1; dy <= 1 && keep; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        if (!(s > score[(size_t)(y + dy) * w + (x + dx)])) {
                            keep = false;
                            break;
                        }
                    }
                if (!keep) continue;
                out.push_back({(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1});
            } else
                out.push_back({(float)x, (float)y, 7.f, -1.f, 0.f, 0, -1});
        }
}

/* ---- cv::fastAtan2 (OpenCV 2.4 core/src/mathfuncs.cpp scalar path), degrees in [0,360) ---- */
float fast_atan2(float y, float x) {
    const float atan2_p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float atan2_p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float atan2_p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float atan2_p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* deterministic sin/cos (shared definition with the device code): Cody-Waite reduction by pi/2 in
 * double, fdlibm kernel polynomials, plain IEEE double mul/add (no FMA), rounded once to float. */
void sincos_det(float angle, float *c_out, float *s_out) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double x = (double)angle;
    double kd = std::nearbyint(x * two_over_pi);
    int k = (int)kd;
    double r = (x - kd * pio2_hi) - kd * pio2_lo;
    double z = r * r;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + (r * z) * ps;
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cs = (1.0 - 0.5 * z) + (z * z) * pc;
    double sv, cv;
    switch (k & 3) {
    case 0: sv = sn; cv = cs; break;
    case 1: sv = cs; cv = -sn; break;
    case 2: sv = -sn; cv = -cs; break;
    default: sv = -cs; cv = sn; break;
    }
    *c_out = (float)cv;
    *s_out = (float)sv;
}

struct ExtractorNode { /* reference include/ORBextractor.h:16-27 */
    std::vector<OrcKeyPoint> vKeys;
    int ULx = 0, ULy = 0, URx = 0, URy = 0, BLx = 0, BLy = 0, BRx = 0, BRy = 0;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore = false;
    long seq = 0; /* creation order: deterministic stand-in for the node's heap address (see :638) */
    void DivideNode(ExtractorNode &n1, ExtractorNode &n2, ExtractorNode &n3, ExtractorNode &n4);
};

/* reference src/ORBextractor.cpp:431-487 */
void ExtractorNode::DivideNode(ExtractorNode &n1, ExtractorNode &n2, ExtractorNode &n3, ExtractorNode &n4) {
    const int halfX = (int)std::ceil(static_cast<float>(URx - ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(BRy - ULy) / 2);
    n1.ULx = ULx; n1.ULy = ULy;
    n1.URx = ULx + halfX; n1.URy = ULy;
    n1.BLx = ULx; n1.BLy = ULy + halfY;
    n1.BRx = ULx + halfX; n1.BRy = ULy + halfY;
    n1.vKeys.reserve(vKeys.size());

    n2.ULx = n1.URx; n2.ULy = n1.URy;
    n2.URx = URx; n2.URy = URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy;
    n2.BRx = URx; n2.BRy = ULy + halfY;
    n2.vKeys.reserve(vKeys.size());

    n3.ULx = n1.BLx; n3.ULy = n1.BLy;
    n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = BLx; n3.BLy = BLy;
    n3.BRx = n1.BRx; n3.BRy = BLy;
    n3.vKeys.reserve(vKeys.size());

    n4.ULx = n3.URx; n4.ULy = n3.URy;
    n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy;
    n4.BRx = BRx; n4.BRy = BRy;
    n4.vKeys.reserve(vKeys.size());

    for (size_t i = 0; i < vKeys.size(); i++) {
        const OrcKeyPoint &kp = vKeys[i];
        if (kp.x < n1.URx) {
            if (kp.y < n1.BRy)
                n1.vKeys.push_back(kp);
            else
                n3.vKeys.push_back(kp);
        } else if (kp.y < n1.BRy)
            n2.vKeys.push_back(kp);
        else
            n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

/* reference src/ORBextractor.cpp:489-718.
 * INDETERMINACY RESOLVED HERE: the reference sorts pair<int, ExtractorNode*> (:638), so
 * equal-size nodes are ordered by heap address, which is allocator dependent.  The oracle (and the
 * device code) order ties by CREATION SEQUENCE instead: a later-created node counts as the "higher
 * address", i.e. among equal sizes the most recently created node is divided first. */
std::vector<OrcKeyPoint> DistributeOctTree(const std::vector<OrcKeyPoint> &vToDistributeKeys, int minX, int maxX,
                                           int minY, int maxY, int N) {
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    std::vector<OrcKeyPoint> vResultKeys;
    if (nIni < 1) return vResultKeys; /* reference divides by zero here (UB); we return nothing */
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<ExtractorNode> lNodes;
    std::vector<ExtractorNode *> vpIniNodes(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        ExtractorNode ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.vKeys.reserve(vToDistributeKeys.size());
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const OrcKeyPoint &kp = vToDistributeKeys[i];
        vpIniNodes[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
    }
    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) {
            lit->bNoMore = true;
            lit++;
        } else if (lit->vKeys.empty())
            lit = lNodes.erase(lit);
        else
            lit++;
    }
    bool bFinish = false;
    typedef std::pair<int, ExtractorNode *> SzNode;
    auto lessSzNode = [](const SzNode &a, const SzNode &b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second->seq < b.second->seq;
    };
    std::vector<SzNode> vSizeAndPointerToNode;
    vSizeAndPointerToNode.reserve(lNodes.size() * 4);
    auto push_child = [&](ExtractorNode &n, bool countExpand, int &nToExpand) {
        if (n.vKeys.size() > 0) {
            n.seq = seq++;
            lNodes.push_front(n);
            if (n.vKeys.size() > 1) {
                if (countExpand) nToExpand++;
                vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
                lNodes.front().lit = lNodes.begin();
            }
        }
    };
    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) {
                lit++;
                continue;
            } else {
                ExtractorNode n1, n2, n3, n4;
                lit->DivideNode(n1, n2, n3, n4);
                push_child(n1, true, nToExpand);
                push_child(n2, true, nToExpand);
                push_child(n3, true, nToExpand);
                push_child(n4, true, nToExpand);
                lit = lNodes.erase(lit);
                continue;
            }
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SzNode> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end(), lessSzNode);
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    ExtractorNode n1, n2, n3, n4;
                    vPrev[j].second->DivideNode(n1, n2, n3, n4);
                    int dummy = 0;
                    push_child(n1, false, dummy);
                    push_child(n2, false, dummy);
                    push_child(n3, false, dummy);
                    push_child(n4, false, dummy);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    vResultKeys.reserve(N + 4);
    for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
        std::vector<OrcKeyPoint> &vNodeKeys = it->vKeys;
        OrcKeyPoint *pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++) {
            if (vNodeKeys[k].response > maxResponse) {
                pKP = &vNodeKeys[k];
                maxResponse = vNodeKeys[k].response;
            }
        }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

struct OrbOracle {
    int nfeatures;
    double scaleFactor; /* include/ORBextractor.h:84: double member initialised from a float */
    int nlevels, iniThFAST, minThFAST;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<Image> pyr, blur;
    std::vector<std::vector<OrcKeyPoint>> cand;
    std::vector<int> levelCount;

    /* reference src/ORBextractor.cpp:360-420 */
    OrbOracle(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
        : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
          minThFAST(_minThFAST) {
        mvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f;
        mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesiredFeaturesPerScale =
            nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesiredFeaturesPerScale *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        pyr.resize(nlevels);
        blur.resize(nlevels);
        cand.resize(nlevels);
        levelCount.assign(nlevels, 0);
    }

    void levelSize(int w, int h, int level, int &lw, int &lh) const {
        float scale = mvInvScaleFactor[level];
        lw = cvRound((float)w * scale);
        lh = cvRound((float)h * scale);
    }

    /* reference :1071-1096.  The 19-px REFLECT_101 apron of mvImagePyramid is not materialised: no
     * later stage reads it (FAST stays inside [16,cols-16), IC_Angle/rBRIEF stay within 15/19 px of a
     * keypoint that is >= 19 px from the edge, and the blur runs on an apron-less clone). */
    void ComputePyramid(const uint8_t *img, int w, int h, int stride) {
        for (int level = 0; level < nlevels; ++level) {
            int lw, lh;
            levelSize(w, h, level, lw, lh);
            pyr[level].create(lw, lh);
            if (level != 0) {
                resize_linear_u8(pyr[level - 1].d.data(), pyr[level - 1].w, pyr[level - 1].h,
                                 pyr[level - 1].w, pyr[level].d.data(), lw, lh, lw);
            } else {
                for (int y = 0; y < h; y++) memcpy(pyr[0].row(y), img + (size_t)y * stride, w);
            }
        }
    }

    /* reference :27-54 */
    float IC_Angle(const Image &image, float ptx, float pty) const {
        int m_01 = 0, m_10 = 0;
        const int step = image.w;
        const uint8_t *center = image.row(cvRound(pty)) + cvRound(ptx);
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fast_atan2((float)m_01, (float)m_10);
    }

    /* reference :720-811 */
    void ComputeKeyPointsOctTree(std::vector<std::vector<OrcKeyPoint>> &allKeypoints) {
        allKeypoints.resize(nlevels);
        const float W = 30;
        for (int level = 0; level < nlevels; ++level) {
            const int minBorderX = EDGE_THRESHOLD - 3;
            const int minBorderY = minBorderX;
            const int maxBorderX = pyr[level].w - EDGE_THRESHOLD + 3;
            const int maxBorderY = pyr[level].h - EDGE_THRESHOLD + 3;
            std::vector<OrcKeyPoint> &vToDistributeKeys = cand[level];
            vToDistributeKeys.clear();
            vToDistributeKeys.reserve(nfeatures * 10);
            const float width = (float)(maxBorderX - minBorderX);
            const float height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W);
            const int nRows = (int)(height / W);
            allKeypoints[level].clear();
            if (nCols < 1 || nRows < 1) continue; /* reference would divide by zero (level too small) */
            const int wCell = (int)std::ceil(width / nCols);
            const int hCell = (int)std::ceil(height / nRows);
            std::vector<OrcKeyPoint> vKeysCell;
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    const uint8_t *roi = pyr[level].row((int)iniY) + (int)iniX;
                    int rw = (int)maxX - (int)iniX, rh = (int)maxY - (int)iniY;
                    fast9_16(roi, rw, rh, pyr[level].w, iniThFAST, true, vKeysCell);
                    if (vKeysCell.empty()) fast9_16(roi, rw, rh, pyr[level].w, minThFAST, true, vKeysCell);
                    for (auto &kp : vKeysCell) {
                        kp.x += j * wCell;
                        kp.y += i * hCell;
                        vToDistributeKeys.push_back(kp);
                    }
                }
            }
            std::vector<OrcKeyPoint> &keypoints = allKeypoints[level];
            keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                          mnFeaturesPerLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (auto &kp : keypoints) {
                kp.x += minBorderX;
                kp.y += minBorderY;
                kp.octave = level;
                kp.size = (float)scaledPatchSize;
            }
        }
        for (int level = 0; level < nlevels; ++level)
            for (auto &kp : allKeypoints[level]) kp.angle = IC_Angle(pyr[level], kp.x, kp.y);
    }

    /* reference :58-97 */
    void computeOrbDescriptor(const OrcKeyPoint &kpt, const Image &img, uint8_t *desc, int trig_mode) const {
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a, b;
        if (trig_mode == ORC_TRIG_LIBM) {
            a = (float)cosf(angle);
            b = (float)sinf(angle);
        } else
            sincos_det(angle, &a, &b);
        const int step = img.w;
        const uint8_t *center = img.row(cvRound(kpt.y)) + cvRound(kpt.x);
        const int *pattern = bit_pattern_31_;
        auto GET_VALUE = [&](int idx) -> int {
            int px = pattern[idx * 2], py = pattern[idx * 2 + 1];
            return center[cvRound(px * b + py * a) * step + cvRound(px * a - py * b)];
        };
        for (int i = 0; i < 32; ++i, pattern += 32) {
            int val = 0;
            for (int k = 0; k < 8; k++) {
                int t0 = GET_VALUE(2 * k), t1 = GET_VALUE(2 * k + 1);
                val |= (t0 < t1) << k;
            }
            desc[i] = (uint8_t)val;
        }
    }

    /* reference :1001-1065 */
    int extract(const uint8_t *img, int w, int h, int stride, OrcKeyPoint *kps, uint8_t *desc, int cap,
                int32_t *n_out, int trig_mode) {
        *n_out = 0;
        if (!img || w <= 0 || h <= 0) return 0; /* :1004-1005 silent return */
        ComputePyramid(img, w, h, stride);
        std::vector<std::vector<OrcKeyPoint>> allKeypoints;
        ComputeKeyPointsOctTree(allKeypoints);
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) {
            levelCount[level] = (int)allKeypoints[level].size();
            nkeypoints += levelCount[level];
        }
        if (nkeypoints > cap) return -2;
        int offset = 0;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<OrcKeyPoint> &keypoints = allKeypoints[level];
            int nl = (int)keypoints.size();
            blur[level].create(pyr[level].w, pyr[level].h);
            /* the reference skips the blur for empty levels (:1041-1042); computing it anyway only
             * fills the inspection buffer */
            gaussian_blur7_u8(pyr[level].d.data(), pyr[level].w, pyr[level].h, pyr[level].w,
                              blur[level].d.data(), pyr[level].w);
            if (nl == 0) continue;
            for (int i = 0; i < nl; i++)
                computeOrbDescriptor(keypoints[i], blur[level], desc + (size_t)(offset + i) * 32, trig_mode);
            if (level != 0) {
                float scale = mvScaleFactor[level];
                for (auto &kp : keypoints) {
                    kp.x *= scale;
                    kp.y *= scale;
                }
            }
            for (int i = 0; i < nl; i++) kps[offset + i] = keypoints[i];
            offset += nl;
        }
        *n_out = nkeypoints;
        return 0;
    }
};

} // namespace

extern "C" {

void *orc_orb_create(const OrcOrbCfg *c) {
    return new OrbOracle(c->nfeatures, c->scaleFactor, c->nlevels, c->iniThFAST, c->minThFAST);
}
void orc_orb_destroy(void *h) { delete (OrbOracle *)h; }
int orc_orb_tables(void *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                   int32_t *fpl, int32_t *umax16) {
    OrbOracle *o = (OrbOracle *)h;
    for (int i = 0; i < o->nlevels; i++) {
        scale[i] = o->mvScaleFactor[i];
        inv_scale[i] = o->mvInvScaleFactor[i];
        sigma2[i] = o->mvLevelSigma2[i];
        inv_sigma2[i] = o->mvInvLevelSigma2[i];
        fpl[i] = o->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = o->umax[i];
    return 0;
}
int orc_orb_level_sizes(void *h, int w, int hgt, int32_t *ws, int32_t *hs) {
    OrbOracle *o = (OrbOracle *)h;
    for (int l = 0; l < o->nlevels; l++) {
        int lw, lh;
        o->levelSize(w, hgt, l, lw, lh);
        ws[l] = lw;
        hs[l] = lh;
    }
    return 0;
}
int orc_orb_extract(void *h, const uint8_t *img, int w, int hgt, int stride, OrcKeyPoint *kps, uint8_t *desc,
                    int cap, int32_t *n_out, int trig_mode) {
    return ((OrbOracle *)h)->extract(img, w, hgt, stride, kps, desc, cap, n_out, trig_mode);
}
int orc_orb_get_level(void *h, int level, uint8_t *out) {
    OrbOracle *o = (OrbOracle *)h;
    memcpy(out, o->pyr[level].d.data(), o->pyr[level].d.size());
    return 0;
}
int orc_orb_get_blur(void *h, int level, uint8_t *out) {
    OrbOracle *o = (OrbOracle *)h;
    memcpy(out, o->blur[level].d.data(), o->blur[level].d.size());
    return 0;
}
int orc_orb_get_candidates(void *h, int level, OrcKeyPoint *out, int cap) {
    OrbOracle *o = (OrbOracle *)h;
    int n = (int)o->cand[level].size();
    for (int i = 0; i < n && i < cap; i++) out[i] = o->cand[level][i];
    return n;
}
int orc_orb_get_level_count(void *h, int level) { return ((OrbOracle *)h)->levelCount[level]; }

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh,
                          int dstride) {
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orc_border_reflect101_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int border) {
    border_reflect101_u8(src, w, h, sstride, dst, border);
}
void orc_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride) {
    gaussian_blur7_u8(src, w, h, sstride, dst, dstride);
}
int orc_fast9_16(const uint8_t *roi, int w, int h, int stride, int threshold, int nonmax, OrcKeyPoint *out,
                 int cap) {
    std::vector<OrcKeyPoint> v;
    fast9_16(roi, w, h, stride, threshold, nonmax != 0, v);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}
int orc_fast_corner_score(const uint8_t *center, int stride, int threshold) {
    return fast_corner_score(center, stride, threshold);
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float orc_ic_angle(const uint8_t *img, int stride, int x, int y) {
    /* standalone IC_Angle with the canonical umax table */
    static const int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0, d = um[v];
        for (int u = -d; u <= d; ++u) {
            int vp = center[u + v * stride], vm = center[u - v * stride];
            v_sum += (vp - vm);
            m_10 += u * (vp + vm);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}
int orc_distribute_octtree(const OrcKeyPoint *in, int n, int minX, int maxX, int minY, int maxY, int N,
                           OrcKeyPoint *out, int cap) {
    std::vector<OrcKeyPoint> v(in, in + n);
    std::vector<OrcKeyPoint> r = DistributeOctTree(v, minX, maxX, minY, maxY, N);
    for (size_t i = 0; i < r.size() && (int)i < cap; i++) out[i] = r[i];
    return (int)r.size();
}
void orc_sincos_det(float angle_rad, float *c, float *s) { sincos_det(angle_rad, c, s); }

} /* extern "C" */
