/*
 * match_oracle.cpp -- CPU ORACLE for ORBmatcher (test infrastructure only).
 *
 * Follows reference src/ORBmatcher.cpp:6-188 and the Frame grid/window semantics that feed it
 * (src/Frame.cpp:144-168, :219-271).  All of that source is present in the reference, so this part
 * is a line-by-line restatement; the only deviations are the two literal Frame bugs listed below.
 *
 * DEVIATIONS: Frame::GetGridId uses miMaxY instead of miMinY (src/Frame.cpp:164) -- the
 * oracle uses the intended miMinY (upstream ORB-SLAM2 PosInGrid).  Frame::UndistortedKeyPoints writes
 * y <- x (src/Frame.cpp:106) -- irrelevant here: the oracle takes undistorted keypoints as input.
 */
#include "oracle.h"

#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

const int HISTO_LENGTH = 30; /* src/ORBmatcher.cpp:6 */
const int TH_LOW = 50;       /* :7 */
const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64; /* include/Frame.h:11-12 */

/* src/ORBmatcher.cpp:128-144 */
int DescriptorDistance(const uint8_t *a, const uint8_t *b) {
    int32_t pa[8], pb[8];
    memcpy(pa, a, 32);
    memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* src/ORBmatcher.cpp:147-188 (takes the bin sizes) */
void ComputeThreeMaxima(const int32_t *histo, const int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            ind3 = ind2;
            ind2 = ind1;
            ind1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            ind3 = ind2;
            ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
}

struct FrameGrid {
    const OrcFrameView *f;
    float invW, invH;
    std::vector<int> grid[FRAME_GRID_COLS][FRAME_GRID_ROWS];

    /* src/Frame.cpp:59-60 (inverse cell sizes), :144-168 (assignment) */
    explicit FrameGrid(const OrcFrameView *fv) : f(fv) {
        invW = static_cast<float>(FRAME_GRID_COLS) / (f->max_x - f->min_x);
        invH = static_cast<float>(FRAME_GRID_ROWS) / (f->max_y - f->min_y);
        for (int i = 0; i < f->n; i++) {
            const double x = f->kps[i].x, y = f->kps[i].y;
            int ix = (int)std::round((x - f->min_x) * invW);
            int iy = (int)std::round((y - f->min_y) * invH); /* reference bug: miMaxY (:164) */
            if (ix < 0 || ix >= FRAME_GRID_COLS || iy < 0 || iy >= FRAME_GRID_ROWS) continue;
            grid[ix][iy].push_back(i);
        }
    }

    /* src/Frame.cpp:219-271 */
    std::vector<size_t> GetFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel) const {
        std::vector<size_t> vIndices;
        vIndices.reserve(f->n);
        const int nMinCellX = std::max(0, (int)std::floor((x - f->min_x - r) * invW));
        if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
        const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)std::ceil((x - f->min_x + r) * invW));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)std::floor((y - f->min_y - r) * invH));
        if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
        const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)std::ceil((y - f->min_y + r) * invH));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<int> &vCell = grid[ix][iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const OrcKeyPoint &kpUn = f->kps[vCell[j]];
                    if (bCheckLevels) {
                        if (kpUn.octave < minLevel) continue;
                        if (maxLevel >= 0)
                            if (kpUn.octave > maxLevel) continue;
                    }
                    const float distx = kpUn.x - x, disty = kpUn.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    }
};

} // namespace

extern "C" {

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) { return DescriptorDistance(a, b); }

void orc_compute_three_maxima(const int32_t *histo, int L, int32_t *i1, int32_t *i2, int32_t *i3) {
    int a = -1, b = -1, c = -1;
    ComputeThreeMaxima(histo, L, a, b, c);
    *i1 = a;
    *i2 = b;
    *i3 = c;
}

/* a12 primitive (build-defined): best / second-best Hamming distance of every row of
 * d1 over all rows of d2, ties -> lowest index, second = second order statistic (duplicates count). */
void orc_hamming_knn2(const uint8_t *d1, int n1, const uint8_t *d2, int n2, int32_t *idx, uint16_t *best,
                      uint16_t *second) {
    for (int i = 0; i < n1; i++) {
        int b1 = INT_MAX, b2 = INT_MAX, bi = -1;
        for (int j = 0; j < n2; j++) {
            int d = DescriptorDistance(d1 + (size_t)i * 32, d2 + (size_t)j * 32);
            if (d < b1) {
                b2 = b1;
                b1 = d;
                bi = j;
            } else if (d < b2)
                b2 = d;
        }
        idx[i] = bi;
        best[i] = (uint16_t)(b1 == INT_MAX ? 0xFFFF : b1);
        second[i] = (uint16_t)(b2 == INT_MAX ? 0xFFFF : b2);
    }
}

/* src/ORBmatcher.cpp:9-126.  window_size < 0: the a12 "brute force" variant -- every i1 (all
 * octaves), candidates = every i2 in ascending order; everything else identical. */
/* Frame::AssignFeaturesToGrid + Frame::GetFeaturesInArea (src/Frame.cpp:144-168, 219-271) for nq queries
 * (x, y, r) with level filters; out_idx[q * cap + k] in the reference's order, out_count[q] = full count */
void orc_features_in_area(const OrcFrameView *F, const float *qxyr, const int32_t *qlevels, int nq, int32_t *out_idx, int cap,
                          int32_t *out_count) {
    FrameGrid g(F);
    for (int q = 0; q < nq; q++) {
        const int lo = qlevels ? qlevels[2 * q] : -1, hi = qlevels ? qlevels[2 * q + 1] : -1;
        const std::vector<size_t> v = g.GetFeaturesInArea(qxyr[3 * q], qxyr[3 * q + 1], qxyr[3 * q + 2], lo, hi);
        out_count[q] = (int32_t)v.size();
        for (size_t k = 0; k < v.size() && (int)k < cap; k++) out_idx[(size_t)q * cap + k] = (int32_t)v[k];
    }
}

int orc_search_for_initialization(const OrcFrameView *F1, const OrcFrameView *F2, float *prev_xy,
                                  int32_t *vnMatches12, int windowSize, float mfNNratio, int mbCheckOrientation,
                                  int clone_mode) {
    int nmatches = 0;
    for (int i = 0; i < F1->n; i++) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
    const float factor = HISTO_LENGTH / 360.0f;
    std::vector<int> vMatchedDistance(F2->n, INT_MAX);
    std::vector<int> vnMatches21(F2->n, -1);
    FrameGrid g2(F2);
    std::vector<uint8_t> cloneBuf;
    auto getDescriptors = [&](const OrcFrameView *F) -> const uint8_t * {
        if (!clone_mode) return F->desc;
        cloneBuf.assign(F->desc, F->desc + (size_t)F->n * 32); /* include/Frame.h:39-42 clone() */
        return cloneBuf.data();
    };
    std::vector<uint8_t> d1buf(32);
    for (size_t i1 = 0, iend1 = (size_t)F1->n; i1 < iend1; i1++) {
        const OrcKeyPoint &kp1 = F1->kps[i1];
        int level1 = kp1.octave;
        std::vector<size_t> vIndices2;
        if (windowSize >= 0) {
            if (level1 > 0) continue;
            vIndices2 = g2.GetFeaturesInArea(prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)windowSize, level1, level1);
        } else {
            vIndices2.resize(F2->n);
            for (int j = 0; j < F2->n; j++) vIndices2[j] = j;
        }
        if (vIndices2.empty()) continue;
        memcpy(d1buf.data(), getDescriptors(F1) + i1 * 32, 32);
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            size_t i2 = vIndices2[k];
            const uint8_t *d2 = getDescriptors(F2) + i2 * 32;
            int dist = DescriptorDistance(d1buf.data(), d2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = (int)i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * mfNNratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = (int)i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (mbCheckOrientation) {
                    float rot = F1->kps[i1].angle - F2->kps[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back((int)i1);
                }
            }
        }
    }
    if (mbCheckOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int32_t sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) {
                    vnMatches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (size_t i1 = 0, iend1 = (size_t)F1->n; i1 < iend1; i1++)
        if (vnMatches12[i1] >= 0) {
            prev_xy[2 * i1] = F2->kps[vnMatches12[i1]].x;
            prev_xy[2 * i1 + 1] = F2->kps[vnMatches12[i1]].y;
        }
    return nmatches;
}

} /* extern "C" */
