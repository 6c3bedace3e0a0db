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


/* ---- ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) ----
 * The reference declares it with an EMPTY body (include/ORBmatcher.h:24) and calls it from
 * Tracking::TrackWithMotionModel (src/Tracking.cpp:344,348): there is nothing in the reference to be identical to, so
 * this restates upstream ORB-SLAM2's monocular semantics on the reference's data model (PARITY UNPINNED by
 * construction): every last-frame feature that holds a map point and is not an outlier is projected with the current
 * pose (x_c = Rcw x_w + tcw; rejected behind the camera or outside the image bounds), candidates =
 * CurrentFrame.GetFeaturesInArea(u, v, th * scaleFactor[octave], octave - 1, octave + 1) (src/Frame.cpp:219-271) that do
 * not already hold a map point with observations, best Hamming distance over them (first minimum in candidate order)
 * accepted if <= TH_HIGH = 100; the accepted feature takes the map point (and, if that point has observations, is
 * closed to later features); rotation-consistency histogram as in the reference's own SearchForInitialization
 * (30 bins, factor 30/360, three maxima, src/ORBmatcher.cpp:79-118).  Deviations from upstream, all forced by the
 * reference's data model: the map point's descriptor is the last frame's descriptor row (MapPoint has no
 * GetDescriptor), and the histogram factor is the reference's (upstream's 1/HISTO_LENGTH is a known slip).
 * last_flags[i]: bit0 = holds a map point and is not an outlier, bit1 = that map point has observations.
 * cur_taken[j] in/out: feature j holds a map point with observations.  cur_match[j] out: index into the last frame. */
int orc_search_by_projection(const OrcFrameView *cur, const OrcFrameView *last, const uint8_t *last_flags, const double *last_xw,
                             const double *Rcw, const double *tcw, double fx, double fy, double cx, double cy,
                             const float *scale_factors, int nlevels, uint8_t *cur_taken, float th, int check_ori,
                             int32_t *cur_match) {
    const int TH_HIGH = 100;
    int nmatches = 0;
    for (int j = 0; j < cur->n; j++) cur_match[j] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    FrameGrid g(cur);
    for (int i = 0; i < last->n; i++) {
        if (!(last_flags[i] & 1)) continue;
        const double *X = last_xw + (size_t)i * 3;
        const double xc = Rcw[0] * X[0] + Rcw[1] * X[1] + Rcw[2] * X[2] + tcw[0];
        const double yc = Rcw[3] * X[0] + Rcw[4] * X[1] + Rcw[5] * X[2] + tcw[1];
        const double zc = Rcw[6] * X[0] + Rcw[7] * X[1] + Rcw[8] * X[2] + tcw[2];
        const double invzc = 1.0 / zc;
        if (invzc < 0) continue;
        const float u = (float)(fx * xc * invzc + cx), v = (float)(fy * yc * invzc + cy);
        if (u < cur->min_x || u > cur->max_x) continue;
        if (v < cur->min_y || v > cur->max_y) continue;
        const int nLastOctave = last->kps[i].octave;
        if (nLastOctave < 0 || nLastOctave >= nlevels) continue;
        const float radius = th * scale_factors[nLastOctave];
        const std::vector<size_t> vIndices2 = g.GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = last->desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const size_t i2 = vIndices2[k];
            if (cur_taken[i2]) continue;
            const int dist = DescriptorDistance(dMP, cur->desc + i2 * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx2 = (int)i2;
            }
        }
        if (bestDist <= TH_HIGH) {
            cur_match[bestIdx2] = i;
            if (last_flags[i] & 2) cur_taken[bestIdx2] = 1;
            nmatches++;
            if (check_ori) {
                float rot = last->kps[i].angle - cur->kps[bestIdx2].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int b = 0; b < HISTO_LENGTH; b++) sizes[b] = (int32_t)rotHist[b].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (size_t k = 0; k < rotHist[b].size(); k++) {
                cur_match[rotHist[b][k]] = -1; /* upstream: CurrentFrame.mvpMapPoints[..] = NULL -- the feature is free again */
                cur_taken[rotHist[b][k]] = 0;
                nmatches--;
            }
        }
    }
    return nmatches;
}

/* ---- ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame F, vector<MapPoint*>& vpMatches) ----
 * Empty body in the reference (include/ORBmatcher.h:22), called from Tracking::TrackReferenceKeyFrame
 * (src/Tracking.cpp:298).  Upstream walks the two BoW feature vectors and compares only features that fall into the
 * same vocabulary node; the reference has no vocabulary at all (KeyFrame::ComputeBoW and Frame::ComputeBOW are
 * empty, src/KeyFrame.cpp:19, src/Frame.cpp:274), i.e. ONE node holding every feature -- which is what this restates
 * (PARITY UNPINNED by construction): for every keyframe feature with a good map point, in index order, best and second
 * best distance over the frame features not matched yet, accepted if best <= TH_LOW and best < ratio * second,
 * rotation histogram as above.  kf_valid[i]: keyframe feature i holds a good map point.  f_match[j] out: keyframe
 * feature whose map point frame feature j receives, or -1. */
int orc_search_by_bow(const OrcFrameView *kf, const uint8_t *kf_valid, const OrcFrameView *f, float nnratio, int check_ori,
                      int32_t *f_match) {
    int nmatches = 0;
    for (int j = 0; j < f->n; j++) f_match[j] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    for (int i = 0; i < kf->n; i++) {
        if (!kf_valid[i]) continue;
        const uint8_t *dKF = kf->desc + (size_t)i * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int j = 0; j < f->n; j++) {
            if (f_match[j] >= 0) continue;
            const int dist = DescriptorDistance(dKF, f->desc + (size_t)j * 32);
            if (dist < bestDist1) {
                bestDist2 = bestDist1;
                bestDist1 = dist;
                bestIdxF = j;
            } else if (dist < bestDist2)
                bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
            f_match[bestIdxF] = i;
            nmatches++;
            if (check_ori) {
                float rot = kf->kps[i].angle - f->kps[bestIdxF].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdxF);
            }
        }
    }
    if (check_ori) {
        int32_t sizes[HISTO_LENGTH];
        for (int b = 0; b < HISTO_LENGTH; b++) sizes[b] = (int32_t)rotHist[b].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (size_t k = 0; k < rotHist[b].size(); k++) {
                f_match[rotHist[b][k]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

} /* extern "C" */
