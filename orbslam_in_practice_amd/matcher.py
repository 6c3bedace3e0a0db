"""Host-side mirror of ORBSlam::ORBmatcher (reference include/ORBmatcher.h:8-32) over the C-ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbFrameView, check, ptr


class FrameFeatures:
    """what ORBmatcher reads from a Frame: GetUnKeyPts(), GetDescriptors() and the image bounds
    (Frame::miMinX.., src/Frame.cpp:111-141)."""

    def __init__(self, keypoints, descriptors, bounds):
        self.kps = np.ascontiguousarray(keypoints, KP_DTYPE)
        self.desc = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        assert len(self.kps) == len(self.desc)
        self.bounds = tuple(float(b) for b in bounds)  # (min_x, max_x, min_y, max_y)

    def view(self):
        return OrbFrameView(self.kps.ctypes.data, self.desc.ctypes.data, len(self.kps), *self.bounds)


class ORBmatcher:
    HISTO_LENGTH = 30  # src/ORBmatcher.cpp:6
    TH_LOW = 50        # src/ORBmatcher.cpp:7

    def __init__(self, nnratio=0.6, checkOri=True, max_keypoints=4096, max_pairs=1, device=0):
        self._L = _lib.lib()
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._m = C.c_void_p()
        check(self._L.orb_matcher_create(max_keypoints, max_pairs, device, C.byref(self._m)))

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            self._L.orb_matcher_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize):
        """src/ORBmatcher.cpp:9-126.  vbPrevMatched: [N1,2] float32, updated in place.
        Returns (nmatches, vnMatches12[int32 N1]).  windowSize < 0 = brute-force variant."""
        n1 = len(F1.kps)
        m12 = np.full(n1, -1, np.int32)
        nm = C.c_int32(0)
        v1, v2 = F1.view(), F2.view()
        pm = None
        if vbPrevMatched is not None:
            assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.shape == (n1, 2) \
                and vbPrevMatched.flags.c_contiguous
            pm = vbPrevMatched
        check(self._L.orb_match_init(self._m, C.byref(v1), C.byref(v2), ptr(pm), ptr(m12), int(windowSize),
                                     self.mfNNratio, int(self.mbCheckOrientation), C.byref(nm)))
        return nm.value, m12

    def GetFeaturesInArea(self, F, queries_xyr, levels=None, cap=None):
        """Frame::GetFeaturesInArea (src/Frame.cpp:219-271, grid of :144-168) for every query row (x, y, r) on the
        device (orb_features_in_area).  levels: (nq, 2) int32 (minLevel, maxLevel) or None = (-1, -1).
        Returns (list of int32 index arrays in the reference's order, counts)."""
        q = np.ascontiguousarray(queries_xyr, np.float32).reshape(-1, 3)
        nq = len(q)
        cap = int(cap if cap is not None else max(len(F.kps), 1))
        lv = None if levels is None else np.ascontiguousarray(levels, np.int32).reshape(nq, 2)
        out = np.full((max(nq, 1), cap), -1, np.int32)
        cnt = np.zeros(max(nq, 1), np.int32)
        v = F.view()
        check(self._L.orb_features_in_area(self._m, C.byref(v), ptr(q) if nq else None, ptr(lv) if lv is not None else None,
                                           nq, ptr(out), cap, ptr(cnt)))
        return [out[i, :min(cnt[i], cap)].copy() for i in range(nq)], cnt[:nq]

    def DescriptorDistance(self, a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return self._L.orb_descriptor_distance(ptr(a), ptr(b))

    def ComputeThreeMaxima(self, histo_sizes):
        s = np.ascontiguousarray(histo_sizes, np.int32)
        a, b, c = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        self._L.orb_compute_three_maxima(ptr(s), len(s), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def knn2(self, d1, d2):
        """brute-force Hamming 2-NN (orb_hamming_knn2)"""
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        n1 = len(d1)
        idx = np.zeros(n1, np.int32)
        best = np.zeros(n1, np.uint16)
        second = np.zeros(n1, np.uint16)
        check(self._L.orb_hamming_knn2(self._m, ptr(d1), n1, ptr(d2), len(d2), ptr(idx), ptr(best), ptr(second)))
        return idx, best, second

    def match_batch_device(self, d_kps, d_desc, d_counts, cap, d_pair_a, d_pair_b, n_pairs, bounds, windowSize,
                           d_matches12, d_nmatches, stream=None):
        check(self._L.orb_match_batch_device(self._m, d_kps, d_desc, d_counts, cap, d_pair_a, d_pair_b, n_pairs,
                                             bounds[0], bounds[1], bounds[2], bounds[3], int(windowSize),
                                             self.mfNNratio, int(self.mbCheckOrientation), d_matches12, d_nmatches,
                                             stream))

    # ---- SURVEY N4 (empty bodies in the reference; upstream semantics, parity unpinned by construction) ----
    def SearchByProjection(self, cur, last, last_flags, last_xw, Rcw, tcw, fx, fy, cx, cy, scale_factors, cur_taken, th):
        """include/ORBmatcher.h:24.  cur / last: FrameFeatures; last_flags bit0 = holds a map point and is not an
        outlier, bit1 = the point has observations; returns (nmatches, cur_match[n_cur], cur_taken[n_cur])."""
        fl = np.ascontiguousarray(last_flags, np.uint8)
        xw = np.ascontiguousarray(last_xw, np.float64)
        R = np.ascontiguousarray(Rcw, np.float64).reshape(9)
        t = np.ascontiguousarray(tcw, np.float64).reshape(3)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        taken = np.ascontiguousarray(cur_taken, np.uint8).copy()
        m = np.full(max(len(cur.kps), 1), -1, np.int32)
        n = C.c_int32(0)
        vc, vl = cur.view(), last.view()
        check(self._L.orb_search_by_projection(self._m, C.byref(vc), C.byref(vl), ptr(fl), ptr(xw), ptr(R), ptr(t), fx, fy, cx, cy,
                                               ptr(sf), len(sf), ptr(taken), float(th), int(self.mbCheckOrientation), ptr(m),
                                               C.byref(n)))
        return n.value, m[:len(cur.kps)], taken

    def search_by_projection_device(self, d_cur_kps, d_cur_desc, n_cur, d_last_kps, d_last_desc, n_last, d_last_flags, d_last_xw,
                                    Rcw, tcw, fx, fy, cx, cy, scale_factors, bounds, d_cur_taken, th, d_cur_match, d_n_matches,
                                    stream=None, d_n_cur=None):
        """orb_search_by_projection_device: every d_* argument is a device pointer (int); enqueued on `stream`, returns
        at once -- projection, candidate scan, walk and rotation histogram on the device, no copy and no wait"""
        R = np.ascontiguousarray(Rcw, np.float64).reshape(9)
        t = np.ascontiguousarray(tcw, np.float64).reshape(3)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        check(self._L.orb_search_by_projection_device(self._m, d_cur_kps, d_cur_desc, int(n_cur), d_n_cur, d_last_kps, d_last_desc, int(n_last),
                                                      d_last_flags, d_last_xw, ptr(R), ptr(t), fx, fy, cx, cy, ptr(sf), len(sf),
                                                      bounds[0], bounds[1], bounds[2], bounds[3], d_cur_taken, float(th),
                                                      int(self.mbCheckOrientation), d_cur_match, d_n_matches, stream))

    def SearchByBoW(self, kf, kf_valid, frame):
        """include/ORBmatcher.h:22 with the reference's (absent) vocabulary = one node.  Returns (nmatches, f_match[n_f])."""
        valid = np.ascontiguousarray(kf_valid, np.uint8)
        m = np.full(max(len(frame.kps), 1), -1, np.int32)
        n = C.c_int32(0)
        vk, vf = kf.view(), frame.view()
        check(self._L.orb_search_by_bow(self._m, C.byref(vk), ptr(valid), C.byref(vf), self.mfNNratio,
                                        int(self.mbCheckOrientation), ptr(m), C.byref(n)))
        return n.value, m[:len(frame.kps)]
