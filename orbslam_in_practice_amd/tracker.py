"""One frame of Tracking::TrackWithMotionModel (reference src/Tracking.cpp:330-372) as ONE device chain: ctypes mirror of
the OrbTracker entry points of include/orbslam_amd.h.  The reference's caller makes three calls per frame -- Frame(im,
&extractor), SearchByProjection, PoseOptimization -- each a host round trip; the tracker chains the same kernels on one
stream (the extracted features stay in HBM as the next call's "last frame") and synchronises once."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


class Tracker:
    def __init__(self, extractor, matcher, optimizer, device=0):
        self._L = _lib.lib()
        self._t = C.c_void_p()
        self._keep = (extractor, matcher, optimizer)   # the tracker borrows their handles
        check(self._L.orb_tracker_create(extractor._h, matcher._m, optimizer._s, device, C.byref(self._t)))
        self.cap = extractor.cap

    def close(self):
        if getattr(self, "_t", None) and self._t.value:
            self._L.orb_tracker_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self._L.orb_tracker_reset(self._t))

    def track(self, gray, prev_flags, prev_xw, pose_pred, fx, fy, cx, cy, bounds, th, check_ori=True, next_gray=None):
        """returns dict(kps, desc, match[n], outlier[n], pose[7], n_matches, n_inliers); prev_flags / prev_xw empty on the
        first frame.  next_gray: the frame the NEXT call will pass (the very same contiguous uint8 array), uploaded and
        extracted beside this frame's search and pose problem (orb_tracker_track_prefetch)"""
        img = np.ascontiguousarray(gray, np.uint8)
        h, w = img.shape
        fl = np.ascontiguousarray(prev_flags, np.uint8)
        xw = np.ascontiguousarray(prev_xw, np.float64).reshape(-1, 3)
        n_prev = len(fl)
        assert len(xw) == n_prev
        pose = np.ascontiguousarray(pose_pred, np.float64).reshape(7)
        cap = max(self.cap, int(self._L.orb_max_keypoints(self._keep[0]._h)))
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        match = np.full(cap, -1, np.int32)
        outl = np.zeros(cap, np.uint8)
        pose_out = np.zeros(7)
        n, nm, ninl = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        nxt = None
        if next_gray is not None:  # look-ahead: the caller keeps this array alive and unchanged until it passes it as `gray`
            nxt = next_gray
            assert nxt.dtype == np.uint8 and nxt.flags["C_CONTIGUOUS"] and nxt.shape == img.shape
        check(self._L.orb_tracker_track_prefetch(self._t, ptr(img), w, h, img.strides[0], ptr(fl) if n_prev else None,
                                                 ptr(xw) if n_prev else None, n_prev, ptr(pose), fx, fy, cx, cy, bounds[0], bounds[1],
                                                 bounds[2], bounds[3], float(th), int(check_ori), ptr(kps), ptr(desc), cap, C.byref(n),
                                                 ptr(match), ptr(outl), ptr(pose_out), C.byref(nm), C.byref(ninl),
                                                 ptr(nxt) if nxt is not None else None))
        k = n.value
        return dict(kps=kps[:k].copy(), desc=desc[:k].copy(), match=match[:k].copy(), outlier=outl[:k].copy(), pose=pose_out,
                    n_matches=nm.value, n_inliers=ninl.value)
