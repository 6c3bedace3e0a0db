"""Host-side mirror of ORBSlam::ORBextractor (reference include/ORBextractor.h:29-97) over the C-ABI.

Same constructor arguments, same call operator, same getters; all arithmetic runs in the HIP kernels
behind orb_extract*.  numpy arrays stand in for cv::Mat / std::vector<cv::KeyPoint> (KP_DTYPE is the
28-byte cv::KeyPoint POD).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbConfig, check, ptr


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1  # include/ORBextractor.h:33

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_w=1280, max_h=720,
                 max_batch=1, device=0):
        self._L = _lib.lib()
        self.cfg = OrbConfig(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        self._h = C.c_void_p()
        check(self._L.orb_create(C.byref(self.cfg), max_w, max_h, max_batch, device, C.byref(self._h)))
        self.nlevels = nlevels
        self.max_batch = max_batch
        self.cap = self._L.orb_max_keypoints(self._h)
        self._wh = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters, include/ORBextractor.h:47-69 ----
    def GetLevels(self):
        return self._L.orb_get_levels(self._h)

    def GetScaleFactor(self):
        return self._L.orb_get_scale_factor(self._h)

    def _tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        check(self._L.orb_get_tables(self._h, ptr(sc), ptr(isc), ptr(s2), ptr(is2), ptr(fpl), ptr(um)))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=um)

    def GetScaleFactors(self):
        return self._tables()["scale"]

    def GetInverseScaleFactors(self):
        return self._tables()["inv_scale"]

    def GetScaleSigmaSquares(self):
        return self._tables()["sigma2"]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()["inv_sigma2"]

    def level_sizes(self, w, h):
        ws = np.zeros(self.nlevels, np.int32)
        hs = np.zeros(self.nlevels, np.int32)
        check(self._L.orb_get_level_sizes(self._h, w, h, ptr(ws), ptr(hs)))
        return ws, hs

    # ---- operator(), src/ORBextractor.cpp:1001-1065 ----
    def __call__(self, image, mask=None):
        """returns (keypoints[KP_DTYPE], descriptors[N,32] u8).  `mask` is ignored like in the reference
        (include/ORBextractor.h:42); an empty image returns empty outputs (:1004-1005)."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (src/ORBextractor.cpp:1008)"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        for attempt in (0, 1):
            kps = np.zeros(self.cap, KP_DTYPE)
            desc = np.zeros((self.cap, 32), np.uint8)
            n = C.c_int32(0)
            rc = self._L.orb_extract(self._h, ptr(image), w, h, image.strides[0], ptr(kps), ptr(desc), self.cap,
                                     C.byref(n))
            if rc == _lib.ORB_E_CAPACITY and attempt == 0 and self._grow_cap():
                continue  # a frame shape whose bound (4 * nIni per level for wide frames) exceeds the creation size's
            check(rc)
            break
        self._wh = (w, h)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def _grow_cap(self):
        """orb_max_keypoints() follows the frame size the handle last saw"""
        cap = self._L.orb_max_keypoints(self._h)
        if cap > self.cap:
            self.cap = cap
            return True
        return False

    def extract_batch(self, images):
        """images: [B,H,W] u8 (host).  Returns list of (kps, desc)."""
        images = np.ascontiguousarray(images, np.uint8)
        B, h, w = images.shape
        ptrs = (C.c_void_p * B)(*[images[i].ctypes.data for i in range(B)])
        for attempt in (0, 1):
            kps = np.zeros((B, self.cap), KP_DTYPE)
            desc = np.zeros((B, self.cap, 32), np.uint8)
            n = np.zeros(B, np.int32)
            rc = self._L.orb_extract_batch(self._h, ptrs, B, w, h, w, ptr(kps), ptr(desc), self.cap, ptr(n))
            if rc == _lib.ORB_E_CAPACITY and attempt == 0 and self._grow_cap():
                continue
            check(rc)
            break
        self._wh = (w, h)
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(B)]

    def extract_batch_device(self, d_imgs_ptr, n_frames, w, h, stride, frame_stride, d_kps_ptr, d_desc_ptr, cap,
                             d_counts_ptr, stream=None):
        """everything device-resident; only enqueues (see orb_extract_batch_device)."""
        check(self._L.orb_extract_batch_device(self._h, d_imgs_ptr, n_frames, w, h, stride, frame_stride, d_kps_ptr,
                                               d_desc_ptr, cap, d_counts_ptr, stream))
        self._wh = (w, h)

    def synchronize(self):
        check(self._L.orb_synchronize(self._h))

    # ---- mvImagePyramid (public member, include/ORBextractor.h:71) and inspection helpers ----
    def image_pyramid_level(self, level, frame=0):
        """level with its 19-px REFLECT_101 apron, as the reference's mvImagePyramid backing Mat"""
        ws, hs = self.level_sizes(*self._wh)
        out = np.zeros((hs[level] + 38, ws[level] + 38), np.uint8)
        check(self._L.orb_download_level(self._h, frame, level, 0, ptr(out)))
        return out

    def blurred_level(self, level, frame=0):
        ws, hs = self.level_sizes(*self._wh)
        out = np.zeros((hs[level], ws[level]), np.uint8)
        check(self._L.orb_download_level(self._h, frame, level, 1, ptr(out)))
        return out

    def candidates(self, level, frame=0):
        n = C.c_int32(0)
        check(self._L.orb_download_candidates(self._h, frame, level, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        check(self._L.orb_download_candidates(self._h, frame, level, ptr(out), n.value, C.byref(n)))
        return out[:n.value]
