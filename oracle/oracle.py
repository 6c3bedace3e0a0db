"""ctypes wrapper of the CPU ORACLE (oracle/liborbslam_oracle.so).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liborbslam_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrbCfg(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scaleFactor", C.c_float), ("nlevels", C.c_int32),
                ("iniThFAST", C.c_int32), ("minThFAST", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int32),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_orb_create.restype = C.c_void_p
        L.orc_orb_create.argtypes = [C.POINTER(OrbCfg)]
        L.orc_orb_destroy.argtypes = [C.c_void_p]
        L.orc_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_orb_level_sizes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_orb_get_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_orb_get_blur.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_orb_get_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_orb_get_level_count.argtypes = [C.c_void_p, C.c_int]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int]
        L.orc_border_reflect101_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_gaussian_blur7_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int]
        L.orc_fast_corner_score.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_void_p, C.c_int]
        L.orc_sincos_det.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hamming_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.orc_search_for_initialization.argtypes = [C.POINTER(FrameView), C.POINTER(FrameView), C.c_void_p,
                                                    C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int]
        L.orc_compute_three_maxima.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _register_ba(L)
        _lib = L
    return _lib


class BAProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32),
                ("poses", C.c_void_p), ("fixed", C.c_void_p), ("points", C.c_void_p), ("obs", C.c_void_p),
                ("cam_idx", C.c_void_p), ("pt_idx", C.c_void_p), ("inv_sigma2", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("huber_delta", C.c_double), ("n_rounds", C.c_int32), ("rounds", C.c_int32 * 4),
                ("chi2_threshold", C.c_double), ("drop_robust_after", C.c_int32), ("stop", C.c_void_p)]


class BAResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("points", C.c_void_p), ("outlier", C.c_void_p),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("iterations", C.c_int32),
                ("lin_solves", C.c_int32)]


def _register_ba(L):
    L.orc_ba_solve.argtypes = [C.POINTER(BAProblem), C.POINTER(BAResult)]
    L.orc_ba_solve.restype = C.c_int
    L.orc_pose_optimization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                        C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.orc_pose_optimization.restype = C.c_int


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


TRIG_LIBM, TRIG_DET = 0, 1


class OrbOracle:
    """CPU restatement of ORBSlam::ORBextractor (reference src/ORBextractor.cpp)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.cfg = OrbCfg(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.nlevels = nlevels
        self.h = lib().orc_orb_create(C.byref(self.cfg))
        self._wh = None

    def __del__(self):
        try:
            if self.h:
                lib().orc_orb_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        lib().orc_orb_tables(self.h, _p(sc), _p(isc), _p(s2), _p(is2), _p(fpl), _p(um))
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=um)

    def level_sizes(self, w, h):
        ws = np.zeros(self.nlevels, np.int32)
        hs = np.zeros(self.nlevels, np.int32)
        lib().orc_orb_level_sizes(self.h, w, h, _p(ws), _p(hs))
        return ws, hs

    def extract(self, img, trig=TRIG_DET, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        # per level the quadtree returns up to max(quota + 3, 4 * nIni) keypoints, nIni = round(W / H) roots of the
        # BORDERED level (W = width - 32, H = height - 32: a 1938 x 115 strip has 23, 26, 31 roots on its first levels)
        if cap is None:
            ws, hs = self.level_sizes(w, h)
            cap = self.cfg.nfeatures + 64 + sum(8 + 4 * (max(int(a) - 32, 0) // max(int(b) - 32, 1) + 2) for a, b in zip(ws, hs))
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        rc = lib().orc_orb_extract(self.h, _p(img), w, h, w, _p(kps), _p(desc), cap, C.byref(n), trig)
        if rc != 0:
            raise RuntimeError("oracle extract rc=%d" % rc)
        self._wh = (w, h)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def level(self, l):
        ws, hs = self.level_sizes(*self._wh)
        out = np.zeros((hs[l], ws[l]), np.uint8)
        lib().orc_orb_get_level(self.h, l, _p(out))
        return out

    def blurred(self, l):
        ws, hs = self.level_sizes(*self._wh)
        out = np.zeros((hs[l], ws[l]), np.uint8)
        lib().orc_orb_get_blur(self.h, l, _p(out))
        return out

    def candidates(self, l):
        n = lib().orc_orb_get_candidates(self.h, l, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        lib().orc_orb_get_candidates(self.h, l, _p(out), n)
        return out[:n]

    def level_count(self, l):
        return lib().orc_orb_get_level_count(self.h, l)


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), dw, dh, dw)
    return dst


def border_reflect101(src, border):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().orc_border_reflect101_u8(_p(src), w, h, w, _p(dst), border)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.shape[1], _p(dst), src.shape[1])
    return dst


def fast9_16(roi, threshold, nonmax=True):
    roi = np.ascontiguousarray(roi, np.uint8)
    h, w = roi.shape
    cap = w * h
    out = np.zeros(max(cap, 1), KP_DTYPE)
    n = lib().orc_fast9_16(_p(roi), w, h, w, threshold, int(nonmax), _p(out), cap)
    return out[:n]


def fast_corner_score(img, x, y, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    base = img.ctypes.data + y * img.shape[1] + x
    return lib().orc_fast_corner_score(C.c_void_p(base), img.shape[1], threshold)


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_ic_angle(_p(img), img.shape[1], x, y)


def distribute_octtree(kps, min_x, max_x, min_y, max_y, N):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros(max(N + 8, 8), KP_DTYPE)
    n = lib().orc_distribute_octtree(_p(kps), len(kps), min_x, max_x, min_y, max_y, N, _p(out), len(out))
    return out[:n]


def sincos_det(angle_rad):
    c, s = C.c_float(), C.c_float()
    lib().orc_sincos_det(float(angle_rad), C.byref(c), C.byref(s))
    return c.value, s.value


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def hamming_knn2(d1, d2):
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    n1 = len(d1)
    idx = np.zeros(n1, np.int32)
    best = np.zeros(n1, np.uint16)
    second = np.zeros(n1, np.uint16)
    lib().orc_hamming_knn2(_p(d1), n1, _p(d2), len(d2), _p(idx), _p(best), _p(second))
    return idx, best, second


def _view(kps, desc, bounds):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    v = FrameView(kps.ctypes.data, desc.ctypes.data, len(kps), *[float(b) for b in bounds])
    return v, (kps, desc)


def search_for_initialization(kps1, desc1, kps2, desc2, bounds, prev_matched=None, window=100, nnratio=0.9,
                              check_ori=True, clone_mode=0):
    """bounds = (min_x, max_x, min_y, max_y).  window < 0 -> brute-force variant (a12).
    Returns (nmatches, matches12[int32 N1], prev_matched[N1,2] float32 updated)."""
    v1, k1 = _view(kps1, desc1, bounds)
    v2, k2 = _view(kps2, desc2, bounds)
    if prev_matched is None:
        prev_matched = np.stack([k1[0]["x"], k1[0]["y"]], axis=1)
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.full(len(kps1), -1, np.int32)
    n = lib().orc_search_for_initialization(C.byref(v1), C.byref(v2), _p(pm), _p(m12), int(window),
                                            float(nnratio), int(check_ori), int(clone_mode))
    return n, m12, pm


def features_in_area(kps, bounds, queries_xyr, levels=None, cap=None):
    """Frame::GetFeaturesInArea for every query row (x, y, r); levels = (nq, 2) int32 (minLevel, maxLevel) or None
    for (-1, -1).  Returns a list of int32 index arrays in the reference's order."""
    v, keep = _view(kps, np.zeros((len(kps), 32), np.uint8), bounds)
    q = np.ascontiguousarray(queries_xyr, np.float32).reshape(-1, 3)
    nq = len(q)
    cap = int(cap if cap is not None else max(len(kps), 1))
    lv = None if levels is None else np.ascontiguousarray(levels, np.int32).reshape(nq, 2)
    out = np.full((nq, cap), -1, np.int32)
    cnt = np.zeros(nq, np.int32)
    L = lib()
    L.orc_features_in_area.argtypes = [C.POINTER(FrameView), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_features_in_area.restype = None
    L.orc_features_in_area(C.byref(v), _p(q), _p(lv) if lv is not None else None, nq, _p(out), cap, _p(cnt))
    del keep
    return [out[i, :min(cnt[i], cap)].copy() for i in range(nq)], cnt


def compute_three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    a, b, c = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
    lib().orc_compute_three_maxima(_p(sizes), len(sizes), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


HUBER_2D = float(np.sqrt(5.99))  # src/Optimizer.cpp:61


def fill_ba_problem(P, w, rounds, robust=True, chi2_threshold=5.991, drop_robust_after=0, keep=None):
    """fill a BAProblem-shaped ctypes struct from a synth.make_ba_window() dict; returns the arrays that
    must stay alive"""
    arrs = dict(poses=np.ascontiguousarray(w["poses"], np.float64), fixed=np.ascontiguousarray(w["fixed"], np.uint8),
                points=np.ascontiguousarray(w["points"], np.float64), obs=np.ascontiguousarray(w["obs"], np.float64),
                cam_idx=np.ascontiguousarray(w["cam_idx"], np.int32), pt_idx=np.ascontiguousarray(w["pt_idx"], np.int32),
                inv_sigma2=np.ascontiguousarray(w["inv_sigma2"], np.float64))
    P.n_cams, P.n_points, P.n_obs = len(arrs["poses"]), len(arrs["points"]), len(arrs["obs"])
    for k, v in arrs.items():
        setattr(P, k, v.ctypes.data)
    P.fx, P.fy, P.cx, P.cy = w["fx"], w["fy"], w["cx"], w["cy"]
    P.huber_delta = HUBER_2D if robust else 0.0
    P.n_rounds = len(rounds)
    for i, r in enumerate(rounds):
        P.rounds[i] = int(r)
    P.chi2_threshold = chi2_threshold
    P.drop_robust_after = drop_robust_after
    P.stop = None
    return arrs


def ba_solve(w, rounds=(5, 10), robust=True, chi2_threshold=5.991, drop_robust_after=0):
    """oracle BA on a synth.make_ba_window() dict.  rounds=(5,10): local BA schedule (a14);
    rounds=(20,): global BA (a13)."""
    P, R = BAProblem(), BAResult()
    keep = fill_ba_problem(P, w, rounds, robust, chi2_threshold, drop_robust_after)
    poses = np.zeros((P.n_cams, 7), np.float64)
    points = np.zeros((P.n_points, 3), np.float64)
    outlier = np.zeros(P.n_obs, np.uint8)
    R.poses, R.points, R.outlier = poses.ctypes.data, points.ctypes.data, outlier.ctypes.data
    rc = lib().orc_ba_solve(C.byref(P), C.byref(R))
    assert rc == 0
    del keep
    return dict(poses=poses, points=points, outlier=outlier, chi2_initial=R.chi2_initial, chi2_final=R.chi2_final,
                iterations=R.iterations, lin_solves=R.lin_solves)


def pose_optimization(pose, xw, obs, inv_sigma2, fx, fy, cx, cy):
    pose = np.ascontiguousarray(pose, np.float64).copy()
    xw = np.ascontiguousarray(xw, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    inv_sigma2 = np.ascontiguousarray(inv_sigma2, np.float64)
    n = len(xw)
    outlier = np.zeros(max(n, 1), np.uint8)
    ninl = C.c_int32(0)
    lib().orc_pose_optimization(_p(pose), _p(xw), _p(obs), _p(inv_sigma2), n, fx, fy, cx, cy, _p(outlier),
                                C.byref(ninl))
    return pose, outlier[:n], ninl.value


def search_by_projection(cur_kps, cur_desc, last_kps, last_desc, bounds, last_flags, last_xw, Rcw, tcw, fx, fy, cx, cy,
                         scale_factors, cur_taken, th, check_ori=True):
    """SURVEY N4 (upstream semantics, parity unpinned): returns (nmatches, cur_match[n_cur], cur_taken[n_cur])"""
    L = lib()
    L.orc_search_by_projection.argtypes = [C.POINTER(FrameView), C.POINTER(FrameView), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    vc, kc = _view(cur_kps, cur_desc, bounds)
    vl, kl = _view(last_kps, last_desc, bounds)
    fl = np.ascontiguousarray(last_flags, np.uint8)
    xw = np.ascontiguousarray(last_xw, np.float64)
    R = np.ascontiguousarray(Rcw, np.float64).reshape(9)
    t = np.ascontiguousarray(tcw, np.float64).reshape(3)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    taken = np.ascontiguousarray(cur_taken, np.uint8).copy()
    m = np.full(max(len(cur_kps), 1), -1, np.int32)
    n = L.orc_search_by_projection(C.byref(vc), C.byref(vl), _p(fl), _p(xw), _p(R), _p(t), fx, fy, cx, cy, _p(sf), len(sf),
                                   _p(taken), float(th), 1 if check_ori else 0, _p(m))
    return n, m[:len(cur_kps)], taken


def search_by_bow(kf_kps, kf_desc, kf_valid, f_kps, f_desc, bounds, nnratio=0.7, check_ori=True):
    """SURVEY N4 (one-node vocabulary, parity unpinned): returns (nmatches, f_match[n_f])"""
    L = lib()
    L.orc_search_by_bow.argtypes = [C.POINTER(FrameView), C.c_void_p, C.POINTER(FrameView), C.c_float, C.c_int, C.c_void_p]
    vk, kk = _view(kf_kps, kf_desc, bounds)
    vf, kf_ = _view(f_kps, f_desc, bounds)
    valid = np.ascontiguousarray(kf_valid, np.uint8)
    m = np.full(max(len(f_kps), 1), -1, np.int32)
    n = L.orc_search_by_bow(C.byref(vk), _p(valid), C.byref(vf), float(nnratio), 1 if check_ori else 0, _p(m))
    return n, m[:len(f_kps)]
