"""Host-side mirror of ORBSlam::Optimizer (reference include/Optimizer.h:8-20) over the C-ABI.

The reference's static methods walk KeyFrame / MapPoint objects (src/Optimizer.cpp:43-122); here the
same graph is handed over as flat arrays (the BAProblem of include/orbslam_amd.h) -- the C++ shim in
include/orbslam_amd/Optimizer.h does that flattening for real KeyFrame* / MapPoint* containers.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BAProblem, BAResult, check, ptr

HUBER_2D = float(np.sqrt(5.99))  # thHuber2D, src/Optimizer.cpp:61
CHI2_2D = 5.991


class Optimizer:
    """all-static in the reference; one instance here owns a device solver handle"""

    def __init__(self, max_cams=128, max_points=8192, max_obs=65536, max_batch=1, device=0):
        self._L = _lib.lib()
        self._s = C.c_void_p()
        self._jobs = []   # submitted, not yet waited-for batches, oldest first: mirrors the C side's queue slot by slot
        check(self._L.ba_create(max_cams, max_points, max_obs, max_batch, device, C.byref(self._s)))

    def close(self):
        if getattr(self, "_s", None) and self._s.value:
            self._L.ba_destroy(self._s)
            self._s = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _ARRAYS = (("poses", np.float64), ("fixed", np.uint8), ("points", np.float64), ("obs", np.float64),
               ("cam_idx", np.int32), ("pt_idx", np.int32), ("inv_sigma2", np.float64))

    @staticmethod
    def _fill(P, w, rounds, robust, chi2_threshold, drop_robust_after, stop):
        """fill the C struct of one window.  The filled struct is remembered ON the window dict (key "_abi"): a caller
        that submits the same window object again (a stream of batches over a fixed set of windows, a retry) pays one
        memmove instead of twenty ctypes attribute stores -- 7 ms per 256-window batch otherwise, a fifth of what the
        device needs for the batch.  The memo is void as soon as any array of the window was replaced."""
        key = (tuple(int(r) for r in rounds), bool(robust), float(chi2_threshold), int(drop_robust_after))
        memo = w.get("_abi") if stop is None else None
        if memo is not None and memo[0] == key and all(w[k] is memo[3][i] for i, (k, _) in enumerate(Optimizer._ARRAYS)):
            C.memmove(C.byref(P), C.byref(memo[1]), C.sizeof(BAProblem))
            return memo[2]
        keep = {k: np.ascontiguousarray(w[k], dt) for k, dt in Optimizer._ARRAYS}
        P.n_cams, P.n_points, P.n_obs = len(keep["poses"]), len(keep["points"]), len(keep["obs"])
        for k, v in keep.items():
            setattr(P, k, v.ctypes.data)
        P.fx, P.fy, P.cx, P.cy = w["fx"], w["fy"], w["cx"], w["cy"]
        P.huber_delta = HUBER_2D if robust else 0.0
        P.n_rounds = len(rounds)
        for i, r in enumerate(rounds):
            P.rounds[i] = int(r)
        P.chi2_threshold = chi2_threshold
        P.drop_robust_after = drop_robust_after
        P.stop = stop.ctypes.data if stop is not None else None
        if stop is None and isinstance(w, dict):
            snap = BAProblem()
            C.memmove(C.byref(snap), C.byref(P), C.sizeof(BAProblem))
            w["_abi"] = (key, snap, keep, tuple(w[k] for k, _ in Optimizer._ARRAYS))
        return keep

    def _prepare(self, windows, rounds, robust, chi2_threshold, drop_robust_after, stop):
        n = len(windows)
        Ps = (BAProblem * n)()
        Rs = (BAResult * n)()
        keep, outs = [], []
        for i, w in enumerate(windows):
            keep.append(self._fill(Ps[i], w, rounds, robust, chi2_threshold, drop_robust_after, stop))
        # one pooled result buffer per field (per-window arrays cost ~0.1 ms each in allocation + ctypes glue)
        oc = np.concatenate([[0], np.cumsum([Ps[i].n_cams for i in range(n)])]).astype(np.int64)
        op = np.concatenate([[0], np.cumsum([Ps[i].n_points for i in range(n)])]).astype(np.int64)
        oe = np.concatenate([[0], np.cumsum([max(Ps[i].n_obs, 1) for i in range(n)])]).astype(np.int64)
        Pb, Xb, Ob = np.empty((int(oc[-1]), 7)), np.empty((max(int(op[-1]), 1), 3)), np.zeros(int(oe[-1]), np.uint8)
        bp, bx, bo = Pb.ctypes.data, Xb.ctypes.data, Ob.ctypes.data
        for i in range(n):
            Rs[i].poses, Rs[i].points, Rs[i].outlier = bp + int(oc[i]) * 56, bx + int(op[i]) * 24, bo + int(oe[i])
            outs.append(dict(poses=Pb[oc[i]:oc[i + 1]], points=Xb[op[i]:op[i + 1]], outlier=Ob[oe[i]:oe[i + 1]]))
        return dict(n=n, Ps=Ps, Rs=Rs, keep=keep, outs=outs, bufs=(Pb, Xb, Ob))

    @staticmethod
    def _collect(job):
        Ps, Rs = job["Ps"], job["Rs"]
        for i, o in enumerate(job["outs"]):
            o["outlier"] = o["outlier"][:Ps[i].n_obs]
            o.update(chi2_initial=Rs[i].chi2_initial, chi2_final=Rs[i].chi2_final, iterations=Rs[i].iterations,
                     lin_solves=Rs[i].lin_solves)
        return job["outs"]

    def solve_batch(self, windows, rounds, robust=True, chi2_threshold=CHI2_2D, drop_robust_after=0, stop=None):
        if self._jobs:
            # ba_solve_batch would drain the submitted batches inside C and leave their result buffers unclaimed here
            raise RuntimeError("solve_batch with %d submitted batch(es) in flight: call wait() first" % len(self._jobs))
        job = self._prepare(windows, rounds, robust, chi2_threshold, drop_robust_after, stop)
        check(self._L.ba_solve_batch(self._s, job["Ps"], job["Rs"], job["n"]))
        return self._collect(job)

    def submit_batch(self, windows, rounds, robust=True, chi2_threshold=CHI2_2D, drop_robust_after=0):
        """ba_submit_batch: stage + enqueue, returns at once (three batches may be in flight); pair with wait()"""
        job = self._prepare(windows, rounds, robust, chi2_threshold, drop_robust_after, None)
        check(self._L.ba_submit_batch(self._s, job["Ps"], job["Rs"], job["n"]))
        self._jobs.append(job)   # keeps every buffer the C side points into alive until wait()

    def wait(self):
        """ba_wait: results of the OLDEST submitted batch.  The C side retires that batch whether or not it
        succeeded, so the job leaves this queue before the status is looked at: the two queues stay aligned."""
        if not self._jobs:
            raise RuntimeError("wait() without a submitted batch")
        rc = self._L.ba_wait(self._s)
        job = self._jobs.pop(0)
        check(rc)
        return self._collect(job)

    def set_profiling(self, on=True):
        check(self._L.ba_set_profiling(self._s, 1 if on else 0))

    def iteration_ms(self):
        """(average device time of an optimize() round in ms, rounds timed, workgroups per window -- 0: the batch ran as
        phase kernels) of the last finished batch"""
        ms, nl, team = C.c_float(0), C.c_int32(0), C.c_int32(0)
        check(self._L.ba_get_iteration_ms(self._s, C.byref(ms), C.byref(nl), C.byref(team)))
        return ms.value, nl.value, team.value

    # ---- the reference's entry points (include/Optimizer.h:11-19) ----
    def BundleAdjustment(self, window, nIterations=5, pbStopFlag=None, nLoopKF=0, bRobust=True):
        """src/Optimizer.cpp:23-129: one optimize(nIterations), robust kernel if bRobust"""
        return self.solve_batch([window], (nIterations,), robust=bRobust, stop=pbStopFlag)[0]

    def GlobalBundleAdjustemnt(self, window, nIterations=5, pbStopFlag=None, nLoopKF=0, bRobust=True):
        """src/Optimizer.cpp:14-21 (sic: the reference's spelling, called at src/Tracking.cpp:254)"""
        return self.BundleAdjustment(window, nIterations, pbStopFlag, nLoopKF, bRobust)

    def LocalBundleAdjustment(self, window, pbStopFlag=None):
        """a14 (absent from the reference, upstream ORB-SLAM2 schedule): 5 its -> drop chi2 > 5.991 or
        negative-depth edges and the robust kernel -> 10 its"""
        return self.solve_batch([window], (5, 10), robust=True, stop=pbStopFlag)[0]

    def PoseOptimization(self, pose, xw, obs, inv_sigma2, fx, fy, cx, cy):
        """src/Optimizer.cpp:131-261.  Returns (pose[7], outlier[n], n_inliers)."""
        pose = np.ascontiguousarray(pose, np.float64).copy()
        xw = np.ascontiguousarray(xw, np.float64)
        obs = np.ascontiguousarray(obs, np.float64)
        inv_sigma2 = np.ascontiguousarray(inv_sigma2, np.float64)
        n = len(xw)
        outlier = np.zeros(max(n, 1), np.uint8)
        ninl = C.c_int32(0)
        check(self._L.ba_pose_optimization(self._s, ptr(pose), ptr(xw), ptr(obs), ptr(inv_sigma2), n, fx, fy, cx, cy,
                                           ptr(outlier), C.byref(ninl)))
        return pose, outlier[:n], ninl.value


    def PoseOptimizationBatch(self, poses, xws, obss, infos, fx, fy, cx, cy):
        """several pose problems in one launch (ba_pose_optimization_batch).  poses [P,7]; xws/obss/infos lists
        of per-problem arrays.  Returns (poses, [outlier arrays], n_inliers[P])."""
        P = len(xws)
        poses = np.ascontiguousarray(poses, np.float64).reshape(P, 7).copy()
        offs = np.zeros(P + 1, np.int32)
        offs[1:] = np.cumsum([len(x) for x in xws])
        n = int(offs[-1])
        xw = np.ascontiguousarray(np.concatenate(xws) if n else np.zeros((0, 3)), np.float64)
        obs = np.ascontiguousarray(np.concatenate(obss) if n else np.zeros((0, 2)), np.float64)
        info = np.ascontiguousarray(np.concatenate(infos) if n else np.zeros(0), np.float64)
        outlier = np.zeros(max(n, 1), np.uint8)
        ninl = np.zeros(P, np.int32)
        check(self._L.ba_pose_optimization_batch(self._s, ptr(poses), ptr(xw), ptr(obs), ptr(info), ptr(offs), P, fx, fy,
                                                 cx, cy, ptr(outlier), ptr(ninl)))
        return poses, [outlier[offs[i]:offs[i + 1]] for i in range(P)], ninl


def solve_window(window, rounds=(5, 10), **kw):
    o = Optimizer()
    try:
        return o.solve_batch([window], rounds, **kw)[0]
    finally:
        o.close()
