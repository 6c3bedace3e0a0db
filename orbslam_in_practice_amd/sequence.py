"""A Tracking + LocalMapping style loop over a synthetic mono sequence, built ONLY from the calls of the hot path
(BASELINE.json configs[4] asks for the full loop of the reference; Tracking / LocalMapping / Initializer themselves
are out of scope, SURVEY.md section 2, and the reference's versions cannot be built here).  It exists to exercise the
in-scope calls the way their callers chain them, one stream, one frame at a time:

    ORBextractor::operator()            every frame                     (src/Tracking.cpp:115-137)
    ORBmatcher::SearchByProjection      against the last frame          (src/Tracking.cpp:341-352, upstream semantics)
    Optimizer::PoseOptimization         on the matched map points       (src/Tracking.cpp:355)
    Optimizer::LocalBundleAdjustment    over the last keyframes         (src/LocalMapping.cpp:16-18)

What the out-of-scope callers would do is replaced by scene knowledge: the scene is a textured plane, so a keypoint
becomes a map point by intersecting its viewing ray with the plane (instead of the Initializer's / LocalMapping's
triangulation), the first frame's pose is given, and a frame predicts its pose with the constant-velocity model.

`backend` is the product (`GpuBackend`: the HIP path through the C-ABI) -- or, in tests/ only, the checker
(tests/sequence_oracle.py: the same five calls on the CPU restatement under oracle/).  Same loop, same arithmetic
around the calls: the two trajectories are compared by tests/test_gpu_sequence.py.
"""
import numpy as np


# ---- the scene: a textured plane Z = 0, a pinhole camera moving above it ----------------------------------------
def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[1 + i] = 0.25 * s
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def make_plane_sequence(seed, n_frames, w=640, h=480, fx=420.0, fy=420.0):
    """(frames u8 [n, h, w], poses [n, 7] = (qw qx qy qz tx ty tz) world -> camera, K = (fx, fy, cx, cy), metres per
    texel).  The camera looks down the +Z axis of a frame whose origin hovers 2.4 .. 2.7 m in front of the plane, drifts
    sideways, rolls and tilts a few degrees: consecutive frames overlap by > 90 %."""
    from . import synth
    rng = np.random.Generator(np.random.PCG64(seed))
    tw, th = 1792, 1344
    tex = synth.make_frame(seed + 17, tw, th, n_shapes=9000).astype(np.float32)
    mpt = 0.0042  # metres per texel
    cx, cy = w / 2 - 0.5, h / 2 - 0.5
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    rays = np.stack([(xx - cx) / fx, (yy - cy) / fy, np.ones_like(xx)], -1)  # camera frame
    frames = np.zeros((n_frames, h, w), np.uint8)
    poses = np.zeros((n_frames, 7))
    a0, a1, a2 = rng.uniform(0, 2 * np.pi, 3)
    for k in range(n_frames):
        s = k / max(n_frames - 1, 1)
        # camera centre in the world (plane Z = 0, camera on the -Z side looking along +Z)
        C = np.array([-0.9 + 1.8 * s + 0.05 * np.sin(7 * s + a0), 0.25 * np.sin(2.2 * np.pi * s + a1), -2.55 + 0.15 * np.cos(5 * s + a2)])
        roll, tx_, ty_ = 0.10 * np.sin(3.1 * s + a1), 0.05 * np.sin(4.3 * s + a2), 0.04 * np.cos(2.7 * s + a0)
        Rz = np.array([[np.cos(roll), -np.sin(roll), 0], [np.sin(roll), np.cos(roll), 0], [0, 0, 1]])
        Rx = np.array([[1, 0, 0], [0, np.cos(tx_), -np.sin(tx_)], [0, np.sin(tx_), np.cos(tx_)]])
        Ry = np.array([[np.cos(ty_), 0, np.sin(ty_)], [0, 1, 0], [-np.sin(ty_), 0, np.cos(ty_)]])
        Rcw = Rz @ Rx @ Ry
        tcw = -Rcw @ C
        d = rays @ Rcw          # world direction of every pixel's ray (R^T r, row-vector form)
        lam = -C[2] / d[..., 2]
        X = C[0] + lam * d[..., 0]
        Y = C[1] + lam * d[..., 1]
        u = X / mpt + tw / 2
        v = Y / mpt + th / 2
        u0 = np.clip(np.floor(u).astype(np.int64), 0, tw - 2)
        v0 = np.clip(np.floor(v).astype(np.int64), 0, th - 2)
        fu = np.clip(u - u0, 0, 1)
        fv = np.clip(v - v0, 0, 1)
        img = (tex[v0, u0] * (1 - fu) * (1 - fv) + tex[v0, u0 + 1] * fu * (1 - fv) + tex[v0 + 1, u0] * (1 - fu) * fv + tex[v0 + 1, u0 + 1] * fu * fv)
        frames[k] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        poses[k, :4] = R_to_quat(Rcw)
        poses[k, 4:] = tcw
    return frames, poses, (fx, fy, cx, cy), mpt


def backproject_to_plane(pose, K, xy):
    """world points (Z = 0) seen at pixels xy [n, 2] from `pose`"""
    fx, fy, cx, cy = K
    R, t = quat_to_R(pose[:4]), pose[4:]
    C = -R.T @ t
    r = np.stack([(xy[:, 0] - cx) / fx, (xy[:, 1] - cy) / fy, np.ones(len(xy))], -1) @ R  # R^T r
    lam = -C[2] / r[:, 2]
    return C[None, :] + lam[:, None] * r


# ---- the backend ---------------------------------------------------------------------------------------------------
class GpuBackend:
    """the product: HIP kernels through the C-ABI (fails loudly when liborbslam_amd.so is missing)"""
    name = "gpu"

    def __init__(self, nfeatures, w, h, device=0):
        import orbslam_in_practice_amd as oa
        self.oa = oa
        self.ext = oa.ORBextractor(nfeatures, 1.2, 8, 20, 7, max_w=w, max_h=h, max_batch=1, device=device)
        self.matcher = oa.ORBmatcher(0.9, True, max_keypoints=self.ext.cap, max_pairs=1, device=device)
        self.opt = oa.Optimizer(max_cams=16, max_points=8192, max_obs=65536, max_batch=1, device=device)
        self.bounds = (0.0, float(w), 0.0, float(h))
        self.scale_factors = np.asarray(self.ext.GetScaleFactors(), np.float32)
        self.inv_sigma2 = np.asarray(self.ext.GetInverseScaleSigmaSquares(), np.float64)

    def extract(self, img):
        return self.ext(img)

    def search_by_projection(self, cur, last, last_flags, last_xw, R, t, K, th):
        F = self.oa.matcher.FrameFeatures
        n, m, _ = self.matcher.SearchByProjection(F(cur[0], cur[1], self.bounds), F(last[0], last[1], self.bounds), last_flags, last_xw, R, t,
                                                  K[0], K[1], K[2], K[3], self.scale_factors, np.zeros(len(cur[0]), np.uint8), th)
        return n, m

    def pose_optimization(self, pose, xw, obs, inv_sigma2, K):
        return self.opt.PoseOptimization(pose, xw, obs, inv_sigma2, *K)

    def local_ba(self, window):
        return self.opt.LocalBundleAdjustment(window)

    def close(self):
        self.opt.close()
        self.matcher.close()
        self.ext.close()


class TrackerBackend(GpuBackend):
    """the same calls, with the per-frame chain extract -> SearchByProjection -> PoseOptimization as ONE device chain
    (orbslam_in_practice_amd/tracker.py, csrc/orb_track.hip): one synchronisation per frame instead of three"""
    name = "gpu-tracker"

    def __init__(self, nfeatures, w, h, device=0):
        super().__init__(nfeatures, w, h, device)
        self.tracker = self.oa.Tracker(self.ext, self.matcher, self.opt, device)

    def track(self, img, last_flags, last_xw, pose_pred, K, th, next_gray=None):
        return self.tracker.track(img, last_flags, last_xw, pose_pred, K[0], K[1], K[2], K[3], self.bounds, th, True, next_gray=next_gray)

    def reset(self):
        self.tracker.reset()

    def close(self):
        self.tracker.close()
        super().close()


# ---- the loop ------------------------------------------------------------------------------------------------------
def run_sequence(backend, frames, K, pose0, kf_every=4, ba_window=5, th=15.0):
    """returns dict(poses [n, 7], n_matches [n], n_inliers [n], matches: list of int32 arrays, ba_runs, n_points)"""
    n = len(frames)
    fx, fy, cx, cy = K
    pts = []            # map: world points, grown by back-projection of unmatched keypoints
    poses = np.zeros((n, 7))
    n_matches = np.zeros(n, np.int32)
    n_inliers = np.zeros(n, np.int32)
    all_matches = []
    kfs = []            # keyframes: dict(frame, pose, pid[n_kp], kps)
    ba_runs = 0

    def new_points(pose, kps, pid):
        free = np.where(pid < 0)[0]
        if len(free) == 0:
            return
        X = backproject_to_plane(pose, K, np.stack([kps["x"][free], kps["y"][free]], -1).astype(np.float64))
        for j, i in enumerate(free):
            pid[i] = len(pts)
            pts.append(X[j])

    fused = hasattr(backend, "track")   # a backend that runs a frame's three calls as one device chain
    # frame 0: pose given, every keypoint becomes a map point
    if fused:
        # (a recorded sequence: frame k + 1 is handed over with frame k and extracted beside its search and pose problem)
        ahead = lambda k: frames[k + 1] if (k + 1 < len(frames) and frames[k + 1].flags["C_CONTIGUOUS"] and frames[k + 1].dtype == np.uint8) else None
        r0 = backend.track(frames[0], np.zeros(0, np.uint8), np.zeros((0, 3)), np.asarray(pose0, np.float64), K, th, next_gray=ahead(0))
        kps, desc = r0["kps"], r0["desc"]
    else:
        kps, desc = backend.extract(frames[0])
    pid = np.full(len(kps), -1, np.int64)
    poses[0] = pose0
    new_points(poses[0], kps, pid)
    last = dict(kps=kps, desc=desc, pid=pid)
    kfs.append(dict(frame=0, pose=poses[0].copy(), pid=pid.copy(), kps=kps))
    all_matches.append(np.zeros(0, np.int32))
    velocity = None
    for k in range(1, n):
        # constant-velocity prediction (Tracking::TrackWithMotionModel)
        Rl, tl = quat_to_R(poses[k - 1][:4]), poses[k - 1][4:]
        if velocity is not None:
            Rp, tp = velocity[0] @ Rl, velocity[0] @ tl + velocity[1]
        else:
            Rp, tp = Rl, tl
        flags = np.where(last["pid"] >= 0, 3, 0).astype(np.uint8)
        xw = np.zeros((len(last["kps"]), 3))
        has = last["pid"] >= 0
        if has.any():
            xw[has] = np.asarray([pts[i] for i in last["pid"][has]])
        pose_pred = np.concatenate([R_to_quat(Rp), tp])
        res = None
        if fused:
            res = backend.track(frames[k], flags, xw, pose_pred, K, th, next_gray=ahead(k))
            kps, desc, nm, m = res["kps"], res["desc"], res["n_matches"], res["match"]
        else:
            kps, desc = backend.extract(frames[k])
            nm, m = backend.search_by_projection((kps, desc), (last["kps"], last["desc"]), flags, xw, Rp, tp, K, th)
        if nm < 20:  # wider window, as Tracking does (src/Tracking.cpp:347-351): through the separate calls in either case
            nm, m = backend.search_by_projection((kps, desc), (last["kps"], last["desc"]), flags, xw, Rp, tp, K, 2 * th)
            res = None
        all_matches.append(m.copy())
        n_matches[k] = nm
        cur_idx = np.where(m >= 0)[0]
        pid = np.full(len(kps), -1, np.int64)
        pid[cur_idx] = last["pid"][m[cur_idx]]
        if res is not None:
            pose, outl, ninl = res["pose"], res["outlier"][cur_idx], res["n_inliers"]
        else:
            Xc = np.asarray([pts[i] for i in pid[cur_idx]]).reshape(-1, 3)
            obs = np.stack([kps["x"][cur_idx], kps["y"][cur_idx]], -1).astype(np.float64)
            info = backend.inv_sigma2[kps["octave"][cur_idx]]
            pose, outl, ninl = backend.pose_optimization(pose_pred, Xc, obs, info, K)
        poses[k] = pose
        n_inliers[k] = ninl
        pid[cur_idx[outl != 0]] = -1        # Tracking discards the outlier associations
        Rc, tc = quat_to_R(pose[:4]), pose[4:]
        velocity = (Rc @ Rl.T, tc - Rc @ Rl.T @ tl)
        if k % kf_every == 0:
            new_points(poses[k], kps, pid)  # LocalMapping::CreateNewMapPoints stand-in
            kfs.append(dict(frame=k, pose=poses[k].copy(), pid=pid.copy(), kps=kps))
            if len(kfs) >= 3:
                win = kfs[-ba_window:]
                # points seen by at least two keyframes of the window
                cnt = {}
                for kf in win:
                    for i in kf["pid"][kf["pid"] >= 0]:
                        cnt[i] = cnt.get(i, 0) + 1
                ids = sorted(i for i, c in cnt.items() if c >= 2)
                if len(ids) >= 30:
                    lid = {p: j for j, p in enumerate(ids)}
                    cam_idx, pt_idx, ob, isg = [], [], [], []
                    for c, kf in enumerate(win):
                        for i in np.where(kf["pid"] >= 0)[0]:
                            p = kf["pid"][i]
                            if p in lid:
                                cam_idx.append(c)
                                pt_idx.append(lid[p])
                                ob.append([kf["kps"]["x"][i], kf["kps"]["y"][i]])
                                isg.append(backend.inv_sigma2[kf["kps"]["octave"][i]])
                    fixed = np.zeros(len(win), np.uint8)
                    fixed[0] = 1
                    if len(win) > 2:
                        fixed[1] = 1      # two fixed keyframes pin the gauge (scale included), like the fixed covisible observers upstream
                    window = dict(poses=np.asarray([kf["pose"] for kf in win]), points=np.asarray([pts[p] for p in ids]),
                                  cam_idx=np.asarray(cam_idx, np.int32), pt_idx=np.asarray(pt_idx, np.int32),
                                  obs=np.asarray(ob, np.float64), inv_sigma2=np.asarray(isg, np.float64), fixed=fixed,
                                  fx=fx, fy=fy, cx=cx, cy=cy)
                    res = backend.local_ba(window)
                    ba_runs += 1
                    for c, kf in enumerate(win):
                        kf["pose"] = res["poses"][c].copy()
                        poses[kf["frame"]] = kf["pose"]
                    for j, p in enumerate(ids):
                        pts[p] = res["points"][j].copy()
                    # (outlier observations are dropped from the keyframes, as LocalBundleAdjustment does)
                    out = np.asarray(res["outlier"]).astype(bool)
                    e = 0
                    for c, kf in enumerate(win):
                        for i in np.where(kf["pid"] >= 0)[0]:
                            if kf["pid"][i] in lid:
                                if out[e]:
                                    kf["pid"][i] = -1
                                e += 1
                    pid = kfs[-1]["pid"].copy()
        last = dict(kps=kps, desc=desc, pid=pid)
    return dict(poses=poses, n_matches=n_matches, n_inliers=n_inliers, matches=all_matches, ba_runs=ba_runs, n_points=len(pts))


def ate_rmse(poses, gt):
    """absolute trajectory error: RMSE of the camera centres (no alignment: frame 0 and the scale are given)"""
    def centres(P):
        return np.asarray([-quat_to_R(p[:4]).T @ p[4:] for p in P])
    d = centres(poses) - centres(gt)
    return float(np.sqrt((d * d).sum(1).mean()))
