"""Seeded synthetic inputs for the hot path: gray frames with FAST corners at every
pyramid scale, similarity-warped streams for matching, and random local-BA windows.

numpy only; used by tests/, bench.py and __graft_entry__.smoke().  The reference ships no data
(SURVEY R1), so every input is synthetic and says so.
"""
import numpy as np


def make_frame(seed, w=640, h=480, n_shapes=4000):
    """u8 gray frame: 128 + low-frequency sinusoid field + n_shapes random rectangles/triangles
    (axis-aligned or rotated, 4..40 px, uniform intensity) + uniform noise in {-3..3}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.full((h, w), 128.0, np.float32)
    for _ in range(8):
        fx, fy = rng.uniform(-0.02, 0.02, 2)
        ph = rng.uniform(0, 2 * np.pi)
        img += rng.uniform(0, 5.0) * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph).astype(np.float32)
    cx = rng.uniform(0, w, n_shapes)
    cy = rng.uniform(0, h, n_shapes)
    sw = rng.uniform(4, 40, n_shapes)
    sh = rng.uniform(4, 40, n_shapes)
    th = np.where(rng.random(n_shapes) < 0.5, 0.0, rng.uniform(0, np.pi, n_shapes))
    tri = rng.random(n_shapes) < 0.4
    val = rng.integers(0, 256, n_shapes).astype(np.float32)
    for k in range(n_shapes):
        r = 0.75 * max(sw[k], sh[k]) + 1
        x0, x1 = int(max(0, cx[k] - r)), int(min(w, cx[k] + r + 1))
        y0, y1 = int(max(0, cy[k] - r)), int(min(h, cy[k] + r + 1))
        if x1 <= x0 or y1 <= y0:
            continue
        X = xx[y0:y1, x0:x1] - cx[k]
        Y = yy[y0:y1, x0:x1] - cy[k]
        c, s = np.cos(th[k]), np.sin(th[k])
        U = (c * X + s * Y) / (0.5 * sw[k])
        V = (-s * X + c * Y) / (0.5 * sh[k])
        m = (np.abs(U) <= 1) & (np.abs(V) <= 1)
        if tri[k]:
            m &= (U + V) <= 0.5
        img[y0:y1, x0:x1][m] = val[k]
    img += rng.integers(-3, 4, (h, w)).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp_similarity(img, seed, max_deg=5.0, max_scale=0.03, max_t=12.0):
    """frame k+1 = frame k under a seeded similarity (nearest neighbour, integer-exact); pixels that
    fall outside sample the REFLECT_101 extension so the frame stays fully textured."""
    rng = np.random.Generator(np.random.PCG64(seed))
    h, w = img.shape
    th = np.deg2rad(rng.uniform(-max_deg, max_deg))
    sc = 1.0 + rng.uniform(-max_scale, max_scale)
    tx, ty = rng.uniform(-max_t, max_t, 2)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    X = xx - w / 2
    Y = yy - h / 2
    c, s = np.cos(th) / sc, np.sin(th) / sc
    sx = np.rint(c * X + s * Y + w / 2 - tx).astype(np.int64)
    sy = np.rint(-s * X + c * Y + h / 2 - ty).astype(np.int64)

    def refl(p, n):
        p = np.abs(p)
        p = np.where(p >= n, 2 * (n - 1) - p, p)
        return np.clip(p, 0, n - 1)

    return img[refl(sy, h), refl(sx, w)].copy()


def make_stream(seed, n, w=640, h=480, n_shapes=4000):
    """n frames: frame 0 from make_frame(seed), frame k+1 = warp(frame k)."""
    frames = [make_frame(seed, w, h, n_shapes)]
    for k in range(1, n):
        frames.append(warp_similarity(frames[-1], seed * 100003 + k))
    return np.stack(frames)


def random_descriptors(seed, n):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def make_ba_window(seed, n_cams=50, n_points=2000, outlier_frac=0.05, fx=525.0, fy=525.0, cx=319.5,
                   cy=239.5, w=640, h=480):
    """Seeded synthetic local-BA window: cameras on a noisy arc of radius 5 m looking at
    the origin, points uniform in a 4 m cube, each seen by 2..10 consecutive cameras, pixel noise
    N(0, 1.2^octave), 5 % uniform outliers, perturbed initial estimates.  All f64.

    Returns dict with poses (Nc,7: qw,qx,qy,qz,tx,ty,tz, world->camera), fixed (Nc,u8), points (Np,3),
    obs (E,2), cam_idx (E,i32), pt_idx (E,i32), inv_sigma2 (E,), gt_poses, gt_points, intrinsics."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def rot_to_quat(R):
        tr = np.trace(R)
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2
            q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
        else:
            i = int(np.argmax(np.diag(R)))
            j, k = (i + 1) % 3, (i + 2) % 3
            s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
            q = [0, 0, 0, 0]
            q[0] = (R[k, j] - R[j, k]) / s
            q[1 + i] = 0.25 * s
            q[1 + j] = (R[j, i] + R[i, j]) / s
            q[1 + k] = (R[k, i] + R[i, k]) / s
        q = np.array(q)
        return q / np.linalg.norm(q) * (1 if q[0] >= 0 else -1)

    def rodrigues(v):
        a = np.linalg.norm(v)
        if a < 1e-12:
            return np.eye(3)
        k = v / a
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K

    Rs, ts = [], []
    for i in range(n_cams):
        ang = (i / max(n_cams - 1, 1) - 0.5) * np.pi * 0.9
        C = np.array([5.0 * np.sin(ang), rng.normal(0, 0.05), -5.0 * np.cos(ang)]) + rng.normal(0, 0.03, 3)
        z = -C / np.linalg.norm(C)
        x = np.cross(np.array([0.0, 1.0, 0.0]), z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])  # world -> camera
        Rs.append(R)
        ts.append(-R @ C)
    Rs, ts = np.array(Rs), np.array(ts)
    pts = rng.uniform(-2, 2, (n_points, 3))
    obs, cam_idx, pt_idx, inv_s2 = [], [], [], []
    for j in range(n_points):
        k = int(rng.integers(2, 11))
        c0 = int(rng.integers(0, max(n_cams - k, 0) + 1))
        for c in range(c0, min(c0 + k, n_cams)):
            pc = Rs[c] @ pts[j] + ts[c]
            if pc[2] <= 0.1:
                continue
            u = fx * pc[0] / pc[2] + cx
            v = fy * pc[1] / pc[2] + cy
            if not (0 <= u < w and 0 <= v < h):
                continue
            octv = int(rng.integers(0, 8))
            s2 = 1.44 ** octv
            if rng.random() < outlier_frac:
                u, v = rng.uniform(0, w), rng.uniform(0, h)
            else:
                u += rng.normal(0, np.sqrt(s2))
                v += rng.normal(0, np.sqrt(s2))
            obs.append([u, v])
            cam_idx.append(c)
            pt_idx.append(j)
            inv_s2.append(1.0 / s2)
    gt_poses = np.concatenate([np.array([rot_to_quat(R) for R in Rs]), ts], axis=1)
    poses = gt_poses.copy()
    for i in range(1, n_cams):
        dR = rodrigues(rng.normal(0, np.deg2rad(1.0) / np.sqrt(3), 3))
        poses[i, :4] = rot_to_quat(dR @ Rs[i])
        poses[i, 4:] = ts[i] + rng.normal(0, 0.02 / np.sqrt(3), 3)
    points = pts + rng.normal(0, 0.03 / np.sqrt(3), pts.shape)
    fixed = np.zeros(n_cams, np.uint8)
    fixed[0] = 1
    return dict(poses=np.ascontiguousarray(poses), fixed=fixed, points=np.ascontiguousarray(points),
                obs=np.array(obs, np.float64).reshape(-1, 2), cam_idx=np.array(cam_idx, np.int32),
                pt_idx=np.array(pt_idx, np.int32), inv_sigma2=np.array(inv_s2, np.float64),
                gt_poses=gt_poses, gt_points=pts, fx=fx, fy=fy, cx=cx, cy=cy)


def _quat_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_ba_window_long_tracks(seed, nc, npt, track):
    """make_ba_window with every point observed by `track` consecutive keyframes (a real local map; the plain generator
    stops at 10): the extra observations are projections of the ground truth + N(0, 1) pixels; two keyframes fixed."""
    w = make_ba_window(seed, nc, npt, outlier_frac=0.02)
    rng = np.random.default_rng(seed)
    R = [_quat_R(q) for q in w["gt_poses"][:, :4]]
    t = w["gt_poses"][:, 4:]
    cam, pt, obs, info = [list(w[k]) for k in ("cam_idx", "pt_idx", "obs", "inv_sigma2")]
    seen = {(int(c), int(p)) for c, p in zip(w["cam_idx"], w["pt_idx"])}
    first = {}
    for c, p in zip(w["cam_idx"], w["pt_idx"]):
        first[int(p)] = min(first.get(int(p), nc), int(c))
    for p in range(npt):
        c0 = min(first.get(p, 0), max(nc - track, 0))
        for c in range(c0, min(nc, c0 + track)):
            if (c, p) in seen:
                continue
            pc = R[c] @ w["gt_points"][p] + t[c]
            if pc[2] <= 0.1:
                continue
            cam.append(c)
            pt.append(p)
            obs.append([w["fx"] * pc[0] / pc[2] + w["cx"] + rng.normal(0, 1), w["fy"] * pc[1] / pc[2] + w["cy"] + rng.normal(0, 1)])
            info.append(1.0)
    w["cam_idx"], w["pt_idx"] = np.asarray(cam, np.int32), np.asarray(pt, np.int32)
    w["obs"], w["inv_sigma2"] = np.asarray(obs, np.float64).reshape(-1, 2), np.asarray(info, np.float64)
    w["fixed"][:2] = 1
    return w
