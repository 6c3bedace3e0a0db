#!/usr/bin/env python3
"""Local-BA windows with LONG tracks (every point seen by `track` consecutive keyframes, as in a real local map; the
synthetic windows of synth.make_ba_window stop at 10): time of a batch through the solver and parity of one window.
    python tools/ba_long_tracks.py [n_cams n_points track batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402


def quat_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def long_track_window(seed, nc, npt, track):
    w = synth.make_ba_window(seed, nc, npt, outlier_frac=0.02)
    rng = np.random.default_rng(seed)
    R = [quat_R(q) for q in w["gt_poses"][:, :4]]
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


def main():
    nc, npt, track, batch = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (40, 1500, 30, 32)
    uniq = [long_track_window(500 + i, nc, npt, track) for i in range(4)]
    wins = [uniq[i % 4] for i in range(batch)]
    E = max(len(w["obs"]) for w in uniq)
    opt = oa.Optimizer(max_cams=nc, max_points=npt, max_obs=E, max_batch=batch)
    opt.solve_batch(wins, (5, 10))
    opt.set_profiling(True)
    t0 = time.perf_counter()
    res = opt.solve_batch(wins, (5, 10))
    dt = time.perf_counter() - t0
    ms, nl, team = opt.iteration_ms()
    print("%d KFs x %d points, tracks of %d (E = %d), batch %d: call %.1f ms, kernel %.1f ms, %s" % (
        nc, npt, track, E, batch, dt * 1e3, ms * nl, "phase kernels" if team == 0 else "team of %d" % team))
    if os.environ.get("CHECK"):
        from oracle import oracle as orc
        ref = orc.ba_solve(uniq[0], rounds=(5, 10))
        print("   vs oracle: rel(points) %.1e, outlier flags equal %s" % (
            np.abs(res[0]["points"] - ref["points"]).max() / np.abs(ref["points"]).max(), np.array_equal(res[0]["outlier"], ref["outlier"])))
    opt.close()


if __name__ == "__main__":
    main()
