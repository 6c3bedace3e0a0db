#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the build container).

The reference ships no fixtures and cannot be built, so the vectors come from the
CPU oracle -- with the reference's own libm cosf/sinf (ORC_TRIG_LIBM) for the descriptor steering, so
that the deterministic-trig path used by the device is pinned against it.  Inputs are stored too
(synthetic frames; numpy's float sin is not guaranteed bit-identical across machines).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(G, exist_ok=True)
    # 1) constructor tables and level geometry
    o = orc.OrbOracle(1000, 1.2, 8, 20, 7)
    t = o.tables()
    o2 = orc.OrbOracle(2000, 1.2, 8, 20, 7)
    ws, hs = o.level_sizes(640, 480)
    ws2, hs2 = o.level_sizes(1280, 720)
    json.dump({"features_per_level_1000": t["features_per_level"].tolist(),
               "features_per_level_2000": o2.tables()["features_per_level"].tolist(),
               "umax": t["umax"].tolist(), "scale": [float(x) for x in t["scale"]],
               "level_sizes_640x480": [ws.tolist(), hs.tolist()],
               "level_sizes_1280x720": [ws2.tolist(), hs2.tolist()]},
              open(os.path.join(G, "tables.json"), "w"), indent=1)
    # 2) frames + full extractor outputs (libm trig) + matches
    frames = synth.make_stream(2024, 2, 640, 480)
    small = synth.make_frame(77, 320, 240, n_shapes=1200)
    out = {"frame0": frames[0], "frame1": frames[1], "small": small}
    feats = []
    for name, img, nf in (("frame0", frames[0], 1000), ("frame1", frames[1], 1000), ("small", small, 500)):
        oo = orc.OrbOracle(nf, 1.2, 8, 20, 7)
        k, d = oo.extract(img, trig=orc.TRIG_LIBM)
        out[name + "_kps"] = k
        out[name + "_desc"] = d
        out[name + "_cand_counts"] = np.array([len(oo.candidates(l)) for l in range(8)], np.int32)
        feats.append((k, d))
    bounds = (0.0, 640.0, 0.0, 480.0)
    for tag, win in (("win100", 100), ("brute", -1)):
        n, m12, pm = orc.search_for_initialization(feats[0][0], feats[0][1], feats[1][0], feats[1][1], bounds, None, win,
                                                   0.9, True)
        out["match_%s" % tag] = m12
        out["match_%s_n" % tag] = np.array([n], np.int32)
    np.savez_compressed(os.path.join(G, "orb_golden.npz"), **out)
    # 3) BA: a small gauge-fixed window and the oracle's estimates
    w = synth.make_ba_window(4242, 12, 300)
    w["fixed"][:2] = 1
    r = orc.ba_solve(w, rounds=(5, 10))
    np.savez_compressed(os.path.join(G, "ba_golden.npz"), **{k: v for k, v in w.items() if isinstance(v, np.ndarray)},
                        intr=np.array([w["fx"], w["fy"], w["cx"], w["cy"]]), out_poses=r["poses"],
                        out_points=r["points"], out_outlier=r["outlier"],
                        out_chi2=np.array([r["chi2_initial"], r["chi2_final"]]))
    print("golden written:", os.listdir(G))


if __name__ == "__main__":
    main()
