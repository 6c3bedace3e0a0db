"""GPU: the HIP path against the COMMITTED golden fixtures (tests/golden/, made by tools/gen_golden.py with
the reference's libm trig).  The C++ drop-in classes are exercised by tests/test_gpu_dropin.py."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(G, "orb_golden.npz"))


@pytest.mark.parametrize("name,nf", [("frame0", 1000), ("frame1", 1000), ("small", 500)])
def test_extract_golden(golden, name, nf):
    import orbslam_in_practice_amd as oa
    img = golden[name]
    e = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_w=img.shape[1], max_h=img.shape[0])
    k, d = e(img)
    assert np.array_equal(k, golden[name + "_kps"])
    assert np.array_equal(d, golden[name + "_desc"])
    assert [len(e.candidates(l)) for l in range(8)] == golden[name + "_cand_counts"].tolist()
    e.close()


def test_match_golden(golden):
    import orbslam_in_practice_amd as oa
    b = (0.0, 640.0, 0.0, 480.0)
    F1 = oa.FrameFeatures(golden["frame0_kps"], golden["frame0_desc"], b)
    F2 = oa.FrameFeatures(golden["frame1_kps"], golden["frame1_desc"], b)
    m = oa.ORBmatcher(0.9, True, max_keypoints=2048)
    for tag, win in (("win100", 100), ("brute", -1)):
        prev = np.stack([F1.kps["x"], F1.kps["y"]], axis=1).astype(np.float32).copy()
        n, m12 = m.SearchForInitialization(F1, F2, prev, win)
        assert n == int(golden["match_%s_n" % tag][0]) and np.array_equal(m12, golden["match_%s" % tag])
    m.close()


def test_ba_golden():
    import orbslam_in_practice_amd as oa
    g = np.load(os.path.join(G, "ba_golden.npz"))
    w = {k: g[k] for k in ("poses", "fixed", "points", "obs", "cam_idx", "pt_idx", "inv_sigma2")}
    w.update(fx=float(g["intr"][0]), fy=float(g["intr"][1]), cx=float(g["intr"][2]), cy=float(g["intr"][3]))
    o = oa.Optimizer()
    r = o.solve_batch([w], (5, 10))[0]
    assert np.abs(r["points"] - g["out_points"]).max() < 1e-5 * np.abs(g["out_points"]).max()
    assert np.abs(r["poses"] - g["out_poses"]).max() < 1e-5
    assert np.array_equal(r["outlier"], g["out_outlier"])
    o.close()
