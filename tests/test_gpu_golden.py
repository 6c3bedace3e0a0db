"""GPU: the HIP path against the COMMITTED golden fixtures (tests/golden/, made by tools/gen_golden.py with
the reference's libm trig), and the C++ drop-in classes end to end."""
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


def test_cpp_dropin_end_to_end(golden, oracle):
    """compile tests/cpp/dropin_main.cpp (reference call sites over the drop-in headers) and run it"""
    lib = os.path.join(ROOT, "orbslam_in_practice_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "dropin_main")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", "dropin_main.cpp"), "-o", exe, "-L", lib,
                               "-lorbslam_amd", "-Wl,-rpath," + lib])
        golden["frame0"].tofile(os.path.join(td, "f0.raw"))
        golden["frame1"].tofile(os.path.join(td, "f1.raw"))
        out = subprocess.check_output([exe, "640", "480", os.path.join(td, "f0.raw"), os.path.join(td, "f1.raw"),
                                       os.path.join(td, "o")], timeout=300).decode()
        assert "dropin ok" in out
        raw = open(os.path.join(td, "o_f0.bin"), "rb").read()
        n = int(np.frombuffer(raw[:4], np.int32)[0])
        kps = np.frombuffer(raw[4:4 + 28 * n], oracle.KP_DTYPE)
        desc = np.frombuffer(raw[4 + 28 * n:], np.uint8).reshape(n, 32)
        o = oracle.OrbOracle(2000, 1.2, 8, 20, 7)   # Tracking's initialisation extractor: 2 * nFeatures
        k0, d0 = o.extract(golden["frame0"])
        k1, d1 = o.extract(golden["frame1"])
        assert np.array_equal(kps, k0) and np.array_equal(desc, d0)
        mraw = np.fromfile(os.path.join(td, "o_match.bin"), np.int32)
        on, om12, _ = oracle.search_for_initialization(k0, d0, k1, d1, (0.0, 640.0, 0.0, 480.0), None, 100, 0.9, True)
        assert mraw[0] == on and np.array_equal(mraw[1:], om12)
        cosang, rms = [float(x) for x in open(os.path.join(td, "o_ba.txt")).read().split()]
        assert cosang > 0.9999 and rms < 0.05     # GlobalBundleAdjustemnt pulled KF 1 back onto the data
