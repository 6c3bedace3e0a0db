"""GPU: the C++ drop-in classes (include/orbslam_amd/*.h) called by the reference's own class names and
signatures -- tests/cpp/dropin_main.cpp instantiates and runs every public method -- with their write-backs
compared against the C-ABI path on identical inputs (same device code: tight tolerance) and the CPU oracle
(north_star tolerance 1e-5)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
REL = 1e-5


def build_dropin(outdir, extra=(), program="dropin_main"):
    """compile + link tests/cpp/<program>.cpp; include path as in the reference tree: the drop-in headers by their
    reference names ("ORBmatcher.h", "Optimizer.h") and a "Frame.h" / "KeyFrame.h" next to them"""
    lib = os.path.join(ROOT, "orbslam_in_practice_amd", "csrc")
    exe = os.path.join(outdir, program)
    subprocess.check_call(["g++", "-std=c++17", "-O1", *extra, "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "include", "orbslam_amd"), "-I", os.path.join(ROOT, "tests", "cpp", "refmodel"),
                           os.path.join(ROOT, "tests", "cpp", program + ".cpp"), "-o", exe, "-L", lib, "-lorbslam_amd",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return exe


def _quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _read_world(path, nc, npt):
    a = np.fromfile(path, np.float64)
    assert a.size == nc * 12 + npt * 3
    kf = a[:nc * 12].reshape(nc, 12)
    return kf[:, :9].reshape(nc, 3, 3), kf[:, 9:], a[nc * 12:].reshape(npt, 3)


def _inv_sigma2_table():
    t, s = [], np.float32(1.0)
    for _ in range(8):                       # OptimizerT::InvLevelSigma2: float arithmetic
        t.append(np.float32(1.0) / (s * s))
        s = s * np.float32(1.2)
    return np.array(t, np.float32).astype(np.float64)


def _as_cpp_sees_it(w):
    """the window after the trip through cv::KeyPoint (float observations, integer octave -> float table) and the
    float intrinsics of KeyFrame::fx.."""
    octv = np.rint(np.log(1.0 / w["inv_sigma2"]) / np.log(1.44)).astype(np.int32)
    w2 = dict(w)
    w2["obs"] = w["obs"].astype(np.float32).astype(np.float64)
    w2["inv_sigma2"] = _inv_sigma2_table()[octv]
    for k in ("fx", "fy", "cx", "cy"):
        w2[k] = float(np.float32(w[k]))
    return w2, octv


def _sub_window(w, cams, fixed_cams, pts):
    """flatten (local KFs, fixed KFs, local MPs) the way OptimizerT::Run does: cameras = free then fixed, points in
    the given order, per point its observations in keyframe order"""
    cam_list = list(cams) + list(fixed_cams)
    cam_of = {c: i for i, c in enumerate(cam_list)}
    pt_of = {p: i for i, p in enumerate(pts)}
    sel = [e for e in np.lexsort((w["cam_idx"], w["pt_idx"])) if w["pt_idx"][e] in pt_of and w["cam_idx"][e] in cam_of]
    sel = np.array(sel, np.int64)
    # points without an edge are removed (src/Optimizer.cpp:115-119)
    used = sorted(set(int(w["pt_idx"][e]) for e in sel))
    pts2 = [p for p in pts if p in set(used)]
    pt_of = {p: i for i, p in enumerate(pts2)}
    fixed = np.zeros(len(cam_list), np.uint8)
    fixed[len(cams):] = 1
    for i, c in enumerate(cam_list):
        if c == 0:
            fixed[i] = 1
    return dict(poses=w["poses"][cam_list].copy(), fixed=fixed, points=w["points"][pts2].copy(), obs=w["obs"][sel].copy(),
                cam_idx=np.array([cam_of[int(c)] for c in w["cam_idx"][sel]], np.int32),
                pt_idx=np.array([pt_of[int(p)] for p in w["pt_idx"][sel]], np.int32), inv_sigma2=w["inv_sigma2"][sel].copy(),
                fx=w["fx"], fy=w["fy"], cx=w["cx"], cy=w["cy"]), cam_list, pts2


def _local_window(w, cur):
    """independent numpy statement of OptimizerT::LocalWindow (rule of src/KeyFrame.cpp:54-113)"""
    nc = len(w["poses"])
    mine = set(w["pt_idx"][w["cam_idx"] == cur].tolist())
    counts = np.zeros(nc, np.int64)
    for c, p in zip(w["cam_idx"], w["pt_idx"]):
        if c != cur and p in mine:
            counts[c] += 1
    local = {cur} | set(np.nonzero(counts > 15)[0].tolist())
    if len(local) == 1 and counts.max() > 0:
        local.add(int(np.argmax(counts)))        # first (lowest id) of the maxima
    local = sorted(local)
    pts = sorted(set(w["pt_idx"][np.isin(w["cam_idx"], local)].tolist()))
    fixed = sorted(set(w["cam_idx"][np.isin(w["pt_idx"], pts)].tolist()) - set(local))
    return local, fixed, pts


def _reproj_residuals(R, t, X, w):
    pc = np.einsum("eij,ej->ei", R[w["cam_idx"]], X[w["pt_idx"]]) + t[w["cam_idx"]]
    return np.stack([w["obs"][:, 0] - (w["fx"] * pc[:, 0] / pc[:, 2] + w["cx"]),
                     w["obs"][:, 1] - (w["fy"] * pc[:, 1] / pc[:, 2] + w["cy"])], axis=1)


@pytest.mark.gpu
def test_cpp_dropin_every_method(oracle, tmp_path):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    golden = np.load(os.path.join(G, "orb_golden.npz"))
    td = str(tmp_path)
    exe = build_dropin(td)
    golden["frame0"].tofile(os.path.join(td, "f0.raw"))
    golden["frame1"].tofile(os.path.join(td, "f1.raw"))
    NC, NPT = 30, 1200
    # inlier noise only: with gross outliers kept on Huber edges for 20 gauge-free iterations the window itself is
    # unstable (tools/ba_sensitivity.py) and not even two runs of one implementation would agree
    w = synth.make_ba_window(31337, NC, NPT, outlier_frac=0.0)
    w2, octv = _as_cpp_sees_it(w)
    with open(os.path.join(td, "win.bin"), "wb") as f:
        np.array([NC, NPT, len(w["obs"])], np.int32).tofile(f)
        w["poses"].astype(np.float64).tofile(f)
        w["points"].astype(np.float64).tofile(f)
        w["obs"].astype(np.float32).tofile(f)
        w["cam_idx"].astype(np.int32).tofile(f)
        w["pt_idx"].astype(np.int32).tofile(f)
        octv.tofile(f)
        np.array([w["fx"], w["fy"], w["cx"], w["cy"]], np.float32).tofile(f)
    out = subprocess.run([exe, "640", "480", os.path.join(td, "f0.raw"), os.path.join(td, "f1.raw"),
                          os.path.join(td, "win.bin"), os.path.join(td, "o")], timeout=600, capture_output=True)
    assert out.returncode == 0, (out.returncode, out.stdout.decode(), out.stderr.decode())
    assert b"dropin ok" in out.stdout

    # ---- ORBextractor::operator() / Frame ctor (src/Frame.cpp:77) ----
    raw = open(os.path.join(td, "o_f0.bin"), "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    kps = np.frombuffer(raw[4:4 + 28 * n], oracle.KP_DTYPE)
    desc = np.frombuffer(raw[4 + 28 * n:], np.uint8).reshape(n, 32)
    o = oracle.OrbOracle(2000, 1.2, 8, 20, 7)   # Tracking's initialisation extractor: 2 * nFeatures
    k0, d0 = o.extract(golden["frame0"])
    k1, d1 = o.extract(golden["frame1"])        # the oracle now holds frame 1's pyramid, as the extractor does
    assert np.array_equal(kps, k0) and np.array_equal(desc, d0)
    # ---- mvImagePyramid read as a vector<Mat>: ROI + BORDER_REFLECT_101 apron (src/ORBextractor.cpp:1078-1093) ----
    pyr = open(os.path.join(td, "o_pyr.bin"), "rb").read()
    nl = int(np.frombuffer(pyr[:4], np.int32)[0])
    assert nl == 8
    off = 4
    for l in range(nl):
        cols, rows, step = np.frombuffer(pyr[off:off + 12], np.int32)
        off += 12
        padded = np.frombuffer(pyr[off:off + (cols + 38) * (rows + 38)], np.uint8).reshape(rows + 38, cols + 38)
        off += (cols + 38) * (rows + 38)
        assert step == cols + 38
        assert np.array_equal(padded, np.pad(o.level(l), 19, mode="reflect")), l
    # ---- ORBmatcher::SearchForInitialization, image bounds never set by hand ----
    mraw = np.fromfile(os.path.join(td, "o_match.bin"), np.int32)
    on, om12, _ = oracle.search_for_initialization(k0, d0, k1, d1, (0.0, 640.0, 0.0, 480.0), None, 100, 0.9, True)
    assert mraw[0] == on and on > 100 and np.array_equal(mraw[1:], om12)

    # ---- SURVEY N4 through the C++ classes: SearchByProjection / SearchByBoW fill mvpMappts / vpMatches like the C-ABI ----
    n4 = np.fromfile(os.path.join(td, "o_n4.bin"), np.int32)
    nproj, nbow, n1 = n4[:3]
    cpp_proj, cpp_bow = n4[3:3 + n1], n4[3 + n1:3 + 2 * n1]
    fxf, fyf, cxf, cyf = (float(np.float32(w[k])) for k in ("fx", "fy", "cx", "cy"))
    idx0 = np.arange(len(k0))
    has = idx0 % 4 != 3
    flags = (has & (idx0 != 5)).astype(np.uint8) | ((has & (idx0 % 2 == 0) & (idx0 != 5)).astype(np.uint8) << 1)
    xw0 = np.stack([(k0["x"].astype(np.float64) - cxf) / fxf * 4.0, (k0["y"].astype(np.float64) - cyf) / fyf * 4.0,
                    np.full(len(k0), 4.0)], axis=1)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2))])).astype(np.float32)
    bnd = (0.0, 640.0, 0.0, 480.0)
    mm = oa.ORBmatcher(0.9, True, max_keypoints=4096)
    gn, gm, _ = mm.SearchByProjection(oa.FrameFeatures(k1, d1, bnd), oa.FrameFeatures(k0, d0, bnd), flags, xw0, np.eye(3),
                                      np.zeros(3), fxf, fyf, cxf, cyf, sf, np.zeros(len(k1), np.uint8), 30.0)
    mm.close()
    assert nproj == gn and gn > 30 and np.array_equal(cpp_proj, gm)      # map point id == frame-0 feature index
    mm = oa.ORBmatcher(0.7, True, max_keypoints=4096)
    bn, bm = mm.SearchByBoW(oa.FrameFeatures(k0, d0, bnd), has.astype(np.uint8), oa.FrameFeatures(k1, d1, bnd))
    mm.close()
    assert nbow == bn and bn > 20 and np.array_equal(cpp_bow, bm)

    opt = oa.Optimizer(max_cams=64, max_points=4096, max_obs=32768, max_batch=1)
    # ---- Optimizer::GlobalBundleAdjustemnt(pMap, 20): the reference's graph, gauge free -> gauge-invariant compare ----
    R, t, X = _read_world(os.path.join(td, "o_gba.bin"), NC, NPT)
    got = opt.solve_batch([w2], (20,))[0]
    ref = oracle.ba_solve(w2, rounds=(20,))
    Rg = np.array([_quat_to_R(q) for q in got["poses"][:, :4]])
    r_cpp = _reproj_residuals(R, t, X, w2)
    r_abi = _reproj_residuals(Rg, got["poses"][:, 4:], got["points"], w2)
    Rr = np.array([_quat_to_R(q) for q in ref["poses"][:, :4]])
    r_orc = _reproj_residuals(Rr, ref["poses"][:, 4:], ref["points"], w2)
    inl = np.abs(r_orc).max(axis=1) < 20
    scale = np.abs(r_orc[inl]).max()
    assert np.abs(r_cpp - r_abi)[inl].max() < 1e-7 * scale
    assert np.abs(r_cpp - r_orc)[inl].max() < REL * scale
    assert np.array_equal(R[0], _quat_to_R(w2["poses"][0, :4])) and np.array_equal(t[0], w2["poses"][0, 4:])  # KF 0 fixed
    # ---- Optimizer::BundleAdjustment(vpKF, vpMP, 5, &stop, 0, false) ----
    R5, t5, X5 = _read_world(os.path.join(td, "o_ba5.bin"), NC, NPT)
    got5 = opt.solve_batch([w2], (5,), robust=False)[0]
    assert np.abs(X5 - got5["points"]).max() < 1e-7 * np.abs(got5["points"]).max()
    assert np.abs(t5 - got5["poses"][:, 4:]).max() < 1e-7 * np.abs(got5["poses"][:, 4:]).max()
    # ---- Optimizer::LocalBundleAdjustment(pKF, &stop, pMap): window from the observations, fixed observers ----
    cur = NC // 2
    lw = np.fromfile(os.path.join(td, "o_lwin.bin"), np.int32)
    nl_, nf_, np_ = lw[:3]
    local, fixed, pts = _local_window(w2, cur)
    assert lw[3:3 + nl_].tolist() == local and lw[3 + nl_:3 + nl_ + nf_].tolist() == fixed
    assert lw[3 + nl_ + nf_:].tolist() == pts and len(local) > 3 and len(fixed) > 0 and len(pts) > 100
    sub, cam_list, pts2 = _sub_window(w2, local, fixed, pts)
    gl = opt.solve_batch([sub], (5, 10))[0]
    rl = oracle.ba_solve(sub, rounds=(5, 10))
    Rl, tl, Xl = _read_world(os.path.join(td, "o_lba.bin"), NC, NPT)
    assert np.abs(Xl[pts2] - gl["points"]).max() < 1e-9 * np.abs(gl["points"]).max()
    assert np.abs(Xl[pts2] - rl["points"]).max() < REL * np.abs(rl["points"]).max()
    free = [i for i, c in enumerate(cam_list) if not sub["fixed"][i]]
    assert np.abs(tl[[cam_list[i] for i in free]] - gl["poses"][free, 4:]).max() < 1e-9 * np.abs(gl["poses"][:, 4:]).max()
    assert np.abs(tl[[cam_list[i] for i in free]] - rl["poses"][free, 4:]).max() < REL * np.abs(rl["poses"][:, 4:]).max()
    for i in free:
        assert np.abs(Rl[cam_list[i]] - _quat_to_R(gl["poses"][i, :4])).max() < 1e-9
    # everything outside the window is untouched, fixed observers included
    outside = sorted(set(range(NC)) - set(local))
    assert all(np.array_equal(tl[c], w2["poses"][c, 4:]) for c in outside)
    notpts = sorted(set(range(NPT)) - set(pts2))
    assert np.array_equal(Xl[notpts], w2["points"][notpts])
    assert np.abs(Xl[pts2] - w2["points"][pts2]).max() > 1e-4          # and the window did move
    # window-explicit form over the whole map (no fixed observers except KF 0)
    Ra, ta, Xa = _read_world(os.path.join(td, "o_lba_all.bin"), NC, NPT)
    ga = opt.solve_batch([w2], (5, 10))[0]
    ra = _reproj_residuals(Ra, ta, Xa, w2)
    rb = _reproj_residuals(np.array([_quat_to_R(q) for q in ga["poses"][:, :4]]), ga["poses"][:, 4:], ga["points"], w2)
    assert np.abs(ra - rb)[inl].max() < 1e-7 * scale
    # ---- Optimizer::PoseOptimization(Frame*): intrinsics through GetCameraPara ----
    po = np.fromfile(os.path.join(td, "o_po.bin"), np.float64)
    T, ret, outl = po[:16].reshape(4, 4), int(po[16]), po[17:].astype(np.uint8)
    sel = np.nonzero(w2["cam_idx"] == cur)[0]
    has = np.array([i % 3 != 2 for i in range(len(sel))])
    xw = w2["points"][w2["pt_idx"][sel[has]]]
    gp, go, gn = opt.PoseOptimization(w2["poses"][cur], xw, w2["obs"][sel[has]], w2["inv_sigma2"][sel[has]],
                                      w2["fx"], w2["fy"], w2["cx"], w2["cy"])
    rp, ro, rn = oracle.pose_optimization(w2["poses"][cur], xw, w2["obs"][sel[has]], w2["inv_sigma2"][sel[has]],
                                          w2["fx"], w2["fy"], w2["cx"], w2["cy"])
    assert ret == gn == rn and 3 < ret < has.sum()
    assert np.array_equal(outl[has], go) and np.array_equal(go, ro)
    assert outl[~has].all()      # slots without a map point are not touched (src/Optimizer.cpp:172-176): the test set them
    assert np.abs(T[:3, :3] - _quat_to_R(gp[:4])).max() < 1e-9 and np.abs(T[:3, 3] - gp[4:]).max() < 1e-9
    assert np.abs(T[:3, 3] - rp[4:]).max() < REL * np.abs(rp[4:]).max()
    assert np.array_equal(T[3], [0, 0, 0, 1])
    opt.close()
