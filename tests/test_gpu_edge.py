"""GPU parity on the edge cases: tiny / degenerate image sizes, extreme feature counts, noise images with
far more candidates than quota, all-zero descriptors, empty frames in the matcher."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(ext, oracle, img, nf, nlevels=8, ini=20, mn=7, scale=1.2):
    o = oracle.OrbOracle(nf, scale, nlevels, ini, mn)
    ok, od = o.extract(img)
    k, d = ext(img)
    assert np.array_equal(k, ok) and np.array_equal(d, od), (img.shape, nf, len(k), len(ok))
    return len(k)


@pytest.mark.parametrize("w,h", [(40, 40), (97, 33), (64, 48), (127, 95), (200, 131), (641, 479), (1000, 200)])
def test_small_and_odd_sizes(oracle, w, h):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    img = synth.make_frame(w * 1000 + h, w, h, n_shapes=max(30, w * h // 80))
    e = oa.ORBextractor(500, 1.2, 8, 20, 7, max_w=w, max_h=h)
    _same(e, oracle, img, 500)
    e.close()


def test_tall_image_reports_aspect_error():
    """(w-32)/(h-32) rounds to 0 root nodes: the reference divides by zero (src/ORBextractor.cpp:493-495)"""
    import orbslam_in_practice_amd as oa
    with pytest.raises(oa.OrbError) as ei:
        oa.ORBextractor(500, 1.2, 8, 20, 7, max_w=100, max_h=400)
    assert ei.value.code == oa._lib.ORB_E_ASPECT


@pytest.mark.parametrize("nf,nlevels,scale", [(50, 8, 1.2), (4000, 8, 1.2), (1000, 4, 1.5), (1000, 1, 1.2), (300, 12, 1.1),
                                             (2000, 2, 1.3), (2000, 1, 1.2)])  # the last two: > 1020 features on one level
def test_feature_counts_and_pyramid_shapes(oracle, nf, nlevels, scale):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    img = synth.make_frame(nf + nlevels, 640, 480)
    e = oa.ORBextractor(nf, scale, nlevels, 20, 7, max_w=640, max_h=480)
    n = _same(e, oracle, img, nf, nlevels=nlevels, scale=scale)
    assert n >= min(nf, 300)
    e.close()


@pytest.mark.parametrize("scale,nlevels", [(2.0, 3), (3.0, 2), (4.5, 2)])
def test_large_scale_steps(oracle, scale, nlevels):
    """scale steps far beyond the reference's 1.2: the resize kernel's LDS window grows with the step and, past its
    budget, the kernel falls back to direct gathers -- both must stay bit-exact"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    img = synth.make_frame(77, 1000, 760)
    e = oa.ORBextractor(600, scale, nlevels, 20, 7, max_w=1000, max_h=760)
    n = _same(e, oracle, img, 600, nlevels=nlevels, scale=scale)
    assert n > 100
    e.close()


def test_noise_image_saturates_every_cell(oracle):
    import orbslam_in_practice_amd as oa
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (480, 640)).astype(np.uint8)     # ~70k FAST candidates per frame
    e = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=640, max_h=480)
    _same(e, oracle, img, 1000)
    # high thresholds: only the minThFAST fallback finds anything in most cells
    img2 = (rng.integers(0, 40, (480, 640)) + 100).astype(np.uint8)
    e2 = oa.ORBextractor(1000, 1.2, 8, 60, 30, max_w=640, max_h=480)
    _same(e2, oracle, img2, 1000, ini=60, mn=30)
    e.close()
    e2.close()


def test_handle_reuse_across_sizes_and_batches(oracle):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    e = oa.ORBextractor(800, 1.2, 8, 20, 7, max_w=320, max_h=240, max_batch=2)
    for (w, h) in [(320, 240), (800, 600), (160, 120), (800, 600)]:   # grows, shrinks, regrows
        _same(e, oracle, synth.make_frame(w + h, w, h, n_shapes=w * h // 100), 800)
    frames = synth.make_stream(9, 2, 320, 240, n_shapes=800)
    res = e.extract_batch(frames)
    o = oracle.OrbOracle(800, 1.2, 8, 20, 7)
    for f, (k, d) in zip(frames, res):
        ok, od = o.extract(f)
        assert np.array_equal(k, ok) and np.array_equal(d, od)
    with pytest.raises(oa.OrbError):
        e.extract_batch(np.stack([frames[0]] * 3))               # batch > max_batch -> ORB_E_CAPACITY
    e.close()


def test_matcher_edge_cases(oracle):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    b = (0.0, 640.0, 0.0, 480.0)
    m = oa.ORBmatcher(0.9, True, max_keypoints=512)
    empty = oa.FrameFeatures(np.zeros(0, oa.KP_DTYPE), np.zeros((0, 32), np.uint8), b)
    kp = np.zeros(40, oa.KP_DTYPE)
    rng = np.random.default_rng(2)
    kp["x"], kp["y"] = rng.uniform(20, 620, 40), rng.uniform(20, 460, 40)
    kp["angle"] = rng.uniform(0, 360, 40)
    d = synth.random_descriptors(3, 40)
    F = oa.FrameFeatures(kp, d, b)
    prev = np.stack([kp["x"], kp["y"]], axis=1).astype(np.float32).copy()
    assert m.SearchForInitialization(empty, F, np.zeros((0, 2), np.float32), 100) == (0, pytest.approx(np.zeros(0)))
    n, m12 = m.SearchForInitialization(F, empty, prev.copy(), 100)
    assert n == 0 and np.all(m12 == -1)
    # identical frames: everything matches itself at distance 0 unless the ratio test kills duplicates
    for win in (100, -1):
        n, m12 = m.SearchForInitialization(F, F, prev.copy(), win)
        on, om12, _ = oracle.search_for_initialization(kp, d, kp, d, b, prev.copy(), win, 0.9, True)
        assert n == on and np.array_equal(m12, om12) and n > 10
    # all-equal descriptors: best == second -> ratio test rejects everything
    dz = np.zeros((40, 32), np.uint8)
    Fz = oa.FrameFeatures(kp, dz, b)
    n, m12 = m.SearchForInitialization(Fz, Fz, prev.copy(), 1000)
    on, om12, _ = oracle.search_for_initialization(kp, dz, kp, dz, b, prev.copy(), 1000, 0.9, True)
    assert n == on == 0 and np.array_equal(m12, om12)
    # keypoints on the grid boundary and outside the image bounds (not assigned to the grid)
    kp2 = kp.copy()
    kp2["x"][:5] = [0.0, 639.9, 640.0, -3.0, 645.0]
    kp2["y"][:5] = [0.0, 479.9, 480.0, 10.0, 470.0]
    F2 = oa.FrameFeatures(kp2, d, b)
    prev2 = np.stack([kp2["x"], kp2["y"]], axis=1).astype(np.float32).copy()
    n, m12 = m.SearchForInitialization(F2, F2, prev2.copy(), 50)
    on, om12, _ = oracle.search_for_initialization(kp2, d, kp2, d, b, prev2.copy(), 50, 0.9, True)
    assert n == on and np.array_equal(m12, om12)
    m.close()


def test_error_text_of_matcher_and_ba_reaches_orb_last_error():
    """one error slot per host thread for the whole library: a BA / matcher failure's text is what
    orb_last_error() (and hence OrbError) reports, not a stale extractor message"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    o = oa.Optimizer(max_cams=16, max_points=256, max_obs=4096, max_batch=2)
    w1, w2 = synth.make_ba_window(1, 6, 100), synth.make_ba_window(2, 6, 100)
    Ps = (oa._lib.BAProblem * 2)()
    Rs = (oa._lib.BAResult * 2)()
    k1 = oa.Optimizer._fill(Ps[0], w1, (5, 10), True, 5.991, 0, None)
    k2 = oa.Optimizer._fill(Ps[1], w2, (3,), True, 5.991, 0, None)   # a different schedule in one batch
    bufs = [np.zeros((6, 7)), np.zeros((100, 3)), np.zeros((6, 7)), np.zeros((100, 3))]
    Rs[0].poses, Rs[0].points, Rs[1].poses, Rs[1].points = [b.ctypes.data for b in bufs]
    with pytest.raises(oa.OrbError) as ei:
        oa._lib.check(o._L.ba_solve_batch(o._s, Ps, Rs, 2))
    assert ei.value.code == oa._lib.ORB_E_INVALID and "one iteration schedule" in str(ei.value)
    o.close()
    m = oa.ORBmatcher(0.9, True, max_keypoints=32768)
    kp = np.zeros(20000, oa.KP_DTYPE)
    F = oa.FrameFeatures(kp, np.zeros((20000, 32), np.uint8), (0.0, 640.0, 0.0, 480.0))
    with pytest.raises(oa.OrbError) as ei:
        m.GetFeaturesInArea(F, np.array([[10.0, 10.0, 5.0]], np.float32))
    assert ei.value.code == oa._lib.ORB_E_CAPACITY and "16384" in str(ei.value)
    m.close()
    del k1, k2


@pytest.mark.parametrize("w,h,nf,nl,nb", [(1346, 154, 100, 7, 3), (1008, 155, 100, 5, 1), (900, 120, 50, 4, 2),
                                          (1938, 115, 5, 3, 3)])  # 23 / 26 / 31 roots per level, 317 keypoints for a budget of 5
def test_wide_frames_return_more_than_their_budget(oracle, w, h, nf, nl, nb):
    """nIni = round(W / H) quadtree roots (src/ORBextractor.cpp:493-534): a wide frame with a small feature budget
    returns up to 4 * nIni keypoints per level -- several times nfeatures.  The capacity bound has to follow the
    frame shape (orb_max_keypoints), in the batch and in the single-frame (latency) path, and through a handle that
    was created for another shape."""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    frames = np.stack([synth.make_frame(1000 + i, w, h, n_shapes=2000) for i in range(nb)])
    o = oracle.OrbOracle(nf, 1.2, nl, 20, 3)
    want = [o.extract(f) for f in frames]
    assert max(len(k) for k, _ in want) > nf + 4 * nl      # more than the naive sum of quotas
    for create_w, create_h in ((w, h), (640, 480)):
        e = oa.ORBextractor(nf, 1.2, nl, 20, 3, max_w=create_w, max_h=create_h, max_batch=nb)
        got = e.extract_batch(frames) if nb > 1 else [e(frames[0])]
        for (k, d), (ok, od) in zip(got, want):
            assert np.array_equal(k, ok) and np.array_equal(d, od)
        e.close()


def test_extractors_on_concurrent_host_threads(oracle):
    """one extractor handle per host thread (the documented model), different frame sizes, growing buffers and changing
    geometry while the others run: every thread's results equal the oracle's (the buffer-generation sink and
    the per-device LDS limits were process-wide and unsynchronised)"""
    import threading
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    shapes = [(640, 480, 1000), (320, 240, 500), (800, 600, 1500)]
    frames = {s: [synth.make_frame(900 + 10 * i + k, s[0], s[1], n_shapes=s[0] * s[1] // 100) for k in range(4)]
              for i, s in enumerate(shapes)}
    refs = {}
    for s in shapes:
        o = oracle.OrbOracle(s[2], 1.2, 8, 20, 7)
        refs[s] = [o.extract(f) for f in frames[s]]
    errors = []

    def worker(s, order):
        try:
            # created small on purpose: the first call of every size has to grow the buffers and rebuild the geometry
            e = oa.ORBextractor(s[2], 1.2, 8, 20, 7, max_w=64, max_h=64, max_batch=1)
            for rep in range(3):
                for k in order:
                    kp, d = e(frames[s][k])
                    rk, rd = refs[s][k]
                    if not (np.array_equal(kp, rk) and np.array_equal(d, rd)):
                        errors.append((s, rep, k))
            e.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((s, repr(ex)))

    threads = [threading.Thread(target=worker, args=(s, [0, 1, 2, 3] if i % 2 == 0 else [3, 2, 1, 0]))
               for i, s in enumerate(shapes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors


def test_tracking_thread_beside_local_mapping_thread(oracle):
    """the deployment the reference is written for: a Tracking thread (extraction -- HIP-graph capture and replay --,
    matching, pose optimisation) and a LocalMapping thread (local bundle adjustment, buffers growing from window to
    window) on their own handles at the same time: every result equals the one computed alone"""
    import threading
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    frames = [synth.make_frame(4100 + k, 640, 480) for k in range(6)]
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    ref_frames = [o.extract(f) for f in frames]
    wins = [synth.make_ba_window(4200 + k, 10 + 6 * k, 300 + 350 * k) for k in range(4)]
    for w in wins:
        w["fixed"][:2] = 1
    ref_ba = [oracle.ba_solve(w, rounds=(5, 10)) for w in wins]
    pw = synth.make_ba_window(4300, 6, 400)
    sel = np.nonzero(pw["cam_idx"] == 2)[0]
    pxw, pobs, pinfo = pw["gt_points"][pw["pt_idx"][sel]], pw["obs"][sel], pw["inv_sigma2"][sel]
    ref_pose = oracle.pose_optimization(pw["poses"][2], pxw, pobs, pinfo, pw["fx"], pw["fy"], pw["cx"], pw["cy"])
    errors = []

    def tracking():
        try:
            e = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=640, max_h=480, max_batch=1)
            po = oa.Optimizer(max_cams=8, max_points=1024, max_obs=4096, max_batch=1)
            for rep in range(8):
                for k, f in enumerate(frames):
                    kp, d = e(f)
                    if not (np.array_equal(kp, ref_frames[k][0]) and np.array_equal(d, ref_frames[k][1])):
                        errors.append(("extract", rep, k))
                    p, outl, ninl = po.PoseOptimization(pw["poses"][2], pxw, pobs, pinfo, pw["fx"], pw["fy"], pw["cx"], pw["cy"])
                    if ninl != ref_pose[2] or not np.array_equal(outl, ref_pose[1]) or np.abs(p - ref_pose[0]).max() > 1e-5:
                        errors.append(("pose", rep, k))
            po.close()
            e.close()
        except Exception as ex:  # noqa: BLE001
            errors.append(("tracking", repr(ex)))

    def mapping():
        try:
            ba = oa.Optimizer(max_cams=8, max_points=256, max_obs=1024, max_batch=1)   # small on purpose: every window grows it
            for rep in range(3):
                for k, w in enumerate(wins):
                    g = ba.LocalBundleAdjustment(w)
                    r = ref_ba[k]
                    if not (np.array_equal(g["outlier"], r["outlier"]) and
                            np.abs(g["points"] - r["points"]).max() <= 1e-5 * np.abs(r["points"]).max() and
                            np.abs(g["poses"] - r["poses"]).max() <= 1e-5 * max(np.abs(r["poses"]).max(), 1.0)):
                        errors.append(("ba", rep, k))
            ba.close()
        except Exception as ex:  # noqa: BLE001
            errors.append(("mapping", repr(ex)))

    ts = [threading.Thread(target=tracking), threading.Thread(target=mapping)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors
