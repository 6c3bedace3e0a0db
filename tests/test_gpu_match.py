"""GPU parity: HIP ORBmatcher (C-ABI) vs the CPU oracle -- distances and match indices bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def feats(oracle, frames640):
    o = oracle.OrbOracle(2000, 1.2, 8, 20, 7)
    return [o.extract(f) for f in frames640]


@pytest.fixture(scope="module")
def matcher():
    import orbslam_in_practice_amd as oa
    m = oa.ORBmatcher(0.9, True, max_keypoints=4096)
    yield m
    m.close()


def test_knn2_random(matcher, oracle):
    from orbslam_in_practice_amd import synth
    for (n1, n2, seed) in [(1000, 1000, 1), (257, 2049, 2), (1, 1, 3), (64, 3, 4), (5, 0, 5), (129, 33, 6), (40, 31, 7), (300, 4000, 8),
                           (2, 2, 9), (128, 32, 10), (33, 1, 11)]:
        d1, d2 = synth.random_descriptors(seed, n1), synth.random_descriptors(seed + 100, n2)
        if n2 > 2:
            d2[1] = d2[0]          # duplicate rows: tie -> lowest index, second == best
            d1[0] = d2[0]
        gi, gb, gs = matcher.knn2(d1, d2)
        oi, ob, os_ = oracle.hamming_knn2(d1, d2)
        assert np.array_equal(gi, oi) and np.array_equal(gb, ob) and np.array_equal(gs, os_)


def test_knn2_real_descriptors(matcher, oracle, feats):
    gi, gb, gs = matcher.knn2(feats[0][1], feats[1][1])
    oi, ob, os_ = oracle.hamming_knn2(feats[0][1], feats[1][1])
    assert np.array_equal(gi, oi) and np.array_equal(gb, ob) and np.array_equal(gs, os_)


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (100, 0.7, False), (30, 0.9, True),
                                              (-1, 0.9, True), (-1, 0.6, False)])
def test_search_for_initialization(oracle, feats, window, ratio, ori):
    import orbslam_in_practice_amd as oa
    bounds = (0.0, 640.0, 0.0, 480.0)
    m = oa.ORBmatcher(ratio, ori, max_keypoints=4096)
    (k1, d1), (k2, d2) = feats[0], feats[1]
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
    on, om12, opm = oracle.search_for_initialization(k1, d1, k2, d2, bounds, prev.copy(), window, ratio, ori)
    F1, F2 = oa.FrameFeatures(k1, d1, bounds), oa.FrameFeatures(k2, d2, bounds)
    gn, gm12 = m.SearchForInitialization(F1, F2, prev, window)
    assert on > 20, "synthetic stream should produce matches (got %d)" % on
    assert gn == on
    assert np.array_equal(gm12, om12)
    assert np.array_equal(prev, opm)
    # second call with the updated prev-matched (the reference's steady state)
    on2, om12b, opm2 = oracle.search_for_initialization(k1, d1, k2, d2, bounds, opm.copy(), window, ratio, ori)
    gn2, gm12b = m.SearchForInitialization(F1, F2, prev, window)
    assert gn2 == on2 and np.array_equal(gm12b, om12b) and np.array_equal(prev, opm2)
    m.close()


def test_host_helpers(matcher, oracle):
    from orbslam_in_practice_amd import synth
    d = synth.random_descriptors(9, 50)
    for i in range(49):
        assert matcher.DescriptorDistance(d[i], d[i + 1]) == oracle.descriptor_distance(d[i], d[i + 1])
    assert matcher.DescriptorDistance(d[0], d[0]) == 0
    rng = np.random.default_rng(0)
    for _ in range(50):
        h = rng.integers(0, 40, 30).astype(np.int32)
        assert matcher.ComputeThreeMaxima(h) == oracle.compute_three_maxima(h)


@pytest.mark.parametrize("n1,n2,nproto,flip,window", [(900, 60, 12, 3, -1), (900, 60, 12, 3, 1000), (700, 700, 25, 6, -1),
                                                       (1500, 200, 5, 10, 80), (64, 3, 1, 20, -1), (2000, 2000, 40, 4, 100)])
def test_search_for_initialization_contention(oracle, n1, n2, nproto, flip, window):
    """Many queries compete for few F2 keypoints: descriptors are a handful of prototypes with a few random bit
    flips, so almost every query's candidate list overlaps its neighbours' -- matches get stolen, vMatchedDistance
    filters whole top-8 lists (the full-rescan path) and the speculative rounds of the resolve kernel conflict all
    the time.  Greedy order matters here, so any deviation from the sequential walk shows up."""
    import orbslam_in_practice_amd as oa
    rng = np.random.default_rng(n1 * 7 + n2 + nproto)
    proto = rng.integers(0, 256, (nproto, 32), dtype=np.uint8)

    def descs(n):
        d = proto[rng.integers(0, nproto, n)].copy()
        for i in range(n):
            for b in rng.integers(0, 256, rng.integers(0, flip + 1)):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        return d

    def kps(n):
        kp = np.zeros(n, oa.KP_DTYPE)
        kp["x"], kp["y"] = rng.uniform(20, 620, n), rng.uniform(20, 460, n)
        kp["angle"] = rng.uniform(0, 360, n)
        kp["octave"] = 0
        return kp

    b = (0.0, 640.0, 0.0, 480.0)
    k1, k2, d1, d2 = kps(n1), kps(n2), descs(n1), descs(n2)
    m = oa.ORBmatcher(0.9, True, max_keypoints=2048)
    for ratio_m in (m,):
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
        on, om12, opm = oracle.search_for_initialization(k1, d1, k2, d2, b, prev.copy(), window, 0.9, True)
        gn, gm12 = ratio_m.SearchForInitialization(oa.FrameFeatures(k1, d1, b), oa.FrameFeatures(k2, d2, b), prev, window)
        assert gn == on and np.array_equal(gm12, om12) and np.array_equal(prev, opm)
    assert on > 0 or nproto == 1  # (one prototype, 20 flips: every best/second pair fails the ratio test)
    m.close()


@pytest.mark.parametrize("n1,n2,ratio,window", [(33, 129, 0.9, -1), (257, 31, 0.05, -1), (1, 1000, 0.9, -1), (1000, 1, 2.5, -1),
                                                 (130, 127, 1.0, 60), (513, 511, 0.3, -1), (96, 160, 0.9, 25), (640, 383, 0.75, 200)])
def test_search_for_initialization_matrix_core_edges(oracle, n1, n2, ratio, window):
    """The matrix-core phase 1 (k_match_knn_mfma) at its edges: set sizes that end inside a 32-row tile / a 128-keypoint
    stage / a 32-query block, ratios from 'every distance matters' (0.05: the threshold saturates at 256 and EVERY pair
    takes the hit path) to 'nothing beyond TH_LOW matters' (2.5), descriptors with no bit or every bit set (popcount 0 /
    256: the per-query threshold at its extremes), exact duplicates (more than eight candidates at distance 0: the key
    lists overflow and only the order key decides), small windows."""
    import orbslam_in_practice_amd as oa
    rng = np.random.default_rng(n1 * 131 + n2 * 7 + int(ratio * 100))
    base = rng.integers(0, 256, (12, 32), dtype=np.uint8)
    base[0] = 0
    base[1] = 255

    def descs(n):
        d = base[rng.integers(0, len(base), n)].copy()
        noisy = rng.random(n) < 0.7
        for i in np.nonzero(noisy)[0]:
            for b in rng.integers(0, 256, rng.integers(1, 40)):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        return d

    def kps(n):
        kp = np.zeros(n, oa.KP_DTYPE)
        kp["x"], kp["y"] = rng.uniform(20, 620, n), rng.uniform(20, 460, n)
        kp["angle"] = rng.uniform(0, 360, n)
        kp["octave"] = rng.integers(0, 2, n) if window >= 0 else 0
        return kp

    b = (0.0, 640.0, 0.0, 480.0)
    k1, k2, d1, d2 = kps(n1), kps(n2), descs(n1), descs(n2)
    m = oa.ORBmatcher(ratio, True, max_keypoints=2048)
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
    on, om12, opm = oracle.search_for_initialization(k1, d1, k2, d2, b, prev.copy(), window, ratio, True)
    gn, gm12 = m.SearchForInitialization(oa.FrameFeatures(k1, d1, b), oa.FrameFeatures(k2, d2, b), prev, window)
    assert gn == on and np.array_equal(gm12, om12) and np.array_equal(prev, opm)
    m.close()


def test_features_in_area_device(oracle, feats):
    """the device GetFeaturesInArea (grid-free, rank-ordered) returns what the oracle's 64x48-cell
    FrameGrid walk returns, index for index, for windows inside, across and outside the image and every level filter"""
    import orbslam_in_practice_amd as oa
    rng = np.random.default_rng(9)
    b = (0.0, 640.0, 0.0, 480.0)
    m = oa.ORBmatcher(0.9, True, max_keypoints=4096)
    k2 = feats[1][0]
    kp = np.concatenate([k2, k2[:50]])   # a few duplicates and out-of-bounds points
    kp["x"][-50:-25] = rng.uniform(-30, 0, 25)
    kp["y"][-25:] = rng.uniform(480, 520, 25)
    F = oa.FrameFeatures(kp, np.zeros((len(kp), 32), np.uint8), b)
    nq = 300
    q = np.stack([rng.uniform(-60, 700, nq), rng.uniform(-60, 540, nq), rng.choice([0.5, 5, 30, 100, 400, 2000], nq)], 1)
    q = q.astype(np.float32)
    lv = np.stack([rng.integers(-1, 5, nq), rng.integers(-1, 8, nq)], 1).astype(np.int32)
    for levels in (None, lv):
        want, wc = oracle.features_in_area(kp, b, q, levels)
        got, gc = m.GetFeaturesInArea(F, q, levels)
        assert np.array_equal(gc, wc)
        for a, c in zip(got, want):
            assert np.array_equal(a, c)
    assert wc.max() > 50 and (wc == 0).any()
    # capacity smaller than the result: the first cap entries, full count reported
    got, gc = m.GetFeaturesInArea(F, q[:20], None, cap=7)
    want, wc = oracle.features_in_area(kp, b, q[:20], None, cap=7)
    assert np.array_equal(gc, wc) and all(np.array_equal(a, c) for a, c in zip(got, want))
    # empty frame / no queries
    E = oa.FrameFeatures(np.zeros(0, oa.KP_DTYPE), np.zeros((0, 32), np.uint8), b)
    got, gc = m.GetFeaturesInArea(E, q[:3])
    assert all(len(a) == 0 for a in got) and np.all(gc == 0)
    m.close()
