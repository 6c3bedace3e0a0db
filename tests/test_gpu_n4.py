"""GPU: SURVEY N4 -- ORBmatcher::SearchByProjection / SearchByBoW.  The reference declares both with EMPTY bodies
(include/ORBmatcher.h:22,24), so there is no reference behaviour to pin: the oracle restates upstream ORB-SLAM2's
monocular semantics on the reference's data model (oracle/match_oracle.cpp) and the device path (candidate scan on the
GPU, sequential walk on the host) must reproduce it exactly -- indices, counts and the updated taken flags.
PARITY UNPINNED against any reference output, by construction."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W, H = 640, 480
FX = FY = 525.0
CX, CY = 319.5, 239.5
BOUNDS = (0.0, float(W), 0.0, float(H))
SF = (1.2 ** np.arange(8)).astype(np.float32)


@pytest.fixture(scope="module")
def feats(oracle):
    from orbslam_in_practice_amd import synth
    frames = synth.make_stream(404, 2, W, H)
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    return [o.extract(f) for f in frames]


def _lift(kps, z=4.0):
    return np.stack([(kps["x"] - CX) / FX * z, (kps["y"] - CY) / FY * z, np.full(len(kps), z)], axis=1).astype(np.float64)


def _rot_z(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


@pytest.mark.parametrize("th,ori,seed", [(15.0, True, 0), (30.0, True, 1), (30.0, False, 2), (7.0, True, 3)])
def test_search_by_projection_matches_oracle(oracle, feats, th, ori, seed):
    import orbslam_in_practice_amd as oa
    (k0, d0), (k1, d1) = feats
    rng = np.random.default_rng(seed)
    flags = ((rng.random(len(k0)) < 0.8).astype(np.uint8)) | ((rng.random(len(k0)) < 0.6).astype(np.uint8) << 1)
    flags[flags == 2] = 0                     # observations without a map point make no sense
    xw = _lift(k0)
    R, t = _rot_z(0.5 * seed), np.array([0.01 * seed, -0.02, 0.05 * (seed % 2)])
    taken = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    on, om, ot = oracle.search_by_projection(k1, d1, k0, d0, BOUNDS, flags, xw, R, t, FX, FY, CX, CY, SF, taken, th, ori)
    m = oa.ORBmatcher(0.9, ori, max_keypoints=2048)
    gn, gm, gt = m.SearchByProjection(oa.FrameFeatures(k1, d1, BOUNDS), oa.FrameFeatures(k0, d0, BOUNDS), flags, xw, R, t,
                                      FX, FY, CX, CY, SF, taken, th)
    m.close()
    assert gn == on and np.array_equal(gm, om) and np.array_equal(gt, ot)
    if th >= 15:
        assert on > 30                        # the synthetic stream moves by a few pixels: real matches exist
    # (the count is upstream's: a feature that is given a second map point because the first has no observations counts
    #  twice, so it may exceed the number of features that end up holding a point)
    assert (gm >= 0).sum() <= on or ori


def test_search_by_projection_exhausted_lists(oracle):
    """more than eight earlier features compete for the same few candidates: the closest-eight list of the later
    ones is all taken and the whole window has to be looked at again"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    rng = np.random.default_rng(5)
    base = synth.random_descriptors(9, 1)[0]
    n_last, n_cur = 40, 30
    k0 = np.zeros(n_last, oracle.KP_DTYPE)
    k1 = np.zeros(n_cur, oracle.KP_DTYPE)
    for k, n in ((k0, n_last), (k1, n_cur)):
        k["x"] = 300 + rng.uniform(-6, 6, n)
        k["y"] = 200 + rng.uniform(-6, 6, n)
        k["size"], k["octave"], k["angle"], k["class_id"] = 31, 0, 10.0, -1
    d0 = np.tile(base, (n_last, 1))
    d1 = np.tile(base, (n_cur, 1))
    for j in range(n_cur):                    # distances 0..3 so that ties and order both matter
        d1[j, j % 32] ^= np.uint8((1 << (j % 3)) - 1 if j % 3 else 0)
    flags = np.full(n_last, 3, np.uint8)
    xw = _lift(k0)
    taken = np.zeros(n_cur, np.uint8)
    on, om, ot = oracle.search_by_projection(k1, d1, k0, d0, BOUNDS, flags, xw, np.eye(3), np.zeros(3), FX, FY, CX, CY, SF,
                                             taken, 15.0, False)
    m = oa.ORBmatcher(0.9, False, max_keypoints=1024)
    gn, gm, gt = m.SearchByProjection(oa.FrameFeatures(k1, d1, BOUNDS), oa.FrameFeatures(k0, d0, BOUNDS), flags, xw, np.eye(3),
                                      np.zeros(3), FX, FY, CX, CY, SF, taken, 15.0)
    m.close()
    assert on == n_cur                        # every current feature ends up taken: later ones went past their top eight
    assert gn == on and np.array_equal(gm, om) and np.array_equal(gt, ot)


@pytest.mark.parametrize("ratio,ori", [(0.7, True), (0.9, False), (0.6, True)])
def test_search_by_bow_matches_oracle(oracle, feats, ratio, ori):
    import orbslam_in_practice_amd as oa
    (k0, d0), (k1, d1) = feats
    rng = np.random.default_rng(11)
    valid = (rng.random(len(k0)) < 0.7).astype(np.uint8)
    on, om = oracle.search_by_bow(k0, d0, valid, k1, d1, BOUNDS, ratio, ori)
    m = oa.ORBmatcher(ratio, ori, max_keypoints=2048)
    gn, gm = m.SearchByBoW(oa.FrameFeatures(k0, d0, BOUNDS), valid, oa.FrameFeatures(k1, d1, BOUNDS))
    m.close()
    assert gn == on and np.array_equal(gm, om) and on > 20
    got = gm[gm >= 0]
    assert len(np.unique(got)) == len(got) and valid[got].all()


def test_search_by_bow_exhausted_lists(oracle):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    base = synth.random_descriptors(3, 2)
    n_kf, n_f = 30, 30
    kk = np.zeros(n_kf, oracle.KP_DTYPE)
    kf = np.zeros(n_f, oracle.KP_DTYPE)
    kk["angle"], kf["angle"] = 20.0, 15.0
    dk = np.tile(base[0], (n_kf, 1))
    df = np.tile(base[0], (n_f, 1))
    for j in range(n_f):
        df[j, 0] = base[0][0] ^ np.uint8(j % 4)          # a few distance classes, many exact ties
    df[20:] = base[1]                                     # ... and a far cluster (second best for the last takers)
    valid = np.ones(n_kf, np.uint8)
    on, om = oracle.search_by_bow(kk, dk, valid, kf, df, BOUNDS, 0.9, False)
    m = oa.ORBmatcher(0.9, False, max_keypoints=1024)
    gn, gm = m.SearchByBoW(oa.FrameFeatures(kk, dk, BOUNDS), valid, oa.FrameFeatures(kf, df, BOUNDS))
    m.close()
    assert gn == on and np.array_equal(gm, om)


def _sbp_case(oracle, feats, seed, th, ori):
    (k0, d0), (k1, d1) = feats
    rng = np.random.default_rng(seed)
    flags = ((rng.random(len(k0)) < 0.8).astype(np.uint8)) | ((rng.random(len(k0)) < 0.6).astype(np.uint8) << 1)
    flags[flags == 2] = 0
    xw = _lift(k0)
    R, t = _rot_z(0.5 * seed), np.array([0.01 * seed, -0.02, 0.05 * (seed % 2)])
    taken = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    ref = oracle.search_by_projection(k1, d1, k0, d0, BOUNDS, flags, xw, R, t, FX, FY, CX, CY, SF, taken, th, ori)
    return (k0, d0, k1, d1, flags, xw, R, t, taken), ref


@pytest.mark.parametrize("th,ori,seed", [(15.0, True, 0), (30.0, True, 1), (30.0, False, 2), (7.0, True, 3)])
def test_search_by_projection_device_walk_equals_host_walk_and_oracle(oracle, feats, monkeypatch, th, ori, seed):
    """the sequential walk and the rotation histogram ON THE DEVICE (k_n4_walk: speculative rounds of 64 queries on one
    wavefront) through the host-pointer entry point: same indices, counts and taken flags as the host walk
    and the oracle"""
    import orbslam_in_practice_amd as oa
    (k0, d0, k1, d1, flags, xw, R, t, taken), (on, om, ot) = _sbp_case(oracle, feats, seed, th, ori)
    m = oa.ORBmatcher(0.9, ori, max_keypoints=2048)
    F1, F0 = oa.FrameFeatures(k1, d1, BOUNDS), oa.FrameFeatures(k0, d0, BOUNDS)
    hn, hm, ht = m.SearchByProjection(F1, F0, flags, xw, R, t, FX, FY, CX, CY, SF, taken, th)
    monkeypatch.setenv("ORBSLAM_AMD_N4_DEVICE_WALK", "1")
    gn, gm, gt = m.SearchByProjection(F1, F0, flags, xw, R, t, FX, FY, CX, CY, SF, taken, th)
    m.close()
    assert (hn, gn) == (on, on) and np.array_equal(hm, om) and np.array_equal(ht, ot)
    assert np.array_equal(gm, om) and np.array_equal(gt, ot)


def test_search_by_projection_device_walk_exhausted_lists(oracle, monkeypatch):
    """the contention case of test_search_by_projection_exhausted_lists on the device walk: queries whose eight closest
    candidates are all taken are resolved alone by the wave-wide rescan"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    monkeypatch.setenv("ORBSLAM_AMD_N4_DEVICE_WALK", "1")
    rng = np.random.default_rng(5)
    base = synth.random_descriptors(9, 1)[0]
    n_last, n_cur = 150, 90                      # (several 64-query chunks, takers and non-takers mixed)
    k0 = np.zeros(n_last, oracle.KP_DTYPE)
    k1 = np.zeros(n_cur, oracle.KP_DTYPE)
    for k, n in ((k0, n_last), (k1, n_cur)):
        k["x"] = 300 + rng.uniform(-6, 6, n)
        k["y"] = 200 + rng.uniform(-6, 6, n)
        k["size"], k["octave"], k["class_id"] = 31, 0, -1
        k["angle"] = rng.choice([10.0, 10.0, 10.0, 130.0], n)
    d0 = np.tile(base, (n_last, 1))
    d1 = np.tile(base, (n_cur, 1))
    for j in range(n_cur):
        d1[j, j % 32] ^= np.uint8((1 << (j % 3)) - 1 if j % 3 else 0)
    flags = np.where(rng.random(n_last) < 0.7, 3, 1).astype(np.uint8)
    xw = _lift(k0)
    taken = np.zeros(n_cur, np.uint8)
    for ori in (False, True):
        on, om, ot = oracle.search_by_projection(k1, d1, k0, d0, BOUNDS, flags, xw, np.eye(3), np.zeros(3), FX, FY, CX, CY, SF,
                                                 taken, 15.0, ori)
        m = oa.ORBmatcher(0.9, ori, max_keypoints=1024)
        gn, gm, gt = m.SearchByProjection(oa.FrameFeatures(k1, d1, BOUNDS), oa.FrameFeatures(k0, d0, BOUNDS), flags, xw, np.eye(3),
                                          np.zeros(3), FX, FY, CX, CY, SF, taken, 15.0)
        m.close()
        assert gn == on and np.array_equal(gm, om) and np.array_equal(gt, ot), ori
    assert on >= n_cur - 5


def test_search_by_projection_on_device_resident_frames(oracle, feats):
    """orb_search_by_projection_device: feature slabs, flags, world points and results all in HBM (torch tensors here),
    three kernels on a stream, no copy inside the call -- same results as the oracle"""
    torch = pytest.importorskip("torch")
    import orbslam_in_practice_amd as oa
    dev = torch.device("cuda", 0)
    for seed, th, ori in ((0, 15.0, True), (2, 30.0, False)):
        (k0, d0, k1, d1, flags, xw, R, t, taken), (on, om, ot) = _sbp_case(oracle, feats, seed, th, ori)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
        t_k1, t_d1, t_k0, t_d0 = up(k1), up(d1), up(k0), up(d0)
        t_fl, t_tk = up(flags), up(taken)
        t_xw = torch.from_numpy(np.ascontiguousarray(xw, np.float64)).to(dev)
        t_m = torch.full((len(k1),), -7, dtype=torch.int32, device=dev)
        t_n = torch.zeros((1,), dtype=torch.int32, device=dev)
        m = oa.ORBmatcher(0.9, ori, max_keypoints=2048)
        st = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        import ctypes as C
        m.search_by_projection_device(t_k1.data_ptr(), t_d1.data_ptr(), len(k1), t_k0.data_ptr(), t_d0.data_ptr(), len(k0),
                                      t_fl.data_ptr(), t_xw.data_ptr(), R, t, FX, FY, CX, CY, SF, BOUNDS, t_tk.data_ptr(), th,
                                      t_m.data_ptr(), t_n.data_ptr(), C.c_void_p(st.cuda_stream))
        st.synchronize()
        m.close()
        assert int(t_n.item()) == on
        assert np.array_equal(t_m.cpu().numpy(), om) and np.array_equal(t_tk.cpu().numpy(), ot)
