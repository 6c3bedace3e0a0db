"""GPU parity of the DEVICE-RESIDENT batch entry points -- exactly what bench.py times: frames in HBM (torch
tensors are only the allocator), orb_extract_batch_device on a caller stream, orb_match_batch_device with
pair lists; plus strided / unaligned inputs that take the non-vectorised level-0 copy."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _np_kps(t, oracle):
    a = t.cpu().numpy()
    return a.reshape(a.shape[0], -1).view(oracle.KP_DTYPE).reshape(a.shape[0], a.shape[1])


@pytest.mark.parametrize("window,nf", [(-1, 1000), (100, 2000)])
def test_device_batch_extract_and_match(oracle, window, nf):
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import shard, synth
    B, W, H = 6, 640, 480
    frames = synth.make_stream(31, B, W, H)
    dev = torch.device("cuda", 0)
    ext = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B)
    cap = ext.cap
    m = oa.ORBmatcher(0.9, True, max_keypoints=cap, max_pairs=B)
    d_img = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    pa_h, pb_h = shard.stream_pairs(0, 1, B)
    pa, pb = torch.from_numpy(pa_h).to(dev), torch.from_numpy(pb_h).to(dev)
    d_m12 = torch.full((B, cap), -7, dtype=torch.int32, device=dev)
    d_nm = torch.zeros((B,), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(st.cuda_stream)
    bounds = (0.0, float(W), 0.0, float(H))
    with torch.cuda.stream(st):
        for _ in range(2):  # twice: buffers are reused, results must not depend on what was there before
            ext.extract_batch_device(d_img.data_ptr(), B, W, H, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                     d_cnt.data_ptr(), sp)
            m.match_batch_device(d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, pa.data_ptr(), pb.data_ptr(),
                                 B, bounds, window, d_m12.data_ptr(), d_nm.data_ptr(), sp)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    kps = _np_kps(d_kps, oracle)
    desc = d_desc.cpu().numpy()
    m12 = d_m12.cpu().numpy()
    nm = d_nm.cpu().numpy()
    o = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    feats = [o.extract(f) for f in frames]
    for i in range(B):
        assert cnt[i] == len(feats[i][0])
        assert np.array_equal(kps[i, :cnt[i]], feats[i][0]) and np.array_equal(desc[i, :cnt[i]], feats[i][1])
    for p in range(B):
        a, b = int(pa_h[p]), int(pb_h[p])
        on, om12, _ = oracle.search_for_initialization(feats[a][0], feats[a][1], feats[b][0], feats[b][1], bounds, None,
                                                       window, 0.9, True)
        assert nm[p] == on and np.array_equal(m12[p, :cnt[a]], om12), (p, nm[p], on)
    assert nm[:-1].min() > 20
    m.close()
    ext.close()


def test_strided_and_unaligned_device_input(oracle):
    """row stride > width and a base pointer that is not 16-byte aligned: the level-0 copy takes its scalar
    path; results must not change"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    W, H, S = 333, 257, 352
    frames = np.stack([synth.make_frame(70 + i, W, H, n_shapes=900) for i in range(2)])
    dev = torch.device("cuda", 0)
    ext = oa.ORBextractor(600, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=2)
    cap = ext.cap
    buf = torch.zeros((2 * S * H + 64,), dtype=torch.uint8, device=dev)
    off = 5   # deliberately misaligned start
    padded = np.zeros((2, H, S), np.uint8)
    padded[:, :, :W] = frames
    buf[off:off + 2 * S * H] = torch.from_numpy(padded.reshape(-1)).to(dev)
    d_kps = torch.zeros((2, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros((2,), dtype=torch.int32, device=dev)
    ext.extract_batch_device(buf.data_ptr() + off, 2, W, H, S, S * H, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                             d_cnt.data_ptr(), None)
    ext.synchronize()
    cnt = d_cnt.cpu().numpy()
    kps = _np_kps(d_kps, oracle)
    desc = d_desc.cpu().numpy()
    o = oracle.OrbOracle(600, 1.2, 8, 20, 7)
    for i in range(2):
        ok, od = o.extract(frames[i])
        assert cnt[i] == len(ok) and np.array_equal(kps[i, :cnt[i]], ok) and np.array_equal(desc[i, :cnt[i]], od)
    # host API with a strided numpy view
    view = padded[0, :, :W]
    assert view.strides[0] == S
    k, d = ext(view)
    ok, od = o.extract(frames[0])
    assert np.array_equal(k, ok) and np.array_equal(d, od)
    # capacity too small is refused up front
    with pytest.raises(oa.OrbError) as ei:
        ext.extract_batch_device(buf.data_ptr() + off, 2, W, H, S, S * H, d_kps.data_ptr(), d_desc.data_ptr(), 10,
                                 d_cnt.data_ptr(), None)
    assert ei.value.code == oa._lib.ORB_E_CAPACITY
    ext.close()


def test_measurement_hooks_report_every_stage():
    """orb_set_profiling / orb_get_stage_ms and orb_matcher_set_profiling / orb_matcher_get_stage_ms (what
    bench.py's roofline is built from): HIP events on the caller's stream, every stage a positive duration, the
    matcher's two phases adding up to its total"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import shard, synth
    B, W, H = 4, 640, 480
    L = oa._lib.lib()
    dev = torch.device("cuda", 0)
    ext = oa.ORBextractor(1000, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B)
    cap = ext.cap
    m = oa.ORBmatcher(0.9, True, max_keypoints=cap, max_pairs=B)
    d_img = torch.from_numpy(synth.make_stream(5, B, W, H)).to(dev)
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    pa_h, pb_h = shard.stream_pairs(0, 1, B)
    pa, pb = torch.from_numpy(pa_h).to(dev), torch.from_numpy(pb_h).to(dev)
    d_m12 = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros((B,), dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(st.cuda_stream)
    oa._lib.check(L.orb_set_profiling(ext._h, 1))
    oa._lib.check(L.orb_matcher_set_profiling(m._m, 1))
    for _ in range(3):
        ext.extract_batch_device(d_img.data_ptr(), B, W, H, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                 d_cnt.data_ptr(), sp)
        m.match_batch_device(d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, pa.data_ptr(), pb.data_ptr(), B,
                             (0.0, float(W), 0.0, float(H)), -1, d_m12.data_ptr(), d_nm.data_ptr(), sp)
    torch.cuda.synchronize()
    ms5 = (C.c_float * 5)()
    oa._lib.check(L.orb_get_stage_ms(ext._h, ms5))
    assert all(0.0 < float(v) < 50.0 for v in ms5), list(ms5)
    ms2 = (C.c_float * 2)()
    oa._lib.check(L.orb_matcher_get_stage_ms(m._m, ms2))
    assert 0.0 < ms2[0] < 50.0 and 0.0 < ms2[1] < 50.0
    # the ring was consumed by the call above: a second read has nothing to average
    assert L.orb_matcher_get_stage_ms(m._m, ms2) == oa._lib.ORB_E_INVALID
    for _ in range(2):
        m.match_batch_device(d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, pa.data_ptr(), pb.data_ptr(), B,
                             (0.0, float(W), 0.0, float(H)), -1, d_m12.data_ptr(), d_nm.data_ptr(), sp)
    tot = C.c_float(0)
    oa._lib.check(L.orb_matcher_get_ms(m._m, C.byref(tot)))
    assert 0.0 < tot.value < 50.0
    m.close()
    ext.close()


def test_config3_per_rank_share_64x1280x720(oracle):
    """BASELINE configs[3] at its per-GPU size: 512 frames of 1280x720 over 8 GPUs = 64 per rank, through
    orb_extract_batch_device + orb_match_batch_device (what a rank of bench.py runs), oracle parity on frames
    spread over the batch, and on the pairs they form with their successors; batch-position independence and
    determinism for every frame."""
    import hashlib
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import shard, synth
    B, W, H, nf = 64, 1280, 720, 1000
    frames = synth.make_stream(333, B, W, H)
    dev = torch.device("cuda", 0)
    ext = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=B)
    cap = ext.cap
    m = oa.ORBmatcher(0.9, True, max_keypoints=cap, max_pairs=B)
    d_img = torch.from_numpy(frames).to(dev)
    pa_h, pb_h = shard.stream_pairs(0, 1, B)
    pa, pb = torch.from_numpy(pa_h).to(dev), torch.from_numpy(pb_h).to(dev)
    st = torch.cuda.Stream(device=dev)
    sp = C.c_void_p(st.cuda_stream)
    bounds = (0.0, float(W), 0.0, float(H))
    runs = []
    for rep in range(2):
        d_kps = torch.full((B, cap, 28), 0xAB if rep else 0, dtype=torch.uint8, device=dev)
        d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
        d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
        d_m12 = torch.full((B, cap), -7, dtype=torch.int32, device=dev)
        d_nm = torch.zeros((B,), dtype=torch.int32, device=dev)
        with torch.cuda.stream(st):
            ext.extract_batch_device(d_img.data_ptr(), B, W, H, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                     d_cnt.data_ptr(), sp)
            m.match_batch_device(d_kps.data_ptr(), d_desc.data_ptr(), d_cnt.data_ptr(), cap, pa.data_ptr(), pb.data_ptr(),
                                 B, bounds, -1, d_m12.data_ptr(), d_nm.data_ptr(), sp)
        torch.cuda.synchronize()
        runs.append((d_cnt.cpu().numpy(), _np_kps(d_kps, oracle), d_desc.cpu().numpy(), d_m12.cpu().numpy(), d_nm.cpu().numpy()))
    cnt, kps, desc, m12, nm = runs[0]
    cnt2, kps2, desc2, m122, nm2 = runs[1]
    assert np.array_equal(cnt, cnt2) and np.array_equal(nm, nm2)
    for i in range(B):   # deterministic, and independent of what the output slabs held before
        assert np.array_equal(kps[i, :cnt[i]], kps2[i, :cnt[i]]) and np.array_equal(desc[i, :cnt[i]], desc2[i, :cnt[i]])
        assert np.array_equal(m12[i, :cnt[pa_h[i]]], m122[i, :cnt[pa_h[i]]])
        assert cnt[i] >= nf
    o = oracle.OrbOracle(nf, 1.2, 8, 20, 7)
    feats = {}
    for i in (0, 1, 21, 22, 42, 43, 62, 63):
        feats[i] = o.extract(frames[i])
        assert cnt[i] == len(feats[i][0])
        assert np.array_equal(kps[i, :cnt[i]], feats[i][0]) and np.array_equal(desc[i, :cnt[i]], feats[i][1]), i
    for p in (0, 21, 42, 62):
        a, b = int(pa_h[p]), int(pb_h[p])
        assert (a, b) == (p, p + 1)
        on, om12, _ = oracle.search_for_initialization(feats[a][0], feats[a][1], feats[b][0], feats[b][1], bounds, None,
                                                       -1, 0.9, True)
        assert nm[p] == on and on > 50 and np.array_equal(m12[p, :cnt[a]], om12), (p, nm[p], on)
    # the single-frame host path gives the same answer for frames the oracle did not see
    e1 = oa.ORBextractor(nf, 1.2, 8, 20, 7, max_w=W, max_h=H, max_batch=1)
    for i in (7, 33, 57):
        k1, d1 = e1(frames[i])
        assert np.array_equal(k1, kps[i, :cnt[i]]) and np.array_equal(d1, desc[i, :cnt[i]])
    e1.close()
    m.close()
    ext.close()
