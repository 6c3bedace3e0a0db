#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite: a few minutes of oracle time): random frame sizes, feature
budgets, scale factors, pyramid depths and FAST thresholds through the extractor, random window / ratio settings
through the matcher, single-frame (latency mode) and batched launches, all compared with the CPU oracle bit for bit.
Usage: python tests/fuzz_parity.py [n_configs] [seed] [wild]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402


def run(n_cfg=30, seed=1, wild=False):
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(n_cfg):
        if wild:  # extreme shapes and budgets
            w, h = int(rng.integers(40, 2000)), int(rng.integers(40, 1100))
            nf = int(rng.choice([5, 30, 100, 700, 1000, 2500, 5000]))
        else:
            w, h = int(rng.integers(120, 1400)), int(rng.integers(100, 800))
            if w < 0.6 * h:
                w = h
            nf = int(rng.choice([100, 500, 1000, 2000, 3000]))
        scale = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nl = int(rng.integers(1, 9))
        ini, mn = int(rng.choice([12, 20, 30, 45])), int(rng.choice([3, 7, 10]))
        nb = int(rng.choice([1, 1, 2, 3, 12]))
        frames = np.stack([synth.make_frame(int(rng.integers(1 << 30)), w, h, n_shapes=max(20, w * h // int(rng.choice([60, 100, 400]))))
                           for _ in range(nb)])
        try:
            e = oa.ORBextractor(nf, scale, nl, ini, mn, max_w=w, max_h=h, max_batch=nb)
        except oa.OrbError as ex:
            print("cfg %d %dx%d nf %d scale %.1f nl %d: create refused (%s)" % (c, w, h, nf, scale, nl, ex))
            continue
        o = orc.OrbOracle(nf, scale, nl, ini, mn)
        got = e.extract_batch(frames) if nb > 1 else [e(frames[0])]
        feats = []
        for f, (k, d) in zip(frames, got):
            try:
                ok, od = o.extract(f)
            except RuntimeError as ex:  # a harness limit of the oracle wrapper, reported with the configuration that hit it
                print("ORACLE ERROR cfg %d %dx%d nf %d scale %.1f nl %d th %d/%d batch %d: %s" % (c, w, h, nf, scale, nl, ini, mn, nb, ex))
                raise
            same = np.array_equal(k, ok) and np.array_equal(d, od)
            bad += 0 if same else 1
            if not same:
                print("MISMATCH extract cfg %d %dx%d nf %d scale %.1f nl %d th %d/%d batch %d" % (c, w, h, nf, scale, nl, ini, mn, nb))
            feats.append((ok, od))
        e.close()
        if nb >= 2 and len(feats[0][0]) > 0:
            win = int(rng.choice([-1, 20, 100, 300]))
            ratio, ori = float(rng.choice([0.6, 0.8, 0.9])), bool(rng.integers(0, 2))
            (k1, d1), (k2, d2) = feats[0], feats[1]
            b = (0.0, float(w), 0.0, float(h))
            prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32).copy()
            on, om12, opm = orc.search_for_initialization(k1, d1, k2, d2, b, prev.copy(), win, ratio, ori)
            m = oa.ORBmatcher(ratio, ori, max_keypoints=max(len(k1), len(k2), 64))
            gn, gm12 = m.SearchForInitialization(oa.FrameFeatures(k1, d1, b), oa.FrameFeatures(k2, d2, b), prev, win)
            m.close()
            same = gn == on and np.array_equal(gm12, om12) and np.array_equal(prev, opm)
            bad += 0 if same else 1
            if not same:
                print("MISMATCH match cfg %d win %d ratio %.1f ori %d n1 %d n2 %d" % (c, win, ratio, ori, len(k1), len(k2)))
        print("cfg %2d %4dx%-4d nf %4d scale %.1f nl %d th %2d/%-2d batch %2d kp %s" % (c, w, h, nf, scale, nl, ini, mn, nb,
                                                                                  [len(k) for k, _ in got][:3]), flush=True)
    # matcher on synthetic features: keypoints of every octave in arbitrary order (not the extractor's level order),
    # positions partly outside the image bounds, descriptors clustered around a few prototypes (heavy contention)
    mbad = 0
    for c in range(n_cfg):
        n1, n2 = int(rng.integers(1, 2048)), int(rng.integers(1, 2048))
        nproto, flip = int(rng.integers(1, 60)), int(rng.integers(0, 12))
        proto = rng.integers(0, 256, (nproto, 32), dtype=np.uint8)

        def mk(n):
            kp = np.zeros(n, oa.KP_DTYPE)
            kp["x"], kp["y"] = rng.uniform(-20, 660, n), rng.uniform(-20, 500, n)
            kp["angle"] = rng.uniform(0, 360, n)
            kp["octave"] = rng.integers(0, 8, n) if rng.integers(0, 2) else np.sort(rng.integers(0, 8, n))
            d = proto[rng.integers(0, nproto, n)].copy()
            nf_ = rng.integers(0, flip + 1, n)
            for i in range(n):
                for b_ in rng.integers(0, 256, nf_[i]):
                    d[i, b_ >> 3] ^= np.uint8(1 << (b_ & 7))
            return kp, d
        (k1, d1), (k2, d2) = mk(n1), mk(n2)
        b = (0.0, 640.0, 0.0, 480.0)
        win = int(rng.choice([-1, 15, 100, 1000]))
        ratio, ori = float(rng.choice([0.6, 0.9, 1.0])), bool(rng.integers(0, 2))
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32) + rng.uniform(-8, 8, (n1, 2)).astype(np.float32)
        on, om12, opm = orc.search_for_initialization(k1, d1, k2, d2, b, prev.copy(), win, ratio, ori)
        m = oa.ORBmatcher(ratio, ori, max_keypoints=int(rng.choice([2048, 4096])))
        gp = prev.copy()
        gn, gm12 = m.SearchForInitialization(oa.FrameFeatures(k1, d1, b), oa.FrameFeatures(k2, d2, b), gp, win)
        m.close()
        same = gn == on and np.array_equal(gm12, om12) and np.array_equal(gp, opm)
        mbad += 0 if same else 1
        if not same:
            print("MISMATCH synthetic match cfg %d n1 %d n2 %d protos %d flip %d win %d ratio %.1f ori %d: %d vs %d" % (
                c, n1, n2, nproto, flip, win, ratio, ori, gn, on))
    print("FUZZ-MATCH %s: %d mismatches in %d configs" % ("FAILED" if mbad else "OK", mbad, n_cfg))
    bad += mbad
    # one handle, a random sequence of calls: frame sizes and batch sizes change from call to call, single frames repeat
    # (HIP-graph capture, replay and invalidation; staging re-growth; geometry switches; latency / batch quadtree)
    sbad = 0
    e = oa.ORBextractor(600, 1.2, 6, 20, 7, max_w=320, max_h=240, max_batch=6)
    o = orc.OrbOracle(600, 1.2, 6, 20, 7)
    shapes = [(320, 240), (400, 300), (256, 256), (640, 200), (200, 180)]
    for c in range(2 * n_cfg):
        w, h = shapes[int(rng.integers(0, len(shapes)))]
        nb = int(rng.choice([1, 1, 1, 2, 6, 9]))
        frames = np.stack([synth.make_frame(int(rng.integers(1 << 30)), w, h, n_shapes=w * h // 90) for _ in range(nb)])
        if nb == 1:
            got = [e(frames[0])]
        elif nb <= 6:
            got = e.extract_batch(frames)
        else:  # larger than max_batch: refused, handle stays usable
            try:
                e.extract_batch(frames)
                got = None
            except oa.OrbError as ex:
                got = [] if ex.code == oa._lib.ORB_E_CAPACITY else None
            if got is None:
                sbad += 1
                print("MISMATCH sequence call %d: batch %d > max_batch was not refused with ORB_E_CAPACITY" % (c, nb))
            continue
        for f, (k, d) in zip(frames, got):
            try:
                ok, od = o.extract(f)
            except RuntimeError as ex:  # a harness limit of the oracle wrapper, reported with the configuration that hit it
                print("ORACLE ERROR cfg %d %dx%d nf %d scale %.1f nl %d th %d/%d batch %d: %s" % (c, w, h, nf, scale, nl, ini, mn, nb, ex))
                raise
            if not (np.array_equal(k, ok) and np.array_equal(d, od)):
                sbad += 1
                print("MISMATCH sequence call %d: %dx%d batch %d" % (c, w, h, nb))
    e.close()
    print("FUZZ-SEQUENCE %s: %d mismatches in %d calls" % ("FAILED" if sbad else "OK", sbad, 2 * n_cfg))
    bad += sbad
    print("FUZZ %s: %d mismatches in %d configs" % ("FAILED" if bad else "OK", bad, n_cfg))
    return 1 if bad else 0


def main():
    return run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
               len(sys.argv) > 3 and sys.argv[3] == "wild")


if __name__ == "__main__":
    sys.exit(main())
