#!/usr/bin/env python3
"""orb_hamming_knn2 (brute-force 2-NN, host pointers): ms per call.  ORBSLAM_AMD_MATCH_NO_MFMA=1 = the xor/popcount scan"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa
from orbslam_in_practice_amd import synth
m = oa.ORBmatcher(0.9, True, max_keypoints=16384)
for n in (1000, 4000, 16000):
    d1, d2 = synth.random_descriptors(1, n), synth.random_descriptors(2, n)
    m.knn2(d1, d2)
    t0 = time.perf_counter()
    for _ in range(20):
        m.knn2(d1, d2)
    print("n1 = n2 = %d: %.3f ms per call (uploads and downloads included)" % (n, (time.perf_counter() - t0) / 20 * 1e3))
m.close()
