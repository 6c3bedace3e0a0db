"""The checker's side of orbslam_in_practice_amd/sequence.py: the five calls of the tracking loop on the CPU oracle.
Test infrastructure (tests/ only: the package itself never imports oracle/)."""
import numpy as np


class OracleBackend:
    """the checker: oracle/ (CPU restatement of the reference's algorithm)"""
    name = "oracle"

    def __init__(self, nfeatures, w, h):
        from oracle import oracle as orc
        self.orc = orc
        self.ext = orc.OrbOracle(nfeatures, 1.2, 8, 20, 7)
        self.bounds = (0.0, float(w), 0.0, float(h))
        t = self.ext.tables()
        self.scale_factors = np.asarray(t["scale"], np.float32)
        self.inv_sigma2 = np.asarray(t["inv_sigma2"], np.float64)

    def extract(self, img):
        return self.ext.extract(img)

    def search_by_projection(self, cur, last, last_flags, last_xw, R, t, K, th):
        n, m, _ = self.orc.search_by_projection(cur[0], cur[1], last[0], last[1], self.bounds, last_flags, last_xw, R, t, K[0], K[1], K[2], K[3],
                                                self.scale_factors, np.zeros(len(cur[0]), np.uint8), th)
        return n, m

    def pose_optimization(self, pose, xw, obs, inv_sigma2, K):
        return self.orc.pose_optimization(pose, xw, obs, inv_sigma2, *K)

    def local_ba(self, window):
        return self.orc.ba_solve(window, rounds=(5, 10))

    def close(self):
        pass
