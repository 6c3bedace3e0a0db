"""GPU parity: device bundle adjustment (in-kernel blocked Cholesky on the Schur complement; PCG kept
behind ORBSLAM_AMD_BA_SOLVER=pcg; C-ABI) vs the CPU oracle
(dense Cholesky, g2o semantics).  Tolerance from BASELINE.json north_star: pose / point estimates
within 1e-5 relative on identical inputs (f64 on both sides)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-5


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def opt():
    import orbslam_in_practice_amd as oa
    o = oa.Optimizer(max_cams=128, max_points=8192, max_obs=65536, max_batch=8)
    yield o
    o.close()


@pytest.mark.parametrize("nc,npt,rounds,nfix", [(8, 150, (5, 10), 1), (50, 2000, (5, 10), 1), (50, 2000, (20,), 2),
                                                 (100, 5000, (5, 10), 1), (100, 5000, (5, 10), 3)])
def test_ba_matches_oracle(opt, oracle, nc, npt, rounds, nfix):
    """nfix = 1 is the reference's graph (only KF 0 fixed, src/Optimizer.cpp:52-53: the scale gauge stays
    free); nfix >= 2 pins the gauge like upstream's local BA (fixed covisible observers)."""
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(nc * 7 + npt, nc, npt)
    w["fixed"][:nfix] = 1
    ref = oracle.ba_solve(w, rounds=rounds)
    got = opt.solve_batch([w], rounds)[0]
    # once converged to machine precision the sign of rho is rounding noise, so the number of LM trials may differ by
    # a few between the two solvers -- and with it the iteration at which LM gives up (ten rejected trials in a row:
    # the 8 x 150 window ends that way on the phase-kernel path); the estimates may not
    if got["iterations"] != ref["iterations"]:
        assert abs(got["iterations"] - ref["iterations"]) <= 1
        assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-9 * ref["chi2_final"]
    assert abs(got["lin_solves"] - ref["lin_solves"]) <= 10
    assert np.array_equal(got["outlier"], ref["outlier"])
    assert _rel(got["points"], ref["points"]) < REL
    assert _rel(got["poses"][:, 4:], ref["poses"][:, 4:]) < REL
    assert np.abs(got["poses"][:, :4] - ref["poses"][:, :4]).max() < REL
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]
    assert abs(got["chi2_initial"] - ref["chi2_initial"]) <= 1e-9 * ref["chi2_initial"]
    assert got["chi2_final"] < got["chi2_initial"]
    if len(rounds) == 2:  # outliers removed after round 0: the cost collapses
        assert got["chi2_final"] < 0.2 * got["chi2_initial"]


def _check_estimates(got, ref, tol=REL, converged_ok=False):
    if converged_ok and got["iterations"] != ref["iterations"]:
        # LM stops early when a step no longer lowers the cost; AT the minimum the sign of the gain ratio is rounding
        # noise, so the stopping iteration is too (the oracle itself stops elsewhere when its input is perturbed by
        # 1e-13) -- legitimate only if both sides have converged to the same cost
        assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-9 * ref["chi2_final"]
    else:
        assert got["iterations"] == ref["iterations"]
    assert np.array_equal(got["outlier"], ref["outlier"])
    assert _rel(got["points"], ref["points"]) < tol
    assert _rel(got["poses"][:, 4:], ref["poses"][:, 4:]) < tol
    assert np.abs(got["poses"][:, :4] - ref["poses"][:, :4]).max() < tol
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= tol * ref["chi2_final"]


# (cameras, points, gross-outlier fraction, iterations, seed).  Every case is WELL-POSED at the tolerance: the oracle run
# on the same window with its points scaled by (1 + 1e-13) reproduces its own estimates to better than 1e-7 (measured;
# tools/ba_sensitivity.py prints the table).  Windows where that self-agreement is worse than the tolerance -- gross
# outliers that the never-switched-off Huber edges drag to |X| ~ 70 in a two-view map, for one -- cannot pin any
# implementation to 1e-5 and are covered by the sensitivity-bounded test below instead.
@pytest.mark.parametrize("nc,npt,outl,its,seed", [(2, 300, 0.05, 20, 8),      # src/Tracking.cpp:254: the initial two-keyframe map
                                                  (2, 300, 0.0, 20, 101),
                                                  (3, 500, 0.05, 10, 102),
                                                  (50, 2000, 0.0, 20, 149),   # BASELINE configs[2] size, inlier noise only
                                                  (50, 2000, 0.05, 12, 149),  # ... with 5 % gross outliers: the well-posed prefix
                                                  (100, 5000, 0.0, 20, 199)])
def test_ba_global_reference_graph(opt, oracle, nc, npt, outl, its, seed):
    """a13 exactly as the reference builds it (src/Optimizer.cpp:43-58, :100-105, :124): ONLY keyframe 0 fixed (the
    monocular scale gauge stays free), robust kernel on every edge, one optimize(nIterations), outliers never
    removed.  Pose and point estimates within 1e-5 relative of the oracle, f64 on both sides."""
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(seed, nc, npt, outlier_frac=outl)
    assert w["fixed"].sum() == 1 and w["fixed"][0] == 1
    ref = oracle.ba_solve(w, rounds=(its,))
    got = opt.solve_batch([w], (its,))[0]
    _check_estimates(got, ref, converged_ok=(outl == 0.0))


def _residuals(r, w):
    q = r["poses"][:, :4]
    qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                  2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                  2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)], axis=1).reshape(-1, 3, 3)
    pc = np.einsum("eij,ej->ei", R[w["cam_idx"]], r["points"][w["pt_idx"]]) + r["poses"][w["cam_idx"], 4:]
    return np.stack([w["obs"][:, 0] - (w["fx"] * pc[:, 0] / pc[:, 2] + w["cx"]),
                     w["obs"][:, 1] - (w["fy"] * pc[:, 1] / pc[:, 2] + w["cy"])], axis=1)


def test_ba_global_reference_graph_outliers_20_iterations(opt, oracle):
    """The same graph at 50 x 2000 with 5 % gross outliers and the full 20 iterations.  Here the PROBLEM is unstable,
    not a solver: the oracle run twice, the second time with its input points scaled by (1 + 1e-13), agrees with itself
    to 1e-11 through iteration 12 and then parts ways by three orders of magnitude every few iterations (lambda has
    fallen to ~1e-10 of the Hessian diagonal, the Huber-weighted outliers are never switched off and the cost keeps
    creeping down a flat valley) -- 3e-3 in the points at iteration 20.  No two f64 implementations can agree to 1e-5
    there (g2o built with another compiler would not either), so the test (a) holds the device to 1e-5 on the
    well-conditioned prefix (parametrised case above), (b) measures the problem's own sensitivity with the oracle and
    requires the device to stay within a small multiple of it, and (c) bounds the gauge-invariant cost by 1e-4."""
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(50 * 7 + 2000, 50, 2000)
    ref = oracle.ba_solve(w, rounds=(20,))
    wp = dict(w)
    wp["points"] = w["points"] * (1 + 1e-13)
    alt = oracle.ba_solve(wp, rounds=(20,))
    got = opt.solve_batch([w], (20,))[0]
    assert got["iterations"] == ref["iterations"] == 20
    K = 50   # rounding differences of the two implementations are not the 1e-13 of the probe; same order of magnitude
    own_c = max(abs(alt["chi2_final"] - ref["chi2_final"]) / ref["chi2_final"], REL / K)
    assert abs(got["chi2_final"] - ref["chi2_final"]) < K * own_c * ref["chi2_final"]
    assert abs(got["chi2_final"] - ref["chi2_final"]) < 1e-4 * ref["chi2_final"]
    own = max(_rel(alt["points"], ref["points"]), REL)
    assert _rel(got["points"], ref["points"]) < K * own
    r_ref, r_alt, r_got = _residuals(ref, w), _residuals(alt, w), _residuals(got, w)
    inl = np.abs(r_ref).max(axis=1) < 20.0
    own_r = max(np.abs(r_alt - r_ref)[inl].max(), REL * np.abs(r_ref[inl]).max())
    assert np.abs(r_got - r_ref)[inl].max() < K * own_r
    import warnings   # a tolerance wider than 1e-5 is a carve-out: counted and shown (VERDICT r2 #1c)
    warnings.warn("BA carve-out (K x own sensitivity, 1 window): device vs oracle rel(points) %.2e, oracle vs itself at 1e-13 %.2e, "
                  "bound %.2e" % (_rel(got["points"], ref["points"]), _rel(alt["points"], ref["points"]), K * own))


def test_ba_degenerate_windows(opt, oracle):
    """all cameras fixed (no reduced system: structure-only refinement), no observations at all, and a
    window whose points each have a single observation (Hll singular until lambda is added)"""
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(8, 6, 200)
    wf = dict(w)
    wf["fixed"] = np.ones(6, np.uint8)
    ref = oracle.ba_solve(wf, rounds=(5, 10))
    got = opt.solve_batch([wf], (5, 10))[0]
    _check_estimates(got, ref, converged_ok=True)   # structure-only refinement converges fully within the schedule
    assert np.array_equal(got["poses"], w["poses"] * np.sign(w["poses"][:, :1]))
    assert got["chi2_final"] < got["chi2_initial"]
    we = dict(w)
    for k in ("obs", "cam_idx", "pt_idx", "inv_sigma2"):
        we[k] = w[k][:0]
    got = opt.solve_batch([we], (5, 10))[0]
    assert got["chi2_final"] == 0.0 and np.array_equal(got["points"], w["points"])
    first = np.unique(w["pt_idx"], return_index=True)[1]
    w1 = dict(w)
    for k in ("obs", "cam_idx", "pt_idx", "inv_sigma2"):
        w1[k] = w[k][first]
    ref = oracle.ba_solve(w1, rounds=(5,))
    got = opt.solve_batch([w1], (5,))[0]
    assert np.isfinite(got["points"]).all() and np.isfinite(got["poses"]).all()
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * max(ref["chi2_final"], 1.0)


def test_ba_batch_equals_single(opt, oracle):
    from orbslam_in_practice_amd import synth
    ws = [synth.make_ba_window(100 + i, 12 + i, 300 + 40 * i) for i in range(5)]
    batch = opt.solve_batch(ws, (5, 10))
    for w, b in zip(ws, batch):
        single = opt.solve_batch([w], (5, 10))[0]
        # no atomics anywhere: bit-reproducible for a given launch shape; a window solved alone runs on a team of
        # workgroups, the batch (four windows or more) as phase kernels -- another summation order and W rebuilt from its
        # edge records: rounding level only, far below the 1e-5 both are held to
        assert _rel(single["points"], b["points"]) < 1e-7
        assert np.array_equal(single["outlier"], b["outlier"])
        ref = oracle.ba_solve(w, rounds=(5, 10))
        assert _rel(b["points"], ref["points"]) < REL


def test_ba_non_robust_and_fixed_cams(opt, oracle):
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(77, 20, 500, outlier_frac=0.0)
    w["fixed"][:3] = 1  # upstream local BA: several fixed observers
    ref = oracle.ba_solve(w, rounds=(10,), robust=False)
    got = opt.solve_batch([w], (10,), robust=False)[0]
    assert _rel(got["points"], ref["points"]) < REL
    assert np.array_equal(got["poses"][:3], w["poses"][:3] * np.sign(w["poses"][:3, :1]))
    assert got["chi2_final"] < 0.25 * got["chi2_initial"]  # no outliers: converges to the noise floor
    assert got["chi2_final"] < 2.5 * len(w["obs"])          # ~2 dof per observation at unit-variance noise


def test_ba_stop_flag(opt):
    from orbslam_in_practice_amd import synth
    import orbslam_in_practice_amd as oa
    w = synth.make_ba_window(5, 10, 200)
    stop = np.ones(1, np.uint8)
    with pytest.raises(oa.OrbError) as ei:
        opt.solve_batch([w], (5, 10), stop=stop)
    assert ei.value.code == oa._lib.ORB_E_STOPPED


def test_pose_optimization(opt, oracle):
    from orbslam_in_practice_amd import synth
    rng = np.random.default_rng(3)
    w = synth.make_ba_window(11, 6, 400, outlier_frac=0.1)
    c = 3
    sel = w["cam_idx"] == c
    xw = w["gt_points"][w["pt_idx"][sel]]
    obs, info = w["obs"][sel], w["inv_sigma2"][sel]
    pose0 = w["poses"][c]
    rp, ro, rn = oracle.pose_optimization(pose0, xw, obs, info, w["fx"], w["fy"], w["cx"], w["cy"])
    gp, go, gn = opt.PoseOptimization(pose0, xw, obs, info, w["fx"], w["fy"], w["cx"], w["cy"])
    assert gn == rn and np.array_equal(go, ro)
    assert np.abs(gp - rp).max() < REL * np.abs(rp).max()
    assert 0 < gn < len(xw)
    # fewer than 3 correspondences: returns 0 and leaves the pose alone (src/Optimizer.cpp:206-207)
    gp2, _, gn2 = opt.PoseOptimization(pose0, xw[:2], obs[:2], info[:2], w["fx"], w["fy"], w["cx"], w["cy"])
    assert gn2 == 0 and np.array_equal(gp2, pose0)


def test_pose_optimization_batch(opt, oracle):
    """N3: several frames' pose problems in one launch equal the oracle one by one (ragged sizes, one tiny)"""
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(21, 8, 500, outlier_frac=0.08)
    poses, xws, obss, infos = [], [], [], []
    for c in range(1, 7):
        sel = w["cam_idx"] == c
        if c == 4:
            sel = sel & (np.cumsum(sel) <= 2)      # fewer than 3 correspondences
        poses.append(w["poses"][c])
        xws.append(w["gt_points"][w["pt_idx"][sel]])
        obss.append(w["obs"][sel])
        infos.append(w["inv_sigma2"][sel])
    gp, go, gn = opt.PoseOptimizationBatch(np.array(poses), xws, obss, infos, w["fx"], w["fy"], w["cx"], w["cy"])
    for i in range(6):
        rp, ro, rn = oracle.pose_optimization(poses[i], xws[i], obss[i], infos[i], w["fx"], w["fy"], w["cx"], w["cy"])
        assert gn[i] == rn and np.array_equal(go[i], ro[:len(go[i])])
        assert np.abs(gp[i] - rp).max() < REL * max(np.abs(rp).max(), 1.0)
    assert gn[3] == 0 and np.array_equal(gp[3], poses[3])


def test_ba_cholesky_equals_pcg_many_sizes(oracle):
    """the two in-kernel linear solvers (blocked Cholesky with f64 MFMA, block-Jacobi PCG) on 48 ragged
    windows -- every reduced-system size 6..150 pads differently into the 16-wide Cholesky blocks"""
    import os
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    ws = [synth.make_ba_window(4000 + i, 3 + (i % 24), 60 + 17 * i) for i in range(48)]
    for w in ws:
        w["fixed"][:2] = 1
    chol = oa.Optimizer(max_cams=32, max_points=1024, max_obs=16384, max_batch=48)
    os.environ["ORBSLAM_AMD_BA_SOLVER"] = "pcg"
    try:
        pcg = oa.Optimizer(max_cams=32, max_points=1024, max_obs=16384, max_batch=48)
    finally:
        del os.environ["ORBSLAM_AMD_BA_SOLVER"]
    a = chol.solve_batch(ws, (5, 10))
    b = pcg.solve_batch(ws, (5, 10))
    for i, (x, y, w) in enumerate(zip(a, b, ws)):
        assert np.array_equal(x["outlier"], y["outlier"]), i
        assert _rel(x["points"], y["points"]) < 1e-7, i
        assert _rel(x["poses"], y["poses"]) < 1e-7, i
    ref = oracle.ba_solve(ws[17], rounds=(5, 10))
    assert _rel(a[17]["points"], ref["points"]) < REL
    chol.close()
    pcg.close()


@pytest.mark.parametrize("nc,npt,seed", [(24, 600, 1), (60, 2500, 2), (100, 5000, 3)])
def test_ba_shuffled_camera_order(opt, oracle, nc, npt, seed):
    """The blocked Cholesky works inside the envelope of S in the caller's camera order.  Shuffling the cameras
    of a chain-like window turns the band into an irregular pattern (tile rows whose first non-zero column is
    NOT monotone, envelope close to dense): same estimates as the oracle, camera by camera."""
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(7000 + seed, nc, npt)
    rng = np.random.default_rng(seed)
    perm = np.concatenate([[0], 1 + rng.permutation(nc - 1)])      # new position -> old camera (KF 0 stays first)
    inv = np.empty(nc, np.int64)
    inv[perm] = np.arange(nc)
    ws = dict(w)
    ws["poses"] = w["poses"][perm].copy()
    ws["fixed"] = w["fixed"][perm].copy()
    ws["fixed"][:2] = 1
    ws["cam_idx"] = inv[w["cam_idx"]].astype(np.int32)
    for k in ("gt_poses",):
        if k in w:
            ws[k] = w[k][perm].copy()
    ref = oracle.ba_solve(ws, rounds=(5, 10))
    got = opt.solve_batch([ws], (5, 10))[0]
    assert got["iterations"] == ref["iterations"]
    assert np.array_equal(got["outlier"], ref["outlier"])
    assert _rel(got["points"], ref["points"]) < REL
    assert _rel(got["poses"][:, 4:], ref["poses"][:, 4:]) < REL
    assert abs(got["chi2_final"] - ref["chi2_final"]) <= 1e-6 * ref["chi2_final"]


def test_ba_many_cameras_envelope_panel(oracle):
    """160 free cameras: the dense reduced system is 960 x 960, its full-height LDS panel (130 KB) would not fit, the
    envelope-sized one (a band of a few tiles for a chain-like window) does -- the blocked Cholesky still runs and
    still agrees with the oracle"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(4242, 162, 4000)
    w["fixed"][:2] = 1
    o = oa.Optimizer(max_cams=170, max_points=4096, max_obs=40000, max_batch=1)
    ref = oracle.ba_solve(w, rounds=(5, 5))
    got = o.solve_batch([w], (5, 5))[0]
    assert got["iterations"] == ref["iterations"]
    assert np.array_equal(got["outlier"], ref["outlier"])
    assert _rel(got["points"], ref["points"]) < REL
    assert _rel(got["poses"][:, 4:], ref["poses"][:, 4:]) < REL
    o.close()


@pytest.mark.parametrize("nc,npt", [(12, 400), (100, 5000)])
def test_ba_team_sizes_agree(oracle, nc, npt, monkeypatch):
    """one window spread over 1, 4 and 64 cooperating workgroups (device-side phase barriers): same estimates to
    rounding, each within 1e-5 of the oracle, and the profiling hook reports the team that ran"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(2024 + nc, nc, npt)
    w["fixed"][:2] = 1
    ref = oracle.ba_solve(w, rounds=(5, 10))
    outs = {}
    for g in (1, 4, 64):
        monkeypatch.setenv("ORBSLAM_AMD_BA_TEAM", str(g))
        o = oa.Optimizer(max_cams=nc, max_points=npt, max_obs=8 * npt, max_batch=1)
        o.set_profiling(True)
        outs[g] = o.solve_batch([w], (5, 10))[0]
        ms, launches, team = o.iteration_ms()
        assert team == g and launches == 2 and ms > 0     # one launch per optimize() round (5 + 10 iterations)
        o.close()
        _check_estimates(outs[g], ref)
    for g in (4, 64):
        assert np.array_equal(outs[g]["outlier"], outs[1]["outlier"])
        assert _rel(outs[g]["points"], outs[1]["points"]) < 1e-6   # another summation order (point chunks vs block lists)
        assert _rel(outs[g]["poses"], outs[1]["poses"]) < 1e-6


def test_ba_submit_wait_pipeline(opt, oracle):
    """ba_submit_batch / ba_wait: three batches in flight, results in submission order and equal to the synchronous
    call; a fourth submit without a wait is refused"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    batches = [[synth.make_ba_window(5000 + 10 * b + i, 10 + i, 250 + 30 * i) for i in range(4)] for b in range(4)]
    sync = [opt.solve_batch(ws, (5, 10)) for ws in batches]
    opt.submit_batch(batches[0], (5, 10))
    opt.submit_batch(batches[1], (5, 10))
    opt.submit_batch(batches[2], (5, 10))
    with pytest.raises(oa.OrbError) as ei:
        opt.submit_batch(batches[3], (5, 10))
    assert ei.value.code == oa._lib.ORB_E_BUSY == -8 and len(opt._jobs) == 3   # the refused batch never reached the device
    with pytest.raises(RuntimeError):
        opt.solve_batch(batches[3], (5, 10))       # would drain the queue inside C and orphan the jobs held here
    assert len(opt._jobs) == 3
    got0 = opt.wait()
    opt.submit_batch(batches[3], (5, 10))          # the freed slot is the oldest one: slots are reused in ring order
    got1 = opt.wait()
    got2 = opt.wait()
    got3 = opt.wait()
    for a, b in zip(got3, sync[3]):
        assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["poses"], b["poses"])
    for got, ref in zip((got0, got1, got2), sync):
        for a, b in zip(got, ref):
            assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["poses"], b["poses"])
            assert np.array_equal(a["outlier"], b["outlier"]) and a["chi2_final"] == b["chi2_final"]
    r = oracle.ba_solve(batches[2][3], rounds=(5, 10))
    assert _rel(got2[3]["points"], r["points"]) < REL
    assert opt._jobs == []
    with pytest.raises(RuntimeError):
        opt.wait()                                 # nothing submitted: a clear error, not an IndexError
    # a C caller that fills only rounds[0 .. n_rounds): garbage past n_rounds must not be compared
    job = opt._prepare(batches[0], (5, 10), True, 5.991, 0, None)
    for i in range(job["n"]):
        job["Ps"][i].rounds[2], job["Ps"][i].rounds[3] = 1000 + i, -i
    oa._lib.check(opt._L.ba_solve_batch(opt._s, job["Ps"], job["Rs"], job["n"]))
    for a, b in zip(opt._collect(job), sync[0]):
        assert np.array_equal(a["points"], b["points"]) and np.array_equal(a["poses"], b["poses"])


def test_ba_more_than_256_free_cameras(oracle):
    """g2o has no camera limit.  270 free cameras (reduced system 1620 > the 1536 rows of the LDS vectors): a
    chain-like window still fits the envelope Cholesky; the same window with its cameras shuffled has a nearly dense
    envelope, no LDS panel fits, and the in-kernel PCG runs with its vectors in HBM.  Both agree with the oracle."""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    w = synth.make_ba_window(4243, 272, 3500)
    w["fixed"][:2] = 1
    ref = oracle.ba_solve(w, rounds=(3, 2))
    o = oa.Optimizer(max_cams=272, max_points=3500, max_obs=32768, max_batch=1)
    got = o.solve_batch([w], (3, 2))[0]
    _check_estimates(got, ref)
    rng = np.random.default_rng(9)
    perm = np.concatenate([[0, 1], 2 + rng.permutation(270)])
    inv = np.empty(272, np.int64)
    inv[perm] = np.arange(272)
    ws = dict(w)
    ws["poses"], ws["fixed"] = w["poses"][perm].copy(), w["fixed"][perm].copy()
    ws["cam_idx"] = inv[w["cam_idx"]].astype(np.int32)
    got2 = o.solve_batch([ws], (3, 2))[0]
    assert got2["iterations"] == ref["iterations"]
    assert np.array_equal(got2["outlier"], ref["outlier"])
    assert _rel(got2["points"], ref["points"]) < REL
    assert _rel(got2["poses"][inv][:, 4:], ref["poses"][:, 4:]) < REL
    o.close()


def test_ba_point_seen_by_every_camera(oracle, monkeypatch):
    """two points per window observed by all 100 keyframes: 4851 contributions each to the reduced system, every one to
    another block -- nineteen times what a point chunk of the LDS-staged Schur pass holds.  Long tracks are what the
    per-block lists are for (ba_window_chunkable): in a batch such windows run as phase kernels with the list pass in
    place of the chunked one, alone on a team, and the fused kernel with one workgroup per window takes the list path
    too.  All agree with the oracle."""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    ws = []
    for seed in (61, 62, 63, 64):
        w = synth.make_ba_window(seed, 100, 600)
        w["fixed"][:2] = 1
        q, t = w["gt_poses"][:, :4], w["gt_poses"][:, 4:]
        rng = np.random.default_rng(seed)
        for pid in (0, 7):          # two such points per window
            X = w["gt_points"][pid]
            have = set(w["cam_idx"][w["pt_idx"] == pid].tolist())
            add_c, add_o = [], []
            for c in range(100):
                if c in have:
                    continue
                qw, qx, qy, qz = q[c]
                R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                              [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                              [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
                pc = R @ X + t[c]
                if pc[2] <= 0.1:
                    continue
                add_c.append(c)
                add_o.append([w["fx"] * pc[0] / pc[2] + w["cx"] + rng.normal(0, 1), w["fy"] * pc[1] / pc[2] + w["cy"] + rng.normal(0, 1)])
            w["cam_idx"] = np.concatenate([w["cam_idx"], np.array(add_c, np.int32)])
            w["pt_idx"] = np.concatenate([w["pt_idx"], np.full(len(add_c), pid, np.int32)])
            w["obs"] = np.concatenate([w["obs"], np.array(add_o).reshape(-1, 2)])
            w["inv_sigma2"] = np.concatenate([w["inv_sigma2"], np.ones(len(add_c))])
            assert (w["pt_idx"] == pid).sum() > 91
        ws.append(w)
    refs = [oracle.ba_solve(w, rounds=(5, 10)) for w in ws]
    for force_team, want_team, sub in ((None, 0, ws), ("1", 1, ws[:2]), (None, None, ws[:1])):
        if force_team is None:
            monkeypatch.delenv("ORBSLAM_AMD_BA_TEAM", raising=False)
        else:
            monkeypatch.setenv("ORBSLAM_AMD_BA_TEAM", force_team)
        o = oa.Optimizer(max_cams=100, max_points=600, max_obs=8192, max_batch=4)
        o.set_profiling(True)
        got = o.solve_batch(sub, (5, 10))
        team = o.iteration_ms()[2]
        o.close()
        if want_team is not None:
            assert team == want_team          # 0: phase kernels with the list pass (four windows); 1: fused kernel, one workgroup per window
        else:
            assert team > 1                   # one window: a team
        for g, r in zip(got, refs):
            _check_estimates(g, r)


@pytest.mark.parametrize("spare", [None, "-100", "0"])
def test_ba_phase_kernel_batch_path(oracle, monkeypatch, spare):
    """a batch of more than 128 windows runs as phase kernels (an LM trial = ten kernels over all windows, the windows'
    positions in g2o's control flow in their state records, W rebuilt from 32-byte edge records, two-part Schur pass):
    every window within 1e-5 of the oracle -- with the default number of enqueued trials, with none (everything left to
    the per-window finishing kernel) and with exactly one per iteration (windows with a rejected trial finish there)"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    if spare is not None:
        monkeypatch.setenv("ORBSLAM_AMD_BA_SPARE", spare)
    shapes = [(8, 150), (12, 400), (30, 900), (20, 300)]
    ws = []
    for i in range(136):
        nc, npt = shapes[i % len(shapes)]
        w = synth.make_ba_window(7000 + i, nc, npt)
        w["fixed"][:1 + i % 3] = 1
        ws.append(w)
    o = oa.Optimizer(max_cams=32, max_points=1024, max_obs=16384, max_batch=len(ws))
    o.set_profiling(True)
    got = o.solve_batch(ws, (5, 10))
    ms, launches, team = o.iteration_ms()
    assert team == 0 and launches == 2 and ms > 0        # phase kernels; one timed interval per optimize() round
    o.close()
    for i, (w, g) in enumerate(zip(ws, got)):
        if i % 4 == 0 or i < 8:                           # (the oracle takes its time: a sample of the batch)
            _check_estimates(g, oracle.ba_solve(w, rounds=(5, 10)), converged_ok=True)
    # the same windows one at a time (workgroup teams, fused kernel): same estimates to rounding
    o1 = oa.Optimizer(max_cams=32, max_points=1024, max_obs=16384, max_batch=1)
    for i in (0, 1, 2, 3, 77):
        s1 = o1.solve_batch([ws[i]], (5, 10))[0]
        assert np.array_equal(s1["outlier"], got[i]["outlier"])
        assert _rel(s1["points"], got[i]["points"]) < 1e-6
        assert _rel(s1["poses"], got[i]["poses"]) < 1e-6
    o1.close()


def test_ba_mixed_batch_of_chunk_and_list_windows(oracle):
    """one batch through the phase kernels with both kinds of windows: short tracks (point chunks, 32-byte edge records, two-part
    Schur pass) and long tracks (per-block lists, the list pass as a phase kernel): every window within 1e-5 of the oracle"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    ws = []
    for i in range(4):
        w = synth.make_ba_window(8100 + i, 20, 400)
        w["fixed"][:2] = 1
        ws.append(w)
        ws.append(synth.make_ba_window_long_tracks(8200 + i, 20, 400, 16 if i % 2 else 26))
    o = oa.Optimizer(max_cams=20, max_points=400, max_obs=16384, max_batch=len(ws))
    o.set_profiling(True)
    got = o.solve_batch(ws, (5, 10))
    assert o.iteration_ms()[2] == 0          # phase kernels
    o.close()
    for w, g in zip(ws, got):
        _check_estimates(g, oracle.ba_solve(w, rounds=(5, 10)), converged_ok=True)


@pytest.mark.parametrize("nc,npt,nwin", [(100, 5000, 8), (50, 2000, 8)])
def test_ba_bench_launch_shape_every_window_vs_oracle(oracle, nc, npt, nwin):
    """VERDICT r2 #1a / ADVICE r2: the launch shape bench.py TIMES -- a batch of 100 x 5000 windows (north_star target)
    or 50 x 2000 windows (BASELINE configs[2]) as phase kernels with point chunks (k_ba_ph_*) -- held to the oracle at
    1e-5 for EVERY window of the batch, through the synchronous call and again through ba_submit_batch / ba_wait (two
    batches in flight, as the bench streams them).  src/Optimizer.cpp:23-129 graph, a14 schedule (5 + 10)."""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    ws = [synth.make_ba_window(900 + i, nc, npt) for i in range(nwin)]       # the seeds bench.py's rank 0 uses
    refs = [oracle.ba_solve(w, rounds=(5, 10)) for w in ws]
    E = max(len(w["obs"]) for w in ws)
    o = oa.Optimizer(max_cams=nc, max_points=npt, max_obs=E, max_batch=nwin)
    o.set_profiling(True)
    got = o.solve_batch(ws, (5, 10))
    ms, launches, team = o.iteration_ms()
    assert team == 0 and launches == 2 and ms > 0      # phase kernels, one timed interval per optimize() round
    o.set_profiling(False)
    for g, r in zip(got, refs):
        _check_estimates(g, r)
    # the streamed form: two batches in flight, the second in reverse order (results must not depend on the slot, the
    # position in the batch or what else is in flight)
    o.submit_batch(ws, (5, 10))
    o.submit_batch(ws[::-1], (5, 10))
    a = o.wait()
    b = o.wait()
    o.close()
    for i in range(nwin):
        _check_estimates(a[i], refs[i])
        _check_estimates(b[nwin - 1 - i], refs[i])
        for k in ("points", "poses", "outlier"):
            assert np.array_equal(a[i][k], got[i][k]) and np.array_equal(b[nwin - 1 - i][k], got[i][k])


@pytest.mark.parametrize("variant", ["nt256", "nt512", "round2_solver"])
def test_ba_cholesky_variants_agree_with_the_oracle(oracle, monkeypatch, variant):
    """the reduced camera system has three solvers on the phase-kernel path -- the band Cholesky with the LDS-resident
    window at 256 threads per window (two windows per CU: batches that fill the device twice) and at 512, and round 2's
    blocked Cholesky on the dense matrix in memory (ORBSLAM_AMD_BA_CHOL_BAND=0, also what a band wider than 8 tile rows
    falls back to) -- each held to the oracle at 1e-5 on narrow bands (M = 4: short tracks), wider ones (long tracks: the
    MAXM = 8 instance) and a shuffled camera order (an irregular envelope, beyond the band kernel: the fallback)"""
    import orbslam_in_practice_amd as oa
    from orbslam_in_practice_amd import synth
    if variant == "round2_solver":
        monkeypatch.setenv("ORBSLAM_AMD_BA_CHOL_BAND", "0")
    else:
        monkeypatch.setenv("ORBSLAM_AMD_BA_CHOL_NT", variant[2:])
    ws = []
    for i in range(3):
        w = synth.make_ba_window(8800 + i, 40, 900)
        w["fixed"][:1 + i] = 1
        ws.append(w)
    for i, track in enumerate((10, 14, 18)):
        ws.append(synth.make_ba_window_long_tracks(8850 + i, 24, 500, track))
    rng = np.random.default_rng(5)
    w = synth.make_ba_window(8890, 30, 700)
    perm = rng.permutation(30)
    inv = np.empty(30, np.int64)
    inv[perm] = np.arange(30)
    w["poses"], w["fixed"] = w["poses"][perm].copy(), w["fixed"][perm].copy()
    w["cam_idx"] = inv[w["cam_idx"]].astype(np.int32)
    ws.append(w)
    o = oa.Optimizer(max_cams=40, max_points=900, max_obs=20000, max_batch=len(ws))
    o.set_profiling(True)
    got = o.solve_batch(ws, (5, 10))
    assert o.iteration_ms()[2] == 0          # phase kernels
    o.close()
    for w, g in zip(ws, got):
        _check_estimates(g, oracle.ba_solve(w, rounds=(5, 10)), converged_ok=True)
