"""CPU: host-side logic added in round 3 that needs no device -- the memo of the filled C struct in the Python mirror of
Optimizer, the packed halo record of shard.HaloExchange (single process, the collective faked), and bench.py's stamping
of profiles/traffic.json (figures of a stage whose kernel source changed are dropped, never reported stale)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_optimizer_fill_memo_is_exact_and_invalidated():
    from orbslam_in_practice_amd import optimizer, synth
    from orbslam_in_practice_amd._lib import BAProblem
    O = optimizer.Optimizer
    w = synth.make_ba_window(3, 6, 80)
    P1, P2, P3 = BAProblem(), BAProblem(), BAProblem()
    keep1 = O._fill(P1, w, (5, 10), True, 5.991, 0, None)
    assert "_abi" in w
    keep2 = O._fill(P2, w, (5, 10), True, 5.991, 0, None)          # from the memo
    assert bytes(P1) == bytes(P2) and keep2 is keep1
    assert P2.n_cams == 6 and P2.n_rounds == 2 and P2.rounds[0] == 5 and P2.rounds[1] == 10
    O._fill(P3, w, (20,), False, 5.991, 0, None)                   # another schedule: not the memo
    assert P3.n_rounds == 1 and P3.rounds[0] == 20 and P3.huber_delta == 0.0 and P3.poses == P1.poses
    w["points"] = w["points"].copy()                               # an array replaced: the memo is void
    P4 = BAProblem()
    O._fill(P4, w, (20,), False, 5.991, 0, None)
    assert P4.points != P3.points and P4.poses == P3.poses
    stop = np.zeros(1, np.uint8)                                   # a stop flag is per call: never memoised
    P5 = BAProblem()
    O._fill(P5, w, (20,), False, 5.991, 0, stop)
    assert P5.stop == stop.ctypes.data
    P6 = BAProblem()
    O._fill(P6, w, (20,), False, 5.991, 0, None)
    assert not P6.stop


def test_optimizer_queue_errors_without_a_device():
    """wait() with nothing submitted is a clear error, also before any device exists"""
    from orbslam_in_practice_amd import optimizer
    o = optimizer.Optimizer.__new__(optimizer.Optimizer)
    o._jobs = []
    with pytest.raises(RuntimeError):
        o.wait()


def test_halo_exchange_packs_one_record_per_rank(monkeypatch):
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from orbslam_in_practice_amd import shard
    world, B, halo, cap = 4, 5, 2, 7
    rows = shard.halo_rows(world, B, halo)
    ranks = []
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        ranks.append((torch.randint(0, 255, (rows, cap, 28), dtype=torch.uint8, generator=g),
                      torch.randint(0, 255, (rows, cap, 32), dtype=torch.uint8, generator=g),
                      torch.randint(0, 5000, (rows,), dtype=torch.int32, generator=g)))
    packs = [torch.cat([k[:halo].reshape(-1), d[:halo].reshape(-1), c[:halo].contiguous().view(torch.uint8)]) for k, d, c in ranks]
    calls = []

    def fake_all_gather(parts, t, group=None):
        calls.append(int(t.numel()))
        for i, p_ in enumerate(parts):
            p_.copy_(packs[i])

    monkeypatch.setattr(dist, "all_gather", fake_all_gather)
    ex = shard.HaloExchange.__new__(shard.HaloExchange)
    ex.world, ex.B, ex.halo, ex.group, ex.staged, ex.collectives, ex._pack, ex._gath = world, B, halo, None, True, 0, None, None
    me = 2
    k, d, c = [t.clone() for t in ranks[me]]
    ex(k, d, c)
    assert ex.collectives == 1 and calls == [halo * shard.slab_bytes(cap)]       # ONE collective of one packed record
    assert ex.bytes_per_rank(cap) == world * halo * shard.slab_bytes(cap)
    for r in range(world):
        lo, hi = B + r * halo, B + (r + 1) * halo
        assert torch.equal(k[lo:hi], ranks[r][0][:halo]) and torch.equal(d[lo:hi], ranks[r][1][:halo]) and torch.equal(c[lo:hi], ranks[r][2][:halo])
    assert torch.equal(k[:B], ranks[me][0][:B]) and torch.equal(c[:B], ranks[me][2][:B])    # own rows untouched


def test_traffic_figures_are_dropped_when_a_kernel_source_changed(tmp_path, monkeypatch):
    import bench
    now = bench.stage_source_hashes()
    assert set(now) == set(bench.STAGE_SOURCE.values()) and all(len(v) == 16 for v in now.values())
    stamped = dict(now)
    stamped["orb_fast.hip"] = "0" * 16                                            # FAST changed since the PMC passes
    fake = {"_batch": 256, "_ba_batch": 256, "_collected": "test", "_src_sha16": stamped,
            "fast": 1, "valu_insts_fast": 11, "blur": 2, "valu_insts_blur": 22, "pyramid": 3, "ba_iteration": 4}
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    (root / "profiles" / "traffic.json").write_text(json.dumps(fake))
    os.symlink(os.path.join(ROOT, "orbslam_in_practice_amd"), root / "orbslam_in_practice_amd")
    monkeypatch.setattr(bench, "ROOT", str(root))
    tj, src = bench.load_traffic()
    assert "fast" not in tj and "valu_insts_fast" not in tj and tj["_stale_valu"] == {"fast": 11}
    assert tj["blur"] == 2 and tj["valu_insts_blur"] == 22 and tj["pyramid"] == 3 and tj["ba_iteration"] == 4
    assert "traffic.json@test" in src
    (root / "profiles" / "traffic.json").write_text(json.dumps({"fast": 1}))   # no stamp at all: everything is stale
    tj, _ = bench.load_traffic()
    assert "fast" not in tj


def test_committed_traffic_file_is_stamped():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert "_src_sha16" in tj and "_collected" in tj


def test_traffic_tally_counts_batch_launches_only(tmp_path):
    """tools/collect_traffic.py: the profiled bench line launches the front-end kernels on single frames too (its one-stream
    legs); those dispatches must not count as launches of the batch stages (they diluted every per-launch figure by their
    share until late round 3).  A synthetic counter file: 4 batch launches and 6 one-frame launches of FAST, resizes of two
    levels, one BA round."""
    import subprocess
    hdr = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"
    rows = []

    def add(name, grid, val, counter):
        rows.append("1,%d,1,1,1,1,%d,1,\"%s\",64,0,0,32,0,32,%s,%g,0,1" % (len(rows) + 1, grid, name, counter, val))
    for counter, unit in (("FETCH_SIZE", 1000.0), ("WRITE_SIZE", 500.0)):
        rows.clear()
        for _ in range(4):
            add("k_fast_cells_wave(OrbGeom)", 815 * 256 * 64, unit, counter)          # a batch of 256 frames
            add("k_copy_level0(OrbGeom)", 150 * 256 * 256, unit, counter)
            add("k_resize(OrbGeom)", 65 * 256 * 256, unit, counter)                   # level 1 of the batch
            add("k_resize(OrbGeom)", 10 * 256 * 256, unit, counter)                   # level 7 of the batch: still a batch launch
        for _ in range(6):
            add("k_fast_cells_wave(OrbGeom)", 815 * 64, unit / 256, counter)           # one frame (tracking loop)
            add("k_copy_level0(OrbGeom)", 150 * 256, unit / 256, counter)
            add("k_resize(OrbGeom)", 65 * 256, unit / 256, counter)
        add("k_ba_ph_hll(BABufs)", 256 * 256, 7 * unit, counter)
        add("k_ba_ph_round(BABufs,int,int)", 256, 0.0, counter)
        d = tmp_path / ("pmc_" + counter)
        d.mkdir()
        (d / "1_counter_collection.csv").write_text(hdr + "\n" + "\n".join(rows) + "\n")
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "collect_traffic.py"), str(tmp_path / "pmc_FETCH_SIZE"),
                    str(tmp_path / "pmc_WRITE_SIZE"), str(out), "256"], check=True, capture_output=True)
    tj = json.load(open(out))
    assert tj["fast"] == int(1000.0 * 1024 * 2 + 500.0 * 1024)                    # per batch launch: fetch doubled (gfx950) + write
    assert tj["pyramid"] == int(3 * (1000.0 * 1024 * 2 + 500.0 * 1024))           # copy + two resize levels, per batch launch
    assert tj["ba_iteration"] == int(7 * (1000.0 * 1024 * 2 + 500.0 * 1024))
