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
