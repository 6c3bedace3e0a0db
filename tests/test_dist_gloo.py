"""CPU, world_size 2 and 3, gloo: the sharding + halo exchange that bench.py runs over RCCL
(orbslam_in_practice_amd/shard.py: HaloExchange, halo_pairs -- the very code bench.py calls).  Each rank extracts
its own block of frames (with the CPU oracle standing in for the device), the first frame slab of every rank is
all-gathered into the tail of the slab tensors, and rank r matches its frames against their successors in the
GLOBAL stream through LOCAL row indices -- the result must equal the single-process run.  The same hand-off with
device-produced slabs and the device matcher is tests/test_gpu_dist.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, CAP, FPR = 320, 240, 600, 2   # frames per rank


def _features(frames):
    from oracle import oracle as orc
    o = orc.OrbOracle(500, 1.2, 8, 20, 7)
    kps = np.zeros((len(frames), CAP), orc.KP_DTYPE)
    desc = np.zeros((len(frames), CAP, 32), np.uint8)
    cnt = np.zeros(len(frames), np.int32)
    for i, f in enumerate(frames):
        k, d = o.extract(f)
        cnt[i] = len(k)
        kps[i, :len(k)] = k
        desc[i, :len(k)] = d
    return kps, desc, cnt


def _match(kps, desc, cnt, a, b):
    from oracle import oracle as orc
    out = []
    for i, j in zip(a, b):
        n, m12, _ = orc.search_for_initialization(kps[i, :cnt[i]], desc[i, :cnt[i]], kps[j, :cnt[j]], desc[j, :cnt[j]],
                                                  (0.0, float(W), 0.0, float(H)), None, -1, 0.9, True)
        out.append((n, m12))
    return out


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orbslam_in_practice_amd import shard, synth
    from oracle import oracle as orc
    frames = synth.make_stream(5, world * FPR, W, H, n_shapes=900)
    lo, hi = shard.owned_range(world * FPR, rank, world)
    assert hi - lo == FPR
    kps, desc, cnt = _features(frames[lo:hi])
    rows = shard.halo_rows(world, FPR)
    s_k = torch.zeros((rows, CAP, 28), dtype=torch.uint8)
    s_d = torch.zeros((rows, CAP, 32), dtype=torch.uint8)
    s_c = torch.zeros((rows,), dtype=torch.int32)
    s_k[:FPR] = torch.from_numpy(kps.view(np.uint8).reshape(FPR, CAP, 28))
    s_d[:FPR] = torch.from_numpy(desc)
    s_c[:FPR] = torch.from_numpy(cnt)
    ex = shard.HaloExchange(world, FPR)
    ex(s_k, s_d, s_c)
    assert ex.collectives == 1          # a single fused all-gather per batch (packed [kps | desc | cnt] record)
    K = s_k.numpy().reshape(rows, CAP * 28).view(orc.KP_DTYPE).reshape(rows, CAP)
    a, b = shard.halo_pairs(rank, world, FPR)
    res = _match(K, s_d.numpy(), s_c.numpy(), a, b)
    q.put((rank, [(int(n), m.tolist()) for n, m in res], s_c.numpy().tolist(), ex.bytes_per_rank(CAP)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ranks_match_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, res, counts, nbytes = q.get(timeout=300)
        got[rank] = (res, counts, nbytes)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from orbslam_in_practice_amd import shard, synth
    frames = synth.make_stream(5, world * FPR, W, H, n_shapes=900)
    kps, desc, cnt = _features(frames)
    for r in range(world):
        # the halo region of every rank holds the FIRST frame of every rank, nothing else travelled
        assert got[r][1][FPR:] == [int(cnt[q_ * FPR]) for q_ in range(world)]
        assert got[r][2] == world * shard.slab_bytes(CAP)
        a, b = shard.stream_pairs(r, world, FPR)
        ref = _match(kps, desc, cnt, a, b)
        assert [(int(n), m.tolist()) for n, m in ref] == got[r][0]
    assert any(n > 10 for n, _ in got[0][0])


def test_halo_pairs_equal_global_stream_pairs():
    from orbslam_in_practice_amd import shard
    for world, B in [(1, 4), (2, 3), (8, 64)]:
        for r in range(world):
            ga, gb = shard.stream_pairs(r, world, B)
            la, lb = shard.halo_pairs(r, world, B)
            # local row -> global frame: own rows are [r*B, (r+1)*B), halo row B + q is frame q*B
            to_global = lambda x: r * B + x if x < B else (x - B) * B
            assert [to_global(int(x)) for x in la] == ga.tolist()
            assert [to_global(int(x)) for x in lb] == gb.tolist()
        assert shard.halo_rows(world, B) == B + (world if world > 1 else 0)


def test_shard_arithmetic():
    from orbslam_in_practice_amd import shard
    for n, w in [(512, 8), (10, 3), (7, 8), (1, 1)]:
        spans = [shard.owned_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    a, b = shard.stream_pairs(1, 2, 4)
    assert a.tolist() == [4, 5, 6, 7] and b.tolist() == [5, 6, 7, 0]
    assert shard.slab_bytes(1032) == 1032 * 60 + 4
