"""How the hot path shards over the GPUs of one node: frames and BA windows are
independent units, rank r owns a contiguous block of them, and the only exchange is an all-gather of the
fixed-capacity per-frame feature slabs so that a rank can match its last frame against the next rank's
first one.  Pure index arithmetic, shared by bench.py and the gloo tests."""
import numpy as np


def owned_range(n_units, rank, world):
    """contiguous block partition; the first (n_units % world) ranks get one extra unit"""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def stream_pairs(rank, world, frames_per_rank):
    """(a, b) global frame indices of the pairs rank `rank` matches: its own frames against their
    successors in the global stream (the last one wraps to the next rank's first frame)."""
    n_global = world * frames_per_rank
    a = np.arange(rank * frames_per_rank, (rank + 1) * frames_per_rank, dtype=np.int32)
    b = ((a + 1) % n_global).astype(np.int32)
    return a, b


def slab_bytes(cap):
    """bytes one frame contributes to the exchange: cap keypoints (28 B) + cap descriptors (32 B) + count"""
    return cap * 28 + cap * 32 + 4


# ---------------------------------------------------------------------------------------------------------------
# The exchange step.  A rank matches frame k against frame k+1; only its LAST frame has a successor on another
# rank, so the only remote data a rank ever reads is the first frame of its right-hand neighbour.  Gathering every
# rank's whole block (world x B frame slabs, 253 MB per rank per step at 8 x 512) would move 500 x what is needed;
# the exchange below all-gathers the first `halo` frames of every rank (world x halo slabs = 0.5 MB at 8 GPUs) as ONE
# RCCL all-gather of a packed [kps | desc | count] record per rank, scattered into the tail of the slab tensors the
# matcher reads:
#
#     slab rows  [0, B)                      the rank's own frames
#                [B, B + world * halo)       halo region: frames [0, halo) of rank 0, rank 1, ...
#
# Pair lists are in these LOCAL row indices.  halo > 1 serves callers that match a frame against several successors.


def halo_rows(world, frames_per_rank, halo=1):
    """rows a slab tensor needs: own frames + (world > 1: the gathered halo region)"""
    return frames_per_rank + (world * halo if world > 1 else 0)


def halo_pairs(rank, world, frames_per_rank, halo=1):
    """(a, b) LOCAL slab rows of the pairs rank `rank` matches: own frame i against its successor in the global
    stream; the last frame's successor is the next rank's first frame, found in the halo region (the global stream
    wraps: the last rank's successor is rank 0's first frame; a single rank wraps onto its own first frame)."""
    B = frames_per_rank
    a = np.arange(B, dtype=np.int32)
    b = a + 1
    b[-1] = 0 if world == 1 else B + ((rank + 1) % world) * halo
    return a, b.astype(np.int32)


class HaloExchange:
    """all-gather of the first `halo` frame slabs of every rank into rows [B, B + world*halo) of the slab tensors
    (kps [rows, cap, 28] u8, desc [rows, cap, 32] u8, cnt [rows] i32) -- ONE collective per batch ("a single
    fused all-gather per batch"): the rank's `halo` frames are packed into one contiguous record
    `[kps rows | desc rows | counts]` (one concatenating copy), `all_gather_into_tensor` moves world x that record, and
    three strided copies scatter the gathered records into the tail rows the matcher reads.  Backend "nccl" (= RCCL)
    gathers device buffers in place; "gloo" (CPU tests, one-GPU rehearsals) goes through host copies of the SAME packed
    record.  The pack / gather buffers belong to this object: calls must be ordered on one stream (bench.py: lane 0's
    match stream)."""

    def __init__(self, world, frames_per_rank, halo=1, group=None):
        self.world, self.B, self.halo, self.group = world, frames_per_rank, halo, group
        self.staged = False
        self.collectives = 0      # all-gathers issued so far (tests: exactly one per call)
        self._pack = self._gath = None
        if world > 1:
            import torch.distributed as dist
            self.staged = dist.get_backend(group) != "nccl"

    def bytes_per_rank(self, cap):
        return self.world * self.halo * slab_bytes(cap) if self.world > 1 else 0

    def _buffers(self, like, nbytes):
        import torch
        if self._pack is None or self._pack.numel() != nbytes or self._pack.device != like.device:
            self._pack = torch.empty((nbytes,), dtype=torch.uint8, device=like.device)
            self._gath = torch.empty((self.world, nbytes), dtype=torch.uint8, device=like.device)
        return self._pack, self._gath

    def __call__(self, kps, desc, cnt):
        if self.world == 1:
            return
        import torch
        import torch.distributed as dist
        B, h, w = self.B, self.halo, self.world
        srcs = [kps[:h].reshape(-1), desc[:h].reshape(-1), cnt[:h].contiguous().view(torch.uint8)]
        sizes = [int(t.numel()) for t in srcs]
        pack, gath = self._buffers(kps, sum(sizes))
        torch.cat(srcs, out=pack)                                      # [kps | desc | cnt] of this rank's halo frames
        if self.staged:
            parts = [torch.empty(pack.shape, dtype=torch.uint8) for _ in range(w)]
            dist.all_gather(parts, pack.cpu(), group=self.group)
            gath.copy_(torch.stack(parts))
        else:
            dist.all_gather_into_tensor(gath.view(-1), pack, group=self.group)
        self.collectives += 1
        o0, o1 = sizes[0], sizes[0] + sizes[1]
        kps[B:B + w * h].view(w, -1).copy_(gath[:, :o0])
        desc[B:B + w * h].view(w, -1).copy_(gath[:, o0:o1])
        cnt[B:B + w * h].view(torch.uint8).view(w, -1).copy_(gath[:, o1:])
