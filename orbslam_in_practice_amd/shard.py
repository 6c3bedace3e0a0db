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
# the exchange below all-gathers the first `halo` frames of every rank (world x halo slabs = 0.5 MB at 8 GPUs), one
# RCCL all-gather per slab array, straight into the tail of the slab tensors the matcher reads:
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
    (kps [rows, cap, 28] u8, desc [rows, cap, 32] u8, cnt [rows] i32).  Backend "nccl" (= RCCL) gathers device
    tensors in place; "gloo" (CPU tests, one-GPU rehearsals) goes through host copies."""

    def __init__(self, world, frames_per_rank, halo=1, group=None):
        import torch.distributed as dist
        self.world, self.B, self.halo, self.group = world, frames_per_rank, halo, group
        self.staged = world > 1 and dist.get_backend(group) != "nccl"

    def bytes_per_rank(self, cap):
        return self.world * self.halo * slab_bytes(cap) if self.world > 1 else 0

    def __call__(self, kps, desc, cnt):
        if self.world == 1:
            return
        import torch.distributed as dist
        B, h, w = self.B, self.halo, self.world
        for t in (kps, desc, cnt):
            src, dst = t[:h], t[B:B + w * h]
            if self.staged:
                parts = [src.new_empty(src.shape, device="cpu") for _ in range(w)]
                dist.all_gather(parts, src.cpu().contiguous(), group=self.group)
                for r in range(w):
                    dst[r * h:(r + 1) * h].copy_(parts[r])
            else:
                dist.all_gather_into_tensor(dst.view(-1), src.reshape(-1), group=self.group)
