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
