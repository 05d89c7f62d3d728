"""Frame/step sharding across ranks: contiguous time shards, no data-path
collective.  Pure index arithmetic, shared by bench.py and the gloo tests.

Rank r of R owns frames [lo, hi) of the bulk indexing (frame h = samples [(h+1)*hop - N, (h+1)*hop)).
Its device image is the padded layout of include/melonix_amd.h with the *true* neighbour samples in
the pads (zeros only beyond the ends of the whole signal), so frames that straddle a shard boundary
read exactly what an unsharded run reads: the input halo is N - hop samples on the left edge.
"""
from __future__ import annotations

from dataclasses import dataclass

MX_AUDIO_PAD = 32768


@dataclass(frozen=True)
class FrameShard:
    rank: int
    world: int
    lo: int            # first global frame
    hi: int            # one past the last global frame
    sample_lo: int     # global index of the shard's first own sample (= lo * hop)
    sample_hi: int     # one past the last own sample
    halo_left: int     # samples of the left neighbour the first frame reads (N - hop, clipped at 0)

    @property
    def frames(self) -> int:
        return self.hi - self.lo


def frame_count(n: int, hop: int) -> int:
    return (n + hop - 1) // hop


def shard_frames(n: int, N: int, hop: int, rank: int, world: int) -> FrameShard:
    """Equal contiguous frame ranges (the last ranks get one frame less when F % world != 0)."""
    F = frame_count(n, hop)
    base, extra = divmod(F, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    s_lo = lo * hop
    s_hi = min(n, hi * hop)
    return FrameShard(rank, world, lo, hi, s_lo, s_hi, min(N - hop, s_lo))


def padded_shard(wav, shard: FrameShard, pad: int = MX_AUDIO_PAD):
    """Host construction of a rank's padded image from the whole signal (numpy)."""
    import numpy as np

    n = len(wav)
    own = shard.sample_hi - shard.sample_lo
    out = np.zeros(own + 2 * pad, dtype=np.float32)
    g0 = shard.sample_lo - pad
    a, b = max(g0, 0), min(shard.sample_hi + pad, n)
    out[a - g0:b - g0] = wav[a:b]
    return out, own


def shard_steps(nsteps: int, rank: int, world: int):
    """Contiguous step ranges for the resynthesis schedule (outputs land at out_offset, so ranks
    write disjoint PCM ranges; nextGrainFirstSample is already in each step record)."""
    base, extra = divmod(nsteps, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
