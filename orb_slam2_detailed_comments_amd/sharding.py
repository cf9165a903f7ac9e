"""Batched-frames mode across the GPUs of one node.

Frames are independent units (the extractor keeps no cross-frame state, reference include/ORBextractor.h:30-35),
so a batch is cut into contiguous blocks, one block per rank, with NO collective on the data path.  The only
exchange is the gather of the per-frame result records -- one fixed-stride all-gather per batch:

    record(frame) = { int32 count ; orbx_keypoint[cap] ; uint8 desc[cap][32] }

`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs); "gloo" is used by the CPU tests.
"""
from __future__ import annotations
import torch
import torch.distributed as dist

KP_BYTES = 28
DESC_BYTES = 32


def shard_range(nframes: int, world: int, rank: int):
    """contiguous block [begin, end) of rank `rank`; blocks differ by at most one frame"""
    base, rem = divmod(nframes, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_ranges_live(nframes: int, world: int, dead=()):
    """Re-sharding around missing GPUs ("survive a missing GPU by re-sharding -- replicas only, no
    state"): the blocks of all `world` ranks when the ranks in `dead` take no frames.  The live ranks split the batch as
    shard_range would for len(live) ranks, in rank order, so the concatenation of the live blocks is the whole batch in
    frame order whatever the set of dead ranks is; a dead rank gets the empty block at the position where its block would
    start.  The path keeps no state between frames (include/ORBextractor.h:30-35), so nothing else has to move."""
    live = [r for r in range(world) if r not in set(dead)]
    if not live:
        raise ValueError("no live rank left")
    out, pos = [], 0
    for r in range(world):
        if r in live:
            b, e = shard_range(nframes, len(live), live.index(r))
            out.append((b, e)); pos = e
        else:
            out.append((pos, pos))
    return out


def record_bytes(cap: int) -> int:
    return 4 + cap * (KP_BYTES + DESC_BYTES)


def pack_records(counts: torch.Tensor, kps: torch.Tensor, desc: torch.Tensor, out: torch.Tensor = None):
    """counts int32[B], kps uint8[B, cap*28] (raw keypoint bytes), desc uint8[B, cap*32] -> uint8[B, record_bytes]"""
    b = counts.shape[0]
    cap = desc.shape[1] // DESC_BYTES
    rb = record_bytes(cap)
    if out is None:
        out = torch.empty((b, rb), dtype=torch.uint8, device=counts.device)
    out[:, 0:4] = counts.view(torch.uint8).view(b, 4)
    out[:, 4:4 + cap * KP_BYTES] = kps
    out[:, 4 + cap * KP_BYTES:] = desc
    return out


def unpack_records(buf: torch.Tensor, cap: int):
    b = buf.shape[0]
    counts = buf[:, 0:4].contiguous().view(torch.int32).view(b)
    kps = buf[:, 4:4 + cap * KP_BYTES]
    desc = buf[:, 4 + cap * KP_BYTES:]
    return counts, kps, desc


class RecordGatherer:
    """One all-gather of equal-sized per-rank record blocks per batch, optionally left in flight
    (async) while the next batch is computed."""

    def __init__(self, frames_per_rank: int, cap: int, device, group=None, mode: str = "gather", root: int = 0):
        """mode "gather": records go to rank `root` only (the consumer of the keypoints; 1/N of the all-gather traffic);
        mode "all_gather": every rank receives every record."""
        assert mode in ("gather", "all_gather")
        self.mode, self.root = mode, root
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.cap = cap
        self.rb = record_bytes(cap)
        self.frames_per_rank = frames_per_rank
        self.send = [torch.empty((frames_per_rank, self.rb), dtype=torch.uint8, device=device) for _ in range(2)]
        need_recv = mode == "all_gather" or self.rank == root
        self.recv = [torch.empty((self.world * frames_per_rank, self.rb) if need_recv else (0, self.rb),
                                 dtype=torch.uint8, device=device) for _ in range(2)]
        self.work = [None, None]
        self.slot = 0

    def gather(self, counts, kps, desc, async_op=False):
        s = self.slot
        if self.work[s] is not None:
            self.work[s].wait()
            self.work[s] = None
        pack_records(counts, kps, desc, out=self.send[s])
        w = self._collective(s, async_op)
        self.work[s] = w if async_op else None
        self.slot ^= 1
        return self.recv[s]

    def _collective(self, s, async_op):
        if self.world == 1:
            self.recv[s].copy_(self.send[s])
            return None
        if self.mode == "all_gather":
            return dist.all_gather_into_tensor(self.recv[s], self.send[s], group=self.group, async_op=async_op)
        parts = list(self.recv[s].view(self.world, self.frames_per_rank, self.rb).unbind(0)) if self.rank == self.root else None
        return dist.gather(self.send[s], parts, dst=self.root, group=self.group, async_op=async_op)

    def collective_only(self):
        """the exchange alone, synchronous, on what the send buffer of slot 0 holds (bench.py times it between two events to
        tell the collective's share of a step from the compute's)"""
        self.wait_all()
        self._collective(0, False)

    def wait_all(self):
        for i in (0, 1):
            if self.work[i] is not None:
                self.work[i].wait()
                self.work[i] = None
