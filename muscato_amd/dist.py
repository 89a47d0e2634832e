"""One process per GPU: shard the unique reads, replicate the target database, gather hits.

The path has no exchange step: every read is matched independently against the replicated
database, so the only collective is the concatenation of the per-rank hit
lists on rank 0 (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for the tests).  Because the reference orders results.txt by read sequence first
(cmd/muscato/main.go:657-659) and shards are contiguous ranges of the sorted unique reads,
concatenating shards in rank order keeps the global order.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n sorted unique reads owned by `rank`."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def gather_hits(local: torch.Tensor, read_base: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """local: int32 [n_local, 4] hits with shard-local read_idx (any device).  Adds
    `read_base` to column 0 and concatenates all ranks' hits on rank `dst` in rank order.
    Returns the [n_total, 4] tensor on `dst`, None elsewhere.

    One all_gather of the counts (8 B per rank) and one padded gather of the tuples."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local = local.reshape(-1, 4)
    if read_base:
        local = local.clone()
        local[:, 0] += read_base
    if world == 1:
        return local
    dev = local.device
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    mx = max(max(counts), 1)
    padded = torch.zeros((mx, 4), dtype=local.dtype, device=dev)
    padded[:local.shape[0]] = local
    if rank == dst:
        bufs: List[torch.Tensor] = [torch.empty((mx, 4), dtype=local.dtype, device=dev) for _ in range(world)]
        dist.gather(padded, gather_list=bufs, dst=dst, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    dist.gather(padded, gather_list=None, dst=dst, group=group)
    return None


class HitGatherer:
    """Streaming gatherv of the tuples of a sequence of passes (bench.py --gpus N, a service
    matching batch after batch): per pass ONE group of point-to-point transfers -- every rank
    `isend`s its buffer to rank `dst`, `dst` posts one `irecv` per peer into that peer's slab of a
    rank-major receive buffer, all of them inside one ncclGroupStart/End (torch's
    batch_isend_irecv), so the seven incoming transfers of an 8-GPU node run side by side on the
    seven xGMI links of `dst` instead of one after the other.  The transfer of pass i runs on the
    communicator's stream while pass i+1 is being matched.

    NOTHING BLOCKS BETWEEN THE MATCH AND THE TRANSFERS (r04): the size of every rank's transfer is
    agreed ONCE, when the gatherer is built -- `rank_caps[r]` rows for rank r (agree_caps(): one
    all_gather of the counts of the sizing pass, + slack) -- so submit() posts its isend / irecv
    without a collective and without reading a device value on the host (r03 agreed the row counts
    with an all_gather + .item() per pass: a host synchronisation of all ranks inside a 2 ms
    step).  Link load follows the tuples as closely as the slack allows: a rank sends its own
    capacity, not the largest rank's.  The count of a pass rides in the buffer's header row.  A
    rank whose list does not fit its capacity sends its buffer with the header -1 and raises; `dst`
    raises when it reads that header (finish / counts): nobody waits in a collective for it.

    Every rank owns `depth` send buffers of 1 + `cap` rows; row 0 carries the tuple count.  With
    `packed=True` a row is one int64 word (the layout of musc_hits_copy_packed: half the bytes on
    the links; fill() must then write words that already include the shard's read base),
    otherwise four int32 (read, gene, pos, nmiss).  Rank `dst` owns `depth` receive buffers of
    world x (1 + cap) rows.  submit() fills the next send buffer through `fill(rows) -> n` (`rows` =
    the buffer behind its header row; e.g. Engine.hits_to), rebases column 0 by `read_base` and starts the transfers; it first
    waits for the ones issued `depth` passes earlier, whose buffers it reuses.  finish() waits for
    everything; on `dst`, counts(k) / last_result() then give the tuples per rank -- rank-order
    concatenation is the global read order."""

    def __init__(self, cap: int, device, depth: int = 2, dst: int = 0, group=None, packed: bool = False,
                 compact_reads: int = 0, rank_caps: Optional[List[int]] = None):
        """compact_reads > 0 selects the compact form (musc_hits_copy_compact; `compact_reads` = the
        largest number of reads any rank holds): a buffer is int32 words [n tuples, n reads] +
        one count byte per read + one word per tuple -- 5 bytes per tuple at one tuple per read,
        against 8 (packed) and 16.  fill(buf) then writes through compact_views(buf) and returns
        (n, n_reads).  rank_caps: tuples rank r may send per pass (<= cap; default: cap for every rank)."""
        self.cap, self.depth, self.dst, self.group, self.packed = int(cap), depth, dst, group, packed
        self.compact_reads = int(compact_reads)
        self.cw = (self.compact_reads + 3) // 4  # words of the count bytes
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rank_caps = [min(int(c), self.cap) for c in rank_caps] if rank_caps is not None else [self.cap] * self.world
        if len(self.rank_caps) != self.world:
            raise ValueError("HitGatherer: rank_caps needs one entry per rank")
        hdr = 2 + self.cw if self.compact_reads else 1
        self.rows = [hdr + c for c in self.rank_caps]  # rows of rank r's transfer, every pass
        shape, dtype = ((1 + self.cap,), torch.int64) if packed else ((1 + self.cap, 4), torch.int32)
        if self.compact_reads:
            shape, dtype = (2 + self.cw + self.cap,), torch.int32
        self.recv = None
        if self.rank == dst:
            # the destination fills its own slab of the receive buffer in place: its shard never
            # touches a link, nor a second buffer
            self.recv = [torch.empty((self.world,) + shape, dtype=dtype, device=device) for _ in range(depth)]
            for r in self.recv:
                r[self.rank].zero_()
            self.send = [r[self.rank] for r in self.recv]
        else:
            self.send = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(depth)]
        # the counts of a pass reach a device buffer through one small pinned staging tensor per
        # buffer set (one asynchronous copy instead of a blocking scalar write per field)
        self.head = None
        if torch.device(device).type == "cuda":
            self.head = [torch.zeros(2, dtype=dtype, pin_memory=True) for _ in range(depth)]
        self.work = [None] * depth
        self.i = 0

    @staticmethod
    def agree_capacity(n_local: int, device, slack: float = 1.05, group=None) -> int:
        """A capacity every rank's hit list fits in: max over ranks, plus slack."""
        t = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return int(int(t.item()) * slack) + 16

    @staticmethod
    def agree_caps(n_local: int, device, slack: float = 1.05, group=None) -> List[int]:
        """Per-rank capacities, agreed ONCE (setup, not per pass): every rank's tuple count of the pass
        that sized the buffers, plus slack.  max() of the list is the buffer capacity."""
        world = dist.get_world_size(group)
        mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
        allr = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, mine, group=group)
        else:
            allr = [mine]
        return [int(int(t.item()) * slack) + 16 for t in allr]

    def _wait(self, k: int) -> None:
        if self.work[k] is not None:
            for w in self.work[k]:
                w.wait()
            if self.send[k].is_cuda:
                torch.cuda.current_stream().synchronize()  # fill() may write from another stream
            self.work[k] = None

    def submit(self, fill, read_base: int) -> int:
        k = self.i % self.depth
        self._wait(k)
        buf = self.send[k]
        n = fill(buf if self.compact_reads else buf[1:])
        nreads = 0
        if self.compact_reads:
            n, nreads = n
        n = int(n)
        mycap = self.rank_caps[self.rank]
        err = None
        if self.compact_reads and nreads > self.compact_reads:
            err = "HitGatherer: %d reads exceed the agreed capacity %d" % (nreads, self.compact_reads)
        elif n > mycap:
            err = "HitGatherer: %d hits exceed the agreed capacity %d" % (n, mycap)
        nh = -1 if err else n  # (the header -1 tells `dst` that this rank gave up; its rows still travel: the sizes are fixed)
        if self.compact_reads:
            if self.head is not None:
                h = self.head[k]  # (free again: the transfers of `depth` passes ago were waited for)
                h[0] = nh
                h[1] = int(nreads) if not err else 0
                buf[:2].copy_(h, non_blocking=True)
            else:
                buf[0] = nh
                buf[1] = int(nreads) if not err else 0
        elif self.packed:
            if self.head is not None:
                self.head[k][0] = nh
                buf[:1].copy_(self.head[k][:1], non_blocking=True)
            else:
                buf[0] = nh
        else:
            if read_base and n and not err:
                buf[1:1 + n, 0] += read_base
            buf[0, 0] = nh
        # the transfers: sizes agreed at setup (self.rows) -- no collective, no device value read on the host
        if self.rank == self.dst:
            ops = [dist.P2POp(dist.irecv, self.recv[k][r][:self.rows[r]], r, self.group) for r in range(self.world) if r != self.dst]
        else:
            ops = [dist.P2POp(dist.isend, buf[:self.rows[self.rank]], self.dst, self.group)]
        self.work[k] = dist.batch_isend_irecv(ops) if ops else []
        self.i += 1
        if err:
            raise RuntimeError(err)
        return n

    def counts(self, k: int) -> List[int]:
        """Tuple counts per rank of buffer set k (rank dst, after its transfers completed)."""
        if self.compact_reads:
            last = self.recv[k][:, 0]
        else:
            last = self.recv[k][:, 0] if self.packed else self.recv[k][:, 0, 0]
        cn = [int(c) for c in last.tolist()]
        bad = [r for r, c in enumerate(cn) if c < 0]
        if bad:
            raise RuntimeError("HitGatherer: rank(s) %s could not fit their tuples into the agreed capacity" % bad)
        return cn

    def compact_views(self, buf: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(count bytes [compact_reads] uint8, tuple words [cap] int32) of a compact buffer."""
        counts = buf[2:2 + self.cw].view(torch.uint8)[:self.compact_reads]
        return counts, buf[2 + self.cw:]

    def unpack_compact(self, slab: torch.Tensor, read_base: int, bits) -> torch.Tensor:
        """One rank's compact buffer -> int64 [n, 4] (read + read_base, gene, pos, nmiss)."""
        n, nreads = int(slab[0]), int(slab[1])
        counts, words = self.compact_views(slab)
        bg, bp, bn = bits
        reads = torch.repeat_interleave(torch.arange(nreads, device=slab.device, dtype=torch.int64) + int(read_base),
                                        counts[:nreads].to(torch.int64))
        w = words[:n].to(torch.int64) & 0xFFFFFFFF
        return torch.stack([reads, (w >> (bp + bn)) & ((1 << bg) - 1), (w >> bn) & ((1 << bp) - 1), w & ((1 << bn) - 1)], dim=1)

    def finish(self) -> Optional[List[int]]:
        """Wait for all outstanding transfers; on dst return the per-rank counts of the last pass."""
        for k in range(self.depth):
            self._wait(k)
        if self.send[0].is_cuda:
            torch.cuda.current_stream().synchronize()
        if self.rank != self.dst or self.i == 0:
            return None
        return self.counts((self.i - 1) % self.depth)

    @staticmethod
    def unpack(words: torch.Tensor, bits) -> torch.Tensor:
        """int64 words of musc_hits_copy_packed -> int64 [n, 4] (read, gene, pos, nmiss)."""
        br, bg, bp, bn = bits
        w = words.to(torch.int64)
        nm = w & ((1 << bn) - 1)
        pos = (w >> bn) & ((1 << bp) - 1)
        gene = (w >> (bn + bp)) & ((1 << bg) - 1)
        read = (w >> (bn + bp + bg)) & ((1 << br) - 1)   # the mask also undoes the sign extension of bit 63
        return torch.stack([read, gene, pos, nm], dim=1)

    def last_result(self) -> Optional[torch.Tensor]:
        """After finish(): the last pass's tuples on dst, concatenated in rank order (int64 words
        when packed -- see unpack())."""
        if self.rank != self.dst or self.i == 0:
            return None
        k = (self.i - 1) % self.depth
        if self.compact_reads:
            raise RuntimeError("compact buffers carry no read numbers: use last_slabs() + unpack_compact()")
        return torch.cat([self.recv[k][r][1:1 + c] for r, c in enumerate(self.counts(k))], dim=0)

    def last_slabs(self) -> Optional[torch.Tensor]:
        """After finish(): rank dst's receive buffer of the last pass, one slab per rank."""
        if self.rank != self.dst or self.i == 0:
            return None
        return self.recv[(self.i - 1) % self.depth]
