"""N>1 path on CPU: world_size-2 gloo run of the read sharding + hit gather (muscato_amd/dist.py).

Each rank matches its contiguous shard of the sorted unique reads against the replicated
targets (the per-shard matcher here is the CPU oracle -- the GPU engine cannot run in this
container); rank 0's gathered hits must equal the single-process result, in global read order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, outdir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cases import make_case
    from muscato_amd.dist import gather_hits, shard_range
    from oracle import muscato_oracle as orc

    cfg, reads, targets = make_case(seed)
    lo, hi = shard_range(len(reads), rank, world)
    local = sorted(orc.best_filter(orc.match_direct(reads[lo:hi], targets, cfg), cfg.MMTol))
    t = torch.tensor(local, dtype=torch.int32).reshape(-1, 4)
    g = gather_hits(t, lo, dst=0)
    if rank == 0:
        np.save(os.path.join(outdir, "gathered.npy"), g.numpy())
    else:
        assert g is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", [2, 5, 14])
def test_two_rank_gather_equals_single_process(tmp_path, seed):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import make_case
    from oracle import muscato_oracle as orc

    mp.spawn(_worker, args=(2, _free_port(), seed, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    cfg, reads, targets = make_case(seed)
    exp = np.array(sorted(orc.best_filter(orc.match_direct(reads, targets, cfg), cfg.MMTol)), dtype=np.int32).reshape(-1, 4)
    assert got.shape == exp.shape
    # rank-order concatenation of per-shard sorted hits is already globally sorted by read
    assert (got == exp).all()


def test_shard_range_partitions():
    from muscato_amd.dist import shard_range
    for n in (0, 1, 7, 100, 45_000_000):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


def test_gather_single_rank_is_identity():
    from muscato_amd.dist import gather_hits
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        t = torch.tensor([[0, 1, 2, 3], [4, 5, 6, 7]], dtype=torch.int32)
        g = gather_hits(t, 10)
        assert g.tolist() == [[10, 1, 2, 3], [14, 5, 6, 7]]
        assert t.tolist() == [[0, 1, 2, 3], [4, 5, 6, 7]]
    finally:
        dist.destroy_process_group()


def _stream_worker(rank, world, port, outdir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from muscato_amd.dist import HitGatherer

    def tuples(p):  # pass p, this rank: a different number of tuples every time
        n = 5 + 3 * rank + 2 * p
        t = torch.arange(n * 4, dtype=torch.int32).reshape(n, 4) + 1000 * p
        t[:, 0] = torch.arange(n, dtype=torch.int32)  # shard-local read index
        return t

    # capacities agreed ONCE, per rank (rank 1 holds more tuples than rank 0): no collective per pass afterwards
    caps = HitGatherer.agree_caps(5 + 3 * rank + 2 * 4, torch.device("cpu"), slack=1.0)
    assert caps == [5 + 3 * r + 8 + 16 for r in range(world)]
    cap = max(caps)
    g = HitGatherer(cap, torch.device("cpu"), depth=2, rank_caps=caps)
    snaps = []
    for p in range(5):
        t = tuples(p)

        def fill(buf, t=t):
            buf[:len(t)] = t
            return len(t)
        assert g.submit(fill, read_base=100 * rank) == len(t)
        if p == 2:  # look at a finished pass in the middle of the stream
            cn = g.finish()
            if rank == 0:
                snaps.append((cn, g.last_result().clone()))
    cn = g.finish()
    if rank == 0:
        snaps.append((cn, g.last_result().clone()))
        torch.save(snaps, os.path.join(outdir, "snaps.pt"))
    # a rank whose list outgrows ITS capacity: it raises, its buffer travels with the header -1 (the sizes are fixed:
    # nobody is left waiting in a collective), and rank 0 raises when it reads that header
    if rank == 1:
        with pytest.raises(RuntimeError):
            g.submit(lambda buf: caps[1] + 1, 0)
    else:
        g.submit(lambda buf: 0, 0)
        with pytest.raises(RuntimeError, match=r"rank\(s\) \[1\]"):
            g.finish()
    dist.barrier()
    dist.destroy_process_group()


def test_streaming_gatherer_two_ranks(tmp_path):
    """HitGatherer: overlapped per-pass gathers with counts riding in the buffers (what
    bench.py --gpus N uses); buffers are reused every `depth` passes."""
    mp.spawn(_stream_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    snaps = torch.load(tmp_path / "snaps.pt")
    for (cn, res), p in zip(snaps, (2, 4)):
        exp = []
        for rank in range(2):
            n = 5 + 3 * rank + 2 * p
            t = torch.arange(n * 4, dtype=torch.int32).reshape(n, 4) + 1000 * p
            t[:, 0] = torch.arange(n, dtype=torch.int32) + 100 * rank
            exp.append(t)
        assert cn == [len(e) for e in exp]
        assert torch.equal(res, torch.cat(exp))


def _packed_worker(rank, world, port, outdir):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from muscato_amd.dist import HitGatherer
    bits = (20, 12, 10, 3)
    g = HitGatherer(64, torch.device("cpu"), depth=2, packed=True)
    for p in range(3):
        n = 7 + rank + p
        t = torch.stack([torch.arange(n) + 1000 * rank, torch.arange(n) * 3 % 4096, torch.arange(n) * 7 % 1024,
                         torch.arange(n) % 8], dim=1).to(torch.int64)
        words = (((t[:, 0] << bits[1] | t[:, 1]) << bits[2] | t[:, 2]) << bits[3]) | t[:, 3]

        def fill(buf, words=words):
            buf[:len(words)] = words
            return len(words)
        g.submit(fill, read_base=0)
    cn = g.finish()
    if rank == 0:
        torch.save((cn, HitGatherer.unpack(g.last_result(), bits)), os.path.join(outdir, "packed.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_streaming_gatherer_packed_words(tmp_path):
    mp.spawn(_packed_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    cn, res = torch.load(tmp_path / "packed.pt")
    exp = []
    for rank in range(2):
        n = 7 + rank + 2
        exp.append(torch.stack([torch.arange(n) + 1000 * rank, torch.arange(n) * 3 % 4096, torch.arange(n) * 7 % 1024,
                                torch.arange(n) % 8], dim=1).to(torch.int64))
    assert cn == [len(e) for e in exp]
    assert torch.equal(res, torch.cat(exp))
