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
