"""The N>1 path end to end on two ranks over gloo: Engine-shaped `fill` callbacks through
HitGatherer with `n_shards` = world, as bench.py --gpus N and a service would drive it.

CPU (`-m "not gpu"`): the per-shard matcher is a stand-in with the Engine's methods backed by the
CPU oracle (this container has no GPU); `-m gpu`: two ranks share the one MI355X of the test box,
each with a real Engine (RCCL refuses two ranks on one device, so the transport stays gloo -- the
collective calls are the same).  Rank 0's concatenation must equal the single-process result in
global read order, in the 16-byte and in the packed 8-byte form."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BITS = (16, 12, 16, 8)
CBITS = (12, 14, 6)  # gene, pos, nmiss of the compact form: 32 bits


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleEngine:
    """The Engine methods the N>1 path uses, on the CPU oracle (tests only)."""

    def __init__(self, ocfg, targets):
        self.ocfg, self.targets, self.reads, self.hits = ocfg, targets, [], np.zeros((0, 4), np.uint32)
        self.n_shards_seen = []

    def load_reads(self, reads):
        self.reads = list(reads)
        self.n_reads = len(self.reads)

    def match_device(self, cfg, apply_mmtol=True, skip_block_check=False, n_shards=1):
        from oracle import muscato_oracle as orc
        assert cfg.to_params(apply_mmtol, skip_block_check, n_shards).n_shards == n_shards
        self.n_shards_seen.append(n_shards)
        h = orc.match_direct(self.reads, self.targets, self.ocfg)
        if apply_mmtol:
            h = orc.best_filter(h, self.ocfg.MMTol)
        self.hits = np.array(sorted(h), dtype=np.uint32).reshape(-1, 4)
        return len(self.hits)

    def hits_to(self, ptr, capacity, on_device):
        import ctypes
        assert capacity >= len(self.hits) and not on_device
        ctypes.memmove(ptr, self.hits.ctypes.data, self.hits.nbytes)

    def hits_to_packed(self, ptr, capacity, on_device, bits, read_base=0):
        import ctypes
        h = self.hits.astype(np.uint64)
        w = ((((h[:, 0] + np.uint64(read_base)) << np.uint64(bits[1]) | h[:, 1]) << np.uint64(bits[2]) | h[:, 2])
             << np.uint64(bits[3])) | h[:, 3]
        w = np.ascontiguousarray(w)
        ctypes.memmove(ptr, w.ctypes.data, w.nbytes)

    def hits_to_compact(self, words_ptr, words_cap, counts_ptr, counts_cap, on_device, bits):
        import ctypes
        h = self.hits.astype(np.uint64)
        assert words_cap >= len(h) and counts_cap >= self.n_reads and not on_device
        w = (((h[:, 1] << np.uint64(bits[1])) | h[:, 2]) << np.uint64(bits[2]) | h[:, 3]).astype(np.uint32)
        cnt = np.bincount(self.hits[:, 0].astype(np.int64), minlength=self.n_reads).astype(np.uint8)
        w = np.ascontiguousarray(w)
        ctypes.memmove(words_ptr, w.ctypes.data, w.nbytes)
        ctypes.memmove(counts_ptr, cnt.ctypes.data, cnt.nbytes)

    def stats(self):
        return {"n_overflow_blocks": 0}

    def close(self):
        pass


def _worker(rank, world, port, seed, outdir, use_gpu, backend="gloo"):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    on_dev = backend == "nccl"  # RCCL moves device buffers: the library fills them in place
    # one GPU per rank where the box has them (RCCL refuses two ranks on one device); else device 0
    gpu = rank if (on_dev and world > 1) else 0
    if on_dev:
        torch.cuda.set_device(gpu)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from cases import make_case
    from muscato_amd import Config
    from muscato_amd.dist import HitGatherer, shard_range

    ocfg, reads, targets = make_case(seed)
    cfg = Config(Windows=list(ocfg.Windows), WindowWidth=ocfg.WindowWidth, PMatch=ocfg.PMatch, MinDinuc=ocfg.MinDinuc,
                 MaxReadLength=ocfg.MaxReadLength, MaxMatches=ocfg.MaxMatches, MMTol=ocfg.MMTol, MatchMode=ocfg.MatchMode)
    lo, hi = shard_range(len(reads), rank, world)
    if use_gpu:
        from muscato_amd import Engine
        eng = Engine(gpu)
        eng.load_targets(targets)
    else:
        eng = OracleEngine(ocfg, targets)
    eng.load_reads(reads[lo:hi])
    n0 = eng.match_device(cfg, apply_mmtol=True, n_shards=world)
    dev = torch.device("cuda", gpu) if on_dev else torch.device("cpu")
    out = {}
    for packed in (False, True):
        caps = HitGatherer.agree_caps(n0, dev)
        g = HitGatherer(max(caps), dev, depth=2, packed=packed, rank_caps=caps)
        for _ in range(3):  # three passes: buffers are reused from the third on
            n = eng.match_device(cfg, apply_mmtol=True, n_shards=world)

            def fill(buf, n=n):
                if n and packed:
                    eng.hits_to_packed(buf.data_ptr(), n, on_dev, BITS, lo)
                elif n:
                    eng.hits_to(buf.data_ptr(), n, on_dev)
                return n
            g.submit(fill, lo)
        cn = g.finish()
        # the MaxMatches proof is per shard: the union holds only if every shard reports 0
        o = torch.tensor([int(eng.stats()["n_overflow_blocks"])], dtype=torch.int64, device=dev)
        dist.all_reduce(o, op=dist.ReduceOp.MAX)
        if rank == 0:
            res = g.last_result()
            out["packed" if packed else "plain"] = (cn, (HitGatherer.unpack(res, BITS) if packed else res.to(torch.int64)).cpu(),
                                                    int(o.item()))
    # the compact form: one count byte per read + one u32 word per tuple
    nmax = torch.tensor([eng.n_reads], dtype=torch.int64, device=dev)
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
    caps = HitGatherer.agree_caps(n0, dev)
    g = HitGatherer(max(caps), dev, depth=2, compact_reads=int(nmax.item()), rank_caps=caps)
    for _ in range(3):
        n = eng.match_device(cfg, apply_mmtol=True, n_shards=world)

        def fillc(buf, n=n):
            counts, words = g.compact_views(buf)
            eng.hits_to_compact(words.data_ptr(), g.cap, counts.data_ptr(), g.compact_reads, on_dev, CBITS)
            return n, eng.n_reads
        g.submit(fillc, lo)
    cn = g.finish()
    bases = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(bases, torch.tensor([lo], dtype=torch.int64, device=dev))
    if rank == 0:
        slabs = g.last_slabs()
        res = torch.cat([g.unpack_compact(slabs[r], int(bases[r]), CBITS) for r in range(world)], dim=0)
        out["compact"] = (cn, res.cpu(), 0)
        torch.save(out, os.path.join(outdir, "out.pt"))
    if not use_gpu:
        assert eng.n_shards_seen and all(s == world for s in eng.n_shards_seen)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def _check(tmp_path, seed, use_gpu, world=2, backend="gloo"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import make_case
    from oracle import muscato_oracle as orc
    mp.spawn(_worker, args=(world, _free_port(), seed, str(tmp_path), use_gpu, backend), nprocs=world, join=True)
    out = torch.load(tmp_path / "out.pt")
    ocfg, reads, targets = make_case(seed)
    exp = sorted(orc.best_filter(orc.match_direct(reads, targets, ocfg), ocfg.MMTol))
    exp = torch.tensor(exp, dtype=torch.int64).reshape(-1, 4)
    for form in ("plain", "packed", "compact"):
        cn, res, overflow = out[form]
        assert overflow == 0 and sum(cn) == len(exp)
        # per-shard tuples come out read-major; order within a read is unspecified
        idx = np.lexsort((res[:, 3].numpy(), res[:, 2].numpy(), res[:, 1].numpy(), res[:, 0].numpy()))
        assert torch.equal(res[idx], exp), form
        assert bool((res[1:, 0] >= res[:-1, 0]).all()), "rank-order concatenation must be in global read order"


@pytest.mark.parametrize("seed", [3, 14])
def test_two_rank_engine_shaped_gather_cpu(tmp_path, seed):
    _check(tmp_path, seed, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 14, 27])
def test_two_rank_engines_share_one_gpu(tmp_path, seed):
    _check(tmp_path, seed, use_gpu=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 14])
def test_one_member_rccl_group_with_device_buffers(tmp_path, seed):
    """What the gloo runs cannot reach: backend "nccl" (RCCL) with the gatherer's buffers in device
    memory, filled in place by musc_hits_copy / _packed / _compact (dst_on_device = 1), counts
    agreed by collectives on device tensors, the communicator's stream against the library's.  One
    GPU allows one RCCL rank, so the group has a single member and no transfer has a peer; the
    8-GPU run adds only torch's own grouped send/recv."""
    _check(tmp_path, seed, use_gpu=True, world=1, backend="nccl")


@pytest.mark.gpu
def test_in_process_gather_entry_points():
    """musc_gather / musc_gather_rccl (one process, one context per GPU -- `muscato --GPUs N`, a Go
    host): on the one GPU of the test box the clique has a single member, so this covers the
    packing, rebasing and copy-out paths and that both entry points agree; two contexts on one
    device are refused by the RCCL form (it needs distinct devices) and accepted by the host form."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import make_case
    from muscato_amd import Config, Engine, gather, sorted_hits
    from muscato_amd.api import MuscatoError
    ocfg, reads, targets = make_case(14)
    cfg = Config(Windows=list(ocfg.Windows), WindowWidth=ocfg.WindowWidth, PMatch=ocfg.PMatch, MinDinuc=ocfg.MinDinuc,
                 MaxReadLength=ocfg.MaxReadLength, MaxMatches=ocfg.MaxMatches, MMTol=ocfg.MMTol, MatchMode=ocfg.MatchMode)
    half = len(reads) // 2
    with Engine(0) as a, Engine(0) as b:
        for e, part in ((a, reads[:half]), (b, reads[half:])):
            e.load_targets(targets)
            e.load_reads(part)
            e.match_device(cfg, apply_mmtol=True, n_shards=2)
        whole = gather([a, b], [0, half], rccl=False)
        one = gather([a], [7], rccl=True)
        ref = gather([a], [7], rccl=False)
        assert (one == ref).all() and len(one) == a.stats()["n_hits"] and (len(one) == 0 or one[:, 0].min() >= 7)
        with pytest.raises(MuscatoError, match="share device"):
            gather([a, b], [0, half], rccl=True)
    with Engine(0) as e:
        e.load_targets(targets)
        e.load_reads(reads)
        exp = sorted_hits(e.match(cfg, apply_mmtol=True))
    assert (sorted_hits(whole) == exp).all()


def _n_gpus():
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: HitGatherer over RCCL with a peer")
@pytest.mark.parametrize("seed", [3, 14])
def test_two_rank_rccl_gather_on_two_gpus(tmp_path, seed):
    """The first thing to run on a multi-GPU box: two ranks, one GPU each, backend "nccl" -- the grouped
    send / recv of HitGatherer with a real peer over xGMI, device buffers filled in place by the
    library, in the 16-byte, packed and compact forms, against the single-process list.  (Skipped
    on the one-GPU test boxes; the same code runs there over gloo and as a one-member RCCL group.)"""
    _check(tmp_path, seed, use_gpu=True, world=2, backend="nccl")


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs: musc_gather_rccl with two contexts on two devices")
def test_in_process_rccl_gather_on_two_gpus():
    """musc_gather_rccl with n > 1 (ncclCommInitAll over two devices, grouped ncclSend / ncclRecv to the
    first context's GPU, rebase there, one copy out) against musc_gather on the same contexts."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import make_case
    from muscato_amd import Config, Engine, gather, sorted_hits
    ocfg, reads, targets = make_case(14)
    cfg = Config(Windows=list(ocfg.Windows), WindowWidth=ocfg.WindowWidth, PMatch=ocfg.PMatch, MinDinuc=ocfg.MinDinuc,
                 MaxReadLength=ocfg.MaxReadLength, MaxMatches=ocfg.MaxMatches, MMTol=ocfg.MMTol, MatchMode=ocfg.MatchMode)
    half = len(reads) // 2
    with Engine(0) as a, Engine(1) as b:
        for e, part in ((a, reads[:half]), (b, reads[half:])):
            e.load_targets(targets)
            e.load_reads(part)
            e.match_device(cfg, apply_mmtol=True, n_shards=2)
        via_host = gather([a, b], [0, half], rccl=False)
        lib = a._lib
        import ctypes
        arr = (ctypes.c_void_p * 2)(a._h, b._h)
        bases = (ctypes.c_uint64 * 2)(0, half)
        ph, cnt = ctypes.c_void_p(), ctypes.c_uint64()
        rc = lib.musc_gather_rccl(arr, 2, bases, ctypes.byref(ph), ctypes.byref(cnt))
        assert rc == 0, lib.musc_last_error(a._h).decode()
        via_rccl = np.zeros((cnt.value, 4), dtype=np.uint32)
        if cnt.value:
            ctypes.memmove(via_rccl.ctypes.data, ph, cnt.value * 16)
        lib.musc_free_hits(ph)
    assert via_rccl.shape == via_host.shape and (via_rccl == via_host).all()
