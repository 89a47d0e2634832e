"""Databases of 2^32 bases or more (BASELINE configs[4]: 10 Gbp per GPU): 40-bit positions
through the index, the pair descriptors and k_confirm.  4.4 M targets x 1000 bp = 4.4 Gbp
(1.03 x 2^32) generated on the device; reads planted verbatim and with substitutions, most of
them beyond base 2^32.  Properties as in test_gpu_scale.py (no oracle at this size)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_TARGETS, TLEN, N_READS, L = 4_400_000, 1000, 200_000, 100


def test_database_beyond_2_32_bases():
    import torch
    from muscato_amd import Config, Engine, sorted_hits
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(424242)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    T = torch.empty((N_TARGETS, TLEN), dtype=torch.uint8, device=dev)
    for s in range(0, N_TARGETS, 200_000):
        e = min(N_TARGETS, s + 200_000)
        T[s:e] = lut[torch.randint(0, 4, (e - s, TLEN), device=dev, generator=g)]
    toff = torch.arange(0, N_TARGETS + 1, dtype=torch.int64, device=dev) * TLEN
    assert int(toff[-1]) > 2 ** 32

    # reads: 3/4 from the last 5 % of the database (global offsets above 2^32), 1/4 anywhere
    gi = torch.randint(0, N_TARGETS, (N_READS,), device=dev, generator=g)
    hi = torch.randint(int(N_TARGETS * 0.98), N_TARGETS, (N_READS,), device=dev, generator=g)
    gi = torch.where(torch.rand((N_READS,), device=dev, generator=g) < 0.75, hi, gi)
    po = torch.randint(0, TLEN - L + 1, (N_READS,), device=dev, generator=g)
    po[:1000] = 0
    po[1000:2000] = TLEN - L
    R = T.reshape(-1)[(gi * TLEN + po)[:, None] + torch.arange(L, device=dev)[None, :]]
    noisy = torch.rand((N_READS,), device=dev, generator=g) < 0.5
    sub = (torch.rand((N_READS, L), device=dev, generator=g) < 0.01) & noisy[:, None]
    R = torch.where(sub, lut[torch.randint(0, 4, (N_READS, L), device=dev, generator=g)], R)
    roff = torch.arange(0, N_READS + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()

    cfg = Config(Windows=[0, 20], WindowWidth=15, PMatch=0.97, MinDinuc=5, MaxReadLength=100)
    with Engine(0) as eng:
        eng.load_targets_device(T.data_ptr(), toff.data_ptr(), N_TARGETS)
        eng.load_reads_device(R.data_ptr(), roff.data_ptr(), N_READS)
        hits = sorted_hits(eng.match(cfg, apply_mmtol=False))
        st = eng.stats()
    assert st["n_overflow_blocks"] == 0 and len(hits) > 0.8 * N_READS
    h = torch.from_numpy(hits.astype(np.int64)).to(dev)
    # every tuple is real and its nmiss is the true Hamming distance
    assert bool((h[:, 2] + L <= TLEN).all()) and bool((h[:, 3] <= 3).all())
    tsub = T.reshape(-1)[(h[:, 1] * TLEN + h[:, 2])[:, None] + torch.arange(L, device=dev)[None, :]]
    assert bool(((tsub != R[h[:, 0]]).sum(dim=1) == h[:, 3]).all())
    # most tuples lie beyond base 2^32
    assert int(((h[:, 1] * TLEN + h[:, 2]) >= 2 ** 32).sum()) > 0.5 * len(hits)
    # completeness: every read whose source placement has <= 3 mismatches and a usable window
    # is reported at its source (a read with > 3 substitutions is legitimately absent)
    src_mm = (T.reshape(-1)[(gi * TLEN + po)[:, None] + torch.arange(L, device=dev)[None, :]] != R).sum(dim=1)
    key = (h[:, 0] << 40) | (h[:, 1] << 12) | h[:, 2]
    want = (torch.arange(N_READS, device=dev) << 40) | (gi << 12) | po
    found = torch.isin(want, key)
    must = (src_mm == 0) & (po != 0)
    missing = torch.nonzero(must & ~found).reshape(-1).cpu().numpy()
    from oracle import muscato_oracle as orc
    ocfg = orc.Config(Windows=[0, 20], WindowWidth=15, MinDinuc=5, MaxReadLength=100)
    Rc = R.cpu().numpy()
    for m in missing[:100]:
        r = bytes(Rc[m])
        assert not orc.window_valid(r, 0, ocfg) and not orc.window_valid(r, 1, ocfg)
    assert len(missing) < 0.01 * N_READS
