"""The host tools (C++): `muscato_prep_targets` against the reference's prep_targets fixtures,
the `muscato` flag surface, and -- on the GPU box -- the reference's five end-to-end cases
(tests/tests.toml:70-139) through the real CLI, byte-identical result files."""
import os
import shutil
import subprocess

import pytest

from muscato_amd import build as mbuild
from oracle import muscato_oracle as orc

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "muscato_amd", "bin")


@pytest.fixture(scope="module", autouse=True)
def _built():
    mbuild.build()


def _sz(data: bytes) -> bytes:
    """Framed snappy with uncompressed chunks and a deliberately unchecked (zero) CRC would be
    rejected; build valid chunks with the real masked CRC-32C."""
    import zlib  # noqa: F401  (crc32c is not in zlib: tiny table-driven one here)
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)

    def crc32c(b):
        c = 0xFFFFFFFF
        for x in b:
            c = tab[(c ^ x) & 0xFF] ^ (c >> 8)
        return c ^ 0xFFFFFFFF

    out = b"\xff\x06\x00\x00sNaPpY"
    for p in range(0, len(data), 65536):
        chunk = data[p:p + 65536]
        c = crc32c(chunk)
        m = (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF
        out += b"\x01" + (len(chunk) + 4).to_bytes(3, "little") + m.to_bytes(4, "little") + chunk
    return out


def run(cmd, cwd, **kw):
    return subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


# tests/tests.toml:1-68
PREP_CASES = [
    ("00", "genes.fasta", False), ("01", "genes.fasta", True),
    ("02", "genes.txt", False), ("03", "genes.txt", True),
    ("04", "genes.txt.gz", False), ("05", "genes.txt.gz", True),
    ("06", "genes.txt.sz", True), ("07", "genes.txt.sz", True),
]


@pytest.mark.parametrize("case,fname,rev", PREP_CASES)
def test_prep_targets_tool(golden_dir, tmp_path, case, fname, rev):
    src = os.path.join(golden_dir, "prep_targets", case)
    shutil.copy(os.path.join(src, fname), tmp_path / fname)
    os.chmod(tmp_path / fname, 0o644)
    cmd = [os.path.join(BIN, "muscato_prep_targets")] + (["-rev"] if rev else []) + [fname]
    r = run(cmd, tmp_path)
    assert r.returncode == 0, r.stderr
    stem = fname[:-3] if fname.endswith((".gz", ".sz")) else fname
    for out, exp in (("musc_%s.sz" % stem, "expected_sequences.txt"), ("musc_ids_%s.sz" % stem, "expected_ids.txt")):
        with open(tmp_path / out, "rb") as f:
            got = orc.snappy_framed_decode(f.read())
        with open(os.path.join(src, exp), "rb") as f:
            assert got == f.read()


def test_sz_roundtrip_through_both_codecs(tmp_path):
    # C++ writer -> python reader is covered above; here python-known Go-snappy streams
    # (fixtures 06/07, compressed chunks with copy ops + CRC) -> C++ reader -> C++ writer.
    data = b"gene1\tACGTNNACGT\ngene2\t" + b"ACGT" * 40000 + b"\n"
    (tmp_path / "g.txt").write_bytes(data)
    r = run([os.path.join(BIN, "muscato_prep_targets"), "g.txt"], tmp_path)
    assert r.returncode == 0, r.stderr
    seqs = orc.snappy_framed_decode((tmp_path / "musc_g.txt.sz").read_bytes())
    assert seqs == b"ACGTXXACGT\n" + b"ACGT" * 40000 + b"\n"
    ids = orc.snappy_framed_decode((tmp_path / "musc_ids_g.txt.sz").read_bytes())
    assert ids == b"00000000000\tgene1\t10\n00000000001\tgene2\t160000\n"
    # feed the tool a .sz it wrote itself (>64 KiB: several chunks, CRC verified on read)
    (tmp_path / "h.txt.sz").write_bytes(_sz(b"g1\t" + b"ACGT" * 40000 + b"\n"))
    r = run([os.path.join(BIN, "muscato_prep_targets"), "-rev", "h.txt.sz"], tmp_path)
    assert r.returncode == 0, r.stderr
    seqs = orc.snappy_framed_decode((tmp_path / "musc_h.txt.sz").read_bytes())
    assert seqs == b"ACGT" * 40000 + b"\n" + b"ACGT" * 40000 + b"\n"


def test_cli_usage_and_argument_errors(tmp_path):
    exe = os.path.join(BIN, "muscato")
    r = run([exe, "--help"], tmp_path)
    assert r.returncode == 0 and b"-WindowWidth int" in r.stdout and b"-PMatch float" in r.stdout
    r = run([exe, "-NoSuchFlag=1"], tmp_path)
    assert r.returncode == 2 and b"flag provided but not defined" in r.stderr
    r = run([exe, "-WindowWidth=abc"], tmp_path)
    assert r.returncode == 2
    r = run([exe, "-GeneFileName=g", "-GeneIdFileName=i", "-Windows=0", "-WindowWidth=4", "-MaxReadLength=10"], tmp_path)
    assert r.returncode == 1 and b"ReadFileName not provided" in r.stderr
    r = run([exe, "-ReadFileName=r.fastq", "-GeneFileName=g", "-GeneIdFileName=i", "-Windows=0", "-WindowWidth=4"], tmp_path)
    assert r.returncode == 1 and b"MaxReadLength not provided" in r.stderr


def _stage_case(golden_dir, tmp_path, case, rev):
    """tests/tests.toml runs from the tests/ directory with paths data/muscato/NN/..."""
    d = tmp_path / "data" / "muscato" / case
    shutil.copytree(os.path.join(golden_dir, "muscato", case), d)
    os.chmod(d, 0o755)  # the repo snapshot on the GPU box may be read-only
    for f in d.iterdir():
        os.chmod(f, 0o644)
    r = run([os.path.join(BIN, "muscato_prep_targets")] + (["-rev"] if rev else []) + ["genes.txt"], d)
    assert r.returncode == 0, r.stderr
    # the fixture configs still use the pre-"musc_" names
    shutil.copy(d / "musc_genes.txt.sz", d / "genes.txt.sz")
    shutil.copy(d / "musc_ids_genes.txt.sz", d / "genes_ids.txt.sz")
    return d


def _sort_k5_key(line: bytes) -> bytes:
    """GNU `sort -k5` under LC_ALL=C: the key runs from the blank that precedes field 5 to the
    end of the line (fields are separated by the empty string between a non-blank and a blank)."""
    p = 0
    for _ in range(4):
        while p < len(line) and line[p:p + 1] in (b" ", b"\t"):
            p += 1
        while p < len(line) and line[p:p + 1] not in (b" ", b"\t"):
            p += 1
    return line[p:]


def expected_genestats(results: bytes) -> bytes:
    """cmd/muscato/main.go:94-150 (`sort -k5` piped into muscato_genestats) +
    cmd/muscato_genestats/main.go: runs of equal fifth fields, `gene\\tcount\\t`."""
    lines = sorted(results.splitlines(), key=lambda l: (_sort_k5_key(l), l))
    out, old, n = [], None, 0
    for l in lines:
        g = l.split()[4]
        if old is not None and g != old:
            out.append(old + b"\t%d\t\n" % n)
            n = 0
        old = g
        n += 1
    if old is not None:
        out.append(old + b"\t%d\t\n" % n)
    return b"".join(out)


def expected_readstats(results: bytes) -> bytes:
    """cmd/muscato_readstats/main.go: per run of equal eighth fields the set of fifth fields
    (the reference prints it in Go map order; the CLI sorts it)."""
    out, old, genes = [], None, set()
    for l in results.splitlines():
        f = l.split()
        if len(f) < 8:
            continue
        if old is not None and f[7] != old:
            out.append(old + b"\t" + b"".join(g + b";" for g in sorted(genes)) + b"\n")
            genes = set()
        old = f[7]
        genes.add(f[4])
    if old is not None:
        out.append(old + b"\t" + b"".join(g + b";" for g in sorted(genes)) + b"\n")
    return b"".join(out)


def _check_outputs(d):
    assert (d / "result.txt").read_bytes() == (d / "result_e.txt").read_bytes()
    assert (d / "result.nonmatch.txt.fastq").read_bytes() == (d / "result.nonmatch_e.txt").read_bytes()
    res = (d / "result.txt").read_bytes()
    assert (d / "result_genestats.txt").read_bytes() == expected_genestats(res)
    assert (d / "result_readstats.txt").read_bytes() == expected_readstats(res)


MUSCATO_CASES = [("00", False), ("01", False), ("02", False), ("03", False), ("04", True)]


@pytest.mark.gpu
@pytest.mark.parametrize("case,rev", MUSCATO_CASES)
def test_cli_reference_fixture(golden_dir, tmp_path, case, rev):
    d = _stage_case(golden_dir, tmp_path, case, rev)
    r = run([os.path.join(BIN, "muscato"), "-ConfigFileName=data/muscato/%s/config.json" % case, "--NoCleanTemp"],
            tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    _check_outputs(d)
    logs = list((tmp_path / "muscato_logs").iterdir())
    assert len(logs) == 1 and (logs[0] / "config.json").exists() and (logs[0] / "muscato.log").exists()
    tmps = list((tmp_path / "muscato_tmp").iterdir())
    assert len(tmps) == 1 and (tmps[0] / "reads_sorted.txt.sz").exists()


@pytest.mark.gpu
def test_cli_pure_flags_case_02(golden_dir, tmp_path):
    """tests/tests.toml:100-110: the same run driven by flags only (pins flag parsing)."""
    d = _stage_case(golden_dir, tmp_path, "02", False)
    b = "data/muscato/02/"
    r = run([os.path.join(BIN, "muscato"), "-ReadFileName=" + b + "reads.fastq", "-GeneFileName=" + b + "genes.txt.sz",
             "-GeneIdFileName=" + b + "genes_ids.txt.sz", "-ResultsFileName=" + b + "result.txt", "-Windows=0,5",
             "-WindowWidth=4", "-BloomSize=4000000", "-NumHash=20", "-PMatch=1", "-MinDinuc=1", "-MinReadLength=0",
             "-MaxMatches=1000", "-MaxConfirmProcs=5", "-MaxReadLength=300", "-MatchMode=best", "-MMTol=1"], tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    _check_outputs(d)
    # temp dir removed without --NoCleanTemp
    assert not any((tmp_path / "muscato_tmp").iterdir())


@pytest.mark.gpu
def test_cli_max_mismatch_addition(golden_dir, tmp_path):
    """--MaxMismatch N is the absolute form of PMatch (BASELINE.json's flag surface)."""
    import json
    d = _stage_case(golden_dir, tmp_path, "03", False)
    # one substitution in read1's last base: both windows stay exact, every placement has nmiss=1
    fq = (d / "reads.fastq").read_bytes().replace(b"GTAGGATATC", b"GTAGGATATG")
    (d / "reads.fastq").write_bytes(fq)
    exe = os.path.join(BIN, "muscato")
    base = ["-ConfigFileName=data/muscato/03/config.json", "-MMTol=5"]
    r = run([exe] + base + ["-PMatch=0.8"], tmp_path)      # int(0.2*10) = 1 (1.9999999999999996)
    assert r.returncode == 0, r.stderr.decode()
    a = (d / "result.txt").read_bytes()
    r = run([exe] + base + ["-MaxMismatch=1"], tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    assert (d / "result.txt").read_bytes() == a
    assert b"GTAGGATATG\tGTAGGATATC\t0\t1\tgene3" in a
    r = run([exe] + base, tmp_path)                         # PMatch=1 from the config: read1 is lost
    assert r.returncode == 0 and b"GTAGGATATG" not in (d / "result.txt").read_bytes()
    # and the oracle agrees on the whole file
    ocfg = orc.Config.from_json(json.loads((d / "config.json").read_bytes()))
    ocfg.PMatch, ocfg.MMTol = 0.8, 5
    seqs, ids = orc.prep_targets_file(str(d / "genes.txt"), False)
    res, nonmatch, _, _ = orc.run_pipeline(fq, seqs, ids, ocfg)
    assert a == res


@pytest.mark.gpu
@pytest.mark.parametrize("mode,seed", [("first", 1), ("best", 1), ("first", 2), ("best", 3)])
def test_cli_replays_maxmatches_truncation(tmp_path, mode, seed):
    """cmd/muscato_confirm/main.go:233-242, 424-448: blocks with more than MaxMatches accepted
    pairs are cut order-dependently.  The CLI must reproduce that cut exactly (GPU tuples +
    host replay) -- expected output from the literal oracle."""
    import json
    import random
    from oracle import literal
    rng = random.Random(seed)
    alpha = b"AC"  # two letters: every 4-mer key is shared by many reads and target positions
    targets = [bytes(rng.choice(alpha) for _ in range(rng.randint(20, 40))) for _ in range(30)]
    reads = sorted({bytes(rng.choice(alpha) for _ in range(rng.randint(10, 14))) for _ in range(25)})
    ocfg = orc.Config(Windows=[0, 5], WindowWidth=4, PMatch=0.7, MinDinuc=0, MaxReadLength=50,
                      MaxMatches=6, MMTol=2, MatchMode=mode)
    with pytest.raises(OverflowError):
        orc.match_direct(reads, targets, ocfg)  # the case really overflows
    d = tmp_path
    (d / "genes.txt").write_bytes(b"".join(b"g%d\t%s\n" % (i, t) for i, t in enumerate(targets)))
    (d / "reads.fastq").write_bytes(b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"F" * len(r)) for i, r in enumerate(reads)))
    r = run([os.path.join(BIN, "muscato_prep_targets"), "genes.txt"], d)
    assert r.returncode == 0, r.stderr
    cfg = {"ReadFileName": "reads.fastq", "GeneFileName": "musc_genes.txt.sz", "GeneIdFileName": "musc_ids_genes.txt.sz",
           "ResultsFileName": "result.txt", "Windows": [0, 5], "WindowWidth": 4, "PMatch": 0.7, "MinDinuc": 0,
           "MaxReadLength": 50, "MaxMatches": 6, "MMTol": 2, "MatchMode": mode}
    (d / "config.json").write_text(json.dumps(cfg))
    r = run([os.path.join(BIN, "muscato"), "-ConfigFileName=config.json"], d)
    assert r.returncode == 0, r.stderr.decode()
    assert b"replaying the reference's truncation" in r.stderr
    # expected: the literal oracle's union, then the reference's post-chain
    seqs, ids = orc.prep_targets_file(str(d / "genes.txt"), False)
    ureads = orc.uniqify(orc.prep_reads(orc.read_fastq((d / "reads.fastq").read_bytes()), ocfg))
    hits = literal.match_literal([u.seq for u in ureads], seqs, ocfg, bloom_size=4000000, num_hash=20)
    exp = orc.results_text(hits, ureads, seqs, ids, ocfg)
    got = (d / "result.txt").read_bytes()
    assert got == exp
    # and it differs from what keeping every match would give
    full = orc.results_text(orc.match_direct([u.seq for u in ureads], seqs, ocfg, check_overflow=False), ureads, seqs, ids, ocfg)
    assert full != exp


@pytest.mark.gpu
def test_cli_read_prep_on_gpu_equals_oracle(tmp_path):
    """Duplicated reads with many names (the 1000-character cut), reads that are prefixes of
    others, non-ACGT letters, MinReadLength / MaxReadLength: the CLI's GPU sort + collapse
    (musc_reads_sort_unique) must give the files the oracle gives."""
    import json
    import random
    rng = random.Random(12)
    targets = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(40, 120))) for _ in range(12)]
    pool = []
    for _ in range(40):
        t = rng.choice(targets)
        p = rng.randint(0, len(t) - 20)
        pool.append(t[p:p + rng.randint(8, min(60, len(t) - p))])
    pool += [r[:len(r) // 2] for r in pool[:8]] + [b"ACNNGT" + pool[0], b"acgt" + pool[1]]
    recs = []
    for i in range(400):
        r = pool[0] if i % 3 == 0 else rng.choice(pool)   # one read 130+ times: names exceed 1000 chars
        recs.append((b"@read_%d some description %d" % (rng.randint(0, 10**6), i), r))
    d = tmp_path
    (d / "genes.txt").write_bytes(b"".join(b"g%d\t%s\n" % (i, t) for i, t in enumerate(targets)))
    (d / "reads.fastq").write_bytes(b"".join(b"%s\n%s\n+\n%s\n" % (n, r, b"F" * len(r)) for n, r in recs))
    assert run([os.path.join(BIN, "muscato_prep_targets"), "genes.txt"], d).returncode == 0
    cfg = {"ReadFileName": "reads.fastq", "GeneFileName": "musc_genes.txt.sz", "GeneIdFileName": "musc_ids_genes.txt.sz",
           "ResultsFileName": "result.txt", "Windows": [0, 4], "WindowWidth": 6, "PMatch": 0.9, "MinDinuc": 2,
           "MinReadLength": 10, "MaxReadLength": 50, "MaxMatches": 100000, "MMTol": 1, "MatchMode": "best"}
    (d / "config.json").write_text(json.dumps(cfg))
    r = run([os.path.join(BIN, "muscato"), "-ConfigFileName=config.json", "--NoCleanTemp"], d)
    assert r.returncode == 0, r.stderr.decode()
    tmps = list((d / "muscato_tmp").iterdir())
    out = ((d / "result.txt").read_bytes(), (d / "result.nonmatch.txt.fastq").read_bytes(),
           orc.snappy_framed_decode((tmps[0] / "reads_sorted.txt.sz").read_bytes()))
    assert (d / "result_genestats.txt").read_bytes() == expected_genestats(out[0])
    assert (d / "result_readstats.txt").read_bytes() == expected_readstats(out[0])
    ocfg = orc.Config(Windows=[0, 4], WindowWidth=6, PMatch=0.9, MinDinuc=2, MinReadLength=10, MaxReadLength=50,
                      MaxMatches=100000, MMTol=1, MatchMode="best")
    ureads = orc.uniqify(orc.prep_reads(orc.read_fastq((d / "reads.fastq").read_bytes()), ocfg))
    exp_sorted = b"".join(b"%s\t%d\t%s\n" % (u.seq, u.count, u.names) for u in ureads)
    assert out[2] == exp_sorted
    assert any(u.names.endswith(b"...") for u in ureads)   # the 1000-character rule was exercised
    seqs, ids = orc.prep_targets_file(str(d / "genes.txt"), False)
    hits = orc.best_filter(orc.match_direct([u.seq for u in ureads], seqs, ocfg), ocfg.MMTol)
    assert out[0] == orc.results_text(hits, ureads, seqs, ids, ocfg)


def test_cli_without_gpu_fails_loudly(golden_dir, tmp_path):
    """No GPU -> non-zero exit and a clear message, never a silent CPU path."""
    import ctypes
    from muscato_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    if lib.musc_init(0, ctypes.byref(h)) == 0:
        lib.musc_destroy(h)
        pytest.skip("a GPU is present")
    _stage_case(golden_dir, tmp_path, "00", False)
    r = run([os.path.join(BIN, "muscato"), "-ConfigFileName=data/muscato/00/config.json"], tmp_path)
    assert r.returncode != 0
    assert b"no CPU fallback" in r.stderr or b"HIP" in r.stderr
