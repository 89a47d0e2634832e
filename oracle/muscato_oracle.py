"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- direct restatement of muscato's hot path.

This module is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``muscato_amd/`` imports it.

It restates, in plain Python, what the reference Go pipeline computes for

    muscato_prep_targets -> muscato_prep_reads | sort | muscato_uniqify
    -> muscato_window_reads -> muscato_screen -> muscato_confirm
    -> combine_filter | sort -u | combine_windows -> joins -> results.txt

following the set formulation D1-D7 of SURVEY.md section 8(a).  All citations
are file:line of the reference checkout (kshedden/muscato).

Parity pinning: every function here is checked against the reference's own
end-to-end fixtures ``tests/data/muscato/00-04`` and ``tests/data/prep_targets
/00-07`` (copied as data to ``tests/golden``) by ``tests/test_oracle_golden.py``.
The Go binaries cannot be built in this image (no Go toolchain, un-vendored
modules), so behaviour the fixtures do not exercise (nmiss>0, X bases, the
literal-100 rule, MaxMatches overflow) is pinned only by source reading and by
agreement with the independent literal C++ restatement in ``oracle/literal.cpp``.
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Sequence, Set, Tuple

# --------------------------------------------------------------------------
# configuration (utils/config.go:10-101; defaults cmd/muscato/main.go:833-904)
# --------------------------------------------------------------------------


@dataclass
class Config:
    Windows: List[int] = field(default_factory=list)
    WindowWidth: int = 0
    PMatch: float = 1.0
    MinDinuc: int = 0
    MinReadLength: int = 0
    MaxReadLength: int = 0
    MaxMatches: int = 1000 * 1000
    MMTol: int = 0
    MatchMode: str = "best"

    @classmethod
    def from_json(cls, d: dict) -> "Config":
        c = cls()
        for k in ("Windows", "WindowWidth", "PMatch", "MinDinuc", "MinReadLength",
                  "MaxReadLength", "MaxMatches", "MMTol", "MatchMode"):
            if k in d and d[k] not in (None, "", 0, []):
                setattr(c, k, d[k])
        if "PMatch" in d and d["PMatch"] == 0:
            c.PMatch = 1.0  # cmd/muscato/main.go:867-870
        return c


# --------------------------------------------------------------------------
# target preparation (cmd/muscato_prep_targets/main.go)
# --------------------------------------------------------------------------

_COMP = {ord("A"): ord("T"), ord("T"): ord("A"), ord("G"): ord("C"),
         ord("C"): ord("G"), ord("X"): ord("X")}


def subx(seq: bytes) -> bytes:
    """Non-ATGC -> X (cmd/muscato_prep_targets/main.go:68-80, prep_reads :33-44)."""
    return bytes(c if c in b"ATGC" else ord("X") for c in seq)


def revcomp(seq: bytes) -> bytes:
    """cmd/muscato_prep_targets/main.go:48-66.  Bytes outside ATGCX map to 0."""
    return bytes(_COMP.get(c, 0) for c in reversed(seq))


def prep_targets_text(raw: bytes, rev: bool) -> Tuple[List[bytes], List[bytes]]:
    """``id<TAB>seq`` text input -> (sequence lines, id lines).

    cmd/muscato_prep_targets/main.go:82-141: stops at the first empty line;
    gene numbering counts reverse complements; ids are ``%011d\\tname\\tlen``.
    """
    seqs: List[bytes] = []
    ids: List[bytes] = []
    lnum = 0
    for line in raw.split(b"\n"):
        if len(line) == 0:
            break
        toks = line.split(b"\t")
        if len(toks) != 2:
            break  # reference logs and exits (:97-101)
        nam, seq = toks[0], subx(toks[1])
        seqs.append(seq)
        if rev:
            seqs.append(revcomp(seq))
        ids.append(b"%011d\t%s\t%d" % (lnum, nam, len(seq)))
        lnum += 1
        if rev:
            ids.append(b"%011d\t%s_r\t%d" % (lnum, nam, len(seq)))
            lnum += 1
    return seqs, ids


def prep_targets_fasta(raw: bytes, rev: bool) -> Tuple[List[bytes], List[bytes]]:
    """FASTA input (cmd/muscato_prep_targets/main.go:143-213).

    Quirks kept: names keep the leading '>'; the LAST record is not
    ``subx``-ed (:204-212); an empty line would panic in the reference
    (``line[0]``), here it raises IndexError.
    """
    seqs: List[bytes] = []
    ids: List[bytes] = []
    seqname = b""
    seq = b""
    lnum = 0

    def flush(s: bytes, r: bool) -> None:
        seqs.append(s)
        ids.append(b"%011d\t%s%s\t%d" % (len(ids), seqname, b"_r" if r else b"", len(s)))

    lines = raw.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    for line in lines:
        if line[0] == ord(">"):
            if len(seq) > 0:
                seq = subx(seq)
                flush(seq, False)
                lnum += 1
                if rev:
                    flush(revcomp(seq), True)
                    lnum += 1
            seqname = line
            seq = b""
            continue
        seq += line
    if len(seq) > 0:
        flush(seq, False)
        if rev:
            flush(revcomp(seq), True)
    return seqs, ids


def prep_targets_file(path: str, rev: bool) -> Tuple[List[bytes], List[bytes]]:
    """Dispatch on the file name like cmd/muscato_prep_targets/main.go:219-241,321-322.

    ``fasta`` is decided on the *original* name (so ``x.fasta.gz`` is parsed as
    text format -- a reference quirk).
    """
    with open(path, "rb") as f:
        raw = f.read()
    low = path.lower()
    if low.endswith(".gz"):
        raw = gzip.decompress(raw)
    elif low.endswith(".sz"):
        raw = snappy_framed_decode(raw)
    if low.endswith("fasta"):
        return prep_targets_fasta(raw, rev)
    return prep_targets_text(raw, rev)


# --------------------------------------------------------------------------
# snappy framed format decoder (golang/snappy NewReader; format description:
# "Snappy framing format" -- stream id ff 06 00 00 sNaPpY, chunk 0x00 =
# compressed, 0x01 = uncompressed, each with a 4-byte masked CRC32C).
# The CRC is not verified here; the product decoder does verify it.
# --------------------------------------------------------------------------


def _snappy_block_decode(buf: bytes) -> bytes:
    pos = 0
    n = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if b < 0x80:
            break
        shift += 7
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        t = tag & 3
        if t == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if t == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif t == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n, "snappy: length mismatch"
    return bytes(out)


def snappy_framed_decode(raw: bytes) -> bytes:
    pos = 0
    out = bytearray()
    while pos < len(raw):
        ctype = raw[pos]
        clen = int.from_bytes(raw[pos + 1:pos + 4], "little")
        body = raw[pos + 4:pos + 4 + clen]
        pos += 4 + clen
        if ctype == 0xFF:
            assert body == b"sNaPpY"
        elif ctype == 0x00:
            out += _snappy_block_decode(body[4:])
        elif ctype == 0x01:
            out += body[4:]
        elif 0x80 <= ctype <= 0xFE:
            continue  # skippable
        else:
            raise ValueError("snappy: reserved unskippable chunk 0x%02x" % ctype)
    return bytes(out)


# --------------------------------------------------------------------------
# read preparation (utils/fastq.go, cmd/muscato_prep_reads, sort, uniqify)
# --------------------------------------------------------------------------

MAX_NAME_LEN = 1000  # cmd/muscato_prep_reads/main.go:17-19


def read_fastq(raw: bytes) -> List[Tuple[bytes, bytes]]:
    """4-line records, name = whole line 1 (utils/fastq.go:35-61).

    A trailing incomplete record is dropped (Next returns false mid-record).
    """
    lines = raw.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    out = []
    for i in range(0, len(lines) - 3, 4):
        out.append((lines[i], lines[i + 1]))
    return out


def prep_reads(records: Iterable[Tuple[bytes, bytes]], cfg: Config) -> List[bytes]:
    """-> ``seq\\tname`` lines (cmd/muscato_prep_reads/main.go:46-92).

    MinReadLength tests the RAW length; truncation to MaxReadLength after subx;
    names longer than 1000 become first 995 bytes + '...'.
    """
    out = []
    for name, seq in records:
        if len(seq) < cfg.MinReadLength:
            continue
        x = subx(seq)
        if len(x) > cfg.MaxReadLength:
            x = x[:cfg.MaxReadLength]
        rn = name
        if len(rn) > MAX_NAME_LEN:
            rn = rn[:MAX_NAME_LEN - 5] + b"..."
        out.append(x + b"\t" + rn)
    return out


@dataclass
class UniqueRead:
    seq: bytes
    count: int
    names: bytes


def uniqify(lines: Sequence[bytes]) -> List[UniqueRead]:
    """``LC_ALL=C sort`` of ``seq\\tname`` then collapse on seq.

    cmd/muscato/main.go:180-189 (sort), cmd/muscato_uniqify/main.go:83-135:
    names joined with ';', >1000 bytes -> first 996 + '...'; the name is
    ``toks[1]`` only (anything after a second tab is dropped).
    """
    out: List[UniqueRead] = []
    cur = None
    names: List[bytes] = []

    def flush() -> None:
        na = b";".join(names)
        if len(na) > 1000:
            na = na[:996] + b"..."
        out.append(UniqueRead(cur, len(names), na))

    for line in sorted(lines):
        toks = line.split(b"\t")
        if cur is None or toks[0] != cur:
            if cur is not None:
                flush()
            cur = toks[0]
            names = []
        names.append(toks[1])
    if cur is not None:
        flush()
    return out


# --------------------------------------------------------------------------
# the hot path, direct formulation
# --------------------------------------------------------------------------


def count_dinuc(seq: bytes) -> int:
    """utils/entropy.go:5-40: #distinct adjacent pairs over {A,T,G,C,other}."""
    code = {ord("A"): 0, ord("T"): 1, ord("G"): 2, ord("C"): 3}
    seen = set()
    last = 0
    for i, x in enumerate(seq):
        v = code.get(x, 4)
        if i > 0:
            seen.add(5 * last + v)
        last = v
    return len(seen)


def nmiss_allowed(pmatch: float, readlen: int) -> int:
    """cmd/muscato_confirm/main.go:198 -- IEEE double, truncation toward zero."""
    return int((1 - pmatch) * float(readlen))


def window_valid(read: bytes, k: int, cfg: Config) -> bool:
    """D2: cmd/muscato_window_reads/main.go:106-118 (= screen buildBloom :174-185)."""
    q1 = cfg.Windows[k]
    q2 = q1 + cfg.WindowWidth
    if len(read) < q2:
        return False
    return count_dinuc(read[q1:q2]) >= cfg.MinDinuc


def _kmer_index(targets: Sequence[bytes], ww: int) -> Dict[bytes, List[Tuple[int, int]]]:
    idx: Dict[bytes, List[Tuple[int, int]]] = {}
    for g, t in enumerate(targets):
        for jx in range(0, len(t) - ww + 1):
            idx.setdefault(t[jx:jx + ww], []).append((g, jx))
    return idx


Hit = Tuple[int, int, int, int]  # (read_idx, gene_idx, pos, nmiss)


def match_direct(reads: Sequence[bytes], targets: Sequence[bytes], cfg: Config,
                 check_overflow: bool = True) -> Set[Hit]:
    """Union over windows of the tuples muscato_confirm accepts (D3-D7).

    Candidate = equal ww-mer at read offset q1 and target offset jx
    (cmd/muscato_screen/main.go:319-365 through the exact merge-join of
    cmd/muscato_confirm/main.go:375-416), p = jx - q1 >= 0.
    Fit rules (D4): jx >= 1 -> p + len(r) <= len(t)
    (cmd/muscato_screen/main.go:347-353 + cmd/muscato_confirm/main.go:201-203);
    jx == 0 (only q1 == 0) -> len(r) <= min(100 - ww, len(t))
    (cmd/muscato_screen/main.go:294-316, the literal 100).
    nx = Hamming distance over the whole read, X == X
    (cmd/muscato_confirm/main.go:151-159, 205-211).

    MaxMatches truncation (D6) is order dependent; this formulation refuses
    (raises) if any (window, key) block would overflow so that a silent
    divergence is impossible -- the literal C++ oracle handles those.
    """
    ww = cfg.WindowWidth
    idx = _kmer_index(targets, ww)
    hits: Set[Hit] = set()
    for k, q1 in enumerate(cfg.Windows):
        q2 = q1 + ww
        block_count: Dict[bytes, int] = {}
        for ri, r in enumerate(reads):
            if not window_valid(r, k, cfg):
                continue
            key = r[q1:q2]
            nm = nmiss_allowed(cfg.PMatch, len(r))
            for g, jx in idx.get(key, ()):
                t = targets[g]
                p = jx - q1
                if p < 0:
                    continue
                if jx == 0:
                    if len(r) > min(100 - q2, len(t)):
                        continue
                else:
                    if p + len(r) > len(t):
                        continue
                sub = t[p:p + len(r)]
                nx = sum(1 for a, b in zip(r, sub) if a != b)
                if nx > nm:
                    continue
                block_count[key] = block_count.get(key, 0) + 1
                hits.add((ri, g, p, nx))
        if check_overflow:
            for key, c in block_count.items():
                if c > cfg.MaxMatches:
                    raise OverflowError(
                        "window %d key %r: %d accepted pairs > MaxMatches=%d; "
                        "use the literal oracle" % (k, key, c, cfg.MaxMatches))
    return hits


def best_filter(hits: Iterable[Hit], mmtol: int) -> Set[Hit]:
    """Per read keep nx <= best + MMTol (cmd/muscato_combine_windows/main.go:36-60)."""
    best: Dict[int, int] = {}
    hl = list(hits)
    for ri, _, _, nx in hl:
        if ri not in best or nx < best[ri]:
            best[ri] = nx
    return {h for h in hl if h[3] <= best[h[0]] + mmtol}


# --------------------------------------------------------------------------
# post chain to results.txt (cmd/muscato/main.go:422-676)
# --------------------------------------------------------------------------


def results_text(hits: Iterable[Hit], ureads: Sequence[UniqueRead],
                 targets: Sequence[bytes], id_lines: Sequence[bytes], cfg: Config) -> bytes:
    """hits (already the union over windows) -> results.txt bytes.

    sort -u + combine_windows (:453-469), sort -k5 + join with the id file +
    cut (:524-611) replace the 11-digit gene number by ``name\\tlen``; the
    final ``sort -k1`` (:657) orders the 6-column lines bytewise as whole lines
    and ``join`` appends ``count\\tnames`` of reads_sorted.
    """
    kept = best_filter(hits, cfg.MMTol)
    idrest = [ln.split(b"\t", 1)[1] for ln in id_lines]
    lines = []
    for ri, g, p, nx in kept:
        r = ureads[ri].seq
        sub = targets[g][p:p + len(r)]
        six = b"\t".join([r, sub, b"%d" % p, b"%d" % nx, idrest[g]])
        lines.append((six, ri))
    lines.sort(key=lambda x: x[0])
    out = bytearray()
    for six, ri in lines:
        out += six + b"\t%d\t" % ureads[ri].count + ureads[ri].names + b"\n"
    return bytes(out)


def nonmatch_text(result_text: bytes, ureads: Sequence[UniqueRead]) -> bytes:
    """cmd/muscato_nonmatch/main.go:95-114 (exact set instead of the Bloom filter)."""
    matched = {ln.split()[0] for ln in result_text.split(b"\n") if ln.strip()}
    out = bytearray()
    for u in ureads:
        if u.seq in matched:
            continue
        f = (u.seq + b"\t%d\t" % u.count + u.names).split()
        out += f[2] + b"#" + f[1] + b"\n" + f[0] + b"\n+\n" + b"!" * len(f[0]) + b"\n"
    return bytes(out)


def run_pipeline(fastq_raw: bytes, target_seqs: Sequence[bytes], id_lines: Sequence[bytes],
                 cfg: Config) -> Tuple[bytes, bytes, List[UniqueRead], Set[Hit]]:
    """Whole reference pipeline on in-memory inputs -> (results, nonmatch, ureads, hits)."""
    ureads = uniqify(prep_reads(read_fastq(fastq_raw), cfg))
    seqs = [u.seq for u in ureads]
    # window_reads exits(1) if a window has no read long enough
    # (cmd/muscato_window_reads/main.go:143-151)
    for k, q1 in enumerate(cfg.Windows):
        if not any(len(s) >= q1 + cfg.WindowWidth for s in seqs):
            raise RuntimeError("Window %d produced no valid reads, exiting" % k)
    hits = match_direct(seqs, target_seqs, cfg)
    res = results_text(hits, ureads, target_seqs, id_lines, cfg)
    return res, nonmatch_text(res, ureads), ureads, hits
