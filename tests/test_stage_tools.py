"""File-level drop-ins: `muscato_screen config.json` and `muscato_confirm config.json k` with the
reference's argv and TempDir file formats (cmd/muscato/main.go:306-316, 387-420).

The stages around them are emulated here at the TEXT level exactly as the reference chains them
(muscato_window_reads -> sort; sort of bmatch_k; a line-by-line restatement of
cmd/muscato_confirm/main.go:171-250), so the test checks that
  * bmatch_k written by our muscato_screen, pushed through the reference's own confirm logic,
    yields the reference result, and
  * rmatch_k written by our muscato_confirm equals what that confirm logic yields."""
import json
import os
import subprocess
from collections import defaultdict

import pytest

from muscato_amd import build as mbuild
from oracle import muscato_oracle as orc

from cases import make_case
from test_cli import _sz, BIN, run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    mbuild.build()


def ref_confirm_text(win_lines, smatch_lines, pmatch):
    """cmd/muscato_confirm/main.go:171-250 on text lines (no MaxMatches overflow here)."""
    src = defaultdict(list)
    for l in win_lines:
        key, left, right = l.split(b"\t")
        src[key].append((left, right))
    out = set()
    for l in smatch_lines:
        key, mlft, mrgt, gene, pos = l.split(b"\t")
        for slft, srgt in src.get(key, ()):
            nmiss = int((1 - pmatch) * float(len(key) + len(slft) + len(srgt)))
            if len(srgt) > len(mrgt):
                continue
            nx = sum(a != b for a, b in zip(mlft, slft)) + sum(a != b for a, b in zip(mrgt[:len(srgt)], srgt))
            if nx > nmiss:
                continue
            out.add(b"\t".join([slft + key + srgt, mlft + key + mrgt[:len(srgt)],
                                b"%d" % (int(pos) - len(mlft)), b"%d" % nx, gene]))
    return out


@pytest.mark.parametrize("seed", [1, 4, 9, 12, 21, 33])
def test_screen_and_confirm_executables(tmp_path, seed):
    ocfg, reads, targets = make_case(seed)
    tmp, logd = tmp_path / "tmp", tmp_path / "logs"
    tmp.mkdir()
    logd.mkdir()
    (tmp_path / "genes.txt.sz").write_bytes(_sz(b"".join(t + b"\n" for t in targets)))
    (tmp / "reads_sorted.txt.sz").write_bytes(_sz(b"".join(r + b"\t1\tn%d\n" % i for i, r in enumerate(reads))))
    cfg = {"ReadFileName": "x.fastq", "GeneFileName": str(tmp_path / "genes.txt.sz"), "GeneIdFileName": "unused",
           "ResultsFileName": "unused", "Windows": ocfg.Windows, "WindowWidth": ocfg.WindowWidth,
           "PMatch": ocfg.PMatch, "MinDinuc": ocfg.MinDinuc, "MaxReadLength": ocfg.MaxReadLength,
           "MaxMatches": ocfg.MaxMatches, "MatchMode": ocfg.MatchMode, "MMTol": ocfg.MMTol,
           "TempDir": str(tmp), "LogDir": str(logd), "BloomSize": 4000000, "NumHash": 20}
    (logd / "config.json").write_text(json.dumps(cfg))

    # ---- muscato_screen
    r = run([os.path.join(BIN, "muscato_screen"), str(logd / "config.json")], tmp_path)
    assert r.returncode == 0, r.stderr.decode()
    assert (logd / "muscato_screen.log").exists()
    ww = ocfg.WindowWidth
    full = orc.match_direct(reads, targets, ocfg)
    union = set()
    per_window = []
    for k, q1 in enumerate(ocfg.Windows):
        bm = [l for l in orc.snappy_framed_decode((tmp / ("bmatch_%d.txt.sz" % k)).read_bytes()).split(b"\n") if l]
        for l in bm:
            f = l.split(b"\t")
            assert len(f) == 5 and len(f[0]) == ww and len(f[3]) == 11
            g, pos = int(f[3]), int(f[4])
            assert targets[g][pos:pos + ww] == f[0]
            assert f[1] == (b"" if pos == 0 else targets[g][pos - q1:pos])
        # muscato_window_reads + sort (cmd/muscato_window_reads/main.go:94-141, cmd/muscato/main.go:237-304)
        win = sorted(rd[q1:q1 + ww] + b"\t" + rd[:q1] + b"\t" + rd[q1 + ww:] for rd in reads if orc.window_valid(rd, k, ocfg))
        (tmp / ("win_%d_sorted.txt.sz" % k)).write_bytes(_sz(b"".join(l + b"\n" for l in win)))
        smatch = sorted(bm)  # sortBloom
        (tmp / ("smatch_%d.txt.sz" % k)).write_bytes(_sz(b"".join(l + b"\n" for l in smatch)))
        rm = ref_confirm_text(win, smatch, ocfg.PMatch)
        per_window.append(rm)
        union |= rm
    exp = {b"\t".join([reads[ri], targets[g][p:p + len(reads[ri])], b"%d" % p, b"%d" % nx, b"%011d" % g])
           for ri, g, p, nx in full}
    assert union == exp  # our bmatch_k through the reference's confirm logic = the reference result

    # ---- muscato_confirm, one process per window like cmd/muscato/main.go:391-419
    for k in range(len(ocfg.Windows)):
        r = run([os.path.join(BIN, "muscato_confirm"), str(logd / "config.json"), str(k)], tmp_path)
        assert r.returncode == 0, r.stderr.decode()
        got = {l for l in orc.snappy_framed_decode((tmp / ("rmatch_%d.txt.sz" % k)).read_bytes()).split(b"\n") if l}
        assert got == per_window[k]
        assert (logd / ("muscato_confirm_%d.log" % k)).exists()


def test_stage_tools_argument_errors(tmp_path):
    for tool in ("muscato_screen", "muscato_confirm"):
        r = run([os.path.join(BIN, tool)], tmp_path)
        assert r.returncode == 1 and b"wrong number of arguments" in r.stderr
