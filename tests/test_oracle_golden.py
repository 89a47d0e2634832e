"""Pin the CPU oracle against the reference's own fixtures.

tests/golden/muscato/00-04 and tests/golden/prep_targets/00-07 are DATA files
copied from the reference's tests/data (inputs + expected outputs driven by its
tests/tests.toml:1-139).  No GPU needed.
"""
import json
import os

import pytest

from oracle import muscato_oracle as orc

# tests/tests.toml:70-139 -- case 04 is prepared with -rev, the others without.
MUSCATO_CASES = [("00", False), ("01", False), ("02", False), ("03", False), ("04", True)]

# tests/tests.toml:1-68 (cases 06 and 07 both run with -rev)
PREP_CASES = [
    ("00", "genes.fasta", False), ("01", "genes.fasta", True),
    ("02", "genes.txt", False), ("03", "genes.txt", True),
    ("04", "genes.txt.gz", False), ("05", "genes.txt.gz", True),
    ("06", "genes.txt.sz", True), ("07", "genes.txt.sz", True),
]


def _read(p):
    with open(p, "rb") as f:
        return f.read()


@pytest.mark.parametrize("case,rev", MUSCATO_CASES)
def test_muscato_fixture(golden_dir, case, rev):
    d = os.path.join(golden_dir, "muscato", case)
    cfg = orc.Config.from_json(json.loads(_read(os.path.join(d, "config.json"))))
    seqs, ids = orc.prep_targets_file(os.path.join(d, "genes.txt"), rev)
    res, nonmatch, _, _ = orc.run_pipeline(_read(os.path.join(d, "reads.fastq")), seqs, ids, cfg)
    assert res == _read(os.path.join(d, "result_e.txt"))
    assert nonmatch == _read(os.path.join(d, "result.nonmatch_e.txt"))


@pytest.mark.parametrize("case,fname,rev", PREP_CASES)
def test_prep_targets_fixture(golden_dir, case, fname, rev):
    d = os.path.join(golden_dir, "prep_targets", case)
    seqs, ids = orc.prep_targets_file(os.path.join(d, fname), rev)
    assert b"".join(s + b"\n" for s in seqs) == _read(os.path.join(d, "expected_sequences.txt"))
    assert b"".join(s + b"\n" for s in ids) == _read(os.path.join(d, "expected_ids.txt"))


def test_nmiss_truncation():
    # cmd/muscato_confirm/main.go:198 in IEEE double: hand-derived values
    assert orc.nmiss_allowed(0.97, 100) == 3
    assert orc.nmiss_allowed(0.93, 100) == 6   # (1-0.93)*100 = 7.000000000000001? no: 6.999999999999996
    assert orc.nmiss_allowed(0.9, 100) == 9    # 9.999999999999998
    assert orc.nmiss_allowed(1.0, 100) == 0
    assert orc.nmiss_allowed(0.8, 10) == 1     # 1.9999999999999996


def test_count_dinuc():
    # utils/entropy.go:5-40
    assert orc.count_dinuc(b"AAAA") == 1
    assert orc.count_dinuc(b"ACGT") == 3
    assert orc.count_dinuc(b"ACAC") == 2
    assert orc.count_dinuc(b"A") == 0
    assert orc.count_dinuc(b"ANNA") == 3  # A-other, other-other, other-A
    assert orc.count_dinuc(b"AXNA") == 3  # X and N are the same letter ("other")
