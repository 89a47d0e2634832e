"""Fuzz of the CLI's MaxMatches replay (not collected by pytest): `python tests/fuzz_cli.py LO HI` runs
test_cli_replays_maxmatches_truncation for more seeds, both match modes.  Round 1: seeds 10..60 clean."""
import os, sys, tempfile, pathlib, traceback
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import test_cli
bad = 0
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    for mode in ("first", "best"):
        with tempfile.TemporaryDirectory() as d:
            try:
                test_cli.test_cli_replays_maxmatches_truncation(pathlib.Path(d), mode, seed)
            except BaseException as e:
                # a seed whose case does not overflow fails the test's own precondition: not a mismatch
                msg = "".join(traceback.format_exception_only(type(e), e)).strip()
                kind = "precondition" if "DID NOT RAISE" in msg or "full != exp" in msg else "MISMATCH"
                if kind == "MISMATCH":
                    bad += 1
                print(kind, "seed", seed, mode, msg[:200], flush=True)
print("fuzz_cli", lo, hi, "bad", bad)
