"""ctypes binding of the literal C++ oracle (TEST INFRASTRUCTURE ONLY).

See oracle/literal.cpp.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_muscato.so")
ORC_MAX_WINDOWS = 32


class OrcParams(ctypes.Structure):
    _fields_ = [
        ("n_windows", ctypes.c_int32),
        ("windows", ctypes.c_int32 * ORC_MAX_WINDOWS),
        ("window_width", ctypes.c_int32),
        ("pmatch", ctypes.c_double),
        ("min_dinuc", ctypes.c_int32),
        ("max_read_length", ctypes.c_int32),
        ("max_matches", ctypes.c_int32),
        ("match_mode_first", ctypes.c_int32),
        ("bloom_size", ctypes.c_uint64),
        ("num_hash", ctypes.c_int32),
        ("nthreads", ctypes.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "literal.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle_muscato.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_match.restype = ctypes.c_int
        _lib.orc_match.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
            ctypes.POINTER(OrcParams), ctypes.POINTER(ctypes.c_void_p),
            ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_void_p]
        _lib.orc_free.argtypes = [ctypes.c_void_p]
        _lib.orc_count_dinuc.argtypes = [ctypes.c_char_p, ctypes.c_int]
    return _lib


def concat(seqs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8).copy()
    return buf, off


def make_params(cfg, bloom_size: int = 4_000_000, num_hash: int = 20, nthreads: int = 1) -> OrcParams:
    p = OrcParams()
    p.n_windows = len(cfg.Windows)
    for i, w in enumerate(cfg.Windows):
        p.windows[i] = w
    p.window_width = cfg.WindowWidth
    p.pmatch = cfg.PMatch
    p.min_dinuc = cfg.MinDinuc
    p.max_read_length = cfg.MaxReadLength
    p.max_matches = cfg.MaxMatches
    p.match_mode_first = 1 if cfg.MatchMode == "first" else 0
    p.bloom_size = bloom_size
    p.num_hash = num_hash
    p.nthreads = nthreads
    return p


def match_arrays(rbuf: np.ndarray, roff: np.ndarray, gbuf: np.ndarray, goff: np.ndarray,
                 params: OrcParams):
    """-> (hits[n,4] uint32 sorted unique, timings[5] seconds, counts[3])."""
    L = lib()
    out = ctypes.c_void_p()
    n = ctypes.c_uint64()
    tim = np.zeros(5, dtype=np.float64)
    cnt = np.zeros(3, dtype=np.uint64)
    rc = L.orc_match(rbuf.ctypes.data, roff.ctypes.data, len(roff) - 1,
                     gbuf.ctypes.data, goff.ctypes.data, len(goff) - 1,
                     ctypes.byref(params), ctypes.byref(out), ctypes.byref(n),
                     tim.ctypes.data, cnt.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_match failed: rc=%d" % rc)
    if n.value:
        arr = np.ctypeslib.as_array(ctypes.cast(out, ctypes.POINTER(ctypes.c_uint32)),
                                    shape=(n.value, 4)).copy()
    else:
        arr = np.zeros((0, 4), np.uint32)
    L.orc_free(out)
    return arr, tim, cnt


def match_literal(reads: Sequence[bytes], targets: Sequence[bytes], cfg, **kw) -> List[Tuple[int, int, int, int]]:
    rbuf, roff = concat(reads)
    gbuf, goff = concat(targets)
    arr, _, _ = match_arrays(rbuf, roff, gbuf, goff, make_params(cfg, **kw))
    return [tuple(int(x) for x in row) for row in arr]
