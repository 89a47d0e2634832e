"""Host-side binding of the hot path: utils.Config mirror + one Engine per GPU.

The reference's interface for this path is the pair of executables
``muscato_screen config.json`` / ``muscato_confirm config.json k``
(cmd/muscato/main.go:306-316, 387-420) driven by ``utils.Config``
(utils/config.go:10-101).  ``Config`` keeps the same field names, JSON form
and defaults; ``Engine`` is the in-process replacement of the two executables
plus the sort between them, through the C ABI of include/muscato_hip.h.
"""
from __future__ import annotations

import ctypes
import json
from dataclasses import dataclass, field, asdict
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


class MuscatoError(RuntimeError):
    pass


@dataclass
class Config:
    """utils/config.go:10-101 -- same names; defaults of cmd/muscato/main.go:833-904."""
    ReadFileName: str = ""
    GeneFileName: str = ""
    GeneIdFileName: str = ""
    ResultsFileName: str = ""
    Windows: List[int] = field(default_factory=list)
    WindowWidth: int = 0
    BloomSize: int = 0
    NumHash: int = 0
    PMatch: float = 0.0
    MinDinuc: int = 0
    TempDir: str = ""
    LogDir: str = ""
    MinReadLength: int = 0
    MaxReadLength: int = 0
    MaxMatches: int = 0
    MaxConfirmProcs: int = 0
    MMTol: int = 0
    MatchMode: str = ""
    SortPar: int = 0
    SortTemp: str = ""
    SortMem: str = ""
    NoCleanTemp: bool = False
    CPUProfile: bool = False
    MaxMismatch: int = -1  # addition: absolute mismatch budget (overrides PMatch when >= 0)

    @classmethod
    def from_json(cls, text) -> "Config":
        d = json.loads(text) if isinstance(text, (str, bytes)) else dict(text)
        c = cls()
        for k, v in d.items():
            if hasattr(c, k) and v is not None:
                setattr(c, k, v)
        return c

    def to_json(self) -> str:
        return json.dumps(asdict(self))

    def with_defaults(self) -> "Config":
        """checkArgs (cmd/muscato/main.go:833-904) for the fields the hot path reads."""
        c = Config(**asdict(self))
        if not c.Windows:
            raise MuscatoError("Windows not provided")
        if c.WindowWidth == 0:
            raise MuscatoError("WindowWidth not provided")
        if c.MaxReadLength == 0:
            raise MuscatoError("MaxReadLength not provided")
        if c.BloomSize == 0:
            c.BloomSize = 4 * 1000 * 1000 * 1000
        if c.NumHash == 0:
            c.NumHash = 20
        if c.PMatch == 0:
            c.PMatch = 1.0
        if c.MaxMatches == 0:
            c.MaxMatches = 1000 * 1000
        if c.MaxConfirmProcs == 0:
            c.MaxConfirmProcs = 3
        if c.MatchMode == "":
            c.MatchMode = "best"
        if c.MatchMode not in ("best", "first"):
            raise MuscatoError("MatchMode must be 'first' or 'best'")
        if c.ResultsFileName == "":
            c.ResultsFileName = "results.txt"
        return c

    def to_params(self, apply_mmtol: bool, skip_block_check: bool = False, n_shards: int = 1) -> _lib.MuscParams:
        """n_shards > 1: this context sees one of n_shards contiguous slices of the reads, so a
        (window,key) block is split across contexts and the MaxMatches check must use
        MaxMatches / n_shards (include/muscato_hip.h, musc_params.n_shards)."""
        c = self.with_defaults()
        if len(c.Windows) > _lib.MUSC_MAX_WINDOWS:
            raise MuscatoError("at most %d windows are supported" % _lib.MUSC_MAX_WINDOWS)
        p = _lib.MuscParams()
        p.n_windows = len(c.Windows)
        for i, w in enumerate(c.Windows):
            p.windows[i] = int(w)
        p.window_width = int(c.WindowWidth)
        p.pmatch = float(c.PMatch)
        p.min_dinuc = int(c.MinDinuc)
        p.max_read_length = int(c.MaxReadLength)
        p.max_matches = int(c.MaxMatches)
        p.match_mode = 1 if c.MatchMode == "first" else 0
        p.mmtol = int(c.MMTol)
        p.apply_mmtol = 1 if apply_mmtol else 0
        p.max_mismatch_p1 = c.MaxMismatch + 1 if c.MaxMismatch >= 0 else 0
        p.skip_block_check = 1 if skip_block_check else 0
        p.n_shards = max(1, int(n_shards))
        return p


def concat(seqs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    """list of ASCII sequences -> (uint8 buffer, uint64 offsets[n+1])."""
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        off[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(seqs) + b"\0" * 8, dtype=np.uint8).copy()
    return buf, off


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def pack_2bit(buf: np.ndarray, nbases: int) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """ASCII bases -> the packed ABI form: 2 bits per base (A0 C1 G2 T3), 4 bases per
    byte little-endian, plus a 1-bit-per-base X mask (None if there is no X)."""
    codes = _CODE[buf[:nbases]]
    isx = codes == 255
    codes = np.where(isx, 0, codes).astype(np.uint8)
    pad = (-nbases) % 4
    c4 = np.concatenate([codes, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    packed = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
    mask = np.packbits(isx, bitorder="little") if isx.any() else None
    return packed, mask


class Engine:
    """One context per GPU: resident target database + index, resident reads, match()."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self._lib.musc_init(int(device), ctypes.byref(h))
        if rc != 0:
            raise MuscatoError("musc_init failed: %s" % self._lib.musc_last_error(None).decode())
        self._h = h
        self.device = device
        self.n_targets = 0
        self.n_reads = 0

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.musc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise MuscatoError("%s failed (%d): %s" % (what, rc, self._lib.musc_last_error(self._h).decode()))

    # ---- database (gene number = index in the list = line index of GeneFileName)
    def reload_env(self) -> None:
        """Re-read the MUSC_* knobs (the library reads them once, at musc_init)."""
        self._check(self._lib.musc_reload_env(self._h), "musc_reload_env")

    def load_targets(self, seqs: Sequence[bytes]) -> None:
        buf, off = concat(seqs)
        self.load_targets_arrays(buf, off)

    def load_targets_arrays(self, buf: np.ndarray, off: np.ndarray) -> None:
        self._check(self._lib.musc_db_load_ascii(self._h, buf.ctypes.data, off.ctypes.data, len(off) - 1, 0),
                    "musc_db_load_ascii")
        self.n_targets = len(off) - 1

    def load_targets_device(self, seqs_ptr: int, off_ptr: int, nseq: int) -> None:
        self._check(self._lib.musc_db_load_ascii(self._h, seqs_ptr, off_ptr, nseq, 1), "musc_db_load_ascii")
        self.n_targets = nseq

    def load_targets_packed(self, seqs: Sequence[bytes]) -> None:
        buf, off = concat(seqs)
        packed, mask = pack_2bit(buf, int(off[-1]))
        packed = np.concatenate([packed, np.zeros(8, np.uint8)])
        mp = mask.ctypes.data if mask is not None else None
        self._check(self._lib.musc_db_load_packed(self._h, packed.ctypes.data, mp, off.ctypes.data, len(off) - 1),
                    "musc_db_load_packed")
        self.n_targets = len(off) - 1

    def build_index(self, window_width: int) -> None:
        self._check(self._lib.musc_db_build_index(self._h, int(window_width)), "musc_db_build_index")

    def build_index_for(self, cfg: "Config", max_read_len: int = 0) -> None:
        """Build the index match() will pick for `cfg` (context buckets when the run fits them)."""
        p = cfg.to_params(True)
        self._check(self._lib.musc_db_build_index_for(self._h, ctypes.byref(p), int(max_read_len)),
                    "musc_db_build_index_for")

    # ---- reads (already prepared: X-substituted, truncated, unique)
    def load_reads(self, seqs: Sequence[bytes]) -> None:
        buf, off = concat(seqs)
        self.load_reads_arrays(buf, off)

    def load_reads_arrays(self, buf: np.ndarray, off: np.ndarray) -> None:
        self._check(self._lib.musc_reads_load_ascii(self._h, buf.ctypes.data, off.ctypes.data, len(off) - 1, 0),
                    "musc_reads_load_ascii")
        self.n_reads = len(off) - 1

    def load_reads_device(self, seqs_ptr: int, off_ptr: int, nreads: int) -> None:
        self._check(self._lib.musc_reads_load_ascii(self._h, seqs_ptr, off_ptr, nreads, 1), "musc_reads_load_ascii")
        self.n_reads = nreads

    def sort_unique_reads(self, seqs: Sequence[bytes]):
        """Read prep on the GPU (musc_reads_sort_unique): `seqs` are prepared reads in input order.
        Loads the distinct sequences in bytewise order as the context's reads and returns
        (order uint32[n], ustart uint32[n_unique + 1]): distinct sequence g stands for the input
        reads order[ustart[g]:ustart[g+1]]."""
        buf, off = concat(seqs)
        return self.sort_unique_reads_arrays(buf.ctypes.data, off.ctypes.data, len(off) - 1, False)

    def sort_unique_reads_arrays(self, seqs_ptr: int, off_ptr: int, nreads: int, on_device: bool):
        po, pu, nu = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        self._check(self._lib.musc_reads_sort_unique(self._h, seqs_ptr, off_ptr, nreads, 1 if on_device else 0,
                                                     ctypes.byref(po), ctypes.byref(pu), ctypes.byref(nu)),
                    "musc_reads_sort_unique")
        try:
            order = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_uint32)), shape=(max(nreads, 1),))[:nreads].copy()
            ustart = np.ctypeslib.as_array(ctypes.cast(pu, ctypes.POINTER(ctypes.c_uint32)), shape=(nu.value + 1,)).copy()
        finally:
            self._lib.musc_free_u32(po)
            self._lib.musc_free_u32(pu)
        self.n_reads = int(nu.value)
        return order, ustart

    def load_reads_packed(self, seqs: Sequence[bytes]) -> None:
        buf, off = concat(seqs)
        packed, mask = pack_2bit(buf, int(off[-1]))
        packed = np.concatenate([packed, np.zeros(8, np.uint8)])
        mp = mask.ctypes.data if mask is not None else None
        self._check(self._lib.musc_reads_load_packed(self._h, packed.ctypes.data, mp, off.ctypes.data, len(off) - 1),
                    "musc_reads_load_packed")
        self.n_reads = len(off) - 1

    def load_reads_packed_ptr(self, bases_ptr: int, mask_ptr: int, off_ptr: int, nreads: int) -> None:
        """musc_reads_load_packed on caller-owned host buffers (e.g. pinned memory): 2-bit bases,
        optional 1-bit X mask (0 = none), uint64 offsets[nreads + 1] in bases."""
        self._check(self._lib.musc_reads_load_packed(self._h, bases_ptr, mask_ptr or None, off_ptr, nreads),
                    "musc_reads_load_packed")
        self.n_reads = nreads

    def load_reads_packed32_ptr(self, bases_ptr: int, mask_ptr: int, lengths_ptr: int, fixed_len: int, nreads: int,
                                async_upload: bool = False) -> None:
        """musc_reads_load_packed32: 2-bit bases back to back, optional 1-bit X mask (0 = none), uint32
        lengths (0 = every read has fixed_len bases).  async_upload (fixed length, no mask): returns
        once the upload is queued; the next match overlaps it batch by batch -- the host buffer must
        stay valid until that match returns."""
        self._check(self._lib.musc_reads_load_packed32(self._h, bases_ptr, mask_ptr or None, lengths_ptr or None,
                                                       fixed_len, nreads, 1 if async_upload else 0),
                    "musc_reads_load_packed32")
        self.n_reads = nreads

    # ---- hot path
    def match_device(self, cfg: Config, apply_mmtol: bool = True, skip_block_check: bool = False,
                     n_shards: int = 1) -> int:
        """Run screen+confirm(+select); hits stay on the device.  Returns the hit count.
        n_shards = number of read shards the (window,key) blocks are split over (world size of a
        one-process-per-GPU run): the MaxMatches proof then holds for the union of the shards."""
        p = cfg.to_params(apply_mmtol, skip_block_check, n_shards)
        n = ctypes.c_uint64()
        self._check(self._lib.musc_match_device(self._h, ctypes.byref(p), ctypes.byref(n)), "musc_match_device")
        return int(n.value)

    def hits_to(self, ptr: int, capacity: int, on_device: bool) -> None:
        self._check(self._lib.musc_hits_copy(self._h, ptr, capacity, 1 if on_device else 0), "musc_hits_copy")

    def hits_to_packed(self, ptr: int, capacity: int, on_device: bool, bits: Sequence[int], read_base: int = 0) -> None:
        """The last pass's tuples as one u64 each (read+base | gene | pos | nmiss, widths `bits`)."""
        b = (ctypes.c_int32 * 4)(*bits)
        self._check(self._lib.musc_hits_copy_packed(self._h, ptr, capacity, 1 if on_device else 0, read_base, b),
                    "musc_hits_copy_packed")

    def hits_to_compact(self, words_ptr: int, words_cap: int, counts_ptr: int, counts_cap: int, on_device: bool,
                        bits: Sequence[int]) -> None:
        """The last pass's tuples as one u32 word each (gene | pos | nmiss, widths `bits`) plus one
        byte per loaded read = its number of tuples (musc_hits_copy_compact: the list is read-major)."""
        b = (ctypes.c_int32 * 3)(*bits)
        self._check(self._lib.musc_hits_copy_compact(self._h, words_ptr, words_cap, counts_ptr, counts_cap,
                                                     1 if on_device else 0, b), "musc_hits_copy_compact")

    def unpack_hits(self, src_ptr: int, n: int, on_device: bool, bits: Sequence[int], dst_ptr: int) -> None:
        b = (ctypes.c_int32 * 4)(*bits)
        self._check(self._lib.musc_hits_unpack(self._h, src_ptr, n, 1 if on_device else 0, b, dst_ptr), "musc_hits_unpack")

    def match(self, cfg: Config, apply_mmtol: bool = True, n_shards: int = 1) -> np.ndarray:
        """-> uint32 array [n, 4] of (read_idx, gene_idx, pos, nmiss), order unspecified."""
        n = self.match_device(cfg, apply_mmtol, n_shards=n_shards)
        out = np.zeros((n, 4), dtype=np.uint32)
        if n:
            self.hits_to(out.ctypes.data, n, False)
        return out

    def overflow_probes(self) -> np.ndarray:
        """(read_idx, window) of the probes whose (window,key) block may exceed MaxMatches after
        the last match (empty when n_overflow_blocks == 0).  uint32 [n, 2]."""
        pr, pw, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        self._check(self._lib.musc_overflow_probes(self._h, ctypes.byref(pr), ctypes.byref(pw), ctypes.byref(n)),
                    "musc_overflow_probes")
        out = np.zeros((n.value, 2), dtype=np.uint32)
        if n.value:
            out[:, 0] = np.ctypeslib.as_array(ctypes.cast(pr, ctypes.POINTER(ctypes.c_uint32)), shape=(n.value,))
            out[:, 1] = np.ctypeslib.as_array(ctypes.cast(pw, ctypes.POINTER(ctypes.c_uint32)), shape=(n.value,))
        self._lib.musc_free_u32(pr)
        self._lib.musc_free_u32(pw)
        return out

    def stats(self) -> dict:
        s = _lib.MuscStats()
        self._check(self._lib.musc_get_stats(self._h, ctypes.byref(s)), "musc_get_stats")
        return {k: getattr(s, k) for k, _ in s._fields_}


def gather(engines: Sequence["Engine"], read_bases: Sequence[int], rccl: bool = False) -> np.ndarray:
    """Concatenate the device-resident hits of several engines of ONE process in order, adding
    read_bases[i] to the read_idx of engine i (musc_gather: every GPU copies its own to the host;
    rccl=True, opt-in: musc_gather_rccl, over RCCL/xGMI to the first engine's GPU and one copy to the
    host -- any failure there falls back to musc_gather).  uint32 [n, 4]."""
    lib = engines[0]._lib
    n = len(engines)
    arr = (ctypes.c_void_p * n)(*[e._h for e in engines])
    bases = (ctypes.c_uint64 * n)(*[int(b) for b in read_bases])
    ph, cnt = ctypes.c_void_p(), ctypes.c_uint64()
    rc = lib.musc_gather_rccl(arr, n, bases, ctypes.byref(ph), ctypes.byref(cnt)) if rccl else 1
    if rccl and rc == 2:  # a caller's mistake (contexts sharing a device, ...): not something to paper over
        raise MuscatoError("musc_gather_rccl failed (%d): %s" % (rc, lib.musc_last_error(engines[0]._h).decode()))
    if rc != 0:
        rc = lib.musc_gather(arr, n, bases, ctypes.byref(ph), ctypes.byref(cnt))
    if rc != 0:
        raise MuscatoError("musc_gather failed (%d): %s" % (rc, lib.musc_last_error(engines[0]._h).decode()))
    out = np.zeros((cnt.value, 4), dtype=np.uint32)
    if cnt.value:
        ctypes.memmove(out.ctypes.data, ph, cnt.value * 16)
    lib.musc_free_hits(ph)
    return out


def sorted_hits(a: np.ndarray) -> np.ndarray:
    """Canonical order (read, gene, pos, nmiss) for comparisons."""
    if len(a) == 0:
        return a.reshape(0, 4)
    idx = np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))
    return a[idx]
