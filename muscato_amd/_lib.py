"""ctypes loader of libmuscato_hip.so.  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MUSC_LIB_PATH: another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("MUSC_LIB_PATH") or os.path.join(HERE, "libmuscato_hip.so")

MUSC_MAX_WINDOWS = 16


class MuscHit(ctypes.Structure):
    _fields_ = [("read_idx", ctypes.c_uint32), ("gene_idx", ctypes.c_uint32),
                ("pos", ctypes.c_uint32), ("nmiss", ctypes.c_uint32)]


class MuscParams(ctypes.Structure):
    _fields_ = [
        ("n_windows", ctypes.c_int32),
        ("windows", ctypes.c_int32 * MUSC_MAX_WINDOWS),
        ("window_width", ctypes.c_int32),
        ("pmatch", ctypes.c_double),
        ("min_dinuc", ctypes.c_int32),
        ("max_read_length", ctypes.c_int32),
        ("max_matches", ctypes.c_int32),
        ("match_mode", ctypes.c_int32),
        ("mmtol", ctypes.c_int32),
        ("apply_mmtol", ctypes.c_int32),
        ("max_mismatch_p1", ctypes.c_int32),
        ("skip_block_check", ctypes.c_int32),
        ("n_shards", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 2),
    ]


class MuscStats(ctypes.Structure):
    _fields_ = [
        ("n_reads", ctypes.c_uint64), ("n_read_windows", ctypes.c_uint64),
        ("n_candidates", ctypes.c_uint64), ("n_pairs", ctypes.c_uint64), ("n_accepted", ctypes.c_uint64),
        ("n_hits", ctypes.c_uint64), ("n_overflow_blocks", ctypes.c_uint64),
        ("confirm_bytes", ctypes.c_uint64),
        ("confirm_launches", ctypes.c_uint32), ("n_batches", ctypes.c_uint32),
        ("ms_screen", ctypes.c_float), ("ms_scan", ctypes.c_float), ("match_variant", ctypes.c_uint32),
        ("ms_confirm", ctypes.c_float), ("ms_select", ctypes.c_float), ("ms_total", ctypes.c_float),
        ("ms_index_build", ctypes.c_float), ("ms_read_prep", ctypes.c_float),
        ("n_descriptors", ctypes.c_uint64),
        ("index_kind", ctypes.c_uint32), ("match_launches", ctypes.c_uint32),
        ("n_overflow_entries", ctypes.c_uint64), ("match_bytes", ctypes.c_uint64),
        ("match_bytes_strict", ctypes.c_uint64), ("index_bytes", ctypes.c_uint64),
    ]


# every symbol include/muscato_hip.h declares
SYMBOLS = [
    "musc_abi_version", "musc_init", "musc_destroy", "musc_last_error", "musc_reload_env",
    "musc_db_load_ascii", "musc_db_load_packed", "musc_db_build_index", "musc_db_build_index_for",
    "musc_reads_load_ascii", "musc_reads_load_packed", "musc_reads_load_packed32", "musc_reads_sort_unique",
    "musc_match_device", "musc_hits_copy", "musc_hits_copy_packed", "musc_hits_copy_compact", "musc_hits_unpack", "musc_match", "musc_free_hits",
    "musc_get_stats", "musc_gather", "musc_gather_rccl", "musc_rccl_probe", "musc_overflow_probes", "musc_free_u32",
]

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library; raise (never fall back) if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "muscato_amd: %s is missing -- build it with `python -m muscato_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    # torch ships a HIP runtime of its own; this library links the system one.  Both work in one
    # process when torch's opens the device first, and torch reports "no ROCm-capable device" when it
    # comes second.  A process that has torch loaded gets that order here, whatever the caller does
    # next (the library itself never imports torch).
    import sys
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise ImportError("muscato_amd: %s does not export %s" % (LIB_PATH, s))
    vp, u64, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32
    lib.musc_abi_version.restype = ctypes.c_int
    lib.musc_init.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.musc_destroy.argtypes = [vp]
    lib.musc_destroy.restype = None
    lib.musc_last_error.argtypes = [vp]
    lib.musc_last_error.restype = ctypes.c_char_p
    lib.musc_reload_env.argtypes = [vp]
    lib.musc_db_load_ascii.argtypes = [vp, vp, vp, ctypes.c_uint32, ctypes.c_int]
    lib.musc_db_load_packed.argtypes = [vp, vp, vp, vp, ctypes.c_uint32]
    lib.musc_db_build_index.argtypes = [vp, i32]
    lib.musc_db_build_index_for.argtypes = [vp, ctypes.POINTER(MuscParams), i32]
    lib.musc_reads_load_ascii.argtypes = [vp, vp, vp, u64, ctypes.c_int]
    lib.musc_reads_load_packed.argtypes = [vp, vp, vp, vp, u64]
    lib.musc_reads_load_packed32.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, u64, ctypes.c_int]
    lib.musc_reads_sort_unique.argtypes = [vp, vp, vp, u64, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                           ctypes.POINTER(u64)]
    lib.musc_match_device.argtypes = [vp, ctypes.POINTER(MuscParams), ctypes.POINTER(u64)]
    lib.musc_hits_copy.argtypes = [vp, vp, u64, ctypes.c_int]
    lib.musc_hits_copy_packed.argtypes = [vp, vp, u64, ctypes.c_int, u64, ctypes.POINTER(i32)]
    lib.musc_hits_copy_compact.argtypes = [vp, vp, u64, vp, u64, ctypes.c_int, ctypes.POINTER(i32)]
    lib.musc_hits_unpack.argtypes = [vp, vp, u64, ctypes.c_int, ctypes.POINTER(i32), vp]
    lib.musc_match.argtypes = [vp, ctypes.POINTER(MuscParams), ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.musc_free_hits.argtypes = [vp]
    lib.musc_free_hits.restype = None
    lib.musc_get_stats.argtypes = [vp, ctypes.POINTER(MuscStats)]
    lib.musc_overflow_probes.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.musc_overflow_probes.restype = ctypes.c_int
    lib.musc_free_u32.argtypes = [vp]
    lib.musc_free_u32.restype = None
    lib.musc_gather.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(u64),
                                ctypes.POINTER(vp), ctypes.POINTER(u64)]
    lib.musc_gather_rccl.argtypes = lib.musc_gather.argtypes
    lib.musc_rccl_probe.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
    for name in ("musc_init", "musc_reload_env", "musc_db_load_ascii", "musc_db_load_packed", "musc_db_build_index", "musc_db_build_index_for", "musc_db_build_index_for",
                 "musc_reads_load_ascii", "musc_reads_load_packed", "musc_reads_load_packed32", "musc_reads_sort_unique", "musc_match_device",
                 "musc_hits_copy", "musc_hits_copy_packed", "musc_hits_copy_compact", "musc_hits_unpack", "musc_match", "musc_get_stats",
                 "musc_gather", "musc_gather_rccl", "musc_rccl_probe"):
        getattr(lib, name).restype = ctypes.c_int
    _lib = lib
    return lib
