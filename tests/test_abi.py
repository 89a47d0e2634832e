"""The C-ABI library loads and exports every symbol include/muscato_hip.h declares (no GPU)."""
import os
import re

from muscato_amd import _lib, build as mbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    mbuild.build()
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "muscato_hip.h")) as f:
        hdr = f.read()
    declared = set(re.findall(r"\b(musc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"musc_ctx", "musc_hit", "musc_params", "musc_stats"}
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.musc_abi_version() == 1


def test_struct_layouts_match_header():
    import ctypes
    assert ctypes.sizeof(_lib.MuscHit) == 16
    # n_windows + 16 windows + ww (+pad) + double + 6 ints + 5 reserved
    assert ctypes.sizeof(_lib.MuscParams) == 4 + 64 + 4 + 8 + 4 * 6 + 4 * 5 + 4
    assert ctypes.sizeof(_lib.MuscStats) == 8 * 8 + 2 * 4 + 8 * 4 + 8


def test_init_without_gpu_fails_loudly():
    import ctypes
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.musc_init(0, ctypes.byref(h))
    if rc == 0:  # a GPU is present: fine, clean up
        lib.musc_destroy(h)
        return
    msg = lib.musc_last_error(None).decode()
    assert "no CPU fallback" in msg or "gfx950" in msg or "HIP" in msg
