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
    assert lib.musc_abi_version() == 3


def test_struct_layouts_match_header():
    import ctypes
    assert ctypes.sizeof(_lib.MuscHit) == 16
    # n_windows + 16 windows + ww (+pad) + double + 6 ints + 5 reserved
    assert ctypes.sizeof(_lib.MuscParams) == 4 + 64 + 4 + 8 + 4 * 6 + 4 * 5 + 4
    assert ctypes.sizeof(_lib.MuscStats) == 8 * 8 + 2 * 4 + 8 * 4 + 8 + 2 * 4 + 4 * 8


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


def test_every_entry_point_has_matching_argtypes():
    """Each prototype of include/muscato_hip.h against the ctypes binding: same number of
    parameters, pointers bound as pointers, 64-bit integers as 64-bit.  A call through a binding
    without argtypes passes Python ints as 32-bit C ints: a pointer or a uint64 count is cut to
    its low half and the library then reads through a wild address on the host (the r01 SIGSEGV
    inside musc_reads_sort_unique, gpurun_out/t54.log: `offsets[nreads]` read through a
    truncated pointer).  This check needs no GPU."""
    import ctypes
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "muscato_hip.h")) as f:
        hdr = re.sub(r"/\*.*?\*/", " ", f.read(), flags=re.S)
    protos = re.findall(r"\b(?:int|void|const char\s*\*)\s+(musc_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", hdr)
    assert {n for n, _ in protos} == set(_lib.SYMBOLS)
    for name, args in protos:
        params = [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"]
        at = getattr(lib, name).argtypes
        assert at is not None or not params, "%s: argtypes not declared" % name
        at = list(at or [])
        assert len(at) == len(params), "%s: %d parameters in the header, %d in the binding" % (name, len(params), len(at))
        for decl, ct in zip(params, at):
            is_ptr = "*" in decl
            bound_ptr = ct is ctypes.c_void_p or ct is ctypes.c_char_p or hasattr(ct, "contents") or \
                issubclass(ct, ctypes._Pointer)
            assert is_ptr == bound_ptr, "%s: `%s` bound as %s" % (name, decl, ct)
            if not is_ptr and "uint64_t" in decl:
                assert ctypes.sizeof(ct) == 8, "%s: `%s` bound as a %d-byte integer" % (name, decl, ctypes.sizeof(ct))


def test_rccl_probe_reports_a_missing_library():
    """rccl_api() called dlerror() twice (the second call returns NULL) and built a
    std::string from it.  With the loader pointed at a file that does not exist the probe must come
    back with code 20 and the loader's message -- in a process of its own, the lookup is cached."""
    import subprocess
    import sys
    code = ("import ctypes\nfrom muscato_amd import _lib\nlib = _lib.load()\nbuf = ctypes.create_string_buffer(256)\n"
            "rc = lib.musc_rccl_probe(buf, 256)\nprint(rc, buf.value.decode())\n")
    env = dict(os.environ, MUSC_RCCL_LIB="/nonexistent/librccl.so.1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, msg = out.stdout.strip().split(" ", 1)
    assert rc == "20" and "cannot load librccl" in msg and "nonexistent" in msg
