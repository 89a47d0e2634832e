"""Build libmuscato_hip.so (HIP, gfx950) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmuscato_hip.so")
OBJ = os.path.join(HERE, "build")
# one translation unit for the host side and most kernels, and one per record stride for k_match_t
# (kernels_match_lane_inst.hpp): they compile side by side
SOURCES = [os.path.join(CSRC, f) for f in ("muscato_hip.hip", "match_lane_rw4.hip", "match_lane_rw8.hip", "match_lane_rw12.hip",
                                           "match_lane_rw8w.hip", "match_lane_rw12w.hip", "match_lane_rw16w.hip", "match_lane_rw8s.hip", "match_dma_rw8.hip")]
HEADERS = [os.path.join(os.path.dirname(HERE), "include", "muscato_hip.h")] + \
    [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]
LANE_ONLY = os.path.join(CSRC, "kernels_match_lane.hpp")  # k_match_t's definition: muscato_hip.hip sees its declaration only
DMA_ONLY = os.path.join(CSRC, "kernels_match_dma.hpp")    # k_match_g's definition, likewise
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmuscato_hip.so cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


HOST = os.path.join(CSRC, "host")
BIN = os.path.join(HERE, "bin")
HOST_HEADERS = [os.path.join(HOST, "muscato_host.hpp"), os.path.join(HOST, "sz.hpp")]
TOOLS = {"muscato": "muscato_cli.cpp", "muscato_prep_targets": "muscato_prep_targets.cpp",
         "muscato_screen": "muscato_screen.cpp", "muscato_confirm": "muscato_confirm.cpp"}
GPU_TOOLS = ("muscato", "muscato_screen", "muscato_confirm")


def build_tools(force: bool = False, verbose: bool = False) -> None:
    """The host executables (C++17, g++): `muscato` links libmuscato_hip.so by rpath."""
    os.makedirs(BIN, exist_ok=True)
    for name, src in TOOLS.items():
        out = os.path.join(BIN, name)
        srcp = os.path.join(HOST, src)
        deps = [srcp] + HOST_HEADERS + HEADERS
        if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
            continue
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-o", out, srcp, "-lz"]
        if name in GPU_TOOLS:
            cmd += ["-L" + HERE, "-lmuscato_hip", "-Wl,-rpath,$ORIGIN/.."]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)


def _compile(src: str, force: bool, verbose: bool, defines=(), tag: str = "") -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + tag + ".o")
    base = os.path.basename(src)
    deps = [src] + [h for h in HEADERS if (h != LANE_ONLY or "match_lane" in base or "match_dma" in base) and (h != DMA_ONLY or "match_dma" in base)]
    if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
        return obj
    cmd = [hipcc()] + FLAGS + list(defines) + ["-c", "-o", obj, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        os.makedirs(OBJ, exist_ok=True)
        with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
            objs = list(ex.map(lambda s: _compile(s, force, verbose), SOURCES))
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_tools(force=force, verbose=verbose)
    return LIB


def variant(name: str, defines) -> str:
    """A library whose k_match_t units are compiled with extra -D flags (tuning sessions:
    build_variants/<name>.so, run with MUSC_LIB_PATH); the main unit's object is shared."""
    os.makedirs(OBJ, exist_ok=True)
    out_dir = os.path.join(os.path.dirname(HERE), "build_variants")
    os.makedirs(out_dir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        # (-DMAIN... flags go to the main unit, everything else to the k_match_t units)
        main_defs = ["-D" + d[len("-DMAIN_"):] for d in defines if d.startswith("-DMAIN_")]
        lane_defs = [d for d in defines if not d.startswith("-DMAIN_")]
        objs = list(ex.map(lambda s: _compile(s, False, False, lane_defs if ("match_lane" in s or "match_dma" in s) else main_defs,
                                              "." + name if (("match_lane" in s or "match_dma" in s) and lane_defs) or ("match_lane" not in s and "match_dma" not in s and main_defs) else ""), SOURCES))
    out = os.path.join(out_dir, name + ".so")
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "variant":
        print(variant(sys.argv[2], sys.argv[3:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
