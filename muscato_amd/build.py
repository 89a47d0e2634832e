"""Build libmuscato_hip.so (HIP, gfx950) in-tree with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmuscato_hip.so")
SOURCES = [os.path.join(CSRC, "muscato_hip.hip")]
HEADERS = [os.path.join(os.path.dirname(HERE), "include", "muscato_hip.h")] + \
    [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")]


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


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function", "-o", LIB] + SOURCES + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_tools(force=force, verbose=verbose)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
