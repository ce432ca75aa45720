"""Build libmatgcn.so (the HIP C-ABI library) in-tree for gfx950.

    python -m multistgraph_amd.build            # rebuild if sources are newer than the .so

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmatgcn.so")
SOURCES = ["matgcn_capi.hip"]
DEPS = ["matgcn_capi.hip", "matgcn_kernels.hip", "matgcn_node16.hip", "matgcn_bwd.hip", "matgcn_bwd_kernels.hip", "matgcn_internal.h",
        os.path.join("..", "..", "include", "matgcn.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def source_id() -> str:
    """12 hex digits over the library's sources: the build a measurement belongs to (bench line, PMC summaries)."""
    import hashlib
    h = hashlib.sha1()
    for d in DEPS:
        with open(os.path.join(CSRC, d), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", *extra_flags,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH]
    if verbose:
        print("[matgcn build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB_PATH


def build_variant(tag: str, extra_flags, verbose: bool = True) -> str:
    """Lab builds (kernel parameter sweeps): lib/libmatgcn_<tag>.so next to the product library, selected at run time
    with MATGCN_LIB=<path> (see _lib.py).  Never used by the product path."""
    os.makedirs(LIB_DIR, exist_ok=True)
    out = os.path.join(LIB_DIR, "libmatgcn_%s.so" % tag)
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           *extra_flags, *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]
    if verbose:
        print("[matgcn build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
