"""Builds glomap_amd/csrc/libgsfm.so (HIP, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as well as on the
MI355X box.  Objects are rebuilt only when a source or header is newer than the object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
INCLUDE = Path(__file__).resolve().parent.parent / "include"
LIB = CSRC / "libgsfm.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found — libgsfm needs the ROCm toolchain")
    return exe


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _headers():
    return sorted(CSRC.glob("*.hpp")) + sorted(INCLUDE.glob("*.h"))


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in _sources() + _headers())


def build_lib(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    hdr_time = max(p.stat().st_mtime for p in _headers())
    objs = []
    jobs = []
    for src in _sources():
        obj = src.with_suffix(".o")
        objs.append(obj)
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_time):
            jobs.append([hipcc, *FLAGS, "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB),
            "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    run(link)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
