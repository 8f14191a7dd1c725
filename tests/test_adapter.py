"""The C++ drop-in adapter (include/gsfm_glomap_adapter.hpp): compiled with g++ against
interface-shaped stand-ins of the GLOMAP headers (tests/adapter/mock) and linked to libgsfm.so.
CPU: it compiles and links.  GPU: the program runs RotationEstimator / GlobalPositioner /
BundleAdjuster — the reference's class names and signatures — end to end from C++ through the C ABI."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "adapter" / "adapter_check.cc"
LIBDIR = ROOT / "glomap_amd" / "csrc"


def _build(tmp_path) -> Path:
    from glomap_amd import build

    build.build_lib(verbose=False)
    exe = tmp_path / "adapter_check"
    cmd = [shutil.which("g++") or "g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'tests' / 'adapter' / 'mock'}",
           f"-I{ROOT / 'include'}", str(SRC), "-o", str(exe), f"-L{LIBDIR}", "-lgsfm", f"-Wl,-rpath,{LIBDIR}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_adapter_compiles_and_links(tmp_path):
    assert _build(tmp_path).exists()


@pytest.mark.gpu
def test_adapter_runs_end_to_end(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "ADAPTER OK" in out.stdout


def test_adapter_numbering_on_the_host(tmp_path):
    """Ascending-id numbering of frames / tracks and the walk orders handed to the library (tests/adapter/pack_check.cc):
    host-only, no libgsfm call."""
    exe = tmp_path / "pack_check"
    cmd = [shutil.which("g++") or "g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{ROOT / 'tests' / 'adapter' / 'mock'}",
           f"-I{ROOT / 'include'}", str(ROOT / "tests" / "adapter" / "pack_check.cc"), "-o", str(exe), f"-L{LIBDIR}", "-lgsfm",
           f"-Wl,-rpath,{LIBDIR}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "PACK OK" in out.stdout
