"""Register / scratch budgets of the hot kernels, read off the compiler metadata of a device-only gfx950 compile
(tools/kernel_resources.py; no GPU needed).

Why a test: twice in round 3 a change far away from a sweep kernel cost it its occupancy — the fisheye / FOV camera
models inlined into every projecting kernel took k_ba_phaseB from 219 to 278 registers (2 -> 1 wave per SIMD, 130 -> 406 us
per launch on configs[3]) and runtime-indexed Jacobian columns put it on scratch; the deflation arrays made the
1 024-thread single-workgroup update kernel spill.  None of that changes a result, so no parity test notices."""
import pathlib
import re
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent

# kernel (demangled prefix, as tools/kernel_resources.py prints it) -> (max VGPR + AGPR, scratch must be zero)
BUDGET = {
    "k_gp_phaseA<true>": 64,           # (non-temporal tile streams; <false> is the A/B twin)
    "k_gp_phaseB": 96,
    "k_gp_phaseB_x<2>": 72,             # two tiles per wave: 7 waves per SIMD
    "k_gp_wsum": 64,                    # 1 024-thread blocks (16 waves); + the dot products with the recycled Ritz vectors
    "k_ba_phaseA<2, true>": 104,
    "k_ba_phaseB<false>": 224,          # 2 waves per SIMD; the WIDE instance (fisheye / FOV) is allowed 1
    "k_ba_cost<false>": 64,
    "k_ba_build_cam<true, false, 2>": 256,   # compact intrinsics columns: 2 waves per SIMD (full width: 504, one wave)
    "k_ba_lin_track<2, false>": 176,
    "k_ba_lin_cam<false, false>": 208,
    "k_cg_update1<3>": 128,             # __launch_bounds__(1024)
    "k_gj_sweep_step": 80,
    "k_sub_apply3": 72,
}


@pytest.fixture(scope="module")
def resources():
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_resources.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    table = {}
    for line in out.stdout.splitlines():
        m = re.match(r"\s+(.+?)\s+vgpr\s+(\d+) agpr\s+(\d+) sgpr\s+\d+ lds\s+\d+ scratch\s+(\d+) spills\s+(\d+)", line)
        if m:
            table.setdefault(m.group(1).strip(), (int(m.group(2)) + int(m.group(3)), int(m.group(4)), int(m.group(5))))
    return table


@pytest.mark.parametrize("kernel", sorted(BUDGET))
def test_hot_kernel_keeps_its_register_budget(resources, kernel):
    assert kernel in resources, f"{kernel} not found (renamed? update the budget table)"
    regs, scratch, spills = resources[kernel]
    assert scratch == 0 and spills == 0, f"{kernel}: scratch {scratch} B, {spills} spills"
    assert regs <= BUDGET[kernel], f"{kernel}: {regs} registers, budget {BUDGET[kernel]}"
