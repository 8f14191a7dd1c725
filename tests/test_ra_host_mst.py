"""Host logic of the rotation-averaging start (glomap_amd/csrc/ra.hip: mst_init — maximum spanning tree on #inliers + BFS
propagation, gra.cc:87-138 / tree.cc:78-153): the function's own text is compiled with g++ twice — as it stands (Kruskal stops
once the tree spans) and with the early exit removed (every edge scanned) — and both must give bit-identical rotations on
connected and disconnected graphs, any root, ties in the inlier counts."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
MAIN = r'''
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>
namespace early { %s }
namespace full { %s }
int main() {
  std::mt19937 rng(1);
  for (int trial = 0; trial < 12; ++trial) {
    const int N = trial < 8 ? 50 + trial * 37 : 4000;
    const long deg = trial < 8 ? 3 + trial : 40;
    const bool disconnected = trial %% 3 == 2;
    std::vector<int> ei, ej, ninl;
    std::vector<double> eq;
    for (int i = 0; i < N; ++i)
      for (int k = 1; k <= deg; ++k) {
        const int j = (i + k) %% N;
        if (disconnected && (i < N / 2) != (j < N / 2)) continue;
        ei.push_back(trial %% 2 ? j : i);
        ej.push_back(trial %% 2 ? i : j);
        ninl.push_back(30 + (int)(rng() %% (trial == 5 ? 3 : 470)));  // trial 5: almost all counts tie
        double q[4], n = 0;
        for (double& c : q) { c = std::normal_distribution<double>()(rng); n += c * c; }
        for (double c : q) eq.push_back(c / std::sqrt(n));
      }
    const long E = (long)ei.size();
    std::vector<double> r1(3 * N, 0.123), r2(3 * N, 0.123);
    const int root = trial %% 4 == 1 ? N / 3 : 0;
    early::mst_init(N, E, ei.data(), ej.data(), eq.data(), ninl.data(), r1.data(), root);
    full::mst_init(N, E, ei.data(), ej.data(), eq.data(), ninl.data(), r2.data(), root);
    if (std::memcmp(r1.data(), r2.data(), sizeof(double) * 3 * N) != 0) return std::printf("trial %%d differs\n", trial), 1;
  }
  std::printf("MST OK\n");
  return 0;
}
'''


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_kruskal_early_exit_gives_the_same_tree_and_rotations(tmp_path):
    src = (ROOT / "glomap_amd" / "csrc" / "ra.hip").read_text()
    body = src[src.index("void mst_init(int N, long E"):src.index("struct RaDevice {")]
    full, n = re.subn(r"\n[^\n]*\n[^\n]*\n\s*if \(\+\+tree_edges == N - 1\) break;", "", body)
    assert n == 1 and "break;" not in full.split("std::vector<Q> q(N")[0].split("for (long e : order)")[1]
    cc = tmp_path / "mst.cc"
    cc.write_text(MAIN % (body, full))
    exe = tmp_path / "mst"
    subprocess.run(["g++", "-O2", "-std=c++17", str(cc), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "MST OK" in out.stdout, out.stdout + out.stderr
