"""Host logic of the recycled-Ritz-vector preconditioner (glomap_amd/csrc/ritz.hpp), compiled with g++ and checked against
numpy: the tridiagonal eigen-solver, and ritz_select on the Lanczos coefficients of a real PCG run — the Ritz values it
returns are eigenvalues of the preconditioned operator, the combination coefficients turn the recorded z's into vectors u
with u^T A u = theta and A u ~ theta M u, and adding sum u u^T / theta to the preconditioner shortens the next solve of a
PERTURBED system without changing its solution (what cg.hpp's CgRecycle does on the device)."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

MAIN = r'''
#include <cstdio>
#include <cstdlib>
#include "ritz.hpp"
using namespace gsfm;
int main(int argc, char** argv) {
  // mode 0: eigen-decomposition of the tridiagonal matrix in stdin (m, d[m], e[m-1]) -> eigenvalues, then Z row by row
  // mode 1: ritz_select (m, cut, conv, gamma[m+1], alpha[m]) -> k, theta[k], coef[m][8]
  const int mode = std::atoi(argv[1]);
  int m;
  if (std::scanf("%d", &m) != 1) return 2;
  if (mode == 0) {
    std::vector<double> d(m), e(m, 0.0), Z;
    for (int i = 0; i < m; ++i) if (std::scanf("%lf", &d[i]) != 1) return 2;
    for (int i = 0; i + 1 < m; ++i) if (std::scanf("%lf", &e[i]) != 1) return 2;
    if (!tridiag_eig(m, d, e, Z)) return 3;
    for (int i = 0; i < m; ++i) std::printf("%.17g\n", d[i]);
    for (size_t i = 0; i < Z.size(); ++i) std::printf("%.17g\n", Z[i]);
    return 0;
  }
  double cut, conv;
  if (std::scanf("%lf %lf", &cut, &conv) != 2) return 2;
  std::vector<double> gamma(m + 1), alpha(m), coef;
  for (int i = 0; i <= m; ++i) if (std::scanf("%lf", &gamma[i]) != 1) return 2;
  for (int i = 0; i < m; ++i) if (std::scanf("%lf", &alpha[i]) != 1) return 2;
  double theta[kRitzMaxHarvest];
  const int k = ritz_select(m, gamma.data(), alpha.data(), cut, conv, theta, coef);
  std::printf("%d\n", k);
  for (int e = 0; e < k; ++e) std::printf("%.17g\n", theta[e]);
  for (int j = 0; j < m; ++j)
    for (int e = 0; e < kRitzMaxHarvest; ++e) std::printf("%.17g\n", k ? coef[(size_t)j * kRitzMaxHarvest + e] : 0.0);
  // the store's policy, smoke: expire by radius / age, eviction of the largest Ritz value
  RitzStore st;
  for (int j = 0; j < kRitzMaxStore; ++j) { st.used[j] = true; st.theta[j] = 0.01 * (j + 1); st.radius[j] = 100.0; }
  if (st.slot_for(0.5) != -1 || st.slot_for(0.001) != kRitzMaxStore - 1) return 4;
  st.expire(100.0, 3.0, 6);
  if (st.count() != kRitzMaxStore) return 5;
  st.expire(20.0, 3.0, 6);
  if (st.count() != 0) return 6;
  return 0;
}
'''


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    d = tmp_path_factory.mktemp("ritz")
    cc = d / "ritz_main.cc"
    cc.write_text(MAIN)
    out = d / "ritz_main"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", str(ROOT / "glomap_amd" / "csrc"), str(cc), "-o", str(out)], check=True)
    return out


def _run(exe, mode, text):
    r = subprocess.run([str(exe), str(mode)], input=text, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, (r.returncode, r.stderr)
    return np.array([float(v) for v in r.stdout.split()])


@pytest.mark.parametrize("m", [1, 2, 5, 40, 127])
def test_tridiag_eig_matches_numpy(exe, m):
    rng = np.random.default_rng(m)
    d = rng.normal(size=m) + 2.0
    e = rng.normal(size=max(m - 1, 0))
    if m > 10:
        e[m // 2] = 0.0  # a decoupled block
    out = _run(exe, 0, f"{m}\n" + " ".join(f"{v:.17g}" for v in d) + "\n" + " ".join(f"{v:.17g}" for v in e) + "\n")
    lam, Z = out[:m], out[m:].reshape(m, m)
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    assert np.allclose(np.sort(lam), np.linalg.eigvalsh(T), atol=1e-12, rtol=1e-12)
    assert np.allclose(Z.T @ Z, np.eye(m), atol=1e-12)
    assert np.allclose(T @ Z, Z * lam[None, :], atol=1e-11)


def _pcg(A, b, minv, extra=None, tol=1e-10, record=False):
    """PCG with preconditioner z = minv * r (+ sum_j u_j (u_j . r) / theta_j); returns x, iterations, (gamma, alpha, Z)."""
    x = np.zeros_like(b)
    r = b.copy()

    def prec(r_):
        z_ = minv * r_
        if extra is not None:
            U, th = extra
            z_ = z_ + U @ ((U.T @ r_) / th)
        return z_

    z = prec(r)
    p = z.copy()
    gam, alp, Zs = [], [], []
    rz = r @ z
    it = 0
    bn = np.linalg.norm(b)
    while it < 500:
        gam.append(rz)
        Zs.append(z.copy())
        w = A @ p
        a = rz / (p @ w)
        alp.append(a)
        x += a * p
        r -= a * w
        it += 1
        if np.linalg.norm(r) <= tol * bn:
            z = prec(r)
            gam.append(r @ z)
            break
        z = prec(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, it, (np.array(gam), np.array(alp), np.array(Zs))


def test_ritz_select_on_a_real_pcg_run_and_recycling_shortens_the_next_solve(exe):
    rng = np.random.default_rng(7)
    n = 400
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    lam = np.concatenate([[0.012, 0.023, 0.04, 0.06, 0.09], rng.uniform(0.3, 1.9, n - 5)])  # a low tail under a bulk
    S = (Q * lam) @ Q.T
    dsc = rng.uniform(0.5, 2.0, n)
    A = S * np.sqrt(dsc)[:, None] * np.sqrt(dsc)[None, :]  # M^-1 A ~ S: Jacobi does not see the tail
    A = 0.5 * (A + A.T)
    minv = 1.0 / dsc
    b = rng.normal(size=n)
    x, it, (gam, alp, Zs) = _pcg(A, b, minv)
    assert np.allclose(A @ x, b, atol=1e-8)
    m = it - 1
    text = f"{m}\n0.3 0.2\n" + " ".join(f"{v:.17g}" for v in gam[: m + 1]) + "\n" + " ".join(f"{v:.17g}" for v in alp[:m]) + "\n"
    out = _run(exe, 1, text)
    k = int(out[0])
    assert 3 <= k <= 8
    theta = out[1 : 1 + k]
    coef = out[1 + k :].reshape(m, 8)[:, :k]
    ev = np.sort(np.linalg.eigvalsh(np.sqrt(minv)[:, None] * A * np.sqrt(minv)[None, :]))
    assert np.allclose(theta[:3], ev[:3], rtol=1e-4), (theta, ev[:5])
    U = Zs[:m].T @ coef  # what k_cgr_harvest forms on the device
    for e in range(3):
        u = U[:, e]
        assert abs(u @ A @ u - theta[e]) < 1e-6 * theta[e] + 1e-9
        assert np.linalg.norm(A @ u - theta[e] * (u / minv)) < 1e-3 * np.linalg.norm(A @ u)
    # the NEXT system: a perturbed operator and another right-hand side; stale (theta, u) in the preconditioner
    P = rng.normal(size=(n, n)) * 1e-3
    A2 = A + 0.5 * (P + P.T) @ A @ (np.eye(n) + 0.5 * (P + P.T))
    A2 = 0.5 * (A2 + A2.T)
    b2 = rng.normal(size=n)
    x_plain, it_plain, _ = _pcg(A2, b2, minv)
    x_rec, it_rec, _ = _pcg(A2, b2, minv, extra=(U, theta))
    assert np.allclose(x_plain, x_rec, atol=1e-7 * np.abs(x_plain).max())
    assert it_rec < 0.8 * it_plain, (it_rec, it_plain)
