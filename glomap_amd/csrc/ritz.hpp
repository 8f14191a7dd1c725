// ritz.hpp — host side of the recycled-Ritz-vector coarse space of the reduced-system PCG (cg.hpp, CgRecycle).
//
// A PCG solve is a Lanczos process in disguise: with gamma_j = r_j . z_j, alpha_j the step lengths and
// beta_j = gamma_{j+1} / gamma_j, the vectors v_j = (-1)^j z_j / sqrt(gamma_j) are orthonormal in the inner product of the
// preconditioner's inverse and V^T A V = T is tridiagonal,
//     T[j][j] = 1 / alpha_j + beta_{j-1} / alpha_{j-1},      T[j][j+1] = sqrt(beta_j) / alpha_j
// (Saad, Iterative Methods for Sparse Linear Systems, 6.7.3).  Its eigenpairs (theta, y) give Ritz vectors u = V y with
// u^T A u = theta: approximations of the eigenvectors of the preconditioned operator, the extreme ones first.  The small
// ones are what the iteration count of the NEXT solve of the same LM problem goes into, so they are kept
// and added to its preconditioner:  M2^-1 = M^-1 + sum_j u_j u_j^T / theta_j.
// Everything here is O(m^3) on an m <= 128 tridiagonal matrix: microseconds of host time per solve.
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

namespace gsfm {

// Eigen-decomposition of a symmetric tridiagonal matrix by QL sweeps with implicit Wilkinson shifts (the classic scheme of
// EISPACK's tql2).  d[0..m): diagonal in, eigenvalues out (unsorted); e[0..m-1): e[i] = T[i][i+1] in (destroyed; e[m-1]
// unused); Z (m x m, row-major): eigenvectors in its COLUMNS.  False when a sweep does not converge.
inline bool tridiag_eig(int m, std::vector<double>& d, std::vector<double>& e, std::vector<double>& Z) {
  Z.assign((size_t)m * m, 0.0);
  for (int i = 0; i < m; ++i) Z[(size_t)i * m + i] = 1.0;
  if (m == 0) return true;
  e.resize(m);
  e[m - 1] = 0.0;
  const double eps = 2.3e-16;
  for (int l = 0; l < m; ++l) {
    int iter = 0;
    while (true) {
      int mm = l;
      for (; mm < m - 1; ++mm) {
        const double dd = std::fabs(d[mm]) + std::fabs(d[mm + 1]);
        if (std::fabs(e[mm]) <= eps * dd) break;
      }
      if (mm == l) break;
      if (++iter > 80) return false;
      double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
      double r = std::hypot(g, 1.0);
      g = d[mm] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
      double s = 1.0, c = 1.0, p = 0.0;
      int i = mm - 1;
      bool underflow = false;
      for (; i >= l; --i) {
        double f = s * e[i];
        const double b = c * e[i];
        r = std::hypot(f, g);
        e[i + 1] = r;
        if (r == 0.0) {
          d[i + 1] -= p;
          e[mm] = 0.0;
          underflow = true;
          break;
        }
        s = f / r;
        c = g / r;
        g = d[i + 1] - p;
        r = (d[i] - g) * s + 2.0 * c * b;
        p = s * r;
        d[i + 1] = g + p;
        g = c * r - b;
        for (int k = 0; k < m; ++k) {
          double* zk = &Z[(size_t)k * m];
          f = zk[i + 1];
          zk[i + 1] = s * zk[i] + c * f;
          zk[i] = c * zk[i] - s * f;
        }
      }
      if (underflow) continue;
      d[l] -= p;
      e[l] = g;
      e[mm] = 0.0;
    }
  }
  return true;
}

constexpr int kRitzMaxStore = 16;   // = kCgMaxRecycle (cg.hpp)
constexpr int kRitzMaxHarvest = 8;  // Ritz vectors taken from one solve

// What the solver remembers about its store of recycled vectors (the vectors themselves live on the device).
struct RitzStore {
  bool used[kRitzMaxStore] = {};
  double theta[kRitzMaxStore] = {};
  double radius[kRitzMaxStore] = {};  // trust-region radius of the solve the vector was harvested from
  int age[kRitzMaxStore] = {};        // solves since
  int count() const {
    int c = 0;
    for (bool u : used) c += u ? 1 : 0;
    return c;
  }
  void clear() { *this = RitzStore(); }
  // Before a solve at `radius_now`: drop what has gone stale — a vector's Ritz value is the curvature of the system it was
  // harvested from, and with the damping a factor `max_ratio` away (or `max_age` linearisations later) u u^T / theta is no
  // longer a correction of the right size: too small a theta turns a slow mode into an eigenvalue far above the bulk.
  void expire(double radius_now, double max_ratio, int max_age) {
    for (int j = 0; j < kRitzMaxStore; ++j) {
      if (!used[j]) continue;
      const double q = radius[j] / radius_now;
      if (!(q <= max_ratio && q >= 1.0 / max_ratio) || age[j] >= max_age) used[j] = false;
      else ++age[j];
    }
  }
  // A slot for a new vector with Ritz value `th`: a free one, else the slot of the LARGEST stored Ritz value (it matters
  // least) provided the new one is smaller; -1: not worth storing.
  int slot_for(double th) const {
    for (int j = 0; j < kRitzMaxStore; ++j)
      if (!used[j]) return j;
    int worst = 0;
    for (int j = 1; j < kRitzMaxStore; ++j)
      if (theta[j] > theta[worst]) worst = j;
    return theta[worst] > th ? worst : -1;
  }
};

// Ritz pairs of one solve.  gamma[0..m], alpha[0..m-1] as recorded by the vector kernel (m Lanczos steps).  Picks at most
// kRitzMaxHarvest pairs with theta < cut whose residual estimate |T[m][m-1] y_m| is below `conv` * theta, smallest first.
// Out: theta[e] and coef[j * kRitzMaxHarvest + e] = (-1)^j y_j / sqrt(gamma_j), the combination of the recorded z_j that
// forms u_e.  Returns the number of pairs.
inline int ritz_select(int m, const double* gamma, const double* alpha, double cut, double conv, double* theta_out,
                       std::vector<double>& coef) {
  if (m < 3) return 0;
  for (int j = 0; j <= m; ++j)
    if (!(gamma[j] > 0.0) || !std::isfinite(gamma[j])) return 0;
  for (int j = 0; j < m; ++j)
    if (!(alpha[j] > 0.0) || !std::isfinite(alpha[j])) return 0;
  std::vector<double> d(m), e(m, 0.0), Z;
  for (int j = 0; j < m; ++j) {
    d[j] = 1.0 / alpha[j] + (j > 0 ? (gamma[j] / gamma[j - 1]) / alpha[j - 1] : 0.0);
    if (j + 1 < m) e[j] = std::sqrt(gamma[j + 1] / gamma[j]) / alpha[j];
  }
  const double tnext = std::sqrt(gamma[m] / gamma[m - 1]) / alpha[m - 1];  // T[m][m-1]: couples the last kept step to the next
  if (!tridiag_eig(m, d, e, Z)) return 0;
  std::vector<int> order(m);
  for (int i = 0; i < m; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
  coef.assign((size_t)m * kRitzMaxHarvest, 0.0);
  int k = 0;
  for (int oi = 0; oi < m && k < kRitzMaxHarvest; ++oi) {
    const int c = order[oi];
    if (!(d[c] > 0.0)) continue;
    if (d[c] >= cut) break;
    if (std::fabs(tnext * Z[(size_t)(m - 1) * m + c]) > conv * d[c]) continue;  // not converged: not an eigenvector yet
    theta_out[k] = d[c];
    for (int j = 0; j < m; ++j)
      coef[(size_t)j * kRitzMaxHarvest + k] = ((j & 1) ? -1.0 : 1.0) * Z[(size_t)j * m + c] / std::sqrt(gamma[j]);
    ++k;
  }
  return k;
}

}  // namespace gsfm
