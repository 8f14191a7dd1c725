// colmap/geometry/pose.h + util/types.h belong to the un-vendored COLMAP dependency; the two functions the rotation averaging
// sources call, restated from their published definitions:
//   AverageQuaternions(quats, weights)  principal eigenvector of sum_i w_i q_i q_i^T / sum_i w_i (Markley et al.), one
//                                        quaternion returned as it is; the sign of an eigenvector is free, the rotation is not
//   ImagePairToPairId(a, b)             kMaxNumImages * min + max with kMaxNumImages = 2^31 - 1
#pragma once
#include <cmath>
#include <vector>

#include "ref_shim_linalg.h"

namespace colmap {
inline glomap::image_pair_t ImagePairToPairId(glomap::image_t image_id1, glomap::image_t image_id2) {
  constexpr uint64_t kMaxNumImages = 2147483647ull;
  return image_id1 > image_id2 ? kMaxNumImages * image_id2 + image_id1 : kMaxNumImages * image_id1 + image_id2;
}

inline Eigen::Quaterniond AverageQuaternions(const std::vector<Eigen::Quaterniond>& quats, const std::vector<double>& weights) {
  if (quats.size() == 1) return quats[0];
  double A[4][4] = {{0}}, V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  double wsum = 0.0;
  for (size_t i = 0; i < quats.size(); ++i) {
    const Eigen::Quaterniond& q = quats[i];
    const double nq = std::sqrt(q.w() * q.w() + q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
    const double c[4] = {q.x() / nq, q.y() / nq, q.z() / nq, q.w() / nq};  // coeffs(): x, y, z, w
    for (int r = 0; r < 4; ++r)
      for (int s = 0; s < 4; ++s) A[r][s] += weights[i] * c[r] * c[s];
    wsum += weights[i];
  }
  for (auto& r : A)
    for (double& x : r) x /= wsum;
  for (int sweep = 0; sweep < 60; ++sweep) {  // cyclic Jacobi on the symmetric 4 x 4
    double off = 0.0;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
        for (int k = 0; k < 4; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
        for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; ++k)
    if (A[k][k] > A[best][best]) best = k;
  return Eigen::Quaterniond(V[3][best], V[0][best], V[1][best], V[2][best]);
}
}  // namespace colmap
