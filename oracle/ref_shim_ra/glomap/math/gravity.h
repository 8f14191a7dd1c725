#pragma once
// glomap/math/gravity.cc needs Eigen's Householder QR and JacobiSVD for GetAlignRot / AverageGravity, which the rotation
// averaging sources do not call (the alignment matrices are handed in ready-made).  The two one-liners they do call:
#include "glomap/math/rigid3d.h"

namespace glomap {
inline double RotUpToAngle(const Eigen::Matrix3d& R_up) { return RotationToAngleAxis(R_up)[1]; }   // gravity.cc:25-27
inline Eigen::Matrix3d AngleToRotUp(double angle) { return AngleAxisToRotation(Eigen::Vector3d(0, angle, 0)); }  // gravity.cc:29-32
}  // namespace glomap
