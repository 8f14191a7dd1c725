// ORACLE (test infrastructure only — never linked into or called by the product path).
//
// orc_ba.cc — multithreaded C++ CPU restatement of GLOMAP's global bundle adjustment for trivial
// rigs.  Same algorithm as oracle/ba.py (cross-validated against it on small problems), written so
// that it runs at the full benchmark size (10k cameras / 5M observations) on all host cores:
//
//   problem      BundleAdjuster::AddPointToCameraConstraints, glomap/estimators/bundle_adjustment.cc:115-190
//                (tracks >= min_num_view_per_track :122)
//   residual     colmap::ReprojErrorCostFunctor<CameraModel> via bundle_adjustment.cc:137-146 (COLMAP @ b6b7b54e
//                is un-vendored; restated from its published definition):
//                x_c = R(q) X + t; (u, v) = CameraModel::ImgFromCam(params, x_c); r = (u, v) - obs  [pixels];
//                residual and Jacobian are zero when the point is not in front of the camera
//   models       SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV, OPENCV_FISHEYE, FOV, SIMPLE_RADIAL_FISHEYE,
//                RADIAL_FISHEYE (colmap/sensor/models.h; the models with at most 8 parameters); FULL_OPENCV, THIN_PRISM_FISHEYE
//                and RAD_TAN_THIN_PRISM_FISHEYE in the 16-wide unit (orc_ba_wide.cc compiles this file with ORC_BA_MAXP 16)
//   manifolds    bundle_adjustment.cc:244-317: EigenQuaternionManifold (q <- [sin|d| d/|d|, cos|d|] * q), first frame
//                constant, optimize_rotations / optimize_translation, principal point frozen by a SubsetManifold
//   loss         Huber(1 px), bundle_adjustment.h:30,34-36
//   ordering     points first (bundle_adjustment.cc:204-208)
//   solver       Ceres LM (orc_lm.hpp); the 3x3 point blocks are eliminated in closed form (what SPARSE_SCHUR does)
//                and the reduced system over poses + intrinsics is solved by block-Jacobi PCG to 1e-14
//
// parity unpinned: compared through converged solutions.
#include "orc_lm.hpp"

namespace orc {

// This file is compiled twice into liboracle_cpu.so: as it stands (intrinsics blocks of MAXP = 8 doubles, entry orc_ba_solve)
// and through orc_ba_wide.cc with ORC_BA_MAXP 16 (entry orc_ba_solve_wide) for the camera models with more than eight
// parameters — FULL_OPENCV, THIN_PRISM_FISHEYE (12), RAD_TAN_THIN_PRISM_FISHEYE (16); everything but the two option / report
// structs (identical in both units) is in the anonymous namespace below.
#ifndef ORC_BA_MAXP
#define ORC_BA_MAXP 8
#define ORC_BA_ENTRY orc_ba_solve
#endif
constexpr int MAXP = ORC_BA_MAXP;  // doubles per intrinsics block
constexpr int MP2 = 2 * MAXP;      // a 2 x MAXP Jacobian row pair
constexpr int MPP = MAXP * MAXP;
constexpr int BJ = 6 + MAXP;       // joint pose + intrinsics block of the preconditioner
constexpr int BJ2 = BJ * BJ;
enum { SIMPLE_PINHOLE = 0, PINHOLE = 1, SIMPLE_RADIAL = 2, RADIAL = 3, OPENCV = 4, OPENCV_FISHEYE = 5, FULL_OPENCV = 6, FOV = 7,
       SIMPLE_RADIAL_FISHEYE = 8, RADIAL_FISHEYE = 9, THIN_PRISM_FISHEYE = 10, RAD_TAN_THIN_PRISM_FISHEYE = 11 };  // COLMAP's CameraModelId values
constexpr int kNumModels = 12;
// (-1 in the 8-wide unit: the models with more than 8 parameters)
const int kNumParams[kNumModels] = {3, 4, 4, 5, 8, 8, MAXP >= 16 ? 12 : -1, 5, 4, 5, MAXP >= 16 ? 12 : -1, MAXP >= 16 ? 16 : -1};
const int kPP[kNumModels][2] = {{1, 2}, {2, 3}, {1, 2}, {1, 2}, {2, 3}, {2, 3}, {2, 3}, {2, 3}, {1, 2}, {1, 2}, {2, 3}, {2, 3}};

struct BaOptionsC {
  int32_t max_num_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius, min_relative_decrease;
  double min_lm_diagonal, max_lm_diagonal;
  int32_t jacobi_scaling, max_num_consecutive_invalid_steps;
  double pcg_relative_tolerance;
  int32_t pcg_max_iterations, order, verbose, line_search;  // line_search: GP only (BA has no bounds)
  double thres_loss_function;
  int32_t optimize_rotations, optimize_translation, optimize_intrinsics, optimize_principal_point, optimize_points;
  int32_t min_num_view_per_track;
  int32_t optimize_rig_poses;
};

struct BaReport {
  int32_t iterations, successful_steps, termination, usable;
  i64 linear_iterations;
  double initial_cost, final_cost, max_linear_residual, seconds_total, seconds_linear;
  int32_t threads, pad;
};

namespace {

inline void quat_to_rot(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z);
  R[1] = 2 * (x * y - w * z);
  R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);
  R[4] = 1 - 2 * (x * x + z * z);
  R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);
  R[7] = 2 * (y * z + w * x);
  R[8] = 1 - 2 * (x * x + y * y);
}

// ImgFromCam with analytic Jacobians: uv[2], Jx[2][3] = d(uv)/d(x_c), Jp[2][8] = d(uv)/d(params).
inline bool project(int model, const double* p, const double* xc, double* uv, double* Jx, double* Jp) {
  for (int i = 0; i < MP2; ++i) Jp[i] = 0.0;
  const double z = xc[2];
  if (!(z > 2.220446049250313e-16)) return false;
  const double u = xc[0] / z, v = xc[1] / z;
  const double r2 = u * u + v * v;
  double J00, J01, J10, J11;  // d(pixel)/d(u, v)
  switch (model) {
    case SIMPLE_PINHOLE: {
      const double f = p[0];
      uv[0] = f * u + p[1];
      uv[1] = f * v + p[2];
      J00 = f; J01 = 0; J10 = 0; J11 = f;
      Jp[0] = u; Jp[MAXP] = v;
      Jp[1] = 1; Jp[MAXP + 2] = 1;
      break;
    }
    case PINHOLE: {
      uv[0] = p[0] * u + p[2];
      uv[1] = p[1] * v + p[3];
      J00 = p[0]; J01 = 0; J10 = 0; J11 = p[1];
      Jp[0] = u; Jp[MAXP + 1] = v;
      Jp[2] = 1; Jp[MAXP + 3] = 1;
      break;
    }
    case SIMPLE_RADIAL:
    case RADIAL: {
      const double f = p[0], k1 = p[3], k2 = model == RADIAL ? p[4] : 0.0;
      const double rad = k1 * r2 + k2 * r2 * r2;
      const double drad = k1 + 2 * k2 * r2;
      const double ud = u * (1 + rad), vd = v * (1 + rad);
      uv[0] = f * ud + p[1];
      uv[1] = f * vd + p[2];
      J00 = f * (1 + rad + 2 * u * u * drad);
      J01 = f * (2 * u * v * drad);
      J10 = J01;
      J11 = f * (1 + rad + 2 * v * v * drad);
      Jp[0] = ud; Jp[MAXP] = vd;
      Jp[1] = 1; Jp[MAXP + 2] = 1;
      Jp[3] = f * u * r2; Jp[MAXP + 3] = f * v * r2;
      if (model == RADIAL) {
        Jp[4] = f * u * r2 * r2;
        Jp[MAXP + 4] = f * v * r2 * r2;
      }
      break;
    }
    case OPENCV: {
      const double fx = p[0], fy = p[1], k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7];
      const double rad = k1 * r2 + k2 * r2 * r2;
      const double drad = k1 + 2 * k2 * r2;
      const double du = u * rad + 2 * p1 * u * v + p2 * (r2 + 2 * u * u);
      const double dv = v * rad + 2 * p2 * u * v + p1 * (r2 + 2 * v * v);
      uv[0] = fx * (u + du) + p[2];
      uv[1] = fy * (v + dv) + p[3];
      const double ddu_du = rad + 2 * u * u * drad + 2 * p1 * v + 6 * p2 * u;
      const double ddu_dv = 2 * u * v * drad + 2 * p1 * u + 2 * p2 * v;
      const double ddv_du = 2 * u * v * drad + 2 * p2 * v + 2 * p1 * u;
      const double ddv_dv = rad + 2 * v * v * drad + 2 * p2 * u + 6 * p1 * v;
      J00 = fx * (1 + ddu_du); J01 = fx * ddu_dv; J10 = fy * ddv_du; J11 = fy * (1 + ddv_dv);
      Jp[0] = u + du; Jp[MAXP + 1] = v + dv;
      Jp[2] = 1; Jp[MAXP + 3] = 1;
      Jp[4] = fx * u * r2; Jp[MAXP + 4] = fy * v * r2;
      Jp[5] = fx * u * r2 * r2; Jp[MAXP + 5] = fy * v * r2 * r2;
      Jp[6] = fx * 2 * u * v; Jp[MAXP + 6] = fy * (r2 + 2 * v * v);
      Jp[7] = fx * (r2 + 2 * u * u); Jp[MAXP + 7] = fy * 2 * u * v;
      break;
    }
    case OPENCV_FISHEYE:
    case SIMPLE_RADIAL_FISHEYE:
    case RADIAL_FISHEYE: {
      // equidistant fisheye: theta = atan(r); pixel = f (u, v) theta_d / r + c, theta_d = theta (1 + sum_j k_j theta^(2j+2))
      const bool full = model == OPENCV_FISHEYE;
      const double fx = p[0], fy = full ? p[1] : p[0];
      const int ic = full ? 2 : 1, ik0 = full ? 4 : 3, nk = full ? 4 : (model == RADIAL_FISHEYE ? 2 : 1);
      const double r = std::sqrt(r2);
      const bool big = r > 2.220446049250313e-16;
      const double th = big ? std::atan(r) : r, th2 = th * th;
      double poly = 1.0, dpoly = 1.0, tp = 1.0, tpow[4] = {0, 0, 0, 0};
      for (int j = 0; j < nk; ++j) {
        tp *= th2;
        tpow[j] = tp;
        poly += p[ik0 + j] * tp;
        dpoly += (2.0 * j + 3.0) * p[ik0 + j] * tp;
      }
      const double m = big ? th * poly / r : poly;
      const double dm_r = big ? (dpoly / (1.0 + r2) - m) / r2 : 0.0;
      uv[0] = fx * u * m + p[ic];
      uv[1] = fy * v * m + p[ic + 1];
      J00 = fx * (m + u * u * dm_r); J01 = fx * u * v * dm_r; J10 = fy * u * v * dm_r; J11 = fy * (m + v * v * dm_r);
      if (full) {
        Jp[0] = u * m; Jp[MAXP + 1] = v * m;
      } else {
        Jp[0] = u * m; Jp[MAXP] = v * m;
      }
      Jp[ic] = 1; Jp[MAXP + ic + 1] = 1;
      const double sf = big ? th / r : 1.0;
      for (int j = 0; j < nk; ++j) {
        Jp[ik0 + j] = fx * u * sf * tpow[j];
        Jp[MAXP + ik0 + j] = fy * v * sf * tpow[j];
      }
      break;
    }
    case FOV: {
      const double fx = p[0], fy = p[1], om = p[4], om2 = om * om;
      double fac, dfac_r2, dfac_om;
      if (om2 < 1e-4) {  // COLMAP's series branches (FOVCameraModel::Distortion)
        fac = om2 * r2 / 3.0 - om2 / 12.0 + 1.0;
        dfac_r2 = om2 / 3.0;
        dfac_om = 2.0 * om * (r2 / 3.0 - 1.0 / 12.0);
      } else if (r2 < 1e-4) {
        const double t = std::tan(0.5 * om), t2 = t * t;
        fac = (-2.0 * t * (4.0 * r2 * t2 - 3.0)) / (3.0 * om);
        dfac_r2 = -8.0 * t * t2 / (3.0 * om);
        dfac_om = (-2.0 * 0.5 * (1.0 + t2) * (12.0 * r2 * t2 - 3.0)) / (3.0 * om) - fac / om;
      } else {
        const double r = std::sqrt(r2), t = std::tan(0.5 * om), a = 2.0 * r * t, num = std::atan(a), da = 1.0 / (1.0 + a * a);
        fac = num / (r * om);
        dfac_r2 = (da * 2.0 * t * r - num) / (r2 * om) / (2.0 * r);
        dfac_om = da * 2.0 * r * 0.5 * (1.0 + t * t) / (r * om) - fac / om;
      }
      uv[0] = fx * u * fac + p[2];
      uv[1] = fy * v * fac + p[3];
      J00 = fx * (fac + 2 * u * u * dfac_r2); J01 = fx * 2 * u * v * dfac_r2; J10 = fy * 2 * u * v * dfac_r2;
      J11 = fy * (fac + 2 * v * v * dfac_r2);
      Jp[0] = u * fac; Jp[MAXP + 1] = v * fac;
      Jp[2] = 1; Jp[MAXP + 3] = 1;
      Jp[4] = fx * u * dfac_om; Jp[MAXP + 4] = fy * v * dfac_om;
      break;
    }
#if ORC_BA_MAXP >= 16
    case FULL_OPENCV:
    case THIN_PRISM_FISHEYE:
    case RAD_TAN_THIN_PRISM_FISHEYE: {
      // (a, b) -> (ad, bd) distortions of (u, v) itself (FULL_OPENCV) or of the equidistant coordinates (a, b) = (u, v) theta / r
      // (the thin-prism fisheye models); D = d(ad, bd) / d(a, b), E = d(a, b) / d(u, v).  Same formulas as oracle/ba.py.
      const double fx = p[0], fy = p[1];
      double a = u, b = v, e0 = 1.0, e1 = 0.0, e2 = 0.0, e3 = 1.0;
      if (model != FULL_OPENCV) {
        const double r = std::sqrt(r2);
        if (r > 2.220446049250313e-16) {
          const double s = std::atan(r) / r, sp = (1.0 / (1.0 + r2) - s) / r2;
          a = s * u; b = s * v;
          e0 = s + u * u * sp; e1 = u * v * sp; e2 = e1; e3 = s + v * v * sp;
        }
      }
      const double a2 = a * a, b2 = b * b, ab = a * b, q2 = a2 + b2;
      double ad, bd, d0, d1, d2, d3;
      double dpa[MAXP], dpb[MAXP];  // d(ad) / d(param), d(bd) / d(param)
      for (int j = 0; j < MAXP; ++j) dpa[j] = dpb[j] = 0.0;
      if (model == FULL_OPENCV) {
        const double k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7], k3 = p[8], k4 = p[9], k5 = p[10], k6 = p[11];
        const double q4 = q2 * q2, q6 = q4 * q2;
        const double num = 1.0 + k1 * q2 + k2 * q4 + k3 * q6, den = 1.0 + k4 * q2 + k5 * q4 + k6 * q6, rad = num / den;
        const double drad = ((k1 + 2.0 * k2 * q2 + 3.0 * k3 * q4) - rad * (k4 + 2.0 * k5 * q2 + 3.0 * k6 * q4)) / den;
        ad = a * rad + 2.0 * p1 * ab + p2 * (q2 + 2.0 * a2);
        bd = b * rad + 2.0 * p2 * ab + p1 * (q2 + 2.0 * b2);
        d0 = rad + 2.0 * a2 * drad + 2.0 * p1 * b + 6.0 * p2 * a;
        d1 = 2.0 * ab * drad + 2.0 * p1 * a + 2.0 * p2 * b;
        d2 = 2.0 * ab * drad + 2.0 * p2 * b + 2.0 * p1 * a;
        d3 = rad + 2.0 * b2 * drad + 2.0 * p2 * a + 6.0 * p1 * b;
        const double qq[3] = {q2, q4, q6};
        const int inum[3] = {4, 5, 8}, iden[3] = {9, 10, 11};
        for (int j = 0; j < 3; ++j) {
          dpa[inum[j]] = a * qq[j] / den; dpb[inum[j]] = b * qq[j] / den;
          dpa[iden[j]] = -a * rad * qq[j] / den; dpb[iden[j]] = -b * rad * qq[j] / den;
        }
        dpa[6] = 2.0 * ab; dpb[6] = q2 + 2.0 * b2;
        dpa[7] = q2 + 2.0 * a2; dpb[7] = 2.0 * ab;
      } else if (model == THIN_PRISM_FISHEYE) {
        const double k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7], k3 = p[8], k4 = p[9], sx1 = p[10], sy1 = p[11];
        const double q4 = q2 * q2, q6 = q4 * q2, q8 = q4 * q4;
        const double rad = k1 * q2 + k2 * q4 + k3 * q6 + k4 * q8, drad = k1 + 2.0 * k2 * q2 + 3.0 * k3 * q4 + 4.0 * k4 * q6;
        ad = a + a * rad + 2.0 * p1 * ab + p2 * (q2 + 2.0 * a2) + sx1 * q2;
        bd = b + b * rad + 2.0 * p2 * ab + p1 * (q2 + 2.0 * b2) + sy1 * q2;
        d0 = 1.0 + rad + 2.0 * a2 * drad + 2.0 * p1 * b + 6.0 * p2 * a + 2.0 * sx1 * a;
        d1 = 2.0 * ab * drad + 2.0 * p1 * a + 2.0 * p2 * b + 2.0 * sx1 * b;
        d2 = 2.0 * ab * drad + 2.0 * p2 * b + 2.0 * p1 * a + 2.0 * sy1 * a;
        d3 = 1.0 + rad + 2.0 * b2 * drad + 2.0 * p2 * a + 6.0 * p1 * b + 2.0 * sy1 * b;
        const double qq[4] = {q2, q4, q6, q8};
        const int ik[4] = {4, 5, 8, 9};
        for (int j = 0; j < 4; ++j) { dpa[ik[j]] = a * qq[j]; dpb[ik[j]] = b * qq[j]; }
        dpa[6] = 2.0 * ab; dpb[6] = q2 + 2.0 * b2;
        dpa[7] = q2 + 2.0 * a2; dpb[7] = 2.0 * ab;
        dpa[10] = q2;
        dpb[11] = q2;
      } else {
        const double p0 = p[10], p1 = p[11], s0 = p[12], s1 = p[13], s2 = p[14], s3 = p[15];
        double qp[6];
        qp[0] = q2;
        for (int i = 1; i < 6; ++i) qp[i] = qp[i - 1] * q2;
        double Rr = 1.0, dR = p[4];
        for (int i = 0; i < 6; ++i) Rr += p[4 + i] * qp[i];
        for (int i = 1; i < 6; ++i) dR += (i + 1.0) * p[4 + i] * qp[i - 1];
        const double uh = Rr * a, vh = Rr * b, uh2 = uh * uh, vh2 = vh * vh, uhvh = uh * vh, h2 = uh2 + vh2, h4 = h2 * h2;
        ad = uh + p0 * (2.0 * uh2 + h2) + 2.0 * p1 * uhvh + s0 * h2 + s1 * h4;
        bd = vh + p1 * (2.0 * vh2 + h2) + 2.0 * p0 * uhvh + s2 * h2 + s3 * h4;
        const double sx = s0 + 2.0 * s1 * h2, sy = s2 + 2.0 * s3 * h2;
        const double T0 = 1.0 + 6.0 * p0 * uh + 2.0 * p1 * vh + 2.0 * uh * sx, T1 = 2.0 * p0 * vh + 2.0 * p1 * uh + 2.0 * vh * sx;
        const double T2 = 2.0 * p1 * uh + 2.0 * p0 * vh + 2.0 * uh * sy, T3 = 1.0 + 6.0 * p1 * vh + 2.0 * p0 * uh + 2.0 * vh * sy;
        const double r0 = Rr + 2.0 * a2 * dR, r1 = 2.0 * ab * dR, r3 = Rr + 2.0 * b2 * dR;
        d0 = T0 * r0 + T1 * r1; d1 = T0 * r1 + T1 * r3; d2 = T2 * r0 + T3 * r1; d3 = T2 * r1 + T3 * r3;
        const double ta = T0 * a + T1 * b, tb = T2 * a + T3 * b;
        for (int i = 0; i < 6; ++i) { dpa[4 + i] = ta * qp[i]; dpb[4 + i] = tb * qp[i]; }
        dpa[10] = 2.0 * uh2 + h2; dpb[10] = 2.0 * uhvh;
        dpa[11] = 2.0 * uhvh; dpb[11] = 2.0 * vh2 + h2;
        dpa[12] = h2; dpa[13] = h4;
        dpb[14] = h2; dpb[15] = h4;
      }
      uv[0] = fx * ad + p[2];
      uv[1] = fy * bd + p[3];
      J00 = fx * (d0 * e0 + d1 * e2); J01 = fx * (d0 * e1 + d1 * e3);
      J10 = fy * (d2 * e0 + d3 * e2); J11 = fy * (d2 * e1 + d3 * e3);
      for (int j = 4; j < MAXP; ++j) { Jp[j] = fx * dpa[j]; Jp[MAXP + j] = fy * dpb[j]; }
      Jp[0] = ad; Jp[MAXP + 1] = bd;
      Jp[2] = 1; Jp[MAXP + 3] = 1;
      break;
    }
#endif
    default:
      return false;
  }
  // d(u,v)/d(x_c) = [[1/z, 0, -u/z], [0, 1/z, -v/z]]
  const double iz = 1.0 / z;
  Jx[0] = J00 * iz; Jx[1] = J01 * iz; Jx[2] = -(J00 * u + J01 * v) * iz;
  Jx[3] = J10 * iz; Jx[4] = J11 * iz; Jx[5] = -(J10 * u + J11 * v) * iz;
  return true;
}

struct Ba : LmProblem {
  i64 N, P, M, K;
  std::vector<int32_t> cam, pt, ik;  // per observation: camera, point, intrinsics block
  std::vector<i64> poff;
  std::vector<double> xy;
  std::vector<double> sens;           // [M][12] known rigs: cam_from_rig (R row-major 9 | t 3) of the observation's image, else empty
  // optimize_rig_poses (RigReprojErrorCostFunctor, ba.cc:161-179): S cam_from_rig blocks, stored as pose blocks N .. N+S-1
  // of q / t and of the reduced vector; sblk[k] = block of observation k's sensor or -1
  i64 S = 0;
  std::vector<int32_t> sblk;
  std::vector<int32_t> sobs, sown;    // observations that have a block, and their block (owner lists of the blocks)
  OwnerLists bysens;
  std::vector<double> Js;             // [M][2][6] Jacobian w.r.t. the sensor block
  std::vector<double> Msblk;          // block-Jacobi: 6x6 per sensor block
  std::vector<int32_t> cam_intr, model;
  OwnerLists bycam, byintr;
  Huber loss;
  std::vector<uint8_t> rot_free, trn_free;
  std::vector<uint8_t> free_par;      // [K][8]
  std::vector<int32_t> icol;          // [K][8] -> compact column or -1
  i64 nfree = 0, nred = 0;            // reduced system = 6 (N + S) + nfree
  double mpt;                         // optimize_points
  double lm_lo, lm_hi;
  bool rev;
  double pcg_tol;
  int pcg_max;
  // joint / separate preconditioner blocks
  std::vector<int32_t> intr_owner;    // [K]: the single camera using this block, or -1 when shared / unused
  // state
  std::vector<double> q, t, X, intr, q2, t2, X2, intr2;
  // linearisation (robustified, tangent space)
  std::vector<double> rt;             // [M][2]
  std::vector<double> Jc;             // [M][2][6]
  std::vector<double> Jp;             // [M][2][3]
  std::vector<double> Ji;             // [M][2][8]
  std::vector<double> gred, gpt;      // J^T r~ : [nred], [3P]
  std::vector<double> hred, hpt;      // column squared norms
  std::vector<double> jred, jpt;      // Jacobi scales
  // per step
  std::vector<double> dred, Hinv /*[P][9]*/, tp, ak /*[M][2]*/;
  std::vector<double> dy_prev;  // ORC_WARM_START experiment: the reduced step of the last solve ...
  bool same_lin = false;        // ... and whether it belongs to the current linearisation
  bool defl_on = true;          // ORC_DEFLATE experiment: deflate the next reduced solve
  std::vector<double> Mblk;           // block-Jacobi: per camera 14x14 (joint) ...
  std::vector<double> Miblk;          // ... per shared intrinsics block 8x8

  inline i64 ccol(i64 n) const { return 6 * n; }
  inline double damp(double h, double j, double radius) const {
    const double j2 = j * j;
    return std::min(std::max(j2 * h, lm_lo), lm_hi) / (radius * j2);
  }

  // Jcam / arig (optional): d(uv)/d(x_c) before the chain through R_s, and R_s x_rig — what the sensor block's rotation acts on
  void residual(i64 k, const std::vector<double>& qq, const std::vector<double>& tt, const std::vector<double>& XX,
                const std::vector<double>& in, double* r, double* R, double* RX, double* Jx, double* Jpar, bool* valid,
                double* Jcam = nullptr, double* arig = nullptr) const {
    const i64 n = cam[k], p = pt[k], b = ik[k];
    quat_to_rot(&qq[4 * n], R);
    const double* x = &XX[3 * p];
    for (int i = 0; i < 3; ++i) RX[i] = R[3 * i] * x[0] + R[3 * i + 1] * x[1] + R[3 * i + 2] * x[2];
    double xc[3] = {RX[0] + tt[3 * n], RX[1] + tt[3 * n + 1], RX[2] + tt[3 * n + 2]};
    const double* S = sens.empty() ? nullptr : &sens[12 * k];
    double Sv[12];
    if (!sblk.empty() && sblk[k] >= 0) {  // the cam_from_rig is the sensor block's current value
      const i64 sb = N + sblk[k];
      quat_to_rot(&qq[4 * sb], Sv);
      for (int i = 0; i < 3; ++i) Sv[9 + i] = tt[3 * sb + i];
      S = Sv;
    }
    if (S) {  // RigReprojError*CostFunctor (bundle_adjustment.cc:147-179): x_c = cam_from_rig * (rig_from_world * X)
      const double xr[3] = {xc[0], xc[1], xc[2]};
      for (int i = 0; i < 3; ++i) {
        const double a = S[3 * i] * xr[0] + S[3 * i + 1] * xr[1] + S[3 * i + 2] * xr[2];
        if (arig) arig[i] = a;
        xc[i] = a + S[9 + i];
      }
    }
    double uv[2];
    *valid = project(model[b], &in[MAXP * b], xc, uv, Jx, Jpar);
    if (Jcam)
      for (int i = 0; i < 6; ++i) Jcam[i] = *valid ? Jx[i] : 0.0;
    if (*valid && S) {  // d(uv)/d(x_rig) = d(uv)/d(x_c) R_s: everything downstream differentiates through the rig-frame point
      double J2[6];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) J2[3 * i + j] = Jx[3 * i] * S[j] + Jx[3 * i + 1] * S[3 + j] + Jx[3 * i + 2] * S[6 + j];
      for (int i = 0; i < 6; ++i) Jx[i] = J2[i];
    }
    if (*valid) {
      r[0] = uv[0] - xy[2 * k];
      r[1] = uv[1] - xy[2 * k + 1];
    } else {
      r[0] = r[1] = 0.0;
      for (int i = 0; i < 6; ++i) Jx[i] = 0.0;
    }
  }

  double cost_at(const std::vector<double>& qq, const std::vector<double>& tt, const std::vector<double>& XX,
                 const std::vector<double>& in) const {
    return 0.5 * chunked_sum(M, [&](i64 k) {
      double r[2], R[9], RX[3], Jx[6], Jpar[MP2];
      bool valid;
      residual(k, qq, tt, XX, in, r, R, RX, Jx, Jpar, &valid);
      double r0, r1;
      loss.eval(r[0] * r[0] + r[1] * r[1], r0, r1);
      return r0;
    });
  }

  double linearize(double* gmax_out) override {
    same_lin = false;
    rt.resize(2 * M);
    Jc.resize(12 * M);
    Jp.resize(6 * M);
    Ji.resize((size_t)MP2 * M);
    if (S > 0) Js.assign(12 * M, 0.0);
    std::vector<double> rho(M);
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) {
      double r[2], R[9], RX[3], Jx[6], Jpar[MP2], Jcam[6], arig[3] = {0, 0, 0};
      bool valid;
      residual(k, q, t, X, intr, r, R, RX, Jx, Jpar, &valid, Jcam, arig);
      double r0, r1;
      loss.eval(r[0] * r[0] + r[1] * r[1], r0, r1);
      rho[k] = r0;
      const double sw = valid ? std::sqrt(r1) : 0.0;
      rt[2 * k] = sw * r[0];
      rt[2 * k + 1] = sw * r[1];
      const i64 n = cam[k], b = ik[k];
      const double fr = rot_free[n] ? 1.0 : 0.0, ft = trn_free[n] ? 1.0 : 0.0;
      // d x_c / d delta_rot = -2 [R X]_x  (EigenQuaternionManifold: rotation by 2|delta| on the left)
      const double a0 = RX[0], a1 = RX[1], a2 = RX[2];
      for (int i = 0; i < 2; ++i) {
        const double j0 = sw * Jx[3 * i], j1 = sw * Jx[3 * i + 1], j2 = sw * Jx[3 * i + 2];
        // (Jx skew)[i][:] with skew = [[0,-a2,a1],[a2,0,-a0],[-a1,a0,0]]
        Jc[12 * k + 6 * i + 0] = fr * -2.0 * (j1 * a2 - j2 * a1);
        Jc[12 * k + 6 * i + 1] = fr * -2.0 * (-j0 * a2 + j2 * a0);
        Jc[12 * k + 6 * i + 2] = fr * -2.0 * (j0 * a1 - j1 * a0);
        Jc[12 * k + 6 * i + 3] = ft * j0;
        Jc[12 * k + 6 * i + 4] = ft * j1;
        Jc[12 * k + 6 * i + 5] = ft * j2;
        for (int j = 0; j < 3; ++j) Jp[6 * k + 3 * i + j] = mpt * (j0 * R[j] + j1 * R[3 + j] + j2 * R[6 + j]);
        for (int j = 0; j < MAXP; ++j) Ji[MP2 * k + MAXP * i + j] = free_par[MAXP * b + j] ? sw * Jpar[MAXP * i + j] : 0.0;
        if (S > 0 && sblk[k] >= 0) {  // x_c = Exp(2 d_rot) (R_s x_rig) + t_s + d_trn: always free (ba.cc:296-309)
          const double c0 = sw * Jcam[3 * i], c1 = sw * Jcam[3 * i + 1], c2 = sw * Jcam[3 * i + 2];
          Js[12 * k + 6 * i + 0] = -2.0 * (c1 * arig[2] - c2 * arig[1]);
          Js[12 * k + 6 * i + 1] = -2.0 * (-c0 * arig[2] + c2 * arig[0]);
          Js[12 * k + 6 * i + 2] = -2.0 * (c0 * arig[1] - c1 * arig[0]);
          Js[12 * k + 6 * i + 3] = c0;
          Js[12 * k + 6 * i + 4] = c1;
          Js[12 * k + 6 * i + 5] = c2;
        }
      }
    }
    const double cost = 0.5 * chunked_sum(M, [&](i64 k) { return rho[k]; });
    gred.assign(nred, 0.0);
    hred.assign(nred, 0.0);
    gpt.assign(3 * P, 0.0);
    hpt.assign(3 * P, 0.0);
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 p = 0; p < P; ++p) {
      double g[3] = {0, 0, 0}, h[3] = {0, 0, 0};
      auto body = [&](i64 k) {
        for (int j = 0; j < 3; ++j) {
          g[j] += Jp[6 * k + j] * rt[2 * k] + Jp[6 * k + 3 + j] * rt[2 * k + 1];
          h[j] += Jp[6 * k + j] * Jp[6 * k + j] + Jp[6 * k + 3 + j] * Jp[6 * k + 3 + j];
        }
      };
      if (!rev)
        for (i64 k = poff[p]; k < poff[p + 1]; ++k) body(k);
      else
        for (i64 k = poff[p + 1] - 1; k >= poff[p]; --k) body(k);
      for (int j = 0; j < 3; ++j) {
        gpt[3 * p + j] = g[j];
        hpt[3 * p + j] = h[j];
      }
    }
    std::vector<double> acc(12 * N);
    bycam.reduce<12>(acc.data(), rev, [&](i64 k, double* a) {
      for (int j = 0; j < 6; ++j) {
        a[j] += Jc[12 * k + j] * rt[2 * k] + Jc[12 * k + 6 + j] * rt[2 * k + 1];
        a[6 + j] += Jc[12 * k + j] * Jc[12 * k + j] + Jc[12 * k + 6 + j] * Jc[12 * k + 6 + j];
      }
    });
    for (i64 n = 0; n < N; ++n)
      for (int j = 0; j < 6; ++j) {
        gred[6 * n + j] = acc[12 * n + j];
        hred[6 * n + j] = acc[12 * n + 6 + j];
      }
    if (S > 0) {
      std::vector<double> accs(12 * S);
      bysens.reduce<12>(accs.data(), rev, [&](i64 e, double* a) {
        const i64 k = sobs[e];
        for (int j = 0; j < 6; ++j) {
          a[j] += Js[12 * k + j] * rt[2 * k] + Js[12 * k + 6 + j] * rt[2 * k + 1];
          a[6 + j] += Js[12 * k + j] * Js[12 * k + j] + Js[12 * k + 6 + j] * Js[12 * k + 6 + j];
        }
      });
      for (i64 sb = 0; sb < S; ++sb)
        for (int j = 0; j < 6; ++j) {
          gred[6 * (N + sb) + j] = accs[12 * sb + j];
          hred[6 * (N + sb) + j] = accs[12 * sb + 6 + j];
        }
    }
    std::vector<double> acci((size_t)MP2 * K);
    byintr.reduce<MP2>(acci.data(), rev, [&](i64 k, double* a) {
      for (int j = 0; j < MAXP; ++j) {
        a[j] += Ji[MP2 * k + j] * rt[2 * k] + Ji[MP2 * k + MAXP + j] * rt[2 * k + 1];
        a[MAXP + j] += Ji[MP2 * k + j] * Ji[MP2 * k + j] + Ji[MP2 * k + MAXP + j] * Ji[MP2 * k + MAXP + j];
      }
    });
    for (i64 b = 0; b < K; ++b)
      for (int j = 0; j < MAXP; ++j)
        if (icol[MAXP * b + j] >= 0) {
          gred[icol[MAXP * b + j]] = acci[MP2 * b + j];
          hred[icol[MAXP * b + j]] = acci[MP2 * b + MAXP + j];
        }
    double gmax = chunked_max(nred, [&](i64 i) { return std::fabs(gred[i]); });
    gmax = std::max(gmax, chunked_max(3 * P, [&](i64 i) { return std::fabs(gpt[i]); }));
    *gmax_out = gmax;
    return cost;
  }

  void set_jacobi_scaling(bool enabled) override {
    jred.assign(nred, 1.0);
    jpt.assign(3 * P, 1.0);
    if (enabled) {
      for (i64 i = 0; i < nred; ++i) jred[i] = 1.0 / (1.0 + std::sqrt(hred[i]));
      for (i64 i = 0; i < 3 * P; ++i) jpt[i] = 1.0 / (1.0 + std::sqrt(hpt[i]));
    }
  }

  // a_k = Jc_k z_c + Ji_k z_i
  inline void cam_side(i64 k, const std::vector<double>& z, double* a) const {
    const i64 n = cam[k], b = ik[k];
    a[0] = a[1] = 0.0;
    for (int j = 0; j < 6; ++j) {
      a[0] += Jc[12 * k + j] * z[6 * n + j];
      a[1] += Jc[12 * k + 6 + j] * z[6 * n + j];
    }
    if (S > 0 && sblk[k] >= 0) {
      const i64 o = 6 * (N + sblk[k]);
      for (int j = 0; j < 6; ++j) {
        a[0] += Js[12 * k + j] * z[o + j];
        a[1] += Js[12 * k + 6 + j] * z[o + j];
      }
    }
    for (int j = 0; j < MAXP; ++j) {
      const int32_t col = icol[MAXP * b + j];
      if (col >= 0) {
        a[0] += Ji[MP2 * k + j] * z[col];
        a[1] += Ji[MP2 * k + MAXP + j] * z[col];
      }
    }
  }

  // tp = Hpp^-1 sum_k Jp_k^T a_k(z); ak <- a_k - Jp_k tp
  void point_pass(const std::vector<double>& z) {
    ak.resize(2 * M);
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 p = 0; p < P; ++p) {
      double b[3] = {0, 0, 0};
      auto body = [&](i64 k) {
        double a[2];
        cam_side(k, z, a);
        ak[2 * k] = a[0];
        ak[2 * k + 1] = a[1];
        for (int j = 0; j < 3; ++j) b[j] += Jp[6 * k + j] * a[0] + Jp[6 * k + 3 + j] * a[1];
      };
      if (!rev)
        for (i64 k = poff[p]; k < poff[p + 1]; ++k) body(k);
      else
        for (i64 k = poff[p + 1] - 1; k >= poff[p]; --k) body(k);
      const double* H = &Hinv[9 * p];
      double tv[3];
      for (int i = 0; i < 3; ++i) tv[i] = H[3 * i] * b[0] + H[3 * i + 1] * b[1] + H[3 * i + 2] * b[2];
      for (int i = 0; i < 3; ++i) tp[3 * p + i] = tv[i];
      for (i64 k = poff[p]; k < poff[p + 1]; ++k) {
        ak[2 * k] -= Jp[6 * k] * tv[0] + Jp[6 * k + 1] * tv[1] + Jp[6 * k + 2] * tv[2];
        ak[2 * k + 1] -= Jp[6 * k + 3] * tv[0] + Jp[6 * k + 4] * tv[1] + Jp[6 * k + 5] * tv[2];
      }
    }
  }

  // out = [Jc Ji]^T e  over all observations, e [M][2]
  void cam_transpose(const std::vector<double>& e, std::vector<double>& out) {
    std::vector<double> acc(6 * N);
    bycam.reduce<6>(acc.data(), rev, [&](i64 k, double* a) {
      for (int j = 0; j < 6; ++j) a[j] += Jc[12 * k + j] * e[2 * k] + Jc[12 * k + 6 + j] * e[2 * k + 1];
    });
    std::copy(acc.begin(), acc.end(), out.begin());
    if (S > 0) {
      std::vector<double> accs(6 * S);
      bysens.reduce<6>(accs.data(), rev, [&](i64 ee, double* a) {
        const i64 k = sobs[ee];
        for (int j = 0; j < 6; ++j) a[j] += Js[12 * k + j] * e[2 * k] + Js[12 * k + 6 + j] * e[2 * k + 1];
      });
      std::copy(accs.begin(), accs.end(), out.begin() + 6 * N);
    }
    if (nfree > 0) {
      std::vector<double> acci((size_t)MAXP * K);
      byintr.reduce<MAXP>(acci.data(), rev, [&](i64 k, double* a) {
        for (int j = 0; j < MAXP; ++j) a[j] += Ji[MP2 * k + j] * e[2 * k] + Ji[MP2 * k + MAXP + j] * e[2 * k + 1];
      });
      for (i64 b = 0; b < K; ++b)
        for (int j = 0; j < MAXP; ++j)
          if (icol[MAXP * b + j] >= 0) out[icol[MAXP * b + j]] = acci[MAXP * b + j];
    }
  }

  void apply(const std::vector<double>& z, std::vector<double>& out) {
    point_pass(z);
    cam_transpose(ak, out);
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < nred; ++i) out[i] += dred[i] * z[i];
  }

  bool step(double radius, double* model_change, double* cand_cost, double* step_norm, double* x_norm, i64* lin,
            double* relres) override {
    dred.resize(nred);
    for (i64 i = 0; i < nred; ++i) dred[i] = damp(hred[i], jred[i], radius);
    Hinv.resize(9 * P);
    tp.assign(3 * P, 0.0);
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 p = 0; p < P; ++p) {
      double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      auto body = [&](i64 k) {
        for (int i = 0; i < 3; ++i)
          for (int j = i; j < 3; ++j) H[3 * i + j] += Jp[6 * k + i] * Jp[6 * k + j] + Jp[6 * k + 3 + i] * Jp[6 * k + 3 + j];
      };
      if (!rev)
        for (i64 k = poff[p]; k < poff[p + 1]; ++k) body(k);
      else
        for (i64 k = poff[p + 1] - 1; k >= poff[p]; --k) body(k);
      H[3] = H[1];
      H[6] = H[2];
      H[7] = H[5];
      for (int j = 0; j < 3; ++j) H[4 * j] += damp(hpt[3 * p + j], jpt[3 * p + j], radius);
      if (!spd_inverse(H, 3)) {
#pragma omp atomic write
        ok = false;
      }
      std::memcpy(&Hinv[9 * p], H, sizeof H);
    }
    if (!ok) return false;
    // rhs = -[Jc Ji]^T (r~ - Jp u), u = Hpp^-1 gpt
    std::vector<double> u(3 * P), e(2 * M), rhs(nred);
#pragma omp parallel for schedule(static)
    for (i64 p = 0; p < P; ++p) {
      const double* H = &Hinv[9 * p];
      for (int i = 0; i < 3; ++i) u[3 * p + i] = H[3 * i] * gpt[3 * p] + H[3 * i + 1] * gpt[3 * p + 1] + H[3 * i + 2] * gpt[3 * p + 2];
      for (i64 k = poff[p]; k < poff[p + 1]; ++k) {
        e[2 * k] = rt[2 * k] - (Jp[6 * k] * u[3 * p] + Jp[6 * k + 1] * u[3 * p + 1] + Jp[6 * k + 2] * u[3 * p + 2]);
        e[2 * k + 1] = rt[2 * k + 1] - (Jp[6 * k + 3] * u[3 * p] + Jp[6 * k + 4] * u[3 * p + 1] + Jp[6 * k + 5] * u[3 * p + 2]);
      }
    }
    cam_transpose(e, rhs);
    for (i64 i = 0; i < nred; ++i) rhs[i] = -rhs[i];
    if (!build_preconditioner()) return false;
    std::vector<double> dy(nred, 0.0);
    *relres = 0.0;
    // experiment (ORC_WARM_START=1, tools/exp_lm_warm_start.py): after a rejected step the linearisation is the same and
    // only the damping moved, so the rejected step is a starting guess for the next solve.  Off: the oracle's solves
    // start from zero.
    static const bool warm = std::getenv("ORC_WARM_START") != nullptr;
    const std::vector<double>* guess = warm && same_lin && (i64)dy_prev.size() == nred ? &dy_prev : nullptr;
    // experiment (ORC_DEFLATE=1, tools/exp_deflation.py): deflate the seven similarity modes of the scene — world
    // translation (dt_n = -R_n a), world rotation (left tangent d_n = -R_n w / 2: the quaternion manifold turns by
    // 2 |d|), scale (dt_n = t_n) — from the PCG.  They are the near-null space of the reduced system: one constant frame
    // anchors six of them with a stiffness of O(1/N), the scale only through the LM damping.
    // ORC_DEFLATE=6 leaves the scale out: it is an EXACT gauge of the cost (held by the LM damping alone), so "solving"
    // its component divides rounding noise by the damping and sends the iterate along the gauge.
    const int deflate = std::getenv("ORC_DEFLATE") ? std::atoi(std::getenv("ORC_DEFLATE")) : 0;  // (read per solve)
    std::vector<std::vector<double>> W;
    if (deflate && S == 0 && defl_on) {  // like ba.hip: short (strongly damped) solves run undeflated
      W.assign(deflate == 6 ? 6 : 7, std::vector<double>(nred, 0.0));
      for (i64 n = 0; n < N; ++n) {
        double R[9];
        quat_to_rot(&q[4 * n], R);
        const double fr = rot_free[n] ? 1.0 : 0.0, ft = trn_free[n] ? 1.0 : 0.0;
        for (int a = 0; a < 3; ++a)
          for (int i = 0; i < 3; ++i) {
            W[a][6 * n + 3 + i] = -ft * R[3 * i + a];
            W[3 + a][6 * n + i] = -0.5 * fr * R[3 * i + a];
          }
        if (W.size() > 6)
          for (int i = 0; i < 3; ++i) W[6][6 * n + 3 + i] = ft * t[3 * n + i];
      }
      // ORC_DEFLATE=9: a few intrinsics blocks shared by many images are a dense border of the reduced system; their
      // unit vectors as modes treat that border exactly (what a Schur complement on the border would do)
      if (deflate == 9 && nfree <= 8)
        for (i64 c = 6 * (N + S); c < nred; ++c) {
          W.emplace_back(nred, 0.0);
          W.back()[c] = 1.0;
        }
    }
    // Intrinsics blocks shared by many images are a dense border of the reduced system: eliminated densely (solve_bordered,
    // orc_lm.hpp) — what SPARSE_SCHUR's factorisation does to the arrowhead.  ORC_BORDER=0 (experiments) keeps round 3's plain
    // block-Jacobi PCG on the whole system.  The A-solves deflate the similarity modes (same solutions, fewer iterations).
    static const bool border_on = !(std::getenv("ORC_BORDER") && std::atoi(std::getenv("ORC_BORDER")) == 0);
    bool shared_border = false;
    for (i64 b = 0; b < K; ++b) shared_border = shared_border || intr_owner[b] < 0;
    bool all_shared = true;
    for (i64 b = 0; b < K; ++b) all_shared = all_shared && intr_owner[b] < 0;
    if (border_on && shared_border && all_shared && nfree > 0 && nfree <= 32 && nred > kDenseMax) {
      std::vector<std::vector<double>> Wb;
      if (S == 0) {
        Wb.assign(7, std::vector<double>(nred, 0.0));
        for (i64 n = 0; n < N; ++n) {
          double R[9];
          quat_to_rot(&q[4 * n], R);
          const double fr = rot_free[n] ? 1.0 : 0.0, ft = trn_free[n] ? 1.0 : 0.0;
          for (int a = 0; a < 3; ++a)
            for (int i = 0; i < 3; ++i) {
              Wb[a][6 * n + 3 + i] = -ft * R[3 * i + a];
              Wb[3 + a][6 * n + i] = -0.5 * fr * R[3 * i + a];
            }
          for (int i = 0; i < 3; ++i) Wb[6][6 * n + 3 + i] = ft * t[3 * n + i];
        }
        // frozen rotations / translations leave some modes identically zero: drop them
        std::vector<std::vector<double>> keep;
        for (auto& w : Wb)
          if (vdot(w, w) > 0.0) keep.push_back(std::move(w));
        Wb.swap(keep);
      }
      *lin = solve_bordered(
          nred, nfree, rhs, dy, pcg_tol, pcg_max, [&](const std::vector<double>& z, std::vector<double>& o) { apply(z, o); },
          [&](const std::vector<double>& r, std::vector<double>& z) { precond(r, z); }, relres, Wb.empty() ? nullptr : &Wb);
    } else
    *lin = solve_reduced(
        nred, rhs, dy, pcg_tol, pcg_max, [&](const std::vector<double>& z, std::vector<double>& o) { apply(z, o); },
        [&](const std::vector<double>& r, std::vector<double>& z) { precond(r, z); }, relres, (double)M, guess,
        W.empty() ? nullptr : &W);
    // deflation pays while a plain solve needs more than ~3 k iterations
    defl_on = W.empty() ? *lin > 3 * 7 : *lin - (i64)W.size() > (i64)W.size();
    if (warm) {
      dy_prev = dy;
      same_lin = true;
    }
    // back-substitution: dX = -u - tp(dy)
    point_pass(dy);
    std::vector<double> dX(3 * P);
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < 3 * P; ++i) dX[i] = -u[i] - tp[i];
    // model change: J delta = a_k(dy) + Jp dX
    std::vector<double> mterm(M);
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) {
      double a[2];
      cam_side(k, dy, a);
      const i64 p = pt[k];
      const double j0 = a[0] + Jp[6 * k] * dX[3 * p] + Jp[6 * k + 1] * dX[3 * p + 1] + Jp[6 * k + 2] * dX[3 * p + 2];
      const double j1 = a[1] + Jp[6 * k + 3] * dX[3 * p] + Jp[6 * k + 4] * dX[3 * p + 1] + Jp[6 * k + 5] * dX[3 * p + 2];
      mterm[k] = j0 * (rt[2 * k] + 0.5 * j0) + j1 * (rt[2 * k + 1] + 0.5 * j1);
    }
    *model_change = -chunked_sum(M, [&](i64 k) { return mterm[k]; });
    // candidate = Plus(x, delta)
    q2 = q;
    t2 = t;
    X2 = X;
    intr2 = intr;
    for (i64 n = 0; n < N + S; ++n) {
      const double* d = &dy[6 * n];
      const double nr = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      const double kk = nr > 0 ? std::sin(nr) / nr : 1.0;
      const double a[4] = {std::cos(nr), kk * d[0], kk * d[1], kk * d[2]};
      const double* b = &q[4 * n];
      q2[4 * n] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
      q2[4 * n + 1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
      q2[4 * n + 2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
      q2[4 * n + 3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
      for (int j = 0; j < 3; ++j) t2[3 * n + j] = t[3 * n + j] + d[3 + j];
    }
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < 3 * P; ++i) X2[i] = X[i] + dX[i];
    for (i64 b = 0; b < K; ++b)
      for (int j = 0; j < MAXP; ++j)
        if (icol[MAXP * b + j] >= 0) intr2[MAXP * b + j] = intr[MAXP * b + j] + dy[icol[MAXP * b + j]];
    auto diff2 = [&](const std::vector<double>& a, const std::vector<double>& b) {
      return chunked_sum((i64)a.size(), [&](i64 i) { const double d = a[i] - b[i]; return d * d; });
    };
    auto sq = [&](const std::vector<double>& a) { return chunked_sum((i64)a.size(), [&](i64 i) { return a[i] * a[i]; }); };
    const double sn = diff2(q2, q) + diff2(t2, t) + diff2(X2, X) + diff2(intr2, intr);
    *step_norm = std::sqrt(sn);
    *x_norm = std::sqrt(sq(q) + sq(t) + sq(X) + sq(intr));
    *cand_cost = cost_at(q2, t2, X2, intr2);
    return std::isfinite(sn);
  }

  // Block-Jacobi preconditioner of the reduced system: per camera a joint block over its 6 pose columns and the
  // free columns of an intrinsics block only it uses; a separate block per shared intrinsics block.
  // Diagonal blocks of S assembled per observation: W^T W - W^T Jp Hpp^-1 Jp^T W, W = [Jc Ji] (exact for the pose
  // part; for shared intrinsics it drops the cross terms between observations of one point — it is a preconditioner).
  bool build_preconditioner() {
    Mblk.assign((size_t)BJ2 * N, 0.0);
    std::vector<double> acc((size_t)BJ2 * N);
    bycam.reduce<BJ2>(acc.data(), rev, [&](i64 k, double* a) {
      const i64 n = cam[k], b = ik[k];
      const bool joint = intr_owner[b] == n;
      double W[2][BJ];
      for (int j = 0; j < 6; ++j) {
        W[0][j] = Jc[12 * k + j];
        W[1][j] = Jc[12 * k + 6 + j];
      }
      for (int j = 0; j < MAXP; ++j) {
        W[0][6 + j] = joint ? Ji[MP2 * k + j] : 0.0;
        W[1][6 + j] = joint ? Ji[MP2 * k + MAXP + j] : 0.0;
      }
      const double* H = &Hinv[9 * (i64)pt[k]];
      // G = Jp Hpp^-1 Jp^T (2x2)
      double JH[2][3];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
          JH[i][j] = Jp[6 * k + 3 * i] * H[j] + Jp[6 * k + 3 * i + 1] * H[3 + j] + Jp[6 * k + 3 * i + 2] * H[6 + j];
      double G[2][2];
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
          G[i][j] = JH[i][0] * Jp[6 * k + 3 * j] + JH[i][1] * Jp[6 * k + 3 * j + 1] + JH[i][2] * Jp[6 * k + 3 * j + 2];
      const double A00 = 1.0 - G[0][0], A01 = -G[0][1], A10 = -G[1][0], A11 = 1.0 - G[1][1];
      const int w = joint ? BJ : 6;
      for (int i = 0; i < w; ++i) {
        const double l0 = W[0][i] * A00 + W[1][i] * A10, l1 = W[0][i] * A01 + W[1][i] * A11;
        for (int j = 0; j < w; ++j) a[BJ * i + j] += l0 * W[0][j] + l1 * W[1][j];
      }
    });
    bool ok = true;
#pragma omp parallel for schedule(static)
    for (i64 n = 0; n < N; ++n) {
      double B[BJ2];
      for (int i = 0; i < BJ; ++i)
        for (int j = 0; j < BJ; ++j) B[BJ * i + j] = 0.5 * (acc[(size_t)BJ2 * n + BJ * i + j] + acc[(size_t)BJ2 * n + BJ * j + i]);
      const i64 b = cam_intr[n];
      const bool joint = intr_owner[b] == n;
      for (int j = 0; j < 6; ++j) B[(BJ + 1) * j] += dred[6 * n + j];
      for (int j = 0; j < MAXP; ++j) {
        const int32_t col = joint ? icol[MAXP * b + j] : -1;
        if (col >= 0)
          B[(BJ + 1) * (6 + j)] += dred[col];
        else {  // not part of this block: identity row / column
          for (int i = 0; i < BJ; ++i) B[BJ * (6 + j) + i] = B[BJ * i + 6 + j] = 0.0;
          B[(BJ + 1) * (6 + j)] = 1.0;
        }
      }
      if (!spd_inverse(B, BJ)) {
#pragma omp atomic write
        ok = false;
      }
      std::memcpy(&Mblk[(size_t)BJ2 * n], B, sizeof B);
    }
    if (!ok) return false;
    if (S > 0) {  // one 6 x 6 block per sensor block: Js^T (I - Jp Hpp^-1 Jp^T) Js summed per observation + damping
      std::vector<double> accs((size_t)36 * S);
      bysens.reduce<36>(accs.data(), rev, [&](i64 ee, double* a) {
        const i64 k = sobs[ee];
        const double* H = &Hinv[9 * (i64)pt[k]];
        double JH[2][3];
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 3; ++j)
            JH[i][j] = Jp[6 * k + 3 * i] * H[j] + Jp[6 * k + 3 * i + 1] * H[3 + j] + Jp[6 * k + 3 * i + 2] * H[6 + j];
        double G[2][2];
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j)
            G[i][j] = JH[i][0] * Jp[6 * k + 3 * j] + JH[i][1] * Jp[6 * k + 3 * j + 1] + JH[i][2] * Jp[6 * k + 3 * j + 2];
        const double A00 = 1.0 - G[0][0], A01 = -G[0][1], A10 = -G[1][0], A11 = 1.0 - G[1][1];
        for (int i = 0; i < 6; ++i) {
          const double l0 = Js[12 * k + i] * A00 + Js[12 * k + 6 + i] * A10, l1 = Js[12 * k + i] * A01 + Js[12 * k + 6 + i] * A11;
          for (int j = 0; j < 6; ++j) a[6 * i + j] += l0 * Js[12 * k + j] + l1 * Js[12 * k + 6 + j];
        }
      });
      Msblk.assign((size_t)36 * S, 0.0);
      for (i64 sb = 0; sb < S; ++sb) {
        double B[36];
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) B[6 * i + j] = 0.5 * (accs[(size_t)36 * sb + 6 * i + j] + accs[(size_t)36 * sb + 6 * j + i]);
        for (int j = 0; j < 6; ++j) B[7 * j] += dred[6 * (N + sb) + j];
        if (!spd_inverse(B, 6)) return false;
        std::memcpy(&Msblk[(size_t)36 * sb], B, sizeof B);
      }
    }
    Miblk.assign((size_t)MPP * K, 0.0);
    bool any_shared = false;
    for (i64 b = 0; b < K; ++b) any_shared = any_shared || intr_owner[b] < 0;
    if (any_shared && nfree > 0) {
      std::vector<double> acci((size_t)MPP * K);
      byintr.reduce<MPP>(acci.data(), rev, [&](i64 k, double* a) {
        if (intr_owner[ik[k]] >= 0) return;
        const double* H = &Hinv[9 * (i64)pt[k]];
        double JH[2][3];
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 3; ++j)
            JH[i][j] = Jp[6 * k + 3 * i] * H[j] + Jp[6 * k + 3 * i + 1] * H[3 + j] + Jp[6 * k + 3 * i + 2] * H[6 + j];
        double G[2][2];
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j)
            G[i][j] = JH[i][0] * Jp[6 * k + 3 * j] + JH[i][1] * Jp[6 * k + 3 * j + 1] + JH[i][2] * Jp[6 * k + 3 * j + 2];
        const double A00 = 1.0 - G[0][0], A01 = -G[0][1], A10 = -G[1][0], A11 = 1.0 - G[1][1];
        for (int i = 0; i < MAXP; ++i) {
          const double l0 = Ji[MP2 * k + i] * A00 + Ji[MP2 * k + MAXP + i] * A10, l1 = Ji[MP2 * k + i] * A01 + Ji[MP2 * k + MAXP + i] * A11;
          for (int j = 0; j < MAXP; ++j) a[MAXP * i + j] += l0 * Ji[MP2 * k + j] + l1 * Ji[MP2 * k + MAXP + j];
        }
      });
      for (i64 b = 0; b < K; ++b) {
        if (intr_owner[b] >= 0) continue;
        double B[MPP];
        for (int i = 0; i < MAXP; ++i)
          for (int j = 0; j < MAXP; ++j) B[MAXP * i + j] = 0.5 * (acci[(size_t)MPP * b + MAXP * i + j] + acci[(size_t)MPP * b + MAXP * j + i]);
        for (int j = 0; j < MAXP; ++j) {
          const int32_t col = icol[MAXP * b + j];
          if (col >= 0)
            B[(MAXP + 1) * j] += dred[col];
          else {
            for (int i = 0; i < MAXP; ++i) B[MAXP * j + i] = B[MAXP * i + j] = 0.0;
            B[(MAXP + 1) * j] = 1.0;
          }
        }
        if (!spd_inverse(B, MAXP)) return false;
        std::memcpy(&Miblk[(size_t)MPP * b], B, sizeof B);
      }
    }
    return true;
  }

  void precond(const std::vector<double>& r, std::vector<double>& z) {
#pragma omp parallel for schedule(static)
    for (i64 n = 0; n < N; ++n) {
      const i64 b = cam_intr[n];
      const bool joint = intr_owner[b] == n;
      double rv[BJ], zv[BJ];
      for (int j = 0; j < 6; ++j) rv[j] = r[6 * n + j];
      for (int j = 0; j < MAXP; ++j) {
        const int32_t col = joint ? icol[MAXP * b + j] : -1;
        rv[6 + j] = col >= 0 ? r[col] : 0.0;
      }
      const double* B = &Mblk[(size_t)BJ2 * n];
      for (int i = 0; i < BJ; ++i) {
        double s = 0.0;
        for (int j = 0; j < BJ; ++j) s += B[BJ * i + j] * rv[j];
        zv[i] = s;
      }
      for (int j = 0; j < 6; ++j) z[6 * n + j] = zv[j];
      for (int j = 0; j < MAXP; ++j) {
        const int32_t col = joint ? icol[MAXP * b + j] : -1;
        if (col >= 0) z[col] = zv[6 + j];
      }
    }
    for (i64 sb = 0; sb < S; ++sb) {
      const double* B = &Msblk[(size_t)36 * sb];
      const i64 o = 6 * (N + sb);
      for (int i = 0; i < 6; ++i) {
        double sum = 0.0;
        for (int j = 0; j < 6; ++j) sum += B[6 * i + j] * r[o + j];
        z[o + i] = sum;
      }
    }
    for (i64 b = 0; b < K; ++b) {
      if (intr_owner[b] >= 0) continue;
      const double* B = &Miblk[(size_t)MPP * b];
      for (int i = 0; i < MAXP; ++i) {
        const int32_t ci = icol[MAXP * b + i];
        if (ci < 0) continue;
        double s = 0.0;
        for (int j = 0; j < MAXP; ++j) {
          const int32_t cj = icol[MAXP * b + j];
          if (cj >= 0) s += B[MAXP * i + j] * r[cj];
        }
        z[ci] = s;
      }
    }
  }

  void accept() override {
    q.swap(q2);
    t.swap(t2);
    X.swap(X2);
    intr.swap(intr2);
  }
};

}  // namespace
}  // namespace orc

extern "C" {
using orc::i64;

// Arrays as gsfm_ba_problem (include/gsfm.h); cam_q (w,x,y,z), intr_params [K][MAXP] (8; 16 in the unit of orc_ba_wide.cc).
// Known rigs: image_frame [I], image_cam_from_rig [I][7] (qw,qx,qy,qz,tx,ty,tz), image_intr [I] (all NULL = trivial
// rigs): obs_cam then indexes images and cam_intr is ignored.
int ORC_BA_ENTRY(int32_t num_cams, int32_t num_intr, int32_t fixed_cam, i64 num_pts, const i64* pt_offset,
                 const int32_t* obs_cam, const double* obs_xy, const int32_t* cam_intr, const int32_t* intr_model,
                 const orc::BaOptionsC* o, double* cam_q_inout, double* cam_t_inout, double* pt_xyz_inout,
                 double* intr_params_inout, orc::BaReport* rep, int32_t num_threads, const int32_t* image_frame,
                 const double* image_cam_from_rig, const int32_t* image_intr, int32_t num_sensors,
                 const int32_t* image_sensor, double* sensor_cam_from_rig_inout) {
  using namespace orc;
  const double t0 = omp_get_wtime();
  if (num_threads > 0) omp_set_num_threads(num_threads);
  Ba g;
  g.N = num_cams;
  g.K = num_intr;
  const bool have_sens = image_frame && num_sensors > 0 && image_sensor && sensor_cam_from_rig_inout;
  g.S = (have_sens && o->optimize_rig_poses) ? num_sensors : 0;
  std::vector<i64> used_pts;
  g.poff.push_back(0);
  for (i64 p = 0; p < num_pts; ++p) {
    if (pt_offset[p + 1] - pt_offset[p] < o->min_num_view_per_track) continue;  // ba.cc:122
    const i64 id = (i64)used_pts.size();
    used_pts.push_back(p);
    for (i64 k = pt_offset[p]; k < pt_offset[p + 1]; ++k) {
      g.pt.push_back((int32_t)id);
      if (image_frame) {
        const i64 im = obs_cam[k];
        g.cam.push_back(image_frame[im]);
        g.ik.push_back(image_intr[im]);
        // the cam_from_rig of an image with a sensor entry is the table's value (constant, or the start of its block)
        const double* cfr = (have_sens && image_sensor[im] >= 0) ? sensor_cam_from_rig_inout + 7 * image_sensor[im]
                                                                   : image_cam_from_rig + 7 * im;
        double R[9];
        quat_to_rot(cfr, R);
        for (int j = 0; j < 9; ++j) g.sens.push_back(R[j]);
        for (int j = 0; j < 3; ++j) g.sens.push_back(cfr[4 + j]);
        if (g.S > 0) {
          g.sblk.push_back(image_sensor[im]);
          if (image_sensor[im] >= 0) {
            g.sobs.push_back((int32_t)g.cam.size() - 1);
            g.sown.push_back(image_sensor[im]);
          }
        }
      } else {
        g.cam.push_back(obs_cam[k]);
        g.ik.push_back(cam_intr[obs_cam[k]]);
      }
      g.xy.push_back(obs_xy[2 * k]);
      g.xy.push_back(obs_xy[2 * k + 1]);
    }
    g.poff.push_back((i64)g.cam.size());
  }
  g.P = (i64)used_pts.size();
  g.M = (i64)g.cam.size();
  std::memset(rep, 0, sizeof *rep);
  rep->threads = omp_get_max_threads();
  if (g.M == 0) return -5;
  if (image_frame)
    g.cam_intr.assign(g.N, 0);  // unused with rigs (every intrinsics block counts as shared: separate preconditioner blocks)
  else
    g.cam_intr.assign(cam_intr, cam_intr + g.N);
  g.model.assign(intr_model, intr_model + g.K);
  for (i64 b = 0; b < g.K; ++b)
    if (g.model[b] < 0 || g.model[b] >= kNumModels || kNumParams[g.model[b]] < 0) return -7;
  g.bycam.build(g.N, g.M, g.cam.data());
  g.byintr.build(g.K, g.M, g.ik.data());
  if (g.S > 0) g.bysens.build(g.S, (i64)g.sobs.size(), g.sown.data());
  g.loss = {o->thres_loss_function, 1.0};
  g.rot_free.assign(g.N, o->optimize_rotations ? 1 : 0);
  g.trn_free.assign(g.N, o->optimize_translation ? 1 : 0);
  if (fixed_cam >= 0) g.rot_free[fixed_cam] = g.trn_free[fixed_cam] = 0;  // ba.cc:261-266
  g.free_par.assign(MAXP * g.K, 0);
  g.icol.assign(MAXP * g.K, -1);
  g.nfree = 0;
  for (i64 b = 0; b < g.K; ++b) {
    if (!o->optimize_intrinsics && !o->optimize_principal_point) continue;  // SetParameterBlockConstant, ba.cc:273-293
    const int m = g.model[b];
    for (int j = 0; j < kNumParams[m]; ++j) g.free_par[MAXP * b + j] = 1;
    if (o->optimize_intrinsics && !o->optimize_principal_point) g.free_par[MAXP * b + kPP[m][0]] = g.free_par[MAXP * b + kPP[m][1]] = 0;
    for (int j = 0; j < MAXP; ++j)
      if (g.free_par[MAXP * b + j]) g.icol[MAXP * b + j] = (int32_t)(6 * (g.N + g.S) + g.nfree++);
  }
  g.nred = 6 * (g.N + g.S) + g.nfree;
  g.intr_owner.assign(g.K, -2);
  if (!image_frame)
    for (i64 n = 0; n < g.N; ++n) {
      int32_t& ow = g.intr_owner[cam_intr[n]];
      ow = (ow == -2) ? (int32_t)n : -1;
    }
  for (i64 b = 0; b < g.K; ++b)
    if (g.intr_owner[b] == -2) g.intr_owner[b] = -1;
  g.mpt = o->optimize_points ? 1.0 : 0.0;
  g.lm_lo = o->min_lm_diagonal;
  g.lm_hi = o->max_lm_diagonal;
  g.rev = o->order != 0;
  g.pcg_tol = o->pcg_relative_tolerance;
  g.pcg_max = o->pcg_max_iterations;
  g.q.assign(cam_q_inout, cam_q_inout + 4 * g.N);
  g.t.assign(cam_t_inout, cam_t_inout + 3 * g.N);
  for (i64 sb = 0; sb < g.S; ++sb) {
    for (int j = 0; j < 4; ++j) g.q.push_back(sensor_cam_from_rig_inout[7 * sb + j]);
    for (int j = 0; j < 3; ++j) g.t.push_back(sensor_cam_from_rig_inout[7 * sb + 4 + j]);
  }
  g.intr.assign(intr_params_inout, intr_params_inout + MAXP * g.K);
  g.X.resize(3 * g.P);
  for (i64 i = 0; i < g.P; ++i)
    for (int j = 0; j < 3; ++j) g.X[3 * i + j] = pt_xyz_inout[3 * used_pts[i] + j];
  LmOptions lo;
  lo.max_num_iterations = o->max_num_iterations;
  lo.function_tolerance = o->function_tolerance;
  lo.gradient_tolerance = o->gradient_tolerance;
  lo.parameter_tolerance = o->parameter_tolerance;
  lo.initial_trust_region_radius = o->initial_trust_region_radius;
  lo.max_trust_region_radius = o->max_trust_region_radius;
  lo.min_trust_region_radius = o->min_trust_region_radius;
  lo.min_relative_decrease = o->min_relative_decrease;
  lo.min_lm_diagonal = o->min_lm_diagonal;
  lo.max_lm_diagonal = o->max_lm_diagonal;
  lo.jacobi_scaling = o->jacobi_scaling;
  lo.max_num_consecutive_invalid_steps = o->max_num_consecutive_invalid_steps;
  lo.verbose = o->verbose;
  LmSummary s;
  lm_minimize(g, lo, &s);
  std::memcpy(cam_q_inout, g.q.data(), sizeof(double) * 4 * g.N);
  std::memcpy(cam_t_inout, g.t.data(), sizeof(double) * 3 * g.N);
  for (i64 sb = 0; sb < g.S; ++sb) {
    for (int j = 0; j < 4; ++j) sensor_cam_from_rig_inout[7 * sb + j] = g.q[4 * (g.N + sb) + j];
    for (int j = 0; j < 3; ++j) sensor_cam_from_rig_inout[7 * sb + 4 + j] = g.t[3 * (g.N + sb) + j];
  }
  std::memcpy(intr_params_inout, g.intr.data(), sizeof(double) * MAXP * g.K);
  for (i64 i = 0; i < g.P; ++i)
    for (int j = 0; j < 3; ++j) pt_xyz_inout[3 * used_pts[i] + j] = g.X[3 * i + j];
  rep->iterations = s.iterations;
  rep->successful_steps = s.successful_steps;
  rep->termination = s.termination;
  rep->usable = s.usable;
  rep->linear_iterations = s.linear_iterations;
  rep->initial_cost = s.initial_cost;
  rep->final_cost = s.final_cost;
  rep->max_linear_residual = s.max_linear_residual;
  rep->seconds_linear = s.seconds_linear;
  rep->seconds_total = omp_get_wtime() - t0;
  return s.usable ? 0 : -6;
}
}
