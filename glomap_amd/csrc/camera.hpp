// camera.hpp — COLMAP camera models (ImgFromCam) with analytic derivatives, device side.
//
// Replaces the autodiff of colmap::ReprojErrorCostFunctor<CameraModel> that the reference
// instantiates through colmap::CreateCameraCostFunction (glomap/estimators/bundle_adjustment.cc:135-146).
// COLMAP (pinned b6b7b54e, thirdparty/CMakeLists.txt:23-28) is un-vendored; the models follow
// their published definitions (colmap/sensor/models.h), parameter order as in COLMAP:
//   SIMPLE_PINHOLE f,cx,cy | PINHOLE fx,fy,cx,cy | SIMPLE_RADIAL f,cx,cy,k | RADIAL f,cx,cy,k1,k2
//   OPENCV fx,fy,cx,cy,k1,k2,p1,p2 | OPENCV_FISHEYE fx,fy,cx,cy,k1,k2,k3,k4 | FOV fx,fy,cx,cy,omega
//   SIMPLE_RADIAL_FISHEYE f,cx,cy,k | RADIAL_FISHEYE f,cx,cy,k1,k2
// Fisheye models: (u, v) = (x/z, y/z), r = |(u, v)|, theta = atan(r); the equidistant coordinates (u, v) theta / r are
// distorted radially in theta: pixel = f (u, v) theta_d / r + c, theta_d = theta (1 + k1 theta^2 + k2 theta^4 + ...)
// (r <= eps: (u, v) itself).  FOV: pixel = f (u, v) factor + c, factor = atan(2 r tan(omega / 2)) / (r omega) with COLMAP's
// two series branches for omega^2 < 1e-4 and r^2 < 1e-4.
//
// Intrinsics blocks are KP doubles wide: KP = 8 (GSFM_CAMERA_MAX_PARAMS, the nine models above) or KP = 16
// (GSFM_CAMERA_MAX_PARAMS_WIDE) for the models with more than eight parameters, which exist only in the KP = 16 instances:
//   FULL_OPENCV fx,fy,cx,cy,k1,k2,p1,p2,k3,k4,k5,k6: (u, v) (1 + k1 r^2 + k2 r^4 + k3 r^6) / (1 + k4 r^2 + k5 r^4 + k6 r^6)
//     + tangential (2 p1 u v + p2 (r^2 + 2 u^2), 2 p2 u v + p1 (r^2 + 2 v^2))
//   THIN_PRISM_FISHEYE fx,fy,cx,cy,k1,k2,p1,p2,k3,k4,sx1,sy1: equidistant (u, v) theta / r, then polynomial radial
//     (k1 r^2 + k2 r^4 + k3 r^6 + k4 r^8) + tangential + thin prism (sx1 r^2, sy1 r^2) in the equidistant coordinates
//   RAD_TAN_THIN_PRISM_FISHEYE fx,fy,cx,cy,k0..k5,p0,p1,s0..s3: equidistant (u, v) theta / r, radial
//     (uh, vh) = (u, v) (1 + k0 r^2 + ... + k5 r^12), then tangential (p0 (2 uh^2 + rh^2) + 2 p1 uh vh,
//     p1 (2 vh^2 + rh^2) + 2 p0 uh vh) and thin prism (s0 rh^2 + s1 rh^4, s2 rh^2 + s3 rh^4) on the radially distorted (uh, vh)
#pragma once

#include "../../include/gsfm.h"
#include "linalg.hpp"

namespace gsfm {

template <int KP>
struct ObsGeomT {
  V3 a;             // R X  (camera-frame point minus translation)
  double px, py;    // projected pixel
  double Jx[2][3];  // d pixel / d x_cam
  double Jp[2][KP]; // d pixel / d params
  bool valid;       // point in front of the camera; otherwise residual and Jacobians are zero
};

// Equidistant fisheye with radial distortion in theta: (xd, yd) = m (u, v), m = theta_d / r, theta = atan(r),
// theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8); dm_r = (dm/dr) / r; sk = theta / r (d m / d k_j =
// sk theta^(2j)); t2 .. t8 = the even powers of theta.  r <= eps: the mapping is the identity there.
__device__ __forceinline__ void fisheye_core(double r2, double k1, double k2, double k3, double k4, double& m, double& dm_r,
                                             double& sk, double& t2, double& t4, double& t6, double& t8) {
  const double r = sqrt(r2);
  const bool big = r > 2.220446049250313e-16;
  const double th = big ? atan(r) : r;
  t2 = th * th;
  t4 = t2 * t2;
  t6 = t4 * t2;
  t8 = t4 * t4;
  const double poly = 1.0 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8;
  const double dpoly = 1.0 + 3.0 * k1 * t2 + 5.0 * k2 * t4 + 7.0 * k3 * t6 + 9.0 * k4 * t8;  // d theta_d / d theta
  if (big) {
    m = th * poly / r;
    dm_r = (dpoly / (1.0 + r2) - m) / r2;
    sk = th / r;
  } else {
    m = poly;
    dm_r = 0.0;
    sk = 1.0;
  }
}

// The models with more than eight parameters (KP = 16 instances only).  All three are "(a, b) -> (ad, bd)" distortions of
// either (u, v) itself (FULL_OPENCV) or of the equidistant coordinates (a, b) = (u, v) theta / r (the two thin-prism fisheye
// models); J = d(ad, bd) / d(a, b) is chained with the equidistant map's Jacobian E = s I + (s' / r) (u, v)(u, v)^T, s = theta / r.
// Returns false for the other model ids.
template <int KP>
__device__ __forceinline__ bool distort_project_wide16(int model, const double* __restrict__ p, double u, double v, double r2,
                                                       double& px, double& py, double (&Juv)[4], double (&Jp)[2][KP]) {
  static_assert(KP >= 16, "the 12 / 16-parameter models need the wide intrinsics block");
  if (model != GSFM_CAMERA_FULL_OPENCV && model != GSFM_CAMERA_THIN_PRISM_FISHEYE &&
      model != GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE)
    return false;
  const double fx = p[0], fy = p[1];
  double a = u, b = v;          // input of the distortion
  double e0 = 1.0, e1 = 0.0, e2 = 0.0, e3 = 1.0;  // d(a, b) / d(u, v)
  if (model != GSFM_CAMERA_FULL_OPENCV) {
    const double r = sqrt(r2);
    if (r > 2.220446049250313e-16) {
      const double s = atan(r) / r;
      const double sp = (1.0 / (1.0 + r2) - s) / r2;  // s' / r
      a = s * u;
      b = s * v;
      e0 = s + u * u * sp;
      e1 = u * v * sp;
      e2 = e1;
      e3 = s + v * v * sp;
    }
  }
  const double a2 = a * a, b2 = b * b, ab = a * b, q2 = a2 + b2;
  double ad, bd, j0, j1, j2, j3;  // distorted point, d(ad, bd) / d(a, b)
  if (model == GSFM_CAMERA_FULL_OPENCV) {
    const double k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7], k3 = p[8], k4 = p[9], k5 = p[10], k6 = p[11];
    const double q4 = q2 * q2, q6 = q4 * q2;
    const double num = 1.0 + k1 * q2 + k2 * q4 + k3 * q6, den = 1.0 + k4 * q2 + k5 * q4 + k6 * q6;
    const double iden = 1.0 / den, rad = num * iden;
    const double drad = ((k1 + 2.0 * k2 * q2 + 3.0 * k3 * q4) - rad * (k4 + 2.0 * k5 * q2 + 3.0 * k6 * q4)) * iden;
    ad = a * rad + 2.0 * p1 * ab + p2 * (q2 + 2.0 * a2);
    bd = b * rad + 2.0 * p2 * ab + p1 * (q2 + 2.0 * b2);
    j0 = rad + 2.0 * a2 * drad + 2.0 * p1 * b + 6.0 * p2 * a;
    j1 = 2.0 * ab * drad + 2.0 * p1 * a + 2.0 * p2 * b;
    j2 = 2.0 * ab * drad + 2.0 * p2 * b + 2.0 * p1 * a;
    j3 = rad + 2.0 * b2 * drad + 2.0 * p2 * a + 6.0 * p1 * b;
    const double n1 = q2 * iden, n2 = q4 * iden, n3 = q6 * iden;  // d rad / d (k1, k2, k3); d rad / d (k4, k5, k6) = -rad (n1, n2, n3)
    Jp[0][4] = fx * a * n1; Jp[1][4] = fy * b * n1;
    Jp[0][5] = fx * a * n2; Jp[1][5] = fy * b * n2;
    Jp[0][6] = fx * 2.0 * ab; Jp[1][6] = fy * (q2 + 2.0 * b2);
    Jp[0][7] = fx * (q2 + 2.0 * a2); Jp[1][7] = fy * 2.0 * ab;
    Jp[0][8] = fx * a * n3; Jp[1][8] = fy * b * n3;
    Jp[0][9] = -fx * a * rad * n1; Jp[1][9] = -fy * b * rad * n1;
    Jp[0][10] = -fx * a * rad * n2; Jp[1][10] = -fy * b * rad * n2;
    Jp[0][11] = -fx * a * rad * n3; Jp[1][11] = -fy * b * rad * n3;
  } else if (model == GSFM_CAMERA_THIN_PRISM_FISHEYE) {
    const double k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7], k3 = p[8], k4 = p[9], sx1 = p[10], sy1 = p[11];
    const double q4 = q2 * q2, q6 = q4 * q2, q8 = q4 * q4;
    const double rad = k1 * q2 + k2 * q4 + k3 * q6 + k4 * q8;
    const double drad = k1 + 2.0 * k2 * q2 + 3.0 * k3 * q4 + 4.0 * k4 * q6;
    ad = a + a * rad + 2.0 * p1 * ab + p2 * (q2 + 2.0 * a2) + sx1 * q2;
    bd = b + b * rad + 2.0 * p2 * ab + p1 * (q2 + 2.0 * b2) + sy1 * q2;
    j0 = 1.0 + rad + 2.0 * a2 * drad + 2.0 * p1 * b + 6.0 * p2 * a + 2.0 * sx1 * a;
    j1 = 2.0 * ab * drad + 2.0 * p1 * a + 2.0 * p2 * b + 2.0 * sx1 * b;
    j2 = 2.0 * ab * drad + 2.0 * p2 * b + 2.0 * p1 * a + 2.0 * sy1 * a;
    j3 = 1.0 + rad + 2.0 * b2 * drad + 2.0 * p2 * a + 6.0 * p1 * b + 2.0 * sy1 * b;
    Jp[0][4] = fx * a * q2; Jp[1][4] = fy * b * q2;
    Jp[0][5] = fx * a * q4; Jp[1][5] = fy * b * q4;
    Jp[0][6] = fx * 2.0 * ab; Jp[1][6] = fy * (q2 + 2.0 * b2);
    Jp[0][7] = fx * (q2 + 2.0 * a2); Jp[1][7] = fy * 2.0 * ab;
    Jp[0][8] = fx * a * q6; Jp[1][8] = fy * b * q6;
    Jp[0][9] = fx * a * q8; Jp[1][9] = fy * b * q8;
    Jp[0][10] = fx * q2;
    Jp[1][11] = fy * q2;
  } else {  // GSFM_CAMERA_RAD_TAN_THIN_PRISM_FISHEYE
    const double p0 = p[10], p1 = p[11], s0 = p[12], s1 = p[13], s2 = p[14], s3 = p[15];
    double qp[6];  // q2^(i + 1)
    qp[0] = q2;
#pragma unroll
    for (int i = 1; i < 6; ++i) qp[i] = qp[i - 1] * q2;
    double R = 1.0, dR = p[4];
#pragma unroll
    for (int i = 0; i < 6; ++i) R += p[4 + i] * qp[i];
#pragma unroll
    for (int i = 1; i < 6; ++i) dR += (double)(i + 1) * p[4 + i] * qp[i - 1];
    const double uh = R * a, vh = R * b;
    const double uh2 = uh * uh, vh2 = vh * vh, uhvh = uh * vh, h2 = uh2 + vh2, h4 = h2 * h2;
    ad = uh + p0 * (2.0 * uh2 + h2) + 2.0 * p1 * uhvh + s0 * h2 + s1 * h4;
    bd = vh + p1 * (2.0 * vh2 + h2) + 2.0 * p0 * uhvh + s2 * h2 + s3 * h4;
    // T = d(ad, bd) / d(uh, vh), then the radial stage d(uh, vh) / d(a, b) = R I + 2 dR (a, b)(a, b)^T
    const double sx = s0 + 2.0 * s1 * h2, sy = s2 + 2.0 * s3 * h2;
    const double T0 = 1.0 + 6.0 * p0 * uh + 2.0 * p1 * vh + 2.0 * uh * sx;
    const double T1 = 2.0 * p0 * vh + 2.0 * p1 * uh + 2.0 * vh * sx;
    const double T2 = 2.0 * p1 * uh + 2.0 * p0 * vh + 2.0 * uh * sy;
    const double T3 = 1.0 + 6.0 * p1 * vh + 2.0 * p0 * uh + 2.0 * vh * sy;
    const double r0 = R + 2.0 * a2 * dR, r1 = 2.0 * ab * dR, r3 = R + 2.0 * b2 * dR;
    j0 = T0 * r0 + T1 * r1;
    j1 = T0 * r1 + T1 * r3;
    j2 = T2 * r0 + T3 * r1;
    j3 = T2 * r1 + T3 * r3;
    const double ta = T0 * a + T1 * b, tb = T2 * a + T3 * b;  // T (a, b): d(ad, bd) / d k_i = T (a, b) q2^(i + 1)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      Jp[0][4 + i] = fx * ta * qp[i];
      Jp[1][4 + i] = fy * tb * qp[i];
    }
    Jp[0][10] = fx * (2.0 * uh2 + h2); Jp[1][10] = fy * 2.0 * uhvh;
    Jp[0][11] = fx * 2.0 * uhvh; Jp[1][11] = fy * (2.0 * vh2 + h2);
    Jp[0][12] = fx * h2;
    Jp[0][13] = fx * h4;
    Jp[1][14] = fy * h2;
    Jp[1][15] = fy * h4;
  }
  px = fx * ad + p[2];
  py = fy * bd + p[3];
  Jp[0][0] = ad; Jp[1][1] = bd;
  Jp[0][2] = 1.0; Jp[1][3] = 1.0;
  Juv[0] = fx * (j0 * e0 + j1 * e2);
  Juv[1] = fx * (j0 * e1 + j1 * e3);
  Juv[2] = fy * (j2 * e0 + j3 * e2);
  Juv[3] = fy * (j2 * e1 + j3 * e3);
  return true;
}

// pixel = f(u, v; params) and its derivatives w.r.t. (u, v) -> Juv[4] = {du/du, du/dv, dv/du, dv/dv}
// The fisheye / FOV models (ids >= 5).  Kept out of the sweeps that never see them: every kernel that projects is
// instantiated twice (template parameter WIDE) and the host picks the lean instance when all cameras use the five polynomial
// models — with the transcendental branches inlined k_ba_phaseB needs 278 instead of 219 registers (1 wave per SIMD) and
// runs three times slower on configs[3].
template <int KP>
__device__ __forceinline__ void distort_project_wide(int model, const double* __restrict__ p, double u, double v, double r2,
                                                     double& px, double& py, double (&Juv)[4], double (&Jp)[2][KP]) {
  if constexpr (KP >= 16) {
    if (distort_project_wide16(model, p, u, v, r2, px, py, Juv, Jp)) return;
  }
  switch (model) {
    case GSFM_CAMERA_OPENCV_FISHEYE: {
      const double fx = p[0], fy = p[1];
      double m, dm_r, sk, t2, t4, t6, t8;
      fisheye_core(r2, p[4], p[5], p[6], p[7], m, dm_r, sk, t2, t4, t6, t8);
      px = fx * u * m + p[2];
      py = fy * v * m + p[3];
      Juv[0] = fx * (m + u * u * dm_r);
      Juv[1] = fx * (u * v * dm_r);
      Juv[2] = fy * (u * v * dm_r);
      Juv[3] = fy * (m + v * v * dm_r);
      Jp[0][0] = u * m; Jp[1][1] = v * m;
      Jp[0][2] = 1.0; Jp[1][3] = 1.0;
      Jp[0][4] = fx * u * sk * t2; Jp[1][4] = fy * v * sk * t2;
      Jp[0][5] = fx * u * sk * t4; Jp[1][5] = fy * v * sk * t4;
      Jp[0][6] = fx * u * sk * t6; Jp[1][6] = fy * v * sk * t6;
      Jp[0][7] = fx * u * sk * t8; Jp[1][7] = fy * v * sk * t8;
      break;
    }
    case GSFM_CAMERA_SIMPLE_RADIAL_FISHEYE:
    case GSFM_CAMERA_RADIAL_FISHEYE: {
      const double f = p[0];
      const double k2 = model == GSFM_CAMERA_RADIAL_FISHEYE ? p[4] : 0.0;
      double m, dm_r, sk, t2, t4, t6, t8;
      fisheye_core(r2, p[3], k2, 0.0, 0.0, m, dm_r, sk, t2, t4, t6, t8);
      px = f * u * m + p[1];
      py = f * v * m + p[2];
      Juv[0] = f * (m + u * u * dm_r);
      Juv[1] = f * (u * v * dm_r);
      Juv[2] = Juv[1];
      Juv[3] = f * (m + v * v * dm_r);
      Jp[0][0] = u * m; Jp[1][0] = v * m;
      Jp[0][1] = 1.0; Jp[1][2] = 1.0;
      Jp[0][3] = f * u * sk * t2; Jp[1][3] = f * v * sk * t2;
      if (model == GSFM_CAMERA_RADIAL_FISHEYE) {
        Jp[0][4] = f * u * sk * t4;
        Jp[1][4] = f * v * sk * t4;
      }
      break;
    }
    case GSFM_CAMERA_FOV: {
      const double fx = p[0], fy = p[1], om = p[4];
      const double om2 = om * om;
      double fac, dfac_r2, dfac_om;  // factor, d factor / d r^2, d factor / d omega
      if (om2 < 1e-4) {
        fac = om2 * r2 / 3.0 - om2 / 12.0 + 1.0;
        dfac_r2 = om2 / 3.0;
        dfac_om = 2.0 * om * (r2 / 3.0 - 1.0 / 12.0);
      } else if (r2 < 1e-4) {
        const double t = tan(0.5 * om), t2 = t * t;
        fac = (-2.0 * t * (4.0 * r2 * t2 - 3.0)) / (3.0 * om);
        dfac_r2 = -8.0 * t * t2 / (3.0 * om);
        const double dt = 0.5 * (1.0 + t2);  // d tan(omega / 2) / d omega
        dfac_om = (-2.0 * dt * (12.0 * r2 * t2 - 3.0)) / (3.0 * om) - fac / om;
      } else {
        const double r = sqrt(r2), t = tan(0.5 * om);
        const double a = 2.0 * r * t, num = atan(a);
        fac = num / (r * om);
        const double da = 1.0 / (1.0 + a * a);
        // d/dr: (da 2 t r - num) / (r^2 om); d r^2 = 2 r dr
        dfac_r2 = (da * 2.0 * t * r - num) / (r2 * om) / (2.0 * r);
        dfac_om = da * 2.0 * r * 0.5 * (1.0 + t * t) / (r * om) - fac / om;
      }
      px = fx * u * fac + p[2];
      py = fy * v * fac + p[3];
      Juv[0] = fx * (fac + 2.0 * u * u * dfac_r2);
      Juv[1] = fx * (2.0 * u * v * dfac_r2);
      Juv[2] = fy * (2.0 * u * v * dfac_r2);
      Juv[3] = fy * (fac + 2.0 * v * v * dfac_r2);
      Jp[0][0] = u * fac; Jp[1][1] = v * fac;
      Jp[0][2] = 1.0; Jp[1][3] = 1.0;
      Jp[0][4] = fx * u * dfac_om; Jp[1][4] = fy * v * dfac_om;
      break;
    }
    default:
      px = py = 0.0;
      Juv[0] = Juv[1] = Juv[2] = Juv[3] = 0.0;
      break;
  }
}

template <bool WIDE, int KP = 8>
__device__ __forceinline__ void distort_project(int model, const double* __restrict__ p, double u, double v,
                                                double& px, double& py, double (&Juv)[4], double (&Jp)[2][KP]) {
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    Jp[0][j] = 0.0;
    Jp[1][j] = 0.0;
  }
  const double r2 = u * u + v * v;
  if constexpr (WIDE) {
    if (model >= GSFM_CAMERA_OPENCV_FISHEYE) {
      distort_project_wide<KP>(model, p, u, v, r2, px, py, Juv, Jp);
      return;
    }
  }
  switch (model) {
    case GSFM_CAMERA_SIMPLE_PINHOLE: {
      const double f = p[0];
      px = f * u + p[1];
      py = f * v + p[2];
      Juv[0] = f; Juv[1] = 0.0; Juv[2] = 0.0; Juv[3] = f;
      Jp[0][0] = u; Jp[1][0] = v;
      Jp[0][1] = 1.0; Jp[1][2] = 1.0;
      break;
    }
    case GSFM_CAMERA_PINHOLE: {
      px = p[0] * u + p[2];
      py = p[1] * v + p[3];
      Juv[0] = p[0]; Juv[1] = 0.0; Juv[2] = 0.0; Juv[3] = p[1];
      Jp[0][0] = u; Jp[1][1] = v;
      Jp[0][2] = 1.0; Jp[1][3] = 1.0;
      break;
    }
    case GSFM_CAMERA_SIMPLE_RADIAL:
    case GSFM_CAMERA_RADIAL: {
      const double f = p[0], k1 = p[3];
      const double k2 = model == GSFM_CAMERA_RADIAL ? p[4] : 0.0;
      const double rad = k1 * r2 + k2 * r2 * r2;
      const double drad = k1 + 2.0 * k2 * r2;
      const double ud = u * (1.0 + rad), vd = v * (1.0 + rad);
      px = f * ud + p[1];
      py = f * vd + p[2];
      Juv[0] = f * (1.0 + rad + 2.0 * u * u * drad);
      Juv[1] = f * (2.0 * u * v * drad);
      Juv[2] = Juv[1];
      Juv[3] = f * (1.0 + rad + 2.0 * v * v * drad);
      Jp[0][0] = ud; Jp[1][0] = vd;
      Jp[0][1] = 1.0; Jp[1][2] = 1.0;
      Jp[0][3] = f * u * r2; Jp[1][3] = f * v * r2;
      if (model == GSFM_CAMERA_RADIAL) {
        Jp[0][4] = f * u * r2 * r2;
        Jp[1][4] = f * v * r2 * r2;
      }
      break;
    }
    default: {  // GSFM_CAMERA_OPENCV
      const double fx = p[0], fy = p[1], k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7];
      const double rad = k1 * r2 + k2 * r2 * r2;
      const double drad = k1 + 2.0 * k2 * r2;
      const double du = u * rad + 2.0 * p1 * u * v + p2 * (r2 + 2.0 * u * u);
      const double dv = v * rad + 2.0 * p2 * u * v + p1 * (r2 + 2.0 * v * v);
      px = fx * (u + du) + p[2];
      py = fy * (v + dv) + p[3];
      Juv[0] = fx * (1.0 + rad + 2.0 * u * u * drad + 2.0 * p1 * v + 6.0 * p2 * u);
      Juv[1] = fx * (2.0 * u * v * drad + 2.0 * p1 * u + 2.0 * p2 * v);
      Juv[2] = fy * (2.0 * u * v * drad + 2.0 * p2 * v + 2.0 * p1 * u);
      Juv[3] = fy * (1.0 + rad + 2.0 * v * v * drad + 2.0 * p2 * u + 6.0 * p1 * v);
      Jp[0][0] = u + du; Jp[1][1] = v + dv;
      Jp[0][2] = 1.0; Jp[1][3] = 1.0;
      Jp[0][4] = fx * u * r2; Jp[1][4] = fy * v * r2;
      Jp[0][5] = fx * u * r2 * r2; Jp[1][5] = fy * v * r2 * r2;
      Jp[0][6] = fx * 2.0 * u * v; Jp[1][6] = fy * (r2 + 2.0 * v * v);
      Jp[0][7] = fx * (r2 + 2.0 * u * u); Jp[1][7] = fy * 2.0 * u * v;
      break;
    }
  }
}

// x_cam = R X + t; pixel = ImgFromCam(params, x_cam).  R9 row-major.
template <bool WIDE, int KP>
__device__ __forceinline__ void obs_geom(const double* __restrict__ R9, const double* __restrict__ t3, const V3& X,
                                         int model, const double* __restrict__ par, ObsGeomT<KP>& g) {
  g.a = V3{R9[0] * X.x + R9[1] * X.y + R9[2] * X.z, R9[3] * X.x + R9[4] * X.y + R9[5] * X.z,
           R9[6] * X.x + R9[7] * X.y + R9[8] * X.z};
  const double xc = g.a.x + t3[0], yc = g.a.y + t3[1], zc = g.a.z + t3[2];
  g.valid = zc > 2.220446049250313e-16;
  const double iz = 1.0 / (g.valid ? zc : 1.0);
  const double u = xc * iz, v = yc * iz;
  double Juv[4];
  distort_project<WIDE, KP>(model, par, u, v, g.px, g.py, Juv, g.Jp);
  // d(u,v)/d x_cam = [1/z, 0, -u/z; 0, 1/z, -v/z]
  g.Jx[0][0] = Juv[0] * iz;
  g.Jx[0][1] = Juv[1] * iz;
  g.Jx[0][2] = -(Juv[0] * u + Juv[1] * v) * iz;
  g.Jx[1][0] = Juv[2] * iz;
  g.Jx[1][1] = Juv[3] * iz;
  g.Jx[1][2] = -(Juv[2] * u + Juv[3] * v) * iz;
}

__device__ __forceinline__ V3 cross(const V3& a, const V3& b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Jx w  (2-vector) and Jx^T g (3-vector)
__device__ __forceinline__ void jx_mul(const double (&Jx)[2][3], const V3& w, double& o0, double& o1) {
  o0 = Jx[0][0] * w.x + Jx[0][1] * w.y + Jx[0][2] * w.z;
  o1 = Jx[1][0] * w.x + Jx[1][1] * w.y + Jx[1][2] * w.z;
}
__device__ __forceinline__ V3 jxT_mul(const double (&Jx)[2][3], double g0, double g1) {
  return V3{Jx[0][0] * g0 + Jx[1][0] * g1, Jx[0][1] * g0 + Jx[1][1] * g1, Jx[0][2] * g0 + Jx[1][2] * g1};
}
__device__ __forceinline__ V3 R_mul(const double* __restrict__ R9, const V3& v) {
  return V3{R9[0] * v.x + R9[1] * v.y + R9[2] * v.z, R9[3] * v.x + R9[4] * v.y + R9[5] * v.z,
            R9[6] * v.x + R9[7] * v.y + R9[8] * v.z};
}
__device__ __forceinline__ V3 RT_mul(const double* __restrict__ R9, const V3& v) {
  return V3{R9[0] * v.x + R9[3] * v.y + R9[6] * v.z, R9[1] * v.x + R9[4] * v.y + R9[7] * v.z,
            R9[2] * v.x + R9[5] * v.y + R9[8] * v.z};
}

}  // namespace gsfm
