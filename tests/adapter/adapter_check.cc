// Drives the three reference-named estimator classes of include/gsfm_glomap_adapter.hpp end to end
// from C++ (the reference's language) through the C ABI of libgsfm.so, on a small synthetic scene
// held in GLOMAP-shaped containers (tests/adapter/mock).  Prints "ADAPTER OK ..." on success.
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <random>

#include "gsfm_glomap_adapter.hpp"

using namespace glomap;

static void rot_y(double a, double R[9]) {
  const double c = std::cos(a), s = std::sin(a);
  const double M[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  for (int i = 0; i < 9; ++i) R[i] = M[i];
}
static mock_eigen::Quaterniond quat_of(const double R[9]) {  // rotation about y only
  const double ang = std::atan2(R[2], R[0]);
  return mock_eigen::Quaterniond(std::cos(0.5 * ang), 0.0, std::sin(0.5 * ang), 0.0);
}

int main() {
  const int N = 16, P = 400;
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::unordered_map<rig_t, Rig> rigs;
  std::unordered_map<camera_t, Camera> cameras;
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  std::unordered_map<track_t, Track> tracks;
  ViewGraph vg;
  Camera cam;
  cam.model_id = 1;  // PINHOLE
  cam.params = {800.0, 800.0, 320.0, 240.0};
  cameras[1] = cam;
  // cameras on a ring of radius 10 looking at the origin
  std::vector<double> Rg(9 * N), tg(3 * N), cg(3 * N);
  for (int n = 0; n < N; ++n) {
    const double th = 2.0 * M_PI * n / N;
    const double c[3] = {10.0 * std::sin(th), 0.0, -10.0 * std::cos(th)};
    double R[9];
    rot_y(th, R);  // world -> camera with R c = (0, 0, -10): the optical axis passes through the origin
    for (int i = 0; i < 9; ++i) Rg[9 * n + i] = R[i];
    for (int i = 0; i < 3; ++i) {
      cg[3 * n + i] = c[i];
      tg[3 * n + i] = -(R[3 * i] * c[0] + R[3 * i + 1] * c[1] + R[3 * i + 2] * c[2]);
    }
    Frame fr;
    fr.is_registered = true;
    Rigid3d pose;
    pose.rotation = quat_of(R);
    pose.translation = mock_eigen::Vector3d(tg[3 * n], tg[3 * n + 1], tg[3 * n + 2]);
    fr.SetRigFromWorld(pose);
    frames[n] = fr;
  }
  for (int n = 0; n < N; ++n) {
    Image im;
    im.image_id = n;
    im.camera_id = 1;
    im.frame_id = n;
    images[n] = im;
  }
  for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames[n];
  // points in a ball of radius 2, observed by every camera
  for (int p = 0; p < P; ++p) {
    Track tr;
    tr.track_id = p;
    const double X[3] = {2.0 * U(rng), 2.0 * U(rng), 2.0 * U(rng)};
    tr.xyz = mock_eigen::Vector3d(X[0] + 0.05 * U(rng), X[1] + 0.05 * U(rng), X[2] + 0.05 * U(rng));
    for (int n = 0; n < N; n += 1 + (p % 3)) {
      const double* R = &Rg[9 * n];
      double xc[3];
      for (int i = 0; i < 3; ++i) xc[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + tg[3 * n + i];
      const double nrm = std::sqrt(xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2]);
      images[n].features.emplace_back(800.0 * xc[0] / xc[2] + 320.0, 800.0 * xc[1] / xc[2] + 240.0);
      images[n].features_undist.emplace_back(xc[0] / nrm, xc[1] / nrm, xc[2] / nrm);
      tr.observations.emplace_back(n, (feature_t)(images[n].features.size() - 1));
    }
    tracks[p] = tr;
  }
  // view graph: each camera linked to its 4 successors, exact relative rotations
  for (int i = 0; i < N; ++i)
    for (int d = 1; d <= 4; ++d) {
      const int j = (i + d) % N;
      ImagePair pr;
      pr.image_id1 = i;
      pr.image_id2 = j;
      double Rij[9];
      rot_y((2.0 * M_PI * j / N) - (2.0 * M_PI * i / N), Rij);  // R_j R_i^T
      pr.cam2_from_cam1.rotation = quat_of(Rij);
      pr.inliers.resize(100 + 7 * d);
      vg.image_pairs[(uint64_t)i * 1000 + j] = pr;
    }

  // 1) rotation averaging from a perturbed start
  auto frames_ra = frames;
  for (auto& [id, fr] : frames_ra) {
    Rigid3d p = fr.RigFromWorld();
    p.rotation = mock_eigen::Quaterniond(1, 0, 0, 0);
    fr.SetRigFromWorld(p);
  }
  for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames_ra[n];
  RotationEstimatorOptions ro;
  gsfm_glomap::RotationEstimator ra(ro);
  if (!ra.EstimateRotations(vg, rigs, frames_ra, images)) return std::printf("RA failed\n"), 1;
  double worst = 0;
  for (int n = 1; n < N; ++n) {  // relative rotation to frame 0 (gauge free): angle about y
    auto q0 = frames_ra[0].RigFromWorld().rotation, qn = frames_ra[n].RigFromWorld().rotation;
    const double a_est = 2.0 * (std::atan2(qn.y(), qn.w()) - std::atan2(q0.y(), q0.w()));
    const double a_ref = 2.0 * M_PI * n / N;
    double d = std::fmod(a_est - a_ref, 2.0 * M_PI);
    if (d > M_PI) d -= 2.0 * M_PI;
    if (d < -M_PI) d += 2.0 * M_PI;
    worst = std::fmax(worst, std::fabs(d));
  }
  if (worst > 1e-6) return std::printf("RA error %.3e rad\n", worst), 1;
  for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames[n];

  // 2) global positioning (random init inside) with the ground-truth rotations
  GlobalPositionerOptions go;
  gsfm_glomap::GlobalPositioner gp(go);
  auto frames_gp = frames;
  for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames_gp[n];
  auto tracks_gp = tracks;
  if (!gp.Solve(vg, rigs, cameras, frames_gp, images, tracks_gp)) return std::printf("GP failed\n"), 1;
  // centres up to similarity: compare ratios of pairwise distances
  auto center = [&](Frame& f, double* c) {
    double tmp[3];
    gsfm_glomap::detail::RotateInv(f.RigFromWorld().rotation, f.RigFromWorld().translation, tmp);
    for (int i = 0; i < 3; ++i) c[i] = -tmp[i];
  };
  double c0[3], c1[3], c8[3];
  center(frames_gp[0], c0);
  center(frames_gp[1], c1);
  center(frames_gp[8], c8);
  auto dist = [](const double* a, const double* b) {
    return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
  };
  const double ratio = dist(c0, c8) / dist(c0, c1);
  const double ratio_ref = 20.0 / (2.0 * 10.0 * std::sin(M_PI / N));
  if (std::fabs(ratio / ratio_ref - 1.0) > 1e-3) return std::printf("GP ratio %.6f vs %.6f\n", ratio, ratio_ref), 1;
  for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames[n];

  // 2b) the other constraint types (gp.cc:55-71, 167-210): camera-to-camera BATA pairs from the view graph's relative
  //     translations (unit length, exact), alone and next to the tracks
  {
    auto vg_t = vg;
    for (auto& [pid, pr] : vg_t.image_pairs) {
      const int i = (int)pr.image_id1, j = (int)pr.image_id2;
      double ci[3], cj[3], t[3];
      for (int a = 0; a < 3; ++a) {  // c = -R^T t
        ci[a] = -(Rg[9 * i + a] * tg[3 * i] + Rg[9 * i + 3 + a] * tg[3 * i + 1] + Rg[9 * i + 6 + a] * tg[3 * i + 2]);
        cj[a] = -(Rg[9 * j + a] * tg[3 * j] + Rg[9 * j + 3 + a] * tg[3 * j + 1] + Rg[9 * j + 6 + a] * tg[3 * j + 2]);
      }
      for (int a = 0; a < 3; ++a)  // t_21 = R_2 (c_1 - c_2)
        t[a] = Rg[9 * j + 3 * a] * (ci[0] - cj[0]) + Rg[9 * j + 3 * a + 1] * (ci[1] - cj[1]) + Rg[9 * j + 3 * a + 2] * (ci[2] - cj[2]);
      const double nrm = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
      pr.cam2_from_cam1.translation = mock_eigen::Vector3d(t[0] / nrm, t[1] / nrm, t[2] / nrm);
    }
    for (auto ctype : {GlobalPositionerOptions::ONLY_CAMERAS, GlobalPositionerOptions::POINTS_AND_CAMERAS_BALANCED,
                       GlobalPositionerOptions::POINTS_AND_CAMERAS}) {
      GlobalPositionerOptions gc;
      gc.constraint_type = ctype;
      gc.constraint_reweight_scale = 2.0;
      gsfm_glomap::GlobalPositioner gpc(gc);
      auto frames_c = frames;
      for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames_c[n];
      auto tracks_c = tracks;
      if (!gpc.Solve(vg_t, rigs, cameras, frames_c, images, tracks_c)) return std::printf("GP constraint type %d failed\n", (int)ctype), 1;
      double a0[3], a1[3], a8[3];
      center(frames_c[0], a0);
      center(frames_c[1], a1);
      center(frames_c[8], a8);
      const double rc = dist(a0, a8) / dist(a0, a1);
      if (std::fabs(rc / ratio_ref - 1.0) > 1e-3) return std::printf("GP constraint type %d: ratio %.6f vs %.6f\n", (int)ctype, rc, ratio_ref), 1;
      if (ctype == GlobalPositionerOptions::ONLY_CAMERAS)  // the tracks are not part of the problem: untouched
        for (auto& [tid, tr] : tracks_c)
          if (tr.xyz[0] != tracks[tid].xyz[0] || tr.is_initialized != tracks[tid].is_initialized) return std::printf("ONLY_CAMERAS touched a track\n"), 1;
      if (ctype == GlobalPositionerOptions::ONLY_CAMERAS) {
        // ... and they may be absent altogether (gp.cc:46-50 asks for tracks only when the type uses them): the packed track
        // arrays are then empty vectors, i.e. null pointers at the C ABI
        auto frames_e = frames;
        for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames_e[n];
        std::unordered_map<track_t, Track> no_tracks;
        if (!gsfm_glomap::GlobalPositioner(gc).Solve(vg_t, rigs, cameras, frames_e, images, no_tracks))
          return std::printf("ONLY_CAMERAS without tracks failed\n"), 1;
        double e0[3], e1[3], e8[3];
        center(frames_e[0], e0);
        center(frames_e[1], e1);
        center(frames_e[8], e8);
        const double re = dist(e0, e8) / dist(e0, e1);
        if (std::fabs(re / ratio_ref - 1.0) > 1e-3) return std::printf("ONLY_CAMERAS without tracks: ratio %.6f vs %.6f\n", re, ratio_ref), 1;
        for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames_c[n];
      }
      GlobalPositionerOptions none = gc;  // no image pairs at all: refused (gp.cc:41-45)
      ViewGraph empty;
      if (gsfm_glomap::GlobalPositioner(none).Solve(empty, rigs, cameras, frames_c, images, tracks_c)) return std::printf("GP accepted an empty view graph\n"), 1;
    }
    for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames[n];
  }

  // 3) bundle adjustment from perturbed points: must reach (numerically) zero reprojection error.
  //    A posed frame that no track observes is added LAST (libstdc++ iterates it FIRST): the reference's gauge is the first
  //    frame that owns a parameter block (bundle_adjustment.cc:253-269), so exactly one OBSERVED frame must stay bit-identical
  //    and the unobserved one must not be touched; a track with 3 raw observations of which one points to an image that is
  //    not in `images` is still optimised (raw-count rule, bundle_adjustment.cc:122-125).
  {
    Frame extra;
    Rigid3d pe;
    pe.translation = mock_eigen::Vector3d(1.0, 2.0, 3.0);
    extra.SetRigFromWorld(pe);
    frames[1000] = extra;
    for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames[n];
  }
  const track_t short_id = 7;  // stride 2 -> 8 observations: cut to 2 usable + 1 dangling
  {
    auto& tr = tracks[short_id];
    tr.observations.resize(2);
    tr.observations.emplace_back((image_t)4242, (feature_t)0);
  }
  const auto short_xyz_before = tracks[short_id].xyz;
  auto frames_before = frames;
  BundleAdjusterOptions bo;
  gsfm_glomap::BundleAdjuster ba(bo);
  if (!ba.Solve(rigs, cameras, frames, images, tracks)) return std::printf("BA failed\n"), 1;
  {
    int untouched = 0;
    for (int n = 0; n < N; ++n) {
      const auto &a = frames[n].RigFromWorld(), &b = frames_before[n].RigFromWorld();
      if (a.rotation.w() == b.rotation.w() && a.rotation.y() == b.rotation.y() && a.translation[0] == b.translation[0] &&
          a.translation[2] == b.translation[2])
        ++untouched;
    }
    if (untouched != 1) return std::printf("BA gauge: %d observed frames constant, expected 1\n", untouched), 1;
    if (frames[1000].RigFromWorld().translation[1] != 2.0) return std::printf("BA touched an unobserved frame\n"), 1;
    const auto& x = tracks[short_id].xyz;
    if (x[0] == short_xyz_before[0] && x[1] == short_xyz_before[1]) return std::printf("BA skipped the 3-raw-observation track\n"), 1;
    tracks[short_id].observations.pop_back();  // the checks below look every image up
  }
  double maxerr = 0;
  for (auto& [tid, tr] : tracks)
    for (auto& ob : tr.observations) {
      Frame& fr = frames[images[ob.first].frame_id];
      double xc[3];
      const double X[3] = {tr.xyz[0], tr.xyz[1], tr.xyz[2]};
      gsfm_glomap::detail::Rotate(fr.RigFromWorld().rotation, X, xc);
      for (int i = 0; i < 3; ++i) xc[i] += fr.RigFromWorld().translation[i];
      const auto& par = cameras[1].params;
      const double u = par[0] * xc[0] / xc[2] + par[2], v = par[1] * xc[1] / xc[2] + par[3];
      const auto& f = images[ob.first].features[ob.second];
      maxerr = std::fmax(maxerr, std::hypot(u - f[0], v - f[1]));
    }
  if (maxerr > 1e-3) return std::printf("BA reprojection error %.3e px\n", maxerr), 1;
  // 3b) a camera model with more than 8 parameters: FULL_OPENCV (12) with zero distortion projects like the PINHOLE above;
  //     the adapter packs 16-wide intrinsics rows (gsfm_ba_problem::intr_stride) and the library's 16-wide unit solves
  //     (colmap::CreateCameraCostFunction dispatches on any CameraModelId, bundle_adjustment.cc:136-139).
  {
    const auto cam_before = cameras[1];
    const auto tracks_before = tracks;
    const auto frames_b = frames;
    cameras[1].model_id = 6;
    cameras[1].params.resize(12, 0.0);
    for (auto& [tid, tr] : tracks)
      if (tr.observations.size() >= 3)  // (shorter tracks are not part of the problem, bundle_adjustment.cc:122)
        for (int i = 0; i < 3; ++i) tr.xyz[i] += 2e-3 * U(rng);
    gsfm_glomap::BundleAdjuster ba12(bo);
    if (!ba12.Solve(rigs, cameras, frames, images, tracks)) return std::printf("BA (FULL_OPENCV) failed\n"), 1;
    const auto& par = cameras[1].params;
    if (par.size() != 12) return std::printf("BA (FULL_OPENCV) resized the parameter block\n"), 1;
    if (par[2] != 320.0 || par[3] != 240.0) return std::printf("BA (FULL_OPENCV) moved the principal point\n"), 1;
    double e12 = 0;
    for (auto& [tid, tr] : tracks)
      for (auto& ob : tr.observations) {
        Frame& fr = frames[images[ob.first].frame_id];
        double xc[3];
        const double X[3] = {tr.xyz[0], tr.xyz[1], tr.xyz[2]};
        gsfm_glomap::detail::Rotate(fr.RigFromWorld().rotation, X, xc);
        for (int i = 0; i < 3; ++i) xc[i] += fr.RigFromWorld().translation[i];
        const double u = xc[0] / xc[2], v = xc[1] / xc[2], r2 = u * u + v * v;
        const double rad = (1 + par[4] * r2 + par[5] * r2 * r2 + par[8] * r2 * r2 * r2) / (1 + par[9] * r2 + par[10] * r2 * r2 + par[11] * r2 * r2 * r2);
        const double ud = u * rad + 2 * par[6] * u * v + par[7] * (r2 + 2 * u * u), vd = v * rad + 2 * par[7] * u * v + par[6] * (r2 + 2 * v * v);
        const auto& f = images[ob.first].features[ob.second];
        e12 = std::fmax(e12, std::hypot(par[0] * ud + par[2] - f[0], par[1] * vd + par[3] - f[1]));
      }
    // the perturbation puts in 0.15 px; with zero distortion the numerator / denominator coefficients k1..k3 / k4..k6 are exactly
    // collinear, the cost is flat along them and the default function_tolerance ends the solve at a few 1e-3 px (measured
    // 4.1e-3; the solver's parity on the wide models is tests/test_ba_wide_models_gpu.py — this scenario is about the packing)
    if (e12 > 2e-2) return std::printf("BA (FULL_OPENCV) reprojection error %.3e px\n", e12), 1;
    std::printf("BA with a 12-parameter camera model: reprojection error %.2e px, %d LM iterations\n", e12, ba12.LastReport().iterations);
    cameras[1] = cam_before;  // the scenarios below continue on the PINHOLE scene
    tracks = tracks_before;
    frames = frames_b;
    for (int n = 0; n < N; ++n) images[n].frame_ptr = &frames[n];
  }
  // 3c) UndistortImages (image_undistorter.cc:7-46) on the adapter: the bearings of the PINHOLE scene are recomputed from its
  //     pixels (clean_points), then the same rays are pushed through a SIMPLE_RADIAL camera with k = 0.05 and recovered; with
  //     clean_points = false an image that already has its bearings is left alone.
  {
    auto imgs = images;
    const auto cams_before = cameras;
    cameras[1].params = {800.0, 800.0, 320.0, 240.0};  // the intrinsics the scene's pixels and bearings were made with (BA moved them)
    for (auto& [iid, im] : imgs) im.features_undist.clear();
    gsfm_glomap::UndistortImages(cameras, imgs, true);
    double e = 0;
    for (auto& [iid, im] : imgs) {
      if (im.features_undist.size() != im.features.size()) return std::printf("UndistortImages left image %d without bearings\n", (int)iid), 1;
      for (size_t i = 0; i < im.features.size(); ++i)
        for (int j = 0; j < 3; ++j) e = std::fmax(e, std::fabs(im.features_undist[i][j] - images[iid].features_undist[i][j]));
    }
    if (e > 1e-12) return std::printf("UndistortImages (PINHOLE): %.3e\n", e), 1;
    cameras[1].model_id = 2;  // SIMPLE_RADIAL f, cx, cy, k
    cameras[1].params = {800.0, 320.0, 240.0, 0.05};
    for (auto& [iid, im] : imgs)
      for (size_t i = 0; i < im.features.size(); ++i) {
        const auto& r = images[iid].features_undist[i];
        const double u = r[0] / r[2], v = r[1] / r[2], d = 1.0 + 0.05 * (u * u + v * v);
        im.features[i] = mock_eigen::Vector2d(800.0 * u * d + 320.0, 800.0 * v * d + 240.0);
      }
    imgs[0].features_undist[0] = mock_eigen::Vector3d(0.0, 0.6, 0.8);  // a stale bearing
    gsfm_glomap::UndistortImages(cameras, imgs, false);  // nothing to do: every image has as many bearings as features
    if (imgs[0].features_undist[0][1] != 0.6) return std::printf("UndistortImages(clean_points = false) recomputed a complete image\n"), 1;
    gsfm_glomap::UndistortImages(cameras, imgs, true);
    double e2 = 0;
    for (auto& [iid, im] : imgs)
      for (size_t i = 0; i < im.features.size(); ++i)
        for (int j = 0; j < 3; ++j) e2 = std::fmax(e2, std::fabs(im.features_undist[i][j] - images[iid].features_undist[i][j]));
    if (e2 > 1e-10) return std::printf("UndistortImages (SIMPLE_RADIAL): %.3e\n", e2), 1;
    std::printf("UndistortImages: PINHOLE %.1e, SIMPLE_RADIAL (k = 0.05) %.1e from the true bearings\n", e, e2);
    cameras = cams_before;
  }
  // 4) processors: a corrupted observation is filtered, a far point has no triangulation angle,
  //    a wrong relative rotation is invalidated, normalisation scales the ring of centres to extent 10
  {
    auto& im0 = images[0];
    im0.features_undist[0] = mock_eigen::Vector3d(0.6, 0.0, 0.8);  // first observation of some track in image 0
    const int changed = gsfm_glomap::TrackFilter::FilterTracksByAngle(vg, cameras, images, tracks, 1.0);
    if (changed != 1) return std::printf("FilterTracksByAngle changed %d tracks\n", changed), 1;
    const int changed2 = gsfm_glomap::TrackFilter::FilterTracksByReprojection(vg, cameras, images, tracks, 1e-2, true);
    if (changed2 != 0) return std::printf("FilterTracksByReprojection changed %d tracks\n", changed2), 1;
    tracks[5].xyz = mock_eigen::Vector3d(1e7, 2e7, 3e7);
    const int removed = gsfm_glomap::TrackFilter::FilterTrackTriangulationAngle(vg, images, tracks, 1.0);
    if (removed != 1 || !tracks[5].observations.empty()) return std::printf("triangulation filter removed %d\n", removed), 1;
    auto& bad = vg.image_pairs.begin()->second;
    bad.cam2_from_cam1.rotation = mock_eigen::Quaterniond(0.0, 1.0, 0.0, 0.0);  // 180 degrees about x
    gsfm_glomap::RelPoseFilter::FilterRotations(vg, images, 5.0);
    int ninvalid = 0;
    for (auto& [id, pr] : vg.image_pairs) ninvalid += pr.is_valid ? 0 : 1;
    if (ninvalid != 1 || bad.is_valid) return std::printf("FilterRotations invalidated %d pairs\n", ninvalid), 1;
    double c0b[3], c8b[3], c0n[3], c8n[3];
    center(frames[0], c0b);
    center(frames[8], c8b);
    const double diam_before = dist(c0b, c8b);
    const auto sim = gsfm_glomap::NormalizeReconstruction(rigs, cameras, frames, images, tracks);
    center(frames[0], c0n);
    center(frames[8], c8n);
    // every distance between camera centres scales by the returned factor
    const double diam = dist(c0n, c8n);
    if (!(sim[0] > 0.0) || std::fabs(diam / (diam_before * sim[0]) - 1.0) > 1e-9)
      return std::printf("normalizer scale %.9f diam %.9f (before %.9f)\n", sim[0], diam, diam_before), 1;
  }
  // 5) producers of the GP / BA inputs: matches derived from the scene's tracks -> EstablishFullTracks must
  //    give the tracks back (as sets of (image, feature)), FindTracksForProblem keeps those seen by >= 3 images,
  //    KeepLargestConnectedComponents drops an isolated pair of frames
  {
    for (auto& [pid, pr] : vg.image_pairs) {
      pr.is_valid = true;
      pr.inliers.clear();
      pr.matches = mock_eigen::MatrixXi();
    }
    size_t expect_tracks = 0, expect_selected = 0;
    for (auto& [tid, tr] : tracks) {
      if (tr.observations.empty()) continue;
      bool linked = false;
      for (size_t a = 0; a < tr.observations.size(); ++a)
        for (size_t b = a + 1; b < tr.observations.size(); ++b) {
          const int i = (int)tr.observations[a].first, j = (int)tr.observations[b].first;
          const int d = (j - i + N) % N;
          ImagePair* pr = nullptr;
          bool fwd = true;
          if (d >= 1 && d <= 4) pr = &vg.image_pairs[(uint64_t)i * 1000 + j];
          else if (N - d >= 1 && N - d <= 4) pr = &vg.image_pairs[(uint64_t)j * 1000 + i], fwd = false;
          if (!pr) continue;
          pr->matches.push_row(-1, -1);  // an outlier row that `inliers` skips
          if (fwd) pr->matches.push_row((int)tr.observations[a].second, (int)tr.observations[b].second);
          else pr->matches.push_row((int)tr.observations[b].second, (int)tr.observations[a].second);
          pr->inliers.push_back(pr->matches.rows() - 1);
          linked = true;
        }
      // stride-1 and stride-2 tracks are chains of neighbours (connected); stride-3 tracks link 0-3-6-...-15 as well
      if (linked) {
        ++expect_tracks;
        if (tr.observations.size() >= 3) ++expect_selected;
      }
    }
    TrackEstablishmentOptions to;
    gsfm_glomap::TrackEngine engine(vg, images, to);
    std::unordered_map<track_t, Track> full, selected;
    const size_t nfull = engine.EstablishFullTracks(full);
    if (nfull != expect_tracks) return std::printf("EstablishFullTracks: %zu tracks, expected %zu\n", nfull, expect_tracks), 1;
    // every established track equals one scene track: same size, and its id is its smallest member
    std::unordered_map<uint64_t, size_t> size_of;
    for (auto& [tid, tr] : tracks)
      for (auto& ob : tr.observations) size_of[((uint64_t)ob.first << 32) | ob.second] = tr.observations.size();
    for (auto& [id, tr] : full) {
      if (tr.observations.empty()) return std::printf("unexpected discarded track\n"), 1;
      const auto& o0 = tr.observations.front();
      if ((((uint64_t)o0.first << 32) | o0.second) != id) return std::printf("track id is not its smallest member\n"), 1;
      for (auto& ob : tr.observations)
        if (size_of[((uint64_t)ob.first << 32) | ob.second] != tr.observations.size()) return std::printf("track members differ\n"), 1;
    }
    const size_t nsel = engine.FindTracksForProblem(full, selected);
    if (nsel != expect_selected) return std::printf("FindTracksForProblem: %zu tracks, expected %zu\n", nsel, expect_selected), 1;
    for (auto& [id, tr] : selected)
      if (tr.track_id != id || tr.observations.size() < 3) return std::printf("bad selected track\n"), 1;
    // two extra frames linked only to each other
    for (int n = N; n < N + 2; ++n) {
      frames[n] = Frame();
      Image im;
      im.image_id = n;
      im.frame_id = n;
      images[n] = im;
    }
    for (int n = 0; n < N + 2; ++n) images[n].frame_ptr = &frames[n];
    ImagePair lone;
    lone.image_id1 = N;
    lone.image_id2 = N + 1;
    vg.image_pairs[(uint64_t)N * 1000 + N + 1] = lone;
    const int kept = gsfm_glomap::KeepLargestConnectedComponents(vg, frames, images);
    if (kept != N || frames[N].is_registered || frames[N + 1].is_registered || !frames[3].is_registered ||
        vg.image_pairs[(uint64_t)N * 1000 + N + 1].is_valid)
      return std::printf("KeepLargestConnectedComponents kept %d images\n", kept), 1;
  }
  // 6) calibrated rigs (the shape of global_mapper_test.cc:89-126): 12 frames of a 2-camera rig — reference sensor
  //    (camera 11) and a second sensor (camera 12) with a KNOWN cam_from_rig (10 degrees about y, a metric baseline).
  //    RotationEstimator, GlobalPositioner and BundleAdjuster must handle them through the same three calls.
  double rig_ra = 0, rig_gp = 0, rig_ba = 0, rig_sens = 0, rig_unk = 0, rig_unk_rot = 0;
  {
    const int NF = 12, NP = 300;
    std::unordered_map<rig_t, Rig> rigs2;
    std::unordered_map<camera_t, Camera> cams2;
    std::unordered_map<frame_t, Frame> fr2;
    std::unordered_map<image_t, Image> im2;
    std::unordered_map<track_t, Track> tr2;
    ViewGraph vg2;
    Camera c2;
    c2.model_id = 1;
    c2.params = {800.0, 800.0, 320.0, 240.0};
    cams2[11] = c2;
    cams2[12] = c2;
    const double s_ang = 10.0 * M_PI / 180.0;
    double Rs[9];
    rot_y(s_ang, Rs);
    const double ts[3] = {0.8, 0.1, -0.2};
    {
      Rig rig;
      rig.SetRigId(1);
      rig.AddRefSensor(sensor_t(SensorType::CAMERA, 11));
      Rigid3d cfr;
      cfr.rotation = quat_of(Rs);
      cfr.translation = mock_eigen::Vector3d(ts[0], ts[1], ts[2]);
      rig.AddSensor(sensor_t(SensorType::CAMERA, 12), cfr);
      rigs2[1] = rig;
    }
    std::vector<double> Rf(9 * NF), tf(3 * NF), cf(3 * NF), Rc(9 * 2 * NF), tc(3 * 2 * NF), ang_c(2 * NF);
    for (int f = 0; f < NF; ++f) {
      const double th = 2.0 * M_PI * f / NF;
      rot_y(th, &Rf[9 * f]);
      const double c[3] = {10.0 * std::sin(th), 0.0, -10.0 * std::cos(th)};
      for (int i = 0; i < 3; ++i) {
        cf[3 * f + i] = c[i];
        tf[3 * f + i] = -(Rf[9 * f + 3 * i] * c[0] + Rf[9 * f + 3 * i + 1] * c[1] + Rf[9 * f + 3 * i + 2] * c[2]);
      }
      Frame fr;
      fr.is_registered = true;
      fr.SetRigId(1);
      Rigid3d pose;
      pose.rotation = quat_of(&Rf[9 * f]);
      pose.translation = mock_eigen::Vector3d(tf[3 * f], tf[3 * f + 1], tf[3 * f + 2]);
      fr.SetRigFromWorld(pose);
      fr2[f] = fr;
      for (int sn = 0; sn < 2; ++sn) {
        const int id = 2 * f + sn;
        Image im;
        im.image_id = id;
        im.camera_id = 11 + sn;
        im.frame_id = f;
        im2[id] = im;
        // cam_from_world = cam_from_rig * rig_from_world
        ang_c[id] = th + (sn ? s_ang : 0.0);
        rot_y(ang_c[id], &Rc[9 * id]);
        for (int i = 0; i < 3; ++i) {
          double v = tf[3 * f + i];
          if (sn) v = Rs[3 * i] * tf[3 * f] + Rs[3 * i + 1] * tf[3 * f + 1] + Rs[3 * i + 2] * tf[3 * f + 2] + ts[i];
          tc[3 * id + i] = v;
        }
      }
    }
    for (int f = 0; f < NF; ++f) {
      fr2[f].SetRigPtr(&rigs2[1]);
      for (int sn = 0; sn < 2; ++sn) {
        im2[2 * f + sn].frame_ptr = &fr2[f];
        fr2[f].AddDataId(data_t(sensor_t(SensorType::CAMERA, 11 + sn), 2 * f + sn));
      }
    }
    std::vector<double> Xg(3 * NP);
    for (int p = 0; p < NP; ++p) {
      Track tr;
      tr.track_id = p;
      for (int i = 0; i < 3; ++i) Xg[3 * p + i] = 2.0 * U(rng);
      tr.xyz = mock_eigen::Vector3d(Xg[3 * p] + 0.05 * U(rng), Xg[3 * p + 1] + 0.05 * U(rng), Xg[3 * p + 2] + 0.05 * U(rng));
      for (int id = 0; id < 2 * NF; id += 1 + (p % 3)) {
        const double* R = &Rc[9 * id];
        double xc[3];
        for (int i = 0; i < 3; ++i) xc[i] = R[3 * i] * Xg[3 * p] + R[3 * i + 1] * Xg[3 * p + 1] + R[3 * i + 2] * Xg[3 * p + 2] + tc[3 * id + i];
        const double nrm = std::sqrt(xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2]);
        im2[id].features.emplace_back(800.0 * xc[0] / xc[2] + 320.0, 800.0 * xc[1] / xc[2] + 240.0);
        im2[id].features_undist.emplace_back(xc[0] / nrm, xc[1] / nrm, xc[2] / nrm);
        tr.observations.emplace_back(id, (feature_t)(im2[id].features.size() - 1));
      }
      tr2[p] = tr;
    }
    for (int a = 0; a < 2 * NF; ++a)
      for (int b = a + 1; b < 2 * NF; ++b) {
        const int d = std::abs(a / 2 - b / 2), dr = std::min(d, NF - d);
        if (dr > 2) continue;
        ImagePair pr;
        pr.image_id1 = a;
        pr.image_id2 = b;
        double Rab[9];
        rot_y(ang_c[b] - ang_c[a], Rab);
        pr.cam2_from_cam1.rotation = quat_of(Rab);
        pr.inliers.resize(50 + (a * 7 + b * 3) % 40);
        vg2.image_pairs[(uint64_t)a * 1000 + b] = pr;
      }
    // RA from identity rotations
    auto fr_ra = fr2;
    for (auto& [id, fr] : fr_ra) {
      Rigid3d p0;
      fr.SetRigFromWorld(p0);
    }
    for (int id = 0; id < 2 * NF; ++id) im2[id].frame_ptr = &fr_ra[id / 2];
    RotationEstimatorOptions ro2;
    gsfm_glomap::RotationEstimator ra2(ro2);
    if (!ra2.EstimateRotations(vg2, rigs2, fr_ra, im2)) return std::printf("rig RA failed\n"), 1;
    for (int f = 1; f < NF; ++f) {
      auto q0 = fr_ra[0].RigFromWorld().rotation, qn = fr_ra[f].RigFromWorld().rotation;
      double dd = std::fmod(2.0 * (std::atan2(qn.y(), qn.w()) - std::atan2(q0.y(), q0.w())) - 2.0 * M_PI * f / NF, 2.0 * M_PI);
      if (dd > M_PI) dd -= 2.0 * M_PI;
      if (dd < -M_PI) dd += 2.0 * M_PI;
      rig_ra = std::fmax(rig_ra, std::fabs(dd));
    }
    if (rig_ra > 1e-6) return std::printf("rig RA error %.3e rad\n", rig_ra), 1;
    // GP with the ground-truth rotations: the metric rig baseline fixes the scale, so distances come out ABSOLUTE
    auto fr_gp = fr2;
    for (int id = 0; id < 2 * NF; ++id) im2[id].frame_ptr = &fr_gp[id / 2];
    auto tr_gp = tr2;
    GlobalPositionerOptions go2;
    gsfm_glomap::GlobalPositioner gp2(go2);
    if (!gp2.Solve(vg2, rigs2, cams2, fr_gp, im2, tr_gp)) return std::printf("rig GP failed\n"), 1;
    double a0[3], a6[3];
    center(fr_gp[0], a0);
    center(fr_gp[6], a6);
    rig_gp = std::fabs(dist(a0, a6) / 20.0 - 1.0);
    if (rig_gp > 3e-2) return std::printf("rig GP: |c0 - c6| = %.6f, expected 20 (metric, to the accuracy of the stopping rule)\n", dist(a0, a6)), 1;
    // BA from perturbed points
    for (int id = 0; id < 2 * NF; ++id) im2[id].frame_ptr = &fr2[id / 2];
    BundleAdjusterOptions bo2;
    gsfm_glomap::BundleAdjuster ba2(bo2);
    if (!ba2.Solve(rigs2, cams2, fr2, im2, tr2)) return std::printf("rig BA failed\n"), 1;
    for (auto& [tid, tr] : tr2)
      for (auto& ob : tr.observations) {
        const auto cw = im2[ob.first].CamFromWorld();
        double xc[3];
        const double X[3] = {tr.xyz[0], tr.xyz[1], tr.xyz[2]};
        gsfm_glomap::detail::Rotate(cw.rotation, X, xc);
        for (int i = 0; i < 3; ++i) xc[i] += cw.translation[i];
        const auto& par = cams2[im2[ob.first].camera_id].params;
        const auto& f = im2[ob.first].features[ob.second];
        rig_ba = std::fmax(rig_ba, std::hypot(par[0] * xc[0] / xc[2] + par[2] - f[0], par[1] * xc[1] / xc[2] + par[3] - f[1]));
      }
    if (rig_ba > 1e-3) return std::printf("rig BA reprojection error %.3e px\n", rig_ba), 1;
    // optimize_rig_poses (RigReprojErrorCostFunctor, ba.cc:161-179): start from a miscalibrated cam_from_rig (0.5 degrees,
    // 5 cm) and let BA refine it; its rotation is observable and must come back, the reprojection error must vanish
    {
      double Rp[9];
      rot_y(s_ang + 0.5 * M_PI / 180.0, Rp);
      Rigid3d bad;
      bad.rotation = quat_of(Rp);
      bad.translation = mock_eigen::Vector3d(ts[0] + 0.05, ts[1] - 0.03, ts[2] + 0.02);
      rigs2[1].SetSensorFromRig(sensor_t(SensorType::CAMERA, 12), bad);
      BundleAdjusterOptions bo3;
      bo3.optimize_rig_poses = true;
      gsfm_glomap::BundleAdjuster ba3(bo3);
      if (!ba3.Solve(rigs2, cams2, fr2, im2, tr2)) return std::printf("rig BA (optimize_rig_poses) failed\n"), 1;
      const auto qs = rigs2[1].SensorFromRig(sensor_t(SensorType::CAMERA, 12)).rotation;
      rig_sens = std::fabs(2.0 * std::atan2(qs.y(), qs.w()) - s_ang);
      if (rig_sens > 1e-6 || std::fabs(qs.x()) > 1e-7 || std::fabs(qs.z()) > 1e-7)
        return std::printf("optimize_rig_poses: cam_from_rig rotation off by %.3e rad\n", rig_sens), 1;
      double worst_px = 0.0;
      for (auto& [tid, tr] : tr2)
        for (auto& ob : tr.observations) {
          const auto cw = im2[ob.first].CamFromWorld();
          double xc[3];
          const double X[3] = {tr.xyz[0], tr.xyz[1], tr.xyz[2]};
          gsfm_glomap::detail::Rotate(cw.rotation, X, xc);
          for (int i = 0; i < 3; ++i) xc[i] += cw.translation[i];
          const auto& par = cams2[im2[ob.first].camera_id].params;
          const auto& f = im2[ob.first].features[ob.second];
          worst_px = std::fmax(worst_px, std::hypot(par[0] * xc[0] / xc[2] + par[2] - f[0], par[1] * xc[1] / xc[2] + par[3] - f[1]));
        }
      if (worst_px > 1e-3) return std::printf("optimize_rig_poses: reprojection error %.3e px\n", worst_px), 1;
    }
    // unknown cam_from_rig translation (NaN, what rotation averaging leaves behind for an estimated sensor):
    // GlobalPositioner estimates it (RigUnknownBATAPairwiseDirectionError, gp.cc:354-368) in the scale of the solution
    {
      // first the rotation: a sensor without any cam_from_rig (rotation_averager_test.cc:214-263) gets a cam block in
      // RotationEstimator (gra.cc:173-191); noise-free data: exact after the spanning-tree start, translation left NaN
      rigs2[1].ResetSensorFromRig(sensor_t(SensorType::CAMERA, 12));
      auto fr_u = fr2;
      for (auto& [id, fr] : fr_u) {
        Rigid3d p0;
        fr.SetRigFromWorld(p0);
      }
      for (int id = 0; id < 2 * NF; ++id) im2[id].frame_ptr = &fr_u[id / 2];
      gsfm_glomap::RotationEstimator ra3(ro2);
      if (!ra3.EstimateRotations(vg2, rigs2, fr_u, im2)) return std::printf("RA with an unknown cam_from_rig failed\n"), 1;
      const auto est = rigs2[1].MaybeSensorFromRig(sensor_t(SensorType::CAMERA, 12));
      if (!est.has_value() || !std::isnan(est->translation[0])) return std::printf("RA: cam_from_rig not written as (rotation, NaN)\n"), 1;
      rig_unk_rot = std::fabs(2.0 * std::atan2(est->rotation.y(), est->rotation.w()) - s_ang);
      if (rig_unk_rot > 1e-6 || std::fabs(est->rotation.x()) > 1e-7 || std::fabs(est->rotation.z()) > 1e-7)
        return std::printf("RA: estimated cam_from_rig rotation off by %.3e rad\n", rig_unk_rot), 1;
      for (int f = 1; f < NF; ++f) {
        auto q0 = fr_u[0].RigFromWorld().rotation, qn = fr_u[f].RigFromWorld().rotation;
        double dd = std::fmod(2.0 * (std::atan2(qn.y(), qn.w()) - std::atan2(q0.y(), q0.w())) - 2.0 * M_PI * f / NF, 2.0 * M_PI);
        if (dd > M_PI) dd -= 2.0 * M_PI;
        if (dd < -M_PI) dd += 2.0 * M_PI;
        if (std::fabs(dd) > 1e-6) return std::printf("RA with an unknown cam_from_rig: frame %d off by %.3e rad\n", f, dd), 1;
      }
      // then the translation, with the rotations just estimated
      auto tr_u = tr2;
      gsfm_glomap::GlobalPositioner gp3(go2);
      if (!gp3.Solve(vg2, rigs2, cams2, fr_u, im2, tr_u)) return std::printf("GP with an unknown cam_from_rig failed\n"), 1;
      const auto te = rigs2[1].SensorFromRig(sensor_t(SensorType::CAMERA, 12)).translation;
      double b0[3], b6[3];
      center(fr_u[0], b0);
      center(fr_u[6], b6);
      const double sc = 20.0 / dist(b0, b6);  // the solution's free scale
      for (int i = 0; i < 3; ++i) rig_unk = std::fmax(rig_unk, std::fabs(sc * te[i] - ts[i]));
      if (!(rig_unk < 5e-2)) return std::printf("unknown cam_from_rig: translation off by %.3e (scaled)\n", rig_unk), 1;
      for (int id = 0; id < 2 * NF; ++id) im2[id].frame_ptr = &fr2[id / 2];
    }
    // a sensor without any cam_from_rig is refused, not mis-solved
    rigs2[1].ResetSensorFromRig(sensor_t(SensorType::CAMERA, 12));
    if (ba2.Solve(rigs2, cams2, fr2, im2, tr2)) return std::printf("BA accepted an uncalibrated rig\n"), 1;
  }
  // 7) gravity-aligned rotation averaging (use_gravity, rotation_averager_test.cc:171-212): 10 trivial frames with general
  //    rotations, every frame carries its gravity (the world's y axis seen from the rig) and starts at R_align times a
  //    slightly wrong angle like PrepareGravity leaves them (:58-60); the 1-DoF solve must bring the exact rotations back.
  double grav_err = 0;
  {
    const int NG = 10;
    std::unordered_map<rig_t, Rig> rigs3;
    std::unordered_map<frame_t, Frame> fr3;
    std::unordered_map<image_t, Image> im3;
    ViewGraph vg3;
    auto qmul = [](const mock_eigen::Quaterniond& a, const mock_eigen::Quaterniond& b) {
      return mock_eigen::Quaterniond(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                                     a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                                     a.w() * b.y() - a.x() * b.z() + a.y() * b.w() + a.z() * b.x(),
                                     a.w() * b.z() + a.x() * b.y() - a.y() * b.x() + a.z() * b.w());
    };
    auto qinv = [](const mock_eigen::Quaterniond& a) { return mock_eigen::Quaterniond(a.w(), -a.x(), -a.y(), -a.z()); };
    std::vector<mock_eigen::Quaterniond> qgt(NG);
    for (int f = 0; f < NG; ++f) {
      // R_f = RotX(tilt) * RotZ(roll) * RotY(heading): a tilted, rolled camera
      const double hd = 0.55 * f, tilt = 0.3 * std::sin(1.3 * f), roll = 0.2 * std::cos(0.7 * f);
      const mock_eigen::Quaterniond qy(std::cos(0.5 * hd), 0, std::sin(0.5 * hd), 0), qx(std::cos(0.5 * tilt), std::sin(0.5 * tilt), 0, 0),
          qz(std::cos(0.5 * roll), 0, 0, std::sin(0.5 * roll));
      qgt[f] = qmul(qx, qmul(qz, qy));
      Frame fr;
      fr.is_registered = true;
      const double ey[3] = {0.0, 1.0, 0.0};
      double g[3];
      gsfm_glomap::detail::Rotate(qgt[f], ey, g);  // gravity in the rig = R_f * (0, 1, 0)
      fr.gravity_info.SetGravity(mock_eigen::Vector3d(g[0], g[1], g[2]));
      // start: R_align * RotY(true angle + offset); R_align^T R_f is a pure rotation about y by construction
      double qa[4];
      gsfm_glomap::detail::MatToQuatWxyz(fr.gravity_info.GetRAlign(), qa);
      const mock_eigen::Quaterniond qal(qa[0], qa[1], qa[2], qa[3]);
      const mock_eigen::Quaterniond qrel = qmul(qinv(qal), qgt[f]);
      const double ang = 2.0 * std::atan2(qrel.y(), qrel.w()) + (f == 0 ? 0.0 : 0.15 * std::sin(2.1 * f));
      Rigid3d pose;
      pose.rotation = qmul(qal, mock_eigen::Quaterniond(std::cos(0.5 * ang), 0, std::sin(0.5 * ang), 0));
      fr.SetRigFromWorld(pose);
      fr3[f] = fr;
      Image im;
      im.image_id = f;
      im.camera_id = 1;
      im.frame_id = f;
      im3[f] = im;
    }
    for (int f = 0; f < NG; ++f) im3[f].frame_ptr = &fr3[f];
    for (int a = 0; a < NG; ++a)
      for (int b = a + 1; b < NG; ++b) {
        if (b - a > 3 && b - a < NG - 3) continue;
        ImagePair pr;
        pr.image_id1 = a;
        pr.image_id2 = b;
        pr.cam2_from_cam1.rotation = qmul(qgt[b], qinv(qgt[a]));
        pr.inliers.resize(60);
        vg3.image_pairs[(uint64_t)a * 1000 + b] = pr;
      }
    RotationEstimatorOptions rog;
    rog.use_gravity = true;
    gsfm_glomap::RotationEstimator rag(rog);
    if (!rag.EstimateRotations(vg3, rigs3, fr3, im3)) return std::printf("gravity RA failed\n"), 1;
    for (int a = 0; a < NG; ++a)
      for (int b = a + 1; b < NG; ++b) {
        const auto e = qmul(qmul(fr3[b].RigFromWorld().rotation, qinv(fr3[a].RigFromWorld().rotation)), qinv(qmul(qgt[b], qinv(qgt[a]))));
        grav_err = std::fmax(grav_err, 2.0 * std::asin(std::fmin(1.0, std::sqrt(e.x() * e.x() + e.y() * e.y() + e.z() * e.z()))));
      }
    if (grav_err > 1e-6) return std::printf("gravity RA: relative rotation off by %.3e rad\n", grav_err), 1;
  }
  // use_gpu == false: the solve goes to the reference class (here: the counting stand-ins), nothing is touched by libgsfm
  {
    GlobalPositionerOptions go;
    go.use_gpu = false;
    BundleAdjusterOptions bo;
    bo.use_gpu = false;
    const int g0 = glomap::GlobalPositioner::calls(), b0 = glomap::BundleAdjuster::calls();
    gsfm_glomap::GlobalPositioner gp_cpu(go);
    gsfm_glomap::BundleAdjuster ba_cpu(bo);
    if (!gp_cpu.Solve(vg, rigs, cameras, frames, images, tracks) || !ba_cpu.Solve(rigs, cameras, frames, images, tracks))
      return std::printf("use_gpu = false: delegated solve failed\n"), 1;
    if (glomap::GlobalPositioner::calls() != g0 + 1 || glomap::BundleAdjuster::calls() != b0 + 1)
      return std::printf("use_gpu = false did not reach the reference estimators\n"), 1;
  }
  std::printf("ADAPTER OK gravity_ra=%.2e rad ", grav_err);
  std::printf("ra=%.2e rad gp_ratio_err=%.2e ba=%.2e px | rigs: ra=%.2e rad gp_scale_err=%.2e ba=%.2e px cam_from_rig=%.2e rad unknown_R=%.2e rad unknown_t=%.2e\n",
              worst, std::fabs(ratio / ratio_ref - 1.0), maxerr, rig_ra, rig_gp, rig_ba, rig_sens, rig_unk_rot, rig_unk);
  return 0;
}
