"""THE DROP-IN AT THE TOP OF THE HOT PATH: the reference's own `GlobalMapper::Solve`.

oracle/_ref/libref_dropin_mapper.so holds glomap/controllers/global_mapper.cc — rotation averaging twice with the rotation filter
and the largest component between, track establishment, global positioning from a random start, the three track filters, the
normaliser, the bundle adjustment rounds (positions only, then everything) with their reprojection filters
(global_mapper.cc:85-278) — compiled from /root/reference, unmodified, THREE times (oracle/Makefile, `ref_mapper`):

  which = 0   as the reference builds it: its controller on its own RotationEstimator / GlobalPositioner / BundleAdjuster and its
              own processors — reference code all the way, on the CPU (the absent Eigen / COLMAP / Ceres / Boost are the stand-in
              headers of oracle/ref_shim*/; Ceres is the SOLVING stand-in, so the estimators run to their end points)
  which = 1   oracle/ref_dropin_mapper_on_gsfm.cc: the same source with the estimator names switched at their use sites —
              INTEGRATION.md section 2's diff — to the classes of include/gsfm_glomap_adapter.hpp: the reference's controller
              driving libgsfm on the GPU from the reference's own std::unordered_map containers and option structs
  which = 2   as 1, and TrackFilter / NormalizeReconstruction / RelPoseFilter switched to libgsfm as well
  which = 3   as 2, and TrackEngine too: the same tracks under other ids — another walk of the `tracks` map, another draw of the
              random start of global positioning — an equally valid run, held to ground truth and to the track SETS of which = 0

The stages outside SURVEY section 8 (preprocessing, view-graph calibration, relative-pose estimation, retriangulation, pruning)
are skipped with the reference's own GlobalMapperOptions::skip_* switches.

CPU: which = 0 recovers synthetic scenes at the pins of the reference's own end-to-end tests (global_mapper_test.cc:82-86: rotations
1e-2 degrees and centres 1e-4 on noise-free data; :211-215: 0.1 degrees / 0.1 with noise and outliers) — the reference's controller,
estimators and processors as compiled here are a working GLOMAP.  GPU: which = 1 and 2 against which = 0 on the same containers:
FINAL camera poses at north_star's bar (1e-4 rad, 1e-3 of the extent), the same frames registered, the same pairs valid, the same
tracks kept.
"""
import numpy as np
import pytest

from glomap_amd import so3, synthetic
from oracle import ref

pytestmark = pytest.mark.skipif(ref.load_mapper() is None, reason="oracle/_ref: neither /root/reference + libgsfm.so nor a prebuilt oracle/_ref/libref_dropin_mapper.so")

SCENES = {
    # name: (generator arguments, mapper options, rotation pin in degrees, centre pin relative to the extent)
    "noise_free": (dict(n_images=14, n_points=80, seed=0), dict(), 1e-2, 1e-4),
    "noisy_outliers": (dict(n_images=30, n_points=400, seed=1, pixel_noise=0.5, rot_outlier_pairs=4, false_match_frac=0.0005), dict(), 0.1, 0.1),
    "isolated_pair": (dict(n_images=16, n_points=120, seed=2, pixel_noise=0.2, isolated_pair=True), dict(), 0.1, 0.1),
    "fixed_intrinsics": (dict(n_images=20, n_points=200, seed=3, pixel_noise=0.3), dict(optimize_intrinsics=0, num_iteration_bundle_adjustment=2), 0.1, 0.1),
    # SIMPLE_RADIAL cameras: UndistortImages goes through COLMAP's iterative undistortion (the device sweep's Newton on the other side)
    "simple_radial": (dict(n_images=18, n_points=150, seed=4, pixel_noise=0.3, simple_radial=-0.08), dict(), 0.1, 0.1),
    # above libgsfm's single-workgroup sizes on the GP side (72 000 observations); ~17 s of reference code on the Ceres stand-in
    "sixty_images": (dict(n_images=60, n_points=1200, seed=5, pixel_noise=0.3, num_succ=8), dict(), 0.1, 0.1),
}


def _scene(gen):
    gen = dict(gen)
    k = gen.pop("simple_radial", None)
    s = synthetic.make_pipeline_scene(**gen)
    if k is not None:  # the same rays seen through two shared SIMPLE_RADIAL cameras (f, cx, cy, k)
        p = s["intr_params"][s["cam_intr"]][np.repeat(np.arange(s["num_images"]), np.diff(s["feat_offset"]))]
        x = s["feat_undist"][:, :2] / s["feat_undist"][:, 2:3]
        xd = x * (1.0 + k * (x * x).sum(1, keepdims=True))
        s["feat_xy"] = p[:, 0:1] * xd + p[:, 2:4]
        s["intr_model"] = np.full(len(s["intr_model"]), 2, dtype=np.int32)
        par = np.zeros_like(s["intr_params"])
        par[:, 0], par[:, 1:3], par[:, 3] = s["intr_params"][:, 0], s["intr_params"][:, 2:4], k
        s["intr_params"] = par
    return s


def _solve(which, s, **options):
    K = len(s["intr_model"])
    cp = np.zeros((K, 12))
    cp[:, :8] = s["intr_params"]
    E = len(s["pair_image1"])
    return ref.mapper_solve(which, s["intr_model"], cp, s["cam_intr"], s["feat_offset"], s["feat_xy"], s["pair_image1"], s["pair_image2"],
                            s["pair_q"], np.zeros((E, 3)), s["pair_offset"], s["match_feat1"], s["match_feat2"], **options)


def _poses(r):
    R = so3.quat_to_rotmat(r["frame_q"])
    return R, -np.einsum("nji,nj->ni", R, r["frame_t"])


@pytest.mark.parametrize("name", list(SCENES))
def test_reference_mapper_as_compiled_here_recovers_the_scene(name):
    """which = 0: GlobalMapper::Solve of the reference, every line of it reference code, at the reference's own pins."""
    gen, opt, pin_deg, pin_c = SCENES[name]
    s = _scene(gen)
    r = _solve(0, s, **opt)
    assert r["ok"]
    N0 = s["num_ring_images"]
    reg = r["frame_registered"]
    assert reg[:N0].all() and not reg[N0:].any()  # the isolated pair is not in the largest component (global_mapper.cc:107-111)
    R, c = _poses(r)
    rot = synthetic.rotation_errors_deg(R[reg], s["gt_R"][reg]).max()
    cen = synthetic.center_errors_after_sim3(c[reg], s["gt_center"][reg]).max() / synthetic.scene_extent(s["gt_center"][reg])
    print(f"[parity] reference GlobalMapper::Solve (compiled unmodified, CPU) {name}: {int(reg.sum())} images, {r['num_tracks']} tracks / "
          f"{r['num_observations']} observations, rotations {rot:.2e} deg, centres {cen:.2e} of the extent vs ground truth")
    assert rot < pin_deg and cen < pin_c
    assert r["num_tracks"] > 0.8 * gen["n_points"] and r["num_initialized"] == r["num_tracks"]
    if gen.get("rot_outlier_pairs"):
        bad = s["pair_rot_outlier"]
        assert bad is not None and not r["pair_valid"][bad].any() and r["pair_valid"][~bad].all()  # RelPoseFilter::FilterRotations


@pytest.mark.gpu
@pytest.mark.parametrize("which", [1, 2])
@pytest.mark.parametrize("name", list(SCENES))
def test_reference_mapper_drives_libgsfm_to_the_reference_poses(name, which, gsfm_ctx):
    """which = 1 / 2 against which = 0: the reference's controller source on the adapter classes ends where it ends on the
    reference's own estimators — final poses at north_star's bar, identical discrete decisions."""
    gen, opt, _, _ = SCENES[name]
    s = _scene(gen)
    a = _solve(0, s, **opt)
    b = _solve(which, s, **opt)
    assert a["ok"] and b["ok"]
    assert np.array_equal(a["frame_registered"], b["frame_registered"])
    assert np.array_equal(a["pair_valid"], b["pair_valid"])
    reg = a["frame_registered"]
    Ra, ca = _poses(a)
    Rb, cb = _poses(b)
    # both runs fix the same gauge (the first frame stays where global positioning and the normaliser left it), so the poses
    # are compared as they are: no alignment
    dR = np.einsum("nij,nkj->nik", Ra[reg], Rb[reg])
    rot = np.linalg.norm(so3.quat_to_aa(so3.rotmat_to_quat(dR)), axis=1).max()
    ext = synthetic.scene_extent(ca[reg])
    cen = np.linalg.norm(ca[reg] - cb[reg], axis=1).max() / ext
    same_tracks = a["num_tracks"] == b["num_tracks"] and np.array_equal(a["track_id"], b["track_id"]) and np.array_equal(a["track_len"], b["track_len"])
    xyz = np.linalg.norm(a["track_xyz"] - b["track_xyz"], axis=1).max() / ext if same_tracks else float("nan")
    intr = np.abs(a["cam_params"] - b["cam_params"]).max() / np.abs(a["cam_params"]).max()
    what = "estimators + UndistortImages" if which == 1 else "estimators + UndistortImages + TrackFilter + NormalizeReconstruction + RelPoseFilter"
    print(f"[parity] DROP-IN GlobalMapper::Solve {name} ({what} on libgsfm) vs the reference's own: rotations {rot:.2e} rad, centres {cen:.2e}, "
          f"points {xyz:.2e} of the extent, intrinsics {intr:.2e} rel; {a['num_tracks']} tracks / {a['num_observations']} observations both")
    assert same_tracks and a["num_observations"] == b["num_observations"]
    assert rot < 1e-4 and cen < 1e-3 and xyz < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_reference_mapper_with_every_switchable_stage_on_libgsfm(name, gsfm_ctx):
    """which = 3: track establishment and selection on libgsfm as well.  gsfm_glomap::TrackEngine names a track by its smallest
    member where the reference uses the union-find root, so the `tracks` map walks in another order and GlobalPositioner's random
    start is another draw: the run is held to the reference's own pins against ground truth, and to the all-reference build in what
    does not depend on the draw — registered frames, valid pairs, the number of tracks and observations entering the pipeline."""
    gen, opt, pin_deg, pin_c = SCENES[name]
    s = _scene(gen)
    a = _solve(0, s, **opt)
    b = _solve(3, s, **opt)
    assert a["ok"] and b["ok"]
    assert np.array_equal(a["frame_registered"], b["frame_registered"]) and np.array_equal(a["pair_valid"], b["pair_valid"])
    reg = b["frame_registered"]
    R, c = _poses(b)
    rot = synthetic.rotation_errors_deg(R[reg], s["gt_R"][reg]).max()
    cen = synthetic.center_errors_after_sim3(c[reg], s["gt_center"][reg]).max() / synthetic.scene_extent(s["gt_center"][reg])
    Ra, ca = _poses(a)
    rot_a = synthetic.rotation_errors_deg(Ra[reg], s["gt_R"][reg]).max()
    cen_a = synthetic.center_errors_after_sim3(ca[reg], s["gt_center"][reg]).max() / synthetic.scene_extent(s["gt_center"][reg])
    print(f"[parity] DROP-IN GlobalMapper::Solve {name}, every switchable stage on libgsfm (other track ids, another random start): vs ground truth "
          f"{rot:.2e} deg / {cen:.2e} (all-reference build: {rot_a:.2e} deg / {cen_a:.2e}); tracks {b['num_tracks']} / {a['num_tracks']}, "
          f"observations {b['num_observations']} / {a['num_observations']}")
    assert rot < pin_deg and cen < pin_c
    assert abs(b["num_tracks"] - a["num_tracks"]) <= 0.01 * a["num_tracks"] and abs(b["num_observations"] - a["num_observations"]) <= 0.01 * a["num_observations"]
