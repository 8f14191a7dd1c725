"""THE DROP-IN, on the reference's own controller code.

oracle/_ref/libref_dropin_ra.so holds glomap/controllers/rotation_averager.cc — SolveRotationAveraging: largest component, the
stratified 1-DoF pre-solve, the trivial-rig pre-pass for unknown cam_from_rig, ConvertRotationsFromImageToRig, the final solve —
compiled from /root/reference, unmodified, TWICE (oracle/Makefile, `ref_dropin`):

  which = 0   as the reference builds it, on its own RotationEstimator (global_rotation_averaging.cc & co. on the stand-ins of
              oracle/ref_shim_ra/): reference code all the way, on the CPU
  which = 1   with oracle/ref_shim_dropin/ first on the include path, whose one header makes the name glomap::RotationEstimator
              the class of include/gsfm_glomap_adapter.hpp — the include / namespace switch INTEGRATION.md describes — i.e. the
              reference's controller driving libgsfm on the GPU through the adapter, from the reference's own
              std::unordered_map containers and its own option structs (the reference's real headers, not tests/adapter/mock)

CPU: the scenarios of the reference's rotation_averager_test.cc (shaped by tests/test_rotation_averager_policy.py::make_scene) on
which = 0 — the reference's pins hold on its own controller + estimator code as compiled here — and the Python mirror of the
controller (glomap_amd/rotation_averager.py, oracle backend) ends with the same registered frames, valid pairs, estimated
sensors and relative rotations.  GPU: which = 1 against which = 0 on noisy data, same containers: the rotations agree to 1e-6 rad.
"""
import copy

import numpy as np
import pytest

from glomap_amd import estimators, rotation_averager as rav, so3
from glomap_amd.scene import Rigid3d
from oracle import ref
from test_rotation_averager_policy import OracleBackend, _errors_deg, make_scene

pytestmark = pytest.mark.skipif(ref.load_dropin() is None, reason="oracle/_ref: neither /root/reference + libgsfm.so nor a prebuilt oracle/_ref/libref_dropin_ra.so")


def _flat(vg, rigs, frames, images):
    """The Python containers as the flat arrays of oracle/ref_glue_ra_scene.h (ids -> indices in ascending id order)."""
    fid = {f: k for k, f in enumerate(sorted(frames))}
    iid = {i: k for k, i in enumerate(sorted(images))}
    rid = {r: k for k, r in enumerate(sorted(rigs))}
    a = dict(rig_ref_cam=[rigs[r].ref_camera_id for r in sorted(rigs)],
             frame_rig=[rid[frames[f].rig_id] for f in sorted(frames)],
             image_frame=[fid[images[i].frame_id] for i in sorted(images)], image_cam=[images[i].camera_id for i in sorted(images)])
    sr, sc, ss, sq = [], [], [], []
    for r in sorted(rigs):
        for cam, cfr in rigs[r].NonRefSensors():
            sr.append(rid[r]); sc.append(cam)
            if cfr is None:
                ss.append(0); sq.append([1.0, 0, 0, 0])
            else:
                ss.append(2 if np.isnan(cfr.translation).any() else 1); sq.append(list(cfr.rotation))
    a.update(sensor_rig=sr, sensor_cam=sc, sensor_state=ss, sensor_q=np.array(sq).reshape(-1, 4) if sq else None)
    a["frame_q"] = np.array([frames[f].rig_from_world.rotation for f in sorted(frames)])
    Ra = np.full((len(frames), 3, 3), np.nan)
    for f in sorted(frames):
        if frames[f].HasGravity():
            Ra[fid[f]] = frames[f].GetRAlign()
    a["frame_R_align"] = Ra
    a["frame_registered"] = [frames[f].is_registered for f in sorted(frames)]
    keys = list(vg.image_pairs)
    a["pair_i"] = [iid[vg.image_pairs[k].image_id1] for k in keys]
    a["pair_j"] = [iid[vg.image_pairs[k].image_id2] for k in keys]
    a["pair_q"] = np.array([vg.image_pairs[k].cam2_from_cam1.rotation for k in keys])
    a["pair_weight"] = [vg.image_pairs[k].weight for k in keys]
    a["pair_ninl"] = [vg.image_pairs[k].inlier_count() for k in keys]
    a["pair_valid"] = [vg.image_pairs[k].is_valid for k in keys]
    return a, keys


def _write_back(r, keys, vg, rigs, frames):
    """Result of ref.ra_policy into (copies of) the Python containers, for _errors_deg."""
    for k, f in enumerate(sorted(frames)):
        frames[f].rig_from_world = Rigid3d(r["frame_q"][k].copy(), np.zeros(3))
        frames[f].is_registered = bool(r["frame_registered"][k])
    s = 0
    for rg in sorted(rigs):
        for cam, _ in rigs[rg].NonRefSensors():
            if r["sensor_has"][s]:
                old = rigs[rg].sensors[cam]
                rigs[rg].sensors[cam] = Rigid3d(r["sensor_q"][s].copy(), np.full(3, np.nan) if old is None else old.translation)
            s += 1
    for k, key in enumerate(keys):
        vg.image_pairs[key].is_valid = bool(r["pair_valid"][k])


def _image_rotations(frames, rigs, images):
    out = {}
    for i, im in images.items():
        fr = frames[im.frame_id]
        if not fr.is_registered:
            continue
        R = so3.quat_to_rotmat(np.asarray(fr.rig_from_world.rotation)[None])[0]
        if not rav.has_trivial_frame(im, frames, rigs):
            R = so3.quat_to_rotmat(np.asarray(rigs[fr.rig_id].MaybeSensorFromRig(im.camera_id).rotation)[None])[0] @ R
        out[i] = R
    return out


def _relative_distance_deg(A, B):
    """Largest difference between the relative rotations of two solutions (gauge-free)."""
    ids = sorted(A)
    assert ids == sorted(B)
    worst = 0.0
    for a in ids:
        for b in ids:
            if b > a:
                worst = max(worst, float(so3.rotation_angle_deg((A[b] @ A[a].T)[None], (B[b] @ B[a].T)[None])[0]))
    return worst


SCENARIOS = {
    "trivial": (dict(num_frames=10, cams_per_rig=1), dict()),
    "known_rig": (dict(num_frames=8, cams_per_rig=2), dict()),
    "known_rig_gravity": (dict(num_frames=8, cams_per_rig=2, gravity=range(8), start="align"), dict(use_gravity=True)),
    "mixed_gravity_stratified": (dict(num_frames=12, cams_per_rig=1, reach=3, gravity=range(0, 12, 2), start="align"), dict(use_gravity=True)),
    "unknown_rig": (dict(num_frames=8, cams_per_rig=2, unknown=True), dict()),
    "partly_calibrated_rig": (dict(num_frames=8, cams_per_rig=3, unknown={2}), dict()),
    "two_unknown_rigs": (dict(num_frames=8, cams_per_rig=3, num_rigs=2, unknown=True, reach=3), dict()),
}


def _scene(name, noise_deg=0.0):
    kw, opt = SCENARIOS[name]
    vg, rigs, frames, images, R_img, R_s = make_scene(**kw)
    if name == "mixed_gravity_stratified":  # frames without gravity start near the truth (no spanning-tree start with use_gravity)
        rng = np.random.default_rng(3)
        for f, fr in frames.items():
            if not fr.HasGravity():
                R = so3.aa_to_rotmat(rng.normal(0, 0.05, (1, 3)))[0] @ R_img[f]
                fr.rig_from_world = Rigid3d(so3.rotmat_to_quat(R[None])[0], np.zeros(3))
    if noise_deg:
        rng = np.random.default_rng(17)
        for pr in vg.image_pairs.values():
            R = so3.aa_to_rotmat(rng.normal(0, np.radians(noise_deg), (1, 3)))[0] @ so3.quat_to_rotmat(np.asarray(pr.cam2_from_cam1.rotation)[None])[0]
            pr.cam2_from_cam1 = Rigid3d(so3.rotmat_to_quat(R[None])[0], np.zeros(3))
    return vg, rigs, frames, images, R_img, opt


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_reference_controller_on_reference_estimator(name):
    """which = 0 on noise-free data: the pins of rotation_averager_test.cc (1e-2 degrees on every relative rotation, :166-167,
    209-210, 260-261) hold for the reference's controller + estimator as compiled here, and the Python mirror of the controller
    ends in the same place."""
    vg, rigs, frames, images, R_img, opt = _scene(name)
    a, keys = _flat(vg, rigs, frames, images)
    r = ref.ra_policy(0, **a, **opt)
    assert r["ok"]
    vg_r, rigs_r, frames_r = copy.deepcopy(vg), copy.deepcopy(rigs), copy.deepcopy(frames)
    _write_back(r, keys, vg_r, rigs_r, frames_r)
    assert _errors_deg(frames_r, rigs_r, images, R_img) < 1e-2
    # the Python mirror (glomap_amd/rotation_averager.py) with the oracle as numerical backend
    assert rav.SolveRotationAveraging(vg, rigs, frames, images, estimators.RotationEstimatorOptions(**opt), backend=OracleBackend())
    assert [frames[f].is_registered for f in sorted(frames)] == [frames_r[f].is_registered for f in sorted(frames)]
    assert [vg.image_pairs[k].is_valid for k in keys] == [vg_r.image_pairs[k].is_valid for k in keys]
    for rg in rigs:
        for cam, cfr in rigs[rg].NonRefSensors():
            cr = rigs_r[rg].MaybeSensorFromRig(cam)
            assert (cfr is None) == (cr is None) and np.isnan(cfr.translation).all() == np.isnan(cr.translation).all()
    d = _relative_distance_deg(_image_rotations(frames, rigs, images), _image_rotations(frames_r, rigs_r, images))
    print(f"[dropin] {name}: reference controller vs Python mirror, relative rotations {d:.2e} deg")
    assert d < 1e-3


def test_reference_controller_unregisters_an_island():
    from glomap_amd.scene import Frame, Image, ImagePair

    vg, rigs, frames, images, R_img, _ = make_scene(8, 1)
    for f in (8, 9):  # two frames linked only to each other
        frames[f] = Frame(f, Rigid3d(), True, 1, [f])
        images[f] = Image(f, 100, f)
    vg.image_pairs[(8, 9)] = ImagePair(8, 9, Rigid3d())
    a, keys = _flat(vg, rigs, frames, images)
    r = ref.ra_policy(0, **a)
    assert r["ok"] and r["frame_registered"].tolist() == [True] * 8 + [False] * 2
    assert not r["pair_valid"][keys.index((8, 9))] and r["pair_valid"].sum() == len(keys) - 1


@pytest.mark.gpu
@pytest.mark.parametrize("images_reversed", [False, True], ids=["same_walk", "images_walk_differs"])
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_reference_controller_on_libgsfm_equals_reference_controller_on_reference_estimator(name, images_reversed):
    """which = 1 vs which = 0, noisy relative rotations (0.5 degrees), the SAME containers and options: every frame rotation and
    every estimated cam_from_rig rotation within 1e-6 rad, the same frames registered and pairs valid.  With images_reversed the
    `images` map iterates in another order than `frames`: the spanning tree's root (first registered image, tree.cc:84-88) and
    the gauge (first registered frame, gra.cc:248-257) then sit in different frames, and the adapter has to follow both walks."""
    vg, rigs, frames, images, R_img, opt = _scene(name, noise_deg=0.5)
    a, keys = _flat(vg, rigs, frames, images)
    r0 = ref.ra_policy(0, **a, **opt, images_reversed=images_reversed)
    r1 = ref.ra_policy(1, **a, **opt, images_reversed=images_reversed)
    assert r0["ok"] and r1["ok"]
    assert np.array_equal(r0["frame_registered"], r1["frame_registered"]) and np.array_equal(r0["pair_valid"], r1["pair_valid"])
    assert np.array_equal(r0["sensor_has"], r1["sensor_has"])

    def dist(qa, qb):
        d = so3.quat_mul(so3.quat_conj(qa), qb)
        return 2.0 * np.arcsin(np.minimum(1.0, np.linalg.norm(d[:, 1:], axis=1)))

    df = dist(r0["frame_q"], r1["frame_q"]).max()
    ds = dist(r0["sensor_q"][r0["sensor_has"]], r1["sensor_q"][r1["sensor_has"]]).max() if r0["sensor_has"].any() else 0.0
    print(f"[parity] DROP-IN {name}{' (images walk differs)' if images_reversed else ''}: reference controller on libgsfm vs on the reference estimator: frames {df:.2e} rad, sensors {ds:.2e} rad")
    assert df < 1e-6 and ds < 1e-6


# ---------------------------------------------------------------------------------------------------------------
# GlobalPositioner / BundleAdjuster of the adapter on the containers the reference's own classes get
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1])
def test_adapter_global_positioner_on_reference_containers(seed):
    """Same flat arrays -> same std::unordered_map containers (oracle/ref_glue_gp_scene.h), once into the reference's
    GlobalPositioner::Solve on the recording Ceres (ref.gp_build) and once into the adapter's (ref.gp_adapter_solve, libgsfm):
    the cost of the start is the same number — the adapter walked the hash maps the way the reference does, drew in its order
    (g++ argument order included: both sides are g++ builds), kept the same tracks — and the results come back through the
    containers the way ConvertResults leaves them."""
    from glomap_amd import synthetic

    p = synthetic.make_gp_problem(num_cams=60, num_pts=2000, seed=seed, uncalibrated_ratio=0.2, dir_noise=1e-3, outlier_ratio=0.02)
    # two-view tracks and an unobserved camera: in the walk, not in the problem
    keep = np.ones(p.num_obs, bool)
    for t in (3, 17, 40):
        keep[p.pt_offset[t] + 2 : p.pt_offset[t + 1]] = False
    keep[p.obs_cam == 5] = False
    lens = np.diff(p.pt_offset)
    trk = np.repeat(np.arange(p.num_pts), lens)
    off = np.zeros(p.num_pts + 1, dtype=np.int64)
    off[1:] = np.cumsum(np.bincount(trk[keep], minlength=p.num_pts))
    obs_cam, obs_dir, obs_cal = p.obs_cam[keep], p.obs_dir[keep], p.obs_calibrated[keep]
    q = so3.rotmat_to_quat(p.cam_R)
    t = -np.einsum("nij,nj->ni", p.cam_R, p.gt_center)
    und = np.einsum("mij,mj->mi", p.cam_R[obs_cam], obs_dir)
    cal = np.ones(p.num_cams, np.uint8)
    cal[obs_cam] = obs_cal
    X0 = np.random.default_rng(seed).normal(size=(p.num_pts, 3))
    r = ref.gp_build(q, t, off, obs_cam, und, X0, cam_calibrated=cal)
    a = ref.gp_adapter_solve(q, t, off, obs_cam, und, X0, cam_calibrated=cal)
    assert a["ok"]
    rel = abs(a["initial_cost"] - r["initial_cost"]) / r["initial_cost"]
    print(f"[parity] DROP-IN GP seed={seed}: adapter start cost {a['initial_cost']:.12e} vs reference builder {r['initial_cost']:.12e} (rel {rel:.1e})")
    assert rel < 1e-12
    short = np.diff(off) < 3
    assert not a["initialized"][short].any() and a["initialized"][~short].all()      # gp.cc:258-264
    assert np.array_equal(a["xyz"][short], X0[short])                                # tracks outside the problem keep their point
    assert np.abs(a["center"][5] - p.gt_center[5]).max() < 1e-12                     # the unconstrained camera keeps its centre (gp.cc:161)
    from glomap_amd import synthetic as syn

    ok_cams = np.arange(p.num_cams) != 5
    assert syn.center_errors_after_sim3(a["center"][ok_cams], p.gt_center[ok_cams]).max() < 0.1  # and the scene is recovered


@pytest.mark.gpu
def test_adapter_bundle_adjuster_on_reference_containers():
    """BundleAdjuster: the adapter's start cost equals the reference builder's on the same containers (which frame is constant
    follows the reference's walk of `frames`), the constant frame comes back bit-identical, the cost goes down."""
    from glomap_amd import synthetic

    p = synthetic.make_ba_problem(num_cams=40, num_pts=1500, seed=2, pixel_noise=0.7, outlier_ratio=0.02, intr_noise=0.01)
    args = (p.intr_model, p.intr_params, p.cam_q, p.cam_t, np.arange(p.num_cams), p.cam_intr, p.pt_offset, p.obs_cam, p.obs_xy, p.pt_xyz)
    kw = dict(rig_ref_cam=np.arange(p.num_intr), frame_rig=p.cam_intr)
    r = ref.ba_build(*args, **kw)
    a = ref.ba_adapter_solve(*args, **kw)
    assert a["ok"]
    rel = abs(a["initial_cost"] - r["initial_cost"]) / r["initial_cost"]
    print(f"[parity] DROP-IN BA: adapter start cost {a['initial_cost']:.12e} vs reference builder {r['initial_cost']:.12e} (rel {rel:.1e})")
    assert rel < 1e-12 and a["final_cost"] < 0.5 * a["initial_cost"]
    order = r["frame_order"]
    fixed = int(order[(r["frame_flags"][order] & 1) != 0][0])
    assert np.array_equal(a["frame_q"][fixed], p.cam_q[fixed]) and np.array_equal(a["frame_t"][fixed], p.cam_t[fixed])
    moved = np.abs(a["frame_t"] - p.cam_t).max(axis=1) > 0
    assert moved.sum() == p.num_cams - 1
    # principal points are held by the subset manifold (ba.cc:273-285)
    assert np.array_equal(a["cam_params"][:, 1:3], p.intr_params[:, 1:3]) and not np.array_equal(a["cam_params"][:, 0], p.intr_params[:, 0])
