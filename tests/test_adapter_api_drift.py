"""Guards the adapter's stand-in headers against drift from the real GLOMAP sources.

include/gsfm_glomap_adapter.hpp is compiled and run against tests/adapter/mock/glomap/mock_types.h because GLOMAP / COLMAP /
Eigen are not installed here.  Every member or method of a GLOMAP / COLMAP scene type the adapter touches is listed below
with the reference file that declares or uses it; the test checks that (1) the name still appears there, (2) the adapter
really uses it and (3) the stand-in declares it — so a rename on either side shows up.  Runs where the reference tree
exists (this container); it is skipped on the GPU box."""
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (identifier, reference file declaring or using it, regex that must match there)
API = [
    # glomap::Image (scene/image.h)
    ("frame_ptr", "glomap/scene/image.h", r"frame_ptr"),
    ("frame_id", "glomap/scene/image.h", r"frame_id"),
    ("camera_id", "glomap/scene/image.h", r"camera_id"),
    ("features", "glomap/scene/image.h", r"std::vector<Eigen::Vector2d>\s+features"),
    ("features_undist", "glomap/scene/image.h", r"std::vector<Eigen::Vector3d>\s+features_undist"),
    ("IsRegistered", "glomap/scene/image.h", r"IsRegistered\(\)"),
    ("HasTrivialFrame", "glomap/scene/image.h", r"HasTrivialFrame\(\)"),
    ("HasGravity", "glomap/scene/image.h", r"HasGravity\(\)"),
    ("CamFromWorld", "glomap/scene/image.h", r"CamFromWorld\(\)"),
    # glomap::Frame / GravityInfo (scene/frame.h)
    ("is_registered", "glomap/scene/frame.h", r"bool\s+is_registered"),
    ("gravity_info", "glomap/scene/frame.h", r"GravityInfo\s+gravity_info"),
    ("has_gravity", "glomap/scene/frame.h", r"bool\s+has_gravity"),
    ("GetRAlign", "glomap/scene/frame.h", r"GetRAlign\(\)"),
    # colmap::Frame / colmap::Rig as GLOMAP uses them
    ("RigFromWorld", "glomap/estimators/global_rotation_averaging.cc", r"\.RigFromWorld\(\)"),
    ("SetRigFromWorld", "glomap/estimators/global_rotation_averaging.cc", r"SetRigFromWorld\("),
    ("HasPose", "glomap/estimators/bundle_adjustment.cc", r"\.HasPose\(\)"),
    ("RigId", "glomap/estimators/global_rotation_averaging.cc", r"\.RigId\(\)"),
    ("NonRefSensors", "glomap/estimators/global_rotation_averaging.cc", r"NonRefSensors\(\)"),
    ("MaybeSensorFromRig", "glomap/estimators/global_rotation_averaging.cc", r"MaybeSensorFromRig\("),
    ("SensorFromRig", "glomap/estimators/bundle_adjustment.cc", r"\.SensorFromRig\("),
    ("SetSensorFromRig", "glomap/estimators/global_rotation_averaging.cc", r"SetSensorFromRig\("),
    ("SensorType::CAMERA", "glomap/estimators/global_rotation_averaging.cc", r"SensorType::CAMERA"),
    # glomap::ImagePair / ViewGraph
    ("image_pairs", "glomap/scene/view_graph.h", r"image_pairs"),
    ("is_valid", "glomap/scene/image_pair.h", r"bool\s+is_valid"),
    ("image_id1", "glomap/scene/image_pair.h", r"image_id1"),
    ("cam2_from_cam1", "glomap/scene/image_pair.h", r"Rigid3d\s+cam2_from_cam1"),
    ("weight", "glomap/scene/image_pair.h", r"double\s+weight"),
    ("inliers", "glomap/scene/image_pair.h", r"std::vector<int>\s+inliers"),
    ("matches", "glomap/scene/image_pair.h", r"Eigen::MatrixXi\s+matches"),
    # glomap::Track / Camera
    ("observations", "glomap/scene/track.h", r"std::vector<Observation>\s+observations"),
    ("xyz", "glomap/scene/track.h", r"Eigen::Vector3d\s+xyz"),
    ("is_initialized", "glomap/scene/track.h", r"bool\s+is_initialized"),
    ("has_prior_focal_length", "glomap/estimators/global_positioning.cc", r"has_prior_focal_length"),
    ("model_id", "glomap/estimators/bundle_adjustment.cc", r"\.model_id"),
    ("params", "glomap/estimators/bundle_adjustment.cc", r"\.params"),
    # option structs
    ("max_num_l1_iterations", "glomap/estimators/global_rotation_averaging.h", r"max_num_l1_iterations"),
    ("irls_loss_parameter_sigma", "glomap/estimators/global_rotation_averaging.h", r"irls_loss_parameter_sigma"),
    ("weight_type", "glomap/estimators/global_rotation_averaging.h", r"weight_type"),
    ("skip_initialization", "glomap/estimators/global_rotation_averaging.h", r"skip_initialization"),
    ("use_weight", "glomap/estimators/global_rotation_averaging.h", r"use_weight"),
    ("use_gravity", "glomap/estimators/global_rotation_averaging.h", r"use_gravity"),
    ("generate_random_positions", "glomap/estimators/global_positioning.h", r"generate_random_positions"),
    ("optimize_scales", "glomap/estimators/global_positioning.h", r"optimize_scales"),
    ("constraint_type", "glomap/estimators/global_positioning.h", r"constraint_type"),
    ("seed", "glomap/estimators/global_positioning.h", r"unsigned\s+seed"),
    ("gpu_index", "glomap/estimators/global_positioning.h", r"gpu_index"),
    ("use_gpu", "glomap/estimators/global_positioning.h", r"bool\s+use_gpu"),
    ("min_num_images_gpu_solver", "glomap/estimators/bundle_adjustment.h", r"int\s+min_num_images_gpu_solver"),
    ("optimize_rig_poses", "glomap/estimators/bundle_adjustment.h", r"optimize_rig_poses"),
    ("optimize_principal_point", "glomap/estimators/bundle_adjustment.h", r"optimize_principal_point"),
    ("min_num_view_per_track", "glomap/estimators/bundle_adjustment.h", r"min_num_view_per_track"),
    ("thres_loss_function", "glomap/estimators/optimization_base.h", r"thres_loss_function"),
    ("solver_options", "glomap/estimators/optimization_base.h", r"solver_options"),
]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("ident,ref_file,pattern", API, ids=[a[0] for a in API])
def test_identifier_exists_in_reference_adapter_and_stand_in(ident, ref_file, pattern):
    with open(os.path.join(REF, ref_file)) as f:
        assert re.search(pattern, f.read()), f"{ident}: not found in reference {ref_file}"
    base = ident.split("::")[-1]
    with open(os.path.join(ROOT, "include", "gsfm_glomap_adapter.hpp")) as f:
        assert re.search(r"\b" + re.escape(base) + r"\b", f.read()), f"{ident}: the adapter does not use it (stale entry)"
    with open(os.path.join(ROOT, "tests", "adapter", "mock", "glomap", "mock_types.h")) as f:
        assert re.search(r"\b" + re.escape(base) + r"\b", f.read()), f"{ident}: missing from the stand-in headers"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_estimator_signatures_match_the_reference_headers():
    """Solve / EstimateRotations take the containers in the reference's order."""
    with open(os.path.join(ROOT, "include", "gsfm_glomap_adapter.hpp")) as f:
        ad = re.sub(r"\s+", " ", f.read())
    for header, method, order in [
        ("glomap/estimators/global_rotation_averaging.h", "EstimateRotations", ["ViewGraph", "rig_t", "frame_t", "image_t"]),
        ("glomap/estimators/global_positioning.h", "Solve", ["ViewGraph", "rig_t", "camera_t", "frame_t", "image_t", "track_t"]),
        ("glomap/estimators/bundle_adjustment.h", "Solve", ["rig_t", "camera_t", "frame_t", "image_t", "track_t"]),
    ]:
        with open(os.path.join(REF, header)) as f:
            ref = re.sub(r"\s+", " ", f.read())
        m = re.search(r"bool " + method + r"\((.*?)\);", ref)
        assert m, (header, method)
        pos = [m.group(1).find(t) for t in order]
        assert all(p >= 0 for p in pos) and pos == sorted(pos), (header, m.group(1))
        hits = [mm.group(1) for mm in re.finditer(r"bool " + method + r"\((.*?)\) \{", ad)]
        assert any(all(a.find(t) >= 0 for t in order) and [a.find(t) for t in order] == sorted(a.find(t) for t in order)
                   for a in hits), (method, hits)
