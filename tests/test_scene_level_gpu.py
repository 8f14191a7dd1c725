"""The scene-level mirrors with their DEFAULT backend (libgsfm on the GPU): rotation_averager.SolveRotationAveraging /
RotationEstimator / KeepLargestConnectedComponents (glomap/controllers/rotation_averager.cc:8-198) and
mapper_estimators.GlobalPositioner / BundleAdjuster on the containers of glomap_amd.scene.

The scenarios are the SAME test functions tests/test_rotation_averager_policy.py and tests/test_mapper_rigs_cpu.py run on the
CPU with the oracle as backend — the nine rotation-averaging scenes shaped after rotation_averager_test.cc (pins
:166-167, 209-210, 258-261: relative rotations within 1e-2 degrees) and the two rig mapper scenes of
global_mapper_test.cc:89-175 (pins :121-125, 170-174: rotations within 1e-2 degrees, centres within 1e-4 after a similarity
alignment).  They are imported here and collected a second time with `make_backend` producing the product backend, which
only adds the bookkeeping the scenarios assert on (which kind of flat call each step made)."""
import numpy as np
import pytest

from glomap_amd import estimators, mapper_estimators as mest, rotation_averager as rav

import test_mapper_rigs_cpu as cpu_mapper
import test_rotation_averager_policy as cpu_policy

pytestmark = pytest.mark.gpu


class RecordingGpuBackend(rav.GpuBackend, mest.GpuBackend):
    """rotation_averager.GpuBackend + mapper_estimators.GpuBackend (both: glomap_amd.estimators through the C ABI) with
    the call log of the oracle backend of the CPU tests.  Nothing numerical happens here."""

    def __init__(self, ctx=None):
        rav.GpuBackend.__init__(self, ctx)
        self.calls = []

    def ra_solve(self, p, opt):
        if p.image_frame is not None:
            self.calls.append("cam_blocks")
        elif opt.use_gravity and p.node_gravity is not None:
            self.calls.append("gravity")
        else:
            self.calls.append("plain")
        return rav.GpuBackend.ra_solve(self, p, opt)


@pytest.fixture
def make_backend(gsfm_ctx):
    return lambda: RecordingGpuBackend(gsfm_ctx)


# ---- rotation averaging policy: the nine scenes -------------------------------------------------------------------
test_trivial_rigs_without_gravity = cpu_policy.test_trivial_rigs_without_gravity
test_known_rig_with_and_without_gravity = cpu_policy.test_known_rig_with_and_without_gravity
test_mixed_gravity_runs_the_stratified_pre_solve = cpu_policy.test_mixed_gravity_runs_the_stratified_pre_solve
test_unknown_rig_goes_through_the_trivial_pre_pass = cpu_policy.test_unknown_rig_goes_through_the_trivial_pre_pass
test_partly_calibrated_rig = cpu_policy.test_partly_calibrated_rig
test_estimator_alone_builds_the_start_for_unknown_sensors = cpu_policy.test_estimator_alone_builds_the_start_for_unknown_sensors
test_gravity_refuses_uncalibrated_rigs = cpu_policy.test_gravity_refuses_uncalibrated_rigs
test_largest_component_unregisters_the_rest = cpu_policy.test_largest_component_unregisters_the_rest

# ---- RA -> GP -> BA on rig scenes ---------------------------------------------------------------------------------
test_rig_scene_through_ra_gp_ba = cpu_mapper.test_rig_scene_through_ra_gp_ba
test_bundle_adjuster_refuses_uncalibrated_sensors = cpu_mapper.test_bundle_adjuster_refuses_uncalibrated_sensors
test_global_positioner_constraint_types_on_a_trivial_scene = cpu_mapper.test_global_positioner_constraint_types_on_a_trivial_scene


def test_default_backend_is_the_gpu_library():
    """No backend argument at all: the constructors and SolveRotationAveraging build their own GpuBackend (own context)."""
    assert isinstance(rav.RotationEstimator(estimators.RotationEstimatorOptions()).backend, rav.GpuBackend)
    assert isinstance(mest.GlobalPositioner(estimators.GlobalPositionerOptions()).backend, mest.GpuBackend)
    assert isinstance(mest.BundleAdjuster(estimators.BundleAdjusterOptions()).backend, mest.GpuBackend)
    vg, rigs, frames, images, R_img, _ = cpu_policy.make_scene(10, 1)
    assert rav.SolveRotationAveraging(vg, rigs, frames, images, estimators.RotationEstimatorOptions())
    assert cpu_policy._errors_deg(frames, rigs, images, R_img) < 1e-2
    # the trivial-rig mapper stage, default backends: RA (above) -> GP -> BA on a noise-free scene
    vg, rigs, cameras, frames, images, tracks, R_cw, c_gt = cpu_mapper.make_rig_scene(False, cams=2, seed=7)
    assert rav.SolveRotationAveraging(vg, rigs, frames, images, estimators.RotationEstimatorOptions())
    assert mest.GlobalPositioner(estimators.GlobalPositionerOptions()).Solve(vg, rigs, cameras, frames, images, tracks)
    assert mest.BundleAdjuster(estimators.BundleAdjusterOptions(optimize_rotations=False)).Solve(rigs, cameras, frames, images, tracks)
    assert mest.BundleAdjuster(estimators.BundleAdjusterOptions()).Solve(rigs, cameras, frames, images, tracks)
    from glomap_amd import synthetic

    R_fin, c_fin = cpu_mapper._image_poses(rigs, frames, images)
    assert synthetic.rotation_errors_deg(R_fin, R_cw).max() < 1e-2
    assert synthetic.center_errors_after_sim3(c_fin, c_gt).max() < 1e-4
