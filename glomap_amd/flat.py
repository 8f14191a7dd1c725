"""Flat structure-of-arrays problems — the exact memory layout the C ABI (include/gsfm.h) takes.

These replace the reference's pointer-linked ``std::unordered_map`` containers
(glomap/scene/view_graph.h:12-35, image_pair.h:13-57, image.h:10-53, frame.h:29-42,
track.h:10-27, camera.h:12-26) at the drop-in boundary.  All index arrays are dense int32
indices (not COLMAP ids); the scene<->flat mapping lives in glomap_amd/estimators.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# Camera model ids follow COLMAP's CameraModelId (colmap/sensor/models.h) for the models supported.
CAMERA_SIMPLE_PINHOLE = 0  # f, cx, cy
CAMERA_PINHOLE = 1  # fx, fy, cx, cy
CAMERA_SIMPLE_RADIAL = 2  # f, cx, cy, k
CAMERA_RADIAL = 3  # f, cx, cy, k1, k2
CAMERA_OPENCV = 4  # fx, fy, cx, cy, k1, k2, p1, p2
CAMERA_OPENCV_FISHEYE = 5  # fx, fy, cx, cy, k1, k2, k3, k4
CAMERA_FOV = 7  # fx, fy, cx, cy, omega
CAMERA_SIMPLE_RADIAL_FISHEYE = 8  # f, cx, cy, k
CAMERA_RADIAL_FISHEYE = 9  # f, cx, cy, k1, k2
CAMERA_FULL_OPENCV = 6  # fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6               ([K,16] intrinsics rows)
CAMERA_THIN_PRISM_FISHEYE = 10  # fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, sx1, sy1     ([K,16])
CAMERA_RAD_TAN_THIN_PRISM_FISHEYE = 11  # fx, fy, cx, cy, k0 .. k5, p0, p1, s0 .. s3   ([K,16])
CAMERA_NUM_PARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8, 7: 5, 8: 4, 9: 5, 6: 12, 10: 12, 11: 16}
CAMERA_PP_IDXS = {0: (1, 2), 1: (2, 3), 2: (1, 2), 3: (1, 2), 4: (2, 3)}
CAMERA_MAX_PARAMS = 8
CAMERA_MAX_PARAMS_WIDE = 16  # FULL_OPENCV, THIN_PRISM_FISHEYE, RAD_TAN_THIN_PRISM_FISHEYE: [K,16] intrinsics rows


@dataclass
class RaProblem:
    """Rotation-averaging view graph (reference reads: gra.cc:263-341, tree.cc:78-153)."""

    num_nodes: int
    edge_i: np.ndarray  # [E] int32, image_id1 index
    edge_j: np.ndarray  # [E] int32, image_id2 index
    edge_q: np.ndarray  # [E,4] f64 (w,x,y,z) = cam2_from_cam1.rotation   (x_j = R x_i)
    edge_weight: np.ndarray  # [E] f64, ImagePair::weight (used iff use_weight)
    edge_ninl: np.ndarray  # [E] int32, ImagePair::inliers.size() (MST weight)
    node_aa0: np.ndarray  # [N,3] f64 initial rig_from_world angle-axis
    fixed_node: int = 0
    gt_R: Optional[np.ndarray] = None  # [N,3,3] ground truth (tests only)
    outlier: Optional[np.ndarray] = None  # [E] bool (tests only)
    # Rigs with cam_from_rig rotations among the unknowns (gra.cc:173-191, 396-446): when given, edge_i / edge_j index
    # IMAGES, num_nodes / node_aa0 / fixed_node are the FRAMES, image_cam names the image's cam block (-1: reference or
    # calibrated sensor, folded into edge_q by the caller), cam_aa0 [C,3] holds the blocks' start values
    image_frame: Optional[np.ndarray] = None  # [I] int32
    image_cam: Optional[np.ndarray] = None  # [I] int32
    cam_aa0: Optional[np.ndarray] = None  # [C,3] f64
    # use_gravity (gra.cc:207-217, 376-418): 1 = the frame has gravity: one unknown, node_aa0[n] = (0, angle, 0); edge_q aligned
    node_gravity: Optional[np.ndarray] = None  # [N] uint8

    @property
    def num_edges(self) -> int:
        return int(self.edge_i.shape[0])


@dataclass
class GpProblem:
    """Global-positioning problem (reference: gp.cc:167-375): tracks and, for the constraint types other than ONLY_POINTS,
    camera-to-camera pairs.

    Observations are stored track-major: track p owns observations
    [pt_offset[p], pt_offset[p+1]).
    """

    num_cams: int
    num_pts: int
    pt_offset: np.ndarray  # [P+1] int64
    obs_cam: np.ndarray  # [M] int32
    obs_dir: np.ndarray  # [M,3] f64: R_cw^T * features_undist (gp.cc:294-296)
    obs_calibrated: np.ndarray  # [M] uint8: cameras[...].has_prior_focal_length (gp.cc:313-316)
    cam_center: np.ndarray  # [N,3] f64 in/out
    pt_xyz: np.ndarray  # [P,3] f64 in/out
    cam_R: Optional[np.ndarray] = None  # [N,3,3] cam_from_world rotations (for tests / conversion)
    gt_center: Optional[np.ndarray] = None
    gt_xyz: Optional[np.ndarray] = None
    # Known (calibrated) rigs, gp.cc:318-350 RigBATAPairwiseDirectionError: when given, obs_cam indexes IMAGES,
    # num_cams / cam_center are the FRAMES (rigs in time) and image_offset = R_cam_from_world^T t_cam_from_rig
    image_frame: Optional[np.ndarray] = None  # [I] int32
    image_offset: Optional[np.ndarray] = None  # [I,3] f64
    # Unknown cam_from_rig, gp.cc:354-368 RigUnknownBATAPairwiseDirectionError: image i of such a sensor has
    # image_sensor[i] >= 0 (its centre block), image_offset[i] = 0 and image_sensor_rot[i] = R_rig_from_world of its frame
    image_sensor: Optional[np.ndarray] = None  # [I] int32, -1 = no block
    image_sensor_rot: Optional[np.ndarray] = None  # [I,3,3] f64
    sensor_center: Optional[np.ndarray] = None  # [S,3] f64 in/out: camera centre in rig coordinates, -R_cfr^T t_cfr
    # Camera-to-camera constraints, gp.cc:167-210 (constraint_type != ONLY_POINTS, trivial frames): frame indices of the
    # valid pairs' two images and pair_dir = -R_cam2_from_world^T t_cam2_from_cam1
    pair_i: Optional[np.ndarray] = None  # [E] int32
    pair_j: Optional[np.ndarray] = None  # [E] int32
    pair_dir: Optional[np.ndarray] = None  # [E,3] f64
    # Order of the random draws (host int32 permutations, optional): the reference's container iteration order when frames /
    # tracks are numbered differently (include/gsfm.h, gsfm_gp_problem)
    cam_draw_order: Optional[np.ndarray] = None  # [N] int32
    pt_draw_order: Optional[np.ndarray] = None  # [P] int32

    @property
    def num_obs(self) -> int:
        return int(self.obs_cam.shape[0])

    @property
    def num_images(self) -> int:
        return 0 if self.image_frame is None else int(self.image_frame.shape[0])


@dataclass
class BaProblem:
    """Bundle-adjustment problem, trivial rigs (reference: ba.cc:115-190, 244-317)."""

    num_cams: int
    num_pts: int
    num_intr: int
    pt_offset: np.ndarray  # [P+1] int64, track-major observations
    obs_cam: np.ndarray  # [M] int32 (frame index)
    obs_xy: np.ndarray  # [M,2] f64 distorted pixel observation (image.features)
    cam_intr: np.ndarray  # [N] int32: intrinsics block used by each frame (camera_id)
    cam_q: np.ndarray  # [N,4] f64 (w,x,y,z) cam_from_world in/out
    cam_t: np.ndarray  # [N,3] f64 in/out
    pt_xyz: np.ndarray  # [P,3] f64 in/out
    intr_model: np.ndarray  # [K] int32 camera model id
    intr_params: np.ndarray  # [K,8] f64 in/out ([K,16] when a camera model has more than 8 parameters)
    fixed_cam: int = 0  # frame whose q and t are held constant (ba.cc:261-266); -1 = none
    gt_q: Optional[np.ndarray] = None
    gt_t: Optional[np.ndarray] = None
    gt_xyz: Optional[np.ndarray] = None
    gt_intr: Optional[np.ndarray] = None
    # Known (calibrated) rigs, ba.cc:147-160 RigReprojErrorConstantRigCostFunctor: when given, obs_cam indexes IMAGES,
    # cam_q / cam_t are the FRAMES' rig_from_world, cam_intr is unused (each image carries its own intrinsics block)
    image_frame: Optional[np.ndarray] = None  # [I] int32
    image_cam_from_rig: Optional[np.ndarray] = None  # [I,7] f64 (qw,qx,qy,qz,tx,ty,tz), identity for reference sensors
    image_intr: Optional[np.ndarray] = None  # [I] int32
    # Sensor blocks, ba.cc:161-179 RigReprojErrorCostFunctor: image i of a non-reference sensor takes its cam_from_rig from
    # sensor_cam_from_rig[image_sensor[i]] (in/out), optimised when BundleAdjusterOptions.optimize_rig_poses, else constant
    image_sensor: Optional[np.ndarray] = None  # [I] int32, -1 = reference sensor / constant entry of image_cam_from_rig
    sensor_cam_from_rig: Optional[np.ndarray] = None  # [S,7] f64 in/out

    @property
    def num_obs(self) -> int:
        return int(self.obs_cam.shape[0])

    @property
    def num_images(self) -> int:
        return 0 if self.image_frame is None else int(self.image_frame.shape[0])

    @property
    def num_sensors(self) -> int:
        return 0 if self.sensor_cam_from_rig is None else int(self.sensor_cam_from_rig.shape[0])

    def copy(self) -> "BaProblem":
        import copy

        return copy.deepcopy(self)
