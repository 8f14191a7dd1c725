"""Host-side mirror of the reference's scene containers (glomap/scene/*), reduced to the fields
the three estimators read or write.  Containers are plain dicts keyed by id, like the
reference's std::unordered_map.  Trivial rigs (one image per frame) need nothing beyond Frame / Image; multi-camera rigs
add `Rig` (colmap::Rig: reference sensor + optional sensor_from_rig per other sensor) and the frame's rig_id / image_ids;
gravity-aligned frames carry `gravity` (GravityInfo, scene/frame.h:11-27).

  ImagePair  glomap/scene/image_pair.h:13-57      ViewGraph  glomap/scene/view_graph.h:12-35
  Image      glomap/scene/image.h:10-53           Frame      glomap/scene/frame.h:29-42
  Track      glomap/scene/track.h:10-27           Camera     glomap/scene/camera.h:12-26
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np


@dataclass
class Rigid3d:
    """cam_from_world: x_cam = R(rotation) x_world + translation.  rotation = (w,x,y,z)."""

    rotation: np.ndarray = field(default_factory=lambda: np.array([1.0, 0.0, 0.0, 0.0]))
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3))


@dataclass
class ImagePair:
    image_id1: int
    image_id2: int
    cam2_from_cam1: Rigid3d = field(default_factory=Rigid3d)
    is_valid: bool = True
    weight: float = -1.0
    inliers: List[int] = field(default_factory=list)
    num_inliers: Optional[int] = None  # stand-in for inliers.size() when the list itself is not kept

    def inlier_count(self) -> int:
        return self.num_inliers if self.num_inliers is not None else len(self.inliers)


@dataclass
class ViewGraph:
    image_pairs: Dict[Tuple[int, int], ImagePair] = field(default_factory=dict)


@dataclass
class Camera:
    camera_id: int
    model_id: int
    params: np.ndarray
    has_prior_focal_length: bool = True


def get_align_rot(gravity: np.ndarray) -> np.ndarray:
    """GetAlignRot (math/gravity.cc:11-24): the second column is the gravity direction, the other two the last two
    columns of the Householder Q of that vector (Eigen and LAPACK build the same reflector), the third flipped if needed
    for a right-handed basis."""
    v = np.asarray(gravity, dtype=np.float64)
    v = v / np.linalg.norm(v)
    Q, _ = np.linalg.qr(v.reshape(3, 1), mode="complete")
    R = np.stack([Q[:, 1], v, Q[:, 2]], axis=1)
    if np.linalg.det(R) < 0:
        R[:, 2] = -R[:, 2]
    return R


@dataclass
class Rig:
    """colmap::Rig as the estimators use it: camera sensors only, keyed by camera_id; None = no sensor_from_rig yet."""

    rig_id: int
    ref_camera_id: int
    sensors: Dict[int, Optional[Rigid3d]] = field(default_factory=dict)  # non-reference sensors

    def IsRefSensor(self, camera_id: int) -> bool:
        return camera_id == self.ref_camera_id

    def MaybeSensorFromRig(self, camera_id: int) -> Optional[Rigid3d]:
        return self.sensors.get(camera_id)

    def SetSensorFromRig(self, camera_id: int, cam_from_rig: Rigid3d) -> None:
        self.sensors[camera_id] = cam_from_rig

    def ResetSensorFromRig(self, camera_id: int) -> None:
        self.sensors[camera_id] = None

    def NonRefSensors(self):
        return sorted(self.sensors.items())  # std::map order


@dataclass
class Frame:
    frame_id: int
    rig_from_world: Rigid3d = field(default_factory=Rigid3d)
    is_registered: bool = True
    rig_id: Optional[int] = None  # None: a trivial frame of its own
    image_ids: List[int] = field(default_factory=list)
    gravity: Optional[np.ndarray] = None  # gravity direction in the rig frame (GravityInfo::SetGravity)

    def HasGravity(self) -> bool:
        return self.gravity is not None

    def GetRAlign(self) -> np.ndarray:
        return get_align_rot(self.gravity) if self.gravity is not None else np.eye(3)


@dataclass
class Image:
    image_id: int
    camera_id: int
    frame_id: int
    features: Optional[np.ndarray] = None  # [F,2] distorted pixels (image.h:29)
    features_undist: Optional[np.ndarray] = None  # [F,3] unit rays (image.h:31)
    is_registered: bool = True

    def IsRegistered(self) -> bool:
        return self.is_registered


@dataclass
class Track:
    track_id: int
    xyz: np.ndarray = field(default_factory=lambda: np.zeros(3))
    observations: List[Tuple[int, int]] = field(default_factory=list)  # (image_id, feature_id)
    is_initialized: bool = False
