"""Host-side mirror of the reference's scene containers (glomap/scene/*), reduced to the fields
the three estimators read or write.  Containers are plain dicts keyed by id, like the
reference's std::unordered_map; trivial rigs only (one image per frame).

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


@dataclass
class Frame:
    frame_id: int
    rig_from_world: Rigid3d = field(default_factory=Rigid3d)
    is_registered: bool = True


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
