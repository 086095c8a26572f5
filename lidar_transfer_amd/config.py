"""Sensor-model and approach YAML files, consumed unchanged.

The reference reads two kinds of YAML (``lidar_deform.py``):

* the *sensor model* -- ``config.yaml`` inside a dataset directory for the source sensor, ``--target`` for the
  target sensor (``lidar_deform.py:231-235, :289-295``): keys ``name, fov_up, fov_down, beams, angle_res_hor,
  fov_hor`` and optionally ``beam_angles``; the image width is ``W = int(fov_hor / angle_res_hor)``
  (``:264-277`` source, ``:302-315`` target);
* the *approach* file ``config/lidar_transfer.yaml`` (``:318-351``): adaption, number_of_scans, voxel_size,
  voxel_bounds, batch_interval, ignore / moving classes, labels and ``color_map``.

Host logic only: what the files say is turned into the arguments of the device path (``create_rays``,
``RaySet``, ``TSDFVolume``, the scan index list).  ``beam_angles`` is sorted in place like the reference does and
is ``None`` when absent; ``create_rays`` ignores it (``laserscan.py:1092-1119``), the projection uses it
(``laserscan.py:233-238``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


@dataclass
class SensorModel:
    """One scanner description (``lidar_deform.py:264-277``)."""
    name: str
    fov_up: float
    fov_down: float
    beams: int
    angle_res_hor: float
    fov_hor: float
    beam_angles: Optional[List[float]] = None
    raw: dict = field(default_factory=dict, repr=False)

    @property
    def H(self) -> int:
        return int(self.beams)

    @property
    def W(self) -> int:
        # W = int(fov_hor / angle_res_hor)  (lidar_deform.py:277, :309) -- float division, truncation
        return int(self.fov_hor / self.angle_res_hor)

    def as_tuple(self):
        """``(name, fov_up, fov_down, H, W, beam_angles)``"""
        return self.name, self.fov_up, self.fov_down, self.H, self.W, self.beam_angles

    def create_rays(self):
        """Host mirror of ``MultiSemLaserScan.create_rays(fov_up, fov_down, H, W)`` for this model:
        ``[H*W, 3]`` float32 (beam_angles are ignored there, as in the reference)."""
        from .laserscan import create_rays
        return create_rays(self.fov_up, self.fov_down, self.H, self.W)


def _load_yaml(path_or_dict):
    if isinstance(path_or_dict, dict):
        return path_or_dict
    import yaml
    with open(path_or_dict, "r") as f:
        return yaml.safe_load(f)


def load_sensor(path_or_dict) -> SensorModel:
    """Read a sensor YAML exactly as ``lidar_deform.py:264-277`` / ``:302-315`` do.

    Missing mandatory keys raise ``KeyError`` (the reference indexes the dict directly); a missing
    ``beam_angles`` means "equidistant angles" (``None``); present ones are sorted ascending."""
    cfg = _load_yaml(path_or_dict)
    name = cfg["name"]
    fov_up = cfg["fov_up"]
    fov_down = cfg["fov_down"]
    beams = cfg["beams"]
    angle_res_hor = cfg["angle_res_hor"]
    fov_hor = cfg["fov_hor"]
    try:
        beam_angles = list(cfg["beam_angles"])
        beam_angles.sort()
    except Exception:
        beam_angles = None
    return SensorModel(name, fov_up, fov_down, beams, angle_res_hor, fov_hor, beam_angles, raw=cfg)


@dataclass
class Approach:
    """``config/lidar_transfer.yaml`` (``lidar_deform.py:318-351``)."""
    adaption: str
    preserve_float: bool
    voxel_size: float
    voxel_bounds: np.ndarray          # [3, 2] (xmin xmax / ymin ymax / zmin zmax), lidar_deform.py:347-350
    number_of_scans: int
    ignore: List[int]
    moving: List[int]
    transformation: List[float]
    batch_interval: int               # default 1 when absent (lidar_deform.py:352-355)
    color_map: Dict[int, List[int]]   # bgr
    labels: Dict[int, str]

    def color_lut(self) -> np.ndarray:
        """``SemLaserScan.__init__``'s look-up table (``laserscan.py:547-555``): ``[max(key) + 1 + 100, 3]`` float32,
        ``color_map`` values / 255."""
        max_key = 0
        for key in self.color_map:
            if key + 1 > max_key:
                max_key = key + 1
        lut = np.zeros((max_key + 100, 3), dtype=np.float32)
        for key, value in self.color_map.items():
            lut[key] = np.array(value, np.float32) / 255.0
        return lut

    def scan_indices(self, n_scan_files: int, offset: int = 0):
        """Scan list of the batch loop (``lidar_deform.py:385-390, :457-459``)."""
        from .dist import scan_indices
        return scan_indices(n_scan_files, self.number_of_scans, offset, self.batch_interval)


def load_approach(path_or_dict) -> Approach:
    cfg = _load_yaml(path_or_dict)
    vb = np.array(cfg["voxel_bounds"])
    try:
        vb = vb.reshape(3, 2)
    except Exception:
        pass
    return Approach(adaption=cfg["adaption"], preserve_float=cfg["preserve_float"], voxel_size=cfg["voxel_size"],
                    voxel_bounds=vb, number_of_scans=cfg["number_of_scans"], ignore=list(cfg["ignore"]),
                    moving=list(cfg["moving"]), transformation=list(cfg["transformation"]),
                    batch_interval=cfg.get("batch_interval", 1), color_map=dict(cfg["color_map"]),
                    labels=dict(cfg.get("labels", {})))
