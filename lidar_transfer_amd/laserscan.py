"""Host-side mirror of the scan-model functions that sit on the hot path
(reference: auxiliary/laserscan.py).  Only what the path needs is here."""
from __future__ import annotations

import numpy as np


def create_rays(fov_up, fov_down, H, W):
    """Unit ray direction per (beam, azimuth) cell, ``float32 [H*W, 3]``, row-major ``h*W + w``.

    Restatement of ``MultiSemLaserScan.create_rays`` (auxiliary/laserscan.py:1092-1119), quirks
    included: ``linspace(0, 360, W)`` contains both end points, so column 0 and column W-1 are
    the same direction; ``beam_angles`` are ignored; float64 trigonometry, cast to float32 last.
    """
    yaw = np.linspace(0, 360, W) + 180
    yaw[yaw > 360] -= 360
    yaw = yaw / 180. * np.pi
    pitch = np.pi / 2 - np.linspace(fov_up, fov_down, H) / 180. * np.pi
    sp, cp = np.sin(pitch), np.cos(pitch)
    beams = np.empty((H, W, 3), dtype=np.float64)
    beams[:, :, 0] = sp[:, None] * np.cos(-yaw)[None, :]
    beams[:, :, 1] = sp[:, None] * np.sin(-yaw)[None, :]
    beams[:, :, 2] = cp[:, None] * np.ones(W)[None, :]
    return np.ascontiguousarray(beams.reshape(H * W, 3).astype(np.float32))
