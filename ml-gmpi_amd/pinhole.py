"""Pinhole camera: per-pixel unit ray directions and world-space rays for a camera-to-world matrix.

Host-side mirror of gmpi/core/camera.py:13-211 (`Camera`) and gmpi/utils/cam_utils.py:16-22
(`gen_cam`).  The bit pattern of `ray_dir` is part of the parity contract:
directions are built in float64 numpy (K^-1 [x+.5, y+.5, 1], normalised), cast once to float32 and
rotated by a float32 `torch.matmul` on the renderer's device -- the same library calls, in the same
precision, as the reference makes.
"""
from typing import Tuple, Union

import numpy as np
import torch


class Camera:
    """height x width pinhole with 3x3 intrinsics; rays leave from pixel centres when `ray_from_pix_center`."""

    def __init__(self, height: int = 480, width: int = 640, intrinsics: np.ndarray = None,
                 ray_from_pix_center: bool = False):
        assert intrinsics.ndim == 2 and intrinsics.shape == (3, 3), \
            "[Camera] Expecting a 3x3 intrinsics matrix, but instead got {}".format(intrinsics.shape)
        self._h, self._w, self._K = height, width, intrinsics
        self._ray_from_pix_center = ray_from_pix_center
        self._cache = {}

    @property
    def intrinsic_matrix(self):
        return self._K

    @property
    def height(self):
        return self._h

    @property
    def width(self):
        return self._w

    def __repr__(self):
        return f"Camera: height={self.height}, width={self.width}, intrinsics=\n{self.intrinsic_matrix}"

    # -- unit directions in the camera frame ---------------------------------------------------------
    def _unit_dirs_np(self, border_only: bool) -> np.ndarray:
        """[3, H*W] float64 (or [3,4] for the four frustum corners), cached."""
        key = ("np", border_only)
        if key not in self._cache:
            if border_only:
                xx, yy = np.meshgrid(np.array([0, self._w]), np.array([0, self._h]), indexing="xy")
            else:
                xx, yy = np.meshgrid(range(int(self._w)), range(int(self._h)), indexing="xy")
                if self._ray_from_pix_center:
                    xx = xx + 0.5
                    yy = yy + 0.5
            pix = np.stack([xx, yy, np.ones(xx.shape)])
            back = np.matmul(np.linalg.inv(self._K), pix.reshape(3, -1))
            back = back.reshape((3,) + xx.shape)
            dirs = back / np.linalg.norm(back, axis=0)
            self._cache[key] = dirs.reshape(3, -1)
        return self._cache[key]

    @property
    def ray_dir_np(self) -> np.ndarray:
        return self._unit_dirs_np(False)

    @property
    def ray_dir_border_np(self) -> np.ndarray:
        return self._unit_dirs_np(True)

    def unit_dirs(self, device, border_only: bool = False) -> torch.Tensor:
        """[3, H*W] float32 on `device`, cached per device."""
        key = ("t", border_only, str(device))
        if key not in self._cache:
            self._cache[key] = torch.FloatTensor(self._unit_dirs_np(border_only)).to(device)
        return self._cache[key]

    # -- world-space rays ---------------------------------------------------------------------------
    def generate_rays(self, tf_c2w: Union[np.ndarray, torch.Tensor], border_only: bool = False) -> Tuple:
        """(ray_dir [3,H,W], eye_pos [3], z_dir [3]) for a 4x4 camera-to-world matrix (numpy or torch)."""
        shape = (3, 2, 2) if border_only else (3, self._h, self._w)
        rot = tf_c2w[:3, :3]
        eye = tf_c2w[:3, 3]
        if isinstance(tf_c2w, np.ndarray):
            rays = (rot @ self._unit_dirs_np(border_only)).reshape(shape)
        elif isinstance(tf_c2w, torch.Tensor):
            rays = torch.matmul(rot, self.unit_dirs(tf_c2w.device, border_only)).reshape(shape)
        else:
            raise ValueError
        return rays, eye, rot[:, 2]


def gen_cam(*, h, w, f, ray_from_pix_center):
    """Camera with principal point at the image centre (w/2, h/2) and focal length f (pixels)."""
    K = np.array([[f, 0.0, w / 2], [0.0, f, h / 2], [0.0, 0.0, 1.0]])
    return Camera(height=h, width=w, intrinsics=K, ray_from_pix_center=ray_from_pix_center)
