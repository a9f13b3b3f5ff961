"""Plane depths and per-plane spatial extents (d, h, w) of the multiplane volume.

Host-side, init-time mirror of gmpi/utils/mpi_utils.py:21-53 (`sample_distance`), :621-649
(`compute_intersection_between_cam_frustum_and_plane`) and :652-917
(`compute_plane_dhws_given_cam_pose_spatial_range[_confined]`), as used by
`MPIRenderer.compute_mpi_spatial_volume` (gmpi/core/mpi_renderer.py:105-152).

The reference walks 100x100+1 camera poses in a Python loop (seconds per renderer construction,
SURVEY.md section 8 f-4).  Here all poses are evaluated in ONE batch: same float32/float64
operations per pose, so the resulting `dhw` table is bit-identical (checked against fixtures made
by the reference in tests/test_host_geometry.py), in tens of milliseconds.

`dhw` feeds the in-kernel coordinate chain, so it is part of the parity contract.
"""
from typing import List, Tuple

import numpy as np
import torch

from .pinhole import Camera
from .poses import gen_sphere_path

_GRID = 100  # poses per axis of the heuristic pose grid


def sample_distance(dmin: float, dmax: float, num_samples: int, method: str, **kwargs) -> np.ndarray:
    """`num_samples` plane distances in [dmin, dmax], nearest first, float32."""
    assert 0 < dmin <= dmax
    assert 1 <= num_samples < 9999
    if method == "uniform":
        d = list(np.linspace(dmin, dmax, num=num_samples))
    elif method == "log-uniform":
        d = np.exp(np.linspace(np.log(dmin), np.log(dmax), num=num_samples)).tolist()
    elif method == "sqrt":
        d = [t ** 2 for t in np.linspace(dmin ** 0.5, dmax ** 0.5, num=num_samples)]
    elif method == "squared":
        d = [np.sqrt(t) for t in np.linspace(dmin ** 2, dmax ** 2, num=num_samples)]
    elif method == "inverse":
        d = [1 / t for t in np.linspace(1 / dmax, 1 / dmin, num=num_samples)][::-1]
    else:
        raise ValueError(method)
    return np.array(d, dtype=np.float32)


def frustum_footprints(camera: Camera, c2w: np.ndarray, z_plane: torch.Tensor) -> Tuple[np.ndarray, ...]:
    """Per-pose bounding box (min_x, max_x, min_y, max_y; float32 [P]) of the camera frustum on the plane z=z_plane.

    c2w: [P,4,4] float64.  The four frustum-corner rays are rotated in float64, cast to float32, and
    intersected with the plane in float32 (scale = (z_plane - cam_z) / ray_z; xyz = cam + ray*scale).
    """
    corners = camera.ray_dir_border_np  # [3,4] float64
    P = c2w.shape[0]
    rays = np.empty((P, 3, 4), dtype=np.float64)
    for i in range(P):  # one 3x3 @ 3x4 product per pose, as the reference does it (identical BLAS call)
        rays[i] = c2w[i, :3, :3] @ corners
    ray = torch.from_numpy(rays).float()                      # [P,3,4]
    cam = torch.from_numpy(np.ascontiguousarray(c2w[:, :3, 3])).float()  # [P,3]
    z_diff = (z_plane - cam[:, 2:3]).view(P, 1, 1)
    scale = z_diff / ray[:, 2:3, :]
    xyz = cam.view(P, 3, 1) + ray * scale
    x, y = xyz[:, 0, :], xyz[:, 1, :]
    return (x.min(dim=1)[0].numpy(), x.max(dim=1)[0].numpy(), y.min(dim=1)[0].numpy(), y.max(dim=1)[0].numpy())


def compute_plane_dhws(camera: Camera, sphere_center: np.ndarray, sphere_r, cam_horizontal_min: float,
                       cam_horizontal_max: float, cam_vertical_min: float, cam_vertical_max: float,
                       cam_pose_n_truncated_stds, plane_zs: torch.Tensor, enlarge_factor: float = 1.0,
                       confined: bool = True, consume_rng_like_reference: bool = True):
    """(dhws [D,3] float64 array holding float32 values, tex_expand_ratio).

    Last plane: symmetric bounding box of every pose's frustum on it, times `enlarge_factor`.
    Other planes: the footprint of the MID pose on the last plane -- constant (`confined`, the
    reference's `use_confined_volume`) or scaled by z / z_last.
    """
    h_mid = (cam_horizontal_min + cam_horizontal_max) / 2
    v_mid = (cam_vertical_min + cam_vertical_max) / 2
    hs = np.linspace(cam_horizontal_min, cam_horizontal_max, _GRID)
    vs = np.linspace(cam_vertical_min, cam_vertical_max, _GRID)
    yaw = np.concatenate([np.repeat(hs, _GRID), [h_mid]])   # horizontal outer loop, vertical inner, mid pose last
    pitch = np.concatenate([np.tile(vs, _GRID), [v_mid]])
    P = yaw.shape[0]
    if consume_rng_like_reference:
        # The reference draws torch.rand((1,1)) twice per pose (std is 0, so the draws do not change
        # the angles); advance the global RNG identically so later seeded pose sampling matches.
        for _ in range(2 * P):
            torch.rand((1, 1))
    # (rand-0.5)*2*n*0 + mean == float32(mean)
    yaws = torch.from_numpy(yaw).to(torch.float32).reshape(P, 1)
    pitches = torch.from_numpy(pitch).to(torch.float32).reshape(P, 1)
    c2w, _, _ = gen_sphere_path(P, sphere_center, sphere_r, given_yaws=yaws, given_pitches=pitches)
    mnx, mxx, mny, mxy = frustum_footprints(camera, c2w, plane_zs[-1])

    # mid pose (last entry): base/confined sizes, all in float32 like the numpy scalars of the reference
    f32 = np.float32
    base_spatial_size = min(mxx[-1] - mnx[-1], mxy[-1] - mny[-1])
    conf_h = f32(2) * np.max([np.abs(mny[-1]), np.abs(mxy[-1])])
    conf_w = f32(2) * np.max([np.abs(mnx[-1]), np.abs(mxx[-1])])

    lo_x, hi_x, lo_y, hi_y = np.min(mnx), np.max(mxx), np.min(mny), np.max(mxy)
    bound = np.max(np.abs([lo_x, hi_x, lo_y, hi_y]))
    assert bound <= 5.0, (
        f"You have MPI's plane whose boundary value is up to {bound}. "
        f"This usually means the camera poses's range is too big, which will cause problems for MPI representation. "
        f"Please reduce h_stddev or v_stddev in curriculums.py or cam_pose_n_truncated_stds in config file."
    )
    # symmetric planes (+X right, +Y down); float32 arithmetic (numpy>=2 scalar promotion)
    last_h = f32(f32(2) * np.max([np.abs(lo_y), np.abs(hi_y)])) * f32(enlarge_factor)
    last_w = f32(f32(2) * np.max([np.abs(lo_x), np.abs(hi_x)])) * f32(enlarge_factor)

    z_last = plane_zs[-1]
    rows: List[List[float]] = []
    for i in range(len(plane_zs) - 1):
        z = plane_zs[i]
        if confined:
            rows.append([float(z), float(conf_h), float(conf_w)])
        else:
            rows.append([float(z), float(conf_h * z / z_last), float(conf_w * z / z_last)])
    rows.append([float(z_last), float(last_h), float(last_w)])
    dhws = np.array(rows)
    tex_expand_ratio = np.max(dhws[:, 1:] / base_spatial_size)
    return dhws, tex_expand_ratio
