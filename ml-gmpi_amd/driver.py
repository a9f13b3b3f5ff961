"""Per-view batch driver: many camera views of one or several MPIs, sharded over the GPUs of a node.

Replaces the reference's host loops that call `render` once per view and copy every frame back
(gmpi/eval/vis/render_video.py:95-130: one render + `.cpu()` per angle; prepare_fake_data.py:58-86;
fid_evaluation.py:86-133: `img_counter = rank; img_counter += world_size`).  Here

  * views are rendered in batches -- one launch per batch, the RGBA volume is indexed, never
    replicated (`views_per_mpi`), and nothing is copied to the host inside the loop;
  * the uint8 / depth-normalisation epilogue of render_video.py:118-126 runs on the device
    (`gmpi_frames_to_uint8_launch`);
  * across ranks the views are partitioned exactly like the reference's rank-strided counter, every
    rank renders its shard independently (no data-path collective), and ONE all_gather of the
    finished frames (RCCL over xGMI when the backend is "nccl") assembles the sequence.
"""
import ctypes
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib


def shard_views(n_views: int, rank: int, world_size: int, mode: str = "strided") -> List[int]:
    """View indices rendered by `rank`.  "strided" = the reference's counter (rank, rank+W, ...);
    "block" = contiguous blocks (better when consecutive views share an MPI)."""
    assert 0 <= rank < world_size
    if mode == "strided":
        return list(range(rank, n_views, world_size))
    if mode == "block":
        per = (n_views + world_size - 1) // world_size
        return list(range(min(rank * per, n_views), min((rank + 1) * per, n_views)))
    raise ValueError(mode)


def frames_to_uint8(rgb_pm1: torch.Tensor, depth: Optional[torch.Tensor], near: float, far: float, out=None):
    """(img8 [N,H,W,3] uint8, dep8 [N,H,W,1] uint8 or None) on the device -- render_video.py:118-126.  `out` = (img8, dep8 or None): contiguous
    device tensors to write into (e.g. slices of a path-long buffer)."""
    if not rgb_pm1.is_cuda:
        raise _lib.GmpiError("frames_to_uint8 needs device tensors (no CPU path)")
    lib = _lib.load_library()
    rgb_pm1 = rgb_pm1.contiguous()
    N, _, H, W = rgb_pm1.shape
    img8 = out[0] if out is not None else torch.empty((N, H, W, 3), dtype=torch.uint8, device=rgb_pm1.device)
    assert img8.is_contiguous() and tuple(img8.shape) == (N, H, W, 3) and img8.dtype is torch.uint8
    dep8 = None
    if depth is not None:
        depth = depth.contiguous()
        dep8 = out[1] if out is not None else torch.empty((N, H, W, 1), dtype=torch.uint8, device=rgb_pm1.device)
        assert dep8.is_contiguous() and tuple(dep8.shape) == (N, H, W, 1) and dep8.dtype is torch.uint8
    with torch.cuda.device(rgb_pm1.device):
        _lib.check(lib.gmpi_frames_to_uint8_launch(
            rgb_pm1.data_ptr(), depth.data_ptr() if depth is not None else None, N, H, W, float(near), float(far),
            img8.data_ptr(), dep8.data_ptr() if dep8 is not None else None,
            torch.cuda.current_stream(rgb_pm1.device).cuda_stream), "gmpi_frames_to_uint8_launch")
    return img8, dep8


def gather_frames(local: torch.Tensor, indices: Sequence[int], n_views: int, group=None) -> torch.Tensor:
    """all_gather of per-rank frame stacks [n_local, ...] into [n_views, ...] ordered by view index.

    Shards may be ragged (n_views not divisible by the world size): every rank pads to the largest
    shard so that a single fixed-size all_gather suffices; the index lists travel with the frames.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        out = torch.empty((n_views,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(list(indices), dtype=torch.long, device=local.device)] = local
        return out
    world = dist.get_world_size(group)
    per = (n_views + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    idx = torch.full((per,), -1, dtype=torch.int64, device=local.device)
    idx[: len(indices)] = torch.as_tensor(list(indices), dtype=torch.int64, device=local.device)
    all_frames = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    all_idx = torch.empty((world * per,), dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(all_frames, pad, group=group)
    dist.all_gather_into_tensor(all_idx, idx, group=group)
    keep = all_idx >= 0
    out = torch.empty((n_views,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    out[all_idx[keep]] = all_frames[keep]
    return out


def render_views_sharded(render_fn: Callable[[List[int]], torch.Tensor], n_views: int, rank: int = 0,
                         world_size: int = 1, mode: str = "strided", gather: bool = True, group=None):
    """Each rank renders `shard_views(...)` through `render_fn(indices) -> [n_local, C, H, W]`; with `gather`
    the full [n_views, C, H, W] stack is returned on every rank, else (local_frames, indices)."""
    indices = shard_views(n_views, rank, world_size, mode)
    local = render_fn(indices)
    assert local.shape[0] == len(indices), (local.shape, len(indices))
    if not gather:
        return local, indices
    return gather_frames(local, indices, n_views, group=group)


def dump_frames(save_dir: str, task: str, start_index: int, img8, angles, depth=None, n_view_per_z: int = 1) -> List[str]:
    """Writes rendered frames in the on-disk layout of the reference's dataset dumps (prepare_fake_data.py:184-190, 226-258):

        <save_dir>/<task>/rgb/<idx:06d>.png      uint8 RGB            (n_view_per_z == 1)
        <save_dir>/<task>/angle/<idx:06d>.npy    [2] = (pitch, yaw)
        <save_dir>/<task>/depth/<idx:06d>.npy    [H, W] float32       (only with `depth`)
        ... and `<idx:06d>_<j>.*` for view j of seed idx when n_view_per_z > 1 (the "consistency" task).

    img8 [N,H,W,3] uint8 (host or device; e.g. `frames_to_uint8` / `ViewBatchDriver.render_path(..., to_uint8=True)["img8"]`), angles [N,2],
    depth [N,H,W] or [N,1,H,W] float32.  N = seeds x n_view_per_z, seed-major; the first seed gets index `start_index`.  Returns the PNG paths."""
    import os

    import numpy as np
    from PIL import Image

    def host(t):
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)

    img8, angles = host(img8), host(angles)
    assert img8.dtype == np.uint8 and img8.ndim == 4 and img8.shape[-1] == 3, (img8.dtype, img8.shape)
    assert img8.shape[0] % n_view_per_z == 0 and angles.shape[0] == img8.shape[0]
    if depth is not None:
        depth = host(depth).astype(np.float32).reshape(img8.shape[0], img8.shape[1], img8.shape[2])
    dirs = {k: os.path.join(save_dir, task, k) for k in ("rgb", "depth", "angle")}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    paths = []
    for n in range(img8.shape[0]):
        seed, j = divmod(n, n_view_per_z)
        name = f"{start_index + seed:06d}" if n_view_per_z == 1 else f"{start_index + seed:06d}_{j}"
        Image.fromarray(img8[n]).save(os.path.join(dirs["rgb"], name + ".png"))
        with open(os.path.join(dirs["angle"], name + ".npy"), "wb") as f:
            np.save(f, angles[n])
        if depth is not None:
            with open(os.path.join(dirs["depth"], name + ".npy"), "wb") as f:
                np.save(f, depth[n])
        paths.append(os.path.join(dirs["rgb"], name + ".png"))
    return paths


class ViewBatchDriver:
    """Renders a camera path / a batch of seeds with an `MPIRenderer`, `batch` views per launch, device-resident."""

    def __init__(self, renderer, batch: int = 8):
        self.renderer = renderer
        self.batch = int(batch)
        self._host = {}
        self._copy_stream = None   # a second HIP stream: frames travel to the host while the next batch renders (render_path(to_host=True))

    def _pinned(self, slot, shape, dtype):
        key = (slot, tuple(shape), dtype)
        buf = self._host.get(key)
        if buf is None:
            while len(self._host) >= 8:
                self._host.pop(next(iter(self._host)))
            buf = self._host[key] = torch.empty(shape, dtype=dtype, pin_memory=True)
        return buf

    def to_host(self, *tensors: torch.Tensor):
        """Device tensors -> PINNED host tensors kept by the driver, one asynchronous copy each and one synchronisation.  A fresh pageable
        `.cpu()` tensor per pass pays the page faults of its allocation on top of the copy (64 uint8 frames of 512^2: 5-9 k frames/s from
        run to run; 9 k steadily this way).  The buffers are reused by the next call with the same shapes: consume (or copy) them first."""
        outs = []
        for i, t in enumerate(tensors):
            buf = self._pinned(i, t.shape, t.dtype) if t.is_cuda else torch.empty(t.shape, dtype=t.dtype)
            buf.copy_(t, non_blocking=True)
            outs.append(buf)
        if tensors and tensors[0].is_cuda:
            torch.cuda.current_stream(tensors[0].device).synchronize()
        return tuple(outs)

    @torch.no_grad()
    def render_path(self, mpi_rgbas: torch.Tensor, render_size: int, yaws: Sequence[float],
                    pitches: Sequence[float], indices: Optional[Sequence[int]] = None, to_uint8: bool = False,
                    depth_range=None, want_transmittance: bool = False, to_host: bool = False):
        """Views `indices` (default all) of ONE MPI [1,D,4,Ht,Wt] along (yaws[i], pitches[i]) -- the loop of
        render_video.py:95-130 (h_mean/v_mean = angle, std 0) as batched launches.

        Returns dict(rgb [n,3,H,W] in [-1,1], depth [n,1,H,W][, T][, img8, dep8]) on the device.
        `to_host=True` (with `to_uint8`; round 6): the uint8 epilogue runs per batch and every batch's frames travel to PINNED host buffers on a second
        HIP stream while the next batch renders -- the copy of the whole path (64 frames of 512^2: 67 MB) then hides behind the render launches instead
        of following them; adds `img8_host`, `dep8_host` (buffers of the driver, reused by the next call of the same shape: consume or copy them).
        """
        r = self.renderer
        assert mpi_rgbas.shape[0] == 1, "render_path draws many views of one MPI"
        idx = list(range(len(yaws))) if indices is None else list(indices)
        dev = mpi_rgbas.device
        n = len(idx)
        rgb = torch.empty((n, 3, render_size, render_size), dtype=torch.float32, device=dev)
        dep = torch.empty((n, 1, render_size, render_size), dtype=torch.float32, device=dev)
        T = torch.empty((n, 1, render_size, render_size), dtype=torch.float32, device=dev) if want_transmittance else None
        status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        if render_size != r.render_h or render_size != r.render_w:
            r.set_cam(r.cam_fov, render_size, render_size)
        dhw = r._dhw_on_device()
        pipe = to_host and to_uint8 and mpi_rgbas.is_cuda
        if pipe:
            near, far = depth_range if depth_range is not None else (r.plane_min_d, r.plane_max_d)
            img8 = torch.empty((n, render_size, render_size, 3), dtype=torch.uint8, device=dev)
            dep8 = torch.empty((n, render_size, render_size, 1), dtype=torch.uint8, device=dev)
            img8_h, dep8_h = self._pinned("path_img8", img8.shape, torch.uint8), self._pinned("path_dep8", dep8.shape, torch.uint8)
            if self._copy_stream is None or self._copy_stream.device != dev:
                self._copy_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self._copy_stream.wait_stream(main)   # (the buffers' previous users are done)
        # Poses and rays of a GROUP of batches in one go (round 6): one pass of the host's pose arithmetic, one copy, one ray kernel for up to
        # ~256 MB of rays -- per batch only the render launch is left on the host (the per-batch pose call was what bounded the loop once the
        # copies had moved off the critical path).  The poses are a pure function of the angles (std 0); the torch RNG advances as for ONE
        # `sample_cam_poses` call per group (the reference's per-view loop advances it once per view: render_video.py:100-113).
        group = max(self.batch, (int(256e6 // (12 * render_size * render_size)) // self.batch) * self.batch)
        ray_g = eye_g = zd_g = None
        for s in range(0, n, self.batch):
            chunk = idx[s:s + self.batch]
            if s % group == 0:
                members = idx[s:s + group]
                gy = torch.tensor([[float(yaws[i])] for i in members], dtype=torch.float32)
                gp = torch.tensor([[float(pitches[i])] for i in members], dtype=torch.float32)
                _, _, c2w, rays, eyes, zdirs = r.sample_cam_poses(len(members), 0.0, 0.0, 0.0, 0.0, False, given_yaws=gy, given_pitches=gp)
                if r._batched_cam is not None and rays is r._batched_cam[0]:
                    ray_g, eye_g, zd_g = r._batched_cam[1:]
                else:
                    ray_g, eye_g, zd_g = torch.cat(rays), torch.cat(eyes), torch.cat(zdirs)
                r._batched_cam = None
            g0 = s % group
            ray_t, eye_t, zd_t = ray_g[g0:g0 + len(chunk)], eye_g[g0:g0 + len(chunk)], zd_g[g0:g0 + len(chunk)]
            out = dict(color=rgb[s:s + len(chunk)], depth=dep[s:s + len(chunk)])
            if T is not None:
                out["T"] = T[s:s + len(chunk)]
            r.mpi.render_views(mpi_rgbas, dhw, ray_t, eye_t, zd_t,
                               views_per_mpi=len(chunk), check_last_plane=True, out_pm1=True,
                               want_transmittance=want_transmittance, status=status, defer_status=True, out=out)
            if pipe:   # this batch's frames: uint8 on the device, then to the host behind an event -- next to the next batch's render
                e = s + len(chunk)
                frames_to_uint8(rgb[s:e], dep[s:e], near, far, out=(img8[s:e], dep8[s:e]))
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(self._copy_stream):
                    self._copy_stream.wait_event(ev)
                    img8_h[s:e].copy_(img8[s:e], non_blocking=True)
                    dep8_h[s:e].copy_(dep8[s:e], non_blocking=True)
        r.mpi.raise_on_status(status)  # one host sync for the whole path
        res = dict(rgb=rgb, depth=dep, T=T)
        if pipe:
            self._copy_stream.synchronize()
            res.update(img8=img8, dep8=dep8, img8_host=img8_h, dep8_host=dep8_h)
        elif to_uint8:
            near, far = depth_range if depth_range is not None else (r.plane_min_d, r.plane_max_d)
            res["img8"], res["dep8"] = frames_to_uint8(rgb, dep, near, far)
        return res

    @torch.no_grad()
    def render_seeds(self, mpi_rgbas: torch.Tensor, render_size: int, views_per_mpi: int = 1, **render_kwargs):
        """Random-pose renders of a stack of MPIs [B,D,4,Ht,Wt] (prepare_fake_data.py:58-66 /
        fid_evaluation.py:116 pattern), `batch` MPIs per launch.  Returns the renderer's 4-tuple, concatenated."""
        from .hip_mpi import flush_status
        outs = []
        render_kwargs.setdefault("defer_status", "lag")   # (no frame leaves this method before every launch's assertions have been looked at)
        for s in range(0, mpi_rgbas.shape[0], self.batch):
            outs.append(self.renderer.render(mpi_rgbas[s:s + self.batch], render_size, render_size,
                                             views_per_mpi=views_per_mpi, **render_kwargs))
        if render_kwargs["defer_status"] == "lag":
            flush_status()
        return tuple(torch.cat([o[i] for o in outs], 0) for i in range(len(outs[0])))
