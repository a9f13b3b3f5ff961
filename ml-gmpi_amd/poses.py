"""Camera poses on the viewing sphere -> camera-to-world matrices in the MPI frame.

Host-side mirror of the pose chain `MPIRenderer.sample_cam_poses` relies on
(reference: gmpi/utils/cam_utils.py:481-568 `sample_camera_positions_sphere`, :571-622
`create_cam2sphere_sys_matrix`, :687-731 `create_sphere2world_sys_matrix_for_coord`, :734-821
`gen_sphere_path`; gmpi/utils/torch_utils.py:51-76 `truncated_normal`, :87-91 `normalize_vecs`).

The numbers produced here are part of the parity contract (a 1-ulp change of c2w moves every
sampling position), so the numerical recipe is kept operation-for-operation: float32 torch ops for
the sphere position and the look-at frame, float64 numpy for the change of frame, and the same
torch RNG calls in the same order (so a seeded run draws the same poses as the reference).
tests/test_host_geometry.py checks bit-equality against fixtures produced by the reference.

Frames.  Sphere frame: origin at the sphere centre, +X backward, +Y right, +Z up; yaw is measured
from +X in the XY plane, pitch from +X in the XZ plane; (yaw, pitch) = (0, 0) faces the MPI.
MPI/world frame: +X right, +Y down, +Z forward.
"""
from typing import Optional, Tuple

import threading

import numpy as np
import torch
from scipy.spatial.transform import Rotation

_CPU = torch.device("cpu")


class host_math:
    """Context for the tiny CPU tensor ops of pose sampling (a few [B,1] tensors): run them on one thread.

    On a 128-core host some of these ops (`max(dim)`, `gather`, advanced indexing) open an OpenMP region over the whole
    intra-op pool for 8 elements; waking 128 sleeping threads costs milliseconds per op and made `render()` and the
    shading augmentation 5-10x slower than their kernels (profiles/r01_aux_kernels.txt).  The intra-op thread count is
    process-global, so it is restored on exit; nested use is a no-op."""
    _depth = 0
    _lock = threading.Lock()

    def __enter__(self):
        with host_math._lock:
            host_math._depth += 1
            if host_math._depth == 1:
                host_math._saved = torch.get_num_threads()
                if host_math._saved != 1:
                    torch.set_num_threads(1)
        return self

    def __exit__(self, *exc):
        with host_math._lock:
            host_math._depth -= 1
            if host_math._depth == 0 and host_math._saved != 1:
                torch.set_num_threads(host_math._saved)
        return False


def _unit(v: torch.Tensor) -> torch.Tensor:
    return v / torch.norm(v, dim=-1, keepdim=True)


def truncated_normal(n: int, mean: float, std: float, n_stds: float, device=_CPU, cand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[n,1] draws from N(mean, std) restricted to mean +- n_stds*std.

    Four candidates per sample (one `normal_()` call on [n,1,4], which is what fixes the RNG
    consumption), the first candidate strictly inside the bounds wins, and whatever is chosen is
    clipped to the closed interval.  With std == 0 every candidate equals `mean`.
    `cand`: standard-normal candidates drawn by the caller ([m,1,4], consumed in place; `draw_angle_noise`).
    """
    assert std >= 0, f"{std}"
    if cand is None:
        cand = torch.empty((n, 1, 4), dtype=torch.float32, device=device).normal_()
    cand.mul_(std).add_(mean)
    lo = mean - 1 * n_stds * std
    hi = mean + n_stds * std
    inside = (cand < hi) & (cand > lo)
    pick = inside.max(-1, keepdim=True)[1]
    out = cand.gather(-1, pick).squeeze(-1)
    out[out <= lo] = lo
    out[out >= hi] = hi
    return out


def draw_angle_noise(n: int, method: str, generator: Optional[torch.Generator] = None, device=_CPU) -> Tuple[torch.Tensor, torch.Tensor]:
    """The raw random numbers of ONE `sample_sphere_angles(random=True)` call, in its RNG order (yaw noise, then pitch noise): the
    only part of pose sampling that touches the generator.  `generator=None` = torch's default CPU generator, like the reference."""
    if method == "uniform":
        return torch.rand((n, 1), device=device, generator=generator), torch.rand((n, 1), device=device, generator=generator)
    if method in ("normal", "gaussian"):
        return torch.randn((n, 1), device=device, generator=generator), torch.randn((n, 1), device=device, generator=generator)
    if method == "truncated_gaussian":
        return (torch.empty((n, 1, 4), dtype=torch.float32, device=device).normal_(generator=generator),
                torch.empty((n, 1, 4), dtype=torch.float32, device=device).normal_(generator=generator))
    raise ValueError(method)


def angles_from_noise(noise_yaw: torch.Tensor, noise_pitch: torch.Tensor, yaw_mean, yaw_std, pitch_mean, pitch_std, *, method: str,
                      n_stds: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(yaws, pitches) [m,1] from the raw draws of `draw_angle_noise` -- of one call, or of several calls concatenated along dim 0
    (every operation is per element / per row, so the rows of a batch equal the rows of the separate calls bit for bit)."""
    m = noise_yaw.shape[0]
    if method == "uniform":
        return (noise_yaw - 0.5) * 2 * n_stds * yaw_std + yaw_mean, (noise_pitch - 0.5) * 2 * n_stds * pitch_std + pitch_mean
    if method in ("normal", "gaussian"):
        return noise_yaw * yaw_std + yaw_mean, noise_pitch * pitch_std + pitch_mean
    if method == "truncated_gaussian":
        return (truncated_normal(m, yaw_mean, yaw_std, n_stds, noise_yaw.device, cand=noise_yaw),
                truncated_normal(m, pitch_mean, pitch_std, n_stds, noise_pitch.device, cand=noise_pitch))
    raise ValueError(method)


def sample_sphere_angles(n: int, yaw_mean, yaw_std, pitch_mean, pitch_std, *, random: bool, method: str,
                         n_stds: float, horizontal_sweep: bool = True, device=_CPU) -> Tuple[torch.Tensor, torch.Tensor]:
    """(yaws, pitches), each [n,1] float32.  RNG order: yaws first, then pitches."""
    if random:
        noise_yaw, noise_pitch = draw_angle_noise(n, method, None, device)
        yaws, pitches = angles_from_noise(noise_yaw, noise_pitch, yaw_mean, yaw_std, pitch_mean, pitch_std, method=method, n_stds=n_stds)
    else:
        sweep = torch.linspace(-n_stds, n_stds, steps=n, device=device).reshape((n, 1))
        if horizontal_sweep:
            yaws = sweep * yaw_std + yaw_mean
            pitches = torch.ones((n, 1), device=device) * pitch_mean
        else:
            yaws = torch.ones((n, 1), device=device) * yaw_mean
            pitches = sweep * pitch_std + pitch_mean
    return yaws, pitches


def sphere_positions(yaws: torch.Tensor, pitches: torch.Tensor, r: float) -> torch.Tensor:
    """[n,3] camera positions in the sphere frame (float32)."""
    n = yaws.shape[0]
    pos = torch.zeros((n, 3), device=yaws.device)
    ring = r * torch.abs(torch.cos(pitches))
    pos[:, 0:1] = ring * torch.cos(yaws)
    pos[:, 1:2] = ring * torch.sin(yaws)
    pos[:, 2:3] = r * torch.sin(pitches)
    return pos


def look_at_centre(pos: torch.Tensor) -> torch.Tensor:
    """[n,4,4] float32 camera-to-sphere matrices of cameras at `pos` looking at the origin.

    Camera axes: +X right, +Y down, +Z forward; columns of the rotation are (right, down, forward).
    """
    n = pos.shape[0]
    dev = pos.device
    fwd = _unit(_unit(-pos))  # the reference normalises twice (gen_sphere_path, then create_cam2sphere_sys_matrix)
    down0 = torch.tensor([0, 0, -1], dtype=torch.float, device=dev).expand_as(fwd)
    right = _unit(torch.cross(down0, fwd, dim=-1))
    down = _unit(torch.cross(fwd, right, dim=-1))
    rot = torch.eye(4, device=dev).unsqueeze(0).repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((right, down, fwd), axis=-1)
    trans = torch.eye(4, device=dev).unsqueeze(0).repeat(n, 1, 1)
    trans[:, :3, 3] = pos
    return trans @ rot


_FRAME_CACHE = {}


def sphere_to_mpi_frame(sphere_center: np.ndarray) -> np.ndarray:
    """4x4 float64: sphere-frame coordinates -> MPI-frame coordinates (rotate axes, then move to the centre).
    Constant per renderer, so the matrix is cached per centre (two scipy Rotation objects per call otherwise)."""
    key = tuple(float(v) for v in np.asarray(sphere_center, dtype=np.float64).reshape(-1))
    hit = _FRAME_CACHE.get(key)
    if hit is not None:
        return hit.copy()
    turn_z = np.eye(4)
    turn_z[:3, :3] = Rotation.from_euler("Z", -90, degrees=True).as_matrix()  # -> +X right, +Y forward, +Z up
    turn_x = np.eye(4)
    turn_x[:3, :3] = Rotation.from_euler("X", 90, degrees=True).as_matrix()   # -> +X right, +Y down, +Z forward
    rot = np.matmul(turn_x, turn_z)
    shift = np.eye(4)
    shift[:3, 3] = np.asarray(sphere_center).reshape(-1)
    out = np.matmul(shift, rot)
    if len(_FRAME_CACHE) < 64:
        _FRAME_CACHE[key] = out.copy()
    return out


def gen_sphere_path(n_cams: int, sphere_center: np.ndarray, sphere_r: Optional[float], yaw_mean=0.0,
                    yaw_std=np.sqrt(np.pi), pitch_mean=0.0, pitch_std=np.sqrt(np.pi), given_yaws=None,
                    given_pitches=None, flag_rnd=True, flag_det_horizontal=True, sample_method="uniform",
                    n_truncated_stds=2, device=_CPU):
    """Same call signature and return value as the reference's `gen_sphere_path` (cam_utils.py:734):
    (c2w [n,4,4] float64 numpy in the MPI frame, yaws [n,1], pitches [n,1])."""
    if sphere_r is None:
        sphere_r = np.linalg.norm(sphere_center, ord=2)
    with host_math():
        if given_yaws is None:
            assert given_pitches is None
            yaws, pitches = sample_sphere_angles(n_cams, yaw_mean, yaw_std, pitch_mean, pitch_std, random=flag_rnd,
                                                 method=sample_method, n_stds=n_truncated_stds,
                                                 horizontal_sweep=flag_det_horizontal, device=device)
        else:
            yaws, pitches = given_yaws, given_pitches
        pos = sphere_positions(yaws, pitches, sphere_r)
        cam2sphere = look_at_centre(pos).cpu().numpy()
    c2w = np.matmul(sphere_to_mpi_frame(sphere_center), cam2sphere)
    return c2w, yaws, pitches


def gen_sphere_paths_ahead(n_calls: int, n_cams: int, sphere_center: np.ndarray, sphere_r: Optional[float], yaw_mean, yaw_std, pitch_mean,
                           pitch_std, sample_method: str, n_truncated_stds, generator: torch.Generator):
    """The poses of the next `n_calls` calls of `gen_sphere_path(n_cams, ..., flag_rnd=True)` as they would come out if the default
    generator were in `generator`'s state: the random draws are taken from `generator` call by call (its state after every call is
    recorded), the arithmetic runs once over all calls.  Returns (c2w [n_calls, n_cams, 4, 4] float64 numpy, yaws, pitches as
    [n_calls, n_cams, 1] float32, states: n_calls + 1 generator states -- states[j] before call j, states[j + 1] after it).
    Bit-identical to n_calls separate calls (tests/test_host_geometry.py)."""
    if sphere_r is None:
        sphere_r = np.linalg.norm(sphere_center, ord=2)
    with host_math():
        states = [generator.get_state()]
        ny, npi = [], []
        for _ in range(n_calls):
            a, b = draw_angle_noise(n_cams, sample_method, generator)
            ny.append(a), npi.append(b)
            states.append(generator.get_state())
        yaws, pitches = angles_from_noise(torch.cat(ny, 0), torch.cat(npi, 0), yaw_mean, yaw_std, pitch_mean, pitch_std,
                                          method=sample_method, n_stds=n_truncated_stds)
        pos = sphere_positions(yaws, pitches, sphere_r)
        cam2sphere = look_at_centre(pos).cpu().numpy()
    c2w = np.matmul(sphere_to_mpi_frame(sphere_center), cam2sphere)
    return (c2w.reshape(n_calls, n_cams, 4, 4), yaws.reshape(n_calls, n_cams, 1), pitches.reshape(n_calls, n_cams, 1), states)


def yaw_pitch_from_w2c(w2c_mat: torch.Tensor, sphere_c: torch.Tensor):
    """(yaws, pitches) [B,1] of cameras given world-to-camera matrices [B,4,4] (diagnostics only;
    mirrors gmpi/utils/cam_utils.py:1005-1050 `compute_pitch_yaw_from_w2c_mat`)."""
    assert sphere_c.ndim <= 2, f"{sphere_c.shape}"
    assert w2c_mat.ndim == 3, f"{w2c_mat.shape}"
    bs = w2c_mat.shape[0]
    world2sphere = torch.inverse(torch.FloatTensor(sphere_to_mpi_frame(sphere_c.numpy()))).unsqueeze(0).expand(bs, -1, -1)
    origin = torch.FloatTensor([0, 0, 0, 1]).reshape((1, 4, 1)).expand(bs, -1, -1)
    cam_in_world = torch.matmul(torch.inverse(w2c_mat), origin)
    cam_in_sphere = torch.matmul(world2sphere, cam_in_world)[:, :3]
    cam_in_sphere = cam_in_sphere / torch.norm(cam_in_sphere, p=2, dim=1, keepdim=True)
    yaws = torch.atan2(cam_in_sphere[:, 1], cam_in_sphere[:, 0])
    pitches = np.pi / 2 - torch.acos(cam_in_sphere[:, 2])
    return yaws, pitches
