"""ctypes front-end of the CPU oracle (oracle/mpi_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product package never does (tests/test_boundary.py greps for it).

Also holds the version-independent synthetic RGBA generator used by the golden fixtures
(`synth_rgba`): pure 64-bit integer hashing in numpy, so the same (seed, shape) gives the same
bytes on any numpy/torch version.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

STATUS_OUT_OF_LAST_PLANE = 1
STATUS_RGBA_RANGE = 2
STATUS_CAMERA_BEHIND_PLANE = 4


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (see oracle/Makefile)."""
    need = force or not all(os.path.isfile(os.path.join(_HERE, n)) for n in ("libgmpi_oracle.so", "libgmpi_oracle_omp.so"))
    src_m = os.path.getmtime(os.path.join(_HERE, "mpi_oracle.c"))
    for n in ("libgmpi_oracle.so", "libgmpi_oracle_omp.so"):
        p = os.path.join(_HERE, n)
        if os.path.isfile(p) and os.path.getmtime(p) < src_m:
            need = True
    if need:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "all"], check=True, capture_output=True)


def _lib(threads: bool):
    name = "libgmpi_oracle_omp.so" if threads else "libgmpi_oracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.isfile(path):
            build()
        lib = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        up = ctypes.POINTER(ctypes.c_uint32)
        lib.gmpi_oracle_render.restype = ctypes.c_int
        lib.gmpi_oracle_render.argtypes = [fp, ip, fp, fp, fp, fp] + [ctypes.c_int] * 8 + [fp, fp, fp, fp, up]
        lib.gmpi_oracle_coords.restype = ctypes.c_int
        lib.gmpi_oracle_coords.argtypes = [ip, fp, fp, fp] + [ctypes.c_int] * 8 + [fp, fp]
        lib.gmpi_oracle_range_check.restype = ctypes.c_uint32
        lib.gmpi_oracle_range_check.argtypes = [fp, ctypes.c_size_t]
        lib.gmpi_oracle_alpha_depth.restype = ctypes.c_int
        lib.gmpi_oracle_alpha_depth.argtypes = [fp, fp] + [ctypes.c_int] * 4 + [fp, fp]
        lib.gmpi_oracle_num_threads.restype = ctypes.c_int
        _LIBS[name] = lib
    return _LIBS[name]


def _f32(a):
    if hasattr(a, "detach"):  # torch tensor (any float dtype; bf16 is upcast like mpi_renderer.py:446)
        a = a.detach().float().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ct=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def num_threads(threads: bool = True) -> int:
    return int(_lib(threads).gmpi_oracle_num_threads())


def render(rgba, dhw, ray_dir, eye, zdir, view_to_mpi=None, align_corners=True, want_T=True, threads=False):
    """Oracle render.

    rgba [M,D,4,Ht,Wt], dhw [M,D,3], ray_dir [N,3,H,W], eye [N,3], zdir [N,3], view_to_mpi [N] or None (identity).
    Returns dict(color [N,3,H,W] in [0,1], depth [N,1,H,W], T [N,1,H,W], uv_minmax [N,4], status int).
    """
    rgba, dhw, ray_dir, eye, zdir = map(_f32, (rgba, dhw, ray_dir, eye, zdir))
    M, D, C, Ht, Wt = rgba.shape
    assert C == 4, rgba.shape
    N, three, H, W = ray_dir.shape
    assert three == 3 and dhw.shape == (M, D, 3) and eye.shape == (N, 3) and zdir.shape == (N, 3)
    if view_to_mpi is None:
        assert N == M
        v2m = np.arange(N, dtype=np.int32)
    else:
        v2m = np.ascontiguousarray(np.asarray(view_to_mpi), dtype=np.int32)
        assert v2m.shape == (N,)
    color = np.empty((N, 3, H, W), np.float32)
    depth = np.empty((N, 1, H, W), np.float32)
    T = np.empty((N, 1, H, W), np.float32) if want_T else None
    uv = np.empty((N, 4), np.float32)
    status = np.zeros(4, np.uint32)
    rc = _lib(threads).gmpi_oracle_render(
        _ptr(rgba), _ptr(v2m, ctypes.c_int32), _ptr(dhw), _ptr(ray_dir), _ptr(eye), _ptr(zdir),
        N, M, D, Ht, Wt, H, W, int(bool(align_corners)),
        _ptr(color), _ptr(depth), _ptr(T) if want_T else None, _ptr(uv), _ptr(status, ctypes.c_uint32))
    if rc != 0:
        raise RuntimeError(f"gmpi_oracle_render failed: {rc}")
    return dict(color=color, depth=depth, T=T, uv_minmax=uv, status=int(status[0]))


def coords(dhw, ray_dir, eye, Ht, Wt, view_to_mpi=None, align_corners=True):
    """ix, iy [N,D,H,W] sampling coordinates (for localising mismatches)."""
    dhw, ray_dir, eye = map(_f32, (dhw, ray_dir, eye))
    M, D, _ = dhw.shape
    N, _, H, W = ray_dir.shape
    v2m = np.arange(N, dtype=np.int32) if view_to_mpi is None else np.ascontiguousarray(view_to_mpi, dtype=np.int32)
    ix = np.empty((N, D, H, W), np.float32)
    iy = np.empty((N, D, H, W), np.float32)
    rc = _lib(False).gmpi_oracle_coords(_ptr(v2m, ctypes.c_int32), _ptr(dhw), _ptr(ray_dir), _ptr(eye),
                                        N, M, D, Ht, Wt, H, W, int(bool(align_corners)), _ptr(ix), _ptr(iy))
    if rc != 0:
        raise RuntimeError(f"gmpi_oracle_coords failed: {rc}")
    return ix, iy


def alpha_depth(mpi_alpha, plane_ds):
    """LightRenderer.compute_depth restated: mpi_alpha [B,D,1,H,W] -> (depth [B,1,H,W], T [B,1,H,W])."""
    a = _f32(mpi_alpha)
    B, D, one, H, W = a.shape
    assert one == 1
    ds = _f32(plane_ds).reshape(-1)
    assert ds.size == D
    depth = np.empty((B, 1, H, W), np.float32)
    T = np.empty((B, 1, H, W), np.float32)
    rc = _lib(False).gmpi_oracle_alpha_depth(_ptr(a), _ptr(ds), B, D, H, W, _ptr(depth), _ptr(T))
    if rc != 0:
        raise RuntimeError(rc)
    return depth, T


def gaussian_kernel1d(ksize: int, sigma: float) -> np.ndarray:
    """torchvision.transforms.functional `_get_gaussian_kernel1d`, float32 step by step."""
    lim = np.float32((ksize - 1) * 0.5)
    x = np.linspace(-lim, lim, ksize, dtype=np.float32)
    pdf = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    return (pdf / pdf.sum(dtype=np.float32)).astype(np.float32)


def light_shade(rgba, plane_ds, xyz_last, light_dir, ka, kd, ksize=9, sigma=None):
    """LightRenderer.render (light_renderer.py:122-199) restated in numpy float32 for a GIVEN light direction [B,3]:
    compute_depth (:82-100) -> GaussianBlur (reflect padding, :51-55) -> compute_pcl (:102-120) -> get_normal (:57-80)
    -> diffuse = clamp(-n.l, 0), shading = ka + kd*diffuse (:163-190) -> clip(rgb*shading, 0, 1) | alpha (:193-198)."""
    f = np.float32
    rgba = _f32(rgba)
    B, D, _, H, W = rgba.shape
    depth, _ = alpha_depth(rgba[:, :, 3:], plane_ds)
    if sigma is None:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    k1 = gaussian_kernel1d(ksize, sigma)
    k2 = (k1[:, None] * k1[None, :]).astype(f)
    r = ksize // 2
    pad = np.pad(depth[:, 0], ((0, 0), (r, r), (r, r)), mode="reflect")
    blur = np.zeros((B, H, W), f)
    for dy in range(ksize):
        for dx in range(ksize):
            blur += pad[:, dy:dy + H, dx:dx + W] * k2[dy, dx]
    xyz = _f32(xyz_last).reshape(1, H, W, 3)
    scale = blur[..., None] / (xyz[..., 2:] + f(1e-8))
    g = (xyz * scale).astype(f)                                     # [B,H,W,3]
    c = g[:, 1:-1, 1:-1]
    up, down, left, right = g[:, :-2, 1:-1], g[:, 2:, 1:-1], g[:, 1:-1, :-2], g[:, 1:-1, 2:]
    n = np.cross(up - c, left - c) + np.cross(left - c, down - c) + np.cross(down - c, right - c) + np.cross(right - c, up - c)
    n = np.pad(n.astype(f), ((0, 0), (1, 1), (1, 1), (0, 0)), mode="edge")
    n = n / (np.sqrt((n ** 2).sum(3, keepdims=True, dtype=f)) + f(1e-8))
    ld = _f32(light_dir).reshape(B, 1, 1, 3)
    diffuse = np.maximum(f(-1.0) * (n * ld).sum(3, dtype=f), f(0.0))
    shading = (f(ka) + diffuse * f(kd)).astype(f)                  # [B,H,W]
    out = rgba.copy()
    out[:, :, :3] = np.clip(rgba[:, :, :3] * shading[:, None, None], f(0.0), f(1.0))
    return out, shading


def range_check(rgba) -> int:
    rgba = _f32(rgba)
    return int(_lib(False).gmpi_oracle_range_check(_ptr(rgba), rgba.size))


# ---------------------------------------------------------------------------------------------
# Version-independent synthetic inputs for the golden fixtures
# ---------------------------------------------------------------------------------------------
def synth_rgba(seed: int, shape, last_alpha_one: bool = False, bf16_round: bool = False) -> np.ndarray:
    """U[0,1) white-noise RGBA stack [M,D,4,Ht,Wt] float32 from a splitmix64-style integer hash.

    Values are k / 2^24 (exactly representable); `bf16_round` additionally truncates the mantissa to
    8 bits (value exactly representable in bfloat16), `last_alpha_one` sets alpha of the last plane
    to 1 (the generator's `background_alpha_full`, networks_cond_on_pos_enc.py:1307-1310).
    """
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    v = ((x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(shape)
    if bf16_round:
        v = (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    if last_alpha_one:
        v[:, -1, 3] = 1.0
    return np.ascontiguousarray(v)
