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
