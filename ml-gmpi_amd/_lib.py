"""ctypes binding of the C-ABI library `libgmpi_render.so` (include/gmpi_render.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  If the shared object is
missing or does not export the ABI this header declares, loading fails loudly with instructions to
build it (`python -c "import __graft_entry__ as g; g.build()"` or `make -C ml-gmpi_amd/csrc`).
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_PKG, "libgmpi_render.so")
_LIB = None

ABI_VERSION = 2
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
FLAG_ALIGN_CORNERS, FLAG_OUT_PM1, FLAG_CHECK_LAST_PLANE, FLAG_CHECK_RANGE, FLAG_STRICT_ORDER, FLAG_HINT_FRONTAL, FLAG_HINT_TILTED = 1, 2, 4, 8, 16, 32, 64
FLAG_HINT_OBLIQUE = 256     # advisory: some camera axis more than 0.35 rad off the MPI normal (views that share an MPI then stay on the tile kernel)
FLAG_GRAD_OVERWRITE = 128   # backward only: grad_rgba's content is not needed (with the backward's workspace every element is written: no zero-fill)
STATUS_OUT_OF_LAST_PLANE, STATUS_RGBA_RANGE, STATUS_CAMERA_BEHIND_PLANE, STATUS_BAD_VIEW_INDEX = 1, 2, 4, 8
STATUS_WORDS = 4
VARIANT_AUTO, VARIANT_GATHER, VARIANT_LDS, VARIANT_WAVE, VARIANT_DMA, VARIANT_BAND = 0, 1, 2, 3, 4, 5
VARIANTS = {"auto": VARIANT_AUTO, "gather": VARIANT_GATHER, "lds": VARIANT_LDS, "wave": VARIANT_WAVE, "band": VARIANT_BAND}  # (4 = the retired GMPI_VARIANT_DMA)

EXPORTS = (
    "gmpi_mpi_render_launch",
    "gmpi_render_workspace_bytes",
    "gmpi_render_backward_workspace_bytes",
    "gmpi_mpi_render_backward_launch",
    "gmpi_last_plane_uv_minmax_launch",
    "gmpi_rgba_range_check_launch",
    "gmpi_frames_to_uint8_launch",
    "gmpi_generate_rays_launch",
    "gmpi_alpha_depth_launch",
    "gmpi_light_blur_launch",
    "gmpi_light_shading_launch",
    "gmpi_light_apply_launch",
    "gmpi_light_apply_backward_launch",
    "gmpi_alpha_depth_backward_launch",
    "gmpi_selftest_division_launch",
    "gmpi_stream_probe_launch",
    "gmpi_query",
    "gmpi_version_string",
)

_ERRORS = {
    -1: "GMPI_E_NULL (required pointer is NULL)",
    -2: "GMPI_E_SHAPE (non-positive or inconsistent extent, or more than 65535 views for the gather kernel / the backward)",
    -3: "GMPI_E_DTYPE (unknown rgba dtype)",
    -4: "GMPI_E_STRIDE (innermost rgba stride must be 1, strides non-negative)",
    -5: "GMPI_E_ABI (GmpiRenderParams size mismatch between binding and library)",
    -6: "GMPI_E_VARIANT (requested kernel variant cannot run this shape)",
    -7: "GMPI_E_FLAGS (undefined bit in GmpiRenderParams.flags)",
}


class GmpiError(RuntimeError):
    pass


class GmpiRenderParams(ctypes.Structure):
    """Field-for-field mirror of `struct GmpiRenderParams` in include/gmpi_render.h."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("flags", ctypes.c_uint32),
        ("variant", ctypes.c_int32), ("rgba_dtype", ctypes.c_int32),
        ("N", ctypes.c_int32), ("M", ctypes.c_int32), ("D", ctypes.c_int32),
        ("Ht", ctypes.c_int32), ("Wt", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("views_per_mpi", ctypes.c_int32),
        ("rgba", ctypes.c_void_p), ("rgba_stride", ctypes.c_int64 * 5),
        ("view_to_mpi", ctypes.c_void_p), ("dhw", ctypes.c_void_p), ("ray_dir", ctypes.c_void_p),
        ("eye_pos", ctypes.c_void_p), ("z_dir", ctypes.c_void_p),
        ("rgb_out", ctypes.c_void_p), ("depth_out", ctypes.c_void_p), ("transmittance_out", ctypes.c_void_p),
        ("status", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_uint64),
    ]


def source_hash() -> str:
    """Short hash of the kernel sources (csrc/*.hip, *.hpp, include/gmpi_render.h): profiles/hbm_traffic.json is keyed by it,
    so that bench.py only quotes PMC-measured traffic that belongs to the kernels it is timing."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_PKG, "csrc", "*.hip")) + glob.glob(os.path.join(_PKG, "csrc", "*.hpp")))
    files.append(os.path.join(os.path.dirname(_PKG), "include", "gmpi_render.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def library_path() -> str:
    return _SO


_STAMP = os.path.join(_PKG, "csrc", ".built_from")


def build_extension(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (ml-gmpi_amd/csrc/Makefile). Returns the .so path.

    `make` trusts time stamps; a snapshot can carry objects whose time stamps are newer than sources they were not built from.  So a
    successful build leaves the hash of the sources it compiled in csrc/.built_from, and a build that finds another hash there (or none)
    recompiles everything (`make -B`), as does force=True or GMPI_BUILD_FORCE=1 in the environment."""
    want = source_hash() + " " + _makefile_hash()
    have = open(_STAMP).read().strip() if os.path.isfile(_STAMP) else None
    force = force or os.environ.get("GMPI_BUILD_FORCE", "") not in ("", "0") or have != want or not os.path.isfile(_SO)
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0 or not os.path.isfile(_SO):
        raise GmpiError("building libgmpi_render.so failed:\n" + res.stderr[-4000:])
    with open(_STAMP, "w") as f:
        f.write(want + "\n")
    return _SO


def _makefile_hash() -> str:
    import hashlib
    with open(os.path.join(_PKG, "csrc", "Makefile"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:8]


def build_report() -> str:
    """One line per object: what build() prints so that a log shows which sources the library in use was built from."""
    import glob
    import time
    rows = [f"sources {source_hash()} (csrc/*.hip, *.hpp, include/gmpi_render.h); stamp {open(_STAMP).read().strip() if os.path.isfile(_STAMP) else None}"]
    for f in sorted(glob.glob(os.path.join(_PKG, "csrc", "*.o"))) + [_SO]:
        if os.path.isfile(f):
            rows.append(f"  {os.path.relpath(f, os.path.dirname(_PKG)):44s} {os.path.getsize(f):9d} B  {time.strftime('%Y-%m-%d %H:%M:%S', time.localtime(os.path.getmtime(f)))}")
    return "\n".join(rows)


def load_library():
    """Load (once) and type the C ABI.  Raises GmpiError when the HIP library is not there."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.isfile(_SO):
        raise GmpiError(
            f"HIP extension not built: {_SO} is missing. This package has no CPU/PyTorch fallback. "
            "Build it with `python -c \"import __graft_entry__ as g; g.build()\"` or `make -C ml-gmpi_amd/csrc`.")
    import torch  # noqa: F401  -- load torch's ROCm runtime first so both sides share one libamdhip64
    try:
        lib = ctypes.CDLL(_SO)
    except OSError as e:
        raise GmpiError(f"could not load {_SO}: {e}") from e
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise GmpiError(f"{_SO} does not export {missing}; rebuild it")
    vp = ctypes.c_void_p
    lib.gmpi_mpi_render_launch.restype = ctypes.c_int
    lib.gmpi_mpi_render_launch.argtypes = [ctypes.POINTER(GmpiRenderParams), vp]
    lib.gmpi_render_workspace_bytes.restype = ctypes.c_uint64
    lib.gmpi_render_workspace_bytes.argtypes = [ctypes.POINTER(GmpiRenderParams)]
    lib.gmpi_render_backward_workspace_bytes.restype = ctypes.c_uint64
    lib.gmpi_render_backward_workspace_bytes.argtypes = [ctypes.POINTER(GmpiRenderParams)]
    lib.gmpi_mpi_render_backward_launch.restype = ctypes.c_int
    lib.gmpi_mpi_render_backward_launch.argtypes = [ctypes.POINTER(GmpiRenderParams), vp, vp, vp,
                                                    ctypes.POINTER(ctypes.c_int64), vp]
    lib.gmpi_last_plane_uv_minmax_launch.restype = ctypes.c_int
    lib.gmpi_last_plane_uv_minmax_launch.argtypes = [ctypes.POINTER(GmpiRenderParams), vp, vp]
    lib.gmpi_rgba_range_check_launch.restype = ctypes.c_int
    lib.gmpi_rgba_range_check_launch.argtypes = [vp, ctypes.c_int32, ctypes.c_int64, vp, vp]
    lib.gmpi_frames_to_uint8_launch.restype = ctypes.c_int
    lib.gmpi_frames_to_uint8_launch.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_double, ctypes.c_double, vp, vp, vp]
    lib.gmpi_generate_rays_launch.restype = ctypes.c_int
    lib.gmpi_generate_rays_launch.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp]
    lib.gmpi_alpha_depth_launch.restype = ctypes.c_int
    lib.gmpi_alpha_depth_launch.argtypes = [vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, vp]
    lib.gmpi_light_blur_launch.restype = ctypes.c_int
    lib.gmpi_light_blur_launch.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int32, vp]
    lib.gmpi_light_shading_launch.restype = ctypes.c_int
    lib.gmpi_light_shading_launch.argtypes = [vp, vp, vp, ctypes.c_float, ctypes.c_float, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32, vp, vp]
    lib.gmpi_light_apply_launch.restype = ctypes.c_int
    lib.gmpi_light_apply_launch.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), vp, vp, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    lib.gmpi_light_apply_backward_launch.restype = ctypes.c_int
    lib.gmpi_light_apply_backward_launch.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), vp, vp, vp, vp,
                                                     ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    lib.gmpi_alpha_depth_backward_launch.restype = ctypes.c_int
    lib.gmpi_alpha_depth_backward_launch.argtypes = [vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, vp, vp, vp,
                                                     vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                     ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp]
    lib.gmpi_stream_probe_launch.restype = ctypes.c_int
    lib.gmpi_stream_probe_launch.argtypes = [vp, ctypes.c_uint64, vp, vp]
    lib.gmpi_selftest_division_launch.restype = ctypes.c_int
    lib.gmpi_selftest_division_launch.argtypes = [ctypes.c_uint64, ctypes.c_uint32, vp, vp]
    lib.gmpi_query.restype = ctypes.c_int
    lib.gmpi_query.argtypes = [ctypes.c_int32]
    lib.gmpi_version_string.restype = ctypes.c_char_p
    lib.gmpi_version_string.argtypes = []
    if lib.gmpi_query(0) != ABI_VERSION or lib.gmpi_query(1) != ctypes.sizeof(GmpiRenderParams):
        raise GmpiError(f"ABI mismatch: library ABI {lib.gmpi_query(0)} / struct {lib.gmpi_query(1)} B, "
                        f"binding ABI {ABI_VERSION} / struct {ctypes.sizeof(GmpiRenderParams)} B")
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc <= -100:
        raise GmpiError(f"{what}: HIP launch failed (hipError_t {-100 - rc})")
    raise GmpiError(f"{what}: {_ERRORS.get(rc, rc)}")
