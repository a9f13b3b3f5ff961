"""Public names of the package (see ml_gmpi_amd/__init__.py for why this is not an __init__)."""
from ._lib import GmpiError, build_extension, library_path, load_library
from .hip_mpi import MPI, HipMPI, flush_status
from .renderer import MPIRenderer, PRESETS, make_renderer
from .driver import ViewBatchDriver, shard_views, render_views_sharded, frames_to_uint8, dump_frames
from .install import install, uninstall
from .light import LightRenderer, compute_depth

__all__ = [
    "GmpiError", "build_extension", "library_path", "load_library",
    "MPI", "HipMPI", "flush_status", "MPIRenderer", "PRESETS", "make_renderer",
    "ViewBatchDriver", "shard_views", "render_views_sharded", "frames_to_uint8", "dump_frames",
    "install", "uninstall", "compute_depth", "LightRenderer",
]
