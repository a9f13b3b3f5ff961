"""`install()` -- swap the reference's renderer for the HIP one, so that an unmodified
`gmpi/eval/vis/render_video.py` (which does `from gmpi.core.mpi_renderer import MPIRenderer`,
render_video.py:11; likewise prepare_fake_data.py:11, train.py:24, eval/common.py:5) runs on it.

Call it after the reference is importable (on PYTHONPATH) and BEFORE the script imports the names:

    import ml_gmpi_amd; ml_gmpi_amd.install()
    runpy.run_path("gmpi/eval/vis/render_video.py", run_name="__main__")
"""
import importlib
import sys

_SAVED = {}


def install(patch_mpi: bool = True, patch_renderer: bool = True, patch_light: bool = True, range_check: str = "full") -> None:
    """`range_check`: what a swapped-in `MPI` asserts by default.  "full" = the reference's behaviour (min/max over the WHOLE
    volume, mpi.py:185-187 / mpi_renderer.py:447-449: one extra streaming pass, ~0.6 ms per 3.2 GB); "touched" = only
    the texels a render samples (free, but a NaN in a texel no view touches goes unnoticed)."""
    from .hip_mpi import MPI
    from .light import LightRenderer
    from .renderer import MPIRenderer

    assert range_check in ("full", "touched", "off"), range_check
    # What the reference's modules get are SUBCLASSES that carry the default: `ml_gmpi_amd.MPI` / `MPIRenderer` themselves -- and every
    # instance a direct user of this package builds, before or after install() -- keep "touched" (no process-global state).
    base_mpi, base_renderer = MPI, MPIRenderer

    class MPI(base_mpi):  # noqa: F811  (same name: reprs and pickles of the swapped-in class read like the reference's)
        DEFAULT_RANGE_CHECK = range_check

    class MPIRenderer(base_renderer):  # noqa: F811
        def __init__(self, **kw):
            kw.setdefault("range_check", range_check)
            super().__init__(**kw)

    core_mpi = importlib.import_module("gmpi.core.mpi")
    core_renderer = importlib.import_module("gmpi.core.mpi_renderer")
    if patch_mpi:
        _SAVED.setdefault(("gmpi.core.mpi", "MPI"), core_mpi.MPI)
        _SAVED.setdefault(("gmpi.core.mpi_renderer", "MPI"), core_renderer.MPI)
        core_mpi.MPI = MPI
        core_renderer.MPI = MPI  # `self.mpi = MPI(...)` in the reference's own MPIRenderer (mpi_renderer.py:47)
    if patch_renderer:
        _SAVED.setdefault(("gmpi.core.mpi_renderer", "MPIRenderer"), core_renderer.MPIRenderer)
        core_renderer.MPIRenderer = MPIRenderer
    if patch_light:
        # train.py:23 `from gmpi.core.light_renderer import LightRenderer`.  The reference module imports torchvision at
        # import time; where that is unavailable the module cannot be imported and there is nothing to patch.
        try:
            core_light = importlib.import_module("gmpi.core.light_renderer")
        except ImportError:
            core_light = None
        if core_light is not None:
            _SAVED.setdefault(("gmpi.core.light_renderer", "LightRenderer"), core_light.LightRenderer)
            core_light.LightRenderer = LightRenderer


def uninstall() -> None:
    for (mod, name), obj in list(_SAVED.items()):
        if mod in sys.modules:
            setattr(sys.modules[mod], name, obj)
    _SAVED.clear()
