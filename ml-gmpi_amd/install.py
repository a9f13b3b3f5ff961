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


# The classes install() hands to the reference's modules: subclasses that carry the default range check, defined ONCE per value at module
# level (importable as ml_gmpi_amd.install.MPI_full / MPIRenderer_full / ...), so that `torch.save` of a module holding one, a spawn-based
# DataLoader or a multiprocessing copy can pickle them, and `isinstance` holds across repeated install() calls.
_INSTALLED = {}


def _installed_classes(range_check: str):
    hit = _INSTALLED.get(range_check)
    if hit is None:
        from .hip_mpi import MPI as base_mpi
        from .renderer import MPIRenderer as base_renderer

        def _init(self, **kw):
            kw.setdefault("range_check", range_check)
            base_renderer.__init__(self, **kw)

        mpi_cls = type(f"MPI_{range_check}", (base_mpi,), {"DEFAULT_RANGE_CHECK": range_check, "__module__": __name__})
        ren_cls = type(f"MPIRenderer_{range_check}", (base_renderer,), {"__init__": _init, "__module__": __name__})
        globals()[mpi_cls.__name__], globals()[ren_cls.__name__] = mpi_cls, ren_cls
        hit = _INSTALLED[range_check] = (mpi_cls, ren_cls)
    return hit


for _rc in ("full", "touched", "off"):  # (defined at import: a pickle is loadable in a process that never called install())
    _installed_classes(_rc)


def install(patch_mpi: bool = True, patch_renderer: bool = True, patch_light: bool = True, range_check: str = "full") -> None:
    """`range_check`: what a swapped-in `MPI` asserts by default.  "full" = the reference's behaviour (min/max over the WHOLE
    volume, mpi.py:185-187 / mpi_renderer.py:447-449: one extra streaming pass, ~0.6 ms per 3.2 GB); "touched" = only
    the texels a render samples (free, but a NaN in a texel no view touches goes unnoticed)."""
    from .light import LightRenderer

    assert range_check in ("full", "touched", "off"), range_check
    # What the reference's modules get are SUBCLASSES that carry the default: `ml_gmpi_amd.MPI` / `MPIRenderer` themselves -- and every
    # instance a direct user of this package builds, before or after install() -- keep "touched" (no process-global state).
    MPI, MPIRenderer = _installed_classes(range_check)

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
