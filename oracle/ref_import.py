"""TEST INFRASTRUCTURE ONLY -- imports the real reference (`/root/reference/gmpi/core`) on CPU.

Used by `oracle/make_golden.py` (golden-fixture generation) and by the CPU tests that are
allowed to run only in the build container (they skip when `/root/reference` is absent, which
is the case on the GPU box).  Nothing in the product package (`ml-gmpi_amd/`) may import this.

The reference package imports three third-party modules at import time that are not installed
here and are not on the arithmetic path:

* `yacs.config.CfgNode`          (gmpi/utils/config.py:6)
* `torch.utils.tensorboard`      (gmpi/utils/tensorboard_utils.py:3)
* `lazy.lazy`                    (gmpi/core/camera.py:10)  -- a cached-property descriptor

We register minimal stand-ins in `sys.modules` before the import.  No reference source is
copied; the reference is executed from where it lies.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GMPI_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "gmpi", "core", "mpi.py"))


class _CachedAttr:
    """Stand-in for `lazy.lazy`: compute once per instance, then cache in the instance dict."""

    def __init__(self, fn):
        self._fn = fn
        self._name = fn.__name__
        self.__doc__ = fn.__doc__

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        val = self._fn(obj)
        obj.__dict__[self._name] = val
        return val


def _install_stubs():
    if "lazy" not in sys.modules:
        m = types.ModuleType("lazy")
        m.lazy = _CachedAttr
        sys.modules["lazy"] = m
    if "yacs" not in sys.modules:
        yacs = types.ModuleType("yacs")
        cfg = types.ModuleType("yacs.config")

        class CfgNode(dict):
            def __init__(self, *a, **k):
                super().__init__()

            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError as e:  # pragma: no cover
                    raise AttributeError(k) from e

            def __setattr__(self, k, v):
                self[k] = v

        cfg.CfgNode = CfgNode
        yacs.config = cfg
        sys.modules["yacs"] = yacs
        sys.modules["yacs.config"] = cfg
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter:  # never instantiated on this path
            def __init__(self, *a, **k):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb


def import_reference():
    """Returns the reference modules as a namespace: .mpi, .mpi_renderer, .camera, .cam_utils, .mpi_utils."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib

    ns = types.SimpleNamespace()
    ns.mpi = importlib.import_module("gmpi.core.mpi")
    ns.mpi_renderer = importlib.import_module("gmpi.core.mpi_renderer")
    ns.camera = importlib.import_module("gmpi.core.camera")
    ns.cam_utils = importlib.import_module("gmpi.utils.cam_utils")
    ns.mpi_utils = importlib.import_module("gmpi.utils.mpi_utils")
    ns.torch_utils = importlib.import_module("gmpi.utils.torch_utils")
    return ns


# Renderer kwargs of the reference's dataset presets (gmpi/curriculums.py:109-116, 133-140,
# 171-178 and configs/gmpi.yml:74-110), as render_video.py:168-189 passes them.
PRESETS = {
    "FFHQ": dict(plane_min_d=0.95, plane_max_d=1.12, cam_fov=12.6, sphere_center_z=1.0, sphere_r=1.0,
                 horizontal_mean=0.0, horizontal_std=0.289, vertical_mean=0.0, vertical_std=0.127,
                 cam_pose_n_truncated_stds=2),
    "MetFaces": dict(plane_min_d=0.95, plane_max_d=1.12, cam_fov=12.6, sphere_center_z=1.0, sphere_r=1.0,
                     horizontal_mean=0.0, horizontal_std=0.339, vertical_mean=0.0, vertical_std=0.133,
                     cam_pose_n_truncated_stds=2),
    "AFHQCat": dict(plane_min_d=2.55, plane_max_d=2.8, cam_fov=13.39, sphere_center_z=2.7, sphere_r=2.7,
                    horizontal_mean=0.0, horizontal_std=0.19, vertical_mean=0.0, vertical_std=0.15,
                    cam_pose_n_truncated_stds=3),
}


def make_reference_renderer(ns, preset="FFHQ", n_planes=32, align_corners=True, confined=True, **over):
    import torch

    kw = dict(PRESETS[preset])
    kw.update(
        n_mpi_planes=n_planes,
        plan_spatial_enlarge_factor=1.001,
        plane_distances_sample_method="inverse",
        cam_sample_method="truncated_gaussian",
        mpi_align_corners=align_corners,
        use_confined_volume=confined,
        device=torch.device("cpu"),
    )
    kw.update(over)
    return ns.mpi_renderer.MPIRenderer(**kw)
