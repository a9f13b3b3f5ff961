"""`install()` swaps the reference's MPI / MPIRenderer for the HIP ones (build container only: needs
/root/reference; skipped on the GPU box).  No kernel is launched here."""
import contextlib
import io

import pytest
import torch

from ref_import import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="needs /root/reference")


def test_install_patches_the_reference_modules_and_uninstall_restores():
    import ref_import
    ns = ref_import.import_reference()
    import ml_gmpi_amd
    orig_mpi, orig_renderer = ns.mpi.MPI, ns.mpi_renderer.MPIRenderer
    try:
        ml_gmpi_amd.install()
        import gmpi.core.mpi as m
        import gmpi.core.mpi_renderer as mr
        assert m.MPI is ml_gmpi_amd.MPI and mr.MPI is ml_gmpi_amd.MPI and mr.MPIRenderer is ml_gmpi_amd.MPIRenderer
        # the way render_video.py binds the name (`from gmpi.core.mpi_renderer import MPIRenderer`) after install()
        from gmpi.core.mpi_renderer import MPIRenderer
        assert MPIRenderer is ml_gmpi_amd.MPIRenderer
        import sys
        if "gmpi.core.light_renderer" in sys.modules:   # importable only where torchvision (or a stand-in) is present
            assert sys.modules["gmpi.core.light_renderer"].LightRenderer is ml_gmpi_amd.LightRenderer
    finally:
        ml_gmpi_amd.uninstall()
    assert ns.mpi.MPI is orig_mpi and ns.mpi_renderer.MPIRenderer is orig_renderer


def test_reference_renderer_with_patched_mpi_keeps_its_geometry():
    """patch_renderer=False: the reference's own MPIRenderer constructs our MPI through its seam (mpi_renderer.py:47)."""
    import ref_import
    ns = ref_import.import_reference()
    import ml_gmpi_amd
    try:
        ml_gmpi_amd.install(patch_renderer=False)
        with contextlib.redirect_stdout(io.StringIO()):
            r = ref_import.make_reference_renderer(ns, "FFHQ", 4)
        assert isinstance(r.mpi, ml_gmpi_amd.MPI) and r.mpi._align_corners is True
        ours = ml_gmpi_amd.make_renderer("FFHQ", n_planes=4, device=torch.device("cpu"))
        assert torch.equal(r.static_mpi_plane_dhws, ours.static_mpi_plane_dhws)
    finally:
        ml_gmpi_amd.uninstall()
