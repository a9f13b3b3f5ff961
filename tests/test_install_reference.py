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
        assert issubclass(m.MPI, ml_gmpi_amd.MPI) and mr.MPI is m.MPI and issubclass(mr.MPIRenderer, ml_gmpi_amd.MPIRenderer)
        assert m.MPI.DEFAULT_RANGE_CHECK == "full" and ml_gmpi_amd.MPI.DEFAULT_RANGE_CHECK == "touched"   # no process-global default
        # the way render_video.py binds the name (`from gmpi.core.mpi_renderer import MPIRenderer`) after install()
        from gmpi.core.mpi_renderer import MPIRenderer
        assert issubclass(MPIRenderer, ml_gmpi_amd.MPIRenderer)
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


class _RecordingLibrary:
    """Stands in for libgmpi_render.so: every launch records a copy of its arguments and returns GMPI_OK."""
    records_only = True

    def __init__(self):
        self.calls = []

    def _copy_params(self, pref):
        import ctypes
        from ml_gmpi_amd import _lib
        src = ctypes.cast(pref, ctypes.POINTER(_lib.GmpiRenderParams)).contents
        dst = _lib.GmpiRenderParams()
        ctypes.memmove(ctypes.byref(dst), ctypes.byref(src), ctypes.sizeof(dst))
        return dst

    def gmpi_mpi_render_launch(self, pref, stream):
        self.calls.append(("render", self._copy_params(pref), stream))
        return 0

    def gmpi_rgba_range_check_launch(self, ptr, dtype, count, status, stream):
        self.calls.append(("range_check", (ptr, dtype, count, status), stream))
        return 0


def test_reference_render_drives_the_hip_mpi_through_its_seam(monkeypatch):
    """The reference's OWN `MPIRenderer.render` (mpi_renderer.py:387-469) with only `MPI` swapped: the call at
    mpi_renderer.py:451-461 (keyword arguments, per-view lists, c2w_mat, sphere_c) must reach `MPI.forward` and be marshalled
    into the GmpiRenderParams the C ABI expects.  The library is replaced by a recorder (no GPU in this container)."""
    import ref_import
    ns = ref_import.import_reference()
    import ml_gmpi_amd
    from ml_gmpi_amd import _lib
    rec = _RecordingLibrary()
    monkeypatch.setattr(_lib, "load_library", lambda: rec)
    seen = {}
    orig_forward = ml_gmpi_amd.MPI.forward

    def spy(self, **kw):
        seen.update(kw)
        return orig_forward(self, **kw)

    monkeypatch.setattr(ml_gmpi_amd.MPI, "forward", spy)
    B, D, S, T = 3, 6, 20, 24
    try:
        ml_gmpi_amd.install(patch_renderer=False)        # default: the reference's whole-volume range assertion
        with contextlib.redirect_stdout(io.StringIO()):
            r = ref_import.make_reference_renderer(ns, "FFHQ", D)
            r.set_cam(r.cam_fov, S, S)
        assert isinstance(r.mpi, ml_gmpi_amd.MPI) and r.mpi.range_check == "full"
        torch.manual_seed(5)
        rgba = torch.rand(B, D, 4, T, T)
        with contextlib.redirect_stdout(io.StringIO()):
            rgb, depth, c2w, angles = r.render(rgba, S, S, assert_not_out_of_last_plane=True)
    finally:
        ml_gmpi_amd.uninstall()
    assert ml_gmpi_amd.MPI.DEFAULT_RANGE_CHECK == "touched"
    # what the reference passed through the seam (mpi_renderer.py:451-461)
    assert set(seen) == {"batch_rgba", "batch_dhw", "batch_ray_dir", "batch_eye_pos", "batch_z_dir", "separate_background",
                         "assert_not_out_of_last_plane", "c2w_mat", "sphere_c"}
    assert seen["separate_background"] is None and seen["assert_not_out_of_last_plane"] is True
    assert len(seen["batch_ray_dir"]) == B and tuple(seen["batch_ray_dir"][0].shape) == (1, 3, S, S)
    assert tuple(seen["c2w_mat"].shape) == (B, 4, 4) and len(seen["sphere_c"]) == 3
    # what reached the C ABI: the exhaustive range check first, then ONE render launch
    assert [c[0] for c in rec.calls] == ["range_check", "render"]
    ptr, dtype, count, status_ptr = rec.calls[0][1]
    assert dtype == _lib.DTYPE_F32 and count == B * D * 4 * T * T
    p = rec.calls[1][1]
    assert (p.N, p.M, p.D, p.Ht, p.Wt, p.H, p.W, p.views_per_mpi) == (B, B, D, T, T, S, S, 1)
    assert p.view_to_mpi is None and p.rgba == ptr and p.status == status_ptr
    assert list(p.rgba_stride) == [D * 4 * T * T, 4 * T * T, T * T, T, 1]
    assert p.rgba_dtype == _lib.DTYPE_F32 and p.variant == _lib.VARIANT_AUTO
    # MPI.forward returns colour in [0,1]: the reference applies 2c-1 itself (mpi_renderer.py:467), so OUT_PM1 is off
    assert p.flags == (_lib.FLAG_ALIGN_CORNERS | _lib.FLAG_CHECK_LAST_PLANE | _lib.FLAG_CHECK_RANGE)
    assert p.transmittance_out is None and p.rgb_out and p.depth_out
    assert tuple(rgb.shape) == (B, 3, S, S) and tuple(depth.shape) == (B, 1, S, S) and tuple(angles.shape) == (B, 2)


def _stub_driver_imports(monkeypatch):
    """render_video.py imports torchvision.utils.save_image, and (through gmpi.utils.io_utils / train_helpers) imageio and torch_ema:
    none is installed here and none is on the render path -- minimal stand-ins, removed again by monkeypatch."""
    import sys
    import types

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        monkeypatch.setitem(sys.modules, name, m)
        return m

    if "torchvision" not in sys.modules:
        tv = stub("torchvision")
        tv.utils = stub("torchvision.utils", save_image=lambda *a, **k: None)
        stub("torchvision.transforms")
    if "imageio" not in sys.modules:
        stub("imageio", mimwrite=lambda *a, **k: None, imwrite=lambda *a, **k: None)
    if "torch_ema" not in sys.modules:
        stub("torch_ema", ExponentialMovingAverage=type("ExponentialMovingAverage", (), {"__init__": lambda self, *a, **k: None}))


class _EyeRecordingLibrary(_RecordingLibrary):
    """Also keeps the eye positions the launch was given (the tensors are alive during the call)."""

    def gmpi_mpi_render_launch(self, pref, stream):
        import ctypes
        p = self._copy_params(pref)
        eye = (ctypes.c_float * (3 * p.N)).from_address(p.eye_pos)
        self.calls.append(("render", p, list(eye)))
        return 0


def test_reference_generate_img_drives_the_installed_renderer(monkeypatch):
    """The reference's video driver itself -- `generate_img` (gmpi/eval/vis/render_video.py:19-132), unmodified, executed from where it
    lies -- after `install()`: a fake generator stands in for the StyleGAN2 network, the library is a recorder (no GPU here).  Every
    camera angle of the path must arrive at the C ABI as ONE render launch of the generator's volume, with the pose the angle asks
    for (h_mean = angle, stddev 0: render_video.py:100-113, 236-237)."""
    import importlib
    import sys
    import numpy as np
    import ref_import
    ref_import.import_reference()
    import ml_gmpi_amd
    from ml_gmpi_amd import _lib
    from ml_gmpi_amd.poses import gen_sphere_path
    _stub_driver_imports(monkeypatch)
    rec = _EyeRecordingLibrary()
    monkeypatch.setattr(_lib, "load_library", lambda: rec)
    D, S, T = 5, 24, 32
    monkeypatch.delitem(sys.modules, "gmpi.eval.vis.render_video", raising=False)
    try:
        ml_gmpi_amd.install()
        rv = importlib.import_module("gmpi.eval.vis.render_video")   # binds `from gmpi.core.mpi_renderer import MPIRenderer` AFTER install()
        assert issubclass(rv.MPIRenderer, ml_gmpi_amd.MPIRenderer)
        kw = dict(ref_import.PRESETS["FFHQ"])
        kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse",
                  cam_sample_method="truncated_gaussian", mpi_align_corners=True, use_xyz_ztype="depth", use_normalized_xyz=False,
                  normalized_xyz_range="-11", use_confined_volume=True, device=torch.device("cpu"))
        with contextlib.redirect_stdout(io.StringIO()):
            renderer = rv.MPIRenderer(**kw)                           # as render_video.py:168-189
            ws = renderer.get_xyz_interpolate_ws(D, D)                # :193
            renderer.set_cam(kw["cam_fov"], S, S)                     # :198
            xyz, xyz_n = renderer.get_xyz(T, T, ret_single_res=False, only_z=False)   # :200-205
        monkeypatch.setattr(rv, "mpi_renderer", renderer, raising=False)   # the module global generate_img reads (:104)
        metadata = dict(img_size=S, h_mean=0.0, v_mean=0.0, h_stddev=0.0, v_stddev=0.0, ray_start=kw["plane_min_d"], ray_end=kw["plane_max_d"])
        gen_calls = []

        def fake_generator(z, c, xyz_input, only_z, n_planes, z_interpolation_ws=None, truncation_psi=1.0):
            gen_calls.append(n_planes)
            g = torch.Generator().manual_seed(11)
            return torch.rand((z.shape[0], n_planes, 4, T, T), generator=g)

        angles = np.linspace(0.5, -0.5, 5).tolist()                   # :236-237 (100 angles there)
        torch.manual_seed(0)
        z = torch.randn((1, 16))
        with contextlib.redirect_stdout(io.StringIO()):
            imgs, tensors, depths, mpi_rgb, mpi_alpha = rv.generate_img(
                device=torch.device("cpu"), face_angles=angles, generator=fake_generator, z=z, mpi_xyz_input=xyz, metadata=metadata,
                horizontal_cam_move=True, mpi_xyz_only_z=False, z_interpolation_ws=ws, n_planes=D, truncation_psi=1.0,
                render_single_image=False, chunk_n_planes=-1, disable_tqdm=True)
    finally:
        ml_gmpi_amd.uninstall()
    assert gen_calls == [D, D]                                        # the whole-volume call and the one chunk (:42-75)
    renders = [c for c in rec.calls if c[0] == "render"]
    assert len(renders) == len(angles) and len(imgs) == len(angles) == len(depths) == len(tensors)
    # install(): the reference's whole-volume assertion (mpi_renderer.py:447-449) -- ONCE for the loop's one unchanged volume (round 6; rounds 1-5 paid the
    # exhaustive pass in front of every one of the path's views: tests/test_range_check_cache.py)
    assert [c[0] for c in rec.calls] == ["range_check", "render"] + ["render"] * (len(angles) - 1)
    for (_, p, eye), yaw in zip(renders, angles):
        assert (p.N, p.M, p.D, p.Ht, p.Wt, p.H, p.W) == (1, 1, D, T, T, S, S)
        assert p.flags & _lib.FLAG_OUT_PM1 and p.flags & _lib.FLAG_CHECK_LAST_PLANE and p.variant == _lib.VARIANT_AUTO
        c2w, yaws, pitches = gen_sphere_path(n_cams=1, sphere_center=renderer.sphere_center, sphere_r=renderer.sphere_r, yaw_mean=yaw,
                                             yaw_std=0.0, pitch_mean=0.0, pitch_std=0.0, n_truncated_stds=renderer.cam_pose_n_truncated_stds,
                                             flag_rnd=True, sample_method=renderer.cam_sample_method, given_yaws=None, given_pitches=None)
        want = torch.as_tensor(c2w)[0, :3, 3].float().tolist()
        assert eye == pytest.approx(want, abs=0), (yaw, eye, want)
    assert imgs[0].shape == (S, S, 3) and imgs[0].dtype == np.uint8 and depths[0].shape == (S, S, 1)
    assert tuple(mpi_rgb.shape) == (D, 3, T, T) and tuple(mpi_alpha.shape) == (D, 1, T, T)
