"""Host-side geometry mirror (poses, pinhole camera, plane extents) vs fixtures produced by the
reference (oracle/make_golden.py -> tests/golden/geometry.npz).  Everything here feeds the in-kernel
coordinate chain, so equality is BIT-EXACT (np.array_equal), not a tolerance.

Reference functions pinned: mpi_utils.py:21 sample_distance, :652/:787 compute_plane_dhws_*,
cam_utils.py:734 gen_sphere_path (+ :481, :571, :687), camera.py:182 _generate_rays_torch,
mpi_renderer.py:337 sample_cam_poses, torch_utils.py:51 truncated_normal (RNG consumption).
"""
import numpy as np
import pytest
import torch

from _util import load_npz
from ml_gmpi_amd.plane_geometry import sample_distance
from ml_gmpi_amd.renderer import MPIRenderer, PRESETS

GEO = load_npz("geometry.npz")
CPU = torch.device("cpu")


def cpu_renderer(preset, D, confined=True, **over):
    kw = dict(PRESETS[preset])
    kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse",
              cam_sample_method="truncated_gaussian", mpi_align_corners=True, use_confined_volume=confined, device=CPU)
    kw.update(over)
    return MPIRenderer(**kw)


@pytest.mark.parametrize("key", sorted(GEO["meta"]["presets"]))
def test_plane_dhws_bit_exact(key):
    m = GEO["meta"]["presets"][key]
    r = cpu_renderer(m["preset"], m["D"], m["confined"])
    assert r.static_mpi_plane_dhws.dtype == torch.float32
    assert np.array_equal(r.static_mpi_plane_dhws.numpy(), GEO[key]), key


@pytest.mark.parametrize("method", ["uniform", "log-uniform", "sqrt", "squared", "inverse"])
def test_sample_distance(method):
    assert np.array_equal(sample_distance(0.95, 1.12, 12, method), GEO[f"dist_{method}"])


def test_given_angle_poses_and_rays_bit_exact():
    r = cpu_renderer("FFHQ", 4)
    for i, m in enumerate(GEO["meta"]["poses"]):
        r.set_cam(r.cam_fov, m["S"], m["S"])
        cam = r.sample_cam_poses(1, 0.0, 0.0, 0.0, 0.0, False, given_yaws=torch.tensor([[m["yaw"]]]),
                                 given_pitches=torch.tensor([[m["pitch"]]]))
        assert np.array_equal(cam[2].numpy(), GEO[f"pose{i}_c2w"])
        assert np.array_equal(cam[3][0].numpy(), GEO[f"pose{i}_ray"])
        assert np.array_equal(cam[4][0].numpy(), GEO[f"pose{i}_eye"])
        assert np.array_equal(cam[5][0].numpy(), GEO[f"pose{i}_zdir"])
    ra = cpu_renderer("AFHQCat", 4)
    ra.set_cam(ra.cam_fov, 6, 6)
    cam = ra.sample_cam_poses(2, 0.0, 0.0, 0.0, 0.0, False, given_yaws=torch.tensor([[0.2], [-0.4]]),
                              given_pitches=torch.tensor([[0.1], [0.3]]))
    assert np.array_equal(cam[2].numpy(), GEO["afhq_c2w"])
    assert np.array_equal(torch.cat(cam[3]).numpy(), GEO["afhq_ray"])


def test_seeded_pose_sampling_draws_the_reference_poses():
    """Same torch RNG consumption as the reference: seeded calls give the same angles AND leave the
    generator at the same position (checked with the next torch.rand)."""
    r = cpu_renderer("FFHQ", 4)
    r.set_cam(r.cam_fov, 4, 4)
    for j, m in enumerate(GEO["meta"]["samples"]):
        r.cam_sample_method = m["method"]
        torch.manual_seed(m["seed"])
        cam = r.sample_cam_poses(m["B"], 0.05, 0.289, -0.02, 0.127, m["random"])
        assert np.array_equal(cam[0].numpy(), GEO[f"sample{j}_yaws"]), m
        assert np.array_equal(cam[1].numpy(), GEO[f"sample{j}_pitches"]), m
        assert np.array_equal(cam[2].numpy(), GEO[f"sample{j}_c2w"]), m
        assert np.array_equal(torch.rand(2).numpy(), GEO[f"sample{j}_after"]), m
    r.cam_sample_method = "truncated_gaussian"
    torch.manual_seed(200)
    cam = r.sample_cam_poses(1, 0.25, 0.0, 0.0, 0.0, True)  # video-style: std 0
    assert np.array_equal(cam[0].numpy(), GEO["video_yaws"])
    assert np.array_equal(cam[2].numpy(), GEO["video_c2w"])
    assert np.array_equal(torch.rand(2).numpy(), GEO["video_after"])


def test_constructor_advances_rng_like_the_reference():
    """The reference draws 2 x 10001 torch.rand((1,1)) while building the plane geometry; a seeded
    script must see the same RNG state afterwards (fixture: reference renderer built under seed 77)."""
    import os
    from ref_import import reference_available
    if not reference_available():
        pytest.skip("needs /root/reference")
    import contextlib, io
    import ref_import
    ns = ref_import.import_reference()
    torch.manual_seed(77)
    with contextlib.redirect_stdout(io.StringIO()):
        ref_import.make_reference_renderer(ns, "FFHQ", 4)
    want = torch.rand(3)
    torch.manual_seed(77)
    cpu_renderer("FFHQ", 4)
    assert torch.equal(torch.rand(3), want)


def test_renderer_attributes_and_helpers():
    r = cpu_renderer("FFHQ", 8)
    for name in ("mpi", "cam", "render_h", "render_w", "static_mpi_plane_dhws", "dynamic_mpi_plane_dhws", "device",
                 "n_mpi_planes", "sphere_center", "sphere_r", "cam_fov"):
        assert hasattr(r, name), name
    ws = r.get_xyz_interpolate_ws(4, 8)
    assert ws.shape == (8, 6) and torch.allclose(ws.sum(1), torch.ones(8), atol=1e-5)
    z, nz = r.get_xyz(16, 16, only_z=True)
    assert z.shape == (8, 1, 1, 1) and float(nz.min()) >= -1 - 1e-6 and float(nz.max()) <= 1 + 1e-6
    xyz, _ = r.get_xyz(16, 16)
    assert xyz.shape == (8, 16, 16, 3)


def test_host_math_context_restores_the_intra_op_thread_count():
    """poses.host_math(): tiny pose tensors run on one thread (the 128-thread pool of the GPU hosts makes them cost
    milliseconds); the process-wide thread count is restored, nesting is a no-op, results do not depend on it."""
    from ml_gmpi_amd import poses
    before = torch.get_num_threads()
    torch.manual_seed(4)
    a = poses.truncated_normal(8, 0.0, 0.3, 2)
    with poses.host_math():
        inner = torch.get_num_threads()
        with poses.host_math():
            assert torch.get_num_threads() == 1
        torch.manual_seed(4)
        b = poses.truncated_normal(8, 0.0, 0.3, 2)
    assert inner == 1 and torch.get_num_threads() == before
    assert torch.equal(a, b)


def test_look_ahead_poses_are_the_poses_the_calls_would_have_drawn():
    """MPIRenderer's guarded pose look-ahead (started by itself when a request repeats, or by prefetch_poses): same poses, same RNG
    stream as drawing inside each call, whatever else the program draws in between."""
    import ml_gmpi_amd
    from ml_gmpi_amd import poses

    def plain(r, n, hm, hs, vm, vs):  # one call the way the reference does it: straight from the default generator
        c2w, yaws, pitches = poses.gen_sphere_path(n_cams=n, sphere_center=r.sphere_center, sphere_r=r.sphere_r, yaw_mean=hm, yaw_std=hs,
                                                   pitch_mean=vm, pitch_std=vs, n_truncated_stds=r.cam_pose_n_truncated_stds, flag_rnd=True,
                                                   sample_method=r.cam_sample_method)
        return yaws, pitches, torch.FloatTensor(c2w)

    for method in ("truncated_gaussian", "uniform", "normal"):
        r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=4, device=torch.device("cpu"), ray_backend="torch", cam_sample_method=method)
        r.set_cam(r.cam_fov, 8, 8)
        args = (3, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std)
        # 1. a loop of identical calls (the look-ahead starts at the second call and is refilled when it runs dry) with an unrelated
        #    draw after every third call (which invalidates the queue: the call after it must draw for itself)
        torch.manual_seed(77)
        want, noise_want = [], []
        for i in range(25):
            want.append(plain(r, *args))
            if i % 3 == 2:
                noise_want.append(torch.rand(2))
        end_want = torch.rand(3)
        torch.manual_seed(77)
        got, noise_got = [], []
        for i in range(25):
            got.append(r.sample_cam_poses(*args, True)[:3])
            if i % 3 == 2:
                noise_got.append(torch.rand(2))
        end_got = torch.rand(3)
        assert torch.equal(end_want, end_got) and all(torch.equal(a, b) for a, b in zip(noise_want, noise_got))
        for w, g in zip(want, got):
            assert all(torch.equal(a, b) for a, b in zip(w, g)), method
        # 2. an undisturbed loop: the queue is actually used (and refilled, deeper each time) -- after the 16 calls for which a request whose
        #    look-ahead was invalidated (part 1) draws for itself
        torch.manual_seed(5)
        want = [plain(r, *args) for _ in range(70)]
        torch.manual_seed(5)
        used = 0
        for w in want:
            before = None if r._spec is None else r._spec["idx"]
            g = r.sample_cam_poses(*args, True)[:3]
            used += int(r._spec is not None and before is not None and r._spec["idx"] == before + 1)
            assert all(torch.equal(a, b) for a, b in zip(w, g)), method
        assert used >= 40 and r._spec is not None and r._spec["n"] >= 16, (used, r._spec and r._spec["n"])
        # 3. prefetch_poses does not advance the generator; re-seeding between calls drops the queue
        torch.manual_seed(9)
        r.prefetch_poses(4, 3)
        probe = torch.get_rng_state()
        torch.manual_seed(9)
        assert torch.equal(probe, torch.get_rng_state())
        a = r.sample_cam_poses(*args, True)
        torch.manual_seed(123)
        w = plain(r, *args)
        torch.manual_seed(123)
        b = r.sample_cam_poses(*args, True)
        assert all(torch.equal(x, y) for x, y in zip(w, b[:3])) and not torch.equal(a[2], b[2])
        # 4. other arguments: the call draws for itself
        torch.manual_seed(5)
        w = plain(r, 2, 0.0, 0.1, 0.0, 0.1)
        torch.manual_seed(5)
        b = r.sample_cam_poses(2, 0.0, 0.1, 0.0, 0.1, True)
        assert all(torch.equal(x, y) for x, y in zip(w, b[:3]))


def test_zero_deviation_draws_are_cached_and_identical_to_the_general_path():
    """render_video.py:95-130 calls render(h_mean=angle, h_stddev=0) per view: a "random" draw whose result is the mean itself.  The renderer
    serves such requests from a per-request cache -- same pose bits and the same generator state afterwards as the general path
    (poses.gen_sphere_path with flag_rnd=True, itself pinned to the reference by golden/geometry.npz), first call and cached call alike."""
    import ml_gmpi_amd
    from ml_gmpi_amd.poses import gen_sphere_path
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=8, device=torch.device("cpu"), ray_backend="torch")
    for ang, vm, B in ((0.37, 0.0, 1), (-0.5, 0.1, 3), (0.37, 0.0, 1)):   # (the third request repeats the first: a cache hit)
        torch.manual_seed(77)
        c2w_ref, yaws_ref, pitches_ref = gen_sphere_path(
            n_cams=B, sphere_center=r.sphere_center, sphere_r=r.sphere_r, yaw_mean=ang, yaw_std=0.0, pitch_mean=vm, pitch_std=0.0,
            n_truncated_stds=r.cam_pose_n_truncated_stds, flag_rnd=True, sample_method=r.cam_sample_method)
        state_ref = torch.get_rng_state()
        torch.manual_seed(77)
        yaws, pitches, c2w, angles = r._draw_poses(B, ang, 0.0, vm, 0.0, True)
        assert torch.equal(torch.get_rng_state(), state_ref)
        assert torch.equal(yaws, yaws_ref) and torch.equal(pitches, pitches_ref)
        assert np.array_equal(c2w.numpy(), torch.FloatTensor(c2w_ref).numpy())
        assert torch.equal(angles, torch.cat([pitches_ref, yaws_ref], -1))
    assert len(r._det_poses) == 2
