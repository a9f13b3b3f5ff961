"""Backward of the fused render at the reference's training sizes (gmpi.yml:78 D=32; curriculums.py:91-93 batch 8/4/4
at 256/512/1024): HIP events around forward and forward+backward through torch autograd (incl. the zero fill of the
gradient volume), for the tile-staged scatter (default) and the one-pixel-per-lane kernel (variant "gather").
usage: python tools/time_backward.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml_gmpi_amd  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for S, B in ((256, 8), (512, 4), (1024, 4)):
    D = 32
    for variant in ("auto", "gather"):
        r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise", kernel_variant=variant)
        vol = torch.rand((B, D, 4, S, S), device=dev)
        vol[:, -1, 3] = 1.0
        vol.requires_grad_(True)
        torch.manual_seed(0)
        r.set_cam(r.cam_fov, S, S)
        cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
        infos = dict(zip(["batch_yaws", "batch_pitches", "batch_tf_c2w", "batch_ray_dir", "batch_eye_pos", "batch_z_dir"], cam))
        g = torch.randn((B, 3, S, S), device=dev)

        def fwd():
            with torch.no_grad():
                return r.render(vol.detach(), S, S, given_cam_infos=infos, defer_status=True)

        def fwdbwd():
            vol.grad = None
            rgb = r.render(vol, S, S, given_cam_infos=infos, defer_status=True)[0]
            (rgb * g).sum().backward()

        tf, tfb = timeit(fwd), timeit(fwdbwd)
        print(f"{S}^2 x {D} planes, batch {B}, variant {variant:6s}: forward {tf:.3f} ms, forward+backward {tfb:.3f} ms "
              f"(backward ~{tfb - tf:.3f} ms = {B * S * S * D / (tfb - tf) / 1e3:.0f} Mpix*planes/s)")

# ---- the shading augmentation in front of it (train.py:535-541): LightRenderer.render forward and forward+backward ----
for S, B in ((256, 8), (512, 4), (1024, 4)):
    D = 32
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
    L = ml_gmpi_amd.LightRenderer(sphere_center_z=1.0, sphere_r=1.0, ka_max=0.9, kd_max=0.1, n_grow_iters=1)
    L.step = 10
    vol = torch.rand((B, D, 4, S, S), device=dev)
    vol[:, -1, 3] = 1.0
    vol.requires_grad_(True)
    xyz, _ = r.get_xyz(S, S, ret_single_res=True)
    g = torch.randn((B, D, 4, S, S), device=dev)

    def lfwd():
        with torch.no_grad():
            return L.render(vol.detach(), r.static_mpi_plane_dhws, xyz)

    def lfwdbwd():
        vol.grad = None
        (L.render(vol, r.static_mpi_plane_dhws, xyz) * g).sum().backward()

    tf, tfb = timeit(lfwd), timeit(lfwdbwd)
    gb = B * D * S * S * 16 / 1e9
    print(f"LightRenderer {S}^2 x {D} planes, batch {B}: forward {tf:.3f} ms, forward+backward {tfb:.3f} ms (volume {gb:.2f} GB fp32)")
