"""The same render through PyTorch-ROCm eager ops on the MI355X (F.grid_sample + cumprod, fp32: the op sequence of the
reference's gmpi/core/mpi.py, restated in tests/_torch_ref.py) next to the fused HIP kernel -- what a port that keeps
the reference's PyTorch path would get on this GPU.  One view at a time (the [D,4,H,W] temporaries).
usage: python tools/time_torch_path.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ml_gmpi_amd  # noqa: E402
from _torch_ref import torch_render  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for S, D in ((256, 96), (512, 96), (1024, 96)):
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
    r.set_cam(r.cam_fov, S, S)
    torch.manual_seed(0)
    vol = torch.rand((1, D, 4, S, S), device=dev)
    vol[:, -1, 3] = 1.0
    cam = r.sample_cam_poses(1, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    ray, eye, zd = torch.cat(cam[3]).to(dev), torch.cat(cam[4]).to(dev), torch.cat(cam[5]).to(dev)
    dhw = r._dhw_on_device().reshape(1, D, 3)
    with torch.no_grad():
        t_torch = timeit(lambda: torch_render(vol, dhw, ray, eye, zd, [0]), 5)
        t_hip = timeit(lambda: r.mpi.render_views(vol, dhw, ray, eye, zd, defer_status=True), 20)
        c_t, d_t = torch_render(vol, dhw, ray, eye, zd, [0])
        out = r.mpi.render_views(vol, dhw, ray, eye, zd, defer_status=True)
    err = float((out["color"] - c_t).abs().max())
    print(f"{S}^2 x {D} planes, 1 view fp32: torch eager ops {t_torch:.3f} ms ({S * S * D / t_torch / 1e3:.0f} Mpix*planes/s), "
          f"fused HIP kernel {t_hip:.3f} ms ({S * S * D / t_hip / 1e3:.0f} Mpix*planes/s), x{t_torch / t_hip:.0f}; max colour diff {err:.1e}")
