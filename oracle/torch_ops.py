"""The reference's OP CHAIN for the render path, restated as the same sequence of PyTorch calls (fp32, CPU).

TEST / BASELINE INFRASTRUCTURE, like oracle/mpi_oracle.c: only tests/ and bench.py's `cpu_baseline` leg may import
this; the product never does.  /root/reference does not exist on the GPU box, so the reference itself cannot be timed
there; this module performs what the reference performs per call -- the same temporaries ([views*planes, ...] tensors),
the same ATen kernels (grid_sampler_2d, cumprod, einsum, the elementwise passes) in the same order:
    MPIRenderer.render  gmpi/core/mpi_renderer.py:444-467   (dhw expand, .float(), range assert, 2c-1)
    MPI.forward         gmpi/core/mpi.py:321-436            (expand per view, flatten views x planes, composite)
    homography          gmpi/core/mpi.py:60-153             (ray/plane intersection, F.grid_sample, depth/disparity)
tests/test_oracle_golden.py pins it against the fixtures the reference produced (<= 2e-6).
"""
import torch
import torch.nn.functional as F

NARROW = 0.95  # mpi.py:23 ALIGN_CORNERS_FALSE_NARROW_SCALE


def _homography(rgba, dhw, eye_pos, ray_dir, z_dir, align_corners):
    n, _, h, w = ray_dir.shape
    distance, height, width = dhw[:, :1], dhw[:, 1:2].unsqueeze(-1), dhw[:, 2:3].unsqueeze(-1)   # mpi.py:61-66
    with torch.no_grad():
        z_eye, z_ray = eye_pos[:, 2:3], ray_dir[:, 2:3]
        assert torch.all(distance >= z_eye[0])                                                 # mpi.py:70-72
        z_diff = (distance - z_eye).view(n, 1, 1, 1).expand(n, 1, h, w)                          # mpi.py:74-75
        scale = z_diff / z_ray                                                                   # mpi.py:76
        xyz = eye_pos.view(-1, 3, 1, 1) + ray_dir * scale                                        # mpi.py:79
        x, y = xyz[:, 0], xyz[:, 1]
        v = 2 * y / height                                                                       # mpi.py:89-90
        u = 2 * x / width
        if not align_corners:                                                                    # mpi.py:98-99
            v[(v >= -1) & (v <= 1)] = v[(v >= -1) & (v <= 1)] * NARROW
            u[(u >= -1) & (u <= 1)] = u[(u >= -1) & (u <= 1)] * NARROW
        grid = torch.stack([u, v], dim=-1)                                                       # mpi.py:101
    smp = F.grid_sample(rgba, grid, align_corners=align_corners, mode="bilinear", padding_mode="zeros")  # mpi.py:136-142
    rgb, alpha = smp[:, :3], smp[:, 3:4]
    with torch.no_grad():
        dist2depth = torch.einsum("nchw,nc->nhw", ray_dir, z_dir)                                # mpi.py:149
        depth = scale * dist2depth.view(n, 1, h, w)                                              # mpi.py:150
        disp = 1 / depth                                                                         # mpi.py:151
    return rgb, disp, alpha


def mpi_forward(batch_rgba, batch_dhw, ray_dir, eye_pos, z_dir, align_corners=True):
    """One view per MPI (the shape of MPIRenderer.render's call): batch_rgba [B,D,4,Ht,Wt], batch_dhw [B,D,3],
    ray_dir [B,3,H,W], eye_pos/z_dir [B,3] -> (color [B,3,H,W] in [0,1], depth [B,1,H,W])."""
    B, D = batch_rgba.shape[:2]
    _, _, H, W = ray_dir.shape
    # mpi.py:362-379: every per-view tensor replicated D-fold and flattened to [B*D, ...]
    flat_rgba = batch_rgba.reshape(B * D, 4, *batch_rgba.shape[-2:])
    flat_dhw = batch_dhw.reshape(B * D, 3)
    flat_ray = ray_dir.unsqueeze(1).expand(B, D, 3, H, W).reshape(B * D, 3, H, W)
    flat_eye = eye_pos.unsqueeze(1).expand(B, D, 3).reshape(B * D, 3)
    flat_z = z_dir.unsqueeze(1).expand(B, D, 3).reshape(B * D, 3)
    rgb, disp, alpha = _homography(flat_rgba, flat_dhw, flat_eye, flat_ray, flat_z, align_corners)   # mpi.py:399-409
    depth = 1 / disp                                                                                   # mpi.py:411
    rgb = rgb.view(B, D, 3, H, W)
    alpha = alpha.view(B, D, 1, H, W)
    depth = depth.view(B, D, 1, H, W)
    alphas_shifted = torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], 1)                 # mpi.py:421
    weights = alpha * torch.cumprod(alphas_shifted, 1)[:, :-1]                                         # mpi.py:423
    color = torch.sum(weights * rgb, 1)                                                                # mpi.py:430
    depth_out = torch.sum(weights * depth, 1)                                                          # mpi.py:434
    return color, depth_out


def renderer_render(batch_mpi_rgbas, static_dhw, ray_dir, eye_pos, z_dir, align_corners=True):
    """MPIRenderer.render after pose sampling (mpi_renderer.py:444-467): returns (rgb in [-1,1], depth)."""
    B = batch_mpi_rgbas.shape[0]
    batch_dhw = static_dhw.unsqueeze(0).expand(B, -1, -1)                                              # :444
    batch_mpi_rgbas = batch_mpi_rgbas.float()                                                          # :446
    assert torch.min(batch_mpi_rgbas) >= 0 and torch.max(batch_mpi_rgbas) <= 1                         # :447-449
    assert torch.min(batch_mpi_rgbas[:, :, 3]) >= 0 and torch.max(batch_mpi_rgbas[:, :, 3]) <= 1       # mpi.py:185-187
    color, depth = mpi_forward(batch_mpi_rgbas, batch_dhw, ray_dir, eye_pos, z_dir, align_corners)
    return 2 * color - 1, depth                                                                         # :467
