"""Test-only PyTorch restatement of the render (float64, differentiable w.r.t. rgba) -- used to check the fused
HIP backward against autograd.  Mirrors gmpi/core/mpi.py:26-153, 308-436 (grid under no_grad, grid_sample bilinear
zeros, cumprod weights).  Never imported by the product."""
import torch
import torch.nn.functional as F


def torch_render(rgba, dhw, ray_dir, eye, zdir, view_to_mpi, align_corners=True):
    """rgba [M,D,4,Ht,Wt] (requires_grad ok), dhw [M,D,3], ray_dir [N,3,H,W], eye/zdir [N,3] -> color [N,3,H,W], depth [N,1,H,W]."""
    N, _, H, W = ray_dir.shape
    M, D = dhw.shape[:2]
    colors, depths = [], []
    for n in range(N):
        m = int(view_to_mpi[n])
        with torch.no_grad():
            d, ph, pw = dhw[m, :, 0], dhw[m, :, 1], dhw[m, :, 2]
            zdiff = (d - eye[n, 2]).view(D, 1, 1)
            s = zdiff / ray_dir[n, 2][None]
            x = eye[n, 0] + ray_dir[n, 0][None] * s
            y = eye[n, 1] + ray_dir[n, 1][None] * s
            u = 2 * x / pw.view(D, 1, 1)
            v = 2 * y / ph.view(D, 1, 1)
            if not align_corners:
                v = torch.where((v >= -1) & (v <= 1), v * 0.95, v)
                u = torch.where((u >= -1) & (u <= 1), u * 0.95, u)
            grid = torch.stack([u, v], dim=-1)                          # [D,H,W,2]
            dot = (ray_dir[n] * zdir[n].view(3, 1, 1)).sum(0)           # [H,W]
            depth_k = s * dot[None]                                     # [D,H,W]
        smp = F.grid_sample(rgba[m], grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)  # [D,4,H,W]
        a = smp[:, 3:4]
        shifted = torch.cat([torch.ones_like(a[:1]), 1 - a + 1e-10], 0)
        w = a * torch.cumprod(shifted, dim=0)[:-1]
        colors.append((w * smp[:, :3]).sum(0))
        depths.append((w * depth_k[:, None]).sum(0))
    return torch.stack(colors), torch.stack(depths)


def torch_light_render(rgba, plane_ds, xyz_last, light_dir, ka, kd, k1d, eps=1e-8):
    """LightRenderer.render (light_renderer.py:82-199) for a given light direction, any float dtype, differentiable
    w.r.t. rgba [B,D,4,H,W]: compute_depth -> GaussianBlur -> compute_pcl -> get_normal -> Lambert -> clip(rgb*s)."""
    alpha = rgba[:, :, 3:]
    shifted = torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], 1)
    w = alpha * torch.cumprod(shifted, dim=1)[:, :-1]
    depth = torch.sum(w * plane_ds.to(rgba).reshape(1, -1, 1, 1, 1), dim=1)          # [B,1,H,W]
    r = k1d.numel() // 2
    k2d = torch.outer(k1d.to(rgba), k1d.to(rgba))[None, None]
    blurred = F.conv2d(F.pad(depth, (r, r, r, r), mode="reflect"), k2d)[:, 0]
    xyz = xyz_last.to(rgba)[None]
    g = xyz * (blurred.unsqueeze(-1) / (xyz[..., 2:] + eps))
    c = g[:, 1:-1, 1:-1]
    up, down, left, right = g[:, :-2, 1:-1], g[:, 2:, 1:-1], g[:, 1:-1, :-2], g[:, 1:-1, 2:]
    n = (torch.cross(up - c, left - c, dim=3) + torch.cross(left - c, down - c, dim=3)
         + torch.cross(down - c, right - c, dim=3) + torch.cross(right - c, up - c, dim=3))
    n = F.pad(n.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="replicate").permute(0, 2, 3, 1)
    n = n / (((n ** 2).sum(3, keepdim=True)) ** 0.5 + eps)
    diffuse = (-1 * (n * light_dir.to(rgba).view(-1, 1, 1, 3)).sum(3)).clamp(min=0)
    s = (ka + diffuse * kd).unsqueeze(1).unsqueeze(1)
    return torch.cat((torch.clip(rgba[:, :, :3] * s, min=0.0, max=1.0), alpha), dim=2)
