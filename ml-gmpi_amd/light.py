"""The reference's shading augmentation (gmpi/core/light_renderer.py `LightRenderer`) on the device.

`compute_depth` (light_renderer.py:82-100): the reference builds `[B, D+1, 1, H, W]` shifted alphas, a cumprod tensor,
the weights and a weighted sum (five full passes over the alpha planes); here the alpha channel is read once by one
streaming HIP kernel and the running transmittance lives in a register -- the renderer's composite with the identity
warp.  `LightRenderer.render` chains it with three more kernels (csrc/light_kernels.hip): Gaussian blur of the depth,
point cloud -> normals -> Lambert shading per texel, and `clip(rgb * shading, 0, 1)` over the volume.

The reference applies the augmentation inside the G-step (train.py:535-541): when the volume requires grad the same
kernels run under a `torch.autograd.Function` whose backward uses two more streaming kernels for the volume-sized ops
(clip(rgb*shading) and the alpha compositing of the depth) and torch autograd for the B*H*W middle (blur, normals, Lambert).
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, poses
from .hip_mpi import _DTYPES

EPS = 1e-8  # light_renderer.py:8


@torch.no_grad()
def compute_depth(mpi_alpha: torch.Tensor, plane_ds: torch.Tensor, want_transmittance: bool = False):
    """mpi_alpha [B, D, 1, H, W] (any float storage dtype; may be the strided view `mpi[:, :, 3:]` of an RGBA volume),
    plane_ds [D] or [D,1] plane distances -> depth [B, 1, H, W] (float32) [, transmittance [B,1,H,W]]."""
    if not mpi_alpha.is_cuda:
        raise _lib.GmpiError("compute_depth needs tensors on a ROCm device (no CPU path)")
    assert mpi_alpha.ndim == 5 and mpi_alpha.shape[2] == 1, f"{mpi_alpha.shape}"
    lib = _lib.load_library()
    if mpi_alpha.dtype not in _DTYPES:
        mpi_alpha = mpi_alpha.float()
    if mpi_alpha.stride(4) != 1 or any(s < 0 for s in mpi_alpha.stride()):
        mpi_alpha = mpi_alpha.contiguous()
    B, D, _, H, W = mpi_alpha.shape
    ds = plane_ds.reshape(-1).to(mpi_alpha.device, torch.float32).contiguous()
    assert ds.numel() == D, f"{ds.shape}, {D}"
    depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=mpi_alpha.device)
    T = torch.empty((B, 1, H, W), dtype=torch.float32, device=mpi_alpha.device) if want_transmittance else None
    with torch.cuda.device(mpi_alpha.device):
        _lib.check(lib.gmpi_alpha_depth_launch(
            mpi_alpha.data_ptr(), _DTYPES[mpi_alpha.dtype], mpi_alpha.stride(0), mpi_alpha.stride(1), mpi_alpha.stride(3),
            ds.data_ptr(), B, D, H, W, depth.data_ptr(), T.data_ptr() if T is not None else None,
            torch.cuda.current_stream(mpi_alpha.device).cuda_stream), "gmpi_alpha_depth_launch")
    return (depth, T) if want_transmittance else depth


def gaussian_kernel1d(ksize: int, sigma: float) -> torch.Tensor:
    """1-D kernel of torchvision.transforms.GaussianBlur (functional `_get_gaussian_kernel1d`): samples of the pdf on
    linspace(-(k-1)/2, (k-1)/2, k), normalised to sum 1, float32."""
    lim = (ksize - 1) * 0.5
    x = torch.linspace(-lim, lim, steps=ksize)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


_BLUR_MATRICES = {}


def _blur_matrix(n: int, k1d: torch.Tensor, device) -> torch.Tensor:
    """[n,n] matrix of the 1-D reflect-padded blur (row i = weights of output i over the inputs); cached."""
    k1d = k1d.detach().cpu()
    key = (n, tuple(k1d.tolist()), str(device))
    m = _BLUR_MATRICES.get(key)
    if m is None:
        k, r = k1d.numel(), k1d.numel() // 2
        src = torch.arange(n).view(n, 1) + torch.arange(k).view(1, k) - r          # [n,k] unpadded source index
        src = src.abs()
        src = torch.where(src >= n, 2 * (n - 1) - src, src)                       # padding_mode="reflect"
        m = torch.zeros((n, n), dtype=torch.float32)
        m.index_put_((torch.arange(n).view(n, 1).expand(n, k), src), k1d.to(torch.float32).view(1, k).expand(n, k), accumulate=True)
        m = m.to(device)
        if len(_BLUR_MATRICES) < 16:
            _BLUR_MATRICES[key] = m
    return m


def _blur_torch(depth: torch.Tensor, my: torch.Tensor, mx: torch.Tensor) -> torch.Tensor:
    """Differentiable restatement of gaussian_blur_kernel ([B,1,H,W]) for the backward of the pipeline's middle: the
    separable blur as two GEMMs with the reflect-padded 1-D operators `_blur_matrix` (equal to the 81-tap sum up to
    rounding; a 9x9 `F.conv2d` on B single-channel images costs milliseconds in MIOpen, the GEMMs microseconds)."""
    return torch.matmul(torch.matmul(my, depth), mx.t())


def _shading_torch(depth_blurred: torch.Tensor, xyz_last: torch.Tensor, light_dir: torch.Tensor, ka: float, kd: float) -> torch.Tensor:
    """torch restatement of light_shading_kernel (compute_pcl :102-120, get_normal :57-80, Lambert :163-190) -> [B,H,W]."""
    xyz = xyz_last.to(depth_blurred)[None]
    g = xyz * (depth_blurred[:, 0].unsqueeze(-1) / (xyz[..., 2:] + EPS))
    c = g[:, 1:-1, 1:-1]
    up, down, left, right = g[:, :-2, 1:-1], g[:, 2:, 1:-1], g[:, 1:-1, :-2], g[:, 1:-1, 2:]
    n = (torch.cross(up - c, left - c, dim=3) + torch.cross(left - c, down - c, dim=3)
         + torch.cross(down - c, right - c, dim=3) + torch.cross(right - c, up - c, dim=3))
    n = F.pad(n.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="replicate").permute(0, 2, 3, 1)
    n = n / (((n ** 2).sum(3, keepdim=True)) ** 0.5 + EPS)
    diffuse = (-1 * (n * light_dir.to(depth_blurred).view(-1, 1, 1, 3)).sum(3)).clamp(min=0)
    return ka + diffuse * kd


class _LightFunction(torch.autograd.Function):
    """LightRenderer.render as one autograd node: forward = the four kernels, backward = apply-backward kernel, torch
    autograd through the B*H*W middle, alpha-depth-backward kernel (adds into the alpha channel of the gradient)."""

    @staticmethod
    def forward(ctx, vol, renderer, plane_ds, xyz_last, light_dir, ka, kd):
        out, depth, T, shading = renderer._forward_kernels(vol.detach(), plane_ds, xyz_last, light_dir, ka, kd)
        dev = vol.device
        H, W = vol.shape[-2:]
        # everything the backward needs is put on the device here: CPU tensor ops inside the autograd worker thread start
        # a second OpenMP pool, and the two pools then fight for the cores (milliseconds per tiny host op)
        ctx.save_for_backward(vol.detach(), depth, T, shading, _blur_matrix(H, renderer._k1d, dev), _blur_matrix(W, renderer._k1d, dev),
                              xyz_last.to(dev, torch.float32), light_dir.to(dev, torch.float32),
                              plane_ds.reshape(-1).to(dev, torch.float32).contiguous())
        ctx.misc = (ka, kd, vol.dtype)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib.load_library()
        vol, depth, T, shading, my, mx, xyz_last, light_dir, ds = ctx.saved_tensors
        ka, kd, in_dtype = ctx.misc
        dev = vol.device
        B, D, _, H, W = vol.shape
        g_out = g_out.to(torch.float32).contiguous()
        g_rgba = torch.empty((B, D, 4, H, W), dtype=torch.float32, device=dev)
        g_shading = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        strides = (ctypes.c_int64 * 5)(*vol.stride())
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(lib.gmpi_light_apply_backward_launch(vol.data_ptr(), _DTYPES[vol.dtype], strides, shading.data_ptr(),
                                                            g_out.data_ptr(), g_rgba.data_ptr(), g_shading.data_ptr(), B, D, H, W,
                                                            stream), "gmpi_light_apply_backward_launch")
        with torch.enable_grad():
            d = depth.detach().requires_grad_(True)
            s = _shading_torch(_blur_torch(d, my, mx), xyz_last, light_dir, ka, kd)
            (g_depth,) = torch.autograd.grad(s, d, g_shading)
        g_depth = g_depth.contiguous()
        alpha = vol[:, :, 3:]
        plane = H * W
        with torch.cuda.device(dev):
            _lib.check(lib.gmpi_alpha_depth_backward_launch(
                alpha.data_ptr(), _DTYPES[vol.dtype], alpha.stride(0), alpha.stride(1), alpha.stride(3), ds.data_ptr(),
                T.data_ptr(), g_depth.data_ptr(), g_rgba.data_ptr() + 3 * plane * 4, D * 4 * plane, 4 * plane, W, B, D, H, W,
                stream), "gmpi_alpha_depth_backward_launch")
        return g_rgba.to(in_dtype), None, None, None, None, None, None


class LightRenderer:
    """Same constructor, attributes (`step`, `cur_ka`, `cur_kd`, `sphere_center`, ...) and method signatures as the
    reference class (light_renderer.py:11-199); tensors live on the ROCm device of `batch_mpi`."""

    def __init__(self, *, sphere_center_z, sphere_r, ka_max=1.0, kd_max=0.0, n_grow_iters=1000, l_h_mean=0.0, l_h_std=0.2,
                 l_v_mean=0.2, l_v_std=0.05, blur_ksize=9):
        self.ka_max, self.kd_max, self.n_grow_iters = ka_max, kd_max, n_grow_iters
        self.cur_ka = self.cur_kd = 0.0
        self.l_h_mean, self.l_h_std, self.l_v_mean, self.l_v_std = l_h_mean, l_h_std, l_v_mean, l_v_std
        self.sphere_center = torch.FloatTensor(np.array([0, 0, sphere_center_z]))
        self.sphere_r = sphere_r
        self.blur_ksize = blur_ksize
        self.blur_sigma = 0.3 * ((blur_ksize - 1) * 0.5 - 1) + 0.8  # OpenCV's rule, light_renderer.py:50
        self._k1d = gaussian_kernel1d(blur_ksize, self.blur_sigma)
        self.step = -1

    # -- pieces (same names as the reference) ------------------------------------------------------------------
    compute_depth = staticmethod(lambda mpi_alpha, plane_ds: compute_depth(mpi_alpha, plane_ds))

    @torch.no_grad()
    def blurrer_func(self, depth: torch.Tensor) -> torch.Tensor:
        """[B,1,H,W] -> blurred [B,1,H,W] (torchvision GaussianBlur: reflect padding, ksize x ksize)."""
        lib = _lib.load_library()
        d = depth.to(torch.float32).contiguous()
        B, _, H, W = d.shape
        out = torch.empty_like(d)
        k = self._k1d.to(d.device)
        with torch.cuda.device(d.device):
            _lib.check(lib.gmpi_light_blur_launch(d.data_ptr(), out.data_ptr(), B, H, W, k.data_ptr(), self.blur_ksize,
                                                  torch.cuda.current_stream(d.device).cuda_stream), "gmpi_light_blur_launch")
        return out

    @torch.no_grad()
    def shading(self, depth_blurred: torch.Tensor, xyz_last: torch.Tensor, light_direction: torch.Tensor, ka: float,
                kd: float) -> torch.Tensor:
        """compute_pcl + get_normal + Lambert term: [B,1,H,W], [H,W,3], [B,3] -> shading [B,H,W] = ka + kd*max(-n.l, 0)."""
        lib = _lib.load_library()
        d = depth_blurred.to(torch.float32).contiguous()
        B, _, H, W = d.shape
        xyz = xyz_last.to(d.device, torch.float32).reshape(H, W, 3).contiguous()
        ld = light_direction.to(d.device, torch.float32).reshape(B, 3).contiguous()
        out = torch.empty((B, H, W), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(lib.gmpi_light_shading_launch(d.data_ptr(), xyz.data_ptr(), ld.data_ptr(), float(ka), float(kd), B, H, W,
                                                     out.data_ptr(), torch.cuda.current_stream(d.device).cuda_stream),
                       "gmpi_light_shading_launch")
        return out

    # -- light_renderer.py:122-199 ------------------------------------------------------------------------------
    @torch.no_grad()
    def _forward_kernels(self, vol, plane_ds, xyz_last, light_direction, ka, kd):
        lib = _lib.load_library()
        dev = vol.device
        B, D, _, H, W = vol.shape
        depth, T = compute_depth(vol[:, :, 3:], plane_ds, want_transmittance=True)
        blurred = self.blurrer_func(depth)
        shading = self.shading(blurred, xyz_last, light_direction, ka, kd)
        out = torch.empty((B, D, 4, H, W), dtype=torch.float32, device=dev)
        strides = (ctypes.c_int64 * 5)(*vol.stride())
        with torch.cuda.device(dev):
            _lib.check(lib.gmpi_light_apply_launch(vol.data_ptr(), _DTYPES[vol.dtype], strides, shading.data_ptr(),
                                                   out.data_ptr(), B, D, H, W, torch.cuda.current_stream(dev).cuda_stream),
                       "gmpi_light_apply_launch")
        return out, depth, T, shading

    def render(self, batch_mpi: torch.Tensor, mpi_plane_dhws: torch.Tensor, mpi_tex_pix_xyz: torch.Tensor) -> torch.Tensor:
        """batch_mpi [B,D,4,H,W], mpi_plane_dhws [D,3], mpi_tex_pix_xyz [D,H,W,>=3] -> shaded MPI [B,D,4,H,W] float32
        (differentiable w.r.t. batch_mpi)."""
        if not batch_mpi.is_cuda:
            raise _lib.GmpiError("LightRenderer.render needs tensors on a ROCm device (no CPU path)")
        self.step += 1
        dev = batch_mpi.device
        vol = batch_mpi if batch_mpi.dtype in _DTYPES else batch_mpi.float()
        if vol.stride(4) != 1 or any(s < 0 for s in vol.stride()):
            vol = vol.contiguous()
        B = vol.shape[0]
        # light position on the sphere (consumes the torch RNG exactly as the reference's gen_sphere_path call)
        c2w, _, _ = poses.gen_sphere_path(n_cams=B, sphere_center=self.sphere_center, sphere_r=self.sphere_r,
                                          yaw_mean=self.l_h_mean, yaw_std=self.l_h_std, pitch_mean=self.l_v_mean,
                                          pitch_std=self.l_v_std, n_truncated_stds=2, flag_rnd=True,
                                          sample_method="truncated_gaussian", given_yaws=None, given_pitches=None)
        with poses.host_math():
            light_pos = c2w[:, :3, 3]
            light_pos = light_pos if isinstance(light_pos, torch.Tensor) else torch.FloatTensor(light_pos)
            light_direction = poses._unit(self.sphere_center.reshape(1, 3) - light_pos)  # towards the sphere centre
        cur_ratio = min(1.0, self.step / self.n_grow_iters)
        self.cur_ka, self.cur_kd = cur_ratio * self.ka_max, cur_ratio * self.kd_max
        plane_ds = mpi_plane_dhws[:, :1].detach().to(dev)
        xyz_last = mpi_tex_pix_xyz[-1, :, :, :3].detach()
        if torch.is_grad_enabled() and vol.requires_grad:
            return _LightFunction.apply(vol, self, plane_ds, xyz_last, light_direction, self.cur_ka, self.cur_kd)
        return self._forward_kernels(vol, plane_ds, xyz_last, light_direction, self.cur_ka, self.cur_kd)[0]
