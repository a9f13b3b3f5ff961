"""`compute_depth` -- the alpha-compositing part of the reference's shading augmentation
(gmpi/core/light_renderer.py:82-100 `LightRenderer.compute_depth`) as one streaming HIP kernel.

The reference builds `[B, D+1, 1, H, W]` shifted alphas, a cumprod tensor, the weights and a weighted sum
(five full passes over the alpha planes); here the alpha channel is read once and the running transmittance lives in
a register.  Same arithmetic as the renderer's composite with the identity warp.
"""
import torch

from . import _lib
from .hip_mpi import _DTYPES


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
