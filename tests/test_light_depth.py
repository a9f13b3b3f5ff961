"""`compute_depth` (LightRenderer.compute_depth, light_renderer.py:82-100): oracle pinned to a fixture made by the
reference; HIP kernel vs oracle on the GPU (bit-exact: same op sequence, no divisions)."""
import json

import numpy as np
import pytest
import torch

import oracle
from _util import load_npz


def _case():
    fx = load_npz("light_compute_depth.npz")
    m = fx["meta"]
    rgba = oracle.synth_rgba(m["seed"], (m["B"], m["D"], 4, m["S"], m["S"]))
    rgba[0, :, 3] = (rgba[0, :, 3] > 0.7).astype(np.float32)
    return fx, rgba


def test_oracle_matches_reference_compute_depth():
    fx, rgba = _case()
    depth, T = oracle.alpha_depth(rgba[:, :, 3:], fx["plane_ds"])
    assert np.abs(depth - fx["ref_depth"]).max() <= 2e-6  # torch.sum order only


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_hip_compute_depth_bit_exact(dtype):
    import ml_gmpi_amd
    fx, rgba = _case()
    vol = torch.from_numpy(rgba).cuda().to(dtype)
    alpha = vol[:, :, 3:]                     # strided view of the RGBA volume, as LightRenderer.render passes it
    assert not alpha.is_contiguous()
    depth, T = ml_gmpi_amd.compute_depth(alpha, torch.from_numpy(fx["plane_ds"]).reshape(-1, 1), want_transmittance=True)
    want_d, want_T = oracle.alpha_depth(alpha.float().cpu().numpy(), fx["plane_ds"])
    assert np.array_equal(depth.cpu().numpy(), want_d) and np.array_equal(T.cpu().numpy(), want_T)
    if dtype == torch.float32:
        assert np.abs(depth.cpu().numpy() - fx["ref_depth"]).max() <= 2e-6


@pytest.mark.gpu
def test_hip_compute_depth_full_size_streams():
    """1024^2 x 96 alpha planes inside an RGBA volume: consistency with the renderer's composite identity
    (sum of weights = 1 - T) and a throughput sanity check."""
    import ml_gmpi_amd
    vol = torch.rand((2, 96, 4, 1024, 1024), device="cuda")
    ds = torch.linspace(0.95, 1.12, 96)
    ones = torch.ones(96)
    depth, T = ml_gmpi_amd.compute_depth(vol[:, :, 3:], ds, want_transmittance=True)
    wsum, _ = ml_gmpi_amd.compute_depth(vol[:, :, 3:], ones, want_transmittance=True)
    assert float((wsum - (1 - T)).abs().max()) <= 2e-6
    assert float(depth.min()) >= 0.0 and float(depth.max()) <= 1.12 + 1e-5
