"""GPU tests of the per-view batch driver (render_video.py:95-130 / prepare_fake_data.py:58-86 patterns)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def test_render_path_equals_per_view_renders_and_oracle():
    from ml_gmpi_amd import ViewBatchDriver, make_renderer
    dev = torch.device("cuda:0")
    D, S = 12, 96
    r = make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
    rgba = torch.from_numpy(oracle.synth_rgba(5, (1, D, 4, S, S), last_alpha_one=True)).to(dev)
    yaws = np.linspace(0.5, -0.5, 7)          # video sweep (render_video.py:236-237 shape), pitch 0
    pitches = np.zeros(7)
    drv = ViewBatchDriver(r, batch=3)          # ragged last batch
    out = drv.render_path(rgba, S, yaws, pitches, to_uint8=True, want_transmittance=True)
    assert out["rgb"].shape == (7, 3, S, S) and out["img8"].shape == (7, S, S, 3) and out["dep8"].shape == (7, S, S, 1)
    # the reference loop: one render per angle, std 0 through the truncated-gaussian sampler
    for i in (0, 3, 6):
        rgb, dep, c2w, ang = r.render(rgba, S, S, horizontal_mean=float(yaws[i]), horizontal_std=0.0,
                                      vertical_mean=0.0, vertical_std=0.0)
        assert torch.equal(rgb[0], out["rgb"][i]) and torch.equal(dep[0], out["depth"][i])
        img = ((rgb[0].permute(1, 2, 0).cpu().numpy() + 1) / 2.0 * 255).astype(np.uint8)
        assert np.array_equal(img, out["img8"][i].cpu().numpy())
    # and against the oracle on the same camera tensors (rays made on the device)
    cam = r.sample_cam_poses(1, 0, 0, 0, 0, False, given_yaws=torch.tensor([[float(yaws[2])]]), given_pitches=torch.zeros(1, 1))
    orc = oracle.render(rgba.cpu(), r.static_mpi_plane_dhws.reshape(1, -1, 3), cam[3][0].cpu(), cam[4][0].cpu(), cam[5][0].cpu())
    assert np.abs(out["rgb"][2].cpu().numpy() - (2 * orc["color"][0] - 1)).max() <= 1e-5
    assert np.abs(out["depth"][2].cpu().numpy() - orc["depth"][0]).max() <= 1e-5
    assert np.abs(out["T"][2].cpu().numpy() - orc["T"][0]).max() <= 1e-5
    # the driver's pinned host buffers: same bytes as a plain copy, the same buffers on the next call
    h_img, h_dep = drv.to_host(out["img8"], out["dep8"])
    assert h_img.is_pinned() and not h_img.is_cuda and torch.equal(h_img, out["img8"].cpu()) and torch.equal(h_dep, out["dep8"].cpu())
    again = drv.to_host(out["img8"], out["dep8"])
    assert again[0].data_ptr() == h_img.data_ptr() and again[1].data_ptr() == h_dep.data_ptr()
    # round 6: the same frames through the pipelined path (uint8 epilogue per batch, copies on a second stream next to the next batch's render)
    piped = drv.render_path(rgba, S, yaws, pitches, to_uint8=True, to_host=True)
    assert piped["img8_host"].is_pinned() and torch.equal(piped["img8_host"], out["img8"].cpu()) and torch.equal(piped["dep8_host"], out["dep8"].cpu())
    assert torch.equal(piped["img8"], out["img8"]) and torch.equal(piped["rgb"], out["rgb"])
    sub = drv.render_path(rgba, S, yaws, pitches, indices=[5, 1, 2, 6], to_uint8=True, to_host=True)   # a rank's shard of the path
    assert torch.equal(sub["img8_host"], out["img8"].cpu()[[5, 1, 2, 6]])


def test_render_seeds_and_views_per_mpi_match_expanded_volume():
    from ml_gmpi_amd import ViewBatchDriver, make_renderer
    dev = torch.device("cuda:0")
    D, S, B, K = 6, 64, 3, 2
    r = make_renderer("MetFaces", n_planes=D, device=dev, on_out_of_plane="raise")
    rgba = torch.from_numpy(oracle.synth_rgba(9, (B, D, 4, S, S))).to(dev)
    torch.manual_seed(1)
    a = r.render(rgba, S, S, views_per_mpi=K)                           # indexed volume, B*K views
    torch.manual_seed(1)
    exp = rgba.unsqueeze(1).expand(-1, K, -1, -1, -1, -1).reshape(B * K, D, 4, S, S)  # prepare_fake_data.py:58-63
    b = r.render(exp, S, S)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    torch.manual_seed(1)
    c = ViewBatchDriver(r, batch=2).render_seeds(rgba, S, views_per_mpi=K)
    assert c[0].shape == a[0].shape  # different batching draws poses in a different order; shapes/finite only
    assert torch.isfinite(c[0]).all()


def test_single_process_gather_on_device():
    from ml_gmpi_amd import render_views_sharded
    dev = torch.device("cuda:0")
    frames = torch.arange(5 * 4 * 3 * 3, dtype=torch.float32, device=dev).reshape(5, 4, 3, 3)
    out = render_views_sharded(lambda idx: frames[idx], 5, rank=0, world_size=1)
    assert torch.equal(out, frames)
