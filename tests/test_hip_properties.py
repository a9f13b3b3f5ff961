"""Size-independent properties of the render path at BASELINE.json's full sizes (the CPU oracle
needs ~10 GB and minutes per 1024^2 x 96 view, so full-size parity is checked through invariants):

  * constant-colour volume:       C = c * (1 - T_final)                 (sum of weights = 1 - prod(1-a))
  * opaque far plane:             T_final == 0 and sum(weights) == 1 -> depth inside [near, far]
  * batch invariance:             N views in one launch == the same views one by one (bit-exact)
  * plane-split associativity:    composite(planes[:k]) (+) composite(planes[k:]) == composite(all)
                                  with (C1,Z1,T1)(+)(C2,Z2,T2) = (C1+T1*C2, Z1+T1*Z2, T1*T2)
  * identity of variants:         gather and LDS kernels agree bit-exactly in strict-order mode
  * sub-region parity:            a 64x64 pixel window of the full-size render == the oracle run on
                                  that window's rays (bit-exact in strict mode)
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def variants():
    from ml_gmpi_amd import _lib as L
    lib = L.load_library()
    # (all shapes of this file store fp32 or bf16: the band kernel takes them; "auto" = band kernel + gated tile kernel at these sizes)
    return (["gather"] + (["lds"] if lib.gmpi_query(3) > 0 else []) + (["wave"] if lib.gmpi_query(6) > 0 else [])
            + (["band"] if lib.gmpi_query(8) > 0 else []) + ["auto"])


def setup(S, D, B, preset="FFHQ", dtype=torch.float32, seed=0, last_alpha_one=False, extreme=False):
    from ml_gmpi_amd import make_renderer
    dev = torch.device(DEV)
    r = make_renderer(preset, n_planes=D, device=dev, on_out_of_plane="raise")
    r.set_cam(r.cam_fov, S, S)
    g = torch.Generator(device=dev).manual_seed(seed)
    rgba = torch.rand((B, D, 4, S, S), device=dev, generator=g, dtype=torch.float32)
    if last_alpha_one:
        rgba[:, -1, 3] = 1.0
    rgba = rgba.to(dtype)
    torch.manual_seed(seed)
    if extreme:  # the 2-sigma corner of the pose distribution: boxes shear, half-tile staging, texture borders in reach
        n = r.cam_pose_n_truncated_stds
        gy = torch.tensor([[(-1) ** i * n * r.horizontal_std] for i in range(B)], dtype=torch.float32)
        gp = torch.tensor([[(-1) ** (i // 2) * n * r.vertical_std] for i in range(B)], dtype=torch.float32)
        cam = r.sample_cam_poses(B, 0, 0, 0, 0, False, given_yaws=gy, given_pitches=gp)
    else:
        cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    dhw = r._dhw_on_device().expand(B, -1, -1).contiguous()
    return r, rgba, dhw, torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])


def run(r, rgba, dhw, ray, eye, zd, variant, strict=False, **kw):
    r.mpi.variant, r.mpi.strict_order = variant, strict
    with torch.no_grad():
        return r.mpi.render_views(rgba, dhw, ray, eye, zd, want_transmittance=True, check_last_plane=True, **kw)


@pytest.mark.parametrize("shape", [dict(S=1024, D=96, B=2, dtype=torch.bfloat16),   # config 3 shape (bf16 storage)
                                   dict(S=256, D=96, B=8, dtype=torch.float32),      # config 2
                                   dict(S=512, D=96, B=2, dtype=torch.float32)])     # config 4 per-GPU slice
def test_constant_colour_and_opaque_background(shape):
    r, rgba, dhw, ray, eye, zd = setup(last_alpha_one=True, seed=3, **shape)
    c = torch.tensor([0.25, 0.5, 0.75], device=rgba.device, dtype=rgba.dtype)
    rgba[:, :, :3] = c.view(1, 1, 3, 1, 1)
    for variant in variants():
        out = run(r, rgba, dhw, ray, eye, zd, variant)
        T, C, Z = out["T"], out["color"], out["depth"]
        want = c.float().view(1, 3, 1, 1) * (1.0 - T)
        assert float((C - want).abs().max()) <= 2e-6, variant
        assert float(T.abs().max()) <= 1e-9  # opaque last plane, every ray hits it (status checked)
        assert float(Z.min()) >= r.plane_min_d * 0.8 and float(Z.max()) <= r.plane_max_d * 1.25, variant


@pytest.mark.parametrize("shape", [dict(S=1024, D=96, B=2, dtype=torch.bfloat16), dict(S=256, D=96, B=8, dtype=torch.float32)])
def test_batch_invariance_and_variant_identity(shape):
    r, rgba, dhw, ray, eye, zd = setup(seed=4, **shape)
    ref = None
    for variant in variants():
        full = run(r, rgba, dhw, ray, eye, zd, variant, strict=True)
        if ref is None:
            ref = full
        else:  # both kernels: same bits in strict mode
            for k in ("color", "depth", "T"):
                assert torch.equal(full[k], ref[k]), (variant, k)
        for i in (0, rgba.shape[0] - 1):
            one = run(r, rgba[i:i + 1], dhw[i:i + 1], ray[i:i + 1], eye[i:i + 1], zd[i:i + 1], variant, strict=True)
            for k in ("color", "depth", "T"):
                assert torch.equal(one[k][0], full[k][i]), (variant, k, i)
        fast = run(r, rgba, dhw, ray, eye, zd, variant, strict=False)
        assert float((fast["color"] - full["color"]).abs().max()) <= 5e-6
        assert float((fast["depth"] - full["depth"]).abs().max()) <= 1e-5


def test_plane_split_associativity_full_size():
    r, rgba, dhw, ray, eye, zd = setup(S=1024, D=96, B=1, dtype=torch.float32, seed=5)
    k = 40
    for variant in variants():
        r.mpi.variant = variant
        whole = run(r, rgba, dhw, ray, eye, zd, variant)
        r.mpi.range_check = "touched"
        front = r.mpi.render_views(rgba[:, :k], dhw[:, :k].contiguous(), ray, eye, zd, want_transmittance=True)
        back = r.mpi.render_views(rgba[:, k:], dhw[:, k:].contiguous(), ray, eye, zd, want_transmittance=True)
        C = front["color"] + front["T"] * back["color"]
        Z = front["depth"] + front["T"] * back["depth"]
        T = front["T"] * back["T"]
        assert float((C - whole["color"]).abs().max()) <= 2e-6, variant
        assert float((Z - whole["depth"]).abs().max()) <= 4e-6, variant
        assert float((T - whole["T"]).abs().max()) <= 1e-6, variant


@pytest.mark.parametrize("shape", [dict(S=1024, D=96, B=1, dtype=torch.bfloat16), dict(S=1024, D=256, B=1, dtype=torch.float32, preset="MetFaces"),
                                   dict(S=1024, D=96, B=2, dtype=torch.bfloat16, extreme=True), dict(S=1024, D=96, B=1, dtype=torch.float32, extreme=True)])
def test_full_size_window_against_oracle(shape):
    """Oracle on the rays of a few 64x64 windows of the full-size image (the volume is full size): strict-order mode bit for bit, and -- round 6 --
    DEFAULT mode, the arithmetic every timed launch runs (divisions through correctly rounded reciprocals, FMA blend, 1 - w weights), within
    5e-6 colour ([0, 1] scale) / 1e-5 depth and transmittance (BASELINE.json's bar: 1e-5), for every kernel variant."""
    r, rgba, dhw, ray, eye, zd = setup(seed=6, **shape)
    vol = rgba.float().cpu().numpy()
    S = shape["S"]
    wins = [(0, 0), (S - 64, S - 64), (S // 2 - 32, S // 2 + 7), (13, S - 64)]
    orcs = {}
    for (y0, x0) in wins:
        win = ray[:, :, y0:y0 + 64, x0:x0 + 64].contiguous().cpu()
        orcs[(y0, x0)] = oracle.render(vol, dhw.cpu(), win, eye.cpu(), zd.cpu(), threads=True)
    for variant in variants():
        out = run(r, rgba, dhw, ray, eye, zd, variant, strict=True)
        fast = run(r, rgba, dhw, ray, eye, zd, variant, strict=False)
        for (y0, x0) in wins:
            orc = orcs[(y0, x0)]
            for key, bar in (("color", 5e-6), ("depth", 1e-5), ("T", 1e-5)):
                got = out[key][:, :, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
                assert np.array_equal(got, orc[key]), (variant, key, y0, x0, np.abs(got - orc[key]).max())
                dflt = fast[key][:, :, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
                assert np.abs(dflt - orc[key]).max() <= bar, ("default mode", variant, key, y0, x0, np.abs(dflt - orc[key]).max())


def test_config5_shape_auto_shares_the_views_between_band_and_tile_kernel():
    """BASELINE configs[4]'s shape (1024^2 x 256, fp32, MetFaces preset, transmittance output) under GMPI_VARIANT_AUTO with one near-frontal view
    and two views at the 2-sigma corner of the pose distribution: the band kernel cannot stage the tilted views, its table kernel hands them to
    the tile kernel through the view gate (gmpi_abi.hip) -- colour, depth AND transmittance of every view against the oracle windows, strict mode
    bit for bit and default mode within the bars; and the gate did what this test is for (workspace header: exactly the two tilted views left
    the band kernel)."""
    from ml_gmpi_amd import make_renderer, hip_mpi
    dev = torch.device(DEV)
    S, D, B = 1024, 256, 3
    r = make_renderer("MetFaces", n_planes=D, device=dev, on_out_of_plane="raise")
    r.set_cam(r.cam_fov, S, S)
    g = torch.Generator(device=dev).manual_seed(11)
    rgba = torch.rand((B, D, 4, S, S), device=dev, generator=g)
    rgba[:, -1, 3] = 1.0
    n = r.cam_pose_n_truncated_stds
    gy = torch.tensor([[0.04], [n * r.horizontal_std], [-n * r.horizontal_std]], dtype=torch.float32)
    gp = torch.tensor([[0.02], [n * r.vertical_std], [-n * r.vertical_std]], dtype=torch.float32)
    cam = r.sample_cam_poses(B, 0, 0, 0, 0, False, given_yaws=gy, given_pitches=gp)
    ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
    dhw = r._dhw_on_device().expand(B, -1, -1).contiguous()
    wins = [(0, 0), (S - 64, S - 64), (S // 2 - 32, S // 2 + 7)]
    out = run(r, rgba, dhw, ray, eye, zd, "auto", strict=True)
    fast = run(r, rgba, dhw, ray, eye, zd, "auto", strict=False)
    torch.cuda.synchronize()
    n_bands_view = (S // 128) * (S // 8)      # fp32 volumes: bands of 128 x 8 pixels
    ws = hip_mpi.workspace_of(dev)   # (of THIS stream: other tests leave workspaces of their own streams in the cache)
    hdr = ws[:4 * n_bands_view * B].view(torch.int32).view(B, n_bands_view)
    assert (hdr != 0).any(dim=1).cpu().tolist() == [False, True, True]
    for v in range(B):   # (one view's volume on the host at a time: 4.3 GB)
        vol = rgba[v:v + 1].cpu().numpy()
        for (y0, x0) in wins:
            win = ray[v:v + 1, :, y0:y0 + 64, x0:x0 + 64].contiguous().cpu()
            orc = oracle.render(vol, dhw[v:v + 1].cpu(), win, eye[v:v + 1].cpu(), zd[v:v + 1].cpu(), threads=True)
            for key, bar in (("color", 5e-6), ("depth", 1e-5), ("T", 1e-5)):
                got = out[key][v:v + 1, :, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
                assert np.array_equal(got, orc[key]), (v, key, y0, x0, np.abs(got - orc[key]).max())
                dflt = fast[key][v:v + 1, :, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
                assert np.abs(dflt - orc[key]).max() <= bar, ("default mode", v, key, y0, x0, np.abs(dflt - orc[key]).max())


def test_full_size_windows_config2_and_config4_against_oracle():
    """BASELINE config 2 (256^2 x 96, 8 views, fp32) and config 4 (512^2 x 96, 8 camera-path views of ONE MPI, yaw sweep
    0.5 ... -0.5 as render_video.py:236-237: the views_per_mpi > 1 tile interleave) at full size: strict mode, windows
    against the oracle, every view."""
    # ---- config 2 ----
    r, rgba, dhw, ray, eye, zd = setup(S=256, D=96, B=8, dtype=torch.float32, seed=8)
    vol = rgba.cpu().numpy()
    for variant in variants():
        out = run(r, rgba, dhw, ray, eye, zd, variant, strict=True)
        for (y0, x0) in [(0, 0), (192, 192), (101, 37)]:
            win = ray[:, :, y0:y0 + 64, x0:x0 + 64].contiguous().cpu()
            orc = oracle.render(vol, dhw.cpu(), win, eye.cpu(), zd.cpu(), threads=True)
            for key in ("color", "depth", "T"):
                got = out[key][:, :, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
                assert np.array_equal(got, orc[key]), ("cfg2", variant, key, y0, x0, np.abs(got - orc[key]).max())
    # ---- config 4: one MPI, 8 views ----
    from ml_gmpi_amd import make_renderer
    dev = torch.device(DEV)
    S, D, B = 512, 96, 8
    r = make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
    r.set_cam(r.cam_fov, S, S)
    g = torch.Generator(device=dev).manual_seed(9)
    rgba = torch.rand((1, D, 4, S, S), device=dev, generator=g)
    yaw = torch.linspace(0.5, -0.5, B).view(-1, 1)
    cam = r.sample_cam_poses(B, 0, 0, 0, 0, False, given_yaws=yaw, given_pitches=torch.zeros(B, 1))
    ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
    dhw = r._dhw_on_device().expand(1, -1, -1).contiguous()
    vol = rgba.cpu().numpy()
    v2m = np.zeros(B, dtype=np.int32)
    for variant in variants():
        out = run(r, rgba, dhw, ray, eye, zd, variant, strict=True, views_per_mpi=B)
        for (y0, x0) in [(0, 0), (S - 64, S - 64), (S // 2 - 32, 200)]:
            win = ray[:, :, y0:y0 + 64, x0:x0 + 64].contiguous().cpu()
            orc = oracle.render(vol, dhw.cpu(), win, eye.cpu(), zd.cpu(), view_to_mpi=v2m, threads=True)
            for key in ("color", "depth", "T"):
                got = out[key][:, :, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
                assert np.array_equal(got, orc[key]), ("cfg4", variant, key, y0, x0, np.abs(got - orc[key]).max())
