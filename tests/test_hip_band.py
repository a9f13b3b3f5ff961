"""GPU parity tests of the round-3 kernels -- the band kernel (GMPI_VARIANT_BAND, render_band.hip: 256 x 8 pixel bands over bf16 volumes, 128 x 8
over fp32 volumes) -- and of GMPI_VARIANT_AUTO's two-kernel launch that shares
the views between the band kernel and the tile kernel on the device.  Same bars as test_hip_parity.py: strict-order mode == oracle bit for bit,
default mode within 1e-5.  A bf16 volume is rendered by the oracle as its exact fp32 upcast (mpi_renderer.py:446)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from test_hip_parity import TOL, _lib, _random_case, hip_render

pytestmark = pytest.mark.gpu

BF = ("band", "auto")
E_VARIANT = -6  # GMPI_E_VARIANT (include/gmpi_render.h)


def _bf16_case(**cfg):
    rgba, dhw, ray, eye, zd = _random_case(**cfg)
    return rgba.to(torch.bfloat16), dhw, ray, eye, zd


def _check(stored, dhw, ray, eye, zd, variants=BF, **kw):
    orc = oracle.render(stored.float(), dhw, ray, eye, zd, threads=True)
    for variant in variants:
        strict = hip_render(stored, dhw, ray, eye, zd, variant=variant, strict=True, **kw)
        for k in ("color", "depth", "T"):
            assert np.array_equal(strict[k], orc[k]), (variant, k, np.abs(strict[k] - orc[k]).max())
        assert int(strict["status"][0]) == 0, variant
        fast = hip_render(stored, dhw, ray, eye, zd, variant=variant, **kw)
        assert np.abs(fast["color"] - orc["color"]).max() <= 0.5 * TOL, variant
        assert np.abs(fast["depth"] - orc["depth"]).max() <= TOL, variant
        assert np.abs(fast["T"] - orc["T"]).max() <= TOL, variant
    return orc


@pytest.mark.parametrize("cfg", [
    dict(seed=26, B=2, D=12, S=256),                        # fp32: two band columns of 128 pixels
    dict(seed=27, B=2, D=7, S=200, T=204),                  # fp32: ragged image, texture != image (a multiple of 4 texels)
    dict(seed=28, B=2, D=9, S=320, T=256, extreme=True),    # fp32: tilted cameras -> boxes that do not fit, rays that leave the texture
])
def test_band_parity_fp32(cfg):
    rgba, dhw, ray, eye, zd = _random_case(**cfg)
    _check(rgba, dhw, ray, eye, zd, variants=("band", "auto"))


def test_auto_shares_views_fp32():
    """The fp32 form of test_auto_shares_views_between_band_and_tile_kernels (AUTO's band path from gmpi_query(10) bands of 128 x 8 pixels)."""
    lib = _lib().load_library()
    S, B, D = 512, 4, 5
    assert B * ((S + 127) // 128) * ((S + 7) // 8) >= lib.gmpi_query(10)
    rgba, dhw, ray, eye, zd = _random_case(seed=33, B=B, D=D, S=S)
    _, _, ray_x, eye_x, zd_x = _random_case(seed=34, B=B, D=D, S=S, extreme=True)
    for n in (0, 2):
        ray[n], eye[n], zd[n] = ray_x[n], eye_x[n], zd_x[n]
    _check(rgba, dhw, ray, eye, zd, variants=("auto", "band"))


@pytest.mark.parametrize("cfg", [
    dict(seed=21, B=2, D=32, S=256),                        # one band column, 32 band rows per view
    dict(seed=22, B=1, D=12, S=512),                        # two band columns
    dict(seed=23, B=2, D=9, S=200, T=208),                  # image not a multiple of the band (256 x 8), texture != image
    dict(seed=24, B=3, D=5, S=72, T=64),                    # image smaller than one band; rays leave the texture (zeros padding)
    dict(seed=25, B=2, D=16, S=320, T=256, extreme=True),   # tilted cameras at the truncation limit: boxes that do not fit -> gather path
])
def test_band_parity_bf16(cfg):
    _check(*_bf16_case(**cfg))


@pytest.mark.parametrize("cfg", [
    dict(seed=71, B=2, D=16, S=256),                        # one band column
    dict(seed=72, B=1, D=9, S=512),                         # two band columns
    dict(seed=73, B=2, D=7, S=200, T=208),                  # ragged image, texture != image
    dict(seed=74, B=2, D=8, S=320, T=256, extreme=True),    # tilted cameras: boxes that do not fit, rays that leave the texture
])
def test_band_parity_fp16(cfg):
    """fp16 volumes (round 4): the d16_hi load yields the texel's fp16 pattern, converted inside the bilinear FMAs (v_fma_mix_f32) -- strict mode
    bit-identical to the oracle run on the exact fp32 upcast of the stored values, default mode within the bar."""
    rgba, dhw, ray, eye, zd = _random_case(**cfg)
    _check(rgba.to(torch.float16), dhw, ray, eye, zd, variants=("band", "auto"))


def test_band_range_check_fp16():
    rgba, dhw, ray, eye, zd = _random_case(seed=75, B=2, D=6, S=256)
    vol = rgba.to(torch.float16)
    for value in (1.25, -0.5, float("nan"), float("inf")):
        bad = vol.clone()
        bad[1, 3, 2, 100:140, 90:150] = value
        with pytest.raises(AssertionError):
            hip_render(bad, dhw, ray, eye, zd, variant="band")
    nz = vol.clone()
    nz[:, :, :, 60:200, 60:200][vol[:, :, :, 60:200, 60:200] < 0.25] = -0.0   # negative zeros are legal
    out = hip_render(nz, dhw, ray, eye, zd, variant="band", strict=True)
    assert int(out["status"][0]) == 0 and np.array_equal(out["color"], oracle.render(nz.float(), dhw, ray, eye, zd)["color"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_band_range_check_has_no_false_alarm_on_nan_and_huge_rays(dtype):
    """The [0,1] test folds every TAP REGISTER (render_band.hip tap_fold8), so a tap that reads something else than staged texels would raise a false
    "alpha out of [0, 1]".  Rays the coordinate chain turns into NaN tap addresses (the conversion saturates NaN to LDS address 0: the head of the
    staging buffers, always texels) or into addresses past the allocation (reads zeros) must leave the status clean -- behind a launch of ANOTHER kernel
    that left arbitrary bytes in LDS -- and must not disturb the other pixels (bit for bit in strict mode).  check_last_plane is off: such rays leave the
    last plane by definition."""
    rgba, dhw, ray, eye, zd = _random_case(seed=77, B=2, D=6, S=256)
    vol = rgba.to(dtype)
    clean = hip_render(vol, dhw, ray, eye, zd, variant="band", strict=True, check_last=False)
    bad = ray.clone()
    bad[0, :, 40:43, 100:170] = float("nan")          # NaN rays: a strip inside one band
    bad[1, 0, 200, 7] = float("inf")                  # x component infinite
    bad[1, :, 13, 250] = torch.tensor([3e30, -3e30, 1e-30])   # finite, absurd: coordinates beyond +-16384 -> the band's boxes are marked unfit (gather path)
    junk = torch.randn(1 << 22, device="cuda:0")
    for strict in (True, False):
        torch.sort(junk)                               # (a library kernel that uses LDS for its own data runs on every CU in front of the render)
        out = hip_render(vol, dhw, bad, eye, zd, variant="band", strict=strict, check_last=False)
        assert int(out["status"][0]) == 0, (strict, out["status"][:4])
        if strict:
            good = np.ones((2,) + clean["color"].shape[-2:], dtype=bool)     # every pixel but the ones with a bad ray
            good[0, 40:43, 100:170] = False
            good[1, 200, 7] = good[1, 13, 250] = False
            for n in range(2):
                for k in ("color", "depth", "T"):
                    assert np.array_equal(out[k][n][:, good[n]], clean[k][n][:, good[n]]), (n, k)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
def test_band_range_check_sees_every_sampled_texel(dtype):
    """The band kernel's [0,1] test runs on the taps (late round 5: a running maximum over the tap registers instead of an LDS read-back of the
    staged items): ONE bad texel in view -- any channel, i.e. either tap batch; rows that fall to either pixel of a thread -- trips it."""
    rgba, dhw, ray, eye, zd = _random_case(seed=76, B=2, D=5, S=256)
    vol = rgba.to(dtype)
    assert int(hip_render(vol, dhw, ray, eye, zd, variant="band")["status"][0]) == 0
    for i, (c, y, x, value) in enumerate([(0, 120, 120, 1.5), (1, 125, 77, -0.25), (2, 131, 190, float("nan")), (3, 122, 141, 1.0078125),
                                          (0, 99, 160, float("inf")), (3, 140, 101, -1e-3)]):
        bad = vol.clone()
        bad[i % 2, i % 5, c, y, x] = value
        with pytest.raises(AssertionError):
            hip_render(bad, dhw, ray, eye, zd, variant="band")
        with pytest.raises(AssertionError):
            hip_render(bad, dhw, ray, eye, zd, variant="band", strict=True)


def test_auto_shares_views_between_band_and_tile_kernels():
    """A launch large enough for AUTO's band path (>= gmpi_query(9) bands) whose views are partly frontal, partly tilted beyond what the band
    kernel stages: every view must come out once, bit-exact, whichever kernel the device-side gate hands it to."""
    lib = _lib().load_library()
    S, B, D = 512, 5, 6
    assert B * ((S + 255) // 256) * ((S + 7) // 8) >= lib.gmpi_query(9)
    rgba, dhw, ray, eye, zd = _bf16_case(seed=31, B=B, D=D, S=S)
    # views 1 and 3: the extreme poses of another draw (tilted); the rest stay as drawn (mild)
    _, _, ray_x, eye_x, zd_x = _random_case(seed=32, B=B, D=D, S=S, extreme=True)
    for n in (1, 3):
        ray[n], eye[n], zd[n] = ray_x[n], eye_x[n], zd_x[n]
    orc = _check(rgba, dhw, ray, eye, zd, variants=("auto", "band"))
    # the same launch with an explicit view -> MPI table (a permutation: view n samples MPI B-1-n) takes AUTO's band path as well
    perm = list(range(B - 1, -1, -1))
    orc_p = oracle.render(rgba.float(), dhw, ray, eye, zd, view_to_mpi=np.array(perm, dtype=np.int32), threads=True)
    out_p = hip_render(rgba, dhw, ray, eye, zd, variant="auto", strict=True, view_to_mpi=perm)
    assert np.array_equal(out_p["color"], orc_p["color"]) and np.array_equal(out_p["depth"], orc_p["depth"])
    # dirty workspace: whatever the scratch held, every view is rendered exactly once (the gate words are stamped per launch)
    from ml_gmpi_amd import hip_mpi
    for ws in hip_mpi._WORKSPACES.values():
        ws.fill_(0xFF)
    again = hip_render(rgba, dhw, ray, eye, zd, variant="auto", strict=True)
    assert np.array_equal(again["color"], orc["color"]) and np.array_equal(again["depth"], orc["depth"])
    for ws in hip_mpi._WORKSPACES.values():
        ws.random_(0, 256)
    again = hip_render(rgba, dhw, ray, eye, zd, variant="auto", strict=True)
    assert np.array_equal(again["color"], orc["color"]) and np.array_equal(again["depth"], orc["depth"])


def test_band_views_sharing_one_mpi():
    """views_per_mpi > 1 (video path: the views of one MPI are interleaved per band position) and the explicit view_to_mpi table."""
    rgba, dhw, ray, eye, zd = _bf16_case(seed=41, B=4, D=8, S=256)
    one, dhw1 = rgba[:2], dhw[:2]
    orc = oracle.render(one.float(), dhw1, ray, eye, zd, view_to_mpi=np.array([0, 0, 1, 1], dtype=np.int32), threads=True)
    for variant in BF:
        out = hip_render(one, dhw1, ray, eye, zd, variant=variant, strict=True, views_per_mpi=2)
        assert np.array_equal(out["color"], orc["color"]) and np.array_equal(out["depth"], orc["depth"]), variant
        out2 = hip_render(one, dhw1, ray, eye, zd, variant=variant, strict=True, view_to_mpi=[0, 0, 1, 1])
        assert np.array_equal(out2["color"], orc["color"]), variant
        out3 = hip_render(one, dhw1, ray[:3], eye[:3], zd[:3], variant=variant, strict=True, views_per_mpi=[2, 1])  # ragged groups
        assert np.array_equal(out3["color"], orc["color"][:3]), variant


def test_band_padded_rows_and_expanded_batch():
    rgba, dhw, ray, eye, zd = _bf16_case(seed=51, B=2, D=6, S=128)
    dev = torch.device("cuda:0")
    big = torch.rand((2, 6, 4, 140, 160), generator=torch.Generator().manual_seed(52)).to(torch.bfloat16).to(dev)
    view = big[:, :, :, 5:133, 16:144]  # row stride 160, 128 x 128 window starting at a 16-byte boundary
    assert not view.is_contiguous() and view.stride(4) == 1
    orc = oracle.render(view.cpu().float().contiguous(), dhw, ray, eye, zd)
    for variant in BF:
        out = hip_render(view, dhw, ray, eye, zd, variant=variant, strict=True)
        assert np.array_equal(out["color"], orc["color"]), variant
    exp = rgba[:1].to(dev).expand(2, -1, -1, -1, -1)
    orc = oracle.render(rgba[:1].float().expand(2, -1, -1, -1, -1).contiguous(), dhw, ray, eye, zd)
    for variant in BF:
        out = hip_render(exp, dhw, ray, eye, zd, variant=variant, strict=True)
        assert np.array_equal(out["color"], orc["color"]), variant
    odd = big[:, :, :, 5:133, 3:131]  # a window that does not start on a 16-byte boundary: refused when forced, rendered by AUTO
    orc = oracle.render(odd.cpu().float().contiguous(), dhw, ray, eye, zd)
    for variant in BF:
        out = hip_render(odd, dhw, ray, eye, zd, variant=variant, strict=True)
        assert np.array_equal(out["color"], orc["color"]), variant


def test_band_range_check_running_maximum():
    """mpi.py:185-187 on the texels the render touches: the band kernel folds the landed items into a running unsigned maximum and takes the
    verdict once per band; -0.0 (a legal value whose bit pattern sits above that of 1.0) sends the band to an exact re-test."""
    rgba, dhw, ray, eye, zd = _bf16_case(seed=61, B=2, D=6, S=256)
    orc = oracle.render(rgba.float(), dhw, ray, eye, zd)
    for variant in BF:
        bad = rgba.clone()
        bad[1, 3, 2, 100:140, 90:150] = 1.25
        with pytest.raises(AssertionError, match="alpha to be within"):
            hip_render(bad, dhw, ray, eye, zd, variant=variant)
        out = hip_render(bad, dhw, ray, eye, zd, variant=variant, range_check="off")
        assert int(out["status"][0]) == 0
        neg = rgba.clone()
        neg[0, 1, 3, 128, 128] = -0.5
        with pytest.raises(AssertionError, match="alpha to be within"):
            hip_render(neg, dhw, ray, eye, zd, variant=variant)
        nan = rgba.clone()
        nan[0, 2, 0, 77, 131] = float("nan")
        with pytest.raises(AssertionError):
            hip_render(nan, dhw, ray, eye, zd, variant=variant)
        nz = rgba.clone()
        nz[:, :, :, 60:200, 60:200][rgba[:, :, :, 60:200, 60:200] < 0.25] = -0.0   # plenty of negative zeros: legal
        orc_nz = oracle.render(nz.float(), dhw, ray, eye, zd)
        out = hip_render(nz, dhw, ray, eye, zd, variant=variant, strict=True)
        assert int(out["status"][0]) == 0 and np.array_equal(out["color"], orc_nz["color"]), variant
        both = nz.clone()
        both[1, 4, 1, 150, 150] = 2.0   # a violation among the negative zeros is still found
        with pytest.raises(AssertionError, match="alpha to be within"):
            hip_render(both, dhw, ray, eye, zd, variant=variant)
    assert orc["color"].shape == (2, 3, 256, 256)


def test_workspace_contract():
    """gmpi_render_workspace_bytes: 0 where no kernel wants scratch; the band kernel refuses to run without it (GMPI_E_VARIANT) and AUTO
    then uses the kernels that need none."""
    L = _lib()
    lib = L.load_library()
    dev = torch.device("cuda:0")
    rgba, dhw, ray, eye, zd = _bf16_case(seed=71, B=2, D=4, S=256)
    d = [t.to(dev).contiguous() for t in (rgba, dhw, ray, eye, zd)]
    color, depth = torch.empty((2, 3, 256, 256), device=dev), torch.empty((2, 1, 256, 256), device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)

    def params(variant, vol):
        p = L.GmpiRenderParams()
        p.struct_size = ctypes.sizeof(L.GmpiRenderParams)
        p.flags = L.FLAG_ALIGN_CORNERS | L.FLAG_STRICT_ORDER
        p.variant = L.VARIANTS[variant]
        p.rgba_dtype = {torch.float32: L.DTYPE_F32, torch.bfloat16: L.DTYPE_BF16, torch.float16: L.DTYPE_F16}[vol.dtype]
        p.N, p.M, p.D, p.Ht, p.Wt, p.H, p.W, p.views_per_mpi = 2, 2, 4, 256, 256, 256, 256, 1
        p.rgba = vol.data_ptr()
        for i, s in enumerate(vol.stride()):
            p.rgba_stride[i] = s
        p.dhw, p.ray_dir, p.eye_pos, p.z_dir = d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr()
        p.rgb_out, p.depth_out, p.status = color.data_ptr(), depth.data_ptr(), status.data_ptr()
        return p

    p = params("band", d[0])
    need = lib.gmpi_render_workspace_bytes(ctypes.byref(p))
    assert need > 0
    assert lib.gmpi_render_workspace_bytes(ctypes.byref(params("auto", d[0]))) == 0   # 64 bands: below AUTO's band threshold
    assert lib.gmpi_render_workspace_bytes(ctypes.byref(params("band", d[0].float()))) > need  # fp32: twice the bands (128 pixels wide)
    assert lib.gmpi_render_workspace_bytes(ctypes.byref(params("band", d[0].to(torch.float16)))) == need  # fp16 volume: the bf16 geometry (round 4)
    assert lib.gmpi_mpi_render_launch(ctypes.byref(p), None) == E_VARIANT            # no workspace
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), need - 1
    assert lib.gmpi_mpi_render_launch(ctypes.byref(p), None) == E_VARIANT            # too small
    p.workspace, p.workspace_bytes = ws.data_ptr() + 16, need
    assert lib.gmpi_mpi_render_launch(ctypes.byref(p), None) == E_VARIANT            # not 256-byte aligned
    p.workspace, p.workspace_bytes = ws.data_ptr(), need
    assert lib.gmpi_mpi_render_launch(ctypes.byref(p), None) == 0
    torch.cuda.synchronize()
    orc = oracle.render(rgba.float(), dhw, ray, eye, zd)
    assert np.array_equal(color.cpu().numpy(), orc["color"]) and int(status[0]) == 0
    vol16 = d[0].to(torch.float16)
    ph = params("band", vol16)
    ph.workspace, ph.workspace_bytes = ws.data_ptr(), need
    assert lib.gmpi_mpi_render_launch(ctypes.byref(ph), None) == 0                   # fp16 volumes: the band kernel's since round 4
    torch.cuda.synchronize()
    orc16 = oracle.render(vol16.float(), dhw, ray, eye, zd)
    assert np.array_equal(color.cpu().numpy(), orc16["color"]) and int(status[0]) == 0


def test_frontal_hint_changes_no_result():
    """GMPI_FLAG_HINT_FRONTAL is advisory: AUTO's choice between the strip and the tile kernel on a small 16-bit launch depends on it, the
    pixels (strict-order mode: bit for bit) do not -- whether the hint is true or not."""
    from ml_gmpi_amd import MPI
    rgba, dhw, ray, eye, zd = _bf16_case(seed=81, B=4, D=6, S=256)   # 1024 strips of 32 x 8 pixels: the launch size the hint decides
    orc = oracle.render(rgba.float(), dhw, ray, eye, zd)
    dev = torch.device("cuda:0")
    mpi = MPI(align_corners=True, variant="auto", strict_order=True, on_out_of_plane="raise")
    args = [t.to(dev) for t in (rgba, dhw, ray, eye, zd)]
    with torch.no_grad():
        for hint in (False, True):
            out = mpi.render_views(*args, check_last_plane=True, frontal_hint=hint)
            assert np.array_equal(out["color"].cpu().numpy(), orc["color"]) and np.array_equal(out["depth"].cpu().numpy(), orc["depth"]), hint


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_auto_renders_views_that_share_an_mpi_with_the_band_kernel_and_keeps_a_group_together(dtype):
    """Round 6: a camera path over ONE MPI (views_per_mpi = 8, render_video.py:95-130) under GMPI_VARIANT_AUTO.  Every view within the band kernel's reach:
    the band kernel renders the group (no gate word written).  One view beyond it: the table kernel hands the WHOLE group to the tile kernel (every view's
    gate word carries the launch's generation) -- two kernels that each render a few views of the group lose to one that renders them all.  Either way, and
    with GMPI_FLAG_HINT_OBLIQUE (the host knows a camera is more than 0.35 rad off the normal: tile kernel at once), the same pixels: strict-order mode
    bit-identical to the oracle, default mode inside the bars."""
    from ml_gmpi_amd import MPI, hip_mpi
    from ml_gmpi_amd.renderer import MPIRenderer, PRESETS
    lib = _lib().load_library()
    dev = torch.device("cuda:0")
    S, V, D = 512, 8, 5
    bw = 128 if dtype == torch.float32 else 256
    n_bands = V * ((S + bw - 1) // bw) * (S // 8)
    assert n_bands >= lib.gmpi_query(10 if dtype == torch.float32 else 9)
    kw = dict(PRESETS["FFHQ"])
    kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
              mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
    r = MPIRenderer(**kw)
    r.set_cam(r.cam_fov, S, S)
    rgba = torch.rand((1, D, 4, S, S), generator=torch.Generator().manual_seed(91)).to(dtype)
    dhw = r.static_mpi_plane_dhws.reshape(1, -1, 3).contiguous()
    for lim, want_gated in ((0.25, False), (0.5, True)):
        cam = r.sample_cam_poses(V, 0, 0, 0, 0, False, given_yaws=torch.linspace(lim, -lim, V).view(-1, 1), given_pitches=torch.zeros(V, 1))
        ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
        orc = oracle.render(rgba.float(), dhw, ray, eye, zd, view_to_mpi=np.zeros(V, dtype=np.int32), threads=True)
        for oblique in (False, True):
            for strict in (True, False):
                mpi = MPI(align_corners=True, variant="auto", strict_order=strict, range_check="touched", on_out_of_plane="raise")
                args = [t.to(dev) for t in (rgba, dhw, ray, eye, zd)]
                with torch.no_grad():
                    mpi.render_views(*args, views_per_mpi=V, check_last_plane=True, want_transmittance=True, oblique_hint=oblique)   # (the workspace exists now)
                    ws = hip_mpi.workspace_of(dev)
                    if ws is not None:
                        ws.zero_()
                    out = mpi.render_views(*args, views_per_mpi=V, check_last_plane=True, want_transmittance=True, oblique_hint=oblique)
                torch.cuda.synchronize()
                got = {k: out[k].cpu().numpy() for k in ("color", "depth", "T")}
                for k in got:
                    if strict:
                        assert np.array_equal(got[k], orc[k]), (lim, oblique, k, np.abs(got[k] - orc[k]).max())
                    else:
                        assert np.abs(got[k] - orc[k]).max() <= TOL, (lim, oblique, k)
                if not oblique:   # which kernel rendered the group: the gate words behind the band headers of the workspace
                    gate = hip_mpi.workspace_of(dev)[4 * n_bands:4 * (n_bands + V)].view(torch.int32).cpu().tolist()
                    if want_gated:
                        assert gate[0] != 0 and all(g == gate[0] for g in gate), gate
                    else:
                        assert all(g == 0 for g in gate), gate
