"""Edge cases of the render path on the GPU: degenerate sizes, ragged shapes, wild inputs.  Everything is
checked against the CPU oracle bit-exactly in strict-order mode (and must never fault)."""
import numpy as np
import pytest
import torch

import oracle
from test_hip_parity import hip_render, variants

pytestmark = pytest.mark.gpu


def _cam(N, H, W, seed=0, tilt=0.1):
    """Synthetic pinhole-ish rays (not from the renderer): unit vectors around +z with a per-view tilt."""
    g = np.random.default_rng(seed)
    ys, xs = np.meshgrid(np.linspace(-0.11, 0.11, H), np.linspace(-0.11, 0.11, W), indexing="ij")
    rays, eyes, zds = [], [], []
    for n in range(N):
        a = tilt * (g.random() - 0.5) * 2
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        d = np.stack([xs, ys, np.ones_like(xs)]).reshape(3, -1)
        d = d / np.linalg.norm(d, axis=0)
        rays.append((R @ d).reshape(3, H, W))
        eyes.append([-np.sin(a), 0.0, 1 - np.cos(a)])
        zds.append(R[:, 2])
    return (np.stack(rays).astype(np.float32), np.array(eyes, np.float32), np.array(zds, np.float32))


def _dhw(M, D, near=0.95, far=1.12, ext=0.25, last=0.5):
    d = 1.0 / np.linspace(1 / near, 1 / far, D) if D > 1 else np.array([far])
    t = np.stack([d, np.full(D, ext), np.full(D, ext)], 1)
    t[-1, 1:] = last
    return np.broadcast_to(t[None], (M, D, 3)).astype(np.float32).copy()


@pytest.mark.parametrize("shape", [
    dict(N=1, D=1, Ht=4, Wt=4, H=1, W=1),          # single plane, single pixel
    dict(N=2, D=3, Ht=8, Wt=16, H=5, W=7),         # tiny, non-square texture and image
    dict(N=1, D=2, Ht=40, Wt=24, H=33, W=65),      # W one past a tile boundary
    dict(N=3, D=97, Ht=32, Wt=32, H=17, W=31),     # one plane more than a geometry chunk (96)
    dict(N=1, D=4, Ht=12, Wt=20, H=300, W=20),     # image much finer than the texture (magnification)
    dict(N=1, D=4, Ht=512, Wt=512, H=24, W=24),    # texture much finer than the image: boxes do not fit -> gather chunk
])
def test_degenerate_and_ragged_shapes(shape):
    N, D, Ht, Wt, H, W = (shape[k] for k in ("N", "D", "Ht", "Wt", "H", "W"))
    rgba = oracle.synth_rgba(31, (N, D, 4, Ht, Wt))
    ray, eye, zd = _cam(N, H, W, seed=3)
    dhw = _dhw(N, D)
    for ac in (True, False):
        orc = oracle.render(rgba, dhw, ray, eye, zd, align_corners=ac)
        for variant in variants():
            out = hip_render(rgba, dhw, ray, eye, zd, ac=ac, variant=variant, strict=True, check_last=False)
            for k in ("color", "depth", "T"):
                assert np.array_equal(out[k], orc[k]), (shape, ac, variant, k, np.abs(out[k] - orc[k]).max())
            fast = hip_render(rgba, dhw, ray, eye, zd, ac=ac, variant=variant, check_last=False)
            assert np.abs(fast["color"] - orc["color"]).max() <= 5e-6 and np.abs(fast["depth"] - orc["depth"]).max() <= 1e-5


def test_zero_views_is_a_no_op():
    from ml_gmpi_amd import MPI
    dev = torch.device("cuda:0")
    mpi = MPI()
    with torch.no_grad():
        out = mpi.render_views(torch.rand(1, 2, 4, 8, 8, device=dev), torch.rand(1, 2, 3, device=dev),
                               torch.empty(0, 3, 8, 8, device=dev), torch.empty(0, 3, device=dev),
                               torch.empty(0, 3, device=dev), views_per_mpi=[0])
    assert out["color"].shape == (0, 3, 8, 8)


def test_wild_rays_do_not_fault_and_match_oracle_where_defined():
    """NaN / zero / huge ray components: the reference produces garbage for such pixels; we must not crash and
    every pixel with sane rays must still be exact."""
    N, D, S = 1, 6, 64
    rgba = oracle.synth_rgba(33, (N, D, 4, S, S))
    ray, eye, zd = _cam(N, S, S, seed=4)
    dhw = _dhw(N, D)
    bad = ray.copy()
    bad[0, :, 3, 5] = np.nan
    bad[0, 2, 10, 10] = 0.0            # ray parallel to the planes -> division by zero
    bad[0, 0, 20, 20] = 1e30
    bad[0, :, 63, 63] = [0.0, 0.0, -1.0]  # looking backwards
    good = np.ones((S, S), bool)
    for (y, x) in [(3, 5), (10, 10), (20, 20), (63, 63)]:
        good[y, x] = False
    orc = oracle.render(rgba, dhw, ray, eye, zd)
    for variant in variants():
        out = hip_render(rgba, dhw, bad, eye, zd, variant=variant, strict=True, check_last=False, range_check="off")
        for k in ("color", "depth", "T"):
            assert np.array_equal(out[k][0][..., good], orc[k][0][..., good]), (variant, k)


def test_alpha_exact_zero_and_one():
    """a == 1 exercises the 1e-10 term of mpi.py:421, a == 0 leaves T untouched."""
    N, D, S = 1, 5, 48
    rgba = oracle.synth_rgba(35, (N, D, 4, S, S))
    rgba[:, :, 3] = (rgba[:, :, 3] > 0.6).astype(np.float32)
    rgba[:, -1, 3] = 1.0
    ray, eye, zd = _cam(N, S, S, seed=5)
    dhw = _dhw(N, D)
    orc = oracle.render(rgba, dhw, ray, eye, zd)
    for variant in variants():
        out = hip_render(rgba, dhw, ray, eye, zd, variant=variant, strict=True, check_last=False)
        assert np.array_equal(out["color"], orc["color"]) and np.array_equal(out["T"], orc["T"])
        assert float(out["T"].max()) <= 1e-6  # opaque last plane: bilinear weights sum to 1 within an ulp


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_guard_band_no_read_outside_the_volume(dtype):
    """The volume as a strided window of a larger buffer that is NaN everywhere else (2 planes before/after, one channel
    image before/after, 2 rows above/below, 8 texels left/right of every row): a load that leaves the texture along any
    axis brings a NaN into the box -- the range check flags it, the output shows it.  Every kernel, tilted cameras whose
    rays leave the planes (zeros padding on all four borders), bounds done by buffer range checks and predicates only."""
    from ml_gmpi_amd import MPI
    dev = torch.device("cuda:0")
    M, D, Ht, Wt, S = 2, 9, 64, 72, 96
    g = torch.Generator().manual_seed(77)
    vol_c = torch.rand((M, D, 4, Ht, Wt), generator=g).to(dtype)
    big = torch.full((M, D + 4, 6, Ht + 4, Wt + 16), float("nan"), dtype=dtype, device=dev)
    window = big[:, 2:-2, 1:5, 2:-2, 8:-8]
    window.copy_(vol_c.to(dev))
    assert not window.is_contiguous() and window.data_ptr() % 16 == 0
    ray, eye, zd = _cam(M, S, S, seed=9, tilt=0.9)
    dhw = _dhw(M, D, ext=0.18, last=0.6)   # small planes: rays of the image border leave planes 0 .. D-2 on every side
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for ac in (True, False):
        ref = None
        for variant in variants():
            if variant == "band" and dtype == torch.float16:
                continue  # (refused for fp16 volumes: GMPI_E_VARIANT)
            for strict in (True, False):
                mpi = MPI(align_corners=ac, variant=variant, strict_order=strict, range_check="touched", on_out_of_plane="raise")
                with torch.no_grad():
                    out = mpi.render_views(window, t(dhw), t(ray), t(eye), t(zd), check_last_plane=False, want_transmittance=True)
                    alone = mpi.render_views(vol_c.to(dev), t(dhw), t(ray), t(eye), t(zd), check_last_plane=False, want_transmittance=True)
                torch.cuda.synchronize()
                for k in ("color", "depth", "T"):
                    assert torch.isfinite(out[k]).all(), (ac, variant, strict, k)
                    assert torch.equal(out[k], alone[k]), (ac, variant, strict, k)
                if strict:
                    if ref is None:
                        ref = out
                    else:
                        assert all(torch.equal(out[k], ref[k]) for k in ("color", "depth", "T")), (ac, variant)
        # the zeros padding is really exercised: some pixels see (almost) nothing of the first planes
        assert float(ref["T"].max()) > 1e-3
