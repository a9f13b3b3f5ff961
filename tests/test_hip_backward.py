"""Backward of the fused render (gradient w.r.t. the RGBA volume; reference: autograd through MPI.forward in the
G-step, gmpi/train.py:740-779).  The HIP backward (fp32, atomics) is compared with (1) fixtures made by the reference's own
autograd (tests/golden/backward_*.npz, oracle/make_golden.py) and (2) torch autograd on a float64 restatement of the same
forward (tests/_torch_ref.py) for the shapes the fixtures do not cover."""
import numpy as np
import pytest
import torch

import oracle
from _torch_ref import torch_render
from test_hip_edge_cases import _cam, _dhw

pytestmark = pytest.mark.gpu


def _setup(N, M, D, Ht, Wt, H, W, seed, v2m=None):
    rgba = oracle.synth_rgba(seed, (M, D, 4, Ht, Wt))
    ray, eye, zd = _cam(N, H, W, seed=seed + 1, tilt=0.3)
    dhw = _dhw(M, D)
    v2m = np.arange(N) % M if v2m is None else np.asarray(v2m)
    return rgba, dhw, ray, eye, zd, v2m


def _ref_grads(rgba, dhw, ray, eye, zd, v2m, gc, gd, ac):
    t = lambda a: torch.from_numpy(np.asarray(a)).double()
    vol = t(rgba).requires_grad_(True)
    color, depth = torch_render(vol, t(dhw), t(ray), t(eye), t(zd), v2m, align_corners=ac)
    loss = (color * t(gc)).sum() + (depth * t(gd)).sum()
    loss.backward()
    return color.detach().numpy(), depth.detach().numpy(), vol.grad.numpy()


@pytest.mark.parametrize("cfg", [
    dict(N=2, M=2, D=6, Ht=24, Wt=28, H=20, W=22, ac=True),
    dict(N=3, M=1, D=5, Ht=16, Wt=16, H=33, W=17, ac=False),        # several views of one MPI accumulate into one gradient
    dict(N=2, M=2, D=8, Ht=64, Wt=64, H=64, W=64, ac=True, variant="lds"),
    dict(N=2, M=2, D=8, Ht=64, Wt=64, H=64, W=64, ac=True, variant="gather"),      # one pixel per lane, global atomics only
    dict(N=2, M=1, D=98, Ht=48, Wt=48, H=40, W=72, ac=True),                       # two table chunks, ragged tiles
    dict(N=1, M=1, D=3, Ht=256, Wt=256, H=24, W=24, ac=False, fwd_tol=1e-4, bwd_tol=1e-4, rel_tol=2e-2),       # minified: boxes do not fit -> direct scatter
                                                                                   # (10 texels/pixel of white noise: fp32 vs float64 coordinates)
])
def test_backward_matches_autograd(cfg):
    from ml_gmpi_amd import MPI
    ac = cfg["ac"]
    rgba, dhw, ray, eye, zd, v2m = _setup(cfg["N"], cfg["M"], cfg["D"], cfg["Ht"], cfg["Wt"], cfg["H"], cfg["W"], seed=41)
    g = np.random.default_rng(5)
    gc = g.standard_normal((cfg["N"], 3, cfg["H"], cfg["W"])).astype(np.float32)
    gd = g.standard_normal((cfg["N"], 1, cfg["H"], cfg["W"])).astype(np.float32)
    ref_c, ref_d, ref_g = _ref_grads(rgba, dhw, ray, eye, zd, v2m, gc, gd, ac)

    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    vol = t(rgba).requires_grad_(True)
    mpi = MPI(align_corners=ac, variant=cfg.get("variant", "auto"), on_out_of_plane="raise")
    out = mpi.render_views(vol, t(dhw), t(ray), t(eye), t(zd), view_to_mpi=t(v2m.astype(np.int32)), check_last_plane=False)
    assert out["color"].requires_grad and out["depth"].requires_grad
    assert np.abs(out["color"].detach().cpu().numpy() - ref_c).max() <= cfg.get("fwd_tol", 1e-5)
    loss = (out["color"] * t(gc)).sum() + (out["depth"] * t(gd)).sum()
    loss.backward()
    got = vol.grad.cpu().numpy()
    scale = np.abs(ref_g).max()
    assert np.abs(got - ref_g).max() <= cfg.get("bwd_tol", 2e-5) * scale + 1e-6, (np.abs(got - ref_g).max(), scale)
    # relative check on the significant entries
    big = np.abs(ref_g) > 1e-3 * scale
    assert np.max(np.abs(got[big] - ref_g[big]) / np.abs(ref_g[big])) <= cfg.get("rel_tol", 2e-3)


def test_backward_behind_opaque_and_nearly_opaque_planes():
    """alpha == 1 (om = 1e-10) and 1 - 1e-6 in the middle of the stack: dL/da of such a texel is the O(1) quantity
    (sum of the 1e-10-scaled weights behind it) / om; a front-to-back difference of sums loses it, torch's cumprod
    backward (reverse cumsum) does not.  Also four exactly opaque planes in a row (final T underflows in fp32)."""
    from ml_gmpi_amd import MPI
    N = M = 1
    D, S = 8, 32
    rgba, dhw, ray, eye, zd, v2m = _setup(N, M, D, S, S, S, S, seed=47)
    rgba[:, 2, 3, :, : S // 2] = 1.0
    rgba[:, 4, 3, :, S // 4:] = 1.0 - 1e-6
    rgba[:, 3:7, 3, : S // 3, :] = 1.0
    g = np.random.default_rng(6)
    gc = g.standard_normal((N, 3, S, S)).astype(np.float32)
    gd = g.standard_normal((N, 1, S, S)).astype(np.float32)
    _, _, ref_g = _ref_grads(rgba, dhw, ray, eye, zd, v2m, gc, gd, True)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for variant in ("auto", "gather"):
        vol = t(rgba).requires_grad_(True)
        mpi = MPI(align_corners=True, variant=variant, on_out_of_plane="raise")
        out = mpi.render_views(vol, t(dhw), t(ray), t(eye), t(zd), view_to_mpi=t(v2m.astype(np.int32)), check_last_plane=False)
        ((out["color"] * t(gc)).sum() + (out["depth"] * t(gd)).sum()).backward()
        got = vol.grad.cpu().numpy()
        scale = np.abs(ref_g).max()
        assert np.isfinite(got).all()
        assert np.abs(got - ref_g).max() <= 1e-4 * scale, (variant, np.abs(got - ref_g).max(), scale)


def test_backward_through_renderer_render_pm1_and_expand():
    """MPIRenderer.render under grad: colour in [-1,1] (factor 2), expanded volume (n_view_per_z pattern, train.py:553-558)."""
    from ml_gmpi_amd import make_renderer
    dev = torch.device("cuda:0")
    D, S, K = 4, 32, 2
    r = make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
    base = torch.from_numpy(oracle.synth_rgba(43, (1, D, 4, S, S))).to(dev).requires_grad_(True)
    vol = base.unsqueeze(0).expand(K, -1, -1, -1, -1, -1).reshape(K, D, 4, S, S)   # materialises, grads sum back into `base`
    torch.manual_seed(3)
    rgb, depth, c2w, ang = r.render(vol, S, S)
    gc = torch.randn_like(rgb)
    (rgb * gc).sum().backward()
    assert base.grad is not None and torch.isfinite(base.grad).all() and float(base.grad.abs().max()) > 0
    # cross-check with autograd on the float64 restatement, same cameras
    torch.manual_seed(3)
    cam = r.sample_cam_poses(K, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    vol64 = base.detach().double().cpu().requires_grad_(True)
    dhw = r.static_mpi_plane_dhws.double().reshape(1, -1, 3)
    color, _ = torch_render(vol64, dhw, torch.cat(cam[3]).double().cpu(), torch.cat(cam[4]).double().cpu(),
                            torch.cat(cam[5]).double().cpu(), [0] * K)
    ((2 * color - 1) * gc.double().cpu()).sum().backward()
    ref = vol64.grad.numpy()
    got = base.grad.cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


def test_two_renders_then_one_backward_keep_their_own_cameras():
    """`render()` keeps its ray / eye / z_dir tensors in buffers reused per (B, H, W, stream) -- but not while a graph
    is being recorded: two calls under grad with DIFFERENT poses followed by one backward must differentiate each graph with its own
    cameras (a multi-view or consistency loss; the reference's train.py happens to call backward after every render)."""
    from ml_gmpi_amd import make_renderer
    dev = torch.device("cuda:0")
    D, S, B = 4, 32, 2
    r = make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
    base = torch.from_numpy(oracle.synth_rgba(59, (B, D, 4, S, S))).to(dev)
    poses = [dict(given_yaws=torch.full((B, 1), y), given_pitches=torch.full((B, 1), p_)) for y, p_ in ((0.25, 0.05), (-0.3, -0.1))]
    g = torch.randn((B, 3, S, S), generator=torch.Generator().manual_seed(5)).to(dev)
    alone = []
    for kw in poses:                                   # each graph on its own
        v = base.clone().requires_grad_(True)
        (r.render(v, S, S, **kw)[0] * g).sum().backward()
        alone.append(v.grad.clone())
    assert float((alone[0] - alone[1]).abs().max()) > 1e-3 * float(alone[0].abs().max())   # (the poses do differ)
    va, vb = base.clone().requires_grad_(True), base.clone().requires_grad_(True)
    rgb_a = r.render(va, S, S, **poses[0])[0]
    rgb_b = r.render(vb, S, S, **poses[1])[0]          # same shape, same stream: would overwrite reused camera buffers
    ((rgb_a * g).sum() + (rgb_b * g).sum()).backward()
    for got, want in ((va.grad, alone[0]), (vb.grad, alone[1])):   # (atomic adds: the summation order varies from run to run)
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    with torch.no_grad():                              # (and without a graph the buffers ARE reused: nothing allocated per call)
        x = r.render(base, S, S, **poses[0])
        y = r.render(base, S, S, **poses[1])
    assert x[0].shape == y[0].shape


def test_no_gradient_to_geometry_and_no_grad_mode():
    from ml_gmpi_amd import MPI
    rgba, dhw, ray, eye, zd, v2m = _setup(1, 1, 3, 16, 16, 16, 16, seed=47)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mpi = MPI(on_out_of_plane="raise")
    with pytest.raises(NotImplementedError):
        mpi.render_views(t(rgba), t(dhw).requires_grad_(True), t(ray), t(eye), t(zd))
    with torch.no_grad():
        out = mpi.render_views(t(rgba).requires_grad_(True), t(dhw), t(ray), t(eye), t(zd))
    assert not out["color"].requires_grad


def _backward_fixtures():
    import glob
    import os
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "backward_*.npz")))


@pytest.mark.parametrize("name", _backward_fixtures())
def test_backward_matches_reference_autograd_fixtures(name):
    """d loss / d rgba of the HIP backward against fixtures made by the REFERENCE'S OWN autograd (oracle/make_golden.py
    `run_backward_cases`: MPIRenderer.render / MPI.forward under grad on the CPU, fp32 -- what the G-step
    back-propagates, train.py:740-779), incl. exactly / nearly opaque planes in the middle of the stack, a texture
    finer than the image with align_corners=False, and a ragged views-per-MPI list whose views accumulate into one MPI."""
    from _util import load_npz
    from ml_gmpi_amd import MPI
    fx = load_npz(name + ".npz")
    meta = fx["meta"]
    pm1 = "ref_rgb_pm1" in fx
    ref_c = fx["ref_rgb_pm1"] if pm1 else fx["ref_color01"]
    ref_g = fx["ref_grad_rgba"]
    N = fx["ray_dir"].shape[0]
    v2m = fx["view_to_mpi"] if "view_to_mpi" in fx else np.arange(N, dtype=np.int32)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    scale = np.abs(ref_g).max()
    for variant in ("auto", "gather"):
        vol = t(fx["rgba"]).requires_grad_(True)
        mpi = MPI(align_corners=meta["ac"], variant=variant, on_out_of_plane="raise")
        out = mpi.render_views(vol, t(fx["dhw"]), t(fx["ray_dir"]), t(fx["eye"]), t(fx["zdir"]), view_to_mpi=t(v2m.astype(np.int32)),
                               check_last_plane=False, out_pm1=pm1)
        assert np.abs(out["color"].detach().cpu().numpy() - ref_c).max() <= 1e-5
        assert np.abs(out["depth"].detach().cpu().numpy() - fx["ref_depth"]).max() <= 1e-5
        ((out["color"] * t(fx["g_rgb"])).sum() + (out["depth"] * t(fx["g_depth"])).sum()).backward()
        got = vol.grad.cpu().numpy()
        assert np.isfinite(got).all()
        # fp32 autograd on the CPU vs fp32 atomics on the GPU: both carry ~1e-6 relative noise per term
        assert np.abs(got - ref_g).max() <= 3e-5 * scale, (variant, np.abs(got - ref_g).max(), scale)
        big = np.abs(ref_g) > 1e-3 * scale
        assert np.max(np.abs(got[big] - ref_g[big]) / np.abs(ref_g[big])) <= 5e-3, variant


def test_inplace_update_between_forward_and_backward_is_detected():
    """The volume, the camera tensors and the transmittance the backward starts from are kept through save_for_backward:
    overwriting the volume before backward() must raise (torch's version counter), not differentiate overwritten memory;
    and a caller-supplied out["T"] buffer that is reused by the next launch must not disturb an earlier graph."""
    from ml_gmpi_amd import MPI
    rgba, dhw, ray, eye, zd, v2m = _setup(2, 2, 6, 32, 32, 32, 32, seed=51)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mpi = MPI(align_corners=True, on_out_of_plane="raise")
    base = t(rgba)
    leaf = base.clone().requires_grad_(True)
    vol = leaf * 1.0
    out = mpi.render_views(vol, t(dhw), t(ray), t(eye), t(zd), check_last_plane=False)
    vol.mul_(0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out["color"].sum().backward()
    # shared output buffers (the batch driver's pattern): two graphs, one T buffer
    shared = dict(T=torch.empty((2, 1, 32, 32), device=dev))
    a = base.clone().requires_grad_(True)
    b = (base * 0.5).clone().requires_grad_(True)
    out_a = mpi.render_views(a, t(dhw), t(ray), t(eye), t(zd), check_last_plane=False, want_transmittance=True, out=shared)
    out_b = mpi.render_views(b, t(dhw), t(ray), t(eye), t(zd), check_last_plane=False, want_transmittance=True, out=shared)
    out_a["color"].sum().backward()
    alone = base.clone().requires_grad_(True)
    mpi.render_views(alone, t(dhw), t(ray), t(eye), t(zd), check_last_plane=False)["color"].sum().backward()
    assert torch.allclose(a.grad, alone.grad, rtol=1e-4, atol=1e-6)
    assert torch.equal(shared["T"], out_b["T"])


def _rot_cam(N, H, W, yaw, pitch, roll, fov=0.11):
    """Pinhole rays of N cameras on the unit sphere around (0, 0, 1), looking at it, with an in-plane roll: the tilt and the rotation
    shear the texel boxes of neighbouring tiles against each other."""
    ys, xs = np.meshgrid(np.linspace(-fov, fov, H), np.linspace(-fov, fov, W), indexing="ij")
    rays, eyes, zds = [], [], []
    for n in range(N):
        sgn = 1 if n % 2 == 0 else -1
        a, b, c = sgn * yaw, (0.5 + 0.5 * n / max(N - 1, 1)) * pitch, sgn * roll
        Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        R = Ry @ Rx @ Rz
        d = np.stack([xs, ys, np.ones_like(xs)]).reshape(3, -1)
        d = d / np.linalg.norm(d, axis=0)
        rays.append((R @ d).reshape(3, H, W))
        eyes.append(np.array([0.0, 0.0, 1.0]) - R[:, 2])
        zds.append(R[:, 2])
    return (np.stack(rays).astype(np.float32), np.array(eyes, np.float32), np.array(zds, np.float32))


@pytest.mark.parametrize("cfg", [
    dict(H=160, W=160, Ht=128, Wt=128, yaw=0.0, pitch=0.0, roll=0.0),      # frontal: 5 x 10 tiles, interior tiles with a whole 5 x 5 block
    dict(H=160, W=192, Ht=192, Wt=160, yaw=0.35, pitch=0.12, roll=0.0),    # tilted: sheared boxes
    dict(H=144, W=160, Ht=128, Wt=128, yaw=0.25, pitch=-0.1, roll=0.5),    # in-plane rotation: neighbours' boxes overlap on two sides
    dict(H=144, W=160, Ht=128, Wt=128, yaw=0.1, pitch=0.1, roll=1.3),      # nearly a quarter turn: "left" tiles lie above
    dict(H=96, W=224, Ht=40, Wt=56, yaw=0.3, pitch=0.0, roll=0.2),         # texture coarser than the image: tiles several columns apart share texels
    dict(H=100, W=130, Ht=300, Wt=260, yaw=0.3, pitch=0.1, roll=-0.3),     # texture finer than the image: large boxes (some not staged)
])
def test_tile_backward_matches_the_all_atomic_kernel(cfg):
    """The round-5 tile backward (pipelined planes, fixed-point boxes, flush waves) on many tiles under tilted and ROTATED cameras -- sheared boxes,
    neighbours that overlap on two sides, textures coarser and finer than the image -- against the one-pixel-per-lane kernel (GMPI_VARIANT_GATHER:
    16 global atomics per pixel and plane, no staging) and against float64 autograd of the same forward.  (Written for round 5's exclusive-cell
    experiment -- plain stores for box cells no other tile touches; correct, slower, not kept: profiles/r05_backward.txt -- and kept as the
    multi-tile cross-check the backward did not have.)"""
    from ml_gmpi_amd import MPI
    N = M = 2
    D = 5
    H, W, Ht, Wt = cfg["H"], cfg["W"], cfg["Ht"], cfg["Wt"]
    rgba = oracle.synth_rgba(61, (M, D, 4, Ht, Wt))
    ray, eye, zd = _rot_cam(N, H, W, cfg["yaw"], cfg["pitch"], cfg["roll"])
    dhw = _dhw(M, D, ext=0.30, last=0.6)
    v2m = np.arange(N)
    g = np.random.default_rng(9)
    gc = g.standard_normal((N, 3, H, W)).astype(np.float32)
    gd = g.standard_normal((N, 1, H, W)).astype(np.float32)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    grads = {}
    for variant in ("auto", "gather"):
        vol = t(rgba).requires_grad_(True)
        mpi = MPI(align_corners=True, variant=variant, on_out_of_plane="raise")
        out = mpi.render_views(vol, t(dhw), t(ray), t(eye), t(zd), views_per_mpi=1, check_last_plane=False)
        ((out["color"] * t(gc)).sum() + (out["depth"] * t(gd)).sum()).backward()
        grads[variant] = vol.grad.cpu().numpy()
    scale = np.abs(grads["gather"]).max()
    assert scale > 0
    assert np.abs(grads["auto"] - grads["gather"]).max() <= 1e-5 * scale, (np.abs(grads["auto"] - grads["gather"]).max(), scale)
    # (float64 coordinates against fp32 ones on white noise, hundreds of pixels per row: a little looser than the small cases above)
    _, _, ref_g = _ref_grads(rgba, dhw, ray, eye, zd, v2m, gc, gd, True)
    assert np.abs(grads["auto"] - ref_g).max() <= 5e-5 * np.abs(ref_g).max() + 1e-6


@pytest.mark.parametrize("cfg", [
    dict(H=160, W=160, Ht=128, Wt=128, yaw=0.0, pitch=0.0, roll=0.0, vpm=1),
    dict(H=160, W=192, Ht=192, Wt=160, yaw=0.35, pitch=0.12, roll=0.0, vpm=1),     # tilted: the pixel boxes of the texel tiles shear
    dict(H=144, W=160, Ht=128, Wt=128, yaw=0.25, pitch=-0.1, roll=0.5, vpm=1),     # in-plane rotation: a texel's candidates are a rotated window
    dict(H=144, W=160, Ht=128, Wt=128, yaw=0.1, pitch=0.1, roll=1.3, vpm=2),       # nearly a quarter turn; two views of every MPI summed in registers
    dict(H=96, W=224, Ht=40, Wt=56, yaw=0.3, pitch=0.0, roll=0.2, vpm=1),          # texture coarser than the image: many pixels per texel (wide windows)
    dict(H=100, W=130, Ht=300, Wt=260, yaw=0.3, pitch=0.1, roll=-0.3, vpm=1),      # texture finer than the image: most texels get nothing
    dict(H=40, W=1800, Ht=32, Wt=64, yaw=0.2, pitch=0.0, roll=0.0, vpm=1),         # 28 pixels per texel: a pixel box wider than a chunk (column blocks), 58 candidates per row
])
def test_gather_backward_matches_autograd_and_is_bit_reproducible(cfg):
    """MPI(backward="gather") (round 6, render_backward_gather.hip: pixel pass + texel gather, no atomics, no zero-fill): against float64 autograd of
    the same forward, against the atomic tile kernel, and twice in a row -- every gradient cell is written once, in a fixed order, so the two runs
    are BIT-identical (the atomic kernels' are not: bench.py's `backward_repeatability`)."""
    from ml_gmpi_amd import MPI
    M, D, vpm = 2, 5, cfg["vpm"]
    N = M * vpm
    H, W, Ht, Wt = cfg["H"], cfg["W"], cfg["Ht"], cfg["Wt"]
    rgba = oracle.synth_rgba(63, (M, D, 4, Ht, Wt))
    ray, eye, zd = _rot_cam(N, H, W, cfg["yaw"], cfg["pitch"], cfg["roll"])
    dhw = _dhw(M, D, ext=0.30, last=0.6)
    g = np.random.default_rng(19)
    gc = g.standard_normal((N, 3, H, W)).astype(np.float32)
    gd = g.standard_normal((N, 1, H, W)).astype(np.float32)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    grads = {}
    for mode in ("gather", "gather again", "atomic"):
        vol = t(rgba).requires_grad_(True)
        mpi = MPI(align_corners=True, on_out_of_plane="raise", backward=mode.split()[0])
        out = mpi.render_views(vol, t(dhw), t(ray), t(eye), t(zd), views_per_mpi=vpm, check_last_plane=False)
        ((out["color"] * t(gc)).sum() + (out["depth"] * t(gd)).sum()).backward()
        grads[mode] = vol.grad.clone()
    assert torch.equal(grads["gather"], grads["gather again"])
    a, b = grads["gather"].cpu().numpy(), grads["atomic"].cpu().numpy()
    scale = np.abs(b).max()
    assert scale > 0 and np.isfinite(a).all()
    assert np.abs(a - b).max() <= 1e-5 * scale, (np.abs(a - b).max(), scale)
    _, _, ref_g = _ref_grads(rgba, dhw, ray, eye, zd, np.repeat(np.arange(M), vpm), gc, gd, True)
    assert np.abs(a - ref_g).max() <= 5e-5 * np.abs(ref_g).max() + 1e-6


def test_gather_backward_falls_back_and_accumulates():
    """align_corners=False and a ragged view list take the atomic path whatever `backward` says (same gradients); through the C ABI, a launch
    WITHOUT GMPI_FLAG_GRAD_OVERWRITE adds into what the buffer holds (read, add, write back -- by the cell's one owner)."""
    import ctypes
    from ml_gmpi_amd import MPI, _lib
    M, D, S = 2, 4, 96
    rgba = oracle.synth_rgba(64, (M, D, 4, S, S))
    ray, eye, zd = _rot_cam(3, S, S, 0.2, 0.1, 0.1)
    dhw = _dhw(M, D, ext=0.30, last=0.6)
    g = np.random.default_rng(20)
    gc = g.standard_normal((3, 3, S, S)).astype(np.float32)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    v2m = np.array([0, 1, 1], dtype=np.int32)
    res = {}
    for mode in ("gather", "atomic"):
        for ac in (True, False):
            vol = t(rgba).requires_grad_(True)
            out = MPI(align_corners=ac, on_out_of_plane="raise", backward=mode).render_views(vol, t(dhw), t(ray), t(eye), t(zd), view_to_mpi=t(v2m), check_last_plane=False)
            (out["color"] * t(gc)).sum().backward()
            res[(mode, ac)] = vol.grad.cpu().numpy()
    for ac in (True, False):
        sc = np.abs(res[("atomic", ac)]).max()
        assert np.abs(res[("gather", ac)] - res[("atomic", ac)]).max() <= 1e-5 * sc
    # accumulate semantics of the gather path (no OVERWRITE flag): grad = 1 + gradient
    lib = _lib.load_library()
    vol = t(rgba[:2])
    ray2, eye2, zd2 = t(ray[:2]), t(eye[:2]), t(zd[:2])
    mpi = MPI(align_corners=True, on_out_of_plane="raise")
    fwd = mpi.render_views(vol, t(dhw), ray2, eye2, zd2, views_per_mpi=1, check_last_plane=False, want_transmittance=True, defer_status=True, _in_autograd_fn=True)
    p, keep = fwd["_bwd"]
    p.rgb_out = p.depth_out = p.status = None
    need = int(lib.gmpi_render_backward_workspace_bytes(ctypes.byref(p)))
    assert need >= 2 * D * S * S * 24
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), need
    gcol = t(gc[:2])
    outs = []
    for flag, fill in ((_lib.FLAG_GRAD_OVERWRITE, float("nan")), (0, 1.0)):
        p.flags = (p.flags & ~_lib.FLAG_GRAD_OVERWRITE) | flag
        grad = torch.full_like(vol, fill)
        gs = (ctypes.c_int64 * 5)(*grad.stride())
        _lib.check(lib.gmpi_mpi_render_backward_launch(ctypes.byref(p), gcol.data_ptr(), None, grad.data_ptr(), gs, torch.cuda.current_stream(dev).cuda_stream), "bwd")
        torch.cuda.synchronize()
        outs.append(grad)
    assert torch.isfinite(outs[0]).all()                      # OVERWRITE: every element written, whatever the buffer held
    assert torch.equal(outs[1], outs[0] + 1.0)                # without it: added to the ones, cell by cell


def test_tile_backward_several_views_per_mpi_keep_their_atomics():
    """Two views of ONE MPI add into the same gradient volume."""
    from ml_gmpi_amd import MPI
    N, M, D, S = 4, 2, 4, 128
    rgba = oracle.synth_rgba(62, (M, D, 4, S, S))
    ray, eye, zd = _rot_cam(N, S, S, 0.2, 0.1, 0.1)
    dhw = _dhw(M, D, ext=0.30, last=0.6)
    g = np.random.default_rng(10)
    gc = g.standard_normal((N, 3, S, S)).astype(np.float32)
    gd = g.standard_normal((N, 1, S, S)).astype(np.float32)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    vol = t(rgba).requires_grad_(True)
    out = MPI(align_corners=True, on_out_of_plane="raise").render_views(vol, t(dhw), t(ray), t(eye), t(zd), views_per_mpi=2, check_last_plane=False)
    ((out["color"] * t(gc)).sum() + (out["depth"] * t(gd)).sum()).backward()
    _, _, ref_g = _ref_grads(rgba, dhw, ray, eye, zd, np.repeat(np.arange(M), 2), gc, gd, True)
    assert np.abs(vol.grad.cpu().numpy() - ref_g).max() <= 2e-5 * np.abs(ref_g).max() + 1e-6
