"""Randomized cross-check of the backward on the GPU box (not part of the test suite): the tile kernel (GMPI_VARIANT_AUTO: round-5 pipelined
kernel) against the one-pixel-per-lane kernel (GMPI_VARIANT_GATHER: 16 global atomics per pixel and plane, no staging) on random image / texture
sizes, plane counts (incl. more than one table chunk of 96), storage types, align_corners, views per MPI (uniform, ragged through view_to_mpi),
tilted and rotated pinhole cameras, exactly / nearly opaque planes, with and without a depth gradient and a forward transmittance.
usage: python tools/fuzz_backward_gpu.py [n_cases] [seed]    (FUZZ_BWD=gather: the atomics-free pair of round 6 in place of the tile kernel)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "/oracle", "/tests"):
    sys.path.insert(0, ROOT + d)
import numpy as np
import torch
from ml_gmpi_amd import MPI, _lib
if os.environ.get("FUZZ_LIB"):   # a profiling build (with GMPI_TUNE_SKIP=64: the round-1 tile kernel)
    _lib._SO = os.path.abspath(os.environ["FUZZ_LIB"])

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")


def cams(N, H, W, yaw, pitch, roll, fov):
    ys, xs = np.meshgrid(np.linspace(-fov, fov, H), np.linspace(-fov, fov, W), indexing="ij")
    rays, eyes, zds = [], [], []
    for n in range(N):
        a, b, c = yaw * rng.uniform(-1, 1), pitch * rng.uniform(-1, 1), roll * rng.uniform(-1, 1)
        Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        R = Ry @ Rx @ Rz
        d = np.stack([xs, ys, np.ones_like(xs)]).reshape(3, -1)
        d = d / np.linalg.norm(d, axis=0)
        rays.append((R @ d).reshape(3, H, W)); eyes.append(np.array([0.0, 0.0, 1.0]) - R[:, 2]); zds.append(R[:, 2])
    return np.stack(rays).astype(np.float32), np.array(eyes, np.float32), np.array(zds, np.float32)


t0 = time.time()
worst = 0.0
for i in range(n_cases):
    big = rng.random() < 0.2
    H, W = (int(rng.integers(100, 400)), int(rng.integers(100, 400))) if big else (int(rng.integers(3, 100)), int(rng.integers(3, 100)))
    Ht, Wt = int(rng.integers(2, 300 if big else 90)), 2 * int(rng.integers(1, 150 if big else 45))
    D = int(rng.choice([1, 2, 5, 9, 32, 97, 120])) if not big else int(rng.choice([2, 5, 12]))
    M = int(rng.integers(1, 4))
    mode = rng.integers(0, 3)
    if rng.random() < 0.5:   # round 6: the flush STORES whole aligned 128-byte lines that are one tile's alone -- needs Wt % 32 == 0 and one view per MPI
        Wt = 32 * int(rng.integers(1, 12 if big else 4))
        if rng.random() < 0.7:
            mode = 0
    if mode == 0:
        vpm, v2m, N = 1, None, M
    elif mode == 1:
        vpm = int(rng.integers(2, 4)); v2m, N = None, M * vpm
    else:
        N = int(rng.integers(1, 6)); v2m = rng.integers(0, M, N).astype(np.int32); vpm = 1
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    ac = bool(rng.integers(0, 2))
    rgba = rng.random((M, D, 4, Ht, Wt), dtype=np.float32)
    if rng.random() < 0.3:
        rgba[:, rng.integers(0, D), 3, : Ht // 2] = 1.0
    if rng.random() < 0.3:
        rgba[:, rng.integers(0, D), 3, :, Wt // 3:] = 1.0 - 1e-6
    ray, eye, zd = cams(N, H, W, rng.choice([0.0, 0.2, 0.5]), rng.choice([0.0, 0.1, 0.3]), rng.choice([0.0, 0.3, 1.2]), rng.choice([0.05, 0.11, 0.2]))
    d = 1.0 / np.linspace(1 / 0.95, 1 / 1.12, D) if D > 1 else np.array([1.12])
    ext = rng.choice([0.2, 0.3, 0.6])
    dhw = np.broadcast_to(np.stack([d, np.full(D, ext), np.full(D, ext)], 1)[None], (M, D, 3)).astype(np.float32).copy()
    gc = rng.standard_normal((N, 3, H, W)).astype(np.float32)
    gd = rng.standard_normal((N, 1, H, W)).astype(np.float32) if rng.random() < 0.7 else None
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    grads = {}
    for variant in ("auto", "gather"):
        vol = t(rgba).to(dtype).requires_grad_(True)
        mpi = MPI(align_corners=ac, variant=variant, on_out_of_plane="raise", range_check="off", backward=os.environ.get("FUZZ_BWD", "atomic") if variant == "auto" else "atomic")
        kw = dict(view_to_mpi=t(v2m)) if v2m is not None else dict(views_per_mpi=vpm)
        out = mpi.render_views(vol, t(dhw), t(ray), t(eye), t(zd), check_last_plane=False, **kw)
        loss = (out["color"] * t(gc)).sum()
        if gd is not None:
            loss = loss + (out["depth"] * t(gd)).sum()
        loss.backward()
        grads[variant] = vol.grad.float().cpu().numpy()
    scale = np.abs(grads["gather"]).max()
    err = np.abs(grads["auto"] - grads["gather"]).max()
    tol = (1e-5 if dtype is torch.float32 else 1e-2) * scale + 1e-7   # (16-bit gradients: both kernels accumulate in fp32, the result is rounded to the storage type)
    if not (np.isfinite(grads["auto"]).all() and err <= tol):
        print(f"MISMATCH case {i}: H {H} W {W} Ht {Ht} Wt {Wt} D {D} M {M} N {N} mode {mode} {dtype} ac {ac}: err {err:.3e} scale {scale:.3e}")
        if dtype is torch.float32 and H * W * D * N < 3e6:   # which of the two is off?  float64 autograd of the same forward (tests/_torch_ref.py)
            from _torch_ref import torch_render
            t64 = lambda a: torch.from_numpy(np.asarray(a)).double()
            vol = t64(rgba).requires_grad_(True)
            idx = v2m if v2m is not None else np.repeat(np.arange(M), vpm)
            color, depth = torch_render(vol, t64(dhw), t64(ray), t64(eye), t64(zd), idx, align_corners=ac)
            loss = (color * t64(gc)).sum() + ((depth * t64(gd)).sum() if gd is not None else 0.0)
            loss.backward()
            ref = vol.grad.numpy()
            print(f"   vs float64 autograd: tile kernel {np.abs(grads['auto'] - ref).max():.3e}, one-pixel-per-lane kernel {np.abs(grads['gather'] - ref).max():.3e} (scale {np.abs(ref).max():.3e})")
            if os.environ.get("FUZZ_DEBUG"):
                ea, eg = np.abs(grads['auto'] - ref), np.abs(grads['gather'] - ref)
                for k in range(D):
                    print(f"     plane {k}: tile err per channel {[float(f'{ea[0, k, c].max():.2e}') for c in range(4)]}  gather {[float(f'{eg[0, k, c].max():.2e}') for c in range(4)]}  |ref| max {[float(f'{np.abs(ref[0, k, c]).max():.2e}') for c in range(4)]} alpha range [{rgba[0, k, 3].min():.3f}, {rgba[0, k, 3].max():.6f}]")
                sys.exit(0)
        n_bad = globals().get("n_bad", 0) + 1
        globals()["n_bad"] = n_bad
        if n_bad >= 5:
            sys.exit(1)
    worst = max(worst, err / max(scale, 1e-30) if dtype is torch.float32 else 0.0)
print(f"backward fuzz ok: {n_cases} cases (tile kernel vs one-pixel-per-lane kernel) in {time.time() - t0:.0f} s; worst fp32 difference {worst:.2e} of the largest gradient")
