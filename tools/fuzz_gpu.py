"""Long randomized parity run on the GPU box (not part of the test suite): random image / texture sizes, plane counts, batch sizes,
presets, align_corners, storage types and poses (random, 2-sigma, beyond); every kernel variant; strict mode must equal the CPU
oracle bit for bit, default mode must stay within 1e-5.  usage: python tools/fuzz_gpu.py [n_cases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "/oracle", "/tests"):
    sys.path.insert(0, ROOT + d)
import numpy as np
import torch
import oracle
from test_hip_parity import hip_render

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
from ml_gmpi_amd.renderer import MPIRenderer, PRESETS
t0 = time.time()
worst = dict(color=0.0, depth=0.0, T=0.0)
n_runs = 0
for i in range(n_cases):
    big = rng.random() < 0.15
    H, W = (int(rng.integers(200, 700)), int(rng.integers(200, 700))) if big else (int(rng.integers(4, 200)), int(rng.integers(4, 200)))
    Ht, Wt = 8 * int(rng.integers(1, 80 if big else 30)), 8 * int(rng.integers(1, 80 if big else 30))
    D = int(rng.integers(1, 40 if big else 130))
    if os.environ.get("FUZZ_LARGE"):  # launches large enough for AUTO's band path (and its view sharing): few planes keep the oracle quick
        big = True
        H, W = int(rng.integers(256, 1100)), int(rng.integers(256, 1100))
        Ht, Wt = 8 * int(rng.integers(16, 140)), 8 * int(rng.integers(16, 140))
        D = int(rng.integers(1, 12))
    B = int(rng.integers(1, 4)) if not (big and rng.random() < 0.3) else int(rng.integers(4, 9))   # (many views: AUTO's band + gated tile launch)
    preset = ["FFHQ", "AFHQCat", "MetFaces"][int(rng.integers(0, 3))]
    ac = bool(rng.integers(0, 2))
    kw = dict(PRESETS[preset])
    kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse",
              cam_sample_method="truncated_gaussian", mpi_align_corners=ac, use_confined_volume=bool(rng.integers(0, 2)), device=torch.device("cpu"))
    r = MPIRenderer(**kw)
    S = max(H, W)
    r.set_cam(r.cam_fov, S, S)
    mode = int(rng.integers(0, 3))
    torch.manual_seed(int(rng.integers(0, 1 << 30)))
    if mode == 0:
        cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    else:
        f = 2.0 if mode == 1 else float(rng.uniform(0.5, 2.6))
        gy = torch.tensor([[(-1) ** b * f * r.horizontal_std * rng.uniform(0.3, 1)] for b in range(B)], dtype=torch.float32)
        gp = torch.tensor([[(-1) ** (b // 2) * f * r.vertical_std * rng.uniform(0.3, 1)] for b in range(B)], dtype=torch.float32)
        cam = r.sample_cam_poses(B, 0, 0, 0, 0, False, given_yaws=gy, given_pitches=gp)
    rgba = torch.rand((B, D, 4, Ht, Wt))
    if rng.random() < 0.3:
        rgba[:, :, 3] = (rgba[:, :, 3] > 0.6).float()
    dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    vol = rgba.to(dtype)
    dhw = r.static_mpi_plane_dhws.reshape(1, -1, 3).expand(B, -1, -1).contiguous()
    ray = torch.cat(cam[3])[:, :, :H, :W].contiguous()
    eye, zd = torch.cat(cam[4]), torch.cat(cam[5])
    orc = oracle.render(vol.float(), dhw, ray, eye, zd, align_corners=ac, threads=True)
    variants = ("gather", "lds", "wave", "band", "auto")
    n_runs += len(variants)
    for variant in variants:
        out = hip_render(vol, dhw, ray, eye, zd, ac=ac, variant=variant, strict=True, check_last=False)
        for k in ("color", "depth", "T"):
            if not np.array_equal(out[k], orc[k]):
                print("STRICT MISMATCH", i, dict(H=H, W=W, Ht=Ht, Wt=Wt, D=D, B=B, preset=preset, ac=ac, dtype=str(dtype), mode=mode), variant, k,
                      float(np.nanmax(np.abs(out[k] - orc[k]))))
                sys.exit(1)
        fast = hip_render(vol, dhw, ray, eye, zd, ac=ac, variant=variant, check_last=False)
        for k, tol in (("color", 5e-6), ("depth", 1e-5), ("T", 1e-5)):
            err = float(np.abs(fast[k] - orc[k]).max())
            worst[k] = max(worst[k], err)
            if not err <= tol:
                print("DEFAULT-MODE MISMATCH", i, dict(H=H, W=W, Ht=Ht, Wt=Wt, D=D, B=B, preset=preset, ac=ac, dtype=str(dtype), mode=mode), variant, k, err)
                sys.exit(1)
print(f"fuzz ok: {n_cases} cases, {n_runs} (case, variant) pairs x 2 modes (gather / lds / wave / band / auto) in {time.time() - t0:.0f} s; "
      f"worst default-mode error {worst}")
