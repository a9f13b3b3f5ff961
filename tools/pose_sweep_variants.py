"""Per-draw timing of a workload under the renderer's pose distribution, per kernel variant (GPU):
    python tools/pose_sweep_variants.py cfg2 [draws]   ->  one line per draw: seed, max |yaw|, max |pitch|, ms per variant"""
import sys, os, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ml_gmpi_amd
from ml_gmpi_amd import _lib
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 32
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["auto", "lds", "wave"]
preset, S, D, n_views, dtype, want_T, desc = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
r = ml_gmpi_amd.make_renderer(preset, n_planes=D, device=dev, on_out_of_plane="raise")
r.set_cam(r.cam_fov, S, S)
g = torch.Generator(device=dev).manual_seed(3000)
rgba = torch.empty((n_views, D, 4, S, S), device=dev, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
for i in range(n_views):
    rgba[i] = torch.rand((D, 4, S, S), device=dev, generator=g).to(rgba.dtype)
rgba[:, -1, 3] = 1.0
dhw = r._dhw_on_device().expand(n_views, -1, -1).contiguous()
status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
out = dict(color=torch.empty((n_views, 3, S, S), device=dev), depth=torch.empty((n_views, 1, S, S), device=dev))
for _ in range(200):  # clock ramp
    r.mpi.render_views(rgba, dhw, *[torch.cat(x) for x in r.sample_cam_poses(n_views, 0, 0, 0, 0, False, given_yaws=torch.zeros(n_views, 1), given_pitches=torch.zeros(n_views, 1))[3:6]], status=status, defer_status=True, out=out)
torch.cuda.synchronize()
tot = {v: [] for v in variants}
for d in range(draws):
    torch.manual_seed(100 + d)
    cam = r.sample_cam_poses(n_views, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
    row = []
    for v in variants:
        r.mpi.variant = v
        def step():
            r.mpi.render_views(rgba, dhw, ray, eye, zd, check_last_plane=True, out_pm1=True, status=status, defer_status=True, out=out, frontal_hint=False)
        step()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for e0, e1 in evs:
            e0.record(); step(); e1.record()
        torch.cuda.synchronize()
        ms = statistics.median(e0.elapsed_time(e1) for e0, e1 in evs)
        tot[v].append(ms); row.append(f"{v} {ms:.4f}")
    print(f"seed {100 + d}: max|yaw| {float(cam[0].abs().max()):.3f} max|pitch| {float(cam[1].abs().max()):.3f}  " + "  ".join(row), flush=True)
r.mpi.raise_on_status(status)
print("mean: " + "  ".join(f"{v} {sum(t) / len(t):.4f}" for v, t in tot.items()) + "   worst: " + "  ".join(f"{v} {max(t):.4f}" for v, t in tot.items()))
