"""Kernel time vs camera tilt for one variant (run on the GPU box): 4 views of 1024^2 x 96 at fixed (yaw, pitch).
usage: python tools/r2_pose.py bf16|f32 variant"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ml_gmpi_amd
from ml_gmpi_amd import _lib
dev = torch.device("cuda")
S, D, B = 1024, 96, 4
dt = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
variant = sys.argv[2]
poses = ((0, 0), (0.15, 0.05), (0.3, 0.1), (0.45, 0.2), (0.578, 0.254)) if len(sys.argv) < 4 else ((0, 0), (0.15, 0.05))
r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, kernel_variant=variant, on_out_of_plane="raise")
rgba = torch.rand((B, D, 4, S, S), device=dev).to(dt); rgba[:, -1, 3] = 1
r.set_cam(r.cam_fov, S, S)
out = []
for yaw, pitch in poses:
    gy = torch.tensor([[yaw], [-yaw], [yaw], [-yaw]], dtype=torch.float32); gp = torch.tensor([[pitch], [pitch], [-pitch], [-pitch]], dtype=torch.float32)
    cam = r.sample_cam_poses(B, 0, 0, 0, 0, False, given_yaws=gy, given_pitches=gp)
    ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
    dhw = r._dhw_on_device().expand(B, -1, -1).contiguous()
    status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
    f = lambda: r.mpi.render_views(rgba, dhw, ray, eye, zd, check_last_plane=True, out_pm1=True, defer_status=True, status=status)
    with torch.no_grad():
        for _ in range(2): f()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(8): f()
        e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 8
    st = status.cpu().tolist()
    out.append(f"({yaw},{pitch}) {ms:.3f} ms [unfit/10 launches {st[3]}]")
print(sys.argv[1], variant, "tune", os.environ.get("GMPI_TUNE_WAVE", "-"), " | ".join(out))
