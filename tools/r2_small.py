"""Kernel time of small launches (single views): plane split on/off.  usage: python tools/r2_small.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ml_gmpi_amd
dev = torch.device("cuda")
for (S, D, B, dt) in ((256, 96, 1, torch.float32), (256, 32, 1, torch.float32), (512, 96, 1, torch.float32), (256, 96, 2, torch.bfloat16), (256, 96, 4, torch.float32), (512, 96, 2, torch.float32)):
    row = []
    for variant in ("lds", "wave"):
        r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, kernel_variant=variant, on_out_of_plane="raise")
        rgba = torch.rand((B, D, 4, S, S), device=dev).to(dt); rgba[:, -1, 3] = 1
        r.set_cam(r.cam_fov, S, S)
        torch.manual_seed(3)
        cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
        ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
        dhw = r._dhw_on_device().expand(B, -1, -1).contiguous()
        f = lambda: r.mpi.render_views(rgba, dhw, ray, eye, zd, check_last_plane=True, out_pm1=True, defer_status=True)
        with torch.no_grad():
            for _ in range(5): f()
            torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50): f()
            e.record(); torch.cuda.synchronize()
        row.append(f"{variant} {s.elapsed_time(e) / 50 * 1e3:.1f} us")
    print(f"{B} x {S}^2 x {D} {str(dt)[6:]} tune {os.environ.get('GMPI_TUNE_WAVE','-')}:", " | ".join(row))
