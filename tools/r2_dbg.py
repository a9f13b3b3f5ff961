import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ml_gmpi_amd
from ml_gmpi_amd import _lib
dev = torch.device("cuda")
for (S, D, B, dt, seed) in ((256, 96, 8, torch.float32, 3), (1024, 96, 4, torch.bfloat16, 3)):
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, kernel_variant="wave", on_out_of_plane="raise")
    rgba = torch.rand((B, D, 4, S, S), device=dev).to(dt); rgba[:, -1, 3] = 1
    r.set_cam(r.cam_fov, S, S)
    torch.manual_seed(seed)
    cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
    dhw = r._dhw_on_device().expand(B, -1, -1).contiguous()
    status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
    with torch.no_grad():
        r.mpi.render_views(rgba, dhw, ray, eye, zd, check_last_plane=True, out_pm1=True, status=status, defer_status=True)
    torch.cuda.synchronize()
    st = status.cpu().tolist()
    print(S, dt, "status", st[0], "strips in half mode", st[2], "half strips in the gather", st[3], "of", B * (S // 32) * (S // 8), "strips")
