"""Dump the camera tensors of bench.py's cfg3 workload (and of a fixed-pose set) for tools/kbench.cpp, the torch-free kernel
timing harness (a gpurun call without `import torch` costs seconds instead of minutes).  Runs on CPU.
    python tools/kbench_dump.py  ->  gpurun_in/kb_<set>.bin   (not committed; travels with the snapshot)
File: int32 N, S, D; float32 focal; float32 dhw[D,3], c2w[N,4,4] (rays are rebuilt by the harness: K^-1 [x+.5, y+.5, 1], normalised, rotated)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ml_gmpi_amd

S, D, B = 1024, 96, 4
os.makedirs(os.path.join(ROOT, "gpurun_in"), exist_ok=True)
r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=torch.device("cpu"), ray_backend="torch") if False else None
from ml_gmpi_amd.renderer import MPIRenderer, PRESETS
kw = dict(PRESETS["FFHQ"])
kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
          mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
r = MPIRenderer(**kw)
r.set_cam(r.cam_fov, S, S)


def dump(name, cam):
    c2w, eye = cam[2].float().numpy(), torch.cat(cam[4]).float().numpy()
    dhw = r.static_mpi_plane_dhws.reshape(-1, 3).float().numpy()
    with open(os.path.join(ROOT, "gpurun_in", f"kb_{name}.bin"), "wb") as f:
        np.array([c2w.shape[0], r.cam.height, r.static_mpi_plane_dhws.reshape(-1, 3).shape[0]], dtype=np.int32).tofile(f)
        np.array([r.cam.intrinsic_matrix[0, 0]], dtype=np.float32).tofile(f)
        dhw.astype(np.float32).tofile(f); c2w.astype(np.float32).tofile(f)
    print(name, c2w.shape, eye.tolist())


torch.manual_seed(3)  # bench.py rank 0
dump("bench", r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True))
gy = torch.tensor([[0.0], [0.15], [-0.3], [0.45]]); gp = torch.tensor([[0.0], [0.05], [0.1], [-0.2]])
dump("fixed", r.sample_cam_poses(B, 0, 0, 0, 0, False, given_yaws=gy, given_pitches=gp))

# config 2: 8 views of 256^2 (bench.py's poses for that workload)
S = 256
r.set_cam(r.cam_fov, S, S)
torch.manual_seed(3)
dump("cfg2", r.sample_cam_poses(8, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True))

S = 512
r.set_cam(r.cam_fov, S, S)
torch.manual_seed(3)
dump("s512", r.sample_cam_poses(8, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True))

# config 5: MetFaces preset, 256 planes, 1024^2
kw = dict(PRESETS["MetFaces"])
kw.update(n_mpi_planes=256, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
          mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
r = MPIRenderer(**kw)
S, D = 1024, 256
r.set_cam(r.cam_fov, S, S)
torch.manual_seed(3)
dump("c5", r.sample_cam_poses(4, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True))

# pose sweep sets: every view at the same yaw, pitch 0 -- 8 views of 256^2 and 2 views of 512^2 per yaw
kw = dict(PRESETS["FFHQ"])
kw.update(n_mpi_planes=96, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
          mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
r = MPIRenderer(**kw)
for S, n in ((256, 8), (512, 2)):
    r.set_cam(r.cam_fov, S, S)
    for yaw in (0.0, 0.3, 0.45, 0.578):
        gy = torch.full((n, 1), yaw); gp = torch.zeros((n, 1))
        dump(f"p{S}_y{int(round(yaw * 1000)):03d}", r.sample_cam_poses(n, 0, 0, 0, 0, False, given_yaws=gy, given_pitches=gp))

# round 6 (band order against the plane count): the FFHQ preset with 256 planes under bench.py's pose draw
kw = dict(PRESETS["FFHQ"])
kw.update(n_mpi_planes=256, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
          mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
r = MPIRenderer(**kw)
r.set_cam(r.cam_fov, 1024, 1024)
torch.manual_seed(3)
dump("b256", r.sample_cam_poses(4, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True))

# config 4: 8 camera-path views of ONE 512^2 x 96 MPI (yaw sweep 0.5 ... -0.5, pitch 0: render_video.py:236-237); run with KB_VPM=8
kw = dict(PRESETS["FFHQ"])
kw.update(n_mpi_planes=96, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
          mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
r = MPIRenderer(**kw)
r.set_cam(r.cam_fov, 512, 512)
dump("c4", r.sample_cam_poses(8, 0, 0, 0, 0, False, given_yaws=torch.linspace(0.5, -0.5, 8).view(-1, 1), given_pitches=torch.zeros(8, 1)))
# ... and narrower sweeps of the same path (which kernel is best for views that share an MPI when every view fits the band kernel's boxes)
for name, lim in (("c4n", 0.25), ("c4m", 0.35)):
    dump(name, r.sample_cam_poses(8, 0, 0, 0, 0, False, given_yaws=torch.linspace(lim, -lim, 8).view(-1, 1), given_pitches=torch.zeros(8, 1)))
# ... and 4 views of 1024^2 that are ALL beyond the band kernel's reach (yaw 0.45) or alternate (0.45 / 0): the gated tile launch against the plain tile kernel
kw = dict(PRESETS["FFHQ"])
kw.update(n_mpi_planes=96, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
          mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
r = MPIRenderer(**kw)
r.set_cam(r.cam_fov, 1024, 1024)
dump("p1024_y450", r.sample_cam_poses(4, 0, 0, 0, 0, False, given_yaws=torch.full((4, 1), 0.45), given_pitches=torch.zeros((4, 1))))
dump("p1024_mix", r.sample_cam_poses(4, 0, 0, 0, 0, False, given_yaws=torch.tensor([[0.45], [0.0], [0.45], [0.0]]), given_pitches=torch.zeros((4, 1))))
