"""Timing of the auxiliary kernels (run on the GPU box): compute_depth, ray generation, range check, uint8 epilogue."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ml_gmpi_amd
from ml_gmpi_amd import _lib

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

dev = torch.device("cuda")
vol = torch.rand((4, 96, 4, 1024, 1024), device=dev)
ds = torch.linspace(0.95, 1.12, 96)
ms = timeit(lambda: ml_gmpi_amd.compute_depth(vol[:, :, 3:], ds))
b = 4 * 96 * 1024 * 1024 * 4
print(f"compute_depth 4x96x1024^2 fp32 alpha (strided view of RGBA): {ms:.3f} ms  {b/ms/1e6:.0f} GB/s of alpha bytes ({b/ms/1e6/8000:.1%} of 8 TB/s)")
r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=96, device=dev)
r.set_cam(r.cam_fov, 1024, 1024)
c2w = torch.eye(4, device=dev).repeat(8, 1, 1)
ms = timeit(lambda: r._generate_rays_hip(c2w))
print(f"generate_rays 8 views 1024^2: {ms:.3f} ms  ({8*1024*1024*12*2/ms/1e6:.0f} GB/s r+w)")
lib = _lib.load_library(); st = torch.zeros(4, dtype=torch.int32, device=dev)
ms = timeit(lambda: lib.gmpi_rgba_range_check_launch(vol.data_ptr(), 0, vol.numel(), st.data_ptr(), torch.cuda.current_stream().cuda_stream))
print(f"range_check full volume 6.4 GB: {ms:.3f} ms  {vol.numel()*4/ms/1e6:.0f} GB/s ({vol.numel()*4/ms/1e6/8000:.1%} of 8 TB/s)")
rgb = torch.rand((8, 3, 1024, 1024), device=dev) * 2 - 1; dep = torch.rand((8, 1, 1024, 1024), device=dev) + 0.5
ms = timeit(lambda: ml_gmpi_amd.frames_to_uint8(rgb, dep, 0.95, 1.12))
print(f"frames_to_uint8 8 frames 1024^2: {ms:.3f} ms")

# ---- backward at the reference's training sizes (gmpi.yml:78 D=32; curriculums.py:91-93 batch 8/4/4 @ 256/512/1024) ----
# (the G-step shapes -- forward + backward -- are bench.py --workload train256 / train512 / train1024 since round 5)

# ---- shading augmentation (LightRenderer.render) on a 4 x 96 x 1024^2 volume, kernel by kernel ----
import ctypes
from ml_gmpi_amd import light
from ml_gmpi_amd.hip_mpi import _DTYPES
del vol
for dt in (torch.float32, torch.bfloat16):
    B, D, S = 4, 96, 1024
    vol = torch.rand((B, D, 4, S, S), device=dev).to(dt)
    L = ml_gmpi_amd.LightRenderer(sphere_center_z=1.0, sphere_r=1.0, ka_max=0.6, kd_max=0.9, n_grow_iters=1)
    t_depth = timeit(lambda: light.compute_depth(vol[:, :, 3:], ds), 5)
    d = light.compute_depth(vol[:, :, 3:], ds)
    t_blur = timeit(lambda: L.blurrer_func(d), 5)
    xyz_last = torch.stack((torch.linspace(-0.25, 0.25, S).expand(S, S), torch.linspace(-0.2, 0.2, S).view(S, 1).expand(S, S),
                            torch.full((S, S), 1.12)), dim=-1).to(dev)
    ld = torch.tensor([[0.1, -0.2, 0.97]] * B)
    t_shade = timeit(lambda: L.shading(d, xyz_last, ld / ld.norm(dim=1, keepdim=True), 0.6, 0.9), 5)
    sh = torch.rand((B, S, S), device=dev)
    out = torch.empty((B, D, 4, S, S), device=dev)
    st_ = (ctypes.c_int64 * 5)(*vol.stride())
    t_apply = timeit(lambda: lib.gmpi_light_apply_launch(vol.data_ptr(), _DTYPES[vol.dtype], st_, sh.data_ptr(), out.data_ptr(), B, D, S, S,
                                                        torch.cuda.current_stream().cuda_stream), 5)
    es = vol.element_size()
    gb_apply = B * D * S * S * (4 * es + 16) / 1e9
    print(f"LightRenderer pieces {B}x{D}x{S}^2 {dt}: compute_depth {t_depth:.3f} ms, blur {t_blur:.3f} ms, shading {t_shade:.3f} ms, "
          f"apply {t_apply:.3f} ms = {gb_apply / t_apply * 1e3:.0f} GB/s read+write ({gb_apply / t_apply / 8:.1%} of 8 TB/s)")
    del vol, out
