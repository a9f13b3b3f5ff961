"""Times the backward launch (and the forward) at the G-step shapes through the C ABI.  usage: python tools/time_backward.py [lib.so]
(GMPI_TUNE_SKIP in the environment selects ablations of a profiling build: 16 no global atomics in the flush, 32 no LDS atomics, 64 the
round-1 tile kernel, 128 no tap loads.)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ml_gmpi_amd  # noqa: E402
from ml_gmpi_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib._SO = os.path.abspath(sys.argv[1])
    _lib._LIB = None
lib = _lib.load_library()
print("library:", _lib._SO, flush=True)
dev = torch.device("cuda:0")
for S, B in ((256, 8), (512, 4), (1024, 4)):
    D = 32
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
    r.set_cam(r.cam_fov, S, S)
    g = torch.Generator(device=dev).manual_seed(7000)
    vol = torch.rand((B, D, 4, S, S), device=dev, generator=g)
    vol[:, -1, 3] = 1.0
    gc = torch.randn((B, 3, S, S), device=dev, generator=g)
    gd = torch.randn((B, 1, S, S), device=dev, generator=g)
    torch.manual_seed(3)
    cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
    dhw = r._dhw_on_device().expand(B, -1, -1).contiguous()
    res = r.mpi.render_views(vol, dhw, ray, eye, zd, views_per_mpi=1, check_last_plane=True, out_pm1=True, want_transmittance=True,
                             defer_status=True, _in_autograd_fn=True)
    p, keep = res["_bwd"]
    p.rgb_out = p.depth_out = None
    need = int(lib.gmpi_render_backward_workspace_bytes(ctypes.byref(p))) if os.environ.get("BWD_GATHER") else 0   # BWD_GATHER=1: the atomics-free pair
    bws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None   # (what the autograd bridge passes: the atomics-free path writes every element)
    if bws is not None:
        p.workspace, p.workspace_bytes = bws.data_ptr(), bws.numel()
        p.flags |= _lib.FLAG_GRAD_OVERWRITE
    print("   backward workspace:", need, "bytes", flush=True)
    prof = torch.zeros(64, dtype=torch.int32, device=dev)
    p.status = prof.data_ptr() if os.environ.get("GMPI_PROF_WORDS") else None
    grad = torch.zeros_like(vol)
    gs = (ctypes.c_int64 * 5)(*grad.stride())
    cs = torch.cuda.current_stream(dev).cuda_stream

    def bwd():
        _lib.check(lib.gmpi_mpi_render_backward_launch(ctypes.byref(p), gc.data_ptr(), gd.data_ptr(), grad.data_ptr(), gs, cs), "bwd")
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:   # clock ramp
        for _ in range(8):
            bwd()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for e0, e1 in evs:
        e0.record(); bwd(); e1.record()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    print(f"backward {S}^2 x {D} x {B}: {ms:.4f} ms  (skip={os.environ.get('GMPI_TUNE_SKIP', '0')})", flush=True)
    if os.environ.get("GMPI_PROF_WORDS"):   # a -DGMPI_PROF build: shader-clock cycles of one pixel wave | one flush wave over the whole tile
        w = prof.cpu().tolist()
        print(f"   pixel wave: scatter {w[8]} grads {w[9]} fetch {w[10]} barrier {w[11]} prologue {w[13]} total {w[14]} | flush wave: flush {w[16]} barrier {w[19]} prologue {w[21]} total {w[22]}")
