#!/usr/bin/env python
"""bench.py -- throughput of the MPI render hot path on MI355X (one JSON line on rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg3_f32|cfg4|cfg5|video]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mpix*planes/s = views*H*W*D / t / 1e6 (whole job, all GPUs), plus views/s and
the fraction of the HBM roofline.  A "step" = ONE pass of the hot path over one batch of views
(one `gmpi_mpi_render_launch`: warp + composite + depth + asserts fused in one render kernel -- for large bf16 / fp32 launches the band
kernel behind its small geometry-table kernel, plus the tile kernel's gated launch for views the band kernel cannot stage), inputs
resident in HBM.  Before the W warm-up steps the
same launch is repeated for --prewarm-ms (default 250 ms, untimed) so that a GPU coming from idle has reached its busy clocks.

Default workload = BASELINE.json configs[2] ("FFHQ1024-shaped: 1024x1024, 96 planes, batch 4 views,
bf16", the configuration the north-star target is quoted on); each rank renders its own batch
(independent views/seeds -- own volumes per rank, the same pose draw on every rank so that the per-GPU work is exactly
fixed -- no data-path collective -> "weak" scaling).  The final frame all_gather
(RCCL) is timed separately (`gather_ms`), outside the K timed steps.

roofline.achieved = ALGORITHMIC bytes per launch / the larger of (average duration of a step's kernels: HIP events around every
launch call on the launch stream; wall time per step).  Algorithmic bytes:
    N*D*4*Ht*Wt*s_in  +  N*H*W*12 (ray_dir)  +  N*H*W*4*(3+1[+1]) (outputs)
cpu_baseline = the CPU oracle (oracle/mpi_oracle.c, OpenMP build, kind "port") timed on this box's
host cores on a bounded sample of the same workload -- a reported baseline, not the target.

Further blocks of the line, all OUTSIDE the K timed steps (rank 0, N = 1 unless noted):
    parity      the timed launch itself (same tensors, same variant) rendered once in strict-order mode and once in default mode and
                compared with the CPU oracle on three 64x64 ray windows per view: strict_bit_exact, max_abs_err_{color,depth,T}, bar 1e-5.
                A failure makes the process exit non-zero (--no-parity skips it: the PMC passes of tools/prof.sh).
    pose_sweep  --pose-draws (default 32) seeded draws of the preset's pose distribution through the same launch: mean / p50 / p90 /
                worst ms per step and the share of views the band kernel handed to the tile kernel (read from the workspace header).
                The headline `value` stays on the draw SURVEY.md 8(d) prescribes (torch.manual_seed(3)).
    e2e_render  `MPIRenderer.render()` through the product's host path (pose draw, ray kernel, launch, lagged status): mean and max over
                fresh draws, and back to back without a synchronisation in between (what a loop of calls pays).
    --workload video  the reference's render_video.py loop shape (one 512^2 view per call over a yaw sweep, `.cpu()` per view) through the
                installed drop-in, next to ViewBatchDriver.render_path on the same path: views/s of both (metric "views/s").
    --dry-run   no GPU: the rank / rendezvous / gather / JSON logic with a launch-counting stand-in library on the gloo backend
                (tests/test_bench_ranks_gloo.py runs it at world size 8); --numa-pin binds a rank to the NUMA node of its GPU.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)

WORKLOADS = {
    # name: (preset, S (=H=W=Ht=Wt), D, views per GPU, storage dtype, want_T, description)
    "cfg2": ("FFHQ", 256, 96, 8, "f32", False, "FFHQ256-shaped: 256x256, 96 planes, batch 8 views, fp32"),
    "cfg3": ("FFHQ", 1024, 96, 4, "bf16", False, "FFHQ1024-shaped: 1024x1024, 96 planes, batch 4 views, bf16 storage"),
    "cfg3_f32": ("FFHQ", 1024, 96, 4, "f32", False, "FFHQ1024-shaped: 1024x1024, 96 planes, batch 4 views, fp32 storage"),
    "cfg4": ("FFHQ", 512, 96, 8, "f32", False, "FFHQ512-shaped: 512x512, 96 planes, 8 camera-path views of ONE MPI per GPU"),
    "cfg5": ("MetFaces", 1024, 256, 4, "f32", True, "MetFaces-shaped: 1024x1024, 256 planes + depth/transmittance, 4 seeds per GPU"),
}


# G-step shapes of the reference's curriculum (curriculums.py:89-91: 32 planes; batch 8 at 256^2, 4 at 512^2 and 1024^2; train.py:740-779: render
# with grad, loss, backward): forward + gradient w.r.t. the RGBA volume, fp32.
TRAIN_WORKLOADS = {
    # name: (preset, S, D, batch, description)
    "train256": ("FFHQ", 256, 32, 8, "G-step render 256x256, 32 planes, batch 8, fp32: forward + backward w.r.t. the RGBA volume"),
    "train512": ("FFHQ", 512, 32, 4, "G-step render 512x512, 32 planes, batch 4, fp32: forward + backward w.r.t. the RGBA volume"),
    "train1024": ("FFHQ", 1024, 32, 4, "G-step render 1024x1024, 32 planes, batch 4, fp32: forward + backward w.r.t. the RGBA volume"),
}


def algorithmic_bytes(n_views, D, S, s_in, want_T):
    return n_views * D * 4 * S * S * s_in + n_views * S * S * 12 + n_views * S * S * 4 * (4 + (1 if want_T else 0))


def footprint_bytes(ray, eye, dhw, S, s_in, want_T):
    """Conservative companion of the algorithmic bytes: only the texel bounding box each view touches on
    each plane (from the 4 image-corner rays, align_corners=True grid), plus the per-pixel bytes."""
    n = ray.shape[0]
    r = ray.double().cpu()[:, :, [0, 0, -1, -1], [0, -1, 0, -1]]          # [N,3,4] corner rays
    e = eye.double().cpu()
    d = dhw.double().cpu()[0]                                               # [D,3] (same planes for every MPI here)
    sc = (d[None, :, 0, None] - e[:, None, 2, None]) / r[:, None, 2, :]      # [N,D,4]
    x = e[:, None, 0, None] + r[:, None, 0, :] * sc
    y = e[:, None, 1, None] + r[:, None, 1, :] * sc
    ix = (2 * x / d[None, :, 2, None] + 1) * (S - 1) / 2
    iy = (2 * y / d[None, :, 1, None] + 1) * (S - 1) / 2
    w = (ix.max(-1).values.floor() + 1).clamp(0, S - 1) - ix.min(-1).values.floor().clamp(0, S - 1) + 1
    h = (iy.max(-1).values.floor() + 1).clamp(0, S - 1) - iy.min(-1).values.floor().clamp(0, S - 1) + 1
    texels = float((w.clamp(min=0) * h.clamp(min=0)).sum())
    return int(texels * 4 * s_in + n * S * S * 12 + n * S * S * 4 * (4 + (1 if want_T else 0)))


def cpu_baseline(preset, S, D, dtype, budget_s=20.0):
    """The CPU side of the line, on ONE view of the workload shape, ~budget_s of CPU wall time in total:
      headline entry (kind "reference-ops"): the reference's own op chain (MPIRenderer.render -> MPI.forward -> homography: the same PyTorch CPU
        calls in the same order, oracle/torch_ops.py, pinned to fixtures made by the reference itself) -- the reference is not on the GPU box,
        its ops are; best of a small sweep of intra-op thread counts, `cores` = the count that won;
      nested `port`: the repo's own restatement (oracle/mpi_oracle.c, OpenMP over all host threads) -- what a hand-written CPU renderer reaches."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from ml_gmpi_amd.renderer import MPIRenderer, PRESETS

    oracle.build()
    kw = dict(PRESETS[preset])
    kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse",
              cam_sample_method="truncated_gaussian", mpi_align_corners=True, use_confined_volume=True,
              device=torch.device("cpu"))
    r = MPIRenderer(**kw)
    r.set_cam(r.cam_fov, S, S)
    torch.manual_seed(0)
    cam = r.sample_cam_poses(1, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    rgba = torch.rand((1, D, 4, S, S))
    if dtype == "bf16":
        rgba = rgba.to(torch.bfloat16).float()
    dhw = r.static_mpi_plane_dhws.reshape(1, -1, 3)
    args = (rgba.numpy(), dhw.numpy(), cam[3][0].numpy(), cam[4][0].numpy(), cam[5][0].numpy())
    oracle.render(*args, threads=True)  # warm-up (page-in)
    reps, t0 = 0, time.perf_counter()
    best = float("inf")
    while True:
        t1 = time.perf_counter()
        oracle.render(*args, threads=True)
        best = min(best, time.perf_counter() - t1)
        reps += 1
        if time.perf_counter() - t0 > 0.35 * budget_s or reps >= 20:
            break
    port = dict(value=round(S * S * D / best / 1e6, 2), unit="Mpix*planes/s", cores=oracle.num_threads(True),
                kind="port", sample=f"1 view {S}x{S}x{D} ({dtype} values), best of {reps} runs of oracle/mpi_oracle.c (OpenMP)",
                views_per_s=round(1.0 / best, 4))
    # the reference's own op chain, all host cores available, one view, warm-up + best of <= 2 per thread count
    try:
        import torch_ops
        nproc = os.cpu_count() or 1
        t_args = (rgba, r.static_mpi_plane_dhws, cam[3][0], cam[4][0], cam[5][0])
        sweep = {}
        with torch.no_grad():
            # PyTorch's intra-op pool oversubscribed (256 threads on a 128-core box) runs this memory-bound chain 3x slower than 8 threads:
            # the baseline is the BEST of a small thread sweep, each count warmed up once and timed at most twice within the budget
            t_sweep = time.perf_counter()
            for nt in sorted({min(n, nproc) for n in (8, 32, 64, nproc)}):
                torch.set_num_threads(nt)
                torch_ops.renderer_render(*t_args)
                best_nt = float("inf")
                for _ in range(2):
                    t1 = time.perf_counter()
                    torch_ops.renderer_render(*t_args)
                    best_nt = min(best_nt, time.perf_counter() - t1)
                    if time.perf_counter() - t_sweep > budget_s:
                        break
                sweep[nt] = best_nt
                if time.perf_counter() - t_sweep > budget_s:
                    break
        nthreads = min(sweep, key=sweep.get)
        best_ops = sweep[nthreads]
        return dict(value=round(S * S * D / best_ops / 1e6, 2), unit="Mpix*planes/s", cores=nthreads, kind="reference-ops",
                    sample=f"1 view {S}x{S}x{D} ({dtype} values): the reference's op chain (mpi_renderer.py:444-467, mpi.py:60-153, 321-436) "
                           "as the same PyTorch CPU calls, oracle/torch_ops.py; best of the thread counts in `thread_sweep`",
                    thread_sweep={str(k): round(S * S * D / v / 1e6, 2) for k, v in sweep.items()}, host_cores=nproc,
                    views_per_s=round(1.0 / best_ops, 4), port=port)
    except Exception as e:  # memory (10 GB per 1024^2 x 96 view) or a missing module must not cost the bench line: the port stands in
        port["reference_ops_error"] = str(e)[:200]
        return port


def run_video(a, dev, rank, world, use_dist):
    """--workload video: the camera-path loop of the reference's video script on the GPU, in its own shape -- ONE MPI (512^2 x 96, fp32), one
    view per `render()` call with `horizontal_mean = angle`, std 0, and a device-to-host copy + uint8 conversion per view
    (eval/vis/render_video.py:95-130, `generate_img`) -- next to the batched driver (`ViewBatchDriver.render_path`: 8 views per launch, uint8
    epilogue on the device, one copy at the end).  A "step" = one pass over this rank's share of the 64-view path (BASELINE config 4)."""
    import numpy as np
    import ml_gmpi_amd
    S, D, n_path = 512, 96, 64
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, kernel_variant=a.variant, on_out_of_plane="raise")
    g = torch.Generator(device=dev).manual_seed(4000)
    rgba = torch.rand((1, D, 4, S, S), device=dev, generator=g)
    rgba[:, -1, 3] = 1.0
    angles = np.linspace(0.5, -0.5, n_path).tolist()[rank::world]
    near, far = r.plane_min_d, r.plane_max_d

    t_parts = [0.0, 0.0]   # seconds inside render() | until both frames are on the host (the rest of a pass is the script's own numpy)

    def per_call_pass():  # render_video.py:95-130, line by line
        frames = []
        for ang in angles:
            t_a = time.perf_counter()
            img, depth, _, _ = r.render(rgba, S, S, horizontal_mean=ang, horizontal_std=0.0, vertical_mean=0.0, vertical_std=0.0,
                                        assert_not_out_of_last_plane=True)
            t_b = time.perf_counter()
            img, depth = img.permute(0, 2, 3, 1).squeeze().cpu(), depth.permute(0, 2, 3, 1).squeeze().cpu()
            t_c = time.perf_counter()
            t_parts[0] += t_b - t_a
            t_parts[1] += t_c - t_b
            img = img.numpy()
            img = ((img + 1) / 2.0 * 255).astype(np.uint8)
            dm = depth.numpy()
            dm = (np.clip((dm - near) / (far - near), 0, 1)[..., None] * 255).astype(np.uint8)
            frames.append((img, dm))
        return frames

    drv = ml_gmpi_amd.ViewBatchDriver(r, batch=8)

    def batched_pass():
        # (round 6: the uint8 frames of a batch travel to pinned host buffers on a second stream while the next batch renders)
        res = drv.render_path(rgba, S, angles, [0.0] * len(angles), to_uint8=True, depth_range=(near, far), to_host=True)
        return res["img8_host"], res["dep8_host"]

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(max(a.warmup, 1)):
            per_call_pass(); batched_pass()
        fence(); t0 = time.perf_counter()
        t_parts[0] = t_parts[1] = 0.0
        for _ in range(a.steps):
            per_call_pass()
        fence(); t_call = time.perf_counter() - t0
        in_render, to_host = t_parts[0], t_parts[1]
        # (the per-call loop keeps the GPU 10 % busy: its clocks have dropped, and a batched pass is only ~7 ms -- untimed passes for
        #  --prewarm-ms first, as the render workloads do; without them the figure is bimodal, 5 000 or 9 000 views/s from run to run)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < a.prewarm_ms:
            batched_pass()
        n_batch = max(a.steps, 20)
        fence(); t0 = time.perf_counter()
        for _ in range(n_batch):
            batched_pass()
        fence(); t_batch = (time.perf_counter() - t0) * a.steps / n_batch
        a_frames, b_frames = per_call_pass(), batched_pass()
    same = all(np.array_equal(f[0], b_frames[0][i].numpy()) for i, f in enumerate(a_frames))  # the two paths give the same uint8 frames
    t = torch.tensor([t_call, t_batch], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_call, t_batch = float(t[0]), float(t[1])
    if rank == 0:
        views = n_path * a.steps
        line = {"metric": "Mpix*planes/s", "value": round(views * S * S * D / t_call / 1e6, 1), "unit": "Mpix*planes/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": round(t_call / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": "FFHQ512-shaped video path: 512x512, 96 planes, 64 camera-path views of ONE MPI, one view per render() call + "
                                       ".cpu() per view (render_video.py:95-130)", "name": "video", "H": S, "W": S, "planes": D, "rgba_storage": "f32",
                           "variant": a.variant, "parallelism": f"views of the path strided over {world} rank(s)"},
                "views_per_s": round(views / t_call, 1), "ms_per_view": round(t_call / views * world * 1e3, 4),
                "per_view_ms": {"render_call_host": round(in_render / (len(angles) * a.steps) * 1e3, 4), "sync_and_copy_to_host": round(to_host / (len(angles) * a.steps) * 1e3, 4),
                                "script_numpy": round((t_call - in_render - to_host) / (len(angles) * a.steps) * 1e3, 4),
                                "what": "render(): host time of the call | .cpu() x 2: waits for the kernel, copies 4 MB | the script's own uint8 conversion (numpy)"},
                "render_and_copy_views_per_s": round(len(angles) * a.steps * world / (in_render + to_host), 1),
                "batched_driver": {"views_per_s": round(views / t_batch, 1), "ms_per_view": round(t_batch / views * world * 1e3, 4), "batch": 8,
                                   "what": "ViewBatchDriver.render_path(to_host=True): 8 views per launch, uint8 epilogue on the device per batch, copies into pinned host buffers on a second stream next to the next batch's render"},
                "frames_identical": bool(same), "roofline": None, "cpu_baseline": None}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


class TrainWorkload:
    """The render of the reference's G-step (train.py:740-779: `MPIRenderer.render` on the generator's RGBA volume under autograd, a loss on the
    frames, `.backward()`) at one of the curriculum's shapes (curriculums.py:89-91: 32 planes, batch 8 / 4 / 4), set up on a device.
    `step()` = forward launch (colour, depth, transmittance) + zero-fill of the gradient volume + `gmpi_mpi_render_backward_launch`, exactly what
    `hip_mpi._RenderFunction` enqueues (driven through the product's own autograd bridge); `parts()` times each of the three alone through the C ABI."""

    def __init__(self, name, dev, rank=0, variant="auto"):
        import ml_gmpi_amd
        preset, S, D, B, desc = TRAIN_WORKLOADS[name]
        self.name, self.S, self.D, self.B, self.desc, self.dev, self.variant = name, S, D, B, desc, dev, variant
        r = self.r = ml_gmpi_amd.make_renderer(preset, n_planes=D, device=dev, kernel_variant=variant, on_out_of_plane="raise")
        r.set_cam(r.cam_fov, S, S)
        g = torch.Generator(device=dev).manual_seed(7000 + rank)
        rgba = torch.rand((B, D, 4, S, S), device=dev, generator=g)
        rgba[:, -1, 3] = 1.0
        self.rgba = rgba.requires_grad_(True)
        self.g_color = torch.randn((B, 3, S, S), device=dev, generator=g)
        self.g_depth = torch.randn((B, 1, S, S), device=dev, generator=g)
        torch.manual_seed(3)
        cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
        self.infos = dict(zip(["batch_yaws", "batch_pitches", "batch_tf_c2w", "batch_ray_dir", "batch_eye_pos", "batch_z_dir"], cam))
        self.ray, self.eye, self.zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
        self.dhw = r._dhw_on_device().expand(B, -1, -1).contiguous()
        vol_b, pix = B * D * 4 * S * S * 4, B * S * S
        # algorithmic bytes: volume read by the forward | volume read by the backward + gradient volume written; per pixel rays 12 + frames out 20
        # (forward), rays 12 + upstream gradients 16 + transmittance 4 (backward)
        self.ab_fwd, self.ab_bwd = vol_b + pix * (12 + 20), 2 * vol_b + pix * (12 + 16 + 4)
        self._abi = None

    def step(self):
        self.rgba.grad = None
        color, depth, _, _ = self.r.render(self.rgba, self.S, self.S, given_cam_infos=self.infos, defer_status=True)
        torch.autograd.backward([color, depth], [self.g_color, self.g_depth])

    def abi(self):
        """The step's launches through the C ABI (what the bridge enqueues): fwd(), and bwd(g_color, g_depth, grad) on a parameter struct of the forward."""
        if self._abi is None:
            import ctypes
            from ml_gmpi_amd import _lib
            lib = _lib.load_library()
            B, S, dev = self.B, self.S, self.dev
            out = dict(color=torch.empty((B, 3, S, S), device=dev), depth=torch.empty((B, 1, S, S), device=dev), T=torch.empty((B, 1, S, S), device=dev))
            status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
            vol = self.rgba.detach()

            def fwd(variant=None):
                mpi = self.r.mpi
                was = mpi.variant
                if variant is not None:
                    mpi.variant = variant
                try:
                    return mpi.render_views(vol, self.dhw, self.ray, self.eye, self.zd, views_per_mpi=1, check_last_plane=True, out_pm1=True,
                                            want_transmittance=True, status=status, defer_status=True, out=out, _in_autograd_fn=True)
                finally:
                    mpi.variant = was
            pstruct, keep = fwd()["_bwd"]
            pstruct.rgb_out = pstruct.depth_out = pstruct.status = None
            grad = torch.zeros_like(vol)
            # what the autograd bridge does: MPI(backward="gather") lends the atomics-free pair its scratch (then no zero-fill); the default adds atomically
            need = int(lib.gmpi_render_backward_workspace_bytes(ctypes.byref(pstruct))) if self.r.mpi.backward == "gather" else 0
            bws = torch.empty(need, dtype=torch.uint8, device=dev) if need else None
            if bws is not None:
                pstruct.workspace, pstruct.workspace_bytes = bws.data_ptr(), bws.numel()
                pstruct.flags |= _lib.FLAG_GRAD_OVERWRITE
            gstride = (ctypes.c_int64 * 5)(*grad.stride())
            cs = torch.cuda.current_stream(dev).cuda_stream

            def bwd(gc=self.g_color, gd=self.g_depth, into=grad, variant=None):
                was = pstruct.variant
                if variant is not None:
                    pstruct.variant = _lib.VARIANTS[variant]
                try:
                    _lib.check(lib.gmpi_mpi_render_backward_launch(ctypes.byref(pstruct), gc.data_ptr(), gd.data_ptr(), into.data_ptr(), gstride, cs),
                               "gmpi_mpi_render_backward_launch")
                finally:
                    pstruct.variant = was
            self._abi = dict(fwd=fwd, bwd=bwd, grad=grad, status=status, keep=keep, pstruct=pstruct, bwd_workspace=bws, needs_zero_fill=bws is None)
        return self._abi

    def parts(self, n=20):
        """Each part alone: n launches behind 3 warm-ups, HIP events on the launch stream."""
        k = self.abi()
        dev = self.dev

        def timed(fn):
            for _ in range(3):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for e0, e1 in evs:
                e0.record(); fn(); e1.record()
            torch.cuda.synchronize(dev)
            return sum(e0.elapsed_time(e1) for e0, e1 in evs) / n
        with torch.no_grad():
            fwd_ms, zero_ms, bwd_ms = timed(k["fwd"]), timed(k["grad"].zero_), timed(k["bwd"])
            # the opt-in atomics-free pair (MPI(backward="gather"): pixel pass + texel gather, no zero-fill, bit-reproducible), for the record
            gather_ms = None
            if k["needs_zero_fill"]:
                import ctypes
                from ml_gmpi_amd import _lib
                lib = _lib.load_library()
                ps = k["pstruct"]
                need = int(lib.gmpi_render_backward_workspace_bytes(ctypes.byref(ps)))
                if need:
                    try:
                        ws = torch.empty(need, dtype=torch.uint8, device=dev)
                        keep = (ps.workspace, ps.workspace_bytes, ps.flags)
                        ps.workspace, ps.workspace_bytes, ps.flags = ws.data_ptr(), need, ps.flags | _lib.FLAG_GRAD_OVERWRITE
                        gather_ms = timed(k["bwd"])
                        ps.workspace, ps.workspace_bytes, ps.flags = keep
                        del ws
                    except torch.cuda.OutOfMemoryError:
                        pass
        self.r.mpi.raise_on_status(k["status"])
        return dict(forward_ms=round(fwd_ms, 4), forward_frac=round(self.ab_fwd / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    grad_zero_fill_ms=round(zero_ms, 4), backward_ms=round(bwd_ms, 4),
                    backward_frac=round(self.ab_bwd / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), backward_algorithmic_bytes=self.ab_bwd,
                    backward_gather_ms=None if gather_ms is None else round(gather_ms, 4),
                    what=f"each part alone: {n} launches, HIP events on the launch stream; backward_gather_ms: the opt-in atomics-free pair (no zero-fill)")

    def parity(self, n_windows=2, w=64):
        """The backward against (1) the all-atomic one-pixel-per-lane kernel (GMPI_VARIANT_GATHER: no staging, no tiles, 16 global fp32 atomics per
        pixel and plane) on the TIMED launch's own upstream gradients -- the whole gradient volume -- and (2) the reference's own op chain under
        torch autograd on the host (oracle/torch_ops.py: F.grid_sample, cumprod, the reference's sums; fp32) for upstream gradients restricted to one
        w x w pixel window per checked view (a full 1024^2 view would need minutes and 20 GB on the host): the same kernel, volume, cameras and
        transmittance input; errors relative to the largest gradient.  Bars: 1e-5 (kernel against kernel, both fp32 sums in different orders),
        5e-5 (fp32 host chain on white noise; tests/test_hip_backward.py pins the same quantity on the reference's fixtures at 3e-5)."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch_ops
        k = self.abi()
        dev, S, B = self.dev, self.S, self.B
        with torch.no_grad():
            k["fwd"]()                                   # (the transmittance the backward starts from)
            k["grad"].zero_(); k["bwd"]()
            other = torch.zeros_like(k["grad"])
            k["bwd"](into=other, variant="gather")
            torch.cuda.synchronize(dev)
            scale = float(other.abs().max())
            err_kernel = float((k["grad"] - other).abs().max()) / scale
            del other
        err_ref = 0.0
        for i in range(min(n_windows, B)):
            n = (i * 3) % B
            y0, x0 = (S // 2 - w // 2, min(S // 2 + 7, S - w)) if i % 2 == 0 else (S - w, 0)
            gc, gd = torch.zeros_like(self.g_color), torch.zeros_like(self.g_depth)
            gc[n, :, y0:y0 + w, x0:x0 + w] = self.g_color[n, :, y0:y0 + w, x0:x0 + w]
            gd[n, :, y0:y0 + w, x0:x0 + w] = self.g_depth[n, :, y0:y0 + w, x0:x0 + w]
            with torch.no_grad():
                k["grad"].zero_(); k["bwd"](gc=gc, gd=gd)
                got = k["grad"][n].cpu()
            vol = self.rgba.detach()[n:n + 1].cpu().requires_grad_(True)
            color, depth = torch_ops.mpi_forward(vol, self.dhw[n:n + 1].cpu(), self.ray[n:n + 1, :, y0:y0 + w, x0:x0 + w].contiguous().cpu(),
                                                 self.eye[n:n + 1].cpu(), self.zd[n:n + 1].cpu())
            # (the forward hands out 2 c - 1, mpi_renderer.py:467: the upstream gradient of the frame reaches c doubled)
            loss = (2.0 * color * gc[n:n + 1, :, y0:y0 + w, x0:x0 + w].cpu()).sum() + (depth * gd[n:n + 1, :, y0:y0 + w, x0:x0 + w].cpu()).sum()
            loss.backward()
            ref = vol.grad[0]
            err_ref = max(err_ref, float((got - ref).abs().max()) / float(ref.abs().max()))
        return dict(ok=bool(err_kernel <= 1e-5 and err_ref <= 5e-5), vs_all_atomic_kernel=float(f"{err_kernel:.3e}"), vs_reference_autograd=float(f"{err_ref:.3e}"),
                    bars=[1e-5, 5e-5], windows=f"{min(n_windows, B)} views x one {w}x{w} pixel window of upstream gradients vs oracle/torch_ops.py under torch autograd (host, fp32); "
                                               "the timed launch's full gradient vs GMPI_VARIANT_GATHER", scale="relative to the largest |gradient|")


def run_train(a, dev, rank, world, use_dist):
    """--workload train256 | train512 | train1024: see TrainWorkload.  The zero-fill the C ABI asks of the caller and the read half of the atomics'
    read-modify-write are traffic, not algorithm: they show in `traffic`, not in `achieved`."""
    from ml_gmpi_amd import _lib
    w = TrainWorkload(a.workload, dev, rank, a.variant)
    S, D, B, desc = w.S, w.D, w.B, w.desc
    step = w.step

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < a.prewarm_ms:
        step()
        torch.cuda.synchronize(dev)
    for _ in range(a.warmup):
        step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    fence()
    t0 = time.perf_counter()
    for e0, e1 in ev:
        e0.record(); step(); e1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / a.steps
    grad_ref = w.rgba.grad.detach().clone()

    parts = w.parts()
    k = w.abi()
    with torch.no_grad():
        k["grad"].zero_(); k["bwd"]()
        torch.cuda.synchronize(dev)
        same = float((k["grad"] - grad_ref).abs().max() / grad_ref.abs().max())   # (atomics: the order of the adds differs from run to run)
    del grad_ref

    t = torch.tensor([elapsed, kern_ms], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(t[0]), float(t[1])
    if rank == 0:
        ab_fwd, ab_bwd = w.ab_fwd, w.ab_bwd
        step_ms = max(kern_ms, elapsed / a.steps * 1e3)
        ach = (ab_fwd + ab_bwd) / (step_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.isfile(prof):
            try:
                pj = json.load(open(prof))
                if pj.get("source_hash") == _lib.source_hash():
                    traffic = (pj.get("workloads", {}).get(a.workload) or {}).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                pass
        line = {"metric": "Mpix*planes/s", "value": round(B * world * S * S * D * a.steps / elapsed / 1e6, 1), "unit": "Mpix*planes/s", "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": desc + " (curriculums.py:89-91, train.py:740-779)", "name": a.workload, "views_per_gpu": B, "H": S, "W": S, "planes": D,
                           "rgba_storage": "f32", "variant": a.variant, "parallelism": f"batch sharded x{world}", "step": "forward + zero-fill + backward, via autograd"},
                "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                             "priced_on_ms": round(step_ms, 4), "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": ab_fwd + ab_bwd, "traffic": traffic,
                             "parts": parts},
                "backward_repeatability": {"max_abs_diff_over_max_grad": float(f"{same:.3e}"), "what": "two runs of the backward (atomic adds in a different order)"},
                "cpu_baseline": None}
        if not a.no_parity:
            line["parity"] = w.parity()
        print(json.dumps(line), flush=True)
        parity_failed = bool(line.get("parity")) and not line["parity"]["ok"]
    else:
        parity_failed = False
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: the backward does not match its references (see `parity` in the line above)")


class _DryLibrary:
    """--dry-run: stands in for libgmpi_render.so (every launch is counted and returns GMPI_OK): the rank logic of this script -- argument
    parsing, per-rank seeding and sharding, barriers, all_reduce(MAX), all_gather_into_tensor, the rank-0 line -- runs on CPU tensors under
    `gloo` at any world size (tests/test_bench_ranks_gloo.py: 8 ranks).  No number of such a run means anything."""
    records_only = True

    def __init__(self):
        self.launches = 0

    def gmpi_mpi_render_launch(self, pref, stream):
        self.launches += 1
        return 0

    def gmpi_rgba_range_check_launch(self, *args):
        return 0

    def gmpi_render_workspace_bytes(self, pref):
        return 0


class _WallEvent:
    """torch.cuda.Event's two methods on the host clock (dry runs)."""
    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def numa_cpus_of_pci(pci: str, sysfs: str = "/sys"):
    """(NUMA node, cpu ids, cpulist text) of a PCI device out of sysfs: <sysfs>/bus/pci/devices/<pci>/numa_node and
    <sysfs>/devices/system/node/node<N>/cpulist ("0-31,64-95").  node < 0: the platform reports no affinity -> (node, None, None)."""
    node = int(open(os.path.join(sysfs, "bus/pci/devices", pci.lower(), "numa_node")).read().strip())
    if node < 0:
        return node, None, None
    cpus = open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read().strip()
    ids = set()
    for part in cpus.split(","):
        lo, _, hi = part.partition("-")
        ids.update(range(int(lo), int(hi or lo) + 1))
    return node, ids, cpus


def numa_pin(local_rank: int, enable: bool, dry: bool = False):
    """Pins this rank's host threads to the cores of the NUMA node its GPU hangs off (sysfs: /sys/class/drm/card*/device/numa_node through the
    device's PCI address; `rocm-smi --showtoponuma` shows the same).  One process per GPU otherwise floats over all sockets and its launch
    latency depends on where the scheduler left it.  Returns a description for the bench line; never fails the run.
    --dry-run (no GPU): the PCI address of rank i comes from GMPI_BENCH_PCI_IDS (comma separated) and the sysfs root from GMPI_BENCH_SYSFS -- the
    lookup and the parsing run (tests/test_bench_ranks_gloo.py builds a fake tree for 8 ranks on 2 nodes), the affinity is reported, not applied."""
    if not enable:
        return None
    try:
        sysfs = os.environ.get("GMPI_BENCH_SYSFS", "/sys") if dry else "/sys"
        if dry:
            ids_env = [x for x in os.environ.get("GMPI_BENCH_PCI_IDS", "").split(",") if x]
            if not ids_env:
                return "unavailable (dry run without GMPI_BENCH_PCI_IDS)"
            pci = ids_env[local_rank % len(ids_env)]
        else:
            pci = torch.cuda.get_device_properties(local_rank).pci_bus_id if hasattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id") else None
            if pci is None:
                import ctypes
                buf = ctypes.create_string_buffer(64)
                hip = ctypes.CDLL("libamdhip64.so")
                if hip.hipDeviceGetPCIBusId(buf, 64, local_rank) != 0:
                    return "unavailable (hipDeviceGetPCIBusId failed)"
                pci = buf.value.decode()
        node, ids, cpus = numa_cpus_of_pci(pci, sysfs)
        if node < 0:
            return f"gpu {local_rank} ({pci}): no NUMA affinity reported"
        if not dry:
            ids &= os.sched_getaffinity(0)
        if not ids:
            return f"gpu {local_rank} ({pci}): node {node} has no allowed cpu"
        if not dry:
            os.sched_setaffinity(0, ids)
        return f"gpu {local_rank} ({pci}) -> NUMA node {node}, {len(ids)} cpus ({cpus})"
    except Exception as e:  # noqa: BLE001 -- a missing sysfs entry must not cost the bench line
        return f"unavailable ({type(e).__name__}: {e})"


def pose_hints(zd):
    """The advisory flags MPIRenderer.render derives from the poses it has drawn (renderer.py: _COS_FRONTAL, _COS_TILTED)."""
    from ml_gmpi_amd import renderer as _r
    m = float(zd[:, 2].min())
    return dict(frontal_hint=m >= _r._COS_FRONTAL, tilted_hint=m < _r._COS_TILTED, oblique_hint=m < _r._COS_OBLIQUE)


PARITY_BAR = 1e-5  # BASELINE.json north_star: "within 1e-5 fp32" (colour on the [-1, 1] scale of MPIRenderer.render, depth in scene units)


def parity_block(r, rgba, dhw, ray, eye, zd, vpm, want_T, S):
    """The bench's OWN launch (same tensors, same flags, GMPI_VARIANT as timed) rendered once more in strict-order mode and once in default
    mode, outside the timed region, and compared with the CPU oracle on three 64 x 64 ray windows of every view (the oracle needs minutes
    per full 1024^2 x 96 view).  Strict mode must be bit-identical, default mode within PARITY_BAR.  """
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle
    from ml_gmpi_amd import _lib
    oracle.build()
    dev = rgba.device
    n_views = ray.shape[0]
    outs = {}
    was = r.mpi.strict_order
    for strict in (True, False):
        r.mpi.strict_order = strict
        st = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        outs[strict] = r.mpi.render_views(rgba, dhw, ray, eye, zd, views_per_mpi=vpm, check_last_plane=True, out_pm1=True,
                                          want_transmittance=True, status=st, defer_status=True, **pose_hints(zd))
        r.mpi.raise_on_status(st)
    r.mpi.strict_order = was
    w = min(64, S)
    wins = sorted({(0, 0), (S - w, S - w), (max(S // 2 - w // 2, 0), min(S // 2 + 7, S - w))})
    res = dict(strict_bit_exact=True, max_abs_err_color=0.0, max_abs_err_depth=0.0, max_abs_err_T=0.0, bar=PARITY_BAR,
               windows=f"{len(wins)} windows of {w}x{w} pixels per view x {n_views} views vs oracle/mpi_oracle.c", color_scale="[-1, 1]")
    vol_cache = {}
    for n in range(n_views):
        m = n // vpm
        if m not in vol_cache:
            vol_cache.clear()
            vol_cache[m] = rgba[m:m + 1].float().cpu().numpy()   # exact upcast of the stored values (mpi_renderer.py:446)
        vol = vol_cache[m]
        for (y0, x0) in wins:
            win = ray[n:n + 1, :, y0:y0 + w, x0:x0 + w].contiguous().cpu()
            orc = oracle.render(vol, dhw[m:m + 1].cpu(), win, eye[n:n + 1].cpu(), zd[n:n + 1].cpu(), threads=True)
            ref = dict(color=np.float32(2.0) * orc["color"] - np.float32(1.0), depth=orc["depth"], T=orc["T"])   # mpi_renderer.py:467
            for key in ("color", "depth", "T"):
                got_s = outs[True][key][n:n + 1, :, y0:y0 + w, x0:x0 + w].cpu().numpy()
                got_d = outs[False][key][n:n + 1, :, y0:y0 + w, x0:x0 + w].cpu().numpy()
                if not np.array_equal(got_s, ref[key]):
                    res["strict_bit_exact"] = False
                    res.setdefault("strict_max_abs_err", 0.0)
                    res["strict_max_abs_err"] = max(res["strict_max_abs_err"], float(np.abs(got_s - ref[key]).max()))
                k = "max_abs_err_" + key
                res[k] = max(res[k], float(np.abs(got_d - ref[key]).max()))
    res["ok"] = bool(res["strict_bit_exact"] and res["max_abs_err_color"] <= PARITY_BAR and res["max_abs_err_depth"] <= PARITY_BAR
                     and res["max_abs_err_T"] <= PARITY_BAR)
    if not want_T:
        res["note"] = "T compared as well (the timed launch does not write it)"
    for k in ("max_abs_err_color", "max_abs_err_depth", "max_abs_err_T"):
        res[k] = float(f"{res[k]:.3e}")
    return res


def pose_sweep(r, rgba, dhw, n_views, vpm, want_T, S, draws, out, status, seed0=100):
    """The timed launch under `draws` seeded draws of the renderer's pose distribution (curriculums.py:109-116 / gmpi.yml:91-96 through
    MPIRenderer.sample_cam_poses): the headline number is ONE draw (seed 3); this is the distribution.  Per draw: 3 event-timed launches
    behind one warm-up, the median counts.  `views_off_band`: views the band kernel handed to the tile kernel (workspace header words)."""
    import statistics
    from ml_gmpi_amd import hip_mpi
    dev = rgba.device
    ms, off_band, n_total = [], 0, 0
    bw = 256 if rgba.dtype == torch.bfloat16 else 128
    n_bands_view = ((S + bw - 1) // bw) * ((S + 7) // 8)
    for d in range(draws):
        torch.manual_seed(seed0 + d)
        cam = r.sample_cam_poses(n_views, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
        ray, eye, zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])

        hints = pose_hints(zd)

        def step():
            r.mpi.render_views(rgba, dhw, ray, eye, zd, views_per_mpi=vpm, check_last_plane=True, out_pm1=True,
                               want_transmittance=want_T, status=status, defer_status=True, out=out, **hints)
        step()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for e0, e1 in evs:
            e0.record(); step(); e1.record()
        torch.cuda.synchronize(dev)
        ms.append(statistics.median(e0.elapsed_time(e1) for e0, e1 in evs))
        ws = hip_mpi.workspace_of(dev)
        if ws is not None and ws.numel() >= 4 * n_bands_view * n_views and r.mpi.variant == "auto":
            hdr = ws[:4 * n_bands_view * n_views].view(torch.int32).view(n_views, n_bands_view)
            off_band += int((hdr != 0).any(dim=1).sum())
            n_total += n_views
    r.mpi.raise_on_status(status)
    q = sorted(ms)
    return dict(draws=draws, seeds=f"{seed0}..{seed0 + draws - 1}", mean_ms=round(sum(ms) / len(ms), 4), p50_ms=round(q[len(q) // 2], 4),
                p90_ms=round(q[min(int(0.9 * len(q)), len(q) - 1)], 4), worst_ms=round(q[-1], 4), best_ms=round(q[0], 4),
                views_off_band_share=None if n_total == 0 else round(off_band / n_total, 4))


class Workload:
    """One render workload set up on a device: synthetic volume(s) resident in HBM, the camera tensors of one pose draw, output buffers, and
    `step()` = ONE pass of the hot path (one `gmpi_mpi_render_launch`) over the batch."""

    def __init__(self, name, dev, rank=0, world=1, variant="auto", strict=False, dry=False, dry_size=64):
        import ml_gmpi_amd
        from ml_gmpi_amd import _lib
        preset, S, D, n_views, dtype, want_T, desc = WORKLOADS[name]
        if dry:
            S, D = dry_size, min(D, 8)
        self.name, self.preset, self.S, self.D, self.n_views, self.dtype, self.want_T, self.desc = name, preset, S, D, n_views, dtype, want_T, desc
        self.variant, self.strict, self.dev = variant, strict, dev
        r = self.r = ml_gmpi_amd.make_renderer(preset, n_planes=D, device=dev, kernel_variant=variant, strict_order=strict,
                                               on_out_of_plane="raise", **({"ray_backend": "torch"} if dry else {}))
        r.set_cam(r.cam_fov, S, S)
        # ---- synthetic inputs, resident in HBM ----
        n_mpis = 1 if name == "cfg4" else n_views
        # config 4 = ONE MPI rendered along a camera path by all ranks: the same volume everywhere; the others: own seeds per rank
        g = torch.Generator(device=dev).manual_seed(1000 * 3 + (0 if name == "cfg4" else rank))
        rgba = torch.empty((n_mpis, D, 4, S, S), device=dev, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
        for i in range(n_mpis):  # per-MPI fill keeps the transient fp32 copy small
            rgba[i] = torch.rand((D, 4, S, S), device=dev, generator=g).to(rgba.dtype)
        rgba[:, -1, 3] = 1.0  # background_alpha_full (networks_cond_on_pos_enc.py:1307-1310)
        self.rgba = rgba
        # camera poses: the same draw from the dataset's pose distribution on every rank (seed 3) -- the volumes differ per rank, the
        # per-GPU work does not (kernel time depends on camera tilt by a few per cent: with per-rank pose seeds the max-over-ranks
        # time of a weak-scaling run would measure pose luck, not scaling)
        torch.manual_seed(3)
        if name == "cfg4":  # video path: yaw sweep, pitch 0 (render_video.py:236-237), this rank's 8 of 8 * world views
            import numpy as np
            yaw = np.linspace(0.5, -0.5, 8 * world)[rank::world]
            cam = r.sample_cam_poses(n_views, 0, 0, 0, 0, False, given_yaws=torch.tensor(yaw, dtype=torch.float32).view(-1, 1),
                                     given_pitches=torch.zeros(n_views, 1))
        else:
            cam = r.sample_cam_poses(n_views, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
        self.ray, self.eye, self.zd = torch.cat(cam[3]), torch.cat(cam[4]), torch.cat(cam[5])
        self.dhw = r._dhw_on_device().expand(n_mpis, -1, -1).contiguous()
        self.vpm = n_views if name == "cfg4" else 1
        self.out = dict(color=torch.empty((n_views, 3, S, S), device=dev), depth=torch.empty((n_views, 1, S, S), device=dev))
        if want_T:
            self.out["T"] = torch.empty((n_views, 1, S, S), device=dev)
        self.status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        self.hints = pose_hints(self.zd)   # (what MPIRenderer.render passes for these poses: GMPI_FLAG_HINT_FRONTAL / _TILTED, advisory)
        self.s_in = 2 if dtype == "bf16" else 4
        self.abytes = algorithmic_bytes(n_views, D, S, self.s_in, want_T)

    def step(self):
        self.r.mpi.render_views(self.rgba, self.dhw, self.ray, self.eye, self.zd, views_per_mpi=self.vpm, check_last_plane=True, out_pm1=True,
                                want_transmittance=self.want_T, status=self.status, defer_status=True, out=self.out, **self.hints)

    def parity(self):
        return parity_block(self.r, self.rgba, self.dhw, self.ray, self.eye, self.zd, self.vpm, self.want_T, self.S)


def companion_lines(a, dev, main_name):
    """Rank 0, N = 1, OUTSIDE the timed region: the configurations that otherwise have no driver-run witness -- config 3 with an fp32 volume (the
    1024^2 x 96 variant whose kernel is memory-side bound), config 2 (BASELINE configs[1]), config 3 in strict-order mode (the bit-identical
    arithmetic the parity tests run) and, since round 6, the per-GPU shards of BASELINE configs[3] (`cfg4_shard`: 8 camera-path views of ONE
    512^2 x 96 MPI, render_video.py:95-130) and configs[4] (`cfg5_shard`: 4 seeds of 1024^2 x 256 with the transmittance output, 17 GB of
    volumes generated in place) and the G-step's render at 1024^2 (`train1024`: forward, zero-fill, backward; train.py:740-779).  Each: its own
    tensors, the same untimed clock ramp as the headline (--prewarm-ms), 5 warm-up + 20 launches with a HIP-event pair around every launch on the
    launch stream, mean of the 20; frac = algorithmic bytes / mean / 8 TB/s, frac_footprint = the texel boxes the views touch / mean / 8 TB/s; and
    the same oracle comparison as the headline's `parity` block.  Not `value`; a witness."""
    res = {}
    for key, name, strict in (("cfg3_f32", "cfg3_f32", False), ("cfg2", "cfg2", False), ("cfg3_strict", "cfg3", True),
                              ("cfg4_shard", "cfg4", False), ("cfg5_shard", "cfg5", False)):
        if name == main_name and not strict:
            continue
        try:
            w = Workload(name, dev, variant=a.variant, strict=strict)
            with torch.no_grad():
                # (the oracle comparison of the block before this one ran on the host for seconds: the GPU's clocks have dropped -- the same
                #  untimed ramp as in front of the headline's warm-up steps)
                t_pre = time.perf_counter()
                while (time.perf_counter() - t_pre) * 1e3 < a.prewarm_ms:
                    for _ in range(8):
                        w.step()
                    torch.cuda.synchronize(dev)
                for _ in range(5):
                    w.step()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
                for e0, e1 in evs:
                    e0.record(); w.step(); e1.record()
                torch.cuda.synchronize(dev)
                w.r.mpi.raise_on_status(w.status)
                ms = [e0.elapsed_time(e1) for e0, e1 in evs]
                mean = sum(ms) / len(ms)
                fbytes = footprint_bytes(w.ray, w.eye, w.dhw, w.S, w.s_in, w.want_T)
                ent = dict(workload=w.desc + (", strict-order arithmetic" if strict else ""), ms=round(mean, 4), min_ms=round(min(ms), 4),
                           launches=len(ms), algorithmic_bytes_per_launch=w.abytes, frac=round(w.abytes / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           footprint_bytes_per_launch=fbytes, frac_footprint=round(fbytes / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           mpix_planes_per_s=round(w.n_views * w.S * w.S * w.D / (mean * 1e-3) / 1e6, 1))
                if not a.no_parity:
                    par = w.parity()
                    ent["parity_ok"] = par["ok"]
                    ent["parity"] = {k: par[k] for k in ("strict_bit_exact", "max_abs_err_color", "max_abs_err_depth", "max_abs_err_T")}
            res[key] = ent
            del w
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001 -- a companion must not cost the headline line
            res[key] = dict(error=f"{type(e).__name__}: {e}"[:300])
    try:
        tw = TrainWorkload("train1024", dev, 0, a.variant)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < a.prewarm_ms:
            tw.step()
            torch.cuda.synchronize(dev)
        parts = tw.parts()
        step_ms = parts["forward_ms"] + parts["grad_zero_fill_ms"] + parts["backward_ms"]
        ent = dict(workload=tw.desc + " (curriculums.py:89-91, train.py:740-779)", forward_ms=parts["forward_ms"], grad_zero_fill_ms=parts["grad_zero_fill_ms"],
                   backward_ms=parts["backward_ms"], backward_gather_ms=parts["backward_gather_ms"], ms=round(step_ms, 4), forward_frac=parts["forward_frac"], backward_frac=parts["backward_frac"],
                   algorithmic_bytes_per_launch=tw.ab_fwd + tw.ab_bwd, frac=round((tw.ab_fwd + tw.ab_bwd) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   what="forward | zero-fill | backward each alone: 20 launches behind 3 warm-ups, HIP events on the launch stream; ms = their sum")
        if not a.no_parity:
            par = tw.parity()
            ent["parity_ok"] = par["ok"]
            ent["parity"] = {k: par[k] for k in ("vs_all_atomic_kernel", "vs_reference_autograd", "bars")}
        res["train1024"] = ent
        del tw
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        res["train1024"] = dict(error=f"{type(e).__name__}: {e}"[:300])
    return res


def parse_smi(out):
    """(engine MHz, package W) out of `rocm-smi --showclocks --showpower`, or None."""
    import re
    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    w = re.search(r"Package Power \(W\): ([\d.]+)", out)
    return (int(m.group(1)), float(w.group(1))) if m and w else None


def power_probe(step, sync, gpu_index, seconds=1.5):
    """Engine clock and package power WHILE the timed launch loops (outside the timed region, rank 0): the render launches of this path run at the
    package power limit (profiles/r05_power.txt), so `ms_per_step` depends on the clock the box's firmware grants at its cap -- this block is the
    witness on the box that produced the line.  Samples `rocm-smi` from a thread while the main thread keeps the launch queue full; None when the tool
    is missing or prints something else."""
    import re
    import shutil
    import subprocess
    import threading
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.isfile(smi):
        return None
    samples, done = [], threading.Event()

    def sampler():
        time.sleep(0.4)   # the clock has settled under the loop by then
        for _ in range(4):
            try:
                out = subprocess.run([smi, "-d", str(gpu_index), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:  # noqa: BLE001
                break
            got = parse_smi(out)
            if got:
                samples.append(got)
        done.set()

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while not done.is_set() and time.perf_counter() - t0 < 20.0:
        for _ in range(16):
            step()
        sync()
    th.join(timeout=15)
    if not samples:
        return None
    cap = None
    try:
        out = subprocess.run([smi, "-d", str(gpu_index), "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", out)
        cap = float(m.group(1)) if m else None
    except Exception:  # noqa: BLE001
        pass
    return {"engine_mhz": [c for c, _ in samples], "package_w": [w for _, w in samples], "cap_w": cap, "engine_mhz_max": 2400,
            "what": "rocm-smi sampled while the timed launch loops (outside the timed region): at the cap the firmware sets the engine clock, and with it ms_per_step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS) + ["video"] + sorted(TRAIN_WORKLOADS))
    ap.add_argument("--variant", default="auto", choices=["auto", "gather", "lds", "wave", "band"])
    ap.add_argument("--strict", action="store_true", help="strict-order arithmetic (bit-identical to the oracle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: CPU tensors, gloo, launches counted instead of executed -- exercises the rank logic only")
    ap.add_argument("--dry-size", type=int, default=64, help="image / texture size of a --dry-run (the volumes live in host memory)")
    ap.add_argument("--numa-pin", action="store_true", help="pin each rank's host threads to the NUMA node of its GPU (mapping printed in the line)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the timed launch (rank 0, outside the timed region)")
    ap.add_argument("--no-companions", action="store_true", help="skip the `companions` block (cfg3_f32 / cfg2 / cfg3 strict, rank 0, N = 1, outside the timed region)")
    ap.add_argument("--pose-draws", type=int, default=32, help="draws of the pose distribution in the `pose_sweep` block (0 = skip)")
    ap.add_argument("--profile-clean", action="store_true",
                    help="for rocprofv3 (tools/prof.sh): the render kernels are launched W + K times and never else -- the clock ramp runs on another "
                         "kernel (the stream probe), no parity / pose sweep / end-to-end / companions / CPU baseline -- so that the average of the "
                         "kernel-trace CSV IS roofline.kernel_ms")
    ap.add_argument("--no-power-probe", action="store_true", help="skip the `power` block (engine clock / package power under the timed launch, rocm-smi, rank 0, outside the timed region)")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed render launches for this long before the W warm-up steps: a GPU coming from idle needs ~0.1 s to reach "
                         "its busy clocks (20 steps are only 23 ms of work); 0 disables")
    a = ap.parse_args()
    if a.profile_clean:
        a.no_parity, a.pose_draws, a.no_cpu_baseline, a.no_companions, a.no_power_probe = True, 0, True, True, True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run (also with one rank: the RCCL path is then exercised)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dry = a.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path; --dry-run exercises the rank logic only)")
    if dry:
        dev = torch.device("cpu")
        if use_dist:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if use_dist:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    pinned = numa_pin(local_rank, a.numa_pin, dry)

    import ml_gmpi_amd
    from ml_gmpi_amd import _lib

    if dry:
        dry_lib = _DryLibrary()
        _lib.load_library = lambda: dry_lib
        a.no_parity, a.pose_draws, a.no_cpu_baseline, a.prewarm_ms, a.no_companions, a.no_power_probe = True, 0, True, 0.0, True, True
    Event = _WallEvent if dry else torch.cuda.Event

    def sync():
        if not dry:
            torch.cuda.synchronize(dev)

    if a.workload == "video":
        assert not dry, "--dry-run covers the render workloads"
        return run_video(a, dev, rank, world, use_dist)
    if a.workload in TRAIN_WORKLOADS:
        assert not dry, "--dry-run covers the render workloads"
        return run_train(a, dev, rank, world, use_dist)
    w = Workload(a.workload, dev, rank, world, a.variant, a.strict, dry, a.dry_size)
    r, rgba, dhw, ray, eye, zd, vpm, out, status = w.r, w.rgba, w.dhw, w.ray, w.eye, w.zd, w.vpm, w.out, w.status
    preset, S, D, n_views, dtype, want_T, desc = w.preset, w.S, w.D, w.n_views, w.dtype, w.want_T, w.desc
    step = w.step
    lib = _lib.load_library()
    vol_bytes = rgba.numel() * rgba.element_size()
    probe_status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)  # (the probes' own words)
    cs = 0 if dry else torch.cuda.current_stream(dev).cuda_stream

    def probe():
        _lib.check(lib.gmpi_stream_probe_launch(rgba.data_ptr(), vol_bytes, probe_status.data_ptr(), cs), "gmpi_stream_probe_launch")

    def fence():
        sync()
        if use_dist:
            dist.barrier()
        sync()

    e2e = None
    with torch.no_grad():
        if a.prewarm_ms > 0:  # clock ramp (untimed, before the W warm-up steps)
            t_pre = time.perf_counter()
            while (time.perf_counter() - t_pre) * 1e3 < a.prewarm_ms:
                for _ in range(8):
                    probe() if a.profile_clean else step()
                sync()
        for _ in range(a.warmup):
            step()
        ev = [(Event(enable_timing=True), Event(enable_timing=True)) for _ in range(a.steps)]
        fence()
        t0 = time.perf_counter()
        for i in range(a.steps):
            ev[i][0].record()  # same stream the kernel is launched on (torch's current stream)
            step()
            ev[i][1].record()
        fence()
        elapsed = time.perf_counter() - t0
        r.mpi.raise_on_status(status)  # asserts of all steps, one read-back
        power = None
        if rank == 0 and not a.no_power_probe:
            with torch.no_grad():
                power = power_probe(step, sync, local_rank)
            r.mpi.raise_on_status(status)
        if not a.profile_clean:
            # end-to-end MPIRenderer.render() as a caller gets it by default: pose sampling on the host + rays + launch + the status read-back
            # in the call (status_mode="sync": the reference's assertion timing)
            r.render(rgba, S, S, views_per_mpi=vpm)  # warm-up of the pose/ray path (first call loads its kernels)
            e2e_t = []
            for _ in range(24):  # every call draws fresh poses: the mean and the worst call are reported (a median would hide the tilted draws)
                sync()
                t1 = time.perf_counter()
                r.render(rgba, S, S, views_per_mpi=vpm)
                sync()
                e2e_t.append((time.perf_counter() - t1) * 1e3)
            # ... and back to back with the LAGGED status check (opt-in, defer_status="lag"): the host runs ahead of the device, one sync at the end
            sync()
            t1 = time.perf_counter()
            for _ in range(24):
                r.render(rgba, S, S, views_per_mpi=vpm, defer_status="lag")
            sync()
            e2e_b2b_ms = (time.perf_counter() - t1) * 1e3 / 24
            ml_gmpi_amd.flush_status()  # (the lagged calls' status bits: looked at now)
            # the same with the poses of the calls drawn ahead of time (MPIRenderer.prefetch_poses): what is left between two
            # launches is the ray kernel, the marshalling of the parameter struct and the status read-back
            r.prefetch_poses(8, n_views)
            r.render(rgba, S, S, views_per_mpi=vpm)
            e2e_pre = []
            for _ in range(7):
                sync()
                t1 = time.perf_counter()
                r.render(rgba, S, S, views_per_mpi=vpm)
                sync()
                e2e_pre.append((time.perf_counter() - t1) * 1e3)
            e2e = dict(mean=sum(e2e_t) / len(e2e_t), max=max(e2e_t), b2b=e2e_b2b_ms, pre=sorted(e2e_pre)[len(e2e_pre) // 2])
            # What `install()` hands the reference's scripts: range_check="full" = the reference's min/max over the WHOLE volume (mpi_renderer.py:447-449,
            # mpi.py:185-187).  The exhaustive pass runs in the first call on a volume and again whenever the volume changed (hip_mpi.MPI._full_check_needed);
            # the reference's video loop renders 100 views of one unchanged MPI.  `first`: a call that runs the pass; `mean`: 24 further calls.
            r_full = ml_gmpi_amd.make_renderer(preset, n_planes=D, device=dev, kernel_variant=a.variant, strict_order=a.strict, on_out_of_plane="raise",
                                               range_check="full")
            r_full.set_cam(r_full.cam_fov, S, S)
            for _ in range(40):                             # (a fresh renderer: its pose look-ahead's first batches stall a call for tens of ms once)
                r_full.render(rgba, S, S, views_per_mpi=vpm)
            full_first = []
            for _ in range(3):
                r_full.mpi._full_check_passed = None        # (as after an in-place update of the volume)
                sync()
                t1 = time.perf_counter()
                r_full.render(rgba, S, S, views_per_mpi=vpm)
                sync()
                full_first.append((time.perf_counter() - t1) * 1e3)
            full_t = []
            for _ in range(24):
                sync()
                t1 = time.perf_counter()
                r_full.render(rgba, S, S, views_per_mpi=vpm)
                sync()
                full_t.append((time.perf_counter() - t1) * 1e3)
            e2e.update(full_first=sorted(full_first)[1], full_mean=sum(full_t) / len(full_t))
            del r_full
        sweep = None
        if a.pose_draws > 0 and a.workload != "cfg4":  # (config 4's poses are a fixed yaw sweep, not a draw)
            sweep = pose_sweep(r, rgba, dhw, n_views, vpm, want_T, S, a.pose_draws, out, status)
        # final gather of the finished frames (the only collective of the job)
        gather_ms = gather_bytes = None
        if use_dist:
            frames = torch.cat([out["color"], out["depth"]] + ([out["T"]] if want_T else []), dim=1)
            buf = torch.empty((world,) + tuple(frames.shape), device=dev)
            dist.all_gather_into_tensor(buf.view(-1, *frames.shape[1:]), frames)   # (first call: communicator set-up, untimed)
            fence()
            tg = time.perf_counter()
            dist.all_gather_into_tensor(buf.view(-1, *frames.shape[1:]), frames)
            fence()
            gather_ms = (time.perf_counter() - tg) * 1e3
            gather_bytes = buf.numel() * buf.element_size()

    kern_ms = sum(s.elapsed_time(e) for s, e in ev) / a.steps
    t = torch.tensor([elapsed, kern_ms], device=dev, dtype=torch.float64)
    per_rank = None
    if use_dist:
        mine = torch.tensor([elapsed / a.steps * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)            # straggler visibility: every rank's own wall time per step
        per_rank = [float(x[0]) for x in allr]
        if a.numa_pin:   # every rank's binding (the line is rank 0's)
            pins = [None] * world
            dist.all_gather_object(pins, pinned)
            pinned = pins
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(t[0]), float(t[1])

    if rank == 0:
        units_per_step = n_views * S * S * D * world  # pixel*planes, all ranks
        value = units_per_step * a.steps / elapsed / 1e6
        s_in = w.s_in
        abytes = w.abytes
        # the roofline is priced on the LARGER of the two clocks (event-timed launches, wall time per step): a launch sequence whose
        # events under-report (queueing, several kernels per step) must not raise the fraction
        step_ms = elapsed / a.steps * 1e3
        roof_ms = max(kern_ms, step_ms)
        achieved = abytes / (roof_ms * 1e-3) / 1e9
        fbytes = footprint_bytes(ray, eye, dhw, S, s_in, want_T)
        # PMC-derived numbers (tools/prof.sh -> profiles/hbm_traffic.json) are quoted only if they were measured on THESE
        # kernel sources (hash of csrc/ + the ABI header) and for this workload / variant; otherwise null
        traffic = valu_floor_ms = None
        traffic_note = "not measured for these sources"
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.isfile(prof):
            try:
                pj = json.load(open(prof))
                ent = pj.get("workloads", {}).get(a.workload)
                if pj.get("source_hash") != _lib.source_hash():
                    traffic_note = f"profiles/hbm_traffic.json is for sources {pj.get('source_hash')}, these are {_lib.source_hash()}"
                elif ent and ent.get("variant", "auto") == a.variant and not a.strict:
                    traffic = ent.get("hbm_bytes_per_launch")
                    traffic_note = ent.get("source", "profiles/hbm_traffic.json")
                    if ent.get("valu_insts_per_launch") and ent.get("valu_ns_per_inst"):
                        # wave64 VALU instructions / 1024 SIMDs x the issue cost of THAT kernel's instruction mix (stored next to the count
                        # by tools/prof.sh: fp32 1.05-1.1 ns, integer / conversions 1.6-1.9 ns per instruction and SIMD with every CU busy at
                        # steady clocks, profiles/r03_probe.txt)
                        valu_floor_ms = round(ent["valu_insts_per_launch"] / 1024 * ent["valu_ns_per_inst"] * 1e-6, 4)
            except Exception as e:
                traffic_note = f"unreadable: {e}"
        # streaming-read ceiling of THIS box, measured in-run: one read-only pass over the same volume with the fastest read pattern of
        # profiles/r03_calibration.txt (gmpi_stream_probe_launch: non-temporal dword loads); the exhaustive range check -- the product's own
        # streaming pass over the volume, install()'s default assertion -- is timed next to it

        def timed(launch):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for e0, e1 in evs:
                e0.record()
                launch()
                e1.record()
            torch.cuda.synchronize(dev)
            return vol_bytes / (min(e0.elapsed_time(e1) for e0, e1 in evs[1:]) * 1e-3) / 1e9

        if dry:
            stream_gbs = range_check_gbs = 1.0
        else:
            stream_gbs = timed(probe)
            range_check_gbs = timed(lambda: _lib.check(lib.gmpi_rgba_range_check_launch(rgba.data_ptr(), {"f32": 0, "bf16": 1}[dtype], rgba.numel(),
                                                                                         probe_status.data_ptr(), cs), "gmpi_rgba_range_check_launch"))
        baseline_cfg = {"cfg4": "BASELINE configs[3]: 64 camera-path views of ONE 512^2 x 96 MPI over 8 GPUs = 8 views per GPU",
                        "cfg5": "BASELINE configs[4]: 32 seeds of 1024^2 x 256 + transmittance over 8 GPUs = 4 seeds per GPU"}.get(a.workload)
        line = {
            "metric": "Mpix*planes/s", "value": round(value, 1), "unit": "Mpix*planes/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "DRY RUN: no GPU, launches counted not executed -- no number of this line means anything" if dry else "synthetic",
            "config": {"workload": desc + (f" ({baseline_cfg}; this job: {n_views * world} views on {world} GPU(s))" if baseline_cfg else ""),
                       "name": a.workload, "views_per_gpu": n_views, "views_total": n_views * world, "H": S, "W": S, "planes": D,
                       "rgba_storage": dtype, "variant": a.variant, "strict_order": a.strict,
                       "outputs": "rgb+depth" + ("+transmittance" if want_T else ""), "parallelism": f"views sharded x{world}",
                       "poses": "cfg4: yaw sweep 0.5..-0.5 split over the ranks" if a.workload == "cfg4" else "truncated-gaussian draw (seed 3) on every rank"},
            "views_per_s": round(n_views * world * a.steps / elapsed, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "priced_on_ms": round(roof_ms, 4), "traffic": traffic, "traffic_source": traffic_note,
                         # companions: against what a pure streaming read reaches on this box, and the VALU issue floor
                         "stream_read_gbs": round(stream_gbs, 1), "frac_of_stream_ceiling": round(achieved / stream_gbs, 4),
                         "range_check_pass_gbs": round(range_check_gbs, 1),
                         "valu_floor_ms": valu_floor_ms,
                         "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": abytes,
                         # conservative companion: only the texel boxes the views actually touch
                         "footprint_bytes_per_launch": fbytes, "frac_footprint": round(fbytes / (roof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         # what a user of the pose distribution sees: the same bytes over the MEAN launch time of the `pose_sweep` draws
                         "frac_pose_mean": None if not sweep else round(abytes / (sweep["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "gather_ms": None if gather_ms is None else round(gather_ms, 3),
            "pose_sweep": sweep, "numa_pin": pinned, "power": power,
        }
        if e2e is not None:
            line.update({"e2e_render_ms": round(e2e["mean"], 3), "e2e_render_max_ms": round(e2e["max"], 3),
                         "e2e_render_back_to_back_lagged_ms": round(e2e["b2b"], 3), "e2e_render_prefetched_poses_ms": round(e2e["pre"], 3),
                         # range_check="full" (install()'s default = the reference's whole-volume assertion): a call that runs the exhaustive pass | the
                         # calls on the unchanged volume that follow (the pass is skipped while the volume is provably unchanged)
                         "e2e_render_install_default_first_ms": round(e2e["full_first"], 3), "e2e_render_install_default_ms": round(e2e["full_mean"], 3)})
        if use_dist:
            # the only collective of the job, and the spread of the ranks (the headline time is the max over ranks)
            full_S = WORKLOADS[a.workload][1]
            line["rccl"] = {"world": dist.get_world_size(), "backend": dist.get_backend(), "gather_ms": round(gather_ms, 3),
                            "gather_bytes": gather_bytes, "gather_gbs": round(gather_bytes / (gather_ms * 1e-3) / 1e9, 2),
                            # (SURVEY 8e's figure for the workload at its full size: world x views x channels x S^2 x 4 B -- equal to gather_bytes except in a --dry-run, whose images are shrunk)
                            "gather_bytes_at_full_size": world * n_views * (4 + (1 if want_T else 0)) * full_S * full_S * 4,
                            "what": "one all_gather_into_tensor of the finished frames [world, views, channels, H, W] fp32, second call (the first sets the communicator up)"}
            line["ms_per_step_ranks"] = {"min": round(min(per_rank), 4), "max": round(max(per_rank), 4), "all": [round(x, 4) for x in per_rank]}
        if dry:
            line["dry_run"] = {"launches_on_rank0": dry_lib.launches, "gathered_shape": None if not use_dist else list(buf.shape)}
        if not a.no_parity:
            line["parity"] = w.parity()
        if not a.no_companions and world == 1 and a.workload == "cfg3" and not a.strict:
            line["companions"] = companion_lines(a, dev, a.workload)
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(preset, S, D, dtype, a.cpu_budget)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
        parity_failed = bool(line.get("parity")) and not line["parity"]["ok"]
        parity_failed = parity_failed or any(c.get("parity_ok") is False for c in (line.get("companions") or {}).values())
    else:
        parity_failed = False
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: the timed launch does not match the oracle (see `parity` in the line above)")


if __name__ == "__main__":
    main()
