"""Generates tests/golden/*.npz by RUNNING THE REFERENCE ITSELF on CPU (fp32).

Run in the build container only (needs /root/reference):   python oracle/make_golden.py

The reference has no golden vectors of its own, so these fixtures are the
pin for the oracle (oracle/mpi_oracle.c), for the host-side geometry mirror and for the HIP path:

  render_*.npz   inputs (rgba by seed via oracle.synth_rgba, camera tensors exactly as the
                 reference's `sample_cam_poses` produced them) + outputs of the reference
                 `MPIRenderer.render` / `MPI.forward` (gmpi/core/mpi_renderer.py:387, mpi.py:308)
  backward_*.npz gradient of the reference w.r.t. the RGBA volume (autograd through MPIRenderer.render / MPI.forward,
                 the G-step's backward, train.py:740-779) for seeded output gradients
  geometry.npz   plane depths / dhws / c2w / rays / sampled poses of the reference's host-side
                 helpers (mpi_utils.py:21,787,652; cam_utils.py:734; camera.py:182;
                 mpi_renderer.py:337) for the dataset presets
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle  # noqa: E402
import ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

# name, preset, D, (Ht, Wt), S (render size), B views, align_corners, pose spec, rgba options
RENDER_CASES = [
    dict(name="ffhq_d8_32_ac1", preset="FFHQ", D=8, tex=(32, 32), S=32, B=2, ac=True, pose="random", seed=11),
    dict(name="ffhq_d8_32_ac0", preset="FFHQ", D=8, tex=(32, 32), S=32, B=2, ac=False, pose="random", seed=12),
    dict(name="ffhq_d6_tex48x40_extreme", preset="FFHQ", D=6, tex=(48, 40), S=24, B=2, ac=True,
         pose=[(0.578, 0.254), (-0.578, -0.254)], seed=13),
    dict(name="metfaces_d16_64_bf16", preset="MetFaces", D=16, tex=(64, 64), S=64, B=1, ac=True, pose="random",
         seed=14, bf16=True, last_alpha_one=True),
    dict(name="afhq_d5_tex20_img36", preset="AFHQCat", D=5, tex=(20, 20), S=36, B=1, ac=False, pose="random",
         seed=15),
    dict(name="ffhq_d4_alpha01", preset="FFHQ", D=4, tex=(16, 16), S=16, B=1, ac=True, pose=[(0.1, -0.05)],
         seed=16, alpha_binary=True),
    dict(name="ffhq_d32_256_cfg1", preset="FFHQ", D=32, tex=(256, 256), S=256, B=1, ac=True, pose="random",
         seed=1000),  # BASELINE.json configs[0]
]


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def build_rgba(case):
    Ht, Wt = case["tex"]
    rgba = oracle.synth_rgba(case["seed"], (case["B"], case["D"], 4, Ht, Wt),
                             last_alpha_one=case.get("last_alpha_one", False), bf16_round=case.get("bf16", False))
    if case.get("alpha_binary"):
        rgba[:, :, 3] = (rgba[:, :, 3] > 0.5).astype(np.float32)  # exercises a==1 (the 1e-10 term) and a==0
    return rgba


def run_render_case(ns, case):
    with quiet():
        r = ref_import.make_reference_renderer(ns, case["preset"], case["D"], align_corners=case["ac"])
        r.set_cam(r.cam_fov, case["S"], case["S"])
    B = case["B"]
    torch.manual_seed(case["seed"])
    if case["pose"] == "random":
        cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
    else:
        yaws = torch.tensor([[p[0]] for p in case["pose"]], dtype=torch.float32)
        pitches = torch.tensor([[p[1]] for p in case["pose"]], dtype=torch.float32)
        cam = r.sample_cam_poses(B, 0.0, 0.0, 0.0, 0.0, False, given_yaws=yaws, given_pitches=pitches)
    keys = ["batch_yaws", "batch_pitches", "batch_tf_c2w", "batch_ray_dir", "batch_eye_pos", "batch_z_dir"]
    infos = dict(zip(keys, cam))
    rgba = build_rgba(case)
    t_rgba = torch.from_numpy(rgba)
    if case.get("bf16"):
        t_rgba = t_rgba.to(torch.bfloat16)  # exact (values were rounded); reference upcasts, mpi_renderer.py:446
    with quiet():
        rgb, depth, c2w, angles = r.render(t_rgba, case["S"], case["S"], given_cam_infos=infos,
                                            assert_not_out_of_last_plane=True)
    # second formulation of the same composite (mpi.py:218-306) as an extra pin
    dhw = r.dynamic_mpi_plane_dhws.reshape(1, -1, 3).expand(B, -1, -1)
    with quiet():
        c_old, d_old = r.mpi.old_forward(batch_rgba=t_rgba.float(), batch_dhw=dhw, batch_ray_dir=cam[3],
                                         batch_eye_pos=cam[4], batch_z_dir=cam[5], separate_background=None)
    meta = {k: v for k, v in case.items()}
    np.savez(
        os.path.join(OUT, f"render_{case['name']}.npz"),
        meta=json.dumps(meta),
        dhw=dhw.contiguous().numpy().astype(np.float32),
        ray_dir=torch.cat(cam[3]).numpy(), eye=torch.cat(cam[4]).numpy(), zdir=torch.cat(cam[5]).numpy(),
        yaws=cam[0].numpy(), pitches=cam[1].numpy(), c2w=cam[2].numpy(),
        ref_rgb_pm1=rgb.numpy(), ref_depth=depth.numpy(), ref_angles=angles.numpy(),
        ref_old_color01=c_old.numpy(), ref_old_depth=d_old.numpy(),
    )
    print("wrote", case["name"], "rgb range", float(rgb.min()), float(rgb.max()))


def run_multiview_case(ns):
    """MPI.forward called directly with a ragged views-per-MPI list (mpi.py:331-354): M=2 MPIs, (2,1) views."""
    with quiet():
        r = ref_import.make_reference_renderer(ns, "FFHQ", 5)
        r.set_cam(r.cam_fov, 20, 20)
    torch.manual_seed(21)
    cam = r.sample_cam_poses(3, 0.0, 0.289, 0.0, 0.127, True)
    rgba = oracle.synth_rgba(21, (2, 5, 4, 24, 28))
    dhw = r.dynamic_mpi_plane_dhws.reshape(1, -1, 3).expand(2, -1, -1).contiguous()
    ray = [torch.cat(cam[3][:2]), cam[3][2]]
    eye = [torch.cat(cam[4][:2]), cam[4][2]]
    zd = [torch.cat(cam[5][:2]), cam[5][2]]
    color, depth = r.mpi(batch_rgba=torch.from_numpy(rgba), batch_dhw=dhw, batch_ray_dir=ray, batch_eye_pos=eye,
                         batch_z_dir=zd, separate_background=None, assert_not_out_of_last_plane=True)
    np.savez(os.path.join(OUT, "forward_ragged_views.npz"),
             meta=json.dumps(dict(seed=21, M=2, D=5, tex=(24, 28), S=20, views_per_mpi=[2, 1], ac=True)),
             dhw=dhw.numpy(), ray_dir=torch.cat(cam[3]).numpy(), eye=torch.cat(cam[4]).numpy(),
             zdir=torch.cat(cam[5]).numpy(), view_to_mpi=np.array([0, 0, 1], np.int32),
             ref_color01=color.numpy(), ref_depth=depth.numpy())
    print("wrote forward_ragged_views")


def run_geometry(ns):
    out = {}
    meta = {"presets": {}, "poses": []}
    for preset in ("FFHQ", "MetFaces", "AFHQCat"):
        for confined in (True, False):
            for D in (4, 32, 96):
                if not confined and D != 32:
                    continue
                with quiet():
                    r = ref_import.make_reference_renderer(ns, preset, D, confined=confined)
                key = f"dhw_{preset}_{'conf' if confined else 'free'}_{D}"
                out[key] = r.static_mpi_plane_dhws.numpy()
                meta["presets"][key] = dict(preset=preset, confined=confined, D=D)
    for method in ("uniform", "log-uniform", "sqrt", "squared", "inverse"):
        out[f"dist_{method}"] = ns.mpi_utils.sample_distance(0.95, 1.12, 12, method)
    # c2w + rays for given angles, 3 sphere setups
    with quiet():
        r = ref_import.make_reference_renderer(ns, "FFHQ", 4)
    for i, (S, yaw, pitch) in enumerate([(8, 0.0, 0.0), (8, 0.3, -0.1), (16, -0.578, 0.254), (5, 0.01, 0.2)]):
        with quiet():
            r.set_cam(r.cam_fov, S, S)
        cam = r.sample_cam_poses(1, 0.0, 0.0, 0.0, 0.0, False, given_yaws=torch.tensor([[yaw]]),
                                 given_pitches=torch.tensor([[pitch]]))
        out[f"pose{i}_c2w"] = cam[2].numpy()
        out[f"pose{i}_ray"] = cam[3][0].numpy()
        out[f"pose{i}_eye"] = cam[4][0].numpy()
        out[f"pose{i}_zdir"] = cam[5][0].numpy()
        meta["poses"].append(dict(S=S, yaw=yaw, pitch=pitch))
    with quiet():
        ra = ref_import.make_reference_renderer(ns, "AFHQCat", 4)
        ra.set_cam(ra.cam_fov, 6, 6)
    cam = ra.sample_cam_poses(2, 0.0, 0.0, 0.0, 0.0, False, given_yaws=torch.tensor([[0.2], [-0.4]]),
                              given_pitches=torch.tensor([[0.1], [0.3]]))
    out["afhq_c2w"] = cam[2].numpy()
    out["afhq_ray"] = torch.cat(cam[3]).numpy()
    # RNG-consuming pose sampling (truncated_gaussian / uniform / gaussian; random + deterministic sweeps)
    with quiet():
        r.set_cam(r.cam_fov, 4, 4)
    for j, (method, rnd, B) in enumerate([("truncated_gaussian", True, 5), ("uniform", True, 3),
                                         ("gaussian", True, 3), ("truncated_gaussian", False, 4)]):
        r.cam_sample_method = method
        torch.manual_seed(100 + j)
        cam = r.sample_cam_poses(B, 0.05, 0.289, -0.02, 0.127, rnd)
        out[f"sample{j}_yaws"] = cam[0].numpy()
        out[f"sample{j}_pitches"] = cam[1].numpy()
        out[f"sample{j}_c2w"] = cam[2].numpy()
        out[f"sample{j}_after"] = torch.rand(2).numpy()  # RNG stream position after the call
        meta.setdefault("samples", []).append(dict(method=method, random=rnd, B=B, seed=100 + j))
    # video-style call: std 0 through the truncated_gaussian sampler (render_video.py:236-237 pattern)
    r.cam_sample_method = "truncated_gaussian"
    torch.manual_seed(200)
    cam = r.sample_cam_poses(1, 0.25, 0.0, 0.0, 0.0, True)
    out["video_yaws"] = cam[0].numpy()
    out["video_c2w"] = cam[2].numpy()
    out["video_after"] = torch.rand(2).numpy()
    np.savez(os.path.join(OUT, "geometry.npz"), meta=json.dumps(meta), **out)
    print("wrote geometry", len(out), "arrays")


def run_light_depth():
    """LightRenderer.compute_depth (light_renderer.py:82-100).  The module imports torchvision at import time (not
    installed here, not on this function's path): a stand-in module is registered first."""
    import importlib
    import types
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.transforms = types.SimpleNamespace(GaussianBlur=lambda **k: None)
        sys.modules["torchvision"] = tv
    if ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REFERENCE_ROOT)
    ref_import._install_stubs()
    lr = importlib.import_module("gmpi.core.light_renderer")
    B, D, S = 2, 12, 40
    rgba = oracle.synth_rgba(77, (B, D, 4, S, S))
    rgba[0, :, 3] = (rgba[0, :, 3] > 0.7).astype(np.float32)  # exact 0/1 alphas in one batch element
    alpha = torch.from_numpy(rgba)[:, :, 3:]
    ds = torch.from_numpy(ref_import.import_reference().mpi_utils.sample_distance(0.95, 1.12, D, "inverse"))
    depth = lr.LightRenderer.compute_depth(None, alpha, ds.reshape(-1, 1))
    np.savez(os.path.join(OUT, "light_compute_depth.npz"), meta=json.dumps(dict(seed=77, B=B, D=D, S=S)),
             plane_ds=ds.numpy(), ref_depth=depth.numpy())
    print("wrote light_compute_depth", tuple(depth.shape))


def _torchvision_stand_in():
    """torchvision is not installed in this image.  `LightRenderer.__init__` only needs
    torchvision.transforms.GaussianBlur(kernel_size, sigma); the stand-in restates torchvision's functional
    `gaussian_blur` (transforms/_functional_tensor.py): 1-D kernels = pdf samples on linspace(-(k-1)/2, (k-1)/2, k)
    normalised to 1, 2-D kernel = their outer product, reflect padding by k//2, depthwise conv2d.  Everything else on
    the path (compute_depth, compute_pcl, get_normal, the shading and the pose sampling) is the reference's own code."""
    import types

    class GaussianBlur(torch.nn.Module):
        def __init__(self, kernel_size, sigma):
            super().__init__()
            self.kernel_size, self.sigma = tuple(kernel_size), tuple(sigma)

        @staticmethod
        def _k1d(k, s):
            lim = (k - 1) * 0.5
            x = torch.linspace(-lim, lim, steps=k)
            pdf = torch.exp(-0.5 * (x / s).pow(2))
            return pdf / pdf.sum()

        def forward(self, img):
            kx, ky = self._k1d(self.kernel_size[0], self.sigma[0]), self._k1d(self.kernel_size[1], self.sigma[1])
            k2d = torch.mm(ky[:, None], kx[None, :]).to(img.dtype)
            c = img.shape[-3]
            k2d = k2d.expand(c, 1, k2d.shape[0], k2d.shape[1])
            pad = [self.kernel_size[0] // 2, self.kernel_size[0] // 2, self.kernel_size[1] // 2, self.kernel_size[1] // 2]
            return torch.nn.functional.conv2d(torch.nn.functional.pad(img, pad, mode="reflect"), k2d, groups=c)

    tv = types.ModuleType("torchvision")
    tv.transforms = types.SimpleNamespace(GaussianBlur=GaussianBlur)
    return tv


def run_light_render():
    """LightRenderer.render (light_renderer.py:122-199) on a small MPI: the reference class end to end (seeded light
    pose), with the torchvision stand-in above for the blur."""
    import importlib
    sys.modules["torchvision"] = _torchvision_stand_in()
    sys.modules.pop("gmpi.core.light_renderer", None)
    if ref_import.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REFERENCE_ROOT)
    ns = ref_import.import_reference()
    lr = importlib.import_module("gmpi.core.light_renderer")
    B, D, S = 2, 6, 32
    r = ref_import.make_reference_renderer(ns, "FFHQ", D)
    xyz, _ = r.get_xyz(S, S, ret_single_res=True)  # [D,S,S,3]
    rgba = oracle.synth_rgba(79, (B, D, 4, S, S), last_alpha_one=True)
    # smooth alpha so that the surface (and its normals) are not pure noise
    a = torch.from_numpy(rgba[:, :, 3:4].reshape(B * D, 1, S, S))
    a = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(a, (3, 3, 3, 3), mode="replicate"), 7, stride=1)
    rgba[:, :, 3] = a.reshape(B, D, S, S).numpy()
    rgba[:, -1, 3] = 1.0
    out = {}
    for name, kw, steps in (("kd", dict(ka_max=0.6, kd_max=0.9, n_grow_iters=4), 3), ("ambient", dict(ka_max=1.0, kd_max=0.0, n_grow_iters=2), 2)):
        L = lr.LightRenderer(sphere_center_z=1.0, sphere_r=1.0, **kw)
        torch.manual_seed(123)
        res = None
        for _ in range(steps):  # `step` ramps ka/kd (light_renderer.py:176-179)
            res = L.render(torch.from_numpy(rgba), r.static_mpi_plane_dhws, xyz)
        out[f"ref_{name}"] = res.numpy()
        torch.manual_seed(123)  # the light direction of the LAST step, re-derived with the reference's own sampler
        for _ in range(steps):
            c2w, _, _ = ns.cam_utils.gen_sphere_path(n_cams=B, sphere_center=L.sphere_center, sphere_r=L.sphere_r,
                                                     yaw_mean=L.l_h_mean, yaw_std=L.l_h_std, pitch_mean=L.l_v_mean,
                                                     pitch_std=L.l_v_std, n_truncated_stds=2, flag_rnd=True,
                                                     sample_method="truncated_gaussian", given_yaws=None, given_pitches=None)
        pos = torch.as_tensor(c2w[:, :3, 3], dtype=torch.float32)
        out[f"light_dir_{name}"] = ns.torch_utils.normalize_vecs(torch.FloatTensor(L.sphere_center).reshape(1, 3) - pos).numpy()
        out[f"ka_kd_{name}"] = np.array([L.cur_ka, L.cur_kd], dtype=np.float64)
        out[f"rng_after_{name}"] = torch.rand(2).numpy()
    np.savez(os.path.join(OUT, "light_render.npz"), meta=json.dumps(dict(seed=79, B=B, D=D, S=S)), rgba=rgba,
             dhw=r.static_mpi_plane_dhws.numpy(), xyz=xyz.numpy(), **out)
    print("wrote light_render", {k: v.shape for k, v in out.items()})


def run_backward_cases(ns):
    """Gradient of the reference w.r.t. the RGBA volume: autograd through the reference's own MPIRenderer.render /
    MPI.forward (mpi.py:308-436), exactly what the G-step back-propagates (train.py:740-779), fp32 on the CPU.
    loss = sum(rgb * g_rgb) + sum(depth * g_depth) with seeded normal g's."""
    keys = ["batch_yaws", "batch_pitches", "batch_tf_c2w", "batch_ray_dir", "batch_eye_pos", "batch_z_dir"]
    cases = [
        dict(name="ffhq_d8_32", D=8, S=32, B=2, seed=31, ac=True),
        dict(name="ffhq_d8_32_opaque", D=8, S=32, B=1, seed=32, ac=True, opaque=True),
        dict(name="ffhq_d6_tex40_img24_ac0", D=6, S=24, tex=40, B=2, seed=33, ac=False),
    ]
    for case in cases:
        with quiet():
            r = ref_import.make_reference_renderer(ns, "FFHQ", case["D"], align_corners=case["ac"])
            r.set_cam(r.cam_fov, case["S"], case["S"])
        B, D, S, T = case["B"], case["D"], case["S"], case.get("tex", case["S"])
        torch.manual_seed(case["seed"])
        cam = r.sample_cam_poses(B, r.horizontal_mean, r.horizontal_std, r.vertical_mean, r.vertical_std, True)
        rgba = oracle.synth_rgba(case["seed"], (B, D, 4, T, T), last_alpha_one=True)
        if case.get("opaque"):  # exactly and nearly opaque planes in the middle of the stack (the 1e-10 term matters)
            rgba[:, 2, 3, :, : T // 2] = 1.0
            rgba[:, 4, 3, :, T // 4:] = np.float32(1.0 - 1e-6)
            rgba[:, 3:6, 3, : T // 3, :] = 1.0
        vol = torch.from_numpy(rgba).requires_grad_(True)
        with quiet():
            rgb, depth, _, _ = r.render(vol, S, S, given_cam_infos=dict(zip(keys, cam)))
        g = torch.Generator().manual_seed(case["seed"] + 100)
        g_rgb, g_depth = torch.randn(rgb.shape, generator=g), torch.randn(depth.shape, generator=g)
        ((rgb * g_rgb).sum() + (depth * g_depth).sum()).backward()
        dhw = r.dynamic_mpi_plane_dhws.reshape(1, -1, 3).expand(B, -1, -1)
        np.savez(os.path.join(OUT, f"backward_{case['name']}.npz"), meta=json.dumps(case), rgba=rgba,
                 dhw=dhw.contiguous().numpy().astype(np.float32), ray_dir=torch.cat(cam[3]).numpy(),
                 eye=torch.cat(cam[4]).numpy(), zdir=torch.cat(cam[5]).numpy(), g_rgb=g_rgb.numpy(), g_depth=g_depth.numpy(),
                 ref_rgb_pm1=rgb.detach().numpy(), ref_depth=depth.detach().numpy(), ref_grad_rgba=vol.grad.numpy())
        print("wrote backward", case["name"], "max |grad|", float(vol.grad.abs().max()))
    # MPI.forward called directly with a ragged views-per-MPI list (M = 2 MPIs, (2, 1) views): gradients of two views
    # accumulate into MPI 0
    with quiet():
        r = ref_import.make_reference_renderer(ns, "FFHQ", 5)
        r.set_cam(r.cam_fov, 20, 20)
    torch.manual_seed(34)
    cam = r.sample_cam_poses(3, 0.0, 0.289, 0.0, 0.127, True)
    rgba = oracle.synth_rgba(34, (2, 5, 4, 24, 28))
    vol = torch.from_numpy(rgba).requires_grad_(True)
    dhw = r.dynamic_mpi_plane_dhws.reshape(1, -1, 3).expand(2, -1, -1).contiguous()
    color, depth = r.mpi(batch_rgba=vol, batch_dhw=dhw, batch_ray_dir=[torch.cat(cam[3][:2]), cam[3][2]],
                         batch_eye_pos=[torch.cat(cam[4][:2]), cam[4][2]], batch_z_dir=[torch.cat(cam[5][:2]), cam[5][2]],
                         separate_background=None)
    g = torch.Generator().manual_seed(134)
    g_rgb, g_depth = torch.randn(color.shape, generator=g), torch.randn(depth.shape, generator=g)
    ((color * g_rgb).sum() + (depth * g_depth).sum()).backward()
    np.savez(os.path.join(OUT, "backward_forward_ragged_views.npz"),
             meta=json.dumps(dict(seed=34, M=2, D=5, tex=(24, 28), S=20, views_per_mpi=[2, 1], ac=True)), rgba=rgba,
             dhw=dhw.numpy(), ray_dir=torch.cat(cam[3]).numpy(), eye=torch.cat(cam[4]).numpy(), zdir=torch.cat(cam[5]).numpy(),
             view_to_mpi=np.array([0, 0, 1], np.int32), g_rgb=g_rgb.numpy(), g_depth=g_depth.numpy(),
             ref_color01=color.detach().numpy(), ref_depth=depth.detach().numpy(), ref_grad_rgba=vol.grad.numpy())
    print("wrote backward_forward_ragged_views")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "backward":
        torch.set_num_threads(1)
        run_backward_cases(ref_import.import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "light":
        os.makedirs(OUT, exist_ok=True)
        torch.set_num_threads(1)
        run_light_depth()
        run_light_render()
        return
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)
    ns = ref_import.import_reference()
    for case in RENDER_CASES:
        run_render_case(ns, case)
    run_multiview_case(ns)
    run_backward_cases(ns)
    run_geometry(ns)
    run_light_depth()
    run_light_render()
    with open(os.path.join(OUT, "PROVENANCE.txt"), "w") as f:
        f.write("Generated by oracle/make_golden.py from the reference at /root/reference "
                "(apple/ml-gmpi @ 2024_08_07), CPU fp32, torch %s, numpy %s.\n" % (torch.__version__, np.__version__))


if __name__ == "__main__":
    main()
