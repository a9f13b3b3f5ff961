"""The render launch under HIP graph capture.  `gmpi_mpi_render_launch` is stream-ordered, allocates nothing, never synchronises and keeps no host
state a result depends on (include/gmpi_render.h): a caller may record it into a graph once and replay it over buffers whose CONTENT changes -- new
poses, a new volume -- which is how a launch-bound loop (small images, a camera path) sheds its per-call host cost on this part.  What a replay
re-uses and a fresh launch would not: the per-launch generation number of AUTO's view gate (gmpi_device.hpp `KParams::gate`); a gate word left by an
earlier replay can only move a view from the band kernel to the tile kernel, never leave it unrendered or rendered twice -- checked here with pose
sets that flip views between the two kernels from replay to replay.  Strict-order mode: bit-identical to an uncaptured launch and to the oracle."""
import numpy as np
import pytest
import torch

import oracle
from test_hip_parity import TOL, _lib, _random_case

pytestmark = pytest.mark.gpu


def _poses(S, B, D, seeds, tilted):
    sets = []
    for seed, tilt in zip(seeds, tilted):
        _, dhw, ray, eye, zd = _random_case(seed=seed, B=B, D=D, S=S)
        if tilt:   # views 1 and 3 from the edge of the pose distribution: boxes the band kernel cannot stage -> AUTO's gate
            _, _, ray_x, eye_x, zd_x = _random_case(seed=seed + 100, B=B, D=D, S=S, extreme=True)
            for n in tilt:
                ray[n], eye[n], zd[n] = ray_x[n], eye_x[n], zd_x[n]
        sets.append((dhw, ray, eye, zd))
    return sets


@pytest.mark.parametrize("dtype,strict", [(torch.bfloat16, True), (torch.bfloat16, False), (torch.float32, True)])
def test_render_launch_replays_from_a_graph(dtype, strict):
    from ml_gmpi_amd import MPI
    lib = _lib().load_library()
    dev = torch.device("cuda:0")
    S, B, D = 512, 4, 6
    bw = 128 if dtype == torch.float32 else 256
    assert B * ((S + bw - 1) // bw) * ((S + 7) // 8) >= lib.gmpi_query(10 if dtype == torch.float32 else 9)   # AUTO takes the band kernel + the gated tile launch
    rgba0, _, _, _, _ = _random_case(seed=71, B=B, D=D, S=S)
    rgba1, _, _, _, _ = _random_case(seed=72, B=B, D=D, S=S)
    sets = _poses(S, B, D, seeds=(73, 74, 75), tilted=((), (1, 3), (0,)))
    mpi = MPI(align_corners=True, variant="auto", strict_order=strict, range_check="touched", on_out_of_plane="raise")
    # the buffers the graph is recorded over
    vol = rgba0.to(dev).to(dtype)
    dhw, ray, eye, zd = (t.to(dev).clone() for t in sets[0])
    out = dict(color=torch.empty((B, 3, S, S), device=dev), depth=torch.empty((B, 1, S, S), device=dev), T=torch.empty((B, 1, S, S), device=dev))
    status = torch.zeros(_lib().STATUS_WORDS, dtype=torch.int32, device=dev)

    def launch():
        mpi.render_views(vol, dhw, ray, eye, zd, views_per_mpi=1, check_last_plane=True, want_transmittance=True, status=status, defer_status=True, out=out)

    def load(volume, pose):
        vol.copy_(volume.to(dev).to(dtype))
        for dst, src in zip((dhw, ray, eye, zd), pose):
            dst.copy_(src.to(dev))

    def result():
        torch.cuda.synchronize()
        assert int(status[0].item()) == 0
        return {k: v.cpu().numpy().copy() for k, v in out.items()}

    with torch.no_grad():
        # uncaptured launches: what every (volume, pose set) pair renders to
        want = {}
        for vi, volume in enumerate((rgba0, rgba1)):
            for pi, pose in enumerate(sets):
                load(volume, pose)
                launch()
                want[vi, pi] = result()
        # record ONE launch, then replay it over changing content -- the order visits fit -> gated -> fit -> other gated views -> fit
        load(rgba0, sets[0])
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            launch()   # (warm-up on the capture stream: its workspace exists before the recording starts)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            launch()
        for vi, pi in ((0, 0), (0, 1), (1, 0), (1, 2), (0, 1), (1, 1), (0, 0)):
            load((rgba0, rgba1)[vi], sets[pi])
            for v in out.values():
                v.fill_(float("nan"))
            graph.replay()
            got = result()
            for k in ("color", "depth", "T"):
                assert not np.isnan(got[k]).any(), (vi, pi, k)
                if strict:
                    assert np.array_equal(got[k], want[vi, pi][k]), (vi, pi, k, np.abs(got[k] - want[vi, pi][k]).max())
                else:   # (a view a stale gate word sends to the tile kernel: the other kernel's default-mode arithmetic, inside the same bar)
                    assert np.abs(got[k] - want[vi, pi][k]).max() <= TOL, (vi, pi, k)
        if strict:   # ... and the last replay against the oracle
            orc = oracle.render(rgba0.to(dtype).float(), *sets[0], threads=True)
            for k in ("color", "depth", "T"):
                assert np.array_equal(got[k], orc[k]), k


def test_forward_and_backward_replay_from_a_graph():
    """The G-step's render (train.py:740-779: forward, then d/d rgba through the autograd bridge) recorded as ONE graph and replayed over a new volume:
    colour bit-identical to the uncaptured step, the gradient to the atomics' order (1e-6 of its largest element).  Every step, the first included, runs on
    the capture stream: a leaf whose first backward ran on another stream makes torch's engine tie the two streams together, which ends a capture."""
    from ml_gmpi_amd import MPI
    dev = torch.device("cuda:0")
    S, B, D = 256, 4, 8
    rgba0, dhw, ray, eye, zd = _random_case(seed=81, B=B, D=D, S=S)
    rgba1 = _random_case(seed=82, B=B, D=D, S=S)[0]
    mpi = MPI(align_corners=True, variant="auto", range_check="touched", on_out_of_plane="raise")
    dhw, ray, eye, zd = (t.to(dev) for t in (dhw, ray, eye, zd))
    g = torch.Generator(device=dev).manual_seed(83)
    gc, gd = torch.randn((B, 3, S, S), device=dev, generator=g), torch.randn((B, 1, S, S), device=dev, generator=g)
    status = torch.zeros(_lib().STATUS_WORDS, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        vol = rgba0.to(dev).clone().requires_grad_(True)

        def step():
            res = mpi.render_views(vol, dhw, ray, eye, zd, views_per_mpi=1, check_last_plane=True, status=status, defer_status=True)
            loss = (res["color"] * gc).sum() + (res["depth"] * gd).sum()
            grad, = torch.autograd.grad(loss, vol)
            return res["color"], grad

        want = {}
        for vi, volume in enumerate((rgba0, rgba1)):
            with torch.no_grad():
                vol.copy_(volume.to(dev))
            c, gr = step()
            torch.cuda.synchronize()
            want[vi] = (c.clone(), gr.clone())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            c_out, g_out = step()
        for vi in (0, 1, 0):
            with torch.no_grad():
                vol.copy_((rgba0, rgba1)[vi].to(dev))
            graph.replay()
            torch.cuda.synchronize()
            assert int(status[0].item()) == 0
            assert torch.equal(c_out, want[vi][0]), vi
            assert float((g_out - want[vi][1]).abs().max() / want[vi][1].abs().max()) <= 1e-6, vi
