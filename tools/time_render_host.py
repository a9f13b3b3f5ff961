"""Host-side cost of MPIRenderer.render() per call (cProfile over 300 calls at config 2 size: the kernel is short, the host work shows).
    python tools/time_render_host.py [S] [D] [views] [dtype]"""
import cProfile, pstats, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml_gmpi_amd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = int(sys.argv[2]) if len(sys.argv) > 2 else 96
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dt = torch.bfloat16 if (len(sys.argv) > 4 and sys.argv[4] == "bf16") else torch.float32
dev = torch.device("cuda:0")
r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise")
r.set_cam(r.cam_fov, S, S)
rgba = torch.rand((B, D, 4, S, S), device=dev).to(dt)
rgba[:, -1, 3] = 1.0
with torch.no_grad():
    for _ in range(50):
        r.render(rgba, S, S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        r.render(rgba, S, S)
    torch.cuda.synchronize()
    print(f"render() back to back: {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per call")
    e = []
    for _ in range(50):
        torch.cuda.synchronize(); t1 = time.perf_counter(); r.render(rgba, S, S); torch.cuda.synchronize(); e.append((time.perf_counter() - t1) * 1e3)
    print(f"render() + sync, median: {sorted(e)[25]:.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        r.render(rgba, S, S)
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
