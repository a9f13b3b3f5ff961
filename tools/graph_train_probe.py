import sys, time, torch
sys.path.insert(0, '.')
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
dev = torch.device('cuda:0')
for name in ("train256", "train512", "train1024"):
    side = torch.cuda.Stream(dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        tw = bench.TrainWorkload(name, dev, 0, "auto")
        def timed(fn, n=30):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                for _ in range(4): fn()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        from ml_gmpi_amd import _lib
        status = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=dev)
        def step2():
            tw.rgba.grad = None
            res = tw.r.mpi.render_views(tw.rgba, tw.dhw, tw.ray, tw.eye, tw.zd, views_per_mpi=1, check_last_plane=True, out_pm1=True, status=status, defer_status=True)
            torch.autograd.backward([res["color"], res["depth"]], [tw.g_color, tw.g_depth])
        host = timed(tw.step)
        eager = timed(step2)
        g_ref = tw.rgba.grad.clone()
        graph = torch.cuda.CUDAGraph()
        tw.rgba.grad = None
        with torch.cuda.graph(graph, stream=side):
            step2()
        rep = timed(graph.replay)
        torch.cuda.synchronize()
        diff = float((tw.rgba.grad - g_ref).abs().max() / g_ref.abs().max())
        parts = tw.parts()
        print(name, "render() under autograd %.4f ms " % host, "render_views eager %.4f ms  graph replay %.4f ms  (parts alone: fwd %.4f + fill %.4f + bwd %.4f = %.4f)  grad diff %.1e" % (eager, rep, parts["forward_ms"], parts["grad_zero_fill_ms"], parts["backward_ms"], parts["forward_ms"] + parts["grad_zero_fill_ms"] + parts["backward_ms"], diff), flush=True)
    del tw, graph
    torch.cuda.empty_cache()
