import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml_gmpi_amd
dev = torch.device('cuda')
S, D, B = 1024, 96, 4
r = ml_gmpi_amd.make_renderer('FFHQ', n_planes=D, device=dev, on_out_of_plane='raise')
rgba = torch.rand((B, D, 4, S, S), device=dev).to(torch.bfloat16); rgba[:, -1, 3] = 1
torch.manual_seed(3)
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = r.render(rgba, S, S)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(i, 'render %.2f ms' % ((t1 - t0) * 1e3), 'yaw', [round(float(a), 2) for a in out[3][:, 1]], 'pitch', [round(float(a), 2) for a in out[3][:, 0]])
