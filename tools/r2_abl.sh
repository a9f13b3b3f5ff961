#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --workload $1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1:', d['roofline']['kernel_ms'], d['ms_per_step'], d['e2e_render_ms'], d['e2e_render_prefetched_poses_ms'])"; }
run cfg3; run cfg2
python -X importtime -c "pass" 2>/dev/null
python - <<'PY'
import time, torch, sys
sys.path.insert(0, '.')
import ml_gmpi_amd, cProfile, pstats
dev = torch.device("cuda")
r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=96, device=dev, on_out_of_plane="raise")
rgba = torch.rand((8, 96, 4, 256, 256), device=dev)
with torch.no_grad():
    for _ in range(5): r.render(rgba, 256, 256)
    r.prefetch_poses(300, 8)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): r.render(rgba, 256, 256)
    pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
PY
