#!/bin/bash
# tools/bench_lines.sh <tag>  -- run on the GPU box: bench lines of all workloads + the default command (with cpu_baseline) + the one-rank
# torchrun line -> gpurun_out/<tag>_bench_lines.jsonl, <tag>_default_bench.json, <tag>_torchrun1_bench.json
# (profiles/hbm_traffic.json must match the sources for `roofline.traffic` to be quoted: run tools/prof.sh first and copy it).
TAG=${1:-r04}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
: > gpurun_out/${TAG}_bench_lines.jsonl
for wl in cfg3 cfg2 cfg3_f32 cfg4 cfg5; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-companions 2>/dev/null | grep "^{" | tail -1 >> gpurun_out/${TAG}_bench_lines.jsonl
done
timeout 300 python bench.py --strict --no-cpu-baseline --no-companions --pose-draws 0 2>/dev/null | grep "^{" | tail -1 >> gpurun_out/${TAG}_bench_lines.jsonl
: > gpurun_out/${TAG}_train_lines.jsonl
for wl in train256 train512 train1024; do
  timeout 300 python bench.py --workload $wl 2>/dev/null | grep "^{" | tail -1 >> gpurun_out/${TAG}_train_lines.jsonl
done
timeout 400 python bench.py 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${TAG}_default_bench.json
timeout 300 python bench.py --workload video --steps 5 --warmup 2 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${TAG}_video_bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-companions 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${TAG}_torchrun1_bench.json
python - $TAG <<'PY'
import json, sys
tag = sys.argv[1]
for f in [f'gpurun_out/{tag}_bench_lines.jsonl', f'gpurun_out/{tag}_default_bench.json', f'gpurun_out/{tag}_torchrun1_bench.json']:
    for l in open(f):
        if not l.strip(): continue
        d = json.loads(l); r = d['roofline']; ps = d.get('pose_sweep') or {}; pa = d.get('parity') or {}
        print(d['config']['name'], 'strict' if d['config'].get('strict_order') else '', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_pose_mean'), r['frac_of_stream_ceiling'], r['traffic'],
              d.get('e2e_render_ms'), d.get('e2e_render_back_to_back_lagged_ms'), d['gather_ms'], (d.get('cpu_baseline') or {}).get('value'),
              'sweep', ps.get('mean_ms'), ps.get('p90_ms'), ps.get('worst_ms'), ps.get('views_off_band_share'), 'parity', pa.get('ok'), pa.get('max_abs_err_color'),
              'companions', {k: (v.get('ms'), v.get('frac'), v.get('parity_ok')) for k, v in (d.get('companions') or {}).items()}, 'rccl', d.get('rccl'))
for l in open(f'gpurun_out/{tag}_train_lines.jsonl'):
    if l.strip():
        d = json.loads(l); print(d['config']['name'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['parts'])
PY
