#!/bin/bash
# bench lines of all workloads + the default command + the one-rank torchrun line (profiles/hbm_traffic.json must match the sources)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
: > gpurun_out/r02_bench_lines.jsonl
for wl in cfg3 cfg2 cfg3_f32 cfg4 cfg5; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 >> gpurun_out/r02_bench_lines.jsonl
done
timeout 300 python bench.py 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_default_bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_torchrun1_bench.json
python tools/r2_pose.py bf16 lds > gpurun_out/r02_pose_sweep.txt; python tools/r2_pose.py bf16 wave >> gpurun_out/r02_pose_sweep.txt; python tools/r2_pose.py f32 lds >> gpurun_out/r02_pose_sweep.txt
python - <<'PY'
import json
for f in ['gpurun_out/r02_bench_lines.jsonl', 'gpurun_out/r02_default_bench.json', 'gpurun_out/r02_torchrun1_bench.json']:
    for l in open(f):
        if not l.strip(): continue
        d = json.loads(l); r = d['roofline']
        print(d['config']['name'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['frac_of_stream_ceiling'], r['traffic'], r['valu_floor_ms'], d['e2e_render_ms'], d['e2e_render_prefetched_poses_ms'], d['gather_ms'], (d.get('cpu_baseline') or {}).get('value'))
PY
