#!/bin/bash
# final evidence of the round: profiles (tools/prof.sh), then bench lines of all workloads and of the default command
cd $GRAFT_REPO_ROOT
bash tools/prof.sh r02 > gpurun_out/prof_r02.log 2>&1
tail -3 gpurun_out/prof_r02.log
cp gpurun_out/prof_r02/hbm_traffic.json profiles/hbm_traffic.json   # so that the bench lines below quote the traffic of THESE sources
sed -i 's#profiles/prof_r02_summary.txt#profiles/r02_prof_summary.txt#' profiles/hbm_traffic.json
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic_r02.json
: > gpurun_out/r02_bench_lines.jsonl
for wl in cfg3 cfg2 cfg3_f32 cfg4 cfg5; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 >> gpurun_out/r02_bench_lines.jsonl
done
timeout 300 python bench.py 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_default_bench.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_torchrun1_bench.json
python tools/r2_pose.py bf16 lds > gpurun_out/r02_pose_sweep.txt; python tools/r2_pose.py bf16 wave >> gpurun_out/r02_pose_sweep.txt; python tools/r2_pose.py f32 lds >> gpurun_out/r02_pose_sweep.txt
python - <<'PY'
import json
for l in open('gpurun_out/r02_bench_lines.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(d['config']['name'], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['frac_of_stream_ceiling'], r['traffic'], r['valu_floor_ms'], d['e2e_render_ms'], d['e2e_render_prefetched_poses_ms'])
print(open('gpurun_out/r02_torchrun1_bench.json').read()[-260:])
PY
