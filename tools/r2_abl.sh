#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/r2_pose.py bf16 wave
run() { timeout 300 python bench.py --workload $1 --variant $2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2:', d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'])"; }
run cfg3 wave; run cfg2 wave
