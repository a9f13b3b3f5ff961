#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --workload $1 --variant $2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2:', d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'])"; }
for wl in cfg3 cfg2 cfg4 cfg5 cfg3_f32; do for v in lds wave; do run $wl $v; done; done
