#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/r2_small.py 2>&1 | sed 's/lds/[forced lds]/'
run() { timeout 300 python bench.py --workload $1 --variant $2 --steps 50 --warmup 10 --no-cpu-baseline 2>&1 | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2:', d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['e2e_render_ms'])"; }
run cfg2 auto; run cfg3 auto; run cfg4 auto
