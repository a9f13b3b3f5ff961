#!/bin/bash
# first GPU contact of the wave kernel: parity subset, then A/B timings
set -u
mkdir -p gpurun_out/r2a
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_edge_cases.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/pytest.log
for wl in cfg3 cfg3_f32; do
  for v in lds wave; do
    echo "== $wl $v" ; timeout 200 python bench.py --workload $wl --variant $v --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
  for t in 1 2 4; do
    echo "== $wl wave tune $t"; GMPI_TUNE_WAVE=$t timeout 200 python bench.py --workload $wl --variant wave --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done 2>&1 | tee gpurun_out/r2a/ab.log
