#!/bin/bash
# usage (from the repo root, on a GPU box): tools/evidence.sh   -- then, back home: tools/collect_evidence.sh r06
# the round's evidence on ONE box: GPU test suite + smoke, profiles (kernel trace + PMC passes), then the bench lines quoting the traffic just measured, then the fuzz runs
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
rocm-smi --showmemorypartition --showcomputepartition 2>/dev/null | grep -E "Partition" > gpurun_out/r06_box.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06_pytest_gpu.txt 2>&1
tools/prof.sh r06 > gpurun_out/r06_prof.log 2>&1
cp gpurun_out/prof_r06/hbm_traffic.json profiles/hbm_traffic.json
tools/bench_lines.sh r06 > gpurun_out/r06_bench_lines.log 2>&1
timeout 900 python tools/fuzz_gpu.py 500 606 > gpurun_out/r06_fuzz_a.txt 2>&1
FUZZ_LARGE=1 timeout 900 python tools/fuzz_gpu.py 120 607 > gpurun_out/r06_fuzz_b.txt 2>&1
timeout 600 python tools/fuzz_backward_gpu.py 600 608 > gpurun_out/r06_fuzz_bwd.txt 2>&1
FUZZ_BWD=gather timeout 600 python tools/fuzz_backward_gpu.py 600 609 > gpurun_out/r06_fuzz_bwd_gather.txt 2>&1
tail -n 4 gpurun_out/r06_pytest_gpu.txt; for f in a b bwd bwd_gather; do tail -n 1 gpurun_out/r06_fuzz_$f.txt; done; tail -15 gpurun_out/r06_bench_lines.log | cut -c1-420
