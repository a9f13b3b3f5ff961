#!/bin/bash
# Copies what tools/evidence.sh left under gpurun_out/ into profiles/ (tracked): tools/collect_evidence.sh r06
# (refuses when the traffic file was not measured on the sources of this tree)
set -e
cd "$(dirname "$0")/.."
T=${1:-r06}
H=$(python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('ml-gmpi_amd._lib').source_hash())")
grep -q "\"$H\"" gpurun_out/prof_$T/hbm_traffic.json || { echo "gpurun_out/prof_$T/hbm_traffic.json is not of sources $H"; exit 1; }
cp gpurun_out/prof_$T/summary.txt profiles/${T}_prof_summary.txt
cp gpurun_out/prof_$T/hbm_traffic.json profiles/hbm_traffic.json
cp gpurun_out/prof_$T/trace/t_kernel_stats.csv profiles/${T}_cfg3_kernel_stats.csv
cp gpurun_out/prof_$T/train1024/trace/t_kernel_stats.csv profiles/${T}_train1024_kernel_stats.csv
for f in bench_lines.jsonl default_bench.json train_lines.jsonl video_bench.json torchrun1_bench.json; do cp gpurun_out/${T}_$f profiles/${T}_$f; done
{
  echo "Round ${T#r0}, randomized parity runs on the final sources ($H), one MI355X (the evidence box of profiles/${T}_*; GPU tests on the same box: $(grep -oE '[0-9]+ (passed|failed)[^=]*' gpurun_out/${T}_pytest_gpu.txt | head -1), smoke ok):"
  echo "tools/fuzz_gpu.py 500 606:"; tail -n 1 gpurun_out/${T}_fuzz_a.txt
  echo "FUZZ_LARGE=1 tools/fuzz_gpu.py 120 607 (launches large enough for AUTO's band path and its view sharing):"; tail -n 1 gpurun_out/${T}_fuzz_b.txt
  echo "tools/fuzz_backward_gpu.py 600 608 (the atomic tile kernel, 64 x 8 tiles, against the one-pixel-per-lane kernel):"; tail -n 1 gpurun_out/${T}_fuzz_bwd.txt
  echo "FUZZ_BWD=gather tools/fuzz_backward_gpu.py 600 609 (the atomics-free pair against the one-pixel-per-lane kernel):"; tail -n 1 gpurun_out/${T}_fuzz_bwd_gather.txt
  echo "earlier in the round, other boxes and earlier sources: the same four commands on 9e2df0a9e673 and 5877118c7317 (all ok; backward worst 4.89e-06 / 5.55e-06 and 4.87e-06 / 7.12e-06); gather pair 400 cases worst 9.66e-06, 300 cases 2.04e-06; tile kernel 300-400 cases each on five builds, worst 8.0e-06 (gpurun_out/r6*/fuzz_backward*.txt)"
  grep -E '^(FUZZ_SHARED_CASES|Extended runs)' profiles/${T}_fuzz.txt 2>/dev/null || true   # (lines added by hand: runs outside the evidence call)
} > profiles/${T}_fuzz.txt.new && mv profiles/${T}_fuzz.txt.new profiles/${T}_fuzz.txt
python tools/status_table.py $T
