#!/bin/bash
# memory-path counters of the wave kernel (cfg3 bf16 and fp32), SYNC vs no-SYNC
cd $GRAFT_REPO_ROOT
for wl in cfg3 cfg3_f32; do
for t in 0 1; do
  for pass in tcp ea ta tlb; do
    GMPI_TUNE_WAVE=$t bash tools/prof_mem.sh r2_${wl}_t$t $pass --workload $wl --variant wave 2>&1 | grep -v "^$" | head -12
  done
done
done
