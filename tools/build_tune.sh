#!/bin/bash
# Profiling build of the library (experiment knobs from the environment, see DESIGN.md "Profiling builds"):
#   tools/build_tune.sh [extra flags]  ->  ml-gmpi_amd/libgmpi_render_tune.so   (objects under build/tune; never shipped)
set -e
cd "$(dirname "$0")/../ml-gmpi_amd/csrc"
B=../../build/tune; mkdir -p $B
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -DGMPI_TUNE $*"
for s in gmpi_abi render_gather render_lds render_dma render_band render_wave render_backward light_kernels; do
  if [ ! -f $B/$s.o ] || [ $s.hip -nt $B/$s.o ] || [ gmpi_device.hpp -nt $B/$s.o ] || [ -n "$FORCE" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c $s.hip -o $B/$s.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgmpi_render_tune.so $B/*.o
ls -la ../libgmpi_render_tune.so
