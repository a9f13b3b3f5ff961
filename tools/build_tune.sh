#!/bin/bash
# Profiling build of the library (experiment knobs from the environment, see DESIGN.md "Profiling builds"):
#   tools/build_tune.sh [extra flags]  ->  ml-gmpi_amd/libgmpi_render_tune.so   (objects under build/tune; never shipped)
#   SUFFIX=_x ONLY="render_band" tools/build_tune.sh -DFOO  ->  ml-gmpi_amd/libgmpi_render_tune_x.so: an A/B build next to the first one
#       (objects under build/tune_x; only the sources named in ONLY are compiled with the extra flags, the others are taken from build/tune)
#   NOTUNE=1: without -DGMPI_TUNE (the product's code paths, e.g. with -DGMPI_PROF)
set -e
cd "$(dirname "$0")/../ml-gmpi_amd/csrc"
B=../../build/tune$SUFFIX; mkdir -p $B
TUNE=-DGMPI_TUNE; [ -n "$NOTUNE" ] && TUNE=
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function $TUNE $*"
ALL="gmpi_abi render_gather render_lds render_band render_wave render_backward render_backward_gather light_kernels"
for s in $ALL; do
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $s "; then cp ../../build/tune/$s.o $B/$s.o; continue; fi
  if [ ! -f $B/$s.o ] || [ $s.hip -nt $B/$s.o ] || [ gmpi_device.hpp -nt $B/$s.o ] || [ gmpi_backward.hpp -nt $B/$s.o ] || [ ../../include/gmpi_render.h -nt $B/$s.o ] || [ -n "$FORCE" ] || [ -n "$ONLY" ]; then
    rm -f $B/$s.o; /opt/rocm/bin/hipcc $FLAGS -c $s.hip -o $B/$s.o &
  fi
done
wait
for s in $ALL; do [ -f $B/$s.o ] || { echo "build_tune: $s failed to compile"; rm -f ../libgmpi_render_tune$SUFFIX.so; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgmpi_render_tune$SUFFIX.so $B/*.o
ls -la ../libgmpi_render_tune$SUFFIX.so
