#!/bin/bash
# Builds variants of ONE translation unit into ab_libs/lib_<tag>.so (the other objects are linked as they are):
#   tools/build_variants.sh kernels_scan "base=" "prio3=-DLR_EPI_PRIO=3" ...
# (cross-compiles here; tools/ab_many.sh then times them in turn on the GPU box)
set -e
cd "$(dirname "$0")/../bulklmm.jl_amd/csrc"
UNIT=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -ffp-contract=on"
make -j8 > /dev/null
OTHERS=$(ls *.o | grep -v "^$UNIT.o$")
mkdir -p ../../ab_libs
pids=()
for spec in "$@"; do
  tag=${spec%%=*}; ex=${spec#*=}
  ( /opt/rocm/bin/hipcc $FLAGS $ex -c $UNIT.hip -o /tmp/${UNIT}_$tag.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab_libs/lib_$tag.so /tmp/${UNIT}_$tag.o $OTHERS -ldl -lpthread && echo "built $tag" ) &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
