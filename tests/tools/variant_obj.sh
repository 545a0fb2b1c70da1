#!/bin/bash
# Build a variant of the library in which ONE translation unit is compiled with extra -D flags:
#   bash tests/tools/variant_obj.sh <tag> <unit> [-DFOO=1 ...]     unit = kernels_pairwise, kernels_fft512, scheduler, ...
# -> bliss-rs_amd/libblissgpu_<tag>.so (git-ignored; travels to the GPU box; compare with tests/tools/kbench)
set -e
R=$(cd $(dirname $0)/../.. && pwd); tag=$1; unit=$2; shift 2
cd $R/bliss-rs_amd/csrc
extra=""; case $unit in kernels_tempo|kernels_finalize|kernels_pairwise|kernels_playlist) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result $extra "$@" -c $unit.hip -o /tmp/${unit}_$tag.o 2>&1 | grep -v "argument unused" || true
objs=""
for f in blissgpu scheduler node kernels_pcm kernels_fft512 kernels_tempo kernels_chroma kernels_finalize kernels_pairwise kernels_playlist; do
  if [ $f = $unit ]; then objs="$objs /tmp/${unit}_$tag.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libblissgpu_$tag.so $objs -ldl -Wl,-rpath,/opt/rocm/lib
echo built libblissgpu_$tag.so
