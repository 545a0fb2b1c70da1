#!/bin/bash
# Build a variant of the WHOLE library with extra -D flags (for switches in the shared headers):
#   bash tests/tools/variant_all.sh <tag> -DCOARSE_SHIFT=18   -> bliss-rs_amd/libblissgpu_<tag>.so
set -e
R=$(cd $(dirname $0)/../.. && pwd); tag=$1; shift
cd $R/bliss-rs_amd/csrc
objs=""
for f in blissgpu scheduler node kernels_pcm kernels_fft512 kernels_tempo kernels_chroma kernels_finalize kernels_pairwise kernels_playlist; do
  extra=""; case $f in kernels_tempo|kernels_finalize|kernels_pairwise|kernels_playlist) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result $extra "$@" -c $f.hip -o /tmp/${f}_$tag.o 2>&1 | grep -v "argument unused" || true &
  objs="$objs /tmp/${f}_$tag.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libblissgpu_$tag.so $objs -ldl -Wl,-rpath,/opt/rocm/lib
echo built libblissgpu_$tag.so
