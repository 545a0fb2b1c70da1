#!/bin/bash
# Build a variant of the library with extra -D flags on the host-side translation units (scheduler / blissgpu / node):
#   bash tests/tools/feed_variant.sh <tag> -DFEED_GROUP_MIB=2048 -DFEED_COPY_STREAMS=1   -> bliss-rs_amd/libblissgpu_<tag>.so
set -e
R=$(cd $(dirname $0)/../.. && pwd); tag=$1; shift
cd $R/bliss-rs_amd/csrc
for f in scheduler blissgpu node; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result "$@" -c $f.hip -o /tmp/${f}_$tag.o 2>&1 | grep -v "argument unused" || true
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libblissgpu_$tag.so /tmp/blissgpu_$tag.o /tmp/scheduler_$tag.o /tmp/node_$tag.o kernels_pcm.o kernels_fft512.o kernels_tempo.o kernels_chroma.o kernels_finalize.o kernels_pairwise.o kernels_playlist.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built libblissgpu_$tag.so
