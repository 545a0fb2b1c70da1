#!/bin/bash
# Build a variant of the library with extra -D flags on kernels_fft512.hip: bash tests/tools/variant512.sh <tag> -DFOO=1 ...
# -> bliss-rs_amd/libblissgpu_<tag>.so (git-ignored; travels to the GPU box; compare with tests/tools/kbench)
set -e
R=$(cd $(dirname $0)/../.. && pwd); tag=$1; shift
cd $R/bliss-rs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result "$@" -c kernels_fft512.hip -o /tmp/k512_$tag.o 2>&1 | grep -v "argument unused" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libblissgpu_$tag.so blissgpu.o scheduler.o node.o kernels_pcm.o /tmp/k512_$tag.o kernels_tempo.o kernels_chroma.o kernels_finalize.o kernels_pairwise.o kernels_playlist.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built libblissgpu_$tag.so
