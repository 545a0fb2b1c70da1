#!/bin/bash
# Build a variant of the library from another chroma translation unit and / or with extra -D flags on it:
#   bash tests/tools/variant.sh <tag> [--tu <file.hip>] [-DFOO=1 ...]
# -> bliss-rs_amd/libblissgpu_<tag>.so (git-ignored; travels to the GPU box; compare with tests/tools/kbench).
# Probe translation units (they include bliss-rs_amd/csrc/kernels_chroma.hip, never the other way round):
#   tests/tools/probes/stft_trace/stft_trace.hip           per-phase cycle table of the FFT-8192 kernel, printed by kbench
#   tests/tools/probes/handpipe/kernels_chroma_handpipe.hip   the withdrawn hand-pipelined contraction
set -e
R=$(cd $(dirname $0)/../.. && pwd); tag=$1; shift
tu=$R/bliss-rs_amd/csrc/kernels_chroma.hip
if [ "$1" = "--tu" ]; then tu=$(cd $R && realpath $2); shift 2; fi
cd $R/bliss-rs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result -I$R/bliss-rs_amd/csrc "$@" -c $tu -o /tmp/kc_$tag.o 2>&1 | grep -v "argument unused" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libblissgpu_$tag.so blissgpu.o scheduler.o node.o kernels_pcm.o kernels_fft512.o kernels_tempo.o /tmp/kc_$tag.o kernels_finalize.o kernels_pairwise.o kernels_playlist.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built libblissgpu_$tag.so
