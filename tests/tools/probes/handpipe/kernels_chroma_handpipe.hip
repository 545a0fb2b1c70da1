// Probe translation unit: the library's kernels_chroma.hip with the WITHDRAWN hand-pipelined contraction of round 2 in place
// of chroma_kernel (see chroma_handpipe.inc).  Built by `tests/tools/variant.sh hp --tu tests/tools/probes/handpipe/kernels_chroma_handpipe.hip`
// into libblissgpu_hp.so; never part of libblissgpu.so.
#define BG_PROBE_REPLACES_CHROMA_KERNEL
#include "../../../../bliss-rs_amd/csrc/kernels_chroma.hip"
namespace bg {
#include "chroma_handpipe.inc"
}  // namespace bg
