// The neighbouring rows of SURVEY.md §8 (f): device end repair and pairwise contig distances (neighbours.inc), the round-trip verifier
// and device decompress (kernels_verify.inc).
#include "graph_impl.hpp"

namespace ac {

#include "neighbours.inc"      // device end repair (f-1) and pairwise contig distances (f-3)
#include "kernels_verify.inc"  // ac_verify_graph: the round-trip verifier at scale (f-4)

}  // namespace ac
