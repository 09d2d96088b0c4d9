"""autocycler_amd — MI355X-native hot path of `autocycler compress` (k-mer de Bruijn graph -> unitig GFA).

The compute lives in libautocycler_hip.so (hand-written HIP for gfx950 behind the C ABI declared in
include/autocycler_hip.h); this package is the thin host-side mirror used by the tests and benchmarks."""
from ._capi import (AutocyclerError, Graph, HipLibraryMissing, LIB_PATH, compress_build, graph_from_gfa, load_library)  # noqa: F401

__version__ = "0.1.0"
