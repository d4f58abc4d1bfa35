"""rdf-fusion_amd — MI355X-native BGP-scan + hash-join + FILTER path behind RDF Fusion's
DataFusion surface.  The product is ``lib/librdfgpu.so`` (hand-written HIP for gfx950 + C++
host orchestration behind the C ABI of ``include/rdfgpu.h``); this package is the thin Python
binding used by the tests and the benchmark."""
from . import abi, plan  # noqa: F401
from .engine import (RdfGpuError, GpuQuadStore, GpuPlan, Comm, NTriples, load_library, library_path, kernel_source_sha16,  # noqa: F401
                     choose_index, scan_score, predicate_and, pushdown_to_scan_predicate, shard_of)

__all__ = ["abi", "plan", "RdfGpuError", "GpuQuadStore", "GpuPlan", "Comm", "NTriples", "shard_of", "load_library", "library_path", "kernel_source_sha16",
           "choose_index", "scan_score", "predicate_and", "pushdown_to_scan_predicate"]
