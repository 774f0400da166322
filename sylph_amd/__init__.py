"""sylph_amd — MI355X-native sketch + profile engine (hot path of bluenote-1577/sylph behind a C ABI).

The product is the shared library `libsylph_hip.so` (hand-written gfx950 HIP kernels, include/sylph_hip.h).  This
package is the thin Python binding used by tests/ and bench.py; it fails loudly when the library is missing — there
is no CPU fallback.
"""
from .binding import (Comm, Context, Database, FastqText, Inflated, PinnedBuffer, Pipeline, ReadSketcher, SylphHipError, SEED_AVX2_COMPAT, SEED_SCALAR, READS_PAIRED,
                      READS_SINGLE, lib_path, load, pack_2bit, shard_bounds)

__all__ = ["Pipeline", "pack_2bit", "Comm", "shard_bounds", "Context", "Database", "FastqText", "Inflated", "PinnedBuffer", "ReadSketcher", "SylphHipError", "SEED_AVX2_COMPAT", "SEED_SCALAR", "READS_PAIRED",
           "READS_SINGLE", "lib_path", "load"]
