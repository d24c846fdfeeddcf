"""agrep_amd -- MI355X-native approximate record scanner (the agrep k-error hot path).

This package is a thin ctypes view of the C-ABI in include/agrep_hip.h (libagrep_hip.so,
hand-written HIP for gfx950).  Python is used by the tests and by bench.py only; the host
side of the product is C (agrep_amd/host).  There is deliberately no fallback: if the shared
library is missing or no HIP device is usable, everything here raises.
"""
from ._ffi import (AghError, Comm, Match, PatternTables, Query, Result, compile_pattern, corpus_fill_device, device_count, lib,  # noqa: F401
                   shard_cuts_fd,
                   probe_read_ms, set_device, ENGINE_FILTER, ENGINE_FULLSCAN, FORCE_FILTER,
                   FORCE_FULLSCAN, FORCE_NUMBERED, COUNT, FILENAMEONLY, INVERT, NO_BYTES, EMIT_HEAD_DELIM, EMIT_TAIL_DELIM, TIME_SWEEP, TIME_SCAN)

__all__ = ["AghError", "Comm", "PatternTables", "compile_pattern", "shard_cuts_fd", "Match", "Query", "Result", "corpus_fill_device", "device_count", "lib",
           "probe_read_ms", "set_device", "ENGINE_FILTER", "ENGINE_FULLSCAN", "FORCE_FILTER",
           "FORCE_FULLSCAN", "FORCE_NUMBERED", "COUNT", "FILENAMEONLY", "INVERT", "NO_BYTES", "EMIT_HEAD_DELIM", "EMIT_TAIL_DELIM", "TIME_SWEEP", "TIME_SCAN"]
