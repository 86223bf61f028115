"""cammiq_amd -- MI355X-native engine for CAMMiQ's read-classification hot path.

The product is ``libcammiq_hip.so`` (C ABI in ``include/cammiq_hip.h``, sources in
``cammiq_amd/csrc``) and the ``cammiq`` command-line shell built on it.  This package is
the thin Python binding used by tests and bench.py, plus synthetic-input helpers.
"""
from .binding import (CammiqError, Comm, Index, MODE_P, MODE_SC, Multi, comm_unique_id,  # noqa: F401
                      host_array, lib, lib_path, pack_reads, pack_reads_tight, shard_range, stride_bytes, stride_words)
