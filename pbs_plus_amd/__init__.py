"""pbs_plus_amd — MI355X-native content-defined chunker + SHA-256 dedup-hash engine.

Host-side mirror (Python over the C ABI in include/pbsgpu.h) of the interface the
pbs-plus reference uses for its pxar stream path:

* ``buzhash.NewConfig`` / ``buzhash.Config`` — github.com/pbs-plus/pxar ``buzhash``
  (reference internal/pxarmount/commit_orchestrate.go:143-149, internal/tapeio/converter.go:248)
* ``Engine`` — batch cut + digest (the chunk loop behind ``WriteEntryReader``)
* ``PayloadStream`` — the payload-stream writer seam (``transfer.ArchiveWriter``)
* ``PageRing`` — many streams, page-granular memory release, persistent SHA-256 service
* ``Chunker`` — upstream-style ``scan`` compatibility
* ``didx`` / ``dedup`` / ``Comm`` — dynamic index records and the cross-GPU digest-set reduce (RCCL, behind the C ABI)

Everything executes in the gfx950 kernels of ``lib/libpbsgpu.so``; there is no CPU path.
"""
import os as _os

# The engine overlaps batches on separate HIP streams (up to 16 slots + copy streams); ROCm's default of 4
# hardware queues would make them share queues, and with 8 two of eight slots still collide (measured: a
# pair of batches then runs back to back). Must be set before the HIP runtime initialises (import this
# package first).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

from . import buzhash  # noqa: F401,E402
from ._lib import RECORD_DTYPE, PbsGpuError  # noqa: F401,E402
from .engine import Chunker, Comm, Engine, PageRing, PayloadStream  # noqa: F401,E402

__all__ = ["buzhash", "Engine", "PayloadStream", "PageRing", "Chunker", "Comm", "RECORD_DTYPE", "PbsGpuError"]
