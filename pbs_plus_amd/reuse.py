"""Chunk-reuse planner: host-side mirror of the commit walk's index arithmetic over a previous
snapshot's dynamic index (reference internal/pxarmount/commit_reuse.go:84-183). Pure index math on the
engine's record lists; runs without a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import RECORD_DTYPE, ReuseChunk, check

CHUNK_PADDING_THRESHOLD = 0.1  # internal/pxarmount/commit_types.go:14
HEADER_SIZE = 16               # format.HeaderSize (commit_types.go:24-32, keepLast_chunk_test.go:105)


def range_end(sort_key: int, file_size: int) -> int:
    """commitEntry.rangeEnd: payload offset just past a file's header + body."""
    return sort_key + file_size + HEADER_SIZE


def pending_refs_range(refs):
    """pendingRefsRange over [(sortKey, fileSize)]: (first sortKey, max rangeEnd)."""
    if not refs:
        return 0, 0
    return refs[0][0], max(range_end(k, n) for k, n in refs)


def lookup_dynamic_entries(index: np.ndarray, range_start: int, range_end_: int):
    """lookupDynamicEntries: (chunks, startPadding, endPadding); chunks = list of dicts."""
    idx = np.ascontiguousarray(index, dtype=RECORD_DTYPE)
    L = _lib.lib()
    n, sp, ep = C.c_uint64(), C.c_uint64(), C.c_uint64()
    ptr = idx.ctypes.data if idx.size else None
    st = L.pbsgpu_reuse_lookup(ptr, idx.size, range_start, range_end_, None, 0, C.byref(n), C.byref(sp), C.byref(ep))
    check(st, "reuse_lookup")
    out = (ReuseChunk * max(n.value, 1))()
    check(L.pbsgpu_reuse_lookup(ptr, idx.size, range_start, range_end_, out, n.value, C.byref(n), C.byref(sp),
                                C.byref(ep)), "reuse_lookup")
    chunks = [{"size": out[i].size, "padding": out[i].padding, "endOffset": out[i].end_offset,
               "digest": bytes(out[i].digest)} for i in range(n.value)]
    return chunks, sp.value, ep.value


def should_reuse(index, refs, saved=None, threshold: float = CHUNK_PADDING_THRESHOLD) -> bool:
    """shouldReuse for pending refs [(sortKey, fileSize)]; index None/empty -> True (nothing to compare)."""
    if index is None or len(index) == 0 or not refs:
        return True
    idx = np.ascontiguousarray(index, dtype=RECORD_DTYPE)
    rs, re_ = pending_refs_range(refs)
    sv = None
    if saved is not None:
        sv = ReuseChunk(saved["size"], saved["padding"], saved["endOffset"], (C.c_uint8 * 32)(*saved["digest"]))
    r = C.c_int()
    check(_lib.lib().pbsgpu_reuse_should(idx.ctypes.data, idx.size, rs, re_, C.byref(sv) if sv is not None else None,
                                         threshold, C.byref(r)), "reuse_should")
    return bool(r.value)
