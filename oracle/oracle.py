"""ctypes front end of the CPU parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package ``pbs_plus_amd`` never does.
PARITY UNPINNED against github.com/pbs-plus/pxar v0.34.0 (see oracle/oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


class Config(C.Structure):
    _fields_ = [
        ("avg", C.c_uint32),
        ("min", C.c_uint32),
        ("max", C.c_uint32),
        ("window", C.c_uint32),
        ("mask", C.c_uint32),
        ("break_min", C.c_uint32),
        ("table", C.c_uint32 * 256),
    ]


class Chunker(C.Structure):
    _fields_ = [
        ("cfg", Config),
        ("h", C.c_uint32),
        ("window_size", C.c_uint32),
        ("chunk_size", C.c_uint64),
        ("window", C.c_uint8 * 64),
    ]


class Segment(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("length", C.c_uint64)]


RECORD_DTYPE = np.dtype(
    [("end", "<u8"), ("digest", "u1", (32,)), ("segment", "<u4"), ("size", "<u4")]
)
assert RECORD_DTYPE.itemsize == 48


def build(force: bool = False) -> str:
    """Compile oracle/_build/liboracle.so with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("buzhash_oracle.c", "sha256_oracle.c", "oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_default_table.restype = C.POINTER(C.c_uint32)
        L.oracle_config_init.argtypes = [C.c_uint64, C.c_void_p, C.POINTER(Config)]
        L.oracle_config_init.restype = C.c_int
        L.oracle_chunker_init.argtypes = [C.POINTER(Chunker), C.POINTER(Config)]
        L.oracle_chunker_init.restype = None
        L.oracle_chunker_scan.argtypes = [C.POINTER(Chunker), C.c_void_p, C.c_size_t]
        L.oracle_chunker_scan.restype = C.c_size_t
        L.oracle_chunk_stream.argtypes = [C.POINTER(Config), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_chunk_stream.restype = C.c_size_t
        L.oracle_candidates.argtypes = [C.POINTER(Config), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_candidates.restype = C.c_size_t
        L.oracle_sha256.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.oracle_sha256.restype = None
        L.oracle_have_shani.restype = C.c_int
        L.oracle_chunk_and_digest.argtypes = [C.POINTER(Config), C.c_void_p, C.POINTER(Segment), C.c_uint32,
                                              C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_chunk_and_digest.restype = C.c_size_t
        L.oracle_chunk_stream_suggested.argtypes = [C.POINTER(Config), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                    C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_chunk_stream_suggested.restype = C.c_size_t
        L.oracle_chunk_stream_suggested_grid.argtypes = [C.POINTER(Config), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                         C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.oracle_chunk_stream_suggested_grid.restype = C.c_size_t
        L.oracle_chunk_and_digest_suggested.argtypes = [C.POINTER(Config), C.c_void_p, C.POINTER(Segment), C.c_uint32,
                                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_chunk_and_digest_suggested.restype = C.c_size_t
        L.oracle_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.oracle_fill.restype = None
        _lib = L
    return _lib


def default_table() -> np.ndarray:
    p = lib().oracle_default_table()
    return np.ctypeslib.as_array(p, shape=(256,)).copy()


def new_config(avg: int, table: np.ndarray | None = None) -> Config:
    cfg = Config()
    tp = None
    if table is not None:
        table = np.ascontiguousarray(table, dtype=np.uint32)
        assert table.shape == (256,)
        tp = table.ctypes.data
    if lib().oracle_config_init(avg, tp, C.byref(cfg)) != 0:
        raise ValueError(f"invalid average chunk size {avg}")
    return cfg


def _buf(data) -> np.ndarray:
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    return np.ascontiguousarray(a, dtype=np.uint8)


class StreamingChunker:
    """Upstream-style `scan(data) -> pos` chunker (0 = need more data)."""

    def __init__(self, cfg: Config):
        self._c = Chunker()
        lib().oracle_chunker_init(C.byref(self._c), C.byref(cfg))

    def scan(self, data) -> int:
        a = _buf(data)
        return int(lib().oracle_chunker_scan(C.byref(self._c), a.ctypes.data, a.size))


def chunk_stream(cfg: Config, data) -> np.ndarray:
    """Chunk END offsets (exclusive) of `data` cut as one stream."""
    a = _buf(data)
    cap = max(16, a.size // max(1, cfg.min) + 2)
    ends = np.empty(cap, dtype=np.uint64)
    n = lib().oracle_chunk_stream(C.byref(cfg), a.ctypes.data, a.size, ends.ctypes.data, cap)
    assert n <= cap
    return ends[:n].copy()


def chunk_stream_suggested(cfg: Config, data, suggested, feed: int = 1, absolute: bool = False) -> np.ndarray:
    """Chunk END offsets of `data` cut as one stream by the payload chunker (ChunkerImpl + suggested boundaries,
    absolute offsets in send order). feed = bytes handed to each scan call: 1 = byte-serial (the engine's
    default), 0 = the whole remaining buffer at once (upstream's second test loop), N = N bytes per call counted from
    the last cut or — absolute — ending at multiples of N from the stream start."""
    a = _buf(data)
    sg = np.ascontiguousarray(suggested, dtype=np.uint64)
    cap = max(16, a.size // max(1, cfg.min) + 2 + sg.size)
    ends = np.empty(cap, dtype=np.uint64)
    n = lib().oracle_chunk_stream_suggested_grid(C.byref(cfg), a.ctypes.data, a.size, sg.ctypes.data if sg.size else None,
                                                 sg.size, feed, int(absolute), ends.ctypes.data, cap)
    assert n <= cap
    return ends[:n].copy()


def candidates(cfg: Config, data) -> np.ndarray:
    """Raw Buzhash candidate END offsets (no min/max rules) — parity target of the scan kernel."""
    a = _buf(data)
    cap = 1024
    while True:
        out = np.empty(cap, dtype=np.uint64)
        n = lib().oracle_candidates(C.byref(cfg), a.ctypes.data, a.size, out.ctypes.data, cap)
        if n <= cap:
            return out[:n].copy()
        cap = int(n)


def sha256(data, impl: int = 1) -> bytes:
    a = _buf(data)
    out = (C.c_uint8 * 32)()
    lib().oracle_sha256(a.ctypes.data if a.size else None, a.size, out, impl)
    return bytes(out)


def chunk_and_digest(cfg: Config, data, segments=None, impl: int = 1) -> np.ndarray:
    """Records (end, digest, segment, size) for every chunk of every segment."""
    a = _buf(data)
    if segments is None:
        segments = [(0, a.size)]
    segs = (Segment * max(1, len(segments)))(*[Segment(int(o), int(n)) for o, n in segments])
    total = sum(int(n) for _, n in segments)
    cap = total // max(1, cfg.min) + 2 * len(segments) + 16
    out = np.zeros(cap, dtype=RECORD_DTYPE)
    n = lib().oracle_chunk_and_digest(C.byref(cfg), a.ctypes.data, segs, len(segments),
                                      out.ctypes.data, cap, impl)
    assert n <= cap
    return out[:n].copy()


def chunk_and_digest_suggested(cfg: Config, data, segments, suggested, impl: int = 1) -> np.ndarray:
    """Like chunk_and_digest, with per-segment suggested boundaries: `suggested` = list (one entry per segment) of
    ascending offsets relative to the segment start."""
    a = _buf(data)
    segs = (Segment * max(1, len(segments)))(*[Segment(int(o), int(n)) for o, n in segments])
    flat = np.concatenate([np.asarray(x, dtype=np.uint64).reshape(-1) for x in suggested]) if len(suggested) else np.zeros(0, np.uint64)
    flat = np.ascontiguousarray(flat, dtype=np.uint64)
    idx = np.zeros(len(segments) + 1, dtype=np.uint32)
    idx[1:] = np.cumsum([len(x) for x in suggested])
    total = sum(int(n) for _, n in segments)
    cap = total // max(1, cfg.min) + 2 * len(segments) + flat.size + 16
    out = np.zeros(cap, dtype=RECORD_DTYPE)
    n = lib().oracle_chunk_and_digest_suggested(C.byref(cfg), a.ctypes.data, segs, len(segments),
                                                flat.ctypes.data if flat.size else None, idx.ctypes.data,
                                                out.ctypes.data, cap, impl)
    assert n <= cap
    return out[:n].copy()


def fill(n: int, seed: int, kind: int = 0, stream_off: int = 0, out: np.ndarray | None = None) -> np.ndarray:
    """Deterministic synthetic bytes (twin of the engine's device fill kernel)."""
    if out is None:
        out = np.empty(n, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= n and out.flags.c_contiguous
    lib().oracle_fill(out.ctypes.data, stream_off, n, seed, kind)
    return out[:n]
