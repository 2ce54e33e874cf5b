"""ctypes binding of libpbsgpu (the C ABI in include/pbsgpu.h).

Loads the in-tree shared library ``pbs_plus_amd/lib/libpbsgpu.so`` (built by
``__graft_entry__.build()`` / ``make -C pbs_plus_amd/csrc``). There is no
fallback of any kind: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBSGPU_LIB_PATH") or os.path.join(_HERE, "lib", "libpbsgpu.so")   # (override: A/B builds, scripts/)
CSRC = os.path.join(_HERE, "csrc")

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_NOMEM, E_CAPACITY, E_BUSY, E_TICKET, _E_RETIRED_8, E_STATE = range(-1, -10, -1)  # (-8 was E_DENSITY until ABI v4)


class PbsGpuError(RuntimeError):
    def __init__(self, status: int, what: str = ""):
        self.status = status
        msg = lib().pbsgpu_strerror(status).decode() if _lib is not None else str(status)
        hip = lib().pbsgpu_last_hip_error() if _lib is not None else 0
        extra = ""
        if _lib is not None and ("comm" in what or "allgather" in what):
            txt = lib().pbsgpu_comm_last_error()
            if txt:
                extra = f"; RCCL: {txt.decode(errors='replace')}"
        super().__init__(f"{what}: {msg} (status {status}, hipError {hip}{extra})")


class Config(C.Structure):
    """pbsgpu_config — mirrors buzhash.Config (reference commit_orchestrate.go:143-149)."""

    _fields_ = [
        ("avg", C.c_uint32),
        ("min", C.c_uint32),
        ("max", C.c_uint32),
        ("window", C.c_uint32),
        ("mask", C.c_uint32),
        ("break_min", C.c_uint32),
        ("table", C.c_uint32 * 256),
    ]


class Segment(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("length", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [
        ("h2d_ms", C.c_float),
        ("scan_ms", C.c_float),
        ("resolve_ms", C.c_float),
        ("sha_ms", C.c_float),
        ("total_ms", C.c_float),
        ("ncandidates", C.c_uint64),
        ("nrecords", C.c_uint64),
        ("retries", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class PayloadFormat(C.Structure):
    _fields_ = [
        ("payload_type", C.c_uint64),
        ("start_type", C.c_uint64),
        ("tail_type", C.c_uint64),
        ("with_start", C.c_uint32),
        ("with_tail", C.c_uint32),
    ]


class ReuseChunk(C.Structure):
    _fields_ = [("size", C.c_uint64), ("padding", C.c_uint64), ("end_offset", C.c_uint64), ("digest", C.c_uint8 * 32)]


class FileHash(C.Structure):
    _fields_ = [("index", C.c_uint64), ("size", C.c_uint64), ("xxh3", C.c_uint64)]


RING_OFF = 0xFFFFFFFF
RING_F_NO_OVERLAP, RING_F_NO_STAGE, RING_F_NO_CUT_PRIO, RING_F_NO_SPLIT_AUTO, RING_F_DEFER_SERVICE, RING_F_FILL_SERIAL = 1, 2, 4, 8, 16, 32
RING_F_DENSE_SERVICE = 64
RING_F_DENSE_LANES = 128
RING_F_TIER_TAG = 256


class RingOptions(C.Structure):
    """pbsgpu_ring_options (ABI v5): positional use keeps working for the first six fields; the rest by keyword."""
    _fields_ = [("arena_bytes", C.c_uint64), ("page_bytes", C.c_uint64), ("max_streams", C.c_uint32),
                ("sha_cus", C.c_uint32), ("round_pages", C.c_uint32), ("express_cus", C.c_uint32),
                ("min_round_pages", C.c_uint32), ("max_inflight", C.c_uint32), ("long_bytes", C.c_uint32),
                ("long_lo_bytes", C.c_uint32), ("long_spill", C.c_uint32), ("poll_every", C.c_uint32), ("flags", C.c_uint32),
                ("lanes_cus", C.c_uint32), ("backlog_mib", C.c_double), ("lone_defer_ms", C.c_double),
                ("idle_timeout_s", C.c_double), ("autopark_ms", C.c_double), ("short_bytes", C.c_uint64), ("reserved", C.c_uint64 * 3)]


class EngineOptions(C.Structure):
    """pbsgpu_engine_options (ABI v5)"""
    _fields_ = [("inflight", C.c_uint32), ("sha_form", C.c_uint32), ("sha_slack_pct", C.c_uint32), ("sha_dense_pct", C.c_uint32),
                ("resolve_par_min", C.c_uint64), ("sha_many_files_per_core", C.c_uint32), ("stream_sha_cus", C.c_uint32),
                ("stream_express_cus", C.c_uint32), ("stream_ring_slots", C.c_uint32), ("stream_ctx_pool", C.c_uint32),
                ("reserved0", C.c_uint32), ("stream_ring_gib", C.c_double), ("stream_page_bytes", C.c_uint64),
                ("reserved", C.c_uint64 * 4)]


class RingStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("page_bytes", "bytes_enqueued", "chunks", "candidates", "pages_enqueued",
                                          "pages_recycled", "service_bytes_last")] + \
               [(k, C.c_uint32) for k in ("pages_total", "pages_free", "sha_cus", "rounds", "rounds_done",
                                          "rounds_in_flight", "streams_opened", "service_launches")] + \
               [("service_ms_last", C.c_double), ("service_ms_total", C.c_double)]


class RingProbe(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("pair_steps", "pair_cycles", "pair_ticks", "express_steps", "express_cycles",
                                          "express_ticks")]


class DedupStats(C.Structure):
    _fields_ = [
        ("nrecords", C.c_uint64),
        ("nunique", C.c_uint64),
        ("total_bytes", C.c_uint64),
        ("unique_bytes", C.c_uint64),
    ]


# pbsgpu_record: 48 bytes, same layout as a DIDX entry + (segment, size)
RECORD_DTYPE = np.dtype([("end", "<u8"), ("digest", "u1", (32,)), ("segment", "<u4"), ("size", "<u4")])
assert RECORD_DTYPE.itemsize == 48

# every symbol include/pbsgpu.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_U64P = C.POINTER(C.c_uint64)
SYMBOLS = {
    "pbsgpu_strerror": (C.c_char_p, [C.c_int]),
    "pbsgpu_abi_version": (C.c_int, []),
    "pbsgpu_last_hip_error": (C.c_int, []),
    "pbsgpu_device_count": (C.c_int, []),
    "pbsgpu_config_init": (C.c_int, [C.c_uint64, _P, C.POINTER(Config)]),
    "pbsgpu_default_table": (C.POINTER(C.c_uint32), []),
    "pbsgpu_engine_create": (C.c_int, [C.c_int, C.POINTER(Config), C.c_uint32, C.POINTER(_P)]),
    "pbsgpu_engine_create_opt": (C.c_int, [C.c_int, C.POINTER(Config), C.POINTER(EngineOptions), C.POINTER(_P)]),
    "pbsgpu_engine_destroy": (None, [_P]),
    "pbsgpu_engine_config": (C.c_int, [_P, C.POINTER(Config)]),
    "pbsgpu_engine_trim": (C.c_int, [_P, _U64P]),
    "pbsgpu_engine_set_suggested_feed": (C.c_int, [_P, C.c_uint64, C.c_int]),
    "pbsgpu_sha256_many_pays": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]),
    "pbsgpu_submit_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _U64P]),
    "pbsgpu_submit_host": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _U64P]),
    "pbsgpu_submit_device_suggested": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P, _P, _U64P]),
    "pbsgpu_submit_host_suggested": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P, _P, _U64P]),
    "pbsgpu_wait": (C.c_int, [_P, C.c_uint64, _U64P]),
    "pbsgpu_ticket_done": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_int)]),
    "pbsgpu_collect": (C.c_int, [_P, C.c_uint64, _P, C.c_uint64, _U64P]),
    "pbsgpu_ticket_timing": (C.c_int, [_P, C.c_uint64, C.POINTER(Timing)]),
    "pbsgpu_candidates_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, _U64P]),
    "pbsgpu_resolve_candidates": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P, C.c_uint64, _U64P]),
    "pbsgpu_chunker_create": (C.c_int, [_P, C.POINTER(_P)]),
    "pbsgpu_chunker_destroy": (None, [_P]),
    "pbsgpu_chunker_scan": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pbsgpu_chunker_reset": (C.c_int, [_P]),
    "pbsgpu_stream_create": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "pbsgpu_stream_destroy": (None, [_P]),
    "pbsgpu_stream_write": (C.c_int, [_P, _P, C.c_size_t]),
    "pbsgpu_stream_reserve": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "pbsgpu_stream_commit": (C.c_int, [_P, C.c_size_t]),
    "pbsgpu_stream_cut": (C.c_int, [_P, C.c_uint64]),
    "pbsgpu_stream_finish": (C.c_int, [_P]),
    "pbsgpu_stream_finish_begin": (C.c_int, [_P]),
    "pbsgpu_stream_done": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "pbsgpu_stream_poll": (C.c_int, [_P, _P, C.c_uint64, _U64P]),
    "pbsgpu_stream_position": (C.c_int, [_P, _U64P]),
    "pbsgpu_stream_bytes_written": (C.c_int, [_P, _U64P]),
    "pbsgpu_stream_suggest": (C.c_int, [_P, C.c_uint64]),
    "pbsgpu_stream_begin_file": (C.c_int, [_P]),
    "pbsgpu_stream_end_file": (C.c_int, [_P, _U64P]),
    "pbsgpu_stream_poll_files": (C.c_int, [_P, _P, C.c_uint64, _U64P]),
    "pbsgpu_stream_begin_entry": (C.c_int, [_P, _P, C.c_uint64, _U64P]),
    "pbsgpu_stream_end_entry": (C.c_int, [_P, _U64P]),
    "pbsgpu_stream_write_marker": (C.c_int, [_P, _P, C.c_int]),
    "pbsgpu_ring_create": (C.c_int, [_P, C.POINTER(RingOptions), C.POINTER(_P)]),
    "pbsgpu_ring_destroy": (None, [_P]),
    "pbsgpu_ring_open": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "pbsgpu_ring_reserve": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), _U64P]),
    "pbsgpu_ring_commit": (C.c_int, [_P, C.c_uint32, C.c_uint64, C.c_int]),
    "pbsgpu_ring_fill": (C.c_int, [_P, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, _U64P]),
    "pbsgpu_ring_pump": (C.c_int, [_P]),
    "pbsgpu_ring_poll": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, _U64P, C.POINTER(C.c_int)]),
    "pbsgpu_ring_poll_any": (C.c_int, [_P, _P, C.c_uint64, _U64P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "pbsgpu_ring_close": (C.c_int, [_P, C.c_uint32]),
    "pbsgpu_ring_quiesce": (C.c_int, [_P]),
    "pbsgpu_ring_park": (C.c_int, [_P]),
    "pbsgpu_ring_fill_pieces": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.c_uint64, C.c_int, _U64P]),
    "pbsgpu_ring_suggest": (C.c_int, [_P, C.c_uint32, C.c_uint64]),
    "pbsgpu_ring_get_stats": (C.c_int, [_P, C.POINTER(RingStats)]),
    "pbsgpu_ring_debug": (C.c_int, [_P, C.c_char_p, C.c_uint64]),
    "pbsgpu_ring_express": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "pbsgpu_ring_get_probe": (C.c_int, [_P, C.POINTER(RingProbe)]),
    "pbsgpu_sha256_many_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P]),
    "pbsgpu_sha256_many_host": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P]),
    "pbsgpu_xxh3_many_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P]),
    "pbsgpu_xxh3_many_host": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, _P]),
    "pbsgpu_dedup_host": (C.c_int, [_P, _P, C.c_uint64, _P, C.POINTER(DedupStats)]),
    "pbsgpu_dedup_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.POINTER(DedupStats)]),
    "pbsgpu_comm_unique_id": (C.c_int, [_P]),
    "pbsgpu_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "pbsgpu_comm_destroy": (None, [_P]),
    "pbsgpu_comm_rank": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pbsgpu_comm_last_error": (C.c_char_p, []),
    "pbsgpu_digest_allgather_dedup": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P, C.POINTER(DedupStats)]),
    "pbsgpu_split_plan": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_uint32] + [C.POINTER(C.c_uint64)] * 4),
    "pbsgpu_comm_split_stream": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pbsgpu_didx_size": (C.c_int, [C.c_uint64, _U64P]),
    "pbsgpu_didx_encode": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_int64, _P, C.c_uint64]),
    "pbsgpu_didx_decode": (C.c_int, [_P, C.c_uint64, _P, C.c_uint64, _U64P, C.POINTER(C.c_int64), _P]),
    "pbsgpu_payload_format_default": (C.c_int, [C.POINTER(PayloadFormat)]),
    "pbsgpu_payload_size": (C.c_int, [_P, C.c_uint32, C.POINTER(PayloadFormat), _U64P]),
    "pbsgpu_payload_pack_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint32, C.POINTER(PayloadFormat), _P, C.c_uint64,
                                             _U64P, _P]),
    "pbsgpu_reuse_lookup": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P, C.c_uint64, _U64P, _U64P, _U64P]),
    "pbsgpu_reuse_should": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, _P, C.c_double, C.POINTER(C.c_int)]),
    "pbsgpu_fill_device": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]),
    "pbsgpu_gather_device": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_uint32]),
    "pbsgpu_measure_h2d": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_double)]),
    "pbsgpu_device_alloc": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "pbsgpu_device_free": (C.c_int, [_P, _P]),
    "pbsgpu_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "pbsgpu_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_uint64]),
}


def build(force: bool = False) -> str:
    """Compile libpbsgpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "-s", "clean"], check=True)
    subprocess.run(["make", "-C", CSRC, "-s", "-j4", "all"], check=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError = the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int, what: str) -> None:
    if status != OK:
        raise PbsGpuError(status, what)
