"""Engine / PayloadStream / Chunker — thin object wrappers over the C ABI.

Device memory: ``Engine`` accepts raw device pointers (ints), anything exposing
``data_ptr()``/``numel()``/``element_size()`` (a CUDA/HIP torch tensor) or host
buffers (bytes / numpy); torch is plumbing only and is never imported here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import RECORD_DTYPE, Segment, check
from .buzhash import Config


def _segs(segments):
    """[(offset, length)] or an (n, 2) uint64 array -> (pointer-compatible object, n).
    pbsgpu_segment is two little-endian u64, i.e. exactly one row of such an array."""
    if segments is None:
        return None, 0
    n = len(segments)
    if n == 0:
        return None, 0
    a = np.ascontiguousarray(segments, dtype=np.uint64).reshape(n, 2)
    return a.ctypes.data_as(C.POINTER(Segment)), n


def _host_view(data) -> np.ndarray:
    a = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1)


def _is_device_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and bool(x.is_cuda)


class DeviceBuffer:
    """Library-owned device allocation (for callers without torch)."""

    def __init__(self, eng: "Engine", nbytes: int):
        self._eng = eng
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(eng._L.pbsgpu_device_alloc(eng._h, self.nbytes, C.byref(p)), "device_alloc")
        self.ptr = p.value

    def free(self):
        if self.ptr:
            check(self._eng._L.pbsgpu_device_free(self._eng._h, self.ptr), "device_free")
            self.ptr = None

    def upload(self, data, offset: int = 0):
        a = _host_view(data)
        assert offset + a.size <= self.nbytes
        check(self._eng._L.pbsgpu_memcpy_h2d(self._eng._h, self.ptr + offset, a.ctypes.data, a.size), "memcpy_h2d")

    def download(self, offset: int = 0, nbytes: int | None = None) -> np.ndarray:
        n = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(n, dtype=np.uint8)
        check(self._eng._L.pbsgpu_memcpy_d2h(self._eng._h, out.ctypes.data, self.ptr + offset, n), "memcpy_d2h")
        return out


class Engine:
    """One engine per (process, GPU). Stands where backupproxy's session owns the
    chunker + hasher built from the buzhash.Config (commit_orchestrate.go:137-149)."""

    def __init__(self, config: Config, device: int = 0, inflight: int = 2, **options):
        """`options`: fields of pbsgpu_engine_options by name (sha_form=2, stream_ring_gib=8.0, ...); none = the defaults."""
        self._L = _lib.lib()
        self.config = config
        h = C.c_void_p()
        if options:
            o = _lib.EngineOptions()
            o.inflight = int(inflight)
            for k, v in options.items():
                setattr(o, k, v)
            check(self._L.pbsgpu_engine_create_opt(int(device), C.byref(config._c), C.byref(o), C.byref(h)), "engine_create_opt")
        else:
            check(self._L.pbsgpu_engine_create(int(device), C.byref(config._c), int(inflight), C.byref(h)),
                  "engine_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.pbsgpu_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_suggested_feed(self, feed_bytes: int = 1, absolute_grid: bool = False) -> None:
        """Reader buffer size the suggested-boundary rule emulates (1 = byte-serial, 0 = everything at once)."""
        check(self._L.pbsgpu_engine_set_suggested_feed(self._h, int(feed_bytes), int(absolute_grid)), "set_suggested_feed")

    # ---- device memory helpers -------------------------------------------------------------
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def fill(self, dptr: int, nbytes: int, seed: int, kind: int = 0, stream_off: int = 0) -> None:
        check(self._L.pbsgpu_fill_device(self._h, dptr, stream_off, nbytes, seed, kind), "fill_device")

    @staticmethod
    def _dev(data, nbytes=None):
        if isinstance(data, DeviceBuffer):
            return data.ptr, data.nbytes if nbytes is None else nbytes
        if _is_device_tensor(data):
            return data.data_ptr(), data.numel() * data.element_size() if nbytes is None else nbytes
        if isinstance(data, int):
            assert nbytes is not None
            return data, nbytes
        return None, None

    # ---- batch path ------------------------------------------------------------------------------
    def submit(self, data, segments=None, nbytes: int | None = None, suggested=None) -> int:
        """Enqueue cut + digest of every segment; returns a ticket. `suggested` = one ascending list of suggested
        boundaries per segment (relative to the segment start; payload-chunker rule, see pbsgpu.h)."""
        segs, nseg = _segs(segments)
        t = C.c_uint64()
        ptr, n = self._dev(data, nbytes)
        if suggested is not None:
            ns = max(nseg, 1)
            assert len(suggested) == ns, "one suggested-boundary list per segment"
            flat = (np.concatenate([np.asarray(x, dtype=np.uint64).reshape(-1) for x in suggested])
                    if ns else np.zeros(0, np.uint64))
            flat = np.ascontiguousarray(flat, dtype=np.uint64)
            idx = np.zeros(ns + 1, dtype=np.uint32)
            idx[1:] = np.cumsum([len(x) for x in suggested])
            fp = flat.ctypes.data if flat.size else None
            if ptr is not None:
                check(self._L.pbsgpu_submit_device_suggested(self._h, ptr, n, segs, nseg, fp, idx.ctypes.data,
                                                             C.byref(t)), "submit_device_suggested")
            else:
                a = _host_view(data)
                check(self._L.pbsgpu_submit_host_suggested(self._h, a.ctypes.data, a.size, segs, nseg, fp,
                                                           idx.ctypes.data, C.byref(t)), "submit_host_suggested")
            return t.value
        if ptr is not None:
            check(self._L.pbsgpu_submit_device(self._h, ptr, n, segs, nseg, C.byref(t)), "submit_device")
        else:
            a = _host_view(data)
            check(self._L.pbsgpu_submit_host(self._h, a.ctypes.data, a.size, segs, nseg, C.byref(t)), "submit_host")
        return t.value

    def h2d_bandwidth(self, nbytes: int = 1 << 30) -> float:
        """Measured pinned host -> device copy rate of this box in GB/s."""
        v = C.c_double()
        check(self._L.pbsgpu_measure_h2d(self._h, int(nbytes), C.byref(v)), "measure_h2d")
        return float(v.value)

    def wait(self, ticket: int) -> int:
        n = C.c_uint64()
        check(self._L.pbsgpu_wait(self._h, ticket, C.byref(n)), "wait")
        return n.value

    def done(self, ticket: int) -> bool:
        """Non-blocking: has everything enqueued for the ticket finished on the device?"""
        d = C.c_int(0)
        check(self._L.pbsgpu_ticket_done(self._h, int(ticket), C.byref(d)), "ticket_done")
        return bool(d.value)

    def timing(self, ticket: int) -> dict:
        t = _lib.Timing()
        check(self._L.pbsgpu_ticket_timing(self._h, ticket, C.byref(t)), "ticket_timing")
        return {k: getattr(t, k) for k, _ in _lib.Timing._fields_ if k != "reserved"}

    def collect(self, ticket: int) -> np.ndarray:
        n = self.wait(ticket)
        out = np.zeros(max(n, 1), dtype=RECORD_DTYPE)
        got = C.c_uint64()
        check(self._L.pbsgpu_collect(self._h, ticket, out.ctypes.data, n, C.byref(got)), "collect")
        return out[: got.value]

    def chunk_and_digest(self, data, segments=None, nbytes: int | None = None, suggested=None) -> np.ndarray:
        return self.collect(self.submit(data, segments, nbytes, suggested))

    def candidates(self, data, nbytes: int | None = None) -> np.ndarray:
        """Raw Buzhash candidates (ascending END offsets) of a device byte range."""
        ptr, n = self._dev(data, nbytes)
        assert ptr is not None, "candidates() wants device memory"
        cnt = C.c_uint64()
        st = self._L.pbsgpu_candidates_device(self._h, ptr, n, None, 0, C.byref(cnt))
        if st == _lib.OK and cnt.value == 0:
            return np.zeros(0, dtype=np.uint64)
        if st != _lib.E_CAPACITY:
            check(st, "candidates_device")
        out = np.empty(cnt.value, dtype=np.uint64)
        check(self._L.pbsgpu_candidates_device(self._h, ptr, n, out.ctypes.data, out.size, C.byref(cnt)),
              "candidates_device")
        return out[: cnt.value]

    def resolve_candidates(self, cands: np.ndarray, stream_len: int) -> np.ndarray:
        """Cut one stream from an explicit ascending candidate list (records without digests)."""
        c = np.ascontiguousarray(cands, dtype=np.uint64)
        n = C.c_uint64()
        st = self._L.pbsgpu_resolve_candidates(self._h, c.ctypes.data if c.size else None, c.size, stream_len, None, 0,
                                               C.byref(n))
        if st not in (_lib.OK, _lib.E_CAPACITY):
            check(st, "resolve_candidates")
        out = np.zeros(max(n.value, 1), dtype=RECORD_DTYPE)
        check(self._L.pbsgpu_resolve_candidates(self._h, c.ctypes.data if c.size else None, c.size, stream_len,
                                                out.ctypes.data, n.value, C.byref(n)), "resolve_candidates")
        return out[: n.value]

    # ---- whole-stream SHA-256 (verification.HashFile for many files) --------------------------------
    def sha256_many(self, data, segments, nbytes: int | None = None) -> np.ndarray:
        segs, nseg = _segs(segments)
        out = np.zeros((max(nseg, 1), 32), dtype=np.uint8)
        ptr, n = self._dev(data, nbytes)
        if ptr is not None:
            check(self._L.pbsgpu_sha256_many_device(self._h, ptr, n, segs, nseg, out.ctypes.data), "sha256_many_device")
        else:
            a = _host_view(data)
            check(self._L.pbsgpu_sha256_many_host(self._h, a.ctypes.data if a.size else None, a.size, segs, nseg,
                                                  out.ctypes.data), "sha256_many_host")
        return out[:nseg]

    # ---- whole-stream XXH3-64 (the xxh3.New() tee of writeBackedFile / verifyBackedFileHashes) -------
    def xxh3_many(self, data, segments, nbytes: int | None = None) -> np.ndarray:
        segs, nseg = _segs(segments)
        out = np.zeros(max(nseg, 1), dtype=np.uint64)
        ptr, n = self._dev(data, nbytes)
        if ptr is not None:
            check(self._L.pbsgpu_xxh3_many_device(self._h, ptr, n, segs, nseg, out.ctypes.data), "xxh3_many_device")
        else:
            a = _host_view(data)
            check(self._L.pbsgpu_xxh3_many_host(self._h, a.ctypes.data if a.size else None, a.size, segs, nseg,
                                                out.ctypes.data), "xxh3_many_host")
        return out[:nseg]

    # ---- payload-stream assembly (.ppxar layout: markers + 16-byte headers + file bodies) ----------
    def payload_pack(self, src, files, dst, with_start: bool = True, with_tail: bool = True):
        """Lay the file bodies `files` = [(offset, length)] of device buffer `src` out as the pxar
        payload stream in device buffer `dst`. Returns (stream length, payload offset of every file's header)."""
        fmt = _lib.PayloadFormat()
        check(self._L.pbsgpu_payload_format_default(C.byref(fmt)), "payload_format_default")
        fmt.with_start, fmt.with_tail = int(with_start), int(with_tail)
        segs, n = _segs(files)
        sp, sn = self._dev(src)
        dp, dn = self._dev(dst)
        out_len = C.c_uint64()
        offs = np.zeros(max(n, 1), dtype=np.uint64)
        check(self._L.pbsgpu_payload_pack_device(self._h, sp, sn, segs, n, C.byref(fmt), dp, dn, C.byref(out_len),
                                                 offs.ctypes.data), "payload_pack_device")
        return out_len.value, offs[:n]

    def gather(self, src, dst, items) -> None:
        """Piece-table copy on the device: items = (n, 3) uint64 rows (src_off, dst_off, len)."""
        a = np.ascontiguousarray(items, dtype=np.uint64).reshape(-1, 3)
        sp, sn = self._dev(src)
        dp, dn = self._dev(dst)
        check(self._L.pbsgpu_gather_device(self._h, sp, sn, dp, dn, a.ctypes.data if a.size else None, a.shape[0]),
              "gather_device")

    # ---- digest set ------------------------------------------------------------------------------------
    def dedup(self, records: np.ndarray):
        """(dup flags, stats) — dup[i] = 1 when an earlier record has the same digest."""
        recs = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        dup = np.zeros(max(recs.size, 1), dtype=np.uint8)
        st = _lib.DedupStats()
        check(self._L.pbsgpu_dedup_host(self._h, recs.ctypes.data if recs.size else None, recs.size, dup.ctypes.data,
                                        C.byref(st)), "dedup_host")
        return dup[: recs.size], {k: getattr(st, k) for k, _ in _lib.DedupStats._fields_}

    def dedup_device(self, dptr: int, n: int, want_flags: bool = True):
        """dedup() on n records that already are in device memory (e.g. an RCCL all-gather's output)."""
        dup = np.zeros(max(n, 1), dtype=np.uint8) if want_flags else None
        st = _lib.DedupStats()
        check(self._L.pbsgpu_dedup_device(self._h, dptr, n, dup.ctypes.data if want_flags else None, C.byref(st)),
              "dedup_device")
        return (dup[:n] if want_flags else None), {k: getattr(st, k) for k, _ in _lib.DedupStats._fields_}

    # ---- dynamic index ------------------------------------------------------------------------------
    def didx_encode(self, records: np.ndarray, uuid: bytes = b"\0" * 16, ctime: int = 0) -> bytes:
        recs = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        nb = C.c_uint64()
        check(self._L.pbsgpu_didx_size(recs.size, C.byref(nb)), "didx_size")
        out = np.zeros(nb.value, dtype=np.uint8)
        u = (C.c_uint8 * 16).from_buffer_copy(uuid)
        check(self._L.pbsgpu_didx_encode(self._h, recs.ctypes.data if recs.size else None, recs.size, u, ctime,
                                         out.ctypes.data, out.size), "didx_encode")
        return out.tobytes()


def didx_decode(blob: bytes):
    """(records, ctime, index_csum) from a .didx image."""
    L = _lib.lib()
    a = np.frombuffer(blob, dtype=np.uint8)
    n = C.c_uint64()
    ct = C.c_int64()
    cs = (C.c_uint8 * 32)()
    st = L.pbsgpu_didx_decode(a.ctypes.data, a.size, None, 0, C.byref(n), C.byref(ct), cs)
    if st not in (_lib.OK, _lib.E_CAPACITY):
        check(st, "didx_decode")
    out = np.zeros(max(n.value, 1), dtype=RECORD_DTYPE)
    check(L.pbsgpu_didx_decode(a.ctypes.data, a.size, out.ctypes.data, n.value, C.byref(n), C.byref(ct), cs),
          "didx_decode")
    return out[: n.value], ct.value, bytes(cs)


class PayloadStream:
    """The payload-stream seam ``WriteEntryReader`` feeds (transfer.ArchiveWriter,
    internal/pxarmount/commit_test.go:33-67): append bytes, get (end, digest) records."""

    def __init__(self, eng: Engine, window_bytes: int = 0):
        self._eng = eng
        self._L = eng._L
        h = C.c_void_p()
        check(self._L.pbsgpu_stream_create(eng._h, window_bytes, C.byref(h)), "stream_create")
        self._h = h

    def write(self, data) -> None:
        a = _host_view(data)
        check(self._L.pbsgpu_stream_write(self._h, a.ctypes.data if a.size else None, a.size), "stream_write")

    def reserve(self) -> np.ndarray:
        """Borrow pinned staging memory as a writable uint8 array; fill a prefix, then commit(n)."""
        buf, cap = C.c_void_p(), C.c_size_t()
        check(self._L.pbsgpu_stream_reserve(self._h, C.byref(buf), C.byref(cap)), "stream_reserve")
        return np.ctypeslib.as_array((C.c_uint8 * cap.value).from_address(buf.value))

    def commit(self, n: int) -> None:
        check(self._L.pbsgpu_stream_commit(self._h, n), "stream_commit")

    def inject(self, nbytes: int = 0) -> None:
        """Forced cut (InjectChunks flushes the open chunk, commit_reuse.go:315-341)."""
        check(self._L.pbsgpu_stream_cut(self._h, nbytes), "stream_cut")

    def finish(self) -> None:
        check(self._L.pbsgpu_stream_finish(self._h), "stream_finish")

    def finish_begin(self) -> None:
        """Close the input without waiting for the last records (the writer goes on with its next archive); poll() keeps
        delivering, done() tells when the last record is out."""
        check(self._L.pbsgpu_stream_finish_begin(self._h), "stream_finish_begin")

    def done(self) -> bool:
        d = C.c_int()
        check(self._L.pbsgpu_stream_done(self._h, C.byref(d)), "stream_done")
        return bool(d.value)

    def poll(self, cap: int = 1 << 16) -> np.ndarray:
        outs = []
        while True:
            out = np.zeros(cap, dtype=RECORD_DTYPE)
            n = C.c_uint64()
            check(self._L.pbsgpu_stream_poll(self._h, out.ctypes.data, cap, C.byref(n)), "stream_poll")
            outs.append(out[: n.value])
            if n.value < cap:
                break
        return np.concatenate(outs) if outs else np.zeros(0, dtype=RECORD_DTYPE)

    def position(self) -> int:
        """Encoder().PayloadPosition(): bytes written + bytes injected (the records' coordinate system)."""
        n = C.c_uint64()
        check(self._L.pbsgpu_stream_position(self._h, C.byref(n)), "stream_position")
        return n.value

    def bytes_written(self) -> int:
        n = C.c_uint64()
        check(self._L.pbsgpu_stream_bytes_written(self._h, C.byref(n)), "stream_bytes_written")
        return n.value

    # ---- per-file XXH3-64 tee + pxar payload entries ---------------------------------------------------------
    def begin_file(self) -> None:
        check(self._L.pbsgpu_stream_begin_file(self._h), "stream_begin_file")

    def end_file(self) -> int:
        i = C.c_uint64()
        check(self._L.pbsgpu_stream_end_file(self._h, C.byref(i)), "stream_end_file")
        return i.value

    def begin_entry(self, content_len: int) -> int:
        """Append the 16-byte payload header of a file body of content_len bytes; returns its payload offset
        (the PAYLOAD_REF / WriteEntryRef value) and opens the file tee."""
        off = C.c_uint64()
        check(self._L.pbsgpu_stream_begin_entry(self._h, None, int(content_len), C.byref(off)), "stream_begin_entry")
        return off.value

    def end_entry(self) -> int:
        i = C.c_uint64()
        check(self._L.pbsgpu_stream_end_entry(self._h, C.byref(i)), "stream_end_entry")
        return i.value

    def write_marker(self, tail: bool = False) -> None:
        check(self._L.pbsgpu_stream_write_marker(self._h, None, int(tail)), "stream_write_marker")

    def poll_files(self, cap: int = 4096) -> list:
        """[(index, size, xxh3)] of the files hashed so far (in file order)."""
        outs = []
        while True:
            buf = (_lib.FileHash * cap)()
            n = C.c_uint64()
            check(self._L.pbsgpu_stream_poll_files(self._h, buf, cap, C.byref(n)), "stream_poll_files")
            outs += [(int(buf[i].index), int(buf[i].size), int(buf[i].xxh3)) for i in range(n.value)]
            if n.value < cap:
                return outs

    def suggest(self, offset: int | None = None) -> None:
        """Suggest a chunk boundary at absolute payload position `offset` (default: here)."""
        check(self._L.pbsgpu_stream_suggest(self._h, self.position() if offset is None else int(offset)),
              "stream_suggest")

    def close(self):
        if getattr(self, "_h", None):
            self._L.pbsgpu_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def split_plan(total_len: int, world: int, rank: int, max_chunk: int):
    """(own_start, own_end, lo, hi) of `rank` for a stream split over `world` ranks (pbsgpu_split_plan: arithmetic only)."""
    v = [C.c_uint64() for _ in range(4)]
    check(_lib.lib().pbsgpu_split_plan(int(total_len), int(world), int(rank), int(max_chunk), *[C.byref(x) for x in v]),
          "split_plan")
    return tuple(int(x.value) for x in v)


class Comm:
    """The multi-GPU digest-set reduce through the C ABI (pbsgpu_comm_*): ONE RCCL all-gather of the ranks' (digest, size)
    records over xGMI + the device dedup — what a Go host binds directly (go/pbsgpu: Comm), no torch involved.
    ``Comm.unique_id()`` on rank 0, ship the 128 bytes to the other ranks, ``Comm(engine, id, rank, world)`` on every rank
    (collective), then ``dedup(records, cap_records)`` on every rank in the same order."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        check(_lib.lib().pbsgpu_comm_unique_id(buf), "comm_unique_id")
        return bytes(buf)

    def __init__(self, eng: Engine, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == Comm.ID_BYTES
        self._eng = eng
        self._L = eng._L
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        idb = (C.c_uint8 * Comm.ID_BYTES).from_buffer_copy(unique_id)
        check(self._L.pbsgpu_comm_create(eng._h, idb, self.rank, self.world, C.byref(h)), "comm_create")
        self._h = h

    def dedup(self, records: np.ndarray, cap_records: int, want_flags: bool = True):
        """(dup flags of THIS rank's records, stats of the union over all ranks). cap_records: the same on every rank."""
        recs = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
        dup = np.zeros(max(recs.size, 1), dtype=np.uint8) if want_flags else None
        st = _lib.DedupStats()
        check(self._L.pbsgpu_digest_allgather_dedup(self._h, recs.ctypes.data if recs.size else None, recs.size,
                                                    int(cap_records), dup.ctypes.data if want_flags else None,
                                                    C.byref(st)), "digest_allgather_dedup")
        return (dup[: recs.size] if want_flags else None), {k: getattr(st, k) for k, _ in _lib.DedupStats._fields_}

    def split_stream(self, local_dptr: int, total_len: int) -> np.ndarray:
        """Cut + hash ONE stream of total_len bytes that is split over the ranks (collective, pbsgpu_comm_split_stream):
        local_dptr = device pointer to this rank's bytes [lo, hi) of split_plan(). The whole stream's records."""
        cap = int(total_len) // max(64, min(self._eng.config.MinSize, 1 << 30)) + 16
        out = np.zeros(cap, dtype=RECORD_DTYPE)
        n = C.c_uint64()
        check(self._L.pbsgpu_comm_split_stream(self._h, int(local_dptr) if local_dptr else None, int(total_len),
                                               out.ctypes.data, cap, C.byref(n)), "comm_split_stream")
        return out[: n.value].copy()

    def dedup_device(self, dptr: int, n: int, cap_records: int, want_flags: bool = True):
        """dedup() on n records that already are in DEVICE memory (they travel device -> device, no host round trip)."""
        dup = np.zeros(max(n, 1), dtype=np.uint8) if want_flags else None
        st = _lib.DedupStats()
        check(self._L.pbsgpu_digest_allgather_dedup(self._h, int(dptr) if n else None, int(n), int(cap_records),
                                                    dup.ctypes.data if want_flags else None, C.byref(st)),
              "digest_allgather_dedup")
        return (dup[:n] if want_flags else None), {k: getattr(st, k) for k, _ in _lib.DedupStats._fields_}

    def close(self):
        if getattr(self, "_h", None):
            self._L.pbsgpu_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Chunker:
    """Upstream-style streaming chunker: ``scan(data) -> pos`` (0 = no boundary yet)."""

    def __init__(self, eng: Engine):
        self._eng = eng
        self._L = eng._L
        h = C.c_void_p()
        check(self._L.pbsgpu_chunker_create(eng._h, C.byref(h)), "chunker_create")
        self._h = h

    def scan(self, data) -> int:
        a = _host_view(data)
        pos = C.c_size_t()
        check(self._L.pbsgpu_chunker_scan(self._h, a.ctypes.data if a.size else None, a.size, C.byref(pos)),
              "chunker_scan")
        return pos.value

    def reset(self) -> None:
        check(self._L.pbsgpu_chunker_reset(self._h), "chunker_reset")

    def close(self):
        if getattr(self, "_h", None):
            self._L.pbsgpu_chunker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PageRing:
    """Many payload streams through one device arena with page-granular memory release and the persistent SHA-256
    service (pbsgpu_ring_*, include/pbsgpu.h): the chunk loop behind ``WriteEntryReader`` for several archives at once
    (internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836) when bytes are in device memory."""

    def __init__(self, eng: Engine, arena_bytes: int = 0, page_bytes: int = 0, max_streams: int = 0, sha_cus: int = 0,
                 round_pages: int = 0, express_cus: int = 0, **options):
        """`options`: the other fields of pbsgpu_ring_options by name (flags=_lib.RING_F_NO_STAGE, lone_defer_ms=-1.0, ...)."""
        self._eng = eng
        self._L = eng._L
        opt = _lib.RingOptions(int(arena_bytes), int(page_bytes), int(max_streams), int(sha_cus), int(round_pages),
                               int(express_cus))
        for k, v in options.items():
            setattr(opt, k, v)
        h = C.c_void_p()
        check(self._L.pbsgpu_ring_create(eng._h, C.byref(opt), C.byref(h)), "ring_create")
        self._h = h
        self.page_bytes = self.stats()["page_bytes"]

    def open(self) -> int:
        s = C.c_uint32()
        check(self._L.pbsgpu_ring_open(self._h, C.byref(s)), "ring_open")
        return s.value

    def reserve(self, stream: int):
        """(device pointer, capacity) of the stream's next page, or None when no page is free right now."""
        p, cap = C.c_void_p(), C.c_uint64()
        st = self._L.pbsgpu_ring_reserve(self._h, stream, C.byref(p), C.byref(cap))
        if st == _lib.E_BUSY:
            return None
        check(st, "ring_reserve")
        return p.value, cap.value

    def commit(self, stream: int, nbytes: int, final: bool = False) -> None:
        check(self._L.pbsgpu_ring_commit(self._h, stream, int(nbytes), int(final)), "ring_commit")

    def fill(self, stream: int, seed: int, kind: int, nbytes: int, final: bool = False) -> int:
        """Synthetic producer: bytes accepted (whole pages that were free); call again with the rest after pump()."""
        t = C.c_uint64()
        check(self._L.pbsgpu_ring_fill(self._h, stream, int(seed), int(kind), int(nbytes), int(final), C.byref(t)),
              "ring_fill")
        return t.value

    def fill_pieces(self, stream: int, pieces, nbytes: int, final: bool = False) -> int:
        """Synthetic producer for an EDITED stream: `pieces` = (n, 4) uint64 rows (dst_off, len, src_off, seed) over
        generator 4 with the stream's first call, None afterwards. Bytes accepted, as fill()."""
        t = C.c_uint64()
        if pieces is not None:
            a = np.ascontiguousarray(pieces, dtype=np.uint64).reshape(-1, 4)
            ptr, n = a.ctypes.data, a.shape[0]
        else:
            ptr, n = None, 0
        check(self._L.pbsgpu_ring_fill_pieces(self._h, stream, ptr, n, int(nbytes), int(final), C.byref(t)), "ring_fill_pieces")
        return t.value

    def pump(self) -> None:
        check(self._L.pbsgpu_ring_pump(self._h), "ring_pump")

    def poll(self, stream: int, cap: int = 4096):
        """(records, finished) — records in stream order, `end` = absolute stream offset."""
        out = np.zeros(cap, dtype=RECORD_DTYPE)
        n, fin = C.c_uint64(), C.c_int()
        check(self._L.pbsgpu_ring_poll(self._h, stream, out.ctypes.data, cap, C.byref(n), C.byref(fin)), "ring_poll")
        return out[: n.value], bool(fin.value)

    def poll_any(self, cap: int = 16384, fcap: int = 4096):
        """(records of any open stream with `segment` = stream id, ids of the streams that have just finished)"""
        if getattr(self, "_any_buf", None) is None or self._any_buf.size < cap:
            self._any_buf = np.zeros(cap, dtype=RECORD_DTYPE)
            self._fin_buf = np.zeros(fcap, dtype=np.uint32)
        n, nf = C.c_uint64(), C.c_uint32()
        check(self._L.pbsgpu_ring_poll_any(self._h, self._any_buf.ctypes.data, cap, C.byref(n), self._fin_buf.ctypes.data,
                                           min(fcap, self._fin_buf.size), C.byref(nf)), "ring_poll_any")
        return self._any_buf[: n.value].copy(), self._fin_buf[: nf.value].copy()

    def close_stream(self, stream: int) -> None:
        check(self._L.pbsgpu_ring_close(self._h, stream), "ring_close")

    def quiesce(self) -> None:
        check(self._L.pbsgpu_ring_quiesce(self._h), "ring_quiesce")

    def park(self) -> None:
        """Stop the service behind everything enqueued WITHOUT waiting for it (the next pump starts a new one)."""
        check(self._L.pbsgpu_ring_park(self._h), "ring_park")

    def suggest(self, stream: int, offset: int) -> None:
        """Suggested boundary at `offset` bytes from the stream's start (ascending, ahead of the bytes around it)."""
        check(self._L.pbsgpu_ring_suggest(self._h, stream, int(offset)), "ring_suggest")

    def stats(self) -> dict:
        st = _lib.RingStats()
        check(self._L.pbsgpu_ring_get_stats(self._h, C.byref(st)), "ring_get_stats")
        return {k: getattr(st, k) for k, _ in _lib.RingStats._fields_}

    def express(self) -> tuple:
        """(CUs of the express service, chunk size from which a chunk takes it) — (0, 0) when the ring has none"""
        cus, lb = C.c_uint32(0), C.c_uint64(0)
        check(self._L.pbsgpu_ring_express(self._h, C.byref(cus), C.byref(lb)), "ring_express")
        return int(cus.value), int(lb.value)

    def probe(self) -> dict:
        """Cumulative regime counters of the two services (pbsgpu_ring_get_probe): steps, shader cycles, 100 MHz ticks."""
        p = _lib.RingProbe()
        check(self._L.pbsgpu_ring_get_probe(self._h, C.byref(p)), "ring_get_probe")
        return {k: int(getattr(p, k)) for k, _ in _lib.RingProbe._fields_}

    @staticmethod
    def probe_delta(a: dict, b: dict) -> dict:
        """What the services did between two probe() readings: ns per block step and shader clock, per service."""
        out = {}
        for svc, blocks in (("pair", 1), ("express", 2)):
            st, cy, tk = (b[f"{svc}_{k}"] - a[f"{svc}_{k}"] for k in ("steps", "cycles", "ticks"))
            if st and tk:
                out[svc] = {"ns_per_block_step": round(tk * 10.0 / st / blocks, 1), "sclk_mhz": round(cy * 100.0 / tk, 1),
                            "wave_steps_sampled": int(st)}
        return out

    def debug(self) -> str:
        buf = C.create_string_buffer(1 << 16)
        check(self._L.pbsgpu_ring_debug(self._h, buf, len(buf)), "ring_debug")
        return buf.value.decode(errors="replace")

    def ingest_synthetic(self, jobs, timeout_s: float = 120.0, concurrent: int | None = None):
        """Drive whole synthetic streams through the ring: jobs = [(seed, kind, nbytes)]; returns one record array per
        job. At most `concurrent` (default: all slots) streams are open at a time."""
        import time

        res = [[] for _ in jobs]
        todo = list(range(len(jobs)))
        active = {}                                   # stream id -> [job index, bytes left]
        limit = concurrent or len(jobs)
        t0 = time.perf_counter()
        while todo or active:
            # `limit` streams are FED at a time; a stream whose bytes are all in only waits for its last chunks
            while todo and sum(1 for v in active.values() if not v[2]) < limit:
                try:
                    sid = self.open()
                except _lib.PbsGpuError as exc:
                    if exc.status != _lib.E_BUSY:
                        raise
                    break
                j = todo.pop(0)
                active[sid] = [j, int(jobs[j][2]), False]
            quota = 16 * self.page_bytes            # pages are dealt out evenly among the open streams
            for sid, a in active.items():
                j, left, sent_final = a
                if not sent_final:
                    want = min(left, quota)
                    got = self.fill(sid, jobs[j][0], jobs[j][1], want, final=(want == left))
                    a[1] -= got
                    if a[1] == 0 and (got == want):
                        a[2] = True
            self.pump()
            for sid in list(active):
                recs, fin = self.poll(sid)
                if recs.size:
                    res[active[sid][0]].append(recs.copy())
                if fin:
                    self.close_stream(sid)
                    del active[sid]
            if time.perf_counter() - t0 > timeout_s:
                raise TimeoutError(f"ring ingest did not finish in {timeout_s} s: {self.stats()}\n{self.debug()}")
        return [np.concatenate(r) if r else np.zeros(0, dtype=RECORD_DTYPE) for r in res]

    def close(self):
        if getattr(self, "_h", None):
            self._L.pbsgpu_ring_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
