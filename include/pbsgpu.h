/*
 * include/pbsgpu.h — C ABI of libpbsgpu: the MI355X (gfx950) content-defined
 * chunker + per-chunk SHA-256 engine.
 *
 * This is the drop-in boundary for the reference's pxar stream path. The
 * reference (pure Go, CGO disabled) reaches the chunker only through the Go
 * package API of github.com/pbs-plus/pxar v0.34.0 (go.mod:30); the entry
 * points below are what a cgo binding inside that module's `buzhash` /
 * `backupproxy` packages would bind (INTEGRATION.md shows the stub). Every
 * declaration cites the reference interface it stands behind.
 *
 * Conventions: plain pointers and sizes only; every function returns an int
 * status (PBSGPU_OK == 0, negative = error, text via pbsgpu_strerror); opaque
 * handles; no thread-local state — a handle may be used from any OS thread
 * (cgo calls arrive on arbitrary threads). An ENGINE may be used from several
 * threads at once (submit / collect / helper calls / any number of streams and
 * chunkers created from it: the library never holds a lock across a device
 * wait or a copy, and helper calls queue for a work context instead of failing);
 * calls on ONE stream or chunker handle must be serialised by the caller, like
 * a Go writer owned by one goroutine (internal/tapeio/converter.go:672-680).
 * Streams and chunkers keep their engine alive: destroy order does not matter. Host memory passed in is never
 * retained after the call returns (cgo pointer rule): it is copied into
 * library-owned pinned staging first. Device pointers are borrowed until the
 * ticket they were submitted under has been collected.
 *
 * There is NO CPU fallback: without a usable HIP device every engine call
 * fails with PBSGPU_E_NO_DEVICE.
 */
#ifndef PBSGPU_H
#define PBSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBSGPU_ABI_VERSION 5

/* ---- status codes ------------------------------------------------------- */
#define PBSGPU_OK 0
#define PBSGPU_E_INVALID (-1)      /* bad argument (NULL, not a power of two, ...) */
#define PBSGPU_E_NO_DEVICE (-2)    /* no HIP device / HIP runtime error at init */
#define PBSGPU_E_HIP (-3)          /* a HIP call failed (see pbsgpu_last_hip_error) */
#define PBSGPU_E_NOMEM (-4)        /* host or device allocation failed */
#define PBSGPU_E_CAPACITY (-5)     /* caller's output buffer too small (needed size reported) */
#define PBSGPU_E_BUSY (-6)         /* pbsgpu_submit_*: all in-flight tickets used, collect one first (nothing else returns it) */
#define PBSGPU_E_TICKET (-7)       /* unknown / already collected ticket */
/* (-8 is retired: it was PBSGPU_E_DENSITY until ABI v4 — no byte content fails a call any more: candidate-dense data is
 * resolved exactly by on-demand re-scans, on every path) */
#define PBSGPU_E_STATE (-9)        /* call not valid in the handle's current state */

const char *pbsgpu_strerror(int status);
int pbsgpu_abi_version(void);
/* hipError_t of the most recent failing HIP call on this process (diagnostic). */
int pbsgpu_last_hip_error(void);
/* Number of visible HIP devices (0 when there is no GPU / no driver). */
int pbsgpu_device_count(void);

/* ---- chunker configuration ---------------------------------------------
 * Mirrors buzhash.NewConfig(avgSize int) (buzhash.Config, error) — reference
 * call sites internal/pxarmount/commit_orchestrate.go:143-149 and
 * internal/tapeio/converter.go:248 (both avg = 4 << 20), tests
 * internal/pxarmount/commit_walk_test.go:25,380 (avg = 4096).
 * Derived values follow the published Proxmox/casync chunker: window 64,
 * min = avg/4, max = avg*4, mask = 2*avg-1, break when (h & mask) >= mask-2.
 * The 256-word table is an INPUT (NULL selects the built-in casync table). */
typedef struct pbsgpu_config {
    uint32_t avg;
    uint32_t min;
    uint32_t max;
    uint32_t window;    /* 64 */
    uint32_t mask;      /* break_test_mask; must be 2^k - 1 */
    uint32_t break_min; /* break_test_minimum */
    uint32_t table[256];
} pbsgpu_config;

/* PBSGPU_E_INVALID unless avg is a power of two in [256, 2^28]
 * (the error return of buzhash.NewConfig). */
int pbsgpu_config_init(uint64_t avg, const uint32_t *table, pbsgpu_config *out);
const uint32_t *pbsgpu_default_table(void);

/* ---- records -------------------------------------------------------------
 * One record per chunk = one dynamic-index entry: datastore.ChunkInfo{End,
 * Digest} (internal/pxarmount/commit_reuse.go:105-115) plus the segment it
 * belongs to and its size (backupproxy.KnownChunkRef{Digest, Size},
 * commit_reuse.go:316-332). `end` is exclusive and relative to the start of
 * the record's segment. Records come out ordered by (segment, end). */
typedef struct pbsgpu_record {
    uint64_t end;
    uint8_t digest[32]; /* SHA-256 of the raw chunk bytes (CryptModeNone, commit_orchestrate.go:157) */
    uint32_t segment;
    uint32_t size;
} pbsgpu_record;

/* A segment is one independent stream inside the submitted byte range: fresh
 * chunker state at `offset`, forced cut at `offset + length`. One segment =
 * one archive payload stream, or the bytes between two InjectChunks calls
 * (commit_reuse.go:315-341), or one file of a many-file batch. Segments must
 * be sorted by offset and must not overlap. */
typedef struct pbsgpu_segment {
    uint64_t offset;
    uint64_t length;
} pbsgpu_segment;

/* ---- engine ---------------------------------------------------------------
 * One engine per (process, device): owns HIP streams, device work buffers and
 * pinned staging. Stands where backupproxy's session holds its chunker +
 * hasher (constructed from the Config at commit_orchestrate.go:137-149). */
typedef struct pbsgpu_engine pbsgpu_engine;

/* `device` = HIP ordinal; `inflight` = number of batches that may be in
 * flight at once (1..16; 0 -> default 2). */
int pbsgpu_engine_create(int device, const pbsgpu_config *cfg, uint32_t inflight, pbsgpu_engine **out);
/* ... with every tuning value an engine has (ABI v5). Until v4 these were process-wide PBSGPU_* environment variables,
 * read once — invisible to a host that runs two engines with different needs. 0 / 0.0 = the default everywhere. The
 * variables named in the comments still work as DEBUG overrides of whatever the caller passed (one table in engine.cpp). */
typedef struct pbsgpu_engine_options {
    uint32_t inflight;               /* batches in flight at once (1..16; 0 = 2) */
    uint32_t sha_form;               /* batch-path hash kernel: 0 = wave pairs (default), 1 = single-wave lanes, 2 = express
                                      * (two lanes per chunk). Parity tests and A/B runs; PBSGPU_SHA_MODE=lane|pair|xpair */
    uint32_t sha_slack_pct;          /* workgroup budget of a batch's hash launch over total work / longest chain, in percent
                                      * + 1 (0 = default 25 %; 1 = none); PBSGPU_SHA_SLACK_PCT */
    uint32_t sha_dense_pct;          /* work per pair lane, in percent of the longest chain, from which a batch's hash launch
                                      * takes the dense form (0 = default 150; 0xffffffff = never); PBSGPU_SHA_DENSE_PCT */
    uint64_t resolve_par_min;        /* smallest single stream whose cut chain is resolved by pointer doubling (0 = default
                                      * 64 MiB; UINT64_MAX = always the serial walk); PBSGPU_RESOLVE_PAR_MIN / _SERIAL */
    uint32_t sha_many_files_per_core;/* pbsgpu_sha256_many_pays: files in flight per host core from which a GPU batch wins
                                      * (0 = default 55); PBSGPU_SHA_MANY_FILES_PER_CORE */
    /* the engine's own page ring behind pbsgpu_stream_* (created by the first stream) */
    uint32_t stream_sha_cus;         /* CUs of its pair service (0 = default 32); PBSGPU_STREAM_SHA_CUS */
    uint32_t stream_express_cus;     /* ... of its express service (0 = default 8; 0xffffffff = none); PBSGPU_STREAM_XP_CUS */
    uint32_t stream_ring_slots;      /* ring streams (sections of all payload streams) open at once (0 = 256); PBSGPU_STREAM_RING_SLOTS */
    uint32_t stream_ctx_pool;        /* closed stream contexts kept for re-use (0 = default 8; 0xffffffff = none); PBSGPU_STREAM_CTX_POOL */
    uint32_t reserved0;
    double stream_ring_gib;          /* its arena (0 = default 48 GiB); PBSGPU_STREAM_RING_GIB */
    uint64_t stream_page_bytes;      /* its page size (0 = default); PBSGPU_STREAM_PAGE_BYTES */
    uint64_t reserved[4];
} pbsgpu_engine_options;
int pbsgpu_engine_create_opt(int device, const pbsgpu_config *cfg, const pbsgpu_engine_options *opt /* NULL = defaults */,
                             pbsgpu_engine **out);
void pbsgpu_engine_destroy(pbsgpu_engine *eng);
int pbsgpu_engine_config(const pbsgpu_engine *eng, pbsgpu_config *out);
/* Release what the engine only holds for re-use: the page ring of its payload streams (arena of PBSGPU_STREAM_RING_GIB,
 * default 48 GiB, created by the first pbsgpu_stream_create; released only while no payload stream is alive) and the
 * parked contexts of closed streams (pinned staging). *freed_bytes = device memory that came back. Waits for the device
 * to go idle: call it between jobs. */
int pbsgpu_engine_trim(pbsgpu_engine *eng, uint64_t *freed_bytes);

/* Batch path (the data-parallel form of WriteEntryReader's chunk loop —
 * internal/pxarmount/commit_reuse.go:457, commit_walk.go:475,
 * internal/tapeio/converter.go:836): cut every segment, hash every chunk.
 * `*_device`: bytes already resident in HBM at `dptr` (borrowed until collect).
 * `*_host`: bytes in host memory, copied H2D through pinned staging.
 * Asynchronous: returns a ticket as soon as the work is enqueued. */
int pbsgpu_submit_device(pbsgpu_engine *eng, const void *dptr, uint64_t nbytes,
                         const pbsgpu_segment *segs, uint32_t nseg, uint64_t *ticket);
int pbsgpu_submit_host(pbsgpu_engine *eng, const void *hptr, uint64_t nbytes,
                       const pbsgpu_segment *segs, uint32_t nseg, uint64_t *ticket);
/* The same with SUGGESTED BOUNDARIES (the payload chunker of the Proxmox lineage: cut at a suggested offset — a file
 * start in the .ppxar stream, internal/pxarmount/commit_types.go:24-32, commit_reuse.go:265 — when the open chunk
 * would then be within [min, max]; otherwise keep scanning). suggested[suggested_index[s] .. suggested_index[s+1])
 * are the ascending boundaries of segment s, relative to its start; suggested_index has nseg + 1 entries (2 when
 * segs == NULL). Definition = the serial payload chunker fed byte by byte (oracle_payload_chunker_scan): after a
 * cut at s the next cut is the EARLIER of the hash/max cut and the first suggested b with min <= b - s <= max;
 * boundaries closer than min to s are dropped. Whether github.com/pbs-plus/pxar v0.34.0 cuts at suggested
 * boundaries is open: the plain entry points above never do. */
int pbsgpu_submit_device_suggested(pbsgpu_engine *eng, const void *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
                                   uint32_t nseg, const uint64_t *suggested, const uint32_t *suggested_index,
                                   uint64_t *ticket);
int pbsgpu_submit_host_suggested(pbsgpu_engine *eng, const void *hptr, uint64_t nbytes, const pbsgpu_segment *segs,
                                 uint32_t nseg, const uint64_t *suggested, const uint32_t *suggested_index,
                                 uint64_t *ticket);
/* Which READER the suggested-boundary rule emulates. Upstream's payload chunker looks at its pending boundary once per
 * `scan` call, before the hash scan of the bytes that call brings: a boundary inside the call's buffer is cut at even
 * if a hash boundary lies EARLIER in the same buffer, while a hash boundary found by an earlier call wins. The result
 * therefore depends on how many bytes one call sees. feed_bytes = 1 (the default): byte-serial feed, the
 * feed-independent limit (the earlier position wins); N > 1: every call sees N bytes; 0: one call sees everything
 * that is left. absolute_grid = 0: the buffer grid restarts at every cut (oracle_chunk_stream_suggested's feeding
 * loop); 1: buffers end at multiples of feed_bytes from the stream start (a reader appending fixed-size reads, as a
 * Go io.Reader loop with a fixed buffer does: internal/agent/verification/handler.go:15-20 uses 256 KiB). Applies to
 * everything enqueued afterwards. Whether github.com/pbs-plus/pxar cuts at suggested boundaries at all is open. */
int pbsgpu_engine_set_suggested_feed(pbsgpu_engine *eng, uint64_t feed_bytes, int absolute_grid);
/* Block until the ticket's work is done; report its record count. */
int pbsgpu_wait(pbsgpu_engine *eng, uint64_t ticket, uint64_t *nrecords);
/* Non-blocking: *done = 1 when everything enqueued for the ticket has finished on the device (collect will not
 * wait, except for the rare density retry), 0 while it is still running. Lets a host that keeps several
 * tickets in flight collect whichever finishes first instead of first-in-first-out. */
int pbsgpu_ticket_done(pbsgpu_engine *eng, uint64_t ticket, int *done);
/* Wait, copy the records out (PBSGPU_E_CAPACITY + *nrecords if cap is too
 * small; the ticket stays valid), release the ticket. */
int pbsgpu_collect(pbsgpu_engine *eng, uint64_t ticket, pbsgpu_record *out, uint64_t cap,
                   uint64_t *nrecords);

/* Per-stage device time of a finished ticket, from HIP events recorded on the
 * stream the kernels ran on (milliseconds). */
typedef struct pbsgpu_timing {
    float h2d_ms;     /* host->device staging (0 for *_device submits) */
    float scan_ms;    /* Buzhash candidate kernel */
    float resolve_ms; /* candidate compaction + min/max resolution */
    float sha_ms;     /* SHA-256 kernel */
    float total_ms;   /* first kernel start -> records ready */
    uint64_t ncandidates;
    uint64_t nrecords;
    uint32_t retries; /* density retries taken */
    uint32_t reserved;
} pbsgpu_timing;
int pbsgpu_ticket_timing(pbsgpu_engine *eng, uint64_t ticket, pbsgpu_timing *out);

/* Kernel-level entry used by the parity tests: raw Buzhash candidates of a
 * flat device byte range, i.e. every END offset e (64 <= e <= nbytes) whose
 * 64-byte window [e-64, e) satisfies the break test, ascending. Synchronous. */
int pbsgpu_candidates_device(pbsgpu_engine *eng, const void *dptr, uint64_t nbytes,
                             uint64_t *out, uint64_t cap, uint64_t *n);

/* The resolve step alone: cut ONE stream of `stream_len` bytes from an explicit ascending list of
 * candidate END offsets (as returned by pbsgpu_candidates_device, in stream coordinates). Used when
 * one stream's candidates were found piecewise — e.g. a stream split over several GPUs, each scanning
 * its byte range and all-gathering the lists. Records come back without digests. */
int pbsgpu_resolve_candidates(pbsgpu_engine *eng, const uint64_t *cands, uint64_t ncand, uint64_t stream_len,
                              pbsgpu_record *out, uint64_t cap, uint64_t *nrecords);

/* ---- upstream-style streaming chunker --------------------------------------
 * `scan` semantics of the chunker behind buzhash.Config (Proxmox
 * ChunkerImpl::scan): consume `len` bytes; *pos = 0 when no boundary was
 * found (everything consumed), else the boundary is after data[*pos - 1] and
 * only *pos bytes were consumed (state reset). Compatibility path: the scan
 * itself runs on the GPU but is only efficient for large buffers. */
typedef struct pbsgpu_chunker pbsgpu_chunker;
int pbsgpu_chunker_create(pbsgpu_engine *eng, pbsgpu_chunker **out);
void pbsgpu_chunker_destroy(pbsgpu_chunker *c);
int pbsgpu_chunker_scan(pbsgpu_chunker *c, const void *data, size_t len, size_t *pos);
int pbsgpu_chunker_reset(pbsgpu_chunker *c);

/* ---- payload-stream writer ---------------------------------------------------
 * The seam WriteEntryReader feeds (transfer.ArchiveWriter, mocked at
 * internal/pxarmount/commit_test.go:33-67): bytes are appended to ONE
 * continuous stream; finished (end, digest) records become available as the
 * stream advances. `end` in the records is the absolute stream offset
 * (payload position: written + injected), `segment` the section (number of
 * pbsgpu_stream_cut calls before the chunk).
 * All payload streams of an engine are clients of ONE engine-owned page ring
 * (see "page ring" below): host bytes are staged in pinned memory and copied
 * straight into a reserved page; cut rounds, the persistent SHA-256 service, page-granular
 * release and record delivery are shared. `window_bytes` is accepted for
 * compatibility and ignored (rounds 1-3: bytes per private device window).
 * No byte content makes a call fail (the reference's WriteEntryReader has no content-dependent
 * error either: internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836): data
 * with more candidates than a scan tile has slots (periodic / crafted: one per 128 bytes and
 * more) is resolved exactly by on-demand re-scans inside the cut round — slower for that
 * stretch, bit-identical to the serial chunker. */
typedef struct pbsgpu_stream pbsgpu_stream;
int pbsgpu_stream_create(pbsgpu_engine *eng, uint64_t window_bytes, pbsgpu_stream **out);
void pbsgpu_stream_destroy(pbsgpu_stream *s);
int pbsgpu_stream_write(pbsgpu_stream *s, const void *data, size_t len);
/* Zero-copy feed: borrow library-owned pinned memory, fill it (e.g. io.ReadFull straight from the
 * source file, the reader side of WriteEntryReader), then commit the first `len` bytes. Saves the
 * caller-buffer -> staging copy of pbsgpu_stream_write. One reservation at a time. */
int pbsgpu_stream_reserve(pbsgpu_stream *s, void **buf, size_t *cap);
int pbsgpu_stream_commit(pbsgpu_stream *s, size_t len);
/* Force a cut at the current position (InjectChunks flushes the open chunk:
 * commit_reuse.go:315-341) and skip `inject_bytes` of injected, already
 * known chunk payload in the stream offsets. */
int pbsgpu_stream_cut(pbsgpu_stream *s, uint64_t inject_bytes);
/* End of stream: the tail becomes the final chunk; returns when every record is available to poll (the serial SHA-256
 * chain of the last chunks: up to ~0.5 s at 16 MiB maximum chunks). */
int pbsgpu_stream_finish(pbsgpu_stream *s);
/* The same without the wait, for a writer that goes on with its NEXT archive while this one drains (one goroutine per
 * archive, archives back to back: internal/tapeio/converter.go:672-680): finish_begin closes the input (tail chunk cut,
 * every chunk handed to the hash jobs; later writes fail with PBSGPU_E_STATE) and returns; records keep arriving
 * through poll; pbsgpu_stream_done is non-blocking, *done = 1 once the last record can be polled. pbsgpu_stream_finish
 * after finish_begin waits for exactly that. */
int pbsgpu_stream_finish_begin(pbsgpu_stream *s);
int pbsgpu_stream_done(pbsgpu_stream *s, int *done);
/* Pop up to `cap` finished records (in stream order). */
int pbsgpu_stream_poll(pbsgpu_stream *s, pbsgpu_record *out, uint64_t cap, uint64_t *n);
/* Encoder().PayloadPosition() (commit_reuse.go:265): bytes written PLUS bytes injected so far — the coordinate
 * system of the records' `end` and of PAYLOAD_REF offsets (InjectChunks advances the position by the injected
 * sizes: keepLast_chunk_test.go mock, enc.Advance(total)). */
int pbsgpu_stream_position(const pbsgpu_stream *s, uint64_t *position);
/* Bytes handed to write/commit only (no injected bytes). */
int pbsgpu_stream_bytes_written(const pbsgpu_stream *s, uint64_t *bytes_written);
/* Suggest a chunk boundary at absolute payload position `offset` (same coordinates as pbsgpu_stream_position;
 * typically the current position = "a file starts here"). Ascending; see pbsgpu_submit_device_suggested. */
int pbsgpu_stream_suggest(pbsgpu_stream *s, uint64_t offset);

/* Per-file XXH3-64 tee of the stream (writeBackedFile: hash := xxh3.New(); tee := io.TeeReader(f, hash);
 * writer.WriteEntryReader(entry, tee, size); backedHashes[path] = hash.Sum64() — internal/pxarmount/
 * commit_reuse.go:450-461): the bytes written between begin_file and end_file are hashed ON THE DEVICE from the
 * same window the chunker reads (one H2D copy, two consumers; a file may span any number of windows). The hash is
 * reported asynchronously — with the window that holds the file's last byte — through poll_files, in file order
 * (the reference only reads backedHashes after the commit: verifyBackedFileHashes, commit_orchestrate.go:485-562). */
typedef struct pbsgpu_file_hash {
    uint64_t index; /* as returned by end_file / end_entry (0, 1, 2, ... per stream) */
    uint64_t size;
    uint64_t xxh3;  /* XXH3-64, seed 0 */
} pbsgpu_file_hash;
int pbsgpu_stream_begin_file(pbsgpu_stream *s);
int pbsgpu_stream_end_file(pbsgpu_stream *s, uint64_t *file_index);
int pbsgpu_stream_poll_files(pbsgpu_stream *s, pbsgpu_file_hash *out, uint64_t cap, uint64_t *n);
/* pxar payload entries written straight into the stream (the layout pbsgpu_payload_pack_device produces for resident
 * files): begin_entry appends the 16-byte {payload_type, 16 + content_len} header, reports the header's payload
 * position (what WriteEntryRef / PAYLOAD_REF records, commit_walk.go:455) and opens the file tee; exactly
 * content_len bytes must follow (write / reserve+commit) before end_entry (else PBSGPU_E_STATE = the Go writer's
 * unexpected EOF). write_marker appends the start (tail = 0) or tail (tail = 1) marker. fmt NULL = defaults. */
struct pbsgpu_payload_format;
int pbsgpu_stream_begin_entry(pbsgpu_stream *s, const struct pbsgpu_payload_format *fmt, uint64_t content_len,
                              uint64_t *payload_offset);
int pbsgpu_stream_end_entry(pbsgpu_stream *s, uint64_t *file_index);
int pbsgpu_stream_write_marker(pbsgpu_stream *s, const struct pbsgpu_payload_format *fmt, int tail);

/* ---- page ring: many payload streams, page-granular memory release, persistent SHA-256 service ------------------
 * The data-parallel form of the chunk loop behind WriteEntryReader for SEVERAL archives at once (one writer per
 * archive: internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836) when the bytes are (or arrive)
 * in device memory. A batch submitted with pbsgpu_submit_device stays resident until its LONGEST chunk is hashed
 * (SHA-256 is serial inside a chunk: up to ~0.45 s for 16 MiB); the ring gives memory back PAGE by page:
 *   - the arena is cut into pages (>= max chunk size); a stream = pages in logical order, anywhere in the arena;
 *   - reserve / commit (or fill, the synthetic producer) hand the stream's next page to the ring; pump cuts the newly
 *     committed pages of all streams in one round on the device and feeds every cut chunk to the SHA-256 service, a
 *     persistent kernel whose lanes take the next chunk the moment they finish one;
 *   - a page is free again as soon as every chunk touching it has been READ by the service (not when a batch ends);
 *   - poll returns the stream's finished (end, digest) records in stream order; `end` is the absolute stream offset.
 * Results are bit-identical to one pbsgpu_submit_* / pbsgpu_stream_* pass over the same bytes.
 * One thread drives a ring. While the service runs, hipDeviceSynchronize / hipFree of the process block (the library's own
 * frees are parked meanwhile): call quiesce or park first. A ring that is not called at all for
 * PBSGPU_RING_IDLE_TIMEOUT_S (default 20 s) while its service has nothing to do — a writer in a blocking tape read,
 * internal/tapeio/converter.go:672-680 — loses NOTHING: the service stops on its own (a handshake with the host
 * guarantees that no chunk is left behind) and the next pump starts it again.
 * Candidate-dense data (more than one candidate per 128 bytes over a whole scan tile: a crafted short period) is cut
 * exactly like everything else: the control kernel re-scans such a tile on demand for the one candidate the cut rule needs
 * (ABI v4 failed the stream with PBSGPU_E_DENSITY). A stream is limited to 4 PiB (52-bit offsets inside a round, 12 bits
 * for the slot); PBSGPU_E_INVALID beyond. */
typedef struct pbsgpu_ring pbsgpu_ring;
typedef struct pbsgpu_ring_options {
    uint64_t arena_bytes;  /* device memory for pages; 0 = what is free minus 8 GiB */
    uint64_t page_bytes;   /* 0 = default: max chunk rounded up to whole scan tiles (16.2 MiB at avg 4 MiB) */
    uint32_t max_streams;  /* streams open at once; 0 = 64 */
    uint32_t sha_cus;      /* CUs of the SHA-256 service; 0 = three quarters of the chip (the rest runs the cut rounds) */
    uint32_t round_pages;  /* most pages one round cuts; 0 = 256 */
    uint32_t express_cus;  /* CUs of the EXPRESS service (two lanes per chunk: the SHA-256 chain of a chunk runs ~1.4x faster at
                            * ~0.65 of the throughput per CU) for the longest chunks (>= 13/16 of the maximum size: 1.3 % of
                            * the chunks of random data; PBSGPU_RING_LONG_BYTES). 0 = the default: 16 CUs out of the default
                            * service share when sha_cus is 0 too (measured: throughput unchanged, a lone 64 GiB file 15 %
                            * sooner, the drain of a burst 0.08 s shorter), none when sha_cus is given. A value given here
                            * comes ON TOP of a given sha_cus. PBSGPU_RING_XP_CUS overrides (0 = off). (`reserved` before.) */
    /* ---- ABI v5: what used to be PBSGPU_RING_* environment variables (still honoured as debug overrides: one table in
     * ring.cpp). 0 / 0.0 = the default; PBSGPU_RING_OFF where "none" must be expressible. ---- */
    uint32_t min_round_pages;  /* a round is launched as soon as this many pages wait (0 = round_pages / 4) */
    uint32_t max_inflight;     /* rounds in flight (0 = 3) */
    uint32_t long_bytes;       /* chunk size from which a chunk takes the express service (0 = 13/16 of the maximum) */
    uint32_t long_lo_bytes;    /* ... while the express pairs are mostly idle (0 = 11/16 of the maximum on bulk rings; OFF = never) */
    uint32_t long_spill;       /* long chunks waiting beyond which the pair lanes help (0 = default rule) */
    uint32_t poll_every;       /* block steps between two looks at the queue of a service wave that carries work (0 = 8..64 by chunker) */
    uint32_t flags;            /* PBSGPU_RING_F_* */
    uint32_t lanes_cus;        /* CUs of the LANES service (one lane per chunk, four independent waves per CU: ~1.1x the bytes per
                                * CU-second of the pair service, every chain ~1.8x slower) for chunks of at most short_bytes, taken
                                * out of sha_cus' share; 0 = none (`reserved0` before). PBSGPU_RING_LANES_CUS overrides. */
    double backlog_mib;        /* bytes waiting in front of the service beyond which no page is handed out (0 = 128 MiB per
                                * service CU; < 0 = no limit) */
    double lone_defer_ms;      /* how long the first rounds of bulk streams on an idle ring are cut ahead of the service (0 = 25; < 0 = never) */
    double idle_timeout_s;     /* the service stops on its own after this long without work or a call (0 = 20 s) */
    double autopark_ms;        /* > 0: the ring parks its service when nothing has been anywhere in it for this long */
    uint64_t short_bytes;      /* largest chunk the lanes service takes (0 = 3/2 of the average chunk size; first word of `reserved` before) */
    uint64_t reserved[3];
} pbsgpu_ring_options;
#define PBSGPU_RING_OFF 0xffffffffu
#define PBSGPU_RING_F_NO_OVERLAP 1u      /* scan of round n + 1 NOT beside the control kernel of round n (one scan set) */
#define PBSGPU_RING_F_NO_STAGE 2u        /* the round's kernels read its tables from mapped host memory (rounds 3-4) */
#define PBSGPU_RING_F_NO_CUT_PRIO 4u     /* refill and scan streams at normal priority */
#define PBSGPU_RING_F_NO_SPLIT_AUTO 8u   /* the pair / express CU split does not follow the data */
#define PBSGPU_RING_F_DEFER_SERVICE 16u  /* profiling: rounds only fill the queue, quiesce runs the services alone */
#define PBSGPU_RING_F_FILL_SERIAL 32u    /* experiments: the synthetic refill in stream order behind the previous scan */
#define PBSGPU_RING_F_DENSE_SERVICE 64u  /* the pair service with FOUR pairs (eight waves) per CU: more bytes per CU-second, every chain slower */
#define PBSGPU_RING_F_DENSE_LANES 128u   /* the lanes service (lanes_cus) with EIGHT waves per CU: 84 instead of 77-81 chain-blocks per us and CU, 6 us per block */
#define PBSGPU_RING_F_TIER_TAG 256u      /* diagnostics: pbsgpu_ring_poll* report the queue a chunk went through in bits 28-29 of `segment` (0 main, 1 long, 2 short) */
typedef struct pbsgpu_ring_stats {
    uint64_t page_bytes, bytes_enqueued, chunks, candidates, pages_enqueued, pages_recycled, service_bytes_last;
    uint32_t pages_total, pages_free, sha_cus, rounds, rounds_done, rounds_in_flight, streams_opened, service_launches;
    double service_ms_last;  /* duration of the most recent service launch (HIP events on its stream), set by quiesce */
    double service_ms_total;
} pbsgpu_ring_stats;
int pbsgpu_ring_create(pbsgpu_engine *eng, const pbsgpu_ring_options *opt /* NULL = defaults */, pbsgpu_ring **out);
void pbsgpu_ring_destroy(pbsgpu_ring *ring);
/* A new stream (fresh chunker state). PBSGPU_E_BUSY when max_streams are open. */
int pbsgpu_ring_open(pbsgpu_ring *ring, uint32_t *stream);
/* The stream's next page: a device pointer to write up to *cap (= page size) bytes to — by a kernel, a DMA, a peer.
 * PBSGPU_E_BUSY when no page is free right now, or when enough bytes already wait in front of the SHA-256 service
 * (committed pages not yet in a round + rounds in flight + published chunks no lane has claimed > the backlog limit,
 * default 128 MiB per service CU, PBSGPU_RING_BACKLOG_MIB, 0 = no limit): pump / poll and retry. */
int pbsgpu_ring_reserve(pbsgpu_ring *ring, uint32_t stream, void **dptr, uint64_t *cap);
/* The first nbytes of the reserved page are the stream's next bytes and are VISIBLE to the device (the producer has
 * finished). Every page but the stream's last must be full; final != 0 ends the stream (nbytes may then be 0, also
 * without a reservation). */
int pbsgpu_ring_commit(pbsgpu_ring *ring, uint32_t stream, uint64_t nbytes, int final);
/* Synthetic producer (benchmarks, parity tests): the stream's next nbytes come from the generator of
 * pbsgpu_fill_device (seed, kind) at the stream's current offset, written by the round itself. Takes as many whole
 * pages as are free: *taken bytes were accepted, call again with the rest after a pump. */
int pbsgpu_ring_fill(pbsgpu_ring *ring, uint32_t stream, uint64_t seed, uint32_t kind, uint64_t nbytes, int final,
                     uint64_t *taken);
/* Synthetic producer for EDITED streams (benchmarks: BASELINE.json configs[4], the re-chunk after byte edits, through the
 * ring): the stream's bytes are a piece table over generator 4 of pbsgpu_fill_device — kept extents of a base file and
 * newly written extents in stream order, pieces contiguous from 0, every offset and length a multiple of 16. The first
 * call hands the table over (copied), later calls pass NULL / 0 and continue; nbytes / final / *taken as pbsgpu_ring_fill. */
typedef struct pbsgpu_fill_piece {
    uint64_t dst_off; /* offset in the stream */
    uint64_t len;
    uint64_t src_off; /* generator offset the piece's first byte comes from */
    uint64_t seed;    /* generator seed (the base file's, or the new bytes') */
} pbsgpu_fill_piece;
int pbsgpu_ring_fill_pieces(pbsgpu_ring *ring, uint32_t stream, const pbsgpu_fill_piece *pieces, uint32_t npieces,
                            uint64_t nbytes, int final, uint64_t *taken);
/* Enqueue the committed pages as cut rounds, collect finished rounds and freed pages. Never blocks. */
int pbsgpu_ring_pump(pbsgpu_ring *ring);
/* Up to cap finished records of the stream, in stream order; *finished = 1 once the stream has ended and every record
 * has been handed out. */
int pbsgpu_ring_poll(pbsgpu_ring *ring, uint32_t stream, pbsgpu_record *out, uint64_t cap, uint64_t *n, int *finished);
/* The same for ANY open stream in one call (each stream's records in its own order, `segment` = stream id): for callers
 * that keep hundreds or thousands of short streams open — one per file of a many-file job (BASELINE.json configs[2]) —
 * and cannot ask each of them after every pump. Streams that have ended and handed out their last record are listed
 * once in `finished` (up to fcap per call); close them afterwards. */
int pbsgpu_ring_poll_any(pbsgpu_ring *ring, pbsgpu_record *out, uint64_t cap, uint64_t *n, uint32_t *finished, uint32_t fcap,
                         uint32_t *nfinished);
/* Release a finished, fully polled stream's slot (PBSGPU_E_STATE before that). */
int pbsgpu_ring_close(pbsgpu_ring *ring, uint32_t stream);
/* Suggested boundary (see pbsgpu_submit_device_suggested) at `offset` bytes from the stream's start; ascending; announce
 * it before the bytes around it are committed. The reader-buffer rule of pbsgpu_engine_set_suggested_feed applies. */
int pbsgpu_ring_suggest(pbsgpu_ring *ring, uint32_t stream, uint64_t offset);
/* Wait until everything enqueued is hashed and stop the service kernel (the device is then idle as far as the ring is
 * concerned); the next pump starts it again. */
int pbsgpu_ring_quiesce(pbsgpu_ring *ring);
/* The same without the wait: the service ends by itself once it has hashed what is enqueued; the next pump starts a new
 * one. For a binding that is about to sit in a blocking read, or that wants hipFree / device-wide synchronisation of the
 * process to be possible again soon. */
int pbsgpu_ring_park(pbsgpu_ring *ring);
int pbsgpu_ring_get_stats(pbsgpu_ring *ring, pbsgpu_ring_stats *out);
/* How the ring's service is laid out: CUs of the express service (0 = none) and the chunk size from which a chunk takes it. */
int pbsgpu_ring_express(pbsgpu_ring *ring, uint32_t *express_cus, uint64_t *long_bytes);
/* Which regime is the ring running in? Cumulative counters of the two SHA-256 services since pbsgpu_ring_create: one
 * wave of each service samples the shader clock and the 100 MHz wall clock once per 4096 block steps and, when it carried a
 * block in every step of the interval, adds the interval to these sums. ticks / steps x 10 = ns per block step of a chain
 * UNDER LOAD (an express step is two blocks of a chunk); cycles / ticks x 100 = the shader clock in MHz that chain ran at.
 * Read it twice and subtract to look at a phase (bench.py: feed phase and drain of the timed region -> `roofline`).
 * Safe while the service runs: the answer then comes from the newest finished cut round's status block (the control kernel
 * copies the counters there, at most one round old) and no HIP call is made beside the persistent kernels; with the service
 * stopped (after pbsgpu_ring_quiesce) the device counters are read directly. Same thread rule as every pbsgpu_ring_* call. */
typedef struct pbsgpu_ring_probe {
    uint64_t pair_steps, pair_cycles, pair_ticks;
    uint64_t express_steps, express_cycles, express_ticks;
} pbsgpu_ring_probe;
int pbsgpu_ring_get_probe(pbsgpu_ring *ring, pbsgpu_ring_probe *out);
/* Diagnostic text snapshot of the ring's device-side state (queue words, page reference counts, stream states). Debugging
 * only: it copies from the device on the null stream, which may wait for a running service's idle time-out. */
int pbsgpu_ring_debug(pbsgpu_ring *ring, char *buf, uint64_t cap);

/* ---- whole-stream SHA-256 batch ---------------------------------------------
 * verification.HashFile (internal/agent/verification/handler.go:36-68) and
 * extractFileHash (internal/server/verification/job.go:1273-1303) for many
 * files at once: digest[i] = SHA-256(base[segs[i].offset .. +length)). */
int pbsgpu_sha256_many_device(pbsgpu_engine *eng, const void *dptr, uint64_t nbytes,
                              const pbsgpu_segment *segs, uint32_t nseg, uint8_t *digests /* 32*nseg */);
int pbsgpu_sha256_many_host(pbsgpu_engine *eng, const void *hptr, uint64_t nbytes,
                            const pbsgpu_segment *segs, uint32_t nseg, uint8_t *digests);
/* POLICY for the two calls above. SHA-256 is serial inside a file: the GPU hashes ONE file per lane at 0.036 GiB/s
 * whatever its size, a SHA-NI host core does ~2 GiB/s, so a batch only wins with more than ~55 files in flight per
 * host core that would otherwise hash (measured). The reference's verify job keeps FOUR files in
 * flight (internal/server/verification/job.go:493): routed to the GPU it would run ~50x slower than sha256-simd.
 * *pays = 1 when a batch of `nfiles` beats `host_cores` (0 = 1) host cores, else 0 — a binding keeps the host hash
 * when it says 0 (go/pbsgpu: Engine.HashFiles returns ErrHostFaster). The hash calls themselves never refuse. */
int pbsgpu_sha256_many_pays(const pbsgpu_engine *eng, uint32_t nfiles, uint32_t host_cores, int *pays);

/* ---- whole-stream XXH3-64 batch ---------------------------------------------------
 * The per-file hash the commit path tees new file bodies through and re-checks afterwards:
 * xxh3.New() ... Sum64() (internal/pxarmount/commit_reuse.go:450-461), verifyBackedFileHashes
 * (internal/pxarmount/commit_orchestrate.go:485-562). out[i] = XXH3-64(seed 0) of segs[i]. */
int pbsgpu_xxh3_many_device(pbsgpu_engine *eng, const void *dptr, uint64_t nbytes,
                            const pbsgpu_segment *segs, uint32_t nseg, uint64_t *out /* nseg */);
int pbsgpu_xxh3_many_host(pbsgpu_engine *eng, const void *hptr, uint64_t nbytes,
                          const pbsgpu_segment *segs, uint32_t nseg, uint64_t *out);

/* ---- digest-set operations (cross-file duplicate detection) -----------------
 * Sort records by digest on the device and flag duplicates: dup[i] = 1 when
 * an earlier record (lower index) carries the same digest. Used on the
 * all-gathered (digest, size) set of all ranks. */
typedef struct pbsgpu_dedup_stats {
    uint64_t nrecords;
    uint64_t nunique;
    uint64_t total_bytes;
    uint64_t unique_bytes;
} pbsgpu_dedup_stats;
int pbsgpu_dedup_host(pbsgpu_engine *eng, const pbsgpu_record *recs, uint64_t n, uint8_t *dup /* n, may be NULL */,
                      pbsgpu_dedup_stats *stats);
/* The same on records that already are in device memory (the receive buffer of the RCCL all-gather): the gathered set
 * never takes a host round trip. `dup` (host, may be NULL) and `stats` as above. */
int pbsgpu_dedup_device(pbsgpu_engine *eng, const void *drecs, uint64_t n, uint8_t *dup /* n, may be NULL */,
                        pbsgpu_dedup_stats *stats);

/* ---- multi-GPU digest-set reduce (RCCL over xGMI) ----------------------------------
 * The path shards at file / archive granularity with no data-path collective: one engine per GPU (one process per GPU, or
 * several engines in one process) ingests its own streams. The one exchange step is the digest-set reduce for cross-file
 * duplicate detection: ONE RCCL all-gather of fixed-size slots [count | records] + the device dedup.
 * A Go host (one session per process, pure Go: internal/tapeio/converter.go:396-439) binds this directly — no Python, no
 * torch: rank 0 asks for a unique id, ships its 128 bytes to the other ranks over the channel it already has (the
 * agent's aRPC session), every rank creates its communicator, and every rank calls the reduce in the same order.
 * RCCL is resolved at run time (dlopen): a single-GPU host never loads it; PBSGPU_E_NO_DEVICE when it is not installed. */
#define PBSGPU_COMM_ID_BYTES 128
typedef struct pbsgpu_comm pbsgpu_comm;
int pbsgpu_comm_unique_id(uint8_t id[PBSGPU_COMM_ID_BYTES]);
/* Collective over `world` ranks (blocks until all of them have called it). The communicator keeps its engine alive. */
int pbsgpu_comm_create(pbsgpu_engine *eng, const uint8_t id[PBSGPU_COMM_ID_BYTES], int rank, int world, pbsgpu_comm **out);
void pbsgpu_comm_destroy(pbsgpu_comm *comm);
int pbsgpu_comm_rank(const pbsgpu_comm *comm, int *rank, int *world);
/* What RCCL reported when a pbsgpu_comm_* call of this process last failed ("" = nothing yet), e.g.
 * "ncclCommInitRank: unhandled system error (ncclResult 2)". The pointer stays valid for the calling thread. */
const char *pbsgpu_comm_last_error(void);
/* Collective. recs[0..n) (HOST or DEVICE memory — device records travel device to device —, n <= cap_records) = this rank's
 * records; cap_records must be the SAME on every
 * rank (bytes per rank / minimum chunk size is a bound every rank can compute). *stats describes the union over all
 * ranks and is identical on every rank; dup_own[i] (may be NULL) = 1 when an earlier record of the union — a lower rank's,
 * or this rank's with a lower index — carries the same digest. ~48 B per chunk travel: 12 MB per TiB of corpus.
 * Errors are collective too: bad arguments (n > cap_records, recs NULL with n > 0), capacities that differ between ranks or
 * an allocation that fails on ONE rank make EVERY rank return that error from this call — the ranks agree over two 64-byte
 * all-gathers before the large one, so no rank is left waiting inside it. (Only comm / stats NULL fail locally.) */
int pbsgpu_digest_allgather_dedup(pbsgpu_comm *comm, const pbsgpu_record *recs, uint64_t n, uint64_t cap_records,
                                  uint8_t *dup_own /* n, may be NULL */, pbsgpu_dedup_stats *stats);
/* One stream split over the ranks (BASELINE.json configs[1] at N > 1 when the stream is larger than one GPU's memory;
 * rounds 2-5 had this in Python only). pbsgpu_split_plan is arithmetic: rank `rank` of `world` OWNS [*own_start, *own_end)
 * and must HOLD [*lo, *hi) in device memory — 63 bytes of window halo to the left, one maximum chunk to the right.
 * pbsgpu_comm_split_stream is collective: `local` = device pointer to this rank's bytes [lo, hi); every rank scans its own
 * bytes, the candidate END offsets and later the digests are all-gathered (two small exchanges, errors collective as
 * above), the cut chain is resolved identically everywhere, each rank hashes the chunks that START in its range.
 * out[0 .. *nrecords) = the WHOLE stream's records in stream order, identical on every rank. */
int pbsgpu_split_plan(uint64_t total_len, int world, int rank, uint32_t max_chunk, uint64_t *own_start, uint64_t *own_end,
                      uint64_t *lo, uint64_t *hi);
int pbsgpu_comm_split_stream(pbsgpu_comm *comm, const void *local, uint64_t total_len, pbsgpu_record *out, uint64_t cap,
                             uint64_t *nrecords);

/* ---- dynamic index (.didx) encoding ------------------------------------------
 * On-disk form of the record list: datastore.NewDynamicIndexWriter(ctime)
 * .Add(end, digest).Finish() / datastore.ParseDynamicIndex
 * (internal/pxarmount/commit_bottleneck_test.go:773-793,
 * commit_orchestrate.go:219). 4096-byte header + 40-byte entries. */
#define PBSGPU_DIDX_HEADER_SIZE 4096u
int pbsgpu_didx_size(uint64_t nrecords, uint64_t *nbytes);
int pbsgpu_didx_encode(pbsgpu_engine *eng, const pbsgpu_record *recs, uint64_t n, const uint8_t uuid[16],
                       int64_t ctime, uint8_t *out, uint64_t cap);
int pbsgpu_didx_decode(const uint8_t *in, uint64_t nbytes, pbsgpu_record *out, uint64_t cap, uint64_t *n,
                       int64_t *ctime, uint8_t index_csum[32]);

/* ---- payload-stream assembly ----------------------------------------------------
 * The step before the chunker: the .ppxar payload stream WriteEntryReader appends to
 * (one continuous stream per archive, files separated by format.HeaderSize = 16-byte
 * headers: internal/pxarmount/commit_types.go:24-32 rangeEnd = offset + FileSize +
 * format.HeaderSize; keepLast_chunk_test.go:105). Layout written to `dst`:
 *   [start marker {start_type, 16}]  { [{payload_type, 16 + len_i}] [file i bytes] }*  [tail {tail_type, 16}]
 * Each header is {type u64 LE, full size u64 LE}. The three type constants default to the
 * pxar v2 values (EXTERNAL, injectable). payload_offsets[i] = stream offset of file i's
 * header = what WriteEntryRef / PAYLOAD_REF records (commit_walk.go:455). */
typedef struct pbsgpu_payload_format {
    uint64_t payload_type;  /* PXAR_PAYLOAD */
    uint64_t start_type;    /* PXAR_PAYLOAD_START_MARKER */
    uint64_t tail_type;     /* PXAR_PAYLOAD_TAIL_MARKER */
    uint32_t with_start;    /* emit the start marker */
    uint32_t with_tail;     /* emit the tail marker */
} pbsgpu_payload_format;
int pbsgpu_payload_format_default(pbsgpu_payload_format *out);
int pbsgpu_payload_size(const pbsgpu_segment *files, uint32_t nfiles, const pbsgpu_payload_format *fmt,
                        uint64_t *nbytes);
int pbsgpu_payload_pack_device(pbsgpu_engine *eng, const void *src, uint64_t src_bytes, const pbsgpu_segment *files,
                               uint32_t nfiles, const pbsgpu_payload_format *fmt, void *dst, uint64_t dst_cap,
                               uint64_t *out_len, uint64_t *payload_offsets /* nfiles or NULL */);

/* ---- chunk-reuse planner (host-side index arithmetic) --------------------------------
 * What the commit walk does with the PREVIOUS snapshot's dynamic index before it decides
 * to splice old chunks in (InjectChunks) or to re-feed the bytes to the chunker:
 *   lookupDynamicEntries(idx, rangeStart, rangeEnd) -> chunks, startPadding, endPadding
 *     (internal/pxarmount/commit_reuse.go:84-135; table test commit_bottleneck_test.go:795-835)
 *   shouldReuse: padding / (range + padding) <= chunkPaddingThreshold = 0.1
 *     (commit_reuse.go:152-183, commit_types.go:14; test commit_bottleneck_test.go:837-895)
 * `idx` is the record list of one stream (ascending `end`, e.g. from pbsgpu_didx_decode). */
typedef struct pbsgpu_reuse_chunk {
    uint64_t size;       /* chunk length */
    uint64_t padding;    /* bytes of the chunk outside the requested range */
    uint64_t end_offset; /* chunk end in the old payload stream */
    uint8_t digest[32];
} pbsgpu_reuse_chunk;
int pbsgpu_reuse_lookup(const pbsgpu_record *idx, uint64_t n, uint64_t range_start, uint64_t range_end,
                        pbsgpu_reuse_chunk *out, uint64_t cap, uint64_t *nchunks, uint64_t *start_padding,
                        uint64_t *end_padding);
/* saved = the chunk kept from the previous batch (keepLastChunk), or NULL. *reuse = 1/0. */
int pbsgpu_reuse_should(const pbsgpu_record *idx, uint64_t n, uint64_t range_start, uint64_t range_end,
                        const pbsgpu_reuse_chunk *saved, double threshold, int *reuse);

/* ---- synthetic corpus generator ----------------------------------------------
 * Fills device memory with the deterministic byte stream the benchmarks and
 * parity tests use (the oracle has the CPU twin). kind: 0 random, 1 zeros,
 * 2 repeating 4 KiB block, 3 random with ~30 % zero extents, 4 random from a cheap ARX generator
 * (two ChaCha quarter-rounds per 16-byte block; the page ring's refill). `stream_off`
 * and `dptr` must be 8-byte aligned. */
int pbsgpu_fill_device(pbsgpu_engine *eng, void *dptr, uint64_t stream_off, uint64_t nbytes, uint64_t seed,
                       uint32_t kind);

/* Piece-table copy inside device memory: dst[dst_off .. +len) = src[src_off .. +len) for every item (items may be
 * given in any order; destination ranges must not overlap). Builds the edited corpus of BASELINE.json configs[4]
 * (overwrite / insert / delete extents applied to a resident base corpus) without a host round trip. */
typedef struct pbsgpu_copy_item {
    uint64_t src_off;
    uint64_t dst_off;
    uint64_t len;
} pbsgpu_copy_item;
int pbsgpu_gather_device(pbsgpu_engine *eng, const void *src, uint64_t src_bytes, void *dst, uint64_t dst_bytes,
                         const pbsgpu_copy_item *items, uint32_t nitems);

/* Host -> device copy rate of this box through pinned memory, in GB/s (what bounds every host-fed entry point). */
int pbsgpu_measure_h2d(pbsgpu_engine *eng, uint64_t nbytes, double *gb_per_s);

/* Library-owned device buffers for callers without their own allocator. */
int pbsgpu_device_alloc(pbsgpu_engine *eng, uint64_t nbytes, void **dptr);
int pbsgpu_device_free(pbsgpu_engine *eng, void *dptr);
int pbsgpu_memcpy_h2d(pbsgpu_engine *eng, void *dptr, const void *hptr, uint64_t nbytes);
int pbsgpu_memcpy_d2h(pbsgpu_engine *eng, void *hptr, const void *dptr, uint64_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* PBSGPU_H */
