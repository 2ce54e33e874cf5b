/*
 * oracle/oracle.h — CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity checker for the MI355X engine. Nothing under
 * pbs_plus_amd/ (the product) may include, link or call it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the algorithm restated here lives in the un-vendored Go
 * module github.com/pbs-plus/pxar v0.34.0 (reference go.mod:30, go.sum:168-169),
 * packages buzhash / backupproxy, which is NOT present under /root/reference
 * and cannot be fetched or compiled here (no Go toolchain, no network).  The
 * reference tree pins no chunk boundary and no chunk digest in any test
 * (SURVEY.md §8c).  What IS pinned: SHA-256 against FIPS 180-4 known answers
 * and Python hashlib; the Buzhash chunker against its own streaming/whole
 * buffer equivalence (the invariant upstream's own test checks).
 *
 * Reference call sites this restates the callee of:
 *   internal/pxarmount/commit_orchestrate.go:143-149  buzhash.NewConfig(4 << 20)
 *   internal/tapeio/converter.go:248                  buzhash.NewConfig(4 << 20)
 *   internal/pxarmount/commit_walk_test.go:25,380     buzhash.NewConfig(4096)
 *   internal/pxarmount/commit_reuse.go:105-115        datastore.ChunkInfo{End, Digest}
 *   internal/agent/verification/handler.go:36-68      whole-file SHA-256
 */
#ifndef PBS_ORACLE_H
#define PBS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_WINDOW 64u

/* Parameters of the Buzhash CDC (published Proxmox/casync lineage:
 * window 64, min = avg/4, max = avg*4, mask = 2*avg-1, break_min = mask-2). */
typedef struct oracle_config {
    uint32_t avg;
    uint32_t min;
    uint32_t max;
    uint32_t window;     /* always 64 */
    uint32_t mask;       /* break_test_mask */
    uint32_t break_min;  /* break_test_minimum */
    uint32_t table[256];
} oracle_config;

/* Streaming chunker state (one per stream; reset at every cut). */
typedef struct oracle_chunker {
    oracle_config cfg;
    uint32_t h;
    uint32_t window_size;
    uint64_t chunk_size;
    uint8_t window[ORACLE_WINDOW];
} oracle_chunker;

/* The default 256-word table (casync / Proxmox BUZHASH_TABLE, recalled from
 * the public sources; not verifiable offline — every API takes the table as
 * an input so a maintainer can inject the module's own constant). */
const uint32_t *oracle_default_table(void);

/* 0 on success, -1 if avg is not a power of two or is < 256 (min >= window
 * is required by the position-independence argument the GPU engine relies
 * on; upstream asserts the power of two). table == NULL -> default table. */
int oracle_config_init(uint64_t avg, const uint32_t *table, oracle_config *out);

void oracle_chunker_init(oracle_chunker *c, const oracle_config *cfg);

/* Upstream `scan` semantics: consume bytes from data; return 0 if no
 * boundary was found (all bytes consumed), else the number of bytes up to
 * and including the last byte of the chunk (state is reset). */
size_t oracle_chunker_scan(oracle_chunker *c, const uint8_t *data, size_t len);

/* Whole-stream helper: cut [data, data+len) as ONE stream (fresh state,
 * tail emitted as the final chunk). Writes chunk END offsets (exclusive,
 * relative to data) into ends[0..cap). Returns the number of chunks (may
 * exceed cap; only cap are written). */
size_t oracle_chunk_stream(const oracle_config *cfg, const uint8_t *data, size_t len,
                           uint64_t *ends, size_t cap);

/* Payload chunker = ChunkerImpl + suggested boundaries (published Proxmox PayloadChunker; EXTERNAL, recalled). */
typedef struct oracle_payload_chunker {
    oracle_chunker c;
    const uint64_t *sugg; /* the channel: absolute stream offsets in send order */
    size_t nsugg, next;
    int have_cur;
    uint64_t cur;
} oracle_payload_chunker;
void oracle_payload_chunker_init(oracle_payload_chunker *p, const oracle_config *cfg, const uint64_t *sugg, size_t nsugg);
/* base = stream offset of the current chunk's first byte, total = bytes of the current chunk so far INCLUDING data */
size_t oracle_payload_chunker_scan(oracle_payload_chunker *p, const uint8_t *data, size_t len, uint64_t base,
                                   uint64_t total);
/* one stream, `feed` bytes per scan call (0 = all that is left; 1 = byte-serial = the engine's definition) */
size_t oracle_chunk_stream_suggested(const oracle_config *cfg, const uint8_t *data, size_t len, const uint64_t *sugg,
                                     size_t nsugg, size_t feed, uint64_t *ends, size_t cap);
size_t oracle_chunk_stream_suggested_grid(const oracle_config *cfg, const uint8_t *data, size_t len, const uint64_t *sugg,
                                          size_t nsugg, size_t feed, int absolute, uint64_t *ends, size_t cap);

/* Raw candidates: every END offset e (64 <= e <= len) whose 64-byte window [e-64, e)
 * passes the break test (no min/max, no resets), ascending. Returns the count. */
size_t oracle_candidates(const oracle_config *cfg, const uint8_t *data, size_t len, uint64_t *out, size_t cap);

/* SHA-256 (FIPS 180-4). impl: 0 = portable scalar, 1 = SHA-NI if the CPU
 * has it (falls back to scalar), both bit-identical. */
typedef struct oracle_sha256_ctx {
    uint32_t state[8];
    uint64_t nbytes;
    uint8_t buf[64];
    uint32_t buflen;
    int impl;
} oracle_sha256_ctx;

void oracle_sha256_init(oracle_sha256_ctx *s, int impl);
void oracle_sha256_update(oracle_sha256_ctx *s, const uint8_t *data, size_t len);
void oracle_sha256_final(oracle_sha256_ctx *s, uint8_t out[32]);
void oracle_sha256(const uint8_t *data, size_t len, uint8_t out[32], int impl);
int oracle_have_shani(void);

/* One record per chunk, identical in layout to the engine's pbsgpu_record. */
typedef struct oracle_record {
    uint64_t end;        /* chunk end offset, exclusive, relative to its segment start */
    uint8_t digest[32];  /* SHA-256 of the raw chunk bytes (crypt mode none) */
    uint32_t segment;
    uint32_t size;       /* chunk length in bytes */
} oracle_record;

typedef struct oracle_segment {
    uint64_t offset;
    uint64_t length;
} oracle_segment;

/* Batch helper: every segment is an independent stream (fresh chunker, forced
 * cut at its end). Returns the number of records (may exceed cap). */
size_t oracle_chunk_and_digest(const oracle_config *cfg, const uint8_t *base,
                               const oracle_segment *segs, uint32_t nseg,
                               oracle_record *out, size_t cap, int sha_impl);

/* ... with suggested boundaries per segment (byte-serial feed): sugg[sugg_idx[s] .. sugg_idx[s+1]) are the
 * ascending boundaries of segment s, relative to its start. sugg == NULL -> none. */
size_t oracle_chunk_and_digest_suggested(const oracle_config *cfg, const uint8_t *base, const oracle_segment *segs,
                                         uint32_t nseg, const uint64_t *sugg, const uint32_t *sugg_idx,
                                         oracle_record *out, size_t cap, int sha_impl);

/* Deterministic synthetic byte generator shared with the engine's device fill
 * kernel (pbsgpu_fill): 8 bytes per counter via splitmix64(seed, index),
 * shaped by `kind`: 0 random, 1 zeros, 2 repeating 4 KiB block,
 * 3 random with ~30 % zero extents (64 KiB granules), 4 random from two ChaCha
 * quarter-rounds per 16-byte block (the cheap generator of the page ring's refill). `stream_off` is the
 * absolute byte offset of dst[0] in the synthetic stream (multiple of 8). */
void oracle_fill(uint8_t *dst, uint64_t stream_off, uint64_t len, uint64_t seed, uint32_t kind);

#ifdef __cplusplus
}
#endif
#endif
