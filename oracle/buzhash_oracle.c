/*
 * oracle/buzhash_oracle.c — scalar restatement of the Buzhash content-defined
 * chunker (TEST INFRASTRUCTURE ONLY; see oracle.h header; PARITY UNPINNED).
 *
 * Restates the callee of buzhash.NewConfig (reference
 * internal/pxarmount/commit_orchestrate.go:143-149, internal/tapeio/converter.go:248):
 * the chunker of github.com/pbs-plus/pxar v0.34.0 (go.mod:30), which is absent
 * from /root/reference. The algorithm below is the published Proxmox Backup
 * `pbs-datastore/src/chunker.rs` ChunkerImpl::scan / shall_break (itself the
 * casync cachunker design), as summarised in SURVEY.md Appendix A: a 32-bit
 * cyclic-polynomial hash over a 64-byte ring, warm-up without a `leave` term
 * after every cut, break test only in the rolling loop.
 *
 * Written byte-at-a-time on purpose: this is the serial statement the GPU
 * engine's candidate/resolve decomposition has to reproduce bit for bit.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

static const uint32_t k_default_table[256] = {
    0x458be752, 0xc10748cc, 0xfbbcdbb8, 0x6ded5b68, 0xb10a82b5, 0x20d75648, 0xdfc5665f, 0xa8428801,
    0x7ebf5191, 0x841135c7, 0x65cc53b3, 0x280a597c, 0x16f60255, 0xc78cbc3e, 0x294415f5, 0xb938d494,
    0xec85c4e6, 0xb7d33edc, 0xe549b544, 0xfdeda5aa, 0x882bf287, 0x3116737c, 0x05569956, 0xe8cc1f68,
    0x0806ac5e, 0x22a14443, 0x15297e10, 0x50d090e7, 0x4ba60f6f, 0xefd9f1a7, 0x5c5c885c, 0x82482f93,
    0x9bfd7c64, 0x0b3e7276, 0xf2688e77, 0x8fad8abc, 0xb0509568, 0xf1ada29f, 0xa53efdfe, 0xcb2b1d00,
    0xf2a9e986, 0x6463432b, 0x95094051, 0x5a223ad2, 0x9be8401b, 0x61e579cb, 0x1a556a14, 0x5840fdc2,
    0x9261ddf6, 0xcde002bb, 0x52432bb0, 0xbf17373e, 0x7b7c222f, 0x2955ed16, 0x9f10ca59, 0xe840c4c9,
    0xccabd806, 0x14543f34, 0x1462417a, 0x0d4a1f9c, 0x087ed925, 0xd7f8f24c, 0x7338c425, 0xcf86c8f5,
    0xb19165cd, 0x9891c393, 0x325384ac, 0x0308459d, 0x86141d7e, 0xc922116a, 0xe2ffa6b6, 0x53f52aed,
    0x2cd86197, 0xf5b9f498, 0xbf319c8f, 0xe0411fae, 0x977eb18c, 0xd8770976, 0x9833466a, 0xc674df7f,
    0x8c297d45, 0x8ca48d26, 0xc49ed8e2, 0x7344f874, 0x556f79c7, 0x6b25eaed, 0xa03e2b42, 0xf68f66a4,
    0x8e8b09a2, 0xf2e0e62a, 0x0d3a9806, 0x9729e493, 0x8c72b0fc, 0x160b94f6, 0x450e4d3d, 0x7a320e85,
    0xbef8f0e1, 0x21d73653, 0x4e3d977a, 0x1e7b3929, 0x1cc6c719, 0xbe478d53, 0x8d752809, 0xe6d8c2c6,
    0x275f0892, 0xc8acc273, 0x4cc21580, 0xecc4a617, 0xf5f7be70, 0xe795248a, 0x375a2fe9, 0x425570b6,
    0x8898dcf8, 0xdc2d97c4, 0x0106114b, 0x364dc22f, 0x1e0cad1f, 0xbe63803c, 0x5f69fac2, 0x4d5afa6f,
    0x1bc0dfb5, 0xfb273589, 0x0ea47f7b, 0x3c1c2b50, 0x21b2a932, 0x6b1223fd, 0x2fe706a8, 0xf9bd6ce2,
    0xa268e64e, 0xe987f486, 0x3eacf563, 0x1ca2018c, 0x65e18228, 0x2207360a, 0x57cf1715, 0x34c37d2b,
    0x1f8f3cde, 0x93b657cf, 0x31a019fd, 0xe69eb729, 0x8bca7b9b, 0x4c9d5bed, 0x277ebeaf, 0xe0d8f8ae,
    0xd150821c, 0x31381871, 0xafc3f1b0, 0x927db328, 0xe95effac, 0x305a47bd, 0x426ba35b, 0x1233af3f,
    0x686a5b83, 0x50e072e5, 0xd9d3bb2a, 0x8befc475, 0x487f0de6, 0xc88dff89, 0xbd664d5e, 0x971b5d18,
    0x63b14847, 0xd7d3c1ce, 0x7f583cf3, 0x72cbcb09, 0xc0d0a81c, 0x7fa3429b, 0xe9158a1b, 0x225ea19a,
    0xd8ca9ea3, 0xc763b282, 0xbb0c6341, 0x020b8293, 0xd4cd299d, 0x58cfa7f8, 0x91b4ee53, 0x37e4d140,
    0x95ec764c, 0x30f76b06, 0x5ee68d24, 0x679c8661, 0xa41979c2, 0xf2b61284, 0x4fac1475, 0x0adb49f9,
    0x19727a23, 0x15a7e374, 0xc43a18d5, 0x3fb1aa73, 0x342fc615, 0x924c0793, 0xbee2d7f0, 0x8a279de9,
    0x4aa2d70c, 0xe24dd37f, 0xbe862c0b, 0x177c22c2, 0x5388e5ee, 0xcd8a7510, 0xf901b4fd, 0xdbc13dbc,
    0x6c0bae5b, 0x64efe8c7, 0x48b02079, 0x80331a49, 0xca3d8ae6, 0xf3546190, 0xfed7108b, 0xc49b941b,
    0x32baf4a9, 0xeb833a4a, 0x88a3f1a5, 0x3a91ce0a, 0x3cc27da1, 0x7112e684, 0x4a3096b1, 0x3794574c,
    0xa3c8b6f3, 0x1d213941, 0x6e0a2e00, 0x233479f1, 0x0f4cd82f, 0x6093edd2, 0x5d7d209e, 0x464fe319,
    0xd4dcac9e, 0x0db845cb, 0xfb5e4bc3, 0xe0256ce1, 0x09fb4ed1, 0x0914be1e, 0xa5bdb2c3, 0xc6eb57bb,
    0x30320350, 0x3f397e91, 0xa67791bc, 0x86bc0e2c, 0xefa0a7e2, 0xe9ff7543, 0xe733612c, 0xd185897b,
    0x329e5388, 0x91dd236b, 0x2ecb0d93, 0xf4d82a3d, 0x35b5c03f, 0xe4e606f0, 0x05b21843, 0x37b45964,
    0x5eff22f4, 0x6027f4cc, 0x77178b3c, 0xae507131, 0x7bf7cabc, 0xf9c18d66, 0x593ade65, 0xd95ddf11,
};

const uint32_t *oracle_default_table(void) { return k_default_table; }

static inline uint32_t rotl32(uint32_t x, unsigned n) { return (x << n) | (x >> (32u - n)); }

int oracle_config_init(uint64_t avg, const uint32_t *table, oracle_config *out) {
    if (!out) return -1;
    /* upstream asserts popcount(avg) == 1; min >= window needs avg >= 256 */
    if (avg < 256 || avg > (1u << 28) || (avg & (avg - 1)) != 0) return -1;
    out->avg = (uint32_t)avg;
    out->min = (uint32_t)(avg >> 2);
    out->max = (uint32_t)(avg << 2);
    out->window = ORACLE_WINDOW;
    out->mask = (uint32_t)(avg * 2 - 1);
    out->break_min = out->mask - 2;
    memcpy(out->table, table ? table : k_default_table, sizeof(out->table));
    return 0;
}

void oracle_chunker_init(oracle_chunker *c, const oracle_config *cfg) {
    memset(c, 0, sizeof(*c));
    c->cfg = *cfg;
}

static inline int shall_break(const oracle_chunker *c) {
    if (c->chunk_size >= c->cfg.max) return 1;
    if (c->chunk_size < c->cfg.min) return 0;
    return (c->h & c->cfg.mask) >= c->cfg.break_min;
}

size_t oracle_chunker_scan(oracle_chunker *c, const uint8_t *data, size_t len) {
    const uint32_t *T = c->cfg.table;
    size_t pos = 0;

    /* warm-up: absorb bytes until the ring is full; no break test here */
    if (c->window_size < ORACLE_WINDOW) {
        size_t need = ORACLE_WINDOW - c->window_size;
        size_t n = need < len ? need : len;
        for (size_t i = 0; i < n; i++) {
            uint8_t b = data[pos++];
            c->window[c->window_size++] = b;
            c->h = rotl32(c->h, 1) ^ T[b];
        }
        c->chunk_size += n;
        if (c->window_size < ORACLE_WINDOW) return 0;
    }

    unsigned idx = (unsigned)(c->chunk_size & 63u);
    while (pos < len) {
        uint8_t enter = data[pos];
        uint8_t leave = c->window[idx];
        /* rotl(T[leave], 64) == T[leave] for a 32-bit word */
        c->h = rotl32(c->h, 1) ^ T[leave] ^ T[enter];
        c->chunk_size++;
        pos++;
        c->window[idx] = enter;
        if (shall_break(c)) {
            c->h = 0;
            c->chunk_size = 0;
            c->window_size = 0;
            return pos;
        }
        idx = (unsigned)(c->chunk_size & 63u);
    }
    return 0;
}

size_t oracle_chunk_stream(const oracle_config *cfg, const uint8_t *data, size_t len,
                           uint64_t *ends, size_t cap) {
    oracle_chunker c;
    oracle_chunker_init(&c, cfg);
    size_t n = 0, pos = 0;
    while (pos < len) {
        size_t k = oracle_chunker_scan(&c, data + pos, len - pos);
        if (k == 0) break;
        pos += k;
        if (n < cap) ends[n] = pos;
        n++;
    }
    if (pos < len) { /* end of stream: the tail is the final chunk */
        if (n < cap) ends[n] = len;
        n++;
    }
    return n;
}

size_t oracle_chunk_and_digest(const oracle_config *cfg, const uint8_t *base,
                               const oracle_segment *segs, uint32_t nseg,
                               oracle_record *out, size_t cap, int sha_impl) {
    size_t n = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        const uint8_t *p = base + segs[s].offset;
        uint64_t len = segs[s].length;
        oracle_chunker c;
        oracle_chunker_init(&c, cfg);
        uint64_t pos = 0, start = 0;
        while (pos < len) {
            size_t k = oracle_chunker_scan(&c, p + pos, len - pos);
            uint64_t end;
            if (k == 0) end = len; else end = pos + k;
            pos = end;
            if (n < cap) {
                out[n].end = end;
                out[n].segment = s;
                out[n].size = (uint32_t)(end - start);
                oracle_sha256(p + start, end - start, out[n].digest, sha_impl);
            }
            n++;
            start = end;
        }
    }
    return n;
}

/* ---- payload chunker: suggested boundaries ------------------------------------------------------
 * Restatement of the published Proxmox `PayloadChunker::scan` (pbs-datastore/src/chunker.rs; EXTERNAL, recalled
 * — whether github.com/pbs-plus/pxar v0.34.0 ports it is open). The caller (upstream's
 * ChunkStream) passes ctx.base = stream offset of the current chunk's first byte and ctx.total = bytes of the
 * current chunk accumulated so far INCLUDING `data`. A suggested boundary b (absolute stream offset = the END
 * of the chunk it would close) is taken when it falls inside the bytes seen so far and the resulting chunk size
 * b - base lies in [min, max]; a boundary that would give a chunk < min is dropped, one in the past is dropped,
 * one still in the future or too far (> max) leaves the decision to the hash scan. */
void oracle_payload_chunker_init(oracle_payload_chunker *p, const oracle_config *cfg, const uint64_t *sugg, size_t nsugg) {
    oracle_chunker_init(&p->c, cfg);
    p->sugg = sugg;
    p->nsugg = nsugg;
    p->next = 0;
    p->have_cur = 0;
    p->cur = 0;
}

size_t oracle_payload_chunker_scan(oracle_payload_chunker *p, const uint8_t *data, size_t len, uint64_t base,
                                   uint64_t total) {
    const uint64_t pos = total - (uint64_t)len; /* bytes of the current chunk before `data` */
    for (;;) {
        if (p->have_cur) {
            const uint64_t b = p->cur;
            if (b < base + pos) { /* boundary in the past: ignore */
                p->have_cur = 0;
                continue;
            }
            if (b > base + total) /* boundary in the future: cannot decide yet */
                return oracle_chunker_scan(&p->c, data, len);
            const uint64_t chunk_size = b - base;
            if (chunk_size < p->c.cfg.min) { /* chunk too small: ignore the boundary */
                p->have_cur = 0;
                continue;
            }
            if (chunk_size <= p->c.cfg.max) {
                p->have_cur = 0;
                const uint64_t n = chunk_size - pos; /* boundary relative to the start of `data` */
                if (n == 0) return oracle_chunker_scan(&p->c, data, len); /* passed: previous scan did not know it yet */
                p->c.h = 0; /* chunker.reset() */
                p->c.chunk_size = 0;
                p->c.window_size = 0;
                return (size_t)n;
            }
            /* chunk too big: regular scan decides (the boundary stays pending) */
            return oracle_chunker_scan(&p->c, data, len);
        }
        if (p->next < p->nsugg) { /* try_recv */
            p->cur = p->sugg[p->next++];
            p->have_cur = 1;
        } else {
            return oracle_chunker_scan(&p->c, data, len);
        }
    }
}

/* Cut one stream with suggested boundaries, handing the chunker `feed` bytes per scan call (0 = everything that
 * is left, like upstream's whole-buffer test; 1 = byte-serial, the feed-independent limit the engine implements:
 * an earlier hash cut wins over a later suggested boundary). */
size_t oracle_chunk_stream_suggested(const oracle_config *cfg, const uint8_t *data, size_t len, const uint64_t *sugg,
                                     size_t nsugg, size_t feed, uint64_t *ends, size_t cap) {
    return oracle_chunk_stream_suggested_grid(cfg, data, len, sugg, nsugg, feed, 0, ends, cap);
}

/* absolute != 0: the reader appends fixed reads of `feed` bytes to its buffer, so a scan call ends at the next multiple
 * of `feed` from the STREAM start (after a cut the rest of the buffered read is scanned first) — the shape of a Go
 * io.Reader loop with a fixed buffer; absolute == 0: `feed` bytes per call counted from the last cut. */
size_t oracle_chunk_stream_suggested_grid(const oracle_config *cfg, const uint8_t *data, size_t len, const uint64_t *sugg,
                                          size_t nsugg, size_t feed, int absolute, uint64_t *ends, size_t cap) {
    oracle_payload_chunker p;
    oracle_payload_chunker_init(&p, cfg, sugg, nsugg);
    size_t n = 0;
    uint64_t base = 0, pos = 0;
    while (pos < len) {
        size_t take = (feed == 0 || feed > len - pos) ? (size_t)(len - pos) : feed;
        if (absolute && feed) {
            const size_t to_grid = feed - (size_t)(pos % feed);
            take = to_grid > len - pos ? (size_t)(len - pos) : to_grid;
        }
        size_t k = oracle_payload_chunker_scan(&p, data + pos, take, base, (pos - base) + take);
        if (k == 0) {
            pos += take;
            continue;
        }
        pos += k;
        base = pos;
        if (n < cap) ends[n] = pos;
        n++;
    }
    if (base < len) {
        if (n < cap) ends[n] = len;
        n++;
    }
    return n;
}

size_t oracle_chunk_and_digest_suggested(const oracle_config *cfg, const uint8_t *base, const oracle_segment *segs,
                                         uint32_t nseg, const uint64_t *sugg, const uint32_t *sugg_idx,
                                         oracle_record *out, size_t cap, int sha_impl) {
    size_t n = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        const uint8_t *p = base + segs[s].offset;
        const uint64_t len = segs[s].length;
        size_t ecap = (size_t)(len / (cfg->min ? cfg->min : 1)) + 4;
        uint64_t *ends = (uint64_t *)malloc(ecap * sizeof(uint64_t));
        if (!ends) return 0;
        const uint64_t *sl = sugg ? sugg + sugg_idx[s] : NULL;
        const size_t ns = sugg ? (size_t)(sugg_idx[s + 1] - sugg_idx[s]) : 0;
        size_t k = oracle_chunk_stream_suggested(cfg, p, (size_t)len, sl, ns, 1, ends, ecap);
        uint64_t start = 0;
        for (size_t i = 0; i < k && i < ecap; i++) {
            if (n < cap) {
                out[n].end = ends[i];
                out[n].segment = s;
                out[n].size = (uint32_t)(ends[i] - start);
                oracle_sha256(p + start, ends[i] - start, out[n].digest, sha_impl);
            }
            n++;
            start = ends[i];
        }
        free(ends);
    }
    return n;
}

/* Every END offset e (64 <= e <= len) whose window [e-64, e) satisfies the break test,
 * ascending: the rolling hash of ChunkerImpl::scan run over the whole buffer WITHOUT
 * resets or min/max rules. This is what the engine's candidate kernel must reproduce;
 * the serial chunker only ever acts on a subset of these positions. */
size_t oracle_candidates(const oracle_config *cfg, const uint8_t *data, size_t len, uint64_t *out, size_t cap) {
    const uint32_t *T = cfg->table;
    size_t n = 0;
    if (len < ORACLE_WINDOW) return 0;
    uint32_t h = 0;
    for (size_t i = 0; i < ORACLE_WINDOW; i++) h = rotl32(h, 1) ^ T[data[i]];
    for (size_t e = ORACLE_WINDOW;; e++) { /* h = hash of [e-64, e) */
        if ((h & cfg->mask) >= cfg->break_min) {
            if (n < cap) out[n] = e;
            n++;
        }
        if (e == len) break;
        h = rotl32(h, 1) ^ T[data[e - ORACLE_WINDOW]] ^ T[data[e]];
    }
    return n;
}

/* ---- synthetic data generator (twin of the engine's device fill kernel) ---- */
static inline uint64_t splitmix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* kind 4: 16-byte blocks from two ChaCha quarter-rounds over {block index, seed} — adds, xors and rotates only, ~1/3
 * of splitmix64's issue slots on the GPU (64-bit multiplies run at quarter rate there). Statistically sound for this
 * purpose: byte histogram chi^2 238 over 64 MiB, candidate and chunk counts at the nominal density (measured against
 * splitmix64 data). Used where the generator runs inside a timed region (the page ring's refill). */
static inline uint32_t rotl32g(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static inline uint64_t chacha2_word(uint64_t widx, uint64_t seed) {
    uint64_t q = widx >> 1;
    uint32_t x0 = (uint32_t)q ^ 0x61707865u, x1 = (uint32_t)(q >> 32) ^ 0x3320646eu;
    uint32_t x2 = (uint32_t)seed ^ 0x79622d32u, x3 = (uint32_t)(seed >> 32) ^ 0x6b206574u;
    for (int r = 0; r < 2; ++r) {
        x0 += x1; x3 = rotl32g(x3 ^ x0, 16);
        x2 += x3; x1 = rotl32g(x1 ^ x2, 12);
        x0 += x1; x3 = rotl32g(x3 ^ x0, 8);
        x2 += x3; x1 = rotl32g(x1 ^ x2, 7);
    }
    return (widx & 1u) ? ((uint64_t)x3 << 32) | x2 : ((uint64_t)x1 << 32) | x0;
}

static inline uint64_t fill_word(uint64_t widx, uint64_t seed, uint32_t kind) {
    switch (kind) {
    case 0: return splitmix64(seed, widx);
    case 4: return chacha2_word(widx, seed);
    case 1: return 0;
    case 2: return splitmix64(seed, widx & 511u); /* 4 KiB period */
    default: {
        uint64_t g = widx >> 13; /* 64 KiB granule */
        uint64_t r = splitmix64(seed ^ 0xA5A5A5A55A5A5A5Aull, g);
        if ((((r >> 32) * 10u) >> 32) < 3u) return 0;
        return splitmix64(seed, widx);
    }
    }
}

void oracle_fill(uint8_t *dst, uint64_t stream_off, uint64_t len, uint64_t seed, uint32_t kind) {
    uint64_t w0 = stream_off >> 3;
    uint64_t i = 0;
    unsigned skip = (unsigned)(stream_off & 7u);
    if (skip) {
        uint64_t w = fill_word(w0, seed, kind);
        for (unsigned b = skip; b < 8 && i < len; b++) dst[i++] = (uint8_t)(w >> (8 * b));
        w0++;
    }
    while (i + 8 <= len) {
        uint64_t w = fill_word(w0++, seed, kind);
        memcpy(dst + i, &w, 8); /* little-endian host */
        i += 8;
    }
    if (i < len) {
        uint64_t w = fill_word(w0, seed, kind);
        for (unsigned b = 0; i < len; b++) dst[i++] = (uint8_t)(w >> (8 * b));
    }
}
