// libpbsgpu host side, part 2: the streaming front ends and the record-set utilities.
//
//  * pbsgpu_stream_*  — the payload-stream seam transfer.ArchiveWriter.WriteEntryReader
//    feeds (reference internal/pxarmount/commit_reuse.go:427-468, commit_walk.go:465-479,
//    internal/tapeio/converter.go:827-842): bytes are appended to one continuous stream and
//    (end, digest) records fall out in order, exactly what the module appends to the .didx.
//  * pbsgpu_chunker_* — upstream `scan(data) -> pos` compatibility (buzhash.Config's chunker).
//  * pbsgpu_dedup_host — digest-set duplicate detection on the device.
//  * pbsgpu_didx_*    — dynamic index encode/decode (commit_bottleneck_test.go:773-793).
// All byte-stream work runs through the same HIP kernels as the batch path.
#include <deque>

#include "engine_internal.h"

using namespace pbse;

// -------------------------------------------------------------------------------------
// streaming writer
// -------------------------------------------------------------------------------------
// Windows of the stream are CUT synchronously (scan + resolve: milliseconds) because the next
// window needs to know where the still-open chunk starts, but HASHED asynchronously: the SHA-256
// kernel of a window is bounded below by the serial chain of its longest chunk (up to ~0.46 s for a
// 16 MiB chunk), so several windows hash concurrently on their own engine slots / HIP streams while
// the writer keeps accepting bytes. Records are delivered strictly in stream order.
namespace {

struct WindowInFlight {
    Slot *slot = nullptr;
    int buf = -1;
    uint64_t base = 0;     // absolute stream offset of the window buffer's byte 0
    uint32_t section = 0;
    uint64_t nemit = 0;    // records to deliver (the open tail chunk of a non-final window is excluded)
};

constexpr size_t kStreamStage = 32u << 20;

}  // namespace

struct pbsgpu_stream {
    pbsgpu_engine *eng = nullptr;
    uint64_t window = 0;           // new bytes per device window
    std::vector<DevBuf> dev;       // window buffers: [carry | new bytes]
    std::vector<char> dev_busy;    // held by a window that is still hashing
    int cur = 0;
    uint64_t carry = 0;            // bytes of the still-open chunk at the front of dev[cur]
    uint64_t fill = 0;             // new bytes already copied behind the carry
    hipStream_t copy_stream = nullptr;
    PinnedBuf stage[2];
    hipEvent_t stage_ev[2] = {};
    int stage_idx = 0;
    int reserved = -1;             // staging buffer handed out by pbsgpu_stream_reserve
    uint64_t base = 0;             // absolute stream offset of dev[cur][0]
    uint64_t written = 0;
    uint64_t inject_total = 0;
    uint32_t section = 0;
    bool finished = false;
    std::deque<WindowInFlight> inflight;
    std::deque<pbsgpu_record> out;
    std::vector<pbsgpu_record> tmp;
};

namespace {

// wait for the oldest hashing window, move its records to the output queue, release slot + buffer
int stream_complete_oldest(pbsgpu_stream *s) {
    WindowInFlight w = s->inflight.front();
    HIPCHK(hipStreamSynchronize(w.slot->stream));
    s->tmp.resize((size_t)w.nemit);
    if (w.nemit) {
        HIPCHK(hipMemcpy(s->tmp.data(), w.slot->recs.p, (size_t)w.nemit * sizeof(pbsgpu_record), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < w.nemit; ++i) {
            pbsgpu_record r = s->tmp[(size_t)i];
            r.end += w.base;
            r.segment = w.section;
            s->out.push_back(r);
        }
    }
    w.slot->busy = false;
    s->dev_busy[w.buf] = 0;
    s->inflight.pop_front();
    return PBSGPU_OK;
}

int stream_free_buffer(pbsgpu_stream *s) {
    for (size_t i = 0; i < s->dev.size(); ++i)
        if (!s->dev_busy[i] && (int)i != s->cur) return (int)i;
    return -1;
}

// cut [carry | fill] of the current window; hash all complete chunks in the background; when not
// `final` the tail chunk stays open and its bytes move to the front of the next window buffer (a cut
// only depends on bytes before it, so re-examining them with more data reproduces the serial chunker)
int stream_flush(pbsgpu_stream *s, bool final) {
    pbsgpu_engine *e = s->eng;
    const uint64_t total = s->carry + s->fill;
    if (total == 0) return PBSGPU_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    HIPCHK(hipStreamSynchronize(s->copy_stream));
    Slot *slot = find_free_slot(e);
    while (!slot) {
        if (s->inflight.empty()) return PBSGPU_E_BUSY;  // slots held by other users of the engine
        CHK(stream_complete_oldest(s));
        slot = find_free_slot(e);
    }
    DevBuf &buf = s->dev[s->cur];
    uint64_t nrec = 0;
    pbsgpu_segment seg{0, total};
    CHK(cut_sync(e, *slot, buf.as<uint8_t>(), total, &seg, 1, &nrec));
    const uint64_t nemit = final ? nrec : (nrec ? nrec - 1 : 0);
    pbsgpu_record open{};
    if (!final && nrec) {
        HIPCHK(hipMemcpy(&open, slot->recs.as<pbsgpu_record>() + (nrec - 1), sizeof(open), hipMemcpyDeviceToHost));
    }
    if (nemit) {
        CHK(hash_async(e, *slot, nemit));
        slot->busy = true;
        slot->ticket = ~0ull;  // owned by the stream, never collectable through the batch API
        s->dev_busy[s->cur] = 1;
        WindowInFlight w;
        w.slot = slot;
        w.buf = s->cur;
        w.base = s->base;
        w.section = s->section;
        w.nemit = nemit;
        s->inflight.push_back(w);
    }
    if (final || nrec == 0) {
        s->base += total;
        s->carry = 0;
        if (nemit) {  // the buffer is still being read by the SHA kernel: continue in another one
            int nb = stream_free_buffer(s);
            while (nb < 0) {
                CHK(stream_complete_oldest(s));
                nb = stream_free_buffer(s);
            }
            s->cur = nb;
        }
    } else {
        const uint64_t open_start = open.end - open.size;
        int nb = stream_free_buffer(s);
        while (nb < 0) {
            CHK(stream_complete_oldest(s));
            nb = stream_free_buffer(s);
        }
        HIPCHK(hipMemcpyAsync(s->dev[nb].p, buf.as<uint8_t>() + open_start, open.size, hipMemcpyDeviceToDevice,
                              s->copy_stream));
        HIPCHK(hipStreamSynchronize(s->copy_stream));
        s->base += open_start;
        s->carry = open.size;
        s->cur = nb;
    }
    s->fill = 0;
    return PBSGPU_OK;
}

// non-blocking: collect windows whose SHA kernel has finished
int stream_reap(pbsgpu_stream *s) {
    std::lock_guard<std::mutex> lk(s->eng->mu);
    CHK(set_device(s->eng));
    while (!s->inflight.empty()) {
        hipError_t q = hipStreamQuery(s->inflight.front().slot->stream);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            break;
        }
        if (q != hipSuccess) {
            g_last_hip_error.store((int)q);
            return PBSGPU_E_HIP;
        }
        CHK(stream_complete_oldest(s));
    }
    return PBSGPU_OK;
}

int stream_drain(pbsgpu_stream *s) {
    std::lock_guard<std::mutex> lk(s->eng->mu);
    CHK(set_device(s->eng));
    while (!s->inflight.empty()) CHK(stream_complete_oldest(s));
    return PBSGPU_OK;
}

}  // namespace

// -------------------------------------------------------------------------------------
// upstream-style chunker
// -------------------------------------------------------------------------------------
struct pbsgpu_chunker {
    pbsgpu_engine *eng = nullptr;
    uint64_t chunk_size = 0;  // bytes consumed since the last cut
    uint8_t tail[63];         // last bytes consumed (window continuity across calls)
    uint32_t tail_len = 0;
    DevBuf dev;
    PinnedBuf host;
    std::vector<uint64_t> cands;
};

namespace {

constexpr size_t kChunkerSlice = 8u << 20;

void chunker_push_tail(pbsgpu_chunker *c, const uint8_t *p, size_t n) {
    if (n >= 63) {
        std::memcpy(c->tail, p + n - 63, 63);
        c->tail_len = 63;
        return;
    }
    const uint32_t keep = std::min<uint32_t>(c->tail_len, (uint32_t)(63 - n));
    std::memmove(c->tail, c->tail + (c->tail_len - keep), keep);
    std::memcpy(c->tail + keep, p, n);
    c->tail_len = keep + (uint32_t)n;
}

}  // namespace

extern "C" {

int pbsgpu_stream_create(pbsgpu_engine *e, uint64_t window_bytes, pbsgpu_stream **out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    *out = nullptr;
    if (window_bytes == 0) window_bytes = 256ull << 20;
    if (window_bytes < e->cfg.max) window_bytes = e->cfg.max;
    pbsgpu_stream *s = new (std::nothrow) pbsgpu_stream();
    if (!s) return PBSGPU_E_NOMEM;
    s->eng = e;
    s->window = window_bytes;
    int st;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        st = set_device(e);
        const size_t devcap = (size_t)window_bytes + (size_t)e->cfg.max + 256;
        const size_t nbuf = e->slots.size() + 1;  // one being filled + one per hashing window
        s->dev.resize(nbuf);
        s->dev_busy.assign(nbuf, 0);
        for (size_t i = 0; i < nbuf && st == PBSGPU_OK; ++i) st = s->dev[i].ensure(devcap);
        for (int i = 0; i < 2 && st == PBSGPU_OK; ++i) st = s->stage[i].ensure(kStreamStage);
        if (st == PBSGPU_OK && hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess) st = PBSGPU_E_HIP;
        for (int i = 0; i < 2 && st == PBSGPU_OK; ++i)
            if (hipEventCreateWithFlags(&s->stage_ev[i], hipEventDisableTiming) != hipSuccess) st = PBSGPU_E_HIP;
    }
    if (st != PBSGPU_OK) {
        pbsgpu_stream_destroy(s);
        return st;
    }
    *out = s;
    return PBSGPU_OK;
}

void pbsgpu_stream_destroy(pbsgpu_stream *s) {
    if (!s) return;
    if (s->eng) {
        (void)stream_drain(s);
        std::lock_guard<std::mutex> lk(s->eng->mu);
        (void)hipSetDevice(s->eng->device);
        if (s->copy_stream) {
            (void)hipStreamSynchronize(s->copy_stream);
            (void)hipStreamDestroy(s->copy_stream);
        }
        for (auto &b : s->dev) b.release();
        for (auto &b : s->stage) b.release();
        for (auto &ev : s->stage_ev)
            if (ev) (void)hipEventDestroy(ev);
    }
    delete s;
}

int pbsgpu_stream_write(pbsgpu_stream *s, const void *data, size_t len) {
    if (!s || (!data && len)) return PBSGPU_E_INVALID;
    if (s->finished || s->reserved >= 0) return PBSGPU_E_STATE;
    const uint8_t *p = static_cast<const uint8_t *>(data);
    while (len) {
        size_t n = (size_t)std::min<uint64_t>(len, s->window - s->fill);
        n = std::min(n, kStreamStage);
        {   // caller bytes -> library-owned pinned staging -> device window (async on the copy stream)
            std::lock_guard<std::mutex> lk(s->eng->mu);
            CHK(set_device(s->eng));
            const int k = s->stage_idx;
            HIPCHK(hipEventSynchronize(s->stage_ev[k]));
            std::memcpy(s->stage[k].p, p, n);
            HIPCHK(hipMemcpyAsync(s->dev[s->cur].as<uint8_t>() + s->carry + s->fill, s->stage[k].p, n,
                                  hipMemcpyHostToDevice, s->copy_stream));
            HIPCHK(hipEventRecord(s->stage_ev[k], s->copy_stream));
            s->stage_idx ^= 1;
        }
        s->fill += n;
        s->written += n;
        p += n;
        len -= n;
        if (s->fill == s->window) CHK(stream_flush(s, false));
    }
    return PBSGPU_OK;
}

int pbsgpu_stream_reserve(pbsgpu_stream *s, void **buf, size_t *cap) {
    if (!s || !buf || !cap) return PBSGPU_E_INVALID;
    if (s->finished || s->reserved >= 0) return PBSGPU_E_STATE;
    std::lock_guard<std::mutex> lk(s->eng->mu);
    CHK(set_device(s->eng));
    const int k = s->stage_idx;
    HIPCHK(hipEventSynchronize(s->stage_ev[k]));  // its previous H2D copy has drained
    s->reserved = k;
    *buf = s->stage[k].p;
    *cap = (size_t)std::min<uint64_t>(kStreamStage, s->window - s->fill);
    return PBSGPU_OK;
}

int pbsgpu_stream_commit(pbsgpu_stream *s, size_t len) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->reserved < 0) return PBSGPU_E_STATE;
    const int k = s->reserved;
    if (len > (size_t)std::min<uint64_t>(kStreamStage, s->window - s->fill)) return PBSGPU_E_INVALID;
    s->reserved = -1;
    if (len == 0) return PBSGPU_OK;
    {
        std::lock_guard<std::mutex> lk(s->eng->mu);
        CHK(set_device(s->eng));
        HIPCHK(hipMemcpyAsync(s->dev[s->cur].as<uint8_t>() + s->carry + s->fill, s->stage[k].p, len,
                              hipMemcpyHostToDevice, s->copy_stream));
        HIPCHK(hipEventRecord(s->stage_ev[k], s->copy_stream));
        s->stage_idx ^= 1;
    }
    s->fill += len;
    s->written += len;
    if (s->fill == s->window) CHK(stream_flush(s, false));
    return PBSGPU_OK;
}

int pbsgpu_stream_cut(pbsgpu_stream *s, uint64_t inject_bytes) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->finished) return PBSGPU_E_STATE;
    CHK(stream_flush(s, true));
    s->base += inject_bytes;
    s->inject_total += inject_bytes;
    s->section++;
    return PBSGPU_OK;
}

int pbsgpu_stream_finish(pbsgpu_stream *s) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->finished) return PBSGPU_OK;
    CHK(stream_flush(s, true));
    CHK(stream_drain(s));
    s->finished = true;
    return PBSGPU_OK;
}

int pbsgpu_stream_poll(pbsgpu_stream *s, pbsgpu_record *out, uint64_t cap, uint64_t *n) {
    if (!s || !n || (!out && cap)) return PBSGPU_E_INVALID;
    CHK(stream_reap(s));
    uint64_t k = 0;
    while (k < cap && !s->out.empty()) {
        out[k++] = s->out.front();
        s->out.pop_front();
    }
    *n = k;
    return PBSGPU_OK;
}

int pbsgpu_stream_position(const pbsgpu_stream *s, uint64_t *bytes_written) {
    if (!s || !bytes_written) return PBSGPU_E_INVALID;
    *bytes_written = s->written;
    return PBSGPU_OK;
}

// ---- chunker ----------------------------------------------------------------------------
int pbsgpu_chunker_create(pbsgpu_engine *e, pbsgpu_chunker **out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    *out = nullptr;
    pbsgpu_chunker *c = new (std::nothrow) pbsgpu_chunker();
    if (!c) return PBSGPU_E_NOMEM;
    c->eng = e;
    int st;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        st = set_device(e);
        if (st == PBSGPU_OK) st = c->dev.ensure(kChunkerSlice + 256);
        if (st == PBSGPU_OK) st = c->host.ensure(kChunkerSlice + 256);
    }
    if (st != PBSGPU_OK) {
        pbsgpu_chunker_destroy(c);
        return st;
    }
    *out = c;
    return PBSGPU_OK;
}

void pbsgpu_chunker_destroy(pbsgpu_chunker *c) {
    if (!c) return;
    if (c->eng) {
        std::lock_guard<std::mutex> lk(c->eng->mu);
        (void)hipSetDevice(c->eng->device);
        c->dev.release();
        c->host.release();
    }
    delete c;
}

int pbsgpu_chunker_reset(pbsgpu_chunker *c) {
    if (!c) return PBSGPU_E_INVALID;
    c->chunk_size = 0;
    c->tail_len = 0;
    return PBSGPU_OK;
}

int pbsgpu_chunker_scan(pbsgpu_chunker *c, const void *data, size_t len, size_t *pos) {
    if (!c || !pos || (!data && len)) return PBSGPU_E_INVALID;
    pbsgpu_engine *e = c->eng;
    const uint8_t *p = static_cast<const uint8_t *>(data);
    *pos = 0;
    size_t off = 0;
    while (off < len) {
        // no boundary is possible before chunk_size reaches effmin: consume without scanning
        if (c->chunk_size + 1 < e->effmin) {
            const size_t skip = (size_t)std::min<uint64_t>(len - off, e->effmin - 1 - c->chunk_size);
            chunker_push_tail(c, p + off, skip);
            c->chunk_size += skip;
            off += skip;
            continue;
        }
        // slice never extends past the forced cut at max
        const uint64_t to_max = (uint64_t)e->cfg.max - c->chunk_size;  // >= 1
        const size_t n = (size_t)std::min<uint64_t>(std::min<uint64_t>(len - off, kChunkerSlice - 64), to_max);
        uint64_t ncand = 0;
        const uint32_t tl = c->tail_len;
        {
            std::lock_guard<std::mutex> lk(e->mu);
            CHK(set_device(e));
            Slot *slot = find_free_slot(e);
            if (!slot) return PBSGPU_E_BUSY;
            uint8_t *h = c->host.as<uint8_t>();
            std::memcpy(h, c->tail, tl);
            std::memcpy(h + tl, p + off, n);
            HIPCHK(hipMemcpyAsync(c->dev.p, h, tl + n, hipMemcpyHostToDevice, slot->stream));
            CHK(candidates_sync(e, *slot, c->dev.as<uint8_t>(), tl + n, &ncand));
            c->cands.resize((size_t)ncand);
            if (ncand) HIPCHK(hipMemcpy(c->cands.data(), slot->dense.p, (size_t)ncand * 8, hipMemcpyDeviceToHost));
        }
        // first candidate end (in slice coordinates) that satisfies the min rule
        uint64_t cut = 0;
        for (uint64_t i = 0; i < ncand; ++i) {
            const uint64_t endc = c->cands[(size_t)i];  // exclusive end in [tail|slice] coordinates
            if (endc <= tl) continue;
            const uint64_t eslice = endc - tl;  // bytes of this slice consumed at the cut
            if (c->chunk_size + eslice >= e->effmin) {
                cut = eslice;
                break;
            }
        }
        if (cut == 0 && n == to_max) cut = n;  // chunk_size reaches max inside this slice
        if (cut) {
            c->chunk_size = 0;
            c->tail_len = 0;
            *pos = off + (size_t)cut;
            return PBSGPU_OK;
        }
        chunker_push_tail(c, p + off, n);
        c->chunk_size += n;
        off += n;
    }
    return PBSGPU_OK;
}

// ---- payload-stream assembly -----------------------------------------------------------------
int pbsgpu_payload_pack_device(pbsgpu_engine *e, const void *src, uint64_t src_bytes, const pbsgpu_segment *files,
                               uint32_t nfiles, const pbsgpu_payload_format *fmt, void *dst, uint64_t dst_cap,
                               uint64_t *out_len, uint64_t *payload_offsets) {
    if (!e || !fmt || !dst || !out_len || (nfiles && (!files || !src))) return PBSGPU_E_INVALID;
    uint64_t need = 0;
    CHK(pbsgpu_payload_size(files, nfiles, fmt, &need));
    *out_len = need;
    if (dst_cap < need) return PBSGPU_E_CAPACITY;
    for (uint32_t i = 0; i < nfiles; ++i)
        if (files[i].length > src_bytes || files[i].offset > src_bytes - files[i].length) return PBSGPU_E_INVALID;
    constexpr uint64_t kPiece = 4ull << 20;
    std::vector<pbsk::PackItem> items;
    items.reserve((size_t)nfiles * 2 + (size_t)(need / kPiece) + 4);
    uint64_t pos = 0;
    auto header = [&](uint64_t type, uint64_t size) {
        items.push_back(pbsk::PackItem{type, pos, size, 1u, 0u});
        pos += 16;
    };
    if (fmt->with_start) header(fmt->start_type, 16);
    for (uint32_t i = 0; i < nfiles; ++i) {
        if (payload_offsets) payload_offsets[i] = pos;
        header(fmt->payload_type, 16 + files[i].length);
        for (uint64_t o = 0; o < files[i].length; o += kPiece) {
            const uint64_t n = std::min<uint64_t>(kPiece, files[i].length - o);
            items.push_back(pbsk::PackItem{files[i].offset + o, pos + o, n, 0u, 0u});
        }
        pos += files[i].length;
    }
    if (fmt->with_tail) header(fmt->tail_type, 16);
    if (items.size() >= (1ull << 31)) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    const size_t bytes = items.size() * sizeof(pbsk::PackItem);
    CHK(s->tile_slots.ensure(bytes + 64));
    CHK(staged_h2d(e, s->tile_slots.p, items.data(), bytes, s->stream));
    HIPCHK(pbsk::launch_pack(static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst),
                             s->tile_slots.as<pbsk::PackItem>(), (uint32_t)items.size(), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PBSGPU_OK;
}

// ---- digest-set duplicate detection ---------------------------------------------------------
int pbsgpu_dedup_host(pbsgpu_engine *e, const pbsgpu_record *recs, uint64_t n, uint8_t *dup,
                      pbsgpu_dedup_stats *stats) {
    if (!e || (!recs && n) || !stats) return PBSGPU_E_INVALID;
    if (n >= (1ull << 32)) return PBSGPU_E_INVALID;
    std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return PBSGPU_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    const size_t tmp_bytes = pbsk::dedup_tmp_bytes(n);
    // layout inside slot buffers: recs | keys | keys_alt | idx | idx_alt | dup | stats | tmp
    CHK(s->recs.ensure((size_t)n * sizeof(pbsgpu_record)));
    CHK(s->dense.ensure((size_t)n * 16 + 64));
    CHK(s->tile_slots.ensure((size_t)n * 8 + 64));
    CHK(s->tile_cnt.ensure((size_t)n + 64));
    CHK(s->scalars.ensure(SC_COUNT * 4 + 64));
    CHK(s->scan_tmp.ensure(tmp_bytes));
    CHK(s->h_scalars.ensure(64));
    CHK(staged_h2d(e, s->recs.p, recs, n * sizeof(pbsgpu_record), s->stream));
    uint64_t *keys = s->dense.as<uint64_t>();
    uint64_t *keys_alt = keys + n;
    uint32_t *idx = s->tile_slots.as<uint32_t>();
    uint32_t *idx_alt = idx + n;
    uint8_t *d_dup = s->tile_cnt.as<uint8_t>();
    uint64_t *d_stats = reinterpret_cast<uint64_t *>(s->scalars.as<uint8_t>() + 32);
    HIPCHK(pbsk::launch_dedup(s->recs.as<pbsgpu_record>(), n, keys, idx, keys_alt, idx_alt, d_dup, d_stats,
                              s->scan_tmp.p, tmp_bytes, s->stream));
    HIPCHK(hipMemcpyAsync(s->h_scalars.p, d_stats, 32, hipMemcpyDeviceToHost, s->stream));
    if (dup) HIPCHK(hipMemcpyAsync(dup, d_dup, (size_t)n, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    const uint64_t *hs = s->h_scalars.as<uint64_t>();
    stats->nrecords = hs[0];
    stats->nunique = hs[1];
    stats->total_bytes = hs[2];
    stats->unique_bytes = hs[3];
    return PBSGPU_OK;
}

// ---- dynamic index (.didx) -----------------------------------------------------------------
// Proxmox Backup dynamic index layout (the format datastore.ParseDynamicIndex reads):
//   header, 4096 bytes: magic[8] | uuid[16] | ctime i64 LE | index_csum[32] | reserved
//   entries, 40 bytes each: end u64 LE | digest[32]
//   index_csum = SHA-256 over the concatenated entries.
extern const uint8_t pbsgpu_didx_magic[8];  // hostonly.cpp

int pbsgpu_didx_encode(pbsgpu_engine *e, const pbsgpu_record *recs, uint64_t n, const uint8_t uuid[16],
                       int64_t ctime, uint8_t *out, uint64_t cap) {
    if (!e || (!recs && n) || !out) return PBSGPU_E_INVALID;
    const uint64_t need = PBSGPU_DIDX_HEADER_SIZE + n * 40;
    if (cap < need) return PBSGPU_E_CAPACITY;
    std::memset(out, 0, PBSGPU_DIDX_HEADER_SIZE);
    std::memcpy(out, pbsgpu_didx_magic, 8);
    if (uuid) std::memcpy(out + 8, uuid, 16);
    for (int i = 0; i < 8; ++i) out[24 + i] = (uint8_t)((uint64_t)ctime >> (8 * i));
    uint8_t *ent = out + PBSGPU_DIDX_HEADER_SIZE;
    uint64_t prev = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (recs[i].end < prev) return PBSGPU_E_INVALID;  // one stream, ascending ends
        prev = recs[i].end;
        for (int b = 0; b < 8; ++b) ent[i * 40 + b] = (uint8_t)(recs[i].end >> (8 * b));
        std::memcpy(ent + i * 40 + 8, recs[i].digest, 32);
    }
    // index checksum on the device (same SHA-256 kernel as the chunk digests)
    pbsgpu_segment seg{0, n * 40};
    return pbsgpu_sha256_many_host(e, ent, n * 40, &seg, 1, out + 32);
}

}  // extern "C"
