// libpbsgpu host side, part 2: the streaming front ends and the record-set utilities.
//
//  * pbsgpu_stream_*  — the payload-stream seam transfer.ArchiveWriter.WriteEntryReader
//    feeds (reference internal/pxarmount/commit_reuse.go:427-468, commit_walk.go:465-479,
//    internal/tapeio/converter.go:827-842): bytes are appended to one continuous stream and
//    (end, digest) records fall out in order, exactly what the module appends to the .didx.
//    A client of the engine's page ring (ring.cpp) since round 4.
//  * pbsgpu_chunker_* — upstream `scan(data) -> pos` compatibility (buzhash.Config's chunker).
//  * pbsgpu_dedup_host — digest-set duplicate detection on the device.
//  * pbsgpu_didx_*    — dynamic index encode/decode (commit_bottleneck_test.go:773-793).
// All byte-stream work runs through the same HIP kernels as the batch path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "ring_internal.h"

using namespace pbse;

// -------------------------------------------------------------------------------------
// caller bytes -> pinned staging, on several threads for large writes
// -------------------------------------------------------------------------------------
namespace pbse {

namespace {
struct CopyPool {
    struct Job { uint8_t *dst; const uint8_t *src; size_t n; };
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Job> jobs;
    size_t pending = 0;
    std::vector<std::thread> threads;
    bool stop = false;
    int nthreads = 0;

    void start(int n) {
        nthreads = n;
        for (int i = 0; i < n; ++i)
            threads.emplace_back([this] {
                for (;;) {
                    Job j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_work.wait(lk, [this] { return stop || !jobs.empty(); });
                        if (stop && jobs.empty()) return;
                        j = jobs.front();
                        jobs.pop_front();
                    }
                    std::memcpy(j.dst, j.src, j.n);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : threads) t.join();
    }
};
}  // namespace

void parallel_memcpy(void *dst, const void *src, size_t n) {
    static const int want = []() {
        const char *v = getenv("PBSGPU_COPY_THREADS");
        return std::min(16, std::max(1, v ? atoi(v) : 4));
    }();
    constexpr size_t kMin = 4u << 20;  // below this one thread is as fast as the hand-over
    if (want <= 1 || n < kMin) {
        std::memcpy(dst, src, n);
        return;
    }
    static CopyPool pool;              // helpers only: the caller copies a slice itself
    static std::once_flag once;
    std::call_once(once, [] { pool.start(want - 1); });
    static std::mutex serial;          // one large copy at a time uses the helpers; concurrent writers fall back to their own thread
    std::unique_lock<std::mutex> only(serial, std::try_to_lock);
    if (!only.owns_lock()) {
        std::memcpy(dst, src, n);
        return;
    }
    const size_t parts = (size_t)want;
    const size_t slice = ((n / parts) + 4095) & ~(size_t)4095;
    uint8_t *d = static_cast<uint8_t *>(dst);
    const uint8_t *s = static_cast<const uint8_t *>(src);
    size_t off = std::min(slice, n);   // [0, off) is the caller's own slice
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        for (size_t o = off; o < n; o += slice) {
            pool.jobs.push_back(CopyPool::Job{d + o, s + o, std::min(slice, n - o)});
            pool.pending++;
        }
    }
    pool.cv_work.notify_all();
    std::memcpy(d, s, off);
    std::unique_lock<std::mutex> lk(pool.mu);
    pool.cv_done.wait(lk, [] { return pool.pending == 0; });
}

}  // namespace pbse

// -------------------------------------------------------------------------------------
// streaming writer — a client of the engine's page ring (ring.cpp)
// -------------------------------------------------------------------------------------
// Rounds 1-3 ran this seam on an engine of its own (a private ring of window buffers per stream, deferred window cuts with
// a headroom carry, six engine-wide hash-job lanes whose launches each lasted one max-size chunk chain). Since round 4 a
// payload stream owns nothing but its STAGING: caller bytes -> pinned staging -> H2D straight into a page reserved from
// the engine's page ring. Cutting (rounds over the new pages of all streams), hashing (the persistent SHA-256 service: a
// chunk starts the moment it is cut), page release (a page is free again when its last chunk has been READ) and record
// delivery are the ring's, shared with pbsgpu_ring_* callers and by all streams of the engine. What the seam adds on top:
//   * sections: InjectChunks (pbsgpu_stream_cut) ends the current ring stream with a forced cut and the next bytes open
//     a new one; records carry the section as `segment` and payload positions (written + injected) as `end`;
//   * suggested boundaries in payload coordinates, forwarded to the section they fall into;
//   * the per-file XXH3-64 tee and the pxar payload entry headers, run on a page's bytes before the page is committed;
//   * records and file hashes are collected by WHICHEVER stream of the engine calls next (one lock per ring), so a writer
//     that sits in a blocking read never holds up the others.
namespace {

struct Section {                         // the bytes between two forced cuts = one stream of the ring
    uint32_t rid = 0;                    // ring stream slot (valid until ring_done)
    uint32_t index = 0;                  // section number: the records' `segment`
    uint64_t base = 0;                   // payload position of its first byte
    bool input_closed = false;           // its last page has been committed
    bool ring_done = false;              // the ring has delivered its last record (slot closed)
    std::deque<pbsgpu_record> recs;      // filled under ring->mu by whoever drains the ring
};

struct FileSpan {                        // a file body inside the stream (begin_file .. end_file), in WRITTEN-byte coordinates
    uint64_t index = 0, w_start = 0, w_end = 0;
    bool closed = false, started = false;
    bool done = false;                   // its last piece has been queued
    uint32_t state = 0;                  // which of the two streaming XXH3 states carries it across pages
    uint32_t pend = 0;                   // host mirror of xxh::State::pend_len (pure arithmetic on the piece lengths)
};

struct TeeLaunch {                       // item tables of one tee launch (a small ring of them: launches overlap)
    PinnedBuf h_items, h_out;            // mapped: item table (read by the upload kernel), hashes (written by the tee kernel)
    DevBuf d_items;
    hipEvent_t done = nullptr;           // recorded behind the launch; the tables are reusable once it has completed
    bool busy = false;
    std::vector<pbsgpu_file_hash> files; // files whose last piece is in this launch (xxh3 filled in at harvest)
    std::vector<uint32_t> out_slot;      // ... and its index in h_out
};

constexpr size_t kStreamStage = 32u << 20;
constexpr int kStreamStages = 3;
constexpr int kTeeLaunches = 8;

}  // namespace

struct pbsgpu_stream {
    pbsgpu_engine *eng = nullptr;
    pbsgpu_ring *ring = nullptr;         // the engine's page ring
    uint64_t window = 0;                 // as given to create (recycling key; the ring's page size is what matters)
    // staging: caller bytes are gathered here (cgo pointer rule) and copied H2D piecewise
    PinnedBuf stage[kStreamStages];
    hipEvent_t stage_ev[kStreamStages][2] = {};   // last copy out of the buffer on each of the engine's copy streams
    bool stage_used[kStreamStages][2] = {};
    int stage_idx = 0;
    size_t stage_fill = 0;               // bytes gathered in stage[stage_idx] and not yet pushed (small writes are coalesced)
    int reserved = -1;                   // staging buffer handed out by pbsgpu_stream_reserve
    // the page being filled
    bool have_page = false;
    uint8_t *page_ptr = nullptr;
    uint64_t page_fill = 0, page_cap = 0;
    uint64_t page_w0 = 0;                // written-byte coordinate of the page's first byte
    int page_cs = 0;                     // engine copy stream all of this page's copies ride on
    // sections
    std::deque<std::unique_ptr<Section>> sections;  // oldest first; the last one may be `cur`
    Section *cur = nullptr;              // section that takes new bytes (null until the first byte behind a cut)
    uint32_t section_index = 0;
    uint64_t written = 0, inject_total = 0;
    uint64_t landed = 0;                 // written bytes that have been copied into pages so far
    bool finished = false;               // input closed
    bool drained = false;                // ... and every record has been moved to `out`
    int error = PBSGPU_OK;               // sticky (a HIP / ring error; never the bytes' fault)
    std::deque<uint64_t> suggested;      // announced boundaries (payload positions, ascending) not yet behind the stream
    size_t sugg_fwd = 0;                 // how many of them the current section's ring stream already knows
    std::deque<pbsgpu_record> out;
    // per-file XXH3-64 tee
    std::deque<FileSpan> files;          // files not yet completely hashed, in stream order
    uint64_t next_file = 0;
    uint32_t n_stateful = 0;
    bool file_open = false;
    uint64_t entry_left = 0;             // begin_entry: content bytes still expected
    bool in_entry = false;
    hipStream_t tee_stream = nullptr;    // one of the engine's (all tee launches of this stream: in order)
    DevBuf tee_states, tee_queue, tee_sums;
    TeeLaunch tee[kTeeLaunches];
    uint32_t tee_next = 0;               // next launch slot
    std::deque<int> tee_pending;         // launch slots in flight, oldest first
    std::deque<pbsgpu_file_hash> file_out;
};

namespace {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- the engine's ring ----------------------------------------------------------------------------------------------------
int engine_ring_get(pbsgpu_engine *e, pbsgpu_ring **out) {
    std::lock_guard<std::mutex> lk(e->sring_mu);
    if (!e->sring) {
        CHK(set_device(e));
        pbsgpu_ring_options o{};
        // Arena: ingest rate x residency. A host-fed engine moves <= ~55 GiB/s (PCIe) and a page stays for queue wait + the
        // chain of the longest chunk touching it (<= 0.46 s, ~0.3 s on average): 48 GiB is generous (pbsgpu_engine_options::stream_ring_gib).
        // Service: 50 GiB/s need ~12 CUs of chains (4.3 GiB/s per CU); 32 leave headroom for bursts and keep 7/8 of the chip
        // for everything else the process runs (pbsgpu_engine_options::stream_sha_cus).
        const pbsgpu_engine_options &eo = e->opt;  // (defaults resolved by pbsgpu_engine_create_opt)
        const double gib = eo.stream_ring_gib;
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        uint64_t want = (uint64_t)(gib * 1073741824.0);
        const uint64_t keep = 4ull << 30;  // tables, other allocations of the process
        if (fr > keep && want > fr - keep) want = fr - keep;
        o.arena_bytes = want;
        int cus = std::max(1, std::min(32, e->num_cus / 2));
        if (eo.stream_sha_cus) cus = std::max(1, std::min((int)eo.stream_sha_cus, e->num_cus - 1));
        o.sha_cus = (uint32_t)cus;
        // ... and 8 more for the EXPRESS service: what an archive waits for at its end is the serial SHA-256 chain of its last
        // long chunks (0.49 s for a 16 MiB chunk on a pair lane); two lanes per chunk finish it in 0.36 s, and a host-fed
        // engine has CUs to spare (12.5 % of 50 GiB/s in chunks >= 10 MiB need 3 express CUs)
        int xp = std::min(8, std::max(0, e->num_cus / 2 - cus));
        if (eo.stream_express_cus) xp = eo.stream_express_cus == 0xffffffffu ? 0 : std::min((int)eo.stream_express_cus, e->num_cus / 2);
        o.express_cus = xp ? (uint32_t)xp : PBSGPU_RING_OFF;
        o.max_streams = eo.stream_ring_slots;
        o.page_bytes = eo.stream_page_bytes;
        // nothing in flight anywhere for 2 ms: stop the service (its CUs, and hipFree / device-wide syncs of the process, come
        // back); the next page starts it again
        o.autopark_ms = 2.0;
        // ... and a writer that simply stops calling (a blocking read) gives them back after 2 s without any call at all
        o.idle_timeout_s = 2.0;
        // host-fed pages trickle in (3 per millisecond at 50 GiB/s) and a round is three launches: cut every 8 pages instead of
        // waiting for a quarter of a full round (64 pages = 20 ms more latency for every chunk, and for the archive's drain)
        o.min_round_pages = 8;
        pbsgpu_ring *r = nullptr;
        // ... for every chunk of at least half the maximum size: what an archive waits for at its end is then a 16 MiB chunk on
        // an express pair (0.36 s) rather than a 13 MiB one on a pair lane (0.40 s); 23 % of 50 GiB/s keep 4-5 express CUs busy
        CHK(ring_create_internal(e, &o, false, &r, xp ? (uint32_t)(e->cfg.max / 2) : 0u));
        e->sring = r;
    }
    e->sring_users++;
    *out = e->sring;
    return PBSGPU_OK;
}

void engine_ring_put(pbsgpu_engine *e) {
    std::lock_guard<std::mutex> lk(e->sring_mu);
    if (e->sring_users > 0) e->sring_users--;
}

// ring->mu held: pump the ring and hand every stream of the engine what is ready for it — whoever calls collects for all
int ring_drain(pbsgpu_ring *r) {
    const int st = pbsgpu_ring_pump(r);
    thread_local std::vector<pbsgpu_record> buf(4096);
    for (uint32_t si = 0; si < r->slots.size(); ++si) {
        StreamSlot &sl = r->slots[si];
        if (!sl.open) continue;
        Section *sec = static_cast<Section *>(sl.owner);  // null: its stream was destroyed in an error state — discard
        for (;;) {
            uint64_t n = 0;
            ring_pop_records(r, si, buf.data(), buf.size(), &n);
            if (sec) sec->recs.insert(sec->recs.end(), buf.begin(), buf.begin() + (long)n);
            if (n < buf.size()) break;
        }
        if (sl.final_done && sl.cells.empty()) {
            if (sec) sec->ring_done = true;
            sl.owner = nullptr;
            (void)pbsgpu_ring_close(r, si);
        }
    }
    while (!r->rounds.empty() && r->rounds.front().reaped && r->rounds.front().live_cells == 0) r->rounds.pop_front();
    return st;
}

// ring->mu held: this stream's delivered records -> out (payload coordinates), finished sections retired
void stream_collect(pbsgpu_stream *s) {
    while (!s->sections.empty()) {
        Section *sec = s->sections.front().get();
        for (auto &rec : sec->recs) {
            pbsgpu_record o = rec;
            o.end = rec.end + sec->base;
            o.segment = sec->index;
            s->out.push_back(o);
        }
        sec->recs.clear();
        if (!sec->ring_done) break;  // records come out in stream order: later sections wait
        if (sec == s->cur) s->cur = nullptr;
        s->sections.pop_front();
    }
}

int stream_pump(pbsgpu_stream *s) {
    std::lock_guard<std::mutex> lk(s->ring->mu);
    const int st = ring_drain(s->ring);
    stream_collect(s);
    if (st != PBSGPU_OK && s->error == PBSGPU_OK) s->error = st;
    return s->error;
}

// ---- tee ---------------------------------------------------------------------------------------------------------------------
// results of finished tee launches -> file_out (in file order); block: wait for all of them
int stream_harvest_tees(pbsgpu_stream *s, bool block) {
    while (!s->tee_pending.empty()) {
        TeeLaunch &t = s->tee[s->tee_pending.front()];
        if (block) {
            HIPCHK(hipEventSynchronize(t.done));
        } else {
            const hipError_t q = hipEventQuery(t.done);
            if (q == hipErrorNotReady) {
                (void)hipGetLastError();
                break;
            }
            HIPCHK(q);
        }
        for (size_t i = 0; i < t.files.size(); ++i) {
            t.files[i].xxh3 = t.h_out.as<uint64_t>()[t.out_slot[i]];
            s->file_out.push_back(t.files[i]);
        }
        t.files.clear();
        t.out_slot.clear();
        t.busy = false;
        s->tee_pending.pop_front();
    }
    return PBSGPU_OK;
}

// Queue the per-file XXH3 pieces of the written-byte range [w0, w1) that lies at `base` in device memory (the page being
// committed; empty range: only zero-length last pieces) on the stream's tee stream, behind `after` (the page's copy).
// *launched = the tee ran something: then `done_ev` (from the ring's pool, already fetched by the caller) has been recorded
// behind it and the page must wait for THAT instead of its copy.
int stream_enqueue_tee(pbsgpu_stream *s, const uint8_t *base, uint64_t w0, uint64_t w1, hipEvent_t after, hipEvent_t done_ev,
                       bool *launched) {
    *launched = false;
    if (s->files.empty()) return PBSGPU_OK;
    pbsgpu_engine *e = s->eng;
    TeeLaunch &t = s->tee[s->tee_next % kTeeLaunches];
    if (t.busy) {  // the oldest launch still owns these tables: take its results first (launches complete in order)
        while (t.busy) CHK(stream_harvest_tees(s, true));
    }
    const size_t maxitems = s->files.size();
    CHK(t.h_items.ensure(std::max<size_t>(maxitems, 1024) * sizeof(pbsk::XxhItem)));
    CHK(t.h_out.ensure(std::max<size_t>(maxitems, 1024) * 8));
    pbsk::XxhItem *items = t.h_items.as<pbsk::XxhItem>();
    uint32_t n = 0;
    uint64_t total_blocks = 0;
    for (auto &f : s->files) {
        if (f.w_start > w1 || (f.w_start == w1 && !(f.closed && f.w_end == f.w_start))) break;  // starts behind this range
        const uint64_t lo = std::max(f.w_start, w0);
        const uint64_t hi = f.closed ? std::min(f.w_end, w1) : w1;
        const uint64_t len = hi > lo ? hi - lo : 0;
        const bool first = !f.started && lo == f.w_start;
        const bool last = f.closed && hi == f.w_end;
        if (len == 0 && !last) continue;  // nothing of it here yet
        pbsk::XxhItem it{};
        it.ptr = base + (lo - w0);
        it.len = len;
        it.flags = (first ? 1u : 0u) | (last ? 2u : 0u);
        if (first && !last) f.state = s->n_stateful++ & 1u;  // at most two files span a page edge at any time
        it.state = f.state;
        // the plan: which 1 KiB blocks this piece completes (XXH3 keeps the final 1..1024 bytes for its tail rules)
        if (first && last) {
            it.pend = 0;
            it.nproc = len > 240 ? (uint32_t)((len - 1) / 1024) : 0u;
        } else {
            if (first) f.pend = 0;
            const uint64_t T = (uint64_t)f.pend + len;
            it.pend = f.pend;
            it.nproc = T ? (uint32_t)((T - 1) / 1024) : 0u;
            f.pend = (uint32_t)(T - (uint64_t)it.nproc * 1024);
        }
        it.s_off = total_blocks;
        total_blocks += it.nproc;
        f.started = true;
        if (last) {
            f.done = true;
            const uint32_t slot = (uint32_t)t.files.size();
            it.out = slot;
            t.files.push_back(pbsgpu_file_hash{f.index, f.w_end - f.w_start, 0});
            t.out_slot.push_back(slot);
        }
        items[n++] = it;
    }
    // (a CLOSED file may still have bytes in staging, beyond the page that is being committed: only `done` retires it)
    while (!s->files.empty() && s->files.front().done) s->files.pop_front();
    if (n == 0) return PBSGPU_OK;
    CHK(t.d_items.ensure(std::max<size_t>(maxitems, 1024) * sizeof(pbsk::XxhItem)));
    CHK(s->tee_sums.ensure((size_t)(total_blocks + 64) * 64));
    hipStream_t ts = s->tee_stream;
    if (after) HIPCHK(hipStreamWaitEvent(ts, after, 0));
    HIPCHK(hipMemsetAsync(s->tee_queue.p, 0, 64, ts));
    HIPCHK(pbsk::launch_publish(t.d_items.p, items, (size_t)n * sizeof(pbsk::XxhItem), ts));  // host -> device by kernel
    HIPCHK(pbsk::launch_xxh3_items(t.d_items.as<pbsk::XxhItem>(), n, total_blocks, s->tee_states.p, s->tee_sums.as<uint64_t>(),
                                   t.h_out.as<uint64_t>(), s->tee_queue.as<uint32_t>(), e->num_cus, ts));
    HIPCHK(hipEventRecord(t.done, ts));
    if (done_ev) HIPCHK(hipEventRecord(done_ev, ts));
    t.busy = true;
    s->tee_pending.push_back((int)(s->tee_next % kTeeLaunches));
    s->tee_next++;
    *launched = true;
    return PBSGPU_OK;
}

// ---- pages -------------------------------------------------------------------------------------------------------------------
// a page to write into: opens the section's ring stream at its first byte; waits (pumping) while the ring has no page or
// no stream slot free — back-pressure: the ingest outruns the SHA-256 service
int stream_acquire_page(pbsgpu_stream *s) {
    pbsgpu_ring *r = s->ring;
    const double t0 = now_ms();
    for (int spin = 0;; ++spin) {
        {
            std::lock_guard<std::mutex> lk(r->mu);
            const int st = ring_drain(r);
            stream_collect(s);
            if (st != PBSGPU_OK) return s->error = st;
            if (s->error != PBSGPU_OK) return s->error;
            bool ok = true;
            if (!s->cur) {
                uint32_t rid = 0;
                const int so = pbsgpu_ring_open(r, &rid);
                if (so == PBSGPU_OK) {
                    std::unique_ptr<Section> sec(new (std::nothrow) Section());
                    if (!sec) return PBSGPU_E_NOMEM;
                    sec->rid = rid;
                    sec->index = s->section_index;
                    sec->base = s->landed + s->inject_total;  // payload position of the next byte to land
                    r->slots[rid].owner = sec.get();
                    r->slots[rid].origin = sec->base;
                    s->cur = sec.get();
                    s->sections.push_back(std::move(sec));
                    s->sugg_fwd = 0;
                } else if (so == PBSGPU_E_BUSY) {
                    ok = false;  // every slot is taken by sections that are still being hashed
                } else {
                    return so;
                }
            }
            if (ok) {
                void *dptr = nullptr;
                uint64_t cap = 0;
                const int sr = pbsgpu_ring_reserve(r, s->cur->rid, &dptr, &cap);
                if (sr == PBSGPU_OK) {
                    s->have_page = true;
                    s->page_ptr = static_cast<uint8_t *>(dptr);
                    s->page_cap = cap;
                    s->page_fill = 0;
                    s->page_w0 = s->landed;
                    s->page_cs = (int)(s->eng->copy_rr.fetch_add(1, std::memory_order_relaxed) % s->eng->copy_streams.size());
                    return PBSGPU_OK;
                }
                if (sr != PBSGPU_E_BUSY) return s->error = sr;
            }
        }
        (void)stream_harvest_tees(s, false);
        if (spin < 64) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (now_ms() - t0 > 120000.0) return s->error = PBSGPU_E_STATE;  // two minutes without a page: the device is gone
    }
}

// announced boundaries the current section's ring stream does not know yet, up to `upto` (payload position); ring->mu held
int stream_forward_suggestions(pbsgpu_stream *s, uint64_t upto) {
    if (!s->cur) return PBSGPU_OK;
    while (!s->suggested.empty() && s->suggested.front() <= s->cur->base) {  // at or before the section's first byte: no cut there
        s->suggested.pop_front();
        if (s->sugg_fwd) s->sugg_fwd--;
    }
    // ... and what the ring already knows and the stream has left more than a maximum chunk behind
    const uint64_t behind = s->landed + s->inject_total;
    while (s->sugg_fwd > 0 && !s->suggested.empty() && s->suggested.front() + s->eng->cfg.max < behind) {
        s->suggested.pop_front();
        s->sugg_fwd--;
    }
    while (s->sugg_fwd < s->suggested.size() && s->suggested[s->sugg_fwd] <= upto) {
        CHK(pbsgpu_ring_suggest(s->ring, s->cur->rid, s->suggested[s->sugg_fwd] - s->cur->base));
        s->sugg_fwd++;
    }
    return PBSGPU_OK;
}

// hand the current page (page_fill bytes; every page but a section's last is full) to the ring
int stream_commit_page(pbsgpu_stream *s, bool final) {
    pbsgpu_ring *r = s->ring;
    pbsgpu_engine *e = s->eng;
    hipEvent_t ev_copy = nullptr, ev_tee = nullptr;
    const bool bytes = s->have_page && s->page_fill > 0;
    const bool tee = !s->files.empty();
    {
        std::lock_guard<std::mutex> lk(r->mu);
        if (bytes) CHK(ring_event_get(r, &ev_copy));
        if (tee) CHK(ring_event_get(r, &ev_tee));
    }
    if (bytes) HIPCHK(hipEventRecord(ev_copy, e->copy_streams[(size_t)s->page_cs]));  // behind the page's last copy
    hipEvent_t dep = ev_copy;
    if (tee) {
        bool launched = false;
        const uint8_t *base = bytes ? s->page_ptr : s->tee_states.as<uint8_t>();
        const uint64_t w0 = bytes ? s->page_w0 : s->landed, w1 = bytes ? s->page_w0 + s->page_fill : s->landed;
        CHK(stream_enqueue_tee(s, base, w0, w1, ev_copy, ev_tee, &launched));
        if (launched && bytes) dep = ev_tee;  // the tee reads the page: the cut (and with it the page's release) waits for it
    }
    std::lock_guard<std::mutex> lk(r->mu);
    if (dep != ev_copy && ev_copy) ring_event_put(r, ev_copy);  // (the tee stream's wait has captured it)
    if (dep != ev_tee && ev_tee) ring_event_put(r, ev_tee);
    if (!s->cur) {  // nothing was ever written behind the last cut: there is no ring stream to end
        if (dep) ring_event_put(r, dep);
        return PBSGPU_OK;
    }
    const uint64_t end_pos = s->landed + s->inject_total;  // payload position behind the committed bytes
    const uint64_t look = e->sugg_feed.load(std::memory_order_relaxed) > 1 ? (uint64_t)e->cfg.max : 0;
    CHK(stream_forward_suggestions(s, end_pos + look));
    const int st = ring_commit_dep(r, s->cur->rid, bytes ? s->page_fill : 0, final ? 1 : 0, bytes ? dep : nullptr);
    if (!bytes && dep) ring_event_put(r, dep);
    if (st != PBSGPU_OK) return s->error = st;
    s->have_page = false;
    s->page_fill = 0;
    if (final) {
        s->cur->input_closed = true;
        s->cur = nullptr;  // (the section stays in `sections` until its last record is out)
    }
    const int sd = ring_drain(r);
    stream_collect(s);
    if (sd != PBSGPU_OK) return s->error = sd;
    return PBSGPU_OK;
}

// H2D of one staged piece of n bytes (stage[k][0..n)) into the stream's pages. `written` was advanced when the bytes were
// accepted.
int stream_push_piece(pbsgpu_stream *s, int k, size_t n) {
    pbsgpu_engine *e = s->eng;
    const uint8_t *src = s->stage[k].as<uint8_t>();
    size_t off = 0;
    s->stage_idx = (k + 1) % kStreamStages;
    while (off < n) {
        if (!s->have_page) CHK(stream_acquire_page(s));
        const size_t m = (size_t)std::min<uint64_t>(n - off, s->page_cap - s->page_fill);
        hipStream_t cs = e->copy_streams[(size_t)s->page_cs];
        HIPCHK(hipMemcpyAsync(s->page_ptr + s->page_fill, src + off, m, hipMemcpyHostToDevice, cs));
        HIPCHK(hipEventRecord(s->stage_ev[k][s->page_cs], cs));  // the staging buffer is reusable after this
        s->stage_used[k][s->page_cs] = true;
        s->page_fill += m;
        s->landed += m;
        off += m;
        if (s->page_fill == s->page_cap) CHK(stream_commit_page(s, false));
    }
    return PBSGPU_OK;
}

// push whatever small writes have gathered in the current staging buffer (before a cut / finish / reserve)
int stream_push_staged(pbsgpu_stream *s) {
    if (s->stage_fill == 0) return PBSGPU_OK;
    const size_t n = s->stage_fill;
    s->stage_fill = 0;
    return stream_push_piece(s, s->stage_idx, n);
}

// staging buffer k is about to be written by the host: its previous H2D copies have drained
int stream_stage_ready(pbsgpu_stream *s, int k) {
    CHK(s->stage[k].ensure(kStreamStage));
    for (int c = 0; c < 2; ++c)
        if (s->stage_used[k][c]) {
            HIPCHK(hipEventSynchronize(s->stage_ev[k][c]));
            s->stage_used[k][c] = false;
        }
    return PBSGPU_OK;
}

// forced cut / end of input: the section's last (short) page goes out with the final flag
int stream_end_section(pbsgpu_stream *s) {
    CHK(stream_push_staged(s));
    if (!s->cur && !s->have_page && s->files.empty()) return PBSGPU_OK;
    return stream_commit_page(s, true);
}

bool stream_all_delivered(const pbsgpu_stream *s) { return s->sections.empty(); }

// this stream has nothing in the ring any more: if nobody else has either, stop the ring's service NOW (not after the idle
// timer) — a caller that synchronises the device or frees memory right after finish() must not wait for a kernel that
// only ends on request
void stream_maybe_park(pbsgpu_stream *s) {
    std::lock_guard<std::mutex> lk(s->ring->mu);
    if (ring_idle(s->ring)) (void)ring_park(s->ring);
}

// wait until every record of every section is in `out` (the serial SHA-256 chain of the last chunks: up to ~0.46 s)
int stream_drain(pbsgpu_stream *s) {
    const double t0 = now_ms();
    for (int spin = 0;; ++spin) {
        const int st = stream_pump(s);
        if (st != PBSGPU_OK) return st;
        if (stream_all_delivered(s)) break;
        if (spin < 64) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (now_ms() - t0 > 120000.0) return s->error = PBSGPU_E_STATE;
    }
    CHK(stream_harvest_tees(s, true));
    stream_maybe_park(s);
    return PBSGPU_OK;
}

}  // namespace

// -------------------------------------------------------------------------------------
// upstream-style chunker
// -------------------------------------------------------------------------------------
struct pbsgpu_chunker {
    pbsgpu_engine *eng = nullptr;
    Slot ctx;                 // private scan context
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

static pbsgpu_stream *stream_unpark(pbsgpu_engine *e, uint64_t window_bytes);  // contexts of closed streams are recycled, see below
void pbsgpu_stream_destroy(pbsgpu_stream *s);

int pbsgpu_stream_create(pbsgpu_engine *e, uint64_t window_bytes, pbsgpu_stream **out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    *out = nullptr;
    (void)window_bytes;  // (rounds 1-3: bytes per device window; the engine's page ring has ONE page size for all its streams)
    pbsgpu_ring *ring = nullptr;
    CHK(engine_ring_get(e, &ring));
    if (pbsgpu_stream *parked = stream_unpark(e, 0)) {
        parked->ring = ring;
        *out = parked;
        return PBSGPU_OK;
    }
    pbsgpu_stream *s = new (std::nothrow) pbsgpu_stream();
    if (!s) {
        engine_ring_put(e);
        return PBSGPU_E_NOMEM;
    }
    engine_ref(e);
    s->eng = e;
    s->ring = ring;
    int st = set_device(e);
    // everything a stream can need is allocated here or grows by doubling: a free later (hipFree / hipHostFree) would
    // wait for the whole device, and is parked anyway while the ring's service runs (engine_internal.h: dev_free)
    for (int k = 0; k < kStreamStages && st == PBSGPU_OK; ++k)
        for (int c = 0; c < 2 && st == PBSGPU_OK; ++c)
            if (hipEventCreateWithFlags(&s->stage_ev[k][c], hipEventDisableTiming) != hipSuccess) st = PBSGPU_E_HIP;
    for (auto &t : s->tee)
        if (st == PBSGPU_OK && hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) st = PBSGPU_E_HIP;
    if (st == PBSGPU_OK) st = s->tee_states.ensure(2 * pbsk::xxh3_state_bytes() + 256);
    if (st == PBSGPU_OK) st = s->tee_queue.ensure(64);
    if (st == PBSGPU_OK) st = s->tee_sums.ensure((size_t)(ring->page_bytes / 1024 + 64) * 64);
    s->tee_stream = e->tee_streams[e->tee_rr.fetch_add(1, std::memory_order_relaxed) % e->tee_streams.size()];
    if (st != PBSGPU_OK) {
        s->error = st;  // (never parked: see pbsgpu_stream_destroy)
        pbsgpu_stream_destroy(s);
        return st;
    }
    *out = s;
    return PBSGPU_OK;
}

// ---- stream contexts are recycled --------------------------------------------------------------------------------
// What a stream owns (3 x 32 MiB pinned staging, the tee's buffers, a dozen events) is parked in the engine when the
// stream closes and handed to the engine's next stream: allocating 100 MB of pinned memory costs tens of milliseconds per
// archive, and freeing it (hipFree / hipHostFree wait for the whole device) would stall a writer that closes archive k
// while archive k+1 is already being hashed. Only contexts that were fully created and never saw an error are parked.
static void stream_free_context(pbsgpu_stream *s) {
    for (auto &b : s->stage) b.release();
    for (auto &row : s->stage_ev)
        for (auto &ev : row)
            if (ev) (void)hipEventDestroy(ev);
    for (auto &t : s->tee) {
        t.h_items.release();
        t.h_out.release();
        t.d_items.release();
        if (t.done) (void)hipEventDestroy(t.done);
    }
    s->tee_states.release();
    s->tee_queue.release();
    s->tee_sums.release();
}

// back to the state pbsgpu_stream_create leaves a stream in (the resources stay)
static void stream_reset_state(pbsgpu_stream *s) {
    for (auto &row : s->stage_used)
        for (auto &u : row) u = false;
    s->stage_idx = 0;
    s->stage_fill = 0;
    s->reserved = -1;
    s->have_page = false;
    s->page_ptr = nullptr;
    s->page_fill = s->page_cap = s->page_w0 = 0;
    s->sections.clear();
    s->cur = nullptr;
    s->section_index = 0;
    s->written = s->inject_total = s->landed = 0;
    s->finished = s->drained = false;
    s->error = PBSGPU_OK;
    s->suggested.clear();
    s->sugg_fwd = 0;
    s->out.clear();
    s->files.clear();
    s->next_file = 0;
    s->n_stateful = 0;
    s->file_open = false;
    s->entry_left = 0;
    s->in_entry = false;
    s->tee_pending.clear();
    for (auto &t : s->tee) {
        t.busy = false;
        t.files.clear();
        t.out_slot.clear();
    }
    s->file_out.clear();
}

static bool stream_park(pbsgpu_engine *e, pbsgpu_stream *s) {
    const size_t limit = e->opt.stream_ctx_pool;
    stream_reset_state(s);
    s->eng = nullptr;  // (a parked context holds no reference: the engine owns it)
    s->ring = nullptr;
    std::lock_guard<std::mutex> lk(e->pool_mu);
    if (e->destroyed || e->stream_pool.size() >= limit) return false;
    e->stream_pool.push_back(s);
    return true;
}

static pbsgpu_stream *stream_unpark(pbsgpu_engine *e, uint64_t) {
    pbsgpu_stream *s = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->pool_mu);
        if (!e->stream_pool.empty()) {
            s = e->stream_pool.back();
            e->stream_pool.pop_back();
        }
    }
    if (!s) return nullptr;
    engine_ref(e);
    s->eng = e;
    return s;
}

}  // extern "C"
namespace pbse {
void stream_pool_release(pbsgpu_engine *e) {
    std::vector<pbsgpu_stream *> v;
    {
        std::lock_guard<std::mutex> lk(e->pool_mu);
        v.swap(e->stream_pool);
    }
    for (pbsgpu_stream *s : v) {
        stream_free_context(s);
        delete s;
    }
}

// caller holds e->sring_mu, or is the engine's teardown
void engine_ring_release(pbsgpu_engine *e) {
    if (e->sring) {
        pbsgpu_ring_destroy(e->sring);
        e->sring = nullptr;
    }
}
}  // namespace pbse
extern "C" {

void pbsgpu_stream_destroy(pbsgpu_stream *s) {
    if (!s) return;
    pbsgpu_engine *e = s->eng;
    if (e) {
        (void)hipSetDevice(e->device);
        bool healthy = s->error == PBSGPU_OK;
        if (s->ring) {
            // An unfinished stream still has to END its ring streams (their pages and slots only come back through a final
            // round), and every section has to hand out its last record before its slot is free: the same wait as finish.
            if (healthy && !s->finished) {
                s->file_open = false;
                s->in_entry = false;
                s->reserved = -1;
                healthy = stream_end_section(s) == PBSGPU_OK;
                s->finished = true;
            }
            if (healthy) healthy = stream_drain(s) == PBSGPU_OK;
            if (!healthy) {  // a failed section / ring: drop what the ring still routes to this stream
                std::lock_guard<std::mutex> lk(s->ring->mu);
                for (auto &sec : s->sections)
                    if (!sec->ring_done && sec->rid < s->ring->slots.size() && s->ring->slots[sec->rid].owner == sec.get()) {
                        StreamSlot &sl = s->ring->slots[sec->rid];
                        sl.owner = nullptr;
                        if (!sl.final_committed) (void)ring_commit_dep(s->ring, sec->rid, 0, 1, nullptr);
                        // (the slot stays open until its ring stream has ended; the next drain by any stream of the
                        // engine discards what it still delivers and closes it — ring_drain)
                    }
            }
            for (int k = 0; k < kStreamStages; ++k)
                for (int c = 0; c < 2; ++c)
                    if (s->stage_used[k][c] && s->stage_ev[k][c]) (void)hipEventSynchronize(s->stage_ev[k][c]);  // copies out of our staging
            (void)stream_harvest_tees(s, true);
            stream_maybe_park(s);
            engine_ring_put(e);
        }
        if (healthy && stream_park(e, s)) {  // the context waits for the engine's next stream: nothing is freed, nothing waits
            engine_unref(e);
            return;
        }
        stream_free_context(s);
    }
    delete s;
    if (e) engine_unref(e);
}

int pbsgpu_stream_write(pbsgpu_stream *s, const void *data, size_t len) {
    if (!s || (!data && len)) return PBSGPU_E_INVALID;
    if (s->finished || s->reserved >= 0) return PBSGPU_E_STATE;
    if (s->error != PBSGPU_OK) return s->error;
    if (s->in_entry && len > s->entry_left) return PBSGPU_E_INVALID;  // more bytes than the entry header announced
    if (len) CHK(set_device(s->eng));
    const uint8_t *p = static_cast<const uint8_t *>(data);
    while (len) {
        // caller bytes -> library-owned pinned staging -> device page (async). Writes are COALESCED in the staging
        // buffer (an io.Copy feeds 32 KiB at a time, a payload header is 16 bytes): one H2D piece per 32 MiB, not per
        // call. The memcpy runs on the caller's thread with no lock held, so several streams copy in parallel.
        const int k = s->stage_idx;
        if (s->stage_fill == 0) CHK(stream_stage_ready(s, k));
        const size_t n = std::min(len, kStreamStage - s->stage_fill);
        parallel_memcpy(s->stage[k].as<uint8_t>() + s->stage_fill, p, n);
        s->stage_fill += n;
        s->written += n;
        if (s->in_entry) s->entry_left -= std::min<uint64_t>(s->entry_left, n);
        p += n;
        len -= n;
        if (s->stage_fill == kStreamStage) CHK(stream_push_staged(s));
    }
    return PBSGPU_OK;
}

int pbsgpu_stream_reserve(pbsgpu_stream *s, void **buf, size_t *cap) {
    if (!s || !buf || !cap) return PBSGPU_E_INVALID;
    if (s->finished || s->reserved >= 0) return PBSGPU_E_STATE;
    if (s->error != PBSGPU_OK) return s->error;
    CHK(set_device(s->eng));
    CHK(stream_push_staged(s));  // bytes gathered by earlier small writes go first; the reserved buffer starts empty
    const int k = s->stage_idx;
    CHK(stream_stage_ready(s, k));
    s->reserved = k;
    *buf = s->stage[k].p;
    uint64_t room = kStreamStage;
    if (s->in_entry) room = std::min(room, s->entry_left);
    *cap = (size_t)room;
    return PBSGPU_OK;
}

int pbsgpu_stream_commit(pbsgpu_stream *s, size_t len) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->reserved < 0) return PBSGPU_E_STATE;
    const int k = s->reserved;
    if (len > kStreamStage) return PBSGPU_E_INVALID;
    if (s->in_entry && len > s->entry_left) return PBSGPU_E_INVALID;
    s->reserved = -1;
    if (len == 0) return PBSGPU_OK;
    CHK(set_device(s->eng));
    s->written += len;
    if (s->in_entry) s->entry_left -= std::min<uint64_t>(s->entry_left, len);
    return stream_push_piece(s, k, len);
}

int pbsgpu_stream_suggest(pbsgpu_stream *s, uint64_t offset) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->finished) return PBSGPU_E_STATE;
    if (!s->suggested.empty() && offset < s->suggested.back()) return PBSGPU_E_INVALID;  // ascending, like the channel
    s->suggested.push_back(offset);  // forwarded to the section it falls into when that section's next page is committed
    return PBSGPU_OK;
}

// ---- per-file XXH3-64 tee (writeBackedFile: tee := io.TeeReader(f, xxh3.New()), commit_reuse.go:450-461) ----------
int pbsgpu_stream_begin_file(pbsgpu_stream *s) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->finished || s->file_open || s->reserved >= 0) return PBSGPU_E_STATE;
    FileSpan f;
    f.index = s->next_file++;
    f.w_start = f.w_end = s->written;
    s->files.push_back(f);
    s->file_open = true;
    return PBSGPU_OK;
}

int pbsgpu_stream_end_file(pbsgpu_stream *s, uint64_t *index) {
    if (!s) return PBSGPU_E_INVALID;
    if (!s->file_open || s->reserved >= 0) return PBSGPU_E_STATE;
    FileSpan &f = s->files.back();
    f.w_end = s->written;
    f.closed = true;
    s->file_open = false;
    if (index) *index = f.index;
    return PBSGPU_OK;
}

int pbsgpu_stream_poll_files(pbsgpu_stream *s, pbsgpu_file_hash *out, uint64_t cap, uint64_t *n) {
    if (!s || !n || (!out && cap)) return PBSGPU_E_INVALID;
    if (s->eng) {
        CHK(set_device(s->eng));
        CHK(stream_harvest_tees(s, false));
    }
    uint64_t k = 0;
    while (k < cap && !s->file_out.empty()) {
        out[k++] = s->file_out.front();
        s->file_out.pop_front();
    }
    *n = k;
    return PBSGPU_OK;
}

// ---- pxar payload entries: 16-byte {type, 16 + size} header in front of every file body ----------------------------
static int stream_write_header(pbsgpu_stream *s, uint64_t type, uint64_t full_size) {
    uint8_t h[16];
    for (int i = 0; i < 8; ++i) {
        h[i] = (uint8_t)(type >> (8 * i));
        h[8 + i] = (uint8_t)(full_size >> (8 * i));
    }
    return pbsgpu_stream_write(s, h, 16);
}

int pbsgpu_stream_write_marker(pbsgpu_stream *s, const pbsgpu_payload_format *fmt, int tail) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->in_entry || s->file_open) return PBSGPU_E_STATE;
    pbsgpu_payload_format f;
    if (fmt) f = *fmt; else (void)pbsgpu_payload_format_default(&f);
    return stream_write_header(s, tail ? f.tail_type : f.start_type, 16);
}

int pbsgpu_stream_begin_entry(pbsgpu_stream *s, const pbsgpu_payload_format *fmt, uint64_t content_len,
                              uint64_t *payload_offset) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->finished || s->in_entry || s->file_open || s->reserved >= 0) return PBSGPU_E_STATE;
    pbsgpu_payload_format f;
    if (fmt) f = *fmt; else (void)pbsgpu_payload_format_default(&f);
    if (payload_offset) *payload_offset = s->written + s->inject_total;  // what WriteEntryRef / PAYLOAD_REF records
    CHK(stream_write_header(s, f.payload_type, 16 + content_len));
    CHK(pbsgpu_stream_begin_file(s));
    s->in_entry = true;
    s->entry_left = content_len;
    return PBSGPU_OK;
}

int pbsgpu_stream_end_entry(pbsgpu_stream *s, uint64_t *file_index) {
    if (!s) return PBSGPU_E_INVALID;
    if (!s->in_entry || s->reserved >= 0) return PBSGPU_E_STATE;
    if (s->entry_left != 0) return PBSGPU_E_STATE;  // short body: WriteEntryReader's "unexpected EOF"
    s->in_entry = false;
    return pbsgpu_stream_end_file(s, file_index);
}

int pbsgpu_stream_cut(pbsgpu_stream *s, uint64_t inject_bytes) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->finished || s->reserved >= 0 || s->in_entry) return PBSGPU_E_STATE;
    if (s->error != PBSGPU_OK) return s->error;
    CHK(set_device(s->eng));
    CHK(stream_end_section(s));  // the open chunk is closed where the stream stands; the next byte opens a new ring stream
    s->inject_total += inject_bytes;
    s->section_index++;
    return PBSGPU_OK;
}

// close the input: the tail chunk is cut, every chunk is with the SHA-256 service; nothing here waits for a hash
static int stream_close_input(pbsgpu_stream *s) {
    if (s->finished) return PBSGPU_OK;
    if (s->reserved >= 0 || s->file_open) return PBSGPU_E_STATE;
    if (s->error != PBSGPU_OK) return s->error;
    CHK(set_device(s->eng));
    CHK(stream_end_section(s));
    s->finished = true;
    return PBSGPU_OK;
}

int pbsgpu_stream_finish(pbsgpu_stream *s) {
    if (!s) return PBSGPU_E_INVALID;
    CHK(stream_close_input(s));
    if (s->drained) return PBSGPU_OK;
    CHK(stream_drain(s));
    s->drained = true;
    return PBSGPU_OK;
}

int pbsgpu_stream_finish_begin(pbsgpu_stream *s) {
    if (!s) return PBSGPU_E_INVALID;
    return stream_close_input(s);
}

int pbsgpu_stream_done(pbsgpu_stream *s, int *done) {
    if (!s || !done) return PBSGPU_E_INVALID;
    *done = 0;
    if (!s->finished) return PBSGPU_OK;
    if (!s->drained) {
        CHK(set_device(s->eng));
        CHK(stream_pump(s));
        CHK(stream_harvest_tees(s, false));
        s->drained = stream_all_delivered(s) && s->tee_pending.empty();
        if (s->drained) stream_maybe_park(s);
    }
    *done = s->drained ? 1 : 0;
    return PBSGPU_OK;
}

int pbsgpu_stream_poll(pbsgpu_stream *s, pbsgpu_record *out, uint64_t cap, uint64_t *n) {
    if (!s || !n || (!out && cap)) return PBSGPU_E_INVALID;
    int st = PBSGPU_OK;
    if (s->eng && s->ring && !s->drained) {
        CHK(set_device(s->eng));
        st = stream_pump(s);
    }
    uint64_t k = 0;
    while (k < cap && !s->out.empty()) {
        out[k++] = s->out.front();
        s->out.pop_front();
    }
    *n = k;
    return (k == 0) ? st : PBSGPU_OK;  // what was cut before a failure is still delivered; the error follows
}

int pbsgpu_stream_position(const pbsgpu_stream *s, uint64_t *position) {
    if (!s || !position) return PBSGPU_E_INVALID;
    // Encoder().PayloadPosition() semantics (commit_reuse.go:265): bytes written PLUS bytes injected — InjectChunks
    // advances the payload position by the injected sizes (keepLast_chunk_test.go: enc.Advance(total)), and the
    // record `end` offsets live in the same coordinates
    *position = s->written + s->inject_total;
    return PBSGPU_OK;
}

int pbsgpu_stream_bytes_written(const pbsgpu_stream *s, uint64_t *bytes_written) {
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
    engine_ref(e);
    c->eng = e;
    int st = set_device(e);
    if (st == PBSGPU_OK) st = c->ctx.init();
    if (st == PBSGPU_OK) st = c->dev.ensure(kChunkerSlice + 256);
    if (st == PBSGPU_OK) st = c->host.ensure(kChunkerSlice + 256);
    if (st != PBSGPU_OK) {
        pbsgpu_chunker_destroy(c);
        return st;
    }
    *out = c;
    return PBSGPU_OK;
}

void pbsgpu_chunker_destroy(pbsgpu_chunker *c) {
    if (!c) return;
    pbsgpu_engine *e = c->eng;
    if (e) {
        (void)hipSetDevice(e->device);
        c->ctx.destroy();
        c->dev.release();
        c->host.release();
    }
    delete c;
    if (e) engine_unref(e);
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
            CHK(set_device(e));
            Slot *slot = &c->ctx;
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
    CHK(set_device(e));
    AuxLease lease(e);
    Slot *s = lease.s;
    const size_t bytes = items.size() * sizeof(pbsk::PackItem);
    CHK(s->tile_slots.ensure(bytes + 64));
    CHK(staged_h2d(*s, s->tile_slots.p, items.data(), bytes, s->stream));
    HIPCHK(pbsk::launch_pack(static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst),
                             s->tile_slots.as<pbsk::PackItem>(), (uint32_t)items.size(), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PBSGPU_OK;
}

// ---- piece-table copy (synthetic corpus editing) -------------------------------------------------
int pbsgpu_gather_device(pbsgpu_engine *e, const void *src, uint64_t src_bytes, void *dst, uint64_t dst_bytes,
                         const pbsgpu_copy_item *its, uint32_t nitems) {
    if (!e || (nitems && (!its || !src || !dst))) return PBSGPU_E_INVALID;
    if (nitems == 0) return PBSGPU_OK;
    constexpr uint64_t kPiece = 4ull << 20;
    std::vector<pbsk::PackItem> items;
    for (uint32_t i = 0; i < nitems; ++i) {
        if (its[i].len > src_bytes || its[i].src_off > src_bytes - its[i].len) return PBSGPU_E_INVALID;
        if (its[i].len > dst_bytes || its[i].dst_off > dst_bytes - its[i].len) return PBSGPU_E_INVALID;
        for (uint64_t o = 0; o < its[i].len; o += kPiece)
            items.push_back(pbsk::PackItem{its[i].src_off + o, its[i].dst_off + o, std::min<uint64_t>(kPiece, its[i].len - o), 0u, 0u});
    }
    if (items.size() >= (1ull << 31)) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    AuxLease lease(e);
    Slot *s = lease.s;
    const size_t bytes = items.size() * sizeof(pbsk::PackItem);
    CHK(s->tile_slots.ensure(bytes + 64));
    CHK(staged_h2d(*s, s->tile_slots.p, items.data(), bytes, s->stream));
    HIPCHK(pbsk::launch_pack(static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst),
                             s->tile_slots.as<pbsk::PackItem>(), (uint32_t)items.size(), s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PBSGPU_OK;
}

// ---- digest-set duplicate detection ---------------------------------------------------------
static int dedup_common(pbsgpu_engine *e, const pbsgpu_record *recs, bool on_device, uint64_t n, uint8_t *dup,
                        pbsgpu_dedup_stats *stats);

int pbsgpu_dedup_host(pbsgpu_engine *e, const pbsgpu_record *recs, uint64_t n, uint8_t *dup,
                      pbsgpu_dedup_stats *stats) {
    return dedup_common(e, recs, false, n, dup, stats);
}

// the records are already in device memory (e.g. the output of an RCCL all-gather): no host round trip of the set
int pbsgpu_dedup_device(pbsgpu_engine *e, const void *drecs, uint64_t n, uint8_t *dup, pbsgpu_dedup_stats *stats) {
    if (n && !is_device_pointer(drecs)) return PBSGPU_E_INVALID;
    return dedup_common(e, static_cast<const pbsgpu_record *>(drecs), true, n, dup, stats);
}

static int dedup_common(pbsgpu_engine *e, const pbsgpu_record *recs, bool on_device, uint64_t n, uint8_t *dup,
                        pbsgpu_dedup_stats *stats) {
    if (!e || (!recs && n) || !stats) return PBSGPU_E_INVALID;
    if (n >= (1ull << 32)) return PBSGPU_E_INVALID;
    std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return PBSGPU_OK;
    CHK(set_device(e));
    AuxLease lease(e);
    Slot *s = lease.s;
    const size_t tmp_bytes = pbsk::dedup_tmp_bytes(n);
    // layout inside slot buffers: recs | keys | keys_alt | idx | idx_alt | dup | stats | tmp
    if (!on_device) CHK(s->recs.ensure((size_t)n * sizeof(pbsgpu_record)));
    CHK(s->dense.ensure((size_t)n * 16 + 64));
    CHK(s->tile_slots.ensure((size_t)n * 8 + 64));
    CHK(s->tile_cnt.ensure((size_t)n + 64));
    CHK(s->scalars.ensure(SC_COUNT * 4 + 64));
    CHK(s->scan_tmp.ensure(tmp_bytes));
    CHK(s->h_scalars.ensure(64));
    const pbsgpu_record *drecs = recs;
    if (!on_device) {
        CHK(staged_h2d(*s, s->recs.p, recs, n * sizeof(pbsgpu_record), s->stream));
        drecs = s->recs.as<pbsgpu_record>();
    }
    uint64_t *keys = s->dense.as<uint64_t>();
    uint64_t *keys_alt = keys + n;
    uint32_t *idx = s->tile_slots.as<uint32_t>();
    uint32_t *idx_alt = idx + n;
    uint8_t *d_dup = s->tile_cnt.as<uint8_t>();
    uint64_t *d_stats = reinterpret_cast<uint64_t *>(s->scalars.as<uint8_t>() + 32);
    HIPCHK(pbsk::launch_dedup(drecs, n, keys, idx, keys_alt, idx_alt, d_dup, d_stats,
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
