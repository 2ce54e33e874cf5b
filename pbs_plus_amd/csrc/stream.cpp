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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "engine_internal.h"

using namespace pbse;

// -------------------------------------------------------------------------------------
// shared SHA-256 jobs (engine-wide)
// -------------------------------------------------------------------------------------
// Windows of every stream append their chunk descriptors to the engine's OPEN job; whoever finds a free hash lane
// seals the job and launches it (one k_sha256_pair launch over all accumulated chunks, longest first). A launch
// cannot finish before the serial chain of its longest chunk (up to ~0.43 s at 16 MiB), so with one launch per
// window the number of concurrent kernels would grow with the ingest rate and exhaust the hardware queues; with
// shared jobs it is bounded by the lane count while every chunk still starts within one lane-turnaround.
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

constexpr int kHashLanes = 6;  // + 2 copy streams + one stream per payload stream: within the 24 hardware queues for 8 writers

int hd_init(pbsgpu_engine *e) {
    HashDispatcher &hd = e->hd;
    hd.num_cus = e->num_cus;
    int nlanes = kHashLanes;
    if (const char *v = getenv("PBSGPU_HASH_LANES")) nlanes = std::min(16, std::max(1, atoi(v)));
    // + reserve lanes, taken only by a job somebody BLOCKS on (the last job of an archive, a writer whose ring is full)
    // when every regular lane is busy: lanes do not stay evenly staggered (jobs last 0.3-0.55 s depending on their longest
    // chunk), so without them such a job waits up to a whole job time for a lane
    int nreserve = 0;  // (measured: no gain for one writer, -15 % for eight, profiles/r03_hostfeed_hash_job_pacing.log)
    if (const char *v = getenv("PBSGPU_HASH_RESERVE_LANES")) nreserve = std::min(8, std::max(0, atoi(v)));
    hd.regular_lanes = nlanes;
    hd.lanes.assign((size_t)(nlanes + nreserve), nullptr);
    hd.lane_job.assign((size_t)(nlanes + nreserve), nullptr);
    // launches are SPACED by 0.8 x (chain of a max-size chunk) / lanes = 61 ms at 16 MiB chunks and 6 lanes (policy and
    // measurements: engine_internal.h); small maximum chunk sizes make the interval vanish
    hd.min_interval_ms = 0.8 * ((double)e->cfg.max / 64.0 * 1.75e-3) / (double)nlanes;
    hd.t0_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (const char *v = getenv("PBSGPU_HASH_INTERVAL_MS")) hd.min_interval_ms = atof(v);
    if (const char *v = getenv("PBSGPU_HASH_BYPASS_GIB")) hd.bypass_bytes = (uint64_t)(atof(v) * 1073741824.0);
    for (auto &st : hd.lanes) HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return PBSGPU_OK;
}

void hd_destroy(pbsgpu_engine *e) {
    HashDispatcher &hd = e->hd;
    for (auto st : hd.lanes)
        if (st) (void)hipStreamSynchronize(st);
    for (auto &j : hd.jobs) {
        j->d_queue.release();
        for (PinnedBuf *b : {&j->h_desc, &j->h_order, &j->h_dig}) b->release();
        if (j->done) (void)hipEventDestroy(j->done);
    }
    hd.jobs.clear();
    for (auto st : hd.lanes)
        if (st) (void)hipStreamDestroy(st);
    hd.lanes.clear();
}

}  // namespace pbse

namespace {

// hd.mu held: a job object that is neither open, running, nor still referenced by a window
HashJob *hd_new_job(HashDispatcher &hd) {
    for (auto &j : hd.jobs) {
        if (j.get() == hd.open || j->refs.load() != 0) continue;
        bool on_lane = false;
        for (auto *lj : hd.lane_job) on_lane |= (lj == j.get());
        if (on_lane) continue;
        j->descs.clear();
        j->n = 0;
        j->lane = -1;
        j->state = HashJob::OPEN;
        return j.get();
    }
    std::unique_ptr<HashJob> j(new (std::nothrow) HashJob());
    if (!j) return nullptr;
    if (hipEventCreateWithFlags(&j->done, hipEventDisableTiming) != hipSuccess) return nullptr;
    hd.jobs.push_back(std::move(j));
    return hd.jobs.back().get();
}

// hd.mu held. Seal the open job and launch it if a lane is free. Returns PBSGPU_OK with *launched = false when every
// lane is still busy (the caller may wait on *busy_ev, outside the lock, and retry).
static double hd_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int hd_try_launch_locked(pbsgpu_engine *e, bool *launched, hipEvent_t *busy_ev, bool force = false) {
    HashDispatcher &hd = e->hd;
    *launched = false;
    if (busy_ev) *busy_ev = nullptr;
    HashJob *j = hd.open;
    if (!j || j->descs.empty()) return PBSGPU_OK;
    // pacing (engine_internal.h): go at once when a good amount of work is waiting, otherwise keep launches spaced
    if (!force && hd.open_bytes < hd.bypass_bytes && hd_now_ms() - hd.last_launch_ms < hd.min_interval_ms) return PBSGPU_OK;
    int lane = -1;
    const int usable = force ? (int)hd.lanes.size() : hd.regular_lanes;
    for (int i = 0; i < usable && lane < 0; ++i) {
        HashJob *lj = hd.lane_job[i];
        if (!lj) {
            lane = i;
        } else {
            const hipError_t q = hipEventQuery(lj->done);
            if (q == hipSuccess) {
                hd.lane_job[i] = nullptr;
                lane = i;
            } else if (q != hipErrorNotReady) {
                g_last_hip_error.store((int)q);
                return PBSGPU_E_HIP;
            } else {
                (void)hipGetLastError();
            }
        }
    }
    if (lane < 0) {
        if (busy_ev) {  // the job launched first is the next to finish (all last about one max-size chunk chain)
            HashJob *oldest = hd.lane_job[0];
            for (int i = 0; i < usable; ++i)
                if (hd.lane_job[i]->launched_ms < oldest->launched_ms) oldest = hd.lane_job[i];
            *busy_ev = oldest->done;
        }
        return PBSGPU_OK;
    }
    const uint32_t n = (uint32_t)j->descs.size();
    hipStream_t st = hd.lanes[lane];
    const size_t room = std::max<size_t>(n, 8192);  // regrowing later would wait for the whole device (hipFree)
    CHK(j->h_desc.ensure(room * sizeof(pbsk::HashDesc)));
    CHK(j->h_order.ensure(room * 4));
    CHK(j->h_dig.ensure(room * 32));
    CHK(j->d_queue.ensure(64));
    std::memcpy(j->h_desc.p, j->descs.data(), (size_t)n * sizeof(pbsk::HashDesc));
    // longest first (the launch lasts as long as its longest chain) + the CU budget that keeps the launch at that bound
    uint32_t *ord = j->h_order.as<uint32_t>();
    for (uint32_t i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord, ord + n, [&](uint32_t a, uint32_t b) { return j->descs[a].len > j->descs[b].len; });
    uint64_t total_blocks = 0, longest = 1;
    for (const auto &d : j->descs) {
        const uint64_t blocks = (d.len + 8) / 64 + 1;
        total_blocks += blocks;
        longest = std::max(longest, blocks);
    }
    uint64_t lanes_needed = (total_blocks * 125 / 100 + longest - 1) / longest;
    unsigned wgs = (unsigned)std::min<uint64_t>((lanes_needed + 127) / 128, (uint64_t)hd.num_cus);
    HIPCHK(hipMemsetAsync(j->d_queue.p, 0, 64, st));
    // No copy engine anywhere in a job: the kernel reads descriptors + order from, and writes the digests to, mapped
    // pinned memory. A small H2D copy would queue behind the streams' payload pieces in the shared SDMA queues, and a
    // D2H copy parked behind this 0.4 s kernel would block that queue for every other stream (kernels.hip, k_publish).
    HIPCHK(pbsk::launch_sha256_descs(j->h_desc.as<pbsk::HashDesc>(), n, j->h_order.as<uint32_t>(), j->h_dig.as<uint8_t>(),
                                     j->d_queue.as<uint32_t>(), wgs,
                                     pbsk::sha256_dense_pays(total_blocks, longest, hd.num_cus), st));
    HIPCHK(hipEventRecord(j->done, st));
    j->n = n;
    j->lane = lane;
    j->state = HashJob::LAUNCHED;
    hd.lane_job[lane] = j;
    hd.open = nullptr;
    hd.open_bytes = 0;
    hd.last_launch_ms = hd_now_ms();
    j->launched_ms = hd.last_launch_ms;
    *launched = true;
    static const bool trace = getenv("PBSGPU_TRACE") != nullptr;
    if (trace) {
        int busy = 0;
        for (auto *lj : hd.lane_job) busy += lj != nullptr;
        fprintf(stderr, "[pbsgpu] t=%.1f ms hash job: %u chunks, %.1f MiB, longest %.1f MiB, %u workgroups, lane %d (%d busy)%s\n",
                hd.last_launch_ms - hd.t0_ms, n, total_blocks / 16384.0, longest / 16384.0, wgs, lane, busy, force ? " forced" : "");
    }
    return PBSGPU_OK;
}

// append `n` descriptors to the open job; returns the job and the index of the first one. Tries to launch at once.
int hd_append(pbsgpu_engine *e, const pbsk::HashDesc *d, uint32_t n, HashJob **job, uint32_t *first) {
    HashDispatcher &hd = e->hd;
    std::lock_guard<std::mutex> lk(hd.mu);
    if (!hd.open) {
        hd.open = hd_new_job(hd);
        if (!hd.open) return PBSGPU_E_NOMEM;
    }
    HashJob *j = hd.open;
    *first = (uint32_t)j->descs.size();
    j->descs.insert(j->descs.end(), d, d + n);
    for (uint32_t i = 0; i < n; ++i) hd.open_bytes += d[i].len;
    j->refs.fetch_add(1);
    *job = j;
    bool launched;
    return hd_try_launch_locked(e, &launched, nullptr);
}

// make sure `j` gets launched (it may still be the open job because every lane was busy when it was filled)
int hd_ensure_launched(pbsgpu_engine *e, HashJob *j, bool block) {
    HashDispatcher &hd = e->hd;
    for (;;) {
        hipEvent_t busy = nullptr;
        {
            std::lock_guard<std::mutex> lk(hd.mu);
            if (j->state == HashJob::LAUNCHED) return PBSGPU_OK;
            bool launched = false;
            CHK(hd_try_launch_locked(e, &launched, &busy, block));  // someone waits for this job: no pacing
            if (launched || j->state == HashJob::LAUNCHED) return PBSGPU_OK;
        }
        if (!block) return PBSGPU_OK;
        if (busy) HIPCHK(hipEventSynchronize(busy));  // a lane frees up when its (oldest) job ends
        else std::this_thread::sleep_for(std::chrono::milliseconds(2));  // pacing: streams finishing together share a job
    }
}

// -------------------------------------------------------------------------------------
// streaming writer
// -------------------------------------------------------------------------------------
// A stream owns everything it touches on the cut side (two private work contexts on ONE HIP stream, its window
// buffers, pinned staging): several streams of one engine run concurrently from different threads without ever
// taking the engine lock. The writer thread never waits for the GPU in steady state:
//   * a full window's cut (scan + resolve + the per-file XXH3 tee) is only ENQUEUED behind its last H2D piece; the
//     writer moves on to the next window buffer at once, writing new bytes behind a max-chunk-sized headroom;
//   * at the NEXT flush the previous cut (long finished) is read back: its complete chunks go to the engine's shared
//     hash jobs, its still-open tail chunk is copied into the headroom in front of the new bytes (a cut only depends
//     on bytes before it, so re-examining the open chunk with more data reproduces the serial chunker);
//   * digests arrive asynchronously with the shared jobs; records are delivered strictly in stream order.
struct WindowInFlight {
    int buf = -1;
    HashJob *job = nullptr;
    uint32_t first = 0;                  // index of this window's first descriptor in the job
    std::vector<pbsgpu_record> recs;     // end (absolute) / size / section filled in; digests arrive with the job
};

struct FileSpan {                        // a file body inside the stream (begin_file .. end_file), in WRITTEN-byte coordinates
    uint64_t index = 0, w_start = 0, w_end = 0;
    bool closed = false, started = false;
    uint32_t state = 0;                  // which of the two streaming XXH3 states carries it across windows
    uint32_t pend = 0;                   // host mirror of xxh::State::pend_len (pure arithmetic on the piece lengths)
};

struct PendingCut {                      // a window whose cut has been enqueued but not read back yet
    bool active = false;
    int ctx = 0, buf = -1;
    uint64_t base = 0;                   // absolute stream offset (incl. injected bytes) of the window's first byte
    uint64_t carry = 0, total = 0;       // the window is [headroom - carry, headroom - carry + total) of its buffer
    uint32_t section = 0;
    bool final = false;
    std::vector<uint64_t> sugg_rel;      // suggested boundaries relative to the window start (for a capacity retry)
    std::vector<pbsgpu_file_hash> files; // files whose last piece was in this window (xxh3 filled in at read-back)
    std::vector<uint32_t> file_out;      // ... and the tee output slot of each
};

constexpr size_t kStreamStage = 32u << 20;
constexpr int kStreamStages = 3;

}  // namespace

struct pbsgpu_stream {
    pbsgpu_engine *eng = nullptr;
    uint64_t window = 0;           // new bytes per device window
    uint64_t headroom = 0;         // = max chunk size: room for the carried open chunk in front of the new bytes
    size_t devcap = 0;
    size_t max_bufs = 0;           // ring limit (back-pressure beyond it)
    std::vector<DevBuf> dev;       // window buffers: [headroom (carry right-aligned) | new bytes]
    std::vector<char> dev_busy;    // held by a window whose chunks are still being hashed, or by the pending cut
    int cur = 0;
    uint64_t carry = 0;            // bytes of the still-open chunk in front of the new bytes of dev[cur]
    uint64_t fill = 0;             // new bytes already copied into dev[cur]
    Slot cut[2];                   // private cut contexts, alternating; cut[1] borrows cut[0]'s HIP stream
    int cut_next = 0;
    hipStream_t hs = nullptr;      // == cut[0].stream: cuts, the tee and the carry copy, in order
    // H2D payload pieces ride the ENGINE's shared copy streams and never wait for a kernel: a copy that depends on a
    // kernel parks at the head of the SDMA queue it shares with other copies and stalls all of them (measured: 8
    // producers with copies on their cut streams reached 22 GiB/s, 2 producers 46). hs waits for the pieces' events.
    hipEvent_t piece_ev[2] = {};   // last piece of the current window on each of the engine's copy streams
    bool piece_used[2] = {false, false};
    PendingCut pend;
    PinnedBuf stage[kStreamStages];
    hipEvent_t stage_ev[kStreamStages] = {};
    int stage_idx = 0;
    size_t stage_fill = 0;         // bytes gathered in stage[stage_idx] and not yet pushed (small writes are coalesced)
    int reserved = -1;             // staging buffer handed out by pbsgpu_stream_reserve
    uint64_t base = 0;             // absolute stream offset (incl. injected bytes) of the carry's first byte
    uint64_t written = 0;
    uint64_t inject_total = 0;
    uint32_t section = 0;
    bool finished = false;  // input closed (the final flush is enqueued)
    bool drained = false;   // ... and every record has been delivered to `out`
    std::deque<uint64_t> suggested;  // pending suggested boundaries (absolute offsets, ascending)
    std::deque<WindowInFlight> inflight;
    std::deque<pbsgpu_record> out;
    std::vector<pbsk::HashDesc> descs;
    // per-file XXH3-64 tee
    std::deque<FileSpan> files;      // files not yet completely hashed, in stream order
    uint64_t next_file = 0;
    uint32_t n_stateful = 0;
    bool file_open = false;
    uint64_t entry_left = 0;         // begin_entry: content bytes still expected
    bool in_entry = false;
    DevBuf tee_states, tee_queue, tee_items, tee_sums;
    PinnedBuf h_tee_items, h_tee_out;  // mapped: read / written by the tee kernel directly
    std::deque<pbsgpu_file_hash> file_out;
};

namespace {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// move the oldest window's records (now with digests) to the output queue, release its buffer
int stream_complete_oldest(pbsgpu_stream *s, bool block) {
    WindowInFlight &w = s->inflight.front();
    CHK(hd_ensure_launched(s->eng, w.job, block));
    {
        std::lock_guard<std::mutex> lk(s->eng->hd.mu);
        if (w.job->state != HashJob::LAUNCHED) return PBSGPU_E_BUSY;  // only when !block
    }
    if (block) {
        HIPCHK(hipEventSynchronize(w.job->done));
    } else {
        const hipError_t q = hipEventQuery(w.job->done);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            return PBSGPU_E_BUSY;
        }
        HIPCHK(q);
    }
    for (size_t i = 0; i < w.recs.size(); ++i) {
        std::memcpy(w.recs[i].digest, w.job->digest(w.first + (uint32_t)i), 32);
        s->out.push_back(w.recs[i]);
    }
    static const bool trace = getenv("PBSGPU_TRACE") != nullptr;
    if (trace && w.job->refs.load() == 1)  // the job's last window: the job is over
        fprintf(stderr, "[pbsgpu] t=%.1f ms hash job of lane %d (launched t=%.1f ms, %u chunks) delivered%s\n",
                hd_now_ms() - s->eng->hd.t0_ms, w.job->lane, w.job->launched_ms - s->eng->hd.t0_ms, w.job->n, block ? " (waited)" : "");
    w.job->refs.fetch_sub(1);
    s->dev_busy[w.buf] = 0;
    s->inflight.pop_front();
    return PBSGPU_OK;
}

// a window buffer of devcap bytes: from the engine's pool of returned buffers if one fits, else a fresh allocation
int stream_buffer_ensure(pbsgpu_stream *s, DevBuf &b) {
    if (b.cap >= s->devcap) return PBSGPU_OK;
    {
        std::lock_guard<std::mutex> lk(s->eng->pool_mu);
        auto &pool = s->eng->win_pool;
        for (size_t i = 0; i < pool.size(); ++i)
            if (pool[i].cap >= s->devcap && pool[i].cap <= s->devcap + s->devcap / 4) {
                b = std::move(pool[i]);
                pool.erase(pool.begin() + (long)i);
                return PBSGPU_OK;
            }
    }
    return b.ensure(s->devcap);
}

// a window buffer that is neither current nor held; grows the ring up to max_bufs, then waits for the oldest window
int stream_free_buffer(pbsgpu_stream *s, int *out) {
    for (;;) {
        bool retry = false;
        for (size_t i = 0; i < s->dev.size() && !retry; ++i)
            if (!s->dev_busy[i] && (int)i != s->cur) {
                const int st = stream_buffer_ensure(s, s->dev[i]);
                if (st == PBSGPU_OK) {
                    *out = (int)i;
                    return PBSGPU_OK;
                }
                // HBM is full (other streams' rings, resident batches): that is back-pressure, not a write error, as long
                // as this stream has windows in flight whose buffers come back. Drop the empty ring entry, stop growing,
                // and wait for the oldest window instead. Only a ring that cannot reach its minimum (current + one
                // more buffer) fails.
                if (st != PBSGPU_E_NOMEM || s->dev[i].p) return st;
                {   // buffers parked by destroyed streams are the first thing to give back
                    std::lock_guard<std::mutex> lk(s->eng->pool_mu);
                    if (!s->eng->win_pool.empty()) {
                        for (auto &b : s->eng->win_pool) b.release();
                        s->eng->win_pool.clear();
                        retry = true;
                        continue;
                    }
                }
                if (s->inflight.empty()) return PBSGPU_E_NOMEM;
                if ((int)i < s->cur) s->cur--;
                for (auto &w : s->inflight)
                    if (w.buf > (int)i) w.buf--;
                if (s->pend.active && s->pend.buf > (int)i) s->pend.buf--;
                s->dev.erase(s->dev.begin() + (long)i);
                s->dev_busy.erase(s->dev_busy.begin() + (long)i);
                s->max_bufs = std::max<size_t>(s->dev.size(), 2);
                CHK(stream_complete_oldest(s, true));
                retry = true;
            }
        if (retry) continue;
        if (s->dev.size() < s->max_bufs) {
            s->dev.emplace_back();
            s->dev_busy.push_back(0);
            continue;
        }
        if (s->inflight.empty()) return PBSGPU_E_STATE;
        CHK(stream_complete_oldest(s, true));  // back-pressure: the ingest outruns the hash jobs
    }
}

// queue the per-file XXH3 pieces of the current window's NEW bytes (written-byte range [w0, w0 + fill))
int stream_enqueue_tee(pbsgpu_stream *s, PendingCut &pc) {
    if (s->files.empty()) return PBSGPU_OK;
    pbsgpu_engine *e = s->eng;
    const uint64_t w0 = s->written - s->fill, w1 = s->written;
    const uint8_t *newp = s->dev[s->cur].as<uint8_t>() + s->headroom;
    const size_t maxitems = s->files.size();
    CHK(s->h_tee_items.ensure(std::max<size_t>(maxitems, 4096) * sizeof(pbsk::XxhItem)));
    CHK(s->h_tee_out.ensure(std::max<size_t>(maxitems, 4096) * 8));
    pbsk::XxhItem *items = s->h_tee_items.as<pbsk::XxhItem>();
    uint32_t n = 0;
    uint64_t total_blocks = 0;
    for (auto &f : s->files) {
        if (f.w_start > w1 || (f.w_start == w1 && !(f.closed && f.w_end == f.w_start))) break;  // starts behind this window
        const uint64_t lo = std::max(f.w_start, w0);
        const uint64_t hi = f.closed ? std::min(f.w_end, w1) : w1;
        const uint64_t len = hi > lo ? hi - lo : 0;
        const bool first = !f.started && lo == f.w_start;
        const bool last = f.closed && hi == f.w_end;
        if (len == 0 && !last) continue;  // nothing of it here yet
        pbsk::XxhItem it{};
        it.ptr = newp + (lo - w0);
        it.len = len;
        it.flags = (first ? 1u : 0u) | (last ? 2u : 0u);
        if (first && !last) f.state = s->n_stateful++ & 1u;  // at most two files span a window edge at any time
        it.state = f.state;
        it.out = n;
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
            pc.files.push_back(pbsgpu_file_hash{f.index, f.w_end - f.w_start, 0});
            pc.file_out.push_back(n);
        }
        items[n++] = it;
    }
    while (!s->files.empty() && s->files.front().closed) s->files.pop_front();  // closed => its last piece is queued now
    if (n == 0) return PBSGPU_OK;
    CHK(s->tee_states.ensure(2 * pbsk::xxh3_state_bytes()));
    CHK(s->tee_queue.ensure(64));
    CHK(s->tee_items.ensure(std::max<size_t>(maxitems, 4096) * sizeof(pbsk::XxhItem)));
    CHK(s->tee_sums.ensure((size_t)(s->window / 1024 + 64) * 64));  // presized at create: never regrown
    HIPCHK(hipMemsetAsync(s->tee_queue.p, 0, 64, s->hs));
    HIPCHK(pbsk::launch_publish(s->tee_items.p, items, (size_t)n * sizeof(pbsk::XxhItem), s->hs));  // host -> device by kernel
    HIPCHK(pbsk::launch_xxh3_items(s->tee_items.as<pbsk::XxhItem>(), n, total_blocks, s->tee_states.p,
                                   s->tee_sums.as<uint64_t>(), s->h_tee_out.as<uint64_t>(), s->tee_queue.as<uint32_t>(),
                                   e->num_cus, s->hs));
    return PBSGPU_OK;
}

// read back the pending window's cut: complete chunks -> shared hash jobs, open tail chunk -> headroom of dev[cur]
int stream_resolve_pending(pbsgpu_stream *s) {
    PendingCut &pc = s->pend;
    if (!pc.active) return PBSGPU_OK;
    pbsgpu_engine *e = s->eng;
    Slot &ctx = s->cut[pc.ctx];
    uint64_t nrec = 0;
    CHK(cut_finish(e, ctx, &nrec));  // waits for the cut AND the tee queued in front of it
    pc.active = false;
    for (size_t i = 0; i < pc.files.size(); ++i) {
        pc.files[i].xxh3 = s->h_tee_out.as<uint64_t>()[pc.file_out[i]];
        s->file_out.push_back(pc.files[i]);
    }
    pc.files.clear();
    pc.file_out.clear();
    uint8_t *const winp = s->dev[pc.buf].as<uint8_t>() + s->headroom - pc.carry;
    const uint64_t nemit = pc.final ? nrec : (nrec ? nrec - 1 : 0);
    const pbsgpu_record *hr = ctx.h_recs.as<pbsgpu_record>();
    if (nemit) {
        WindowInFlight w;
        w.buf = pc.buf;
        w.recs.resize((size_t)nemit);
        s->descs.resize((size_t)nemit);
        for (uint64_t i = 0; i < nemit; ++i) {
            s->descs[(size_t)i] = pbsk::HashDesc{winp + (hr[i].end - hr[i].size), hr[i].size};
            pbsgpu_record r{};
            r.end = hr[i].end + pc.base;
            r.size = hr[i].size;
            r.segment = pc.section;
            w.recs[(size_t)i] = r;
        }
        CHK(hd_append(e, s->descs.data(), (uint32_t)nemit, &w.job, &w.first));
        s->inflight.push_back(std::move(w));  // keeps pc.buf busy until its chunks are hashed
    } else {
        s->dev_busy[pc.buf] = 0;  // (a carry copy below still reads it: later writes to it follow on the same HIP stream)
    }
    if (pc.final || nrec == 0) {
        s->base = pc.base + pc.total;
        s->carry = 0;
    } else {
        const pbsgpu_record &open = hr[nrec - 1];
        const uint64_t open_start = open.end - open.size;
        HIPCHK(hipMemcpyAsync(s->dev[s->cur].as<uint8_t>() + s->headroom - open.size, winp + open_start, open.size,
                              hipMemcpyDeviceToDevice, s->hs));
        // a window without a complete chunk released its buffer above: the copy must have read it before new pieces
        // (other HIP stream) may overwrite it. Cannot happen in steady state (a full window holds >= max bytes).
        if (!nemit) HIPCHK(hipStreamSynchronize(s->hs));
        s->base = pc.base + open_start;
        s->carry = open.size;
    }
    return PBSGPU_OK;
}

// A window is complete (or the stream is being cut / finished): resolve the previous window, enqueue this one's cut
// and tee, continue in a fresh buffer. `final`: the tail chunk is closed too (forced cut), resolved immediately.
int stream_flush(pbsgpu_stream *s, bool final) {
    pbsgpu_engine *e = s->eng;
    CHK(set_device(e));
    static const bool trace = getenv("PBSGPU_TRACE") != nullptr;
    const double t0 = trace ? now_ms() : 0;
    CHK(stream_resolve_pending(s));
    const double t1 = trace ? now_ms() : 0;
    const uint64_t total = s->carry + s->fill;
    PendingCut &pc = s->pend;
    for (int c = 0; c < 2; ++c)  // the cut and the tee read this window's pieces: device-side wait for the copy streams
        if (s->piece_used[c]) {
            HIPCHK(hipStreamWaitEvent(s->hs, s->piece_ev[c], 0));
            s->piece_used[c] = false;
        }
    pc.sugg_rel.clear();
    CHK(stream_enqueue_tee(s, pc));
    if (total == 0) {
        if (!pc.files.empty()) {  // zero-length last pieces only (end_file right after a flush): no cut to ride on
            HIPCHK(hipStreamSynchronize(s->hs));
            for (size_t i = 0; i < pc.files.size(); ++i) {
                pc.files[i].xxh3 = s->h_tee_out.as<uint64_t>()[pc.file_out[i]];
                s->file_out.push_back(pc.files[i]);
            }
            pc.files.clear();
            pc.file_out.clear();
        }
        return PBSGPU_OK;
    }
    // suggested boundaries that can still matter: behind the open chunk's start, inside this window
    while (!s->suggested.empty() && s->suggested.front() <= s->base) s->suggested.pop_front();
    // (with a reader-buffer rule in force — pbsgpu_engine_set_suggested_feed — also the announced boundaries up to one
    // max chunk BEYOND the window: one of them may pre-empt a hash cut inside the window, see Suggested::open_end)
    const uint64_t look = e->sugg_feed.load(std::memory_order_relaxed) > 1 ? (uint64_t)e->cfg.max : 0;
    for (uint64_t b : s->suggested) {
        if (b > s->base + total + look) break;
        pc.sugg_rel.push_back(b - s->base);
    }
    const uint32_t sidx[2] = {0u, (uint32_t)pc.sugg_rel.size()};
    SuggestedHost sg{pc.sugg_rel.data(), sidx, s->base, !final};
    pbsgpu_segment seg{0, total};
    pc.ctx = s->cut_next;
    s->cut_next ^= 1;
    pc.buf = s->cur;
    pc.base = s->base;
    pc.carry = s->carry;
    pc.total = total;
    pc.section = s->section;
    pc.final = final;
    CHK(cut_enqueue(e, s->cut[pc.ctx], s->dev[s->cur].as<uint8_t>() + s->headroom - s->carry, total, &seg, 1,
                    pc.sugg_rel.empty() ? nullptr : &sg));
    pc.active = true;
    s->dev_busy[s->cur] = 1;  // until the cut has been read back (then: until its chunks are hashed, or free)
    int nb = -1;
    CHK(stream_free_buffer(s, &nb));
    s->cur = nb;
    s->carry = 0;  // known again once the pending cut is resolved
    s->fill = 0;
    if (final) CHK(stream_resolve_pending(s));
    if (trace)
        fprintf(stderr, "[pbsgpu] stream %p window %.1f MiB: previous cut read back in %.2f ms, enqueue %.2f ms; %zu windows "
                        "in flight, ring %zu\n", (void *)s, total / 1048576.0, t1 - t0, now_ms() - t1, s->inflight.size(),
                s->dev.size());
    return PBSGPU_OK;
}

// non-blocking: deliver windows whose hash job has finished
int stream_reap(pbsgpu_stream *s) {
    CHK(set_device(s->eng));
    if (s->pend.active) {  // read the previous window's cut back early if it is done (never wait here)
        const hipError_t q = hipEventQuery(s->cut[s->pend.ctx].ev[EV_SHA1]);
        if (q == hipSuccess) CHK(stream_resolve_pending(s));
        else if (q == hipErrorNotReady) (void)hipGetLastError();
        else HIPCHK(q);
    }
    while (!s->inflight.empty()) {
        const int st = stream_complete_oldest(s, false);
        if (st == PBSGPU_E_BUSY) break;
        CHK(st);
    }
    return PBSGPU_OK;
}

int stream_drain(pbsgpu_stream *s) {
    CHK(set_device(s->eng));
    CHK(stream_resolve_pending(s));
    // The NEWEST window's job first: it is usually still the open job, and nothing launches it while this thread sits in
    // the older jobs' events below — it then started only after the last of them (traced: +0.34 s on every finish). Now it
    // takes the first lane that frees up and runs beside the older jobs.
    if (!s->inflight.empty()) CHK(hd_ensure_launched(s->eng, s->inflight.back().job, true));
    while (!s->inflight.empty()) CHK(stream_complete_oldest(s, true));
    return PBSGPU_OK;
}

// H2D of one staged piece of n bytes (stage[k][0..n)) behind the window's current fill. `written` was advanced when the
// bytes were accepted; `fill` advances here.
int stream_push_piece(pbsgpu_stream *s, int k, size_t n) {
    pbsgpu_engine *e = s->eng;
    const int c = (int)(e->copy_rr.fetch_add(1, std::memory_order_relaxed) % e->copy_streams.size());
    hipStream_t cs = e->copy_streams[(size_t)c];
    HIPCHK(hipMemcpyAsync(s->dev[s->cur].as<uint8_t>() + s->headroom + s->fill, s->stage[k].p, n, hipMemcpyHostToDevice,
                          cs));
    HIPCHK(hipEventRecord(s->stage_ev[k], cs));   // the staging buffer is reusable after this
    HIPCHK(hipEventRecord(s->piece_ev[c], cs));   // ... and the window's cut waits for the last piece per copy stream
    s->piece_used[c] = true;
    s->stage_idx = (k + 1) % kStreamStages;
    s->fill += n;
    if (s->fill == s->window) CHK(stream_flush(s, false));
    return PBSGPU_OK;
}

// push whatever small writes have gathered in the current staging buffer (before a cut / finish / reserve)
int stream_push_staged(pbsgpu_stream *s) {
    if (s->stage_fill == 0) return PBSGPU_OK;
    const size_t n = s->stage_fill;
    s->stage_fill = 0;
    return stream_push_piece(s, s->stage_idx, n);
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

static size_t stream_ring_limit(size_t devcap) {  // window buffers per stream (PBSGPU_STREAM_RING_GIB, default 32 GiB)
    size_t budget_gib = 32;
    if (const char *v = getenv("PBSGPU_STREAM_RING_GIB")) budget_gib = (size_t)std::max(1L, atol(v));
    const size_t nb = (size_t)((budget_gib << 30) / devcap);
    return std::min<size_t>(std::max<size_t>(nb, 4), 256);
}

int pbsgpu_stream_create(pbsgpu_engine *e, uint64_t window_bytes, pbsgpu_stream **out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    *out = nullptr;
    if (window_bytes == 0) window_bytes = 256ull << 20;
    if (window_bytes < e->cfg.max) window_bytes = e->cfg.max;
    if (pbsgpu_stream *parked = stream_unpark(e, window_bytes)) {
        parked->max_bufs = stream_ring_limit(parked->devcap);  // (a previous life under memory pressure may have lowered it)
        int st = set_device(e);
        if (st == PBSGPU_OK) st = stream_buffer_ensure(parked, parked->dev[0]);
        if (st != PBSGPU_OK) {
            pbsgpu_stream_destroy(parked);
            return st;
        }
        *out = parked;
        return PBSGPU_OK;
    }
    pbsgpu_stream *s = new (std::nothrow) pbsgpu_stream();
    if (!s) return PBSGPU_E_NOMEM;
    engine_ref(e);
    s->eng = e;
    s->window = window_bytes;
    s->headroom = ((uint64_t)e->cfg.max + 255) & ~255ull;
    s->devcap = (size_t)window_bytes + (size_t)s->headroom + 256;
    // Ring limit. Chunks of up to 16 MiB hash for ~0.45 s, so a stream needs (ingest rate x ~0.6 s) of windows in
    // flight: 32 GiB carries ~50 GiB/s, the H2D rate (16 GiB held one fast writer at 26 GiB/s). Buffers are allocated on
    // demand — a slow producer never grows its ring — and a failed allocation is back-pressure, not an error.
    s->max_bufs = stream_ring_limit(s->devcap);
    int st = set_device(e);
    hipStream_t shared = nullptr;
    if (st == PBSGPU_OK) {  // opt-in: engine-wide cut streams (engine_internal.h, cut_streams)
        static const int nshared = []() {
            const char *v = getenv("PBSGPU_SHARED_CUT_STREAMS");
            return v ? std::min(16, std::max(0, atoi(v))) : 0;
        }();
        if (nshared > 0) {
            std::lock_guard<std::mutex> lk(e->pool_mu);
            while ((int)e->cut_streams.size() < nshared && st == PBSGPU_OK) {
                hipStream_t cs = nullptr;
                if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) st = PBSGPU_E_HIP;
                else e->cut_streams.push_back(cs);
            }
            if (st == PBSGPU_OK) shared = e->cut_streams[e->cut_rr++ % e->cut_streams.size()];
        }
    }
    if (st == PBSGPU_OK) st = s->cut[0].init(shared);
    if (st == PBSGPU_OK) st = s->cut[1].init(s->cut[0].stream);
    s->hs = s->cut[0].stream;
    for (auto &ev : s->piece_ev)
        if (st == PBSGPU_OK && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) st = PBSGPU_E_HIP;
    // everything a window can need is allocated here: a regrow later (hipFree / hipHostFree) would wait for the whole
    // device, i.e. for other windows' running hash jobs
    for (auto &c : s->cut) {
        c.mapped_ctrl = true;
        if (st == PBSGPU_OK) st = presize_cut(e, c, s->devcap);
    }
    if (st == PBSGPU_OK) {
        s->dev.resize(2);
        s->dev_busy.assign(2, 0);
        st = stream_buffer_ensure(s, s->dev[0]);
    }
    for (int i = 0; i < kStreamStages && st == PBSGPU_OK; ++i)
        if (hipEventCreateWithFlags(&s->stage_ev[i], hipEventDisableTiming) != hipSuccess) st = PBSGPU_E_HIP;
    if (st != PBSGPU_OK) {
        pbsgpu_stream_destroy(s);
        return st;
    }
    *out = s;
    return PBSGPU_OK;
}

// ---- stream contexts are recycled --------------------------------------------------------------------------------
// Everything a stream owns besides its window buffers (two cut contexts with their device tables and mapped pinned
// result buffers, 3 x 32 MiB pinned staging, the tee's buffers, events, one HIP stream) is parked in the engine when the
// stream closes and handed to the engine's next stream of the same window size. Freeing it (hipFree / hipHostFree) waits
// for the whole device — for a writer that closes archive k while archive k+1 is already being hashed that is a stall
// of up to one chunk chain (0.45 s) on the thread that should be writing.
static void stream_free_context(pbsgpu_stream *s) {
    for (auto &ev : s->piece_ev)
        if (ev) (void)hipEventDestroy(ev);
    s->cut[1].destroy();
    s->cut[0].destroy();
    for (auto &b : s->dev) b.release();
    for (auto &b : s->stage) b.release();
    s->tee_states.release();
    s->tee_queue.release();
    s->tee_items.release();
    s->tee_sums.release();
    s->h_tee_items.release();
    s->h_tee_out.release();
    for (auto &ev : s->stage_ev)
        if (ev) (void)hipEventDestroy(ev);
}

// back to the state pbsgpu_stream_create leaves a stream in (the resources stay)
static void stream_reset_state(pbsgpu_stream *s) {
    s->dev.clear();
    s->dev.resize(2);
    s->dev_busy.assign(2, 0);
    s->cur = 0;
    s->carry = 0;
    s->fill = 0;
    s->cut_next = 0;
    s->piece_used[0] = s->piece_used[1] = false;
    s->pend = PendingCut{};
    s->stage_idx = 0;
    s->stage_fill = 0;
    s->reserved = -1;
    s->base = 0;
    s->written = 0;
    s->inject_total = 0;
    s->section = 0;
    s->finished = false;
    s->drained = false;
    s->suggested.clear();
    s->inflight.clear();
    s->out.clear();
    s->descs.clear();
    s->files.clear();
    s->next_file = 0;
    s->n_stateful = 0;
    s->file_open = false;
    s->entry_left = 0;
    s->in_entry = false;
    s->file_out.clear();
}

static bool stream_park(pbsgpu_engine *e, pbsgpu_stream *s) {
    static const size_t limit = []() -> size_t {
        const char *v = getenv("PBSGPU_STREAM_CTX_POOL");
        return (size_t)(v ? std::max(0L, atol(v)) : 8L);
    }();
    stream_reset_state(s);
    s->eng = nullptr;  // (a parked context holds no reference: the engine owns it)
    std::lock_guard<std::mutex> lk(e->pool_mu);
    if (e->destroyed || e->stream_pool.size() >= limit) return false;
    e->stream_pool.push_back(s);
    return true;
}

static pbsgpu_stream *stream_unpark(pbsgpu_engine *e, uint64_t window_bytes) {
    pbsgpu_stream *s = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->pool_mu);
        for (size_t i = 0; i < e->stream_pool.size(); ++i)
            if (e->stream_pool[i]->window == window_bytes) {
                s = e->stream_pool[i];
                e->stream_pool.erase(e->stream_pool.begin() + (long)i);
                break;
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
}  // namespace pbse
extern "C" {

void pbsgpu_stream_destroy(pbsgpu_stream *s) {
    if (!s) return;
    pbsgpu_engine *e = s->eng;
    if (e) {
        (void)hipSetDevice(e->device);
        (void)stream_drain(s);
        while (!s->inflight.empty()) {  // drain failed (HIP error): drop the references so the jobs can be reused
            s->inflight.front().job->refs.fetch_sub(1);
            s->inflight.pop_front();
        }
        for (int k = 0; k < kStreamStages; ++k)
            if (s->stage_ev[k]) (void)hipEventSynchronize(s->stage_ev[k]);  // pieces still copying from our staging
        if (s->hs) (void)hipStreamSynchronize(s->hs);
        {   // window buffers go back to the engine for the next stream (hipFree waits for the whole device, i.e. for other
            // streams' running hash jobs). The pool is bounded in BYTES (one default ring, PBSGPU_STREAM_POOL_GIB): a
            // long-lived engine that opens one stream per backup job must not pin tens of GiB of HBM for ever; what does
            // not fit is freed below. pbsgpu_engine_trim() empties the pool on request.
            static const uint64_t pool_limit = []() -> uint64_t {
                const char *v = getenv("PBSGPU_STREAM_POOL_GIB");
                return (uint64_t)(v ? std::max(0L, atol(v)) : 32L) << 30;
            }();
            std::lock_guard<std::mutex> lk(e->pool_mu);
            uint64_t held = 0;
            for (auto &b : e->win_pool) held += b.cap;
            for (auto &b : s->dev)
                if (b.p && held + b.cap <= pool_limit) {
                    held += b.cap;
                    e->win_pool.push_back(std::move(b));
                }
        }
        for (auto &b : s->dev) b.release();
        if (stream_park(e, s)) {  // the context waits for the engine's next stream: nothing else is freed, nothing waits
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
    if (s->in_entry && len > s->entry_left) return PBSGPU_E_INVALID;  // more bytes than the entry header announced
    if (len) CHK(set_device(s->eng));
    const uint8_t *p = static_cast<const uint8_t *>(data);
    while (len) {
        // caller bytes -> library-owned pinned staging -> device window (async). Writes are COALESCED in the staging
        // buffer (an io.Copy feeds 32 KiB at a time, a payload header is 16 bytes): one H2D piece per 32 MiB or per
        // window edge, not per call. The memcpy runs on the caller's thread with no lock held, so several streams copy
        // in parallel.
        const int k = s->stage_idx;
        if (s->stage_fill == 0) {
            CHK(s->stage[k].ensure(kStreamStage));
            HIPCHK(hipEventSynchronize(s->stage_ev[k]));  // its previous H2D copy has drained
        }
        size_t n = std::min(len, kStreamStage - s->stage_fill);
        n = (size_t)std::min<uint64_t>(n, s->window - s->fill - s->stage_fill);
        parallel_memcpy(s->stage[k].as<uint8_t>() + s->stage_fill, p, n);
        s->stage_fill += n;
        s->written += n;
        if (s->in_entry) s->entry_left -= std::min<uint64_t>(s->entry_left, n);
        p += n;
        len -= n;
        if (s->stage_fill == kStreamStage || s->fill + s->stage_fill == s->window) CHK(stream_push_staged(s));
    }
    return PBSGPU_OK;
}

int pbsgpu_stream_reserve(pbsgpu_stream *s, void **buf, size_t *cap) {
    if (!s || !buf || !cap) return PBSGPU_E_INVALID;
    if (s->finished || s->reserved >= 0) return PBSGPU_E_STATE;
    CHK(set_device(s->eng));
    CHK(stream_push_staged(s));  // bytes gathered by earlier small writes go first; the reserved buffer starts empty
    const int k = s->stage_idx;
    CHK(s->stage[k].ensure(kStreamStage));
    HIPCHK(hipEventSynchronize(s->stage_ev[k]));  // its previous H2D copy has drained
    s->reserved = k;
    *buf = s->stage[k].p;
    uint64_t room = std::min<uint64_t>(kStreamStage, s->window - s->fill);
    if (s->in_entry) room = std::min(room, s->entry_left);
    *cap = (size_t)room;
    return PBSGPU_OK;
}

int pbsgpu_stream_commit(pbsgpu_stream *s, size_t len) {
    if (!s) return PBSGPU_E_INVALID;
    if (s->reserved < 0) return PBSGPU_E_STATE;
    const int k = s->reserved;
    if (len > (size_t)std::min<uint64_t>(kStreamStage, s->window - s->fill)) return PBSGPU_E_INVALID;
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
    s->suggested.push_back(offset);  // boundaries at or before the open chunk's start are dropped at the next flush
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
    CHK(set_device(s->eng));
    CHK(stream_push_staged(s));
    CHK(stream_flush(s, true));
    s->base += inject_bytes;
    s->inject_total += inject_bytes;
    s->section++;
    return PBSGPU_OK;
}

// close the input: the tail chunk is cut, every chunk is with the hash jobs; nothing here waits for a hash
static int stream_close_input(pbsgpu_stream *s) {
    if (s->finished) return PBSGPU_OK;
    if (s->reserved >= 0 || s->file_open) return PBSGPU_E_STATE;
    CHK(set_device(s->eng));
    CHK(stream_push_staged(s));
    CHK(stream_flush(s, true));
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
        CHK(stream_reap(s));
        s->drained = !s->pend.active && s->inflight.empty();
    }
    *done = s->drained ? 1 : 0;
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
