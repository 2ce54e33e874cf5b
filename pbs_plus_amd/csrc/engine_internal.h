// Internal declarations shared by the libpbsgpu translation units (not part of the C ABI).
//
// Concurrency model (cgo calls arrive on arbitrary OS threads, several goroutines may share one engine):
//   * pbsgpu_engine::mu guards only bookkeeping — which pool slot is free, the ticket table, the aux leases, the
//     child count. It is NEVER held across a HIP synchronisation, a memcpy or a kernel enqueue.
//   * a Slot (device work context: HIP stream, events, work buffers, pinned staging) is owned by exactly one
//     logical user at a time: a batch ticket (pool slots), one synchronous helper call (aux slots, leased; callers
//     WAIT for a lease instead of failing), or one stream / chunker handle (private slot, never shared).
//   * Slot::op serialises concurrent operations on the same ticket (wait / collect / timing from two threads).
//   * streaming writers (pbsgpu_stream_*) are clients of ONE engine-owned page ring (ring_internal.h; own mutex): pages,
//     cut rounds, the persistent SHA-256 service and record delivery are shared by all streams of the engine.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "kernels.h"

namespace pbse {

extern std::atomic<int> g_last_hip_error;

#define HIPCHK(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) {                        \
            pbse::g_last_hip_error.store((int)_e);     \
            return PBSGPU_E_HIP;                       \
        }                                              \
    } while (0)

#define CHK(expr)                       \
    do {                                \
        int _s = (expr);                \
        if (_s != PBSGPU_OK) return _s; \
    } while (0)

// hipFree / hipHostFree wait for the WHOLE device — including a page ring's persistent SHA-256 service, which only ends
// on request: a free issued while a service runs would block until someone stops it (for ever, if the caller is the
// thread that would). Every release in this library therefore goes through dev_free / host_free: while a service runs ON
// THE DEVICE THE MEMORY BELONGS TO the pointer is parked in that device's graveyard and really freed when the device's last
// service has stopped (ring.cpp: quiesce / park / the service's own stop); a device without a running service frees at once
// (round 5: services and graveyard are tracked per device — a service on GPU 0 no longer parks the frees of GPU 1).
// Parked DEVICE bytes are bounded: beyond the cap (an eighth of the device's memory, PBSGPU_GRAVEYARD_MIB) the device's
// rings are asked to park their services (service_park_generation, honoured once per request by the next pump), the last one to end frees
// everything and the services start again with the next round.
void service_started(int device);     // a page-ring service is about to be launched (waits for a flush in progress)
void service_ended(int device);       // ... is known to have ended; frees the device's parked memory when it was the last
int services_running(int device);
uint32_t service_park_generation(int device);  // 0 = no request pending, else the number of the pending request
void dev_free(void *p);
void host_free(void *p);

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
        return *this;
    }
    int ensure(size_t bytes) {
        if (bytes <= cap) return PBSGPU_OK;
        if (p) {
            dev_free(p);  // device-wide wait: see PinnedBuf::ensure
            p = nullptr;
            bytes = std::max(bytes, std::min<size_t>(cap * 2, cap + (256u << 20)));
            cap = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            g_last_hip_error.store((int)e);
            (void)hipGetLastError();
            p = nullptr;
            return PBSGPU_E_NOMEM;
        }
        cap = want;
        return PBSGPU_OK;
    }
    void release() {
        if (p) dev_free(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    PinnedBuf(PinnedBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    // NOTE: hipHostFree / hipFree wait for the whole device to go idle — a regrow while another stream's 0.4 s SHA
    // launch is running stalls the caller that long (measured on the streaming writer). Hot paths therefore size their
    // buffers once for the largest case (ensure_once / presize) and growth doubles.
    int ensure(size_t bytes) {
        if (bytes <= cap) return PBSGPU_OK;
        if (p) {
            host_free(p);
            bytes = std::max(bytes, cap * 2);
        }
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            g_last_hip_error.store((int)e);
            (void)hipGetLastError();
            p = nullptr;
            return PBSGPU_E_NOMEM;
        }
        cap = bytes;
        return PBSGPU_OK;
    }
    void release() {
        if (p) host_free(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// device scalars of one slot (uint32 each)
enum : int { SC_NCAND = 0, SC_NREC = 1, SC_MAXCNT = 2, SC_QUEUE = 3, SC_WGLIMIT = 4, SC_ZERO = 5 /* stays 0 */,
             SC_TILEQ = 6 /* u64 */, SC_PARFB = 8 /* parallel resolve handed the job back */, SC_PARHOPS = 9, SC_COUNT = 10 };

enum : int { EV_BEGIN = 0, EV_SCAN0, EV_SCAN1, EV_RESOLVE1, EV_SHA1, EV_COUNT };

constexpr size_t kStageBytes = 32u << 20;

struct Slot {
    hipStream_t stream = nullptr;
    bool own_stream = true;  // false: the stream belongs to a sibling context (a payload stream's second cut context)
    hipEvent_t ev[EV_COUNT] = {};
    DevBuf data;  // staged copy of host submits
    DevBuf tile_cnt, tile_off, tile_slots, dense, scan_tmp, scalars, segs, seg_cnt, seg_off, recs, order;
    DevBuf sugg, sugg_idx;  // suggested boundaries (optional)
    DevBuf par;             // scratch of the parallel single-stream resolve (doubling tables)
    PinnedBuf h_scalars;    // readback of SC_*
    PinnedBuf h_segs;       // pinned copy of the segment table
    PinnedBuf h_sugg;       // pinned copy of suggested offsets + index
    PinnedBuf h_recs;       // mapped pinned copy of the finished records (published by kernel, not by the copy engine)
    bool recs_published = false;
    PinnedBuf stage[2];     // host -> device staging of this slot (lazily allocated)
    hipEvent_t stage_ev[2] = {};
    // Control tables (segment table, suggested boundaries) are either copied to the device (batch tickets: thousands
    // of segments read by many waves) or, for the streaming writers, READ BY THE KERNELS STRAIGHT FROM the mapped pinned
    // copies: a small H2D copy would queue behind the megabytes of payload pieces in the shared SDMA queues.
    bool mapped_ctrl = false;
    const pbsgpu_segment *segs_dev() const {
        return mapped_ctrl ? h_segs.as<pbsgpu_segment>() : segs.as<pbsgpu_segment>();
    }
    std::mutex op;          // serialises operations on the ticket that owns this slot
    // in-flight state (owner only)
    bool busy = false;      // under engine mu
    bool ready = false;     // ticket published (under engine mu)
    bool synced = false;
    uint64_t ticket = 0;
    const uint8_t *dptr = nullptr;
    uint64_t nbytes = 0;
    uint32_t nseg = 0;
    uint32_t cap = 0;
    uint64_t rec_cap = 0;
    uint64_t nsugg = 0;     // suggested offsets staged for this batch (0 = none)
    uint64_t sugg_origin = 0;
    bool sugg_open_end = false;
    bool host_submit = false;
    uint32_t retries = 0;
    bool dense_mode = false;  // scanned at the capacity limit: the resolve walks re-scan overflowed tiles on demand (DenseTiles)
    uint64_t nrec = 0, ncand = 0;

    int init(hipStream_t borrowed = nullptr);  // stream (unless borrowed) + events
    void destroy();  // frees everything (device must be current)
};

}  // namespace pbse

struct pbsgpu_stream;
struct pbsgpu_ring;
struct pbsgpu_engine {
    int device = 0;
    int num_cus = 256;
    // SHA workgroup budget slack (k_order): 25 % keeps one batch's makespan at its longest chunk; with more
    // than 4 batches in flight the chip is oversubscribed anyway and 0 % (fewest CUs per batch) carries more
    uint32_t sha_slack_pct = 25;
    pbsgpu_engine_options opt{};  // as given to pbsgpu_engine_create_opt, defaults resolved (engine.cpp: resolve_engine_options)
    pbsgpu_config cfg{};
    uint32_t bits = 0;     // mask == 2^bits - 1
    uint32_t thr = 0;      // break_min << (32 - bits)
    uint32_t effmin = 0;   // max(min, 65)
    // reader buffer size the suggested-boundary rule emulates (pbsgpu_engine_set_suggested_feed): 1 = byte-serial
    std::atomic<uint64_t> sugg_feed{1};
    std::atomic<uint32_t> sugg_feed_abs{0};
    uint32_t *d_table_rot = nullptr;
    std::vector<std::unique_ptr<pbse::Slot>> slots;  // batch-ticket pool
    std::vector<std::unique_ptr<pbse::Slot>> aux;    // leased to synchronous helper calls
    std::vector<char> aux_busy;
    uint64_t next_ticket = 1;
    std::atomic<int> tickets_out{0};  // pool slots in use (read without the lock as a scheduling hint)
    std::atomic<uint32_t> cap_hint{0};       // per-tile slot capacity that a density retry settled on
    std::atomic<uint32_t> cap_hint_tile{0};  // ... for this tile size
    std::mutex mu;
    std::condition_variable cv;  // an aux lease was returned
    // Host -> device payload copies of ALL streams go through these few engine-wide HIP streams (a page's copies stay on
    // one of them). Measured: one HIP copy stream per payload stream maps 8 streams unevenly onto the SDMA engines
    // (17 GiB/s aggregate, some streams 4x slower than others); two always-busy copy queues carry ~50 GiB/s whatever
    // the number of producers. Only ready copies are ever enqueued here (nothing that waits for a kernel).
    std::vector<hipStream_t> copy_streams;
    std::atomic<uint32_t> copy_rr{0};
    // ... and the per-file XXH3 tees of all streams through these (short kernels behind a page's copy; a stream keeps
    // to one of them so that its pieces run in order). Few engine-wide HIP streams instead of one per payload stream keep a
    // process under the ~20 hardware queues beyond which every kernel pays 19 %.
    std::vector<hipStream_t> tee_streams;
    std::atomic<uint32_t> tee_rr{0};
    // The engine's page ring: every payload stream of the engine is a client of it (stream.cpp). Created by the first
    // pbsgpu_stream_create, destroyed with the engine (or by pbsgpu_engine_trim when no stream is alive).
    std::mutex sring_mu;             // creation / destruction only; the ring has its own lock for use
    pbsgpu_ring *sring = nullptr;
    int sring_users = 0;             // live payload streams (under sring_mu)
    // whole stream contexts (pinned staging, tee buffers, events) of closed streams, re-used by the engine's next stream:
    // allocating 100 MB of pinned memory per archive costs tens of milliseconds, freeing it would wait for the device
    std::mutex pool_mu;
    std::vector<pbsgpu_stream *> stream_pool;
    int refs = 1;                // owner + live streams / chunkers (under mu); freed when it drops to 0
    bool destroyed = false;      // pbsgpu_engine_destroy was called (children may still be alive)
};

namespace pbse {

uint32_t default_cap(const pbsgpu_engine *e, uint64_t nbytes);
// largest per-tile slot capacity a scan is given: one candidate per 128 bytes (or twice the chunker's nominal provisioning,
// whichever is larger). Tiles beyond it are resolved by on-demand re-scans (DenseTiles, kernels.h), never by more memory.
uint32_t cap_limit(const pbsgpu_engine *e, uint32_t tile_bytes);
int set_device(const pbsgpu_engine *e);
bool is_device_pointer(const void *p);

// leases one of the engine's aux slots for the duration of a synchronous helper call (waits for a free one)
struct AuxLease {
    pbsgpu_engine *e;
    Slot *s = nullptr;
    int idx = -1;
    explicit AuxLease(pbsgpu_engine *eng);
    ~AuxLease();
    AuxLease(const AuxLease &) = delete;
    AuxLease &operator=(const AuxLease &) = delete;
};

void engine_ref(pbsgpu_engine *e);
void engine_unref(pbsgpu_engine *e);  // frees the engine when the last reference goes

struct SuggestedHost {  // caller's suggested boundaries: offsets[index[s] .. index[s+1]) ascending, relative to segment s
    const uint64_t *offsets = nullptr;
    const uint32_t *index = nullptr;
    uint64_t origin = 0;  // stream offset of the (single) segment's first byte: only the absolute feed grid needs it
    bool open_end = false;  // the segment ends where the bytes seen so far end, not where the stream ends
};

int enqueue_candidates(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, uint32_t cap,
                       uint64_t nseg_hint = 0);
int staged_h2d(Slot &s, void *dst, const void *src, uint64_t nbytes, hipStream_t st);
int stage_segments(pbsgpu_engine *e, Slot &s, const pbsgpu_segment *segs, uint32_t nseg, uint64_t nbytes,
                   const SuggestedHost *sg);
// synchronous helper for the upstream-style chunker (the caller owns the slot)
int candidates_sync(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, uint64_t *count);
// Caller bytes -> pinned staging. One thread copies ~12 GB/s, the H2D engine moves 57 GB/s: a single writer (the
// reference drives ONE goroutine per archive, internal/tapeio/converter.go:672-680) was memcpy-bound at 24-28 GiB/s.
// Large copies are therefore split over a few helper threads of a process-wide pool (started at the first large
// write; PBSGPU_COPY_THREADS, default 4 incl. the caller, 1 = off). Small writes stay on the caller's thread.
void parallel_memcpy(void *dst, const void *src, size_t n);

// stream contexts parked in pbsgpu_engine::stream_pool (stream.cpp): really free them (engine teardown, trim)
void stream_pool_release(pbsgpu_engine *e);

// the engine's page ring (stream.cpp): destroyed at engine teardown / trim
void engine_ring_release(pbsgpu_engine *e);

}  // namespace pbse
