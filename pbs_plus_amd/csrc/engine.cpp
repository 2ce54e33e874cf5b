// libpbsgpu host side: C-ABI entry points (include/pbsgpu.h) over the gfx950 kernels.
//
// Stands where github.com/pbs-plus/pxar's backupproxy session runs its chunk loop for
// the writers the reference drives (internal/pxarmount/commit_orchestrate.go:137-177,
// internal/tapeio/converter.go:386-439). No CPU fallback exists in this file: every data
// path launches the HIP kernels; without a device the engine cannot be created.
#include <chrono>
#include "engine_internal.h"

#include <cstdio>
#include <cstdlib>

// The engine runs up to 16 batches on separate HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues
// (default 4) and reads the variable when the HIP runtime initialises, i.e. at the first HIP call of the process.
// Setting a default when this library is loaded covers hosts that bind the C ABI directly (cgo, JNI, ctypes)
// without going through the Python package; an explicit setting of the host always wins.
__attribute__((constructor)) static void pbsgpu_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "20", 0); }

namespace pbse {
std::atomic<int> g_last_hip_error{0};

namespace {
constexpr int kMaxDevices = 64;
struct Graveyard {
    std::mutex svc_mu;               // service start <-> flush: no service may start on the device while parked memory is being
                                     // freed (a hipFree issued then would wait for a kernel that only ends on request)
    std::atomic<int> services{0};    // page-ring services launched on the device and not yet known to have ended
    std::atomic<int> park_request{0};
    std::atomic<uint32_t> park_gen{0};  // requests raised so far (a ring honours each request once)
    double park_t = 0, park_backoff_ms = 0;  // (under mu) when the pending request was raised last, and how long until it is raised again
    std::mutex mu;                   // the lists below
    std::vector<void *> dev, host;
    uint64_t dev_bytes = 0;
    std::atomic<uint64_t> cap_bytes{0};  // default cap of this device (an eighth of its memory), found at the first parked free
};
Graveyard g_grave[kMaxDevices];

// the device a pointer belongs to (device memory, or pinned host memory allocated under a device); the calling thread's
// current device when HIP cannot tell
int device_of(const void *p) {
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) == hipSuccess && a.device >= 0 && a.device < kMaxDevices) return a.device;
    (void)hipGetLastError();
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) (void)hipGetLastError();
    return (d >= 0 && d < kMaxDevices) ? d : 0;
}

// really free what was parked on `device` — only with no service running there, and with none able to start meanwhile
void graveyard_flush(int device) {
    Graveyard &g = g_grave[device];
    std::lock_guard<std::mutex> hold(g.svc_mu);
    if (g.services.load(std::memory_order_acquire) > 0) return;  // (one started since the caller looked: its end flushes)
    std::vector<void *> d, h;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        d.swap(g.dev);
        h.swap(g.host);
        g.dev_bytes = 0;
    }
    g.park_request.store(0, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(g.mu);
        g.park_backoff_ms = 0;
    }
    if (d.empty() && h.empty()) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (have_cur && cur != device) (void)hipSetDevice(device);
    for (void *p : d) (void)hipFree(p);
    for (void *p : h) (void)hipHostFree(p);
    if (have_cur && cur != device) (void)hipSetDevice(cur);
}
}  // namespace

void service_started(int device) {
    if (device < 0 || device >= kMaxDevices) return;
    Graveyard &g = g_grave[device];
    std::lock_guard<std::mutex> hold(g.svc_mu);  // (a flush in progress finishes first: milliseconds)
    g.services.fetch_add(1, std::memory_order_acq_rel);
}

void service_ended(int device) {
    if (device < 0 || device >= kMaxDevices) return;
    if (g_grave[device].services.fetch_sub(1, std::memory_order_acq_rel) == 1) graveyard_flush(device);
}

int services_running(int device) {
    return (device >= 0 && device < kMaxDevices) ? g_grave[device].services.load(std::memory_order_acquire) : 0;
}

uint32_t service_park_generation(int device) {
    if (device < 0 || device >= kMaxDevices) return 0;
    Graveyard &g = g_grave[device];
    return g.park_request.load(std::memory_order_acquire) ? g.park_gen.load(std::memory_order_acquire) : 0u;
}

void dev_free(void *p) {
    if (!p) return;
    const int device = device_of(p);
    Graveyard &g = g_grave[device];
    if (g.services.load(std::memory_order_acquire) > 0) {
        // how much is parked decides when the device's rings are asked to let go of their services
        size_t size = 0;
        hipDeviceptr_t base = nullptr;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
            (void)hipGetLastError();
            size = 64u << 20;  // unknown: count it as a sizeable block rather than as nothing (the cap errs on the early side)
        }
        // the cap: PBSGPU_GRAVEYARD_MIB (debug override, read once) or an eighth of the memory of the device the POINTER
        // belongs to — not of whatever device the calling thread has current
        static const uint64_t env_cap = []() -> uint64_t {
            const char *v = getenv("PBSGPU_GRAVEYARD_MIB");
            return v ? (uint64_t)(std::max(1.0, atof(v)) * 1048576.0) : 0ull;
        }();
        uint64_t dflt = 0;
        if (env_cap == 0 && g.cap_bytes.load() == 0) {  // (outside g.mu: two HIP calls)
            int cur = 0;
            const bool have_cur = hipGetDevice(&cur) == hipSuccess;
            if (have_cur && cur != device) (void)hipSetDevice(device);
            size_t fr = 0, tot = 0;
            dflt = hipMemGetInfo(&fr, &tot) == hipSuccess ? std::max<uint64_t>(tot / 8, 1ull << 30) : (8ull << 30);
            if (have_cur && cur != device) (void)hipSetDevice(cur);
        }
        std::lock_guard<std::mutex> lk(g.mu);
        if (g.services.load(std::memory_order_acquire) > 0) {  // (still: a flush takes g.mu after the count reached zero)
            if (g.cap_bytes.load() == 0 && dflt) g.cap_bytes.store(dflt);
            const uint64_t cap = env_cap ? env_cap : (g.cap_bytes.load() ? g.cap_bytes.load() : (8ull << 30));
            g.dev.push_back(p);
            g.dev_bytes += size;
            if (g.dev_bytes > cap) {
                // A request every ring honours ONCE can fail: ring A lets go, waits its grace period for ring B, B's service is
                // still finishing its chunks when the grace ends, A starts again — and when B ends the count is back at one. The
                // request would then stay pending, unanswerable, until a service ends for another reason, while frees keep
                // being parked (one run in five of tests/test_gpu_round5.py's two-ring case ran out of memory that way). It is
                // raised AGAIN, as a new generation, while frees arrive over the cap: after 0.5 s, then 1, 2, 4, 8 s — bounded,
                // because a ring that nobody calls any more (a leaked engine) can never answer, and each new generation costs
                // every other ring one more grace period.
                const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
                const bool pending = g.park_request.load(std::memory_order_relaxed) != 0;
                if (!pending || t - g.park_t > g.park_backoff_ms) {
                    g.park_backoff_ms = pending ? std::min(g.park_backoff_ms * 2.0, 8000.0) : 500.0;
                    g.park_t = t;
                    g.park_gen.fetch_add(1, std::memory_order_acq_rel);
                    g.park_request.store(1, std::memory_order_release);
                }
            }
            return;
        }
    }
    (void)hipFree(p);
}

void host_free(void *p) {
    if (!p) return;
    const int device = device_of(p);
    Graveyard &g = g_grave[device];
    if (g.services.load(std::memory_order_acquire) > 0) {
        std::lock_guard<std::mutex> lk(g.mu);
        if (g.services.load(std::memory_order_acquire) > 0) {
            g.host.push_back(p);
            return;
        }
    }
    (void)hipHostFree(p);
}
}  // namespace pbse
using namespace pbse;


namespace pbse {

int Slot::init(hipStream_t borrowed) {
    if (borrowed) {
        stream = borrowed;
        own_stream = false;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    for (auto &e : stage_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return PBSGPU_OK;
}

void Slot::destroy() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (DevBuf *b : {&data, &tile_cnt, &tile_off, &tile_slots, &dense, &scan_tmp, &scalars, &segs, &seg_cnt, &seg_off,
                      &recs, &order, &sugg, &sugg_idx, &par})
        b->release();
    h_scalars.release();
    h_segs.release();
    h_sugg.release();
    h_recs.release();
    for (auto &b : stage) b.release();
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : stage_ev)
        if (e) (void)hipEventDestroy(e);
    if (stream && own_stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
}

uint32_t default_cap(const pbsgpu_engine *e, uint64_t nbytes) {
    const uint32_t tile_bytes = pbsk::scan_tile_bytes(nbytes);
    // a corpus that needed a larger per-tile capacity once tends to need it again: start there
    const uint32_t hint = e->cap_hint.load(std::memory_order_relaxed);
    // expected candidates per wave tile = 3 * tile / (mask + 1); leave generous headroom
    const double lambda = 3.0 * tile_bytes / ((double)e->cfg.mask + 1.0);
    double want = 4.0 * lambda + 16.0;
    uint32_t cap = 8;
    while (cap < want) cap <<= 1;
    if (hint > cap && e->cap_hint_tile.load(std::memory_order_relaxed) == tile_bytes) return std::min(hint, cap_limit(e, tile_bytes));
    return cap;
}

uint32_t cap_limit(const pbsgpu_engine *e, uint32_t tile_bytes) {
    const double lambda = 3.0 * tile_bytes / ((double)e->cfg.mask + 1.0);
    uint32_t capv = 8;
    while (capv < 4.0 * lambda + 16.0) capv <<= 1;
    return std::min<uint32_t>(std::max<uint32_t>(capv * 2, tile_bytes / 128), tile_bytes);
}

int set_device(const pbsgpu_engine *e) {
    HIPCHK(hipSetDevice(e->device));
    return PBSGPU_OK;
}

AuxLease::AuxLease(pbsgpu_engine *eng) : e(eng) {
    std::unique_lock<std::mutex> lk(e->mu);
    for (;;) {
        for (size_t i = 0; i < e->aux.size(); ++i)
            if (!e->aux_busy[i]) {
                e->aux_busy[i] = 1;
                idx = (int)i;
                s = e->aux[i].get();
                return;
            }
        e->cv.wait(lk);  // helper calls are synchronous and self-contained: a lease always comes back
    }
}

AuxLease::~AuxLease() {
    {
        std::lock_guard<std::mutex> lk(e->mu);
        e->aux_busy[idx] = 0;
    }
    e->cv.notify_one();
}

static void free_engine(pbsgpu_engine *e) {
    (void)hipSetDevice(e->device);
    engine_ring_release(e);  // (stops its SHA-256 service: everything below may free)
    for (auto cs : e->copy_streams) {
        (void)hipStreamSynchronize(cs);
        (void)hipStreamDestroy(cs);
    }
    for (auto cs : e->tee_streams) {
        (void)hipStreamSynchronize(cs);
        (void)hipStreamDestroy(cs);
    }
    stream_pool_release(e);
    for (auto &s : e->slots)
        if (s) s->destroy();  // a failed create leaves a null entry behind (new(nothrow) Slot)
    for (auto &s : e->aux)
        if (s) s->destroy();
    if (e->d_table_rot) dev_free(e->d_table_rot);
    delete e;
}

void engine_ref(pbsgpu_engine *e) {
    std::lock_guard<std::mutex> lk(e->mu);
    e->refs++;
}

void engine_unref(pbsgpu_engine *e) {
    bool last;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        last = (--e->refs == 0);
    }
    if (last) free_engine(e);
}

static uint64_t record_upper_bound(const pbsgpu_engine *e, const pbsgpu_segment *segs, uint32_t nseg, uint64_t nsugg) {
    uint64_t n = nsugg;  // every accepted suggested boundary adds at most one cut
    for (uint32_t i = 0; i < nseg; ++i) n += segs[i].length / std::min(e->effmin, e->cfg.min) + 1;
    return n;
}

static int validate_segments(const pbsgpu_segment *segs, uint32_t nseg, uint64_t nbytes) {
    uint64_t prev_end = 0;
    for (uint32_t i = 0; i < nseg; ++i) {
        if (segs[i].offset < prev_end) return PBSGPU_E_INVALID;
        if (segs[i].length > nbytes || segs[i].offset > nbytes - segs[i].length) return PBSGPU_E_INVALID;
        prev_end = segs[i].offset + segs[i].length;
    }
    return PBSGPU_OK;
}

// enqueue scan -> compaction (candidates land dense + ascending in slot.dense, count in SC_NCAND)
int enqueue_candidates(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, uint32_t cap,
                       uint64_t nseg_hint) {
    const uint32_t lead = (uint32_t)((uintptr_t)dptr & 127u);  // scan from the 128-byte line containing dptr[0]
    const uint64_t extent = nbytes + lead;
    const uint32_t tile_bytes = pbsk::scan_tile_bytes(nbytes);
    const uint64_t ntiles = (extent + tile_bytes - 1) / tile_bytes;
    if (ntiles * (uint64_t)cap >= (1ull << 32)) return PBSGPU_E_CAPACITY;  // (> 0.5 TB in one batch at the capacity limit)
    CHK(s.tile_cnt.ensure((size_t)ntiles * 4 + 16));
    CHK(s.tile_off.ensure((size_t)ntiles * 4 + 16));
    CHK(s.tile_slots.ensure((size_t)ntiles * cap * 4 + 16));
    CHK(s.dense.ensure((size_t)ntiles * cap * 8 + 16));
    // every buffer the queued kernels touch is sized BEFORE the first launch: growing one later would
    // free memory an already-queued kernel still uses
    CHK(s.scan_tmp.ensure(pbsk::scan_tmp_words(std::max<uint64_t>(std::max<uint64_t>(ntiles, nseg_hint), 1)) * 4));
    CHK(s.scalars.ensure(SC_COUNT * 4));
    HIPCHK(hipMemsetAsync(s.scalars.p, 0, SC_COUNT * 4, s.stream));

    pbsk::ScanParams p{};
    p.data_al = dptr - lead;
    p.lead = lead;
    p.nbytes = nbytes;
    p.ntiles = ntiles;
    p.tile_bytes = tile_bytes;
    p.table_rot = e->d_table_rot;
    p.thr = e->thr;
    p.cap = cap;
    p.tile_cnt = s.tile_cnt.as<uint32_t>();
    p.tile_slots = s.tile_slots.as<uint32_t>();
    p.tile_queue = reinterpret_cast<unsigned long long *>(s.scalars.as<uint32_t>() + SC_TILEQ);
    p.shared_chip = e->tickets_out.load(std::memory_order_relaxed) > 1 ? 1u : 0u;  // a hint: other batches are hashing
    HIPCHK(hipEventRecord(s.ev[EV_SCAN0], s.stream));
    HIPCHK(pbsk::launch_scan(p, e->num_cus, s.stream));
    HIPCHK(hipEventRecord(s.ev[EV_SCAN1], s.stream));
    uint32_t *sc = s.scalars.as<uint32_t>();
    HIPCHK(pbsk::launch_exclusive_scan(s.tile_cnt.as<uint32_t>(), ntiles, cap, s.tile_off.as<uint32_t>(),
                                       sc + SC_NCAND, sc + SC_MAXCNT, s.scan_tmp.as<uint32_t>(), s.stream));
    HIPCHK(pbsk::launch_compact(s.tile_cnt.as<uint32_t>(), s.tile_off.as<uint32_t>(), s.tile_slots.as<uint32_t>(),
                                cap, ntiles, lead, nbytes, s.dense.as<uint64_t>(), ntiles * (uint64_t)cap, tile_bytes,
                                s.stream));
    return PBSGPU_OK;
}

// phase 1 of a batch: candidates -> compaction -> min/max (+ suggested boundary) resolution (records without digests)
static int enqueue_cut(pbsgpu_engine *e, Slot &s, uint32_t cap) {
    s.cap = cap;
    s.dense_mode = cap >= cap_limit(e, pbsk::scan_tile_bytes(s.nbytes));
    CHK(s.seg_cnt.ensure((size_t)s.nseg * 4 + 16));
    CHK(s.seg_off.ensure((size_t)s.nseg * 4 + 16));
    CHK(s.recs.ensure((size_t)s.rec_cap * sizeof(pbsgpu_record) + 64));
    CHK(s.order.ensure((size_t)s.rec_cap * pbsk::kQueueDescBytes + 64));
    CHK(enqueue_candidates(e, s, s.dptr, s.nbytes, cap, s.nseg));
    uint32_t *sc = s.scalars.as<uint32_t>();
    const pbsgpu_segment *dsegs = s.segs_dev();
    pbsk::Suggested sg{};
    if (s.nsugg) {
        if (s.mapped_ctrl) {
            sg.offsets = s.h_sugg.as<uint64_t>();
            sg.index = reinterpret_cast<const uint32_t *>(s.h_sugg.as<uint8_t>() + (size_t)s.nsugg * 8);
        } else {
            sg.offsets = s.sugg.as<uint64_t>();
            sg.index = s.sugg_idx.as<uint32_t>();
        }
        sg.cmin = e->cfg.min;
        sg.feed = e->sugg_feed.load(std::memory_order_relaxed);
        sg.absolute = e->sugg_feed_abs.load(std::memory_order_relaxed);
        sg.origin = s.sugg_origin;
        sg.open_end = s.sugg_open_end ? 1u : 0u;
    }
    // one long stream, no suggested boundaries: follow the cut chain by pointer doubling (one workgroup, ~20 rounds)
    // instead of walking it chunk by chunk on one wave (4.6 ms per 64 GiB, 11 ms next to SHA waves)
    constexpr uint32_t kParNodes = 1u << 18, kParLevels = 19;
    const uint64_t par_min = e->opt.resolve_par_min;  // smallest stream that takes the parallel path (~0: never)
    const bool par_off = par_min == ~0ull;
    // expected candidates = 3 per (mask + 1) bytes; twice that is the node budget (dense / crafted inputs fall back)
    const uint64_t expect = (uint64_t)(3.0 * (double)s.nbytes / ((double)e->cfg.mask + 1.0)) + 64;
    uint32_t big_nodes = 0, big_levels = 0;
    if (2 * expect + 2 > kParNodes && 2 * expect + 2 <= (1ull << 22)) {  // many candidates: the grid-wide variant
        big_nodes = 1u << 19;
        while (big_nodes < 2 * expect + 2) big_nodes <<= 1;
        while ((1ull << big_levels) < (uint64_t)big_nodes + 1) ++big_levels;
    }
    // at the capacity limit some tile may overflow: only the serial walks know how to ask such a tile (DenseTiles)
    pbsk::DenseTiles dzv{};
    const uint32_t lead = (uint32_t)((uintptr_t)s.dptr & 127u);
    const uint32_t tile_bytes = pbsk::scan_tile_bytes(s.nbytes);
    const uint64_t ntiles = (s.nbytes + lead + tile_bytes - 1) / tile_bytes;
    if (s.dense_mode) {
        dzv.tile_cnt = s.tile_cnt.as<uint32_t>();
        dzv.cap = cap;
        dzv.tile_bytes = tile_bytes;
        dzv.table_rot = e->d_table_rot;
        dzv.thr = e->thr;
        dzv.base = s.dptr;
    }
    const pbsk::DenseTiles *dz = s.dense_mode ? &dzv : nullptr;
    const bool par_ok = s.nseg == 1 && !s.nsugg && !par_off && s.nbytes >= par_min && !s.dense_mode;
    if (par_ok && big_nodes && s.par.ensure(pbsk::resolve_par_scratch_bytes(big_nodes, big_levels)) == PBSGPU_OK) {
        HIPCHK(pbsk::launch_resolve_single_par_grid(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, e->effmin, e->cfg.max,
                                                    sc + SC_ZERO, sc + SC_NREC, s.recs.as<pbsgpu_record>(), s.rec_cap,
                                                    s.par.p, big_nodes, big_levels, sc + SC_PARFB, sc + SC_PARHOPS,
                                                    s.stream));
    } else if (par_ok && !big_nodes && s.par.ensure(pbsk::resolve_par_scratch_bytes(kParNodes, kParLevels)) == PBSGPU_OK) {
        HIPCHK(pbsk::launch_resolve_single_par(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, e->effmin, e->cfg.max,
                                               sc + SC_ZERO, sc + SC_NREC, s.recs.as<pbsgpu_record>(), s.rec_cap, s.par.p,
                                               kParNodes, kParLevels, sc + SC_PARFB, s.stream));
    } else if (s.nseg == 1) {  // one stream: records start at 0, a single walk writes them and their count
        HIPCHK(pbsk::launch_resolve_single(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, e->effmin, e->cfg.max,
                                           sc + SC_ZERO, sc + SC_NREC, s.recs.as<pbsgpu_record>(), s.rec_cap, sg,
                                           s.stream, dz, lead, ntiles));
    } else {
        HIPCHK(pbsk::launch_resolve_count(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, s.nseg, e->effmin,
                                          e->cfg.max, s.seg_cnt.as<uint32_t>(), sg, s.stream, dz, lead, ntiles));
        HIPCHK(pbsk::launch_exclusive_scan(s.seg_cnt.as<uint32_t>(), s.nseg, 0xffffffffu, s.seg_off.as<uint32_t>(),
                                           sc + SC_NREC, nullptr, s.scan_tmp.as<uint32_t>(), s.stream));
        HIPCHK(pbsk::launch_resolve_write(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, s.nseg, e->effmin,
                                          e->cfg.max, s.seg_off.as<uint32_t>(), s.recs.as<pbsgpu_record>(),
                                          s.rec_cap, sg, s.stream, dz, lead, ntiles));
    }
    HIPCHK(hipEventRecord(s.ev[EV_RESOLVE1], s.stream));
    return PBSGPU_OK;
}

// phase 2: longest-first queue + SHA-256 of the first *SC_NREC records
static int enqueue_hash(pbsgpu_engine *e, Slot &s) {
    uint32_t *sc = s.scalars.as<uint32_t>();
    const pbsgpu_segment *dsegs = s.segs_dev();
    HIPCHK(pbsk::launch_order(s.dptr, dsegs, s.recs.as<pbsgpu_record>(), sc + SC_NREC, e->cfg.max, s.order.as<uint4>(),
                              sc + SC_WGLIMIT, e->num_cus, s.dense_mode ? nullptr : sc + SC_MAXCNT, s.cap, e->sha_slack_pct,
                              s.stream));  // (dense mode: the cut list is exact whatever the tiles found — hash it)
    // form of the hash kernel: issue-bound (dense) or chain-bound (sparse), from what the host knows at submit time —
    // the bytes of the batch and the longest chain the chunker can produce
    const uint64_t longest = std::min<uint64_t>(e->cfg.max, std::max<uint64_t>(s.nbytes, 1)) / 64 + 1;
    const bool dense = pbsk::sha256_dense_pays(s.nbytes / 64, longest, e->num_cus, e->opt.sha_dense_pct);
    HIPCHK(pbsk::launch_sha256_records(s.recs.as<pbsgpu_record>(), sc + SC_NREC, sc + SC_QUEUE, s.order.as<uint4>(),
                                       sc + SC_WGLIMIT, e->num_cus, dense, (int)e->opt.sha_form, s.stream));
    HIPCHK(hipEventRecord(s.ev[EV_SHA1], s.stream));
    return PBSGPU_OK;
}

// enqueue the whole pipeline for the slot's current (dptr, nbytes, segs) at capacity `cap`
static int enqueue_pipeline(pbsgpu_engine *e, Slot &s, uint32_t cap) {
    CHK(enqueue_cut(e, s, cap));
    CHK(enqueue_hash(e, s));
    HIPCHK(pbsk::launch_publish(s.h_scalars.p, s.scalars.p, SC_COUNT * 4, s.stream));
    // the finished records too (a ticket's results are then a plain host memcpy away); very large record sets
    // (tiny average chunk over a huge batch) keep the copy engine
    s.recs_published = false;
    if (s.rec_cap * sizeof(pbsgpu_record) <= (64u << 20) && s.h_recs.ensure((size_t)s.rec_cap * sizeof(pbsgpu_record) + 64) == PBSGPU_OK) {
        HIPCHK(pbsk::launch_publish_records(s.h_recs.as<pbsgpu_record>(), s.recs.as<pbsgpu_record>(),
                                            s.scalars.as<uint32_t>() + SC_NREC, s.rec_cap, s.stream));
        s.recs_published = true;
    }
    return PBSGPU_OK;
}

static Slot *acquire_pool_slot(pbsgpu_engine *e) {
    std::lock_guard<std::mutex> lk(e->mu);
    for (auto &s : e->slots)
        if (!s->busy) {
            s->busy = true;
            s->ready = false;
            e->tickets_out.fetch_add(1, std::memory_order_relaxed);
            return s.get();
        }
    return nullptr;
}

// `ticket` != 0: release only if the slot still belongs to that ticket. A second collect of the same ticket that got
// past with_ticket's lookup before the first one finished must not free a slot that a NEW submit has acquired meanwhile.
static void release_pool_slot(pbsgpu_engine *e, Slot *s, uint64_t ticket = 0) {
    std::lock_guard<std::mutex> lk(e->mu);
    if (ticket && s->ticket != ticket) return;
    if (s->busy) e->tickets_out.fetch_sub(1, std::memory_order_relaxed);
    s->busy = false;
    s->ready = false;
    s->ticket = 0;
}

// run fn on the slot that owns `ticket`, serialised against other calls on the same ticket
template <typename F>
static int with_ticket(pbsgpu_engine *e, uint64_t ticket, F fn) {
    Slot *s = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        for (auto &c : e->slots)
            if (c->busy && c->ready && c->ticket == ticket) s = c.get();
    }
    if (!s) return PBSGPU_E_TICKET;
    std::lock_guard<std::mutex> op(s->op);
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!s->busy || !s->ready || s->ticket != ticket) return PBSGPU_E_TICKET;  // collected meanwhile
    }
    CHK(set_device(e));
    return fn(*s);
}

// copy the caller's segment table (or the implicit single segment) and suggested boundaries to the slot
int stage_segments(pbsgpu_engine *e, Slot &s, const pbsgpu_segment *segs, uint32_t nseg, uint64_t nbytes,
                   const SuggestedHost *sg) {
    pbsgpu_segment whole{0, nbytes};
    if (segs == nullptr || nseg == 0) {
        segs = &whole;
        nseg = 1;
    } else {
        CHK(validate_segments(segs, nseg, nbytes));
    }
    CHK(s.h_segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    std::memcpy(s.h_segs.p, segs, (size_t)nseg * sizeof(pbsgpu_segment));
    if (!s.mapped_ctrl) {
        CHK(s.segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
        HIPCHK(hipMemcpyAsync(s.segs.p, s.h_segs.p, (size_t)nseg * sizeof(pbsgpu_segment), hipMemcpyHostToDevice,
                              s.stream));
    }
    s.nseg = nseg;
    s.nsugg = 0;
    s.sugg_origin = sg ? sg->origin : 0;
    s.sugg_open_end = sg ? sg->open_end : false;
    if (sg && sg->offsets && sg->index) {
        if (sg->index[0] != 0) return PBSGPU_E_INVALID;
        for (uint32_t i = 0; i < nseg; ++i) {
            if (sg->index[i + 1] < sg->index[i]) return PBSGPU_E_INVALID;
            for (uint32_t k = sg->index[i] + 1; k < sg->index[i + 1]; ++k)
                if (sg->offsets[k] < sg->offsets[k - 1]) return PBSGPU_E_INVALID;  // ascending per segment
        }
        const uint64_t n = sg->index[nseg];
        if (n) {
            const size_t ob = (size_t)n * 8, ib = ((size_t)nseg + 1) * 4;
            CHK(s.h_sugg.ensure(ob + ib));
            std::memcpy(s.h_sugg.p, sg->offsets, ob);
            std::memcpy(s.h_sugg.as<uint8_t>() + ob, sg->index, ib);
            if (!s.mapped_ctrl) {
                CHK(s.sugg.ensure(ob + 16));
                CHK(s.sugg_idx.ensure(ib + 16));
                HIPCHK(hipMemcpyAsync(s.sugg.p, s.h_sugg.p, ob, hipMemcpyHostToDevice, s.stream));
                HIPCHK(hipMemcpyAsync(s.sugg_idx.p, s.h_sugg.as<uint8_t>() + ob, ib, hipMemcpyHostToDevice, s.stream));
            }
            s.nsugg = n;
        }
    }
    s.rec_cap = record_upper_bound(e, s.h_segs.as<pbsgpu_segment>(), nseg, s.nsugg);
    return PBSGPU_OK;
}

// host -> device through the slot's two pinned staging buffers (caller memory is not referenced after return)
int staged_h2d(Slot &s, void *dst, const void *src, uint64_t nbytes, hipStream_t st) {
    const uint8_t *h = static_cast<const uint8_t *>(src);
    uint8_t *d = static_cast<uint8_t *>(dst);
    int which = 0;
    uint64_t off = 0;
    while (off < nbytes) {
        const size_t n = (size_t)std::min<uint64_t>(kStageBytes, nbytes - off);
        CHK(s.stage[which].ensure(std::min<uint64_t>(kStageBytes, std::max<uint64_t>(nbytes, 4096))));
        HIPCHK(hipEventSynchronize(s.stage_ev[which]));
        std::memcpy(s.stage[which].p, h + off, n);
        HIPCHK(hipMemcpyAsync(d + off, s.stage[which].p, n, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(s.stage_ev[which], st));
        off += n;
        which ^= 1;
    }
    return PBSGPU_OK;
}

// *_device entry points borrow a DEVICE pointer; a host slice handed in by mistake must not reach a kernel
bool is_device_pointer(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;  // unregistered host memory
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged || attr.type == hipMemoryTypeUnified;
}

static int submit_common(pbsgpu_engine *e, const void *ptr, bool host, uint64_t nbytes, const pbsgpu_segment *segs,
                         uint32_t nseg, const SuggestedHost *sg, uint64_t *ticket) {
    if (!e || !ticket || (!ptr && nbytes)) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    if (!host && nbytes && !is_device_pointer(ptr)) return PBSGPU_E_INVALID;
    Slot *s = acquire_pool_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    int st = PBSGPU_OK;
    {
        std::lock_guard<std::mutex> op(s->op);
        st = [&]() -> int {
            CHK(s->h_scalars.ensure(SC_COUNT * 4 + 64));
            CHK(stage_segments(e, *s, segs, nseg, nbytes, sg));
            HIPCHK(hipEventRecord(s->ev[EV_BEGIN], s->stream));
            if (host) {
                CHK(s->data.ensure((size_t)nbytes + 64));
                CHK(staged_h2d(*s, s->data.p, ptr, nbytes, s->stream));
                s->dptr = s->data.as<uint8_t>();
            } else {
                s->dptr = static_cast<const uint8_t *>(ptr);
            }
            s->nbytes = nbytes;
            s->host_submit = host;
            s->retries = 0;
            s->synced = false;
            return enqueue_pipeline(e, *s, default_cap(e, s->nbytes));
        }();
        if (st != PBSGPU_OK) (void)hipStreamSynchronize(s->stream);
    }
    if (st != PBSGPU_OK) {
        release_pool_slot(e, s);
        return st;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    s->ticket = e->next_ticket++;
    s->ready = true;
    *ticket = s->ticket;
    return PBSGPU_OK;
}

// wait for a slot; re-run with a larger per-tile capacity if any tile overflowed — up to the capacity limit (cap_limit): a
// batch scanned THERE is resolved exactly whatever its tiles found (enqueue_cut: dense_mode), so at most one more run follows
static int sync_slot(pbsgpu_engine *e, Slot &s) {
    if (s.synced) return PBSGPU_OK;
    const uint32_t tile_bytes = pbsk::scan_tile_bytes(s.nbytes);
    const uint32_t limit = cap_limit(e, tile_bytes);
    for (;;) {
        HIPCHK(hipStreamSynchronize(s.stream));
        const uint32_t *hs = s.h_scalars.as<uint32_t>();
        if (hs[SC_MAXCNT] <= s.cap || s.dense_mode) {
            if (s.dense_mode && hs[SC_MAXCNT] <= s.cap / 2 && e->cap_hint_tile.load(std::memory_order_relaxed) == tile_bytes) {
                // the corpus has calmed down: the next batch starts lower again (and gets the parallel resolve back)
                uint32_t cap = 8;
                while (cap < hs[SC_MAXCNT]) cap <<= 1;
                e->cap_hint.store(cap, std::memory_order_relaxed);  // (default_cap never goes below the nominal figure)
            }
            s.ncand = hs[SC_NCAND];
            s.nrec = hs[SC_NREC];
            break;
        }
        uint32_t cap = s.cap;
        while (cap < hs[SC_MAXCNT] && cap < limit) cap <<= 1;
        if (cap > limit) cap = limit;
        e->cap_hint_tile.store(tile_bytes, std::memory_order_relaxed);
        e->cap_hint.store(cap, std::memory_order_relaxed);
        s.retries++;
        int st = enqueue_pipeline(e, s, cap);
        if (st != PBSGPU_OK) return st;
    }
    if (s.nrec > s.rec_cap) return PBSGPU_E_STATE;
    s.synced = true;
    return PBSGPU_OK;
}

// run scan + compaction on `s` and wait; grows the per-tile capacity until nothing overflowed
int candidates_sync(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, uint64_t *count) {
    CHK(s.h_scalars.ensure(SC_COUNT * 4 + 64));
    uint32_t tcap = default_cap(e, nbytes);
    for (;;) {
        CHK(enqueue_candidates(e, s, dptr, nbytes, tcap));
        HIPCHK(pbsk::launch_publish(s.h_scalars.p, s.scalars.p, SC_COUNT * 4, s.stream));
        HIPCHK(hipStreamSynchronize(s.stream));
        const uint32_t *hs = s.h_scalars.as<uint32_t>();
        if (hs[SC_MAXCNT] <= tcap) break;
        while (tcap < hs[SC_MAXCNT]) tcap <<= 1;
        if (tcap > pbsk::scan_tile_bytes(nbytes)) tcap = pbsk::scan_tile_bytes(nbytes);
    }
    *count = s.h_scalars.as<uint32_t>()[SC_NCAND];
    return PBSGPU_OK;
}

}  // namespace pbse

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

int pbsgpu_last_hip_error(void) { return g_last_hip_error.load(); }

int pbsgpu_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_last_hip_error.store((int)e);
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// DEBUG overrides of pbsgpu_engine_options by environment (the variable names of rounds 1-5): ONE table, one getenv
static void engine_env_overrides(pbsgpu_engine_options &o) {
    enum Kind { U32, U32_ZERO_OFF, U64, F64, SHA_MODE, RESOLVE_SERIAL };
    struct Entry {
        const char *name;
        Kind kind;
        void *field;
    };
    const Entry table[] = {
        {"PBSGPU_SHA_MODE", SHA_MODE, &o.sha_form},
        {"PBSGPU_SHA_SLACK_PCT", U32, &o.sha_slack_pct},  // (value + 1, see below)
        {"PBSGPU_SHA_DENSE_PCT", U32_ZERO_OFF, &o.sha_dense_pct},
        {"PBSGPU_RESOLVE_PAR_MIN", U64, &o.resolve_par_min},
        {"PBSGPU_RESOLVE_SERIAL", RESOLVE_SERIAL, &o.resolve_par_min},
        {"PBSGPU_SHA_MANY_FILES_PER_CORE", U32, &o.sha_many_files_per_core},
        {"PBSGPU_STREAM_SHA_CUS", U32, &o.stream_sha_cus},
        {"PBSGPU_STREAM_XP_CUS", U32_ZERO_OFF, &o.stream_express_cus},
        {"PBSGPU_STREAM_RING_SLOTS", U32, &o.stream_ring_slots},
        {"PBSGPU_STREAM_CTX_POOL", U32_ZERO_OFF, &o.stream_ctx_pool},
        {"PBSGPU_STREAM_RING_GIB", F64, &o.stream_ring_gib},
        {"PBSGPU_STREAM_PAGE_BYTES", U64, &o.stream_page_bytes},
    };
    for (const Entry &e : table) {
        const char *v = getenv(e.name);
        if (!v || !*v) continue;
        switch (e.kind) {
        case U32:
            *static_cast<uint32_t *>(e.field) = (uint32_t)std::max(0L, atol(v)) + (e.field == &o.sha_slack_pct ? 1u : 0u);
            break;
        case U32_ZERO_OFF: *static_cast<uint32_t *>(e.field) = atol(v) <= 0 ? 0xffffffffu : (uint32_t)atol(v); break;
        case U64: *static_cast<uint64_t *>(e.field) = strtoull(v, nullptr, 10); break;
        case F64: *static_cast<double *>(e.field) = std::max(0.0, atof(v)); break;
        case SHA_MODE: *static_cast<uint32_t *>(e.field) = v[0] == 'l' ? 1u : v[0] == 'x' ? 2u : 0u; break;
        case RESOLVE_SERIAL: *static_cast<uint64_t *>(e.field) = ~0ull; break;
        }
    }
}

int pbsgpu_engine_create(int device, const pbsgpu_config *cfg, uint32_t inflight, pbsgpu_engine **out) {
    pbsgpu_engine_options o{};
    o.inflight = inflight;
    return pbsgpu_engine_create_opt(device, cfg, &o, out);
}

int pbsgpu_engine_create_opt(int device, const pbsgpu_config *cfg, const pbsgpu_engine_options *opt, pbsgpu_engine **out) {
    if (!cfg || !out) return PBSGPU_E_INVALID;
    *out = nullptr;
    pbsgpu_engine_options o{};
    if (opt) o = *opt;
    engine_env_overrides(o);
    uint32_t inflight = o.inflight;
    if (o.sha_form > 2) return PBSGPU_E_INVALID;
    // the candidate/resolve split needs: window 64, mask = 2^k - 1, min >= window, max > min
    if (cfg->window != pbsk::kWindow) return PBSGPU_E_INVALID;
    const uint64_t m1 = (uint64_t)cfg->mask + 1;
    if ((m1 & (m1 - 1)) != 0 || m1 < 2 || m1 > (1ull << 31)) return PBSGPU_E_INVALID;
    if (cfg->break_min > cfg->mask) return PBSGPU_E_INVALID;
    if (cfg->min < pbsk::kWindow || cfg->max <= cfg->min || cfg->max < 128) return PBSGPU_E_INVALID;
    if (inflight == 0) inflight = 2;
    if (inflight > 16) return PBSGPU_E_INVALID;

    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_last_hip_error.store((int)he);
        (void)hipGetLastError();
        return PBSGPU_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    pbsgpu_engine *e = new (std::nothrow) pbsgpu_engine();
    if (!e) return PBSGPU_E_NOMEM;
    e->device = device;
    e->cfg = *cfg;
    uint32_t bits = 0;
    while ((1ull << bits) < m1) ++bits;
    e->bits = bits;
    e->thr = cfg->break_min << (32 - bits);
    e->effmin = std::max<uint32_t>(cfg->min, pbsk::kWindow + 1);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        if (prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
        // the code objects in this library are gfx950 (MI355X / CDNA4) only: fail here, not at the first launch
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && getenv("PBSGPU_ALLOW_ANY_ARCH") == nullptr) {
            delete e;
            return PBSGPU_E_NO_DEVICE;
        }
    }

    uint32_t rot[256];
    const uint32_t r = (32 - bits) & 31;
    for (int i = 0; i < 256; ++i) rot[i] = r ? ((cfg->table[i] << r) | (cfg->table[i] >> (32 - r))) : cfg->table[i];
    int st = PBSGPU_OK;
    do {
        if (hipMalloc(reinterpret_cast<void **>(&e->d_table_rot), sizeof(rot)) != hipSuccess) { st = PBSGPU_E_NOMEM; break; }
        if (hipMemcpy(e->d_table_rot, rot, sizeof(rot), hipMemcpyHostToDevice) != hipSuccess) { st = PBSGPU_E_HIP; break; }
        e->sha_slack_pct = inflight > 4 ? 0u : 25u;
        if (o.sha_slack_pct) e->sha_slack_pct = std::min(o.sha_slack_pct - 1u, 400u);
        // every default resolved once: the rest of the library reads e->opt
        if (o.sha_dense_pct == 0) o.sha_dense_pct = 150;
        if (o.sha_dense_pct == 0xffffffffu) o.sha_dense_pct = 0;
        if (o.resolve_par_min == 0) o.resolve_par_min = 64ull << 20;
        if (o.sha_many_files_per_core == 0) o.sha_many_files_per_core = 55;
        if (o.stream_ring_slots == 0) o.stream_ring_slots = 256;
        o.stream_ring_slots = std::max(4u, std::min(o.stream_ring_slots, 4096u));
        if (o.stream_ctx_pool == 0) o.stream_ctx_pool = 8;
        if (o.stream_ctx_pool == 0xffffffffu) o.stream_ctx_pool = 0;
        if (o.stream_ring_gib <= 0) o.stream_ring_gib = 48.0;
        o.inflight = inflight;
        e->opt = o;
        for (uint32_t i = 0; i < inflight && st == PBSGPU_OK; ++i) {
            e->slots.emplace_back(new (std::nothrow) Slot());
            st = e->slots.back() ? e->slots.back()->init() : PBSGPU_E_NOMEM;
        }
        constexpr int kAux = 4;  // concurrent synchronous helper calls (verification keeps 4 files in flight)
        for (int i = 0; i < kAux && st == PBSGPU_OK; ++i) {
            e->aux.emplace_back(new (std::nothrow) Slot());
            st = e->aux.back() ? e->aux.back()->init() : PBSGPU_E_NOMEM;
        }
        e->aux_busy.assign(e->aux.size(), 0);
        for (int i = 0; i < 2 && st == PBSGPU_OK; ++i) {
            hipStream_t cs = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) st = PBSGPU_E_HIP;
            else e->copy_streams.push_back(cs);
        }
        for (int i = 0; i < 2 && st == PBSGPU_OK; ++i) {
            hipStream_t ts = nullptr;
            if (hipStreamCreateWithFlags(&ts, hipStreamNonBlocking) != hipSuccess) st = PBSGPU_E_HIP;
            else e->tee_streams.push_back(ts);
        }
    } while (0);
    if (st != PBSGPU_OK) {
        pbsgpu_engine_destroy(e);
        return st;
    }
    *out = e;
    return PBSGPU_OK;
}

// Streams and chunkers created from the engine keep it alive: the engine's memory is released when the last of
// them has been destroyed too (a Go finalizer or Python __del__ may run in any order).
void pbsgpu_engine_destroy(pbsgpu_engine *e) {
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->destroyed) return;
        e->destroyed = true;
    }
    engine_unref(e);
}

// Give back what the engine only keeps for re-use: the page ring of its payload streams (when no stream is alive) and the
// parked contexts of closed streams (hipFree waits for the device to go idle, so this is for quiet moments — between
// backup jobs, or when another allocation has failed). *freed_bytes = device memory that came back.
int pbsgpu_engine_trim(pbsgpu_engine *e, uint64_t *freed_bytes) {
    if (!e) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    uint64_t freed = 0;
    {   // the engine's page ring (arena + tables), unless a payload stream is alive
        std::lock_guard<std::mutex> lk(e->sring_mu);
        if (e->sring && e->sring_users == 0) {
            size_t f0 = 0, f1 = 0, tot = 0;
            (void)hipMemGetInfo(&f0, &tot);
            engine_ring_release(e);
            (void)hipMemGetInfo(&f1, &tot);
            if (f1 > f0) freed += f1 - f0;
        }
    }
    stream_pool_release(e);
    if (freed_bytes) *freed_bytes = freed;
    return PBSGPU_OK;
}

// SHA-256 is serial inside a file: one GPU lane hashes one file at 64 B per ~1.66 us = 0.036 GiB/s whatever its size,
// a SHA-NI host core does ~2 GiB/s. A whole-file batch therefore only beats `host_cores` cores with more than
// ~55 files per core in flight (measured: profiles/r02_verify_workload.log). The reference's verify
// job keeps 4 files in flight (internal/server/verification/job.go:493): a drop-in that sent those to the GPU would
// be ~50x slower than the code it replaces, so bindings ask first.
int pbsgpu_sha256_many_pays(const pbsgpu_engine *e, uint32_t nfiles, uint32_t host_cores, int *pays) {
    if (!e || !pays) return PBSGPU_E_INVALID;
    const uint32_t per_core = std::max(1u, e->opt.sha_many_files_per_core);
    if (host_cores == 0) host_cores = 1;
    *pays = (uint64_t)nfiles > (uint64_t)per_core * host_cores ? 1 : 0;
    return PBSGPU_OK;
}

// Which reader the suggested-boundary rule emulates (see pbsgpu.h). Applies to *_suggested submits and
// pbsgpu_stream_suggest cuts enqueued after the call.
int pbsgpu_engine_set_suggested_feed(pbsgpu_engine *e, uint64_t feed_bytes, int absolute_grid) {
    if (!e) return PBSGPU_E_INVALID;
    e->sugg_feed.store(feed_bytes == 0 ? ~0ull : feed_bytes, std::memory_order_relaxed);
    e->sugg_feed_abs.store(absolute_grid ? 1u : 0u, std::memory_order_relaxed);
    return PBSGPU_OK;
}

int pbsgpu_engine_config(const pbsgpu_engine *e, pbsgpu_config *out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    *out = e->cfg;
    return PBSGPU_OK;
}

int pbsgpu_submit_device(pbsgpu_engine *e, const void *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
                         uint32_t nseg, uint64_t *ticket) {
    return submit_common(e, dptr, false, nbytes, segs, nseg, nullptr, ticket);
}

int pbsgpu_submit_host(pbsgpu_engine *e, const void *hptr, uint64_t nbytes, const pbsgpu_segment *segs,
                       uint32_t nseg, uint64_t *ticket) {
    return submit_common(e, hptr, true, nbytes, segs, nseg, nullptr, ticket);
}

int pbsgpu_submit_device_suggested(pbsgpu_engine *e, const void *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
                                   uint32_t nseg, const uint64_t *suggested, const uint32_t *suggested_index,
                                   uint64_t *ticket) {
    SuggestedHost sg{suggested, suggested_index};
    return submit_common(e, dptr, false, nbytes, segs, nseg, &sg, ticket);
}

int pbsgpu_submit_host_suggested(pbsgpu_engine *e, const void *hptr, uint64_t nbytes, const pbsgpu_segment *segs,
                                 uint32_t nseg, const uint64_t *suggested, const uint32_t *suggested_index,
                                 uint64_t *ticket) {
    SuggestedHost sg{suggested, suggested_index};
    return submit_common(e, hptr, true, nbytes, segs, nseg, &sg, ticket);
}

int pbsgpu_wait(pbsgpu_engine *e, uint64_t ticket, uint64_t *nrecords) {
    if (!e) return PBSGPU_E_INVALID;
    return with_ticket(e, ticket, [&](Slot &s) -> int {
        CHK(sync_slot(e, s));
        if (nrecords) *nrecords = s.nrec;
        return PBSGPU_OK;
    });
}

int pbsgpu_ticket_done(pbsgpu_engine *e, uint64_t ticket, int *done) {
    if (!e || !done) return PBSGPU_E_INVALID;
    Slot *s = nullptr;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        for (auto &c : e->slots)
            if (c->busy && c->ready && c->ticket == ticket) s = c.get();
    }
    if (!s) return PBSGPU_E_TICKET;
    // never blocks: if another thread is waiting on / collecting this very ticket, it simply is not done yet for us
    std::unique_lock<std::mutex> op(s->op, std::try_to_lock);
    if (!op.owns_lock()) {
        *done = 0;
        return PBSGPU_OK;
    }
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!s->busy || !s->ready || s->ticket != ticket) return PBSGPU_E_TICKET;
    }
    if (s->synced) {
        *done = 1;
        return PBSGPU_OK;
    }
    CHK(set_device(e));
    const hipError_t q = hipStreamQuery(s->stream);
    if (q == hipErrorNotReady) {
        (void)hipGetLastError();
        *done = 0;
        return PBSGPU_OK;
    }
    HIPCHK(q);
    *done = 1;
    return PBSGPU_OK;
}

int pbsgpu_collect(pbsgpu_engine *e, uint64_t ticket, pbsgpu_record *out, uint64_t cap, uint64_t *nrecords) {
    if (!e) return PBSGPU_E_INVALID;
    // The ticket is released on success and on every hard error (E_CAPACITY keeps it valid for a retry) — INSIDE the
    // ticket's critical section (Slot::op held by with_ticket): a concurrent collect / wait of the same ticket is either
    // serialised in front of this one or finds the ticket gone (E_TICKET), never a half-released slot.
    int st = with_ticket(e, ticket, [&](Slot &s) -> int {
        const int r = [&]() -> int {
        CHK(sync_slot(e, s));
        if (nrecords) *nrecords = s.nrec;
        if (s.nrec > cap || (!out && s.nrec)) return PBSGPU_E_CAPACITY;
        static const bool trace = getenv("PBSGPU_TRACE") != nullptr;  // ingest log line, like tapeio's MB/s progress
        if (trace) {
            float scan = 0, res = 0, sha = 0;
            (void)hipEventElapsedTime(&scan, s.ev[EV_SCAN0], s.ev[EV_SCAN1]);
            (void)hipEventElapsedTime(&res, s.ev[EV_SCAN1], s.ev[EV_RESOLVE1]);
            (void)hipEventElapsedTime(&sha, s.ev[EV_RESOLVE1], s.ev[EV_SHA1]);
            (void)hipGetLastError();
            fprintf(stderr, "[pbsgpu] ticket %llu: %.2f MiB, %llu candidates, %llu chunks, scan %.3f ms, resolve %.3f ms, "
                            "sha256 %.3f ms, retries %u\n",
                    (unsigned long long)s.ticket, s.nbytes / 1048576.0, (unsigned long long)s.ncand,
                    (unsigned long long)s.nrec, scan, res, sha, s.retries);
        }
        if (s.nrec) {
            if (s.recs_published) {
                std::memcpy(out, s.h_recs.p, (size_t)s.nrec * sizeof(pbsgpu_record));
            } else {
                HIPCHK(hipMemcpyAsync(out, s.recs.p, (size_t)s.nrec * sizeof(pbsgpu_record), hipMemcpyDeviceToHost, s.stream));
                HIPCHK(hipStreamSynchronize(s.stream));
            }
        }
        return PBSGPU_OK;
        }();
        if (r != PBSGPU_E_CAPACITY) release_pool_slot(e, &s, ticket);
        return r;
    });
    return st;
}

int pbsgpu_ticket_timing(pbsgpu_engine *e, uint64_t ticket, pbsgpu_timing *out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    return with_ticket(e, ticket, [&](Slot &s) -> int {
        CHK(sync_slot(e, s));
        std::memset(out, 0, sizeof(*out));
        float ms = 0;
        if (s.retries == 0 && hipEventElapsedTime(&ms, s.ev[EV_BEGIN], s.ev[EV_SCAN0]) == hipSuccess) out->h2d_ms = ms;
        if (hipEventElapsedTime(&ms, s.ev[EV_SCAN0], s.ev[EV_SCAN1]) == hipSuccess) out->scan_ms = ms;
        if (hipEventElapsedTime(&ms, s.ev[EV_SCAN1], s.ev[EV_RESOLVE1]) == hipSuccess) out->resolve_ms = ms;
        if (hipEventElapsedTime(&ms, s.ev[EV_RESOLVE1], s.ev[EV_SHA1]) == hipSuccess) out->sha_ms = ms;
        if (hipEventElapsedTime(&ms, s.ev[EV_SCAN0], s.ev[EV_SHA1]) == hipSuccess) out->total_ms = ms;
        (void)hipGetLastError();
        out->ncandidates = s.ncand;
        out->nrecords = s.nrec;
        out->retries = s.retries;
        return PBSGPU_OK;
    });
}

int pbsgpu_candidates_device(pbsgpu_engine *e, const void *dptr, uint64_t nbytes, uint64_t *out, uint64_t cap,
                             uint64_t *n) {
    if (!e || !n || (!dptr && nbytes)) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    if (nbytes && !is_device_pointer(dptr)) return PBSGPU_E_INVALID;
    AuxLease lease(e);
    Slot *s = lease.s;
    uint64_t cnt = 0;
    CHK(candidates_sync(e, *s, static_cast<const uint8_t *>(dptr), nbytes, &cnt));
    *n = cnt;
    if (cnt > cap || (!out && cnt)) return PBSGPU_E_CAPACITY;
    if (cnt) HIPCHK(hipMemcpy(out, s->dense.p, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    return PBSGPU_OK;
}

int pbsgpu_resolve_candidates(pbsgpu_engine *e, const uint64_t *cands, uint64_t ncand, uint64_t stream_len,
                              pbsgpu_record *out, uint64_t cap, uint64_t *nrecords) {
    if (!e || !nrecords || (ncand && !cands) || ncand >= (1ull << 32)) return PBSGPU_E_INVALID;
    for (uint64_t i = 1; i < ncand; ++i)
        if (cands[i] <= cands[i - 1]) return PBSGPU_E_INVALID;  // strictly ascending
    CHK(set_device(e));
    AuxLease lease(e);
    Slot *s = lease.s;
    pbsgpu_segment whole{0, stream_len};
    CHK(s->h_scalars.ensure(SC_COUNT * 4 + 64));
    CHK(stage_segments(e, *s, &whole, 1, stream_len, nullptr));
    CHK(s->recs.ensure((size_t)s->rec_cap * sizeof(pbsgpu_record) + 64));
    CHK(s->dense.ensure((size_t)ncand * 8 + 64));
    CHK(s->scalars.ensure(SC_COUNT * 4));
    HIPCHK(hipMemsetAsync(s->scalars.p, 0, SC_COUNT * 4, s->stream));
    if (ncand) CHK(staged_h2d(*s, s->dense.p, cands, ncand * 8, s->stream));
    uint32_t *hn = s->h_scalars.as<uint32_t>() + SC_COUNT;
    *hn = (uint32_t)ncand;
    uint32_t *sc = s->scalars.as<uint32_t>();
    HIPCHK(hipMemcpyAsync(sc + SC_NCAND, hn, 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(pbsk::launch_resolve_single(s->dense.as<uint64_t>(), sc + SC_NCAND, s->segs.as<pbsgpu_segment>(), e->effmin,
                                       e->cfg.max, sc + SC_ZERO, sc + SC_NREC, s->recs.as<pbsgpu_record>(), s->rec_cap,
                                       pbsk::Suggested{}, s->stream));
    HIPCHK(pbsk::launch_publish(s->h_scalars.p, s->scalars.p, SC_COUNT * 4, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    const uint64_t n = s->h_scalars.as<uint32_t>()[SC_NREC];
    *nrecords = n;
    if (n > cap || (!out && n)) return PBSGPU_E_CAPACITY;
    if (n) {
        HIPCHK(hipMemcpy(out, s->recs.p, (size_t)n * sizeof(pbsgpu_record), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; ++i) std::memset(out[i].digest, 0, 32);  // not hashed here
    }
    return PBSGPU_OK;
}

// shared front half of the whole-range hash batches: segment table (+ host bytes) onto an aux slot
static int stage_ranges(pbsgpu_engine *e, Slot *s, const void *ptr, bool host, uint64_t nbytes, const pbsgpu_segment *segs,
                        uint32_t nseg, const uint8_t **d) {
    CHK(s->h_segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    std::memcpy(s->h_segs.p, segs, (size_t)nseg * sizeof(pbsgpu_segment));
    CHK(s->segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    HIPCHK(hipMemcpyAsync(s->segs.p, s->h_segs.p, (size_t)nseg * sizeof(pbsgpu_segment), hipMemcpyHostToDevice, s->stream));
    *d = static_cast<const uint8_t *>(ptr);
    if (host) {
        CHK(s->data.ensure((size_t)nbytes + 64));
        CHK(staged_h2d(*s, s->data.p, ptr, nbytes, s->stream));
        *d = s->data.as<uint8_t>();
    }
    CHK(s->scalars.ensure(SC_COUNT * 4));
    HIPCHK(hipMemsetAsync(s->scalars.p, 0, SC_COUNT * 4, s->stream));
    return PBSGPU_OK;
}

// results of a synchronous helper: device -> mapped pinned memory by kernel (not by the shared copy queue, which a
// copy waiting behind a long hash kernel would block for every other stream), then a host memcpy to the caller
static int fetch_result(Slot *s, void *dst, const void *src_dev, size_t nbytes) {
    CHK(s->h_recs.ensure(nbytes + 64));
    HIPCHK(pbsk::launch_publish(s->h_recs.p, src_dev, nbytes, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    std::memcpy(dst, s->h_recs.p, nbytes);
    return PBSGPU_OK;
}

static int sha256_many(pbsgpu_engine *e, const void *ptr, bool host, uint64_t nbytes, const pbsgpu_segment *segs,
                       uint32_t nseg, uint8_t *digests) {
    if (!e || (!ptr && nbytes) || (nseg && (!segs || !digests))) return PBSGPU_E_INVALID;
    if (nseg == 0) return PBSGPU_OK;
    for (uint32_t i = 0; i < nseg; ++i)
        if (segs[i].length > nbytes || segs[i].offset > nbytes - segs[i].length) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    AuxLease lease(e);
    Slot *s = lease.s;
    const uint8_t *d = nullptr;
    CHK(stage_ranges(e, s, ptr, host, nbytes, segs, nseg, &d));
    CHK(s->recs.ensure((size_t)nseg * 32));
    uint64_t total_blocks = 0, longest = 1;
    for (uint32_t i = 0; i < nseg; ++i) {
        const uint64_t blocks = (segs[i].length + 8) / 64 + 1;
        total_blocks += blocks;
        longest = std::max(longest, blocks);
    }
    HIPCHK(pbsk::launch_sha256_segments(d, s->segs.as<pbsgpu_segment>(), nseg, s->recs.as<uint8_t>(),
                                        s->scalars.as<uint32_t>() + SC_QUEUE, e->num_cus,
                                        pbsk::sha256_dense_pays(total_blocks, longest, e->num_cus, e->opt.sha_dense_pct),
                                        (int)e->opt.sha_form, s->stream));
    return fetch_result(s, digests, s->recs.p, (size_t)nseg * 32);
}

int pbsgpu_sha256_many_device(pbsgpu_engine *e, const void *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
                              uint32_t nseg, uint8_t *digests) {
    return sha256_many(e, dptr, false, nbytes, segs, nseg, digests);
}

int pbsgpu_sha256_many_host(pbsgpu_engine *e, const void *hptr, uint64_t nbytes, const pbsgpu_segment *segs,
                            uint32_t nseg, uint8_t *digests) {
    return sha256_many(e, hptr, true, nbytes, segs, nseg, digests);
}

static int xxh3_many(pbsgpu_engine *e, const void *ptr, bool host, uint64_t nbytes, const pbsgpu_segment *segs,
                     uint32_t nseg, uint64_t *out) {
    if (!e || (!ptr && nbytes) || (nseg && (!segs || !out))) return PBSGPU_E_INVALID;
    if (nseg == 0) return PBSGPU_OK;
    for (uint32_t i = 0; i < nseg; ++i)
        if (segs[i].length > nbytes || segs[i].offset > nbytes - segs[i].length) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    AuxLease lease(e);
    Slot *s = lease.s;
    const uint8_t *d = nullptr;
    CHK(stage_ranges(e, s, ptr, host, nbytes, segs, nseg, &d));
    // work items + plan (which blocks exist) on the host, block sums and chains on the device
    std::vector<pbsk::XxhItem> items(nseg);
    for (uint32_t i = 0; i < nseg; ++i) {
        items[i] = pbsk::XxhItem{};
        items[i].ptr = d + segs[i].offset;
        items[i].len = segs[i].length;
        items[i].flags = 3u;
        items[i].out = i;
    }
    const uint64_t total_blocks = pbsk::xxh3_plan_whole(items.data(), nseg);
    CHK(s->tile_slots.ensure((size_t)nseg * sizeof(pbsk::XxhItem) + 64));
    CHK(s->dense.ensure((size_t)total_blocks * 64 + 64));
    CHK(s->recs.ensure((size_t)nseg * 8 + 64));
    CHK(staged_h2d(*s, s->tile_slots.p, items.data(), (size_t)nseg * sizeof(pbsk::XxhItem), s->stream));
    HIPCHK(pbsk::launch_xxh3_items(s->tile_slots.as<pbsk::XxhItem>(), nseg, total_blocks, nullptr, s->dense.as<uint64_t>(),
                                   s->recs.as<uint64_t>(), s->scalars.as<uint32_t>() + SC_QUEUE, e->num_cus, s->stream));
    return fetch_result(s, out, s->recs.p, (size_t)nseg * 8);
}

int pbsgpu_xxh3_many_device(pbsgpu_engine *e, const void *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
                            uint32_t nseg, uint64_t *out) {
    return xxh3_many(e, dptr, false, nbytes, segs, nseg, out);
}

int pbsgpu_xxh3_many_host(pbsgpu_engine *e, const void *hptr, uint64_t nbytes, const pbsgpu_segment *segs,
                          uint32_t nseg, uint64_t *out) {
    return xxh3_many(e, hptr, true, nbytes, segs, nseg, out);
}

int pbsgpu_fill_device(pbsgpu_engine *e, void *dptr, uint64_t stream_off, uint64_t nbytes, uint64_t seed,
                       uint32_t kind) {
    if (!e || (!dptr && nbytes) || ((uintptr_t)dptr & 7u) || (stream_off & 7u) || kind > 4) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    AuxLease lease(e);
    HIPCHK(pbsk::launch_fill(dptr, stream_off, nbytes, seed, kind, lease.s->stream));
    HIPCHK(hipStreamSynchronize(lease.s->stream));
    return PBSGPU_OK;
}

int pbsgpu_device_alloc(pbsgpu_engine *e, uint64_t nbytes, void **dptr) {
    if (!e || !dptr) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    hipError_t he = hipMalloc(dptr, nbytes ? nbytes : 1);
    if (he != hipSuccess) {
        g_last_hip_error.store((int)he);
        (void)hipGetLastError();
        *dptr = nullptr;
        return PBSGPU_E_NOMEM;
    }
    return PBSGPU_OK;
}

int pbsgpu_device_free(pbsgpu_engine *e, void *dptr) {
    if (!e) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    if (dptr) dev_free(dptr);  // (parked while a page-ring service runs: hipFree would wait for it)
    return PBSGPU_OK;
}

int pbsgpu_memcpy_h2d(pbsgpu_engine *e, void *dptr, const void *hptr, uint64_t nbytes) {
    if (!e || ((!dptr || !hptr) && nbytes)) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    if (nbytes) HIPCHK(hipMemcpy(dptr, hptr, nbytes, hipMemcpyHostToDevice));
    return PBSGPU_OK;
}

int pbsgpu_memcpy_d2h(pbsgpu_engine *e, void *hptr, const void *dptr, uint64_t nbytes) {
    if (!e || ((!dptr || !hptr) && nbytes)) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    if (nbytes) HIPCHK(hipMemcpy(hptr, dptr, nbytes, hipMemcpyDeviceToHost));
    return PBSGPU_OK;
}

// Host -> device copy rate of this box through pinned memory (GB/s): what bounds every host-fed entry point.
int pbsgpu_measure_h2d(pbsgpu_engine *e, uint64_t nbytes, double *gb_per_s) {
    if (!e || !gb_per_s || nbytes < (1u << 20)) return PBSGPU_E_INVALID;
    CHK(set_device(e));
    AuxLease lease(e);
    Slot *s = lease.s;
    PinnedBuf h;
    DevBuf d;
    CHK(h.ensure(nbytes));
    int st = d.ensure(nbytes);
    if (st == PBSGPU_OK) {
        std::memset(h.p, 0x5a, nbytes);
        st = [&]() -> int {
            HIPCHK(hipMemcpyAsync(d.p, h.p, nbytes, hipMemcpyHostToDevice, s->stream));  // warm-up
            HIPCHK(hipEventRecord(s->ev[EV_BEGIN], s->stream));
            for (int i = 0; i < 4; ++i) HIPCHK(hipMemcpyAsync(d.p, h.p, nbytes, hipMemcpyHostToDevice, s->stream));
            HIPCHK(hipEventRecord(s->ev[EV_SHA1], s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, s->ev[EV_BEGIN], s->ev[EV_SHA1]));
            *gb_per_s = 4.0 * (double)nbytes / (ms * 1e-3) / 1e9;
            return PBSGPU_OK;
        }();
    }
    h.release();
    d.release();
    return st;
}

}  // extern "C"
