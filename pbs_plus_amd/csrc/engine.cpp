// libpbsgpu host side: C-ABI entry points (include/pbsgpu.h) over the gfx950 kernels.
//
// Stands where github.com/pbs-plus/pxar's backupproxy session runs its chunk loop for
// the writers the reference drives (internal/pxarmount/commit_orchestrate.go:137-177,
// internal/tapeio/converter.go:386-439). No CPU fallback exists in this file: every data
// path launches the HIP kernels; without a device the engine cannot be created.
#include "engine_internal.h"

#include <cstdio>
#include <cstdlib>

// The engine runs up to 16 batches on separate HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues
// (default 4) and reads the variable when the HIP runtime initialises, i.e. at the first HIP call of the process.
// Setting a default when this library is loaded covers hosts that bind the C ABI directly (cgo, JNI, ctypes)
// without going through the Python package; an explicit setting of the host always wins.
__attribute__((constructor)) static void pbsgpu_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "24", 0); }

namespace pbse {
std::atomic<int> g_last_hip_error{0};
}
using namespace pbse;


namespace pbse {

constexpr size_t kStageBytes = 32u << 20;

uint32_t default_cap(const pbsgpu_engine *e, uint64_t nbytes) {
    const uint32_t tile_bytes = pbsk::scan_tile_bytes(nbytes);
    // a corpus that needed a larger per-tile capacity once tends to need it again: start there
    if (e->cap_hint_tile == tile_bytes && e->cap_hint) return e->cap_hint;
    // expected candidates per wave tile = 3 * tile / (mask + 1); leave generous headroom
    const double lambda = 3.0 * tile_bytes / ((double)e->cfg.mask + 1.0);
    double want = 4.0 * lambda + 16.0;
    uint32_t cap = 8;
    while (cap < want) cap <<= 1;
    return cap;
}

int set_device(const pbsgpu_engine *e) {
    HIPCHK(hipSetDevice(e->device));
    return PBSGPU_OK;
}

uint64_t record_upper_bound(const pbsgpu_engine *e, const pbsgpu_segment *segs, uint32_t nseg) {
    uint64_t n = 0;
    for (uint32_t i = 0; i < nseg; ++i) n += segs[i].length / e->effmin + 1;
    return n;
}

int validate_segments(const pbsgpu_segment *segs, uint32_t nseg, uint64_t nbytes) {
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
    if (ntiles * (uint64_t)cap >= (1ull << 32)) return PBSGPU_E_DENSITY;
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

// enqueue the whole pipeline for the slot's current (dptr, nbytes, segs) at capacity `cap`
// phase 1 of a batch: candidates -> compaction -> min/max resolution (records without digests)
int enqueue_cut(pbsgpu_engine *e, Slot &s, uint32_t cap) {
    s.cap = cap;
    CHK(s.seg_cnt.ensure((size_t)s.nseg * 4 + 16));
    CHK(s.seg_off.ensure((size_t)s.nseg * 4 + 16));
    CHK(s.recs.ensure((size_t)s.rec_cap * sizeof(pbsgpu_record) + 64));
    CHK(s.order.ensure((size_t)s.rec_cap * 4 + 64));
    CHK(enqueue_candidates(e, s, s.dptr, s.nbytes, cap, s.nseg));
    uint32_t *sc = s.scalars.as<uint32_t>();
    const pbsgpu_segment *dsegs = s.segs.as<pbsgpu_segment>();
    if (s.nseg == 1) {  // one stream: records start at 0, a single walk writes them and their count
        HIPCHK(pbsk::launch_resolve_single(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, e->effmin, e->cfg.max,
                                           sc + SC_ZERO, sc + SC_NREC, s.recs.as<pbsgpu_record>(), s.rec_cap,
                                           s.stream));
    } else {
        HIPCHK(pbsk::launch_resolve_count(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, s.nseg, e->effmin,
                                          e->cfg.max, s.seg_cnt.as<uint32_t>(), s.stream));
        HIPCHK(pbsk::launch_exclusive_scan(s.seg_cnt.as<uint32_t>(), s.nseg, 0xffffffffu, s.seg_off.as<uint32_t>(),
                                           sc + SC_NREC, nullptr, s.scan_tmp.as<uint32_t>(), s.stream));
        HIPCHK(pbsk::launch_resolve_write(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, s.nseg, e->effmin,
                                          e->cfg.max, s.seg_off.as<uint32_t>(), s.recs.as<pbsgpu_record>(),
                                          s.rec_cap, s.stream));
    }
    HIPCHK(hipEventRecord(s.ev[EV_RESOLVE1], s.stream));
    return PBSGPU_OK;
}

// phase 2: longest-first queue + SHA-256 of the first *SC_NREC records
int enqueue_hash(pbsgpu_engine *e, Slot &s) {
    uint32_t *sc = s.scalars.as<uint32_t>();
    const pbsgpu_segment *dsegs = s.segs.as<pbsgpu_segment>();
    HIPCHK(pbsk::launch_order(s.recs.as<pbsgpu_record>(), sc + SC_NREC, e->cfg.max, s.order.as<uint32_t>(),
                              sc + SC_WGLIMIT, e->num_cus, sc + SC_MAXCNT, s.cap, e->sha_slack_pct, s.stream));
    HIPCHK(pbsk::launch_sha256_records(s.dptr, dsegs, s.recs.as<pbsgpu_record>(), sc + SC_NREC, sc + SC_QUEUE,
                                       s.order.as<uint32_t>(), sc + SC_WGLIMIT, e->num_cus, s.stream));
    HIPCHK(hipEventRecord(s.ev[EV_SHA1], s.stream));
    return PBSGPU_OK;
}

// enqueue the whole pipeline for the slot's current (dptr, nbytes, segs) at capacity `cap`
int enqueue_pipeline(pbsgpu_engine *e, Slot &s, uint32_t cap) {
    CHK(enqueue_cut(e, s, cap));
    CHK(enqueue_hash(e, s));
    HIPCHK(hipMemcpyAsync(s.h_scalars.p, s.scalars.p, SC_COUNT * 4, hipMemcpyDeviceToHost, s.stream));
    return PBSGPU_OK;
}

Slot *find_free_slot(pbsgpu_engine *e) {
    for (auto &s : e->slots)
        if (!s.busy) return &s;
    return nullptr;
}

Slot *find_ticket(pbsgpu_engine *e, uint64_t ticket) {
    for (auto &s : e->slots)
        if (s.busy && s.ticket == ticket) return &s;
    return nullptr;
}

// copy the caller's segment table (or the implicit single segment) to the slot
int stage_segments(pbsgpu_engine *e, Slot &s, const pbsgpu_segment *segs, uint32_t nseg, uint64_t nbytes) {
    pbsgpu_segment whole{0, nbytes};
    if (segs == nullptr || nseg == 0) {
        segs = &whole;
        nseg = 1;
    } else {
        CHK(validate_segments(segs, nseg, nbytes));
    }
    CHK(s.h_segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    std::memcpy(s.h_segs.p, segs, (size_t)nseg * sizeof(pbsgpu_segment));
    CHK(s.segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    HIPCHK(hipMemcpyAsync(s.segs.p, s.h_segs.p, (size_t)nseg * sizeof(pbsgpu_segment), hipMemcpyHostToDevice,
                          s.stream));
    s.nseg = nseg;
    s.rec_cap = record_upper_bound(e, s.h_segs.as<pbsgpu_segment>(), nseg);
    return PBSGPU_OK;
}

// host -> device through the engine's two pinned staging buffers (caller memory is not
// referenced after return)
int staged_h2d(pbsgpu_engine *e, void *dst, const void *src, uint64_t nbytes, hipStream_t st) {
    const uint8_t *h = static_cast<const uint8_t *>(src);
    uint8_t *d = static_cast<uint8_t *>(dst);
    int which = 0;
    uint64_t off = 0;
    while (off < nbytes) {
        const size_t n = (size_t)std::min<uint64_t>(kStageBytes, nbytes - off);
        CHK(e->stage[which].ensure(kStageBytes));
        HIPCHK(hipEventSynchronize(e->stage_ev[which]));
        std::memcpy(e->stage[which].p, h + off, n);
        HIPCHK(hipMemcpyAsync(d + off, e->stage[which].p, n, hipMemcpyHostToDevice, st));
        HIPCHK(hipEventRecord(e->stage_ev[which], st));
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

int submit_common(pbsgpu_engine *e, const void *ptr, bool host, uint64_t nbytes, const pbsgpu_segment *segs,
                  uint32_t nseg, uint64_t *ticket) {
    if (!e || !ticket || (!ptr && nbytes)) return PBSGPU_E_INVALID;
    if (!host && nbytes) {
        if (set_device(e) != PBSGPU_OK || !is_device_pointer(ptr)) return PBSGPU_E_INVALID;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    CHK(s->h_scalars.ensure(SC_COUNT * 4));
    CHK(stage_segments(e, *s, segs, nseg, nbytes));
    HIPCHK(hipEventRecord(s->ev[EV_BEGIN], s->stream));
    if (host) {
        CHK(s->data.ensure((size_t)nbytes + 64));
        CHK(staged_h2d(e, s->data.p, ptr, nbytes, s->stream));
        s->dptr = s->data.as<uint8_t>();
    } else {
        s->dptr = static_cast<const uint8_t *>(ptr);
    }
    s->nbytes = nbytes;
    s->host_submit = host;
    s->retries = 0;
    s->synced = false;
    int st = enqueue_pipeline(e, *s, default_cap(e, s->nbytes));
    if (st != PBSGPU_OK) {
        (void)hipStreamSynchronize(s->stream);
        return st;
    }
    s->busy = true;
    s->ticket = e->next_ticket++;
    *ticket = s->ticket;
    return PBSGPU_OK;
}

// wait for a slot; re-run with a larger per-tile capacity if any tile overflowed
int sync_slot(pbsgpu_engine *e, Slot &s) {
    if (s.synced) return PBSGPU_OK;
    for (;;) {
        HIPCHK(hipStreamSynchronize(s.stream));
        const uint32_t *hs = s.h_scalars.as<uint32_t>();
        if (hs[SC_MAXCNT] <= s.cap) {
            s.ncand = hs[SC_NCAND];
            s.nrec = hs[SC_NREC];
            break;
        }
        uint32_t cap = s.cap;
        while (cap < hs[SC_MAXCNT]) cap <<= 1;
        if (cap > pbsk::scan_tile_bytes(s.nbytes)) cap = pbsk::scan_tile_bytes(s.nbytes);
        e->cap_hint = cap;
        e->cap_hint_tile = pbsk::scan_tile_bytes(s.nbytes);
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
    CHK(s.h_scalars.ensure(SC_COUNT * 4));
    uint32_t tcap = default_cap(e, nbytes);
    for (;;) {
        CHK(enqueue_candidates(e, s, dptr, nbytes, tcap));
        HIPCHK(hipMemcpyAsync(s.h_scalars.p, s.scalars.p, SC_COUNT * 4, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(hipStreamSynchronize(s.stream));
        const uint32_t *hs = s.h_scalars.as<uint32_t>();
        if (hs[SC_MAXCNT] <= tcap) break;
        while (tcap < hs[SC_MAXCNT]) tcap <<= 1;
        if (tcap > pbsk::scan_tile_bytes(nbytes)) tcap = pbsk::scan_tile_bytes(nbytes);
    }
    *count = s.h_scalars.as<uint32_t>()[SC_NCAND];
    return PBSGPU_OK;
}

// full pipeline on device-resident bytes, synchronous; records stay in s.recs
int batch_sync(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
               uint32_t nseg, uint64_t *nrec) {
    CHK(s.h_scalars.ensure(SC_COUNT * 4));
    CHK(stage_segments(e, s, segs, nseg, nbytes));
    HIPCHK(hipEventRecord(s.ev[EV_BEGIN], s.stream));
    s.dptr = dptr;
    s.nbytes = nbytes;
    s.host_submit = false;
    s.retries = 0;
    s.synced = false;
    CHK(enqueue_pipeline(e, s, default_cap(e, s.nbytes)));
    CHK(sync_slot(e, s));
    *nrec = s.nrec;
    return PBSGPU_OK;
}

// phase 1 only, synchronous (streaming writer): records WITHOUT digests stay in s.recs
int cut_sync(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
             uint32_t nseg, uint64_t *nrec) {
    CHK(s.h_scalars.ensure(SC_COUNT * 4 + 64));
    CHK(stage_segments(e, s, segs, nseg, nbytes));
    s.dptr = dptr;
    s.nbytes = nbytes;
    s.host_submit = false;
    s.retries = 0;
    s.synced = false;
    uint32_t cap = default_cap(e, nbytes);
    for (;;) {
        CHK(enqueue_cut(e, s, cap));
        HIPCHK(hipMemcpyAsync(s.h_scalars.p, s.scalars.p, SC_COUNT * 4, hipMemcpyDeviceToHost, s.stream));
        HIPCHK(hipStreamSynchronize(s.stream));
        const uint32_t *hs = s.h_scalars.as<uint32_t>();
        if (hs[SC_MAXCNT] <= cap) break;
        while (cap < hs[SC_MAXCNT]) cap <<= 1;
        if (cap > pbsk::scan_tile_bytes(nbytes)) cap = pbsk::scan_tile_bytes(nbytes);
        s.retries++;
    }
    s.nrec = s.h_scalars.as<uint32_t>()[SC_NREC];
    s.ncand = s.h_scalars.as<uint32_t>()[SC_NCAND];
    if (s.nrec > s.rec_cap) return PBSGPU_E_STATE;
    *nrec = s.nrec;
    return PBSGPU_OK;
}

// phase 2 for the first `nhash` records of a slot that went through cut_sync; asynchronous
int hash_async(pbsgpu_engine *e, Slot &s, uint64_t nhash) {
    uint32_t *hn = s.h_scalars.as<uint32_t>() + SC_COUNT;  // pinned scratch word behind the readback area
    *hn = (uint32_t)nhash;
    HIPCHK(hipMemcpyAsync(s.scalars.as<uint32_t>() + SC_NREC, hn, 4, hipMemcpyHostToDevice, s.stream));
    return enqueue_hash(e, s);
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

int pbsgpu_engine_create(int device, const pbsgpu_config *cfg, uint32_t inflight, pbsgpu_engine **out) {
    if (!cfg || !out) return PBSGPU_E_INVALID;
    *out = nullptr;
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
        e->slots.resize(inflight);
        e->sha_slack_pct = inflight > 4 ? 0u : 25u;
        if (const char *sl = getenv("PBSGPU_SHA_SLACK_PCT")) {  // experiments
            const int v = atoi(sl);
            if (v >= 0 && v <= 400) e->sha_slack_pct = (uint32_t)v;
        }
        for (auto &s : e->slots) {
            if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) { st = PBSGPU_E_HIP; break; }
            for (auto &ev : s.ev)
                if (hipEventCreate(&ev) != hipSuccess) { st = PBSGPU_E_HIP; break; }
        }
        for (auto &ev : e->stage_ev)
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) st = PBSGPU_E_HIP;
    } while (0);
    if (st != PBSGPU_OK) {
        pbsgpu_engine_destroy(e);
        return st;
    }
    *out = e;
    return PBSGPU_OK;
}

void pbsgpu_engine_destroy(pbsgpu_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (auto &s : e->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        for (DevBuf *b : {&s.data, &s.tile_cnt, &s.tile_off, &s.tile_slots, &s.dense, &s.scan_tmp, &s.scalars, &s.segs,
                          &s.seg_cnt, &s.seg_off, &s.recs, &s.order})
            b->release();
        s.h_scalars.release();
        s.h_segs.release();
        for (auto &ev : s.ev)
            if (ev) (void)hipEventDestroy(ev);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    for (auto &b : e->stage) b.release();
    for (auto &ev : e->stage_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->d_table_rot) (void)hipFree(e->d_table_rot);
    delete e;
}

int pbsgpu_engine_config(const pbsgpu_engine *e, pbsgpu_config *out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    *out = e->cfg;
    return PBSGPU_OK;
}

int pbsgpu_submit_device(pbsgpu_engine *e, const void *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
                         uint32_t nseg, uint64_t *ticket) {
    return submit_common(e, dptr, false, nbytes, segs, nseg, ticket);
}

int pbsgpu_submit_host(pbsgpu_engine *e, const void *hptr, uint64_t nbytes, const pbsgpu_segment *segs,
                       uint32_t nseg, uint64_t *ticket) {
    return submit_common(e, hptr, true, nbytes, segs, nseg, ticket);
}

int pbsgpu_wait(pbsgpu_engine *e, uint64_t ticket, uint64_t *nrecords) {
    if (!e) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_ticket(e, ticket);
    if (!s) return PBSGPU_E_TICKET;
    CHK(sync_slot(e, *s));
    if (nrecords) *nrecords = s->nrec;
    return PBSGPU_OK;
}

int pbsgpu_ticket_done(pbsgpu_engine *e, uint64_t ticket, int *done) {
    if (!e || !done) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_ticket(e, ticket);
    if (!s) return PBSGPU_E_TICKET;
    if (s->synced) {
        *done = 1;
        return PBSGPU_OK;
    }
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
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_ticket(e, ticket);
    if (!s) return PBSGPU_E_TICKET;
    int st = sync_slot(e, *s);
    if (st != PBSGPU_OK) {
        s->busy = false;
        return st;
    }
    if (nrecords) *nrecords = s->nrec;
    if (s->nrec > cap || (!out && s->nrec)) return PBSGPU_E_CAPACITY;
    static const bool trace = getenv("PBSGPU_TRACE") != nullptr;  // ingest log line, like tapeio's MB/s progress
    if (trace) {
        float scan = 0, res = 0, sha = 0;
        (void)hipEventElapsedTime(&scan, s->ev[EV_SCAN0], s->ev[EV_SCAN1]);
        (void)hipEventElapsedTime(&res, s->ev[EV_SCAN1], s->ev[EV_RESOLVE1]);
        (void)hipEventElapsedTime(&sha, s->ev[EV_RESOLVE1], s->ev[EV_SHA1]);
        (void)hipGetLastError();
        fprintf(stderr, "[pbsgpu] ticket %llu: %.2f MiB, %llu candidates, %llu chunks, scan %.3f ms, resolve %.3f ms, "
                        "sha256 %.3f ms, retries %u\n",
                (unsigned long long)s->ticket, s->nbytes / 1048576.0, (unsigned long long)s->ncand,
                (unsigned long long)s->nrec, scan, res, sha, s->retries);
    }
    if (s->nrec) {
        HIPCHK(hipMemcpyAsync(out, s->recs.p, (size_t)s->nrec * sizeof(pbsgpu_record), hipMemcpyDeviceToHost,
                              s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    s->busy = false;
    return PBSGPU_OK;
}

int pbsgpu_ticket_timing(pbsgpu_engine *e, uint64_t ticket, pbsgpu_timing *out) {
    if (!e || !out) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_ticket(e, ticket);
    if (!s) return PBSGPU_E_TICKET;
    CHK(sync_slot(e, *s));
    std::memset(out, 0, sizeof(*out));
    float ms = 0;
    if (s->retries == 0 && hipEventElapsedTime(&ms, s->ev[EV_BEGIN], s->ev[EV_SCAN0]) == hipSuccess) out->h2d_ms = ms;
    if (hipEventElapsedTime(&ms, s->ev[EV_SCAN0], s->ev[EV_SCAN1]) == hipSuccess) out->scan_ms = ms;
    if (hipEventElapsedTime(&ms, s->ev[EV_SCAN1], s->ev[EV_RESOLVE1]) == hipSuccess) out->resolve_ms = ms;
    if (hipEventElapsedTime(&ms, s->ev[EV_RESOLVE1], s->ev[EV_SHA1]) == hipSuccess) out->sha_ms = ms;
    if (hipEventElapsedTime(&ms, s->ev[EV_SCAN0], s->ev[EV_SHA1]) == hipSuccess) out->total_ms = ms;
    (void)hipGetLastError();
    out->ncandidates = s->ncand;
    out->nrecords = s->nrec;
    out->retries = s->retries;
    return PBSGPU_OK;
}

int pbsgpu_candidates_device(pbsgpu_engine *e, const void *dptr, uint64_t nbytes, uint64_t *out, uint64_t cap,
                             uint64_t *n) {
    if (!e || !n || (!dptr && nbytes)) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    if (nbytes && !is_device_pointer(dptr)) return PBSGPU_E_INVALID;
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
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
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    pbsgpu_segment whole{0, stream_len};
    CHK(s->h_scalars.ensure(SC_COUNT * 4 + 64));
    CHK(stage_segments(e, *s, &whole, 1, stream_len));
    CHK(s->recs.ensure((size_t)s->rec_cap * sizeof(pbsgpu_record) + 64));
    CHK(s->dense.ensure((size_t)ncand * 8 + 64));
    CHK(s->scalars.ensure(SC_COUNT * 4));
    HIPCHK(hipMemsetAsync(s->scalars.p, 0, SC_COUNT * 4, s->stream));
    if (ncand) CHK(staged_h2d(e, s->dense.p, cands, ncand * 8, s->stream));
    uint32_t *hn = s->h_scalars.as<uint32_t>() + SC_COUNT;
    *hn = (uint32_t)ncand;
    uint32_t *sc = s->scalars.as<uint32_t>();
    HIPCHK(hipMemcpyAsync(sc + SC_NCAND, hn, 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(pbsk::launch_resolve_single(s->dense.as<uint64_t>(), sc + SC_NCAND, s->segs.as<pbsgpu_segment>(), e->effmin,
                                       e->cfg.max, sc + SC_ZERO, sc + SC_NREC, s->recs.as<pbsgpu_record>(), s->rec_cap,
                                       s->stream));
    HIPCHK(hipMemcpyAsync(s->h_scalars.p, s->scalars.p, SC_COUNT * 4, hipMemcpyDeviceToHost, s->stream));
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

static int sha256_many(pbsgpu_engine *e, const void *ptr, bool host, uint64_t nbytes, const pbsgpu_segment *segs,
                       uint32_t nseg, uint8_t *digests) {
    if (!e || (!ptr && nbytes) || (nseg && (!segs || !digests))) return PBSGPU_E_INVALID;
    if (nseg == 0) return PBSGPU_OK;
    for (uint32_t i = 0; i < nseg; ++i)
        if (segs[i].length > nbytes || segs[i].offset > nbytes - segs[i].length) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    CHK(s->h_segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    std::memcpy(s->h_segs.p, segs, (size_t)nseg * sizeof(pbsgpu_segment));
    CHK(s->segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    HIPCHK(hipMemcpyAsync(s->segs.p, s->h_segs.p, (size_t)nseg * sizeof(pbsgpu_segment), hipMemcpyHostToDevice,
                          s->stream));
    const uint8_t *d = static_cast<const uint8_t *>(ptr);
    if (host) {
        CHK(s->data.ensure((size_t)nbytes + 64));
        CHK(staged_h2d(e, s->data.p, ptr, nbytes, s->stream));
        d = s->data.as<uint8_t>();
    }
    CHK(s->recs.ensure((size_t)nseg * 32));
    CHK(s->scalars.ensure(SC_COUNT * 4));
    HIPCHK(hipMemsetAsync(s->scalars.p, 0, SC_COUNT * 4, s->stream));
    HIPCHK(pbsk::launch_sha256_segments(d, s->segs.as<pbsgpu_segment>(), nseg, s->recs.as<uint8_t>(),
                                        s->scalars.as<uint32_t>() + SC_QUEUE, e->num_cus, s->stream));
    HIPCHK(hipMemcpyAsync(digests, s->recs.p, (size_t)nseg * 32, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PBSGPU_OK;
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
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    CHK(s->h_segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    std::memcpy(s->h_segs.p, segs, (size_t)nseg * sizeof(pbsgpu_segment));
    CHK(s->segs.ensure((size_t)nseg * sizeof(pbsgpu_segment)));
    HIPCHK(hipMemcpyAsync(s->segs.p, s->h_segs.p, (size_t)nseg * sizeof(pbsgpu_segment), hipMemcpyHostToDevice,
                          s->stream));
    const uint8_t *d = static_cast<const uint8_t *>(ptr);
    if (host) {
        CHK(s->data.ensure((size_t)nbytes + 64));
        CHK(staged_h2d(e, s->data.p, ptr, nbytes, s->stream));
        d = s->data.as<uint8_t>();
    }
    CHK(s->recs.ensure((size_t)nseg * 8 + 64));
    CHK(s->scalars.ensure(SC_COUNT * 4));
    HIPCHK(hipMemsetAsync(s->scalars.p, 0, SC_COUNT * 4, s->stream));
    HIPCHK(pbsk::launch_xxh3(d, s->segs.as<pbsgpu_segment>(), nseg, s->recs.as<uint64_t>(),
                             s->scalars.as<uint32_t>() + SC_QUEUE, e->num_cus, s->stream));
    HIPCHK(hipMemcpyAsync(out, s->recs.p, (size_t)nseg * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return PBSGPU_OK;
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
    if (!e || (!dptr && nbytes) || ((uintptr_t)dptr & 7u) || (stream_off & 7u) || kind > 3) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    hipStream_t st = e->slots[0].stream;
    HIPCHK(pbsk::launch_fill(dptr, stream_off, nbytes, seed, kind, st));
    HIPCHK(hipStreamSynchronize(st));
    return PBSGPU_OK;
}

int pbsgpu_device_alloc(pbsgpu_engine *e, uint64_t nbytes, void **dptr) {
    if (!e || !dptr) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
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
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    if (dptr) HIPCHK(hipFree(dptr));
    return PBSGPU_OK;
}

int pbsgpu_memcpy_h2d(pbsgpu_engine *e, void *dptr, const void *hptr, uint64_t nbytes) {
    if (!e || ((!dptr || !hptr) && nbytes)) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    if (nbytes) HIPCHK(hipMemcpy(dptr, hptr, nbytes, hipMemcpyHostToDevice));
    return PBSGPU_OK;
}

int pbsgpu_memcpy_d2h(pbsgpu_engine *e, void *hptr, const void *dptr, uint64_t nbytes) {
    if (!e || ((!dptr || !hptr) && nbytes)) return PBSGPU_E_INVALID;
    std::lock_guard<std::mutex> lk(e->mu);
    CHK(set_device(e));
    if (nbytes) HIPCHK(hipMemcpy(hptr, dptr, nbytes, hipMemcpyDeviceToHost));
    return PBSGPU_OK;
}

}  // extern "C"
