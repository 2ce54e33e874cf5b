// libpbsgpu host side: C-ABI entry points (include/pbsgpu.h) over the gfx950 kernels.
//
// Stands where github.com/pbs-plus/pxar's backupproxy session runs its chunk loop for
// the writers the reference drives (internal/pxarmount/commit_orchestrate.go:137-177,
// internal/tapeio/converter.go:386-439). No CPU fallback exists in this file: every data
// path launches the HIP kernels; without a device the engine cannot be created.
#include "engine_internal.h"

namespace pbse {
std::atomic<int> g_last_hip_error{0};
}
using namespace pbse;

namespace {

const uint32_t kDefaultTable[256] = {
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

}  // namespace

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
    HIPCHK(pbsk::launch_resolve_count(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, s.nseg, e->effmin, e->cfg.max,
                                      s.seg_cnt.as<uint32_t>(), s.stream));
    HIPCHK(pbsk::launch_exclusive_scan(s.seg_cnt.as<uint32_t>(), s.nseg, 0xffffffffu, s.seg_off.as<uint32_t>(),
                                       sc + SC_NREC, nullptr, s.scan_tmp.as<uint32_t>(), s.stream));
    HIPCHK(pbsk::launch_resolve_write(s.dense.as<uint64_t>(), sc + SC_NCAND, dsegs, s.nseg, e->effmin, e->cfg.max,
                                      s.seg_off.as<uint32_t>(), s.recs.as<pbsgpu_record>(), s.rec_cap, s.stream));
    HIPCHK(hipEventRecord(s.ev[EV_RESOLVE1], s.stream));
    return PBSGPU_OK;
}

// phase 2: longest-first queue + SHA-256 of the first *SC_NREC records
int enqueue_hash(pbsgpu_engine *e, Slot &s) {
    uint32_t *sc = s.scalars.as<uint32_t>();
    const pbsgpu_segment *dsegs = s.segs.as<pbsgpu_segment>();
    HIPCHK(pbsk::launch_order(s.recs.as<pbsgpu_record>(), sc + SC_NREC, e->cfg.max, s.order.as<uint32_t>(),
                              sc + SC_WGLIMIT, e->num_cus, sc + SC_MAXCNT, s.cap, s.stream));
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

int submit_common(pbsgpu_engine *e, const void *ptr, bool host, uint64_t nbytes, const pbsgpu_segment *segs,
                  uint32_t nseg, uint64_t *ticket) {
    if (!e || !ticket || (!ptr && nbytes)) return PBSGPU_E_INVALID;
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

const char *pbsgpu_strerror(int status) {
    switch (status) {
    case PBSGPU_OK: return "ok";
    case PBSGPU_E_INVALID: return "invalid argument";
    case PBSGPU_E_NO_DEVICE: return "no usable HIP device";
    case PBSGPU_E_HIP: return "HIP runtime error";
    case PBSGPU_E_NOMEM: return "out of memory";
    case PBSGPU_E_CAPACITY: return "output buffer too small";
    case PBSGPU_E_BUSY: return "all in-flight slots busy";
    case PBSGPU_E_TICKET: return "unknown ticket";
    case PBSGPU_E_DENSITY: return "candidate density exceeds capacity";
    case PBSGPU_E_STATE: return "invalid state";
    default: return "unknown status";
    }
}

int pbsgpu_abi_version(void) { return PBSGPU_ABI_VERSION; }
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

const uint32_t *pbsgpu_default_table(void) { return kDefaultTable; }

int pbsgpu_config_init(uint64_t avg, const uint32_t *table, pbsgpu_config *out) {
    if (!out) return PBSGPU_E_INVALID;
    if (avg < 256 || avg > (1ull << 28) || (avg & (avg - 1)) != 0) return PBSGPU_E_INVALID;
    out->avg = (uint32_t)avg;
    out->min = (uint32_t)(avg >> 2);
    out->max = (uint32_t)(avg << 2);
    out->window = pbsk::kWindow;
    out->mask = (uint32_t)(avg * 2 - 1);
    out->break_min = out->mask - 2;
    std::memcpy(out->table, table ? table : kDefaultTable, sizeof(out->table));
    return PBSGPU_OK;
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
    if (inflight > 8) return PBSGPU_E_INVALID;

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
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        e->num_cus = prop.multiProcessorCount;

    uint32_t rot[256];
    const uint32_t r = (32 - bits) & 31;
    for (int i = 0; i < 256; ++i) rot[i] = r ? ((cfg->table[i] << r) | (cfg->table[i] >> (32 - r))) : cfg->table[i];
    int st = PBSGPU_OK;
    do {
        if (hipMalloc(reinterpret_cast<void **>(&e->d_table_rot), sizeof(rot)) != hipSuccess) { st = PBSGPU_E_NOMEM; break; }
        if (hipMemcpy(e->d_table_rot, rot, sizeof(rot), hipMemcpyHostToDevice) != hipSuccess) { st = PBSGPU_E_HIP; break; }
        e->slots.resize(inflight);
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
    Slot *s = find_free_slot(e);
    if (!s) return PBSGPU_E_BUSY;
    uint64_t cnt = 0;
    CHK(candidates_sync(e, *s, static_cast<const uint8_t *>(dptr), nbytes, &cnt));
    *n = cnt;
    if (cnt > cap || (!out && cnt)) return PBSGPU_E_CAPACITY;
    if (cnt) HIPCHK(hipMemcpy(out, s->dense.p, (size_t)cnt * 8, hipMemcpyDeviceToHost));
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
