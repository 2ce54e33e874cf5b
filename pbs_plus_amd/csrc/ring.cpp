// libpbsgpu host side, part 3: the PAGE RING — many payload streams through one device arena with page-granular
// memory release and a persistent cross-stream SHA-256 service. Since round 4 it is THE engine behind every streaming
// entry point: pbsgpu_ring_* drives it directly (bytes already in device memory), pbsgpu_stream_* (stream.cpp) is a client
// that feeds host bytes into reserved pages.
//
// Why it exists. SHA-256 is serial inside a chunk (a 16 MiB chunk = 262 144 dependent compressions ~ 0.43 s on one
// lane), so a batch submitted with pbsgpu_submit_device keeps ALL its bytes resident until its longest chunk is done:
// throughput <= resident bytes / 0.45 s whatever the chip could hash. The ring removes the batch as the unit of
// residency. It stands where the chunk loop behind WriteEntryReader runs for every archive of a multi-archive ingest
// (internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836):
//   * device memory is an arena of fixed PAGES (>= max chunk, a whole number of scan tiles). A stream is a sequence
//     of pages in logical order; physically they lie wherever a page was free;
//   * newly filled pages of all streams are cut in ROUNDS (ring_kernels.inc): scan of the new pages only, multi-stream
//     resolve continuing from each stream's open chunk (device-resident state: no host round trip between rounds);
//   * every cut chunk goes into ONE device-resident FIFO that the SHA-256 SERVICE — a persistent kernel on a fixed set
//     of CUs — drains: a lane takes the next chunk the moment it finishes one, across rounds and streams;
//   * each page counts the chunks that still have to be read from it (plus a hold while the stream's open chunk
//     reaches into it). The lane that loads a chunk's last block drops the reference; a page that reaches zero is
//     reported to the host through mapped pinned memory and can take new bytes at once — residency per page is the
//     hash time of the longest chunk that touches THAT page, not of the longest chunk of a 64 GiB batch;
//   * digests and record fields land in host-visible record cells; pbsgpu_ring_poll hands them out per stream, in order.
// One thread drives a ring (like one goroutine owns a writer, internal/tapeio/converter.go:672-680).
//
// No byte content fails a stream: a scan tile that finds more candidates than it has slots (periodic / crafted data) is
// resolved exactly by an on-demand re-scan inside the control kernel (DenseTiles, kernels.h) — like the reference's writer,
// which has no content-dependent error (internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836).
// A host that stops calling for longer than the idle timeout finds the service gone and the ring healthy — the next round
// starts it again.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "ring_internal.h"

using namespace pbse;

namespace {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

uint32_t pow2_at_least(uint64_t v) {
    uint32_t p = 1;
    while (p < v && p < (1u << 31)) p <<= 1;
    return p;
}

volatile uint32_t *hb_words(const pbsgpu_ring *r) { return r->heartbeat.as<volatile uint32_t>(); }

// the service launched last is known to have ended: statistics, the service count, parked frees
void ring_service_ended(pbsgpu_ring *r) {
    if (r->svc == SvcState::Stopped) return;
    r->svc = SvcState::Stopped;
    float ms = 0;
    if (hipEventElapsedTime(&ms, r->ev_svc0, r->ev_svc1) == hipSuccess) {
        r->st.service_ms_last = ms;
        r->st.service_ms_total += ms;
    } else {
        (void)hipGetLastError();
    }
    r->st.service_bytes_last = r->st.bytes_enqueued - r->svc_bytes0;
    volatile uint32_t *hb = hb_words(r);
    hb[pbsk::kHbIntent] = 0;
    hb[pbsk::kHbCommitted] = 0;
    // head := tail, stop := 0 — NOW, while nothing is published behind the service's back: its lanes may have left holding
    // claims beyond the tail, and rounds enqueued from here on (even before the next service starts: a lone stream's rounds
    // are cut ahead at full chip width, ring_enqueue_round) must find the queue ready to be served from exactly this point
    if (r->cs) {
        (void)pbsk::launch_ring_reset(r->ctl.as<pbsk::RingCtl>(), r->cs);
        (void)hipEventRecord(r->ev_reset, r->cs);
    }
    r->parked_for_flush = false;
    service_ended(r->eng->device);  // (the device's last service: memory parked by dev_free / host_free is freed now)
}

// Did the service stop on its own (idle timeout, kernels.hip)? Then wait the few microseconds until the kernel is gone.
// With `gate` (a round is about to be enqueued) also settle an ANNOUNCED stop: the caller has bumped heartbeat and round
// count before coming here, so the service either withdraws or commits within a PCIe round trip.
int ring_service_check(pbsgpu_ring *r, bool gate) {
    if (r->svc != SvcState::Running) return PBSGPU_OK;
    volatile uint32_t *hb = hb_words(r);
    if (gate && hb[pbsk::kHbIntent] != 0 && hb[pbsk::kHbCommitted] == 0) {
        const double t0 = now_ms();
        while (hb[pbsk::kHbIntent] != 0 && hb[pbsk::kHbCommitted] == 0) {
            if (now_ms() - t0 > 5000.0) return r->error = PBSGPU_E_STATE;  // neither withdrawn nor committed: the device is gone
            std::this_thread::yield();
        }
    }
    if (hb[pbsk::kHbCommitted] != 0) {
        std::atomic_thread_fence(std::memory_order_acquire);
        HIPCHK(hipStreamSynchronize(r->ss));
        ring_service_ended(r);
    }
    return PBSGPU_OK;
}

// every ring call: tell the service the host is alive
void ring_heartbeat(pbsgpu_ring *r) {
    volatile uint32_t *hb = hb_words(r);
    hb[pbsk::kHbBeat] = hb[pbsk::kHbBeat] + 1u;
}

// pages the service has handed back since the last call
void ring_reap_free(pbsgpu_ring *r) {
    volatile unsigned long long *f = r->free_fifo.as<volatile unsigned long long>();
    for (;;) {
        const unsigned long long e = f[r->free_read & (r->nfree - 1)];
        if ((uint32_t)(e >> 32) != r->free_read + 1u) break;
        r->free_pages.push_back((uint32_t)e);
        r->free_read++;
        r->st.pages_recycled++;
    }
}

// round results in order: record cells to their streams, finished streams, input tables reusable
void ring_reap_rounds(pbsgpu_ring *r) {
    for (auto &ri : r->rounds) {
        if (ri.reaped) continue;
        volatile pbsk::RingRoundStatus *hs = r->in_status(ri.input);
        if (hs->seq != ri.seq) break;  // rounds complete in order
        std::atomic_thread_fence(std::memory_order_acquire);
        if (hs->error) {
            r->error = PBSGPU_E_STATE;  // record / cell capacity of a round exceeded: the ring's own bound was wrong
        } else {
            const uint32_t n = hs->nrec;
            const uint8_t *cells = r->cells.as<uint8_t>();
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t c = (uint32_t)((ri.cell_base + i) & (r->ncells - 1));
                const uint32_t *cw = reinterpret_cast<const uint32_t *>(cells + (size_t)c * 64);
                if (cw[11] == 0) continue;  // the round's open chunk: no record
                const uint32_t slot = cw[10];
                if (slot >= r->slots.size()) continue;
                r->slots[slot].cells.push_back(CellRef{c, ri.seq});
                ri.live_cells++;
                r->st.chunks++;
                r->obs_bytes += (double)cw[11];
                if (r->long_bytes && cw[11] >= r->long_bytes) r->obs_long_bytes += (double)cw[11];
            }
            r->st.candidates += hs->ncand;
            r->pub_positions += (uint32_t)(hs->tail - r->tail_seen);
            r->pub_bytes += ri.new_bytes;
            r->tail_seen = hs->tail;
            for (int i = 0; i < 6; ++i) r->probe_seen[i] = hs->probe[i];
            r->probe_seen_valid = true;
            {   // the control kernel's phase times, summed per round size class (pbsgpu_ring_debug): small (< 100 pages) / large rounds
                const int cls = ri.new_bytes < 100ull * r->page_bytes ? 0 : 1;
                for (int i = 0; i < 5; ++i) r->ctl_phase_ticks[cls][i] += hs->phase_ticks[i];
                r->ctl_phase_rounds[cls]++;
            }
        }
        for (uint32_t s : ri.finals) r->slots[s].final_done = true;
        r->inflight_bytes -= ri.new_bytes;
        ri.reaped = true;
        r->input_busy[ri.input] = false;
        r->st.rounds_done++;
    }
    while (!r->rounds.empty() && r->rounds.front().reaped && r->rounds.front().live_cells == 0) r->rounds.pop_front();
}

// the pair service on `ss` and, when the ring has one, the express service on `xs`; ev_svc1 = BOTH have ended
int ring_launch_services(pbsgpu_ring *r) {
    HIPCHK(hipStreamWaitEvent(r->ss, r->ev_reset, 0));  // (never recorded before the first launch: no wait)
    HIPCHK(hipEventRecord(r->ev_svc0, r->ss));
    HIPCHK(pbsk::launch_ring_service(r->source(), r->sha_cus, r->ss, r->dense_service));
    if (r->xp_cus) {
        HIPCHK(hipStreamWaitEvent(r->xs, r->ev_reset, 0));
        HIPCHK(pbsk::launch_ring_service_xp(r->source(), r->xp_cus, r->xs));
        HIPCHK(hipEventRecord(r->ev_xsvc1, r->xs));
        HIPCHK(hipStreamWaitEvent(r->ss, r->ev_xsvc1, 0));
    }
    if (r->lanes_cus) {
        HIPCHK(hipStreamWaitEvent(r->ls, r->ev_reset, 0));
        HIPCHK(pbsk::launch_ring_service_lanes(r->source(), r->lanes_cus, r->ls, r->dense_lanes));
        HIPCHK(hipEventRecord(r->ev_lsvc1, r->ls));
        HIPCHK(hipStreamWaitEvent(r->ss, r->ev_lsvc1, 0));
    }
    HIPCHK(hipEventRecord(r->ev_svc1, r->ss));
    return PBSGPU_OK;
}

// The express form hashes a chunk 1.37x sooner at 0.70 of the pair form's throughput per CU (2.99 vs 4.28 GiB/s per CU,
// DESIGN.md 5.6). For a share s of the bytes in long chunks both services are equally busy when the express service holds
//   (s / 2.99) / (s / 2.99 + (1 - s) / 4.28)   of the services' CUs:
// 7 % = 16 CUs for random data (s = 0.05: the default), 59 % = 112 CUs for a corpus with half its bytes in zero runs or
// periodic files (BASELINE configs[2]: measured 429 GiB/s with 16 express CUs, 441 / 475 / 490 with 48 / 80 / 112 — every
// page of such a file is held for the chain of a 16 MiB chunk, 0.49 s on a pair lane, 0.34 s on an express pair, and the
// arena binds). Called when a service starts: the observation window is what was published since the last decision
// (at least 8 GiB), halved afterwards so that the split follows the data with some memory.
void ring_adapt_split(pbsgpu_ring *r) {
    if (!r->split_auto || r->xs == nullptr || r->long_bytes == 0 || r->obs_bytes < 8.0 * 1073741824.0) return;
    const double s = std::min(1.0, r->obs_long_bytes / r->obs_bytes);
    const double fx = (s / 2.99) / (s / 2.99 + (1.0 - s) / 4.28);
    // (x 0.9: the pair lanes help out with long chunks whenever every express pair is busy, the express pairs never take a
    // short chunk — too few express CUs cost little, too many leave the pair service short: configs[2] measured 490 GiB/s at
    // 112 express CUs, 469 at 120, 445 at 128)
    int xp = (int)(0.9 * fx * (double)r->svc_cus / 8.0 + 0.5) * 8;  // whole XCD rows: other counts leave the XCDs unevenly loaded
    xp = std::max(16, std::min(xp, (int)r->svc_cus - 32));
    r->obs_bytes *= 0.5;
    r->obs_long_bytes *= 0.5;
    if (std::abs(xp - (int)r->xp_cus) < 16) return;  // hysteresis
    r->xp_cus = (uint32_t)xp;
    r->sha_cus = r->svc_cus - r->xp_cus - r->lanes_cus;
    r->st.sha_cus = r->sha_cus;
    if (r->backlog_auto) r->backlog_limit = (uint64_t)(r->sha_cus + r->lanes_cus) << 27;  // (the gate is sized by the CUs that serve the main and the short queue)
}

// (`force`: quiesce / destroy — the caller is about to wait for the queue to drain, a service must run now)
int ring_start_service(pbsgpu_ring *r, bool force = false) {
    if (r->svc == SvcState::Running) return PBSGPU_OK;
    if (r->svc == SvcState::Stopping && r->parked_for_flush) {
        // parked because memory waits in the device's graveyard for the services to END: a new launch queued right behind
        // the old one would keep the device's service count above zero for ever — wait for the old one (at most the chain
        // of the chunks its lanes hold, ~0.5 s; a rare memory-pressure event), let the frees happen, then start afresh
        HIPCHK(hipStreamSynchronize(r->ss));
        ring_service_ended(r);
    }
    const uint32_t park_gen = service_park_generation(r->eng->device);
    if (r->svc == SvcState::Stopped && !force && park_gen != 0 && park_gen != r->park_grace_gen) {
        // The park request is STILL pending: other rings of the device (the stream writer's engine ring beside a bulk ring,
        // two engines) have not let go of their services yet. Starting ours now would put the device's service count back
        // before theirs has dropped — with several busy rings the count then never reaches zero, nothing is ever freed and
        // every ring pays its restart stall for nothing (round 5's cap only worked with ONE ring per device). Stay stopped:
        // rounds are still cut (like the cut-ahead of a lone stream), the next pump asks again, and the last ring to end
        // flushes. Bounded, ONCE per request: a ring that is never called again (its service stops by its idle timeout, but
        // nobody observes the end — an engine a host leaked) keeps the request pending for good, and must cost every other
        // ring one grace period, not one per service start (a stream writer's ring restarts its service after every idle
        // 2 ms: the first version of this wait made a whole test suite crawl behind one leaked engine).
        const double t = now_ms();
        if (r->park_wait_t0 == 0) r->park_wait_t0 = t;
        if (t - r->park_wait_t0 < 250.0) return PBSGPU_OK;
        r->park_grace_gen = park_gen;
    }
    r->park_wait_t0 = 0;
    if (r->svc == SvcState::Stopping) {
        // parked, its end not yet observed: the new service goes behind the old one's END and a reset (its lanes may hold
        // claims beyond the tail that the reset hands out again) — all on the device, nobody waits here
        HIPCHK(hipStreamWaitEvent(r->cs, r->ev_svc1, 0));
        HIPCHK(pbsk::launch_ring_reset(r->ctl.as<pbsk::RingCtl>(), r->cs));
        HIPCHK(hipEventRecord(r->ev_reset, r->cs));
    } else {
        service_started(r->eng->device);  // (Stopped: the queue was reset when the last service ended)
    }
    r->svc = SvcState::Running;  // (from here on an error leaves a service count behind that quiesce / destroy settle)
    ring_adapt_split(r);
    r->defer_t0 = 0;
    hb_words(r)[pbsk::kHbClaim] = r->tail_seen;
    CHK(ring_launch_services(r));
    r->svc_t0 = now_ms();
    r->svc_bytes0 = r->st.bytes_enqueued - r->deferred_bytes;  // (bytes cut ahead of this launch are its work too)
    r->deferred_bytes = 0;
    r->st.service_launches++;
    return PBSGPU_OK;
}

// build + enqueue one round from the committed pages; *did = false when there is nothing to do or no room
int ring_enqueue_round(pbsgpu_ring *r, bool *did) {
    *did = false;
    pbsgpu_engine *e = r->eng;
    if (r->error != PBSGPU_OK) return r->error;
    bool any = false, any_final = false;
    size_t ready = 0;
    for (auto &s : r->slots) {
        if (!s.open) continue;
        any |= !s.ready.empty() || s.zero_final;
        any_final |= s.zero_final || (!s.ready.empty() && s.ready.back().final);
        ready += s.ready.size();
    }
    if (!any) return PBSGPU_OK;
    // A round costs a few dependent launches whatever it holds: while earlier rounds keep the device busy, wait until
    // a quarter of a full round has gathered (pages come back from the SHA service one by one). A stream's end and an
    // idle device go at once.
    {
        size_t inflight = 0;
        for (auto &ri : r->rounds) inflight += ri.reaped ? 0 : 1;
        if (inflight > 0 && !any_final && ready < r->min_round_pages) return PBSGPU_OK;
        // ... and never queue rounds deep: a page that waits behind several queued rounds is resident without being
        // worked on (measured: 16 rounds deep = ~80 ms of extra residency per page, a quarter of the whole)
        if (inflight >= r->max_inflight) return PBSGPU_OK;
    }
    int in = -1;
    for (uint32_t i = 0; i < kRingInputs; ++i)
        if (!r->input_busy[i]) { in = (int)i; break; }
    if (in < 0) return PBSGPU_OK;
    // record cells: a round takes a contiguous (modulo the ring) range; wait while older rounds still own what the
    // largest possible round would overwrite (only a caller that never polls gets here)
    {
        const uint64_t oldest = r->rounds.empty() ? r->cell_cursor : r->rounds.front().cell_base;
        if (r->cell_cursor + r->rec_cap - oldest > r->ncells) return PBSGPU_OK;
    }
    pbsk::RingPage *pg = r->in_pages((uint32_t)in);
    pbsk::RingSeg *sg = r->in_segs((uint32_t)in);
    uint32_t *recbase = r->in_recbase((uint32_t)in);
    uint32_t *suggidx = r->in_suggidx((uint32_t)in);
    uint32_t np = 0, ns = 0;
    uint64_t cells_needed = 0, new_bytes = 0;
    RoundInfo ri;
    std::vector<hipEvent_t> deps;
    std::vector<uint64_t> sugg;
    const uint32_t minsz = std::min(e->effmin, e->cfg.min);
    const uint64_t feed = e->sugg_feed.load(std::memory_order_relaxed);
    const uint64_t look = feed > 1 ? (uint64_t)e->cfg.max : 0;  // reader-buffer rule: boundaries just beyond the bytes matter too
    // While the services run the cut side has a quarter of the chip, and there a FULL round costs it 1.5x more per byte than a
    // half one (the refill of round n + 1 then runs beside the scan of round n on the same CUs, and nothing of a 4.3 GB round
    // is still in the last-level cache when the scan comes to it): measured with 184 + 8 service CUs, where every round was
    // full, 680 GiB/s of feed phase at 256 pages per round, 763 at 128, 780 at 32 (profiles/r06_round_size_bistability.log).
    // A ring that falls behind once grows its rounds, which makes it fall behind further: round 5's "unexplained" 184 + 8
    // cliff and its few-large-rounds regime. So a round takes at most HALF the configured pages while a service runs; the
    // whole-chip cut-ahead of an idle ring (no service yet: 64 GiB in ~20 ms) keeps full rounds.
    // (one stream alone is cut-bound on the cut side's quarter of the chip once its service has started, with most lanes
    // idle: there the full round is the faster one — one file alone 407 vs 395 ms — and nothing can pile up behind it)
    uint32_t streams_waiting = 0;
    for (auto &s : r->slots) streams_waiting += (s.open && !s.ready.empty()) ? 1u : 0u;
    const uint32_t round_cap = (r->svc == SvcState::Stopped || streams_waiting < 2)
                                   ? r->round_pages
                                   : std::max(std::min(r->round_pages, r->min_round_pages), r->round_pages / 2);
    for (uint32_t si = 0; si < r->slots.size() && np < round_cap; ++si) {
        StreamSlot &s = r->slots[si];
        if (!s.open || (s.ready.empty() && !s.zero_final)) continue;
        pbsk::RingSeg g{};
        g.slot = si;
        g.first_page = np;
        g.reset = s.fresh ? 1u : 0u;
        g.origin = s.origin;
        const uint64_t end_old = s.bytes_enqueued;
        uint64_t end = end_old;
        uint32_t take = 0;
        while (!s.ready.empty() && take < kPagesPerStreamRound && np < round_cap) {
            const PageReq &q = s.ready.front();
            pbsk::RingPage p{};
            p.phys = q.phys;
            p.phys_off = (uint64_t)q.phys * r->stride + 128u;
            p.logical = ((uint64_t)si << pbsk::kRingOffBits) | (q.k * r->page_bytes);
            p.valid = q.valid;
            p.slot = si;
            p.seg = ns;
            p.do_fill = q.do_fill ? 1u : 0u;
            p.fill_seed = q.seed;
            p.fill_off = q.fill_off;
            p.fill_kind = q.kind;
            p.fill_tab = q.tab;
            p.fill_ntab = q.ntab;
            p.prev_phys = (q.k > 0 && s.last_phys >= 0) ? (uint32_t)s.last_phys : 0xffffffffu;
            s.last_phys = (int64_t)q.phys;
            pg[np++] = p;
            if (q.dep) deps.push_back(q.dep);
            end += q.valid;
            new_bytes += q.valid;
            if (q.final) g.final = 1;
            s.ready.pop_front();
            ++take;
        }
        if (s.zero_final && s.ready.empty()) {
            g.final = 1;
            s.zero_final = false;
        }
        g.npages = take;
        g.new_end = end;
        s.bytes_enqueued = end;
        s.fresh = false;
        if (g.final) {
            s.final_enqueued = true;
            ri.finals.push_back(si);
        }
        // suggested boundaries that can still matter: behind the open chunk's start (>= end_old - max, the chunk is
        // shorter than max), up to the new end (+ one max chunk under the reader-buffer rule)
        suggidx[ns] = (uint32_t)sugg.size();
        {
            const uint64_t lower = end_old > e->cfg.max ? end_old - e->cfg.max : 0;
            while (!s.sugg.empty() && s.sugg.front() < lower) s.sugg.pop_front();
            for (uint64_t b : s.sugg) {
                if (b > end + look) break;
                sugg.push_back(b);
            }
        }
        recbase[ns] = (uint32_t)cells_needed;
        cells_needed += ((uint64_t)take * r->page_bytes + e->cfg.max) / minsz + 2;
        sg[ns++] = g;
    }
    if (ns == 0) return PBSGPU_OK;
    suggidx[ns] = (uint32_t)sugg.size();
    if (cells_needed > r->rec_cap) cells_needed = r->rec_cap;  // (the bound above is never larger: rec_cap is the same formula for a full round)
    recbase[ns] = (uint32_t)cells_needed;
    // From here on the popped pages and the streams' new lengths exist only in this round: any failure is the ring's (sticky).
    auto fail = [&](int st) {
        r->error = st;
        for (hipEvent_t ev : deps) ring_event_put(r, ev);
        return st;
    };
    const uint64_t *sugg_dev = nullptr;
    if (!sugg.empty()) {
        PinnedBuf &sb = r->sugg_in[in];
        if (sb.ensure(std::max<size_t>(sugg.size() * 8, 4096)) != PBSGPU_OK) return fail(PBSGPU_E_NOMEM);
        std::memcpy(sb.p, sugg.data(), sugg.size() * 8);
        sugg_dev = sb.as<uint64_t>();
    }
    ri.cell_base = r->cell_cursor;
    ri.cell_cap = (uint32_t)cells_needed;
    r->cell_cursor += cells_needed;
    uint8_t *cells = r->cells.as<uint8_t>();
    for (uint64_t i = 0; i < cells_needed; ++i)
        std::memset(cells + (size_t)((ri.cell_base + i) & (r->ncells - 1)) * 64, 0, 64);
    ri.seq = r->next_seq++;
    ri.input = (uint32_t)in;
    pbsk::RingRoundStatus *hs = r->in_status((uint32_t)in);
    hs->seq = 0;
    // the service's self-stop handshake (kernels.hip): count and heartbeat first, THEN look at its intent flag
    volatile uint32_t *hb = hb_words(r);
    hb[pbsk::kHbRoundsEnq] = ++r->rounds_enq;
    ring_heartbeat(r);
    std::atomic_thread_fence(std::memory_order_seq_cst);
    {
        const int st = ring_service_check(r, true);
        if (st != PBSGPU_OK) return fail(st);
    }

    pbsk::RingRound rr{};
    rr.arena = r->arena.as<uint8_t>();
    rr.page_bytes = (uint32_t)r->page_bytes;
    rr.stride = (uint32_t)r->stride;
    rr.tile_bytes = r->tile_bytes;
    rr.tpp = r->tpp;
    rr.effmin = e->effmin;
    rr.cmin = e->cfg.min;
    rr.maxsz = e->cfg.max;
    rr.cap = r->cap;
    rr.thr = e->thr;
    rr.table_rot = e->d_table_rot;
    rr.pages = pg;
    rr.npages = np;
    rr.segs_in = sg;
    rr.nseg = ns;
    rr.seq = ri.seq;
    rr.cell_base = (uint32_t)(ri.cell_base & (r->ncells - 1));
    rr.cell_cap = ri.cell_cap;
    rr.cell_mask = r->ncells - 1;
    rr.scan_blocks = (uint32_t)std::max(1, e->num_cus - (int)r->svc_cus);
    rr.status = hs;
    rr.streams = r->streams.as<pbsk::RingStreamState>();
    rr.q = r->source();
    rr.desc_w = r->desc.as<uint4>();
    rr.ldesc_w = r->ldesc.as<uint4>();
    rr.sdesc_w = r->lanes_cus ? r->sdesc.as<uint4>() : nullptr;
    const uint32_t set = r->ps ? r->scan_set : 0u;
    rr.scalars = r->scalars.as<uint32_t>();
    rr.tile_queue = r->tileq.as<unsigned long long>() + set * 8u;
    rr.tile_cnt = set ? r->tile_cnt2.as<uint32_t>() : r->tile_cnt.as<uint32_t>();
    rr.tile_off = r->tile_off.as<uint32_t>();
    rr.tile_slots = set ? r->tile_slots2.as<uint32_t>() : r->tile_slots.as<uint32_t>();
    rr.scan_tmp = r->scan_tmp.as<uint32_t>();
    rr.dense = r->dense.as<uint64_t>();
    rr.dense_cap = r->dense_cap;
    rr.segs = r->segs.as<pbsgpu_segment>();
    rr.seg_cnt = r->seg_cnt.as<uint32_t>();
    rr.seg_off = r->seg_off.as<uint32_t>();
    rr.recs = r->recs.as<pbsgpu_record>();
    rr.rec_cap = r->rec_cap;
    rr.seg_newc = r->seg_newc.as<uint64_t>();
    rr.seg_open = r->seg_open.as<uint32_t>();
    rr.seg_ecand_in = r->seg_ecand_in.as<uint64_t>();
    rr.seg_ecand = r->seg_ecand.as<uint64_t>();
    rr.sugg = sugg_dev;
    rr.sugg_idx = sugg_dev ? suggidx : nullptr;
    rr.sugg_feed = feed;
    rr.sugg_abs = e->sugg_feed_abs.load(std::memory_order_relaxed);
    rr.seg_rec_base = recbase;
    std::atomic_thread_fence(std::memory_order_release);
    // Start the service with this round — unless it is worth cutting AHEAD of it: a LONE stream that delivers pages in bulk
    // (bytes already in device memory) cannot keep the service's lanes busy anyway (a 64 GiB file is 17 k chunks for 24 k
    // lanes: it is bound by the chain of its longest chunk), but its scan can use the WHOLE chip while no service holds
    // three quarters of it: 64 GiB are cut in ~20 ms at full width instead of ~75 ms on the cut side's quarter. The
    // service starts when the stream's bytes are all in, when another stream shows up, or after lone_defer_ms — and
    // finds every chunk queued (the queue was reset when the previous service ended, not now).
    // The same holds for the first rounds of SEVERAL bulk streams on an idle ring: until the service's lanes could all be
    // busy (lanes x average chunk size: ~96 GiB at avg 4 MiB) it only holds CUs the cut side could use — 96 GiB are cut in
    // ~26 ms on the whole chip, ~105 ms on a quarter of it with lanes waiting all the while. A trickle of pages (host-fed
    // streams: a few pages per pump) never defers: np is small there.
    bool defer = r->defer_service;
    if (!defer && r->svc == SvcState::Stopped && r->lone_defer_ms > 0 && !any_final &&
        np >= std::max<uint32_t>(1, r->round_pages / 2)) {
        const uint64_t lanes_worth = (uint64_t)r->sha_cus * (r->dense_service ? 256u : 128u) * (uint64_t)e->cfg.avg;
        const double t = now_ms();
        if (r->defer_t0 == 0) r->defer_t0 = t;
        defer = t - r->defer_t0 < r->lone_defer_ms && r->deferred_bytes + new_bytes < lanes_worth;
    }
    if (!defer) {
        const int st = ring_start_service(r);
        if (st != PBSGPU_OK) return fail(st);
        // the start may have moved the pair / express split (ring_adapt_split): this round publishes against the split it
        // will be served by, not the one before it (xp_pairs, long_spill, long_lo: heuristics only, records do not depend on them)
        rr.q = r->source();
    }
    if (r->svc == SvcState::Stopped) r->deferred_bytes += new_bytes;  // cut ahead of the service (or the start waits for a graveyard flush)
    if (r->svc == SvcState::Stopped) rr.scan_blocks = (uint32_t)std::max(1, e->num_cus);  // nobody else on the chip: full width
    else if (r->ps && rr.scan_blocks > 8) rr.scan_blocks -= 1;  // (one CU stays free for the control kernel that runs beside the scan)
    hipStream_t scan_st = r->ps ? r->ps : r->cs;
    for (hipEvent_t ev : deps)  // host-fed pages: the cut waits for their copies (device-side wait; the copies never wait for a kernel)
        if (hipStreamWaitEvent(scan_st, ev, 0) != hipSuccess) return fail(PBSGPU_E_HIP);
    for (hipEvent_t ev : deps) ring_event_put(r, ev);
    deps.clear();
    if (r->ps && r->ctl_used[set])  // this scan set's previous user must have been resolved before the scan overwrites it
        if (hipStreamWaitEvent(r->ps, r->ev_ctl[set], 0) != hipSuccess) return fail(PBSGPU_E_HIP);
    {
        // (PBSGPU_RING_F_FILL_SERIAL, experiments: the synthetic refill in stream order behind the previous round's scan instead
        // of beside it on its own stream)
        const bool fill_serial = r->fill_serial;
        pbsk::RingStage stg{};
        if (r->stage_inputs) {
            const uint8_t *hb = r->in((uint32_t)in);
            uint8_t *db = r->in_dev((uint32_t)in);
            stg.src = hb;
            stg.dst = db;
            stg.off[0] = (uint32_t)r->in_pages_off;   stg.len[0] = np * (uint32_t)sizeof(pbsk::RingPage);
            stg.off[1] = (uint32_t)r->in_segs_off;    stg.len[1] = ns * (uint32_t)sizeof(pbsk::RingSeg);
            stg.off[2] = (uint32_t)r->in_recbase_off; stg.len[2] = (ns + 1u) * 4u;
            stg.off[3] = (uint32_t)r->in_suggidx_off; stg.len[3] = rr.sugg_idx ? (ns + 1u) * 4u : 0u;
            stg.pages_host = pg;
            auto mirror = [&](const void *hp) { return db + (reinterpret_cast<const uint8_t *>(hp) - hb); };
            rr.pages = reinterpret_cast<const pbsk::RingPage *>(mirror(pg));
            rr.segs_in = reinterpret_cast<const pbsk::RingSeg *>(mirror(sg));
            rr.seg_rec_base = reinterpret_cast<const uint32_t *>(mirror(recbase));
            if (rr.sugg_idx) rr.sugg_idx = reinterpret_cast<const uint32_t *>(mirror(suggidx));
        }
        const hipError_t he = pbsk::launch_ring_round(rr, e->num_cus, r->cs, fill_serial ? nullptr : r->fs, r->ev_fill[in], r->ps,
                                                      r->ps ? r->ev_scan[in] : nullptr, r->stage_inputs ? &stg : nullptr);
        if (he != hipSuccess) {
            g_last_hip_error.store((int)he);
            return fail(PBSGPU_E_HIP);
        }
    }
    if (r->ps) {
        if (hipEventRecord(r->ev_ctl[set], r->cs) != hipSuccess) return fail(PBSGPU_E_HIP);
        r->ctl_used[set] = true;
        r->scan_set ^= 1u;
    }
    r->input_busy[in] = true;
    ri.new_bytes = new_bytes;
    r->ready_bytes -= new_bytes;
    r->inflight_bytes += new_bytes;
    r->rounds.push_back(std::move(ri));
    r->st.rounds++;
    r->st.bytes_enqueued += new_bytes;
    r->st.pages_enqueued += np;
    *did = true;
    return PBSGPU_OK;
}

int ring_take_page(pbsgpu_ring *r, uint32_t *phys) {
    if (r->backlog_limit && r->svc == SvcState::Running) {
        // published and unclaimed: queue tail of the last reaped round minus the service's claim progress (a word of the
        // heartbeat block, written whenever the head crosses a multiple of 256 — hence the slack: idle lanes claim ahead
        // of the tail, so with nothing left to claim the difference always falls below 256)
        const int32_t behind = (int32_t)(r->tail_seen - hb_words(r)[pbsk::kHbClaim]) - 256;
        double wait = (double)r->ready_bytes + (double)r->inflight_bytes;
        if (behind > 0 && r->pub_positions) wait += (double)behind * ((double)r->pub_bytes / (double)r->pub_positions);
        if (wait > (double)r->backlog_limit) return PBSGPU_E_BUSY;
    }
    if (r->free_pages.empty()) ring_reap_free(r);
    if (r->free_pages.empty()) return PBSGPU_E_BUSY;
    *phys = r->free_pages.back();
    r->free_pages.pop_back();
    return PBSGPU_OK;
}

// DEBUG overrides of pbsgpu_ring_options by environment: the variable names of rounds 3-5, ONE table, one getenv. What a
// host passes in the options is what counts (two engines of one process may want different rings); these exist so that an
// unmodified binary can be A/B-tested (scripts/, the parity tests that force a code path).
void ring_env_overrides(pbsgpu_ring_options &o) {
    enum Kind { U32, U32_ZERO_OFF, U64, F64, F64_ZERO_NEG, FLAG_IF_ZERO, FLAG_IF_SET };
    struct Entry {
        const char *name;
        Kind kind;
        void *field;
        uint32_t flag;
    };
    const Entry table[] = {
        {"PBSGPU_RING_SHA_CUS", U32, &o.sha_cus, 0},
        {"PBSGPU_RING_XP_CUS", U32_ZERO_OFF, &o.express_cus, 0},
        {"PBSGPU_RING_LANES_CUS", U32, &o.lanes_cus, 0},
        {"PBSGPU_RING_SHORT_BYTES", U64, &o.short_bytes, 0},
        {"PBSGPU_RING_ROUND_PAGES", U32, &o.round_pages, 0},
        {"PBSGPU_RING_MIN_ROUND_PAGES", U32, &o.min_round_pages, 0},
        {"PBSGPU_RING_MAX_INFLIGHT", U32, &o.max_inflight, 0},
        {"PBSGPU_RING_LONG_BYTES", U32_ZERO_OFF, &o.long_bytes, 0},
        {"PBSGPU_RING_LONG_LO_BYTES", U32_ZERO_OFF, &o.long_lo_bytes, 0},
        {"PBSGPU_RING_LONG_SPILL", U32, &o.long_spill, 0},
        {"PBSGPU_RING_POLL_EVERY", U32, &o.poll_every, 0},
        {"PBSGPU_RING_BACKLOG_MIB", F64_ZERO_NEG, &o.backlog_mib, 0},
        {"PBSGPU_RING_LONE_DEFER_MS", F64_ZERO_NEG, &o.lone_defer_ms, 0},
        {"PBSGPU_RING_IDLE_TIMEOUT_S", F64, &o.idle_timeout_s, 0},
        {"PBSGPU_RING_AUTOPARK_MS", F64, &o.autopark_ms, 0},
        {"PBSGPU_RING_OVERLAP", FLAG_IF_ZERO, &o.flags, PBSGPU_RING_F_NO_OVERLAP},
        {"PBSGPU_RING_STAGE_INPUTS", FLAG_IF_ZERO, &o.flags, PBSGPU_RING_F_NO_STAGE},
        {"PBSGPU_RING_CUT_PRIO", FLAG_IF_ZERO, &o.flags, PBSGPU_RING_F_NO_CUT_PRIO},
        {"PBSGPU_RING_SPLIT_AUTO", FLAG_IF_ZERO, &o.flags, PBSGPU_RING_F_NO_SPLIT_AUTO},
        {"PBSGPU_RING_DEFER_SERVICE", FLAG_IF_SET, &o.flags, PBSGPU_RING_F_DEFER_SERVICE},
        {"PBSGPU_RING_FILL_SERIAL", FLAG_IF_SET, &o.flags, PBSGPU_RING_F_FILL_SERIAL},
        {"PBSGPU_RING_DENSE_SERVICE", FLAG_IF_SET, &o.flags, PBSGPU_RING_F_DENSE_SERVICE},
        {"PBSGPU_RING_DENSE_LANES", FLAG_IF_SET, &o.flags, PBSGPU_RING_F_DENSE_LANES},
        {"PBSGPU_RING_TIER_TAG", FLAG_IF_SET, &o.flags, PBSGPU_RING_F_TIER_TAG},
    };
    for (const Entry &e : table) {
        const char *v = getenv(e.name);
        if (!v || !*v) continue;
        switch (e.kind) {
        case U32: *static_cast<uint32_t *>(e.field) = (uint32_t)std::max(0L, atol(v)); break;
        case U32_ZERO_OFF: *static_cast<uint32_t *>(e.field) = atol(v) <= 0 ? PBSGPU_RING_OFF : (uint32_t)atol(v); break;
        case U64: *static_cast<uint64_t *>(e.field) = strtoull(v, nullptr, 10); break;
        case F64: *static_cast<double *>(e.field) = std::max(0.0, atof(v)); break;
        case F64_ZERO_NEG: *static_cast<double *>(e.field) = atof(v) <= 0.0 ? -1.0 : atof(v); break;
        case FLAG_IF_ZERO: if (atoi(v) == 0) *static_cast<uint32_t *>(e.field) |= e.flag; break;
        case FLAG_IF_SET: if (atoi(v) != 0) *static_cast<uint32_t *>(e.field) |= e.flag; break;
        }
    }
}

}  // namespace

pbsk::RingSource pbsgpu_ring::source() const {
    pbsk::RingSource q{};
    q.desc = desc.as<uint4>();
    q.qmask = qslots - 1;
    q.probe = probe.as<unsigned long long>();
    q.ldesc = ldesc.as<uint4>();
    q.lmask = lslots - 1;
    q.long_bytes = long_bytes;
    q.sdesc = sdesc.as<uint4>();
    q.smask = sslots ? sslots - 1 : 0;
    q.short_bytes = lanes_cus ? short_bytes : 0u;
    // entries that may wait for the lanes service: its lanes take ~2 chunks per CU and round (256 lanes x 1.2 ms / ~0.15 s per
    // chunk); 32 per CU rides out a dozen rounds and fills an idle service within ten
    q.short_room = lanes_cus * (dense_lanes ? 64u : 32u);
    q.xp = xp_cus ? 1u : 0u;
    // the pair lanes take a long chunk only while EVERY express pair is busy (RingCtl::xp_busy): random data keeps the express
    // service just busy (2.6 long chunks per ms against the 2.8 it can take), a corpus whose files are mostly zero runs or
    // periodic (configs[2]: half the bytes in 16 MiB chunks) swamps 16 CUs at once — and a long chunk that WAITS for an express
    // pair (0.36 s + 0.36 s) finishes later than on a pair lane that is free now (0.49 s), holding its page all the while
    // (measured with a queue-length rule instead: ring_manyfiles 441 -> 418 GiB/s, drain 0.50 -> 0.58 s)
    q.long_spill = 0u;
    q.xp_pairs = xp_cus * 64u;
    // ... unless the express service is LARGE (the split has followed a corpus of long chunks, ring_adapt_split): with
    // thousands of express pairs one comes free every few dozen microseconds (6 656 pairs x 0.34 s per 16 MiB chunk: one per
    // 50 us), so a long chunk near the head of the queue is on an express pair — and done 0.15 s sooner than on a pair lane —
    // almost at once. The pair lanes then only see what is queued beyond xp_pairs / 16 (<= 21 ms of waiting).
    if (xp_cus >= 32) q.long_spill = q.xp_pairs / 16u;
    // light load (fewer than 3/4 of the express pairs taken): chunks from 11/16 of the maximum on go express too — bulk rings
    // only (the stream writer's ring already sends every chunk >= half the maximum express). One 64 GiB file alone: ~400 chunks
    // >= 11 MiB for 1 024 pairs; its last record then waits for the express chain of a 16 MiB chunk (0.34 s), not for the pair
    // chain of a 12.9 MiB one (0.37 s).
    q.long_lo = (xp_cus && long_lo_auto) ? (uint32_t)((uint64_t)eng->cfg.max * 11 / 16) : 0u;
    if (opt_long_lo) q.long_lo = (xp_cus && opt_long_lo != PBSGPU_RING_OFF) ? opt_long_lo : 0u;
    if (q.long_lo >= q.long_bytes) q.long_lo = 0u;
    if (opt_long_spill) q.long_spill = opt_long_spill;
    q.ctl = ctl.as<pbsk::RingCtl>();
    q.cells = cells.as<uint8_t>();
    q.pending = pending.as<uint32_t>();
    q.free_fifo = free_fifo.as<unsigned long long>();
    q.free_mask = nfree - 1;
    const double idle_s = idle_timeout_s > 0 ? std::max(0.05, idle_timeout_s) : 20.0;
    q.idle_ticks = (unsigned long long)(idle_s * 100e6);  // wall_clock64 runs at 100 MHz
    q.heartbeat = heartbeat.as<uint32_t>();
    {   // poll period of waves that carry work (power of two; 1 = every step, the behaviour before round 4)
        // A free lane waits up to `every` block steps (1.75 us each) for its next look at the queue: kept below 0.4 % of the
        // chain of a MINIMUM-size chunk — 8 for small chunkers (tests), 64 at the production average of 4 MiB (min 1 MiB =
        // 16 384 steps). Measured on the driver's command: every 8th step 610.5 / 609.7, every 32nd 617.0 GiB/s, drain
        // 0.430 -> 0.409 s (profiles/r05_ab_long_spill_and_poll.log; round 4: 1 -> 8: 606 -> 615).
        const uint64_t min_steps = std::min<uint64_t>(eng->effmin, eng->cfg.min) / 64u;
        int every = (int)std::min<uint64_t>(64, std::max<uint64_t>(8, pow2_at_least(min_steps / 256u + 1u) / 2u));
        if (opt_poll_every) every = (int)opt_poll_every;
        q.poll_mask = pow2_at_least((uint64_t)every) - 1u;
    }
    return q;
}

namespace pbse {

// records of one stream that are ready, in order, into out[*n ..)
void ring_pop_records(pbsgpu_ring *r, uint32_t slot, pbsgpu_record *out, uint64_t cap, uint64_t *n) {
    StreamSlot &s = r->slots[slot];
    const uint8_t *cells = r->cells.as<uint8_t>();
    while (*n < cap && !s.cells.empty()) {
        const CellRef cr = s.cells.front();
        const uint8_t *c = cells + (size_t)cr.cell * 64;
        const volatile uint32_t *flag = reinterpret_cast<const volatile uint32_t *>(c + 48);
        if (*flag != 1u) break;  // its chunk is still being hashed: records come out in stream order
        std::atomic_thread_fence(std::memory_order_acquire);
        pbsgpu_record rec;
        std::memcpy(&rec, c, sizeof(rec));
        rec.segment = slot;
        if (r->tier_tag) rec.segment |= (reinterpret_cast<const uint32_t *>(c)[13] & 3u) << 28;  // PBSGPU_RING_F_TIER_TAG
        out[(*n)++] = rec;
        s.cells.pop_front();
        s.records_out++;
        for (auto &ri : r->rounds)
            if (ri.seq == cr.round_idx) {
                ri.live_cells--;
                break;
            }
    }
}

int ring_event_get(pbsgpu_ring *r, hipEvent_t *ev) {
    if (!r->ev_pool.empty()) {
        *ev = r->ev_pool.back();
        r->ev_pool.pop_back();
        return PBSGPU_OK;
    }
    HIPCHK(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return PBSGPU_OK;
}

void ring_event_put(pbsgpu_ring *r, hipEvent_t ev) {
    if (ev) r->ev_pool.push_back(ev);
}

bool ring_idle(const pbsgpu_ring *r) {
    if (!r->rounds.empty() || r->ready_bytes || r->inflight_bytes) return false;
    for (auto &s : r->slots)
        if (s.open && (!s.ready.empty() || s.zero_final || !s.cells.empty())) return false;
    return true;
}

// Stop the service behind everything enqueued so far without waiting for it: the kernel drains what is published and
// ends; the next round starts a new one (which first waits, on the device, for the old one's end).
int ring_park(pbsgpu_ring *r) {
    if (r->svc != SvcState::Running) return PBSGPU_OK;
    CHK(ring_service_check(r, false));
    if (r->svc != SvcState::Running) return PBSGPU_OK;
    HIPCHK(pbsk::launch_ring_stop(r->ctl.as<pbsk::RingCtl>(), r->cs));
    r->svc = SvcState::Stopping;
    return PBSGPU_OK;
}

int ring_commit_dep(pbsgpu_ring *r, uint32_t stream, uint64_t nbytes, int final, hipEvent_t dep) {
    if (!r || stream >= r->slots.size() || !r->slots[stream].open) return PBSGPU_E_INVALID;
    StreamSlot &s = r->slots[stream];
    if (s.final_committed) return PBSGPU_E_STATE;
    if (nbytes > r->page_bytes || (nbytes != r->page_bytes && !final)) return PBSGPU_E_INVALID;  // only a stream's last page is short
    if (s.bytes_committed + nbytes > pbsk::kRingMaxStream) return PBSGPU_E_INVALID;  // 4 PiB per stream (52-bit logical offsets)
    if (nbytes == 0) {
        if (s.reserved >= 0) {  // nothing written: the page goes back
            r->free_pages.push_back((uint32_t)s.reserved);
            s.reserved = -1;
        }
        s.zero_final = true;
        s.final_committed = true;
        return PBSGPU_OK;
    }
    if (s.reserved < 0) return PBSGPU_E_STATE;
    PageReq q;
    q.phys = (uint32_t)s.reserved;
    q.k = s.next_k++;
    q.valid = (uint32_t)nbytes;
    q.final = final != 0;
    q.dep = dep;
    s.ready.push_back(q);
    r->ready_bytes += nbytes;
    s.reserved = -1;
    s.bytes_committed += nbytes;
    if (final) s.final_committed = true;
    return PBSGPU_OK;
}

int ring_create_internal(pbsgpu_engine *e, const pbsgpu_ring_options *opt, bool hold_engine_ref, pbsgpu_ring **out,
                         uint32_t long_bytes_hint) {
    if (!e || !out) return PBSGPU_E_INVALID;
    *out = nullptr;
    CHK(set_device(e));
    pbsgpu_ring_options o{};
    if (opt) o = *opt;
    ring_env_overrides(o);
    pbsgpu_ring *r = new (std::nothrow) pbsgpu_ring();
    if (!r) return PBSGPU_E_NOMEM;
    r->holds_engine_ref = hold_engine_ref;
    if (hold_engine_ref) engine_ref(e);
    r->eng = e;
    int st = [&]() -> int {
        // page geometry: a whole number of scan tiles, >= the largest chunk (so a chunk touches at most two pages)
        const uint32_t big = 64u * 34u * 128u, small = 64u * 4u * 128u;
        uint64_t page = o.page_bytes;
        if (page == 0) {
            const uint32_t tile = e->cfg.max >= (1u << 20) ? big : small;
            page = ((uint64_t)e->cfg.max + tile - 1) / tile * tile;
            if (page < 2ull * tile && tile == small) page = 2ull * tile;
        }
        if (page % big == 0) r->tile_bytes = big;
        else if (page % small == 0) r->tile_bytes = small;
        else return PBSGPU_E_INVALID;
        if (page < e->cfg.max || page >= (1ull << 31)) return PBSGPU_E_INVALID;
        r->page_bytes = page;
        r->tpp = (uint32_t)(page / r->tile_bytes);
        r->stride = page + 256;
        uint64_t arena_bytes = o.arena_bytes;
        if (arena_bytes == 0) {
            size_t fr = 0, tot = 0;
            HIPCHK(hipMemGetInfo(&fr, &tot));
            arena_bytes = fr > (12ull << 30) ? fr - (8ull << 30) : fr / 2;
        }
        r->npages = (uint32_t)std::min<uint64_t>(arena_bytes / r->stride, 65534);  // 16-bit page ids in the queue descriptors
        {   // the chunk FIFO and the record cells are sized for every chunk the arena can hold (arena / min chunk size):
            // with small average chunk sizes that bound, not HBM, limits the arena (4 M resident chunks = 128 MB of
            // descriptors + 1 GB of pinned record cells)
            const uint64_t lim = (4ull << 20) * std::min(e->effmin, e->cfg.min) / r->page_bytes;
            if (r->npages > lim) r->npages = (uint32_t)std::max<uint64_t>(lim, 4);
        }
        if (r->npages < 4) return PBSGPU_E_INVALID;
        r->max_streams = o.max_streams ? o.max_streams : 64;
        if (r->max_streams > 4096) return PBSGPU_E_INVALID;
        // 3/4 of the chip hashes, 1/4 cuts: per GiB the cut rounds cost ~70 CU-ms (scan 52 + refill 18), the service
        // ~233 CU-ms (128 chains per CU at 1.66-1.75 us per 64-byte block) — measured optimum 192 of 256 CUs
        // (profiles/r03_ring_sweep_*.log: 184 -> 580, 192 -> 597, 200 -> 528, 208 -> 504 GiB/s)
        int sha = o.sha_cus ? (int)o.sha_cus : std::max(1, e->num_cus - e->num_cus / 4);
        // EXPRESS service: that many CUs run k_sha256_xpair — two lanes per chunk: the chain of a chunk 1.37x faster (1.28
        // vs 1.75 us per block, profiles/r04_chain_time_pair_vs_express.log) at 0.65 of the throughput per CU — on the
        // chunks of at least long_bytes. Bulk rings (default service share): 16 CUs for chunks >= 13/16 of the maximum
        // (1.3 % of random data's chunks, 5 % of its bytes): the driver's line is unchanged within noise (the drain gets
        // 0.08 s shorter, the feed phase 4 % slower: profiles/r04_ab_express_service.log), one file alone is 15 % sooner.
        // The CUs come out of the pair service's share unless that was given explicitly.
        int xp = o.express_cus == PBSGPU_RING_OFF ? 0 : (int)o.express_cus;
        if (o.express_cus == 0 && !o.sha_cus && e->num_cus >= 128) xp = 16;
        xp = std::min(xp, std::max(0, e->num_cus / 2));
        if (xp && !o.sha_cus) sha = std::max(1, sha - xp);
        // The cut side needs its share: with 13/16 of the chip (208 of 256 CUs) in service workgroups the driver's line falls
        // to 525-545 GiB/s, with 216 the cut kernels barely find a CU (a warm-up of 5 files took 120 s:
        // profiles/r04_ab_cu_split.log). Whatever was asked for, the services together get at most 3/4 of the chip + 8.
        const int svc_max = std::max(2, e->num_cus - e->num_cus / 4 + (e->num_cus >= 64 ? 8 : 0));
        if (sha + xp > svc_max) sha = std::max(1, svc_max - xp);
        r->xp_cus = (uint32_t)xp;
        r->sha_cus = (uint32_t)std::min(std::max(sha, 1), std::max(1, e->num_cus - 1 - xp));
        // LANES service: out of the pair service's share (at most 3/4 of it)
        r->lanes_cus = std::min<uint32_t>(o.lanes_cus, r->sha_cus * 3u / 4u);
        r->sha_cus -= r->lanes_cus;
        r->short_bytes = o.short_bytes ? (uint32_t)std::min<uint64_t>(o.short_bytes, e->cfg.max) : (uint32_t)((uint64_t)e->cfg.avg * 3 / 2);
        r->svc_cus = r->sha_cus + r->xp_cus + r->lanes_cus;
        r->split_auto = xp > 0 && !o.sha_cus && !o.express_cus && r->svc_cus >= 96 && !(o.flags & PBSGPU_RING_F_NO_SPLIT_AUTO);
        r->round_pages = o.round_pages ? o.round_pages : 256;
        r->round_pages = std::min(r->round_pages, r->npages);
        r->min_round_pages = o.min_round_pages ? std::max(1u, std::min(o.min_round_pages, std::max(1u, r->round_pages / 4)))
                                               : std::max(1u, r->round_pages / 4);
        // ~30 ms of the service's throughput (4.3 GiB/s per CU measured) is plenty to ride out the gaps between rounds
        r->backlog_limit = o.backlog_mib < 0 ? 0 : o.backlog_mib > 0 ? (uint64_t)(o.backlog_mib * 1048576.0)   : (uint64_t)(r->sha_cus + r->lanes_cus) << 27;
        r->backlog_auto = o.backlog_mib == 0;
        if (o.max_inflight) r->max_inflight = std::min<uint32_t>(std::max(1u, o.max_inflight), kRingInputs);
        r->autopark_ms = std::max(0.0, o.autopark_ms);
        r->defer_service = (o.flags & PBSGPU_RING_F_DEFER_SERVICE) != 0;
        r->fill_serial = (o.flags & PBSGPU_RING_F_FILL_SERIAL) != 0;
        r->dense_service = (o.flags & PBSGPU_RING_F_DENSE_SERVICE) != 0;
        r->dense_lanes = (o.flags & PBSGPU_RING_F_DENSE_LANES) != 0;
        r->tier_tag = (o.flags & PBSGPU_RING_F_TIER_TAG) != 0;
        r->lone_defer_ms = o.lone_defer_ms < 0 ? 0.0 : o.lone_defer_ms > 0 ? o.lone_defer_ms : 25.0;
        r->idle_timeout_s = o.idle_timeout_s;
        r->opt_long_lo = o.long_lo_bytes;
        r->opt_long_spill = o.long_spill;
        r->opt_poll_every = o.poll_every;
        // Candidate slots per scan tile. The batch path starts small and RE-RUNS a batch whose tile overflowed; a ring round
        // cannot be re-run (later rounds continue from it), so the ring provisions for periodic data up front: one
        // candidate per 128 bytes (a repeating block of >= 128 bytes whose every period holds a candidate — BASELINE
        // configs[2]'s repeating 4 KiB files have one per 4 KiB). 12 bytes per slot: ~400 MB at the default round size.
        // A tile beyond that (a crafted short period) keeps what fits and is re-scanned on demand by the resolve walk
        // (DenseTiles, kernels.h): slower for that stretch, exact all the same.
        const double lambda = 3.0 * r->tile_bytes / ((double)e->cfg.mask + 1.0);
        uint32_t capv = 8;
        while (capv < 4.0 * lambda + 16.0) capv <<= 1;
        r->cap = std::min<uint32_t>(std::max<uint32_t>(capv * 2, r->tile_bytes / 128), r->tile_bytes);
        const uint64_t ntiles = (uint64_t)r->round_pages * r->tpp;
        if (ntiles * r->cap >= (1ull << 32)) return PBSGPU_E_INVALID;  // (round_pages far beyond anything an arena holds)
        const uint32_t minsz = std::min(e->effmin, e->cfg.min);
        r->dense_cap = ntiles * r->cap;
        r->rec_cap = ((uint64_t)r->round_pages * r->page_bytes + (uint64_t)r->max_streams * e->cfg.max) / minsz +
                     4ull * r->max_streams + 64;
        const uint64_t resident_chunks = (uint64_t)r->npages * r->page_bytes / minsz + 2ull * r->npages + r->rec_cap;
        r->qslots = pow2_at_least(2 * resident_chunks + 4096);
        r->ncells = pow2_at_least(4 * resident_chunks + 4 * r->rec_cap);
        r->nfree = pow2_at_least(4ull * r->npages + 64);

        CHK(r->arena.ensure((size_t)r->npages * r->stride + 512));
        CHK(r->ctl.ensure(256));
        CHK(r->probe.ensure(128));
        CHK(r->streams.ensure((size_t)r->max_streams * sizeof(pbsk::RingStreamState)));
        CHK(r->pending.ensure((size_t)r->npages * 4 + 64));
        CHK(r->desc.ensure((size_t)r->qslots * 32));
        // optional long-chunk queue (PBSGPU_RING_LONG_BYTES, e.g. half the maximum chunk size = 8 % of the chunks, 30 % of the
        // bytes of random data): idle lanes look at it first
        // (OFF by default: measured +0.8 % on the bench line for +40 ms of single-file latency — the drain is not made of
        // late-starting long chunks; kept as a switch)
        r->long_bytes = r->xp_cus ? (long_bytes_hint ? long_bytes_hint : (uint32_t)((uint64_t)e->cfg.max * 13 / 16)) : 0;
        r->long_lo_auto = long_bytes_hint == 0;  // (a ring with its own threshold — the stream writer's — keeps it)
        if (o.long_bytes) r->long_bytes = o.long_bytes == PBSGPU_RING_OFF ? 0u : o.long_bytes;
        if (r->xp_cus && r->long_bytes == 0) r->xp_cus = 0;  // (no long queue: nothing the express service could take)
        r->lslots = pow2_at_least(2 * ((uint64_t)r->npages * r->page_bytes / std::max<uint32_t>(r->long_bytes, minsz) + r->rec_cap) + 1024);
        CHK(r->ldesc.ensure((size_t)r->lslots * 32));
        r->sslots = r->lanes_cus ? r->qslots : 2;
        CHK(r->sdesc.ensure((size_t)r->sslots * 32));
        CHK(r->scalars.ensure(pbsk::kRsCount * 4 + 64));
        CHK(r->tile_cnt.ensure((size_t)ntiles * 4 + 16));
        CHK(r->tile_off.ensure((size_t)ntiles * 4 + 16));
        CHK(r->tile_slots.ensure((size_t)ntiles * r->cap * 4 + 16));
        CHK(r->tileq.ensure(128));
        const bool overlap = !(o.flags & PBSGPU_RING_F_NO_OVERLAP);
        if (overlap) {
            CHK(r->tile_cnt2.ensure((size_t)ntiles * 4 + 16));
            CHK(r->tile_slots2.ensure((size_t)ntiles * r->cap * 4 + 16));
        }
        CHK(r->dense.ensure((size_t)r->dense_cap * 8 + 16));
        CHK(r->scan_tmp.ensure(pbsk::scan_tmp_words(std::max<uint64_t>(ntiles, r->max_streams)) * 4 + 64));
        CHK(r->segs.ensure((size_t)r->max_streams * sizeof(pbsgpu_segment)));
        CHK(r->seg_cnt.ensure((size_t)r->max_streams * 4 + 16));
        CHK(r->seg_off.ensure((size_t)r->max_streams * 4 + 16));
        CHK(r->seg_newc.ensure((size_t)r->max_streams * 8 + 16));
        CHK(r->seg_open.ensure((size_t)r->max_streams * 4 + 16));
        CHK(r->seg_ecand_in.ensure((size_t)r->max_streams * 8 + 16));
        CHK(r->seg_ecand.ensure((size_t)r->max_streams * 8 + 16));
        CHK(r->recs.ensure((size_t)r->rec_cap * sizeof(pbsgpu_record) + 64));
        CHK(r->cells.ensure((size_t)r->ncells * 64));
        CHK(r->heartbeat.ensure(256));
        std::memset(r->heartbeat.p, 0, 256);
        CHK(r->free_fifo.ensure((size_t)r->nfree * 8));
        std::memset(r->free_fifo.p, 0, (size_t)r->nfree * 8);
        auto al64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
        r->in_pages_off = 0;
        r->in_segs_off = al64((size_t)r->round_pages * sizeof(pbsk::RingPage));
        r->in_recbase_off = al64(r->in_segs_off + (size_t)r->max_streams * sizeof(pbsk::RingSeg));
        r->in_suggidx_off = al64(r->in_recbase_off + ((size_t)r->max_streams + 1) * 4);
        r->in_status_off = al64(r->in_suggidx_off + ((size_t)r->max_streams + 1) * 4);
        r->input_stride = r->in_status_off + 128;
        CHK(r->inputs.ensure(r->input_stride * kRingInputs));
        // PBSGPU_RING_F_NO_STAGE: the round's kernels read their tables from mapped host memory (rounds 3-4), for A/B runs
        r->stage_inputs = !(o.flags & PBSGPU_RING_F_NO_STAGE);
        if (r->stage_inputs) CHK(r->inputs_dev.ensure(r->input_stride * kRingInputs));
        std::memset(r->inputs.p, 0, r->input_stride * kRingInputs);
        // Three priorities: the services highest (their own hardware-queue pool: nothing may queue behind a kernel that only
        // ends on request), the CONTROL side of the rounds normal, the producers of bulk work — synthetic refill, scan —
        // lowest: the one-workgroup control kernel and its prep launch find a CU as soon as one frees up instead of waiting
        // behind the next round's scan workgroups (kernel trace of the driver's command: k_ring_prep 1.2 ms and
        // k_ring_control 1.1 ms per launch for 10 us / 170 us of work; PBSGPU_RING_CUT_PRIO=0: all normal).
        int prio_lo = 0, prio_hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        const bool cut_prio = !(o.flags & PBSGPU_RING_F_NO_CUT_PRIO);
        r->bulk_prio = cut_prio ? prio_lo : 0;
        HIPCHK(hipStreamCreateWithFlags(&r->cs, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithPriority(&r->fs, hipStreamNonBlocking, r->bulk_prio));
        // The device-side state starts from zero — cleared ON THE CONTROL STREAM and waited for. A plain hipMemset is
        // queued on the null stream and returns at once for device memory; the ring's streams are non-blocking, i.e. not
        // ordered against the null stream: when the null stream was held up (it waits for every blocking stream's earlier
        // work, e.g. the previous engine's last kernels) the clear arrived AFTER the first rounds had run and wiped queue
        // tail, stream states and page reference counts — pages never came back, lanes waited at positions the tail had
        // been reset below (the one-in-a-few-hundred hang of the small-ring tests).
        HIPCHK(hipMemsetAsync(r->ctl.p, 0, 256, r->cs));
        HIPCHK(hipMemsetAsync(r->probe.p, 0, 128, r->cs));
        HIPCHK(hipMemsetAsync(r->streams.p, 0, (size_t)r->max_streams * sizeof(pbsk::RingStreamState), r->cs));
        HIPCHK(hipMemsetAsync(r->pending.p, 0, (size_t)r->npages * 4 + 64, r->cs));
        HIPCHK(hipMemsetAsync(r->scalars.p, 0, pbsk::kRsCount * 4 + 64, r->cs));
        HIPCHK(hipMemsetAsync(r->tileq.p, 0, 128, r->cs));
        HIPCHK(hipStreamSynchronize(r->cs));
        for (auto &ev : r->ev_fill) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        if (overlap) {
            HIPCHK(hipStreamCreateWithPriority(&r->ps, hipStreamNonBlocking, r->bulk_prio));
            for (auto &ev : r->ev_scan) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            for (auto &ev : r->ev_ctl) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        // the service must never share a hardware queue with a stream that enqueues behind it (packets of one queue
        // run in order: work queued behind a kernel that only ends on request would never start). HIP keeps one queue
        // pool per priority: the service gets the highest priority to itself.
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&r->ss, hipStreamNonBlocking, hi));
        if (r->xp_cus) {
            HIPCHK(hipStreamCreateWithPriority(&r->xs, hipStreamNonBlocking, hi));
            HIPCHK(hipEventCreateWithFlags(&r->ev_xsvc1, hipEventDisableTiming));
        }
        if (r->lanes_cus) {
            HIPCHK(hipStreamCreateWithPriority(&r->ls, hipStreamNonBlocking, hi));
            HIPCHK(hipEventCreateWithFlags(&r->ev_lsvc1, hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(&r->ev_reset, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&r->ev_svc0));
        HIPCHK(hipEventCreate(&r->ev_svc1));
        r->slots.resize(r->max_streams);
        r->piece_tab.resize(r->max_streams);
        r->piece_n.assign(r->max_streams, 0);
        r->free_pages.reserve(r->npages);
        for (uint32_t p = r->npages; p-- > 0;) r->free_pages.push_back(p);  // page 0 is handed out first
        r->st.pages_total = r->npages;
        r->st.page_bytes = r->page_bytes;
        r->st.sha_cus = r->sha_cus;
        return PBSGPU_OK;
    }();
    if (st != PBSGPU_OK) {
        pbsgpu_ring_destroy(r);
        return st;
    }
    *out = r;
    return PBSGPU_OK;
}

}  // namespace pbse

extern "C" {

int pbsgpu_ring_create(pbsgpu_engine *e, const pbsgpu_ring_options *opt, pbsgpu_ring **out) {
    return ring_create_internal(e, opt, true, out);
}

// Stop the SHA service behind everything enqueued so far and wait until the device holds no ring work: every chunk of
// every enqueued round is hashed, the persistent kernel has ended (hipDeviceSynchronize / hipFree can return again).
// The next pump starts the service again.
int pbsgpu_ring_quiesce(pbsgpu_ring *r) {
    if (!r) return PBSGPU_E_INVALID;
    CHK(set_device(r->eng));
    ring_heartbeat(r);
    if (r->defer_service && r->svc == SvcState::Stopped) {
        // Profiling mode (PBSGPU_RING_DEFER_SERVICE=1; scripts/r4_ring_pmc.py): the rounds have only filled the queue. Now
        // the service runs ALONE over everything published, `stop` already raised, and ends — one ordinary dispatch that
        // rocprofv3's counter passes (which serialise dispatches) can measure: k_sha256_pair<RingSource,false> with every
        // lane busy, reading chunks that cross pages, releasing pages. (The arena must hold what was fed: no page comes
        // back before this point.)
        HIPCHK(hipStreamSynchronize(r->cs));
        HIPCHK(pbsk::launch_ring_stop(r->ctl.as<pbsk::RingCtl>(), r->cs));
        HIPCHK(hipEventRecord(r->ev_reset, r->cs));
        CHK(ring_launch_services(r));
        service_started(r->eng->device);
        r->svc = SvcState::Running;
        r->st.service_launches++;
        HIPCHK(hipStreamSynchronize(r->ss));
        ring_service_ended(r);  // (head := tail, stop := 0 for the next batch of rounds)
        HIPCHK(hipStreamSynchronize(r->cs));
        r->svc_bytes0 = r->st.bytes_enqueued;
        ring_reap_free(r);
        ring_reap_rounds(r);
        return r->error;
    }
    if (r->svc == SvcState::Stopped && r->deferred_bytes > 0) {
        // rounds were cut AHEAD of the service (a lone bulk stream on an idle ring: ring_enqueue_round): their chunks sit in
        // the queue with no service to hash them. "Everything enqueued is hashed" needs one: start it, then stop it behind
        // what is published, like a running one (round 5; before that fill -> pump -> quiesce -> poll saw no records)
        CHK(ring_start_service(r, true));
    }
    if (r->svc == SvcState::Running) {
        CHK(ring_service_check(r, false));  // (it may have stopped on its own in the meantime)
        if (r->svc == SvcState::Running) HIPCHK(pbsk::launch_ring_stop(r->ctl.as<pbsk::RingCtl>(), r->cs));
    }
    HIPCHK(hipStreamSynchronize(r->cs));
    if (r->svc != SvcState::Stopped) {
        HIPCHK(hipStreamSynchronize(r->ss));
        HIPCHK(hipStreamSynchronize(r->fs));
        ring_service_ended(r);
    }
    ring_reap_free(r);
    ring_reap_rounds(r);
    return r->error;
}

int pbsgpu_ring_park(pbsgpu_ring *r) {
    if (!r) return PBSGPU_E_INVALID;
    CHK(set_device(r->eng));
    ring_heartbeat(r);
    CHK(ring_park(r));
    return r->error;
}

void pbsgpu_ring_destroy(pbsgpu_ring *r) {
    if (!r) return;
    pbsgpu_engine *e = r->eng;
    if (e) {
        (void)hipSetDevice(e->device);
        if (r->cs && r->ss) (void)pbsgpu_ring_quiesce(r);
        if (r->svc != SvcState::Stopped) {  // quiesce failed half-way (HIP error): the count must not leak
            r->svc = SvcState::Stopped;
            service_ended(r->eng->device);
        }
        if (r->ss) (void)hipStreamDestroy(r->ss);
        if (r->xs) (void)hipStreamDestroy(r->xs);
        if (r->ev_xsvc1) (void)hipEventDestroy(r->ev_xsvc1);
        if (r->ls) (void)hipStreamDestroy(r->ls);
        if (r->ev_lsvc1) (void)hipEventDestroy(r->ev_lsvc1);
        if (r->cs) (void)hipStreamDestroy(r->cs);
        if (r->ps) (void)hipStreamDestroy(r->ps);
        if (r->fs) (void)hipStreamDestroy(r->fs);
        for (auto ev : r->ev_scan)
            if (ev) (void)hipEventDestroy(ev);
        for (auto ev : r->ev_ctl)
            if (ev) (void)hipEventDestroy(ev);
        for (auto ev : r->ev_fill)
            if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : {r->ev_reset, r->ev_svc0, r->ev_svc1})
            if (ev) (void)hipEventDestroy(ev);
        for (auto &s : r->slots)
            for (auto &q : s.ready)
                if (q.dep) r->ev_pool.push_back(q.dep);
        for (auto ev : r->ev_pool) (void)hipEventDestroy(ev);
        for (DevBuf *b : {&r->arena, &r->ctl, &r->probe, &r->streams, &r->pending, &r->desc, &r->ldesc, &r->sdesc, &r->scalars, &r->tile_cnt, &r->tile_off,
                          &r->tile_slots, &r->tile_cnt2, &r->tile_slots2, &r->tileq, &r->scan_tmp, &r->dense, &r->segs, &r->seg_cnt, &r->seg_off, &r->recs, &r->seg_newc,
                          &r->seg_open, &r->seg_ecand_in, &r->seg_ecand, &r->inputs_dev})
            b->release();
        r->cells.release();
        r->heartbeat.release();
        r->free_fifo.release();
        r->inputs.release();
        for (auto &b : r->sugg_in) b.release();
        for (auto &b : r->piece_tab) b.release();
    }
    const bool unref = r->holds_engine_ref;
    delete r;
    if (e && unref) engine_unref(e);
}

int pbsgpu_ring_open(pbsgpu_ring *r, uint32_t *stream) {
    if (!r || !stream) return PBSGPU_E_INVALID;
    for (uint32_t i = 0; i < r->slots.size(); ++i)
        if (!r->slots[i].open) {
            r->slots[i] = StreamSlot{};
            r->slots[i].open = true;
            if (i < r->piece_n.size()) r->piece_n[i] = 0;  // (a piece table left behind by the slot's previous stream is not this one's)
            *stream = i;
            r->st.streams_opened++;
            return PBSGPU_OK;
        }
    return PBSGPU_E_BUSY;
}

int pbsgpu_ring_close(pbsgpu_ring *r, uint32_t stream) {
    if (!r || stream >= r->slots.size() || !r->slots[stream].open) return PBSGPU_E_INVALID;
    StreamSlot &s = r->slots[stream];
    if (!s.final_done || !s.cells.empty()) return PBSGPU_E_STATE;  // finish it and poll its records first
    s.open = false;
    return PBSGPU_OK;
}

int pbsgpu_ring_reserve(pbsgpu_ring *r, uint32_t stream, void **dptr, uint64_t *cap) {
    if (!r || !dptr || !cap || stream >= r->slots.size() || !r->slots[stream].open) return PBSGPU_E_INVALID;
    StreamSlot &s = r->slots[stream];
    if (s.final_committed || s.reserved >= 0) return PBSGPU_E_STATE;
    if (r->error != PBSGPU_OK) return r->error;
    uint32_t phys = 0;
    CHK(ring_take_page(r, &phys));
    s.reserved = phys;
    *dptr = r->arena.as<uint8_t>() + (uint64_t)phys * r->stride + 128u;
    *cap = r->page_bytes;
    return PBSGPU_OK;
}

int pbsgpu_ring_commit(pbsgpu_ring *r, uint32_t stream, uint64_t nbytes, int final) {
    return ring_commit_dep(r, stream, nbytes, final, nullptr);
}

int pbsgpu_ring_suggest(pbsgpu_ring *r, uint32_t stream, uint64_t offset) {
    if (!r || stream >= r->slots.size() || !r->slots[stream].open) return PBSGPU_E_INVALID;
    StreamSlot &s = r->slots[stream];
    if (s.final_committed) return PBSGPU_E_STATE;
    if (!s.sugg.empty() && offset < s.sugg.back()) return PBSGPU_E_INVALID;  // ascending
    // a boundary at or before bytes that are already in a round can no longer take part in those rounds' cuts: it must be
    // announced before the bytes around it are committed (the stream writer announces at the current position or ahead)
    s.sugg.push_back(offset);
    return PBSGPU_OK;
}

int pbsgpu_ring_fill(pbsgpu_ring *r, uint32_t stream, uint64_t seed, uint32_t kind, uint64_t nbytes, int final,
                     uint64_t *taken) {
    if (!r || !taken || stream >= r->slots.size() || !r->slots[stream].open || kind > 4) return PBSGPU_E_INVALID;
    StreamSlot &s = r->slots[stream];
    *taken = 0;
    if (s.final_committed || s.reserved >= 0) return PBSGPU_E_STATE;
    if (r->error != PBSGPU_OK) return r->error;
    if (s.bytes_committed + nbytes > pbsk::kRingMaxStream) return PBSGPU_E_INVALID;
    if (nbytes == 0) {
        if (final) {
            s.zero_final = true;
            s.final_committed = true;
        }
        return PBSGPU_OK;
    }
    while (*taken < nbytes) {
        const uint64_t n = std::min<uint64_t>(r->page_bytes, nbytes - *taken);
        const bool last = (*taken + n == nbytes);
        if (n < r->page_bytes && !(last && final)) break;  // a short page only as the stream's last one
        uint32_t phys = 0;
        if (ring_take_page(r, &phys) != PBSGPU_OK) break;   // no page free right now: the caller pumps and retries
        PageReq q;
        q.phys = phys;
        q.k = s.next_k++;
        q.valid = (uint32_t)n;
        q.final = last && final;
        q.do_fill = true;
        q.seed = seed;
        q.kind = kind;
        q.fill_off = s.bytes_committed;  // the generator's stream offset = the page's offset in its stream
        s.ready.push_back(q);
        r->ready_bytes += n;
        s.bytes_committed += n;
        *taken += n;
        if (q.final) s.final_committed = true;
    }
    return PBSGPU_OK;
}

// Synthetic producer for EDITED streams (BASELINE.json configs[4] through the ring): the stream's bytes are defined by a
// piece table over generator 4 — kept extents of a base file and newly written extents, in stream order. The first call
// hands the table over (it is copied; pieces ascending and contiguous from 0, offsets and lengths multiples of 16), later
// calls pass NULL / 0 and continue. Otherwise like pbsgpu_ring_fill.
int pbsgpu_ring_fill_pieces(pbsgpu_ring *r, uint32_t stream, const pbsgpu_fill_piece *pieces, uint32_t npieces, uint64_t nbytes,
                            int final, uint64_t *taken) {
    if (!r || !taken || stream >= r->slots.size() || !r->slots[stream].open) return PBSGPU_E_INVALID;
    static_assert(sizeof(pbsgpu_fill_piece) == sizeof(pbsk::FillPiece), "piece layout");
    StreamSlot &s = r->slots[stream];
    *taken = 0;
    if (s.final_committed || s.reserved >= 0) return PBSGPU_E_STATE;
    if (r->error != PBSGPU_OK) return r->error;
    if (pieces) {
        if (npieces == 0 || s.bytes_committed != 0) return PBSGPU_E_INVALID;  // the table comes with the stream's first bytes
        uint64_t pos = 0;
        for (uint32_t i = 0; i < npieces; ++i) {
            if (pieces[i].dst_off != pos || pieces[i].len == 0 || ((pieces[i].dst_off | pieces[i].len | pieces[i].src_off) & 15u))
                return PBSGPU_E_INVALID;
            pos += pieces[i].len;
        }
        CHK(r->piece_tab[stream].ensure(std::max<size_t>((size_t)npieces * sizeof(pbsk::FillPiece), 4096)));
        std::memcpy(r->piece_tab[stream].p, pieces, (size_t)npieces * sizeof(pbsk::FillPiece));
        r->piece_n[stream] = npieces;
    }
    const uint32_t nt = r->piece_n[stream];
    if (nt == 0) return PBSGPU_E_STATE;
    const pbsk::FillPiece *tab = r->piece_tab[stream].as<pbsk::FillPiece>();
    const uint64_t total = tab[nt - 1].dst_off + tab[nt - 1].len;
    if (s.bytes_committed + nbytes > total || s.bytes_committed + nbytes > pbsk::kRingMaxStream) return PBSGPU_E_INVALID;
    if (nbytes == 0) {
        if (final) {
            s.zero_final = true;
            s.final_committed = true;
        }
        return PBSGPU_OK;
    }
    while (*taken < nbytes) {
        const uint64_t n = std::min<uint64_t>(r->page_bytes, nbytes - *taken);
        const bool last = (*taken + n == nbytes);
        if (n < r->page_bytes && !(last && final)) break;
        uint32_t phys = 0;
        if (ring_take_page(r, &phys) != PBSGPU_OK) break;
        PageReq q;
        q.phys = phys;
        q.k = s.next_k++;
        q.valid = (uint32_t)n;
        q.final = last && final;
        q.do_fill = true;
        q.kind = 5;
        q.tab = tab;
        q.ntab = nt;
        q.fill_off = s.bytes_committed;
        s.ready.push_back(q);
        r->ready_bytes += n;
        s.bytes_committed += n;
        *taken += n;
        if (q.final) s.final_committed = true;
    }
    return PBSGPU_OK;
}

int pbsgpu_ring_pump(pbsgpu_ring *r) {
    if (!r) return PBSGPU_E_INVALID;
    CHK(set_device(r->eng));
    ring_heartbeat(r);
    CHK(ring_service_check(r, false));
    if (r->svc == SvcState::Stopping) {  // parked: has the kernel gone yet? (then parked frees can go ahead)
        const hipError_t q = hipEventQuery(r->ev_svc1);
        if (q == hipSuccess) ring_service_ended(r);
        else (void)hipGetLastError();
    }
    ring_reap_free(r);
    ring_reap_rounds(r);
    if (r->svc == SvcState::Running) {
        // parked frees on this device have passed their cap (dev_free): every ring lets go of its service once per request;
        // the last one to end frees the memory, the next round starts the service again
        const uint32_t gen = service_park_generation(r->eng->device);
        if (gen && gen != r->park_gen_seen) {
            r->park_gen_seen = gen;
            CHK(ring_park(r));
            if (r->svc == SvcState::Stopping) r->parked_for_flush = true;
        }
    }
    bool any_round = false;
    for (int i = 0; i < 4; ++i) {
        bool did = false;
        CHK(ring_enqueue_round(r, &did));
        if (!did) break;
        any_round = true;
    }
    if (!r->defer_service && r->svc == SvcState::Stopped && r->deferred_bytes > 0) {
        // rounds were cut ahead of the service (lone stream): start it once nothing more is waiting to be cut right now, or
        // the deferral has lasted long enough
        bool waiting = false;
        for (auto &s : r->slots) waiting |= s.open && !s.ready.empty();
        const uint64_t lanes_worth = (uint64_t)r->sha_cus * (r->dense_service ? 256u : 128u) * (uint64_t)r->eng->cfg.avg;
        // (Round 5 tried a 1 ms grace before "nothing waiting" ends the deferral — a feeder that commits a round's worth, pumps
        // and polls in a loop has nothing waiting after every pump, so its cut-ahead ends with the first poll: the file's last
        // bytes are cut ~7 ms sooner, its first chunks start ~25 ms later; one file alone 487-500 vs 496-502 ms on the same box,
        // the driver's line 589-592 vs 593-595: no gain, not kept. profiles/r05_ab_defer_grace.log)
        if ((!any_round && !waiting) || now_ms() - r->defer_t0 >= r->lone_defer_ms || r->deferred_bytes >= lanes_worth)
            CHK(ring_start_service(r));
    }
    if (r->autopark_ms > 0 && r->svc == SvcState::Running) {  // nothing anywhere in the ring: give the CUs (and hipFree) back
        if (ring_idle(r)) {
            const double t = now_ms();
            if (r->idle_since_ms == 0) r->idle_since_ms = t;
            else if (t - r->idle_since_ms >= r->autopark_ms) CHK(ring_park(r));
        } else {
            r->idle_since_ms = 0;
        }
    }
    return r->error;
}

int pbsgpu_ring_poll(pbsgpu_ring *r, uint32_t stream, pbsgpu_record *out, uint64_t cap, uint64_t *n, int *finished) {
    if (!r || !n || stream >= r->slots.size() || !r->slots[stream].open || (!out && cap)) return PBSGPU_E_INVALID;
    StreamSlot &s = r->slots[stream];
    ring_heartbeat(r);
    ring_reap_rounds(r);
    *n = 0;
    ring_pop_records(r, stream, out, cap, n);
    while (!r->rounds.empty() && r->rounds.front().reaped && r->rounds.front().live_cells == 0) r->rounds.pop_front();
    if (finished) *finished = (s.final_done && s.cells.empty()) ? 1 : 0;
    return r->error;
}

// Records of ANY open stream (each stream's in its own order), `segment` = stream id — for callers that drive hundreds
// or thousands of short streams (one per file) and cannot afford to ask every one of them after every pump. A stream
// that has ended and handed out its last record is reported once in `finished`.
int pbsgpu_ring_poll_any(pbsgpu_ring *r, pbsgpu_record *out, uint64_t cap, uint64_t *n, uint32_t *finished, uint32_t fcap,
                         uint32_t *nfinished) {
    if (!r || !n || !nfinished || (!out && cap) || (!finished && fcap)) return PBSGPU_E_INVALID;
    ring_heartbeat(r);
    ring_reap_rounds(r);
    *n = 0;
    *nfinished = 0;
    for (uint32_t si = 0; si < r->slots.size(); ++si) {
        StreamSlot &s = r->slots[si];
        if (!s.open || s.reported) continue;
        ring_pop_records(r, si, out, cap, n);
        if (s.final_done && s.cells.empty() && *nfinished < fcap) {
            finished[(*nfinished)++] = si;
            s.reported = true;
        }
    }
    while (!r->rounds.empty() && r->rounds.front().reaped && r->rounds.front().live_cells == 0) r->rounds.pop_front();
    return r->error;
}

// Diagnostic snapshot (text) of the device-side state: queue control words, per-page reference counts, stream states
// and what the host thinks. Safe while the service runs (plain copies on the null stream; the ring's streams are
// non-blocking).
int pbsgpu_ring_debug(pbsgpu_ring *r, char *buf, uint64_t cap) {
    if (!r || !buf || cap < 64) return PBSGPU_E_INVALID;
    CHK(set_device(r->eng));
    pbsk::RingCtl ctl{};
    std::vector<uint32_t> pend(r->npages);
    std::vector<pbsk::RingStreamState> sts(r->max_streams);
    HIPCHK(hipMemcpy(&ctl, r->ctl.p, sizeof(ctl), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pend.data(), r->pending.p, (size_t)r->npages * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sts.data(), r->streams.p, (size_t)r->max_streams * sizeof(pbsk::RingStreamState), hipMemcpyDeviceToHost));
    size_t o = 0;
    auto put = [&](const char *fmt, auto... a) {
        if (o + 1 < cap) o += (size_t)std::max(0, snprintf(buf + o, (size_t)(cap - o), fmt, a...));
        if (o >= cap) o = (size_t)cap - 1;
    };
    volatile uint32_t *hb = hb_words(r);
    put("ctl: tail=%u stop=%u head=%u ltail=%u lhead=%u free_count=%u error=%u rounds_done=%u | host: free_read=%u free_pages=%zu "
        "rounds=%zu next_seq=%u rounds_enq=%u service=%d tail_seen=%u claim_progress=%u backlog_limit=%llu intent=%u committed=%u\n",
        ctl.tail, ctl.stop, ctl.head, ctl.ltail, ctl.lhead, ctl.free_count, ctl.error, ctl.rounds_done, r->free_read,
        r->free_pages.size(), r->rounds.size(), r->next_seq, r->rounds_enq, (int)r->svc, r->tail_seen, hb[pbsk::kHbClaim],
        (unsigned long long)r->backlog_limit, hb[pbsk::kHbIntent], hb[pbsk::kHbCommitted]);
    {
        unsigned long long pr[16] = {};
        HIPCHK(hipMemcpy(pr, r->probe.p, sizeof(pr), hipMemcpyDeviceToHost));
        put("lanes service: cus=%u short_bytes=%u stail=%u shead=%u steps_sampled=%llu ns_per_block_step=%.1f lanes_busy=%.3f | pair cus=%u "
            "lanes_busy=%.3f (probe wave, intervals with a block in every step) express cus=%u\n",
            r->lanes_cus, r->short_bytes, ctl.stail, ctl.shead, pr[6], pr[6] ? (double)pr[7] * 10.0 / (double)pr[6] : 0.0,
            pr[6] ? (double)pr[9] / (64.0 * (double)pr[6]) : 0.0, r->sha_cus, pr[0] ? (double)pr[8] / (64.0 * (double)pr[0]) : 0.0, r->xp_cus);
        for (int cls = 0; cls < 2; ++cls)
            if (r->ctl_phase_rounds[cls])
                put("control kernel, %s rounds: %llu rounds, us per round: tiles+compaction %.1f, resolve %.1f, numbering %.1f, records %.1f, publish %.1f\n",
                    cls ? "large (>= 100 pages)" : "small (< 100 pages)", (unsigned long long)r->ctl_phase_rounds[cls],
                    r->ctl_phase_ticks[cls][0] * 0.01 / r->ctl_phase_rounds[cls], r->ctl_phase_ticks[cls][1] * 0.01 / r->ctl_phase_rounds[cls],
                    r->ctl_phase_ticks[cls][2] * 0.01 / r->ctl_phase_rounds[cls], r->ctl_phase_ticks[cls][3] * 0.01 / r->ctl_phase_rounds[cls],
                    r->ctl_phase_ticks[cls][4] * 0.01 / r->ctl_phase_rounds[cls]);
        put("probe raw: pair_steps=%llu pair_active=%llu lanes_steps=%llu lanes_active=%llu\n", pr[0], pr[8], pr[6], pr[9]);
    }
    uint32_t nz = 0;
    for (uint32_t p = 0; p < r->npages && nz < 64; ++p)
        if (pend[p]) {
            put("page %u pending=%u\n", p, pend[p]);
            ++nz;
        }
    for (uint32_t i = 0; i < r->slots.size(); ++i) {
        const StreamSlot &s = r->slots[i];
        if (!s.open) continue;
        put("stream %u: committed=%llu enqueued=%llu final_committed=%d final_enqueued=%d final_done=%d ready=%zu cells=%zu "
            "out=%llu | device c=%llu end=%llu\n", i, (unsigned long long)s.bytes_committed,
            (unsigned long long)s.bytes_enqueued, (int)s.final_committed, (int)s.final_enqueued, (int)s.final_done,
            s.ready.size(), s.cells.size(), (unsigned long long)s.records_out, (unsigned long long)sts[i].c,
            (unsigned long long)sts[i].end);
        if (!s.cells.empty()) {
            const uint8_t *c = r->cells.as<uint8_t>() + (size_t)s.cells.front().cell * 64;
            pbsgpu_record rec;
            std::memcpy(&rec, c, sizeof(rec));
            put("  waiting for cell %u: end=%llu size=%u flag=%u\n", s.cells.front().cell, (unsigned long long)rec.end, rec.size,
                *reinterpret_cast<const uint32_t *>(c + 48));
        }
    }
    return PBSGPU_OK;
}

int pbsgpu_ring_express(pbsgpu_ring *r, uint32_t *express_cus, uint64_t *long_bytes) {
    if (!r) return PBSGPU_E_INVALID;
    if (express_cus) *express_cus = r->xp_cus;
    if (long_bytes) *long_bytes = r->xp_cus ? r->long_bytes : 0;
    return PBSGPU_OK;
}

int pbsgpu_ring_get_probe(pbsgpu_ring *r, pbsgpu_ring_probe *out) {
    if (!r || !out) return PBSGPU_E_INVALID;
    static_assert(sizeof(pbsgpu_ring_probe) == 48, "six counters");
    // While a service runs the answer comes from the newest reaped round's status block (k_ring_control copies the counters
    // there, mapped pinned: at most one round — about a millisecond — old; the probe waves sample every ~7 ms): no HIP call
    // of the host beside a persistent kernel. (Until the end of round 6 this was a hipMemcpy on the null stream inside
    // bench.py's timed region; pbsgpu_ring_debug, which still copies that way, was seen to stall for the service's idle
    // timeout in 2 of 6 diagnostic runs.) With the service stopped the device counters are read directly: exact.
    if (r->svc == SvcState::Running) {
        ring_reap_rounds(r);
        if (r->probe_seen_valid) {
            std::memcpy(out, r->probe_seen, sizeof(*out));
            return PBSGPU_OK;
        }
    }
    CHK(set_device(r->eng));
    HIPCHK(hipMemcpy(out, r->probe.p, sizeof(*out), hipMemcpyDeviceToHost));
    std::memcpy(r->probe_seen, out, sizeof(*out));  // (the counters only grow: a later answer from a status block is never older)
    r->probe_seen_valid = true;
    return PBSGPU_OK;
}

int pbsgpu_ring_get_stats(pbsgpu_ring *r, pbsgpu_ring_stats *out) {
    if (!r || !out) return PBSGPU_E_INVALID;
    ring_reap_free(r);
    r->st.pages_free = (uint32_t)r->free_pages.size();
    r->st.rounds_in_flight = 0;
    for (auto &ri : r->rounds) r->st.rounds_in_flight += ri.reaped ? 0 : 1;
    *out = r->st;
    return PBSGPU_OK;
}

}  // extern "C"
