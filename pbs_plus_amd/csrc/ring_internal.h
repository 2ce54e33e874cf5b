// The page ring's host-side state (ring.cpp), shared with the payload-stream writer (stream.cpp), which is a CLIENT of an
// engine-owned ring: pbsgpu_stream_* = staging + H2D into reserved pages + the bookkeeping of sections, tees and entries;
// cutting, hashing, page release and record delivery are the ring's. Not part of the C ABI.
#pragma once

#include <deque>
#include <mutex>
#include <vector>

#include "engine_internal.h"

namespace pbse {

constexpr uint32_t kRingInputs = 16;            // rounds whose host-written tables may be in flight
constexpr uint32_t kPagesPerStreamRound = 256;  // < kRingPT - 2 (open chunk). It was 48 until a lone 64 GiB file turned out to
                                                // go through 83 small rounds, none of them large enough for the cut-ahead at
                                                // full chip width (ring_enqueue_round): a whole default round per stream now

struct PageReq {                          // a committed page waiting for its round
    uint32_t phys = 0;
    uint64_t k = 0;                       // logical page index in its stream
    uint32_t valid = 0;
    bool final = false;
    bool do_fill = false;
    uint64_t seed = 0, fill_off = 0;
    uint32_t kind = 0;
    const pbsk::FillPiece *tab = nullptr;  // kind 5: the stream's piece table
    uint32_t ntab = 0;
    hipEvent_t dep = nullptr;             // the page's bytes are there once this event has completed (host-fed pages: the
                                          // H2D copy, or the XXH3 tee behind it); from the ring's event pool, returned at enqueue
};

struct CellRef {
    uint32_t cell;
    uint32_t round_idx;                   // RoundInfo::seq of the round that produced it
};

struct StreamSlot {
    bool open = false;
    bool fresh = true;                    // no round has carried this stream yet (device state starts from zero)
    uint64_t next_k = 0;                  // next logical page
    uint64_t bytes_committed = 0;
    uint64_t bytes_enqueued = 0;          // stream length after the rounds enqueued so far
    bool final_committed = false, final_enqueued = false, final_done = false;
    bool zero_final = false;              // final commit of 0 bytes still to be carried by a round
    int64_t reserved = -1;                // physical page handed out by reserve
    int64_t last_phys = -1;               // physical page of the last page that went into a round (the next page's head pad source)
    std::deque<PageReq> ready;
    std::deque<CellRef> cells;            // record cells in stream order (round results reaped)
    uint64_t records_out = 0;
    bool reported = false;                // poll_any has announced the end of this stream
    uint64_t origin = 0;                  // payload position of the stream's byte 0 (absolute reader grid of suggested boundaries)
    std::deque<uint64_t> sugg;            // suggested boundaries still of interest (stream offsets, ascending)
    void *owner = nullptr;                // the stream writer's section that feeds this slot (stream.cpp)
};

struct RoundInfo {
    uint32_t seq = 0;                     // round number + 1
    uint32_t input = 0;
    uint64_t cell_base = 0;               // monotonic
    uint32_t cell_cap = 0;
    uint32_t live_cells = 0;              // cells handed to streams and not yet polled
    bool reaped = false;
    std::vector<uint32_t> finals;         // slots whose stream ended with this round
    uint64_t new_bytes = 0;               // stream bytes this round added
};

enum class SvcState { Stopped, Running, Stopping };

}  // namespace pbse

struct pbsgpu_ring {
    pbsgpu_engine *eng = nullptr;
    bool holds_engine_ref = true;         // false for the engine's own ring (the engine owns it, not the reverse)
    std::mutex mu;                        // taken by the stream writer around every use (the C ABI's ring calls are
                                          // single-threaded by contract and do not lock)
    // geometry
    uint64_t page_bytes = 0, stride = 0;
    uint32_t tile_bytes = 0, tpp = 0, npages = 0, max_streams = 0, sha_cus = 0, round_pages = 0, min_round_pages = 0, cap = 0;
    uint32_t max_inflight = 3;
    // backlog gate: no page is handed out while more than this many bytes wait in front of the service — committed pages
    // not yet in a round, rounds in flight, published chunks no lane has claimed. A full arena of unhashed chunks feeds
    // the service no faster than a short queue does; it only adds its length to every latency (and to the final drain).
    uint64_t backlog_limit = 0;
    uint64_t ready_bytes = 0, inflight_bytes = 0;
    uint32_t tail_seen = 0;               // queue tail after the last reaped round
    uint64_t pub_positions = 0, pub_bytes = 0;  // queue positions / stream bytes of all reaped rounds (average chunk size)
    uint64_t rec_cap = 0, dense_cap = 0;
    uint32_t qslots = 0, ncells = 0, nfree = 0;
    // device
    pbse::DevBuf arena, ctl, streams, pending, desc, ldesc, sdesc, probe;
    unsigned long long probe_seen[6] = {};  // the probe counters as the newest reaped round reported them (pbsgpu_ring_get_probe)
    bool probe_seen_valid = false;
    unsigned long long ctl_phase_ticks[2][5] = {};  // k_ring_control's phase times summed over the reaped rounds (small / large rounds)
    unsigned long long ctl_phase_rounds[2] = {};
    uint32_t lslots = 0, long_bytes = 0;
    uint32_t sslots = 0, short_bytes = 0, lanes_cus = 0;  // the LANES service and its short-chunk queue (kernels.h: RingSource::sdesc)
    pbse::DevBuf scalars, tile_cnt, tile_off, tile_slots, scan_tmp, dense, segs, seg_cnt, seg_off, recs, seg_newc, seg_open;
    pbse::DevBuf tile_cnt2, tile_slots2, tileq;  // second set of the scan side (rounds alternate) + the two tile-queue counters
    pbse::DevBuf seg_ecand_in, seg_ecand;
    pbse::DevBuf inputs_dev;  // device mirror of `inputs` (the round tables are staged once per round: k_ring_stage)
    bool stage_inputs = true;
    bool long_lo_auto = false;  // RingSource::long_lo from the chunker's maximum (bulk rings)
    // mapped pinned
    pbse::PinnedBuf cells, free_fifo, inputs, heartbeat;
    std::vector<pbse::PinnedBuf> piece_tab;      // per stream slot: the piece table of a synthetic edited stream (fill_pieces)
    std::vector<uint32_t> piece_n;
    pbse::PinnedBuf sugg_in[pbse::kRingInputs];  // suggested offsets of the round built in input i (grown on demand)
    size_t input_stride = 0, in_pages_off = 0, in_segs_off = 0, in_recbase_off = 0, in_suggidx_off = 0, in_status_off = 0;
    hipStream_t cs = nullptr, ss = nullptr, fs = nullptr;  // cut rounds, SHA service, synthetic producer
    hipStream_t ls = nullptr;             // the LANES service when lanes_cus > 0
    hipStream_t xs = nullptr;             // the EXPRESS service (two lanes per chunk, long chunks only) when xp_cus > 0
    hipStream_t ps = nullptr;             // scan side of the cut rounds (head pads + scan): round n + 1 is scanned while round n's
                                          // control kernel runs on cs (PBSGPU_RING_OVERLAP=0: everything on cs)
    hipEvent_t ev_scan[pbse::kRingInputs] = {};  // scan of the round built in input i done
    hipEvent_t ev_ctl[2] = {};            // control kernel of the last round that used scan set 0 / 1 done
    bool ctl_used[2] = {false, false};
    uint32_t scan_set = 0;                // scan set of the next round
    int bulk_prio = 0;                    // HIP priority of the refill and scan streams (lowest: the control side goes first)
    uint32_t xp_cus = 0;
    // The split between the two services follows the DATA (round 5): the share of the published bytes that sits in long
    // chunks (>= long_bytes) is observed as rounds are reaped, and every service START — the ring was idle, nothing is in
    // flight, the change is free — picks the express share that balances the two services' CU-time for that share
    // (ring_adapt_split). Only when neither the options nor the environment fixed the counts.
    bool split_auto = false;
    uint32_t svc_cus = 0;                 // CUs of both services together (constant)
    double obs_bytes = 0, obs_long_bytes = 0;
    hipEvent_t ev_reset = nullptr, ev_svc0 = nullptr, ev_svc1 = nullptr, ev_xsvc1 = nullptr, ev_lsvc1 = nullptr;
    hipEvent_t ev_fill[pbse::kRingInputs] = {};
    std::vector<hipEvent_t> ev_pool;      // page dependency events (ring_event_get / ring_event_put)
    pbse::SvcState svc = pbse::SvcState::Stopped;
    uint32_t rounds_enq = 0;              // mirrored into the heartbeat block for the service's self-stop handshake
    double autopark_ms = 0;               // > 0: stop the service when the ring has been idle this long (engine ring of the stream writer)
    double idle_since_ms = 0;
    bool dense_service = false;           // PBSGPU_RING_F_DENSE_SERVICE
    bool dense_lanes = false;             // PBSGPU_RING_F_DENSE_LANES
    bool tier_tag = false;                // PBSGPU_RING_F_TIER_TAG
    bool backlog_auto = true;             // backlog_limit derived from sha_cus (follows ring_adapt_split)
    uint32_t park_gen_seen = 0;           // last graveyard park request this ring honoured (engine_internal.h: dev_free)
    uint32_t park_grace_gen = 0;          // the park request this ring has already waited its grace period for (ring_start_service)
    double park_wait_t0 = 0;              // since when a start has been waiting for the other rings of the device to let go (ring_start_service)
    bool parked_for_flush = false;        // ... and its service was parked for it: the next start waits for that service's END
    bool fill_serial = false;             // PBSGPU_RING_F_FILL_SERIAL
    uint32_t opt_long_lo = 0, opt_long_spill = 0, opt_poll_every = 0;  // pbsgpu_ring_options (0 = the default rule)
    bool defer_service = false;           // PBSGPU_RING_F_DEFER_SERVICE (profiling): rounds only enqueue; quiesce runs the service ALONE
    double lone_defer_ms = 25.0;          // a lone bulk stream's rounds are cut ahead of the service start for at most this long (0 = off)
    double defer_t0 = 0;                  // when the current deferral began (0 = none)
    uint64_t deferred_bytes = 0;          // bytes cut while no service was running (they are the next launch's work)
    double idle_timeout_s = 0;            // > 0: the service's own idle stop (pbsgpu_ring_options::idle_timeout_s; default 20 s)
    // host bookkeeping
    std::vector<uint32_t> free_pages;
    uint32_t free_read = 0;               // entries of the free FIFO consumed
    std::vector<pbse::StreamSlot> slots;
    std::deque<pbse::RoundInfo> rounds;   // enqueued, oldest first; popped when reaped AND all their cells were polled
    bool input_busy[pbse::kRingInputs] = {};
    uint32_t next_seq = 1;
    uint64_t cell_cursor = 0;
    int error = PBSGPU_OK;
    pbsgpu_ring_stats st{};
    double svc_t0 = 0;
    uint64_t svc_bytes0 = 0;

    uint8_t *in(uint32_t i) const { return inputs.as<uint8_t>() + (size_t)i * input_stride; }
    uint8_t *in_dev(uint32_t i) const { return inputs_dev.as<uint8_t>() + (size_t)i * input_stride; }
    pbsk::RingPage *in_pages(uint32_t i) const { return reinterpret_cast<pbsk::RingPage *>(in(i) + in_pages_off); }
    pbsk::RingSeg *in_segs(uint32_t i) const { return reinterpret_cast<pbsk::RingSeg *>(in(i) + in_segs_off); }
    uint32_t *in_recbase(uint32_t i) const { return reinterpret_cast<uint32_t *>(in(i) + in_recbase_off); }
    uint32_t *in_suggidx(uint32_t i) const { return reinterpret_cast<uint32_t *>(in(i) + in_suggidx_off); }
    pbsk::RingRoundStatus *in_status(uint32_t i) const {
        return reinterpret_cast<pbsk::RingRoundStatus *>(in(i) + in_status_off);
    }
    pbsk::RingSource source() const;
};

namespace pbse {

// ring.cpp internals the stream writer uses (caller holds ring->mu)
// (long_bytes_hint: from which chunk size a chunk takes the express service; 0 = the default 13/16 of the maximum)
int ring_create_internal(pbsgpu_engine *e, const pbsgpu_ring_options *opt, bool hold_engine_ref, pbsgpu_ring **out,
                         uint32_t long_bytes_hint = 0);
// an event from the ring's pool; record it behind the page's last copy / tee and hand it to ring_commit_dep
int ring_event_get(pbsgpu_ring *r, hipEvent_t *ev);
void ring_event_put(pbsgpu_ring *r, hipEvent_t ev);
// pbsgpu_ring_commit with a dependency: the page's bytes are there once `dep` has completed (nullptr = they are now)
int ring_commit_dep(pbsgpu_ring *r, uint32_t stream, uint64_t nbytes, int final, hipEvent_t dep);
// records of one stream that are ready, in order, appended to out[*n ..) (segment = slot)
void ring_pop_records(pbsgpu_ring *r, uint32_t slot, pbsgpu_record *out, uint64_t cap, uint64_t *n);
// stop the service behind everything enqueued so far WITHOUT waiting for it; the next round starts it again
int ring_park(pbsgpu_ring *r);
bool ring_idle(const pbsgpu_ring *r);

}  // namespace pbse
