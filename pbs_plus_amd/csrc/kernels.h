// Internal launch interface between the engine (host logic) and the gfx950 kernels.
// Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pbsgpu.h"

namespace pbsk {

// ---- Buzhash candidate scan geometry -------------------------------------------------
// One wave owns one "wave tile" of 64 lanes x (LINES x 128) bytes (k_scan3: LINES = 34 -> 278 528 B, 4 -> 32 768 B).
constexpr int kWindow = 64;

struct ScanParams {
    const uint8_t *data_al;   // 16-byte aligned base (<= caller's pointer)
    uint32_t lead;            // caller's pointer - data_al (0..15)
    uint64_t nbytes;          // caller's byte count
    uint64_t ntiles;          // wave tiles covering [0, nbytes + lead)
    uint32_t tile_bytes;      // bytes per wave tile (selects the kernel variant)
    const uint32_t *table_rot;// device: table pre-rotated left by (32 - bits)
    uint32_t thr;             // break_min << (32 - bits)
    uint32_t cap;             // slots per tile
    uint32_t *tile_cnt;       // [ntiles] true count (may exceed cap)
    uint32_t *tile_slots;     // [ntiles * cap] end offset within tile (1..tile_bytes), a-coords
    unsigned long long *tile_queue;  // device counter, zero at launch: next tile to hand out
    uint32_t tiles_per_wave;         // k_scan3: 0 = persistent workgroups, else a wave retires after this many tiles
    uint32_t shared_chip;            // other batches of the engine are in flight (their SHA chains are running)
    // page-ring rounds (k_scan3 only): tile t lies in page entry t / ring_tpp at tile index t % ring_tpp of that page;
    // data_al = arena base. Null for flat byte ranges.
    const struct RingPage *ring_pages;
    uint32_t ring_tpp;
    uint32_t max_blocks;             // 0 = no extra limit on the launch's workgroups
};

hipError_t launch_scan(const ScanParams &p, int num_cus, hipStream_t st);
// tile size the scan will use for a batch of nbytes (kernel variant is chosen from it)
uint32_t scan_tile_bytes(uint64_t nbytes);

// exclusive scan of min(in[i], clamp) -> out[i]; *total = sum; *maxval = max(in[i]) (atomicMax'd)
// tmp must hold at least scan_tmp_words(n) uint32.
size_t scan_tmp_words(uint64_t n);
hipError_t launch_exclusive_scan(const uint32_t *in, uint64_t n, uint32_t clamp, uint32_t *out,
                                 uint32_t *total, uint32_t *maxval, uint32_t *tmp, hipStream_t st);

// dense, ascending candidate END offsets (caller coordinates) from the per-tile slots
hipError_t launch_compact(const uint32_t *tile_cnt, const uint32_t *tile_off, const uint32_t *tile_slots,
                          uint32_t cap, uint64_t ntiles, uint32_t lead, uint64_t nbytes, uint64_t *dense,
                          uint64_t dense_cap, uint32_t tile_bytes, hipStream_t st);

// one long stream without suggested boundaries: the cut chain followed by pointer doubling (kernels.hip); `scratch` holds
// resolve_par_scratch_bytes(node_cap, levels); falls back to the serial walk when there are more candidates than node_cap - 1
size_t resolve_par_scratch_bytes(uint32_t node_cap, uint32_t levels);
hipError_t launch_resolve_single_par(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs, uint32_t effmin,
                                     uint32_t maxsz, const uint32_t *zero_off, uint32_t *nrec, pbsgpu_record *recs,
                                     uint64_t rec_cap, void *scratch, uint32_t node_cap, uint32_t levels, uint32_t *fallback,
                                     hipStream_t st);

// ... the same for many candidates (small average chunk sizes): one grid-wide launch per phase / doubling level
hipError_t launch_resolve_single_par_grid(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                          uint32_t effmin, uint32_t maxsz, const uint32_t *zero_off, uint32_t *nrec,
                                          pbsgpu_record *recs, uint64_t rec_cap, void *scratch, uint32_t node_cap, uint32_t levels,
                                          uint32_t *fallback, uint32_t *hops, hipStream_t st);

// optional suggested boundaries (payload chunker): offsets[index[s] .. index[s+1]) ascending, relative to segment s
struct Suggested {
    const uint64_t *offsets = nullptr;  // device
    const uint32_t *index = nullptr;    // device, nseg + 1 entries
    uint32_t cmin = 0;                  // the config's true min (a suggested cut needs chunk_size >= min, not >= 65)
    // How many bytes one `scan` call of the reference's payload chunker sees (its reader's buffer): a suggested boundary
    // that lies in the CURRENT buffer is taken without running the hash scan on that buffer, i.e. it pre-empts an earlier
    // hash cut of the same buffer; a hash cut found in an earlier buffer still wins. feed <= 1 = byte-serial feed (the
    // feed-independent limit: the earlier position wins); ~0 = the whole rest in one call. `absolute`: buffers end at
    // multiples of `feed` from the STREAM start (a reader that appends fixed-size reads to its buffer; `origin` = stream
    // offset of the segment start) instead of restarting at every cut (oracle_chunk_stream_suggested's feeding loop).
    uint64_t feed = 1;
    uint64_t origin = 0;
    uint32_t absolute = 0;
    // the segment's end is NOT the stream's end (a window of the stream writer): a known boundary that lies beyond the
    // bytes seen so far but in the same reader buffer as a hash cut still pre-empts that cut — the walk stops there and
    // the rest is the open chunk the next window re-examines
    uint32_t open_end = 0;
};
struct SuggFeed {
    uint64_t feed, origin;
    uint32_t absolute, open_end;
};
// Page-ring rounds (ring_kernels.inc): segments are the streams' open chunks + new pages in LOGICAL coordinates
// ((slot << kRingOffBits) | offset), suggested offsets are relative to the stream's byte 0, a segment's end is the stream's end only
// when its RingSeg says final, and the walk reports per segment what the round leaves behind (all null otherwise).
// logical coordinates of a ring round: (stream slot << kRingOffBits) | offset. A ring has at most 4096 stream slots (12 bits),
// which leaves 52 bits = 4 PiB per stream (rounds 3-4: 40 bits = 1 TiB, which a payload stream without a forced cut — a
// fresh multi-TiB backup, a tape conversion — could exceed)
constexpr unsigned kRingOffBits = 52;
constexpr uint64_t kRingOffMask = (1ull << kRingOffBits) - 1ull;
struct RingPage;
// Exact handling of candidate-DENSE scan tiles (periodic / crafted data). A tile records at most `cap` of its candidates; a
// tile that found more (tile_cnt[t] > cap) leaves an arbitrary SUBSET in the dense list — every entry a true candidate, some
// missing. The cut rule only ever needs ONE thing from such a tile: the first candidate at or behind a given position. So the
// resolve walk re-scans on demand (kernels.hip: dense_refine / dense_first_hit): whenever the stretch it is about to skip —
// [s + effmin, first listed candidate) — touches an overflowed tile, one wave computes the window hashes of that part of the
// tile (64 lanes x 64 positions per step, the scan's own pre-rotated table and threshold) and takes the first hit. Nothing
// fails and nothing is re-run: the reference's writer never fails on byte content either
// (transfer.ArchiveWriter.WriteEntryReader: internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836).
struct DenseTiles {
    const uint32_t *tile_cnt = nullptr;  // true candidate count per tile; null = no tile of this launch overflowed
    uint32_t cap = 0, tile_bytes = 0;
    const uint32_t *table_rot = nullptr;
    uint32_t thr = 0;
    const uint8_t *base = nullptr;       // flat byte range: caller byte x at base[x] (any alignment); page ring: the arena
    const RingPage *pages = nullptr;     // page ring: the round's page table (tile t lies in entry t / tpp)
    uint32_t tpp = 0;
};
// where the tiles of ONE segment's walk lie: END offset E (the walk's coordinates) belongs to tile
// first_tile + (E - 1 - L0 + lead) / tile_bytes, for E > L0 (flat range: L0 = 0, lead = caller pointer & 127, the scan's
// tiles start at the 128-byte line of byte 0; page ring: L0 = logical offset of the segment's first NEW page, lead = 0)
struct DenseSeg {
    uint64_t L0 = 0, first_tile = 0, ntiles = 0;
    uint32_t lead = 0;
};

// min/max resolution, one wave per segment. count pass -> seg_cnt; write pass -> recs[seg_off[s] + k]
// (`dz`: the batch was scanned at the capacity limit and some tile overflowed — the walks consult the tiles, see DenseTiles;
// `maxcnt` = the device word that holds the largest tile count, `lead` = caller pointer & 127)
hipError_t launch_resolve_count(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                uint32_t nseg, uint32_t effmin, uint32_t maxsz, uint32_t *seg_cnt,
                                const Suggested &sg, hipStream_t st, const DenseTiles *dz = nullptr, uint32_t lead = 0,
                                uint64_t ntiles = 0);
hipError_t launch_resolve_write(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                uint32_t nseg, uint32_t effmin, uint32_t maxsz, const uint32_t *seg_off,
                                pbsgpu_record *recs, uint64_t rec_cap, const Suggested &sg, hipStream_t st,
                                const DenseTiles *dz = nullptr, uint32_t lead = 0, uint64_t ntiles = 0);

hipError_t launch_resolve_single(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                 uint32_t effmin, uint32_t maxsz, const uint32_t *zero_off, uint32_t *nrec,
                                 pbsgpu_record *recs, uint64_t rec_cap, const Suggested &sg, hipStream_t st,
                                 const DenseTiles *dz = nullptr, uint32_t lead = 0, uint64_t ntiles = 0);

// SHA-256 of every record's chunk: one lane per chunk, lanes pull records from a shared queue.
// `queue` is a device uint32 that must be zero at launch.
// (`form`: pbsgpu_engine_options::sha_form — 0 wave pairs, 1 single-wave lanes, 2 express)
hipError_t launch_sha256_records(pbsgpu_record *recs, const uint32_t *nrec, uint32_t *queue, const uint4 *qdesc,
                                 const uint32_t *wg_limit, int num_cus, bool dense, int form, hipStream_t st);
// longest-first queue order (counting sort by size class) + workgroup budget for the SHA kernel:
// lanes = (1 + slack_pct/100) x total blocks / longest chunk's blocks
// (writes the queue as 16-byte descriptors {address lo, hi, size, record index}: kQueueDescBytes per record)
constexpr size_t kQueueDescBytes = 16;
hipError_t launch_order(const uint8_t *data, const pbsgpu_segment *segs, const pbsgpu_record *recs, const uint32_t *nrec,
                        uint32_t max_chunk, uint4 *qdesc, uint32_t *wg_limit, int num_cus, const uint32_t *maxcnt, uint32_t cap, uint32_t slack_pct,
                        hipStream_t st);
// true when a hash launch of `total_blocks` 64-byte blocks whose longest item has `longest_blocks` is bound by issue
// slots rather than by that longest chain (the device-side twin of this test lives in k_order)
bool sha256_dense_pays(uint64_t total_blocks, uint64_t longest_blocks, int num_cus, uint32_t dense_pct);
// SHA-256 of whole segments (verification path): digests[32*i] for segs[i]
hipError_t launch_sha256_segments(const uint8_t *data, const pbsgpu_segment *segs, uint32_t nseg,
                                  uint8_t *digests, uint32_t *queue, int num_cus, bool dense, int form, hipStream_t st);

// XXH3-64 (seed 0) of whole segments: out[i] for segs[i]; `queue` zero at launch
// XXH3-64 (seed 0), two phases (kernels.hip): every 1 KiB block of every input is summed by some wave of the grid
// (k_xxh3_sums -> `sums`, 64 bytes per block), then one wave per input runs the short serial scramble chain + tail.
// Work items: flags bit0 = first piece of its input, bit1 = last piece (both = a whole input); pieces of one input carry
// state in states[state] and must be launched in order on one stream; the hash of a finished input lands in out[out].
// pend / nproc / s_off are the host's plan: bytes pending in front of the piece, full blocks to consume now, index of the
// item's first entry in `sums` (items in s_off order).
struct XxhItem {
    const uint8_t *ptr;
    uint64_t len;
    uint32_t flags;
    uint32_t state;
    uint32_t out;
    uint32_t pend;
    uint64_t s_off;
    uint32_t nproc;
    uint32_t pad;
};
uint64_t xxh3_plan_whole(XxhItem *items, uint32_t n);  // plan for whole inputs; returns the total block count
size_t xxh3_state_bytes();
hipError_t launch_xxh3_items(const XxhItem *items, uint32_t nitems, uint64_t total_blocks, void *states, uint64_t *sums,
                             uint64_t *out, uint32_t *queue, int num_cus, hipStream_t st);

// device -> mapped pinned host memory by kernel// device -> mapped pinned host memory by kernel (never through the shared SDMA copy queues; see kernels.hip)
hipError_t launch_publish(void *dst_host_mapped, const void *src, uint64_t nbytes, hipStream_t st);
hipError_t launch_publish_records(pbsgpu_record *dst_host_mapped, const pbsgpu_record *src, const uint32_t *nrec,
                                  uint64_t cap, hipStream_t st);

hipError_t launch_fill(void *dptr, uint64_t stream_off, uint64_t nbytes, uint64_t seed, uint32_t kind,
                       hipStream_t st);

// ---- page ring (ring.cpp): persistent SHA-256 service + cut rounds over non-adjacent pages ----------------------
struct alignas(64) RingCtl {   // device memory, one 64-byte line
    union {
        struct {
            uint32_t tail;     // positions < tail are published
            uint32_t stop;     // nothing will be published beyond tail
        };
        unsigned long long tail_stop;  // ... read together as one 64-bit word by the service's lanes
    };
    union {
        struct {
            uint32_t ltail;    // LONG-chunk queue: positions < ltail are published
            uint32_t lhead;    // ... positions < lhead have been taken (CAS: never runs ahead of ltail)
        };
        unsigned long long lq;
    };
    uint32_t head;         // next queue position to hand out
    uint32_t free_count;   // pages reported free so far
    uint32_t error;        // sticky, ring-wide: a round overflowed its record / cell capacity (the host's bound was wrong)
    uint32_t rounds_done;  // rounds published so far (k_ring_publish); the service compares it with the host's count of
                           // enqueued rounds before it stops on its own
    uint32_t xp_busy;      // express service: lane pairs that hold a chunk right now (the pair service's lanes take a long
                           // chunk only while every express pair is busy: waiting for one would cost more than it saves)
    uint32_t pad0;
    union {
        struct {
            uint32_t stail;    // SHORT-chunk queue (the lanes service, RingSource::sdesc): positions < stail are published
            uint32_t shead;    // ... positions < shead have been taken (CAS, like lhead)
        };
        unsigned long long sq;
    };
    uint32_t pad[4];
};
static_assert(sizeof(RingCtl) == 64, "RingCtl is one 64-byte line");
struct RingSource {
    static constexpr bool kRing = true;
    const uint4 *desc;         // ring of positions, 2 x uint4 each: {p1.lo, p1.hi, len, len1} {p2v.lo, p2v.hi, cell, pages}
    uint32_t qmask;            // positions - 1 (power of two)
    // Which regime did a run land in? ONE producer wave of each service (workgroup 0, first producer) samples clock64() (the
    // shader clock) and wall_clock64() (100 MHz) once per kRingProbeSteps block steps and, if it carried a block in EVERY step
    // of the interval, adds {steps, shader cycles, wall ticks} to these device counters (3 x u64 per service: pair [0..2],
    // express [3..5]): ns per block step of a chain under load and the shader clock it ran at (pbsgpu_ring_get_probe). One
    // wave only: counting steps in EVERY producer wave cost the driver's line 1.2 % (profiles/r06_ab_walk_load_in_branch.log:
    // the producer bounds the loaded chain, every instruction of its step shows); the others pay one scalar branch.
    unsigned long long *probe;
    // Chunks of at least `long_bytes` go through a second, smaller queue that idle lanes look at FIRST: a max-size chunk
    // hashes for ~0.45 s on one lane, so the later it starts the longer the ring's drain (bench: the last such chunk used
    // to start behind ~0.3 s of queued short chunks). Lanes never wait on this queue (compare-and-swap on lhead only
    // while lhead < ltail); a lane may take a long chunk while its claim on the main queue stays valid.
    const uint4 *ldesc;
    uint32_t lmask;
    uint32_t long_bytes;       // 0 = no second queue
    // Chunks of at most `short_bytes` may go through a THIRD queue, served by the LANES service (k_sha256_lanes: one lane per
    // chunk, schedule and rounds in one wave, four independent waves per CU: 81 chain-blocks per us and CU against the pair
    // form's 74, every chain at 3.2 instead of 1.73 us per block — profiles/r06_sha_forms_full_lanes.log). The control kernel
    // sends a short chunk there only while fewer than `short_room` entries wait (the lanes are kept busy, nothing queues up
    // behind them); everything else stays with the pair service. CAS queue like the long one: lanes never wait at a position.
    const uint4 *sdesc;
    uint32_t smask;
    uint32_t short_bytes;      // 0 = no lanes service
    uint32_t short_room;
    RingCtl *ctl;
    uint8_t *cells;            // mapped pinned: 64-byte record cells {end, digest[32], segment, size, flag, pad}
    uint32_t *pending;         // per physical page: chunks not yet loaded + holds of open chunks
    unsigned long long *free_fifo;  // mapped pinned: (sequence << 32) | page
    uint32_t free_mask;
    // A kernel that only ends on request must not outlive a host that died or sits in a blocking read: every ring call
    // bumps word 0 of the `heartbeat` block (mapped pinned); when the service's designated wave (workgroup 0, first
    // producer) has seen neither work nor a heartbeat change for idle_ticks (wall-clock ticks, 100 MHz) it stops the
    // service ON ITS OWN — a handshake with the host makes that safe (kernels.hip, k_sha256_pair): the ring stays
    // healthy, the next pump finds the service gone and starts it again.
    // Heartbeat block (uint32 words): [0] heartbeat (host), [16] stop intent (device), [17] stop committed (device),
    // [32] claim progress (device, for the backlog gate), [33] rounds enqueued so far (host).
    const uint32_t *heartbeat;
    unsigned long long idle_ticks;
    uint32_t poll_mask;        // a wave that still carries chunks looks at the queue when (step & poll_mask) == 0 (0 = every step)
    // An EXPRESS service (k_sha256_xpair on its own CUs) owns the long-chunk queue: the pair service's lanes then take a long
    // chunk only while more than `long_spill` of them wait (the express lanes are all busy), never give up main-queue
    // claims for it, and leave without looking at it.
    uint32_t xp;
    uint32_t long_spill;
    uint32_t xp_pairs;         // lane pairs of the express service (its CUs x 64)
    // While fewer than 3/4 of the express pairs hold or await a chunk (one file alone, the first rounds of a burst) chunks from
    // `long_lo` bytes on go express too: under load the threshold is what 16 CUs can take (13/16 of the maximum), but a lone
    // 64 GiB file ends with the pair chain of its longest chunk BELOW that threshold (12.9 MiB: 0.37 s) although hundreds
    // of express pairs sit idle (0 = off).
    uint32_t long_lo;
};
constexpr int kHbBeat = 0, kHbIntent = 16, kHbCommitted = 17, kHbClaim = 32, kHbRoundsEnq = 33;
constexpr uint32_t kRingProbeSteps = 4096;  // block steps between two samples of a service's probe wave (~7 ms)


// scalar slots of a round (same numbering as engine_internal.h's SC_*)
enum : int { kRsNcand = 0, kRsNrec = 1, kRsMaxcnt = 2, kRsNlong = 3, kRsNshort = 4, kRsShortRoom = 5, kRsTileq = 6 /* u64 */, kRsNmain = 8, kRsCount = 10 };

// One physical page of a round (host-written into mapped pinned memory; the round's kernels read the device copy k_ring_stage makes).
struct RingPage {
    uint64_t phys_off;     // byte offset of the page BODY from the arena base (a 128-byte pad precedes and follows it)
    uint64_t logical;      // (stream slot << kRingOffBits) | offset of the page's first byte within its stream
    uint32_t valid;        // bytes of the page that belong to the stream (== page size except a stream's last page)
    uint32_t slot;         // stream slot
    uint32_t phys;         // physical page index
    uint32_t seg;          // index of the stream's segment entry in this round
    uint64_t fill_seed;    // synthetic producer (bench / tests): generator seed, kind and stream offset
    uint64_t fill_off;
    uint32_t fill_kind;    // 0..4: pbsgpu_fill_device's generators; 5: piece table (fill_tab) over generator 4
    uint32_t do_fill;
    const struct FillPiece *fill_tab;  // kind 5: the stream's piece table (mapped pinned), ascending dst_off, contiguous
    uint32_t fill_ntab;
    uint32_t prev_phys;    // physical page that holds the stream's PREVIOUS logical page (~0: this is the stream's first page).
                           // Host-known, so the page's head pad — the scan's window warm-up — needs no device state:
                           // scan(n + 1) can run while control(n) still resolves (launch_ring_round)
};
// One piece of a synthetic EDITED stream (BASELINE.json configs[4] through the ring): stream bytes [dst_off, dst_off + len)
// are generator 4's bytes (seed) at [src_off, src_off + len) — a kept extent of the base file, or newly written bytes.
// All three offsets / lengths are multiples of 16 (the generator's block).
struct FillPiece {
    uint64_t dst_off, len, src_off, seed;
};
// One stream of a round.
struct RingSeg {
    uint32_t slot;         // stream slot
    uint32_t first_page;   // first entry of the stream's new pages in the round's page table (ascending logical order)
    uint32_t npages;       // may be 0 (a stream that is only being finished)
    uint32_t final;        // the stream ends with this round: its tail becomes the final chunk
    uint64_t new_end;      // logical length of the stream after this round
    uint32_t reset;        // first round of a new stream in this slot: state starts from zero
    uint32_t pad;
    uint64_t origin;       // payload position of the stream's byte 0 (suggested boundaries on the absolute reader grid)
};
// Device-resident state of a stream slot.
constexpr uint64_t kRingMaxStream = 1ull << kRingOffBits;  // logical coordinates are (stream slot << kRingOffBits) | offset
constexpr uint32_t kRingPT = 512;      // page-table window per stream (open chunk <= 2 pages + new pages of one round)
struct RingStreamState {
    uint64_t c;            // start of the open chunk (logical offset in the stream)
    uint64_t end;          // bytes received so far
    uint64_t ecand;        // ~0 or: the hash candidate inside the open chunk that a suggested boundary beyond the bytes
                           // seen so far pre-empts (reader-buffer rule); it is not rescanned, so it is carried here
    uint32_t pad[2];
    uint32_t pt[kRingPT];  // logical page k -> physical page, at [k % kRingPT]
};
struct RingRoundStatus {   // mapped pinned: written last by a round
    uint32_t seq;          // round number + 1
    uint32_t nrec;         // record cells written (open chunks included as void cells)
    uint32_t ncand;
    uint32_t error;        // 2 = record / cell capacity (ring-wide: the host's own bound was wrong, never the data's fault)
    uint32_t tail;         // queue tail after this round
    uint32_t pad[3];
    // the services' probe counters (RingSource::probe, words 0..5) as this round found them: pbsgpu_ring_get_probe answers from
    // the newest reaped round while a service runs — no HIP call of the host beside a persistent kernel
    unsigned long long probe[6];
    // wall-clock ticks (100 MHz) the control kernel spent in its phases: [0] tile prefix + compaction, [1] resolve walks,
    // [2] numbering, [3] records -> cells / descriptors / page references, [4] publish (holds, releases, tail); [5] its start
    // (low 32 bits of the wall clock). pbsgpu_ring_debug sums them per ring.
    uint32_t phase_ticks[6];
};
static_assert(sizeof(RingRoundStatus) == 104, "host slot: 128 bytes (ring.cpp: input_stride)");
struct RingRound {
    // geometry / constants
    uint8_t *arena;            // device: [pad | page 0 | pad][pad | page 1 | pad] ...
    uint32_t page_bytes, stride, tile_bytes, tpp;
    uint32_t effmin, cmin, maxsz, cap;
    uint32_t thr;
    const uint32_t *table_rot;
    // this round's inputs (device copies of the host-written tables; the mapped pinned originals with PBSGPU_RING_STAGE_INPUTS=0)
    const RingPage *pages;
    uint32_t npages;
    const RingSeg *segs_in;
    uint32_t nseg;
    uint32_t seq;              // round number + 1
    uint32_t cell_base, cell_cap, cell_mask;
    uint32_t scan_blocks;      // workgroups the scan may use (the CUs the SHA service leaves free)
    RingRoundStatus *status;   // mapped pinned
    // persistent device state
    RingStreamState *streams;
    RingSource q;
    uint4 *desc_w;             // writable view of q.desc
    uint4 *ldesc_w;            // ... and of q.ldesc
    uint4 *sdesc_w;            // ... and of q.sdesc (nullptr: no lanes service)
    // work buffers. The SCAN side (tile_cnt, tile_slots, tile_queue) exists twice: round n + 1 is scanned on its own HIP
    // stream while round n's control kernel still reads round n's candidates; everything behind the scan runs in order on
    // the control stream and has one set.
    unsigned long long *tile_queue;  // the scan's dynamic tile counter
    uint32_t *scalars;         // SC_* layout of engine_internal.h
    uint32_t *tile_cnt, *tile_off, *tile_slots, *scan_tmp;
    uint64_t *dense;
    uint64_t dense_cap;
    pbsgpu_segment *segs;
    uint32_t *seg_cnt, *seg_off;
    pbsgpu_record *recs;
    uint64_t rec_cap;
    uint64_t *seg_newc;        // per segment: the open chunk's start after this round
    uint32_t *seg_open;        // per segment: 1 = the round left an open chunk
    uint64_t *seg_ecand_in;    // per segment: RingStreamState::ecand before / after this round
    uint64_t *seg_ecand;
    // suggested boundaries (optional): sugg[sugg_idx[s] .. sugg_idx[s+1]) ascending, offsets within stream s
    const uint64_t *sugg;
    const uint32_t *sugg_idx;
    uint64_t sugg_feed;        // reader-buffer rule (Suggested::feed / absolute)
    uint32_t sugg_abs;
    uint32_t pad0;
    // fused control kernel: records of segment s go to recs[seg_rec_base[s] .. seg_rec_base[s + 1]) — the host's bound on
    // what the stream's open chunk + new pages can produce (mapped pinned, nseg + 1 entries)
    const uint32_t *seg_rec_base;
};
// enqueue one cut round on `st` (fill -> pads/segments -> scan -> compaction -> resolve -> descriptors -> publish)
// (`fill_st` / `fill_ev`: the synthetic producer's own stream and the event the cut waits for; null = same stream)
// (`scan_st` / `scan_ev`: the scan's own stream — head pads + scan of this round may overlap the control kernel of the
// previous one — and the event the control stream waits for; null = everything in order on `st`)
// (`stage`: the round's host-written tables — pages, segments, record bases, suggested-offset index — are copied from mapped
// pinned memory into device memory by the round's FIRST kernel, and r.pages / r.segs_in / r.seg_rec_base / r.sugg_idx point
// at the copies; null = the kernels read the mapped host memory itself, as in rounds 3-4)
struct RingStage {
    const uint8_t *src;          // the input block in mapped pinned memory
    uint8_t *dst;                // its device mirror (same layout)
    uint32_t off[4], len[4];     // spans to copy (64-byte aligned offsets, lengths rounded up to 16 inside the block's padding)
    const RingPage *pages_host;  // host view of the page table (launch_ring_round looks at do_fill)
};
hipError_t launch_ring_round(const RingRound &r, int num_cus, hipStream_t st, hipStream_t fill_st = nullptr,
                             hipEvent_t fill_ev = nullptr, hipStream_t scan_st = nullptr, hipEvent_t scan_ev = nullptr,
                             const RingStage *stage = nullptr);
// the persistent SHA-256 service: `workgroups` x (2 producer + 2 consumer waves), one per CU
hipError_t launch_ring_service(const RingSource &q, unsigned workgroups, hipStream_t st, bool dense = false);
hipError_t launch_ring_service_xp(const RingSource &q, unsigned workgroups, hipStream_t st);
hipError_t launch_ring_service_lanes(const RingSource &q, unsigned workgroups, hipStream_t st, bool dense);
// raise `stop` behind everything enqueued so far on `st`
hipError_t launch_ring_stop(RingCtl *ctl, hipStream_t st);
hipError_t launch_ring_reset(RingCtl *ctl, hipStream_t st);

// digest-set: sort keys/index pairs by the first 8 digest bytes (big-endian) and flag duplicates
hipError_t launch_dedup(const pbsgpu_record *recs, uint64_t n, uint64_t *keys, uint32_t *idx,
                        uint64_t *keys_alt, uint32_t *idx_alt, uint8_t *dup, uint64_t *stats4,
                        void *tmp, size_t tmp_bytes, hipStream_t st);
size_t dedup_tmp_bytes(uint64_t n);

// payload-stream assembly: kind 0 = copy len bytes from src_base+src_off; kind 1 = 16-byte
// {type = src_off, size = len} header at dst_off
struct PackItem {
    uint64_t src_off;
    uint64_t dst_off;
    uint64_t len;
    uint32_t kind;
    uint32_t pad;
};
hipError_t launch_pack(const uint8_t *src_base, uint8_t *dst, const PackItem *items, uint32_t nitems, hipStream_t st);

}  // namespace pbsk
