// Internal declarations shared by the libpbsgpu translation units (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
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

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PBSGPU_OK;
        if (p) {
            (void)hipFree(p);
            p = nullptr;
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
            p = nullptr;
            return PBSGPU_E_NOMEM;
        }
        cap = want;
        return PBSGPU_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return PBSGPU_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            g_last_hip_error.store((int)e);
            p = nullptr;
            return PBSGPU_E_NOMEM;
        }
        cap = bytes;
        return PBSGPU_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

// device scalars of one slot (uint32 each)
enum : int { SC_NCAND = 0, SC_NREC = 1, SC_MAXCNT = 2, SC_QUEUE = 3, SC_WGLIMIT = 4, SC_ZERO = 5 /* stays 0 */,
             SC_TILEQ = 6 /* u64 */, SC_COUNT = 8 };

enum : int { EV_BEGIN = 0, EV_SCAN0, EV_SCAN1, EV_RESOLVE1, EV_SHA1, EV_COUNT };

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[EV_COUNT] = {};
    DevBuf data;  // staged copy of host submits
    DevBuf tile_cnt, tile_off, tile_slots, dense, scan_tmp, scalars, segs, seg_cnt, seg_off, recs, order;
    PinnedBuf h_scalars;  // readback of SC_*
    PinnedBuf h_segs;     // pinned copy of the segment table
    // in-flight state
    bool busy = false;
    bool synced = false;
    uint64_t ticket = 0;
    const uint8_t *dptr = nullptr;
    uint64_t nbytes = 0;
    uint32_t nseg = 0;
    uint32_t cap = 0;
    uint64_t rec_cap = 0;
    bool host_submit = false;
    uint32_t retries = 0;
    uint64_t nrec = 0, ncand = 0;
};

}  // namespace pbse

struct pbsgpu_engine {
    int device = 0;
    int num_cus = 256;
    // SHA workgroup budget slack (k_order): 25 % keeps one batch's makespan at its longest chunk; with more
    // than 4 batches in flight the chip is oversubscribed anyway and 0 % (fewest CUs per batch) carries more
    uint32_t sha_slack_pct = 25;
    pbsgpu_config cfg{};
    uint32_t bits = 0;     // mask == 2^bits - 1
    uint32_t thr = 0;      // break_min << (32 - bits)
    uint32_t effmin = 0;   // max(min, 65)
    uint32_t *d_table_rot = nullptr;
    std::vector<pbse::Slot> slots;
    uint64_t next_ticket = 1;
    uint32_t cap_hint = 0;       // per-tile slot capacity that a density retry settled on
    uint32_t cap_hint_tile = 0;  // ... for this tile size
    pbse::PinnedBuf stage[2];
    hipEvent_t stage_ev[2] = {};
    std::mutex mu;
};


namespace pbse {

uint32_t default_cap(const pbsgpu_engine *e, uint64_t nbytes);
int set_device(const pbsgpu_engine *e);
Slot *find_free_slot(pbsgpu_engine *e);
bool is_device_pointer(const void *p);
int enqueue_candidates(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, uint32_t cap,
                       uint64_t nseg_hint = 0);
int staged_h2d(pbsgpu_engine *e, void *dst, const void *src, uint64_t nbytes, hipStream_t st);
// synchronous helpers for the streaming front ends (caller holds e->mu)
int candidates_sync(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, uint64_t *count);
int batch_sync(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
               uint32_t nseg, uint64_t *nrec);
int cut_sync(pbsgpu_engine *e, Slot &s, const uint8_t *dptr, uint64_t nbytes, const pbsgpu_segment *segs,
             uint32_t nseg, uint64_t *nrec);
int hash_async(pbsgpu_engine *e, Slot &s, uint64_t nhash);

}  // namespace pbse
