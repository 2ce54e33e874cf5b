// libpbsgpu host side, part 3: the chunk-reuse planner — pure index arithmetic over the engine's
// (end, digest) record lists. Restates what the commit walk computes from the previous snapshot's
// dynamic index (reference internal/pxarmount/commit_reuse.go:84-135 lookupDynamicEntries,
// :152-183 shouldReuse, threshold internal/pxarmount/commit_types.go:14) so that callers of the
// GPU engine can decide between InjectChunks (forced cut, pbsgpu_stream_cut) and re-chunking.
#include <cstring>

#include "../../include/pbsgpu.h"

namespace {

// datastore.DynamicIndexReader.ChunkFromOffset: index of the chunk that contains `off`
// (first entry whose end lies beyond it); false past the end of the index.
bool chunk_from_offset(const pbsgpu_record *idx, uint64_t n, uint64_t off, uint64_t *out) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (idx[mid].end <= off) lo = mid + 1; else hi = mid;
    }
    if (lo >= n) return false;
    *out = lo;
    return true;
}

}  // namespace

extern "C" {

int pbsgpu_reuse_lookup(const pbsgpu_record *idx, uint64_t n, uint64_t range_start, uint64_t range_end,
                        pbsgpu_reuse_chunk *out, uint64_t cap, uint64_t *nchunks, uint64_t *start_padding,
                        uint64_t *end_padding) {
    if (!nchunks || !start_padding || !end_padding || (n && !idx)) return PBSGPU_E_INVALID;
    *nchunks = 0;
    *start_padding = 0;
    *end_padding = 0;
    if (n == 0 || range_start >= range_end) return PBSGPU_OK;
    uint64_t first = 0;
    if (!chunk_from_offset(idx, n, range_start, &first)) return PBSGPU_OK;
    uint64_t prev_end = first ? idx[first - 1].end : 0;
    const uint64_t spad = range_start - prev_end;
    uint64_t epad = 0, k = 0;
    for (uint64_t i = first; i < n; ++i) {
        const uint64_t end = idx[i].end;
        if (k < cap && out) {
            out[k].size = end - prev_end;
            out[k].padding = 0;
            out[k].end_offset = end;
            std::memcpy(out[k].digest, idx[i].digest, 32);
        }
        prev_end = end;
        const bool beyond = range_end < end;  // note: a range ending exactly ON a chunk end pulls in the next chunk too
        if (beyond) epad = end - range_end;
        ++k;
        if (beyond) break;
    }
    *nchunks = k;
    *start_padding = spad;
    *end_padding = epad;
    if (k > cap) return out ? PBSGPU_E_CAPACITY : PBSGPU_OK;
    if (out && k) {
        out[0].padding += spad;
        out[k - 1].padding += epad;
    }
    return PBSGPU_OK;
}

int pbsgpu_reuse_should(const pbsgpu_record *idx, uint64_t n, uint64_t range_start, uint64_t range_end,
                        const pbsgpu_reuse_chunk *saved, double threshold, int *reuse) {
    if (!reuse || (n && !idx)) return PBSGPU_E_INVALID;
    *reuse = 1;
    if (n == 0 || range_end <= range_start) return PBSGPU_OK;  // no index / empty range: reuse
    pbsgpu_reuse_chunk first{};
    uint64_t k = 0, spad = 0, epad = 0;
    int st = pbsgpu_reuse_lookup(idx, n, range_start, range_end, &first, 1, &k, &spad, &epad);
    if (st != PBSGPU_OK && st != PBSGPU_E_CAPACITY) return st;
    if (k == 0) return PBSGPU_OK;
    uint64_t padding = spad + epad;
    if (saved && std::memcmp(saved->digest, first.digest, 32) == 0 && saved->end_offset == first.end_offset) {
        const uint64_t used = saved->size - saved->padding;
        padding = (used > padding) ? 0 : padding - used;
    }
    const uint64_t total = (range_end - range_start) + padding;
    if (total == 0) return PBSGPU_OK;
    *reuse = ((double)padding / (double)total <= threshold) ? 1 : 0;
    return PBSGPU_OK;
}

}  // extern "C"
