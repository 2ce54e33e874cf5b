// C++ mirror, host-only part (no GPU): buzhash::NewConfig error return and the DynamicIndexReader
// calls the commit walk makes (Count / ChunkInfo / ChunkFromOffset: internal/pxarmount/commit_reuse.go:84-125)
// on a synthetic index built like buildSyntheticDIDX (commit_bottleneck_test.go:773-793).
#include <cassert>
#include <cstdio>

#include "../../include/pbsgpu.hpp"

using namespace pbsgpu;

static std::vector<uint8_t> synthetic_didx(int numChunks, uint64_t chunkSize, int64_t ctime) {
    std::vector<uint8_t> img(4096 + (size_t)numChunks * 40, 0);
    const uint8_t magic[8] = {28, 145, 78, 165, 25, 186, 179, 205};
    std::memcpy(img.data(), magic, 8);
    for (int b = 0; b < 8; ++b) img[24 + b] = (uint8_t)((uint64_t)ctime >> (8 * b));
    uint64_t off = 0;
    for (int i = 0; i < numChunks; ++i) {
        off += chunkSize;
        uint8_t *e = img.data() + 4096 + (size_t)i * 40;
        for (int b = 0; b < 8; ++b) e[b] = (uint8_t)(off >> (8 * b));
        e[8] = (uint8_t)i;
        e[9] = (uint8_t)(i >> 8);
    }
    return img;
}

int main() {
    auto cfg = buzhash::NewConfig(4 << 20);  // commit_orchestrate.go:144
    assert(cfg && cfg.value.MinSize == (1 << 20) && cfg.value.MaxSize == (16 << 20) && cfg.value.WindowSize == 64);
    assert(cfg.value.BreakTestMask == 0x7FFFFF && cfg.value.BreakTestMinimum == 0x7FFFFD);
    auto bad = buzhash::NewConfig(3 << 20);  // not a power of two -> error, like the Go constructor
    assert(!bad && !bad.err.empty());
    assert(!buzhash::NewConfig(-5) && !buzhash::NewConfig(64));

    auto idx = datastore::ParseDynamicIndex(synthetic_didx(5, 100, 1700000000));
    assert(idx && idx.value->Count() == 5 && idx.value->CTime() == 1700000000);
    auto [info, ok] = idx.value->ChunkInfoAt(2);
    assert(ok && info.End == 300 && info.Digest_[0] == 2);
    assert(!idx.value->ChunkInfoAt(5).second && !idx.value->ChunkInfoAt(-1).second);
    struct { uint64_t off; int want; bool ok; } q[] = {{0, 0, true}, {99, 0, true}, {100, 1, true}, {250, 2, true}, {499, 4, true}, {500, 0, false}, {600, 0, false}};
    for (auto &t : q) {
        auto [i, found] = idx.value->ChunkFromOffset(t.off);
        assert(found == t.ok && (!found || i == t.want));
    }
    // the walk of lookupDynamicEntries, written against the mirror (range [50, 350) of TestLookupDynamicEntries)
    uint64_t rangeStart = 50, rangeEnd = 350, prevEnd = 0, endPadding = 0;
    auto [startIdx, found] = idx.value->ChunkFromOffset(rangeStart);
    assert(found && startIdx == 0);
    int chunks = 0;
    for (int i = startIdx; i < idx.value->Count(); ++i) {
        auto ci = idx.value->ChunkInfoAt(i).first;
        prevEnd = ci.End;
        ++chunks;
        if (rangeEnd < ci.End) { endPadding = ci.End - rangeEnd; break; }
    }
    assert(chunks == 4 && endPadding == 50 && prevEnd == 400);

    auto trunc = synthetic_didx(3, 10, 0);
    trunc.pop_back();
    assert(!datastore::ParseDynamicIndex(trunc));
    std::puts("cpp-reader-ok");
    return 0;
}
