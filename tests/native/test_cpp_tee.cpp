// C++ mirror on the GPU, round 2: the payload half of WriteEntryReader for regular files the way writeBackedFile
// drives it (internal/pxarmount/commit_reuse.go:427-468): start marker, { payload header + body + XXH3 tee }*, tail
// marker, InjectChunks in between, suggested boundaries at file starts. Prints "F <index> <size> <xxh3 hex> <offset>"
// per file and "C <end> <size> <sha256 hex>" per chunk; the Python test checks both against xxhash / the oracle.
#include <cstdio>
#include <sstream>

#include "../../include/pbsgpu.hpp"

using namespace pbsgpu;

static uint64_t splitmix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main() {
    auto cfg = buzhash::NewConfig(4096);
    if (!cfg) return 2;
    auto eng = Engine::New(0, cfg.value, 2);
    if (!eng) { std::fprintf(stderr, "%s\n", eng.err.c_str()); return 3; }
    std::vector<std::pair<datastore::ChunkInfo, uint32_t>> got;
    auto w = transfer::PayloadWriter::New(eng.value, [&](const datastore::ChunkInfo &ci, uint32_t size) { got.push_back({ci, size}); }, 1 << 16);
    if (!w) return 4;
    const uint64_t sizes[] = {300000, 0, 17, 70000, 1024, 200001};
    std::string e = w.value->WriteMarker(false);
    std::vector<uint64_t> offs;
    for (size_t k = 0; k < sizeof(sizes) / sizeof(sizes[0]) && e.empty(); ++k) {
        std::string body(sizes[k], '\0');  // oracle_fill kind 0, seed 100 + k
        for (uint64_t i = 0; i < sizes[k]; ++i) body[i] = (char)(splitmix64(100 + k, i >> 3) >> (8 * (i & 7)));
        std::istringstream r(body);
        uint64_t off = 0, idx = 0;
        if (k == 3) e = w.value->SuggestBoundary();  // "a file starts here"
        if (e.empty()) e = w.value->WritePayloadEntry(r, sizes[k], &off, &idx);
        if (e.empty() && idx != k) e = "file index mismatch";
        offs.push_back(off);
        if (e.empty() && k == 1) e = w.value->InjectChunks({datastore::KnownChunkRef{{}, 123456}});
    }
    if (e.empty()) e = w.value->WriteMarker(true);
    if (e.empty()) e = w.value->Finish();
    if (!e.empty()) { std::fprintf(stderr, "%s\n", e.c_str()); return 5; }
    auto files = w.value->BackedHashes();
    if (files.size() != 6) return 6;
    for (auto &f : files)
        std::printf("F %llu %llu %016llx %llu\n", (unsigned long long)f.index, (unsigned long long)f.size,
                    (unsigned long long)f.xxh3, (unsigned long long)offs[(size_t)f.index]);
    for (auto &g : got) {
        std::printf("C %llu %u ", (unsigned long long)g.first.End, g.second);
        for (uint8_t b : g.first.Digest_) std::printf("%02x", b);
        std::printf("\n");
    }
    std::printf("cpp-tee-ok %llu\n", (unsigned long long)w.value->PayloadPosition());
    return 0;
}
