// C++ mirror on the GPU: the way the reference's tests drive the module (commit_walk_test.go:21-147:
// NewConfig(4096) -> writer -> small + large entries -> Finish -> read the index back), written
// against include/pbsgpu.hpp. Prints one line per chunk: "<end> <size> <sha256 hex>"; the Python test
// compares them with the oracle.
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

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 700001;
    auto cfg = buzhash::NewConfig(4096);  // commit_walk_test.go:25
    if (!cfg) { std::fprintf(stderr, "%s\n", cfg.err.c_str()); return 2; }
    auto eng = Engine::New(0, cfg.value, 2);
    if (!eng) { std::fprintf(stderr, "%s\n", eng.err.c_str()); return 3; }

    std::string payload(n, '\0');  // the synthetic stream (same generator as oracle_fill kind 0, seed 77)
    for (uint64_t i = 0; i < n; ++i) payload[i] = (char)(splitmix64(77, i >> 3) >> (8 * (i & 7)));

    std::vector<std::pair<datastore::ChunkInfo, uint32_t>> got;
    auto w = transfer::PayloadWriter::New(eng.value, [&](const datastore::ChunkInfo &ci, uint32_t size) { got.push_back({ci, size}); }, 1 << 16);
    if (!w) { std::fprintf(stderr, "%s\n", w.err.c_str()); return 4; }
    // three "files": a tiny one via WriteEntry, the rest via WriteEntryReader
    std::string e = w.value->WriteEntry(payload.data(), 17);
    std::istringstream r1(payload.substr(17, 300000)), r2(payload.substr(300017));
    if (e.empty()) e = w.value->WriteEntryReader(r1, 300000);
    if (e.empty()) e = w.value->WriteEntryReader(r2, n - 300017);
    if (e.empty() && w.value->PayloadPosition() != n) e = "payload position mismatch";
    std::istringstream shortr("abc");
    if (e.empty() && w.value->WriteEntryReader(shortr, 0).size()) e = "zero-size entry failed";
    if (e.empty()) e = w.value->Finish();
    if (!e.empty()) { std::fprintf(stderr, "%s\n", e.c_str()); return 5; }

    // dynamic index round trip through the mirror (NewDynamicIndexWriter(ctime).Add(...).Finish())
    auto iw = datastore::NewDynamicIndexWriter(1700000123);
    for (auto &g : got) iw.Add(g.first.End, g.first.Digest_);
    auto blob = iw.Finish(eng.value->handle());
    if (!blob) { std::fprintf(stderr, "%s\n", blob.err.c_str()); return 6; }
    auto idx = datastore::ParseDynamicIndex(blob.value);
    if (!idx || idx.value->Count() != (int)got.size() || idx.value->CTime() != 1700000123) return 7;
    for (int i = 0; i < idx.value->Count(); ++i) {
        auto ci = idx.value->ChunkInfoAt(i).first;
        if (ci.End != got[(size_t)i].first.End || ci.Digest_ != got[(size_t)i].first.Digest_) return 8;
    }
    for (auto &g : got) {
        std::printf("%llu %u ", (unsigned long long)g.first.End, g.second);
        for (uint8_t b : g.first.Digest_) std::printf("%02x", b);
        std::printf("\n");
    }
    std::printf("cpp-writer-ok %zu\n", got.size());
    return 0;
}
