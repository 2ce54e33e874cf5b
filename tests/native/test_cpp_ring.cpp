// A C++ program written against include/pbsgpu.hpp's PageRing (the page ring: several archives at once on bytes in
// device memory). Three synthetic streams (generator seed / kind / length on the command line are fixed here) go through
// a small arena; every (stream, end, size, digest) entry is printed for the Python test to compare with the oracle.
#include <cstdio>
#include <map>

#include "../../include/pbsgpu.hpp"

int main() {
    auto cfg = pbsgpu::buzhash::NewConfig(65536);
    if (!cfg) return 1;
    auto eng = pbsgpu::Engine::New(0, cfg.value, 1);
    if (!eng) { std::fprintf(stderr, "%s\n", eng.err.c_str()); return 2; }
    std::map<uint32_t, int> job_of;
    auto sink = [&](uint32_t stream, const pbsgpu::datastore::ChunkInfo &ci, uint32_t size) {
        std::printf("C %d %llu %u ", job_of[stream], (unsigned long long)ci.End, size);
        for (uint8_t b : ci.Digest_) std::printf("%02x", b);
        std::printf("\n");
    };
    pbsgpu_ring_options opt{};
    opt.page_bytes = 262144;
    opt.arena_bytes = 48ull * (262144 + 256);
    opt.max_streams = 4;
    opt.sha_cus = 8;
    opt.round_pages = 8;
    auto ring = pbsgpu::transfer::PageRing::New(eng.value, sink, &opt);
    if (!ring) { std::fprintf(stderr, "%s\n", ring.err.c_str()); return 3; }
    struct Job { uint64_t seed; uint32_t kind; uint64_t len, left; uint32_t sid; bool done; };
    Job jobs[3] = {{71, 0, 5u * 262144 + 777, 0, 0, false}, {72, 3, 9u * 262144, 0, 0, false}, {73, 4, 100, 0, 0, false}};
    for (int j = 0; j < 3; ++j) {
        auto s = ring.value->Open();
        if (!s) return 4;
        jobs[j].sid = s.value;
        jobs[j].left = jobs[j].len;
        job_of[s.value] = j;
    }
    int open = 3;
    for (int spin = 0; open > 0 && spin < 2000000; ++spin) {
        for (auto &j : jobs) {
            if (j.done) continue;
            if (j.left) {
                auto t = ring.value->FillSynthetic(j.sid, j.seed, j.kind, j.left, true);
                if (!t) { std::fprintf(stderr, "%s\n", t.err.c_str()); return 5; }
                j.left -= t.value;
            }
            bool fin = false;
            const std::string e = ring.value->Pump(j.sid, &fin);
            if (!e.empty()) { std::fprintf(stderr, "%s\n", e.c_str()); return 6; }
            if (fin) {
                if (!ring.value->CloseStream(j.sid).empty()) return 7;
                j.done = true;
                --open;
            }
        }
    }
    if (open) return 8;
    if (!ring.value->Quiesce().empty()) return 9;
    std::printf("cpp-ring-ok\n");
    return 0;
}
