// include/pbsgpu.hpp — header-only C++17 host mirror of the reference's Go interface for the pxar
// stream path, over the C ABI in pbsgpu.h (the reference is compiled Go; no Go toolchain exists in
// the build image, so the host side above the C ABI is C++ — names, argument meaning and error
// behaviour follow the module API as pbs-plus uses it):
//
//   buzhash::NewConfig(avg)                      internal/pxarmount/commit_orchestrate.go:143-149,
//                                                internal/tapeio/converter.go:248
//   datastore::NewDynamicIndexWriter(ctime).Add(end, digest).Finish()
//                                                internal/pxarmount/commit_bottleneck_test.go:773-793
//   datastore::ParseDynamicIndex / DynamicIndexReader{Count, ChunkInfo, ChunkFromOffset}
//                                                internal/pxarmount/commit_reuse.go:84-135, commit_orchestrate.go:219
//   transfer::PayloadWriter{WriteEntryReader, InjectChunks, Finish}  (the payload half of transfer.ArchiveWriter)
//                                                internal/pxarmount/commit_test.go:33-67, commit_reuse.go:315-341,457
//
// Go's (value, error) returns become a Result<T>{value, err}: err.empty() means success.
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <istream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "pbsgpu.h"

namespace pbsgpu {

template <typename T> struct Result {
    T value{};
    std::string err;  // empty = nil error
    explicit operator bool() const { return err.empty(); }
};

inline std::string errorf(const char *what, int status) {
    return std::string("pbsgpu: ") + what + ": " + pbsgpu_strerror(status);
}

namespace buzhash {

// buzhash.Config: a plain value handed to NewPBSStore / NewLocalStore / BackupConfig.ChunkConfig
struct Config {
    int AvgSize = 0, MinSize = 0, MaxSize = 0, WindowSize = 0;
    uint32_t BreakTestMask = 0, BreakTestMinimum = 0;
    pbsgpu_config c{};
};

// buzhash.NewConfig(avgSize int) (Config, error)
inline Result<Config> NewConfig(int avgSize) {
    Result<Config> r;
    const int st = pbsgpu_config_init(avgSize < 0 ? 0 : (uint64_t)avgSize, nullptr, &r.value.c);
    if (st != PBSGPU_OK) {
        r.err = "buzhash: average chunk size must be a power of two in [256, 2^28]";
        return r;
    }
    const pbsgpu_config &c = r.value.c;
    r.value.AvgSize = (int)c.avg;
    r.value.MinSize = (int)c.min;
    r.value.MaxSize = (int)c.max;
    r.value.WindowSize = (int)c.window;
    r.value.BreakTestMask = c.mask;
    r.value.BreakTestMinimum = c.break_min;
    return r;
}

}  // namespace buzhash

namespace datastore {

using Digest = std::array<uint8_t, 32>;

// datastore.ChunkInfo{End, Digest}
struct ChunkInfo {
    uint64_t End = 0;
    Digest Digest_{};
};

// backupproxy.KnownChunkRef{Digest, Size}
struct KnownChunkRef {
    Digest Digest_{};
    uint64_t Size = 0;
};

// datastore.DynamicIndexReader
class DynamicIndexReader {
  public:
    int Count() const { return (int)recs_.size(); }
    // (info, ok)
    std::pair<ChunkInfo, bool> ChunkInfoAt(int i) const {
        ChunkInfo ci;
        if (i < 0 || i >= Count()) return {ci, false};
        ci.End = recs_[(size_t)i].end;
        std::memcpy(ci.Digest_.data(), recs_[(size_t)i].digest, 32);
        return {ci, true};
    }
    // index of the chunk containing `offset`; ok = false past the end (commit_reuse.go:89-92)
    std::pair<int, bool> ChunkFromOffset(uint64_t offset) const {
        size_t lo = 0, hi = recs_.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (recs_[mid].end <= offset) lo = mid + 1; else hi = mid;
        }
        if (lo >= recs_.size()) return {0, false};
        return {(int)lo, true};
    }
    int64_t CTime() const { return ctime_; }
    const Digest &IndexCsum() const { return csum_; }
    const std::vector<pbsgpu_record> &Records() const { return recs_; }

  private:
    friend Result<std::shared_ptr<DynamicIndexReader>> ParseDynamicIndex(const std::vector<uint8_t> &data);
    std::vector<pbsgpu_record> recs_;
    int64_t ctime_ = 0;
    Digest csum_{};
};

// datastore.ParseDynamicIndex(data) (*DynamicIndexReader, error)
inline Result<std::shared_ptr<DynamicIndexReader>> ParseDynamicIndex(const std::vector<uint8_t> &data) {
    Result<std::shared_ptr<DynamicIndexReader>> r;
    uint64_t n = 0;
    int st = pbsgpu_didx_decode(data.data(), data.size(), nullptr, 0, &n, nullptr, nullptr);
    if (st != PBSGPU_OK && st != PBSGPU_E_CAPACITY) {
        r.err = errorf("parse dynamic index", st);
        return r;
    }
    auto idx = std::make_shared<DynamicIndexReader>();
    idx->recs_.resize((size_t)n);
    st = pbsgpu_didx_decode(data.data(), data.size(), idx->recs_.data(), n, &n, &idx->ctime_, idx->csum_.data());
    if (st != PBSGPU_OK) {
        r.err = errorf("parse dynamic index", st);
        return r;
    }
    r.value = std::move(idx);
    return r;
}

// datastore.NewDynamicIndexWriter(ctime).Add(end, digest).Finish() — the index checksum is computed
// by the engine's SHA-256 kernel, so Finish needs an engine.
class DynamicIndexWriter {
  public:
    explicit DynamicIndexWriter(int64_t ctime) : ctime_(ctime) {}
    void Add(uint64_t endOffset, const Digest &digest) {
        pbsgpu_record r{};
        r.end = endOffset;
        std::memcpy(r.digest, digest.data(), 32);
        r.size = (uint32_t)(endOffset - (recs_.empty() ? 0 : recs_.back().end));
        recs_.push_back(r);
    }
    Result<std::vector<uint8_t>> Finish(pbsgpu_engine *eng, const std::array<uint8_t, 16> &uuid = {}) {
        Result<std::vector<uint8_t>> r;
        uint64_t nb = 0;
        pbsgpu_didx_size(recs_.size(), &nb);
        r.value.resize((size_t)nb);
        const int st = pbsgpu_didx_encode(eng, recs_.data(), recs_.size(), uuid.data(), ctime_, r.value.data(), nb);
        if (st != PBSGPU_OK) {
            r.value.clear();
            r.err = errorf("finish dynamic index", st);
        }
        return r;
    }

  private:
    int64_t ctime_;
    std::vector<pbsgpu_record> recs_;
};

inline DynamicIndexWriter NewDynamicIndexWriter(int64_t ctime) { return DynamicIndexWriter(ctime); }

}  // namespace datastore

// One engine per (process, GPU): stands where backupproxy's store/session owns chunker + hasher.
class Engine {
  public:
    static Result<std::shared_ptr<Engine>> New(int device, const buzhash::Config &cfg, unsigned inflight = 2) {
        Result<std::shared_ptr<Engine>> r;
        pbsgpu_engine *h = nullptr;
        const int st = pbsgpu_engine_create(device, &cfg.c, inflight, &h);
        if (st != PBSGPU_OK) {
            r.err = errorf("engine create", st);
            return r;
        }
        r.value = std::shared_ptr<Engine>(new Engine(h));
        return r;
    }
    // ... with every per-engine tuning value (pbsgpu_engine_options, ABI v5; zero fields = the defaults)
    static Result<std::shared_ptr<Engine>> New(int device, const buzhash::Config &cfg, const pbsgpu_engine_options &opt) {
        Result<std::shared_ptr<Engine>> r;
        pbsgpu_engine *h = nullptr;
        const int st = pbsgpu_engine_create_opt(device, &cfg.c, &opt, &h);
        if (st != PBSGPU_OK) {
            r.err = errorf("engine create", st);
            return r;
        }
        r.value = std::shared_ptr<Engine>(new Engine(h));
        return r;
    }
    ~Engine() { pbsgpu_engine_destroy(h_); }
    pbsgpu_engine *handle() const { return h_; }

  private:
    explicit Engine(pbsgpu_engine *h) : h_(h) {}
    Engine(const Engine &) = delete;
    pbsgpu_engine *h_;
};

namespace transfer {

// The payload half of transfer.ArchiveWriter: WriteEntryReader pulls `size` bytes from a reader and
// appends them to the payload stream; finished chunks arrive at the sink in stream order as
// (ChunkInfo, size) — exactly what the session appends to the .ppxar.didx.
class PayloadWriter {
  public:
    using Sink = std::function<void(const datastore::ChunkInfo &, uint32_t size)>;

    static Result<std::unique_ptr<PayloadWriter>> New(std::shared_ptr<Engine> eng, Sink sink, uint64_t windowBytes = 0) {
        Result<std::unique_ptr<PayloadWriter>> r;
        pbsgpu_stream *s = nullptr;
        const int st = pbsgpu_stream_create(eng->handle(), windowBytes, &s);
        if (st != PBSGPU_OK) {
            r.err = errorf("stream create", st);
            return r;
        }
        r.value.reset(new PayloadWriter(std::move(eng), s, std::move(sink)));
        return r;
    }
    ~PayloadWriter() { pbsgpu_stream_destroy(s_); }

    // WriteEntryReader(entry, r, size) error — reads exactly `size` bytes from r (zero-copy into the
    // library's pinned staging), error on a short read like the Go writer
    std::string WriteEntryReader(std::istream &r, uint64_t size) {
        uint64_t left = size;
        while (left) {
            void *buf = nullptr;
            size_t cap = 0;
            int st = pbsgpu_stream_reserve(s_, &buf, &cap);
            if (st != PBSGPU_OK) return errorf("write entry", st);
            const size_t want = (size_t)std::min<uint64_t>(cap, left);
            r.read(static_cast<char *>(buf), (std::streamsize)want);
            const size_t got = (size_t)r.gcount();
            st = pbsgpu_stream_commit(s_, got);
            if (st != PBSGPU_OK) return errorf("write entry", st);
            left -= got;
            if (got < want) return "write entry: unexpected EOF";
            if (std::string e = drain(); !e.empty()) return e;
        }
        return {};
    }
    // The whole payload half of ArchiveWriter.WriteEntryReader for a regular file: 16-byte pxar payload header, `size`
    // bytes from r (zero-copy), per-file XXH3-64 tee (writeBackedFile, commit_reuse.go:427-468). *payloadOffset = the
    // header's payload position (what WriteEntryRef / PAYLOAD_REF records, commit_walk.go:455); the hash arrives
    // later through BackedHashes() under *fileIndex.
    std::string WritePayloadEntry(std::istream &r, uint64_t size, uint64_t *payloadOffset, uint64_t *fileIndex) {
        int st = pbsgpu_stream_begin_entry(s_, nullptr, size, payloadOffset);
        if (st != PBSGPU_OK) return errorf("begin entry", st);
        if (std::string e = WriteEntryReader(r, size); !e.empty()) return e;
        st = pbsgpu_stream_end_entry(s_, fileIndex);
        return st == PBSGPU_OK ? drain() : errorf("end entry", st);
    }
    std::string WriteMarker(bool tail) {
        const int st = pbsgpu_stream_write_marker(s_, nullptr, tail ? 1 : 0);
        return st == PBSGPU_OK ? std::string() : errorf("write marker", st);
    }
    // backedHashes (commit_reuse.go:461): (file index, size, XXH3-64) of every file whose last byte has been cut
    std::vector<pbsgpu_file_hash> BackedHashes() {
        std::vector<pbsgpu_file_hash> out;
        pbsgpu_file_hash buf[256];
        for (;;) {
            uint64_t n = 0;
            if (pbsgpu_stream_poll_files(s_, buf, 256, &n) != PBSGPU_OK) break;
            out.insert(out.end(), buf, buf + n);
            if (n < 256) break;
        }
        return out;
    }
    // WriteEntry(entry, content []byte)
    std::string WriteEntry(const void *data, size_t len) {
        const int st = pbsgpu_stream_write(s_, data, len);
        if (st != PBSGPU_OK) return errorf("write entry", st);
        return drain();
    }
    // InjectChunks(refs): the open chunk is flushed, offsets skip the injected payload (commit_reuse.go:315-341)
    std::string InjectChunks(const std::vector<datastore::KnownChunkRef> &refs) {
        uint64_t total = 0;
        for (const auto &k : refs) total += k.Size;
        const int st = pbsgpu_stream_cut(s_, total);
        if (st != PBSGPU_OK) return errorf("inject chunks", st);
        return drain();
    }
    // Encoder().PayloadPosition() (commit_reuse.go:265): payload bytes written PLUS bytes injected so far — the
    // coordinate system of PAYLOAD_REF offsets and of the records' End (InjectChunks advances the position:
    // keepLast_chunk_test.go mock, enc.Advance(total))
    uint64_t PayloadPosition() const {
        uint64_t n = 0;
        pbsgpu_stream_position(s_, &n);
        return n;
    }
    // payload chunker: suggest a chunk boundary at the current position ("a file starts here"); taken when the open
    // chunk is then within [min, max], see pbsgpu_submit_device_suggested
    std::string SuggestBoundary() {
        const int st = pbsgpu_stream_suggest(s_, PayloadPosition());
        return st == PBSGPU_OK ? std::string() : errorf("suggest boundary", st);
    }
    std::string Finish() {
        const int st = pbsgpu_stream_finish(s_);
        if (st != PBSGPU_OK) return errorf("finish", st);
        return drain();
    }
    // the same in two steps, for a writer that starts its next archive while this one's last chunks are hashed:
    // FinishBegin closes the input and returns; Done() delivers what has arrived to the sink and reports whether the
    // last entry is out (never blocks)
    std::string FinishBegin() {
        const int st = pbsgpu_stream_finish_begin(s_);
        return st == PBSGPU_OK ? std::string() : errorf("finish begin", st);
    }
    Result<bool> Done() {
        Result<bool> r;
        int d = 0;
        const int st = pbsgpu_stream_done(s_, &d);
        if (st != PBSGPU_OK) {
            r.err = errorf("done", st);
            return r;
        }
        r.err = drain();
        r.value = d != 0;
        return r;
    }

  private:
    PayloadWriter(std::shared_ptr<Engine> e, pbsgpu_stream *s, Sink sink) : eng_(std::move(e)), s_(s), sink_(std::move(sink)) {}
    std::string drain() {
        pbsgpu_record buf[256];
        for (;;) {
            uint64_t n = 0;
            const int st = pbsgpu_stream_poll(s_, buf, 256, &n);
            if (st != PBSGPU_OK) return errorf("poll", st);
            for (uint64_t i = 0; i < n; ++i) {
                datastore::ChunkInfo ci;
                ci.End = buf[i].end;
                std::memcpy(ci.Digest_.data(), buf[i].digest, 32);
                sink_(ci, buf[i].size);
            }
            if (n < 256) return {};
        }
    }
    std::shared_ptr<Engine> eng_;
    pbsgpu_stream *s_;
    Sink sink_;
};

// Several archives at once on bytes that are (or arrive) in device memory: the page ring (pbsgpu_ring_*). One writer
// per archive in the reference (commit_reuse.go:457, converter.go:836); here each archive is a stream of the ring and
// the sink receives its (End, Digest) entries in stream order. One thread drives a PageRing.
class PageRing {
  public:
    using Sink = std::function<void(uint32_t stream, const datastore::ChunkInfo &, uint32_t size)>;

    static Result<std::unique_ptr<PageRing>> New(std::shared_ptr<Engine> eng, Sink sink, const pbsgpu_ring_options *opt = nullptr) {
        Result<std::unique_ptr<PageRing>> r;
        pbsgpu_ring *h = nullptr;
        const int st = pbsgpu_ring_create(eng->handle(), opt, &h);
        if (st != PBSGPU_OK) {
            r.err = errorf("ring create", st);
            return r;
        }
        r.value.reset(new PageRing(std::move(eng), h, std::move(sink)));
        return r;
    }
    ~PageRing() { pbsgpu_ring_destroy(r_); }

    Result<uint32_t> Open() {
        Result<uint32_t> r;
        const int st = pbsgpu_ring_open(r_, &r.value);
        if (st != PBSGPU_OK) r.err = errorf("ring open", st);
        return r;
    }
    // the stream's next page (device pointer, capacity); value.first == nullptr when no page is free right now
    Result<std::pair<void *, uint64_t>> Reserve(uint32_t stream) {
        Result<std::pair<void *, uint64_t>> r;
        r.value = {nullptr, 0};
        const int st = pbsgpu_ring_reserve(r_, stream, &r.value.first, &r.value.second);
        if (st != PBSGPU_OK && st != PBSGPU_E_BUSY) r.err = errorf("ring reserve", st);
        if (st == PBSGPU_E_BUSY) r.value = {nullptr, 0};
        return r;
    }
    std::string Commit(uint32_t stream, uint64_t nbytes, bool final) {
        const int st = pbsgpu_ring_commit(r_, stream, nbytes, final ? 1 : 0);
        return st == PBSGPU_OK ? std::string() : errorf("ring commit", st);
    }
    // synthetic producer (benchmarks / tests): bytes accepted
    Result<uint64_t> FillSynthetic(uint32_t stream, uint64_t seed, uint32_t kind, uint64_t nbytes, bool final) {
        Result<uint64_t> r;
        const int st = pbsgpu_ring_fill(r_, stream, seed, kind, nbytes, final ? 1 : 0, &r.value);
        if (st != PBSGPU_OK) r.err = errorf("ring fill", st);
        return r;
    }
    // enqueue what has been committed, deliver finished entries of `stream` to the sink; *done: the stream has ended
    // and everything has been delivered
    std::string Pump(uint32_t stream, bool *done) {
        int st = pbsgpu_ring_pump(r_);
        if (st != PBSGPU_OK) return errorf("ring pump", st);
        pbsgpu_record buf[256];
        for (;;) {
            uint64_t n = 0;
            int fin = 0;
            st = pbsgpu_ring_poll(r_, stream, buf, 256, &n, &fin);
            if (st != PBSGPU_OK) return errorf("ring poll", st);
            for (uint64_t i = 0; i < n; ++i) {
                datastore::ChunkInfo ci;
                ci.End = buf[i].end;
                std::memcpy(ci.Digest_.data(), buf[i].digest, 32);
                sink_(stream, ci, buf[i].size);
            }
            if (done) *done = fin != 0;
            if (n < 256) return {};
        }
    }
    std::string CloseStream(uint32_t stream) {
        const int st = pbsgpu_ring_close(r_, stream);
        return st == PBSGPU_OK ? std::string() : errorf("ring close", st);
    }
    std::string Quiesce() {
        const int st = pbsgpu_ring_quiesce(r_);
        return st == PBSGPU_OK ? std::string() : errorf("ring quiesce", st);
    }
    // Quiesce without the wait: for a writer about to sit in a blocking read (internal/tapeio/converter.go:672-680)
    std::string Park() {
        const int st = pbsgpu_ring_park(r_);
        return st == PBSGPU_OK ? std::string() : errorf("ring park", st);
    }
    // a suggested chunk boundary `offset` bytes into the stream ("a file starts here"), announced ahead of its bytes
    std::string Suggest(uint32_t stream, uint64_t offset) {
        const int st = pbsgpu_ring_suggest(r_, stream, offset);
        return st == PBSGPU_OK ? std::string() : errorf("ring suggest", st);
    }
    // {CUs of the express service (two lanes per chunk, the longest chunks), chunk size from which a chunk takes it}; {0, 0}: none
    std::pair<uint32_t, uint64_t> Express() const {
        std::pair<uint32_t, uint64_t> r{0, 0};
        (void)pbsgpu_ring_express(r_, &r.first, &r.second);
        return r;
    }

    // cumulative regime counters of the two services (ns per block step under load, shader clock): pbsgpu_ring_get_probe
    pbsgpu_ring_probe Probe() const {
        pbsgpu_ring_probe p{};
        (void)pbsgpu_ring_get_probe(r_, &p);
        return p;
    }

  private:
    PageRing(std::shared_ptr<Engine> eng, pbsgpu_ring *r, Sink sink) : eng_(std::move(eng)), r_(r), sink_(std::move(sink)) {}
    PageRing(const PageRing &) = delete;
    std::shared_ptr<Engine> eng_;
    pbsgpu_ring *r_;
    Sink sink_;
};

}  // namespace transfer

// The cross-GPU digest-set reduce (pbsgpu_comm_*): one Engine per GPU ingests its own archives (the reference runs one
// session per process, internal/tapeio/converter.go:396-439), the (digest, size) records of all ranks meet in ONE RCCL
// all-gather + device dedup. Every call is collective: all ranks, same order.
namespace distributed {

using CommID = std::array<uint8_t, PBSGPU_COMM_ID_BYTES>;

inline Result<CommID> NewCommID() {  // rank 0 only; ship the bytes to the other ranks
    Result<CommID> r;
    const int st = pbsgpu_comm_unique_id(r.value.data());
    if (st != PBSGPU_OK) r.err = errorf("comm unique id", st);
    return r;
}

inline std::string CommLastError() { return pbsgpu_comm_last_error(); }  // RCCL's own words for the last failure

class Comm {
  public:
    static Result<std::unique_ptr<Comm>> New(std::shared_ptr<Engine> eng, const CommID &id, int rank, int world) {
        Result<std::unique_ptr<Comm>> r;
        pbsgpu_comm *h = nullptr;
        const int st = pbsgpu_comm_create(eng->handle(), id.data(), rank, world, &h);
        if (st != PBSGPU_OK) {
            r.err = errorf("comm create", st);
            return r;
        }
        r.value.reset(new Comm(std::move(eng), h));
        return r;
    }
    ~Comm() { pbsgpu_comm_destroy(c_); }
    // this rank's records in, the statistics of the union out; dup[i] = an earlier record of the union has recs[i]'s digest
    Result<pbsgpu_dedup_stats> Dedup(const std::vector<pbsgpu_record> &recs, uint64_t cap_records, std::vector<uint8_t> *dup = nullptr) {
        Result<pbsgpu_dedup_stats> r;
        if (dup) dup->assign(recs.size(), 0);
        const int st = pbsgpu_digest_allgather_dedup(c_, recs.empty() ? nullptr : recs.data(), recs.size(), cap_records,
                                                     dup && !recs.empty() ? dup->data() : nullptr, &r.value);
        if (st != PBSGPU_OK) r.err = errorf("digest all-gather + dedup", st);
        return r;
    }

  private:
    Comm(std::shared_ptr<Engine> eng, pbsgpu_comm *c) : eng_(std::move(eng)), c_(c) {}
    Comm(const Comm &) = delete;
    std::shared_ptr<Engine> eng_;
    pbsgpu_comm *c_;
};

}  // namespace distributed
}  // namespace pbsgpu
