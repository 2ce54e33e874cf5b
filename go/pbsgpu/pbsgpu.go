//go:build cgo && gpuchunk

// Package pbsgpu is the cgo binding of libpbsgpu (include/pbsgpu.h), the MI355X engine that
// stands in for the chunk loop of github.com/pbs-plus/pxar (buzhash scan + per-chunk SHA-256)
// behind the writers pbs-plus drives:
//
//	internal/pxarmount/commit_orchestrate.go:143-149  buzhash.NewConfig(4 << 20) -> NewPBSStore
//	internal/tapeio/converter.go:248,399,420          buzhash.NewConfig(4 << 20) -> New{Local,PBS}Store
//	internal/pxarmount/commit_reuse.go:457            writer.WriteEntryReader(entry, tee, size)
//
// SOURCE ONLY: this image has no Go toolchain and pbs-plus builds with CGO_ENABLED=0
// (.goreleaser.yaml:30), so the binding is opt-in behind the `gpuchunk` build tag; see
// INTEGRATION.md for the replace directive and the fallback file.
package pbsgpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../pbs_plus_amd/lib -lpbsgpu -Wl,-rpath,${SRCDIR}/../../pbs_plus_amd/lib
#include <stdlib.h>
#include "pbsgpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"io"
	"runtime"
	"unsafe"
)

// Config mirrors buzhash.Config: a plain value handed to NewPBSStore / NewLocalStore /
// BackupConfig.ChunkConfig (commit_orchestrate.go:137-149, converter.go:399-438).
type Config struct {
	AvgSize, MinSize, MaxSize, WindowSize int
	BreakTestMask, BreakTestMinimum       uint32
	c                                     C.pbsgpu_config
}

// NewConfig mirrors buzhash.NewConfig(avgSize int) (Config, error).
func NewConfig(avgSize int) (Config, error) {
	var cfg Config
	if st := C.pbsgpu_config_init(C.uint64_t(avgSize), nil, &cfg.c); st != C.PBSGPU_OK {
		return Config{}, fmt.Errorf("buzhash: invalid average chunk size %d: %s", avgSize, strerror(st))
	}
	cfg.AvgSize, cfg.MinSize, cfg.MaxSize = int(cfg.c.avg), int(cfg.c.min), int(cfg.c.max)
	cfg.WindowSize = int(cfg.c.window)
	cfg.BreakTestMask, cfg.BreakTestMinimum = uint32(cfg.c.mask), uint32(cfg.c.break_min)
	return cfg, nil
}

// WithTable injects the module's own BUZHASH table (the built-in default is the casync table).
func (c Config) WithTable(t [256]uint32) Config {
	for i, v := range t {
		c.c.table[i] = C.uint32_t(v)
	}
	return c
}

func strerror(st C.int) string { return C.GoString(C.pbsgpu_strerror(st)) }

func check(st C.int, what string) error {
	if st == C.PBSGPU_OK {
		return nil
	}
	return fmt.Errorf("pbsgpu: %s: %s (hip %d)", what, strerror(st), int(C.pbsgpu_last_hip_error()))
}

// ChunkInfo is one dynamic-index entry: datastore.ChunkInfo{End, Digest}
// (internal/pxarmount/commit_reuse.go:105-115) plus the chunk size (KnownChunkRef.Size).
type ChunkInfo struct {
	End     uint64
	Digest  [32]byte
	Segment uint32
	Size    uint32
}

// Engine owns the device state for one GPU. It may be shared by any number of goroutines: batch submits,
// helper calls and any number of Streams / Chunkers created from it run concurrently (the library holds no lock
// across device waits); one Stream or Chunker is for one goroutine at a time, like the reference's writers
// (internal/tapeio/converter.go:672-680).
type Engine struct{ h *C.pbsgpu_engine }

func NewEngine(device int, cfg Config, inflight int) (*Engine, error) {
	return NewEngineOpt(device, cfg, EngineOptions{Inflight: uint32(inflight)})
}

// EngineOptions mirrors pbsgpu_engine_options (ABI v5): every tuning value an engine has, per engine — two engines of one
// process may differ (rounds 1-5 read them from process-wide PBSGPU_* environment variables). Zero values = the defaults.
type EngineOptions struct {
	Inflight            uint32  // batches in flight at once (1..16; 0 = 2)
	ShaForm             uint32  // batch-path hash kernel: 0 wave pairs, 1 single-wave lanes, 2 express (tests, A/B)
	ShaSlackPct         uint32  // percent + 1 (0 = default)
	ShaDensePct         uint32  // 0 = default 150; ^uint32(0) = never
	ResolveParMin       uint64  // 0 = default 64 MiB; ^uint64(0) = always the serial walk
	ShaManyFilesPerCore uint32  // HashFiles policy: files in flight per host core from which the GPU wins (0 = 55)
	StreamShaCUs        uint32  // the engine's own ring behind Stream: CUs of its pair service (0 = 32)
	StreamExpressCUs    uint32  // ... of its express service (0 = 8; ^uint32(0) = none)
	StreamRingSlots     uint32  // ring streams open at once (0 = 256)
	StreamCtxPool       uint32  // closed stream contexts kept for re-use (0 = 8; ^uint32(0) = none)
	StreamRingGiB       float64 // its arena (0 = 48)
	StreamPageBytes     uint64  // its page size (0 = default)
}

func NewEngineOpt(device int, cfg Config, o EngineOptions) (*Engine, error) {
	e := &Engine{}
	co := C.pbsgpu_engine_options{inflight: C.uint32_t(o.Inflight), sha_form: C.uint32_t(o.ShaForm),
		sha_slack_pct: C.uint32_t(o.ShaSlackPct), sha_dense_pct: C.uint32_t(o.ShaDensePct),
		resolve_par_min: C.uint64_t(o.ResolveParMin), sha_many_files_per_core: C.uint32_t(o.ShaManyFilesPerCore),
		stream_sha_cus: C.uint32_t(o.StreamShaCUs), stream_express_cus: C.uint32_t(o.StreamExpressCUs),
		stream_ring_slots: C.uint32_t(o.StreamRingSlots), stream_ctx_pool: C.uint32_t(o.StreamCtxPool),
		stream_ring_gib: C.double(o.StreamRingGiB), stream_page_bytes: C.uint64_t(o.StreamPageBytes)}
	if err := check(C.pbsgpu_engine_create_opt(C.int(device), &cfg.c, &co, &e.h), "engine_create"); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(e, func(e *Engine) { e.Close() })
	return e, nil
}

// Close releases the handle. Streams and Chunkers created from the engine keep the device state alive until
// they are closed too (the C library reference-counts), so finalizer order does not matter.
func (e *Engine) Close() {
	runtime.SetFinalizer(e, nil)
	if e.h != nil {
		C.pbsgpu_engine_destroy(e.h)
		e.h = nil
	}
}

func toSegments(offsets, lengths []uint64) ([]C.pbsgpu_segment, error) {
	if len(offsets) != len(lengths) {
		return nil, errors.New("pbsgpu: offsets/lengths differ in length")
	}
	segs := make([]C.pbsgpu_segment, len(offsets))
	for i := range segs {
		segs[i].offset, segs[i].length = C.uint64_t(offsets[i]), C.uint64_t(lengths[i])
	}
	return segs, nil
}

func fromRecords(buf []C.pbsgpu_record, n int) []ChunkInfo {
	out := make([]ChunkInfo, n)
	for i := range out {
		out[i].End, out[i].Segment, out[i].Size = uint64(buf[i].end), uint32(buf[i].segment), uint32(buf[i].size)
		copy(out[i].Digest[:], C.GoBytes(unsafe.Pointer(&buf[i].digest[0]), 32))
	}
	return out
}

func toRecords(in []ChunkInfo) []C.pbsgpu_record {
	out := make([]C.pbsgpu_record, len(in))
	for i, ci := range in {
		out[i].end, out[i].segment, out[i].size = C.uint64_t(ci.End), C.uint32_t(ci.Segment), C.uint32_t(ci.Size)
		for b := 0; b < 32; b++ {
			out[i].digest[b] = C.uint8_t(ci.Digest[b])
		}
	}
	return out
}

// ---- batch path: many files / segments of one buffer at once (BASELINE configs[2]) -------------------------------

// Ticket identifies an asynchronous batch.
type Ticket uint64

// ErrBusy: every in-flight ticket slot is taken; Collect one first.
var ErrBusy = errors.New("pbsgpu: all in-flight tickets used")

// Submit cuts and hashes every segment buf[offsets[i] : offsets[i]+lengths[i]] as an independent stream (fresh
// chunker state, forced cut at its end). The Go slice is copied to pinned staging before the call returns.
// suggested (optional): one ascending list of suggested boundaries per segment, relative to the segment start.
func (e *Engine) Submit(buf []byte, offsets, lengths []uint64, suggested [][]uint64) (Ticket, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	segs, err := toSegments(offsets, lengths)
	if err != nil {
		return 0, err
	}
	var base unsafe.Pointer
	if len(buf) > 0 {
		base = unsafe.Pointer(&buf[0])
	}
	var sp *C.pbsgpu_segment
	if len(segs) > 0 {
		sp = &segs[0]
	}
	var t C.uint64_t
	var st C.int
	if suggested == nil {
		st = C.pbsgpu_submit_host(e.h, base, C.uint64_t(len(buf)), sp, C.uint32_t(len(segs)), &t)
	} else {
		nseg := len(segs)
		if nseg == 0 {
			nseg = 1
		}
		if len(suggested) != nseg {
			return 0, errors.New("pbsgpu: one suggested-boundary list per segment")
		}
		idx := make([]C.uint32_t, nseg+1)
		var flat []C.uint64_t
		for i, l := range suggested {
			for _, v := range l {
				flat = append(flat, C.uint64_t(v))
			}
			idx[i+1] = C.uint32_t(len(flat))
		}
		var fp *C.uint64_t
		if len(flat) > 0 {
			fp = &flat[0]
		}
		st = C.pbsgpu_submit_host_suggested(e.h, base, C.uint64_t(len(buf)), sp, C.uint32_t(len(segs)), fp, &idx[0], &t)
	}
	runtime.KeepAlive(buf)
	if st == C.PBSGPU_E_BUSY {
		return 0, ErrBusy
	}
	return Ticket(t), check(st, "submit_host")
}

// Done reports, without blocking, whether Collect would return at once.
func (e *Engine) Done(t Ticket) (bool, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	var d C.int
	err := check(C.pbsgpu_ticket_done(e.h, C.uint64_t(t), &d), "ticket_done")
	return d != 0, err
}

// Collect waits for the batch and returns its records ordered by (segment, end); the ticket is released.
func (e *Engine) Collect(t Ticket) ([]ChunkInfo, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	var n C.uint64_t
	if err := check(C.pbsgpu_wait(e.h, C.uint64_t(t), &n), "wait"); err != nil {
		return nil, err
	}
	buf := make([]C.pbsgpu_record, int(n)+1)
	if err := check(C.pbsgpu_collect(e.h, C.uint64_t(t), &buf[0], n, &n), "collect"); err != nil {
		return nil, err
	}
	return fromRecords(buf, int(n)), nil
}

// ---- digest set, dynamic index, reuse planner ---------------------------------------------------------------------

// DedupStats mirrors pbsgpu_dedup_stats.
type DedupStats struct{ Records, Unique, TotalBytes, UniqueBytes uint64 }

// Dedup flags every record whose digest already occurred at a lower index (device sort + compare): the digest-set
// reduce of cross-file duplicate detection, run on the all-gathered records of all GPUs.
func (e *Engine) Dedup(recs []ChunkInfo) ([]bool, DedupStats, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	if len(recs) == 0 {
		return nil, DedupStats{}, nil
	}
	cr := toRecords(recs)
	dup := make([]C.uint8_t, len(recs))
	var st C.pbsgpu_dedup_stats
	if err := check(C.pbsgpu_dedup_host(e.h, &cr[0], C.uint64_t(len(cr)), &dup[0], &st), "dedup_host"); err != nil {
		return nil, DedupStats{}, err
	}
	out := make([]bool, len(recs))
	for i := range out {
		out[i] = dup[i] != 0
	}
	return out, DedupStats{uint64(st.nrecords), uint64(st.nunique), uint64(st.total_bytes), uint64(st.unique_bytes)}, nil
}

// EncodeDynamicIndex is datastore.NewDynamicIndexWriter(ctime).Add(end, digest)...Finish()
// (internal/pxarmount/commit_bottleneck_test.go:773-793): the .didx image of one stream's records.
func (e *Engine) EncodeDynamicIndex(recs []ChunkInfo, uuid [16]byte, ctime int64) ([]byte, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	var nb C.uint64_t
	C.pbsgpu_didx_size(C.uint64_t(len(recs)), &nb)
	out := make([]byte, int(nb))
	cr := toRecords(recs)
	var rp *C.pbsgpu_record
	if len(cr) > 0 {
		rp = &cr[0]
	}
	err := check(C.pbsgpu_didx_encode(e.h, rp, C.uint64_t(len(cr)), (*C.uint8_t)(unsafe.Pointer(&uuid[0])), C.int64_t(ctime),
		(*C.uint8_t)(unsafe.Pointer(&out[0])), nb), "didx_encode")
	return out, err
}

// ParseDynamicIndex is datastore.ParseDynamicIndex (internal/pxarmount/commit_orchestrate.go:219).
func ParseDynamicIndex(blob []byte) (recs []ChunkInfo, ctime int64, csum [32]byte, err error) {
	if len(blob) == 0 {
		return nil, 0, csum, errors.New("pbsgpu: empty index")
	}
	var n C.uint64_t
	var ct C.int64_t
	st := C.pbsgpu_didx_decode((*C.uint8_t)(unsafe.Pointer(&blob[0])), C.uint64_t(len(blob)), nil, 0, &n, &ct,
		(*C.uint8_t)(unsafe.Pointer(&csum[0])))
	if st != C.PBSGPU_OK && st != C.PBSGPU_E_CAPACITY {
		return nil, 0, csum, check(st, "didx_decode")
	}
	buf := make([]C.pbsgpu_record, int(n)+1)
	if err = check(C.pbsgpu_didx_decode((*C.uint8_t)(unsafe.Pointer(&blob[0])), C.uint64_t(len(blob)), &buf[0], n, &n, &ct,
		(*C.uint8_t)(unsafe.Pointer(&csum[0]))), "didx_decode"); err != nil {
		return nil, 0, csum, err
	}
	return fromRecords(buf, int(n)), int64(ct), csum, nil
}

// ReuseChunk mirrors the chunks lookupDynamicEntries returns (internal/pxarmount/commit_reuse.go:84-135).
type ReuseChunk struct {
	Size, Padding, EndOffset uint64
	Digest                   [32]byte
}

// LookupDynamicEntries is lookupDynamicEntries(idx, rangeStart, rangeEnd).
func LookupDynamicEntries(idx []ChunkInfo, rangeStart, rangeEnd uint64) (chunks []ReuseChunk, startPadding, endPadding uint64, err error) {
	cr := toRecords(idx)
	var rp *C.pbsgpu_record
	if len(cr) > 0 {
		rp = &cr[0]
	}
	buf := make([]C.pbsgpu_reuse_chunk, len(idx)+1)
	var n, sp, ep C.uint64_t
	if err = check(C.pbsgpu_reuse_lookup(rp, C.uint64_t(len(cr)), C.uint64_t(rangeStart), C.uint64_t(rangeEnd), &buf[0],
		C.uint64_t(len(buf)), &n, &sp, &ep), "reuse_lookup"); err != nil {
		return nil, 0, 0, err
	}
	chunks = make([]ReuseChunk, int(n))
	for i := range chunks {
		chunks[i].Size, chunks[i].Padding, chunks[i].EndOffset = uint64(buf[i].size), uint64(buf[i].padding), uint64(buf[i].end_offset)
		copy(chunks[i].Digest[:], C.GoBytes(unsafe.Pointer(&buf[i].digest[0]), 32))
	}
	return chunks, uint64(sp), uint64(ep), nil
}

// ShouldReuse is shouldReuse with chunkPaddingThreshold (commit_reuse.go:152-183, commit_types.go:14).
func ShouldReuse(idx []ChunkInfo, rangeStart, rangeEnd uint64, saved *ReuseChunk, threshold float64) (bool, error) {
	cr := toRecords(idx)
	var rp *C.pbsgpu_record
	if len(cr) > 0 {
		rp = &cr[0]
	}
	var sv *C.pbsgpu_reuse_chunk
	var tmp C.pbsgpu_reuse_chunk
	if saved != nil {
		tmp.size, tmp.padding, tmp.end_offset = C.uint64_t(saved.Size), C.uint64_t(saved.Padding), C.uint64_t(saved.EndOffset)
		for b := 0; b < 32; b++ {
			tmp.digest[b] = C.uint8_t(saved.Digest[b])
		}
		sv = &tmp
	}
	var r C.int
	err := check(C.pbsgpu_reuse_should(rp, C.uint64_t(len(cr)), C.uint64_t(rangeStart), C.uint64_t(rangeEnd), sv,
		C.double(threshold), &r), "reuse_should")
	return r != 0, err
}

// ---- payload-stream writer -----------------------------------------------------------------------------------------

// Stream is the payload-stream seam WriteEntryReader feeds: Write appends bytes, Poll returns
// finished (end, digest) entries in stream order, Inject mirrors ArchiveWriter.InjectChunks
// (commit_reuse.go:315-341: the open chunk is flushed, offsets skip the injected payload).
type Stream struct {
	h   *C.pbsgpu_stream
	eng *Engine // keeps the Go handle reachable while the stream lives
}

func (e *Engine) NewStream(windowBytes uint64) (*Stream, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	s := &Stream{eng: e}
	if err := check(C.pbsgpu_stream_create(e.h, C.uint64_t(windowBytes), &s.h), "stream_create"); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(s, func(s *Stream) { s.Close() })
	return s, nil
}

// Write implements io.Writer. The Go slice is not retained: the library copies it into its own
// pinned staging before returning (cgo pointer rule).
func (s *Stream) Write(p []byte) (int, error) {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	if len(p) == 0 {
		return 0, nil
	}
	err := check(C.pbsgpu_stream_write(s.h, unsafe.Pointer(&p[0]), C.size_t(len(p))), "stream_write")
	runtime.KeepAlive(p)
	if err != nil {
		return 0, err
	}
	return len(p), nil
}

// ReadFrom implements io.ReaderFrom without the extra copy of Write: the reader fills the
// library's pinned staging memory directly (zero-copy feed of WriteEntryReader's io.Reader).
func (s *Stream) ReadFrom(r io.Reader) (int64, error) {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	var total int64
	for {
		var buf unsafe.Pointer
		var capacity C.size_t
		if err := check(C.pbsgpu_stream_reserve(s.h, &buf, &capacity), "stream_reserve"); err != nil {
			return total, err
		}
		if capacity == 0 { // inside an entry whose announced size has been reached
			_ = C.pbsgpu_stream_commit(s.h, 0)
			return total, nil
		}
		n, rerr := io.ReadFull(r, unsafe.Slice((*byte)(buf), int(capacity)))
		if err := check(C.pbsgpu_stream_commit(s.h, C.size_t(n)), "stream_commit"); err != nil {
			return total, err
		}
		total += int64(n)
		if rerr == io.EOF || rerr == io.ErrUnexpectedEOF {
			return total, nil
		}
		if rerr != nil {
			return total, rerr
		}
	}
}

// WriteEntryReader is the payload half of ArchiveWriter.WriteEntryReader(entry, r, size): 16-byte payload header,
// exactly size bytes from r (zero-copy), per-file XXH3-64 tee. Returns the entry's payload offset (the PAYLOAD_REF /
// WriteEntryRef value, commit_walk.go:455) and the file index under which PollFiles reports the hash — what
// writeBackedFile keeps in backedHashes[path] (commit_reuse.go:450-461).
func (s *Stream) WriteEntryReader(r io.Reader, size uint64) (payloadOffset, fileIndex uint64, err error) {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	var off, idx C.uint64_t
	if err = check(C.pbsgpu_stream_begin_entry(s.h, nil, C.uint64_t(size), &off), "stream_begin_entry"); err != nil {
		return 0, 0, err
	}
	n, err := s.ReadFrom(io.LimitReader(r, int64(size)))
	if err != nil {
		return uint64(off), 0, err
	}
	if uint64(n) != size {
		return uint64(off), 0, io.ErrUnexpectedEOF
	}
	err = check(C.pbsgpu_stream_end_entry(s.h, &idx), "stream_end_entry")
	return uint64(off), uint64(idx), err
}

// BeginFile / EndFile bracket a file body for the XXH3 tee when the caller writes the bytes itself.
func (s *Stream) BeginFile() error {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	return check(C.pbsgpu_stream_begin_file(s.h), "stream_begin_file")
}
func (s *Stream) EndFile() (uint64, error) {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	var idx C.uint64_t
	err := check(C.pbsgpu_stream_end_file(s.h, &idx), "stream_end_file")
	return uint64(idx), err
}

// FileHash is one finished file of the tee.
type FileHash struct{ Index, Size, XXH3 uint64 }

func (s *Stream) PollFiles(max int) ([]FileHash, error) {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	if max <= 0 {
		return nil, errors.New("pbsgpu: PollFiles(max <= 0)")
	}
	buf := make([]C.pbsgpu_file_hash, max)
	var n C.uint64_t
	if err := check(C.pbsgpu_stream_poll_files(s.h, &buf[0], C.uint64_t(max), &n), "stream_poll_files"); err != nil {
		return nil, err
	}
	out := make([]FileHash, int(n))
	for i := range out {
		out[i] = FileHash{uint64(buf[i].index), uint64(buf[i].size), uint64(buf[i].xxh3)}
	}
	return out, nil
}

// WriteMarker appends the payload start (tail = false) or tail marker.
func (s *Stream) WriteMarker(tail bool) error {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	t := C.int(0)
	if tail {
		t = 1
	}
	return check(C.pbsgpu_stream_write_marker(s.h, nil, t), "stream_write_marker")
}

// Inject mirrors InjectChunks: forced cut, the payload position advances by the injected sizes.
func (s *Stream) Inject(injectedBytes uint64) error {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	return check(C.pbsgpu_stream_cut(s.h, C.uint64_t(injectedBytes)), "stream_cut")
}

// PayloadPosition is Encoder().PayloadPosition() (commit_reuse.go:265): written + injected bytes.
func (s *Stream) PayloadPosition() uint64 {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	var n C.uint64_t
	C.pbsgpu_stream_position(s.h, &n)
	return uint64(n)
}

// SuggestBoundary suggests a chunk boundary at the current position (payload chunker; a file starts here).
func (s *Stream) SuggestBoundary() error {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	return check(C.pbsgpu_stream_suggest(s.h, C.uint64_t(s.PayloadPosition())), "stream_suggest")
}

func (s *Stream) Finish() error {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	return check(C.pbsgpu_stream_finish(s.h), "stream_finish")
}

// FinishBegin closes the input without waiting for the last chunks' digests: the goroutine goes on with its next
// archive while this one drains; Poll keeps delivering, Done reports when the last entry is out.
func (s *Stream) FinishBegin() error {
	defer runtime.KeepAlive(s)
	return check(C.pbsgpu_stream_finish_begin(s.h), "stream_finish_begin")
}

// Done never blocks: true once every entry of a stream closed by Finish / FinishBegin can be polled.
func (s *Stream) Done() (bool, error) {
	defer runtime.KeepAlive(s)
	var d C.int
	if err := check(C.pbsgpu_stream_done(s.h, &d), "stream_done"); err != nil {
		return false, err
	}
	return d != 0, nil
}

func (s *Stream) Poll(max int) ([]ChunkInfo, error) {
	defer runtime.KeepAlive(s) // the finalizer must not run Close while the C call is executing
	if max <= 0 {
		return nil, errors.New("pbsgpu: Poll(max <= 0)")
	}
	buf := make([]C.pbsgpu_record, max)
	var n C.uint64_t
	if err := check(C.pbsgpu_stream_poll(s.h, &buf[0], C.uint64_t(max), &n), "stream_poll"); err != nil {
		return nil, err
	}
	return fromRecords(buf, int(n)), nil
}

func (s *Stream) Close() {
	runtime.SetFinalizer(s, nil)
	if s.h != nil {
		C.pbsgpu_stream_destroy(s.h)
		s.h = nil
	}
	s.eng = nil
}

// Chunker mirrors the module's streaming chunker: Scan returns 0 when no boundary was found
// in data (all of it consumed) or the boundary position (bytes consumed, state reset).
type Chunker struct {
	h   *C.pbsgpu_chunker
	eng *Engine
}

func (e *Engine) NewChunker() (*Chunker, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	c := &Chunker{eng: e}
	if err := check(C.pbsgpu_chunker_create(e.h, &c.h), "chunker_create"); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(c, func(c *Chunker) { c.Close() })
	return c, nil
}

func (c *Chunker) Scan(data []byte) (int, error) {
	defer runtime.KeepAlive(c) // the finalizer must not run Close while the C call is executing
	if len(data) == 0 {
		return 0, nil
	}
	var pos C.size_t
	err := check(C.pbsgpu_chunker_scan(c.h, unsafe.Pointer(&data[0]), C.size_t(len(data)), &pos), "chunker_scan")
	runtime.KeepAlive(data)
	return int(pos), err
}

func (c *Chunker) Reset() error {
	defer runtime.KeepAlive(c) // the finalizer must not run Close while the C call is executing
	return check(C.pbsgpu_chunker_reset(c.h), "chunker_reset")
}

func (c *Chunker) Close() {
	runtime.SetFinalizer(c, nil)
	if c.h != nil {
		C.pbsgpu_chunker_destroy(c.h)
		c.h = nil
	}
	c.eng = nil
}

// ---- whole-file hashes (verification) -------------------------------------------------------------------------------

// HashFiles is verification.HashFile (internal/agent/verification/handler.go:36-68) for many
// files of one buffer: digests[i] = SHA-256(buf[offsets[i] : offsets[i]+lengths[i]]). One GPU lane per file:
// worth it for MANY files per call (see DESIGN.md, crossover), not for a handful of huge ones.
//
// POLICY: SHA-256 is serial inside a file (one GPU lane per file, 0.036 GiB/s each), so a batch only beats the host's
// SHA-NI cores with more than ~55 files per core in flight. The reference's verify job keeps FOUR files in flight
// (internal/server/verification/job.go:493) — routed here it would be ~50x slower than sha256-simd. HashFiles therefore
// returns ErrHostFaster (and hashes nothing) when pbsgpu_sha256_many_pays says the host wins for this batch size on
// runtime.NumCPU() cores; HashFilesForced skips the question.
func (e *Engine) HashFiles(buf []byte, offsets, lengths []uint64) ([][32]byte, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	var pays C.int
	if err := check(C.pbsgpu_sha256_many_pays(e.h, C.uint32_t(len(offsets)), C.uint32_t(runtime.NumCPU()), &pays), "sha256_many_pays"); err != nil {
		return nil, err
	}
	if pays == 0 {
		return nil, ErrHostFaster
	}
	return e.HashFilesForced(buf, offsets, lengths)
}

// ErrHostFaster: the batch is too small for the GPU to beat the host's SHA-NI cores; hash on the host.
var ErrHostFaster = errors.New("pbsgpu: too few files in flight for the GPU to beat host SHA-256; hash on the host")

// Trim releases device memory the engine only keeps for re-use (window buffers of closed streams).
func (e *Engine) Trim() (uint64, error) {
	defer runtime.KeepAlive(e)
	var n C.uint64_t
	err := check(C.pbsgpu_engine_trim(e.h, &n), "engine_trim")
	return uint64(n), err
}

// HashFilesForced hashes on the GPU whatever the batch size.
func (e *Engine) HashFilesForced(buf []byte, offsets, lengths []uint64) ([][32]byte, error) {
	defer runtime.KeepAlive(e)
	segs, err := toSegments(offsets, lengths)
	if err != nil || len(segs) == 0 {
		return nil, errors.New("pbsgpu: HashFiles needs matching, non-empty offsets/lengths")
	}
	out := make([][32]byte, len(segs))
	var base unsafe.Pointer
	if len(buf) > 0 {
		base = unsafe.Pointer(&buf[0])
	}
	err = check(C.pbsgpu_sha256_many_host(e.h, base, C.uint64_t(len(buf)), &segs[0], C.uint32_t(len(segs)),
		(*C.uint8_t)(unsafe.Pointer(&out[0][0]))), "sha256_many_host")
	runtime.KeepAlive(buf)
	return out, err
}

// XXH3Files is the per-file XXH3-64 of verifyBackedFileHashes (internal/pxarmount/commit_orchestrate.go:485-562).
func (e *Engine) XXH3Files(buf []byte, offsets, lengths []uint64) ([]uint64, error) {
	defer runtime.KeepAlive(e) // the finalizer must not run Close while the C call is executing
	segs, err := toSegments(offsets, lengths)
	if err != nil || len(segs) == 0 {
		return nil, errors.New("pbsgpu: XXH3Files needs matching, non-empty offsets/lengths")
	}
	out := make([]uint64, len(segs))
	var base unsafe.Pointer
	if len(buf) > 0 {
		base = unsafe.Pointer(&buf[0])
	}
	err = check(C.pbsgpu_xxh3_many_host(e.h, base, C.uint64_t(len(buf)), &segs[0], C.uint32_t(len(segs)),
		(*C.uint64_t)(unsafe.Pointer(&out[0]))), "xxh3_many_host")
	runtime.KeepAlive(buf)
	return out, err
}

// ---- page ring: several archives at once, page-granular memory release ---------------------------------------------

// Ring runs the chunk loop for SEVERAL payload streams through one device arena (pbsgpu_ring_*): memory is given
// back page by page as soon as the chunks touching a page have been read by the persistent SHA-256 service, not when a
// whole batch has been hashed. One goroutine drives a Ring (like one goroutine owns a writer,
// internal/tapeio/converter.go:672-680); bytes reach a stream's pages through Reserve/Commit (a device pointer for
// a DMA, a peer GPU or a kernel). HOST bytes go through PayloadStream (pbsgpu_stream_*), which is a client of the engine's
// own ring: pinned staging, H2D straight into a reserved page, the same cut rounds and SHA-256 service.
// The goroutine may block in a Read for as long as it likes: a ring that is not called for the idle timeout stops its idle
// service by itself and the next Pump starts it again (nothing is lost); Park does so at once. No byte content fails
// a stream: candidate-dense data (a crafted short period) is cut exactly, by on-demand re-scans inside the cut round.
type Ring struct {
	h   *C.pbsgpu_ring
	eng *Engine
}

// RingOptions mirrors pbsgpu_ring_options; zero values select the library defaults.
type RingOptions struct {
	ArenaBytes, PageBytes         uint64
	MaxStreams, ShaCUs, RoundPages uint32
	// ExpressCUs run the two-lanes-per-chunk SHA-256 form on the long chunks (>= 5/8 of the maximum size): the chain of a
	// chunk ~1.4x faster at ~0.65 of the throughput per CU. For rings whose latency matters more than their CU-time.
	ExpressCUs uint32
	// ABI v5: what used to be PBSGPU_RING_* environment variables (pbsgpu.h); zero = default, RingOff = "none"
	MinRoundPages, MaxInflight, LongBytes, LongLoBytes, LongSpill, PollEvery, Flags uint32
	BacklogMiB, LoneDeferMs, IdleTimeoutS, AutoparkMs                                float64
	// LanesCUs of the pair service's share run the LANES service (one lane per chunk) for chunks of at most ShortBytes
	// (0 = 3/2 of the average chunk size). Off by default: measured neutral on bulk rings (DESIGN.md 5.5).
	LanesCUs   uint32
	ShortBytes uint64
}

// RingOff expresses "none" for RingOptions fields whose zero value means "default" (PBSGPU_RING_OFF).
const RingOff = ^uint32(0)

func (e *Engine) NewRing(o RingOptions) (*Ring, error) {
	defer runtime.KeepAlive(e)
	co := C.pbsgpu_ring_options{arena_bytes: C.uint64_t(o.ArenaBytes), page_bytes: C.uint64_t(o.PageBytes),
		max_streams: C.uint32_t(o.MaxStreams), sha_cus: C.uint32_t(o.ShaCUs), round_pages: C.uint32_t(o.RoundPages),
		express_cus: C.uint32_t(o.ExpressCUs), min_round_pages: C.uint32_t(o.MinRoundPages),
		max_inflight: C.uint32_t(o.MaxInflight), long_bytes: C.uint32_t(o.LongBytes), long_lo_bytes: C.uint32_t(o.LongLoBytes),
		long_spill: C.uint32_t(o.LongSpill), poll_every: C.uint32_t(o.PollEvery), flags: C.uint32_t(o.Flags),
		backlog_mib: C.double(o.BacklogMiB), lone_defer_ms: C.double(o.LoneDeferMs), idle_timeout_s: C.double(o.IdleTimeoutS),
		autopark_ms: C.double(o.AutoparkMs), lanes_cus: C.uint32_t(o.LanesCUs), short_bytes: C.uint64_t(o.ShortBytes)}
	r := &Ring{eng: e}
	if err := check(C.pbsgpu_ring_create(e.h, &co, &r.h), "ring_create"); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(r, func(r *Ring) { r.Close() })
	return r, nil
}

// Open starts a new stream (fresh chunker state); ErrBusy when MaxStreams are open.
func (r *Ring) Open() (uint32, error) {
	defer runtime.KeepAlive(r)
	var s C.uint32_t
	st := C.pbsgpu_ring_open(r.h, &s)
	if st == C.PBSGPU_E_BUSY {
		return 0, ErrBusy
	}
	return uint32(s), check(st, "ring_open")
}

// Reserve returns the device address and capacity of the stream's next page; ErrBusy when no page is free right now.
func (r *Ring) Reserve(stream uint32) (dptr uintptr, capacity uint64, err error) {
	defer runtime.KeepAlive(r)
	var p unsafe.Pointer
	var c C.uint64_t
	st := C.pbsgpu_ring_reserve(r.h, C.uint32_t(stream), &p, &c)
	if st == C.PBSGPU_E_BUSY {
		return 0, 0, ErrBusy
	}
	return uintptr(p), uint64(c), check(st, "ring_reserve")
}

// Commit hands the first nbytes of the reserved page to the ring; final ends the stream.
func (r *Ring) Commit(stream uint32, nbytes uint64, final bool) error {
	defer runtime.KeepAlive(r)
	f := C.int(0)
	if final {
		f = 1
	}
	return check(C.pbsgpu_ring_commit(r.h, C.uint32_t(stream), C.uint64_t(nbytes), f), "ring_commit")
}

// FillSynthetic is the benchmark producer (pbsgpu_ring_fill); returns the bytes accepted.
func (r *Ring) FillSynthetic(stream uint32, seed uint64, kind uint32, nbytes uint64, final bool) (uint64, error) {
	defer runtime.KeepAlive(r)
	f := C.int(0)
	if final {
		f = 1
	}
	var taken C.uint64_t
	err := check(C.pbsgpu_ring_fill(r.h, C.uint32_t(stream), C.uint64_t(seed), C.uint32_t(kind), C.uint64_t(nbytes), f, &taken),
		"ring_fill")
	return uint64(taken), err
}

// Pump enqueues the committed pages as cut rounds and collects finished rounds and freed pages; never blocks.
func (r *Ring) Pump() error {
	defer runtime.KeepAlive(r)
	return check(C.pbsgpu_ring_pump(r.h), "ring_pump")
}

// Poll returns up to max finished (end, digest) entries of the stream in stream order; done once the stream has ended
// and every entry has been handed out.
func (r *Ring) Poll(stream uint32, max int) (recs []ChunkInfo, done bool, err error) {
	defer runtime.KeepAlive(r)
	if max <= 0 {
		return nil, false, errors.New("pbsgpu: Poll(max <= 0)")
	}
	buf := make([]C.pbsgpu_record, max)
	var n C.uint64_t
	var fin C.int
	if err = check(C.pbsgpu_ring_poll(r.h, C.uint32_t(stream), &buf[0], C.uint64_t(max), &n, &fin), "ring_poll"); err != nil {
		return nil, false, err
	}
	return fromRecords(buf, int(n)), fin != 0, nil
}

// PollAny returns finished entries of ANY open stream (Segment = stream id, each stream's entries in order) and the ids
// of the streams that have just handed out their last entry — for jobs with one stream per file.
func (r *Ring) PollAny(max, maxFinished int) (recs []ChunkInfo, finished []uint32, err error) {
	defer runtime.KeepAlive(r)
	if max <= 0 || maxFinished <= 0 {
		return nil, nil, errors.New("pbsgpu: PollAny(max <= 0)")
	}
	buf := make([]C.pbsgpu_record, max)
	fin := make([]C.uint32_t, maxFinished)
	var n C.uint64_t
	var nf C.uint32_t
	if err = check(C.pbsgpu_ring_poll_any(r.h, &buf[0], C.uint64_t(max), &n, &fin[0], C.uint32_t(maxFinished), &nf), "ring_poll_any"); err != nil {
		return nil, nil, err
	}
	finished = make([]uint32, int(nf))
	for i := range finished {
		finished[i] = uint32(fin[i])
	}
	return fromRecords(buf, int(n)), finished, nil
}

// CloseStream releases a finished, fully polled stream's slot.
func (r *Ring) CloseStream(stream uint32) error {
	defer runtime.KeepAlive(r)
	return check(C.pbsgpu_ring_close(r.h, C.uint32_t(stream)), "ring_close")
}

// Quiesce waits until everything enqueued is hashed and stops the service kernel; the next Pump restarts it.
func (r *Ring) Quiesce() error {
	defer runtime.KeepAlive(r)
	return check(C.pbsgpu_ring_quiesce(r.h), "ring_quiesce")
}

// Park is Quiesce without the wait: the service ends by itself once it has hashed what is enqueued and the next Pump
// starts a new one. Call it before the driving goroutine sits in a blocking Read (internal/tapeio/converter.go:672-680)
// if hipFree / device-wide synchronisation elsewhere in the process must not wait for this ring. Not calling it is safe
// too: a ring that is not called for PBSGPU_RING_IDLE_TIMEOUT_S stops its idle service on its own and loses nothing.
func (r *Ring) Park() error {
	defer runtime.KeepAlive(r)
	return check(C.pbsgpu_ring_park(r.h), "ring_park")
}

// Suggest announces a suggested chunk boundary `offset` bytes into the stream (ascending, ahead of the bytes around it).
func (r *Ring) Suggest(stream uint32, offset uint64) error {
	defer runtime.KeepAlive(r)
	return check(C.pbsgpu_ring_suggest(r.h, C.uint32_t(stream), C.uint64_t(offset)), "ring_suggest")
}

// RingStats mirrors the counters of pbsgpu_ring_stats a caller sizes its feed with.
type RingStats struct {
	PageBytes, BytesEnqueued, Chunks uint64
	PagesTotal, PagesFree, Rounds    uint32
	ServiceMsLast                    float64
}

func (r *Ring) Stats() (RingStats, error) {
	defer runtime.KeepAlive(r)
	var st C.pbsgpu_ring_stats
	if err := check(C.pbsgpu_ring_get_stats(r.h, &st), "ring_get_stats"); err != nil {
		return RingStats{}, err
	}
	return RingStats{uint64(st.page_bytes), uint64(st.bytes_enqueued), uint64(st.chunks), uint32(st.pages_total),
		uint32(st.pages_free), uint32(st.rounds), float64(st.service_ms_last)}, nil
}

// Express reports the CUs of the ring's express service (0 = none) and the chunk size from which a chunk takes it.
func (r *Ring) Express() (cus uint32, longBytes uint64, err error) {
	defer runtime.KeepAlive(r)
	var c C.uint32_t
	var l C.uint64_t
	if err = check(C.pbsgpu_ring_express(r.h, &c, &l), "ring_express"); err != nil {
		return 0, 0, err
	}
	return uint32(c), uint64(l), nil
}

// RingProbe: cumulative regime counters of the ring's two SHA-256 services (pbsgpu_ring_get_probe). Ticks / Steps x 10 = ns
// per block step of a chain under load (an express step is two blocks), Cycles / Ticks x 100 = the shader clock in MHz.
type RingProbe struct {
	PairSteps, PairCycles, PairTicks          uint64
	ExpressSteps, ExpressCycles, ExpressTicks uint64
}

// Probe reads the counters (safe while the service runs); read twice and subtract to look at a phase.
func (r *Ring) Probe() (RingProbe, error) {
	defer runtime.KeepAlive(r)
	var p C.pbsgpu_ring_probe
	if err := check(C.pbsgpu_ring_get_probe(r.h, &p), "ring_get_probe"); err != nil {
		return RingProbe{}, err
	}
	return RingProbe{uint64(p.pair_steps), uint64(p.pair_cycles), uint64(p.pair_ticks), uint64(p.express_steps),
		uint64(p.express_cycles), uint64(p.express_ticks)}, nil
}

func (r *Ring) Close() {
	runtime.SetFinalizer(r, nil)
	if r.h != nil {
		C.pbsgpu_ring_destroy(r.h)
		r.h = nil
	}
	r.eng = nil
}

// ---- multi-GPU digest-set reduce (RCCL over xGMI, behind the C ABI) ---------------------------------------------------

// CommIDBytes is the size of the opaque id rank 0 creates and ships to the other ranks (over the agent's own RPC).
const CommIDBytes = C.PBSGPU_COMM_ID_BYTES

// Comm is one rank's end of the cross-GPU digest-set reduce (pbsgpu_comm_*): the path shards at archive granularity with
// no data-path collective — one Engine per GPU ingests its own streams (internal/tapeio/converter.go:396-439: one session
// per process) — and the (digest, size) records of all ranks meet in ONE all-gather + device dedup for cross-file
// duplicate detection. Every method is collective: all ranks call it, in the same order.
type Comm struct {
	h   *C.pbsgpu_comm
	eng *Engine
}

// NewCommID is called on rank 0 only.
func NewCommID() ([CommIDBytes]byte, error) {
	var id [CommIDBytes]byte
	err := check(C.pbsgpu_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))), "comm_unique_id")
	return id, err
}

func (e *Engine) NewComm(id [CommIDBytes]byte, rank, world int) (*Comm, error) {
	defer runtime.KeepAlive(e)
	c := &Comm{eng: e}
	if err := check(C.pbsgpu_comm_create(e.h, (*C.uint8_t)(unsafe.Pointer(&id[0])), C.int(rank), C.int(world), &c.h), "comm_create"); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(c, (*Comm).Close)
	return c, nil
}

// Dedup contributes this rank's records (at most capRecords, the same bound on every rank: bytes per rank / minimum chunk
// size) and returns, for each of them, whether an earlier record of the union carries the same digest, plus the statistics
// of the union (identical on every rank).
func (c *Comm) Dedup(recs []ChunkInfo, capRecords uint64) ([]bool, DedupStats, error) {
	defer runtime.KeepAlive(c)
	var st C.pbsgpu_dedup_stats
	var rp *C.pbsgpu_record
	var dp *C.uint8_t
	cr := toRecords(recs)
	dup := make([]C.uint8_t, len(recs))
	if len(recs) > 0 {
		rp, dp = &cr[0], &dup[0]
	}
	if err := check(C.pbsgpu_digest_allgather_dedup(c.h, rp, C.uint64_t(len(recs)), C.uint64_t(capRecords), dp, &st), "digest_allgather_dedup"); err != nil {
		return nil, DedupStats{}, err
	}
	out := make([]bool, len(recs))
	for i := range out {
		out[i] = dup[i] != 0
	}
	return out, DedupStats{uint64(st.nrecords), uint64(st.nunique), uint64(st.total_bytes), uint64(st.unique_bytes)}, nil
}

// SplitPlan says which bytes of ONE stream of totalLen bytes rank `rank` of `world` owns, [ownStart, ownEnd), and which it
// must hold in device memory, [lo, hi): 63 bytes of window halo to the left, one maximum chunk to the right (arithmetic only).
func SplitPlan(totalLen uint64, world, rank int, maxChunk uint32) (ownStart, ownEnd, lo, hi uint64, err error) {
	var a, b, l, h C.uint64_t
	err = check(C.pbsgpu_split_plan(C.uint64_t(totalLen), C.int(world), C.int(rank), C.uint32_t(maxChunk), &a, &b, &l, &h), "split_plan")
	return uint64(a), uint64(b), uint64(l), uint64(h), err
}

// SplitStream cuts and hashes ONE stream of totalLen bytes that is split over the ranks of the communicator (collective;
// BASELINE configs[1] at N > 1 when the stream does not fit one GPU). local = DEVICE pointer to this rank's bytes [lo, hi) of
// SplitPlan. Every rank gets the whole stream's records, in stream order.
func (c *Comm) SplitStream(local unsafe.Pointer, totalLen uint64, capRecords uint64) ([]ChunkInfo, error) {
	defer runtime.KeepAlive(c)
	buf := make([]C.pbsgpu_record, capRecords+1)
	var n C.uint64_t
	if err := check(C.pbsgpu_comm_split_stream(c.h, local, C.uint64_t(totalLen), &buf[0], C.uint64_t(capRecords), &n), "comm_split_stream"); err != nil {
		return nil, err
	}
	return fromRecords(buf, int(n)), nil
}

// CommLastError is what RCCL reported when a communicator call of this process last failed ("" = nothing yet): a
// PBSGPU_E_HIP from NewComm / Dedup does not say whether the bootstrap found no network interface or the device faulted.
func CommLastError() string { return C.GoString(C.pbsgpu_comm_last_error()) }

func (c *Comm) Close() {
	runtime.SetFinalizer(c, nil)
	if c.h != nil {
		C.pbsgpu_comm_destroy(c.h)
		c.h = nil
	}
	c.eng = nil
}

// DedupDevice flags duplicates among n records that already are in device memory (the receive buffer of an RCCL
// all-gather): the digest-set reduce without a host round trip of the set.
func (e *Engine) DedupDevice(drecs uintptr, n uint64) ([]bool, DedupStats, error) {
	defer runtime.KeepAlive(e)
	if n == 0 {
		return nil, DedupStats{}, nil
	}
	dup := make([]C.uint8_t, n)
	var st C.pbsgpu_dedup_stats
	if err := check(C.pbsgpu_dedup_device(e.h, unsafe.Pointer(drecs), C.uint64_t(n), &dup[0], &st), "dedup_device"); err != nil {
		return nil, DedupStats{}, err
	}
	out := make([]bool, n)
	for i := range out {
		out[i] = dup[i] != 0
	}
	return out, DedupStats{uint64(st.nrecords), uint64(st.nunique), uint64(st.total_bytes), uint64(st.unique_bytes)}, nil
}
