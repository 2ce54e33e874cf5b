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

// Engine owns the device state for one GPU. One goroutine at a time per Engine, like the
// single writer goroutine of the reference (internal/tapeio/converter.go:672-680).
type Engine struct{ h *C.pbsgpu_engine }

func NewEngine(device int, cfg Config, inflight int) (*Engine, error) {
	e := &Engine{}
	if err := check(C.pbsgpu_engine_create(C.int(device), &cfg.c, C.uint32_t(inflight), &e.h), "engine_create"); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(e, func(e *Engine) { e.Close() })
	return e, nil
}

func (e *Engine) Close() {
	if e.h != nil {
		C.pbsgpu_engine_destroy(e.h)
		e.h = nil
	}
}

// Stream is the payload-stream seam WriteEntryReader feeds: Write appends bytes, Poll returns
// finished (end, digest) entries in stream order, Inject mirrors ArchiveWriter.InjectChunks
// (commit_reuse.go:315-341: the open chunk is flushed, offsets skip the injected payload).
type Stream struct{ h *C.pbsgpu_stream }

func (e *Engine) NewStream(windowBytes uint64) (*Stream, error) {
	s := &Stream{}
	if err := check(C.pbsgpu_stream_create(e.h, C.uint64_t(windowBytes), &s.h), "stream_create"); err != nil {
		return nil, err
	}
	return s, nil
}

// Write implements io.Writer. The Go slice is not retained: the library copies it into its own
// pinned staging before returning (cgo pointer rule).
func (s *Stream) Write(p []byte) (int, error) {
	if len(p) == 0 {
		return 0, nil
	}
	if err := check(C.pbsgpu_stream_write(s.h, unsafe.Pointer(&p[0]), C.size_t(len(p))), "stream_write"); err != nil {
		return 0, err
	}
	return len(p), nil
}

// ReadFrom implements io.ReaderFrom without the extra copy of Write: the reader fills the
// library's pinned staging memory directly (zero-copy feed of WriteEntryReader's io.Reader).
func (s *Stream) ReadFrom(r io.Reader) (int64, error) {
	var total int64
	for {
		var buf unsafe.Pointer
		var capacity C.size_t
		if err := check(C.pbsgpu_stream_reserve(s.h, &buf, &capacity), "stream_reserve"); err != nil {
			return total, err
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

func (s *Stream) Inject(injectedBytes uint64) error {
	return check(C.pbsgpu_stream_cut(s.h, C.uint64_t(injectedBytes)), "stream_cut")
}

func (s *Stream) Finish() error { return check(C.pbsgpu_stream_finish(s.h), "stream_finish") }

func (s *Stream) Poll(max int) ([]ChunkInfo, error) {
	if max <= 0 {
		return nil, errors.New("pbsgpu: Poll(max <= 0)")
	}
	buf := make([]C.pbsgpu_record, max)
	var n C.uint64_t
	if err := check(C.pbsgpu_stream_poll(s.h, &buf[0], C.uint64_t(max), &n), "stream_poll"); err != nil {
		return nil, err
	}
	out := make([]ChunkInfo, int(n))
	for i := range out {
		out[i].End, out[i].Segment, out[i].Size = uint64(buf[i].end), uint32(buf[i].segment), uint32(buf[i].size)
		copy(out[i].Digest[:], C.GoBytes(unsafe.Pointer(&buf[i].digest[0]), 32))
	}
	return out, nil
}

func (s *Stream) Close() {
	if s.h != nil {
		C.pbsgpu_stream_destroy(s.h)
		s.h = nil
	}
}

// Chunker mirrors the module's streaming chunker: Scan returns 0 when no boundary was found
// in data (all of it consumed) or the boundary position (bytes consumed, state reset).
type Chunker struct{ h *C.pbsgpu_chunker }

func (e *Engine) NewChunker() (*Chunker, error) {
	c := &Chunker{}
	if err := check(C.pbsgpu_chunker_create(e.h, &c.h), "chunker_create"); err != nil {
		return nil, err
	}
	return c, nil
}

func (c *Chunker) Scan(data []byte) (int, error) {
	if len(data) == 0 {
		return 0, nil
	}
	var pos C.size_t
	err := check(C.pbsgpu_chunker_scan(c.h, unsafe.Pointer(&data[0]), C.size_t(len(data)), &pos), "chunker_scan")
	return int(pos), err
}

func (c *Chunker) Close() {
	if c.h != nil {
		C.pbsgpu_chunker_destroy(c.h)
		c.h = nil
	}
}

// HashFiles is verification.HashFile (internal/agent/verification/handler.go:36-68) for many
// files of one buffer: digests[i] = SHA-256(buf[offsets[i] : offsets[i]+lengths[i]]).
func (e *Engine) HashFiles(buf []byte, offsets, lengths []uint64) ([][32]byte, error) {
	n := len(offsets)
	if n == 0 || n != len(lengths) {
		return nil, errors.New("pbsgpu: HashFiles needs matching offsets/lengths")
	}
	segs := make([]C.pbsgpu_segment, n)
	for i := range segs {
		segs[i].offset, segs[i].length = C.uint64_t(offsets[i]), C.uint64_t(lengths[i])
	}
	out := make([][32]byte, n)
	var base unsafe.Pointer
	if len(buf) > 0 {
		base = unsafe.Pointer(&buf[0])
	}
	err := check(C.pbsgpu_sha256_many_host(e.h, base, C.uint64_t(len(buf)), &segs[0], C.uint32_t(n),
		(*C.uint8_t)(unsafe.Pointer(&out[0][0]))), "sha256_many_host")
	return out, err
}
