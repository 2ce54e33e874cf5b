//go:build !(cgo && gpuchunk)

// Without cgo or without the gpuchunk tag (the project's default CGO_ENABLED=0 builds,
// .goreleaser.yaml:30) this package only reports that the GPU engine is not compiled in, so
// the fork of github.com/pbs-plus/pxar keeps using its pure-Go chunker. Nothing here chunks.
// The file mirrors the WHOLE exported surface of pbsgpu.go (tests/test_binding_sources.py checks it), so code written
// against the GPU build still compiles in the default build and gets ErrNotBuilt at run time.
package pbsgpu

import (
	"errors"
	"io"
	"unsafe"
)

// ErrNotBuilt is returned by every entry point in non-GPU builds.
var ErrNotBuilt = errors.New("pbsgpu: built without cgo/gpuchunk; GPU engine unavailable")

var (
	ErrBusy       = errors.New("pbsgpu: all in-flight tickets used")
	ErrHostFaster = errors.New("pbsgpu: too few files in flight for the GPU to beat host SHA-256; hash on the host")
)

// CommIDBytes mirrors PBSGPU_COMM_ID_BYTES.
const CommIDBytes = 128

type (
	Config struct {
		AvgSize, MinSize, MaxSize, WindowSize int
		BreakTestMask, BreakTestMinimum       uint32
	}
	ChunkInfo struct {
		End     uint64
		Digest  [32]byte
		Segment uint32
		Size    uint32
	}
	Engine     struct{}
	Stream     struct{}
	Chunker    struct{}
	Ring       struct{}
	Comm       struct{}
	Ticket     uint64
	DedupStats struct{ Records, Unique, TotalBytes, UniqueBytes uint64 }
	ReuseChunk struct {
		Size, Padding, EndOffset uint64
		Digest                   [32]byte
	}
	FileHash    struct{ Index, Size, XXH3 uint64 }
	RingOptions struct {
		ArenaBytes, PageBytes                                                            uint64
		MaxStreams, ShaCUs, RoundPages                                                   uint32
		ExpressCUs                                                                       uint32
		MinRoundPages, MaxInflight, LongBytes, LongLoBytes, LongSpill, PollEvery, Flags uint32
		BacklogMiB, LoneDeferMs, IdleTimeoutS, AutoparkMs                                float64
		LanesCUs                                                                         uint32
		ShortBytes                                                                       uint64
	}
	EngineOptions struct {
		Inflight, ShaForm, ShaSlackPct, ShaDensePct                      uint32
		ResolveParMin                                                    uint64
		ShaManyFilesPerCore                                              uint32
		StreamShaCUs, StreamExpressCUs, StreamRingSlots, StreamCtxPool   uint32
		StreamRingGiB                                                    float64
		StreamPageBytes                                                  uint64
	}
	RingStats struct {
		PageBytes, BytesEnqueued, Chunks uint64
		PagesTotal, PagesFree, Rounds    uint32
		ServiceMsLast                    float64
	}
	RingProbe struct {
		PairSteps, PairCycles, PairTicks          uint64
		ExpressSteps, ExpressCycles, ExpressTicks uint64
	}
)

func NewConfig(int) (Config, error)               { return Config{}, ErrNotBuilt }
func (c Config) WithTable([256]uint32) Config     { return c }
func NewEngine(int, Config, int) (*Engine, error) { return nil, ErrNotBuilt }
func NewEngineOpt(int, Config, EngineOptions) (*Engine, error) { return nil, ErrNotBuilt }

const RingOff = ^uint32(0)
func ParseDynamicIndex([]byte) ([]ChunkInfo, int64, [32]byte, error) {
	return nil, 0, [32]byte{}, ErrNotBuilt
}
func LookupDynamicEntries([]ChunkInfo, uint64, uint64) ([]ReuseChunk, uint64, uint64, error) {
	return nil, 0, 0, ErrNotBuilt
}
func ShouldReuse([]ChunkInfo, uint64, uint64, *ReuseChunk, float64) (bool, error) { return false, ErrNotBuilt }

func (e *Engine) Close()                                                          {}
func (e *Engine) Submit([]byte, []uint64, []uint64, [][]uint64) (Ticket, error)   { return 0, ErrNotBuilt }
func (e *Engine) Done(Ticket) (bool, error)                                       { return false, ErrNotBuilt }
func (e *Engine) Collect(Ticket) ([]ChunkInfo, error)                             { return nil, ErrNotBuilt }
func (e *Engine) Dedup([]ChunkInfo) ([]bool, DedupStats, error)                   { return nil, DedupStats{}, ErrNotBuilt }
func (e *Engine) DedupDevice(uintptr, uint64) ([]bool, DedupStats, error)         { return nil, DedupStats{}, ErrNotBuilt }
func (e *Engine) EncodeDynamicIndex([]ChunkInfo, [16]byte, int64) ([]byte, error) { return nil, ErrNotBuilt }
func (e *Engine) NewStream(uint64) (*Stream, error)                               { return nil, ErrNotBuilt }
func (e *Engine) NewChunker() (*Chunker, error)                                   { return nil, ErrNotBuilt }
func (e *Engine) NewRing(RingOptions) (*Ring, error)                              { return nil, ErrNotBuilt }
func (e *Engine) NewComm([CommIDBytes]byte, int, int) (*Comm, error)              { return nil, ErrNotBuilt }
func (e *Engine) HashFiles([]byte, []uint64, []uint64) ([][32]byte, error)        { return nil, ErrNotBuilt }
func (e *Engine) HashFilesForced([]byte, []uint64, []uint64) ([][32]byte, error)  { return nil, ErrNotBuilt }
func (e *Engine) XXH3Files([]byte, []uint64, []uint64) ([]uint64, error)          { return nil, ErrNotBuilt }
func (e *Engine) Trim() (uint64, error)                                           { return 0, ErrNotBuilt }

func (s *Stream) Write([]byte) (int, error)                                  { return 0, ErrNotBuilt }
func (s *Stream) ReadFrom(io.Reader) (int64, error)                          { return 0, ErrNotBuilt }
func (s *Stream) WriteEntryReader(io.Reader, uint64) (uint64, uint64, error) { return 0, 0, ErrNotBuilt }
func (s *Stream) BeginFile() error                                           { return ErrNotBuilt }
func (s *Stream) EndFile() (uint64, error)                                   { return 0, ErrNotBuilt }
func (s *Stream) PollFiles(int) ([]FileHash, error)                          { return nil, ErrNotBuilt }
func (s *Stream) WriteMarker(bool) error                                     { return ErrNotBuilt }
func (s *Stream) Inject(uint64) error                                        { return ErrNotBuilt }
func (s *Stream) PayloadPosition() uint64                                    { return 0 }
func (s *Stream) SuggestBoundary() error                                     { return ErrNotBuilt }
func (s *Stream) Finish() error                                              { return ErrNotBuilt }
func (s *Stream) FinishBegin() error                                         { return ErrNotBuilt }
func (s *Stream) Done() (bool, error)                                        { return false, ErrNotBuilt }
func (s *Stream) Poll(int) ([]ChunkInfo, error)                              { return nil, ErrNotBuilt }
func (s *Stream) Close()                                                     {}

func (c *Chunker) Scan([]byte) (int, error) { return 0, ErrNotBuilt }
func (c *Chunker) Reset() error             { return ErrNotBuilt }
func (c *Chunker) Close()                   {}

func NewCommID() ([CommIDBytes]byte, error)                                      { return [CommIDBytes]byte{}, ErrNotBuilt }
func (c *Comm) Dedup([]ChunkInfo, uint64) ([]bool, DedupStats, error)            { return nil, DedupStats{}, ErrNotBuilt }
func (c *Comm) Close()                                                           {}
func (c *Comm) SplitStream(unsafe.Pointer, uint64, uint64) ([]ChunkInfo, error)  { return nil, ErrNotBuilt }
func SplitPlan(uint64, int, int, uint32) (uint64, uint64, uint64, uint64, error) { return 0, 0, 0, 0, ErrNotBuilt }
func CommLastError() string                                                      { return "" }
func (r *Ring) Open() (uint32, error)                                            { return 0, ErrNotBuilt }
func (r *Ring) Reserve(uint32) (uintptr, uint64, error)                          { return 0, 0, ErrNotBuilt }
func (r *Ring) Commit(uint32, uint64, bool) error                                { return ErrNotBuilt }
func (r *Ring) FillSynthetic(uint32, uint64, uint32, uint64, bool) (uint64, error) { return 0, ErrNotBuilt }
func (r *Ring) Pump() error                                                      { return ErrNotBuilt }
func (r *Ring) Poll(uint32, int) ([]ChunkInfo, bool, error)                      { return nil, false, ErrNotBuilt }
func (r *Ring) PollAny(int, int) ([]ChunkInfo, []uint32, error)                  { return nil, nil, ErrNotBuilt }
func (r *Ring) CloseStream(uint32) error                                         { return ErrNotBuilt }
func (r *Ring) Quiesce() error                                                   { return ErrNotBuilt }
func (r *Ring) Park() error                                                      { return ErrNotBuilt }
func (r *Ring) Suggest(uint32, uint64) error                                     { return ErrNotBuilt }
func (r *Ring) Stats() (RingStats, error)                                        { return RingStats{}, ErrNotBuilt }
func (r *Ring) Express() (uint32, uint64, error)                                 { return 0, 0, ErrNotBuilt }
func (r *Ring) Probe() (RingProbe, error)                                        { return RingProbe{}, ErrNotBuilt }
func (r *Ring) Close()                                                           {}
