//go:build !(cgo && gpuchunk)

// Without cgo or without the gpuchunk tag (the project's default CGO_ENABLED=0 builds,
// .goreleaser.yaml:30) this package only reports that the GPU engine is not compiled in, so
// the fork of github.com/pbs-plus/pxar keeps using its pure-Go chunker. Nothing here chunks.
package pbsgpu

import "errors"

// ErrNotBuilt is returned by every constructor in non-GPU builds.
var ErrNotBuilt = errors.New("pbsgpu: built without cgo/gpuchunk; GPU engine unavailable")

type (
	Config  struct{ AvgSize, MinSize, MaxSize, WindowSize int }
	Engine  struct{}
	Stream  struct{}
	Chunker struct{}
	Ticket  uint64
)

func NewConfig(int) (Config, error)               { return Config{}, ErrNotBuilt }
func NewEngine(int, Config, int) (*Engine, error) { return nil, ErrNotBuilt }
func (e *Engine) NewStream(uint64) (*Stream, error) { return nil, ErrNotBuilt }
func (e *Engine) NewChunker() (*Chunker, error)     { return nil, ErrNotBuilt }
func (e *Engine) Close()                            {}
