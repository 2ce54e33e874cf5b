// tools/golden: emits golden (segment, end, size, sha256) vectors from the REAL Go module
// github.com/pbs-plus/pxar v0.34.0 in the schema of tests/golden/chunks_v1.json, so the
// "parity unpinned" status of oracle/ can be closed by a maintainer who has Go + the module
// (neither exists in the build image: SURVEY.md §0.5).
//
//	go run ./tools/golden > tests/golden/chunks_go.json
//
// The synthetic inputs are the same splitmix64 streams as oracle_fill / pbsgpu_fill_device.
// NOTE (SURVEY.md Appendix E.1): the exported surface of the module's buzhash package beyond
// NewConfig/Config is not visible from pbs-plus; adjust `newChunker`/`scan` below to the real
// names (Proxmox's chunker exposes Scan(data) int with 0 = no boundary).
package main

import (
	"crypto/sha256"
	"encoding/hex"
	"encoding/json"
	"os"

	"github.com/pbs-plus/pxar/buzhash"
)

type segSpec struct {
	Seed   uint64 `json:"seed"`
	Kind   uint32 `json:"kind"`
	Length uint64 `json:"length"`
}

type goldenCase struct {
	Name     string          `json:"name"`
	Avg      int             `json:"avg"`
	Segments []segSpec       `json:"segments"`
	Records  [][]interface{} `json:"records"`
}

func splitmix64(seed, idx uint64) uint64 {
	z := seed + (idx+1)*0x9E3779B97F4A7C15
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
	z = (z ^ (z >> 27)) * 0x94D049BB133111EB
	return z ^ (z >> 31)
}

func fillWord(w, seed uint64, kind uint32) uint64 {
	switch kind {
	case 0:
		return splitmix64(seed, w)
	case 1:
		return 0
	case 2:
		return splitmix64(seed, w&511)
	default:
		r := splitmix64(seed^0xA5A5A5A55A5A5A5A, w>>13)
		if ((r>>32)*10)>>32 < 3 {
			return 0
		}
		return splitmix64(seed, w)
	}
}

func fill(n, seed uint64, kind uint32) []byte {
	out := make([]byte, n)
	for i := uint64(0); i < n; i++ {
		out[i] = byte(fillWord(i>>3, seed, kind) >> (8 * (i & 7)))
	}
	return out
}

func main() {
	cases := []goldenCase{
		{Name: "rand_avg4k", Avg: 4096, Segments: []segSpec{{11, 0, 1 << 20}}},
		{Name: "rand_avg64k", Avg: 65536, Segments: []segSpec{{31, 0, 8 << 20}}},
		{Name: "zero_extents_avg64k", Avg: 65536, Segments: []segSpec{{41, 3, 6 << 20}, {42, 1, 1 << 20}}},
		{Name: "rand_avg4m", Avg: 4 << 20, Segments: []segSpec{{51, 0, 48 << 20}}},
	}
	for ci := range cases {
		c := &cases[ci]
		cfg, err := buzhash.NewConfig(c.Avg)
		if err != nil {
			panic(err)
		}
		for si, s := range c.Segments {
			data := fill(s.Length, s.Seed, s.Kind)
			ch := buzhash.NewChunker(cfg) // adjust to the module's constructor
			start, pos := 0, 0
			emit := func(end int) {
				d := sha256.Sum256(data[start:end])
				c.Records = append(c.Records, []interface{}{si, end, end - start, hex.EncodeToString(d[:])})
				start = end
			}
			for pos < len(data) {
				k := ch.Scan(data[pos:]) // 0 = no boundary in the rest
				if k == 0 {
					break
				}
				pos += k
				emit(pos)
			}
			if start < len(data) {
				emit(len(data))
			}
		}
	}
	enc := json.NewEncoder(os.Stdout)
	_ = enc.Encode(map[string]interface{}{"schema": "pbsgpu-golden-v1", "generator": "github.com/pbs-plus/pxar v0.34.0", "cases": cases})
}
