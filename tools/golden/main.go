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
	"fmt"
	"os"
	"reflect"

	"github.com/pbs-plus/pxar/buzhash"
)

// configFields dumps whatever buzhash.Config really contains (SURVEY.md Appendix E.1/E.2: the field set — min, max,
// window, mask, break minimum, table — is not visible from pbs-plus). Reflection keeps this compiling whatever the names.
func configFields(cfg interface{}) map[string]interface{} {
	out := map[string]interface{}{}
	v := reflect.ValueOf(cfg)
	if v.Kind() == reflect.Ptr {
		v = v.Elem()
	}
	if v.Kind() != reflect.Struct {
		out["value"] = fmt.Sprint(cfg)
		return out
	}
	for i := 0; i < v.NumField(); i++ {
		f := v.Type().Field(i)
		if f.PkgPath != "" { // unexported: print what fmt can see
			out[f.Name] = fmt.Sprintf("%v", v.Field(i))
			continue
		}
		out[f.Name] = v.Field(i).Interface()
	}
	return out
}

// suggestedCase: the payload chunker's suggested boundaries (Appendix E.3). `Feed` = bytes handed to each Scan call
// (0 = everything that is left): upstream's result depends on it, the engine emulates it with
// pbsgpu_engine_set_suggested_feed. Adjust newPayloadChunker / Scan to the module's real names; if the module has no
// payload chunker, delete the suggested cases — that itself answers E.3.
type suggestedCase struct {
	Name      string   `json:"name"`
	Avg       int      `json:"avg"`
	Seed      uint64   `json:"seed"`
	Length    uint64   `json:"length"`
	Suggested []uint64 `json:"suggested"`
	Feed      int      `json:"feed"`
	Ends      []uint64 `json:"ends"`
}

type segSpec struct {
	Seed   uint64 `json:"seed"`
	Kind   uint32 `json:"kind"`
	Length uint64 `json:"length"`
	// Kind 100 (round 6): a crafted period — the pattern's bytes repeated. Candidate-DENSE data: with the casync table a
	// 64-byte pattern whose window hash is 0xFFFFFFFF makes every position a candidate, one that passes the break test at one
	// phase gives a candidate per period (tests/dense_inputs.py). The serial chunker cuts such data at the minimum size; the
	// engine must too (it answered PBSGPU_E_DENSITY on its streaming paths until ABI v4). The patterns are read from the
	// committed fixture so that both sides chunk the same bytes.
	Pattern string `json:"pattern,omitempty"`
}

// craftedCases: the "period*" cases of tests/golden/chunks_v1.json (names, averages, segment specs incl. the patterns)
func craftedCases(path string) []goldenCase {
	raw, err := os.ReadFile(path)
	if err != nil {
		fmt.Fprintln(os.Stderr, "no crafted cases:", err)
		return nil
	}
	var fx struct {
		Cases []struct {
			Name     string          `json:"name"`
			Avg      int             `json:"avg"`
			Segments json.RawMessage `json:"segments"`
		} `json:"cases"`
	}
	if err := json.Unmarshal(raw, &fx); err != nil {
		panic(err)
	}
	var out []goldenCase
	for _, c := range fx.Cases {
		if len(c.Name) < 6 || c.Name[:6] != "period" {
			continue
		}
		var segs []segSpec
		if err := json.Unmarshal(c.Segments, &segs); err != nil {
			panic(err)
		}
		out = append(out, goldenCase{Name: c.Name, Avg: c.Avg, Segments: segs})
	}
	return out
}

func segmentBytes(s segSpec) []byte {
	if s.Pattern == "" {
		return fill(s.Length, s.Seed, s.Kind)
	}
	pat, err := hex.DecodeString(s.Pattern)
	if err != nil || len(pat) == 0 {
		panic("bad pattern")
	}
	out := make([]byte, s.Length)
	for i := range out {
		out[i] = pat[i%len(pat)]
	}
	return out
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
		{Name: "rand_avg4k", Avg: 4096, Segments: []segSpec{{Seed: 11, Kind: 0, Length: 1 << 20}}},
		{Name: "rand_avg64k", Avg: 65536, Segments: []segSpec{{Seed: 31, Kind: 0, Length: 8 << 20}}},
		{Name: "zero_extents_avg64k", Avg: 65536, Segments: []segSpec{{Seed: 41, Kind: 3, Length: 6 << 20}, {Seed: 42, Kind: 1, Length: 1 << 20}}},
		{Name: "rand_avg4m", Avg: 4 << 20, Segments: []segSpec{{Seed: 51, Kind: 0, Length: 48 << 20}}},
	}
	cases = append(cases, craftedCases("../../tests/golden/chunks_v1.json")...) // (run from tools/golden: `make golden-go`)
	for ci := range cases {
		c := &cases[ci]
		cfg, err := buzhash.NewConfig(c.Avg)
		if err != nil {
			panic(err)
		}
		for si, s := range c.Segments {
			data := segmentBytes(s)
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
	// the module's actual Config for the two production / test averages (closes Appendix E.1 and E.2)
	cfgDump := map[string]interface{}{}
	for _, avg := range []int{4096, 4 << 20} {
		cfg, err := buzhash.NewConfig(avg)
		if err != nil {
			panic(err)
		}
		cfgDump[fmt.Sprint(avg)] = configFields(cfg)
	}
	// suggested boundaries: the LE-u32 counter buffer of upstream's test_suggested_boundary (avg 64 KiB; expected sizes
	// recalled from upstream: 32768, 110609, 229376, 32768, 262144, 262144, 118767) and a random stream, per feed size
	var sugg []suggestedCase
	counter := make([]byte, 1<<20)
	for i := 0; i < len(counter)/4; i++ {
		counter[4*i], counter[4*i+1], counter[4*i+2], counter[4*i+3] = byte(i), byte(i>>8), byte(i>>16), byte(i>>24)
	}
	for _, feed := range []int{1, 4096, 65536, 0} {
		for _, in := range []struct {
			name string
			data []byte
			seed uint64
			sg   []uint64
		}{
			{"counter_avg64k", counter, 0, []uint64{32 * 1024, 32 * 1024, 372753, 405521}},
			{"rand_avg64k", fill(4<<20, 61, 0), 61, []uint64{100000, 150000, 700001, 1 << 20, 3000000}},
		} {
			cfg, _ := buzhash.NewConfig(65536)
			pc := buzhash.NewPayloadChunker(cfg, in.sg) // adjust: upstream feeds boundaries through a channel
			c := suggestedCase{Name: in.name, Avg: 65536, Seed: in.seed, Length: uint64(len(in.data)), Suggested: in.sg, Feed: feed}
			pos, base := 0, 0
			for pos < len(in.data) {
				take := len(in.data) - pos
				if feed > 0 && feed < take {
					take = feed
				}
				k := pc.Scan(in.data[pos:pos+take], uint64(base), uint64(pos-base+take)) // (data, chunk base, bytes of the chunk so far)
				if k == 0 {
					pos += take
					continue
				}
				pos += k
				base = pos
				c.Ends = append(c.Ends, uint64(pos))
			}
			if base < len(in.data) {
				c.Ends = append(c.Ends, uint64(len(in.data)))
			}
			sugg = append(sugg, c)
		}
	}
	enc := json.NewEncoder(os.Stdout)
	_ = enc.Encode(map[string]interface{}{"schema": "pbsgpu-golden-v2", "generator": "github.com/pbs-plus/pxar v0.34.0",
		"cases": cases, "config": cfgDump, "suggested": sugg})
}
