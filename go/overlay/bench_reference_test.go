//go:build bloomgpu || benchref

package bloomsearch

// The reference's own CPU path timed in isolation, for the `cpu_baseline` of bench.py (kind "reference"):
//   BenchmarkReferenceProbeLoop — per (query, block): parseFilterSection (CRC32C + bloom.ReadFrom x3) then
//     evaluateBloomFilters, i.e. the body of evaluateBlockFilters' loop (query_exec.go:572-615) over NDJSON-built blocks
//   BenchmarkReferenceBuildLoop — indexRow per row + buildFilters per block (ingest.go:450, flush.go:204)
// Input: BLOOMSEARCH_NDJSON = rows as bloomsearch_amd/synth.py::rows_json writes them (one JSON object per line),
// BLOOMSEARCH_ROWS_PER_BLOCK (default 10000).  Run with
//   go test -tags benchref -run '^$' -bench Reference -benchtime 1x -cpu $(nproc)
// and feed the printed probes/s into bench.py --cpu-reference-json.

import (
	"bufio"
	"os"
	"runtime"
	"strconv"
	"sync"
	"testing"
)

func loadNDJSONBlocks(b *testing.B) [][][]byte {
	path := os.Getenv("BLOOMSEARCH_NDJSON")
	if path == "" {
		b.Skip("BLOOMSEARCH_NDJSON not set")
	}
	per := 10000
	if v := os.Getenv("BLOOMSEARCH_ROWS_PER_BLOCK"); v != "" {
		per, _ = strconv.Atoi(v)
	}
	f, err := os.Open(path)
	if err != nil {
		b.Fatal(err)
	}
	defer f.Close()
	sc := bufio.NewScanner(f)
	sc.Buffer(make([]byte, 1<<20), 1<<26)
	var blocks [][][]byte
	var cur [][]byte
	for sc.Scan() {
		cur = append(cur, append([]byte(nil), sc.Bytes()...))
		if len(cur) == per {
			blocks, cur = append(blocks, cur), nil
		}
	}
	if len(cur) > 0 {
		blocks = append(blocks, cur)
	}
	return blocks
}

func BenchmarkReferenceBuildLoop(b *testing.B) {
	blocks := loadNDJSONBlocks(b)
	rows := 0
	for _, blk := range blocks {
		rows += len(blk)
	}
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		var wg sync.WaitGroup
		jobs := make(chan [][]byte, len(blocks))
		for _, blk := range blocks {
			jobs <- blk
		}
		close(jobs)
		for w := 0; w < runtime.GOMAXPROCS(0); w++ {
			wg.Add(1)
			go func() {
				defer wg.Done()
				for blk := range jobs {
					set := newBloomEntrySets()
					for _, r := range blk {
						set.indexRow(r, BasicWhitespaceLowerTokenizer)
					}
					_ = set.buildFilters(0.001)
				}
			}()
		}
		wg.Wait()
	}
	b.ReportMetric(float64(rows*b.N)/b.Elapsed().Seconds(), "rows/s")
	b.ReportMetric(float64(runtime.GOMAXPROCS(0)), "cores")
}

func BenchmarkReferenceProbeLoop(b *testing.B) {
	blocks := loadNDJSONBlocks(b)
	sections := make([][]byte, len(blocks))
	for i, blk := range blocks {
		set := newBloomEntrySets()
		for _, r := range blk {
			set.indexRow(r, BasicWhitespaceLowerTokenizer)
		}
		f := set.buildFilters(0.001)
		sec, err := encodeFilterSection(&f)
		if err != nil {
			b.Fatal(err)
		}
		sections[i] = sec
	}
	// the C2 batch shape: 3-term And(FieldToken) queries over the low-cardinality fields, one in four values absent
	levels := []string{"debug", "info", "warn", "error", "absent-level-0"}
	services := []string{"auth", "payment", "search", "gateway", "billing", "absent-svc-0"}
	var queries []*BloomQuery
	for q := 0; q < 256; q++ {
		e := And(FieldToken("level", levels[q%len(levels)]), FieldToken("service", services[(q/5)%len(services)]),
			FieldToken("nested.region", "region-"+strconv.Itoa(q%11)))
		queries = append(queries, &BloomQuery{Expression: &e})
	}
	engine := &BloomSearchEngine{}
	b.ResetTimer()
	for i := 0; i < b.N; i++ {
		var wg sync.WaitGroup
		jobs := make(chan *BloomQuery, len(queries))
		for _, q := range queries {
			jobs <- q
		}
		close(jobs)
		for w := 0; w < runtime.GOMAXPROCS(0); w++ {
			wg.Add(1)
			go func() {
				defer wg.Done()
				for q := range jobs {
					for _, sec := range sections {
						f, err := parseFilterSection(sec)
						if err != nil {
							panic(err)
						}
						_ = engine.evaluateBloomFilters(f.FieldBloomFilter, f.TokenBloomFilter, f.FieldTokenBloomFilter, q)
					}
				}
			}()
		}
		wg.Wait()
	}
	probes := float64(len(queries)*len(sections)*3) * float64(b.N)
	b.ReportMetric(probes/b.Elapsed().Seconds(), "probes/s")
	b.ReportMetric(float64(runtime.GOMAXPROCS(0)), "cores")
}
