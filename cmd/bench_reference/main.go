// bench_reference — BASELINE configs[0] on the reference's own Go path through its PUBLIC API only: NDJSON rows (as
// bloomsearch_amd/synth.py::rows_json writes them) -> IngestRows in batches -> Flush every --rows-per-file rows ->
// Query(FieldToken("level","error")) -> one JSON line with wall times, QueryStats and the core count.  The isolated
// parse+probe and index+build loops need unexported functions and live in go/overlay/bench_reference_test.go.
package main

import (
	"bufio"
	"bytes"
	"context"
	"encoding/json"
	"flag"
	"fmt"
	"os"
	"runtime"
	"time"

	"github.com/danthegoodman1/bloomsearch"
)

func main() {
	ndjson := flag.String("ndjson", "", "rows, one JSON object per line")
	dir := flag.String("dir", "", "FileSystemDataStore directory (default: a temp dir)")
	perFile := flag.Int("rows-per-file", 10000, "rows per flush (=> files of one block at the default MaxRowGroupRows)")
	batch := flag.Int("batch", 1000, "rows per IngestRows call")
	field := flag.String("field", "level", "FieldToken field")
	token := flag.String("token", "error", "FieldToken token")
	flag.Parse()
	if *ndjson == "" {
		fmt.Fprintln(os.Stderr, "usage: bench_reference -ndjson rows.ndjson")
		os.Exit(2)
	}
	root := *dir
	if root == "" {
		var err error
		if root, err = os.MkdirTemp("", "bloomsearch-bench"); err != nil {
			panic(err)
		}
		defer os.RemoveAll(root)
	}
	store := bloomsearch.NewFileSystemDataStore(root)
	cfg := bloomsearch.DefaultBloomSearchEngineConfig()
	cfg.MaxBufferedTime = time.Hour
	cfg.MaxBufferedRows = *perFile * 2 // flushes are explicit
	cfg.MaxBufferedBytes = 1 << 30
	cfg.MaxQueryConcurrency = runtime.NumCPU()
	engine, err := bloomsearch.NewBloomSearchEngine(cfg, store, store)
	if err != nil {
		panic(err)
	}
	engine.Start()
	ctx := context.Background()

	f, err := os.Open(*ndjson)
	if err != nil {
		panic(err)
	}
	defer f.Close()
	sc := bufio.NewScanner(f)
	sc.Buffer(make([]byte, 1<<20), 1<<26)
	var rows []map[string]any
	total, sinceFlush := 0, 0
	tIngest := time.Duration(0)
	send := func() {
		if len(rows) == 0 {
			return
		}
		done := make(chan error, 1)
		t0 := time.Now()
		if err := engine.IngestRows(ctx, rows, done); err != nil {
			panic(err)
		}
		if err := <-done; err != nil {
			panic(err)
		}
		sinceFlush += len(rows)
		if sinceFlush >= *perFile {
			if err := engine.Flush(ctx); err != nil {
				panic(err)
			}
			sinceFlush = 0
		}
		tIngest += time.Since(t0)
		total += len(rows)
		rows = rows[:0:0]
	}
	for sc.Scan() {
		var row map[string]any
		dec := json.NewDecoder(bytes.NewReader(sc.Bytes()))
		dec.UseNumber() // keep integer literals as written
		if err := dec.Decode(&row); err != nil {
			panic(err)
		}
		rows = append(rows, row)
		if len(rows) == *batch {
			send()
		}
	}
	send()
	t0 := time.Now()
	if err := engine.Flush(ctx); err != nil {
		panic(err)
	}
	tIngest += time.Since(t0)

	t0 = time.Now()
	results, err := engine.Query(ctx, bloomsearch.NewQuery().FieldToken(*field, *token).Build())
	if err != nil {
		panic(err)
	}
	matched := 0
	for results.Next() {
		matched++
	}
	if err := results.Err(); err != nil {
		panic(err)
	}
	stats := results.Stats()
	results.Close()
	tQuery := time.Since(t0)
	_ = engine.Stop(ctx)

	out := map[string]any{
		"rows": total, "ingest_flush_s": tIngest.Seconds(), "rows_per_s_ingest": float64(total) / tIngest.Seconds(),
		"query_s": tQuery.Seconds(), "rows_matched": matched, "blocks_processed": stats.BlocksProcessed,
		"blocks_skipped": stats.BlocksSkipped, "rows_scanned": stats.RowsScanned, "cores": runtime.NumCPU(),
		"go": runtime.Version(), "query": fmt.Sprintf("FieldToken(%q, %q)", *field, *token),
	}
	enc, _ := json.Marshal(out)
	fmt.Println(string(enc))
}
