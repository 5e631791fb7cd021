#!/usr/bin/env bash
# The reference's Go CPU path timed for bench.py's cpu_baseline (kind "reference").
#   go/run_bench_reference.sh /path/to/bloomsearch rows.ndjson [rows_per_block]
# prints the `go test -bench` lines (probes/s, rows/s, cores) and the public-API JSON line of cmd/bench_reference.
set -euo pipefail
REF=${1:?reference checkout}; NDJSON=${2:?rows.ndjson}; PER=${3:-10000}
REPO=$(cd "$(dirname "$0")/.." && pwd)
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$REF"/. "$WORK"/ref
cp "$REPO"/go/overlay/bench_reference_test.go "$WORK"/ref/
(cd "$WORK"/ref && BLOOMSEARCH_NDJSON="$NDJSON" BLOOMSEARCH_ROWS_PER_BLOCK="$PER" \
    go test -tags benchref -run '^$' -bench Reference -benchtime 1x -cpu "$(nproc)" .)
cp -r "$REPO"/cmd/bench_reference "$WORK"/cmd
(cd "$WORK"/cmd && go mod edit -replace github.com/danthegoodman1/bloomsearch="$WORK"/ref && go mod tidy >/dev/null 2>&1 && \
    go run . -ndjson "$NDJSON" -rows-per-file "$PER")
