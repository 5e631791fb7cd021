#!/usr/bin/env bash
# ONE command for the first box that has Go: parity of libbloomgpu against the REAL bits-and-blooms/bloom/v3 and the
# reference's own indexRow / encodeFilterSection, then the reference's WHOLE test suite with the GPU seams live.
# Needs: a Go toolchain (>= the reference's go.mod), bloom/v3 v3.7.0 + gjson + klauspost/compress in the module cache (or
# network), a gfx950 GPU, the built libbloomgpu.so.  Works on a scratch COPY of the reference checkout: nothing is
# written to it.
#   go/run_parity.sh /path/to/bloomsearch [extra go test args]
# Steps: 1 static check of the Go side against the header and the reference (tools/check_go.py)
#        2 engine_gpu.patch + overlay onto the copy; `go build` and the stock suite WITHOUT the tag (the stubs: zero behaviour change)
#        3 `go vet` + the binding's own tests
#        4 the parity tests (-run GPU): bitsets Equal, (m, k), TestString, section bytes, indexRow counts
#        5 the reference's whole suite, -tags bloomgpu, BLOOMSEARCH_GPU_DEVICES=0: flush / merge / query through the device
#        6 the same with BLOOMSEARCH_GPU_INGEST=1: rows walked on the device as well
set -euo pipefail
REF=${1:?usage: run_parity.sh /path/to/reference-checkout [go test args]}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
python3 "$REPO/tools/check_go.py" --reference "$REF"
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$REF"/. "$WORK"/
cd "$WORK"
patch -p1 < "$REPO/go/overlay/engine_gpu.patch"
cp "$REPO"/go/overlay/*.go "$WORK"/
go mod edit -require=bloomsearch_amd/go/bloomgpu@v0.0.0 -replace=bloomsearch_amd/go/bloomgpu="$REPO/go/bloomgpu"
export CGO_CFLAGS="-I$REPO/include"
export CGO_LDFLAGS="-L$REPO/bloomsearch_amd/csrc -lbloomgpu -Wl,-rpath,$REPO/bloomsearch_amd/csrc"
go build ./...                                   # without the tag: gpu_engine_stub.go, the engine as it was
go test -count=1 .                               # ... and its suite still passes
go vet -tags bloomgpu . "$REPO/go/bloomgpu" || true
(cd "$REPO/go/bloomgpu" && go test ./...)
go test -tags bloomgpu -run 'GPU' -count=1 "$@" .
BLOOMSEARCH_GPU_DEVICES=${BLOOMSEARCH_GPU_DEVICES:-0} go test -tags bloomgpu -count=1 "$@" ./...
BLOOMSEARCH_GPU_DEVICES=${BLOOMSEARCH_GPU_DEVICES:-0} BLOOMSEARCH_GPU_INGEST=1 go test -tags bloomgpu -count=1 "$@" ./...
