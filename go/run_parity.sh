#!/usr/bin/env bash
# Parity of libbloomgpu against the REAL bits-and-blooms/bloom/v3 and the reference's own indexRow / encodeFilterSection.
# Needs: a Go toolchain (>= the reference's go.mod), bloom/v3 v3.7.0 + gjson + klauspost/compress in the module cache (or
# network), a gfx950 GPU, the built libbloomgpu.so.  Works on a scratch COPY of the reference checkout: nothing is
# written to it.
#   go/run_parity.sh /path/to/bloomsearch [extra go test args]
set -euo pipefail
REF=${1:?usage: run_parity.sh /path/to/reference-checkout [go test args]}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
WORK=$(mktemp -d)
trap 'rm -rf "$WORK"' EXIT
cp -r "$REF"/. "$WORK"/
cp "$REPO"/go/overlay/*.go "$WORK"/
cd "$WORK"
go mod edit -require=bloomsearch_amd/go/bloomgpu@v0.0.0 -replace=bloomsearch_amd/go/bloomgpu="$REPO/go/bloomgpu"
export CGO_CFLAGS="-I$REPO/include"
export CGO_LDFLAGS="-L$REPO/bloomsearch_amd/csrc -lbloomgpu -Wl,-rpath,$REPO/bloomsearch_amd/csrc"
go vet -tags bloomgpu . "$REPO/go/bloomgpu" || true
(cd "$REPO/go/bloomgpu" && go test ./...)
go test -tags bloomgpu -run 'GPU' -count=1 "$@" .
