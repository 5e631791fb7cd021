module bloomsearch_amd/cmd/bench_reference

go 1.26.0

require github.com/danthegoodman1/bloomsearch v0.0.0

// point this at a checkout of the reference (go/run_bench_reference.sh does):
//   go mod edit -replace github.com/danthegoodman1/bloomsearch=/path/to/bloomsearch && go mod tidy
