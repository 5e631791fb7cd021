module bloomsearch_amd/go/bloomgpu

go 1.21
