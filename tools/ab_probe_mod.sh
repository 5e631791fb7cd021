#!/bin/bash
# A/B on ONE box of two builds of the library (box-to-box spread is larger than the effect).  Build the two libraries first:
#   BSG_EXTRA_CXXFLAGS=-DBSG_LAB_PROBE_FP64 python -m bloomsearch_amd.build --force && cp bloomsearch_amd/csrc/libbloomgpu.so tools/lab/libbloomgpu_b.so
#   python -m bloomsearch_amd.build --force && cp bloomsearch_amd/csrc/libbloomgpu.so tools/lab/libbloomgpu_a.so
# then: gpurun -- bash tools/ab_probe_mod.sh
L=bloomsearch_amd/csrc/libbloomgpu.so
run() {
  python bench.py --steps 20 --warmup 5 --cpu-budget 0 --no-decode --no-q1 --no-concurrent --no-big-filters --no-single --ingest-blocks 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 c2', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], 'c4', d['c4']['kernel_ms'], d['c4']['frac'], d['c4']['ms_per_step'])"
  python bench.py --workload needle --steps 20 --warmup 5 --cpu-budget 0 --no-decode --no-q1 --no-concurrent --no-big-filters --no-single --ingest-blocks 0 --c4-files 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 needle', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
}
for i in 1 2; do
cp tools/lab/libbloomgpu_a.so $L; run a
cp tools/lab/libbloomgpu_b.so $L; run b
done
cp tools/lab/libbloomgpu_a.so $L
