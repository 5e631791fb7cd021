//go:build !bloomgpu

// gpu_engine_stub.go — what engine_gpu.patch's hooks resolve to when the engine is built WITHOUT -tags bloomgpu: a
// gpuEngine that is always nil, so every hook takes the stock path and the patched engine behaves exactly as the
// unpatched one.  Setting BloomSearchEngineConfig.GPUDevices in such a build is a configuration error.
package bloomsearch

import (
	"errors"
	"io"
	"log/slog"
	"time"
)

type gpuEngine struct{}

func openGPUEngine(config BloomSearchEngineConfig, _ *slog.Logger) (*gpuEngine, error) {
	if len(config.GPUDevices) > 0 {
		return nil, errors.New("GPUDevices is set but the engine was built without -tags bloomgpu")
	}
	return nil, nil
}

func (e *gpuEngine) close()                  {}
func (e *gpuEngine) forget(filePointer []byte) {}
func (e *gpuEngine) keepsRows() bool         { return false }

func (e *gpuEngine) buildFilters(entries *bloomEntrySets, fpr float64) BloomFilters {
	return entries.buildFilters(fpr)
}

type gpuFlushFilters struct{}

func (e *gpuEngine) flushFilters(map[string]*partitionBuffer, ValueTokenizerFunc, float64) *gpuFlushFilters {
	return nil
}
func (f *gpuFlushFilters) block(pb *partitionBuffer, fpr float64) BloomFilters { return pb.entries.buildFilters(fpr) }
func (f *gpuFlushFilters) blockCounts(pb *partitionBuffer) BloomEntryCounts    { return pb.entries.counts() }
func (f *gpuFlushFilters) file(e *gpuEngine, entries *bloomEntrySets, fpr float64) BloomFilters {
	return entries.buildFilters(fpr)
}
func (f *gpuFlushFilters) fileCounts(entries *bloomEntrySets) BloomEntryCounts { return entries.counts() }

func (e *gpuEngine) evaluateBlockFilters(*Results, io.ReadSeeker, fileFilterJob, []DataBlockMetadata, int64, int64, *BloomQuery, time.Duration,
	[]blockScanCandidate) ([]blockScanCandidate, bool, bool) {
	return nil, false, true
}

type gpuRowVerdicts struct{}

func (e *gpuEngine) matchBlock([]byte, *compiledRowMatcher) *gpuRowVerdicts { return nil }
func (v *gpuRowVerdicts) matches(_ int64, m *compiledRowMatcher, rowBytes []byte, scratch *rowMatchScratch) bool {
	return m.matchRowBytes(rowBytes, scratch)
}
