// Package bloomgpu is the cgo binding of libbloomgpu.so (include/bloomgpu.h): bloomsearch's hierarchical
// bloom-filter construct + probe hot path on AMD MI355X (gfx950).
//
// It is a thin, allocation-conscious wrapper: every method is one call into the C-ABI.  The reference engine
// (github.com/danthegoodman1/bloomsearch, package bloomsearch) binds it at its internal seams through the files in
// ../overlay; this package itself depends on nothing but cgo, so `go vet` and `go test` run on it alone.
//
// cgo rules honoured by the library: every pointer argument is read or written only for the duration of the call and
// never retained (exception, stated where it applies: memory from PinnedAlloc handed to an asynchronous probe); no
// Go-pointer-to-Go-pointer arguments (descriptors, terms and conditions are flat structs); no callbacks into Go; calls
// may come from any goroutine on any OS thread (no thread-local "current device", no runtime.LockOSThread).
//
// Errors: a failing call records its message on the handle it was made on.  A Context shared by concurrent goroutines
// has one slot, so give every goroutine-confined caller its own Scope (cheap; pool them): the message read after a
// failure is then the caller's own, whichever OS thread either call ran on.
//
// Build: CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/bloomsearch_amd/csrc -lbloomgpu -Wl,-rpath,<repo>/bloomsearch_amd/csrc"
package bloomgpu

/*
#include <stdlib.h>
#include "bloomgpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"unsafe"
)

// Filter kinds (BloomField / BloomToken / BloomFieldToken, query.go:480-484).
const (
	KindField      uint32 = C.BSG_KIND_FIELD
	KindToken      uint32 = C.BSG_KIND_TOKEN
	KindFieldToken uint32 = C.BSG_KIND_FIELD_TOKEN
)

// Program opcodes: op = opcode<<28 | arg, postfix order (see bloomgpu.h).
const (
	OpTerm  uint32 = C.BSG_OP_TERM
	OpAnd   uint32 = C.BSG_OP_AND
	OpOr    uint32 = C.BSG_OP_OR
	OpTrue  uint32 = C.BSG_OP_TRUE
	OpFalse uint32 = C.BSG_OP_FALSE
)

// Op packs one program word.
func Op(opcode, arg uint32) uint32 { return opcode<<28 | arg&0x0FFFFFFF }

// Probe / ingest flags.
const (
	ProbeAsync        uint32 = C.BSG_PROBE_ASYNC
	ProbeTimed        uint32 = C.BSG_PROBE_TIMED
	ProbeNoFuse       uint32 = C.BSG_PROBE_NOFUSE
	ProbeRowsPacked   uint32 = C.BSG_PROBE_ROWS_PACKED
	IngestTrustedJSON uint32 = C.BSG_INGEST_TRUSTED_JSON
)

// FilterDesc mirrors bsg_filter_desc (24 bytes).  M == 0 marks an absent (nil) filter.
type FilterDesc struct {
	WordOff  uint64
	M        uint64
	K        uint32
	Reserved uint32
}

// Term mirrors bsg_term (40 bytes): the four bloom/v3 base hashes of the probed string and its filter kind.
type Term struct {
	H        [4]uint64
	Kind     uint32
	Reserved uint32
}

// MatchCond is one condition of the device row matcher: kind + the field and token strings (empty where the kind has
// none).  The library hashes and fingerprints them itself.
type MatchCond struct {
	Kind  uint32
	Field string
	Token string
}

// IngestStats mirrors bsg_ingest_stats.
type IngestStats struct {
	Rows, FallbackRows, TableGrows, Reserved uint32
	RowBytes, TableBytes                     uint64
	MsWalk, MsUnion, MsBuild, MsEncode       float32
}

// The Go mirrors must have the C layouts: checked once at start-up.
func init() {
	if unsafe.Sizeof(FilterDesc{}) != unsafe.Sizeof(C.bsg_filter_desc{}) ||
		unsafe.Sizeof(Term{}) != unsafe.Sizeof(C.bsg_term{}) ||
		unsafe.Sizeof(IngestStats{}) != unsafe.Sizeof(C.bsg_ingest_stats{}) {
		panic("bloomgpu: struct layout differs from bloomgpu.h")
	}
}

// Arena is a resident, decoded set of block filters (bsg_arena_load*).
type Arena struct {
	ID     uint64
	Blocks uint32
}

// Batch is a compiled, uploaded query batch (bsg_batch_create).
type Batch struct {
	ID      uint64
	Queries uint32
}

// Context wraps a bsg_ctx, or an error scope of one (Scope).
type Context struct{ c *C.bsg_ctx }

// DeviceCount is bsg_device_count.
func DeviceCount() int { return int(C.bsg_device_count()) }

// Open opens the library on the given devices (engine construction).  There is no CPU fallback: without a gfx950
// device it fails.
func Open(deviceIDs []int32) (*Context, error) {
	if len(deviceIDs) == 0 {
		return nil, errors.New("bloomgpu: no device ids")
	}
	var ctx *C.bsg_ctx
	var msg [512]C.char
	rc := C.bsg_open_err((*C.int32_t)(unsafe.Pointer(&deviceIDs[0])), C.int32_t(len(deviceIDs)), &ctx, &msg[0], C.uint64_t(len(msg)))
	if rc != C.BSG_OK {
		return nil, fmt.Errorf("bloomgpu: open failed (%d): %s", int(rc), C.GoString(&msg[0]))
	}
	return &Context{ctx}, nil
}

// Close releases the context (or the scope).  Close scopes before the context they alias.
func (g *Context) Close() {
	if g != nil && g.c != nil {
		C.bsg_close(g.c)
		g.c = nil
	}
}

// Scope returns an alias of the context with an error slot of its own (bsg_scope_open).
func (g *Context) Scope() (*Context, error) {
	var s *C.bsg_ctx
	if rc := C.bsg_scope_open(g.c, &s); rc != C.BSG_OK {
		return nil, g.err(rc)
	}
	return &Context{s}, nil
}

// Error is a failed library call.
type Error struct {
	Code    int
	Message string
}

func (e *Error) Error() string { return fmt.Sprintf("bloomgpu: %s (%d)", e.Message, e.Code) }

func (g *Context) err(rc C.int32_t) error {
	if rc == C.BSG_OK {
		return nil
	}
	var msg [512]C.char
	C.bsg_last_error_copy(g.c, &msg[0], C.uint64_t(len(msg)))
	return &Error{Code: int(rc), Message: C.GoString(&msg[0])}
}

// Sync waits for everything enqueued on the context's devices.
func (g *Context) Sync() error { return g.err(C.bsg_sync(g.c)) }

func u8p(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

func u32p(b []uint32) *C.uint32_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&b[0]))
}

func u64p(b []uint64) *C.uint64_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint64_t)(unsafe.Pointer(&b[0]))
}

func descp(d []FilterDesc) *C.bsg_filter_desc {
	if len(d) == 0 {
		return nil
	}
	return (*C.bsg_filter_desc)(unsafe.Pointer(&d[0]))
}

// EstimateParameters is bsg_estimate_parameters.  A Go host should keep calling bloom.EstimateParameters (sizing stays
// in Go so that libm differences can never change (m, k)); this exists for tests and non-Go hosts.
func EstimateParameters(n uint64, p float64) (m, k uint64, err error) {
	var cm, ck C.uint64_t
	if rc := C.bsg_estimate_parameters(C.uint64_t(n), C.double(p), &cm, &ck); rc != C.BSG_OK {
		return 0, 0, &Error{Code: int(rc), Message: C.GoString(C.bsg_last_error(nil))}
	}
	return uint64(cm), uint64(ck), nil
}

// PackEntries lays strings out as bsg_hash_entries / bsg_build take them.
func PackEntries(entries []string) (bytes []byte, offsets []uint32) {
	offsets = make([]uint32, 1, len(entries)+1)
	for _, e := range entries {
		bytes = append(bytes, e...)
		offsets = append(offsets, uint32(len(bytes)))
	}
	return bytes, offsets
}

// HashEntries returns bloom/v3 baseHashes of every entry (bytes[offsets[e]:offsets[e+1]]).
func (g *Context) HashEntries(bytes []byte, offsets []uint32) ([][4]uint64, error) {
	n := len(offsets) - 1
	if n <= 0 {
		return nil, nil
	}
	out := make([][4]uint64, n)
	rc := C.bsg_hash_entries(g.c, u8p(bytes), u32p(offsets), C.uint32_t(n), (*C.uint64_t)(unsafe.Pointer(&out[0])))
	return out, g.err(rc)
}

// Build is bsg_build: the AddString loops of buildSizedBloomFilter (ingest.go:139-145) for many filters at once.
// Filter f owns entries [filterEntryStart[f], filterEntryStart[f+1]); the returned words are the bitsets at
// desc[f].WordOff (native little-endian u64, exactly a bitset.BitSet's []uint64).
func (g *Context) Build(bytes []byte, offsets, filterEntryStart []uint32, desc []FilterDesc, nWords uint64) ([]uint64, error) {
	words := make([]uint64, nWords)
	rc := C.bsg_build(g.c, u8p(bytes), u32p(offsets), C.uint32_t(len(offsets)-1), u32p(filterEntryStart), descp(desc),
		C.uint32_t(len(desc)), u64p(words), C.uint64_t(nWords))
	return words, g.err(rc)
}

// SectionsSize is bsg_sections_size.
func SectionsSize(desc []FilterDesc) (uint64, error) {
	var total C.uint64_t
	if rc := C.bsg_sections_size(descp(desc), C.uint32_t(len(desc)/3), &total); rc != C.BSG_OK {
		return 0, &Error{Code: int(rc), Message: C.GoString(C.bsg_last_error(nil))}
	}
	return uint64(total), nil
}

// BuildSections is bsg_build_sections: Build followed by encodeFilterSection (file_format.go:343-384) on the device.
// region[secOff[b]:secOff[b+1]] is block b's section, byte for byte what the reference writes.
func (g *Context) BuildSections(bytes []byte, offsets, filterEntryStart []uint32, desc []FilterDesc, nWords uint64) (region []byte, secOff []uint64, err error) {
	size, err := SectionsSize(desc)
	if err != nil {
		return nil, nil, err
	}
	region = make([]byte, size)
	secOff = make([]uint64, len(desc)/3+1)
	rc := C.bsg_build_sections(g.c, u8p(bytes), u32p(offsets), C.uint32_t(len(offsets)-1), u32p(filterEntryStart), descp(desc),
		C.uint32_t(len(desc)), C.uint64_t(nWords), u8p(region), C.uint64_t(len(region)), u64p(secOff))
	return region, secOff, g.err(rc)
}

// ArenaLoad uploads decoded filters: desc[b*3+kind] addresses words (bsg_arena_load).
func (g *Context) ArenaLoad(words []uint64, desc []FilterDesc) (Arena, error) {
	var id C.uint64_t
	rc := C.bsg_arena_load(g.c, u64p(words), C.uint64_t(len(words)), descp(desc), C.uint32_t(len(desc)/3), &id)
	return Arena{uint64(id), uint32(len(desc) / 3)}, g.err(rc)
}

// ArenaLoadSections uploads a file's filter region as stored (bsg_arena_load_sections): CRC32C and the big-endian
// decode run on the device; status[b] != 0 is parseFilterSection's failure for block b, whose filters become nil
// (fail-open) without affecting the other blocks (query_exec.go:580-590).
func (g *Context) ArenaLoadSections(region []byte, secOff []uint64) (Arena, []int32, error) {
	n := len(secOff) - 1
	if n < 0 {
		n = 0
	}
	status := make([]int32, n)
	var sp *C.int32_t
	if n > 0 {
		sp = (*C.int32_t)(unsafe.Pointer(&status[0]))
	}
	var id C.uint64_t
	rc := C.bsg_arena_load_sections(g.c, u8p(region), C.uint64_t(len(region)), u64p(secOff), C.uint32_t(n), sp, &id)
	return Arena{uint64(id), uint32(n)}, status, g.err(rc)
}

// ArenaStream is a filter region being handed over chunk by chunk (bsg_arena_stream_*): the device-side counterpart of
// blockFilterCursor (file_format.go:511-662).
type ArenaStream struct {
	g      *Context
	id     C.uint64_t
	blocks int
}

// ArenaStreamBegin opens a stream for the candidate blocks whose sections lie at file offsets
// [secBegin[b], secEnd[b]) (equal = the block has no section).
func (g *Context) ArenaStreamBegin(secBegin, secEnd []uint64) (*ArenaStream, error) {
	if len(secBegin) != len(secEnd) {
		return nil, errors.New("bloomgpu: secBegin and secEnd differ in length")
	}
	var id C.uint64_t
	rc := C.bsg_arena_stream_begin(g.c, u64p(secBegin), u64p(secEnd), C.uint32_t(len(secBegin)), &id)
	if err := g.err(rc); err != nil {
		return nil, err
	}
	return &ArenaStream{g, id, len(secBegin)}, nil
}

// Append hands over file bytes [fileOffset, fileOffset+len(chunk)) — one read of the region cursor (<= 4 MiB in the
// reference), in any order; the bytes are copied out before the call returns.
func (s *ArenaStream) Append(fileOffset uint64, chunk []byte) error {
	return s.g.err(C.bsg_arena_stream_append(s.g.c, s.id, C.uint64_t(fileOffset), u8p(chunk), C.uint64_t(len(chunk))))
}

// Finish returns the resident arena and parseFilterSection's status per block (0 ok, -2 ErrInvalidHash, -7 never read ...).
func (s *ArenaStream) Finish() (Arena, []int32, error) {
	status := make([]int32, s.blocks)
	var sp *C.int32_t
	if s.blocks > 0 {
		sp = (*C.int32_t)(unsafe.Pointer(&status[0]))
	}
	var id C.uint64_t
	rc := C.bsg_arena_stream_finish(s.g.c, s.id, sp, &id)
	return Arena{uint64(id), uint32(s.blocks)}, status, s.g.err(rc)
}

// Abort discards the stream.
func (s *ArenaStream) Abort() error { return s.g.err(C.bsg_arena_stream_abort(s.g.c, s.id)) }

// ArenaFree is bsg_arena_free.
func (g *Context) ArenaFree(a Arena) error { return g.err(C.bsg_arena_free(g.c, C.uint64_t(a.ID))) }

// ---- resident file arenas across queries (the library's cache: bsg_file_arena_*) ----

// FileLease is a query's hold on a file's arena: Rows[i] is candidate i's row (block index) in Arena.
type FileLease struct {
	ID    uint64
	Arena Arena
	Rows  []uint32
}

// SetArenaBudget bounds the device memory the resident arenas may hold (bsg_set_arena_budget).
func (g *Context) SetArenaBudget(bytes uint64) error {
	return g.err(C.bsg_set_arena_budget(g.c, C.uint64_t(bytes)))
}

// FileArenaAcquire leases the file's resident arena when it covers every candidate block (blockKeys strictly ascending:
// RowDataOffset); nil when it does not.
func (g *Context) FileArenaAcquire(key []byte, blockKeys []uint64) (*FileLease, error) {
	rows := make([]uint32, len(blockKeys))
	var rp *C.uint32_t
	if len(rows) > 0 {
		rp = (*C.uint32_t)(unsafe.Pointer(&rows[0]))
	}
	var lease, arena C.uint64_t
	var arenaBlocks C.uint32_t
	rc := C.bsg_file_arena_acquire(g.c, u8p(key), C.uint32_t(len(key)), u64p(blockKeys), C.uint32_t(len(blockKeys)), &lease, &arena, &arenaBlocks, rp)
	if err := g.err(rc); err != nil || lease == 0 {
		return nil, err
	}
	return &FileLease{ID: uint64(lease), Arena: Arena{ID: uint64(arena), Blocks: uint32(arenaBlocks)}, Rows: rows}, nil
}

// FileArenaHave returns what the file's resident arena holds: block keys and the sections' extents in the file.
func (g *Context) FileArenaHave(key []byte) (blockKeys, secBegin, secEnd []uint64, err error) {
	var n C.uint32_t
	if err = g.err(C.bsg_file_arena_have(g.c, u8p(key), C.uint32_t(len(key)), nil, nil, nil, 0, &n)); err != nil || n == 0 {
		return nil, nil, nil, err
	}
	for {
		c := int(n)
		blockKeys, secBegin, secEnd = make([]uint64, c), make([]uint64, c), make([]uint64, c)
		rc := C.bsg_file_arena_have(g.c, u8p(key), C.uint32_t(len(key)), u64p(blockKeys), u64p(secBegin), u64p(secEnd), C.uint32_t(c), &n)
		if rc == 0 && int(n) <= c {
			return blockKeys[:n], secBegin[:n], secEnd[:n], nil
		}
		if int(n) <= c { // a failure, not a wider arena published in between
			return nil, nil, nil, g.err(rc)
		}
	}
}

// FileArenaPublish hands a freshly decoded arena of exactly these blocks to the cache, which owns it from here on; the lease
// keeps it alive for this query whether or not it became the file's resident arena.
func (g *Context) FileArenaPublish(key []byte, a Arena, blockKeys, secBegin, secEnd []uint64, status []int32) (*FileLease, bool, error) {
	var sp *C.int32_t
	if len(status) > 0 {
		sp = (*C.int32_t)(unsafe.Pointer(&status[0]))
	}
	var lease C.uint64_t
	var resident C.int32_t
	rc := C.bsg_file_arena_publish(g.c, u8p(key), C.uint32_t(len(key)), C.uint64_t(a.ID), u64p(blockKeys), u64p(secBegin), u64p(secEnd), sp,
		C.uint32_t(len(blockKeys)), &lease, &resident)
	if err := g.err(rc); err != nil {
		return nil, false, err
	}
	rows := make([]uint32, len(blockKeys))
	for i := range rows {
		rows[i] = uint32(i)
	}
	return &FileLease{ID: uint64(lease), Arena: a, Rows: rows}, resident != 0, nil
}

// FileArenaRelease ends a lease (bsg_file_arena_release).
func (g *Context) FileArenaRelease(l *FileLease) error {
	return g.err(C.bsg_file_arena_release(g.c, C.uint64_t(l.ID)))
}

// FileArenaForget drops a tombstoned file's arena (bsg_file_arena_forget); its last user frees it.
func (g *Context) FileArenaForget(key []byte) error {
	return g.err(C.bsg_file_arena_forget(g.c, u8p(key), C.uint32_t(len(key))))
}

// BatchCreate compiles and uploads a query batch: distinct terms + per query the postfix program
// progOps[progOff[q]:progOff[q+1]] (bsg_batch_create).
func (g *Context) BatchCreate(terms []Term, progOps, progOff []uint32) (Batch, error) {
	var tp *C.bsg_term
	if len(terms) > 0 {
		tp = (*C.bsg_term)(unsafe.Pointer(&terms[0]))
	}
	var id C.uint64_t
	nq := len(progOff) - 1
	if nq < 0 {
		nq = 0
	}
	rc := C.bsg_batch_create(g.c, tp, C.uint32_t(len(terms)), u32p(progOps), u32p(progOff), C.uint32_t(nq), &id)
	return Batch{uint64(id), uint32(nq)}, g.err(rc)
}

// BatchFree is bsg_batch_free.
func (g *Context) BatchFree(b Batch) error { return g.err(C.bsg_batch_free(g.c, C.uint64_t(b.ID))) }

// ProbeMany evaluates every query of the batch against every block of every arena (bsg_probe_many) and returns, per
// arena, survivors[q*ceil(blocks/64) + b>>6] bit b&63 == evaluateBloomFilters(block b's filters, query q).
func (g *Context) ProbeMany(arenas []Arena, b Batch) ([][]uint64, error) {
	if len(arenas) == 0 || b.Queries == 0 {
		return nil, nil
	}
	ids := make([]uint64, len(arenas))
	total := 0
	for i, a := range arenas {
		ids[i] = a.ID
		total += int(b.Queries) * int((a.Blocks+63)/64)
	}
	flat := make([]uint64, total+1)
	rc := C.bsg_probe_many(g.c, u64p(ids), C.uint32_t(len(ids)), C.uint64_t(b.ID), 0, u64p(flat))
	if err := g.err(rc); err != nil {
		return nil, err
	}
	out := make([][]uint64, len(arenas))
	o := 0
	for i, a := range arenas {
		n := int(b.Queries) * int((a.Blocks+63)/64)
		out[i] = flat[o : o+n : o+n]
		o += n
	}
	return out, nil
}

// Probe is the one-shot convenience bsg_probe for a single arena.
func (g *Context) Probe(a Arena, terms []Term, progOps, progOff []uint32) ([]uint64, error) {
	b, err := g.BatchCreate(terms, progOps, progOff)
	if err != nil {
		return nil, err
	}
	defer g.BatchFree(b)
	out, err := g.ProbeMany([]Arena{a}, b)
	if err != nil || len(out) == 0 {
		return nil, err
	}
	return out[0], nil
}

// Query is one interactive Query() in ONE call (bsg_query): the probed STRINGS go in — keys[t] with kinds[t], i.e. Field, Token or
// Field + "::" + Token — and, per arena, survivors[q*ceil(blocks/64) + b>>6] bit b&63 comes out.  The strings are hashed on the
// host inside the call; for a handful of terms the hashes and programs ride in the kernel arguments of one dispatch per device: no
// batch object, no upload.  Larger batches take the batch path inside the same call.
func (g *Context) Query(arenas []Arena, keys []string, kinds, progOps, progOff []uint32) ([][]uint64, error) {
	nq := len(progOff) - 1
	if len(arenas) == 0 || nq <= 0 {
		return nil, nil
	}
	if len(kinds) != len(keys) {
		return nil, errors.New("bloomgpu: one kind per key")
	}
	bytes, offsets := PackEntries(keys)
	ids := make([]uint64, len(arenas))
	total := 0
	for i, a := range arenas {
		ids[i] = a.ID
		total += nq * int((a.Blocks+63)/64)
	}
	flat := make([]uint64, total+1)
	rc := C.bsg_query(g.c, u64p(ids), C.uint32_t(len(ids)), u8p(bytes), u32p(offsets), u32p(kinds), C.uint32_t(len(keys)),
		u32p(progOps), u32p(progOff), C.uint32_t(nq), u64p(flat))
	if err := g.err(rc); err != nil {
		return nil, err
	}
	out := make([][]uint64, len(arenas))
	o := 0
	for i, a := range arenas {
		n := nq * int((a.Blocks+63)/64)
		out[i] = flat[o : o+n : o+n]
		o += n
	}
	return out, nil
}

// SurvivorList is bsg_survivor_list: the surviving block indices of one query's survivor row, ascending — the order
// evaluateBlockFilters appends blockScanCandidate{index} in (query_exec.go:603) for an arena loaded in that block order.
func SurvivorList(row []uint64, blocks uint32) ([]uint32, error) {
	if blocks == 0 {
		return nil, nil
	}
	out := make([]uint32, blocks)
	var n C.uint32_t
	if rc := C.bsg_survivor_list(u64p(row), C.uint32_t(blocks), u32p(out), C.uint32_t(blocks), &n); rc != C.BSG_OK {
		return nil, &Error{Code: int(rc), Message: C.GoString(C.bsg_last_error(nil))}
	}
	return out[:n], nil
}

// SurvivorRows is the result of ProbeManyRows for one arena: a header per query (tag << 30 | surviving-block count) and the
// row slots in bsg_probe_many's dense layout; what a slot holds depends on the row's tag (see bsg_probe_many_rows).  Both
// slices are views of page-locked C memory owned by the RowsBuffer they came from.
type SurvivorRows struct {
	Blocks uint32
	Hdr    []uint32
	Rows   []uint64
}

// Blocks of query q, ascending (blockScanCandidate order, query_exec.go:321,603), whatever the row's tag.
func (s SurvivorRows) List(q int) ([]uint32, error) {
	g := int((s.Blocks + 63) / 64)
	hdr := s.Hdr[q]
	n := hdr & 0x3FFFFFFF
	if n == 0 {
		return nil, nil
	}
	out := make([]uint32, n)
	var got C.uint32_t
	var row *C.uint64_t
	if g > 0 {
		row = (*C.uint64_t)(unsafe.Pointer(&s.Rows[q*g]))
	}
	if rc := C.bsg_survivor_row_list(C.uint32_t(hdr), row, C.uint32_t(s.Blocks), u32p(out), C.uint32_t(n), &got); rc != C.BSG_OK {
		return nil, &Error{Code: int(rc), Message: C.GoString(C.bsg_last_error(nil))}
	}
	return out[:got], nil
}

// RowsBuffer is page-locked C memory (bsg_pinned_alloc) the device writes survivor rows into; reuse it across calls.
type RowsBuffer struct {
	g    *Context
	rows unsafe.Pointer
	hdr  unsafe.Pointer
	nRow int // u64 capacity
	nHdr int // u32 capacity
}

// NewRowsBuffer allocates room for `rowWords` u64 of row slots and `headers` u32 of headers.
func (g *Context) NewRowsBuffer(rowWords, headers int) (*RowsBuffer, error) {
	b := &RowsBuffer{g: g, nRow: rowWords, nHdr: headers}
	if err := g.err(C.bsg_pinned_alloc(g.c, C.uint64_t(8*(rowWords+1)), &b.rows)); err != nil {
		return nil, err
	}
	if err := g.err(C.bsg_pinned_alloc(g.c, C.uint64_t(4*(headers+1)), &b.hdr)); err != nil {
		C.bsg_pinned_free(g.c, b.rows)
		return nil, err
	}
	return b, nil
}

// Free returns the buffer's memory.
func (b *RowsBuffer) Free() {
	C.bsg_pinned_free(b.g.c, b.rows)
	C.bsg_pinned_free(b.g.c, b.hdr)
	b.rows, b.hdr = nil, nil
}

// ProbeManyRows is bsg_probe_many_rows: like ProbeMany, but what crosses PCIe is a 4-byte header per (arena, query) plus
// block ids or words only for the rows that need them (single-device contexts).
func (g *Context) ProbeManyRows(arenas []Arena, b Batch, buf *RowsBuffer) ([]SurvivorRows, error) {
	if len(arenas) == 0 || b.Queries == 0 {
		return nil, nil
	}
	ids := make([]uint64, len(arenas))
	total := 0
	for i, a := range arenas {
		ids[i] = a.ID
		total += int(b.Queries) * int((a.Blocks+63)/64)
	}
	if total > buf.nRow || len(arenas)*int(b.Queries) > buf.nHdr {
		return nil, &Error{Code: int(C.BSG_E_INVALID), Message: "rows buffer too small"}
	}
	rc := C.bsg_probe_many_rows(g.c, u64p(ids), C.uint32_t(len(ids)), C.uint64_t(b.ID), 0, (*C.uint64_t)(buf.rows), (*C.uint32_t)(buf.hdr))
	if err := g.err(rc); err != nil {
		return nil, err
	}
	rows := unsafe.Slice((*uint64)(buf.rows), buf.nRow+1)
	hdr := unsafe.Slice((*uint32)(buf.hdr), buf.nHdr+1)
	out := make([]SurvivorRows, len(arenas))
	o := 0
	for i, a := range arenas {
		n := int(b.Queries) * int((a.Blocks+63)/64)
		out[i] = SurvivorRows{Blocks: a.Blocks, Hdr: hdr[i*int(b.Queries) : (i+1)*int(b.Queries)], Rows: rows[o : o+n : o+n]}
		o += n
	}
	return out, nil
}

// QueryStats is bsg_query_stats: how the eligible Query calls were served since the last reset — concurrent calls share dispatches
// inside the library (a hot arena streamed once for all its callers, the rest one job-list dispatch).
type QueryStats struct {
	Calls, SoloCalls, Cycles, CycleCalls, Dispatches, HotArenas, MaxCallsPerCycle uint64
	PrepareNs, EnqueueNs, WaitNs, DealNs, WakeNs, ScatterNs, FreeNs, RetireNs       uint64
}

// QueryStats reads (and optionally resets) the combiner's counters.
func (g *Context) QueryStats(reset bool) (QueryStats, error) {
	var qstats C.bsg_query_stats
	r := C.int32_t(0)
	if reset {
		r = 1
	}
	if err := g.err(C.bsg_query_stats_read(g.c, &qstats, r)); err != nil {
		return QueryStats{}, err
	}
	return QueryStats{Calls: uint64(qstats.calls), SoloCalls: uint64(qstats.solo_calls), Cycles: uint64(qstats.cycles), CycleCalls: uint64(qstats.cycle_calls),
		Dispatches: uint64(qstats.dispatches), HotArenas: uint64(qstats.hot_arenas), MaxCallsPerCycle: uint64(qstats.max_calls_per_cycle),
		PrepareNs: uint64(qstats.ns_prepare), EnqueueNs: uint64(qstats.ns_enqueue), WaitNs: uint64(qstats.ns_wait), DealNs: uint64(qstats.ns_deal),
		WakeNs: uint64(qstats.ns_wake), ScatterNs: uint64(qstats.ns_scatter), FreeNs: uint64(qstats.ns_free), RetireNs: uint64(qstats.ns_retire)}, nil
}

// RowsOnDevices is the result of ProbeManyRowsOnDevices: the rows every device of the context wrote for its shards (local block
// numbers, a slice of the buffer per device).  List merges the shards' rows of one (arena, query) into GLOBAL block order.
type RowsOnDevices struct {
	g      *Context
	ids    []uint64
	batch  Batch
	buf    *RowsBuffer
	blocks []uint32
	packed bool // written with BSG_PROBE_ROWS_PACKED: read with bsg_survivor_rows_list_packed
}

// ProbeManyRowsOnDevices is bsg_probe_many_rows on a context opened over several GPUs (the Go host's shape: one process, all 8
// GPUs): block b of every arena lives on device b % n, every device writes its shards' rows into its own slice of buf, and the
// host-side gather of surviving block ids (query_exec.go:603) is RowsOnDevices.List.  Works on a single-device context too.
func (g *Context) ProbeManyRowsOnDevices(arenas []Arena, b Batch, buf *RowsBuffer) (*RowsOnDevices, error) {
	if len(arenas) == 0 || b.Queries == 0 {
		return nil, nil
	}
	r := &RowsOnDevices{g: g, ids: make([]uint64, len(arenas)), batch: b, buf: buf, blocks: make([]uint32, len(arenas))}
	for i, a := range arenas {
		r.ids[i], r.blocks[i] = a.ID, a.Blocks
	}
	var rowWords, hdrWords C.uint64_t
	if err := g.err(C.bsg_survivor_rows_size(g.c, u64p(r.ids), C.uint32_t(len(r.ids)), C.uint64_t(b.ID), &rowWords, &hdrWords)); err != nil {
		return nil, err
	}
	if int(rowWords) > buf.nRow || int(hdrWords) > buf.nHdr {
		return nil, &Error{Code: int(C.BSG_E_INVALID), Message: "rows buffer too small"}
	}
	// The packed form first (a run's LIST / DENSE payloads leave the device as one contiguous stretch instead of one PCIe write per
	// row); an arena beyond 1 024 blocks per device is refused before anything is launched, and takes the dense layout.
	r.packed = true
	rc := C.bsg_probe_many_rows(g.c, u64p(r.ids), C.uint32_t(len(r.ids)), C.uint64_t(b.ID), C.BSG_PROBE_ROWS_PACKED, (*C.uint64_t)(buf.rows), (*C.uint32_t)(buf.hdr))
	if rc == C.BSG_E_UNSUPPORTED {
		r.packed = false
		rc = C.bsg_probe_many_rows(g.c, u64p(r.ids), C.uint32_t(len(r.ids)), C.uint64_t(b.ID), 0, (*C.uint64_t)(buf.rows), (*C.uint32_t)(buf.hdr))
	}
	if err := g.err(rc); err != nil {
		return nil, err
	}
	return r, nil
}

// List returns the surviving blocks of query q in arena i, ascending global block indices (blockScanCandidate order).
func (r *RowsOnDevices) List(i, q int) ([]uint32, error) {
	if r.blocks[i] == 0 {
		return nil, nil
	}
	out := make([]uint32, r.blocks[i])
	var n C.uint32_t
	var rc C.int32_t
	if r.packed {
		rc = C.bsg_survivor_rows_list_packed(r.g.c, u64p(r.ids), C.uint32_t(len(r.ids)), C.uint64_t(r.batch.ID), (*C.uint64_t)(r.buf.rows), (*C.uint32_t)(r.buf.hdr),
			C.uint32_t(i), C.uint32_t(q), u32p(out), C.uint32_t(len(out)), &n)
	} else {
		rc = C.bsg_survivor_rows_list(r.g.c, u64p(r.ids), C.uint32_t(len(r.ids)), C.uint64_t(r.batch.ID), (*C.uint64_t)(r.buf.rows), (*C.uint32_t)(r.buf.hdr),
			C.uint32_t(i), C.uint32_t(q), u32p(out), C.uint32_t(len(out)), &n)
	}
	if err := r.g.err(rc); err != nil {
		return nil, err
	}
	return out[:n], nil
}

// PeerAccess is bsg_peer_access: m[i][j] is true when device i of the context reaches device j's memory directly (xGMI peer
// access); copies between the other pairs are staged through host memory by the runtime.
func (g *Context) PeerAccess(nDevices int) ([][]bool, error) {
	flat := make([]byte, nDevices*nDevices)
	if err := g.err(C.bsg_peer_access(g.c, (*C.uint8_t)(unsafe.Pointer(&flat[0])), C.uint32_t(nDevices))); err != nil {
		return nil, err
	}
	m := make([][]bool, nDevices)
	for i := range m {
		m[i] = make([]bool, nDevices)
		for j := range m[i] {
			m[i][j] = flat[i*nDevices+j] != 0
		}
	}
	return m, nil
}

// DeviceCalls is bsg_device_calls: construct / match parts each device of the context has served so far.
func (g *Context) DeviceCalls(nDevices int) ([]uint64, error) {
	out := make([]uint64, nDevices)
	return out, g.err(C.bsg_device_calls(g.c, u64p(out), C.uint32_t(nDevices)))
}

// OrReduce is bsg_or_reduce (north-star extension: OR of fixed-geometry filters; the reference rebuilds instead).
func (g *Context) OrReduce(a Arena, kind uint32, nWords uint64) ([]uint64, error) {
	out := make([]uint64, nWords)
	return out, g.err(C.bsg_or_reduce(g.c, C.uint64_t(a.ID), C.uint32_t(kind), u64p(out), C.uint64_t(nWords)))
}

// CommUniqueID is bsg_comm_unique_id (rank 0 of a one-process-per-GPU job; hand the bytes to the other ranks).
func CommUniqueID() ([]byte, error) {
	id := make([]byte, C.BSG_COMM_ID_BYTES)
	if rc := C.bsg_comm_unique_id(u8p(id)); rc != C.BSG_OK {
		return nil, &Error{Code: int(rc), Message: C.GoString(C.bsg_last_error(nil))}
	}
	return id, nil
}

// CommInit joins the RCCL communicator (id from CommUniqueID), or — with a nil id — makes every device of a
// multi-device context a rank of a communicator of its own.
func (g *Context) CommInit(id []byte, rank, world int) error {
	return g.err(C.bsg_comm_init(g.c, u8p(id), C.int32_t(rank), C.int32_t(world)))
}

// CommInfo is bsg_comm_info: the number of ranks and this context's rank as the communicator library itself reports
// them (ncclCommCount / ncclCommUserRank); fromLibrary is false when the bound library lacks the two symbols.
func (g *Context) CommInfo() (world, rank int, fromLibrary bool, err error) {
	var w, r, lib C.int32_t
	if err = g.err(C.bsg_comm_info(g.c, &w, &r, &lib)); err != nil {
		return 0, 0, false, err
	}
	return int(w), int(r), lib != 0, nil
}

// CommDestroy is bsg_comm_destroy.
func (g *Context) CommDestroy() error { return g.err(C.bsg_comm_destroy(g.c)) }

// OrAllreduce is bsg_or_allreduce: OrReduce over every rank of the communicator (ncclAllGather over xGMI + local OR).
func (g *Context) OrAllreduce(a Arena, kind uint32, nWords uint64) ([]uint64, error) {
	out := make([]uint64, nWords)
	return out, g.err(C.bsg_or_allreduce(g.c, C.uint64_t(a.ID), C.uint32_t(kind), u64p(out), C.uint64_t(nWords)))
}

// PinnedAlloc returns n bytes of page-locked C memory as a Go slice (bsg_pinned_alloc): marshal rows, entries or
// sections straight into it and the copies to the device become plain DMA.  Free it with PinnedFree.
func (g *Context) PinnedAlloc(n int) ([]byte, error) {
	var p unsafe.Pointer
	if rc := C.bsg_pinned_alloc(g.c, C.uint64_t(n), &p); rc != C.BSG_OK {
		return nil, g.err(rc)
	}
	return unsafe.Slice((*byte)(p), n), nil
}

// PinnedFree releases memory from PinnedAlloc.
func (g *Context) PinnedFree(b []byte) error {
	if cap(b) == 0 {
		return nil
	}
	return g.err(C.bsg_pinned_free(g.c, unsafe.Pointer(unsafe.SliceData(b[:1]))))
}

// Ingest is a device ingest in progress (bsg_ingest_*): rows -> distinct bloom entries -> exact counts -> bitsets.
type Ingest struct {
	g        *Context
	id       C.uint64_t
	nSets    int
	nParents int
}

// IngestRows walks, tokenizes, hashes and deduplicates marshaled JSON rows on the device (bsg_ingest_rows): the work
// of bloomEntrySets.indexRow (ingest.go:55-102) for the DEFAULT tokenizer.  Set s owns rows
// [setFirstRow[s], setFirstRow[s+1]); parentOfSet[s] names its file-level union or is 0xFFFFFFFF.
func (g *Context) IngestRows(rows []byte, rowOff []uint64, setFirstRow, parentOfSet []uint32, nParents int, flags uint32) (*Ingest, error) {
	var id C.uint64_t
	nSets := len(setFirstRow) - 1
	rc := C.bsg_ingest_rows(g.c, u8p(rows), u64p(rowOff), C.uint32_t(len(rowOff)-1), u32p(setFirstRow), C.uint32_t(nSets),
		u32p(parentOfSet), C.uint32_t(nParents), nil, C.uint32_t(flags), &id)
	if err := g.err(rc); err != nil {
		return nil, err
	}
	return &Ingest{g, id, nSets, nParents}, nil
}

// FallbackRows lists the rows the host walker must finish (bsg_ingest_fallback_rows).
func (in *Ingest) FallbackRows() ([]uint32, error) {
	var n C.uint32_t
	if rc := C.bsg_ingest_fallback_rows(in.g.c, in.id, nil, 0, &n); rc != C.BSG_OK {
		return nil, in.g.err(rc)
	}
	if n == 0 {
		return nil, nil
	}
	rows := make([]uint32, n)
	rc := C.bsg_ingest_fallback_rows(in.g.c, in.id, u32p(rows), n, &n)
	return rows[:n], in.g.err(rc)
}

// AddEntries hands over the entries the host walker produced for the fallback rows (bsg_ingest_add_entries).
func (in *Ingest) AddEntries(bytes []byte, offsets, setOfEntry, kindOfEntry []uint32) error {
	if len(setOfEntry) == 0 {
		return nil
	}
	return in.g.err(C.bsg_ingest_add_entries(in.g.c, in.id, u8p(bytes), u32p(offsets), C.uint32_t(len(setOfEntry)),
		u32p(setOfEntry), u32p(kindOfEntry)))
}

// Finish unions the sets into their parents and returns the exact distinct counts [(nSets+nParents)*3] and a status per
// set (2: the set must be rebuilt on the host path) (bsg_ingest_finish).
func (in *Ingest) Finish() (counts []uint64, status []uint32, err error) {
	n := in.nSets + in.nParents
	counts = make([]uint64, 3*n)
	status = make([]uint32, n)
	err = in.g.err(C.bsg_ingest_finish(in.g.c, in.id, u64p(counts), u32p(status)))
	return counts, status, err
}

// Build writes every table's bitset (bsg_ingest_build); desc[(nSets+nParents)*3] carries the caller's (m, k).
func (in *Ingest) Build(desc []FilterDesc, nWords uint64) ([]uint64, error) {
	words := make([]uint64, nWords)
	return words, in.g.err(C.bsg_ingest_build(in.g.c, in.id, descp(desc), u64p(words), C.uint64_t(nWords)))
}

// BuildSections serialises every set's filters as one filter section on the device and, when keepResident is set,
// leaves them resident as probe arenas: sets ("block" i = set i) and parents (bsg_ingest_build_sections).
func (in *Ingest) BuildSections(desc []FilterDesc, keepResident bool) (region []byte, secOff []uint64, sets, parents Arena, err error) {
	size, err := SectionsSize(desc)
	if err != nil {
		return nil, nil, Arena{}, Arena{}, err
	}
	region = make([]byte, size)
	secOff = make([]uint64, in.nSets+in.nParents+1)
	var a, b C.uint64_t
	var ap, bp *C.uint64_t
	if keepResident {
		ap, bp = &a, &b
	}
	rc := C.bsg_ingest_build_sections(in.g.c, in.id, descp(desc), u8p(region), C.uint64_t(len(region)), u64p(secOff), ap, bp)
	return region, secOff, Arena{uint64(a), uint32(in.nSets)}, Arena{uint64(b), uint32(in.nParents)}, in.g.err(rc)
}

// Stats is bsg_ingest_stats_read.
func (in *Ingest) Stats() (IngestStats, error) {
	var st IngestStats
	rc := C.bsg_ingest_stats_read(in.g.c, in.id, (*C.bsg_ingest_stats)(unsafe.Pointer(&st)))
	return st, in.g.err(rc)
}

// Free is bsg_ingest_free.
func (in *Ingest) Free() error { return in.g.err(C.bsg_ingest_free(in.g.c, in.id)) }

// MatchRows is the final row test on the device (bsg_match_rows): bits[r>>6] bit r&63 <=> row r matches; rows listed
// in hostRows must be decided by matchRowBytes (outside the device walker's envelope, or a hash collision with a
// condition string that only a byte compare can settle).
func (g *Context) MatchRows(rows []byte, rowOff []uint64, conds []MatchCond, progOps []uint32) (bits []uint64, hostRows []uint32, err error) {
	n := len(rowOff) - 1
	if n <= 0 {
		return nil, nil, nil
	}
	bits = make([]uint64, (n+63)/64)
	hostRows = make([]uint32, n)
	var cbytes []byte
	coff := make([]uint32, 1, 2*len(conds)+1)
	kinds := make([]uint32, len(conds))
	for i, c := range conds {
		cbytes = append(cbytes, c.Field...)
		coff = append(coff, uint32(len(cbytes)))
		cbytes = append(cbytes, c.Token...)
		coff = append(coff, uint32(len(cbytes)))
		kinds[i] = c.Kind
	}
	var nfb C.uint32_t
	rc := C.bsg_match_rows(g.c, u8p(rows), u64p(rowOff), C.uint32_t(n), u8p(cbytes), u32p(coff), u32p(kinds), C.uint32_t(len(conds)),
		u32p(progOps), C.uint32_t(len(progOps)), u64p(bits), u32p(hostRows), C.uint32_t(n), &nfb)
	return bits, hostRows[:nfb], g.err(rc)
}
