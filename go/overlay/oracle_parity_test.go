//go:build bloomgpu

package bloomsearch

// First thing to run on a box that has BOTH a Go toolchain (with bloom/v3 v3.7.0 in its module cache) and a gfx950 GPU:
// it turns the assumptions B1-B5 of SURVEY.md 8c — on which the repository's CPU oracle rests — into facts, by comparing
// libbloomgpu with the real github.com/bits-and-blooms/bloom/v3 bit for bit.  go/run_parity.sh runs it.

import (
	"bytes"
	"encoding/json"
	"fmt"
	"math/rand"
	"testing"

	"github.com/bits-and-blooms/bloom/v3"

	"bloomsearch_amd/go/bloomgpu"
)

func openOrSkip(t testing.TB) *bloomgpu.Context {
	g, err := bloomgpu.Open([]int32{0})
	if err != nil {
		t.Skip(err)
	}
	return g
}

func randomSets(rng *rand.Rand, n int) *bloomEntrySets {
	set := newBloomEntrySets()
	for i := 0; i < n; i++ {
		b := make([]byte, rng.Intn(40)) // every murmur tail length incl. 15 -> 16 (B2)
		rng.Read(b)
		set.tokens[string(b)] = struct{}{}
		set.fieldTokens[fmt.Sprintf("f%d::%x", i%7, b)] = struct{}{}
	}
	set.fields["user.name"] = struct{}{}
	return set
}

// B1 (EstimateParameters), B2 (sum256), B3 (location), B4 (Add) and B5's word layout at once: the device-built bitset
// must Equal the one bloom/v3 builds from the same entries.
func TestGPUBuildParityAgainstBloomV3(t *testing.T) {
	g := openOrSkip(t)
	defer g.Close()
	rng := rand.New(rand.NewSource(1))
	for _, n := range []int{0, 1, 2, 9, 100, 1000, 20000, 70000, 200000} {
		for _, fpr := range []float64{0.001, 0.01, 1e-6} {
			set := randomSets(rng, n)
			want := set.buildFilters(fpr) // real bloom/v3
			got, err := buildFiltersGPU(g, []*bloomEntrySets{set}, fpr)
			if err != nil {
				t.Fatal(err)
			}
			for name, pair := range map[string][2]*bloom.BloomFilter{
				"field": {want.FieldBloomFilter, got[0].FieldBloomFilter},
				"token": {want.TokenBloomFilter, got[0].TokenBloomFilter},
				"ft":    {want.FieldTokenBloomFilter, got[0].FieldTokenBloomFilter}} {
				if pair[0].Cap() != pair[1].Cap() || pair[0].K() != pair[1].K() {
					t.Fatalf("n=%d fpr=%g %s: geometry (%d,%d) vs (%d,%d)", n, fpr, name, pair[0].Cap(), pair[0].K(), pair[1].Cap(), pair[1].K())
				}
				if !pair[0].Equal(pair[1]) {
					t.Fatalf("n=%d fpr=%g %s: GPU bitset differs from bloom/v3", n, fpr, name)
				}
			}
			// B1 for non-Go hosts: bsg_estimate_parameters must agree with the library too
			m, k, err := bloomgpu.EstimateParameters(uint64(max(len(set.tokens), 1)), fpr)
			if err != nil || uint(m) != want.TokenBloomFilter.Cap() || uint(k) != want.TokenBloomFilter.K() {
				t.Fatalf("bsg_estimate_parameters(%d, %g) = (%d, %d), bloom/v3 (%d, %d)", len(set.tokens), fpr, m, k,
					want.TokenBloomFilter.Cap(), want.TokenBloomFilter.K())
			}
		}
	}
}

// TestString parity: the device probe of library-built filters answers as TestString does, members and non-members.
func TestGPUProbeParityAgainstBloomV3(t *testing.T) {
	g := openOrSkip(t)
	defer g.Close()
	rng := rand.New(rand.NewSource(2))
	var blocks []*BloomFilters
	for b := 0; b < 70; b++ {
		f := randomSets(rng, 50+rng.Intn(3000)).buildFilters(0.01)
		if b%9 == 0 {
			f.TokenBloomFilter = nil // nil filter: fail-open
		}
		blocks = append(blocks, &f)
	}
	arena, err := loadArena(g, blocks)
	if err != nil {
		t.Fatal(err)
	}
	defer g.ArenaFree(arena)
	for i := 0; i < 300; i++ {
		probe := fmt.Sprintf("probe%d", i)
		expr := Or(Token(probe), And(Field("user.name"), FieldToken("f3", probe)))
		survivors, _, err := probeBlocks(g, []bloomgpu.Arena{arena}, &BloomQuery{Expression: &expr})
		if err != nil {
			t.Fatal(err)
		}
		for b, f := range blocks {
			tok := f.TokenBloomFilter == nil || f.TokenBloomFilter.TestString(probe)
			want := tok || (f.FieldBloomFilter.TestString("user.name") && f.FieldTokenBloomFilter.TestString(makeFieldTokenKey("f3", probe)))
			if got := survivors[0][b>>6]>>(uint(b)&63)&1 == 1; got != want {
				t.Fatalf("probe %q block %d: GPU %v, bloom/v3 %v", probe, b, got, want)
			}
		}
	}
}

// B5 (wire format): the section bytes the device writes are encodeFilterSection's, and the device decodes the
// reference's bytes into filters that Equal the originals.
func TestGPUSectionBytesParity(t *testing.T) {
	g := openOrSkip(t)
	defer g.Close()
	rng := rand.New(rand.NewSource(3))
	sets := []*bloomEntrySets{randomSets(rng, 5000), randomSets(rng, 0), randomSets(rng, 70000)}
	var want [][]byte
	var region []byte
	secOff := []uint64{0}
	for _, s := range sets {
		f := s.buildFilters(0.001)
		sec, err := encodeFilterSection(&f)
		if err != nil {
			t.Fatal(err)
		}
		want = append(want, sec)
		region = append(region, sec...)
		secOff = append(secOff, uint64(len(region)))
	}
	// device encode
	var entries []byte
	offsets, fstart := []uint32{0}, []uint32{0}
	var desc []bloomgpu.FilterDesc
	var cursor uint64
	for _, s := range sets {
		for _, set := range []map[string]struct{}{s.fields, s.tokens, s.fieldTokens} {
			m, k := sizeFor(len(set), 0.001)
			desc = append(desc, bloomgpu.FilterDesc{WordOff: cursor, M: m, K: k})
			cursor += ((m+63)/64 + 15) / 16 * 16
			for e := range set {
				entries = append(entries, e...)
				offsets = append(offsets, uint32(len(entries)))
			}
			fstart = append(fstart, uint32(len(offsets)-1))
		}
	}
	got, gotOff, err := g.BuildSections(entries, offsets, fstart, desc, cursor)
	if err != nil {
		t.Fatal(err)
	}
	for b := range sets {
		if !bytes.Equal(got[gotOff[b]:gotOff[b+1]], want[b]) {
			t.Fatalf("section %d: device bytes differ from encodeFilterSection", b)
		}
	}
	// device decode of the reference's bytes, then probe parity on a member of every set
	arena, status, err := g.ArenaLoadSections(region, secOff)
	if err != nil {
		t.Fatal(err)
	}
	defer g.ArenaFree(arena)
	for b, st := range status {
		if st != 0 {
			t.Fatalf("section %d: device parse status %d on clean bytes", b, st)
		}
	}
	expr := Field("user.name")
	survivors, _, err := probeBlocks(g, []bloomgpu.Arena{arena}, &BloomQuery{Expression: &expr})
	if err != nil || survivors[0][0]&7 != 7 {
		t.Fatalf("decoded arena lost a member: %v %b", err, survivors[0][0])
	}
}

// indexRow parity: the device walker + dedup must produce the reference's exact distinct counts and filters.
func TestGPUIngestParityAgainstIndexRow(t *testing.T) {
	g := openOrSkip(t)
	defer g.Close()
	rng := rand.New(rand.NewSource(7))
	var rows [][][]byte
	for b := 0; b < 4; b++ {
		var buf [][]byte
		for i := 0; i < 500; i++ {
			row := map[string]any{
				"timestamp": int64(1700000000 + b*500 + i), "level": []string{"debug", "INFO", "Warn", "error"}[rng.Intn(4)],
				"message": fmt.Sprintf("Kelvin İstanbul <html>&amp; %d x", rng.Intn(50)), "user.id": rng.Intn(1000),
				"nested": map[string]any{"a.b": []any{rng.Intn(9), nil, true, 1.5e6}}, "tags": []any{"x", "y z"},
			}
			bts, err := json.Marshal(row)
			if err != nil {
				t.Fatal(err)
			}
			buf = append(buf, bts)
		}
		rows = append(rows, buf)
	}
	filters, counts, err := buildFiltersFromRowsGPU(g, rows, 0.001)
	if err != nil {
		t.Fatal(err)
	}
	file := newBloomEntrySets()
	for b, buf := range rows {
		set := newBloomEntrySets()
		for _, r := range buf {
			set.indexRow(r, BasicWhitespaceLowerTokenizer)
		}
		set.unionInto(file)
		if counts[b] != set.counts() {
			t.Fatalf("buffer %d: device counts %+v, indexRow %+v", b, counts[b], set.counts())
		}
		want := set.buildFilters(0.001)
		if !want.FieldBloomFilter.Equal(filters[b].FieldBloomFilter) || !want.TokenBloomFilter.Equal(filters[b].TokenBloomFilter) ||
			!want.FieldTokenBloomFilter.Equal(filters[b].FieldTokenBloomFilter) {
			t.Fatalf("buffer %d: device-built filters differ from buildFilters", b)
		}
	}
	if counts[len(rows)] != file.counts() {
		t.Fatalf("file level: device counts %+v, unionInto %+v", counts[len(rows)], file.counts())
	}
}
