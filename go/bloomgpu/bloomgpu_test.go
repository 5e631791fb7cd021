package bloomgpu

import "testing"

// The public MurmurHash3_x64_128 vector for "hello" (seed 0) through bsg_hash_entries: h0, h1 of bloom/v3's
// baseHashes are exactly that digest.  Needs a gfx950 device; skipped otherwise.
func TestHashEntriesPublicVector(t *testing.T) {
	g, err := Open([]int32{0})
	if err != nil {
		t.Skip(err)
	}
	defer g.Close()
	bytes, off := PackEntries([]string{"hello", ""})
	h, err := g.HashEntries(bytes, off)
	if err != nil {
		t.Fatal(err)
	}
	if h[0][0] != 0xcbd8a7b341bd9b02 || h[0][1] != 0x5b1e906a48ae1d19 {
		t.Fatalf("murmur3(hello) = %016x %016x", h[0][0], h[0][1])
	}
	if h[1][0] != 0 || h[1][1] != 0 {
		t.Fatalf("murmur3(\"\") = %016x %016x", h[1][0], h[1][1])
	}
}

func TestEstimateParameters(t *testing.T) {
	for _, c := range []struct {
		n    uint64
		p    float64
		m, k uint64
	}{{100, 0.01, 959, 7}, {1, 0.001, 15, 11}, {20000, 0.001, 287552, 10}} {
		m, k, err := EstimateParameters(c.n, c.p)
		if err != nil || m != c.m || k != c.k {
			t.Fatalf("EstimateParameters(%d, %g) = %d, %d, %v", c.n, c.p, m, k, err)
		}
	}
}

// A failing call's message is read back from the scope it was made on, from another goroutine.
func TestScopesOwnTheirErrors(t *testing.T) {
	g, err := Open([]int32{0})
	if err != nil {
		t.Skip(err)
	}
	defer g.Close()
	a, _ := g.Scope()
	b, _ := g.Scope()
	defer a.Close()
	defer b.Close()
	errA := a.ArenaFree(Arena{ID: 0xDEAD})
	errB := b.BatchFree(Batch{ID: 0xBEEF})
	done := make(chan [2]string)
	go func() { done <- [2]string{errA.Error(), errB.Error()} }()
	got := <-done
	if got[0] == got[1] || errA == nil || errB == nil {
		t.Fatalf("scopes share an error slot: %q / %q", got[0], got[1])
	}
}
