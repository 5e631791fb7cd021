// ingest.hip.h — CDNA4 (gfx950) device code for the construct side of the path:
// marshaled-JSON rows -> distinct bloom entries (fields / tokens / field::token) per set,
// exact distinct counts, then bitsets straight from the distinct sets.
//
// Reference functions this replaces on the device:
//   pathWalker.walk / walkValue / emitKeyPrefixPaths ........ row_matcher.go:51-135 (twin: tokenizer.go:51-113)
//   leafTokenInput ........................................... tokenizer.go:120-133
//   BasicWhitespaceLowerTokenizer (forEachWord/appendFoldedWord) tokenizer.go:141-143, row_matcher.go:142-202
//   bloomEntrySets.indexRow / addFieldToken / unionInto / counts  ingest.go:24-123 (key: tokenizer.go:509-511)
//   buildFilters' AddString loop ............................. ingest.go:127-145
//
// Scope of the device walker ("walker-lite"): rows of valid UTF-8 without raw control bytes, whose JSON escapes are the
// simple ones (\" \\ \/ \b \f \n \r \t) or \uXXXX incl. well-formed surrogate pairs (json.Marshal's \u003c \u003e \u0026 included);
// nesting <= kMaxDepth, paths <= kPathCap bytes.  Words are split on Unicode white space and lower-cased with
// unicode.ToLower's one-rune mapping (a 512 KB direct table built from the host walker's own data); keys are copied as
// they are, like the host does.  A row that leaves that envelope — or is malformed — is appended to the fallback list
// and finished by the host walker (walker.hpp), which owns invalid UTF-8 (U+FFFD per byte), lone surrogate escapes and the
// lenient error semantics.
// The kernel walks every row twice: a validation pass (automaton only, no hashing) decides whether the row is the
// device's, and only rows that pass are walked again to emit — a row handed to the host has inserted nothing.
//
// A "set" is the distinct-entry state of one partition buffer (child) or of one file (parent, the
// union of its children): three open-addressing tables of 32-byte slots holding the four bloom/v3 base
// hashes of an entry.  Distinctness is decided on all 256 bits.  h0 == 0 marks an empty slot; an entry
// whose own h0..h3 contains a zero word cannot be represented and flags the table (status 2) so the
// caller rebuilds that set on the host path (probability 2^-62 per entry).
#pragma once
#include "kernels.hip.h"

namespace bsg {

constexpr int kIngestThreads = 256;
constexpr uint32_t kPathCap = 96;       // longest path the device walker keeps (bytes)
constexpr uint32_t kMaxDepth = 16;      // container nesting handled on the device
constexpr uint32_t kLaneLds = 116;      // 96 path + 16 stack + pad = 29 dwords (odd stride: lanes spread over LDS banks)
constexpr uint32_t kMaxProbe = 96;      // linear-probe bound before a table is declared full
constexpr uint32_t kSpinLimit = 1u << 20;   // re-reads of a claimed slot whose h1..h3 are still in flight (a claimer writes them right
                                        // after its CAS; only a wave that was context-switched out in between can make this run long)

constexpr uint32_t kTableOk = 0, kTableOverflow = 1, kTableExotic = 2;

struct IngestTable {
    uint64_t *slots;   // [mask + 1][4]
    uint64_t *fps;     // [mask + 1]: keyed fingerprint of the entry stored in the slot (0 while its claimer is still writing)
    uint32_t mask;     // capacity - 1 (a power of two for tables that are inserted into; a DENSE list of n entries has mask = n - 1)
    uint32_t shift;    // 32 - log2(capacity): an entry's home slot is the TOP log2(capacity) bits of table_key(h)
};

// An entry's position key: 32 bits of h1 (h0 is the claim word).  The home slot is its TOP bits, so that for any two tables —
// whatever their capacities — "the entries whose key starts with these b bits" is ONE contiguous run of slots (plus the linear-
// probe spill behind it): k_union_partitions unions a file's children partition by partition without moving an entry first.
__device__ __forceinline__ uint32_t table_key(const uint64_t h[4]) { return (uint32_t)(h[1] >> 20); }

// Exactness where the reference is exact.  Go's map compares BYTES; the tables compare the 256 bits of bloom/v3's sum256,
// and MurmurHash3_x64_128 has seed-independent internal-state collisions: for any entry of >= 24 bytes an attacker can
// write a second one with the same four base hashes (tests/test_collisions.py constructs such pairs).  The device would
// count them once where the reference counts twice (a different m, different bytes on disk), and the row matcher would
// accept a row matchRowBytes rejects.  So every entry also carries a 64-bit fingerprint under a per-context SECRET key
// (multiply-xor over the same 8-byte words murmur consumes, keys drawn at bsg_open): equal hashes with different
// fingerprints cannot be crafted without the key.  The device never resolves such a pair itself — the table is flagged
// (status 2: the caller rebuilds that set on its host path), the matcher hands the row to the host matcher.
struct FpKey { uint64_t k0, k1, k2; };   // k1, k2 odd

__device__ __forceinline__ uint64_t fp_word(uint64_t f, uint64_t w, uint64_t k) { return (f ^ w) * k; }
__device__ __forceinline__ uint64_t fp_final(uint64_t f, uint64_t len, const FpKey &key)
{
    f = (f ^ len) * key.k1;
    f ^= f >> 31;
    return f ? f : 1;     // 0 is the "still being written" mark of a table slot
}

// ---------------- streaming murmur3_x64_128 pair (bloom/v3 sum256) ----------------
struct HashStream {
    uint64_t h1, h2, lo, hi;
    uint64_t f;        // running keyed fingerprint over the completed 16-byte blocks
    uint32_t n;
};

__device__ __forceinline__ void hs_init(HashStream &s, const FpKey &key) { s.h1 = s.h2 = s.lo = s.hi = 0; s.f = key.k0; s.n = 0; }
__device__ __forceinline__ void hs_block(HashStream &s, const FpKey &key)
{
    bmix(s.h1, s.h2, s.lo, s.hi);
    s.f = fp_word(fp_word(s.f, s.lo, key.k1), s.hi, key.k2);
}

__device__ __forceinline__ void hs_absorb(HashStream &s, uint32_t byte, const FpKey &key)
{
    const uint32_t t = s.n & 15u;
    const uint64_t v = (uint64_t)byte << ((t & 7u) * 8u);
    if (t < 8u) s.lo |= v; else s.hi |= v;
    s.n += 1;
    if (t == 15u) { hs_block(s, key); s.lo = 0; s.hi = 0; }
}

// (h0,h1) = murmur(d), (h2,h3) = murmur(d || 0x01): same tail handling as base_hashes (kernels.hip.h)
// ... and the entry's keyed fingerprint: the blocks' running value, the tail words, the length
__device__ __forceinline__ uint64_t hs_finish(const HashStream &s, uint64_t h[4], const FpKey &key)
{
    const uint32_t t = s.n & 15u;
    uint64_t k1 = s.lo, k2 = s.hi;
    uint64_t f = s.f;
    if (t > 0) f = fp_word(f, k1, key.k1);
    if (t > 8) f = fp_word(f, k2, key.k2);
    f = fp_final(f, s.n, key);
    {
        uint64_t a1 = s.h1, a2 = s.h2;
        if (t > 8) mix_k2(a2, k2);
        if (t > 0) mix_k1(a1, k1);
        murmur_finalize(a1, a2, s.n, h[0], h[1]);
    }
    {
        if (t < 8) k1 |= 1ULL << (8 * t);
        else       k2 |= 1ULL << (8 * (t - 8));
        uint64_t b1 = s.h1, b2 = s.h2;
        if (t == 15) {
            bmix(b1, b2, k1, k2);
        } else {
            if (t + 1 > 8) mix_k2(b2, k2);
            mix_k1(b1, k1);
        }
        murmur_finalize(b1, b2, (uint64_t)s.n + 1, h[2], h[3]);
    }
    return f;
}

// The same fingerprint straight from an entry's bytes (host-walked entries, condition strings): blob with >= 16 readable
// bytes after the last entry, as base_hashes_words.
__device__ __forceinline__ uint64_t entry_fp(const uint8_t *p, uint32_t len, const FpKey &key)
{
    uint64_t f = key.k0;
    const uint32_t nb = len >> 4;
    for (uint32_t i = 0; i < nb; ++i) f = fp_word(fp_word(f, load_u64_unaligned(p + 16 * i), key.k1), load_u64_unaligned(p + 16 * i + 8), key.k2);
    const uint32_t t = len & 15u;
    uint64_t k1 = load_u64_unaligned(p + 16 * nb), k2 = load_u64_unaligned(p + 16 * nb + 8);
    if (t < 8) { k1 = t ? (k1 & (~0ULL >> (64 - 8 * t))) : 0; k2 = 0; }
    else       { k2 = t > 8 ? (k2 & (~0ULL >> (64 - 8 * (t - 8)))) : 0; }
    if (t > 0) f = fp_word(f, k1, key.k1);
    if (t > 8) f = fp_word(f, k2, key.k2);
    return fp_final(f, len, key);
}

// Absorbs n (1..8) bytes held little-endian in the low bytes of v (bytes above n must be zero).
__device__ __forceinline__ void hs_absorb_n(HashStream &s, uint64_t v, uint32_t n, const FpKey &key)
{
    const uint32_t t = s.n & 15u;
    s.n += n;
    if (t < 8u) {
        s.lo |= v << (t * 8u);
        if (t != 0u && t + n > 8u) s.hi |= v >> ((8u - t) * 8u);
    } else {
        s.hi |= v << ((t - 8u) * 8u);
        if (t + n >= 16u) {
            hs_block(s, key);
            s.lo = (t + n > 16u) ? v >> ((16u - t) * 8u) : 0;   // t > 8 here, so the shift is < 64
            s.hi = 0;
        }
    }
}

// SWAR over the 8 bytes of a chunk.  Each predicate returns 0x80 in the byte lanes that match; bits above the
// LOWEST match may be false positives (borrow propagation), so only the lowest set bit is ever used.
constexpr uint64_t kOnes = 0x0101010101010101ULL, kHighs = 0x8080808080808080ULL;
__device__ __forceinline__ uint64_t swar_eq(uint64_t v, uint32_t c)
{
    const uint64_t x = v ^ (kOnes * c);
    return (x - kOnes) & ~x & kHighs;
}
// bytes that end a run of word bytes inside a string: space, quote, backslash, DEL, controls (< 0x20), non-ASCII (>= 0x80)
__device__ __forceinline__ uint64_t swar_str_stops(uint64_t v)
{
    return swar_eq(v, ' ') | swar_eq(v, '"') | swar_eq(v, '\\') | swar_eq(v, 0x7F) | ((v - kOnes * 0x20u) & ~v & kHighs) | (v & kHighs);
}
// bytes that end a run of key bytes: quote, backslash, DEL, controls (a space is an ordinary key byte, and so is any
// byte >= 0x80: keys are never tokenized, the host copies them undecoded too).  Non-ASCII bytes are replaced by 'a'
// before the comparisons, which are only exact for bytes < 0x80.
__device__ __forceinline__ uint64_t swar_key_stops(uint64_t v)
{
    const uint64_t mm = ((v & kHighs) >> 7) * 0xFFu;         // 0xFF in every non-ASCII byte lane
    const uint64_t x = (v & ~mm) | (mm & (kOnes * 'a'));
    return swar_eq(x, '"') | swar_eq(x, '\\') | swar_eq(x, 0x7F) | ((x - kOnes * 0x20u) & ~x & kHighs);
}
// ASCII A-Z -> a-z in every byte lane (all bytes < 0x80 here)
__device__ __forceinline__ uint64_t swar_lower(uint64_t v)
{
    const uint64_t ge_a = v + kOnes * (0x80u - 'A'), gt_z = v + kOnes * (0x80u - 'Z' - 1u);
    return v | (((ge_a & ~gt_z) & kHighs) >> 2);
}

// ---------------- distinct set insert ----------------
// table slots live in global memory: say so, or every access is a flat_* instruction (the pointer comes out of a struct)
typedef __attribute__((address_space(1))) uint64_t glb_u64;
__device__ __forceinline__ uint64_t ld_agent(const glb_u64 *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(glb_u64 *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One insert in flight: probe position and outcome.
struct InsertState {
    uint32_t idx, probes, spins;
    bool done, present;   // present: the entry was found already stored (a duplicate)
    bool fresh;           // this call stored it
};

// *count += 1 for every lane with `flag`.  The flagged lanes of a wave nearly always share the counter (a wave's rows
// belong to one set; a union workgroup feeds one parent): then one lane adds the popcount.  A single parent counter
// took 1.1 M same-address atomics per 100 blocks before this (k_ingest_union 4.5 ms, all of it that counter).
__device__ __forceinline__ void count_add(uint32_t *count, bool flag)
{
    if (flag) {
        const uint64_t p = (uint64_t)count;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
        const uint64_t p0 = ((uint64_t)hi << 32) | lo;
        const uint64_t active = __ballot(true);
        if (__ballot(p != p0) == 0ull) {
            if (__lane_id() == (uint32_t)__builtin_ctzll(active))
                __hip_atomic_fetch_add(count, (uint32_t)__popcll(active), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ void insert_begin(InsertState &x, const IngestTable t, const uint64_t h[4], bool active, uint32_t *status)
{
    x.idx = table_key(h) >> t.shift;            // h0 is the claim word; the home slot is the top bits of the key (see table_key)
    x.probes = x.spins = 0;
    x.present = false;
    x.fresh = false;
    x.done = !active;
    if (active && (h[0] == 0 || h[1] == 0 || h[2] == 0 || h[3] == 0)) {
        __hip_atomic_fetch_max(status, kTableExotic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x.done = true;
    }
}

// Inserts up to two entries per lane (A into ta, B into tb), their table round trips overlapped: each trip loads
// both candidate slots (all four words: one cache line each), then issues both claims, then settles both.
// present_x reports that the entry was found already stored (a duplicate), not that this call stored it.
// SIMT note: the loop leaves only when EVERY active lane is done (wave-uniform __ballot exit) and the winner's
// stores of h1..h3 sit inside the loop body.  With a per-lane exit the compiler places the winner's "store, then
// leave" block after the loop — executed once the whole wave has left it — so lanes of the same wave racing on one
// entry never see h1..h3 and insert duplicates (measured: 64 equal rows -> 64 "distinct").  Every trip does a
// bounded amount of work (no inner wait).
// COUNT = false leaves the distinct counters alone and reports through fresh_x instead (the caller aggregates).
// fa / fb: the entries' keyed fingerprints.  The winner of a slot stores it after h1..h3; a reader that finds all four hash
// words equal waits for the fingerprint like it waits for h1..h3, and a DIFFERENT fingerprint under equal hashes — a
// murmur3 state collision, i.e. an adversarial entry — flags the table (status 2: the caller rebuilds the set on the host).
template <bool COUNT = true>
__device__ __forceinline__ void set_insert2(const IngestTable ta, const uint64_t ha[4], uint64_t fa, bool active_a, uint32_t *count_a, uint32_t *status_a,
                                            bool &present_a,
                                            const IngestTable tb, const uint64_t hb[4], uint64_t fb, bool active_b, uint32_t *count_b, uint32_t *status_b,
                                            bool &present_b, bool *fresh_a = nullptr, bool *fresh_b = nullptr)
{
    InsertState A, B;
    insert_begin(A, ta, ha, active_a, status_a);
    insert_begin(B, tb, hb, active_b, status_b);
    do {
        glb_u64 *sa = (glb_u64 *)ta.slots + (uint64_t)A.idx * 4, *sb = (glb_u64 *)tb.slots + (uint64_t)B.idx * 4;
        glb_u64 *pa = (glb_u64 *)ta.fps + A.idx, *pb = (glb_u64 *)tb.fps + B.idx;
        uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
        if (!A.done) { a0 = ld_agent(sa); a1 = ld_agent(sa + 1); a2 = ld_agent(sa + 2); a3 = ld_agent(sa + 3); a4 = ld_agent(pa); }
        if (!B.done) { b0 = ld_agent(sb); b1 = ld_agent(sb + 1); b2 = ld_agent(sb + 2); b3 = ld_agent(sb + 3); b4 = ld_agent(pb); }
        const bool cas_a = !A.done && a0 == 0, cas_b = !B.done && b0 == 0;
        uint64_t old_a = 0, old_b = 0;
        bool won_a = false, won_b = false;
        if (cas_a) won_a = __hip_atomic_compare_exchange_strong(sa, &old_a, ha[0], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cas_b) won_b = __hip_atomic_compare_exchange_strong(sb, &old_b, hb[0], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#define BSG_SETTLE(X, won, cas, s, pf, h, f, c0, c1, c2, c3, c4, tab, status)                                   \
        if (!X.done) {                                                                                                   \
            if (won) {                                                                                                   \
                st_agent(s + 1, h[1]); st_agent(s + 2, h[2]); st_agent(s + 3, h[3]); st_agent(pf, f);                    \
                X.done = true; X.fresh = true;   /* a first insert: present stays false, see the cache policy */         \
            } else if (!cas) {            /* (a lost claim looks at the same slot again on the next trip) */             \
                bool advance = true;                                                                                     \
                if (c0 == h[0]) {                                                                                        \
                    if (c1 == h[1] && c2 == h[2] && c3 == h[3] && c4 == f) { X.done = true; X.present = true; advance = false; } \
                    else if (c1 == h[1] && c2 == h[2] && c3 == h[3] && c4 != 0) {   /* equal hashes, another entry */       \
                        __hip_atomic_fetch_max(status, kTableExotic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        \
                        X.done = true; advance = false;                                                                  \
                    }                                                                                                    \
                    else if (c1 == 0 || c2 == 0 || c3 == 0 || (c1 == h[1] && c2 == h[2] && c3 == h[3])) {  /* the claimer's words are still in flight */ \
                        advance = false;                                                                                 \
                        if (++X.spins > kSpinLimit) {              /* never a duplicate slot: give the set to the host */ \
                            __hip_atomic_fetch_max(status, kTableExotic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        \
                            X.done = true;                                                                               \
                        }                                                                                                \
                    }                                                                                                    \
                }                         /* else: a different entry (possibly with the same h0) */                      \
                if (advance) {                                                                                           \
                    X.idx = (X.idx + 1) & tab.mask;                                                                      \
                    X.probes += 1;                                                                                       \
                    if (X.probes > kMaxProbe) {                                                                          \
                        __hip_atomic_fetch_max(status, kTableOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          \
                        X.done = true;                                                                                   \
                    } else if ((X.probes & 15u) == 0u &&                                                                 \
                               __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kTableOk) {      \
                        X.done = true;    /* the table is already flagged: the launch is repeated after it has grown */  \
                    }                                                                                                    \
                }                                                                                                        \
            }                                                                                                            \
        }
        BSG_SETTLE(A, won_a, cas_a, sa, pa, ha, fa, a0, a1, a2, a3, a4, ta, status_a)
        BSG_SETTLE(B, won_b, cas_b, sb, pb, hb, fb, b0, b1, b2, b3, b4, tb, status_b)
#undef BSG_SETTLE
    } while (__ballot(!A.done || !B.done) != 0ull);
    if (COUNT) {
        count_add(count_a, A.fresh);
        count_add(count_b, B.fresh);
    } else {
        *fresh_a = A.fresh;
        *fresh_b = B.fresh;
    }
    present_a = A.present;
    present_b = B.present;
}

__device__ __forceinline__ void set_insert(const IngestTable t, const uint64_t h[4], uint64_t f, uint32_t *count, uint32_t *status)
{
    bool pa, pb;
    set_insert2(t, h, f, true, count, status, pa, t, h, f, false, count, status, pb);
}

// ---------------- row walker ----------------
// One lane per row.  The walker is a RESUMABLE byte automaton: walker_step() runs until the lane has a request
// (hash this path / finish this word), has used up its 8-byte chunk of row bytes, is done, or must hand the row to
// the host.  k_ingest_rows drives a wave in rounds:
//     (A) every lane parses on (divergent, registers + LDS only; chunk loads converged) until it has a request
//     (B) all requests are hashed and inserted together (murmur finalisations and table round trips for 64 lanes at once)
// Measured on MI355X, 1 M log rows: inserts from inside the divergent parse 12.6 ms; this scheme 10.8 ms; with one
// counter add per wave instead of one per stored entry (count_add) 2.85 ms — see profiles/README.md.
struct IngestArgs {
    const uint8_t *rows;            // 8-byte aligned, >= 16 readable bytes after the last row
    const uint64_t *row_off;        // [n_rows + 1]
    const uint32_t *set_first_row;  // [n_sets + 1], ascending
    const IngestTable *tables;      // [n_sets_total * 3]
    uint32_t *counts;               // per table
    uint32_t *status;               // per table
    uint32_t *fallback_rows;        // rows the host walker must finish
    uint32_t *n_fallback;
    const uint32_t *lower;          // [0x20000]: unicode.ToLower(cp) where it differs from cp, 0 = unchanged, 0xFFFFFFFF = unknown code point (host)
    uint32_t n_rows;
    uint32_t n_sets;
    uint32_t validate;              // 1: run the validation pass first (rows of unknown provenance)
    uint32_t row_first, row_end;    // this launch walks rows [row_first, row_end): a chunk whose bytes have landed
    uint32_t rows_per_wave;         // multiple of 64: wave w of the launch walks rows [row_first + w * rows_per_wave, + rows_per_wave)
    FpKey key;                      // the context's fingerprint key
};

typedef __attribute__((address_space(3))) uint8_t lds_u8;

enum : uint32_t {
    S_VALUE, S_VALUE_OR_CLOSE, S_KEY_OR_CLOSE, S_KEY_OPEN, S_KEY, S_COLON, S_PREFIX, S_STR, S_NUM, S_LIT, S_AFTER,
    S_STR_ESC, S_STR_U, S_KEY_ESC, S_KEY_U,  // after a backslash / inside \uXXXX, in a string value / in a key
    S_STR_UTF8,                              // inside a multi-byte UTF-8 sequence of a string value
    S_STR_SUR_BS, S_STR_SUR_U,               // after a \uD8xx high surrogate of a string value: the backslash and the 'u' of its low half
    S_KEY_SUR_BS, S_KEY_SUR_U                // the same inside a key
};
enum : uint32_t { R_CONTINUE, R_DONE, R_FAIL };
// What a lane asks the converged part of the loop to do for it (hashing is the expensive part of an emission —
// two murmur3 finalisations per hash — so it runs once per round for all lanes instead of inside the divergent parse).
enum : uint32_t {
    Q_NONE,
    Q_FIELD,   // path[0, len) is a field entry (container path or key prefix)
    Q_LEAF,    // path[0, len) is a field entry AND the leaf's words continue from hash state(path + "::")
    Q_WORD,    // a word ended: tok -> token entry, ft -> field::token entry
};

struct Walker {
    uint64_t pos, end;
    uint64_t cur;            // row bytes [pos & ~7, +8)
    lds_u8 *path;            // this lane's path buffer [kPathCap] followed by the container stack [kMaxDepth]
    uint32_t path_len, depth, objmask, st;
    uint32_t aux;            // S_PREFIX: scan index; S_NUM: phase; S_LIT: matched chars
    uint32_t key_len;        // path length including the key being read (S_KEY .. S_PREFIX)
    uint32_t lit;            // S_LIT: 0 true, 1 false, 2 null; S_*_U: hex digits seen << 16 | value so far
    uint32_t req, req_len;   // pending request
    const uint32_t *lower;   // IngestArgs::lower
    FpKey key;               // fingerprint key of the context (uniform)
    bool quiet;              // leaf without a path (scalars at the root): nothing is emitted
    bool in_token;
    bool ft_on;              // keep the path::word stream (the ingest walker); the row matcher only needs the word's own hash
#ifdef BSG_INGEST_PROF
    uint32_t iters;          // lab: trips of the walker_step loop
#endif
    HashStream ps, tok, ft;  // path + "::" prefix state; current word; current path::word
};

__device__ __forceinline__ bool is_plain(uint32_t c) { return c >= 0x20u && c < 0x7Fu && c != '\\'; }

// EMIT = false is the validation pass: same automaton, no hashing, no requests.
template <bool EMIT>
__device__ __forceinline__ void leaf_start(Walker &w)
{
    w.quiet = !EMIT || w.path_len == 0;
    w.in_token = false;
    if (!w.quiet) { w.req = Q_LEAF; w.req_len = w.path_len; }
}

__device__ __forceinline__ void word_byte(Walker &w, uint32_t c)
{
    if (w.quiet) return;
    if (!w.in_token) { hs_init(w.tok, w.key); w.ft = w.ps; w.in_token = true; }
    if (c - 'A' < 26u) c += 32;                    // ASCII fold (appendFoldedWord fast path, row_matcher.go:187-202)
    hs_absorb(w.tok, c, w.key);
    if (w.ft_on) hs_absorb(w.ft, c, w.key);
}

// n (1..8) word bytes, little-endian in v, upper bytes zero
__device__ __forceinline__ void word_run(Walker &w, uint64_t v, uint32_t n)
{
    if (w.quiet) return;
    if (!w.in_token) { hs_init(w.tok, w.key); w.ft = w.ps; w.in_token = true; }
    v = swar_lower(v);                             // ASCII fold (appendFoldedWord fast path, row_matcher.go:187-202)
    hs_absorb_n(w.tok, v, n, w.key);
    if (w.ft_on) hs_absorb_n(w.ft, v, n, w.key);
}

__device__ __forceinline__ uint32_t c_at(uint64_t v, uint32_t i) { return (uint32_t)(v >> (i * 8u)) & 0xFFu; }

__device__ __forceinline__ bool word_end(Walker &w)
{
    if (w.quiet || !w.in_token) return false;
    w.in_token = false;
    w.req = Q_WORD;
    return true;
}

// Number grammar -? digits (. digits)? ([eE] [+-]? digits)? as phases: 7 start, 0 after '-', 1 int digits,
// 2 after '.', 3 fraction digits, 4 after e, 5 after exponent sign, 6 exponent digits.  Next phase or 0xFF.
__device__ __forceinline__ uint32_t num_next(uint32_t phase, uint32_t c)
{
    const bool digit = c - '0' <= 9u;
    switch (phase) {
    case 7: return digit ? 1u : (c == '-' ? 0u : 0xFFu);
    case 0: return digit ? 1u : 0xFFu;
    case 1: return digit ? 1u : (c == '.' ? 2u : ((c | 0x20u) == 'e' ? 4u : 0xFFu));
    case 2: return digit ? 3u : 0xFFu;
    case 3: return digit ? 3u : ((c | 0x20u) == 'e' ? 4u : 0xFFu);
    case 4: return digit ? 6u : ((c == '+' || c == '-') ? 5u : 0xFFu);
    case 5: return digit ? 6u : 0xFFu;
    default: return digit ? 6u : 0xFFu;
    }
}
__device__ __forceinline__ bool num_accepting(uint32_t phase) { return phase == 1u || phase == 3u || phase == 6u; }

// JSON escapes the device decodes itself: \" \\ \/ \b \f \n \r \t and \u00XX below 0x80 (Go's json.Marshal writes <, >, &
// as \u003c, \u003e, \u0026, so ordinary log text is full of them).  Anything that decodes to a non-ASCII rune goes to
// the host walker, which owns UTF-8 encoding and the Unicode space / case tables.
__device__ __forceinline__ uint32_t simple_escape(uint32_t c)
{
    switch (c) {
    case '"': return '"';
    case '\\': return '\\';
    case '/': return '/';
    case 'b': return 0x08;
    case 'f': return 0x0C;
    case 'n': return 0x0A;
    case 'r': return 0x0D;
    case 't': return 0x09;
    default: return 0xFFFFu;
    }
}
__device__ __forceinline__ uint32_t hex_value(uint32_t c)
{
    if (c - '0' <= 9u) return c - '0';
    const uint32_t l = c | 0x20u;
    return (l - 'a' <= 5u) ? l - 'a' + 10u : 0xFFu;
}
// unicode.IsSpace below 0x80: \t \n \v \f \r and space (forEachWord, row_matcher.go:142-181)
__device__ __forceinline__ bool ascii_space(uint32_t b) { return b == ' ' || (b - 9u) <= 4u; }

// a decoded (escaped) byte of a string value: white space ends the word, anything else belongs to it
__device__ __forceinline__ bool str_decoded_byte(Walker &w, uint32_t b)
{
    if (ascii_space(b)) return word_end(w);
    word_byte(w, b);
    return false;
}

// unicode.IsSpace above 0x7F (text.hpp is_space)
__device__ __forceinline__ bool rune_space(uint32_t r)
{
    return r == 0x85u || r == 0xA0u || r == 0x1680u || (r - 0x2000u) <= 0xAu || r == 0x2028u || r == 0x2029u || r == 0x202Fu ||
           r == 0x205Fu || r == 0x3000u;
}
// UTF-8 encoding of a rune >= 0x80 (not a surrogate, <= 0x10FFFF) as little-endian bytes in a u32; n = its length
__device__ __forceinline__ uint32_t rune_utf8(uint32_t r, uint32_t &n)
{
    if (r < 0x800u) { n = 2; return (0xC0u | (r >> 6)) | ((0x80u | (r & 0x3Fu)) << 8); }
    if (r < 0x10000u) { n = 3; return (0xE0u | (r >> 12)) | ((0x80u | ((r >> 6) & 0x3Fu)) << 8) | ((0x80u | (r & 0x3Fu)) << 16); }
    n = 4;
    return (0xF0u | (r >> 18)) | ((0x80u | ((r >> 12) & 0x3Fu)) << 8) | ((0x80u | ((r >> 6) & 0x3Fu)) << 16) | ((0x80u | (r & 0x3Fu)) << 24);
}
// A non-ASCII rune of a string value (raw UTF-8 or \uXXXX): Unicode white space ends the word; anything else joins it
// as the UTF-8 bytes of unicode.ToLower(rune) — the simple one-rune mapping, looked up in the table the host walker
// folds with (the lower-case form may be shorter or longer than the original, or plain ASCII: U+212A -> 'k').
// Returns R_CONTINUE + request, or 0xFF = keep going.
__device__ __forceinline__ uint32_t str_rune(Walker &w, uint32_t r)
{
    if (rune_space(r)) return word_end(w) ? R_CONTINUE : 0xFFu;
    // (the validation pass looks the rune up too: a code point these tables do not know sends the row to the host, whose
    // own unicode.ToLower decides — a newer Unicode may have made it a cased letter)
    const uint32_t lo = r < 0x20000u ? w.lower[r] : 0u;
    if (lo == 0xFFFFFFFFu) return R_FAIL;
    if (!w.quiet) {
        if (lo != 0u) r = lo;
        uint32_t n = 1, bytes = r;
        if (r >= 0x80u) bytes = rune_utf8(r, n);
        if (!w.in_token) { hs_init(w.tok, w.key); w.ft = w.ps; w.in_token = true; }
        hs_absorb_n(w.tok, bytes, n, w.key);
        if (w.ft_on) hs_absorb_n(w.ft, bytes, n, w.key);
    }
    return 0xFFu;
}

struct ChunkCursor {
    const uint64_t *chunks;
    uint64_t ci;     // index of the chunk in Walker::cur
    uint64_t nxt;    // one chunk ahead: its latency hides behind the work on the current one
};

// Runs until the lane has a request pending (w.req), is done, or must go to the host.  A lane that uses up its 8-byte
// chunk takes the next one — already in a register, requested one chunk ago — and requests the one after, right here:
// returning to the wave-converged loop for it put the lanes of a wave out of phase with each other at every chunk
// boundary (different lanes cross theirs at different steps), and every extra trip runs all the states some lane is in
// (10 M rows: 31.7 -> 27 ms; the row matcher 15.9 -> 12 ms).  Measured and dropped on top of it: skipping S_COLON /
// S_KEY_OPEN when their byte sits in the same chunk and a has-dots flag for S_PREFIX (no change), the rare escape /
// UTF-8 states behind one test (slower: 8.9 -> 11.1 ms per 3 M rows, the structurizer's layout got worse), S_STR / S_KEY
// runs continuing into the next chunk within one trip (8.76 -> 9.05 ms).
template <bool EMIT>
__device__ __forceinline__ uint32_t walker_step(Walker &w, ChunkCursor &cc)
{
    lds_u8 *stack = w.path + kPathCap;
    if ((w.pos >> 3) != cc.ci) {                      // the step before ended on the chunk boundary
        cc.ci += 1;
        w.cur = cc.nxt;
        cc.nxt = cc.chunks[cc.ci + 1];
    }
    uint64_t chunk_end = (cc.ci + 1) * 8;             // bytes of w.cur end here
    for (;;) {
#ifdef BSG_INGEST_PROF
        ++w.iters;
#endif
        if (w.st == S_PREFIX) {
            // every "."-split prefix of the key is a field entry, empty paths skipped (row_matcher.go:103-135)
            while (EMIT && w.aux < w.key_len) {
                const uint32_t j = w.aux++;
                if (w.path[j] == '.' && j > 0) { w.req = Q_FIELD; w.req_len = j; return R_CONTINUE; }
            }
            w.path_len = w.key_len;
            w.st = S_VALUE;
        }
        if (w.pos >= w.end) {
            if (w.st == S_NUM && num_accepting(w.aux)) {          // a number ends with the row
                w.st = S_AFTER;
                if (word_end(w)) return R_CONTINUE;
            }
            return (w.st == S_AFTER && w.depth == 0) ? R_DONE : R_FAIL;
        }
        if (w.pos >= chunk_end) { cc.ci += 1; w.cur = cc.nxt; cc.nxt = cc.chunks[cc.ci + 1]; chunk_end += 8; }
        const uint32_t c = (uint32_t)(w.cur >> ((w.pos & 7u) * 8u)) & 0xFFu;
        switch (w.st) {
        case S_VALUE_OR_CLOSE:
            if (c == ' ') { ++w.pos; break; }
            if (c == ']') { ++w.pos; --w.depth; w.st = S_AFTER; break; }
            w.st = S_VALUE;
            break;
        case S_KEY_OR_CLOSE:
            if (c == ' ') { ++w.pos; break; }
            if (c == '}') { ++w.pos; --w.depth; w.st = S_AFTER; break; }
            w.st = S_KEY_OPEN;
            break;
        case S_VALUE:
            if (c == ' ') { ++w.pos; break; }
            if (c == '{' || c == '[') {
                if (w.depth >= kMaxDepth) return R_FAIL;
                stack[w.depth] = (uint8_t)w.path_len;
                if (c == '{') w.objmask |= 1u << w.depth; else w.objmask &= ~(1u << w.depth);
                ++w.depth;
                ++w.pos;
                w.st = (c == '{') ? S_KEY_OR_CLOSE : S_VALUE_OR_CLOSE;
                if (EMIT && w.path_len > 0) { w.req = Q_FIELD; w.req_len = w.path_len; return R_CONTINUE; }   // container path: non-leaf emission
                break;
            }
            // a primitive: its first byte stays unconsumed until the leaf request has been served (the words' hash
            // state continues from the path's)
            if (c == '"') { ++w.pos; w.st = S_STR; }
            else if (c == 't' || c == 'f' || c == 'n') { w.lit = c == 't' ? 0u : (c == 'f' ? 1u : 2u); w.aux = 0; w.st = S_LIT; }
            else if (c == '-' || c - '0' <= 9u) { w.aux = 7; w.st = S_NUM; }
            else return R_FAIL;
            leaf_start<EMIT>(w);
            if (w.req != Q_NONE) return R_CONTINUE;
            break;
        case S_KEY_OPEN:
            if (c == ' ') { ++w.pos; break; }
            if (c != '"') return R_FAIL;
            ++w.pos;
            w.key_len = w.path_len;
            if (w.key_len > 0) {
                if (w.key_len >= kPathCap) return R_FAIL;
                w.path[w.key_len++] = '.';
            }
            w.aux = w.key_len;                                      // first byte of the key inside the path buffer
            w.st = S_KEY;
            break;
        case S_KEY: {
            // the whole run of key bytes left in this chunk at once
            const uint32_t k = (uint32_t)(w.pos & 7u);
            uint32_t avail = 8u - k;
            if (w.end - w.pos < avail) avail = (uint32_t)(w.end - w.pos);
            const uint64_t v = w.cur >> (k * 8u);
            const uint64_t stops = swar_key_stops(v);
            uint32_t n = stops ? (uint32_t)(__builtin_ctzll(stops) >> 3) : 8u;
            if (n > avail) n = avail;
            if (w.key_len + n > kPathCap) return R_FAIL;
            for (uint32_t i = 0; i < n; ++i) w.path[w.key_len + i] = (uint8_t)(v >> (i * 8u));
            w.key_len += n;
            w.pos += n;
            if (n < avail) {                                        // stopped on a byte inside the chunk
                const uint32_t b = c_at(v, n);
                ++w.pos;
                if (b == '"') w.st = S_COLON;
                else if (b == '\\') w.st = S_KEY_ESC;
                else return R_FAIL;
            }
            break;
        }
        case S_KEY_ESC:
        case S_STR_ESC: {
            ++w.pos;
            const bool key = w.st == S_KEY_ESC;
            if (c == 'u') { w.lit = 0; w.req_len = 0; w.st = key ? S_KEY_U : S_STR_U; break; }   // req_len (free while no request is pending) = high surrogate waiting
            const uint32_t b = simple_escape(c);
            if (b == 0xFFFFu) return R_FAIL;
            w.st = key ? S_KEY : S_STR;
            if (key) {
                if (w.key_len >= kPathCap) return R_FAIL;
                w.path[w.key_len++] = (uint8_t)b;
            } else if (str_decoded_byte(w, b)) {
                return R_CONTINUE;
            }
            break;
        }
        case S_KEY_U:
        case S_STR_U: {
            const uint32_t h = hex_value(c);
            if (h == 0xFFu) return R_FAIL;
            ++w.pos;
            const uint32_t seen = (w.lit >> 16) + 1, value = ((w.lit & 0xFFFFu) << 4) | h;
            w.lit = (seen << 16) | value;
            if (seen < 4) break;
            const bool key = w.st == S_KEY_U;
            uint32_t cp = value;
            if (w.req_len != 0u) {                                  // second half of a surrogate pair
                if ((value - 0xDC00u) >= 0x400u) return R_FAIL;     // not a low surrogate: the high one was lone (-> U+FFFD): host
                cp = 0x10000u + ((w.req_len - 0xD800u) << 10) + (value - 0xDC00u);
                w.req_len = 0;
            } else if ((value - 0xD800u) < 0x800u) {
                if (value >= 0xDC00u) return R_FAIL;                // a lone low surrogate: the host walker's
                w.req_len = value;                                  // a high surrogate: its low half must follow immediately
                w.st = key ? S_KEY_SUR_BS : S_STR_SUR_BS;
                break;
            }
            w.st = key ? S_KEY : S_STR;
            if (key) {
                uint32_t n = 1, bytes = cp;
                if (cp >= 0x80u) bytes = rune_utf8(cp, n);
                if (w.key_len + n > kPathCap) return R_FAIL;
                for (uint32_t i = 0; i < n; ++i) w.path[w.key_len++] = (uint8_t)(bytes >> (8u * i));
            } else if (cp < 0x80u) {
                if (str_decoded_byte(w, cp)) return R_CONTINUE;
            } else {
                const uint32_t r = str_rune(w, cp);
                if (r != 0xFFu) return r;
            }
            break;
        }
        case S_KEY_SUR_BS:
        case S_STR_SUR_BS:
            if (c != '\\') return R_FAIL;
            ++w.pos;
            w.st = w.st == S_KEY_SUR_BS ? S_KEY_SUR_U : S_STR_SUR_U;
            break;
        case S_KEY_SUR_U:
        case S_STR_SUR_U:
            if (c != 'u') return R_FAIL;
            ++w.pos;
            w.lit = 0;
            w.st = w.st == S_KEY_SUR_U ? S_KEY_U : S_STR_U;         // w.req_len != 0 tells S_*_U which half this is
            break;
        case S_STR_UTF8: {
            // w.lit: bytes still expected << 28 | lowest / highest value allowed for THIS byte << 8 / << 16 (the second byte
            // of E0, ED, F0, F4 is restricted: no overlongs, no surrogates, nothing above U+10FFFF) ; w.aux: rune so far
            const uint32_t lo = (w.lit >> 8) & 0xFFu, hi = (w.lit >> 16) & 0xFFu, left = w.lit >> 28;
            if (c < lo || c > hi) return R_FAIL;                    // invalid UTF-8 becomes U+FFFD per byte: the host walker's
            ++w.pos;
            w.aux = (w.aux << 6) | (c & 0x3Fu);
            if (left > 1) { w.lit = ((left - 1) << 28) | (0xBFu << 16) | (0x80u << 8); break; }
            w.st = S_STR;
            const uint32_t r = str_rune(w, w.aux);
            if (r != 0xFFu) return r;
            break;
        }
        case S_COLON:
            if (c == ' ') { ++w.pos; break; }
            if (c != ':') return R_FAIL;
            ++w.pos;
            w.st = S_PREFIX;                                        // w.aux still points at the key's first byte
            break;
        case S_STR: {
            // the run of word bytes left in this chunk goes into both hash streams at once
            const uint32_t k = (uint32_t)(w.pos & 7u);
            uint32_t avail = 8u - k;
            if (w.end - w.pos < avail) avail = (uint32_t)(w.end - w.pos);
            const uint64_t v = w.cur >> (k * 8u);
            const uint64_t stops = swar_str_stops(v);
            uint32_t n = stops ? (uint32_t)(__builtin_ctzll(stops) >> 3) : 8u;
            if (n > avail) n = avail;
            if (n != 0u) {
                word_run(w, n == 8u ? v : (v & ((1ULL << (n * 8u)) - 1ULL)), n);
                w.pos += n;
            }
            if (n < avail) {                                        // stopped on a byte inside the chunk
                const uint32_t b = c_at(v, n);
                ++w.pos;
                if (b == '"') { w.st = S_AFTER; if (word_end(w)) return R_CONTINUE; }
                else if (b == ' ') { if (word_end(w)) return R_CONTINUE; }   // raw white space other than 0x20 fails here
                else if (b == '\\') w.st = S_STR_ESC;
                else if (b >= 0xC2u && b <= 0xF4u) {                // UTF-8 lead byte (decode_rune, text.hpp)
                    uint32_t left, lo = 0x80u, hi = 0xBFu;
                    if (b < 0xE0u) { left = 1; w.aux = b & 0x1Fu; }
                    else if (b < 0xF0u) { left = 2; w.aux = b & 0x0Fu; if (b == 0xE0u) lo = 0xA0u; if (b == 0xEDu) hi = 0x9Fu; }
                    else { left = 3; w.aux = b & 0x07u; if (b == 0xF0u) lo = 0x90u; if (b == 0xF4u) hi = 0x8Fu; }
                    w.lit = (left << 28) | (hi << 16) | (lo << 8);
                    w.st = S_STR_UTF8;
                }
                else return R_FAIL;
            }
            break;
        }
        case S_NUM: {                                               // the token is the RAW literal (tokenizer.go:124-125)
            const uint32_t nx = num_next(w.aux, c);
            if (nx != 0xFFu) { w.aux = nx; word_byte(w, c); ++w.pos; break; }
            if (!num_accepting(w.aux)) return R_FAIL;
            w.st = S_AFTER;                                         // c is not part of the number: S_AFTER looks at it
            if (word_end(w)) return R_CONTINUE;
            break;
        }
        case S_LIT: {
            // "true" / "fals"(+e) / "null" packed little-endian: no table lookup in divergent code
            const uint32_t packed = w.lit == 0u ? 0x65757274u : (w.lit == 1u ? 0x736c6166u : 0x6c6c756eu);
            const uint32_t want = w.aux == 4u ? (uint32_t)'e' : ((packed >> (8u * w.aux)) & 0xFFu);
            if (c != want) return R_FAIL;
            ++w.pos;
            ++w.aux;
            if (w.lit != 2u) word_byte(w, c);                       // null: field existence only (tokenizer.go:130-131)
            if (w.aux == (w.lit == 1u ? 5u : 4u)) {
                w.st = S_AFTER;
                if (word_end(w)) return R_CONTINUE;
            }
            break;
        }
        default: {  // S_AFTER
            if (c == ' ') { ++w.pos; break; }
            if (w.depth == 0) return R_FAIL;                        // nothing but spaces may follow the row's value
            w.path_len = stack[w.depth - 1];                        // array elements reuse the array's path; members restart from the object's
            const bool obj = (w.objmask >> (w.depth - 1)) & 1u;
            if (c == ',') { ++w.pos; w.st = obj ? S_KEY_OPEN : S_VALUE; break; }
            if (c == (obj ? '}' : ']')) { ++w.pos; --w.depth; break; }
            return R_FAIL;
        }
        }
    }
}

// ---------------- workgroup dedup cache ----------------
// Most emissions repeat entries this workgroup has just inserted (field paths, levels, message words ...).
// An LDS cache of confirmed inserts answers those without touching the table.  Lanes update it without locking: a
// torn entry can only produce a false hit for an entry that agrees with unrelated entries on 64 hash bits each AND on
// the keyed fingerprint.
#ifndef BSG_INGEST_CACHE_SETS
#define BSG_INGEST_CACHE_SETS 448
#endif
#ifndef BSG_INGEST_WPE
#define BSG_INGEST_WPE 3
#endif
// 2-way sets of (h0 ^ table, h1, fingerprint): the first murmur3 hash of the entry and its keyed fingerprint — what the
// table itself needs to tell two entries apart once their hashes agree.  Way 0 is the most recently confirmed entry of
// the set; a new one pushes it to way 1, so two hot entries that share a set do not evict each other every round.
// (By itself that moved 15.7 -> 14.7 of a row's 32 rounds to the table: most of them came from COLD caches, which the
// contiguous runs of rows per wave fix — see k_ingest_rows.)
constexpr uint32_t kCacheSets = BSG_INGEST_CACHE_SETS;  // lab: -DBSG_INGEST_CACHE_SETS / -DBSG_INGEST_WPE (waves per SIMD the kernel is compiled for)
constexpr uint32_t kCacheEntryWords = 3;
constexpr uint32_t kCacheWords = kCacheSets * 2 * kCacheEntryWords;
constexpr uint32_t kIngestLdsBytes = kCacheWords * 8 + kIngestThreads * kLaneLds;   // cache, then the lanes' path buffers
typedef __attribute__((address_space(3))) uint64_t lds_u64i;

__device__ __forceinline__ uint32_t cache_set(const uint64_t h[4]) { return ((((uint32_t)(h[1] >> 8)) & 0xFFFFu) * kCacheSets) >> 16; }
__device__ __forceinline__ uint64_t cache_tag(uint32_t table_id, const uint64_t h[4]) { return h[0] ^ ((uint64_t)(table_id + 1) * 0x9E3779B97F4A7C15ULL); }

__device__ __forceinline__ bool cache_hit(const lds_u64i *cache, uint32_t table_id, const uint64_t h[4], uint64_t fp)
{
    const uint64_t tag = cache_tag(table_id, h);
    const lds_u64i *e = cache + (size_t)cache_set(h) * 2 * kCacheEntryWords;
    // (an entry whose hashes agree with a cached one but whose fingerprint differs goes to the table, which flags the pair)
    return (e[0] == tag && e[1] == h[1] && e[2] == fp) || (e[3] == tag && e[4] == h[1] && e[5] == fp);
}
__device__ __forceinline__ void cache_put(lds_u64i *cache, uint32_t table_id, const uint64_t h[4], uint64_t fp)
{
    const uint64_t tag = cache_tag(table_id, h);
    lds_u64i *e = cache + (size_t)cache_set(h) * 2 * kCacheEntryWords;
    const uint64_t o0 = e[0], o1 = e[1], o2 = e[2];
    if (o0 == tag && o1 == h[1] && o2 == fp) return;
    e[3] = o0; e[4] = o1; e[5] = o2;
    e[0] = tag; e[1] = h[1]; e[2] = fp;
}


// lab only (-DBSG_INGEST_PROF): per-phase wave cycles accumulated behind *n_fallback (slots 1..7 as u64)
#ifdef BSG_INGEST_PROF
#define BSG_PROF_DECL uint64_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BSG_PROF_T(var) const uint64_t var = __builtin_readcyclecounter()
#define BSG_PROF_ADD(slot, t0, t1) prof_acc[slot] += (uint64_t)((t1) - (t0))
#define BSG_PROF_FLUSH() do { if ((threadIdx.x & 63u) == 0u) { for (int pi = 1; pi < 8; ++pi) atomicAdd((unsigned long long *)a.n_fallback + pi, (unsigned long long)prof_acc[pi]); atomicAdd((unsigned long long *)a.n_fallback + 8, (unsigned long long)prof_acc[0]); } } while (0)
#else
#define BSG_PROF_DECL
#define BSG_PROF_T(var)
#define BSG_PROF_ADD(slot, t0, t1)
#define BSG_PROF_FLUSH()
#endif

__device__ __forceinline__ void walker_reset(Walker &w, ChunkCursor &cc, uint64_t pos, uint64_t end, bool live)
{
    w.pos = pos; w.end = end;
    w.path_len = w.depth = w.objmask = 0;
    w.st = S_VALUE;
    w.aux = w.key_len = w.lit = 0;
    w.req = Q_NONE; w.req_len = 0;
    w.quiet = true;
    w.in_token = false;
    cc.ci = pos >> 3;
    w.cur = live ? cc.chunks[cc.ci] : 0;
    cc.nxt = live ? cc.chunks[cc.ci + 1] : 0;
}

template <bool EMIT>
__device__ __forceinline__ uint32_t advance(Walker &w, ChunkCursor &cc) { return walker_step<EMIT>(w, cc); }

// One wave walks a contiguous run of rows_per_wave rows, 64 at a time (lane = row); the four waves of a workgroup take
// neighbouring runs, so a workgroup stays inside one block (or two) and its dedup cache stays warm: a workgroup that
// sees each lane's row only once sends ~15 of a row's 32 emission rounds to the table (every entry is new to IT), one
// that has walked a few rows per lane only the 2-3 rounds whose entries are new to the block (timestamps, ids).
__global__ __launch_bounds__(kIngestThreads, BSG_INGEST_WPE) void k_ingest_rows(const IngestArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    lds_u64i *cache = (lds_u64i *)lds_raw;
    for (uint32_t i = threadIdx.x; i < kCacheWords; i += kIngestThreads) cache[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (kIngestThreads / 64) + (threadIdx.x >> 6);
    const uint64_t run_begin = a.row_first + wave * a.rows_per_wave;
    const uint32_t run_end = (uint32_t)(run_begin + a.rows_per_wave < a.row_end ? run_begin + a.rows_per_wave : a.row_end);
    Walker w;
    ChunkCursor cc;
    cc.chunks = reinterpret_cast<const uint64_t *>(a.rows);
    w.path = (lds_u8 *)lds_raw + kCacheWords * 8 + threadIdx.x * kLaneLds;
    w.lower = a.lower;
    w.key = a.key;
#ifdef BSG_LAB_NOFT      // lab only: what the field::token stream costs inside the parse (the sets come out wrong)
    w.ft_on = false;
#else
    w.ft_on = true;
#endif
    BSG_PROF_DECL;
    for (uint64_t tile = run_begin; tile < run_end; tile += 64) {
    const uint32_t r = (uint32_t)tile + lane;
    const bool live = r < run_end;
    // the set this row belongs to: last s with set_first_row[s] <= r
    uint32_t lo = 0, hi = a.n_sets;
    while (live && hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.set_first_row[mid] <= r) lo = mid; else hi = mid;
    }
    const uint32_t t0 = lo * 3;
    const uint64_t row_begin = live ? a.row_off[r] : 0, row_end = live ? a.row_off[r + 1] : 0;
    hs_init(w.ps, w.key); hs_init(w.tok, w.key); hs_init(w.ft, w.key);

    // pass 1: validate.  A row the device walker cannot finish contributes NOTHING here; it goes to the host walker whole.
    BSG_PROF_T(p0);
    uint32_t res = R_DONE;
    if (a.validate) {
        walker_reset(w, cc, row_begin, row_end, live);
        res = live ? R_CONTINUE : R_DONE;
        while (__ballot(res == R_CONTINUE) != 0ull)
            if (res == R_CONTINUE) res = advance<false>(w, cc);
        if (res == R_FAIL) {
            const uint32_t slot = __hip_atomic_fetch_add(a.n_fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.fallback_rows[slot] = r;
        }
    }

    // pass 2: emit.  Rounds of  (A) every lane parses on until it has a request,  (B) all requests are hashed and
    // inserted together.
    const bool go = live && res == R_DONE;
    BSG_PROF_T(p1);
    BSG_PROF_ADD(1, p0, p1);
    walker_reset(w, cc, row_begin, row_end, go);
    res = go ? R_CONTINUE : R_DONE;
    while (__ballot(res == R_CONTINUE || w.req != Q_NONE) != 0ull) {
        BSG_PROF_T(ta0);
#ifdef BSG_INGEST_PROF
        w.iters = 0;
#endif
        while (__ballot(res == R_CONTINUE && w.req == Q_NONE) != 0ull)
            if (res == R_CONTINUE && w.req == Q_NONE) res = advance<true>(w, cc);
        BSG_PROF_T(ta1);
        BSG_PROF_ADD(2, ta0, ta1);
        BSG_PROF_ADD(5, 0, 1);
#ifdef BSG_INGEST_PROF
        {   // trips of the wave = those of its slowest lane; and the lanes' own mean
            uint32_t mx = w.iters, sm = w.iters;
            for (int o = 32; o > 0; o >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); sm += (uint32_t)__shfl_xor((int)sm, o); }
            prof_acc[7] += mx;
            prof_acc[0] += sm;
        }
#endif
        const uint32_t q = w.req;
        w.req = Q_NONE;
        // (B1) path hashes for Q_FIELD / Q_LEAF
        HashStream s;
        hs_init(s, w.key);
        const uint32_t plen = (q == Q_FIELD || q == Q_LEAF) ? w.req_len : 0u;
        for (uint32_t i = 0; __ballot(i < plen) != 0ull; ++i)
            if (i < plen) hs_absorb(s, w.path[i], w.key);
        if (q == Q_WORD) s = w.tok;
        uint64_t ha[4], hb[4], fa = 0, fb = 0;
        if (q != Q_NONE) fa = hs_finish(s, ha, w.key);
        if (q == Q_LEAF) {                          // the leaf's words continue from path + "::" (makeFieldTokenKey, tokenizer.go:509-511)
            w.ps = s;
            hs_absorb(w.ps, ':', w.key);
            hs_absorb(w.ps, ':', w.key);
        }
        if (q == Q_WORD) fb = hs_finish(w.ft, hb, w.key);
        BSG_PROF_T(tb1);
        BSG_PROF_ADD(3, ta1, tb1);
        // (B2) inserts.  Only entries met as duplicates are cached: a first insert is usually a row-unique value
        // (timestamp, id) that would only push hot entries out of the cache.
        const uint32_t ta = t0 + (q == Q_WORD ? 1u : 0u), tb = t0 + 2u;
        const bool need_a = q != Q_NONE && !cache_hit(cache, ta, ha, fa);
        const bool need_b = q == Q_WORD && !cache_hit(cache, tb, hb, fb);
        if (__ballot(need_a || need_b) != 0ull) {
            bool pa, pb;
            set_insert2(a.tables[ta], ha, fa, need_a, a.counts + ta, a.status + ta, pa,
                        a.tables[tb], hb, fb, need_b, a.counts + tb, a.status + tb, pb);
            if (pa) cache_put(cache, ta, ha, fa);
            if (pb) cache_put(cache, tb, hb, fb);
            BSG_PROF_ADD(6, 0, 1);
        }
        BSG_PROF_T(tb2);
        BSG_PROF_ADD(4, tb1, tb2);
    }
    // without the validation pass a row is flagged when the emitting walk gives up on it: what it inserted before that
    // is a subset of what the host walker inserts for it PROVIDED the row is valid JSON (the caller's promise).
    // (After a validation pass the emitting walk cannot fail — both run the same automaton — but a row is never dropped.)
    if (res == R_FAIL) {
        const uint32_t slot = __hip_atomic_fetch_add(a.n_fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.fallback_rows[slot] = r;
    }
    }
    BSG_PROF_FLUSH();
}

// ---------------- host-walked entries (fallback rows) ----------------
__global__ __launch_bounds__(256) void k_ingest_add(const uint8_t *bytes, const uint32_t *off, const uint32_t *table_of_entry,
                                                    uint32_t n, const IngestTable *tables, uint32_t *counts, uint32_t *status, const FpKey key)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    uint64_t h[4];
    base_hashes_at(bytes, off[e], off[e + 1] - off[e], h);
    const uint64_t f = entry_fp(bytes + off[e], off[e + 1] - off[e], key);
    const uint32_t t = table_of_entry[e];
    set_insert(tables[t], h, f, counts + t, status + t);
}

// hashes + fingerprints of packed strings (the row matcher's condition strings)
__global__ __launch_bounds__(256) void k_hash_fp_entries(const uint8_t *bytes, const uint32_t *off, uint32_t n, uint64_t *out_h, uint64_t *out_fp,
                                                         const FpKey key)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    uint64_t h[4];
    base_hashes_at(bytes, off[e], off[e + 1] - off[e], h);
    for (int j = 0; j < 4; ++j) out_h[(uint64_t)e * 4 + j] = h[j];
    out_fp[e] = entry_fp(bytes + off[e], off[e + 1] - off[e], key);
}

// ---------------- union / rehash: every occupied slot of src[item] is inserted into dst[item] ----------------
struct UnionItem { uint32_t src, dst; };

__global__ __launch_bounds__(256) void k_ingest_union(const IngestTable *src_tables, const IngestTable *dst_tables,
                                                      const UnionItem *items, uint32_t *dst_counts, uint32_t *dst_status)
{
    __shared__ uint32_t wg_fresh;
    __shared__ uint32_t queue[4][256];                   // per wave: the occupied slots of its 4 x 64 scanned ones
    if (threadIdx.x == 0) wg_fresh = 0;
    __syncthreads();
    const UnionItem it = items[blockIdx.y];
    const IngestTable s = src_tables[it.src];
    const IngestTable d = dst_tables[it.dst];
    const uint64_t cap = (uint64_t)s.mask + 1;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t *q = queue[threadIdx.x >> 6];
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    uint32_t fresh = 0;
    // occupied <=> fingerprint != 0 (the walk is over): scan the 8-byte fingerprints, compact, insert densely — the source
    // tables are ~0.2 full, so walking their 32-byte slots moves 5x the bytes and keeps 1 lane in 6 busy in set_insert2
    for (uint64_t base = wave * 256; base < cap; base += n_waves * 256) {
        uint64_t f[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint64_t i = base + u * 64 + lane;
            f[u] = i < cap ? s.fps[i] : 0;
        }
        uint32_t n = 0;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint64_t mask = __ballot(f[u] != 0);
            if (f[u] != 0) q[n + lane_rank(mask)] = u * 64 + lane;
            n += (uint32_t)__builtin_popcountll(mask);
        }
        __builtin_amdgcn_wave_barrier();                 // a wave's LDS writes are visible to its own later reads (program order)
        for (uint32_t e = lane; e < n; e += 128) {
            const uint32_t e2 = e + 64;
            const bool act_b = e2 < n;
            const uint64_t i = base + q[e], j = base + (act_b ? q[e2] : q[e]);
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(s.slots + i * 4);
            const ulonglong2 *r = reinterpret_cast<const ulonglong2 *>(s.slots + j * 4);
            const ulonglong2 x0 = p[0], y0 = p[1], x1 = r[0], y1 = r[1];
            const uint64_t ha[4] = {x0.x, x0.y, y0.x, y0.y}, hb[4] = {x1.x, x1.y, y1.x, y1.y};
            const uint64_t fpa = s.fps[i], fpb = s.fps[j];
            bool pa, pb, fa, fb;
            set_insert2<false>(d, ha, fpa, true, nullptr, dst_status + it.dst, pa, d, hb, fpb, act_b, nullptr, dst_status + it.dst, pb, &fa, &fb);
            fresh += (uint32_t)fa + (uint32_t)fb;
        }
    }
    // one counter add per workgroup (a parent's counter is a single address for the whole launch)
    if (fresh) atomicAdd(&wg_fresh, fresh);
    __syncthreads();
    if (threadIdx.x == 0 && wg_fresh)
        __hip_atomic_fetch_add(dst_counts + it.dst, wg_fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------- the file-level union, partition by partition, deduplicated in LDS ----------------
// flush.go:221,253: the file's entry sets are the unions of its blocks' sets, and their exact sizes size the file-level
// filters.  Round 2 inserted every child entry into a global hash table per parent: 39 M CAS round trips at C3 (4.5 ms, all
// of it the CAS), tables of 2 x the children's entries rounded up to a power of two (5.4 GB), and a memset of all of it.
// Here a workgroup owns ONE PARTITION of one parent — the entries whose table_key starts with log2P given bits.  In every
// child table those entries sit in one run of slots (home slot = top bits of the key) plus whatever linear probing pushed
// behind it, so the workgroup walks that run in each child — lane = child — until the first empty slot at or past its end,
// keeps the entries whose key really is the partition's, and deduplicates them in an LDS table (exact: four hash words + the
// keyed fingerprint; equal hashes under different fingerprints flag the parent like the global tables do).  What is left is
// the partition's distinct entries: the workgroup adds their number to the parent's counter — ONE global atomic per workgroup,
// and the counter ends up as the exact distinct count — and writes them densely at the offset the add returned.  The parent
// is then a dense list (mask = count - 1, every fingerprint non-zero): k_build_sets, the binned build, k_table_compact and
// k_ingest_union read it like any table.  Memory: 40 bytes per CHILD entry (an upper bound of the union), no memset.
constexpr uint32_t kPartSlots = 2048;                 // LDS table of a partition: 80 KB of dynamic LDS, two workgroups per CU
constexpr uint32_t kPartFill = kPartSlots * 3 / 4;    // more distinct entries than this: the partitioning was too coarse (status overflow)
constexpr uint32_t kPartTarget = 1280;                // child entries per partition the host aims at (duplicates included): at C3 a partition is then
                                                      // 4 home slots per child = one 32-byte sector of fingerprints; finer partitions re-read sectors
constexpr uint32_t kPartThreads = 512;
constexpr uint32_t kPartProbes = 48;                  // linear-probe bound inside the LDS table
constexpr uint32_t kPartAhead = 4;                    // slots of a child a lane fetches per trip
constexpr uint32_t kPartLdsBytes = kPartSlots * 5 * 8;

struct PartParent {
    uint64_t *out_slots, *out_fps;   // dense output, room for cap_out entries
    uint32_t cap_out;
    uint32_t log2P;                  // partitions = 1 << log2P
    uint32_t child_begin, child_end; // this parent's children in child_list
    uint32_t table;                  // the parent's own table index: counts[table] / status[table]
    uint32_t pad;
};

__global__ __launch_bounds__(kPartThreads) void k_union_partitions(const IngestTable *tables, const uint32_t *child_list, const PartParent *parents,
                                                                   uint32_t *counts, uint32_t *status)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t part_lds[];
    uint64_t *L0 = part_lds, *L1 = L0 + kPartSlots, *L2 = L1 + kPartSlots, *L3 = L2 + kPartSlots, *LF = L3 + kPartSlots;
    __shared__ uint32_t n_fresh, out_base, cursor, overflow;
    const PartParent pp = parents[blockIdx.y];
    const uint32_t p = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
    if (p >= (1u << pp.log2P)) return;
    for (uint32_t i = tid; i < kPartSlots; i += kPartThreads) { L0[i] = 0; LF[i] = 0; }
    if (tid == 0) { n_fresh = 0; cursor = 0; overflow = 0; }
    __syncthreads();
    uint32_t fresh = 0;
    const uint32_t n_children = pp.child_end - pp.child_begin;
    for (uint32_t c0 = 0; c0 < n_children; c0 += kPartThreads) {
        const bool have = c0 + tid < n_children;
        IngestTable t{nullptr, nullptr, 0, 0};
        if (have) t = tables[child_list[pp.child_begin + c0 + tid]];
        const uint64_t cap = (uint64_t)t.mask + 1;
        // the partition's run of home slots in this child: [floor(p cap / P), ceil((p + 1) cap / P))
        uint64_t s = ((uint64_t)p * cap) >> pp.log2P;
        const uint64_t e = (((uint64_t)(p + 1) * cap) + ((1ull << pp.log2P) - 1)) >> pp.log2P;
        bool scanning = have;
        uint32_t scanned = 0;
        while (__ballot(scanning) != 0ull) {
            // ---- every lane looks at kPartAhead consecutive slots of its child: their fingerprints in one go, then the hashes of
            // the occupied ones together — two dependent round trips per trip instead of two per slot (a run is 1-2 home slots
            // plus the probe spill, so one trip nearly always finishes a child) ----
            if (scanning && *(volatile uint32_t *)&overflow) scanning = false;      // the partition is already known to be too coarse: stop reading
            uint64_t f[kPartAhead];
            ulonglong2 hx[kPartAhead], hy[kPartAhead];
            bool cand[kPartAhead];
#pragma unroll
            for (uint32_t u = 0; u < kPartAhead; ++u) f[u] = scanning ? t.fps[(uint32_t)(s + u) & t.mask] : 0;     // (the last partition's spill wraps to slot 0)
            // a slot past the run's end can only hold one of ours if every slot from the run's last one up to it is occupied
            // (linear probing leaves no gap between an entry's home and its place): the others' hashes are not even fetched,
            // and the first gap at or after the run's last slot ends the child
            bool chain = true;
            bool look[kPartAhead];
#pragma unroll
            for (uint32_t u = 0; u < kPartAhead; ++u) {
                hx[u] = hy[u] = make_ulonglong2(0, 0);
                look[u] = scanning && f[u] != 0 && (s + u < e || chain);
                if (look[u]) {
                    const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(t.slots + (uint64_t)((uint32_t)(s + u) & t.mask) * 4);
                    hx[u] = q[0]; hy[u] = q[1];
                }
                if (s + u + 1 >= e && f[u] == 0) chain = false;   // (from the run's last slot on)
            }
#pragma unroll
            for (uint32_t u = 0; u < kPartAhead; ++u) {
                cand[u] = false;
                if (look[u]) {
                    const uint32_t key = (uint32_t)(hx[u].y >> 20);                   // table_key
                    cand[u] = pp.log2P == 0 || (key >> (32u - pp.log2P)) == p;
                }
            }
            scanned += kPartAhead;
            if (scanned > cap) scanning = false;                                      // a table without an empty slot: one lap
            s += kPartAhead;
            if (s >= e && !chain) scanning = false;                   // the run is done and there is a gap at or behind its last slot: nothing of ours lies beyond
            // ---- the candidates of the wave go into the LDS table together (wave-uniform exit: see set_insert2's SIMT note) ----
#pragma unroll
            for (uint32_t u = 0; u < kPartAhead; ++u) {
                if (__ballot(cand[u]) == 0ull) continue;
                const uint64_t h0 = hx[u].x, h1 = hx[u].y, h2 = hy[u].x, h3 = hy[u].y, fp = f[u];
                bool done = !cand[u];
                uint32_t idx = (uint32_t)(h2 >> 17) & (kPartSlots - 1), probes = 0, spins = 0;
                do {
                    if (!done) {
                        unsigned long long old = 0;
                        const bool won = __hip_atomic_compare_exchange_strong((unsigned long long *)&L0[idx], &old, (unsigned long long)h0, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                              __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (won) {
                            L1[idx] = h1; L2[idx] = h2; L3[idx] = h3;
                            __hip_atomic_store(&LF[idx], fp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            done = true;
                            ++fresh;
                        } else if (old == h0) {
                            const uint64_t f2 = __hip_atomic_load(&LF[idx], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (f2 == 0) {                            // the claimer's words are still in flight
                                if (++spins > kSpinLimit) { __hip_atomic_fetch_max(status + pp.table, kTableExotic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); done = true; }
                            } else if (L1[idx] == h1 && L2[idx] == h2 && L3[idx] == h3) {
                                if (f2 != fp) __hip_atomic_fetch_max(status + pp.table, kTableExotic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // equal hashes, another entry
                                done = true;                          // a duplicate
                            } else {
                                idx = (idx + 1) & (kPartSlots - 1); ++probes;
                            }
                        } else {
                            idx = (idx + 1) & (kPartSlots - 1); ++probes;
                        }
                        if (!done && probes >= kPartProbes) { overflow = 1; done = true; }     // a chain this long: the table is as good as full
                    }
                } while (__ballot(!done) != 0ull);
            }
        }
    }
    if (fresh) atomicAdd(&n_fresh, fresh);
    __syncthreads();
    if (n_fresh > kPartFill) overflow = 1;
    __syncthreads();
    if (overflow) {                                               // too coarse a partitioning: the host repeats the launch with more partitions
        if (tid == 0) __hip_atomic_fetch_max(status + pp.table, kTableOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (tid == 0 && n_fresh) out_base = __hip_atomic_fetch_add(counts + pp.table, n_fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (n_fresh == 0) return;
    for (uint32_t i0 = 0; i0 < kPartSlots; i0 += kPartThreads) {
        const uint32_t i = i0 + tid;
        const uint64_t f = LF[i];
        const uint64_t mask = __ballot(f != 0);
        if (mask == 0) continue;
        uint32_t at = 0;
        if (lane == (uint32_t)__builtin_ctzll(mask)) at = atomicAdd(&cursor, (uint32_t)__builtin_popcountll(mask));
        at = __shfl(at, (int)__builtin_ctzll(mask), 64);
        if (f != 0) {
            const uint32_t pos = out_base + at + lane_rank(mask);
            if (pos < pp.cap_out) {
                ulonglong2 *o = reinterpret_cast<ulonglong2 *>(pp.out_slots + (uint64_t)pos * 4);
                o[0] = make_ulonglong2(L0[i], L1[i]);
                o[1] = make_ulonglong2(L2[i], L3[i]);
                pp.out_fps[pos] = f;
            }
        }
    }
}

// ---------------- occupied slots of a table, densely packed (the form a partial file-level set travels between devices in) ----------------
// out_slots[pos][4] / out_fps[pos] for pos < *counter: order is whatever the waves' reservations make it (a set has none).
// A partial parent table is ~0.25 full and 40 bytes per slot: what crosses xGMI is count x 40 bytes instead of capacity x 40.
__global__ __launch_bounds__(256) void k_table_compact(const IngestTable t, uint64_t *out_slots, uint64_t *out_fps, uint32_t *counter, uint32_t cap_out)
{
    const uint64_t cap = (uint64_t)t.mask + 1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    for (uint64_t base = wave * 64; base < cap; base += n_waves * 64) {
        const uint64_t i = base + lane;
        const uint64_t f = i < cap ? t.fps[i] : 0;
        const uint64_t mask = __ballot(f != 0);
        if (mask == 0) continue;
        uint32_t at = 0;
        if (lane == (uint32_t)__builtin_ctzll(mask)) at = __hip_atomic_fetch_add(counter, (uint32_t)__builtin_popcountll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        at = __shfl(at, (int)__builtin_ctzll(mask), 64);
        if (f != 0) {
            const uint32_t pos = at + lane_rank(mask);
            if (pos < cap_out) {
                const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(t.slots + i * 4);
                ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out_slots + (uint64_t)pos * 4);
                o[0] = p[0]; o[1] = p[1];
                out_fps[pos] = f;
            }
        }
    }
}

// ---------------- bitsets from distinct sets ----------------
struct SetBuildItem {
    uint32_t table;       // source table == filter index (desc[table])
    uint32_t staged;      // 1: whole bitset assembled in LDS by this workgroup
    uint64_t slot_begin, slot_end;
};

struct SetBuildArgs {
    const IngestTable *tables;
    const SetBuildItem *items;
    const DevDesc *desc;
    uint64_t *out;
};

constexpr uint32_t kBuildSetsThreads = 1024;   // one workgroup per table: 16 waves keep more slot loads in flight than 8 (1.16 -> see profiles)

// A table is mostly empty slots (sized for 4 entries per row, filled to ~0.2): walking the 32-byte slots themselves
// moves 5x the bytes that matter and leaves 5 of 6 lanes idle in the Barrett / atomic part.  So a tile of slots is
// scanned through the 8-byte fingerprints (non-zero <=> occupied once the walk has ended), the occupied indices are
// compacted into an LDS list (ballot + mbcnt, one LDS add per wave), and the list is worked off densely.
constexpr uint32_t kSetTile = 4 * kBuildSetsThreads;            // slots scanned per trip
constexpr uint32_t kSetListBytes = kSetTile * 4 + 16;            // static LDS of k_build_sets beside the staged bitset

template <int MODE, typename BITS32>
__device__ __forceinline__ void build_from_slots(const IngestTable t, const SetBuildItem &it, const DevDesc &d, BITS32 bits, uint32_t tid,
                                                 uint32_t *list, uint32_t *n_list)
{
    ModF64 fm{};
    if (MODE == kModFp64) fm = make_modf64(d.m, d.magic);
    for (uint64_t base = it.slot_begin; base < it.slot_end; base += kSetTile) {
        if (tid == 0) *n_list = 0;
        __syncthreads();
        uint64_t f[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint64_t i = base + u * kBuildSetsThreads + tid;
            f[u] = i < it.slot_end ? t.fps[i] : 0;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint64_t mask = __ballot(f[u] != 0);
            if (mask == 0) continue;
            uint32_t at = 0;
            if ((tid & 63u) == 0u) at = atomicAdd(n_list, (uint32_t)__builtin_popcountll(mask));
            at = __builtin_amdgcn_readfirstlane(at);
            if (f[u] != 0) list[at + lane_rank(mask)] = u * kBuildSetsThreads + tid;
        }
        __syncthreads();
        const uint32_t n = *n_list;
        // two entries per trip, all four 16-byte halves loaded before the first is used
        for (uint32_t e = tid; e < n; e += 2 * kBuildSetsThreads) {
            const uint32_t e2 = e + kBuildSetsThreads;
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(t.slots + (base + list[e]) * 4);
            const ulonglong2 x0 = p[0], y0 = p[1];
            ulonglong2 x1 = make_ulonglong2(0, 0), y1 = x1;
            if (e2 < n) {
                const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(t.slots + (base + list[e2]) * 4);
                x1 = q[0]; y1 = q[1];
            }
            set_entry_bits<MODE>(bits, d, fm, x0.x, x0.y, y0.x, y0.y);
            if (e2 < n) set_entry_bits<MODE>(bits, d, fm, x1.x, x1.y, y1.x, y1.y);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBuildSetsThreads) void k_build_sets(const SetBuildArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    __shared__ uint32_t list[kSetTile];
    __shared__ uint32_t n_list;
    const SetBuildItem it = a.items[blockIdx.x];
    const DevDesc d = a.desc[it.table];
    const IngestTable t = a.tables[it.table];
    const uint32_t tid = threadIdx.x;
    if (d.m == 0) return;
    const uint64_t nw = (d.m + 63) >> 6;
    const int mode = mod_mode(d.m);
    if (it.staged) {
        for (uint32_t i = tid; i < nw; i += kBuildSetsThreads) lds64[i] = 0;
        lds_u32 *bits = (lds_u32 *)lds64;                        // (build_from_slots starts with a barrier)
        if (mode == kModFp64)           build_from_slots<kModFp64>(t, it, d, bits, tid, list, &n_list);
        else if (mode == kModBarrett32) build_from_slots<kModBarrett32>(t, it, d, bits, tid, list, &n_list);
        else                            build_from_slots<kModBarrett64>(t, it, d, bits, tid, list, &n_list);
        uint64_t *dst = a.out + d.word_off;
        for (uint32_t i = tid; i < nw; i += kBuildSetsThreads) dst[i] = lds64[i];
    } else {
        uint32_t *bits = reinterpret_cast<uint32_t *>(a.out + d.word_off);
        if (mode == kModBarrett64) build_from_slots<kModBarrett64>(t, it, d, bits, tid, list, &n_list);
        else                       build_from_slots<kModBarrett32>(t, it, d, bits, tid, list, &n_list);
    }
}

}  // namespace bsg
